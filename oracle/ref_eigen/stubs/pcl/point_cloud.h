// TEST INFRASTRUCTURE ONLY (oracle/ref_eigen): stand-in so that the reference's include/common_lib.h compiles without PCL.
// common_lib.h:54,56,163,172,193 use pcl::PointCloud<T>, its ::Ptr (reset(new ...)) and ->points (size(), back()).
#pragma once
#include <memory>
#include <vector>
#include <Eigen/Core>
namespace pcl {
template <typename PointT>
struct PointCloud {
    typedef std::shared_ptr<PointCloud<PointT>> Ptr;
    typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
    std::vector<PointT, Eigen::aligned_allocator<PointT>> points;
    std::size_t size() const { return points.size(); }
    void clear() { points.clear(); }
};
}  // namespace pcl
