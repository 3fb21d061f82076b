// TEST INFRASTRUCTURE ONLY (oracle/ref_eigen): stand-in so that the reference's sources compile without PCL.
// common_lib.h:54,56,163,172,193 use pcl::PointCloud<T>, its ::Ptr (reset(new ...)) and ->points (size(), back()); the Mode-18
// loop (laserMapping.cpp:1506-1732) additionally swaps a cloud with an empty one and calls resize / reserve on it; UndistortPcl
// (IMU_Processing.cpp:611-809) clears one and push_backs into it.
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
    void push_back(const PointT &p) { points.push_back(p); }
    void resize(std::size_t n) { points.resize(n); }
    void reserve(std::size_t n) { points.reserve(n); }
    void swap(PointCloud &o) { points.swap(o.points); }
};
}  // namespace pcl
