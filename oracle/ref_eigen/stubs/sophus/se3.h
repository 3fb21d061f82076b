// TEST INFRASTRUCTURE ONLY (oracle/ref_eigen): stand-in so that the reference's include/common_lib.h compiles without Sophus
// (common_lib.h:17,23: `using namespace Sophus;`).  On the pinned path Sophus appears once: `SE3(Rcw, Pcw)` stored into
// new_frame_->T_f_w_ (lidar_selection.cpp:910) -- a rotation matrix and a translation kept as given.
#pragma once
#include <Eigen/Core>
namespace Sophus {
class SE3 {
public:
    SE3() : R_(Eigen::Matrix3d::Identity()), t_(0, 0, 0) {}
    SE3(const Eigen::Matrix3d &R, const Eigen::Vector3d &t) : R_(R), t_(t) {}
    const Eigen::Matrix3d &rotation_matrix() const { return R_; }
    const Eigen::Vector3d &translation() const { return t_; }
private:
    Eigen::Matrix3d R_;
    Eigen::Vector3d t_;
};
}  // namespace Sophus
