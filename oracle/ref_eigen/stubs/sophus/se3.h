// TEST INFRASTRUCTURE ONLY (oracle/ref_eigen): stand-in so that the reference's sources compile without Sophus
// (common_lib.h:17,23: `using namespace Sophus;`).  The pinned text uses SE3(R, t) (lidar_selection.cpp:910), T * p, T * T,
// T.inverse() and translation() (:537-538, frame.h:89,98,107, feature.h:58).  Stated with rotation matrices as oracle/orc_select.c
// states them; Sophus a621ff keeps a unit quaternion: its results agree with these to rounding, not bitwise.
#pragma once
#include <Eigen/Core>
namespace Sophus {
class SE3 {
public:
    SE3() : R_(Eigen::Matrix3d::Identity()), t_(0, 0, 0) {}
    SE3(const Eigen::Matrix3d &R, const Eigen::Vector3d &t) : R_(R), t_(t) {}
    const Eigen::Matrix3d &rotation_matrix() const { return R_; }
    const Eigen::Vector3d &translation() const { return t_; }
    Eigen::Vector3d operator*(const Eigen::Vector3d &p) const { return R_ * p + t_; }
    SE3 operator*(const SE3 &o) const { return SE3(R_ * o.R_, R_ * o.t_ + t_); }
    SE3 inverse() const { const Eigen::Matrix3d Rt = R_.transpose(); return SE3(Rt, Rt * (t_ * -1.)); }
private:
    Eigen::Matrix3d R_;
    Eigen::Vector3d t_;
};
}  // namespace Sophus
