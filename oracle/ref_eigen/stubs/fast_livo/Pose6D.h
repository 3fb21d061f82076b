// TEST INFRASTRUCTURE ONLY (oracle/ref_eigen): stand-in so that the reference's include/common_lib.h compiles without ROS message generation.
// msg/Pose6D.msg of the reference: float64 offset_time, float64[3] acc gyr vel pos, float64[9] rot (set_pose6d, common_lib.h:395-412).
#pragma once
namespace fast_livo {
struct Pose6D {
    double offset_time;
    double acc[3], gyr[3], vel[3], pos[3], rot[9];
};
}  // namespace fast_livo
