// TEST INFRASTRUCTURE ONLY (oracle/ref_eigen): stand-in so that the reference's sources compile without ROS.  common_lib.h:13
// includes tf/transform_broadcaster.h and uses nothing of it; the Mode-18 loop calls tf::createQuaternionMsgFromRollPitchYaw
// once (laserMapping.cpp:1718) into a message that only the ROS publishers read.
#pragma once
#include <cmath>
namespace geometry_msgs { struct Quaternion { double x, y, z, w; }; }
namespace tf {
inline geometry_msgs::Quaternion createQuaternionMsgFromRollPitchYaw(double roll, double pitch, double yaw)
{
    const double cr = std::cos(roll * 0.5), sr = std::sin(roll * 0.5), cp = std::cos(pitch * 0.5), sp = std::sin(pitch * 0.5);
    const double cy = std::cos(yaw * 0.5), sy = std::sin(yaw * 0.5);
    geometry_msgs::Quaternion q;
    q.x = sr * cp * cy - cr * sp * sy;
    q.y = cr * sp * cy + sr * cp * sy;
    q.z = cr * cp * sy - sr * sp * cy;
    q.w = cr * cp * cy + sr * sp * sy;
    return q;
}
}  // namespace tf
