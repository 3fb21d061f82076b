// TEST INFRASTRUCTURE ONLY (oracle/ref_eigen): stand-in so that the reference's sources compile without ROS.
// common_lib.h:148,186: deque<sensor_msgs::Imu::ConstPtr>, (*it)->header.stamp.toSec(); ImuProcess::UndistortPcl
// (IMU_Processing.cpp:611-809) additionally reads angular_velocity.{x,y,z} and linear_acceleration.{x,y,z} (geometry_msgs/Vector3:
// three float64).
#pragma once
#include <memory>
namespace ros { struct Time { double t = 0.0; double toSec() const { return t; } }; struct NodeHandle {}; }
namespace std_msgs { struct Header { ros::Time stamp; }; }
namespace geometry_msgs { struct Vector3 { double x = 0.0, y = 0.0, z = 0.0; }; }
namespace sensor_msgs {
struct Imu {
    typedef std::shared_ptr<const Imu> ConstPtr;
    typedef std::shared_ptr<Imu> Ptr;
    std_msgs::Header header;
    geometry_msgs::Vector3 angular_velocity, linear_acceleration;
};
typedef Imu::ConstPtr ImuConstPtr;
}  // namespace sensor_msgs
#ifndef ROS_WARN
#define ROS_WARN(...) ((void)0)
#endif
