// TEST INFRASTRUCTURE ONLY (oracle/ref_eigen): stand-in so that the reference's include/common_lib.h compiles without ROS.
// common_lib.h:148,186: deque<sensor_msgs::Imu::ConstPtr>, (*it)->header.stamp.toSec().
#pragma once
#include <memory>
namespace ros { struct Time { double t = 0.0; double toSec() const { return t; } }; }
namespace std_msgs { struct Header { ros::Time stamp; }; }
namespace sensor_msgs {
struct Imu {
    typedef std::shared_ptr<const Imu> ConstPtr;
    typedef std::shared_ptr<Imu> Ptr;
    std_msgs::Header header;
};
}  // namespace sensor_msgs
#ifndef ROS_WARN
#define ROS_WARN(...) ((void)0)
#endif
