// TEST INFRASTRUCTURE ONLY (oracle/ref_eigen): stand-in so that the reference's include/common_lib.h compiles without ROS (eigen_conversions/eigen_msg.h is included by common_lib.h:9-16 and nothing of it is used there).
#pragma once
