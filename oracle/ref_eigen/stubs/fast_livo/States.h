// TEST INFRASTRUCTURE ONLY (oracle/ref_eigen): stand-in so that the reference's include/common_lib.h compiles without ROS message generation (fast_livo/States is included at common_lib.h:8 and not used by it).
#pragma once
namespace fast_livo { struct States {}; }
