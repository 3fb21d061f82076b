// TEST INFRASTRUCTURE ONLY (oracle/ref_eigen): stand-in so that the reference's include/common_lib.h compiles without OpenCV.
// common_lib.h:149,202 hold cv::Mat members; nothing on the pinned functions touches them.
#pragma once
namespace cv { struct Mat {}; }
