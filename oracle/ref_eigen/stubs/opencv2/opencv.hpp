// TEST INFRASTRUCTURE ONLY (oracle/ref_eigen): stand-in so that the reference's include/common_lib.h compiles without OpenCV.
// common_lib.h:149,202 hold cv::Mat members; LidarSelector::UpdateState reads `img.data` as a row-major u8 buffer
// (lidar_selection.cpp:821) and nothing else of it.
#pragma once
// MAX is OpenCV's macro (opencv2/core/cvdef.h); IMU_Processing.cpp:621 uses it on two doubles and nothing else defines it.
#ifndef MAX
#define MAX(a, b) ((a) < (b) ? (b) : (a))
#endif
namespace cv {
struct Mat {
    unsigned char *data;
    int rows, cols;
    Mat() : data(nullptr), rows(0), cols(0) {}
    Mat(int r, int c, unsigned char *d) : data(d), rows(r), cols(c) {}
};
}  // namespace cv
