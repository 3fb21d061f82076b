// TEST INFRASTRUCTURE ONLY (oracle/ref_eigen): stand-in so that the reference's sources compile without OpenCV.
// common_lib.h:149,202 hold cv::Mat members; the pinned text reads `img.data` as a row-major u8 buffer (lidar_selection.cpp:113,134,821),
// `cols` / `rows` (:289), copies Mats around (a header over shared pixels, as in OpenCV) and allocates one zeroed float image
// (`cv::Mat::zeros(height, width, CV_32FC1)`, :366).
#pragma once
#include <memory>
#include <vector>
// MAX is OpenCV's macro (opencv2/core/cvdef.h); IMU_Processing.cpp:621 uses it on two doubles and nothing else defines it.
#ifndef MAX
#define MAX(a, b) ((a) < (b) ? (b) : (a))
#endif
#define CV_8UC1 0
#define CV_32FC1 5
namespace cv {
struct Mat {
    unsigned char *data;
    int rows, cols;
    std::shared_ptr<std::vector<unsigned char>> own;      // set when the Mat allocated its pixels itself
    Mat() : data(nullptr), rows(0), cols(0) {}
    Mat(int r, int c, unsigned char *d) : data(d), rows(r), cols(c) {}
    static Mat zeros(int r, int c, int type)
    {
        Mat m;
        m.rows = r; m.cols = c;
        m.own.reset(new std::vector<unsigned char>((size_t)r * (size_t)c * (type == CV_32FC1 ? 4u : 1u), 0));
        m.data = m.own->data();
        return m;
    }
};
}  // namespace cv
