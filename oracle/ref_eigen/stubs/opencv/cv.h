// TEST INFRASTRUCTURE ONLY (oracle/ref_eigen): stand-in so that the reference's include/common_lib.h compiles without OpenCV (IKFoM_toolkit/mtk/types/wrapped_cv_mat.hpp:41; not included by use-ikfom.hpp, kept for completeness).
#pragma once
#include <opencv2/opencv.hpp>
