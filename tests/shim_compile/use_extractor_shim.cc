#include "vieo_shim.hpp"
#ifndef VIEO_SHIM_HAVE_OPENCV
#error "mock OpenCV not found"
#endif
int use(VIEO_SLAM::ORBextractor& e, cv::Mat& im, std::vector<cv::KeyPoint>& k, cv::Mat& d) { return e(im, cv::Mat(), k, d) + e.GetLevels(); }
