// vieo_shim.hpp -- VIEO_SLAM::ORBextractor with the reference's public interface
// (include/ORBextractor.h:27-80) implemented on the C-ABI of vieo_hot.h, so src/Frame.cc and
// src/Tracking.cc compile and behave unchanged when this header replaces include/ORBextractor.h
// and libvieo_hot.so replaces src/ORBextractor.cc.  Needs OpenCV headers (cv::Mat, cv::KeyPoint);
// where they are absent (this repository's image) the file compiles to nothing.
// ORBmatcher / Optimizer forwarding is shown in INTEGRATION.md.
#pragma once
#if defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>)
#define VIEO_SHIM_HAVE_OPENCV 1
#endif
#endif

#ifdef VIEO_SHIM_HAVE_OPENCV
#include <opencv2/core/core.hpp>

#include <cassert>
#include <stdexcept>
#include <vector>

#include "vieo_hot.h"

namespace VIEO_SLAM {

class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
      : nlevels_(nlevels), nfeatures_(nfeatures), ini_th_(iniThFAST), min_th_(minThFAST) {
    if (vieo_orb_create(&h_, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) != VIEO_OK)
      throw std::runtime_error(vieo_last_error());  // no CPU fallback
    mvImagePyramid.resize(nlevels);
  }
  ~ORBextractor() { vieo_orb_destroy(h_); }
  ORBextractor(const ORBextractor&) = delete;
  ORBextractor& operator=(const ORBextractor&) = delete;

  // ORBextractor.cc:968-1058; mask is ignored as in the reference
  int operator()(cv::InputArray _image, cv::InputArray, std::vector<cv::KeyPoint>& keypoints,
                 cv::OutputArray descriptors, const std::vector<int>* pvLappingArea = nullptr) {
    if (_image.empty()) return -1;
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    if (deferred_) {  // shim/Tracking_hot.cc: the frame's extraction happens inside vieo_track_frame
      deferred_image_ = image;
      keypoints.clear();
      descriptors.release();
      return 0;
    }
    const int cap = vieo_orb_max_keypoints(h_);
    static_assert(sizeof(cv::KeyPoint) == sizeof(vieo_keypoint), "cv::KeyPoint layout");
    keypoints.resize(cap);
    cv::Mat desc(cap, 32, CV_8U);
    int n = 0, mono = 0;
    const int rc = vieo_orb_extract(h_, image.data, image.cols, image.rows, (int)image.step,
                                    pvLappingArea ? pvLappingArea->data() : nullptr,
                                    reinterpret_cast<vieo_keypoint*>(keypoints.data()), desc.data, cap,
                                    &n, &mono);
    if (rc == VIEO_E_EMPTY) return -1;
    if (rc != VIEO_OK) throw std::runtime_error(vieo_last_error());
    keypoints.resize(n);
    if (n == 0)
      descriptors.release();
    else
      desc.rowRange(0, n).copyTo(descriptors);
    // mvImagePyramid (ORBextractor.h:54) is read by Frame::ComputeStereoMatches only (src/Frame.cc:457,536-557), and
    // shim/Frame_hot.cc runs that on the device where the planes already are: the eight bordered planes (1.4 MB for a
    // 752 x 480 image) are copied back only on request -- FetchImagePyramid(), or always with VIEO_SHIM_EAGER_PYRAMID.
    // CONTRACT CHANGE against the reference's operator(): after the call mvImagePyramid holds EMPTY cv::Mat's unless one
    // of the two is used.  No other reader exists in the reference today (grep mvImagePyramid: ORBextractor.cc itself and
    // the two places in Frame.cc named above); code added later that reads the member must ask for the planes.
#ifdef VIEO_SHIM_EAGER_PYRAMID
    FetchImagePyramid();
#else
    for (auto& m : mvImagePyramid) m = cv::Mat();
#endif
    return mono;
  }

  int inline GetLevels() { return vieo_orb_levels(h_); }
  float inline GetScaleFactor() { return vieo_orb_scale_factor(h_); }
  std::vector<float> inline GetScaleFactors() { return tab(vieo_orb_scale_factors); }
  std::vector<float> inline GetInverseScaleFactors() { return tab(vieo_orb_inv_scale_factors); }
  std::vector<float> inline GetScaleSigmaSquares() { return tab(vieo_orb_level_sigma2); }
  std::vector<float> inline GetInverseScaleSigmaSquares() { return tab(vieo_orb_inv_level_sigma2); }

  std::vector<cv::Mat> mvImagePyramid;
  // ROI views into bordered planes (ComputePyramid, ORBextractor.cc:1060-1081) of the last extraction
  void FetchImagePyramid() {
    for (int l = 0; l < nlevels_; ++l) {
      int w, h;
      vieo_orb_level_size(h_, l, &w, &h);
      cv::Mat temp(h + 38, w + 38, CV_8UC1);
      vieo_orb_get_level(h_, 0, l, 1, temp.data, (int)temp.step);
      mvImagePyramid[l] = temp(cv::Rect(19, 19, w, h));
    }
  }
  vieo_orb* handle() { return h_; }  // the resident frame: shim/Frame_hot.cc, shim/ORBmatcher_hot.cc
  // One call per frame (shim/Tracking_hot.cc): while deferred, operator() only keeps the image (a cv::Mat header) and
  // returns no keys -- Frame::Frame then stops after ExtractORB (src/Frame.cc:282 `if (!N) return;`) and the tracker
  // binding fills the frame from vieo_track_frame's outputs.
  // (the constructor's arguments, for the binding that creates a tracker with the same extractor)
  int HotFeatures() const { return nfeatures_; }
  int HotIniThFAST() const { return ini_th_; }
  int HotMinThFAST() const { return min_th_; }
  void Defer(bool on) { deferred_ = on; }
  bool Deferred() const { return deferred_; }
  const cv::Mat& DeferredImage() const { return deferred_image_; }

 private:
  std::vector<float> tab(int (*fn)(const vieo_orb*, float*)) {
    std::vector<float> v(nlevels_);
    fn(h_, v.data());
    return v;
  }
  vieo_orb* h_ = nullptr;
  int nlevels_, nfeatures_, ini_th_, min_th_;
  bool deferred_ = false;
  cv::Mat deferred_image_;
};

}  // namespace VIEO_SLAM
#endif  // VIEO_SHIM_HAVE_OPENCV
