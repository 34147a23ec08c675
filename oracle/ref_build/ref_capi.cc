// TEST INFRASTRUCTURE -- part of the pinning recipe (oracle/ref_build/CMakeLists.txt), never of the product.
// C hooks with the names and signatures of the restated oracle's (oracle/orb_extractor.cc: vo_orb_*), implemented on
// the REFERENCE's own VIEO_SLAM::ORBextractor (include/ORBextractor.h:27-80, src/ORBextractor.cc) and a real OpenCV.
// Only what the real class exposes is offered: the extraction itself, the public pyramid, the scale tables.
// Not compiled in the authoring image (no OpenCV); see the CMake file's header.
#include <cstdint>
#include <cstring>
#include <vector>

#include <opencv2/core/core.hpp>

#include "ORBextractor.h"

namespace {
struct RefKeyPoint {  // = vieo_keypoint = the fields of cv::KeyPoint in declaration order (28 bytes)
  float x, y, size, angle, response;
  int32_t octave, class_id;
};
struct Handle {
  VIEO_SLAM::ORBextractor ex;
  int nlevels;
  Handle(int nf, float sf, int nl, int ini, int mn) : ex(nf, sf, nl, ini, mn), nlevels(nl) {}
};
}  // namespace

extern "C" {

void* vo_orb_create(int nfeatures, float scale, int nlevels, int ini, int mn) {
  return new Handle(nfeatures, scale, nlevels, ini, mn);
}
void vo_orb_destroy(void* h) { delete (Handle*)h; }

// returns the mono index (operator()'s return value), -1 on an empty image, -2 when cap is too small
int vo_orb_extract(void* h, const uint8_t* img, int w, int hgt, int stride, const int* lapping, void* kps_out,
                   uint8_t* desc_out, int cap, int* n) {
  Handle* e = (Handle*)h;
  cv::Mat im(hgt, w, CV_8UC1, (void*)img, (size_t)stride);
  std::vector<cv::KeyPoint> kps;
  cv::Mat desc;
  std::vector<int> lap;
  if (lapping) lap = {lapping[0], lapping[1]};
  const int mono = e->ex(im, cv::Mat(), kps, desc, lapping ? &lap : nullptr);
  if (mono < 0) return -1;
  *n = (int)kps.size();
  if ((int)kps.size() > cap) return -2;
  RefKeyPoint* out = (RefKeyPoint*)kps_out;
  for (size_t i = 0; i < kps.size(); i++) {
    const cv::KeyPoint& k = kps[i];
    out[i] = RefKeyPoint{k.pt.x, k.pt.y, k.size, k.angle, k.response, k.octave, k.class_id};
    std::memcpy(desc_out + 32 * i, desc.ptr<uint8_t>((int)i), 32);
  }
  return mono;
}

float vo_orb_scale_factor(void* h, int level) { return ((Handle*)h)->ex.GetScaleFactors()[level]; }

void vo_orb_level_size(void* h, int level, int* w, int* hgt) {
  const cv::Mat& m = ((Handle*)h)->ex.mvImagePyramid[level];
  *w = m.cols, *hgt = m.rows;
}

// which: 0 = pyramid level as Frame::ComputeStereoMatches reads it (the ROI view); others are not public in the
// reference and return 0.  out: w x h bytes, tightly packed
int vo_orb_get_plane(void* h, int level, int which, uint8_t* out) {
  if (which != 0) return 0;
  const cv::Mat& m = ((Handle*)h)->ex.mvImagePyramid[level];
  for (int y = 0; y < m.rows; y++) std::memcpy(out + (size_t)y * m.cols, m.ptr<uint8_t>(y), m.cols);
  return 1;
}

}  // extern "C"
