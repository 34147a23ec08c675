// Frame_hot.cc -- replacement DEFINITIONS of the two stereo matchers of VIEO_SLAM::Frame on top of the C-ABI of
// libvieo_hot.so.  Compiled inside the reference tree against the reference's own include/Frame.h (declarations
// untouched); the same two members are compiled out of src/Frame.cc with `#ifndef VIEO_HOT`.
// Frame::Frame (src/Frame.cc:218-320) calls them unchanged, right after the per-camera ExtractORB threads.
//
//   void Frame::ComputeStereoMatches()                                    src/Frame.cc:451-611
//   void Frame::ComputeStereoFishEyeMatches(const float th_far_pts)       src/Frame.cc:613-779
//
// The extractor handles the frame points at (mpORBextractors[c], include/vieo_shim.hpp) still hold the keys,
// descriptors and pyramids they have just produced: the rectified matcher reads them there (the resident form: nothing
// goes up, uright / depth come back, uright stays in HBM for the frame's projection searches).  A frame whose handles
// hold something else by now -- never the case inside Frame::Frame -- is served by the host-pointer entry.
// No CPU fallback: a failing C-ABI call aborts like the reference's CV_Assert.
#include "Frame.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "vieo_flatten.hpp"

namespace VIEO_SLAM {

namespace {

static_assert(sizeof(cv::KeyPoint) == sizeof(vieo_keypoint), "cv::KeyPoint layout (28 bytes) is the C-ABI's");

[[noreturn]] void hot_fail(const char* what, int rc) {
  std::fprintf(stderr, "vieo_hot: %s failed (%d): %s\n", what, rc, vieo_last_error());
  std::abort();
}
#define HOT_CHECK(call)                       \
  do {                                        \
    const int rc_ = (call);                   \
    if (rc_ != VIEO_OK) hot_fail(#call, rc_); \
  } while (0)

inline const vieo_keypoint* kp(const std::vector<cv::KeyPoint>& v) { return reinterpret_cast<const vieo_keypoint*>(v.data()); }

}  // namespace

// ---------------------------------------------------------------- src/Frame.cc:451-611
void Frame::ComputeStereoMatches() {
  stereoinfo_.vuright_ = vector<float>(N, -1.0f);
  stereoinfo_.vdepth_ = vector<float>(N, -1.0f);
  if (N <= 0) return;
  vieo_orb* hl = mpORBextractors[0]->handle();
  vieo_orb* hr = mpORBextractors[1]->handle();
  const std::vector<cv::KeyPoint>& kl = vvkeys_[0];
  const std::vector<cv::KeyPoint>& kr = vvkeys_[1];
  const float baseline = stereoinfo_.baseline_bf_[0], bf = stereoinfo_.baseline_bf_[1];
  if (vieo_orb_holds(hl, kp(kl), (int)kl.size()) && vieo_orb_holds(hr, kp(kr), (int)kr.size())) {
    HOT_CHECK(vieo_stereo_match_rectified_resident(hl, hr, baseline, bf, stereoinfo_.vuright_.data(), stereoinfo_.vdepth_.data()));
    return;
  }
  HOT_CHECK(vieo_stereo_match_rectified(hl, hr, kp(kl), mDescriptors.ptr<unsigned char>(0), (int)kl.size(), kp(kr),
                                        kr.empty() ? nullptr : vdescriptors_[1].ptr<unsigned char>(0), (int)kr.size(), baseline, bf,
                                        stereoinfo_.vuright_.data(), stereoinfo_.vdepth_.data()));
}

// ---------------------------------------------------------------- src/Frame.cc:613-779
void Frame::ComputeStereoFishEyeMatches(const float th_far_pts) {
  const size_t n_cams = vvkeys_.size();
  if (n_cams < 2 || n_cams > 4 || mpCameras.size() < n_cams) hot_fail("ComputeStereoFishEyeMatches: 2..4 cameras", VIEO_E_INVALID);
  // ---- the matches, the groups (mvidxsMatches / goodmatches_ / mapcamidx2idxs_ / v3dpoints_) and the keys' depths: one call
  vieo_fisheye_params P;
  std::memset(&P, 0, sizeof(P));
  vieo_camera cams[4];
  double Trc[4][12], Tcr[4][12];
  const Eigen::Matrix3d I = Eigen::Matrix3d::Identity();
  const Eigen::Vector3d z = Eigen::Vector3d::Zero();
  const vieo_keypoint* keys[4] = {nullptr, nullptr, nullptr, nullptr};
  const uint8_t* descs[4] = {nullptr, nullptr, nullptr, nullptr};
  int32_t n_keys[4] = {0, 0, 0, 0}, mono[4] = {0, 0, 0, 0};
  size_t n_total = 0;
  for (size_t c = 0; c < n_cams; ++c) {
    if (!vieo_shim::to_pod(mpCameras[c].get(), I, z, cams[c])) hot_fail("ComputeStereoFishEyeMatches: camera model", VIEO_E_INVALID);
    vieo_shim::se3_to_3x4(mpCameras[c]->GetTrc(), Trc[c]);
    vieo_shim::se3_to_3x4(mpCameras[c]->GetTcr(), Tcr[c]);
    keys[c] = kp(vvkeys_[c]);
    descs[c] = vvkeys_[c].empty() ? nullptr : vdescriptors_[c].ptr<unsigned char>(0);
    n_keys[c] = (int32_t)vvkeys_[c].size(), mono[c] = (int32_t)num_mono[c];
    n_total += vvkeys_[c].size();
  }
  P.n_cams = (int32_t)n_cams, P.n_levels = (int32_t)scalepyrinfo_.vlevelsigma2_.size();
  P.bf = stereoinfo_.baseline_bf_[1], P.th_far_pts = th_far_pts;
  P.cams = cams, P.Trc = &Trc[0][0], P.Tcr = &Tcr[0][0], P.level_sigma2 = scalepyrinfo_.vlevelsigma2_.data();
  const int32_t gcap = (int32_t)std::max<size_t>(n_total, 1);
  std::vector<float> depth(std::max<size_t>(n_total, 1), -1.f);
  std::vector<int32_t> key_group(std::max<size_t>(n_total, 1), -1), group_idx((size_t)gcap * n_cams, -1);
  std::vector<uint8_t> good(gcap, 0);
  std::vector<double> p3d((size_t)gcap * 3, 0.0);
  int32_t n_groups = 0, n_matches = 0;
  if (n_total > 0)
    HOT_CHECK(vieo_stereo_fisheye_match(&P, keys, descs, n_keys, mono, gcap, depth.data(), key_group.data(), group_idx.data(),
                                        good.data(), p3d.data(), &n_groups, &n_matches));
  mvidxsMatches.assign(n_groups, vector<size_t>(n_cams, (size_t)-1));
  stereoinfo_.goodmatches_.assign(n_groups, false);
  stereoinfo_.v3dpoints_.resize(n_groups);
  stereoinfo_.mapcamidx2idxs_.clear();
  for (int32_t g = 0; g < n_groups; ++g) {
    for (size_t c = 0; c < n_cams; ++c) {
      const int32_t k = group_idx[(size_t)g * n_cams + c];
      if (k >= 0) mvidxsMatches[g][c] = (size_t)k;
    }
    stereoinfo_.goodmatches_[g] = good[g] != 0;
    for (int r = 0; r < 3; ++r) stereoinfo_.v3dpoints_[g](r) = p3d[(size_t)g * 3 + r];
  }
  // ---- mvKeys / mDescriptors / vdepth_ / vuright_ / mapn2in_ in camera-major order, and the maps back (:738-778)
  size_t n = 0;
  for (size_t c = 0; c < n_cams; ++c) {
    if (vdescriptors_.size() <= c || vdescriptors_[c].empty()) {
      n += vvkeys_[c].size();  // (an extractor never returns keys without descriptors; keeps the flat index in step)
      continue;
    }
    if (mDescriptors.empty())
      mDescriptors = vdescriptors_[c].clone();
    else
      cv::vconcat(mDescriptors, vdescriptors_[c], mDescriptors);
    for (size_t k = 0; k < vvkeys_[c].size(); ++k, ++n) {
      const std::pair<size_t, size_t> camidx(c, k);
      if (key_group[n] >= 0) stereoinfo_.mapcamidx2idxs_[camidx] = (size_t)key_group[n];
      stereoinfo_.vdepth_.push_back(depth[n]);
      stereoinfo_.vuright_.push_back(-1);
      mvKeys.push_back(vvkeys_[c][k]);
      mapn2in_.push_back(camidx);
    }
  }
  if (mapin2n_.size() < n_cams) mapin2n_.resize(n_cams);
  mapidxs2n_.resize(stereoinfo_.v3dpoints_.size(), (size_t)-1);
  for (size_t k = 0; k < mvKeys.size(); ++k) {
    const size_t cami = std::get<0>(mapn2in_[k]);
    if (mapin2n_[cami].size() < vvkeys_[cami].size()) mapin2n_[cami].resize(vvkeys_[cami].size());
    mapin2n_[cami][std::get<1>(mapn2in_[k])] = k;
    auto it = stereoinfo_.mapcamidx2idxs_.find(mapn2in_[k]);
    if (it != stereoinfo_.mapcamidx2idxs_.end()) mapidxs2n_[it->second] = k;
  }
  N = (int)mvKeys.size();
}

}  // namespace VIEO_SLAM
