// ORBmatcher_hot.cc -- replacement DEFINITIONS of the VIEO_SLAM::ORBmatcher members on the hot path, on top of the
// C-ABI of libvieo_hot.so.  Compiled inside the reference tree against the reference's own include/ORBmatcher.h
// (the class declaration, i.e. every call signature, is untouched); the same members are compiled out of
// src/ORBmatcher.cc.  Tracking.cc / LocalMapping.cc call them unchanged.
//
//   int SearchByProjection(Frame&, const Frame&, th, bMono, th_far_pts)                 ORBmatcher.cc:1303-1467
//   int SearchByProjection(Frame&, const vector<MapPoint*>&, th, th_far_pts)            ORBmatcher.cc:230-335
//   int SearchByProjection(Frame&, KeyFrame*, const set<MapPoint*>&, th, ORBdist, far)  ORBmatcher.cc:1471-1606
//   int SearchForTriangulation(KeyFrame*, KeyFrame*, vMatchedPairs, bOnlyStereo)        ORBmatcher.cc:896-1150
//   int Fuse(KeyFrame*, const vector<MapPoint*>&, th) via SearchByProjectionBase        ORBmatcher.cc:26-227, 1152-1165
//
// What stays on the host is exactly what mutates caller objects (AddMapPoint / EraseMapPointMatch / FuseMP), applied
// in the reference's order.  No CPU fallback: a failing C-ABI call aborts like the reference's CV_Assert.
#include "ORBmatcher.h"

#include <climits>
#include <cstdio>
#include <cstdlib>

#include "vieo_flatten.hpp"

namespace VIEO_SLAM {

namespace {

static_assert(sizeof(cv::KeyPoint) == sizeof(vieo_keypoint), "cv::KeyPoint layout (28 bytes) is the C-ABI's");

[[noreturn]] void hot_fail(const char* what, int rc) {
  std::fprintf(stderr, "vieo_hot: %s failed (%d): %s\n", what, rc, vieo_last_error());
  std::abort();  // the reference's error convention for programming errors (CV_Assert)
}
#define HOT_CHECK(call)                        \
  do {                                         \
    const int rc_ = (call);                    \
    if (rc_ != VIEO_OK) hot_fail(#call, rc_);  \
  } while (0)

// protected MapPoint::mfMaxDistance / mfMinDistance (PredictScale reads them): pointer-to-member formed in a derived
// class, which [class.protected] allows; its type is float MapPoint::*
struct MapPointAccess : public MapPoint {
  static float MapPoint::*max_distance() { return &MapPointAccess::mfMaxDistance; }
  static float MapPoint::*min_distance() { return &MapPointAccess::mfMinDistance; }
};

inline void copy_desc(const cv::Mat& d, uint8_t* out) { std::memcpy(out, d.ptr<unsigned char>(0), 32); }

void fill_sbp_camera(const Frame& F, float th, float th_far, bool mono, vieo_sbp_camera& C) {
  std::memset(&C, 0, sizeof(C));
  vieo_shim::se3_to_3x4(F.GetTcwCst(), C.Tcw_cur);
  C.bf = F.stereoinfo_.baseline_bf_[1], C.baseline = F.stereoinfo_.baseline_bf_[0];
  C.th = th, C.th_far = th_far, C.mono = mono ? 1 : 0;
  C.nlevels = (int)F.scalepyrinfo_.vscalefactor_.size();
  for (int l = 0; l < C.nlevels && l < 16; ++l) C.scale[l] = F.scalepyrinfo_.vscalefactor_[l];
}

// The extractor handle that still holds THIS frame's keys (the resident frame of include/vieo_hot.h), or nullptr: a
// one-camera key list (rectified stereo, monocular, RGB-D) whose search keys -- mvKeysUn when the frame is undistorted,
// identical to the extracted keys for pinhole cameras -- are what the frame's first extractor has just returned.  The
// current frame of Tracking always qualifies; a Frame copied long ago does not and takes the host-pointer entry.
vieo_orb* resident_handle(const Frame& F, const std::vector<cv::KeyPoint>& keys, int n_cams) {
  if (n_cams != 1 || F.mpORBextractors.empty() || !F.mpORBextractors[0]) return nullptr;
  vieo_orb* h = F.mpORBextractors[0]->handle();
  return vieo_orb_holds(h, reinterpret_cast<const vieo_keypoint*>(keys.data()), F.N) ? h : nullptr;
}

void taken_flags(const Frame& F, int mode, std::vector<uint8_t>& taken) {
  const auto& cur = F.GetMapPointMatches();
  taken.assign(F.N, 0);
  for (int i = 0; i < F.N; ++i)
    if (cur[i]) taken[i] = (mode == VIEO_SBP_RELOC) ? 1 : (cur[i]->Observations() > 0 ? 1 : 0);
}

// the write-back of the reference: AddMapPoint per accepted query, EraseMapPointMatch for the rotation-histogram losers
template <class MpOf>
void apply_assign(Frame& F, const std::vector<int32_t>& assign, MpOf mp_of) {
  for (int i = 0; i < F.N; ++i) {
    if (assign[i] >= 0)
      F.AddMapPoint(mp_of(assign[i]), i);
    else if (assign[i] == VIEO_SBP_ERASED)
      F.EraseMapPointMatch(i);
  }
}

// the search on caller-built queries + the write-back.  mp_of(q) = the map point of query q.  Resident frames: nothing of
// the frame goes up (vieo_search_by_projection_resident), the window grid of the frame's first search is reused.
template <class MpOf>
int search_and_apply(int mode, Frame& F, const std::vector<vieo_proj_query>& q, float ratio, bool check_ori,
                     const vieo_sbp_rig& rig, MpOf mp_of) {
  const int N = F.N, nc = rig.n_cams;
  if (N <= 0 || q.empty()) return 0;
  const std::vector<cv::KeyPoint>& keys = !Frame::usedistort_ ? F.mvKeysUn : F.mvKeys;  // FrameBase.cpp:120
  std::vector<uint8_t> taken;
  taken_flags(F, mode, taken);
  std::vector<int32_t> assign(N);
  int32_t nmatches = 0;
  if (vieo_orb* h = resident_handle(F, keys, nc)) {
    HOT_CHECK(vieo_search_by_projection_resident(mode, h, q.data(), (int)q.size(), F.stereoinfo_.vuright_.data(), taken.data(),
                                                 &rig.bounds[0][0], ratio, check_ori ? 1 : 0, assign.data(), &nmatches));
  } else {
    int32_t cam_first[5];
    vieo_shim::cam_first_of(F, nc, cam_first);
    HOT_CHECK(vieo_search_by_projection_rig(mode, q.data(), (int)q.size(), reinterpret_cast<const vieo_keypoint*>(keys.data()),
                                            F.stereoinfo_.vuright_.data(), F.mDescriptors.ptr<unsigned char>(0),
                                            taken.data(), N, cam_first, &rig.bounds[0][0], nc, ratio, check_ori ? 1 : 0,
                                            assign.data(), &nmatches));
  }
  apply_assign(F, assign, mp_of);
  return nmatches;
}

}  // namespace

// ---------------------------------------------------------------- ORBmatcher.cc:1303-1467
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono,
                                   const float th_far_pts) {
  vieo_sbp_rig rig;
  if (!vieo_shim::rig_to_pod(CurrentFrame, rig)) hot_fail("rig_to_pod", VIEO_E_INVALID);
  vieo_sbp_camera cam;
  fill_sbp_camera(CurrentFrame, th, th_far_pts, bMono, cam);
  vieo_shim::se3_to_3x4(LastFrame.GetTcwCst(), cam.Tcw_last);
  const int n = LastFrame.N, nc = rig.n_cams;
  if (n <= 0) return 0;
  const auto& lfmps = LastFrame.GetMapPointMatches();
  std::vector<vieo_last_frame_point> pts(n);
  std::memset(pts.data(), 0, sizeof(vieo_last_frame_point) * n);
  for (int i = 0; i < n; ++i) {
    MapPoint* pMP = lfmps[i];
    if (!pMP || LastFrame.mvbOutlier[i]) continue;
    const auto Xw = pMP->GetWorldPos();
    vieo_last_frame_point& p = pts[i];
    p.Xw[0] = Xw(0), p.Xw[1] = Xw(1), p.Xw[2] = Xw(2);
    p.octave = LastFrame.mvKeys[i].octave, p.angle = LastFrame.mvKeys[i].angle;
    p.flags = 1 | (pMP->Observations() > 0 ? 2 : 0);
    copy_desc(pMP->GetDescriptor(), p.desc);
  }
  // the resident frame: projection and search as ONE call on the keys still in the extractor handle
  const std::vector<cv::KeyPoint>& keys = !Frame::usedistort_ ? CurrentFrame.mvKeysUn : CurrentFrame.mvKeys;
  if (vieo_orb* h = (CurrentFrame.N > 0 && !Frame::usedistort_) ? resident_handle(CurrentFrame, keys, nc) : nullptr) {
    cam.fx = rig.cams[0].fx, cam.fy = rig.cams[0].fy, cam.cx = rig.cams[0].cx, cam.cy = rig.cams[0].cy;
    std::memcpy(cam.bounds, rig.bounds[0], sizeof(cam.bounds));
    std::vector<uint8_t> taken;
    taken_flags(CurrentFrame, VIEO_SBP_LAST_FRAME, taken);
    bool any_taken = false;
    for (uint8_t t : taken) any_taken = any_taken || t;
    if (!any_taken) {  // (TrackWithIMU / TrackWithMotionModel call with an empty frame; else the two-call form below)
      std::vector<int32_t> assign(CurrentFrame.N);
      int32_t nmatches = 0;
      HOT_CHECK(vieo_search_by_projection_last_frame_resident(h, pts.data(), n, &cam, CurrentFrame.stereoinfo_.vuright_.data(),
                                                              mfNNratio, mbCheckOrientation ? 1 : 0, assign.data(), &nmatches));
      apply_assign(CurrentFrame, assign, [&](int k) { return lfmps[k]; });
      return nmatches;
    }
  }
  std::vector<vieo_proj_query> q((size_t)n * nc);
  HOT_CHECK(vieo_sbp_project_last_frame_rig(pts.data(), n, &cam, &rig, q.data()));
  return search_and_apply(VIEO_SBP_LAST_FRAME, CurrentFrame, q, mfNNratio, mbCheckOrientation, rig,
                          [&](int k) { return lfmps[k / nc]; });
}

// ---------------------------------------------------------------- ORBmatcher.cc:230-335
int ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, const float th, const float th_far_pts) {
  vieo_sbp_rig rig;
  if (!vieo_shim::rig_to_pod(F, rig)) hot_fail("rig_to_pod", VIEO_E_INVALID);
  const bool bFactor = th != 1.0;
  std::vector<vieo_proj_query> q;
  std::vector<MapPoint*> owner;
  q.reserve(vpMapPoints.size()), owner.reserve(vpMapPoints.size());
  for (size_t iMP = 0; iMP < vpMapPoints.size(); ++iMP) {
    MapPoint* pMP = vpMapPoints[iMP];
    auto& trackinfo = pMP->GetTrackInfoRef();
    if (!trackinfo.btrack_inview_) continue;
    if (th_far_pts > 0 && trackinfo.track_depth_ > th_far_pts) continue;
    if (pMP->isBad()) continue;
    auto it_lvl = trackinfo.vtrack_scalelevel_.begin();
    auto it_cos = trackinfo.vtrack_viewcos_.begin();
    auto it_u = trackinfo.vtrack_proj_[0].begin(), it_v = trackinfo.vtrack_proj_[1].begin(),
         it_ur = trackinfo.vtrack_proj_[2].begin();
    uint8_t desc[32];
    bool have_desc = false;
    const int obs_bit = pMP->Observations() > 0 ? 2 : 0;
    for (auto it_cam = trackinfo.vtrack_cami_.begin(); it_cam != trackinfo.vtrack_cami_.end();
         ++it_cam, ++it_lvl, ++it_cos, ++it_u, ++it_v, ++it_ur) {
      const int lvl = *it_lvl;
      float r = RadiusByViewingCos(*it_cos);
      if (bFactor) r *= th;
      if (!have_desc) copy_desc(pMP->GetDescriptor(), desc), have_desc = true;
      vieo_proj_query k;
      std::memset(&k, 0, sizeof(k));
      k.u = *it_u, k.v = *it_v, k.ur = *it_ur;
      k.radius = r * F.scalepyrinfo_.vscalefactor_[lvl];
      k.level_min = lvl - 1, k.level_max = lvl;
      k.flags = 1 | obs_bit | ((int)*it_cam << 8);
      std::memcpy(k.desc, desc, 32);
      q.push_back(k), owner.push_back(pMP);
    }
  }
  return search_and_apply(VIEO_SBP_LOCAL_MAP, F, q, mfNNratio, false, rig, [&](int k) { return owner[k]; });
}

// ---------------------------------------------------------------- ORBmatcher.cc:1471-1606
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, const float th,
                                   const int ORBdist, const float th_far_pts) {
  vieo_sbp_rig rig;
  if (!vieo_shim::rig_to_pod(CurrentFrame, rig)) hot_fail("rig_to_pod", VIEO_E_INVALID);
  vieo_sbp_camera cam;
  fill_sbp_camera(CurrentFrame, th, th_far_pts, false, cam);
  const vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
  const int n = (int)vpMPs.size(), nc = rig.n_cams;
  if (n <= 0) return 0;
  std::vector<vieo_keyframe_point> pts(n);
  std::memset(pts.data(), 0, sizeof(vieo_keyframe_point) * n);
  for (int i = 0; i < n; ++i) {
    MapPoint* pMP = vpMPs[i];
    if (!pMP) continue;
    if (pMP->isBad() || sAlreadyFound.end() != sAlreadyFound.find(pMP)) continue;
    const auto Xw = pMP->GetWorldPos();
    vieo_keyframe_point& p = pts[i];
    p.Xw[0] = Xw(0), p.Xw[1] = Xw(1), p.Xw[2] = Xw(2);
    p.angle = pKF->mvKeys[i].angle;
    p.flags = 1 | (pMP->Observations() > 0 ? 2 : 0);
    p.max_distance = pMP->*MapPointAccess::max_distance();
    p.min_distance = pMP->*MapPointAccess::min_distance();
    copy_desc(pMP->GetDescriptor(), p.desc);
  }
  std::vector<vieo_proj_query> q((size_t)n * nc);
  HOT_CHECK(vieo_sbp_project_keyframe(pts.data(), n, &cam, &rig, CurrentFrame.scalepyrinfo_.flogscalefactor_, q.data()));
  return search_and_apply(VIEO_SBP_RELOC, CurrentFrame, q, (float)ORBdist, mbCheckOrientation, rig,
                          [&](int k) { return vpMPs[k / nc]; });
}

// ---------------------------------------------------------------- ORBmatcher.cc:896-1150
namespace {
struct TriKF {  // storage behind one vieo_tri_keyframe
  std::vector<uint8_t> has_mp, key_cam;
  std::vector<uint32_t> node_id;
  std::vector<int32_t> node_first, node_feat;
  std::vector<vieo_camera> cams;
  std::vector<double> Tcr, Trc;
};

void fill_tri_keyframe(KeyFrame* pKF, vieo_tri_keyframe& K, TriKF& S) {
  std::memset(&K, 0, sizeof(K));
  vieo_shim::se3_to_3x4(pKF->GetTcw(), K.Tcw);
  const bool distort = KeyFrame::usedistort_;
  const std::vector<cv::KeyPoint>& keys = !distort ? pKF->mvKeysUn : pKF->mvKeys;
  K.n_keys = pKF->N;
  K.keys = reinterpret_cast<const vieo_keypoint*>(keys.data());
  K.descriptors = pKF->mDescriptors.ptr<unsigned char>(0);
  K.uright = pKF->stereoinfo_.vuright_.data();
  S.has_mp.assign(pKF->N, 0);
  for (int i = 0; i < pKF->N; ++i) S.has_mp[i] = pKF->GetMapPoint(i) ? 1 : 0;
  K.has_mappoint = S.has_mp.data();
  S.node_first.assign(1, 0);
  for (auto it = pKF->mFeatVec.begin(); it != pKF->mFeatVec.end(); ++it) {  // std::map: ascending node ids
    S.node_id.push_back((uint32_t)it->first);
    for (auto f : it->second) S.node_feat.push_back((int32_t)f);
    S.node_first.push_back((int32_t)S.node_feat.size());
  }
  K.n_nodes = (int)S.node_id.size();
  K.node_id = S.node_id.data(), K.node_first = S.node_first.data(), K.node_feat = S.node_feat.data();
  K.scale_factor = pKF->scalepyrinfo_.vscalefactor_.data();
  K.level_sigma2 = pKF->scalepyrinfo_.vlevelsigma2_.data();
  K.n_levels = (int)pKF->scalepyrinfo_.vscalefactor_.size();
  const auto& P = pKF->mpCameras[0]->GetParameters();
  K.fx = P[0], K.fy = P[1], K.cx = P[2], K.cy = P[3];
  if (distort) {
    const int nc = (int)pKF->mpCameras.size();
    K.n_cams = nc;
    S.cams.resize(nc), S.Tcr.resize(12 * nc), S.Trc.resize(12 * nc), S.key_cam.assign(pKF->N, 0);
    const Eigen::Matrix3d I = Eigen::Matrix3d::Identity();
    const Eigen::Vector3d z = Eigen::Vector3d::Zero();
    for (int c = 0; c < nc; ++c) {
      if (!vieo_shim::to_pod(pKF->mpCameras[c].get(), I, z, S.cams[c])) hot_fail("camera", VIEO_E_INVALID);
      vieo_shim::se3_to_3x4(pKF->mpCameras[c]->GetTcr(), &S.Tcr[12 * c]);
      vieo_shim::se3_to_3x4(pKF->mpCameras[c]->GetTrc(), &S.Trc[12 * c]);
    }
    for (int i = 0; i < pKF->N && i < (int)pKF->mapn2in_.size(); ++i) S.key_cam[i] = (uint8_t)std::get<0>(pKF->mapn2in_[i]);
    K.cams = S.cams.data(), K.Tcr = S.Tcr.data(), K.Trc = S.Trc.data(), K.key_cam = S.key_cam.data();
  }
}
}  // namespace

int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, vector<vector<vector<size_t>>>& vMatchedPairs,
                                       const bool bOnlyStereo) {
  vieo_tri_keyframe k1, k2;
  TriKF s1, s2;
  fill_tri_keyframe(pKF1, k1, s1);
  fill_tri_keyframe(pKF2, k2, s2);
  const int nc1 = k1.n_cams ? k1.n_cams : 1, nc2 = k2.n_cams ? k2.n_cams : 1, stride = nc1 + nc2;
  const int cap = std::max(pKF1->N, 1);
  std::vector<int32_t> pairs((size_t)cap * stride);
  int32_t n_pairs = 0, n_matches = 0;
  HOT_CHECK(vieo_search_for_triangulation(&k1, &k2, 1, bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0, cap, stride,
                                          pairs.data(), &n_pairs, &n_matches));
  vMatchedPairs.clear();
  vMatchedPairs.reserve(n_pairs);
  for (int m = 0; m < n_pairs; ++m) {
    vector<vector<size_t>> matchedpair(2);
    const int32_t* row = &pairs[(size_t)m * stride];
    for (int c = 0; c < nc1; ++c) matchedpair[0].push_back((size_t)row[c]);  // -1 stays (size_t)-1 like the reference
    for (int c = 0; c < nc2; ++c) matchedpair[1].push_back((size_t)row[nc1 + c]);
    vMatchedPairs.push_back(matchedpair);
  }
  return n_matches;
}

// ---------------------------------------------------------------- ORBmatcher.cc:26-227 (the search) + :195-224 (the mutations)
void ORBmatcher::SearchByProjectionBase(const vector<MapPoint*>& vpMapPoints1, cv::Mat Rcrw_cv, cv::Mat tcrw_cv, KeyFrame* pKF,
                                        const float th_radius, const float th_bestdist, bool bCheckViewingAngle,
                                        const float* pbf, int* pnfused, char mode,
                                        vector<vector<bool>>* pvbAlreadyMatched1, vector<set<int>>* pvnMatch1) {
  const bool only1match = !(SBPMatchMultiCam & mode), fuselater = SBPFuseLater & mode;
  const size_t N1 = vpMapPoints1.size();
  if (pvnMatch1) {
    pvnMatch1->clear();
    pvnMatch1->resize(N1, set<int>());
  }
  const int nc = pKF->mpCameras.empty() ? 1 : (int)pKF->mpCameras.size();
  if (N1 == 0 || nc > 4) return;
  // ---- the frame (vieo_fuse_frame): float casts as the reference makes them (:30-31, :47)
  vieo_fuse_frame FF;
  std::memset(&FF, 0, sizeof(FF));
  vieo_frustum_frame& B = FF.base;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) B.Rcrw[r * 3 + c] = Rcrw_cv.at<float>(r, c);
    B.tcrw[r] = tcrw_cv.at<float>(r, 0);
  }
  const cv::Mat Ow = pKF->GetCameraCenter();
  for (int r = 0; r < 3; ++r) B.Ow[r] = Ow.at<float>(r, 0);
  B.n_cams = nc, B.use_distort = KeyFrame::usedistort_ ? 1 : 0;
  vieo_camera cams[4];
  const Eigen::Matrix3d I = Eigen::Matrix3d::Identity();
  const Eigen::Vector3d z = Eigen::Vector3d::Zero();
  int32_t cam_first[5];
  vieo_shim::cam_first_of(*pKF, nc, cam_first);
  for (int c = 0; c < nc; ++c) {
    if (!vieo_shim::to_pod(pKF->mpCameras[c].get(), I, z, cams[c])) hot_fail("camera", VIEO_E_INVALID);
    const auto Tcr = pKF->mpCameras[c]->GetTcr();
    const auto R = Tcr.rotationMatrix();
    const auto t = Tcr.translation();
    for (int r = 0; r < 3; ++r) {
      for (int k = 0; k < 3; ++k) B.Tcr[c][r * 4 + k] = R(r, k);
      B.Tcr[c][r * 4 + 3] = t(r);
    }
    const auto trc = pKF->mpCameras[c]->GetTrc().translation();
    for (int k = 0; k < 3; ++k) B.trc[c][k] = trc(k);
    for (int k = 0; k < 4; ++k) B.bounds[c][k] = vieo_shim::GridAccess::bounds()[c][k];
  }
  B.cams = cams;
  B.bf = pbf ? *pbf : 0.f;
  B.log_scale_factor = pKF->scalepyrinfo_.flogscalefactor_;
  B.n_levels = (int)pKF->scalepyrinfo_.vscalefactor_.size();
  for (int l = 0; l < B.n_levels && l < 16; ++l) {
    FF.scale_factors[l] = pKF->scalepyrinfo_.vscalefactor_[l];
    FF.inv_level_sigma2[l] = pKF->scalepyrinfo_.vinvlevelsigma2_[l];
  }
  FF.th_radius = th_radius, FF.check_viewing_angle = bCheckViewingAngle ? 1 : 0, FF.use_bf = pbf ? 1 : 0;
  // ---- the points; a point that turns bad / enters the key frame DURING the replay below is re-checked there
  std::vector<vieo_fuse_point> pts(N1);
  std::memset(pts.data(), 0, sizeof(vieo_fuse_point) * N1);
  for (size_t i1 = 0; i1 < N1; ++i1) {
    MapPoint* pMP = vpMapPoints1[i1];
    vieo_fuse_point& p = pts[i1];
    if (!pMP || pMP->isBad()) {
      p.skip_mask = INT_MIN;  // bit 31
      continue;
    }
    const auto Xw = pMP->GetWorldPos();
    const auto Pn = pMP->GetNormal();
    for (int k = 0; k < 3; ++k) p.Xw[k] = Xw(k), p.normal[k] = Pn(k);
    p.max_distance = pMP->*MapPointAccess::max_distance();
    p.min_distance = pMP->*MapPointAccess::min_distance();
    copy_desc(pMP->GetDescriptor(), p.desc);
    if (pvbAlreadyMatched1)
      for (int c = 0; c < nc; ++c)
        if ((*pvbAlreadyMatched1)[i1][c]) p.skip_mask |= 1 << c;
  }
  const std::vector<cv::KeyPoint>& keys = !KeyFrame::usedistort_ ? pKF->mvKeysUn : pKF->mvKeys;
  const vieo_keypoint* kp[4];
  const float* ur[4];
  const uint8_t* dp[4];
  int32_t nk[4];
  for (int c = 0; c < nc; ++c) {
    kp[c] = reinterpret_cast<const vieo_keypoint*>(keys.data()) + cam_first[c];
    ur[c] = pKF->stereoinfo_.vuright_.data() + cam_first[c];
    dp[c] = pKF->mDescriptors.ptr<unsigned char>(0) + (size_t)cam_first[c] * 32;
    nk[c] = cam_first[c + 1] - cam_first[c];
  }
  std::vector<int32_t> best_idx(N1 * nc), best_dist(N1 * nc);
  HOT_CHECK(vieo_fuse_search(&FF, kp, ur, dp, nk, pts.data(), (int)N1, best_idx.data(), best_dist.data()));
  // ---- the order-dependent rest of the reference's loop (:195-224), unchanged in meaning
  for (size_t i1 = 0; i1 < N1; ++i1) {
    MapPoint* pMP = vpMapPoints1[i1];
    if (!pMP || pMP->isBad()) continue;
    if (!fuselater && pMP->IsInKeyFrame(pKF)) continue;
    int bestDistInCams = INT_MAX, bestIdxInCams = -1;
    for (int cami = 0; cami < nc; ++cami) {
      if (pvbAlreadyMatched1 && (*pvbAlreadyMatched1)[i1][cami]) continue;
      if (best_idx[i1 * nc + cami] < 0) continue;
      const int bestIdx = best_idx[i1 * nc + cami] + cam_first[cami];  // index in the key frame's key list
      const int bestDist = best_dist[i1 * nc + cami];
      if (!only1match) {
        if (bestDist <= th_bestdist) {
          if (pvnMatch1) (*pvnMatch1)[i1].insert(bestIdx);
          if (!fuselater) pKF->FuseMP(bestIdx, pMP);
          if (pnfused) ++*pnfused;
        }
      } else if (bestDist < bestDistInCams) {
        auto pMP2 = pKF->GetMapPoint(bestIdx);
        if (pMP2 && !pMP2->isBad()) bestDistInCams = bestDist, bestIdxInCams = bestIdx;
      }
    }
    if (only1match && bestDistInCams <= th_bestdist) {
      auto pMP2 = pKF->GetMapPoint(bestIdxInCams);
      if (pMP2 && !pMP2->isBad()) {
        auto idxsOf1mp = pMP2->GetIndexInKeyFrame(pKF);
        for (auto iter = idxsOf1mp.begin(), iterend = idxsOf1mp.end(); iter != iterend; ++iter) {
          if (pvnMatch1) (*pvnMatch1)[i1].insert(*iter);
          if (!fuselater) pKF->FuseMP(bestIdxInCams, vpMapPoints1[i1]);
          if (pnfused) ++*pnfused;
        }
      }
    }
  }
}

// ORBmatcher.cc:1152-1165
int ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, const float th) {
  cv::Mat Rcw = pKF->GetRotation();
  cv::Mat tcw = pKF->GetTranslation();
  int nFused = 0;
  const float& bf = pKF->stereoinfo_.baseline_bf_[1];
  SearchByProjectionBase(vpMapPoints, Rcw, tcw, pKF, th, TH_LOW, true, &bf, &nFused);
  return nFused;
}

}  // namespace VIEO_SLAM
