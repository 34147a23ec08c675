// Tracking_hot.cc -- the ONE-CALL binding: a frame's whole tracking (ExtractORB x n_cams, the stereo matcher,
// PredictNavStateByIMU, SearchByProjection(last frame) -> PoseOptimization -> SearchLocalPoints -> SearchByProjection(local
// map) -> PoseOptimization(bComputeMarg)) as one vieo_track_frame call, behind replacement DEFINITIONS of
//
//   bool Tracking::TrackWithIMU(bool bMapUpdated)           src/Tracking.cc:261-378     (stereo + IMU; rectified and rigs)
//   bool Tracking::TrackLocalMapWithIMU(bool bMapUpdated)   src/Tracking.cc:453-547
//   bool Tracking::TrackWithMotionModel()                   src/Tracking.cc:1843-1922   (stereo without IMU, configs[0])
//   bool Tracking::TrackLocalMap()                          src/Tracking.cc:1924-2008
//   void Tracking::SearchLocalPoints()                      src/Tracking.cc:2308-2370   (the member-by-member path: Frame::isInFrustum
//                                                                                        of all local points as ONE call)
//
// compiled inside the reference tree against its own include/Tracking.h (declarations untouched; the four members are
// compiled out of src/Tracking.cc with `#ifndef VIEO_HOT`).  This is the path that changes the
// caller: the other shims (vieo_shim.hpp, Frame_hot.cc, ORBmatcher_hot.cc, Optimizer_hot.cc) need no change in
// Tracking.cc and are what examples/dropin_replay.cc times; this file buys the last 0.4 ms per frame.
//
// How it fits without touching Tracking::Track / GrabImageStereo / Frame::Frame:
//  * While the tracker is in its steady state the ORBextractor shims are DEFERRED (include/vieo_shim.hpp): operator()
//    keeps the image and returns no keys, Frame::Frame stops at `if (!N) return;` (src/Frame.cc:282; a rig frame runs
//    shim/Frame_hot.cc's ComputeStereoFishEyeMatches on empty inputs and ends with N = 0 as well), and Track() reaches
//    TrackWithIMU / TrackWithMotionModel with an empty frame.  The replacement member fills it from the call's outputs
//    and finishes what the constructor skipped (mvKeysUn, mvpMapPoints, mvbOutlier, image bounds, the grid).
//  * Both optimisations run inside the call.  TrackWithIMU writes back the first stage and parks the second;
//    TrackLocalMapWithIMU applies the parked stage and then does the reference's bookkeeping (IncreaseFound, the inlier
//    gates).  UpdateLocalMap() runs AFTER the frame, for the next one: the local-map candidates of a call are the
//    previous frame's mvpLocalMapPoints (they change per key frame; the candidate table is re-uploaded only then).
//  * A frame that arrives extracted (the first frames after initialisation / relocalisation, when deferral is off) takes
//    the member-by-member path below, which is the reference's sequence on the other shims.
//  * A deferred frame that leaves the steady state (pre-integration failed, tracking lost) is re-made by the ordinary
//    constructor from the kept images before any other member of Tracking looks at it.
#include "Tracking.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>

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

// protected MapPoint::mfMaxDistance / mfMinDistance (isInFrustum reads them), as in ORBmatcher_hot.cc
struct MapPointAccess : public MapPoint {
  static float MapPoint::*max_distance() { return &MapPointAccess::mfMaxDistance; }
  static float MapPoint::*min_distance() { return &MapPointAccess::mfMinDistance; }
};

// One binding per Tracking object (the class cannot grow a member without a header change).
struct HotBinding {
  vieo_tracker* trk = nullptr;
  bool vision_only = false, rig = false;
  int n_cams = 1, key_cap = 0;
  // the local-map candidate table of the last upload
  std::vector<MapPoint*> local;
  std::vector<vieo_frustum_point> local_pts;
  std::vector<uint8_t> local_desc;
  int local_version = 0;
  // per-call scratch
  std::vector<vieo_last_frame_point> last_pts;
  std::vector<float> last_depth;
  std::vector<int32_t> alias;
  std::vector<vieo_imu_sample> samples;
  // the parked second stage of the frame TrackWithIMU / TrackWithMotionModel has just run
  bool parked = false;
  std::vector<MapPoint*> stage2_points;  // per key: the map point after the local-map search (nullptr: none)
  std::vector<uint8_t> stage2_outlier;
  std::vector<float> stage2_depth;       // per key: mTrackDepth of a point found in the local map (NaN: keep)
  std::vector<MapPoint*> stage1_points;  // the points the first search + optimisation left on the keys (SearchLocalPoints' head)
  std::vector<uint8_t> local_seen;       // per candidate of `local`: Frame::isInFrustum said yes (IncreaseVisible, Tracking.cc:2342)
  vieo_vio_result second;
  int map_change_idx = -1;               // Map::GetLastChangeIdx() the candidate table was flattened at

  ~HotBinding() { vieo_tracker_destroy(trk); }

  static std::map<const Tracking*, std::unique_ptr<HotBinding>>& all() {
    static std::map<const Tracking*, std::unique_ptr<HotBinding>> m;
    return m;
  }
  static HotBinding& of(const Tracking* t) {
    auto& p = all()[t];
    if (!p) p.reset(new HotBinding());
    return *p;
  }

  // can this configuration take the one-call path?  rectified: two extractors, undistorted keys = the keys (all cameras
  // pinhole: Frame::UndistortKeyPoints copies); rig: Frame::usedistort_ with 2..4 cameras, one extractor each
  static bool supported(const std::vector<ORBextractor*>& ext, const std::vector<camm::Camera::Ptr>& cams) {
    if (Frame::usedistort_) return cams.size() >= 2 && cams.size() <= 4 && ext.size() == cams.size();
    if (ext.size() != 2 || cams.empty()) return false;
    for (const auto& c : cams)
      if (c->camera_model() != camm::Camera::kPinhole) return false;
    return true;
  }

  void create(const Frame& proto, std::vector<ORBextractor*>& ext, int width, int height, bool no_imu, float th_far,
              const cv::Mat& gw, float th_last, float th_local) {
    vieo_tracker_params P;
    std::memset(&P, 0, sizeof(P));
    P.width = width, P.height = height;
    P.n_features = ext[0]->HotFeatures(), P.n_levels = ext[0]->GetLevels();
    P.ini_th_fast = ext[0]->HotIniThFAST(), P.min_th_fast = ext[0]->HotMinThFAST();
    P.scale_factor = ext[0]->GetScaleFactor();
    const std::vector<float>& k = proto.mpCameras[0]->GetParameters();
    P.fx = k[0], P.fy = k[1], P.cx = k[2], P.cy = k[3];
    P.bf = proto.stereoinfo_.baseline_bf_[1], P.baseline = proto.stereoinfo_.baseline_bf_[0];
    P.th_depth = proto.mThDepth;
    P.th_last = th_last, P.th_local = th_local, P.nn_last = 0.9f, P.nn_local = 0.8f;
    P.max_local_points = 16384;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) P.Rcb[r * 3 + c] = Frame::meigRcb(r, c);
      P.tcb[r] = Frame::meigtcb(r);
    }
    if (!no_imu) {
      for (int r = 0; r < 3; ++r) P.gw[r] = gw.at<float>(r, 0);
      P.inv_sigma_bg2 = IMUDataBase::mInvSigmabg2, P.inv_sigma_ba2 = IMUDataBase::mInvSigmaba2;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
          P.noise.sigma_g[r * 3 + c] = IMUDataBase::mSigmag(r, c), P.noise.sigma_a[r * 3 + c] = IMUDataBase::mSigmaa(r, c);
      P.noise.freq_ref = IMUDataBase::mFreqRef, P.noise.dt_cov_noise_fixed = IMUDataBase::mdt_cov_noise_fixed;
    }
    P.vision_only = no_imu ? 1 : 0;
    vision_only = no_imu, rig = Frame::usedistort_;
    n_cams = rig ? (int)proto.mpCameras.size() : 1;
    if (rig) {
      vieo_tracker_rig R;
      std::memset(&R, 0, sizeof(R));
      R.n_cams = n_cams, R.th_far_pts = th_far;
      for (int c = 0; c < n_cams; ++c) {
        if (!vieo_shim::to_pod(proto.mpCameras[c].get(), Frame::meigRcb, Frame::meigtcb, R.cams[c])) hot_fail("camera model", VIEO_E_INVALID);
        vieo_shim::se3_to_3x4(proto.mpCameras[c]->GetTrc(), R.Trc[c]);
        vieo_shim::se3_to_3x4(proto.mpCameras[c]->GetTcr(), R.Tcr[c]);
      }
      if (proto.mpCameras[0]->camera_model() == camm::Camera::kKB8) {
        const std::vector<int>& lap = std::static_pointer_cast<camm::KB8Camera>(proto.mpCameras[0])->GetvLappingArea();
        R.use_lapping = 1, R.lapping[0] = lap[0], R.lapping[1] = lap[1];
      }
      HOT_CHECK(vieo_tracker_create_rig(&trk, &P, &R));
    } else
      HOT_CHECK(vieo_tracker_create(&trk, &P));
    key_cap = vieo_tracker_key_capacity(trk);
  }

  // One tracker per mode: a vision-only tracker integrates no IMU samples and applies no inertial edge or prior, an IMU
  // tracker without samples reports PREINT_FAILED -- a Tracking object that crosses the IMU initialisation (or a
  // relocalisation's bias recomputation) hands its frames to the other member, and the tracker follows.
  void ensure_mode(bool no_imu) {
    if (trk && vision_only != no_imu) {
      vieo_tracker_destroy(trk);
      trk = nullptr;
      local.clear();  // the new tracker has seen no candidate table: the next set_local_map uploads
      map_change_idx = -1;
      parked = false;
    }
  }

  // mvpLocalMapPoints -> the candidate table (re-uploaded by the tracker only when local_version changes).  The table
  // holds VALUES (positions, normals, distance limits, descriptors): it is rebuilt when the pointer list differs and
  // whenever the map changed since it was flattened (a local BA / loop correction moved the points, Map::InformNewChange;
  // the reference reads GetWorldPos() afresh in every frame).
  void set_local_map(const std::vector<MapPoint*>& pts, int change_idx) {
    std::vector<MapPoint*> live;
    live.reserve(pts.size());
    for (MapPoint* p : pts)
      if (p && !p->isBad()) live.push_back(p);
    if (live.size() > 16384) live.resize(16384);
    bool same = live.size() == local.size() && change_idx == map_change_idx;
    for (size_t i = 0; same && i < live.size(); ++i) same = live[i] == local[i];
    if (same) return;
    map_change_idx = change_idx;
    local.swap(live);
    local_pts.resize(local.size()), local_desc.resize(local.size() * 32);
    for (size_t j = 0; j < local.size(); ++j) {
      MapPoint* p = local[j];
      const auto X = p->GetWorldPos();
      const auto nrm = p->GetNormal();
      for (int r = 0; r < 3; ++r) local_pts[j].Xw[r] = X(r), local_pts[j].normal[r] = nrm(r);
      local_pts[j].max_distance = p->*MapPointAccess::max_distance();
      local_pts[j].min_distance = p->*MapPointAccess::min_distance();
      std::memcpy(&local_desc[j * 32], p->GetDescriptor().ptr<unsigned char>(0), 32);
    }
    ++local_version;
  }

  // One frame.  `cur` arrives empty (deferred extraction); `pred` (vision only): the predicted Tcw as a NavState whose
  // p / q are the BODY pose.  On return `cur` is the frame Frame::Frame + the first tracking stage would have left.
  // returns the call's status (VIEO_TRACK_*); n_matches = the first search's return value.
  int run(Frame& cur, const Frame& last, const std::vector<ORBextractor*>& ext, const NavState& ns_ref, double t_ref,
          const NavState* ns_prior, const Matrix<double, 15, 15>* H_prior, const NavState* pred, int* n_matches) {
    vieo_track_input in;
    std::memset(&in, 0, sizeof(in));
    const cv::Mat& im0 = ext[0]->DeferredImage();
    in.stride = (int)im0.step;
    if (rig)
      for (int c = 0; c < n_cams; ++c) in.images[c] = ext[c]->DeferredImage().data;
    else
      in.left = im0.data, in.right = ext[1]->DeferredImage().data;
    // the samples FrameBase::PreIntegration selected for [t_ref, t_cur] (Tracking::PreIntegration ran on the host: its
    // list logic -- iterijFind, the culling of old data -- is Tracking's own); the device integrates them again inside
    // the call, beside the extraction
    samples.clear();
    if (!vision_only)
      for (const auto& d : cur.GetIMUPreInt().GetRawDataRef()) {
        vieo_imu_sample s;
        s.t = d.mtm;
        for (int r = 0; r < 3; ++r) s.w[r] = d.mw(r), s.a[r] = d.ma(r);
        samples.push_back(s);
      }
    in.imu = samples.data(), in.n_imu = (int)samples.size();
    in.t_ref = t_ref, in.t_cur = cur.ftimestamp_;
    vieo_shim::to_pod(pred ? *pred : ns_ref, in.nav_ref);
    vieo_shim::to_pod(last.GetNavState(), in.nav_last);
    vieo_navstate prior_pod;
    double Hp[225];
    if (ns_prior && H_prior) {
      vieo_shim::to_pod(*ns_prior, prior_pod);
      for (int r = 0; r < 15; ++r)
        for (int c = 0; c < 15; ++c) Hp[r * 15 + c] = (*H_prior)(r, c);
      in.nav_prior = &prior_pod, in.H_prior = Hp;
    }
    // mLastFrame.mvpMapPoints flattened; the keys of one stereo group of a rig frame hold the SAME MapPoint
    const auto& lmps = last.GetMapPointMatches();
    const int nl = last.N;
    last_pts.assign(nl, vieo_last_frame_point());
    std::memset(last_pts.data(), 0, sizeof(vieo_last_frame_point) * nl);
    last_depth.assign(nl, std::numeric_limits<float>::infinity());
    std::map<MapPoint*, int> first_key;
    for (int i = 0; i < nl; ++i) {
      vieo_last_frame_point& p = last_pts[i];
      p.octave = last.mvKeys[i].octave, p.angle = last.mvKeys[i].angle;
      MapPoint* mp = lmps[i];
      if (!mp || last.mvbOutlier[i]) continue;
      const auto X = mp->GetWorldPos();
      p.Xw[0] = X(0), p.Xw[1] = X(1), p.Xw[2] = X(2);
      p.flags = 1 | (mp->Observations() > 0 ? 2 : 0);
      std::memcpy(p.desc, mp->GetDescriptor().ptr<unsigned char>(0), 32);
      last_depth[i] = mp->GetTrackInfoRef().track_depth_;
      auto it = first_key.find(mp);
      if (it == first_key.end())
        first_key[mp] = i;
      else
        p.reserved[0] = 1 + it->second;
    }
    in.n_last = nl, in.last_points = last_pts.data(), in.last_track_depth = last_depth.data();
    alias.assign(local.size(), -1);
    for (size_t j = 0; j < local.size(); ++j) {
      auto it = first_key.find(local[j]);
      if (it != first_key.end()) alias[j] = it->second;
    }
    in.n_local = (int)local.size(), in.local_version = local_version;
    in.local_points = local_pts.data(), in.local_desc = local_desc.data(), in.local_alias = alias.data();
    vieo_track_output out;
    HOT_CHECK(vieo_track_frame(trk, &in, &out));
    *n_matches = out.n_matches_last;
    // ---- Frame::Frame's outputs (src/Frame.cc:259-320, 451-779)
    const int N = out.n_keys;
    cur.N = N;
    cur.mvKeys.assign(reinterpret_cast<const cv::KeyPoint*>(out.keys), reinterpret_cast<const cv::KeyPoint*>(out.keys) + N);
    cur.mDescriptors = cv::Mat(N, 32, CV_8U);
    if (N > 0) std::memcpy(cur.mDescriptors.ptr<unsigned char>(0), out.desc, (size_t)N * 32);
    cur.stereoinfo_.vuright_.assign(out.uright, out.uright + N);
    cur.stereoinfo_.vdepth_.assign(out.depth, out.depth + N);
    cur.mapn2in_.clear();
    if (rig) {
      cur.vvkeys_.assign(n_cams, std::vector<cv::KeyPoint>());
      cur.vdescriptors_.assign(n_cams, cv::Mat());
      cur.num_mono.assign(n_cams, 0);
      for (int c = 0; c < n_cams; ++c) {
        const int a = out.cam_first[c], b = out.cam_first[c + 1];
        cur.vvkeys_[c].assign(cur.mvKeys.begin() + a, cur.mvKeys.begin() + b);
        cur.vdescriptors_[c] = cur.mDescriptors.rowRange(a, b);
        cur.num_mono[c] = (size_t)out.mono_index[c];
        for (int k = a; k < b; ++k) cur.mapn2in_.push_back(std::make_pair((size_t)c, (size_t)(k - a)));
      }
      cur.mvidxsMatches.assign(out.n_groups, vector<size_t>(n_cams, (size_t)-1));
      cur.stereoinfo_.goodmatches_.assign(out.n_groups, false);
      cur.stereoinfo_.v3dpoints_.resize(out.n_groups);
      cur.stereoinfo_.mapcamidx2idxs_.clear();
      for (int g = 0; g < out.n_groups; ++g) {
        for (int c = 0; c < n_cams; ++c)
          if (out.group_idx[g * n_cams + c] >= 0) cur.mvidxsMatches[g][c] = (size_t)out.group_idx[g * n_cams + c];
        cur.stereoinfo_.goodmatches_[g] = out.group_good[g] != 0;
        for (int r = 0; r < 3; ++r) cur.stereoinfo_.v3dpoints_[g](r) = out.group_p3d[g * 3 + r];
      }
      cur.mapin2n_.assign(n_cams, std::vector<size_t>());
      cur.mapidxs2n_.assign(out.n_groups, (size_t)-1);
      for (int k = 0; k < N; ++k) {
        const auto& ci = cur.mapn2in_[k];
        if (cur.mapin2n_[ci.first].size() <= ci.second) cur.mapin2n_[ci.first].resize(cur.vvkeys_[ci.first].size());
        cur.mapin2n_[ci.first][ci.second] = (size_t)k;
        if (out.key_group[k] >= 0) cur.stereoinfo_.mapcamidx2idxs_[ci] = (size_t)out.key_group[k], cur.mapidxs2n_[out.key_group[k]] = (size_t)k;
      }
    } else {
      cur.vvkeys_[0] = cur.mvKeys;
      cur.vdescriptors_[0] = cur.mDescriptors;
      cur.mvKeysUn = cur.mvKeys;  // Frame::UndistortKeyPoints for pinhole cameras (src/Frame.cc:424-427)
    }
    cur.GetMapPointsRef().assign(N, static_cast<MapPoint*>(nullptr));
    cur.mvbOutlier.assign(N, false);
    cur.ComputeImageBounds(std::vector<int>({im0.cols, im0.rows}));
    cur.AssignFeaturesToGrid();
    if (out.status == VIEO_TRACK_PREINT_FAILED) return out.status;
    // ---- the first stage: the last frame's points the search put on the keys, the first optimisation's state and
    //      outlier verdicts are implied by the second stage's table (an outlier of the first optimisation lost its point)
    auto point_of = [&](int ref) -> MapPoint* {
      if (ref < 0) return nullptr;
      return ref < out.key_cap ? lmps[ref] : local[ref - out.key_cap];
    };
    stage2_points.assign(N, nullptr), stage2_outlier.assign(out.outlier, out.outlier + N);
    stage2_depth.assign(N, std::numeric_limits<float>::quiet_NaN());
    stage1_points.assign(N, nullptr);
    auto& cmps = cur.GetMapPointsRef();
    for (int i = 0; i < N; ++i) {
      const int ref = out.point_ref[i];
      MapPoint* mp = point_of(ref);
      stage2_points[i] = mp;
      if (mp && ref < out.key_cap) cmps[i] = mp, stage1_points[i] = mp;  // held since the first search and still held: TrackWithIMU's survivors
      if (mp && ref >= out.key_cap) stage2_depth[i] = out.local_track_depth[ref - out.key_cap];
    }
    // Frame::isInFrustum's verdict per candidate (the device evaluated it for every one: depth < 0 = outside every camera)
    local_seen.assign(local.size(), 0);
    for (size_t j = 0; j < local.size(); ++j) local_seen[j] = out.local_track_depth[j] >= 0.f;
    second = out.second;
    parked = out.status == VIEO_TRACK_OK;
    if (out.status == VIEO_TRACK_OK) {
      NavState& ns = cur.GetNavStateRef();
      if (!vision_only) ns = ns_ref;  // bg / ba of the reference state; p, R, v, dbg, dba from the optimisation
      vieo_shim::from_pod(out.first.base.status == 0 ? out.first.base.nav : out.nav_pred, ns);
      cur.UpdatePoseFromNS();
    }
    return out.status;
  }

  // the parked second stage onto the frame (TrackLocalMap[WithIMU]'s search + optimisation)
  void apply_second_stage(Frame& cur) {
    auto& cmps = cur.GetMapPointsRef();
    const int N = cur.N;
    // SearchLocalPoints' bookkeeping (src/Tracking.cc:2308-2345): the points the frame already holds are visible and
    // marked as seen in this frame; every other live candidate inside the frustum is visible.  (GetFoundRatio() =
    // found / visible feeds LocalMapping::MapPointCulling: without these calls the ratio only grows.)
    for (int i = 0; i < N && i < (int)stage1_points.size(); ++i) {
      MapPoint* mp = stage1_points[i];
      if (!mp) continue;
      if (mp->isBad()) {
        cur.EraseMapPointMatch(i);
        continue;
      }
      mp->IncreaseVisible();
      mp->GetTrackInfoRef().Reset(&cur);
    }
    for (size_t j = 0; j < local.size() && j < local_seen.size(); ++j) {
      MapPoint* mp = local[j];
      if (mp->GetTrackInfoRef().last_seen_frameid_ == cur.nid_) continue;  // held by the frame: counted above
      if (mp->isBad()) continue;
      if (local_seen[j]) mp->IncreaseVisible();
    }
    for (int i = 0; i < N && i < (int)stage2_points.size(); ++i) {
      MapPoint* mp = stage2_points[i];
      if (mp && !cmps[i]) {
        cur.AddMapPoint(mp, i);
        if (!std::isnan(stage2_depth[i])) mp->GetTrackInfoRef().track_depth_ = stage2_depth[i];
      }
      cur.mvbOutlier[i] = stage2_outlier[i] != 0;
    }
    if (second.base.status == 0) {
      vieo_shim::from_pod(second.base.nav, cur.GetNavStateRef());
      cur.UpdatePoseFromNS();
    }
    if (!vision_only && second.has_marg) {  // Optimizer.h:755,809-811
      cur.mNavStatePrior = cur.GetNavState();
      for (int r = 0; r < 15; ++r)
        for (int c = 0; c < 15; ++c) cur.mMargCovInv(r, c) = second.H_marg[r * 15 + c];
      cur.mbPrior = true;
    }
    parked = false;
  }
};

bool frame_is_deferred(const Frame& f, const std::vector<ORBextractor*>& ext) {
  if (f.N != 0 || ext.empty()) return false;
  for (ORBextractor* e : ext)
    if (!e || !e->Deferred() || e->DeferredImage().empty()) return false;
  return true;
}
void defer_all(const std::vector<ORBextractor*>& ext, bool on) {
  for (ORBextractor* e : ext)
    if (e) e->Defer(on);
}

}  // namespace

// A deferred frame that cannot stay on the one-call path: the ordinary constructor on the kept images
// (the arguments of Tracking::GrabImageStereo, src/Tracking.cc:923-935).
namespace {
void remake_frame(Frame& cur, const std::vector<ORBextractor*>& ext, ORBVocabulary* voc, const std::vector<camm::Camera::Ptr>& cams,
                  float bf, float th_depth, float th_far) {
  std::vector<cv::Mat> ims;
  for (ORBextractor* e : ext) ims.push_back(e->DeferredImage());
  defer_all(ext, false);
  const double ts = cur.timestamp_;
  cur = Frame(ims, ts, ext, voc, cams, bf, th_depth, nullptr, nullptr, Frame::usedistort_, th_far);
}
}  // namespace

// ---------------------------------------------------------------- src/Tracking.cc:261-378
bool Tracking::TrackWithIMU(bool bMapUpdated) {
  ORBmatcher matcher(0.9, true);
  HotBinding& H = HotBinding::of(this);
  const int th = mSensor != System::STEREO ? 15 : 7;
  int nmatches = 0;
  if (frame_is_deferred(mCurrentFrame, mpORBextractors)) {
    // PredictNavStateByIMU's head on the host (the state the prediction starts from, Tracking's own pre-integration list
    // handling); the prediction itself, the searches and both optimisations in the one call
    NavState& ns = mCurrentFrame.GetNavStateRef();
    const FrameBase* ref = bMapUpdated ? static_cast<const FrameBase*>(plast_kf_) : static_cast<const FrameBase*>(&mLastFrame);
    ns = bMapUpdated ? plast_kf_->GetNavState() : mLastFrame.GetNavStateRef();
    const NavState ns_ref = ns;
    PreIntegration(bMapUpdated ? 3 : 1);
    if (mCurrentFrame.GetIMUPreInt().mdeltatij == 0) {  // no IMU data between the frames: the reference's fallback
      remake_frame(mCurrentFrame, mpORBextractors, mpORBVocabulary, mpCameras, mbf, mThDepth, mpLocalMapper->th_far_pts_);
      ns = ns_ref;
      ns.mbg += ns.mdbg, ns.mba += ns.mdba;
      ns.mdbg = ns.mdba = Eigen::Vector3d::Zero();
      if (mVelocity.empty()) return false;
      const bool ok = TrackWithMotionModel();
      if (ok) mCurrentFrame.UpdateNavStatePVRFromTcw();
      return ok;
    }
    H.ensure_mode(false);
    if (!H.trk) {
      const cv::Mat& im = mpORBextractors[0]->DeferredImage();
      H.create(mLastFrame, mpORBextractors, im.cols, im.rows, false, mpLocalMapper->th_far_pts_, mpIMUInitiator->GetGravityVec(), (float)th,
               mSensor == System::RGBD ? 3.f : 2.f);
    }
    H.set_local_map(mvpLocalMapPoints, mpMap->GetLastChangeIdx());  // (no-op unless the list or the map changed)
    const bool prior = !bMapUpdated && mLastFrame.mbPrior;
    const int st = H.run(mCurrentFrame, mLastFrame, mpORBextractors, ns_ref, ref->ftimestamp_, prior ? &mLastFrame.mNavStatePrior : nullptr,
                         prior ? &mLastFrame.mMargCovInv : nullptr, nullptr, &nmatches);
    if (st != VIEO_TRACK_OK) {  // lost (fewer than 10 matches after the wider window, :311) or no pre-integration
      defer_all(mpORBextractors, false);
      return false;
    }
  } else {
    // member by member: the reference's sequence on the other shims
    if (!PredictNavStateByIMU(bMapUpdated)) {
      if (mVelocity.empty()) return false;
      const bool ok = TrackWithMotionModel();
      if (ok) mCurrentFrame.UpdateNavStatePVRFromTcw();
      return ok;
    }
    nmatches = matcher.SearchByProjection(mCurrentFrame, mLastFrame, th, mSensor == System::MONOCULAR, mpLocalMapper->th_far_pts_);
    if (nmatches < 20) {
      auto& ref = mCurrentFrame.GetMapPointsRef();
      std::fill(ref.begin(), ref.end(), static_cast<MapPoint*>(nullptr));
      nmatches = matcher.SearchByProjection(mCurrentFrame, mLastFrame, 2 * th, mSensor == System::MONOCULAR, mpLocalMapper->th_far_pts_);
    }
    if (nmatches < 10) return false;
    if (bMapUpdated)
      Optimizer::PoseOptimization(&mCurrentFrame, plast_kf_, mpIMUInitiator->GetGravityVec(), false);
    else
      Optimizer::PoseOptimization(&mCurrentFrame, &mLastFrame, mpIMUInitiator->GetGravityVec(), false);
    // discard the optimisation's outliers (:346-363)
    const auto& cur = mCurrentFrame.GetMapPointMatches();
    for (int i = 0; i < mCurrentFrame.N; ++i)
      if (cur[i] && mCurrentFrame.mvbOutlier[i]) {
        MapPoint* pMP = cur[i];
        mCurrentFrame.EraseMapPointMatch(i);
        mCurrentFrame.mvbOutlier[i] = false;
        pMP->GetTrackInfoRef().Reset(&mCurrentFrame);
        --nmatches;
      }
  }
  int nmatchesMap = 0;
  const auto& cur = mCurrentFrame.GetMapPointMatches();
  for (int i = 0; i < mCurrentFrame.N; ++i)
    if (cur[i] && cur[i]->Observations() > 0) ++nmatchesMap;
  mnMatchesInliers = nmatchesMap;
  if (mbOnlyTracking) {
    mbVO = nmatchesMap < 6;
    return nmatches > 12;
  }
  return nmatchesMap >= 6;
}

// ---------------------------------------------------------------- src/Tracking.cc:453-547
bool Tracking::TrackLocalMapWithIMU(bool bMapUpdated) {
  HotBinding& H = HotBinding::of(this);
  if (H.parked) {
    H.apply_second_stage(mCurrentFrame);
    UpdateLocalMap();  // for the next frame's call
    H.set_local_map(mvpLocalMapPoints, mpMap->GetLastChangeIdx());
  } else {
    UpdateLocalMap();
    SearchLocalPoints();
    if (mCurrentFrame.GetIMUPreInt().mdeltatij == 0) {
      Optimizer::PoseOptimization(&mCurrentFrame, &mLastFrame);
      mCurrentFrame.UpdateNavStatePVRFromTcw();
    } else if (bMapUpdated)
      Optimizer::PoseOptimization(&mCurrentFrame, plast_kf_, mpIMUInitiator->GetGravityVec(), true);
    else
      Optimizer::PoseOptimization(&mCurrentFrame, &mLastFrame, mpIMUInitiator->GetGravityVec(), true);
  }
  // map-point statistics and the inlier gates (:490-546)
  mnMatchesInliers = 0;
  const auto& cur = mCurrentFrame.GetMapPointMatches();
  for (int i = 0; i < mCurrentFrame.N; ++i) {
    if (!cur[i]) continue;
    if (!mCurrentFrame.mvbOutlier[i]) {
      cur[i]->IncreaseFound();
      if (mbOnlyTracking || cur[i]->Observations() > 0) ++mnMatchesInliers;
    } else if (mSensor == System::STEREO)
      mCurrentFrame.EraseMapPointMatch(i);
  }
  // steady state again?  the next Frame::Frame may then skip its extraction
  // (Track() sends the next frame to TrackWithIMU only when the IMU is initialised AND no bias recomputation after a
  // relocalisation is pending, src/Tracking.cc:1020,1164: otherwise it goes to TrackWithMotionModel, which has no
  // samples for an IMU tracker)
  const bool steady = mpIMUInitiator->GetVINSInited() && !mbRelocBiasPrepare && HotBinding::supported(mpORBextractors, mpCameras) &&
                      mSensor == System::STEREO;
  bool ok;
  if (mCurrentFrame.nid_ < mnLastRelocFrameId + mMaxFrames && mnMatchesInliers < 50)
    ok = false;
  else if (mnMatchesInliers > 10 && mState == ODOMOK)
    ok = true;
  else if (mnMatchesInliers < 15)
    ok = false;
  else
    ok = !(mCurrentFrame.GetIMUPreInt().mdeltatij == 0 && mnMatchesInliers < 30);
  defer_all(mpORBextractors, ok && steady);
  return ok;
}

// ---------------------------------------------------------------- src/Tracking.cc:1843-1922 (stereo without IMU)
bool Tracking::TrackWithMotionModel() {
  ORBmatcher matcher(0.9, true);
  HotBinding& H = HotBinding::of(this);
  UpdateLastFrame();
  const int th = mSensor != System::STEREO ? 15 : 7;
  int nmatches = 0;
  if (frame_is_deferred(mCurrentFrame, mpORBextractors) && !Frame::usedistort_) {
    mCurrentFrame.SetPose(mVelocity * mLastFrame.GetTcwRef());
    mCurrentFrame.UpdateNavStatePVRFromTcw();  // the predicted pose as a body state: what the vision-only call starts from
    const NavState pred = mCurrentFrame.GetNavState();
    H.ensure_mode(true);
    if (!H.trk) {
      const cv::Mat& im = mpORBextractors[0]->DeferredImage();
      H.create(mLastFrame, mpORBextractors, im.cols, im.rows, true, 0.f, cv::Mat(), (float)th, mSensor == System::RGBD ? 3.f : 1.f);
    }
    H.set_local_map(mvpLocalMapPoints, mpMap->GetLastChangeIdx());  // (no-op unless the list or the map changed)
    const int st = H.run(mCurrentFrame, mLastFrame, mpORBextractors, pred, mLastFrame.ftimestamp_, nullptr, nullptr, &pred, &nmatches);
    if (st != VIEO_TRACK_OK) {  // fewer than 20 matches after the wider window (:1878)
      defer_all(mpORBextractors, false);
      return false;
    }
  } else {
    if (frame_is_deferred(mCurrentFrame, mpORBextractors)) remake_frame(mCurrentFrame, mpORBextractors, mpORBVocabulary, mpCameras, mbf, mThDepth, mpLocalMapper->th_far_pts_);
    mCurrentFrame.SetPose(mVelocity * mLastFrame.GetTcwRef());
    nmatches = matcher.SearchByProjection(mCurrentFrame, mLastFrame, th, mSensor == System::MONOCULAR);
    if (nmatches < 20) {
      auto& ref = mCurrentFrame.GetMapPointsRef();
      std::fill(ref.begin(), ref.end(), static_cast<MapPoint*>(nullptr));
      nmatches = matcher.SearchByProjection(mCurrentFrame, mLastFrame, 2 * th, mSensor == System::MONOCULAR);
    }
    if (nmatches < 20) return false;
    Optimizer::PoseOptimization(&mCurrentFrame, &mLastFrame);
    const auto& cur = mCurrentFrame.GetMapPointMatches();
    for (int i = 0; i < mCurrentFrame.N; ++i)
      if (cur[i] && mCurrentFrame.mvbOutlier[i]) {
        MapPoint* pMP = cur[i];
        mCurrentFrame.EraseMapPointMatch(i);
        mCurrentFrame.mvbOutlier[i] = false;
        pMP->GetTrackInfoRef().Reset(&mCurrentFrame);
        --nmatches;
      }
  }
  int nmatchesMap = 0;
  const auto& cur = mCurrentFrame.GetMapPointMatches();
  for (int i = 0; i < mCurrentFrame.N; ++i)
    if (cur[i] && cur[i]->Observations() > 0) ++nmatchesMap;
  mnMatchesInliers = nmatchesMap;
  if (mbOnlyTracking) {
    mbVO = nmatchesMap < 10;
    return nmatches > 20;
  }
  return nmatchesMap >= 10;
}

// ---------------------------------------------------------------- src/Tracking.cc:1924-2008
bool Tracking::TrackLocalMap() {
  HotBinding& H = HotBinding::of(this);
  if (H.parked) {
    H.apply_second_stage(mCurrentFrame);
    UpdateLocalMap();
    H.set_local_map(mvpLocalMapPoints, mpMap->GetLastChangeIdx());
  } else {
    UpdateLocalMap();
    SearchLocalPoints();
    Optimizer::PoseOptimization(&mCurrentFrame, &mLastFrame);
  }
  mnMatchesInliers = 0;
  const auto& cur = mCurrentFrame.GetMapPointMatches();
  for (int i = 0; i < mCurrentFrame.N; ++i) {
    if (!cur[i]) continue;
    if (!mCurrentFrame.mvbOutlier[i]) {
      cur[i]->IncreaseFound();
      if (mbOnlyTracking || cur[i]->Observations() > 0) ++mnMatchesInliers;
    } else if (mSensor == System::STEREO)
      mCurrentFrame.EraseMapPointMatch(i);
  }
  const int reloc_gate = (mbOnlyTracking && mCurrentFrame.GetEncPreInt().mdeltatij > 0) ? 25 : 50;
  bool ok;
  if (mnLastRelocFrameId && mCurrentFrame.nid_ < mnLastRelocFrameId + mMaxFrames && mnMatchesInliers < reloc_gate)
    ok = false;
  else
    ok = mnMatchesInliers >= 15;
  const bool steady = mSensor == System::STEREO && !Frame::usedistort_ && HotBinding::supported(mpORBextractors, mpCameras) &&
                      (!mpIMUInitiator->GetVINSInited() || mbRelocBiasPrepare) && !mVelocity.empty();
  defer_all(mpORBextractors, ok && steady);
  return ok;
}

// ---------------------------------------------------------------- src/Tracking.cc:2308-2370
// The loop `for (pMP : mvpLocalMapPoints) if (mCurrentFrame.isInFrustum(pMP, 0.5)) ...` as one vieo_is_in_frustum_batch call
// (Frame::isInFrustum, src/Frame.cc:335-416, per point on the device); what the member writes into the point's
// TrackFastMatchInfo is written here from the call's records, in the same push order.
void Tracking::SearchLocalPoints() {
  const auto& curfmps = mCurrentFrame.GetMapPointMatches();
  for (size_t i = 0; i < curfmps.size(); ++i) {  // :2313-2327
    MapPoint* pMP = curfmps[i];
    if (!pMP) continue;
    if (pMP->isBad())
      mCurrentFrame.EraseMapPointMatch(i);
    else {
      pMP->IncreaseVisible();
      pMP->GetTrackInfoRef().Reset(&mCurrentFrame);
    }
  }
  std::vector<MapPoint*> cand;
  cand.reserve(mvpLocalMapPoints.size());
  for (MapPoint* pMP : mvpLocalMapPoints) {  // :2332-2337
    if (pMP->GetTrackInfoRef().last_seen_frameid_ == mCurrentFrame.nid_) continue;
    if (pMP->isBad()) continue;
    cand.push_back(pMP);
  }
  int nToMatch = 0;
  if (!cand.empty()) {
    Frame& F = mCurrentFrame;
    const int nc = F.mpCameras.empty() ? 1 : (int)F.mpCameras.size();
    if (nc > 4) hot_fail("SearchLocalPoints: at most 4 cameras", VIEO_E_INVALID);
    vieo_frustum_frame B;
    std::memset(&B, 0, sizeof(B));
    const cv::Mat& Tcw = F.GetTcwRef();  // CV_32F (Frame.cc:346-349 reads it through double and back: the same floats)
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) B.Rcrw[r * 3 + c] = Tcw.at<float>(r, c);
      B.tcrw[r] = Tcw.at<float>(r, 3);
    }
    const cv::Mat Ow = F.GetCameraCenter();
    for (int r = 0; r < 3; ++r) B.Ow[r] = Ow.at<float>(r, 0);
    B.n_cams = nc, B.use_distort = Frame::usedistort_ ? 1 : 0;
    vieo_camera cams[4];
    const Eigen::Matrix3d I = Eigen::Matrix3d::Identity();
    const Eigen::Vector3d z = Eigen::Vector3d::Zero();
    for (int c = 0; c < nc; ++c) {
      if (!vieo_shim::to_pod(F.mpCameras[c].get(), I, z, cams[c])) hot_fail("camera model", VIEO_E_INVALID);
      {
        const auto Tcr = F.mpCameras[c]->GetTcr();
        const auto R = Tcr.rotationMatrix();
        const auto t = Tcr.translation();
        for (int r = 0; r < 3; ++r) {
          for (int k = 0; k < 3; ++k) B.Tcr[c][r * 4 + k] = R(r, k);
          B.Tcr[c][r * 4 + 3] = t(r);
        }
      }
      const auto trc = F.mpCameras[c]->GetTrc().translation();
      for (int k = 0; k < 3; ++k) B.trc[c][k] = trc(k);
      for (int k = 0; k < 4; ++k) B.bounds[c][k] = vieo_shim::GridAccess::bounds()[c][k];
    }
    B.cams = cams;
    B.bf = F.stereoinfo_.baseline_bf_[1];
    B.log_scale_factor = F.scalepyrinfo_.flogscalefactor_;
    B.n_levels = (int)F.scalepyrinfo_.vscalefactor_.size();
    B.viewing_cos_limit = 0.5f;
    std::vector<vieo_frustum_point> pts(cand.size());
    for (size_t j = 0; j < cand.size(); ++j) {
      MapPoint* p = cand[j];
      const auto X = p->GetWorldPos();
      const auto nrm = p->GetNormal();
      for (int r = 0; r < 3; ++r) pts[j].Xw[r] = X(r), pts[j].normal[r] = nrm(r);
      pts[j].max_distance = p->*MapPointAccess::max_distance();
      pts[j].min_distance = p->*MapPointAccess::min_distance();
    }
    std::vector<vieo_track_info> info(cand.size());
    HOT_CHECK(vieo_is_in_frustum_batch(&B, pts.data(), (int)pts.size(), info.data()));
    for (size_t j = 0; j < cand.size(); ++j) {
      auto& ti = cand[j]->GetTrackInfoRef();
      ti.Reset();
      const vieo_track_info& T = info[j];
      for (int k = 0; k < T.n; ++k) {  // Frame.cc:399-406
        ti.vtrack_proj_[0].push_back(T.u[k]);
        ti.vtrack_proj_[1].push_back(T.v[k]);
        ti.vtrack_proj_[2].push_back(T.ur[k]);
        ti.vtrack_scalelevel_.push_back(T.level[k]);
        ti.vtrack_viewcos_.push_back(T.viewcos[k]);
        ti.vtrack_cami_.push_back((size_t)T.cam[k]);
      }
      if (T.n > 0) ti.track_depth_ = T.track_depth;
      ti.btrack_inview_ = T.n > 0;
      if (T.n > 0) {
        cand[j]->IncreaseVisible();
        ++nToMatch;
      }
    }
  }
  if (nToMatch > 0) {  // :2346-2367
    ORBmatcher matcher(0.8);
    int th = 1;
    if (mSensor == System::RGBD) th = 3;
    if (mpIMUInitiator->GetVINSInited()) th = 2;
    if (mCurrentFrame.nid_ < mnLastRelocFrameId + 2) th = 5;
    if (ODOMOK == mState) th = 15;
    nToMatch = matcher.SearchByProjection(mCurrentFrame, mvpLocalMapPoints, th, mpLocalMapper->th_far_pts_);
  }
}

}  // namespace VIEO_SLAM
