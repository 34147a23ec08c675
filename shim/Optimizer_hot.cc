// Optimizer_hot.cc -- replacement DEFINITIONS of the VIEO_SLAM::Optimizer members on the hot path, on top of the C-ABI
// of libvieo_hot.so.  Compiled inside the reference tree against the reference's own include/Optimizer.h; the same
// members are compiled out of src/Optimizer.cc, and the body of the header template PoseOptimization<KeyFrame> is
// replaced by the two explicit specialisations below (INTEGRATION.md section 4 shows the three-line header patch).
//
//   int  PoseOptimization(Frame*, Frame* = NULL)                                        Optimizer.cc:1611-1874
//   int  PoseOptimization<Frame|KeyFrame>(Frame*, T*, gw, bComputeMarg, bNoMPs)         Optimizer.h:208-816
//   void LocalBundleAdjustment(KeyFrame*, bool* pbStopFlag, Map*, int Nlocal)           Optimizer.cc:1876-2307
//   void LocalBundleAdjustmentNavStatePRV(KeyFrame*, Nlocal, bool*, Map*, gw, bLarge, bRecInit, th_dist_far)  :21-769
//   int  GlobalBundleAdjustmentNavStatePRV(Map*, gw, nIterations, bool*, nLoopKF, bRobust, bScaleOpt, pimu_initator)  :771-1345
//   void BundleAdjustment(vpKF, vpMP, nIterations, bool*, nLoopKF, bRobust, bEnc) / GlobalBundleAdjustment(Map*, ...)  :1346-1609
//
// Everything that touches caller objects keeps the reference's order and locks: MapPoint::mGlobalMutex while the
// point positions of a frame are read (Optimizer.cc:1700, Optimizer.h:403), pMap->mMutexMapUpdate around the local-BA
// write-back (Optimizer.cc:704, :2270), *pbStopFlag polled where the reference polls it (mirrored into the int the
// C-ABI reads by a watcher: see StopMirror).
#include "Optimizer.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "FrameBase_impl.h"  // ErasePairObs
#include "vieo_flatten.hpp"

namespace VIEO_SLAM {

namespace {

[[noreturn]] void hot_fail(const char* what, int rc) {
  std::fprintf(stderr, "vieo_hot: %s failed (%d): %s\n", what, rc, vieo_last_error());
  std::abort();
}
#define HOT_CHECK(call)                        \
  do {                                         \
    const int rc_ = (call);                    \
    if (rc_ != VIEO_OK) hot_fail(#call, rc_);  \
  } while (0)

// bool* pbStopFlag (LocalMapping::mbAbortBA) -> the `volatile const int*` of the C-ABI.  The reference hands the
// bool's address to g2o (setForceStopFlag); here a watcher thread mirrors it while a bundle adjustment runs.
struct StopMirror {
  std::atomic<int> flag{0};
  std::atomic<bool> done{false};
  std::thread th;
  explicit StopMirror(bool* p) {
    if (!p) return;
    flag = *p ? 1 : 0;
    th = std::thread([this, p] {
      while (!done.load(std::memory_order_relaxed)) {
        if (*p) flag.store(1, std::memory_order_relaxed);
        std::this_thread::sleep_for(std::chrono::microseconds(50));
      }
    });
  }
  ~StopMirror() {
    done = true;
    if (th.joinable()) th.join();
  }
  volatile const int* ptr() { return reinterpret_cast<volatile const int*>(&flag); }
};

// the rig of a (key) frame as vieo_camera[n_cams] with EdgeReproject::SetParams applied (g2otypes.h:409-416)
template <class FB>
int fill_cameras(FB* f, const Eigen::Matrix3d& Rcb, const Eigen::Vector3d& tcb, vieo_camera* cams) {
  const int nc = (int)f->mpCameras.size();
  if (nc < 1 || nc > 4) hot_fail("fill_cameras: 1..4 cameras", VIEO_E_INVALID);
  for (int c = 0; c < nc; ++c)
    if (!vieo_shim::to_pod(f->mpCameras[c].get(), Rcb, tcb, cams[c])) hot_fail("fill_cameras: camera model", VIEO_E_INVALID);
  return nc;
}

// vieo_pose_frame from a Frame; cams: storage for the rig (used when Frame::usedistort_)
void fill_pose_frame(Frame* pFrame, vieo_pose_frame& f, vieo_camera* cams) {
  std::memset(&f, 0, sizeof(f));
  vieo_shim::to_pod(pFrame->GetNavStateRef(), f.nav);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) f.Rcb[r * 3 + c] = pFrame->meigRcb(r, c);
    f.tcb[r] = pFrame->meigtcb(r);
  }
  const auto& P = pFrame->mpCameras[0]->GetParameters();
  f.fx = P[0], f.fy = P[1], f.cx = P[2], f.cy = P[3];
  f.bf = pFrame->stereoinfo_.baseline_bf_[1];
  if (Frame::usedistort_) {
    f.n_cams = fill_cameras(pFrame, pFrame->meigRcb, pFrame->meigtcb, cams);
    f.cams = cams;
  }
}

// the observation gathering at the top of both pose optimisations (Optimizer.cc:1704-1786, Optimizer.h:404-470):
// one vieo_pose_obs per key that holds a map point, in key order; mvbOutlier[i] = false for them
void gather_obs(Frame* pFrame, bool vio, std::vector<vieo_pose_obs>& obs, std::vector<int>& key) {
  const int N = pFrame->N;
  const bool distort = Frame::usedistort_;
  const auto& mps = pFrame->GetMapPointsRef();
  obs.clear(), key.clear();
  obs.reserve(N), key.reserve(N);
  std::unique_lock<std::mutex> lock(MapPoint::mGlobalMutex);
  for (int i = 0; i < N; ++i) {
    MapPoint* pMP = mps[i];
    if (!pMP) continue;
    pFrame->mvbOutlier[i] = false;
    const cv::KeyPoint& kp = !distort ? pFrame->mvKeysUn[i] : pFrame->mvKeys[i];
    const auto Xw = pMP->GetWorldPos();
    vieo_pose_obs o;
    o.Xw[0] = Xw(0), o.Xw[1] = Xw(1), o.Xw[2] = Xw(2);
    o.u = kp.pt.x, o.v = kp.pt.y, o.ur = pFrame->stereoinfo_.vuright_[i];
    o.inv_sigma2 = pFrame->scalepyrinfo_.vinvlevelsigma2_[kp.octave];
    o.flags = 0;
    if (vio && pMP->GetTrackInfoRef().track_depth_ < (10 < pFrame->mThDepth ? pFrame->mThDepth : 10)) o.flags |= 1;  // bClose
    if (distort && (int)pFrame->mapn2in_.size() > i) o.flags |= (int)std::get<0>(pFrame->mapn2in_[i]) << 8;
    obs.push_back(o), key.push_back(i);
  }
}

void fill_enc(Frame* pFrame, const NavState& last, vieo_pose_enc& e) {
  std::memset(&e, 0, sizeof(e));
  vieo_shim::to_pod(pFrame->GetEncPreInt(), e.enc);
  vieo_shim::tbe_to_pod(Frame::mTbc, Frame::mTce, e.qRbe, e.pbe);
  vieo_navstate l;
  vieo_shim::to_pod(last, l);
  for (int i = 0; i < 3; ++i) e.p_last[i] = l.p[i];
  for (int i = 0; i < 4; ++i) e.q_last[i] = l.q[i];
}

}  // namespace

// ---------------------------------------------------------------- Optimizer.cc:1611-1874
int Optimizer::PoseOptimization(Frame* pFrame, Frame* pLastF) {
  pFrame->UpdateNavStatePVRFromTcw();
  vieo_camera cams[4];
  vieo_pose_frame f;
  fill_pose_frame(pFrame, f, cams);
  vieo_pose_enc enc;
  if (pLastF != NULL && pFrame->GetEncPreInt().mdeltatij && !pLastF->GetTcwRef().empty()) {
    pLastF->UpdateNavStatePVRFromTcw();
    fill_enc(pFrame, pLastF->GetNavStateRef(), enc);
    f.enc = &enc;
  }
  std::vector<vieo_pose_obs> obs;
  std::vector<int> key;
  gather_obs(pFrame, false, obs, key);
  if (obs.size() < 3) return 0;  // Optimizer.cc:1789
  f.n_obs = (int)obs.size();
  std::vector<uint8_t> outl(obs.size());
  vieo_pose_result r;
  HOT_CHECK(vieo_pose_optimization(&f, obs.data(), outl.data(), &r));
  if (r.status == VIEO_POSE_TOO_FEW) return 0;
  for (size_t k = 0; k < key.size(); ++k) pFrame->mvbOutlier[key[k]] = outl[k] != 0;
  vieo_shim::from_pod(r.nav, pFrame->GetNavStateRef());  // Optimizer.cc:1869-1871
  pFrame->UpdatePoseFromNS();
  return r.n_inliers;
}

// ---------------------------------------------------------------- Optimizer.h:208-816
namespace {
template <class LastT>
int pose_optimization_vio(Frame* pFrame, LastT* pLastKF, const cv::Mat& gw, const bool bComputeMarg, const bool bNoMPs) {
  vieo_camera cams[4];
  vieo_vio_frame f;
  std::memset(&f, 0, sizeof(f));
  fill_pose_frame(pFrame, f.base, cams);
  vieo_shim::to_pod(pLastKF->GetNavState(), f.nav_last);
  f.last_has_prior = pLastKF->mbPrior ? 1 : 0;
  if (pLastKF->mbPrior) {
    vieo_shim::to_pod(pLastKF->mNavStatePrior, f.nav_prior);
    for (int r = 0; r < 15; ++r)
      for (int c = 0; c < 15; ++c) f.H_prior[r * 15 + c] = pLastKF->mMargCovInv(r, c);
  }
  vieo_shim::to_pod(pFrame->GetIMUPreInt(), false, f.imu);
  for (int i = 0; i < 3; ++i) f.gw[i] = gw.at<float>(i, 0);  // Converter::toVector3d(gw)
  f.inv_sigma_bg2 = IMUDataBase::mInvSigmabg2, f.inv_sigma_ba2 = IMUDataBase::mInvSigmaba2;
  f.dt_frames = pFrame->ftimestamp_ - pLastKF->ftimestamp_;
  f.th_depth = pFrame->mThDepth;
  f.compute_marg = bComputeMarg ? 1 : 0, f.no_mps = bNoMPs ? 1 : 0;
  vieo_pose_enc enc;
  if (pFrame->GetEncPreInt().mdeltatij) {  // Optimizer.h:345-363
    fill_enc(pFrame, pLastKF->GetNavState(), enc);
    f.base.enc = &enc;
  }
  std::vector<vieo_pose_obs> obs;
  std::vector<int> key;
  gather_obs(pFrame, true, obs, key);
  if (obs.size() < 3 && !bNoMPs) return 0;  // Optimizer.h:499-503
  f.base.n_obs = (int)obs.size();
  std::vector<uint8_t> outl(std::max<size_t>(obs.size(), 1));
  vieo_vio_result r;
  HOT_CHECK(vieo_pose_optimization_vio(&f, obs.data(), outl.data(), &r));
  if (r.base.status == VIEO_POSE_TOO_FEW) return 0;
  for (size_t k = 0; k < key.size(); ++k) pFrame->mvbOutlier[key[k]] = outl[k] != 0;
  NavState& nsj = pFrame->GetNavStateRef();
  vieo_shim::from_pod(r.base.nav, nsj);  // p, R, v of the PVR vertex, dbg / dba of the bias vertex (Optimizer.h:651-659)
  pFrame->UpdatePoseFromNS();
  if (bComputeMarg && r.has_marg) {  // Optimizer.h:755, 809-811
    for (int a = 0; a < 15; ++a)
      for (int b = 0; b < 15; ++b) pFrame->mMargCovInv(a, b) = r.H_marg[a * 15 + b];
    pFrame->mNavStatePrior = nsj;
    pFrame->mbPrior = true;
  }
  return r.base.n_inliers;
}
}  // namespace

template <>
int Optimizer::PoseOptimization<Frame>(Frame* pFrame, Frame* pLastKF, const cv::Mat& gw, const bool bComputeMarg,
                                       const bool bNoMPs) {
  return pose_optimization_vio(pFrame, pLastKF, gw, bComputeMarg, bNoMPs);
}
template <>
int Optimizer::PoseOptimization<KeyFrame>(Frame* pFrame, KeyFrame* pLastKF, const cv::Mat& gw, const bool bComputeMarg,
                                          const bool bNoMPs) {
  return pose_optimization_vio(pFrame, pLastKF, gw, bComputeMarg, bNoMPs);
}

// ---------------------------------------------------------------- the two local bundle adjustments
namespace {

struct Window {  // the flattened window + the pointers the write-back needs
  std::vector<KeyFrame*> kf_ptr;  // local key frames first, then the fixed ones
  size_t n_local = 0;
  std::vector<MapPoint*> mp_ptr;
  std::vector<vieo_lba_keyframe> kfs;
  std::vector<float> X;
  std::vector<uint8_t> close;
  std::vector<vieo_lba_obs> obs;
  std::vector<KeyFrame*> obs_kf;
  std::vector<MapPoint*> obs_mp;
  vieo_camera cams[4];
  float thresh_depth_close = 10;
};

// points + observations in the reference's edge insertion order (Optimizer.cc:395-515 / :2066-2166): per local map
// point, per observing key frame of the window (std::map order = pointer order), per key index of that observation
void flatten_points(const std::list<MapPoint*>& lLocalMapPoints, KeyFrame* pKF, std::map<KeyFrame*, int>& kf_index,
                    bool vio, Window& W) {
  const bool distort = Frame::usedistort_;
  for (MapPoint* pMP : lLocalMapPoints) {
    const int m = (int)W.mp_ptr.size();
    W.mp_ptr.push_back(pMP);
    const auto Xw = pMP->GetWorldPos();
    W.X.push_back(Xw(0)), W.X.push_back(Xw(1)), W.X.push_back(Xw(2));
    const std::map<KeyFrame*, std::set<size_t>> observations = pMP->GetObservations();
    for (auto mit = observations.begin(); mit != observations.end(); ++mit) {
      KeyFrame* pKFi = mit->first;
      auto it = kf_index.find(pKFi);
      if (it == kf_index.end()) continue;  // LIMIT_KFS_NUM: not a key frame of the window (:417)
      if (pKFi->isBad()) continue;
      if (vio && W.thresh_depth_close < pKFi->mThDepth) W.thresh_depth_close = pKFi->mThDepth;
      for (size_t idx : mit->second) {
        const cv::KeyPoint& kp = !distort ? pKFi->mvKeysUn[idx] : pKFi->mvKeys[idx];
        vieo_lba_obs o;
        o.kf = it->second, o.mp = m;
        if (distort && pKFi->mapn2in_.size() > idx) o.kf |= (int)std::get<0>(pKFi->mapn2in_[idx]) << 24;
        o.u = kp.pt.x, o.v = kp.pt.y, o.ur = pKFi->stereoinfo_.vuright_[idx];
        o.inv_sigma2 = pKFi->scalepyrinfo_.vinvlevelsigma2_[kp.octave];
        W.obs.push_back(o), W.obs_kf.push_back(pKFi), W.obs_mp.push_back(pMP);
      }
    }
  }
  if (vio) {
    W.close.resize(W.mp_ptr.size());
    for (size_t m = 0; m < W.mp_ptr.size(); ++m)
      W.close[m] = W.mp_ptr[m]->GetTrackInfoRef().track_depth_ < W.thresh_depth_close ? 1 : 0;  // :603-611
  }
}

void fill_lba_params(KeyFrame* pKF, int its0, int its1, Window& W, vieo_lba_params& P) {
  std::memset(&P, 0, sizeof(P));
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) P.Rcb[r * 3 + c] = Frame::meigRcb(r, c);
    P.tcb[r] = Frame::meigtcb(r);
  }
  const auto& par = pKF->mpCameras[0]->GetParameters();
  P.fx = par[0], P.fy = par[1], P.cx = par[2], P.cy = par[3];
  P.bf = pKF->stereoinfo_.baseline_bf_[1];
  P.its0 = its0, P.its1 = its1;
  if (Frame::usedistort_) {
    P.n_cams = fill_cameras(pKF, Frame::meigRcb, Frame::meigtcb, W.cams);
    P.cams = W.cams;
  }
}

// protected members of MapPoint the batched UpdateNormalAndDepth writes (as MapPointAccess in ORBmatcher_hot.cc reads two)
struct MapPointWrite : public MapPoint {
  static float MapPoint::*max_distance() { return &MapPointWrite::mfMaxDistance; }
  static float MapPoint::*min_distance() { return &MapPointWrite::mfMinDistance; }
  static MapPoint::Vector3data MapPoint::*normal() { return &MapPointWrite::mNormalVector; }
  static std::mutex MapPoint::*mutex_pos() { return &MapPointWrite::mMutexPos; }
};

// void MapPoint::UpdateNormalAndDepth() (src/MapPoint.cc:424-480) for all points of a write-back as ONE call
// (vieo_update_normal_and_depth_batch) instead of one host call per point: the observations' camera centres
// (GetCameraCenter() + Rcrw^T Trc.translation() of the observing camera, :447-451) are gathered once per (key frame,
// camera), the mean viewing direction and the two scale-invariance distances come back, and the members are written
// under the point's own lock as the reference does (:474-479).  Points that are bad or have no observation are skipped
// as there (:431,:437).
void update_normal_and_depth(const std::vector<MapPoint*>& pts) {
  std::vector<MapPoint*> live;
  std::vector<float> X, centres, ref_scale;
  std::vector<int32_t> first(1, 0), obs_centre, ref_centre;
  std::map<std::pair<KeyFrame*, size_t>, int> centre_of;
  auto centre = [&](KeyFrame* pKF, size_t cami) -> int {
    auto it = centre_of.find(std::make_pair(pKF, cami));
    if (it != centre_of.end()) return it->second;
    const cv::Mat Ow = pKF->GetCameraCenter();
    double c[3] = {Ow.at<float>(0, 0), Ow.at<float>(1, 0), Ow.at<float>(2, 0)};
    if (pKF->mpCameras.size() > cami) {  // twc += Rcrw^T * Trc.translation()
      const cv::Mat R = pKF->GetRotation();
      const auto t = pKF->mpCameras[cami]->GetTrc().translation();
      for (int r = 0; r < 3; ++r)
        c[r] += (double)R.at<float>(0, r) * t(0) + (double)R.at<float>(1, r) * t(1) + (double)R.at<float>(2, r) * t(2);
    }
    const int id = (int)(centres.size() / 3);
    for (int r = 0; r < 3; ++r) centres.push_back((float)c[r]);
    centre_of[std::make_pair(pKF, cami)] = id;
    return id;
  };
  float scale_last = 1.f;
  for (MapPoint* pMP : pts) {
    if (!pMP || pMP->isBad()) continue;
    const std::map<KeyFrame*, std::set<size_t>> observations = pMP->GetObservations();
    KeyFrame* pRef = pMP->GetReferenceKeyFrame();
    if (observations.empty() || !pRef) continue;
    auto ref_it = observations.find(pRef);
    if (ref_it == observations.end() || ref_it->second.empty()) continue;  // (CV_Assert in the reference, :466)
    for (auto mit = observations.begin(); mit != observations.end(); ++mit)
      for (size_t idx : mit->second) {
        KeyFrame* pKF = mit->first;
        const size_t cami = pKF->mapn2in_.size() <= idx ? 0 : std::get<0>(pKF->mapn2in_[idx]);
        obs_centre.push_back(centre(pKF, cami));
      }
    first.push_back((int32_t)obs_centre.size());
    const auto P = pMP->GetWorldPos();
    X.push_back(P(0)), X.push_back(P(1)), X.push_back(P(2));
    // the reference key frame's own centre (camera 0: GetCameraCenter(), :461) and the scale of the key's level (:467-468)
    ref_centre.push_back(centre(pRef, (size_t)-1));
    const int level = pRef->mvKeys[*ref_it->second.begin()].octave;
    ref_scale.push_back(pRef->scalepyrinfo_.vscalefactor_[level]);
    scale_last = pRef->scalepyrinfo_.vscalefactor_.back();
    live.push_back(pMP);
  }
  if (live.empty()) return;
  const int n = (int)live.size();
  std::vector<float> nrm(3 * (size_t)n), dmax(n), dmin(n);
  HOT_CHECK(vieo_update_normal_and_depth_batch(X.data(), first.data(), obs_centre.data(), centres.data(), (int)(centres.size() / 3),
                                               ref_centre.data(), ref_scale.data(), scale_last, n, nrm.data(), dmax.data(), dmin.data()));
  for (int i = 0; i < n; ++i) {
    MapPoint* pMP = live[i];
    auto& ti = pMP->GetTrackInfoRef();
    if (INFINITY == ti.track_depth_) ti.track_depth_ = dmax[i] / ref_scale[i];  // dist (:463-464)
    std::unique_lock<std::mutex> lock(pMP->*MapPointWrite::mutex_pos());
    MapPoint::Vector3data& N = pMP->*MapPointWrite::normal();
    N(0) = nrm[3 * i], N(1) = nrm[3 * i + 1], N(2) = nrm[3 * i + 2];
    pMP->*MapPointWrite::max_distance() = dmax[i];
    pMP->*MapPointWrite::min_distance() = dmin[i];
  }
}

// Optimizer.cc:704-768 / :2251-2300 with the map lock held by the caller
void write_back(Window& W, const std::vector<vieo_navstate>& navs, const std::vector<float>& Xo,
                const std::vector<uint8_t>& erase, bool vio) {
  for (size_t e = 0; e < W.obs.size(); ++e)
    if (erase[e]) ErasePairObs(W.obs_kf[e], W.obs_mp[e], -1);
  for (size_t k = 0; k < W.n_local; ++k) {
    NavState ns = W.kf_ptr[k]->GetNavState();
    if (vio)
      vieo_shim::from_pod(navs[k], ns);
    else {  // VertexNavStatePR: pose only
      NavState pr = ns;
      vieo_shim::from_pod(navs[k], pr);
      ns.mpwb = pr.mpwb, ns.mRwb = pr.mRwb;
    }
    W.kf_ptr[k]->SetNavState(ns);
  }
  for (size_t m = 0; m < W.mp_ptr.size(); ++m) {
    MapPoint::Vector3data Pos;
    Pos(0) = Xo[3 * m], Pos(1) = Xo[3 * m + 1], Pos(2) = Xo[3 * m + 2];
    W.mp_ptr[m]->SetWorldPos(Pos);
  }
  update_normal_and_depth(W.mp_ptr);
}

}  // namespace

// ---------------------------------------------------------------- Optimizer.cc:21-769
void Optimizer::LocalBundleAdjustmentNavStatePRV(KeyFrame* pKF, int Nlocal, bool* pbStopFlag, Map* pMap, cv::Mat gw,
                                                 bool bLarge, bool bRecInit, const float th_dist_far) {
  int optit[2];
  if (bLarge) {  // ORB3_STRATEGY_OPT_WIDER with bDoMore = true (:41-49)
    Nlocal *= 2.5;
    optit[0] = 2, optit[1] = 4 - 2;
  } else
    optit[0] = 4, optit[1] = 6;
  const int maxFixKF = 200;
  // ---- window collection, as the reference does it (:52-118)
  std::list<KeyFrame*> lLocalKeyFrames;
  KeyFrame* pKFlocal = pKF;
  do {
    pKFlocal->mnBALocalForKF = pKF->nid_;
    lLocalKeyFrames.push_front(pKFlocal);
    pKFlocal = pKFlocal->GetPrevKeyFrame();
  } while (--Nlocal > 0 && pKFlocal != NULL);
  std::list<MapPoint*> lLocalMapPoints;
  for (KeyFrame* k : lLocalKeyFrames) {
    std::vector<MapPoint*> vpMPs = k->GetMapPointMatches();
    for (MapPoint* pMP : vpMPs)
      if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pKF->nid_) {
        lLocalMapPoints.push_back(pMP);
        pMP->mnBALocalForKF = pKF->nid_;
      }
  }
  std::list<KeyFrame*> lFixedCameras;
  KeyFrame* pKFPrevLocal = pKFlocal;
  if (pKFPrevLocal) {
    pKFPrevLocal->mnBAFixedForKF = pKF->nid_;
    if (!pKFPrevLocal->isBad()) lFixedCameras.push_back(pKFPrevLocal);
  }
  for (MapPoint* pMP : lLocalMapPoints) {
    auto observations = pMP->GetObservations();
    for (auto mit = observations.begin(); mit != observations.end(); ++mit) {
      KeyFrame* pKFi = mit->first;
      if (pKFi->mnBALocalForKF != pKF->nid_ && pKFi->mnBAFixedForKF != pKF->nid_) {
        pKFi->mnBAFixedForKF = pKF->nid_;
        if (!pKFi->isBad()) lFixedCameras.push_back(pKFi);
      }
      if ((int)lFixedCameras.size() >= maxFixKF) break;
    }
  }
  // ---- flatten
  Window W;
  std::map<KeyFrame*, int> kf_index;
  bool bdimPoses = false;
  for (KeyFrame* k : lLocalKeyFrames) {
    vieo_lba_keyframe r;
    std::memset(&r, 0, sizeof(r));
    vieo_shim::to_pod(k->GetNavState(), r.nav);
    r.fixed = k->nid_ == 0;
    if (!r.fixed) bdimPoses = true;
    kf_index[k] = (int)W.kfs.size();
    W.kfs.push_back(r), W.kf_ptr.push_back(k);
  }
  W.n_local = W.kfs.size();
  if (!bdimPoses) return;  // :178
  for (KeyFrame* k : lFixedCameras) {
    vieo_lba_keyframe r;
    std::memset(&r, 0, sizeof(r));
    vieo_shim::to_pod(k->GetNavState(), r.nav);
    r.fixed = 1;
    kf_index[k] = (int)W.kfs.size();
    W.kfs.push_back(r), W.kf_ptr.push_back(k);
  }
  std::vector<vieo_lba_imu_edge> imu;
  for (KeyFrame* pKF1 : lLocalKeyFrames) {  // :229-347
    KeyFrame* pKF0 = pKF1->GetPrevKeyFrame();
    if (!pKF0) continue;
    auto it0 = kf_index.find(pKF0);
    if (it0 == kf_index.end()) continue;
    vieo_lba_imu_edge e;
    std::memset(&e, 0, sizeof(e));
    e.kf_i = it0->second, e.kf_j = kf_index[pKF1];
    e.dt_kf = pKF1->ftimestamp_ - pKF0->ftimestamp_;
    vieo_shim::to_pod(pKF1->GetIMUPreInt(), true, e.imu);
    vieo_shim::to_pod(pKF1->GetEncPreInt(), e.enc);
    imu.push_back(e);
  }
  flatten_points(lLocalMapPoints, pKF, kf_index, true, W);
  if (pbStopFlag && *pbStopFlag) return;  // :524-528
  vieo_lba_vio_params P;
  std::memset(&P, 0, sizeof(P));
  fill_lba_params(pKF, optit[0], optit[1], W, P.base);
  for (int i = 0; i < 3; ++i) P.gw[i] = gw.at<float>(i, 0);
  P.inv_sigma_bg2 = IMUDataBase::mInvSigmabg2, P.inv_sigma_ba2 = IMUDataBase::mInvSigmaba2;
  P.lambda_init = bLarge ? 1e-2 : 1e0;
  P.rec_init = bRecInit ? 1 : 0, P.large = bLarge ? 1 : 0;
  P.th_dist_far = th_dist_far;
  vieo_shim::tbe_to_pod(Frame::mTbc, Frame::mTce, P.qRbe, P.pbe);
  std::vector<vieo_navstate> navs(W.kfs.size());
  std::vector<float> Xo(W.X.size());
  std::vector<uint8_t> erase(std::max<size_t>(W.obs.size(), 1));
  vieo_lba_result r;
  {
    StopMirror stop(pbStopFlag);
    HOT_CHECK(vieo_local_bundle_adjustment_vio(&P, W.kfs.data(), (int)W.kfs.size(), W.X.data(), W.close.data(),
                                               (int)W.mp_ptr.size(), W.obs.data(), (int)W.obs.size(), imu.data(),
                                               (int)imu.size(), stop.ptr(), navs.data(), Xo.data(), erase.data(), &r));
  }
  if (r.status == VIEO_LBA_DIVERGED || r.status == VIEO_LBA_NO_FREE_POSE) return;  // :660-666, :178: no write-back
  std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);  // :704
  write_back(W, navs, Xo, erase, true);
  pMap->InformNewChange();
}

// ---------------------------------------------------------------- Optimizer.cc:1876-2307
void Optimizer::LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int Nlocal) {
  // ---- window collection (:1880-1965): last-N chain when Nlocal > 0, covisibility neighbours otherwise
  std::list<KeyFrame*> lLocalKeyFrames;
  KeyFrame* pKFlocal = NULL;
  if (Nlocal > 0) {
    int NlocalCnt = Nlocal;
    pKFlocal = pKF;
    do {
      pKFlocal->mnBALocalForKF = pKF->nid_;
      lLocalKeyFrames.push_front(pKFlocal);
      pKFlocal = pKFlocal->GetPrevKeyFrame();
    } while (--NlocalCnt > 0 && pKFlocal != NULL);
  } else {
    lLocalKeyFrames.push_back(pKF);
    pKF->mnBALocalForKF = pKF->nid_;
    const std::vector<KeyFrame*> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
    for (KeyFrame* pKFi : vNeighKFs) {
      pKFi->mnBALocalForKF = pKF->nid_;
      if (!pKFi->isBad()) lLocalKeyFrames.push_back(pKFi);
    }
  }
  std::list<MapPoint*> lLocalMapPoints;
  for (KeyFrame* k : lLocalKeyFrames) {
    std::vector<MapPoint*> vpMPs = k->GetMapPointMatches();
    for (MapPoint* pMP : vpMPs)
      if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pKF->nid_) {
        lLocalMapPoints.push_back(pMP);
        pMP->mnBALocalForKF = pKF->nid_;
      }
  }
  std::list<KeyFrame*> lFixedCameras;
  if (Nlocal > 0 && pKFlocal) {  // the key frame before the window (:1925-1933)
    pKFlocal->mnBAFixedForKF = pKF->nid_;
    if (!pKFlocal->isBad()) lFixedCameras.push_back(pKFlocal);
  }
  for (MapPoint* pMP : lLocalMapPoints) {
    auto observations = pMP->GetObservations();
    for (auto mit = observations.begin(); mit != observations.end(); ++mit) {
      KeyFrame* pKFi = mit->first;
      if (pKFi->mnBALocalForKF != pKF->nid_ && pKFi->mnBAFixedForKF != pKF->nid_) {
        pKFi->mnBAFixedForKF = pKF->nid_;
        if (!pKFi->isBad()) lFixedCameras.push_back(pKFi);
      }
    }
  }
  Window W;
  std::map<KeyFrame*, int> kf_index;
  bool bdimPoses = false;
  for (KeyFrame* k : lLocalKeyFrames) {
    vieo_lba_keyframe r;
    std::memset(&r, 0, sizeof(r));
    vieo_shim::to_pod(k->GetNavState(), r.nav);
    r.fixed = k->nid_ == 0;
    if (!r.fixed) bdimPoses = true;
    kf_index[k] = (int)W.kfs.size();
    W.kfs.push_back(r), W.kf_ptr.push_back(k);
  }
  W.n_local = W.kfs.size();
  if (!bdimPoses) return;  // :1993
  for (KeyFrame* k : lFixedCameras) {
    vieo_lba_keyframe r;
    std::memset(&r, 0, sizeof(r));
    vieo_shim::to_pod(k->GetNavState(), r.nav);
    r.fixed = 1;
    kf_index[k] = (int)W.kfs.size();
    W.kfs.push_back(r), W.kf_ptr.push_back(k);
  }
  // encoder edges between consecutive local key frames (:2008-2042)
  std::vector<vieo_lba_enc_edge> enc_edges;
  for (KeyFrame* pKF1 : lLocalKeyFrames) {
    KeyFrame* pKF0 = pKF1->GetPrevKeyFrame();
    if (!pKF0) continue;
    const EncPreIntegrator encpreint = pKF1->GetEncPreInt();
    if (encpreint.mdeltatij == 0) continue;
    auto it0 = kf_index.find(pKF0);
    if (it0 == kf_index.end()) continue;
    vieo_lba_enc_edge e;
    std::memset(&e, 0, sizeof(e));
    e.kf_i = it0->second, e.kf_j = kf_index[pKF1];
    vieo_shim::to_pod(encpreint, e.enc);
    enc_edges.push_back(e);
  }
  flatten_points(lLocalMapPoints, pKF, kf_index, false, W);
  if (pbStopFlag && *pbStopFlag) return;  // :2174-2177
  vieo_lba_params P;
  fill_lba_params(pKF, 5, 10, W, P);
  vieo_lba_enc enc;
  std::memset(&enc, 0, sizeof(enc));
  enc.n_edges = (int)enc_edges.size(), enc.edges = enc_edges.data();
  vieo_shim::tbe_to_pod(Frame::mTbc, Frame::mTce, enc.qRbe, enc.pbe);
  std::vector<vieo_navstate> navs(W.kfs.size());
  std::vector<float> Xo(W.X.size());
  std::vector<uint8_t> erase(std::max<size_t>(W.obs.size(), 1));
  vieo_lba_result r;
  {
    StopMirror stop(pbStopFlag);
    HOT_CHECK(vieo_local_bundle_adjustment_enc(&P, W.kfs.data(), (int)W.kfs.size(), W.X.data(), (int)W.mp_ptr.size(),
                                               W.obs.data(), (int)W.obs.size(), enc.n_edges ? &enc : nullptr, stop.ptr(),
                                               navs.data(), Xo.data(), erase.data(), &r));
  }
  if (r.status == VIEO_LBA_NO_FREE_POSE) return;
  std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);  // :2270
  write_back(W, navs, Xo, erase, false);
  pMap->InformNewChange();
}

// ================================================================ the full bundle adjustments
namespace {

// every good key frame (vertex order = nid_ order, as g2o's ids give it), every good point with an observation in one of
// them, the observations in the reference's insertion order (point by point, std::map order over the key frames, key
// index order): Optimizer.cc:808-842,1060-1222 / :1375-1535
struct FullMap {
  Window W;
  std::vector<size_t> mp_index;  // point m of the flattened problem is vpMP[mp_index[m]]
  std::map<KeyFrame*, int> kf_index;
  bool bdimPoses = false;
};

void flatten_map(const std::vector<KeyFrame*>& vpKFs, const std::vector<MapPoint*>& vpMP, bool vio, FullMap& M) {
  std::vector<KeyFrame*> kfs;
  for (KeyFrame* k : vpKFs)
    if (k && !k->isBad()) kfs.push_back(k);
  std::sort(kfs.begin(), kfs.end(), [](KeyFrame* a, KeyFrame* b) { return a->nid_ < b->nid_; });
  for (KeyFrame* k : kfs) {
    vieo_lba_keyframe r;
    std::memset(&r, 0, sizeof(r));
    if (!vio) k->UpdateNavStatePVRFromTcw();  // :1386
    vieo_shim::to_pod(k->GetNavState(), r.nav);
    r.fixed = k->nid_ == 0;
    if (!r.fixed) M.bdimPoses = true;
    M.kf_index[k] = (int)M.W.kfs.size();
    M.W.kfs.push_back(r), M.W.kf_ptr.push_back(k);
  }
  M.W.n_local = M.W.kfs.size();
  const bool distort = Frame::usedistort_;
  for (size_t i = 0; i < vpMP.size(); ++i) {
    MapPoint* pMP = vpMP[i];
    if (!pMP || pMP->isBad()) continue;
    const std::map<KeyFrame*, std::set<size_t>> observations = pMP->GetObservations();
    const size_t obs0 = M.W.obs.size();
    const int m = (int)M.W.mp_ptr.size();
    for (auto mit = observations.begin(); mit != observations.end(); ++mit) {
      KeyFrame* pKFi = mit->first;
      auto it = M.kf_index.find(pKFi);
      if (it == M.kf_index.end()) continue;  // bad key frame / not part of this problem
      for (size_t idx : mit->second) {
        const cv::KeyPoint& kp = !distort ? pKFi->mvKeysUn[idx] : pKFi->mvKeys[idx];
        vieo_lba_obs o;
        o.kf = it->second, o.mp = m;
        if (distort && pKFi->mapn2in_.size() > idx) o.kf |= (int)std::get<0>(pKFi->mapn2in_[idx]) << 24;
        o.u = kp.pt.x, o.v = kp.pt.y, o.ur = pKFi->stereoinfo_.vuright_[idx];
        o.inv_sigma2 = pKFi->scalepyrinfo_.vinvlevelsigma2_[kp.octave];
        M.W.obs.push_back(o), M.W.obs_kf.push_back(pKFi), M.W.obs_mp.push_back(pMP);
      }
    }
    if (M.W.obs.size() == obs0) continue;  // nEdges == 0: the vertex is removed, vbNotIncludedMP (:1225-1231)
    M.W.mp_ptr.push_back(pMP), M.mp_index.push_back(i);
    const auto Xw = pMP->GetWorldPos();
    M.W.X.push_back(Xw(0)), M.W.X.push_back(Xw(1)), M.W.X.push_back(Xw(2));
  }
}

// Tcw (4 x 4, CV_32F) of a body state: Tcb * Twb^-1 (Optimizer.cc:1284-1296 / :1569-1577)
cv::Mat tcw_of(const vieo_navstate& n) {
  const double w = n.q[0], x = n.q[1], y = n.q[2], z = n.q[3];
  const double Rwb[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z),
                         2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
  cv::Mat T(4, 4, CV_32F);
  for (int r = 0; r < 3; ++r) {
    double t = Frame::meigtcb(r);
    for (int c = 0; c < 3; ++c) {
      double v = 0;
      for (int k = 0; k < 3; ++k) v += Frame::meigRcb(r, k) * Rwb[c * 3 + k];  // Rcw = Rcb * Rwb^T
      T.at<float>(r, c) = (float)v;
      t -= v * n.p[c];
    }
    T.at<float>(r, 3) = (float)t;
  }
  for (int c = 0; c < 4; ++c) T.at<float>(3, c) = c == 3 ? 1.f : 0.f;
  return T;
}

}  // namespace

// ---------------------------------------------------------------- Optimizer.cc:771-1345
int Optimizer::GlobalBundleAdjustmentNavStatePRV(Map* pMap, const cv::Mat& cvgw, int nIterations, bool* pbStopFlag,
                                                 const unsigned long nLoopKF, const bool bRobust, bool bScaleOpt,
                                                 IMUInitialization* pimu_initiator) {
  if (pimu_initiator) {
    // The IMU initialiser's form adds a gravity-direction vertex (VertexGThetaXYRwI, EdgeNavStatePRVG, a bias prior:
    // Optimizer.cc:852-905,954-972) that the C-ABI does not carry (IMU initialisation is out of scope).
    // INTEGRATION.md 4 keeps the reference's definition for that caller under the name below.
#ifdef VIEO_HOT_KEEPS_INIT_GBA
    return GlobalBundleAdjustmentNavStatePRVInit(pMap, cvgw, nIterations, pbStopFlag, nLoopKF, bRobust, bScaleOpt, pimu_initiator);
#else
    hot_fail("GlobalBundleAdjustmentNavStatePRV with pimu_initiator (build with VIEO_HOT_KEEPS_INIT_GBA, INTEGRATION.md 4)", VIEO_E_INVALID);
#endif
  }
  const std::vector<KeyFrame*> vpKFs = pMap->GetAllKeyFrames();
  const std::vector<MapPoint*> vpMP = pMap->GetAllMapPoints();
  FullMap M;
  flatten_map(vpKFs, vpMP, true, M);
  Window& W = M.W;
  const int nInitialCorrespondences = (int)W.obs.size();
  // inertial + bias (+ encoder) edges between consecutive good key frames (:923-1057)
  std::vector<vieo_lba_imu_edge> imu;
  for (KeyFrame* pKF1 : W.kf_ptr) {
    KeyFrame* pKF0 = pKF1->GetPrevKeyFrame();
    if (!pKF0) continue;
    auto it0 = M.kf_index.find(pKF0);
    if (it0 == M.kf_index.end()) continue;
    vieo_lba_imu_edge e;
    std::memset(&e, 0, sizeof(e));
    e.kf_i = it0->second, e.kf_j = M.kf_index[pKF1];
    e.dt_kf = pKF1->ftimestamp_ - pKF0->ftimestamp_;
    vieo_shim::to_pod(pKF1->GetIMUPreInt(), true, e.imu);
    vieo_shim::to_pod(pKF1->GetEncPreInt(), e.enc);
    imu.push_back(e);
  }
  std::vector<vieo_navstate> navs(W.kfs.size());
  std::vector<float> Xo(W.X.size());
  double scale = 1.0;
  if ((M.bdimPoses || bScaleOpt) && !W.kfs.empty()) {  // :850, :1236
    vieo_lba_vio_params P;
    std::memset(&P, 0, sizeof(P));
    fill_lba_params(W.kf_ptr[0], 0, 0, W, P.base);
    for (int i = 0; i < 3; ++i) P.gw[i] = cvgw.at<float>(i, 0);
    P.inv_sigma_bg2 = IMUDataBase::mInvSigmabg2, P.inv_sigma_ba2 = IMUDataBase::mInvSigmaba2;
    vieo_shim::tbe_to_pod(Frame::mTbc, Frame::mTce, P.qRbe, P.pbe);
    vieo_lba_result r;
    StopMirror stop(pbStopFlag);
    HOT_CHECK(vieo_global_bundle_adjustment_vio_scale(&P, nIterations, bRobust ? 1 : 0, bScaleOpt ? 1 : 0, W.kfs.data(), (int)W.kfs.size(),
                                                      W.X.data(), (int)W.mp_ptr.size(), W.obs.data(), (int)W.obs.size(), imu.data(),
                                                      (int)imu.size(), stop.ptr(), navs.data(), Xo.data(), &r, &scale));
  } else {  // nothing to optimise: the estimates are written back as they are
    for (size_t k = 0; k < W.kfs.size(); ++k) navs[k] = W.kfs[k].nav;
    Xo = W.X;
  }
  std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate, std::defer_lock);
  if (nLoopKF == 0) lock.lock();  // :1238-1241
  for (size_t k = 0; k < W.kf_ptr.size(); ++k) {
    KeyFrame* pKFi = W.kf_ptr[k];
    NavState ns = pKFi->GetNavState();
    vieo_shim::from_pod(navs[k], ns);  // PR, V, dbg / dba (ns_recov, :1270-1276)
    if (nLoopKF == 0)
      pKFi->SetNavState(ns);
    else {
      pKFi->mNavStateGBA = ns;
      pKFi->mTcwGBA = tcw_of(navs[k]);
      pKFi->mnBAGlobalForKF = nLoopKF;
    }
  }
  std::vector<MapPoint*> moved;
  for (size_t m = 0; m < W.mp_ptr.size(); ++m) {  // (the points come back as (float)scale * (float)Xh already, :1321)
    MapPoint* pMP = W.mp_ptr[m];
    if (pMP->isBad()) continue;
    MapPoint::Vector3data Pos;
    Pos(0) = Xo[3 * m], Pos(1) = Xo[3 * m + 1], Pos(2) = Xo[3 * m + 2];
    if (nLoopKF == 0) {
      pMP->SetWorldPos(Pos);
      moved.push_back(pMP);
    } else {
      pMP->mPosGBA = Pos;
      pMP->mnBAGlobalForKF = nLoopKF;
    }
  }
  update_normal_and_depth(moved);
  return nInitialCorrespondences;
}

// ---------------------------------------------------------------- Optimizer.cc:1353-1609
void Optimizer::BundleAdjustment(const std::vector<KeyFrame*>& vpKFs, const std::vector<MapPoint*>& vpMP, int nIterations,
                                 bool* pbStopFlag, const unsigned long nLoopKF, const bool bRobust, const bool bEnc) {
  FullMap M;
  flatten_map(vpKFs, vpMP, false, M);
  Window& W = M.W;
  if (W.kfs.empty()) return;
  std::vector<vieo_lba_enc_edge> enc_edges;
  if (bEnc)  // EdgeEncNavStatePR between consecutive key frames (:1400-1445)
    for (KeyFrame* pKF1 : W.kf_ptr) {
      KeyFrame* pKF0 = pKF1->GetPrevKeyFrame();
      if (!pKF0) continue;
      auto it0 = M.kf_index.find(pKF0);
      if (it0 == M.kf_index.end()) continue;
      vieo_lba_enc_edge e;
      std::memset(&e, 0, sizeof(e));
      e.kf_i = it0->second, e.kf_j = M.kf_index[pKF1];
      vieo_shim::to_pod(pKF1->GetEncPreInt(), e.enc);
      if (e.enc.dt != 0) enc_edges.push_back(e);
    }
  std::vector<vieo_navstate> navs(W.kfs.size());
  std::vector<float> Xo(W.X.size());
  if (M.bdimPoses) {  // :1550
    vieo_lba_params P;
    fill_lba_params(W.kf_ptr[0], 0, 0, W, P);
    vieo_lba_enc E;
    std::memset(&E, 0, sizeof(E));
    E.n_edges = (int)enc_edges.size(), E.edges = enc_edges.data();
    vieo_shim::tbe_to_pod(Frame::mTbc, Frame::mTce, E.qRbe, E.pbe);
    vieo_lba_result r;
    StopMirror stop(pbStopFlag);
    HOT_CHECK(vieo_bundle_adjustment_enc(&P, nIterations, bRobust ? 1 : 0, W.kfs.data(), (int)W.kfs.size(), W.X.data(), (int)W.mp_ptr.size(),
                                         W.obs.data(), (int)W.obs.size(), enc_edges.empty() ? nullptr : &E, stop.ptr(), navs.data(), Xo.data(),
                                         &r));
  } else {
    for (size_t k = 0; k < W.kfs.size(); ++k) navs[k] = W.kfs[k].nav;
    Xo = W.X;
  }
  for (size_t k = 0; k < W.kf_ptr.size(); ++k) {  // :1555-1580 (no map lock in the reference)
    KeyFrame* pKF = W.kf_ptr[k];
    if (nLoopKF == 0) {
      NavState ns = pKF->GetNavState();
      NavState pr = ns;
      vieo_shim::from_pod(navs[k], pr);
      ns.mpwb = pr.mpwb, ns.mRwb = pr.mRwb;  // VertexNavStatePR: pose only
      pKF->SetNavState(ns);
    } else {
      pKF->mTcwGBA = tcw_of(navs[k]);
      pKF->mnBAGlobalForKF = nLoopKF;
    }
  }
  std::vector<MapPoint*> moved;
  for (size_t m = 0; m < W.mp_ptr.size(); ++m) {
    MapPoint* pMP = W.mp_ptr[m];
    if (pMP->isBad()) continue;
    MapPoint::Vector3data Pos;
    Pos(0) = Xo[3 * m], Pos(1) = Xo[3 * m + 1], Pos(2) = Xo[3 * m + 2];
    if (nLoopKF == 0) {
      pMP->SetWorldPos(Pos);
      moved.push_back(pMP);
    } else {
      pMP->mPosGBA = Pos;
      pMP->mnBAGlobalForKF = nLoopKF;
    }
  }
  update_normal_and_depth(moved);
}

// ---------------------------------------------------------------- Optimizer.cc:1346-1351
void Optimizer::GlobalBundleAdjustment(Map* pMap, int nIterations, bool* pbStopFlag, const unsigned long nLoopKF, const bool bRobust,
                                       const bool bEnc) {
  const std::vector<KeyFrame*> vpKFs = pMap->GetAllKeyFrames();
  const std::vector<MapPoint*> vpMP = pMap->GetAllMapPoints();
  BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust, bEnc);
}

}  // namespace VIEO_SLAM
