// Optimizer_hot.cc -- replacement DEFINITIONS of the VIEO_SLAM::Optimizer members on the hot path, on top of the C-ABI
// of libvieo_hot.so.  Compiled inside the reference tree against the reference's own include/Optimizer.h; the same
// members are compiled out of src/Optimizer.cc, and the body of the header template PoseOptimization<KeyFrame> is
// replaced by the two explicit specialisations below (INTEGRATION.md section 4 shows the three-line header patch).
//
//   int  PoseOptimization(Frame*, Frame* = NULL)                                        Optimizer.cc:1611-1874
//   int  PoseOptimization<Frame|KeyFrame>(Frame*, T*, gw, bComputeMarg, bNoMPs)         Optimizer.h:208-816
//   void LocalBundleAdjustment(KeyFrame*, bool* pbStopFlag, Map*, int Nlocal)           Optimizer.cc:1876-2307
//   void LocalBundleAdjustmentNavStatePRV(KeyFrame*, Nlocal, bool*, Map*, gw, bLarge, bRecInit, th_dist_far)  :21-769
//
// Everything that touches caller objects keeps the reference's order and locks: MapPoint::mGlobalMutex while the
// point positions of a frame are read (Optimizer.cc:1700, Optimizer.h:403), pMap->mMutexMapUpdate around the local-BA
// write-back (Optimizer.cc:704, :2270), *pbStopFlag polled where the reference polls it (mirrored into the int the
// C-ABI reads by a watcher: see StopMirror).
#include "Optimizer.h"

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
    W.mp_ptr[m]->UpdateNormalAndDepth();
  }
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

}  // namespace VIEO_SLAM
