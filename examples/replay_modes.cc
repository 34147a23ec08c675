// replay_modes.cc -- the sequential replays of the OTHER configurations of BASELINE.json without Python: the C++ twin of
// vieo_slam_amd/replay_modes.py (RigTrackerReplay / VisionTrackerReplay), as examples/replay_main.cc is the twin of
// tracker.TrackerReplay.
//
//   rig     a "VSEQ0002" file: 2..4 distorted cameras + IMU (the reference's default MH05 set-up, configs[3], configs[4]).
//           Per frame ONE vieo_track_frame call on a rig tracker (ExtractORB x n_cams, ComputeStereoFishEyeMatches, the
//           camera loops of both SearchByProjection overloads, PoseOptimization x 2); per key frame new map points from
//           the good stereo groups (Tracking::CreateNewKeyFrame, src/Tracking.cc:2168-2306: one point per group, held
//           and observed by one key per camera), the key-frame pair's pre-integration and
//           vieo_local_bundle_adjustment_vio with per-camera edges (camera of an observation in bits 24..27).
//   vision  a "VSEQ0001" file with --vision: configs[0], rectified stereo WITHOUT IMU, 1000 features.
//           TrackWithMotionModel + TrackLocalMap (src/Tracking.cc:1843-2008) as one vieo_track_frame call on a
//           vision-only tracker; the constant-velocity prediction (mVelocity, :1178-1189) and UpdateLastFrame
//           (:1790-1841: the last frame's pose follows its reference key frame) are this caller's 4x4 products;
//           per key frame vieo_local_bundle_adjustment (Optimizer::LocalBundleAdjustment, src/Optimizer.cc:1876-2307).
//
// Same simplifications as the other replays: a key frame every `kf_every` frames, local map = the points of the last
// `n_local_kfs` key frames, first frame initialised with the true state, LocalMapping on its own host thread with its
// write-back reaching the tracker `--lba-lag` frames after the key frame (0: inline).  The map lives here on the host.
//
//   python tools/write_sequence.py rig.vseq --rig radtan --cams 2 --features 1200 --frames 100
//   ./examples/replay_modes rig.vseq [traj.bin] [--frames N] [--warmup M] [--lba-lag L] [--prefetch 0|1] [--quiet]
//   ./examples/replay_modes seq.vseq [traj.bin] --vision ...
// Prints one JSON line (whole-loop ms per frame, the tracking call, its GPU part, the local BAs, error against the true
// trajectory); traj.bin receives the vieo_navstate of every frame.  Built by __graft_entry__.build().
#include "replay_common.hpp"

using namespace vieo_replay;

namespace {

// ---- 4x4 rigid transforms, row-major (the vision mode's motion model)
void T_of(const vieo_navstate& n, double* T) {
  double R[9];
  quat_to_R(n.q, R);
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T[r * 4 + c] = R[r * 3 + c];
    T[r * 4 + 3] = n.p[r];
  }
  T[12] = T[13] = T[14] = 0, T[15] = 1;
}
void mul4(const double* A, const double* B, double* C) {
  double t[16];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) t[r * 4 + c] = A[r * 4] * B[c] + A[r * 4 + 1] * B[4 + c] + A[r * 4 + 2] * B[8 + c] + A[r * 4 + 3] * B[12 + c];
  std::memcpy(C, t, sizeof(t));
}
void inv_rigid(const double* T, double* I) {
  double t[16];
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) t[r * 4 + c] = T[c * 4 + r];
    t[r * 4 + 3] = -(T[r] * T[3] + T[4 + r] * T[7] + T[8 + r] * T[11]);
  }
  t[12] = t[13] = t[14] = 0, t[15] = 1;
  std::memcpy(I, t, sizeof(t));
}
// synth_ba._R_to_quat: (w, x, y, z), normalised
void R_to_quat(const double* T, double* q) {
  auto R = [&](int r, int c) { return T[r * 4 + c]; };
  const double tr = R(0, 0) + R(1, 1) + R(2, 2);
  if (tr > 0) {
    const double s = std::sqrt(tr + 1.0) * 2;
    q[0] = 0.25 * s, q[1] = (R(2, 1) - R(1, 2)) / s, q[2] = (R(0, 2) - R(2, 0)) / s, q[3] = (R(1, 0) - R(0, 1)) / s;
  } else {
    int i = 0;
    if (R(1, 1) > R(i, i)) i = 1;
    if (R(2, 2) > R(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    const double s = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0) * 2;
    q[0] = (R(k, j) - R(j, k)) / s, q[1 + i] = 0.25 * s, q[1 + j] = (R(j, i) + R(i, j)) / s, q[1 + k] = (R(k, i) + R(i, k)) / s;
  }
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int a = 0; a < 4; a++) q[a] /= n;
}
vieo_navstate nav_of(const double* T, const vieo_navstate& like) {
  vieo_navstate n = like;
  n.p[0] = T[3], n.p[1] = T[7], n.p[2] = T[11];
  R_to_quat(T, n.q);
  return n;
}

// one frame; a key frame IS its frame (shared with `last`, as in the Python driver)
struct MFrame {
  int k = 0, id = -1, N = 0;
  double t = 0;
  std::vector<vieo_keypoint> keys;
  std::vector<uint8_t> desc, outlier;
  std::vector<float> uright, depth, track_depth;
  std::vector<long> mp_ref;
  vieo_navstate nav;
  bool has_prior = false;
  vieo_navstate prior_nav;
  double H_prior[225];
  double Rwc[9], twc[3];
  bool has_edge = false;
  vieo_imu_preint edge;
  // rig frames: keys in mvKeys (camera-major) order, the stereo groups of ComputeStereoFishEyeMatches
  int cam_first[5] = {0, 0, 0, 0, 0};
  std::vector<int32_t> key_cam, group_idx;
  std::vector<uint8_t> group_good;
  std::vector<double> group_p3d;
  int n_groups = 0;
  // vision-only frames: the reference key frame and the pose relative to it (mpReferenceKF, mlRelativeFramePoses)
  int ref_kf = -1;
  double T_rel[16];
};
typedef std::shared_ptr<MFrame> MFramePtr;

struct Row { int kid, key; };
struct LbaJob {
  std::vector<int> local;
  std::vector<long> pts;
  std::vector<vieo_lba_keyframe> K;
  std::vector<vieo_lba_obs> obs;
  std::vector<Row> rows;
  std::vector<vieo_lba_imu_edge> edges;
  vieo_lba_vio_params P;
  std::vector<float> X, Xo;
  std::vector<uint8_t> close, erase;
  std::vector<vieo_navstate> navs;
  bool deferred = false, vision = false;
  std::vector<long> nd_ids;
  std::vector<float> nd_nrm, nd_mx, nd_mn;
  vieo_lba_result res;
  double ms = 0, ms_preint = 0;
  int rc = 0;
  bool need_edge = false, edge_done = false;  // edge_done: the pre-integration has run (on the helper thread)
  int edge_kf = -1;
  std::vector<vieo_imu_sample> samples;
  vieo_imu_noise noise;
  double ti = 0, tj = 0, bg[3], ba[3];
  vieo_imu_preint edge;
};

struct Replay {
  const Sequence& S;
  const bool vision;
  const int nc;  // cameras whose keys make up a frame (1: the rectified pair's left image)
  int kf_every = 10, n_local = 10, n_local_kfs = 10;
  float th_last = 7.0f, th_local = 2.0f, th_depth = 35.0f;
  double Tcb[16];
  float scale[16], inv_sigma2[16];
  vieo_tracker* trk = nullptr;
  // the map: a point is observed by a key frame through one key (rectified) or one key per camera (rig)
  std::vector<float> mp_X, mp_normal, mp_maxd, mp_mind;
  std::vector<uint8_t> mp_desc, mp_bad;
  std::vector<std::map<int, std::vector<int>>> mp_obs;  // key-frame id -> its keys that hold the point, ascending
  std::vector<MFramePtr> kfs;
  std::vector<vieo_navstate> traj;
  MFramePtr last;
  bool map_updated = false;
  double velocity[16];  // T_b(k-1)<-b(k): the constant-velocity model in the body frame (vision)
  int n_lba = 0, n_lba_applied = 0, local_version = 0, widened = 0;
  int lba_lag = 0, lba_due = -1;
  std::unique_ptr<LbaJob> job;
  std::thread lba_thread;
  std::mutex lba_m;
  std::condition_variable lba_cv;
  LbaJob* lba_todo = nullptr;
  bool lba_busy = false, lba_quit = false;
  std::thread pre_thread;  // its helper: the key-frame pair's pre-integration beside the window's flattening
  std::mutex pre_m;
  std::condition_variable pre_cv;
  LbaJob* pre_todo = nullptr;
  bool pre_busy = false, pre_quit = false;
  std::vector<long> lp;
  std::vector<vieo_frustum_point> lp_pts;
  std::vector<uint8_t> lp_desc;
  size_t lp_key_kfs = (size_t)-1;
  int lp_key_lba = -1;
  bool prefetch = false, prefetched = false;
  int last_frame = -1;
  double next_bias[6];
  double ms_track = 0, ms_gpu = 0, ms_lba = 0, ms_kf_preint = 0;
  std::vector<double> frame_ms;
  long win_kfs = 0, win_fixed = 0, win_points = 0, win_obs = 0;
  int win_max_kfs = 0, win_max_fixed = 0;

  Replay(const Sequence& s, bool vision_only) : S(s), vision(vision_only), nc(vision_only ? 1 : s.n_cams) {
    std::memcpy(Tcb, S.Tcb, sizeof(Tcb));
    vieo_orb* ext = nullptr;  // (only asked for the scale factors)
    CHECK(vieo_orb_create(&ext, 1000, SCALE, NLEVELS, INI_TH, MIN_TH));
    CHECK(vieo_orb_scale_factors(ext, scale));
    vieo_orb_destroy(ext);
    for (int l = 0; l < NLEVELS; l++) inv_sigma2[l] = 1.0f / (scale[l] * scale[l]);
    if (vision) {
      th_local = 1.0f, th_depth = S.th_depth;
      vieo_tracker_params P;
      std::memset(&P, 0, sizeof(P));
      P.width = S.W, P.height = S.H, P.n_features = 1000, P.n_levels = NLEVELS, P.ini_th_fast = INI_TH, P.min_th_fast = MIN_TH;
      P.scale_factor = SCALE;
      P.fx = S.fx, P.fy = S.fy, P.cx = S.cx, P.cy = S.cy, P.bf = S.bf, P.baseline = S.baseline, P.th_depth = S.th_depth;
      P.th_last = th_last, P.th_local = th_local, P.nn_last = 0.9f, P.nn_local = 0.8f, P.max_local_points = 16384;
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) P.Rcb[r * 3 + c] = Tcb[r * 4 + c];
        P.tcb[r] = Tcb[r * 4 + 3];
      }
      std::memcpy(P.gw, GRAVITY, 24);
      P.inv_sigma_bg2 = 1.0 / (IMU_SIGMA[2] * IMU_SIGMA[2]), P.inv_sigma_ba2 = 1.0 / (IMU_SIGMA[3] * IMU_SIGMA[3]);
      P.noise = S.noise;
      P.vision_only = 1;
      CHECK(vieo_tracker_create(&trk, &P));
    } else {
      th_last = S.trk_params.th_last, th_local = S.trk_params.th_local, th_depth = S.trk_params.th_depth;
      CHECK(vieo_tracker_create_rig(&trk, &S.trk_params, &S.rig));
    }
  }
  ~Replay() {
    if (lba_thread.joinable()) {
      {
        std::lock_guard<std::mutex> g(lba_m);
        lba_quit = true;
      }
      lba_cv.notify_all();
      lba_thread.join();
    }
    if (pre_thread.joinable()) {
      {
        std::lock_guard<std::mutex> g(pre_m);
        pre_quit = true;
      }
      pre_cv.notify_all();
      pre_thread.join();
    }
    vieo_tracker_destroy(trk);
  }
  Replay(const Replay&) = delete;
  Replay& operator=(const Replay&) = delete;

  void pose_of(MFrame& f) const {  // Twc = Twb Tbc
    double Rwb[9];
    quat_to_R(f.nav.q, Rwb);
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) f.Rwc[r * 3 + c] = Rwb[r * 3] * S.Tbc[c] + Rwb[r * 3 + 1] * S.Tbc[4 + c] + Rwb[r * 3 + 2] * S.Tbc[8 + c];
      f.twc[r] = f.nav.p[r] + (Rwb[r * 3] * S.Tbc[3] + Rwb[r * 3 + 1] * S.Tbc[7] + Rwb[r * 3 + 2] * S.Tbc[11]);
    }
  }
  void centre_of(const vieo_navstate& nav, double* twc) const {
    double Rwb[9];
    quat_to_R(nav.q, Rwb);
    for (int r = 0; r < 3; r++) twc[r] = nav.p[r] + (Rwb[r * 3] * S.Tbc[3] + Rwb[r * 3 + 1] * S.Tbc[7] + Rwb[r * 3 + 2] * S.Tbc[11]);
  }
  void add_obs(long m, int kid, int key) {
    std::vector<int>& v = mp_obs[m][kid];
    const auto it = std::lower_bound(v.begin(), v.end(), key);
    if (it == v.end() || *it != key) v.insert(it, key);
  }
  long new_point(const MFrame& kf, const double* Xw, int first_key) {
    double d[3];
    for (int r = 0; r < 3; r++) d[r] = Xw[r] - kf.twc[r];
    const double dist = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const long m = (long)mp_bad.size();
    for (int r = 0; r < 3; r++) mp_X.push_back((float)Xw[r]), mp_normal.push_back((float)(d[r] / dist));
    mp_desc.insert(mp_desc.end(), kf.desc.begin() + (size_t)first_key * 32, kf.desc.begin() + (size_t)(first_key + 1) * 32);
    mp_bad.push_back(0);
    mp_obs.emplace_back();
    const float maxd = (float)(dist * (double)scale[kf.keys[first_key].octave]);
    mp_maxd.push_back(maxd), mp_mind.push_back(maxd / scale[NLEVELS - 1]);
    return m;
  }

  // Tracking::CreateNewKeyFrame + LocalMapping::ProcessNewKeyFrame
  void insert_keyframe(const MFramePtr& f, const vieo_navstate& nav, const vieo_imu_preint* edge) {
    MFrame& kf = *f;
    kf.id = (int)kfs.size();
    kf.nav = nav;
    pose_of(kf);
    kf.has_edge = edge != nullptr;
    if (edge) kf.edge = *edge;
    kfs.push_back(f);
    for (int i = 0; i < kf.N; i++)  // AddObservation
      if (kf.mp_ref[i] >= 0) add_obs(kf.mp_ref[i], kf.id, i);
    if (vision) {
      // new points from stereo depth: keys with depth and without a point, nearest first; all close ones, at least 100
      std::vector<int> cand;
      for (int i = 0; i < kf.N; i++)
        if (kf.depth[i] > 0 && kf.mp_ref[i] < 0) cand.push_back(i);
      std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return kf.depth[a] < kf.depth[b]; });
      int n_close = 0;
      for (int i : cand) n_close += kf.depth[i] <= th_depth;
      cand.resize(std::max(n_close, std::min(100, (int)cand.size())));
      for (int i : cand) {
        const double z = (double)kf.depth[i];
        const double Xc[3] = {(double)(kf.keys[i].x - S.cx) * z / S.intr[0], (double)(kf.keys[i].y - S.cy) * z / S.intr[1], z};
        double Xw[3];
        for (int r = 0; r < 3; r++) Xw[r] = Xc[0] * kf.Rwc[r * 3] + Xc[1] * kf.Rwc[r * 3 + 1] + Xc[2] * kf.Rwc[r * 3 + 2] + kf.twc[r];
        const long m = new_point(kf, Xw, i);
        kf.mp_ref[i] = m, add_obs(m, kf.id, i);
      }
      return;
    }
    // a point per good stereo group none of whose keys holds one; all the keys of the group then hold and observe it
    struct G { double z; int g; };
    std::vector<G> groups;
    auto keys_of = [&](int g, int* ks) {
      int n = 0;
      for (int c = 0; c < nc; c++)
        if (kf.group_idx[(size_t)g * nc + c] >= 0) ks[n++] = kf.cam_first[c] + kf.group_idx[(size_t)g * nc + c];
      return n;
    };
    for (int g = 0; g < kf.n_groups; g++) {
      if (!kf.group_good[g]) continue;
      int ks[4];
      const int n = keys_of(g, ks);
      bool free_keys = n > 0;
      for (int a = 0; a < n; a++) free_keys = free_keys && kf.mp_ref[ks[a]] < 0;
      if (free_keys) groups.push_back(G{kf.group_p3d[(size_t)g * 3 + 2], g});
    }
    std::stable_sort(groups.begin(), groups.end(), [](const G& a, const G& b) { return a.z < b.z; });
    int n_close = 0;
    for (const G& g : groups) n_close += g.z <= (double)th_depth;
    groups.resize(std::max(n_close, std::min(100, (int)groups.size())));
    for (const G& gg : groups) {
      int ks[4];
      const int n = keys_of(gg.g, ks);
      const double* P3 = &kf.group_p3d[(size_t)gg.g * 3];
      double Xw[3];
      for (int r = 0; r < 3; r++) Xw[r] = P3[0] * kf.Rwc[r * 3] + P3[1] * kf.Rwc[r * 3 + 1] + P3[2] * kf.Rwc[r * 3 + 2] + kf.twc[r];
      const long m = new_point(kf, Xw, ks[0]);
      for (int a = 0; a < n; a++) kf.mp_ref[ks[a]] = m, add_obs(m, kf.id, ks[a]);
    }
  }

  // ---- the local bundle adjustment of the last n_local key frames: flatten (a snapshot of the map), solve, write back
  void lba_build_into(LbaJob& J) {
    J.vision = vision;
    const int nk = (int)kfs.size(), first = std::max(0, nk - n_local);
    std::vector<int>& local = J.local;
    for (int k = first; k < nk; k++) local.push_back(k);
    std::vector<char> is_local(nk, 0);
    for (int k : local) is_local[k] = 1;
    std::vector<long>& pts = J.pts;
    {
      std::vector<char> seen(mp_bad.size(), 0);
      for (int k : local)
        for (long m : kfs[k]->mp_ref)
          if (m >= 0 && !seen[m] && !mp_bad[m]) seen[m] = 1, pts.push_back(m);
    }
    std::vector<int> fixed_ids;
    std::vector<char> is_fixed(nk, 0);
    if (!vision && first > 0) fixed_ids.push_back(first - 1), is_fixed[first - 1] = 1;  // (the inertial chain's anchor)
    for (long m : pts)
      for (const auto& kv : mp_obs[m])
        if (!is_local[kv.first] && !is_fixed[kv.first]) fixed_ids.push_back(kv.first), is_fixed[kv.first] = 1;
    std::vector<int> order(local);
    order.insert(order.end(), fixed_ids.begin(), fixed_ids.end());
    std::vector<int> index(nk, -1);
    for (size_t i = 0; i < order.size(); i++) index[order[i]] = (int)i;
    J.K.resize(order.size());
    std::memset(J.K.data(), 0, J.K.size() * sizeof(vieo_lba_keyframe));
    for (size_t i = 0; i < order.size(); i++) {
      J.K[i].nav = kfs[order[i]]->nav;
      J.K[i].fixed = (i >= local.size() || order[i] == 0) ? 1 : 0;
    }
    for (size_t j = 0; j < pts.size(); j++)
      for (const auto& kv : mp_obs[pts[j]])
        if (index[kv.first] >= 0) {
          const MFrame& k = *kfs[kv.first];
          for (int key : kv.second) {
            vieo_lba_obs o;
            o.kf = vision ? index[kv.first] : (index[kv.first] | (k.key_cam[key] << 24));
            o.mp = (int)j;
            o.u = k.keys[key].x, o.v = k.keys[key].y, o.ur = vision ? k.uright[key] : -1.0f;
            o.inv_sigma2 = inv_sigma2[k.keys[key].octave];
            J.obs.push_back(o), J.rows.push_back(Row{kv.first, key});
          }
        }
    if (!vision)
      for (int k : local)
        if (k > 0 && index[k - 1] >= 0 && kfs[k]->has_edge) {
          vieo_lba_imu_edge e;
          std::memset(&e, 0, sizeof(e));
          e.kf_i = index[k - 1], e.kf_j = index[k];
          e.dt_kf = kfs[k]->t - kfs[k - 1]->t;
          e.imu = kfs[k]->edge;
          J.edges.push_back(e);
        }
    vieo_lba_vio_params& P = J.P;
    std::memset(&P, 0, sizeof(P));
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) P.base.Rcb[r * 3 + c] = Tcb[r * 4 + c];
      P.base.tcb[r] = Tcb[r * 4 + 3];
    }
    P.base.fx = S.fx, P.base.fy = S.fy, P.base.cx = S.cx, P.base.cy = S.cy, P.base.bf = S.bf;
    if (vision) {
      P.base.its0 = 5, P.base.its1 = 10;  // src/Optimizer.cc:2179-2258
    } else {
      P.base.its0 = 4, P.base.its1 = 6;
      P.base.n_cams = nc, P.base.cams = S.rig.cams;
      std::memcpy(P.gw, GRAVITY, 24);
      P.inv_sigma_bg2 = 1.0 / (IMU_SIGMA[2] * IMU_SIGMA[2]), P.inv_sigma_ba2 = 1.0 / (IMU_SIGMA[3] * IMU_SIGMA[3]);
      P.lambda_init = 1.0;
      P.qRbe[0] = 1.0;
    }
    J.X.resize(pts.size() * 3), J.Xo.resize(pts.size() * 3);
    for (size_t j = 0; j < pts.size(); j++)
      for (int r = 0; r < 3; r++) J.X[3 * j + r] = mp_X[3 * pts[j] + r];
    J.close.assign(pts.size(), 0), J.erase.assign(std::max<size_t>(J.obs.size(), 1), 0);
    J.navs.resize(order.size());
    int n_fixed = 0;
    for (const vieo_lba_keyframe& k : J.K) n_fixed += k.fixed != 0;
    win_kfs += (long)J.K.size(), win_fixed += n_fixed, win_points += (long)pts.size(), win_obs += (long)J.obs.size();
    win_max_kfs = std::max(win_max_kfs, (int)J.K.size()), win_max_fixed = std::max(win_max_fixed, n_fixed);
  }
  // MapPoint::UpdateNormalAndDepth of the window's points from the solve's results, on the LocalMapping thread: the key
  // frames that still observe a point after the erase flags (a point's rows are contiguous: key frames ascending, a key
  // frame's keys ascending), the centres with the optimised states in place
  void lba_post(LbaJob& J) {
    J.nd_ids.clear();
    if (J.rc != 0 || J.res.status != 0) return;
    std::vector<float> centres(kfs.size() * 3);
    for (size_t k = 0; k < kfs.size(); k++)
      for (int r = 0; r < 3; r++) centres[3 * k + r] = (float)kfs[k]->twc[r];
    for (size_t i = 0; i < J.local.size(); i++)
      if (!J.K[i].fixed) {
        double twc[3];
        centre_of(J.navs[i], twc);
        for (int r = 0; r < 3; r++) centres[3 * J.local[i] + r] = (float)twc[r];
      }
    std::vector<int32_t> first(1, 0), obs_centre, ref;
    std::vector<float> pts, ref_scale;
    for (size_t r0 = 0; r0 < J.obs.size();) {
      const int j = J.obs[r0].mp;
      size_t r1 = r0;
      const size_t before = obs_centre.size();
      int ref_row = -1;
      for (; r1 < J.obs.size() && J.obs[r1].mp == j; r1++)
        if (!J.erase[r1]) {
          if (ref_row < 0) ref_row = (int)r1;
          if (obs_centre.size() == before || obs_centre.back() != J.rows[r1].kid) obs_centre.push_back(J.rows[r1].kid);
        }
      r0 = r1;
      if (obs_centre.size() == before) continue;  // every observation erased: the point goes bad at the write-back
      J.nd_ids.push_back(J.pts[j]);
      first.push_back((int32_t)obs_centre.size());
      ref.push_back(J.rows[ref_row].kid);
      ref_scale.push_back(scale[kfs[J.rows[ref_row].kid]->keys[J.rows[ref_row].key].octave]);
      for (int r = 0; r < 3; r++) pts.push_back(J.Xo[3 * j + r]);
    }
    const size_t n = J.nd_ids.size();
    if (n == 0) return;
    J.nd_nrm.resize(n * 3), J.nd_mx.resize(n), J.nd_mn.resize(n);
    const int rc = vieo_update_normal_and_depth_batch(pts.data(), first.data(), obs_centre.data(), centres.data(), (int)kfs.size(), ref.data(),
                                                      ref_scale.data(), scale[NLEVELS - 1], (int)n, J.nd_nrm.data(), J.nd_mx.data(),
                                                      J.nd_mn.data());
    if (rc != 0) J.rc = rc;
  }
  // the key-frame pair's pre-integration (LocalMapping::ProcessNewKeyFrame's in the reference, not the optimiser's: timed apart)
  static void lba_preintegrate(LbaJob* J) {  // (any host thread)
    const auto t0 = std::chrono::steady_clock::now();
    const int32_t first[2] = {0, (int32_t)J->samples.size()};
    double prv[81];
    int32_t st = 0;
    J->rc = vieo_imu_preintegrate_batch(&J->noise, J->samples.data(), first, &J->ti, &J->tj, J->bg, J->ba, 1, &J->edge, prv, &st);
    if (J->rc != 0 || st != 0) J->rc = J->rc ? J->rc : -1;
    std::memcpy(J->edge.Sigma, prv, sizeof(prv));  // mSigmaijPRV for the local BA
    J->ms_preint = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    J->edge_done = true;
  }
  static void lba_solve(LbaJob* J) {  // (any host thread)
    if (J->need_edge) {
      if (!J->edge_done) lba_preintegrate(J);
      if (J->rc != 0) return;
      J->edges.back().imu = J->edge;  // (the newest key frame's edge is the last one built)
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (J->vision)
      J->rc = vieo_local_bundle_adjustment(&J->P.base, J->K.data(), (int)J->K.size(), J->X.data(), (int)J->pts.size(), J->obs.data(),
                                           (int)J->obs.size(), nullptr, J->navs.data(), J->Xo.data(), J->erase.data(), &J->res);
    else
      J->rc = vieo_local_bundle_adjustment_vio(&J->P, J->K.data(), (int)J->K.size(), J->X.data(), J->close.data(), (int)J->pts.size(),
                                               J->obs.data(), (int)J->obs.size(), J->edges.data(), (int)J->edges.size(), nullptr,
                                               J->navs.data(), J->Xo.data(), J->erase.data(), &J->res);
    J->ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  void lba_apply(LbaJob& J) {
    if (J.rc != 0) std::fprintf(stderr, "local BA / key-frame pre-integration failed: %s\n", vieo_last_error()), std::exit(1);
    if (J.need_edge) kfs[J.edge_kf]->edge = J.edge;
    ms_lba += J.ms, ms_kf_preint += J.ms_preint;
    n_lba_applied++;
    if (J.res.status != 0) return;
    for (size_t r = 0; r < J.obs.size(); r++)  // ErasePairObs: this key's observation of the point
      if (J.erase[r]) {
        const long m = J.pts[J.obs[r].mp];
        const auto it = mp_obs[m].find(J.rows[r].kid);
        if (it != mp_obs[m].end()) {
          std::vector<int>& v = it->second;
          v.erase(std::remove(v.begin(), v.end(), J.rows[r].key), v.end());
          if (v.empty()) mp_obs[m].erase(it);
        }
        kfs[J.rows[r].kid]->mp_ref[J.rows[r].key] = -1;
        if (mp_obs[m].empty()) mp_bad[m] = 1;
      }
    for (size_t i = 0; i < J.local.size(); i++)
      if (!J.K[i].fixed) {
        kfs[J.local[i]]->nav = J.navs[i];
        pose_of(*kfs[J.local[i]]);
      }
    for (size_t j = 0; j < J.pts.size(); j++)
      for (int r = 0; r < 3; r++) mp_X[3 * J.pts[j] + r] = J.Xo[3 * j + r];
    for (size_t j = 0; j < J.nd_ids.size(); j++) {
      const long m = J.nd_ids[j];
      for (int r = 0; r < 3; r++) mp_normal[3 * m + r] = J.nd_nrm[3 * j + r];
      mp_maxd[m] = J.nd_mx[j], mp_mind[m] = J.nd_mn[j];
    }
  }
  void local_ba() {  // inline: LocalMapping before the next frame
    std::unique_ptr<LbaJob> J(new LbaJob());
    lba_build_into(*J);
    lba_solve(J.get());
    lba_post(*J);
    n_lba++;
    lba_apply(*J);
  }
  void lba_worker() {
    CHECK(vieo_lba_set_stream_priority(0));
    for (;;) {
      LbaJob* J;
      {
        std::unique_lock<std::mutex> g(lba_m);
        lba_cv.wait(g, [&] { return lba_todo || lba_quit; });
        if (lba_quit) return;
        J = lba_todo, lba_todo = nullptr;
      }
      if (J->need_edge && J->deferred) {  // the pre-integration on the helper thread, beside the flattening
        if (!pre_thread.joinable()) pre_thread = std::thread(&Replay::pre_worker, this);
        {
          std::lock_guard<std::mutex> g(pre_m);
          pre_todo = J, pre_busy = true;
        }
        pre_cv.notify_all();
      }
      if (J->deferred) lba_build_into(*J);
      if (J->need_edge && J->deferred) {
        std::unique_lock<std::mutex> g(pre_m);
        pre_cv.wait(g, [&] { return !pre_busy; });
      }
      lba_solve(J);
      lba_post(*J);
      {
        std::lock_guard<std::mutex> g(lba_m);
        lba_busy = false;
      }
      lba_cv.notify_all();
    }
  }
  void pre_worker() {
    for (;;) {
      LbaJob* J;
      {
        std::unique_lock<std::mutex> g(pre_m);
        pre_cv.wait(g, [&] { return pre_todo || pre_quit; });
        if (pre_quit) return;
        J = pre_todo, pre_todo = nullptr;
      }
      lba_preintegrate(J);
      {
        std::lock_guard<std::mutex> g(pre_m);
        pre_busy = false;
      }
      pre_cv.notify_all();
    }
  }
  void lba_submit(LbaJob* J) {
    if (!lba_thread.joinable()) lba_thread = std::thread(&Replay::lba_worker, this);
    {
      std::lock_guard<std::mutex> g(lba_m);
      lba_todo = J, lba_busy = true;
    }
    lba_cv.notify_all();
  }
  void before_frame(int k) {  // the pending write-back reaches the tracker before frame k
    if (job && k >= lba_due) {
      {
        std::unique_lock<std::mutex> g(lba_m);
        lba_cv.wait(g, [&] { return !lba_busy; });
      }
      lba_apply(*job);
      job.reset();
      map_updated = true;
    }
  }

  void local_points() {  // all points of the local key frames (rebuilt when a key frame came in / a local BA was applied)
    if (lp_key_kfs == kfs.size() && lp_key_lba == n_lba_applied) return;
    lp_key_kfs = kfs.size(), lp_key_lba = n_lba_applied;
    lp.clear();
    std::vector<char> seen(mp_bad.size(), 0);
    for (size_t k = kfs.size() > (size_t)n_local_kfs ? kfs.size() - n_local_kfs : 0; k < kfs.size(); k++)
      for (long m : kfs[k]->mp_ref)
        if (m >= 0 && !seen[m] && !mp_bad[m]) seen[m] = 1, lp.push_back(m);
    lp_pts.resize(lp.size()), lp_desc.resize(lp.size() * 32);
    for (size_t j = 0; j < lp.size(); j++) {
      const long m = lp[j];
      for (int r = 0; r < 3; r++) lp_pts[j].Xw[r] = mp_X[3 * m + r], lp_pts[j].normal[r] = mp_normal[3 * m + r];
      lp_pts[j].max_distance = mp_maxd[m], lp_pts[j].min_distance = mp_mind[m];
      std::memcpy(&lp_desc[j * 32], &mp_desc[(size_t)m * 32], 32);
    }
    local_version++;
  }

  // ---- the tracker call and the frame it returns
  void set_images(vieo_track_input& in, int k, bool next) const {
    if (vision) {
      (next ? in.next_left : in.left) = S.image(k, 0), (next ? in.next_right : in.right) = S.image(k, 1);
    } else {
      for (int c = 0; c < nc; c++) (next ? in.next_images : in.images)[c] = S.image(k, c);
    }
  }
  MFramePtr frame_of(int k, const vieo_track_output& out) const {
    MFramePtr f = std::make_shared<MFrame>();
    f->k = k, f->t = S.time(k), f->N = out.n_keys;
    const int N = out.n_keys;
    f->keys.assign(out.keys, out.keys + N), f->desc.assign(out.desc, out.desc + (size_t)N * 32);
    f->uright.assign(out.uright, out.uright + N), f->depth.assign(out.depth, out.depth + N);
    f->mp_ref.assign(N, -1), f->track_depth.assign(N, std::numeric_limits<float>::infinity());
    f->outlier.assign(N, 0);
    if (!vision) {
      for (int c = 0; c <= nc; c++) f->cam_first[c] = out.cam_first[c];
      f->key_cam.resize(N);
      for (int c = 0; c < nc; c++)
        for (int i = f->cam_first[c]; i < std::min(f->cam_first[c + 1], N); i++) f->key_cam[i] = c;
      f->n_groups = out.n_groups;
      f->group_idx.assign(out.group_idx, out.group_idx + (size_t)out.n_groups * nc);
      f->group_good.assign(out.group_good, out.group_good + out.n_groups);
      f->group_p3d.assign(out.group_p3d, out.group_p3d + (size_t)out.n_groups * 3);
    }
    return f;
  }
  // Frame::Frame of the first frame: a call without a last frame or a local map (its searches find nothing and it says so;
  // the extraction and stereo outputs are valid: include/vieo_hot.h, VIEO_TRACK_LOST / VIEO_TRACK_PREINT_FAILED)
  void initialise() {
    vieo_navstate nav;
    std::memset(&nav, 0, sizeof(nav));
    const double* tr = &S.truth[0];
    std::memcpy(nav.p, tr, 24), std::memcpy(nav.q, tr + 3, 32), std::memcpy(nav.v, tr + 7, 24);
    std::memcpy(nav.bg, S.bg, 24), std::memcpy(nav.ba, S.ba, 24);
    vieo_track_input in;
    std::memset(&in, 0, sizeof(in));
    set_images(in, 0, false);
    in.stride = S.W, in.t_ref = in.t_cur = S.time(0), in.nav_ref = nav, in.nav_last = nav;
    in.local_version = ++local_version;
    vieo_track_output out;
    CHECK(vieo_track_frame(trk, &in, &out));
    if (out.status == VIEO_TRACK_OK || (!vision && out.stereo_status != 0)) {
      std::fprintf(stderr, "first frame: unexpected status %d (stereo %d)\n", out.status, out.stereo_status);
      std::exit(1);
    }
    MFramePtr f = frame_of(0, out);
    insert_keyframe(f, nav, nullptr);
    f->nav = nav, f->has_prior = false;
    last = f, map_updated = true;
    traj.push_back(nav);
    if (vision) {
      set_ref(*f);
      for (int i = 0; i < 16; i++) velocity[i] = (i % 5 == 0) ? 1.0 : 0.0;
    }
  }
  void set_ref(MFrame& f) const {
    f.ref_kf = (int)kfs.size() - 1;
    double Tk[16], Tf[16];
    T_of(kfs[f.ref_kf]->nav, Tk), T_of(f.nav, Tf);
    inv_rigid(Tk, Tk);
    mul4(Tk, Tf, f.T_rel);
  }

  void step(int k) {
    MFrame& L = *last;
    const double t = S.time(k);
    vieo_track_input in;
    std::memset(&in, 0, sizeof(in));
    set_images(in, k, false);
    in.stride = S.W;
    vieo_navstate nav_last_v;
    int i0 = 0, ni = 0;
    if (vision) {
      // Tracking::UpdateLastFrame + the constant-velocity prediction
      double Twb_last[16], Tp[16];
      T_of(kfs[L.ref_kf]->nav, Twb_last);
      mul4(Twb_last, L.T_rel, Twb_last);
      nav_last_v = nav_of(Twb_last, L.nav);
      mul4(Twb_last, velocity, Tp);
      in.nav_ref = nav_of(Tp, L.nav), in.nav_last = nav_last_v;
      in.t_ref = L.t, in.t_cur = t;
    } else {
      const MFrame& ref = map_updated ? *kfs.back() : L;
      S.imu_between(ref.t, t, &i0, &ni);
      in.imu = S.imu.data() + i0, in.n_imu = ni;
      in.t_ref = ref.t, in.t_cur = t;
      in.nav_ref = ref.nav, in.nav_last = L.nav;
      if (!map_updated && L.has_prior) in.nav_prior = &L.prior_nav, in.H_prior = L.H_prior;
    }
    if (prefetch) {
      in.use_prefetched = prefetched ? 1 : 0;
      prefetched = k + 1 <= last_frame;
      if (prefetched) {
        set_images(in, k + 1, true);
        if (!vision) {
          // (as in examples/replay_main.cc: when LocalMapping's write-back comes in between, the next prediction starts at
          // the newest key frame with the bias the local BA gave it -- known once that solve has finished)
          int j0, nj;
          if (job && k + 1 >= lba_due) {
            bool done;
            {
              std::lock_guard<std::mutex> g(lba_m);
              done = !lba_busy;
            }
            if (done && job->rc == 0 && job->res.status == 0) {
              const MFrame& kf = *kfs.back();
              const vieo_navstate* nav = &kf.nav;
              for (size_t i = 0; i < job->local.size(); i++)
                if (job->local[i] == kf.id && !job->K[i].fixed) nav = &job->navs[i];
              std::memcpy(next_bias, nav->bg, 48);  // bg[3], ba[3] are adjacent in vieo_navstate
              S.imu_between(kf.t, S.time(k + 1), &j0, &nj);
              in.next_ref_bias = next_bias, in.next_t_ref = kf.t;
              in.next_imu = S.imu.data() + j0, in.next_n_imu = nj, in.next_t_cur = S.time(k + 1);
            }
          } else {
            S.imu_between(t, S.time(k + 1), &j0, &nj);
            in.next_imu = S.imu.data() + j0, in.next_n_imu = nj, in.next_t_cur = S.time(k + 1);
          }
        }
      }
    }
    std::vector<vieo_last_frame_point> pts(L.N);
    std::memset(pts.data(), 0, pts.size() * sizeof(vieo_last_frame_point));
    std::vector<int32_t> where(mp_bad.size(), -1);
    for (int i = 0; i < L.N; i++) {
      vieo_last_frame_point& p = pts[i];
      p.octave = L.keys[i].octave, p.angle = L.keys[i].angle;
      const long m = L.mp_ref[i];
      if (m >= 0 && !L.outlier[i] && !mp_bad[m]) {
        for (int r = 0; r < 3; r++) p.Xw[r] = mp_X[3 * m + r];
        p.flags = 3;
        std::memcpy(p.desc, &mp_desc[(size_t)m * 32], 32);
        if (where[m] < 0) where[m] = i;  // the first key that holds the point
        // the keys of one stereo group hold the same MapPoint: they name the first of them as their table entry
        if (!vision) p.reserved[0] = where[m] + 1;
      }
    }
    local_points();
    std::vector<int32_t> alias(lp.size());
    for (size_t j = 0; j < lp.size(); j++) alias[j] = where[lp[j]];
    in.n_last = L.N, in.last_points = pts.data(), in.last_track_depth = L.track_depth.data();
    in.n_local = (int)lp.size(), in.local_version = local_version;
    in.local_points = lp_pts.data(), in.local_desc = lp_desc.data(), in.local_alias = alias.data();
    vieo_track_output out;
    CHECK(vieo_track_frame(trk, &in, &out));
    if (out.status != VIEO_TRACK_OK || (!vision && out.stereo_status != 0)) {
      std::fprintf(stderr, "frame %d: tracking failed (status %d, pre-integration %d, stereo %d)\n", k, out.status, out.preint_status,
                   out.stereo_status);
      std::exit(1);
    }
    ms_track += out.ms_host, ms_gpu += out.ms_gpu, widened += out.widened;
    MFramePtr f = frame_of(k, out);
    const int N = out.n_keys, cap = out.key_cap;
    f->outlier.assign(out.outlier, out.outlier + N);
    for (int i = 0; i < N; i++) {
      const int r = out.point_ref[i];
      if (r >= 0 && r < cap)
        f->mp_ref[i] = L.mp_ref[r], f->track_depth[i] = L.track_depth[r];
      else if (r >= cap)
        f->mp_ref[i] = lp[r - cap], f->track_depth[i] = out.local_track_depth[r - cap];
    }
    f->nav = out.second.base.status == 0 ? out.second.base.nav : out.first.base.nav;
    f->has_prior = !vision && out.second.has_marg != 0;
    if (f->has_prior) f->prior_nav = f->nav, std::memcpy(f->H_prior, out.second.H_marg, sizeof(f->H_prior));
    map_updated = false;
    if (vision) {  // mVelocity = Tcw_cur * Twc_last (src/Tracking.cc:1178-1189), here between the body poses
      double Tl[16], Tf[16];
      T_of(nav_last_v, Tl), T_of(f->nav, Tf);
      inv_rigid(Tl, Tl);
      mul4(Tl, Tf, velocity);
      set_ref(*f);
    }
    finish_frame(k, f);
  }

  // NeedNewKeyFrame / CreateNewKeyFrame / LocalMapping after a tracked frame
  void finish_frame(int k, const MFramePtr& f) {
    const double t = f->t;
    if (k % kf_every == 0) {
      for (int i = 0; i < f->N; i++)
        if (f->outlier[i]) f->mp_ref[i] = -1;
      const MFrame& kp = *kfs.back();
      int i0 = 0, ni = 0;
      if (!vision) S.imu_between(kp.t, t, &i0, &ni);
      if (lba_lag <= 0) {
        if (vision) {
          insert_keyframe(f, f->nav, nullptr);
        } else {
          const int32_t first[2] = {0, ni};
          vieo_imu_preint im;
          double prv[81];
          int32_t st = 0;
          CHECK(vieo_imu_preintegrate_batch(&S.noise, S.imu.data() + i0, first, &kp.t, &t, kp.nav.bg, kp.nav.ba, 1, &im, prv, &st));
          if (st != 0) std::fprintf(stderr, "key-frame pre-integration failed\n"), std::exit(1);
          std::memcpy(im.Sigma, prv, sizeof(prv));
          insert_keyframe(f, f->nav, &im);
        }
        local_ba();  // (f IS the key frame: its state follows the optimisation, UpdateLastFrame)
        if (!vision) map_updated = true;
      } else {
        const double kp_t = kp.t;
        double kbg[3], kba[3];
        std::memcpy(kbg, kp.nav.bg, 24), std::memcpy(kba, kp.nav.ba, 24);
        vieo_imu_preint placeholder;
        std::memset(&placeholder, 0, sizeof(placeholder));
        insert_keyframe(f, f->nav, vision ? nullptr : &placeholder);
        before_frame(k + lba_lag + kf_every);  // (a job still pending is applied first)
        job.reset(new LbaJob());
        job->deferred = true;
        if (!vision) {
          job->need_edge = true, job->edge_kf = (int)kfs.size() - 1;
          job->samples.assign(S.imu.begin() + i0, S.imu.begin() + i0 + ni);
          job->noise = S.noise, job->ti = kp_t, job->tj = t;
          std::memcpy(job->bg, kbg, 24), std::memcpy(job->ba, kba, 24);
        }
        n_lba++;
        lba_due = k + lba_lag;
        lba_submit(job.get());
      }
      std::fill(f->outlier.begin(), f->outlier.end(), 0);
      if (vision) set_ref(*f);  // the key frame is its own reference: T_rel = identity
    }
    last = f;
    traj.push_back(f->nav);
  }

  std::string run_shape_json() const {
    const size_t n = frame_ms.size(), tail = std::min<size_t>(n, 200);
    double lastsum = 0;
    for (size_t i = 0; i < n; i++) lastsum += i + tail >= n ? frame_ms[i] : 0.0;
    std::vector<double> v(frame_ms);
    std::sort(v.begin(), v.end());
    char buf[512];
    std::snprintf(buf, sizeof(buf),
                  "\"ms_per_frame_last_200\": %.4f, \"ms_per_frame_median\": %.4f, \"ms_per_frame_p99\": %.4f, "
                  "\"lba_windows\": {\"mean_key_frames\": %.2f, \"mean_fixed_key_frames\": %.2f, \"max_key_frames\": %d, "
                  "\"max_fixed_key_frames\": %d, \"mean_points\": %.1f, \"mean_observations\": %.1f}",
                  tail ? lastsum / tail : 0.0, n ? v[n / 2] : 0.0, n ? v[std::min(n - 1, (size_t)(0.99 * n))] : 0.0,
                  n_lba ? (double)win_kfs / n_lba : 0.0, n_lba ? (double)win_fixed / n_lba : 0.0, win_max_kfs, win_max_fixed,
                  n_lba ? (double)win_points / n_lba : 0.0, n_lba ? (double)win_obs / n_lba : 0.0);
    return buf;
  }
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s seq.vseq [traj.bin] [--vision] [--frames N] [--warmup M] [--lba-lag L] [--prefetch 0|1] [--quiet]\n", argv[0]);
    return 2;
  }
  const char* traj_path = nullptr;
  int n_frames = -1, warmup = 0, lba_lag = 0, prefetch = 0;
  bool quiet = false, vision = false;
  for (int i = 2; i < argc; i++) {
    if (!std::strcmp(argv[i], "--frames") && i + 1 < argc)
      n_frames = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--prefetch") && i + 1 < argc)
      prefetch = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--warmup") && i + 1 < argc)
      warmup = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--lba-lag") && i + 1 < argc)
      lba_lag = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--vision"))
      vision = true;
    else if (!std::strcmp(argv[i], "--quiet"))
      quiet = true;
    else
      traj_path = argv[i];
  }
  if (!vieo_device_available()) {
    std::fprintf(stderr, "no gfx950 device: %s (there is no CPU fallback)\n", vieo_last_error());
    return 2;
  }
  Sequence S;
  if (!load_sequence(argv[1], S)) {
    std::fprintf(stderr, "cannot read %s\n", argv[1]);
    return 2;
  }
  if (vision == S.has_rig) {
    std::fprintf(stderr, "%s\n", vision ? "--vision takes a rectified pair's file (VSEQ0001)"
                                        : "a rectified pair's file: --vision here, or examples/replay_main for the stereo-inertial replay");
    return 2;
  }
  const int n = n_frames > 0 ? std::min(n_frames, S.n_frames) : S.n_frames;
  if (warmup > 1) {
    Replay Wm(S, vision);
    Wm.lba_lag = lba_lag, Wm.prefetch = prefetch != 0, Wm.last_frame = std::min(warmup, S.n_frames) - 1;
    Wm.initialise();
    for (int k = 1; k < std::min(warmup, S.n_frames); k++) Wm.before_frame(k), Wm.step(k);
    Wm.before_frame(1 << 30);
  }
  Replay R(S, vision);
  R.lba_lag = lba_lag, R.prefetch = prefetch != 0, R.last_frame = n - 1;
  R.initialise();
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 1; k < n; k++) {
    const auto tk = std::chrono::steady_clock::now();
    R.before_frame(k);
    R.step(k);
    R.frame_ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tk).count());
    if (!quiet && k % 10 == 0) {
      const double* tr = &S.truth[(size_t)k * 10];
      const vieo_navstate& v = R.traj.back();
      const double e = std::sqrt((v.p[0] - tr[0]) * (v.p[0] - tr[0]) + (v.p[1] - tr[1]) * (v.p[1] - tr[1]) + (v.p[2] - tr[2]) * (v.p[2] - tr[2]));
      std::fprintf(stderr, "frame %d: %zu key frames, %zu points, position error %.2e m\n", k, R.kfs.size(), R.mp_bad.size(), e);
    }
  }
  const double ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  R.before_frame(1 << 30);
  double emax = 0, e2 = 0;
  for (int k = 0; k < n; k++) {
    const double* tr = &S.truth[(size_t)k * 10];
    const vieo_navstate& v = R.traj[k];
    const double e = (v.p[0] - tr[0]) * (v.p[0] - tr[0]) + (v.p[1] - tr[1]) * (v.p[1] - tr[1]) + (v.p[2] - tr[2]) * (v.p[2] - tr[2]);
    e2 += e, emax = std::max(emax, std::sqrt(e));
  }
  if (traj_path) {
    FILE* f = std::fopen(traj_path, "wb");
    if (!f || std::fwrite(R.traj.data(), sizeof(vieo_navstate), R.traj.size(), f) != R.traj.size()) {
      std::fprintf(stderr, "cannot write %s\n", traj_path);
      return 1;
    }
    std::fclose(f);
  }
  const int nf = n - 1;
  std::printf("{\"mode\": \"%s\", \"cameras\": %d, \"frames\": %d, \"ms_per_frame\": %.4f, \"frames_per_s\": %.2f, \"ms_track_call\": %.4f, "
              "\"ms_track_gpu\": %.4f, \"local_bas\": %d, \"ms_per_local_ba\": %.4f, \"ms_per_key_frame_preintegration\": %.4f, \"key_frames\": %zu, \"map_points\": %zu, "
              "\"widened\": %d, \"lba_lag\": %d, \"prefetch\": %d, \"ate_rmse_vs_truth_m\": %.6e, \"max_err_vs_truth_m\": %.6e, %s}\n",
              vision ? "vision" : "rig", vision ? 2 : S.n_cams, nf, ms_total / nf, 1e3 * nf / ms_total, R.ms_track / nf, R.ms_gpu / nf, R.n_lba,
              R.n_lba_applied ? R.ms_lba / R.n_lba_applied : 0.0, R.n_lba_applied ? R.ms_kf_preint / R.n_lba_applied : 0.0, R.kfs.size(), R.mp_bad.size(), R.widened, lba_lag, prefetch, std::sqrt(e2 / n),
              emax, R.run_shape_json().c_str());
  return 0;
}
