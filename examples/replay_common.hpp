// replay_common.hpp -- what the two C++ replays share: the sequence file, the host-side map (key frames, points,
// observations: the part Map / KeyFrame / MapPoint play in the reference), Tracking::CreateNewKeyFrame, and LocalMapping's
// local bundle adjustment on its own host thread.  examples/replay_main.cc drives it with ONE vieo_track_frame call per
// frame, examples/dropin_replay.cc with the per-member entries the shims behind the reference's own signatures call.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "vieo_hot.h"

#define CHECK(call)                                                                \
  do {                                                                             \
    const int rc_ = (call);                                                        \
    if (rc_ != VIEO_OK) {                                                          \
      std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, vieo_last_error()); \
      std::exit(1);                                                                \
    }                                                                              \
  } while (0)

namespace vieo_replay {

const int NFEAT = 1200, NLEVELS = 8, INI_TH = 20, MIN_TH = 7;
const float SCALE = 1.2f;
const double GRAVITY[3] = {0.0, 0.0, -9.81};
const double IMU_SIGMA[4] = {1.6968e-4, 2.0e-3, 1.9393e-5, 3.0e-3};  // EuRoC_VIO.yaml:13-17

struct Sequence {
  int n_frames = 0, W = 0, H = 0, n_imu = 0;
  double dt = 0, t0 = 0;
  vieo_imu_noise noise;
  double bg[3], ba[3], Tbc[16], Tcb[16];
  double intr[8];  // fx, fy, cx, cy, bf, baseline, th_depth, pad as the Python driver holds them (double)
  float fx, fy, cx, cy, bf, baseline, th_depth;  // ... and as float32 where it computes in float32
  std::vector<vieo_imu_sample> imu;
  std::vector<double> truth;  // [n][10]
  std::vector<uint8_t> images;
  // "VSEQ0002" (tools/write_sequence.py write_rig_sequence): a rig of n_cams distorted cameras and its tracker's parameters
  int n_cams = 2, nfeatures = NFEAT;
  bool has_rig = false;
  vieo_tracker_params trk_params;
  vieo_tracker_rig rig;
  double time(int k) const { return t0 + k * dt; }
  const uint8_t* image(int k, int cam) const { return images.data() + ((size_t)k * n_cams + cam) * W * H; }
  // the samples PreIntegration gets for [ti, tj]: from the last one at or before ti to the first one at or after tj
  void imu_between(double ti, double tj, int* first, int* count) const {
    int a = 0, b = n_imu - 1;
    {  // searchsorted(t, ti, side="right") - 1
      int lo = 0, hi = n_imu;
      while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if (imu[mid].t <= ti) lo = mid + 1; else hi = mid;
      }
      a = std::max(lo - 1, 0);
    }
    {  // searchsorted(t, tj, side="left")
      int lo = 0, hi = n_imu;
      while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if (imu[mid].t < tj) lo = mid + 1; else hi = mid;
      }
      b = std::min(lo, n_imu - 1);
    }
    *first = a, *count = b - a + 1;
  }
};

inline bool load_sequence(const char* path, Sequence& S) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  char magic[8];
  bool ok = std::fread(magic, 1, 8, f) == 8 && (std::memcmp(magic, "VSEQ0001", 8) == 0 || std::memcmp(magic, "VSEQ0002", 8) == 0);
  S.has_rig = ok && magic[7] == '2';
  int hdr[4];
  ok = ok && std::fread(hdr, 4, 4, f) == 4;
  S.n_frames = hdr[0], S.W = hdr[1], S.H = hdr[2], S.n_imu = hdr[3];
  ok = ok && std::fread(&S.dt, 8, 1, f) == 1 && std::fread(&S.t0, 8, 1, f) == 1;
  ok = ok && std::fread(&S.noise, sizeof(S.noise), 1, f) == 1;
  ok = ok && std::fread(S.bg, 8, 3, f) == 3 && std::fread(S.ba, 8, 3, f) == 3 && std::fread(S.Tbc, 8, 16, f) == 16 &&
       std::fread(S.Tcb, 8, 16, f) == 16;
  ok = ok && std::fread(S.intr, 8, 8, f) == 8;
  if (!ok) return std::fclose(f), false;
  S.fx = (float)S.intr[0], S.fy = (float)S.intr[1], S.cx = (float)S.intr[2], S.cy = (float)S.intr[3];
  S.bf = (float)S.intr[4], S.baseline = (float)S.intr[5], S.th_depth = (float)S.intr[6];
  if (S.has_rig) {
    int32_t h2[2];
    ok = std::fread(h2, 4, 2, f) == 2 && std::fread(&S.trk_params, sizeof(S.trk_params), 1, f) == 1 &&
         std::fread(&S.rig, sizeof(S.rig), 1, f) == 1 && h2[0] >= 2 && h2[0] <= 4 && S.rig.n_cams == h2[0];
    if (!ok) return std::fclose(f), false;
    S.n_cams = h2[0], S.nfeatures = h2[1];
  }
  S.imu.resize(S.n_imu), S.truth.resize((size_t)S.n_frames * 10), S.images.resize((size_t)S.n_frames * S.n_cams * S.W * S.H);
  ok = std::fread(S.imu.data(), sizeof(vieo_imu_sample), S.n_imu, f) == (size_t)S.n_imu;
  ok = ok && std::fread(S.truth.data(), 8, S.truth.size(), f) == S.truth.size();
  ok = ok && std::fread(S.images.data(), 1, S.images.size(), f) == S.images.size();
  std::fclose(f);
  return ok;
}

inline void quat_to_R(const double* q, double* R) {  // q = (w, x, y, z), as synth_ba.quat_to_R
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z), R[1] = 2 * (x * y - z * w), R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w), R[4] = 1 - 2 * (x * x + z * z), R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w), R[7] = 2 * (y * z + x * w), R[8] = 1 - 2 * (x * x + y * y);
}

// one frame; a key frame IS its frame (the last frame and the newest key frame may be the same object, as in the
// Python driver and as mLastFrame / its reference key frame share map points in the reference)
struct Frame {
  int k = 0, id = -1, N = 0;
  double t = 0;
  std::vector<vieo_keypoint> keys;
  std::vector<uint8_t> desc;
  std::vector<float> uright, depth, track_depth;
  std::vector<long> mp_ref;
  std::vector<uint8_t> outlier;
  vieo_navstate nav;
  bool has_prior = false;
  vieo_navstate prior_nav;
  double H_prior[225];
  // key-frame part
  double Rwc[9], twc[3];
  bool has_edge = false;
  vieo_imu_preint edge;  // from the previous key frame, Sigma in (p, Phi, v) order for the local BA
};
typedef std::shared_ptr<Frame> FramePtr;
struct ReplayBase {
  const Sequence& S;
  int kf_every = 10, n_local = 10, n_local_kfs = 10;
  float th_last = 7.0f, th_local = 2.0f;
  double Tcb[16];
  float scale[16], inv_sigma2[16];
  vieo_orb *extL = nullptr, *extR = nullptr;
  // the map
  std::vector<float> mp_X, mp_normal, mp_maxd, mp_mind;
  std::vector<uint8_t> mp_desc, mp_bad;
  std::vector<std::map<int, int>> mp_obs;  // key-frame id -> key index
  std::vector<FramePtr> kfs;
  std::vector<vieo_navstate> traj;
  FramePtr last;
  bool map_updated = false;
  int n_lba = 0, n_lba_applied = 0, lba_version = 0, local_version = 0, widened = 0;
  // LocalMapping beside Tracking (src/LocalMapping.cc:113-139): the local BA of the key frame made at frame k is solved on
  // its own host thread (the library call is re-entrant, its kernels run on the bundle-adjustment stream below the
  // tracker's) and its write-back reaches the tracker before frame k + lba_lag; lba_lag = 0: inline, before frame k + 1.
  int lba_lag = 0, lba_due = -1;
  struct LbaJob;
  std::unique_ptr<LbaJob> job;
  // the LocalMapping thread: ONE persistent host thread (the library keeps its scratch buffers per calling thread; a
  // thread per job would allocate and free them every key frame)
  std::thread lba_thread;
  std::mutex lba_m;
  std::condition_variable lba_cv;
  LbaJob* lba_todo = nullptr;
  bool lba_busy = false, lba_quit = false;
  // ... and a helper of it: the key-frame pair's pre-integration (0.7 ms: one wavefront's serial chain over ~100 samples) runs
  // beside the flattening of the window instead of behind it (persistent for the same reason as the thread above)
  std::thread pre_thread;
  std::mutex pre_m;
  std::condition_variable pre_cv;
  LbaJob* pre_todo = nullptr;
  bool pre_busy = false, pre_quit = false;

  std::vector<long> lp;  // cached local-map candidates
  std::vector<vieo_frustum_point> lp_pts;
  std::vector<uint8_t> lp_desc;
  size_t lp_key_kfs = (size_t)-1;
  int lp_key_lba = -1;
  double ms_track = 0, ms_gpu = 0, ms_lba = 0, ms_kf_preint = 0, ms_lba_job = 0, ms_frames = 0;
  double ms_wait_lm = 0, ms_apply = 0;  // the write-back on the tracking thread: waiting for LocalMapping / copying its results
  int n_tracked = 0;
  // what the run looked like: wall time of every frame of the timed loop (main() fills it), and the local-BA windows'
  // shapes (key frames, fixed ones among them, points, observations) -- a steady-state run has 10 free key frames plus
  // the fixed observers of their points
  std::vector<double> frame_ms;
  long win_kfs = 0, win_fixed = 0, win_points = 0, win_obs = 0;
  int win_max_kfs = 0, win_max_fixed = 0;
  // JSON fragment shared by the two programs' result lines
  std::string run_shape_json() const {
    const size_t n = frame_ms.size(), tail = std::min<size_t>(n, 200);
    double all = 0, last = 0;
    for (size_t i = 0; i < n; i++) all += frame_ms[i], last += i + tail >= n ? frame_ms[i] : 0.0;
    std::vector<double> v(frame_ms);
    std::sort(v.begin(), v.end());
    char buf[512];
    std::snprintf(buf, sizeof(buf),
                  "\"ms_per_frame_last_200\": %.4f, \"frames_in_last_200\": %zu, \"ms_per_frame_median\": %.4f, \"ms_per_frame_p99\": %.4f, "
                  "\"lba_windows\": {\"mean_key_frames\": %.2f, \"mean_fixed_key_frames\": %.2f, \"max_key_frames\": %d, "
                  "\"max_fixed_key_frames\": %d, \"mean_points\": %.1f, \"mean_observations\": %.1f}",
                  tail ? last / tail : 0.0, tail, n ? v[n / 2] : 0.0, n ? v[std::min(n - 1, (size_t)(0.99 * n))] : 0.0,
                  n_lba ? (double)win_kfs / n_lba : 0.0, n_lba ? (double)win_fixed / n_lba : 0.0, win_max_kfs, win_max_fixed,
                  n_lba ? (double)win_points / n_lba : 0.0, n_lba ? (double)win_obs / n_lba : 0.0);
    return buf;
  }

  explicit ReplayBase(const Sequence& s) : S(s) {
    std::memcpy(Tcb, S.Tcb, sizeof(Tcb));
    CHECK(vieo_orb_create(&extL, NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH));
    CHECK(vieo_orb_create(&extR, NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH));
    CHECK(vieo_orb_scale_factors(extL, scale));
    for (int l = 0; l < NLEVELS; l++) inv_sigma2[l] = 1.0f / (scale[l] * scale[l]);
  }
  virtual ~ReplayBase() {
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
    vieo_orb_destroy(extL), vieo_orb_destroy(extR);
  }
  ReplayBase(const ReplayBase&) = delete;
  ReplayBase& operator=(const ReplayBase&) = delete;

  void pose_of(Frame& f) {  // Twc = Twb Tbc
    double Rwb[9];
    quat_to_R(f.nav.q, Rwb);
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) f.Rwc[r * 3 + c] = Rwb[r * 3] * S.Tbc[c] + Rwb[r * 3 + 1] * S.Tbc[4 + c] + Rwb[r * 3 + 2] * S.Tbc[8 + c];
      f.twc[r] = f.nav.p[r] + (Rwb[r * 3] * S.Tbc[3] + Rwb[r * 3 + 1] * S.Tbc[7] + Rwb[r * 3 + 2] * S.Tbc[11]);
    }
  }
  void centre_of(const vieo_navstate& nav, double* twc) const {  // (pose_of's translation, for a state not written back yet)
    double Rwb[9];
    quat_to_R(nav.q, Rwb);
    for (int r = 0; r < 3; r++) twc[r] = nav.p[r] + (Rwb[r * 3] * S.Tbc[3] + Rwb[r * 3 + 1] * S.Tbc[7] + Rwb[r * 3 + 2] * S.Tbc[11]);
  }

  // Frame::Frame of the first frame: the two extractions and ComputeStereoMatches through the stage entries
  FramePtr make_frame0() {
    FramePtr f = std::make_shared<Frame>();
    f->k = 0, f->t = S.time(0);
    const int cap = vieo_orb_max_keypoints(extL);
    std::vector<vieo_keypoint> kl(cap), kr(cap);
    std::vector<uint8_t> dl((size_t)cap * 32), dr((size_t)cap * 32);
    int nl = 0, nr = 0, mono = 0;
    CHECK(vieo_orb_extract(extL, S.image(0, 0), S.W, S.H, S.W, nullptr, kl.data(), dl.data(), cap, &nl, &mono));
    CHECK(vieo_orb_extract(extR, S.image(0, 1), S.W, S.H, S.W, nullptr, kr.data(), dr.data(), cap, &nr, &mono));
    f->N = nl;
    f->keys.assign(kl.begin(), kl.begin() + nl), f->desc.assign(dl.begin(), dl.begin() + (size_t)nl * 32);
    f->uright.resize(nl), f->depth.resize(nl);
    CHECK(vieo_stereo_match_rectified(extL, extR, kl.data(), dl.data(), nl, kr.data(), dr.data(), nr, S.baseline, S.bf,
                                      f->uright.data(), f->depth.data()));
    f->mp_ref.assign(nl, -1), f->track_depth.assign(nl, std::numeric_limits<float>::infinity()), f->outlier.assign(nl, 0);
    return f;
  }

  // Tracking::CreateNewKeyFrame + LocalMapping::ProcessNewKeyFrame
  void insert_keyframe(const FramePtr& f, const vieo_navstate& nav, const vieo_imu_preint* edge) {
    Frame& kf = *f;
    kf.id = (int)kfs.size();
    kf.nav = nav;
    pose_of(kf);
    kf.has_edge = edge != nullptr;
    if (edge) kf.edge = *edge;
    kfs.push_back(f);
    for (int i = 0; i < kf.N; i++)  // AddObservation
      if (kf.mp_ref[i] >= 0) mp_obs[kf.mp_ref[i]][kf.id] = i;
    // new points from stereo: keys with depth and without a point, nearest first; all close ones, at least 100
    std::vector<int> cand;
    for (int i = 0; i < kf.N; i++)
      if (kf.depth[i] > 0 && kf.mp_ref[i] < 0) cand.push_back(i);
    std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return kf.depth[a] < kf.depth[b]; });
    int n_close = 0;
    for (int i : cand) n_close += kf.depth[i] <= S.th_depth;
    const int n_take = std::max(n_close, std::min(100, (int)cand.size()));
    cand.resize(n_take);
    for (int i : cand) {
      const double z = (double)kf.depth[i];
      // float32 differences first, as the driver's numpy expression evaluates them
      const double Xc[3] = {(double)(kf.keys[i].x - S.cx) * z / S.intr[0], (double)(kf.keys[i].y - S.cy) * z / S.intr[1], z};
      double Xw[3], d[3];
      for (int r = 0; r < 3; r++) Xw[r] = Xc[0] * kf.Rwc[r * 3] + Xc[1] * kf.Rwc[r * 3 + 1] + Xc[2] * kf.Rwc[r * 3 + 2] + kf.twc[r];
      for (int r = 0; r < 3; r++) d[r] = Xw[r] - kf.twc[r];
      const double dist = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      const long m = (long)mp_bad.size();
      for (int r = 0; r < 3; r++) mp_X.push_back((float)Xw[r]), mp_normal.push_back((float)(d[r] / dist));
      mp_desc.insert(mp_desc.end(), kf.desc.begin() + (size_t)i * 32, kf.desc.begin() + (size_t)(i + 1) * 32);
      mp_bad.push_back(0);
      mp_obs.emplace_back();
      mp_obs.back()[kf.id] = i;
      const float maxd = (float)(dist * (double)scale[kf.keys[i].octave]);
      mp_maxd.push_back(maxd), mp_mind.push_back(maxd / scale[NLEVELS - 1]);
      kf.mp_ref[i] = m;
    }
  }

  // MapPoint::UpdateNormalAndDepth for the given points: observations in key-frame order, reference = the oldest observer
  void update_normal_depth(const std::vector<long>& ids_in) {
    std::vector<long> ids;
    for (long m : ids_in)
      if (!mp_bad[m] && !mp_obs[m].empty()) ids.push_back(m);
    if (ids.empty()) return;
    std::vector<int32_t> first(ids.size() + 1, 0), obs_centre, ref(ids.size());
    std::vector<float> pts(ids.size() * 3), ref_scale(ids.size()), centres(kfs.size() * 3);
    for (size_t j = 0; j < ids.size(); j++) {
      const auto& ob = mp_obs[ids[j]];
      for (const auto& kv : ob) obs_centre.push_back(kv.first);
      first[j + 1] = (int32_t)obs_centre.size();
      ref[j] = ob.begin()->first;
      ref_scale[j] = scale[kfs[ob.begin()->first]->keys[ob.begin()->second].octave];
      for (int r = 0; r < 3; r++) pts[3 * j + r] = mp_X[3 * ids[j] + r];
    }
    for (size_t k = 0; k < kfs.size(); k++)
      for (int r = 0; r < 3; r++) centres[3 * k + r] = (float)kfs[k]->twc[r];
    std::vector<float> nrm(ids.size() * 3), mx(ids.size()), mn(ids.size());
    CHECK(vieo_update_normal_and_depth_batch(pts.data(), first.data(), obs_centre.data(), centres.data(), (int)kfs.size(),
                                             ref.data(), ref_scale.data(), scale[NLEVELS - 1], (int)ids.size(), nrm.data(),
                                             mx.data(), mn.data()));
    for (size_t j = 0; j < ids.size(); j++) {
      for (int r = 0; r < 3; r++) mp_normal[3 * ids[j] + r] = nrm[3 * j + r];
      mp_maxd[ids[j]] = mx[j], mp_mind[ids[j]] = mn[j];
    }
  }

  // Optimizer::LocalBundleAdjustmentNavStatePRV on the last n_local key frames: the flattened problem (a snapshot of the
  // map when the key frame was made), the solve, the write-back
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
    // LocalMapping's work around the solve, on its thread: the window is flattened there (deferred) and
    // MapPoint::UpdateNormalAndDepth of its points is computed there from the results; the write-back only copies
    bool deferred = false;
    std::vector<long> nd_ids;
    std::vector<float> nd_nrm, nd_mx, nd_mn;
    vieo_lba_result res;
    double ms = 0, ms_preint = 0, ms_job = 0;  // the local BA call; the key-frame pair's pre-integration; the whole job on the LocalMapping thread
    int rc = 0;
    // the newest key frame's inertial edge, pre-integrated on the LocalMapping thread too (only the local BA reads it)
    bool need_edge = false, edge_done = false;  // edge_done: the pre-integration below has run (on the helper thread)
    int edge_kf = -1;
    std::vector<vieo_imu_sample> samples;
    vieo_imu_noise noise;
    double ti = 0, tj = 0, bg[3], ba[3];
    vieo_imu_preint edge;
  };
  std::unique_ptr<LbaJob> lba_build() {
    std::unique_ptr<LbaJob> Jp(new LbaJob());
    lba_build_into(*Jp);
    return Jp;
  }
  // (reads the map: on the LocalMapping thread only between a key frame's insertion and its write-back, when the tracking
  // thread does not write to it)
  void lba_build_into(LbaJob& J) {
    const int nk = (int)kfs.size(), first = std::max(0, nk - n_local);
    std::vector<int>& local = J.local;
    for (int k = first; k < nk; k++) local.push_back(k);
    std::vector<char> is_local(nk, 0);
    for (int k : local) is_local[k] = 1;
    std::vector<long>& pts = J.pts;
    {
      std::vector<char> seen(mp_bad.size(), 0);
      for (int k : local)  // lLocalMapPoints: key frames oldest first, keys in order
        for (long m : kfs[k]->mp_ref)
          if (m >= 0 && !seen[m] && !mp_bad[m]) seen[m] = 1, pts.push_back(m);
    }
    std::vector<int> fixed_ids;
    std::vector<char> is_fixed(nk, 0);
    if (first > 0) fixed_ids.push_back(first - 1), is_fixed[first - 1] = 1;
    for (long m : pts)
      for (const auto& kv : mp_obs[m])
        if (!is_local[kv.first] && !is_fixed[kv.first]) fixed_ids.push_back(kv.first), is_fixed[kv.first] = 1;
    std::vector<int> order(local);
    order.insert(order.end(), fixed_ids.begin(), fixed_ids.end());
    std::vector<int> index(nk, -1);
    for (size_t i = 0; i < order.size(); i++) index[order[i]] = (int)i;
    std::vector<vieo_lba_keyframe>& K = J.K;
    K.resize(order.size());
    std::memset(K.data(), 0, K.size() * sizeof(vieo_lba_keyframe));
    for (size_t i = 0; i < order.size(); i++) {
      K[i].nav = kfs[order[i]]->nav;
      K[i].fixed = (i >= local.size() || order[i] == 0) ? 1 : 0;
    }
    for (size_t j = 0; j < pts.size(); j++)
      for (const auto& kv : mp_obs[pts[j]])
        if (index[kv.first] >= 0) {
          const Frame& k = *kfs[kv.first];
          vieo_lba_obs o;
          o.kf = index[kv.first], o.mp = (int)j;
          o.u = k.keys[kv.second].x, o.v = k.keys[kv.second].y, o.ur = k.uright[kv.second];
          o.inv_sigma2 = inv_sigma2[k.keys[kv.second].octave];
          J.obs.push_back(o), J.rows.push_back(Row{kv.first, kv.second});
        }
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
    P.base.its0 = 4, P.base.its1 = 6;
    std::memcpy(P.gw, GRAVITY, 24);
    P.inv_sigma_bg2 = 1.0 / (IMU_SIGMA[2] * IMU_SIGMA[2]), P.inv_sigma_ba2 = 1.0 / (IMU_SIGMA[3] * IMU_SIGMA[3]);
    P.lambda_init = 1.0;
    P.qRbe[0] = 1.0;
    J.X.resize(pts.size() * 3), J.Xo.resize(pts.size() * 3);
    for (size_t j = 0; j < pts.size(); j++)
      for (int r = 0; r < 3; r++) J.X[3 * j + r] = mp_X[3 * pts[j] + r];
    J.close.assign(pts.size(), 0), J.erase.assign(std::max<size_t>(J.obs.size(), 1), 0);
    J.navs.resize(order.size());
    int n_fixed = 0;
    for (const vieo_lba_keyframe& k : K) n_fixed += k.fixed != 0;
    win_kfs += (long)K.size(), win_fixed += n_fixed, win_points += (long)pts.size(), win_obs += (long)J.obs.size();
    win_max_kfs = std::max(win_max_kfs, (int)K.size()), win_max_fixed = std::max(win_max_fixed, n_fixed);
  }
  // MapPoint::UpdateNormalAndDepth of the window's points from the solve's results, before they are written back: the
  // observations that survive the erase flags (a point's rows are contiguous, key frames ascending like the map's
  // std::map), the key frames' centres with the optimised states in place
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
          obs_centre.push_back(J.rows[r1].kid);
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
    J->rc = vieo_local_bundle_adjustment_vio(&J->P, J->K.data(), (int)J->K.size(), J->X.data(), J->close.data(), (int)J->pts.size(),
                                             J->obs.data(), (int)J->obs.size(), J->edges.data(), (int)J->edges.size(), nullptr,
                                             J->navs.data(), J->Xo.data(), J->erase.data(), &J->res);
    J->ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  void lba_apply(LbaJob& J) {
    if (J.rc != 0) std::fprintf(stderr, "local BA / key-frame pre-integration failed: %s\n", vieo_last_error()), std::exit(1);
    if (J.need_edge) kfs[J.edge_kf]->edge = J.edge;
    ms_lba += J.ms, ms_kf_preint += J.ms_preint, ms_lba_job += J.ms_job;
    n_lba_applied++;
    if (J.res.status != 0) return;
    for (size_t r = 0; r < J.obs.size(); r++)  // ErasePairObs
      if (J.erase[r]) {
        const long m = J.pts[J.obs[r].mp];
        mp_obs[m].erase(J.rows[r].kid);
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
    for (size_t j = 0; j < J.nd_ids.size(); j++) {  // (lba_post computed them on the LocalMapping thread)
      const long m = J.nd_ids[j];
      for (int r = 0; r < 3; r++) mp_normal[3 * m + r] = J.nd_nrm[3 * j + r];
      mp_maxd[m] = J.nd_mx[j], mp_mind[m] = J.nd_mn[j];
    }
  }
  void local_ba() {  // inline: LocalMapping before the next frame
    std::unique_ptr<LbaJob> J = lba_build();
    lba_solve(J.get());
    lba_post(*J);
    n_lba++;
    lba_apply(*J);
  }
  void lba_worker() {
    // one window beside the tracker: not behind everything else on the device (VIEO_REPLAY_LBA_PRIORITY=-1 / 1: A/B runs)
    const char* pe = std::getenv("VIEO_REPLAY_LBA_PRIORITY");
    CHECK(vieo_lba_set_stream_priority(pe ? std::atoi(pe) : 0));
    for (;;) {
      LbaJob* J;
      {
        std::unique_lock<std::mutex> g(lba_m);
        lba_cv.wait(g, [&] { return lba_todo || lba_quit; });
        if (lba_quit) return;
        J = lba_todo, lba_todo = nullptr;
      }
      const auto tj = std::chrono::steady_clock::now();
      if (J->need_edge && J->deferred) {  // the pre-integration beside the flattening
        if (!pre_thread.joinable()) pre_thread = std::thread(&ReplayBase::pre_worker, this);
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
      J->ms_job = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tj).count();
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
    if (!lba_thread.joinable()) lba_thread = std::thread(&ReplayBase::lba_worker, this);
    {
      std::lock_guard<std::mutex> g(lba_m);
      lba_todo = J, lba_busy = true;
    }
    lba_cv.notify_all();
  }
  // the pending write-back reaches the tracker before frame k
  void before_frame(int k) {
    if (job && k >= lba_due) {
      const auto t0 = std::chrono::steady_clock::now();
      {
        std::unique_lock<std::mutex> g(lba_m);
        lba_cv.wait(g, [&] { return !lba_busy; });
      }
      const auto t1 = std::chrono::steady_clock::now();
      lba_apply(*job);
      job.reset();  // (frees the job's arrays)
      map_updated = true;
      ms_wait_lm += std::chrono::duration<double, std::milli>(t1 - t0).count();
      ms_apply += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    }
  }

  void initialise() {
    FramePtr f = make_frame0();
    vieo_navstate nav;
    std::memset(&nav, 0, sizeof(nav));
    const double* tr = &S.truth[0];
    std::memcpy(nav.p, tr, 24), std::memcpy(nav.q, tr + 3, 32), std::memcpy(nav.v, tr + 7, 24);
    std::memcpy(nav.bg, S.bg, 24), std::memcpy(nav.ba, S.ba, 24);  // IMU initialisation done: biases known at the start
    insert_keyframe(f, nav, nullptr);
    f->nav = nav, f->has_prior = false;
    std::fill(f->outlier.begin(), f->outlier.end(), 0);
    last = f, map_updated = true;
    traj.push_back(nav);
  }

  void local_points() {  // all points of the local key frames (rebuilt when a key frame came in / a local BA ran)
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
  // NeedNewKeyFrame / CreateNewKeyFrame / LocalMapping after a tracked frame (shared by both replays): a key frame every
  // kf_every frames, its local BA inline (lba_lag = 0) or on the LocalMapping thread
  void finish_frame(int k, const FramePtr& f) {
    const double t = f->t;
    int i0, ni;
    if (k % kf_every == 0) {
      const Frame& kp = *kfs.back();
      S.imu_between(kp.t, t, &i0, &ni);
      for (int i = 0; i < f->N; i++)
        if (f->outlier[i]) f->mp_ref[i] = -1;
      if (lba_lag <= 0) {
        const int32_t first[2] = {0, ni};
        vieo_imu_preint im;
        double prv[81];
        int32_t st = 0;
        CHECK(vieo_imu_preintegrate_batch(&S.noise, S.imu.data() + i0, first, &kp.t, &t, kp.nav.bg, kp.nav.ba, 1, &im, prv, &st));
        if (st != 0) std::fprintf(stderr, "key-frame pre-integration failed\n"), std::exit(1);
        std::memcpy(im.Sigma, prv, sizeof(prv));  // mSigmaijPRV for the local BA
        insert_keyframe(f, f->nav, &im);
        local_ba();
        f->nav = kfs.back()->nav;  // mLastFrame follows its reference key frame (UpdateLastFrame)
        map_updated = true;
      } else {
        // LocalMapping's work goes to its thread whole: the key-frame-to-key-frame pre-integration (an input of the local
        // BA only) and the solve; the edge is stored with the write-back
        const double kp_t = kp.t;
        double kbg[3], kba[3];
        std::memcpy(kbg, kp.nav.bg, 24), std::memcpy(kba, kp.nav.ba, 24);
        vieo_imu_preint placeholder;
        std::memset(&placeholder, 0, sizeof(placeholder));
        insert_keyframe(f, f->nav, &placeholder);
        before_frame(k + lba_lag + kf_every);  // (a job still pending is applied first)
        job.reset(new LbaJob());
        job->deferred = true;  // (flattened on the LocalMapping thread: nothing writes to the map until the write-back)
        job->need_edge = true, job->edge_kf = (int)kfs.size() - 1;
        job->samples.assign(S.imu.begin() + i0, S.imu.begin() + i0 + ni);
        job->noise = S.noise, job->ti = kp_t, job->tj = t;
        std::memcpy(job->bg, kbg, 24), std::memcpy(job->ba, kba, 24);
        n_lba++;
        lba_due = k + lba_lag;
        lba_submit(job.get());
      }
      std::fill(f->outlier.begin(), f->outlier.end(), 0);
    }
    last = f;
    traj.push_back(f->nav);
  }
};

}  // namespace vieo_replay
