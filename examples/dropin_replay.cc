// dropin_replay.cc -- the sequential stereo-inertial replay driven ONLY through the entries that sit behind the reference's
// own class members, called in Tracking's order -- the path a maintainer gets by linking shim/*.cc and include/vieo_shim.hpp
// WITHOUT touching Tracking.cc / LocalMapping.cc (north_star: "so Tracking/LocalMapping drop it in unchanged"):
//
//   Frame::Frame                    ExtractORB per camera, each on its own host thread (src/Frame.cc:259-278)
//                                     -> ORBextractor::operator()            = vieo_orb_extract               (vieo_shim.hpp)
//                                   ComputeStereoMatches (src/Frame.cc:451-611) = vieo_stereo_match_rectified[_resident] (shim/Frame_hot.cc)
//   Tracking::TrackWithIMU          FrameBase::PreIntegration (src/Tracking.cc:385-391) = vieo_imu_preintegrate_batch
//     (src/Tracking.cc:261-378)     PredictNavStateByIMU: host arithmetic, as in the reference (:392-451)
//                                   ORBmatcher::SearchByProjection(Frame&, const Frame&, th, ...) (:296)
//                                     = vieo_search_by_projection_last_frame_resident | vieo_sbp_project_last_frame + vieo_search_by_projection
//                                   Optimizer::PoseOptimization<Frame>(..., bComputeMarg = false) (:321) = vieo_pose_optimization_vio
//   Tracking::TrackLocalMapWithIMU  SearchLocalPoints (:2308-2370): Frame::isInFrustum = vieo_is_in_frustum_batch, then
//     (src/Tracking.cc:453-488)     ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th) = vieo_search_by_projection[_resident]
//                                   Optimizer::PoseOptimization<...>(..., bComputeMarg = true) (:475,479) = vieo_pose_optimization_vio
//   LocalMapping::Run               Optimizer::LocalBundleAdjustmentNavStatePRV (src/LocalMapping.cc:122-136) on its own host thread
//
// --resident 1 (default): the shims' fast form -- a frame's keys / descriptors / pyramid / uright / window grid stay in the
// extractor handles the Frame points at, every call uploads only what the pointer graph forces and synchronises once
// (include/vieo_hot.h, "the resident frame").  --resident 0: the round-4 form, every call re-uploads the frame.
// Same sequence file, same map logic and same output as examples/replay_main.cc (which needs Tracking::Track rewritten
// around vieo_track_frame); the Python twin is replay.Replay with HipStages, the oracle twin tests/replay_oracle.py.
//
//   ./examples/dropin_replay seq.vseq [traj.bin] [--frames N] [--warmup M] [--lba-lag L] [--resident 0|1] [--quiet]
// Built by __graft_entry__.build().
#include "replay_common.hpp"

using namespace vieo_replay;

namespace {

typedef std::chrono::steady_clock Clock;
inline double ms_since(const Clock::time_point& t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

// Twc = Twb Tbc, Tcw = Twc^-1 as a row-major 3 x 4
void Tcw_of(const vieo_navstate& nav, const double* Tbc, double* Tcw) {
  double Rwb[9], Rwc[9], twc[3];
  quat_to_R(nav.q, Rwb);
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) Rwc[r * 3 + c] = Rwb[r * 3] * Tbc[c] + Rwb[r * 3 + 1] * Tbc[4 + c] + Rwb[r * 3 + 2] * Tbc[8 + c];
    twc[r] = nav.p[r] + (Rwb[r * 3] * Tbc[3] + Rwb[r * 3 + 1] * Tbc[7] + Rwb[r * 3 + 2] * Tbc[11]);
  }
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) Tcw[r * 4 + c] = Rwc[c * 3 + r];
    Tcw[r * 4 + 3] = -(Rwc[r] * twc[0] + Rwc[3 + r] * twc[1] + Rwc[6 + r] * twc[2]);
  }
}

void mat3_mul(const double* A, const double* B, double* C) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}
void mat3_vec(const double* A, const double* x, double* y) {
  for (int r = 0; r < 3; r++) y[r] = A[r * 3] * x[0] + A[r * 3 + 1] * x[1] + A[r * 3 + 2] * x[2];
}
void so3_exp(const double* w, double* R) {
  const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  double q[4] = {1, 0, 0, 0};
  if (th >= 1e-12) {
    const double s = std::sin(th / 2) / th;
    q[0] = std::cos(th / 2), q[1] = s * w[0], q[2] = s * w[1], q[3] = s * w[2];
  }
  quat_to_R(q, R);
}
void R_to_quat(const double* R, double* q) {
  const double t = R[0] + R[4] + R[8];
  if (t > 0) {
    const double s = std::sqrt(t + 1.0) * 2;
    q[0] = 0.25 * s, q[1] = (R[7] - R[5]) / s, q[2] = (R[2] - R[6]) / s, q[3] = (R[3] - R[1]) / s;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    const double s = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0) * 2;
    q[0] = (R[k * 3 + j] - R[j * 3 + k]) / s;
    q[1 + i] = 0.25 * s;
    q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) / s;
    q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) / s;
  }
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int a = 0; a < 4; a++) q[a] /= n;
}

// Tracking::PredictNavStateByIMU (src/Tracking.cc:385-451; IMUPreIntegratorBase members as vieo_imu_preint)
vieo_navstate predict(const vieo_navstate& ref, const vieo_imu_preint& im) {
  vieo_navstate ns = ref;
  const double dt = im.dt;
  double Rwb[9], a[3], b[3], c[3], s[3];
  quat_to_R(ns.q, Rwb);
  mat3_vec(im.Jgp, ns.dbg, a), mat3_vec(im.Jap, ns.dba, b);
  for (int r = 0; r < 3; r++) s[r] = im.pij[r] + a[r] + b[r];
  mat3_vec(Rwb, s, c);
  double p[3], v[3];
  for (int r = 0; r < 3; r++) p[r] = ns.p[r] + ns.v[r] * dt + GRAVITY[r] * (dt * dt / 2) + c[r];
  mat3_vec(im.Jgv, ns.dbg, a), mat3_vec(im.Jav, ns.dba, b);
  for (int r = 0; r < 3; r++) s[r] = im.vij[r] + a[r] + b[r];
  mat3_vec(Rwb, s, c);
  for (int r = 0; r < 3; r++) v[r] = ns.v[r] + GRAVITY[r] * dt + c[r];
  double w[3], E[9], RR[9], R[9];
  mat3_vec(im.JgR, ns.dbg, w);
  so3_exp(w, E);
  mat3_mul(Rwb, im.Rij, RR), mat3_mul(RR, E, R);
  for (int r = 0; r < 3; r++) ns.p[r] = p[r], ns.v[r] = v[r];
  R_to_quat(R, ns.q);
  for (int r = 0; r < 3; r++) ns.bg[r] += ns.dbg[r], ns.ba[r] += ns.dba[r], ns.dbg[r] = 0, ns.dba[r] = 0;
  return ns;
}

enum Stage { ST_EXTRACT, ST_STEREO, ST_PREINT, ST_SBP_LAST, ST_POSE1, ST_FRUSTUM, ST_SBP_LOCAL, ST_POSE2, ST_N };
const char* kStageName[ST_N] = {"extract", "stereo", "preintegrate", "sbp_last_frame", "pose1", "is_in_frustum", "sbp_local_map", "pose2"};

struct Dropin : ReplayBase {
  bool resident = true;
  double ms_stage[ST_N] = {0};
  vieo_camera pinhole;
  // reusable host arrays of a frame (the members a Frame / ORBmatcher call would touch)
  std::vector<vieo_last_frame_point> pts;
  std::vector<vieo_proj_query> q1, q2;
  std::vector<int32_t> a1, a2, owner, idx;
  std::vector<uint8_t> taken, outl;
  std::vector<vieo_pose_obs> obs;
  std::vector<vieo_frustum_point> fp;
  std::vector<vieo_track_info> info;
  std::vector<long> cand;

  explicit Dropin(const Sequence& s) : ReplayBase(s) {
    std::memset(&pinhole, 0, sizeof(pinhole));
    pinhole.fx = S.fx, pinhole.fy = S.fy, pinhole.cx = S.cx, pinhole.cy = S.cy;
  }

  // Frame::Frame (src/Frame.cc:225-320)
  FramePtr make_frame(int k) {
    FramePtr f = std::make_shared<Frame>();
    f->k = k, f->t = S.time(k);
    const int cap = vieo_orb_max_keypoints(extL);
    std::vector<vieo_keypoint> kr(cap);
    std::vector<uint8_t> dr((size_t)cap * 32);
    f->keys.resize(cap), f->desc.resize((size_t)cap * 32);
    int nl = 0, nr = 0, rcl = 0, rcr = 0;
    auto t0 = Clock::now();
    {  // one host thread per camera, as Frame.cc:259-278
      int mono_l = 0, mono_r = 0;
      std::thread tl([&] { rcl = vieo_orb_extract(extL, S.image(k, 0), S.W, S.H, S.W, nullptr, f->keys.data(), f->desc.data(), cap, &nl, &mono_l); });
      std::thread tr([&] { rcr = vieo_orb_extract(extR, S.image(k, 1), S.W, S.H, S.W, nullptr, kr.data(), dr.data(), cap, &nr, &mono_r); });
      tl.join(), tr.join();
    }
    if (rcl != VIEO_OK || rcr != VIEO_OK) std::fprintf(stderr, "vieo_orb_extract failed: %s\n", vieo_last_error()), std::exit(1);
    ms_stage[ST_EXTRACT] += ms_since(t0);
    f->N = nl;
    f->keys.resize(nl), f->desc.resize((size_t)nl * 32);
    f->uright.resize(nl), f->depth.resize(nl);
    t0 = Clock::now();
    if (resident && vieo_orb_holds(extL, f->keys.data(), nl) && vieo_orb_holds(extR, kr.data(), nr))
      CHECK(vieo_stereo_match_rectified_resident(extL, extR, S.baseline, S.bf, f->uright.data(), f->depth.data()));
    else
      CHECK(vieo_stereo_match_rectified(extL, extR, f->keys.data(), f->desc.data(), nl, kr.data(), dr.data(), nr, S.baseline, S.bf,
                                        f->uright.data(), f->depth.data()));
    ms_stage[ST_STEREO] += ms_since(t0);
    f->mp_ref.assign(nl, -1), f->track_depth.assign(nl, std::numeric_limits<float>::infinity()), f->outlier.assign(nl, 0);
    return f;
  }

  void vio_frame(vieo_vio_frame& F, const vieo_navstate& nav, const vieo_navstate& ref_nav, const vieo_imu_preint& im, const Frame* prior,
                 double dt_frames, int marg, int n_obs) {
    std::memset(&F, 0, sizeof(F));
    F.base.nav = nav;
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) F.base.Rcb[r * 3 + c] = Tcb[r * 4 + c];
      F.base.tcb[r] = Tcb[r * 4 + 3];
    }
    F.base.fx = S.fx, F.base.fy = S.fy, F.base.cx = S.cx, F.base.cy = S.cy, F.base.bf = S.bf;
    F.base.n_obs = n_obs;
    F.nav_last = ref_nav;
    F.imu = im;
    std::memcpy(F.gw, GRAVITY, 24);
    F.inv_sigma_bg2 = 1.0 / (IMU_SIGMA[2] * IMU_SIGMA[2]), F.inv_sigma_ba2 = 1.0 / (IMU_SIGMA[3] * IMU_SIGMA[3]);
    F.dt_frames = dt_frames, F.th_depth = S.th_depth;
    F.compute_marg = marg;
    if (prior) F.nav_prior = prior->prior_nav, std::memcpy(F.H_prior, prior->H_prior, sizeof(F.H_prior)), F.last_has_prior = 1;
  }

  // what PoseOptimization reads from the frame (Optimizer.h:403-498): one observation per key that holds a map point
  void build_obs(const Frame& f) {
    obs.clear(), idx.clear();
    const float close = std::max(10.0f, S.th_depth);
    for (int i = 0; i < f.N; i++) {
      const long m = f.mp_ref[i];
      if (m < 0) continue;
      vieo_pose_obs o;
      for (int r = 0; r < 3; r++) o.Xw[r] = mp_X[3 * m + r];
      o.u = f.keys[i].x, o.v = f.keys[i].y, o.ur = f.uright[i];
      o.inv_sigma2 = inv_sigma2[f.keys[i].octave];
      o.flags = f.track_depth[i] < close ? 1 : 0;
      obs.push_back(o), idx.push_back(i);
    }
    outl.assign(std::max<size_t>(obs.size(), 1), 0);
  }

  // one frame: Tracking::Track for the stereo-inertial steady state, member by member
  void step(int k) {
    const auto t_frame = Clock::now();
    Frame& L = *last;
    FramePtr fp_ = make_frame(k);
    Frame& f = *fp_;
    const Frame& ref = map_updated ? *kfs.back() : L;  // Tracking.cc:392-409
    const vieo_navstate ref_nav = ref.nav;
    const Frame* prior = (!map_updated && L.has_prior) ? &L : nullptr;
    const double t_ref = ref.t;
    // ---- FrameBase::PreIntegration + PredictNavStateByIMU
    auto t0 = Clock::now();
    int i0, ni;
    S.imu_between(t_ref, f.t, &i0, &ni);
    vieo_imu_preint im;
    {
      const int32_t first[2] = {0, ni};
      double prv[81];
      int32_t st = 0;
      CHECK(vieo_imu_preintegrate_batch(&S.noise, S.imu.data() + i0, first, &t_ref, &f.t, ref_nav.bg, ref_nav.ba, 1, &im, prv, &st));
      if (st != 0) std::fprintf(stderr, "frame %d: IMU pre-integration failed (%d)\n", k, st), std::exit(1);
    }
    ms_stage[ST_PREINT] += ms_since(t0);
    const vieo_navstate nav_pred = predict(ref_nav, im);
    // ---- TrackWithIMU: SearchByProjection(mCurrentFrame, mLastFrame, th) + PoseOptimization
    t0 = Clock::now();
    vieo_sbp_camera cam;
    std::memset(&cam, 0, sizeof(cam));
    Tcw_of(nav_pred, S.Tbc, cam.Tcw_cur), Tcw_of(L.nav, S.Tbc, cam.Tcw_last);
    cam.fx = S.fx, cam.fy = S.fy, cam.cx = S.cx, cam.cy = S.cy;
    cam.bounds[0] = 0, cam.bounds[1] = (float)S.W, cam.bounds[2] = 0, cam.bounds[3] = (float)S.H;
    cam.bf = S.bf, cam.baseline = S.baseline, cam.th = th_last, cam.th_far = 0, cam.mono = 0, cam.nlevels = NLEVELS;
    for (int l = 0; l < NLEVELS; l++) cam.scale[l] = scale[l];
    pts.assign(L.N, vieo_last_frame_point());
    std::memset(pts.data(), 0, pts.size() * sizeof(vieo_last_frame_point));
    for (int i = 0; i < L.N; i++) {
      vieo_last_frame_point& p = pts[i];
      p.octave = L.keys[i].octave, p.angle = L.keys[i].angle;
      const long m = L.mp_ref[i];
      if (m >= 0 && !L.outlier[i] && !mp_bad[m]) {
        for (int r = 0; r < 3; r++) p.Xw[r] = mp_X[3 * m + r];
        p.flags = 3;
        std::memcpy(p.desc, &mp_desc[(size_t)m * 32], 32);
      }
    }
    a1.assign(std::max(f.N, 1), -1);
    int32_t n1 = 0;
    const bool res = resident && vieo_orb_holds(extL, f.keys.data(), f.N);
    auto search_last = [&] {
      if (res) {
        CHECK(vieo_search_by_projection_last_frame_resident(extL, pts.data(), L.N, &cam, nullptr, 0.9f, 1, a1.data(), &n1));
      } else {
        q1.resize(std::max(L.N, 1));
        CHECK(vieo_sbp_project_last_frame(pts.data(), L.N, &cam, q1.data()));
        CHECK(vieo_search_by_projection(VIEO_SBP_LAST_FRAME, q1.data(), L.N, f.keys.data(), f.uright.data(), f.desc.data(), nullptr, f.N,
                                        cam.bounds, 0.9f, 1, a1.data(), &n1));
      }
    };
    search_last();
    if (n1 < 20) {  // the wider window of Tracking.cc:301-309
      cam.th = 2 * th_last;
      widened++;
      search_last();
    }
    for (int i = 0; i < f.N; i++)
      if (a1[i] >= 0) f.mp_ref[i] = L.mp_ref[a1[i]], f.track_depth[i] = L.track_depth[a1[i]];
    ms_stage[ST_SBP_LAST] += ms_since(t0);
    t0 = Clock::now();
    build_obs(f);
    vieo_vio_frame F1;
    vio_frame(F1, nav_pred, ref_nav, im, prior, f.t - t_ref, 0, (int)obs.size());
    vieo_vio_result r1;
    CHECK(vieo_pose_optimization_vio(&F1, obs.data(), outl.data(), &r1));
    for (size_t j = 0; j < obs.size(); j++)
      if (outl[j]) f.mp_ref[idx[j]] = -1;  // Discard outliers
    ms_stage[ST_POSE1] += ms_since(t0);
    // ---- TrackLocalMapWithIMU: SearchLocalPoints + PoseOptimization(bComputeMarg)
    const vieo_navstate nav1 = r1.base.status == 0 ? r1.base.nav : nav_pred;
    t0 = Clock::now();
    double Tcw1[12];
    Tcw_of(nav1, S.Tbc, Tcw1);
    // UpdateLocalMap: the points of the last n_local_kfs key frames, not yet in the frame
    {
      std::vector<char> seen(mp_bad.size(), 0);
      for (int i = 0; i < f.N; i++)
        if (f.mp_ref[i] >= 0) seen[f.mp_ref[i]] = 1;
      cand.clear();
      for (size_t kk = kfs.size() > (size_t)n_local_kfs ? kfs.size() - n_local_kfs : 0; kk < kfs.size(); kk++)
        for (long m : kfs[kk]->mp_ref)
          if (m >= 0 && !seen[m] && !mp_bad[m]) seen[m] = 1, cand.push_back(m);
    }
    int32_t n2 = 0;
    if (!cand.empty()) {
      vieo_frustum_frame FF;
      std::memset(&FF, 0, sizeof(FF));
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) FF.Rcrw[r * 3 + c] = (float)Tcw1[r * 4 + c];
        FF.tcrw[r] = (float)Tcw1[r * 4 + 3];
        FF.Ow[r] = (float)(-(Tcw1[r] * Tcw1[3] + Tcw1[4 + r] * Tcw1[7] + Tcw1[8 + r] * Tcw1[11]));
      }
      FF.n_cams = 1, FF.use_distort = 0, FF.cams = &pinhole;
      FF.Tcr[0][0] = FF.Tcr[0][5] = FF.Tcr[0][10] = 1.f;
      FF.bounds[0][0] = 0, FF.bounds[0][1] = (float)S.W, FF.bounds[0][2] = 0, FF.bounds[0][3] = (float)S.H;
      FF.bf = S.bf, FF.n_levels = NLEVELS, FF.viewing_cos_limit = 0.5f;
      FF.log_scale_factor = logf(SCALE);
      fp.resize(cand.size()), info.resize(cand.size());
      for (size_t j = 0; j < cand.size(); j++) {
        const long m = cand[j];
        for (int r = 0; r < 3; r++) fp[j].Xw[r] = mp_X[3 * m + r], fp[j].normal[r] = mp_normal[3 * m + r];
        fp[j].max_distance = mp_maxd[m], fp[j].min_distance = mp_mind[m];
      }
      CHECK(vieo_is_in_frustum_batch(&FF, fp.data(), (int)cand.size(), info.data()));
      ms_stage[ST_FRUSTUM] += ms_since(t0);
      t0 = Clock::now();
      // the head of SearchByProjection(Frame&, vector<MapPoint*>&, th) (ORBmatcher.cc:237-266): one window query per
      // (point in view, camera), point-major
      q2.clear(), owner.clear();
      for (size_t j = 0; j < cand.size(); j++) {
        const vieo_track_info& ti = info[j];
        const int cnt = std::min(std::max(ti.n, 0), 4);
        for (int sl = 0; sl < cnt; sl++) {
          vieo_proj_query q;
          std::memset(&q, 0, sizeof(q));
          const int lvl = ti.level[sl];
          float r = ti.viewcos[sl] > 0.998f ? 2.5f : 4.0f;
          if (th_local != 1.0f) r = r * th_local;
          q.u = ti.u[sl], q.v = ti.v[sl], q.ur = ti.ur[sl];
          q.radius = r * scale[lvl];
          q.level_min = lvl - 1, q.level_max = lvl, q.angle = 0;
          q.flags = 1 | 2 | (ti.cam[sl] << 8);
          std::memcpy(q.desc, &mp_desc[(size_t)cand[j] * 32], 32);
          q2.push_back(q), owner.push_back((int32_t)j);
        }
      }
      if (!q2.empty()) {
        taken.assign(f.N, 0);
        for (int i = 0; i < f.N; i++) taken[i] = f.mp_ref[i] >= 0;
        a2.assign(std::max(f.N, 1), -1);
        if (res)
          CHECK(vieo_search_by_projection_resident(VIEO_SBP_LOCAL_MAP, extL, q2.data(), (int)q2.size(), nullptr, taken.data(), cam.bounds,
                                                   0.8f, 1, a2.data(), &n2));
        else
          CHECK(vieo_search_by_projection(VIEO_SBP_LOCAL_MAP, q2.data(), (int)q2.size(), f.keys.data(), f.uright.data(), f.desc.data(),
                                          taken.data(), f.N, cam.bounds, 0.8f, 1, a2.data(), &n2));
        for (int i = 0; i < f.N; i++)
          if (a2[i] >= 0) f.mp_ref[i] = cand[owner[a2[i]]], f.track_depth[i] = info[owner[a2[i]]].track_depth;
      }
      ms_stage[ST_SBP_LOCAL] += ms_since(t0);
    } else
      ms_stage[ST_FRUSTUM] += ms_since(t0);
    t0 = Clock::now();
    build_obs(f);
    vieo_vio_frame F2;
    vio_frame(F2, nav1, ref_nav, im, prior, f.t - t_ref, 1, (int)obs.size());
    vieo_vio_result r2;
    CHECK(vieo_pose_optimization_vio(&F2, obs.data(), outl.data(), &r2));
    std::fill(f.outlier.begin(), f.outlier.end(), 0);
    for (size_t j = 0; j < obs.size(); j++)
      if (outl[j]) f.outlier[idx[j]] = 1;
    f.nav = r2.base.status == 0 ? r2.base.nav : nav1;
    f.has_prior = r2.has_marg != 0;
    if (f.has_prior) f.prior_nav = f.nav, std::memcpy(f.H_prior, r2.H_marg, sizeof(f.H_prior));
    ms_stage[ST_POSE2] += ms_since(t0);
    map_updated = false;
    n_tracked++;
    ms_frames += ms_since(t_frame);
    finish_frame(k, fp_);
  }
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s seq.vseq [traj.bin] [--frames N] [--warmup M] [--lba-lag L] [--resident 0|1] [--quiet]\n", argv[0]);
    return 2;
  }
  const char* traj_path = nullptr;
  int n_frames = -1, warmup = 0, lba_lag = 0, resident = 1;
  bool quiet = false;
  for (int i = 2; i < argc; i++) {
    if (!std::strcmp(argv[i], "--frames") && i + 1 < argc)
      n_frames = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--warmup") && i + 1 < argc)
      warmup = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--lba-lag") && i + 1 < argc)
      lba_lag = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--resident") && i + 1 < argc)
      resident = std::atoi(argv[++i]);
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
  const int n = n_frames > 0 ? std::min(n_frames, S.n_frames) : S.n_frames;
  if (warmup > 1) {
    Dropin Wm(S);
    Wm.lba_lag = lba_lag, Wm.resident = resident != 0;
    Wm.initialise();
    for (int k = 1; k < std::min(warmup, S.n_frames); k++) Wm.before_frame(k), Wm.step(k);
    Wm.before_frame(1 << 30);
  }
  Dropin R(S);
  R.lba_lag = lba_lag, R.resident = resident != 0;
  R.initialise();
  const auto t0 = Clock::now();
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
  const double ms_total = ms_since(t0);
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
  std::printf("{\"frames\": %d, \"resident\": %d, \"ms_per_frame\": %.4f, \"frames_per_s\": %.2f, \"ms_frame_without_local_ba\": %.4f, "
              "\"local_bas\": %d, \"ms_per_local_ba\": %.4f, \"key_frames\": %zu, \"map_points\": %zu, \"widened\": %d, \"lba_lag\": %d, "
              "\"ate_rmse_vs_truth_m\": %.6e, \"max_err_vs_truth_m\": %.6e, \"sequential_calls_per_frame\": %d, \"stage_ms_per_frame\": {",
              nf, resident, ms_total / nf, 1e3 * nf / ms_total, R.ms_frames / nf, R.n_lba, R.n_lba ? R.ms_lba / R.n_lba : 0.0, R.kfs.size(),
              R.mp_bad.size(), R.widened, lba_lag, std::sqrt(e2 / n), emax, 8);
  for (int s = 0; s < ST_N; s++) std::printf("%s\"%s\": %.4f", s ? ", " : "", kStageName[s], R.ms_stage[s] / nf);
  std::printf("}, %s}\n", R.run_shape_json().c_str());
  return 0;
}
