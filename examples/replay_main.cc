// replay_main.cc -- the sequential stereo-inertial replay WITHOUT Python: a C++ host that plays the part of the
// reference's Tracking / LocalMapping threads around the C-ABI of include/vieo_hot.h.
//
//   per frame     vieo_track_frame                 (Frame::Frame + TrackWithIMU + TrackLocalMapWithIMU, one call)
//   per key frame vieo_imu_preintegrate_batch      (the key-frame pair's pre-integration, LocalMapping side)
//                 vieo_local_bundle_adjustment_vio (Optimizer::LocalBundleAdjustmentNavStatePRV)
//                 vieo_update_normal_and_depth_batch (MapPoint::UpdateNormalAndDepth after the write-back)
//
// It is the C++ twin of vieo_slam_amd/replay.py (class Replay / tracker.TrackerReplay), same simplifications: a key
// frame every `kf_every` frames, new points only from stereo depth (CreateNewKeyFrame, src/Tracking.cc:2180-2250),
// local map = the points of the last `n_local_kfs` key frames, first frame initialised with the true state.  The map
// (key frames, points, observations) lives here on the host, as it lives in Map / KeyFrame / MapPoint in the reference.
//
//   tools/write_sequence.py seq.vseq --frames 100      # the rendered sequence, once
//   ./examples/replay_main seq.vseq [traj.bin] [--frames N] [--warmup M] [--quiet]
// --warmup M: M frames are replayed once before the timed run (code objects loaded, scratch buffers allocated).
// Prints one JSON line: frames, ms per frame (whole loop / tracking call / its GPU part), local BAs, error against the
// sequence's true trajectory; traj.bin receives the optimised vieo_navstate of every frame (176 B each).
// Built by __graft_entry__.build().
#include "replay_common.hpp"

using namespace vieo_replay;

namespace {

struct Replay : ReplayBase {
  vieo_tracker* trk = nullptr;
  // frame pipelining (vieo_track_input.next_left / next_right): a dataset player has frame k + 1's images in hand while
  // frame k is tracked; their extraction + stereo stage then run beside frame k's searches and optimisations
  bool prefetch = false, prefetched = false;
  int last_frame = -1;
  double ms_prep = 0, ms_post = 0, ms_finish = 0;  // the caller's own work per frame: before / after the call, bookkeeping
  double next_bias[6];

  explicit Replay(const Sequence& s) : ReplayBase(s) {
    vieo_tracker_params P;
    std::memset(&P, 0, sizeof(P));
    P.width = S.W, P.height = S.H, P.n_features = NFEAT, P.n_levels = NLEVELS, P.ini_th_fast = INI_TH, P.min_th_fast = MIN_TH;
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
    CHECK(vieo_tracker_create(&trk, &P));
  }
  ~Replay() override { vieo_tracker_destroy(trk); }

  // one frame: Tracking::Track for the stereo-inertial steady state
  void step(int k) {
    const auto t0 = std::chrono::steady_clock::now();
    Frame& L = *last;
    const Frame& ref = map_updated ? *kfs.back() : L;
    const double t = S.time(k);
    vieo_track_input in;
    std::memset(&in, 0, sizeof(in));
    in.left = S.image(k, 0), in.right = S.image(k, 1), in.stride = S.W;
    if (prefetch) {
      in.use_prefetched = prefetched ? 1 : 0;
      prefetched = k + 1 <= last_frame;
      if (prefetched) {
        in.next_left = S.image(k + 1, 0), in.next_right = S.image(k + 1, 1);
        // ... and its pre-integration, which starts at this frame -- unless LocalMapping's write-back is applied in between:
        // then the next prediction starts at the newest key frame, with the bias the local BA gave it.  When that solve has
        // finished by now its result is known and the run-ahead integration is told its reference (next_ref_bias); when it
        // has not, the next call integrates on its own (either way its outputs are the same bits).
        int j0, nj;
        if (job && k + 1 >= lba_due) {
          bool done;
          {
            std::lock_guard<std::mutex> g(lba_m);
            done = !lba_busy;
          }
          if (done && job->rc == 0 && job->res.status == 0) {
            const Frame& kf = *kfs.back();
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
    int i0, ni;
    S.imu_between(ref.t, t, &i0, &ni);
    in.imu = S.imu.data() + i0, in.n_imu = ni;
    in.t_ref = ref.t, in.t_cur = t;
    in.nav_ref = ref.nav, in.nav_last = L.nav;
    if (!map_updated && L.has_prior) in.nav_prior = &L.prior_nav, in.H_prior = L.H_prior;
    std::vector<vieo_last_frame_point> pts(L.N);
    std::memset(pts.data(), 0, pts.size() * sizeof(vieo_last_frame_point));
    std::vector<int32_t> where(mp_bad.size(), -1);
    for (int i = 0; i < L.N; i++) {
      vieo_last_frame_point& p = pts[i];
      p.octave = L.keys[i].octave, p.angle = L.keys[i].angle;
      const long m = L.mp_ref[i];
      const bool has = m >= 0 && !L.outlier[i] && !mp_bad[m];
      if (has) {
        for (int r = 0; r < 3; r++) p.Xw[r] = mp_X[3 * m + r];
        p.flags = 3;
        std::memcpy(p.desc, &mp_desc[(size_t)m * 32], 32);
        if (where[m] < 0) where[m] = i;  // the first key that holds the point
      }
    }
    local_points();
    std::vector<int32_t> alias(lp.size());
    for (size_t j = 0; j < lp.size(); j++) alias[j] = where[lp[j]];
    in.n_last = L.N, in.last_points = pts.data(), in.last_track_depth = L.track_depth.data();
    in.n_local = (int)lp.size(), in.local_version = local_version;
    in.local_points = lp_pts.data(), in.local_desc = lp_desc.data(), in.local_alias = alias.data();
    vieo_track_output out;
    const auto t1 = std::chrono::steady_clock::now();
    CHECK(vieo_track_frame(trk, &in, &out));
    const auto t2 = std::chrono::steady_clock::now();
    if (out.status != VIEO_TRACK_OK) {
      std::fprintf(stderr, "frame %d: IMU pre-integration failed (%d)\n", k, out.preint_status);
      std::exit(1);
    }
    ms_track += out.ms_host, ms_gpu += out.ms_gpu, n_tracked++, widened += out.widened;
    FramePtr f = std::make_shared<Frame>();
    f->k = k, f->t = t, f->N = out.n_keys;
    const int N = out.n_keys, cap = out.key_cap;
    f->keys.assign(out.keys, out.keys + N), f->desc.assign(out.desc, out.desc + (size_t)N * 32);
    f->uright.assign(out.uright, out.uright + N), f->depth.assign(out.depth, out.depth + N);
    f->mp_ref.assign(N, -1), f->track_depth.assign(N, std::numeric_limits<float>::infinity());
    f->outlier.assign(out.outlier, out.outlier + N);
    for (int i = 0; i < N; i++) {
      const int r = out.point_ref[i];
      if (r >= 0 && r < cap)
        f->mp_ref[i] = L.mp_ref[r], f->track_depth[i] = L.track_depth[r];
      else if (r >= cap)
        f->mp_ref[i] = lp[r - cap], f->track_depth[i] = out.local_track_depth[r - cap];
    }
    f->nav = out.second.base.status == 0 ? out.second.base.nav : out.first.base.nav;
    f->has_prior = out.second.has_marg != 0;
    if (f->has_prior) f->prior_nav = f->nav, std::memcpy(f->H_prior, out.second.H_marg, sizeof(f->H_prior));
    map_updated = false;
    const auto t3 = std::chrono::steady_clock::now();
    ms_frames += std::chrono::duration<double, std::milli>(t3 - t0).count();
    finish_frame(k, f);
    ms_prep += std::chrono::duration<double, std::milli>(t1 - t0).count(), ms_post += std::chrono::duration<double, std::milli>(t3 - t2).count();
    ms_finish += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t3).count();
  }
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s seq.vseq [traj.bin] [--frames N] [--warmup M] [--lba-lag L] [--prefetch 0|1] [--trackers T] [--quiet]\n", argv[0]);
    return 2;
  }
  const char* traj_path = nullptr;
  int n_frames = -1, warmup = 0, lba_lag = 0, n_trackers = 1, kf_every = 10, prefetch = 0;
  bool quiet = false;
  for (int i = 2; i < argc; i++) {
    if (!std::strcmp(argv[i], "--frames") && i + 1 < argc)
      n_frames = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--trackers") && i + 1 < argc)
      n_trackers = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--prefetch") && i + 1 < argc)
      prefetch = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--kf-every") && i + 1 < argc)  // (experiments: a huge value = no key frames, no local BA)
      kf_every = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--warmup") && i + 1 < argc)
      warmup = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--lba-lag") && i + 1 < argc)
      lba_lag = std::atoi(argv[++i]);
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
    Replay Wm(S);
    Wm.lba_lag = lba_lag, Wm.prefetch = prefetch != 0, Wm.last_frame = std::min(warmup, S.n_frames) - 1;
    Wm.initialise();
    for (int k = 1; k < std::min(warmup, S.n_frames); k++) Wm.before_frame(k), Wm.step(k);
    Wm.before_frame(1 << 30);
  }
  if (n_trackers > 1) {
    // N independent sequences on ONE GPU (BASELINE configs[3] "8 sequences": what a GPU of that run serves when it gets
    // more than one): every tracker has its own host thread, streams, map and LocalMapping thread; all replay the same
    // file, so every trajectory must equal the first one's bit for bit.
    std::vector<std::unique_ptr<Replay>> Rs;
    for (int i = 0; i < n_trackers; i++) {
      Rs.emplace_back(new Replay(S));
      Rs.back()->lba_lag = lba_lag, Rs.back()->kf_every = kf_every, Rs.back()->prefetch = prefetch != 0, Rs.back()->last_frame = n - 1;
      Rs.back()->initialise();
    }
    std::mutex m;
    std::condition_variable cv;
    int ready = 0;
    bool go = false;
    std::vector<double> wall(n_trackers, 0.0);
    std::vector<std::thread> th;
    for (int i = 0; i < n_trackers; i++)
      th.emplace_back([&, i] {
        {
          std::unique_lock<std::mutex> g(m);
          ready++;
          cv.notify_all();
          cv.wait(g, [&] { return go; });
        }
        const auto t0 = std::chrono::steady_clock::now();
        Replay& R = *Rs[i];
        for (int k = 1; k < n; k++) {
          const auto tk = std::chrono::steady_clock::now();
          R.before_frame(k);
          R.step(k);
          R.frame_ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tk).count());
        }
        wall[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        R.before_frame(1 << 30);
      });
    std::chrono::steady_clock::time_point t0;
    {
      std::unique_lock<std::mutex> g(m);
      cv.wait(g, [&] { return ready == n_trackers; });
      go = true;
      t0 = std::chrono::steady_clock::now();
    }
    cv.notify_all();
    for (auto& t : th) t.join();
    const double ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    bool same = true;
    for (int i = 1; i < n_trackers; i++)
      same = same && Rs[i]->traj.size() == Rs[0]->traj.size() &&
             !std::memcmp(Rs[i]->traj.data(), Rs[0]->traj.data(), Rs[0]->traj.size() * sizeof(vieo_navstate));
    std::vector<double> all;
    double gpu = 0, call = 0;
    for (auto& R : Rs) all.insert(all.end(), R->frame_ms.begin(), R->frame_ms.end()), gpu += R->ms_gpu, call += R->ms_track;
    std::sort(all.begin(), all.end());
    const double nf = (double)(n - 1);
    double mean = 0;
    for (double v : all) mean += v;
    std::printf("{\"trackers\": %d, \"frames_per_tracker\": %d, \"frames_per_s_all_trackers\": %.2f, \"ms_per_frame_latency_mean\": %.4f, "
                "\"ms_per_frame_latency_median\": %.4f, \"ms_per_frame_latency_p99\": %.4f, \"ms_track_call_mean\": %.4f, "
                "\"ms_track_gpu_mean\": %.4f, \"slowest_tracker_ms_per_frame\": %.4f, \"identical_trajectories\": %s, \"lba_lag\": %d}\n",
                n_trackers, n - 1, 1e3 * n_trackers * nf / ms_total, mean / all.size(), all[all.size() / 2],
                all[std::min(all.size() - 1, (size_t)(0.99 * all.size()))], call / (n_trackers * nf), gpu / (n_trackers * nf),
                *std::max_element(wall.begin(), wall.end()) / nf, same ? "true" : "false", lba_lag);
    if (traj_path) {
      FILE* f = std::fopen(traj_path, "wb");
      if (f) std::fwrite(Rs[0]->traj.data(), sizeof(vieo_navstate), Rs[0]->traj.size(), f), std::fclose(f);
    }
    return same ? 0 : 1;
  }
  Replay R(S);
  R.lba_lag = lba_lag, R.kf_every = kf_every, R.prefetch = prefetch != 0, R.last_frame = n - 1;
  R.initialise();
  const auto t0 = std::chrono::steady_clock::now();
  double ms_before = 0;
  for (int k = 1; k < n; k++) {
    const auto tk = std::chrono::steady_clock::now();
    R.before_frame(k);
    ms_before += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tk).count();
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
  R.before_frame(1 << 30);  // (a solve still running is waited for outside the timed frames: its frames are not in the run)
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
  std::printf("{\"frames\": %d, \"ms_per_frame\": %.4f, \"frames_per_s\": %.2f, \"ms_track_call\": %.4f, \"ms_track_gpu\": %.4f, "
              "\"ms_frame_without_local_ba\": %.4f, \"local_bas\": %d, \"ms_per_local_ba\": %.4f, \"ms_per_key_frame_preintegration\": %.4f, \"ms_per_local_mapping_job\": %.4f, \"key_frames\": %zu, "
              "\"map_points\": %zu, \"widened\": %d, \"lba_lag\": %d, \"prefetch\": %d, \"ate_rmse_vs_truth_m\": %.6e, \"max_err_vs_truth_m\": %.6e, "
              "\"caller_ms_per_frame\": {\"map_write_back\": %.4f, \"of_which_waiting_for_local_mapping\": %.4f, \"before_call\": %.4f, \"after_call\": %.4f, \"key_frame_bookkeeping\": %.4f}, %s}\n",
              nf, ms_total / nf, 1e3 * nf / ms_total, R.ms_track / nf, R.ms_gpu / nf, R.ms_frames / nf, R.n_lba,
              R.n_lba ? R.ms_lba / R.n_lba : 0.0, R.n_lba ? R.ms_kf_preint / R.n_lba : 0.0, R.n_lba ? R.ms_lba_job / R.n_lba : 0.0, R.kfs.size(), R.mp_bad.size(), R.widened, lba_lag, prefetch, std::sqrt(e2 / n), emax,
              ms_before / nf, R.ms_wait_lm / nf, R.ms_prep / nf, R.ms_post / nf, R.ms_finish / nf, R.run_shape_json().c_str());
  return 0;
}
