// tracker.hip -- one stereo-inertial frame's tracking as ONE call behind the C-ABI (vieo_track_frame).
//
// What Tracking::Track does per frame in the steady state (reference: src/Tracking.cc:261-378 TrackWithIMU,
// :385-451 PredictNavStateByIMU, :453-488 TrackLocalMapWithIMU, :2308-2370 SearchLocalPoints; Frame::Frame
// src/Frame.cc:259-320, ComputeStereoMatches :451-611) as a chain of launches on the extractor's stream:
//
//   H2D (one block: header, IMU samples, both images, last frame's points)        [+ local-map block when it changed]
//   stream B:  k_imu_preint (one wavefront; runs beside the extraction)  ----event---+
//   stream A:  extract x 2 -> stereo                                                 v
//              k_track_predict   PredictNavStateByIMU: nav_pred, both optimiser problems, Tcw of the search
//              sbp_project -> search(last frame) -> merge -> build_obs -> PoseOptimization
//              after_pose -> mark_held -> local queries (isInFrustum) -> search(local map) -> merge -> build_obs
//              PoseOptimization(bComputeMarg) -> k_track_finish (per-key outlier flags)
//   D2H (three pieces), ONE host synchronisation.
//
// The order-free bookkeeping between the stages is the vieo_track_* glue of track_glue.hip; nothing here computes
// on the host beyond filling the upload block.  The rare wider-window branch (fewer than 20 matches in the first
// search, Tracking.cc:301-309) re-runs the chain from the projection with 2 x th.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <vector>

#include "imu_device.h"
#include "orb_internal.h"

namespace vieo {

// upload header: everything small the chain reads, one struct so that it travels with the images in one copy
struct TrkHdr {
  vieo_sbp_camera cam;       // Tcw_cur / Tcw_last are written by k_track_predict / k_track_set_pose
  vieo_vio_frame f1, f2;     // base.nav / imu written by k_track_predict, n_obs / obs_begin by k_track_build_obs
                             // (vision-only trackers use f1.base / f2.base as the vieo_pose_frames)
  vieo_navstate nav_ref, nav_last;
  vieo_imu_noise noise;
  double ti, tj, bg[3], ba[3];
  int32_t first[2];
  int32_t npts[4];           // [0] = n_last, [1] = n_last * n_cams (queries of the first search)
  float consts[32];          // inv_sigma2[16], scale[16]
};

// The next frame's pre-integration, run ahead (vieo_track_input.next_imu): what goes up for it.  In HBM the block is
// [SpecImu | samples | bias (6 doubles, written by k_track_predict) | vieo_imu_preint | Sigma_prv (81) | status].
struct SpecImu {
  vieo_imu_noise noise;
  double ti, tj;
  int32_t first[2];
  double xbias[6];  // next_ref_bias: the reference's bias when it is not this frame's predicted one
};

// download header
struct TrkOut {
  int32_t cnt[8];            // extractor counts per image: {n, mono}
  int32_t nm[4];             // [0] matches of the first search, [1] of the second
  int32_t nq[4];
  int32_t preint_status[4];
  vieo_vio_result r1, r2;
  vieo_navstate nav_pred;
  vieo_imu_preint imu;
  double sigma_prv[81];
  int32_t nobs2[4];          // observations of the second optimisation
  int32_t fe_hdr[8];         // rig: header of the stereo stage (groups, matches, threshold, status, ...)
  int32_t cam_first[8];      // rig: first key of every camera in mvKeys, [n_cams] = N
  int32_t fcnt[2];           // rig: {N, 0}
};

// The last frame's part of the tail's two point tables ([last frame's points | local candidates]: positions, track depths)
// comes up with the header; blocks 1.. of the prediction kernels move it in place (two copies and an event less to hand
// to the second stream per frame).
constexpr int kTableBlocks = 16;
struct TrkTables {
  const float *xyz_in, *dep_in;
  float *xyz_out, *dep_out;
};
__device__ __forceinline__ void track_fill_tables(const TrkTables& T, int n_last) {
  const int i0 = (blockIdx.x - 1) * 64 + threadIdx.x, step = kTableBlocks * 64;
  for (int i = i0; i < 3 * n_last; i += step) T.xyz_out[i] = T.xyz_in[i];
  for (int i = i0; i < n_last; i += step) T.dep_out[i] = T.dep_in[i];
}

// PredictNavStateByIMU (Tracking.cc:385-451) from the pre-integration in HBM; fills the two optimiser problems and
// the projection search's camera.  One wavefront; lane 0 does the (double) arithmetic, all lanes copy.
__global__ void __launch_bounds__(64)
k_track_predict(TrkHdr* __restrict__ H, TrkOut* __restrict__ O, const vieo_imu_preint* __restrict__ pre,
                const double* __restrict__ sigma_prv, const int32_t* __restrict__ status, double* __restrict__ next_bias,
                TrkTables tables) {
  if (blockIdx.x > 0) return track_fill_tables(tables, H->npts[0]);
  __shared__ vieo_navstate s_nav;
  const int lane = threadIdx.x;
  const vieo_imu_preint& M = *pre;
  if (lane == 0) {
    vieo_navstate ns = H->nav_ref;
    const double dt = M.dt;
    if (dt != 0) {
      const Qd q{ns.q[0], ns.q[1], ns.q[2], ns.q[3]};
      double Rwb[9], t0[3], t1[3], t2[3], r[3];
      q_to_R(q, Rwb);
      // p += v dt + g dt^2 / 2 + Rwb (pij + Jgp dbg + Jap dba)
      mv3(M.Jgp, ns.dbg, t0), mv3(M.Jap, ns.dba, t1);
      for (int k = 0; k < 3; k++) t2[k] = M.pij[k] + t0[k] + t1[k];
      mv3(Rwb, t2, r);
      double pn[3], vn[3];
      for (int k = 0; k < 3; k++) pn[k] = ns.p[k] + (ns.v[k] * dt + H->f1.gw[k] * (dt * dt / 2) + r[k]);
      mv3(M.Jgv, ns.dbg, t0), mv3(M.Jav, ns.dba, t1);
      for (int k = 0; k < 3; k++) t2[k] = M.vij[k] + t0[k] + t1[k];
      mv3(Rwb, t2, r);
      for (int k = 0; k < 3; k++) vn[k] = ns.v[k] + (H->f1.gw[k] * dt + r[k]);
      // Rwb *= Rij Exp(JgR dbg)
      double w[3], E[9], A[9], Rn[9];
      mv3(M.JgR, ns.dbg, w);
      q_to_R(so3_exp_q(w), E);
      mm3(M.Rij, E, A);
      mm3(Rwb, A, Rn);
      const Qd qn = R_to_q(Rn);
      for (int k = 0; k < 3; k++) ns.p[k] = pn[k], ns.v[k] = vn[k];
      ns.q[0] = qn.w, ns.q[1] = qn.x, ns.q[2] = qn.y, ns.q[3] = qn.z;
    }
    for (int k = 0; k < 3; k++) {  // bj_bar = bi_bar + dbi, also when the pre-integration failed (Tracking.cc:413-419)
      ns.bg[k] += ns.dbg[k], ns.ba[k] += ns.dba[k];
      ns.dbg[k] = 0, ns.dba[k] = 0;
    }
    s_nav = ns;
    // (bj_bar: the bias the NEXT frame's pre-integration runs with when this frame is its reference -- SpecImu below)
    for (int k = 0; k < 3; k++) next_bias[k] = ns.bg[k], next_bias[3 + k] = ns.ba[k];
    // Tcw = Tcb Twb^-1 of the predicted and of the last frame's state (UpdatePoseFromNS)
    for (int which = 0; which < 2; which++) {
      const vieo_navstate& n = which == 0 ? ns : H->nav_last;
      const Qd q{n.q[0], n.q[1], n.q[2], n.q[3]};
      double Rwb[9];
      q_to_R(q, Rwb);
      const double* Rcb = H->f1.base.Rcb;
      double* T = which == 0 ? H->cam.Tcw_cur : H->cam.Tcw_last;
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++)
          T[r * 4 + c] = Rcb[r * 3] * Rwb[c * 3] + Rcb[r * 3 + 1] * Rwb[c * 3 + 1] + Rcb[r * 3 + 2] * Rwb[c * 3 + 2];
        T[r * 4 + 3] = H->f1.base.tcb[r] - (T[r * 4] * n.p[0] + T[r * 4 + 1] * n.p[1] + T[r * 4 + 2] * n.p[2]);
      }
    }
    O->preint_status[0] = status[0];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __syncthreads();
  const double* sn = (const double*)&s_nav;
  for (int i = lane; i < (int)(sizeof(vieo_navstate) / 8); i += 64) {
    ((double*)&H->f1.base.nav)[i] = sn[i], ((double*)&H->f2.base.nav)[i] = sn[i];
    ((double*)&O->nav_pred)[i] = sn[i];
  }
  const double* sm = (const double*)pre;
  for (int i = lane; i < (int)(sizeof(vieo_imu_preint) / 8); i += 64) {
    ((double*)&H->f1.imu)[i] = sm[i], ((double*)&H->f2.imu)[i] = sm[i];
    ((double*)&O->imu)[i] = sm[i];
  }
  for (int i = lane; i < 81; i += 64) O->sigma_prv[i] = sigma_prv[i];
}

// The vision-only tracker's prediction comes from the host (mVelocity * mLastFrame.Tcw, Tracking.cc:1852): the two
// optimiser problems start from it, the projection search gets Tcw of it and of the last frame.
__global__ void __launch_bounds__(64)
k_track_set_pose(TrkHdr* __restrict__ H, TrkOut* __restrict__ O, TrkTables tables) {
  if (blockIdx.x > 0) return track_fill_tables(tables, H->npts[0]);
  const int lane = threadIdx.x;
  if (lane < 2) {
    const vieo_navstate& n = lane == 0 ? H->nav_ref : H->nav_last;
    const Qd q{n.q[0], n.q[1], n.q[2], n.q[3]};
    double Rwb[9];
    q_to_R(q, Rwb);
    const double* Rcb = H->f1.base.Rcb;
    double* T = lane == 0 ? H->cam.Tcw_cur : H->cam.Tcw_last;
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++)
        T[r * 4 + c] = Rcb[r * 3] * Rwb[c * 3] + Rcb[r * 3 + 1] * Rwb[c * 3 + 1] + Rcb[r * 3 + 2] * Rwb[c * 3 + 2];
      T[r * 4 + 3] = H->f1.base.tcb[r] - (T[r * 4] * n.p[0] + T[r * 4 + 1] * n.p[1] + T[r * 4 + 2] * n.p[2]);
    }
  }
  const double* sn = (const double*)&H->nav_ref;
  for (int i = lane; i < (int)(sizeof(vieo_navstate) / 8); i += 64) {
    ((double*)&H->f1.base.nav)[i] = sn[i], ((double*)&H->f2.base.nav)[i] = sn[i];
    ((double*)&O->nav_pred)[i] = sn[i];
  }
  if (lane == 0) O->preint_status[0] = 0;
}

// per-key outlier flags of the second optimisation (mvbOutlier), and its observation count for the host
__global__ void __launch_bounds__(256)
k_track_finish(const int32_t* __restrict__ obs_key, const uint8_t* __restrict__ outl, const vieo_vio_frame* __restrict__ f2,
               uint8_t* __restrict__ key_outlier, int key_cap, TrkOut* __restrict__ O) {
  const int n = f2->base.n_obs;
  for (int i = threadIdx.x; i < key_cap; i += 256) key_outlier[i] = 0;
  __syncthreads();
  for (int j = threadIdx.x; j < n; j += 256)
    if (outl[j]) key_outlier[obs_key[j]] = 1;
  if (threadIdx.x == 0) O->nobs2[0] = n;
}

// A prefetched frame becomes the current one: the slot the third stream extracted into -> the arrays the chain reads
// (both images' keys and descriptors, their counts, uright / depth of the stereo stage).  One launch instead of five copies.
__global__ void __launch_bounds__(256)
k_track_adopt(const uint4* __restrict__ s_kp, uint4* __restrict__ d_kp, int n_kp16, const uint4* __restrict__ s_desc,
              uint4* __restrict__ d_desc, int n_desc16, const uint4* __restrict__ s_ur, uint4* __restrict__ d_ur,
              const uint4* __restrict__ s_dp, uint4* __restrict__ d_dp, int n_f16, const int32_t* __restrict__ s_cnt,
              int32_t* __restrict__ d_cnt, int n_cnt) {
  const int stride = gridDim.x * 256;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_kp16; i += stride) d_kp[i] = s_kp[i];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_desc16; i += stride) d_desc[i] = s_desc[i];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_f16; i += stride) d_ur[i] = s_ur[i], d_dp[i] = s_dp[i];
  if (blockIdx.x == 0 && threadIdx.x < n_cnt) d_cnt[threadIdx.x] = s_cnt[threadIdx.x];
}

}  // namespace vieo

using namespace vieo;

struct vieo_tracker {
  vieo_tracker_params P;
  vieo_tracker_rig R;
  bool rig = false, vision = false;
  int n_img = 2;     // images per frame
  int nc = 1;        // cameras the searches loop over (1: the rectified pair's left camera)
  int kc = 0;        // key capacity of a frame: cap, or n_cams * cap of a rig (mvKeys)
  vieo_orb* ext = nullptr;
  vieo_fisheye* fe = nullptr;
  hipStream_t st = nullptr, st_imu = nullptr, st_pref = nullptr;
  hipEvent_t ev_up = nullptr, ev_imu = nullptr, ev_t0 = nullptr, ev_t1 = nullptr, ev_ext = nullptr, ev_fe = nullptr, ev_kd = nullptr;
  hipEvent_t ev_head = nullptr, ev_pref = nullptr;  // this frame's stereo stage is done / the next frame is extracted
  hipEvent_t ev_spec = nullptr;                     // the pre-integration run ahead (second stream) is done
  hipEvent_t ev_h2d = nullptr;                      // the prefetched images have left the pinned planes
  bool h2d_pending = false;
  hipEvent_t ev_tab = nullptr;                      // a changed local map (second stream) is in place
  bool tab_pending = false;
  // frame pipelining (vieo_track_input.next_left / next_right): the next frame's images and what its extraction and
  // stereo stage produce, on the third stream
  uint8_t *h_next = nullptr, *d_next = nullptr, *d_slot = nullptr;
  size_t s_kp = 0, s_desc = 0, s_ur = 0, s_dp = 0, s_cnt = 0;
  bool pref_valid = false;
  int pref_frames = 0;
  // ... and its pre-integration (next_imu): pinned / device blocks, what was integrated (the next call compares), counts
  uint8_t *h_spec = nullptr, *d_spec = nullptr;
  size_t sp_samples = 0, sp_bias = 0, sp_pre = 0, sp_prv = 0, sp_pst = 0, sp_up = 0;
  std::vector<vieo_imu_sample> spec_samples;
  double spec_ti = 0, spec_tj = 0, spec_bias[6] = {0, 0, 0, 0, 0, 0};
  int spec_n = 0;
  bool spec_valid = false;
  int spec_used = 0;
  int cap = 0, ccap = 0, pcap = 0, gcap = 0, imu_cap = 512;
  int local_version = -1, n_local_dev = 0;
  int replica_repeats = 0;    // frames whose optimisations were repeated on one workgroup (a replica did not arrive)
  float side_ratio = 0.f;     // create_side_stream: elapsed(both spin kernels) / elapsed(one): 1 side by side, 2 in series
  int side_probes = 1, side_checks = 0;
  // the frames' GPU times: a ring for the running median, and how many frames in a row sat 30 % above it -- the sign that
  // the two streams have come to share a hardware queue (streams the process created since: the mapping is the runtime's)
  float gpu_ring[32] = {};
  int gpu_n = 0, slow_run = 0, frames_since_check = 0;
  float scale[16], inv_sigma2[16];
  vieo_camera pin_cam;
  vieo_frustum_frame ff;
  float bounds[4][4];
  // pinned blocks and their device twins (same layout)
  uint8_t *h_up = nullptr, *d_up = nullptr;      // per-frame upload
  uint8_t *h_loc = nullptr, *d_loc = nullptr;    // local-map candidates (uploaded when they change)
  uint8_t *h_out = nullptr, *d_out = nullptr;    // download
  uint8_t* d_work = nullptr;                     // device-only scratch
  uint8_t* d_const = nullptr;                    // rig: vieo_sbp_rig | vieo_camera[4]
  // offsets in the upload block
  size_t o_hdr, o_imu, o_img, o_pts, o_xyz, o_dep, o_alias, up_fixed, up_small;
  // offsets in the local block
  size_t l_cpt, l_cdesc, l_xyz;
  // offsets in the download block
  size_t q_hdr, q_ur, q_dp, q_mpref, q_outl, q_kg, q_gidx, q_good, q_p3d, q_small_end, q_kp, q_desc, q_cdep, out_bytes;
  // offsets in the work block
  size_t w_kp, w_desc, w_kcat, w_dcat, w_q1, w_q1c, w_qsrc, w_q2, w_assign, w_taken, w_held, w_obs, w_obskey, w_outl, w_xyz, w_dep, w_pre, w_prv, w_pst;
};

static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

// The second stream of a tracker (pre-integration, a rig frame's stereo bookkeeping, the copies back) must be served by
// another hardware queue than the first: on one queue their kernels run one after the other (observed with two streams
// of equal priority: the "parallel" pre-integration of every frame sat in front of its extraction, 100 us; in a process
// that holds many streams the assignment is anybody's guess: a 2-camera frame 1.06 -> 1.55 ms with eight idle torch
// streams around).  HIP does not say which queue a stream gets, so candidates are TRIED: a ~40 us spin kernel on each of
// the two streams, started together -- a candidate whose kernel ends when the first one's does runs beside it.  High
// priority first (its own pool of queues where the runtime has one), a handful of attempts, the best one is kept.
__global__ void k_track_spin(long long cycles) {
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
}
// elapsed(both spin kernels, started together on the two streams) / elapsed(one): ~1 side by side, ~2 one after the other
static bool side_stream_ratio(hipStream_t main_stream, hipStream_t c, hipEvent_t e0, hipEvent_t e1, hipEvent_t e2, float* ratio) {
  const long long spin = 100000;  // ~40 us
  float both = 0, one = 0;
  bool ok = true;
  for (int rep = 0; rep < 2 && ok; rep++) {  // (the first round also creates the candidate's queue)
    ok = hipEventRecord(e0, main_stream) == hipSuccess && hipStreamWaitEvent(c, e0, 0) == hipSuccess;
    hipLaunchKernelGGL(k_track_spin, dim3(1), dim3(64), 0, main_stream, spin);
    hipLaunchKernelGGL(k_track_spin, dim3(1), dim3(64), 0, c, spin);
    ok = ok && hipEventRecord(e1, main_stream) == hipSuccess && hipEventRecord(e2, c) == hipSuccess &&
         hipStreamSynchronize(main_stream) == hipSuccess && hipStreamSynchronize(c) == hipSuccess &&
         hipEventElapsedTime(&one, e0, e1) == hipSuccess && hipEventElapsedTime(&both, e0, e2) == hipSuccess;
  }
  if (ok) *ratio = both / (one > 0 ? one : 1.f);
  return ok;
}
static const float kSideRatioOk = 1.35f;
// *ratio: what the kept stream measured (reported by vieo_tracker_get_stats; > kSideRatioOk = no candidate ran beside
// the main stream, the best of them is kept and the frame's second stream is then in series with the first)
static hipError_t create_side_stream(hipStream_t* out, hipStream_t main_stream, float* ratio_out) {
  int lo = 0, hi = 0;
  const bool prio = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi;
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
  *ratio_out = 0.f;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreate(&e2) != hipSuccess) {
    for (hipEvent_t e : {e0, e1, e2})
      if (e) (void)hipEventDestroy(e);
    return prio ? hipStreamCreateWithPriority(out, hipStreamNonBlocking, hi) : hipStreamCreateWithFlags(out, hipStreamNonBlocking);
  }
  hipStream_t best = nullptr, held[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  float best_ratio = 1e9f;
  for (int k = 0; k < 6; k++) {
    hipStream_t c = nullptr;
    // (normal-priority candidates first: the main stream is a high-priority one unless VIEO_TRACKER_PRIORITY=0)
    const hipError_t err = (prio && k % 2 == 1) ? hipStreamCreateWithPriority(&c, hipStreamNonBlocking, hi)
                                                : hipStreamCreateWithFlags(&c, hipStreamNonBlocking);
    if (err != hipSuccess) break;
    held[k] = c;  // (kept until the end: a destroyed candidate's queue would be handed to the next one)
    float ratio = 0;
    if (!side_stream_ratio(main_stream, c, e0, e1, e2, &ratio)) continue;
    if (ratio < best_ratio) best_ratio = ratio, best = c;
    if (ratio < kSideRatioOk) break;
  }
  for (hipStream_t c : held)
    if (c && c != best) (void)hipStreamDestroy(c);
  for (hipEvent_t e : {e0, e1, e2}) (void)hipEventDestroy(e);
  if (!best) return hipErrorUnknown;
  *out = best, *ratio_out = best_ratio;
  return hipSuccess;
}

// The third stream (the next frame's extraction).  VIEO_PREFETCH_PRIORITY = -1 (lowest: the bundle adjustment's pool of
// hardware queues), 0 (normal, the default), 1 (highest): A/B runs.
static hipError_t create_prefetch_stream(hipStream_t* out) {
  const char* e = getenv("VIEO_PREFETCH_PRIORITY");
  const int want = e ? atoi(e) : 0;
  int lo = 0, hi = 0;
  if (want != 0 && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
    return hipStreamCreateWithPriority(out, hipStreamNonBlocking, want < 0 ? lo : hi);
  return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

extern "C" {

void vieo_tracker_destroy(vieo_tracker* t) {
  if (!t) return;
  if (t->st) (void)hipStreamSynchronize(t->st);
  if (t->st_imu) (void)hipStreamSynchronize(t->st_imu), (void)hipStreamDestroy(t->st_imu);
  if (t->st_pref) (void)hipStreamSynchronize(t->st_pref), (void)hipStreamDestroy(t->st_pref);
  for (hipEvent_t e : {t->ev_up, t->ev_imu, t->ev_t0, t->ev_t1, t->ev_ext, t->ev_fe, t->ev_kd, t->ev_head, t->ev_pref, t->ev_tab, t->ev_h2d, t->ev_spec})
    if (e) (void)hipEventDestroy(e);
  for (uint8_t* p : {t->h_up, t->h_loc, t->h_out, t->h_next, t->h_spec})
    if (p) (void)hipHostFree(p);
  for (uint8_t* p : {t->d_up, t->d_loc, t->d_out, t->d_work, t->d_const, t->d_next, t->d_slot, t->d_spec})
    if (p) (void)hipFree(p);
  if (t->fe) vieo_fisheye_destroy(t->fe);
  if (t->ext) vieo_orb_destroy(t->ext);
  delete t;
}

int vieo_tracker_create_rig(vieo_tracker** out, const vieo_tracker_params* P, const vieo_tracker_rig* R) {
  if (!out || !P || P->width <= 0 || P->height <= 0 || P->n_levels < 1 || P->n_levels > 16 || P->max_local_points < 0)
    return VIEO_E_INVALID;
  if (R && (R->n_cams < 2 || R->n_cams > 4 || P->vision_only)) {
    set_error("vieo_tracker_create_rig: %d cameras (2..4), visual-inertial", R->n_cams);
    return VIEO_E_INVALID;
  }
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  vieo_tracker* t = new vieo_tracker();
  t->P = *P;
  t->rig = R != nullptr, t->vision = P->vision_only != 0;
  if (R) t->R = *R;
  t->n_img = R ? R->n_cams : 2, t->nc = R ? R->n_cams : 1;
  // the main stream from the high-priority queues, the second one from the normal ones, the bundle adjustment's at the
  // lowest level: three queue pools, no sharing (orb_create_with_priority).  VIEO_TRACKER_PRIORITY=0: all normal (A/B).
  static const int main_prio = [] {
    const char* e = getenv("VIEO_TRACKER_PRIORITY");
    return e ? atoi(e) : 1;
  }();
  if ((rc = vieo::orb_create_with_priority(&t->ext, P->n_features, P->scale_factor, P->n_levels, P->ini_th_fast, P->min_th_fast,
                                           main_prio)) != VIEO_OK) {
    delete t;
    return rc;
  }
  t->st = (hipStream_t)vieo_orb_stream(t->ext);
  t->cap = vieo_orb_max_keypoints(t->ext);
  t->kc = t->nc * t->cap;
  t->ccap = std::max(P->max_local_points, 64);
  t->pcap = t->kc + t->ccap;
  vieo_orb_scale_factors(t->ext, t->scale);
  vieo_orb_inv_level_sigma2(t->ext, t->inv_sigma2);
  if (R) {
    float sig2[16];
    vieo_orb_level_sigma2(t->ext, sig2);
    vieo_fisheye_params fp;
    memset(&fp, 0, sizeof(fp));
    fp.n_cams = R->n_cams, fp.n_levels = P->n_levels, fp.bf = P->bf, fp.th_far_pts = R->th_far_pts;
    fp.cams = R->cams, fp.Trc = &R->Trc[0][0], fp.Tcr = &R->Tcr[0][0], fp.level_sigma2 = sig2;
    if ((rc = vieo_fisheye_create(&t->fe, &fp, t->cap, 1)) != VIEO_OK) {
      vieo_tracker_destroy(t);
      return rc;
    }
    t->gcap = vieo_fisheye_group_capacity(t->fe);
  }
  const size_t npx = (size_t)P->width * P->height;
  const int cap = t->cap, ccap = t->ccap, kc = t->kc, nc = t->nc;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t r = o;
    o = al256(o + bytes);
    return r;
  };
  // ---- upload block: [header | IMU samples | last points | their xyz | their depth | alias | images]: the images last, so
  // that a call whose frame was prefetched uploads the head only
  t->o_hdr = take(sizeof(TrkHdr)), t->o_imu = take((size_t)t->imu_cap * sizeof(vieo_imu_sample));
  t->o_pts = take((size_t)kc * sizeof(vieo_last_frame_point));
  t->o_xyz = take((size_t)kc * 12), t->o_dep = take((size_t)kc * 4), t->o_alias = take((size_t)ccap * 4);
  t->up_fixed = t->o_alias;
  t->up_small = o;
  t->o_img = take(t->n_img * npx);
  const size_t up_bytes = o;
  // ---- the prefetch slot: both images' keys / descriptors, counts, uright / depth of the left image
  o = 0;
  t->s_kp = take((size_t)t->n_img * cap * sizeof(vieo_keypoint)), t->s_desc = take((size_t)t->n_img * cap * 32);
  t->s_ur = take((size_t)cap * 4 + 16), t->s_dp = take((size_t)cap * 4 + 16), t->s_cnt = take(64);
  const size_t slot_bytes = o;
  o = 0;
  t->l_cpt = take((size_t)ccap * sizeof(vieo_frustum_point)), t->l_cdesc = take((size_t)ccap * 32), t->l_xyz = take((size_t)ccap * 12);
  const size_t loc_bytes = o;
  o = 0;
  t->q_hdr = take(sizeof(TrkOut));
  t->q_ur = take((size_t)kc * 4), t->q_dp = take((size_t)kc * 4), t->q_mpref = take((size_t)kc * 4), t->q_outl = take(kc);
  t->q_kg = t->q_gidx = t->q_good = t->q_p3d = o;
  if (R) {
    t->q_kg = take((size_t)kc * 4), t->q_gidx = take((size_t)t->gcap * nc * 4), t->q_good = take(t->gcap);
    t->q_p3d = take((size_t)t->gcap * 24);
  }
  t->q_small_end = o;
  t->q_kp = take((size_t)kc * sizeof(vieo_keypoint)), t->q_desc = take((size_t)kc * 32), t->q_cdep = take((size_t)ccap * 4);
  t->out_bytes = o;
  o = 0;
  t->w_kp = take((size_t)t->n_img * cap * sizeof(vieo_keypoint)), t->w_desc = take((size_t)t->n_img * cap * 32);
  t->w_kcat = t->w_kp, t->w_dcat = t->w_desc;
  if (R) t->w_kcat = take((size_t)kc * sizeof(vieo_keypoint)), t->w_dcat = take((size_t)kc * 32);
  t->w_q1 = take((size_t)kc * nc * sizeof(vieo_proj_query)), t->w_q2 = take((size_t)ccap * nc * sizeof(vieo_proj_query));
  t->w_q1c = t->w_q1, t->w_qsrc = 0;
  if (R) t->w_q1c = take((size_t)kc * nc * sizeof(vieo_proj_query)), t->w_qsrc = take((size_t)kc * nc * 4);
  t->w_assign = take((size_t)kc * 4), t->w_taken = take(kc), t->w_held = take(t->pcap);
  t->w_obs = take((size_t)kc * sizeof(vieo_pose_obs)), t->w_obskey = take((size_t)kc * 4), t->w_outl = take(kc);
  t->w_xyz = take((size_t)t->pcap * 12), t->w_dep = take((size_t)t->pcap * 4);
  t->w_pre = take(sizeof(vieo_imu_preint)), t->w_prv = take(81 * 8), t->w_pst = take(16);
  const size_t work_bytes = o;
  o = 0;
  (void)take(sizeof(SpecImu));
  t->sp_samples = take((size_t)t->imu_cap * sizeof(vieo_imu_sample)), t->sp_up = o;
  t->sp_bias = take(6 * 8), t->sp_pre = take(sizeof(vieo_imu_preint)), t->sp_prv = take(81 * 8), t->sp_pst = take(16);
  const size_t spec_bytes = o;
  const size_t const_bytes = al256(sizeof(vieo_sbp_rig)) + al256(sizeof(vieo_camera) * 4);
  bool ok = hipHostMalloc((void**)&t->h_up, up_bytes, hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc((void**)&t->h_loc, loc_bytes, hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc((void**)&t->h_out, t->out_bytes, hipHostMallocDefault) == hipSuccess &&
            hipMalloc((void**)&t->d_up, up_bytes) == hipSuccess && hipMalloc((void**)&t->d_loc, loc_bytes) == hipSuccess &&
            hipMalloc((void**)&t->d_out, t->out_bytes) == hipSuccess && hipMalloc((void**)&t->d_work, work_bytes) == hipSuccess &&
            hipMalloc((void**)&t->d_const, const_bytes) == hipSuccess &&
            hipHostMalloc((void**)&t->h_spec, t->sp_up, hipHostMallocDefault) == hipSuccess &&
            hipMalloc((void**)&t->d_spec, spec_bytes) == hipSuccess && hipMemset(t->d_spec, 0, spec_bytes) == hipSuccess &&
            hipHostMalloc((void**)&t->h_next, t->n_img * npx, hipHostMallocDefault) == hipSuccess &&
            hipMalloc((void**)&t->d_next, t->n_img * npx) == hipSuccess && hipMalloc((void**)&t->d_slot, slot_bytes) == hipSuccess &&
            create_prefetch_stream(&t->st_pref) == hipSuccess &&
            hipEventCreateWithFlags(&t->ev_pref, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&t->ev_h2d, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&t->ev_spec, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&t->ev_head, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&t->ev_tab, hipEventDisableTiming) == hipSuccess &&
            create_side_stream(&t->st_imu, t->st, &t->side_ratio) == hipSuccess &&
            hipEventCreateWithFlags(&t->ev_up, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&t->ev_imu, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&t->ev_ext, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&t->ev_fe, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&t->ev_kd, hipEventDisableTiming) == hipSuccess &&
            hipEventCreate(&t->ev_t0) == hipSuccess && hipEventCreate(&t->ev_t1) == hipSuccess;
  if (!ok) {
    set_error("vieo_tracker_create: allocation failed (%s)", hipGetErrorString(hipGetLastError()));
    vieo_tracker_destroy(t);
    return VIEO_E_HIP;
  }
  memset(t->h_up, 0, up_bytes), memset(t->h_loc, 0, loc_bytes), memset(t->h_out, 0, t->out_bytes);
  (void)hipMemsetAsync(t->d_work, 0, work_bytes, t->st);
  (void)hipMemsetAsync(t->d_out, 0, t->out_bytes, t->st);
  // constant parts of the header
  TrkHdr& H = *(TrkHdr*)(t->h_up + t->o_hdr);
  const vieo_camera* c0 = R ? &R->cams[0] : nullptr;
  const float fx = c0 ? c0->fx : P->fx, fy = c0 ? c0->fy : P->fy, cx = c0 ? c0->cx : P->cx, cy = c0 ? c0->cy : P->cy;
  H.cam.fx = fx, H.cam.fy = fy, H.cam.cx = cx, H.cam.cy = cy;
  H.cam.bounds[0] = 0, H.cam.bounds[1] = (float)P->width, H.cam.bounds[2] = 0, H.cam.bounds[3] = (float)P->height;
  H.cam.bf = P->bf, H.cam.baseline = P->baseline, H.cam.th = P->th_last, H.cam.th_far = R ? R->th_far_pts : 0;
  H.cam.mono = 0, H.cam.nlevels = P->n_levels;
  for (int l = 0; l < P->n_levels; l++) H.cam.scale[l] = t->scale[l], H.consts[l] = t->inv_sigma2[l], H.consts[16 + l] = t->scale[l];
  for (int c = 0; c < 4; c++)
    t->bounds[c][0] = 0, t->bounds[c][1] = (float)P->width, t->bounds[c][2] = 0, t->bounds[c][3] = (float)P->height;
  vieo_camera* d_cams = (vieo_camera*)(t->d_const + al256(sizeof(vieo_sbp_rig)));
  for (vieo_vio_frame* f : {&H.f1, &H.f2}) {
    memcpy(f->base.Rcb, P->Rcb, 72), memcpy(f->base.tcb, P->tcb, 24);
    f->base.fx = fx, f->base.fy = fy, f->base.cx = cx, f->base.cy = cy, f->base.bf = P->bf;
    if (R) f->base.n_cams = R->n_cams, f->base.cams = d_cams;
    memcpy(f->gw, P->gw, 24);
    f->inv_sigma_bg2 = P->inv_sigma_bg2, f->inv_sigma_ba2 = P->inv_sigma_ba2, f->th_depth = P->th_depth;
  }
  H.f2.compute_marg = 1;
  H.noise = P->noise;
  memset(&t->pin_cam, 0, sizeof(t->pin_cam));
  t->pin_cam.fx = P->fx, t->pin_cam.fy = P->fy, t->pin_cam.cx = P->cx, t->pin_cam.cy = P->cy;
  memset(&t->ff, 0, sizeof(t->ff));
  t->ff.n_cams = 1, t->ff.use_distort = 0, t->ff.cams = &t->pin_cam;
  t->ff.Tcr[0][0] = t->ff.Tcr[0][5] = t->ff.Tcr[0][10] = 1.f;
  for (int c = 0; c < 4; c++) memcpy(t->ff.bounds[c], t->bounds[c], 16);
  t->ff.bf = P->bf, t->ff.n_levels = P->n_levels, t->ff.viewing_cos_limit = 0.5f;
  t->ff.log_scale_factor = logf(P->scale_factor);
  if (R) {
    // the searches' rig (Tcr / trc cast to double as mpCameras[c]->GetTcr().cast<double>()) and the cameras, once
    vieo_sbp_rig sr;
    memset(&sr, 0, sizeof(sr));
    sr.n_cams = R->n_cams, sr.use_distort = 1;
    t->ff.n_cams = R->n_cams, t->ff.use_distort = 1, t->ff.cams = t->R.cams;
    for (int c = 0; c < R->n_cams; c++) {
      sr.cams[c] = R->cams[c];
      memcpy(sr.Tcr[c], R->Tcr[c], 96);
      for (int r = 0; r < 3; r++) sr.trc[c][r] = R->Trc[c][r * 4 + 3], t->ff.trc[c][r] = (float)R->Trc[c][r * 4 + 3];
      for (int i = 0; i < 12; i++) t->ff.Tcr[c][i] = (float)R->Tcr[c][i];
      memcpy(sr.bounds[c], t->bounds[c], 16);
    }
    if (hipMemcpy(t->d_const, &sr, sizeof(sr), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_cams, R->cams, sizeof(vieo_camera) * R->n_cams, hipMemcpyHostToDevice) != hipSuccess) {
      set_error("vieo_tracker_create: constant upload failed");
      vieo_tracker_destroy(t);
      return VIEO_E_HIP;
    }
  }
  *out = t;
  return VIEO_OK;
}

int vieo_tracker_create(vieo_tracker** out, const vieo_tracker_params* P) { return vieo_tracker_create_rig(out, P, nullptr); }

int vieo_tracker_image_buffers(vieo_tracker* t, uint8_t** left, uint8_t** right) {
  if (!t || !left || !right) return VIEO_E_INVALID;
  *left = t->h_up + t->o_img;
  *right = *left + (size_t)t->P.width * t->P.height;
  return VIEO_OK;
}

int vieo_tracker_image_buffer(vieo_tracker* t, int image_index, uint8_t** plane) {
  if (!t || !plane || image_index < 0 || image_index >= t->n_img) return VIEO_E_INVALID;
  *plane = t->h_up + t->o_img + (size_t)image_index * t->P.width * t->P.height;
  return VIEO_OK;
}

int vieo_tracker_scale_factors(const vieo_tracker* t, float* h_out) {
  if (!t || !h_out) return VIEO_E_INVALID;
  for (int l = 0; l < t->P.n_levels; l++) h_out[l] = t->scale[l];
  return VIEO_OK;
}

// The second stream again: measured first, replaced only when it no longer runs beside the main one.  Both streams are
// idle here (every vieo_track_frame returns synchronised).
int vieo_tracker_reprobe(vieo_tracker* t) {
  if (!t) return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  VIEO_HIP_CHECK(hipStreamSynchronize(t->st));
  VIEO_HIP_CHECK(hipStreamSynchronize(t->st_imu));
  hipEvent_t e[3] = {nullptr, nullptr, nullptr};
  for (auto& ev : e) VIEO_HIP_CHECK(hipEventCreate(&ev));
  float ratio = 0;
  const bool ok = side_stream_ratio(t->st, t->st_imu, e[0], e[1], e[2], &ratio);
  for (auto& ev : e) (void)hipEventDestroy(ev);
  t->side_checks++;
  if (ok) t->side_ratio = ratio;
  if (ok && ratio < kSideRatioOk) return VIEO_OK;
  hipStream_t fresh = nullptr;
  float fr = 0;
  if (create_side_stream(&fresh, t->st, &fr) != hipSuccess) return VIEO_OK;  // (keep what there is)
  if (ok && fr >= ratio) {  // nothing better to be had
    (void)hipStreamDestroy(fresh);
    return VIEO_OK;
  }
  (void)hipStreamDestroy(t->st_imu);
  t->st_imu = fresh, t->side_ratio = fr, t->side_probes++;
  return VIEO_OK;
}

int vieo_tracker_get_stats(const vieo_tracker* t, vieo_tracker_stats* out) {
  if (!t || !out) return VIEO_E_INVALID;
  memset(out, 0, sizeof(*out));
  out->side_stream_ratio = t->side_ratio, out->side_stream_selections = t->side_probes, out->side_stream_checks = t->side_checks;
  out->replica_repeats = t->replica_repeats;
  out->frames_prefetched = t->pref_frames, out->preints_ahead_used = t->spec_used;
  const int n = std::min(t->gpu_n, 32);
  if (n > 0) {
    float v[32];
    memcpy(v, t->gpu_ring, sizeof(float) * n);
    std::nth_element(v, v + n / 2, v + n);
    out->ms_gpu_median = v[n / 2];
  }
  out->slow_frames_in_a_row = t->slow_run;
  return VIEO_OK;
}

int vieo_tracker_key_capacity(const vieo_tracker* t) { return t ? t->kc : 0; }
int vieo_tracker_group_capacity(const vieo_tracker* t) { return t ? t->gcap : 0; }

int vieo_tracker_get_level(vieo_tracker* t, int image_index, int level, int with_border, uint8_t* h_dst, int dst_stride) {
  if (!t) return VIEO_E_INVALID;
  return vieo_orb_get_level(t->ext, image_index, level, with_border, h_dst, dst_stride);
}

// the part of the chain behind the prediction: both searches and both optimisations
// the first search's queries from the predicted pose: SearchByProjection's projection of the last frame's points and, for
// rigs, the compaction of the valid (point, camera) pairs
static int track_project(vieo_tracker* t, hipStream_t s) {
  const int kc = t->kc, nc = t->nc;
  TrkHdr* dH = (TrkHdr*)(t->d_up + t->o_hdr);
  TrkOut* dO = (TrkOut*)(t->d_out + t->q_hdr);
  uint8_t* W = t->d_work;
  vieo_proj_query* d_q1 = (vieo_proj_query*)(W + t->w_q1);
  int rc;
  if (t->rig) {
    if ((rc = vieo_sbp_project_last_frame_rig_batch_device((const vieo_last_frame_point*)(t->d_up + t->o_pts), dH->npts, kc, 1,
                                                           &dH->cam, (const vieo_sbp_rig*)t->d_const, nc, d_q1, s)) != VIEO_OK)
      return rc;
    // one query per (last-frame key, camera): most project outside their camera -- the search walks the valid ones only
    return vieo_track_compact_queries_batch_device(d_q1, dH->npts + 1, kc * nc, 1, (vieo_proj_query*)(W + t->w_q1c),
                                                   (int32_t*)(W + t->w_qsrc), dO->nq + 1, s);
  }
  return vieo_sbp_project_last_frame_batch_device((const vieo_last_frame_point*)(t->d_up + t->o_pts), dH->npts, kc, 1, &dH->cam,
                                                  d_q1, s);
}

}  // extern "C" (a template cannot have C linkage)

// projected: the first search's queries are there already (the prediction, the projection of the last frame's points and,
// for rigs, their compaction ran on the second stream beside the extraction); false for the repeat with the wider window
// side_rest: what the caller still has to hand to the second stream (recorded into ev_tab / ev_kd / ev_fe there); called
// behind the first optimisation's launch
template <class SideRest>
static int track_chain_tail(vieo_tracker* t, int nc_local, bool projected, SideRest&& side_rest) {
  const vieo_tracker_params& P = t->P;
  const int kc = t->kc, nc = t->nc;
  hipStream_t st = t->st;
  TrkHdr* dH = (TrkHdr*)(t->d_up + t->o_hdr);
  TrkOut* dO = (TrkOut*)(t->d_out + t->q_hdr);
  uint8_t* W = t->d_work;
  const vieo_keypoint* d_kp = (const vieo_keypoint*)(W + t->w_kcat);  // mvKeys of the frame
  const uint8_t* d_desc = W + t->w_dcat;
  float* d_ur = (float*)(t->d_out + t->q_ur);
  int32_t* d_mpref = (int32_t*)(t->d_out + t->q_mpref);
  // the frame's key count in the {n, -} layout the glue reads: the left image's counts / the rig's {N, 0}
  const int32_t* d_cnt = t->rig ? dO->fcnt : dO->cnt;
  vieo_proj_query* d_q1 = (vieo_proj_query*)(W + t->w_q1);
  vieo_proj_query* d_q2 = (vieo_proj_query*)(W + t->w_q2);
  int32_t* d_assign = (int32_t*)(W + t->w_assign);
  uint8_t* d_taken = W + t->w_taken;
  uint8_t* d_held = W + t->w_held;
  vieo_pose_obs* d_obs = (vieo_pose_obs*)(W + t->w_obs);
  int32_t* d_obskey = (int32_t*)(W + t->w_obskey);
  uint8_t* d_outl = W + t->w_outl;
  float* d_xyz = (float*)(W + t->w_xyz);
  float* d_dep = (float*)(W + t->w_dep);
  const float close = std::max(10.0f, P.th_depth);
  const int vio = t->vision ? 0 : 1;
  void* f1 = t->vision ? (void*)&dH->f1.base : (void*)&dH->f1;
  void* f2 = t->vision ? (void*)&dH->f2.base : (void*)&dH->f2;
  void* r1 = t->vision ? (void*)&dO->r1.base : (void*)&dO->r1;
  int rc;
#define TRK(call)                       \
  do {                                  \
    if ((rc = (call)) != VIEO_OK) return rc; \
  } while (0)
  auto search = [&](int mode, const vieo_proj_query* q, const int32_t* d_nq, int q_cap, const uint8_t* taken, float nn, int32_t* d_nm) {
    if (t->rig)
      return vieo_search_by_projection_rig_batch_device(mode, q, d_nq, q_cap, 1, d_kp, d_ur, d_desc, taken, dO->cam_first, kc,
                                                        &t->bounds[0][0], nc, nn, 1, d_assign, d_nm, st);
    return vieo_search_by_projection_batch_device(mode, q, d_nq, q_cap, 1, d_kp, d_ur, d_desc, taken, d_cnt, kc, 0, 2,
                                                  t->bounds[0], nn, 1, d_assign, d_nm, st);
  };
  // the search's assignment merged into the frame's point table and the observations gathered from it: one launch
  auto merge_build_obs = [&](void* frame, int point_offset, int reset, const vieo_last_frame_point* same_point, const int32_t* query_src,
                             int q_cap) {
    return vieo_track_merge_build_obs_batch_device(d_assign, d_mpref, point_offset, reset, nc, same_point, query_src, q_cap, d_xyz,
                                                   t->vision ? nullptr : d_dep, close, t->pcap, d_kp, d_ur, d_cnt,
                                                   t->rig ? dO->cam_first : nullptr, nc, kc, 1, 0, t->rig ? 1 : 2, dH->consts, d_obs,
                                                   d_obskey, frame, vio, st);
  };
  auto pose = [&](void* frame, void* result) {
    if (t->vision)
      return vieo_pose_optimization_batch_device_ex((const vieo_pose_frame*)frame, 1, d_obs, d_outl, (vieo_pose_result*)result,
                                                    VIEO_POSE_CAMS_RECTIFIED, st);
    return vieo_pose_optimization_vio_batch_device_ex((const vieo_vio_frame*)frame, 1, d_obs, d_outl, (vieo_vio_result*)result,
                                                      t->rig ? VIEO_POSE_CAMS_RIG : VIEO_POSE_CAMS_RECTIFIED, VIEO_POSE_ENC_NONE, st);
  };
  if (!projected) TRK(track_project(t, st));
  if (t->rig) {
    vieo_proj_query* d_q1c = (vieo_proj_query*)(W + t->w_q1c);
    int32_t* d_qsrc = (int32_t*)(W + t->w_qsrc);
    TRK(search(VIEO_SBP_LAST_FRAME, d_q1c, dO->nq + 1, kc * nc, nullptr, P.nn_last, dO->nm));
    TRK(merge_build_obs(f1, 0, 1, (const vieo_last_frame_point*)(t->d_up + t->o_pts), d_qsrc, kc * nc));
  } else {
    TRK(search(VIEO_SBP_LAST_FRAME, d_q1, dH->npts + 1, kc * nc, nullptr, P.nn_last, dO->nm));
    TRK(merge_build_obs(f1, 0, 1, nullptr, nullptr, 0));
  }
  TRK(pose(f1, r1));
  // a changed local map travels on the second stream; nothing before this line reads it
  TRK(side_rest());
  if (t->tab_pending) {
    if (hipStreamWaitEvent(st, t->ev_tab, 0) != hipSuccess) return VIEO_E_HIP;
    t->tab_pending = false;
  }
  TRK(vieo_track_after_pose_held_batch_device(d_mpref, d_obskey, d_outl, f1, r1, vio, kc, 1, f2, d_taken, d_cnt, 0, 2, d_held, t->pcap, st));
  // (the kernel reads only the leading vieo_pose_frame / vieo_pose_result of its two arguments)
  TRK(vieo_track_local_queries_device(&t->ff, (const vieo_vio_frame*)f1, (const vieo_vio_result*)r1,
                                      (const vieo_frustum_point*)(t->d_loc + t->l_cpt), t->d_loc + t->l_cdesc,
                                      (const int32_t*)(t->d_up + t->o_alias), d_held, t->pcap, nc_local, P.th_local,
                                      t->rig ? t->R.th_far_pts : 0.f, dH->consts + 16, d_q2, d_dep + kc, dO->nq, st));
  (void)vieo_sbp_keep_grid(1);  // the frame's keys have not changed since the first search
  TRK(search(VIEO_SBP_LOCAL_MAP, d_q2, dO->nq, t->ccap * nc, d_taken, P.nn_local, dO->nm + 1));
  TRK(merge_build_obs(f2, kc, 0, nullptr, nullptr, 0));
  TRK(pose(f2, t->vision ? (void*)&dO->r2.base : (void*)&dO->r2));
#undef TRK
  hipLaunchKernelGGL(k_track_finish, dim3(1), dim3(256), 0, st, d_obskey, d_outl, &dH->f2, t->d_out + t->q_outl, kc, dO);
  VIEO_HIP_CHECK(hipGetLastError());
  // results: [header | uright | depth | point_ref | outlier | (rig: key -> group, the groups)] and the candidates' depths
  // (the keys / descriptors are copied by the caller, once)
  if (t->rig) VIEO_HIP_CHECK(hipStreamWaitEvent(st, t->ev_fe, 0));  // the stereo groups of the frame (second stream)
  VIEO_HIP_CHECK(hipMemcpyAsync(t->h_out, t->d_out, t->q_small_end, hipMemcpyDeviceToHost, st));
  if (nc_local > 0) VIEO_HIP_CHECK(hipMemcpyAsync(t->h_out + t->q_cdep, d_dep + kc, (size_t)nc_local * 4, hipMemcpyDeviceToHost, st));
  return VIEO_OK;
}

extern "C" {

// an error in the middle of the chain: work queued on the two streams still reads the pinned blocks, which the next
// call would overwrite -- wait for it before handing the error back
static int track_fail(vieo_tracker* t, int rc) {
  if (t->st_pref) (void)hipStreamSynchronize(t->st_pref), t->pref_valid = false;
  t->spec_valid = false;
  (void)hipStreamSynchronize(t->st_imu);
  (void)hipStreamSynchronize(t->st);
  return rc;
}

int vieo_track_frame(vieo_tracker* t, const vieo_track_input* in, vieo_track_output* out) {
  if (!t || !in || !out || in->stride < t->P.width || in->n_imu < 0 || (in->n_imu > 0 && !in->imu) ||
      in->n_last < 0 || (in->n_last > 0 && (!in->last_points || !in->last_track_depth)) || in->n_local < 0 ||
      (in->n_local > 0 && !in->local_alias))
    return VIEO_E_INVALID;
  const uint8_t* imgs[4] = {in->left, in->right, nullptr, nullptr};
  if (t->rig)
    for (int c = 0; c < 4; c++) imgs[c] = in->images[c];
  if (!in->use_prefetched)  // (a prefetched frame's images are not read: vieo_hot.h says they may be null then)
    for (int c = 0; c < t->n_img; c++)
      if (!imgs[c]) {
        set_error("vieo_track_frame: image %d is null (only a call with use_prefetched = 1 may leave the images out)", c);
        return VIEO_E_INVALID;
      }
  const auto t_enter = std::chrono::steady_clock::now();
  const vieo_tracker_params& P = t->P;
  const int cap = t->cap, kc = t->kc, W = P.width, Hh = P.height;
  if (in->n_last > kc || in->n_local > t->ccap || in->n_imu > t->imu_cap) {
    set_error("vieo_track_frame: %d last-frame points / %d local points / %d IMU samples exceed the capacities %d / %d / %d",
              in->n_last, in->n_local, in->n_imu, kc, t->ccap, t->imu_cap);
    return VIEO_E_CAPACITY;
  }
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  hipStream_t st = t->st;
  const size_t npx = (size_t)W * Hh;
  const int nl = in->n_last, nc = in->n_local;
  // frame pipelining: was this frame extracted beside the previous call's tail?
  if (in->use_prefetched && !t->pref_valid) {
    set_error("vieo_track_frame: use_prefetched without a pending prefetch (the previous call carried no next_left / next_right)");
    return VIEO_E_INVALID;
  }
  // the next frame's images, if the caller has them: next_left / next_right (rectified pair) or next_images[c] (rig)
  const uint8_t* nx[4] = {in->next_left, in->next_right, nullptr, nullptr};
  if (t->rig)
    for (int c = 0; c < 4; c++) nx[c] = in->next_images[c];
  int n_next = 0;
  for (int c = 0; c < t->n_img; c++) n_next += nx[c] != nullptr;
  if (n_next != 0 && n_next != t->n_img) {
    set_error("vieo_track_frame: the next frame's images come complete (%d of %d given)", n_next, t->n_img);
    return VIEO_E_INVALID;
  }
  const bool pref = in->use_prefetched != 0;
  if (!pref && t->pref_valid) {  // a pending prefetch the caller does not want: let it finish, forget it
    (void)hipStreamSynchronize(t->st_pref);
    t->pref_valid = false;
  }
  // ---- the upload block
  TrkHdr& H = *(TrkHdr*)(t->h_up + t->o_hdr);
  H.cam.th = P.th_last;
  H.nav_ref = in->nav_ref, H.nav_last = in->nav_last;
  for (vieo_vio_frame* f : {&H.f1, &H.f2}) {
    f->nav_last = in->nav_ref;
    f->dt_frames = in->t_cur - in->t_ref;
    f->last_has_prior = in->nav_prior && in->H_prior ? 1 : 0;
    if (f->last_has_prior) f->nav_prior = *in->nav_prior, memcpy(f->H_prior, in->H_prior, sizeof(f->H_prior));
    f->base.n_obs = 0, f->base.obs_begin = 0;
  }
  H.ti = in->t_ref, H.tj = in->t_cur;
  for (int k = 0; k < 3; k++) H.bg[k] = in->nav_ref.bg[k], H.ba[k] = in->nav_ref.ba[k];
  H.first[0] = 0, H.first[1] = in->n_imu;
  H.npts[0] = nl, H.npts[1] = nl * t->nc;
  if (in->n_imu) memcpy(t->h_up + t->o_imu, in->imu, (size_t)in->n_imu * sizeof(vieo_imu_sample));
  uint8_t* img = t->h_up + t->o_img;
  for (int c = 0; c < (pref ? 0 : t->n_img); c++) {
    const uint8_t* src = imgs[c];
    uint8_t* dst = img + c * npx;
    if (src == dst) continue;  // decoded straight into the pinned plane
    if (in->stride == W)
      memcpy(dst, src, npx);
    else
      for (int y = 0; y < Hh; y++) memcpy(dst + (size_t)y * W, src + (size_t)y * in->stride, W);
  }
  if (nl) {
    memcpy(t->h_up + t->o_pts, in->last_points, (size_t)nl * sizeof(vieo_last_frame_point));
    float* xyz = (float*)(t->h_up + t->o_xyz);
    for (int i = 0; i < nl; i++) {
      const float* X = in->last_points[i].Xw;
      xyz[3 * i] = X[0], xyz[3 * i + 1] = X[1], xyz[3 * i + 2] = X[2];
    }
    memcpy(t->h_up + t->o_dep, in->last_track_depth, (size_t)nl * 4);
  }
  if (nc) memcpy(t->h_up + t->o_alias, in->local_alias, (size_t)nc * 4);
  const bool new_local = nc > 0 && (in->local_version != t->local_version || nc != t->n_local_dev);
  if (new_local) {
    if (!in->local_points || !in->local_desc) return VIEO_E_INVALID;
    memcpy(t->h_loc + t->l_cpt, in->local_points, (size_t)nc * sizeof(vieo_frustum_point));
    memcpy(t->h_loc + t->l_cdesc, in->local_desc, (size_t)nc * 32);
    float* xyz = (float*)(t->h_loc + t->l_xyz);
    for (int i = 0; i < nc; i++) {
      const float* X = in->local_points[i].Xw;
      xyz[3 * i] = X[0], xyz[3 * i + 1] = X[1], xyz[3 * i + 2] = X[2];
    }
  }
  // ---- one copy up (+ the local map when it changed), the chain, the copies back
  uint8_t* Wk = t->d_work;
#define TRK_HIP(expr)                                                                                 \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess) {                                                                           \
      vieo::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_));           \
      return track_fail(t, VIEO_E_HIP);                                                               \
    }                                                                                                 \
  } while (0)
  TRK_HIP(hipEventRecord(t->ev_t0, st));
  TRK_HIP(hipMemcpyAsync(t->d_up, t->h_up, t->up_fixed + (size_t)nc * 4, hipMemcpyHostToDevice, st));
  TRK_HIP(hipEventRecord(t->ev_up, st));
  TrkHdr* dH = (TrkHdr*)(t->d_up + t->o_hdr);
  TrkOut* dO = (TrkOut*)(t->d_out + t->q_hdr);
  vieo_keypoint* d_kp = (vieo_keypoint*)(Wk + t->w_kp);
  uint8_t* d_desc = Wk + t->w_desc;
  const int* lapping = t->rig && t->R.use_lapping ? t->R.lapping : nullptr;
  if (pref) {
    // the frame was extracted (and its stereo stage run) on the third stream beside the previous call's tail: adopt it
    TRK_HIP(hipStreamWaitEvent(st, t->ev_pref, 0));
    const int n_kp16 = (int)(((size_t)t->n_img * cap * sizeof(vieo_keypoint) + 15) / 16), n_desc16 = t->n_img * cap * 2;
    const int n_f16 = t->rig ? 0 : (cap + 3) / 4;  // (a rig frame's stereo stage runs in the call: it fills mvKeys-order tables)
    hipLaunchKernelGGL(k_track_adopt, dim3(32), dim3(256), 0, st, (const uint4*)(t->d_slot + t->s_kp), (uint4*)d_kp, n_kp16,
                       (const uint4*)(t->d_slot + t->s_desc), (uint4*)d_desc, n_desc16, (const uint4*)(t->d_slot + t->s_ur),
                       (uint4*)(t->d_out + t->q_ur), (const uint4*)(t->d_slot + t->s_dp), (uint4*)(t->d_out + t->q_dp), n_f16,
                       (const int32_t*)(t->d_slot + t->s_cnt), dO->cnt, 2 * t->n_img);
    t->pref_valid = false;
  } else {
    TRK_HIP(hipMemcpyAsync(t->d_up + t->o_img, t->h_up + t->o_img, t->n_img * npx, hipMemcpyHostToDevice, st));
    if ((rc = vieo_orb_extract_batch_device(t->ext, t->d_up + t->o_img, t->n_img, W, Hh, W, npx, lapping, d_kp, d_desc, cap, dO->cnt)) != VIEO_OK)
      return track_fail(t, rc);
  }
  // Everything that needs nothing of the new images goes to the second stream, and is handed to it AFTER the extraction's
  // launches: those are the head of the critical path (the host spends 20-30 us on the second stream's five to eight
  // launches, and the first pyramid level used to wait for them).
  const TrkTables tables{(const float*)(t->d_up + t->o_xyz), (const float*)(t->d_up + t->o_dep), (float*)(Wk + t->w_xyz), (float*)(Wk + t->w_dep)};
  // (a frame extracted ahead: there is no extraction to run beside, the prediction and the projection are the next links of
  // the chain itself -- on this stream, without the two event hops to the second one and back: ~12 us)
  const hipStream_t s_head = pref ? st : t->st_imu;
  // the next frame's pre-integration run ahead: with this frame as its reference (behind this frame's prediction, in
  // side_rest below) or with a reference the caller names (spec_ref, started right behind the prediction's launch)
  const bool spec_any = !t->vision && in->next_imu && in->next_n_imu > 0 && in->next_n_imu <= t->imu_cap;
  const bool spec_ref = spec_any && in->next_ref_bias != nullptr, spec = spec_any && !spec_ref;
  const auto spec_launch = [&](double t_ref, const double* ref_bias) -> int {  // ref_bias: null = this frame's predicted bias
    SpecImu& S = *(SpecImu*)t->h_spec;
    S.noise = H.noise, S.ti = t_ref, S.tj = in->next_t_cur, S.first[0] = 0, S.first[1] = in->next_n_imu;
    for (int k = 0; k < 6; k++) S.xbias[k] = ref_bias ? ref_bias[k] : 0.0;
    memcpy(t->h_spec + t->sp_samples, in->next_imu, (size_t)in->next_n_imu * sizeof(vieo_imu_sample));
    VIEO_HIP_CHECK(hipMemcpyAsync(t->d_spec, t->h_spec, t->sp_samples + (size_t)in->next_n_imu * sizeof(vieo_imu_sample), hipMemcpyHostToDevice, t->st_imu));
    SpecImu* dS = (SpecImu*)t->d_spec;
    const double* bias = ref_bias ? dS->xbias : (const double*)(t->d_spec + t->sp_bias);
    const int rc_pre = vieo_imu_preintegrate_batch_device(&dS->noise, (const vieo_imu_sample*)(t->d_spec + t->sp_samples), dS->first, &dS->ti,
                                                          &dS->tj, bias, bias + 3, 1, (vieo_imu_preint*)(t->d_spec + t->sp_pre),
                                                          (double*)(t->d_spec + t->sp_prv), (int32_t*)(t->d_spec + t->sp_pst), t->st_imu);
    if (rc_pre != VIEO_OK) return rc_pre;
    VIEO_HIP_CHECK(hipEventRecord(t->ev_spec, t->st_imu));
    t->spec_samples.assign(in->next_imu, in->next_imu + in->next_n_imu);
    t->spec_n = in->next_n_imu, t->spec_ti = t_ref, t->spec_tj = in->next_t_cur;
    return VIEO_OK;
  };
  if (!t->vision) {
    // the pre-integration beside the extraction -- or, run ahead by the previous call (next_imu), if what that call
    // integrated is bit for bit what this call asks for
    const bool ahead = t->spec_valid && in->n_imu == t->spec_n && in->t_ref == t->spec_ti && in->t_cur == t->spec_tj &&
                       (in->n_imu == 0 || memcmp(in->imu, t->spec_samples.data(), (size_t)in->n_imu * sizeof(vieo_imu_sample)) == 0) &&
                       memcmp(in->nav_ref.bg, t->spec_bias, 24) == 0 && memcmp(in->nav_ref.ba, t->spec_bias + 3, 24) == 0;
    t->spec_valid = false;
    const uint8_t* pre_at = ahead ? t->d_spec + t->sp_pre : Wk + t->w_pre;
    const uint8_t* prv_at = ahead ? t->d_spec + t->sp_prv : Wk + t->w_prv;
    const uint8_t* pst_at = ahead ? t->d_spec + t->sp_pst : Wk + t->w_pst;
    if (!pref) TRK_HIP(hipStreamWaitEvent(t->st_imu, t->ev_up, 0));
    if (ahead) {
      t->spec_used++;
      // (it ran on the second stream -- which vieo_tracker_reprobe may have replaced since: wait for it by its event either way)
      TRK_HIP(hipStreamWaitEvent(s_head, t->ev_spec, 0));
    } else if ((rc = vieo_imu_preintegrate_batch_device(&dH->noise, (const vieo_imu_sample*)(t->d_up + t->o_imu), dH->first, &dH->ti,
                                                        &dH->tj, dH->bg, dH->ba, 1, (vieo_imu_preint*)(Wk + t->w_pre),
                                                        (double*)(Wk + t->w_prv), (int32_t*)(Wk + t->w_pst), s_head)) != VIEO_OK)
      return track_fail(t, rc);
    hipLaunchKernelGGL(k_track_predict, dim3(1 + kTableBlocks), dim3(64), 0, s_head, dH, dO, (const vieo_imu_preint*)pre_at,
                       (const double*)prv_at, (const int32_t*)pst_at, (double*)(t->d_spec + t->sp_bias), tables);
  } else {
    if (!pref) TRK_HIP(hipStreamWaitEvent(t->st_imu, t->ev_up, 0));
    hipLaunchKernelGGL(k_track_set_pose, dim3(1 + kTableBlocks), dim3(64), 0, s_head, dH, dO, tables);
  }
  // PredictNavStateByIMU and the projection of the last frame's points need nothing of the new images: beside the extraction
  TRK_HIP(hipGetLastError());
  if ((rc = track_project(t, s_head)) != VIEO_OK) return track_fail(t, rc);
  if (!pref) TRK_HIP(hipEventRecord(t->ev_imu, t->st_imu));  // the prediction and the first search's queries: what the search waits for
  // ComputeStereoFishEyeMatches (Frame.cc:613-779) into mvKeys order: keys / descriptors in the work block, the tables in
  // the download block.
  // What tracking reads of it -- the concatenated keys and descriptors, the cameras' ranges, uright = -1 -- does not
  // depend on the matches: that part stays on this stream (one short kernel), the matches / groups / depths, which are
  // outputs of the frame only (knn-2, pairs, FillMatchesFromPair's walk, re-triangulation: 0.3 ms of a 4-camera frame),
  // run on the second stream beside the two searches and optimisations and are joined before the copy back.
  auto fe_part = [&](int part, hipStream_t s) {
    return vieo_stereo_fisheye_match_batch_device_part(
        t->fe, d_kp, d_desc, dO->cnt, 1, (vieo_keypoint*)(Wk + t->w_kcat), Wk + t->w_dcat, dO->cam_first, dO->fcnt,
        (float*)(t->d_out + t->q_dp), (float*)(t->d_out + t->q_ur), (int32_t*)(t->d_out + t->q_kg),
        (int32_t*)(t->d_out + t->q_gidx), t->d_out + t->q_good, (double*)(t->d_out + t->q_p3d), dO->fe_hdr, part, s);
  };
  if (t->rig)
    rc = fe_part(VIEO_FISHEYE_CONCAT, st);
  else if (!pref)
    rc = vieo_stereo_match_rectified_batch_device(t->ext, 1, d_kp, d_desc, dO->cnt, cap, P.baseline, P.bf, (float*)(t->d_out + t->q_ur),
                                                  (float*)(t->d_out + t->q_dp));
  if (rc != VIEO_OK) return track_fail(t, rc);
  TRK_HIP(hipEventRecord(t->ev_head, st));  // the extractor's pyramids and scratch are free from here on
  // the frame's keys / descriptors (mvKeys / mDescriptors: the left image's, or the rig's concatenation) are final here:
  // what the second stream reads of this one (the rig's groups, the keys' / descriptors' copies back)
  TRK_HIP(hipEventRecord(t->ev_ext, st));
  // The rest of the second stream's work is read by the tail behind the first optimisation at the earliest: the host hands
  // it over AFTER that kernel's launch.  (Once the frame was extracted ahead the host's launches are the head of the
  // critical path: twenty runtime calls at 2-5 us each used to stand between k_track_adopt and the first search kernel;
  // the first optimisation's 0.25 ms is where the host gets ahead again.)
  const auto side_rest = [&]() -> int {
    if (new_local) {
      TRK_HIP(hipMemcpyAsync(t->d_loc + t->l_cpt, t->h_loc + t->l_cpt, (size_t)nc * sizeof(vieo_frustum_point), hipMemcpyHostToDevice, t->st_imu));
      TRK_HIP(hipMemcpyAsync(t->d_loc + t->l_cdesc, t->h_loc + t->l_cdesc, (size_t)nc * 32, hipMemcpyHostToDevice, t->st_imu));
      TRK_HIP(hipMemcpyAsync(Wk + t->w_xyz + (size_t)kc * 12, t->h_loc + t->l_xyz, (size_t)nc * 12, hipMemcpyHostToDevice, t->st_imu));
      t->local_version = in->local_version, t->n_local_dev = nc;
      TRK_HIP(hipEventRecord(t->ev_tab, t->st_imu));  // (joined by the tail before the local-map queries)
      t->tab_pending = true;
    }
    TRK_HIP(hipStreamWaitEvent(t->st_imu, t->ev_ext, 0));
    if (t->rig) {
      const int rc_fe = fe_part(VIEO_FISHEYE_GROUPS, t->st_imu);
      if (rc_fe != VIEO_OK) return rc_fe;
      TRK_HIP(hipEventRecord(t->ev_fe, t->st_imu));
    }
    if (spec) {  // the next frame's samples go up with this frame's copies (the pinned block is free again when the call returns)
      SpecImu& S = *(SpecImu*)t->h_spec;
      S.noise = H.noise, S.ti = in->t_cur, S.tj = in->next_t_cur, S.first[0] = 0, S.first[1] = in->next_n_imu;
      memcpy(t->h_spec + t->sp_samples, in->next_imu, (size_t)in->next_n_imu * sizeof(vieo_imu_sample));
      TRK_HIP(hipMemcpyAsync(t->d_spec, t->h_spec, t->sp_samples + (size_t)in->next_n_imu * sizeof(vieo_imu_sample), hipMemcpyHostToDevice, t->st_imu));
    }
    TRK_HIP(hipMemcpyAsync(t->h_out + t->q_kp, Wk + t->w_kcat, (size_t)kc * sizeof(vieo_keypoint), hipMemcpyDeviceToHost, t->st_imu));
    TRK_HIP(hipMemcpyAsync(t->h_out + t->q_desc, Wk + t->w_dcat, (size_t)kc * 32, hipMemcpyDeviceToHost, t->st_imu));
    TRK_HIP(hipEventRecord(t->ev_kd, t->st_imu));
    if (spec) {
      // PreIntegration of [t_cur, next_t_cur] with bj_bar (k_track_predict left it in the block), behind this frame's
      // copies on the second stream and beside its tail; the next call's k_track_predict is behind it on the same stream
      SpecImu* dS = (SpecImu*)t->d_spec;
      const double* bias = (const double*)(t->d_spec + t->sp_bias);
      const int rc_pre = vieo_imu_preintegrate_batch_device(&dS->noise, (const vieo_imu_sample*)(t->d_spec + t->sp_samples), dS->first,
                                                            &dS->ti, &dS->tj, bias, bias + 3, 1, (vieo_imu_preint*)(t->d_spec + t->sp_pre),
                                                            (double*)(t->d_spec + t->sp_prv), (int32_t*)(t->d_spec + t->sp_pst), t->st_imu);
      if (rc_pre != VIEO_OK) return rc_pre;
      TRK_HIP(hipEventRecord(t->ev_spec, t->st_imu));
      t->spec_samples.assign(in->next_imu, in->next_imu + in->next_n_imu);
      t->spec_n = in->next_n_imu, t->spec_ti = in->t_cur, t->spec_tj = in->next_t_cur;
    }
    // ... or of [next_t_ref, next_t_cur] with the bias the caller names (next_ref_bias: the next call's reference is a key
    // frame).  Measured and dropped: starting this one right behind the prediction -- nothing of this frame enters it -- on
    // the second stream (tracking call 0.730 against 0.717 ms here) or on a stream of its own (a fourth stream of the
    // tracker shares a hardware queue with somebody: the local BA beside it went from 3.2 to 3.9 ms).
    if (spec_ref) {
      const int rc_pre = spec_launch(in->next_t_ref, in->next_ref_bias);
      if (rc_pre != VIEO_OK) return rc_pre;
    }
    return VIEO_OK;
  };
  if (!pref) TRK_HIP(hipStreamWaitEvent(st, t->ev_imu, 0));  // the prediction and the first search's queries (second stream)
  if ((rc = track_chain_tail(t, nc, true, side_rest)) != VIEO_OK) return track_fail(t, rc);
  TRK_HIP(hipStreamWaitEvent(st, t->ev_kd, 0));  // (the keys' / descriptors' copies)
  TRK_HIP(hipEventRecord(t->ev_t1, st));
  if (n_next) {
    // ---- the NEXT frame's Frame::Frame on the third stream, beside this frame's searches and optimisations (which are
    // queued by now): copy its images to the pinned planes (the host is otherwise about to wait), up, ExtractORB x n_img,
    // (rectified pairs: ComputeStereoMatches) into the slot the next call adopts
    // (the previous prefetch's copy up left these planes long ago -- it was queued behind that frame's stereo stage and this
    // host thread has waited for that frame's whole chain since --, but nothing in the stream order says so: ask)
    if (t->h2d_pending) TRK_HIP(hipEventSynchronize(t->ev_h2d));
    for (int c = 0; c < t->n_img; c++) {
      uint8_t* dst = t->h_next + c * npx;
      if (in->stride == W)
        memcpy(dst, nx[c], npx);
      else
        for (int y = 0; y < Hh; y++) memcpy(dst + (size_t)y * W, nx[c] + (size_t)y * in->stride, W);
    }
    hipStream_t sp = t->st_pref;
    TRK_HIP(hipStreamWaitEvent(sp, t->ev_head, 0));
    TRK_HIP(hipMemcpyAsync(t->d_next, t->h_next, t->n_img * npx, hipMemcpyHostToDevice, sp));
    TRK_HIP(hipEventRecord(t->ev_h2d, sp));
    t->h2d_pending = true;
    vieo_keypoint* s_kp = (vieo_keypoint*)(t->d_slot + t->s_kp);
    uint8_t* s_desc = t->d_slot + t->s_desc;
    int32_t* s_cnt = (int32_t*)(t->d_slot + t->s_cnt);
    hipStream_t keep = t->ext->stream;
    t->ext->stream = sp;  // (the extractor and the stereo matcher launch on the handle's stream)
    rc = vieo_orb_extract_batch_device(t->ext, t->d_next, t->n_img, W, Hh, W, npx, lapping, s_kp, s_desc, cap, s_cnt);
    if (rc == VIEO_OK && !t->rig)
      rc = vieo_stereo_match_rectified_batch_device(t->ext, 1, s_kp, s_desc, s_cnt, cap, P.baseline, P.bf, (float*)(t->d_slot + t->s_ur),
                                                    (float*)(t->d_slot + t->s_dp));
    t->ext->stream = keep;
    if (rc != VIEO_OK) {
      (void)hipStreamSynchronize(sp);
      return track_fail(t, rc);
    }
    TRK_HIP(hipEventRecord(t->ev_pref, sp));
    t->pref_valid = true, t->pref_frames++;
  }
  TRK_HIP(hipStreamSynchronize(st));
  const TrkOut* O = (const TrkOut*)(t->h_out + t->q_hdr);
  const bool pre_ok = t->vision || (O->preint_status[0] == 0 && O->imu.dt != 0);
  int widened = 0;
  if (O->nm[0] < 20 && pre_ok) {
    // Tracking.cc:301-309 / :1869-1876: the wider window.  Only the search threshold changes; everything before the
    // projection is still in HBM
    widened = 1;
    const float th2 = 2 * P.th_last;
    TRK_HIP(hipMemcpyAsync(&dH->cam.th, &th2, 4, hipMemcpyHostToDevice, st));
    if ((rc = track_chain_tail(t, nc, false, [] { return (int)VIEO_OK; })) != VIEO_OK) return track_fail(t, rc);
    TRK_HIP(hipEventRecord(t->ev_t1, st));
    TRK_HIP(hipStreamSynchronize(st));
  }
  if (t->rig && (O->r1.base.status == VIEO_E_HIP || O->r2.base.status == VIEO_E_HIP)) {
    // a replica of one of the two optimisations never became resident (vieo_pose_set_replicas in include/vieo_hot.h: the
    // device is shared with other work): the tail again with one workgroup per optimisation -- the frame is late, not lost
    const int was = vieo_pose_set_replicas(0);
    rc = track_chain_tail(t, nc, false, [] { return (int)VIEO_OK; });
    (void)vieo_pose_set_replicas(was);
    if (rc != VIEO_OK) return track_fail(t, rc);
    TRK_HIP(hipEventRecord(t->ev_t1, st));
    TRK_HIP(hipStreamSynchronize(st));
    t->replica_repeats++;
  }
#undef TRK_HIP
  if (spec || spec_ref) {  // (the bias the run-ahead integration used: the next call's nav_ref must carry exactly it)
    for (int k = 0; k < 3; k++) t->spec_bias[k] = O->nav_pred.bg[k], t->spec_bias[3 + k] = O->nav_pred.ba[k];
    if (spec_ref)
      for (int k = 0; k < 6; k++) t->spec_bias[k] = in->next_ref_bias[k];
    t->spec_valid = true;
  }
  memset(out, 0, sizeof(*out));
  out->preint_status = O->preint_status[0];
  // Tracking.cc:311 (fewer than 10 matches with the IMU) / :1878 (fewer than 20 without): the reference returns before
  // the optimisations; here they have run, their outputs are to be ignored
  out->status = !pre_ok ? VIEO_TRACK_PREINT_FAILED : (O->nm[0] < (t->vision ? 20 : 10) ? VIEO_TRACK_LOST : VIEO_TRACK_OK);
  if (t->rig) {
    out->n_keys = std::min(O->cam_first[t->nc], kc);
    for (int c = 0; c <= t->nc; c++) out->cam_first[c] = O->cam_first[c];
    for (int c = 0; c < t->nc; c++) out->mono_index[c] = O->cnt[2 * c + 1];
    out->stereo_status = O->fe_hdr[3], out->n_groups = O->fe_hdr[3] ? 0 : O->fe_hdr[0], out->n_stereo_matches = O->fe_hdr[1];
    out->key_group = (const int32_t*)(t->h_out + t->q_kg), out->group_idx = (const int32_t*)(t->h_out + t->q_gidx);
    out->group_good = t->h_out + t->q_good, out->group_p3d = (const double*)(t->h_out + t->q_p3d);
  } else {
    out->n_keys = std::min(O->cnt[0], cap);
    out->cam_first[1] = out->n_keys;
  }
  out->key_cap = kc;
  out->keys = (const vieo_keypoint*)(t->h_out + t->q_kp), out->desc = t->h_out + t->q_desc;
  out->uright = (const float*)(t->h_out + t->q_ur), out->depth = (const float*)(t->h_out + t->q_dp);
  out->point_ref = (const int32_t*)(t->h_out + t->q_mpref), out->outlier = t->h_out + t->q_outl;
  out->local_track_depth = (const float*)(t->h_out + t->q_cdep);
  out->n_matches_last = O->nm[0], out->n_matches_local = O->nm[1], out->widened = widened;
  out->nav_pred = O->nav_pred, out->imu = O->imu;
  out->first = O->r1, out->second = O->r2;
  (void)hipEventElapsedTime(&out->ms_gpu, t->ev_t0, t->ev_t1);
  out->ms_host = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_enter).count();
  // watch the chain's GPU time: eight frames in a row 30 % above the running median -> look at the second stream again
  // (at most once per 64 frames; a workload that simply grew moves the median instead)
  {
    const int n = std::min(t->gpu_n, 32);
    float med = 0;
    if (n >= 16) {
      float v[32];
      memcpy(v, t->gpu_ring, sizeof(float) * n);
      std::nth_element(v, v + n / 2, v + n);
      med = v[n / 2];
    }
    t->slow_run = (med > 0 && out->ms_gpu > 1.3f * med && !widened) ? t->slow_run + 1 : 0;
    t->gpu_ring[t->gpu_n % 32] = out->ms_gpu, t->gpu_n++;
    t->frames_since_check++;
    // (not while the next frame's extraction is still running on the prefetch stream: the probe's spin kernels would
    // share the device with it, the ratio would be skewed and the side stream swapped for nothing -- it waits for a frame
    // without a prefetch in flight)
    if (t->slow_run >= 8 && t->frames_since_check >= 64 && !t->pref_valid) {
      (void)vieo_tracker_reprobe(t);
      t->frames_since_check = 0, t->slow_run = 0, t->gpu_n = 0;
    }
  }
  return VIEO_OK;
}

}  // extern "C"
