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
#include <chrono>
#include <cmath>
#include <cstring>

#include "imu_device.h"

namespace vieo {

// upload header: everything small the chain reads, one struct so that it travels with the images in one copy
struct TrkHdr {
  vieo_sbp_camera cam;       // Tcw_cur / Tcw_last are written by k_track_predict
  vieo_vio_frame f1, f2;     // base.nav / imu written by k_track_predict, n_obs / obs_begin by k_track_build_obs
  vieo_navstate nav_ref, nav_last;
  vieo_imu_noise noise;
  double ti, tj, bg[3], ba[3];
  int32_t first[2];
  int32_t npts[4];           // [0] = n_last
  float consts[32];          // inv_sigma2[16], scale[16]
};

// download header
struct TrkOut {
  int32_t cnt[4];            // extractor counts: {n_left, mono_left, n_right, mono_right}
  int32_t nm[4];             // [0] matches of the first search, [1] of the second
  int32_t nq[4];
  int32_t preint_status[4];
  vieo_vio_result r1, r2;
  vieo_navstate nav_pred;
  vieo_imu_preint imu;
  double sigma_prv[81];
  int32_t nobs2[4];          // observations of the second optimisation
};

// PredictNavStateByIMU (Tracking.cc:385-451) from the pre-integration in HBM; fills the two optimiser problems and
// the projection search's camera.  One wavefront; lane 0 does the (double) arithmetic, all lanes copy.
__global__ void __launch_bounds__(64)
k_track_predict(TrkHdr* __restrict__ H, TrkOut* __restrict__ O, const vieo_imu_preint* __restrict__ pre,
                const double* __restrict__ sigma_prv, const int32_t* __restrict__ status) {
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

}  // namespace vieo

using namespace vieo;

struct vieo_tracker {
  vieo_tracker_params P;
  vieo_orb* ext = nullptr;
  hipStream_t st = nullptr, st_imu = nullptr;
  hipEvent_t ev_up = nullptr, ev_imu = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
  int cap = 0, ccap = 0, pcap = 0, imu_cap = 512;
  int local_version = -1, n_local_dev = 0;
  float scale[16], inv_sigma2[16];
  vieo_camera pin_cam;
  vieo_frustum_frame ff;
  // pinned blocks and their device twins (same layout)
  uint8_t *h_up = nullptr, *d_up = nullptr;      // per-frame upload
  uint8_t *h_loc = nullptr, *d_loc = nullptr;    // local-map candidates (uploaded when they change)
  uint8_t *h_out = nullptr, *d_out = nullptr;    // download
  uint8_t* d_work = nullptr;                     // device-only scratch
  // offsets in the upload block
  size_t o_hdr, o_imu, o_img, o_pts, o_xyz, o_dep, o_alias, up_fixed;
  // offsets in the local block
  size_t l_cpt, l_cdesc, l_xyz;
  // offsets in the download block
  size_t q_hdr, q_ur, q_dp, q_mpref, q_outl, q_kp, q_desc, q_cdep, out_bytes;
  // offsets in the work block
  size_t w_kp, w_desc, w_q1, w_q2, w_assign, w_taken, w_held, w_obs, w_obskey, w_outl, w_xyz, w_dep, w_pre, w_prv, w_pst;
};

static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" {

void vieo_tracker_destroy(vieo_tracker* t) {
  if (!t) return;
  if (t->st) (void)hipStreamSynchronize(t->st);
  if (t->st_imu) (void)hipStreamSynchronize(t->st_imu), (void)hipStreamDestroy(t->st_imu);
  for (hipEvent_t e : {t->ev_up, t->ev_imu, t->ev_t0, t->ev_t1})
    if (e) (void)hipEventDestroy(e);
  for (uint8_t* p : {t->h_up, t->h_loc, t->h_out})
    if (p) (void)hipHostFree(p);
  for (uint8_t* p : {t->d_up, t->d_loc, t->d_out, t->d_work})
    if (p) (void)hipFree(p);
  if (t->ext) vieo_orb_destroy(t->ext);
  delete t;
}

int vieo_tracker_create(vieo_tracker** out, const vieo_tracker_params* P) {
  if (!out || !P || P->width <= 0 || P->height <= 0 || P->n_levels < 1 || P->n_levels > 16 || P->max_local_points < 0)
    return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  vieo_tracker* t = new vieo_tracker();
  t->P = *P;
  if ((rc = vieo_orb_create(&t->ext, P->n_features, P->scale_factor, P->n_levels, P->ini_th_fast, P->min_th_fast)) != VIEO_OK) {
    delete t;
    return rc;
  }
  t->st = (hipStream_t)vieo_orb_stream(t->ext);
  t->cap = vieo_orb_max_keypoints(t->ext);
  t->ccap = std::max(P->max_local_points, 64);
  t->pcap = t->cap + t->ccap;
  vieo_orb_scale_factors(t->ext, t->scale);
  vieo_orb_inv_level_sigma2(t->ext, t->inv_sigma2);
  const size_t npx = (size_t)P->width * P->height;
  const int cap = t->cap, ccap = t->ccap;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t r = o;
    o = al256(o + bytes);
    return r;
  };
  // ---- upload block: [header | IMU samples | images | last points | their xyz | their depth | alias]
  t->o_hdr = take(sizeof(TrkHdr)), t->o_imu = take((size_t)t->imu_cap * sizeof(vieo_imu_sample)), t->o_img = take(2 * npx);
  t->o_pts = take((size_t)cap * sizeof(vieo_last_frame_point));
  t->o_xyz = take((size_t)cap * 12), t->o_dep = take((size_t)cap * 4), t->o_alias = take((size_t)ccap * 4);
  t->up_fixed = t->o_alias;
  const size_t up_bytes = o;
  o = 0;
  t->l_cpt = take((size_t)ccap * sizeof(vieo_frustum_point)), t->l_cdesc = take((size_t)ccap * 32), t->l_xyz = take((size_t)ccap * 12);
  const size_t loc_bytes = o;
  o = 0;
  t->q_hdr = take(sizeof(TrkOut));
  t->q_ur = take((size_t)cap * 4), t->q_dp = take((size_t)cap * 4), t->q_mpref = take((size_t)cap * 4), t->q_outl = take(cap);
  t->q_kp = take((size_t)cap * sizeof(vieo_keypoint)), t->q_desc = take((size_t)cap * 32), t->q_cdep = take((size_t)ccap * 4);
  t->out_bytes = o;
  o = 0;
  t->w_kp = take((size_t)2 * cap * sizeof(vieo_keypoint)), t->w_desc = take((size_t)2 * cap * 32);
  t->w_q1 = take((size_t)cap * sizeof(vieo_proj_query)), t->w_q2 = take((size_t)ccap * sizeof(vieo_proj_query));
  t->w_assign = take((size_t)cap * 4), t->w_taken = take(cap), t->w_held = take(t->pcap);
  t->w_obs = take((size_t)cap * sizeof(vieo_pose_obs)), t->w_obskey = take((size_t)cap * 4), t->w_outl = take(cap);
  t->w_xyz = take((size_t)t->pcap * 12), t->w_dep = take((size_t)t->pcap * 4);
  t->w_pre = take(sizeof(vieo_imu_preint)), t->w_prv = take(81 * 8), t->w_pst = take(16);
  const size_t work_bytes = o;
  bool ok = hipHostMalloc((void**)&t->h_up, up_bytes, hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc((void**)&t->h_loc, loc_bytes, hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc((void**)&t->h_out, t->out_bytes, hipHostMallocDefault) == hipSuccess &&
            hipMalloc((void**)&t->d_up, up_bytes) == hipSuccess && hipMalloc((void**)&t->d_loc, loc_bytes) == hipSuccess &&
            hipMalloc((void**)&t->d_out, t->out_bytes) == hipSuccess && hipMalloc((void**)&t->d_work, work_bytes) == hipSuccess &&
            hipStreamCreateWithFlags(&t->st_imu, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&t->ev_up, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&t->ev_imu, hipEventDisableTiming) == hipSuccess &&
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
  H.cam.fx = P->fx, H.cam.fy = P->fy, H.cam.cx = P->cx, H.cam.cy = P->cy;
  H.cam.bounds[0] = 0, H.cam.bounds[1] = (float)P->width, H.cam.bounds[2] = 0, H.cam.bounds[3] = (float)P->height;
  H.cam.bf = P->bf, H.cam.baseline = P->baseline, H.cam.th = P->th_last, H.cam.th_far = 0;
  H.cam.mono = 0, H.cam.nlevels = P->n_levels;
  for (int l = 0; l < P->n_levels; l++) H.cam.scale[l] = t->scale[l], H.consts[l] = t->inv_sigma2[l], H.consts[16 + l] = t->scale[l];
  for (vieo_vio_frame* f : {&H.f1, &H.f2}) {
    memcpy(f->base.Rcb, P->Rcb, 72), memcpy(f->base.tcb, P->tcb, 24);
    f->base.fx = P->fx, f->base.fy = P->fy, f->base.cx = P->cx, f->base.cy = P->cy, f->base.bf = P->bf;
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
  t->ff.bounds[0][0] = 0, t->ff.bounds[0][1] = (float)P->width, t->ff.bounds[0][2] = 0, t->ff.bounds[0][3] = (float)P->height;
  t->ff.bf = P->bf, t->ff.n_levels = P->n_levels, t->ff.viewing_cos_limit = 0.5f;
  t->ff.log_scale_factor = logf(P->scale_factor);
  *out = t;
  return VIEO_OK;
}

int vieo_tracker_image_buffers(vieo_tracker* t, uint8_t** left, uint8_t** right) {
  if (!t || !left || !right) return VIEO_E_INVALID;
  *left = t->h_up + t->o_img;
  *right = *left + (size_t)t->P.width * t->P.height;
  return VIEO_OK;
}

int vieo_tracker_scale_factors(const vieo_tracker* t, float* h_out) {
  if (!t || !h_out) return VIEO_E_INVALID;
  for (int l = 0; l < t->P.n_levels; l++) h_out[l] = t->scale[l];
  return VIEO_OK;
}

int vieo_tracker_get_level(vieo_tracker* t, int image_index, int level, int with_border, uint8_t* h_dst, int dst_stride) {
  if (!t) return VIEO_E_INVALID;
  return vieo_orb_get_level(t->ext, image_index, level, with_border, h_dst, dst_stride);
}

// the part of the chain behind the prediction: both searches and both optimisations
static int track_chain_tail(vieo_tracker* t, int nc) {
  const vieo_tracker_params& P = t->P;
  const int cap = t->cap;
  hipStream_t st = t->st;
  TrkHdr* dH = (TrkHdr*)(t->d_up + t->o_hdr);
  TrkOut* dO = (TrkOut*)(t->d_out + t->q_hdr);
  uint8_t* W = t->d_work;
  vieo_keypoint* d_kp = (vieo_keypoint*)(W + t->w_kp);
  uint8_t* d_desc = W + t->w_desc;
  float* d_ur = (float*)(t->d_out + t->q_ur);
  int32_t* d_mpref = (int32_t*)(t->d_out + t->q_mpref);
  int32_t* d_cnt = dO->cnt;
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
  const float bounds[4] = {0.f, (float)P.width, 0.f, (float)P.height};
  const float close = std::max(10.0f, P.th_depth);
  int rc;
#define TRK(call)                       \
  do {                                  \
    if ((rc = (call)) != VIEO_OK) return rc; \
  } while (0)
  TRK(vieo_sbp_project_last_frame_batch_device((const vieo_last_frame_point*)(t->d_up + t->o_pts), dH->npts, cap, 1, &dH->cam, d_q1, st));
  TRK(vieo_search_by_projection_batch_device(VIEO_SBP_LAST_FRAME, d_q1, dH->npts, cap, 1, d_kp, d_ur, d_desc, nullptr, d_cnt, cap, 0, 2,
                                             bounds, P.nn_last, 1, d_assign, dO->nm, st));
  TRK(vieo_track_merge_assign_batch_device(d_assign, d_mpref, d_cnt, cap, 1, 0, 2, 0, 1, st));
  TRK(vieo_track_build_obs_depth_batch_device(d_mpref, d_xyz, d_dep, close, t->pcap, d_kp, d_ur, d_cnt, cap, 1, 0, 2, dH->consts,
                                              d_obs, d_obskey, &dH->f1, 1, st));
  TRK(vieo_pose_optimization_vio_batch_device_ex(&dH->f1, 1, d_obs, d_outl, &dO->r1, VIEO_POSE_CAMS_RECTIFIED, VIEO_POSE_ENC_NONE, st));
  TRK(vieo_track_after_pose_batch_device(d_mpref, d_obskey, d_outl, &dH->f1, &dO->r1, 1, cap, 1, &dH->f2, d_taken, st));
  TRK(vieo_track_mark_held_batch_device(d_mpref, d_cnt, cap, 1, 0, 2, d_held, t->pcap, st));
  TRK(vieo_track_local_queries_device(&t->ff, &dH->f1, &dO->r1, (const vieo_frustum_point*)(t->d_loc + t->l_cpt), t->d_loc + t->l_cdesc,
                                      (const int32_t*)(t->d_up + t->o_alias), d_held, t->pcap, nc, P.th_local, 0.f, dH->consts + 16, d_q2,
                                      d_dep + cap, dO->nq, st));
  TRK(vieo_search_by_projection_batch_device(VIEO_SBP_LOCAL_MAP, d_q2, dO->nq, t->ccap, 1, d_kp, d_ur, d_desc, d_taken, d_cnt, cap, 0, 2,
                                             bounds, P.nn_local, 1, d_assign, dO->nm + 1, st));
  TRK(vieo_track_merge_assign_batch_device(d_assign, d_mpref, d_cnt, cap, 1, 0, 2, cap, 0, st));
  TRK(vieo_track_build_obs_depth_batch_device(d_mpref, d_xyz, d_dep, close, t->pcap, d_kp, d_ur, d_cnt, cap, 1, 0, 2, dH->consts,
                                              d_obs, d_obskey, &dH->f2, 1, st));
  TRK(vieo_pose_optimization_vio_batch_device_ex(&dH->f2, 1, d_obs, d_outl, &dO->r2, VIEO_POSE_CAMS_RECTIFIED, VIEO_POSE_ENC_NONE, st));
#undef TRK
  hipLaunchKernelGGL(k_track_finish, dim3(1), dim3(256), 0, st, d_obskey, d_outl, &dH->f2, t->d_out + t->q_outl, cap, dO);
  VIEO_HIP_CHECK(hipGetLastError());
  // results: [header | uright | depth | point_ref | outlier] and the candidates' depths (the left keys / descriptors are
  // copied by the caller, once)
  VIEO_HIP_CHECK(hipMemcpyAsync(t->h_out, t->d_out, t->q_kp, hipMemcpyDeviceToHost, st));
  if (nc > 0) VIEO_HIP_CHECK(hipMemcpyAsync(t->h_out + t->q_cdep, d_dep + cap, (size_t)nc * 4, hipMemcpyDeviceToHost, st));
  return VIEO_OK;
}

int vieo_track_frame(vieo_tracker* t, const vieo_track_input* in, vieo_track_output* out) {
  if (!t || !in || !out || !in->left || !in->right || in->stride < t->P.width || in->n_imu < 0 || (in->n_imu > 0 && !in->imu) ||
      in->n_last < 0 || (in->n_last > 0 && (!in->last_points || !in->last_track_depth)) || in->n_local < 0 ||
      (in->n_local > 0 && !in->local_alias))
    return VIEO_E_INVALID;
  const auto t_enter = std::chrono::steady_clock::now();
  const vieo_tracker_params& P = t->P;
  const int cap = t->cap, W = P.width, Hh = P.height;
  if (in->n_last > cap || in->n_local > t->ccap || in->n_imu > t->imu_cap) {
    set_error("vieo_track_frame: %d last-frame points / %d local points / %d IMU samples exceed the capacities %d / %d / %d",
              in->n_last, in->n_local, in->n_imu, cap, t->ccap, t->imu_cap);
    return VIEO_E_CAPACITY;
  }
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  hipStream_t st = t->st;
  const size_t npx = (size_t)W * Hh;
  const int nl = in->n_last, nc = in->n_local;
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
  H.npts[0] = nl;
  if (in->n_imu) memcpy(t->h_up + t->o_imu, in->imu, (size_t)in->n_imu * sizeof(vieo_imu_sample));
  uint8_t* img = t->h_up + t->o_img;
  for (int c = 0; c < 2; c++) {
    const uint8_t* src = c == 0 ? in->left : in->right;
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
  VIEO_HIP_CHECK(hipEventRecord(t->ev_t0, st));
  VIEO_HIP_CHECK(hipMemcpyAsync(t->d_up, t->h_up, t->up_fixed + (size_t)nc * 4, hipMemcpyHostToDevice, st));
  VIEO_HIP_CHECK(hipEventRecord(t->ev_up, st));
  TrkHdr* dH = (TrkHdr*)(t->d_up + t->o_hdr);
  TrkOut* dO = (TrkOut*)(t->d_out + t->q_hdr);
  // the pre-integration beside the extraction
  VIEO_HIP_CHECK(hipStreamWaitEvent(t->st_imu, t->ev_up, 0));
  if ((rc = vieo_imu_preintegrate_batch_device(&dH->noise, (const vieo_imu_sample*)(t->d_up + t->o_imu), dH->first, &dH->ti, &dH->tj,
                                               dH->bg, dH->ba, 1, (vieo_imu_preint*)(Wk + t->w_pre), (double*)(Wk + t->w_prv),
                                               (int32_t*)(Wk + t->w_pst), t->st_imu)) != VIEO_OK)
    return rc;
  VIEO_HIP_CHECK(hipEventRecord(t->ev_imu, t->st_imu));
  if (new_local) {
    VIEO_HIP_CHECK(hipMemcpyAsync(t->d_loc + t->l_cpt, t->h_loc + t->l_cpt, (size_t)nc * sizeof(vieo_frustum_point), hipMemcpyHostToDevice, st));
    VIEO_HIP_CHECK(hipMemcpyAsync(t->d_loc + t->l_cdesc, t->h_loc + t->l_cdesc, (size_t)nc * 32, hipMemcpyHostToDevice, st));
    VIEO_HIP_CHECK(hipMemcpyAsync(Wk + t->w_xyz + (size_t)cap * 12, t->h_loc + t->l_xyz, (size_t)nc * 12, hipMemcpyHostToDevice, st));
    t->local_version = in->local_version, t->n_local_dev = nc;
  }
  // the last frame's part of the two point tables
  if (nl) {
    VIEO_HIP_CHECK(hipMemcpyAsync(Wk + t->w_xyz, t->d_up + t->o_xyz, (size_t)nl * 12, hipMemcpyDeviceToDevice, st));
    VIEO_HIP_CHECK(hipMemcpyAsync(Wk + t->w_dep, t->d_up + t->o_dep, (size_t)nl * 4, hipMemcpyDeviceToDevice, st));
  }
  vieo_keypoint* d_kp = (vieo_keypoint*)(Wk + t->w_kp);
  uint8_t* d_desc = Wk + t->w_desc;
  if ((rc = vieo_orb_extract_batch_device(t->ext, t->d_up + t->o_img, 2, W, Hh, W, npx, nullptr, d_kp, d_desc, cap, dO->cnt)) != VIEO_OK)
    return rc;
  if ((rc = vieo_stereo_match_rectified_batch_device(t->ext, 1, d_kp, d_desc, dO->cnt, cap, P.baseline, P.bf, (float*)(t->d_out + t->q_ur),
                                                     (float*)(t->d_out + t->q_dp))) != VIEO_OK)
    return rc;
  VIEO_HIP_CHECK(hipStreamWaitEvent(st, t->ev_imu, 0));
  hipLaunchKernelGGL(k_track_predict, dim3(1), dim3(64), 0, st, dH, dO, (const vieo_imu_preint*)(Wk + t->w_pre),
                     (const double*)(Wk + t->w_prv), (const int32_t*)(Wk + t->w_pst));
  VIEO_HIP_CHECK(hipGetLastError());
  if ((rc = track_chain_tail(t, nc)) != VIEO_OK) return rc;
  // the left image's keys / descriptors (the extractor's arrays hold both images)
  VIEO_HIP_CHECK(hipMemcpyAsync(t->h_out + t->q_kp, d_kp, (size_t)cap * sizeof(vieo_keypoint), hipMemcpyDeviceToHost, st));
  VIEO_HIP_CHECK(hipMemcpyAsync(t->h_out + t->q_desc, d_desc, (size_t)cap * 32, hipMemcpyDeviceToHost, st));
  VIEO_HIP_CHECK(hipEventRecord(t->ev_t1, st));
  VIEO_HIP_CHECK(hipStreamSynchronize(st));
  const TrkOut* O = (const TrkOut*)(t->h_out + t->q_hdr);
  int widened = 0;
  if (O->nm[0] < 20 && O->preint_status[0] == 0 && O->imu.dt != 0) {
    // Tracking.cc:301-309: the wider window.  Only the search threshold changes; everything before the projection is
    // still in HBM
    widened = 1;
    const float th2 = 2 * P.th_last;
    VIEO_HIP_CHECK(hipMemcpyAsync(&dH->cam.th, &th2, 4, hipMemcpyHostToDevice, st));
    if ((rc = track_chain_tail(t, nc)) != VIEO_OK) return rc;
    VIEO_HIP_CHECK(hipEventRecord(t->ev_t1, st));
    VIEO_HIP_CHECK(hipStreamSynchronize(st));
  }
  memset(out, 0, sizeof(*out));
  out->preint_status = O->preint_status[0];
  out->status = (O->preint_status[0] != 0 || O->imu.dt == 0) ? VIEO_TRACK_PREINT_FAILED : VIEO_TRACK_OK;
  out->n_keys = std::min(O->cnt[0], cap), out->key_cap = cap;
  out->keys = (const vieo_keypoint*)(t->h_out + t->q_kp), out->desc = t->h_out + t->q_desc;
  out->uright = (const float*)(t->h_out + t->q_ur), out->depth = (const float*)(t->h_out + t->q_dp);
  out->point_ref = (const int32_t*)(t->h_out + t->q_mpref), out->outlier = t->h_out + t->q_outl;
  out->local_track_depth = (const float*)(t->h_out + t->q_cdep);
  out->n_matches_last = O->nm[0], out->n_matches_local = O->nm[1], out->widened = widened;
  out->nav_pred = O->nav_pred, out->imu = O->imu;
  out->first = O->r1, out->second = O->r2;
  (void)hipEventElapsedTime(&out->ms_gpu, t->ev_t0, t->ev_t1);
  out->ms_host = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_enter).count();
  return VIEO_OK;
}

}  // extern "C"
