// pose_opt.hip -- motion-only bundle adjustment on gfx950: Optimizer::PoseOptimization
// (reference: src/Optimizer.cc:1611-1874; edges src/Odom/g2otypes.h:321-547; g2o LM
// optimization_algorithm_levenberg.cpp:61-207).
//
// One persistent workgroup per frame runs the WHOLE reference procedure on the device: 4 rounds x
// optimize(10) Levenberg-Marquardt iterations with up to 10 lambda trials each, chi2
// re-classification after every round, no host round trip.  A frame has a few hundred edges and a
// 6x6 system, so the work is latency bound; throughput comes from running one workgroup per frame
// of a batch.  Per pass every thread evaluates its edges (projection rounded to float exactly as
// the reference does), per-wave shuffle reductions + one LDS exchange give every thread the
// identical 27 doubles of J^T W J / J^T W e, and every thread then solves the 6x6 system
// redundantly (uniform control flow, no broadcast needed).
// FP64 throughout; parity with oracle/pose_opt.cc is <= 1e-4 on SE(3) (summation order differs).
#include "imu_device.h"

namespace vieo {

// 6x6 LDL^T solve (LinearSolverDense, Eigen LDLT); false if a pivot is not positive
__device__ __forceinline__ bool ldlt6(const double* H, const double* b, double* x) {
  double L[36], D[6], y[6];
  for (int j = 0; j < 6; j++) {
    double d = H[j * 6 + j];
    for (int k = 0; k < j; k++) d -= L[j * 6 + k] * L[j * 6 + k] * D[k];
    if (!(d > 0)) return false;
    D[j] = d;
    for (int i = j + 1; i < 6; i++) {
      double s = H[i * 6 + j];
      for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k] * D[k];
      L[i * 6 + j] = s / d;
    }
  }
  for (int i = 0; i < 6; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= L[i * 6 + k] * y[k];
    y[i] = s;
  }
  for (int i = 0; i < 6; i++) y[i] /= D[i];
  for (int i = 5; i >= 0; i--) {
    double s = y[i];
    for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * x[k];
    x[i] = s;
  }
  return true;
}

// ---- the optional encoder edge (EdgeEncNavStatePR between the fixed last frame and the frame, Optimizer.cc:1650-1674,
// Huber sqrt(12.592) in all four rounds).  One 6-row edge: thread 0 evaluates it out of line (the visual loops keep
// their registers) and publishes chi2 and its J^T (rho' Omega) J, J^T (-rho' Omega e) through LDS.
struct PoseEncShared {
  double Info[36], H[36], b[6], chi;
};

__device__ __noinline__ void pose_enc_setup(const vieo_pose_enc* pe, PoseEncShared* S) {
  double M[6][12];  // Gauss-Jordan with partial pivoting, as the host does for the BA edges
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) M[i][j] = pe->enc.Sigma[i * 6 + j], M[i][6 + j] = (i == j) ? 1.0 : 0.0;
  for (int c = 0; c < 6; c++) {
    int piv = c;
    for (int r = c + 1; r < 6; r++)
      if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
    if (piv != c)
      for (int j = 0; j < 12; j++) {
        const double t = M[c][j];
        M[c][j] = M[piv][j], M[piv][j] = t;
      }
    const double d = M[c][c];
    for (int j = 0; j < 12; j++) M[c][j] /= d;
    for (int r = 0; r < 6; r++)
      if (r != c) {
        const double f = M[r][c];
        if (f != 0)
          for (int j = 0; j < 12; j++) M[r][j] -= f * M[c][j];
      }
  }
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) S->Info[i * 6 + j] = M[i][6 + j];
}

__device__ __noinline__ void pose_enc_eval(const vieo_pose_enc* pe, PoseEncShared* S, const Est* est, int jac) {
  NSd si, sj;
  for (int k = 0; k < 3; k++) si.p[k] = pe->p_last[k], sj.p[k] = est->p[k];
  si.qw = pe->q_last[0], si.qx = pe->q_last[1], si.qy = pe->q_last[2], si.qz = pe->q_last[3];
  sj.qw = est->qw, sj.qx = est->qx, sj.qy = est->qy, sj.qz = est->qz;
  double err[6], Ji[36], Jj[36];
  enc_edge_eval(si, sj, pe->enc.delx, pe->qRbe, pe->pbe, err, jac ? Ji : nullptr, jac ? Jj : nullptr);
  double we[6], c2 = 0;
  for (int a = 0; a < 6; a++) {
    double t = 0;
    for (int q = 0; q < 6; q++) t += S->Info[a * 6 + q] * err[q];
    we[a] = t;
    c2 += err[a] * t;
  }
  const double dE = (double)(float)sqrt(12.592);
  double r0, r1;
  huber(c2, dE, dE * dE, &r0, &r1);
  S->chi = r0;
  if (!jac) return;
  double T[36];
  for (int a = 0; a < 6; a++)
    for (int c = 0; c < 6; c++) {
      double u = 0;
      for (int q = 0; q < 6; q++) u += (r1 * S->Info[a * 6 + q]) * Jj[q * 6 + c];
      T[a * 6 + c] = u;
    }
  for (int i = 0; i < 6; i++) {
    for (int j = 0; j < 6; j++) {
      double u = 0;
      for (int a = 0; a < 6; a++) u += Jj[a * 6 + i] * T[a * 6 + j];
      S->H[i * 6 + j] = u;
    }
    double u = 0;
    for (int a = 0; a < 6; a++) u += Jj[a * 6 + i] * (-we[a] * r1);
    S->b[i] = u;
  }
}

// observations per frame: one bit per edge in a 64-bit mask per lane -> 64 x threads (4096 / 16384)
#define kMaxObs (64 * BS)

// BS threads per frame: 256 for a few frames (lowest latency), 64 = one wavefront per frame for large
// batches (four frames per CU in flight), as in pose_opt_vio.hip
// MC = true handles the frames of a distorted multi-camera rig (n_cams > 0, a20), MC = false all others;
// each instance leaves the other's frames alone, so a mixed batch takes both launches.
template <int BS, bool MC>
__global__ void __launch_bounds__(BS)
k_pose_opt(const vieo_pose_frame* __restrict__ frames, const vieo_pose_obs* __restrict__ obs_all,
           uint8_t* __restrict__ outlier_all, vieo_pose_result* __restrict__ results, int other_launched) {
  __shared__ double s_red[4 * 27];
  __shared__ double s_tr[28 * (BS / 32) * 34], s_vis[28];  // block_sum_lds' transposition buffer, the 28 sums of an iteration
  __shared__ __align__(8) unsigned char s_cam_store[sizeof(CamD) * (MC ? 4 : 1)];  // CamD has initialisers
  CamD* s_cams = reinterpret_cast<CamD*>(s_cam_store);
  __shared__ int s_bad;
  __shared__ PoseEncShared s_enc;
  const int f = blockIdx.x, tid = threadIdx.x;
  const vieo_pose_frame& F = frames[f];
  const int N = F.n_obs;
  const vieo_pose_obs* obs = obs_all + F.obs_begin;
  uint8_t* outl = outlier_all + F.obs_begin;
  vieo_pose_result* R = results + f;
  if ((F.n_cams > 0) != MC) {
    if (!other_launched) {  // vieo_pose_set_camera_mode promised frames of the other kind only
      for (int i = tid; i < N; i += BS) outl[i] = 0;
      if (tid == 0) {
        R->nav = F.nav;
        R->n_inliers = 0, R->status = VIEO_E_INVALID, R->lm_iterations = 0, R->reserved = 0;
      }
    }
    return;
  }
  if (N < 3 || N > kMaxObs) {  // Optimizer.cc:1789
    for (int i = tid; i < N; i += BS) outl[i] = 0;
    if (tid == 0) {
      R->nav = F.nav;
      R->n_inliers = 0;
      R->status = N < 3 ? VIEO_POSE_TOO_FEW : VIEO_E_CAPACITY;
      R->lm_iterations = 0;
      R->reserved = 0;
    }
    return;
  }
  CamD c;
  c.fx = F.fx, c.fy = F.fy, c.cx = F.cx, c.cy = F.cy, c.bf = F.bf;
  for (int i = 0; i < 9; i++) c.Rcb[i] = F.Rcb[i];
  for (int i = 0; i < 3; i++) c.tcb[i] = F.tcb[i];
  if (MC) {
    if (tid == 0) s_bad = F.n_cams > 4 || !F.cams;
    __syncthreads();
    if (tid < 4 && !s_bad) {
      const bool ok = cam_from_abi(F.cams[tid < F.n_cams ? tid : 0], s_cams[tid]);
      if (!ok) s_bad = 1;
    }
    __syncthreads();
    if (s_bad) {
      for (int i = tid; i < N; i += BS) outl[i] = 0;
      if (tid == 0) {
        R->nav = F.nav;
        R->n_inliers = 0, R->status = VIEO_E_INVALID, R->lm_iterations = 0, R->reserved = 0;
      }
      return;
    }
  }
  const vieo_pose_enc* pe = F.enc;
  const bool has_enc = pe != nullptr && pe->enc.dt != 0;
  if (has_enc) {
    if (tid == 0) pose_enc_setup(pe, &s_enc);
    __syncthreads();
  }
  // `const float deltaMono = sqrt(5.991)`: double sqrt rounded to float (Optimizer.cc:1689-1690)
  const double deltaMono = (double)(float)sqrt(5.991), deltaStereo = (double)(float)sqrt(7.815);
  const float chi2Mono = 5.991f, chi2Stereo = 7.815f;
  Est init;
  init.p[0] = F.nav.p[0], init.p[1] = F.nav.p[1], init.p[2] = F.nav.p[2];
  init.qw = F.nav.q[0], init.qx = F.nav.q[1], init.qy = F.nav.q[2], init.qz = F.nav.q[3];
  Est est = init;
  unsigned long long levelmask = 0;  // bit k: edge tid + 256k is at level 1 (outlier)
  int nBad = 0, total_iters = 0;
  for (int it = 0; it < 4; it++) {
    est = init;
    const bool robust = it < 3;
    // active-edge count decides whether optimize() does anything
    double cnt[1] = {0};
    for (int k = 0, i = tid; i < N; k++, i += BS) cnt[0] += ((levelmask >> k) & 1) ? 0. : 1.;
    block_sum_bs<1, BS>(cnt, s_red, tid);
    Est est_err = est;  // estimate at the last computeActiveErrors (g2o does not pop edge errors)
    if (cnt[0] > 0 || has_enc) {
      double lambda = -1, ni = 2;
      int nBadLM = 0;
      for (int iter = 0; iter < 10; iter++) {
        total_iters++;
        PoseXf X;
        make_xf(c, est, X);
        // computeActiveErrors + activeRobustChi2 + buildSystem in one pass
        double acc[28];  // the system's 21 + 6 sums, the robust chi2
#pragma unroll
        for (int i = 0; i < 28; i++) acc[i] = 0;
        vieo_pose_obs o_next = obs[min(tid, N - 1)];  // the next edge's record is in flight while this one is evaluated
        for (int k = 0, i = tid; i < N; k++, i += BS) {
          const vieo_pose_obs o = o_next;
          if (i + BS < N) o_next = obs[i + BS];
          if ((levelmask >> k) & 1) continue;
          double err[3], Pc[3];
          double J[18];
          const double chi2 = edge_eval<MC>(c, s_cams, X, est.p, o, err, Pc, J);
          const bool stereo = o.ur >= 0;
          double r0 = chi2, r1 = 1.;
          if (robust) {
            const double dl = stereo ? deltaStereo : deltaMono;
            huber(chi2, dl, dl * dl, &r0, &r1);
          }
          acc[27] += r0;
          visual_accumulate(J, err, (double)o.inv_sigma2, r1, stereo, acc);
        }
        est_err = est;
        // one LDS transpose for the 28 sums (28 butterfly sums were ~500 instructions per wavefront: ba_device.h);
        // every thread then reads them back, the 6 x 6 solve below is done by all of them side by side
        block_sum_lds<28, BS>(acc, s_tr, s_vis, tid);
        double currentChi = s_vis[27];
        double H[36], b[6];
        {
          int t = 0;
          for (int a = 0; a < 6; a++)
            for (int bb = a; bb < 6; bb++, t++) H[a * 6 + bb] = H[bb * 6 + a] = s_vis[t];
          for (int a = 0; a < 6; a++) b[a] = s_vis[21 + a];
        }
        if (has_enc) {
          __syncthreads();
          if (tid == 0) pose_enc_eval(pe, &s_enc, &est, 1);
          __syncthreads();
          currentChi += s_enc.chi;
          for (int a = 0; a < 36; a++) H[a] += s_enc.H[a];
          for (int a = 0; a < 6; a++) b[a] += s_enc.b[a];
        }
        const double iniChi = currentChi;
        if (iter == 0) {  // computeLambdaInit
          double mx = 0;
          for (int j = 0; j < 6; j++) mx = fmax(fabs(H[j * 6 + j]), mx);
          lambda = 1e-5 * mx;
          ni = 2;
          nBadLM = 0;
        }
        double rho = 0;
        int qmax = 0;
        do {
          const Est backup = est;
          double Hl[36], x[6] = {0, 0, 0, 0, 0, 0};
          for (int i = 0; i < 36; i++) Hl[i] = H[i];
          for (int j = 0; j < 6; j++) Hl[j * 6 + j] += lambda;
          const bool ok2 = ldlt6(Hl, b, x);
          inc_small_pr(est, x);
          PoseXf X2;
          make_xf(c, est, X2);
          double tc[1] = {0};
          vieo_pose_obs o_next = obs[min(tid, N - 1)];  // the next edge's record is in flight while this one is evaluated
          for (int k = 0, i = tid; i < N; k++, i += BS) {
            const vieo_pose_obs o = o_next;
            if (i + BS < N) o_next = obs[i + BS];
            if ((levelmask >> k) & 1) continue;
            double err[3], Pc[3];
            const double chi2 = edge_eval<MC>(c, s_cams, X2, est.p, o, err, Pc, nullptr);
            double r0 = chi2, r1 = 1.;
            if (robust) {
              const double dl = o.ur >= 0 ? deltaStereo : deltaMono;
              huber(chi2, dl, dl * dl, &r0, &r1);
            }
            tc[0] += r0;
          }
          est_err = est;
          block_sum_bs<1, BS>(tc, s_red, tid);
          double tempChi = tc[0];
          if (has_enc) {
            __syncthreads();
            if (tid == 0) pose_enc_eval(pe, &s_enc, &est, 0);
            __syncthreads();
            tempChi += s_enc.chi;
          }
          if (!ok2) tempChi = DBL_MAX;
          rho = currentChi - tempChi;
          double scale = 0;
          for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
          scale += 1e-3;
          rho /= scale;
          if (rho > 0 && isfinite(tempChi)) {
            double alpha = 1. - pow(2 * rho - 1, 3);
            alpha = fmin(alpha, 2. / 3.);
            const double sf = fmax(1. / 3., alpha);
            lambda *= sf;
            ni = 2;
            currentChi = tempChi;
          } else {
            lambda *= ni;
            ni *= 2;
            est = backup;
          }
          qmax++;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0) break;  // Terminate
        if ((iniChi - currentChi) * 1e3 < iniChi)
          nBadLM++;
        else
          nBadLM = 0;
        if (nBadLM >= 3) break;
      }
    }
    // ---- classification (Optimizer.cc:1810-1861)
    PoseXf Xe, Xc;
    make_xf(c, est_err, Xe);  // inliers keep the error of the last computeActiveErrors
    make_xf(c, est, Xc);      // outliers are re-evaluated at the current estimate
    double nb[1] = {0};
    for (int k = 0, i = tid; i < N; k++, i += BS) {
      const vieo_pose_obs o = obs[i];
      const bool was_out = (levelmask >> k) & 1;
      double err[3], Pc[3];
      const float chi2 = (float)edge_eval<MC>(c, s_cams, was_out ? Xc : Xe, was_out ? est.p : est_err.p, o, err, Pc, nullptr);
      const float th = o.ur >= 0 ? chi2Stereo : chi2Mono;
      if (chi2 > th) {
        levelmask |= (1ull << k);
        nb[0] += 1;
      } else
        levelmask &= ~(1ull << k);
    }
    block_sum_bs<1, BS>(nb, s_red, tid);
    nBad = (int)nb[0];
    if (N + (has_enc ? 1 : 0) < 10) break;  // optimizer.edges().size() < 10
  }
  for (int k = 0, i = tid; i < N; k++, i += BS) outl[i] = (levelmask >> k) & 1;
  if (tid == 0) {
    R->nav = F.nav;
    R->nav.p[0] = est.p[0], R->nav.p[1] = est.p[1], R->nav.p[2] = est.p[2];
    R->nav.q[0] = est.qw, R->nav.q[1] = est.qx, R->nav.q[2] = est.qy, R->nav.q[3] = est.qz;
    R->n_inliers = N - nBad;
    R->status = VIEO_POSE_OK;
    R->lm_iterations = total_iters;
    R->reserved = 0;
  }
}

}  // namespace vieo

using namespace vieo;

extern "C" {

static int pose_launch(const vieo_pose_frame* d_frames, int n_frames, const vieo_pose_obs* d_obs,
                       uint8_t* d_outlier, vieo_pose_result* d_results, int which, void* stream) {
  if (!d_frames || n_frames <= 0 || !d_obs || !d_outlier || !d_results) return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  static const int forced = [] {  // VIEO_POSE_THREADS=64|256 overrides the choice (tuning / tests)
    const char* e = getenv("VIEO_POSE_THREADS");
    return e ? atoi(e) : 0;
  }();
  const bool narrow = forced == 64 || (forced != 256 && n_frames > 256);
  const int both = which == 3;
  if (which & 1) {
    if (narrow)
      hipLaunchKernelGGL((k_pose_opt<64, false>), dim3(n_frames), dim3(64), 0, (hipStream_t)stream, d_frames, d_obs,
                         d_outlier, d_results, both);
    else
      hipLaunchKernelGGL((k_pose_opt<256, false>), dim3(n_frames), dim3(256), 0, (hipStream_t)stream, d_frames,
                         d_obs, d_outlier, d_results, both);
  }
  if (which & 2) {
    if (false)  // rig frames carry n_cams x the observations: always the wide form (up to 16384 edges)
      hipLaunchKernelGGL((k_pose_opt<64, true>), dim3(n_frames), dim3(64), 0, (hipStream_t)stream, d_frames, d_obs,
                         d_outlier, d_results, both);
    else
      hipLaunchKernelGGL((k_pose_opt<256, true>), dim3(n_frames), dim3(256), 0, (hipStream_t)stream, d_frames,
                         d_obs, d_outlier, d_results, both);
  }
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

int vieo_pose_optimization_batch_device(const vieo_pose_frame* d_frames, int n_frames,
                                        const vieo_pose_obs* d_obs, uint8_t* d_outlier,
                                        vieo_pose_result* d_results, void* stream) {
  return pose_launch(d_frames, n_frames, d_obs, d_outlier, d_results, vieo::pose_rig_launches(), stream);
}

int vieo_pose_optimization_batch_device_ex(const vieo_pose_frame* d_frames, int n_frames, const vieo_pose_obs* d_obs,
                                           uint8_t* d_outlier, vieo_pose_result* d_results, int cams_mode, void* stream) {
  if (cams_mode < VIEO_POSE_CAMS_AUTO || cams_mode > VIEO_POSE_CAMS_RIG) return VIEO_E_INVALID;
  return pose_launch(d_frames, n_frames, d_obs, d_outlier, d_results, vieo::pose_launch_mask(cams_mode), stream);
}

int vieo_pose_optimization(const vieo_pose_frame* h_frame, const vieo_pose_obs* h_obs,
                           uint8_t* h_outlier, vieo_pose_result* h_result) {
  if (!h_frame || !h_result || (h_frame->n_obs > 0 && (!h_obs || !h_outlier))) return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  static thread_local DevBuf dF, dO, dU, dR, dC;
  const int n = h_frame->n_obs, nc = h_frame->n_cams;
  if (nc < 0 || nc > 4 || (nc > 0 && !h_frame->cams)) {
    set_error("PoseOptimization: n_cams = %d (0..4) needs `cams`", nc);
    return VIEO_E_INVALID;
  }
  if ((rc = dF.ensure(sizeof(vieo_pose_frame))) != VIEO_OK) return rc;
  if ((rc = dC.ensure(4 * sizeof(vieo_camera))) != VIEO_OK) return rc;
  if ((rc = dO.ensure((size_t)std::max(n, 1) * sizeof(vieo_pose_obs))) != VIEO_OK) return rc;
  if ((rc = dU.ensure(std::max(n, 1))) != VIEO_OK) return rc;
  if ((rc = dR.ensure(sizeof(vieo_pose_result))) != VIEO_OK) return rc;
  vieo_pose_frame F = *h_frame;
  const vieo_pose_obs* src = h_obs + h_frame->obs_begin;
  F.obs_begin = 0;
  if (nc > 0) {
    VIEO_HIP_CHECK(hipMemcpy(dC.p, h_frame->cams, (size_t)nc * sizeof(vieo_camera), hipMemcpyHostToDevice));
    F.cams = dC.as<vieo_camera>();
  }
  if (h_frame->enc) {
    static thread_local DevBuf dE;
    if ((rc = dE.ensure(sizeof(vieo_pose_enc))) != VIEO_OK) return rc;
    VIEO_HIP_CHECK(hipMemcpy(dE.p, h_frame->enc, sizeof(vieo_pose_enc), hipMemcpyHostToDevice));
    F.enc = dE.as<vieo_pose_enc>();
  }
  VIEO_HIP_CHECK(hipMemcpy(dF.p, &F, sizeof(F), hipMemcpyHostToDevice));
  if (n > 0) VIEO_HIP_CHECK(hipMemcpy(dO.p, src, (size_t)n * sizeof(vieo_pose_obs), hipMemcpyHostToDevice));
  rc = pose_launch(dF.as<vieo_pose_frame>(), 1, dO.as<vieo_pose_obs>(), dU.as<uint8_t>(),
                   dR.as<vieo_pose_result>(), nc > 0 ? 2 : 1, nullptr);
  if (rc != VIEO_OK) return rc;
  VIEO_HIP_CHECK(hipMemcpy(h_result, dR.p, sizeof(vieo_pose_result), hipMemcpyDeviceToHost));
  if (n > 0) VIEO_HIP_CHECK(hipMemcpy(h_outlier + h_frame->obs_begin, dU.p, n, hipMemcpyDeviceToHost));
  if (h_result->status == VIEO_E_CAPACITY) {
    set_error("PoseOptimization: %d observations exceed the kernel's capacity (16384)", n);
    return VIEO_E_CAPACITY;
  }
  return VIEO_OK;
}

}  // extern "C"
