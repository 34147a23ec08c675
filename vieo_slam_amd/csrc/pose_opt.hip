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
#include <cfloat>

#include "common.h"

namespace vieo {

struct Est {
  double p[3];
  double qw, qx, qy, qz;
};

struct CamD {
  double fx, fy, cx, cy, bf;  // float parameters widened once
  double Rcb[9], tcb[3];
};

__device__ __forceinline__ void quat_to_R(const Est& s, double* R) {
  const double tx = 2 * s.qx, ty = 2 * s.qy, tz = 2 * s.qz;
  const double twx = tx * s.qw, twy = ty * s.qw, twz = tz * s.qw;
  const double txx = tx * s.qx, txy = ty * s.qx, txz = tz * s.qx;
  const double tyy = ty * s.qy, tyz = tz * s.qy, tzz = tz * s.qz;
  R[0] = 1 - (tyy + tzz), R[1] = txy - twz, R[2] = txz + twy;
  R[3] = txy + twz, R[4] = 1 - (txx + tzz), R[5] = tyz - twx;
  R[6] = txz - twy, R[7] = tyz + twx, R[8] = 1 - (txx + tyy);
}

// NavState::IncSmall (NavState.h:47-58): p += Rwb*dp ; Rwb *= Exp(dphi), SO3ex::exp
// (so3_extra.h:121-142) with its 1e-5 small-angle branch.
__device__ __forceinline__ void inc_small_pr(Est& s, const double* d) {
  double R[9];
  quat_to_R(s, R);
  for (int i = 0; i < 3; i++) s.p[i] += R[i * 3] * d[0] + R[i * 3 + 1] * d[1] + R[i * 3 + 2] * d[2];
  const double theta = sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
  double imag, real;
  if (theta < 1e-5) {
    const double t2 = theta * theta;
    imag = 0.5 - t2 / 48.;
    real = 1.0 - t2 / 8.;
  } else {
    const double half = 0.5 * theta;
    imag = sin(half) / theta;
    real = cos(half);
  }
  double ew = real, ex = imag * d[3], ey = imag * d[4], ez = imag * d[5];
  double n = sqrt(ew * ew + ex * ex + ey * ey + ez * ez);
  ew /= n, ex /= n, ey /= n, ez /= n;
  const double w = s.qw * ew - s.qx * ex - s.qy * ey - s.qz * ez;
  const double x = s.qw * ex + s.qx * ew + s.qy * ez - s.qz * ey;
  const double y = s.qw * ey + s.qy * ew + s.qz * ex - s.qx * ez;
  const double z = s.qw * ez + s.qz * ew + s.qx * ey - s.qy * ex;
  n = sqrt(w * w + x * x + y * y + z * z);
  s.qw = w / n, s.qx = x / n, s.qy = y / n, s.qz = z / n;
}

struct PoseXf {  // per-estimate transforms shared by all edges
  double Rcw[9], tcw[3], Rwb[9];
};

__device__ __forceinline__ void make_xf(const CamD& c, const Est& s, PoseXf& X) {
  quat_to_R(s, X.Rwb);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)  // Rcw = Rcb * Rwb^T
      X.Rcw[i * 3 + j] = c.Rcb[i * 3] * X.Rwb[j * 3] + c.Rcb[i * 3 + 1] * X.Rwb[j * 3 + 1] +
                         c.Rcb[i * 3 + 2] * X.Rwb[j * 3 + 2];
  for (int i = 0; i < 3; i++)
    X.tcw[i] = -(X.Rcw[i * 3] * s.p[0] + X.Rcw[i * 3 + 1] * s.p[1] + X.Rcw[i * 3 + 2] * s.p[2]) + c.tcb[i];
}

// EdgeReproject::computeError (g2otypes.h:400-406) with PinholeCamera::Project's float rounding
// (camera_pinhole.h:70-84).  Returns chi2 = e . (info * e).
__device__ __forceinline__ double edge_error(const CamD& c, const PoseXf& X, const vieo_pose_obs& o,
                                             double* err, double* Pc) {
  const double Xw0 = o.Xw[0], Xw1 = o.Xw[1], Xw2 = o.Xw[2];
  for (int i = 0; i < 3; i++)
    Pc[i] = X.Rcw[i * 3] * Xw0 + X.Rcw[i * 3 + 1] * Xw1 + X.Rcw[i * 3 + 2] * Xw2 + X.tcw[i];
  const double invz = 1. / Pc[2];
  const double u = (double)(float)(c.fx * Pc[0] * invz + c.cx);
  const double v = (double)(float)(c.fy * Pc[1] * invz + c.cy);
  err[0] = (double)o.u - u;
  err[1] = (double)o.v - v;
  const double info = (double)o.inv_sigma2;
  double chi2 = err[0] * (info * err[0]) + err[1] * (info * err[1]);
  if (o.ur >= 0) {
    err[2] = (double)o.ur - (u - c.bf / Pc[2]);
    chi2 += err[2] * (info * err[2]);
  } else
    err[2] = 0;
  return chi2;
}

// RobustKernelHuber::robustify (robust_kernel_impl.cpp:78-91): rho[0], rho[1]
__device__ __forceinline__ void huber(double e, double delta, double dsqr, double* r0, double* r1) {
  if (e <= dsqr) {
    *r0 = e, *r1 = 1.;
  } else {
    const double sq = sqrt(e);
    *r0 = 2 * sq * delta - dsqr;
    *r1 = delta / sq;
  }
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// all threads receive the identical block-wide sums of vals[0..n)
template <int N>
__device__ __forceinline__ void block_sum(double* vals, double* s_red, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int i = 0; i < N; i++) vals[i] = wave_sum_d(vals[i]);
  __syncthreads();  // previous readers of s_red are done
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < N; i++) s_red[wave * N + i] = vals[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; i++) vals[i] = (s_red[i] + s_red[N + i]) + (s_red[2 * N + i] + s_red[3 * N + i]);
}

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

static const int kMaxEPT = 8;  // edges per thread: n_obs <= 2048

__global__ void __launch_bounds__(256)
k_pose_opt(const vieo_pose_frame* __restrict__ frames, const vieo_pose_obs* __restrict__ obs_all,
           uint8_t* __restrict__ outlier_all, vieo_pose_result* __restrict__ results) {
  __shared__ double s_red[4 * 27];
  const int f = blockIdx.x, tid = threadIdx.x;
  const vieo_pose_frame& F = frames[f];
  const int N = F.n_obs;
  const vieo_pose_obs* obs = obs_all + F.obs_begin;
  uint8_t* outl = outlier_all + F.obs_begin;
  vieo_pose_result* R = results + f;
  if (N < 3 || N > 256 * kMaxEPT) {  // Optimizer.cc:1789
    for (int i = tid; i < N; i += 256) outl[i] = 0;
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
  // `const float deltaMono = sqrt(5.991)`: double sqrt rounded to float (Optimizer.cc:1689-1690)
  const double deltaMono = (double)(float)sqrt(5.991), deltaStereo = (double)(float)sqrt(7.815);
  const float chi2Mono = 5.991f, chi2Stereo = 7.815f;
  Est init;
  init.p[0] = F.nav.p[0], init.p[1] = F.nav.p[1], init.p[2] = F.nav.p[2];
  init.qw = F.nav.q[0], init.qx = F.nav.q[1], init.qy = F.nav.q[2], init.qz = F.nav.q[3];
  Est est = init;
  unsigned levelmask = 0;  // bit k: edge tid + 256k is at level 1 (outlier)
  int nBad = 0, total_iters = 0;
  for (int it = 0; it < 4; it++) {
    est = init;
    const bool robust = it < 3;
    // active-edge count decides whether optimize() does anything
    double cnt[1] = {0};
    for (int k = 0, i = tid; i < N; k++, i += 256) cnt[0] += ((levelmask >> k) & 1) ? 0. : 1.;
    block_sum<1>(cnt, s_red, tid);
    Est est_err = est;  // estimate at the last computeActiveErrors (g2o does not pop edge errors)
    if (cnt[0] > 0) {
      double lambda = -1, ni = 2;
      int nBadLM = 0;
      for (int iter = 0; iter < 10; iter++) {
        total_iters++;
        PoseXf X;
        make_xf(c, est, X);
        // computeActiveErrors + activeRobustChi2 + buildSystem in one pass
        double acc[27];
#pragma unroll
        for (int i = 0; i < 27; i++) acc[i] = 0;
        double chi = 0;
        for (int k = 0, i = tid; i < N; k++, i += 256) {
          if ((levelmask >> k) & 1) continue;
          const vieo_pose_obs o = obs[i];
          double err[3], Pc[3];
          const double chi2 = edge_error(c, X, o, err, Pc);
          const bool stereo = o.ur >= 0;
          double r0 = chi2, r1 = 1.;
          if (robust) {
            const double dl = stereo ? deltaStereo : deltaMono;
            huber(chi2, dl, dl * dl, &r0, &r1);
          }
          chi += r0;
          // linearizeOplus (g2otypes.h:439-498)
          const double invz = 1 / Pc[2], invz2 = invz * invz;
          double Jp[9];
          Jp[0] = -(c.fx * invz), Jp[1] = 0, Jp[2] = -(-c.fx * Pc[0] * invz2);
          Jp[3] = 0, Jp[4] = -(c.fy * invz), Jp[5] = -(-c.fy * Pc[1] * invz2);
          Jp[6] = Jp[0], Jp[7] = Jp[1], Jp[8] = Jp[2] - c.bf * invz2;
          const double dP0 = (double)o.Xw[0] - est.p[0], dP1 = (double)o.Xw[1] - est.p[1],
                       dP2 = (double)o.Xw[2] - est.p[2];
          double Pa[3];
          for (int m = 0; m < 3; m++) Pa[m] = X.Rwb[m] * dP0 + X.Rwb[3 + m] * dP1 + X.Rwb[6 + m] * dP2;
          // RcbH = Rcb * hat(Paux)
          double RH[9];
          for (int m = 0; m < 3; m++) {
            const double a = c.Rcb[m * 3], b = c.Rcb[m * 3 + 1], d = c.Rcb[m * 3 + 2];
            RH[m * 3 + 0] = b * Pa[2] - d * Pa[1];
            RH[m * 3 + 1] = -a * Pa[2] + d * Pa[0];
            RH[m * 3 + 2] = a * Pa[1] - b * Pa[0];
          }
          const int de = stereo ? 3 : 2;
          double J[18];
          for (int r = 0; r < 3; r++)
            for (int q = 0; q < 3; q++) {
              J[r * 6 + q] = -(Jp[r * 3] * c.Rcb[q] + Jp[r * 3 + 1] * c.Rcb[3 + q] + Jp[r * 3 + 2] * c.Rcb[6 + q]);
              J[r * 6 + 3 + q] = Jp[r * 3] * RH[q] + Jp[r * 3 + 1] * RH[3 + q] + Jp[r * 3 + 2] * RH[6 + q];
            }
          const double info = (double)o.inv_sigma2;
          const double w = r1 * info;
          int t = 0;
          for (int a = 0; a < 6; a++) {
            for (int b = a; b < 6; b++, t++) {
              double s = J[a] * w * J[b] + J[6 + a] * w * J[6 + b];
              if (de == 3) s += J[12 + a] * w * J[12 + b];
              acc[t] += s;
            }
            double s = J[a] * (-(info * err[0]) * r1) + J[6 + a] * (-(info * err[1]) * r1);
            if (de == 3) s += J[12 + a] * (-(info * err[2]) * r1);
            acc[21 + a] += s;
          }
        }
        est_err = est;
        block_sum<27>(acc, s_red, tid);
        double c1[1] = {chi};
        block_sum<1>(c1, s_red, tid);
        double currentChi = c1[0];
        const double iniChi = currentChi;
        double H[36], b[6];
        {
          int t = 0;
          for (int a = 0; a < 6; a++)
            for (int bb = a; bb < 6; bb++, t++) H[a * 6 + bb] = H[bb * 6 + a] = acc[t];
          for (int a = 0; a < 6; a++) b[a] = acc[21 + a];
        }
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
          for (int k = 0, i = tid; i < N; k++, i += 256) {
            if ((levelmask >> k) & 1) continue;
            const vieo_pose_obs o = obs[i];
            double err[3], Pc[3];
            const double chi2 = edge_error(c, X2, o, err, Pc);
            double r0 = chi2, r1 = 1.;
            if (robust) {
              const double dl = o.ur >= 0 ? deltaStereo : deltaMono;
              huber(chi2, dl, dl * dl, &r0, &r1);
            }
            tc[0] += r0;
          }
          est_err = est;
          block_sum<1>(tc, s_red, tid);
          double tempChi = tc[0];
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
    for (int k = 0, i = tid; i < N; k++, i += 256) {
      const vieo_pose_obs o = obs[i];
      const bool was_out = (levelmask >> k) & 1;
      double err[3], Pc[3];
      const float chi2 = (float)edge_error(c, was_out ? Xc : Xe, o, err, Pc);
      const float th = o.ur >= 0 ? chi2Stereo : chi2Mono;
      if (chi2 > th) {
        levelmask |= (1u << k);
        nb[0] += 1;
      } else
        levelmask &= ~(1u << k);
    }
    block_sum<1>(nb, s_red, tid);
    nBad = (int)nb[0];
    if (N < 10) break;  // optimizer.edges().size() < 10
  }
  for (int k = 0, i = tid; i < N; k++, i += 256) outl[i] = (levelmask >> k) & 1;
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

int vieo_pose_optimization_batch_device(const vieo_pose_frame* d_frames, int n_frames,
                                        const vieo_pose_obs* d_obs, uint8_t* d_outlier,
                                        vieo_pose_result* d_results, void* stream) {
  if (!d_frames || n_frames <= 0 || !d_obs || !d_outlier || !d_results) return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  hipLaunchKernelGGL(k_pose_opt, dim3(n_frames), dim3(256), 0, (hipStream_t)stream, d_frames, d_obs,
                     d_outlier, d_results);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

int vieo_pose_optimization(const vieo_pose_frame* h_frame, const vieo_pose_obs* h_obs,
                           uint8_t* h_outlier, vieo_pose_result* h_result) {
  if (!h_frame || !h_result || (h_frame->n_obs > 0 && (!h_obs || !h_outlier))) return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  static thread_local DevBuf dF, dO, dU, dR;
  const int n = h_frame->n_obs;
  if ((rc = dF.ensure(sizeof(vieo_pose_frame))) != VIEO_OK) return rc;
  if ((rc = dO.ensure((size_t)std::max(n, 1) * sizeof(vieo_pose_obs))) != VIEO_OK) return rc;
  if ((rc = dU.ensure(std::max(n, 1))) != VIEO_OK) return rc;
  if ((rc = dR.ensure(sizeof(vieo_pose_result))) != VIEO_OK) return rc;
  vieo_pose_frame F = *h_frame;
  const vieo_pose_obs* src = h_obs + h_frame->obs_begin;
  F.obs_begin = 0;
  VIEO_HIP_CHECK(hipMemcpy(dF.p, &F, sizeof(F), hipMemcpyHostToDevice));
  if (n > 0) VIEO_HIP_CHECK(hipMemcpy(dO.p, src, (size_t)n * sizeof(vieo_pose_obs), hipMemcpyHostToDevice));
  rc = vieo_pose_optimization_batch_device(dF.as<vieo_pose_frame>(), 1, dO.as<vieo_pose_obs>(),
                                           dU.as<uint8_t>(), dR.as<vieo_pose_result>(), nullptr);
  if (rc != VIEO_OK) return rc;
  VIEO_HIP_CHECK(hipMemcpy(h_result, dR.p, sizeof(vieo_pose_result), hipMemcpyDeviceToHost));
  if (n > 0) VIEO_HIP_CHECK(hipMemcpy(h_outlier + h_frame->obs_begin, dU.p, n, hipMemcpyDeviceToHost));
  return VIEO_OK;
}

}  // extern "C"
