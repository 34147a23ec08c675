// ba_device.h -- device helpers shared by the bundle-adjustment kernels (pose_opt.hip,
// pose_opt_vio.hip): NavState retractions, the reprojection edge (EdgeReproject<DE,DV,2>,
// reference src/Odom/g2otypes.h:321-547 with PinholeCamera::Project's float rounding,
// common/camera_models/camera_pinhole.h:70-106), Huber kernel and block reductions.
#pragma once
#include <cfloat>

#include "common.h"
#include "wave_ops.h"

namespace vieo {

struct Est {
  double p[3];
  double qw, qx, qy, qz;
};

struct CamD {
  double fx, fy, cx, cy, bf;  // float parameters widened once
  double Rcb[9], tcb[3];
  int model = 0, num_k = 0;   // VIEO_CAM_*; distortion coefficients (float, widened)
  double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// camm::{Pinhole,Radtan,KB8}Camera::Project (common/camera_models/camera_*.h): the image point as the
// FLOAT the reference returns (Vec2data), and the 2x3 Jacobian d(u, v)/dPc when J != nullptr.
// Parameters are float (Tdata), the arithmetic double (Tcalc).
__device__ __forceinline__ void cam_project(const CamD& c, const double* P, double* uv, double* J) {
  const double x = P[0], y = P[1], z = P[2];
  if (c.model == 1) {  // Radtan, camera_radtan.h:61-129
    const double invz = 1 / z;
    double xn = x * invz, yn = y * invz;
    const double x2 = xn * xn, y2 = yn * yn, xy = xn * yn, r2 = x2 + y2;
    const double* kk = c.k;
    const double* pp = c.k + c.num_k;
    double fd = 1, term_r = 1;
    for (int i = 0; i < c.num_k; ++i) {
      term_r *= r2;
      fd += kk[i] * term_r;
    }
    if (J) {
      double fd2 = 0, coeff2 = 0;
      term_r = 1;
      for (int i = 2; i < c.num_k; ++i) {
        coeff2 += 2;
        fd2 += coeff2 * kk[i] * term_r;
        term_r *= r2;
      }
      const double du_dx = c.fx * invz * (fd + fd2 * x2 + 2 * (pp[0] * yn + 3 * pp[1] * xn));
      const double du_dy = c.fx * invz * (fd2 * xy + 2 * (pp[0] * xn + pp[1] * yn));
      const double du_dz = -(xn * du_dx + yn * du_dy);
      const double dv_dx = du_dy * c.fy / c.fx;
      const double dv_dy = c.fy * invz * (fd + fd2 * y2 + 2 * (pp[1] * xn + 3 * pp[0] * yn));
      const double dv_dz = -(xn * dv_dx + yn * dv_dy);
      J[0] = du_dx, J[1] = du_dy, J[2] = du_dz, J[3] = dv_dx, J[4] = dv_dy, J[5] = dv_dz;
    }
    const double xd = xn * fd + 2 * pp[0] * xy + pp[1] * (r2 + 2 * x2);
    const double yd = yn * fd + 2 * pp[1] * xy + pp[0] * (r2 + 2 * y2);
    uv[0] = (double)(float)(c.fx * xd * (1. / 1.) + c.cx);
    uv[1] = (double)(float)(c.fy * yd * (1. / 1.) + c.cy);
    return;
  }
  if (c.model == 2) {  // KB8, camera_kb8.h:68-157
    const double x2 = x * x, y2 = y * y, r2 = x2 + y2, r = sqrt(r2);
    if (r > (double)1e-5f) {
      const double theta = atan2(r, z), theta2 = theta * theta;
      double thetad = c.k[3] * theta2;
      thetad += c.k[2], thetad *= theta2, thetad += c.k[1], thetad *= theta2, thetad += c.k[0];
      thetad *= theta2, thetad += 1, thetad *= theta;
      const double mx = x * thetad / r, my = y * thetad / r;
      uv[0] = (double)(float)(c.fx * mx * (1. / 1.) + c.cx);
      uv[1] = (double)(float)(c.fy * my * (1. / 1.) + c.cy);
      if (J) {
        const double invr = 1. / r, d_r_d_x = x * invr, d_r_d_y = y * invr;
        const double tmp = 1. / (z * z + r2);
        const double d_thetad_x = d_r_d_x * z * tmp, d_thetad_y = d_r_d_y * z * tmp;
        double dd = 9.0 * c.k[3] * theta2;
        dd += 7.0 * c.k[2], dd *= theta2, dd += 5.0 * c.k[1], dd *= theta2, dd += 3.0 * c.k[0];
        dd *= theta2, dd += 1.0;
        const double invr2 = invr * invr;
        const double tr = thetad * invr;  // thetad / r once (a double-precision division is a dozen instructions)
        const double jxy = x * (dd * d_thetad_y * r - y * tr) * invr2;
        J[0] = c.fx * (x * r * dd * d_thetad_x + y2 * tr) * invr2;
        J[1] = c.fx * jxy;
        J[2] = -c.fx * x * dd * tmp;
        J[3] = c.fy * jxy;  // = J[1] * fy / fx
        J[4] = c.fy * (y * r * dd * d_thetad_y + x2 * tr) * invr2;
        J[5] = -c.fy * y * dd * tmp;
      }
      return;
    }
  }
  const double invz = 1. / z;  // pinhole (also KB8's degenerate case)
  uv[0] = (double)(float)(c.fx * x * invz + c.cx);
  uv[1] = (double)(float)(c.fy * y * invz + c.cy);
  if (J) {
    const double invz2 = invz * invz;
    J[0] = c.fx * invz, J[1] = 0, J[2] = -c.fx * x * invz2;
    J[3] = 0, J[4] = c.fy * invz, J[5] = -c.fy * y * invz2;
  }
}

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
    double sh, ch;
    sincos(half, &sh, &ch);
    imag = sh / theta;
    real = ch;
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

__device__ __forceinline__ double wave_sum_d(double v) { return wave_sum_f64(v); }  // DPP, fixed association order
// the xor butterfly through ds_bpermute: kept for the one-wavefront-per-frame instances of the pose kernels (with the
// DPP form those instances, at ~390 registers, abort on the device; the four-wavefront instances are fine)
__device__ __forceinline__ double wave_sum_d_bfly(double v) {
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

// the same for a workgroup of BS threads (BS = 64: one wavefront, no LDS round trip)
template <int N, int BS>
__device__ __forceinline__ void block_sum_bs(double* vals, double* s_red, int tid) {
  if (BS == 64) {
#pragma unroll
#ifdef VIEO_POSE_DPP64  // experiment: the DPP sums in the one-wavefront instances (tools/probe_dpp64.sh)
    for (int i = 0; i < N; i++) vals[i] = wave_sum_d(vals[i]);
#else
    for (int i = 0; i < N; i++) vals[i] = wave_sum_d_bfly(vals[i]);
#endif
  } else if (BS == 256)
    block_sum<N>(vals, s_red, tid);
  else {  // BS / 64 wavefronts: the same two steps, the partial sums added in a fixed tree order (s_red: BS / 64 x N)
    constexpr int W = BS / 64;
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int i = 0; i < N; i++) vals[i] = wave_sum_d(vals[i]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
      for (int i = 0; i < N; i++) s_red[wave * N + i] = vals[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; i++) {
      double t[W];
#pragma unroll
      for (int w = 0; w < W; w++) t[w] = s_red[w * N + i];
#pragma unroll
      for (int h = W / 2; h > 0; h >>= 1)
#pragma unroll
        for (int w = 0; w < h; w++) t[w] = t[w] + t[w + h];
      vals[i] = t[0];
    }
  }
}

// Block-wide sums of N values per thread through an LDS transpose: out[v] = sum over the BS threads of vals[v].
// A wavefront issues one instruction per 4 cycles however few lanes are useful, so N butterfly sums cost N x 6 steps x 3
// instructions per wavefront; here every thread stores its N values (thread-major inside blocks of 32, stride 34
// doubles: 16-byte aligned, at most 2 lanes per bank), thread (v, part) adds the 32 values of its block with four
// independent chains, and the BS / 32 parts of a value -- neighbouring lanes -- meet in up to three DPP steps: about
// 90 instructions per thread for N = 28 instead of about 500.  The association order is fixed (same result every run).
// tr: N * (BS / 32) * 34 doubles nobody else uses between the two barriers inside; out: N doubles, valid after return.
template <int N, int BS>
__device__ __forceinline__ void block_sum_lds(const double* vals, double* tr, double* out, int tid) {
  constexpr int P = BS / 32, CS = 34;
  static_assert(N * P <= BS && (P == 2 || P == 4 || P == 8), "one thread per (value, part)");
  {
    double* w = tr + (tid >> 5) * CS + (tid & 31);
#pragma unroll
    for (int v = 0; v < N; v++) w[v * P * CS] = vals[v];
  }
  __syncthreads();
  if (tid < N * P) {
    const double* r = tr + tid * CS;  // block (v * P + part) = tid
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
    for (int k = 0; k < 32; k += 4) s0 += r[k], s1 += r[k + 1], s2 += r[k + 2], s3 += r[k + 3];
    double s = (s0 + s1) + (s2 + s3);
#define VIEO_DPP_ADD(ctrl)                                                                              \
  {                                                                                                     \
    const int l2 = VIEO_DPP(0, __double2loint(s), ctrl, 0xF), h2 = VIEO_DPP(0, __double2hiint(s), ctrl, 0xF); \
    s += __hiloint2double(h2, l2);                                                                      \
  }
    VIEO_DPP_ADD(0xB1)               // quad_perm [1, 0, 3, 2]: lane ^ 1
    if (P >= 4) VIEO_DPP_ADD(0x4E)   // quad_perm [2, 3, 0, 1]: lane ^ 2
    if (P >= 8) VIEO_DPP_ADD(0x141)  // row_half_mirror: the other quad of the eight
#undef VIEO_DPP_ADD
    if ((tid & (P - 1)) == 0) out[tid / P] = s;
  }
  __syncthreads();
}

// EdgeReproject::linearizeOplus (g2otypes.h:439-498): Jacobian of the (up to 3) residual rows
// w.r.t. (dp, dphi) of the body pose, J[r*6 + 0..2] = d/dp, J[r*6 + 3..5] = d/dphi.
__device__ __forceinline__ void visual_jacobian(const CamD& c, const PoseXf& X, const double* p,
                                                const vieo_pose_obs& o, const double* Pc, double* J) {
  const double invz = 1 / Pc[2], invz2 = invz * invz;
  double Jp[9];
  Jp[0] = -(c.fx * invz), Jp[1] = 0, Jp[2] = -(-c.fx * Pc[0] * invz2);
  Jp[3] = 0, Jp[4] = -(c.fy * invz), Jp[5] = -(-c.fy * Pc[1] * invz2);
  Jp[6] = Jp[0], Jp[7] = Jp[1], Jp[8] = Jp[2] - c.bf * invz2;
  const double dP0 = (double)o.Xw[0] - p[0], dP1 = (double)o.Xw[1] - p[1], dP2 = (double)o.Xw[2] - p[2];
  double Pa[3];
  for (int m = 0; m < 3; m++) Pa[m] = X.Rwb[m] * dP0 + X.Rwb[3 + m] * dP1 + X.Rwb[6 + m] * dP2;
  double RH[9];  // Rcb * hat(Rwb^T (Xw - pwb))
  for (int m = 0; m < 3; m++) {
    const double a = c.Rcb[m * 3], b = c.Rcb[m * 3 + 1], d = c.Rcb[m * 3 + 2];
    RH[m * 3 + 0] = b * Pa[2] - d * Pa[1];
    RH[m * 3 + 1] = -a * Pa[2] + d * Pa[0];
    RH[m * 3 + 2] = a * Pa[1] - b * Pa[0];
  }
  for (int r = 0; r < 3; r++)
    for (int q = 0; q < 3; q++) {
      J[r * 6 + q] = -(Jp[r * 3] * c.Rcb[q] + Jp[r * 3 + 1] * c.Rcb[3 + q] + Jp[r * 3 + 2] * c.Rcb[6 + q]);
      J[r * 6 + 3 + q] = Jp[r * 3] * RH[q] + Jp[r * 3 + 1] * RH[3 + q] + Jp[r * 3 + 2] * RH[6 + q];
    }
}

// ---- a20: frames of a distorted multi-camera rig (Frame::usedistort_) ----------------------------
// vieo_camera -> CamD (parameters widened once); false when the model / coefficient count is unknown
__device__ __host__ inline bool cam_from_abi(const vieo_camera& c, CamD& d) {
  d.fx = c.fx, d.fy = c.fy, d.cx = c.cx, d.cy = c.cy, d.bf = 0;
  for (int i = 0; i < 9; i++) d.Rcb[i] = c.Rcb[i];
  for (int i = 0; i < 3; i++) d.tcb[i] = c.tcb[i];
  d.model = c.model, d.num_k = c.model == VIEO_CAM_RADTAN ? c.num_k : 0;
  for (int q = 0; q < 8; q++) d.k[q] = (double)c.dist[q];
  return !(c.model < 0 || c.model > 2 || (c.model == VIEO_CAM_RADTAN && (c.num_k < 2 || c.num_k > 6)));
}

// One reprojection edge of a pose optimisation: error (and the Jacobian rows when J != nullptr).
// MC = false: the frame's single rectified pinhole camera c0 with the per-estimate transforms X0.
// MC = true: the observation's own camera cams[(flags >> 8) & 15] (Radtan / KB8 / pinhole, monocular
// edges), whose Rcw / tcw are formed per edge from X0.Rwb and the body position p.
// camera c's world -> camera transform at the estimate X0 / p: xf[0..8] = Rcw = Rcb Rwb^T, xf[9..11] = tcw
__device__ __forceinline__ void rig_cam_xf(const CamD& c, const PoseXf& X0, const double* p, double* xf) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      xf[i * 3 + j] = c.Rcb[i * 3] * X0.Rwb[j * 3] + c.Rcb[i * 3 + 1] * X0.Rwb[j * 3 + 1] + c.Rcb[i * 3 + 2] * X0.Rwb[j * 3 + 2];
  for (int i = 0; i < 3; i++) xf[9 + i] = -(xf[i * 3] * p[0] + xf[i * 3 + 1] * p[1] + xf[i * 3 + 2] * p[2]) + c.tcb[i];
}

// cam_xf: the rig's per-camera transforms [n_cams][12] formed once per pass (rig_cam_xf, in LDS), or null: per edge
template <bool MC>
__device__ __forceinline__ double edge_eval(const CamD& c0, const CamD* cams, const PoseXf& X0, const double* p,
                                            const vieo_pose_obs& o, double* err, double* Pc, double* J,
                                            const double* cam_xf = nullptr) {
  if (!MC) {
    const double chi2 = edge_error(c0, X0, o, err, Pc);
    if (J) visual_jacobian(c0, X0, p, o, Pc, J);
    return chi2;
  }
  const int ci = (o.flags >> 8) & 3;
  const CamD& c = cams[ci];
  double Rcw[9], tcw[3];
  if (cam_xf) {
    for (int i = 0; i < 9; i++) Rcw[i] = cam_xf[ci * 12 + i];
    for (int i = 0; i < 3; i++) tcw[i] = cam_xf[ci * 12 + 9 + i];
  } else {
    double xf[12];
    rig_cam_xf(c, X0, p, xf);
    for (int i = 0; i < 9; i++) Rcw[i] = xf[i];
    for (int i = 0; i < 3; i++) tcw[i] = xf[9 + i];
  }
  const double Xw0 = o.Xw[0], Xw1 = o.Xw[1], Xw2 = o.Xw[2];
  for (int i = 0; i < 3; i++) Pc[i] = Rcw[i * 3] * Xw0 + Rcw[i * 3 + 1] * Xw1 + Rcw[i * 3 + 2] * Xw2 + tcw[i];
  double uv[2], Jc[6];
  cam_project(c, Pc, uv, J ? Jc : nullptr);
  err[0] = (double)o.u - uv[0];
  err[1] = (double)o.v - uv[1];
  const double info = (double)o.inv_sigma2;
  double chi2 = err[0] * (info * err[0]) + err[1] * (info * err[1]);
  const bool stereo = o.ur >= 0;
  if (stereo) {
    err[2] = (double)o.ur - (uv[0] - c.bf / Pc[2]);
    chi2 += err[2] * (info * err[2]);
  } else
    err[2] = 0;
  if (J) {
    const double invz = 1 / Pc[2], invz2 = invz * invz;
    double Jp[9];
    for (int i = 0; i < 6; i++) Jp[i] = -Jc[i];
    Jp[6] = Jp[0], Jp[7] = Jp[1], Jp[8] = Jp[2] - c.bf * invz2;
    const double dP0 = Xw0 - p[0], dP1 = Xw1 - p[1], dP2 = Xw2 - p[2];
    double Pa[3];
    for (int m = 0; m < 3; m++) Pa[m] = X0.Rwb[m] * dP0 + X0.Rwb[3 + m] * dP1 + X0.Rwb[6 + m] * dP2;
    double RH[9];
    for (int m = 0; m < 3; m++) {
      const double a = c.Rcb[m * 3], b = c.Rcb[m * 3 + 1], d = c.Rcb[m * 3 + 2];
      RH[m * 3 + 0] = b * Pa[2] - d * Pa[1];
      RH[m * 3 + 1] = -a * Pa[2] + d * Pa[0];
      RH[m * 3 + 2] = a * Pa[1] - b * Pa[0];
    }
    for (int r = 0; r < 3; r++)
      for (int q = 0; q < 3; q++) {
        J[r * 6 + q] = -(Jp[r * 3] * c.Rcb[q] + Jp[r * 3 + 1] * c.Rcb[3 + q] + Jp[r * 3 + 2] * c.Rcb[6 + q]);
        J[r * 6 + 3 + q] = Jp[r * 3] * RH[q] + Jp[r * 3 + 1] * RH[3 + q] + Jp[r * 3 + 2] * RH[6 + q];
      }
  }
  return chi2;
}

// accumulate J^T (w*info) J (21 upper entries) and J^T (-(info*err)*w) (6) of one edge
__device__ __forceinline__ void visual_accumulate(const double* J, const double* err, double info,
                                                  double r1, bool stereo, double* acc) {
  const double w = r1 * info;
  int t = 0;
  for (int a = 0; a < 6; a++) {
    for (int b = a; b < 6; b++, t++) {
      double s = J[a] * w * J[b] + J[6 + a] * w * J[6 + b];
      if (stereo) s += J[12 + a] * w * J[12 + b];
      acc[t] += s;
    }
    double s = J[a] * (-(info * err[0]) * r1) + J[6 + a] * (-(info * err[1]) * r1);
    if (stereo) s += J[12 + a] * (-(info * err[2]) * r1);
    acc[21 + a] += s;
  }
}

}  // namespace vieo
