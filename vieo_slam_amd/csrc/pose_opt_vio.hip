// pose_opt_vio.hip -- visual-inertial motion BA on gfx950:
//   template<class KeyFrame> int Optimizer::PoseOptimization(Frame*, KeyFrame*, gw, bComputeMarg, bNoMPs)
//   (reference include/Optimizer.h:208-816; marginal prior FillCovInv :126-206; edges
//   src/Odom/g2otypes.h:703-884 (EdgeNavStatePVR), g2otypes.cpp:14-34,84-124).
//
// One persistent workgroup per frame, as in pose_opt.hip.  The system is 15-dim (last state
// fixed) or 30-dim (last state carries a prior and is optimised too) and lives in LDS:
//   * visual edges: every thread evaluates its edges, wave + LDS reduction of 27 doubles;
//   * IMU / bias / prior edges: one lane evaluates the residual + Jacobians (a few hundred
//     flops), all threads then form J^T (rho' Omega) J entry-parallel;
//   * (H + lambda I) x = b by a right-looking LDL^T run by one wavefront on the LDS copy;
//   * LM control flow is evaluated redundantly by every thread from the same LDS scalars.
// FP64 throughout; parity with oracle/pose_opt_vio.cc <= 1e-4 on SE(3).
// This translation unit lets the compiler fuse a * b + c (the library is otherwise built with -ffp-contract=off for the
// bit-exact integer / float paths): the optimiser's parity bar is 1e-4 on SE(3), the kernel runs long single-wavefront
// chains where every instruction costs its 4 issue cycles and every dependent link ~7, and a fused multiply-add is one of
// each instead of two.
#pragma clang fp contract(fast)

#include <type_traits>
#include <vector>

#include "imu_device.h"

namespace vieo {

// ---- the single-lane edges, round 6: what an error evaluation has computed is not computed again by the linearisation
// that follows it at the same state (the LM loop linearises at the state of the last accepted -- i.e. last -- evaluation).
// A lane's time here is the latency of its transcendental chains (sincos 374 cycles, atan 211, sqrt 101, a division 76,
// tools/micro/lat_bench.hip), so the rules are: (1) a quaternion that is a product of unit quaternions is re-normalised
// by the series of 1 / sqrt(1 + e) (|e| ~ 1e-16: exact to rounding) instead of sqrt + division; (2) Exp(-Log(q)) is
// conj(q); (3) JrInv(Log(q)) and Jr(w) take sin / cos of the angle from the half-angle values the quaternion / the
// exponential already hold (sin t = 2 s c, 1 + cos t = 2 c^2, 1 - cos t = 2 s^2); (4) constants of the call (the
// quaternion of the pre-integrated rotation) are formed once.  Values agree with the plain forms (imu_device.h, kept
// for the local BA) to rounding; the optimiser's parity bar is 1e-4 on SE(3).
// (5) Inside |angle| < ~0.5 rad -- every increment, bias correction and edge error of a tracked frame -- Exp, Log, Jr and
// JrInv are polynomials in the squared angle (Taylor series to double rounding: the remainders are below 1e-17 on the
// stated domains): no sqrt, no sincos, no atan, one or two reciprocals.  Outside the domain the plain forms run.
//   cos(t/2) = sum (-1)^k u^k / (4^k (2k)!),  sin(t/2) / t = sum (-1)^k u^k / (2 4^k (2k+1)!),  u = t^2 < 1/4
//   Jr(w) = I - A hat(w) + B hat(w)^2,  A = (1 - cos t) / t^2 = sum (-1)^k u^k / (2k+2)!,  B = (t - sin t) / t^3 = sum (-1)^k u^k / (2k+3)!
//   Log(w, v) = (2 P / w) v with x^2 = |v|^2 / w^2 < 1/16, P = atan(x) / x = 1 - x^2 Q, Q = 1/3 - x^2/5 + x^4/7 - ...
//   JrInv(Log q) = I + hat(e) / 2 + g hat(e)^2,  g = (1 - t (1 + cos t) / (2 sin t)) / t^2 = Q / (4 P^2)   (-> 1/12 at 0)
constexpr double kCosH[9] = {1.0, -0.125, 0.0026041666666666665, -2.170138888888889e-05, 9.68812003968254e-08, -2.691144455467372e-10, 5.096864498991235e-13, -7.001187498614334e-16, 7.292903644389931e-19};
constexpr double kSinHT[9] = {0.5, -0.020833333333333332, 0.00026041666666666666, -1.5500992063492063e-06, 5.382288910934745e-09, -1.2232474797578965e-11, 1.9603324996120133e-14, -2.333729166204778e-17, 2.1449716601146855e-20};
constexpr double kJrA[9] = {0.5, -0.041666666666666664, 0.001388888888888889, -2.48015873015873e-05, 2.755731922398589e-07, -2.08767569878681e-09, 1.1470745597729725e-11, -4.779477332387385e-14, 1.5619206968586225e-16};
constexpr double kJrB[9] = {0.16666666666666666, -0.008333333333333333, 0.0001984126984126984, -2.7557319223985893e-06, 2.505210838544172e-08, -1.6059043836821613e-10, 7.647163731819816e-13, -2.8114572543455206e-15, 8.22063524662433e-18};
constexpr double kAtanQ[16] = {0.3333333333333333, -0.2, 0.14285714285714285, -0.1111111111111111, 0.09090909090909091, -0.07692307692307693, 0.06666666666666667, -0.058823529411764705, 0.05263157894736842, -0.047619047619047616, 0.043478260869565216, -0.04, 0.037037037037037035, -0.034482758620689655, 0.03225806451612903, -0.030303030303030304};
// (6) The hot forms are STRAIGHT-LINE code: a lane's error evaluation is one basic block in which the scheduler
// interleaves the independent chains (position / velocity rows beside the quaternion chain of the rotation rows) --
// with a branch per domain test each piece was a block of its own and the lane ran them one dependent instruction at a
// time.  Every polynomial is evaluated unconditionally, the domain tests only clear `ok`, and a caller whose evaluation
// ends with ok == false (angles beyond the domains, a quaternion that is not unit: never in a tracked sequence) runs
// the plain forms of imu_device.h after it.
__device__ __forceinline__ Qd q_renorm(const Qd& q, bool& ok) {
  const double e = (q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z) - 1.0;
  ok = ok && fabs(e) < 1e-5;
  const double r = 1.0 - e * (0.5 - 0.375 * e);
  return Qd{q.w * r, q.x * r, q.y * r, q.z * r};
}
template <int N>
__device__ __forceinline__ double horner(const double (&c)[N], double u) {
  double r = c[N - 1];
#pragma unroll
  for (int k = N - 2; k >= 0; k--) r = __builtin_fma(r, u, c[k]);
  return r;
}
__device__ __forceinline__ double rcp_nr(double d) {  // 1 / d: v_rcp_f64 + two Newton steps (as the solver's pivots)
  double inv = __builtin_amdgcn_rcp(d);
  inv = __builtin_fma(__builtin_fma(-d, inv, 1.0), inv, inv);
  return __builtin_fma(__builtin_fma(-d, inv, 1.0), inv, inv);
}
// SO3ex::exp (so3_extra.h:121-142) as a unit quaternion; domain |w|^2 < 1/4
__device__ __forceinline__ Qd so3_exp_unit(const double* w, bool& ok) {
  const double u = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  ok = ok && u < 0.25;
  const double real = horner(kCosH, u), imag = horner(kSinHT, u);
  return q_renorm(Qd{real, imag * w[0], imag * w[1], imag * w[2]}, ok);
}
// SO3ex::log (so3_extra.h:144-190) of a unit quaternion, and g of JacobianRInv(log) = I + hat / 2 + g hat^2;
// domain tan^2(half angle) < 1/16
__device__ __forceinline__ void so3_log_unit(const Qd& q, double* out, double* g, bool& ok) {
  const double n2 = q.x * q.x + q.y * q.y + q.z * q.z, w2 = q.w * q.w;
  ok = ok && n2 < 0.0625 * w2;
  const double iw = rcp_nr(q.w), x2 = n2 * (iw * iw);
  const double Q = horner(kAtanQ, x2), P = __builtin_fma(-x2, Q, 1.0);
  const double f = 2.0 * P * iw;
  out[0] = f * q.x, out[1] = f * q.y, out[2] = f * q.z;
  *g = Q * rcp_nr(4.0 * P * P);
}
// g of JacobianRInv for an angle outside the polynomial's domain (the plain forms' fallback)
__device__ __noinline__ double so3_JrInv_g_of(const double* e) {
  const double th2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2], th = sqrt(th2);
  if (th < 1e-5) return 1. / 12.;
  double sth, cth;
  sincos(th, &sth, &cth);
  return (1.0 - (1.0 + cth) * th / (2.0 * sth)) / th2;
}
__device__ __forceinline__ void so3_JrInv_g(const double* e, double g, double* J) {
  // hat(e)^2 = e e^T - |e|^2 I
  const double xx = e[0] * e[0], yy = e[1] * e[1], zz = e[2] * e[2];
  const double xy = g * e[0] * e[1], xz = g * e[0] * e[2], yz = g * e[1] * e[2];
  J[0] = 1.0 - g * (yy + zz), J[1] = xy - 0.5 * e[2], J[2] = xz + 0.5 * e[1];
  J[3] = xy + 0.5 * e[2], J[4] = 1.0 - g * (xx + zz), J[5] = yz - 0.5 * e[0];
  J[6] = xz - 0.5 * e[1], J[7] = yz + 0.5 * e[0], J[8] = 1.0 - g * (xx + yy);
}
// JacobianR(w) (so3_extra.h:254-270); domain |w|^2 < 1/4 (the caller knows from the cache)
__device__ __forceinline__ void so3_Jr_unit(const double* w, double* J) {
  const double u = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double A = horner(kJrA, u), B = horner(kJrB, u);
  const double xx = w[0] * w[0], yy = w[1] * w[1], zz = w[2] * w[2];
  const double xy = B * w[0] * w[1], xz = B * w[0] * w[2], yz = B * w[1] * w[2];
  J[0] = 1.0 - B * (yy + zz), J[1] = xy + A * w[2], J[2] = xz - A * w[1];
  J[3] = xy - A * w[2], J[4] = 1.0 - B * (xx + zz), J[5] = yz + A * w[0];
  J[6] = xz + A * w[1], J[7] = yz - A * w[0], J[8] = 1.0 - B * (xx + yy);
}
struct RotCache {  // written by an error evaluation, read by the linearisation at the same state
  Qd qe, qb;        // inertial edge: the error quaternion (err_R = Log qe) and conj(q_i) q_j
  double g_e;       // JrInv(err_R) = I + hat / 2 + g_e hat^2
  double w[3];      // JgR dbg_i
  Qd qp;            // prior edge: conj(q_prior) q_i, err_P[6..9) = Log qp
  double g_p;
  int w_small;      // |w|^2 < 1/4: Jr(w) by its polynomial
};

// EdgeNavStatePriorPVRBias (g2otypes.cpp:84-124); J is 15 x 15: cols 0..8 PVR_i, 9..14 Bias_i
__device__ __forceinline__ void prior_error(const NSd& pr, const NSd& si, double* err, RotCache& C) {
  double Rb[9], d[3];
  q_to_R(q_of(pr), Rb);
  for (int k = 0; k < 3; k++) d[k] = si.p[k] - pr.p[k];
  mTv3(Rb, d, err);
  bool ok = true;
  Qd q = q_renorm(q_mul(q_conj(q_of(pr)), q_of(si)), ok);
  double g;
  so3_log_unit(q, err + 6, &g, ok);
  for (int k = 0; k < 3; k++) {
    err[3 + k] = si.v[k] - pr.v[k];
    err[9 + k] = si.bg[k] + si.dbg[k] - (pr.bg[k] + pr.dbg[k]);
    err[12 + k] = si.ba[k] + si.dba[k] - (pr.ba[k] + pr.dba[k]);
  }
  if (!ok) {  // the plain forms
    q = q_norm(q_mul(q_conj(q_of(pr)), q_of(si)));
    so3_log_q(q, err + 6);
    g = so3_JrInv_g_of(err + 6);
  }
  C.qp = q, C.g_p = g;
}
// The Jacobian's constant part (zeros, the three identity blocks) is written once per call (prior_jacobian_init, all
// threads); a linearisation writes the two blocks that depend on the state: R_prior^T R_i = R(conj(q_prior) q_i), the
// quaternion the error evaluation formed, and JrInv of the rotation error.
__device__ __forceinline__ void prior_jacobian_init(double* J, int tid, int nthreads) {
  for (int e = tid; e < 225; e += nthreads) {
    const int r = e / 15, c = e - r * 15;
    J[e] = (r == c && ((r >= 3 && r < 6) || r >= 9)) ? 1.0 : 0.0;
  }
}
__device__ __forceinline__ void prior_linearize(const double* err, double* J, const RotCache& C) {
  double tmp[9], Jrinv[9];
  q_to_R(C.qp, tmp);
  set3(J, 15, 0, 0, tmp, 1.0);
  so3_JrInv_g(err + 6, C.g_p, Jrinv);
  set3(J, 15, 6, 6, Jrinv, 1.0);
}

// EdgeNavStateI computeError, rotation rows (imu_device.h imu_error part 2) with the intermediates kept; qRij = the
// quaternion of the pre-integrated rotation (R_to_q(M.Rij), once per call)
__device__ __forceinline__ void imu_error_rot(const vieo_imu_preint& M, const Qd& qRij, const NSd& si, const NSd& sj, double* err, RotCache& C) {
  double w[3];
  mv3(M.JgR, si.dbg, w);
  bool ok = true;
  const Qd qx = so3_exp_unit(w, ok);
  const bool w_small = ok;
  Qd qb = q_renorm(q_mul(q_conj(q_of(si)), q_of(sj)), ok);  // (independent of the chain through qx)
  const Qd qa = q_renorm(q_mul(qRij, qx), ok);
  Qd qe = q_renorm(q_mul(q_conj(qa), qb), ok);
  double g;
  so3_log_unit(qe, err + 6, &g, ok);
  if (!ok) {  // the plain forms
    qb = q_norm(q_mul(q_conj(q_of(si)), q_of(sj)));
    qe = q_norm(q_mul(q_conj(q_norm(q_mul(qRij, so3_exp_q(w)))), qb));
    so3_log_q(qe, err + 6);
    g = so3_JrInv_g_of(err + 6);
  }
  C.qe = qe, C.qb = qb, C.g_e = g;
  C.w[0] = w[0], C.w[1] = w[1], C.w[2] = w[2], C.w_small = w_small ? 1 : 0;
}
// The inertial Jacobian (9 x 24, imu_device.h imu_linearize with (idR, idV) = (6, 3)) has 14 non-zero 3 x 3 blocks, five
// of them constants of the call (-I and the four bias Jacobians of the pre-integration): those and the zeros are
// written once (imu_jacobian_init, all threads); a linearisation writes the nine blocks that depend on the states.
__device__ __forceinline__ void imu_jacobian_init(const vieo_imu_preint& M, double* J, int tid, int nthreads) {
  for (int e = tid; e < 9 * 24; e += nthreads) {
    const int r = e / 24, c = e - r * 24;
    double v = 0.0;
    if (r < 3) {
      if (c >= 9 && c < 12) v = (c - 9 == r) ? -1.0 : 0.0;
      if (c >= 18 && c < 21) v = -M.Jgp[r * 3 + c - 18];
      if (c >= 21) v = -M.Jap[r * 3 + c - 21];
    } else if (r < 6) {
      if (c >= 18 && c < 21) v = -M.Jgv[(r - 3) * 3 + c - 18];
      if (c >= 21) v = -M.Jav[(r - 3) * 3 + c - 21];
    }
    J[e] = v;
  }
}
// position and velocity rows (imu_linearize part 1 without its constant blocks)
__device__ __forceinline__ void imu_linearize_pv(const vieo_imu_preint& M, const double* gw, const NSd& si, const NSd& sj, double* J) {
  const int ld = 24, cj = 0, ci = 9, idR = 6, idV = 3;
  double Ri[9], RiT[9], Rj[9], t[3], r[3], Hm[9], tmp[9];
  q_to_R(q_of(si), Ri);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) RiT[i * 3 + j] = Ri[j * 3 + i];
  q_to_R(q_of(sj), Rj);
  const double dt = M.dt;
  for (int k = 0; k < 3; k++) t[k] = sj.p[k] - si.p[k] - si.v[k] * dt - gw[k] * (dt * dt / 2);
  mv3(RiT, t, r);
  hat3(r, Hm);
  set3(J, ld, 0, ci + idR, Hm, 1.0);
  set3(J, ld, 0, ci + idV, RiT, -dt);
  mm3(RiT, Rj, tmp);
  set3(J, ld, 0, cj + 0, tmp, 1.0);
  for (int k = 0; k < 3; k++) t[k] = sj.v[k] - si.v[k] - gw[k] * dt;
  mv3(RiT, t, r);
  hat3(r, Hm);
  set3(J, ld, idV, ci + idR, Hm, 1.0);
  set3(J, ld, idV, ci + idV, RiT, -1.0);
  set3(J, ld, idV, cj + idV, RiT, 1.0);
}
// EdgeNavStateI linearizeOplus, rotation rows (imu_device.h imu_linearize part 2; the caller has cleared J) from the
// cache of the error evaluation at the same state
__device__ __forceinline__ void imu_linearize_rot(const vieo_imu_preint& M, const double* err, double* J, const RotCache& C) {
  const int ld = 24, cj = 0, ci = 9, cb = 18, idR = 6;
  double Jrinv[9], Rji[9], tmp[9], tmp2[9], E[9], Jr[9];
  so3_JrInv_g(err + idR, C.g_e, Jrinv);
  q_to_R(q_conj(C.qb), Rji);                    // R_j^T R_i
  mm3(Jrinv, Rji, tmp);
  set3(J, ld, idR, ci + idR, tmp, -1.0);
  q_to_R(q_conj(C.qe), E);                      // Exp(-err_R)
  so3_Jr_unit(C.w, Jr);
  if (!C.w_small) so3_Jr_d(C.w, Jr);
  mm3(Jrinv, E, tmp);
  mm3(tmp, Jr, tmp2);
  mm3(tmp2, M.JgR, tmp);
  set3(J, ld, idR, cb + 0, tmp, -1.0);
  set3(J, ld, idR, cj + idR, Jrinv, 1.0);
}
// NavState::IncSmall(dPVR) + IncSmallBias (imu_device.h ns_inc)
__device__ __forceinline__ void ns_inc_unit(NSd& s, const double* d, const double* db) {
  double R[9], Rd[3];
  q_to_R(q_of(s), R);
  mv3(R, d, Rd);
  bool ok = true;
  Qd q = q_renorm(q_mul(q_of(s), so3_exp_unit(d + 6, ok)), ok);
  if (!ok) q = q_norm(q_mul(q_of(s), so3_exp_q(d + 6)));
  for (int i = 0; i < 3; i++) s.p[i] += Rd[i], s.v[i] += d[3 + i];
  s.qw = q.w, s.qx = q.x, s.qy = q.y, s.qz = q.z;
  for (int i = 0; i < 3; i++) s.dbg[i] += db[i], s.dba[i] += db[3 + i];
}

// ---- the visual edges of a one-camera frame as STRAIGHT-LINE code (round 6).  A pass over the edges was a loop with a
// `continue` for the edges of the wrong level, a branch for the stereo row and one inside the Huber kernel: every edge a
// chain of basic blocks in which one wavefront per SIMD runs its ~320 double-precision instructions one dependent link
// after the other (~9 cycles each against 4 of issue).  Here an edge is evaluated unconditionally -- a masked or
// out-of-range slot with weight 0, the stereo row of a monocular edge as zeros, the Huber kernel by selection -- so that
// TWO edges form one block and the scheduler interleaves their chains.  A valid edge goes through the operations of
// edge_error / visual_jacobian / visual_accumulate (ba_device.h) in their order; the zeros added for the other slots do
// not change a sum.
struct VisFlat {
  double err[3], Pc[3], info, chi2;
  bool stereo;
};
__device__ __forceinline__ void vis_error_flat(const CamD& c, const PoseXf& X, const vieo_pose_obs& o, bool valid, VisFlat& E) {
  const double Xw0 = o.Xw[0], Xw1 = o.Xw[1], Xw2 = o.Xw[2];
  for (int i = 0; i < 3; i++) E.Pc[i] = X.Rcw[i * 3] * Xw0 + X.Rcw[i * 3 + 1] * Xw1 + X.Rcw[i * 3 + 2] * Xw2 + X.tcw[i];
  if (!valid) E.Pc[2] = 1.0;  // (a slot that does not count must not divide by a zero depth: inf x weight 0 = NaN)
  const double invz = 1. / E.Pc[2];
  const double u = (double)(float)(c.fx * E.Pc[0] * invz + c.cx);
  const double v = (double)(float)(c.fy * E.Pc[1] * invz + c.cy);
  E.err[0] = (double)o.u - u;
  E.err[1] = (double)o.v - v;
  E.info = (double)o.inv_sigma2;
  E.stereo = o.ur >= 0;
  const double e2 = (double)o.ur - (u - c.bf / E.Pc[2]);
  E.err[2] = E.stereo ? e2 : 0.0;
  const double chi = E.err[0] * (E.info * E.err[0]) + E.err[1] * (E.info * E.err[1]);
  const double chis = chi + E.err[2] * (E.info * E.err[2]);
  E.chi2 = E.stereo ? chis : chi;
}
// RobustKernelHuber::robustify by selection (huber() of ba_device.h: same values)
__device__ __forceinline__ void huber_flat(double e, double delta, double dsqr, bool robust, double* r0, double* r1) {
  const bool big = robust && !(e <= dsqr);
  const double sq = sqrt(big ? e : dsqr);
  *r0 = big ? 2 * sq * delta - dsqr : e;
  *r1 = big ? delta / sq : 1.;
}
// visual_jacobian with the stereo row of a monocular edge as zeros
__device__ __forceinline__ void vis_jacobian_flat(const CamD& c, const PoseXf& X, const double* p, const vieo_pose_obs& o,
                                                  const VisFlat& E, double* J) {
  const double invz = 1 / E.Pc[2], invz2 = invz * invz;
  double Jp[9];
  Jp[0] = -(c.fx * invz), Jp[1] = 0, Jp[2] = -(-c.fx * E.Pc[0] * invz2);
  Jp[3] = 0, Jp[4] = -(c.fy * invz), Jp[5] = -(-c.fy * E.Pc[1] * invz2);
  Jp[6] = E.stereo ? Jp[0] : 0.0, Jp[7] = 0, Jp[8] = E.stereo ? Jp[2] - c.bf * invz2 : 0.0;
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
// visual_accumulate with all three rows (the third is zero for a monocular edge) and the weight of the slot
__device__ __forceinline__ void vis_accumulate_flat(const double* J, const double* err, double info, double r1, double* acc) {
  const double w = r1 * info;
  int t = 0;
  for (int a = 0; a < 6; a++) {
    for (int b = a; b < 6; b++, t++) {
      double s = J[a] * w * J[b] + J[6 + a] * w * J[6 + b];
      s += J[12 + a] * w * J[12 + b];
      acc[t] += s;
    }
    double s = J[a] * (-(info * err[0]) * r1) + J[6 + a] * (-(info * err[1]) * r1);
    s += J[12 + a] * (-(info * err[2]) * r1);
    acc[21 + a] += s;
  }
}

// LDS hand-over between the lanes of one wavefront
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Right-looking LDL^T of the n x n matrix A (LDS, row-major, overwritten; n <= 64) and solve A x = b,
// executed by ONE wavefront (lane = threadIdx & 63).  Returns false on a non-positive pivot.
// col = un-normalised column j, lcol = multipliers col / d (one division per row, all rows at once).
__device__ bool wave_ldlt_solve(double* A, const double* b, double* x, double* col, double* lcol, double* D,
                                int n, int lane) {
  bool ok = true;
  for (int j = 0; j < n; j++) {
    const double d = A[j * n + j];
    if (!(d > 0)) {
      ok = false;
      break;
    }
    if (lane > j && lane < n) {
      const double c = A[lane * n + j];
      col[lane] = c, lcol[lane] = c / d;
    }
    if (lane == 0) D[j] = d;
    wave_sync();
    for (int i = j + 1 + (lane >> 3); i < n; i += 8) {  // lower triangle, 8 x 8 lanes
      const double li = lcol[i];
      for (int k = j + 1 + (lane & 7); k <= i; k += 8) A[i * n + k] -= li * col[k];
    }
    if (lane > j && lane < n) A[lane * n + j] = lcol[lane];  // L(i,j)
    wave_sync();
  }
  if (!ok) return false;
  // forward: y = L^-1 b (column oriented)
  for (int i = lane; i < n; i += 64) x[i] = b[i];
  wave_sync();
  for (int j = 0; j < n; j++) {
    const double yj = x[j];
    if (lane > j && lane < n) x[lane] -= A[lane * n + j] * yj;
    wave_sync();
  }
  if (lane < n) x[lane] /= D[lane];
  wave_sync();
  for (int j = n - 1; j >= 0; j--) {
    const double xj = x[j];
    if (lane < j) x[lane] -= A[j * n + lane] * xj;
    wave_sync();
  }
  return true;
}

// The same factorisation with the matrix in registers: lane i owns row i of (H + lambda I), column
// values travel by v_readlane (the column index is a compile-time constant after unrolling), so the
// N pivot steps need no LDS round trip and no barrier.  Same operations in the same order as above:
// l = c / d, then a_ik -= l_i * c_k.  L is written to Ls (row-major N x N) for the back substitution,
// whose column reads are all issued up front.
__device__ __forceinline__ double readlane_d(double v, int srclane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)b, srclane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), srclane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// The system's structure (round 4): the bias of the current state (unknowns 9..14) meets the rest of the system only
// through the random-walk edge, one diagonal entry w_t per component and -w_t towards the last state's bias (24 + t) when
// that is free -- the inertial edge's Jacobian has no column for it, the visual, prior and encoder edges neither.  Its six
// unknowns are eliminated in closed form (pivot d_t = w_t + lambda) and the factorisation runs on the remaining NR = 24
// (last state free) or 9 (last state fixed) unknowns: 0.65x / 0.25x of the dependent pivot chain of the 30 / 15-dim form.
// Reduced index r -> system index: r < 9 ? r : r + 6.  Lane r owns row r and keeps 1 / d_r; the columns broadcast during
// the factorisation stay in LDS (Ls, NR x 34 doubles) and serve the back substitution.
// (the "TP:" comments mark the phases tools/micro/gen_solve_bench.py puts its s_memtime probes at)
// The function is out of line (two instances, one call site each); its arrays are named in the LDS address space so that
// every access is a plain ds_ instruction -- through generic pointers each one carried a 64-bit address add and a null
// test of the flat -> LDS cast (3 extra instructions per access, about a third of the function).
typedef __attribute__((address_space(3))) double lds_f64;
template <int NR>
__device__ bool wave_solve_vio(const lds_f64* H, int n, double lambda, const lds_f64* b, lds_f64* x, lds_f64* Ls, int lane) {
  // the lane masks below are formed here, per call: hoisted out of the caller's trial loop they would live in ~100
  // scalar registers, i.e. be spilled to lanes of a VGPR and read back by v_readlane, two instructions per use
  asm volatile("" : "+v"(lane));
  double a[NR];
  // lanes 32..63 mirror lanes 0..31 (same row, same values, same LDS addresses): no execution mask anywhere
  const int l32 = lane & 31;
  const int row = l32 < NR ? l32 : 0;
  const int frow = row < 9 ? row : row + 6;
#pragma unroll
  for (int k = 0; k < NR; k++) a[k] = H[frow * n + (k < 9 ? k : k + 6)];  // lambda joins the pivots as they are read
  double y = l32 < NR ? b[frow] : 0.0;
  bool ok = true;
  // the six eliminated unknowns: lane t < 6 keeps its pivot's reciprocal, lanes 18 + t of the 24-dim form take the update
  double dbias = 1.0, off = 0.0;
  {
    const int t = NR == 24 ? (l32 >= 18 && l32 < 24 ? l32 - 18 : (l32 < 6 ? l32 : 0)) : (l32 < 6 ? l32 : 0);
    const double d = H[(9 + t) * n + 9 + t] + lambda;
    if (!(d > 0)) ok = false;
    double inv = __builtin_amdgcn_rcp(d);
    inv = __builtin_fma(__builtin_fma(-d, inv, 1.0), inv, inv);
    inv = __builtin_fma(__builtin_fma(-d, inv, 1.0), inv, inv);
    dbias = inv;
    if (NR == 24) {
      off = H[(9 + t) * n + 24 + t];
      const double l = l32 >= 18 && l32 < 24 ? off * inv : 0.0;
#pragma unroll
      for (int k = 18; k < 24; k++) a[k] = __builtin_fma(k == l32 ? -l : 0.0, off, a[k]);
      y = __builtin_fma(-l, b[9 + t], y);
    }
  }
  ok = __all(ok);
  // TP:factor
  // Column j travels to the other rows through LDS (one ds_write per lane, then wave-uniform -- broadcast -- reads, two
  // values per instruction) instead of two v_readlane per value: a wavefront issues one instruction every 4 cycles
  // (double precision included: tools/micro/lat_bench.hip) whatever the number of useful lanes, so the count of
  // instructions is the cost, and this form has a quarter of them.  Every column keeps its own 34 doubles of Ls, zero at and above the diagonal: together they are
  // L^T un-normalised, which is what the backward substitution reads (lane i its own column i) -- L is never stored.
  lds_f64* colbuf = Ls;  // NR x kCS doubles: 34 keeps pairs 16-byte aligned and spreads the backward pass's per-lane
  constexpr int kCS = 34;  // column reads over the banks (32 would put every lane on one bank: 32 passes per read)
  // The pivot chain -- d_j -> 1 / d_j (v_rcp_f64 + two Newton steps) -> l -> d_(j+1) -- never waits for LDS: row j + 1's
  // own update of its diagonal needs only its own two values (A[j+1][j] is its own column entry), so the next pivot is
  // formed from registers and read by v_readlane while the broadcast of column j is still on its way.
  // The forward substitution y = L^-1 b rides along (b is one more column of the row: y_j is final once pivot j starts).
  // The broadcast of column j is consumed one pivot late (cp / lp): its ds_reads are issued at the end of pivot j and
  // their values first needed after pivot j + 1's reciprocal chain, which covers the LDS round trip (~77 cycles); the
  // empty asm ties that first use to the chain's result so that the scheduler does not pull the wait up.  The order of
  // the updates each entry receives is that of the plain right-looking form.
  double d = readlane_d(a[0], 0) + lambda;
  double dinv = 1.0;  // lane j < NR: 1 / d_j
  double cp[NR], lp = 0.0;
#pragma unroll
  for (int j = 0; j < NR; j++) {
    const double c = l32 > j ? a[j] : 0.0;  // un-normalised column entry of this row; rows <= j are done
    colbuf[j * kCS + l32] = c;
    if (!(d > 0)) ok = false;
    // 1 / d once per column (v_rcp_f64 + two Newton steps, as k_lba_ldlt16 does) instead of a division per row
    double inv = __builtin_amdgcn_rcp(d);
    inv = __builtin_fma(__builtin_fma(-d, inv, 1.0), inv, inv);
    inv = __builtin_fma(__builtin_fma(-d, inv, 1.0), inv, inv);
    dinv = l32 == j ? inv : dinv;
    const double l = c * inv;
    if (j + 1 < NR) {
      const int j1 = j + 1 < NR ? j + 1 : j;
      if (j >= 1) {
        asm volatile("" : "+v"(cp[j1]) : "v"(inv));
        a[j1] = __builtin_fma(-lp, cp[j1], a[j1]);  // pivot j - 1's update of column j + 1
      }
      // column j + 1 is the NEXT broadcast: its update takes A[j+1][j] by v_readlane, so the next ds_write does not wait
      // for this column's trip through LDS
      a[j1] = __builtin_fma(-l, readlane_d(c, j + 1), a[j1]);
      d = readlane_d(a[j1], j + 1) + lambda;  // lane j + 1: its new diagonal
    }
    y = __builtin_fma(-l, readlane_d(y, j), y);
    wave_sync();
    if (j >= 1)
#pragma unroll
      for (int k = j + 2; k < NR; k++) a[k] = __builtin_fma(-lp, cp[k], a[k]);
#pragma unroll
    for (int k = j + 2; k < NR; k++) cp[k] = colbuf[j * kCS + k];
    lp = l;
  }
  if (!ok) return false;
  // TP:backward
  // backward: x_i = (y_i - sum_(j > i) A_ji x_j) / d_i with A_ji = colbuf[i][j] (lane i: its own column, fetched up front)
  double lt[NR];
#pragma unroll
  for (int k = 1; k < NR; k++) lt[k] = colbuf[row * kCS + k] * dinv;
  y *= dinv;
#pragma unroll
  for (int j = NR - 1; j > 0; j--) y = __builtin_fma(-lt[j], readlane_d(y, j), y);
  // TP:write
  if (lane < NR) x[frow] = y;
  wave_sync();
  // the eliminated unknowns: x_(9+t) = (b_(9+t) - off_t x_(24+t)) / d_t
  if (lane < 6) x[9 + lane] = (b[9 + lane] - (NR == 24 ? off * x[24 + lane] : 0.0)) * dbias;
  wave_sync();
  // TP:end
  return true;
}

// observations per frame: one bit per edge in a 64-bit mask per lane -> 64 x threads (4096 / 16384)
#define kVioMaxObs (64 * BS)

struct VioShared {
  NSd nsj, nsi, bkj, bki, prior;
  // The system and the solver's column buffers | the marginalisation's blocks after the last optimize() (the system
  // is dead by then).  LDS per frame decides how many one-wavefront frames a CU holds: 32 KB -> 4 (40 KB would be 3).
  union {
    struct {
      double H[900], L[900];
    };
    struct {
      double cov[225], C[225], E[225], Cinv[225 * 2];
    };
  };
  double b[32], x[32], tr_tail[64];  // (H .. tr_tail: block_sum_lds' buffer of the 64-thread instances)
  double red[4 * 28], vis[28];
  double errI[9], errB[6], errP[15], wI[9], wP[15];
  double JI[9 * 24], JP[225], InfoI[81], T[225], TP[225], Hp[225];  // Hp: the frame's H_prior, staged once
  double chiq[8];       // the generic edges' quadratic forms after the robust kernel and rho': (I, B, P) x (rho0, rho1)
  double xv[4];         // a scalar on its way through PoseXchg
  int xfail;            // an exchange timed out (a replica never arrived): the frame is reported as failed
  RotCache rc;          // intermediates of the last error evaluation (inertial edge's rotation rows, prior edge)
  double qRij[4];       // quaternion of the pre-integrated rotation (constant of the call)
  double gw[4];         // gravity (a pointer into LDS for the out-of-line edge functions; a local array would sit in scratch)
  vieo_imu_preint imu;  // the frame's pre-integration without Sigma, staged once: the single-lane edge evaluations of
                        // every trial read it, and a trip to L2 per dependent batch of loads was a fifth of their time
  int ok;
};

// EdgeEncNavStatePVR (g2otypes.h:591-668, Optimizer.h:345-363) of the ENC instance: one lane evaluates the
// 6-row edge, the block folds J = [Jj | Ji] (columns (dp, dphi) of the two PVR vertices) into the system.
struct VioEncShared {
  double Info[36], err[6], we[6], chi, J[6 * 12], T[6 * 12];
};

__device__ __noinline__ void vio_enc_setup(const vieo_pose_enc* pe, VioEncShared* S) {
  double M[6][12];  // Sigma_E^-1 by Gauss-Jordan with partial pivoting
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

__device__ __noinline__ void vio_enc_eval(const vieo_pose_enc* pe, VioEncShared* S, const NSd* nsi, const NSd* nsj,
                                          int jac) {
  const NSd si = *nsi, sj = *nsj;
  double err[6], Ji[36], Jj[36];
  enc_edge_eval(si, sj, pe->enc.delx, pe->qRbe, pe->pbe, err, jac ? Ji : nullptr, jac ? Jj : nullptr);
  double c2 = 0;
  for (int a = 0; a < 6; a++) {
    double t = 0;
    for (int q = 0; q < 6; q++) t += S->Info[a * 6 + q] * err[q];
    S->err[a] = err[a], S->we[a] = t;
    c2 += err[a] * t;
  }
  S->chi = c2;
  if (!jac) return;
  for (int a = 0; a < 6; a++)
    for (int q = 0; q < 6; q++) S->J[a * 12 + q] = Jj[a * 6 + q], S->J[a * 12 + 6 + q] = Ji[a * 6 + q];
}

// BS threads per frame: 256 (four wavefronts share a frame: lowest latency for a few frames) or 64
// (one wavefront per frame, four frames per CU in flight: highest throughput for large batches)
// MC as in pose_opt.hip: the instance for frames of a distorted multi-camera rig (n_cams > 0, a20)
// ENC: the instance for frames that carry an encoder measurement (base.enc with dt != 0, a16)
// other_launched: bit 0 the other camera kind, bit 1 the other encoder kind has its own launch in this batch
#ifdef VIEO_POSE_PROBE
__device__ unsigned long long g_pose_probe[24];
#define RT(i, stmt) do { const unsigned long long a_ = __builtin_amdgcn_s_memtime(); stmt; atomicAdd(&g_pose_probe[i], __builtin_amdgcn_s_memtime() - a_); } while (0)
#define PP(i) do { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_pose_probe[i], t_ - pp_last); pp_last = t_; } } while (0)
#else
#define PP(i)
#define RT(i, stmt) stmt
#endif
// A frame on several workgroups (rig frames of a small call: thousands of visual edges, one frame).  The kXG workgroups
// of a frame are REPLICAS: each runs the whole optimisation -- the same serial mathematics on the same data, hence the
// same decisions -- but goes through only its share of the visual edges; wherever a sum over the visual edges is taken
// (28 per linearisation, 1 per trial, 1 per classification) the replicas exchange their partial sums through this record
// and add them in the same order.  No master, no commands: a workgroup only ever waits for the others to arrive.
// Protocol (MI355X: the per-XCD L2s are not coherent, a CU's L1 is never refreshed by another CU's stores): every
// partial sum travels as ONE 16-byte granule written with a system-coherent store and read with system-coherent loads
// (`sc0 sc1` on both sides: no fence, no counter).  The granule is two 8-byte halves, each (32 value bits, 32-bit tag),
// tag = (launch number & 0xFFFFF) << 12 | exchange number: a granule counts as arrived only when BOTH halves carry the
// tag, so the protocol needs single-copy atomicity of aligned 8-byte stores only -- a 16-byte store that reached memory
// as two halves shows a stale tag in one of them and is polled again, never read as (fresh tag, stale value).  A granule
// is its own "ready" flag and nothing is cleared between launches (a slot is rewritten by every launch, so the stale
// tags a poll can meet are the previous launch's: 20 bits of launch number tell them apart).  A thread publishes its
// value, then polls the kXG granules of its index until all carry the tag, and adds them in replica order.  Granules are
// double-buffered by the parity of the exchange (a fast replica may be one exchange ahead, never two: it needs every
// replica's granule of this exchange before it can leave it).  A replica that waits in vain (its peers are not
// resident: another process fills the device) gives up after kXchgSpins polls, POISONS its slots -- tag field all ones
// under the launch number -- and finishes without exchanging; a replica that meets a poisoned granule does the same at
// once, so a failed launch costs one time-out, not one per exchange, and the host repeats the frame on one workgroup.
// First form of this exchange -- atomic stores, an arrival counter, polls, atomic loads, three barriers -- cost ~7 us per
// exchange; this one ~3.
constexpr int kXG = 16;
constexpr int kXchgSpins = 1 << 18;  // polls (each 16 loads in flight + a short sleep: ~0.5 s in all) before a replica gives up
typedef unsigned long long xq_t __attribute__((ext_vector_type(2)));
struct PoseXchg {
  xq_t cell[2][kXG][32];
};
__device__ __forceinline__ void xq_store(xq_t* p, xq_t v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
// the kXG replicas' granules of one index (512 bytes apart), all in flight at once
__device__ __forceinline__ void xq_load_all(const xq_t* p, xq_t* v) {
  static_assert(kXG == 16, "sixteen loads below");
  const xq_t* p2 = p + 32 * 8;  // (the instruction's offset field ends at 4095)
  asm volatile(
      "global_load_dwordx4 %0, %16, off sc0 sc1\n\t"
      "global_load_dwordx4 %1, %16, off offset:512 sc0 sc1\n\t"
      "global_load_dwordx4 %2, %16, off offset:1024 sc0 sc1\n\t"
      "global_load_dwordx4 %3, %16, off offset:1536 sc0 sc1\n\t"
      "global_load_dwordx4 %4, %16, off offset:2048 sc0 sc1\n\t"
      "global_load_dwordx4 %5, %16, off offset:2560 sc0 sc1\n\t"
      "global_load_dwordx4 %6, %16, off offset:3072 sc0 sc1\n\t"
      "global_load_dwordx4 %7, %16, off offset:3584 sc0 sc1\n\t"
      "global_load_dwordx4 %8, %17, off sc0 sc1\n\t"
      "global_load_dwordx4 %9, %17, off offset:512 sc0 sc1\n\t"
      "global_load_dwordx4 %10, %17, off offset:1024 sc0 sc1\n\t"
      "global_load_dwordx4 %11, %17, off offset:1536 sc0 sc1\n\t"
      "global_load_dwordx4 %12, %17, off offset:2048 sc0 sc1\n\t"
      "global_load_dwordx4 %13, %17, off offset:2560 sc0 sc1\n\t"
      "global_load_dwordx4 %14, %17, off offset:3072 sc0 sc1\n\t"
      "global_load_dwordx4 %15, %17, off offset:3584 sc0 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15])
      : "v"(p), "v"(p2)
      : "memory");
}
template <int BS, bool MC, bool ENC>
__global__ void __launch_bounds__(BS)
k_pose_opt_vio(const vieo_vio_frame* __restrict__ frames, const vieo_pose_obs* __restrict__ obs_all,
               uint8_t* __restrict__ outlier_all, vieo_vio_result* __restrict__ results, int other_launched,
               PoseXchg* __restrict__ xchg, unsigned launch_id) {
  __shared__ VioShared S;
  __shared__ double s_tr[BS == 64 ? 1 : 28 * (BS / 32) * 34];  // block_sum_lds' buffer (four wavefronts: 61 KB of its own)
  __shared__ double s_xf[MC ? 48 * (BS / 64) : 1];  // rig: every camera's Rcw | tcw at the estimate of the current pass, one
                                                    // copy per wavefront (formed by its own lanes: no workgroup barrier)
  __shared__ __align__(8) unsigned char s_cam_store[sizeof(CamD) * (MC ? 4 : 1)];  // CamD has initialisers
  CamD* s_cams = reinterpret_cast<CamD*>(s_cam_store);
  __shared__ __align__(8) unsigned char s_enc_store[ENC ? sizeof(VioEncShared) : 8];
  VioEncShared* SE = reinterpret_cast<VioEncShared*>(s_enc_store);
  // The observations of a one-camera frame of up to kVioSplitObs edges, staged once (24 KB): every pass over the visual
  // edges -- two per LM iteration, ~35 per call -- started with a trip to L2 for its first record and kept one more in
  // flight per edge; from LDS a record is two ds_read_b128.
  __shared__ __align__(16) vieo_pose_obs s_obs[(BS == 256 && !MC) ? 768 : 1];
  // ... and their level (1: an outlier of the last classification, not an active edge) as bytes beside the lanes' bit
  // masks: the error pass of a trial runs the edges over TWO wavefronts (the third then carries the prior edge alone and
  // is not the last to arrive), i.e. in another edge -> lane mapping than the passes that own the masks
  __shared__ uint8_t s_lvl[(BS == 256 && !MC) ? 768 : 4];
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T1 = BS > 64 ? 64 : 0, T2 = BS > 128 ? 128 : 0;  // lanes of the second / third serial role
  const int T3 = BS > 192 ? 192 : 0;  // fourth: the rotation rows of the inertial Jacobian (its two halves are independent)
  const vieo_vio_frame& F = frames[f];
  const int N = F.base.n_obs;
  // The visual edges belong to the first VT threads, edge i = tid + k VT <-> bit k of the thread's masks.  Four
  // wavefronts and up to kVioSplitObs edges (a tracked frame of one stereo pair): the first two wavefronts; the
  // single-lane edges (inertial, prior, bias, encoder) sit on lanes of the other two and are evaluated WHILE the visual
  // edges are -- separate instruction streams -- instead of after them.  Beyond that (rig frames: thousands of edges)
  // the visual passes are the longer part and take all four wavefronts; the single-lane edges then follow on theirs.
  // (VT is a compile-time constant of the optimisation below, which exists twice in the four-wavefront instances of
  // one-camera frames -- picked per frame by its number of edges; as a run-time stride it cost 7 % of the kernel.
  // Rig instances: rig_xf()'s barriers need every thread, and their frames are beyond the limit anyway.)
  constexpr int kVioSplitObs = 768;
  constexpr int kVioReplicaMinObs = 700;
  const vieo_pose_obs* obs = obs_all + F.base.obs_begin;
  uint8_t* outl = outlier_all + F.base.obs_begin;
  vieo_vio_result* R = results + f;
  const vieo_pose_enc* pe = F.base.enc;
  const bool cam_other = (F.base.n_cams > 0) != MC, enc_other = (pe != nullptr && pe->enc.dt != 0) != ENC;
  if ((cam_other && (other_launched & 1)) || (enc_other && (other_launched & 2))) return;
  const bool other_kind = cam_other || enc_other;
  bool bad_cams = other_kind;  // vieo_pose_set_camera_mode / _encoder_mode promised frames of the other kind only
  if (MC && !other_kind) {
    if (tid == 0) S.ok = !(F.base.n_cams > 4 || !F.base.cams);
    __syncthreads();
    if (tid < 4 && S.ok)
      if (!cam_from_abi(F.base.cams[tid < F.base.n_cams ? tid : 0], s_cams[tid])) S.ok = 0;
    __syncthreads();
    bad_cams = !S.ok;
    __syncthreads();
  }
  if (bad_cams || (N < 3 && !F.no_mps) || N > kVioMaxObs) {  // Optimizer.h:499-503
    for (int i = tid; i < N; i += BS) outl[i] = 0;
    for (int i = tid; i < 225; i += BS) R->H_marg[i] = 0;
    if (tid == 0) {
      R->base.nav = F.base.nav;
      R->base.n_inliers = 0;
      R->base.status = bad_cams ? VIEO_E_INVALID : N > kVioMaxObs ? VIEO_E_CAPACITY : VIEO_POSE_TOO_FEW;
      R->base.lm_iterations = 0;
      R->base.reserved = 0;
      R->has_marg = 0;
      R->reserved = 0;
    }
    return;
  }
  CamD c;
  c.fx = F.base.fx, c.fy = F.base.fy, c.cx = F.base.cx, c.cy = F.base.cy, c.bf = F.base.bf;
  for (int i = 0; i < 9; i++) c.Rcb[i] = F.base.Rcb[i];
  for (int i = 0; i < 3; i++) c.tcb[i] = F.base.tcb[i];
  const bool fixedLast = !F.last_has_prior, hasImu = F.imu.dt != 0;
  // rig frames: the cameras' transforms at the estimate of a pass, once (a lane per camera) instead of once per edge
  const int n = fixedLast ? 15 : 30;
  const bool bodom = hasImu || ENC;
  // ---- constant edge data
  if (tid == 0) {
    ns_load(S.nsj, F.base.nav);
    ns_load(S.nsi, F.nav_last);
    ns_load(S.prior, F.nav_prior);
  }
  if (ENC && tid == T2) vio_enc_setup(pe, SE);
  // IMU information = Sigma^-1 (x 1e-2 when the last state is fixed): Gauss-Jordan by one wave
  if (hasImu && wave == 0) {
    double* M = S.Cinv;  // 9 x 18 augmented
    for (int e = lane; e < 9 * 18; e += 64) {
      const int i = e / 18, j = e % 18;
      M[e] = j < 9 ? F.imu.Sigma[i * 9 + j] : (j - 9 == i ? 1.0 : 0.0);
    }
    wave_sync();
    for (int cidx = 0; cidx < 9; cidx++) {
      int piv = cidx;
      double best = fabs(M[cidx * 18 + cidx]);
      for (int r = cidx + 1; r < 9; r++)
        if (fabs(M[r * 18 + cidx]) > best) best = fabs(M[r * 18 + cidx]), piv = r;
      if (piv != cidx) {
        if (lane < 18) {
          const double t = M[cidx * 18 + lane];
          M[cidx * 18 + lane] = M[piv * 18 + lane];
          M[piv * 18 + lane] = t;
        }
        wave_sync();
      }
      const double d = M[cidx * 18 + cidx];
      wave_sync();
      if (lane < 18) M[cidx * 18 + lane] /= d;
      wave_sync();
      double fr[3];
      for (int h = 0; h < 3; h++) {
        const int e = lane + 64 * h;
        fr[h] = e < 162 ? M[(e / 18) * 18 + cidx] : 0;
      }
      wave_sync();
      for (int h = 0; h < 3; h++) {
        const int e = lane + 64 * h;
        if (e < 162 && e / 18 != cidx) M[e] -= fr[h] * M[cidx * 18 + e % 18];
      }
      wave_sync();
    }
    for (int e = lane; e < 81; e += 64) S.InfoI[e] = M[(e / 9) * 18 + 9 + e % 9] * (fixedLast ? 1e-2 : 1.0);
  }
  if (!fixedLast)
    for (int e = tid; e < 225; e += BS) S.Hp[e] = F.H_prior[e];
  if (tid < 3) S.gw[tid] = F.gw[tid];
  prior_jacobian_init(S.JP, tid, BS);
  if (hasImu) imu_jacobian_init(F.imu, S.JI, tid, BS);
  if (tid == BS - 1 && hasImu) {
    const Qd q = R_to_q(F.imu.Rij);
    S.qRij[0] = q.w, S.qRij[1] = q.x, S.qRij[2] = q.y, S.qRij[3] = q.z;
  }
  for (int e = tid; e < (int)(offsetof(vieo_imu_preint, Sigma) / 8); e += BS)
    reinterpret_cast<double*>(&S.imu)[e] = reinterpret_cast<const double*>(&F.imu)[e];
  __syncthreads();
  // the system's entries that no linearisation writes (right of the diagonal; see the assembly) stay zero from here on
  // (the information matrix's elimination above used this memory)
  if (BS == 256) {
    for (int i = tid; i < 900; i += BS) S.H[i] = 0.0;
    __syncthreads();
  }
  const double deltatij = F.imu.dt ? F.imu.dt : F.dt_frames;
  const double infoBg = F.inv_sigma_bg2 / deltatij * (fixedLast ? 1e-2 : 1.0);
  const double infoBa = F.inv_sigma_ba2 / deltatij * (fixedLast ? 1e-2 : 1.0);
  const double dI = sqrt(16.919), dB = sqrt(12.592), dP = sqrt(25.0), dE = sqrt(12.592);
  const double deltaMono = (double)(float)sqrt(5.991), deltaStereo = (double)(float)sqrt(7.815);
  const float chi2Mono = 5.991f, chi2Stereo = 7.815f;
  auto optimise = [&](auto vt_c, auto g_c) __attribute__((always_inline)) {
  constexpr int VT = decltype(vt_c)::value;
  constexpr int G = decltype(g_c)::value;  // replicas of the frame (1, or kXG on blockIdx.y)
  const int g = G > 1 ? (int)blockIdx.y : 0;
  const int Nv = tid < VT ? N : 0;  // (loop bound of the visual loops: no trip for the other threads)
  const int i0 = g * VT + tid;       // this thread's first visual edge; the next ones GV further
  constexpr int GV = G * VT;
  PoseXchg* xb = G > 1 ? xchg + f : nullptr;
  // the prior edge's lane: on the third wavefront (behind its visual edges when it carries some: on the fourth, behind
  // the inertial edge, it measured 4 % slower -- that wavefront is the last to arrive as it is)
  const int TPr = T2;
  constexpr bool kObsLds = BS == 256 && !MC && VT == 192 && G == 1;  // (N <= kVioSplitObs = the array's size)
  if constexpr (kObsLds) {
    static_assert(sizeof(vieo_pose_obs) == 32, "two 16-byte halves per record");
    const uint4* src = reinterpret_cast<const uint4*>(obs);
    uint4* dst = reinterpret_cast<uint4*>(s_obs);
    for (int i = tid; i < 2 * N; i += BS) dst[i] = src[i];
    for (int i = tid; i < N; i += BS) s_lvl[i] = 0;
    __syncthreads();
  }
  auto ld_obs = [&](int i) -> vieo_pose_obs {
    if constexpr (kObsLds)
      return s_obs[i];
    else
      return obs[i];
  };
  // Rig: the cameras' transforms at the estimate of a pass.  All threads on the visual edges: lanes 0 .. n_cams - 1 of
  // the workgroup form them between two barriers.  Visual edges on wavefronts 0-1 only (replicas): every wavefront
  // forms its own copy between wavefront barriers -- the other wavefronts are busy with the single-lane edges and cannot
  // come to a workgroup barrier here.  (The copy is named at every use: through a pointer variable the compiler loses
  // the LDS address space.)
  auto rig_xf = [&](const PoseXf& X, const double* p) {
    if (MC && VT == BS) {
      __syncthreads();  // the previous pass's readers are done
      if (tid < F.base.n_cams) rig_cam_xf(s_cams[tid], X, p, s_xf + 12 * tid);
      __syncthreads();
    } else if (MC) {
      wave_sync();
      if (lane < F.base.n_cams) rig_cam_xf(s_cams[lane], X, p, s_xf + 48 * wave + 12 * lane);
      wave_sync();
    }
  };
  int epoch = 0;
  if (G > 1) {
    if (tid == 0) S.xfail = 0;
    __syncthreads();
  }
  // v[0 .. n) in LDS, complete and visible (a barrier has passed) -> the sums over the replicas, in every replica
  auto grid_sum = [&](double* v, int n) {
    if (G == 1) return;
    if (S.xfail) return;  // (uniform: set before a barrier below) the launch has failed, the host repeats the frame
    if (tid < n) {
      const unsigned tag = ((launch_id & 0xFFFFFu) << 12) | (unsigned)((epoch + 1) & 0xFFF);
      const unsigned poison = ((launch_id & 0xFFFFFu) << 12) | 0xFFFu;  // (an exchange number never reaches 0xFFF)
      xq_t* cells = &xb->cell[epoch & 1][0][tid];  // replica q's granule: cells + 32 q
      const unsigned long long bits = (unsigned long long)__double_as_longlong(v[tid]);
      xq_t mine;
      mine.x = (bits & 0xFFFFFFFFull) | ((unsigned long long)tag << 32), mine.y = (bits >> 32) | ((unsigned long long)tag << 32);
      xq_store(cells + 32 * g, mine);
      xq_t in[kXG];
      int spins = 0;
      bool failed = false;
      for (;;) {
        xq_load_all(cells, in);
        bool all = true;
#pragma unroll
        for (int q = 0; q < kXG; q++) {
          const unsigned tx = (unsigned)(in[q].x >> 32), ty = (unsigned)(in[q].y >> 32);
          all = all && tx == tag && ty == tag;
          failed = failed || tx == poison || ty == poison;
        }
        if (all && !failed) break;
        if (failed || ++spins > kXchgSpins || epoch >= 0xFF0) {  // a peer gave up / is not coming: fail, do not hang
          failed = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (failed) {
        S.xfail = 1;
      } else {
        double t = 0;
#pragma unroll
        for (int q = 0; q < kXG; q++)  // replica order: the same sum everywhere
          t += __longlong_as_double((long long)((in[q].x & 0xFFFFFFFFull) | (in[q].y << 32)));
        v[tid] = t;
      }
    }
    epoch++;
    __syncthreads();
    if (S.xfail && tid < 64) {  // tell the peers: every slot of this replica, both parities
      const unsigned long long pz = (unsigned long long)(((launch_id & 0xFFFFFu) << 12) | 0xFFFu) << 32;
      xq_t pq;
      pq.x = pz, pq.y = pz;
      xq_store(&xb->cell[tid >> 5][g][tid & 31], pq);
    }
  };
  unsigned long long levelmask = 0;
  bool vis_robust = true;
  int nBad = 0, total_iters = 0, total_trials = 0;
  const int n_edges_total = N + (hasImu ? 1 : 0) + 1 + (fixedLast ? 0 : 1) + (ENC ? 1 : 0);
  double rhoE = 1.0;  // rho' of the encoder edge at the last all_errors()

#ifdef VIEO_POSE_PROBE
  unsigned long long pp_last = __builtin_amdgcn_s_memtime();
#endif
  // Errors of the edges at the LDS state.  The single-lane edges sit on lanes of the wavefronts that hold no visual
  // edge (four wavefronts: the inertial edge's rotation rows on the fourth, its position / velocity rows, the prior, the
  // bias and the encoder edge on the third) and are evaluated while the first two wavefronts go through the visual edges
  // (with_visual: a trial; without: the first linearisation of an optimize(), which sums the visual edges itself).
  // Returns the robust chi2 of the generic edges (and rho' of I, B, P); *vis = that of the active visual edges.
  constexpr int VTE = kObsLds ? 128 : VT;  // visual threads of the error pass
  auto all_errors = [&](double* rhoI, double* rhoB, double* rhoP, bool with_visual, double* vis) -> double {
    if (with_visual && tid < VTE) {  // (whole wavefronts)
      Est e;
      e.p[0] = S.nsj.p[0], e.p[1] = S.nsj.p[1], e.p[2] = S.nsj.p[2];
      e.qw = S.nsj.qw, e.qx = S.nsj.qx, e.qy = S.nsj.qy, e.qz = S.nsj.qz;
      PoseXf X;
      make_xf(c, e, X);
      rig_xf(X, e.p);
      double tc = 0;
      if constexpr (kObsLds) {  // straight-line form, two edges per block (see vis_error_flat)
        const int K = (N + VTE - 1) / VTE;  // slots per lane (uniform)
        auto slots = [&](int k0, auto u_c) __attribute__((always_inline)) {
          constexpr int U = decltype(u_c)::value;
          VisFlat E[U];
          bool valid[U];
#pragma unroll
          for (int u = 0; u < U; u++) {
            const int i = min(tid + (k0 + u) * VTE, N - 1);
            valid[u] = tid + (k0 + u) * VTE < N && !s_lvl[i];
            vis_error_flat(c, X, s_obs[i], valid[u], E[u]);
          }
#pragma unroll
          for (int u = 0; u < U; u++) {
            double r0, r1;
            const double dl = E[u].stereo ? deltaStereo : deltaMono;
            huber_flat(E[u].chi2, dl, dl * dl, vis_robust, &r0, &r1);
            tc += valid[u] ? r0 : 0.0;
          }
        };
        int k = 0;
        for (; k + 1 < K; k += 2) slots(k, std::integral_constant<int, 2>{});
        if (k < K) slots(k, std::integral_constant<int, 1>{});
      } else {
      vieo_pose_obs o_next = ld_obs(min(i0, N - 1));  // the next edge's record is in flight while this one is evaluated
      for (int k = 0, i = i0; i < Nv; k++, i += GV) {
        const vieo_pose_obs o = o_next;
        if (i + GV < N) o_next = ld_obs(i + GV);
        if ((levelmask >> k) & 1) continue;
        double err[3], Pc[3];
        const double chi2 = edge_eval<MC>(c, s_cams, X, e.p, o, err, Pc, nullptr, MC ? s_xf + (VT == BS ? 0 : 48 * wave) : nullptr);
        double r0 = chi2, r1 = 1.;
        if (vis_robust) {
          const double dl = o.ur >= 0 ? deltaStereo : deltaMono;
          huber(chi2, dl, dl * dl, &r0, &r1);
        }
        tc += r0;
      }
      }
      tc = BS == 64 ? wave_sum_d_bfly(tc) : wave_sum_d(tc);
      if (lane == 0) S.red[wave] = tc;  // (S.red's last readers are barriers away)
    }
    if (BS > 192) {
      // (three visual wavefronts: the third also carries the prior's lane, so both halves of the inertial edge sit on
      // the fourth; two visual wavefronts: its position / velocity rows go with the prior on the third)
      if (VT > 128) {  // one lane, one block: the scheduler interleaves the two halves' chains
        if (tid == T3 && hasImu) {
          RT(17, imu_error(S.imu, S.gw, S.nsi, S.nsj, S.errI, 6, 1);
             imu_error_rot(S.imu, Qd{S.qRij[0], S.qRij[1], S.qRij[2], S.qRij[3]}, S.nsi, S.nsj, S.errI, S.rc));
        }
      } else {
        if (tid == T3 && hasImu) RT(17, imu_error_rot(S.imu, Qd{S.qRij[0], S.qRij[1], S.qRij[2], S.qRij[3]}, S.nsi, S.nsj, S.errI, S.rc));
        if (tid == T2 && hasImu) RT(18, imu_error(S.imu, S.gw, S.nsi, S.nsj, S.errI, 6, 1));
      }
    } else if (tid == 0 && hasImu) {
      imu_error(S.imu, S.gw, S.nsi, S.nsj, S.errI, 6, 1);
      imu_error_rot(S.imu, Qd{S.qRij[0], S.qRij[1], S.qRij[2], S.qRij[3]}, S.nsi, S.nsj, S.errI, S.rc);
    }
    if (tid == TPr && !fixedLast) RT(19, prior_error(S.prior, S.nsi, S.errP, S.rc));
    if (tid == T2) {
      for (int k = 0; k < 3; k++) {
        S.errB[k] = (S.nsj.bg[k] + S.nsj.dbg[k]) - (S.nsi.bg[k] + S.nsi.dbg[k]);
        S.errB[3 + k] = (S.nsj.ba[k] + S.nsj.dba[k]) - (S.nsi.ba[k] + S.nsi.dba[k]);
      }
      if (ENC) vio_enc_eval(pe, SE, &S.nsi, &S.nsj, 0);
    }
    // Four wavefronts with both halves of the inertial edge on the fourth (kTail): the information products, the
    // quadratic forms and the robust kernels of the inertial / prior / bias edges are formed by their own wavefronts
    // right behind the errors -- the products lane-parallel, each form by one lane in the order of the loop further
    // down (same bits) -- instead of by every thread behind a second barrier.
    constexpr bool kTail = BS == 256 && VT == 192 && !ENC;
    if (kTail) {
      if (wave == 3 && hasImu) {
        wave_sync();
        if (lane < 9) {
          double t = 0;
#pragma unroll
          for (int j = 0; j < 9; j++) t += S.InfoI[lane * 9 + j] * S.errI[j];
          S.wI[lane] = t;
        }
        wave_sync();
        if (lane == 0) {
          double eI = 0;
#pragma unroll
          for (int i = 0; i < 9; i++) eI += S.errI[i] * S.wI[i];
          double r0 = eI, r1 = 1.0;
          if (fixedLast) huber(eI, dI, dI * dI, &r0, &r1);
          S.chiq[0] = r0, S.chiq[1] = r1;
        }
      }
      if (wave == 2) {
        wave_sync();
        if (!fixedLast && lane < 15) {
          double t = 0;
#pragma unroll
          for (int j = 0; j < 15; j++) t += S.Hp[lane * 15 + j] * S.errP[j];
          S.wP[lane] = t;
        }
        wave_sync();
        if (lane == 0) {
          double eB = 0;
#pragma unroll
          for (int i = 0; i < 3; i++) eB += S.errB[i] * (infoBg * S.errB[i]);
#pragma unroll
          for (int i = 3; i < 6; i++) eB += S.errB[i] * (infoBa * S.errB[i]);
          double r0 = eB, r1 = 1.0;
          if (fixedLast) huber(eB, dB, dB * dB, &r0, &r1);
          S.chiq[2] = r0, S.chiq[3] = r1;
          if (!fixedLast) {
            double eP = 0;
#pragma unroll
            for (int i = 0; i < 15; i++) eP += S.errP[i] * S.wP[i];
            huber(eP, dP, dP * dP, &r0, &r1);
            S.chiq[4] = r0, S.chiq[5] = r1;
          }
        }
      }
    }
    PP(13);
    __syncthreads();
    PP(14);
    if (kTail) {
      PP(15);
      double chi = 0;
      *rhoI = *rhoB = *rhoP = 1.0;
      if (hasImu) chi += S.chiq[0], *rhoI = S.chiq[1];
      chi += S.chiq[2], *rhoB = S.chiq[3];
      if (!fixedLast) chi += S.chiq[4], *rhoP = S.chiq[5];
      if (with_visual) {
        double v = S.red[0];
        for (int w = 1; w < VTE / 64; w++) v += S.red[w];
        *vis = v;
      }
      return chi;
    }
    if (G > 1 && with_visual) {  // the replicas' shares of the visual chi2
      if (tid == 0) {
        double v = S.red[0];
        for (int w = 1; w < VTE / 64; w++) v += S.red[w];
        S.xv[0] = v;
      }
      __syncthreads();
      grid_sum(S.xv, 1);
    }
    if (hasImu && tid < 9) {
      double t = 0;
      for (int j = 0; j < 9; j++) t += S.InfoI[tid * 9 + j] * S.errI[j];
      S.wI[tid] = t;
    }
    if (!fixedLast && tid >= T1 && tid < T1 + 15) {
      const int i = tid - T1;
      double t = 0;
      for (int j = 0; j < 15; j++) t += S.Hp[i * 15 + j] * S.errP[j];
      S.wP[i] = t;
    }
    __syncthreads();
    PP(15);
    double chi = 0;
    *rhoI = *rhoB = *rhoP = 1.0;
    // the three quadratic forms first (independent chains the scheduler can interleave), then the kernels
    double eI = 0, eB = 0, eP = 0;
    if (hasImu)
      for (int i = 0; i < 9; i++) eI += S.errI[i] * S.wI[i];
    for (int i = 0; i < 3; i++) eB += S.errB[i] * (infoBg * S.errB[i]);
    for (int i = 3; i < 6; i++) eB += S.errB[i] * (infoBa * S.errB[i]);
    if (!fixedLast)
      for (int i = 0; i < 15; i++) eP += S.errP[i] * S.wP[i];
    if (hasImu) {
      double r0 = eI;
      if (fixedLast) huber(eI, dI, dI * dI, &r0, rhoI);
      chi += r0;
    }
    {
      double r0 = eB;
      if (fixedLast) huber(eB, dB, dB * dB, &r0, rhoB);
      chi += r0;
    }
    if (!fixedLast) {
      double r0 = eP;
      huber(eP, dP, dP * dP, &r0, rhoP);
      chi += r0;
    }
    if (ENC) {
      double r0;
      huber(SE->chi, dE, dE * dE, &r0, &rhoE);
      chi += r0;
    }
    if (with_visual) {
      double v = S.red[0];
      for (int w = 1; w < VTE / 64; w++) v += S.red[w];
      *vis = G > 1 ? S.xv[0] : v;
    }
    return chi;
  };

  for (int it = 0; it < 4; it++) {
    __syncthreads();
    if (!bodom && tid == 0) {  // Optimizer.h:538-545
      ns_load(S.nsj, F.base.nav);
      if (!fixedLast) ns_load(S.nsi, F.nav_last);
    }
    __syncthreads();
    double lambda = -1, ni = 2;
    int nBadLM = 0;
    // an iteration after the first starts at the state its predecessor's last trial was ACCEPTED at (a rejected last
    // trial ends the optimize()), and that trial evaluated the generic edges there: errors, weights and rho' are still
    // in LDS / registers, computing them again gives the same bits
    double accChi = 0, accI = 1, accB = 1, accP = 1, accE = 1;
    for (int iter = 0; iter < 10; iter++) {
      total_iters++;
      // ---- computeActiveErrors + activeRobustChi2 + buildSystem
      double rhoI, rhoB, rhoP;
      PP(0);
      double chiG;
      if (iter == 0)
        chiG = all_errors(&rhoI, &rhoB, &rhoP, false, nullptr);
      else
        chiG = accChi, rhoI = accI, rhoB = accB, rhoP = accP, rhoE = accE;
      PP(1);
      double acc[28];
#pragma unroll
      for (int i = 0; i < 28; i++) acc[i] = 0;
      if (tid < VT) {  // (whole wavefronts)
      Est e;
      e.p[0] = S.nsj.p[0], e.p[1] = S.nsj.p[1], e.p[2] = S.nsj.p[2];
      e.qw = S.nsj.qw, e.qx = S.nsj.qx, e.qy = S.nsj.qy, e.qz = S.nsj.qz;
      PoseXf X;
      make_xf(c, e, X);
      rig_xf(X, e.p);
      if constexpr (kObsLds) {  // straight-line form, two edges per block (see vis_error_flat)
        const int K = (N + VT - 1) / VT;  // slots per lane (uniform)
        auto slots = [&](int k0, auto u_c) __attribute__((always_inline)) {
          constexpr int U = decltype(u_c)::value;
          VisFlat E[U];
          bool valid[U];
          double J[U][18];
#pragma unroll
          for (int u = 0; u < U; u++) {
            const int i = tid + (k0 + u) * VT;
            valid[u] = i < N && !((levelmask >> (k0 + u)) & 1);
            const vieo_pose_obs o = s_obs[min(i, N - 1)];
            vis_error_flat(c, X, o, valid[u], E[u]);
            vis_jacobian_flat(c, X, e.p, o, E[u], J[u]);
          }
#pragma unroll
          for (int u = 0; u < U; u++) {
            double r0, r1;
            const double dl = E[u].stereo ? deltaStereo : deltaMono;
            huber_flat(E[u].chi2, dl, dl * dl, vis_robust, &r0, &r1);
            acc[27] += valid[u] ? r0 : 0.0;
            vis_accumulate_flat(J[u], E[u].err, valid[u] ? E[u].info : 0.0, r1, acc);
          }
        };
        int k = 0;
        for (; k + 1 < K; k += 2) slots(k, std::integral_constant<int, 2>{});
        if (k < K) slots(k, std::integral_constant<int, 1>{});
      } else {
      vieo_pose_obs o_next = ld_obs(min(i0, N - 1));  // the next edge's record is in flight while this one is evaluated
      for (int k = 0, i = i0; i < Nv; k++, i += GV) {
        const vieo_pose_obs o = o_next;
        if (i + GV < N) o_next = ld_obs(i + GV);
        if ((levelmask >> k) & 1) continue;
        double err[3], Pc[3];
        double J[18];
        const double chi2 = edge_eval<MC>(c, s_cams, X, e.p, o, err, Pc, J, MC ? s_xf + (VT == BS ? 0 : 48 * wave) : nullptr);
        const bool stereo = o.ur >= 0;
        double r0 = chi2, r1 = 1.;
        if (vis_robust) {
          const double dl = stereo ? deltaStereo : deltaMono;
          huber(chi2, dl, dl * dl, &r0, &r1);
        }
        acc[27] += r0;
        visual_accumulate(J, err, (double)o.inv_sigma2, r1, stereo, acc);
      }
      }
      }
      // the single-lane Jacobians, on the wavefronts without visual edges, meanwhile
      // (round 6) ... and what follows from them alone: the inertial edge's (rho' Info) J and J^T T -- i.e. every entry of
      // the system before the other edges add to it -- on the inertial edge's wavefront, the prior's (rho' H_prior) J and
      // J^T T' (parked in the solver's column buffer, dead between two solves) on the prior's wavefront, both WHILE
      // wavefronts 0-1 go through the visual edges and their sums.  What is left behind the sums' barrier is one phase of
      // additions (C below).  The entries and the order of their terms are those of the three-phase form (kept for the
      // one-wavefront instances, where the phases are the same wavefront anyway): same bits.
      constexpr bool kOverlap = BS == 256;
      if (BS > 192) {
        if (wave == 3) {
          if (hasImu) {  // lane 0 writes the inertial Jacobian's state-dependent blocks (the others are constants of the call)
            if (lane == 0) {
              RT(20, imu_linearize_pv(S.imu, S.gw, S.nsi, S.nsj, S.JI);
                 imu_linearize_rot(S.imu, S.errI, S.JI, S.rc));
            }
            wave_sync();
          }
          // The inertial edge's part of the system, J^T (rho' Info) J and its gradient, on the FP64 matrix cores of this
          // wavefront (v_mfma_f64_16x16x4_f64; lane l supplies A[l & 15][4 ks + (l >> 4)] and B[4 ks + (l >> 4)][l & 15],
          // accumulator register r is D[(l >> 4) + 4 r][l & 15]):
          //   T = (rho' Info) J, 9 x 24 as one 16-row tile by two 16-column tiles, K = 9 in three steps of 4 -- its
          //   accumulator registers r = 0..2 are, as they stand, the B operands (rows 4 r + (l >> 4)) of
          //   H = J^T T, whose A operands are the J values already loaded as B operands of the first product;
          //   column 24 of T (padding) carries -rho' w, so column 24 of H is the gradient J^T (-rho' Info e).
          // The solver reads the diagonal and the entries below it only, so tile (0, 1) serves the gradient alone.
          // Entries receive the sum of their nine terms in the matrix core's order instead of the scalar loop's: same
          // value up to the last bits (parity bar 1e-4).  The rest of the system is zero-filled first.
          if (!hasImu) {  // no inertial edge: nothing overwrites the entries the other edges add to
            for (int i = lane; i < n * n; i += 64) S.H[i] = 0.0;
            if (lane < n) S.b[lane] = 0.0;
          }
          if (hasImu) {
            typedef double v4d __attribute__((ext_vector_type(4)));
            const int l15 = lane & 15, l4 = lane >> 4;
            double aI[3], bJ[2][3];
#pragma unroll
            for (int ks = 0; ks < 3; ks++) {
              const int k = 4 * ks + l4;
              aI[ks] = (l15 < 9 && k < 9) ? rhoI * S.InfoI[l15 * 9 + k] : 0.0;
              bJ[0][ks] = k < 9 ? S.JI[k * 24 + l15] : 0.0;
              bJ[1][ks] = (k < 9 && l15 < 8) ? S.JI[k * 24 + 16 + l15] : 0.0;
            }
            v4d T0 = {0, 0, 0, 0}, T1 = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 3; ks++) {
              T0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aI[ks], bJ[0][ks], T0, 0, 0, 0);
              T1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aI[ks], bJ[1][ks], T1, 0, 0, 0);
            }
            if (l15 == 8) {  // column 24: -rho' w_k, k = 4 r + (l >> 4)
#pragma unroll
              for (int r = 0; r < 3; r++) {
                const int k = 4 * r + l4;
                T1[r] = k < 9 ? -S.wI[k] * rhoI : 0.0;
              }
            }
            v4d H00 = {0, 0, 0, 0}, H01 = {0, 0, 0, 0}, H10 = {0, 0, 0, 0}, H11 = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 3; ks++) {
              H00 = __builtin_amdgcn_mfma_f64_16x16x4f64(bJ[0][ks], T0[ks], H00, 0, 0, 0);
              H01 = __builtin_amdgcn_mfma_f64_16x16x4f64(bJ[0][ks], T1[ks], H01, 0, 0, 0);
              H10 = __builtin_amdgcn_mfma_f64_16x16x4f64(bJ[1][ks], T0[ks], H10, 0, 0, 0);
              H11 = __builtin_amdgcn_mfma_f64_16x16x4f64(bJ[1][ks], T1[ks], H11, 0, 0, 0);
            }
            const int m = fixedLast ? 9 : 24;  // reduced unknowns of the inertial edge
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int row = l4 + 4 * r;
              // tiles (0, 0), (1, 0), (1, 1): reduced (c1, c2) with c2 <= c1; tiles (0, 1), (1, 1) column 8: the gradient
              {
                const int c1 = row, c2 = l15;
                if (c1 < m && c2 <= c1) S.H[(c1 < 9 ? c1 : c1 + 6) * n + (c2 < 9 ? c2 : c2 + 6)] = 0.0 + H00[r];
                if (c1 < m && l15 == 8) S.b[c1 < 9 ? c1 : c1 + 6] = 0.0 + H01[r];
              }
              if (!fixedLast) {
                const int c1 = 16 + row;
                if (c1 < 24) {
                  S.H[(c1 + 6) * n + (l15 < 9 ? l15 : l15 + 6)] = 0.0 + H10[r];
                  if (l15 < 8 && 16 + l15 <= c1) S.H[(c1 + 6) * n + 16 + l15 + 6] = 0.0 + H11[r];
                  if (l15 == 8) S.b[c1 + 6] = 0.0 + H11[r];
                }
              }
            }
          }
        }
      } else if (hasImu) {  // one wavefront per frame: it clears the Jacobian, lane 0 fills it
        for (int i = lane; i < 9 * 24; i += 64) S.JI[i] = 0;
        wave_sync();
        if (tid == 0) {
          imu_linearize(S.imu, S.gw, S.nsi, S.nsj, S.errI, S.JI, 6, 3, 1);
          imu_linearize_rot(S.imu, S.errI, S.JI, S.rc);
        }
      }
      if (tid == TPr && !fixedLast) RT(22, prior_linearize(S.errP, S.JP, S.rc));
      if (kOverlap && wave == (TPr >> 6) && !fixedLast) {
        wave_sync();
        // the prior's J^T (rho' H_prior) J and gradient the same way: one 16 x 16 tile, K = 15 in four steps; column 15 of
        // T' (padding) carries -rho' w'.  Parked in L[0 .. 225) and L[225 .. 240).
        typedef double v4d __attribute__((ext_vector_type(4)));
        const int l15 = lane & 15, l4 = lane >> 4;
        double aP[4], bP[4];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
          const int k = 4 * ks + l4;
          aP[ks] = (l15 < 15 && k < 15) ? rhoP * S.Hp[l15 * 15 + k] : 0.0;
          bP[ks] = (l15 < 15 && k < 15) ? S.JP[k * 15 + l15] : 0.0;
        }
        v4d TPv = {0, 0, 0, 0}, HPv = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 4; ks++) TPv = __builtin_amdgcn_mfma_f64_16x16x4f64(aP[ks], bP[ks], TPv, 0, 0, 0);
        if (l15 == 15) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int k = 4 * r + l4;
            TPv[r] = k < 15 ? -S.wP[k] * rhoP : 0.0;
          }
        }
#pragma unroll
        for (int ks = 0; ks < 4; ks++) HPv = __builtin_amdgcn_mfma_f64_16x16x4f64(bP[ks], TPv[ks], HPv, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = l4 + 4 * r;
          if (row < 15) S.L[l15 < 15 ? row * 15 + l15 : 225 + row] = HPv[r];
        }
      }
      if (ENC && tid == T2) vio_enc_eval(pe, SE, &S.nsi, &S.nsj, 1);
      PP(2);
      // the 28 sums land in S.vis (the system's visual block) straight from the transpose; H .. tr_tail are dead here
      // (the four-wavefront instances transpose through s_tr: H is being written meanwhile).
      // Its barriers are also where the Jacobians above meet the threads that assemble the system.
      block_sum_lds<28, BS>(acc, BS == 64 ? S.H : s_tr, S.vis, tid);
      grid_sum(S.vis, 28);
      PP(3);
      double currentChi = chiG + S.vis[27];
      const double iniChi = currentChi;
      const double rhoE0 = rhoE;  // the trial evaluations below overwrite rhoE
      PP(4);
      // One-wavefront instances: three more barriers (were seven): (A) the products (rho' Info) J of the inertial, prior and
      // encoder edges side by side;
      // (B) every entry of the system is WRITTEN once -- J^T T of the inertial edge over its 24 (9) unknowns, zero on
      // the rows and columns of the current bias -- so nothing is cleared first; (C) the visual block, the prior and
      // the bias edge add to entries no other role of this phase touches (the one shared diagonal, prior + bias of the
      // last state, stays with the prior's thread).  Each entry receives its terms in the order inertial, visual |
      // prior, bias, i.e. the sums are those of the one-edge-after-the-other form, bit for bit.
      // Four-wavefront instances: (A) and (B) are done (above); only the encoder edge's product is formed here.
      if (!kOverlap && hasImu) {  // T = (rho' Info) J  (9 x 24)
        const int cnt = fixedLast ? 81 : 216;
        for (int eidx = tid; eidx < cnt; eidx += BS) {
          const int a = fixedLast ? eidx / 9 : eidx / 24, cc = eidx - a * (fixedLast ? 9 : 24);
          double t = 0;
#pragma unroll
          for (int q = 0; q < 9; q++) t += (rhoI * S.InfoI[a * 9 + q]) * S.JI[q * 24 + cc];
          S.T[a * 24 + cc] = t;
        }
      }
      if (!kOverlap && !fixedLast)  // T' = (rho' H_prior) J (15 x 15)
        for (int eidx = tid; eidx < 225; eidx += BS) {
          const int a = eidx / 15, cc = eidx % 15;
          double t = 0;
#pragma unroll
          for (int q = 0; q < 15; q++) t += (rhoP * S.Hp[a * 15 + q]) * S.JP[q * 15 + cc];
          S.TP[a * 15 + cc] = t;
        }
      if (ENC)  // T'' = (rho' Info) [Jj | Ji]
        for (int eidx = tid; eidx < 72; eidx += BS) {
          const int a = eidx / 12, cc = eidx % 12;
          double t = 0;
          for (int q = 0; q < 6; q++) t += (rhoE0 * SE->Info[a * 6 + q]) * SE->J[q * 12 + cc];
          SE->T[eidx] = t;
        }
      if (!kOverlap) {
        __syncthreads();
        for (int i = tid; i < n * n; i += BS) {
          const int s1 = fixedLast ? i / 15 : i / 30, s2 = i - s1 * n;
          const bool bias = (s1 >= 9 && s1 < 15) || (s2 >= 9 && s2 < 15);
          double t = 0;
          if (hasImu && !bias) {
            const int c1 = s1 < 9 ? s1 : s1 - 6, c2 = s2 < 9 ? s2 : s2 - 6;
#pragma unroll
            for (int a = 0; a < 9; a++) t += S.JI[a * 24 + c1] * S.T[a * 24 + c2];
          }
          S.H[i] = 0.0 + t;  // (0 + t: the sign of a zero sum as before)
        }
        if (tid < n) {
          const bool bias = tid >= 9 && tid < 15;
          double t = 0;
          if (hasImu && !bias) {
            const int c = tid < 9 ? tid : tid - 6;
#pragma unroll
            for (int a = 0; a < 9; a++) t += S.JI[a * 24 + c] * (-S.wI[a] * rhoI);
          }
          S.b[tid] = 0.0 + t;
        }
        __syncthreads();
      }
      // visual block: (dp, dphi) -> system rows/cols {0,1,2,6,7,8}
      if (tid < 36) {
        const int a = tid / 6, bq = tid % 6;
        const int lo = a < bq ? a : bq, hi = a < bq ? bq : a;
        const int t = lo * 6 - lo * (lo - 1) / 2 + (hi - lo);
        const int ra = a < 3 ? a : a + 3, rb = bq < 3 ? bq : bq + 3;
        if (!kOverlap || ra >= rb) S.H[ra * n + rb] += S.vis[t];  // (the solver reads the diagonal and below)
      }
      if (tid >= T1 && tid < T1 + 6) {
        const int a = tid - T1;
        S.b[a < 3 ? a : a + 3] += S.vis[21 + a];
      }
      if (!fixedLast) {  // prior: H[15.., 15..] += J^T T', with the bias edge's weight on the last state's bias diagonal
        for (int eidx = tid; eidx < 225; eidx += BS) {
          const int c1 = eidx / 15, c2 = eidx % 15;
          double t = 0;
          if (kOverlap)
            t = S.L[eidx];
          else {
#pragma unroll
            for (int a = 0; a < 15; a++) t += S.JP[a * 15 + c1] * S.TP[a * 15 + c2];
          }
          if (kOverlap && c1 < c2) continue;
          double h = S.H[(15 + c1) * n + 15 + c2] + t;
          if (c1 == c2 && c1 >= 9) h += (c1 < 12 ? infoBg : infoBa) * rhoB;
          S.H[(15 + c1) * n + 15 + c2] = h;
        }
        if (tid < 15) {
          double t = 0;
          if (kOverlap)
            t = S.L[225 + tid];
          else {
#pragma unroll
            for (int a = 0; a < 15; a++) t += S.JP[a * 15 + tid] * (-S.wP[a] * rhoP);
          }
          double h = S.b[15 + tid] + t;
          if (tid >= 9) h += (tid < 12 ? infoBg : infoBa) * S.errB[tid - 9] * rhoB;
          S.b[15 + tid] = h;
        }
      }
      if (tid >= T2 && tid < T2 + 6) {  // bias edge: J_j = +I (cols 9..14), J_i = -I (cols 24..29)
        const int k = tid - T2;
        const double w = (k < 3 ? infoBg : infoBa) * rhoB;
        const double we = (k < 3 ? infoBg : infoBa) * S.errB[k] * rhoB;
        // (this thread is the only writer of these entries: rows / columns of the current bias)
        S.H[(9 + k) * n + 9 + k] = 0.0 + w;
        S.b[9 + k] = 0.0 + -we;
        if (!fixedLast) {
          S.H[(9 + k) * n + 24 + k] = 0.0 - w;
          S.H[(24 + k) * n + 9 + k] = 0.0 - w;
        }
      }
      __syncthreads();
      if (ENC) {  // H += J^T T'' on the (p, phi) rows of the PVR vertices
        const int nc = fixedLast ? 6 : 12;
        for (int eidx = tid; eidx < nc * nc; eidx += BS) {
          const int c1 = eidx / nc, c2 = eidx % nc;
          double t = 0;
          for (int a = 0; a < 6; a++) t += SE->J[a * 12 + c1] * SE->T[a * 12 + c2];
          const int s1 = (c1 < 6 ? 0 : 9) + c1 + (c1 % 6 < 3 ? 0 : 3), s2 = (c2 < 6 ? 0 : 9) + c2 + (c2 % 6 < 3 ? 0 : 3);
          if (!kOverlap || s1 >= s2) S.H[s1 * n + s2] += t;
        }
        if (tid < nc) {
          double t = 0;
          for (int a = 0; a < 6; a++) t += SE->J[a * 12 + tid] * (-SE->we[a] * rhoE0);
          S.b[(tid < 6 ? 0 : 9) + tid + (tid % 6 < 3 ? 0 : 3)] += t;
        }
        __syncthreads();
      }
      PP(5);
      if (iter == 0) {
        double mx = 0;
        for (int j = 0; j < n; j++) mx = fmax(fabs(S.H[j * n + j]), mx);
        lambda = 1e-5 * mx;
        ni = 2;
        nBadLM = 0;
      }
      double rho = 0;
      int qmax = 0;
      do {
        __syncthreads();
        // the two states' backup (a lane per double) on the last wavefront, beside the factorisation on the first
        if (tid >= BS - 64) {
          constexpr int kNs = (int)(sizeof(NSd) / 8);
          if (lane < kNs) ((double*)&S.bkj)[lane] = ((const double*)&S.nsj)[lane];
          if (lane < kNs && BS == 64) ((double*)&S.bki)[lane] = ((const double*)&S.nsi)[lane];
          if (BS > 64 && lane >= 32 && lane < 32 + kNs) ((double*)&S.bki)[lane - 32] = ((const double*)&S.nsi)[lane - 32];
        }
        if (BS == 64) wave_sync();
        PP(11);
        if (wave == 0) {
          const bool ok = n == 15 ? wave_solve_vio<9>((const lds_f64*)S.H, n, lambda, (const lds_f64*)S.b, (lds_f64*)S.x, (lds_f64*)S.L, lane)
                                  : wave_solve_vio<24>((const lds_f64*)S.H, n, lambda, (const lds_f64*)S.b, (lds_f64*)S.x, (lds_f64*)S.L, lane);
          if (lane == 0) S.ok = ok ? 1 : 0;
        }
        PP(12);
        __syncthreads();
        PP(6);
        const bool ok2 = S.ok != 0;
        if (!ok2) {  // (uniform)
          if (tid < n) S.x[tid] = 0;
          __syncthreads();
        }
        // the two states are retracted side by side on different wavefronts (one lane each)
        if (tid == 0) ns_inc_unit(S.nsj, S.x, S.x + 9);
        if (tid == T1 && !fixedLast) ns_inc_unit(S.nsi, S.x + 15, S.x + 24);
        if (tid == T2) {  // the gain ratio's denominator, beside the retractions (it was every thread's loop behind the trial)
          double sc = 0;  // (all loads up front: trip count known per branch)
          if (fixedLast) {
#pragma unroll
            for (int j = 0; j < 15; j++) sc += S.x[j] * (lambda * S.x[j] + S.b[j]);
          } else {
#pragma unroll
            for (int j = 0; j < 30; j++) sc += S.x[j] * (lambda * S.x[j] + S.b[j]);
          }
          S.chiq[6] = sc;
        }
        __syncthreads();
        PP(7);
        double r1, r2, r3, visChi;
        const double tempChiG = all_errors(&r1, &r2, &r3, true, &visChi);
        double tempChi = tempChiG + visChi;
        PP(9);
        if (!ok2) tempChi = DBL_MAX;
        rho = currentChi - tempChi;
        double scale = S.chiq[6];
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && isfinite(tempChi)) {
          const double r21 = 2 * rho - 1;
          double alpha = 1. - r21 * r21 * r21;  // (pow(., 3) of the reference: the last bit of lambda may differ)
          alpha = fmin(alpha, 2. / 3.);
          lambda *= fmax(1. / 3., alpha);
          ni = 2;
          currentChi = tempChi;
          accChi = tempChiG, accI = r1, accB = r2, accP = r3, accE = rhoE;
        } else {
          lambda *= ni;
          ni *= 2;
          __syncthreads();
          {
            constexpr int kNs = (int)(sizeof(NSd) / 8);
            if (tid < kNs) ((double*)&S.nsj)[tid] = ((const double*)&S.bkj)[tid];
            if (tid >= 32 && tid < 32 + kNs) ((double*)&S.nsi)[tid - 32] = ((const double*)&S.bki)[tid - 32];
          }
        }
        qmax++;
        total_trials++;
      } while (rho < 0 && qmax < 10);
      __syncthreads();
      if (qmax == 10 || rho == 0) break;
      if ((iniChi - currentChi) * 1e3 < iniChi)
        nBadLM++;
      else
        nBadLM = 0;
      if (nBadLM >= 3) break;
    }
    PP(10);
    // ---- classification at the current estimate (Optimizer.h:554-611)
    __syncthreads();
    Est e;
    e.p[0] = S.nsj.p[0], e.p[1] = S.nsj.p[1], e.p[2] = S.nsj.p[2];
    e.qw = S.nsj.qw, e.qx = S.nsj.qx, e.qy = S.nsj.qy, e.qz = S.nsj.qz;
    PoseXf X;
    make_xf(c, e, X);
    rig_xf(X, e.p);
    const float chi2close = (float)(1.5 * (double)chi2Mono);
    double nb[1] = {0};
    for (int k = 0, i = i0; i < Nv; k++, i += GV) {
      const vieo_pose_obs o = ld_obs(i);
      double err[3], Pc[3];
      const float chi2 = (float)edge_eval<MC>(c, s_cams, X, e.p, o, err, Pc, nullptr, MC ? s_xf + (VT == BS ? 0 : 48 * wave) : nullptr);
      bool bad;
      if (o.ur < 0)
        bad = chi2 > ((o.flags & 1) ? chi2close : chi2Mono) || !(Pc[2] > 0.);
      else
        bad = chi2 > chi2Stereo;
      if (bad) {
        levelmask |= (1ull << k);
        nb[0] += 1;
      } else
        levelmask &= ~(1ull << k);
      if constexpr (kObsLds) s_lvl[i] = bad ? 1 : 0;
    }
    block_sum_bs<1, BS>(nb, S.red, tid);
    if (G > 1) {
      __syncthreads();  // (the last trial's readers of xv)
      if (tid == 0) S.xv[0] = nb[0];
      __syncthreads();
      grid_sum(S.xv, 1);
      nb[0] = S.xv[0];
    }
    nBad = (int)nb[0];
    if (it == 2) vis_robust = false;
    if (n_edges_total < 10) break;
  }
  unsigned long long outmask = levelmask;  // mvbOutlier
  if (N - nBad < 30) {           // rescue pass, Optimizer.h:621-648
    Est e;
    e.p[0] = S.nsj.p[0], e.p[1] = S.nsj.p[1], e.p[2] = S.nsj.p[2];
    e.qw = S.nsj.qw, e.qx = S.nsj.qx, e.qy = S.nsj.qy, e.qz = S.nsj.qz;
    PoseXf X;
    make_xf(c, e, X);
    rig_xf(X, e.p);
    double nb[1] = {0};
    for (int k = 0, i = i0; i < Nv; k++, i += GV) {
      const vieo_pose_obs o = ld_obs(i);
      double err[3], Pc[3];
      const double chi2 = edge_eval<MC>(c, s_cams, X, e.p, o, err, Pc, nullptr, MC ? s_xf + (VT == BS ? 0 : 48 * wave) : nullptr);
      if (chi2 < (double)(o.ur < 0 ? 18.f : 24.f)) {
        levelmask &= ~(1ull << k);
        outmask &= ~(1ull << k);
        if constexpr (kObsLds) s_lvl[i] = 0;
      } else
        nb[0] += 1;
    }
    block_sum_bs<1, BS>(nb, S.red, tid);
    if (G > 1) {
      __syncthreads();  // (the last trial's readers of xv)
      if (tid == 0) S.xv[0] = nb[0];
      __syncthreads();
      grid_sum(S.xv, 1);
      nb[0] = S.xv[0];
    }
    nBad = (int)nb[0];
  }
  for (int k = 0, i = i0; i < Nv; k++, i += GV) outl[i] = (outmask >> k) & 1;
  // ---- marginal prior (Optimizer.h:663-813, FillCovInv :126-206, exact_mode = kExactRobust)
  if (F.compute_marg) {
    double rhoI, rhoB, rhoP;
    all_errors(&rhoI, &rhoB, &rhoP, false, nullptr);
    Est e;
    e.p[0] = S.nsj.p[0], e.p[1] = S.nsj.p[1], e.p[2] = S.nsj.p[2];
    e.qw = S.nsj.qw, e.qx = S.nsj.qx, e.qy = S.nsj.qy, e.qz = S.nsj.qz;
    PoseXf X;
    make_xf(c, e, X);
    rig_xf(X, e.p);
    double acc[27];
#pragma unroll
    for (int i = 0; i < 27; i++) acc[i] = 0;
    vieo_pose_obs o_next = ld_obs(min(i0, N - 1));  // the next edge's record is in flight while this one is evaluated
    for (int k = 0, i = i0; i < Nv; k++, i += GV) {
      const vieo_pose_obs o = o_next;
      if (i + GV < N) o_next = ld_obs(i + GV);
      if ((levelmask >> k) & 1) continue;
      double err[3], Pc[3];
      double J[18];
      const double chi2 = edge_eval<MC>(c, s_cams, X, e.p, o, err, Pc, J, MC ? s_xf + (VT == BS ? 0 : 48 * wave) : nullptr);
      const bool stereo = o.ur >= 0;
      double r0 = chi2, r1 = 1.;
      if (vis_robust) {
        const double dl = stereo ? deltaStereo : deltaMono;
        huber(chi2, dl, dl * dl, &r0, &r1);
      }
      visual_accumulate(J, err, (double)o.inv_sigma2, r1, stereo, acc);
    }
    block_sum_bs<27, BS>(acc, S.red, tid);
    for (int i = tid; i < 9 * 24; i += BS) S.JI[i] = 0;
    __syncthreads();
    if (tid < 27) {
      double v = 0;
#pragma unroll
      for (int t = 0; t < 27; t++)
        if (tid == t) v = acc[t];
      S.vis[tid] = v;
    }
    if (BS > 192) {
      if (tid == 0 && hasImu) imu_linearize(S.imu, S.gw, S.nsi, S.nsj, S.errI, S.JI, 6, 3, 1);
      if (tid == T3 && hasImu) imu_linearize_rot(S.imu, S.errI, S.JI, S.rc);
    } else if (tid == 0 && hasImu) {
      imu_linearize(S.imu, S.gw, S.nsi, S.nsj, S.errI, S.JI, 6, 3, 1);
      imu_linearize_rot(S.imu, S.errI, S.JI, S.rc);
    }
    if (tid == T1 && !fixedLast) prior_linearize(S.errP, S.JP, S.rc);
    if (ENC && tid == T2) vio_enc_eval(pe, SE, &S.nsi, &S.nsj, 1);
    for (int i = tid; i < 225; i += BS) S.cov[i] = 0, S.C[i] = 0, S.E[i] = 0;
    __syncthreads();
    grid_sum(S.vis, 27);
    if (tid < 36) {
      const int a = tid / 6, bq = tid % 6;
      const int lo = a < bq ? a : bq, hi = a < bq ? bq : a;
      const int t = lo * 6 - lo * (lo - 1) / 2 + (hi - lo);
      S.cov[(a < 3 ? a : a + 3) * 15 + (bq < 3 ? bq : bq + 3)] = S.vis[t];
    }
    __syncthreads();
    if (hasImu) {
      for (int eidx = tid; eidx < 9 * 24; eidx += BS) {
        const int a = eidx / 24, cc = eidx % 24;
        double t = 0;
        for (int q = 0; q < 9; q++) t += (rhoI * S.InfoI[a * 9 + q]) * S.JI[q * 24 + cc];
        S.T[a * 24 + cc] = t;
      }
      __syncthreads();
      for (int eidx = tid; eidx < 24 * 24; eidx += BS) {
        const int c1 = eidx / 24, c2 = eidx % 24;
        double t = 0;
        for (int a = 0; a < 9; a++) t += S.JI[a * 24 + c1] * S.T[a * 24 + c2];
        if (c1 < 9 && c2 < 9) S.cov[c1 * 15 + c2] += t;                         // B: (cur, cur)
        if (c1 >= 9 && c2 >= 9) S.C[(c1 - 9) * 15 + (c2 - 9)] += t;             // C: (last, last)
        if (c1 < 9 && c2 >= 9) S.E[c1 * 15 + (c2 - 9)] += t;                    // E: (cur, last)
      }
      __syncthreads();
    }
    if (tid < 6) {
      const double w = (tid < 3 ? infoBg : infoBa) * rhoB;
      S.cov[(9 + tid) * 15 + 9 + tid] = w;
      S.C[(9 + tid) * 15 + 9 + tid] += w;
      S.E[(9 + tid) * 15 + 9 + tid] = -w;
    }
    __syncthreads();
    if (ENC) {  // getHessianXj -> B, Xi -> C, Xji -> E (FillCovInv :195-204)
      for (int eidx = tid; eidx < 72; eidx += BS) {
        const int a = eidx / 12, cc = eidx % 12;
        double t = 0;
        for (int q = 0; q < 6; q++) t += (rhoE * SE->Info[a * 6 + q]) * SE->J[q * 12 + cc];
        SE->T[eidx] = t;
      }
      __syncthreads();
      for (int eidx = tid; eidx < 144; eidx += BS) {
        const int c1 = eidx / 12, c2 = eidx % 12;
        double t = 0;
        for (int a = 0; a < 6; a++) t += SE->J[a * 12 + c1] * SE->T[a * 12 + c2];
        const int k1 = c1 % 6, k2 = c2 % 6;
        const int d = (k1 < 3 ? k1 : k1 + 3) * 15 + (k2 < 3 ? k2 : k2 + 3);
        if (c1 < 6 && c2 < 6) S.cov[d] += t;
        if (c1 >= 6 && c2 >= 6) S.C[d] += t;
        if (c1 < 6 && c2 >= 6) S.E[d] += t;
      }
      __syncthreads();
    }
    if (!fixedLast) {
      for (int eidx = tid; eidx < 225; eidx += BS) {
        const int a = eidx / 15, cc = eidx % 15;
        double t = 0;
        for (int q = 0; q < 15; q++) t += (rhoP * S.Hp[a * 15 + q]) * S.JP[q * 15 + cc];
        S.T[a * 15 + cc] = t;
      }
      __syncthreads();
      for (int eidx = tid; eidx < 225; eidx += BS) {
        const int c1 = eidx / 15, c2 = eidx % 15;
        double t = 0;
        for (int a = 0; a < 15; a++) t += S.JP[a * 15 + c1] * S.T[a * 15 + c2];
        S.C[c1 * 15 + c2] += t;
      }
      __syncthreads();
      // C^-1 by Gauss-Jordan with partial pivoting (one wave), then cov -= E C^-1 E^T
      if (wave == 0) {
        double* M = S.Cinv;  // 15 x 30
        for (int eidx = lane; eidx < 450; eidx += 64) {
          const int i = eidx / 30, j = eidx % 30;
          M[eidx] = j < 15 ? S.C[i * 15 + j] : (j - 15 == i ? 1.0 : 0.0);
        }
        wave_sync();
        for (int cidx = 0; cidx < 15; cidx++) {
          int piv = cidx;
          double best = fabs(M[cidx * 30 + cidx]);
          for (int r = cidx + 1; r < 15; r++)
            if (fabs(M[r * 30 + cidx]) > best) best = fabs(M[r * 30 + cidx]), piv = r;
          if (piv != cidx) {
            if (lane < 30) {
              const double t = M[cidx * 30 + lane];
              M[cidx * 30 + lane] = M[piv * 30 + lane];
              M[piv * 30 + lane] = t;
            }
            wave_sync();
          }
          const double d = M[cidx * 30 + cidx];
          wave_sync();
          if (lane < 30) M[cidx * 30 + lane] /= d;
          wave_sync();
          double fr[8];
          for (int h = 0; h < 8; h++) {
            const int eidx = lane + 64 * h;
            fr[h] = eidx < 450 ? M[(eidx / 30) * 30 + cidx] : 0;
          }
          wave_sync();
          for (int h = 0; h < 8; h++) {
            const int eidx = lane + 64 * h;
            if (eidx < 450 && eidx / 30 != cidx) M[eidx] -= fr[h] * M[cidx * 30 + eidx % 30];
          }
          wave_sync();
        }
      }
      __syncthreads();
      for (int eidx = tid; eidx < 225; eidx += BS) {  // T = E * C^-1
        const int i = eidx / 15, j = eidx % 15;
        double t = 0;
        for (int k = 0; k < 15; k++) t += S.E[i * 15 + k] * S.Cinv[k * 30 + 15 + j];
        S.T[i * 15 + j] = t;
      }
      __syncthreads();
      for (int eidx = tid; eidx < 225; eidx += BS) {
        const int i = eidx / 15, j = eidx % 15;
        double t = 0;
        for (int k = 0; k < 15; k++) t += S.T[i * 15 + k] * S.E[j * 15 + k];
        S.cov[eidx] -= t;
      }
      __syncthreads();
    }
    for (int i = tid; i < 225; i += BS) R->H_marg[i] = S.cov[i];
  } else {
    for (int i = tid; i < 225; i += BS) R->H_marg[i] = 0;
  }
  if (tid == 0 && g == 0) {  // (the replicas hold the same values)
    R->base.nav = F.base.nav;
    ns_store(S.nsj, R->base.nav);
    R->base.n_inliers = N - nBad;
    R->base.status = (G > 1 && S.xfail) ? VIEO_E_HIP : VIEO_POSE_OK;
    R->base.lm_iterations = total_iters;
    R->base.reserved = total_trials;  // lambda trials over the four rounds (diagnostic)
    R->has_marg = F.compute_marg ? 1 : 0;
    R->reserved = 0;
  }
  };  // optimise
  using one_c = std::integral_constant<int, 1>;
  if constexpr (BS == 256 && !MC) {
    if (N <= kVioSplitObs)
      optimise(std::integral_constant<int, 192>{}, one_c{});
    else
      optimise(std::integral_constant<int, BS>{}, one_c{});
  } else if constexpr (BS == 256 && MC) {
    // a replicated launch: below kVioReplicaMinObs edges the ~36 exchanges (3 us each) cost more than the shared
    // visual passes save -- replica 0 takes the frame alone; each replica's share is small, so the single-lane edges
    // run beside the visual ones (128 visual threads x 64 mask bits x kXG replicas >= any frame the kernel accepts).
    // (ONE call site per instance of the body: called from two it stays out of line, and its by-reference captures --
    // every local above -- then live in scratch memory: 4-camera frame 2.8 -> 3.3 ms.)
    bool rep = gridDim.y == (unsigned)kXG && xchg != nullptr;
    // (test hook, VIEO_POSE_REPLICA_DROP=1: bit 31 of the launch number makes replica 5 stay away, as a workgroup that
    // never becomes resident would -- the others time out, poison their slots, and the host repeats the frame)
    if (rep && (launch_id >> 31) && blockIdx.y == 5 && N >= kVioReplicaMinObs) return;
    if (rep && N < kVioReplicaMinObs) {
      if (blockIdx.y != 0) return;
      rep = false;
    }
    if (rep)
      optimise(std::integral_constant<int, 128>{}, std::integral_constant<int, kXG>{});
    else
      optimise(std::integral_constant<int, BS>{}, one_c{});
  } else
    optimise(std::integral_constant<int, BS>{}, one_c{});
}

}  // namespace vieo

using namespace vieo;

// The exchange records of the replicated launches, one set per (host thread, device, stream): launches on one stream
// are ordered, so a record is never shared by two kernels in flight.  Zeroed once, on the launch stream (tag 0 is never
// used: launches count from 1); freed when the host thread ends.
static PoseXchg* vio_xchg_records(hipStream_t stream, int n_frames, unsigned* launch_id) {
  struct Rec {
    hipStream_t st;
    int dev;
    PoseXchg* p;
    int n;
    unsigned launches;
  };
  struct Recs {
    std::vector<Rec> v;
    ~Recs() {
      for (Rec& r : v)
        if (r.p) {
          int cur = 0;
          (void)hipGetDevice(&cur);
          if (cur != r.dev) (void)hipSetDevice(r.dev);
          (void)hipFree(r.p);
          if (cur != r.dev) (void)hipSetDevice(cur);
        }
    }
  };
  static thread_local Recs recs;
  int dev = 0;
  (void)hipGetDevice(&dev);
  auto fresh = [&](Rec& r) -> PoseXchg* {
    r.p = nullptr, r.n = 0;
    if (hipMalloc(&r.p, sizeof(PoseXchg) * n_frames) != hipSuccess) return nullptr;
    if (hipMemsetAsync(r.p, 0, sizeof(PoseXchg) * n_frames, stream) != hipSuccess) {
      (void)hipFree(r.p);
      r.p = nullptr;
      return nullptr;
    }
    r.n = n_frames;
    return r.p;
  };
  for (Rec& r : recs.v)
    if (r.st == stream && r.dev == dev) {
      *launch_id = ++r.launches;
      // A granule's tag carries 20 bits of the launch number, and not every slot is rewritten by every launch (the records
      // of frames a call does not have, launches whose frames stay below kVioReplicaMinObs): before the number wraps, the
      // records are zeroed on the launch stream -- tag 0 is never used -- so a granule left 2^20 launches ago cannot be
      // taken for a fresh one.
      if ((r.launches & 0x7FFFFu) == 0 && r.p) (void)hipMemsetAsync(r.p, 0, sizeof(PoseXchg) * r.n, stream);
      if (r.n >= n_frames) return r.p;
      (void)hipStreamSynchronize(stream);
      (void)hipFree(r.p);
      return fresh(r);
    }
  Rec r{stream, dev, nullptr, 0, 1};
  PoseXchg* p = fresh(r);
  if (!p) return nullptr;
  recs.v.push_back(r);
  *launch_id = 1;
  return p;
}

// rig frames of a small call (the one-call tracker: one frame): kXG replicas per frame share the visual edges.
// vieo_pose_set_replicas(0) / VIEO_POSE_REPLICAS=0 keeps one workgroup per frame (measurements, tests of both forms).
static int& vio_replicas() {
  static thread_local int on = [] {
    const char* e = getenv("VIEO_POSE_REPLICAS");
    return (e && atoi(e) == 0) ? 0 : 1;
  }();
  return on;
}
constexpr int kVioReplicaFrames = 4;
// The replicas of a launch wait for each other, so all of them must be resident at once: n_frames x kXG workgroups against
// what the device holds of this kernel (occupancy x CUs; a partitioned or smaller device, or a kernel instance that
// outgrew a CU's LDS, then keeps the one-workgroup form).  Other work on the device can still delay a replica: that is
// what the time-out + poison + the host's repeat on one workgroup are for.
template <bool MC, bool ENC>
static bool vio_replicas_fit(int n_frames) {
  static thread_local int cached_dev = -1, cached_blocks = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev != cached_dev) {
    int per_cu = 0, cus = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pose_opt_vio<256, MC, ENC>, 256, 0) != hipSuccess) per_cu = 0;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    cached_dev = dev, cached_blocks = per_cu * cus;
  }
  return n_frames * kXG <= cached_blocks;
}

template <bool MC, bool ENC>
static void vio_launch_kind(bool narrow, const vieo_vio_frame* d_frames, int n_frames, const vieo_pose_obs* d_obs,
                            uint8_t* d_outlier, vieo_vio_result* d_results, int others, hipStream_t stream) {
  const bool replicas = vio_replicas() != 0 && MC && n_frames <= kVioReplicaFrames && vio_replicas_fit<MC, ENC>(n_frames);
  if (narrow && !MC)  // rig frames carry n_cams x the observations: always the wide form (up to 16384 edges)
    hipLaunchKernelGGL((k_pose_opt_vio<64, MC, ENC>), dim3(n_frames), dim3(64), 0, stream, d_frames, d_obs,
                       d_outlier, d_results, others, (PoseXchg*)nullptr, 0u);
  else {
    unsigned launch_id = 0;
    PoseXchg* xb = replicas ? vio_xchg_records(stream, kVioReplicaFrames, &launch_id) : nullptr;
    if (xb) {
      const char* drop = getenv("VIEO_POSE_REPLICA_DROP");  // (read per call: the test switches it in-process)
      if (drop && atoi(drop) > 0) launch_id |= 0x80000000u;
    }
    hipLaunchKernelGGL((k_pose_opt_vio<256, MC, ENC>), dim3(n_frames, xb ? kXG : 1), dim3(256), 0, stream, d_frames, d_obs,
                       d_outlier, d_results, others, xb, launch_id);
  }
}

extern "C" {

int vieo_pose_set_replicas(int on) {
  const int was = vio_replicas();
  vio_replicas() = on ? 1 : 0;
  return was;
}

// which / which_enc: bit 0 the rectified / encoder-less instance, bit 1 the rig / encoder instance
static int vio_launch(const vieo_vio_frame* d_frames, int n_frames, const vieo_pose_obs* d_obs,
                      uint8_t* d_outlier, vieo_vio_result* d_results, int which, int which_enc, void* stream) {
  if (!d_frames || n_frames <= 0 || !d_obs || !d_outlier || !d_results) return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  // one CU holds four wavefront-sized frames; below that a frame gets four wavefronts.
  // VIEO_POSE_THREADS=64|256 overrides the choice (tuning / tests).
  static const int forced = [] {
    const char* e = getenv("VIEO_POSE_THREADS");
    return e ? atoi(e) : 0;
  }();
  const bool narrow = forced == 64 || (forced != 256 && n_frames > 256);
  const int others = (which == 3 ? 1 : 0) | (which_enc == 3 ? 2 : 0);
  hipStream_t st = (hipStream_t)stream;
  if ((which & 1) && (which_enc & 1)) vio_launch_kind<false, false>(narrow, d_frames, n_frames, d_obs, d_outlier, d_results, others, st);
  if ((which & 2) && (which_enc & 1)) vio_launch_kind<true, false>(narrow, d_frames, n_frames, d_obs, d_outlier, d_results, others, st);
  if ((which & 1) && (which_enc & 2)) vio_launch_kind<false, true>(narrow, d_frames, n_frames, d_obs, d_outlier, d_results, others, st);
  if ((which & 2) && (which_enc & 2)) vio_launch_kind<true, true>(narrow, d_frames, n_frames, d_obs, d_outlier, d_results, others, st);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

int vieo_pose_optimization_vio_batch_device(const vieo_vio_frame* d_frames, int n_frames,
                                            const vieo_pose_obs* d_obs, uint8_t* d_outlier,
                                            vieo_vio_result* d_results, void* stream) {
  return vio_launch(d_frames, n_frames, d_obs, d_outlier, d_results, vieo::pose_rig_launches(),
                    vieo::pose_enc_launches(), stream);
}

int vieo_pose_optimization_vio_batch_device_ex(const vieo_vio_frame* d_frames, int n_frames, const vieo_pose_obs* d_obs,
                                               uint8_t* d_outlier, vieo_vio_result* d_results, int cams_mode,
                                               int enc_mode, void* stream) {
  if (cams_mode < VIEO_POSE_CAMS_AUTO || cams_mode > VIEO_POSE_CAMS_RIG || enc_mode < VIEO_POSE_ENC_AUTO ||
      enc_mode > VIEO_POSE_ENC_ALL)
    return VIEO_E_INVALID;
  return vio_launch(d_frames, n_frames, d_obs, d_outlier, d_results, vieo::pose_launch_mask(cams_mode),
                    vieo::pose_launch_mask(enc_mode), stream);
}

int vieo_pose_optimization_vio(const vieo_vio_frame* h_frame, const vieo_pose_obs* h_obs,
                               uint8_t* h_outlier, vieo_vio_result* h_result) {
  if (!h_frame || !h_result || (h_frame->base.n_obs > 0 && (!h_obs || !h_outlier))) return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  const int n = h_frame->base.n_obs, nc = h_frame->base.n_cams;
  if (nc < 0 || nc > 4 || (nc > 0 && !h_frame->base.cams)) {
    set_error("PoseOptimization (VIO): n_cams = %d (0..4) needs `cams`", nc);
    return VIEO_E_INVALID;
  }
  const bool has_enc = h_frame->base.enc && h_frame->base.enc->enc.dt != 0;
  // one block up (frame, cameras, encoder edge, observations), one back (result, outlier flags).  The device
  // pointers inside the frame record are only known once the block is placed, so the record is patched in the
  // pinned copy before the upload.
  static thread_local Staging G;
  G.reset();
  vieo_vio_frame F = *h_frame;
  F.base.obs_begin = 0;
  const size_t o_f = G.in(&F, sizeof(F));
  const size_t o_c = G.in(h_frame->base.cams, nc > 0 ? (size_t)nc * sizeof(vieo_camera) : 0);
  const size_t o_e = G.in(h_frame->base.enc, has_enc ? sizeof(vieo_pose_enc) : 0);
  const size_t o_o = G.in(h_obs ? h_obs + h_frame->base.obs_begin : nullptr, (size_t)std::max(n, 0) * sizeof(vieo_pose_obs));
  const size_t o_r = G.out(sizeof(vieo_vio_result)), o_u = G.out((size_t)std::max(n, 1));
  if ((rc = G.pin.ensure(G.used)) != VIEO_OK || (rc = G.dev.ensure(G.used)) != VIEO_OK) return rc;
  F.base.cams = nc > 0 ? G.d<vieo_camera>(o_c) : nullptr;
  F.base.enc = has_enc ? G.d<vieo_pose_enc>(o_e) : nullptr;
  if ((rc = G.upload(nullptr)) != VIEO_OK) return rc;
  rc = vio_launch(G.d<vieo_vio_frame>(o_f), 1, G.d<vieo_pose_obs>(o_o), G.d<uint8_t>(o_u), G.d<vieo_vio_result>(o_r),
                  nc > 0 ? 2 : 1, has_enc ? 2 : 1, nullptr);
  if (rc != VIEO_OK) return rc;
  if ((rc = G.download(o_r, nullptr)) != VIEO_OK) return rc;
  if (((const vieo_vio_result*)G.h(o_r))->base.status == VIEO_E_HIP && vio_replicas()) {
    // a replica of the frame never became resident (the device is shared with other work): the same optimisation on
    // one workgroup -- slower, same result up to the association order of the visual sums
    const int was = vieo_pose_set_replicas(0);
    rc = vio_launch(G.d<vieo_vio_frame>(o_f), 1, G.d<vieo_pose_obs>(o_o), G.d<uint8_t>(o_u), G.d<vieo_vio_result>(o_r),
                    nc > 0 ? 2 : 1, has_enc ? 2 : 1, nullptr);
    (void)vieo_pose_set_replicas(was);
    if (rc != VIEO_OK) return rc;
    if ((rc = G.download(o_r, nullptr)) != VIEO_OK) return rc;
  }
  memcpy(h_result, G.h(o_r), sizeof(vieo_vio_result));
  if (n > 0) memcpy(h_outlier + h_frame->base.obs_begin, G.h(o_u), n);
  if (h_result->base.status == VIEO_E_CAPACITY) {
    set_error("PoseOptimization (VIO): %d observations exceed the kernel's capacity (16384)", n);
    return VIEO_E_CAPACITY;
  }
  return VIEO_OK;
}

}  // extern "C"

#ifdef VIEO_POSE_PROBE
extern "C" int vieo_debug_pose_probe(unsigned long long* out) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(vieo::g_pose_probe), sizeof(unsigned long long) * 24);
  unsigned long long z[24] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(vieo::g_pose_probe), z, sizeof(z));
  return 0;
}
#endif
