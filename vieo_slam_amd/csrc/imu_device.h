// imu_device.h -- NavState, SO(3) closed forms (common/so3_extra.h) and the inertial edge
// EdgeNavStateI (src/Odom/g2otypes.h:703-884) shared by pose_opt_vio.hip and lba.hip.
#pragma once
#include "ba_device.h"

namespace vieo {

struct NSd {
  double p[3], v[3], qw, qx, qy, qz, bg[3], ba[3], dbg[3], dba[3];
};

__device__ __forceinline__ void ns_load(NSd& s, const vieo_navstate& n) {
  for (int i = 0; i < 3; i++) {
    s.p[i] = n.p[i], s.v[i] = n.v[i], s.bg[i] = n.bg[i], s.ba[i] = n.ba[i];
    s.dbg[i] = n.dbg[i], s.dba[i] = n.dba[i];
  }
  s.qw = n.q[0], s.qx = n.q[1], s.qy = n.q[2], s.qz = n.q[3];
}
__device__ __forceinline__ void ns_store(const NSd& s, vieo_navstate& n) {
  for (int i = 0; i < 3; i++) {
    n.p[i] = s.p[i], n.v[i] = s.v[i], n.bg[i] = s.bg[i], n.ba[i] = s.ba[i];
    n.dbg[i] = s.dbg[i], n.dba[i] = s.dba[i];
  }
  n.q[0] = s.qw, n.q[1] = s.qx, n.q[2] = s.qy, n.q[3] = s.qz;
}

struct Qd {
  double w, x, y, z;
};
__device__ __forceinline__ Qd q_of(const NSd& s) { return Qd{s.qw, s.qx, s.qy, s.qz}; }
// (one division and four products instead of four divisions: a double-precision division is a dozen dependent
// instructions, ~76 cycles on a single-lane chain (tools/micro/lat_bench.hip); the results differ from Eigen's normalize() in the last bit, far inside the
// 1e-4 parity tolerance of the optimisers)
__device__ __forceinline__ Qd q_norm(Qd q) {
  const double r = 1.0 / sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  return Qd{q.w * r, q.x * r, q.y * r, q.z * r};
}
__device__ __forceinline__ Qd q_mul(const Qd& a, const Qd& b) {
  return Qd{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ Qd q_conj(const Qd& q) { return Qd{q.w, -q.x, -q.y, -q.z}; }
__device__ __forceinline__ void q_to_R(const Qd& q, double* R) {
  Est e;
  e.qw = q.w, e.qx = q.x, e.qy = q.y, e.qz = q.z;
  quat_to_R(e, R);
}
__device__ __forceinline__ Qd R_to_q(const double* R) {  // Eigen Quaternion(Matrix3) + normalize
  Qd q;
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (R[7] - R[5]) * t, q.y = (R[2] - R[6]) * t, q.z = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    double v[3];
    v[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (R[k * 3 + j] - R[j * 3 + k]) * t;
    v[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    v[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    q.x = v[0], q.y = v[1], q.z = v[2];
  }
  return q_norm(q);
}
// SO3ex::exp / log / JacobianR / JacobianRInv (common/so3_extra.h:121-190,254-288)
__device__ __forceinline__ Qd so3_exp_q(const double* w) {
  const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  double imag, real;
  if (th < 1e-5) {
    const double t2 = th * th;
    imag = 0.5 - t2 / 48., real = 1.0 - t2 / 8.;
  } else {
    const double h = 0.5 * th;
    double sh, ch;
    sincos(h, &sh, &ch);  // one argument reduction for both
    imag = sh / th, real = ch;
  }
  return q_norm(Qd{real, imag * w[0], imag * w[1], imag * w[2]});
}
__device__ __forceinline__ void so3_log_q(const Qd& q, double* out) {
  const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z), w = q.w, sw = w * w;
  double f;
  if (n < 1e-5) {
    f = 2. / w - 2. / 3 * (n * n) / (w * sw);
  } else if (fabs(w) < 1e-5) {
    f = (w > 0 ? M_PI : -M_PI) / n;
    const double n2 = n * n, n4 = n2 * n2;
    f -= 2 * w / n2 - 2. / 3 * (w * sw) / n4;
  } else
    f = 2 * atan(n / w) / n;
  out[0] = f * q.x, out[1] = f * q.y, out[2] = f * q.z;
}
__device__ __forceinline__ void hat3(const double* w, double* O) {
  O[0] = 0, O[1] = -w[2], O[2] = w[1], O[3] = w[2], O[4] = 0, O[5] = -w[0], O[6] = -w[1], O[7] = w[0], O[8] = 0;
}
__device__ __forceinline__ void mm3(const double* A, const double* B, double* C) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
  for (int i = 0; i < 9; i++) C[i] = t[i];
}
__device__ __forceinline__ void mv3(const double* A, const double* v, double* r) {
  const double a = A[0] * v[0] + A[1] * v[1] + A[2] * v[2], b = A[3] * v[0] + A[4] * v[1] + A[5] * v[2],
               c = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
  r[0] = a, r[1] = b, r[2] = c;
}
__device__ __forceinline__ void mTv3(const double* A, const double* v, double* r) {
  const double a = A[0] * v[0] + A[3] * v[1] + A[6] * v[2], b = A[1] * v[0] + A[4] * v[1] + A[7] * v[2],
               c = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
  r[0] = a, r[1] = b, r[2] = c;
}
__device__ __forceinline__ void so3_Jr_d(const double* w, double* J) {
  const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  double O[9], O2[9];
  if (th < 1e-5) {
    hat3(w, O);
    mm3(O, O, O2);
    for (int i = 0; i < 9; i++) J[i] = ((i % 4) == 0 ? 1.0 : 0.0) - 0.5 * O[i] + O2[i] / 6.;
  } else {
    const double k[3] = {w[0] / th, w[1] / th, w[2] / th};
    hat3(k, O);
    mm3(O, O, O2);
    double sth, cth;
    sincos(th, &sth, &cth);
    const double a = (1 - cth) / th, b = 1 - sth / th;
    for (int i = 0; i < 9; i++) J[i] = ((i % 4) == 0 ? 1.0 : 0.0) - a * O[i] + b * O2[i];
  }
}
__device__ __forceinline__ void so3_JrInv_d(const double* w, double* J) {
  const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  double O[9], O2[9];
  hat3(w, O);
  if (th < 1e-5) {
    mm3(O, O, O2);
    for (int i = 0; i < 9; i++) J[i] = ((i % 4) == 0 ? 1.0 : 0.0) + 0.5 * O[i] + (1. / 12.) * O2[i];
  } else {
    const double k[3] = {w[0] / th, w[1] / th, w[2] / th};
    double K[9];
    hat3(k, K);
    mm3(K, K, O2);
    double sth, cth;
    sincos(th, &sth, &cth);
    const double c = 1.0 - (1.0 + cth) * th / (2.0 * sth);
    for (int i = 0; i < 9; i++) J[i] = ((i % 4) == 0 ? 1.0 : 0.0) + 0.5 * O[i] + c * O2[i];
  }
}

// NavState::IncSmall(dPVR) + IncSmallBias (NavState.h:64-83)
__device__ __forceinline__ void ns_inc(NSd& s, const double* d, const double* db) {
  double R[9], Rd[3];
  q_to_R(q_of(s), R);
  mv3(R, d, Rd);
  for (int i = 0; i < 3; i++) s.p[i] += Rd[i], s.v[i] += d[3 + i];
  const Qd q = q_norm(q_mul(q_of(s), so3_exp_q(d + 6)));
  s.qw = q.w, s.qx = q.x, s.qy = q.y, s.qz = q.z;
  for (int i = 0; i < 3; i++) s.dbg[i] += db[i], s.dba[i] += db[3 + i];
}

// EdgeNavStateI<NV>::computeError (g2otypes.h:733-776): rows [r_p, then r_R at idR, r_v at 9 - idR]
// (idR = 6: EdgeNavStatePVR of PoseOptimization; idR = 3: EdgeNavStatePRV of the local BA)
// part: 0 = all nine rows; 1 = the position and velocity rows, 2 = the rotation rows (the quaternion chain with its exp /
// log) -- the two halves share nothing but the inputs, so two lanes of different wavefronts evaluate them side by side.
static __device__ void imu_error(const vieo_imu_preint& M, const double* gw, const NSd& si, const NSd& sj,
                          double* err, int idR = 6, int part = 0) {
  if (part != 2) {
    double Ri[9], t[3], r[3], Jb[3], Ja[3];
    q_to_R(q_of(si), Ri);
    const double dt = M.dt;
    for (int k = 0; k < 3; k++) t[k] = sj.p[k] - si.p[k] - si.v[k] * dt - gw[k] * (dt * dt / 2);
    mTv3(Ri, t, r);
    mv3(M.Jgp, si.dbg, Jb);
    mv3(M.Jap, si.dba, Ja);
    for (int k = 0; k < 3; k++) err[k] = r[k] - (M.pij[k] + Jb[k] + Ja[k]);
    for (int k = 0; k < 3; k++) t[k] = sj.v[k] - si.v[k] - gw[k] * dt;
    mTv3(Ri, t, r);
    mv3(M.Jgv, si.dbg, Jb);
    mv3(M.Jav, si.dba, Ja);
    for (int k = 0; k < 3; k++) err[9 - idR + k] = r[k] - (M.vij[k] + Jb[k] + Ja[k]);
  }
  if (part != 1) {
    double w[3];
    mv3(M.JgR, si.dbg, w);
    const Qd qa = q_norm(q_mul(R_to_q(M.Rij), so3_exp_q(w)));
    const Qd qb = q_norm(q_mul(q_conj(q_of(si)), q_of(sj)));
    so3_log_q(q_norm(q_mul(q_conj(qa), qb)), err + idR);
  }
}

__device__ __forceinline__ void set3(double* J, int ld, int r0, int c0, const double* M, double s) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) J[(r0 + i) * ld + c0 + j] = s * M[i * 3 + j];
}

// EdgeNavStateI<NV>::linearizeOplus (g2otypes.h:777-884).  J is 9 x 24:
// columns 0..8 = state j, 9..17 = state i (each p, then R at idR, V at idV), 18..23 = Bias_i;
// rows as in imu_error.  (idR, idV) = (6, 3) for PVR, (3, 6) for PRV.
// part: 0 = the whole Jacobian (J is cleared first); 1 = the position and velocity rows, 2 = the rotation rows -- the two
// halves share no intermediate, so two lanes of different wavefronts can form them side by side (the caller clears J).
static __device__ void imu_linearize(const vieo_imu_preint& M, const double* gw, const NSd& si, const NSd& sj,
                              const double* err, double* J, int idR = 6, int idV = 3, int part = 0) {
  const int ld = 24, cj = 0, ci = 9, cb = 18;
  if (part == 0)
    for (int i = 0; i < 9 * 24; i++) J[i] = 0;
  double tmp[9], tmp2[9];
  if (part != 2) {
    double Ri[9], RiT[9], Rj[9], t[3], r[3], Hm[9], I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    q_to_R(q_of(si), Ri);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) RiT[i * 3 + j] = Ri[j * 3 + i];
    q_to_R(q_of(sj), Rj);
    const double dt = M.dt;
    for (int k = 0; k < 3; k++) t[k] = sj.p[k] - si.p[k] - si.v[k] * dt - gw[k] * (dt * dt / 2);
    mv3(RiT, t, r);
    hat3(r, Hm);
    set3(J, ld, 0, ci + idR, Hm, 1.0);
    set3(J, ld, 0, ci + 0, I3, -1.0);
    set3(J, ld, 0, ci + idV, RiT, -dt);
    set3(J, ld, 0, cb + 0, M.Jgp, -1.0);
    set3(J, ld, 0, cb + 3, M.Jap, -1.0);
    mm3(RiT, Rj, tmp);
    set3(J, ld, 0, cj + 0, tmp, 1.0);
    for (int k = 0; k < 3; k++) t[k] = sj.v[k] - si.v[k] - gw[k] * dt;
    mv3(RiT, t, r);
    hat3(r, Hm);
    set3(J, ld, idV, ci + idR, Hm, 1.0);
    set3(J, ld, idV, ci + idV, RiT, -1.0);
    set3(J, ld, idV, cb + 0, M.Jgv, -1.0);
    set3(J, ld, idV, cb + 3, M.Jav, -1.0);
    set3(J, ld, idV, cj + idV, RiT, 1.0);
  }
  if (part != 1) {
    double Jrinv[9], Rji[9];
    const double* eR = err + idR;
    so3_JrInv_d(eR, Jrinv);
    q_to_R(q_norm(q_mul(q_conj(q_of(sj)), q_of(si))), Rji);
    mm3(Jrinv, Rji, tmp);
    set3(J, ld, idR, ci + idR, tmp, -1.0);
    const double meR[3] = {-eR[0], -eR[1], -eR[2]};
    double E[9], w[3], Jr[9];
    q_to_R(so3_exp_q(meR), E);
    mv3(M.JgR, si.dbg, w);
    so3_Jr_d(w, Jr);
    mm3(Jrinv, E, tmp);
    mm3(tmp, Jr, tmp2);
    mm3(tmp2, M.JgR, tmp);
    set3(J, ld, idR, cb + 0, tmp, -1.0);
    set3(J, ld, idR, cj + idR, Jrinv, 1.0);
  }
}

// EdgeEncNavState<DV>::computeError / linearizeOplus (g2otypes.h:606-665, USE_P_PLUS_RDP on): the 6-row encoder
// edge between the PR parts of two states.  err = (eR, ep); Ji / Jj: 6 x 6, columns (dp, dphi); J may be null.
static __device__ void enc_edge_eval(const NSd& si, const NSd& sj, const double* meas, const double* qRbe4,
                                     const double* pbe, double* err, double* Ji, double* Jj) {
  const Qd qi = q_of(si), qj = q_of(sj), qbe = Qd{qRbe4[0], qRbe4[1], qRbe4[2], qRbe4[3]};
  const Qd qRiw = q_conj(qi), qReb = q_conj(qbe);
  const Qd qRij = q_mul(qRiw, qj);
  const Qd qe = q_norm(q_mul(q_mul(qReb, qRij), qbe));
  const Qd ql = q_norm(q_mul(q_conj(so3_exp_q(meas)), qe));
  so3_log_q(ql, err);
  double Riw[9], Rij[9], Reb[9];
  q_to_R(qRiw, Riw), q_to_R(qRij, Rij), q_to_R(qReb, Reb);
  const double dpw[3] = {sj.p[0] - si.p[0], sj.p[1] - si.p[1], sj.p[2] - si.p[2]};
  double a[3], b[3], c[3], dp[3];
  mv3(Riw, dpw, a);
  mv3(Rij, pbe, b);
  for (int k = 0; k < 3; k++) c[k] = a[k] - pbe[k] + b[k];
  mv3(Reb, c, dp);
  for (int k = 0; k < 3; k++) err[3 + k] = dp[k] - meas[3 + k];
  if (!Ji) return;
  double JrInv[9], RijT[9], T[9], T2[9], RebRiw[9], Rwi[9], Rwj[9], hv[9], v[3], RebRij[9];
  so3_JrInv_d(err, JrInv);
  for (int r = 0; r < 3; r++)
    for (int q = 0; q < 3; q++) RijT[r * 3 + q] = Rij[q * 3 + r];
  for (int k = 0; k < 36; k++) Ji[k] = 0, Jj[k] = 0;
  mm3(JrInv, Reb, T);
  mm3(T, RijT, T2);
  set3(Ji, 6, 0, 3, T2, -1.0);       // JeR_dphii = -Jrinv(eR) Reb Rij^T
  mm3(Reb, Riw, RebRiw);
  q_to_R(qi, Rwi), q_to_R(qj, Rwj);
  mm3(RebRiw, Rwi, T);
  set3(Ji, 6, 3, 0, T, -1.0);        // Jep_dpi = -Reb Rbiw Rwbi
  for (int k = 0; k < 3; k++) v[k] = b[k] + a[k];
  hat3(v, hv);
  mm3(Reb, hv, T);
  set3(Ji, 6, 3, 3, T, 1.0);         // Jep_dphii = Reb [Rij pbe + Riw (pwj - pwi)]^
  mm3(JrInv, Reb, T);
  set3(Jj, 6, 0, 3, T, 1.0);         // JeR_dphij = Jrinv(eR) Reb
  mm3(RebRiw, Rwj, T);
  set3(Jj, 6, 3, 0, T, 1.0);         // Jep_dpj = Reb Rbiw Rwbj
  mm3(Reb, Rij, RebRij);
  hat3(pbe, hv);
  mm3(RebRij, hv, T);
  set3(Jj, 6, 3, 3, T, -1.0);        // Jep_dphij = -Reb Rij pbe^
}

}  // namespace vieo
