// ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.  EdgeEncNavState<DV>::computeError / linearizeOplus (reference
// src/Odom/g2otypes.h:606-665, USE_P_PLUS_RDP on: NavState.h:8) restated on the closed forms of smallmat.hpp.
#pragma once
#include <cmath>
#include <utility>

#include "smallmat.hpp"

namespace vo {

// inverse of a small dense matrix, Gauss-Jordan with partial pivoting (n <= 9)
static inline bool gj_inverse(const double* A, double* Ainv, int n) {
  double M[9][18];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) M[i][j] = A[i * n + j], M[i][n + j] = (i == j);
  for (int c = 0; c < n; c++) {
    int piv = c;
    for (int r = c + 1; r < n; r++)
      if (std::fabs(M[r][c]) > std::fabs(M[piv][c])) piv = r;
    if (M[piv][c] == 0) return false;
    if (piv != c)
      for (int j = 0; j < 2 * n; j++) std::swap(M[c][j], M[piv][j]);
    const double d = M[c][c];
    for (int j = 0; j < 2 * n; j++) M[c][j] /= d;
    for (int r = 0; r < n; r++)
      if (r != c) {
        const double f = M[r][c];
        if (f != 0)
          for (int j = 0; j < 2 * n; j++) M[r][j] -= f * M[c][j];
      }
  }
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) Ainv[i * n + j] = M[i][n + j];
  return true;
}

struct EncPose {  // the PR part of a NavState
  double p[3];
  Quat q;  // Rwb
};

// err[6] = (eR, ep); Ji / Jj: 6 x 6, columns (dp, dphi) of vertex i / j.  J may be null.
static inline void enc_edge_eval(const EncPose& si, const EncPose& sj, const double* meas, const Quat& qRbe,
                                 const double* pbe, double* err, double* Ji, double* Jj) {
  Quat qRiw = si.q;
  qRiw.x = -qRiw.x, qRiw.y = -qRiw.y, qRiw.z = -qRiw.z;
  Quat qReb = qRbe;
  qReb.x = -qReb.x, qReb.y = -qReb.y, qReb.z = -qReb.z;
  const Quat qRij = quat_mul(qRiw, sj.q);
  Quat qe = quat_mul(quat_mul(qReb, qRij), qRbe);
  quat_normalize(qe);  // Sophus::SO3exd(q)
  Quat qm = so3_exp(meas);
  qm.x = -qm.x, qm.y = -qm.y, qm.z = -qm.z;  // .inverse()
  Quat ql = quat_mul(qm, qe);
  quat_normalize(ql);
  so3_log(ql, err);
  double Riw[9], Rij[9], Reb[9];
  quat_to_R(qRiw, Riw), quat_to_R(qRij, Rij), quat_to_R(qReb, Reb);
  const double dpw[3] = {sj.p[0] - si.p[0], sj.p[1] - si.p[1], sj.p[2] - si.p[2]};
  double a[3], b[3], c[3];
  m3_v(Riw, dpw, a);   // Rbiw (pwj - pwi)
  m3_v(Rij, pbe, b);   // Rbiw Rwbj pbe
  for (int k = 0; k < 3; k++) c[k] = a[k] - pbe[k] + b[k];
  double dp[3];
  m3_v(Reb, c, dp);
  for (int k = 0; k < 3; k++) err[3 + k] = dp[k] - meas[3 + k];
  if (!Ji) return;
  double JrInv[9], RijT[9], T[9], RebRiw[9], Rwi[9], Rwj[9], hv[9], v[3];
  so3_JrInv(err, JrInv);
  m3_T(Rij, RijT);
  for (int k = 0; k < 36; k++) Ji[k] = 0, Jj[k] = 0;
  auto put = [](double* J, int r0, int c0, const double* M, double s) {
    for (int r = 0; r < 3; r++)
      for (int q = 0; q < 3; q++) J[(r0 + r) * 6 + c0 + q] = s * M[r * 3 + q];
  };
  // JeR_dphii = -Jrinv(eR) * Reb * Rij^T
  m3_mul(JrInv, Reb, T);
  double T2[9];
  m3_mul(T, RijT, T2);
  put(Ji, 0, 3, T2, -1.0);
  // Jep_dpi = -Reb * Rbiw * Rwbi (P + R dp model)
  m3_mul(Reb, Riw, RebRiw);
  quat_to_R(si.q, Rwi), quat_to_R(sj.q, Rwj);
  m3_mul(RebRiw, Rwi, T);
  put(Ji, 3, 0, T, -1.0);
  // Jep_dphii = Reb * hat(Rij * pbe + Riw * (pwj - pwi))
  for (int k = 0; k < 3; k++) v[k] = b[k] + a[k];
  hat(v, hv);
  m3_mul(Reb, hv, T);
  put(Ji, 3, 3, T, 1.0);
  // JeR_dphij = Jrinv(eR) * Reb
  m3_mul(JrInv, Reb, T);
  put(Jj, 0, 3, T, 1.0);
  // Jep_dpj = Reb * Rbiw * Rwbj
  m3_mul(RebRiw, Rwj, T);
  put(Jj, 3, 0, T, 1.0);
  // Jep_dphij = -(Reb * Rij) * hat(pbe)
  double RebRij[9];
  m3_mul(Reb, Rij, RebRij);
  hat(pbe, hv);
  m3_mul(RebRij, hv, T);
  put(Jj, 3, 3, T, -1.0);
}

}  // namespace vo
