// ORACLE -- TEST INFRASTRUCTURE ONLY (see ocv_prims.hpp header).  PARITY UNPINNED.
//
// CPU restatement of Optimizer::LocalBundleAdjustmentNavStatePRV (reference: src/Optimizer.cc:21-769;
// no encoder edges, th_dist_far = INFINITY) on a flattened window, with the pieces it runs through:
//   VertexNavStatePR / V / Bias and their oplus           src/Odom/g2otypes.h:271-288,553-567, NavState.h:47-83
//   EdgeNavStatePRV = EdgeNavStateI<5> (idR = 3, idV = 6)  src/Odom/g2otypes.h:703-884
//   EdgeNavStateBias                                        src/Odom/g2otypes.cpp:14-34
//   EdgeReprojectPR / PRStereo                              src/Odom/g2otypes.h:400-541
//   GraphOperator::Chi2LargeSetLevel                        optimizer/optimizer_ba/g2o_graph_operator.h:23-40
//   BlockSolverX Schur complement, LM with user lambda init g2o/core/block_solver.hpp, optimization_algorithm_levenberg.cpp
//   LinearSolverEigen -> dense LDL^T of the reduced (15 per key frame) system, same solution up to rounding.
// Reduced-system order: key frames in window order, [PR (dp, dphi), V, Bias (dbg, dba)] each.
// Full BA with bScaleOpt (System::FinalGBA, src/System.cc:24-33 -> Optimizer.cc:842-851,1131-1137,1190-1196,1256-1335):
//   VertexScale (g2otypes.h:292-311, 1 dim, s <- s + ds, id after every key-frame vertex = last column of the pose
//   system) and EdgeReprojectPRS / PRSStereo = EdgeReproject<DE, 6, 3, MODE_OPT_VAR = 1> (g2otypes.h:321-541,548-550):
//   Xw = s * Xh with the point vertex unscaled, J_s = (Jproj Rcw) Xh, J_Xh = s (Jproj Rcw); the points are written back
//   as s * Xh, the key frames as they are (the rescaling of p_wb is commented out in the reference, :1286-1289).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "../include/vieo_hot.h"
#include "cam_models.hpp"
#include "enc_edge.hpp"
#include "smallmat.hpp"

namespace vov {
using namespace vo;

struct KF {
  double p[3], v[3], dbg[3], dba[3], bg[3], ba[3];
  Quat q;
  bool fixed;
  int col;  // offset of PR in the pose system (V at +6, Bias at +9), -1 = fixed
};

struct VEdge {
  int kf, mp, de, level = 0, cam = 0;
  double obs[3], info, delta, dsqr;
  bool robust = true;
  double err[3] = {0, 0, 0};
};

struct IEdge {  // EdgeNavStatePRV + EdgeNavStateBias of one key-frame pair
  int i, j;
  const vieo_imu_preint* M;
  bool has_imu, robust;
  double InfoI[81];  // already scaled
  double infoBg, infoBa;
  double errI[9], errB[6];
  double J[9 * 24];  // columns: PR_i 0..5, PR_j 6..11, V_i 12..14, V_j 15..17, Bias_i 18..23
  // EdgeEncNavStatePR of the same pair (Optimizer.cc:323-347)
  bool has_enc = false, enc_robust = true;
  double measE[6], InfoE[36], errE[6], JEi[36], JEj[36];
};

static Quat qconj(const Quat& q) {
  Quat r = q;
  r.x = -q.x, r.y = -q.y, r.z = -q.z;
  return r;
}

static void hub(double e, double delta, double dsqr, double* rho) {
  if (e <= dsqr) {
    rho[0] = e, rho[1] = 1.;
  } else {
    double s = std::sqrt(e);
    rho[0] = 2 * s * delta - dsqr;
    rho[1] = delta / s;
  }
}

static bool inv3(const double* A, double* B) {  // Eigen 3x3 inverse (cofactors)
  const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  const double id = 1.0 / det;
  B[0] = c00 * id, B[1] = (A[2] * A[7] - A[1] * A[8]) * id, B[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  B[3] = c01 * id, B[4] = (A[0] * A[8] - A[2] * A[6]) * id, B[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  B[6] = c02 * id, B[7] = (A[1] * A[6] - A[0] * A[7]) * id, B[8] = (A[0] * A[4] - A[1] * A[3]) * id;
  return det != 0;
}

// general n x n inverse by Gauss-Jordan with partial pivoting (Eigen inverse() of a 9x9)
static bool mat_inverse(const double* A, double* Ainv, int n) {
  std::vector<double> M((size_t)n * 2 * n);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) M[i * 2 * n + j] = A[i * n + j], M[i * 2 * n + n + j] = (i == j);
  for (int c = 0; c < n; c++) {
    int piv = c;
    for (int r = c + 1; r < n; r++)
      if (std::fabs(M[r * 2 * n + c]) > std::fabs(M[piv * 2 * n + c])) piv = r;
    if (M[piv * 2 * n + c] == 0) return false;
    if (piv != c)
      for (int j = 0; j < 2 * n; j++) std::swap(M[c * 2 * n + j], M[piv * 2 * n + j]);
    const double d = M[c * 2 * n + c];
    for (int j = 0; j < 2 * n; j++) M[c * 2 * n + j] /= d;
    for (int r = 0; r < n; r++)
      if (r != c) {
        const double f = M[r * 2 * n + c];
        if (f != 0)
          for (int j = 0; j < 2 * n; j++) M[r * 2 * n + j] -= f * M[c * 2 * n + j];
      }
  }
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) Ainv[i * n + j] = M[i * 2 * n + n + j];
  return true;
}

struct W {
  const vieo_lba_vio_params* P;
  bool gba = false;  // GlobalBundleAdjustmentNavStatePRV: g2o's own initial lambda
  bool scale_opt = false;  // bScaleOpt: VertexScale + EdgeReprojectPRS[Stereo]
  double scale = 1.0;      // VertexScale::estimate(), setEstimate(1.) (Optimizer.cc:845)
  OCam cams[4];
  std::vector<KF> kf;
  std::vector<double> X;
  std::vector<VEdge> E;
  std::vector<IEdge> I;
  std::vector<int> mp_first, mp_count;

  // ---- visual edge (identical to the vision-only LBA)
  void project(const VEdge& e, double* proj, double* Pc_out, double* Rcw_out) const {
    const KF& s = kf[e.kf];
    const OCam& C = cams[e.cam];
    double Rwb[9], Rbw[9], Rcw[9], t[3], Pc[3];
    quat_to_R(s.q, Rwb);
    m3_T(Rwb, Rbw);
    m3_mul(C.Rcb, Rbw, Rcw);
    m3_v(Rcw, s.p, t);
    const double Xw[3] = {X[3 * e.mp] * scale, X[3 * e.mp + 1] * scale, X[3 * e.mp + 2] * scale};  // g2otypes.h:376
    m3_v(Rcw, Xw, Pc);
    for (int i = 0; i < 3; i++) Pc[i] += -t[i] + C.tcb[i];
    float uv[2];
    ocam_project(C, Pc, uv, nullptr);
    proj[0] = uv[0], proj[1] = uv[1];
    if (e.de > 2) proj[2] = proj[0] - (double)C.bf / Pc[2];
    if (Pc_out) memcpy(Pc_out, Pc, 24);
    if (Rcw_out) memcpy(Rcw_out, Rcw, 72);
  }
  void v_error(VEdge& e) const {
    double proj[3];
    project(e, proj, nullptr, nullptr);
    for (int i = 0; i < e.de; i++) e.err[i] = e.obs[i] - proj[i];
  }
  static double v_chi2(const VEdge& e) {
    double s = 0;
    for (int i = 0; i < e.de; i++) s += e.err[i] * (e.info * e.err[i]);
    return s;
  }
  bool depth_positive(const VEdge& e) const {
    double proj[3], Pc[3];
    project(e, proj, Pc, nullptr);
    return Pc[2] > 0.;
  }
  // Js (de x 1, may be null): Jacobian w.r.t. the scale vertex, _jacobianOplus[0] * Ph_unscale before the point block
  // is multiplied by the scale (g2otypes.h:514-518)
  void v_linearize(const VEdge& e, double* Jp, double* Jx, double* Js = nullptr) const {
    const OCam& C = cams[e.cam];
    double proj[3], Pc[3], Rcw[9];
    project(e, proj, Pc, Rcw);
    const KF& s = kf[e.kf];
    const double invz = 1 / Pc[2], invz_2 = invz * invz;
    double J[9] = {0}, Jc[6];
    ocam_project(C, Pc, nullptr, Jc);
    for (int i = 0; i < 6; i++) J[i] = -Jc[i];
    if (e.de > 2) J[6] = J[0], J[7] = J[1], J[8] = J[2] - (double)C.bf * invz_2;
    double Rwb[9], dP[3], Paux[3], H[9], RcbH[9];
    quat_to_R(s.q, Rwb);
    for (int i = 0; i < 3; i++) dP[i] = X[3 * e.mp + i] * scale - s.p[i];  // Pw = s * Xh
    m3T_v(Rwb, dP, Paux);
    hat(Paux, H);
    m3_mul(C.Rcb, H, RcbH);
    for (int r = 0; r < e.de; r++)
      for (int k = 0; k < 3; k++) {
        double a = 0, b = 0, c = 0;
        for (int m = 0; m < 3; m++) {
          a += J[r * 3 + m] * (-C.Rcb[m * 3 + k]);
          b += J[r * 3 + m] * RcbH[m * 3 + k];
          c += J[r * 3 + m] * Rcw[m * 3 + k];
        }
        Jp[r * 6 + k] = a;
        Jp[r * 6 + 3 + k] = b;
        Jx[r * 3 + k] = c;
      }
    if (Js)
      for (int r = 0; r < e.de; r++)
        Js[r] = Jx[r * 3] * X[3 * e.mp] + Jx[r * 3 + 1] * X[3 * e.mp + 1] + Jx[r * 3 + 2] * X[3 * e.mp + 2];
    if (scale_opt)
      for (int r = 0; r < e.de; r++)
        for (int k = 0; k < 3; k++) Jx[r * 3 + k] *= scale;
  }

  // ---- EdgeNavStatePRV::computeError (g2otypes.h:733-776, idR = 3) + EdgeNavStateBias
  void enc_eval(IEdge& e, bool jac) const {
    const KF &si = kf[e.i], &sj = kf[e.j];
    EncPose a, b;
    memcpy(a.p, si.p, 24), memcpy(b.p, sj.p, 24);
    a.q = si.q, b.q = sj.q;
    Quat qbe;
    qbe.w = P->qRbe[0], qbe.x = P->qRbe[1], qbe.y = P->qRbe[2], qbe.z = P->qRbe[3];
    enc_edge_eval(a, b, e.measE, qbe, P->pbe, e.errE, jac ? e.JEi : nullptr, jac ? e.JEj : nullptr);
  }
  static double chi2_E(const IEdge& e) {
    double s = 0;
    for (int a = 0; a < 6; a++) {
      double t = 0;
      for (int b = 0; b < 6; b++) t += e.InfoE[a * 6 + b] * e.errE[b];
      s += e.errE[a] * t;
    }
    return s;
  }
  void i_error(IEdge& e) const {
    const KF &si = kf[e.i], &sj = kf[e.j];
    if (e.has_enc) enc_eval(e, false);
    if (e.has_imu) {
      const vieo_imu_preint& M = *e.M;
      double Ri[9], RiT[9];
      quat_to_R(si.q, Ri);
      m3_T(Ri, RiT);
      const double dt = M.dt;
      double t[3], r[3], Jb[3], Ja[3];
      for (int k = 0; k < 3; k++) t[k] = sj.p[k] - si.p[k] - si.v[k] * dt - P->gw[k] * (dt * dt / 2);
      m3_v(RiT, t, r);
      m3_v(M.Jgp, si.dbg, Jb);
      m3_v(M.Jap, si.dba, Ja);
      for (int k = 0; k < 3; k++) e.errI[k] = r[k] - (M.pij[k] + Jb[k] + Ja[k]);
      double w[3];
      m3_v(M.JgR, si.dbg, w);
      Quat qa = quat_mul(R_to_quat(M.Rij), so3_exp(w));
      quat_normalize(qa);
      Quat qb = quat_mul(qconj(si.q), sj.q);
      quat_normalize(qb);
      Quat qe = quat_mul(qconj(qa), qb);
      quat_normalize(qe);
      so3_log(qe, &e.errI[3]);
      for (int k = 0; k < 3; k++) t[k] = sj.v[k] - si.v[k] - P->gw[k] * dt;
      m3_v(RiT, t, r);
      m3_v(M.Jgv, si.dbg, Jb);
      m3_v(M.Jav, si.dba, Ja);
      for (int k = 0; k < 3; k++) e.errI[6 + k] = r[k] - (M.vij[k] + Jb[k] + Ja[k]);
    }
    for (int k = 0; k < 3; k++) {
      e.errB[k] = (sj.bg[k] + sj.dbg[k]) - (si.bg[k] + si.dbg[k]);
      e.errB[3 + k] = (sj.ba[k] + sj.dba[k]) - (si.ba[k] + si.dba[k]);
    }
  }
  static double chi2_I(const IEdge& e) {
    double s = 0;
    for (int a = 0; a < 9; a++) {
      double t = 0;
      for (int b = 0; b < 9; b++) t += e.InfoI[a * 9 + b] * e.errI[b];
      s += e.errI[a] * t;
    }
    return s;
  }
  static double chi2_B(const IEdge& e) {
    double s = 0;
    for (int k = 0; k < 3; k++) s += e.errB[k] * (e.infoBg * e.errB[k]);
    for (int k = 3; k < 6; k++) s += e.errB[k] * (e.infoBa * e.errB[k]);
    return s;
  }
  static void set3(double* J, int r0, int c0, const double* M, double s) {
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) J[(r0 + a) * 24 + c0 + b] = s * M[a * 3 + b];
  }
  // EdgeNavStatePRV::linearizeOplus (g2otypes.h:777-884), idR = 3, idV = 6
  void i_linearize(IEdge& e) const {
    const KF &si = kf[e.i], &sj = kf[e.j];
    const vieo_imu_preint& M = *e.M;
    const int cPRi = 0, cPRj = 6, cVi = 12, cVj = 15, cB = 18;
    for (int k = 0; k < 9 * 24; k++) e.J[k] = 0;
    double Ri[9], RiT[9], Rj[9], t[3], r[3], Hm[9], I3[9], tmp[9], tmp2[9];
    m3_identity(I3);
    quat_to_R(si.q, Ri);
    m3_T(Ri, RiT);
    quat_to_R(sj.q, Rj);
    const double dt = M.dt;
    // rows 0..2: r_p
    for (int k = 0; k < 3; k++) t[k] = sj.p[k] - si.p[k] - si.v[k] * dt - P->gw[k] * (dt * dt / 2);
    m3_v(RiT, t, r);
    hat(r, Hm);
    set3(e.J, 0, cPRi + 3, Hm, 1.0);   // d r_p / d phi_i
    set3(e.J, 0, cPRi + 0, I3, -1.0);  // d r_p / d p_i  (p <- p + R dp)
    set3(e.J, 0, cVi, RiT, -dt);       // d r_p / d v_i
    set3(e.J, 0, cB + 0, M.Jgp, -1.0);
    set3(e.J, 0, cB + 3, M.Jap, -1.0);
    m3_mul(RiT, Rj, tmp);
    set3(e.J, 0, cPRj + 0, tmp, 1.0);  // d r_p / d p_j
    // rows 6..8: r_v
    for (int k = 0; k < 3; k++) t[k] = sj.v[k] - si.v[k] - P->gw[k] * dt;
    m3_v(RiT, t, r);
    hat(r, Hm);
    set3(e.J, 6, cPRi + 3, Hm, 1.0);
    set3(e.J, 6, cVi, RiT, -1.0);
    set3(e.J, 6, cB + 0, M.Jgv, -1.0);
    set3(e.J, 6, cB + 3, M.Jav, -1.0);
    set3(e.J, 6, cVj, RiT, 1.0);
    // rows 3..5: r_R
    const double* eR = &e.errI[3];
    double Jrinv[9], Rji[9];
    so3_JrInv(eR, Jrinv);
    Quat qji = quat_mul(qconj(sj.q), si.q);
    quat_normalize(qji);
    quat_to_R(qji, Rji);
    m3_mul(Jrinv, Rji, tmp);
    set3(e.J, 3, cPRi + 3, tmp, -1.0);
    const double meR[3] = {-eR[0], -eR[1], -eR[2]};
    double Ex[9], w[3], Jr[9];
    quat_to_R(so3_exp(meR), Ex);
    m3_v(M.JgR, si.dbg, w);
    so3_Jr(w, Jr);
    m3_mul(Jrinv, Ex, tmp);
    m3_mul(tmp, Jr, tmp2);
    m3_mul(tmp2, M.JgR, tmp);
    set3(e.J, 3, cB + 0, tmp, -1.0);
    set3(e.J, 3, cPRj + 3, Jrinv, 1.0);
  }
};

static void inc_pr(KF& s, const double* d) {  // NavState::IncSmall(dPR), P + R dp model
  double R[9], Rd[3];
  quat_to_R(s.q, R);
  m3_v(R, d, Rd);
  for (int i = 0; i < 3; i++) s.p[i] += Rd[i];
  s.q = quat_mul(s.q, so3_exp(d + 3));
  quat_normalize(s.q);
}



struct Sums {
  double last_trial_chi = 0;  // activeRobustChi2 of the errors left in the edges
};

// one SparseOptimizer::optimize(iterations) over the level-0 edges
static void optimize(W& B, int iterations, volatile const int* stop, vieo_lba_result& R, bool first, Sums& S) {
  const int nk = (int)B.kf.size(), nm = (int)B.mp_first.size();
  const float thI = std::sqrt(16.919), thB = std::sqrt(12.592);  // Optimizer.cc:219-222 (const float)
  std::vector<int> act;
  for (size_t i = 0; i < B.E.size(); i++)
    if (B.E[i].level == 0) act.push_back((int)i);
  std::vector<char> mp_act(nm, 0);
  for (int i : act) mp_act[B.E[i].mp] = 1;
  int np = 0;
  for (int k = 0; k < nk; k++) {
    if (!B.kf[k].fixed)
      B.kf[k].col = np, np += 15;
    else
      B.kf[k].col = -1;
  }
  // VertexScale: id_scale = maxKFid + 1 (Optimizer.cc:846), i.e. the last non-marginalised vertex
  const int sc = B.scale_opt ? np : -1;
  if (B.scale_opt) np += 1;
  if (np == 0) return;
  auto computeActiveErrors = [&]() {
    for (int i : act) B.v_error(B.E[i]);
    for (auto& e : B.I) B.i_error(e);
  };
  auto activeRobustChi2 = [&]() {
    double chi = 0, rho[2];
    for (int i : act) {
      const VEdge& e = B.E[i];
      if (e.robust) {
        hub(W::v_chi2(e), e.delta, e.dsqr, rho);
        chi += rho[0];
      } else
        chi += W::v_chi2(e);
    }
    for (const auto& e : B.I) {
      if (e.has_imu) {
        const double c = W::chi2_I(e);
        if (e.robust) {
          hub(c, (double)thI, (double)thI * (double)thI, rho);
          chi += rho[0];
        } else
          chi += c;
      }
      const double c = W::chi2_B(e);
      if (e.robust) {
        hub(c, (double)thB, (double)thB * (double)thB, rho);
        chi += rho[0];
      } else
        chi += c;
      if (e.has_enc) {
        const double ce = W::chi2_E(e);
        if (e.enc_robust) {
          hub(ce, (double)thB, (double)thB * (double)thB, rho);  // sqrt(12.592), Optimizer.cc:343
          chi += rho[0];
        } else
          chi += ce;
      }
    }
    return chi;
  };
  double lambda = -1, ni = 2;
  int nBad = 0;
  for (int it = 0; it < iterations; it++) {
    if (stop && *stop) break;
    R.lm_iterations++;
    computeActiveErrors();
    double currentChi = activeRobustChi2();
    if (first && it == 0) R.chi2_initial = currentChi;
    double tempChi = currentChi;
    const double iniChi = currentChi;
    // ---- buildSystem
    std::vector<double> H((size_t)np * np, 0.0), b(np, 0.0), Hll((size_t)nm * 9, 0.0), bl((size_t)nm * 3, 0.0);
    std::vector<double> Bpl(B.E.size() * 18, 0.0);
    std::vector<double> Bsl(sc >= 0 ? (size_t)nm * 3 : 0, 0.0);  // (scale, point) block, summed over the point's edges
    for (int i : act) {
      const VEdge& e = B.E[i];
      double Jp[18], Jx[9], Js[3] = {0, 0, 0};
      B.v_linearize(e, Jp, Jx, sc >= 0 ? Js : nullptr);
      double wr = 1.0;
      if (e.robust) {
        double rho[2];
        hub(W::v_chi2(e), e.delta, e.dsqr, rho);
        wr = rho[1];
      }
      const double w = wr * e.info;
      const int c = B.kf[e.kf].col;
      for (int a = 0; a < 3; a++) {
        for (int b2 = 0; b2 < 3; b2++) {
          double s = 0;
          for (int r = 0; r < e.de; r++) s += Jx[r * 3 + a] * w * Jx[r * 3 + b2];
          Hll[(size_t)e.mp * 9 + a * 3 + b2] += s;
        }
        double s = 0;
        for (int r = 0; r < e.de; r++) s += Jx[r * 3 + a] * (-(e.info * e.err[r]) * wr);
        bl[(size_t)e.mp * 3 + a] += s;
      }
      if (sc >= 0) {  // BaseMultiEdge::constructQuadraticForm over (point, PR, scale): the scale's own blocks
        double hss = 0, g = 0;
        for (int r = 0; r < e.de; r++) hss += Js[r] * w * Js[r], g += Js[r] * (-(e.info * e.err[r]) * wr);
        H[(size_t)sc * np + sc] += hss;
        b[sc] += g;
        for (int b2 = 0; b2 < 3; b2++) {
          double t = 0;
          for (int r = 0; r < e.de; r++) t += Js[r] * w * Jx[r * 3 + b2];
          Bsl[(size_t)e.mp * 3 + b2] += t;
        }
        if (c >= 0)
          for (int a = 0; a < 6; a++) {
            double t = 0;
            for (int r = 0; r < e.de; r++) t += Jp[r * 6 + a] * w * Js[r];
            H[(size_t)(c + a) * np + sc] += t;
            H[(size_t)sc * np + c + a] += t;
          }
      }
      if (c >= 0) {
        for (int a = 0; a < 6; a++) {
          for (int b2 = 0; b2 < 6; b2++) {
            double s = 0;
            for (int r = 0; r < e.de; r++) s += Jp[r * 6 + a] * w * Jp[r * 6 + b2];
            H[(size_t)(c + a) * np + c + b2] += s;
          }
          double s = 0;
          for (int r = 0; r < e.de; r++) s += Jp[r * 6 + a] * (-(e.info * e.err[r]) * wr);
          b[c + a] += s;
          for (int b2 = 0; b2 < 3; b2++) {
            double t = 0;
            for (int r = 0; r < e.de; r++) t += Jp[r * 6 + a] * w * Jx[r * 3 + b2];
            Bpl[(size_t)i * 18 + a * 3 + b2] = t;
          }
        }
      }
    }
    for (auto& e : B.I) {
      // system column of the 24 local Jacobian columns (-1: fixed vertex)
      int map[24];
      const int ci = B.kf[e.i].col, cj = B.kf[e.j].col;
      for (int k = 0; k < 6; k++) map[k] = ci >= 0 ? ci + k : -1, map[6 + k] = cj >= 0 ? cj + k : -1;
      for (int k = 0; k < 3; k++) map[12 + k] = ci >= 0 ? ci + 6 + k : -1, map[15 + k] = cj >= 0 ? cj + 6 + k : -1;
      for (int k = 0; k < 6; k++) map[18 + k] = ci >= 0 ? ci + 9 + k : -1;
      if (e.has_imu) {
        B.i_linearize(e);
        double rho[2] = {0, 1.0};
        if (e.robust) hub(W::chi2_I(e), (double)thI, (double)thI * (double)thI, rho);
        double we[9], T[9 * 24];
        for (int a = 0; a < 9; a++) {
          double t = 0;
          for (int q = 0; q < 9; q++) t += e.InfoI[a * 9 + q] * e.errI[q];
          we[a] = -t * rho[1];
          for (int c = 0; c < 24; c++) {
            double u = 0;
            for (int q = 0; q < 9; q++) u += (rho[1] * e.InfoI[a * 9 + q]) * e.J[q * 24 + c];
            T[a * 24 + c] = u;
          }
        }
        for (int c1 = 0; c1 < 24; c1++) {
          if (map[c1] < 0) continue;
          for (int c2 = 0; c2 < 24; c2++) {
            if (map[c2] < 0) continue;
            double t = 0;
            for (int a = 0; a < 9; a++) t += e.J[a * 24 + c1] * T[a * 24 + c2];
            H[(size_t)map[c1] * np + map[c2]] += t;
          }
          double t = 0;
          for (int a = 0; a < 9; a++) t += e.J[a * 24 + c1] * we[a];
          b[map[c1]] += t;
        }
      }
      {  // bias edge: J_i = -I, J_j = +I
        double rho[2] = {0, 1.0};
        if (e.robust) hub(W::chi2_B(e), (double)thB, (double)thB * (double)thB, rho);
        for (int k = 0; k < 6; k++) {
          const double w = (k < 3 ? e.infoBg : e.infoBa) * rho[1];
          const double we = (k < 3 ? e.infoBg : e.infoBa) * e.errB[k] * rho[1];
          const int ri = ci >= 0 ? ci + 9 + k : -1, rj = cj >= 0 ? cj + 9 + k : -1;
          if (ri >= 0) H[(size_t)ri * np + ri] += w, b[ri] += we;
          if (rj >= 0) H[(size_t)rj * np + rj] += w, b[rj] += -we;
          if (ri >= 0 && rj >= 0) H[(size_t)ri * np + rj] -= w, H[(size_t)rj * np + ri] -= w;
        }
      }
      if (e.has_enc) {  // encoder edge: 6 rows, columns PR_i (map 0..5) and PR_j (map 6..11)
        B.enc_eval(e, true);
        double rho[2] = {0, 1.0};
        if (e.enc_robust) hub(W::chi2_E(e), (double)thB, (double)thB * (double)thB, rho);
        double JE[6 * 12], TE[6 * 12], weE[6];
        for (int a = 0; a < 6; a++)
          for (int c = 0; c < 6; c++) JE[a * 12 + c] = e.JEi[a * 6 + c], JE[a * 12 + 6 + c] = e.JEj[a * 6 + c];
        for (int a = 0; a < 6; a++) {
          double t = 0;
          for (int q = 0; q < 6; q++) t += e.InfoE[a * 6 + q] * e.errE[q];
          weE[a] = -t * rho[1];
          for (int c = 0; c < 12; c++) {
            double u = 0;
            for (int q = 0; q < 6; q++) u += (rho[1] * e.InfoE[a * 6 + q]) * JE[q * 12 + c];
            TE[a * 12 + c] = u;
          }
        }
        for (int c1 = 0; c1 < 12; c1++) {
          if (map[c1] < 0) continue;
          for (int c2 = 0; c2 < 12; c2++) {
            if (map[c2] < 0) continue;
            double t = 0;
            for (int a = 0; a < 6; a++) t += JE[a * 12 + c1] * TE[a * 12 + c2];
            H[(size_t)map[c1] * np + map[c2]] += t;
          }
          double t = 0;
          for (int a = 0; a < 6; a++) t += JE[a * 12 + c1] * weE[a];
          b[map[c1]] += t;
        }
      }
    }
    if (it == 0) {
      lambda = B.P->lambda_init;  // computeLambdaInit: _userLambdaInit > 0
      if (B.gba) {  // tau * max |diagonal| over the pose-side and the landmark blocks
        double maxDiag = 0;
        for (int j = 0; j < np; j++) maxDiag = std::max(std::fabs(H[(size_t)j * np + j]), maxDiag);
        for (int m = 0; m < nm; m++)
          if (mp_act[m])
            for (int a = 0; a < 3; a++) maxDiag = std::max(std::fabs(Hll[(size_t)m * 9 + a * 4]), maxDiag);
        lambda = 1e-5 * maxDiag;
      }
      ni = 2;
      nBad = 0;
    }
    double rho = 0;
    int qmax = 0;
    do {
      R.lm_trials++;
      std::vector<KF> bk = B.kf;
      std::vector<double> bX = B.X;
      const double bscale = B.scale;
      std::vector<double> Hs = H, bs = b, Dinv((size_t)nm * 9, 0.0), xp(np, 0.0);
      for (int j = 0; j < np; j++) Hs[(size_t)j * np + j] += lambda;
      for (int m = 0; m < nm; m++) {
        if (!mp_act[m]) continue;
        double D[9];
        memcpy(D, &Hll[(size_t)m * 9], 72);
        D[0] += lambda, D[4] += lambda, D[8] += lambda;
        inv3(D, &Dinv[(size_t)m * 9]);
        const double* Di = &Dinv[(size_t)m * 9];
        double db[3];
        m3_v(Di, &bl[(size_t)m * 3], db);
        double SD[3] = {0, 0, 0};
        if (sc >= 0) {  // the scale vertex sees every point
          const double* Bs = &Bsl[(size_t)m * 3];
          for (int b2 = 0; b2 < 3; b2++) SD[b2] = Bs[0] * Di[b2] + Bs[1] * Di[3 + b2] + Bs[2] * Di[6 + b2];
          Hs[(size_t)sc * np + sc] -= SD[0] * Bs[0] + SD[1] * Bs[1] + SD[2] * Bs[2];
          bs[sc] -= Bs[0] * db[0] + Bs[1] * db[1] + Bs[2] * db[2];
        }
        for (int i1 = B.mp_first[m]; i1 < B.mp_first[m] + B.mp_count[m]; i1++) {
          const VEdge& e1 = B.E[i1];
          const int c1 = B.kf[e1.kf].col;
          if (e1.level != 0 || c1 < 0) continue;
          const double* B1 = &Bpl[(size_t)i1 * 18];
          double BD[18];
          for (int a = 0; a < 6; a++)
            for (int b2 = 0; b2 < 3; b2++)
              BD[a * 3 + b2] = B1[a * 3] * Di[b2] + B1[a * 3 + 1] * Di[3 + b2] + B1[a * 3 + 2] * Di[6 + b2];
          for (int a = 0; a < 6; a++)
            bs[c1 + a] -= B1[a * 3] * db[0] + B1[a * 3 + 1] * db[1] + B1[a * 3 + 2] * db[2];
          if (sc >= 0)
            for (int a = 0; a < 6; a++) {
              const double t = B1[a * 3] * SD[0] + B1[a * 3 + 1] * SD[1] + B1[a * 3 + 2] * SD[2];
              Hs[(size_t)(c1 + a) * np + sc] -= t;
              Hs[(size_t)sc * np + c1 + a] -= t;
            }
          for (int i2 = B.mp_first[m]; i2 < B.mp_first[m] + B.mp_count[m]; i2++) {
            const VEdge& e2 = B.E[i2];
            const int c2 = B.kf[e2.kf].col;
            if (e2.level != 0 || c2 < 0) continue;
            const double* B2 = &Bpl[(size_t)i2 * 18];
            for (int a = 0; a < 6; a++)
              for (int b2 = 0; b2 < 6; b2++)
                Hs[(size_t)(c1 + a) * np + c2 + b2] -=
                    BD[a * 3] * B2[b2 * 3] + BD[a * 3 + 1] * B2[b2 * 3 + 1] + BD[a * 3 + 2] * B2[b2 * 3 + 2];
          }
        }
      }
      bool ok2 = ldlt_solve(Hs.data(), bs.data(), xp.data(), np);
      std::vector<double> xl((size_t)nm * 3, 0.0);
      if (ok2) {
        for (int m = 0; m < nm; m++) {
          if (!mp_act[m]) continue;
          double cl[3] = {bl[(size_t)m * 3], bl[(size_t)m * 3 + 1], bl[(size_t)m * 3 + 2]};
          for (int i1 = B.mp_first[m]; i1 < B.mp_first[m] + B.mp_count[m]; i1++) {
            const VEdge& e1 = B.E[i1];
            const int c1 = B.kf[e1.kf].col;
            if (e1.level != 0 || c1 < 0) continue;
            const double* B1 = &Bpl[(size_t)i1 * 18];
            for (int b2 = 0; b2 < 3; b2++)
              for (int a = 0; a < 6; a++) cl[b2] -= B1[a * 3 + b2] * xp[c1 + a];
          }
          if (sc >= 0)
            for (int b2 = 0; b2 < 3; b2++) cl[b2] -= Bsl[(size_t)m * 3 + b2] * xp[sc];
          m3_v(&Dinv[(size_t)m * 9], cl, &xl[(size_t)m * 3]);
        }
      }
      for (int k = 0; k < nk; k++) {
        KF& s = B.kf[k];
        if (s.col < 0) continue;
        inc_pr(s, &xp[s.col]);
        for (int a = 0; a < 3; a++) s.v[a] += xp[s.col + 6 + a];
        for (int a = 0; a < 3; a++) s.dbg[a] += xp[s.col + 9 + a], s.dba[a] += xp[s.col + 12 + a];
      }
      if (sc >= 0) B.scale += xp[sc];  // VertexScale::oplusImpl (g2otypes.h:310)
      for (int m = 0; m < nm; m++)
        if (mp_act[m])
          for (int a = 0; a < 3; a++) B.X[(size_t)m * 3 + a] += xl[(size_t)m * 3 + a];
      computeActiveErrors();
      tempChi = activeRobustChi2();
      S.last_trial_chi = tempChi;
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      rho = currentChi - tempChi;
      double scale = 0;
      for (int j = 0; j < np; j++) scale += xp[j] * (lambda * xp[j] + b[j]);
      for (int m = 0; m < nm; m++)
        if (mp_act[m])
          for (int a = 0; a < 3; a++)
            scale += xl[(size_t)m * 3 + a] * (lambda * xl[(size_t)m * 3 + a] + bl[(size_t)m * 3 + a]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        lambda *= std::max(1. / 3., alpha);
        ni = 2;
        currentChi = tempChi;
      } else {
        lambda *= ni;
        ni *= 2;
        B.kf = bk;
        B.X = bX;
        B.scale = bscale;
      }
      qmax++;
    } while (rho < 0 && qmax < 10 && !(stop && *stop));
    if (qmax == 10 || rho == 0) break;
    if ((iniChi - currentChi) * 1e3 < iniChi)
      nBad++;
    else
      nBad = 0;
    if (nBad >= 3) break;
  }
}

// gba_iterations >= 0: Optimizer::GlobalBundleAdjustmentNavStatePRV (Optimizer.cc:771-1345; no scale / gravity
// vertex, no encoder edges) instead -- one optimize(), Huber on every edge iff gba_robust (sqrt(5.99) /
// sqrt(7.815) :1063-1064, sqrt(16.919) / sqrt(12.592) :910-911), no Chi2LargeSetLevel, no classification, no
// divergence guard
static void local_ba_vio(const vieo_lba_vio_params& P, const vieo_lba_keyframe* kfs, int n_kf,
                         const float* points, const uint8_t* close, int n_mp, const vieo_lba_obs* obs,
                         int n_obs, const vieo_lba_imu_edge* imu, int n_imu, volatile const int* stop,
                         vieo_navstate* navs_out, float* points_out, uint8_t* erase, vieo_lba_result& R,
                         int gba_iterations = -1, bool gba_robust = false, bool scale_opt = false,
                         double* scale_out = nullptr) {
  const bool gba = gba_iterations >= 0;
  if (scale_out) *scale_out = 1.;
  memset(&R, 0, sizeof(R));
  for (int k = 0; k < n_kf; k++) navs_out[k] = kfs[k].nav;
  memcpy(points_out, points, (size_t)n_mp * 12);
  memset(erase, 0, n_obs);
  W B;
  B.P = &P;
  B.gba = gba;
  B.scale_opt = gba && scale_opt;
  ocams_from_params(P.base, B.cams);
  B.kf.resize(n_kf);
  bool any_free = false;
  for (int k = 0; k < n_kf; k++) {
    KF& s = B.kf[k];
    const vieo_navstate& n = kfs[k].nav;
    memcpy(s.p, n.p, 24), memcpy(s.v, n.v, 24), memcpy(s.bg, n.bg, 24), memcpy(s.ba, n.ba, 24);
    memcpy(s.dbg, n.dbg, 24), memcpy(s.dba, n.dba, 24);
    s.q.w = n.q[0], s.q.x = n.q[1], s.q.y = n.q[2], s.q.z = n.q[3];
    s.fixed = kfs[k].fixed != 0;
    any_free |= !s.fixed;
  }
  if (B.scale_opt) any_free = true;  // bdimPoses = true with the scale vertex (Optimizer.cc:850)
  if (!any_free) {  // if (!bdimPoses) return;  Optimizer.cc:178
    R.status = VIEO_LBA_NO_FREE_POSE;
    return;
  }
  B.X.resize((size_t)n_mp * 3);
  for (int i = 0; i < n_mp * 3; i++) B.X[i] = (double)points[i];
  // inertial edges (Optimizer.cc:226-311)
  B.I.resize(n_imu);
  for (int t = 0; t < n_imu; t++) {
    IEdge& e = B.I[t];
    e.i = imu[t].kf_i, e.j = imu[t].kf_j, e.M = &imu[t].imu;
    const bool bfixedkf = B.kf[e.i].fixed;
    e.has_imu = e.M->dt != 0;
    e.robust = gba ? gba_robust : (bfixedkf || P.rec_init);
    if (e.has_imu) {
      mat_inverse(e.M->Sigma, e.InfoI, 9);  // GetProcessedInfoijPRV
      if (bfixedkf)
        for (int k = 0; k < 81; k++) e.InfoI[k] *= 1e-2;
    }
    double deltatij = e.M->dt ? e.M->dt : imu[t].dt_kf;
    const float EPS_MIN_DT = 1e-6f;
    if (deltatij <= EPS_MIN_DT) deltatij = 15;
    e.infoBg = P.inv_sigma_bg2 / deltatij * (bfixedkf ? 1e-2 : 1.0);
    e.infoBa = P.inv_sigma_ba2 / deltatij * (bfixedkf ? 1e-2 : 1.0);
    memset(e.errI, 0, sizeof(e.errI)), memset(e.errB, 0, sizeof(e.errB));
    e.has_enc = imu[t].enc.dt != 0;
    if (e.has_enc) {
      memcpy(e.measE, imu[t].enc.delx, 48);
      mat_inverse(imu[t].enc.Sigma, e.InfoE, 6);
      if (bfixedkf)
        for (int k = 0; k < 36; k++) e.InfoE[k] *= 1e-2;
      e.enc_robust = gba ? gba_robust : true;
      memset(e.errE, 0, sizeof(e.errE));
    }
  }
  B.E.resize(n_obs);
  B.mp_first.assign(n_mp, 0);
  B.mp_count.assign(n_mp, 0);
  const float chi2Mono = 5.991;
  const float thHuberMono = gba ? (float)sqrt(5.99) : sqrt(chi2Mono), thHuberStereo = sqrt(7.815);
  for (int i = 0; i < n_obs; i++) {
    VEdge& e = B.E[i];
    if (gba) e.robust = gba_robust;
    e.kf = obs[i].kf & 0xFFFFFF, e.cam = (obs[i].kf >> 24) & 15, e.mp = obs[i].mp;
    e.obs[0] = obs[i].u, e.obs[1] = obs[i].v, e.obs[2] = obs[i].ur;
    e.de = obs[i].ur < 0 ? 2 : 3;
    e.info = (double)obs[i].inv_sigma2;
    e.delta = e.de == 2 ? (double)thHuberMono : (double)thHuberStereo;
    e.dsqr = e.delta * e.delta;
    if (B.mp_count[e.mp] == 0) B.mp_first[e.mp] = i;
    B.mp_count[e.mp]++;
  }
  if (stop && *stop) {  // Optimizer.cc:524-528
    R.status = VIEO_LBA_ABORTED;
    return;
  }
  // th_dist_far (Optimizer.cc:395,454,513-517): a point without a monocular edge closer than the limit loses its
  // monocular edges (level 1); GetDepth() = third row of Rcw * Xw + tcw at the initial estimates
  if (!gba && P.th_dist_far > 0 && std::isfinite(P.th_dist_far)) {
    for (int m = 0; m < n_mp; m++) {
      bool ok = false, any_mono = false;
      for (int i = B.mp_first[m]; i < B.mp_first[m] + B.mp_count[m]; i++) {
        VEdge& e = B.E[i];
        if (e.de != 2) continue;
        any_mono = true;
        double proj[3], Pc[3];
        B.project(e, proj, Pc, nullptr);
        if (Pc[2] < (double)P.th_dist_far) ok = true;
      }
      if (any_mono && !ok)
        for (int i = B.mp_first[m]; i < B.mp_first[m] + B.mp_count[m]; i++)
          if (B.E[i].de == 2) B.E[i].level = 1;
    }
  }
  // Chi2LargeSetLevel (rat_vis_check = 100): chi2_sig5_[2] = 5.991f, [3] = 7.815f, float product
  for (auto& e : B.E) {
    if (gba) break;
    B.v_error(e);
    const float th = 100.f * (e.de == 2 ? 5.991f : 7.815f);
    if (W::v_chi2(e) > th) e.level = 1;
  }
  Sums S;
  optimize(B, gba ? gba_iterations : P.base.its0, stop, R, true, S);
  const float err = (float)R.chi2_initial;
  bool bDoMore = !(stop && *stop);
  auto bad = [&](const VEdge& e) {
    if (e.de == 2)
      return W::v_chi2(e) > (close[e.mp] ? 1.5 * chi2Mono : chi2Mono) || !B.depth_positive(e);
    return W::v_chi2(e) > 7.815 || !B.depth_positive(e);
  };
  if (bDoMore && !gba) {
    for (auto& e : B.E) {
      if (bad(e)) e.level = 1;
      e.robust = false;
    }
    optimize(B, P.base.its1, stop, R, false, S);
  }
  const float err_end = (float)S.last_trial_chi;
  R.chi2_final = err_end;
  R.chi2_initial = err;
  if ((2 * err < err_end || std::isnan(err) || std::isnan(err_end)) && !P.large && !gba) {  // Optimizer.cc:660-666
    R.status = VIEO_LBA_DIVERGED;
    return;
  }
  for (int i = 0; i < n_obs && !gba; i++)
    if (bad(B.E[i])) erase[i] = 1, R.n_erase++;
  for (int k = 0; k < n_kf; k++) {
    const KF& s = B.kf[k];
    if (s.fixed) continue;
    vieo_navstate& n = navs_out[k];
    memcpy(n.p, s.p, 24), memcpy(n.v, s.v, 24), memcpy(n.dbg, s.dbg, 24), memcpy(n.dba, s.dba, 24);
    n.q[0] = s.q.w, n.q[1] = s.q.x, n.q[2] = s.q.y, n.q[3] = s.q.z;
  }
  if (B.scale_opt) {  // SetWorldPos(scale * vPoint->estimate().cast<float>()): the scalar meets a float vector
    const float sf = (float)B.scale;
    for (int i = 0; i < n_mp * 3; i++) points_out[i] = sf * (float)B.X[i];
    if (scale_out) *scale_out = B.scale;
  } else
    for (int i = 0; i < n_mp * 3; i++) points_out[i] = (float)B.X[i];
}

}  // namespace vov

extern "C" {

void vo_local_bundle_adjustment_vio(const vieo_lba_vio_params* params, const vieo_lba_keyframe* kfs, int n_kf,
                                    const float* points, const uint8_t* close, int n_mp,
                                    const vieo_lba_obs* obs, int n_obs, const vieo_lba_imu_edge* imu, int n_imu,
                                    const int* stop, vieo_navstate* navs_out, float* points_out,
                                    uint8_t* erase, vieo_lba_result* result) {
  vov::local_ba_vio(*params, kfs, n_kf, points, close, n_mp, obs, n_obs, imu, n_imu, stop, navs_out,
                    points_out, erase, *result);
}

// test hook: error (9 + 6) and Jacobian (9 x 24) of one inertial edge at the given states
void vo_global_bundle_adjustment_vio(const vieo_lba_vio_params* params, int n_iterations, int robust,
                                     const vieo_lba_keyframe* kfs, int n_kf, const float* points, int n_mp,
                                     const vieo_lba_obs* obs, int n_obs, const vieo_lba_imu_edge* imu, int n_imu,
                                     const int* stop, vieo_navstate* navs_out, float* points_out,
                                     vieo_lba_result* result) {
  std::vector<uint8_t> erase((size_t)n_obs + 1), close((size_t)n_mp + 1, 0);
  vov::local_ba_vio(*params, kfs, n_kf, points, close.data(), n_mp, obs, n_obs, imu, n_imu, stop, navs_out,
                    points_out, erase.data(), *result, n_iterations, robust != 0);
}

// System::FinalGBA's form (src/System.cc:24-33): bScaleOpt = true.  scale_out: the recovered VertexScale estimate.
void vo_global_bundle_adjustment_vio_scale(const vieo_lba_vio_params* params, int n_iterations, int robust,
                                           int scale_opt, const vieo_lba_keyframe* kfs, int n_kf, const float* points,
                                           int n_mp, const vieo_lba_obs* obs, int n_obs, const vieo_lba_imu_edge* imu,
                                           int n_imu, const int* stop, vieo_navstate* navs_out, float* points_out,
                                           vieo_lba_result* result, double* scale_out) {
  std::vector<uint8_t> erase((size_t)n_obs + 1), close((size_t)n_mp + 1, 0);
  vov::local_ba_vio(*params, kfs, n_kf, points, close.data(), n_mp, obs, n_obs, imu, n_imu, stop, navs_out,
                    points_out, erase.data(), *result, n_iterations, robust != 0, scale_opt != 0, scale_out);
}

// test hook: error and Jacobians of one EdgeReprojectPRS / PRSStereo (ur < 0: 2 rows) at a key-frame state, an
// unscaled point and a scale.  Jp [3][6] (dp, dphi), Jx [3][3], Js [3]
void vo_lba_prs_edge_eval(const vieo_lba_vio_params* params, const vieo_navstate* ns, const double* Xh, double scale,
                          const vieo_lba_obs* obs, double* err3, double* Jp, double* Jx, double* Js) {
  vov::W B;
  B.P = params;
  B.scale_opt = true, B.scale = scale;
  vov::ocams_from_params(params->base, B.cams);
  B.kf.resize(1);
  vov::KF& s = B.kf[0];
  memcpy(s.p, ns->p, 24);
  s.q.w = ns->q[0], s.q.x = ns->q[1], s.q.y = ns->q[2], s.q.z = ns->q[3];
  s.fixed = false, s.col = 0;
  B.X.assign(Xh, Xh + 3);
  vov::VEdge e;
  e.kf = 0, e.mp = 0, e.cam = (obs->kf >> 24) & 15;
  e.obs[0] = obs->u, e.obs[1] = obs->v, e.obs[2] = obs->ur;
  e.de = obs->ur < 0 ? 2 : 3;
  e.info = (double)obs->inv_sigma2;
  B.v_error(e);
  memcpy(err3, e.err, 24);
  if (Jp) {
    double jp[18] = {0}, jx[9] = {0}, js[3] = {0};
    B.v_linearize(e, jp, jx, js);
    memcpy(Jp, jp, sizeof(jp)), memcpy(Jx, jx, sizeof(jx)), memcpy(Js, js, sizeof(js));
  }
}

void vo_enc_edge_eval(const vieo_navstate* nsi, const vieo_navstate* nsj, const double* meas6, const double* qRbe4,
                      const double* pbe3, double* err6, double* Ji36, double* Jj36) {
  vo::EncPose a, b;
  memcpy(a.p, nsi->p, 24), memcpy(b.p, nsj->p, 24);
  a.q.w = nsi->q[0], a.q.x = nsi->q[1], a.q.y = nsi->q[2], a.q.z = nsi->q[3];
  b.q.w = nsj->q[0], b.q.x = nsj->q[1], b.q.y = nsj->q[2], b.q.z = nsj->q[3];
  vo::Quat qbe;
  qbe.w = qRbe4[0], qbe.x = qRbe4[1], qbe.y = qRbe4[2], qbe.z = qRbe4[3];
  vo::enc_edge_eval(a, b, meas6, qbe, pbe3, err6, Ji36, Jj36);
}

void vo_lba_imu_edge_eval(const vieo_lba_vio_params* params, const vieo_lba_imu_edge* edge,
                          const vieo_navstate* nsi, const vieo_navstate* nsj, double* err15, double* J) {
  vov::W B;
  B.P = params;
  B.kf.resize(2);
  const vieo_navstate* ns[2] = {nsi, nsj};
  for (int k = 0; k < 2; k++) {
    vov::KF& s = B.kf[k];
    const vieo_navstate& n = *ns[k];
    memcpy(s.p, n.p, 24), memcpy(s.v, n.v, 24), memcpy(s.bg, n.bg, 24), memcpy(s.ba, n.ba, 24);
    memcpy(s.dbg, n.dbg, 24), memcpy(s.dba, n.dba, 24);
    s.q.w = n.q[0], s.q.x = n.q[1], s.q.y = n.q[2], s.q.z = n.q[3];
    s.fixed = false, s.col = 0;
  }
  vov::IEdge e;
  e.i = 0, e.j = 1, e.M = &edge->imu, e.has_imu = true, e.robust = false;
  B.i_error(e);
  memcpy(err15, e.errI, 72), memcpy(err15 + 9, e.errB, 48);
  if (J) {
    B.i_linearize(e);
    memcpy(J, e.J, sizeof(e.J));
  }
}

// test hook: VertexNavStatePR / V / Bias oplus
void vo_lba_navstate_inc(vieo_navstate* n, const double* d15) {
  vov::KF s;
  memcpy(s.p, n->p, 24);
  s.q.w = n->q[0], s.q.x = n->q[1], s.q.y = n->q[2], s.q.z = n->q[3];
  vov::inc_pr(s, d15);
  memcpy(n->p, s.p, 24);
  n->q[0] = s.q.w, n->q[1] = s.q.x, n->q[2] = s.q.y, n->q[3] = s.q.z;
  for (int a = 0; a < 3; a++) n->v[a] += d15[6 + a], n->dbg[a] += d15[9 + a], n->dba[a] += d15[12 + a];
}

}  // extern "C"
