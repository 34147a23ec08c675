// ORACLE -- TEST INFRASTRUCTURE ONLY (see ocv_prims.hpp header).  PARITY UNPINNED.
//
// CPU restatement of the visual-inertial motion BA
//   template<class KeyFrame> int Optimizer::PoseOptimization(Frame*, KeyFrame*, gw, bComputeMarg, bNoMPs)
//   (reference: include/Optimizer.h:208-816, marginal prior FillCovInv :126-206)
// with its edges restated in the reference's order of operations:
//   EdgeNavStateI<3> = EdgeNavStatePVR      src/Odom/g2otypes.h:703-884   (9-D IMU residual + Jacobians)
//   EdgeNavStateBias                        src/Odom/g2otypes.cpp:14-34
//   EdgeNavStatePriorPVRBias                src/Odom/g2otypes.cpp:84-124
//   EdgeReprojectPVR / PVRStereo            src/Odom/g2otypes.h:321-547
//   NavState::IncSmall(dPVR), IncSmallBias  src/Odom/NavState.h:64-83
//   EdgeEncNavState<9> = EdgeEncNavStatePVR src/Odom/g2otypes.h:591-668   (enc_edge.hpp; Optimizer.h:345-363)
// and the g2o machinery as in pose_opt.cc (LM, Huber, dense LDLT).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "../include/vieo_hot.h"
#include "cam_models.hpp"
#include "enc_edge.hpp"
#include "smallmat.hpp"

namespace vo {

struct NS {  // NavState
  double p[3], v[3];
  Quat q;
  double bg[3], ba[3], dbg[3], dba[3];
};
static NS ns_from(const vieo_navstate& n) {
  NS s;
  memcpy(s.p, n.p, 24), memcpy(s.v, n.v, 24);
  s.q.w = n.q[0], s.q.x = n.q[1], s.q.y = n.q[2], s.q.z = n.q[3];
  memcpy(s.bg, n.bg, 24), memcpy(s.ba, n.ba, 24), memcpy(s.dbg, n.dbg, 24), memcpy(s.dba, n.dba, 24);
  return s;
}
static void ns_to(const NS& s, vieo_navstate& n) {
  memcpy(n.p, s.p, 24), memcpy(n.v, s.v, 24);
  n.q[0] = s.q.w, n.q[1] = s.q.x, n.q[2] = s.q.y, n.q[3] = s.q.z;
  memcpy(n.bg, s.bg, 24), memcpy(n.ba, s.ba, 24), memcpy(n.dbg, s.dbg, 24), memcpy(n.dba, s.dba, 24);
}
// NavState::IncSmall(dPVR) (NavState.h:64-79)
static void inc_pvr(NS& s, const double* d) {
  double R[9], Rd[3];
  quat_to_R(s.q, R);
  m3_v(R, d, Rd);
  for (int i = 0; i < 3; i++) s.p[i] += Rd[i];
  for (int i = 0; i < 3; i++) s.v[i] += d[3 + i];
  s.q = quat_mul(s.q, so3_exp(d + 6));
  quat_normalize(s.q);
}
static void inc_bias(NS& s, const double* d) {  // NavState.h:80-83
  for (int i = 0; i < 3; i++) s.dbg[i] += d[i], s.dba[i] += d[3 + i];
}
static Quat quat_conj(const Quat& q) {
  Quat r = q;
  r.x = -q.x, r.y = -q.y, r.z = -q.z;
  return r;
}

// general n x n inverse by Gauss-Jordan with partial pivoting (Eigen inverse() / PartialPivLU)
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

struct Dense {  // small row-major matrix
  int r, c;
  std::vector<double> a;
  Dense(int r_ = 0, int c_ = 0) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
  double& operator()(int i, int j) { return a[(size_t)i * c + j]; }
  double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
  void set3(int i0, int j0, const double* M, double s = 1.0) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) (*this)(i0 + i, j0 + j) = s * M[i * 3 + j];
  }
};

struct GEdge {  // a generic edge of the 4-vertex problem (vertex ids 0 PVRj, 1 Bj, 2 PVRi, 3 Bi)
  int D = 0;
  int nv = 0;
  int vid[3];
  std::vector<double> err;
  Dense info;
  Dense J[3];
  bool robust = false;
  double delta = 0, dsqr = 0;
  double chi2() const {  // _error.dot(information()*_error)
    double s = 0;
    for (int i = 0; i < D; i++) {
      double t = 0;
      for (int j = 0; j < D; j++) t += info(i, j) * err[j];
      s += err[i] * t;
    }
    return s;
  }
};

static void huber2(double e, double delta, double dsqr, double* rho) {
  if (e <= dsqr) {
    rho[0] = e, rho[1] = 1., rho[2] = 0.;
  } else {
    double sqrte = std::sqrt(e);
    rho[0] = 2 * sqrte * delta - dsqr;
    rho[1] = delta / sqrte;
    rho[2] = -0.5 * rho[1] / e;
  }
}

struct VisEdge {
  double Xw[3], obs[3], info;
  int de, level = 0, idx, cam = 0;
  bool robust = true, close = false;
  double delta, dsqr;
  double err[3] = {0, 0, 0};
};

struct Problem {
  const vieo_vio_frame* F;
  OCam cams[4];  // cams[0] = the rectified pinhole camera when n_cams == 0 (a20)
  NS nsj, nsi;       // current estimates
  NS prior;          // measurement of the prior edge
  bool fixedLast, hasImu;
  double InfoI[81];  // IMU information (already scaled)
  GEdge eI, eB, eP, eE;
  bool hasEnc = false;  // EdgeEncNavStatePVR between PVRi (vertex 0 of the edge) and PVRj
  Quat qRbe;
  double pbe[3], measE[6];
  std::vector<VisEdge> vis;
  int ndim;  // 15 or 30

  // ---- visual edge (EdgeReprojectPVR[Stereo])
  void vis_project(const double* Xw, int de, double* proj, double* Pc_out, double* Rcw_out, int ci = 0) const {
    const OCam& C = cams[ci];
    double Rwb[9], Rbw[9], Rcw[9], t[3], tcw[3], Pc[3];
    quat_to_R(nsj.q, Rwb);
    m3_T(Rwb, Rbw);
    m3_mul(C.Rcb, Rbw, Rcw);
    m3_v(Rcw, nsj.p, t);
    for (int i = 0; i < 3; i++) tcw[i] = -t[i] + C.tcb[i];
    m3_v(Rcw, Xw, Pc);
    for (int i = 0; i < 3; i++) Pc[i] += tcw[i];
    float uv[2];
    ocam_project(C, Pc, uv, nullptr);
    proj[0] = uv[0];
    proj[1] = uv[1];
    if (de > 2) proj[2] = proj[0] - (double)C.bf / Pc[2];
    if (Pc_out) memcpy(Pc_out, Pc, 24);
    if (Rcw_out) memcpy(Rcw_out, Rcw, 72);
  }
  void vis_error(VisEdge& e) const {
    double proj[3];
    vis_project(e.Xw, e.de, proj, nullptr, nullptr, e.cam);
    for (int i = 0; i < e.de; i++) e.err[i] = e.obs[i] - proj[i];
  }
  static double vis_chi2(const VisEdge& e) {
    double s = 0;
    for (int i = 0; i < e.de; i++) s += e.err[i] * (e.info * e.err[i]);
    return s;
  }
  bool vis_depth_positive(const VisEdge& e) const {  // g2otypes.h:421-427
    double proj[3], Pc[3];
    vis_project(e.Xw, e.de, proj, Pc, nullptr, e.cam);
    return Pc[2] > 0.;
  }
  void vis_linearize(const VisEdge& e, double* J /* de x 9 */) const {
    double proj[3], Pc[3], Rcw[9];
    vis_project(e.Xw, e.de, proj, Pc, Rcw, e.cam);
    const OCam& C = cams[e.cam];
    const double* Rcb = C.Rcb;
    const double invz = 1 / Pc[2], invz_2 = invz * invz;
    double Jp[9] = {0}, Jc[6];
    ocam_project(C, Pc, nullptr, Jc);
    for (int i = 0; i < 6; i++) Jp[i] = -Jc[i];
    if (e.de > 2) Jp[6] = Jp[0], Jp[7] = Jp[1], Jp[8] = Jp[2] - (double)C.bf * invz_2;
    (void)invz;
    double Rwb[9], dP[3], Paux[3], H[9], RcbH[9];
    quat_to_R(nsj.q, Rwb);
    for (int i = 0; i < 3; i++) dP[i] = e.Xw[i] - nsj.p[i];
    m3T_v(Rwb, dP, Paux);
    hat(Paux, H);
    m3_mul(Rcb, H, RcbH);
    for (int r = 0; r < e.de; r++)
      for (int k = 0; k < 3; k++) {
        double a = 0, b = 0;
        for (int m = 0; m < 3; m++) {
          a += Jp[r * 3 + m] * (-Rcb[m * 3 + k]);
          b += Jp[r * 3 + m] * RcbH[m * 3 + k];
        }
        J[r * 9 + k] = a;
        J[r * 9 + 3 + k] = 0;
        J[r * 9 + 6 + k] = b;
      }
  }

  // ---- EdgeNavStatePVR (g2otypes.h:733-776)
  void imu_error() {
    const vieo_imu_preint& M = F->imu;
    double Ri[9], RiT[9];
    quat_to_R(nsi.q, Ri);
    m3_T(Ri, RiT);
    const double dt = M.dt;
    double t[3], r[3], Jb[3], Ja[3];
    for (int k = 0; k < 3; k++) t[k] = nsj.p[k] - nsi.p[k] - nsi.v[k] * dt - F->gw[k] * (dt * dt / 2);
    m3_v(RiT, t, r);
    m3_v(M.Jgp, nsi.dbg, Jb);
    m3_v(M.Jap, nsi.dba, Ja);
    for (int k = 0; k < 3; k++) eI.err[k] = r[k] - (M.pij[k] + Jb[k] + Ja[k]);
    // eR = Log((dRij * Exp(JgR*dbg))^-1 * (Ri^-1 * Rj))
    double w[3];
    m3_v(M.JgR, nsi.dbg, w);
    Quat qa = quat_mul(R_to_quat(M.Rij), so3_exp(w));
    quat_normalize(qa);
    Quat qb = quat_mul(quat_conj(nsi.q), nsj.q);
    quat_normalize(qb);
    Quat qe = quat_mul(quat_conj(qa), qb);
    quat_normalize(qe);
    so3_log(qe, &eI.err[6]);
    for (int k = 0; k < 3; k++) t[k] = nsj.v[k] - nsi.v[k] - F->gw[k] * dt;
    m3_v(RiT, t, r);
    m3_v(M.Jgv, nsi.dbg, Jb);
    m3_v(M.Jav, nsi.dba, Ja);
    for (int k = 0; k < 3; k++) eI.err[3 + k] = r[k] - (M.vij[k] + Jb[k] + Ja[k]);
  }
  // g2otypes.h:777-884, PVR layout: idR = 6, idV = 3
  void imu_linearize() {
    const vieo_imu_preint& M = F->imu;
    const int idR = 6, idV = 3;
    double Ri[9], RiT[9], Rj[9];
    quat_to_R(nsi.q, Ri);
    m3_T(Ri, RiT);
    quat_to_R(nsj.q, Rj);
    const double dt = M.dt;
    Dense &Ji = eI.J[0], &Jj = eI.J[1], &JB = eI.J[2];
    Ji = Dense(9, 9), Jj = Dense(9, 9), JB = Dense(9, 6);
    const double* eR = &eI.err[idR];
    double t[3], r[3], Hm[9], I3[9], tmp[9], tmp2[9];
    m3_identity(I3);
    for (int k = 0; k < 3; k++) t[k] = nsj.p[k] - nsi.p[k] - nsi.v[k] * dt - F->gw[k] * (dt * dt / 2);
    m3_v(RiT, t, r);
    hat(r, Hm);
    Ji.set3(0, idR, Hm);
    Ji.set3(0, 0, I3, -1.0);
    Ji.set3(0, idV, RiT, -dt);
    JB.set3(0, 0, M.Jgp, -1.0);
    JB.set3(0, 3, M.Jap, -1.0);
    m3_mul(RiT, Rj, tmp);
    Jj.set3(0, 0, tmp);
    for (int k = 0; k < 3; k++) t[k] = nsj.v[k] - nsi.v[k] - F->gw[k] * dt;
    m3_v(RiT, t, r);
    hat(r, Hm);
    Ji.set3(idV, idR, Hm);
    Ji.set3(idV, idV, RiT, -1.0);
    JB.set3(idV, 0, M.Jgv, -1.0);
    JB.set3(idV, 3, M.Jav, -1.0);
    Jj.set3(idV, idV, RiT);
    double Jrinv[9];
    so3_JrInv(eR, Jrinv);
    // -Jrinv * (Rj^-1 * Ri)
    Quat qji = quat_mul(quat_conj(nsj.q), nsi.q);
    quat_normalize(qji);
    double Rji[9];
    quat_to_R(qji, Rji);
    m3_mul(Jrinv, Rji, tmp);
    Ji.set3(idR, idR, tmp, -1.0);
    // -Jrinv * Exp(-eR) * Jr(JgR*dbg) * JgR
    double meR[3] = {-eR[0], -eR[1], -eR[2]}, ExpmeR[9], w[3], Jr[9];
    quat_to_R(so3_exp(meR), ExpmeR);
    m3_v(M.JgR, nsi.dbg, w);
    so3_Jr(w, Jr);
    m3_mul(Jrinv, ExpmeR, tmp);
    m3_mul(tmp, Jr, tmp2);
    m3_mul(tmp2, M.JgR, tmp);
    JB.set3(idR, 0, tmp, -1.0);
    Jj.set3(idR, idR, Jrinv);
  }
  // EdgeNavStateBias (g2otypes.cpp:14-34): v0 = Bi, v1 = Bj
  void bias_error() {
    for (int k = 0; k < 3; k++) {
      eB.err[k] = (nsj.bg[k] + nsj.dbg[k]) - (nsi.bg[k] + nsi.dbg[k]);
      eB.err[3 + k] = (nsj.ba[k] + nsj.dba[k]) - (nsi.ba[k] + nsi.dba[k]);
    }
  }
  // EdgeNavStatePriorPVRBias (g2otypes.cpp:84-124): v0 = PVRi, v1 = Bi
  void prior_error() {
    double Rb[9], RbT[9], d[3], r[3];
    quat_to_R(prior.q, Rb);
    m3_T(Rb, RbT);
    for (int k = 0; k < 3; k++) d[k] = nsi.p[k] - prior.p[k];
    m3_v(RbT, d, r);
    for (int k = 0; k < 3; k++) eP.err[k] = r[k];
    Quat qe = quat_mul(quat_conj(prior.q), nsi.q);
    quat_normalize(qe);
    so3_log(qe, &eP.err[6]);
    for (int k = 0; k < 3; k++) {
      eP.err[3 + k] = nsi.v[k] - prior.v[k];
      eP.err[9 + k] = nsi.bg[k] + nsi.dbg[k] - (prior.bg[k] + prior.dbg[k]);
      eP.err[12 + k] = nsi.ba[k] + nsi.dba[k] - (prior.ba[k] + prior.dba[k]);
    }
  }
  void prior_linearize() {
    Dense &Jx = eP.J[0], &Jb = eP.J[1];
    Jx = Dense(15, 9), Jb = Dense(15, 6);
    double Rb[9], RbT[9], Ri[9], tmp[9], I3[9], Jrinv[9];
    m3_identity(I3);
    quat_to_R(prior.q, Rb);
    m3_T(Rb, RbT);
    quat_to_R(nsi.q, Ri);
    m3_mul(RbT, Ri, tmp);
    Jx.set3(0, 0, tmp);
    Jx.set3(3, 3, I3);
    so3_JrInv(&eP.err[6], Jrinv);
    Jx.set3(6, 6, Jrinv);
    Jb.set3(9, 0, I3);
    Jb.set3(12, 3, I3);
  }

  // column offset of vertex id in the Hessian (-1 = fixed)
  int col(int vid) const {
    static const int off[4] = {0, 9, 15, 24};
    if (vid >= 2 && fixedLast) return -1;
    return off[vid];
  }
  static int vdim(int vid) { return (vid & 1) ? 6 : 9; }

  void apply_update(const double* x) {
    inc_pvr(nsj, x);
    inc_bias(nsj, x + 9);
    if (!fixedLast) {
      inc_pvr(nsi, x + 15);
      inc_bias(nsi, x + 24);
    }
  }
  // ---- encoder edge: enc_edge.hpp gives (dp, dphi) columns; PVR vertices hold them at 0..2 and 6..8
  void enc_eval(bool jac) {
    EncPose si, sj;
    memcpy(si.p, nsi.p, 24), memcpy(sj.p, nsj.p, 24);
    si.q = nsi.q, sj.q = nsj.q;
    double Ji[36], Jj[36];
    enc_edge_eval(si, sj, measE, qRbe, pbe, eE.err.data(), jac ? Ji : nullptr, jac ? Jj : nullptr);
    if (!jac) return;
    eE.J[0] = Dense(6, 9), eE.J[1] = Dense(6, 9);
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < 3; c++) {
        eE.J[0](r, c) = Ji[r * 6 + c], eE.J[0](r, 6 + c) = Ji[r * 6 + 3 + c];
        eE.J[1](r, c) = Jj[r * 6 + c], eE.J[1](r, 6 + c) = Jj[r * 6 + 3 + c];
      }
  }
  std::vector<GEdge*> generic_edges() {
    std::vector<GEdge*> g;
    if (hasImu) g.push_back(&eI);
    g.push_back(&eB);
    if (!fixedLast) g.push_back(&eP);
    if (hasEnc) g.push_back(&eE);
    return g;
  }
  void compute_generic_errors() {
    if (hasImu) imu_error();
    bias_error();
    if (!fixedLast) prior_error();
    if (hasEnc) enc_eval(false);
  }
  void linearize_generic() {
    if (hasImu) imu_linearize();
    if (!fixedLast) prior_linearize();
    if (hasEnc) enc_eval(true);
  }
};

static void accumulate(Problem& P, const GEdge& e, std::vector<double>& H, std::vector<double>& b) {
  const int n = P.ndim;
  double rho1 = 1.0;
  if (e.robust) {
    double rho[3];
    huber2(e.chi2(), e.delta, e.dsqr, rho);
    rho1 = rho[1];
  }
  // omega_r = -(info * err) * rho1 ; W = rho1 * info
  std::vector<double> wr(e.D);
  for (int i = 0; i < e.D; i++) {
    double t = 0;
    for (int j = 0; j < e.D; j++) t += e.info(i, j) * e.err[j];
    wr[i] = -t * rho1;
  }
  for (int a = 0; a < e.nv; a++) {
    const int ca = P.col(e.vid[a]);
    if (ca < 0) continue;
    const Dense& A = e.J[a];
    // AtO = A^T * W
    Dense AtO(A.c, e.D);
    for (int i = 0; i < A.c; i++)
      for (int j = 0; j < e.D; j++) {
        double s = 0;
        for (int k = 0; k < e.D; k++) s += A(k, i) * (rho1 * e.info(k, j));
        AtO(i, j) = s;
      }
    for (int i = 0; i < A.c; i++) {
      double s = 0;
      for (int k = 0; k < e.D; k++) s += A(k, i) * wr[k];
      b[ca + i] += s;
    }
    for (int bb = a; bb < e.nv; bb++) {
      const int cb = P.col(e.vid[bb]);
      if (cb < 0) continue;
      const Dense& B = e.J[bb];
      for (int i = 0; i < A.c; i++)
        for (int j = 0; j < B.c; j++) {
          double s = 0;
          for (int k = 0; k < e.D; k++) s += AtO(i, k) * B(k, j);
          H[(size_t)(ca + i) * n + cb + j] += s;
          if (bb != a) H[(size_t)(cb + j) * n + ca + i] += s;
        }
    }
  }
}

struct LM2 {
  double lambda = -1, ni = 2;
  int nBad = 0;
  int trials = 0;  // lambda trials of this optimize() (diagnostic: vieo_pose_result.reserved)
};

static int lm_solve_vio(Problem& P, std::vector<VisEdge*>& active, int iteration, LM2& lm) {
  const int n = P.ndim;
  auto gen = P.generic_edges();
  auto computeActiveErrors = [&]() {
    P.compute_generic_errors();
    for (auto* e : active) P.vis_error(*e);
  };
  auto activeRobustChi2 = [&]() {
    double chi = 0, rho[3];
    for (auto* g : gen) {
      if (g->robust) {
        huber2(g->chi2(), g->delta, g->dsqr, rho);
        chi += rho[0];
      } else
        chi += g->chi2();
    }
    for (auto* e : active) {
      if (e->robust) {
        huber2(Problem::vis_chi2(*e), e->delta, e->dsqr, rho);
        chi += rho[0];
      } else
        chi += Problem::vis_chi2(*e);
    }
    return chi;
  };
  computeActiveErrors();
  double currentChi = activeRobustChi2();
  double tempChi = currentChi;
  const double iniChi = currentChi;
  std::vector<double> H((size_t)n * n, 0.0), b(n, 0.0);
  P.linearize_generic();
  for (auto* g : gen) accumulate(P, *g, H, b);
  for (auto* e : active) {
    double J[27];
    P.vis_linearize(*e, J);
    double wr = 1.0;
    if (e->robust) {
      double rho[3];
      huber2(Problem::vis_chi2(*e), e->delta, e->dsqr, rho);
      wr = rho[1];
    }
    for (int i = 0; i < 9; i++) {
      for (int j = 0; j < 9; j++) {
        double s = 0;
        for (int r = 0; r < e->de; r++) s += J[r * 9 + i] * (wr * e->info) * J[r * 9 + j];
        H[(size_t)i * n + j] += s;
      }
      double s = 0;
      for (int r = 0; r < e->de; r++) s += J[r * 9 + i] * (-(e->info * e->err[r]) * wr);
      b[i] += s;
    }
  }
  if (iteration == 0) {
    double maxDiag = 0;
    for (int j = 0; j < n; j++) maxDiag = std::max(std::fabs(H[(size_t)j * n + j]), maxDiag);
    lm.lambda = 1e-5 * maxDiag;
    lm.ni = 2;
    lm.nBad = 0;
  }
  double rho = 0;
  int qmax = 0;
  do {
    const NS bj = P.nsj, bi = P.nsi;
    std::vector<double> Hl = H, x(n, 0.0);
    for (int j = 0; j < n; j++) Hl[(size_t)j * n + j] += lm.lambda;
    bool ok2 = ldlt_solve(Hl.data(), b.data(), x.data(), n);
    P.apply_update(x.data());
    computeActiveErrors();
    tempChi = activeRobustChi2();
    if (!ok2) tempChi = std::numeric_limits<double>::max();
    rho = (currentChi - tempChi);
    double scale = 0;
    for (int j = 0; j < n; j++) scale += x[j] * (lm.lambda * x[j] + b[j]);
    scale += 1e-3;
    rho /= scale;
    if (rho > 0 && std::isfinite(tempChi)) {
      double alpha = 1. - std::pow((2 * rho - 1), 3);
      alpha = std::min(alpha, 2. / 3.);
      double scaleFactor = std::max(1. / 3., alpha);
      lm.lambda *= scaleFactor;
      lm.ni = 2;
      currentChi = tempChi;
    } else {
      lm.lambda *= lm.ni;
      lm.ni *= 2;
      P.nsj = bj, P.nsi = bi;
    }
    qmax++;
    lm.trials++;
  } while (rho < 0 && qmax < 10);
  if (qmax == 10 || rho == 0) return 1;
  if ((iniChi - currentChi) * 1e3 < iniChi)
    lm.nBad++;
  else
    lm.nBad = 0;
  if (lm.nBad >= 3) return 1;
  return 0;
}

// J^T (w * info) K helper for the marginal prior blocks
static void JtWK(const Dense& J, const Dense& info, double w, const Dense& K, double* out, int ldo,
                 bool add) {
  for (int i = 0; i < J.c; i++)
    for (int j = 0; j < K.c; j++) {
      double s = 0;
      for (int a = 0; a < J.r; a++) {
        double t = 0;
        for (int bq = 0; bq < J.r; bq++) t += (w * info(a, bq)) * K(bq, j);
        s += J(a, i) * t;
      }
      if (add)
        out[i * ldo + j] += s;
      else
        out[i * ldo + j] = s;
    }
}
static double edge_rho1(const GEdge& e) {
  if (!e.robust) return 1.0;
  double rho[3];
  huber2(e.chi2(), e.delta, e.dsqr, rho);
  return rho[1];
}

static void pose_optimization_vio(const vieo_vio_frame& F, const vieo_pose_obs* obs, uint8_t* outlier,
                                  vieo_vio_result& R) {
  memset(&R, 0, sizeof(R));
  R.base.nav = F.base.nav;
  R.base.status = VIEO_POSE_OK;
  Problem P;
  P.F = &F;
  {
    vieo_lba_params prm;
    memset(&prm, 0, sizeof(prm));
    memcpy(prm.Rcb, F.base.Rcb, 72), memcpy(prm.tcb, F.base.tcb, 24);
    prm.fx = F.base.fx, prm.fy = F.base.fy, prm.cx = F.base.cx, prm.cy = F.base.cy, prm.bf = F.base.bf;
    prm.n_cams = F.base.n_cams, prm.cams = F.base.cams;
    ocams_from_params(prm, P.cams);
  }
  P.fixedLast = !F.last_has_prior;  // Optimizer.h:212-213
  P.hasImu = F.imu.dt != 0;
  P.ndim = P.fixedLast ? 15 : 30;
  const NS nsj0 = ns_from(F.base.nav), nsi0 = ns_from(F.nav_last);
  P.nsj = nsj0, P.nsi = nsi0;
  P.prior = ns_from(F.nav_prior);
  bool bodom_edge = false;
  // ---- IMU edge (Optimizer.h:273-299)
  if (P.hasImu) {
    bodom_edge = true;
    GEdge& e = P.eI;
    e.D = 9, e.nv = 3;
    e.vid[0] = 2, e.vid[1] = 0, e.vid[2] = 3;
    e.err.assign(9, 0.0);
    e.info = Dense(9, 9);
    double Inv[81];
    mat_inverse(F.imu.Sigma, Inv, 9);  // GetProcessedInfoij
    for (int i = 0; i < 81; i++) e.info.a[i] = P.fixedLast ? Inv[i] * 1e-2 : Inv[i];
    if (P.fixedLast) {
      e.robust = true;
      e.delta = sqrt(16.919);
      e.dsqr = e.delta * e.delta;
    }
  }
  {  // ---- bias random-walk edge (Optimizer.h:301-323)
    GEdge& e = P.eB;
    e.D = 6, e.nv = 2;
    e.vid[0] = 3, e.vid[1] = 1;
    e.err.assign(6, 0.0);
    e.info = Dense(6, 6);
    const double deltatij = F.imu.dt ? F.imu.dt : F.dt_frames;
    for (int i = 0; i < 3; i++) {
      e.info(i, i) = F.inv_sigma_bg2 / deltatij * (P.fixedLast ? 1e-2 : 1.0);
      e.info(3 + i, 3 + i) = F.inv_sigma_ba2 / deltatij * (P.fixedLast ? 1e-2 : 1.0);
    }
    if (P.fixedLast) {
      e.robust = true;
      e.delta = sqrt(12.592);
      e.dsqr = e.delta * e.delta;
    }
    e.J[0] = Dense(6, 6), e.J[1] = Dense(6, 6);
    for (int i = 0; i < 6; i++) e.J[0](i, i) = -1.0, e.J[1](i, i) = 1.0;
  }
  if (!P.fixedLast) {  // ---- prior edge (Optimizer.h:325-343)
    GEdge& e = P.eP;
    e.D = 15, e.nv = 2;
    e.vid[0] = 2, e.vid[1] = 3;
    e.err.assign(15, 0.0);
    e.info = Dense(15, 15);
    for (int i = 0; i < 225; i++) e.info.a[i] = F.H_prior[i];
    e.robust = true;
    e.delta = sqrt(25);
    e.dsqr = e.delta * e.delta;
  }
  if (F.base.enc && F.base.enc->enc.dt != 0) {  // ---- encoder edge (Optimizer.h:345-363)
    bodom_edge = true;
    P.hasEnc = true;
    const vieo_pose_enc& pe = *F.base.enc;
    GEdge& e = P.eE;
    e.D = 6, e.nv = 2;
    e.vid[0] = 2, e.vid[1] = 0;
    e.err.assign(6, 0.0);
    e.info = Dense(6, 6);
    double Inv[36];
    gj_inverse(pe.enc.Sigma, Inv, 6);
    for (int i = 0; i < 36; i++) e.info.a[i] = Inv[i];
    memcpy(P.measE, pe.enc.delx, 48), memcpy(P.pbe, pe.pbe, 24);
    P.qRbe.w = pe.qRbe[0], P.qRbe.x = pe.qRbe[1], P.qRbe.y = pe.qRbe[2], P.qRbe.z = pe.qRbe[3];
    e.robust = true;
    e.delta = sqrt(12.592);
    e.dsqr = e.delta * e.delta;
  }
  // ---- visual edges
  const int N = F.base.n_obs;
  P.vis.resize(N);
  const float deltaMono = sqrt(5.991), deltaStereo = sqrt(7.815);
  int nInitialCorrespondences = 0;
  const float thClose = 10 < F.th_depth ? F.th_depth : 10;
  (void)thClose;
  for (int i = 0; i < N; i++) {
    const vieo_pose_obs& o = obs[i];
    VisEdge& e = P.vis[i];
    for (int k = 0; k < 3; k++) e.Xw[k] = (double)o.Xw[k];
    e.obs[0] = o.u, e.obs[1] = o.v, e.obs[2] = o.ur;
    e.de = o.ur < 0 ? 2 : 3;
    e.info = (double)o.inv_sigma2;
    e.delta = e.de == 2 ? (double)deltaMono : (double)deltaStereo;
    e.dsqr = e.delta * e.delta;
    e.idx = i;
    e.close = (o.flags & 1) != 0;
    e.cam = (o.flags >> 8) & 15;
    nInitialCorrespondences++;
    outlier[i] = 0;
  }
  int nBad = 0;
  if (nInitialCorrespondences < 3 && !F.no_mps) {  // Optimizer.h:499-503: returns 0
    R.base.n_inliers = 0;
    R.base.status = VIEO_POSE_TOO_FEW;
    return;
  }
  const float chi2Mono[4] = {5.991, 5.991, 5.991, 5.991};
  const float chi2Stereo[4] = {7.815, 7.815, 7.815, 7.815};
  const int n_edges_total = N + (P.hasImu ? 1 : 0) + 1 + (P.fixedLast ? 0 : 1) + (P.hasEnc ? 1 : 0);
  for (size_t it = 0; it < 4; it++) {
    if (!bodom_edge) {  // Optimizer.h:538-545
      P.nsj = nsj0;
      if (!P.fixedLast) P.nsi = nsi0;
    }
    std::vector<VisEdge*> active;
    for (auto& e : P.vis)
      if (e.level == 0) active.push_back(&e);
    LM2 lm;
    for (int i = 0; i < 10; i++) {
      int res = lm_solve_vio(P, active, i, lm);
      R.base.lm_iterations++;
      if (res != 0) break;
    }
    R.base.reserved += lm.trials;
    float chi2close = 1.5 * chi2Mono[it];
    nBad = 0;
    for (int pass = 0; pass < 2; pass++)
      for (auto& e : P.vis) {
        if ((pass == 0) != (e.de == 2)) continue;
        P.vis_error(e);  // exact_mode < kNotExact: always recomputed (Optimizer.h:563,592)
        const float chi2 = Problem::vis_chi2(e);
        bool bad;
        if (e.de == 2)
          bad = chi2 > (e.close ? chi2close : chi2Mono[it]) || !P.vis_depth_positive(e);
        else
          bad = chi2 > chi2Stereo[it];
        if (bad) {
          outlier[e.idx] = 1;
          e.level = 1;
          nBad++;
        } else {
          outlier[e.idx] = 0;
          e.level = 0;
        }
        if (it == 2) e.robust = false;
      }
    if (n_edges_total < 10) break;
  }
  int nInliers = nInitialCorrespondences - nBad;
  if (nInliers < 30) {  // Optimizer.h:621-648
    nBad = 0;
    const float chi2MonoOut = 18.f, chi2StereoOut = 24.f;
    for (int pass = 0; pass < 2; pass++)
      for (auto& e : P.vis) {
        if ((pass == 0) != (e.de == 2)) continue;
        P.vis_error(e);
        if (Problem::vis_chi2(e) < (e.de == 2 ? chi2MonoOut : chi2StereoOut)) {
          e.level = 0;
          outlier[e.idx] = 0;
        } else
          nBad++;
      }
  }
  // ---- recover (Optimizer.h:651-659)
  NS out = P.nsj;  // PVR + (dbg, dba) of the bias vertex; bg/ba unchanged
  ns_to(out, R.base.nav);
  R.base.n_inliers = nInitialCorrespondences - nBad;
  // ---- marginal prior (Optimizer.h:663-813, FillCovInv :126-206), exact_mode = kExactRobust
  if (F.compute_marg) {
    double cov[225] = {0};
    if (P.hasImu) P.imu_error();
    P.bias_error();
    if (P.hasEnc) P.enc_eval(true);  // Optimizer.h:672 computeError; the Jacobians depend on the state only
    double encXj[81], encXi[81], encXji[81];
    if (P.hasEnc) {  // getHessianXj / Xi / Xji of FillCovInv :195-204
      const double w = edge_rho1(P.eE);
      JtWK(P.eE.J[1], P.eE.info, w, P.eE.J[1], encXj, 9, false);
      JtWK(P.eE.J[0], P.eE.info, w, P.eE.J[0], encXi, 9, false);
      JtWK(P.eE.J[1], P.eE.info, w, P.eE.J[0], encXji, 9, false);
    }
    // FillCovInv(schur_bec = 0)
    if (P.hasImu) {
      P.imu_linearize();
      JtWK(P.eI.J[1], P.eI.info, edge_rho1(P.eI), P.eI.J[1], cov, 15, false);
    }
    {
      const double w = edge_rho1(P.eB);
      for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) cov[(9 + i) * 15 + 9 + j] = w * P.eB.info(i, j);
    }
    for (int pass = 0; pass < 2; pass++)
      for (auto& e : P.vis) {
        if ((pass == 0) != (e.de == 2)) continue;
        if (e.level) continue;
        double J[27];
        P.vis_linearize(e, J);
        double wr = 1.0;
        if (e.robust) {
          double rho[3];
          huber2(Problem::vis_chi2(e), e.delta, e.dsqr, rho);
          wr = rho[1];
        }
        for (int i = 0; i < 9; i++)
          for (int j = 0; j < 9; j++) {
            double s = 0;
            for (int r = 0; r < e.de; r++) s += J[r * 9 + i] * (wr * e.info) * J[r * 9 + j];
            cov[i * 15 + j] += s;
          }
      }
    if (P.hasEnc)
      for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++) cov[i * 15 + j] += encXj[i * 9 + j];
    if (!P.fixedLast) {
      double C[225] = {0}, E[225] = {0};
      P.prior_error();
      // schur_bec = 2 : the last state's own block
      if (P.hasImu) {
        const double w = edge_rho1(P.eI);
        double blk[81], blk2[54], blk3[36];
        JtWK(P.eI.J[0], P.eI.info, w, P.eI.J[0], blk, 9, false);
        JtWK(P.eI.J[0], P.eI.info, w, P.eI.J[2], blk2, 6, false);
        JtWK(P.eI.J[2], P.eI.info, w, P.eI.J[2], blk3, 6, false);
        for (int i = 0; i < 9; i++)
          for (int j = 0; j < 9; j++) C[i * 15 + j] = blk[i * 9 + j];
        for (int i = 0; i < 9; i++)
          for (int j = 0; j < 6; j++) C[i * 15 + 9 + j] = blk2[i * 6 + j], C[(9 + j) * 15 + i] = blk2[i * 6 + j];
        for (int i = 0; i < 6; i++)
          for (int j = 0; j < 6; j++) C[(9 + i) * 15 + 9 + j] = blk3[i * 6 + j];
      }
      {
        const double w = edge_rho1(P.eB);  // getHessianXi: (-I)^T W (-I)
        for (int i = 0; i < 6; i++)
          for (int j = 0; j < 6; j++) C[(9 + i) * 15 + 9 + j] += w * P.eB.info(i, j);
      }
      {
        P.prior_linearize();
        const double w = edge_rho1(P.eP);
        double b1[81], b2[36], b3[54];
        JtWK(P.eP.J[0], P.eP.info, w, P.eP.J[0], b1, 9, false);
        JtWK(P.eP.J[1], P.eP.info, w, P.eP.J[1], b2, 6, false);
        JtWK(P.eP.J[0], P.eP.info, w, P.eP.J[1], b3, 6, false);
        for (int i = 0; i < 9; i++)
          for (int j = 0; j < 9; j++) C[i * 15 + j] += b1[i * 9 + j];
        for (int i = 0; i < 6; i++)
          for (int j = 0; j < 6; j++) C[(9 + i) * 15 + 9 + j] += b2[i * 6 + j];
        for (int i = 0; i < 9; i++)
          for (int j = 0; j < 6; j++) C[i * 15 + 9 + j] += b3[i * 6 + j];
        for (int i = 0; i < 9; i++)
          for (int j = 0; j < 6; j++) C[(9 + j) * 15 + i] = C[i * 15 + 9 + j];
      }
      if (P.hasEnc)
        for (int i = 0; i < 9; i++)
          for (int j = 0; j < 9; j++) C[i * 15 + j] += encXi[i * 9 + j];
      // schur_bec = 1 : cross block (cur, last)
      if (P.hasImu) {
        const double w = edge_rho1(P.eI);
        double blk[81], blk2[54];
        JtWK(P.eI.J[1], P.eI.info, w, P.eI.J[0], blk, 9, false);
        JtWK(P.eI.J[1], P.eI.info, w, P.eI.J[2], blk2, 6, false);
        for (int i = 0; i < 9; i++)
          for (int j = 0; j < 9; j++) E[i * 15 + j] = blk[i * 9 + j];
        for (int i = 0; i < 9; i++)
          for (int j = 0; j < 6; j++) E[i * 15 + 9 + j] = blk2[i * 6 + j];
      }
      {
        const double w = edge_rho1(P.eB);  // getHessianXji = Jxj^T W Jxi = -W
        for (int i = 0; i < 6; i++)
          for (int j = 0; j < 6; j++) E[(9 + i) * 15 + 9 + j] = -(w * P.eB.info(i, j));
      }
      if (P.hasEnc)
        for (int i = 0; i < 9; i++)
          for (int j = 0; j < 9; j++) E[i * 15 + j] += encXji[i * 9 + j];
      // margH = B - E C^-1 E^T  (C^-1 through the SVD pseudo-inverse without threshold = inverse)
      double Cinv[225], T[225];
      mat_inverse(C, Cinv, 15);
      for (int i = 0; i < 15; i++)
        for (int j = 0; j < 15; j++) {
          double s = 0;
          for (int k = 0; k < 15; k++) s += E[i * 15 + k] * Cinv[k * 15 + j];
          T[i * 15 + j] = s;
        }
      for (int i = 0; i < 15; i++)
        for (int j = 0; j < 15; j++) {
          double s = 0;
          for (int k = 0; k < 15; k++) s += T[i * 15 + k] * E[j * 15 + k];
          cov[i * 15 + j] -= s;
        }
    }
    memcpy(R.H_marg, cov, sizeof(cov));
    R.has_marg = 1;
  }
}

}  // namespace vo

extern "C" {

void vo_pose_optimization_vio(const vieo_vio_frame* frame, const vieo_pose_obs* obs,
                              uint8_t* outlier, vieo_vio_result* result) {
  vo::pose_optimization_vio(*frame, obs + frame->base.obs_begin, outlier + frame->base.obs_begin,
                            *result);
}

// test helper: IMU edge residual (9) and Jacobians (9x9, 9x9, 9x6 row-major) at given states
void vo_imu_edge_eval(const vieo_vio_frame* F, const vieo_navstate* nsi, const vieo_navstate* nsj,
                      double* err9, double* Ji81, double* Jj81, double* JB54) {
  vo::Problem P;
  P.F = F;
  P.fixedLast = false, P.hasImu = true, P.ndim = 30;
  P.nsi = vo::ns_from(*nsi), P.nsj = vo::ns_from(*nsj);
  P.eI.err.assign(9, 0.0);
  P.imu_error();
  memcpy(err9, P.eI.err.data(), 72);
  if (Ji81) {
    P.imu_linearize();
    memcpy(Ji81, P.eI.J[0].a.data(), 81 * 8);
    memcpy(Jj81, P.eI.J[1].a.data(), 81 * 8);
    memcpy(JB54, P.eI.J[2].a.data(), 54 * 8);
  }
}
// test helper: apply IncSmall(dPVR) / IncSmallBias to a navstate
void vo_navstate_inc(vieo_navstate* ns, const double* dpvr9, const double* dbias6) {
  vo::NS s = vo::ns_from(*ns);
  if (dpvr9) vo::inc_pvr(s, dpvr9);
  if (dbias6) vo::inc_bias(s, dbias6);
  vo::ns_to(s, *ns);
}

}  // extern "C"
