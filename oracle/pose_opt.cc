// ORACLE -- TEST INFRASTRUCTURE ONLY (see ocv_prims.hpp header).  PARITY UNPINNED.
//
// CPU restatement of Optimizer::PoseOptimization(Frame*, Frame*) -- the vision-only motion BA
// (reference: src/Optimizer.cc:1611-1874) -- together with the pieces of the vendored g2o it
// exercises, restated sequentially in the reference's own order of operations:
//   EdgeReproject<DE,6,2>::computeError / linearizeOplus     src/Odom/g2otypes.h:400-541
//   PinholeCamera::Project (+ 2x3 Jacobian), float params    common/camera_models/camera_pinhole.h:70-106
//   NavState::IncSmall (p += R*dp, R *= Exp(dphi))            src/Odom/NavState.h:47-58
//   BaseMultiEdge::constructQuadraticForm, RobustKernelHuber  g2o/core/base_multi_edge.hpp:34-46,161-211,
//                                                             robust_kernel_impl.cpp:78-91
//   BlockSolver::buildSystem / setLambda, LinearSolverDense   g2o/core/block_solver.hpp:501-589,
//                                                             solvers/linear_solver_dense.h:65-113
//   OptimizationAlgorithmLevenberg::solve                     g2o/core/optimization_algorithm_levenberg.cpp:61-207
//   SparseOptimizer::optimize / activeRobustChi2 / push / pop g2o/core/sparse_optimizer.cpp:61-113,354-419
// Eigen (LDLT, small products) and Sophus are third-party and absent here; smallmat.hpp restates
// the closed forms.  Floating-point summation order inside 3x3 products may differ from Eigen's
// (the reference itself is compiled -march=native with FMA contraction), hence parity on poses
// is a 1e-4 tolerance (BASELINE.json), not bit-exactness.
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

struct PoseState {  // the part of NavState a VertexNavStatePR touches
  double p[3];
  Quat q;
};

// NavState::IncSmall(dPR) with USE_P_PLUS_RDP (NavState.h:8,47-58)
static void inc_small_pr(PoseState& s, const double* d) {
  double R[9], Rd[3];
  quat_to_R(s.q, R);
  m3_v(R, d, Rd);
  for (int i = 0; i < 3; i++) s.p[i] += Rd[i];
  Quat e = so3_exp(d + 3);
  s.q = quat_mul(s.q, e);
  quat_normalize(s.q);  // SO3ex::operator*= normalises (so3_extra.h:91-95)
}

struct ReprojEdge {
  double Xw[3];
  double obs[3];
  double info;  // information = I * invSigma2
  int de;       // 2 mono, 3 stereo
  int level = 0;
  bool robust = true;
  double delta, dsqr;  // RobustKernelHuber
  double err[3] = {0, 0, 0};
  int idx;  // keypoint slot
  int cam = 0;  // camera of the observation (bits 8..11 of vieo_pose_obs.flags)
};

// the cameras of a frame (a20): cams[0] is the rectified pinhole camera of vieo_pose_frame when n_cams == 0
struct Cam {
  OCam cams[4];
};

// EdgeReproject::GetTcw_wX + cam_project (g2otypes.h:338-406), NV=2, MODE 0
static void edge_project(const Cam& cc, const PoseState& s, const double* Xw, int de, double* proj,
                         double* Pc_out, double* Rcw_out, int ci = 0) {
  const OCam& c = cc.cams[ci];
  double Rwb[9], Rbw[9], Rcw[9], t[3], tcw[3], Pc[3];
  quat_to_R(s.q, Rwb);
  m3_T(Rwb, Rbw);
  m3_mul(c.Rcb, Rbw, Rcw);
  m3_v(Rcw, s.p, t);
  for (int i = 0; i < 3; i++) tcw[i] = -t[i] + c.tcb[i];
  m3_v(Rcw, Xw, Pc);
  for (int i = 0; i < 3; i++) Pc[i] += tcw[i];
  // camm::Camera::Project: image point as float (Tdata)
  float uv[2];
  ocam_project(c, Pc, uv, nullptr);
  proj[0] = uv[0];
  proj[1] = uv[1];
  if (de > 2) proj[2] = proj[0] - (double)c.bf / Pc[2];
  if (Pc_out) memcpy(Pc_out, Pc, 24);
  if (Rcw_out) memcpy(Rcw_out, Rcw, 72);
}

static void edge_compute_error(const Cam& c, const PoseState& s, ReprojEdge& e) {
  double proj[3];
  edge_project(c, s, e.Xw, e.de, proj, nullptr, nullptr, e.cam);
  for (int i = 0; i < e.de; i++) e.err[i] = e.obs[i] - proj[i];
}

static double edge_chi2(const ReprojEdge& e) {  // _error.dot(information()*_error)
  double s = 0;
  for (int i = 0; i < e.de; i++) s += e.err[i] * (e.info * e.err[i]);
  return s;
}

static void huber(double e, double delta, double dsqr, double* rho) {  // robust_kernel_impl.cpp:78-91
  if (e <= dsqr) {
    rho[0] = e, rho[1] = 1., rho[2] = 0.;
  } else {
    double sqrte = std::sqrt(e);
    rho[0] = 2 * sqrte * delta - dsqr;
    rho[1] = delta / sqrte;
    rho[2] = -0.5 * rho[1] / e;
  }
}

// EdgeReproject::linearizeOplus (g2otypes.h:439-541), Jacobian w.r.t. the PR vertex (de x 6)
static void edge_linearize(const Cam& cc, const PoseState& s, const ReprojEdge& e, double* J) {
  const OCam& c = cc.cams[e.cam];
  double proj[3], Pc[3], Rcw[9];
  edge_project(cc, s, e.Xw, e.de, proj, Pc, Rcw, e.cam);
  const double invz = 1 / Pc[2], invz_2 = invz * invz;
  double Jproj[9] = {0};
  {  // Project Jacobian of the edge's camera, then Jproj = -J
    double Jt[6];
    ocam_project(c, Pc, nullptr, Jt);
    for (int i = 0; i < 6; i++) Jproj[i] = -Jt[i];
  }
  if (e.de > 2) {
    Jproj[6] = Jproj[0];
    Jproj[7] = Jproj[1];
    Jproj[8] = Jproj[2] - (double)c.bf * invz_2;
  }
  // JdPwb = Jproj * (-Rcb)
  double Rwb[9], dP[3], Paux[3], H[9], RcbH[9];
  quat_to_R(s.q, Rwb);
  for (int i = 0; i < 3; i++) dP[i] = e.Xw[i] - s.p[i];
  m3T_v(Rwb, dP, Paux);  // Rwb^T (Pw - pwb)
  hat(Paux, H);
  m3_mul(c.Rcb, H, RcbH);
  for (int r = 0; r < e.de; r++) {
    for (int k = 0; k < 3; k++) {
      double a = 0, b = 0;
      for (int m = 0; m < 3; m++) {
        a += Jproj[r * 3 + m] * (-c.Rcb[m * 3 + k]);
        b += Jproj[r * 3 + m] * RcbH[m * 3 + k];
      }
      J[r * 6 + k] = a;      // d/dp
      J[r * 6 + 3 + k] = b;  // d/dphi
    }
  }
}

// the optional EdgeEncNavStatePR between the fixed last frame (vertex i) and the frame (vertex j)
// (Optimizer.cc:1650-1674): Huber sqrt(12.592) in all four rounds
struct EncEdge {
  bool on = false;
  EncPose last;
  Quat qRbe;
  double pbe[3], meas[6], Info[36], err[6], Jj[36];
  double chi2() const {
    double s = 0;
    for (int a = 0; a < 6; a++) {
      double t = 0;
      for (int b = 0; b < 6; b++) t += Info[a * 6 + b] * err[b];
      s += err[a] * t;
    }
    return s;
  }
  void eval(const PoseState& s, bool jac) {
    EncPose cur;
    memcpy(cur.p, s.p, 24);
    cur.q = s.q;
    double Ji[36];
    enc_edge_eval(last, cur, meas, qRbe, pbe, err, jac ? Ji : nullptr, jac ? Jj : nullptr);
  }
};
static const float kThEnc = std::sqrt(12.592);

struct LMState {
  double lambda = -1, ni = 2;
  int nBad = 0;
};

// one OptimizationAlgorithmLevenberg::solve(iteration); returns 0 OK, 1 Terminate
static int lm_solve(const Cam& c, PoseState& est, std::vector<ReprojEdge*>& active, int iteration,
                    LMState& lm, EncEdge& enc) {
  auto computeActiveErrors = [&]() {
    for (auto* e : active) edge_compute_error(c, est, *e);
    if (enc.on) enc.eval(est, false);
  };
  auto activeRobustChi2 = [&]() {
    double chi = 0, rho[3];
    for (auto* e : active) {
      if (e->robust) {
        huber(edge_chi2(*e), e->delta, e->dsqr, rho);
        chi += rho[0];
      } else
        chi += edge_chi2(*e);
    }
    if (enc.on) {
      huber(enc.chi2(), (double)kThEnc, (double)kThEnc * (double)kThEnc, rho);
      chi += rho[0];
    }
    return chi;
  };
  computeActiveErrors();
  double currentChi = activeRobustChi2();
  double tempChi = currentChi;
  const double iniChi = currentChi;
  // buildSystem
  double H[36] = {0}, b[6] = {0};
  for (auto* e : active) {
    double J[18];
    edge_linearize(c, est, *e, J);
    double w = e->info, wr = 1.0;
    if (e->robust) {
      double rho[3];
      huber(edge_chi2(*e), e->delta, e->dsqr, rho);
      wr = rho[1];
    }
    // omega_r = -info*err*rho1 ; H += J^T (rho1*info) J ; b += J^T omega_r
    for (int i = 0; i < 6; i++) {
      for (int j = 0; j < 6; j++) {
        double s = 0;
        for (int r = 0; r < e->de; r++) s += J[r * 6 + i] * (wr * w) * J[r * 6 + j];
        H[i * 6 + j] += s;
      }
      double s = 0;
      for (int r = 0; r < e->de; r++) s += J[r * 6 + i] * (-(w * e->err[r]) * wr);
      b[i] += s;
    }
  }
  if (enc.on) {
    enc.eval(est, true);
    double rho[3];
    huber(enc.chi2(), (double)kThEnc, (double)kThEnc * (double)kThEnc, rho);
    double T[36], we[6];
    for (int a = 0; a < 6; a++) {
      double t = 0;
      for (int q = 0; q < 6; q++) t += enc.Info[a * 6 + q] * enc.err[q];
      we[a] = -t * rho[1];
      for (int c2 = 0; c2 < 6; c2++) {
        double u = 0;
        for (int q = 0; q < 6; q++) u += (rho[1] * enc.Info[a * 6 + q]) * enc.Jj[q * 6 + c2];
        T[a * 6 + c2] = u;
      }
    }
    for (int i = 0; i < 6; i++) {
      for (int j = 0; j < 6; j++) {
        double s2 = 0;
        for (int a = 0; a < 6; a++) s2 += enc.Jj[a * 6 + i] * T[a * 6 + j];
        H[i * 6 + j] += s2;
      }
      double s2 = 0;
      for (int a = 0; a < 6; a++) s2 += enc.Jj[a * 6 + i] * we[a];
      b[i] += s2;
    }
  }
  if (iteration == 0) {  // computeLambdaInit: tau * max diagonal
    double maxDiag = 0;
    for (int j = 0; j < 6; j++) maxDiag = std::max(std::fabs(H[j * 6 + j]), maxDiag);
    lm.lambda = 1e-5 * maxDiag;
    lm.ni = 2;
    lm.nBad = 0;
  }
  double rho = 0;
  int qmax = 0;
  const int maxTrials = 10;
  do {
    PoseState backup = est;  // push
    double Hl[36];
    memcpy(Hl, H, sizeof(H));
    for (int j = 0; j < 6; j++) Hl[j * 6 + j] += lm.lambda;
    double x[6] = {0};
    bool ok2 = ldlt_solve(Hl, b, x, 6);
    inc_small_pr(est, x);  // update
    computeActiveErrors();
    tempChi = activeRobustChi2();
    if (!ok2) tempChi = std::numeric_limits<double>::max();
    rho = (currentChi - tempChi);
    double scale = 0;
    for (int j = 0; j < 6; j++) scale += x[j] * (lm.lambda * x[j] + b[j]);
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
      est = backup;  // pop (edge errors are NOT restored, as in g2o)
    }
    qmax++;
  } while (rho < 0 && qmax < maxTrials);
  if (qmax == maxTrials || rho == 0) return 1;
  if ((iniChi - currentChi) * 1e3 < iniChi)
    lm.nBad++;
  else
    lm.nBad = 0;
  if (lm.nBad >= 3) return 1;
  return 0;
}

// Optimizer::PoseOptimization(Frame*, Frame*) (Optimizer.cc:1611-1874)
static void pose_optimization(const vieo_pose_frame& F, const vieo_pose_obs* obs, uint8_t* outlier,
                              vieo_pose_result& R) {
  R.nav = F.nav;
  R.status = VIEO_POSE_OK;
  R.lm_iterations = 0;
  Cam c;
  {
    vieo_lba_params prm;
    memset(&prm, 0, sizeof(prm));
    memcpy(prm.Rcb, F.Rcb, 72), memcpy(prm.tcb, F.tcb, 24);
    prm.fx = F.fx, prm.fy = F.fy, prm.cx = F.cx, prm.cy = F.cy, prm.bf = F.bf;
    prm.n_cams = F.n_cams, prm.cams = F.cams;
    ocams_from_params(prm, c.cams);
  }
  const int N = F.n_obs;
  std::vector<ReprojEdge> edges(N);
  const float deltaMono = sqrt(5.991), deltaStereo = sqrt(7.815);
  int nInitialCorrespondences = 0;
  for (int i = 0; i < N; i++) {
    const vieo_pose_obs& o = obs[i];
    ReprojEdge& e = edges[i];
    for (int k = 0; k < 3; k++) e.Xw[k] = (double)o.Xw[k];
    e.obs[0] = o.u, e.obs[1] = o.v, e.obs[2] = o.ur;
    e.de = o.ur < 0 ? 2 : 3;
    e.info = (double)o.inv_sigma2;
    e.delta = e.de == 2 ? (double)deltaMono : (double)deltaStereo;
    e.dsqr = e.delta * e.delta;
    e.idx = i;
    e.cam = (o.flags >> 8) & 15;
    ++nInitialCorrespondences;
    outlier[i] = 0;
  }
  if (nInitialCorrespondences < 3) {
    R.n_inliers = 0;
    R.status = VIEO_POSE_TOO_FEW;
    return;
  }
  const float chi2Mono[4] = {5.991, 5.991, 5.991, 5.991};
  const float chi2Stereo[4] = {7.815, 7.815, 7.815, 7.815};
  const int its[4] = {10, 10, 10, 10};
  PoseState init;
  memcpy(init.p, F.nav.p, 24);
  init.q.w = F.nav.q[0], init.q.x = F.nav.q[1], init.q.y = F.nav.q[2], init.q.z = F.nav.q[3];
  PoseState est = init;
  EncEdge enc;
  if (F.enc && F.enc->enc.dt != 0) {  // Optimizer.cc:1650-1674
    const vieo_pose_enc& E = *F.enc;
    enc.on = true;
    memcpy(enc.last.p, E.p_last, 24);
    enc.last.q.w = E.q_last[0], enc.last.q.x = E.q_last[1], enc.last.q.y = E.q_last[2], enc.last.q.z = E.q_last[3];
    enc.qRbe.w = E.qRbe[0], enc.qRbe.x = E.qRbe[1], enc.qRbe.y = E.qRbe[2], enc.qRbe.z = E.qRbe[3];
    memcpy(enc.pbe, E.pbe, 24), memcpy(enc.meas, E.enc.delx, 48);
    gj_inverse(E.enc.Sigma, enc.Info, 6);
    memset(enc.err, 0, sizeof(enc.err));
  }
  int nBad = 0;
  for (size_t it = 0; it < 4; it++) {
    est = init;  // vns->setEstimate(pFrame->GetNavStateRef())
    std::vector<ReprojEdge*> active;
    for (auto& e : edges)
      if (e.level == 0) active.push_back(&e);
    // optimizer.optimize(its[it])
    if (!active.empty() || enc.on) {
      LMState lm;
      for (int i = 0; i < its[it]; i++) {
        int res = lm_solve(c, est, active, i, lm, enc);
        R.lm_iterations++;
        if (res != 0) break;
      }
    }
    nBad = 0;
    // the reference visits all mono edges first, then all stereo edges; decisions are per edge
    for (int pass = 0; pass < 2; pass++)
      for (auto& e : edges) {
        if ((pass == 0) != (e.de == 2)) continue;
        if (outlier[e.idx]) edge_compute_error(c, est, e);
        const float chi2 = edge_chi2(e);
        const float th = e.de == 2 ? chi2Mono[it] : chi2Stereo[it];
        if (chi2 > th) {
          outlier[e.idx] = 1;
          e.level = 1;
          nBad++;
        } else {
          outlier[e.idx] = 0;
          e.level = 0;
        }
        if (it == 2) e.robust = false;
      }
    if (edges.size() + (enc.on ? 1 : 0) < 10) break;
  }
  memcpy(R.nav.p, est.p, 24);
  R.nav.q[0] = est.q.w, R.nav.q[1] = est.q.x, R.nav.q[2] = est.q.y, R.nav.q[3] = est.q.z;
  R.n_inliers = nInitialCorrespondences - nBad;
}

}  // namespace vo

extern "C" {

void vo_pose_optimization(const vieo_pose_frame* frame, const vieo_pose_obs* obs, uint8_t* outlier,
                          vieo_pose_result* result) {
  vo::pose_optimization(*frame, obs + frame->obs_begin, outlier + frame->obs_begin, *result);
}

// test helpers: residual + analytic Jacobian of one edge, for finite-difference checks
void vo_pose_edge_eval(const vieo_pose_frame* F, const vieo_pose_obs* o, const double* delta6,
                       double* err3, double* J18) {
  vo::Cam cc;
  {
    vieo_lba_params prm;
    memset(&prm, 0, sizeof(prm));
    memcpy(prm.Rcb, F->Rcb, 72), memcpy(prm.tcb, F->tcb, 24);
    prm.fx = F->fx, prm.fy = F->fy, prm.cx = F->cx, prm.cy = F->cy, prm.bf = F->bf;
    vo::ocams_from_params(prm, cc.cams);
  }
  const vo::OCam& c = cc.cams[0];
  vo::PoseState s;
  memcpy(s.p, F->nav.p, 24);
  s.q.w = F->nav.q[0], s.q.x = F->nav.q[1], s.q.y = F->nav.q[2], s.q.z = F->nav.q[3];
  if (delta6) vo::inc_small_pr(s, delta6);
  vo::ReprojEdge e;
  for (int k = 0; k < 3; k++) e.Xw[k] = o->Xw[k];
  e.obs[0] = o->u, e.obs[1] = o->v, e.obs[2] = o->ur;
  e.de = o->ur < 0 ? 2 : 3;
  e.info = o->inv_sigma2;
  // residual WITHOUT the float rounding of the projection, so central differences are smooth
  double Rwb[9], Rbw[9], Rcw[9], t[3], Pc[3];
  vo::quat_to_R(s.q, Rwb);
  vo::m3_T(Rwb, Rbw);
  vo::m3_mul(c.Rcb, Rbw, Rcw);
  vo::m3_v(Rcw, s.p, t);
  vo::m3_v(Rcw, e.Xw, Pc);
  for (int i = 0; i < 3; i++) Pc[i] += -t[i] + c.tcb[i];
  err3[0] = e.obs[0] - ((double)c.fx * Pc[0] / Pc[2] + c.cx);
  err3[1] = e.obs[1] - ((double)c.fy * Pc[1] / Pc[2] + c.cy);
  err3[2] = e.de > 2 ? e.obs[2] - (((double)c.fx * Pc[0] / Pc[2] + c.cx) - c.bf / Pc[2]) : 0;
  if (J18) vo::edge_linearize(cc, s, e, J18);
}

void vo_so3_exp(const double* w, double* q4) {
  vo::Quat q = vo::so3_exp(w);
  q4[0] = q.w, q4[1] = q.x, q4[2] = q.y, q4[3] = q.z;
}
void vo_so3_log(const double* q4, double* w) {
  vo::Quat q;
  q.w = q4[0], q.x = q4[1], q.y = q4[2], q.z = q4[3];
  vo::so3_log(q, w);
}
void vo_so3_jr(const double* w, double* J, int inverse) {
  if (inverse)
    vo::so3_JrInv(w, J);
  else
    vo::so3_Jr(w, J);
}

}  // extern "C"
