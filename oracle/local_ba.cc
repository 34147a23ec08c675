// ORACLE -- TEST INFRASTRUCTURE ONLY (see ocv_prims.hpp header).  PARITY UNPINNED.
//
// CPU restatement of Optimizer::LocalBundleAdjustment (vision-only; reference: src/Optimizer.cc:1876-2307)
// on a flattened window, with the vendored g2o pieces it runs through:
//   EdgeEncNavStatePR between consecutive key frames       src/Odom/g2otypes.h:591-668 (enc_edge.hpp),
//     Optimizer.cc:2008-2042 (local BA), :1401-1438 (BundleAdjustment, bEnc)
//   EdgeReprojectPR / PRStereo incl. the point Jacobian   src/Odom/g2otypes.h:400-541
//   BlockSolver<6,3>::buildSystem / setLambda / solve     g2o/core/block_solver.hpp:501-589,353-486
//     (Hpp, Hll, Hpl blocks; Schur complement  Hschur = Hpp - sum_l B D^-1 B^T,
//      bschur = bp - sum_l B D^-1 bl;  xl = D^-1 (bl - B^T xp))
//   OptimizationAlgorithmLevenberg::solve                  g2o/core/optimization_algorithm_levenberg.cpp:61-207
//   LinearSolverEigen (SimplicialLDLT)                     g2o/solvers/linear_solver_eigen.h:94-124
//     -> restated as a dense LDL^T of the reduced system (same solution up to rounding).
// g2o keeps edge errors of rejected LM trials (pop() restores vertices only) and the reference
// classifies with e->chi2() without recomputing (Optimizer.cc:2191-2212,2227-2249): kept.
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

struct KFState {
  double p[3];
  Quat q;
  bool fixed;
  int col;  // offset in the pose system, -1 fixed / inactive
};

struct LEdge {
  int kf, mp, de, level = 0, cam = 0;
  double obs[3], info, delta, dsqr;
  bool robust = true;
  double err[3] = {0, 0, 0};
};

struct PairEdge {  // EdgeEncNavStatePR: vertex 0 = kf i (previous), vertex 1 = kf j
  int i, j;
  double meas[6], Info[36], err[6] = {0, 0, 0, 0, 0, 0};
  bool robust = true;
  double chi2() const {
    double s = 0;
    for (int a = 0; a < 6; a++) {
      double t = 0;
      for (int b = 0; b < 6; b++) t += Info[a * 6 + b] * err[b];
      s += err[a] * t;
    }
    return s;
  }
};
static const double kDeltaEnc = std::sqrt(12.592);

struct LBA {
  const vieo_lba_params* P;
  std::vector<PairEdge> G;
  Quat qRbe;
  double pbe[3];
  void enc_eval(PairEdge& g, double* Ji, double* Jj) const {
    EncPose si, sj;
    memcpy(si.p, kf[g.i].p, 24), memcpy(sj.p, kf[g.j].p, 24);
    si.q = kf[g.i].q, sj.q = kf[g.j].q;
    enc_edge_eval(si, sj, g.meas, qRbe, pbe, g.err, Ji, Jj);
  }
  OCam cams[4];
  std::vector<KFState> kf;
  std::vector<double> X;  // points, 3 per mp
  std::vector<LEdge> E;
  std::vector<int> mp_first, mp_count;  // edge range per point (obs sorted by mp)

  void project(const LEdge& e, double* proj, double* Pc_out, double* Rcw_out) const {
    const KFState& s = kf[e.kf];
    const OCam& C = cams[e.cam];
    double Rwb[9], Rbw[9], Rcw[9], t[3], Pc[3];
    quat_to_R(s.q, Rwb);
    m3_T(Rwb, Rbw);
    m3_mul(C.Rcb, Rbw, Rcw);
    m3_v(Rcw, s.p, t);
    m3_v(Rcw, &X[3 * e.mp], Pc);
    for (int i = 0; i < 3; i++) Pc[i] += -t[i] + C.tcb[i];
    float uv[2];
    ocam_project(C, Pc, uv, nullptr);
    proj[0] = uv[0], proj[1] = uv[1];
    if (e.de > 2) proj[2] = proj[0] - (double)C.bf / Pc[2];
    if (Pc_out) memcpy(Pc_out, Pc, 24);
    if (Rcw_out) memcpy(Rcw_out, Rcw, 72);
  }
  void compute_error(LEdge& e) const {
    double proj[3];
    project(e, proj, nullptr, nullptr);
    for (int i = 0; i < e.de; i++) e.err[i] = e.obs[i] - proj[i];
  }
  static double chi2(const LEdge& e) {
    double s = 0;
    for (int i = 0; i < e.de; i++) s += e.err[i] * (e.info * e.err[i]);
    return s;
  }
  bool depth_positive(const LEdge& e) const {
    double proj[3], Pc[3];
    project(e, proj, Pc, nullptr);
    return Pc[2] > 0.;
  }
  // Jp: de x 6 (dp, dphi of the key frame), Jx: de x 3 (point)
  void linearize(const LEdge& e, double* Jp, double* Jx) const {
    double proj[3], Pc[3], Rcw[9];
    project(e, proj, Pc, Rcw);
    const KFState& s = kf[e.kf];
    const OCam& C = cams[e.cam];
    const double invz = 1 / Pc[2], invz_2 = invz * invz;
    double J[9] = {0}, Jc[6];
    ocam_project(C, Pc, nullptr, Jc);  // Jproj = -Jproj_tmp (g2otypes.h:453-459)
    for (int i = 0; i < 6; i++) J[i] = -Jc[i];
    if (e.de > 2) J[6] = J[0], J[7] = J[1], J[8] = J[2] - (double)C.bf * invz_2;
    double Rwb[9], dP[3], Paux[3], H[9], RcbH[9];
    quat_to_R(s.q, Rwb);
    for (int i = 0; i < 3; i++) dP[i] = X[3 * e.mp + i] - s.p[i];
    m3T_v(Rwb, dP, Paux);
    hat(Paux, H);
    m3_mul(C.Rcb, H, RcbH);
    for (int r = 0; r < e.de; r++)
      for (int k = 0; k < 3; k++) {
        double a = 0, b = 0, c = 0;
        for (int m = 0; m < 3; m++) {
          a += J[r * 3 + m] * (-C.Rcb[m * 3 + k]);
          b += J[r * 3 + m] * RcbH[m * 3 + k];
          c += J[r * 3 + m] * Rcw[m * 3 + k];  // _jacobianOplus[0] = Jproj * Rcw
        }
        Jp[r * 6 + k] = a;
        Jp[r * 6 + 3 + k] = b;
        Jx[r * 3 + k] = c;
      }
  }
};

static void hub(double e, double delta, double dsqr, double* rho) {
  if (e <= dsqr) {
    rho[0] = e, rho[1] = 1.;
  } else {
    double s = std::sqrt(e);
    rho[0] = 2 * s * delta - dsqr;
    rho[1] = delta / s;
  }
}

static void inc_pose(KFState& s, const double* d) {
  double R[9], Rd[3];
  quat_to_R(s.q, R);
  m3_v(R, d, Rd);
  for (int i = 0; i < 3; i++) s.p[i] += Rd[i];
  s.q = quat_mul(s.q, so3_exp(d + 3));
  quat_normalize(s.q);
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

struct LMs {
  double lambda = -1, ni = 2;
  int nBad = 0;
};

// one optimize(iterations) over the active (level 0) edges; returns iterations executed
static int optimize(LBA& B, int iterations, volatile const int* stop, vieo_lba_result& R, bool first) {
  // active sets
  const int nk = (int)B.kf.size(), nm = (int)B.mp_first.size();
  std::vector<int> act;
  for (size_t i = 0; i < B.E.size(); i++)
    if (B.E[i].level == 0) act.push_back((int)i);
  std::vector<char> kf_act(nk, 0), mp_act(nm, 0);
  for (int i : act) kf_act[B.E[i].kf] = 1, mp_act[B.E[i].mp] = 1;
  for (const PairEdge& g : B.G) kf_act[g.i] = kf_act[g.j] = 1;  // level-0 edges: their vertices are active
  int np = 0;
  for (int k = 0; k < nk; k++) {
    if (!B.kf[k].fixed && kf_act[k])
      B.kf[k].col = np, np += 6;
    else
      B.kf[k].col = -1;
  }
  if (np == 0 || act.empty()) return 0;
  auto computeActiveErrors = [&]() {
    for (int i : act) B.compute_error(B.E[i]);
    for (PairEdge& g : B.G) B.enc_eval(g, nullptr, nullptr);
  };
  auto activeRobustChi2 = [&]() {
    double chi = 0, rho[2];
    for (int i : act) {
      const LEdge& e = B.E[i];
      if (e.robust) {
        hub(LBA::chi2(e), e.delta, e.dsqr, rho);
        chi += rho[0];
      } else
        chi += LBA::chi2(e);
    }
    for (const PairEdge& g : B.G) {
      if (g.robust) {
        hub(g.chi2(), kDeltaEnc, kDeltaEnc * kDeltaEnc, rho);
        chi += rho[0];
      } else
        chi += g.chi2();
    }
    return chi;
  };
  LMs lm;
  int done = 0;
  for (int it = 0; it < iterations; it++) {
    if (stop && *stop) break;
    done++;
    R.lm_iterations++;
    computeActiveErrors();
    double currentChi = activeRobustChi2();
    if (first && it == 0) R.chi2_initial = currentChi;
    double tempChi = currentChi;
    const double iniChi = currentChi;
    // ---- buildSystem
    std::vector<double> Hpp((size_t)np * np, 0.0), bp(np, 0.0), Hll((size_t)nm * 9, 0.0), bl((size_t)nm * 3, 0.0);
    std::vector<double> Bpl(B.E.size() * 18, 0.0);  // per edge: 6x3 block (free pose only)
    for (int i : act) {
      const LEdge& e = B.E[i];
      double Jp[18], Jx[9];
      B.linearize(e, Jp, Jx);
      double wr = 1.0;
      if (e.robust) {
        double rho[2];
        hub(LBA::chi2(e), e.delta, e.dsqr, rho);
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
      if (c >= 0) {
        for (int a = 0; a < 6; a++) {
          for (int b2 = 0; b2 < 6; b2++) {
            double s = 0;
            for (int r = 0; r < e.de; r++) s += Jp[r * 6 + a] * w * Jp[r * 6 + b2];
            Hpp[(size_t)(c + a) * np + c + b2] += s;
          }
          double s = 0;
          for (int r = 0; r < e.de; r++) s += Jp[r * 6 + a] * (-(e.info * e.err[r]) * wr);
          bp[c + a] += s;
          for (int b2 = 0; b2 < 3; b2++) {
            double t = 0;
            for (int r = 0; r < e.de; r++) t += Jp[r * 6 + a] * w * Jx[r * 3 + b2];
            Bpl[(size_t)i * 18 + a * 3 + b2] = t;
          }
        }
      }
    }
    for (PairEdge& g : B.G) {  // J^T (rho' Omega) J of the pair, both blocks and the cross block
      double J[2][36];
      B.enc_eval(g, J[0], J[1]);
      double wr = 1.0;
      if (g.robust) {
        double rho[2];
        hub(g.chi2(), kDeltaEnc, kDeltaEnc * kDeltaEnc, rho);
        wr = rho[1];
      }
      double we[6];
      for (int a = 0; a < 6; a++) {
        we[a] = 0;
        for (int q = 0; q < 6; q++) we[a] += g.Info[a * 6 + q] * g.err[q];
      }
      const int col[2] = {B.kf[g.i].col, B.kf[g.j].col};
      for (int u = 0; u < 2; u++) {
        if (col[u] < 0) continue;
        for (int a = 0; a < 6; a++) {
          double sb = 0;
          for (int r = 0; r < 6; r++) sb += J[u][r * 6 + a] * (-we[r] * wr);
          bp[col[u] + a] += sb;
        }
        for (int v = 0; v < 2; v++) {
          if (col[v] < 0) continue;
          for (int a = 0; a < 6; a++)
            for (int b2 = 0; b2 < 6; b2++) {
              double sH = 0;
              for (int r = 0; r < 6; r++) {
                double t = 0;
                for (int q = 0; q < 6; q++) t += (wr * g.Info[r * 6 + q]) * J[v][q * 6 + b2];
                sH += J[u][r * 6 + a] * t;
              }
              Hpp[(size_t)(col[u] + a) * np + col[v] + b2] += sH;
            }
        }
      }
    }
    if (it == 0) {
      double maxDiag = 0;
      for (int j = 0; j < np; j++) maxDiag = std::max(std::fabs(Hpp[(size_t)j * np + j]), maxDiag);
      for (int m = 0; m < nm; m++)
        if (mp_act[m])
          for (int a = 0; a < 3; a++) maxDiag = std::max(std::fabs(Hll[(size_t)m * 9 + a * 4]), maxDiag);
      lm.lambda = 1e-5 * maxDiag;
      lm.ni = 2;
      lm.nBad = 0;
    }
    double rho = 0;
    int qmax = 0;
    do {
      R.lm_trials++;
      std::vector<KFState> bk = B.kf;
      std::vector<double> bX = B.X;
      // ---- Schur complement with lambda on both diagonals
      std::vector<double> Hs = Hpp, bs = bp, Dinv((size_t)nm * 9, 0.0), xp(np, 0.0);
      for (int j = 0; j < np; j++) Hs[(size_t)j * np + j] += lm.lambda;
      for (int m = 0; m < nm; m++) {
        if (!mp_act[m]) continue;
        double D[9];
        memcpy(D, &Hll[(size_t)m * 9], 72);
        D[0] += lm.lambda, D[4] += lm.lambda, D[8] += lm.lambda;
        inv3(D, &Dinv[(size_t)m * 9]);
        const double* Di = &Dinv[(size_t)m * 9];
        double db[3];
        m3_v(Di, &bl[(size_t)m * 3], db);
        for (int i1 = B.mp_first[m]; i1 < B.mp_first[m] + B.mp_count[m]; i1++) {
          const LEdge& e1 = B.E[i1];
          const int c1 = B.kf[e1.kf].col;
          if (e1.level != 0 || c1 < 0) continue;
          const double* B1 = &Bpl[(size_t)i1 * 18];
          double BD[18];
          for (int a = 0; a < 6; a++)
            for (int b2 = 0; b2 < 3; b2++)
              BD[a * 3 + b2] = B1[a * 3] * Di[b2] + B1[a * 3 + 1] * Di[3 + b2] + B1[a * 3 + 2] * Di[6 + b2];
          for (int a = 0; a < 6; a++)
            bs[c1 + a] -= B1[a * 3] * db[0] + B1[a * 3 + 1] * db[1] + B1[a * 3 + 2] * db[2];
          for (int i2 = B.mp_first[m]; i2 < B.mp_first[m] + B.mp_count[m]; i2++) {
            const LEdge& e2 = B.E[i2];
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
            const LEdge& e1 = B.E[i1];
            const int c1 = B.kf[e1.kf].col;
            if (e1.level != 0 || c1 < 0) continue;
            const double* B1 = &Bpl[(size_t)i1 * 18];
            for (int b2 = 0; b2 < 3; b2++)
              for (int a = 0; a < 6; a++) cl[b2] -= B1[a * 3 + b2] * xp[c1 + a];
          }
          m3_v(&Dinv[(size_t)m * 9], cl, &xl[(size_t)m * 3]);
        }
      }
      // update (poses by IncSmall, points additively)
      for (int k = 0; k < nk; k++)
        if (B.kf[k].col >= 0) inc_pose(B.kf[k], &xp[B.kf[k].col]);
      for (int m = 0; m < nm; m++)
        if (mp_act[m])
          for (int a = 0; a < 3; a++) B.X[(size_t)m * 3 + a] += xl[(size_t)m * 3 + a];
      computeActiveErrors();
      tempChi = activeRobustChi2();
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      rho = currentChi - tempChi;
      double scale = 0;
      for (int j = 0; j < np; j++) scale += xp[j] * (lm.lambda * xp[j] + bp[j]);
      for (int m = 0; m < nm; m++)
        if (mp_act[m])
          for (int a = 0; a < 3; a++)
            scale += xl[(size_t)m * 3 + a] * (lm.lambda * xl[(size_t)m * 3 + a] + bl[(size_t)m * 3 + a]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        lm.lambda *= std::max(1. / 3., alpha);
        lm.ni = 2;
        currentChi = tempChi;
      } else {
        lm.lambda *= lm.ni;
        lm.ni *= 2;
        B.kf = bk;
        B.X = bX;
      }
      qmax++;
    } while (rho < 0 && qmax < 10 && !(stop && *stop));
    R.chi2_final = currentChi;
    if (qmax == 10 || rho == 0) break;
    if ((iniChi - currentChi) * 1e3 < iniChi)
      lm.nBad++;
    else
      lm.nBad = 0;
    if (lm.nBad >= 3) break;
  }
  return done;
}

// gba_iterations >= 0: Optimizer::BundleAdjustment (Optimizer.cc:1353-1609) instead -- one optimize(), Huber
// (sqrt(5.99) / sqrt(7.815), :1445-1446) iff gba_robust, no classification, every point written back
static void local_ba(const vieo_lba_params& P, const vieo_lba_keyframe* kfs, int n_kf,
                     const float* points, int n_mp, const vieo_lba_obs* obs, int n_obs,
                     volatile const int* stop, vieo_navstate* navs_out, float* points_out,
                     uint8_t* erase, vieo_lba_result& R, int gba_iterations = -1, bool gba_robust = false,
                     const vieo_lba_enc* enc = nullptr) {
  const bool gba = gba_iterations >= 0;
  memset(&R, 0, sizeof(R));
  for (int k = 0; k < n_kf; k++) navs_out[k] = kfs[k].nav;
  memcpy(points_out, points, (size_t)n_mp * 12);
  memset(erase, 0, n_obs);
  LBA B;
  B.P = &P;
  ocams_from_params(P, B.cams);
  B.kf.resize(n_kf);
  bool any_free = false;
  for (int k = 0; k < n_kf; k++) {
    memcpy(B.kf[k].p, kfs[k].nav.p, 24);
    B.kf[k].q.w = kfs[k].nav.q[0], B.kf[k].q.x = kfs[k].nav.q[1];
    B.kf[k].q.y = kfs[k].nav.q[2], B.kf[k].q.z = kfs[k].nav.q[3];
    B.kf[k].fixed = kfs[k].fixed != 0;
    any_free |= !B.kf[k].fixed;
  }
  if (!any_free) {
    R.status = VIEO_LBA_NO_FREE_POSE;
    return;
  }
  B.X.resize((size_t)n_mp * 3);
  for (int i = 0; i < n_mp * 3; i++) B.X[i] = (double)points[i];
  B.E.resize(n_obs);
  B.mp_first.assign(n_mp, 0);
  B.mp_count.assign(n_mp, 0);
  const float thHuberMono = sqrt(gba ? 5.99 : 5.991), thHuberStereo = sqrt(7.815);
  for (int i = 0; i < n_obs; i++) {
    LEdge& e = B.E[i];
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
  if (enc) {
    B.qRbe.w = enc->qRbe[0], B.qRbe.x = enc->qRbe[1], B.qRbe.y = enc->qRbe[2], B.qRbe.z = enc->qRbe[3];
    memcpy(B.pbe, enc->pbe, 24);
    for (int t = 0; t < enc->n_edges; t++) {
      const vieo_lba_enc_edge& e = enc->edges[t];
      if (e.enc.dt == 0) continue;
      PairEdge g;
      g.i = e.kf_i, g.j = e.kf_j;
      memcpy(g.meas, e.enc.delx, 48);
      gj_inverse(e.enc.Sigma, g.Info, 6);
      if (B.kf[g.i].fixed)
        for (int q = 0; q < 36; q++) g.Info[q] *= 1e-2;
      g.robust = gba ? gba_robust : true;
      B.G.push_back(g);
    }
  }
  if (stop && *stop) {
    R.status = VIEO_LBA_ABORTED;
    return;
  }
  optimize(B, gba ? gba_iterations : P.its0, stop, R, true);
  bool bDoMore = !(stop && *stop);
  if (gba) {
  } else if (bDoMore) {
    for (auto& e : B.E) {
      const double th = e.de == 2 ? 5.991 : 7.815;
      if (LBA::chi2(e) > th || !B.depth_positive(e)) e.level = 1;
      e.robust = false;
    }
    optimize(B, P.its1, stop, R, false);
  } else
    R.status = VIEO_LBA_ABORTED;
  for (int i = 0; i < n_obs && !gba; i++) {
    const LEdge& e = B.E[i];
    const double th = e.de == 2 ? 5.991 : 7.815;
    if (LBA::chi2(e) > th || !B.depth_positive(e)) erase[i] = 1, R.n_erase++;
  }
  for (int k = 0; k < n_kf; k++) {
    if (B.kf[k].fixed) continue;
    memcpy(navs_out[k].p, B.kf[k].p, 24);
    navs_out[k].q[0] = B.kf[k].q.w, navs_out[k].q[1] = B.kf[k].q.x;
    navs_out[k].q[2] = B.kf[k].q.y, navs_out[k].q[3] = B.kf[k].q.z;
  }
  for (int i = 0; i < n_mp * 3; i++) points_out[i] = (float)B.X[i];  // SetWorldPos(cast<float>)
}

}  // namespace vo

extern "C" void vo_local_bundle_adjustment(const vieo_lba_params* params, const vieo_lba_keyframe* kfs,
                                           int n_kf, const float* points, int n_mp,
                                           const vieo_lba_obs* obs, int n_obs, const int* stop,
                                           vieo_navstate* navs_out, float* points_out, uint8_t* erase,
                                           vieo_lba_result* result) {
  vo::local_ba(*params, kfs, n_kf, points, n_mp, obs, n_obs, stop, navs_out, points_out, erase, *result);
}

// test hook: camm::Camera::Project of one camera (image point as float, 2x3 Jacobian)
extern "C" void vo_cam_project(const vieo_camera* cam, const double* P, float* uv, double* J) {
  vieo_lba_params prm;
  memset(&prm, 0, sizeof(prm));
  prm.n_cams = 1, prm.cams = cam;
  vo::OCam c[4];
  vo::ocams_from_params(prm, c);
  vo::ocam_project(c[0], P, uv, J);
}

extern "C" void vo_bundle_adjustment(const vieo_lba_params* params, int n_iterations, int robust,
                                     const vieo_lba_keyframe* kfs, int n_kf, const float* points, int n_mp,
                                     const vieo_lba_obs* obs, int n_obs, const int* stop, vieo_navstate* navs_out,
                                     float* points_out, vieo_lba_result* result) {
  std::vector<uint8_t> erase((size_t)n_obs + 1);
  vo::local_ba(*params, kfs, n_kf, points, n_mp, obs, n_obs, stop, navs_out, points_out, erase.data(), *result,
               n_iterations, robust != 0);
}

extern "C" void vo_local_bundle_adjustment_enc(const vieo_lba_params* params, const vieo_lba_keyframe* kfs, int n_kf,
                                               const float* points, int n_mp, const vieo_lba_obs* obs, int n_obs,
                                               const vieo_lba_enc* enc, const int* stop, vieo_navstate* navs_out,
                                               float* points_out, uint8_t* erase, vieo_lba_result* result) {
  vo::local_ba(*params, kfs, n_kf, points, n_mp, obs, n_obs, stop, navs_out, points_out, erase, *result, -1, false,
               enc);
}

extern "C" void vo_bundle_adjustment_enc(const vieo_lba_params* params, int n_iterations, int robust,
                                         const vieo_lba_keyframe* kfs, int n_kf, const float* points, int n_mp,
                                         const vieo_lba_obs* obs, int n_obs, const vieo_lba_enc* enc, const int* stop,
                                         vieo_navstate* navs_out, float* points_out, vieo_lba_result* result) {
  std::vector<uint8_t> erase((size_t)n_obs + 1);
  vo::local_ba(*params, kfs, n_kf, points, n_mp, obs, n_obs, stop, navs_out, points_out, erase.data(), *result,
               n_iterations, robust != 0, enc);
}
