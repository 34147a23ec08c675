// ORACLE -- TEST INFRASTRUCTURE ONLY (parity unpinned: no golden vectors in the reference; OpenCV / Eigen /
// Sophus absent, so the reference cannot be built here).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use this file; the product never links or calls it.
//
// Sequential restatement of IMUPreIntegratorBase::PreIntegration + update (reference
// src/Odom/OdomPreIntegrator.h:226-506; USE_PREINT_EULA off, forward time order) with SO3ex::Exp / JacobianR /
// normalizeRotationM (common/so3_extra.h:121-142,226-229,255-270).
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../include/vieo_hot.h"
#include "smallmat.hpp"

namespace vo {

struct PreInt {
  double R[9], v[3], p[3], JgR[9], Jgv[9], Jav[9], Jgp[9], Jap[9], S[81], Sprv[81], dt;
  void reset() {
    memset(this, 0, sizeof(*this));
    R[0] = R[4] = R[8] = 1;
  }
};

static void mat9_sandwich(const double* A, double* S) {  // S <- A S A^T
  double T[81];
  for (int i = 0; i < 9; i++)
    for (int j = 0; j < 9; j++) {
      double s = 0;
      for (int k = 0; k < 9; k++) s += A[i * 9 + k] * S[k * 9 + j];
      T[i * 9 + j] = s;
    }
  for (int i = 0; i < 9; i++)
    for (int j = 0; j < 9; j++) {
      double s = 0;
      for (int k = 0; k < 9; k++) s += T[i * 9 + k] * A[j * 9 + k];
      S[i * 9 + j] = s;
    }
}
// S += B N B^T for a 9x3 B whose only non-zero 3x3 blocks start at the rows in rows[] (nb of them)
static void add_noise(double* S, const double* B, const double* N) {
  double T[27];
  for (int i = 0; i < 9; i++)
    for (int j = 0; j < 3; j++) T[i * 3 + j] = B[i * 3] * N[j] + B[i * 3 + 1] * N[3 + j] + B[i * 3 + 2] * N[6 + j];
  for (int i = 0; i < 9; i++)
    for (int j = 0; j < 9; j++) S[i * 9 + j] += T[i * 3] * B[j * 3] + T[i * 3 + 1] * B[j * 3 + 1] + T[i * 3 + 2] * B[j * 3 + 2];
}
static void set_block(double* M, int ld, int r0, int c0, const double* B, double s) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M[(r0 + i) * ld + c0 + j] = B[i * 3 + j] * s;
}

static void update(PreInt& P, const vieo_imu_noise& N, const double* omega, const double* acc, double dt) {
  const double dt2div2 = dt * dt / 2;
  const double wdt[3] = {omega[0] * dt, omega[1] * dt, omega[2] * dt};
  double dR[9], Jr[9], skewa[9], dRt[9], Rsk[9];
  quat_to_R(so3_exp(wdt), dR);
  so3_Jr(wdt, Jr);
  hat(acc, skewa);
  m3_T(dR, dRt);
  m3_mul(P.R, skewa, Rsk);
  double Ng[9], Na[9];
  for (int i = 0; i < 9; i++) {
    if (N.dt_cov_noise_fixed)
      Ng[i] = N.sigma_g[i], Na[i] = N.sigma_a[i];
    else if (!N.freq_ref || dt < 1.5 / N.freq_ref)
      Ng[i] = N.sigma_g[i] / dt, Na[i] = N.sigma_a[i] / dt;
    else
      Ng[i] = N.sigma_g[i] * N.freq_ref, Na[i] = N.sigma_a[i] * N.freq_ref;
  }
  const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  // order p, Phi, v (mSigmaijPRV) then p, v, Phi (mSigmaij): rows (iR, iV) = (3, 6) / (6, 3)
  for (int pass = 0; pass < 2; pass++) {
    const int iR = pass == 0 ? 3 : 6, iV = pass == 0 ? 6 : 3;
    double A[81] = {0}, Bg[27] = {0}, Ba[27] = {0};
    for (int i = 0; i < 9; i++) A[i * 9 + i] = 1;
    set_block(A, 9, iR, iR, dRt, 1.0);
    set_block(A, 9, iV, iR, Rsk, -dt);
    set_block(A, 9, 0, iR, Rsk, -dt2div2);
    set_block(A, 9, 0, iV, I3, dt);
    set_block(Bg, 3, iR, 0, Jr, dt);
    set_block(Ba, 3, iV, 0, P.R, dt);
    set_block(Ba, 3, 0, 0, P.R, dt2div2);
    double* S = pass == 0 ? P.Sprv : P.S;
    mat9_sandwich(A, S);
    add_noise(S, Bg, Ng);
    add_noise(S, Ba, Na);
  }
  double RskJ[9];
  m3_mul(Rsk, P.JgR, RskJ);
  for (int i = 0; i < 9; i++) P.Jap[i] += P.Jav[i] * dt - P.R[i] * dt2div2;
  for (int i = 0; i < 9; i++) P.Jgp[i] += P.Jgv[i] * dt - RskJ[i] * dt2div2;
  for (int i = 0; i < 9; i++) P.Jav[i] += -P.R[i] * dt;
  for (int i = 0; i < 9; i++) P.Jgv[i] += -RskJ[i] * dt;
  double t9[9];
  m3_mul(dRt, P.JgR, t9);
  for (int i = 0; i < 9; i++) P.JgR[i] = t9[i] - Jr[i] * dt;
  double a2[3] = {acc[0] * dt2div2, acc[1] * dt2div2, acc[2] * dt2div2}, a1[3] = {acc[0] * dt, acc[1] * dt, acc[2] * dt}, r[3];
  m3_v(P.R, a2, r);
  for (int i = 0; i < 3; i++) P.p[i] += P.v[i] * dt + r[i];
  m3_v(P.R, a1, r);
  for (int i = 0; i < 3; i++) P.v[i] += r[i];
  double RdR[9];
  m3_mul(P.R, dR, RdR);
  Quat q = R_to_quat(RdR);  // normalizeRotationM: Quaternion(R), w >= 0, normalized, back to a matrix
  if (q.w < 0) q.w = -q.w, q.x = -q.x, q.y = -q.y, q.z = -q.z;
  quat_normalize(q);
  quat_to_R(q, P.R);
  P.dt += dt;
}

// returns the VIEO_PREINT_* status
static int preintegrate(const vieo_imu_noise& N, const vieo_imu_sample* L, int K, double ti, double tj,
                        const double* bg, const double* ba, PreInt& P) {
  P.reset();
  if (K <= 0) return VIEO_PREINT_EMPTY;
  // timeStampi > timeStampj (map reuse): the samples are walked backwards with negative steps (:241-262)
  const bool back = ti > tj;
  const double timemin = back ? tj : ti, timemax = back ? ti : tj;
  int iter_start = 0, iter_stop = K;
  for (int j = 0; j != K && L[j].t <= timemin; iter_start = j++) {
  }
  for (int j = K; j != 0;) {
    iter_stop = j--;
    if (L[j].t >= timemax) continue;
    break;
  }
  if (back) {
    if (iter_stop == K) --iter_stop;
    std::swap(iter_start, iter_stop);
    if (L[iter_stop].t > timemin) iter_stop = K;  // assert(iter_stop == iterBegin): run down to the first sample
  }
  for (int j = iter_start; j != iter_stop;) {
    const int jm1 = j;
    if (!back)
      ++j;
    else if (j == 0)
      j = iter_stop;
    else
      --j;
    const double tj_1 = jm1 == iter_start ? ti : L[jm1].t;
    const double tjj = j == iter_stop ? tj : L[j].t;
    double dt = tjj - tj_1;
    if (dt == 0) continue;
    if (std::fabs(dt) > 1.5) {
      P.dt = 0;
      return VIEO_PREINT_GAP;
    }
    vieo_imu_sample imu = L[jm1], imu_now = j != K ? L[j] : imu;
    if (j != K) {
      if (j == iter_stop) {
        const double dt_tmp = L[j].t - tj;
        if (back ? dt_tmp < 0 : dt_tmp > 0) {
          const double rat = dt_tmp / (L[j].t - L[jm1].t);
          for (int a = 0; a < 3; a++)
            imu_now.w[a] = rat * imu.w[a] + (1 - rat) * imu_now.w[a], imu_now.a[a] = rat * imu.a[a] + (1 - rat) * imu_now.a[a];
        }
      }
      if (jm1 == iter_start) {
        const double dt_tmp = ti - L[jm1].t;
        if (back ? dt_tmp < 0 : dt_tmp > 0) {
          const double rat = dt_tmp / (L[j].t - L[jm1].t);
          for (int a = 0; a < 3; a++)
            imu.w[a] = (1 - rat) * imu.w[a] + rat * imu_now.w[a], imu.a[a] = (1 - rat) * imu.a[a] + rat * imu_now.a[a];
        }
      }
    }
    double w[3], a[3];
    if (jm1 == iter_start) {
      const double dt_comple = L[jm1].t - ti;
      if (back ? dt_comple < 0 : dt_comple > 0) {
        for (int q = 0; q < 3; q++) w[q] = imu.w[q] - bg[q], a[q] = imu.a[q] - ba[q];
        update(P, N, w, a, dt_comple);
        dt -= dt_comple;
        if (!dt) continue;
      }
    }
    double dt_comple_stop = 0;
    if (j == iter_stop) {
      dt_comple_stop = tj - imu_now.t;
      if (back ? dt_comple_stop < 0 : dt_comple_stop > 0) dt -= dt_comple_stop;
    }
    for (int q = 0; q < 3; q++) w[q] = (imu_now.w[q] + imu.w[q]) / 2 - bg[q], a[q] = (imu_now.a[q] + imu.a[q]) / 2 - ba[q];
    update(P, N, w, a, dt);
    if (back ? dt_comple_stop < 0 : dt_comple_stop > 0) {
      for (int q = 0; q < 3; q++) w[q] = imu_now.w[q] - bg[q], a[q] = imu_now.a[q] - ba[q];
      update(P, N, w, a, dt_comple_stop);
    }
  }
  return VIEO_PREINT_OK;
}

}  // namespace vo

extern "C" void vo_imu_preintegrate_batch(const vieo_imu_noise* noise, const vieo_imu_sample* samples,
                                          const int32_t* first, const double* ti, const double* tj,
                                          const double* bg, const double* ba, int n, vieo_imu_preint* out,
                                          double* sigma_prv, int32_t* status) {
  for (int k = 0; k < n; k++) {
    vo::PreInt P;
    status[k] = vo::preintegrate(*noise, samples + first[k], first[k + 1] - first[k], ti[k], tj[k], bg + 3 * k,
                                 ba + 3 * k, P);
    vieo_imu_preint& o = out[k];
    o.dt = P.dt;
    memcpy(o.Rij, P.R, 72), memcpy(o.vij, P.v, 24), memcpy(o.pij, P.p, 24);
    memcpy(o.JgR, P.JgR, 72), memcpy(o.Jgv, P.Jgv, 72), memcpy(o.Jav, P.Jav, 72);
    memcpy(o.Jgp, P.Jgp, 72), memcpy(o.Jap, P.Jap, 72), memcpy(o.Sigma, P.S, 648);
    if (sigma_prv) memcpy(sigma_prv + 81 * (size_t)k, P.Sprv, 648);
  }
}
