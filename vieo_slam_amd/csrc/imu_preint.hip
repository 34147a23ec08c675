// imu_preint.hip -- IMUPreIntegratorBase::PreIntegration + update (reference src/Odom/OdomPreIntegrator.h:226-506,
// USE_PREINT_EULA off, forward time order) for a batch of intervals: one lane per interval walks its
// samples (the recursion over samples is sequential; the batch is the parallel axis), FP64 throughout.  The
// 9 x 9 covariance recursions exploit nothing: A Sigma A^T is formed densely in the lane's private arrays -- a few
// thousand flops per sample, the input producer of the pose optimisations, not a throughput kernel.
#include "imu_device.h"

namespace vieo {

struct PreIntD {
  double R[9], v[3], p[3], JgR[9], Jgv[9], Jav[9], Jgp[9], Jap[9], S[81], Sprv[81], dt;
};

__device__ void preint_sandwich(const double* A, double* S) {  // S <- A S A^T
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
__device__ void preint_noise(double* S, const double* B, const double* N) {  // S += B N B^T, B 9x3
  double T[27];
  for (int i = 0; i < 9; i++)
    for (int j = 0; j < 3; j++) T[i * 3 + j] = B[i * 3] * N[j] + B[i * 3 + 1] * N[3 + j] + B[i * 3 + 2] * N[6 + j];
  for (int i = 0; i < 9; i++)
    for (int j = 0; j < 9; j++) S[i * 9 + j] += T[i * 3] * B[j * 3] + T[i * 3 + 1] * B[j * 3 + 1] + T[i * 3 + 2] * B[j * 3 + 2];
}
__device__ __forceinline__ void preint_block(double* M, int ld, int r0, int c0, const double* B, double s) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M[(r0 + i) * ld + c0 + j] = B[i * 3 + j] * s;
}

// IMUPreIntegratorBase::update (OdomPreIntegrator.h:430-506)
__device__ void preint_update(PreIntD& P, const vieo_imu_noise& N, const double* omega, const double* acc, double dt) {
  const double dt2div2 = dt * dt / 2;
  const double wdt[3] = {omega[0] * dt, omega[1] * dt, omega[2] * dt};
  double dR[9], Jr[9], skewa[9], dRt[9], Rsk[9];
  q_to_R(so3_exp_q(wdt), dR);
  so3_Jr_d(wdt, Jr);
  hat3(acc, skewa);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) dRt[i * 3 + j] = dR[j * 3 + i];
  mm3(P.R, skewa, Rsk);
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
  for (int pass = 0; pass < 2; pass++) {  // mSigmaijPRV (p, Phi, v), then mSigmaij (p, v, Phi)
    const int iR = pass == 0 ? 3 : 6, iV = pass == 0 ? 6 : 3;
    double A[81], Bg[27], Ba[27];
    for (int i = 0; i < 81; i++) A[i] = (i % 10) == 0 ? 1.0 : 0.0;
    for (int i = 0; i < 27; i++) Bg[i] = 0, Ba[i] = 0;
    preint_block(A, 9, iR, iR, dRt, 1.0);
    preint_block(A, 9, iV, iR, Rsk, -dt);
    preint_block(A, 9, 0, iR, Rsk, -dt2div2);
    preint_block(A, 9, 0, iV, I3, dt);
    preint_block(Bg, 3, iR, 0, Jr, dt);
    preint_block(Ba, 3, iV, 0, P.R, dt);
    preint_block(Ba, 3, 0, 0, P.R, dt2div2);
    double* S = pass == 0 ? P.Sprv : P.S;
    preint_sandwich(A, S);
    preint_noise(S, Bg, Ng);
    preint_noise(S, Ba, Na);
  }
  double RskJ[9], t9[9];
  mm3(Rsk, P.JgR, RskJ);
  for (int i = 0; i < 9; i++) P.Jap[i] += P.Jav[i] * dt - P.R[i] * dt2div2;
  for (int i = 0; i < 9; i++) P.Jgp[i] += P.Jgv[i] * dt - RskJ[i] * dt2div2;
  for (int i = 0; i < 9; i++) P.Jav[i] += -P.R[i] * dt;
  for (int i = 0; i < 9; i++) P.Jgv[i] += -RskJ[i] * dt;
  mm3(dRt, P.JgR, t9);
  for (int i = 0; i < 9; i++) P.JgR[i] = t9[i] - Jr[i] * dt;
  const double a2[3] = {acc[0] * dt2div2, acc[1] * dt2div2, acc[2] * dt2div2};
  const double a1[3] = {acc[0] * dt, acc[1] * dt, acc[2] * dt};
  double r[3];
  mv3(P.R, a2, r);
  for (int i = 0; i < 3; i++) P.p[i] += P.v[i] * dt + r[i];
  mv3(P.R, a1, r);
  for (int i = 0; i < 3; i++) P.v[i] += r[i];
  double RdR[9];
  mm3(P.R, dR, RdR);
  Qd q = R_to_q(RdR);  // SO3ex::normalizeRotationM (so3_extra.h:217-229)
  if (q.w < 0) q.w = -q.w, q.x = -q.x, q.y = -q.y, q.z = -q.z;
  q_to_R(q_norm(q), P.R);
  P.dt += dt;
}

__global__ void __launch_bounds__(64)
k_imu_preint(const vieo_imu_noise* __restrict__ noise, const vieo_imu_sample* __restrict__ samples,
             const int32_t* __restrict__ first, const double* __restrict__ ti_, const double* __restrict__ tj_,
             const double* __restrict__ bg_, const double* __restrict__ ba_, int n, vieo_imu_preint* __restrict__ out,
             double* __restrict__ sigma_prv, int32_t* __restrict__ status) {
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= n) return;
  const vieo_imu_noise N = *noise;
  const vieo_imu_sample* L = samples + first[k];
  const int K = first[k + 1] - first[k];
  const double ti = ti_[k], tj = tj_[k];
  const double bg[3] = {bg_[3 * k], bg_[3 * k + 1], bg_[3 * k + 2]}, ba[3] = {ba_[3 * k], ba_[3 * k + 1], ba_[3 * k + 2]};
  PreIntD P;
  for (int i = 0; i < 9; i++) P.R[i] = (i % 4) == 0 ? 1.0 : 0.0, P.JgR[i] = P.Jgv[i] = P.Jav[i] = P.Jgp[i] = P.Jap[i] = 0;
  for (int i = 0; i < 3; i++) P.v[i] = P.p[i] = 0;
  for (int i = 0; i < 81; i++) P.S[i] = P.Sprv[i] = 0;
  P.dt = 0;
  int st = VIEO_PREINT_OK;
  if (K <= 0)
    st = VIEO_PREINT_EMPTY;
  else {
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
      const int t = iter_start;
      iter_start = iter_stop, iter_stop = t;
      if (L[iter_stop].t > timemin) iter_stop = K;  // (only at the first sample) run down to it, then stop
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
      if (fabs(dt) > 1.5) {
        P.dt = 0;
        st = VIEO_PREINT_GAP;
        break;
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
          preint_update(P, N, w, a, dt_comple);
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
      preint_update(P, N, w, a, dt);
      if (back ? dt_comple_stop < 0 : dt_comple_stop > 0) {
        for (int q = 0; q < 3; q++) w[q] = imu_now.w[q] - bg[q], a[q] = imu_now.a[q] - ba[q];
        preint_update(P, N, w, a, dt_comple_stop);
      }
    }
  }
  status[k] = st;
  vieo_imu_preint& o = out[k];
  o.dt = P.dt;
  for (int i = 0; i < 9; i++) o.Rij[i] = P.R[i], o.JgR[i] = P.JgR[i], o.Jgv[i] = P.Jgv[i], o.Jav[i] = P.Jav[i], o.Jgp[i] = P.Jgp[i], o.Jap[i] = P.Jap[i];
  for (int i = 0; i < 3; i++) o.vij[i] = P.v[i], o.pij[i] = P.p[i];
  for (int i = 0; i < 81; i++) o.Sigma[i] = P.S[i];
  if (sigma_prv)
    for (int i = 0; i < 81; i++) sigma_prv[81 * (size_t)k + i] = P.Sprv[i];
}

}  // namespace vieo

using namespace vieo;

extern "C" int vieo_imu_preintegrate_batch(const vieo_imu_noise* noise, const vieo_imu_sample* h_samples,
                                           const int32_t* h_first, const double* h_ti, const double* h_tj,
                                           const double* h_bg, const double* h_ba, int n, vieo_imu_preint* h_out,
                                           double* h_sigma_prv, int32_t* h_status) {
  if (!noise || n < 0 || (n > 0 && (!h_first || !h_ti || !h_tj || !h_bg || !h_ba || !h_out || !h_status)))
    return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  if (n == 0) return VIEO_OK;
  const int total = h_first[n];
  if (total < 0 || (total > 0 && !h_samples)) return VIEO_E_INVALID;
  for (int k = 0; k < n; k++)
    if (h_first[k + 1] < h_first[k]) return VIEO_E_INVALID;
  static thread_local DevBuf dN, dS, dF, dT, dB, dO, dP, dSt;
#define ENS(b, bytes) \
  if ((rc = (b).ensure(std::max<size_t>(bytes, 8))) != VIEO_OK) return rc
  ENS(dN, sizeof(vieo_imu_noise));
  ENS(dS, (size_t)total * sizeof(vieo_imu_sample));
  ENS(dF, (size_t)(n + 1) * 4);
  ENS(dT, (size_t)n * 16);
  ENS(dB, (size_t)n * 48);
  ENS(dO, (size_t)n * sizeof(vieo_imu_preint));
  ENS(dP, (size_t)n * 81 * 8);
  ENS(dSt, (size_t)n * 4);
#undef ENS
  VIEO_HIP_CHECK(hipMemcpy(dN.p, noise, sizeof(vieo_imu_noise), hipMemcpyHostToDevice));
  if (total > 0) VIEO_HIP_CHECK(hipMemcpy(dS.p, h_samples, (size_t)total * sizeof(vieo_imu_sample), hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(hipMemcpy(dF.p, h_first, (size_t)(n + 1) * 4, hipMemcpyHostToDevice));
  double* dti = dT.as<double>();
  double* dtj = dti + n;
  double* dbg = dB.as<double>();
  double* dba = dbg + 3 * (size_t)n;
  VIEO_HIP_CHECK(hipMemcpy(dti, h_ti, (size_t)n * 8, hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(hipMemcpy(dtj, h_tj, (size_t)n * 8, hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(hipMemcpy(dbg, h_bg, (size_t)n * 24, hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(hipMemcpy(dba, h_ba, (size_t)n * 24, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_imu_preint, dim3((n + 63) / 64), dim3(64), 0, 0, dN.as<vieo_imu_noise>(),
                     dS.as<vieo_imu_sample>(), dF.as<int32_t>(), dti, dtj, dbg, dba, n, dO.as<vieo_imu_preint>(),
                     dP.as<double>(), dSt.as<int32_t>());
  VIEO_HIP_CHECK(hipGetLastError());
  VIEO_HIP_CHECK(hipMemcpy(h_out, dO.p, (size_t)n * sizeof(vieo_imu_preint), hipMemcpyDeviceToHost));
  if (h_sigma_prv) VIEO_HIP_CHECK(hipMemcpy(h_sigma_prv, dP.p, (size_t)n * 81 * 8, hipMemcpyDeviceToHost));
  VIEO_HIP_CHECK(hipMemcpy(h_status, dSt.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  return VIEO_OK;
}
