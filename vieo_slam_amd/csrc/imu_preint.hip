// imu_preint.hip -- IMUPreIntegratorBase::PreIntegration + update (reference src/Odom/OdomPreIntegrator.h:226-506,
// USE_PREINT_EULA off, forward time order) for a batch of intervals: one lane per interval walks its
// samples (the recursion over samples is sequential; the batch is the parallel axis), FP64 throughout.  The
// 9 x 9 covariance recursions exploit nothing: A Sigma A^T is formed densely in the lane's private arrays -- a few
// thousand flops per sample, the input producer of the pose optimisations, not a throughput kernel.
// A single interval (the sequential replay: one call per frame) is latency, not throughput: for fewer intervals than
// kWaveBelow the WAVE instantiation gives each interval a wavefront, keeps the two covariances in LDS and spreads
// the 81 entries of every 9 x 9 product over the lanes -- every entry still summed by one lane in the scalar
// version's order, so the result is bit-identical (2.1 -> 0.4 ms per call of ~100 samples).
#include <cstring>
#include "imu_device.h"

namespace vieo {

struct PreIntD {
  double R[9], v[3], p[3], JgR[9], Jgv[9], Jav[9], Jgp[9], Jap[9], dt;
  double *S, *Sprv;  // 9 x 9 covariances: the lane's private arrays, or the wavefront's LDS (WAVE)
  double* lds;       // WAVE: see preint_cov_wave
  int lane;
  const double* sig;  // WAVE: the 18 entries of sigma_g | sigma_a in global memory
};

__device__ __forceinline__ void preint_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The covariance step on a wavefront, both orderings at once: m = 0 is mSigmaijPRV (p, Phi, v: iR = 3, iV = 6), m = 1
// mSigmaij (p, v, Phi: iR = 6, iV = 3).  The two are independent, so every stage works on both between two
// wave_syncs (8 ordering points per step instead of 16, the 2 x 81 entries of a stage in three rounds of 64 lanes
// instead of four); each entry is the same sum in the same order as one pass after the other.
// LDS block of matrix m at P.lds + m * kCovSet: A[81] | T[81] | Bg[27] | Ba[27] | TB[27]; the noise matrices
// NgL[9] | NaL[9] behind both sets (written by the caller, entry e by lane e).
static const int kCovSet = 81 + 81 + 27 * 3, kCovNoise = 2 * kCovSet;
__device__ void preint_cov_wave(PreIntD& P, const double* dRt, const double* Rsk, const double* Jr, double dt, double dt2div2) {
  const int lane = P.lane;
  const double* NgL = P.lds + kCovNoise;
  for (int e2 = lane; e2 < 162; e2 += 64) {
    const int m = e2 >= 81, e = e2 - 81 * m;
    (P.lds + m * kCovSet)[e] = (e % 10) == 0 ? 1.0 : 0.0;
  }
  if (lane < 54) {
    const int m = lane >= 27, l = lane - 27 * m;
    double* Bg = P.lds + m * kCovSet + 162;
    Bg[l] = 0, Bg[27 + l] = 0;
  }
  preint_wave_sync();
  // lane e writes element e of the 3 x 3 blocks: compile-time element indices (dRt[lane] & co. would put the
  // arrays, which every lane holds in registers, into scratch memory)
#pragma unroll
  for (int e = 0; e < 9; e++)
    if (lane == e) {
      const int i = e / 3, j = e - 3 * i;
#pragma unroll
      for (int m = 0; m < 2; m++) {
        const int iR = m == 0 ? 3 : 6, iV = m == 0 ? 6 : 3;
        double* A = P.lds + m * kCovSet;
        double* Bg = A + 162;
        double* Ba = Bg + 27;
        A[(iR + i) * 9 + iR + j] = dRt[e] * 1.0;
        A[(iV + i) * 9 + iR + j] = Rsk[e] * -dt;
        A[i * 9 + iR + j] = Rsk[e] * -dt2div2;
        A[i * 9 + iV + j] = (i == j ? 1.0 : 0.0) * dt;
        Bg[(iR + i) * 3 + j] = Jr[e] * dt;
        Ba[(iV + i) * 3 + j] = P.R[e] * dt;
        Ba[i * 3 + j] = P.R[e] * dt2div2;
      }
    }
  preint_wave_sync();
  for (int e2 = lane; e2 < 162; e2 += 64) {
    const int m = e2 >= 81, e = e2 - 81 * m;
    const double* A = P.lds + m * kCovSet;
    const double* S = m == 0 ? P.Sprv : P.S;
    const int i = e / 9, j = e - 9 * i;
    double s = 0;
    for (int k = 0; k < 9; k++) s += A[i * 9 + k] * S[k * 9 + j];
    (P.lds + m * kCovSet + 81)[e] = s;
  }
  preint_wave_sync();
  for (int e2 = lane; e2 < 162; e2 += 64) {
    const int m = e2 >= 81, e = e2 - 81 * m;
    const double* A = P.lds + m * kCovSet;
    const double* T = A + 81;
    double* S = m == 0 ? P.Sprv : P.S;
    const int i = e / 9, j = e - 9 * i;
    double s = 0;
    for (int k = 0; k < 9; k++) s += T[i * 9 + k] * A[j * 9 + k];
    S[e] = s;
  }
  preint_wave_sync();
  for (int which = 0; which < 2; which++) {  // S += Bg Ng Bg^T, then S += Ba Na Ba^T
    const double* N = NgL + 9 * which;
    if (lane < 54) {
      const int m = lane >= 27, l = lane - 27 * m;
      const double* B = P.lds + m * kCovSet + 162 + 27 * which;
      const int i = l / 3, j = l - 3 * i;
      (P.lds + m * kCovSet + 216)[l] = B[i * 3] * N[j] + B[i * 3 + 1] * N[3 + j] + B[i * 3 + 2] * N[6 + j];
    }
    preint_wave_sync();
    for (int e2 = lane; e2 < 162; e2 += 64) {
      const int m = e2 >= 81, e = e2 - 81 * m;
      const double* B = P.lds + m * kCovSet + 162 + 27 * which;
      const double* TB = P.lds + m * kCovSet + 216;
      double* S = m == 0 ? P.Sprv : P.S;
      const int i = e / 9, j = e - 9 * i;
      S[e] += TB[i * 3] * B[j * 3] + TB[i * 3 + 1] * B[j * 3 + 1] + TB[i * 3 + 2] * B[j * 3 + 2];
    }
    preint_wave_sync();
  }
}

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
// (noinline: three inlined copies per kernel made the lane-per-interval instantiation drop the partial first step of
// some intervals -- caught by tests/test_imu_preint.py::test_wave_and_lane_instantiations_agree_bitwise; the
// out-of-line call is what round 1 shipped and what the parity test pins)
template <bool WAVE>
__device__ __forceinline__ void preint_update_body(PreIntD& P, const vieo_imu_noise& N, const double* omega, const double* acc, double dt) {
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
  if (WAVE) {
    // the step's noise matrices: entry e on lane e (the 18 divisions by dt were 18 x ~30 instructions on every lane),
    // straight into the wavefront's LDS block, where preint_cov_wave reads them (ordered by its first wave_sync)
    if (P.lane < 18) {
      const double sg = P.sig[P.lane];
      P.lds[kCovNoise + P.lane] = N.dt_cov_noise_fixed ? sg : (!N.freq_ref || dt < 1.5 / N.freq_ref) ? sg / dt : sg * N.freq_ref;
    }
  } else
    for (int i = 0; i < 9; i++) {
      if (N.dt_cov_noise_fixed)
        Ng[i] = N.sigma_g[i], Na[i] = N.sigma_a[i];
      else if (!N.freq_ref || dt < 1.5 / N.freq_ref)
        Ng[i] = N.sigma_g[i] / dt, Na[i] = N.sigma_a[i] / dt;
      else
        Ng[i] = N.sigma_g[i] * N.freq_ref, Na[i] = N.sigma_a[i] * N.freq_ref;
    }
  const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (WAVE) preint_cov_wave(P, dRt, Rsk, Jr, dt, dt2div2);
  for (int pass = 0; pass < 2 && !WAVE; pass++) {  // mSigmaijPRV (p, Phi, v), then mSigmaij (p, v, Phi)
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

// The lane-per-interval instantiation calls the update out of line (see above); the wavefront-per-interval one inlines it:
// out of line the integrator state `P` lives in scratch memory and every access inside the (serial) recursion is a
// memory round trip -- 265 us for one interval of ten samples.
__device__ __attribute__((noinline)) void preint_update_lane(PreIntD& P, const vieo_imu_noise& N, const double* omega, const double* acc, double dt) {
  preint_update_body<false>(P, N, omega, acc, dt);
}
template <bool WAVE>
__device__ __forceinline__ void preint_update(PreIntD& P, const vieo_imu_noise& N, const double* omega, const double* acc, double dt) {
  if constexpr (WAVE)
    preint_update_body<true>(P, N, omega, acc, dt);
  else
    preint_update_lane(P, N, omega, acc, dt);
}

static_assert(offsetof(vieo_imu_noise, sigma_a) == offsetof(vieo_imu_noise, sigma_g) + 9 * sizeof(double), "PreIntD::sig");

template <bool WAVE>
__global__ void __launch_bounds__(64)
k_imu_preint(const vieo_imu_noise* __restrict__ noise, const vieo_imu_sample* __restrict__ samples,
             const int32_t* __restrict__ first, const double* __restrict__ ti_, const double* __restrict__ tj_,
             const double* __restrict__ bg_, const double* __restrict__ ba_, int n, vieo_imu_preint* __restrict__ out,
             double* __restrict__ sigma_prv, int32_t* __restrict__ status) {
  const int k = WAVE ? blockIdx.x : blockIdx.x * 64 + threadIdx.x;
  if (k >= n) return;
  __shared__ double s_cov[WAVE ? 2 * 81 + 2 * kCovSet + 18 : 1];
  double S_priv[WAVE ? 1 : 81], Sprv_priv[WAVE ? 1 : 81];
  const vieo_imu_noise N = *noise;
  const vieo_imu_sample* L = samples + first[k];
  const int K = first[k + 1] - first[k];
  const double ti = ti_[k], tj = tj_[k];
  const double bg[3] = {bg_[3 * k], bg_[3 * k + 1], bg_[3 * k + 2]}, ba[3] = {ba_[3 * k], ba_[3 * k + 1], ba_[3 * k + 2]};
  PreIntD P;
  P.lane = threadIdx.x;
  P.sig = noise->sigma_g;  // (sigma_a follows it in vieo_imu_noise)
  P.S = WAVE ? s_cov : S_priv, P.Sprv = WAVE ? s_cov + 81 : Sprv_priv, P.lds = s_cov + 162;
  for (int i = 0; i < 9; i++) P.R[i] = (i % 4) == 0 ? 1.0 : 0.0, P.JgR[i] = P.Jgv[i] = P.Jav[i] = P.Jgp[i] = P.Jap[i] = 0;
  for (int i = 0; i < 3; i++) P.v[i] = P.p[i] = 0;
  if (WAVE) {
    for (int i = threadIdx.x; i < 162; i += 64) s_cov[i] = 0;
    preint_wave_sync();
  } else
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
      // up to three updates per sample pair (partial first step, the mid-point step, partial last step), collected
      // first and run by ONE loop body: three inlined copies of the update were 100 KB of code for a single wavefront
      double wq[3][3], aq[3][3], dtq[3];
      int nsub = 0;
      if (jm1 == iter_start) {
        const double dt_comple = L[jm1].t - ti;
        if (back ? dt_comple < 0 : dt_comple > 0) {
          for (int q = 0; q < 3; q++) wq[0][q] = imu.w[q] - bg[q], aq[0][q] = imu.a[q] - ba[q];
          dtq[0] = dt_comple;
          nsub = 1;
          dt -= dt_comple;
        }
      }
      if (dt != 0 || nsub == 0) {
        double dt_comple_stop = 0;
        if (j == iter_stop) {
          dt_comple_stop = tj - imu_now.t;
          if (back ? dt_comple_stop < 0 : dt_comple_stop > 0) dt -= dt_comple_stop;
        }
        for (int q = 0; q < 3; q++) {
          const double wm = (imu_now.w[q] + imu.w[q]) / 2 - bg[q], am = (imu_now.a[q] + imu.a[q]) / 2 - ba[q];
          if (nsub == 0) wq[0][q] = wm, aq[0][q] = am; else wq[1][q] = wm, aq[1][q] = am;
        }
        if (nsub == 0) dtq[0] = dt; else dtq[1] = dt;
        nsub++;
        if (back ? dt_comple_stop < 0 : dt_comple_stop > 0) {
          for (int q = 0; q < 3; q++) {
            const double wl = imu_now.w[q] - bg[q], al = imu_now.a[q] - ba[q];
            if (nsub == 1) wq[1][q] = wl, aq[1][q] = al; else wq[2][q] = wl, aq[2][q] = al;
          }
          if (nsub == 1) dtq[1] = dt_comple_stop; else dtq[2] = dt_comple_stop;
          nsub++;
        }
      }
#pragma nounroll
      for (int u = 0; u < nsub; u++) {
        double w[3], a[3];
        for (int q = 0; q < 3; q++) w[q] = u == 0 ? wq[0][q] : u == 1 ? wq[1][q] : wq[2][q], a[q] = u == 0 ? aq[0][q] : u == 1 ? aq[1][q] : aq[2][q];
        const double dtu = u == 0 ? dtq[0] : u == 1 ? dtq[1] : dtq[2];
        preint_update<WAVE>(P, N, w, a, dtu);
      }
    }
  }
  if (WAVE) {
    for (int i = threadIdx.x; i < 81; i += 64) {
      out[k].Sigma[i] = P.S[i];
      if (sigma_prv) sigma_prv[81 * (size_t)k + i] = P.Sprv[i];
    }
    if (threadIdx.x != 0) return;
  }
  status[k] = st;
  vieo_imu_preint& o = out[k];
  o.dt = P.dt;
  for (int i = 0; i < 9; i++) o.Rij[i] = P.R[i], o.JgR[i] = P.JgR[i], o.Jgv[i] = P.Jgv[i], o.Jav[i] = P.Jav[i], o.Jgp[i] = P.Jgp[i], o.Jap[i] = P.Jap[i];
  for (int i = 0; i < 3; i++) o.vij[i] = P.v[i], o.pij[i] = P.p[i];
  if (WAVE) return;
  for (int i = 0; i < 81; i++) o.Sigma[i] = P.S[i];
  if (sigma_prv)
    for (int i = 0; i < 81; i++) sigma_prv[81 * (size_t)k + i] = P.Sprv[i];
}

static const int kWaveBelow = 1024;  // intervals per call below which every interval gets a wavefront

}  // namespace vieo

using namespace vieo;

extern "C" int vieo_imu_preintegrate_batch_device(const vieo_imu_noise* d_noise, const vieo_imu_sample* d_samples,
                                                  const int32_t* d_first, const double* d_ti, const double* d_tj,
                                                  const double* d_bg, const double* d_ba, int n, vieo_imu_preint* d_out,
                                                  double* d_sigma_prv, int32_t* d_status, void* stream) {
  if (!d_noise || n <= 0 || !d_first || !d_ti || !d_tj || !d_bg || !d_ba || !d_out || !d_status) return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  if (n < kWaveBelow)
    hipLaunchKernelGGL(k_imu_preint<true>, dim3(n), dim3(64), 0, (hipStream_t)stream, d_noise, d_samples, d_first, d_ti,
                       d_tj, d_bg, d_ba, n, d_out, d_sigma_prv, d_status);
  else
    hipLaunchKernelGGL(k_imu_preint<false>, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_noise, d_samples,
                       d_first, d_ti, d_tj, d_bg, d_ba, n, d_out, d_sigma_prv, d_status);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

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
  // one device block, one pinned staging block: [noise | ti | tj | bg | ba | first | samples] up,
  // [out | sigma_prv | status] down (ten synchronous copies of pageable memory were 2 ms per call)
  static thread_local DevBuf dev;
  static thread_local PinnedBuf pin;
  auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
  const size_t o_noise = 0, o_ti = al(sizeof(vieo_imu_noise)), o_tj = o_ti + al((size_t)n * 8);
  const size_t o_bg = o_tj + al((size_t)n * 8), o_ba = o_bg + al((size_t)n * 24), o_first = o_ba + al((size_t)n * 24);
  const size_t o_samples = o_first + al((size_t)(n + 1) * 4);
  const size_t up = o_samples + al((size_t)total * sizeof(vieo_imu_sample));
  const size_t o_out = up, o_prv = o_out + al((size_t)n * sizeof(vieo_imu_preint));
  const size_t o_st = o_prv + al((size_t)n * 81 * 8), all = o_st + al((size_t)n * 4);
  if ((rc = dev.ensure(all)) != VIEO_OK || (rc = pin.ensure(all)) != VIEO_OK) return rc;
  uint8_t* h = (uint8_t*)pin.p;
  uint8_t* d = (uint8_t*)dev.p;
  memcpy(h + o_noise, noise, sizeof(vieo_imu_noise));
  memcpy(h + o_ti, h_ti, (size_t)n * 8), memcpy(h + o_tj, h_tj, (size_t)n * 8);
  memcpy(h + o_bg, h_bg, (size_t)n * 24), memcpy(h + o_ba, h_ba, (size_t)n * 24);
  memcpy(h + o_first, h_first, (size_t)(n + 1) * 4);
  if (total > 0) memcpy(h + o_samples, h_samples, (size_t)total * sizeof(vieo_imu_sample));
  VIEO_HIP_CHECK(hipMemcpyAsync(d, h, up, hipMemcpyHostToDevice, 0));
  if (n < kWaveBelow)
    hipLaunchKernelGGL(k_imu_preint<true>, dim3(n), dim3(64), 0, 0, (const vieo_imu_noise*)(d + o_noise),
                       (const vieo_imu_sample*)(d + o_samples), (const int32_t*)(d + o_first), (const double*)(d + o_ti),
                       (const double*)(d + o_tj), (const double*)(d + o_bg), (const double*)(d + o_ba), n,
                       (vieo_imu_preint*)(d + o_out), (double*)(d + o_prv), (int32_t*)(d + o_st));
  else
    hipLaunchKernelGGL(k_imu_preint<false>, dim3((n + 63) / 64), dim3(64), 0, 0, (const vieo_imu_noise*)(d + o_noise),
                       (const vieo_imu_sample*)(d + o_samples), (const int32_t*)(d + o_first), (const double*)(d + o_ti),
                       (const double*)(d + o_tj), (const double*)(d + o_bg), (const double*)(d + o_ba), n,
                       (vieo_imu_preint*)(d + o_out), (double*)(d + o_prv), (int32_t*)(d + o_st));
  VIEO_HIP_CHECK(hipGetLastError());
  VIEO_HIP_CHECK(hipMemcpyAsync(h + o_out, d + o_out, all - o_out, hipMemcpyDeviceToHost, 0));
  VIEO_HIP_CHECK(hipStreamSynchronize(0));
  memcpy(h_out, h + o_out, (size_t)n * sizeof(vieo_imu_preint));
  if (h_sigma_prv) memcpy(h_sigma_prv, h + o_prv, (size_t)n * 81 * 8);
  memcpy(h_status, h + o_st, (size_t)n * 4);
  return VIEO_OK;
}
