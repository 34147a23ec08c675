// lba.hip -- Optimizer::LocalBundleAdjustment on gfx950 (reference: src/Optimizer.cc:1876-2307;
// g2o BlockSolver<6,3> with Schur complement, block_solver.hpp:353-589; LM
// optimization_algorithm_levenberg.cpp:61-207; edges src/Odom/g2otypes.h:321-547).
//
// Device side (all FP64):
//   k_lba_error     edge-parallel residuals + robust chi2 (per-block partial sums)
//   k_lba_linearize landmark-parallel (16 lanes per landmark, observations of a point are
//                   contiguous): Jacobians, H_ll (3x3), b_l, and the 6x3 block B = Jp^T W Jx of every
//                   observation
//   k_lba_pose      key-frame-parallel H_pp (6x6) and b_p over each key frame's edge list
//   k_lba_schur     one wavefront per landmark: D^-1, then the dense landmark-block contraction
//                   [B_1 D^-1; ...; B_k D^-1] x [B_1; ...; B_k]^T on the FP64 matrix cores
//                   (v_mfma_f64_16x16x4_f64, K = 3 padded to 4, 6k rows tiled by 16), accumulated into a
//                   per-workgroup LDS copy of the reduced pose system, flushed with one atomic pass
//   k_lba_ldlt      one workgroup: dense LDL^T of the reduced system (<= 6 x #free key frames)
//   k_lba_update    back-substitution x_l = D^-1 (b_l - B^T x_p), point / pose retraction, scale terms
// The Levenberg-Marquardt control flow (lambda policy, accept / reject, stop flag polling) runs on
// the host exactly as g2o's does; it only reads a handful of scalars per trial.
#include <vector>

#include "ba_device.h"

namespace vieo {

struct LbaKf {
  double p[3], qw, qx, qy, qz;
  int col;  // offset in the reduced pose system, -1 = fixed / inactive
  int pad;
};

struct LbaDev {
  const vieo_lba_obs* obs;
  int n_obs, n_mp, n_kf, np;
  LbaKf* kf;
  double* X;               // [n_mp][3]
  double* err;             // [n_obs][3]
  unsigned char* level;    // [n_obs]
  const int* mp_first;     // [n_mp]
  const int* mp_count;     // [n_mp]
  const unsigned char* mp_act;  // [n_mp]
  double *Bpl, *Hll, *bl, *Hpp, *bp, *Hs, *bs, *Dinv, *xp, *xl;
  double* part;            // per-block partial sums
  CamD cam;
  int robust;
  double dMono, dStereo;
};

__device__ __forceinline__ void kf_xf(const CamD& c, const LbaKf& k, PoseXf& X) {
  Est e;
  e.p[0] = k.p[0], e.p[1] = k.p[1], e.p[2] = k.p[2];
  e.qw = k.qw, e.qx = k.qx, e.qy = k.qy, e.qz = k.qz;
  make_xf(c, e, X);
}

__device__ __forceinline__ vieo_pose_obs as_pose_obs(const vieo_lba_obs& o, const double* X) {
  vieo_pose_obs p;
  p.Xw[0] = 0, p.Xw[1] = 0, p.Xw[2] = 0;  // double position passed separately
  p.u = o.u, p.v = o.v, p.ur = o.ur, p.inv_sigma2 = o.inv_sigma2, p.flags = 0;
  (void)X;
  return p;
}

// residual with a double-precision point (the LBA point vertex is double, unlike PoseOpt's)
__device__ __forceinline__ double lba_edge_error(const CamD& c, const PoseXf& X, const vieo_lba_obs& o,
                                                 const double* Xw, double* err, double* Pc) {
  for (int i = 0; i < 3; i++)
    Pc[i] = X.Rcw[i * 3] * Xw[0] + X.Rcw[i * 3 + 1] * Xw[1] + X.Rcw[i * 3 + 2] * Xw[2] + X.tcw[i];
  const double invz = 1. / Pc[2];
  const double u = (double)(float)(c.fx * Pc[0] * invz + c.cx);
  const double v = (double)(float)(c.fy * Pc[1] * invz + c.cy);
  err[0] = (double)o.u - u;
  err[1] = (double)o.v - v;
  const double info = (double)o.inv_sigma2;
  double chi2 = err[0] * (info * err[0]) + err[1] * (info * err[1]);
  if (o.ur >= 0) {
    err[2] = (double)o.ur - (u - c.bf / Pc[2]);
    chi2 += err[2] * (info * err[2]);
  } else
    err[2] = 0;
  return chi2;
}

// Jp (3x6) and Jx (3x3) of one edge
__device__ __forceinline__ void lba_jacobians(const CamD& c, const PoseXf& X, const double* kfp,
                                              const double* Xw, const double* Pc, double* Jp, double* Jx) {
  const double invz = 1 / Pc[2], invz2 = invz * invz;
  double J[9];
  J[0] = -(c.fx * invz), J[1] = 0, J[2] = -(-c.fx * Pc[0] * invz2);
  J[3] = 0, J[4] = -(c.fy * invz), J[5] = -(-c.fy * Pc[1] * invz2);
  J[6] = J[0], J[7] = J[1], J[8] = J[2] - c.bf * invz2;
  const double d0 = Xw[0] - kfp[0], d1 = Xw[1] - kfp[1], d2 = Xw[2] - kfp[2];
  double Pa[3], RH[9];
  for (int m = 0; m < 3; m++) Pa[m] = X.Rwb[m] * d0 + X.Rwb[3 + m] * d1 + X.Rwb[6 + m] * d2;
  for (int m = 0; m < 3; m++) {
    const double a = c.Rcb[m * 3], b = c.Rcb[m * 3 + 1], d = c.Rcb[m * 3 + 2];
    RH[m * 3 + 0] = b * Pa[2] - d * Pa[1];
    RH[m * 3 + 1] = -a * Pa[2] + d * Pa[0];
    RH[m * 3 + 2] = a * Pa[1] - b * Pa[0];
  }
  for (int r = 0; r < 3; r++)
    for (int q = 0; q < 3; q++) {
      Jp[r * 6 + q] = -(J[r * 3] * c.Rcb[q] + J[r * 3 + 1] * c.Rcb[3 + q] + J[r * 3 + 2] * c.Rcb[6 + q]);
      Jp[r * 6 + 3 + q] = J[r * 3] * RH[q] + J[r * 3 + 1] * RH[3 + q] + J[r * 3 + 2] * RH[6 + q];
      Jx[r * 3 + q] = J[r * 3] * X.Rcw[q] + J[r * 3 + 1] * X.Rcw[3 + q] + J[r * 3 + 2] * X.Rcw[6 + q];
    }
}

// ---- residuals + robust chi2 of the active edges; mode 1: classify (level) instead
__global__ void __launch_bounds__(256) k_lba_error(LbaDev D) {
  __shared__ double s_red[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double v[1] = {0};
  if (i < D.n_obs && D.level[i] == 0) {
    const vieo_lba_obs o = D.obs[i];
    PoseXf X;
    kf_xf(D.cam, D.kf[o.kf], X);
    double err[3], Pc[3];
    const double chi2 = lba_edge_error(D.cam, X, o, D.X + 3 * (size_t)o.mp, err, Pc);
    D.err[3 * (size_t)i] = err[0], D.err[3 * (size_t)i + 1] = err[1], D.err[3 * (size_t)i + 2] = err[2];
    double r0 = chi2, r1;
    if (D.robust) {
      const double dl = o.ur >= 0 ? D.dStereo : D.dMono;
      huber(chi2, dl, dl * dl, &r0, &r1);
    }
    v[0] = r0;
  }
  block_sum<1>(v, s_red, threadIdx.x);
  if (threadIdx.x == 0) D.part[blockIdx.x] = v[0];
}

// chi2 (from the STORED error, as the reference does) / depth classification.
// what = 0: set level 1 for outliers (Optimizer.cc:2191-2212); what = 1: write erase flags
__global__ void __launch_bounds__(256) k_lba_classify(LbaDev D, int what, unsigned char* erase) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= D.n_obs) return;
  const vieo_lba_obs o = D.obs[i];
  const double info = (double)o.inv_sigma2;
  const double* e = D.err + 3 * (size_t)i;
  double chi2 = e[0] * (info * e[0]) + e[1] * (info * e[1]);
  if (o.ur >= 0) chi2 += e[2] * (info * e[2]);
  PoseXf X;
  kf_xf(D.cam, D.kf[o.kf], X);
  const double* Xw = D.X + 3 * (size_t)o.mp;
  const double z = X.Rcw[6] * Xw[0] + X.Rcw[7] * Xw[1] + X.Rcw[8] * Xw[2] + X.tcw[2];
  const bool bad = chi2 > (o.ur >= 0 ? 7.815 : 5.991) || !(z > 0.);
  if (what == 0) {
    if (bad) D.level[i] = 1;
  } else
    erase[i] = bad ? 1 : 0;
}

// ---- landmark-parallel linearisation: 16 lanes per landmark
__global__ void __launch_bounds__(256) k_lba_linearize(LbaDev D) {
  const int sub = threadIdx.x & 15;
  const int m = blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool valid_m = m < D.n_mp && D.mp_act[m];
  double acc[9];  // Hll upper (6) + bl (3)
#pragma unroll
  for (int i = 0; i < 9; i++) acc[i] = 0;
  if (valid_m) {
    const int first = D.mp_first[m], cnt = D.mp_count[m];
    const double* Xw = D.X + 3 * (size_t)m;
    for (int j = sub; j < cnt; j += 16) {
      const int i = first + j;
      if (D.level[i]) continue;
      const vieo_lba_obs o = D.obs[i];
      const LbaKf k = D.kf[o.kf];
      PoseXf X;
      kf_xf(D.cam, k, X);
      double err[3], Pc[3];
      const double chi2 = lba_edge_error(D.cam, X, o, Xw, err, Pc);
      const bool stereo = o.ur >= 0;
      double r0, r1 = 1.;
      if (D.robust) {
        const double dl = stereo ? D.dStereo : D.dMono;
        huber(chi2, dl, dl * dl, &r0, &r1);
      }
      double Jp[18], Jx[9];
      lba_jacobians(D.cam, X, k.p, Xw, Pc, Jp, Jx);
      const double info = (double)o.inv_sigma2, w = r1 * info;
      const int de = stereo ? 3 : 2;
      int t = 0;
      for (int a = 0; a < 3; a++) {
        for (int b = a; b < 3; b++, t++) {
          double s = 0;
          for (int r = 0; r < de; r++) s += Jx[r * 3 + a] * w * Jx[r * 3 + b];
          acc[t] += s;
        }
        double s = 0;
        for (int r = 0; r < de; r++) s += Jx[r * 3 + a] * (-(info * err[r]) * r1);
        acc[6 + a] += s;
      }
      double* B = D.Bpl + 18 * (size_t)i;
      if (k.col >= 0)
        for (int a = 0; a < 6; a++)
          for (int b = 0; b < 3; b++) {
            double s = 0;
            for (int r = 0; r < de; r++) s += Jp[r * 6 + a] * w * Jx[r * 3 + b];
            B[a * 3 + b] = s;
          }
    }
  }
#pragma unroll
  for (int i = 0; i < 9; i++)
    for (int o = 8; o > 0; o >>= 1) acc[i] += __shfl_xor(acc[i], o, 16);
  if (valid_m && sub == 0) {
    double* H = D.Hll + 9 * (size_t)m;
    H[0] = acc[0], H[1] = acc[1], H[2] = acc[2];
    H[3] = acc[1], H[4] = acc[3], H[5] = acc[4];
    H[6] = acc[2], H[7] = acc[4], H[8] = acc[5];
    D.bl[3 * (size_t)m] = acc[6], D.bl[3 * (size_t)m + 1] = acc[7], D.bl[3 * (size_t)m + 2] = acc[8];
  }
}

// ---- key-frame-parallel Hpp / bp.  grid (chunks, n_free); edge lists sorted by key frame
__global__ void __launch_bounds__(256)
k_lba_pose(LbaDev D, const int* __restrict__ kf_list, const int* __restrict__ kf_edge_first,
           const int* __restrict__ kf_edge_idx) {
  __shared__ double s_red[4 * 27];
  const int kfi = kf_list[blockIdx.y];
  const LbaKf k = D.kf[kfi];
  const int first = kf_edge_first[kfi], cnt = kf_edge_first[kfi + 1] - first;
  double acc[27];
#pragma unroll
  for (int i = 0; i < 27; i++) acc[i] = 0;
  PoseXf X;
  kf_xf(D.cam, k, X);
  for (int j = blockIdx.x * 256 + threadIdx.x; j < cnt; j += gridDim.x * 256) {
    const int i = kf_edge_idx[first + j];
    if (D.level[i]) continue;
    const vieo_lba_obs o = D.obs[i];
    const double* Xw = D.X + 3 * (size_t)o.mp;
    double err[3], Pc[3];
    const double chi2 = lba_edge_error(D.cam, X, o, Xw, err, Pc);
    const bool stereo = o.ur >= 0;
    double r0, r1 = 1.;
    if (D.robust) {
      const double dl = stereo ? D.dStereo : D.dMono;
      huber(chi2, dl, dl * dl, &r0, &r1);
    }
    double Jp[18], Jx[9];
    lba_jacobians(D.cam, X, k.p, Xw, Pc, Jp, Jx);
    visual_accumulate(Jp, err, (double)o.inv_sigma2, r1, stereo, acc);
  }
  block_sum<27>(acc, s_red, threadIdx.x);
  if (threadIdx.x < 27) {
    const int c = k.col, np = D.np;
    if (threadIdx.x < 21) {
      int a = 0, t = threadIdx.x;
      while (t >= 6 - a) t -= 6 - a, a++;
      const int b = a + t;
      atomicAdd(&D.Hpp[(size_t)(c + a) * np + c + b], acc[threadIdx.x]);
      if (a != b) atomicAdd(&D.Hpp[(size_t)(c + b) * np + c + a], acc[threadIdx.x]);
    } else
      atomicAdd(&D.bp[c + threadIdx.x - 21], acc[threadIdx.x]);
  }
}

// ---- Schur complement: one wavefront per landmark, FP64 MFMA for the block contraction.
typedef double double4_t __attribute__((ext_vector_type(4)));
static const int kSchurMaxObs = 32;  // free observers of one landmark handled by the MFMA tiling

// use_lds = 0: windows whose reduced system does not fit LDS accumulate straight into global memory.
__global__ void __launch_bounds__(256)
k_lba_schur(LbaDev D, double lambda, int lds_np, int* overflow, int use_lds) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int np = D.np;
  double* sH = use_lds ? smem : D.Hs;              // [ld][ld] accumulated -(B D^-1 B^T)
  double* sb = use_lds ? smem + (size_t)lds_np * lds_np : D.bs;
  double* stage = use_lds ? smem + (size_t)lds_np * lds_np + lds_np : smem;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (!use_lds) lds_np = np;
  if (use_lds)
    for (int i = threadIdx.x; i < lds_np * lds_np + lds_np; i += 256) sH[i] = 0;
  __syncthreads();
  double* sA = stage + (size_t)wave * (192 * 4 * 2 + 192);
  double* sB = sA + 192 * 4;
  int* sC = (int*)(sB + 192 * 4);
  for (int m = blockIdx.x * 4 + wave; m < D.n_mp; m += gridDim.x * 4) {
    if (!D.mp_act[m]) continue;
    // D^-1 = (Hll + lambda I)^-1 (every lane redundantly)
    const double* H = D.Hll + 9 * (size_t)m;
    const double a00 = H[0] + lambda, a01 = H[1], a02 = H[2], a10 = H[3], a11 = H[4] + lambda, a12 = H[5],
                 a20 = H[6], a21 = H[7], a22 = H[8] + lambda;
    const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
    const double id = 1.0 / (a00 * c00 + a01 * c01 + a02 * c02);
    double Di[9];
    Di[0] = c00 * id, Di[1] = (a02 * a21 - a01 * a22) * id, Di[2] = (a01 * a12 - a02 * a11) * id;
    Di[3] = c01 * id, Di[4] = (a00 * a22 - a02 * a20) * id, Di[5] = (a02 * a10 - a00 * a12) * id;
    Di[6] = c02 * id, Di[7] = (a01 * a20 - a00 * a21) * id, Di[8] = (a00 * a11 - a01 * a10) * id;
    if (lane < 9) D.Dinv[9 * (size_t)m + lane] = Di[lane];
    const double* blm = D.bl + 3 * (size_t)m;
    const double db0 = Di[0] * blm[0] + Di[1] * blm[1] + Di[2] * blm[2];
    const double db1 = Di[3] * blm[0] + Di[4] * blm[1] + Di[5] * blm[2];
    const double db2 = Di[6] * blm[0] + Di[7] * blm[1] + Di[8] * blm[2];
    // gather the free, active observers of this landmark (ordered compaction)
    const int first = D.mp_first[m], cnt = D.mp_count[m];
    int k = 0;
    for (int j0 = 0; j0 < cnt; j0 += 64) {
      const int j = j0 + lane;
      bool use = false;
      int col = -1;
      if (j < cnt) {
        const int i = first + j;
        col = D.kf[D.obs[i].kf].col;
        use = D.level[i] == 0 && col >= 0;
      }
      const unsigned long long bal = __ballot(use);
      if (use) {
        const int pos = k + __popcll(bal & ((1ull << lane) - 1ull));
        if (pos < kSchurMaxObs) {
          const double* B = D.Bpl + 18 * (size_t)(first + j);
          for (int a = 0; a < 6; a++) {
            const double b0 = B[a * 3], b1 = B[a * 3 + 1], b2 = B[a * 3 + 2];
            const int row = pos * 6 + a;
            sB[row * 4 + 0] = b0, sB[row * 4 + 1] = b1, sB[row * 4 + 2] = b2, sB[row * 4 + 3] = 0;
            sA[row * 4 + 0] = b0 * Di[0] + b1 * Di[3] + b2 * Di[6];
            sA[row * 4 + 1] = b0 * Di[1] + b1 * Di[4] + b2 * Di[7];
            sA[row * 4 + 2] = b0 * Di[2] + b1 * Di[5] + b2 * Di[8];
            sA[row * 4 + 3] = 0;
            sC[row] = col + a;
            // bschur -= B * (D^-1 bl)
            atomicAdd(&sb[col + a], -(b0 * db0 + b1 * db1 + b2 * db2));
          }
        }
      }
      k += __popcll(bal);
    }
    if (k > kSchurMaxObs) {
      if (lane == 0) atomicExch(overflow, 1);
      k = kSchurMaxObs;
    }
    const int rows = 6 * k, nt = (rows + 15) >> 4;
    // zero-pad the last tile
    for (int r = rows + lane; r < nt * 16; r += 64) {
      sA[r * 4] = sA[r * 4 + 1] = sA[r * 4 + 2] = sA[r * 4 + 3] = 0;
      sB[r * 4] = sB[r * 4 + 1] = sB[r * 4 + 2] = sB[r * 4 + 3] = 0;
      sC[r] = -1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // C(ti, tj) = A_ti (16x4) * B_tj^T (4x16) on the matrix core; scatter -C into the LDS system
    for (int ti = 0; ti < nt; ti++)
      for (int tj = 0; tj < nt; tj++) {
        const double av = sA[(ti * 16 + (lane & 15)) * 4 + (lane >> 4)];
        const double bv = sB[(tj * 16 + (lane & 15)) * 4 + (lane >> 4)];
        double4_t c = {0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c, 0, 0, 0);
        const int gc = sC[tj * 16 + (lane & 15)];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int gr = sC[ti * 16 + (lane >> 4) + 4 * r];  // f64 C/D map: row = (lane>>4) + 4*reg
          if (gr >= 0 && gc >= 0) atomicAdd(&sH[gr * lds_np + gc], -c[r]);
        }
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  if (!use_lds) return;
  for (int i = threadIdx.x; i < np * np; i += 256) {
    const double v = sH[(i / np) * lds_np + (i % np)];
    if (v != 0.0) atomicAdd(&D.Hs[i], v);
  }
  for (int i = threadIdx.x; i < np; i += 256)
    if (sb[i] != 0.0) atomicAdd(&D.bs[i], sb[i]);
}

// Hs = Hpp + lambda I ; bs = bp
__global__ void __launch_bounds__(256) k_lba_init_reduced(LbaDev D, double lambda) {
  const int i = blockIdx.x * 256 + threadIdx.x, np = D.np;
  if (i < np * np) D.Hs[i] = D.Hpp[i] + ((i / np) == (i % np) ? lambda : 0.0);
  if (i < np) D.bs[i] = D.bp[i];
}

// ---- dense LDL^T solve of the reduced system by one workgroup (matrix in global memory / L2)
__global__ void __launch_bounds__(256) k_lba_ldlt(double* A, const double* b, double* x, int n, int* ok_out) {
  __shared__ double s_col[512];
  __shared__ double s_D[512];
  __shared__ int s_ok;
  const int tid = threadIdx.x;
  if (tid == 0) s_ok = 1;
  __syncthreads();
  for (int j = 0; j < n; j++) {
    const double d = A[(size_t)j * n + j];
    if (!(d > 0)) {
      if (tid == 0) s_ok = 0;
      break;
    }
    for (int i = j + 1 + tid; i < n; i += 256) s_col[i] = A[(size_t)i * n + j];
    if (tid == 0) s_D[j] = d;
    __syncthreads();
    const int m = n - j - 1;
    for (int e = tid; e < m * m; e += 256) {
      const int i = j + 1 + e / m, k = j + 1 + e % m;
      A[(size_t)i * n + k] -= (s_col[i] / d) * s_col[k];
    }
    for (int i = j + 1 + tid; i < n; i += 256) A[(size_t)i * n + j] = s_col[i] / d;
    __syncthreads();
  }
  __syncthreads();
  if (!s_ok) {
    for (int i = tid; i < n; i += 256) x[i] = 0;
    if (tid == 0) *ok_out = 0;
    return;
  }
  double* y = s_col;
  for (int i = tid; i < n; i += 256) y[i] = b[i];
  __syncthreads();
  for (int j = 0; j < n; j++) {
    const double yj = y[j];
    for (int i = j + 1 + tid; i < n; i += 256) y[i] -= A[(size_t)i * n + j] * yj;
    __syncthreads();
  }
  for (int i = tid; i < n; i += 256) y[i] /= s_D[i];
  __syncthreads();
  for (int j = n - 1; j >= 0; j--) {
    const double xj = y[j];
    for (int i = tid; i < j; i += 256) y[i] -= A[(size_t)j * n + i] * xj;
    __syncthreads();
  }
  for (int i = tid; i < n; i += 256) x[i] = y[i];
  if (tid == 0) *ok_out = 1;
}

// ---- back-substitution + update of the points, scale terms of the LM gain ratio
__global__ void __launch_bounds__(256) k_lba_update_points(LbaDev D, double lambda) {
  __shared__ double s_red[4];
  const int m = blockIdx.x * 256 + threadIdx.x;
  double sc[1] = {0};
  if (m < D.n_mp && D.mp_act[m]) {
    const int first = D.mp_first[m], cnt = D.mp_count[m];
    double cl[3] = {D.bl[3 * (size_t)m], D.bl[3 * (size_t)m + 1], D.bl[3 * (size_t)m + 2]};
    for (int j = 0; j < cnt; j++) {
      const int i = first + j;
      if (D.level[i]) continue;
      const int col = D.kf[D.obs[i].kf].col;
      if (col < 0) continue;
      const double* B = D.Bpl + 18 * (size_t)i;
      for (int a = 0; a < 6; a++) {
        const double xa = D.xp[col + a];
        cl[0] -= B[a * 3] * xa, cl[1] -= B[a * 3 + 1] * xa, cl[2] -= B[a * 3 + 2] * xa;
      }
    }
    const double* Di = D.Dinv + 9 * (size_t)m;
    for (int a = 0; a < 3; a++) {
      const double x = Di[a * 3] * cl[0] + Di[a * 3 + 1] * cl[1] + Di[a * 3 + 2] * cl[2];
      D.xl[3 * (size_t)m + a] = x;
      D.X[3 * (size_t)m + a] += x;
      sc[0] += x * (lambda * x + D.bl[3 * (size_t)m + a]);
    }
  }
  block_sum<1>(sc, s_red, threadIdx.x);
  if (threadIdx.x == 0) D.part[blockIdx.x] = sc[0];
}

__global__ void k_lba_update_poses(LbaDev D) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= D.n_kf) return;
  LbaKf kf = D.kf[k];
  if (kf.col < 0) return;
  Est e;
  e.p[0] = kf.p[0], e.p[1] = kf.p[1], e.p[2] = kf.p[2];
  e.qw = kf.qw, e.qx = kf.qx, e.qy = kf.qy, e.qz = kf.qz;
  inc_small_pr(e, D.xp + kf.col);
  kf.p[0] = e.p[0], kf.p[1] = e.p[1], kf.p[2] = e.p[2];
  kf.qw = e.qw, kf.qx = e.qx, kf.qy = e.qy, kf.qz = e.qz;
  D.kf[k] = kf;
}

// ================================================================== host-side LM driver
struct LbaHost {
  DevBuf obs, kf, kf_bak, X, X_bak, err, level, mp_first, mp_count, mp_act, Bpl, Hll, bl, Hpp, bp, Hs, bs,
      Dinv, xp, xl, part, kf_list, kf_edge_first, kf_edge_idx, flags, erase;
};
static thread_local LbaHost g_lba;
static thread_local hipStream_t g_lba_stream = nullptr;

// synchronous copy on the calling thread's own stream (so concurrent LBA calls from several host
// threads, and the frame pipelines on their streams, do not serialise on the null stream)
static inline hipError_t lba_copy(void* dst, const void* src, size_t n, hipMemcpyKind kind) {
  hipError_t e = hipMemcpyAsync(dst, src, n, kind, g_lba_stream);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(g_lba_stream);
}

#define LBA_ENS(b, n) \
  if ((rc = (b).ensure(std::max<size_t>((n), 8))) != VIEO_OK) return rc

static int sum_partials(LbaHost& S, int n, double* out) {
  std::vector<double> h(n);
  VIEO_HIP_CHECK(lba_copy(h.data(), S.part.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  double s = 0;
  for (double v : h) s += v;
  *out = s;
  return VIEO_OK;
}

}  // namespace vieo

using namespace vieo;

extern "C" int vieo_local_bundle_adjustment(const vieo_lba_params* P, const vieo_lba_keyframe* h_kfs,
                                            int n_kf, const float* h_points, int n_mp,
                                            const vieo_lba_obs* h_obs, int n_obs,
                                            volatile const int* stop, vieo_navstate* h_navs_out,
                                            float* h_points_out, uint8_t* h_erase,
                                            vieo_lba_result* R) {
  if (!P || !h_kfs || n_kf <= 0 || !h_points || n_mp <= 0 || !h_obs || n_obs <= 0 || !h_navs_out ||
      !h_points_out || !h_erase || !R)
    return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  if (!g_lba_stream) VIEO_HIP_CHECK(hipStreamCreateWithFlags(&g_lba_stream, hipStreamNonBlocking));
  memset(R, 0, sizeof(*R));
  for (int k = 0; k < n_kf; k++) h_navs_out[k] = h_kfs[k].nav;
  memcpy(h_points_out, h_points, (size_t)n_mp * 12);
  memset(h_erase, 0, n_obs);
  bool any_free = false;
  for (int k = 0; k < n_kf; k++) any_free |= !h_kfs[k].fixed;
  if (!any_free) {
    R->status = VIEO_LBA_NO_FREE_POSE;  // Optimizer.cc:1993
    return VIEO_OK;
  }
  // ---- index structures (observations must be grouped by map point)
  std::vector<int> mp_first(n_mp, 0), mp_count(n_mp, 0), kf_cnt(n_kf + 1, 0);
  for (int i = 0; i < n_obs; i++) {
    const int m = h_obs[i].mp, k = h_obs[i].kf;
    if (m < 0 || m >= n_mp || k < 0 || k >= n_kf || (i > 0 && m < h_obs[i - 1].mp)) {
      set_error("vieo_local_bundle_adjustment: observations must be sorted by map point");
      return VIEO_E_INVALID;
    }
    if (mp_count[m] == 0) mp_first[m] = i;
    mp_count[m]++;
    kf_cnt[k + 1]++;
  }
  std::vector<int> kf_edge_first(n_kf + 1, 0), kf_edge_idx(n_obs), fill(n_kf, 0);
  for (int k = 0; k < n_kf; k++) kf_edge_first[k + 1] = kf_edge_first[k] + kf_cnt[k + 1];
  for (int i = 0; i < n_obs; i++) kf_edge_idx[kf_edge_first[h_obs[i].kf] + fill[h_obs[i].kf]++] = i;
  std::vector<LbaKf> kf(n_kf);
  for (int k = 0; k < n_kf; k++) {
    memcpy(kf[k].p, h_kfs[k].nav.p, 24);
    kf[k].qw = h_kfs[k].nav.q[0], kf[k].qx = h_kfs[k].nav.q[1], kf[k].qy = h_kfs[k].nav.q[2],
    kf[k].qz = h_kfs[k].nav.q[3];
    kf[k].col = -1, kf[k].pad = 0;
  }
  std::vector<double> X((size_t)n_mp * 3);
  for (int i = 0; i < n_mp * 3; i++) X[i] = (double)h_points[i];
  LbaHost& S = g_lba;
  const int nblk_e = (n_obs + 255) / 256, nblk_m = (n_mp + 255) / 256;
  const int np_max = 6 * n_kf;
  if (np_max > 512) {
    set_error("local BA: more than 85 key frames");
    return VIEO_E_CAPACITY;
  }
  LBA_ENS(S.obs, (size_t)n_obs * sizeof(vieo_lba_obs));
  LBA_ENS(S.kf, (size_t)n_kf * sizeof(LbaKf));
  LBA_ENS(S.kf_bak, (size_t)n_kf * sizeof(LbaKf));
  LBA_ENS(S.X, (size_t)n_mp * 24);
  LBA_ENS(S.X_bak, (size_t)n_mp * 24);
  LBA_ENS(S.err, (size_t)n_obs * 24);
  LBA_ENS(S.level, (size_t)n_obs);
  LBA_ENS(S.erase, (size_t)n_obs);
  LBA_ENS(S.mp_first, (size_t)n_mp * 4);
  LBA_ENS(S.mp_count, (size_t)n_mp * 4);
  LBA_ENS(S.mp_act, (size_t)n_mp);
  LBA_ENS(S.Bpl, (size_t)n_obs * 18 * 8);
  LBA_ENS(S.Hll, (size_t)n_mp * 72);
  LBA_ENS(S.bl, (size_t)n_mp * 24);
  LBA_ENS(S.Dinv, (size_t)n_mp * 72);
  LBA_ENS(S.xl, (size_t)n_mp * 24);
  LBA_ENS(S.Hpp, (size_t)np_max * np_max * 8);
  LBA_ENS(S.Hs, (size_t)np_max * np_max * 8);
  LBA_ENS(S.bp, (size_t)np_max * 8);
  LBA_ENS(S.bs, (size_t)np_max * 8);
  LBA_ENS(S.xp, (size_t)np_max * 8);
  LBA_ENS(S.part, (size_t)std::max(nblk_e, nblk_m) * 8);
  LBA_ENS(S.kf_list, (size_t)n_kf * 4);
  LBA_ENS(S.kf_edge_first, (size_t)(n_kf + 1) * 4);
  LBA_ENS(S.kf_edge_idx, (size_t)n_obs * 4);
  LBA_ENS(S.flags, 16);
  VIEO_HIP_CHECK(lba_copy(S.obs.p, h_obs, (size_t)n_obs * sizeof(vieo_lba_obs), hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(lba_copy(S.X.p, X.data(), X.size() * 8, hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(lba_copy(S.mp_first.p, mp_first.data(), (size_t)n_mp * 4, hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(lba_copy(S.mp_count.p, mp_count.data(), (size_t)n_mp * 4, hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(lba_copy(S.kf_edge_first.p, kf_edge_first.data(), (size_t)(n_kf + 1) * 4, hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(lba_copy(S.kf_edge_idx.p, kf_edge_idx.data(), (size_t)n_obs * 4, hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(hipMemsetAsync(S.level.p, 0, n_obs, g_lba_stream));
  VIEO_HIP_CHECK(hipMemsetAsync(S.err.p, 0, (size_t)n_obs * 24, g_lba_stream));
  LbaDev D;
  D.obs = S.obs.as<vieo_lba_obs>();
  D.n_obs = n_obs, D.n_mp = n_mp, D.n_kf = n_kf, D.np = 0;
  D.kf = S.kf.as<LbaKf>(), D.X = S.X.as<double>(), D.err = S.err.as<double>();
  D.level = S.level.as<unsigned char>();
  D.mp_first = S.mp_first.as<int>(), D.mp_count = S.mp_count.as<int>();
  D.mp_act = S.mp_act.as<unsigned char>();
  D.Bpl = S.Bpl.as<double>(), D.Hll = S.Hll.as<double>(), D.bl = S.bl.as<double>();
  D.Hpp = S.Hpp.as<double>(), D.bp = S.bp.as<double>(), D.Hs = S.Hs.as<double>(), D.bs = S.bs.as<double>();
  D.Dinv = S.Dinv.as<double>(), D.xp = S.xp.as<double>(), D.xl = S.xl.as<double>();
  D.part = S.part.as<double>();
  D.cam.fx = P->fx, D.cam.fy = P->fy, D.cam.cx = P->cx, D.cam.cy = P->cy, D.cam.bf = P->bf;
  memcpy(D.cam.Rcb, P->Rcb, 72);
  memcpy(D.cam.tcb, P->tcb, 24);
  D.robust = 1;
  D.dMono = (double)(float)sqrt(5.991), D.dStereo = (double)(float)sqrt(7.815);
  hipStream_t st = g_lba_stream;
  std::vector<unsigned char> level(n_obs, 0), mp_act(n_mp);
  std::vector<int> kf_list;

  auto robust_chi2 = [&](double* out) -> int {
    hipLaunchKernelGGL(k_lba_error, dim3(nblk_e), dim3(256), 0, st, D);
    return sum_partials(S, nblk_e, out);
  };
  // one SparseOptimizer::optimize(iterations)
  auto optimize = [&](int iterations, bool first) -> int {
    // initializeOptimization(0): active edges -> active vertices -> reduced-system columns
    std::vector<char> kf_act(n_kf, 0);
    std::fill(mp_act.begin(), mp_act.end(), 0);
    bool any = false;
    for (int i = 0; i < n_obs; i++)
      if (!level[i]) kf_act[h_obs[i].kf] = 1, mp_act[h_obs[i].mp] = 1, any = true;
    int np = 0;
    kf_list.clear();
    for (int k = 0; k < n_kf; k++) {
      if (!h_kfs[k].fixed && kf_act[k]) {
        kf[k].col = np, np += 6;
        kf_list.push_back(k);
      } else
        kf[k].col = -1;
    }
    if (!any || np == 0) return VIEO_OK;
    // keep the device poses, refresh only the column map
    std::vector<LbaKf> cur(n_kf);
    if (!first) {
      VIEO_HIP_CHECK(lba_copy(cur.data(), S.kf.p, (size_t)n_kf * sizeof(LbaKf), hipMemcpyDeviceToHost));
      for (int k = 0; k < n_kf; k++) cur[k].col = kf[k].col;
    } else
      cur = kf;
    VIEO_HIP_CHECK(lba_copy(S.kf.p, cur.data(), (size_t)n_kf * sizeof(LbaKf), hipMemcpyHostToDevice));
    VIEO_HIP_CHECK(lba_copy(S.mp_act.p, mp_act.data(), n_mp, hipMemcpyHostToDevice));
    VIEO_HIP_CHECK(lba_copy(S.kf_list.p, kf_list.data(), kf_list.size() * 4, hipMemcpyHostToDevice));
    D.np = np;
    const int lds_np = np | 1;  // odd leading dimension: spreads the LDS atomics over banks
    const size_t stage_lds = 4 * (192 * 4 * 2 + 192) * 8 + 64;
    size_t schur_lds = ((size_t)lds_np * lds_np + lds_np) * 8 + stage_lds;
    const int use_lds = schur_lds <= 150 * 1024;
    if (!use_lds) schur_lds = stage_lds;
    VIEO_HIP_CHECK(hipFuncSetAttribute((const void*)k_lba_schur, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)schur_lds));
    const int schur_blocks = std::max(1, std::min(256, (n_mp + 3) / 4));
    double lambda = -1, ni = 2;
    int nBad = 0;
    for (int it = 0; it < iterations; it++) {
      if (stop && *stop) break;
      R->lm_iterations++;
      double currentChi;
      if ((rc = robust_chi2(&currentChi)) != VIEO_OK) return rc;
      if (first && it == 0) R->chi2_initial = currentChi;
      const double iniChi = currentChi;
      // ---- buildSystem
      VIEO_HIP_CHECK(hipMemsetAsync(S.Hpp.p, 0, (size_t)np * np * 8, st));
      VIEO_HIP_CHECK(hipMemsetAsync(S.bp.p, 0, (size_t)np * 8, st));
      hipLaunchKernelGGL(k_lba_linearize, dim3((n_mp + 15) / 16), dim3(256), 0, st, D);
      hipLaunchKernelGGL(k_lba_pose, dim3(8, (unsigned)kf_list.size()), dim3(256), 0, st, D,
                         S.kf_list.as<int>(), S.kf_edge_first.as<int>(), S.kf_edge_idx.as<int>());
      if (it == 0) {  // computeLambdaInit: tau * max diagonal over poses and landmarks
        std::vector<double> Hpp((size_t)np * np), Hll((size_t)n_mp * 9);
        VIEO_HIP_CHECK(lba_copy(Hpp.data(), S.Hpp.p, Hpp.size() * 8, hipMemcpyDeviceToHost));
        VIEO_HIP_CHECK(lba_copy(Hll.data(), S.Hll.p, Hll.size() * 8, hipMemcpyDeviceToHost));
        double mx = 0;
        for (int j = 0; j < np; j++) mx = std::max(std::fabs(Hpp[(size_t)j * np + j]), mx);
        for (int m = 0; m < n_mp; m++)
          if (mp_act[m])
            for (int a = 0; a < 3; a++) mx = std::max(std::fabs(Hll[(size_t)m * 9 + a * 4]), mx);
        lambda = 1e-5 * mx;
        ni = 2;
        nBad = 0;
      }
      std::vector<double> bp(np);
      VIEO_HIP_CHECK(lba_copy(bp.data(), S.bp.p, (size_t)np * 8, hipMemcpyDeviceToHost));
      double rho = 0;
      int qmax = 0;
      do {
        R->lm_trials++;
        VIEO_HIP_CHECK(hipMemcpyAsync(S.kf_bak.p, S.kf.p, (size_t)n_kf * sizeof(LbaKf), hipMemcpyDeviceToDevice, st));
        VIEO_HIP_CHECK(hipMemcpyAsync(S.X_bak.p, S.X.p, (size_t)n_mp * 24, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_lba_init_reduced, dim3((np * np + 255) / 256), dim3(256), 0, st, D, lambda);
        VIEO_HIP_CHECK(hipMemsetAsync(S.flags.p, 0, 16, st));
        hipLaunchKernelGGL(k_lba_schur, dim3(schur_blocks), dim3(256), schur_lds, st, D, lambda, lds_np,
                           S.flags.as<int>() + 1, use_lds);
        hipLaunchKernelGGL(k_lba_ldlt, dim3(1), dim3(256), 0, st, D.Hs, D.bs, D.xp, np, S.flags.as<int>());
        hipLaunchKernelGGL(k_lba_update_points, dim3(nblk_m), dim3(256), 0, st, D, lambda);
        hipLaunchKernelGGL(k_lba_update_poses, dim3((n_kf + 63) / 64), dim3(64), 0, st, D);
        double scale_l;
        if ((rc = sum_partials(S, nblk_m, &scale_l)) != VIEO_OK) return rc;
        int flags[2];
        std::vector<double> xp(np);
        VIEO_HIP_CHECK(lba_copy(flags, S.flags.p, 8, hipMemcpyDeviceToHost));
        VIEO_HIP_CHECK(lba_copy(xp.data(), S.xp.p, (size_t)np * 8, hipMemcpyDeviceToHost));
        if (flags[1]) {
          set_error("local BA: a map point has more than %d free observers", kSchurMaxObs);
          return VIEO_E_CAPACITY;
        }
        const bool ok2 = flags[0] != 0;
        double tempChi;
        if ((rc = robust_chi2(&tempChi)) != VIEO_OK) return rc;
        if (!ok2) tempChi = DBL_MAX;
        rho = currentChi - tempChi;
        double scale = ok2 ? scale_l : 0.0;
        for (int j = 0; j < np; j++) scale += xp[j] * (lambda * xp[j] + bp[j]);
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && std::isfinite(tempChi)) {
          double alpha = 1. - std::pow(2 * rho - 1, 3);
          alpha = std::min(alpha, 2. / 3.);
          lambda *= std::max(1. / 3., alpha);
          ni = 2;
          currentChi = tempChi;
        } else {
          lambda *= ni;
          ni *= 2;
          VIEO_HIP_CHECK(hipMemcpyAsync(S.kf.p, S.kf_bak.p, (size_t)n_kf * sizeof(LbaKf), hipMemcpyDeviceToDevice, st));
          VIEO_HIP_CHECK(hipMemcpyAsync(S.X.p, S.X_bak.p, (size_t)n_mp * 24, hipMemcpyDeviceToDevice, st));
        }
        qmax++;
      } while (rho < 0 && qmax < 10 && !(stop && *stop));
      R->chi2_final = currentChi;
      if (qmax == 10 || rho == 0) break;
      if ((iniChi - currentChi) * 1e3 < iniChi)
        nBad++;
      else
        nBad = 0;
      if (nBad >= 3) break;
    }
    return VIEO_OK;
  };

  if (stop && *stop) {
    R->status = VIEO_LBA_ABORTED;
    return VIEO_OK;
  }
  if ((rc = optimize(P->its0, true)) != VIEO_OK) return rc;
  if (!(stop && *stop)) {
    hipLaunchKernelGGL(k_lba_classify, dim3(nblk_e), dim3(256), 0, st, D, 0, S.erase.as<unsigned char>());
    VIEO_HIP_CHECK(lba_copy(level.data(), S.level.p, n_obs, hipMemcpyDeviceToHost));
    D.robust = 0;
    if ((rc = optimize(P->its1, false)) != VIEO_OK) return rc;
  } else
    R->status = VIEO_LBA_ABORTED;
  hipLaunchKernelGGL(k_lba_classify, dim3(nblk_e), dim3(256), 0, st, D, 1, S.erase.as<unsigned char>());
  VIEO_HIP_CHECK(lba_copy(h_erase, S.erase.p, n_obs, hipMemcpyDeviceToHost));
  for (int i = 0; i < n_obs; i++) R->n_erase += h_erase[i];
  std::vector<LbaKf> out(n_kf);
  VIEO_HIP_CHECK(lba_copy(out.data(), S.kf.p, (size_t)n_kf * sizeof(LbaKf), hipMemcpyDeviceToHost));
  VIEO_HIP_CHECK(lba_copy(X.data(), S.X.p, X.size() * 8, hipMemcpyDeviceToHost));
  for (int k = 0; k < n_kf; k++) {
    if (h_kfs[k].fixed) continue;
    memcpy(h_navs_out[k].p, out[k].p, 24);
    h_navs_out[k].q[0] = out[k].qw, h_navs_out[k].q[1] = out[k].qx, h_navs_out[k].q[2] = out[k].qy,
    h_navs_out[k].q[3] = out[k].qz;
  }
  for (int i = 0; i < n_mp * 3; i++) h_points_out[i] = (float)X[i];  // SetWorldPos(cast<float>)
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}
