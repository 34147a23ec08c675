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

// control word of a window for one round of the lock-step batch driver
enum {
  LBA_TRIAL = 1,    // solve + update + evaluate one lambda trial
  LBA_BUILD = 2,    // re-linearise (start of an LM iteration)
  LBA_RESTORE = 4,  // last trial was rejected: restore the backed-up estimates first
  LBA_ERROR = 8,    // residual pass only (start of an optimize())
  LBA_CLASS0 = 16,  // chi2 / depth gates -> level 1 (between the two optimisations)
  LBA_CLASS1 = 32,  // final erase flags
};
struct WinCtl {
  int flags, pad;
  double lambda;
};
struct WinOut {
  double chi2, scale_l, scale_p, maxdiag;
  int ok, overflow;
};

struct LbaDev {
  const vieo_lba_obs* obs;
  int n_obs, n_mp, n_kf, np;
  int n_free, pad0;
  const int *kf_list, *kf_edge_first, *kf_edge_idx;
  LbaKf* kf_bak;
  double* X_bak;
  double* part_m;          // per-block partial sums of the point pass
  unsigned char* erase;
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

// Every kernel below is launched for ALL windows of a batch (blockIdx.y / .z / .x = window) and
// returns at once for windows whose control word does not ask for that step.

// ---- residuals + robust chi2 of the active edges
__global__ void __launch_bounds__(256)
k_lba_error(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl) {
  __shared__ double s_red[4];
  const int w = blockIdx.y;
  if (!(ctl[w].flags & (LBA_TRIAL | LBA_ERROR))) return;
  const LbaDev& D = devs[w];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= D.n_obs) return;
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

// one workgroup per window: fold the per-block partials into the window's output record
__global__ void __launch_bounds__(256)
k_lba_reduce(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, WinOut* __restrict__ out) {
  __shared__ double s_red[4 * 2];
  const int w = blockIdx.x, fl = ctl[w].flags;
  if (!(fl & (LBA_TRIAL | LBA_ERROR))) return;
  const LbaDev& D = devs[w];
  double v[2] = {0, 0};
  for (int i = threadIdx.x; i < (D.n_obs + 255) / 256; i += 256) v[0] += D.part[i];
  if (fl & LBA_TRIAL)
    for (int i = threadIdx.x; i < (D.n_mp + 255) / 256; i += 256) v[1] += D.part_m[i];
  block_sum<2>(v, s_red, threadIdx.x);
  if (threadIdx.x == 0) out[w].chi2 = v[0], out[w].scale_l = v[1];
}

// chi2 (from the STORED error, as the reference does) / depth classification
// (Optimizer.cc:2191-2212 -> level 1; :2227-2249 -> vToErase)
__global__ void __launch_bounds__(256)
k_lba_classify(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl) {
  const int w = blockIdx.y, fl = ctl[w].flags;
  if (!(fl & (LBA_CLASS0 | LBA_CLASS1))) return;
  const LbaDev& D = devs[w];
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
  if (fl & LBA_CLASS0) {
    if (bad) D.level[i] = 1;
  } else
    D.erase[i] = bad ? 1 : 0;
}

// ---- backup (before a trial) / restore (after a rejected trial) of poses and points
__global__ void __launch_bounds__(256)
k_lba_backup_restore(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, int restore) {
  const int w = blockIdx.y;
  if (!(ctl[w].flags & (restore ? LBA_RESTORE : LBA_TRIAL))) return;
  const LbaDev& D = devs[w];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < D.n_mp * 3) {
    if (restore)
      D.X[i] = D.X_bak[i];
    else
      D.X_bak[i] = D.X[i];
  }
  if (i < D.n_kf) {
    if (restore) {
      const int col = D.kf[i].col;
      D.kf[i] = D.kf_bak[i];
      D.kf[i].col = col;
    } else
      D.kf_bak[i] = D.kf[i];
  }
}

// ---- landmark-parallel linearisation: 16 lanes per landmark
__global__ void __launch_bounds__(256)
k_lba_linearize(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl) {
  const int w = blockIdx.y;
  if (!(ctl[w].flags & LBA_BUILD)) return;
  const LbaDev& D = devs[w];
  if (blockIdx.x * 16 >= D.n_mp) return;
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
      const double info = (double)o.inv_sigma2, ww = r1 * info;
      const int de = stereo ? 3 : 2;
      int t = 0;
      for (int a = 0; a < 3; a++) {
        for (int b = a; b < 3; b++, t++) {
          double s = 0;
          for (int r = 0; r < de; r++) s += Jx[r * 3 + a] * ww * Jx[r * 3 + b];
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
            for (int r = 0; r < de; r++) s += Jp[r * 6 + a] * ww * Jx[r * 3 + b];
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

// Hpp = 0, bp = 0 of the windows that re-linearise
__global__ void __launch_bounds__(256)
k_lba_zero_pose(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl) {
  const int w = blockIdx.y;
  if (!(ctl[w].flags & LBA_BUILD)) return;
  const LbaDev& D = devs[w];
  const int i = blockIdx.x * 256 + threadIdx.x, np = D.np;
  if (i < np * np) D.Hpp[i] = 0;
  if (i < np) D.bp[i] = 0;
}

// ---- key-frame-parallel Hpp / bp.  grid (chunks, max free key frames, windows)
__global__ void __launch_bounds__(256)
k_lba_pose(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl) {
  __shared__ double s_red[4 * 27];
  const int w = blockIdx.z;
  if (!(ctl[w].flags & LBA_BUILD)) return;
  const LbaDev& D = devs[w];
  if ((int)blockIdx.y >= D.n_free) return;
  const int kfi = D.kf_list[blockIdx.y];
  const LbaKf k = D.kf[kfi];
  const int first = D.kf_edge_first[kfi], cnt = D.kf_edge_first[kfi + 1] - first;
  double acc[27];
#pragma unroll
  for (int i = 0; i < 27; i++) acc[i] = 0;
  PoseXf X;
  kf_xf(D.cam, k, X);
  for (int j = blockIdx.x * 256 + threadIdx.x; j < cnt; j += gridDim.x * 256) {
    const int i = D.kf_edge_idx[first + j];
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

// computeLambdaInit: max |diagonal| over the pose and landmark blocks (one workgroup per window)
__global__ void __launch_bounds__(256)
k_lba_maxdiag(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, WinOut* __restrict__ out) {
  __shared__ double s_m[256];
  const int w = blockIdx.x;
  if (!(ctl[w].flags & LBA_ERROR)) return;
  const LbaDev& D = devs[w];
  double mx = 0;
  for (int j = threadIdx.x; j < D.np; j += 256) mx = fmax(mx, fabs(D.Hpp[(size_t)j * D.np + j]));
  for (int m = threadIdx.x; m < D.n_mp; m += 256)
    if (D.mp_act[m])
      for (int a = 0; a < 3; a++) mx = fmax(mx, fabs(D.Hll[9 * (size_t)m + 4 * a]));
  s_m[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) s_m[threadIdx.x] = fmax(s_m[threadIdx.x], s_m[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[w].maxdiag = s_m[0];
}

// ---- Schur complement: one wavefront per landmark, FP64 MFMA for the block contraction.
typedef double double4_t __attribute__((ext_vector_type(4)));
static const int kSchurMaxObs = 32;  // free observers of one landmark handled by the MFMA tiling

// use_lds = 0: windows whose reduced system does not fit LDS accumulate straight into global memory.
__global__ void __launch_bounds__(256)
k_lba_schur(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, WinOut* __restrict__ out,
            int lds_np_max, int use_lds) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int w = blockIdx.y;
  if (!(ctl[w].flags & LBA_TRIAL)) return;
  const LbaDev& D = devs[w];
  const double lambda = ctl[w].lambda;
  const int np = D.np;
  int lds_np = use_lds ? (np | 1) : np;
  double* sH = use_lds ? smem : D.Hs;  // [ld][ld] accumulated -(B D^-1 B^T)
  double* sb = use_lds ? smem + (size_t)lds_np_max * lds_np_max : D.bs;
  double* stage = use_lds ? smem + (size_t)lds_np_max * lds_np_max + lds_np_max : smem;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (use_lds) {
    for (int i = threadIdx.x; i < lds_np * lds_np; i += 256) sH[i] = 0;
    for (int i = threadIdx.x; i < np; i += 256) sb[i] = 0;
  }
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
            atomicAdd(&sb[col + a], -(b0 * db0 + b1 * db1 + b2 * db2));  // bschur -= B (D^-1 bl)
          }
        }
      }
      k += __popcll(bal);
    }
    if (k > kSchurMaxObs) {
      if (lane == 0) atomicExch(&out[w].overflow, 1);
      k = kSchurMaxObs;
    }
    const int rows = 6 * k, nt = (rows + 15) >> 4;
    for (int r = rows + lane; r < nt * 16; r += 64) {  // zero-pad the last tile
      sA[r * 4] = sA[r * 4 + 1] = sA[r * 4 + 2] = sA[r * 4 + 3] = 0;
      sB[r * 4] = sB[r * 4 + 1] = sB[r * 4 + 2] = sB[r * 4 + 3] = 0;
      sC[r] = -1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // C(ti, tj) = A_ti (16x4) * B_tj^T (4x16) on the matrix core; scatter -C into the system
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
__global__ void __launch_bounds__(256)
k_lba_init_reduced(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, WinOut* __restrict__ out) {
  const int w = blockIdx.y;
  if (!(ctl[w].flags & LBA_TRIAL)) return;
  const LbaDev& D = devs[w];
  const double lambda = ctl[w].lambda;
  const int i = blockIdx.x * 256 + threadIdx.x, np = D.np;
  if (i < np * np) D.Hs[i] = D.Hpp[i] + ((i / np) == (i % np) ? lambda : 0.0);
  if (i < np) D.bs[i] = D.bp[i];
  if (i == 0) out[w].overflow = 0;
}

// ---- dense LDL^T solve of the reduced system, one workgroup per window (matrix in L2)
__global__ void __launch_bounds__(256)
k_lba_ldlt(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, WinOut* __restrict__ out) {
  __shared__ double s_col[512];
  __shared__ double s_D[512];
  __shared__ double s_red[4];
  __shared__ int s_ok;
  const int w = blockIdx.x;
  if (!(ctl[w].flags & LBA_TRIAL)) return;
  const LbaDev& D = devs[w];
  double* A = D.Hs;
  const double* b = D.bs;
  double* x = D.xp;
  const int n = D.np, tid = threadIdx.x;
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
    if (tid == 0) out[w].ok = 0, out[w].scale_p = 0;
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
  double sp[1] = {0};
  const double lambda = ctl[w].lambda;
  for (int i = tid; i < n; i += 256) {
    x[i] = y[i];
    sp[0] += y[i] * (lambda * y[i] + D.bp[i]);  // pose part of computeScale()
  }
  block_sum<1>(sp, s_red, tid);
  if (tid == 0) out[w].ok = 1, out[w].scale_p = sp[0];
}

// ---- back-substitution + update of the points, landmark part of the LM gain-ratio scale
__global__ void __launch_bounds__(256)
k_lba_update_points(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl) {
  __shared__ double s_red[4];
  const int w = blockIdx.y;
  if (!(ctl[w].flags & LBA_TRIAL)) return;
  const LbaDev& D = devs[w];
  if (blockIdx.x * 256 >= D.n_mp) return;
  const double lambda = ctl[w].lambda;
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
  if (threadIdx.x == 0) D.part_m[blockIdx.x] = sc[0];
}

__global__ void __launch_bounds__(64)
k_lba_update_poses(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl) {
  const int w = blockIdx.y;
  if (!(ctl[w].flags & LBA_TRIAL)) return;
  const LbaDev& D = devs[w];
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

// ================================================================== host-side lock-step LM driver
static thread_local hipStream_t g_lba_stream = nullptr;
static thread_local DevBuf g_arena, g_devs, g_ctl, g_out;

// synchronous copy on the calling thread's own stream (concurrent callers and the frame pipelines
// on their streams do not serialise on the null stream)
static inline hipError_t lba_copy(void* dst, const void* src, size_t n, hipMemcpyKind kind) {
  hipError_t e = hipMemcpyAsync(dst, src, n, kind, g_lba_stream);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(g_lba_stream);
}

struct WinHost {  // per-window LM state machine, exactly g2o's (optimization_algorithm_levenberg.cpp)
  const vieo_lba_params* P;
  const vieo_lba_keyframe* kfs;
  const vieo_lba_obs* obs;
  const float* points;
  int n_kf, n_mp, n_obs;
  LbaDev D;                 // device view (host copy)
  std::vector<int> mp_first, mp_count, kf_edge_first, kf_edge_idx, kf_list;
  std::vector<unsigned char> level, mp_act;
  std::vector<LbaKf> kf;
  // device sub-allocations that change between the two optimisations
  int *d_kf_list;
  unsigned char* d_mp_act;
  // state
  int stage = 0;            // 0: optimize(its0), 1: optimize(its1), 2: finished
  int it = 0, iters = 0;    // iteration inside the current optimize()
  int phase = 0;            // 0: needs the initial error pass, 1: in trials
  double lambda = -1, ni = 2, currentChi = 0, iniChi = 0;
  int nBad = 0, qmax = 0;
  bool need_build = false, need_restore = false, skip = false;
  vieo_lba_result* R;
};

}  // namespace vieo

using namespace vieo;

extern "C" {

int vieo_local_bundle_adjustment_batch(int n_windows, const vieo_lba_params* const* params,
                                       const vieo_lba_keyframe* const* h_kfs, const int* n_kf,
                                       const float* const* h_points, const int* n_mp,
                                       const vieo_lba_obs* const* h_obs, const int* n_obs,
                                       volatile const int* stop, vieo_navstate* const* h_navs_out,
                                       float* const* h_points_out, uint8_t* const* h_erase,
                                       vieo_lba_result* h_results) {
  if (n_windows <= 0 || !params || !h_kfs || !n_kf || !h_points || !n_mp || !h_obs || !n_obs ||
      !h_navs_out || !h_points_out || !h_erase || !h_results)
    return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  if (!g_lba_stream) VIEO_HIP_CHECK(hipStreamCreateWithFlags(&g_lba_stream, hipStreamNonBlocking));
  hipStream_t st = g_lba_stream;
  const int W = n_windows;
  std::vector<WinHost> win(W);
  // ---- host-side index structures + arena layout
  size_t arena = 0;
  auto take = [&](size_t bytes) {
    const size_t off = arena;
    arena += (bytes + 255) / 256 * 256;
    return off;
  };
  struct Off {
    size_t obs, kf, kf_bak, X, X_bak, err, level, erase, mp_first, mp_count, mp_act, Bpl, Hll, bl, Dinv, xl,
        Hpp, Hs, bp, bs, xp, part, part_m, kf_list, kf_edge_first, kf_edge_idx;
  };
  std::vector<Off> off(W);
  int max_obs = 0, max_mp = 0, max_kf = 0, max_np = 0;
  for (int w = 0; w < W; w++) {
    WinHost& H = win[w];
    H.P = params[w], H.kfs = h_kfs[w], H.obs = h_obs[w], H.points = h_points[w];
    H.n_kf = n_kf[w], H.n_mp = n_mp[w], H.n_obs = n_obs[w];
    H.R = &h_results[w];
    memset(H.R, 0, sizeof(*H.R));
    if (!H.P || !H.kfs || H.n_kf <= 0 || !H.points || H.n_mp <= 0 || !H.obs || H.n_obs <= 0) return VIEO_E_INVALID;
    for (int k = 0; k < H.n_kf; k++) h_navs_out[w][k] = H.kfs[k].nav;
    memcpy(h_points_out[w], H.points, (size_t)H.n_mp * 12);
    memset(h_erase[w], 0, H.n_obs);
    bool any_free = false;
    for (int k = 0; k < H.n_kf; k++) any_free |= !H.kfs[k].fixed;
    if (!any_free) {
      H.R->status = VIEO_LBA_NO_FREE_POSE;  // Optimizer.cc:1993
      H.skip = true, H.stage = 2;
    }
    if (6 * H.n_kf > 512) {
      set_error("local BA: more than 85 key frames in a window");
      return VIEO_E_CAPACITY;
    }
    H.mp_first.assign(H.n_mp, 0), H.mp_count.assign(H.n_mp, 0);
    std::vector<int> kf_cnt(H.n_kf + 1, 0), fill(H.n_kf, 0);
    for (int i = 0; i < H.n_obs; i++) {
      const int m = H.obs[i].mp, k = H.obs[i].kf;
      if (m < 0 || m >= H.n_mp || k < 0 || k >= H.n_kf || (i > 0 && m < H.obs[i - 1].mp)) {
        set_error("vieo_local_bundle_adjustment: observations must be sorted by map point");
        return VIEO_E_INVALID;
      }
      if (H.mp_count[m] == 0) H.mp_first[m] = i;
      H.mp_count[m]++;
      kf_cnt[k + 1]++;
    }
    H.kf_edge_first.assign(H.n_kf + 1, 0);
    H.kf_edge_idx.resize(H.n_obs);
    for (int k = 0; k < H.n_kf; k++) H.kf_edge_first[k + 1] = H.kf_edge_first[k] + kf_cnt[k + 1];
    for (int i = 0; i < H.n_obs; i++) H.kf_edge_idx[H.kf_edge_first[H.obs[i].kf] + fill[H.obs[i].kf]++] = i;
    H.kf.resize(H.n_kf);
    for (int k = 0; k < H.n_kf; k++) {
      memcpy(H.kf[k].p, H.kfs[k].nav.p, 24);
      H.kf[k].qw = H.kfs[k].nav.q[0], H.kf[k].qx = H.kfs[k].nav.q[1];
      H.kf[k].qy = H.kfs[k].nav.q[2], H.kf[k].qz = H.kfs[k].nav.q[3];
      H.kf[k].col = -1, H.kf[k].pad = 0;
    }
    H.level.assign(H.n_obs, 0), H.mp_act.assign(H.n_mp, 0);
    const int npm = 6 * H.n_kf;
    Off& o = off[w];
    o.obs = take((size_t)H.n_obs * sizeof(vieo_lba_obs));
    o.kf = take((size_t)H.n_kf * sizeof(LbaKf)), o.kf_bak = take((size_t)H.n_kf * sizeof(LbaKf));
    o.X = take((size_t)H.n_mp * 24), o.X_bak = take((size_t)H.n_mp * 24);
    o.err = take((size_t)H.n_obs * 24), o.level = take(H.n_obs), o.erase = take(H.n_obs);
    o.mp_first = take((size_t)H.n_mp * 4), o.mp_count = take((size_t)H.n_mp * 4), o.mp_act = take(H.n_mp);
    o.Bpl = take((size_t)H.n_obs * 144), o.Hll = take((size_t)H.n_mp * 72), o.bl = take((size_t)H.n_mp * 24);
    o.Dinv = take((size_t)H.n_mp * 72), o.xl = take((size_t)H.n_mp * 24);
    o.Hpp = take((size_t)npm * npm * 8), o.Hs = take((size_t)npm * npm * 8);
    o.bp = take((size_t)npm * 8), o.bs = take((size_t)npm * 8), o.xp = take((size_t)npm * 8);
    o.part = take((size_t)((H.n_obs + 255) / 256) * 8), o.part_m = take((size_t)((H.n_mp + 255) / 256) * 8);
    o.kf_list = take((size_t)H.n_kf * 4), o.kf_edge_first = take((size_t)(H.n_kf + 1) * 4);
    o.kf_edge_idx = take((size_t)H.n_obs * 4);
    max_obs = std::max(max_obs, H.n_obs), max_mp = std::max(max_mp, H.n_mp);
    max_kf = std::max(max_kf, H.n_kf), max_np = std::max(max_np, npm);
  }
  if ((rc = g_arena.ensure(arena)) != VIEO_OK) return rc;
  if ((rc = g_devs.ensure((size_t)W * sizeof(LbaDev))) != VIEO_OK) return rc;
  if ((rc = g_ctl.ensure((size_t)W * sizeof(WinCtl))) != VIEO_OK) return rc;
  if ((rc = g_out.ensure((size_t)W * sizeof(WinOut))) != VIEO_OK) return rc;
  uint8_t* base = g_arena.as<uint8_t>();
  for (int w = 0; w < W; w++) {
    WinHost& H = win[w];
    const Off& o = off[w];
    LbaDev& D = H.D;
    memset(&D, 0, sizeof(D));
    D.obs = (const vieo_lba_obs*)(base + o.obs);
    D.n_obs = H.n_obs, D.n_mp = H.n_mp, D.n_kf = H.n_kf, D.np = 0, D.n_free = 0;
    D.kf = (LbaKf*)(base + o.kf), D.kf_bak = (LbaKf*)(base + o.kf_bak);
    D.X = (double*)(base + o.X), D.X_bak = (double*)(base + o.X_bak), D.err = (double*)(base + o.err);
    D.level = base + o.level, D.erase = base + o.erase;
    D.mp_first = (const int*)(base + o.mp_first), D.mp_count = (const int*)(base + o.mp_count);
    D.mp_act = base + o.mp_act;
    H.d_mp_act = base + o.mp_act, H.d_kf_list = (int*)(base + o.kf_list);
    D.Bpl = (double*)(base + o.Bpl), D.Hll = (double*)(base + o.Hll), D.bl = (double*)(base + o.bl);
    D.Dinv = (double*)(base + o.Dinv), D.xl = (double*)(base + o.xl);
    D.Hpp = (double*)(base + o.Hpp), D.Hs = (double*)(base + o.Hs), D.bp = (double*)(base + o.bp);
    D.bs = (double*)(base + o.bs), D.xp = (double*)(base + o.xp);
    D.part = (double*)(base + o.part), D.part_m = (double*)(base + o.part_m);
    D.kf_list = (const int*)(base + o.kf_list), D.kf_edge_first = (const int*)(base + o.kf_edge_first);
    D.kf_edge_idx = (const int*)(base + o.kf_edge_idx);
    D.cam.fx = H.P->fx, D.cam.fy = H.P->fy, D.cam.cx = H.P->cx, D.cam.cy = H.P->cy, D.cam.bf = H.P->bf;
    memcpy(D.cam.Rcb, H.P->Rcb, 72);
    memcpy(D.cam.tcb, H.P->tcb, 24);
    D.robust = 1;
    D.dMono = (double)(float)sqrt(5.991), D.dStereo = (double)(float)sqrt(7.815);
    std::vector<double> X((size_t)H.n_mp * 3);
    for (int i = 0; i < H.n_mp * 3; i++) X[i] = (double)H.points[i];
    VIEO_HIP_CHECK(hipMemcpyAsync(base + o.obs, H.obs, (size_t)H.n_obs * sizeof(vieo_lba_obs), hipMemcpyHostToDevice, st));
    VIEO_HIP_CHECK(hipMemcpyAsync(base + o.X, X.data(), X.size() * 8, hipMemcpyHostToDevice, st));
    VIEO_HIP_CHECK(hipMemcpyAsync(base + o.mp_first, H.mp_first.data(), (size_t)H.n_mp * 4, hipMemcpyHostToDevice, st));
    VIEO_HIP_CHECK(hipMemcpyAsync(base + o.mp_count, H.mp_count.data(), (size_t)H.n_mp * 4, hipMemcpyHostToDevice, st));
    VIEO_HIP_CHECK(hipMemcpyAsync(base + o.kf_edge_first, H.kf_edge_first.data(), (size_t)(H.n_kf + 1) * 4, hipMemcpyHostToDevice, st));
    VIEO_HIP_CHECK(hipMemcpyAsync(base + o.kf_edge_idx, H.kf_edge_idx.data(), (size_t)H.n_obs * 4, hipMemcpyHostToDevice, st));
    VIEO_HIP_CHECK(hipMemsetAsync(base + o.level, 0, H.n_obs, st));
    VIEO_HIP_CHECK(hipMemsetAsync(base + o.err, 0, (size_t)H.n_obs * 24, st));
    VIEO_HIP_CHECK(hipStreamSynchronize(st));  // X goes out of scope
  }
  const int lds_np_max = max_np | 1;
  const size_t stage_lds = 4 * (192 * 4 * 2 + 192) * 8 + 64;
  size_t schur_lds = ((size_t)lds_np_max * lds_np_max + lds_np_max) * 8 + stage_lds;
  const int use_lds = schur_lds <= 150 * 1024;
  if (!use_lds) schur_lds = stage_lds;
  VIEO_HIP_CHECK(hipFuncSetAttribute((const void*)k_lba_schur, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)schur_lds));
  std::vector<WinCtl> ctl(W);
  std::vector<WinOut> out(W);
  std::vector<LbaDev> devs(W);
  bool first_upload = true;

  // initializeOptimization(0) for a window: active vertices, reduced-system columns
  auto begin_optimize = [&](WinHost& H, int iterations) -> int {
    std::vector<char> kf_act(H.n_kf, 0);
    std::fill(H.mp_act.begin(), H.mp_act.end(), 0);
    bool any = false;
    for (int i = 0; i < H.n_obs; i++)
      if (!H.level[i]) kf_act[H.obs[i].kf] = 1, H.mp_act[H.obs[i].mp] = 1, any = true;
    int np = 0;
    H.kf_list.clear();
    for (int k = 0; k < H.n_kf; k++) {
      if (!H.kfs[k].fixed && kf_act[k]) {
        H.kf[k].col = np, np += 6;
        H.kf_list.push_back(k);
      } else
        H.kf[k].col = -1;
    }
    H.D.np = np, H.D.n_free = (int)H.kf_list.size();
    H.it = 0, H.iters = iterations, H.phase = 0, H.need_restore = false;
    std::vector<LbaKf> cur(H.n_kf);
    if (H.stage == 0)
      cur = H.kf;
    else {
      VIEO_HIP_CHECK(lba_copy(cur.data(), H.D.kf, (size_t)H.n_kf * sizeof(LbaKf), hipMemcpyDeviceToHost));
      for (int k = 0; k < H.n_kf; k++) cur[k].col = H.kf[k].col;
    }
    VIEO_HIP_CHECK(hipMemcpyAsync(H.D.kf, cur.data(), (size_t)H.n_kf * sizeof(LbaKf), hipMemcpyHostToDevice, st));
    VIEO_HIP_CHECK(hipMemcpyAsync(H.d_mp_act, H.mp_act.data(), H.n_mp, hipMemcpyHostToDevice, st));
    VIEO_HIP_CHECK(hipMemcpyAsync(H.d_kf_list, H.kf_list.data(), H.kf_list.size() * 4, hipMemcpyHostToDevice, st));
    VIEO_HIP_CHECK(hipStreamSynchronize(st));
    return (!any || np == 0 || iterations <= 0) ? 1 : 0;  // 1: nothing to optimise
  };

  const bool stopped0 = stop && *stop;
  for (int w = 0; w < W; w++) {
    WinHost& H = win[w];
    if (H.skip) continue;
    if (stopped0) {
      H.R->status = VIEO_LBA_ABORTED, H.stage = 2;
      continue;
    }
    int r = begin_optimize(H, H.P->its0);
    if (r < 0) return r;
    if (r == 1) H.phase = 2;  // empty optimisation: go straight to the stage transition
  }

  // ---- lock-step rounds
  for (int guard = 0; guard < 400; guard++) {
    bool any_work = false, any_class = false;
    for (int w = 0; w < W; w++) {
      WinHost& H = win[w];
      WinCtl& c = ctl[w];
      c.flags = 0, c.pad = 0, c.lambda = H.lambda;
      if (H.stage >= 2) continue;
      if (H.phase == 2) {  // optimize() finished -> stage transition handled below
        continue;
      }
      if (H.phase == 0) {
        c.flags = LBA_ERROR | LBA_BUILD;  // computeActiveErrors + buildSystem of iteration 0
      } else {
        c.flags = LBA_TRIAL | (H.need_build ? LBA_BUILD : 0) | (H.need_restore ? LBA_RESTORE : 0);
      }
      any_work = true;
    }
    // stage transitions (classification) for windows whose optimize() ended
    for (int w = 0; w < W; w++) {
      WinHost& H = win[w];
      if (H.stage < 2 && H.phase == 2) {
        ctl[w].flags = (H.stage == 0 && !(stop && *stop)) ? LBA_CLASS0 : LBA_CLASS1;
        if (H.need_restore) ctl[w].flags |= LBA_RESTORE;
        any_class = true;
      }
    }
    if (!any_work && !any_class) break;
    for (int w = 0; w < W; w++) devs[w] = win[w].D;
    VIEO_HIP_CHECK(hipMemcpyAsync(g_devs.p, devs.data(), (size_t)W * sizeof(LbaDev), hipMemcpyHostToDevice, st));
    VIEO_HIP_CHECK(hipMemcpyAsync(g_ctl.p, ctl.data(), (size_t)W * sizeof(WinCtl), hipMemcpyHostToDevice, st));
    (void)first_upload;
    const LbaDev* dD = g_devs.as<LbaDev>();
    const WinCtl* dC = g_ctl.as<WinCtl>();
    WinOut* dO = g_out.as<WinOut>();
    const int ge = (max_obs + 255) / 256, gm = (max_mp + 255) / 256;
    const int gx = std::max((max_mp * 3 + 255) / 256, (max_kf + 255) / 256);
    hipLaunchKernelGGL(k_lba_backup_restore, dim3(gx, W), dim3(256), 0, st, dD, dC, 1);
    if (any_class) hipLaunchKernelGGL(k_lba_classify, dim3(ge, W), dim3(256), 0, st, dD, dC);
    if (any_work) {
      // phase-0 windows: residuals first (their chi2 is the iteration's currentChi)
      hipLaunchKernelGGL(k_lba_zero_pose, dim3((max_np * max_np + 255) / 256, W), dim3(256), 0, st, dD, dC);
      hipLaunchKernelGGL(k_lba_linearize, dim3((max_mp + 15) / 16, W), dim3(256), 0, st, dD, dC);
      hipLaunchKernelGGL(k_lba_pose, dim3(8, max_kf, W), dim3(256), 0, st, dD, dC);
      hipLaunchKernelGGL(k_lba_maxdiag, dim3(W), dim3(256), 0, st, dD, dC, dO);
      hipLaunchKernelGGL(k_lba_backup_restore, dim3(gx, W), dim3(256), 0, st, dD, dC, 0);
      hipLaunchKernelGGL(k_lba_init_reduced, dim3((max_np * max_np + 255) / 256, W), dim3(256), 0, st, dD, dC, dO);
      hipLaunchKernelGGL(k_lba_schur, dim3(std::max(1, std::min(std::max(16, 512 / W), (max_mp + 3) / 4)), W), dim3(256),
                         schur_lds, st, dD, dC, dO, lds_np_max, use_lds);
      hipLaunchKernelGGL(k_lba_ldlt, dim3(W), dim3(256), 0, st, dD, dC, dO);
      hipLaunchKernelGGL(k_lba_update_points, dim3(gm, W), dim3(256), 0, st, dD, dC);
      hipLaunchKernelGGL(k_lba_update_poses, dim3((max_kf + 63) / 64, W), dim3(64), 0, st, dD, dC);
      hipLaunchKernelGGL(k_lba_error, dim3(ge, W), dim3(256), 0, st, dD, dC);
      hipLaunchKernelGGL(k_lba_reduce, dim3(W), dim3(256), 0, st, dD, dC, dO);
      VIEO_HIP_CHECK(lba_copy(out.data(), g_out.p, (size_t)W * sizeof(WinOut), hipMemcpyDeviceToHost));
    }
    VIEO_HIP_CHECK(hipGetLastError());
    // ---- per-window policy (optimization_algorithm_levenberg.cpp:61-164)
    for (int w = 0; w < W; w++) {
      WinHost& H = win[w];
      const int fl = ctl[w].flags;
      if (fl & (LBA_CLASS0 | LBA_CLASS1)) {
        H.need_restore = false;
        if (fl & LBA_CLASS0) {
          VIEO_HIP_CHECK(lba_copy(H.level.data(), H.D.level, H.n_obs, hipMemcpyDeviceToHost));
          H.stage = 1;
          H.D.robust = 0;
          int r = begin_optimize(H, H.P->its1);
          if (r < 0) return r;
          H.phase = r == 1 ? 2 : 0;
        } else {
          if (H.stage == 0) H.R->status = VIEO_LBA_ABORTED;  // stop flag between the two stages
          H.stage = 2;
        }
        continue;
      }
      if (fl & LBA_ERROR) {  // start of an optimize(): error pass + first linearisation are done
        H.R->lm_iterations++;
        H.currentChi = out[w].chi2;
        if (H.stage == 0) H.R->chi2_initial = H.currentChi;
        H.iniChi = H.currentChi;
        H.lambda = 1e-5 * out[w].maxdiag;  // computeLambdaInit
        H.ni = 2, H.nBad = 0, H.qmax = 0;
        H.phase = 1, H.need_build = false, H.need_restore = false;
        continue;
      }
      if (!(fl & LBA_TRIAL)) continue;
      if (out[w].overflow) {
        set_error("local BA: a map point has more than %d free observers", kSchurMaxObs);
        return VIEO_E_CAPACITY;
      }
      H.R->lm_trials++;
      H.need_build = false;
      const bool ok2 = out[w].ok != 0;
      double tempChi = ok2 ? out[w].chi2 : DBL_MAX;
      double rho = H.currentChi - tempChi;
      double scale = (ok2 ? out[w].scale_l + out[w].scale_p : 0.0) + 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow(2 * rho - 1, 3);
        alpha = std::min(alpha, 2. / 3.);
        H.lambda *= std::max(1. / 3., alpha);
        H.ni = 2;
        H.currentChi = tempChi;
        H.need_restore = false;
      } else {
        H.lambda *= H.ni;
        H.ni *= 2;
        H.need_restore = true;
      }
      H.qmax++;
      H.R->chi2_final = H.currentChi;
      const bool again = rho < 0 && H.qmax < 10 && !(stop && *stop);
      if (again) continue;  // next lambda trial of the same iteration
      // ---- the iteration is over
      bool terminate = (H.qmax == 10 || rho == 0);
      if (!terminate) {
        if ((H.iniChi - H.currentChi) * 1e3 < H.iniChi)
          H.nBad++;
        else
          H.nBad = 0;
        if (H.nBad >= 3) terminate = true;
      }
      H.it++;
      if (terminate || H.it >= H.iters || (stop && *stop)) {
        H.phase = 2;
      } else {
        H.R->lm_iterations++;
        H.iniChi = H.currentChi;
        H.qmax = 0;
        H.need_build = true;  // buildSystem at the accepted state (errors there are already stored)
      }
    }
  }
  // ---- results
  for (int w = 0; w < W; w++) {
    WinHost& H = win[w];
    if (H.skip || (stopped0)) continue;
    VIEO_HIP_CHECK(lba_copy(h_erase[w], H.D.erase, H.n_obs, hipMemcpyDeviceToHost));
    for (int i = 0; i < H.n_obs; i++) H.R->n_erase += h_erase[w][i];
    std::vector<LbaKf> o(H.n_kf);
    std::vector<double> X((size_t)H.n_mp * 3);
    VIEO_HIP_CHECK(lba_copy(o.data(), H.D.kf, (size_t)H.n_kf * sizeof(LbaKf), hipMemcpyDeviceToHost));
    VIEO_HIP_CHECK(lba_copy(X.data(), H.D.X, X.size() * 8, hipMemcpyDeviceToHost));
    for (int k = 0; k < H.n_kf; k++) {
      if (H.kfs[k].fixed) continue;
      memcpy(h_navs_out[w][k].p, o[k].p, 24);
      h_navs_out[w][k].q[0] = o[k].qw, h_navs_out[w][k].q[1] = o[k].qx;
      h_navs_out[w][k].q[2] = o[k].qy, h_navs_out[w][k].q[3] = o[k].qz;
    }
    for (int i = 0; i < H.n_mp * 3; i++) h_points_out[w][i] = (float)X[i];  // SetWorldPos(cast<float>)
  }
  return VIEO_OK;
}

int vieo_local_bundle_adjustment(const vieo_lba_params* P, const vieo_lba_keyframe* h_kfs, int n_kf,
                                 const float* h_points, int n_mp, const vieo_lba_obs* h_obs, int n_obs,
                                 volatile const int* stop, vieo_navstate* h_navs_out,
                                 float* h_points_out, uint8_t* h_erase, vieo_lba_result* R) {
  if (!P || !h_kfs || n_kf <= 0 || !h_points || n_mp <= 0 || !h_obs || n_obs <= 0 || !h_navs_out ||
      !h_points_out || !h_erase || !R)
    return VIEO_E_INVALID;
  return vieo_local_bundle_adjustment_batch(1, &P, &h_kfs, &n_kf, &h_points, &n_mp, &h_obs, &n_obs, stop,
                                            &h_navs_out, &h_points_out, &h_erase, R);
}

}  // extern "C"
