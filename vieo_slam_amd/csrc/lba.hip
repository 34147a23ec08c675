// lba.hip -- Optimizer::LocalBundleAdjustment on gfx950 (reference: src/Optimizer.cc:1876-2307;
// g2o BlockSolver<6,3> with Schur complement, block_solver.hpp:353-589; LM
// optimization_algorithm_levenberg.cpp:61-207; edges src/Odom/g2otypes.h:321-547).
//
// Several independent windows advance in lock step; every kernel covers all windows of the batch
// (grid y / x = window) and returns at once for windows whose control word does not ask for the step.
// Device side (all FP64, no floating-point atomics: every sum has a fixed order, so a window's result
// does not depend on what it is batched with):
//   k_lba_begin      initializeOptimization(): active key frames / points from the edge levels, the
//                    reduced-system column of every free key frame, the (free kf, point) -> edge table
//   k_lba_error      edge-parallel residuals + robust chi2 (per-block partial sums)
//   k_lba_build      buildSystem: one thread per point over its observations (H_ll, b_l); one
//                    workgroup per free key frame over its edge list (H_pp, b_p) which also writes the
//                    6 x 3 blocks B = Jp^T W Jx of its edges, COMPACT: one 144-byte block per observation
//                    (CB[edge]); `tab` maps (free key frame, point) to the edge that carries the block
//   k_lba_lambda     computeLambdaInit (tau * max diagonal) for windows starting an optimize()
//   k_lba_schur      the Schur complement as what it is, a GEMM: (BB D^-1) x [BB; b_l]^T with
//                    D = blockdiag(H_ll + lambda I), K = 3 x points split over workgroups, 64x64 output
//                    tiles on the FP64 matrix cores (v_mfma_f64_16x16x4_f64); the dense operand tiles exist
//                    only in LDS: a K-chunk is staged from the compact blocks through `tab` (zeros where a key
//                    frame does not see a point), D^-1 applied on the way; per-split partial products
//   k_lba_assemble   Hs = H_pp + lambda I - sum of the partials (fixed order), bs likewise
//   k_lba_ldlt16     one workgroup per window: blocked LDL^T (FP64 MFMA) of the reduced system in LDS, solve, pose
//                    retraction (with backup) and the pose part of the gain-ratio scale
//   k_lba_ldlt       the same with column panels, for systems of 160 .. 510 unknowns
//   k_lba_update_points  back-substitution x_l = D^-1 (b_l - B^T x_p), point update (with backup)
//   k_lba_restore / k_lba_classify  rejected-step rollback; chi2 / depth gates
// The Levenberg-Marquardt policy (lambda, accept / reject, termination, stop flag) runs on the host
// exactly as g2o's does, from one 56-byte record per window and round.
//
// Full BA with bScaleOpt (System::FinalGBA, src/System.cc:24-33; Optimizer.cc:842-851,1131-1137,1190-1196,1256-1335):
// the VertexScale (g2otypes.h:292-311) is one more 1-dim column at the END of the pose system and the visual edges are
// EdgeReprojectPRS / PRSStereo (g2otypes.h:321-541, MODE_OPT_VAR == 1): Xw = s Xh, J_s = (Jproj Rcw) Xh,
// J_Xh = s (Jproj Rcw).  The scale sees every point, so it is one more (dense) row of BB -- row 6 x free key frames --
// and the Schur GEMM, the pack / all-reduce and the solves take it like any other row; its own blocks (H_ss, b_s,
// H_ps) are summed in fixed order by k_lba_build / k_lba_scale_fold and gathered by k_lba_assemble.
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include "imu_device.h"
#include "rccl_dl.h"

namespace vieo {

struct LbaKf {
  double p[3], qw, qx, qy, qz;
  int col;    // offset of the key frame's block in the reduced pose system, -1 = fixed / inactive
  int fixed;
  double v[3], dbg[3], dba[3], bg[3], ba[3];  // visual-inertial windows (a18) only
};

// one key-frame pair of a visual-inertial window: EdgeNavStatePRV + EdgeNavStateBias
struct LbaImu {
  int i, j;          // key frames (kf_i = previous)
  int has_imu, robust;
  double InfoI[81];  // Sigma_PRV^-1, x 1e-2 when kf_i is fixed (Optimizer.cc:247-259)
  double infoBg, infoBa;
  vieo_imu_preint M;
  int has_enc, enc_robust;  // EdgeEncNavStatePR of the pair (Optimizer.cc:323-347)
  double measE[6], InfoE[36];
};

// control word of a window for one round of the lock-step driver
enum {
  LBA_TRIAL = 1,    // solve + update + evaluate one lambda trial
  LBA_BUILD = 2,    // re-linearise (start of an LM iteration)
  LBA_RESTORE = 4,  // the last trial was rejected: restore the backed-up estimates first
  LBA_BEGIN = 8,    // start of an optimize(): active sets, initial chi2, lambda init
  LBA_CLASS0 = 16,  // chi2 / depth gates -> level 1 (between the two optimisations)
  LBA_CLASS1 = 32,  // final erase flags
  LBA_ROBUST = 64,  // Huber kernels on (first optimisation)
  LBA_PRELEVEL = 128,  // GraphOperator::Chi2LargeSetLevel before the first optimisation (a18)
};
struct WinCtl {
  int flags, pad;
  double lambda;  // < 0: take the device-computed initial lambda
};
struct WinOut {
  double chi0, chi2, scale_l, scale_p, lambda;
  int ok, np;
  double chig0, chig;  // landmark-sharded windows: the (replicated) inertial part, kept out of the reduction
};

static const int kBuildChunk = 512;  // edges of a key frame per run (= workgroup) of k_lba_build's key-frame half (2 per thread)

struct LbaDev {
  const vieo_lba_obs* obs;
  int n_obs, n_mp, n_kf, nf_cap;  // nf_cap: non-fixed key frames = rows of `tab`
  int np, n_free, npv;            // written by k_lba_begin: np = pd * n_free, npv = 6 * n_free
  int pd;                         // reduced-system dims per free key frame: 6 (PR) or 15 (PR + V + Bias)
  int n_imu;
  const LbaImu* imu;              // [n_imu]
  const int *kf_in, *kf_out;      // [n_kf] inertial edge ending / starting at the key frame, -1 = none
  double* Ae;                     // [n_imu][930] generic Hessian 30x30 + gradient 30, local order
                                  //   [kf_i: PR V Bias | kf_j: PR V Bias]
  double *gchi0, *gchi;           // [n_imu] robust chi2 of the inertial edges (at linearisation / after a trial)
  double* bfull;                  // [np] gradient of the pose block (visual + inertial)
  double* red;                    // landmark-sharded window: [S npv x (npv+1) | Hpp | bp], summed over the ranks
  double* red_sc;                 //   and its scalars [chi0, chi2, scale_l, pad]
  const unsigned char* close;     // [n_mp] bClose flags or null
  double thMono, thMonoClose, thStereo;  // chi2 gates of the classification
  double gw[3];
  double th_dist_far;             // > 0: the far-point rule of the visual-inertial local BA is on
  double qRbe[4], pbe[3];         // body <- encoder extrinsics of the encoder edges
  int ldS;                        // leading dimension of a partial Schur product
  int* kf_list;                   // [n_free] free + active key frames in column order
  int* kf_act;                    // [n_kf] scratch of k_lba_begin: the key frame has an active edge
  const int *kf_edge_first, *kf_edge_idx;  // edges grouped by key frame
  int* tab;                       // [nf_cap][n_mp] edge of (free kf ordinal, point), -1 = none
  LbaKf *kf, *kf_bak;
  double *X, *X_bak;              // [n_mp][3]
  double* err;                    // [n_obs][3]
  unsigned char *level, *erase, *mp_act;
  const int *mp_first, *mp_count;  // [n_mp]
  double* CB;                     // [n_obs][18] B = Jp^T W Jx of an edge (6 x 3, row-major); valid where `tab` points
  double* Bs;                     // [n_mp][3] the scale vertex's row (bScaleOpt): it sees every point
  double* Sp;                     // [ksplit][sp_rows][ldS] partial Schur products
  size_t sp_stride;
  int ksplit;                     // K splits of this window's Schur GEMM (a function of the window alone)
  double *Hll, *bl, *Hpp, *bp, *Hs, *bs, *xp;
  unsigned char* occ;             // [(npv + 64) / 64][chunks of 16 landmarks]: the row tile has an edge in the chunk
  int use_occ;                    // full BA only: a local window is dense at that granularity, its Schur GEMM does not look
  double *Hb, *Wp;                // tiled solve of a large reduced system: padded copy [nb][nb], panel [nb][64]
  int* big_fail;                  //   and its "not positive definite" flag
  int nb;
  double *part0, *part, *part_m, *pmax;  // per-block partials
  double* part_t;                 // [blocks of 64 points] robust chi2 of a trial, summed per point block (k_lba_tail)
  int* tail_cnt;                  // arrival counter of k_lba_tail's workgroups (the last one folds the partials)
  // key-frame half of k_lba_build: a key frame's edge list in chunks of kBuildChunk edges, one workgroup each
  int n_chunks;                   // written by k_lba_begin: chunks of the free + active key frames
  int fold_kernel;                // batches: the runs' sums are added by k_lba_build_fold, not by the last workgroup to arrive
                                  // (its device-scope fence, an L2 write-back per workgroup on this multi-XCD part, cost a
                                  // 205-window step 3.2 -> 13.6 ms of k_lba_build)
  int *chunk_first, *chunk_kf;    // [n_free + 1] first chunk of kf_list[a]; [n_chunks] the a of a chunk
  int* chunk_cnt;                 // [n_free] arrival counters (the last workgroup of a key frame folds its chunks' partials)
  double* chunk_part;             // [n_chunks][33] H_pp (21) + b_p (6) + H_ps (6) of a chunk
  CamD cam;                       // the single rectified pinhole camera (n_cams == 0) ...
  CamD cams[4];                   // ... or the physical cameras of a distorted multi-camera rig
  int n_cams;
  const unsigned char* ocam;      // [n_obs] camera of every observation (n_cams > 0)
  double dMono, dStereo;
  int solver;                     // which solve kernel owns the window in a mixed batch: 0 blocked LDS (<= 159 unknowns),
                                  // 1 column panels (LDS up to ~186 unknowns), 2 tiled multi-workgroup LDL^T
  int scale_opt;                  // bScaleOpt: the VertexScale is the last column of the pose system
  double* scl;                    // [2] VertexScale estimate, its backup (push / pop)
  double* sc_sys;                 // [6 nf_cap + 2] H_ps per free key frame (Jp^T W Js), then H_ss, b_s
  double* psc;                    // [blocks of 64 points][2] partial sums of H_ss, b_s
};

__device__ __forceinline__ double win_scale(const LbaDev& D) { return D.scale_opt ? D.scl[0] : 1.0; }
// s * Xh: the point the edges project (g2otypes.h:376; s == 1 without the scale vertex)
__device__ __forceinline__ void scaled_point(const LbaDev& D, int m, double* Xs) {
  const double s = win_scale(D);
  Xs[0] = D.X[3 * (size_t)m] * s, Xs[1] = D.X[3 * (size_t)m + 1] * s, Xs[2] = D.X[3 * (size_t)m + 2] * s;
}

__device__ __forceinline__ const CamD& obs_cam(const LbaDev& D, int i) { return D.n_cams ? D.cams[D.ocam[i]] : D.cam; }

__device__ __forceinline__ void kf_xf(const CamD& c, const LbaKf& k, PoseXf& X) {
  Est e;
  e.p[0] = k.p[0], e.p[1] = k.p[1], e.p[2] = k.p[2];
  e.qw = k.qw, e.qx = k.qx, e.qy = k.qy, e.qz = k.qz;
  make_xf(c, e, X);
}

// residual with a double-precision point (the LBA point vertex is double, unlike PoseOpt's)
__device__ __forceinline__ double lba_edge_error(const CamD& c, const PoseXf& X, const vieo_lba_obs& o,
                                                 const double* Xw, double* err, double* Pc) {
  for (int i = 0; i < 3; i++)
    Pc[i] = X.Rcw[i * 3] * Xw[0] + X.Rcw[i * 3 + 1] * Xw[1] + X.Rcw[i * 3 + 2] * Xw[2] + X.tcw[i];
  double uv[2];
  cam_project(c, Pc, uv, nullptr);
  const double u = uv[0], v = uv[1];
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
  double J[9], Jc[6], uv[2];
  cam_project(c, Pc, uv, Jc);  // Jproj = -d(u, v)/dPc (g2otypes.h:453-459)
  J[0] = -Jc[0], J[1] = -Jc[1], J[2] = -Jc[2];
  J[3] = -Jc[3], J[4] = -Jc[4], J[5] = -Jc[5];
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


__device__ __forceinline__ double win_lambda(const WinCtl& c, const WinOut& o) {
  return c.lambda >= 0 ? c.lambda : o.lambda;
}

// (H_ll + lambda I)^-1 of one landmark
__device__ __forceinline__ void landmark_dinv(const double* H, double lambda, double* Di) {
  const double a00 = H[0] + lambda, a01 = H[1], a02 = H[2], a10 = H[3], a11 = H[4] + lambda, a12 = H[5],
               a20 = H[6], a21 = H[7], a22 = H[8] + lambda;
  const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
  const double id = 1.0 / (a00 * c00 + a01 * c01 + a02 * c02);
  Di[0] = c00 * id, Di[1] = (a02 * a21 - a01 * a22) * id, Di[2] = (a01 * a12 - a02 * a11) * id;
  Di[3] = c01 * id, Di[4] = (a00 * a22 - a02 * a20) * id, Di[5] = (a02 * a10 - a00 * a12) * id;
  Di[6] = c02 * id, Di[7] = (a01 * a20 - a00 * a21) * id, Di[8] = (a00 * a11 - a01 * a10) * id;
}

// ---- rollback of a rejected trial (g2o pop()): only what the trial changed
__global__ void __launch_bounds__(256)
k_lba_restore(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl) {
  const int w = blockIdx.y;
  if (!(ctl[w].flags & LBA_RESTORE)) return;
  const LbaDev& D = devs[w];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < D.n_mp && D.mp_act[i])
    for (int a = 0; a < 3; a++) D.X[3 * (size_t)i + a] = D.X_bak[3 * (size_t)i + a];
  if (i < D.n_kf && D.kf[i].col >= 0) D.kf[i] = D.kf_bak[i];
  if (i == 0 && D.scale_opt) D.scl[0] = D.scl[1];
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
  kf_xf(obs_cam(D, i), D.kf[o.kf], X);
  double Xw[3];
  scaled_point(D, o.mp, Xw);
  const double z = X.Rcw[6] * Xw[0] + X.Rcw[7] * Xw[1] + X.Rcw[8] * Xw[2] + X.tcw[2];
  const double th = o.ur >= 0 ? D.thStereo : ((D.close && D.close[o.mp]) ? D.thMonoClose : D.thMono);
  const bool bad = chi2 > th || !(z > 0.);
  if (fl & LBA_CLASS0) {
    if (bad) D.level[i] = 1;
  } else
    D.erase[i] = bad ? 1 : 0;
}

// GraphOperator::Chi2LargeSetLevel(edges, dim, 100.f, false) (g2o_graph_operator.h:23-40): every edge's
// error is computed and stored; chi2 > 100 * chi2_95(dim) puts the edge on level 1
__global__ void __launch_bounds__(256)
k_lba_prelevel(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, int phase) {
  // phase 0: clear the per-point marks (mp_act is free until k_lba_begin); phase 1: Chi2LargeSetLevel, and mark the
  // points a monocular edge sees closer than th_dist_far; phase 2: th_dist_far (Optimizer.cc:395,454,513-517) --
  // the monocular edges of an unmarked point go to level 1
  const int w = blockIdx.y;
  if (!(ctl[w].flags & LBA_PRELEVEL)) return;
  const LbaDev& D = devs[w];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool far_rule = D.th_dist_far > 0;
  if (phase == 0) {
    if (far_rule)
      for (int m = i; m < D.n_mp; m += gridDim.x * 256) D.mp_act[m] = 0;
    return;
  }
  if (i >= D.n_obs) return;
  const vieo_lba_obs o = D.obs[i];
  if (phase == 2) {
    if (far_rule && o.ur < 0 && !D.mp_act[o.mp]) D.level[i] = 1;
    return;
  }
  PoseXf X;
  const CamD& C = obs_cam(D, i);
  kf_xf(C, D.kf[o.kf], X);
  double err[3], Pc[3], Xs[3];
  scaled_point(D, o.mp, Xs);
  const double chi2 = lba_edge_error(C, X, o, Xs, err, Pc);
  D.err[3 * (size_t)i] = err[0], D.err[3 * (size_t)i + 1] = err[1], D.err[3 * (size_t)i + 2] = err[2];
  if (far_rule && o.ur < 0 && Pc[2] < D.th_dist_far) D.mp_act[o.mp] = 1;  // same value from every writer
  const float th = 100.f * (o.ur >= 0 ? 7.815f : 5.991f);
  if (chi2 > (double)th) D.level[i] = 1;
}

// tab = -1 (and the zero block of CB) for the windows that start an optimize()
__global__ void __launch_bounds__(256)
k_lba_zero(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl) {
  const int w = blockIdx.y;
  if (!(ctl[w].flags & LBA_BEGIN)) return;
  const LbaDev& D = devs[w];
  const size_t nt = (size_t)D.nf_cap * D.n_mp;
  const size_t step = (size_t)gridDim.x * 256;
  if (blockIdx.x == 0 && threadIdx.x < 18) D.CB[18 * (size_t)D.n_obs + threadIdx.x] = 0.0;  // the zero block of absent pairs
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nt; i += step) D.tab[i] = -1;
}

// ---- initializeOptimization(0): one workgroup per window
__global__ void __launch_bounds__(1024)
k_lba_begin(LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, WinOut* __restrict__ out) {
  const int w = blockIdx.x, tid = threadIdx.x;
  if (!(ctl[w].flags & LBA_BEGIN)) return;
  LbaDev& D = devs[w];
  const int n_mp = D.n_mp, n_obs = D.n_obs, n_kf = D.n_kf;
  int* s_act = D.kf_act;
  for (int k = tid; k < n_kf; k += 1024) s_act[k] = 0;
  for (int m = tid; m < n_mp; m += 1024) D.mp_act[m] = 0;
  __syncthreads();
  for (int i = tid; i < n_obs; i += 1024)
    if (D.level[i] == 0) {
      const vieo_lba_obs o = D.obs[i];
      s_act[o.kf] = 1;
      D.mp_act[o.mp] = 1;
    }
  if (D.pd == 6)  // encoder edges of a vision-only window are active edges too (their vertices join the system)
    for (int e = tid; e < D.n_imu; e += 1024) s_act[D.imu[e].i] = 1, s_act[D.imu[e].j] = 1;
  __syncthreads();
  if (tid == 0) {
    // vision-only window: free key frames with an active edge; visual-inertial window: the inertial
    // edges keep every free key frame active
    int np = 0, nf = 0;
    const int pd = D.pd;
    for (int k = 0; k < n_kf; k++) {
      if (!D.kf[k].fixed && (s_act[k] || pd == 15)) {
        D.kf[k].col = np, np += pd;
        D.kf_list[nf++] = k;
      } else
        D.kf[k].col = -1;
    }
    if (D.scale_opt) np += 1;  // id_scale = maxKFid + 1: after every key-frame vertex
    D.np = np, D.n_free = nf, D.npv = 6 * nf + (D.scale_opt ? 1 : 0);
    out[w].np = np;
    *D.tail_cnt = 0;  // (scratch memory: k_lba_tail's arrival counter starts an optimize() at zero)
    int nchk = 0;
    for (int a = 0; a < nf; a++) {
      const int k = D.kf_list[a];
      D.chunk_first[a] = nchk, D.chunk_cnt[a] = 0;
      nchk += max(1, (D.kf_edge_first[k + 1] - D.kf_edge_first[k] + kBuildChunk - 1) / kBuildChunk);
    }
    D.chunk_first[nf] = nchk, D.n_chunks = nchk;
  }
  __syncthreads();
  for (int a = tid; a < D.n_free; a += 1024)
    for (int c = D.chunk_first[a]; c < D.chunk_first[a + 1]; c++) D.chunk_kf[c] = a;
  for (int i = tid; i < n_obs; i += 1024)
    if (D.level[i] == 0) {
      const vieo_lba_obs o = D.obs[i];
      const int c = D.kf[o.kf].col;
      if (c >= 0) D.tab[(size_t)(c / D.pd) * n_mp + o.mp] = i;
    }
}

// ---- residuals + robust chi2 of the active edges.  which = 0: at the start of an optimize()
// (computeActiveErrors before the first iteration), which = 1: after a trial step
__global__ void __launch_bounds__(256)
k_lba_error(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, int which) {
  __shared__ double s_red[4];
  const int w = blockIdx.y, fl = ctl[w].flags;
  if (!(fl & (which ? LBA_TRIAL : LBA_BEGIN))) return;
  const LbaDev& D = devs[w];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= D.n_obs || D.np == 0) return;  // np == 0: optimize() returns before any error pass
  double v[1] = {0};
  if (i < D.n_obs && D.level[i] == 0) {
    const vieo_lba_obs o = D.obs[i];
    PoseXf X;
    const CamD& C = obs_cam(D, i);
    kf_xf(C, D.kf[o.kf], X);
    double err[3], Pc[3], Xs[3];
    scaled_point(D, o.mp, Xs);
    const double chi2 = lba_edge_error(C, X, o, Xs, err, Pc);
    D.err[3 * (size_t)i] = err[0], D.err[3 * (size_t)i + 1] = err[1], D.err[3 * (size_t)i + 2] = err[2];
    double r0 = chi2, r1;
    if (fl & LBA_ROBUST) {
      const double dl = o.ur >= 0 ? D.dStereo : D.dMono;
      huber(chi2, dl, dl * dl, &r0, &r1);
    }
    v[0] = r0;
  }
  block_sum<1>(v, s_red, threadIdx.x);
  if (threadIdx.x == 0) (which ? D.part : D.part0)[blockIdx.x] = v[0];
}

// one workgroup per window: fold the per-block partials into the window's output record
// per_point: the trial's chi2 partials are k_lba_tail's (one per block of 64 points) instead of k_lba_error's
__device__ __forceinline__ void lba_reduce_dev(const LbaDev& D, const WinCtl* __restrict__ ctl, WinOut* __restrict__ out, int w,
                                               int fl, bool per_point, double* s_red) {
  double v[3] = {0, 0, 0};
  if ((fl & LBA_BEGIN) && D.np > 0)
    for (int i = threadIdx.x; i < (D.n_obs + 255) / 256; i += 256) v[0] += D.part0[i];
  if ((fl & LBA_TRIAL) && D.np > 0) {
    if (per_point)
      for (int i = threadIdx.x; i < (D.n_mp + 63) / 64; i += 256) v[1] += D.part_t[i];
    else
      for (int i = threadIdx.x; i < (D.n_obs + 255) / 256; i += 256) v[1] += D.part[i];
    for (int i = threadIdx.x; i < (D.n_mp + 63) / 64; i += 256) v[2] += D.part_m[i];
  }
  block_sum<3>(v, s_red, threadIdx.x);
  if (threadIdx.x == 0) {
    double g0 = 0, g1 = 0;  // inertial edges, fixed order
    for (int e = 0; e < D.n_imu; e++) g0 += D.gchi0[e], g1 += D.gchi[e];
    if (D.red) {  // visual parts go through the all-reduce, the inertial part is the same on every rank
      double* sc = D.red_sc;
      sc[0] = v[0], sc[1] = v[1], sc[2] = v[2];
      sc[3] = ctl[w].pad ? 1.0 : 0.0;  // this rank's stop request: summed with the others', every rank sees the same count
      out[w].chig0 = g0, out[w].chig = (fl & LBA_TRIAL) ? g1 : 0.0;
    } else
      out[w].chi0 = v[0] + g0, out[w].chi2 = v[1] + ((fl & LBA_TRIAL) ? g1 : 0.0), out[w].scale_l = v[2];
  }
}
__global__ void __launch_bounds__(256)
k_lba_reduce(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, WinOut* __restrict__ out, int tail) {
  __shared__ double s_red[4 * 3];
  const int w = blockIdx.x, fl = ctl[w].flags;
  if (!(fl & (LBA_TRIAL | LBA_BEGIN))) return;
  lba_reduce_dev(devs[w], ctl, out, w, fl, tail != 0, s_red);  // tail: the trial's chi2 comes from k_lba_tail's partials
}

// 6x3 block Jp^T (rho' Omega) Jx of one active edge (multi-camera rigs: a key frame can see a point in
// several cameras, the (key frame, point) block of BB is the sum over those edges)
__device__ __forceinline__ void lba_edge_B(const LbaDev& D, int i, const LbaKf& k, bool robust, double* B) {
  const vieo_lba_obs o = D.obs[i];
  const CamD& C = obs_cam(D, i);
  PoseXf X;
  kf_xf(C, k, X);
  double Xw[3];
  scaled_point(D, o.mp, Xw);
  const double sc = win_scale(D);
  double err[3], Pc[3];
  const double chi2 = lba_edge_error(C, X, o, Xw, err, Pc);
  const bool stereo = o.ur >= 0;
  double r0, r1 = 1.;
  if (robust) {
    const double dl = stereo ? D.dStereo : D.dMono;
    huber(chi2, dl, dl * dl, &r0, &r1);
  }
  double Jp[18], Jx[9];
  lba_jacobians(C, X, k.p, Xw, Pc, Jp, Jx);
  const double ww = r1 * (double)o.inv_sigma2;
  if (!stereo) Jx[6] = Jx[7] = Jx[8] = 0;
  if (D.scale_opt)
    for (int t = 0; t < 9; t++) Jx[t] *= sc;  // J_Xh = s (Jproj Rcw)
#pragma unroll
  for (int q = 0; q < 6; q++)
#pragma unroll
    for (int b = 0; b < 3; b++) B[q * 3 + b] = Jp[q] * ww * Jx[b] + Jp[6 + q] * ww * Jx[3 + b] + Jp[12 + q] * ww * Jx[6 + b];
}

// ---- buildSystem (block_solver.hpp:451-520 with EdgeReprojectPR[Stereo]::linearizeOplus).
// blocks [0, gm): four lanes per point over its (contiguous) observations -> H_ll, b_l;
// blocks [gm, gm + free key frames): one workgroup per key frame over its edge list -> H_pp, b_p and the
// rows of BB = Jp^T W Jx it owns.  Every sum has a fixed order.
// MULTICAM: windows with several (distorted) cameras per key frame; the single rectified camera keeps the
// lean instantiation.  SCALE: a batch with a scale-vertex window (EdgeReprojectPRS[Stereo]): the point half also forms
// the point's entry of the scale row of BB (sum over its edges of Js^T W Jx) and its terms of H_ss / b_s, the key-frame
// half H_ps = sum Jp^T W Js.
// The two halves are two launches (KFHALF): the point half needs 162 registers, the key-frame half 254, and in one
// kernel the point half's 64 % of the workgroups ran at the key-frame half's two wavefronts per SIMD.
// The key-frame half takes a key frame's edge list in chunks of kBuildChunk edges, one workgroup per chunk (a rig key
// frame has thousands of edges -- 4 cameras x 1500 features: one workgroup per key frame was 1.08 ms of a trial's 1.4);
// a chunk's sums go to chunk_part, the workgroup that arrives last at the key frame's counter adds the chunks in chunk
// order.  A key frame of one chunk (<= kBuildChunk edges) writes its sums directly, as before.
// The one-launch instance (HALF = 2, calls of a few windows) also carries the generic (inertial / encoder) edges'
// linearisation, one workgroup per edge behind the gk chunk workgroups (k_lba_generic(0) otherwise: 44 us of single-lane
// chains that then run beside the chunks instead of behind them).
__device__ __forceinline__ void lba_generic_dev(const LbaDev& D, int e, int lane, int mode);
// HALF = 0: the point half, 1: the key-frame half (+ the generic edges), 2: both in one launch -- gp point workgroups, then
// the key-frame half's (calls of a few windows: the register-rich instance's occupancy does not matter there, a launch
// less does).
template <bool MULTICAM, bool SCALE, int HALF>
__global__ void __launch_bounds__(256, 2)
k_lba_build(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, int gk, int gp) {
  int bx = blockIdx.x;
  __shared__ double s_red[4 * 27];
  const int w = blockIdx.y, fl = ctl[w].flags;
  if (!(fl & LBA_BUILD)) return;
  const LbaDev& D = devs[w];
  const bool point_half = HALF == 0 || (HALF == 2 && bx < gp);
  if (HALF == 2 && !point_half) bx -= gp;
  if (HALF == 2 && !point_half && bx >= gk) {  // (this instance only: the generic path's 1.7 KB of scratch per lane would
    const int e = bx - gk;                     //  cost the batched key-frame half its occupancy: 3.2 -> 15.6 ms per step)
    if (e < D.n_imu && threadIdx.x < 64) lba_generic_dev(D, e, threadIdx.x, 0);
    return;
  }
  if (D.np == 0) return;
  const bool robust = fl & LBA_ROBUST;
  if (point_half) {
    if (bx * 64 >= D.n_mp) return;
    const int m = bx * 64 + (threadIdx.x >> 2), sub = threadIdx.x & 3;  // 4 lanes per point
    const bool act = m < D.n_mp && D.mp_act[m];
    double mx = 0;
    double acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++) acc[t] = 0;
    double asc[5] = {0, 0, 0, 0, 0};  // SCALE: Bs[3], H_ss, b_s of the point
    const bool scl = SCALE && D.scale_opt;
    const double sc = scl ? D.scl[0] : 1.0;
    if (act) {
      const double Xh[3] = {D.X[3 * (size_t)m], D.X[3 * (size_t)m + 1], D.X[3 * (size_t)m + 2]};
      const double Xw[3] = {Xh[0] * sc, Xh[1] * sc, Xh[2] * sc};
      const int first = D.mp_first[m], cnt = D.mp_count[m];
      // A lane's edges four at a time: their records and level bytes first, then the four key-frame poses, then the
      // arithmetic -- three dependent round trips per FOUR edges instead of per edge (level -> record -> key frame).
      // Same edges in the same order per lane, so the sums are bit-identical.
      for (int j0 = sub; j0 < cnt; j0 += 16) {
        vieo_lba_obs ou[4];
        unsigned char lv[4], oc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = first + min(j0 + 4 * u, cnt - 1);
          ou[u] = D.obs[i], lv[u] = D.level[i];
          oc[u] = (MULTICAM && D.n_cams) ? D.ocam[i] : (unsigned char)0;
        }
        double kp[4][7];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const LbaKf& K = D.kf[ou[u].kf];
          kp[u][0] = K.p[0], kp[u][1] = K.p[1], kp[u][2] = K.p[2];
          kp[u][3] = K.qw, kp[u][4] = K.qx, kp[u][5] = K.qy, kp[u][6] = K.qz;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
        if (j0 + 4 * u >= cnt || lv[u]) continue;
        const vieo_lba_obs o = ou[u];
        PoseXf X;
        const CamD& C = (MULTICAM && D.n_cams) ? D.cams[oc[u]] : D.cam;
        {
          Est e;
          e.p[0] = kp[u][0], e.p[1] = kp[u][1], e.p[2] = kp[u][2];
          e.qw = kp[u][3], e.qx = kp[u][4], e.qy = kp[u][5], e.qz = kp[u][6];
          make_xf(C, e, X);
        }
        double err[3], Pc[3];
        const double chi2 = lba_edge_error(C, X, o, Xw, err, Pc);
        const bool stereo = o.ur >= 0;
        double r0, r1 = 1.;
        if (robust) {
          const double dl = stereo ? D.dStereo : D.dMono;
          huber(chi2, dl, dl * dl, &r0, &r1);
        }
        const double invz = 1 / Pc[2], invz2 = invz * invz;
        double J[9], Jx[9], Jc[6], uv[2];
        cam_project(C, Pc, uv, Jc);
        J[0] = -Jc[0], J[1] = -Jc[1], J[2] = -Jc[2];
        J[3] = -Jc[3], J[4] = -Jc[4], J[5] = -Jc[5];
        J[6] = J[0], J[7] = J[1], J[8] = J[2] - C.bf * invz2;
        for (int r = 0; r < 3; r++)
          for (int q = 0; q < 3; q++)
            Jx[r * 3 + q] = J[r * 3] * X.Rcw[q] + J[r * 3 + 1] * X.Rcw[3 + q] + J[r * 3 + 2] * X.Rcw[6 + q];
        const double info = (double)o.inv_sigma2, ww = r1 * info;
        if (!stereo) Jx[6] = Jx[7] = Jx[8] = 0;  // monocular edge: no third row (err[2] is 0 already)
        if (scl) {  // _jacobianOplus[scale] = _jacobianOplus[0] * Ph_unscale, then _jacobianOplus[0] *= scale
          double Js[3];
#pragma unroll
          for (int r = 0; r < 3; r++) Js[r] = Jx[r * 3] * Xh[0] + Jx[r * 3 + 1] * Xh[1] + Jx[r * 3 + 2] * Xh[2];
#pragma unroll
          for (int q = 0; q < 9; q++) Jx[q] *= sc;
#pragma unroll
          for (int b = 0; b < 3; b++) asc[b] += Js[0] * ww * Jx[b] + Js[1] * ww * Jx[3 + b] + Js[2] * ww * Jx[6 + b];
          asc[3] += Js[0] * ww * Js[0] + Js[1] * ww * Js[1] + Js[2] * ww * Js[2];
          asc[4] += Js[0] * (-(info * err[0]) * r1) + Js[1] * (-(info * err[1]) * r1) + Js[2] * (-(info * err[2]) * r1);
        }
        int t = 0;
#pragma unroll
        for (int a = 0; a < 3; a++) {
#pragma unroll
          for (int b = a; b < 3; b++, t++)
            acc[t] += Jx[a] * ww * Jx[b] + Jx[3 + a] * ww * Jx[3 + b] + Jx[6 + a] * ww * Jx[6 + b];
          acc[6 + a] += Jx[a] * (-(info * err[0]) * r1) + Jx[3 + a] * (-(info * err[1]) * r1) +
                        Jx[6 + a] * (-(info * err[2]) * r1);
        }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 9; t++) acc[t] = quad_sum_f64(acc[t]);
    if (act && sub == 0) {
      double* H = D.Hll + 9 * (size_t)m;
      H[0] = acc[0], H[1] = acc[1], H[2] = acc[2];
      H[3] = acc[1], H[4] = acc[3], H[5] = acc[4];
      H[6] = acc[2], H[7] = acc[4], H[8] = acc[5];
      D.bl[3 * (size_t)m] = acc[6], D.bl[3 * (size_t)m + 1] = acc[7], D.bl[3 * (size_t)m + 2] = acc[8];
      mx = fmax(fabs(acc[0]), fmax(fabs(acc[3]), fabs(acc[5])));
    }
    if (scl) {
#pragma unroll
      for (int t = 0; t < 5; t++) asc[t] = quad_sum_f64(asc[t]);
      if (m < D.n_mp && sub == 0) {  // the scale's row of BB: zero for a point without an active edge
        double* B = D.Bs + 3 * (size_t)m;
        B[0] = asc[0], B[1] = asc[1], B[2] = asc[2];
      }
      double v[2] = {sub == 0 ? asc[3] : 0.0, sub == 0 ? asc[4] : 0.0};
      block_sum<2>(v, s_red + 8, threadIdx.x);
      if (threadIdx.x == 0) D.psc[2 * bx] = v[0], D.psc[2 * bx + 1] = v[1];
    }
    // landmark part of computeLambdaInit: block maximum
    mx = wave_max_f64(mx);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) D.pmax[bx] = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
    return;
  }
  if (bx >= D.n_chunks) return;
  const int a = D.chunk_kf[bx], chunk0 = D.chunk_first[a], nchunk = D.chunk_first[a + 1] - chunk0;
  const int kfi = D.kf_list[a];
  const LbaKf k = D.kf[kfi];
  const int first = D.kf_edge_first[kfi], cnt = D.kf_edge_first[kfi + 1] - first;
  // The sums are formed per run of kBuildChunk edges (a workgroup each) and the runs' sums added in run order: by the last
  // workgroup of the key frame to arrive in calls of a few windows, by k_lba_build_fold in batches (D.fold_kernel: a
  // device-scope fence per workgroup does not scale) -- the same association either way, a window's result does not depend
  // on what it is batched with.
  const int j_lo = (bx - chunk0) * kBuildChunk, j_hi = min(cnt, j_lo + kBuildChunk);
  double acc[27];
  double aps[6] = {0, 0, 0, 0, 0, 0};  // SCALE: H_ps = sum Jp^T W Js
#pragma unroll
  for (int t = 0; t < 27; t++) acc[t] = 0;
  const bool scl = SCALE && D.scale_opt;
  const double sc = scl ? D.scl[0] : 1.0;
  PoseXf X;
  kf_xf(D.cam, k, X);  // one camera: the transform is the same for all edges of the key frame
  const int* tab_a = D.tab + (size_t)a * D.n_mp;
  // A thread's edges two at a time: both list entries, then both records and level bytes, then both points are loaded
  // before the arithmetic (three dependent round trips per PAIR of edges instead of per edge; a key frame's ~600 edges are
  // 2.3 per thread).  Same edges in the same order per thread: the sums are bit-identical.
  for (int j0 = j_lo + threadIdx.x; j0 < j_hi; j0 += 512) {  // (one trip: a run is 2 edges per thread)
    int ii[2];
    vieo_lba_obs oo[2];
    unsigned char lvv[2];
    double Xp[2][3];
#pragma unroll
    for (int u = 0; u < 2; u++) ii[u] = D.kf_edge_idx[first + min(j0 + 256 * u, cnt - 1)];
#pragma unroll
    for (int u = 0; u < 2; u++) oo[u] = D.obs[ii[u]], lvv[u] = D.level[ii[u]];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const double* q = D.X + 3 * (size_t)oo[u].mp;
      Xp[u][0] = q[0], Xp[u][1] = q[1], Xp[u][2] = q[2];
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
    const int j = j0 + 256 * u;
    if (j >= j_hi) continue;
    const int i = ii[u];
    const vieo_lba_obs o = oo[u];
    const bool active = !lvv[u];
    double Bk[18];
#pragma unroll
    for (int t = 0; t < 18; t++) Bk[t] = 0;
    if (active) {
      const double* Xh = Xp[u];
      const double Xw[3] = {Xh[0] * sc, Xh[1] * sc, Xh[2] * sc};
      double err[3], Pc[3];
      const CamD& C = obs_cam(D, i);
      if (MULTICAM && D.n_cams) kf_xf(C, k, X);
      const double chi2 = lba_edge_error(C, X, o, Xw, err, Pc);
      const bool stereo = o.ur >= 0;
      double r0, r1 = 1.;
      if (robust) {
        const double dl = stereo ? D.dStereo : D.dMono;
        huber(chi2, dl, dl * dl, &r0, &r1);
      }
      double Jp[18], Jx[9];
      lba_jacobians(C, X, k.p, Xw, Pc, Jp, Jx);
      const double info = (double)o.inv_sigma2, ww = r1 * info;
      visual_accumulate(Jp, err, info, r1, stereo, acc);
      if (!stereo) Jx[6] = Jx[7] = Jx[8] = 0;
      if (scl) {
        double Js[3];
#pragma unroll
        for (int r = 0; r < 3; r++) Js[r] = Jx[r * 3] * Xh[0] + Jx[r * 3 + 1] * Xh[1] + Jx[r * 3 + 2] * Xh[2];
#pragma unroll
        for (int q = 0; q < 9; q++) Jx[q] *= sc;
#pragma unroll
        for (int q = 0; q < 6; q++) {
          double t = Jp[q] * ww * Js[0] + Jp[6 + q] * ww * Js[1];
          if (stereo) t += Jp[12 + q] * ww * Js[2];
          aps[q] += t;
        }
      }
#pragma unroll
      for (int q = 0; q < 6; q++)
#pragma unroll
        for (int b = 0; b < 3; b++)
          Bk[q * 3 + b] = Jp[q] * ww * Jx[b] + Jp[6 + q] * ww * Jx[3 + b] + Jp[12 + q] * ww * Jx[6 + b];
    }
    // the (key frame, point) block lives at the edge `tab` points to; with several cameras per key frame that edge
    // sums the run of the pair's edges (adjacent in the list) in list order
    bool write = active;
    if (MULTICAM && D.n_cams) {
      write = false;
      if (tab_a[o.mp] == i) {
        int js = j;
        while (js > 0 && D.obs[D.kf_edge_idx[first + js - 1]].mp == o.mp) js--;
        double Bsum[18];
#pragma unroll
        for (int t = 0; t < 18; t++) Bsum[t] = 0;
        bool first_term = true;
        for (int jj = js; jj < cnt; jj++) {
          const int i2 = D.kf_edge_idx[first + jj];
          if (D.obs[i2].mp != o.mp) break;
          if (D.level[i2]) continue;
          double B2[18];
          if (jj == j) {
#pragma unroll
            for (int t = 0; t < 18; t++) B2[t] = Bk[t];
          } else
            lba_edge_B(D, i2, k, robust, B2);
#pragma unroll
          for (int t = 0; t < 18; t++) Bsum[t] = first_term ? B2[t] : Bsum[t] + B2[t];
          first_term = false;
          write = true;
        }
#pragma unroll
        for (int t = 0; t < 18; t++) Bk[t] = Bsum[t];
      }
    }
    if (write) {
      double* B = D.CB + 18 * (size_t)i;
#pragma unroll
      for (int t = 0; t < 18; t++) B[t] = Bk[t];
    }
    }
  }
  block_sum<27>(acc, s_red, threadIdx.x);
  if (scl) block_sum<6>(aps, s_red, threadIdx.x);
  __shared__ int s_last;
  if (nchunk > 1) {  // this chunk's sums; the last workgroup of the key frame adds the chunks in order
    if (threadIdx.x < 33) {
      double v = 0;
#pragma unroll
      for (int t = 0; t < 27; t++)
        if ((int)threadIdx.x == t) v = acc[t];
#pragma unroll
      for (int t = 0; t < 6; t++)
        if ((int)threadIdx.x == 27 + t) v = aps[t];
      D.chunk_part[33 * (size_t)bx + threadIdx.x] = v;
    }
    if (D.fold_kernel) return;  // (batches: k_lba_build_fold adds the runs)
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      const int old = atomicAdd(&D.chunk_cnt[a], 1);
      s_last = old == nchunk - 1;
      if (s_last) D.chunk_cnt[a] = 0;  // (for the next launch: nobody else touches it any more)
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (threadIdx.x < 33) {
      double v = 0;
      for (int c = 0; c < nchunk; c++) v += D.chunk_part[33 * (size_t)(chunk0 + c) + threadIdx.x];
#pragma unroll
      for (int t = 0; t < 27; t++)
        if ((int)threadIdx.x == t) acc[t] = v;
#pragma unroll
      for (int t = 0; t < 6; t++)
        if ((int)threadIdx.x == 27 + t) aps[t] = v;
    }
  }
  if (threadIdx.x < 27) {
    double v = 0;
#pragma unroll
    for (int t = 0; t < 27; t++)  // select, not acc[threadIdx.x]: keeps the sums in registers
      if ((int)threadIdx.x == t) v = acc[t];
    if (threadIdx.x < 21) {
      int r = 0, t = threadIdx.x;
      while (t >= 6 - r) t -= 6 - r, r++;
      const int c = r + t;
      D.Hpp[36 * (size_t)a + r * 6 + c] = v;
      D.Hpp[36 * (size_t)a + c * 6 + r] = v;
    } else
      D.bp[6 * a + threadIdx.x - 21] = v;
  }
  if (scl && threadIdx.x >= 27 && threadIdx.x < 33) {
    double v = 0;
#pragma unroll
    for (int t = 0; t < 6; t++)
      if ((int)threadIdx.x == 27 + t) v = aps[t];
    D.sc_sys[6 * a + threadIdx.x - 27] = v;
  }
}

// The runs' sums of a key frame added in run order (batches; see k_lba_build): one wavefront per free key frame
__global__ void __launch_bounds__(64)
k_lba_build_fold(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl) {
  const int w = blockIdx.y;
  if (!(ctl[w].flags & LBA_BUILD)) return;
  const LbaDev& D = devs[w];
  const int a = blockIdx.x;
  if (D.np == 0 || a >= D.n_free) return;
  const int chunk0 = D.chunk_first[a], nchunk = D.chunk_first[a + 1] - chunk0;
  if (nchunk <= 1 || threadIdx.x >= 33) return;
  double v = 0;
  for (int c = 0; c < nchunk; c++) v += D.chunk_part[33 * (size_t)(chunk0 + c) + threadIdx.x];
  if (threadIdx.x < 21) {
    int r = 0, t = threadIdx.x;
    while (t >= 6 - r) t -= 6 - r, r++;
    const int c = r + t;
    D.Hpp[36 * (size_t)a + r * 6 + c] = v;
    D.Hpp[36 * (size_t)a + c * 6 + r] = v;
  } else if (threadIdx.x < 27)
    D.bp[6 * a + threadIdx.x - 21] = v;
  else if (D.scale_opt)
    D.sc_sys[6 * a + threadIdx.x - 27] = v;
}

// H_ss, b_s of the scale vertex: the per-block partials of k_lba_build in fixed order (one workgroup per window)
__global__ void __launch_bounds__(256)
k_lba_scale_fold(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl) {
  __shared__ double s_red[4 * 2];
  const int w = blockIdx.x;
  if (!(ctl[w].flags & LBA_BUILD)) return;
  const LbaDev& D = devs[w];
  if (!D.scale_opt || D.np == 0) return;
  double v[2] = {0, 0};
  for (int i = threadIdx.x; i < (D.n_mp + 63) / 64; i += 256) v[0] += D.psc[2 * i], v[1] += D.psc[2 * i + 1];
  block_sum<2>(v, s_red, threadIdx.x);
  if (threadIdx.x == 0) D.sc_sys[6 * D.n_free] = v[0], D.sc_sys[6 * D.n_free + 1] = v[1];
}

// computeLambdaInit: tau * max |diagonal| over the pose and landmark blocks (one workgroup per window)
__global__ void __launch_bounds__(256)
k_lba_lambda(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, WinOut* __restrict__ out) {
  __shared__ double s_m[256];
  const int w = blockIdx.x;
  if (!(ctl[w].flags & LBA_BEGIN)) return;
  const LbaDev& D = devs[w];
  double mx = 0;
  if (D.scale_opt && threadIdx.x == 0) mx = fabs(D.sc_sys[6 * D.n_free]);  // H_ss
  if (D.pd == 6 && D.n_imu == 0) {
    for (int j = threadIdx.x; j < 6 * D.n_free; j += 256) mx = fmax(mx, fabs(D.Hpp[36 * (size_t)(j / 6) + 7 * (j % 6)]));
  } else {  // PR + V + Bias vertices: visual block + the inertial edges' diagonal (as k_lba_assemble adds them)
    for (int j = threadIdx.x; j < D.pd * D.n_free; j += 256) {
      const int pd = D.pd, a = j / pd, ra = j - a * pd, ka = D.kf_list[a];
      const int ein = D.kf_in[ka], eout = D.kf_out[ka];
      double v = ra < 6 ? D.Hpp[36 * (size_t)a + 7 * ra] : 0.0;
      if (ein >= 0) v += D.Ae[930 * (size_t)ein + (15 + ra) * 30 + 15 + ra];
      if (eout >= 0) v += D.Ae[930 * (size_t)eout + ra * 30 + ra];
      mx = fmax(mx, fabs(v));
    }
  }
  for (int b = threadIdx.x; b < (D.n_mp + 63) / 64; b += 256) mx = fmax(mx, D.pmax[b]);
  s_m[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) s_m[threadIdx.x] = fmax(s_m[threadIdx.x], s_m[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[w].lambda = 1e-5 * s_m[0];
}

// ---- Schur complement GEMM.  S = (BB D^-1) [BB; bl]^T, S is np x (np + 1), tiled 64 x 64 (upper
// block-tiles only), K = 3 x points cut into chunks of 16 points; grid x = block-tile * ksplit + split.
typedef double double4_t __attribute__((ext_vector_type(4)));
static const int kChunkLm = 16, kLd = 3 * kChunkLm + 2;  // +2: conflict-free b64 fragment reads
#ifndef VIEO_SCHUR_AB
#define VIEO_SCHUR_AB 0  // timing experiments only (wrong results): 1 no MFMAs, 2 no LDS staging, 4 no block loads
#endif

// Which 16-landmark chunks a 64-row tile of BB touches (from `tab`, rebuilt at every optimize()): the Schur
// GEMM skips the all-zero ones -- a map of hundreds of key frames is block-sparse, a local window is dense.
__global__ void __launch_bounds__(256)
k_lba_occ(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl) {
  const int w = blockIdx.y;
  if (!(ctl[w].flags & LBA_BEGIN)) return;
  const LbaDev& D = devs[w];
  if (!D.use_occ) return;
  const int np = D.npv, CB = (np + 64) >> 6, nchunks = (D.n_mp + kChunkLm - 1) / kChunkLm;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (np == 0 || e >= CB * nchunks) return;
  const int bi = e / nchunks, ch = e - bi * nchunks;
  const int a0 = bi * 64 / 6, a1 = min((bi * 64 + 63) / 6, D.n_free - 1);
  int any = D.scale_opt && (np - 1) >= bi * 64 && (np - 1) < bi * 64 + 64;  // the scale's row sees every point
  for (int a = a0; a <= a1 && !any; a++)
    for (int j = 0; j < kChunkLm; j++) {
      const int m = ch * kChunkLm + j;
      if (m < D.n_mp && D.tab[(size_t)a * D.n_mp + m] >= 0) {
        any = 1;
        break;
      }
    }
  D.occ[e] = (unsigned char)any;
}

// Two instances: the diagonal tiles (bi == bj: one register set of blocks, three wavefronts per SIMD = the three
// workgroups a CU's LDS holds) and the off-diagonal ones (two register sets; at 168 registers they spilled the blocks
// they had just loaded, i.e. waited for them at once: two wavefronts per SIMD).  An ordinary window is one diagonal tile.
template <bool OFFDIAG>
__global__ void __launch_bounds__(256, OFFDIAG ? 2 : 3)
k_lba_schur(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, const WinOut* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) double sT[64 * kLd];
  __shared__ __attribute__((aligned(16))) double sB[64 * kLd];
  __shared__ double sDi[2][kChunkLm * 9];
  const int w = blockIdx.y;
  if (!(ctl[w].flags & LBA_TRIAL)) return;
  const LbaDev& D = devs[w];
  const int np = D.npv;  // the landmarks touch the PR blocks only
  if (np == 0) return;
  const int RB = (np + 63) >> 6, CB = (np + 64) >> 6;  // CB covers the extra column bl
  const int ksplit = D.ksplit;
  int bt = blockIdx.x / ksplit;
  const int split = blockIdx.x % ksplit;
  int bi = 0, bj;
  if (OFFDIAG) {  // tiles (bi, bj), bi < bj < CB, row by row
    while (bi < RB && bt >= CB - bi - 1) bt -= CB - bi - 1, bi++;
    if (bi >= RB) return;
    bj = bi + 1 + bt;
  } else {
    if (bt >= RB) return;
    bi = bj = bt;
  }
  const int nchunks = (D.n_mp + kChunkLm - 1) / kChunkLm, cps = (nchunks + ksplit - 1) / ksplit;
  const int c0 = split * cps, c1 = min(nchunks, c0 + cps);
  if (c0 >= c1) return;  // k_lba_assemble uses the same split arithmetic
  const double lambda = win_lambda(ctl[w], out[w]);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_mp = D.n_mp;
  double4_t acc[4];
#pragma unroll
  for (int q = 0; q < 4; q++) acc[q] = (double4_t){0, 0, 0, 0};
  const unsigned char* occ_i = D.occ + (size_t)bi * nchunks;
  const unsigned char* occ_j = D.occ + (size_t)bj * nchunks;
  const bool has_bl = np >= bj * 64 && np < bj * 64 + 64;  // this column tile carries b_l: dense
  // the chunks this workgroup works on: a zero factor adds nothing (workgroup-uniform)
  // (local windows: every chunk -- the flag bytes would be one more dependent global load per chunk)
  const bool use_occ = D.use_occ != 0;
  auto next_chunk = [&](int ch) {
    if (use_occ)
      while (ch < c1 && (!occ_i[ch] || (!has_bl && !occ_j[ch]))) ch++;
    return ch;
  };
  // Staging.  The operand tiles T = (BB D^-1)[rows of tile bi] and B = [BB; b_l][rows of tile bj], 64 x 48 per chunk
  // of 16 landmarks, exist only in LDS.  One thread per (key-frame slot of the tile, landmark of the chunk) pair --
  // 12 slots cover the 64 rows whatever 6 a mod 64 is: wavefronts 0..2 -- looks the pair's edge up in `tab` and writes
  // its 6 x 3 block or zeros: nothing to clear, two barriers per chunk.  The slot after the last free key frame carries
  // the scale vertex's row (bScaleOpt) and, on the B side, the extra column b_l.  Wavefront 3 has no pairs: its first
  // 16 lanes form (H_ll + lambda I)^-1 of the NEXT chunk's landmarks while the others stage the current one.
  // Software pipeline: blocks of chunk c + 1 in registers, `tab` entries of chunk c + 2, H_ll of chunk c + 2.
  // The loop has NO load whose address or predicate depends on another load of the same iteration (s_memtime probes,
  // tools/probe_schur.sh: such a chain -- activity byte -> H_ll, tab entry -> "edge present?" branch -> block -- put a
  // full vmcnt(0) wait, i.e. the whole HBM latency, into every chunk: 16.5 k cycles per chunk against 2.9 k of MFMAs).
  // Absent pairs read a zero block (CB[n_obs], cleared by k_lba_zero) instead of branching, indices are clamped
  // instead of guarded, predicates are applied to the VALUES.
  typedef double dbl2_t __attribute__((ext_vector_type(2)));
  typedef const dbl2_t __attribute__((address_space(1)))* gd2p;
  typedef const double __attribute__((address_space(1)))* gdp;
  typedef const int __attribute__((address_space(1)))* gip;
  typedef const unsigned char __attribute__((address_space(1)))* gbp;
  const gd2p gCB = (gd2p)D.CB;
  const gdp gBs = (gdp)D.Bs, gbl = (gdp)D.bl, gHll = (gdp)D.Hll;
  const gip gtab = (gip)D.tab;
  const gbp gact = (gbp)D.mp_act;
  const int ps = tid >> 4, pj = tid & 15;
  const int nfree = D.n_free, n_obs = D.n_obs;
  const bool sc_opt = D.scale_opt != 0;
  constexpr bool offdiag = OFFDIAG;
  const int aT = (64 * bi) / 6 + ps, aB = (64 * bj) / 6 + ps;
  const int rT = 6 * aT - 64 * bi, rB = 6 * aB - 64 * bj;  // tile row of the pair's first row (-5 .. 66)
  const int aTc = min(aT, max(nfree, 1) - 1), aBc = min(aB, max(nfree, 1) - 1);
  // does this wavefront hold the slot after the last key frame (per side)?  wave-uniform
  const bool spT = wv < 3 && nfree >= (64 * bi) / 6 + 4 * wv && nfree < (64 * bi) / 6 + 4 * wv + 4;
  const bool spB = wv < 3 && nfree >= (64 * bj) / 6 + 4 * wv && nfree < (64 * bj) / 6 + 4 * wv + 4;
  auto tab_of = [&](int a, int ac, int ch) -> int {  // unconditional load, the predicate picks the value
    const int m = ch * kChunkLm + pj;
    const int t = gtab[(size_t)ac * n_mp + min(m, n_mp - 1)];
    return (ch < c1 && a < nfree && m < n_mp) ? t : -1;
  };
  auto load_blk = [&](int e, double* v) {  // e < 0: the zero block
    const gd2p p = gCB + 9 * (size_t)min((unsigned)e, (unsigned)n_obs);
#pragma unroll
    for (int t = 0; t < 9; t++) {
      const dbl2_t x = p[t];
      v[2 * t] = x[0], v[2 * t + 1] = x[1];
    }
  };
  // the slot after the last key frame: the scale vertex's row and b_l of the landmark (zeros for the other threads)
  auto load_special = [&](int a, int ch, bool with_bl, double* sp, double* blv) {
    const int m = ch * kChunkLm + pj, mc = min(m, n_mp - 1);
    const bool mine = a == nfree && m < n_mp && ch < c1;
    double s0 = 0, s1 = 0, s2 = 0;
    if (sc_opt) s0 = gBs[3 * (size_t)mc], s1 = gBs[3 * (size_t)mc + 1], s2 = gBs[3 * (size_t)mc + 2];
    sp[0] = mine ? s0 : 0.0, sp[1] = mine ? s1 : 0.0, sp[2] = mine ? s2 : 0.0;
    if (with_bl) {
      const double b0 = gbl[3 * (size_t)mc], b1 = gbl[3 * (size_t)mc + 1], b2 = gbl[3 * (size_t)mc + 2];
      const bool on = mine && gact[mc] != 0;
      blv[0] = on ? b0 : 0.0, blv[1] = on ? b1 : 0.0, blv[2] = on ? b2 : 0.0;
    }
  };
  double ta[18], tb[18], spT3[3] = {0, 0, 0}, spB3[3] = {0, 0, 0}, blv[3] = {0, 0, 0}, hl[9];
  unsigned char hl_on = 0;
  auto load_hll = [&](int ch) {  // wavefront 3, lanes 0..15 (the others load the same clamped landmark)
    const int m = ch * kChunkLm + (lane & 15), mc = min(m, n_mp - 1);
#pragma unroll
    for (int t = 0; t < 9; t++) hl[t] = gHll[9 * (size_t)mc + t];
    hl_on = (ch < c1 && m < n_mp) ? gact[mc] : (unsigned char)0;
  };
  auto store_dinv = [&](int buf) {  // wavefront 3, lanes 0..15
    double Di[9];
    landmark_dinv(hl, lambda, Di);
    if (lane < kChunkLm) {
#pragma unroll
      for (int t = 0; t < 9; t++) sDi[buf][lane * 9 + t] = hl_on ? Di[t] : 0.0;
    }
  };
  // rows of the pair -> LDS (a slot that straddles the tile's edge has rows outside it)
  auto stage_T = [&](const double* v, const double* Di) {
    const double d0 = Di[0], d1 = Di[1], d2 = Di[2], d3 = Di[3], d4 = Di[4], d5 = Di[5], d6 = Di[6], d7 = Di[7], d8 = Di[8];
#pragma unroll
    for (int q = 0; q < 6; q++) {
      const unsigned r = (unsigned)(rT + q);
      if (r < 64u) {
        double b0 = v[3 * q], b1 = v[3 * q + 1], b2 = v[3 * q + 2];
        if (q == 0) b0 += spT3[0], b1 += spT3[1], b2 += spT3[2];  // the scale vertex's row (zeros elsewhere)
        double* d = sT + r * kLd + 3 * pj;
        d[0] = __builtin_fma(b2, d6, __builtin_fma(b1, d3, b0 * d0));
        d[1] = __builtin_fma(b2, d7, __builtin_fma(b1, d4, b0 * d1));
        d[2] = __builtin_fma(b2, d8, __builtin_fma(b1, d5, b0 * d2));
      }
    }
  };
  auto stage_B = [&](const double* v, const double* sp) {
#pragma unroll
    for (int q = 0; q < 6; q++) {
      const unsigned r = (unsigned)(rB + q);
      if (r < 64u) {
        double* d = sB + r * kLd + 3 * pj;
        double b0 = v[3 * q], b1 = v[3 * q + 1], b2 = v[3 * q + 2];
        if (q == 0) b0 += sp[0], b1 += sp[1], b2 += sp[2];
        if (q < 2 && (q == 1) == sc_opt) b0 += blv[0], b1 += blv[1], b2 += blv[2];  // (uniform) the b_l column's row
        d[0] = b0, d[1] = b1, d[2] = b2;
      }
    }
  };
#ifdef VIEO_SCHUR_PROBE
  long long tp[6] = {0, 0, 0, 0, 0, 0}, t0 = __builtin_readcyclecounter(), t1;
#define SCHUR_TICK(i) t1 = __builtin_readcyclecounter(), tp[i] += t1 - t0, t0 = t1;
#else
#define SCHUR_TICK(i)
#endif
  int ch = next_chunk(c0), buf = 0;
  int nx = ch < c1 ? next_chunk(ch + 1) : c1;
  int nx2 = nx < c1 ? next_chunk(nx + 1) : c1;
  int eT = -1, eB = -1;
  if (wv < 3) {
    load_blk(tab_of(aT, aTc, ch), ta);
    if (offdiag) load_blk(tab_of(aB, aBc, ch), tb);
    if (spT) load_special(aT, ch, !offdiag, spT3, blv);
    if (offdiag && spB) load_special(aB, ch, true, spB3, blv);
    eT = tab_of(aT, aTc, nx);
    if (offdiag) eB = tab_of(aB, aBc, nx);
  } else {
    load_hll(ch);
    store_dinv(0);
    load_hll(nx);
  }
  while (ch < c1) {
    __syncthreads();  // the previous chunk's fragments have been read; sDi[buf] is complete
    SCHUR_TICK(0)
    if (wv < 3) {
      if (!(VIEO_SCHUR_AB & 2)) {
        stage_T(ta, sDi[buf] + pj * 9);
        if (offdiag)
          stage_B(tb, spB3);
        else
          stage_B(ta, spT3);
      }
    } else {
      store_dinv(buf ^ 1);  // chunk nx, from the H_ll loaded an iteration ago
      load_hll(nx2);
    }
    SCHUR_TICK(1)
    __syncthreads();
    SCHUR_TICK(2)
    const int nx3 = nx2 < c1 ? next_chunk(nx2 + 1) : c1;
    if (wv < 3 && nx < c1) {
      load_blk(eT, ta);
      if (offdiag) load_blk(eB, tb);
      if (spT) load_special(aT, nx, !offdiag, spT3, blv);
      if (offdiag && spB) load_special(aB, nx, true, spB3, blv);
      eT = tab_of(aT, aTc, nx2);
      if (offdiag) eB = tab_of(aB, aBc, nx2);
    }
    SCHUR_TICK(3)
    {
      const double* pa = sT + (wv * 16 + (lane & 15)) * kLd + (lane >> 4);
      const double* pb = sB + (lane & 15) * kLd + (lane >> 4);
#if !(VIEO_SCHUR_AB & 1)
#pragma unroll
      for (int ks = 0; ks < 3 * kChunkLm / 4; ks++) {
        const double av = pa[ks * 4];
#pragma unroll
        for (int q = 0; q < 4; q++)
          acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, pb[q * 16 * kLd + ks * 4], acc[q], 0, 0, 0);
      }
#else
      acc[0][0] += pa[0] + pb[0];
#endif
    }
    SCHUR_TICK(4)
    buf ^= 1, ch = nx, nx = nx2, nx2 = nx3;
  }
#ifdef VIEO_SCHUR_PROBE
  if ((blockIdx.x == 3 || blockIdx.x == 40) && (blockIdx.y == 0 || blockIdx.y == 3) && (tid == 0 || tid == 64 || tid == 200))
    printf("schur probe w %d bx %d tid %d tile (%d,%d) chunks %d: wait_A %lld stage %lld wait_B %lld issue_loads %lld mfma %lld\n",
           w, (int)blockIdx.x, tid, bi, bj, c1 - c0, tp[0], tp[1], tp[2], tp[3], tp[4]);
#endif
  // f64 C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
  double* S = D.Sp + (size_t)split * D.sp_stride;
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int r = 0; r < 4; r++)
      S[(size_t)(bi * 64 + wv * 16 + (lane >> 4) + 4 * r) * D.ldS + bj * 64 + q * 16 + (lane & 15)] = acc[q][r];
}

// Landmark-sharded window: this rank's part of everything the ranks have to sum -- the
// Schur product over its landmarks (K-split partials folded in fixed order), H_pp and b_p of its edges
// -- packed contiguously for ONE all-reduce.
__global__ void __launch_bounds__(256)
k_lba_pack(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl) {
  const int w = blockIdx.y;
  if (!(ctl[w].flags & LBA_TRIAL)) return;
  const LbaDev& D = devs[w];
  if (!D.red) return;
  const int npv = D.npv, nS = npv * (npv + 1), nH = 36 * D.n_free, nb = 6 * D.n_free;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= nS + nH + nb + (D.scale_opt ? nb + 2 : 0)) return;
  if (e < nS) {
    const int ksplit = D.ksplit;
    const int nchunks = (D.n_mp + kChunkLm - 1) / kChunkLm, cps = (nchunks + ksplit - 1) / ksplit;
    const int ns = (nchunks + cps - 1) / cps;
    const int r = e / (npv + 1), c = e - r * (npv + 1);
    const int rr = c < npv ? min(r, c) : r, cc = c < npv ? max(r, c) : npv;
    double s = 0;
    for (int k = 0; k < ns; k++) s += D.Sp[(size_t)k * D.sp_stride + (size_t)rr * D.ldS + cc];
    D.red[e] = s;
  } else if (e < nS + nH)
    D.red[e] = D.Hpp[e - nS];
  else if (e < nS + nH + nb)
    D.red[e] = D.bp[e - nS - nH];
  else
    D.red[e] = D.sc_sys[e - nS - nH - nb];  // H_ps, H_ss, b_s of the scale vertex
}

// Reduced pose system.  PR x PR entries: Hpp + lambda I - S (both triangles from the upper block-tiles of
// the Schur partials); visual-inertial windows add the inertial edges' 30x30 blocks, gathered per entry
// (a key frame has at most one inertial edge in and one out).  bs = b - S[:, npv], bfull = b.
__global__ void __launch_bounds__(256)
k_lba_assemble(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, const WinOut* __restrict__ out,
               const int* __restrict__ wins) {
  const int w = wins[blockIdx.y];  // the windows of one solver class: the grid is sized for that class's systems
  if (!(ctl[w].flags & LBA_TRIAL)) return;
  const LbaDev& D = devs[w];
  const int np = D.np, npv = D.npv, pd = D.pd;
  const double lambda = win_lambda(ctl[w], out[w]);
  const int ksplit = D.ksplit;
  const int nchunks = (D.n_mp + kChunkLm - 1) / kChunkLm, cps = (nchunks + ksplit - 1) / ksplit;
  const int ns = (nchunks + cps - 1) / cps;
  // (one launch per solver class, over that class's windows only: sizing one grid for the largest window of a mixed
  // batch launched 112 k workgroups of which the ordinary windows' 90 % returned at once, a small fixed grid left the
  // bLarge windows' threads nine dependent gathers each.  The solve kernels read the 16 x 16 blocks on and below the
  // diagonal only.)
  for (int e = blockIdx.x * 256 + threadIdx.x; e < np * np; e += gridDim.x * 256) {
  const int r = e / np, c = e % np;
  if ((c >> 4) > (r >> 4)) continue;
  // the scale vertex (bScaleOpt) is the last row / column; its row of the visual system is the last one as well
  const bool rs = D.scale_opt && r == np - 1, cs = D.scale_opt && c == np - 1;
  const int a = r / pd, ra = r - a * pd, b = c / pd, cb = c - b * pd;
  const int vr = rs ? npv - 1 : (ra < 6 ? 6 * a + ra : -1), vc = cs ? npv - 1 : (cb < 6 ? 6 * b + cb : -1);
  double v = 0;
  const double* redS = D.red;  // sums over the K splits and over the ranks, see k_lba_pack
  const double* redH = D.red ? D.red + (size_t)npv * (npv + 1) : D.Hpp;
  const double* redb = D.red ? redH + 36 * (size_t)D.n_free : D.bp;
  const double* redsc = D.red ? redb + 6 * (size_t)D.n_free : D.sc_sys;
  if (vr >= 0 && vc >= 0) {
    const int rr = min(vr, vc), cc = max(vr, vc);
    double s = 0;
    if (redS)
      s = redS[(size_t)rr * (npv + 1) + cc];
    else
      for (int k = 0; k < ns; k++) s += D.Sp[(size_t)k * D.sp_stride + (size_t)rr * D.ldS + cc];
    v = -s;
    if (rs && cs)
      v += redsc[6 * D.n_free];  // H_ss
    else if (rs)
      v += redsc[6 * b + cb];  // H_ps^T
    else if (cs)
      v += redsc[6 * a + ra];
    else if (a == b)
      v += redH[36 * (size_t)a + ra * 6 + cb];
  }
  int ein = -1, eout = -1;
  if ((pd == 15 || D.n_imu > 0) && !rs) {  // pair edges: inertial (+ encoder) of a 15-dim window, encoder only of a 6-dim one
    const int ka = D.kf_list[a];
    ein = D.kf_in[ka], eout = D.kf_out[ka];
    if (!cs) {
      const int kb = D.kf_list[b];
      if (a == b) {
        if (ein >= 0) v += D.Ae[930 * (size_t)ein + (15 + ra) * 30 + 15 + cb];
        if (eout >= 0) v += D.Ae[930 * (size_t)eout + ra * 30 + cb];
      } else {
        if (eout >= 0 && D.imu[eout].j == kb) v += D.Ae[930 * (size_t)eout + ra * 30 + 15 + cb];
        if (ein >= 0 && D.imu[ein].i == kb) v += D.Ae[930 * (size_t)ein + (15 + ra) * 30 + cb];
      }
    }
  }
  if (r == c) v += lambda;
  D.Hs[e] = v;
  if (c == 0) {
    double g = 0, t = 0;
    if (vr >= 0) {
      g = rs ? redsc[6 * D.n_free + 1] : redb[6 * a + ra];
      if (redS)
        t = redS[(size_t)vr * (npv + 1) + npv];
      else
        for (int k = 0; k < ns; k++) t += D.Sp[(size_t)k * D.sp_stride + (size_t)vr * D.ldS + npv];
    }
    if (ein >= 0) g += D.Ae[930 * (size_t)ein + 900 + 15 + ra];
    if (eout >= 0) g += D.Ae[930 * (size_t)eout + 900 + ra];
    D.bfull[r] = g;
    D.bs[r] = g - t;
  }
  }
}

// ---- inertial edges of a visual-inertial window: one wavefront per key-frame pair.
// mode 0: linearise at the current state (30x30 block J^T (rho' Omega) J and gradient, both edges of the
// pair) and robust chi2 -> gchi0; mode 1: robust chi2 after a trial -> gchi.
// (the body, for one wavefront: k_lba_generic's workgroups and the pair workgroups of k_lba_tail)
__device__ __forceinline__ void lba_generic_dev(const LbaDev& D, int e, int lane, int mode) {
  __shared__ double sJ[9 * 30], sT[9 * 30], sErr[9 + 6], sWe[9], sRho[3];
  __shared__ double sJE[6 * 30], sTE[6 * 30], sWeE[6];  // encoder edge: J in the local order, (rho' Info) J, Info e
  // the edge's record staged in LDS by the whole wavefront: lane 0's chains below read it field by field, and every
  // dependent batch of those reads was a trip to L2
  static_assert(sizeof(LbaImu) % 8 == 0, "staged in 8-byte words");
  __shared__ __align__(16) double sE_store[sizeof(LbaImu) / 8];
  {
    const double* src = reinterpret_cast<const double*>(&D.imu[e]);
    for (int i = lane; i < (int)(sizeof(LbaImu) / 8); i += 64) sE_store[i] = src[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  const LbaImu& E = *reinterpret_cast<const LbaImu*>(sE_store);
  const double dI = (double)(float)sqrt(16.919), dB = (double)(float)sqrt(12.592);  // Optimizer.cc:219-222
  if (lane == 0) {
    NSd si, sj;
    const LbaKf &ki = D.kf[E.i], &kj = D.kf[E.j];
    for (int k = 0; k < 3; k++) {
      si.p[k] = ki.p[k], si.v[k] = ki.v[k], si.bg[k] = ki.bg[k], si.ba[k] = ki.ba[k];
      si.dbg[k] = ki.dbg[k], si.dba[k] = ki.dba[k];
      sj.p[k] = kj.p[k], sj.v[k] = kj.v[k], sj.bg[k] = kj.bg[k], sj.ba[k] = kj.ba[k];
      sj.dbg[k] = kj.dbg[k], sj.dba[k] = kj.dba[k];
    }
    si.qw = ki.qw, si.qx = ki.qx, si.qy = ki.qy, si.qz = ki.qz;
    sj.qw = kj.qw, sj.qx = kj.qx, sj.qy = kj.qy, sj.qz = kj.qz;
    double chi = 0, rI = 1.0, rB = 1.0;
    if (E.has_imu) {
      imu_error(E.M, D.gw, si, sj, sErr, 3);
      double c2 = 0;
      for (int a = 0; a < 9; a++) {
        double t = 0;
        for (int b = 0; b < 9; b++) t += E.InfoI[a * 9 + b] * sErr[b];
        sWe[a] = t;
        c2 += sErr[a] * t;
      }
      double r0 = c2;
      if (E.robust) huber(c2, dI, dI * dI, &r0, &rI);
      chi += r0;
    }
    for (int k = 0; k < 3; k++) {
      sErr[9 + k] = (sj.bg[k] + sj.dbg[k]) - (si.bg[k] + si.dbg[k]);
      sErr[12 + k] = (sj.ba[k] + sj.dba[k]) - (si.ba[k] + si.dba[k]);
    }
    {
      double c2 = 0;
      for (int k = 0; k < 3; k++) c2 += sErr[9 + k] * (E.infoBg * sErr[9 + k]);
      for (int k = 3; k < 6; k++) c2 += sErr[9 + k] * (E.infoBa * sErr[9 + k]);
      double r0 = c2;
      if (E.robust) huber(c2, dB, dB * dB, &r0, &rB);
      chi += r0;
    }
    double rE = 1.0;
    if (E.has_enc) {
      double errE[6], JEi[36], JEj[36];
      enc_edge_eval(si, sj, E.measE, D.qRbe, D.pbe, errE, mode == 0 ? JEi : nullptr, mode == 0 ? JEj : nullptr);
      double c2 = 0;
      for (int a = 0; a < 6; a++) {
        double t = 0;
        for (int b = 0; b < 6; b++) t += E.InfoE[a * 6 + b] * errE[b];
        sWeE[a] = t;
        c2 += errE[a] * t;
      }
      double r0 = c2;
      if (E.enc_robust) huber(c2, dB, dB * dB, &r0, &rE);  // sqrt(12.592), Optimizer.cc:343
      chi += r0;
      if (mode == 0)
        for (int a = 0; a < 6; a++) {
          for (int c = 0; c < 30; c++) sJE[a * 30 + c] = 0;
          for (int c = 0; c < 6; c++) sJE[a * 30 + c] = JEi[a * 6 + c], sJE[a * 30 + 15 + c] = JEj[a * 6 + c];
        }
    }
    (mode ? D.gchi : D.gchi0)[e] = chi;
    sRho[0] = rI, sRho[1] = rB, sRho[2] = rE;
    if (mode == 0 && E.has_imu) {
      // J (9 x 24, [PRV_j | PRV_i | Bias_i]) -> local order [i: PR V Bias | j: PR V Bias]
      double* J24 = sT;  // (LDS, free until the products below: as a local array it lived in scratch, 1.7 KB per lane)
      imu_linearize(E.M, D.gw, si, sj, sErr, J24, 3, 6);
      for (int a = 0; a < 9; a++) {
        for (int c = 0; c < 30; c++) sJ[a * 30 + c] = 0;
        for (int c = 0; c < 9; c++) sJ[a * 30 + 15 + c] = J24[a * 24 + c];       // state j: PR, V
        for (int c = 0; c < 9; c++) sJ[a * 30 + c] = J24[a * 24 + 9 + c];        // state i: PR, V
        for (int c = 0; c < 6; c++) sJ[a * 30 + 9 + c] = J24[a * 24 + 18 + c];   // Bias_i
      }
    }
  }
  if (mode) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  double* A = D.Ae + 930 * (size_t)e;
  const double rI = sRho[0], rB = sRho[1], rE = sRho[2];
  if (E.has_enc) {
    for (int t = lane; t < 180; t += 64) {
      const int a = t / 30, c = t - a * 30;
      double u = 0;
      for (int q = 0; q < 6; q++) u += (rE * E.InfoE[a * 6 + q]) * sJE[q * 30 + c];
      sTE[t] = u;
    }
  }
  if (E.has_imu) {
    for (int t = lane; t < 270; t += 64) {  // T = (rho' Info) J
      const int a = t / 30, c = t - a * 30;
      double u = 0;
      for (int q = 0; q < 9; q++) u += (rI * E.InfoI[a * 9 + q]) * sJ[q * 30 + c];
      sT[t] = u;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int t = lane; t < 900; t += 64) {
    const int c1 = t / 30, c2 = t - c1 * 30;
    double u = 0;
    if (E.has_imu)
      for (int a = 0; a < 9; a++) u += sJ[a * 30 + c1] * sT[a * 30 + c2];
    if (E.has_enc)
      for (int a = 0; a < 6; a++) u += sJE[a * 30 + c1] * sTE[a * 30 + c2];
    // bias edge: J_i = -I on rows/cols 9..14, J_j = +I on 24..29
    const int b1 = c1 >= 24 ? c1 - 24 : (c1 >= 9 && c1 < 15 ? c1 - 9 : -1);
    const int b2 = c2 >= 24 ? c2 - 24 : (c2 >= 9 && c2 < 15 ? c2 - 9 : -1);
    if (b1 >= 0 && b1 == b2) {
      const double wgt = (b1 < 3 ? E.infoBg : E.infoBa) * rB;
      u += ((c1 >= 24) == (c2 >= 24)) ? wgt : -wgt;
    }
    A[t] = u;
  }
  if (lane < 30) {
    double u = 0;
    if (E.has_imu)
      for (int a = 0; a < 9; a++) u += sJ[a * 30 + lane] * (-sWe[a] * rI);
    if (E.has_enc)
      for (int a = 0; a < 6; a++) u += sJE[a * 30 + lane] * (-sWeE[a] * rE);
    const int b1 = lane >= 24 ? lane - 24 : (lane >= 9 && lane < 15 ? lane - 9 : -1);
    if (b1 >= 0) {
      const double we = (b1 < 3 ? E.infoBg : E.infoBa) * sErr[9 + b1] * rB;
      u += lane >= 24 ? -we : we;
    }
    A[900 + lane] = u;
  }
}
__global__ void __launch_bounds__(64)
k_lba_generic(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, int mode) {
  const int w = blockIdx.y, fl = ctl[w].flags;
  if (!(fl & (mode ? LBA_TRIAL : LBA_BUILD))) return;
  const LbaDev& D = devs[w];
  if ((int)blockIdx.x >= D.n_imu) return;
  lba_generic_dev(D, blockIdx.x, threadIdx.x, mode);
}

// ---- dense LDL^T solve of the reduced system + pose update.
// The end of a solve, shared by the single-workgroup LDL^T and the tiled one: x -> D.xp, the pose part of
// computeScale(), push() + oplus on the free key frames.  y: the solution (LDS or global), 256 threads.
// NT: threads of the workgroup; the first 256 do the work (and sum in the same order whatever NT is).
template <int NT = 256>
__device__ __forceinline__ void lba_apply_step(const LbaDev& D, const double* y, double lambda, bool ok, WinOut& o,
                                               double* s_red, int tid) {
  const int n = D.np;
  double sp[1] = {0};
  if (tid < 256)
    for (int i = tid; i < n; i += 256) {
      D.xp[i] = y[i];
      sp[0] += y[i] * (lambda * y[i] + D.bfull[i]);  // pose part of computeScale()
    }
  if (NT == 256)
    block_sum<1>(sp, s_red, tid);
  else {
    sp[0] = wave_sum_d(sp[0]);
    __syncthreads();
    if ((tid & 63) == 0 && tid < 256) s_red[tid >> 6] = sp[0];
    __syncthreads();
    sp[0] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  }
  if (tid == 0) {
    o.ok = ok ? 1 : 0, o.scale_p = ok ? sp[0] : 0.0;
    if (D.scale_opt) D.scl[1] = D.scl[0], D.scl[0] += y[n - 1];  // push(), VertexScale::oplusImpl (g2otypes.h:310)
  }
  if (tid >= 256) return;
  // oplus on the free key frames (push() first): VertexNavStatePR (+ V, Bias in a visual-inertial window)
  for (int k = tid; k < D.n_kf; k += 256) {
    LbaKf kf = D.kf[k];
    if (kf.col < 0) continue;
    D.kf_bak[k] = kf;
    Est e;
    e.p[0] = kf.p[0], e.p[1] = kf.p[1], e.p[2] = kf.p[2];
    e.qw = kf.qw, e.qx = kf.qx, e.qy = kf.qy, e.qz = kf.qz;
    inc_small_pr(e, y + kf.col);
    kf.p[0] = e.p[0], kf.p[1] = e.p[1], kf.p[2] = e.p[2];
    kf.qw = e.qw, kf.qx = e.qx, kf.qy = e.qy, kf.qz = e.qz;
    if (D.pd == 15)
      for (int a = 0; a < 3; a++)
        kf.v[a] += y[kf.col + 6 + a], kf.dbg[a] += y[kf.col + 9 + a], kf.dba[a] += y[kf.col + 12 + a];
    D.kf[k] = kf;
  }
}

__device__ __forceinline__ double big_readlane(double v, int srclane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)b, srclane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), srclane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// block t of a lower triangle counted row by row -> (row, column)
struct TriMap {
  unsigned char i[64], j[64];
  constexpr TriMap() : i(), j() {
    int t = 0;
    for (int r = 0; r < 11 && t < 64; r++)
      for (int c = 0; c <= r && t < 64; c++, t++) i[t] = (unsigned char)r, j[t] = (unsigned char)c;
  }
};
__constant__ TriMap kTriMap;

// ---- the same solve for reduced systems of up to 159 unknowns (ten key frames of a visual-inertial window, 26 of a
// vision-only one), blocked by 16 on the FP64 matrix cores.  The lower triangle of [H b; b^T 1] lives in LDS as
// 16 x 16 blocks (row pitch 17): row n carries the right-hand side, so the forward substitution falls out of the
// factorisation.  Per block column k, two phases with a barrier each:
//   panel      one thread per row below: W = A_ik L_kk^-T (un-normalised, kept in sWp), L_ik = W D_k^-1 in place
//   trailing   A_ij -= W_ik L_jk^T, one v_mfma_f64_16x16x4_f64 chain of four per block, blocks dealt over the
//              four wavefronts; wavefront 0 takes A_(k+1)(k+1) first and factorises it in registers right after
//              (one lane per row, v_readlane for the column broadcasts), hidden behind the other wavefronts' blocks
// then L^T x = z from the bottom block up (wavefront 0 solves the 16 x 16 triangle, the others fold the block into
// the earlier entries of z).
// Measured (MI355X, 150 unknowns, s_memtime inside the kernel): 75 us per solve against 200 us for the column-panel
// kernel above -- load 8.5, per block column 1.9 (panel) + 3.2 .. 5.2 (diagonal factor 2.7 on the critical path),
// back substitution 7.8, pose update 4.3.  What did NOT help: DPP row_newbcast instead of v_readlane in the
// diagonal factor (5.8 us instead of 2.9), letting every wavefront walk the whole block list and skip the blocks
// of the others (the scalar loop of 16 wavefronts serialises on the CU's scalar unit: +3.5 us at k = 0).
static const int kLdP = 17, kLdBlk = 16 * kLdP, kLd16MaxBlocks = 10, kLd16Threads = 1024;

__device__ __forceinline__ int ld16_blk(int i, int j) { return (i * (i + 1) / 2 + j) * kLdBlk; }

static size_t ld16_lds_bytes(int nbm) { return ((size_t)(nbm * (nbm + 1) / 2 + nbm) * kLdBlk + 16 + nbm * 16) * 8; }

template <int NT>
__global__ void __launch_bounds__(NT)
k_lba_ldlt16(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, WinOut* __restrict__ out, int nb_max) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  __shared__ double s_red[4];
  __shared__ int s_bad;
  constexpr int NW = NT / 64;
  const int w = blockIdx.x;
  if (!(ctl[w].flags & LBA_TRIAL)) return;
  const LbaDev& D = devs[w];
  if (D.solver != 0) return;
  const int n = D.np, tid = threadIdx.x, lane = tid & 63;
  // wave-uniform for the compiler: the MFMA blocks of the other wavefronts must be branched over, not masked
  // (v_mfma ignores EXEC and would run its passes anyway)
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (n == 0) {
    if (tid == 0) out[w].ok = 1, out[w].scale_p = 0;
    return;
  }
  const double lambda = win_lambda(ctl[w], out[w]);
  const int nb = (n + 16) >> 4;  // blocks of the bordered matrix, n + 1 rows
  double* sA = s_dyn;
  double* sWp = sA + (size_t)(nb_max * (nb_max + 1) / 2) * kLdBlk;
  double* sD = sWp + (size_t)nb_max * kLdBlk;
  double* y = sD + 16;
  if (tid == 0) s_bad = 0;
  {  // H, coalesced: the blocks on and below the diagonal
    const float inv_n = 1.0f / (float)n;
#pragma unroll 8
    for (int i = tid; i < n * n; i += NT) {
      const int r = (int)(((float)i + 0.5f) * inv_n), c = i - r * n;  // exact: n * n < 2^15
      if ((c >> 4) <= (r >> 4)) sA[ld16_blk(r >> 4, c >> 4) + (r & 15) * kLdP + (c & 15)] = D.Hs[i];
    }
    // border: row n = b^T with pivot 1, identity padding below
    const int rows = nb * 16 - n;
    for (int i = tid; i < rows * nb * 16; i += NT) {
      const int r = n + i / (nb * 16), c = i % (nb * 16);
      if ((c >> 4) > (r >> 4)) continue;
      const double v = r == n ? (c < n ? D.bs[c] : (c == n ? 1.0 : 0.0)) : (r == c ? 1.0 : 0.0);
      sA[ld16_blk(r >> 4, c >> 4) + (r & 15) * kLdP + (c & 15)] = v;
    }
    // columns >= n of the rows above the border in the last block column
    for (int i = tid; i < 16 * 16; i += NT) {
      const int r = (n >> 4) * 16 + (i >> 4), c = (n >> 4) * 16 + (i & 15);
      if (r < n && c >= n) sA[ld16_blk(n >> 4, n >> 4) + (r & 15) * kLdP + (c & 15)] = 0.0;
    }
  }
  __syncthreads();
  typedef double double4_t __attribute__((ext_vector_type(4)));
  // A_kk = L D L^T in the registers of wavefront 0 (lanes 16..63 mirror lanes 0..15 so that the readlanes stay
  // wave-uniform); 1 / d_c (v_rcp_f64 + two Newton steps) goes to sD for the panel
  auto factor_diag = [&](int k) {
    double* Akk = sA + ld16_blk(k, k);
    const int r = lane & 15;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; c++) a[c] = Akk[r * kLdP + c];
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 16; c++) {
      const double d = big_readlane(a[c], c);
      if (!(d > 0) && k * 16 + c < n) bad = true;  // the right-hand-side row and the padding are not pivots
      double inv = __builtin_amdgcn_rcp(d);
      inv = __builtin_fma(__builtin_fma(-d, inv, 1.0), inv, inv);
      inv = __builtin_fma(__builtin_fma(-d, inv, 1.0), inv, inv);
      const double l = a[c] * inv;
#pragma unroll
      for (int q = c + 1; q < 16; q++) a[q] -= l * big_readlane(a[c], q);  // w_q = A[q][c], un-normalised
      if (r > c) a[c] = l;
      if (r == c) a[c] = inv;
    }
    if (lane < 16) {
#pragma unroll
      for (int c = 0; c < 16; c++) {
        if (c < r) Akk[r * kLdP + c] = a[c];
        if (c == r) sD[c] = a[c];
      }
      if (bad) s_bad = 1;
    }
  };
  if (wv == 0) factor_diag(0);
  __syncthreads();
  bool ok = true;
  for (int k = 0; k < nb; k++) {
    if (s_bad) {
      ok = false;
      break;
    }
    const double* Akk = sA + ld16_blk(k, k);
    if (tid < (nb - k - 1) * 16) {
      const int i = k + 1 + (tid >> 4), r = tid & 15;
      double* src = sA + ld16_blk(i, k) + r * kLdP;
      double a[16];
#pragma unroll
      for (int c = 0; c < 16; c++) a[c] = src[c];
      // W L_kk^T = A: W[c] = A[c] - sum_{m < c} W[m] L_kk[c][m]   (LDS reads are broadcasts)
#pragma unroll
      for (int c = 1; c < 16; c++) {
        double v = a[c];
#pragma unroll
        for (int m = 0; m < c; m++) v -= a[m] * Akk[c * kLdP + m];
        a[c] = v;
      }
      double* wp = sWp + i * kLdBlk + r * kLdP;
#pragma unroll
      for (int c = 0; c < 16; c++) {
        wp[c] = a[c];
        src[c] = a[c] * sD[c];
      }
    }
    __syncthreads();
    // trailing update; the wavefront that owns A_(k+1)(k+1) factorises it straight away, one step ahead
    const int m = nb - k - 1;
    for (int t = wv; t < m * (m + 1) / 2; t += NW) {
      const int i = k + 1 + kTriMap.i[t], j = k + 1 + kTriMap.j[t];
      double* C = sA + ld16_blk(i, j) + (lane >> 4) * kLdP + (lane & 15);
      double4_t acc;
#pragma unroll
      for (int q = 0; q < 4; q++) acc[q] = C[q * 4 * kLdP];
      const double* pa = sWp + i * kLdBlk + (lane & 15) * kLdP + (lane >> 4);
      const double* pb = sA + ld16_blk(j, k) + (lane & 15) * kLdP + (lane >> 4);
#pragma unroll
      for (int ks = 0; ks < 4; ks++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[ks * 4], pb[ks * 4], acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; q++) C[q * 4 * kLdP] = acc[q];
      if (t == 0) factor_diag(k + 1);  // wavefront 0; its own LDS traffic is in order
    }
    __syncthreads();
  }
  if (ok) {
    // z = D^-1 L^-1 b is row n of the factor; L^T x = z from the bottom block up.  Wavefront 0 folds x of block
    // kb + 1 into block kb and solves its triangle at once, the others fold it into the entries above.
    for (int i = tid; i < nb * 16; i += NT) y[i] = i < n ? sA[ld16_blk(n >> 4, i >> 4) + (n & 15) * kLdP + (i & 15)] : 0.0;
    __syncthreads();
    const int top = (n - 1) >> 4;
    for (int kb = top; kb >= 0; kb--) {
      if (wv == 0) {
        const int r = lane & 15, j = kb * 16 + r;
        double s = y[j];
        if (kb < top) {
          const double* blk = sA + ld16_blk(kb + 1, kb) + r;
#pragma unroll
          for (int q = 0; q < 16; q++) s -= blk[q * kLdP] * y[(kb + 1) * 16 + q];  // entries >= n are zero
        }
        const double* Akk = sA + ld16_blk(kb, kb);
        double Lc[16];
#pragma unroll
        for (int rr = 1; rr < 16; rr++) Lc[rr] = (r < rr && kb * 16 + rr < n) ? Akk[rr * kLdP + r] : 0.0;
#pragma unroll
        for (int rr = 15; rr >= 1; rr--) s -= Lc[rr] * big_readlane(s, rr);
        if (lane < 16 && j < n) y[j] = s;
      } else if (kb < top && tid - 64 < kb * 16) {
        const int j = tid - 64;
        const double* blk = sA + ld16_blk(kb + 1, j >> 4) + (j & 15);
        double v = y[j];
#pragma unroll
        for (int q = 0; q < 16; q++) v -= blk[q * kLdP] * y[(kb + 1) * 16 + q];
        y[j] = v;
      }
      __syncthreads();
    }
  } else {
    for (int i = tid; i < n; i += NT) y[i] = 0;
    __syncthreads();
  }
  lba_apply_step<NT>(D, y, lambda, ok, out[w], s_red, tid);
}

// ---- the same solve for reduced systems of 160 .. 639 unknowns (a bLarge window: 25 key frames x 15 = 375; a vision-only
// window of up to 106 key frames), one workgroup per window.  The lower triangle no longer fits LDS (567 KB at 375), so
// the factor lives in global memory -- L2-resident, 1.2 MB -- as 16 x 16 blocks stored COLUMN-major: lane l of a
// v_mfma_f64_16x16x4_f64 operand is element [l & 15][4 ks + (l >> 4)] of its block, i.e. double ks * 64 + l, so every
// operand load of a wavefront is one contiguous 512-byte read.  LEFT-looking by block column k:
//   gather     A_ik - sum_{j<k} W_ij L_kj^T for the row blocks i >= k: a chain of 4 k MFMAs per block with the
//              accumulator in registers (nothing is written back until the block is final; the right-looking form
//              would read and write every trailing block at every step).  What bounds it is the L2 round trip per
//              operand group, so a wavefront takes up to three row blocks at a time and shares the L_kj operands
//              between them; results go to LDS in row layout
//   diagonal   wavefront 0 gathers A_kk alone and factorises it in registers while the others still gather (row per
//              lane, v_readlane broadcasts)
//   panel      one thread per row below: W = A L_kk^-T (un-normalised) and L = W D_k^-1 -> global, column-major blocks
// Row n of the bordered matrix [H b; b^T 1] carries the right-hand side (forward substitution for free); L^T x = z from
// the bottom block up as in k_lba_ldlt16.  Same-workgroup visibility of the global writes: the wavefronts of a
// workgroup share the CU's vector L1 (write-through), so the barrier's workgroup-scope fence is enough.
// It replaces the multi-workgroup tiled solve for this size class (6 x k_big_panel + 5 x k_big_syrk + 6 x k_big_back_step
// per trial: 1.1 ms for the 51 bLarge windows of a bench step) and the column-panel kernel crawling in L2 (2.4 ms).
static const int kLdGMaxBlocks = 40, kLdGThreads = 512;
static size_t ldg_lds_bytes(int nbm) { return ((size_t)(nbm + 1) * kLdBlk + 16 + (size_t)nbm * 16) * 8; }
static size_t ldg_scratch_doubles(int nbm) { return (size_t)nbm * (nbm + 1) / 2 * 256 * 2; }

// element (row, col), col <= row's block, of the bordered matrix [H b; b^T 1] with identity padding
__device__ __forceinline__ double ldg_bordered(const LbaDev& D, int n, int row, int col) {
  if (row < n) return col < n ? D.Hs[(size_t)row * n + col] : 0.0;
  if (row == n) return col < n ? D.bs[col] : (col == n ? 1.0 : 0.0);
  return row == col ? 1.0 : 0.0;
}

// CNT row blocks i0, i0 + step, ... of block column k: A_ik - sum_{j<k} W_ij L_kj^T -> sP (row layout).  Four j per
// round trip to L2: the 16 operand loads of L_kj serve all CNT blocks, the CNT x 16 of W_ij go out with them, then
// 16 CNT MFMAs on 2 CNT independent accumulators.
template <int CNT>
__device__ __forceinline__ void ldg_gather(const LbaDev& D, int n, int k, int i0, int step, const double* Wg,
                                           const double* Lg, double* sP, int lane) {
  typedef double double4_t __attribute__((ext_vector_type(4)));
  double4_t acc[CNT][2];
  const double* pw[CNT];
#pragma unroll
  for (int c = 0; c < CNT; c++) {
    const int i = i0 + c * step;
#pragma unroll
    for (int q = 0; q < 4; q++) acc[c][0][q] = ldg_bordered(D, n, i * 16 + (lane >> 4) + 4 * q, k * 16 + (lane & 15));
    acc[c][1] = (double4_t){0, 0, 0, 0};
    pw[c] = Wg + (size_t)(i * (i + 1) / 2) * 256 + lane;
  }
  const double* pl = Lg + (size_t)(k * (k + 1) / 2) * 256 + lane;
  for (int j0 = 0; j0 < k; j0 += 4) {
    double b[16], a[CNT][16];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const bool in = j0 + u < k;
      const size_t o = (size_t)(in ? j0 + u : 0) * 256;
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        b[u * 4 + ks] = in ? pl[o + ks * 64] : 0.0;
#pragma unroll
        for (int c = 0; c < CNT; c++) a[c][u * 4 + ks] = in ? pw[c][o + ks * 64] : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int ks = 0; ks < 4; ks++)
#pragma unroll
        for (int c = 0; c < CNT; c++)
          acc[c][u & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[c][u * 4 + ks], b[u * 4 + ks], acc[c][u & 1], 0, 0, 0);
  }
#pragma unroll
  for (int c = 0; c < CNT; c++) {
    double* C = sP + (size_t)(i0 + c * step) * kLdBlk + (lane >> 4) * kLdP + (lane & 15);
#pragma unroll
    for (int q = 0; q < 4; q++) C[q * 4 * kLdP] = acc[c][0][q] + acc[c][1][q];
  }
}

template <int NT>
__global__ void __launch_bounds__(NT)
k_lba_ldltg(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, WinOut* __restrict__ out, int nb_max) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  __shared__ double s_red[4];
  __shared__ int s_bad;
  constexpr int NW = NT / 64;
  const int w = blockIdx.x;
  if (!(ctl[w].flags & LBA_TRIAL)) return;
  const LbaDev& D = devs[w];
  if (D.solver != 1) return;
  const int n = D.np, tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: MFMA blocks are branched over, not masked
  if (n == 0) {
    if (tid == 0) out[w].ok = 1, out[w].scale_p = 0;
    return;
  }
  const double lambda = win_lambda(ctl[w], out[w]);
  const int nb = (n + 16) >> 4;  // blocks of the bordered matrix, n + 1 rows
  double* sP = s_dyn;                         // [nb] the block column after the gather, row layout, pitch kLdP
  double* sKK = sP + (size_t)nb_max * kLdBlk;  // L_kk (unit lower)
  double* sD = sKK + kLdBlk;                   // 1 / d of the block's pivots
  double* y = sD + 16;
  // (plain generic pointers on purpose: with address_space(1) loads the kernel measured 523 instead of 405 us)
  double* Wg = D.Hb;                                           // un-normalised W = L D, blocks (i, j), j <= i
  double* Lg = Wg + (size_t)D.nb * (D.nb + 1) / 2 * 256;      // L (D.nb: block capacity of the window)
  if (tid == 0) s_bad = 0;
  __syncthreads();
  bool ok = true;
  for (int k = 0; k < nb; k++) {
    // ---- gather: wavefront 0 takes the diagonal block alone and factorises it at once; the others share the row
    // blocks below, up to three at a time with the L_kj operands loaded once for all of them
    if (wv == 0) {
      ldg_gather<1>(D, n, k, k, 0, Wg, Lg, sP, lane);
      // ---- diagonal block: A_kk = L D L^T in the registers of wavefront 0 (lanes 16..63 mirror lanes 0..15 so
      // that the readlanes stay wave-uniform); its own LDS traffic is in order
      const double* Akk = sP + (size_t)k * kLdBlk;
      const int r = lane & 15;
      double a[16];
#pragma unroll
      for (int c = 0; c < 16; c++) a[c] = Akk[r * kLdP + c];
      bool bad = false;
#pragma unroll
      for (int c = 0; c < 16; c++) {
        const double d = big_readlane(a[c], c);
        if (!(d > 0) && k * 16 + c < n) bad = true;  // the right-hand-side row and the padding are not pivots
        double inv = __builtin_amdgcn_rcp(d);
        inv = __builtin_fma(__builtin_fma(-d, inv, 1.0), inv, inv);
        inv = __builtin_fma(__builtin_fma(-d, inv, 1.0), inv, inv);
        const double l = a[c] * inv;
#pragma unroll
        for (int q = c + 1; q < 16; q++) a[q] -= l * big_readlane(a[c], q);  // w_q = A[q][c], un-normalised
        if (r > c) a[c] = l;
        if (r == c) a[c] = inv;
      }
      if (lane < 16) {
        double* Lkk = Lg + (size_t)(k * (k + 1) / 2 + k) * 256;
#pragma unroll
        for (int c = 0; c < 16; c++) {
          if (c < r) sKK[r * kLdP + c] = a[c], Lkk[c * 16 + r] = a[c];
          if (c == r) sD[c] = a[c];
        }
        if (bad) s_bad = 1;
      }
    } else {
      for (int i0 = k + wv; i0 < nb; i0 += 3 * (NW - 1)) {
        const int cnt = (nb - i0 + NW - 2) / (NW - 1);  // row blocks i0, i0 + NW - 1, ... below nb
        if (cnt >= 3)
          ldg_gather<3>(D, n, k, i0, NW - 1, Wg, Lg, sP, lane);
        else if (cnt == 2)
          ldg_gather<2>(D, n, k, i0, NW - 1, Wg, Lg, sP, lane);
        else
          ldg_gather<1>(D, n, k, i0, NW - 1, Wg, Lg, sP, lane);
      }
    }
    __syncthreads();
    if (s_bad) {
      ok = false;
      break;
    }
    // ---- panel: one thread per row below the diagonal block
    for (int t = tid; t < (nb - k - 1) * 16; t += NT) {
      const int i = k + 1 + (t >> 4), r = t & 15;
      const double* src = sP + (size_t)i * kLdBlk + r * kLdP;
      double a[16];
#pragma unroll
      for (int c = 0; c < 16; c++) a[c] = src[c];
      // W L_kk^T = A: W[c] = A[c] - sum_{m < c} W[m] L_kk[c][m]   (LDS reads are broadcasts)
#pragma unroll
      for (int c = 1; c < 16; c++) {
        double v = a[c];
#pragma unroll
        for (int m = 0; m < c; m++) v -= a[m] * sKK[c * kLdP + m];
        a[c] = v;
      }
      double* wg = Wg + (size_t)(i * (i + 1) / 2 + k) * 256 + r;
      double* lg = Lg + (size_t)(i * (i + 1) / 2 + k) * 256 + r;
#pragma unroll
      for (int c = 0; c < 16; c++) {
        wg[c * 16] = a[c];
        lg[c * 16] = a[c] * sD[c];
      }
    }
    __threadfence_block();
    __syncthreads();
  }
  if (ok) {
    // z = D^-1 L^-1 b is row n of the factor; L^T x = z from the bottom block up
    const int bn = n >> 4, rn = n & 15;
    for (int i = tid; i < nb * 16; i += NT)
      y[i] = i < n ? Lg[(size_t)(bn * (bn + 1) / 2 + (i >> 4)) * 256 + (i & 15) * 16 + rn] : 0.0;
    __syncthreads();
    const int top = (n - 1) >> 4;
    for (int kb = top; kb >= 0; kb--) {
      if (wv == 0) {
        const int r = lane & 15, j = kb * 16 + r;
        double s = y[j];
        if (kb < top) {
          const double* blk = Lg + (size_t)((kb + 1) * (kb + 2) / 2 + kb) * 256 + r * 16;
#pragma unroll
          for (int q = 0; q < 16; q++) s -= blk[q] * y[(kb + 1) * 16 + q];  // entries >= n are zero
        }
        const double* Lkk = Lg + (size_t)(kb * (kb + 1) / 2 + kb) * 256 + r * 16;
        double Lc[16];
#pragma unroll
        for (int rr = 1; rr < 16; rr++) Lc[rr] = (r < rr && kb * 16 + rr < n) ? Lkk[rr] : 0.0;
#pragma unroll
        for (int rr = 15; rr >= 1; rr--) s -= Lc[rr] * big_readlane(s, rr);
        if (lane < 16 && j < n) y[j] = s;
      } else if (kb < top) {
        for (int j = tid - 64; j < kb * 16; j += NT - 64) {
          const double* blk = Lg + (size_t)((kb + 1) * (kb + 2) / 2 + (j >> 4)) * 256 + (j & 15) * 16;
          double v = y[j];
#pragma unroll
          for (int q = 0; q < 16; q++) v -= blk[q] * y[(kb + 1) * 16 + q];
          y[j] = v;
        }
      }
      __syncthreads();
    }
  } else {
    for (int i = tid; i < n; i += NT) y[i] = 0;
    __syncthreads();
  }
  lba_apply_step<NT>(D, y, lambda, ok, out[w], s_red, tid);
}

// ---- tiled LDL^T for reduced systems that do not fit one workgroup (global BA: hundreds of key frames).
// Right-looking over 64-column panels of the padded lower triangle Hb [nb][nb]; row np carries the right-hand
// side, so the forward substitution falls out of the factorisation (its row of L is D^-1 L^-1 b).
//   k_big_init   Hb <- Hs, bs; identity on the padding
//   k_big_panel  step k: every workgroup factorises the 64 x 64 diagonal tile in LDS (redundantly: 87 kflop) and
//                solves its own 256 rows below it, one row per lane held in registers: W = A L_kk^-T -> Wp
//                (un-normalised), L = W D^-1 in place
//   k_big_syrk   A_ij -= W_ik L_jk^T for the tiles right of the panel: FP64 MFMA, 64 x 64 x 64 per workgroup
//   k_big_back_step  L^T x = z, one 64-row block per launch from the bottom; k_big_finish: lba_apply_step
static const int kNB = 64, kBigLd = kNB + 2;

__global__ void __launch_bounds__(256)
k_big_init(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl) {
  const int w = blockIdx.y;
  if (!(ctl[w].flags & LBA_TRIAL)) return;
  const LbaDev& D = devs[w];
  if (D.solver != 2) return;
  const int n = D.np, nb = D.nb;
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e == 0) *D.big_fail = 0;
  if (n == 0 || e >= (size_t)nb * nb) return;
  const int r = (int)(e / nb), c = (int)(e - (size_t)r * nb);
  if (c > r) return;
  double v = r == c ? (r == n ? 1e300 : 1.0) : 0.0;  // the right-hand-side row never limits a pivot
  if (r < n)
    v = D.Hs[(size_t)r * n + c];
  else if (r == n && c < n)
    v = D.bs[c];
  D.Hb[e] = v;
}

__global__ void __launch_bounds__(256)
k_big_panel(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, int k) {
  __shared__ double sA[kNB * kBigLd];  // the diagonal tile, then its unit-lower factor
  __shared__ double sD[kNB];
  const int w = blockIdx.y;
  if (!(ctl[w].flags & LBA_TRIAL)) return;
  const LbaDev& D = devs[w];
  if (D.solver != 2) return;
  const int n = D.np, nb = D.nb, tid = threadIdx.x;
  const int r0 = (k + 1) * kNB + blockIdx.x * 256 - 256;  // block 0: the diagonal tile itself, then 256 rows each
  if (n == 0 || k * kNB >= nb || (blockIdx.x > 0 && r0 >= nb)) return;
  double* Hb = D.Hb;
  const size_t ld = nb;
  {  // the 16 loads of a thread in flight, then the LDS stores
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int i = tid + 256 * u, r = i >> 6, c = i & 63;
      v[u] = c <= r ? Hb[(size_t)(k * kNB + r) * ld + k * kNB + c] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int i = tid + 256 * u;
      sA[(i >> 6) * kBigLd + (i & 63)] = v[u];
    }
  }
  __syncthreads();
  // LDL^T of the tile by ONE wavefront, register resident: lane r holds row r, column c is broadcast with
  // v_readlane (no LDS traffic, no barriers); lanes r <= c carry don't-care values in a[q], q > r.
  bool bad = false;
  if (tid < kNB) {
    double a[kNB];
#pragma unroll
    for (int c = 0; c < kNB; c++) a[c] = sA[tid * kBigLd + c];
#pragma unroll
    for (int c = 0; c < kNB; c++) {
      const double d = big_readlane(a[c], c);
      if (!(d > 0) && k * kNB + c < n) bad = true;  // the right-hand-side row and the padding are not pivots
      const double l = a[c] / d;
#pragma unroll
      for (int q = c + 1; q < kNB; q++) a[q] -= l * big_readlane(a[c], q);  // w_q = A[q][c], un-normalised
      if (tid > c) a[c] = l;
    }
#pragma unroll
    for (int c = 0; c < kNB; c++) {
      if (c < tid) sA[tid * kBigLd + c] = a[c];
      if (c == tid) sD[c] = a[c];
    }
    if (bad && blockIdx.x == 0) *D.big_fail = 1;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    for (int i = tid; i < kNB * kNB; i += 256) {
      const int r = i >> 6, c = i & 63;
      if (c < r) Hb[(size_t)(k * kNB + r) * ld + k * kNB + c] = sA[r * kBigLd + c];
      if (c == r) Hb[(size_t)(k * kNB + r) * ld + k * kNB + c] = sD[c];
    }
    return;
  }
  const int row = r0 + tid;
  if (row >= nb) return;
  double a[kNB];
  double* src = Hb + (size_t)row * ld + k * kNB;
#pragma unroll
  for (int c = 0; c < kNB; c++) a[c] = src[c];
  // W L_kk^T = A: W[c] = A[c] - sum_{m < c} W[m] L_kk[c][m]   (LDS reads are broadcasts)
#pragma unroll
  for (int c = 1; c < kNB; c++) {
    double v = a[c];
#pragma unroll
    for (int m = 0; m < c; m++) v -= a[m] * sA[c * kBigLd + m];
    a[c] = v;
  }
  double* wp = D.Wp + (size_t)row * kNB;
#pragma unroll
  for (int c = 0; c < kNB; c++) {
    wp[c] = a[c];
    src[c] = a[c] / sD[c];
  }
}

__global__ void __launch_bounds__(256)
k_big_syrk(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, int k) {
  __shared__ __attribute__((aligned(16))) double sW[kNB * kBigLd];
  __shared__ __attribute__((aligned(16))) double sL[kNB * kBigLd];
  const int w = blockIdx.y;
  if (!(ctl[w].flags & LBA_TRIAL)) return;
  const LbaDev& D = devs[w];
  if (D.solver != 2) return;
  const int nb = D.nb, nt = nb / kNB, m = nt - k - 1;
  if (D.np == 0 || m <= 0) return;
  int t = blockIdx.x, ti = 0;  // tile (k + 1 + ti, k + 1 + tj), tj <= ti, row-major over the lower triangle
  if (t >= m * (m + 1) / 2) return;
  while (t > ti) t -= ti + 1, ti++;
  const int bi = k + 1 + ti, bj = k + 1 + t;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const size_t ld = nb;
  {  // both tiles in flight (32 loads per thread), then the LDS stores
    double vw[16], vl[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int i = tid + 256 * u, r = i >> 6, c = i & 63;
      vw[u] = D.Wp[(size_t)(bi * kNB + r) * kNB + c];
      vl[u] = D.Hb[(size_t)(bj * kNB + r) * ld + k * kNB + c];
    }
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int i = tid + 256 * u, r = i >> 6, c = i & 63;
      sW[r * kBigLd + c] = vw[u];
      sL[r * kBigLd + c] = vl[u];
    }
  }
  __syncthreads();
  typedef double double4_t __attribute__((ext_vector_type(4)));
  double4_t acc[4];
#pragma unroll
  for (int q = 0; q < 4; q++) acc[q] = (double4_t){0, 0, 0, 0};
  const double* pa = sW + (wv * 16 + (lane & 15)) * kBigLd + (lane >> 4);
  const double* pb = sL + (lane & 15) * kBigLd + (lane >> 4);
#pragma unroll
  for (int ks = 0; ks < kNB / 4; ks++) {
    const double av = pa[ks * 4];
#pragma unroll
    for (int q = 0; q < 4; q++)
      acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, pb[q * 16 * kBigLd + ks * 4], acc[q], 0, 0, 0);
  }
  // f64 C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int gr = bi * kNB + wv * 16 + (lane >> 4) + 4 * r, gc = bj * kNB + q * 16 + (lane & 15);
      if (gc <= gr) D.Hb[(size_t)gr * ld + gc] -= acc[q][r];
    }
}

// step s of L^T x = z from the bottom: block kb = last - s.  Every workgroup solves the 64 x 64 block in LDS
// (one wavefront, redundantly); workgroup 0 publishes x of the block, the others fold it into their 256 columns
// of z (kept in Wp, free after the factorisation): z[c] -= sum_r L[c0 + r][c] x[r].
__global__ void __launch_bounds__(256)
k_big_back_step(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, int s) {
  __shared__ double sA[kNB * kBigLd];
  __shared__ double sx[kNB];
  const int w = blockIdx.y;
  if (!(ctl[w].flags & LBA_TRIAL)) return;
  const LbaDev& D = devs[w];
  if (D.solver != 2) return;
  const int n = D.np, nb = D.nb, tid = threadIdx.x;
  if (n == 0) return;
  const int kb = (n - 1) / kNB - s;
  if (kb < 0) return;
  const int c0 = kb * kNB, cn = min(kNB, n - c0);
  if (blockIdx.x > 0 && (int)(blockIdx.x - 1) * 256 >= c0) return;
  const size_t ld = nb;
  const double* z = s == 0 ? D.Hb + (size_t)n * ld : D.Wp;  // z = D^-1 L^-1 b: the right-hand-side row of L
  {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int i = tid + 256 * u, r = i >> 6, c = i & 63;
      v[u] = (c < r && r < cn) ? D.Hb[(size_t)(c0 + r) * ld + c0 + c] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int i = tid + 256 * u;
      sA[(i >> 6) * kBigLd + (i & 63)] = v[u];
    }
  }
  if (tid < kNB) sx[tid] = tid < cn ? z[c0 + tid] : 0.0;
  __syncthreads();
  if (tid < kNB)  // x[r] is final once the rows below it have been applied
    for (int r = cn - 1; r > 0; r--) {
      const double xr = sx[r];
      if (tid < r) sx[tid] -= sA[r * kBigLd + tid] * xr;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  __syncthreads();
  if (blockIdx.x == 0) {
    if (tid < cn) D.xp[c0 + tid] = sx[tid];
    return;
  }
  const int c = (blockIdx.x - 1) * 256 + tid;
  if (c >= c0) return;
  double v = z[c];
  for (int r0 = 0; r0 < cn; r0 += 16) {  // 16 rows in flight; the subtraction order stays r ascending
    double l[16];
#pragma unroll
    for (int u = 0; u < 16; u++) l[u] = r0 + u < cn ? D.Hb[(size_t)(c0 + r0 + u) * ld + c] : 0.0;
#pragma unroll
    for (int u = 0; u < 16; u++)
      if (r0 + u < cn) v -= l[u] * sx[r0 + u];
  }
  D.Wp[c] = v;
}

__global__ void __launch_bounds__(256)
k_big_finish(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, WinOut* __restrict__ out) {
  __shared__ double s_red[4];
  const int w = blockIdx.x;
  if (!(ctl[w].flags & LBA_TRIAL)) return;
  const LbaDev& D = devs[w];
  if (D.solver != 2) return;
  const int n = D.np, tid = threadIdx.x;
  if (n == 0) {
    if (tid == 0) out[w].ok = 1, out[w].scale_p = 0;
    return;
  }
  const bool ok = *D.big_fail == 0;
  if (!ok) {
    for (int i = tid; i < n; i += 256) D.xp[i] = 0;
    __syncthreads();
  }
  lba_apply_step(D, D.xp, win_lambda(ctl[w], out[w]), ok, out[w], s_red, tid);
}

// ---- back-substitution + update of the points, landmark part of the LM gain-ratio scale
// (the body: k_lba_update_points, and k_lba_tail which goes on with the points' edges)
// Xn (optional): the updated point on the quad's lane 0, zeros on the other lanes and for an inactive point
__device__ __forceinline__ void lba_update_points_dev(const LbaDev& D, double lambda, int blk, double* s_red, double* Xn = nullptr) {
  // FOUR lanes per point, each over every fourth free key frame: a point's chain was one dependent `tab` -> block round
  // trip per free key frame (10 .. 25 of them, 73 us per launch whatever the batch); the quad's partial sums are added
  // in a fixed order
  const int m = blk * 64 + (threadIdx.x >> 2), sub = threadIdx.x & 3;
  const bool act = m < D.n_mp && D.mp_act[m];
  double sc[1] = {0};
  double cl[3] = {0, 0, 0};
  if (act) {
    if (sub == 0) cl[0] = D.bl[3 * (size_t)m], cl[1] = D.bl[3 * (size_t)m + 1], cl[2] = D.bl[3 * (size_t)m + 2];
    for (int a = sub; a < D.n_free; a += 4) {
      const int e = D.tab[(size_t)a * D.n_mp + m];
      if (e < 0) continue;
      const double* B = D.CB + 18 * (size_t)e;
      for (int r = 0; r < 6; r++, B += 3) {
        const double xa = D.xp[D.pd * a + r];
        cl[0] -= B[0] * xa, cl[1] -= B[1] * xa, cl[2] -= B[2] * xa;
      }
    }
    if (D.scale_opt && sub == 0) {  // the (scale, point) block
      const double* B = D.Bs + 3 * (size_t)m;
      const double xa = D.xp[D.np - 1];
      cl[0] -= B[0] * xa, cl[1] -= B[1] * xa, cl[2] -= B[2] * xa;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) cl[k] = quad_sum_f64(cl[k]);
  if (act && sub == 0) {
    double Di[9];
    landmark_dinv(D.Hll + 9 * (size_t)m, lambda, Di);
    for (int a = 0; a < 3; a++) {
      const double x = Di[a * 3] * cl[0] + Di[a * 3 + 1] * cl[1] + Di[a * 3 + 2] * cl[2];
      const double old = D.X[3 * (size_t)m + a];
      D.X_bak[3 * (size_t)m + a] = old;
      D.X[3 * (size_t)m + a] = old + x;
      if (Xn) Xn[a] = old + x;
      sc[0] += x * (lambda * x + D.bl[3 * (size_t)m + a]);
    }
  }
  block_sum<1>(sc, s_red, threadIdx.x);
  if (threadIdx.x == 0) D.part_m[blk] = sc[0];  // one partial per 64 points (k_lba_reduce)
}
__global__ void __launch_bounds__(256)
k_lba_update_points(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl,
                    const WinOut* __restrict__ out) {
  __shared__ double s_red[4];
  const int w = blockIdx.y;
  if (!(ctl[w].flags & LBA_TRIAL)) return;
  const LbaDev& D = devs[w];
  if (blockIdx.x * 64 >= D.n_mp || D.np == 0) return;
  lba_update_points_dev(D, win_lambda(ctl[w], out[w]), blockIdx.x, s_red);
}

// ================================================================== g2o's LM policy, per window, on the device
// What the host loop below does between two rounds (optimization_algorithm_levenberg.cpp:61-164 + the two-stage
// optimize() sequence of the local BAs), as a kernel: the round trip trial -> copy back -> host decision -> copy up ->
// next launch (40 us of every ~200 us round of a single window) disappears, the host queues several rounds blind and
// looks at the windows' state every few rounds.  The arithmetic is the host's line for line; pow() is the device
// library's (the damping factor may differ from the host's in its last bit: far below the solve's own rounding).
struct WinPol {
  int stage, phase, it, iters, its1, nBad, qmax;
  int need_build, need_restore, prelevel_pending, vio, robust0;
  int lm_iterations, lm_trials, aborted, pad;
  double lambda, ni, currentChi, iniChi, lastTrialChi, lambda_init, chi2_initial, chi2_final;
};

// first: only the next round's flags (the first round).  allow_begin: the round being prepared contains the kernels of a
// stage start (classification, active sets, initial chi2, lambda); a window that needs them in a round without waits.
// restore_here: the caller rolls a rejected trial back itself, at once (k_lba_tail's last workgroup) -- returns 1 then --
// instead of a LBA_RESTORE flag for the next round's k_lba_restore.
// ctl: the flags of the round that has just run; ctl_next: where the next round's go (the same array for k_lba_policy, which
// is a launch of its own; the other of two for k_lba_tail, whose workgroups read the current flags while one of them decides).
__device__ __forceinline__ int lba_policy_dev(WinPol* __restrict__ pol, const WinCtl* ctl, WinCtl* ctl_next, const WinOut* __restrict__ out, int w,
                                              int first, int allow_begin, int stop_now, bool restore_here) {
  WinPol H = pol[w];
  int restore_now = 0;
  const int fl = first ? 0 : ctl[w].flags;
  if (fl & LBA_TRIAL) {
    bool run = true;
    if (fl & LBA_BEGIN) {
      if (out[w].np == 0) {  // no active free vertex: optimize() returns at once
        H.phase = 2;
        run = false;
      } else {
        H.lm_iterations++;
        H.currentChi = out[w].chi0;
        if (H.stage == 0) H.chi2_initial = H.currentChi;
        H.iniChi = H.currentChi;
        H.lambda = H.vio ? H.lambda_init : out[w].lambda;
        H.ni = 2, H.nBad = 0, H.qmax = 0, H.it = 0;
        H.phase = 1;
      }
    }
    if (run) {
      H.lm_trials++;
      H.need_build = 0;
      const bool ok2 = out[w].ok != 0;
      H.lastTrialChi = out[w].chi2;
      const double tempChi = ok2 ? out[w].chi2 : DBL_MAX;
      double rho = H.currentChi - tempChi;
      const double scale = (ok2 ? out[w].scale_l + out[w].scale_p : 0.0) + 1e-3;
      rho /= scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow(2 * rho - 1, 3.0);
        alpha = fmin(alpha, 2. / 3.);
        H.lambda *= fmax(1. / 3., alpha);
        H.ni = 2;
        H.currentChi = tempChi;
      } else {
        H.lambda *= H.ni;
        H.ni *= 2;
        if (restore_here)
          restore_now = 1;
        else
          H.need_restore = 1;
      }
      H.qmax++;
      H.chi2_final = H.currentChi;
      if (!(rho < 0 && H.qmax < 10 && !stop_now)) {  // (else: the next lambda trial of the same iteration)
        bool terminate = H.qmax == 10 || rho == 0;
        if (!terminate) {
          if ((H.iniChi - H.currentChi) * 1e3 < H.iniChi)
            H.nBad++;
          else
            H.nBad = 0;
          terminate = H.nBad >= 3;
        }
        H.it++;
        if (terminate || H.it >= H.iters || stop_now)
          H.phase = 2;
        else {
          H.lm_iterations++;
          H.iniChi = H.currentChi;
          H.qmax = 0;
          H.need_build = 1;  // buildSystem at the accepted state
        }
      }
    }
  }
  // ---- the next round
  int f = 0;
  double lam = H.lambda;
  const bool starts = H.stage < 2 && (H.phase == 0 || (H.phase == 2 && H.stage == 0 && !stop_now));
  if (starts && !allow_begin && !first) {
    // (an optimize() would start in the round being prepared and that round has no stage-start kernels: wait a round)
  } else {
    if (H.stage < 2 && H.phase == 2) {  // an optimize() is over
      if (H.need_restore) f |= LBA_RESTORE, H.need_restore = 0;
      if (H.stage == 0 && !stop_now) {
        f |= LBA_CLASS0;
        H.stage = 1, H.iters = H.its1, H.phase = H.iters > 0 ? 0 : 2;
      } else {
        if (H.stage == 0) H.aborted = 1;  // stop flag between the two stages
        f |= LBA_CLASS1;
        H.stage = 2;
      }
    }
    if (H.stage < 2 && H.phase == 0) {
      f |= LBA_BEGIN | LBA_BUILD | LBA_TRIAL | ((H.stage == 0 && H.robust0) ? LBA_ROBUST : 0);
      lam = H.vio ? H.lambda_init : -1;
      if (H.prelevel_pending) f |= LBA_PRELEVEL, H.prelevel_pending = 0;
    } else if (H.stage < 2 && H.phase == 1) {
      f |= LBA_TRIAL | ((H.stage == 0 && H.robust0) ? LBA_ROBUST : 0);
      if (H.need_build) f |= LBA_BUILD;
      if (H.need_restore) f |= LBA_RESTORE, H.need_restore = 0;
    }
  }
  ctl_next[w].flags = f, ctl_next[w].pad = 0, ctl_next[w].lambda = lam;
  pol[w] = H;
  return restore_now;
}
__global__ void __launch_bounds__(64)
k_lba_policy(WinPol* __restrict__ pol, WinCtl* __restrict__ ctl, const WinOut* __restrict__ out, int W, int first,
             int allow_begin, int stop_now) {
  const int w = blockIdx.x * 64 + threadIdx.x;
  if (w >= W) return;
  (void)lba_policy_dev(pol, ctl, ctl, out, w, first, allow_begin, stop_now, false);
}

// ---- the tail of a trial as ONE launch (round 6; were k_lba_update_points, k_lba_error(1), k_lba_generic(1),
// k_lba_reduce: four dependent launches of 5-6 us each that are mostly launch ramp).
//   workgroups [0, gq): 64 points each -- back-substitution and update of the points as above, then the residuals and the
//     robust chi2 of THOSE points' edges (a point's observations are contiguous; the quad of a point takes every fourth;
//     the key-frame poses were updated by the solve kernel before this launch): no workgroup waits for another;
//   workgroups [gq, gq + pairs): the inertial / encoder pair edges' chi2 after the trial (one wavefront each);
//   the LAST workgroup of a window to arrive (a counter in global memory behind a device-scope fence) folds the
//     partials into the window's output record in k_lba_reduce's fixed order -- and, when the LM policy runs on the
//     device (pol != null), takes the window's decision right there (lba_policy_dev: the next round's flags and
//     damping) and rolls a rejected trial back at once: no policy launch, no restore launch, no host round trip.
// The trial's visual chi2 is summed per block of 64 points instead of per block of 256 edges: another (fixed)
// association order than the four-launch form, same terms.
__global__ void __launch_bounds__(256)
k_lba_tail(const LbaDev* __restrict__ devs, const WinCtl* __restrict__ ctl, WinOut* __restrict__ out, int gq, WinPol* __restrict__ pol,
           WinCtl* __restrict__ ctl_next, int allow_begin, int stop_now) {
  __shared__ double s_red[4 * 3];
  __shared__ int s_last;
  const int w = blockIdx.y, fl = ctl[w].flags;
  if (!(fl & LBA_TRIAL)) {  // (a window between two stages, waiting for a round with the stage-start kernels, or done)
    if (pol && blockIdx.x == 0 && threadIdx.x == 0) (void)lba_policy_dev(pol, ctl, ctl_next, out, w, 0, allow_begin, stop_now, true);
    return;
  }
  const LbaDev& D = devs[w];
  const int nblk = max((D.n_mp + 63) / 64, 1);  // (block 0 of an empty landmark shard still arrives)
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < gq) {
    if ((int)blockIdx.x >= nblk) return;
    if (D.np > 0 && (int)blockIdx.x * 64 < D.n_mp) {
      double Xn[3] = {0, 0, 0};
      lba_update_points_dev(D, win_lambda(ctl[w], out[w]), blockIdx.x, s_red, Xn);
      // the edges of the block's points at the trial state (the new point travels from the quad's lane 0 in registers:
      // x + 0 + 0 + 0 is exact)
      const int m = blockIdx.x * 64 + (tid >> 2), sub = tid & 3;
#pragma unroll
      for (int a = 0; a < 3; a++) Xn[a] = quad_sum_f64(Xn[a]);
      double v[1] = {0};
      if (m < D.n_mp && D.mp_act[m]) {  // (a point without an active edge has nothing to evaluate)
        const int first = D.mp_first[m], cnt = D.mp_count[m];
        const double scl = win_scale(D);
        const double Xs[3] = {Xn[0] * scl, Xn[1] * scl, Xn[2] * scl};
        for (int k = sub; k < cnt; k += 4) {
          const int i = first + k;
          if (D.level[i] != 0) continue;
          const vieo_lba_obs o = D.obs[i];
          PoseXf X;
          const CamD& C = obs_cam(D, i);
          kf_xf(C, D.kf[o.kf], X);
          double err[3], Pc[3];
          const double chi2 = lba_edge_error(C, X, o, Xs, err, Pc);
          D.err[3 * (size_t)i] = err[0], D.err[3 * (size_t)i + 1] = err[1], D.err[3 * (size_t)i + 2] = err[2];
          double r0 = chi2, r1;
          if (fl & LBA_ROBUST) {
            const double dl = o.ur >= 0 ? D.dStereo : D.dMono;
            huber(chi2, dl, dl * dl, &r0, &r1);
          }
          v[0] += r0;
        }
      }
      v[0] = quad_sum_f64(v[0]);
      if (sub != 0) v[0] = 0;
      block_sum<1>(v, s_red, tid);
      if (tid == 0) D.part_t[blockIdx.x] = v[0];
    }
  } else {
    const int e = blockIdx.x - gq;
    if (e >= D.n_imu) return;
    if (tid < 64) lba_generic_dev(D, e, tid, 1);
  }
  if (D.fold_kernel) return;  // (batches: k_lba_reduce(tail) folds the partials -- no device-scope fence per workgroup)
  // arrival: this workgroup's results are visible device-wide before its count is
  __threadfence();
  __syncthreads();
  if (tid == 0) {
    const int total = nblk + D.n_imu;
    const int old = atomicAdd(D.tail_cnt, 1);
    s_last = old == total - 1;
    if (s_last) *D.tail_cnt = 0;  // (for the next launch: nobody else touches it any more)
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  lba_reduce_dev(D, ctl, out, w, fl, true, s_red);
  if (!pol) return;
  // (every workgroup of the window has read its flags -- they all arrived -- so the control word can change now)
  if (tid == 0) s_last = lba_policy_dev(pol, ctl, ctl_next, out, w, 0, allow_begin, stop_now, true);
  __syncthreads();
  if (!s_last) return;
  for (int i = tid; i < D.n_mp; i += 256)  // k_lba_restore
    if (D.mp_act[i])
      for (int a = 0; a < 3; a++) D.X[3 * (size_t)i + a] = D.X_bak[3 * (size_t)i + a];
  for (int i = tid; i < D.n_kf; i += 256)
    if (D.kf[i].col >= 0) D.kf[i] = D.kf_bak[i];
  if (tid == 0 && D.scale_opt) D.scl[0] = D.scl[1];
}

// ================================================================== host-side lock-step LM driver
static thread_local hipStream_t g_lba_stream = nullptr;
static thread_local int g_lba_stream_dev = -1;  // the device the stream was created on
static thread_local int g_lba_priority = -2;    // vieo_lba_set_stream_priority of this host thread (-2: VIEO_LBA_PRIORITY / -1)
static thread_local DevBuf g_arena, g_small;
static thread_local PinnedBuf g_stage, g_small_h;

struct WinHost {  // per-window LM state machine, exactly g2o's (optimization_algorithm_levenberg.cpp)
  const vieo_lba_params* P;
  const vieo_lba_vio_params* VP = nullptr;  // visual-inertial window (a18)
  const vieo_lba_enc* ENC = nullptr;        // encoder edges of a vision-only window (a17)
  int n_kf, n_mp, n_obs, n_imu = 0;
  double lastTrialChi = 0;  // activeRobustChi2 of the errors left in the edges (err_end)
  bool prelevel_pending = false;
  int stage = 0;  // 0: optimize(its0), 1: optimize(its1), 2: finished
  int phase = 0;  // 0: the next round starts an optimize(), 1: in trials, 2: optimize() is over
  int it = 0, iters = 0;
  double lambda = -1, ni = 2, currentChi = 0, iniChi = 0;
  int nBad = 0, qmax = 0;
  bool need_build = false, need_restore = false, skip = false;
  vieo_lba_result* R;
  size_t o_kf, o_X, o_erase, o_scl;  // offsets of the results in the staging buffer
};

// 9x9 inverse by Gauss-Jordan with partial pivoting (GetProcessedInfoijPRV: mSigmaijPRV.inverse())
static bool inverse9(const double* A, double* Ainv) {
  double M[9][18];
  for (int i = 0; i < 9; i++)
    for (int j = 0; j < 9; j++) M[i][j] = A[i * 9 + j], M[i][9 + j] = (i == j);
  for (int c = 0; c < 9; c++) {
    int piv = c;
    for (int r = c + 1; r < 9; r++)
      if (std::fabs(M[r][c]) > std::fabs(M[piv][c])) piv = r;
    if (M[piv][c] == 0) return false;
    if (piv != c)
      for (int j = 0; j < 18; j++) std::swap(M[c][j], M[piv][j]);
    const double d = M[c][c];
    for (int j = 0; j < 18; j++) M[c][j] /= d;
    for (int r = 0; r < 9; r++)
      if (r != c) {
        const double f = M[r][c];
        if (f != 0)
          for (int j = 0; j < 18; j++) M[r][j] -= f * M[c][j];
      }
  }
  for (int i = 0; i < 9; i++)
    for (int j = 0; j < 9; j++) Ainv[i * 9 + j] = M[i][9 + j];
  return true;
}

// the same elimination for an n x n block (n <= 9): the encoder covariance
static bool inverse_n(const double* A, double* Ainv, int n) {
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

struct LbaShard {  // landmark-sharded run: every rank holds all key frames and its share of the points
  vieo_allreduce_sum_f64_fn fn;
  void* ctx;
  double* d_buf;
  size_t cap;
};

// one exchange of a sharded run: the caller's callback (complete on return; the stream is drained first) or, with
// fn == nullptr, ncclAllReduce on the stream itself (ctx = the communicator of vieo_rccl_comm_create)
static int shard_exchange(const LbaShard* sh, double* d_buf, size_t n, hipStream_t st) {
  if (sh->fn) {
    VIEO_HIP_CHECK(hipStreamSynchronize(st));
    if (sh->fn(sh->ctx, d_buf, n) != 0) {
      set_error("sharded local BA: the all-reduce callback failed");
      return VIEO_E_INVALID;
    }
    return VIEO_OK;
  }
  return rccl_allreduce_sum_f64(sh->ctx, d_buf, n, st);
}

// packed reduced visual system of a window with nf free key frames (k_lba_pack); sc: with the scale vertex's row,
// H_ps, H_ss, b_s
// All ranks of a sharded run agree on go / no-go: the sum of the ranks' failure flags through the run's own exchange.
// Every rank must call it the same number of times.  *sum > 0: some rank failed.
// stop / *stop_any: the ranks' stop requests travel in the same number (4096 per request: exact in a double for any
// realistic job size), so that pbStopFlag raised on one rank aborts the call on all of them together.
static int shard_agree(const LbaShard* sh, bool ok, double* sum, bool stop = false, bool* stop_any = nullptr) {
  const double flag = (ok ? 0.0 : 1.0) + (stop ? 4096.0 : 0.0);
  *sum = 1.0;
  VIEO_HIP_CHECK(hipMemcpy(sh->d_buf, &flag, 8, hipMemcpyHostToDevice));
  const int xrc = shard_exchange(sh, sh->d_buf, 1, nullptr);
  if (xrc != VIEO_OK) return xrc;
  VIEO_HIP_CHECK(hipStreamSynchronize(nullptr));
  double total = 0;
  VIEO_HIP_CHECK(hipMemcpy(&total, sh->d_buf, 8, hipMemcpyDeviceToHost));
  const double stops = std::floor(total / 4096.0);
  if (stop_any) *stop_any = stops > 0;
  *sum = total - 4096.0 * stops;
  return VIEO_OK;
}

// A rank that leaves lba_run between the first agreement and the rounds (a staging error only it ran into: allocation
// failure, a singular covariance) still takes part in the second agreement -- with a failure flag -- so that the
// other ranks return too instead of waiting for it in the first all-reduce of the rounds.
struct ShardStagingGuard {
  const LbaShard* sh;
  bool done = false;
  ~ShardStagingGuard() {
    if (sh && !done) {
      double sum;
      (void)shard_agree(sh, false, &sum);
    }
  }
};

static size_t shard_sys_doubles(int nf, int sc = 0) {
  const size_t nv = (size_t)6 * nf + (sc ? 1 : 0);
  return nv * (nv + 1) + 36 * (size_t)nf + 6 * (size_t)nf + (sc ? 6 * (size_t)nf + 2 : 0);
}

// Both local BAs: vparams == nullptr -> Optimizer::LocalBundleAdjustment (params), otherwise
// LocalBundleAdjustmentNavStatePRV (vparams, h_close, h_imu, n_imu).  sh != nullptr: this process is one
// rank of a landmark-sharded run.

// reduced systems beyond one workgroup's LDL^T take the tiled solve (VIEO_LBA_BIG_SOLVE=1 forces it: tests)
static const int kSmallSolveMax = 16 * kLdGMaxBlocks - 1, kBigSolveMax = 16320;
// ---- optional kernel-class timing (vieo_lba_enable_timing): HIP events around every launch on the BA stream, folded
// into process-wide totals after each round's synchronisation.  bench.py reads them for the roofline of the whole path.
enum LbaKClass { KC_BUILD, KC_GENERIC, KC_SCHUR, KC_ASSEMBLE, KC_LDLT, KC_UPDATE, KC_ERROR, KC_BEGIN, KC_OTHER, KC_N };
static const char* const kLbaKClassName[KC_N] = {"lba.build", "lba.generic", "lba.schur", "lba.assemble", "lba.ldlt",
                                                 "lba.update_points", "lba.error", "lba.begin", "lba.other"};
static std::atomic<int> g_lba_ktiming{0};
static std::mutex g_lba_kt_mutex;
static double g_lba_kt_ms[KC_N];
static long long g_lba_kt_launches[KC_N];
static double g_lba_kt_schur_flops;  // dense FLOPs of the timed k_lba_schur launches

struct LbaKTimer {
  bool on = false, count_only = false;  // count_only (mode 2): launches per class without events
  hipStream_t st = nullptr;
  std::vector<hipEvent_t> pool;
  std::vector<int> cls;  // class of pair i (events 2i, 2i + 1)
  void begin(hipStream_t s) {
    const int mode = g_lba_ktiming.load();
    on = mode != 0, count_only = mode == 2;
    st = s;
    cls.clear();
  }
  template <class F>
  void launch(int c, F&& f) {
    if (!on) {
      f();
      return;
    }
    if (count_only) {
      f();
      cls.push_back(c);
      return;
    }
    const size_t i = cls.size();
    while (pool.size() < 2 * (i + 1)) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) {
        on = false;
        f();
        return;
      }
      pool.push_back(e);
    }
    (void)hipEventRecord(pool[2 * i], st);
    f();
    (void)hipEventRecord(pool[2 * i + 1], st);
    cls.push_back(c);
  }
  void fold(double schur_flops_per_launch) {  // after the stream was synchronised
    if (!on || cls.empty()) return;
    std::lock_guard<std::mutex> g(g_lba_kt_mutex);
    bool schur_seen = false;  // a round's Schur complement is two launches (diagonal / off-diagonal tiles): FLOPs once
    for (size_t i = 0; i < cls.size(); i++) {
      float ms = 0;
      if (!count_only && hipEventElapsedTime(&ms, pool[2 * i], pool[2 * i + 1]) != hipSuccess) continue;
      g_lba_kt_ms[cls[i]] += ms, g_lba_kt_launches[cls[i]]++;
      if (cls[i] == KC_SCHUR && !schur_seen) g_lba_kt_schur_flops += schur_flops_per_launch, schur_seen = true;
    }
    cls.clear();
  }
};

static bool ldlt16_disabled() {  // VIEO_LBA_LDLT16=0: the column-panel kernel instead (A/B runs)
  static const int off = [] {
    const char* e = getenv("VIEO_LBA_LDLT16");
    return e && atoi(e) == 0;
  }();
  return off;
}

static bool big_solve(int n) {
  static const int forced = [] {
    const char* e = getenv("VIEO_LBA_BIG_SOLVE");
    return e ? atoi(e) : 0;
  }();
  return forced > 0 || n > kSmallSolveMax;
}

// The solve kernel of a window with n unknowns.  Windows of different classes share a lock-step batch: every class's
// kernel is launched (when one of its windows takes a trial this round) and skips the others' windows (a bLarge
// window of 25 key frames next to ordinary ones of 10 is the usual LocalMapping mix).
//   0  k_lba_ldlt16   n <= 159: blocked on the matrix cores, whole triangle in LDS
//   1  k_lba_ldltg    n <= 639: one workgroup, left-looking, the factor in L2 as column-major 16 x 16 blocks
//   2  k_big_*        the tiled LDL^T over many workgroups (full BA: hundreds of key frames)
static int solver_class(int n) {
  if (big_solve(n)) return 2;
  if (((n + 16) >> 4) <= kLd16MaxBlocks && !ldlt16_disabled()) return 0;
  return 1;
}

// Optimizer::BundleAdjustment / GlobalBundleAdjustmentNavStatePRV (Optimizer.cc:1353-1609, 771-1345) on the same
// engine: ONE optimize(iterations), Huber kernels on every edge iff `robust`, no chi2 classification, no
// Chi2LargeSetLevel, g2o's own initial lambda, no divergence guard.
struct GbaMode {
  int iterations, robust;
  int scale_opt = 0;          // bScaleOpt (System::FinalGBA): VertexScale + EdgeReprojectPRS[Stereo]
  double* scale_out = nullptr;  // the recovered scale (1 when the call returns early)
};

// The argument checks of lba_run without side effects.  A landmark-sharded run calls it first and lets all ranks agree
// on the outcome with one all-reduce: a rank that returned early on its own while the others were already waiting in
// the collective would hang them.  `sharded`: a rank may own no point (or no observation) of a window.
static bool lba_args_ok(bool sharded, bool vio, bool gba, int W, const vieo_lba_params* const* params,
                        const vieo_lba_vio_params* const* vparams, const vieo_lba_keyframe* const* h_kfs, const int* n_kf,
                        const float* const* h_points, const uint8_t* const* h_close, const int* n_mp,
                        const vieo_lba_obs* const* h_obs, const int* n_obs, const vieo_lba_imu_edge* const* h_imu,
                        const int* n_imu, vieo_navstate* const* h_navs_out, float* const* h_points_out,
                        uint8_t* const* h_erase, const vieo_lba_result* h_results, int pd, int sco) {
  if (W <= 0 || (!vio && !params) || !h_kfs || !n_kf || !h_points || !n_mp || !h_obs || !n_obs || !h_navs_out ||
      !h_points_out || !h_erase || !h_results || (vio && (!h_close || !h_imu || !n_imu)))
    return false;
  for (int w = 0; w < W; w++) {
    const vieo_lba_params* P = vio ? (vparams[w] ? &vparams[w]->base : nullptr) : params[w];
    if (!P || !h_kfs[w] || n_kf[w] <= 0 || n_mp[w] < 0 || n_obs[w] < 0 || !h_navs_out[w]) return false;
    if (!sharded && (n_mp[w] == 0 || n_obs[w] == 0)) return false;
    if (n_mp[w] > 0 && (!h_points[w] || !h_points_out[w])) return false;
    if (n_obs[w] > 0 && (!h_obs[w] || !h_erase[w])) return false;
    if (vio && ((!h_close[w] && !gba && n_mp[w] > 0) || n_imu[w] < 0 || (n_imu[w] > 0 && !h_imu[w]))) return false;
    int n_free = 0;
    for (int k = 0; k < n_kf[w]; k++) n_free += !h_kfs[w][k].fixed;
    if (pd * n_free + sco > kBigSolveMax || n_kf[w] >= (1 << 24)) return false;
    if (vio) {
      std::vector<char> in(n_kf[w], 0), outk(n_kf[w], 0);
      for (int t = 0; t < n_imu[w]; t++) {
        const int a = h_imu[w][t].kf_i, b = h_imu[w][t].kf_j;
        if (a < 0 || a >= n_kf[w] || b < 0 || b >= n_kf[w] || a == b || outk[a] || in[b]) return false;
        outk[a] = 1, in[b] = 1;
      }
    }
    const int nc = P->n_cams;
    if (nc < 0 || nc > 4 || (nc > 0 && !P->cams)) return false;
    const vieo_lba_obs* ob = h_obs[w];
    for (int i = 0; i < n_obs[w]; i++) {
      const int kfi = ob[i].kf & 0xFFFFFF, ci = (ob[i].kf >> 24) & 15;
      if (ob[i].mp < 0 || ob[i].mp >= n_mp[w] || ob[i].kf < 0 || kfi >= n_kf[w] || (i > 0 && ob[i].mp < ob[i - 1].mp) ||
          (nc == 0 ? ci != 0 : (ci >= nc || ob[i].ur >= 0)))
        return false;
    }
  }
  return true;
}

static int lba_run(const LbaShard* sh, const GbaMode* gba, int n_windows, const vieo_lba_params* const* params,
                   const vieo_lba_vio_params* const* vparams, const vieo_lba_keyframe* const* h_kfs,
                   const int* n_kf, const float* const* h_points, const uint8_t* const* h_close,
                   const int* n_mp, const vieo_lba_obs* const* h_obs, const int* n_obs,
                   const vieo_lba_imu_edge* const* h_imu, const int* n_imu, volatile const int* stop,
                   vieo_navstate* const* h_navs_out, float* const* h_points_out, uint8_t* const* h_erase,
                   vieo_lba_result* h_results, const vieo_lba_enc* const* encs = nullptr) {
  const bool vio = vparams != nullptr;
  const int pd = vio ? 15 : 6;
  const int sco = gba && gba->scale_opt ? 1 : 0;
  if (gba && gba->scale_out) *gba->scale_out = 1.0;
  // (configuration errors, the same on every rank of a job; `stop` is rank-local and asynchronous: a sharded run never
  // acts on its own copy -- the ranks' requests are summed in the exchanges the run makes anyway (the two agreements
  // at entry, the fourth scalar of every trial), so all ranks take the abort at the same point: Optimizer.cc:524-528,570-571)
  if (sh && (!vio || (!sh->fn && !sh->ctx) || !sh->d_buf)) {
    set_error("sharded local BA: visual-inertial windows only, with a reduction callback and buffer");
    return VIEO_E_INVALID;
  }
  bool shard_stop = false;  // a sharded run's collective view of the stop flag (latched)
  if (sh) {
    // every rank reaches this collective whatever its own arguments look like; the sum of the failure flags decides
    // for all of them (needs a device: a rank without one cannot take part in the job at all)
    int rcd = require_device();
    if (rcd != VIEO_OK) return rcd;
    const bool ok = sh->cap >= 1 && lba_args_ok(true, vio, gba != nullptr, n_windows, params, vparams, h_kfs, n_kf,
                                                h_points, h_close, n_mp, h_obs, n_obs, h_imu, n_imu, h_navs_out,
                                                h_points_out, h_erase, h_results, pd, sco);
    double sum = 1.0;
    if (sh->cap >= 1) {
      const int xrc = shard_agree(sh, ok, &sum, stop && *stop, &shard_stop);
      if (xrc != VIEO_OK) return xrc;
    }
    if (sum != 0.0) {
      set_error(ok ? "sharded local BA: another rank rejected its arguments, all ranks return"
                   : "sharded local BA: invalid arguments on this rank (all ranks return)");
      return VIEO_E_INVALID;
    }
  }
  ShardStagingGuard staging_guard{sh};  // from here to the second agreement every early return reports to the other ranks
  if (n_windows <= 0 || (!vio && !params) || !h_kfs || !n_kf || !h_points || !n_mp || !h_obs || !n_obs ||
      !h_navs_out || !h_points_out || !h_erase || !h_results || (vio && (!h_close || !h_imu || !n_imu)))
    return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  {
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (g_lba_stream && g_lba_stream_dev != cur) g_lba_stream = nullptr;  // the thread moved to another GPU
    g_lba_stream_dev = cur;
  }
  if (!g_lba_stream) {
    // VIEO_LBA_PRIORITY = -1 / 0 / 1: lowest / default / highest stream priority for the bundle-adjustment stream.
    // Lowest by default: local mapping is the background thread of the reference and tracking must not wait for
    // it.  Next to a batched front end on another stream the bench step measured (30 steps, 205 windows per step):
    // highest 48.9 k frames/s with k_fast at 9.3 ms per launch and 0.17 ms per window; default 50.1 k / 7.6 / 0.79;
    // lowest 53.0 k / 7.7 (what k_fast takes alone) / 0.73 -- the windows' kernels are short and many and fill what the
    // front end leaves free instead of taking CUs from it.
    // One window beside a tracker is the other case: the sequential replay measured 1.03 ms per frame with the stream at
    // the DEFAULT priority against 1.08 at the lowest and 1.07 at the highest (the solve finishes sooner and the tracker
    // waits less for its write-back): the LocalMapping thread of such a host asks for it with vieo_lba_set_stream_priority(0).
    int lo = 0, hi = 0;
    const char* e = getenv("VIEO_LBA_PRIORITY");
    const int want = g_lba_priority != -2 ? g_lba_priority : (e ? atoi(e) : -1);
    const char* cm = getenv("VIEO_LBA_CU_MASK");  // experiment: "first,count" -> the engine's stream on those CUs only
    int cu_first = 0, cu_count = 0;
    if (cm && sscanf(cm, "%d,%d", &cu_first, &cu_count) == 2 && cu_count > 0) {
      uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = cu_first; i < cu_first + cu_count && i < 256; i++) mask[i >> 5] |= 1u << (i & 31);
      VIEO_HIP_CHECK(hipExtStreamCreateWithCUMask(&g_lba_stream, 8, mask));
    } else if (want == 0 || hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || lo == hi)
      VIEO_HIP_CHECK(hipStreamCreateWithFlags(&g_lba_stream, hipStreamNonBlocking));
    else
      VIEO_HIP_CHECK(hipStreamCreateWithPriority(&g_lba_stream, hipStreamNonBlocking, want < 0 ? lo : hi));
  }
  hipStream_t st = g_lba_stream;
  // VIEO_LBA_TIMING=1: host phases of a call on stderr (staging, rounds, results)
  static const bool host_timing = getenv("VIEO_LBA_TIMING") != nullptr;
  const auto t_enter = std::chrono::steady_clock::now();
  auto ms_since = [](std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
  };
  const int W = n_windows;
  std::vector<WinHost> win(W);
  std::vector<LbaDev> devs(W);
  const bool stopped0 = sh ? shard_stop : (stop && *stop);
  int n_live = 0;
  for (int w = 0; w < W; w++) {
    WinHost& H = win[w];
    if (vio) {
      if (!vparams[w] || (!h_close[w] && !gba) || n_imu[w] < 0 || (n_imu[w] > 0 && !h_imu[w])) return VIEO_E_INVALID;
      H.VP = vparams[w], H.P = &vparams[w]->base, H.n_imu = n_imu[w];
    } else {
      H.P = params[w];
      if (encs && encs[w]) {  // EdgeEncNavStatePR of a vision-only window (Optimizer.cc:2008-2042, 1401-1438)
        if (encs[w]->n_edges < 0 || (encs[w]->n_edges > 0 && !encs[w]->edges)) return VIEO_E_INVALID;
        H.ENC = encs[w], H.n_imu = encs[w]->n_edges;
      }
    }
    H.n_kf = n_kf[w], H.n_mp = n_mp[w], H.n_obs = n_obs[w];
    H.R = &h_results[w];
    if (!H.P || !h_kfs[w] || H.n_kf <= 0 || H.n_mp < 0 || H.n_obs < 0 || !h_navs_out[w] ||
        (!sh && (H.n_mp == 0 || H.n_obs == 0)) || (H.n_mp > 0 && (!h_points[w] || !h_points_out[w])) ||
        (H.n_obs > 0 && (!h_obs[w] || !h_erase[w])))
      return VIEO_E_INVALID;  // a rank of a sharded run may own no point of a window: it still takes part in every exchange
    memset(H.R, 0, sizeof(*H.R));
    {
      int n_free = 0;
      for (int k = 0; k < H.n_kf; k++) n_free += !h_kfs[w][k].fixed;
      if (pd * n_free + sco > kBigSolveMax || H.n_kf >= (1 << 24)) {
        set_error("bundle adjustment: %d free key frames exceed the reduced-system limit of %d unknowns", n_free,
                  kBigSolveMax);
        return VIEO_E_CAPACITY;
      }
    }
    if (vio || H.ENC) {  // a chain: at most one pre-integration into and one out of every key frame
      std::vector<char> in(H.n_kf, 0), outk(H.n_kf, 0);
      for (int t = 0; t < H.n_imu; t++) {
        const int a = vio ? h_imu[w][t].kf_i : H.ENC->edges[t].kf_i, b = vio ? h_imu[w][t].kf_j : H.ENC->edges[t].kf_j;
        if (a < 0 || a >= H.n_kf || b < 0 || b >= H.n_kf || a == b || outk[a] || in[b]) {
          set_error("local BA: the inertial / encoder edges must chain the key frames");
          return VIEO_E_INVALID;
        }
        outk[a] = 1, in[b] = 1;
      }
    }
    for (int k = 0; k < H.n_kf; k++) h_navs_out[w][k] = h_kfs[w][k].nav;
    if (H.n_mp > 0) memcpy(h_points_out[w], h_points[w], (size_t)H.n_mp * 12);
    if (H.n_obs > 0) memset(h_erase[w], 0, H.n_obs);
    bool any_free = false;
    for (int k = 0; k < H.n_kf; k++) any_free |= !h_kfs[w][k].fixed;
    if (sco) any_free = true;  // bdimPoses = true with the scale vertex (Optimizer.cc:850)
    if (!any_free) {
      H.R->status = VIEO_LBA_NO_FREE_POSE;  // Optimizer.cc:1993
      H.skip = true, H.stage = 2;
    } else if (stopped0) {
      H.R->status = VIEO_LBA_ABORTED;
      H.skip = true, H.stage = 2;
    } else
      n_live++;
    const vieo_lba_obs* ob = h_obs[w];
    const int nc = H.P->n_cams;
    if (nc < 0 || nc > 4 || (nc > 0 && !H.P->cams)) {
      set_error("local BA: n_cams must be 0..4");
      return VIEO_E_INVALID;
    }
    (void)ob;  // the observations are checked while they are staged (fill_window)
  }
  if (!n_live) {  // (key frames and stop state are replicated: the ranks of a sharded run all leave here together)
    staging_guard.done = true;
    return VIEO_OK;
  }
  // ---- arena layout: [inputs | results (kf, X, erase) | zero-initialised | scratch]
  size_t arena = 0;
  auto take = [&](size_t bytes) {
    const size_t off = arena;
    arena += (bytes + 255) / 256 * 256;
    return off;
  };
  struct Off {
    size_t obs, mp_first, mp_count, kf_edge_first, kf_edge_idx, imu, kf_in, kf_out, close, ocam, kf, X, erase, scl, level, err;
  };
  std::vector<Off> off(W);
  for (int w = 0; w < W; w++) {
    const WinHost& H = win[w];
    if (H.skip) continue;
    Off& o = off[w];
    o.obs = take((size_t)H.n_obs * sizeof(vieo_lba_obs));
    o.mp_first = take((size_t)H.n_mp * 4), o.mp_count = take((size_t)H.n_mp * 4);
    o.kf_edge_first = take((size_t)(H.n_kf + 1) * 4), o.kf_edge_idx = take((size_t)H.n_obs * 4);
    o.ocam = take(H.n_obs);
    if (vio || H.ENC) {
      o.imu = take((size_t)std::max(H.n_imu, 1) * sizeof(LbaImu));
      o.kf_in = take((size_t)H.n_kf * 4), o.kf_out = take((size_t)H.n_kf * 4), o.close = take(H.n_mp);
    }
  }
  const size_t res_begin = arena;
  for (int w = 0; w < W; w++) {
    WinHost& H = win[w];
    if (H.skip) continue;
    Off& o = off[w];
    o.kf = take((size_t)H.n_kf * sizeof(LbaKf)), o.X = take((size_t)H.n_mp * 24), o.erase = take(H.n_obs);
    o.scl = take(16);
    H.o_kf = o.kf, H.o_X = o.X, H.o_erase = o.erase, H.o_scl = o.scl;
  }
  const size_t res_end = arena;
  for (int w = 0; w < W; w++) {
    const WinHost& H = win[w];
    if (H.skip) continue;
    off[w].level = take(H.n_obs), off[w].err = take((size_t)H.n_obs * 24);
  }
  const size_t zero_end = arena;
  if ((rc = g_stage.ensure(res_end)) != VIEO_OK) return rc;
  uint8_t* hs = (uint8_t*)g_stage.p;
  int max_obs = 0, max_mp = 0, max_kf = 0, max_nf = 0, max_imu = 0, max_chunks = 1;
  bool any_multicam = false;
  for (int w = 0; w < W; w++) {
    if (win[w].skip) continue;
    int nf = 0;
    for (int k = 0; k < win[w].n_kf; k++) nf += !h_kfs[w][k].fixed;
    max_nf = std::max(max_nf, nf), max_mp = std::max(max_mp, win[w].n_mp);
  }
  // Schur GEMM decomposition: 64x64 block-tiles (upper) x K splits.  The number of splits is a function of the window
  // alone (about 8 chunks of 16 landmarks per workgroup, VIEO_LBA_CPS), so that the summation order -- and with it the
  // window's result -- does not depend on what the window is batched with; a mixed batch (ordinary windows of one
  // tile next to bLarge ones of six) launches the largest tile x split count and the others' workgroups exit.
  // Measured per call of 205 windows, VIEO_LBA_CPS = 3 / 6 / 12 / 24: Schur 4.9 / 4.6 / 4.7 / 6.1 ms, k_lba_assemble
  // (which sums the partials) 2.1 / 1.6 / 1.2 / 1.1 ms.  The full BA keeps few splits: its tiles are many and
  // mostly skipped (k_lba_occ).
  static const int cps_target = [] {
    const char* e = getenv("VIEO_LBA_CPS");
    return e && atoi(e) > 0 ? atoi(e) : 8;
  }();
  auto schur_tiles = [&](int nf) {
    const int npm = 6 * nf + sco, RB = (npm + 63) / 64, CB = (npm + 64) / 64;
    return RB * CB - RB * (RB - 1) / 2;
  };
  auto schur_ksplit = [&](int nf, int nmp) {
    const int nch = (nmp + kChunkLm - 1) / kChunkLm, nbt = std::max(1, schur_tiles(nf));
    if (gba) return std::max(1, std::min(std::min(nch, 16), 768 / nbt));
    return std::max(1, std::min(std::min((nch + cps_target - 1) / cps_target, 32), std::max(1, 512 / nbt)));
  };
  int schur_grid = 0, schur_grid_off = 0;  // diagonal / off-diagonal tiles x splits, largest over the windows
  std::vector<size_t> scratch_off(W);
  struct Scr {
    size_t kf_bak, X_bak, mp_act, BB, Bs, Sp, Hll, bl, Hpp, Hs, bp, bs, xp, part0, part, part_m, pmax, kf_list, tab, Ae, gchi0,
        gchi, bfull, Hb, Wp, big_fail, kf_act, occ, sc_sys, psc, part_t, tail_cnt, chunk_first, chunk_kf, chunk_cnt, chunk_part;
    int nb;
  };
  std::vector<Scr> scr(W);
  // ---- staging, part 1 (sequential, cheap): scratch layout of every window and the scalar part of its descriptor
  for (int w = 0; w < W; w++) {
    WinHost& H = win[w];
    if (H.skip) continue;
    const vieo_lba_keyframe* kfs = h_kfs[w];
    int nf = 0;
    for (int k = 0; k < H.n_kf; k++) nf += !kfs[k].fixed;
    // scratch
    Scr& s = scr[w];
    const int npm = 6 * nf + sco;  // rows of the visual system: PR blocks (+ the scale vertex)
    s.kf_bak = take((size_t)H.n_kf * sizeof(LbaKf)), s.X_bak = take((size_t)H.n_mp * 24);
    s.mp_act = take(H.n_mp);
    const int sp_rows = (npm + 63) / 64 * 64, ldS = (npm + 64) / 64 * 64;
    s.BB = take(((size_t)H.n_obs + 1) * 144);  // + the zero block
    s.Bs = sco ? take((size_t)std::max(H.n_mp, 1) * 24) : 0;
    const int ksplit = schur_ksplit(nf, H.n_mp);
    {
      const int RBw = (npm + 63) / 64;
      schur_grid = std::max(schur_grid, RBw * ksplit);
      schur_grid_off = std::max(schur_grid_off, (schur_tiles(nf) - RBw) * ksplit);
    }
    s.Sp = take((size_t)ksplit * sp_rows * ldS * 8);
    s.Hll = take((size_t)H.n_mp * 72), s.bl = take((size_t)H.n_mp * 24);
    const int npf = pd * nf + sco;  // full reduced system
    s.Hpp = take((size_t)std::max(nf, 1) * 36 * 8), s.Hs = take((size_t)npf * npf * 8);
    s.nb = (npf + 1 + kNB - 1) / kNB * kNB;  // + the right-hand-side row
    s.Hb = s.Wp = s.big_fail = 0;
    if (solver_class(npf) == 2) {
      s.Hb = take((size_t)s.nb * s.nb * 8), s.Wp = take((size_t)s.nb * kNB * 8), s.big_fail = take(256);
    } else if (solver_class(npf) == 1) {
      s.nb = (npf + 16) >> 4;  // 16 x 16 blocks of the bordered matrix
      s.Hb = take(ldg_scratch_doubles(s.nb) * 8);
    }
    s.bp = take((size_t)npm * 8), s.bs = take((size_t)npf * 8), s.xp = take((size_t)npf * 8);
    s.bfull = take((size_t)npf * 8);
    s.Ae = take((size_t)std::max(H.n_imu, 1) * 930 * 8);
    s.gchi0 = take((size_t)std::max(H.n_imu, 1) * 8), s.gchi = take((size_t)std::max(H.n_imu, 1) * 8);
    s.part0 = take((size_t)((H.n_obs + 255) / 256) * 8), s.part = take((size_t)((H.n_obs + 255) / 256) * 8);
    s.part_m = take((size_t)((H.n_mp + 63) / 64) * 8), s.pmax = take((size_t)((H.n_mp + 63) / 64) * 8);
    s.part_t = take((size_t)std::max((H.n_mp + 63) / 64, 1) * 8), s.tail_cnt = take(256);
    {  // chunks of the key frames' edge lists: sum over the free key frames of ceil(edges / kBuildChunk) <= this bound
      const size_t nchk = (size_t)H.n_obs / kBuildChunk + nf + 1;
      s.chunk_first = take((size_t)(nf + 1) * 4), s.chunk_cnt = take((size_t)(nf + 1) * 4);
      s.chunk_kf = take(nchk * 4), s.chunk_part = take(nchk * 33 * 8);
      max_chunks = std::max(max_chunks, (int)nchk);
    }
    s.kf_list = take((size_t)H.n_kf * 4), s.tab = take((size_t)std::max(nf, 1) * H.n_mp * 4);
    s.sc_sys = take((size_t)(6 * nf + 2) * 8), s.psc = take((size_t)((H.n_mp + 63) / 64) * 16);
    s.kf_act = take((size_t)H.n_kf * 4);
    s.occ = take((size_t)((npm + 64) / 64) * ((H.n_mp + kChunkLm - 1) / kChunkLm));
    LbaDev& D = devs[w];
    memset(&D, 0, sizeof(D));
    D.n_obs = H.n_obs, D.n_mp = H.n_mp, D.n_kf = H.n_kf, D.nf_cap = nf;
    D.ldS = ldS, D.sp_stride = (size_t)sp_rows * ldS, D.ksplit = ksplit;
    D.cam.fx = H.P->fx, D.cam.fy = H.P->fy, D.cam.cx = H.P->cx, D.cam.cy = H.P->cy, D.cam.bf = H.P->bf;
    memcpy(D.cam.Rcb, H.P->Rcb, 72);
    memcpy(D.cam.tcb, H.P->tcb, 24);
    D.n_cams = H.P->n_cams;
    any_multicam |= H.P->n_cams > 0;
    for (int ci = 0; ci < H.P->n_cams; ci++) {
      const vieo_camera& c = H.P->cams[ci];
      CamD& d = D.cams[ci];
      d.fx = c.fx, d.fy = c.fy, d.cx = c.cx, d.cy = c.cy, d.bf = 0;
      memcpy(d.Rcb, c.Rcb, 72), memcpy(d.tcb, c.tcb, 24);
      d.model = c.model, d.num_k = c.model == VIEO_CAM_RADTAN ? c.num_k : 0;
      if (c.model < 0 || c.model > 2 || (c.model == VIEO_CAM_RADTAN && (c.num_k < 2 || c.num_k > 6))) {
        set_error("local BA: camera %d has an unknown model or coefficient count", ci);
        return VIEO_E_INVALID;
      }
      for (int q = 0; q < 8; q++) d.k[q] = (double)c.dist[q];
    }
    // thHuberMono = sqrt(5.991) in the local BAs, thHuber2D = sqrt(5.99) in the global ones (Optimizer.cc:1063,1445)
    D.dMono = (double)(float)sqrt(gba ? 5.99 : 5.991), D.dStereo = (double)(float)sqrt(7.815);
    D.pd = pd, D.n_imu = H.n_imu;
    D.scale_opt = sco;
    D.use_occ = gba ? 1 : 0;
    D.solver = solver_class(npf);
    if (vio) {  // const float chi2Mono = 5.991; 1.5 * chi2Mono; literal 7.815 (Optimizer.cc:347,603-620)
      D.thMono = (double)5.991f, D.thMonoClose = 1.5 * (double)5.991f, D.thStereo = 7.815;
      memcpy(D.gw, H.VP->gw, 24);
      memcpy(D.qRbe, H.VP->qRbe, 32), memcpy(D.pbe, H.VP->pbe, 24);
      D.th_dist_far = (!gba && H.VP->th_dist_far > 0 && std::isfinite(H.VP->th_dist_far)) ? (double)H.VP->th_dist_far : 0.0;
      H.prelevel_pending = !gba;
    } else {
      D.thMono = D.thMonoClose = 5.991, D.thStereo = 7.815;
      if (H.ENC) memcpy(D.qRbe, H.ENC->qRbe, 32), memcpy(D.pbe, H.ENC->pbe, 24);
    }
    max_imu = std::max(max_imu, H.n_imu);
    max_obs = std::max(max_obs, H.n_obs), max_mp = std::max(max_mp, H.n_mp);
    max_kf = std::max(max_kf, H.n_kf), max_nf = std::max(max_nf, nf);
    H.iters = gba ? gba->iterations : H.P->its0;
    H.phase = H.iters > 0 ? 0 : 2;
  }
  // ---- staging, part 2: the inputs of every window into the pinned copy of the arena (index structures, key frames,
  // inertial edges with their information matrices, points).  Windows are independent and the work is memory copies,
  // so a large batch is split over a few host threads (6.6 ms on one thread for the 103 windows of a bench step).
  const double ms_layout = ms_since(t_enter);
  auto fill_window = [&](int w) -> int {
    WinHost& H = win[w];
    const Off& o = off[w];
    const vieo_lba_obs* ob = h_obs[w];
    const vieo_lba_keyframe* kfs = h_kfs[w];
    // host-side index structures, written straight into the pinned staging copy of the arena
    {  // the device sees plain key-frame indices; the camera index travels in its own byte array
      vieo_lba_obs* so = (vieo_lba_obs*)(hs + o.obs);
      const int nc = H.P->n_cams;
      for (int i = 0; i < H.n_obs; i++) {
        const int kfi = ob[i].kf & 0xFFFFFF, ci = (ob[i].kf >> 24) & 15;
        if (ob[i].mp < 0 || ob[i].mp >= H.n_mp || ob[i].kf < 0 || kfi >= H.n_kf || (i > 0 && ob[i].mp < ob[i - 1].mp) ||
            (nc == 0 ? ci != 0 : (ci >= nc || ob[i].ur >= 0))) {
          set_error("vieo_local_bundle_adjustment: observations must be sorted by map point, key frame and camera "
                    "indices in range, distorted observations monocular");
          return VIEO_E_INVALID;
        }
        so[i] = ob[i];
        so[i].kf = kfi;
        hs[o.ocam + i] = (uint8_t)ci;
      }
      ob = so;
    }
    int* mp_first = (int*)(hs + o.mp_first);
    int* mp_count = (int*)(hs + o.mp_count);
    int* kf_first = (int*)(hs + o.kf_edge_first);
    int* kf_idx = (int*)(hs + o.kf_edge_idx);
    memset(mp_first, 0, (size_t)H.n_mp * 4), memset(mp_count, 0, (size_t)H.n_mp * 4);
    memset(kf_first, 0, (size_t)(H.n_kf + 1) * 4);
    for (int i = 0; i < H.n_obs; i++) {
      const int m = ob[i].mp;
      if (mp_count[m] == 0) mp_first[m] = i;
      mp_count[m]++;
      kf_first[ob[i].kf + 1]++;
    }
    for (int k = 0; k < H.n_kf; k++) kf_first[k + 1] += kf_first[k];
    std::vector<int> fill(kf_first, kf_first + H.n_kf);
    for (int i = 0; i < H.n_obs; i++) kf_idx[fill[ob[i].kf]++] = i;
    LbaKf* kf = (LbaKf*)(hs + o.kf);
    for (int k = 0; k < H.n_kf; k++) {
      memcpy(kf[k].p, kfs[k].nav.p, 24);
      kf[k].qw = kfs[k].nav.q[0], kf[k].qx = kfs[k].nav.q[1];
      kf[k].qy = kfs[k].nav.q[2], kf[k].qz = kfs[k].nav.q[3];
      kf[k].col = -1, kf[k].fixed = kfs[k].fixed ? 1 : 0;
      memcpy(kf[k].v, kfs[k].nav.v, 24), memcpy(kf[k].dbg, kfs[k].nav.dbg, 24), memcpy(kf[k].dba, kfs[k].nav.dba, 24);
      memcpy(kf[k].bg, kfs[k].nav.bg, 24), memcpy(kf[k].ba, kfs[k].nav.ba, 24);
    }
    if (vio) {  // inertial edges (Optimizer.cc:226-311)
      LbaImu* im = (LbaImu*)(hs + o.imu);
      int* kin = (int*)(hs + o.kf_in);
      int* kout = (int*)(hs + o.kf_out);
      for (int k = 0; k < H.n_kf; k++) kin[k] = kout[k] = -1;
      for (int t = 0; t < H.n_imu; t++) {
        const vieo_lba_imu_edge& e = h_imu[w][t];
        LbaImu& d = im[t];
        d.i = e.kf_i, d.j = e.kf_j, d.M = e.imu;
        const bool bfixedkf = kfs[e.kf_i].fixed != 0;
        d.has_imu = e.imu.dt != 0, d.robust = gba ? gba->robust != 0 : (bfixedkf || H.VP->rec_init);
        memset(d.InfoI, 0, sizeof(d.InfoI));
        if (d.has_imu) {
          if (!inverse9(e.imu.Sigma, d.InfoI)) {
            set_error("visual-inertial local BA: singular pre-integration covariance");
            return VIEO_E_INVALID;
          }
          if (bfixedkf)
            for (int q = 0; q < 81; q++) d.InfoI[q] *= 1e-2;
        }
        double deltatij = e.imu.dt ? e.imu.dt : e.dt_kf;
        const float EPS_MIN_DT = 1e-6f;
        if (deltatij <= EPS_MIN_DT) deltatij = 15;  // Optimizer.cc:271-275
        d.infoBg = H.VP->inv_sigma_bg2 / deltatij * (bfixedkf ? 1e-2 : 1.0);
        d.infoBa = H.VP->inv_sigma_ba2 / deltatij * (bfixedkf ? 1e-2 : 1.0);
        d.has_enc = e.enc.dt != 0, d.enc_robust = gba ? gba->robust != 0 : 1;
        memset(d.InfoE, 0, sizeof(d.InfoE)), memset(d.measE, 0, sizeof(d.measE));
        if (d.has_enc) {
          memcpy(d.measE, e.enc.delx, 48);
          if (!inverse_n(e.enc.Sigma, d.InfoE, 6)) {
            set_error("visual-inertial local BA: singular encoder covariance");
            return VIEO_E_INVALID;
          }
          if (bfixedkf)
            for (int q = 0; q < 36; q++) d.InfoE[q] *= 1e-2;
        }
        kout[e.kf_i] = t, kin[e.kf_j] = t;
      }
      if (h_close[w])
        memcpy(hs + o.close, h_close[w], H.n_mp);
      else
        memset(hs + o.close, 0, H.n_mp);
    } else if (H.ENC) {  // encoder edges only: no inertial residual, no bias random walk
      LbaImu* im = (LbaImu*)(hs + o.imu);
      int* kin = (int*)(hs + o.kf_in);
      int* kout = (int*)(hs + o.kf_out);
      for (int k = 0; k < H.n_kf; k++) kin[k] = kout[k] = -1;
      for (int t = 0; t < H.n_imu; t++) {
        const vieo_lba_enc_edge& e = H.ENC->edges[t];
        LbaImu& d = im[t];
        memset(&d, 0, sizeof(d));
        d.i = e.kf_i, d.j = e.kf_j;
        d.has_enc = e.enc.dt != 0, d.enc_robust = gba ? gba->robust != 0 : 1;
        if (d.has_enc) {
          memcpy(d.measE, e.enc.delx, 48);
          if (!inverse_n(e.enc.Sigma, d.InfoE, 6)) {
            set_error("local BA: singular encoder covariance");
            return VIEO_E_INVALID;
          }
          if (kfs[e.kf_i].fixed)  // Optimizer.cc:2030-2033
            for (int q = 0; q < 36; q++) d.InfoE[q] *= 1e-2;
        }
        kout[e.kf_i] = t, kin[e.kf_j] = t;
      }
      memset(hs + o.close, 0, H.n_mp);
    }
    double* X = (double*)(hs + o.X);
    for (int i = 0; i < H.n_mp * 3; i++) X[i] = (double)h_points[w][i];
    memset(hs + o.erase, 0, H.n_obs);
    ((double*)(hs + o.scl))[0] = ((double*)(hs + o.scl))[1] = 1.0;  // pvScale->setEstimate(1.) (Optimizer.cc:845)
    return VIEO_OK;
  };
  {
    std::vector<int> live;
    for (int w = 0; w < W; w++)
      if (!win[w].skip) live.push_back(w);
    const int hw = (int)std::thread::hardware_concurrency();
    const int nt = std::max(1, std::min(std::min(8, hw > 0 ? hw : 1), (int)live.size() / 8));
    std::vector<int> rcs(live.size(), VIEO_OK);
    if (nt <= 1) {
      for (size_t i = 0; i < live.size(); i++)
        if ((rcs[i] = fill_window(live[i])) != VIEO_OK) return rcs[i];
    } else {
      std::vector<std::thread> th;
      for (int t = 0; t < nt; t++)
        th.emplace_back([&, t] {
          for (size_t i = t; i < live.size(); i += nt) rcs[i] = fill_window(live[i]);
        });
      for (auto& x : th) x.join();
      for (size_t i = 0; i < live.size(); i++)
        if (rcs[i] != VIEO_OK) return fill_window(live[i]);  // again on this thread: the error text is thread-local
    }
  }
  const double ms_filled = ms_since(t_enter);
  const size_t small_bytes = (size_t)W * (sizeof(LbaDev) + sizeof(WinCtl) + sizeof(WinOut) + sizeof(int));
  if ((rc = g_arena.ensure(arena)) != VIEO_OK) return rc;
  if ((rc = g_small.ensure(small_bytes)) != VIEO_OK) return rc;
  if ((rc = g_small_h.ensure((size_t)W * (sizeof(WinCtl) + sizeof(WinOut) + 32))) != VIEO_OK) return rc;
  uint8_t* base = g_arena.as<uint8_t>();
  for (int w = 0; w < W; w++) {
    if (win[w].skip) continue;
    const Off& o = off[w];
    const Scr& s = scr[w];
    LbaDev& D = devs[w];
    D.obs = (const vieo_lba_obs*)(base + o.obs);
    D.mp_first = (const int*)(base + o.mp_first), D.mp_count = (const int*)(base + o.mp_count);
    D.kf_edge_first = (const int*)(base + o.kf_edge_first), D.kf_edge_idx = (const int*)(base + o.kf_edge_idx);
    D.kf = (LbaKf*)(base + o.kf), D.X = (double*)(base + o.X), D.erase = base + o.erase;
    D.level = base + o.level, D.err = (double*)(base + o.err);
    D.ocam = base + o.ocam;
    D.kf_bak = (LbaKf*)(base + s.kf_bak), D.X_bak = (double*)(base + s.X_bak), D.mp_act = base + s.mp_act;
    D.CB = (double*)(base + s.BB), D.Bs = (double*)(base + s.Bs), D.Sp = (double*)(base + s.Sp);
    D.Hll = (double*)(base + s.Hll), D.bl = (double*)(base + s.bl);
    D.Hpp = (double*)(base + s.Hpp), D.Hs = (double*)(base + s.Hs), D.bp = (double*)(base + s.bp);
    D.Hb = (double*)(base + s.Hb), D.Wp = (double*)(base + s.Wp), D.big_fail = (int*)(base + s.big_fail);
    D.nb = s.nb;
    D.bs = (double*)(base + s.bs), D.xp = (double*)(base + s.xp);
    D.part0 = (double*)(base + s.part0), D.part = (double*)(base + s.part);
    D.part_m = (double*)(base + s.part_m), D.pmax = (double*)(base + s.pmax);
    D.part_t = (double*)(base + s.part_t), D.tail_cnt = (int*)(base + s.tail_cnt);
    D.fold_kernel = W <= 4 ? 0 : 1;
    D.chunk_first = (int*)(base + s.chunk_first), D.chunk_cnt = (int*)(base + s.chunk_cnt);
    D.chunk_kf = (int*)(base + s.chunk_kf), D.chunk_part = (double*)(base + s.chunk_part);
    D.kf_list = (int*)(base + s.kf_list), D.tab = (int*)(base + s.tab);
    D.kf_act = (int*)(base + s.kf_act), D.occ = base + s.occ;
    D.bfull = (double*)(base + s.bfull), D.Ae = (double*)(base + s.Ae);
    D.gchi0 = (double*)(base + s.gchi0), D.gchi = (double*)(base + s.gchi);
    D.scl = (double*)(base + o.scl), D.sc_sys = (double*)(base + s.sc_sys), D.psc = (double*)(base + s.psc);
    if (vio || win[w].ENC) {
      D.imu = (const LbaImu*)(base + o.imu), D.close = base + o.close;
      D.kf_in = (const int*)(base + o.kf_in), D.kf_out = (const int*)(base + o.kf_out);
    }
  }
  size_t shard_sys = 0;
  if (sh) {
    for (int w = 0; w < W; w++) {
      if (win[w].skip) continue;
      devs[w].red = sh->d_buf + shard_sys;
      shard_sys += shard_sys_doubles(devs[w].nf_cap, sco);
    }
    if (shard_sys + 4 * (size_t)W > sh->cap) {
      set_error("sharded local BA: reduction buffer too small (%zu doubles needed)", shard_sys + 4 * (size_t)W);
      return VIEO_E_CAPACITY;
    }
    for (int w = 0; w < W; w++) devs[w].red_sc = sh->d_buf + shard_sys + 4 * (size_t)w;
  }
  LbaDev* dD = g_small.as<LbaDev>();
  WinCtl* dC = (WinCtl*)(dD + W);
  WinOut* dO = (WinOut*)(dC + W);
  int* dWins = (int*)(dO + W);  // the windows grouped by solver class: [class 0 | class 1 | class 2]
  WinCtl* ctl = (WinCtl*)g_small_h.p;
  WinOut* out = (WinOut*)(ctl + W);
  double* h_sc = (double*)(out + W);  // reduced scalars of a sharded run
  const double ms_described = ms_since(t_enter);
  VIEO_HIP_CHECK(hipMemcpyAsync(base, hs, res_end, hipMemcpyHostToDevice, st));
  VIEO_HIP_CHECK(hipMemsetAsync(base + res_end, 0, zero_end - res_end, st));
  VIEO_HIP_CHECK(hipMemcpyAsync(dD, devs.data(), (size_t)W * sizeof(LbaDev), hipMemcpyHostToDevice, st));
  VIEO_HIP_CHECK(hipMemsetAsync(dO, 0, (size_t)W * sizeof(WinOut), st));
  const int occ_max = ((6 * max_nf + sco + 64) / 64) * ((max_mp + kChunkLm - 1) / kChunkLm);
  // per solver class: the largest system of the class in this batch (0: the class is empty)
  int cls_max[3] = {0, 0, 0};
  for (int w = 0; w < W; w++)
    if (!win[w].skip) cls_max[devs[w].solver] = std::max(cls_max[devs[w].solver], pd * devs[w].nf_cap + sco);
  const bool big = cls_max[2] > 0, ldlt16 = cls_max[0] > 0, panels = cls_max[1] > 0;
  int cls_first[4] = {0, 0, 0, 0};
  std::vector<int> cls_order;  // (alive until the call returns: the copy below is asynchronous)
  for (int c = 0; c < 3; c++) {
    cls_first[c] = (int)cls_order.size();
    for (int w = 0; w < W; w++)
      if (!win[w].skip && devs[w].solver == c) cls_order.push_back(w);
  }
  cls_first[3] = (int)cls_order.size();
  if (!cls_order.empty())
    VIEO_HIP_CHECK(hipMemcpyAsync(dWins, cls_order.data(), cls_order.size() * sizeof(int), hipMemcpyHostToDevice, st));
  const int n_max_b = cls_max[2];
  const int nbg = (cls_max[1] + 16) >> 4;
  if (panels)
    VIEO_HIP_CHECK(hipFuncSetAttribute((const void*)k_lba_ldltg<kLdGThreads>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)ldg_lds_bytes(nbg)));
  const int nb16 = (cls_max[0] + 16) >> 4;
  if (ldlt16)
    VIEO_HIP_CHECK(hipFuncSetAttribute((const void*)k_lba_ldlt16<kLd16Threads>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)ld16_lds_bytes(nb16)));
  // (at least one workgroup each: an empty landmark shard still launches everything)
  const int ge = std::max(1, (max_obs + 255) / 256), gm = std::max(1, (max_mp + 255) / 256), gq = std::max(1, (max_mp + 63) / 64);
  const int gr = std::max(gm, (max_kf + 255) / 256);

  if (sh) {  // second agreement: every rank staged its windows (ShardStagingGuard reports the ones that did not)
    VIEO_HIP_CHECK(hipStreamSynchronize(st));
    double sum = 1.0;
    staging_guard.done = true;
    const int xrc = shard_agree(sh, true, &sum);
    if (xrc != VIEO_OK) return xrc;
    if (sum != 0.0) {
      set_error("sharded local BA: another rank failed while staging its windows, all ranks return");
      return VIEO_E_INVALID;
    }
  }
  // ---- lock-step rounds
  const double ms_staged = ms_since(t_enter);
  const auto t_rounds = std::chrono::steady_clock::now();
  int n_rounds = 0;
  double ms_wait = 0;
  static thread_local LbaKTimer KT;
  KT.begin(st);
  // dense count 2 np (np + 1) 3 n_mp of the visual part of a window's Schur complement
  auto schur_flops_of = [&](int w) {
    return win[w].skip ? 0.0 : 2.0 * (6 * devs[w].nf_cap) * (6 * devs[w].nf_cap + 1) * 3.0 * devs[w].n_mp;
  };
  // ---- the LM policy on the device (local BAs without an exchange step): VIEO_LBA_DEVICE_POLICY=1.  Off by default: it
  // passes the same parity tests, and it measured the SAME time as the host loop below (one window beside the tracker
  // 5.2 vs 5.2 ms, alone 3.56 vs 3.54): what it saves between two rounds (copy back, synchronise, copy up: ~35 us) the blind
  // round's extra kernels cost again (restore + classification + the stage-start group in the rounds where a stage may
  // end: 6-8 launches that find nothing to do, ~4 us each).  Round 6 put the decision and the rollback INSIDE the trial's
  // last launch (k_lba_tail's last workgroup: no policy launch, no restore launch, control words double-buffered by round
  // parity) -- and it still measured 5.1-5.2 ms against the host loop's 4.9 beside the tracker (tools/r6_lba_check.sh):
  // the rounds queued blind behind a finished stage and the stage-start groups cost more than the round trip they save.
  // The host loop keeps glibc's pow() in the damping update.
  // k_lba_tail (one launch for the tail of a trial) for calls of a few windows: one window beside the tracker 5.0 -> 4.9 ms,
  // W = 4 equal, but W = 16 / 64 windows 2.80 -> 3.02 / 5.07 -> 6.47 ms per call -- its residual pass runs four lanes per
  // point over the point's edges (that is what makes it independent of the other workgroups), which is latency-bound and
  // loses to k_lba_error's lane per edge once the launch ramps are amortised over many windows.
  // VIEO_LBA_FUSED_TAIL=0 forces the four-launch form everywhere (A/B runs, tests of both forms).
  static const int fused_tail_env = [] {
    const char* e = getenv("VIEO_LBA_FUSED_TAIL");
    return e ? atoi(e) : -1;
  }();
  // calls of a few windows: the tail's last workgroup folds (fused_tail); batches: the four-launch form (its per-edge
  // residual kernel has four times the tail's parallelism: the tail without the fold + k_lba_reduce over its partials,
  // VIEO_LBA_FUSED_TAIL=2, measured 3.9 against 2.9 ms of kernel time per 205-window step); =0: four launches everywhere
  const bool fused_tail = fused_tail_env != 0 && W <= 4;
  const bool split_tail = fused_tail_env == 2 && W > 4;
  static const int fused_build_env = [] {  // VIEO_LBA_FUSED_BUILD=0 / 1: both halves of k_lba_build in one launch (A/B runs)
    const char* e = getenv("VIEO_LBA_FUSED_BUILD");
    return e ? (atoi(e) != 0 ? 1 : 0) : -1;
  }();
  const bool fused_build = fused_build_env >= 0 ? fused_build_env != 0 : W <= 4;
  static const bool dev_policy_env = [] {
    const char* e = getenv("VIEO_LBA_DEVICE_POLICY");
    return e && atoi(e) != 0;
  }();
  const bool dev_policy = dev_policy_env && !sh && !gba && !KT.on;
  if (dev_policy) {
    static thread_local DevBuf g_pol;
    static thread_local PinnedBuf g_pol_h;
    if ((rc = g_pol.ensure((size_t)W * (sizeof(WinPol) + sizeof(WinCtl)))) != VIEO_OK || (rc = g_pol_h.ensure((size_t)W * (sizeof(WinPol) + sizeof(WinCtl)))) != VIEO_OK)
      return rc;
    WinPol* hp = (WinPol*)g_pol_h.p;
    WinCtl* hc = (WinCtl*)(hp + W);
    WinPol* dP = g_pol.as<WinPol>();
    // (k_lba_tail decides inside the round: the flags of a round and of the next one live in two arrays, by round parity)
    WinCtl* const cbuf[2] = {dC, (WinCtl*)(dP + W)};
    int min_first_stage_rounds = 1 << 30;
    for (int w = 0; w < W; w++) {
      const WinHost& H = win[w];
      WinPol& Q = hp[w];
      memset(&Q, 0, sizeof(Q));
      Q.stage = H.stage, Q.phase = H.phase, Q.iters = H.iters, Q.its1 = H.P->its1, Q.prelevel_pending = H.prelevel_pending ? 1 : 0;
      Q.vio = vio ? 1 : 0, Q.robust0 = 1, Q.lambda = H.lambda, Q.ni = 2;
      Q.lambda_init = vio ? H.VP->lambda_init : 0.0;
      if (!H.skip) min_first_stage_rounds = std::min(min_first_stage_rounds, std::max(1, std::min(H.iters, 3)));
    }
    VIEO_HIP_CHECK(hipMemcpyAsync(dP, hp, (size_t)W * sizeof(WinPol), hipMemcpyHostToDevice, st));
    // One round = the superset of what a window can ask for at that point: the kernels read the windows' flags and leave at
    // once where they have nothing to do.  The stage-start kernels (classification, active sets, initial chi2, lambda) are
    // in round 1 and in every round from the first one in which an optimize() can end (three iterations, or its0 of them)
    // until the host has seen every window inside its second optimize(); the policy kernel is told whether the round it
    // prepares has them and lets a window that needs them wait a round otherwise.  The classification kernel alone (final
    // erase flags) is in every round.
    auto begin_allowed = [&](int round, bool all_in_second) { return round == 1 || (!all_in_second && round > min_first_stage_rounds); };
    auto launch_round = [&](int round, bool with_begin, bool next_begin) -> int {
      WinCtl* const dC = fused_tail ? cbuf[round & 1] : cbuf[0];  // (shadows the one-array name used by the launches below)
      WinCtl* const dCn = cbuf[(round + 1) & 1];
      if (!fused_tail) hipLaunchKernelGGL(k_lba_restore, dim3(gr, W), dim3(256), 0, st, dD, dC);
      hipLaunchKernelGGL(k_lba_classify, dim3(ge, W), dim3(256), 0, st, dD, dC);
      if (round == 1 && vio)
        for (int ph = 0; ph < 3; ph++) hipLaunchKernelGGL(k_lba_prelevel, dim3(ge, W), dim3(256), 0, st, dD, dC, ph);
      if (with_begin) {
        hipLaunchKernelGGL(k_lba_zero, dim3(64, W), dim3(256), 0, st, dD, dC);
        hipLaunchKernelGGL(k_lba_begin, dim3(W), dim3(1024), 0, st, dD, dC, dO);
        hipLaunchKernelGGL(k_lba_error, dim3(ge, W), dim3(256), 0, st, dD, dC, 0);
      }
      auto build2 = [&](auto mc, auto sc) {  // point half, key-frame half
        constexpr bool MC = decltype(mc)::value, SC = decltype(sc)::value;
        if (fused_build) {
          hipLaunchKernelGGL((k_lba_build<MC, SC, 2>), dim3(gq + max_chunks + max_imu, W), dim3(256), 0, st, dD, dC, max_chunks, gq);
          if (W > 4) hipLaunchKernelGGL(k_lba_build_fold, dim3(std::max(1, max_nf), W), dim3(64), 0, st, dD, dC);
          return;
        }
        hipLaunchKernelGGL((k_lba_build<MC, SC, 0>), dim3(gq, W), dim3(256), 0, st, dD, dC, 0, 0);
        hipLaunchKernelGGL((k_lba_build<MC, SC, 1>), dim3(max_chunks, W), dim3(256), 0, st, dD, dC, max_chunks, 0);
        if (W > 4) hipLaunchKernelGGL(k_lba_build_fold, dim3(std::max(1, max_nf), W), dim3(64), 0, st, dD, dC);
      };
      if (any_multicam)
        build2(std::true_type(), std::false_type());
      else
        build2(std::false_type(), std::false_type());
      if (!fused_build && max_imu > 0) hipLaunchKernelGGL(k_lba_generic, dim3(max_imu, W), dim3(64), 0, st, dD, dC, 0);
      if (with_begin) hipLaunchKernelGGL(k_lba_lambda, dim3(W), dim3(256), 0, st, dD, dC, dO);
      hipLaunchKernelGGL(k_lba_schur<false>, dim3(std::max(1, schur_grid), W), dim3(256), 0, st, dD, dC, dO);
      if (schur_grid_off > 0) hipLaunchKernelGGL(k_lba_schur<true>, dim3(schur_grid_off, W), dim3(256), 0, st, dD, dC, dO);
      for (int c = 0; c < 3; c++) {
        const int nw = cls_first[c + 1] - cls_first[c];
        if (nw <= 0) continue;
        const unsigned gx = (unsigned)(((size_t)cls_max[c] * cls_max[c] + 255) / 256);
        hipLaunchKernelGGL(k_lba_assemble, dim3(gx, nw), dim3(256), 0, st, dD, dC, dO, dWins + cls_first[c]);
      }
      if (big && cls_first[3] > cls_first[2]) {
        const int nbm = (n_max_b + 1 + kNB - 1) / kNB * kNB, ntm = nbm / kNB;
        hipLaunchKernelGGL(k_big_init, dim3((unsigned)(((size_t)nbm * nbm + 255) / 256), W), dim3(256), 0, st, dD, dC);
        for (int k = 0; k < ntm; k++) {
          const int below = nbm - (k + 1) * kNB, m = ntm - k - 1;
          hipLaunchKernelGGL(k_big_panel, dim3(1 + (below + 255) / 256, W), dim3(256), 0, st, dD, dC, k);
          if (m > 0) hipLaunchKernelGGL(k_big_syrk, dim3(m * (m + 1) / 2, W), dim3(256), 0, st, dD, dC, k);
        }
        for (int sb = 0; sb < (n_max_b + kNB - 1) / kNB; sb++)
          hipLaunchKernelGGL(k_big_back_step, dim3(1 + (n_max_b + 255) / 256, W), dim3(256), 0, st, dD, dC, sb);
        hipLaunchKernelGGL(k_big_finish, dim3(W), dim3(256), 0, st, dD, dC, dO);
      }
      if (ldlt16 && cls_first[1] > cls_first[0])
        hipLaunchKernelGGL(k_lba_ldlt16<kLd16Threads>, dim3(W), dim3(kLd16Threads), ld16_lds_bytes(nb16), st, dD, dC, dO, nb16);
      if (panels && cls_first[2] > cls_first[1])
        hipLaunchKernelGGL(k_lba_ldltg<kLdGThreads>, dim3(W), dim3(kLdGThreads), ldg_lds_bytes(nbg), st, dD, dC, dO, nbg);
      if (fused_tail)  // ... with the window's LM decision and the rollback of a rejected trial in its last workgroup
        hipLaunchKernelGGL(k_lba_tail, dim3(gq + max_imu, W), dim3(256), 0, st, dD, dC, dO, gq, dP, dCn, next_begin ? 1 : 0, (stop && *stop) ? 1 : 0);
      else {
        hipLaunchKernelGGL(k_lba_update_points, dim3(gq, W), dim3(256), 0, st, dD, dC, dO);
        hipLaunchKernelGGL(k_lba_error, dim3(ge, W), dim3(256), 0, st, dD, dC, 1);
        if (max_imu > 0) hipLaunchKernelGGL(k_lba_generic, dim3(max_imu, W), dim3(64), 0, st, dD, dC, 1);
        hipLaunchKernelGGL(k_lba_reduce, dim3(W), dim3(256), 0, st, dD, dC, dO, 0);
      }
      VIEO_HIP_CHECK(hipGetLastError());
      return VIEO_OK;
    };
    const unsigned pg = (unsigned)((W + 63) / 64);
    int round = 1;
    bool all_in_second = false, with_begin = true;  // with_begin: what the policy kernel was told about the round it prepared
    hipLaunchKernelGGL(k_lba_policy, dim3(pg), dim3(64), 0, st, dP, fused_tail ? cbuf[1] : cbuf[0], dO, W, 1, 1, (stop && *stop) ? 1 : 0);
    constexpr int kBlindRounds = 3;
    for (bool done = false; !done;) {
      for (int k = 0; k < kBlindRounds; k++, round++) {
        const bool next_begin = begin_allowed(round + 1, all_in_second);
        if ((rc = launch_round(round, with_begin, next_begin)) != VIEO_OK) return rc;
        with_begin = next_begin;
        if (!fused_tail)
          hipLaunchKernelGGL(k_lba_policy, dim3(pg), dim3(64), 0, st, dP, dC, dO, W, 0, with_begin ? 1 : 0, (stop && *stop) ? 1 : 0);
        n_rounds++;
      }
      VIEO_HIP_CHECK(hipMemcpyAsync(hp, dP, (size_t)W * sizeof(WinPol), hipMemcpyDeviceToHost, st));
      VIEO_HIP_CHECK(hipMemcpyAsync(hc, fused_tail ? cbuf[round & 1] : cbuf[0], (size_t)W * sizeof(WinCtl), hipMemcpyDeviceToHost, st));
      {
        const auto t_w = std::chrono::steady_clock::now();
        VIEO_HIP_CHECK(hipStreamSynchronize(st));
        ms_wait += ms_since(t_w);
      }
      done = true, all_in_second = true;
      for (int w = 0; w < W; w++) {
        if (win[w].skip) continue;
        if (hp[w].stage < 2 || hc[w].flags != 0) done = false;
        if (hp[w].stage == 0 || (hp[w].stage == 1 && hp[w].phase != 1)) all_in_second = false;
      }
      if (round > 4000) {
        set_error("local BA: the device policy did not finish in %d rounds", round);
        return VIEO_E_HIP;
      }
    }
    for (int w = 0; w < W; w++) {
      WinHost& H = win[w];
      if (H.skip) continue;
      const WinPol& Q = hp[w];
      H.R->lm_iterations += Q.lm_iterations, H.R->lm_trials += Q.lm_trials;
      H.R->chi2_initial = Q.chi2_initial, H.R->chi2_final = Q.chi2_final;
      H.lastTrialChi = Q.lastTrialChi;
      if (Q.aborted) H.R->status = VIEO_LBA_ABORTED;
      H.stage = 2;
    }
  } else
  for (;;) {
    n_rounds++;
    const bool stop_now = sh ? shard_stop : (stop && *stop);
    const int stop_req = stop && *stop ? 1 : 0;  // (sharded: travels as ctl.pad -> the trial's fourth scalar)
    int any = 0;
    bool cls_trial[3] = {false, false, false};  // which solve kernels have a window this round
    for (int w = 0; w < W; w++) {
      WinHost& H = win[w];
      int f = 0;
      double lam = H.lambda;
      if (H.stage < 2 && H.phase == 2) {  // an optimize() is over
        if (H.need_restore) f |= LBA_RESTORE, H.need_restore = false;
        if (gba)
          H.stage = 2;
        else if (H.stage == 0 && !stop_now) {
          f |= LBA_CLASS0;
          H.stage = 1, H.iters = H.P->its1, H.phase = H.iters > 0 ? 0 : 2;
        } else {
          if (H.stage == 0) H.R->status = VIEO_LBA_ABORTED;  // stop flag between the two stages
          f |= LBA_CLASS1;
          H.stage = 2;
        }
      }
      if (H.stage < 2 && H.phase == 0) {
        f |= LBA_BEGIN | LBA_BUILD | LBA_TRIAL | ((gba ? gba->robust != 0 : H.stage == 0) ? LBA_ROBUST : 0);
        lam = vio && !gba ? H.VP->lambda_init : -1;  // setUserLambdaInit (Optimizer.cc:131-138)
        if (H.prelevel_pending) f |= LBA_PRELEVEL, H.prelevel_pending = false;
      } else if (H.stage < 2 && H.phase == 1) {
        f |= LBA_TRIAL | ((gba ? gba->robust != 0 : H.stage == 0) ? LBA_ROBUST : 0);
        if (H.need_build) f |= LBA_BUILD;
        if (H.need_restore) f |= LBA_RESTORE, H.need_restore = false;
      }
      ctl[w].flags = f, ctl[w].pad = sh ? stop_req : 0, ctl[w].lambda = lam;
      any |= f;
      if ((f & LBA_TRIAL) && !H.skip) cls_trial[devs[w].solver] = true;
    }
    if (!any) break;
    VIEO_HIP_CHECK(hipMemcpyAsync(dC, ctl, (size_t)W * sizeof(WinCtl), hipMemcpyHostToDevice, st));
    if (any & LBA_RESTORE) KT.launch(KC_OTHER, [&] { hipLaunchKernelGGL(k_lba_restore, dim3(gr, W), dim3(256), 0, st, dD, dC); });
    if (any & (LBA_CLASS0 | LBA_CLASS1)) KT.launch(KC_OTHER, [&] { hipLaunchKernelGGL(k_lba_classify, dim3(ge, W), dim3(256), 0, st, dD, dC); });
    if (any & LBA_PRELEVEL)
      for (int ph = 0; ph < 3; ph++) KT.launch(KC_OTHER, [&] { hipLaunchKernelGGL(k_lba_prelevel, dim3(ge, W), dim3(256), 0, st, dD, dC, ph); });
    if (any & LBA_BEGIN) {
      KT.launch(KC_BEGIN, [&] { hipLaunchKernelGGL(k_lba_zero, dim3(64, W), dim3(256), 0, st, dD, dC); });
      KT.launch(KC_BEGIN, [&] { hipLaunchKernelGGL(k_lba_begin, dim3(W), dim3(1024), 0, st, dD, dC, dO); });
      if (gba) KT.launch(KC_BEGIN, [&] { hipLaunchKernelGGL(k_lba_occ, dim3(std::max(1, (occ_max + 255) / 256), W), dim3(256), 0, st, dD, dC); });
      KT.launch(KC_ERROR, [&] { hipLaunchKernelGGL(k_lba_error, dim3(ge, W), dim3(256), 0, st, dD, dC, 0); });
    }
    if (any & LBA_BUILD) {
      auto build2 = [&](auto mc, auto sc) {  // point half, key-frame half
        constexpr bool MC = decltype(mc)::value, SC = decltype(sc)::value;
        if (fused_build) {  // one launch for both halves
          KT.launch(KC_BUILD, [&] { hipLaunchKernelGGL((k_lba_build<MC, SC, 2>), dim3(gq + max_chunks + max_imu, W), dim3(256), 0, st, dD, dC, max_chunks, gq); });
          if (W > 4) KT.launch(KC_BUILD, [&] { hipLaunchKernelGGL(k_lba_build_fold, dim3(std::max(1, max_nf), W), dim3(64), 0, st, dD, dC); });
          return;
        }
        KT.launch(KC_BUILD, [&] { hipLaunchKernelGGL((k_lba_build<MC, SC, 0>), dim3(gq, W), dim3(256), 0, st, dD, dC, 0, 0); });
        KT.launch(KC_BUILD, [&] { hipLaunchKernelGGL((k_lba_build<MC, SC, 1>), dim3(max_chunks, W), dim3(256), 0, st, dD, dC, max_chunks, 0); });
        if (W > 4) KT.launch(KC_BUILD, [&] { hipLaunchKernelGGL(k_lba_build_fold, dim3(std::max(1, max_nf), W), dim3(64), 0, st, dD, dC); });
      };
      if (sco) {
        if (any_multicam)
          build2(std::true_type(), std::true_type());
        else
          build2(std::false_type(), std::true_type());
        KT.launch(KC_BUILD, [&] { hipLaunchKernelGGL(k_lba_scale_fold, dim3(W), dim3(256), 0, st, dD, dC); });
      } else if (any_multicam)
        build2(std::true_type(), std::false_type());
      else
        build2(std::false_type(), std::false_type());
      if (!fused_build && max_imu > 0) KT.launch(KC_GENERIC, [&] { hipLaunchKernelGGL(k_lba_generic, dim3(max_imu, W), dim3(64), 0, st, dD, dC, 0); });
    }
    if (any & LBA_BEGIN) KT.launch(KC_BEGIN, [&] { hipLaunchKernelGGL(k_lba_lambda, dim3(W), dim3(256), 0, st, dD, dC, dO); });
    if (any & LBA_TRIAL) {
      KT.launch(KC_SCHUR, [&] { hipLaunchKernelGGL(k_lba_schur<false>, dim3(std::max(1, schur_grid), W), dim3(256), 0, st, dD, dC, dO); });
      if (schur_grid_off > 0)
        KT.launch(KC_SCHUR, [&] { hipLaunchKernelGGL(k_lba_schur<true>, dim3(schur_grid_off, W), dim3(256), 0, st, dD, dC, dO); });
      if (sh) {  // the one exchange step of the path: sum the reduced visual system over the ranks
        const int nv = 6 * max_nf + sco;
        KT.launch(KC_OTHER, [&] { hipLaunchKernelGGL(k_lba_pack, dim3((unsigned)((shard_sys_doubles(max_nf, sco) + 255) / 256), W), dim3(256), 0, st,
                           dD, dC); });
        (void)nv;
        if ((rc = shard_exchange(sh, sh->d_buf, shard_sys, st)) != VIEO_OK) return rc;
      }
      for (int c = 0; c < 3; c++) {
        const int nw = cls_first[c + 1] - cls_first[c];
        if (nw <= 0 || !cls_trial[c]) continue;
        const unsigned gx = (unsigned)(((size_t)cls_max[c] * cls_max[c] + 255) / 256);
        KT.launch(KC_ASSEMBLE, [&] { hipLaunchKernelGGL(k_lba_assemble, dim3(gx, nw), dim3(256), 0, st, dD, dC, dO, dWins + cls_first[c]); });
      }
      if (big && cls_trial[2]) {
        const int nbm = (n_max_b + 1 + kNB - 1) / kNB * kNB, ntm = nbm / kNB;
        KT.launch(KC_LDLT, [&] { hipLaunchKernelGGL(k_big_init, dim3((unsigned)(((size_t)nbm * nbm + 255) / 256), W), dim3(256), 0, st, dD, dC); });
        for (int k = 0; k < ntm; k++) {
          const int below = nbm - (k + 1) * kNB, m = ntm - k - 1;
          KT.launch(KC_LDLT, [&] { hipLaunchKernelGGL(k_big_panel, dim3(1 + (below + 255) / 256, W), dim3(256), 0, st, dD, dC, k); });
          if (m > 0) KT.launch(KC_LDLT, [&] { hipLaunchKernelGGL(k_big_syrk, dim3(m * (m + 1) / 2, W), dim3(256), 0, st, dD, dC, k); });
        }
        for (int sb = 0; sb < (n_max_b + kNB - 1) / kNB; sb++)
          KT.launch(KC_LDLT, [&] { hipLaunchKernelGGL(k_big_back_step, dim3(1 + (n_max_b + 255) / 256, W), dim3(256), 0, st, dD, dC, sb); });
        KT.launch(KC_LDLT, [&] { hipLaunchKernelGGL(k_big_finish, dim3(W), dim3(256), 0, st, dD, dC, dO); });
      }
      if (ldlt16 && cls_trial[0])
        KT.launch(KC_LDLT, [&] { hipLaunchKernelGGL(k_lba_ldlt16<kLd16Threads>, dim3(W), dim3(kLd16Threads), ld16_lds_bytes(nb16), st, dD, dC, dO, nb16); });
      if (panels && cls_trial[1])
        KT.launch(KC_LDLT, [&] { hipLaunchKernelGGL(k_lba_ldltg<kLdGThreads>, dim3(W), dim3(kLdGThreads), ldg_lds_bytes(nbg), st, dD, dC, dO, nbg); });
      if (fused_tail)
        KT.launch(KC_UPDATE, [&] { hipLaunchKernelGGL(k_lba_tail, dim3(gq + max_imu, W), dim3(256), 0, st, dD, dC, dO, gq, (WinPol*)nullptr, (WinCtl*)nullptr, 0, 0); });
      else if (split_tail) {
        KT.launch(KC_UPDATE, [&] { hipLaunchKernelGGL(k_lba_tail, dim3(gq + max_imu, W), dim3(256), 0, st, dD, dC, dO, gq, (WinPol*)nullptr, (WinCtl*)nullptr, 0, 0); });
        KT.launch(KC_OTHER, [&] { hipLaunchKernelGGL(k_lba_reduce, dim3(W), dim3(256), 0, st, dD, dC, dO, 1); });
      } else {
        KT.launch(KC_UPDATE, [&] { hipLaunchKernelGGL(k_lba_update_points, dim3(gq, W), dim3(256), 0, st, dD, dC, dO); });
        KT.launch(KC_ERROR, [&] { hipLaunchKernelGGL(k_lba_error, dim3(ge, W), dim3(256), 0, st, dD, dC, 1); });
        if (max_imu > 0) KT.launch(KC_GENERIC, [&] { hipLaunchKernelGGL(k_lba_generic, dim3(max_imu, W), dim3(64), 0, st, dD, dC, 1); });
        KT.launch(KC_OTHER, [&] { hipLaunchKernelGGL(k_lba_reduce, dim3(W), dim3(256), 0, st, dD, dC, dO, 0); });
      }
      if (sh) {  // chi2 and the landmark part of the gain-ratio scale
        if ((rc = shard_exchange(sh, sh->d_buf + shard_sys, 4 * (size_t)W, st)) != VIEO_OK) return rc;
        VIEO_HIP_CHECK(hipMemcpyAsync(h_sc, sh->d_buf + shard_sys, 32 * (size_t)W, hipMemcpyDeviceToHost, st));
      }
      VIEO_HIP_CHECK(hipMemcpyAsync(out, dO, (size_t)W * sizeof(WinOut), hipMemcpyDeviceToHost, st));
    }
    VIEO_HIP_CHECK(hipGetLastError());
    {
      const auto t_w = std::chrono::steady_clock::now();
      VIEO_HIP_CHECK(hipStreamSynchronize(st));
      ms_wait += ms_since(t_w);
    }
    if (KT.on) {
      double fl = 0;  // the windows this round's k_lba_schur launch worked on
      for (int w = 0; w < W; w++)
        if (ctl[w].flags & LBA_TRIAL) fl += schur_flops_of(w);
      KT.fold(fl);
    }
    if (sh)  // the ranks' stop requests of this round (the same count in every window that ran a trial)
      for (int w = 0; w < W; w++)
        if ((ctl[w].flags & (LBA_TRIAL | LBA_BEGIN)) && h_sc[4 * w + 3] > 0) shard_stop = true;
    // ---- per-window policy (optimization_algorithm_levenberg.cpp:61-164)
    for (int w = 0; w < W; w++) {
      WinHost& H = win[w];
      const int fl = ctl[w].flags;
      if (!(fl & LBA_TRIAL)) continue;
      if (sh) {  // totals = all ranks' visual edges + the inertial edges; the ranks' stop requests
        out[w].chi0 = h_sc[4 * w] + out[w].chig0, out[w].chi2 = h_sc[4 * w + 1] + out[w].chig,
        out[w].scale_l = h_sc[4 * w + 2];
      }
      if (fl & LBA_BEGIN) {
        if (out[w].np == 0) {  // no active free vertex: optimize() returns at once
          H.phase = 2;
          continue;
        }
        H.R->lm_iterations++;
        H.currentChi = out[w].chi0;
        if (H.stage == 0) H.R->chi2_initial = H.currentChi;
        H.iniChi = H.currentChi;
        H.lambda = vio && !gba ? H.VP->lambda_init : out[w].lambda;
        H.ni = 2, H.nBad = 0, H.qmax = 0, H.it = 0;
        H.phase = 1;
      }
      H.R->lm_trials++;
      H.need_build = false;
      const bool ok2 = out[w].ok != 0;
      H.lastTrialChi = out[w].chi2;
      const double tempChi = ok2 ? out[w].chi2 : DBL_MAX;
      double rho = H.currentChi - tempChi;
      const double scale = (ok2 ? out[w].scale_l + out[w].scale_p : 0.0) + 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow(2 * rho - 1, 3);
        alpha = std::min(alpha, 2. / 3.);
        H.lambda *= std::max(1. / 3., alpha);
        H.ni = 2;
        H.currentChi = tempChi;
      } else {
        H.lambda *= H.ni;
        H.ni *= 2;
        H.need_restore = true;
      }
      H.qmax++;
      H.R->chi2_final = H.currentChi;
      const bool stopped = sh ? shard_stop : (stop && *stop);
      if (rho < 0 && H.qmax < 10 && !stopped) continue;  // next lambda trial of the same iteration
      bool terminate = H.qmax == 10 || rho == 0;
      if (!terminate) {
        if ((H.iniChi - H.currentChi) * 1e3 < H.iniChi)
          H.nBad++;
        else
          H.nBad = 0;
        terminate = H.nBad >= 3;
      }
      H.it++;
      if (terminate || H.it >= H.iters || stopped)
        H.phase = 2;
      else {
        H.R->lm_iterations++;
        H.iniChi = H.currentChi;
        H.qmax = 0;
        H.need_build = true;  // buildSystem at the accepted state
      }
    }
  }
  // ---- results: one copy for all windows
  const double ms_rounds = ms_since(t_rounds);
  const auto t_res = std::chrono::steady_clock::now();
  VIEO_HIP_CHECK(hipMemcpyAsync(hs + res_begin, base + res_begin, res_end - res_begin, hipMemcpyDeviceToHost, st));
  VIEO_HIP_CHECK(hipStreamSynchronize(st));
  if (host_timing)
    fprintf(stderr, "lba_run: %d windows, staging %.3f ms, %d rounds %.3f ms (of which waiting for the stream %.3f), "
                    "results copy %.3f ms; staging: layout %.3f, windows filled %.3f, descriptors %.3f, enqueued %.3f\n", W, ms_staged,
            n_rounds, ms_rounds, ms_wait, ms_since(t_res), ms_layout, ms_filled, ms_described, ms_staged);
  for (int w = 0; w < W; w++) {
    WinHost& H = win[w];
    if (H.skip) continue;
    if (vio) {  // float err / err_end and the divergence guard (Optimizer.cc:531-533,655-666)
      const float err = (float)H.R->chi2_initial, err_end = (float)H.lastTrialChi;
      H.R->chi2_initial = err, H.R->chi2_final = err_end;
      if ((2 * err < err_end || std::isnan(err) || std::isnan(err_end)) && !H.VP->large && !gba) {
        H.R->status = VIEO_LBA_DIVERGED;
        continue;  // returns without write-back: outputs stay equal to the inputs
      }
    }
    if (H.n_obs > 0) memcpy(h_erase[w], hs + H.o_erase, H.n_obs);
    for (int i = 0; i < H.n_obs; i++) H.R->n_erase += h_erase[w][i];
    const LbaKf* o = (const LbaKf*)(hs + H.o_kf);
    const double* X = (const double*)(hs + H.o_X);
    for (int k = 0; k < H.n_kf; k++) {
      if (h_kfs[w][k].fixed) continue;
      memcpy(h_navs_out[w][k].p, o[k].p, 24);
      h_navs_out[w][k].q[0] = o[k].qw, h_navs_out[w][k].q[1] = o[k].qx;
      h_navs_out[w][k].q[2] = o[k].qy, h_navs_out[w][k].q[3] = o[k].qz;
      if (vio) {  // ns_recov: v of the V vertex, dbg / dba of the Bias vertex (Optimizer.cc:716-733)
        memcpy(h_navs_out[w][k].v, o[k].v, 24);
        memcpy(h_navs_out[w][k].dbg, o[k].dbg, 24), memcpy(h_navs_out[w][k].dba, o[k].dba, 24);
      }
    }
    if (sco) {  // SetWorldPos(scale * vPoint->estimate().cast<float>()) (Optimizer.cc:1321): a float product
      const double scale = *(const double*)(hs + H.o_scl);
      const float sf = (float)scale;
      for (int i = 0; i < H.n_mp * 3; i++) h_points_out[w][i] = sf * (float)X[i];
      if (gba->scale_out) *gba->scale_out = scale;
    } else
      for (int i = 0; i < H.n_mp * 3; i++) h_points_out[w][i] = (float)X[i];  // SetWorldPos(cast<float>)
  }
  return VIEO_OK;
}

}  // namespace vieo

using namespace vieo;

extern "C" {

int vieo_lba_set_stream_priority(int priority) {
  if (priority < -1 || priority > 1) return VIEO_E_INVALID;
  if (priority != g_lba_priority && g_lba_stream) {  // the next call creates the stream again
    (void)hipStreamSynchronize(g_lba_stream);
    (void)hipStreamDestroy(g_lba_stream);
    g_lba_stream = nullptr;
  }
  g_lba_priority = priority;
  return VIEO_OK;
}

int vieo_local_bundle_adjustment_batch(int n_windows, const vieo_lba_params* const* params,
                                       const vieo_lba_keyframe* const* h_kfs, const int* n_kf,
                                       const float* const* h_points, const int* n_mp,
                                       const vieo_lba_obs* const* h_obs, const int* n_obs,
                                       volatile const int* stop, vieo_navstate* const* h_navs_out,
                                       float* const* h_points_out, uint8_t* const* h_erase,
                                       vieo_lba_result* h_results) {
  return lba_run(nullptr, nullptr, n_windows, params, nullptr, h_kfs, n_kf, h_points, nullptr, n_mp, h_obs, n_obs, nullptr, nullptr,
                 stop, h_navs_out, h_points_out, h_erase, h_results);
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

int vieo_local_bundle_adjustment_vio_batch(int n_windows, const vieo_lba_vio_params* const* params,
                                           const vieo_lba_keyframe* const* h_kfs, const int* n_kf,
                                           const float* const* h_points, const uint8_t* const* h_close,
                                           const int* n_mp, const vieo_lba_obs* const* h_obs, const int* n_obs,
                                           const vieo_lba_imu_edge* const* h_imu, const int* n_imu,
                                           volatile const int* stop, vieo_navstate* const* h_navs_out,
                                           float* const* h_points_out, uint8_t* const* h_erase,
                                           vieo_lba_result* h_results) {
  if (!params) return VIEO_E_INVALID;
  return lba_run(nullptr, nullptr, n_windows, nullptr, params, h_kfs, n_kf, h_points, h_close, n_mp, h_obs, n_obs, h_imu, n_imu,
                 stop, h_navs_out, h_points_out, h_erase, h_results);
}

size_t vieo_lba_sharded_buffer_doubles(int n_windows, const int* n_free_kf) {
  size_t n = 0;
  for (int w = 0; w < n_windows; w++) n += shard_sys_doubles(n_free_kf ? n_free_kf[w] : 0, 1) + 4;  // (room for the scale vertex)
  return n;
}

int vieo_local_bundle_adjustment_vio_sharded(int n_windows, const vieo_lba_vio_params* const* params,
                                             const vieo_lba_keyframe* const* h_kfs, const int* n_kf,
                                             const float* const* h_points, const uint8_t* const* h_close,
                                             const int* n_mp, const vieo_lba_obs* const* h_obs, const int* n_obs,
                                             const vieo_lba_imu_edge* const* h_imu, const int* n_imu,
                                             double* d_reduce_buf, size_t reduce_cap_doubles,
                                             vieo_allreduce_sum_f64_fn allreduce, void* ctx,
                                             vieo_navstate* const* h_navs_out, float* const* h_points_out,
                                             uint8_t* const* h_erase, vieo_lba_result* h_results) {
  return vieo_local_bundle_adjustment_vio_sharded_stop(n_windows, params, h_kfs, n_kf, h_points, h_close, n_mp, h_obs, n_obs, h_imu,
                                                       n_imu, d_reduce_buf, reduce_cap_doubles, allreduce, ctx, nullptr, h_navs_out,
                                                       h_points_out, h_erase, h_results);
}

int vieo_local_bundle_adjustment_vio_sharded_stop(int n_windows, const vieo_lba_vio_params* const* params,
                                                  const vieo_lba_keyframe* const* h_kfs, const int* n_kf,
                                                  const float* const* h_points, const uint8_t* const* h_close,
                                                  const int* n_mp, const vieo_lba_obs* const* h_obs, const int* n_obs,
                                                  const vieo_lba_imu_edge* const* h_imu, const int* n_imu,
                                                  double* d_reduce_buf, size_t reduce_cap_doubles,
                                                  vieo_allreduce_sum_f64_fn allreduce, void* ctx, volatile const int* stop,
                                                  vieo_navstate* const* h_navs_out, float* const* h_points_out,
                                                  uint8_t* const* h_erase, vieo_lba_result* h_results) {
  if (!params) return VIEO_E_INVALID;
  LbaShard sh{allreduce, ctx, d_reduce_buf, reduce_cap_doubles};
  return lba_run(&sh, nullptr, n_windows, nullptr, params, h_kfs, n_kf, h_points, h_close, n_mp, h_obs, n_obs, h_imu,
                 n_imu, stop, h_navs_out, h_points_out, h_erase, h_results);
}

int vieo_global_bundle_adjustment_vio_sharded_scale(const vieo_lba_vio_params* params, int n_iterations, int robust,
                                                    int scale_opt, const vieo_lba_keyframe* h_kfs, int n_kf,
                                                    const float* h_points, int n_mp, const vieo_lba_obs* h_obs, int n_obs,
                                                    const vieo_lba_imu_edge* h_imu, int n_imu, double* d_reduce_buf,
                                                    size_t reduce_cap_doubles, vieo_allreduce_sum_f64_fn allreduce,
                                                    void* ctx, vieo_navstate* h_navs_out, float* h_points_out,
                                                    vieo_lba_result* h_result, double* h_scale_out) {
  if (!params || n_iterations < 0 || !h_result || n_obs < 0) return VIEO_E_INVALID;
  GbaMode g = {n_iterations, robust};
  g.scale_opt = scale_opt != 0, g.scale_out = h_scale_out;
  LbaShard sh{allreduce, ctx, d_reduce_buf, reduce_cap_doubles};
  std::vector<uint8_t> erase((size_t)std::max(n_obs, 1));
  uint8_t* er = erase.data();
  const uint8_t* no_close = nullptr;
  return lba_run(&sh, &g, 1, nullptr, &params, &h_kfs, &n_kf, &h_points, &no_close, &n_mp, &h_obs, &n_obs, &h_imu,
                 &n_imu, nullptr, &h_navs_out, &h_points_out, &er, h_result);
}

int vieo_global_bundle_adjustment_vio_sharded(const vieo_lba_vio_params* params, int n_iterations, int robust,
                                              const vieo_lba_keyframe* h_kfs, int n_kf, const float* h_points,
                                              int n_mp, const vieo_lba_obs* h_obs, int n_obs,
                                              const vieo_lba_imu_edge* h_imu, int n_imu, double* d_reduce_buf,
                                              size_t reduce_cap_doubles, vieo_allreduce_sum_f64_fn allreduce,
                                              void* ctx, vieo_navstate* h_navs_out, float* h_points_out,
                                              vieo_lba_result* h_result) {
  return vieo_global_bundle_adjustment_vio_sharded_scale(params, n_iterations, robust, 0, h_kfs, n_kf, h_points, n_mp,
                                                         h_obs, n_obs, h_imu, n_imu, d_reduce_buf, reduce_cap_doubles,
                                                         allreduce, ctx, h_navs_out, h_points_out, h_result, nullptr);
}

int vieo_local_bundle_adjustment_vio(const vieo_lba_vio_params* P, const vieo_lba_keyframe* h_kfs, int n_kf,
                                     const float* h_points, const uint8_t* h_close, int n_mp,
                                     const vieo_lba_obs* h_obs, int n_obs, const vieo_lba_imu_edge* h_imu,
                                     int n_imu, volatile const int* stop, vieo_navstate* h_navs_out,
                                     float* h_points_out, uint8_t* h_erase, vieo_lba_result* R) {
  if (!P || !h_kfs || n_kf <= 0 || !h_points || !h_close || n_mp <= 0 || !h_obs || n_obs <= 0 || n_imu < 0 ||
      (n_imu > 0 && !h_imu) || !h_navs_out || !h_points_out || !h_erase || !R)
    return VIEO_E_INVALID;
  return vieo_local_bundle_adjustment_vio_batch(1, &P, &h_kfs, &n_kf, &h_points, &h_close, &n_mp, &h_obs, &n_obs,
                                                &h_imu, &n_imu, stop, &h_navs_out, &h_points_out, &h_erase, R);
}

int vieo_bundle_adjustment(const vieo_lba_params* params, int n_iterations, int robust,
                           const vieo_lba_keyframe* h_kfs, int n_kf, const float* h_points, int n_mp,
                           const vieo_lba_obs* h_obs, int n_obs, volatile const int* stop, vieo_navstate* h_navs_out,
                           float* h_points_out, vieo_lba_result* h_result) {
  if (!params || n_iterations < 0 || !h_result || n_obs < 0) return VIEO_E_INVALID;
  const GbaMode g = {n_iterations, robust};
  std::vector<uint8_t> erase((size_t)std::max(n_obs, 1));
  uint8_t* er = erase.data();
  return lba_run(nullptr, &g, 1, &params, nullptr, &h_kfs, &n_kf, &h_points, nullptr, &n_mp, &h_obs, &n_obs, nullptr,
                 nullptr, stop, &h_navs_out, &h_points_out, &er, h_result);
}

int vieo_local_bundle_adjustment_batch_enc(int n_windows, const vieo_lba_params* const* params,
                                           const vieo_lba_keyframe* const* h_kfs, const int* n_kf,
                                           const float* const* h_points, const int* n_mp,
                                           const vieo_lba_obs* const* h_obs, const int* n_obs,
                                           const vieo_lba_enc* const* encs, volatile const int* stop,
                                           vieo_navstate* const* h_navs_out, float* const* h_points_out,
                                           uint8_t* const* h_erase, vieo_lba_result* h_results) {
  return lba_run(nullptr, nullptr, n_windows, params, nullptr, h_kfs, n_kf, h_points, nullptr, n_mp, h_obs, n_obs, nullptr, nullptr,
                 stop, h_navs_out, h_points_out, h_erase, h_results, encs);
}

int vieo_local_bundle_adjustment_enc(const vieo_lba_params* P, const vieo_lba_keyframe* h_kfs, int n_kf,
                                     const float* h_points, int n_mp, const vieo_lba_obs* h_obs, int n_obs,
                                     const vieo_lba_enc* enc, volatile const int* stop, vieo_navstate* h_navs_out,
                                     float* h_points_out, uint8_t* h_erase, vieo_lba_result* R) {
  if (!P || !h_kfs || n_kf <= 0 || !h_points || n_mp <= 0 || !h_obs || n_obs <= 0 || !h_navs_out ||
      !h_points_out || !h_erase || !R)
    return VIEO_E_INVALID;
  return lba_run(nullptr, nullptr, 1, &P, nullptr, &h_kfs, &n_kf, &h_points, nullptr, &n_mp, &h_obs, &n_obs, nullptr,
                 nullptr, stop, &h_navs_out, &h_points_out, &h_erase, R, &enc);
}

int vieo_bundle_adjustment_enc(const vieo_lba_params* params, int n_iterations, int robust,
                               const vieo_lba_keyframe* h_kfs, int n_kf, const float* h_points, int n_mp,
                               const vieo_lba_obs* h_obs, int n_obs, const vieo_lba_enc* enc,
                               volatile const int* stop, vieo_navstate* h_navs_out, float* h_points_out,
                               vieo_lba_result* h_result) {
  if (!params || n_iterations < 0 || !h_result || n_obs < 0) return VIEO_E_INVALID;
  const GbaMode g = {n_iterations, robust};
  std::vector<uint8_t> erase((size_t)std::max(n_obs, 1));
  uint8_t* er = erase.data();
  return lba_run(nullptr, &g, 1, &params, nullptr, &h_kfs, &n_kf, &h_points, nullptr, &n_mp, &h_obs, &n_obs, nullptr,
                 nullptr, stop, &h_navs_out, &h_points_out, &er, h_result, &enc);
}

int vieo_global_bundle_adjustment_vio(const vieo_lba_vio_params* params, int n_iterations, int robust,
                                      const vieo_lba_keyframe* h_kfs, int n_kf, const float* h_points, int n_mp,
                                      const vieo_lba_obs* h_obs, int n_obs, const vieo_lba_imu_edge* h_imu, int n_imu,
                                      volatile const int* stop, vieo_navstate* h_navs_out, float* h_points_out,
                                      vieo_lba_result* h_result) {
  return vieo_global_bundle_adjustment_vio_scale(params, n_iterations, robust, 0, h_kfs, n_kf, h_points, n_mp, h_obs, n_obs,
                                                 h_imu, n_imu, stop, h_navs_out, h_points_out, h_result, nullptr);
}

int vieo_global_bundle_adjustment_vio_scale(const vieo_lba_vio_params* params, int n_iterations, int robust, int scale_opt,
                                            const vieo_lba_keyframe* h_kfs, int n_kf, const float* h_points, int n_mp,
                                            const vieo_lba_obs* h_obs, int n_obs, const vieo_lba_imu_edge* h_imu,
                                            int n_imu, volatile const int* stop, vieo_navstate* h_navs_out,
                                            float* h_points_out, vieo_lba_result* h_result, double* h_scale_out) {
  if (!params || n_iterations < 0 || !h_result || n_obs < 0) return VIEO_E_INVALID;
  GbaMode g = {n_iterations, robust};
  g.scale_opt = scale_opt != 0, g.scale_out = h_scale_out;
  std::vector<uint8_t> erase((size_t)std::max(n_obs, 1));
  uint8_t* er = erase.data();
  const uint8_t* no_close = nullptr;
  return lba_run(nullptr, &g, 1, nullptr, &params, &h_kfs, &n_kf, &h_points, &no_close, &n_mp, &h_obs, &n_obs, &h_imu,
                 &n_imu, stop, &h_navs_out, &h_points_out, &er, h_result);
}


// Kernel-class timing of the bundle-adjustment engine (bench.py's roofline over the whole path).
void vieo_lba_enable_timing(int on) {
  std::lock_guard<std::mutex> g(vieo::g_lba_kt_mutex);
  vieo::g_lba_ktiming.store(on == 2 ? 2 : (on ? 1 : 0));
  for (int i = 0; i < vieo::KC_N; i++) vieo::g_lba_kt_ms[i] = 0, vieo::g_lba_kt_launches[i] = 0;
  vieo::g_lba_kt_schur_flops = 0;
}
int vieo_lba_kernel_classes(void) { return vieo::KC_N; }
const char* vieo_lba_kernel_class_name(int i) { return i >= 0 && i < vieo::KC_N ? vieo::kLbaKClassName[i] : ""; }
void vieo_lba_kernel_times(double* ms, long long* launches, double* schur_flops) {
  std::lock_guard<std::mutex> g(vieo::g_lba_kt_mutex);
  for (int i = 0; i < vieo::KC_N; i++) ms[i] = vieo::g_lba_kt_ms[i], launches[i] = vieo::g_lba_kt_launches[i];
  *schur_flops = vieo::g_lba_kt_schur_flops;
}
}  // extern "C"
