// fisheye_stereo.hip -- Frame::ComputeStereoFishEyeMatches (reference src/Frame.cc:613-779), the stereo stage
// of the distorted multi-camera configurations (a10).
//
//   device   k_knn2 (matching.hip) for every camera pair; k_fe_pairs: Lowe ratio + the pair triangulation of
//            GeometricCamera::FillMatchesFromPair (common/camera_models/camera_base.h:408-574 ->
//            TriangulateMatches :199-285 -> Triangulate :576-608) for every query row, one lane per row -- the
//            triangulation is a pure function of the pair, so it is evaluated for all rows at once and the
//            order-dependent part only reads its verdicts; k_fe_groups: the all-camera re-triangulation of every
//            group when n_cams > 2 (Frame.cc:704-737).
//   host     the group bookkeeping of FillMatchesFromPair under USE_STRATEGY_MIN_DIST (common/config.h:12):
//            inherently sequential (each match reads what the previous ones wrote), a few thousand steps over
//            flat tables -- as in the reference it runs on the calling thread.
// Eigen::JacobiSVD's last right singular vector (camera_base.h:599-600) is obtained by one-sided Jacobi
// rotations on the columns of A (FP64), which is branch-light and register resident for a 4-column matrix.
#include <algorithm>
#include <cmath>
#include <vector>

#include "ba_device.h"
#include "cam_unproject.h"
#include "match_groups.h"

namespace vieo {

struct FeRig {
  int n_cams;
  CamD cam[4];
  double Rrc[4][9];   // rotation of Trc
  double Tcw[4][12];  // Trc^-1 (3x4), inverted in double like Twi[i].inverse()
  float th[2];        // the two parallax thresholds as FillMatchesFromPair receives them (float)
};

// right singular vector of the smallest singular value of A (M x 4), one-sided Jacobi; A is destroyed
template <int M>
__device__ __forceinline__ void null_vector4(double (&A)[M][4], double* x4) {
  double V[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) V[r][c] = r == c ? 1. : 0.;
  for (int sweep = 0; sweep < 40; ++sweep) {
    bool rotated = false;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        double a = 0, b = 0, g = 0;
#pragma unroll
        for (int r = 0; r < M; ++r) a += A[r][p] * A[r][p], b += A[r][q] * A[r][q], g += A[r][p] * A[r][q];
        if (g == 0 || fabs(g) <= 1e-15 * sqrt(a * b)) continue;
        rotated = true;
        const double zeta = (b - a) / (2 * g);
        const double t = (zeta >= 0 ? 1. : -1.) / (fabs(zeta) + sqrt(1 + zeta * zeta));
        const double cs = 1 / sqrt(1 + t * t), sn = cs * t;
#pragma unroll
        for (int r = 0; r < M; ++r) {
          const double u = A[r][p], v = A[r][q];
          A[r][p] = cs * u - sn * v, A[r][q] = sn * u + cs * v;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double u = V[r][p], v = V[r][q];
          V[r][p] = cs * u - sn * v, V[r][q] = sn * u + cs * v;
        }
      }
    if (!rotated) break;
  }
  double nb = INFINITY;
  x4[0] = x4[1] = x4[2] = x4[3] = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double n = 0;
#pragma unroll
    for (int r = 0; r < M; ++r) n += A[r][c] * A[r][c];
    if (n < nb) {
      nb = n;
#pragma unroll
      for (int r = 0; r < 4; ++r) x4[r] = V[r][c];
    }
  }
}

// GeometricCamera::TriangulateMatches over N cameras ci[] with key points kp[] (camera_base.h:199-285).
// gate[k]: whether the parallax test passes for threshold th[k]; the rest does not depend on the threshold.
// returns false for the reference's empty vector (apart from the parallax gate); czs = depths.
template <int N>
__device__ bool triangulate_matches(const FeRig& R, const int* ci, const float (*kp)[2], const float* sig,
                                    bool* gate, double* p3d, float* czs) {
  double nP[N][3];
#pragma unroll
  for (int i = 0; i < N; ++i) cam_unproject(R.cam[ci[i]], kp[i][0], kp[i][1], nP[i]);
  // "bret" of the reference: every pair has cos > th  =>  no usable parallax
  bool all_above[2] = {true, true};
#pragma unroll
  for (int i = 0; i < N - 1; ++i)
#pragma unroll
    for (int j = i + 1; j < N; ++j) {
      const double* Ri = R.Rrc[ci[i]];
      const double* Rj = R.Rrc[ci[j]];
      double w[3], v[3];
      for (int r = 0; r < 3; ++r) w[r] = Rj[r * 3] * nP[j][0] + Rj[r * 3 + 1] * nP[j][1] + Rj[r * 3 + 2] * nP[j][2];
      for (int r = 0; r < 3; ++r) v[r] = Ri[r] * w[0] + Ri[3 + r] * w[1] + Ri[6 + r] * w[2];
      const double dot = nP[i][0] * v[0] + nP[i][1] * v[1] + nP[i][2] * v[2];
      const double ni = sqrt(nP[i][0] * nP[i][0] + nP[i][1] * nP[i][1] + nP[i][2] * nP[i][2]);
      const double nj = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      const float cosr = (float)(dot / (ni * nj));
      if (cosr <= R.th[0]) all_above[0] = false;
      if (cosr <= R.th[1]) all_above[1] = false;
    }
  gate[0] = !(R.th[0] < 1.f && all_above[0]);
  gate[1] = !(R.th[1] < 1.f && all_above[1]);
  if (!gate[0] && !gate[1]) return false;
  double A[2 * N][4];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double* T = R.Tcw[ci[i]];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      A[2 * i][c] = nP[i][0] * T[8 + c] - T[c];
      A[2 * i + 1][c] = nP[i][1] * T[8 + c] - T[4 + c];
    }
  }
  double x4[4];
  null_vector4<2 * N>(A, x4);
  if (!x4[3]) return false;
  const double X[3] = {x4[0] / x4[3], x4[1] / x4[3], x4[2] / x4[3]};
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double* T = R.Tcw[ci[i]];
    czs[i] = (float)(T[8] * X[0] + T[9] * X[1] + T[10] * X[2] + T[11]);
    if (czs[i] <= 0) return false;
    double Pc[3], uv[2];
    for (int r = 0; r < 3; ++r) Pc[r] = (T[r * 4] * X[0] + T[r * 4 + 1] * X[1] + T[r * 4 + 2] * X[2]) + T[r * 4 + 3];
    cam_project(R.cam[ci[i]], Pc, uv, nullptr);
    const float e0 = (float)uv[0] - kp[i][0], e1 = (float)uv[1] - kp[i][1];
    if (e0 * e0 + e1 * e1 > 5.991f * sig[i]) return false;
  }
  p3d[0] = X[0], p3d[1] = X[1], p3d[2] = X[2];
  return true;
}

struct FePairRec {  // verdict of one knn row
  int32_t idxj;     // absolute key index in camera j, -1: ratio test failed / fewer than two neighbours
  float dist;
  int32_t ok;       // bit k: FillMatchesFromPair's triangulation accepts the pair under threshold k
  int32_t pad;
  double p3d[3];
};

struct FeArgs {
  const vieo_keypoint* keys;  // [cam][cap]
  const int32_t* knn_idx;     // [pair][cap][2]
  const int32_t* knn_dist;
  const float* level_sigma2;
  FePairRec* rec;             // [pair][cap]
  int cap, n_pairs;
  int pair_i[6], pair_j[6], nq[6], mono[4];
};

__global__ void __launch_bounds__(64)
k_fe_pairs(const FeRig* __restrict__ rig, FeArgs A) {
  const int p = blockIdx.y, q = blockIdx.x * 64 + threadIdx.x;
  if (q >= A.nq[p]) return;
  FePairRec r;
  r.idxj = -1, r.dist = 0, r.ok = 0, r.pad = 0, r.p3d[0] = r.p3d[1] = r.p3d[2] = 0;
  const int32_t* id = A.knn_idx + ((size_t)p * A.cap + q) * 2;
  const int32_t* dd = A.knn_dist + ((size_t)p * A.cap + q) * 2;
  if (id[0] >= 0 && id[1] >= 0) {
    const float d0 = (float)dd[0], d1 = (float)dd[1];
    // Lowe ratio, Frame.cc:661-663 (float distance against double products)
    if ((double)d0 < (double)d1 * 0.7 || (d0 < 75.f && (double)d0 < (double)d1 * 0.9)) {
      const int ci[2] = {A.pair_i[p], A.pair_j[p]};
      const int ia = q + A.mono[ci[0]], ib = id[0] + A.mono[ci[1]];
      const vieo_keypoint ka = A.keys[(size_t)ci[0] * A.cap + ia], kb = A.keys[(size_t)ci[1] * A.cap + ib];
      const float kp[2][2] = {{ka.x, ka.y}, {kb.x, kb.y}};
      const float sig[2] = {A.level_sigma2[ka.octave], A.level_sigma2[kb.octave]};
      bool gate[2];
      float czs[2];
      r.idxj = ib, r.dist = d0;
      if (triangulate_matches<2>(*rig, ci, kp, sig, gate, r.p3d, czs) && czs[0] > 0.0001f && czs[1] > 0.0001f)
        r.ok = (gate[0] ? 1 : 0) | (gate[1] ? 2 : 0);
    }
  }
  A.rec[(size_t)p * A.cap + q] = r;
}

struct FeGroupOut {
  double p3d[3];
  int32_t ok, pad;
};

// Frame.cc:704-737: every good group is triangulated again from all of its cameras (threshold `which`)
__global__ void __launch_bounds__(64)
k_fe_groups(const FeRig* __restrict__ rig, const vieo_keypoint* __restrict__ keys, int cap,
            const float* __restrict__ level_sigma2, const int32_t* __restrict__ gidx, const uint8_t* __restrict__ good,
            int n_groups, int which, FeGroupOut* __restrict__ out) {
  const int g = blockIdx.x * 64 + threadIdx.x;
  if (g >= n_groups) return;
  FeGroupOut o;
  o.ok = 0, o.pad = 0, o.p3d[0] = o.p3d[1] = o.p3d[2] = 0;
  if (good[g]) {
    const int nc = rig->n_cams;
    int ci[4] = {0, 0, 0, 0}, n = 0;
    float kp[4][2], sig[4], czs[4] = {1, 1, 1, 1};
    for (int k = 0; k < nc; ++k) {
      const int ix = gidx[(size_t)g * nc + k];
      if (ix < 0) continue;
      const vieo_keypoint kk = keys[(size_t)k * cap + ix];
      // select chain instead of ci[n]: keeps the small arrays in registers
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (s == n) ci[s] = k, kp[s][0] = kk.x, kp[s][1] = kk.y, sig[s] = level_sigma2[kk.octave];
      ++n;
    }
    bool gate[2] = {false, false}, ok = false;
    if (n == 2)
      ok = triangulate_matches<2>(*rig, ci, kp, sig, gate, o.p3d, czs);
    else if (n == 3)
      ok = triangulate_matches<3>(*rig, ci, kp, sig, gate, o.p3d, czs);
    else if (n == 4)
      ok = triangulate_matches<4>(*rig, ci, kp, sig, gate, o.p3d, czs);
    ok = ok && gate[which];
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (s < n && czs[s] <= 0.0001f) ok = false;
    o.ok = ok;
  }
  out[g] = o;
}

struct FeScratch {
  DevBuf keys, desc, idx, dist, rec, rig, sig, gidx, good, gout;
};
static thread_local FeScratch g_fe;

}  // namespace vieo

using namespace vieo;

extern "C" {

int vieo_stereo_fisheye_match(const vieo_fisheye_params* P, const vieo_keypoint* const* h_keys,
                              const uint8_t* const* h_desc, const int32_t* n_keys, const int32_t* num_mono,
                              int32_t group_capacity, float* h_depth, int32_t* h_key_group, int32_t* h_group_idx,
                              uint8_t* h_group_good, double* h_group_p3d, int32_t* n_groups, int32_t* n_matches) {
  if (!P || !h_keys || !h_desc || !n_keys || !num_mono || !h_depth || !h_key_group || !h_group_idx || !h_group_good ||
      !h_group_p3d || !n_groups || !n_matches || group_capacity < 0)
    return VIEO_E_INVALID;
  const int nc = P->n_cams;
  if (nc < 2 || nc > 4 || !P->cams || !P->Trc || !P->Tcr || !P->level_sigma2 || P->n_levels <= 0) {
    set_error("ComputeStereoFishEyeMatches: n_cams = %d (2..4) with cameras, Trc, Tcr and level sigmas", nc);
    return VIEO_E_INVALID;
  }
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  int cap = 1;
  for (int c = 0; c < nc; ++c) {
    if (n_keys[c] < 0 || num_mono[c] < 0 || (n_keys[c] > 0 && (!h_keys[c] || !h_desc[c]))) return VIEO_E_INVALID;
    cap = std::max(cap, n_keys[c]);
    for (int k = 0; k < n_keys[c]; ++k)
      if (h_keys[c][k].octave < 0 || h_keys[c][k].octave >= P->n_levels) {
        set_error("ComputeStereoFishEyeMatches: key %d of camera %d has octave %d", k, c, h_keys[c][k].octave);
        return VIEO_E_INVALID;
      }
  }
  FeRig R;
  memset(&R, 0, sizeof(R));
  R.n_cams = nc;
  for (int c = 0; c < nc; ++c) {
    if (!cam_from_abi(P->cams[c], R.cam[c])) {
      set_error("ComputeStereoFishEyeMatches: camera %d has an unknown model or coefficient count", c);
      return VIEO_E_INVALID;
    }
    const double* T = P->Trc + 12 * c;
    for (int r = 0; r < 3; ++r) {
      for (int q = 0; q < 3; ++q) R.Rrc[c][r * 3 + q] = T[r * 4 + q], R.Tcw[c][r * 4 + q] = T[q * 4 + r];
      R.Tcw[c][r * 4 + 3] = -(T[0 * 4 + r] * T[3] + T[1 * 4 + r] * T[7] + T[2 * 4 + r] * T[11]);
    }
  }
  // Frame.cc:636-644
  const float f_bar = (P->cams[0].fx + P->cams[0].fy) / 2.;
  double th[2] = {0.9998, 1. - 1e-6};
  if (P->th_far_pts > 0)
    for (int i = 0; i < 2; ++i) th[i] = std::min(1. - std::pow(P->bf / f_bar / P->th_far_pts, 2) / 2., th[i]);
  R.th[0] = (float)th[0], R.th[1] = (float)th[1];

  FeScratch& S = g_fe;
  const int n_pairs = nc * (nc - 1) / 2;
#define ENS(b, n) \
  if ((rc = (b).ensure(n)) != VIEO_OK) return rc
  ENS(S.keys, (size_t)nc * cap * sizeof(vieo_keypoint));
  ENS(S.desc, (size_t)nc * cap * 32);
  ENS(S.idx, (size_t)n_pairs * cap * 8);
  ENS(S.dist, (size_t)n_pairs * cap * 8);
  ENS(S.rec, (size_t)n_pairs * cap * sizeof(FePairRec));
  ENS(S.rig, sizeof(FeRig));
  ENS(S.sig, (size_t)P->n_levels * 4);
  for (int c = 0; c < nc; ++c)
    if (n_keys[c] > 0) {
      VIEO_HIP_CHECK(hipMemcpy(S.keys.as<vieo_keypoint>() + (size_t)c * cap, h_keys[c],
                               (size_t)n_keys[c] * sizeof(vieo_keypoint), hipMemcpyHostToDevice));
      VIEO_HIP_CHECK(hipMemcpy(S.desc.as<uint8_t>() + (size_t)c * cap * 32, h_desc[c], (size_t)n_keys[c] * 32,
                               hipMemcpyHostToDevice));
    }
  VIEO_HIP_CHECK(hipMemcpy(S.rig.p, &R, sizeof(R), hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(hipMemcpy(S.sig.p, P->level_sigma2, (size_t)P->n_levels * 4, hipMemcpyHostToDevice));
  // brute force between the key points of all image pairs (Frame.cc:618-628)
  FeArgs A;
  memset(&A, 0, sizeof(A));
  int32_t counts[8], pairs[12];
  int max_nq = 0;
  for (int c = 0; c < nc; ++c) counts[2 * c] = n_keys[c], counts[2 * c + 1] = num_mono[c], A.mono[c] = num_mono[c];
  for (int i = 0, p = 0; i < nc - 1; ++i)
    for (int j = i + 1; j < nc; ++j, ++p) {
      pairs[2 * p] = i, pairs[2 * p + 1] = j;
      A.pair_i[p] = i, A.pair_j[p] = j;
      A.nq[p] = (num_mono[i] >= n_keys[i] || num_mono[j] >= n_keys[j]) ? 0 : n_keys[i] - num_mono[i];
      max_nq = std::max(max_nq, A.nq[p]);
    }
  std::vector<FePairRec> rec;
  if (max_nq > 0) {
    // rows of pairs without a search stay unwritten: nq = 0 keeps them out of every later step
    rc = vieo_hamming_knn2_batch_device(S.desc.as<uint8_t>(), counts, cap, pairs, n_pairs, S.idx.as<int32_t>(),
                                        S.dist.as<int32_t>(), nullptr);
    if (rc != VIEO_OK) return rc;
    A.keys = S.keys.as<vieo_keypoint>(), A.knn_idx = S.idx.as<int32_t>(), A.knn_dist = S.dist.as<int32_t>();
    A.level_sigma2 = S.sig.as<float>(), A.rec = S.rec.as<FePairRec>(), A.cap = cap, A.n_pairs = n_pairs;
    hipLaunchKernelGGL(k_fe_pairs, dim3((max_nq + 63) / 64, n_pairs), dim3(64), 0, 0, S.rig.as<FeRig>(), A);
    VIEO_HIP_CHECK(hipGetLastError());
    rec.resize((size_t)n_pairs * cap);
    VIEO_HIP_CHECK(hipMemcpy(rec.data(), S.rec.p, rec.size() * sizeof(FePairRec), hipMemcpyDeviceToHost));
  }
  // Frame.cc:650-694: one or two passes of the sequential bookkeeping
  FeGroups G;
  int nMatches = 0, which = 0;
  const int tries = th[1] == th[0] ? 1 : 2;
  for (int k = 0; k < tries; ++k) {
    which = k;
    G.reset(nc, n_keys);
    for (int p = 0; p < n_pairs; ++p)
      for (int q = 0; q < A.nq[p]; ++q) {
        const FePairRec& r = rec[(size_t)p * cap + q];
        if (r.idxj < 0) continue;
        if (fe_fill(G, A.pair_i[p], q + num_mono[A.pair_i[p]], A.pair_j[p], r.idxj, r.dist, (r.ok >> k) & 1, r.p3d))
          ++nMatches;
      }
    if (nMatches >= 30) break;
  }
  const int ng = G.size();
  if (ng > group_capacity) {
    set_error("ComputeStereoFishEyeMatches: %d groups, capacity %d", ng, group_capacity);
    return VIEO_E_CAPACITY;
  }
  for (int g = 0; g < ng; ++g) {  // Frame.cc:695-701
    int cnt = 0;
    for (int t = 0; t < nc; ++t) cnt += G.idxs[(size_t)g * nc + t] >= 0;
    if (cnt < 2) G.good[g] = 0;
  }
  if (nc > 2 && ng > 0) {  // Frame.cc:704-737
    ENS(S.gidx, (size_t)ng * nc * 4);
    ENS(S.good, (size_t)ng);
    ENS(S.gout, (size_t)ng * sizeof(FeGroupOut));
    VIEO_HIP_CHECK(hipMemcpy(S.gidx.p, G.idxs.data(), (size_t)ng * nc * 4, hipMemcpyHostToDevice));
    VIEO_HIP_CHECK(hipMemcpy(S.good.p, G.good.data(), (size_t)ng, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fe_groups, dim3((ng + 63) / 64), dim3(64), 0, 0, S.rig.as<FeRig>(),
                       S.keys.as<vieo_keypoint>(), cap, S.sig.as<float>(), S.gidx.as<int32_t>(), S.good.as<uint8_t>(),
                       ng, which, S.gout.as<FeGroupOut>());
    VIEO_HIP_CHECK(hipGetLastError());
    std::vector<FeGroupOut> out(ng);
    VIEO_HIP_CHECK(hipMemcpy(out.data(), S.gout.p, (size_t)ng * sizeof(FeGroupOut), hipMemcpyDeviceToHost));
    nMatches = 0;
    for (int g = 0; g < ng; ++g) {
      if (!G.good[g]) continue;
      if (out[g].ok) {
        memcpy(&G.p3d[(size_t)g * 3], out[g].p3d, 24);
        ++nMatches;
      } else
        G.good[g] = 0;
    }
  }
#undef ENS
  *n_groups = ng, *n_matches = nMatches;
  if (ng > 0) {
    memcpy(h_group_idx, G.idxs.data(), (size_t)ng * nc * 4);
    memcpy(h_group_good, G.good.data(), (size_t)ng);
    memcpy(h_group_p3d, G.p3d.data(), (size_t)ng * 24);
  }
  // vdepth_ of the concatenated key list (Frame.cc:742-764)
  size_t n = 0;
  for (int c = 0; c < nc; ++c) {
    const double* T = P->Tcr + 12 * c;
    for (int k = 0; k < n_keys[c]; ++k, ++n) {
      const int g = G.key2g[c][k];
      h_key_group[n] = g, h_depth[n] = -1;
      if (g >= 0 && G.good[g]) {
        const double* X = &G.p3d[(size_t)g * 3];
        h_depth[n] = (float)(T[8] * X[0] + T[9] * X[1] + T[10] * X[2] + T[11]);
      }
    }
  }
  return VIEO_OK;
}

}  // extern "C"
