// tri_search.hip -- ORBmatcher::SearchForTriangulation (reference src/ORBmatcher.cc:896-1150) for key
// frames with one undistorted pinhole camera, a batch of neighbours pKF2 of one pKF1 per call.
//   k_tri_gates   one lane per (node-shared key of pKF1) "query": walks the node's keys of pKF2 and keeps the ones
//                 that pass every gate that does not depend on earlier matches -- Hamming <= TH_LOW, the epipole gate
//                 for two monocular keys, GeometricCamera::epipolarConstrain (camera_base.h:287-406, fundamental
//                 matrix branch; Tdata float / Tcalc double as in common/config.h:23-24)
//   host          the reference's loop order over those candidates: keys of pKF2 already taken are skipped, best
//                 distance (ties to the later key), FillMatchesFromPair (match_groups.h), rotation histogram
// The candidate segment of a query is as long as its node's list in pKF2, so the kernel writes without atomics.
// HBM traffic is the descriptors of both node lists once (32 B per key); the kernel is a gather, latency-bound at
// these sizes (a few thousand queries per neighbour), which is why the neighbours of a key frame go in one launch.
#include <algorithm>
#include <cmath>
#include <vector>

#include "common.h"
#include "match_groups.h"

namespace vieo {

static const int kTriThLow = 50, kTriHisto = 30;

struct TriPairDev {  // per neighbour
  double F12[9];
  float ex, ey;
  int key_off, feat_off, lvl_off, pad;  // offsets of pKF2's arrays in the concatenated buffers
};

struct TriQuery {
  int idx1, pair, first2, count2, out_off;
};

__device__ __forceinline__ int tri_hamming(const uint4* a, const uint4* b) {
  const uint4 a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
         __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__global__ void __launch_bounds__(64)
k_tri_gates(const TriQuery* __restrict__ queries, int n_queries, const TriPairDev* __restrict__ pairs,
            const vieo_keypoint* __restrict__ keys1, const uint8_t* __restrict__ desc1, const float* __restrict__ ur1,
            const vieo_keypoint* __restrict__ keys2, const uint8_t* __restrict__ desc2, const float* __restrict__ ur2,
            const uint8_t* __restrict__ mp2, const int* __restrict__ feat2, const float* __restrict__ scale2,
            const float* __restrict__ sigma2, int only_stereo, int2* __restrict__ cand, int* __restrict__ cand_n) {
  const int q = blockIdx.x * 64 + threadIdx.x;
  if (q >= n_queries) return;
  const TriQuery Q = queries[q];
  const TriPairDev& P = pairs[Q.pair];
  const vieo_keypoint kp1 = keys1[Q.idx1];
  const bool st1 = ur1[Q.idx1] >= 0;
  const uint4* d1 = (const uint4*)(desc1 + 32 * (size_t)Q.idx1);
  const double p1x = kp1.x, p1y = kp1.y;
  // the epipolar line of key 1 in image 2 (Tdata = float)
  const float a = (float)(p1x * P.F12[0] + p1y * P.F12[3] + P.F12[6]);
  const float b = (float)(p1x * P.F12[1] + p1y * P.F12[4] + P.F12[7]);
  const float c = (float)(p1x * P.F12[2] + p1y * P.F12[5] + P.F12[8]);
  const float den = a * a + b * b;
  int n = 0;
  int2* out = cand + Q.out_off;
  for (int k = 0; k < Q.count2; k++) {
    const int idx2 = feat2[P.feat_off + Q.first2 + k];
    const int g2 = P.key_off + idx2;
    if (mp2[g2]) continue;
    const bool st2 = ur2[g2] >= 0;
    if (only_stereo && !st2) continue;
    const int dist = tri_hamming(d1, (const uint4*)(desc2 + 32 * (size_t)g2));
    if (dist > kTriThLow) continue;
    const vieo_keypoint kp2 = keys2[g2];
    if (!st1 && !st2) {
      const float distex = P.ex - kp2.x, distey = P.ey - kp2.y;
      if (distex * distex + distey * distey < 100 * scale2[P.lvl_off + kp2.octave]) continue;
    }
    const float num = (float)((double)a * (double)kp2.x + (double)b * (double)kp2.y + (double)c);
    if (den == 0) continue;
    const float dsqr = num * num / den;
    if (!(dsqr < 3.84f * sigma2[P.lvl_off + kp2.octave])) continue;
    out[n++] = make_int2(idx2, dist);
  }
  cand_n[q] = n;
}

// Tr1r2 = (Tcw1 * Twc2).cast<float>(), F12 = K1^-T [t12]x R12 K2^-1, the epipole of camera 1 in image 2
static void tri_pair_setup(const vieo_tri_keyframe& A, const vieo_tri_keyframe& B, TriPairDev& P) {
  const double *T1 = A.Tcw, *T2 = B.Tcw;
  double R[9], R12[9], t12[3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += T1[i * 4 + k] * T2[j * 4 + k];
      R[i * 3 + j] = s, R12[i * 3 + j] = (double)(float)s;
    }
  for (int i = 0; i < 3; i++) {
    double s = 0;
    for (int j = 0; j < 3; j++) s += R[i * 3 + j] * T2[j * 4 + 3];
    t12[i] = (double)(float)(T1[i * 4 + 3] - s);
  }
  const double fx1 = A.fx, fy1 = A.fy, cx1 = A.cx, cy1 = A.cy, fx2 = B.fx, fy2 = B.fy, cx2 = B.cx, cy2 = B.cy;
  const double K1it[9] = {1 / fx1, 0, 0, 0, 1 / fy1, 0, -cx1 / fx1, -cy1 / fy1, 1};
  const double K2i[9] = {1 / fx2, 0, -cx2 / fx2, 0, 1 / fy2, -cy2 / fy2, 0, 0, 1};
  const double H[9] = {0, -t12[2], t12[1], t12[2], 0, -t12[0], -t12[1], t12[0], 0};
  auto mul = [](const double* X, const double* Y, double* Z) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Z[i * 3 + j] = X[i * 3] * Y[j] + X[i * 3 + 1] * Y[3 + j] + X[i * 3 + 2] * Y[6 + j];
  };
  double M1[9], M2[9];
  mul(K1it, H, M1), mul(M1, R12, M2), mul(M2, K2i, P.F12);
  float R1f[9], t1f[3], R2f[9], t2f[3], Cw[3], C2[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R1f[i * 3 + j] = (float)T1[i * 4 + j], R2f[i * 3 + j] = (float)T2[i * 4 + j];
    t1f[i] = (float)T1[i * 4 + 3], t2f[i] = (float)T2[i * 4 + 3];
  }
  for (int i = 0; i < 3; i++)
    Cw[i] = (float)(-((double)R1f[i] * t1f[0] + (double)R1f[3 + i] * t1f[1] + (double)R1f[6 + i] * t1f[2]));
  for (int i = 0; i < 3; i++)
    C2[i] = (float)((double)R2f[i * 3] * Cw[0] + (double)R2f[i * 3 + 1] * Cw[1] + (double)R2f[i * 3 + 2] * Cw[2] +
                    (double)t2f[i]);
  const float invz = 1.0f / C2[2];
  const float xn = C2[0] * invz, yn = C2[1] * invz;
  P.ex = (B.fx * xn + 0.0f * yn) + B.cx;
  P.ey = (0.0f * xn + B.fy * yn) + B.cy;
}

static bool tri_kf_ok(const vieo_tri_keyframe& K) {
  if (K.n_keys < 0 || K.n_nodes < 0 || K.n_levels <= 0 || !K.scale_factor || !K.level_sigma2) return false;
  if (K.n_keys > 0 && (!K.keys || !K.descriptors || !K.uright || !K.has_mappoint)) return false;
  if (K.n_nodes > 0 && (!K.node_id || !K.node_first || (K.node_first[K.n_nodes] > 0 && !K.node_feat))) return false;
  for (int n = 0; n < K.n_nodes; n++) {
    if (K.node_first[n + 1] < K.node_first[n] || (n > 0 && K.node_id[n] <= K.node_id[n - 1])) return false;
  }
  for (int i = 0; i < (K.n_nodes ? K.node_first[K.n_nodes] : 0); i++)
    if (K.node_feat[i] < 0 || K.node_feat[i] >= K.n_keys) return false;
  for (int i = 0; i < K.n_keys; i++)
    if (K.keys[i].octave < 0 || K.keys[i].octave >= K.n_levels) return false;
  return true;
}

static void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {  // ORBmatcher.cc:1608-1641
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = (int)histo[i].size();
    if (s > max1)
      max3 = max2, max2 = max1, max1 = s, ind3 = ind2, ind2 = ind1, ind1 = i;
    else if (s > max2)
      max3 = max2, max2 = s, ind3 = ind2, ind2 = i;
    else if (s > max3)
      max3 = s, ind3 = i;
  }
  if (max2 < 0.1f * (float)max1)
    ind2 = -1, ind3 = -1;
  else if (max3 < 0.1f * (float)max1)
    ind3 = -1;
}

struct TriScratch {
  DevBuf q, pairs, k1, d1, u1, k2, d2, u2, m2, f2, s2, g2, cand, cn;
};
static thread_local TriScratch g_tri;

}  // namespace vieo

using namespace vieo;

extern "C" int vieo_search_for_triangulation(const vieo_tri_keyframe* kf1, const vieo_tri_keyframe* kf2s, int n_kf2,
                                             int only_stereo, int check_orientation, int32_t pair_capacity,
                                             int32_t* h_pairs, int32_t* h_n_pairs, int32_t* h_n_matches) {
  if (!kf1 || !kf2s || n_kf2 <= 0 || pair_capacity < 0 || (pair_capacity > 0 && !h_pairs) || !h_n_pairs || !h_n_matches)
    return VIEO_E_INVALID;
  if (!tri_kf_ok(*kf1)) {
    set_error("SearchForTriangulation: pKF1 is inconsistent (nodes ascending, feature indices and octaves in range)");
    return VIEO_E_INVALID;
  }
  for (int p = 0; p < n_kf2; p++)
    if (!tri_kf_ok(kf2s[p])) {
      set_error("SearchForTriangulation: neighbour %d is inconsistent", p);
      return VIEO_E_INVALID;
    }
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  const vieo_tri_keyframe& A = *kf1;
  // ---- queries in the reference's order (shared nodes ascending, keys of pKF1 in the node's order)
  std::vector<TriPairDev> pairs(n_kf2);
  std::vector<TriQuery> queries;
  std::vector<int> q_begin(n_kf2 + 1, 0);
  size_t keys2 = 0, feats2 = 0, lvls2 = 0, n_cand = 0;
  for (int p = 0; p < n_kf2; p++) {
    const vieo_tri_keyframe& B = kf2s[p];
    tri_pair_setup(A, B, pairs[p]);
    pairs[p].key_off = (int)keys2, pairs[p].feat_off = (int)feats2, pairs[p].lvl_off = (int)lvls2, pairs[p].pad = 0;
    keys2 += B.n_keys, feats2 += B.n_nodes ? B.node_first[B.n_nodes] : 0, lvls2 += B.n_levels;
    q_begin[p] = (int)queries.size();
    int n1 = 0, n2 = 0;
    while (n1 < A.n_nodes && n2 < B.n_nodes) {
      if (A.node_id[n1] == B.node_id[n2]) {
        const int first2 = B.node_first[n2], count2 = B.node_first[n2 + 1] - first2;
        for (int i1 = A.node_first[n1]; i1 < A.node_first[n1 + 1] && count2 > 0; i1++) {
          const int idx1 = A.node_feat[i1];
          if (A.has_mappoint[idx1] || (only_stereo && !(A.uright[idx1] >= 0))) continue;
          queries.push_back({idx1, p, first2, count2, (int)n_cand});
          n_cand += count2;
        }
        n1++, n2++;
      } else if (A.node_id[n1] < B.node_id[n2])
        n1 = (int)(std::lower_bound(A.node_id + n1, A.node_id + A.n_nodes, B.node_id[n2]) - A.node_id);
      else
        n2 = (int)(std::lower_bound(B.node_id + n2, B.node_id + B.n_nodes, A.node_id[n1]) - B.node_id);
    }
  }
  q_begin[n_kf2] = (int)queries.size();
  const int nq = (int)queries.size();
  std::vector<int> cand_n(nq, 0);
  std::vector<int2> cand(std::max<size_t>(n_cand, 1));
  if (nq > 0) {
    TriScratch& S = g_tri;
    const size_t k1 = std::max(A.n_keys, 1), k2 = std::max<size_t>(keys2, 1);
    if ((rc = S.q.ensure(nq * sizeof(TriQuery))) != VIEO_OK || (rc = S.pairs.ensure(n_kf2 * sizeof(TriPairDev))) != VIEO_OK ||
        (rc = S.k1.ensure(k1 * sizeof(vieo_keypoint))) != VIEO_OK || (rc = S.d1.ensure(k1 * 32)) != VIEO_OK ||
        (rc = S.u1.ensure(k1 * 4)) != VIEO_OK || (rc = S.k2.ensure(k2 * sizeof(vieo_keypoint))) != VIEO_OK ||
        (rc = S.d2.ensure(k2 * 32)) != VIEO_OK || (rc = S.u2.ensure(k2 * 4)) != VIEO_OK || (rc = S.m2.ensure(k2)) != VIEO_OK ||
        (rc = S.f2.ensure(std::max<size_t>(feats2, 1) * 4)) != VIEO_OK || (rc = S.s2.ensure(lvls2 * 4)) != VIEO_OK ||
        (rc = S.g2.ensure(lvls2 * 4)) != VIEO_OK || (rc = S.cand.ensure(cand.size() * sizeof(int2))) != VIEO_OK ||
        (rc = S.cn.ensure(nq * 4)) != VIEO_OK)
      return rc;
    VIEO_HIP_CHECK(hipMemcpy(S.q.p, queries.data(), nq * sizeof(TriQuery), hipMemcpyHostToDevice));
    VIEO_HIP_CHECK(hipMemcpy(S.pairs.p, pairs.data(), n_kf2 * sizeof(TriPairDev), hipMemcpyHostToDevice));
    VIEO_HIP_CHECK(hipMemcpy(S.k1.p, A.keys, (size_t)A.n_keys * sizeof(vieo_keypoint), hipMemcpyHostToDevice));
    VIEO_HIP_CHECK(hipMemcpy(S.d1.p, A.descriptors, (size_t)A.n_keys * 32, hipMemcpyHostToDevice));
    VIEO_HIP_CHECK(hipMemcpy(S.u1.p, A.uright, (size_t)A.n_keys * 4, hipMemcpyHostToDevice));
    for (int p = 0; p < n_kf2; p++) {
      const vieo_tri_keyframe& B = kf2s[p];
      const TriPairDev& P = pairs[p];
      const size_t nk = B.n_keys, nf = B.n_nodes ? B.node_first[B.n_nodes] : 0;
      if (nk) {
        VIEO_HIP_CHECK(hipMemcpy(S.k2.as<vieo_keypoint>() + P.key_off, B.keys, nk * sizeof(vieo_keypoint), hipMemcpyHostToDevice));
        VIEO_HIP_CHECK(hipMemcpy(S.d2.as<uint8_t>() + 32 * (size_t)P.key_off, B.descriptors, nk * 32, hipMemcpyHostToDevice));
        VIEO_HIP_CHECK(hipMemcpy(S.u2.as<float>() + P.key_off, B.uright, nk * 4, hipMemcpyHostToDevice));
        VIEO_HIP_CHECK(hipMemcpy(S.m2.as<uint8_t>() + P.key_off, B.has_mappoint, nk, hipMemcpyHostToDevice));
      }
      if (nf) VIEO_HIP_CHECK(hipMemcpy(S.f2.as<int>() + P.feat_off, B.node_feat, nf * 4, hipMemcpyHostToDevice));
      VIEO_HIP_CHECK(hipMemcpy(S.s2.as<float>() + P.lvl_off, B.scale_factor, (size_t)B.n_levels * 4, hipMemcpyHostToDevice));
      VIEO_HIP_CHECK(hipMemcpy(S.g2.as<float>() + P.lvl_off, B.level_sigma2, (size_t)B.n_levels * 4, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_tri_gates, dim3((nq + 63) / 64), dim3(64), 0, nullptr, S.q.as<TriQuery>(), nq,
                       S.pairs.as<TriPairDev>(), S.k1.as<vieo_keypoint>(), S.d1.as<uint8_t>(), S.u1.as<float>(),
                       S.k2.as<vieo_keypoint>(), S.d2.as<uint8_t>(), S.u2.as<float>(), S.m2.as<uint8_t>(), S.f2.as<int>(),
                       S.s2.as<float>(), S.g2.as<float>(), only_stereo, S.cand.as<int2>(), S.cn.as<int>());
    VIEO_HIP_CHECK(hipGetLastError());
    VIEO_HIP_CHECK(hipMemcpy(cand_n.data(), S.cn.p, nq * 4, hipMemcpyDeviceToHost));
    VIEO_HIP_CHECK(hipMemcpy(cand.data(), S.cand.p, cand.size() * sizeof(int2), hipMemcpyDeviceToHost));
  }
  // ---- the order-dependent part, per neighbour (ORBmatcher.cc:962-1146)
  const float factor = 1.0f / kTriHisto;
  FeGroups G;
  for (int p = 0; p < n_kf2; p++) {
    const vieo_tri_keyframe& B = kf2s[p];
    const int32_t nk[2] = {A.n_keys, B.n_keys};
    G.reset(2, nk);
    std::vector<int> rotHist[kTriHisto];
    int nmatches = 0;
    for (int q = q_begin[p]; q < q_begin[p + 1]; q++) {
      const TriQuery& Q = queries[q];
      int bestDist = kTriThLow, bestIdx2 = -1;
      for (int k = 0; k < cand_n[q]; k++) {
        const int2 c = cand[Q.out_off + k];
        const int g = G.key2g[1][c.x];
        if (g >= 0 && G.idxs[(size_t)g * 2] != -1) continue;  // pKF2's key already belongs to a match
        if (c.y > bestDist) continue;
        bestIdx2 = c.x, bestDist = c.y;
      }
      if (bestIdx2 < 0) continue;
      if (fe_fill(G, 0, Q.idx1, 1, bestIdx2, (float)bestDist, true, nullptr)) ++nmatches;
      if (check_orientation) {
        float rot = A.keys[Q.idx1].angle - B.keys[bestIdx2].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == kTriHisto) bin = 0;
        if (bin < 0 || bin >= kTriHisto) {
          set_error("SearchForTriangulation: key angles outside [0, 360)");
          return VIEO_E_INVALID;
        }
        rotHist[bin].push_back(Q.idx1);
      }
    }
    if (check_orientation) {
      int ind1 = -1, ind2 = -1, ind3 = -1;
      three_maxima(rotHist, kTriHisto, ind1, ind2, ind3);
      for (int i = 0; i < kTriHisto; i++) {
        if (i == ind1 || i == ind2 || i == ind3) continue;
        for (int idx1 : rotHist[i]) {
          const int g = G.key2g[0][idx1];
          if (g < 0) continue;
          G.good[g] = 0;
          nmatches--;
        }
      }
    }
    int np = 0;
    for (int g = 0; g < G.size(); g++) {
      const int i1 = G.idxs[(size_t)g * 2], i2 = G.idxs[(size_t)g * 2 + 1];
      if (i1 < 0 || i2 < 0 || !G.good[g]) continue;
      if (np < pair_capacity) h_pairs[((size_t)p * pair_capacity + np) * 2] = i1, h_pairs[((size_t)p * pair_capacity + np) * 2 + 1] = i2;
      np++;
    }
    h_n_pairs[p] = np, h_n_matches[p] = nmatches;
    if (np > pair_capacity) {
      set_error("SearchForTriangulation: neighbour %d has %d pairs, capacity %d", p, np, pair_capacity);
      return VIEO_E_CAPACITY;
    }
  }
  return VIEO_OK;
}
