// tri_search.hip -- ORBmatcher::SearchForTriangulation (reference src/ORBmatcher.cc:896-1150), a batch of
// neighbours pKF2 of one pKF1 per call; undistorted pinhole key frames and distorted camera rigs.
//   k_tri_points  rigs: K * UnProject(key) of every key once (the image point epipolarConstrain works on)
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

#include "cam_unproject.h"
#include "common.h"
#include "match_groups.h"

namespace vieo {

static const int kTriThLow = 50, kTriHisto = 30;

struct TriPairDev {  // per neighbour
  double F12[4][4][9];  // per (camera of pKF1, camera of pKF2)
  float C2[3];          // pKF1's reference camera centre in pKF2's reference frame (float like the reference)
  float ex, ey;         // its image = the epipole (pinhole: from the host; rig: k_tri_epipole)
  int key_off, feat_off, lvl_off;  // offsets of pKF2's arrays in the concatenated buffers
  int rig;
};

struct TriKfDev {  // per key frame (0 = pKF1, 1 + p = neighbour p)
  vieo_camera cam[4];
  int n_cams, key_off, n_keys, pad;
};

struct TriQuery {
  int idx1, pair, first2, count2, out_off;
};

__device__ __forceinline__ int tri_hamming(const uint4* a, const uint4* b) {
  const uint4 a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
         __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// the image point of every key: the key itself (undistorted), or K * UnProject(key) (bkp_distort, camera_base.h:372-384)
__global__ void __launch_bounds__(64)
k_tri_points(const TriKfDev* __restrict__ kfs, const vieo_keypoint* __restrict__ keys, const uint8_t* __restrict__ key_cam,
             double2* __restrict__ pts, uint8_t* __restrict__ ok) {
  const TriKfDev& K = kfs[blockIdx.y];
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= K.n_keys) return;
  const int g = K.key_off + i;
  const vieo_keypoint kp = keys[g];
  if (K.n_cams == 0) {
    pts[g] = make_double2((double)kp.x, (double)kp.y), ok[g] = 1;
    return;
  }
  CamD c;
  cam_from_abi(K.cam[key_cam[g] & 3], c);
  double X[3];
  cam_unproject(c, kp.x, kp.y, X);
  const double q0 = (c.fx * X[0] + 0.0 * X[1]) + c.cx * X[2], q1 = (0.0 * X[0] + c.fy * X[1]) + c.cy * X[2];
  const double q2 = (0.0 * X[0] + 0.0 * X[1]) + 1.0 * X[2];
  const bool fin = isfinite(q0) && isfinite(q1) && isfinite(q2);
  const double invz = 1. / q2;
  pts[g] = make_double2(q0 * invz, q1 * invz), ok[g] = fin ? 1 : 0;
}

// rigs: pKF2->mpCameras[0]->Project(C2) (ORBmatcher.cc:922-927)
__global__ void k_tri_epipole(TriPairDev* __restrict__ pairs, const TriKfDev* __restrict__ kfs, int n_pairs) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs || !pairs[p].rig) return;
  CamD c;
  cam_from_abi(kfs[1 + p].cam[0], c);
  const double C2[3] = {(double)pairs[p].C2[0], (double)pairs[p].C2[1], (double)pairs[p].C2[2]};
  double uv[2];
  cam_project(c, C2, uv, nullptr);
  pairs[p].ex = (float)uv[0], pairs[p].ey = (float)uv[1];
}

__global__ void __launch_bounds__(64)
k_tri_gates(const TriQuery* __restrict__ queries, int n_queries, const TriPairDev* __restrict__ pairs,
            const vieo_keypoint* __restrict__ keys, const uint8_t* __restrict__ desc, const float* __restrict__ ur,
            const uint8_t* __restrict__ mp, const uint8_t* __restrict__ key_cam, const double2* __restrict__ pts,
            const uint8_t* __restrict__ pt_ok, const int* __restrict__ feat2, const float* __restrict__ scale2,
            const float* __restrict__ sigma2, int only_stereo, int2* __restrict__ cand, int* __restrict__ cand_n) {
  const int q = blockIdx.x * 64 + threadIdx.x;
  if (q >= n_queries) return;
  const TriQuery Q = queries[q];
  const TriPairDev& P = pairs[Q.pair];
  const bool st1 = ur[Q.idx1] >= 0;  // pKF1's keys sit at offset 0
  const uint4* d1 = (const uint4*)(desc + 32 * (size_t)Q.idx1);
  const double2 p1 = pts[Q.idx1];
  const bool ok1 = pt_ok[Q.idx1] != 0;
  const int cam1 = P.rig ? (key_cam[Q.idx1] & 3) : 0;
  int n = 0;
  int2* out = cand + Q.out_off;
  for (int k = 0; k < Q.count2; k++) {
    const int idx2 = feat2[P.feat_off + Q.first2 + k];
    const int g2 = P.key_off + idx2;
    if (mp[g2]) continue;
    const bool st2 = ur[g2] >= 0;
    if (only_stereo && !st2) continue;
    const int dist = tri_hamming(d1, (const uint4*)(desc + 32 * (size_t)g2));
    if (dist > kTriThLow) continue;
    const vieo_keypoint kp2 = keys[g2];
    if (!st1 && !st2) {
      const float distex = P.ex - kp2.x, distey = P.ey - kp2.y;
      if (distex * distex + distey * distey < 100 * scale2[P.lvl_off + kp2.octave]) continue;
    }
    if (!ok1 || !pt_ok[g2]) continue;
    // the epipolar line of key 1 in image 2 (Tdata = float)
    const double* F = P.F12[cam1][P.rig ? (key_cam[g2] & 3) : 0];
    const float a = (float)(p1.x * F[0] + p1.y * F[3] + F[6]);
    const float b = (float)(p1.x * F[1] + p1.y * F[4] + F[7]);
    const float c = (float)(p1.x * F[2] + p1.y * F[5] + F[8]);
    const double2 p2 = pts[g2];
    const float num = (float)((double)a * p2.x + (double)b * p2.y + (double)c);
    const float den = a * a + b * b;
    if (den == 0) continue;
    const float dsqr = num * num / den;
    if (!(dsqr < 3.84f * sigma2[P.lvl_off + kp2.octave])) continue;
    out[n++] = make_int2(idx2, dist);
  }
  cand_n[q] = n;
}

// Tr1r2 = (Tcw1 * Twc2).cast<float>(), T12 = Tcr(cam1) * Tr1r2 * Trc(cam2), F12 = K1^-T [t12]x R12 K2^-1,
// C2 = pKF1's camera centre in pKF2's frame; the epipole of an undistorted pair
static void tri_pair_setup(const vieo_tri_keyframe& A, const vieo_tri_keyframe& B, TriPairDev& P) {
  const bool rig = A.n_cams > 0;
  const int nc1 = rig ? A.n_cams : 1, nc2 = rig ? B.n_cams : 1;
  P.rig = rig ? 1 : 0;
  const double *T1 = A.Tcw, *T2 = B.Tcw;
  double Rr[9], tr[3];
  for (int i = 0; i < 3; i++) {
    double s = 0;
    for (int j = 0; j < 3; j++) {
      double r = 0;
      for (int k = 0; k < 3; k++) r += T1[i * 4 + k] * T2[j * 4 + k];
      Rr[i * 3 + j] = (double)(float)r;
      s += r * T2[j * 4 + 3];
    }
    tr[i] = (double)(float)(T1[i * 4 + 3] - s);
  }
  auto mul = [](const double* X, const double* Y, double* Z) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Z[i * 3 + j] = X[i * 3] * Y[j] + X[i * 3 + 1] * Y[3 + j] + X[i * 3 + 2] * Y[6 + j];
  };
  memset(P.F12, 0, sizeof(P.F12));
  for (int c1 = 0; c1 < nc1; c1++)
    for (int c2 = 0; c2 < nc2; c2++) {
      double R12[9], t12[3];
      if (!rig) {
        memcpy(R12, Rr, 72), memcpy(t12, tr, 24);
      } else {
        const double* Tcr = A.Tcr + 12 * c1;
        const double* Trc = B.Trc + 12 * c2;
        double Ra[9], Rb[9], M[9], v[3];
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) Ra[i * 3 + j] = Tcr[i * 4 + j], Rb[i * 3 + j] = Trc[i * 4 + j];
        mul(Ra, Rr, M), mul(M, Rb, R12);
        for (int i = 0; i < 3; i++) v[i] = tr[i] + (Rr[i * 3] * Trc[3] + Rr[i * 3 + 1] * Trc[7] + Rr[i * 3 + 2] * Trc[11]);
        for (int i = 0; i < 3; i++) t12[i] = Tcr[i * 4 + 3] + (Ra[i * 3] * v[0] + Ra[i * 3 + 1] * v[1] + Ra[i * 3 + 2] * v[2]);
        for (int i = 0; i < 9; i++) R12[i] = (double)(float)R12[i];
        for (int i = 0; i < 3; i++) t12[i] = (double)(float)t12[i];
      }
      const double fx1 = rig ? A.cams[c1].fx : A.fx, fy1 = rig ? A.cams[c1].fy : A.fy;
      const double cx1 = rig ? A.cams[c1].cx : A.cx, cy1 = rig ? A.cams[c1].cy : A.cy;
      const double fx2 = rig ? B.cams[c2].fx : B.fx, fy2 = rig ? B.cams[c2].fy : B.fy;
      const double cx2 = rig ? B.cams[c2].cx : B.cx, cy2 = rig ? B.cams[c2].cy : B.cy;
      const double K1it[9] = {1 / fx1, 0, 0, 0, 1 / fy1, 0, -cx1 / fx1, -cy1 / fy1, 1};
      const double K2i[9] = {1 / fx2, 0, -cx2 / fx2, 0, 1 / fy2, -cy2 / fy2, 0, 0, 1};
      const double H[9] = {0, -t12[2], t12[1], t12[2], 0, -t12[0], -t12[1], t12[0], 0};
      double M1[9], M2[9];
      mul(K1it, H, M1), mul(M1, R12, M2), mul(M2, K2i, P.F12[c1][c2]);
    }
  float R1f[9], t1f[3], R2f[9], t2f[3], Cw[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R1f[i * 3 + j] = (float)T1[i * 4 + j], R2f[i * 3 + j] = (float)T2[i * 4 + j];
    t1f[i] = (float)T1[i * 4 + 3], t2f[i] = (float)T2[i * 4 + 3];
  }
  for (int i = 0; i < 3; i++)
    Cw[i] = (float)(-((double)R1f[i] * t1f[0] + (double)R1f[3 + i] * t1f[1] + (double)R1f[6 + i] * t1f[2]));
  for (int i = 0; i < 3; i++)
    P.C2[i] = (float)((double)R2f[i * 3] * Cw[0] + (double)R2f[i * 3 + 1] * Cw[1] + (double)R2f[i * 3 + 2] * Cw[2] +
                      (double)t2f[i]);
  P.ex = P.ey = 0;
  if (!rig) {
    const float invz = 1.0f / P.C2[2];
    const float xn = P.C2[0] * invz, yn = P.C2[1] * invz;
    P.ex = (B.fx * xn + 0.0f * yn) + B.cx;
    P.ey = (0.0f * xn + B.fy * yn) + B.cy;
  }
}

static bool tri_kf_ok(const vieo_tri_keyframe& K) {
  if (K.n_keys < 0 || K.n_nodes < 0 || K.n_levels <= 0 || !K.scale_factor || !K.level_sigma2) return false;
  if (K.n_keys > 0 && (!K.keys || !K.descriptors || !K.uright || !K.has_mappoint)) return false;
  if (K.n_cams < 0 || K.n_cams > 4 || (K.n_cams > 0 && (!K.cams || !K.Tcr || !K.Trc || (K.n_keys > 0 && !K.key_cam)))) return false;
  if (K.n_nodes > 0 && (!K.node_id || !K.node_first || (K.node_first[K.n_nodes] > 0 && !K.node_feat))) return false;
  for (int n = 0; n < K.n_nodes; n++) {
    if (K.node_first[n + 1] < K.node_first[n] || (n > 0 && K.node_id[n] <= K.node_id[n - 1])) return false;
  }
  for (int i = 0; i < (K.n_nodes ? K.node_first[K.n_nodes] : 0); i++)
    if (K.node_feat[i] < 0 || K.node_feat[i] >= K.n_keys) return false;
  for (int i = 0; i < K.n_keys; i++)
    if (K.keys[i].octave < 0 || K.keys[i].octave >= K.n_levels || (K.n_cams > 0 && K.key_cam[i] >= K.n_cams)) return false;
  for (int c = 0; c < K.n_cams; c++) {
    CamD d;
    if (!cam_from_abi(K.cams[c], d)) return false;
  }
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
  DevBuf q, pairs, kfs, k, d, u, m, kc, pts, ok, f2, s2, g2, cand, cn;
};
static thread_local TriScratch g_tri;

}  // namespace vieo

using namespace vieo;

extern "C" int vieo_search_for_triangulation(const vieo_tri_keyframe* kf1, const vieo_tri_keyframe* kf2s, int n_kf2,
                                             int only_stereo, int check_orientation, int32_t pair_capacity,
                                             int32_t pair_stride, int32_t* h_pairs, int32_t* h_n_pairs,
                                             int32_t* h_n_matches) {
  if (!kf1 || !kf2s || n_kf2 <= 0 || pair_capacity < 0 || (pair_capacity > 0 && !h_pairs) || !h_n_pairs || !h_n_matches)
    return VIEO_E_INVALID;
  if (!tri_kf_ok(*kf1)) {
    set_error("SearchForTriangulation: pKF1 is inconsistent (nodes ascending, feature / camera indices and octaves in range)");
    return VIEO_E_INVALID;
  }
  const vieo_tri_keyframe& A = *kf1;
  const int nc1 = A.n_cams > 0 ? A.n_cams : 1;
  for (int p = 0; p < n_kf2; p++) {
    if (!tri_kf_ok(kf2s[p])) {
      set_error("SearchForTriangulation: neighbour %d is inconsistent", p);
      return VIEO_E_INVALID;
    }
    if ((kf2s[p].n_cams > 0) != (A.n_cams > 0)) {  // assert(!usedistort[1]) / assert(usedistort[1]), ORBmatcher.cc:949,952
      set_error("SearchForTriangulation: neighbour %d and pKF1 are not of one kind (undistorted / rig)", p);
      return VIEO_E_INVALID;
    }
    if (pair_stride < nc1 + (kf2s[p].n_cams > 0 ? kf2s[p].n_cams : 1)) {
      set_error("SearchForTriangulation: pair_stride %d is smaller than the cameras of pair %d", pair_stride, p);
      return VIEO_E_INVALID;
    }
  }
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  // ---- queries in the reference's order (shared nodes ascending, keys of pKF1 in the node's order)
  std::vector<TriPairDev> pairs(n_kf2);
  std::vector<TriKfDev> kfd(1 + n_kf2);
  std::vector<TriQuery> queries;
  std::vector<int> q_begin(n_kf2 + 1, 0);
  size_t keys_all = A.n_keys, feats2 = 0, lvls2 = 0, n_cand = 0;
  int max_keys = A.n_keys;
  auto fill_kf = [](const vieo_tri_keyframe& K, int key_off, TriKfDev& d) {
    memset(&d, 0, sizeof(d));
    for (int c = 0; c < K.n_cams; c++) d.cam[c] = K.cams[c];
    d.n_cams = K.n_cams, d.key_off = key_off, d.n_keys = K.n_keys;
  };
  fill_kf(A, 0, kfd[0]);
  for (int p = 0; p < n_kf2; p++) {
    const vieo_tri_keyframe& B = kf2s[p];
    tri_pair_setup(A, B, pairs[p]);
    pairs[p].key_off = (int)keys_all, pairs[p].feat_off = (int)feats2, pairs[p].lvl_off = (int)lvls2;
    fill_kf(B, (int)keys_all, kfd[1 + p]);
    keys_all += B.n_keys, feats2 += B.n_nodes ? B.node_first[B.n_nodes] : 0, lvls2 += B.n_levels;
    max_keys = std::max(max_keys, B.n_keys);
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
    const size_t ka = std::max<size_t>(keys_all, 1);
    if ((rc = S.q.ensure(nq * sizeof(TriQuery))) != VIEO_OK || (rc = S.pairs.ensure(n_kf2 * sizeof(TriPairDev))) != VIEO_OK ||
        (rc = S.kfs.ensure((1 + n_kf2) * sizeof(TriKfDev))) != VIEO_OK ||
        (rc = S.k.ensure(ka * sizeof(vieo_keypoint))) != VIEO_OK || (rc = S.d.ensure(ka * 32)) != VIEO_OK ||
        (rc = S.u.ensure(ka * 4)) != VIEO_OK || (rc = S.m.ensure(ka)) != VIEO_OK || (rc = S.kc.ensure(ka)) != VIEO_OK ||
        (rc = S.pts.ensure(ka * sizeof(double2))) != VIEO_OK || (rc = S.ok.ensure(ka)) != VIEO_OK ||
        (rc = S.f2.ensure(std::max<size_t>(feats2, 1) * 4)) != VIEO_OK || (rc = S.s2.ensure(lvls2 * 4)) != VIEO_OK ||
        (rc = S.g2.ensure(lvls2 * 4)) != VIEO_OK || (rc = S.cand.ensure(cand.size() * sizeof(int2))) != VIEO_OK ||
        (rc = S.cn.ensure(nq * 4)) != VIEO_OK)
      return rc;
    VIEO_HIP_CHECK(hipMemcpy(S.q.p, queries.data(), nq * sizeof(TriQuery), hipMemcpyHostToDevice));
    VIEO_HIP_CHECK(hipMemcpy(S.pairs.p, pairs.data(), n_kf2 * sizeof(TriPairDev), hipMemcpyHostToDevice));
    VIEO_HIP_CHECK(hipMemcpy(S.kfs.p, kfd.data(), (1 + n_kf2) * sizeof(TriKfDev), hipMemcpyHostToDevice));
    for (int f = 0; f <= n_kf2; f++) {
      const vieo_tri_keyframe& K = f == 0 ? A : kf2s[f - 1];
      const size_t nk = K.n_keys, off = kfd[f].key_off;
      if (!nk) continue;
      VIEO_HIP_CHECK(hipMemcpy(S.k.as<vieo_keypoint>() + off, K.keys, nk * sizeof(vieo_keypoint), hipMemcpyHostToDevice));
      VIEO_HIP_CHECK(hipMemcpy(S.d.as<uint8_t>() + 32 * off, K.descriptors, nk * 32, hipMemcpyHostToDevice));
      VIEO_HIP_CHECK(hipMemcpy(S.u.as<float>() + off, K.uright, nk * 4, hipMemcpyHostToDevice));
      VIEO_HIP_CHECK(hipMemcpy(S.m.as<uint8_t>() + off, K.has_mappoint, nk, hipMemcpyHostToDevice));
      if (K.n_cams > 0)
        VIEO_HIP_CHECK(hipMemcpy(S.kc.as<uint8_t>() + off, K.key_cam, nk, hipMemcpyHostToDevice));
      else
        VIEO_HIP_CHECK(hipMemset(S.kc.as<uint8_t>() + off, 0, nk));
    }
    for (int p = 0; p < n_kf2; p++) {
      const vieo_tri_keyframe& B = kf2s[p];
      const TriPairDev& P = pairs[p];
      const size_t nf = B.n_nodes ? B.node_first[B.n_nodes] : 0;
      if (nf) VIEO_HIP_CHECK(hipMemcpy(S.f2.as<int>() + P.feat_off, B.node_feat, nf * 4, hipMemcpyHostToDevice));
      VIEO_HIP_CHECK(hipMemcpy(S.s2.as<float>() + P.lvl_off, B.scale_factor, (size_t)B.n_levels * 4, hipMemcpyHostToDevice));
      VIEO_HIP_CHECK(hipMemcpy(S.g2.as<float>() + P.lvl_off, B.level_sigma2, (size_t)B.n_levels * 4, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_tri_points, dim3((std::max(max_keys, 1) + 63) / 64, 1 + n_kf2), dim3(64), 0, nullptr,
                       S.kfs.as<TriKfDev>(), S.k.as<vieo_keypoint>(), S.kc.as<uint8_t>(), S.pts.as<double2>(), S.ok.as<uint8_t>());
    hipLaunchKernelGGL(k_tri_epipole, dim3((n_kf2 + 63) / 64), dim3(64), 0, nullptr, S.pairs.as<TriPairDev>(),
                       S.kfs.as<TriKfDev>(), n_kf2);
    hipLaunchKernelGGL(k_tri_gates, dim3((nq + 63) / 64), dim3(64), 0, nullptr, S.q.as<TriQuery>(), nq,
                       S.pairs.as<TriPairDev>(), S.k.as<vieo_keypoint>(), S.d.as<uint8_t>(), S.u.as<float>(),
                       S.m.as<uint8_t>(), S.kc.as<uint8_t>(), S.pts.as<double2>(), S.ok.as<uint8_t>(), S.f2.as<int>(),
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
    const bool rig = A.n_cams > 0;
    const int nc2 = rig ? B.n_cams : 1, nc = nc1 + nc2;
    int32_t nk[8];
    for (int c = 0; c < nc; c++) nk[c] = c < nc1 ? A.n_keys : B.n_keys;  // tables indexed by the key's global index
    G.reset(nc, nk);
    std::vector<int> rotHist[kTriHisto];
    int nmatches = 0;
    for (int q = q_begin[p]; q < q_begin[p + 1]; q++) {
      const TriQuery& Q = queries[q];
      const int cam1 = rig ? A.key_cam[Q.idx1] : 0;
      int bestDist[4] = {kTriThLow, kTriThLow, kTriThLow, kTriThLow}, bestIdx2[4] = {-1, -1, -1, -1};  // per image
      for (int k = 0; k < cand_n[q]; k++) {
        const int2 c = cand[Q.out_off + k];
        const int img = rig ? B.key_cam[c.x] : 0;
        const int g = G.key2g[nc1 + img][c.x];
        if (g >= 0 && G.idxs[(size_t)g * nc + cam1] != -1) continue;  // pKF2's key already matched to this camera
        if (c.y > bestDist[img]) continue;
        bestIdx2[img] = c.x, bestDist[img] = c.y;
      }
      for (int img = 0; img < nc2; img++) {
        if (bestIdx2[img] < 0) continue;
        if (fe_fill(G, cam1, Q.idx1, nc1 + img, bestIdx2[img], (float)bestDist[img], true, nullptr)) ++nmatches;
        if (check_orientation) {
          float rot = A.keys[Q.idx1].angle - B.keys[bestIdx2[img]].angle;
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
    }
    if (check_orientation) {
      int ind1 = -1, ind2 = -1, ind3 = -1;
      three_maxima(rotHist, kTriHisto, ind1, ind2, ind3);
      for (int i = 0; i < kTriHisto; i++) {
        if (i == ind1 || i == ind2 || i == ind3) continue;
        for (int idx1 : rotHist[i]) {
          const int g = G.key2g[rig ? A.key_cam[idx1] : 0][idx1];
          if (g < 0) continue;
          G.good[g] = 0;
          nmatches--;
        }
      }
    }
    int np = 0;
    for (int g = 0; g < G.size(); g++) {
      int cnt = 0;
      for (int c = 0; c < nc; c++) cnt += G.idxs[(size_t)g * nc + c] != -1;
      if (cnt < 2 || !G.good[g]) continue;
      if (np < pair_capacity) {
        int32_t* row = h_pairs + ((size_t)p * pair_capacity + np) * pair_stride;
        for (int c = 0; c < pair_stride; c++) row[c] = c < nc ? G.idxs[(size_t)g * nc + c] : -1;
      }
      np++;
    }
    h_n_pairs[p] = np, h_n_matches[p] = nmatches;
    if (np > pair_capacity) {
      set_error("SearchForTriangulation: neighbour %d has %d matches, capacity %d", p, np, pair_capacity);
      return VIEO_E_CAPACITY;
    }
  }
  return VIEO_OK;
}
