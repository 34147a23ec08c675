// ORACLE -- TEST INFRASTRUCTURE ONLY (see ocv_prims.hpp header).  PARITY UNPINNED.
//
// CPU restatement of int ORBmatcher::SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo)
// (reference src/ORBmatcher.cc:896-1150) for key frames with one undistorted pinhole camera each, with what it calls:
//   GeometricCamera::epipolarConstrain, fundamental-matrix branch      common/camera_models/camera_base.h:287-406
//     (USE_DIR_EPI_ERR is commented out: :295; Tdata = float, Tcalc = double: common/config.h:23-24)
//   GeometricCamera::FillMatchesFromPair without key points / sigmas   camera_base.h:408-574  (no triangulation,
//     USE_STRATEGY_MIN_DIST bookkeeping only)
//   ORBmatcher::ComputeThreeMaxima                                      src/ORBmatcher.cc:1608-1641
// The loops run in the reference's order.  Where the reference goes through float cv::Mat / Sophus float casts
// (camera centre, Tr1r2), the values are rounded to float at the same places; the order of the float operations
// inside those library calls is not observable from /root/reference, so gates whose value sits within a float ulp
// of its threshold may differ from a real run (they cannot differ between this file and the HIP path, which
// follow the same expressions).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

#include "../include/vieo_hot.h"

namespace vo {

static const int kThLow = 50, kHistoLength = 30;

struct TriPair {  // per (pKF1, pKF2): what the loops need besides the keys
  double F12[9];
  float ex, ey;
};

static inline float f32(double v) { return (float)v; }

// Tr1r2 = (Tcw1 * Twc2).cast<float>(), F12 = K1^-T [t12]x R12 K2^-1 in double, the epipole of camera 1 in image 2
static void tri_pair_setup(const vieo_tri_keyframe& A, const vieo_tri_keyframe& B, TriPair& P) {
  const double* T1 = A.Tcw;
  const double* T2 = B.Tcw;
  double R12[9], t12[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += T1[i * 4 + k] * T2[j * 4 + k];  // R1 R2^T
      R12[i * 3 + j] = (double)f32(s);
    }
  }
  for (int i = 0; i < 3; i++) {
    double s = 0;  // t1 - R12 t2, with the unrounded rotation
    for (int j = 0; j < 3; j++) {
      double r = 0;
      for (int k = 0; k < 3; k++) r += T1[i * 4 + k] * T2[j * 4 + k];
      s += r * T2[j * 4 + 3];
    }
    t12[i] = (double)f32(T1[i * 4 + 3] - s);
  }
  const double fx1 = A.fx, fy1 = A.fy, cx1 = A.cx, cy1 = A.cy, fx2 = B.fx, fy2 = B.fy, cx2 = B.cx, cy2 = B.cy;
  const double K1it[9] = {1 / fx1, 0, 0, 0, 1 / fy1, 0, -cx1 / fx1, -cy1 / fy1, 1};  // (K1^T)^-1
  const double K2i[9] = {1 / fx2, 0, -cx2 / fx2, 0, 1 / fy2, -cy2 / fy2, 0, 0, 1};
  const double H[9] = {0, -t12[2], t12[1], t12[2], 0, -t12[0], -t12[1], t12[0], 0};
  double M1[9], M2[9];
  auto mul = [](const double* X, const double* Y, double* Z) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Z[i * 3 + j] = X[i * 3] * Y[j] + X[i * 3 + 1] * Y[3 + j] + X[i * 3 + 2] * Y[6 + j];
  };
  mul(K1it, H, M1);
  mul(M1, R12, M2);
  mul(M2, K2i, P.F12);
  // Cw = -R1^T t1 (float cv::Mat), C2 = R2w Cw + t2w, epipole = K2 (C2 / C2z)
  float R1f[9], t1f[3], R2f[9], t2f[3], Cw[3], C2[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R1f[i * 3 + j] = f32(T1[i * 4 + j]), R2f[i * 3 + j] = f32(T2[i * 4 + j]);
    t1f[i] = f32(T1[i * 4 + 3]), t2f[i] = f32(T2[i * 4 + 3]);
  }
  for (int i = 0; i < 3; i++)
    Cw[i] = f32(-((double)R1f[i] * t1f[0] + (double)R1f[3 + i] * t1f[1] + (double)R1f[6 + i] * t1f[2]));
  for (int i = 0; i < 3; i++)
    C2[i] = f32((double)R2f[i * 3] * Cw[0] + (double)R2f[i * 3 + 1] * Cw[1] + (double)R2f[i * 3 + 2] * Cw[2] + (double)t2f[i]);
  const float invz = 1.0f / C2[2];
  const float xn = C2[0] * invz, yn = C2[1] * invz;
  P.ex = (B.fx * xn + 0.0f * yn) + B.cx;
  P.ey = (0.0f * xn + B.fy * yn) + B.cy;
}

static inline int hamming256(const uint8_t* a, const uint8_t* b) {
  int d = 0;
  for (int i = 0; i < 32; i++) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
  return d;
}

// every gate of the inner loop that does not depend on earlier matches; returns the Hamming distance or -1
static int tri_gates(const vieo_tri_keyframe& A, const vieo_tri_keyframe& B, const TriPair& P, int idx1, int idx2) {
  const int dist = hamming256(A.descriptors + 32 * (size_t)idx1, B.descriptors + 32 * (size_t)idx2);
  if (dist > kThLow) return -1;
  const vieo_keypoint &kp1 = A.keys[idx1], &kp2 = B.keys[idx2];
  const bool st1 = A.uright[idx1] >= 0, st2 = B.uright[idx2] >= 0;
  if (!st1 && !st2) {
    const float distex = P.ex - kp2.x, distey = P.ey - kp2.y;
    if (distex * distex + distey * distey < 100 * B.scale_factor[kp2.octave]) return -1;
  }
  const double p1x = kp1.x, p1y = kp1.y, p2x = kp2.x, p2y = kp2.y;
  const double* F = P.F12;
  const float a = f32(p1x * F[0] + p1y * F[3] + F[6]);
  const float b = f32(p1x * F[1] + p1y * F[4] + F[7]);
  const float c = f32(p1x * F[2] + p1y * F[5] + F[8]);
  const float num = f32((double)a * p2x + (double)b * p2y + (double)c);
  const float den = a * a + b * b;
  if (den == 0) return -1;
  const float dsqr = num * num / den;
  return dsqr < 3.84f * B.level_sigma2[kp2.octave] ? dist : -1;
}

struct TriGroups {  // two cameras: 0 = pKF1's, 1 = pKF2's
  std::vector<int> idx[2];
  std::vector<float> last[2];
  std::vector<bool> good;
  std::map<std::pair<int, int>, int> map;
};

static bool tri_fill(TriGroups& G, int idxi, int idxj, float dist) {  // FillMatchesFromPair, bdepth_ok == true
  const std::pair<int, int> ki(0, idxi), kj(1, idxj);
  auto iteri = G.map.find(ki), iterj = G.map.find(kj);
  if (iteri == G.map.end() && iterj != G.map.end()) iteri = iterj;
  int check[2] = {0, 0}, contradict = 0, g = -1;
  if (iteri != G.map.end()) {
    g = iteri->second;
    contradict = (iterj != G.map.end() && iterj->second != g) ? 2 : 0;
    if (contradict) {
      const int gj = iterj->second;
      float sum[2] = {0, 0};
      int cnt[2] = {0, 0};
      for (int t = 0; t < 2; t++) {
        if (G.idx[t][g] >= 0) sum[0] += G.last[t][g], ++cnt[0];
        if (G.idx[t][gj] >= 0) sum[1] += G.last[t][gj], ++cnt[1];
      }
      if (sum[1] * cnt[0] < sum[0] * cnt[1]) g = gj, contradict = 1;
    }
    if (G.idx[0][g] < 0 || (idxi != G.idx[0][g] && G.last[0][g] > dist)) check[0] = 2;
    if (G.idx[1][g] < 0 || (idxj != G.idx[1][g] && G.last[1][g] > dist)) check[1] = 2;
  } else
    check[0] = check[1] = 1;
  if (!(check[0] || check[1])) return false;
  if (check[0] == 1) {
    g = (int)G.good.size();
    G.map.emplace(ki, g), G.map.emplace(kj, g);
    G.idx[0].push_back(idxi), G.idx[1].push_back(idxj);
    G.last[0].push_back(dist), G.last[1].push_back(dist);
    G.good.push_back(true);
    return true;
  }
  if (contradict) {
    const int gc = contradict == 1 ? iteri->second : iterj->second;
    if (idxi == G.idx[0][gc]) G.map.erase(ki), G.last[0][gc] = INFINITY, G.idx[0][gc] = -1;
    if (idxj == G.idx[1][gc]) G.map.erase(kj), G.last[1][gc] = INFINITY, G.idx[1][gc] = -1;
  }
  const int id[2] = {idxi, idxj};
  for (int t = 0; t < 2; t++) {
    if (check[t] == 2) {
      if (id[t] != G.idx[t][g]) {
        if (G.idx[t][g] >= 0) G.map.erase(std::make_pair(t, G.idx[t][g]));
        G.map.emplace(std::make_pair(t, id[t]), g);
        G.idx[t][g] = id[t];
      }
      G.last[t][g] = dist;
    } else if (G.last[t][g] > dist)
      G.last[t][g] = dist;
  }
  return true;
}

static void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = (int)histo[i].size();
    if (s > max1) {
      max3 = max2, max2 = max1, max1 = s;
      ind3 = ind2, ind2 = ind1, ind1 = i;
    } else if (s > max2) {
      max3 = max2, max2 = s;
      ind3 = ind2, ind2 = i;
    } else if (s > max3) {
      max3 = s, ind3 = i;
    }
  }
  if (max2 < 0.1f * (float)max1) {
    ind2 = -1, ind3 = -1;
  } else if (max3 < 0.1f * (float)max1) {
    ind3 = -1;
  }
}

static int search_for_triangulation(const vieo_tri_keyframe& A, const vieo_tri_keyframe& B, bool only_stereo,
                                    bool check_orientation, std::vector<std::pair<int, int>>& pairs) {
  TriPair P;
  tri_pair_setup(A, B, P);
  int nmatches = 0;
  std::vector<int> rotHist[kHistoLength];
  const float factor = 1.0f / kHistoLength;
  TriGroups G;
  int n1 = 0, n2 = 0;
  while (n1 < A.n_nodes && n2 < B.n_nodes) {
    if (A.node_id[n1] == B.node_id[n2]) {
      for (int i1 = A.node_first[n1]; i1 < A.node_first[n1 + 1]; i1++) {
        const int idx1 = A.node_feat[i1];
        if (A.has_mappoint[idx1]) continue;
        const bool st1 = A.uright[idx1] >= 0;
        if (only_stereo && !st1) continue;
        int bestDist = kThLow, bestIdx2 = -1;
        for (int i2 = B.node_first[n2]; i2 < B.node_first[n2 + 1]; i2++) {
          const int idx2 = B.node_feat[i2];
          if (B.has_mappoint[idx2]) continue;
          auto it = G.map.find(std::make_pair(1, idx2));
          if (it != G.map.end() && G.idx[0][it->second] != -1) continue;  // already matched to a key of pKF1
          if (only_stereo && !(B.uright[idx2] >= 0)) continue;
          const int dist = tri_gates(A, B, P, idx1, idx2);
          if (dist < 0 || dist > bestDist) continue;
          bestIdx2 = idx2, bestDist = dist;
        }
        if (bestIdx2 >= 0) {
          if (tri_fill(G, idx1, bestIdx2, (float)bestDist)) ++nmatches;
          if (check_orientation) {
            float rot = A.keys[idx1].angle - B.keys[bestIdx2].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)std::round(rot * factor);
            if (bin == kHistoLength) bin = 0;
            rotHist[bin].push_back(idx1);
          }
        }
      }
      n1++, n2++;
    } else if (A.node_id[n1] < B.node_id[n2]) {
      while (n1 < A.n_nodes && A.node_id[n1] < B.node_id[n2]) n1++;  // lower_bound
    } else {
      while (n2 < B.n_nodes && B.node_id[n2] < A.node_id[n1]) n2++;
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, kHistoLength, ind1, ind2, ind3);
    for (int i = 0; i < kHistoLength; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx1 : rotHist[i]) {
        auto it = G.map.find(std::make_pair(0, idx1));
        if (it == G.map.end()) continue;
        G.good[it->second] = false;
        nmatches--;
      }
    }
  }
  pairs.clear();
  for (size_t g = 0; g < G.good.size(); g++) {
    const int cnt = (G.idx[0][g] != -1) + (G.idx[1][g] != -1);
    if (cnt < 2) G.good[g] = false;
    if (!G.good[g]) continue;
    pairs.emplace_back(G.idx[0][g], G.idx[1][g]);
  }
  return nmatches;
}

}  // namespace vo

extern "C" void vo_search_for_triangulation(const vieo_tri_keyframe* kf1, const vieo_tri_keyframe* kf2s, int n_kf2,
                                            int only_stereo, int check_orientation, int32_t pair_capacity,
                                            int32_t* pairs, int32_t* n_pairs, int32_t* n_matches) {
  for (int p = 0; p < n_kf2; p++) {
    std::vector<std::pair<int, int>> v;
    n_matches[p] = vo::search_for_triangulation(*kf1, kf2s[p], only_stereo != 0, check_orientation != 0, v);
    n_pairs[p] = (int32_t)v.size();
    for (size_t i = 0; i < v.size() && (int)i < pair_capacity; i++)
      pairs[((size_t)p * pair_capacity + i) * 2] = v[i].first, pairs[((size_t)p * pair_capacity + i) * 2 + 1] = v[i].second;
  }
}

// test hook: the pure gates of one (key1, key2) pair; -1 or the Hamming distance
extern "C" int vo_tri_gates(const vieo_tri_keyframe* kf1, const vieo_tri_keyframe* kf2, int idx1, int idx2, float* epipole) {
  vo::TriPair P;
  vo::tri_pair_setup(*kf1, *kf2, P);
  if (epipole) epipole[0] = P.ex, epipole[1] = P.ey;
  return vo::tri_gates(*kf1, *kf2, P, idx1, idx2);
}
