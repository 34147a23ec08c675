// ORACLE -- TEST INFRASTRUCTURE ONLY (see ocv_prims.hpp header).  PARITY UNPINNED.
//
// CPU restatement of int ORBmatcher::SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo)
// (reference src/ORBmatcher.cc:896-1150), undistorted pinhole key frames and distorted camera rigs, with what it calls:
//   GeometricCamera::epipolarConstrain, fundamental-matrix branch      common/camera_models/camera_base.h:287-406
//     (USE_DIR_EPI_ERR is commented out: :295; Tdata = float, Tcalc = double: common/config.h:23-24)
//   GeometricCamera::FillMatchesFromPair without key points / sigmas   camera_base.h:408-574  (no triangulation,
//     USE_STRATEGY_MIN_DIST bookkeeping only)
//   {Pinhole,Radtan,KB8}Camera::UnProject / Project of the rig's keys      cam_models.hpp
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
#include "cam_models.hpp"

namespace vo {

static const int kThLow = 50, kHistoLength = 30;

struct TriPair {  // per (pKF1, pKF2): what the loops need besides the keys
  bool rig;
  int nc[2];
  OCam cams[2][4];
  double F12[4][4][9];  // per (camera of pKF1, camera of pKF2)
  float ex, ey;
};

static inline float f32(double v) { return (float)v; }

static void tri_cams(const vieo_tri_keyframe& K, OCam* out) {
  vieo_lba_params prm;
  memset(&prm, 0, sizeof(prm));
  prm.fx = K.fx, prm.fy = K.fy, prm.cx = K.cx, prm.cy = K.cy;
  prm.n_cams = K.n_cams, prm.cams = K.cams;
  ocams_from_params(prm, out);
}

// Tr1r2 = (Tcw1 * Twc2).cast<float>(), T12 = Tcr(cam1) * Tr1r2 * Trc(cam2) (float SE3s: the product is taken in
// double on the float values and rounded once), F12 = K1^-T [t12]x R12 K2^-1 in double, the epipole of pKF1's
// reference camera in image 0 of pKF2
static void tri_pair_setup(const vieo_tri_keyframe& A, const vieo_tri_keyframe& B, TriPair& P) {
  P.rig = A.n_cams > 0;
  P.nc[0] = A.n_cams > 0 ? A.n_cams : 1, P.nc[1] = B.n_cams > 0 ? B.n_cams : 1;
  tri_cams(A, P.cams[0]), tri_cams(B, P.cams[1]);
  const double* T1 = A.Tcw;
  const double* T2 = B.Tcw;
  double Rr[9], tr[3];
  for (int i = 0; i < 3; i++) {
    double s = 0;
    for (int j = 0; j < 3; j++) {
      double r = 0;
      for (int k = 0; k < 3; k++) r += T1[i * 4 + k] * T2[j * 4 + k];  // R1 R2^T
      Rr[i * 3 + j] = (double)f32(r);
      s += r * T2[j * 4 + 3];
    }
    tr[i] = (double)f32(T1[i * 4 + 3] - s);  // t1 - R t2, with the unrounded rotation
  }
  auto mul = [](const double* X, const double* Y, double* Z) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Z[i * 3 + j] = X[i * 3] * Y[j] + X[i * 3 + 1] * Y[3 + j] + X[i * 3 + 2] * Y[6 + j];
  };
  for (int c1 = 0; c1 < P.nc[0]; c1++)
    for (int c2 = 0; c2 < P.nc[1]; c2++) {
      double R12[9], t12[3];
      if (!P.rig) {
        memcpy(R12, Rr, 72), memcpy(t12, tr, 24);
      } else {
        const double* Tcr = A.Tcr + 12 * c1;
        const double* Trc = B.Trc + 12 * c2;
        double Ra[9], Rb[9], M[9], v[3];
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) Ra[i * 3 + j] = Tcr[i * 4 + j], Rb[i * 3 + j] = Trc[i * 4 + j];
        mul(Ra, Rr, M);
        mul(M, Rb, R12);
        for (int i = 0; i < 3; i++) v[i] = tr[i] + (Rr[i * 3] * Trc[3] + Rr[i * 3 + 1] * Trc[7] + Rr[i * 3 + 2] * Trc[11]);
        for (int i = 0; i < 3; i++) t12[i] = Tcr[i * 4 + 3] + (Ra[i * 3] * v[0] + Ra[i * 3 + 1] * v[1] + Ra[i * 3 + 2] * v[2]);
        for (int i = 0; i < 9; i++) R12[i] = (double)f32(R12[i]);
        for (int i = 0; i < 3; i++) t12[i] = (double)f32(t12[i]);
      }
      const OCam &k1 = P.cams[0][c1], &k2 = P.cams[1][c2];
      const double fx1 = (float)k1.fx, fy1 = (float)k1.fy, cx1 = (float)k1.cx, cy1 = (float)k1.cy;
      const double fx2 = (float)k2.fx, fy2 = (float)k2.fy, cx2 = (float)k2.cx, cy2 = (float)k2.cy;
      const double K1it[9] = {1 / fx1, 0, 0, 0, 1 / fy1, 0, -cx1 / fx1, -cy1 / fy1, 1};  // (K1^T)^-1
      const double K2i[9] = {1 / fx2, 0, -cx2 / fx2, 0, 1 / fy2, -cy2 / fy2, 0, 0, 1};
      const double H[9] = {0, -t12[2], t12[1], t12[2], 0, -t12[0], -t12[1], t12[0], 0};
      double M1[9], M2[9];
      mul(K1it, H, M1);
      mul(M1, R12, M2);
      mul(M2, K2i, P.F12[c1][c2]);
    }
  // Cw = -R1^T t1 (float cv::Mat), C2 = R2w Cw + t2w, epipole = projection of C2 by camera 0 of pKF2
  float R1f[9], t1f[3], R2f[9], t2f[3], Cw[3], C2[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R1f[i * 3 + j] = f32(T1[i * 4 + j]), R2f[i * 3 + j] = f32(T2[i * 4 + j]);
    t1f[i] = f32(T1[i * 4 + 3]), t2f[i] = f32(T2[i * 4 + 3]);
  }
  for (int i = 0; i < 3; i++)
    Cw[i] = f32(-((double)R1f[i] * t1f[0] + (double)R1f[3 + i] * t1f[1] + (double)R1f[6 + i] * t1f[2]));
  for (int i = 0; i < 3; i++)
    C2[i] = f32((double)R2f[i * 3] * Cw[0] + (double)R2f[i * 3 + 1] * Cw[1] + (double)R2f[i * 3 + 2] * Cw[2] + (double)t2f[i]);
  if (!P.rig) {
    const float invz = 1.0f / C2[2];
    const float xn = C2[0] * invz, yn = C2[1] * invz;
    P.ex = (B.fx * xn + 0.0f * yn) + B.cx;
    P.ey = (0.0f * xn + B.fy * yn) + B.cy;
  } else {
    const double C2d[3] = {C2[0], C2[1], C2[2]};
    float uv[2];
    ocam_project(P.cams[1][0], C2d, uv, nullptr);
    P.ex = uv[0], P.ey = uv[1];
  }
}

static inline int hamming256(const uint8_t* a, const uint8_t* b) {
  int d = 0;
  for (int i = 0; i < 32; i++) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
  return d;
}

// the image point the epipolar test uses: the key itself, or K * UnProject(key) of a distorted key (bkp_distort)
static bool tri_key_point(const TriPair& P, int side, const vieo_tri_keyframe& K, int idx, double* pt) {
  const vieo_keypoint& kp = K.keys[idx];
  if (!P.rig) {
    pt[0] = kp.x, pt[1] = kp.y;
    return true;
  }
  const OCam& c = P.cams[side][K.key_cam[idx]];
  const float uv[2] = {kp.x, kp.y};
  double X[3];
  ocam_unproject(c, uv, X);
  const double fx = (float)c.fx, fy = (float)c.fy, cx = (float)c.cx, cy = (float)c.cy;
  const double q0 = (fx * X[0] + 0.0 * X[1]) + cx * X[2], q1 = (0.0 * X[0] + fy * X[1]) + cy * X[2];
  const double q2 = (0.0 * X[0] + 0.0 * X[1]) + 1.0 * X[2];
  if (!(std::isfinite(q0) && std::isfinite(q1) && std::isfinite(q2))) return false;
  const double invz = 1. / q2;
  pt[0] = q0 * invz, pt[1] = q1 * invz;
  return true;
}

// every gate of the inner loop that does not depend on earlier matches; returns the Hamming distance or -1
static int tri_gates(const vieo_tri_keyframe& A, const vieo_tri_keyframe& B, const TriPair& P, int idx1, int idx2) {
  const int dist = hamming256(A.descriptors + 32 * (size_t)idx1, B.descriptors + 32 * (size_t)idx2);
  if (dist > kThLow) return -1;
  const vieo_keypoint& kp2 = B.keys[idx2];
  const bool st1 = A.uright[idx1] >= 0, st2 = B.uright[idx2] >= 0;
  if (!st1 && !st2) {
    const float distex = P.ex - kp2.x, distey = P.ey - kp2.y;
    if (distex * distex + distey * distey < 100 * B.scale_factor[kp2.octave]) return -1;
  }
  double p1[2], p2[2];
  if (!tri_key_point(P, 0, A, idx1, p1)) return -1;
  if (!tri_key_point(P, 1, B, idx2, p2)) return -1;
  const double* F = P.F12[P.rig ? A.key_cam[idx1] : 0][P.rig ? B.key_cam[idx2] : 0];
  const float a = f32(p1[0] * F[0] + p1[1] * F[3] + F[6]);
  const float b = f32(p1[0] * F[1] + p1[1] * F[4] + F[7]);
  const float c = f32(p1[0] * F[2] + p1[1] * F[5] + F[8]);
  const float num = f32((double)a * p2[0] + (double)b * p2[1] + (double)c);
  const float den = a * a + b * b;
  if (den == 0) return -1;
  const float dsqr = num * num / den;
  return dsqr < 3.84f * B.level_sigma2[kp2.octave] ? dist : -1;
}

struct TriGroups {  // cameras of pKF1 first, then those of pKF2
  int nc = 2;
  std::vector<std::vector<int>> idx;
  std::vector<std::vector<float>> last;
  std::vector<bool> good;
  std::map<std::pair<int, int>, int> map;
};

// FillMatchesFromPair with bdepth_ok == true (no key points / sigmas passed: camera_base.h:497)
static bool tri_fill(TriGroups& G, int cami, int idxi, int camj, int idxj, float dist) {
  const std::pair<int, int> ki(cami, idxi), kj(camj, idxj);
  auto iteri = G.map.find(ki), iterj = G.map.find(kj);
  if (iteri == G.map.end() && iterj != G.map.end()) iteri = iterj;
  int check[2] = {0, 0}, contradict = 0, g = -1;
  const int cam[2] = {cami, camj}, id[2] = {idxi, idxj};
  if (iteri != G.map.end()) {
    g = iteri->second;
    contradict = (iterj != G.map.end() && iterj->second != g) ? 2 : 0;
    if (contradict) {
      const int gj = iterj->second;
      float sum[2] = {0, 0};
      int cnt[2] = {0, 0};
      for (int t = 0; t < G.nc; t++) {
        if (G.idx[g][t] != -1) sum[0] += G.last[g][t], ++cnt[0];
        if (G.idx[gj][t] != -1) sum[1] += G.last[gj][t], ++cnt[1];
      }
      if (sum[1] * cnt[0] < sum[0] * cnt[1]) g = gj, contradict = 1;
    }
    for (int t = 0; t < 2; t++)
      if (G.idx[g][cam[t]] == -1 || (id[t] != G.idx[g][cam[t]] && G.last[g][cam[t]] > dist)) check[t] = 2;
  } else
    check[0] = check[1] = 1;
  if (!(check[0] || check[1])) return false;
  if (check[0] == 1) {
    g = (int)G.good.size();
    G.map.emplace(ki, g), G.map.emplace(kj, g);
    G.idx.emplace_back(G.nc, -1), G.last.emplace_back(G.nc, INFINITY);
    G.idx[g][cami] = idxi, G.idx[g][camj] = idxj;
    G.last[g][cami] = dist, G.last[g][camj] = dist;
    G.good.push_back(true);
    return true;
  }
  if (contradict) {
    const int gc = contradict == 1 ? iteri->second : iterj->second;
    if (idxi == G.idx[gc][cami]) G.map.erase(ki), G.last[gc][cami] = INFINITY, G.idx[gc][cami] = -1;
    if (idxj == G.idx[gc][camj]) G.map.erase(kj), G.last[gc][camj] = INFINITY, G.idx[gc][camj] = -1;
  }
  for (int t = 0; t < 2; t++) {
    if (check[t] == 2) {
      if (id[t] != G.idx[g][cam[t]]) {
        if (G.idx[g][cam[t]] != -1) G.map.erase(std::make_pair(cam[t], G.idx[g][cam[t]]));
        G.map.emplace(std::make_pair(cam[t], id[t]), g);
        G.idx[g][cam[t]] = id[t];
      }
      G.last[g][cam[t]] = dist;
    } else if (G.last[g][cam[t]] > dist)
      G.last[g][cam[t]] = dist;
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
                                    bool check_orientation, std::vector<std::vector<int>>& rows) {
  TriPair P;
  tri_pair_setup(A, B, P);
  int nmatches = 0;
  std::vector<int> rotHist[kHistoLength];
  const float factor = 1.0f / kHistoLength;
  TriGroups G;
  const int nc1 = P.nc[0];
  G.nc = P.nc[0] + P.nc[1];
  auto cam_of = [&](const vieo_tri_keyframe& K, int idx) { return P.rig ? (int)K.key_cam[idx] : 0; };
  int n1 = 0, n2 = 0;
  while (n1 < A.n_nodes && n2 < B.n_nodes) {
    if (A.node_id[n1] == B.node_id[n2]) {
      for (int i1 = A.node_first[n1]; i1 < A.node_first[n1 + 1]; i1++) {
        const int idx1 = A.node_feat[i1];
        if (A.has_mappoint[idx1]) continue;
        const bool st1 = A.uright[idx1] >= 0;
        if (only_stereo && !st1) continue;
        std::vector<int> vbestDist(1, kThLow), vbestIdx2(1, -1);  // per image of pKF2 (MATCH_KNN_IN_EACH_IMG)
        const int cam1 = cam_of(A, idx1);
        for (int i2 = B.node_first[n2]; i2 < B.node_first[n2 + 1]; i2++) {
          const int idx2 = B.node_feat[i2];
          if (B.has_mappoint[idx2]) continue;
          const int img = cam_of(B, idx2), cam2 = img + nc1;
          auto it = G.map.find(std::make_pair(cam2, idx2));
          if (it != G.map.end() && G.idx[it->second][cam1] != -1) continue;  // injective matches
          if (only_stereo && !(B.uright[idx2] >= 0)) continue;
          const int dist = tri_gates(A, B, P, idx1, idx2);  // (the Hamming distance is taken before the resize)
          if (P.rig && (int)vbestDist.size() <= img) vbestDist.resize(img + 1, kThLow), vbestIdx2.resize(img + 1, -1);
          if (dist < 0 || dist > vbestDist[img]) continue;
          vbestIdx2[img] = idx2, vbestDist[img] = dist;
        }
        for (int img = 0; img < (int)vbestDist.size(); img++) {
          if (vbestIdx2[img] < 0) continue;
          const int idx2 = vbestIdx2[img];
          if (tri_fill(G, cam1, idx1, cam_of(B, idx2) + nc1, idx2, (float)vbestDist[img])) ++nmatches;
          if (check_orientation) {
            float rot = A.keys[idx1].angle - B.keys[idx2].angle;
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
        auto it = G.map.find(std::make_pair(cam_of(A, idx1), idx1));
        if (it == G.map.end()) continue;
        G.good[it->second] = false;
        nmatches--;
      }
    }
  }
  rows.clear();
  for (size_t g = 0; g < G.good.size(); g++) {
    int cnt = 0;
    for (int t = 0; t < G.nc; t++) cnt += G.idx[g][t] != -1;
    if (cnt < 2) G.good[g] = false;
    if (!G.good[g]) continue;
    rows.push_back(G.idx[g]);
  }
  return nmatches;
}

}  // namespace vo

extern "C" int vo_search_for_triangulation(const vieo_tri_keyframe* kf1, const vieo_tri_keyframe* kf2s, int n_kf2,
                                           int only_stereo, int check_orientation, int32_t pair_capacity,
                                           int32_t pair_stride, int32_t* pairs, int32_t* n_pairs, int32_t* n_matches) {
  for (int p = 0; p < n_kf2; p++) {
    std::vector<std::vector<int>> v;
    n_matches[p] = vo::search_for_triangulation(*kf1, kf2s[p], only_stereo != 0, check_orientation != 0, v);
    n_pairs[p] = (int32_t)v.size();
    for (size_t i = 0; i < v.size() && (int)i < pair_capacity; i++) {
      int32_t* row = pairs + ((size_t)p * pair_capacity + i) * pair_stride;
      for (int c = 0; c < pair_stride; c++) row[c] = c < (int)v[i].size() ? v[i][c] : -1;
    }
  }
  return 0;
}

// test hook: the pure gates of one (key1, key2) pair; -1 or the Hamming distance
extern "C" int vo_tri_gates(const vieo_tri_keyframe* kf1, const vieo_tri_keyframe* kf2, int idx1, int idx2, float* epipole) {
  vo::TriPair P;
  vo::tri_pair_setup(*kf1, *kf2, P);
  if (epipole) epipole[0] = P.ex, epipole[1] = P.ey;
  return vo::tri_gates(*kf1, *kf2, P, idx1, idx2);
}
