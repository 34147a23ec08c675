// ORACLE -- TEST INFRASTRUCTURE ONLY (parity unpinned: no golden vectors in the reference; OpenCV / Eigen /
// Sophus absent, so the reference cannot be built here).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use this file; the product never links or calls it.
//
// Sequential restatement of Frame::ComputeStereoFishEyeMatches (reference src/Frame.cc:613-779) and of what
// it calls in common/camera_models: GeometricCamera::FillMatchesFromPair (camera_base.h:408-574, compiled
// with USE_STRATEGY_MIN_DIST, common/config.h:12), TriangulateMatches (:199-285), Triangulate (:576-608),
// {Pinhole,Radtan,KB8}Camera::UnProject (camera_pinhole.h:108-125, camera_radtan.h:132-178,
// camera_kb8.h:159-195,278-312; kUnProject2Plane, 10 iterations, precision 1e-8: camera_base.h:121-123).
//
// Third-party arithmetic restated: Eigen::JacobiSVD's right singular vector of the smallest singular value
// (camera_base.h:599-600) is computed by one-sided (Hestenes) Jacobi rotations on the columns of A in double;
// Eigen 3.3.7 uses a QR-preconditioned two-sided Jacobi.  Both are backward stable; the null vector agrees to
// rounding, and the ratio x/w removes the sign ambiguity.  Sophus::SE3 is used through its 3x4 matrices.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

#include "../include/vieo_hot.h"
#include "cam_models.hpp"

extern "C" void vo_knn2_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist);

namespace vo {

// right singular vector of the smallest singular value of A (m x 4, row-major), one-sided Jacobi
static void null_vector4(const double* A_in, int m, double* x4) {
  double A[8 * 4], V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  memcpy(A, A_in, sizeof(double) * m * 4);
  for (int sweep = 0; sweep < 40; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        double a = 0, b = 0, g = 0;
        for (int r = 0; r < m; ++r) a += A[r * 4 + p] * A[r * 4 + p], b += A[r * 4 + q] * A[r * 4 + q], g += A[r * 4 + p] * A[r * 4 + q];
        if (g == 0 || std::fabs(g) <= 1e-15 * std::sqrt(a * b)) continue;
        rotated = true;
        const double zeta = (b - a) / (2 * g);
        const double t = (zeta >= 0 ? 1. : -1.) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
        const double cs = 1 / std::sqrt(1 + t * t), sn = cs * t;
        for (int r = 0; r < m; ++r) {
          const double u = A[r * 4 + p], v = A[r * 4 + q];
          A[r * 4 + p] = cs * u - sn * v, A[r * 4 + q] = sn * u + cs * v;
        }
        for (int r = 0; r < 4; ++r) {
          const double u = V[r * 4 + p], v = V[r * 4 + q];
          V[r * 4 + p] = cs * u - sn * v, V[r * 4 + q] = sn * u + cs * v;
        }
      }
    if (!rotated) break;
  }
  int best = 0;
  double nb = INFINITY;
  for (int c = 0; c < 4; ++c) {
    double n = 0;
    for (int r = 0; r < m; ++r) n += A[r * 4 + c] * A[r * 4 + c];
    if (n < nb) nb = n, best = c;
  }
  for (int r = 0; r < 4; ++r) x4[r] = V[r * 4 + best];
}

struct Rig {
  int n = 0;
  OCam cam[4];
  double Rrc[4][9], Tcw[4][12], Tcr[4][12];  // Twi rotation; Twi.inverse() 3x4 (double inverse of Trc); GetTcr()
};

// GeometricCamera::TriangulateMatches for the cameras `ci[0..n)`; depths -> czs, returns false for "empty"
static bool triangulate_matches(const Rig& R, const int* ci, int n, const float (*kp)[2], const float* sig,
                                float th_cos, double* p3d, float* czs) {
  double nP[4][3];
  for (int i = 0; i < n; ++i) ocam_unproject(R.cam[ci[i]], kp[i], nP[i]);
  if (th_cos < 1.) {
    bool bret = true;
    for (int i = 0; i < n - 1 && bret; ++i)
      for (int j = i + 1; j < n; ++j) {
        const double* Ri = R.Rrc[ci[i]];
        const double* Rj = R.Rrc[ci[j]];
        double w[3], v[3];
        for (int r = 0; r < 3; ++r) w[r] = Rj[r * 3] * nP[j][0] + Rj[r * 3 + 1] * nP[j][1] + Rj[r * 3 + 2] * nP[j][2];
        for (int r = 0; r < 3; ++r) v[r] = Ri[r] * w[0] + Ri[3 + r] * w[1] + Ri[6 + r] * w[2];  // Ri^T w
        const double dot = nP[i][0] * v[0] + nP[i][1] * v[1] + nP[i][2] * v[2];
        const double ni = std::sqrt(nP[i][0] * nP[i][0] + nP[i][1] * nP[i][1] + nP[i][2] * nP[i][2]);
        const double nj = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const float cosr = (float)(dot / (ni * nj));
        if (cosr <= th_cos) {
          bret = false;
          break;
        }
      }
    if (bret) return false;
  }
  double A[8 * 4];
  for (int i = 0; i < n; ++i) {
    const double* T = R.Tcw[ci[i]];
    for (int c = 0; c < 4; ++c) {
      A[(2 * i) * 4 + c] = nP[i][0] * T[8 + c] - T[c];
      A[(2 * i + 1) * 4 + c] = nP[i][1] * T[8 + c] - T[4 + c];
    }
  }
  double x4[4];
  null_vector4(A, 2 * n, x4);
  if (!x4[3]) return false;
  const double X[3] = {x4[0] / x4[3], x4[1] / x4[3], x4[2] / x4[3]};
  for (int i = 0; i < n; ++i) {
    const double* T = R.Tcw[ci[i]];
    czs[i] = (float)(T[8] * X[0] + T[9] * X[1] + T[10] * X[2] + T[11]);
    if (czs[i] <= 0) return false;
    double Pc[3];
    for (int r = 0; r < 3; ++r) Pc[r] = (T[r * 4] * X[0] + T[r * 4 + 1] * X[1] + T[r * 4 + 2] * X[2]) + T[r * 4 + 3];
    float uv[2];
    ocam_project(R.cam[ci[i]], Pc, uv, nullptr);
    const float e0 = uv[0] - kp[i][0], e1 = uv[1] - kp[i][1];
    const float thresh_chi2 = 5.991f;
    if (e0 * e0 + e1 * e1 > thresh_chi2 * sig[i]) return false;
  }
  p3d[0] = X[0], p3d[1] = X[1], p3d[2] = X[2];
  return true;
}

typedef std::pair<size_t, size_t> CamIdx;
// branch coverage of the bookkeeping, read by the tests: new group, extension of a group, replacement of a
// member (MIN_DIST), contradiction kept / swapped
static long g_branch[5] = {0, 0, 0, 0, 0};

struct Groups {
  std::vector<std::vector<size_t>> idxs;
  std::vector<bool> good;
  std::vector<double> p3d;  // 3 per group
  std::vector<std::vector<float>> lastdists;
  std::map<CamIdx, size_t> map;
  void clear() { idxs.clear(), good.clear(), p3d.clear(), lastdists.clear(), map.clear(); }
};

static bool fill_matches_from_pair(const Rig& R, size_t n_tot, size_t cami, size_t idxi, size_t camj, size_t idxj,
                                   float dist, Groups& G, float th_cos, const float (*kp)[2], const float* sig,
                                   int* pcount) {
  const size_t NONE = (size_t)-1;
  const CamIdx ki(cami, idxi), kj(camj, idxj);
  auto iteri = G.map.find(ki), iterj = G.map.find(kj);
  if (iteri == G.map.end() && iterj != G.map.end()) iteri = iterj;
  uint8_t checkdepth[2] = {0, 0};
  size_t ididxs = 0;
  uint8_t contradict = 0;
  if (iteri != G.map.end()) {
    ididxs = iteri->second;
    contradict = (iterj != G.map.end() && iterj->second != ididxs) ? 2 : 0;
    std::vector<size_t> idxs = G.idxs[ididxs];
    if (contradict) {
      const auto& idxsj = G.idxs[iterj->second];
      float dists_sum[2] = {0, 0};
      size_t count_num[2] = {0, 0};
      for (size_t t = 0; t < n_tot; ++t) {
        if (NONE != idxs[t]) dists_sum[0] += G.lastdists[ididxs][t], ++count_num[0];
        if (NONE != idxsj[t]) dists_sum[1] += G.lastdists[iterj->second][t], ++count_num[1];
      }
      if (dists_sum[1] * count_num[0] < dists_sum[0] * count_num[1]) {
        idxs = idxsj;
        ididxs = iterj->second;
        contradict = 1;
      }
    }
    if (NONE == idxs[cami] || (idxi != idxs[cami] && G.lastdists[ididxs][cami] > dist)) checkdepth[0] = 2;
    if (NONE == idxs[camj] || (idxj != idxs[camj] && G.lastdists[ididxs][camj] > dist)) checkdepth[1] = 2;
  } else
    checkdepth[0] = checkdepth[1] = 1;
  if (pcount) ++*pcount;
  if (!(checkdepth[0] || checkdepth[1])) return false;
  double p3D[3];
  float depths[2];
  const int ci[2] = {(int)cami, (int)camj};
  if (!triangulate_matches(R, ci, 2, kp, sig, th_cos, p3D, depths)) return false;
  if (!(depths[0] > 0.0001f && depths[1] > 0.0001f)) return false;
  if (1 == checkdepth[0])
    ++g_branch[0];
  else {
    const auto& cur = G.idxs[ididxs];
    if ((2 == checkdepth[0] && NONE != cur[cami]) || (2 == checkdepth[1] && NONE != cur[camj]))
      ++g_branch[2];
    else
      ++g_branch[1];
    if (contradict) ++g_branch[2 + contradict];
  }
  if (1 == checkdepth[0]) {
    std::vector<size_t> idxs(n_tot, NONE);
    idxs[cami] = idxi, idxs[camj] = idxj;
    ididxs = G.idxs.size();
    G.map.emplace(ki, ididxs);
    G.map.emplace(kj, ididxs);
    G.idxs.push_back(idxs);
    G.p3d.resize(G.idxs.size() * 3);
    G.good.push_back(true);
    std::vector<float> d(n_tot, INFINITY);
    d[cami] = dist, d[camj] = dist;
    G.lastdists.push_back(d);
  } else if (2 == checkdepth[0] || 2 == checkdepth[1]) {
    if (contradict) {
      const size_t idc = 1 == contradict ? iteri->second : iterj->second;
      auto& idxs = G.idxs[idc];
      if (idxi == idxs[cami]) {
        G.map.erase(ki);
        G.lastdists[idc][cami] = INFINITY;
        idxs[cami] = NONE;
      }
      if (idxj == idxs[camj]) {
        G.map.erase(kj);
        G.lastdists[idc][camj] = INFINITY;
        idxs[camj] = NONE;
      }
    }
    auto& idxs = G.idxs[ididxs];
    if (2 == checkdepth[0]) {
      if (idxi != idxs[cami]) {
        if (NONE != idxs[cami]) G.map.erase(CamIdx(cami, idxs[cami]));
        G.map.emplace(ki, ididxs);
        idxs[cami] = idxi;
      }
      G.lastdists[ididxs][cami] = dist;
    } else if (G.lastdists[ididxs][cami] > dist)
      G.lastdists[ididxs][cami] = dist;
    if (2 == checkdepth[1]) {
      if (idxj != idxs[camj]) {
        if (NONE != idxs[camj]) G.map.erase(CamIdx(camj, idxs[camj]));
        G.map.emplace(kj, ididxs);
        idxs[camj] = idxj;
      }
      G.lastdists[ididxs][camj] = dist;
    } else if (G.lastdists[ididxs][camj] > dist)
      G.lastdists[ididxs][camj] = dist;
  }
  memcpy(&G.p3d[ididxs * 3], p3D, 24);
  return true;
}

}  // namespace vo

extern "C" {

// returns 0, or -1 when more than group_capacity groups form (outputs then undefined)
int vo_stereo_fisheye_match(const vieo_fisheye_params* P, const vieo_keypoint* const* keys,
                            const uint8_t* const* desc, const int32_t* n_keys, const int32_t* num_mono,
                            int32_t group_capacity, float* depth, int32_t* key_group, int32_t* group_idx,
                            uint8_t* group_good, double* group_p3d, int32_t* n_groups, int32_t* n_matches) {
  using namespace vo;
  const size_t n_cams = P->n_cams;
  Rig R;
  R.n = (int)n_cams;
  for (size_t i = 0; i < n_cams; ++i) {
    const vieo_camera& s = P->cams[i];
    OCam& c = R.cam[i];
    c.model = s.model, c.num_k = s.model == VIEO_CAM_RADTAN ? s.num_k : 0;
    c.fx = s.fx, c.fy = s.fy, c.cx = s.cx, c.cy = s.cy, c.bf = 0;
    for (int q = 0; q < 8; ++q) c.dist[q] = s.dist[q];
    const double* T = P->Trc + 12 * i;
    for (int r = 0; r < 3; ++r)
      for (int q = 0; q < 3; ++q) R.Rrc[i][r * 3 + q] = T[r * 4 + q];
    // Twi.inverse(): R^T, -R^T t
    for (int r = 0; r < 3; ++r) {
      for (int q = 0; q < 3; ++q) R.Tcw[i][r * 4 + q] = T[q * 4 + r];
      R.Tcw[i][r * 4 + 3] = -(T[0 * 4 + r] * T[3] + T[1 * 4 + r] * T[7] + T[2 * 4 + r] * T[11]);
    }
    memcpy(R.Tcr[i], P->Tcr + 12 * i, 96);
  }
  // brute force between the key points of all image pairs (Frame.cc:618-628)
  std::vector<std::vector<int32_t>> kidx, kdist;
  std::vector<int> knq;
  for (size_t i = 0; i + 1 < n_cams; ++i)
    for (size_t j = i + 1; j < n_cams; ++j) {
      kidx.emplace_back(), kdist.emplace_back();
      knq.push_back(0);
      if (num_mono[i] >= n_keys[i] || num_mono[j] >= n_keys[j]) continue;
      const int nq = n_keys[i] - num_mono[i], nt = n_keys[j] - num_mono[j];
      kidx.back().resize((size_t)nq * 2), kdist.back().resize((size_t)nq * 2);
      knq.back() = nq;
      vo_knn2_hamming(desc[i] + (size_t)num_mono[i] * 32, nq, desc[j] + (size_t)num_mono[j] * 32, nt,
                      kidx.back().data(), kdist.back().data());
    }
  int nMatches = 0, descMatches = 0;
  const float f_bar = (P->cams[0].fx + P->cams[0].fy) / 2.;
  double th[2] = {0.9998, 1. - 1e-6};
  if (P->th_far_pts > 0)
    for (int i = 0; i < 2; ++i) th[i] = std::min(1. - std::pow(P->bf / f_bar / P->th_far_pts, 2) / 2., th[i]);
  Groups G;
  const int tries = th[1] == th[0] ? 1 : 2;
  for (int k = 0; k < tries; ++k) {
    G.clear();
    descMatches = 0;
    size_t idm = 0;
    for (size_t i = 0; i + 1 < n_cams; ++i)
      for (size_t j = i + 1; j < n_cams; ++j, ++idm)
        for (int q = 0; q < knq[idm]; ++q) {
          const int32_t* id = &kidx[idm][(size_t)q * 2];
          const int32_t* dd = &kdist[idm][(size_t)q * 2];
          if (id[0] < 0 || id[1] < 0) continue;  // (*it).size() >= 2
          const float d0 = (float)dd[0], d1 = (float)dd[1];
          const int thOrbDist = (100 + 50) / 2;
          if (!(d0 < d1 * 0.7 || (d0 < thOrbDist && d0 < d1 * 0.9))) continue;
          const size_t idxi = q + num_mono[i], idxj = id[0] + num_mono[j];
          const vieo_keypoint &ka = keys[i][idxi], &kb = keys[j][idxj];
          const float sig[2] = {P->level_sigma2[ka.octave], P->level_sigma2[kb.octave]};
          const float kp[2][2] = {{ka.x, ka.y}, {kb.x, kb.y}};
          if (fill_matches_from_pair(R, n_cams, i, idxi, j, idxj, d0, G, (float)th[0], kp, sig, &descMatches)) ++nMatches;
        }
    if (nMatches >= 30) break;
    th[0] = th[1];
  }
  const size_t NONE = (size_t)-1;
  for (size_t g = 0; g < G.idxs.size(); ++g) {
    size_t cnt = 0;
    for (size_t t = 0; t < n_cams; ++t)
      if (NONE != G.idxs[g][t]) ++cnt;
    if (cnt < 2) G.good[g] = false;
  }
  if (n_cams > 2) {
    nMatches = 0;
    for (size_t g = 0; g < G.idxs.size(); ++g) {
      if (!G.good[g]) continue;
      int ci[4], n = 0;
      float sig[4], kp[4][2], czs[4];
      for (size_t k = 0; k < n_cams; ++k)
        if (NONE != G.idxs[g][k]) {
          const vieo_keypoint& kk = keys[k][G.idxs[g][k]];
          ci[n] = (int)k, sig[n] = P->level_sigma2[kk.octave], kp[n][0] = kk.x, kp[n][1] = kk.y;
          ++n;
        }
      double p3D[3];
      bool ok = triangulate_matches(R, ci, n, kp, sig, (float)th[0], p3D, czs);
      for (int t = 0; ok && t < n; ++t)
        if (czs[t] <= 0.0001f) ok = false;
      if (ok) {
        memcpy(&G.p3d[g * 3], p3D, 24);
        nMatches++;
      } else
        G.good[g] = false;
    }
  }
  if ((int)G.idxs.size() > group_capacity) return -1;
  *n_groups = (int)G.idxs.size();
  *n_matches = nMatches;
  for (size_t g = 0; g < G.idxs.size(); ++g) {
    for (size_t t = 0; t < n_cams; ++t) group_idx[g * n_cams + t] = NONE == G.idxs[g][t] ? -1 : (int32_t)G.idxs[g][t];
    group_good[g] = G.good[g];
    memcpy(group_p3d + g * 3, &G.p3d[g * 3], 24);
  }
  size_t n = 0;
  for (size_t i = 0; i < n_cams; ++i)
    for (int k = 0; k < n_keys[i]; ++k, ++n) {
      auto it = G.map.find(CamIdx(i, (size_t)k));
      depth[n] = -1, key_group[n] = -1;
      if (it == G.map.end()) continue;
      key_group[n] = (int32_t)it->second;
      if (G.good[it->second]) {
        const double* T = R.Tcr[i];
        const double* X = &G.p3d[it->second * 3];
        depth[n] = (float)(T[8] * X[0] + T[9] * X[1] + T[10] * X[2] + T[11]);
      }
    }
  return 0;
}

// test hooks
void vo_fisheye_branch_counts(long* out5, int reset) {
  for (int i = 0; i < 5; ++i) out5[i] = vo::g_branch[i];
  if (reset)
    for (int i = 0; i < 5; ++i) vo::g_branch[i] = 0;
}
void vo_cam_unproject(const vieo_camera* cam, const float* uv, double* P3) {
  vo::OCam c;
  c.model = cam->model, c.num_k = cam->model == VIEO_CAM_RADTAN ? cam->num_k : 0;
  c.fx = cam->fx, c.fy = cam->fy, c.cx = cam->cx, c.cy = cam->cy;
  for (int q = 0; q < 8; ++q) c.dist[q] = cam->dist[q];
  vo::ocam_unproject(c, uv, P3);
}
void vo_null_vector4(const double* A, int m, double* x4) { vo::null_vector4(A, m, x4); }

}  // extern "C"
