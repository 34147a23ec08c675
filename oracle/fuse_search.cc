// ORACLE -- TEST INFRASTRUCTURE ONLY (parity unpinned: no golden vectors in the reference; OpenCV / Eigen /
// Sophus absent, so the reference cannot be built here).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use this file; the product never links or calls it.
//
// Sequential restatement of the search of ORBmatcher::SearchByProjectionBase (reference src/ORBmatcher.cc:26-193)
// with FrameBase::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea / IsInImage (src/FrameBase.cpp:95-174) and
// MapPoint::PredictScale (src/MapPoint.cc:491-509; logf taken as the correctly rounded double logarithm, see
// mappoint.cc).  Float arithmetic as in the reference.
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/vieo_hot.h"
#include "cam_models.hpp"

extern "C" int vo_descriptor_distance(const uint8_t* a, const uint8_t* b);

extern "C" void vo_fuse_search(const vieo_fuse_frame* FF, const vieo_keypoint* const* keys, const float* const* uright,
                               const uint8_t* const* desc, const int32_t* n_keys, const vieo_fuse_point* pts, int n,
                               int32_t* best_idx, int32_t* best_dist) {
  using namespace vo;
  const vieo_frustum_frame& F = FF->base;
  const int GC = 64, GR = 48;  // FrameBase.h:224-225
  const int nc = F.n_cams;
  OCam cams[4];
  std::vector<std::vector<std::vector<int>>> grid(nc, std::vector<std::vector<int>>(GC * GR));
  float winv[4], hinv[4];
  for (int c = 0; c < nc; ++c) {
    const vieo_camera& s = F.cams[c];
    cams[c].model = s.model, cams[c].num_k = s.model == VIEO_CAM_RADTAN ? s.num_k : 0;
    cams[c].fx = s.fx, cams[c].fy = s.fy, cams[c].cx = s.cx, cams[c].cy = s.cy;
    for (int q = 0; q < 8; ++q) cams[c].dist[q] = s.dist[q];
    const float* b = F.bounds[c];
    winv[c] = (float)GC / (b[1] - b[0]), hinv[c] = (float)GR / (b[3] - b[2]);
    for (int i = 0; i < n_keys[c]; ++i) {  // AssignFeaturesToGrid
      const int px = (int)std::round((keys[c][i].x - b[0]) * winv[c]), py = (int)std::round((keys[c][i].y - b[2]) * hinv[c]);
      if (px < 0 || px >= GC || py < 0 || py >= GR) continue;
      grid[c][px * GR + py].push_back(i);
    }
  }
  const float* R = F.Rcrw;
  for (int m = 0; m < n; ++m) {
    const vieo_fuse_point& P = pts[m];
    for (int c = 0; c < nc; ++c) best_idx[m * nc + c] = -1, best_dist[m * nc + c] = INT_MAX;
    if (P.skip_mask & (1u << 31)) continue;
    float Pcr[3];
    for (int r = 0; r < 3; ++r) Pcr[r] = (R[r * 3] * P.Xw[0] + R[r * 3 + 1] * P.Xw[1] + R[r * 3 + 2] * P.Xw[2]) + F.tcrw[r];
    for (int cami = 0; cami < nc; ++cami) {
      if (P.skip_mask & (1 << cami)) continue;
      const float* Tc = F.Tcr[cami];
      float Pc[3], twc[3];
      for (int r = 0; r < 3; ++r)
        Pc[r] = (Tc[r * 4] * Pcr[0] + Tc[r * 4 + 1] * Pcr[1] + Tc[r * 4 + 2] * Pcr[2]) + Tc[r * 4 + 3];
      const float* t = F.trc[cami];
      for (int r = 0; r < 3; ++r) twc[r] = F.Ow[r] + (R[r] * t[0] + R[3 + r] * t[1] + R[6 + r] * t[2]);
      if (Pc[2] <= 0.0) continue;
      const float invz = 1 / Pc[2];
      float u, v;
      if (!F.use_distort) {
        const float p0 = Pc[0] * invz, p1 = Pc[1] * invz;
        u = (cams[cami].fx * p0 + 0.f * p1) + cams[cami].cx * 1.f;
        v = (0.f * p0 + cams[cami].fy * p1) + cams[cami].cy * 1.f;
      } else {
        const double Pd[3] = {Pc[0], Pc[1], Pc[2]};
        float uv[2];
        ocam_project(cams[cami], Pd, uv, nullptr);
        u = uv[0], v = uv[1];
      }
      const float* b = F.bounds[cami];
      if (!(u >= b[0] && u < b[1] && v >= b[2] && v < b[3])) continue;  // IsInImage
      const float PO[3] = {P.Xw[0] - twc[0], P.Xw[1] - twc[1], P.Xw[2] - twc[2]};
      const float dist3D = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
      const float maxDistance = 1.2f * P.max_distance, minDistance = 0.8f * P.min_distance;
      if (dist3D < minDistance || dist3D > maxDistance) continue;
      if (FF->check_viewing_angle) {
        // PO.dot(Pn) < 0.5 * dist3D: float dot against the double product 0.5 * dist3D
        if ((double)(PO[0] * P.normal[0] + PO[1] * P.normal[1] + PO[2] * P.normal[2]) < 0.5 * dist3D) continue;
      }
      const float ratio = P.max_distance / dist3D;
      int lvl = (int)std::ceil((float)std::log((double)ratio) / F.log_scale_factor);
      if (lvl < 0)
        lvl = 0;
      else if (lvl >= F.n_levels)
        lvl = F.n_levels - 1;
      const float radius = FF->th_radius * FF->scale_factors[lvl];
      // GetFeaturesInArea(cami, u, v, radius)
      const int min_cellx = std::max(0, (int)std::floor((u - b[0] - radius) * winv[cami]));
      if (min_cellx >= GC) continue;
      const int max_cellx = std::min(GC - 1, (int)std::ceil((u - b[0] + radius) * winv[cami]));
      if (max_cellx < 0) continue;
      const int min_celly = std::max(0, (int)std::floor((v - b[2] - radius) * hinv[cami]));
      if (min_celly >= GR) continue;
      const int max_celly = std::min(GR - 1, (int)std::ceil((v - b[2] + radius) * hinv[cami]));
      if (max_celly < 0) continue;
      int bestDist = INT_MAX, bestIdx = -1;
      for (int ix = min_cellx; ix <= max_cellx; ++ix)
        for (int iy = min_celly; iy <= max_celly; ++iy)
          for (int idx : grid[cami][ix * GR + iy]) {
            const vieo_keypoint& kp = keys[cami][idx];
            if (!(std::fabs(kp.x - u) < radius && std::fabs(kp.y - v) < radius)) continue;
            const int kpLevel = kp.octave;
            if (kpLevel < lvl - 1 || kpLevel > lvl) continue;
            if (FF->use_bf) {
              const float kpr = uright[cami] ? uright[cami][idx] : -1.f;
              const float ex = u - kp.x, ey = v - kp.y;
              if (kpr >= 0) {
                const float ur = u - F.bf * invz;
                const float er = ur - kpr;
                const float e2 = ex * ex + ey * ey + er * er;
                if ((double)(e2 * FF->inv_level_sigma2[kpLevel]) > 7.8) continue;
              } else {
                const float e2 = ex * ex + ey * ey;
                if ((double)(e2 * FF->inv_level_sigma2[kpLevel]) > 5.99) continue;
              }
            }
            const int d = vo_descriptor_distance(P.desc, desc[cami] + (size_t)idx * 32);
            if (d < bestDist) bestDist = d, bestIdx = idx;
          }
      best_idx[m * nc + cami] = bestIdx, best_dist[m * nc + cami] = bestDist;
    }
  }
}
