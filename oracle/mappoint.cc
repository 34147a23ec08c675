// ORACLE -- TEST INFRASTRUCTURE ONLY (parity unpinned: no golden vectors in the reference; OpenCV / Eigen /
// Sophus absent, so the reference cannot be built here).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use this file; the product never links or calls it.
//
// Sequential restatement of the map-point steps next to the hot path:
//   Frame::isInFrustum                        reference src/Frame.cc:335-416 (+ MapPoint::PredictScale, MapPoint.cc:491-509)
//   MapPoint::ComputeDistinctiveDescriptors   src/MapPoint.cc:314-378
//   MapPoint::UpdateNormalAndDepth            src/MapPoint.cc:424-480
// Float arithmetic throughout (MapPoint::Tcalc = float, MapPoint.h:29).  Deviation kept in one place:
// PredictScale's logf(ratio) is evaluated as (float)log((double)ratio) -- the correctly rounded value -- so that
// the device (whose logf is not glibc's) takes the same ceil() decisions; glibc's logf (< 0.82 ULP) differs from
// it only when it misrounds, and then only matters if the quotient sits on an integer.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/vieo_hot.h"
#include "cam_models.hpp"

extern "C" int vo_descriptor_distance(const uint8_t* a, const uint8_t* b);

extern "C" {

void vo_is_in_frustum_batch(const vieo_frustum_frame* F, const vieo_frustum_point* pts, int n, vieo_track_info* out) {
  using namespace vo;
  OCam cams[4];
  for (int c = 0; c < F->n_cams; ++c) {
    const vieo_camera& s = F->cams[c];
    cams[c].model = s.model, cams[c].num_k = s.model == VIEO_CAM_RADTAN ? s.num_k : 0;
    cams[c].fx = s.fx, cams[c].fy = s.fy, cams[c].cx = s.cx, cams[c].cy = s.cy;
    for (int q = 0; q < 8; ++q) cams[c].dist[q] = s.dist[q];
  }
  for (int m = 0; m < n; ++m) {
    const vieo_frustum_point& P = pts[m];
    vieo_track_info& T = out[m];
    memset(&T, 0, sizeof(T));
    const float maxDistance = 1.2f * P.max_distance, minDistance = 0.8f * P.min_distance;
    const float* R = F->Rcrw;
    float Pcr[3];
    for (int r = 0; r < 3; ++r) Pcr[r] = (R[r * 3] * P.Xw[0] + R[r * 3 + 1] * P.Xw[1] + R[r * 3 + 2] * P.Xw[2]) + F->tcrw[r];
    float sum_depth = 0;
    for (int cami = 0; cami < F->n_cams; ++cami) {
      const float* Tc = F->Tcr[cami];
      float Pc[3], twc[3];
      for (int r = 0; r < 3; ++r)
        Pc[r] = (Tc[r * 4] * Pcr[0] + Tc[r * 4 + 1] * Pcr[1] + Tc[r * 4 + 2] * Pcr[2]) + Tc[r * 4 + 3];
      const float* t = F->trc[cami];
      for (int r = 0; r < 3; ++r) twc[r] = F->Ow[r] + (R[r] * t[0] + R[3 + r] * t[1] + R[6 + r] * t[2]);  // Rcrw^T trc
      const float PcZ = Pc[2];
      if (PcZ < 0.0f) continue;
      const float invz = 1.0f / PcZ;
      float u, v;
      if (!F->use_distort) {
        const float p0 = Pc[0] * invz, p1 = Pc[1] * invz;
        u = (cams[cami].fx * p0 + 0.f * p1) + cams[cami].cx * 1.f;  // K.cast<float>() * (x/z, y/z, 1)
        v = (0.f * p0 + cams[cami].fy * p1) + cams[cami].cy * 1.f;
      } else {
        const double Pd[3] = {Pc[0], Pc[1], Pc[2]};
        float uv[2];
        ocam_project(cams[cami], Pd, uv, nullptr);
        u = uv[0], v = uv[1];
      }
      const float* b = F->bounds[cami];
      if (u < b[0] || u > b[1]) continue;
      if (v < b[2] || v > b[3]) continue;
      const float PO[3] = {P.Xw[0] - twc[0], P.Xw[1] - twc[1], P.Xw[2] - twc[2]};
      const float dist3D = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
      if (dist3D < minDistance || dist3D > maxDistance) continue;
      const float viewCos = (PO[0] * P.normal[0] + PO[1] * P.normal[1] + PO[2] * P.normal[2]) / dist3D;
      if (viewCos < F->viewing_cos_limit) continue;
      // MapPoint::PredictScale
      const float ratio = P.max_distance / dist3D;
      int nscale = (int)std::ceil((float)std::log((double)ratio) / F->log_scale_factor);
      if (nscale < 0)
        nscale = 0;
      else if (nscale >= F->n_levels)
        nscale = F->n_levels - 1;
      const int k = T.n++;
      T.u[k] = u, T.v[k] = v, T.ur[k] = u - F->bf * invz;
      T.level[k] = nscale, T.viewcos[k] = viewCos, T.cam[k] = cami;
      sum_depth += dist3D;
    }
    T.track_depth = T.n ? sum_depth / T.n : -1.f;
  }
}

void vo_distinctive_descriptors_batch(const uint8_t* desc, const int32_t* first, int n, int32_t* best) {
  for (int p = 0; p < n; ++p) {
    const int N = first[p + 1] - first[p];
    best[p] = -1;
    if (N <= 0) continue;
    const uint8_t* D = desc + (size_t)first[p] * 32;
    std::vector<float> Dist((size_t)N * N);
    for (int i = 0; i < N; ++i) {
      Dist[(size_t)i * N + i] = 0;
      for (int j = i + 1; j < N; ++j) {
        const int d = vo_descriptor_distance(D + (size_t)i * 32, D + (size_t)j * 32);
        Dist[(size_t)i * N + j] = d, Dist[(size_t)j * N + i] = d;
      }
    }
    int BestMedian = INT_MAX, BestIdx = 0;
    for (int i = 0; i < N; ++i) {
      std::vector<int> v(Dist.begin() + (size_t)i * N, Dist.begin() + (size_t)(i + 1) * N);
      std::sort(v.begin(), v.end());
      const int median = v[(size_t)(0.5 * (N - 1))];
      if (median < BestMedian) BestMedian = median, BestIdx = i;
    }
    best[p] = BestIdx;
  }
}

void vo_update_normal_and_depth_batch(const float* pts, const int32_t* first, const int32_t* obs_centre,
                                      const float* centres, const int32_t* ref_centre, const float* ref_scale,
                                      float scale_last, int n, float* normal, float* max_d, float* min_d) {
  for (int p = 0; p < n; ++p) {
    const float* Pos = pts + 3 * p;
    float nrm[3] = {0, 0, 0};
    int cnt = 0;
    for (int i = first[p]; i < first[p + 1]; ++i) {
      const float* c = centres + 3 * obs_centre[i];
      const float d[3] = {Pos[0] - c[0], Pos[1] - c[1], Pos[2] - c[2]};
      const float nn = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      for (int r = 0; r < 3; ++r) nrm[r] = nrm[r] + d[r] / nn;
      cnt++;
    }
    if (!cnt) {  // observations.empty(): returns without touching the members
      normal[3 * p] = normal[3 * p + 1] = normal[3 * p + 2] = 0, max_d[p] = min_d[p] = -1;
      continue;
    }
    const float* rc = centres + 3 * ref_centre[p];
    const float PC[3] = {Pos[0] - rc[0], Pos[1] - rc[1], Pos[2] - rc[2]};
    const float dist = std::sqrt(PC[0] * PC[0] + PC[1] * PC[1] + PC[2] * PC[2]);
    for (int r = 0; r < 3; ++r) normal[3 * p + r] = nrm[r] / cnt;
    max_d[p] = dist * ref_scale[p];
    min_d[p] = max_d[p] / scale_last;
  }
}

}  // extern "C"
