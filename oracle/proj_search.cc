// ORACLE -- TEST INFRASTRUCTURE ONLY (see ocv_prims.hpp header).  PARITY UNPINNED.
//
// CPU restatement of the tracking-side projection searches on flattened inputs:
//   FrameBase::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea / IsInImage   src/FrameBase.cpp:95-174
//   ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono, th_far)       src/ORBmatcher.cc:1303-1467
//   ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th, th_far)        src/ORBmatcher.cc:230-335
//   ORBmatcher::ComputeThreeMaxima                                                src/ORBmatcher.cc:1608-1641
//   ... and their camera loops for frames of a rig (mpCameras[camj]->GetTcr(), Project() when usedistort_,
//   GetFeaturesInArea(cami, ...) on per-camera grids): ORBmatcher.cc:1339-1366, :257-266, :1491-1543
// The sequential dependence of the reference (a later query sees the keypoints earlier queries
// claimed through AddMapPoint) is kept literally.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/vieo_hot.h"
#include "cam_models.hpp"

extern "C" int vo_descriptor_distance(const uint8_t* a, const uint8_t* b);

namespace vo {

static const int FRAME_GRID_ROWS = 48, FRAME_GRID_COLS = 64;  // FrameBase.h:224-225
static const int TH_HIGH_ = 100, HISTO_LENGTH = 30;           // ORBmatcher.cc:20-22

struct FrameFeat {
  const vieo_keypoint* keys;
  const float* uright;
  const uint8_t* desc;
  int N;
  // per camera (FrameBase::gridinfo_, vgrids_[cami]); one camera for an undistorted / single-camera frame
  int n_cams = 1;
  float minx[4], maxx[4], miny[4], maxy[4], winv[4], hinv[4];
  std::vector<std::vector<size_t>> grid[4];
  // mvpMapPoints state during a search: -1 none, -3 pre-existing observed map point,
  // >= 0 query index placed by AddMapPoint in this call
  std::vector<int> mp;
  std::vector<uint8_t> mp_observed;  // Observations() > 0 of the map point currently there

  // cam_first[c] .. cam_first[c + 1]: the keys of camera c in mvKeys (Frame.cc:738-764 concatenates camera-major,
  // mapn2in_[i] = (camera, index in the camera)); null: one camera owning all keys
  void build(const float* bounds, int ncams = 1, const int32_t* cam_first = nullptr) {
    n_cams = ncams;
    for (int c = 0; c < n_cams; ++c) {
      const float* b = bounds + 4 * c;
      minx[c] = b[0], maxx[c] = b[1], miny[c] = b[2], maxy[c] = b[3];
      winv[c] = FRAME_GRID_COLS / (maxx[c] - minx[c]);  // FrameBase.cpp:214-217
      hinv[c] = FRAME_GRID_ROWS / (maxy[c] - miny[c]);
      grid[c].assign(FRAME_GRID_COLS * FRAME_GRID_ROWS, {});
    }
    int cami = 0;
    for (int i = 0; i < N; ++i) {  // AssignFeaturesToGrid + PosInGrid
      if (cam_first)
        while (cami + 1 < n_cams && i >= cam_first[cami + 1]) ++cami;
      int posX = (int)round((keys[i].x - minx[cami]) * winv[cami]);
      int posY = (int)round((keys[i].y - miny[cami]) * hinv[cami]);
      if (posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS) continue;
      grid[cami][posX * FRAME_GRID_ROWS + posY].push_back(i);
    }
  }

  // FrameBase.cpp:95-141
  std::vector<size_t> GetFeaturesInArea(int cami, float x, float y, float r, int minlevel, int maxlevel) const {
    std::vector<size_t> vIndices;
    const int min_cellx = std::max(0, (int)floor((x - minx[cami] - r) * winv[cami]));
    if (min_cellx >= FRAME_GRID_COLS) return vIndices;
    const int max_cellx = std::min((int)FRAME_GRID_COLS - 1, (int)ceil((x - minx[cami] + r) * winv[cami]));
    if (max_cellx < 0) return vIndices;
    const int min_celly = std::max(0, (int)floor((y - miny[cami] - r) * hinv[cami]));
    if (min_celly >= FRAME_GRID_ROWS) return vIndices;
    const int max_celly = std::min((int)FRAME_GRID_ROWS - 1, (int)ceil((y - miny[cami] + r) * hinv[cami]));
    if (max_celly < 0) return vIndices;
    const bool bchecklevel = (minlevel > 0) || (maxlevel >= 0);
    for (int ix = min_cellx; ix <= max_cellx; ++ix)
      for (int iy = min_celly; iy <= max_celly; ++iy) {
        const std::vector<size_t>& vCell = grid[cami][ix * FRAME_GRID_ROWS + iy];
        for (size_t j = 0, jend = vCell.size(); j < jend; ++j) {
          const vieo_keypoint& kpUn = keys[vCell[j]];
          if (bchecklevel) {
            if (kpUn.octave < minlevel) continue;
            if (maxlevel >= 0)
              if (kpUn.octave > maxlevel) continue;
          }
          const float distx = kpUn.x - x;
          const float disty = kpUn.y - y;
          if (fabs(distx) < r && fabs(disty) < r) vIndices.push_back(vCell[j]);
        }
      }
    return vIndices;
  }
};

// ORBmatcher.cc:1608-1641
static void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {
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

static int search(int mode, const vieo_proj_query* Q, int nq, FrameFeat& F, const uint8_t* taken,
                  float nnratio, bool checkOri, int32_t* assign) {
  int nmatches = 0;
  F.mp.assign(F.N, -1);
  F.mp_observed.assign(F.N, 0);
  for (int i = 0; i < F.N; i++) {
    assign[i] = VIEO_SBP_UNCHANGED;
    if (taken && taken[i]) F.mp[i] = -3, F.mp_observed[i] = 1;
  }
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  for (int q = 0; q < nq; q++) {
    const vieo_proj_query& p = Q[q];
    if (!(p.flags & 1)) continue;
    const int cami = (p.flags >> 8) & 15;  // the camera the query was projected into (camj / *iter_cami)
    if (cami >= F.n_cams) continue;
    const std::vector<size_t> vIndices = F.GetFeaturesInArea(cami, p.u, p.v, p.radius, p.level_min, p.level_max);
    if (vIndices.empty()) continue;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (size_t k = 0; k < vIndices.size(); k++) {
      const size_t idx = vIndices[k];
      if (mode == VIEO_SBP_RELOC) {
        if (F.mp[idx] != -1) continue;  // if (curfmps[i2]) continue;  ORBmatcher.cc:1553-1555
      } else if (F.mp[idx] != -1)
        if (F.mp_observed[idx]) continue;
      if (mode != VIEO_SBP_RELOC && F.uright[idx] > 0) {
        const float er = fabs(p.ur - F.uright[idx]);
        if (er > p.radius) continue;
      }
      const int dist = vo_descriptor_distance(p.desc, F.desc + idx * 32);
      if (mode != VIEO_SBP_LOCAL_MAP) {
        if (dist < bestDist) {
          bestDist = dist;
          bestIdx = (int)idx;
        }
      } else {
        if (dist < bestDist) {
          bestDist2 = bestDist;
          bestDist = dist;
          bestLevel2 = bestLevel;
          bestLevel = F.keys[idx].octave;
          bestIdx = (int)idx;
        } else if (dist < bestDist2) {
          bestLevel2 = F.keys[idx].octave;
          bestDist2 = dist;
        }
      }
    }
    if (bestDist <= (mode == VIEO_SBP_RELOC ? (int)nnratio : TH_HIGH_)) {  // ORBdist, ORBmatcher.cc:1571
      if (mode == VIEO_SBP_LOCAL_MAP)
        if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
      // AddMapPoint(pMP, bestIdx)
      F.mp[bestIdx] = q;
      F.mp_observed[bestIdx] = (p.flags & 2) ? 1 : 0;
      assign[bestIdx] = q;
      nmatches++;
      if (mode != VIEO_SBP_LOCAL_MAP && checkOri) {
        float rot = p.angle - F.keys[bestIdx].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin].push_back(bestIdx);
      }
    }
  }
  if (mode != VIEO_SBP_LOCAL_MAP && checkOri) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
          assign[rotHist[i][j]] = VIEO_SBP_ERASED;  // EraseMapPointMatch
          nmatches--;
        }
  }
  return nmatches;
}

// ORBmatcher.cc:1313-1378: projection of the last frame's map points -> queries
static void project_last_frame(const vieo_last_frame_point* P, int n, const vieo_sbp_camera& C,
                               vieo_proj_query* Q) {
  // Tlrcr = Tlrw * Tcrw^-1 ; translation = tlrw - Rlrw * Rcrw^T * tcrw
  const double* Tc = C.Tcw_cur;
  const double* Tl = C.Tcw_last;
  double Rlc[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      Rlc[i * 3 + j] = Tl[i * 4] * Tc[j * 4] + Tl[i * 4 + 1] * Tc[j * 4 + 1] + Tl[i * 4 + 2] * Tc[j * 4 + 2];
  const double tz = Tl[2 * 4 + 3] - (Rlc[6] * Tc[3] + Rlc[7] * Tc[7] + Rlc[8] * Tc[11]);
  const bool bForward = tz > C.baseline && !C.mono;
  const bool bBackward = -tz > C.baseline && !C.mono;
  for (int i = 0; i < n; i++) {
    vieo_proj_query& q = Q[i];
    memset(&q, 0, sizeof(q));
    const vieo_last_frame_point& p = P[i];
    if (!(p.flags & 1)) continue;
    const double X = p.Xw[0], Y = p.Xw[1], Z = p.Xw[2];
    const double x3 = Tc[0] * X + Tc[1] * Y + Tc[2] * Z + Tc[3];
    const double y3 = Tc[4] * X + Tc[5] * Y + Tc[6] * Z + Tc[7];
    const double z3 = Tc[8] * X + Tc[9] * Y + Tc[10] * Z + Tc[11];
    if (C.th_far > 0 && z3 > C.th_far) continue;
    const float xc = x3, yc = y3;
    const float invzc = 1.0 / z3;
    if (invzc < 0) continue;
    // uv = K.cast<float>() * (xc*invzc, yc*invzc, 1)
    const float pnx = xc * invzc, pny = yc * invzc;
    const float u = C.fx * pnx + 0.f * pny + C.cx * 1.f;
    const float v = 0.f * pnx + C.fy * pny + C.cy * 1.f;
    if (!(u >= C.bounds[0] && u < C.bounds[1] && v >= C.bounds[2] && v < C.bounds[3])) continue;
    const int nLastOctave = p.octave;
    q.u = u, q.v = v;
    q.ur = u - C.bf * invzc;
    q.radius = C.th * C.scale[nLastOctave];
    if (bForward)
      q.level_min = 0, q.level_max = nLastOctave;
    else if (bBackward)
      q.level_min = nLastOctave, q.level_max = -1;
    else
      q.level_min = nLastOctave - 1, q.level_max = nLastOctave + 1;
    q.angle = p.angle;
    q.flags = 1 | (p.flags & 2);
    memcpy(q.desc, p.desc, 32);
  }
}

static void rig_cams(const vieo_sbp_rig& R, OCam* cams) {
  for (int c = 0; c < R.n_cams; ++c) {
    const vieo_camera& s = R.cams[c];
    cams[c].model = s.model, cams[c].num_k = s.model == VIEO_CAM_RADTAN ? s.num_k : 0;
    cams[c].fx = s.fx, cams[c].fy = s.fy, cams[c].cx = s.cx, cams[c].cy = s.cy;
    for (int q = 0; q < 8; ++q) cams[c].dist[q] = s.dist[q];
  }
}

// u, v of a point in camera coordinates: ORBmatcher.cc:1352-1363 / :1503-1516
static void rig_uv(const vieo_sbp_rig& R, const OCam& cam, const double* x3Dc, float invzc, float* u, float* v) {
  if (R.use_distort) {
    float pt[2];
    ocam_project(cam, x3Dc, pt, nullptr);  // mpCameras[camj]->Project(x3Dc.cast<double>(), &pt)
    *u = pt[0], *v = pt[1];
  } else {
    const float xc = x3Dc[0], yc = x3Dc[1];
    const float pnx = xc * invzc, pny = yc * invzc;  // K.cast<float>() * (xc*invzc, yc*invzc, 1)
    *u = cam.fx * pnx + 0.f * pny + cam.cx * 1.f;
    *v = 0.f * pnx + cam.fy * pny + cam.cy * 1.f;
  }
}

// ORBmatcher.cc:1313-1378 with the camera loop: query (i, camj) at Q[i * n_cams + camj]
static void project_last_frame_rig(const vieo_last_frame_point* P, int n, const vieo_sbp_camera& C,
                                   const vieo_sbp_rig& R, vieo_proj_query* Q) {
  const double* Tc = C.Tcw_cur;
  const double* Tl = C.Tcw_last;
  double Rlc[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      Rlc[i * 3 + j] = Tl[i * 4] * Tc[j * 4] + Tl[i * 4 + 1] * Tc[j * 4 + 1] + Tl[i * 4 + 2] * Tc[j * 4 + 2];
  const double tz = Tl[2 * 4 + 3] - (Rlc[6] * Tc[3] + Rlc[7] * Tc[7] + Rlc[8] * Tc[11]);
  const bool bForward = tz > C.baseline && !C.mono;
  const bool bBackward = -tz > C.baseline && !C.mono;
  OCam cams[4];
  rig_cams(R, cams);
  const int nc = R.n_cams;
  memset(Q, 0, sizeof(vieo_proj_query) * (size_t)n * nc);
  for (int i = 0; i < n; i++) {
    const vieo_last_frame_point& p = P[i];
    if (!(p.flags & 1)) continue;
    const double X = p.Xw[0], Y = p.Xw[1], Z = p.Xw[2];
    const double x3Dr[3] = {Tc[0] * X + Tc[1] * Y + Tc[2] * Z + Tc[3], Tc[4] * X + Tc[5] * Y + Tc[6] * Z + Tc[7],
                            Tc[8] * X + Tc[9] * Y + Tc[10] * Z + Tc[11]};
    if (C.th_far > 0 && x3Dr[2] > C.th_far) continue;
    for (int camj = 0; camj < nc; ++camj) {
      const double* T = R.Tcr[camj];
      double x3Dc[3];
      for (int r = 0; r < 3; ++r) x3Dc[r] = T[r * 4] * x3Dr[0] + T[r * 4 + 1] * x3Dr[1] + T[r * 4 + 2] * x3Dr[2] + T[r * 4 + 3];
      const float invzc = 1.0 / x3Dc[2];
      if (invzc < 0) continue;
      float u, v;
      rig_uv(R, cams[camj], x3Dc, invzc, &u, &v);
      const float* b = R.bounds[camj];
      if (!(u >= b[0] && u < b[1] && v >= b[2] && v < b[3])) continue;  // IsInImage(camj, u, v)
      const int nLastOctave = p.octave;
      vieo_proj_query& q = Q[(size_t)i * nc + camj];
      q.u = u, q.v = v;
      q.ur = u - C.bf * invzc;
      q.radius = C.th * C.scale[nLastOctave];
      if (bForward)
        q.level_min = 0, q.level_max = nLastOctave;
      else if (bBackward)
        q.level_min = nLastOctave, q.level_max = -1;
      else
        q.level_min = nLastOctave - 1, q.level_max = nLastOctave + 1;
      q.angle = p.angle;
      q.flags = 1 | (p.flags & 2) | (camj << 8);
      memcpy(q.desc, p.desc, 32);
    }
  }
}

// ORBmatcher.cc:1487-1543 (relocalisation overload): no positive-depth test, double camera centre, PredictScale
static void project_keyframe(const vieo_keyframe_point* P, int n, const vieo_sbp_camera& C, const vieo_sbp_rig* Rp,
                             float log_scale_factor, vieo_proj_query* Q) {
  vieo_sbp_rig one;
  if (!Rp) {  // the rectified camera of C as a rig of one
    memset(&one, 0, sizeof(one));
    one.n_cams = 1;
    one.cams[0].fx = C.fx, one.cams[0].fy = C.fy, one.cams[0].cx = C.cx, one.cams[0].cy = C.cy;
    one.Tcr[0][0] = one.Tcr[0][5] = one.Tcr[0][10] = 1.0;
    memcpy(one.bounds[0], C.bounds, sizeof(C.bounds));
    Rp = &one;
  }
  const vieo_sbp_rig& R = *Rp;
  const double* Tc = C.Tcw_cur;
  // Twcr = Tcrw.inverse(): Rwcr = Rcrw^T, twcr = Rwcr * (-tcrw)
  double Rw[9], tw[3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Rw[i * 3 + j] = Tc[j * 4 + i];
  for (int i = 0; i < 3; i++) tw[i] = Rw[i * 3] * (Tc[3] * -1.0) + Rw[i * 3 + 1] * (Tc[7] * -1.0) + Rw[i * 3 + 2] * (Tc[11] * -1.0);
  OCam cams[4];
  rig_cams(R, cams);
  const int nc = R.n_cams;
  memset(Q, 0, sizeof(vieo_proj_query) * (size_t)n * nc);
  for (int i = 0; i < n; i++) {
    const vieo_keyframe_point& p = P[i];
    if (!(p.flags & 1)) continue;
    const double Xw[3] = {p.Xw[0], p.Xw[1], p.Xw[2]};
    double x3Dcr[3];
    for (int r = 0; r < 3; ++r) x3Dcr[r] = Tc[r * 4] * Xw[0] + Tc[r * 4 + 1] * Xw[1] + Tc[r * 4 + 2] * Xw[2] + Tc[r * 4 + 3];
    if (C.th_far > 0 && x3Dcr[2] > C.th_far) continue;
    for (int cami = 0; cami < nc; ++cami) {
      const double* T = R.Tcr[cami];
      double Pc[3], twc[3];
      for (int r = 0; r < 3; ++r) Pc[r] = T[r * 4] * x3Dcr[0] + T[r * 4 + 1] * x3Dcr[1] + T[r * 4 + 2] * x3Dcr[2] + T[r * 4 + 3];
      const double* t = R.trc[cami];
      for (int r = 0; r < 3; ++r) twc[r] = tw[r] + (Rw[r * 3] * t[0] + Rw[r * 3 + 1] * t[1] + Rw[r * 3 + 2] * t[2]);
      const float invzc = 1.0 / Pc[2];
      float u, v;
      rig_uv(R, cams[cami], Pc, invzc, &u, &v);
      const float* b = R.bounds[cami];
      if (!(u >= b[0] && u < b[1] && v >= b[2] && v < b[3])) continue;
      const double PO[3] = {Xw[0] - twc[0], Xw[1] - twc[1], Xw[2] - twc[2]};
      const float dist3D = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
      const float maxDistance = 1.2f * p.max_distance, minDistance = 0.8f * p.min_distance;
      if (dist3D < minDistance || dist3D > maxDistance) continue;
      const float ratio = p.max_distance / dist3D;  // MapPoint::PredictScale, see oracle/mappoint.cc for the log
      int lvl = (int)std::ceil((float)std::log((double)ratio) / log_scale_factor);
      if (lvl < 0)
        lvl = 0;
      else if (lvl >= C.nlevels)
        lvl = C.nlevels - 1;
      vieo_proj_query& q = Q[(size_t)i * nc + cami];
      q.u = u, q.v = v, q.ur = u - C.bf * invzc;  // ur is not read by this variant
      q.radius = C.th * C.scale[lvl];
      q.level_min = lvl - 1, q.level_max = lvl + 1;
      q.angle = p.angle;
      q.flags = 1 | (p.flags & 2) | (cami << 8);
      memcpy(q.desc, p.desc, 32);
    }
  }
}

}  // namespace vo

extern "C" {

void vo_sbp_project_last_frame(const vieo_last_frame_point* pts, int n, const vieo_sbp_camera* cam,
                               vieo_proj_query* queries) {
  vo::project_last_frame(pts, n, *cam, queries);
}

int vo_search_by_projection(int mode, const vieo_proj_query* queries, int nq,
                            const vieo_keypoint* keys, const float* uright, const uint8_t* desc,
                            const uint8_t* taken, int n_keys, const float* bounds, float nn_ratio,
                            int check_orientation, int32_t* assign) {
  vo::FrameFeat F;
  F.keys = keys, F.uright = uright, F.desc = desc, F.N = n_keys;
  F.build(bounds);
  return vo::search(mode, queries, nq, F, taken, nn_ratio, check_orientation != 0, assign);
}

void vo_sbp_project_last_frame_rig(const vieo_last_frame_point* pts, int n, const vieo_sbp_camera* cam,
                                   const vieo_sbp_rig* rig, vieo_proj_query* queries) {
  vo::project_last_frame_rig(pts, n, *cam, *rig, queries);
}

void vo_sbp_project_keyframe(const vieo_keyframe_point* pts, int n, const vieo_sbp_camera* cam, const vieo_sbp_rig* rig,
                             float log_scale_factor, vieo_proj_query* queries) {
  vo::project_keyframe(pts, n, *cam, rig, log_scale_factor, queries);
}

int vo_search_by_projection_rig(int mode, const vieo_proj_query* queries, int nq, const vieo_keypoint* keys,
                                const float* uright, const uint8_t* desc, const uint8_t* taken, int n_keys,
                                const int32_t* cam_first, const float* bounds, int n_cams, float nn_ratio,
                                int check_orientation, int32_t* assign) {
  vo::FrameFeat F;
  F.keys = keys, F.uright = uright, F.desc = desc, F.N = n_keys;
  F.build(bounds, n_cams, cam_first);
  return vo::search(mode, queries, nq, F, taken, nn_ratio, check_orientation != 0, assign);
}

}  // extern "C"
