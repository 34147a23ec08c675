// ORACLE -- TEST INFRASTRUCTURE ONLY (see ocv_prims.hpp header).  PARITY UNPINNED.
//
// CPU restatement of the descriptor-matching pieces of the hot path:
//   ORBmatcher::DescriptorDistance          src/ORBmatcher.cc:1645-1667
//   Frame::ComputeStereoMatches             src/Frame.cc:451-611      (rectified stereo)
//   cv::BFMatcher(NORM_HAMMING).knnMatch    src/Frame.cc:18,620-628   (OpenCV batchDistance, k=2)
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "ocv_prims.hpp"

namespace vo {

struct KeyPoint {  // cv::KeyPoint layout (28 B); same struct as in orb_extractor.cc
  float x, y, size, angle, response;
  int32_t octave, class_id;
};

// ORBmatcher.cc:1645-1667 (SWAR popcount over 8 x uint32)
static int DescriptorDistance(const uint8_t* a, const uint8_t* b) {
  const int32_t* pa = (const int32_t*)a;
  const int32_t* pb = (const int32_t*)b;
  int dist = 0;
  for (int i = 0; i < 8; i++, pa++, pb++) {
    unsigned int v = *pa ^ *pb;
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

static const int TH_HIGH = 100, TH_LOW = 50;  // ORBmatcher.cc:20-21

// Frame.cc:451-611.  pyrL/pyrR: borderless pyramid planes (mvImagePyramid ROI views).
static void ComputeStereoMatches(const std::vector<Plane>& pyrL, const std::vector<Plane>& pyrR,
                                 const KeyPoint* keysL, int N, const uint8_t* descL,
                                 const KeyPoint* keysR, int Nr, const uint8_t* descR,
                                 const float* vscalefactor, const float* mvInvScaleFactors,
                                 float baseline, float bf, float* vuright, float* vdepth) {
  for (int i = 0; i < N; i++) vuright[i] = -1.0f, vdepth[i] = -1.0f;
  const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
  const int nRows = pyrL[0].h;
  std::vector<std::vector<size_t>> vRowIndices(nRows, std::vector<size_t>());
  for (int iR = 0; iR < Nr; iR++) {
    const KeyPoint& kp = keysR[iR];
    const float& kpY = kp.y;
    const float r = 2.0f * vscalefactor[keysR[iR].octave];
    const int maxr = ceil(kpY + r);
    const int minr = floor(kpY - r);
    for (int yi = minr; yi <= maxr; yi++)
      if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR);  // (reference: unchecked)
  }
  const float minZ = baseline;
  const float minD = 0;
  const float maxD = bf / minZ;
  std::vector<std::pair<int, int>> vDistIdx;
  vDistIdx.reserve(N);
  for (int iL = 0; iL < N; iL++) {
    const KeyPoint& kpL = keysL[iL];
    const int& levelL = kpL.octave;
    const float& vL = kpL.y;
    const float& uL = kpL.x;
    const std::vector<size_t>& vCandidates = vRowIndices[(size_t)vL];
    if (vCandidates.empty()) continue;
    const float minU = uL - maxD;
    const float maxU = uL - minD;
    if (maxU < 0) continue;
    int bestDist = TH_HIGH;
    size_t bestIdxR = 0;
    const uint8_t* dL = descL + (size_t)iL * 32;
    for (size_t iC = 0; iC < vCandidates.size(); iC++) {
      const size_t iR = vCandidates[iC];
      const KeyPoint& kpR = keysR[iR];
      if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
      const float& uR = kpR.x;
      if (uR >= minU && uR <= maxU) {
        const int dist = DescriptorDistance(dL, descR + iR * 32);
        if (dist < bestDist) {
          bestDist = dist;
          bestIdxR = iR;
        }
      }
    }
    if (bestDist < thOrbDist) {
      const float uR0 = keysR[bestIdxR].x;
      const float scaleFactor = mvInvScaleFactors[kpL.octave];
      const float scaleduL = round(kpL.x * scaleFactor);
      const float scaledvL = round(kpL.y * scaleFactor);
      const float scaleduR0 = round(uR0 * scaleFactor);
      const int w = 5;
      const Plane& PL = pyrL[kpL.octave];
      const Plane& PR = pyrR[kpL.octave];
      // IL = patch - centre value, as float (exact small integers)
      float IL[11][11];
      {
        const int r0 = (int)(scaledvL - w), c0 = (int)(scaleduL - w);
        const float cv = PL.row(r0 + w)[c0 + w];
        for (int y = 0; y < 11; y++)
          for (int x = 0; x < 11; x++) IL[y][x] = (float)PL.row(r0 + y)[c0 + x] - cv;
      }
      int bestDistS = INT_MAX;
      int bestincR = 0;
      const int L = 5;
      std::vector<float> vDists(2 * L + 1);
      const float iniu = scaleduR0 + L - w;
      const float endu = scaleduR0 + L + w + 1;
      if (iniu < 0 || endu >= PR.w) continue;
      for (int incR = -L; incR <= +L; incR++) {
        const int r0 = (int)(scaledvL - w), c0 = (int)(scaleduR0 + incR - w);
        const float cv = PR.row(r0 + w)[c0 + w];
        double acc = 0;  // cv::norm(NORM_L1) on CV_32F accumulates in double
        for (int y = 0; y < 11; y++)
          for (int x = 0; x < 11; x++)
            acc += std::abs((double)(IL[y][x] - ((float)PR.row(r0 + y)[c0 + x] - cv)));
        float dist = (float)acc;
        if (dist < bestDistS) {
          bestDistS = dist;
          bestincR = incR;
        }
        vDists[L + incR] = dist;
      }
      if (bestincR == -L || bestincR == L) continue;
      const float dist1 = vDists[L + bestincR - 1];
      const float dist2 = vDists[L + bestincR];
      const float dist3 = vDists[L + bestincR + 1];
      const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
      if (deltaR < -1 || deltaR > 1) continue;
      float bestuR = vscalefactor[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
      float disparity = (uL - bestuR);
      if (disparity >= minD && disparity < maxD) {
        if (disparity <= 0) {
          disparity = 0.01;
          bestuR = uL - 0.01;
        }
        vdepth[iL] = bf / disparity;
        vuright[iL] = bestuR;
        vDistIdx.push_back(std::pair<int, int>(bestDistS, iL));
      }
    }
  }
  if (vDistIdx.empty()) return;  // (reference: undefined behaviour on an empty vector)
  std::sort(vDistIdx.begin(), vDistIdx.end());
  const float median = vDistIdx[vDistIdx.size() / 2].first;
  const float thDist = 1.5f * 1.4f * median;
  for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
    if (vDistIdx[i].first < thDist)
      break;
    else {
      vuright[vDistIdx[i].second] = -1;
      vdepth[vDistIdx[i].second] = -1;
    }
  }
}

// cv::batchDistance(K=2, NORM_HAMMING) as used by BFMatcher::knnMatch: stable insertion, so
// equal distances keep the lower train index first; unfilled slots stay (-1, INT_MAX).
static void knn2Hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx,
                        int32_t* dist) {
  const int K = 2;
  for (int i = 0; i < nq; i++) {
    int* nidxptr = idx + (size_t)i * K;
    int* distptr = dist + (size_t)i * K;
    for (int k = 0; k < K; k++) nidxptr[k] = -1, distptr[k] = INT_MAX;
    for (int j = 0; j < nt; j++) {
      int d = DescriptorDistance(q + (size_t)i * 32, t + (size_t)j * 32);
      if (d < distptr[K - 1]) {
        int k;
        for (k = K - 2; k >= 0 && distptr[k] > d; k--) {
          nidxptr[k + 1] = nidxptr[k];
          distptr[k + 1] = distptr[k];
        }
        nidxptr[k + 1] = j;
        distptr[k + 1] = d;
      }
    }
  }
}

}  // namespace vo

// pyramid access of the extractor oracle (orb_extractor.cc)
extern "C" void vo_orb_level_size(void* h, int level, int* w, int* hgt);
extern "C" void vo_orb_get_plane(void* h, int level, int which, uint8_t* dst);

extern "C" {

int vo_descriptor_distance(const uint8_t* a, const uint8_t* b) {
  return vo::DescriptorDistance(a, b);
}

void vo_knn2_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx,
                     int32_t* dist) {
  vo::knn2Hamming(q, nq, t, nt, idx, dist);
}

// extL / extR: oracle extractor handles that just processed the left / right image.
void vo_stereo_match_rectified(void* extL, void* extR, int nlevels, const void* keysL, int N,
                               const uint8_t* descL, const void* keysR, int Nr,
                               const uint8_t* descR, const float* scale, const float* inv_scale,
                               float baseline, float bf, float* vuright, float* vdepth) {
  std::vector<vo::Plane> pl(nlevels), pr(nlevels);
  for (int l = 0; l < nlevels; l++) {
    int w, h;
    vo_orb_level_size(extL, l, &w, &h);
    pl[l] = vo::Plane(w, h);
    vo_orb_get_plane(extL, l, 0, pl[l].px.data());
    vo_orb_level_size(extR, l, &w, &h);
    pr[l] = vo::Plane(w, h);
    vo_orb_get_plane(extR, l, 0, pr[l].px.data());
  }
  vo::ComputeStereoMatches(pl, pr, (const vo::KeyPoint*)keysL, N, descL,
                           (const vo::KeyPoint*)keysR, Nr, descR, scale, inv_scale, baseline, bf,
                           vuright, vdepth);
}

}  // extern "C"
