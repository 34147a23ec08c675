// ORACLE -- TEST INFRASTRUCTURE ONLY (see ocv_prims.hpp header).  PARITY UNPINNED.
//
// Literal CPU restatement of the reference ORB extractor, following
//   /root/reference/src/ORBextractor.cc  and  include/ORBextractor.h
// function by function (line numbers cited at each function).  It keeps the reference's
// sequential structure on purpose (std::list quadtree with push_front ordering, one FAST call
// per cell, per-level blur of a clone) so the data-parallel HIP path has an independent checker.
//
// One deliberate, documented determinisation: ORBextractor.cc:647 sorts
// pair<int, ExtractorNode*> and therefore breaks size ties BY HEAP ADDRESS, which is
// allocator-dependent in the reference.  Here ties are broken by node creation order
// (later-created node compares greater), i.e. the address order of a bump allocator.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <list>
#include <vector>

#include "../include/vieo_orb_pattern_31.h"
#include "ocv_prims.hpp"

namespace vo {

static const int PATCH_SIZE = 31;       // ORBextractor.cc:51
static const int HALF_PATCH_SIZE = 15;  // :52
static const int EDGE_THRESHOLD = 19;   // :53

struct KeyPoint {  // cv::KeyPoint layout (28 B)
  float x, y, size, angle, response;
  int32_t octave, class_id;
};

// ORBextractor.cc:55-80
static float IC_Angle(const Plane& image, float ptx, float pty, const std::vector<int>& u_max) {
  int m_01 = 0, m_10 = 0;
  const int cx = cvRound(ptx), cy = cvRound(pty);
  const int step = image.w;
  const uint8_t* center = image.row(cy) + cx;
  for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
  for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
    int v_sum = 0;
    int d = u_max[v];
    for (int u = -d; u <= d; ++u) {
      int val_plus = center[u + v * step], val_minus = center[u - v * step];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return fastAtan2((float)m_01, (float)m_10);
}

// ORBextractor.cc:82-127.  `cos(angle)` with a float argument and `using namespace std`
// resolves to the float overload (cosf).
static const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
static void computeOrbDescriptor(const KeyPoint& kpt, const Plane& img, uint8_t* desc) {
  float angle = (float)kpt.angle * factorPI;
  float a = cosf(angle), b = sinf(angle);
  const int cx = cvRound(kpt.x), cy = cvRound(kpt.y);
  const int step = img.w;
  const uint8_t* center = img.row(cy) + cx;
  const signed char* pat = VIEO_ORB_PATTERN_31;
  for (int i = 0; i < 32; ++i) {
    int val = 0;
    for (int t = 0; t < 8; ++t, pat += 4) {
      // GET_VALUE(idx): center[cvRound(x*b + y*a)*step + cvRound(x*a - y*b)]
      float x0 = pat[0], y0 = pat[1], x1 = pat[2], y1 = pat[3];
      int t0 = center[cvRound(x0 * b + y0 * a) * step + cvRound(x0 * a - y0 * b)];
      int t1 = center[cvRound(x1 * b + y1 * a) * step + cvRound(x1 * a - y1 * b)];
      val |= (t0 < t1) << t;
    }
    desc[i] = (uint8_t)val;
  }
}

struct NodeKey {
  float x, y, response;
  int idx;  // position in vToDistributeKeys (diagnostics)
};

struct ExtractorNode {  // include/ORBextractor.h:14-25
  std::vector<NodeKey> vKeys;
  int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
  std::list<ExtractorNode>::iterator lit;
  bool bNoMore = false;
  long seq = 0;  // creation order: stands in for the heap address in the (size, ptr) sort
  void DivideNode(ExtractorNode& n1, ExtractorNode& n2, ExtractorNode& n3, ExtractorNode& n4);
};

// ORBextractor.cc:467-516
void ExtractorNode::DivideNode(ExtractorNode& n1, ExtractorNode& n2, ExtractorNode& n3,
                               ExtractorNode& n4) {
  const int halfX = (int)std::ceil(static_cast<float>(URx - ULx) / 2);
  const int halfY = (int)std::ceil(static_cast<float>(BRy - ULy) / 2);
  n1.ULx = ULx, n1.ULy = ULy;
  n1.URx = ULx + halfX, n1.URy = ULy;
  n1.BLx = ULx, n1.BLy = ULy + halfY;
  n1.BRx = ULx + halfX, n1.BRy = ULy + halfY;
  n2.ULx = n1.URx, n2.ULy = n1.URy;
  n2.URx = URx, n2.URy = URy;
  n2.BLx = n1.BRx, n2.BLy = n1.BRy;
  n2.BRx = URx, n2.BRy = ULy + halfY;
  n3.ULx = n1.BLx, n3.ULy = n1.BLy;
  n3.URx = n1.BRx, n3.URy = n1.BRy;
  n3.BLx = BLx, n3.BLy = BLy;
  n3.BRx = n1.BRx, n3.BRy = BLy;
  n4.ULx = n3.URx, n4.ULy = n3.URy;
  n4.URx = n2.BRx, n4.URy = n2.BRy;
  n4.BLx = n3.BRx, n4.BLy = n3.BRy;
  n4.BRx = BRx, n4.BRy = BRy;
  for (size_t i = 0; i < vKeys.size(); i++) {
    const NodeKey& kp = vKeys[i];
    if (kp.x < n1.URx) {
      if (kp.y < n1.BRy)
        n1.vKeys.push_back(kp);
      else
        n3.vKeys.push_back(kp);
    } else if (kp.y < n1.BRy)
      n2.vKeys.push_back(kp);
    else
      n4.vKeys.push_back(kp);
  }
  if (n1.vKeys.size() == 1) n1.bNoMore = true;
  if (n2.vKeys.size() == 1) n2.bNoMore = true;
  if (n3.vKeys.size() == 1) n3.bNoMore = true;
  if (n4.vKeys.size() == 1) n4.bNoMore = true;
}

struct Extractor {
  int nfeatures, nlevels, iniThFAST, minThFAST;
  double scaleFactor;  // include/ORBextractor.h:67 (double member, float ctor argument)
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
  std::vector<int> mnFeaturesPerLevel, umax;
  std::vector<Plane> mvImagePyramid;  // borderless ROI views (the 19-px border is derived)
  // taps for stage-wise parity tests
  std::vector<std::vector<NodeKey>> tapCandidates;  // vToDistributeKeys per level
  std::vector<std::vector<KeyPoint>> tapLevelKeys;  // after DistributeOctTree + orientation
  std::vector<Plane> tapBlurred;
  long seqCounter = 0;
  long tieCount = 0;  // number of (size) ties met by the sort at ORBextractor.cc:647

  // ORBextractor.cc:391-456
  Extractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
      : nfeatures(_nfeatures),
        nlevels(_nlevels),
        iniThFAST(_iniThFAST),
        minThFAST(_minThFAST),
        scaleFactor(_scaleFactor) {
    mvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvScaleFactor[0] = 1.0f;
    mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
      mvScaleFactor[i] = mvScaleFactor[i - 1] * scaleFactor;  // float * double -> float
      mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
    }
    mvInvScaleFactor.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
      mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
      mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
    }
    mvImagePyramid.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    float factor = 1.0f / scaleFactor;
    float nDesiredFeaturesPerScale =
        nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sumFeatures = 0;
    for (int level = 0; level < nlevels - 1; level++) {
      mnFeaturesPerLevel[level] = cvRound(nDesiredFeaturesPerScale);
      sumFeatures += mnFeaturesPerLevel[level];
      nDesiredFeaturesPerScale *= factor;
    }
    mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);

    umax.resize(HALF_PATCH_SIZE + 1);
    int v, v0, vmax = cvFloor(HALF_PATCH_SIZE * sqrt(2.f) / 2 + 1);
    int vmin = cvCeil(HALF_PATCH_SIZE * sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) umax[v] = cvRound(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
  }

  // ORBextractor.cc:1060-1081 (border is not materialised: nothing on the path reads it)
  void ComputePyramid(const Plane& image) {
    for (int level = 0; level < nlevels; ++level) {
      float scale = mvInvScaleFactor[level];
      int szw = cvRound((float)image.w * scale), szh = cvRound((float)image.h * scale);
      if (level != 0) {
        mvImagePyramid[level] = Plane(szw, szh);
        resizeLinearU8(mvImagePyramid[level - 1], mvImagePyramid[level]);
      } else {
        mvImagePyramid[level] = image;
      }
    }
  }

  // ORBextractor.cc:518-721
  std::vector<NodeKey> DistributeOctTree(const std::vector<NodeKey>& vToDistributeKeys, int minX,
                                         int maxX, int minY, int maxY, int N) {
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::list<ExtractorNode> lNodes;
    std::vector<ExtractorNode*> vpIniNodes(nIni);
    for (int i = 0; i < nIni; i++) {
      ExtractorNode ni;
      ni.ULx = (int)(hX * static_cast<float>(i)), ni.ULy = 0;
      ni.URx = (int)(hX * static_cast<float>(i + 1)), ni.URy = 0;
      ni.BLx = ni.ULx, ni.BLy = maxY - minY;
      ni.BRx = ni.URx, ni.BRy = maxY - minY;
      ni.seq = seqCounter++;
      lNodes.push_back(ni);
      vpIniNodes[i] = &lNodes.back();
    }
    for (size_t i = 0; i < vToDistributeKeys.size(); i++) {
      const NodeKey& kp = vToDistributeKeys[i];
      vpIniNodes[(int)(kp.x / hX)]->vKeys.push_back(kp);
    }
    auto lit = lNodes.begin();
    while (lit != lNodes.end()) {
      if (lit->vKeys.size() == 1) {
        lit->bNoMore = true;
        lit++;
      } else if (lit->vKeys.empty())
        lit = lNodes.erase(lit);
      else
        lit++;
    }
    bool bFinish = false;
    typedef std::pair<int, ExtractorNode*> SizePtr;
    auto lessSizePtr = [this](const SizePtr& a, const SizePtr& b) {
      if (a.first != b.first) return a.first < b.first;
      return a.second->seq < b.second->seq;  // stand-in for pointer compare
    };
    std::vector<SizePtr> vSizeAndPointerToNode;
    auto pushChild = [&](ExtractorNode& n, bool count, int& nToExpand) {
      if (n.vKeys.size() > 0) {
        n.seq = seqCounter++;
        lNodes.push_front(n);
        if (n.vKeys.size() > 1) {
          if (count) nToExpand++;
          vSizeAndPointerToNode.push_back(std::make_pair((int)n.vKeys.size(), &lNodes.front()));
          lNodes.front().lit = lNodes.begin();
        }
      }
    };
    while (!bFinish) {
      int prevSize = (int)lNodes.size();
      lit = lNodes.begin();
      int nToExpand = 0;
      vSizeAndPointerToNode.clear();
      while (lit != lNodes.end()) {
        if (lit->bNoMore) {
          lit++;
          continue;
        } else {
          ExtractorNode n1, n2, n3, n4;
          lit->DivideNode(n1, n2, n3, n4);
          pushChild(n1, true, nToExpand);
          pushChild(n2, true, nToExpand);
          pushChild(n3, true, nToExpand);
          pushChild(n4, true, nToExpand);
          lit = lNodes.erase(lit);
          continue;
        }
      }
      if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
        bFinish = true;
      } else if (((int)lNodes.size() + nToExpand * 3) > N) {
        while (!bFinish) {
          prevSize = (int)lNodes.size();
          std::vector<SizePtr> vPrev = vSizeAndPointerToNode;
          vSizeAndPointerToNode.clear();
          std::sort(vPrev.begin(), vPrev.end(), lessSizePtr);
          for (size_t q = 1; q < vPrev.size(); q++)
            if (vPrev[q].first == vPrev[q - 1].first) tieCount++;
          for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
            ExtractorNode n1, n2, n3, n4;
            vPrev[j].second->DivideNode(n1, n2, n3, n4);
            int dummy = 0;
            pushChild(n1, false, dummy);
            pushChild(n2, false, dummy);
            pushChild(n3, false, dummy);
            pushChild(n4, false, dummy);
            lNodes.erase(vPrev[j].second->lit);
            if ((int)lNodes.size() >= N) break;
          }
          if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
        }
      }
    }
    std::vector<NodeKey> vResultKeys;
    vResultKeys.reserve(nfeatures);
    for (auto it = lNodes.begin(); it != lNodes.end(); it++) {
      std::vector<NodeKey>& vNodeKeys = it->vKeys;
      NodeKey* pKP = &vNodeKeys[0];
      float maxResponse = pKP->response;
      for (size_t k = 1; k < vNodeKeys.size(); k++) {
        if (vNodeKeys[k].response > maxResponse) {
          pKP = &vNodeKeys[k];
          maxResponse = vNodeKeys[k].response;
        }
      }
      vResultKeys.push_back(*pKP);
    }
    return vResultKeys;
  }

  // ORBextractor.cc:723-802
  void ComputeKeyPointsOctTree(std::vector<std::vector<KeyPoint>>& allKeypoints) {
    allKeypoints.resize(nlevels);
    tapCandidates.assign(nlevels, {});
    const float W = 35;
    for (int level = 0; level < nlevels; ++level) {
      const int minBorderX = EDGE_THRESHOLD - 3;
      const int minBorderY = minBorderX;
      const int maxBorderX = mvImagePyramid[level].w - EDGE_THRESHOLD + 3;
      const int maxBorderY = mvImagePyramid[level].h - EDGE_THRESHOLD + 3;
      std::vector<NodeKey> vToDistributeKeys;
      vToDistributeKeys.reserve(nfeatures * 10);
      const float width = (maxBorderX - minBorderX);
      const float height = (maxBorderY - minBorderY);
      const int nCols = width / W;
      const int nRows = height / W;
      const int wCell = ceil(width / nCols);
      const int hCell = ceil(height / nRows);
      std::vector<FastKp> vKeysCell;
      for (int i = 0; i < nRows; i++) {
        const float iniY = minBorderY + i * hCell;
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = maxBorderY;
        for (int j = 0; j < nCols; j++) {
          const float iniX = minBorderX + j * wCell;
          float maxX = iniX + wCell + 6;
          if (iniX >= maxBorderX - 6) continue;
          if (maxX > maxBorderX) maxX = maxBorderX;
          fast9_16(mvImagePyramid[level], (int)iniX, (int)iniY, (int)maxX, (int)maxY, iniThFAST,
                   vKeysCell);
          if (vKeysCell.empty())
            fast9_16(mvImagePyramid[level], (int)iniX, (int)iniY, (int)maxX, (int)maxY,
                     minThFAST, vKeysCell);
          for (auto& k : vKeysCell) {
            NodeKey nk;
            nk.x = (float)k.x + j * wCell;
            nk.y = (float)k.y + i * hCell;
            nk.response = (float)k.score;
            nk.idx = (int)vToDistributeKeys.size();
            vToDistributeKeys.push_back(nk);
          }
        }
      }
      tapCandidates[level] = vToDistributeKeys;
      std::vector<NodeKey> sel = DistributeOctTree(vToDistributeKeys, minBorderX, maxBorderX,
                                                   minBorderY, maxBorderY,
                                                   mnFeaturesPerLevel[level]);
      const int scaledPatchSize = PATCH_SIZE * mvScaleFactor[level];
      std::vector<KeyPoint>& keypoints = allKeypoints[level];
      keypoints.clear();
      for (auto& s : sel) {
        KeyPoint kp;
        kp.x = s.x + minBorderX;
        kp.y = s.y + minBorderY;
        kp.size = scaledPatchSize;
        kp.angle = -1;
        kp.response = s.response;
        kp.octave = level;
        kp.class_id = -1;
        keypoints.push_back(kp);
      }
    }
    for (int level = 0; level < nlevels; ++level)
      for (auto& kp : allKeypoints[level])
        kp.angle = IC_Angle(mvImagePyramid[level], kp.x, kp.y, umax);
  }

  // ORBextractor.cc:968-1058
  int run(const Plane& image, const int* pvLappingArea, std::vector<KeyPoint>& _keypoints,
          std::vector<uint8_t>& descriptors) {
    if (image.w == 0 || image.h == 0) return -1;
    ComputePyramid(image);
    std::vector<std::vector<KeyPoint>> allKeypoints;
    ComputeKeyPointsOctTree(allKeypoints);
    tapLevelKeys = allKeypoints;
    tapBlurred.assign(nlevels, Plane());
    int nkeypoints = 0;
    for (int level = 0; level < nlevels; ++level) nkeypoints += (int)allKeypoints[level].size();
    descriptors.assign((size_t)nkeypoints * 32, 0);
    _keypoints.clear();
    if (pvLappingArea) _keypoints.resize(nkeypoints);
    int offset = 0;
    int monoIndex = 0, stereoIndex = nkeypoints - 1;
    for (int level = 0; level < nlevels; ++level) {
      std::vector<KeyPoint>& keypoints = allKeypoints[level];
      int nkeypointsLevel = (int)keypoints.size();
      if (nkeypointsLevel == 0) continue;
      Plane workingMat;
      gaussianBlur7(mvImagePyramid[level], workingMat);
      tapBlurred[level] = workingMat;
      std::vector<uint8_t> desc((size_t)nkeypointsLevel * 32);
      for (int i = 0; i < nkeypointsLevel; i++)
        computeOrbDescriptor(keypoints[i], workingMat, &desc[(size_t)i * 32]);
      if (!pvLappingArea)
        memcpy(&descriptors[(size_t)offset * 32], desc.data(), desc.size());
      offset += nkeypointsLevel;
      float scale = mvScaleFactor[level];
      int i = 0;
      for (auto& kp : keypoints) {
        if (level != 0) {
          kp.x *= scale;
          kp.y *= scale;
        }
        if (pvLappingArea) {
          if (kp.x >= pvLappingArea[0] && kp.x <= pvLappingArea[1]) {
            _keypoints.at(stereoIndex) = kp;
            memcpy(&descriptors[(size_t)stereoIndex * 32], &desc[(size_t)i * 32], 32);
            stereoIndex--;
          } else {
            _keypoints.at(monoIndex) = kp;
            memcpy(&descriptors[(size_t)monoIndex * 32], &desc[(size_t)i * 32], 32);
            monoIndex++;
          }
          i++;
        }
      }
      if (!pvLappingArea) _keypoints.insert(_keypoints.end(), keypoints.begin(), keypoints.end());
    }
    return monoIndex;
  }
};

}  // namespace vo

// ---------------------------------------------------------------- C interface (ctypes / bench)
extern "C" {

void* vo_orb_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
  return new vo::Extractor(nfeatures, scaleFactor, nlevels, iniTh, minTh);
}
void vo_orb_destroy(void* h) { delete (vo::Extractor*)h; }

// returns mono index (>=0) or -1 on empty image; *n = number of keypoints (may exceed cap:
// then nothing is copied and -2 is returned)
int vo_orb_extract(void* h, const uint8_t* img, int w, int hgt, int stride, const int* lapping,
                   void* kps_out, uint8_t* desc_out, int cap, int* n) {
  vo::Extractor* e = (vo::Extractor*)h;
  vo::Plane im(w, hgt);
  for (int y = 0; y < hgt; y++) memcpy(im.row(y), img + (size_t)y * stride, w);
  std::vector<vo::KeyPoint> kps;
  std::vector<uint8_t> desc;
  int mono = e->run(im, lapping, kps, desc);
  if (mono < 0) return -1;
  *n = (int)kps.size();
  if ((int)kps.size() > cap) return -2;
  if (!kps.empty()) {
    memcpy(kps_out, kps.data(), kps.size() * sizeof(vo::KeyPoint));
    memcpy(desc_out, desc.data(), desc.size());
  }
  return mono;
}

int vo_orb_features_per_level(void* h, int level) {
  return ((vo::Extractor*)h)->mnFeaturesPerLevel[level];
}
float vo_orb_scale_factor(void* h, int level) { return ((vo::Extractor*)h)->mvScaleFactor[level]; }
int vo_orb_umax(void* h, int v) { return ((vo::Extractor*)h)->umax[v]; }
long vo_orb_tie_count(void* h) { return ((vo::Extractor*)h)->tieCount; }

void vo_orb_level_size(void* h, int level, int* w, int* hgt) {
  vo::Extractor* e = (vo::Extractor*)h;
  *w = e->mvImagePyramid[level].w;
  *hgt = e->mvImagePyramid[level].h;
}
// which: 0 = pyramid level (borderless), 1 = blurred level, 2 = pyramid level with the 19-px
// REFLECT_101 border (dst must be (w+38)*(h+38))
void vo_orb_get_plane(void* h, int level, int which, uint8_t* dst) {
  vo::Extractor* e = (vo::Extractor*)h;
  if (which == 2) {
    vo::Plane b = vo::copyMakeBorder101(e->mvImagePyramid[level], vo::EDGE_THRESHOLD);
    memcpy(dst, b.px.data(), b.px.size());
    return;
  }
  const vo::Plane& p = which == 0 ? e->mvImagePyramid[level] : e->tapBlurred[level];
  memcpy(dst, p.px.data(), p.px.size());
}
// FAST candidates of a level in vToDistributeKeys order: int32 triplets (x, y, response)
int vo_orb_get_candidates(void* h, int level, int32_t* dst, int cap) {
  vo::Extractor* e = (vo::Extractor*)h;
  const auto& c = e->tapCandidates[level];
  if ((int)c.size() > cap) return -(int)c.size();
  for (size_t i = 0; i < c.size(); i++) {
    dst[i * 3] = (int)c[i].x;
    dst[i * 3 + 1] = (int)c[i].y;
    dst[i * 3 + 2] = (int)c[i].response;
  }
  return (int)c.size();
}
// per-level keypoints after the quadtree + orientation, level coordinates (before scaling)
int vo_orb_get_level_keys(void* h, int level, void* dst, int cap) {
  vo::Extractor* e = (vo::Extractor*)h;
  const auto& c = e->tapLevelKeys[level];
  if ((int)c.size() > cap) return -(int)c.size();
  if (!c.empty()) memcpy(dst, c.data(), c.size() * sizeof(vo::KeyPoint));
  return (int)c.size();
}

// stand-alone primitives for known-answer tests
void vo_resize_linear_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
  vo::Plane s(sw, sh), d(dw, dh);
  memcpy(s.px.data(), src, s.px.size());
  vo::resizeLinearU8(s, d);
  memcpy(dst, d.px.data(), d.px.size());
}
void vo_gaussian_blur7(const uint8_t* src, int w, int h, uint8_t* dst) {
  vo::Plane s(w, h), d;
  memcpy(s.px.data(), src, s.px.size());
  vo::gaussianBlur7(s, d);
  memcpy(dst, d.px.data(), d.px.size());
}
int vo_fast(const uint8_t* src, int w, int h, int threshold, int32_t* dst, int cap) {
  vo::Plane s(w, h);
  memcpy(s.px.data(), src, s.px.size());
  std::vector<vo::FastKp> k;
  vo::fast9_16(s, 0, 0, w, h, threshold, k);
  if ((int)k.size() > cap) return -(int)k.size();
  for (size_t i = 0; i < k.size(); i++) {
    dst[i * 3] = k[i].x;
    dst[i * 3 + 1] = k[i].y;
    dst[i * 3 + 2] = k[i].score;
  }
  return (int)k.size();
}
float vo_fast_atan2(float y, float x) { return vo::fastAtan2(y, x); }
int vo_cv_round_f(float v) { return vo::cvRound(v); }
void vo_sincos_ref(float angle_deg, float* c, float* s) {
  float a = angle_deg * vo::factorPI;
  *c = cosf(a);
  *s = sinf(a);
}

}  // extern "C"

// Stand-alone DistributeOctTree (ORBextractor.cc:518-721) on int32 (x, y, response) triplets in
// vToDistributeKeys order; out receives the selected triplets in output order.
extern "C" int vo_distribute_octtree(const int32_t* xyr, int K, int minX, int maxX, int minY,
                                     int maxY, int N, int32_t* out, int cap) {
  vo::Extractor e(1000, 1.2f, 8, 20, 7);
  std::vector<vo::NodeKey> keys(K);
  for (int i = 0; i < K; i++) {
    keys[i].x = (float)xyr[i * 3];
    keys[i].y = (float)xyr[i * 3 + 1];
    keys[i].response = (float)xyr[i * 3 + 2];
    keys[i].idx = i;
  }
  std::vector<vo::NodeKey> r = e.DistributeOctTree(keys, minX, maxX, minY, maxY, N);
  if ((int)r.size() > cap) return -(int)r.size();
  for (size_t i = 0; i < r.size(); i++) {
    out[i * 3] = (int)r[i].x;
    out[i * 3 + 1] = (int)r[i].y;
    out[i * 3 + 2] = (int)r[i].response;
  }
  return (int)r.size();
}
