// Host-side group tables of camm::GeometricCamera::FillMatchesFromPair (reference
// common/camera_models/camera_base.h:408-574, compiled with USE_STRATEGY_MIN_DIST: common/config.h:12), shared by
// ComputeStereoFishEyeMatches (fisheye_stereo.hip) and SearchForTriangulation (tri_search.hip).  The bookkeeping is
// order dependent (a later pair can take a key from an earlier group), so it runs on the host in the reference's
// order over verdicts the device computed for every pair at once.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace vieo {

// ---- host: the group tables of FillMatchesFromPair (camera_base.h:408-574, USE_STRATEGY_MIN_DIST) ----------
struct FeGroups {
  int nc = 0;
  std::vector<int32_t> idxs;   // [g][nc], -1: none          (mvidxsMatches)
  std::vector<float> last;     // [g][nc], INFINITY: none    (lastdists)
  std::vector<uint8_t> good;   //                            (goodmatches_)
  std::vector<double> p3d;     // [g][3]                     (v3dpoints_)
  std::vector<int32_t> key2g[8];  // (camera, key) -> group, -1: none   (mapcamidx2idxs_)
  void reset(int n_cams, const int32_t* n_keys) {
    nc = n_cams;
    idxs.clear(), last.clear(), good.clear(), p3d.clear();
    for (int c = 0; c < n_cams; ++c) key2g[c].assign(n_keys[c], -1);
  }
  int size() const { return (int)good.size(); }
};

// one ratio-accepted knn row; `tri_ok`: the triangulation verdict under the current threshold
static inline bool fe_fill(FeGroups& G, int cami, int idxi, int camj, int idxj, float dist, bool tri_ok, const double* p3d) {
  const int nc = G.nc;
  int gi = G.key2g[cami][idxi];
  const int gj = G.key2g[camj][idxj];
  if (gi < 0 && gj >= 0) gi = gj;  // iteri = iterj
  int check0 = 0, check1 = 0, contradict = 0, g = -1;
  const int g_first = gi;  // iteri->second
  if (gi >= 0) {
    g = gi;
    contradict = (gj >= 0 && gj != g) ? 2 : 0;
    if (contradict) {  // keep the group whose members were matched at the smaller mean distance
      float sum[2] = {0, 0};
      int cnt[2] = {0, 0};
      for (int t = 0; t < nc; ++t) {
        if (G.idxs[(size_t)g * nc + t] >= 0) sum[0] += G.last[(size_t)g * nc + t], ++cnt[0];
        if (G.idxs[(size_t)gj * nc + t] >= 0) sum[1] += G.last[(size_t)gj * nc + t], ++cnt[1];
      }
      if (sum[1] * cnt[0] < sum[0] * cnt[1]) g = gj, contradict = 1;
    }
    const int32_t* ix = &G.idxs[(size_t)g * nc];
    const float* ld = &G.last[(size_t)g * nc];
    if (ix[cami] < 0 || (idxi != ix[cami] && ld[cami] > dist)) check0 = 2;
    if (ix[camj] < 0 || (idxj != ix[camj] && ld[camj] > dist)) check1 = 2;
  } else
    check0 = check1 = 1;
  if (!(check0 || check1)) return false;
  if (!tri_ok) return false;
  if (check0 == 1) {
    g = G.size();
    G.idxs.insert(G.idxs.end(), nc, -1);
    G.last.insert(G.last.end(), nc, INFINITY);
    G.idxs[(size_t)g * nc + cami] = idxi, G.idxs[(size_t)g * nc + camj] = idxj;
    G.last[(size_t)g * nc + cami] = dist, G.last[(size_t)g * nc + camj] = dist;
    // map::emplace keeps an existing entry; neither key has one here
    G.key2g[cami][idxi] = g, G.key2g[camj][idxj] = g;
    G.good.push_back(1);
    G.p3d.resize((size_t)G.size() * 3);
  } else {
    if (contradict) {
      const int gc = contradict == 1 ? g_first : gj;
      int32_t* ix = &G.idxs[(size_t)gc * nc];
      if (idxi == ix[cami]) G.key2g[cami][idxi] = -1, G.last[(size_t)gc * nc + cami] = INFINITY, ix[cami] = -1;
      if (idxj == ix[camj]) G.key2g[camj][idxj] = -1, G.last[(size_t)gc * nc + camj] = INFINITY, ix[camj] = -1;
    }
    int32_t* ix = &G.idxs[(size_t)g * nc];
    float* ld = &G.last[(size_t)g * nc];
    if (check0 == 2) {
      if (idxi != ix[cami]) {
        if (ix[cami] >= 0) G.key2g[cami][ix[cami]] = -1;
        if (G.key2g[cami][idxi] < 0) G.key2g[cami][idxi] = g;  // emplace
        ix[cami] = idxi;
      }
      ld[cami] = dist;
    } else if (ld[cami] > dist)
      ld[cami] = dist;
    if (check1 == 2) {
      if (idxj != ix[camj]) {
        if (ix[camj] >= 0) G.key2g[camj][ix[camj]] = -1;
        if (G.key2g[camj][idxj] < 0) G.key2g[camj][idxj] = g;
        ix[camj] = idxj;
      }
      ld[camj] = dist;
    } else if (ld[camj] > dist)
      ld[camj] = dist;
  }
  if (p3d) memcpy(&G.p3d[(size_t)g * 3], p3d, 24);
  return true;
}

}  // namespace vieo
