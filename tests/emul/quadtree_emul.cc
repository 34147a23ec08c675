// Test-only host build of vieo_slam_amd/csrc/quadtree.inl (the phase-structured text of the HIP
// quadtree kernel) with QT_PHASE = loop over 256 logical threads.  Lets the CPU test-suite check
// the data-parallel formulation against the oracle's literal std::list version.  Never linked
// into the product library.
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#include "../../vieo_slam_amd/csrc/quadtree.inl"

extern "C" int emul_distribute(const int32_t* xyr, int K, int minX, int maxX, int minY, int maxY,
                               int N, int32_t* out, int cap) {
  const int regW = maxX - minX, regH = maxY - minY;
  const int nIni = (int)std::round((float)regW / (float)regH);
  if (nIni < 1) return -1000000;
  const float hX = (float)regW / (float)nIni;
  const int ncap = (N > 4 * nIni ? N : 4 * nIni) + 8;
  const int scap = 2 * ncap;
  std::vector<unsigned> keys(K > 0 ? K : 1);
  std::vector<unsigned short> kslot(K > 0 ? K : 1);
  std::vector<unsigned char> kq(K > 0 ? K : 1);
  QtShared sh;
  memset(&sh, 0, sizeof(sh));
  std::vector<short> x0(ncap), y0(ncap), x1(ncap), y1(ncap);
  std::vector<int> cnt(ncap, 0), cc(ncap * 4, 0), cs0(ncap), cs1(ncap);
  std::vector<unsigned short> child(ncap * 4), mark(ncap), l0(ncap), l1(ncap), c0(ncap), c1(ncap),
      order(ncap);
  std::vector<unsigned> best(ncap);
  std::vector<unsigned long long> sa(scap), sb(scap);
  std::vector<unsigned char> flag(ncap);
  QtMem m;
  m.s = &sh;
  m.x0 = x0.data(), m.y0 = y0.data(), m.x1 = x1.data(), m.y1 = y1.data();
  m.cnt = cnt.data(), m.cc = cc.data(), m.child = child.data(), m.mark = mark.data();
  m.best = best.data();
  m.list[0] = l0.data(), m.list[1] = l1.data();
  m.scanA = sa.data(), m.scanB = sb.data(), m.flag = flag.data();
  m.cand_slot[0] = c0.data(), m.cand_slot[1] = c1.data();
  m.cand_size[0] = cs0.data(), m.cand_size[1] = cs1.data();
  m.order = order.data();
  m.ncap = ncap, m.scap = scap;
  sh.K = K;
  for (int k = 0; k < K; k++) {
    keys[k] = (unsigned)xyr[k * 3] | ((unsigned)xyr[k * 3 + 1] << 12) | ((unsigned)xyr[k * 3 + 2] << 24);
    int ini = (int)((float)xyr[k * 3] / hX);
    kslot[k] = (unsigned short)ini;
    cnt[ini]++;
  }
  std::vector<unsigned> o(ncap);
  int n = qt_distribute(m, keys.data(), kslot.data(), kq.data(), regW, regH, nIni, hX, N, o.data());
  if (sh.error) return -2000000;
  if (n > cap) return -n;
  for (int i = 0; i < n; i++) {
    out[i * 3] = QT_KEY_X(o[i]);
    out[i * 3 + 1] = QT_KEY_Y(o[i]);
    out[i * 3 + 2] = QT_KEY_R(o[i]);
  }
  return n;
}
