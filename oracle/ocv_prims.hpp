// ORACLE -- TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is shipped or measured as the product.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
//
// CPU restatement of the OpenCV 4.5 image primitives that the reference extractor calls
// (reference: src/ORBextractor.cc:39-41 includes; call sites :60,79,87,91,763,767,1013-1015,
// 1063,1070-1077).  OpenCV itself is a third-party dependency that is NOT vendored in
// /root/reference (CMakeLists.txt:64-65 `find_package(OpenCV 4.5)`), and it is not installed in
// this image, so these functions restate OpenCV's *published* algorithms:
//   - cvRound / cvFloor / cvCeil                      (core/fast_math.hpp)
//   - copyMakeBorder BORDER_REFLECT_101 index rule    (core/copy.cpp borderInterpolate)
//   - resize INTER_LINEAR, CV_8UC1 fixed point path   (imgproc/resize.cpp: HResizeLinear /
//                                                      VResizeLinear<uchar,int,short,...>)
//   - FAST-9/16 + cornerScore<16> + 3x3 NMS           (features2d/fast.cpp, fast_score.cpp)
//   - GaussianBlur 8U fixed-point (Q8.8) path         (imgproc/smooth.dispatch.cpp,
//                                                      getGaussianKernelFixedPoint_ED)
//   - fastAtan2                                       (core/mathfuncs.cpp atan_f32)
// PARITY UNPINNED: the reference holds no golden vectors for this path and OpenCV
// cannot be run here, so "bit-exact" means bit-exact against this restatement.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace vo {

// core/fast_math.hpp: cvRound = round-half-to-even (SSE cvtsd2si / lrint in default rounding mode)
static inline int cvRound(double v) { return (int)std::lrint(v); }
static inline int cvRound(float v) { return (int)std::lrintf(v); }
static inline int cvFloor(double v) {
  int i = (int)v;
  return i - (i > v);
}
static inline int cvCeil(double v) {
  int i = (int)v;
  return i + (i < v);
}

// borderInterpolate(p, len, BORDER_REFLECT_101): -k -> k, len-1+k -> len-1-k
static inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0)
      p = -p;
    else
      p = 2 * (len - 1) - p;
  }
  return p;
}

struct Plane {  // borderless 8-bit image, row-major
  int w = 0, h = 0;
  std::vector<uint8_t> px;
  Plane() {}
  Plane(int w_, int h_) : w(w_), h(h_), px((size_t)w_ * h_) {}
  uint8_t* row(int y) { return px.data() + (size_t)y * w; }
  const uint8_t* row(int y) const { return px.data() + (size_t)y * w; }
};

// resize(src, dst, Size(dw,dh), 0, 0, INTER_LINEAR) for CV_8UC1.
// INTER_RESIZE_COEF_BITS = 11; horizontal pass keeps int32 (value*2048), vertical pass is the
// uchar specialisation  (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.
static inline void resizeLinearU8(const Plane& src, Plane& dst) {
  const int sw = src.w, sh = src.h, dw = dst.w, dh = dst.h;
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cvFloor(fx);
    fx -= sx;
    if (sx < 0) fx = 0, sx = 0;
    if (sx >= sw - 1) fx = 0, sx = sw - 1;
    xofs[dx] = sx;
    ialpha[dx * 2] = (short)cvRound((1.f - fx) * 2048);
    ialpha[dx * 2 + 1] = (short)cvRound(fx * 2048);
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cvFloor(fy);
    fy -= sy;
    yofs[dy] = sy;
    ibeta[dy * 2] = (short)cvRound((1.f - fy) * 2048);
    ibeta[dy * 2 + 1] = (short)cvRound(fy * 2048);
  }
  auto clip = [](int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; };
  std::vector<int> r0(dw), r1(dw);
  auto hline = [&](int sy, std::vector<int>& out) {
    const uint8_t* S = src.row(sy);
    for (int dx = 0; dx < dw; dx++) {
      int sx = xofs[dx];
      if (sx + 1 < sw)
        out[dx] = S[sx] * ialpha[dx * 2] + S[sx + 1] * ialpha[dx * 2 + 1];
      else
        out[dx] = S[sx] * 2048;
    }
  };
  for (int dy = 0; dy < dh; dy++) {
    int sy0 = clip(yofs[dy], 0, sh), sy1 = clip(yofs[dy] + 1, 0, sh);
    hline(sy0, r0);
    hline(sy1, r1);
    int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
    uint8_t* D = dst.row(dy);
    for (int x = 0; x < dw; x++)
      D[x] = (uint8_t)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
  }
}

// Bordered copy: dst is (w+2b)x(h+2b), REFLECT_101.
static inline Plane copyMakeBorder101(const Plane& src, int b) {
  Plane d(src.w + 2 * b, src.h + 2 * b);
  for (int y = 0; y < d.h; y++) {
    const uint8_t* S = src.row(reflect101(y - b, src.h));
    uint8_t* D = d.row(y);
    for (int x = 0; x < d.w; x++) D[x] = S[reflect101(x - b, src.w)];
  }
  return d;
}

struct FastKp {
  int x, y, score;
};

// FAST_t<16>(img, kps, threshold, nonmax=true) on a sub-rectangle [x0,x1) x [y0,y1) of `im`
// treated as a stand-alone image (the reference passes rowRange/colRange views,
// ORBextractor.cc:763).  Output coordinates are relative to (x0,y0), row-major order.
static inline void fast9_16(const Plane& im, int x0, int y0, int x1, int y1, int threshold,
                            std::vector<FastKp>& out) {
  static const int off[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1},
                                 {2, -2}, {1, -3},  {0, -3},  {-1, -3}, {-2, -2}, {-3, -1},
                                 {-3, 0}, {-3, 1},  {-2, 2},  {-1, 3}};
  out.clear();
  const int cols = x1 - x0, rows = y1 - y0;
  if (cols < 7 || rows < 7) return;
  const int K = 8, N = 25;
  threshold = std::min(std::max(threshold, 0), 255);
  std::vector<uint8_t> score((size_t)cols * rows, 0);  // 0 = not a corner
  std::vector<uint8_t> iscorner((size_t)cols * rows, 0);
  auto P = [&](int x, int y) -> int { return im.row(y0 + y)[x0 + x]; };
  for (int i = 3; i < rows - 3; i++) {
    for (int j = 3; j < cols - 3; j++) {
      int v = P(j, i);
      int ring[N];
      for (int k = 0; k < N; k++) ring[k] = P(j + off[k & 15][0], i + off[k & 15][1]);
      bool corner = false;
      {
        int vt = v - threshold, count = 0;
        for (int k = 0; k < N; k++) {
          if (ring[k] < vt) {
            if (++count > K) {
              corner = true;
              break;
            }
          } else
            count = 0;
        }
      }
      if (!corner) {
        int vt = v + threshold, count = 0;
        for (int k = 0; k < N; k++) {
          if (ring[k] > vt) {
            if (++count > K) {
              corner = true;
              break;
            }
          } else
            count = 0;
        }
      }
      if (!corner) continue;
      // cornerScore<16>: (fast_score.cpp) the early `continue`s there are pure pruning.
      int d[N];
      for (int k = 0; k < N; k++) d[k] = v - ring[k];
      int a0 = threshold;
      for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, d[k + 4]);
        a = std::min(a, d[k + 5]);
        a = std::min(a, d[k + 6]);
        a = std::min(a, d[k + 7]);
        a = std::min(a, d[k + 8]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
      }
      int b0 = -a0;
      for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        b = std::max(b, d[k + 3]);
        b = std::max(b, d[k + 4]);
        b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, d[k + 6]);
        b = std::max(b, d[k + 7]);
        b = std::max(b, d[k + 8]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
      }
      iscorner[(size_t)i * cols + j] = 1;
      score[(size_t)i * cols + j] = (uint8_t)(-b0 - 1);
    }
  }
  for (int i = 3; i < rows - 3; i++)
    for (int j = 3; j < cols - 3; j++) {
      if (!iscorner[(size_t)i * cols + j]) continue;
      int s = score[(size_t)i * cols + j];
      auto S = [&](int x, int y) -> int { return score[(size_t)y * cols + x]; };
      if (s > S(j + 1, i) && s > S(j - 1, i) && s > S(j - 1, i - 1) && s > S(j, i - 1) &&
          s > S(j + 1, i - 1) && s > S(j - 1, i + 1) && s > S(j, i + 1) && s > S(j + 1, i + 1))
        out.push_back({j, i, s});
    }
}

// Q8.8 kernel for GaussianBlur(Size(7,7), sigma=2): getGaussianKernelFixedPoint_ED applied to
// exp(-x^2/8)/sum  (x256: 17.96, 33.56, 48.82, 55.32) with error diffusion, centre = 256 - rest.
// PROVISIONAL TABLE (kept in one place): an OpenCV older than the error-diffusion change would
// give {18,34,49,55,...} (sum 257).
static const int kGauss7Q8[7] = {18, 34, 48, 56, 48, 34, 18};

// GaussianBlur(src, dst, Size(7,7), 2, 2, BORDER_REFLECT_101) for CV_8UC1, fixed-point path:
// horizontal pass exact in u16 (Q8.8), vertical pass Q16.16 rounded: (sum + 2^15) >> 16.
static inline void gaussianBlur7(const Plane& src, Plane& dst) {
  const int w = src.w, h = src.h;
  std::vector<uint16_t> H((size_t)w * h);
  for (int y = 0; y < h; y++) {
    const uint8_t* S = src.row(y);
    for (int x = 0; x < w; x++) {
      uint32_t acc = 0;
      for (int k = 0; k < 7; k++) acc += kGauss7Q8[k] * S[reflect101(x + k - 3, w)];
      H[(size_t)y * w + x] = (uint16_t)acc;
    }
  }
  dst = Plane(w, h);
  for (int y = 0; y < h; y++) {
    uint8_t* D = dst.row(y);
    for (int x = 0; x < w; x++) {
      uint32_t acc = 0;
      for (int k = 0; k < 7; k++) acc += kGauss7Q8[k] * H[(size_t)reflect101(y + k - 3, h) * w + x];
      D[x] = (uint8_t)((acc + (1u << 15)) >> 16);
    }
  }
}

// cv::fastAtan2(y, x) in degrees; float arithmetic, no FMA contraction (build with
// -ffp-contract=off).
static inline float fastAtan2(float y, float x) {
  static const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
  static const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
  static const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
  static const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
  const float eps = (float)2.2204460492503131e-16;
  float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + eps);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + eps);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

}  // namespace vo
