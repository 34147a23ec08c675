// sincosf_exact.h -- float sin/cos with the exact arithmetic of glibc >= 2.28 sinf/cosf
// (the algorithm of ARM optimized-routines `sincosf`: double-precision range reduction by pi/2
// and degree-8/7 minimax polynomials, evaluated in the published operation order).
//
// Why: the reference computes the BRIEF steering vector as cosf/sinf of a float angle
// (src/ORBextractor.cc:84-85) and feeds it to cvRound(x*b + y*a).  libm's cosf is NOT correctly
// rounded (~0.56 ULP), so a "better" device cosine would flip descriptor bits against the CPU
// path.  Re-doing libm's own double-precision sequence on the device (IEEE double add/mul are
// exact-rounded on gfx950, contraction disabled with -ffp-contract=off) reproduces its results
// bit for bit; tests/test_sincos.py checks that against the host libm.
//
// Valid for |y| < 120 (the extractor only passes angles in [0, 2*pi]).
#ifndef VIEO_SINCOSF_EXACT_H
#define VIEO_SINCOSF_EXACT_H
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define VIEO_HD __host__ __device__ __forceinline__
#else
#define VIEO_HD static inline
#endif

#ifndef VIEO_SINCOS_FMA
#define VIEO_SINCOS_FMA 0
#endif

VIEO_HD double vieo_mad_(double a, double b, double c) {
#if VIEO_SINCOS_FMA
  return __builtin_fma(a, b, c);
#else
  return a * b + c;
#endif
}

VIEO_HD uint32_t vieo_abstop12_(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  return (u >> 20) & 0x7ff;
}

// n&1 == 0: sine polynomial, else cosine polynomial; neg selects the negated-cosine table.
VIEO_HD float vieo_sinf_poly_(double x, double x2, int n, int neg) {
  const double C0 = neg ? -0x1p0 : 0x1p0;
  const double C1 = neg ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2;
  const double C2 = neg ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5;
  const double C3 = neg ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10;
  const double C4 = neg ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
  const double S1 = -0x1.555545995a603p-3;
  const double S2 = 0x1.1107605230bc4p-7;
  const double S3 = -0x1.994eb3774cf24p-13;
  if ((n & 1) == 0) {
    double x3 = x * x2;
    double s1 = vieo_mad_(x2, S3, S2);
    double x7 = x3 * x2;
    double s = vieo_mad_(x3, S1, x);
    return (float)vieo_mad_(x7, s1, s);
  } else {
    double x4 = x2 * x2;
    double c2 = vieo_mad_(x2, C4, C3);
    double c1 = vieo_mad_(x2, C1, C0);
    double x6 = x4 * x2;
    double c = vieo_mad_(x4, C2, c1);
    return (float)vieo_mad_(x6, c2, c);
  }
}

VIEO_HD void vieo_sincosf_exact(float y, float* sinp, float* cosp) {
  const double hpi_inv = 0x1.45F306DC9C883p+23;  // 2/pi * 2^24
  const double hpi = 0x1.921FB54442D18p0;
  double x = y;
  if (vieo_abstop12_(y) < vieo_abstop12_(0x1.921FB6p-1f)) {  // |y| < pi/4
    double x2 = x * x;
    if (vieo_abstop12_(y) < vieo_abstop12_(0x1p-12f)) {
      *sinp = y;
      *cosp = 1.0f;
      return;
    }
    *sinp = vieo_sinf_poly_(x, x2, 0, 0);
    *cosp = vieo_sinf_poly_(x, x2, 1, 0);
    return;
  }
  double r = x * hpi_inv;
  int n = ((int32_t)r + 0x800000) >> 24;
  x = x - n * hpi;  // reduce_fast: n*hpi is a separate (rounded) product in the non-FMA build
  const double sgn = (n & 3) == 1 || (n & 3) == 2 ? -1.0 : 1.0;  // sign[] = {1,-1,-1,1}
  const int neg = (n & 2) ? 1 : 0;
  const double x2 = x * x;
  // sinf: sinf_poly(x*s, x2, p, n);  cosf: s = sign[(n) & 3] with n for cos = n, poly n^1
  *sinp = vieo_sinf_poly_(x * sgn, x2, n, neg);
  // cosf uses p->sign[n & 3] as well, table switched on (n & 2), polynomial index n ^ 1
  *cosp = vieo_sinf_poly_(x * sgn, x2, n ^ 1, neg);
}
#endif
