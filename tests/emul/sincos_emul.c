/* Test-only host build of vieo_slam_amd/csrc/sincosf_exact.h: compares the device routine's
 * arithmetic (compiled for the host, no FMA contraction) with the host libm's sinf/cosf, which is
 * what the reference calls (src/ORBextractor.cc:84-85). */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../vieo_slam_amd/csrc/sincosf_exact.h"

void emul_sincosf(float y, float* s, float* c) { vieo_sincosf_exact(y, s, c); }

/* counts mismatches against libm over bit patterns [u0, u1] with the given stride */
long emul_sincosf_sweep(uint32_t u0, uint32_t u1, uint32_t stride, long* n_checked) {
  long bad = 0, n = 0;
  for (uint64_t u = u0; u <= u1; u += stride) {
    uint32_t uu = (uint32_t)u;
    float y, s, c;
    memcpy(&y, &uu, 4);
    vieo_sincosf_exact(y, &s, &c);
    float rs = sinf(y), rc = cosf(y);
    bad += memcmp(&s, &rs, 4) != 0;
    bad += memcmp(&c, &rc, 4) != 0;
    n++;
  }
  *n_checked = n;
  return bad;
}
