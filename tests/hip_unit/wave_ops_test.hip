// Stand-alone check of vieo_slam_amd/csrc/wave_ops.h (built and run by tests/test_wave_ops.py on the GPU box):
// the DPP reductions against sequential host sums, for several seeds; exit code = number of mismatching lanes.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>

#include "../../vieo_slam_amd/csrc/wave_ops.h"

__global__ void k(const int* a, const double* d, int* oi, unsigned* om, double* od, double* oq, double* omx) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  oi[i] = vieo::wave_sum_i32(a[i]);
  om[i] = vieo::wave_min_u32((unsigned)a[i]);
  od[i] = vieo::wave_sum_f64(d[i]);
  oq[i] = vieo::quad_sum_f64(d[i]);
  omx[i] = vieo::wave_max_f64(d[i]);
}

int main() {
  const int W = 16, N = 64 * W;
  int ha[N];
  double hd[N];
  unsigned s = 12345u;
  for (int i = 0; i < N; i++) {
    s = s * 1664525u + 1013904223u;
    ha[i] = (int)(s >> 8) % 100000 - 30000;
    s = s * 1664525u + 1013904223u;
    hd[i] = ((double)(s >> 4) / 268435456.0 - 0.5) * 1e3;
  }
  int *a, *oi;
  unsigned* om;
  double *d, *od, *oq, *omx;
  if (hipMalloc(&a, N * 4) || hipMalloc(&oi, N * 4) || hipMalloc(&om, N * 4) || hipMalloc(&d, N * 8) || hipMalloc(&od, N * 8) ||
      hipMalloc(&oq, N * 8) || hipMalloc(&omx, N * 8))
    return 200;
  (void)hipMemcpy(a, ha, N * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d, hd, N * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(W), dim3(64), 0, 0, a, d, oi, om, od, oq, omx);
  static int hoi[N];
  static unsigned hom[N];
  static double hod[N], hoq[N], homx[N];
  if (hipMemcpy(hoi, oi, N * 4, hipMemcpyDeviceToHost) || hipMemcpy(hom, om, N * 4, hipMemcpyDeviceToHost) ||
      hipMemcpy(hod, od, N * 8, hipMemcpyDeviceToHost) || hipMemcpy(hoq, oq, N * 8, hipMemcpyDeviceToHost) ||
      hipMemcpy(homx, omx, N * 8, hipMemcpyDeviceToHost))
    return 201;
  int bad = 0;
  for (int w = 0; w < W; w++) {
    int si = 0;
    unsigned mn = ~0u;
    double sd = 0, mx = -1e300, sabs = 0;
    for (int i = 64 * w; i < 64 * w + 64; i++) {
      si += ha[i];
      if ((unsigned)ha[i] < mn) mn = (unsigned)ha[i];
      sd += hd[i], sabs += fabs(hd[i]);
      if (hd[i] > mx) mx = hd[i];
    }
    for (int i = 64 * w; i < 64 * w + 64; i++) {
      const int q0 = i & ~3;
      const double q = (hd[q0] + hd[q0 + 1]) + (hd[q0 + 2] + hd[q0 + 3]);  // the helper's own association: bit-equal
      if (hoi[i] != si || hom[i] != mn || fabs(hod[i] - sd) > 1e-13 * sabs || homx[i] != mx || hoq[i] != q) bad++;
      if (hod[i] != hod[64 * w]) bad++;  // uniform over the wavefront
    }
  }
  printf("wave_ops: %d lanes checked, %d bad\n", N, bad);
  return bad > 199 ? 199 : bad;
}
