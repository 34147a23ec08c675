// Dependent-chain latencies on gfx950, one wavefront, s_memtime around 512 links of each chain.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lat_bench.hip -o tools/micro/lat_bench
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double readlane_d(double v, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
#define N 512
__global__ void k(double* out, unsigned long long* t, double seed, int which) {
  __shared__ double s[64];
  const int lane = threadIdx.x;
  double y = seed + lane * 1e-3, m = 0.999 + lane * 1e-9;
  float yf = (float)y, mf = (float)m;
  s[lane] = y;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (which == 0) {
#pragma unroll
    for (int i = 0; i < N; i++) y = __builtin_fma(y, m, 1e-9);
  } else if (which == 1) {
#pragma unroll
    for (int i = 0; i < N; i++) y = __builtin_fma(-m, readlane_d(y, i & 31), y);
  } else if (which == 2) {
#pragma unroll
    for (int i = 0; i < N; i++) yf = __builtin_fmaf(yf, mf, 1e-9f);
  } else if (which == 3) {
#pragma unroll
    for (int i = 0; i < N; i++) yf = __builtin_fmaf(-mf, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yf), i & 31)), yf);
  } else if (which == 4) {  // LDS: write own, read a neighbour's (wave-uniform address)
#pragma unroll
    for (int i = 0; i < N; i++) {
      s[lane] = y;
      __builtin_amdgcn_wave_barrier();
      y = __builtin_fma(-m, s[i & 31], y);
      __builtin_amdgcn_wave_barrier();
    }
  } else if (which == 5) {  // DPP row_shr:1 within rows of 16
#pragma unroll
    for (int i = 0; i < N; i++) {
      int lo = __builtin_amdgcn_update_dpp(0, __double2loint(y), 0x111, 0xf, 0xf, false);
      int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(y), 0x111, 0xf, 0xf, false);
      y = __builtin_fma(-m, __hiloint2double(hi, lo), y);
    }
  } else if (which == 6) {  // two independent fma per readlane pair (issue-bound reference)
    double z = y + 1;
#pragma unroll
    for (int i = 0; i < N; i++) y = __builtin_fma(y, m, 1e-9), z = __builtin_fma(z, m, 1e-9);
    y += z;
  } else if (which == 7) {  // v_rcp_f64 chain
#pragma unroll
    for (int i = 0; i < N; i++) y = __builtin_amdgcn_rcp(y);
  } else if (which == 8) {
#pragma unroll 8
    for (int i = 0; i < N; i++) y = 1.0 / (y + 0.5);
  } else if (which == 9) {
#pragma unroll 8
    for (int i = 0; i < N; i++) y = sqrt(y + 0.5);
  } else if (which == 10) {
#pragma unroll 4
    for (int i = 0; i < N; i++) {
      double sn, cs;
      sincos(y * 0.3, &sn, &cs);
      y = sn + cs;
    }
  } else if (which == 11) {
#pragma unroll 4
    for (int i = 0; i < N; i++) y = atan(y) + 0.7;
  } else if (which == 12) {
#pragma unroll 4
    for (int i = 0; i < N; i++) y = rsqrt(y + 0.5);
  } else if (which == 13) {
#pragma unroll 4
    for (int i = 0; i < N; i++) y = sin(y * 0.3) + 0.9;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[lane] = y + yf;
  if (lane == 0) t[which] = t1 - t0;
}
int main() {
  double* out;
  unsigned long long *t, h[16];
  hipMalloc(&out, 512), hipMalloc(&t, 128);
  const char* nm[] = {"v_fma_f64 -> v_fma_f64", "v_fma_f64 -> 2 v_readlane -> v_fma_f64", "v_fma_f32 -> v_fma_f32",
                      "v_fma_f32 -> v_readlane -> v_fma_f32", "v_fma_f64 -> ds_write, ds_read -> v_fma_f64",
                      "v_fma_f64 -> 2 dpp mov -> v_fma_f64", "2 independent v_fma_f64 chains (per pair)", "v_rcp_f64 -> v_rcp_f64", "1.0 / x", "sqrt(x)", "sincos(x)", "atan(x)", "rsqrt(x)", "sin(x)"};
  for (int w = 0; w < 14; w++)
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k, 1, 64, 0, 0, out, t, 1.0, w);
  hipDeviceSynchronize();
  hipMemcpy(h, t, 128, hipMemcpyDeviceToHost);
  for (int w = 0; w < 14; w++) printf("%-46s %6.1f cycles per link\n", nm[w], (double)h[w] / N);
  return 0;
}
