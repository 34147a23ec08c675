// call_overhead.hip -- what one synchronous "small call" of the drop-in path costs around its kernel on this box:
// N bytes up, a kernel of T microseconds, M bytes back, one synchronisation.  Variants:
//   A  hipMemcpyAsync (pinned) up, kernel, hipMemcpyAsync (pinned) back, hipStreamSynchronize      [round-4 form]
//   B  the kernel reads the pinned host block and writes the pinned host block itself (zero copy)
//   C  a copy kernel up, the kernel, a copy kernel back (no SDMA engine in the chain)
//   D  as A on the NULL stream
//   E  as B, but completion is polled by the host on a flag in the pinned block (no hipStreamSynchronize)
// hipcc --offload-arch=gfx950 -O3 tools/micro/call_overhead.hip -o tools/micro/call_overhead && tools/micro/call_overhead
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
      std::exit(1);                                                            \
    }                                                                          \
  } while (0)

__global__ void k_work(const unsigned* __restrict__ in, int n_in, unsigned* __restrict__ out, int n_out, long long spin_cycles,
                       volatile unsigned* flag, unsigned ticket) {
  unsigned acc = 0;
  for (int i = threadIdx.x; i < n_in; i += blockDim.x) acc += in[i];
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < spin_cycles) __builtin_amdgcn_s_sleep(4);
  for (int i = threadIdx.x; i < n_out; i += blockDim.x) out[i] = acc + i;
  if (flag) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) *flag = ticket;
  }
}
__global__ void k_copy(const unsigned* __restrict__ src, unsigned* __restrict__ dst, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

int main(int argc, char** argv) {
  const int up = argc > 1 ? atoi(argv[1]) : 20 * 1024, down = argc > 2 ? atoi(argv[2]) : 2 * 1024;
  const int iters = 2000;
  unsigned *h_in, *h_out, *d_in, *d_out;
  CK(hipHostMalloc((void**)&h_in, up, hipHostMallocDefault));
  CK(hipHostMalloc((void**)&h_out, down + 64, hipHostMallocDefault));
  CK(hipMalloc((void**)&d_in, up));
  CK(hipMalloc((void**)&d_out, down));
  memset(h_in, 1, up);
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  volatile unsigned* h_flag = h_out + down / 4;
  for (long long spin_us : {0LL, 300LL}) {
    const long long cyc = spin_us * 100;  // s_memtime ticks at 100 MHz
    auto run = [&](const char* name, auto body) {
      for (int i = 0; i < 50; i++) body(i + 1);
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < iters; i++) body(100 + i);
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
      std::printf("kernel %3lld us  %-62s %8.2f us per call  (overhead %.2f)\n", spin_us, name, us, us - spin_us);
    };
    run("A memcpyAsync up + kernel + memcpyAsync back + sync", [&](int) {
      CK(hipMemcpyAsync(d_in, h_in, up, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, st, d_in, up / 4, d_out, down / 4, cyc, nullptr, 0u);
      CK(hipMemcpyAsync(h_out, d_out, down, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
    });
    run("B kernel reads / writes the pinned blocks (zero copy) + sync", [&](int) {
      hipLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, st, h_in, up / 4, h_out, down / 4, cyc, nullptr, 0u);
      CK(hipStreamSynchronize(st));
    });
    run("C copy kernel + kernel + copy kernel + sync", [&](int) {
      hipLaunchKernelGGL(k_copy, dim3(8), dim3(256), 0, st, h_in, d_in, up / 4);
      hipLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, st, d_in, up / 4, d_out, down / 4, cyc, nullptr, 0u);
      hipLaunchKernelGGL(k_copy, dim3(2), dim3(256), 0, st, d_out, h_out, down / 4);
      CK(hipStreamSynchronize(st));
    });
    run("D as A on the null stream", [&](int) {
      CK(hipMemcpyAsync(d_in, h_in, up, hipMemcpyHostToDevice, nullptr));
      hipLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, nullptr, d_in, up / 4, d_out, down / 4, cyc, nullptr, 0u);
      CK(hipMemcpyAsync(h_out, d_out, down, hipMemcpyDeviceToHost, nullptr));
      CK(hipStreamSynchronize(nullptr));
    });
    run("E zero copy, the host polls a flag in the pinned block", [&](int i) {
      hipLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, st, h_in, up / 4, h_out, down / 4, cyc, (volatile unsigned*)h_flag, (unsigned)i);
      while (*h_flag != (unsigned)i) {
      }
    });
    run("F memcpyAsync up + kernel writes pinned + sync", [&](int) {
      CK(hipMemcpyAsync(d_in, h_in, up, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, st, d_in, up / 4, h_out, down / 4, cyc, nullptr, 0u);
      CK(hipStreamSynchronize(st));
    });
  }
  CK(hipStreamSynchronize(st));
  return 0;
}
