// copy_bench.hip -- host <-> device copies of the sizes a tracked frame moves (an image 361 KB, a frame's upload block
// ~0.9 MB, keys + descriptors 76 KB, small result blocks), as the runtime does them and as a copy KERNEL that reads /
// writes the pinned block directly with many wavefronts in flight.  Time = call + hipStreamSynchronize, microseconds.
// hipcc --offload-arch=gfx950 -O3 tools/micro/copy_bench.hip -o tools/micro/copy_bench
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                       \
  do {                                                              \
    hipError_t e_ = (x);                                            \
    if (e_ != hipSuccess) {                                         \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
      std::exit(1);                                                 \
    }                                                               \
  } while (0)

__global__ void __launch_bounds__(256) k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
// four loads in flight per lane
__global__ void __launch_bounds__(256) k_copy16x4(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a, dst[i + stride] = b, dst[i + 2 * stride] = c, dst[i + 3 * stride] = d;
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}

int main() {
  const size_t maxb = 4 << 20;
  uint8_t *h, *d, *pageable;
  CK(hipHostMalloc((void**)&h, maxb, hipHostMallocDefault));
  CK(hipMalloc((void**)&d, maxb));
  pageable = (uint8_t*)malloc(maxb);
  memset(h, 1, maxb), memset(pageable, 2, maxb);
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const int iters = 300;
  auto run = [&](auto body) {
    for (int i = 0; i < 20; i++) body();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; i++) body();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
  };
  std::printf("%10s | %9s %9s %9s %9s %9s | %9s %9s %9s\n", "bytes", "H2D rt", "H2D page", "H2D k16", "H2D k16x4", "H2D k64wg", "D2H rt",
              "D2H k16", "D2H k16x4");
  for (size_t bytes : {(size_t)2048, (size_t)16384, (size_t)77824, (size_t)368640, (size_t)921600, (size_t)(4 << 20)}) {
    const size_t n16 = bytes / 16;
    const int grid = (int)std::min<size_t>((n16 + 255) / 256, 1024), grid4 = (int)std::min<size_t>((n16 + 1023) / 1024, 512);
    const double a = run([&] { CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); });
    const double b = run([&] { CK(hipMemcpyAsync(d, pageable, bytes, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); });
    const double c = run([&] { hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, st, (const uint4*)h, (uint4*)d, n16); CK(hipStreamSynchronize(st)); });
    const double e = run([&] { hipLaunchKernelGGL(k_copy16x4, dim3(std::max(grid4, 1)), dim3(256), 0, st, (const uint4*)h, (uint4*)d, n16); CK(hipStreamSynchronize(st)); });
    const double f = run([&] { hipLaunchKernelGGL(k_copy16x4, dim3(64), dim3(256), 0, st, (const uint4*)h, (uint4*)d, n16); CK(hipStreamSynchronize(st)); });
    const double g = run([&] { CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); });
    const double i = run([&] { hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, st, (const uint4*)d, (uint4*)h, n16); CK(hipStreamSynchronize(st)); });
    const double j = run([&] { hipLaunchKernelGGL(k_copy16x4, dim3(std::max(grid4, 1)), dim3(256), 0, st, (const uint4*)d, (uint4*)h, n16); CK(hipStreamSynchronize(st)); });
    std::printf("%10zu | %9.1f %9.1f %9.1f %9.1f %9.1f | %9.1f %9.1f %9.1f\n", bytes, a, b, c, e, f, g, i, j);
  }
  // an empty kernel + synchronise: the floor
  const double z = run([&] { hipLaunchKernelGGL(k_copy16, dim3(1), dim3(256), 0, st, (const uint4*)d, (uint4*)d, (size_t)0); CK(hipStreamSynchronize(st)); });
  std::printf("empty kernel + synchronise: %.1f us\n", z);
  return 0;
}
