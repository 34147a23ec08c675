// Instruction-fetch cost of straight-line code on gfx950: one workgroup, W wavefronts, each running a loop whose body is
// KB kilobytes of 8-byte v_fma_f64 (dependent chain of length CH = 1 or 2 independent chains), REP passes.  A CU pair
// shares a 64 KB instruction cache (MI355X_MICROARCH.md): a loop body beyond it is fetched from L2 on every pass.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/icache_bench.hip -o tools/micro/icache_bench
// Output: cycles per instruction for each (body size, wavefronts, same / different code per wavefront).
// The question behind it: k_pose_opt_vio<256> is 207 KB of code (profiles/r6_pose_code_size.txt); how much of its time
// is instruction fetch?
#include <hip/hip_runtime.h>
#include <cstdio>

// (the instructions are spelled out, not .rept: the compiler sizes an asm statement by its lines, and a loop around a
// body it believes small gets a short branch that cannot reach)
#define I2 "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
#define I8 I2 I2 I2 I2
#define I32 I8 I8 I8 I8
#define I128 I32 I32 I32 I32
#define I512 I128 I128 I128 I128
#define B512 asm volatile(I512 I512 : "+v"(x), "+v"(y) : "v"(a), "v"(b))  /* 1024 instructions = 8 KB */
#define B1024 B512; B512
#define B2048 B1024; B1024
#define B3072 B2048; B1024
#define B4096 B2048; B2048
#define B6144 B4096; B2048
#define B8192 B4096; B4096
#define B16384 B8192; B8192
#define BODY(NINST) { B##NINST; }

// KB of code per pass = NINST pairs * 16 bytes / 1024.  DIFF: every wavefront runs its own copy of the body.
template <int KB, bool DIFF>
__global__ void __launch_bounds__(256) k(double* out, unsigned long long* t, int rep, double a, double b) {
  double x = threadIdx.x * 1e-3, y = x + 1;
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < rep; r++) {
    if (!DIFF || wave == 0) {
      if constexpr (KB == 8) BODY(512)
      if constexpr (KB == 16) BODY(1024)
      if constexpr (KB == 32) BODY(2048)
      if constexpr (KB == 48) BODY(3072)
      if constexpr (KB == 64) BODY(4096)
      if constexpr (KB == 96) BODY(6144)
      if constexpr (KB == 128) BODY(8192)
      if constexpr (KB == 256) BODY(16384)
    } else if (wave == 1) {
      if constexpr (KB == 8) BODY(512)
      if constexpr (KB == 16) BODY(1024)
      if constexpr (KB == 32) BODY(2048)
      if constexpr (KB == 48) BODY(3072)
      if constexpr (KB == 64) BODY(4096)
      if constexpr (KB == 96) BODY(6144)
      if constexpr (KB == 128) BODY(8192)
      if constexpr (KB == 256) BODY(16384)
    } else if (wave == 2) {
      if constexpr (KB == 8) BODY(512)
      if constexpr (KB == 16) BODY(1024)
      if constexpr (KB == 32) BODY(2048)
      if constexpr (KB == 48) BODY(3072)
      if constexpr (KB == 64) BODY(4096)
      if constexpr (KB == 96) BODY(6144)
      if constexpr (KB == 128) BODY(8192)
      if constexpr (KB == 256) BODY(16384)
    } else {
      if constexpr (KB == 8) BODY(512)
      if constexpr (KB == 16) BODY(1024)
      if constexpr (KB == 32) BODY(2048)
      if constexpr (KB == 48) BODY(3072)
      if constexpr (KB == 64) BODY(4096)
      if constexpr (KB == 96) BODY(6144)
      if constexpr (KB == 128) BODY(8192)
      if constexpr (KB == 256) BODY(16384)
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) t[wave] = t1 - t0;
  out[threadIdx.x] = x + y;
}

template <int KB, bool DIFF>
static void run(int waves, double* d_out, unsigned long long* d_t) {
  const int rep = 64 * 64 / KB;  // the same instruction count for every size
  unsigned long long h[4] = {0, 0, 0, 0};
  for (int pass = 0; pass < 2; pass++) {  // the second pass is the one reported
    hipLaunchKernelGGL((k<KB, DIFF>), dim3(1), dim3(64 * waves), 0, 0, d_out, d_t, rep, 0.999, 1e-9);
    hipDeviceSynchronize();
    hipMemcpy(h, d_t, sizeof(h), hipMemcpyDeviceToHost);
  }
  const double insts = (double)rep * KB * 1024 / 8;
  unsigned long long mx = 0;
  for (int w = 0; w < waves; w++) mx = h[w] > mx ? h[w] : mx;
  printf("body %3d KB  waves %d  %s  %.2f s_memtime counts per instruction per wavefront\n", KB, waves,
         DIFF ? "own code per wave " : "same code all waves", (double)mx / insts);
}

int main() {
  double* d_out;
  unsigned long long* d_t;
  hipMalloc(&d_out, 256 * 8);
  hipMalloc(&d_t, 4 * 8);
  for (int waves : {1, 4}) {
    run<8, false>(waves, d_out, d_t);
    run<16, false>(waves, d_out, d_t);
    run<32, false>(waves, d_out, d_t);
    run<48, false>(waves, d_out, d_t);
    run<64, false>(waves, d_out, d_t);
    run<96, false>(waves, d_out, d_t);
    run<128, false>(waves, d_out, d_t);
    run<256, false>(waves, d_out, d_t);
  }
  run<8, true>(4, d_out, d_t);
  run<16, true>(4, d_out, d_t);
  run<32, true>(4, d_out, d_t);
  run<64, true>(4, d_out, d_t);
  return 0;
}
