// valu_rate.hip -- issue rate of single VALU opcodes on gfx950 (wave64): which ones run at 2 cycles per
// wave-instruction (SIMD-32 rate) and which at 4 or more.  Each kernel runs N_ITER x 32 independent instances of one
// opcode per lane (8 accumulators x 4 unrolled), 8 waves per SIMD resident, and reports cycles per wave-instruction
// per SIMD = elapsed_cycles * SIMDs / wave_instructions.   Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_ITER 4096

#define OP_KERNEL(NAME, ASM)                                                                         \
  __global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned seed) {                        \
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;             \
    unsigned a4 = a0 ^ 0x55, a5 = a1 ^ 0x33, a6 = a2 ^ 0x0f, a7 = a3 ^ 0xf0;                         \
    unsigned b = seed | 0x00010001u, c = seed * 9 + 0x00030003u;                                     \
    for (int i = 0; i < N_ITER; i++) {                                                               \
      asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                           \
                   ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                           \
                   ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                           \
                   ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                           \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)  \
                   : "v"(b), "v"(c) : "vcc", "s10", "s11");                                          \
    }                                                                                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;              \
  }

#define A_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define A_PKFMA32(k) "v_add_u32 %" #k ", %" #k ", %8\n"
#define A_AND(k) "v_and_b32 %" #k ", %" #k ", %8\n"
#define A_PERM(k) "v_perm_b32 %" #k ", %" #k ", %8, %9\n"
#define A_ALIGN(k) "v_alignbyte_b32 %" #k ", %" #k ", %8, 1\n"
#define A_PKMAXI16(k) "v_pk_max_i16 %" #k ", %" #k ", %8\n"
#define A_PKSUBI16(k) "v_pk_sub_i16 %" #k ", %" #k ", %8\n"
#define A_PKMAXF16(k) "v_pk_max_f16 %" #k ", %" #k ", %8\n"
#define A_PKADDF16(k) "v_pk_add_f16 %" #k ", %" #k ", %8\n"
#define A_MIN3(k) "v_min3_i32 %" #k ", %" #k ", %8, %9\n"
#define A_MAXI32(k) "v_max_i32 %" #k ", %" #k ", %8\n"
#define A_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n"
#define A_MAD24(k) "v_mad_u32_u24 %" #k ", %" #k ", %8, %9\n"
#define A_BCNT(k) "v_bcnt_u32_b32 %" #k ", %" #k ", %8\n"
#define A_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 2, %8\n"
#define A_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 3, 5\n"
#define A_SAD(k) "v_sad_u8 %" #k ", %" #k ", %8, %9\n"
#define A_DOT4(k) "v_dot4_u32_u8 %" #k ", %" #k ", %8, %9\n"
#define A_MOV(k) "v_mov_b32 %" #k ", %8\n"
#define A_PKADDF32(k) "v_add_f32 %" #k ", %" #k ", %8\n"
#define A_MAXF32(k) "v_max_f32 %" #k ", %" #k ", %8\n"
#define A_MINU16(k) "v_min_u16 %" #k ", %" #k ", %8\n"
#define A_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define A_CNDMASK64(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[10:11]\n"
#define A_CMPCND(k) "v_cmp_le_u32_e32 vcc, %8, %" #k "\n v_cndmask_b32_e32 %" #k ", %" #k ", %9, vcc\n"
#define A_CMPCND64(k) "v_cmp_le_u32_e64 s[10:11], %8, %" #k "\n v_cndmask_b32_e64 %" #k ", %" #k ", %9, s[10:11]\n"
#define A_ASHR(k) "v_ashrrev_i32 %" #k ", 31, %" #k "\n"
#define A_LSHL(k) "v_lshlrev_b32 %" #k ", 3, %" #k "\n"
#define A_SUB(k) "v_sub_u32 %" #k ", %" #k ", %8\n"
#define A_BITOP3(k) "v_bitop3_b32 %" #k ", %" #k ", %8, %9 bitop3:0xcf\n"
#define A_OR3(k) "v_or3_b32 %" #k ", %" #k ", %8, %9\n"
#define A_XAD(k) "v_add3_u32 %" #k ", %" #k ", %8, %9\n"
#define A_PKMADI16(k) "v_pk_mad_i16 %" #k ", %" #k ", %8, %9\n"
#define A_PKMULF16(k) "v_pk_mul_f16 %" #k ", %" #k ", %8\n"
#define A_PKFMAF16(k) "v_pk_fma_f16 %" #k ", %" #k ", %8, %9\n"

OP_KERNEL(k_fma_f32, A_FMA)
OP_KERNEL(k_add_u32, A_PKFMA32)
OP_KERNEL(k_and_b32, A_AND)
OP_KERNEL(k_perm_b32, A_PERM)
OP_KERNEL(k_alignbyte, A_ALIGN)
OP_KERNEL(k_pk_max_i16, A_PKMAXI16)
OP_KERNEL(k_pk_sub_i16, A_PKSUBI16)
OP_KERNEL(k_pk_max_f16, A_PKMAXF16)
OP_KERNEL(k_pk_add_f16, A_PKADDF16)
OP_KERNEL(k_pk_mul_f16, A_PKMULF16)
OP_KERNEL(k_pk_fma_f16, A_PKFMAF16)
OP_KERNEL(k_pk_mad_i16, A_PKMADI16)
OP_KERNEL(k_min3_i32, A_MIN3)
OP_KERNEL(k_max_i32, A_MAXI32)
OP_KERNEL(k_mul_lo_u32, A_MULLO)
OP_KERNEL(k_mad_u32_u24, A_MAD24)
OP_KERNEL(k_bcnt, A_BCNT)
OP_KERNEL(k_lshl_add, A_LSHLADD)
OP_KERNEL(k_bfe, A_BFE)
OP_KERNEL(k_sad_u8, A_SAD)
OP_KERNEL(k_dot4_u8, A_DOT4)
OP_KERNEL(k_mov, A_MOV)
OP_KERNEL(k_add_f32, A_PKADDF32)
OP_KERNEL(k_max_f32, A_MAXF32)
OP_KERNEL(k_min_u16, A_MINU16)
OP_KERNEL(k_cndmask, A_CNDMASK)
OP_KERNEL(k_cndmask64, A_CNDMASK64)
OP_KERNEL(k_cmpcnd, A_CMPCND)
OP_KERNEL(k_cmpcnd64, A_CMPCND64)
OP_KERNEL(k_ashr, A_ASHR)
OP_KERNEL(k_lshl, A_LSHL)
OP_KERNEL(k_sub, A_SUB)
OP_KERNEL(k_bitop3, A_BITOP3)
OP_KERNEL(k_or3, A_OR3)
OP_KERNEL(k_add3, A_XAD)

typedef void (*kern_t)(unsigned*, unsigned);
struct Entry { const char* name; kern_t k; };

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * 8;  // 8 blocks x 4 waves per CU = 8 waves per SIMD
  unsigned* d;
  hipMalloc(&d, (size_t)blocks * 256 * 4);
  const Entry tab[] = {
      {"v_fma_f32", k_fma_f32}, {"v_add_f32", k_add_f32}, {"v_max_f32", k_max_f32}, {"v_mov_b32", k_mov},
      {"v_add_u32", k_add_u32}, {"v_and_b32", k_and_b32}, {"v_max_i32", k_max_i32}, {"v_min3_i32", k_min3_i32},
      {"v_add3_u32", k_add3}, {"v_or3_b32", k_or3}, {"v_bitop3_b32", k_bitop3}, {"v_cndmask_b32", k_cndmask},
      {"v_cndmask_b32_e64 (sgpr pair)", k_cndmask64}, {"v_cmp + v_cndmask (vcc), 2 instr", k_cmpcnd},
      {"v_cmp + v_cndmask (sgpr pair), 2 instr", k_cmpcnd64}, {"v_ashrrev_i32", k_ashr}, {"v_lshlrev_b32", k_lshl}, {"v_sub_u32", k_sub},
      {"v_lshl_add_u32", k_lshl_add}, {"v_bfe_u32", k_bfe}, {"v_bcnt_u32_b32", k_bcnt},
      {"v_perm_b32", k_perm_b32}, {"v_alignbyte_b32", k_alignbyte}, {"v_min_u16", k_min_u16},
      {"v_pk_max_i16", k_pk_max_i16}, {"v_pk_sub_i16", k_pk_sub_i16}, {"v_pk_mad_i16", k_pk_mad_i16},
      {"v_pk_max_f16", k_pk_max_f16}, {"v_pk_add_f16", k_pk_add_f16}, {"v_pk_mul_f16", k_pk_mul_f16},
      {"v_pk_fma_f16", k_pk_fma_f16}, {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mul_lo_u32", k_mul_lo_u32},
      {"v_sad_u8", k_sad_u8}, {"v_dot4_u32_u8", k_dot4_u8},
  };
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  int clk_khz = 0;
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  std::printf("device %s, %d CUs, clock attribute %d kHz; %d blocks x 256 threads, %d x 32 ops per lane\n",
              prop.gcnArchName, cus, clk_khz, blocks, N_ITER);
  double t_fma = 0;
  for (const Entry& e : tab) {
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 1u);  // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    const double wave_instr = (double)blocks * 4 * N_ITER * 32;  // per launch
    const double per_simd = wave_instr / (cus * 4.0);
    const double ns_per = ms * 1e6 / per_simd;
    if (t_fma == 0) t_fma = ns_per;
    std::printf("%-18s %8.3f ms  %6.3f ns per wave-instruction per SIMD  = %.2f x v_fma_f32 (2 cycles)  => %.1f cycles\n",
                e.name, ms, ns_per, ns_per / t_fma, 2.0 * ns_per / t_fma);
  }
  return 0;
}
