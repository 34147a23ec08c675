// wave_ops.h -- wavefront-wide reductions on DPP (data-parallel primitives: the VALU reads a neighbouring lane's
// register directly), result uniform over the 64 lanes.
//
// `__shfl_xor` compiles to ds_bpermute_b32 on gfx950: every step of a butterfly is a trip through the LDS crossbar
// (address VGPR, lgkmcnt wait, ~100 cycles of latency) and a reduction is six dependent ones.  The kernels that are one
// or four wavefronts deep and latency-bound (the claim replay of the projection search, IC_Angle, the SAD of the stereo
// refinement, the block sums of the pose optimisations and of the local BA) spent a large part of their time there.
// The DPP forms below need no LDS and no address: 4 steps inside a row of 16 lanes (two quad permutes, the two row
// mirrors), two row broadcasts, one v_readlane.
//
// All 64 lanes must be active at the call (the result is read from lane 63, and a DPP read of an inactive lane returns
// the `old` operand): call them in wave-uniform control flow only.
#pragma once
#include <hip/hip_runtime.h>

namespace vieo {

// dpp_ctrl encodings (LLVM AMDGPU DPP): quad_perm = the four 2-bit selectors; the named ones below
#define VIEO_DPP_QUAD_XOR1 0xB1   /* quad_perm:[1,0,3,2] */
#define VIEO_DPP_QUAD_XOR2 0x4E   /* quad_perm:[2,3,0,1] */
#define VIEO_DPP_ROW_HALF_MIRROR 0x141
#define VIEO_DPP_ROW_MIRROR 0x140
#define VIEO_DPP_ROW_BCAST15 0x142
#define VIEO_DPP_ROW_BCAST31 0x143

// lanes that receive nothing (row_mask) keep `old`
#define VIEO_DPP(old, v, ctrl, row_mask) __builtin_amdgcn_update_dpp((int)(old), (int)(v), ctrl, row_mask, 0xF, false)

__device__ __forceinline__ int wave_sum_i32(int v) {
  v += VIEO_DPP(0, v, VIEO_DPP_QUAD_XOR1, 0xF);
  v += VIEO_DPP(0, v, VIEO_DPP_QUAD_XOR2, 0xF);
  v += VIEO_DPP(0, v, VIEO_DPP_ROW_HALF_MIRROR, 0xF);
  v += VIEO_DPP(0, v, VIEO_DPP_ROW_MIRROR, 0xF);   // every lane: the sum of its row of 16
  v += VIEO_DPP(0, v, VIEO_DPP_ROW_BCAST15, 0xA);  // rows 1, 3 += rows 0, 2
  v += VIEO_DPP(0, v, VIEO_DPP_ROW_BCAST31, 0xC);  // rows 2, 3 += row 1 (= rows 0 + 1)
  return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  v = min(v, (unsigned)VIEO_DPP(v, v, VIEO_DPP_QUAD_XOR1, 0xF));
  v = min(v, (unsigned)VIEO_DPP(v, v, VIEO_DPP_QUAD_XOR2, 0xF));
  v = min(v, (unsigned)VIEO_DPP(v, v, VIEO_DPP_ROW_HALF_MIRROR, 0xF));
  v = min(v, (unsigned)VIEO_DPP(v, v, VIEO_DPP_ROW_MIRROR, 0xF));
  v = min(v, (unsigned)VIEO_DPP(v, v, VIEO_DPP_ROW_BCAST15, 0xA));
  v = min(v, (unsigned)VIEO_DPP(v, v, VIEO_DPP_ROW_BCAST31, 0xC));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// double: the two halves travel as two 32-bit DPP moves.  The association order is fixed (quads, half rows, rows, row
// pairs, halves), the same for every lane and every launch.
__device__ __forceinline__ double wave_sum_f64(double v) {
#define VIEO_DPP_ADD_F64(ctrl, row_mask)                                  \
  {                                                                       \
    const int lo = __double2loint(v), hi = __double2hiint(v);             \
    const int l2 = VIEO_DPP(0, lo, ctrl, row_mask), h2 = VIEO_DPP(0, hi, ctrl, row_mask); \
    v += __hiloint2double(h2, l2);                                        \
  }
  VIEO_DPP_ADD_F64(VIEO_DPP_QUAD_XOR1, 0xF)
  VIEO_DPP_ADD_F64(VIEO_DPP_QUAD_XOR2, 0xF)
  VIEO_DPP_ADD_F64(VIEO_DPP_ROW_HALF_MIRROR, 0xF)
  VIEO_DPP_ADD_F64(VIEO_DPP_ROW_MIRROR, 0xF)
  VIEO_DPP_ADD_F64(VIEO_DPP_ROW_BCAST15, 0xA)
  VIEO_DPP_ADD_F64(VIEO_DPP_ROW_BCAST31, 0xC)
#undef VIEO_DPP_ADD_F64
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// sum over the four lanes of a quad (every lane of the quad receives it); same association as two xor butterflies
__device__ __forceinline__ double quad_sum_f64(double v) {
  {
    const int l2 = VIEO_DPP(0, __double2loint(v), VIEO_DPP_QUAD_XOR1, 0xF), h2 = VIEO_DPP(0, __double2hiint(v), VIEO_DPP_QUAD_XOR1, 0xF);
    v += __hiloint2double(h2, l2);
  }
  {
    const int l2 = VIEO_DPP(0, __double2loint(v), VIEO_DPP_QUAD_XOR2, 0xF), h2 = VIEO_DPP(0, __double2hiint(v), VIEO_DPP_QUAD_XOR2, 0xF);
    v += __hiloint2double(h2, l2);
  }
  return v;
}

__device__ __forceinline__ double wave_max_f64(double v) {
#define VIEO_DPP_MAX_F64(ctrl, row_mask)                                                      \
  {                                                                                           \
    const int lo = __double2loint(v), hi = __double2hiint(v);                                 \
    const int l2 = VIEO_DPP(lo, lo, ctrl, row_mask), h2 = VIEO_DPP(hi, hi, ctrl, row_mask);   \
    v = fmax(v, __hiloint2double(h2, l2));                                                    \
  }
  VIEO_DPP_MAX_F64(VIEO_DPP_QUAD_XOR1, 0xF)
  VIEO_DPP_MAX_F64(VIEO_DPP_QUAD_XOR2, 0xF)
  VIEO_DPP_MAX_F64(VIEO_DPP_ROW_HALF_MIRROR, 0xF)
  VIEO_DPP_MAX_F64(VIEO_DPP_ROW_MIRROR, 0xF)
  VIEO_DPP_MAX_F64(VIEO_DPP_ROW_BCAST15, 0xA)
  VIEO_DPP_MAX_F64(VIEO_DPP_ROW_BCAST31, 0xC)
#undef VIEO_DPP_MAX_F64
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

}  // namespace vieo
