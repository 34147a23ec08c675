// matching.hip -- Hamming matching kernels of the hot path (gfx950).
//
//   k_knn2            cv::BFMatcher(NORM_HAMMING).knnMatch(k=2)  (reference: src/Frame.cc:18,620-628,
//                     the dense N x N search of ComputeStereoFishEyeMatches)
//   k_stereo_rect     Frame::ComputeStereoMatches                (src/Frame.cc:451-597)
//   k_stereo_median   its median-SAD outlier rejection           (src/Frame.cc:599-610)
//   ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1645-1667) is xor + v_bcnt_u32_b32 on the
//   8 dwords of a 32-byte row.
// One wavefront per query / left key; the 32-byte train rows are streamed through L2 (a frame's
// descriptor matrix is 38 KB).  Integer work: results are bit-exact against oracle/matching.cc.
#include <climits>

#include "orb_internal.h"
#include "wave_ops.h"

namespace vieo {

static const int TH_HIGH = 100, TH_LOW = 50;  // ORBmatcher.cc:20-21

__device__ __forceinline__ int hamming32(const uint4 a0, const uint4 a1, const uint8_t* b) {
  const uint4 b0 = ((const uint4*)b)[0], b1 = ((const uint4*)b)[1];
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
         __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// lexicographic (dist, idx) ordering == cv::batchDistance's stable insertion order
__device__ __forceinline__ bool lex_less(int d0, int i0, int d1, int i1) {
  return d0 < d1 || (d0 == d1 && i0 < i1);
}

struct Knn2Job {
  const uint8_t* q;  // query rows
  const uint8_t* t;  // train rows
  int nq, nt;
  int out_off;  // row offset into idx/dist outputs
};

// Where the jobs of a launch come from: a table in HBM (host-built), or the extractor's arrays of a batch of camera-rig
// frames -- pair p of frame f searches camera pi[p]'s rows [num_mono, n) in camera pj[p]'s, the counts read on the
// device (Frame.cc:620-628), so that a rig frame's stereo stage needs no host round trip.
struct Knn2Src {
  const Knn2Job* jobs;     // non-NULL: job blockIdx.y of the table
  const uint8_t* desc;     // [frame][n_cams][cap][32]
  const int32_t* counts;   // [frame][n_cams][2] = {n, num_mono}
  int cap, n_cams, n_pairs;
  signed char pi[6], pj[6];
};

// grid (ceil(max_nq / 64), n_jobs | n_pairs, n_frames), 256 threads.  A lane owns one query row (8 dwords in registers);
// the train rows pass through LDS in tiles of 256 (one coalesced copy per workgroup instead of one 32-byte gather per
// (query, train) pair and lane), wavefront w walks rows w*64.. of every tile with wave-uniform LDS addresses (a
// broadcast read, no bank conflicts).  best / second travel as ONE packed word (distance << 16 | train index): the
// lexicographic order of (distance, index) == cv::batchDistance's stable insertion order is the order of the words, so
// an update is a v_med3_u32 and a v_min_u32, and the four wavefronts' pairs merge with min / max.
__global__ void __launch_bounds__(256)
k_knn2(Knn2Src S, int32_t* __restrict__ idx, int32_t* __restrict__ dist) {
  __shared__ uint4 s_t[256 * 2];
  __shared__ unsigned s_m[3][64][2];
  Knn2Job J;
  if (S.jobs)
    J = S.jobs[blockIdx.y];
  else {
    const int f = blockIdx.z, p = blockIdx.y, ci = S.pi[p], cj = S.pj[p];
    const int32_t* c = S.counts + (size_t)f * S.n_cams * 2;
    const int ni = c[2 * ci], mi = c[2 * ci + 1], nj = c[2 * cj], mj = c[2 * cj + 1];
    const bool skip = mi >= ni || mj >= nj;  // Frame.cc:623
    J.q = S.desc + (((size_t)f * S.n_cams + ci) * S.cap + mi) * 32;
    J.t = S.desc + (((size_t)f * S.n_cams + cj) * S.cap + mj) * 32;
    J.nq = skip ? 0 : min(ni, S.cap) - mi, J.nt = skip ? 0 : min(nj, S.cap) - mj;
    J.out_off = ((int)f * S.n_pairs + p) * S.cap;
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int q0 = blockIdx.x * 64;
  if (q0 >= J.nq) return;  // (uniform over the workgroup)
  const int qi = q0 + lane;
  uint4 a0 = {0, 0, 0, 0}, a1 = a0;
  if (qi < J.nq) a0 = ((const uint4*)(J.q + (size_t)qi * 32))[0], a1 = ((const uint4*)(J.q + (size_t)qi * 32))[1];
  unsigned k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;  // best, second: distance << 16 | train row
  const uint4* T = (const uint4*)J.t;
  for (int t0 = 0; t0 < J.nt; t0 += 256) {
    __syncthreads();
    const int rows = min(256, J.nt - t0);
    for (int e = threadIdx.x; e < rows * 2; e += 256) s_t[e] = T[(size_t)t0 * 2 + e];
    __syncthreads();
    const int r1 = min(rows, wave * 64 + 64);
    for (int r = wave * 64; r < r1; r++) {
      const uint4 b0 = s_t[2 * r], b1 = s_t[2 * r + 1];
      const unsigned d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                         __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
      const unsigned key = (d << 16) | (unsigned)(t0 + r);
      k1 = min(max(key, k0), k1);  // second smallest of {key, k0 <= k1} (v_med3_u32)
      k0 = min(k0, key);
    }
  }
  if (wave > 0) s_m[wave - 1][lane][0] = k0, s_m[wave - 1][lane][1] = k1;
  __syncthreads();
  if (wave == 0 && qi < J.nq) {
#pragma unroll
    for (int w = 0; w < 3; w++) {
      const unsigned b0 = s_m[w][lane][0], b1 = s_m[w][lane][1];
      const unsigned lo = min(k0, b0), hi = max(k0, b0);
      k1 = min(hi, min(k1, b1));
      k0 = lo;
    }
    int32_t* oi = idx + ((size_t)J.out_off + qi) * 2;
    int32_t* od = dist + ((size_t)J.out_off + qi) * 2;
    oi[0] = k0 == 0xFFFFFFFFu ? -1 : (int)(k0 & 0xFFFF), od[0] = k0 == 0xFFFFFFFFu ? INT_MAX : (int)(k0 >> 16);
    oi[1] = k1 == 0xFFFFFFFFu ? -1 : (int)(k1 & 0xFFFF), od[1] = k1 == 0xFFFFFFFFu ? INT_MAX : (int)(k1 >> 16);
  }
}

// ---- the same search on the matrix cores (round 5).  Hamming(a, b) = |a| + |b| - 2 a.b with the descriptors' 256 bits as
// 0 / 1 int8 vectors: a.b for a 32 x 32 block of (train row, query) pairs is eight v_mfma_i32_32x32x32_i8.  The K order is
// free as long as both operands use the same one, so a lane takes the 16 bytes [16 g, 16 g + 16) of its row (g = lane >> 5)
// and MFMA m the bits of bytes 2 m, 2 m + 1 of those -- no knowledge of the instruction's K layout is needed, only that the
// A operand's lane l is row l & 31, the B operand's lane l column l & 31, and D's lane l holds column l & 31, rows
// (r & 3) + 8 (r >> 2) + 4 (l >> 5) (cdna_hip_programming.md).
// Work: a workgroup = 256 queries (a wavefront keeps TWO sets of 32 as expanded B fragments, 64 registers, for the whole
// kernel), the train rows pass in tiles of 32, expanded ONCE per workgroup into LDS in fragment order (a lane's operand of
// MFMA m is one conflict-free ds_read_b128), double-buffered, one barrier per tile.  The tile also carries one word per
// row, (|a| + 512) << 16 | row: key = that - (a.b << 17) is the packed (distance - |b| + 512, row) word in ONE v_mad --
// |b| is the same for all keys of a lane and is added at the end -- and best / second are the v_med3 / v_min of k_knn2.
// Per 32 x 32 pairs and wavefront: 8 MFMA (~260 cycles of the matrix pipe) against 48 key instructions + a quarter of the
// tile's expansion; k_knn2 spends 19 vector instructions per PAIR and lane.
typedef int knn_v16i __attribute__((ext_vector_type(16)));
typedef int knn_v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned knn_spread4(unsigned nib) { return (nib * 0x00204081u) & 0x01010101u; }  // 4 bits -> 4 bytes 0 / 1
__device__ __forceinline__ knn_v4i knn_spread16(unsigned h) {
  knn_v4i r;
  r[0] = (int)knn_spread4(h & 15u), r[1] = (int)knn_spread4((h >> 4) & 15u);
  r[2] = (int)knn_spread4((h >> 8) & 15u), r[3] = (int)knn_spread4((h >> 12) & 15u);
  return r;
}

#ifndef VIEO_KNN2_TILE_ROWS
#define VIEO_KNN2_TILE_ROWS 32  // train rows per barrier (32 or 64).  64 halves the barriers but needs four accumulators: 210 registers,
                                // two wavefronts per SIMD instead of three -- 1.00 against 0.92 ms per 1536 searches (tools/ab_knn2_tile.sh)
#endif
__global__ void __launch_bounds__(256)
k_knn2_mfma(Knn2Src S, int32_t* __restrict__ idx, int32_t* __restrict__ dist) {
  constexpr int TR = VIEO_KNN2_TILE_ROWS, NSUB = TR / 32, CH = TR * 16;  // chunks of 16 expanded bytes per tile
  // chunk ((sub * 8 + m) * 2 + g) * 32 + i = the 16 expanded bytes of train row sub * 32 + i for MFMA m, half g
  __shared__ knn_v4i s_a[2][CH];
  __shared__ __attribute__((aligned(16))) unsigned s_L[2][TR];
  Knn2Job J;
  if (S.jobs)
    J = S.jobs[blockIdx.y];
  else {
    const int f = blockIdx.z, p = blockIdx.y, ci = S.pi[p], cj = S.pj[p];
    const int32_t* c = S.counts + (size_t)f * S.n_cams * 2;
    const int ni = c[2 * ci], mi = c[2 * ci + 1], nj = c[2 * cj], mj = c[2 * cj + 1];
    const bool skip = mi >= ni || mj >= nj;  // Frame.cc:623
    J.q = S.desc + (((size_t)f * S.n_cams + ci) * S.cap + mi) * 32;
    J.t = S.desc + (((size_t)f * S.n_cams + cj) * S.cap + mj) * 32;
    J.nq = skip ? 0 : min(ni, S.cap) - mi, J.nt = skip ? 0 : min(nj, S.cap) - mj;
    J.out_off = ((int)f * S.n_pairs + p) * S.cap;
  }
  if ((int)blockIdx.x * 256 >= J.nq) return;  // (uniform over the workgroup)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, g = lane >> 5;
  const int q0 = blockIdx.x * 256 + wave * 64;
  // ---- this wavefront's queries as B fragments
  knn_v4i B[2][8];
  unsigned pb[2];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const int qi = q0 + 32 * s + j;
    uint4 h = {0, 0, 0, 0};
    if (qi < J.nq) h = ((const uint4*)(J.q + (size_t)qi * 32))[g];
    const unsigned half = __popc(h.x) + __popc(h.y) + __popc(h.z) + __popc(h.w);
    pb[s] = half + (unsigned)__shfl_xor((int)half, 32);
    const unsigned w[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
    for (int m = 0; m < 8; m++) B[s][m] = knn_spread16((w[m >> 1] >> (16 * (m & 1))) & 0xFFFFu);
  }
  unsigned k0[2] = {0xFFFFFFFFu, 0xFFFFFFFFu}, k1[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
  // ---- staging roles: thread t expands chunks t + 256 u of a tile; threads 0 .. TR - 1 its rows' words
  const int n_tiles = (J.nt + TR - 1) / TR;
  constexpr int NU = CH / 256;
  unsigned h16[NU];
  uint4 ra = {0, 0, 0, 0}, rb = ra;
  auto fetch = [&](int T) {
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const int cc = tid + 256 * u, sub = cc >> 9, m = (cc >> 6) & 7, gg = (cc >> 5) & 1, row = T * TR + sub * 32 + (cc & 31);
      h16[u] = row < J.nt ? (unsigned)*(const unsigned short*)(J.t + (size_t)row * 32 + 16 * gg + 2 * m) : 0u;
    }
    if (tid < TR) {
      const int row = T * TR + tid;
      ra = rb = make_uint4(0, 0, 0, 0);
      if (row < J.nt) ra = ((const uint4*)(J.t + (size_t)row * 32))[0], rb = ((const uint4*)(J.t + (size_t)row * 32))[1];
    }
  };
  auto stage = [&](int T, int b) {
#pragma unroll
    for (int u = 0; u < NU; u++) s_a[b][tid + 256 * u] = knn_spread16(h16[u]);
    if (tid < TR) {
      const int row = T * TR + tid;
      const unsigned pa = __popc(ra.x) + __popc(ra.y) + __popc(ra.z) + __popc(ra.w) + __popc(rb.x) + __popc(rb.y) + __popc(rb.z) + __popc(rb.w);
      s_L[b][tid] = row < J.nt ? ((pa + 512u) << 16) | (unsigned)row : 0xFFFFFFFFu;
    }
  };
  fetch(0);
  stage(0, 0);
  __syncthreads();
  for (int T = 0; T < n_tiles; T++) {
    const int b = T & 1;
    if (T + 1 < n_tiles) fetch(T + 1);  // (in flight during this tile's MFMAs)
#pragma unroll
    for (int sub = 0; sub < NSUB; sub++) {
      knn_v16i acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const knn_v4i a = s_a[b][((sub * 8 + m) * 2 + g) * 32 + j];
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B[0][m], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B[1][m], acc1, 0, 0, 0);
      }
#pragma unroll
      for (int q4 = 0; q4 < 4; q4++) {
        const uint4 Lw = *(const uint4*)&s_L[b][sub * 32 + 8 * q4 + 4 * g];
        const unsigned L[4] = {Lw.x, Lw.y, Lw.z, Lw.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const unsigned ka = L[e] - ((unsigned)acc0[4 * q4 + e] << 17), kb = L[e] - ((unsigned)acc1[4 * q4 + e] << 17);
          k1[0] = min(max(ka, k0[0]), k1[0]), k0[0] = min(k0[0], ka);
          k1[1] = min(max(kb, k0[1]), k1[1]), k0[1] = min(k0[1], kb);
        }
      }
    }
    if (T + 1 < n_tiles) stage(T + 1, b ^ 1);
    __syncthreads();
  }
  // ---- the two lanes of a query (rows 4 g .. of every group of 8) merge; |b| joins the distances
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const unsigned o0 = (unsigned)__shfl_xor((int)k0[s], 32), o1 = (unsigned)__shfl_xor((int)k1[s], 32);
    const unsigned lo = min(k0[s], o0), hi = max(k0[s], o0);
    const unsigned second = min(hi, min(k1[s], o1));
    const int qi = q0 + 32 * s + j;
    if (g == 0 && qi < J.nq) {
      int32_t* oi = idx + ((size_t)J.out_off + qi) * 2;
      int32_t* od = dist + ((size_t)J.out_off + qi) * 2;
      oi[0] = lo == 0xFFFFFFFFu ? -1 : (int)(lo & 0xFFFF), od[0] = lo == 0xFFFFFFFFu ? INT_MAX : (int)(lo >> 16) - 512 + (int)pb[s];
      oi[1] = second == 0xFFFFFFFFu ? -1 : (int)(second & 0xFFFF);
      od[1] = second == 0xFFFFFFFFu ? INT_MAX : (int)(second >> 16) - 512 + (int)pb[s];
    }
  }
}

// VIEO_KNN2_MFMA=0: the popcount kernel (A/B timing)
static bool knn2_mfma() {
  static const bool on = [] {
    const char* e = getenv("VIEO_KNN2_MFMA");
    return !e || atoi(e) != 0;
  }();
  return on;
}
static void knn2_launch(const Knn2Src& S, int max_nq, int ny, int nz, int32_t* d_idx, int32_t* d_dist, hipStream_t st) {
  if (knn2_mfma())
    hipLaunchKernelGGL(k_knn2_mfma, dim3((max_nq + 255) / 256, ny, nz), dim3(256), 0, st, S, d_idx, d_dist);
  else
    hipLaunchKernelGGL(k_knn2, dim3((max_nq + 63) / 64, ny, nz), dim3(256), 0, st, S, d_idx, d_dist);
}

// the rig form for the device-resident stereo stage of camera-rig frames (fisheye_stereo.hip)
int knn2_rig_launch(const uint8_t* d_desc, const int32_t* d_counts, int cap, int n_cams, int n_frames, int32_t* d_idx,
                    int32_t* d_dist, hipStream_t st) {
  Knn2Src S;
  memset(&S, 0, sizeof(S));
  S.desc = d_desc, S.counts = d_counts, S.cap = cap, S.n_cams = n_cams;
  int p = 0;
  for (int i = 0; i < n_cams - 1; i++)
    for (int j = i + 1; j < n_cams; j++, p++) S.pi[p] = (signed char)i, S.pj[p] = (signed char)j;
  S.n_pairs = p;
  if (p == 0 || n_frames <= 0) return VIEO_OK;
  knn2_launch(S, cap, p, n_frames, d_idx, d_dist, st);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

// ------------------------------------------------------------------ rectified stereo
struct StereoArgs {
  OrbParams P;     // level geometry (both cameras identical)
  ImgSet IL, IR;   // planes of the left / right extractor batch
  int l_first, l_step, r_first, r_step;  // image index of frame f inside each batch
  const vieo_keypoint *kpL, *kpR;        // [image][cap]
  const uint8_t *descL, *descR;          // [image][cap][32]
  const int *cntL, *cntR;                // [image][2]
  int capL, capR;
  float baseline, bf;
  float* uright;  // [frame][capL]
  float* depth;   // [frame][capL]
  int* sad;       // [frame][capL]  best SAD of accepted matches, -1 otherwise
  int* row_start; // [frame][H + 1]  vRowIndices as CSR: right keys whose row band covers row y
  int2* row_list;  // [frame][list_cap]: {right key | octave << 20, its x as float bits}: the candidate loop needs no key record
  int H, list_cap;
};

__device__ __forceinline__ int wave_sum_i(int v) { return wave_sum_i32(v); }

// vRowIndices (Frame.cc:461-480): for every image row the right keys whose band
// [floor(y - r), ceil(y + r)], r = 2 * scale(octave), covers it.  One workgroup per frame: LDS
// histogram, prefix sum, fill.  The order inside a row is irrelevant: the best match is the
// minimum of (distance, index).
// NT threads: 1024 for a call of a few frames (one or two trips to the keys per pass instead of five), 256 for batches
template <int NT>
__global__ void __launch_bounds__(NT) k_stereo_rows(StereoArgs A) {
  extern __shared__ int s_rows[];  // cnt[H + 1], then fill cursor[H]
  __shared__ int s_part[16];
  const int f = blockIdx.x, tid = threadIdx.x, H = A.H;
  int* cnt = s_rows;
  int* cur = s_rows + H + 1;
  const int imR = A.r_first + f * A.r_step;
  const int Nr = min(A.cntR[2 * imR], A.capR);
  const vieo_keypoint* KR = A.kpR + (size_t)imR * A.capR;
  for (int y = tid; y <= H; y += NT) cnt[y] = 0;
  __syncthreads();
  for (int j = tid; j < Nr; j += NT) {
    const float y = KR[j].y, r = 2.0f * A.P.lv[KR[j].octave].scale;
    const int maxr = min((int)ceilf(y + r), H - 1), minr = max((int)floorf(y - r), 0);
    for (int yi = minr; yi <= maxr; yi++) atomicAdd(&cnt[yi], 1);
  }
  __syncthreads();
  // exclusive prefix sum of cnt[0..H)
  const int per = (H + NT - 1) / NT, y0 = min(H, tid * per), y1 = min(H, y0 + per);
  int sum = 0;
  for (int y = y0; y < y1; y++) sum += cnt[y];
  int inc = sum;
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o);
    if ((tid & 63) >= o) inc += t;
  }
  if ((tid & 63) == 63) s_part[tid >> 6] = inc;
  __syncthreads();
  int acc = inc - sum;
  for (int w = 0; w < (tid >> 6); w++) acc += s_part[w];
  int* rs = A.row_start + (size_t)f * (H + 1);
  for (int y = y0; y < y1; y++) {
    const int v = cnt[y];
    rs[y] = acc, cur[y] = acc;
    acc += v;
  }
  if (tid == NT - 1) rs[H] = acc;  // (the running sum of the last thread: everything)
  __syncthreads();
  int2* list = A.row_list + (size_t)f * A.list_cap;
  for (int j = tid; j < Nr; j += NT) {
    const vieo_keypoint kj = KR[j];
    const float y = kj.y, r = 2.0f * A.P.lv[kj.octave].scale;
    const int maxr = min((int)ceilf(y + r), H - 1), minr = max((int)floorf(y - r), 0);
    for (int yi = minr; yi <= maxr; yi++) {
      const int pos = atomicAdd(&cur[yi], 1);
      if (pos < A.list_cap) list[pos] = make_int2(j | (kj.octave << 20), __float_as_int(kj.x));
    }
  }
}

// grid (ceil(capL/16), n_frames); FOUR left keys per wavefront, one per row of 16 lanes (round 3).  A key's row holds
// about 20 candidates and its refinement is 121 + 231 bytes of patches: with a wavefront per key most lanes idled through
// a chain of four dependent round trips (key -> row list -> descriptors -> patches) and 2.5 M wavefronts queued for
// them.  Same arithmetic: the best match is the minimum of (distance, index), the SADs are integer sums.
__device__ __forceinline__ unsigned row_min_u32(unsigned v) {  // minimum over the lane's row of 16
  v = min(v, (unsigned)VIEO_DPP(v, v, VIEO_DPP_QUAD_XOR1, 0xF));
  v = min(v, (unsigned)VIEO_DPP(v, v, VIEO_DPP_QUAD_XOR2, 0xF));
  v = min(v, (unsigned)VIEO_DPP(v, v, VIEO_DPP_ROW_HALF_MIRROR, 0xF));
  v = min(v, (unsigned)VIEO_DPP(v, v, VIEO_DPP_ROW_MIRROR, 0xF));
  return v;
}
__device__ __forceinline__ int row_sum_i32(int v) {  // sum over the lane's row of 16 (every lane receives it)
  v += VIEO_DPP(0, v, VIEO_DPP_QUAD_XOR1, 0xF);
  v += VIEO_DPP(0, v, VIEO_DPP_QUAD_XOR2, 0xF);
  v += VIEO_DPP(0, v, VIEO_DPP_ROW_HALF_MIRROR, 0xF);
  v += VIEO_DPP(0, v, VIEO_DPP_ROW_MIRROR, 0xF);
  return v;
}

__global__ void __launch_bounds__(256) k_stereo_rect(StereoArgs A) {
  const int f = blockIdx.y, lane = threadIdx.x & 63, g = lane >> 4, gl = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int iL = (blockIdx.x * 4 + wave) * 4 + g;
  const int imL = A.l_first + f * A.l_step, imR = A.r_first + f * A.r_step;
  const int N = min(A.cntL[2 * imL], A.capL);
  bool alive = iL < N;
  const int iLc = alive ? iL : 0;
  float* o_ur = A.uright + (size_t)f * A.capL + iLc;
  float* o_dp = A.depth + (size_t)f * A.capL + iLc;
  int* o_sad = A.sad + (size_t)f * A.capL + iLc;
  if (alive && gl == 0) *o_ur = -1.0f, *o_dp = -1.0f, *o_sad = -1;
  if (!__any(alive)) return;
  const vieo_keypoint kL = A.kpL[(size_t)imL * A.capL + iLc];
  const int levelL = kL.octave;
  const float vL = kL.y, uL = kL.x;
  const int rowL = (int)vL;  // vRowIndices[vL]
  const float minD = 0.f, maxD = A.bf / A.baseline;
  const float minU = uL - maxD, maxU = uL - minD;
  alive = alive && !(maxU < 0) && rowL >= 0 && rowL < A.H;
  const int rowc = min(max(rowL, 0), A.H - 1);
  const uint8_t* dL = A.descL + ((size_t)imL * A.capL + iLc) * 32;
  const uint4 a0 = ((const uint4*)dL)[0], a1 = ((const uint4*)dL)[1];
  const uint8_t* DR = A.descR + (size_t)imR * A.capR * 32;
  int bestDist = TH_HIGH, bestIdx = INT_MAX;
  float bestX = 0.f;  // column of the lane's best right key (it travels with the row list)
  const int* rs = A.row_start + (size_t)f * (A.H + 1);
  const int2* list = A.row_list + (size_t)f * A.list_cap;
  const int c0 = rs[rowc], c1 = alive ? min(rs[rowc + 1], A.list_cap) : c0;
  // two list entries per lane and pass (32 candidates per key), both entries and then both descriptors in flight
  for (int cb = c0; __any(cb < c1); cb += 32) {
    int2 e[2];
    bool in[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int c = cb + gl + 16 * u;
      in[u] = c < c1;
      e[u] = in[u] ? list[c] : make_int2(0, 0);
    }
    int dd[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int j = e[u].x & 0xFFFFF, octR = e[u].x >> 20;
      const float xR = __int_as_float(e[u].y);
      in[u] = in[u] && !(octR < levelL - 1 || octR > levelL + 1) && (xR >= minU && xR <= maxU);
      dd[u] = in[u] ? hamming32(a0, a1, DR + (size_t)j * 32) : TH_HIGH;
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int j = e[u].x & 0xFFFFF;
      if (in[u] && dd[u] < TH_HIGH && lex_less(dd[u], j, bestDist, bestIdx))
        bestDist = dd[u], bestIdx = j, bestX = __int_as_float(e[u].y);
    }
  }
  const int myIdx = bestIdx;
  {  // lexicographic minimum of (distance, index) as ONE key (distance <= 256, index < 2^20), per row of 16 lanes
    const unsigned m = row_min_u32(bestIdx == INT_MAX ? 0xFFFFFFFFu : ((unsigned)bestDist << 20) | (unsigned)bestIdx);
    bestIdx = m == 0xFFFFFFFFu ? INT_MAX : (int)(m & 0xFFFFF);
    bestDist = m == 0xFFFFFFFFu ? INT_MAX : (int)(m >> 20);
  }
  alive = alive && bestIdx != INT_MAX && bestDist < (TH_HIGH + TH_LOW) / 2;
  // ---- sub-pixel refinement by 11 SADs of 11x11 patches at the key's pyramid level
  // the winner's column from the lane of the row that found it
  const unsigned long long own = __ballot(alive && myIdx == bestIdx);
  const unsigned own_g = (unsigned)((own >> (16 * g)) & 0xFFFFull);
  const int src_lane = g * 16 + (own_g ? __builtin_ffs((int)own_g) - 1 : 0);
  const float uR0 = __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane * 4, __float_as_int(bestX)));
  const float sF = 1.0f / A.P.lv[levelL].scale;  // mvInvScaleFactors[octave]
  const float scaleduL = roundf(kL.x * sF), scaledvL = roundf(kL.y * sF);
  const float scaleduR0 = roundf(uR0 * sF);
  const int w = 5, L = 5;
  const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
  const int lvw = A.P.lv[levelL].w, lvh = A.P.lv[levelL].h;
  alive = alive && !(iniu < 0 || endu >= (float)lvw);
  if (!__any(alive)) return;
  int pitchL, pitchR;
  const uint8_t* PL = plane_ptr(A.P, A.IL, imL, levelL, &pitchL);
  const uint8_t* PR = plane_ptr(A.P, A.IR, imR, levelL, &pitchR);
  const int r0 = (int)(scaledvL - w), cL0 = (int)(scaleduL - w), cR0 = (int)(scaleduR0 - w);
  // (addresses of rows that are not refined are clamped into the plane; their values are never used)
  auto clampy = [&](int yy) { return min(max(yy, 0), lvh - 1); };
  auto clampx = [&](int xx) { return min(max(xx, 0), lvw - 1); };
  const int cvL = alive ? (int)PL[(size_t)(r0 + w) * pitchL + cL0 + w] : 0;
  // each lane owns up to eight of the 121 patch positions
  int py[8], px[8], il[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int p = gl + 16 * k;
    py[k] = p / 11, px[k] = p - py[k] * 11;
    il[k] = (alive && p < 121) ? (int)PL[(size_t)clampy(r0 + py[k]) * pitchL + clampx(cL0 + px[k])] - cvL : 0;
  }
  // The 11 shifted 11 x 11 windows of the right image overlap: together they are one 11 x 21 strip (231 bytes); the row
  // of lanes loads it once (15 byte loads per lane) into its LDS slice
  __shared__ uint8_t s_strip[16][11 * 21 + 25];
  uint8_t* strip = s_strip[wave * 4 + g];
  {
    unsigned v[15];
#pragma unroll
    for (int k = 0; k < 15; k++) {
      const int p = gl + 16 * k, row = p / 21, col = p - row * 21;
      v[k] = (alive && p < 231) ? PR[(size_t)clampy(r0 + row) * pitchR + clampx(cR0 - L + col)] : 0;
    }
#pragma unroll
    for (int k = 0; k < 15; k++)
      if (gl + 16 * k < 231) strip[gl + 16 * k] = (uint8_t)v[k];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  int bestS = INT_MAX, bestInc = 0, sads[11];
#pragma unroll
  for (int inc = -L; inc <= L; inc++) {
    const int cvR = strip[w * 21 + inc + L + w];
    int acc = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (gl + 16 * k < 121) {
        const int ir = (int)strip[py[k] * 21 + inc + L + px[k]] - cvR;
        acc += abs(il[k] - ir);
      }
    }
    acc = row_sum_i32(acc);
    sads[inc + L] = acc;
    if (acc < bestS) bestS = acc, bestInc = inc;
  }
  if (!alive || bestInc == -L || bestInc == L) return;
  float dist1 = 0, dist2 = 0, dist3 = 0;
#pragma unroll
  for (int k = 1; k < 10; k++)
    if (k == bestInc + L) dist1 = (float)sads[k - 1], dist2 = (float)sads[k], dist3 = (float)sads[k + 1];
  const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
  if (deltaR < -1 || deltaR > 1) return;
  float bestuR = A.P.lv[levelL].scale * ((float)scaleduR0 + (float)bestInc + deltaR);
  float disparity = uL - bestuR;
  if (disparity >= minD && disparity < maxD) {
    if (disparity <= 0) {
      disparity = 0.01f;
      bestuR = (float)((double)uL - 0.01);
    }
    if (gl == 0) {
      *o_dp = A.bf / disparity;
      *o_ur = bestuR;
      *o_sad = bestS;
    }
  }
}

// Frame.cc:599-610: median of the accepted SADs (element size/2 of the sorted list), reject
// matches with SAD >= 1.5*1.4*median.  One workgroup per frame; the k-th smallest is found by a
// 17-step bisection on the value (SADs are integers below 2^17).
__global__ void __launch_bounds__(256) k_stereo_median(StereoArgs A) {
  // The frame's SADs stay in registers (8 per thread: 2048 keys; more fall back to re-reading), a bisection round is a
  // count per thread, one DPP sum per wavefront and four partial sums through LDS -- was: the SADs re-read from HBM and a
  // 256-entry serial sum by one thread in each of the 17 rounds (31 us for one frame, now a few).
  __shared__ int s_part[2][4];
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int imL = A.l_first + f * A.l_step;
  const int N = min(A.cntL[2 * imL], A.capL);
  int* sad = A.sad + (size_t)f * A.capL;
  constexpr int kR = 8;
  int v[kR];
#pragma unroll
  for (int r = 0; r < kR; r++) {
    const int i = tid + 256 * r;
    v[r] = i < N ? sad[i] : -1;
  }
  const bool spill = N > 256 * kR;
  auto block_count = [&](int bound, int slot) -> int {  // number of SADs s with 0 <= s <= bound
    int c = 0;
#pragma unroll
    for (int r = 0; r < kR; r++) c += (v[r] >= 0 && v[r] <= bound);
    if (spill)
      for (int i = tid + 256 * kR; i < N; i += 256) {
        const int s = sad[i];
        c += (s >= 0 && s <= bound);
      }
    c = wave_sum_i32(c);
    if (lane == 0) s_part[slot][wave] = c;
    __syncthreads();
    return s_part[slot][0] + s_part[slot][1] + s_part[slot][2] + s_part[slot][3];
  };
  const int n = block_count((1 << 30), 0);
  if (n == 0) return;
  const int k = n / 2;  // 0-based rank of the median element
  int lo = 0, hi = (1 << 17) - 1;  // smallest v with count(sad <= v) >= k+1
  int slot = 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const int cnt = block_count(mid, slot);  // (alternating slots: one barrier per round is enough)
    slot ^= 1;
    if (cnt >= k + 1)
      hi = mid;
    else
      lo = mid + 1;
  }
  const float median = (float)lo;
  const float thDist = 1.5f * 1.4f * median;
  for (int i = tid; i < N; i += 256) {
    const int s = sad[i];
    if (s >= 0 && !((float)s < thDist)) {
      A.uright[(size_t)f * A.capL + i] = -1;
      A.depth[(size_t)f * A.capL + i] = -1;
    }
  }
}

static thread_local DevBuf g_row_start, g_row_list;

static int launch_stereo(StereoArgs A, int n_frames, hipStream_t st) {
  int rc;
  float smax = 1.f;
  for (int l = 0; l < A.P.nlevels; l++) smax = std::max(smax, A.P.lv[l].scale);
  A.H = A.P.lv[0].h;
  if (A.capR >= (1 << 20)) {  // the row lists carry the key index in 20 bits
    set_error("ComputeStereoMatches: more than 2^20 keys per image");
    return VIEO_E_CAPACITY;
  }
  A.list_cap = A.capR * (2 * (int)ceilf(2.0f * smax) + 3);
  if ((rc = g_row_start.ensure((size_t)n_frames * (A.H + 1) * 4)) != VIEO_OK) return rc;
  if ((rc = g_row_list.ensure((size_t)n_frames * A.list_cap * 8)) != VIEO_OK) return rc;
  A.row_start = g_row_start.as<int>(), A.row_list = g_row_list.as<int2>();
  if (n_frames <= 16)
    hipLaunchKernelGGL(k_stereo_rows<1024>, dim3(n_frames), dim3(1024), (size_t)(2 * A.H + 1) * 4, st, A);
  else
    hipLaunchKernelGGL(k_stereo_rows<256>, dim3(n_frames), dim3(256), (size_t)(2 * A.H + 1) * 4, st, A);
  hipLaunchKernelGGL(k_stereo_rect, dim3((A.capL + 15) / 16, n_frames), dim3(256), 0, st, A);
  hipLaunchKernelGGL(k_stereo_median, dim3(n_frames), dim3(256), 0, st, A);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

static int same_geometry(const vieo_orb* a, const vieo_orb* b) {
  if (a->w != b->w || a->h != b->h || a->nlevels != b->nlevels || a->w == 0) return 0;
  for (int l = 0; l < a->nlevels; l++)
    if (a->P.lv[l].w != b->P.lv[l].w || a->P.lv[l].h != b->P.lv[l].h ||
        a->P.lv[l].scale != b->P.lv[l].scale)
      return 0;
  return 1;
}

// scratch shared by the host-pointer entry points (serialised by a mutex-free single stream use)
struct MatchScratch {
  DevBuf kpL, kpR, dL, dR, cL, cR, ur, dp, sad, jobs, idx, dist;
};
static thread_local MatchScratch g_ms;

}  // namespace vieo

using namespace vieo;

extern "C" {

int vieo_stereo_match_rectified_batch_device(vieo_orb* e, int n_frames,
                                             const vieo_keypoint* d_keypoints,
                                             const uint8_t* d_descriptors, const int32_t* d_counts,
                                             int capacity, float baseline, float bf,
                                             float* d_uright, float* d_depth) {
  if (!e || n_frames <= 0 || 2 * n_frames > e->last_B || !d_keypoints || !d_descriptors ||
      !d_counts || !d_uright || !d_depth || !(baseline > 0) || !(bf > 0)) {
    set_error("vieo_stereo_match_rectified_batch_device: invalid arguments (the extractor's last "
              "batch must hold 2*n_frames images, left/right interleaved)");
    return VIEO_E_INVALID;
  }
  int rc;
  static thread_local DevBuf sad;
  if ((rc = sad.ensure((size_t)n_frames * capacity * 4)) != VIEO_OK) return rc;
  StereoArgs A;
  A.P = e->P;
  A.IL = A.IR = e->last_imgs;
  A.l_first = 0, A.l_step = 2, A.r_first = 1, A.r_step = 2;
  A.kpL = A.kpR = d_keypoints;
  A.descL = A.descR = d_descriptors;
  A.cntL = A.cntR = d_counts;
  A.capL = A.capR = capacity;
  A.baseline = baseline;
  A.bf = bf;
  A.uright = d_uright;
  A.depth = d_depth;
  A.sad = sad.as<int>();
  return launch_stereo(A, n_frames, e->stream);
}

int vieo_stereo_match_rectified(vieo_orb* left, vieo_orb* right, const vieo_keypoint* h_kpL,
                                const uint8_t* h_descL, int nL, const vieo_keypoint* h_kpR,
                                const uint8_t* h_descR, int nR, float baseline, float bf,
                                float* h_uright, float* h_depth) {
  if (!left || !right || nL < 0 || nR < 0 || !h_uright || !h_depth || !(baseline > 0) || !(bf > 0))
    return VIEO_E_INVALID;
  if (left->last_B < 1 || right->last_B < 1 || !same_geometry(left, right)) {
    set_error("vieo_stereo_match_rectified: both extractors must have processed equally sized "
              "images (their pyramids are read in place)");
    return VIEO_E_INVALID;
  }
  if (nL == 0) return VIEO_OK;
  int rc;
  MatchScratch& S = g_ms;
  const int capL = nL, capR = std::max(nR, 1);
#define ENS(b, n) \
  if ((rc = (b).ensure(n)) != VIEO_OK) return rc
  ENS(S.kpL, (size_t)capL * sizeof(vieo_keypoint));
  ENS(S.kpR, (size_t)capR * sizeof(vieo_keypoint));
  ENS(S.dL, (size_t)capL * 32);
  ENS(S.dR, (size_t)capR * 32);
  ENS(S.cL, 8);
  ENS(S.cR, 8);
  ENS(S.ur, (size_t)capL * 4);
  ENS(S.dp, (size_t)capL * 4);
  ENS(S.sad, (size_t)capL * 4);
#undef ENS
  hipStream_t st = left->stream;
  VIEO_HIP_CHECK(hipStreamSynchronize(right->stream));  // right pyramid complete
  const int cl[2] = {nL, 0}, cr[2] = {nR, 0};
  VIEO_HIP_CHECK(hipMemcpyAsync(S.kpL.p, h_kpL, (size_t)nL * sizeof(vieo_keypoint), hipMemcpyHostToDevice, st));
  VIEO_HIP_CHECK(hipMemcpyAsync(S.dL.p, h_descL, (size_t)nL * 32, hipMemcpyHostToDevice, st));
  if (nR > 0) {
    VIEO_HIP_CHECK(hipMemcpyAsync(S.kpR.p, h_kpR, (size_t)nR * sizeof(vieo_keypoint), hipMemcpyHostToDevice, st));
    VIEO_HIP_CHECK(hipMemcpyAsync(S.dR.p, h_descR, (size_t)nR * 32, hipMemcpyHostToDevice, st));
  }
  VIEO_HIP_CHECK(hipMemcpyAsync(S.cL.p, cl, 8, hipMemcpyHostToDevice, st));
  VIEO_HIP_CHECK(hipMemcpyAsync(S.cR.p, cr, 8, hipMemcpyHostToDevice, st));
  StereoArgs A;
  A.P = left->P;
  A.IL = left->last_imgs;
  A.IR = right->last_imgs;
  A.l_first = 0, A.l_step = 0, A.r_first = 0, A.r_step = 0;
  A.kpL = S.kpL.as<vieo_keypoint>(), A.kpR = S.kpR.as<vieo_keypoint>();
  A.descL = S.dL.as<uint8_t>(), A.descR = S.dR.as<uint8_t>();
  A.cntL = S.cL.as<int>(), A.cntR = S.cR.as<int>();
  A.capL = capL, A.capR = capR;
  A.baseline = baseline, A.bf = bf;
  A.uright = S.ur.as<float>(), A.depth = S.dp.as<float>(), A.sad = S.sad.as<int>();
  if ((rc = launch_stereo(A, 1, st)) != VIEO_OK) return rc;
  VIEO_HIP_CHECK(hipMemcpyAsync(h_uright, S.ur.p, (size_t)nL * 4, hipMemcpyDeviceToHost, st));
  VIEO_HIP_CHECK(hipMemcpyAsync(h_depth, S.dp.p, (size_t)nL * 4, hipMemcpyDeviceToHost, st));
  VIEO_HIP_CHECK(hipStreamSynchronize(st));
  return VIEO_OK;
}

// Frame::ComputeStereoMatches (Frame.cc:451-611) of the frame the two handles have just extracted: keys, descriptors
// and pyramids are read where vieo_orb_extract left them, nothing goes up; uright / depth come back in one block and
// uright stays in the left handle for the frame's projection searches.
int vieo_stereo_match_rectified_resident(vieo_orb* left, vieo_orb* right, float baseline, float bf, float* h_uright,
                                         float* h_depth) {
  if (!left || !right || !h_uright || !h_depth || !(baseline > 0) || !(bf > 0)) return VIEO_E_INVALID;
  if (left->res_n < 0 || right->res_n < 0) {
    set_error("vieo_stereo_match_rectified_resident: a handle holds no frame (vieo_orb_extract first)");
    return VIEO_E_INVALID;
  }
  if (left->last_B < 1 || right->last_B < 1 || !same_geometry(left, right)) {
    set_error("vieo_stereo_match_rectified_resident: both extractors must have processed equally sized images");
    return VIEO_E_INVALID;
  }
  const int nL = left->res_n, capL = vieo_orb_max_keypoints(left), capR = vieo_orb_max_keypoints(right);
  int rc;
  if ((rc = left->d_uright.ensure((size_t)capL * 4)) != VIEO_OK || (rc = left->d_depth.ensure((size_t)capL * 4)) != VIEO_OK ||
      (rc = left->d_sad.ensure((size_t)capL * 4)) != VIEO_OK || (rc = left->h_io.ensure((size_t)capL * 8)) != VIEO_OK)
    return rc;
  left->uright_epoch = left->epoch;
  left->uright_host.clear();
  if (nL == 0) return VIEO_OK;
  hipStream_t st = left->stream;
  VIEO_HIP_CHECK(hipStreamSynchronize(right->stream));  // (its extraction returned synchronised: a formality)
  StereoArgs A;
  A.P = left->P;
  A.IL = left->last_imgs;
  A.IR = right->last_imgs;
  A.l_first = 0, A.l_step = 0, A.r_first = 0, A.r_step = 0;
  A.kpL = left->d_kp.as<vieo_keypoint>(), A.kpR = right->d_kp.as<vieo_keypoint>();
  A.descL = left->d_desc.as<uint8_t>(), A.descR = right->d_desc.as<uint8_t>();
  A.cntL = left->d_counts.as<int>(), A.cntR = right->d_counts.as<int>();
  A.capL = capL, A.capR = capR;
  A.baseline = baseline, A.bf = bf;
  A.uright = left->d_uright.as<float>(), A.depth = left->d_depth.as<float>(), A.sad = left->d_sad.as<int>();
  if ((rc = launch_stereo(A, 1, st)) != VIEO_OK) return rc;
  float* H = (float*)left->h_io.p;
  VIEO_HIP_CHECK(hipMemcpyAsync(H, left->d_uright.p, (size_t)nL * 4, hipMemcpyDeviceToHost, st));
  VIEO_HIP_CHECK(hipMemcpyAsync(H + capL, left->d_depth.p, (size_t)nL * 4, hipMemcpyDeviceToHost, st));
  VIEO_HIP_CHECK(hipStreamSynchronize(st));
  memcpy(h_uright, H, (size_t)nL * 4);
  memcpy(h_depth, H + capL, (size_t)nL * 4);
  left->uright_host.assign(H, H + nL);
  return VIEO_OK;
}

int vieo_hamming_knn2_batch_device(const uint8_t* d_descriptors, const int32_t* h_counts,
                                   int capacity, const int32_t* h_pairs, int n_pairs,
                                   int32_t* d_idx, int32_t* d_dist, void* stream) {
  if (!d_descriptors || !h_counts || !h_pairs || n_pairs <= 0 || !d_idx || !d_dist || capacity <= 0 || capacity > 65535)
    return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  std::vector<Knn2Job> jobs(n_pairs);
  int max_nq = 0;
  for (int p = 0; p < n_pairs; p++) {
    const int qi = h_pairs[2 * p], ti = h_pairs[2 * p + 1];
    // descriptors[num_mono:] of both cameras (Frame.cc:620-628)
    const int qm = h_counts[2 * qi + 1], tm = h_counts[2 * ti + 1];
    jobs[p].q = d_descriptors + ((size_t)qi * capacity + qm) * 32;
    jobs[p].t = d_descriptors + ((size_t)ti * capacity + tm) * 32;
    jobs[p].nq = std::max(h_counts[2 * qi] - qm, 0);
    jobs[p].nt = std::max(h_counts[2 * ti] - tm, 0);
    jobs[p].out_off = p * capacity;
    max_nq = std::max(max_nq, jobs[p].nq);
  }
  MatchScratch& S = g_ms;
  if ((rc = S.jobs.ensure(jobs.size() * sizeof(Knn2Job))) != VIEO_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  VIEO_HIP_CHECK(hipMemcpyAsync(S.jobs.p, jobs.data(), jobs.size() * sizeof(Knn2Job),
                                hipMemcpyHostToDevice, st));
  VIEO_HIP_CHECK(hipStreamSynchronize(st));  // jobs vector goes out of scope
  if (max_nq > 0) {
    Knn2Src K;
    memset(&K, 0, sizeof(K));
    K.jobs = S.jobs.as<Knn2Job>();
    knn2_launch(K, max_nq, n_pairs, 1, d_idx, d_dist, st);
  }
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

int vieo_hamming_knn2_rig_batch_device(const uint8_t* d_descriptors, const int32_t* d_counts, int capacity, int n_cams,
                                       int n_frames, int32_t* d_idx, int32_t* d_dist, void* stream) {
  if (!d_descriptors || !d_counts || capacity <= 0 || capacity > 65535 || n_cams < 2 || n_cams > 4 || n_frames <= 0 || !d_idx ||
      !d_dist)
    return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  return knn2_rig_launch(d_descriptors, d_counts, capacity, n_cams, n_frames, d_idx, d_dist, (hipStream_t)stream);
}

int vieo_hamming_knn2(const uint8_t* h_query, int nq, const uint8_t* h_train, int nt,
                      int32_t* h_idx, int32_t* h_dist) {
  if (nq < 0 || nt < 0 || (nq > 0 && (!h_query || !h_idx || !h_dist))) return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  if (nq == 0) return VIEO_OK;
  if (nt > 65535) {  // the packed (distance, index) word of k_knn2
    set_error("vieo_hamming_knn2: %d train rows, at most 65535", nt);
    return VIEO_E_CAPACITY;
  }
  MatchScratch& S = g_ms;
#define ENS(b, n) \
  if ((rc = (b).ensure(n)) != VIEO_OK) return rc
  ENS(S.dL, (size_t)nq * 32);
  ENS(S.dR, (size_t)std::max(nt, 1) * 32);
  ENS(S.idx, (size_t)nq * 8);
  ENS(S.dist, (size_t)nq * 8);
  ENS(S.jobs, sizeof(Knn2Job));
#undef ENS
  VIEO_HIP_CHECK(hipMemcpy(S.dL.p, h_query, (size_t)nq * 32, hipMemcpyHostToDevice));
  if (nt > 0) VIEO_HIP_CHECK(hipMemcpy(S.dR.p, h_train, (size_t)nt * 32, hipMemcpyHostToDevice));
  Knn2Job j;
  j.q = S.dL.as<uint8_t>(), j.t = S.dR.as<uint8_t>(), j.nq = nq, j.nt = nt, j.out_off = 0;
  VIEO_HIP_CHECK(hipMemcpy(S.jobs.p, &j, sizeof(j), hipMemcpyHostToDevice));
  Knn2Src K;
  memset(&K, 0, sizeof(K));
  K.jobs = S.jobs.as<Knn2Job>();
  knn2_launch(K, nq, 1, 1, S.idx.as<int32_t>(), S.dist.as<int32_t>(), 0);
  VIEO_HIP_CHECK(hipGetLastError());
  VIEO_HIP_CHECK(hipMemcpy(h_idx, S.idx.p, (size_t)nq * 8, hipMemcpyDeviceToHost));
  VIEO_HIP_CHECK(hipMemcpy(h_dist, S.dist.p, (size_t)nq * 8, hipMemcpyDeviceToHost));
  return VIEO_OK;
}

}  // extern "C"
