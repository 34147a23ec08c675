// orb_extractor.hip -- MI355X (gfx950) ORB extractor behind the C-ABI of include/vieo_hot.h.
//
// Replaces VIEO_SLAM::ORBextractor (reference: src/ORBextractor.cc, include/ORBextractor.h).
// Frames are processed in batches that stay resident in HBM; per batch the launch sequence is
//   k_resize x (nlevels-1)  ComputePyramid            (ORBextractor.cc:1060-1081)
//   k_fast                  per-cell FAST 20/7 + NMS   (:723-779)  one wavefront per cell,
//                                                       cell tile staged in LDS
//   k_quadtree              DistributeOctTree          (:518-721)  one workgroup per (image,level)
//   k_blur                  GaussianBlur 7x7 s=2       (:1012-1015) LDS-tiled separable Q8.8
//   k_describe              IC_Angle + steered BRIEF   (:55-127, 1024-1054) one wavefront / key
//   k_lapping               lapping-area reorder       (:1041-1052) only when pvLappingArea
// All arithmetic is integer or strictly-ordered float (-ffp-contract=off), so keypoints and
// descriptors are bit-exact against oracle/ (tests/test_orb_parity.py).
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>

#include "../../include/vieo_orb_pattern_31.h"
#include "common.h"
#include "sincosf_exact.h"
#include "wave_ops.h"

#define QT_DEVICE
#include "quadtree.inl"

#include "orb_internal.h"

namespace vieo {

// ------------------------------------------------------------------ pyramid (cv::resize)
// xtab: {sx0, sx1, a0, a1}; ytab: {sy0, sy1, b0, b1}; INTER_LINEAR 8U fixed point:
// H = S[sx0]*a0 + S[sx1]*a1 (x2048), D = (((b0*(H0>>4))>>16) + ((b1*(H1>>4))>>16) + 2) >> 2.
// One workgroup = 256 output columns x kResizeRows output rows.  The source rows it needs are staged
// in LDS with coalesced dword loads (the 4 taps per pixel then cost LDS byte reads, not global ones).
#ifndef VIEO_RESIZE_ROWS
#define VIEO_RESIZE_ROWS 32
#endif
static const int kResizeRows = VIEO_RESIZE_ROWS;
#ifndef VIEO_QT_GROUPS
#define VIEO_QT_GROUPS {1, 64}  // k_quadtree: level 0 and levels 1.. are two launches, each with its own LDS size ({64} 0.37, {1,64} 0.28, {1,3,64} 0.31, a launch per level 0.37 ms per 1024 images: tools/ab_qt_split.sh)
#endif

__global__ void __launch_bounds__(256, 8)  // 51 instead of 74 registers, no spills: 8 wavefronts per SIMD
k_resize(OrbParams P, int l, ImgSet I, const short4* __restrict__ xtab,
         const short4* __restrict__ ytab, int lds_pitch) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const LevelDesc& D = P.lv[l];
  const int b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int dy0 = blockIdx.y * kResizeRows, dy1 = min(dy0 + kResizeRows, D.h);
  const int dx0 = blockIdx.x * 256, dx1 = min(dx0 + 256, D.w);
  int spitch;
  const uint8_t* src = plane_ptr(P, I, b, l - 1, &spitch);
  const short4* xt = xtab + D.xtab_off;
  const short4* yt = ytab + D.ytab_off;
  const int sy_lo = yt[dy0].x, nrows = yt[dy1 - 1].y - sy_lo + 1;
  const int sx_lo = xt[dx0].x & ~3, ndw = ((xt[dx1 - 1].y + 4) >> 2) - (sx_lo >> 2);
  {  // all source rows of the band in flight at once (12 rows x 2 dword columns per lane), then the stores: a
     // dependent load -> store loop costs one HBM round trip per row
    unsigned v[12][2];
#pragma unroll
    for (int k = 0; k < 12; k++) {
      const int r = wave + 4 * k;
      const unsigned* row = (const unsigned*)(src + (size_t)(sy_lo + r) * spitch + sx_lo);
      if (r < nrows && lane < ndw) v[k][0] = row[lane];
      if (r < nrows && lane + 64 < ndw) v[k][1] = row[lane + 64];
    }
#pragma unroll
    for (int k = 0; k < 12; k++) {
      const int r = wave + 4 * k;
      if (r < nrows && lane < ndw) *(unsigned*)(smem + r * lds_pitch + 4 * lane) = v[k][0];
      if (r < nrows && lane + 64 < ndw) *(unsigned*)(smem + r * lds_pitch + 4 * (lane + 64)) = v[k][1];
    }
    for (int r = wave; r < nrows; r += 4) {  // whatever a wider / taller band leaves over
      const unsigned* row = (const unsigned*)(src + (size_t)(sy_lo + r) * spitch + sx_lo);
      for (int c = lane + (r < 48 ? 128 : 0); c < ndw; c += 64) *(unsigned*)(smem + r * lds_pitch + 4 * c) = row[c];
    }
  }
  __syncthreads();
  const int dx4 = dx0 + lane * 4;
  if (dx4 >= D.w) return;
  // per lane: the four columns' taps as LDS offsets inside a band row, and which of the four bytes exist
  int x0[4], x1[4], a0[4], a1[4];
  unsigned keep = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const short4 t = xt[min(dx4 + j, D.w - 1)];
    x0[j] = t.x - sx_lo, x1[j] = t.y - sx_lo, a0[j] = t.z, a1[j] = t.w;
    if (dx4 + j < D.w) keep |= 0xFFu << (8 * j);
  }
  uint8_t* dst = I.pyr + (size_t)b * I.pyr_img + D.off;
  // the row's table entry is the same for the whole wavefront: with a wave-uniform row index it is a scalar load
  // (a per-lane global load followed by its wait was one L2 round trip per output row)
  // the row's table entry is the same for the whole wavefront: with a wave-uniform row index it is a scalar load
  // (a per-lane global load followed by its wait was one L2 round trip per output row).  Measured and dropped:
  // consecutive rows per wavefront with the horizontal sums of the last two source rows kept in registers (a source
  // row serves 1.67 output rows) -- 0.81 ms per 1024 images against 0.78 for this form, whose 16 LDS reads per row
  // are all in flight at once.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  for (int dy = dy0 + wave_u; dy < dy1; dy += 4) {
    const short4 y = yt[dy];
    const uint8_t* r0 = smem + (y.x - sy_lo) * lds_pitch;
    const uint8_t* r1 = smem + (y.y - sy_lo) * lds_pitch;
    const int b0 = y.z, b1 = y.w;
    unsigned out = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int h0 = r0[x0[j]] * a0[j] + r0[x1[j]] * a1[j];
      const int h1 = r1[x0[j]] * a0[j] + r1[x1[j]] * a1[j];
      // h >> 4 <= 255 * 2048 / 16 fits 16 bits: saying so lets the compiler use the 24-bit multiply (v_mul_lo_u32 is quarter rate)
      const int v = (((b0 * (int)(unsigned short)(h0 >> 4)) >> 16) + ((b1 * (int)(unsigned short)(h1 >> 4)) >> 16) + 2) >> 2;
      out |= (unsigned)(v & 0xFF) << (8 * j);
    }
    *(unsigned*)(dst + (size_t)dy * D.pitch + dx4) = out & keep;
  }
}

// ---- two pyramid levels per launch (round 6).  The resize chain was 47 % of the extractor's HBM traffic: every level is
// written, then read again as the next resize's source.  Here a workgroup produces a tile of level l + 1 (l = the second
// level of the pair) from the tile of level l it has just computed and still holds in LDS: level l is read from HBM by
// FAST and the descriptor kernel only, no longer by the resize, and the chain is 4 launches instead of 7.
//   tile     kR2W x kR2H pixels of level l + 1;
//   region   the level-l pixels its taps touch (rows [yt2[dy0].x, yt2[dy1 - 1].y], columns likewise, the left edge rounded
//            down to a dword): at most 256 x kR2Rows, computed from level l - 1 exactly as k_resize does (same table-driven
//            fixed point, same bytes) into LDS;
//   owned    the part of the region that THIS workgroup writes to level l in HBM: from its region's start to the next
//            tile's region start (dword-aligned in x, so no dword of the plane has two writers); neighbouring regions overlap
//            by one or two rows / up to five columns, which are computed twice and written once.
// The host plans the pairs (orb_plan) and keeps k_resize for a last odd level and for geometries whose regions do not fit.
static const int kR2W = 200, kR2H = 26, kR2Rows = 36;
__global__ void __launch_bounds__(256, 4)
k_resize2(OrbParams P, int l, ImgSet I, const short4* __restrict__ xtab, const short4* __restrict__ ytab, int lds_pitch,
          int l1_off) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* l1 = smem + l1_off;  // the level-l region: kR2Rows rows of 256 bytes (+ 4 of padding per row)
  constexpr int kL1P = 260;
  const LevelDesc& D1 = P.lv[l];
  const LevelDesc& D2 = P.lv[l + 1];
  const int b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const short4* xt1 = xtab + D1.xtab_off;
  const short4* yt1 = ytab + D1.ytab_off;
  const short4* xt2 = xtab + D2.xtab_off;
  const short4* yt2 = ytab + D2.ytab_off;
  // ---- the tile of level l + 1, its region of level l and what of it this workgroup owns
  const int ex0 = blockIdx.x * kR2W, ex1 = min(ex0 + kR2W, D2.w);
  const int ey0 = blockIdx.y * kR2H, ey1 = min(ey0 + kR2H, D2.h);
  const bool last_x = ex1 == D2.w, last_y = ey1 == D2.h;
  const int rx0 = blockIdx.x == 0 ? 0 : (xt2[ex0].x & ~3), rx1 = last_x ? D1.w : min((int)xt2[ex1 - 1].y + 1, D1.w);
  const int ry0 = blockIdx.y == 0 ? 0 : (int)yt2[ey0].x, ry1 = last_y ? D1.h : min((int)yt2[ey1 - 1].y + 1, D1.h);
  const int ox1 = last_x ? D1.w : (xt2[ex1].x & ~3), oy1 = last_y ? D1.h : (int)yt2[ey1].x;
  // ---- level l - 1 rows of the region into LDS (as k_resize)
  int spitch;
  const uint8_t* src = plane_ptr(P, I, b, l - 1, &spitch);
  const int sy_lo = yt1[ry0].x, nrows = yt1[ry1 - 1].y - sy_lo + 1;
  const int sx_lo = xt1[rx0].x & ~3, ndw = ((xt1[rx1 - 1].y + 4) >> 2) - (sx_lo >> 2);
  {
    unsigned v[12][2];
#pragma unroll
    for (int k = 0; k < 12; k++) {
      const int r = wave + 4 * k;
      const unsigned* row = (const unsigned*)(src + (size_t)(sy_lo + r) * spitch + sx_lo);
      if (r < nrows && lane < ndw) v[k][0] = row[lane];
      if (r < nrows && lane + 64 < ndw) v[k][1] = row[lane + 64];
    }
#pragma unroll
    for (int k = 0; k < 12; k++) {
      const int r = wave + 4 * k;
      if (r < nrows && lane < ndw) *(unsigned*)(smem + r * lds_pitch + 4 * lane) = v[k][0];
      if (r < nrows && lane + 64 < ndw) *(unsigned*)(smem + r * lds_pitch + 4 * (lane + 64)) = v[k][1];
    }
    for (int r = wave; r < nrows; r += 4) {
      const unsigned* row = (const unsigned*)(src + (size_t)(sy_lo + r) * spitch + sx_lo);
      for (int c = lane + (r < 48 ? 128 : 0); c < ndw; c += 64) *(unsigned*)(smem + r * lds_pitch + 4 * c) = row[c];
    }
  }
  __syncthreads();
  // ---- level l: the region into LDS, its owned part to HBM
  {
    const int dx4 = rx0 + lane * 4;
    if (dx4 < rx1) {
      int x0[4], x1[4], a0[4], a1[4];
      unsigned keep = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const short4 t = xt1[min(dx4 + j, D1.w - 1)];
        x0[j] = t.x - sx_lo, x1[j] = t.y - sx_lo, a0[j] = t.z, a1[j] = t.w;
        if (dx4 + j < D1.w) keep |= 0xFFu << (8 * j);
      }
      uint8_t* dst = I.pyr + (size_t)b * I.pyr_img + D1.off;
      const bool own_x = dx4 < ox1;  // (ox1 is a multiple of 4 or the plane's width: whole dwords)
      const int wave_u = __builtin_amdgcn_readfirstlane(wave);
      for (int dy = ry0 + wave_u; dy < ry1; dy += 4) {
        const short4 y = yt1[dy];
        const uint8_t* r0 = smem + (y.x - sy_lo) * lds_pitch;
        const uint8_t* r1 = smem + (y.y - sy_lo) * lds_pitch;
        const int b0 = y.z, b1 = y.w;
        unsigned out = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int h0 = r0[x0[j]] * a0[j] + r0[x1[j]] * a1[j];
          const int h1 = r1[x0[j]] * a0[j] + r1[x1[j]] * a1[j];
          const int v = (((b0 * (int)(unsigned short)(h0 >> 4)) >> 16) + ((b1 * (int)(unsigned short)(h1 >> 4)) >> 16) + 2) >> 2;
          out |= (unsigned)(v & 0xFF) << (8 * j);
        }
        out &= keep;
        *(unsigned*)(l1 + (dy - ry0) * kL1P + 4 * lane) = out;
        if (own_x && dy < oy1) *(unsigned*)(dst + (size_t)dy * D1.pitch + dx4) = out;
      }
    }
  }
  __syncthreads();
  // ---- level l + 1: the tile from the region in LDS
  const int dx4 = ex0 + lane * 4;
  if (dx4 >= ex1) return;
  int x0[4], x1[4], a0[4], a1[4];
  unsigned keep = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const short4 t = xt2[min(dx4 + j, D2.w - 1)];
    x0[j] = t.x - rx0, x1[j] = t.y - rx0, a0[j] = t.z, a1[j] = t.w;
    if (dx4 + j < D2.w) keep |= 0xFFu << (8 * j);
  }
  uint8_t* dst2 = I.pyr + (size_t)b * I.pyr_img + D2.off;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  for (int dy = ey0 + wave_u; dy < ey1; dy += 4) {
    const short4 y = yt2[dy];
    const uint8_t* r0 = l1 + (y.x - ry0) * kL1P;
    const uint8_t* r1 = l1 + (y.y - ry0) * kL1P;
    const int b0 = y.z, b1 = y.w;
    unsigned out = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int h0 = r0[x0[j]] * a0[j] + r0[x1[j]] * a1[j];
      const int h1 = r1[x0[j]] * a0[j] + r1[x1[j]] * a1[j];
      const int v = (((b0 * (int)(unsigned short)(h0 >> 4)) >> 16) + ((b1 * (int)(unsigned short)(h1 >> 4)) >> 16) + 2) >> 2;
      out |= (unsigned)(v & 0xFF) << (8 * j);
    }
    *(unsigned*)(dst2 + (size_t)dy * D2.pitch + dx4) = out & keep;
  }
}

// Workgroups are dealt to the 8 XCDs round-robin by linear id, and each XCD has its own L2.  Neighbouring
// cells / tiles share 128-byte lines of the plane, so they should share an L2: the launch index is
// mapped so that runs of G consecutive items stay on one XCD while the runs themselves are still
// interleaved over the XCDs (a contiguous eighth per XCD would unbalance them: pyramid levels differ
// in work per cell).  The grid holds 8 * G * ceil(n / (8 G)) blocks.
#ifndef VIEO_XCD_RUN
#define VIEO_XCD_RUN 32
#endif
static const int kXcdRun = VIEO_XCD_RUN;
__device__ __forceinline__ int xcd_grouped(int bid, int G) {
  const int xcd = bid & 7, q = bid >> 3;
  return ((q / G) * 8 + xcd) * G + q % G;
}

// ------------------------------------------------------------------ FAST-9/16 per cell
// Threshold-free corner strength r = max over the 16 arcs of 9 contiguous ring pixels of
// min(v - x) (dark arcs) and of min(x - v) (bright arcs).  cv::FAST(t) declares a corner iff
// r > t and cornerScore<16> is then r - 1, for every t: one evaluation serves both thresholds.
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

__device__ __forceinline__ int fast_strength(const uint8_t* t, int p) {
  const int v = t[0];
  int d[16];
  d[0] = v - t[3 * p];
  d[1] = v - t[3 * p + 1];
  d[2] = v - t[2 * p + 2];
  d[3] = v - t[p + 3];
  d[4] = v - t[3];
  d[5] = v - t[-p + 3];
  d[6] = v - t[-2 * p + 2];
  d[7] = v - t[-3 * p + 1];
  d[8] = v - t[-3 * p];
  d[9] = v - t[-3 * p - 1];
  d[10] = v - t[-2 * p - 2];
  d[11] = v - t[-p - 3];
  d[12] = v - t[-3];
  d[13] = v - t[p - 3];
  d[14] = v - t[2 * p - 2];
  d[15] = v - t[3 * p - 1];
  int lo3[16], hi3[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    lo3[k] = min3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
    hi3[k] = max3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
  }
  int A = -256, B = 256;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    A = max(A, min3i(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]));
    B = min(B, max3i(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]));
  }
  return max(max(A, -B), 0);
}

#ifdef VIEO_FAST_STATS  // tools/fast_stats.py: what the cells of a workload cost (a private build, not the product)
__device__ unsigned long long g_fast_stats[8];
#define FAST_STAT(i, v) do { if (lane == 0) atomicAdd(&g_fast_stats[i], (unsigned long long)(v)); } while (0)
#else
#define FAST_STAT(i, v) do { } while (0)
#endif

// Necessary condition for strength > t on the 4 compass points of the ring (positions 0, 4, 8, 12):
// an arc of 9 contiguous ring pixels contains position 0 or 8 and position 4 or 12, all of one
// polarity.  Darker: x < v - t, brighter: x > v + t.  Each comparison is one v_cmp into a lane mask.
__device__ __forceinline__ bool fast_compass(const uint8_t* t, int p, int th) {
  const int v = t[0], lo = v - th, hi = v + th;
  const int x0 = t[3 * p], x4 = t[3], x8 = t[-3 * p], x12 = t[-3];
  const bool dk = ((x0 < lo) | (x8 < lo)) & ((x4 < lo) | (x12 < lo));
  const bool br = ((x0 > hi) | (x8 > hi)) & ((x4 > hi) | (x12 > hi));
  return dk | br;
}

// Ordering point for ONE wavefront working on its own slice of LDS: its LDS operations complete in order, so all that
// is needed is that earlier stores have been issued and waited for before later loads are scheduled.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The three passes of one cell on its tile in LDS (the tile must be complete and visible): compass test +
// compaction, exact strength of the survivors, 3x3 non-maximum suppression; writes the cell's keys and count.
__device__ __forceinline__ void fast_cell(const OrbParams& P, const CellDesc& cd, int b, int c, uint8_t* tile, uint8_t* sc,
                                          unsigned short* cand, int cand_cap, int tpitch, int iniTh, int minTh, int lane,
                                          unsigned* __restrict__ cell_keys, int* __restrict__ cell_counts) {
  const int x0a = cd.x0 & ~3;
  const int vw = cd.cw - 6, vh = cd.ch - 6;
  const int npx = (vw > 0 && vh > 0) ? vw * vh : 0;
  const int sp = vw + 2;
  if (npx > 0)  // score_bytes is a multiple of 16 and covers (vh + 2) * sp
    for (int idx = lane; idx < ((vh + 2) * sp + 15) / 16; idx += 64) ((uint4*)sc)[idx] = make_uint4(0, 0, 0, 0);
  wave_sync();
  const int xo = cd.x0 - x0a;
  unsigned* out = cell_keys + ((size_t)b * P.ncells + c) * P.cell_cap;
  int base = 0;
  // cv::FAST at iniThFAST, and only for a cell without any corner again at minThFAST
  // (ORBextractor.cc:752-762).  The strengths are threshold-free, so the second round only adds
  // the pixels between the two thresholds.
  for (int round = 0; round < 2 && base == 0; round++) {
    if (round == 1 && minTh >= iniTh) break;
    const int th = round == 0 ? iniTh : minTh;
    int na = 0, nc = 0;  // cand[0, nc): corners (strength > th); cand[nc, na): compass survivors still to be scored
    bool overflow = false;
    FAST_STAT(round, 1);
    // ---- pass B: exact strength of the pending survivors cand[nc, na); those above the threshold are compacted
    // in place behind the corners already there (still row-major: a wavefront reads its 64 entries before it writes
    // any).  Runs once after pass A, and inside pass A whenever the list is about to exceed its LDS capacity.
    auto pass_b = [&]() {
      int w = nc;
      FAST_STAT(3, (na - nc + 63) / 64);
      FAST_STAT(6, na - nc);
      for (int i0 = nc; i0 < na; i0 += 64) {
        const int i = i0 + lane;
        bool pass = false;
        int p = 0;
        if (i < na) {
          p = cand[i];
          const int y = p >> 6, x = p & 63;
          const int r = fast_strength(tile + (y + 3) * tpitch + (x + 3 + xo), tpitch);
          sc[(y + 1) * sp + x + 1] = (uint8_t)r;
          pass = r > th;
        }
        const unsigned long long m = __ballot(pass);
        if (pass) cand[w + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)p;
        w += __popcll(m);
      }
      nc = na = w;
    };
    // ---- pass A: compass test (see fast_compass) on FOUR horizontally adjacent pixels per lane.  The groups of
    // four are numbered row-major over the cell (g = y * ng + x / 4) and a step takes 64 consecutive ones, so all
    // lanes are busy whatever the cell width (a 36-pixel cell has 9 groups per row, not a power of two).  The
    // centre row and the rows 3 above / below are read as dwords and realigned with v_alignbyte (the byte offset
    // is the same for every lane); the test itself stays on whole dwords, four pixels per operation, with the
    // full-rate 32-bit add / and / bitop3 (measured, tools/ubench/valu_rate.hip: the packed 16-bit, perm, min / max
    // and three-operand forms issue at half that rate on gfx950):
    //   lo = C -sat t, hi = C +sat t per byte;  dark <=> (U < lo | D < lo) & (L < lo | R < lo),
    //   bright <=> (U > hi | D > hi) & (L > hi | R > hi);  bytewise unsigned a < b from d = (a | H) - (b & ~H):
    //   bit 7 of the byte = (~a & b) | (~(a ^ b) & ~d), one v_bitop3 (H = 0x80808080).
    // Ordered compaction of the surviving pixels (y << 6 | x).
    if (npx > 0) {
      // centre byte of pixel x sits at 4 * (lx4 + kq) + s; s is the same for the whole cell: one copy of the loop per
      // value, chosen by a scalar branch (as a lane value it cost a chain of exec-mask branches per step)
      const int s = __builtin_amdgcn_readfirstlane((3 + xo) & 3), kq = (3 + xo) >> 2;
      const int ng = (vw + 3) >> 2, G = ng * vh;
      const int qy = 64 / ng, rx = 64 - qy * ng;        // a step of 64 groups = qy rows and rx groups
      const unsigned H = 0x80808080u, T = (unsigned)th * 0x01010101u;
      const unsigned Tl = T & ~H;
      // bytewise a < b (bit 7 of every byte): see above; 0x4D = truth table of (~a & b) | (~(a ^ b) & ~d) over (a, b, d)
      auto ltu = [&](unsigned a, unsigned bb) -> unsigned {
        const unsigned d = (a | H) - (bb & ~H);
        return __builtin_amdgcn_bitop3_b32(a, bb, d, 0x4D);
      };
      auto pass_a = [&](auto SC) {
        constexpr int S = decltype(SC)::value;
        int y = lane / ng, lx4 = lane - y * ng;
        // byte offset of the lane's group in the tile, advanced by adds (a step = qy rows + rx groups, one more row
        // when the group index wraps); the three rows it reads are uniform bases + this offset
        int toff = y * tpitch + 4 * lx4;
        const int step_off = qy * tpitch + 4 * rx, wrap_off = tpitch - 4 * ng;
        const uint8_t* base_u = tile + 4 * kq;
        const uint8_t* base_c = base_u + 3 * tpitch;
        const uint8_t* base_d = base_u + 6 * tpitch;
        // the last group of a row may hang over the cell: its bytes beyond the last pixel never pass
        const unsigned Htail = (unsigned)(0x80808080ull & ((1ull << (8 * (vw - 4 * (ng - 1)))) - 1ull));
        for (int g0 = 0; g0 < G; g0 += 64) {
          if (na + 256 > cand_cap) {  // a step appends up to 256 entries: score what is pending first (uniform branch)
            wave_sync();
            pass_b();
            wave_sync();
            if (na + 256 > cand_cap) {
              overflow = true;
              break;
            }
          }
          unsigned m4 = 0;
          const int x0 = 4 * lx4;
          if (g0 + lane < G) {
            const unsigned* rc = (const unsigned*)(base_c + toff);
            const unsigned* ru = (const unsigned*)(base_u + toff);
            const unsigned* rd = (const unsigned*)(base_d + toff);
            unsigned C, L, R, U, D;
            if constexpr (S == 0) {
              const unsigned wm = rc[-1], w0 = rc[0], w1 = rc[1];
              C = w0, U = ru[0], D = rd[0], L = __builtin_amdgcn_alignbyte(w0, wm, 1), R = __builtin_amdgcn_alignbyte(w1, w0, 3);
            } else if constexpr (S == 1) {
              const unsigned wm = rc[-1], w0 = rc[0], w1 = rc[1];
              C = __builtin_amdgcn_alignbyte(w1, w0, 1), U = __builtin_amdgcn_alignbyte(ru[1], ru[0], 1),
              D = __builtin_amdgcn_alignbyte(rd[1], rd[0], 1), L = __builtin_amdgcn_alignbyte(w0, wm, 2), R = w1;
            } else if constexpr (S == 2) {
              const unsigned wm = rc[-1], w0 = rc[0], w1 = rc[1], w2 = rc[2];
              C = __builtin_amdgcn_alignbyte(w1, w0, 2), U = __builtin_amdgcn_alignbyte(ru[1], ru[0], 2),
              D = __builtin_amdgcn_alignbyte(rd[1], rd[0], 2), L = __builtin_amdgcn_alignbyte(w0, wm, 3),
              R = __builtin_amdgcn_alignbyte(w2, w1, 1);
            } else {
              const unsigned w0 = rc[0], w1 = rc[1], w2 = rc[2];
              C = __builtin_amdgcn_alignbyte(w1, w0, 3), U = __builtin_amdgcn_alignbyte(ru[1], ru[0], 3),
              D = __builtin_amdgcn_alignbyte(rd[1], rd[0], 3), L = w0, R = __builtin_amdgcn_alignbyte(w2, w1, 2);
            }
            // lo = C -sat t: bytes with C >= t keep C - t, the others become 0
            const unsigned ge = ~ltu(C, T) & H;                 // bit 7: C >= t
            const unsigned gem = (ge - (ge >> 7)) | ge;          // 0xFF in those bytes
            const unsigned dif = ((C | H) - Tl) ^ ((C ^ ~T) & H);  // bytewise C - t (mod 256)
            const unsigned lo = dif & gem;
            // hi = C +sat t: bytes whose sum carries become 255
            const unsigned suml = (C & ~H) + Tl;                 // low 7 bits + carry into bit 7
            const unsigned sum = suml ^ ((C ^ T) & H);           // bytewise C + t (mod 256)
            const unsigned cy = __builtin_amdgcn_bitop3_b32(C, T, suml, 0xE8) & H;  // carry out = majority(C7, t7, carry in)
            const unsigned hi = sum | ((cy - (cy >> 7)) | cy);
            const unsigned dk = (ltu(U, lo) | ltu(D, lo)) & (ltu(L, lo) | ltu(R, lo));
            const unsigned br = (ltu(hi, U) | ltu(hi, D)) & (ltu(hi, L) | ltu(hi, R));
            const unsigned f = (dk | br) & (lx4 == ng - 1 ? Htail : H);  // bit 7 of byte j: pixel x0 + j passes
            // bits 7, 15, 23, 31 -> bits 0..3: one multiply (the partial products land on distinct bits, no carries)
            m4 = ((f >> 7) * 0x01020408u) >> 24;
          }
          // positions: inclusive scan of the lanes' counts on DPP (four shifted adds inside a row of 16 lanes, zeros
          // shift in; two row broadcasts), minus the lane's own count
          const int cnt = __popc(m4);
          int inc = cnt;
          inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xF, 0xF, true);  // row_shr:1
          inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xF, 0xF, true);  // row_shr:2
          inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xF, 0xF, true);  // row_shr:4
          inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xF, 0xF, true);  // row_shr:8
          inc += VIEO_DPP(0, inc, VIEO_DPP_ROW_BCAST15, 0xA);
          inc += VIEO_DPP(0, inc, VIEO_DPP_ROW_BCAST31, 0xC);
          int pos = na + inc - cnt;
          const unsigned key = (unsigned)((y << 6) | x0);
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (m4 & (1u << j)) cand[pos++] = (unsigned short)(key + j);
          na += __builtin_amdgcn_readlane(inc, 63);
          // the next 64 groups
          lx4 += rx, y += qy, toff += step_off;
          if (lx4 >= ng) lx4 -= ng, y++, toff += wrap_off;
        }
      };
      if (s == 0) pass_a(std::integral_constant<int, 0>());
      else if (s == 1) pass_a(std::integral_constant<int, 1>());
      else if (s == 2) pass_a(std::integral_constant<int, 2>());
      else pass_a(std::integral_constant<int, 3>());
    }
    wave_sync();
    if (!overflow) pass_b();
    else {
      // More corners than the list holds (noise, synthetic patterns: never a camera image).  The list is dropped and
      // the strength of EVERY pixel of the cell is evaluated; pass C then scans the strength tile itself.  Same
      // result: the compass test is a necessary condition, so the pixels it removes have strength <= th, which
      // pass C reads as 0 either way.
      FAST_STAT(2, 1);
      for (int i0 = 0; i0 < npx; i0 += 64) {
        const int i = i0 + lane;
        if (i < npx) {
          const int y = i / vw, x = i - y * vw;
          sc[(y + 1) * sp + x + 1] = (uint8_t)fast_strength(tile + (y + 3) * tpitch + (x + 3 + xo), tpitch);
        }
      }
    }
    wave_sync();
    // ---- pass C: 3x3 non-maximum suppression over the corners (still in row-major order)
    const int nC = overflow ? npx : nc;
    FAST_STAT(4, (nC + 63) / 64);
    for (int i0 = 0; i0 < nC; i0 += 64) {
      const int i = i0 + lane;
      bool keep = false;
      unsigned key = 0;
      if (i < nC) {
        int x, y;
        if (overflow) {
          y = i / vw, x = i - y * vw;
        } else {
          const int p = cand[i];
          y = p >> 6, x = p & 63;
        }
        const uint8_t* s = sc + (y + 1) * sp + x + 1;
        const int r = s[0];
        if (r > th) {
          const int sv = r - 1;
#define NB(o) ((s[o] > th) ? (int)s[o] - 1 : 0)
          keep = sv > NB(1) && sv > NB(-1) && sv > NB(-sp - 1) && sv > NB(-sp) && sv > NB(-sp + 1) &&
                 sv > NB(sp - 1) && sv > NB(sp) && sv > NB(sp + 1);
#undef NB
          key = (unsigned)(x + 3 + cd.offx) | ((unsigned)(y + 3 + cd.offy) << 12) |
                ((unsigned)sv << 24);
        }
      }
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        if (pos < P.cell_cap) out[pos] = key;
      }
      base += __popcll(m);
    }
    wave_sync();
  }
  FAST_STAT(7, base);
  if (lane == 0) cell_counts[(size_t)b * P.ncells + c] = min(base, P.cell_cap);
}

// One wavefront (= one workgroup) per cell, no workgroup barrier anywhere.  What the kernel lives on is the number of
// resident wavefronts: it is a chain of short dependent phases (record -> plane -> tile -> three passes), and a CU takes
// as many of these workgroups as their LDS allows.  Measured on 1024 images: 6.5 KB of LDS per cell (candidate list
// sized for the worst case) 2.05 ms, 4.8 KB (list capped, see fast_cell) 1.78 ms, 8.8 KB 2.2 - 2.4 ms -- which is also
// why a second tile buffer for prefetching the next cell (by LDS-DMA or through registers) lost more than it hid.
#ifndef VIEO_FAST_WAVES
#define VIEO_FAST_WAVES 1  // wavefronts (= cells) per workgroup; each keeps its own LDS slice, no workgroup barrier
#endif
__global__ void __launch_bounds__(64 * VIEO_FAST_WAVES)
k_fast(OrbParams P, ImgSet I, const CellDesc* __restrict__ cells, unsigned* __restrict__ cell_keys,
       int* __restrict__ cell_counts, int iniTh, int minTh, int tpitch, int tile_bytes,
       int score_bytes, int cand_cap, int n_images, int lds_per_wave) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_all[];
  const int lane = threadIdx.x & 63;
  const int wave_in_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t* smem = smem_all + wave_in_wg * lds_per_wave;
  const int item = xcd_grouped(blockIdx.x, kXcdRun) * VIEO_FAST_WAVES + wave_in_wg;  // item = image * ncells + cell
  if (item >= P.ncells * n_images) return;
  const int b = item / P.ncells, c = item - b * P.ncells;
  const CellDesc cd = cells[c];
  int pitch;
  const uint8_t* src = plane_ptr(P, I, b, cd.level, &pitch);
  uint8_t* tile = smem;
  uint8_t* sc = smem + tile_bytes;
  unsigned short* cand = (unsigned short*)(smem + tile_bytes + score_bytes);
  const int x0a = cd.x0 & ~3;
  const int ndw = ((cd.x0 + cd.cw + 3) >> 2) - (x0a >> 2);
  {  // ndw <= 17: 16 dword columns x 4 rows per step, the odd 17th column afterwards
    const int dc = lane & 15, dr = lane >> 4;
    // Every row of the cell in flight at once (a dependent load -> store loop costs one HBM round trip per four
    // rows).  Addresses are a uniform base (scalar registers) plus a 32-bit lane offset that grows by one add per
    // step, and the steps that lie wholly inside the cell are taken on a uniform condition: per-step 64-bit
    // address arithmetic and lane masks were 120 of this kernel's vector instructions.
    const uint8_t* gbase = src + (size_t)cd.y0 * pitch + x0a;  // uniform
    const unsigned gstep = 4u * (unsigned)pitch, tstep = 4u * (unsigned)tpitch;
    unsigned goff = (unsigned)dr * (unsigned)pitch + 4u * dc;
    uint8_t* t = tile + dr * tpitch + 4 * dc;
    const int kfull = cd.ch >> 2;  // steps whose four rows all exist (uniform)
    if (dc < ndw) {
      unsigned v[16];
#pragma unroll
      for (int k = 0; k < 16; k++) {
        if (k < kfull)
          v[k] = *(const unsigned*)(gbase + goff);
        else if (dr + 4 * k < cd.ch)
          v[k] = *(const unsigned*)(gbase + goff);
        goff += gstep;
      }
#pragma unroll
      for (int k = 0; k < 16; k++) {
        if (k < kfull)
          *(unsigned*)t = v[k];
        else if (dr + 4 * k < cd.ch)
          *(unsigned*)t = v[k];
        t += tstep;
      }
      for (int r = dr + 64; r < cd.ch; r += 4, goff += gstep, t += tstep) *(unsigned*)t = *(const unsigned*)(gbase + goff);
    }
    for (int idx = lane; idx < cd.ch * (ndw - 16); idx += 64) {  // columns 16.. (cells wider than 61)
      const int r = idx / (ndw - 16), dcol = 16 + idx % (ndw - 16);
      *(unsigned*)(tile + r * tpitch + 4 * dcol) =
          *(const unsigned*)(src + (size_t)(cd.y0 + r) * pitch + x0a + 4 * dcol);
    }
  }
  fast_cell(P, cd, b, c, tile, sc, cand, cand_cap, tpitch, iniTh, minTh, lane, cell_keys, cell_counts);
}

// ------------------------------------------------------------------ quadtree
__global__ void __launch_bounds__(1024)
k_quadtree(OrbParams P, const unsigned* __restrict__ cell_keys,
           const int* __restrict__ cell_counts, unsigned* __restrict__ keys,
           unsigned short* __restrict__ kslot, unsigned char* __restrict__ kq,
           unsigned* __restrict__ sel, int* __restrict__ sel_count, int ncap_max, int scap_max, int l0) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int l = l0 + blockIdx.x, b = blockIdx.y;
  const LevelDesc& D = P.lv[l];
  // carve LDS (8-byte items first)
  uint8_t* q = smem;
  QtMem m;
  m.scanA = (unsigned long long*)q;
  q += sizeof(unsigned long long) * scap_max;
  m.scanB = (unsigned long long*)q;
  q += sizeof(unsigned long long) * scap_max;
  m.cnt = (int*)q;
  q += 4 * ncap_max;
  m.cc = (int*)q;
  q += 16 * ncap_max;
  m.best = (unsigned*)q;
  q += 4 * ncap_max;
  m.cand_size[0] = (int*)q;
  q += 4 * ncap_max;
  m.cand_size[1] = (int*)q;
  q += 4 * ncap_max;
  m.s = (QtShared*)q;
  q += 64;
  m.x0 = (short*)q;
  q += 2 * ncap_max;
  m.y0 = (short*)q;
  q += 2 * ncap_max;
  m.x1 = (short*)q;
  q += 2 * ncap_max;
  m.y1 = (short*)q;
  q += 2 * ncap_max;
  m.child = (unsigned short*)q;
  q += 8 * ncap_max;
  m.mark = (unsigned short*)q;
  q += 2 * ncap_max;
  m.list[0] = (unsigned short*)q;
  q += 2 * ncap_max;
  m.list[1] = (unsigned short*)q;
  q += 2 * ncap_max;
  m.cand_slot[0] = (unsigned short*)q;
  q += 2 * ncap_max;
  m.cand_slot[1] = (unsigned short*)q;
  q += 2 * ncap_max;
  m.order = (unsigned short*)q;
  q += 2 * ncap_max;
  m.flag = (unsigned char*)q;
  m.ncap = D.ncap;
  m.scap = scap_max;

  const int tid = threadIdx.x;
  const int ncell = D.cell_end - D.cell_begin;
  const int* cnts = cell_counts + (size_t)b * P.ncells + D.cell_begin;
  // ---- concatenate the per-cell lists in cell order = vToDistributeKeys order
  for (int i = tid; i < ncell; i += blockDim.x) m.scanA[i] = (unsigned long long)cnts[i];
  for (int i = tid; i < D.nIni; i += blockDim.x) m.cnt[i] = 0;
  __syncthreads();
  unsigned long long* inc = qt_scan(m.scanA, m.scanB, ncell);
  if (tid == 0) m.s->K = ncell > 0 ? (int)inc[ncell - 1] : 0;
  __syncthreads();
  const int K = m.s->K;
  unsigned* kk = keys + (size_t)b * P.keys_per_image + D.key_off;
  unsigned short* ks = kslot + (size_t)b * P.keys_per_image + D.key_off;
  unsigned char* kqq = kq + (size_t)b * P.keys_per_image + D.key_off;
  {  // one thread per output key, four keys in flight: the cell of key t is found by bisection on the inclusive
     // scan in LDS (a wavefront per cell walked ~90 cells one dependent HBM round trip after the other)
    const unsigned* cells0 = cell_keys + ((size_t)b * P.ncells + D.cell_begin) * P.cell_cap;
    const int nth = blockDim.x;
    for (int t0 = tid; t0 < K; t0 += 4 * nth) {
      unsigned key[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int t = t0 + u * nth;
        key[u] = 0;
        if (t < K) {
          int lo = 0, hi = ncell - 1;  // first cell with inc[c] > t
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((int)inc[mid] > t)
              hi = mid;
            else
              lo = mid + 1;
          }
          const int first = lo ? (int)inc[lo - 1] : 0;
          key[u] = cells0[(size_t)lo * P.cell_cap + (t - first)];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int t = t0 + u * nth;
        if (t < K) {
          kk[t] = key[u];
          // vpIniNodes[kp.pt.x / hX]  (ORBextractor.cc:549)
          const int ini = (int)((float)QT_KEY_X(key[u]) / D.hX);
          ks[t] = (unsigned short)ini;
          atomicAdd(&m.cnt[ini], 1);
        }
      }
    }
  }
  __syncthreads();
  unsigned* out = sel + (size_t)b * P.sel_per_image + D.sel_off;
  const int n = qt_distribute(m, kk, ks, kqq, D.regW, D.regH, D.nIni, D.hX, D.nfeat, out);
  if (tid == 0) sel_count[b * kMaxLevels + l] = m.s->error ? -1 : n;
}

// VIEO_ORB_FUSED (default 1): blur + descriptor in one kernel (k_describe_fused), 0: k_blur + k_describe
static bool orb_fused() {
  static const bool on = [] {
    const char* e = getenv("VIEO_ORB_FUSED");
    return !e || atoi(e) != 0;
  }();
  return on;
}

// ------------------------------------------------------------------ Gaussian blur 7x7, sigma 2
// Q8.8 kernel {18,34,48,56,48,34,18}: horizontal pass exact in 16 bits, vertical pass
// (sum + 2^15) >> 16; BORDER_REFLECT_101 on the level itself.
__device__ __forceinline__ int reflect101(int p, int len) {
  if (p < 0) p = -p;
  if (p >= len) p = 2 * (len - 1) - p;
  return p;
}


// Tile = 64 x kBlurTH outputs.  The source tile (kBlurTH + 6 rows x 72 bytes, column 0 <-> x = ox - 4) is staged
// as dwords; the horizontal pass produces 4 outputs per work item with v_alignbyte + 2x
// v_dot4_u32_u8 each (exact 16-bit results, stored packed), the vertical pass 4 outputs per work
// item from 7 x 8-byte LDS reads.  The sum is the same integer as OpenCV's, only the order of the
// two exact passes' additions differs.
__global__ void __launch_bounds__(256)
k_blur(OrbParams P, ImgSet I, const BlurTile* __restrict__ tiles, int n_tiles, int n_images) {
  constexpr int SP = 72, SH = kBlurTH + 6;  // source pitch (bytes), rows
  __shared__ __attribute__((aligned(16))) uint8_t s_src[SH * SP];
  __shared__ __attribute__((aligned(16))) unsigned s_h[(SH / 2) * kBlurTW];  // row pairs, see below
  const int item = xcd_grouped(blockIdx.x, kXcdRun);  // item = image * n_tiles + tile
  if (item >= n_tiles * n_images) return;
  const int b = item / n_tiles;
  const BlurTile t = tiles[item - b * n_tiles];
  const int tid = threadIdx.x;
  const LevelDesc& D = P.lv[t.level];
  int pitch;
  const uint8_t* src = plane_ptr(P, I, b, t.level, &pitch);
  const int ox = t.tx * kBlurTW, oy = t.ty * kBlurTH;
  const bool interior = ox >= 4 && ox + 68 <= D.w && oy >= 3 && oy + kBlurTH + 3 <= D.h &&
                        ((pitch & 3) == 0) && ((((uintptr_t)src) & 3) == 0);
  if (interior) {
    constexpr int NIT = (SH * (SP / 4) + 255) / 256;  // all loads of the tile in flight, then the stores
    unsigned v[NIT];
#pragma unroll
    for (int k = 0; k < NIT; k++) {
      const int idx = tid + 256 * k, r = idx / (SP / 4), c4 = idx - r * (SP / 4);
      if (idx < SH * (SP / 4)) v[k] = *(const unsigned*)(src + (size_t)(oy + r - 3) * pitch + ox - 4 + 4 * c4);
    }
#pragma unroll
    for (int k = 0; k < NIT; k++)
      if (tid + 256 * k < SH * (SP / 4)) ((unsigned*)s_src)[tid + 256 * k] = v[k];
  } else {
    // border tiles (a third of all tiles over the pyramid): whole dwords wherever the four columns lie inside
    // the image, REFLECT_101 byte by byte only for the dwords that straddle a border
    const bool aligned = ((pitch & 3) == 0) && ((((uintptr_t)src) & 3) == 0);
    constexpr int NIT = (SH * (SP / 4) + 255) / 256;
    unsigned v[NIT];
#pragma unroll
    for (int k = 0; k < NIT; k++) {
      const int idx = tid + 256 * k, r = idx / (SP / 4), c4 = idx - r * (SP / 4);
      v[k] = 0;
      if (idx < SH * (SP / 4)) {
        const int gy = reflect101(min(oy + r - 3, D.h + 2), D.h);
        const int gx0 = ox - 4 + 4 * c4;
        const uint8_t* row = src + (size_t)gy * pitch;
        if (aligned && gx0 >= 0 && gx0 + 4 <= D.w)
          v[k] = *(const unsigned*)(row + gx0);
        else {
#pragma unroll
          for (int j = 0; j < 4; j++) v[k] |= (unsigned)row[reflect101(min(gx0 + j, D.w + 2), D.w)] << (8 * j);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NIT; k++)
      if (tid + 256 * k < SH * (SP / 4)) ((unsigned*)s_src)[tid + 256 * k] = v[k];
  }
  __syncthreads();
  const unsigned K0123 = 18u | (34u << 8) | (48u << 16) | (56u << 24);
  const unsigned K456 = 48u | (34u << 8) | (18u << 16);
  // horizontal pass: one item = two source rows x four columns; the 16-bit row sums of vertically
  // adjacent rows share a dword (row 2p low half, row 2p+1 high half) so the vertical pass can use
  // v_dot2_u32_u16
#ifndef VIEO_BLUR_AB
#define VIEO_BLUR_AB 0  // timing experiment only (wrong results): 1 = no arithmetic, the tile goes from LDS straight out
#endif
  for (int idx = tid; idx < ((VIEO_BLUR_AB & 1) ? 0 : (SH / 2) * (kBlurTW / 4)); idx += 256) {
    const int pr = idx / (kBlurTW / 4), g = idx - pr * (kBlurTW / 4);
    unsigned h[2][4];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const unsigned* w = (const unsigned*)(s_src + (2 * pr + q) * SP) + g;
      const unsigned w0 = w[0], w1 = w[1], w2 = w[2];
      // output column c = 4g + j uses source bytes 4g + 1 + j .. 4g + 7 + j
      const unsigned a0 = __builtin_amdgcn_alignbyte(w1, w0, 1), b0 = __builtin_amdgcn_alignbyte(w2, w1, 1);
      const unsigned a1 = __builtin_amdgcn_alignbyte(w1, w0, 2), b1 = __builtin_amdgcn_alignbyte(w2, w1, 2);
      const unsigned a2 = __builtin_amdgcn_alignbyte(w1, w0, 3), b2 = __builtin_amdgcn_alignbyte(w2, w1, 3);
      h[q][0] = __builtin_amdgcn_udot4(b0, K456, __builtin_amdgcn_udot4(a0, K0123, 0u, false), false);
      h[q][1] = __builtin_amdgcn_udot4(b1, K456, __builtin_amdgcn_udot4(a1, K0123, 0u, false), false);
      h[q][2] = __builtin_amdgcn_udot4(b2, K456, __builtin_amdgcn_udot4(a2, K0123, 0u, false), false);
      h[q][3] = __builtin_amdgcn_udot4(w2, K456, __builtin_amdgcn_udot4(w1, K0123, 0u, false), false);
    }
    uint4 o;
    o.x = h[0][0] | (h[1][0] << 16), o.y = h[0][1] | (h[1][1] << 16);
    o.z = h[0][2] | (h[1][2] << 16), o.w = h[0][3] | (h[1][3] << 16);
    *(uint4*)(s_h + pr * kBlurTW + 4 * g) = o;
  }
  __syncthreads();
  // vertical pass: one item = two output rows x four columns (exactly one item per thread)
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  uint8_t* dst = I.blur + (size_t)b * I.blur_img + D.boff;
  for (int idx = tid; idx < (kBlurTH / 2) * (kBlurTW / 4); idx += 256) {
    const int q = idx / (kBlurTW / 4), c4 = (idx - q * (kBlurTW / 4)) * 4;
    const int gy = oy + 2 * q, gx = ox + c4;
    if (gy >= D.h || gx >= D.w) continue;
#if VIEO_BLUR_AB & 1
    *(unsigned*)(dst + (size_t)gy * D.pitch + gx) = *(const unsigned*)(s_src + (2 * q + 3) * SP + c4 + 4);
    if (gy + 1 < D.h) *(unsigned*)(dst + (size_t)(gy + 1) * D.pitch + gx) = *(const unsigned*)(s_src + (2 * q + 4) * SP + c4 + 4);
    continue;
#endif
    // even row 2q: source rows 2q .. 2q+6; odd row 2q+1: source rows 2q+1 .. 2q+7
    const unsigned We[4] = {18u | (34u << 16), 48u | (56u << 16), 48u | (34u << 16), 18u};
    const unsigned Wo[4] = {18u << 16, 34u | (48u << 16), 56u | (48u << 16), 34u | (18u << 16)};
    unsigned e[4] = {1u << 15, 1u << 15, 1u << 15, 1u << 15}, o[4] = {1u << 15, 1u << 15, 1u << 15, 1u << 15};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint4 v = *(const uint4*)(s_h + (q + k) * kBlurTW + c4);
      const unsigned vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        e[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, vv[j]), __builtin_bit_cast(us2, We[k]), e[j], false);
        o[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, vv[j]), __builtin_bit_cast(us2, Wo[k]), o[j], false);
      }
    }
    const unsigned oe = ((e[0] >> 16) & 0xFFu) | (((e[1] >> 16) & 0xFFu) << 8) | (((e[2] >> 16) & 0xFFu) << 16) |
                        (((e[3] >> 16) & 0xFFu) << 24);
    const unsigned oo = ((o[0] >> 16) & 0xFFu) | (((o[1] >> 16) & 0xFFu) << 8) | (((o[2] >> 16) & 0xFFu) << 16) |
                        (((o[3] >> 16) & 0xFFu) << 24);
    *(unsigned*)(dst + (size_t)gy * D.pitch + gx) = oe;
    if (gy + 1 < D.h) *(unsigned*)(dst + (size_t)(gy + 1) * D.pitch + gx) = oo;
  }
}

// ------------------------------------------------------------------ orientation + descriptor
__device__ __forceinline__ int wave_sum(int v) { return wave_sum_i32(v); }

// cv::fastAtan2 (degrees), float arithmetic in the published order
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
  const float eps = (float)2.2204460492503131e-16;
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + eps);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + eps);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// (key, level) of every selected key point in the level-concatenated order (ORBextractor.cc:1005-1054), and the
// per-image counts: k_describe then starts from ONE record load.
__global__ void __launch_bounds__(256)
k_desc_index(OrbParams P, const unsigned* __restrict__ sel, const int* __restrict__ sel_count,
             uint2* __restrict__ krec, int out_cap, int* __restrict__ counts) {
  const int b = blockIdx.y, g = blockIdx.x * 256 + threadIdx.x;
  int level = -1, idx = 0, total = 0;
  for (int l = 0; l < P.nlevels; l++) {
    const int n = max(sel_count[b * kMaxLevels + l], 0);
    if (level < 0 && g < total + n) {
      level = l;
      idx = g - total;
    }
    total += n;
  }
  if (g == 0) {
    counts[2 * b] = min(total, out_cap);
    counts[2 * b + 1] = 0;
  }
  if (g >= P.kp_cap) return;
  unsigned key = 0;
  if (level >= 0) key = sel[(size_t)b * P.sel_per_image + P.lv[level].sel_off + idx];
  krec[(size_t)b * P.kp_cap + g] = make_uint2(key, (unsigned)level);
}

#ifndef VIEO_DESC_KPW
#define VIEO_DESC_KPW 1  // key points per wavefront (records, then ALL patches in flight together): 1 / 2 / 3 / 4 measured 0.93 / 0.95 / 1.11 / 1.32 ms per 1024 images -- the kernel is not waiting for its two round trips
#endif
__global__ void __launch_bounds__(256)
k_describe(OrbParams P, ImgSet I, const uint2* __restrict__ krec, const int* __restrict__ pattern,
           vieo_keypoint* __restrict__ kp_out, uint8_t* __restrict__ desc_out, int out_cap,
           int groups_per_image, int n_images) {
  constexpr int KPW = VIEO_DESC_KPW;
  // all key points of an image on one XCD: their overlapping patches then share that XCD's L2
  const int item = xcd_grouped(blockIdx.x, groups_per_image);
  const int b = item / groups_per_image;
  if (b >= n_images) return;
  const int lane = threadIdx.x & 63;
  // wave-uniform for the compiler: the key's record, its level and the level's descriptor then come through scalar
  // loads instead of three dependent per-lane global loads
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g0 = ((item - b * groups_per_image) * (blockDim.x >> 6) + wv) * KPW;
  const int gmax = min(P.kp_cap, out_cap);
  if (g0 >= gmax) return;
  // (key, level) of the key points, written by k_desc_index: one load instead of the level search + the key load.
  // A wavefront takes KPW consecutive key points: the chain record -> patches -> arithmetic is two dependent round
  // trips per WAVEFRONT, and with one key point each 4.9 M wavefronts per step queued for them.
  uint2 kr[KPW];
#pragma unroll
  for (int u = 0; u < KPW; u++) kr[u] = g0 + u < gmax ? krec[(size_t)b * P.kp_cap + g0 + u] : make_uint2(0, 0xFFFFFFFFu);
  // The two patches a key point reads -- 31 x 31 of the level image for the orientation, 39 x 39 of
  // the blurred level for the steered pattern (|rotated offset| <= 19 = EDGE_THRESHOLD) -- are
  // staged in this wavefront's LDS slice as whole dwords, row by row.
  constexpr int AP = 40, BR = 19, BP = 44;
  __shared__ __attribute__((aligned(16))) uint8_t s_patch[4][KPW][31 * AP + (2 * BR + 1) * BP];
  // the lane's four pattern words: issued ahead of the patch loads, so that they are there when the angle is
  int pt4[4];
#pragma unroll
  for (int gq = 0; gq < 4; gq++) pt4[gq] = pattern[gq * 64 + lane];  // x0 | y0<<8 | x1<<16 | y1<<24 (int8 each)
  int cxs[KPW], cys[KPW], xas[KPW], xbs[KPW];
  unsigned va[KPW][8], vb[KPW][10];
  const int c = lane & 15, r0 = lane >> 4;  // 16 dword columns x 4 rows per step
#pragma unroll
  for (int u = 0; u < KPW; u++) {
    const int level = (int)kr[u].y;
    if (level < 0) continue;  // (wave-uniform)
    const unsigned key = kr[u].x;
    const LevelDesc& D = P.lv[level];
    const int cx = QT_KEY_X(key) + (kEdge - 3), cy = QT_KEY_Y(key) + (kEdge - 3);
    int pitch;
    const uint8_t* img = plane_ptr(P, I, b, level, &pitch);
    const int xa = (cx - kHalfPatch) & ~3, xb = (cx - BR) & ~3;
    cxs[u] = cx, cys[u] = cy, xas[u] = xa, xbs[u] = xb;
    const uint8_t* bl0 = I.blur + (size_t)b * I.blur_img + D.boff;
    const int bp = D.pitch;
    const uint8_t* ga = img + (size_t)(cy - kHalfPatch + r0) * pitch + xa + 4 * c;
    const uint8_t* gb = bl0 + (size_t)(cy - BR + r0) * bp + xb + 4 * c;
    // both patches in flight at once, then the LDS stores (one memory round trip instead of two)
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (c < 9 && r0 + 4 * k < 31) va[u][k] = *(const unsigned*)(ga + (size_t)(4 * k) * pitch);
#pragma unroll
    for (int k = 0; k < 10; k++)
      if (c < 11 && r0 + 4 * k < 2 * BR + 1) vb[u][k] = *(const unsigned*)(gb + (size_t)(4 * k) * bp);
  }
#pragma unroll
  for (int u = 0; u < KPW; u++) {
    if ((int)kr[u].y < 0) continue;
    uint8_t* sa = s_patch[wv][u];
    uint8_t* sb = sa + 31 * AP;
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (c < 9 && r0 + 4 * k < 31) *(unsigned*)(sa + (r0 + 4 * k) * AP + 4 * c) = va[u][k];
#pragma unroll
    for (int k = 0; k < 10; k++)
      if (c < 11 && r0 + 4 * k < 2 * BR + 1) *(unsigned*)(sb + (r0 + 4 * k) * BP + 4 * c) = vb[u][k];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int u = 0; u < KPW; u++) {
    const int level = (int)kr[u].y;
    if (level < 0) continue;
    const int g = g0 + u;
    const unsigned key = kr[u].x;
    const LevelDesc& D = P.lv[level];
    const int cx = cxs[u], cy = cys[u], xa = xas[u], xb = xbs[u];
    const uint8_t* sa = s_patch[wv][u];
    const uint8_t* sb = sa + 31 * AP;
    // ---- IC_Angle: two lanes per row of the radius-15 disc
    int m10 = 0, m01 = 0;
    if (lane < 62) {
      const int v = (lane >> 1) - kHalfPatch;
      const int d = P.umax[v < 0 ? -v : v];
      const uint8_t* row = sa + (v + kHalfPatch) * AP + (cx - xa);
      const int u0 = (lane & 1) ? 0 : -d, u1 = (lane & 1) ? d : -1;
      int sI = 0;
      // at most 16 pixels per lane (d <= 15): unrolled, so the byte reads are all issued before the first one is used
      // (as a loop with a lane-dependent trip count every iteration waited for its own LDS read)
#pragma unroll
      for (int t = 0; t < 16; t++) {
        const int uu = u0 + t;
        if (uu <= u1) {
          const int val = row[uu];
          m10 += uu * val;
          sI += val;
        }
      }
      m01 = v * sI;
    }
    m10 = wave_sum(m10);
    m01 = wave_sum(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    // ---- steered BRIEF on the blurred plane (ORBextractor.cc:83-127)
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float a, bsin;
    vieo_sincosf_exact(angle * factorPI, &bsin, &a);
    const uint8_t* bl = sb + BR * BP + (cx - xb);
    unsigned long long bits[4];
#pragma unroll
    for (int gq = 0; gq < 4; gq++) {
      const int pt = pt4[gq];
      const float x0 = (float)(signed char)(pt & 0xFF), y0 = (float)(signed char)((pt >> 8) & 0xFF);
      const float x1 = (float)(signed char)((pt >> 16) & 0xFF), y1 = (float)(signed char)((pt >> 24) & 0xFF);
      const int t0 = bl[__float2int_rn(x0 * bsin + y0 * a) * BP + __float2int_rn(x0 * a - y0 * bsin)];
      const int t1 = bl[__float2int_rn(x1 * bsin + y1 * a) * BP + __float2int_rn(x1 * a - y1 * bsin)];
      bits[gq] = __ballot(t0 < t1);
    }
    if (lane < 4) ((unsigned long long*)(desc_out + ((size_t)b * out_cap + g) * 32))[lane] = bits[lane];
    if (lane == 0) {
      vieo_keypoint k;
      const float fx = (float)cx, fy = (float)cy;
      k.x = level ? fx * D.scale : fx;
      k.y = level ? fy * D.scale : fy;
      k.size = (float)D.patch;
      k.angle = angle;
      k.response = (float)QT_KEY_R(key);
      k.octave = level;
      k.class_id = -1;
      kp_out[(size_t)b * out_cap + g] = k;
    }
  }
}

// ------------------------------------------------------------------ orientation + blur + descriptor, one kernel
// The descriptor reads the blurred level at 512 steered positions within 18 pixels of the key point (the pattern's
// largest radius is 18.38, cvRound'ed per coordinate): a 37 x 37 patch of the blurred level, which is the 7 x 7 Gaussian
// of a 43 x 43 patch of the level itself.  A frame's ~1200 key points per image cover about twice the pyramid's pixels
// that way, so the blur's arithmetic doubles -- but the blurred pyramid (1.1 MB per image written, then gathered back)
// never exists: the wavefront stages the raw patch once (it holds the orientation's 31 x 31 disc too), blurs it in LDS
// with k_blur's two exact passes (same integers: Q8.8 rows, (sum + 2^15) >> 16 columns) and samples the result there.
// Raw pixels outside the level are its REFLECT_101 continuation, which is what GaussianBlur reads at the border of the
// (borderless) clone the reference blurs (ORBextractor.cc:1128-1131); blurred pixels outside the level -- a key point
// 16 or 17 pixels from the border whose steered offset reaches 17 or 18 -- are out of bounds of that clone in the
// reference (it reads whatever lies there); here they are the blur of the continuation.
#ifndef VIEO_FUSED_AB
#define VIEO_FUSED_AB 0  // timing experiments only (wrong results): 1 = no blur passes, 2 = no patch loads, 4 = no angle / sampling
#endif
constexpr int kFR = 18, kFRaw = kFR + 3;                // blurred / raw patch radius
constexpr int kFRawRows = 2 * kFRaw + 1, kFRawPitch = 48;  // 43 rows of 12 dwords
constexpr int kFHPairs = (kFRawRows + 1) / 2, kFHCols = 40;  // horizontal sums: 22 row pairs x 40 columns (dwords)
constexpr int kFBlRows = 2 * kFR + 1, kFBlPitch = 40;       // blurred patch: 37 rows x 40 bytes
constexpr int kFWaveLds = kFRawRows * kFRawPitch + 16 + kFHPairs * kFHCols * 4;  // the blurred patch overlays the raw one
static_assert(kFBlRows * kFBlPitch <= kFRawRows * kFRawPitch, "the blurred patch fits where the raw one was");

#ifndef VIEO_FUSED_KPW
#define VIEO_FUSED_KPW 4  // key points per wavefront in large batches, one after the other (the next one's patch loads in flight meanwhile)
#endif
// (a frame or two at a time -- the one-call tracker, the per-image host entry -- there are fewer key points than wavefront
// slots: one key point per wavefront then, 604 wavefronts of four were 28 us per stereo frame against 20 for the two kernels)
static inline int fused_kpw(int n_images) { return n_images >= 32 ? VIEO_FUSED_KPW : 1; }
__global__ void __launch_bounds__(256)
k_describe_fused(OrbParams P, ImgSet I, const uint2* __restrict__ krec, const int* __restrict__ pattern,
                 vieo_keypoint* __restrict__ kp_out, uint8_t* __restrict__ desc_out, int out_cap,
                 int groups_per_image, int n_images, int KPW) {
  // all key points of an image on one XCD: their overlapping patches then share that XCD's L2
  const int item = xcd_grouped(blockIdx.x, groups_per_image);
  const int b = item / groups_per_image;
  if (b >= n_images) return;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // A wavefront takes KPW consecutive key points one after the other: the wave's start-up, the record and pattern loads
  // are paid once, and the NEXT key point's patch loads are in flight while this one is blurred and sampled (with one key
  // point per wavefront 1.7 of the kernel's 5.7 ms were start-up and 0.9 the wait for the patch).
  const int g0 = ((item - b * groups_per_image) * (blockDim.x >> 6) + wv) * KPW;
  const int gmax = min(P.kp_cap, out_cap);
  if (g0 >= gmax) return;
  __shared__ __attribute__((aligned(16))) uint8_t s_all[4][kFWaveLds];
  int pt4[4];
#pragma unroll
  for (int gq = 0; gq < 4; gq++) pt4[gq] = pattern[gq * 64 + lane];  // x0 | y0<<8 | x1<<16 | y1<<24 (int8 each)
  uint8_t* s_raw = s_all[wv];
  unsigned* s_h = (unsigned*)(s_raw + kFRawRows * kFRawPitch + 16);
  // ---- the raw patch: 43 rows, staged so that its column 0 is byte 0 of the LDS row (12 dwords).  A step = 4 rows x 16
  // lanes: lane c loads dword c of the aligned row (13 of them hold the patch), takes dword c + 1 from its neighbour
  // (DPP row shift) and stores the two realigned; all loads in flight, then (an iteration later) the stores
  constexpr int NLD = (kFRawRows + 3) / 4;
  unsigned v[NLD];
  bool interior = false;
  int sh = 0;
  const int c = lane & 15, r0 = lane >> 4;
  auto rec_of = [&](int u) { return (u < KPW && g0 + u < gmax) ? krec[(size_t)b * P.kp_cap + g0 + u] : make_uint2(0u, 0xFFFFFFFFu); };
  auto fetch = [&](const uint2 kr) {  // the loads of key point kr's patch (wave-uniform control flow)
    const int level = (int)kr.y;
    if (level < 0) return;
    const LevelDesc& D = P.lv[level];
    const int cx = QT_KEY_X(kr.x) + (kEdge - 3), cy = QT_KEY_Y(kr.x) + (kEdge - 3);
    int pitch;
    const uint8_t* img = plane_ptr(P, I, b, level, &pitch);
    const int x_lo = cx - kFRaw, y_lo = cy - kFRaw;
    const int xa = x_lo & ~3;
    sh = x_lo - xa;  // the patch's column 0 is byte `sh` of the aligned rows
    interior = x_lo >= 0 && y_lo >= 0 && cx + kFRaw < D.w && cy + kFRaw < D.h && xa + 52 <= pitch && ((pitch & 3) == 0) &&
               ((((uintptr_t)img) & 3) == 0);
    if (VIEO_FUSED_AB & 2) {
#pragma unroll
      for (int k = 0; k < NLD; k++) v[k] = kr.x + k;
      interior = false;
    } else if (interior) {
      const uint8_t* gp = img + (size_t)(y_lo + r0) * pitch + xa + 4 * c;
#pragma unroll
      for (int k = 0; k < NLD; k++) {
        v[k] = 0;
        if (c < 13 && r0 + 4 * k < kFRawRows) v[k] = *(const unsigned*)(gp + (size_t)(4 * k) * pitch);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NLD; k++) {
        v[k] = 0;
        if (c < 12 && r0 + 4 * k < kFRawRows) {
          const uint8_t* row = img + (size_t)reflect101(min(max(y_lo + r0 + 4 * k, -(D.h - 1)), 2 * D.h - 2), D.h) * pitch;
#pragma unroll
          for (int jj = 0; jj < 4; jj++) {
            const int x = min(max(x_lo + 4 * c + jj, -(D.w - 1)), 2 * D.w - 2);
            v[k] |= (unsigned)row[reflect101(x, D.w)] << (8 * jj);
          }
        }
      }
    }
  };
  uint2 kr = rec_of(0);
  fetch(kr);
#pragma unroll 1
  for (int u = 0; u < KPW; u++) {
    const uint2 cur = kr;
    const int level = (int)cur.y;
    if (level < 0) break;  // (the records behind an image's last key point are all empty)
    const unsigned key = cur.x;
    const int g = g0 + u;
    const LevelDesc& D = P.lv[level];
    const int cx = QT_KEY_X(key) + (kEdge - 3), cy = QT_KEY_Y(key) + (kEdge - 3);
    if (interior) {
#pragma unroll
      for (int k = 0; k < NLD; k++) {
        const unsigned nxt = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v[k], 0x101, 0xF, 0xF, false);  // row_shl:1 = lane + 1's
        v[k] = __builtin_amdgcn_alignbyte(nxt, v[k], sh);
      }
    }
#pragma unroll
    for (int k = 0; k < NLD; k++)
      if (c < 12 && r0 + 4 * k < kFRawRows) *(unsigned*)(s_raw + (r0 + 4 * k) * kFRawPitch + 4 * c) = v[k];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    kr = rec_of(u + 1);
    fetch(kr);  // the next key point's patch: in flight during this one's arithmetic
    // ---- horizontal pass: one item = two raw rows x EIGHT columns from four dwords per row; the seven taps as v_dot4 with
    // the weights shifted instead of the bytes (20 per row: no realignment), 16-bit sums of vertically adjacent rows packed
    {
      constexpr unsigned A0 = 18u | (34u << 8) | (48u << 16) | (56u << 24), A1 = 48u | (34u << 8) | (18u << 16);
      constexpr unsigned B0 = (18u << 8) | (34u << 16) | (48u << 24), B1 = 56u | (48u << 8) | (34u << 16) | (18u << 24);
      constexpr unsigned C0 = (18u << 16) | (34u << 24), C1 = 48u | (56u << 8) | (48u << 16) | (34u << 24), C2 = 18u;
      constexpr unsigned E0 = 18u << 24, E1 = 34u | (48u << 8) | (56u << 16) | (48u << 24), E2 = 34u | (18u << 8);
      for (int idx = lane; idx < ((VIEO_FUSED_AB & 1) ? 0 : kFHPairs * (kFHCols / 8)); idx += 64) {
        const int pr = idx / (kFHCols / 8), g8 = idx - pr * (kFHCols / 8);
        unsigned h[2][8];
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const int r = min(2 * pr + q, kFRawRows - 1);  // (the 44th row does not exist: its sums are never read)
          const uint2* w = (const uint2*)(s_raw + r * kFRawPitch + 8 * g8);
          const uint2 lo = w[0], hi = w[1];
          const unsigned d0 = lo.x, d1 = lo.y, d2 = hi.x, d3 = hi.y;
#define DOT4(x, k, acc) __builtin_amdgcn_udot4(x, k, acc, false)
          h[q][0] = DOT4(d1, A1, DOT4(d0, A0, 0u));
          h[q][1] = DOT4(d1, B1, DOT4(d0, B0, 0u));
          h[q][2] = DOT4(d2, C2, DOT4(d1, C1, DOT4(d0, C0, 0u)));
          h[q][3] = DOT4(d2, E2, DOT4(d1, E1, DOT4(d0, E0, 0u)));
          h[q][4] = DOT4(d2, A1, DOT4(d1, A0, 0u));
          h[q][5] = DOT4(d2, B1, DOT4(d1, B0, 0u));
          h[q][6] = DOT4(d3, C2, DOT4(d2, C1, DOT4(d1, C0, 0u)));
          h[q][7] = DOT4(d3, E2, DOT4(d2, E1, DOT4(d1, E0, 0u)));
#undef DOT4
        }
        uint4 o0, o1;
        o0.x = h[0][0] | (h[1][0] << 16), o0.y = h[0][1] | (h[1][1] << 16);
        o0.z = h[0][2] | (h[1][2] << 16), o0.w = h[0][3] | (h[1][3] << 16);
        o1.x = h[0][4] | (h[1][4] << 16), o1.y = h[0][5] | (h[1][5] << 16);
        o1.z = h[0][6] | (h[1][6] << 16), o1.w = h[0][7] | (h[1][7] << 16);
        uint4* dst = (uint4*)(s_h + pr * kFHCols + 8 * g8);
        dst[0] = o0, dst[1] = o1;
      }
    }
    // ---- IC_Angle on the raw patch (its centre is row kFRaw, byte kFRaw): two lanes per row of the radius-15 disc
    int m10 = 0, m01 = 0;
    if (lane < ((VIEO_FUSED_AB & 4) ? 1 : 62)) {
      const int vv = (lane >> 1) - kHalfPatch;
      const int d = P.umax[vv < 0 ? -vv : vv];
      const uint8_t* row = s_raw + (vv + kFRaw) * kFRawPitch + kFRaw;
      const int u0 = (lane & 1) ? 0 : -d, u1 = (lane & 1) ? d : -1;
      int sI = 0;
#pragma unroll
      for (int t = 0; t < 16; t++) {
        const int uu = u0 + t;
        if (uu <= u1) {
          const int val = row[uu];
          m10 += uu * val;
          sI += val;
        }
      }
      m01 = vv * sI;
    }
    m10 = wave_sum(m10);
    m01 = wave_sum(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- vertical pass (k_blur's): one item = two blurred rows x four columns, into the space the raw patch had
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    uint8_t* s_bl = s_raw;
    for (int idx = lane; idx < ((VIEO_FUSED_AB & 1) ? 0 : ((kFBlRows + 1) / 2) * (kFBlPitch / 4)); idx += 64) {
      const int q = idx / (kFBlPitch / 4), c4 = (idx - q * (kFBlPitch / 4)) * 4;
      const unsigned We[4] = {18u | (34u << 16), 48u | (56u << 16), 48u | (34u << 16), 18u};
      const unsigned Wo[4] = {18u << 16, 34u | (48u << 16), 56u | (48u << 16), 34u | (18u << 16)};
      unsigned e[4] = {1u << 15, 1u << 15, 1u << 15, 1u << 15}, o[4] = {1u << 15, 1u << 15, 1u << 15, 1u << 15};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint4 hv = *(const uint4*)(s_h + (q + k) * kFHCols + c4);
        const unsigned vv[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
          e[jj] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, vv[jj]), __builtin_bit_cast(us2, We[k]), e[jj], false);
          o[jj] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, vv[jj]), __builtin_bit_cast(us2, Wo[k]), o[jj], false);
        }
      }
      // byte 2 of each sum (the sums stay below 2^24): three v_perm per four pixels
      const unsigned oe = __builtin_amdgcn_perm(__builtin_amdgcn_perm(e[3], e[2], 0x0c0c0602u), __builtin_amdgcn_perm(e[1], e[0], 0x0c0c0602u), 0x05040100u);
      const unsigned oo = __builtin_amdgcn_perm(__builtin_amdgcn_perm(o[3], o[2], 0x0c0c0602u), __builtin_amdgcn_perm(o[1], o[0], 0x0c0c0602u), 0x05040100u);
      *(unsigned*)(s_bl + (2 * q) * kFBlPitch + c4) = oe;
      if (2 * q + 1 < kFBlRows) *(unsigned*)(s_bl + (2 * q + 1) * kFBlPitch + c4) = oo;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- steered BRIEF on the blurred patch (ORBextractor.cc:83-127)
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float a, bsin;
    vieo_sincosf_exact(angle * factorPI, &bsin, &a);
    const uint8_t* bl = s_bl + kFR * kFBlPitch + kFR;
    unsigned long long bits[4];
#pragma unroll
    for (int gq = 0; gq < ((VIEO_FUSED_AB & 4) ? 1 : 4); gq++) {
      const int pt = pt4[gq];
      const float x0 = (float)(signed char)(pt & 0xFF), y0 = (float)(signed char)((pt >> 8) & 0xFF);
      const float x1 = (float)(signed char)((pt >> 16) & 0xFF), y1 = (float)(signed char)((pt >> 24) & 0xFF);
      const int t0 = bl[__float2int_rn(x0 * bsin + y0 * a) * kFBlPitch + __float2int_rn(x0 * a - y0 * bsin)];
      const int t1 = bl[__float2int_rn(x1 * bsin + y1 * a) * kFBlPitch + __float2int_rn(x1 * a - y1 * bsin)];
      bits[gq] = __ballot(t0 < t1);
    }
    if (lane < 4) ((unsigned long long*)(desc_out + ((size_t)b * out_cap + g) * 32))[lane] = bits[lane];
    if (lane == 0) {
      vieo_keypoint k;
      const float fx = (float)cx, fy = (float)cy;
      k.x = level ? fx * D.scale : fx;
      k.y = level ? fy * D.scale : fy;
      k.size = (float)D.patch;
      k.angle = angle;
      k.response = (float)QT_KEY_R(key);
      k.octave = level;
      k.class_id = -1;
      kp_out[(size_t)b * out_cap + g] = k;
    }
    // (the next iteration overwrites the patch this one sampled)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------ lapping-area reorder
// ORBextractor.cc:1041-1052: keys with lap0 <= x <= lap1 fill the output from the back, the
// others from the front, both in level-concatenated order.  One workgroup per image.
__global__ void __launch_bounds__(256)
k_lapping(const vieo_keypoint* __restrict__ kin, const uint8_t* __restrict__ din, int in_cap,
          vieo_keypoint* __restrict__ kout, uint8_t* __restrict__ dout, int out_cap,
          const int* __restrict__ tmp_counts, int* __restrict__ counts, int lap0, int lap1) {
  __shared__ int s_part[256];
  __shared__ int s_total;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = tmp_counts[2 * b];
  const int per = (n + 255) / 256;
  const int i0 = tid * per, i1 = min(n, i0 + per);
  int mono = 0;
  for (int i = i0; i < i1; i++) {
    const float x = kin[(size_t)b * in_cap + i].x;
    mono += !(x >= (float)lap0 && x <= (float)lap1);
  }
  s_part[tid] = mono;
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int t = 0; t < 256; t++) {
      const int v = s_part[t];
      s_part[t] = acc;
      acc += v;
    }
    s_total = acc;
  }
  __syncthreads();
  int mi = s_part[tid];
  for (int i = i0; i < i1; i++) {
    const vieo_keypoint k = kin[(size_t)b * in_cap + i];
    const bool st = (k.x >= (float)lap0 && k.x <= (float)lap1);
    const int pos = st ? (n - 1 - (i - mi)) : mi;
    if (!st) mi++;
    if (pos < out_cap) {
      kout[(size_t)b * out_cap + pos] = k;
      const uint4* s = (const uint4*)(din + ((size_t)b * in_cap + i) * 32);
      uint4* d = (uint4*)(dout + ((size_t)b * out_cap + pos) * 32);
      d[0] = s[0];
      d[1] = s[1];
    }
  }
  if (tid == 0) {
    counts[2 * b] = min(n, out_cap);
    counts[2 * b + 1] = s_total;
  }
}

}  // namespace vieo

using namespace vieo;

// ================================================================== host side
namespace vieo {

static inline int cv_round_f(float v) { return (int)lrintf(v); }  // cvRound: half-to-even
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor_d(double v) {
  int i = (int)v;
  return i - (i > v);
}
static inline int cv_ceil_d(double v) {
  int i = (int)v;
  return i + (i < v);
}

// cv::resize coefficient tables for one axis (INTER_LINEAR, 8U: 11-bit fixed point).
static void resize_axis_table(int ssize, int dsize, bool horizontal, std::vector<short>& tab) {
  const double inv_scale = (double)dsize / ssize;
  const double sc = 1. / inv_scale;
  tab.resize((size_t)dsize * 4);
  for (int d = 0; d < dsize; d++) {
    float f = (float)((d + 0.5) * sc - 0.5);
    int s = cv_floor_d(f);
    f -= s;
    int s0, s1;
    if (horizontal) {
      if (s < 0) f = 0, s = 0;
      if (s >= ssize - 1) f = 0, s = ssize - 1;
      s0 = s;
      s1 = std::min(s + 1, ssize - 1);
    } else {
      s0 = std::min(std::max(s, 0), ssize - 1);
      s1 = std::min(std::max(s + 1, 0), ssize - 1);
    }
    short a0 = (short)cv_round_f((1.f - f) * 2048);
    short a1 = (short)cv_round_f(f * 2048);
    if (horizontal && s >= ssize - 1) a0 = 2048, a1 = 0;
    tab[d * 4 + 0] = (short)s0;
    tab[d * 4 + 1] = (short)s1;
    tab[d * 4 + 2] = a0;
    tab[d * 4 + 3] = a1;
  }
}

static int plan_geometry(vieo_orb* e, int w, int h, int B) {
  OrbParams& P = e->P;
  memset(&P, 0, sizeof(P));
  P.nlevels = e->nlevels;
  for (int i = 0; i <= kHalfPatch; i++) P.umax[i] = e->umax[i];
  e->cells.clear();
  e->tiles.clear();
  std::vector<short> xtab, ytab;
  int resize_rows = 1, resize_dw = 1;
  size_t pyr_off = 0, blur_off = 0;
  int max_cw = 0, max_ch = 0, key_off = 0, sel_off = 0, ncap_max = 0, max_ncell = 0;
  for (int l = 0; l < e->nlevels; l++) {
    LevelDesc& D = P.lv[l];
    const float s = e->inv_scale[l];
    D.w = cv_round_f((float)w * s);  // ORBextractor.cc:1063
    D.h = cv_round_f((float)h * s);
    if (D.w >= 4096 || D.h >= 4096) {
      set_error("image level %d (%dx%d) exceeds the 4095-pixel key packing", l, D.w, D.h);
      return VIEO_E_INVALID;
    }
    D.pitch = align_up(D.w, 16);
    if (l > 0) {
      D.off = (int)pyr_off;
      pyr_off += (size_t)D.pitch * D.h;
      pyr_off = align_up_sz(pyr_off, 256);
    }
    D.boff = (int)blur_off;
    blur_off += (size_t)D.pitch * D.h;
    blur_off = align_up_sz(blur_off, 256);
    D.nfeat = e->feats[l];
    D.scale = e->scale[l];
    D.patch = (int)(kPatchSize * e->scale[l]);  // ORBextractor.cc:787
    // ---- cells (ORBextractor.cc:729-760)
    const int minB = kEdge - 3;
    const int maxBX = D.w - kEdge + 3, maxBY = D.h - kEdge + 3;
    D.regW = maxBX - minB;
    D.regH = maxBY - minB;
    D.cell_begin = (int)e->cells.size();
    if (D.regW >= 35 && D.regH >= 35) {
      const float W = 35;
      const float width = (float)D.regW, height = (float)D.regH;
      const int nCols = (int)(width / W), nRows = (int)(height / W);
      const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
      for (int i = 0; i < nRows; i++) {
        const float iniY = minB + i * hCell;
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBY - 3) continue;
        if (maxY > maxBY) maxY = maxBY;
        for (int j = 0; j < nCols; j++) {
          const float iniX = minB + j * wCell;
          float maxX = iniX + wCell + 6;
          if (iniX >= maxBX - 6) continue;
          if (maxX > maxBX) maxX = maxBX;
          CellDesc c;
          c.level = (short)l;
          c.x0 = (short)iniX, c.y0 = (short)iniY;
          c.cw = (short)((int)maxX - (int)iniX), c.ch = (short)((int)maxY - (int)iniY);
          c.offx = (short)(j * wCell), c.offy = (short)(i * hCell);
          c.pad = 0;
          e->cells.push_back(c);
          max_cw = std::max(max_cw, (int)c.cw);
          max_ch = std::max(max_ch, (int)c.ch);
        }
      }
      D.nIni = (int)roundf((float)D.regW / (float)D.regH);  // ORBextractor.cc:524
      if (D.nIni < 1) {
        set_error("level %d region %dx%d too tall for DistributeOctTree (nIni=0)", l, D.regW, D.regH);
        return VIEO_E_INVALID;
      }
      D.hX = (float)D.regW / (float)D.nIni;
    } else {
      set_error("image too small: level %d is %dx%d", l, D.w, D.h);
      return VIEO_E_INVALID;
    }
    D.cell_end = (int)e->cells.size();
    max_ncell = std::max(max_ncell, D.cell_end - D.cell_begin);
    D.ncap = std::max(D.nfeat, 4 * D.nIni) + 8;
    ncap_max = std::max(ncap_max, D.ncap);
    D.sel_off = sel_off;
    sel_off += D.ncap;
    // ---- resize tables
    if (l > 0) {
      std::vector<short> t;
      D.xtab_off = (int)(xtab.size() / 4);
      resize_axis_table(P.lv[l - 1].w, D.w, true, t);
      xtab.insert(xtab.end(), t.begin(), t.end());
      D.ytab_off = (int)(ytab.size() / 4);
      resize_axis_table(P.lv[l - 1].h, D.h, false, t);
      ytab.insert(ytab.end(), t.begin(), t.end());
      // LDS footprint of one k_resize workgroup at this level
      const short* xt = xtab.data() + 4 * (size_t)D.xtab_off;
      const short* yt = ytab.data() + 4 * (size_t)D.ytab_off;
      for (int y0 = 0; y0 < D.h; y0 += kResizeRows)
        resize_rows = std::max(resize_rows, yt[4 * (std::min(y0 + kResizeRows, D.h) - 1) + 1] - yt[4 * y0] + 1);
      for (int x0 = 0; x0 < D.w; x0 += 256)
        resize_dw = std::max(resize_dw, ((xt[4 * (std::min(x0 + 256, D.w) - 1) + 1] + 4) >> 2) - (xt[4 * x0] >> 2));
    }
    // ---- blur tiles
    D.tile_begin = (int)e->tiles.size();
    D.tiles_x = (D.w + kBlurTW - 1) / kBlurTW;
    for (int ty = 0; ty < (D.h + kBlurTH - 1) / kBlurTH; ty++)
      for (int tx = 0; tx < D.tiles_x; tx++) e->tiles.push_back({(short)l, (short)tx, (short)ty, 0});
    D.tile_end = (int)e->tiles.size();
  }
  P.ncells = (int)e->cells.size();
  if (max_cw - 6 > 64 || max_ch - 6 > 1023) {  // k_fast: one lane per interior column, y << 6 | x keys
    set_error("FAST cell wider than 64 pixels (image narrower than one 30-pixel cell?)");
    return VIEO_E_INVALID;
  }
  // strict 8-neighbour local maxima cannot be adjacent: at most ceil(vw/2)*ceil(vh/2) per cell
  P.cell_cap = std::max(16, ((max_cw - 6 + 1) / 2) * ((max_ch - 6 + 1) / 2));
  for (int l = 0; l < e->nlevels; l++) {
    LevelDesc& D = P.lv[l];
    D.key_off = key_off;
    D.key_cap = (D.cell_end - D.cell_begin) * P.cell_cap;
    key_off += D.key_cap;
    if (D.key_cap >= (1 << 24)) {
      set_error("too many candidate keys per level");
      return VIEO_E_INVALID;
    }
  }
  P.keys_per_image = key_off;
  P.sel_per_image = sel_off;
  P.kp_cap = sel_off;
  e->ncap_max = ncap_max;
  e->scap_max = std::max(2 * ncap_max, max_ncell);
  e->pyr_img = align_up_sz(pyr_off, 256);
  e->blur_img = align_up_sz(blur_off, 256);
  e->resize_pitch = 4 * resize_dw + 4;  // bytes; +4 keeps consecutive rows on different banks
  e->resize_lds = resize_rows * e->resize_pitch;
  {  // k_resize2: the (l, l + 1) pairs l = 1, 3, 5, ...: every tile's level-l region and its source band must fit the kernel
    bool ok = e->nlevels >= 3;
    int rows2 = 1, dw2 = 1;
    for (int l = 1; ok && l + 1 < e->nlevels; l += 2) {
      const LevelDesc &D1 = P.lv[l], &D2 = P.lv[l + 1];
      const short* xt1 = xtab.data() + 4 * (size_t)D1.xtab_off;
      const short* yt1 = ytab.data() + 4 * (size_t)D1.ytab_off;
      const short* xt2 = xtab.data() + 4 * (size_t)D2.xtab_off;
      const short* yt2 = ytab.data() + 4 * (size_t)D2.ytab_off;
      for (int ex0 = 0; ok && ex0 < D2.w; ex0 += kR2W) {
        const int ex1 = std::min(ex0 + kR2W, D2.w);
        const int rx0 = ex0 == 0 ? 0 : (xt2[4 * ex0] & ~3), rx1 = ex1 == D2.w ? D1.w : std::min(xt2[4 * (ex1 - 1) + 1] + 1, D1.w);
        const int ox1 = ex1 == D2.w ? D1.w : (xt2[4 * ex1] & ~3);
        const int ndw = ((xt1[4 * (rx1 - 1) + 1] + 4) >> 2) - ((xt1[4 * rx0] & ~3) >> 2);
        ok = rx1 - rx0 <= 256 && ox1 <= rx1 && ox1 > rx0 && ndw <= 128;
        dw2 = std::max(dw2, ndw);
      }
      for (int ey0 = 0; ok && ey0 < D2.h; ey0 += kR2H) {
        const int ey1 = std::min(ey0 + kR2H, D2.h);
        const int ry0 = ey0 == 0 ? 0 : yt2[4 * ey0], ry1 = ey1 == D2.h ? D1.h : std::min(yt2[4 * (ey1 - 1) + 1] + 1, D1.h);
        const int oy1 = ey1 == D2.h ? D1.h : yt2[4 * ey1];
        const int nrows = yt1[4 * (ry1 - 1) + 1] - yt1[4 * ry0] + 1;
        ok = ry1 - ry0 <= kR2Rows && oy1 <= ry1 && oy1 > ry0 && nrows <= 48;
        rows2 = std::max(rows2, nrows);
      }
    }
    e->resize2_ok = ok;
    e->resize2_pitch = 4 * dw2 + 4;
    e->resize2_l1_off = (rows2 * e->resize2_pitch + 15) & ~15;
    e->resize2_lds = e->resize2_l1_off + kR2Rows * 260;
  }
  // FAST LDS: cell tile (dword-aligned columns) + strength tile
  e->tpitch = align_up(max_cw + 3, 4) + 4;
  e->tile_bytes = align_up(e->tpitch * max_ch, 16);
  e->score_bytes = align_up((max_cw - 4) * (max_ch - 4), 16) + 16;
  // candidate list: at most 512 entries in LDS (fast_cell scores the pending ones when it fills up and falls back to a
  // dense evaluation of the cell when even the corners alone do not fit); VIEO_FAST_CAND_CAP lets the tests force both
  // (at least 256: pass A appends up to 256 entries per step and would otherwise drop to the dense fallback at once)
  e->fast_cand_cap = std::max(256, std::min((max_cw - 6) * (max_ch - 6), 512));
  if (const char* cc = getenv("VIEO_FAST_CAND_CAP")) e->fast_cand_cap = std::max(256, atoi(cc));
  e->fast_lds = e->tile_bytes + e->score_bytes + align_up(2 * e->fast_cand_cap, 16) + 16;
  e->qt_lds = 16 * e->scap_max + (4 + 16 + 4 + 4 + 4) * ncap_max + 64 + (2 * 4 + 8 + 2 * 6) * ncap_max +
              ncap_max + 64;
  // ---- device buffers
  int rc;
#define ENS(buf, bytes)                 \
  if ((rc = (buf).ensure(bytes)) != VIEO_OK) return rc
  ENS(e->d_pyr, e->pyr_img * B);
  if (!orb_fused()) ENS(e->d_blur, e->blur_img * B);  // (the fused descriptor kernel has no blurred pyramid)
  ENS(e->d_cells, e->cells.size() * sizeof(CellDesc));
  ENS(e->d_tiles, e->tiles.size() * sizeof(BlurTile));
  ENS(e->d_xtab, std::max<size_t>(xtab.size() * 2, 8));
  ENS(e->d_ytab, std::max<size_t>(ytab.size() * 2, 8));
  ENS(e->d_cell_keys, (size_t)B * P.ncells * P.cell_cap * 4);
  ENS(e->d_cell_counts, (size_t)B * P.ncells * 4);
  ENS(e->d_keys, (size_t)B * P.keys_per_image * 4);
  ENS(e->d_kslot, (size_t)B * P.keys_per_image * 2);
  ENS(e->d_kq, (size_t)B * P.keys_per_image);
  ENS(e->d_sel, (size_t)B * P.sel_per_image * 4);
  ENS(e->d_sel_count, (size_t)B * kMaxLevels * 4);
  ENS(e->d_tmp_kp, (size_t)B * P.kp_cap * sizeof(vieo_keypoint));
  ENS(e->d_tmp_desc, (size_t)B * P.kp_cap * 32);
  ENS(e->d_tmp_counts, (size_t)B * 2 * 4);
  ENS(e->d_krec, (size_t)B * P.kp_cap * 8);
#undef ENS
  VIEO_HIP_CHECK(hipMemcpyAsync(e->d_cells.p, e->cells.data(), e->cells.size() * sizeof(CellDesc),
                                hipMemcpyHostToDevice, e->stream));
  VIEO_HIP_CHECK(hipMemcpyAsync(e->d_tiles.p, e->tiles.data(), e->tiles.size() * sizeof(BlurTile),
                                hipMemcpyHostToDevice, e->stream));
  if (!xtab.empty()) {
    VIEO_HIP_CHECK(hipMemcpyAsync(e->d_xtab.p, xtab.data(), xtab.size() * 2, hipMemcpyHostToDevice,
                                  e->stream));
    VIEO_HIP_CHECK(hipMemcpyAsync(e->d_ytab.p, ytab.data(), ytab.size() * 2, hipMemcpyHostToDevice,
                                  e->stream));
  }
  VIEO_HIP_CHECK(hipStreamSynchronize(e->stream));  // host vectors go out of scope
  if (e->qt_lds > 64 * 1024)
    VIEO_HIP_CHECK(hipFuncSetAttribute((const void*)k_quadtree,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, e->qt_lds));
  if (e->qt_lds > 160 * 1024) {
    set_error("nfeatures too large for the LDS quadtree (%d bytes)", e->qt_lds);
    return VIEO_E_INVALID;
  }
  e->w = w;
  e->h = h;
  e->B = B;
  return VIEO_OK;
}

static int run_batch(vieo_orb* e, const uint8_t* d_images, int B, int w, int h, int stride,
                     size_t image_pitch, const int* lapping, vieo_keypoint* d_kp, uint8_t* d_desc,
                     int capacity, int32_t* d_counts) {
  int rc;
  if (w != e->w || h != e->h || B > e->B) {
    if ((rc = plan_geometry(e, w, h, std::max(B, (w == e->w && h == e->h) ? e->B : 0))) != VIEO_OK)
      return rc;
  }
  if ((stride & 3) || ((uintptr_t)d_images & 3) || (image_pitch & 3)) {
    set_error("device images need 4-byte aligned base, stride and pitch");
    return VIEO_E_INVALID;
  }
  const OrbParams& P = e->P;
  ImgSet I;
  I.img0 = d_images;
  I.stride0 = stride;
  I.img_pitch = image_pitch;
  I.pyr = e->d_pyr.as<uint8_t>();
  I.pyr_img = e->pyr_img;
  I.blur = e->d_blur.as<uint8_t>();
  I.blur_img = e->blur_img;
  e->last_imgs = I;
  e->last_B = B;
  hipStream_t st = e->stream;
  Timing& T = e->tm;
  int evi = 0;
  hipEvent_t* evs = T.ev[T.steps % kTimingRing];
#define STAMP()                                                  \
  if (T.on) VIEO_HIP_CHECK(hipEventRecord(evs[evi++], st))
  STAMP();
  // two levels per launch where the geometry allows (k_resize2; VIEO_RESIZE_PAIRS=0: one level per launch, A/B runs)
  static const bool resize_pairs = [] {
    const char* v = getenv("VIEO_RESIZE_PAIRS");
    return !(v && atoi(v) == 0);
  }();
  for (int l = 1; l < P.nlevels;) {
    if (resize_pairs && e->resize2_ok && (l & 1) && l + 1 < P.nlevels) {
      const LevelDesc& D2 = P.lv[l + 1];
      dim3 grd((D2.w + kR2W - 1) / kR2W, (D2.h + kR2H - 1) / kR2H, B);
      hipLaunchKernelGGL(k_resize2, grd, dim3(256), e->resize2_lds, st, P, l, I, e->d_xtab.as<short4>(), e->d_ytab.as<short4>(),
                         e->resize2_pitch, e->resize2_l1_off);
      l += 2;
      continue;
    }
    const LevelDesc& D = P.lv[l];
    dim3 grd((D.w + 255) / 256, (D.h + kResizeRows - 1) / kResizeRows, B);
    hipLaunchKernelGGL(k_resize, grd, dim3(256), e->resize_lds, st, P, l, I, e->d_xtab.as<short4>(),
                       e->d_ytab.as<short4>(), e->resize_pitch);
    l++;
  }
  STAMP();
  const auto xcd_grid = [](long long items) {
    return (unsigned)((items + 8 * kXcdRun - 1) / (8 * kXcdRun) * (8 * kXcdRun));
  };
  if ((long long)P.ncells * B > 0x7FFF0000LL || (long long)e->tiles.size() * B > 0x7FFF0000LL) {
    set_error("batch too large for one launch");
    return VIEO_E_CAPACITY;
  }
  hipLaunchKernelGGL(k_fast, dim3(xcd_grid(((long long)P.ncells * B + VIEO_FAST_WAVES - 1) / VIEO_FAST_WAVES)), dim3(64 * VIEO_FAST_WAVES),
                     (size_t)align_up(e->fast_lds, 16) * VIEO_FAST_WAVES, st, P, I,
                     e->d_cells.as<CellDesc>(), e->d_cell_keys.as<unsigned>(), e->d_cell_counts.as<int>(),
                     e->iniTh, e->minTh, e->tpitch, e->tile_bytes, e->score_bytes, e->fast_cand_cap, B,
                     (int)align_up(e->fast_lds, 16));
  STAMP();
  // Several launches over groups of levels: the LDS a workgroup carves is sized by its level's node / cell capacities,
  // and sized for level 0 (25 KB) the smaller levels ran six workgroups per CU instead of eight (one launch 0.37 ms per
  // 1024 images, split after level 0: 0.28)
  {
    static const int bounds_batch[] = VIEO_QT_GROUPS;  // level group boundaries, ascending, the last one >= nlevels
    static const int bounds_one[] = {64};
    // (a few images -- the single-stream frame -- are latency: one launch, 39 us, instead of two of 39 us each)
    const bool batch = B >= 32;
    const int* bounds = batch ? bounds_batch : bounds_one;
    const size_t n_bounds = batch ? sizeof(bounds_batch) / sizeof(bounds_batch[0]) : 1;
    auto qt_caps = [&](int la, int lb, int* ncap, int* scap) {
      int nc = 1, mc = 1;
      for (int l = la; l < lb; l++) nc = std::max(nc, P.lv[l].ncap), mc = std::max(mc, P.lv[l].cell_end - P.lv[l].cell_begin);
      *ncap = nc, *scap = std::max(2 * nc, mc);
    };
    int la = 0;
    // (a few images: latency, a workgroup's passes over its keys on more threads; a batch: four workgroups per CU)
    static const int qt_env = [] {  // (k_quadtree's launch bound is 1024; whole wavefronts)
      const char* e = getenv("VIEO_QT_THREADS");
      const int v = e ? atoi(e) : 0;
      return v > 0 ? std::min(std::max(v / 64 * 64, 64), 1024) : 0;
    }();
    const int qt_threads = qt_env > 0 ? qt_env : (B <= 16 ? 512 : 256);
    for (size_t gi = 0; gi < n_bounds && la < P.nlevels; gi++) {
      const int lb = std::min(bounds[gi], P.nlevels);
      if (lb <= la) continue;
      int nc, sc;
      qt_caps(la, lb, &nc, &sc);
      const size_t lds = (size_t)16 * sc + (4 + 16 + 4 + 4 + 4) * nc + 64 + (2 * 4 + 8 + 2 * 6) * nc + nc + 64;
      hipLaunchKernelGGL(k_quadtree, dim3(lb - la, B), dim3(qt_threads), lds, st, P,
                         e->d_cell_keys.as<unsigned>(), e->d_cell_counts.as<int>(),
                         e->d_keys.as<unsigned>(), e->d_kslot.as<unsigned short>(),
                         e->d_kq.as<unsigned char>(), e->d_sel.as<unsigned>(), e->d_sel_count.as<int>(), nc, sc, la);
      la = lb;
    }
  }
  STAMP();
  // VIEO_ORB_FUSED=0: the two-kernel form (the whole pyramid blurred into HBM, k_describe gathers from it) -- kept for A/B
  // timing and as the producer of the blurred planes the parity tap reads
  const bool fused = orb_fused();
  if (!fused) {
    int rc_b = e->d_blur.ensure(e->blur_img * (size_t)e->B);
    if (rc_b != VIEO_OK) return rc_b;
    I.blur = e->d_blur.as<uint8_t>(), e->last_imgs = I;
    hipLaunchKernelGGL(k_blur, dim3(xcd_grid((long long)e->tiles.size() * B)), dim3(256), 0, st, P, I,
                       e->d_tiles.as<BlurTile>(), (int)e->tiles.size(), B);
  }
  STAMP();
  const int per_group = fused ? 4 * fused_kpw(B) : 4 * VIEO_DESC_KPW;
  const int ngroups = (std::min(P.kp_cap, capacity) + per_group - 1) / per_group;
  auto describe = [&](unsigned grid, vieo_keypoint* kp, uint8_t* desc, int cap, int ng) {
    if (fused)
      hipLaunchKernelGGL(k_describe_fused, dim3(grid), dim3(256), 0, st, P, I, e->d_krec.as<uint2>(), e->d_pattern.as<int>(), kp, desc,
                         cap, ng, B, fused_kpw(B));
    else
      hipLaunchKernelGGL(k_describe, dim3(grid), dim3(256), 0, st, P, I, e->d_krec.as<uint2>(), e->d_pattern.as<int>(), kp, desc, cap,
                         ng, B);
  };
  if (!lapping) {
    hipLaunchKernelGGL(k_desc_index, dim3((P.kp_cap + 255) / 256, B), dim3(256), 0, st, P, e->d_sel.as<unsigned>(),
                       e->d_sel_count.as<int>(), e->d_krec.as<uint2>(), capacity, d_counts);
    describe((unsigned)ngroups * 8 * ((B + 7) / 8), d_kp, d_desc, capacity, ngroups);
  } else {
    const int ng = (P.kp_cap + per_group - 1) / per_group;
    hipLaunchKernelGGL(k_desc_index, dim3((P.kp_cap + 255) / 256, B), dim3(256), 0, st, P, e->d_sel.as<unsigned>(),
                       e->d_sel_count.as<int>(), e->d_krec.as<uint2>(), P.kp_cap, e->d_tmp_counts.as<int>());
    describe((unsigned)ng * 8 * ((B + 7) / 8), e->d_tmp_kp.as<vieo_keypoint>(), e->d_tmp_desc.as<uint8_t>(), P.kp_cap, ng);
    hipLaunchKernelGGL(k_lapping, dim3(B), dim3(256), 0, st, e->d_tmp_kp.as<vieo_keypoint>(),
                       e->d_tmp_desc.as<uint8_t>(), P.kp_cap, d_kp, d_desc, capacity,
                       e->d_tmp_counts.as<int>(), d_counts, lapping[0], lapping[1]);
  }
  STAMP();
#undef STAMP
  if (T.on) T.steps++;
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

}  // namespace vieo

extern "C" {

int vieo_orb_create(vieo_orb** out, int nfeatures, float scale_factor, int nlevels, int ini_th,
                    int min_th) {
  // VIEO_ORB_PRIORITY=1: the extractor's (= the frame pipeline's) stream at the highest priority (A/B runs)
  const char* pe = getenv("VIEO_ORB_PRIORITY");
  return vieo::orb_create_with_priority(out, nfeatures, scale_factor, nlevels, ini_th, min_th, pe && atoi(pe) > 0 ? 1 : 0);
}

}  // extern "C"

// priority 1: the handle's stream comes from the runtime's HIGH-priority queues.  The runtime keeps one pool of hardware
// queues per priority level, so streams of different priority never share a queue: the one-call tracker puts its main
// stream there, its second stream at the normal level and the bundle adjustment runs at the lowest -- three chains that
// cannot end up in series behind each other whatever other streams the process holds.
int vieo::orb_create_with_priority(vieo_orb** out, int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th,
                                   int priority) {
  if (!out || nfeatures <= 0 || nlevels <= 0 || nlevels > kMaxLevels || !(scale_factor > 1.0f)) {
    set_error("vieo_orb_create: invalid arguments");
    return VIEO_E_INVALID;
  }
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  vieo_orb* e = new vieo_orb();
  e->nfeatures = nfeatures;
  e->nlevels = nlevels;
  e->iniTh = std::min(std::max(ini_th, 0), 255);
  e->minTh = std::min(std::max(min_th, 0), 255);
  e->scaleFactor = scale_factor;
  // ORBextractor.cc:397-431
  e->scale.resize(nlevels);
  e->sigma2.resize(nlevels);
  e->inv_scale.resize(nlevels);
  e->inv_sigma2.resize(nlevels);
  e->scale[0] = 1.0f;
  e->sigma2[0] = 1.0f;
  for (int i = 1; i < nlevels; i++) {
    e->scale[i] = (float)(e->scale[i - 1] * e->scaleFactor);
    e->sigma2[i] = e->scale[i] * e->scale[i];
  }
  for (int i = 0; i < nlevels; i++) {
    e->inv_scale[i] = 1.0f / e->scale[i];
    e->inv_sigma2[i] = 1.0f / e->sigma2[i];
  }
  e->feats.resize(nlevels);
  const float factor = (float)(1.0f / e->scaleFactor);
  float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
  int sum = 0;
  for (int l = 0; l < nlevels - 1; l++) {
    e->feats[l] = cv_round_f(nDesired);
    sum += e->feats[l];
    nDesired *= factor;
  }
  e->feats[nlevels - 1] = std::max(nfeatures - sum, 0);
  // ORBextractor.cc:439-455
  {
    int v, v0;
    const int vmax = cv_floor_d(kHalfPatch * sqrt(2.f) / 2 + 1);
    const int vmin = cv_ceil_d(kHalfPatch * sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (v = 0; v <= kHalfPatch; v++) e->umax[v] = 0;
    for (v = 0; v <= vmax; ++v) e->umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
      while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
      e->umax[v] = v0;
      ++v0;
    }
  }
  hipError_t he;
  {
    int lo = 0, hi = 0;
    if (priority > 0 && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
      he = hipStreamCreateWithPriority(&e->stream, hipStreamNonBlocking, hi);
    else
      he = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
  }
  if (he != hipSuccess) {
    set_error("hipStreamCreate: %s", hipGetErrorString(he));
    delete e;
    return VIEO_E_HIP;
  }
  // BRIEF pattern packed as one int32 per test
  std::vector<int> pat(256);
  for (int i = 0; i < 256; i++) {
    const signed char* p = VIEO_ORB_PATTERN_31 + i * 4;
    pat[i] = (int)((unsigned)(uint8_t)p[0] | ((unsigned)(uint8_t)p[1] << 8) |
                   ((unsigned)(uint8_t)p[2] << 16) | ((unsigned)(uint8_t)p[3] << 24));
  }
  if ((rc = e->d_pattern.ensure(1024)) != VIEO_OK) {
    delete e;
    return rc;
  }
  he = hipMemcpy(e->d_pattern.p, pat.data(), 1024, hipMemcpyHostToDevice);
  if (he != hipSuccess) {
    set_error("pattern upload: %s", hipGetErrorString(he));
    delete e;
    return VIEO_E_HIP;
  }
  for (auto& set : e->tm.ev)
    for (auto& ev : set) (void)hipEventCreate(&ev);
  *out = e;
  return VIEO_OK;
}

extern "C" {

void vieo_orb_destroy(vieo_orb* e) {
  if (!e) return;
  (void)hipStreamSynchronize(e->stream);
  DevBuf* bufs[] = {&e->d_pyr,   &e->d_blur,      &e->d_cells,    &e->d_tiles,      &e->d_xtab,
                    &e->d_ytab,  &e->d_cell_keys, &e->d_cell_counts, &e->d_keys,    &e->d_kslot,
                    &e->d_kq,    &e->d_sel,       &e->d_sel_count, &e->d_pattern,   &e->d_in,
                    &e->d_kp,    &e->d_desc,      &e->d_counts,   &e->d_tmp_kp,     &e->d_tmp_desc,
                    &e->d_tmp_counts, &e->d_krec,  &e->d_io,       &e->d_uright,     &e->d_depth,
                    &e->d_sad,   &e->g_start,     &e->g_rec,      &e->g_ang};
  for (DevBuf* b : bufs) b->release();
  e->h_in.release(), e->h_res.release(), e->h_io.release();
  for (auto& set : e->tm.ev)
    for (auto& ev : set)
      if (ev) (void)hipEventDestroy(ev);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

int vieo_orb_levels(const vieo_orb* e) { return e->nlevels; }
float vieo_orb_scale_factor(const vieo_orb* e) { return (float)e->scaleFactor; }
static int copy_tab(const std::vector<float>& v, float* out) {
  if (!out) return VIEO_E_INVALID;
  memcpy(out, v.data(), v.size() * sizeof(float));
  return VIEO_OK;
}
int vieo_orb_scale_factors(const vieo_orb* e, float* o) { return copy_tab(e->scale, o); }
int vieo_orb_inv_scale_factors(const vieo_orb* e, float* o) { return copy_tab(e->inv_scale, o); }
int vieo_orb_level_sigma2(const vieo_orb* e, float* o) { return copy_tab(e->sigma2, o); }
int vieo_orb_inv_level_sigma2(const vieo_orb* e, float* o) { return copy_tab(e->inv_sigma2, o); }
int vieo_orb_features_per_level(const vieo_orb* e, int* o) {
  if (!o) return VIEO_E_INVALID;
  memcpy(o, e->feats.data(), e->feats.size() * sizeof(int));
  return VIEO_OK;
}
int vieo_orb_max_keypoints(const vieo_orb* e) {
  // DistributeOctTree stops at >= N nodes and one split adds at most 3; root nodes up to 4*nIni
  int s = 0;
  for (int l = 0; l < e->nlevels; l++) s += std::max(e->feats[l], 16) + 8;
  return s;
}

int vieo_orb_extract_batch_device(vieo_orb* e, const uint8_t* d_images, int n_images, int width,
                                  int height, int stride, size_t image_pitch_bytes,
                                  const int* h_lapping, vieo_keypoint* d_keypoints,
                                  uint8_t* d_descriptors, int capacity, int32_t* d_counts) {
  if (!e || !d_images || n_images <= 0 || !d_keypoints || !d_descriptors || !d_counts ||
      capacity <= 0) {
    set_error("vieo_orb_extract_batch_device: invalid arguments");
    return VIEO_E_INVALID;
  }
  if (width <= 0 || height <= 0) return VIEO_E_EMPTY;
  return run_batch(e, d_images, n_images, width, height, stride, image_pitch_bytes, h_lapping,
                   d_keypoints, d_descriptors, capacity, d_counts);
}

#ifdef VIEO_FAST_STATS
// [0] cells, [1] cells that went on to minThFAST, [2] cells evaluated densely, [3] batches of 64 exact strengths,
// [4] batches of pass C, [6] compass survivors scored, [7] corners kept; reset on read
int vieo_fast_stats(unsigned long long* out8) {
  unsigned long long z[8] = {0};
  VIEO_HIP_CHECK(hipDeviceSynchronize());
  VIEO_HIP_CHECK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(vieo::g_fast_stats), sizeof(z)));
  VIEO_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(vieo::g_fast_stats), z, sizeof(z)));
  return VIEO_OK;
}
#endif

void* vieo_orb_stream(vieo_orb* e) { return e ? (void*)e->stream : nullptr; }

int vieo_orb_sync(vieo_orb* e) {
  VIEO_HIP_CHECK(hipStreamSynchronize(e->stream));
  return VIEO_OK;
}

int vieo_orb_extract(vieo_orb* e, const uint8_t* h_image, int width, int height, int stride,
                     const int* h_lapping, vieo_keypoint* h_keypoints, uint8_t* h_descriptors,
                     int capacity, int* n_keypoints, int* mono_index) {
  if (!e || !n_keypoints) return VIEO_E_INVALID;
  if (!h_image || width <= 0 || height <= 0) return VIEO_E_EMPTY;  // ORBextractor.cc:970
  if (stride < width) return VIEO_E_INVALID;
  int rc;
  const int pitch = align_up(width, 16);
  const int cap = vieo_orb_max_keypoints(e);
  e->res_n = -1;  // (whatever happens below, the handle no longer holds the previous frame)
  e->epoch++;
  if ((rc = e->d_in.ensure((size_t)pitch * height)) != VIEO_OK) return rc;
  if ((rc = e->d_kp.ensure((size_t)cap * sizeof(vieo_keypoint))) != VIEO_OK) return rc;
  if ((rc = e->d_desc.ensure((size_t)cap * 32)) != VIEO_OK) return rc;
  if ((rc = e->d_counts.ensure(8)) != VIEO_OK) return rc;
  // The image goes up through the handle's own pinned plane (a pageable source makes the runtime stage it in chunks, with
  // a synchronisation per chunk), and counts | keys | descriptors come back as ONE block behind ONE synchronisation
  // (round 4: the counts first, a synchronisation, then the two arrays and another).
  const size_t o_kp = 256, o_desc = o_kp + align_up_sz((size_t)cap * sizeof(vieo_keypoint), 256);
  if ((rc = e->h_in.ensure((size_t)pitch * height)) != VIEO_OK) return rc;
  if ((rc = e->h_res.ensure(o_desc + (size_t)cap * 32)) != VIEO_OK) return rc;
  {
    uint8_t* dst = (uint8_t*)e->h_in.p;
    if (stride == pitch)
      memcpy(dst, h_image, (size_t)pitch * (height - 1) + width);
    else
      for (int y = 0; y < height; y++) memcpy(dst + (size_t)y * pitch, h_image + (size_t)y * stride, width);
  }
  VIEO_HIP_CHECK(hipMemcpyAsync(e->d_in.p, e->h_in.p, (size_t)pitch * height, hipMemcpyHostToDevice, e->stream));
  rc = run_batch(e, e->d_in.as<uint8_t>(), 1, width, height, pitch, (size_t)pitch * height,
                 h_lapping, e->d_kp.as<vieo_keypoint>(), e->d_desc.as<uint8_t>(), cap,
                 e->d_counts.as<int32_t>());
  if (rc != VIEO_OK) return rc;
  uint8_t* R = (uint8_t*)e->h_res.p;
  VIEO_HIP_CHECK(hipMemcpyAsync(R, e->d_counts.p, 8, hipMemcpyDeviceToHost, e->stream));
  VIEO_HIP_CHECK(hipMemcpyAsync(R + o_kp, e->d_kp.p, (size_t)cap * sizeof(vieo_keypoint), hipMemcpyDeviceToHost, e->stream));
  VIEO_HIP_CHECK(hipMemcpyAsync(R + o_desc, e->d_desc.p, (size_t)cap * 32, hipMemcpyDeviceToHost, e->stream));
  VIEO_HIP_CHECK(hipStreamSynchronize(e->stream));
  const int* cnt = (const int*)R;
  const int n = cnt[0];
  *n_keypoints = n;
  if (mono_index) *mono_index = cnt[1];
  if (n > capacity || n > cap) {
    set_error("vieo_orb_extract: %d keypoints, capacity %d", n, std::min(capacity, cap));
    return VIEO_E_CAPACITY;
  }
  if (n > 0) {
    if (!h_keypoints || !h_descriptors) return VIEO_E_INVALID;
    memcpy(h_keypoints, R + o_kp, (size_t)n * sizeof(vieo_keypoint));
    memcpy(h_descriptors, R + o_desc, (size_t)n * 32);
  }
  // the frame stays resident: its identity = the count and the first / last eight keys
  const vieo_keypoint* K = (const vieo_keypoint*)(R + o_kp);
  memset(e->res_sample, 0, sizeof(e->res_sample));
  for (int i = 0; i < std::min(n, 8); i++) e->res_sample[i] = K[i], e->res_sample[8 + i] = K[n - 1 - i];
  e->res_n = n, e->res_mono = cnt[1];
  return VIEO_OK;
}

// Does the handle still hold THESE keys (the frame a Frame object describes)?  The count and the first / last eight
// keys are compared with what the last vieo_orb_extract returned: 1 = yes, the *_resident entries may be used for it.
int vieo_orb_holds(const vieo_orb* e, const vieo_keypoint* h_keys, int n_keys) {
  if (!e || e->res_n < 0 || n_keys != e->res_n || (n_keys > 0 && !h_keys)) return 0;
  for (int i = 0; i < std::min(n_keys, 8); i++)
    if (memcmp(&h_keys[i], &e->res_sample[i], sizeof(vieo_keypoint)) ||
        memcmp(&h_keys[n_keys - 1 - i], &e->res_sample[8 + i], sizeof(vieo_keypoint)))
      return 0;
  return 1;
}

int vieo_orb_resident_keys(const vieo_orb* e) { return e ? e->res_n : -1; }

int vieo_orb_level_size(const vieo_orb* e, int level, int* width, int* height) {
  if (!e || level < 0 || level >= e->nlevels || e->w == 0) return VIEO_E_INVALID;
  if (width) *width = e->P.lv[level].w;
  if (height) *height = e->P.lv[level].h;
  return VIEO_OK;
}

int vieo_orb_level_device(vieo_orb* e, int image_index, int level, const uint8_t** d_ptr,
                          int* pitch) {
  if (!e || level < 0 || level >= e->nlevels || image_index < 0 || image_index >= e->last_B)
    return VIEO_E_INVALID;
  const ImgSet& I = e->last_imgs;
  if (level == 0) {
    *d_ptr = I.img0 + (size_t)image_index * I.img_pitch;
    *pitch = I.stride0;
  } else {
    *d_ptr = I.pyr + (size_t)image_index * I.pyr_img + e->P.lv[level].off;
    *pitch = e->P.lv[level].pitch;
  }
  return VIEO_OK;
}

static int fetch_plane(vieo_orb* e, const uint8_t* d_ptr, int pitch, int w, int h, int border,
                       uint8_t* h_dst, int dst_stride) {
  VIEO_HIP_CHECK(hipStreamSynchronize(e->stream));
  if (!border) {
    VIEO_HIP_CHECK(hipMemcpy2D(h_dst, dst_stride, d_ptr, pitch, w, h, hipMemcpyDeviceToHost));
    return VIEO_OK;
  }
  // copyMakeBorder(BORDER_REFLECT_101) of the ROI (ORBextractor.cc:1072-1077): the border is a
  // pure function of the ROI, so it is synthesised while copying out.
  std::vector<uint8_t> tmp((size_t)w * h);
  VIEO_HIP_CHECK(hipMemcpy2D(tmp.data(), w, d_ptr, pitch, w, h, hipMemcpyDeviceToHost));
  auto refl = [](int p, int len) {
    if (p < 0) p = -p;
    if (p >= len) p = 2 * (len - 1) - p;
    return p;
  };
  for (int y = 0; y < h + 2 * kEdge; y++) {
    const uint8_t* s = tmp.data() + (size_t)refl(y - kEdge, h) * w;
    uint8_t* d = h_dst + (size_t)y * dst_stride;
    for (int x = 0; x < w + 2 * kEdge; x++) d[x] = s[refl(x - kEdge, w)];
  }
  return VIEO_OK;
}

int vieo_orb_get_level(vieo_orb* e, int image_index, int level, int with_border, uint8_t* h_dst,
                       int dst_stride) {
  const uint8_t* p;
  int pitch;
  int rc = vieo_orb_level_device(e, image_index, level, &p, &pitch);
  if (rc != VIEO_OK || !h_dst) return VIEO_E_INVALID;
  return fetch_plane(e, p, pitch, e->P.lv[level].w, e->P.lv[level].h, with_border, h_dst,
                     dst_stride);
}

int vieo_orb_enable_timing(vieo_orb* e, int on) {
  e->tm.on = on != 0;
  e->tm.steps = 0;
  return VIEO_OK;
}

int vieo_orb_timed_steps(vieo_orb* e) { return (int)std::min<long>(e->tm.steps, kTimingRing); }

// steps_back = 0 is the most recent stamped batch call
int vieo_orb_stage_ms(vieo_orb* e, int steps_back, float* h_ms) {
  if (steps_back < 0 || steps_back >= vieo_orb_timed_steps(e)) {
    set_error("timing not enabled or step %d not recorded", steps_back);
    return VIEO_E_INVALID;
  }
  hipEvent_t* ev = e->tm.ev[(e->tm.steps - 1 - steps_back) % kTimingRing];
  VIEO_HIP_CHECK(hipEventSynchronize(ev[VIEO_ORB_NSTAGES - 1]));
  for (int i = 0; i < VIEO_ORB_NSTAGES - 1; i++)
    VIEO_HIP_CHECK(hipEventElapsedTime(&h_ms[i], ev[i], ev[i + 1]));
  VIEO_HIP_CHECK(hipEventElapsedTime(&h_ms[VIEO_ORB_NSTAGES - 1], ev[0], ev[VIEO_ORB_NSTAGES - 1]));
  return VIEO_OK;
}

int vieo_orb_last_stage_ms(vieo_orb* e, float* h_ms) { return vieo_orb_stage_ms(e, 0, h_ms); }

// ---------------------------------------------------------------- test taps
int vieo_orb_tap_plane(vieo_orb* e, int image_index, int level, int which, uint8_t* h_dst,
                       int dst_stride) {
  if (!e || which != 1 || level < 0 || level >= e->nlevels || image_index < 0 ||
      image_index >= e->last_B)
    return VIEO_E_INVALID;
  const LevelDesc& D = e->P.lv[level];
  if (vieo::orb_fused()) {
    // the extraction did not leave a blurred pyramid behind: blur the last batch's levels now (k_blur, the two-kernel
    // form's producer; the fused kernel runs the same two passes per key-point patch)
    int rc = e->d_blur.ensure(e->blur_img * (size_t)e->B);
    if (rc != VIEO_OK) return rc;
    e->last_imgs.blur = e->d_blur.as<uint8_t>(), e->last_imgs.blur_img = e->blur_img;
    hipLaunchKernelGGL(vieo::k_blur, dim3((unsigned)(((long long)e->tiles.size() * e->last_B + 8 * vieo::kXcdRun - 1) / (8 * vieo::kXcdRun) * (8 * vieo::kXcdRun))), dim3(256), 0, e->stream, e->P,
                       e->last_imgs, e->d_tiles.as<vieo::BlurTile>(), (int)e->tiles.size(), e->last_B);
    VIEO_HIP_CHECK(hipGetLastError());
  }
  const uint8_t* p = e->last_imgs.blur + (size_t)image_index * e->blur_img + D.boff;
  return fetch_plane(e, p, D.pitch, D.w, D.h, 0, h_dst, dst_stride);
}

int vieo_orb_tap_candidates(vieo_orb* e, int image_index, int level, int32_t* h_dst, int cap) {
  if (!e || level < 0 || level >= e->nlevels || image_index < 0 || image_index >= e->last_B)
    return VIEO_E_INVALID;
  VIEO_HIP_CHECK(hipStreamSynchronize(e->stream));
  const OrbParams& P = e->P;
  const LevelDesc& D = P.lv[level];
  const int ncell = D.cell_end - D.cell_begin;
  std::vector<int> cnt(ncell);
  std::vector<unsigned> keys((size_t)ncell * P.cell_cap);
  VIEO_HIP_CHECK(hipMemcpy(cnt.data(),
                           e->d_cell_counts.as<int>() + (size_t)image_index * P.ncells + D.cell_begin,
                           ncell * 4, hipMemcpyDeviceToHost));
  VIEO_HIP_CHECK(hipMemcpy(
      keys.data(),
      e->d_cell_keys.as<unsigned>() + ((size_t)image_index * P.ncells + D.cell_begin) * P.cell_cap,
      keys.size() * 4, hipMemcpyDeviceToHost));
  int n = 0;
  for (int c = 0; c < ncell; c++)
    for (int i = 0; i < cnt[c]; i++) {
      if (n < cap) {
        const unsigned k = keys[(size_t)c * P.cell_cap + i];
        h_dst[n * 3] = QT_KEY_X(k);
        h_dst[n * 3 + 1] = QT_KEY_Y(k);
        h_dst[n * 3 + 2] = QT_KEY_R(k);
      }
      n++;
    }
  return n;
}

int vieo_orb_tap_level_keys(vieo_orb* e, int image_index, int level, vieo_keypoint* h_dst, int cap) {
  if (!e || level < 0 || level >= e->nlevels || image_index < 0 || image_index >= e->last_B)
    return VIEO_E_INVALID;
  VIEO_HIP_CHECK(hipStreamSynchronize(e->stream));
  const OrbParams& P = e->P;
  const LevelDesc& D = P.lv[level];
  int n = 0;
  VIEO_HIP_CHECK(hipMemcpy(&n, e->d_sel_count.as<int>() + image_index * kMaxLevels + level, 4,
                           hipMemcpyDeviceToHost));
  if (n < 0) {
    set_error("quadtree capacity overflow at level %d", level);
    return VIEO_E_CAPACITY;
  }
  std::vector<unsigned> keys(std::max(n, 1));
  VIEO_HIP_CHECK(hipMemcpy(keys.data(),
                           e->d_sel.as<unsigned>() + (size_t)image_index * P.sel_per_image + D.sel_off,
                           (size_t)n * 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < n && i < cap; i++) {
    vieo_keypoint k;
    k.x = (float)(QT_KEY_X(keys[i]) + kEdge - 3);
    k.y = (float)(QT_KEY_Y(keys[i]) + kEdge - 3);
    k.size = (float)D.patch;
    k.angle = -1.f;
    k.response = (float)QT_KEY_R(keys[i]);
    k.octave = level;
    k.class_id = -1;
    h_dst[i] = k;
  }
  return n;
}

}  // extern "C"
