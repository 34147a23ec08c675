// proj_search.hip -- tracking-side ORBmatcher::SearchByProjection on gfx950
// (reference: src/ORBmatcher.cc:230-335 and :1303-1467, grid query src/FrameBase.cpp:95-174).
//
// The reference is sequential over map points because a later query must skip keypoints that an
// earlier query already claimed (AddMapPoint).  Split accordingly:
//   k_sbp_project     (a12 only) last frame's map points -> window queries           [parallel]
//   k_sbp_candidates  one wavefront per query: window + octave test over the frame's keys,
//                     Hamming distances, candidates put in GetFeaturesInArea order     [parallel]
//   k_sbp_assign      one wavefront per frame replays the queries in order against the
//                     "claimed" flags kept in LDS: best / second-best, accept rules,
//                     rotation histogram + ComputeThreeMaxima                          [sequential]
// Integer/index work: results are bit-exact against oracle/proj_search.cc.
#include <climits>

#include "common.h"

namespace vieo {

static const int kGridRows = 48, kGridCols = 64;  // FrameBase.h:224-225
static const int kThHigh = 100, kHistoLen = 30;   // ORBmatcher.cc:20-22
static const int kCandCap = 128;                  // candidates kept per query (2 per lane)
static const int kMaxKeys = 4096;                 // 12-bit key index packing

// ORBmatcher.cc:1313-1378
__global__ void __launch_bounds__(256)
k_sbp_project(const vieo_last_frame_point* __restrict__ pts, const int* __restrict__ n_pts,
              int p_cap, const vieo_sbp_camera* __restrict__ cams, vieo_proj_query* __restrict__ out) {
  const int f = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p_cap) return;
  vieo_proj_query q;
  memset(&q, 0, sizeof(q));
  vieo_proj_query* dst = out + (size_t)f * p_cap + i;
  if (i >= n_pts[f]) {
    *dst = q;
    return;
  }
  const vieo_sbp_camera& C = cams[f];
  const vieo_last_frame_point p = pts[(size_t)f * p_cap + i];
  bool ok = p.flags & 1;
  const double* Tc = C.Tcw_cur;
  const double* Tl = C.Tcw_last;
  // tlrcr(2): z of the translation of Tlrw * Tcrw^-1
  const double r6 = Tl[8] * Tc[0] + Tl[9] * Tc[1] + Tl[10] * Tc[2];
  const double r7 = Tl[8] * Tc[4] + Tl[9] * Tc[5] + Tl[10] * Tc[6];
  const double r8 = Tl[8] * Tc[8] + Tl[9] * Tc[9] + Tl[10] * Tc[10];
  const double tz = Tl[11] - (r6 * Tc[3] + r7 * Tc[7] + r8 * Tc[11]);
  const bool bForward = tz > C.baseline && !C.mono;
  const bool bBackward = -tz > C.baseline && !C.mono;
  const double X = p.Xw[0], Y = p.Xw[1], Z = p.Xw[2];
  const double x3 = Tc[0] * X + Tc[1] * Y + Tc[2] * Z + Tc[3];
  const double y3 = Tc[4] * X + Tc[5] * Y + Tc[6] * Z + Tc[7];
  const double z3 = Tc[8] * X + Tc[9] * Y + Tc[10] * Z + Tc[11];
  if (C.th_far > 0 && z3 > C.th_far) ok = false;
  const float xc = (float)x3, yc = (float)y3;
  const float invzc = (float)(1.0 / z3);
  if (invzc < 0) ok = false;
  const float pnx = xc * invzc, pny = yc * invzc;
  const float u = C.fx * pnx + 0.f * pny + C.cx * 1.f;
  const float v = 0.f * pnx + C.fy * pny + C.cy * 1.f;
  if (!(u >= C.bounds[0] && u < C.bounds[1] && v >= C.bounds[2] && v < C.bounds[3])) ok = false;
  if (ok) {
    const int oct = p.octave;
    q.u = u, q.v = v;
    q.ur = u - C.bf * invzc;
    q.radius = C.th * C.scale[oct];
    if (bForward)
      q.level_min = 0, q.level_max = oct;
    else if (bBackward)
      q.level_min = oct, q.level_max = -1;
    else
      q.level_min = oct - 1, q.level_max = oct + 1;
    q.angle = p.angle;
    q.flags = 1 | (p.flags & 2);
    for (int k = 0; k < 32; k++) q.desc[k] = p.desc[k];
  }
  *dst = q;
}

struct SbpArgs {
  int mode;
  const vieo_proj_query* queries;  // [frame][q_cap]
  const int* nq;                   // [frame]
  int q_cap;
  const vieo_keypoint* keys;       // [image][key_cap]
  const float* uright;             // [frame][key_cap]
  const uint8_t* desc;             // [image][key_cap][32]
  const uint8_t* taken;            // [frame][key_cap] or null
  const int* counts;               // [image][2]
  int key_cap, img_first, img_step;
  float minx, maxx, miny, maxy;
  float nn_ratio;
  int check_ori;
  const int* cell_start;            // [frame][kGridCols * kGridRows + 1]  Frame::mGrid as CSR
  const unsigned short* cell_list;  // [frame][key_cap] key indices, ascending inside a cell
  unsigned* cand;      // [frame][q_cap][kCandCap]  idx | dist<<12 | level<<21
  int* cand_n;         // [frame][q_cap]   (-1: overflow)
  int* assign;         // [frame][key_cap]
  int* nmatches;       // [frame]
};

__device__ __forceinline__ int hamming32q(const uint4 a0, const uint4 a1, const uint8_t* b) {
  const uint4 b0 = ((const uint4*)b)[0], b1 = ((const uint4*)b)[1];
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
         __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// Frame::AssignFeaturesToGrid (Frame.cc / FrameBase.cpp:60-93) as a CSR: one workgroup per frame,
// LDS histogram over the 64 x 48 cells, prefix sum, fill, then every cell's short list is put in
// ascending key order (= the reference's push_back order).
static const int kGridCells = kGridCols * kGridRows;

__global__ void __launch_bounds__(256) k_sbp_grid(SbpArgs A, int* __restrict__ cell_start,
                                                  unsigned short* __restrict__ cell_list) {
  __shared__ int s_cnt[kGridCells + 1];
  __shared__ int s_cur[kGridCells];
  __shared__ unsigned short s_list[kMaxKeys];
  __shared__ int s_part[256];
  const int f = blockIdx.x, tid = threadIdx.x;
  const int img = A.img_first + f * A.img_step;
  const int N = min(A.counts[2 * img], A.key_cap);
  const vieo_keypoint* K = A.keys + (size_t)img * A.key_cap;
  const float winv = (float)kGridCols / (A.maxx - A.minx), hinv = (float)kGridRows / (A.maxy - A.miny);
  for (int c = tid; c <= kGridCells; c += 256) s_cnt[c] = 0;
  __syncthreads();
  for (int j = tid; j < N; j += 256) {
    const int posX = (int)roundf((K[j].x - A.minx) * winv), posY = (int)roundf((K[j].y - A.miny) * hinv);
    if (posX >= 0 && posX < kGridCols && posY >= 0 && posY < kGridRows) atomicAdd(&s_cnt[posX * kGridRows + posY], 1);
  }
  __syncthreads();
  const int per = kGridCells / 256, c0 = tid * per;  // 3072 = 256 * 12
  int sum = 0;
  for (int c = c0; c < c0 + per; c++) sum += s_cnt[c];
  s_part[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int t = 0; t < 256; t++) {
      const int v = s_part[t];
      s_part[t] = acc;
      acc += v;
    }
  }
  __syncthreads();
  int acc = s_part[tid];
  int* cs = cell_start + (size_t)f * (kGridCells + 1);
  for (int c = c0; c < c0 + per; c++) {
    const int v = s_cnt[c];
    cs[c] = acc, s_cur[c] = acc;
    acc += v;
  }
  if (tid == 255) cs[kGridCells] = acc;
  __syncthreads();
  for (int j = tid; j < N; j += 256) {
    const int posX = (int)roundf((K[j].x - A.minx) * winv), posY = (int)roundf((K[j].y - A.miny) * hinv);
    if (posX >= 0 && posX < kGridCols && posY >= 0 && posY < kGridRows)
      s_list[atomicAdd(&s_cur[posX * kGridRows + posY], 1)] = (unsigned short)j;
  }
  __syncthreads();
  unsigned short* out = cell_list + (size_t)f * A.key_cap;
  for (int c = c0; c < c0 + per; c++) {
    const int e = s_cur[c], b = e - s_cnt[c];
    for (int i = b + 1; i < e; i++) {  // insertion sort, lists are a handful of entries
      const unsigned short v = s_list[i];
      int k = i - 1;
      while (k >= b && s_list[k] > v) s_list[k + 1] = s_list[k], k--;
      s_list[k + 1] = v;
    }
    for (int i = b; i < e; i++) out[i] = s_list[i];
  }
}

__device__ __forceinline__ int wave_excl_scan_i(int v, int lane, int* total) {
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o);
    if (lane >= o) x += y;
  }
  *total = __shfl(x, 63);
  return x - v;
}

// grid (ceil(q_cap/4), n_frames): one wave per query.  GetFeaturesInArea (FrameBase.cpp:95-174):
// the lanes take the cells of the window in (ix, iy) order, so the candidates come out in the
// reference's order without a sort.
__global__ void __launch_bounds__(256) k_sbp_candidates(SbpArgs A) {
  const int f = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + wave;
  if (q >= A.q_cap) return;
  int* out_n = A.cand_n + (size_t)f * A.q_cap + q;
  if (q >= A.nq[f]) {
    if (lane == 0) *out_n = 0;
    return;
  }
  const vieo_proj_query& Q = A.queries[(size_t)f * A.q_cap + q];
  const int img = A.img_first + f * A.img_step;
  if (!(Q.flags & 1)) {
    if (lane == 0) *out_n = 0;
    return;
  }
  const float x = Q.u, y = Q.v, r = Q.radius;
  const float winv = (float)kGridCols / (A.maxx - A.minx), hinv = (float)kGridRows / (A.maxy - A.miny);
  // FrameBase.cpp:102-115
  const int min_cellx = max(0, (int)floorf((x - A.minx - r) * winv));
  const int max_cellx = min(kGridCols - 1, (int)ceilf((x - A.minx + r) * winv));
  const int min_celly = max(0, (int)floorf((y - A.miny - r) * hinv));
  const int max_celly = min(kGridRows - 1, (int)ceilf((y - A.miny + r) * hinv));
  if (min_cellx >= kGridCols || max_cellx < 0 || min_celly >= kGridRows || max_celly < 0 ||
      min_cellx > max_cellx || min_celly > max_celly) {
    if (lane == 0) *out_n = 0;
    return;
  }
  const int minlevel = Q.level_min, maxlevel = Q.level_max;
  const bool bchecklevel = (minlevel > 0) || (maxlevel >= 0);
  const uint4 a0 = ((const uint4*)Q.desc)[0], a1 = ((const uint4*)Q.desc)[1];
  const vieo_keypoint* K = A.keys + (size_t)img * A.key_cap;
  const uint8_t* D = A.desc + (size_t)img * A.key_cap * 32;
  const int* cs = A.cell_start + (size_t)f * (kGridCells + 1);
  const unsigned short* cl = A.cell_list + (size_t)f * A.key_cap;
  unsigned* dst = A.cand + ((size_t)f * A.q_cap + q) * kCandCap;
  const int ny = max_celly - min_celly + 1, ncell = (max_cellx - min_cellx + 1) * ny;
  int n = 0;
  for (int c0 = 0; c0 < ncell; c0 += 64) {
    const int c = c0 + lane;
    int s = 0, e = 0;
    if (c < ncell) {
      const int cx = c / ny, cell = (min_cellx + cx) * kGridRows + min_celly + (c - cx * ny);
      s = cs[cell], e = cs[cell + 1];
    }
    int cnt = 0;
    for (int t = s; t < e; t++) {
      const vieo_keypoint& k = K[cl[t]];
      bool pass = true;
      if (bchecklevel) {
        if (k.octave < minlevel) pass = false;
        if (maxlevel >= 0 && k.octave > maxlevel) pass = false;
      }
      if (pass) pass = fabsf(k.x - x) < r && fabsf(k.y - y) < r;
      cnt += pass;
    }
    int total;
    int w = n + wave_excl_scan_i(cnt, lane, &total);
    for (int t = s; t < e && cnt > 0; t++) {
      const int j = cl[t];
      const vieo_keypoint& k = K[j];
      bool pass = true;
      if (bchecklevel) {
        if (k.octave < minlevel) pass = false;
        if (maxlevel >= 0 && k.octave > maxlevel) pass = false;
      }
      if (pass) pass = fabsf(k.x - x) < r && fabsf(k.y - y) < r;
      if (pass) {
        const int d = hamming32q(a0, a1, D + (size_t)j * 32);
        if (w < kCandCap) dst[w] = (unsigned)j | ((unsigned)d << 12) | ((unsigned)(k.octave & 15) << 21);
        w++;
      }
    }
    n += total;
  }
  if (lane == 0) *out_n = n > kCandCap ? -1 : n;
}

// one wave per frame
__global__ void __launch_bounds__(64) k_sbp_assign(SbpArgs A) {
  __shared__ uint8_t s_state[kMaxKeys];  // bit0: holds a map point, bit1: it has Observations()>0
  __shared__ unsigned short s_log_idx[kMaxKeys];
  __shared__ uint8_t s_log_bin[kMaxKeys];
  __shared__ int s_hist[kHistoLen];
  const int f = blockIdx.x, lane = threadIdx.x;
  const int img = A.img_first + f * A.img_step;
  const int N = min(A.counts[2 * img], A.key_cap);
  const int nq = min(A.nq[f], A.q_cap);
  int* assign = A.assign + (size_t)f * A.key_cap;
  const uint8_t* taken = A.taken ? A.taken + (size_t)f * A.key_cap : nullptr;
  const float* uright = A.uright + (size_t)f * A.key_cap;
  const vieo_keypoint* K = A.keys + (size_t)img * A.key_cap;
  for (int i = lane; i < N; i += 64) {
    assign[i] = VIEO_SBP_UNCHANGED;
    s_state[i] = (taken && taken[i]) ? 3 : 0;
  }
  if (lane < kHistoLen) s_hist[lane] = 0;
  __syncthreads();
  int nmatches = 0, nlog = 0, overflow = 0;
  const float factor = 1.0f / kHistoLen;
  for (int q = 0; q < nq; q++) {
    const int n = A.cand_n[(size_t)f * A.q_cap + q];
    if (n == 0) continue;
    if (n < 0) {
      overflow = 1;
      continue;
    }
    const vieo_proj_query& Q = A.queries[(size_t)f * A.q_cap + q];
    const unsigned* cand = A.cand + ((size_t)f * A.q_cap + q) * kCandCap;
    // two smallest (dist, order) among usable candidates; lane owns positions lane, lane+64
    unsigned b0 = 0xFFFFFFFFu, b1 = 0xFFFFFFFFu;  // dist<<20 | pos<<12... packed below
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int pos = lane + 64 * h;
      if (pos < n) {
        const unsigned c = cand[pos];
        const int idx = c & 0xFFF, d = (c >> 12) & 0x1FF;
        bool use = !((s_state[idx] & 1) && (s_state[idx] & 2));
        if (use && uright[idx] > 0) {
          const float er = fabsf(Q.ur - uright[idx]);
          if (er > Q.radius) use = false;
        }
        if (use) {
          const unsigned key = ((unsigned)d << 8) | (unsigned)pos;  // (dist, order)
          if (key < b0) {
            b1 = b0;
            b0 = key;
          } else if (key < b1)
            b1 = key;
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned e0 = __shfl_xor(b0, o), e1 = __shfl_xor(b1, o);
      if (e0 < b0) {
        b1 = min(b0, e1);
        b0 = e0;
      } else
        b1 = min(b1, e0);
    }
    if (b0 == 0xFFFFFFFFu) continue;
    const int bestDist = b0 >> 8, bestPos = b0 & 0xFF;
    const unsigned cb = cand[bestPos];
    const int bestIdx = cb & 0xFFF, bestLevel = (cb >> 21) & 15;
    if (bestDist > kThHigh) continue;
    if (A.mode == VIEO_SBP_LOCAL_MAP && b1 != 0xFFFFFFFFu) {
      const int bestDist2 = b1 >> 8;
      const int bestLevel2 = (cand[b1 & 0xFF] >> 21) & 15;
      if (bestLevel == bestLevel2 && (float)bestDist > A.nn_ratio * (float)bestDist2) continue;
    }
    // AddMapPoint(pMP, bestIdx)
    if (lane == 0) {
      s_state[bestIdx] = 1 | ((Q.flags & 2) ? 2 : 0);
      assign[bestIdx] = q;
    }
    nmatches++;
    if (A.mode == VIEO_SBP_LAST_FRAME && A.check_ori) {
      float rot = Q.angle - K[bestIdx].angle;
      if (rot < 0.0f) rot += 360.0f;
      int bin = (int)roundf(rot * factor);
      if (bin == kHistoLen) bin = 0;
      if (lane == 0) {
        s_log_idx[nlog] = (unsigned short)bestIdx;
        s_log_bin[nlog] = (uint8_t)bin;
        s_hist[bin]++;
      }
      nlog++;
    }
    __syncthreads();  // single wave: orders the LDS state update before the next query
  }
  if (A.mode == VIEO_SBP_LAST_FRAME && A.check_ori) {
    __syncthreads();
    // ComputeThreeMaxima (ORBmatcher.cc:1608-1641), evaluated redundantly by every lane
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < kHistoLen; i++) {
      const int s = s_hist[i];
      if (s > max1) {
        max3 = max2, max2 = max1, max1 = s;
        ind3 = ind2, ind2 = ind1, ind1 = i;
      } else if (s > max2) {
        max3 = max2, max2 = s;
        ind3 = ind2, ind2 = i;
      } else if (s > max3) {
        max3 = s, ind3 = i;
      }
    }
    if (max2 < 0.1f * (float)max1) {
      ind2 = -1, ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
      ind3 = -1;
    }
    for (int k = lane; k < nlog; k += 64) {
      const int bin = s_log_bin[k];
      if (bin != ind1 && bin != ind2 && bin != ind3) assign[s_log_idx[k]] = VIEO_SBP_ERASED;
    }
    for (int i = 0; i < kHistoLen; i++)
      if (i != ind1 && i != ind2 && i != ind3) nmatches -= s_hist[i];
  }
  if (lane == 0) A.nmatches[f] = overflow ? -1 : nmatches;
}

struct SbpScratch {
  DevBuf cand, cand_n, q, nq, keys, ur, desc, taken, counts, assign, nm, pts, cam, cell_start, cell_list;
};
static thread_local SbpScratch g_sbp;

static int run_search(SbpArgs& A, int n_frames, hipStream_t st) {
  int rc;
  SbpScratch& S = g_sbp;
  if ((rc = S.cand.ensure((size_t)n_frames * A.q_cap * kCandCap * 4)) != VIEO_OK) return rc;
  if ((rc = S.cand_n.ensure((size_t)n_frames * A.q_cap * 4)) != VIEO_OK) return rc;
  A.cand = S.cand.as<unsigned>();
  A.cand_n = S.cand_n.as<int>();
  if (A.key_cap > kMaxKeys) {
    set_error("search_by_projection: more than %d keypoints per frame", kMaxKeys);
    return VIEO_E_CAPACITY;
  }
  if ((rc = S.cell_start.ensure((size_t)n_frames * (kGridCells + 1) * 4)) != VIEO_OK) return rc;
  if ((rc = S.cell_list.ensure((size_t)n_frames * A.key_cap * 2)) != VIEO_OK) return rc;
  A.cell_start = S.cell_start.as<int>(), A.cell_list = S.cell_list.as<unsigned short>();
  hipLaunchKernelGGL(k_sbp_grid, dim3(n_frames), dim3(256), 0, st, A, S.cell_start.as<int>(),
                     S.cell_list.as<unsigned short>());
  hipLaunchKernelGGL(k_sbp_candidates, dim3((A.q_cap + 3) / 4, n_frames), dim3(256), 0, st, A);
  hipLaunchKernelGGL(k_sbp_assign, dim3(n_frames), dim3(64), 0, st, A);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

}  // namespace vieo

using namespace vieo;

extern "C" {

int vieo_sbp_project_last_frame_batch_device(const vieo_last_frame_point* d_points,
                                             const int32_t* d_n, int p_cap, int n_frames,
                                             const vieo_sbp_camera* d_cams,
                                             vieo_proj_query* d_queries, void* stream) {
  if (!d_points || !d_n || p_cap <= 0 || n_frames <= 0 || !d_cams || !d_queries) return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  hipLaunchKernelGGL(k_sbp_project, dim3((p_cap + 255) / 256, n_frames), dim3(256), 0,
                     (hipStream_t)stream, d_points, d_n, p_cap, d_cams, d_queries);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

int vieo_search_by_projection_batch_device(int mode, const vieo_proj_query* d_queries,
                                           const int32_t* d_nq, int q_cap, int n_frames,
                                           const vieo_keypoint* d_keys, const float* d_uright,
                                           const uint8_t* d_desc, const uint8_t* d_taken,
                                           const int32_t* d_counts, int key_cap, int img_first,
                                           int img_step, const float* h_bounds, float nn_ratio,
                                           int check_orientation, int32_t* d_assign,
                                           int32_t* d_nmatches, void* stream) {
  if (!d_queries || !d_nq || q_cap <= 0 || n_frames <= 0 || !d_keys || !d_uright || !d_desc ||
      !d_counts || !h_bounds || !d_assign || !d_nmatches ||
      (mode != VIEO_SBP_LAST_FRAME && mode != VIEO_SBP_LOCAL_MAP))
    return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  SbpArgs A;
  A.mode = mode;
  A.queries = d_queries, A.nq = d_nq, A.q_cap = q_cap;
  A.keys = d_keys, A.uright = d_uright, A.desc = d_desc, A.taken = d_taken, A.counts = d_counts;
  A.key_cap = key_cap, A.img_first = img_first, A.img_step = img_step;
  A.minx = h_bounds[0], A.maxx = h_bounds[1], A.miny = h_bounds[2], A.maxy = h_bounds[3];
  A.nn_ratio = nn_ratio, A.check_ori = check_orientation;
  A.assign = d_assign, A.nmatches = d_nmatches;
  return run_search(A, n_frames, (hipStream_t)stream);
}

int vieo_sbp_project_last_frame(const vieo_last_frame_point* h_points, int n,
                                const vieo_sbp_camera* h_cam, vieo_proj_query* h_queries) {
  if (n < 0 || !h_cam || (n > 0 && (!h_points || !h_queries))) return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  if (n == 0) return VIEO_OK;
  SbpScratch& S = g_sbp;
  if ((rc = S.pts.ensure((size_t)n * sizeof(vieo_last_frame_point))) != VIEO_OK) return rc;
  if ((rc = S.cam.ensure(sizeof(vieo_sbp_camera))) != VIEO_OK) return rc;
  if ((rc = S.q.ensure((size_t)n * sizeof(vieo_proj_query))) != VIEO_OK) return rc;
  if ((rc = S.nq.ensure(4)) != VIEO_OK) return rc;
  VIEO_HIP_CHECK(hipMemcpy(S.pts.p, h_points, (size_t)n * sizeof(vieo_last_frame_point), hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(hipMemcpy(S.cam.p, h_cam, sizeof(vieo_sbp_camera), hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(hipMemcpy(S.nq.p, &n, 4, hipMemcpyHostToDevice));
  rc = vieo_sbp_project_last_frame_batch_device(S.pts.as<vieo_last_frame_point>(), S.nq.as<int>(), n, 1,
                                                S.cam.as<vieo_sbp_camera>(), S.q.as<vieo_proj_query>(),
                                                nullptr);
  if (rc != VIEO_OK) return rc;
  VIEO_HIP_CHECK(hipMemcpy(h_queries, S.q.p, (size_t)n * sizeof(vieo_proj_query), hipMemcpyDeviceToHost));
  return VIEO_OK;
}

int vieo_search_by_projection(int mode, const vieo_proj_query* h_queries, int nq,
                              const vieo_keypoint* h_keys, const float* h_uright,
                              const uint8_t* h_desc, const uint8_t* h_taken, int n_keys,
                              const float* h_bounds, float nn_ratio, int check_orientation,
                              int32_t* h_assign, int32_t* nmatches) {
  if (nq < 0 || n_keys < 0 || !h_bounds || !nmatches || (n_keys > 0 && (!h_keys || !h_uright || !h_desc || !h_assign)))
    return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  *nmatches = 0;
  for (int i = 0; i < n_keys; i++) h_assign[i] = VIEO_SBP_UNCHANGED;
  if (nq == 0 || n_keys == 0) return VIEO_OK;
  SbpScratch& S = g_sbp;
#define ENS(b, n) \
  if ((rc = (b).ensure(n)) != VIEO_OK) return rc
  ENS(S.q, (size_t)nq * sizeof(vieo_proj_query));
  ENS(S.nq, 4);
  ENS(S.keys, (size_t)n_keys * sizeof(vieo_keypoint));
  ENS(S.ur, (size_t)n_keys * 4);
  ENS(S.desc, (size_t)n_keys * 32);
  ENS(S.taken, (size_t)n_keys);
  ENS(S.counts, 8);
  ENS(S.assign, (size_t)n_keys * 4);
  ENS(S.nm, 4);
#undef ENS
  const int cnt[2] = {n_keys, 0};
  VIEO_HIP_CHECK(hipMemcpy(S.q.p, h_queries, (size_t)nq * sizeof(vieo_proj_query), hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(hipMemcpy(S.nq.p, &nq, 4, hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(hipMemcpy(S.keys.p, h_keys, (size_t)n_keys * sizeof(vieo_keypoint), hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(hipMemcpy(S.ur.p, h_uright, (size_t)n_keys * 4, hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(hipMemcpy(S.desc.p, h_desc, (size_t)n_keys * 32, hipMemcpyHostToDevice));
  if (h_taken) VIEO_HIP_CHECK(hipMemcpy(S.taken.p, h_taken, (size_t)n_keys, hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(hipMemcpy(S.counts.p, cnt, 8, hipMemcpyHostToDevice));
  rc = vieo_search_by_projection_batch_device(
      mode, S.q.as<vieo_proj_query>(), S.nq.as<int>(), nq, 1, S.keys.as<vieo_keypoint>(),
      S.ur.as<float>(), S.desc.as<uint8_t>(), h_taken ? S.taken.as<uint8_t>() : nullptr,
      S.counts.as<int>(), n_keys, 0, 0, h_bounds, nn_ratio, check_orientation, S.assign.as<int>(),
      S.nm.as<int>(), nullptr);
  if (rc != VIEO_OK) return rc;
  VIEO_HIP_CHECK(hipMemcpy(h_assign, S.assign.p, (size_t)n_keys * 4, hipMemcpyDeviceToHost));
  VIEO_HIP_CHECK(hipMemcpy(nmatches, S.nm.p, 4, hipMemcpyDeviceToHost));
  if (*nmatches < 0) {
    set_error("search_by_projection: more than %d window candidates for one query", kCandCap);
    return VIEO_E_CAPACITY;
  }
  return VIEO_OK;
}

}  // extern "C"
