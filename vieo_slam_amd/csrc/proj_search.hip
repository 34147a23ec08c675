// proj_search.hip -- tracking-side ORBmatcher::SearchByProjection on gfx950
// (reference: src/ORBmatcher.cc:230-335 and :1303-1467, grid query src/FrameBase.cpp:95-174).
//
// The reference is sequential over map points because a later query must skip keypoints that an
// earlier query already claimed (AddMapPoint).  Split accordingly:
//   k_sbp_project     (a12 only) last frame's map points -> window queries           [parallel]
//   k_sbp_grid        Frame::mGrid (64 x 48 cells) as a CSR, built per frame            [parallel]
//   k_sbp_candidates  one wavefront per query: the window's cells in GetFeaturesInArea order,
//                     octave / window / stereo-coordinate tests, Hamming distance and rotation
//                     bin per candidate, appended to the frame's candidate pool        [parallel]
//   k_sbp_assign      one wavefront per frame copies the pool to LDS and replays the queries in
//                     order against the "claimed" flags: best / second-best, accept rules,
//                     rotation histogram + ComputeThreeMaxima                          [sequential]
// Integer/index work: results are bit-exact against oracle/proj_search.cc.
#include <climits>
#include <cstring>

#include "ba_device.h"
#include "orb_internal.h"
#include "wave_ops.h"

namespace vieo {

static const int kGridRows = 48, kGridCols = 64;  // FrameBase.h:224-225
static const int kThHigh = 100, kHistoLen = 30;   // ORBmatcher.cc:20-22
static const int kCandCap = 128;                  // candidates kept per query (2 per lane)
static const int kMaxKeys = 8192;                 // 13-bit key index packing (4 cameras x 1500 features fit)
static const int kMaxCams = 4;

// the cameras of a rig frame on the device (vieo_sbp_rig widened once per thread block user)
__device__ __forceinline__ void rig_uv(const vieo_sbp_rig& R, int cam, const double* x3Dc, float invzc, float* u, float* v) {
  const vieo_camera& c = R.cams[cam];
  if (R.use_distort) {
    CamD d;
    cam_from_abi(c, d);
    double uv[2];
    cam_project(d, x3Dc, uv, nullptr);
    *u = (float)uv[0], *v = (float)uv[1];
  } else {
    const float xc = (float)x3Dc[0], yc = (float)x3Dc[1];
    const float pnx = xc * invzc, pny = yc * invzc;
    *u = c.fx * pnx + 0.f * pny + c.cx * 1.f;
    *v = 0.f * pnx + c.fy * pny + c.cy * 1.f;
  }
}

// ORBmatcher.cc:1313-1378.  One thread per (last-frame point, camera): rigs == nullptr is the single rectified
// camera of vieo_sbp_camera (n_cams = 1), otherwise query (i, camj) goes to out[i * n_cams + camj].
__global__ void __launch_bounds__(256)
k_sbp_project(const vieo_last_frame_point* __restrict__ pts, const int* __restrict__ n_pts,
              int p_cap, const vieo_sbp_camera* __restrict__ cams, const vieo_sbp_rig* __restrict__ rigs,
              int n_cams, vieo_proj_query* __restrict__ out) {
  const int f = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p_cap * n_cams) return;
  const int i = t / n_cams, camj = t - i * n_cams;
  vieo_proj_query q;
  memset(&q, 0, sizeof(q));
  vieo_proj_query* dst = out + (size_t)f * p_cap * n_cams + t;
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
  double x3[3];
  x3[0] = Tc[0] * X + Tc[1] * Y + Tc[2] * Z + Tc[3];
  x3[1] = Tc[4] * X + Tc[5] * Y + Tc[6] * Z + Tc[7];
  x3[2] = Tc[8] * X + Tc[9] * Y + Tc[10] * Z + Tc[11];
  if (C.th_far > 0 && x3[2] > C.th_far) ok = false;
  float u, v, invzc;
  const float* bnd = C.bounds;
  if (!rigs) {
    const float xc = (float)x3[0], yc = (float)x3[1];
    invzc = (float)(1.0 / x3[2]);
    const float pnx = xc * invzc, pny = yc * invzc;
    u = C.fx * pnx + 0.f * pny + C.cx * 1.f;
    v = 0.f * pnx + C.fy * pny + C.cy * 1.f;
  } else {
    const vieo_sbp_rig& R = rigs[f];
    const double* T = R.Tcr[camj];
    double xc[3];
    for (int r = 0; r < 3; ++r) xc[r] = T[r * 4] * x3[0] + T[r * 4 + 1] * x3[1] + T[r * 4 + 2] * x3[2] + T[r * 4 + 3];
    invzc = (float)(1.0 / xc[2]);
    u = v = 0.f;
    if (!(invzc < 0)) rig_uv(R, camj, xc, invzc, &u, &v);
    bnd = R.bounds[camj];
  }
  if (invzc < 0) ok = false;
  if (!(u >= bnd[0] && u < bnd[1] && v >= bnd[2] && v < bnd[3])) ok = false;
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
    q.flags = 1 | (p.flags & 2) | (camj << 8);
    for (int k = 0; k < 32; k++) q.desc[k] = p.desc[k];
  }
  *dst = q;
}

// ORBmatcher.cc:1487-1543, the projection of the relocalisation overload (see oracle/proj_search.cc)
__global__ void __launch_bounds__(256)
k_sbp_project_kf(const vieo_keyframe_point* __restrict__ pts, int n, const vieo_sbp_camera* __restrict__ cam,
                 const vieo_sbp_rig* __restrict__ rig, int n_cams, float log_scale_factor,
                 vieo_proj_query* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * n_cams) return;
  const int i = t / n_cams, cami = t - i * n_cams;
  vieo_proj_query q;
  memset(&q, 0, sizeof(q));
  const vieo_sbp_camera& C = *cam;
  const vieo_keyframe_point p = pts[i];
  bool ok = p.flags & 1;
  const double* Tc = C.Tcw_cur;
  const double Xw[3] = {p.Xw[0], p.Xw[1], p.Xw[2]};
  double x3[3];
  for (int r = 0; r < 3; ++r) x3[r] = Tc[r * 4] * Xw[0] + Tc[r * 4 + 1] * Xw[1] + Tc[r * 4 + 2] * Xw[2] + Tc[r * 4 + 3];
  if (C.th_far > 0 && x3[2] > C.th_far) ok = false;
  // Twcr = Tcrw.inverse()
  double tw[3];
  for (int r = 0; r < 3; ++r) tw[r] = Tc[r] * (Tc[3] * -1.0) + Tc[4 + r] * (Tc[7] * -1.0) + Tc[8 + r] * (Tc[11] * -1.0);
  double Pc[3], twc[3];
  float u, v, invzc;
  const float* bnd = C.bounds;
  if (!rig) {
    for (int r = 0; r < 3; ++r) Pc[r] = x3[r], twc[r] = tw[r];  // identity Tcr, zero trc
    invzc = (float)(1.0 / Pc[2]);
    const float xc = (float)Pc[0], yc = (float)Pc[1];
    const float pnx = xc * invzc, pny = yc * invzc;
    u = C.fx * pnx + 0.f * pny + C.cx * 1.f;
    v = 0.f * pnx + C.fy * pny + C.cy * 1.f;
  } else {
    const double* T = rig->Tcr[cami];
    for (int r = 0; r < 3; ++r) Pc[r] = T[r * 4] * x3[0] + T[r * 4 + 1] * x3[1] + T[r * 4 + 2] * x3[2] + T[r * 4 + 3];
    const double* tr = rig->trc[cami];
    for (int r = 0; r < 3; ++r) twc[r] = tw[r] + (Tc[r] * tr[0] + Tc[4 + r] * tr[1] + Tc[8 + r] * tr[2]);
    invzc = (float)(1.0 / Pc[2]);
    rig_uv(*rig, cami, Pc, invzc, &u, &v);
    bnd = rig->bounds[cami];
  }
  if (!(u >= bnd[0] && u < bnd[1] && v >= bnd[2] && v < bnd[3])) ok = false;
  const double PO[3] = {Xw[0] - twc[0], Xw[1] - twc[1], Xw[2] - twc[2]};
  const float dist3D = (float)sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
  if (dist3D < 0.8f * p.min_distance || dist3D > 1.2f * p.max_distance) ok = false;
  if (ok) {
    const float ratio = p.max_distance / dist3D;
    int lvl = (int)ceilf((float)log((double)ratio) / log_scale_factor);  // PredictScale, see oracle/mappoint.cc
    if (lvl < 0)
      lvl = 0;
    else if (lvl >= C.nlevels)
      lvl = C.nlevels - 1;
    q.u = u, q.v = v, q.ur = u - C.bf * invzc;
    q.radius = C.th * C.scale[lvl];
    q.level_min = lvl - 1, q.level_max = lvl + 1;
    q.angle = p.angle;
    q.flags = 1 | (p.flags & 2) | (cami << 8);
    for (int k = 0; k < 32; k++) q.desc[k] = p.desc[k];
  }
  out[t] = q;
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
  int n_cams;                       // cameras of a frame (1 for the single-camera entries)
  const int* cam_first;             // [frame][n_cams + 1] key ranges of the cameras, or null: one camera, N = counts
  float bounds[kMaxCams][4];        // gridinfo_.minmax_xy_[cam]
  float nn_ratio;
  int check_ori;
  const int* cell_start;            // [frame][cam][kGridCols * kGridRows + 1]  vgrids_[cam] as CSR (camera-local offsets)
  const float4* cell_rec;           // [frame][key_cap] keys in (camera, cell) order: x, y, uright, idx | octave << 16
  const float* cell_ang;            // [frame][key_cap] their angles
  unsigned* pool;      // [frame][pool_cap] candidates of all queries: idx | dist<<13 | level<<22 | bin<<26
  int* cursor;         // [frame] entries used in the pool
  int2* qrec;          // [frame][q_cap] {offset in the pool, count (-1: overflow) | has-observations << 16}
  int pool_cap, pool_lds;
  int* assign;         // [frame][key_cap]
  int* nmatches;       // [frame]
  int* claim;          // [frame][q_cap] scratch of k_sbp_assign_par: the key every query claims (-1 none)
  int* need_seq;       // [frame] 1: k_sbp_assign_par gave up, the sequential replay (k_sbp_assign) does the frame
  int max_rounds;      // of the optimistic assignment
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

// keys [k0, k1) of camera `cam` of frame f
__device__ __forceinline__ void cam_range(const SbpArgs& A, int f, int cam, int* k0, int* k1) {
  if (A.cam_first) {
    const int* cf = A.cam_first + (size_t)f * (A.n_cams + 1);
    *k0 = min(cf[cam], A.key_cap), *k1 = min(cf[cam + 1], A.key_cap);
  } else {
    *k0 = 0, *k1 = min(A.counts[2 * (A.img_first + f * A.img_step)], A.key_cap);
  }
}

// NT threads: 1024 for a call of a few frames, 256 for batches
template <int NT>
__global__ void __launch_bounds__(NT) k_sbp_grid(SbpArgs A, int* __restrict__ cell_start,
                                                  float4* __restrict__ cell_rec,
                                                  float* __restrict__ cell_ang) {
  __shared__ int s_cnt[kGridCells + 1];
  __shared__ int s_cur[kGridCells];
  __shared__ unsigned short s_list[kMaxKeys];
  __shared__ int s_part[16];
  const int f = blockIdx.x / A.n_cams, cam = blockIdx.x - f * A.n_cams, tid = threadIdx.x;
  const int img = A.img_first + f * A.img_step;
  int k0, k1;
  cam_range(A, f, cam, &k0, &k1);
  const int N = k1 - k0;
  const vieo_keypoint* K = A.keys + (size_t)img * A.key_cap + k0;
  const float minx = A.bounds[cam][0], miny = A.bounds[cam][2];
  const float winv = (float)kGridCols / (A.bounds[cam][1] - minx), hinv = (float)kGridRows / (A.bounds[cam][3] - miny);
  for (int c = tid; c <= kGridCells; c += NT) s_cnt[c] = 0;
  __syncthreads();
  for (int j = tid; j < N; j += NT) {
    const int posX = (int)roundf((K[j].x - minx) * winv), posY = (int)roundf((K[j].y - miny) * hinv);
    if (posX >= 0 && posX < kGridCols && posY >= 0 && posY < kGridRows) atomicAdd(&s_cnt[posX * kGridRows + posY], 1);
  }
  __syncthreads();
  const int per = kGridCells / NT, c0 = tid * per;  // 3072 = 1024 * 3
  int sum = 0;
  for (int c = c0; c < c0 + per; c++) sum += s_cnt[c];
  // exclusive scan of the partial sums: within the wavefronts, then over the four of them (one thread walking the
  // 256 values was 256 dependent LDS round trips, ~10 us of this kernel's 23)
  int inc = sum;
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o);
    if ((tid & 63) >= o) inc += t;
  }
  if ((tid & 63) == 63) s_part[tid >> 6] = inc;
  __syncthreads();
  int acc = inc - sum;
  for (int w = 0; w < (tid >> 6); w++) acc += s_part[w];
  int* cs = cell_start + (size_t)blockIdx.x * (kGridCells + 1);
  for (int c = c0; c < c0 + per; c++) {
    const int v = s_cnt[c];
    cs[c] = acc, s_cur[c] = acc;
    acc += v;
  }
  if (tid == NT - 1) {
    cs[kGridCells] = acc;
    if (cam == 0) A.cursor[f] = 0;
  }
  __syncthreads();
  for (int j = tid; j < N; j += NT) {
    const int posX = (int)roundf((K[j].x - minx) * winv), posY = (int)roundf((K[j].y - miny) * hinv);
    if (posX >= 0 && posX < kGridCols && posY >= 0 && posY < kGridRows)
      s_list[atomicAdd(&s_cur[posX * kGridRows + posY], 1)] = (unsigned short)j;
  }
  __syncthreads();
  for (int c = c0; c < c0 + per; c++) {
    const int e = s_cur[c], b = e - s_cnt[c];
    for (int i = b + 1; i < e; i++) {  // insertion sort, lists are a handful of entries
      const unsigned short v = s_list[i];
      int k = i - 1;
      while (k >= b && s_list[k] > v) s_list[k + 1] = s_list[k], k--;
      s_list[k + 1] = v;
    }
  }
  __syncthreads();
  // the keys in cell order, with everything a window test reads in one 16-byte record; the index is the key's
  // position in the frame's list (mvKeys), i.e. camera-local + k0
  const int n_in = s_cur[kGridCells - 1];
  const float* uright = A.uright + (size_t)f * A.key_cap + k0;
  float4* rec = cell_rec + (size_t)f * A.key_cap + k0;
  float* ang = cell_ang + (size_t)f * A.key_cap + k0;
  for (int i = tid; i < n_in; i += NT) {
    const int j = s_list[i];
    const vieo_keypoint k = K[j];
    rec[i] = make_float4(k.x, k.y, uright[j], __int_as_float((j + k0) | (k.octave << 16)));
    ang[i] = k.angle;
  }
}

__device__ __forceinline__ int wave_excl_scan_i(int v, int lane, int* total) {
  // inclusive scan on DPP: four shifted adds inside a row of 16 lanes (zeros shift in), then the two row broadcasts
  int x = v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);  // row_shr:8
  x += VIEO_DPP(0, x, VIEO_DPP_ROW_BCAST15, 0xA);
  x += VIEO_DPP(0, x, VIEO_DPP_ROW_BCAST31, 0xC);
  (void)lane;
  *total = __builtin_amdgcn_readlane(x, 63);
  return x - v;
}

// grid (kSbpBlocks, n_frames), 4 wavefronts per workgroup, each walking the frame's queries with a
// stride; the frame's cell offsets sit in LDS, the next query is fetched while the current one is
// processed.  GetFeaturesInArea (FrameBase.cpp:95-174): cells of one grid column are consecutive in
// the CSR, so a window is nx contiguous runs of records, already in the reference's order (ix outer,
// iy inner, key index inside a cell) -- no sort.
static const int kSbpBlocks = 8;      // workgroups per frame of a batch
static const int kSbpBlocksFew = 128; // ... of a call with a frame or two (the one-call tracker): latency, not occupancy

__global__ void __launch_bounds__(256) k_sbp_candidates(SbpArgs A) {
  extern __shared__ unsigned short s_cs[];  // [n_cams][kGridCells + 1] camera-local offsets (< kMaxKeys: 16 bits)
  constexpr int kBigCap = 1024;
  __shared__ int s_big[kBigCap];  // the queries pass 1 leaves to pass 2
  __shared__ int s_nbig;
  if (threadIdx.x == 0) s_nbig = 0;
  // (wave-uniform wavefront index: the query walk then runs on scalar registers and scalar loads)
  const int f = blockIdx.y, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int nq = min(A.nq[f], A.q_cap);
  const int img = A.img_first + f * A.img_step;
  {
    const int* cs = A.cell_start + (size_t)f * A.n_cams * (kGridCells + 1);
    for (int i = threadIdx.x; i < A.n_cams * (kGridCells + 1); i += 256) s_cs[i] = (unsigned short)cs[i];
  }
  // queries past nq: empty records (the replay never reads them, but keep the buffer defined)
  const int n_blocks = gridDim.x;
  for (int q = nq + blockIdx.x * 256 + threadIdx.x; q < A.q_cap; q += n_blocks * 256)
    A.qrec[(size_t)f * A.q_cap + q] = make_int2(0, 0);
  __syncthreads();
  const uint8_t* D = A.desc + (size_t)img * A.key_cap * 32;
  const float4* rec_f = A.cell_rec + (size_t)f * A.key_cap;
  const float* ang_f = A.cell_ang + (size_t)f * A.key_cap;
  unsigned* pool = A.pool + (size_t)f * A.pool_cap;
  const float factor = 1.0f / kHistoLen;
  const uint4* QQ = (const uint4*)(A.queries + (size_t)f * A.q_cap);
  // Pool space comes in slabs of kCandCap entries per wavefront (a query keeps at most kCandCap candidates): one global
  // atomic per slab instead of one per query -- the atomic's round trip was one of the three dependent ones of every
  // query (window records -> reservation -> descriptors).  The tail of a slab stays unused; the queries' records carry
  // their own base, nothing reads the pool front to back.
  int slab_cur = 0, slab_end = 0;  // wave-uniform
  auto reserve = [&](int total) -> int {
    if (slab_cur + total > slab_end) {
      int b = 0;
      if (lane == 0) b = atomicAdd(&A.cursor[f], kCandCap);
      slab_cur = __builtin_amdgcn_readfirstlane(b), slab_end = slab_cur + kCandCap;
    }
    const int base = slab_cur;
    slab_cur += total;
    return base;
  };
  // ---- pass 1: FOUR queries per wavefront, one per row of 16 lanes.  A window holds a handful of keys (1200 keys over
  // 3072 cells), so with a wavefront per query most lanes idled through three dependent round trips per query (window
  // records -> reservation -> descriptors) and a wavefront walked ~134 queries one after the other.  A row takes a query
  // whose window spans at most 16 grid columns and 64 entries (`small`, four steps of 16); the others are left to pass 2, the
  // wavefront-per-query form below, which recomputes the same predicate and skips the small ones.
  {
    const int g = lane >> 4, gl = lane & 15;
    const int stride4 = n_blocks * 16;
    uint4 n0 = {0, 0, 0, 0}, n1 = n0, n2 = n0, n3 = n0;  // the next iteration's record, fetched an iteration ahead
    {
      const int qf = (blockIdx.x * 4 + wave) * 4 + g;
      if (qf < nq) n0 = QQ[4 * (size_t)qf], n1 = QQ[4 * (size_t)qf + 1], n2 = QQ[4 * (size_t)qf + 2], n3 = QQ[4 * (size_t)qf + 3];
    }
    for (int q0 = (blockIdx.x * 4 + wave) * 4; q0 < nq; q0 += stride4) {
      const int qg = q0 + g;
      const bool qin = qg < nq;
      const uint4 g0 = n0, g1 = n1, e0 = n2, e1 = n3;
      {
        const int qn = qg + stride4;
        if (qn < nq) n0 = QQ[4 * (size_t)qn], n1 = QQ[4 * (size_t)qn + 1], n2 = QQ[4 * (size_t)qn + 2], n3 = QQ[4 * (size_t)qn + 3];
      }
      const float x = __uint_as_float(g0.x), y = __uint_as_float(g0.y), q_ur = __uint_as_float(g0.z);
      const float r = __uint_as_float(g0.w);
      const int minlevel = (int)g1.x, maxlevel = (int)g1.y, flags = (int)g1.w;
      const float q_angle = __uint_as_float(g1.z);
      const int cam = (flags >> 8) & 15;
      bool valid = qin && (flags & 1) && cam < A.n_cams;
      const int camc = valid ? cam : 0;
      const float minx = A.bounds[camc][0], miny = A.bounds[camc][2];
      const float winv = (float)kGridCols / (A.bounds[camc][1] - minx), hinv = (float)kGridRows / (A.bounds[camc][3] - miny);
      int k0 = 0;
      if (A.cam_first) k0 = min(A.cam_first[(size_t)f * (A.n_cams + 1) + camc], A.key_cap);
      const unsigned short* cs = s_cs + camc * (kGridCells + 1);
      const int min_cellx = max(0, (int)floorf((x - minx - r) * winv));
      const int max_cellx = min(kGridCols - 1, (int)ceilf((x - minx + r) * winv));
      const int min_celly = max(0, (int)floorf((y - miny - r) * hinv));
      const int max_celly = min(kGridRows - 1, (int)ceilf((y - miny + r) * hinv));
      const bool empty = min_cellx >= kGridCols || max_cellx < 0 || min_celly >= kGridRows || max_celly < 0 ||
                         min_cellx > max_cellx || min_celly > max_celly;
      const int nx = max_cellx - min_cellx + 1;
      const bool narrow = valid && !empty && nx <= 16;
      int seg_s = 0, seg_l = 0;
      if (narrow && gl < nx) {
        const int col = (min_cellx + gl) * kGridRows;
        seg_s = cs[col + min_celly];
        seg_l = cs[col + max_celly + 1] - seg_s;
      }
      // inclusive scan inside the row of 16 lanes (zeros shift in)
      int inc = seg_l;
      inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xF, 0xF, true);  // row_shr:1
      inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xF, 0xF, true);  // row_shr:2
      inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xF, 0xF, true);  // row_shr:4
      inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xF, 0xF, true);  // row_shr:8
      const int n_ent = __builtin_amdgcn_ds_bpermute((g * 16 + 15) * 4, inc);
      const bool small = narrow && n_ent <= 64;
      // up to four steps of 16 entries (uniform trip count: the largest small window of the four)
      const int ne_s = small ? n_ent : 0;
      const int ne_max = max(max(__builtin_amdgcn_readlane(ne_s, 0), __builtin_amdgcn_readlane(ne_s, 16)),
                             max(__builtin_amdgcn_readlane(ne_s, 32), __builtin_amdgcn_readlane(ne_s, 48)));
      const int nit = (ne_max + 15) >> 4;
      const float4* rec = rec_f + k0;
      const bool bchecklevel = (minlevel > 0) || (maxlevel >= 0);
      int sl_it[4], pk_it[4];
      unsigned long long m_it[4] = {0, 0, 0, 0};
#pragma unroll
      for (int it = 0; it < 4; it++) {
        sl_it[it] = 0, pk_it[it] = 0;
        if (it >= nit) continue;  // uniform
        // entry t of the window: the run it falls into = the number of runs that end at or before t (their inclusive
        // ends come round the row on DPP rotations; lanes past nx hold n_ent, which is > t for an entry)
        const int t = it * 16 + gl;
        int idx = (inc <= t) ? 1 : 0;
#define VIEO_ROR(n) idx += (__builtin_amdgcn_update_dpp(0, inc, 0x120 + (n), 0xF, 0xF, false) <= t) ? 1 : 0;
        VIEO_ROR(1) VIEO_ROR(2) VIEO_ROR(3) VIEO_ROR(4) VIEO_ROR(5) VIEO_ROR(6) VIEO_ROR(7) VIEO_ROR(8)
        VIEO_ROR(9) VIEO_ROR(10) VIEO_ROR(11) VIEO_ROR(12) VIEO_ROR(13) VIEO_ROR(14) VIEO_ROR(15)
#undef VIEO_ROR
        idx = min(idx, 15);
        const int src = (g * 16 + idx) * 4;
        const int run_s = __builtin_amdgcn_ds_bpermute(src, seg_s), run_inc = __builtin_amdgcn_ds_bpermute(src, inc),
                  run_l = __builtin_amdgcn_ds_bpermute(src, seg_l);
        const int sl = run_s + (t - (run_inc - run_l));
        bool pass = small && t < n_ent;
        if (pass) {
          const float4 k = rec[sl];
          const int pk = __float_as_int(k.w), oct = pk >> 16;
          pk_it[it] = pk;
          if (bchecklevel && (oct < minlevel || (maxlevel >= 0 && oct > maxlevel))) pass = false;
          if (!(fabsf(k.x - x) < r && fabsf(k.y - y) < r)) pass = false;
          if (A.mode != VIEO_SBP_RELOC && k.z > 0 && fabsf(q_ur - k.z) > r) pass = false;
        }
        sl_it[it] = sl;
        m_it[it] = __ballot(pass);
      }
      // totals per row (query) and the rows' offsets inside one reservation
      const int sh = 16 * g;
      int tg[4], tot_g = 0;
#pragma unroll
      for (int it = 0; it < 4; it++) tg[it] = __popc((unsigned)((m_it[it] >> sh) & 0xFFFFull)), tot_g += tg[it];
      const int T0 = __builtin_amdgcn_readlane(tot_g, 0), T1 = __builtin_amdgcn_readlane(tot_g, 16),
                T2 = __builtin_amdgcn_readlane(tot_g, 32), T3 = __builtin_amdgcn_readlane(tot_g, 48);
      // (a row whose window holds more than kCandCap candidates cannot happen here: at most 64 entries)
      const int tot_all = T0 + T1 + T2 + T3;
      const int off_g = g == 0 ? 0 : (g == 1 ? T0 : (g == 2 ? T0 + T1 : T0 + T1 + T2));
      int base = 0;
      bool over = false;
      if (tot_all > 0) {  // (uniform) one reservation for the four queries; 4 x 64 entries at most = two slabs
        if (tot_all > kCandCap) {  // larger than a slab: its own reservation
          int bb = 0;
          if (lane == 0) bb = atomicAdd(&A.cursor[f], tot_all);
          base = __builtin_amdgcn_readfirstlane(bb);
        } else
          base = reserve(tot_all);
        over = base + tot_all > A.pool_cap;
      }
      int wpos = base + off_g;
#pragma unroll
      for (int it = 0; it < 4; it++) {
        if (it >= nit) continue;
        const unsigned gm = (unsigned)((m_it[it] >> sh) & 0xFFFFull);
        if (((gm >> gl) & 1u) && !over) {
          const int packed = pk_it[it], sl = sl_it[it];
          const int j = packed & 0xFFFF, oct = packed >> 16;
          const int d = hamming32q(e0, e1, D + (size_t)j * 32);
          float rot = q_angle - (ang_f + k0)[sl];
          if (rot < 0.0f) rot += 360.0f;
          int bin = (int)roundf(rot * factor);
          if (bin == kHistoLen) bin = 0;
          pool[wpos + __popc(gm & ((1u << gl) - 1u))] =
              (unsigned)j | ((unsigned)d << 13) | ((unsigned)(oct & 15) << 22) | ((unsigned)(bin & 31) << 26);
        }
        wpos += tg[it];
      }
      if (gl == 0 && qin) {
        int2* out = A.qrec + (size_t)f * A.q_cap + qg;
        if (!valid || empty)
          *out = make_int2(0, 0);
        else if (small)
          *out = tot_g == 0 ? make_int2(0, 0) : (over ? make_int2(0, -1) : make_int2(base + off_g, tot_g | ((flags & 2) ? 1 << 16 : 0)));
        else {  // a wide or crowded window: pass 2
          const int k = atomicAdd(&s_nbig, 1);
          if (k < kBigCap) s_big[k] = qg;
        }
      }
    }
  }
  __syncthreads();
  // ---- pass 2: a wavefront per query for the windows pass 1 left (more than 16 columns or 64 entries)
  // (the block's own list of them; if it overflowed, every query of the stride is looked at again and the small ones skipped)
  // The fallback walks exactly the queries pass 1 gave this wavefront (groups of four, kSbpBlocks * 16 apart): pass 1
  // wrote no record for their wide ones and no other block looks at them.
  const int nbig = s_nbig;
  const bool listed = nbig <= kBigCap;
  const int own0 = (blockIdx.x * 4 + wave) * 4;
  for (int k2 = listed ? wave : 0;; k2 += listed ? 4 : 1) {
    int q;
    if (listed) {
      if (k2 >= nbig) break;
      q = s_big[k2];
    } else {
      const int q0 = own0 + (k2 >> 2) * (n_blocks * 16);
      if (q0 >= nq) break;
      q = q0 + (k2 & 3);
      if (q >= nq) continue;
    }
    const uint4 h0 = QQ[4 * (size_t)q], h1 = QQ[4 * (size_t)q + 1];
    const uint4 d0 = QQ[4 * (size_t)q + 2], d1 = QQ[4 * (size_t)q + 3];
    const float x = __uint_as_float(h0.x), y = __uint_as_float(h0.y), q_ur = __uint_as_float(h0.z);
    const float r = __uint_as_float(h0.w);
    const int minlevel = (int)h1.x, maxlevel = (int)h1.y, flags = (int)h1.w;
    const float q_angle = __uint_as_float(h1.z);
    int2* out = A.qrec + (size_t)f * A.q_cap + q;
    const int cam = (flags >> 8) & 15;
    if (!(flags & 1) || cam >= A.n_cams) continue;  // pass 1 wrote its empty record
    // the query's camera: bounds, its slice of the records (camera c's records start at its first key)
    const float minx = A.bounds[cam][0], miny = A.bounds[cam][2];
    const float winv = (float)kGridCols / (A.bounds[cam][1] - minx), hinv = (float)kGridRows / (A.bounds[cam][3] - miny);
    int k0 = 0;
    if (A.cam_first) k0 = min(A.cam_first[(size_t)f * (A.n_cams + 1) + cam], A.key_cap);
    const float4* rec = rec_f + k0;
    const float* ang = ang_f + k0;
    const unsigned short* cs = s_cs + cam * (kGridCells + 1);
    // FrameBase.cpp:102-115
    const int min_cellx = max(0, (int)floorf((x - minx - r) * winv));
    const int max_cellx = min(kGridCols - 1, (int)ceilf((x - minx + r) * winv));
    const int min_celly = max(0, (int)floorf((y - miny - r) * hinv));
    const int max_celly = min(kGridRows - 1, (int)ceilf((y - miny + r) * hinv));
    if (min_cellx >= kGridCols || max_cellx < 0 || min_celly >= kGridRows || max_celly < 0 ||
        min_cellx > max_cellx || min_celly > max_celly)
      continue;  // (pass 1 wrote its empty record)
    const bool bchecklevel = (minlevel > 0) || (maxlevel >= 0);
    const int nx = max_cellx - min_cellx + 1;
    int seg_s = 0, seg_l = 0;
    if (lane < nx) {
      const int col = (min_cellx + lane) * kGridRows;
      seg_s = cs[col + min_celly];
      seg_l = cs[col + max_celly + 1] - seg_s;
    }
    int n_ent;
    const int seg_o = wave_excl_scan_i(seg_l, lane, &n_ent);  // first window entry of the run
    if (nx <= 16 && n_ent <= 64) continue;  // pass 1 did it
    // entry t of the window -> record, tests.  A key inside the window is a candidate unless its
    // stereo coordinate disagrees with the query's (ORBmatcher.cc:1421-1426 / :278-283) -- the one test of
    // the inner loop that does not depend on what earlier queries claimed, so it is applied here.
    auto test = [&](int t, int* slot, int* packed) -> bool {
      int sl = -1;
      for (int i = 0; i < nx; i++) {
        const int o = __builtin_amdgcn_readlane(seg_o, i), l = __builtin_amdgcn_readlane(seg_l, i), s0 = __builtin_amdgcn_readlane(seg_s, i);
        if (t >= o && t < o + l) sl = s0 + (t - o);
      }
      *slot = sl;
      if (t >= n_ent || sl < 0) return false;
      const float4 k = rec[sl];
      const int pk = __float_as_int(k.w), oct = pk >> 16;
      *packed = pk;
      if (bchecklevel) {
        if (oct < minlevel) return false;
        if (maxlevel >= 0 && oct > maxlevel) return false;
      }
      if (!(fabsf(k.x - x) < r && fabsf(k.y - y) < r)) return false;
      if (A.mode != VIEO_SBP_RELOC && k.z > 0 && fabsf(q_ur - k.z) > r) return false;
      return true;
    };
    auto encode = [&](int slot, int packed) -> unsigned {
      const int j = packed & 0xFFFF, oct = packed >> 16;
      const int d = hamming32q(d0, d1, D + (size_t)j * 32);
      float rot = q_angle - ang[slot];  // ORBmatcher.cc:1453-1460, used only with mbCheckOrientation
      if (rot < 0.0f) rot += 360.0f;
      int bin = (int)roundf(rot * factor);
      if (bin == kHistoLen) bin = 0;
      return (unsigned)j | ((unsigned)d << 13) | ((unsigned)(oct & 15) << 22) | ((unsigned)(bin & 31) << 26);
    };
    const int obs_bit = (flags & 2) ? 1 << 16 : 0;
    if (n_ent <= 64) {  // the usual case: one evaluation
      int slot, packed = 0;
      const bool pass = test(lane, &slot, &packed);
      const unsigned long long m = __ballot(pass);
      const int total = __popcll(m);
      if (total == 0) {
        if (lane == 0) *out = make_int2(0, 0);
        continue;
      }
      const int base = reserve(total);
      if (base + total > A.pool_cap) {
        if (lane == 0) *out = make_int2(0, -1);
        continue;
      }
      if (pass) pool[base + __popcll(m & ((1ull << lane) - 1ull))] = encode(slot, packed);
      if (lane == 0) *out = make_int2(base, total | obs_bit);
      continue;
    }
    int total = 0;
    for (int t0 = 0; t0 < n_ent; t0 += 64) {
      int slot, packed = 0;
      total += __popcll(__ballot(test(t0 + lane, &slot, &packed)));
    }
    if (total == 0 || total > kCandCap) {
      if (lane == 0) *out = make_int2(0, total == 0 ? 0 : -1);
      continue;
    }
    const int base = reserve(total);
    if (base + total > A.pool_cap) {
      if (lane == 0) *out = make_int2(0, -1);
      continue;
    }
    int w = base;
    for (int t0 = 0; t0 < n_ent; t0 += 64) {
      int slot, packed = 0;
      const bool pass = test(t0 + lane, &slot, &packed);
      const unsigned long long m = __ballot(pass);
      if (pass) pool[w + __popcll(m & ((1ull << lane) - 1ull))] = encode(slot, packed);
      w += __popcll(m);
    }
    if (lane == 0) *out = make_int2(base, total | obs_bit);
  }
}

// ---- the assignment, optimistic-parallel (round 3).  The reference walks the queries in order and a key that an
// earlier query's map point took (AddMapPoint, Observations() > 0) is skipped by the later ones (ORBmatcher.cc:289-291,
// 1430-1433), so query q's outcome depends on the queries before it -- but only through the few keys they claim.
// Fixed-point form: in every round ALL queries pick their best / second candidate in parallel, treating key k as taken
// iff the PREVIOUS round's claims hold an earlier blocking query on it (min over the claimers, one atomicMin per claim).
// By induction query j is right from round j + 1 on, and a round that changes no claim is the sequential answer (every
// query then agrees with the claims of the queries before it); real frames settle in a handful of rounds because the
// dependency chains are as short as the overlaps of neighbouring search windows.  One workgroup of 1024 threads per
// frame, one thread per query; the state lives in LDS.  A frame that has not settled after max_rounds is left to the
// sequential replay below (need_seq).  Same results bit for bit: the candidate order key (distance, position) and the
// tests are the replay's.
__global__ void __launch_bounds__(1024) k_sbp_assign_par(SbpArgs A) {
  extern __shared__ unsigned s_pool[];
  int* s_min0 = (int*)(s_pool + A.pool_lds);  // [key_cap] earliest blocking claimer of the key, previous / current round
  int* s_min1 = s_min0 + A.key_cap;
  unsigned* s_bins = (unsigned*)(s_min1 + A.key_cap);
  uint8_t* s_taken = (uint8_t*)(s_bins + A.key_cap);
  __shared__ int s_hist[kHistoLen];
  __shared__ int s_changed[2], s_nm, s_overflow;  // s_changed: one flag per round parity
  const int f = blockIdx.x, tid = threadIdx.x;
  int N;
  {
    int k0;
    cam_range(A, f, A.n_cams - 1, &k0, &N);
  }
  const int nq = min(A.nq[f], A.q_cap);
  int* assign = A.assign + (size_t)f * A.key_cap;
  const uint8_t* taken = A.taken ? A.taken + (size_t)f * A.key_cap : nullptr;
  const unsigned* pool = A.pool + (size_t)f * A.pool_cap;
  const int2* qrec = A.qrec + (size_t)f * A.q_cap;
  int* claim = A.claim + (size_t)f * A.q_cap;
  const int n_lds = min(min(A.cursor[f], A.pool_cap), A.pool_lds);
  for (int i = tid; i < n_lds; i += 1024) s_pool[i] = pool[i];
  for (int i = tid; i < N; i += 1024) {
    s_taken[i] = taken ? taken[i] : (uint8_t)0;
    s_min0[i] = INT_MAX, s_min1[i] = INT_MAX, s_bins[i] = 0u;
    assign[i] = VIEO_SBP_UNCHANGED;
  }
  for (int q = tid; q < nq; q += 1024) claim[q] = -1;
  if (tid < kHistoLen) s_hist[tid] = 0;
  if (tid == 0) s_nm = 0, s_overflow = 0, s_changed[0] = s_changed[1] = 0;
  __syncthreads();
  const bool reloc = A.mode == VIEO_SBP_RELOC;
  // best / second of query q when key k is blocked iff it was taken at the start or mn[k] < q; returns the claimed
  // key (-1: no match) and the winner's candidate word
  auto eval = [&](int q, const int* mn, unsigned* word, int* blocker) -> int {
    const int2 r = qrec[q];
    const int off = r.x, ny = r.y;
    if (ny == 0) return -1;
    if (ny < 0) {
      s_overflow = 1;
      return -1;
    }
    const int n = ny & 0xFFFF;
    unsigned b0 = 0xFFFFFFFFu, b1 = 0xFFFFFFFFu, c0 = 0, c1 = 0;
    for (int pos = 0; pos < n; pos++) {
      const unsigned c = off + pos < n_lds ? s_pool[off + pos] : pool[off + pos];
      const int idx = c & 0x1FFF;
      if (s_taken[idx] || mn[idx] < q) continue;
      const unsigned key = (((c >> 13) & 0x1FFu) << 8) | (unsigned)pos;  // (dist, order)
      if (key < b0)
        b1 = b0, c1 = c0, b0 = key, c0 = c;
      else if (key < b1)
        b1 = key, c1 = c;
    }
    if (b0 == 0xFFFFFFFFu) return -1;
    const int bestDist = b0 >> 8, bestLevel = (c0 >> 22) & 15;
    if (bestDist > (reloc ? (int)A.nn_ratio : kThHigh)) return -1;
    if (A.mode == VIEO_SBP_LOCAL_MAP && b1 != 0xFFFFFFFFu) {
      const int bestDist2 = b1 >> 8, bestLevel2 = (c1 >> 22) & 15;
      if (bestLevel == bestLevel2 && (float)bestDist > A.nn_ratio * (float)bestDist2) return -1;
    }
    *word = c0;
    *blocker = reloc || (ny >> 16);  // its key is closed to later queries (any holder in the relocalisation mode)
    return (int)(c0 & 0x1FFF);
  };
  bool settled = false;
  for (int round = 0; round < A.max_rounds; round++) {
    int* prev = (round & 1) ? s_min1 : s_min0;
    int* cur = (round & 1) ? s_min0 : s_min1;
    int changed = 0;
    for (int q = tid; q < nq; q += 1024) {
      unsigned w;
      int blk = 0;
      const int k = eval(q, prev, &w, &blk);
      if (k != claim[q]) claim[q] = k, changed = 1;
      if (k >= 0 && blk) atomicMin(&cur[k], q);
    }
    if (changed) s_changed[round & 1] = 1;
    __syncthreads();
    const int any = s_changed[round & 1];
    if (tid == 0) s_changed[(round + 1) & 1] = 0;  // (last read a round ago)
    for (int i = tid; i < N; i += 1024) prev[i] = INT_MAX;  // becomes the next round's `cur`
    __syncthreads();
    if (!any) {  // `cur` == the claims everybody just agreed with
      settled = true;
      // results: AddMapPoint in query order = the last accepted claimer of a key stays
      const bool ori = A.mode != VIEO_SBP_LOCAL_MAP && A.check_ori;
      int* s_asg = prev;  // (free now) -1 == VIEO_SBP_UNCHANGED after the reset below
      for (int i = tid; i < N; i += 1024) s_asg[i] = -1;
      __syncthreads();
      int nm = 0;
      for (int q = tid; q < nq; q += 1024) {
        const int k = claim[q];
        if (k < 0) continue;
        unsigned w = 0;
        int blk = 0;
        (void)eval(q, cur, &w, &blk);  // the winner's word again (rotation bin)
        atomicMax(&s_asg[k], q);
        nm++;
        if (ori) {
          const int bin = (w >> 26) & 31;
          atomicOr(&s_bins[k], 1u << bin);
          atomicAdd(&s_hist[bin], 1);
        }
      }
      if (nm) atomicAdd(&s_nm, nm);
      __syncthreads();
      int nmatches = s_nm;
      unsigned losers = 0;
      if (ori) {  // ComputeThreeMaxima (ORBmatcher.cc:1608-1641), evaluated redundantly by every thread
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < kHistoLen; i++) {
          const int sv = s_hist[i];
          if (sv > max1) {
            max3 = max2, max2 = max1, max1 = sv;
            ind3 = ind2, ind2 = ind1, ind1 = i;
          } else if (sv > max2) {
            max3 = max2, max2 = sv;
            ind3 = ind2, ind2 = i;
          } else if (sv > max3) {
            max3 = sv, ind3 = i;
          }
        }
        if (max2 < 0.1f * (float)max1) {
          ind2 = -1, ind3 = -1;
        } else if (max3 < 0.1f * (float)max1) {
          ind3 = -1;
        }
        losers = (1u << kHistoLen) - 1u;
        if (ind1 >= 0) losers &= ~(1u << ind1);
        if (ind2 >= 0) losers &= ~(1u << ind2);
        if (ind3 >= 0) losers &= ~(1u << ind3);
        for (int i = 0; i < kHistoLen; i++)
          if (i != ind1 && i != ind2 && i != ind3) nmatches -= s_hist[i];
      }
      for (int k = tid; k < N; k += 1024) assign[k] = (s_bins[k] & losers) ? VIEO_SBP_ERASED : s_asg[k];
      if (tid == 0) A.nmatches[f] = s_overflow ? -1 : nmatches;
      break;
    }
  }
  if (tid == 0) A.need_seq[f] = settled ? 0 : 1;
}

// (one wavefront: the body is also the tail of k_sbp_assign_cam's last workgroup when a frame did not settle)
__device__ __forceinline__ void seq_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
__device__ void sbp_assign_seq(const SbpArgs& A, int f, unsigned* s_pool, int lane) {
  // LDS from s_pool on: pool copy (pool_lds words) | per key the rotation bins it was accepted with (bit b = bin b; a key
  // can be accepted more than once when its holder has no observations) | key state (key_cap bytes)
  unsigned* s_bins = s_pool + A.pool_lds;
  uint8_t* s_state = (uint8_t*)(s_bins + A.key_cap);  // bit0: holds a map point, bit1: it has Observations()>0
  __shared__ int s_hist[kHistoLen];
  int N;
  {
    int k0;
    cam_range(A, f, A.n_cams - 1, &k0, &N);  // end of the last camera = number of keys of the frame
  }
  const int nq = min(A.nq[f], A.q_cap);
  int* assign = A.assign + (size_t)f * A.key_cap;
  const uint8_t* taken = A.taken ? A.taken + (size_t)f * A.key_cap : nullptr;
  const unsigned* pool = A.pool + (size_t)f * A.pool_cap;
  const int2* qrec = A.qrec + (size_t)f * A.q_cap;
  const int n_lds = min(min(A.cursor[f], A.pool_cap), A.pool_lds);
  for (int i0 = lane; i0 < n_lds; i0 += 16 * 64) {  // 16 loads in flight per lane, then the LDS stores
    unsigned v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = i0 + 64 * u < n_lds ? pool[i0 + 64 * u] : 0u;
#pragma unroll
    for (int u = 0; u < 16; u++)
      if (i0 + 64 * u < n_lds) s_pool[i0 + 64 * u] = v[u];
  }
  for (int i0 = lane; i0 < N; i0 += 8 * 64) {
    uint8_t t[8];
#pragma unroll
    for (int u = 0; u < 8; u++) t[u] = (taken && i0 + 64 * u < N) ? taken[i0 + 64 * u] : (uint8_t)0;
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (i0 + 64 * u < N) {
        assign[i0 + 64 * u] = VIEO_SBP_UNCHANGED;
        s_state[i0 + 64 * u] = t[u] ? 3 : 0;
        s_bins[i0 + 64 * u] = 0u;
      }
  }
  if (lane < kHistoLen) s_hist[lane] = 0;
  seq_sync();
  int nmatches = 0, overflow = 0;
  const bool ori = A.mode != VIEO_SBP_LOCAL_MAP && A.check_ori;
  for (int q0 = 0; q0 < nq; q0 += 64) {
    const int2 mine = q0 + lane < nq ? qrec[q0 + lane] : make_int2(0, 0);
    const int qn = min(64, nq - q0);
    for (int qq = 0; qq < qn; qq++) {
      const int off = __builtin_amdgcn_readlane(mine.x, qq), ny = __builtin_amdgcn_readlane(mine.y, qq);  // qq is uniform
      if (ny == 0) continue;
      if (ny < 0) {
        overflow = 1;
        continue;
      }
      const int n = ny & 0xFFFF, q = q0 + qq;
      // two smallest (dist, order) among usable candidates; lane owns positions lane, lane+64.  The keys are unique
      // (they carry the position), so the second smallest is the minimum once the winner's lane puts its other key
      // forward; the winners' candidate words are read back from their lanes' registers.
      unsigned b0 = 0xFFFFFFFFu, b1 = 0xFFFFFFFFu;
      unsigned cand2[2] = {0u, 0u};  // the lane's candidate words: the winners are read back from here
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int pos = lane + 64 * h;
        if (pos < n) {
          const unsigned c = off + pos < n_lds ? s_pool[off + pos] : pool[off + pos];
          cand2[h] = c;
          const int idx = c & 0x1FFF, d = (c >> 13) & 0x1FF;
          const int st = s_state[idx];
          if (A.mode == VIEO_SBP_RELOC ? !(st & 1) : !((st & 1) && (st & 2))) {
            const unsigned key = ((unsigned)d << 8) | (unsigned)pos;  // (dist, order)
            if (key < b0)
              b1 = b0, b0 = key;
            else
              b1 = key;
          }
        }
      }
      const unsigned m0 = wave_min_u32(b0);
      const unsigned m1 = wave_min_u32(b0 == m0 ? b1 : b0);
      b0 = m0, b1 = m1;
      // (uniform position -> that lane's register: v_readlane instead of another LDS round trip)
      unsigned c0 = 0, c1 = 0;
      if (b0 != 0xFFFFFFFFu) c0 = (unsigned)__builtin_amdgcn_readlane((int)((b0 & 64u) ? cand2[1] : cand2[0]), (int)(b0 & 63u));
      if (b1 != 0xFFFFFFFFu) c1 = (unsigned)__builtin_amdgcn_readlane((int)((b1 & 64u) ? cand2[1] : cand2[0]), (int)(b1 & 63u));
      if (b0 == 0xFFFFFFFFu) continue;
      const int bestDist = b0 >> 8;
      const int bestIdx = c0 & 0x1FFF, bestLevel = (c0 >> 22) & 15;
      if (bestDist > (A.mode == VIEO_SBP_RELOC ? (int)A.nn_ratio : kThHigh)) continue;
      if (A.mode == VIEO_SBP_LOCAL_MAP && b1 != 0xFFFFFFFFu) {
        const int bestDist2 = b1 >> 8;
        const int bestLevel2 = (c1 >> 22) & 15;
        if (bestLevel == bestLevel2 && (float)bestDist > A.nn_ratio * (float)bestDist2) continue;
      }
      // AddMapPoint(pMP, bestIdx)
      if (lane == 0) {
        s_state[bestIdx] = 1 | ((ny >> 16) ? 2 : 0);
        assign[bestIdx] = q;
      }
      nmatches++;
      if (ori) {
        const int bin = (c0 >> 26) & 31;
        if (lane == 0) {  // rotHist[bin].push_back(bestIdx2)
          s_bins[bestIdx] |= 1u << bin;
          s_hist[bin]++;
        }
      }
      seq_sync();  // single wave: orders the LDS state update before the next query
    }
  }
  if (ori) {
    seq_sync();
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
    // EraseMapPointMatch of every key logged in a bin that is not kept
    unsigned losers = (1u << kHistoLen) - 1u;
    if (ind1 >= 0) losers &= ~(1u << ind1);
    if (ind2 >= 0) losers &= ~(1u << ind2);
    if (ind3 >= 0) losers &= ~(1u << ind3);
    for (int k = lane; k < N; k += 64)
      if (s_bins[k] & losers) assign[k] = VIEO_SBP_ERASED;
    for (int i = 0; i < kHistoLen; i++)
      if (i != ind1 && i != ind2 && i != ind3) nmatches -= s_hist[i];
  }
  if (lane == 0) A.nmatches[f] = overflow ? -1 : nmatches;
}

// ---- the same assignment for camera rigs, one workgroup per (frame, camera) (round 4).  A query belongs to one
// camera and its candidates are keys of that camera, so the order-dependent walk decomposes exactly: the queries of
// camera c, in their original order, against the keys of camera c.  Two things made the one-workgroup form slow on rig
// frames (31 us per round of 10 k queries, 8 rounds): (1) 13 bytes of state per key of ALL cameras in LDS left room for
// 5 k of the frame's 35-45 k candidate words, the rest was read through L2 in every round; (2) one thread per query
// walks its list candidate by candidate -- three dependent reads each -- and a wavefront waits for its longest list
// (up to kCandCap = 128 entries where the texture is dense, 3 on average).  Here the camera's lists are copied to LDS
// once and a round is FLAT over the (query, candidate) pairs: every thread takes the same number of pairs, tests
// "blocked?" and puts the pair's key (distance, position) into its query's slot with one LDS atomicMin (a second pass
// for the runner-up where the mode looks at it); the query's owner then applies the tests and claims.  The cameras run
// side by side on their own CUs.  What spans the cameras -- the rotation histogram (ComputeThreeMaxima over the matches
// of all cameras, ORBmatcher.cc:1608-1641), the match count, "not settled" -- is finished by the LAST workgroup of the
// frame to arrive (agent-scope release / acquire around one counter).  fin: [frame][48] ints, zero between launches
// (the finisher clears it); kbins: [frame][key_cap] rotation bins of the accepted keys.
// LDS: pairs (word u32, owner u16) x pair_cap | per key 13 bytes x cam_cap | per query of the camera 12 bytes x kCamQ.
static const int kFinStride = 48, kCamQ = 4096;
__global__ void __launch_bounds__(1024) k_sbp_assign_cam(SbpArgs A, int cam_cap, int* __restrict__ fin, unsigned* __restrict__ kbins,
                                                        int lds_bytes) {
  extern __shared__ unsigned s_pool[];             // [pool_lds] the camera's candidate words, list after list
  unsigned* s_best = s_pool + A.pool_lds;          // [kCamQ] smallest unblocked key of the query this round
  unsigned* s_second = s_best + kCamQ;             // [kCamQ] runner-up (local-map mode)
  int* s_min0 = (int*)(s_second + kCamQ);          // [cam_cap] earliest blocking claimer of the key, previous / current round
  int* s_min1 = s_min0 + cam_cap;
  unsigned* s_bins = (unsigned*)(s_min1 + cam_cap);
  unsigned short* s_qof = (unsigned short*)(s_bins + cam_cap);  // [pool_lds] pair -> local query
  unsigned short* s_qid = s_qof + A.pool_lds;      // [kCamQ] local query -> query
  unsigned short* s_qstart = s_qid + kCamQ;        // [kCamQ] its first pair
  uint8_t* s_taken = (uint8_t*)(s_qstart + kCamQ);  // [cam_cap]
  __shared__ int s_hist[kHistoLen];
  __shared__ int s_changed[2], s_nm, s_overflow, s_fill, s_nql, s_last;
  const int f = blockIdx.x, cam = blockIdx.y, tid = threadIdx.x;
  int k0, k1;
  cam_range(A, f, cam, &k0, &k1);
  const int Nc = k1 - k0;
  const int nq = min(A.nq[f], A.q_cap);
  int* assign = A.assign + (size_t)f * A.key_cap;
  const uint8_t* taken = A.taken ? A.taken + (size_t)f * A.key_cap : nullptr;
  const unsigned* pool = A.pool + (size_t)f * A.pool_cap;
  const int2* qrec = A.qrec + (size_t)f * A.q_cap;
  int* F = fin + (size_t)f * kFinStride;  // [0, 30) histogram, [32] matches, [33] overflow, [34] not settled, [35] arrivals
  unsigned* KB = kbins + (size_t)f * A.key_cap;
  const bool ori = A.mode != VIEO_SBP_LOCAL_MAP && A.check_ori;
  const bool reloc = A.mode == VIEO_SBP_RELOC, second = A.mode == VIEO_SBP_LOCAL_MAP;
  // A thread OWNS the camera's queries me = tid, tid + 1024, ... (kCamQ / 1024 = 4 slots: their state lives in registers);
  // which query sits in which slot is decided while the lists are gathered and does not matter -- the order of the walk is
  // the query number s_qid[me].  (Round 4 kept a register slot for EVERY query of the frame, 12 per thread: a 4-camera
  // fisheye sequence reaches 14 k valid (point, camera) queries once its map has grown, and every frame beyond 12 288 fell
  // to the sequential replay, 6 ms instead of 50 us.)
  constexpr int kOwn = kCamQ / 1024;
  bool settled = false;
  // (a camera with more keys than its share of key_cap, more queries than fit: the sequential replay takes the frame)
  if (Nc <= cam_cap && nq < 65536) {
    for (int i = tid; i < Nc; i += 1024) {
      s_taken[i] = taken ? taken[k0 + i] : (uint8_t)0;
      s_min0[i] = INT_MAX, s_min1[i] = INT_MAX, s_bins[i] = 0u;
    }
    for (int i = tid; i < kCamQ; i += 1024) s_best[i] = 0xFFFFFFFFu, s_second[i] = 0xFFFFFFFFu;
    if (tid < kHistoLen) s_hist[tid] = 0;
    if (tid == 0) s_nm = 0, s_overflow = 0, s_fill = 0, s_nql = 0, s_changed[0] = s_changed[1] = 0;
    __syncthreads();
    // ---- all queries of the frame, four records in flight per thread: which are this camera's (their first candidate is
    // one of its keys); those get a slot, their lists go to LDS (s_second keeps the record's count word until the owner
    // has taken it)
    for (int q0 = tid; q0 < nq; q0 += 4096) {
      int2 rq[4];
      unsigned first[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int q = q0 + 1024 * j;
        rq[j] = q < nq ? qrec[q] : make_int2(0, 0);
      }
#pragma unroll
      for (int j = 0; j < 4; j++) first[j] = rq[j].y > 0 ? pool[rq[j].x] : 0u;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (rq[j].y < 0) s_overflow = 1;  // (its camera is unknown: every camera reports it)
        const int n = rq[j].y > 0 ? (rq[j].y & 0xFFFF) : 0, idx = (int)(first[j] & 0x1FFF);
        if (n > 0 && idx >= k0 && idx < k1) {
          const int lo = atomicAdd(&s_fill, n), me = atomicAdd(&s_nql, 1);
          if (lo + n <= A.pool_lds && me < kCamQ) {
            s_qid[me] = (unsigned short)(q0 + 1024 * j), s_qstart[me] = (unsigned short)lo;
            s_second[me] = (unsigned)rq[j].y;
            for (int p0 = 0; p0 < n; p0 += 8) {  // eight words in flight
              unsigned cw[8];
#pragma unroll
              for (int u = 0; u < 8; u++) cw[u] = pool[rq[j].x + min(p0 + u, n - 1)];
#pragma unroll
              for (int u = 0; u < 8; u++)
                if (p0 + u < n) s_pool[lo + p0 + u] = cw[u], s_qof[lo + p0 + u] = (unsigned short)me;
            }
          }
        }
      }
    }
    __syncthreads();
    // the owners take their queries' records
    int lq[kOwn], cny[kOwn], cl[kOwn];
    unsigned wl[kOwn];  // the winner's candidate word at the last round (rotation bin)
    {
      const int nql = min(s_nql, kCamQ);
#pragma unroll
      for (int j = 0; j < kOwn; j++) {
        const int me = tid + 1024 * j;
        cl[j] = -1, wl[j] = 0, lq[j] = me < nql ? me : -1, cny[j] = 0;
        if (me < nql) cny[j] = (int)s_second[me], s_second[me] = 0xFFFFFFFFu;
      }
    }
    __syncthreads();
    const int n_pairs = s_fill;
    const bool fits = n_pairs <= A.pool_lds && s_nql <= kCamQ && n_pairs < 65536;
    for (int round = 0; fits && round < A.max_rounds; round++) {
      int* prev = (round & 1) ? s_min1 : s_min0;
      int* cur = (round & 1) ? s_min0 : s_min1;
      // pairs: the key (distance, position) of every unblocked candidate into its query's slot
      for (int p = tid; p < n_pairs; p += 1024) {
        const unsigned c = s_pool[p];
        const int me = s_qof[p], idx = (int)(c & 0x1FFF) - k0;
        const int q = s_qid[me], pos = p - s_qstart[me];
        if (!(s_taken[idx] || prev[idx] < q)) atomicMin(&s_best[me], (((c >> 13) & 0x1FFu) << 8) | (unsigned)pos);
      }
      __syncthreads();
      if (second) {  // the runner-up: the smallest key that is not the best
        for (int p = tid; p < n_pairs; p += 1024) {
          const unsigned c = s_pool[p];
          const int me = s_qof[p], idx = (int)(c & 0x1FFF) - k0;
          const int q = s_qid[me], pos = p - s_qstart[me];
          const unsigned key = (((c >> 13) & 0x1FFu) << 8) | (unsigned)pos;
          if (key != s_best[me] && !(s_taken[idx] || prev[idx] < q)) atomicMin(&s_second[me], key);
        }
        __syncthreads();
      }
      // queries: the tests behind best / second, the claim
      int changed = 0;
#pragma unroll
      for (int j = 0; j < kOwn; j++)
        if (lq[j] >= 0) {
          const int me = lq[j], q = s_qid[me], ny = cny[j];
          const unsigned b0 = s_best[me], b1 = s_second[me];
          s_best[me] = 0xFFFFFFFFu, s_second[me] = 0xFFFFFFFFu;  // (for the next round)
          int k = -1;
          if (b0 != 0xFFFFFFFFu) {
            const unsigned c0 = s_pool[s_qstart[me] + (b0 & 0xFF)];
            const int bestDist = b0 >> 8, bestLevel = (c0 >> 22) & 15;
            bool ok = bestDist <= (reloc ? (int)A.nn_ratio : kThHigh);
            if (ok && second && b1 != 0xFFFFFFFFu) {
              const unsigned c1 = s_pool[s_qstart[me] + (b1 & 0xFF)];
              const int bestDist2 = b1 >> 8, bestLevel2 = (c1 >> 22) & 15;
              if (bestLevel == bestLevel2 && (float)bestDist > A.nn_ratio * (float)bestDist2) ok = false;
            }
            if (ok) k = (int)(c0 & 0x1FFF) - k0, wl[j] = c0;
          }
          if (k != cl[j]) cl[j] = k, changed = 1;
          // its key is closed to later queries when its holder has observations (any holder in the relocalisation mode)
          if (k >= 0 && (reloc || (ny >> 16))) atomicMin(&cur[k], q);
        }
      if (changed) s_changed[round & 1] = 1;
      __syncthreads();
      const int any = s_changed[round & 1];
      if (tid == 0) s_changed[(round + 1) & 1] = 0;  // (last read a round ago)
      for (int i = tid; i < Nc; i += 1024) prev[i] = INT_MAX;  // becomes the next round's `cur`
      __syncthreads();
      if (!any) {  // `cur` == the claims everybody just agreed with
        settled = true;
        int* s_asg = prev;  // (free now) -1 == VIEO_SBP_UNCHANGED
        for (int i = tid; i < Nc; i += 1024) s_asg[i] = -1;
        __syncthreads();
        int nm = 0;
#pragma unroll
        for (int j = 0; j < kOwn; j++)
          if (lq[j] >= 0 && cl[j] >= 0) {
            atomicMax(&s_asg[cl[j]], (int)s_qid[lq[j]]);  // AddMapPoint in query order = the last accepted claimer of a key stays
            nm++;
            if (ori) {
              const int bin = (wl[j] >> 26) & 31;
              atomicOr(&s_bins[cl[j]], 1u << bin);
              atomicAdd(&s_hist[bin], 1);
            }
          }
        if (nm) atomicAdd(&s_nm, nm);
        __syncthreads();
        for (int i = tid; i < Nc; i += 1024) assign[k0 + i] = s_asg[i], KB[k0 + i] = s_bins[i];
        if (ori && tid < kHistoLen && s_hist[tid]) atomicAdd(&F[tid], s_hist[tid]);
        if (tid == 0 && s_nm) atomicAdd(&F[32], s_nm);
        break;
      }
    }
    if (tid == 0 && s_overflow == 1) atomicOr(&F[33], 1);
  }
  if (tid == 0 && !settled) atomicOr(&F[34], 1);
  // ---- the frame's last workgroup finishes what spans the cameras
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (A.n_cams > 1) {  // (one camera: this workgroup is the frame's only one -- its own stores are visible to it)
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int old = __hip_atomic_fetch_add(&F[35], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = old == A.n_cams - 1;
      if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!s_last) return;
  }
  if (tid < kHistoLen) s_hist[tid] = __hip_atomic_load(&F[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0) {
    s_nm = __hip_atomic_load(&F[32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_overflow = __hip_atomic_load(&F[33], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_fill = __hip_atomic_load(&F[34], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  int nmatches = s_nm;
  if (ori && !s_fill) {  // ComputeThreeMaxima (ORBmatcher.cc:1608-1641), evaluated redundantly by every thread
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < kHistoLen; i++) {
      const int sv = s_hist[i];
      if (sv > max1) {
        max3 = max2, max2 = max1, max1 = sv;
        ind3 = ind2, ind2 = ind1, ind1 = i;
      } else if (sv > max2) {
        max3 = max2, max2 = sv;
        ind3 = ind2, ind2 = i;
      } else if (sv > max3) {
        max3 = sv, ind3 = i;
      }
    }
    if (max2 < 0.1f * (float)max1) {
      ind2 = -1, ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
      ind3 = -1;
    }
    unsigned losers = (1u << kHistoLen) - 1u;
    if (ind1 >= 0) losers &= ~(1u << ind1);
    if (ind2 >= 0) losers &= ~(1u << ind2);
    if (ind3 >= 0) losers &= ~(1u << ind3);
    for (int i = 0; i < kHistoLen; i++)
      if (i != ind1 && i != ind2 && i != ind3) nmatches -= s_hist[i];
    int kend;
    {
      int kb;
      cam_range(A, f, A.n_cams - 1, &kb, &kend);
    }
    for (int k = tid; k < kend; k += 1024)
      if (KB[k] & losers) assign[k] = VIEO_SBP_ERASED;
  }
  const bool replay = s_fill != 0;  // some camera did not settle (or does not fit): the sequential replay, right here
  if (tid == 0) {
    A.nmatches[f] = s_overflow ? -1 : nmatches;
    A.need_seq[f] = 0;
  }
  if (tid < kFinStride) F[tid] = 0;  // (the next launch on this stream finds it clear)
  __syncthreads();
  if (replay && tid < 64) {
    SbpArgs S = A;
    S.pool_lds = max(0, min(A.pool_cap, (lds_bytes - A.key_cap * 5 - 64) / 4));
    sbp_assign_seq(S, f, s_pool, tid);
  }
}

// one wave per frame: the frame's candidate pool is copied to LDS once, after that the replay of
// the queries touches no global memory except the accepted assignments.  Since round 3 the fallback of
// k_sbp_assign_par (frames whose claims did not settle) and the reference form for the tests (VIEO_SBP_ASSIGN=seq).
__global__ void __launch_bounds__(64) k_sbp_assign(SbpArgs A) {
  extern __shared__ unsigned s_pool[];
  if (A.need_seq && !A.need_seq[blockIdx.x]) return;  // the parallel assignment settled this frame
  sbp_assign_seq(A, blockIdx.x, s_pool, threadIdx.x);
}

// ---------------------------------------------------------------- SearchByProjectionBase (Fuse) -------------
// ORBmatcher.cc:26-193 per (map point, camera): one wavefront per point; the projection and its gates are
// evaluated by every lane (cheap, uniform), the window is walked column by column over the CSR grid with the
// lanes across a column's run, best (distance, position in the reference's candidate order) reduced over the
// wavefront.  The mutations that follow in the reference (FuseMP, only-one-match) are the caller's.
struct FuseDev {
  vieo_fuse_frame F;
  CamD cams[4];
};

__global__ void __launch_bounds__(256)
k_fuse_search(const FuseDev* __restrict__ fd, int cami, const int* __restrict__ cell_start,
              const float4* __restrict__ cell_rec, const uint8_t* __restrict__ desc,
              const vieo_fuse_point* __restrict__ pts, int n, int32_t* __restrict__ best_idx,
              int32_t* __restrict__ best_dist) {
  const int m = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // wave-uniform
  if (m >= n) return;
  const vieo_fuse_frame& FF = fd->F;
  const vieo_frustum_frame& F = FF.base;
  const int nc = F.n_cams;
  int32_t* oi = best_idx + (size_t)m * nc + cami;
  int32_t* od = best_dist + (size_t)m * nc + cami;
  if (lane == 0) *oi = -1, *od = INT_MAX;
  const vieo_fuse_point& P = pts[m];
  const int skip = P.skip_mask;
  if ((skip & (1u << 31)) || (skip & (1 << cami))) return;
  const float X0 = P.Xw[0], X1 = P.Xw[1], X2 = P.Xw[2];
  const float* R = F.Rcrw;
  float Pcr[3];
  for (int r = 0; r < 3; ++r) Pcr[r] = (R[r * 3] * X0 + R[r * 3 + 1] * X1 + R[r * 3 + 2] * X2) + F.tcrw[r];
  const float* Tc = F.Tcr[cami];
  float Pc[3], twc[3];
  for (int r = 0; r < 3; ++r) Pc[r] = (Tc[r * 4] * Pcr[0] + Tc[r * 4 + 1] * Pcr[1] + Tc[r * 4 + 2] * Pcr[2]) + Tc[r * 4 + 3];
  const float* t = F.trc[cami];
  for (int r = 0; r < 3; ++r) twc[r] = F.Ow[r] + (R[r] * t[0] + R[3 + r] * t[1] + R[6 + r] * t[2]);
  if (Pc[2] <= 0.0f) return;
  const float invz = 1.0f / Pc[2];
  float u, v;
  const CamD& C = fd->cams[cami];
  if (!F.use_distort) {
    const float p0 = Pc[0] * invz, p1 = Pc[1] * invz;
    u = ((float)C.fx * p0 + 0.f * p1) + (float)C.cx * 1.f;
    v = (0.f * p0 + (float)C.fy * p1) + (float)C.cy * 1.f;
  } else {
    const double Pd[3] = {Pc[0], Pc[1], Pc[2]};
    double uv[2];
    cam_project(C, Pd, uv, nullptr);
    u = (float)uv[0], v = (float)uv[1];
  }
  const float* b = F.bounds[cami];
  if (!(u >= b[0] && u < b[1] && v >= b[2] && v < b[3])) return;  // FrameBase::IsInImage
  const float PO[3] = {X0 - twc[0], X1 - twc[1], X2 - twc[2]};
  const float dist3D = sqrtf(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
  if (dist3D < 0.8f * P.min_distance || dist3D > 1.2f * P.max_distance) return;
  if (FF.check_viewing_angle &&
      (double)(PO[0] * P.normal[0] + PO[1] * P.normal[1] + PO[2] * P.normal[2]) < 0.5 * (double)dist3D)
    return;
  const float ratio = P.max_distance / dist3D;
  int lvl = (int)ceilf((float)log((double)ratio) / F.log_scale_factor);  // PredictScale, see oracle/mappoint.cc
  if (lvl < 0)
    lvl = 0;
  else if (lvl >= F.n_levels)
    lvl = F.n_levels - 1;
  const float radius = FF.th_radius * FF.scale_factors[lvl];
  const float winv = (float)kGridCols / (b[1] - b[0]), hinv = (float)kGridRows / (b[3] - b[2]);
  const int min_cellx = max(0, (int)floorf((u - b[0] - radius) * winv));
  const int max_cellx = min(kGridCols - 1, (int)ceilf((u - b[0] + radius) * winv));
  const int min_celly = max(0, (int)floorf((v - b[2] - radius) * hinv));
  const int max_celly = min(kGridRows - 1, (int)ceilf((v - b[2] + radius) * hinv));
  if (min_cellx >= kGridCols || max_cellx < 0 || min_celly >= kGridRows || max_celly < 0) return;
  const uint4 d0 = ((const uint4*)P.desc)[0], d1 = ((const uint4*)P.desc)[1];
  unsigned best = 0xFFFFFFFFu;  // dist << 20 | position in the candidate order
  int bidx = -1, order = 0;
  for (int ix = min_cellx; ix <= max_cellx; ++ix) {
    const int s0 = cell_start[ix * kGridRows + min_celly], s1 = cell_start[ix * kGridRows + max_celly + 1];
    for (int e0 = s0; e0 < s1; e0 += 64) {
      const int e = e0 + lane;
      if (e < s1) {
        const float4 k = cell_rec[e];
        const int pk = __float_as_int(k.w), j = pk & 0xFFFF, oct = pk >> 16;
        bool ok = fabsf(k.x - u) < radius && fabsf(k.y - v) < radius && oct >= lvl - 1 && oct <= lvl;
        if (ok && FF.use_bf) {
          const float ex = u - k.x, ey = v - k.y;
          if (k.z >= 0) {
            const float er = (u - F.bf * invz) - k.z;
            ok = !((double)((ex * ex + ey * ey + er * er) * FF.inv_level_sigma2[oct]) > 7.8);
          } else
            ok = !((double)((ex * ex + ey * ey) * FF.inv_level_sigma2[oct]) > 5.99);
        }
        if (ok) {
          const unsigned d = (unsigned)hamming32q(d0, d1, desc + (size_t)j * 32);
          const unsigned key = (d << 20) | (unsigned)min(order + (e - s0), (1 << 20) - 1);
          if (key < best) best = key, bidx = j;
        }
      }
    }
    order += s1 - s0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned e = __shfl_xor(best, o);
    const int j = __shfl_xor(bidx, o);
    if (e < best) best = e, bidx = j;
  }
  if (lane == 0 && bidx >= 0) *oi = bidx, *od = (int)(best >> 20);
}

struct SbpScratch {
  DevBuf pool, cursor, qrec, q, nq, keys, ur, desc, taken, counts, assign, nm, pts, cam, cell_start, cell_rec, cell_ang, claim, need, fin, kbins;
};
static thread_local SbpScratch g_sbp;
static thread_local struct {
  bool valid = false;
  const void *keys = nullptr, *ur = nullptr, *counts = nullptr, *rec = nullptr;
  int n_frames = 0, key_cap = 0, n_cams = 0;
  hipStream_t st = nullptr;
  float bounds[kMaxCams][4] = {};
} g_grid;
static thread_local bool g_grid_keep = false;
// the sticky request of vieo_sbp_keep_grid is consumed by the NEXT search call whatever becomes of that call (an early
// return used to leave it set for the search after it)
static bool take_grid_keep() {
  const bool k = g_grid_keep;
  g_grid_keep = false;
  return k;
}

// A grid that outlives the call: the window grid of a RESIDENT frame (vieo_search_by_projection*_resident) lives in the
// frame's extractor handle, is built by the frame's first search and read by the later ones.
struct SbpGridRef {
  int* cell_start;
  float4* cell_rec;
  float* cell_ang;
  bool built;  // in: already built for these keys; the caller marks it built after a successful call
};

static int run_search(SbpArgs& A, int n_frames, hipStream_t st, bool keep_grid, SbpGridRef* ext_grid = nullptr) {
  int rc;
  SbpScratch& S = g_sbp;
  if (A.n_cams < 1 || A.n_cams > kMaxCams) {
    set_error("search_by_projection: n_cams = %d (1..%d)", A.n_cams, kMaxCams);
    return VIEO_E_INVALID;
  }
  // candidate pool: 32 per query on average (a single query may hold up to kCandCap), plus the unused tail of one slab
  // per wavefront of k_sbp_candidates
  A.pool_cap = std::max(std::min(A.q_cap, 2 * kMaxKeys) * 32, 2 * kCandCap) + kSbpBlocksFew * 4 * kCandCap;
  static const int pool_lds_env = [] {
    const char* e = getenv("VIEO_SBP_POOL_LDS");
    return e ? atoi(e) : 0;
  }();
  A.pool_lds = std::min(A.pool_cap, pool_lds_env > 0 ? pool_lds_env : 5120);  // 20 KB + ~5 KB of state: 4+ frames / CU
  if ((rc = S.pool.ensure((size_t)n_frames * A.pool_cap * 4)) != VIEO_OK) return rc;
  if ((rc = S.cursor.ensure((size_t)n_frames * 4)) != VIEO_OK) return rc;
  if ((rc = S.qrec.ensure((size_t)n_frames * A.q_cap * sizeof(int2))) != VIEO_OK) return rc;
  A.pool = S.pool.as<unsigned>(), A.cursor = S.cursor.as<int>(), A.qrec = S.qrec.as<int2>();
  if (A.q_cap > (1 << 20)) {
    set_error("search_by_projection: more than %d queries per frame", 1 << 20);
    return VIEO_E_CAPACITY;
  }
  if (A.key_cap > kMaxKeys) {
    set_error("search_by_projection: more than %d keypoints per frame", kMaxKeys);
    return VIEO_E_CAPACITY;
  }
  if (ext_grid) {
    A.cell_start = ext_grid->cell_start, A.cell_rec = ext_grid->cell_rec, A.cell_ang = ext_grid->cell_ang;
  } else {
    if ((rc = S.cell_start.ensure((size_t)n_frames * A.n_cams * (kGridCells + 1) * 4)) != VIEO_OK) return rc;
    if ((rc = S.cell_rec.ensure((size_t)n_frames * A.key_cap * sizeof(float4))) != VIEO_OK) return rc;
    if ((rc = S.cell_ang.ensure((size_t)n_frames * A.key_cap * 4)) != VIEO_OK) return rc;
    A.cell_start = S.cell_start.as<int>(), A.cell_rec = S.cell_rec.as<float4>(), A.cell_ang = S.cell_ang.as<float>();
  }
  // The grid (Frame::mGrid as a CSR + the records in cell order) is a function of the frame's keys alone: the second search
  // of a frame (local map after last frame) reuses the first one's when the caller says the keys are the same
  // (vieo_sbp_keep_grid: the one-call tracker) and the arrays, geometry and stream match.
  const bool same = ext_grid ? ext_grid->built
                             : (keep_grid && g_grid.valid && g_grid.keys == (const void*)A.keys && g_grid.ur == (const void*)A.uright &&
                                g_grid.counts == (const void*)(A.cam_first ? (const void*)A.cam_first : (const void*)A.counts) &&
                                g_grid.n_frames == n_frames && g_grid.key_cap == A.key_cap && g_grid.n_cams == A.n_cams &&
                                g_grid.st == st && g_grid.rec == S.cell_rec.p && !memcmp(g_grid.bounds, A.bounds, sizeof(g_grid.bounds)));
  if (!same) {
    if (n_frames * A.n_cams <= 16)
      hipLaunchKernelGGL(k_sbp_grid<1024>, dim3(n_frames * A.n_cams), dim3(1024), 0, st, A, (int*)A.cell_start, (float4*)A.cell_rec,
                         (float*)A.cell_ang);
    else
      hipLaunchKernelGGL(k_sbp_grid<256>, dim3(n_frames * A.n_cams), dim3(256), 0, st, A, (int*)A.cell_start, (float4*)A.cell_rec,
                         (float*)A.cell_ang);
  }
  else if (ext_grid)  // the grid kernel is what empties the frames' candidate pools: a resident frame may be searched any
    VIEO_HIP_CHECK(hipMemsetAsync(A.cursor, 0, (size_t)n_frames * 4, st));  // number of times on one grid (the tracker's
                                                                            // second search appends behind its first)
  if (!ext_grid) {
    g_grid.valid = true, g_grid.keys = A.keys, g_grid.ur = A.uright;
    g_grid.counts = A.cam_first ? (const void*)A.cam_first : (const void*)A.counts;
    g_grid.n_frames = n_frames, g_grid.key_cap = A.key_cap, g_grid.n_cams = A.n_cams, g_grid.st = st, g_grid.rec = S.cell_rec.p;
    memcpy(g_grid.bounds, A.bounds, sizeof(g_grid.bounds));
  }
  // (VIEO_SBP_BLOCKS: tests pin the block count to reach the per-block list's overflow path with few queries)
  const char* e_blocks = getenv("VIEO_SBP_BLOCKS");
  const int n_blocks = e_blocks && atoi(e_blocks) > 0 ? std::min(atoi(e_blocks), kSbpBlocksFew) : (n_frames <= 2 ? kSbpBlocksFew : kSbpBlocks);
  hipLaunchKernelGGL(k_sbp_candidates, dim3(n_blocks, n_frames), dim3(256),
                     (size_t)A.n_cams * (kGridCells + 1) * sizeof(unsigned short), st, A);
  // VIEO_SBP_ASSIGN=seq: only the sequential replay (A/B runs, tests); VIEO_SBP_MAX_ROUNDS: rounds before a frame is
  // handed to it (1 = every frame with any dependency falls back)
  // (read at every call: the tests switch them in-process)
  const char* e_mode = getenv("VIEO_SBP_ASSIGN");
  const bool seq_only = e_mode && !strcmp(e_mode, "seq");
  const char* e_rounds = getenv("VIEO_SBP_MAX_ROUNDS");
  const int max_rounds = e_rounds && atoi(e_rounds) > 0 ? atoi(e_rounds) : 48;
  const size_t lds = (size_t)A.pool_lds * 4 + (size_t)A.key_cap * 5;
  A.need_seq = nullptr, A.claim = nullptr, A.max_rounds = max_rounds;
  if (!seq_only) {
    if ((rc = S.claim.ensure((size_t)n_frames * A.q_cap * 4)) != VIEO_OK) return rc;
    if ((rc = S.need.ensure((size_t)n_frames * 4)) != VIEO_OK) return rc;
    A.claim = S.claim.as<int>(), A.need_seq = S.need.as<int>();
    SbpArgs P = A;
    size_t lds_par = (size_t)P.pool_lds * 4 + (size_t)P.key_cap * 13;
    if (lds_par > 150 * 1024) P.pool_lds = 0, lds_par = (size_t)P.key_cap * 13;  // many-camera frames: the pool stays in L2
    // (the attribute belongs to the function on the device, not to the calling thread, and a set overwrites it: every
    // launch asks for the one fixed ceiling, so concurrent tracker threads with different key_cap cannot lower each other's)
    if (lds_par > 64 * 1024)
      VIEO_HIP_CHECK(hipFuncSetAttribute((const void*)k_sbp_assign_par, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    // rigs, and the few frames of a tracking call: a workgroup per (frame, camera) with the camera's lists in LDS and
    // rounds flat over the candidate pairs (k_sbp_assign_cam: all of a CU's LDS).  Large batches of one-camera frames
    // keep the thread-per-query form, three frames per CU.  VIEO_SBP_FLAT=0 / 1 forces one or the other (tests, A/B).
    const char* e_flat = getenv("VIEO_SBP_FLAT");
    const int cam_cap = (((A.key_cap + A.n_cams - 1) / A.n_cams) + 3) & ~3;
    const size_t state = (size_t)cam_cap * 13 + (size_t)kCamQ * 12;  // per key | per query of the camera
    // (a camera of more than ~6 000 keys leaves no room for its lists beside its state: the thread-per-query form)
    const bool flat = (e_flat ? atoi(e_flat) != 0 : (P.n_cams > 1 || n_frames <= 16)) && state + 64 + 6 * 4096 <= 150 * 1024;
    if (flat) {
      P.pool_lds = (int)(std::min<size_t>((size_t)P.pool_cap, (150 * 1024 - state - 64) / 6) & ~(size_t)1);  // pairs: 6 bytes
      const size_t lds_cam = (size_t)P.pool_lds * 6 + state;
      const void* fin_was = S.fin.p;
      if ((rc = S.fin.ensure((size_t)n_frames * kFinStride * 4)) != VIEO_OK) return rc;
      if ((rc = S.kbins.ensure((size_t)n_frames * A.key_cap * 4)) != VIEO_OK) return rc;
      if (S.fin.p != fin_was) VIEO_HIP_CHECK(hipMemsetAsync(S.fin.p, 0, S.fin.cap, st));  // (the kernel leaves it clear)
      VIEO_HIP_CHECK(hipFuncSetAttribute((const void*)k_sbp_assign_cam, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      hipLaunchKernelGGL(k_sbp_assign_cam, dim3(n_frames, P.n_cams), dim3(1024), lds_cam, st, P, cam_cap, S.fin.as<int>(),
                         S.kbins.as<unsigned>(), (int)lds_cam);
      VIEO_HIP_CHECK(hipGetLastError());
      return VIEO_OK;  // (a frame that does not settle is replayed by its last workgroup: no second launch)
    }
    hipLaunchKernelGGL(k_sbp_assign_par, dim3(n_frames), dim3(1024), lds_par, st, P);
  }
  hipLaunchKernelGGL(k_sbp_assign, dim3(n_frames), dim3(64), lds, st, A);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

}  // namespace vieo

using namespace vieo;

extern "C" {

int vieo_sbp_keep_grid(int on) {
  g_grid_keep = on != 0;
  return VIEO_OK;
}

int vieo_sbp_project_last_frame_rig_batch_device(const vieo_last_frame_point* d_points, const int32_t* d_n,
                                                 int p_cap, int n_frames, const vieo_sbp_camera* d_cams,
                                                 const vieo_sbp_rig* d_rigs, int n_cams,
                                                 vieo_proj_query* d_queries, void* stream) {
  if (!d_points || !d_n || p_cap <= 0 || n_frames <= 0 || !d_cams || !d_queries || n_cams < 1 || n_cams > kMaxCams ||
      (!d_rigs && n_cams != 1))
    return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  hipLaunchKernelGGL(k_sbp_project, dim3((p_cap * n_cams + 255) / 256, n_frames), dim3(256), 0,
                     (hipStream_t)stream, d_points, d_n, p_cap, d_cams, d_rigs, n_cams, d_queries);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

int vieo_sbp_project_last_frame_batch_device(const vieo_last_frame_point* d_points,
                                             const int32_t* d_n, int p_cap, int n_frames,
                                             const vieo_sbp_camera* d_cams,
                                             vieo_proj_query* d_queries, void* stream) {
  return vieo_sbp_project_last_frame_rig_batch_device(d_points, d_n, p_cap, n_frames, d_cams, nullptr, 1, d_queries,
                                                      stream);
}

static int search_batch(int mode, const vieo_proj_query* d_queries, const int32_t* d_nq, int q_cap, int n_frames,
                        const vieo_keypoint* d_keys, const float* d_uright, const uint8_t* d_desc,
                        const uint8_t* d_taken, const int32_t* d_counts, const int32_t* d_cam_first, int key_cap,
                        int img_first, int img_step, const float* h_bounds, int n_cams, float nn_ratio,
                        int check_orientation, int32_t* d_assign, int32_t* d_nmatches, void* stream) {
  const bool keep_grid = take_grid_keep();
  if (!d_queries || !d_nq || q_cap <= 0 || n_frames <= 0 || !d_keys || !d_uright || !d_desc ||
      (!d_counts && !d_cam_first) || !h_bounds || !d_assign || !d_nmatches || key_cap <= 0 ||
      (mode != VIEO_SBP_LAST_FRAME && mode != VIEO_SBP_LOCAL_MAP && mode != VIEO_SBP_RELOC))
    return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  SbpArgs A;
  memset(&A, 0, sizeof(A));
  A.mode = mode;
  A.queries = d_queries, A.nq = d_nq, A.q_cap = q_cap;
  A.keys = d_keys, A.uright = d_uright, A.desc = d_desc, A.taken = d_taken, A.counts = d_counts;
  A.key_cap = key_cap, A.img_first = img_first, A.img_step = img_step;
  A.n_cams = n_cams, A.cam_first = d_cam_first;
  if (n_cams >= 1 && n_cams <= kMaxCams) memcpy(A.bounds, h_bounds, sizeof(float) * 4 * n_cams);
  A.nn_ratio = nn_ratio, A.check_ori = check_orientation;
  A.assign = d_assign, A.nmatches = d_nmatches;
  return run_search(A, n_frames, (hipStream_t)stream, keep_grid);
}

int vieo_search_by_projection_batch_device(int mode, const vieo_proj_query* d_queries,
                                           const int32_t* d_nq, int q_cap, int n_frames,
                                           const vieo_keypoint* d_keys, const float* d_uright,
                                           const uint8_t* d_desc, const uint8_t* d_taken,
                                           const int32_t* d_counts, int key_cap, int img_first,
                                           int img_step, const float* h_bounds, float nn_ratio,
                                           int check_orientation, int32_t* d_assign,
                                           int32_t* d_nmatches, void* stream) {
  if (!d_counts) return VIEO_E_INVALID;
  return search_batch(mode, d_queries, d_nq, q_cap, n_frames, d_keys, d_uright, d_desc, d_taken, d_counts, nullptr,
                      key_cap, img_first, img_step, h_bounds, 1, nn_ratio, check_orientation, d_assign, d_nmatches,
                      stream);
}

int vieo_search_by_projection_rig_batch_device(int mode, const vieo_proj_query* d_queries, const int32_t* d_nq,
                                               int q_cap, int n_frames, const vieo_keypoint* d_keys,
                                               const float* d_uright, const uint8_t* d_desc,
                                               const uint8_t* d_taken, const int32_t* d_cam_first, int key_cap,
                                               const float* h_bounds, int n_cams, float nn_ratio,
                                               int check_orientation, int32_t* d_assign, int32_t* d_nmatches,
                                               void* stream) {
  if (!d_cam_first) return VIEO_E_INVALID;
  return search_batch(mode, d_queries, d_nq, q_cap, n_frames, d_keys, d_uright, d_desc, d_taken, nullptr, d_cam_first,
                      key_cap, 0, 1, h_bounds, n_cams, nn_ratio, check_orientation, d_assign, d_nmatches, stream);
}

// host-pointer form of the three projections: last frame (rig or not) and key frame
static int project_host(const void* h_points, int n, const vieo_sbp_camera* h_cam, const vieo_sbp_rig* h_rig,
                        int keyframe, float log_scale_factor, vieo_proj_query* h_queries) {
  if (n < 0 || !h_cam || (n > 0 && (!h_points || !h_queries))) return VIEO_E_INVALID;
  const int nc = h_rig ? h_rig->n_cams : 1;
  if (nc < 1 || nc > kMaxCams) {
    set_error("sbp_project: n_cams = %d (1..%d)", nc, kMaxCams);
    return VIEO_E_INVALID;
  }
  if (h_rig)
    for (int c = 0; c < nc; ++c) {
      CamD d;
      if (!cam_from_abi(h_rig->cams[c], d)) {
        set_error("sbp_project: camera %d has an unknown model or coefficient count", c);
        return VIEO_E_INVALID;
      }
    }
  if (h_cam->nlevels < 1 || h_cam->nlevels > 16) {
    set_error("sbp_project: nlevels = %d (1..16)", h_cam->nlevels);
    return VIEO_E_INVALID;
  }
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  if (n == 0) return VIEO_OK;
  SbpScratch& S = g_sbp;
  static thread_local DevBuf dRig;
  if ((rc = S.pts.ensure((size_t)n * 64)) != VIEO_OK) return rc;
  if ((rc = S.cam.ensure(sizeof(vieo_sbp_camera))) != VIEO_OK) return rc;
  if ((rc = S.q.ensure((size_t)n * nc * sizeof(vieo_proj_query))) != VIEO_OK) return rc;
  if ((rc = S.nq.ensure(4)) != VIEO_OK) return rc;
  if ((rc = dRig.ensure(sizeof(vieo_sbp_rig))) != VIEO_OK) return rc;
  VIEO_HIP_CHECK(hipMemcpy(S.pts.p, h_points, (size_t)n * 64, hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(hipMemcpy(S.cam.p, h_cam, sizeof(vieo_sbp_camera), hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(hipMemcpy(S.nq.p, &n, 4, hipMemcpyHostToDevice));
  if (h_rig) VIEO_HIP_CHECK(hipMemcpy(dRig.p, h_rig, sizeof(vieo_sbp_rig), hipMemcpyHostToDevice));
  const vieo_sbp_rig* d_rig = h_rig ? dRig.as<vieo_sbp_rig>() : nullptr;
  if (keyframe) {
    hipLaunchKernelGGL(k_sbp_project_kf, dim3((n * nc + 255) / 256), dim3(256), 0, 0, S.pts.as<vieo_keyframe_point>(),
                       n, S.cam.as<vieo_sbp_camera>(), d_rig, nc, log_scale_factor, S.q.as<vieo_proj_query>());
    VIEO_HIP_CHECK(hipGetLastError());
  } else {
    rc = vieo_sbp_project_last_frame_rig_batch_device(S.pts.as<vieo_last_frame_point>(), S.nq.as<int>(), n, 1,
                                                      S.cam.as<vieo_sbp_camera>(), d_rig, nc,
                                                      S.q.as<vieo_proj_query>(), nullptr);
    if (rc != VIEO_OK) return rc;
  }
  VIEO_HIP_CHECK(hipMemcpy(h_queries, S.q.p, (size_t)n * nc * sizeof(vieo_proj_query), hipMemcpyDeviceToHost));
  return VIEO_OK;
}

int vieo_sbp_project_last_frame(const vieo_last_frame_point* h_points, int n,
                                const vieo_sbp_camera* h_cam, vieo_proj_query* h_queries) {
  return project_host(h_points, n, h_cam, nullptr, 0, 0.f, h_queries);
}

int vieo_sbp_project_last_frame_rig(const vieo_last_frame_point* h_points, int n, const vieo_sbp_camera* h_cam,
                                    const vieo_sbp_rig* h_rig, vieo_proj_query* h_queries) {
  if (!h_rig) return VIEO_E_INVALID;
  return project_host(h_points, n, h_cam, h_rig, 0, 0.f, h_queries);
}

int vieo_sbp_project_keyframe(const vieo_keyframe_point* h_points, int n, const vieo_sbp_camera* h_cam,
                              const vieo_sbp_rig* h_rig, float log_scale_factor, vieo_proj_query* h_queries) {
  if (!(log_scale_factor > 0)) return VIEO_E_INVALID;
  return project_host(h_points, n, h_cam, h_rig, 1, log_scale_factor, h_queries);
}

int vieo_search_by_projection_rig(int mode, const vieo_proj_query* h_queries, int nq, const vieo_keypoint* h_keys,
                                  const float* h_uright, const uint8_t* h_desc, const uint8_t* h_taken, int n_keys,
                                  const int32_t* h_cam_first, const float* h_bounds, int n_cams, float nn_ratio,
                                  int check_orientation, int32_t* h_assign, int32_t* nmatches) {
  if (nq < 0 || n_keys < 0 || !h_bounds || !nmatches || n_cams < 1 || n_cams > kMaxCams || !h_cam_first ||
      (n_keys > 0 && (!h_keys || !h_uright || !h_desc || !h_assign)) || (nq > 0 && !h_queries))
    return VIEO_E_INVALID;
  if (h_cam_first[0] != 0 || h_cam_first[n_cams] != n_keys) {
    set_error("search_by_projection: cam_first must run from 0 to n_keys");
    return VIEO_E_INVALID;
  }
  for (int c = 0; c < n_cams; ++c)
    if (h_cam_first[c] > h_cam_first[c + 1]) {
      set_error("search_by_projection: cam_first is not ascending");
      return VIEO_E_INVALID;
    }
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  *nmatches = 0;
  for (int i = 0; i < n_keys; i++) h_assign[i] = VIEO_SBP_UNCHANGED;
  if (nq == 0 || n_keys == 0) return VIEO_OK;
  static thread_local Staging G;
  G.reset();
  const size_t o_q = G.in(h_queries, (size_t)nq * sizeof(vieo_proj_query)), o_nq = G.in(&nq, 4);
  const size_t o_keys = G.in(h_keys, (size_t)n_keys * sizeof(vieo_keypoint)), o_ur = G.in(h_uright, (size_t)n_keys * 4);
  const size_t o_desc = G.in(h_desc, (size_t)n_keys * 32);
  const size_t o_taken = G.in(h_taken, h_taken ? (size_t)n_keys : 0);
  const size_t o_cf = G.in(h_cam_first, (size_t)(n_cams + 1) * 4);
  const size_t o_assign = G.out((size_t)n_keys * 4), o_nm = G.out(4);
  if ((rc = G.upload(nullptr)) != VIEO_OK) return rc;
  rc = vieo_search_by_projection_rig_batch_device(
      mode, G.d<vieo_proj_query>(o_q), G.d<int>(o_nq), nq, 1, G.d<vieo_keypoint>(o_keys), G.d<float>(o_ur),
      G.d<uint8_t>(o_desc), h_taken ? G.d<uint8_t>(o_taken) : nullptr, G.d<int32_t>(o_cf), n_keys, h_bounds, n_cams,
      nn_ratio, check_orientation, G.d<int>(o_assign), G.d<int>(o_nm), nullptr);
  if (rc != VIEO_OK) return rc;
  if ((rc = G.download(o_assign, nullptr)) != VIEO_OK) return rc;
  memcpy(h_assign, G.h(o_assign), (size_t)n_keys * 4);
  memcpy(nmatches, G.h(o_nm), 4);
  if (*nmatches < 0) {
    set_error("search_by_projection: more than %d window candidates for one query", kCandCap);
    return VIEO_E_CAPACITY;
  }
  return VIEO_OK;
}

int vieo_search_by_projection(int mode, const vieo_proj_query* h_queries, int nq,
                              const vieo_keypoint* h_keys, const float* h_uright,
                              const uint8_t* h_desc, const uint8_t* h_taken, int n_keys,
                              const float* h_bounds, float nn_ratio, int check_orientation,
                              int32_t* h_assign, int32_t* nmatches) {
  const int32_t cam_first[2] = {0, n_keys};
  return vieo_search_by_projection_rig(mode, h_queries, nq, h_keys, h_uright, h_desc, h_taken, n_keys, cam_first,
                                       h_bounds, 1, nn_ratio, check_orientation, h_assign, nmatches);
}

// ---- the projection searches of a RESIDENT frame (round 5) ---------------------------------------------------------
// The frame's keys, descriptors and (after vieo_stereo_match_rectified_resident) uright are still in the extractor handle
// that produced them; a search uploads only what the caller's pointer graph forces -- the last frame's points or the
// window queries, the taken flags -- in ONE block on the handle's stream, builds the window grid at the frame's first
// search and keeps it in the handle, and returns behind ONE synchronisation.
static int resident_common(vieo_orb* fr, const float*& h_uright, const char* who) {
  if (!fr || fr->res_n < 0) {
    set_error("%s: the handle holds no frame (vieo_orb_extract first)", who);
    return VIEO_E_INVALID;
  }
  // a caller that hands back the values the resident matcher returned (the shim passes stereoinfo_.vuright_ always) is
  // served from HBM like one that passes NULL
  if (h_uright && fr->uright_epoch == fr->epoch && (int)fr->uright_host.size() == fr->res_n &&
      (fr->res_n == 0 || !memcmp(h_uright, fr->uright_host.data(), (size_t)fr->res_n * 4)))
    h_uright = nullptr;
  if (!h_uright && fr->uright_epoch != fr->epoch) {
    set_error("%s: no resident uright for this frame (vieo_stereo_match_rectified_resident first, or pass h_uright)", who);
    return VIEO_E_INVALID;
  }
  return require_device();
}

static int resident_search(int mode, vieo_orb* fr, const vieo_last_frame_point* h_points, const vieo_sbp_camera* h_cam,
                           const vieo_proj_query* h_queries, int nq, const float* h_uright, const uint8_t* h_taken,
                           const float* h_bounds, float nn_ratio, int check_orientation, int32_t* h_assign, int32_t* nmatches) {
  const int n_keys = fr->res_n, cap = vieo_orb_max_keypoints(fr);
  *nmatches = 0;
  for (int i = 0; i < n_keys; i++) h_assign[i] = VIEO_SBP_UNCHANGED;
  if (nq == 0 || n_keys == 0) return VIEO_OK;
  int rc;
  // one block: [points or queries | camera | nq | taken | uright] up, [assign | nmatches] back; the projected queries
  // of the last-frame form are device-only
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t r = o;
    o = (o + bytes + 255) & ~(size_t)255;
    return r;
  };
  const size_t o_in = take((size_t)nq * 64), o_cam = take(sizeof(vieo_sbp_camera)), o_nq = take(4);
  const size_t o_taken = take(h_taken ? (size_t)n_keys : 0), o_ur = take(h_uright ? (size_t)n_keys * 4 : 0);
  const size_t in_end = o;
  const size_t o_assign = take((size_t)cap * 4), o_nm = take(4);
  const size_t io_end = o;
  const size_t o_q = take(h_points ? (size_t)nq * sizeof(vieo_proj_query) : 0);
  if ((rc = fr->h_io.ensure(io_end)) != VIEO_OK || (rc = fr->d_io.ensure(o)) != VIEO_OK) return rc;
  if ((rc = fr->g_start.ensure((size_t)(kGridCells + 1) * 4)) != VIEO_OK || (rc = fr->g_rec.ensure((size_t)cap * sizeof(float4))) != VIEO_OK ||
      (rc = fr->g_ang.ensure((size_t)cap * 4)) != VIEO_OK)
    return rc;
  uint8_t* H = (uint8_t*)fr->h_io.p;
  uint8_t* D = (uint8_t*)fr->d_io.p;
  memcpy(H + o_in, h_points ? (const void*)h_points : (const void*)h_queries, (size_t)nq * 64);
  if (h_cam) memcpy(H + o_cam, h_cam, sizeof(vieo_sbp_camera));
  memcpy(H + o_nq, &nq, 4);
  if (h_taken) memcpy(H + o_taken, h_taken, n_keys);
  if (h_uright) memcpy(H + o_ur, h_uright, (size_t)n_keys * 4);
  hipStream_t st = fr->stream;
  VIEO_HIP_CHECK(hipMemcpyAsync(D, H, in_end, hipMemcpyHostToDevice, st));
  const vieo_proj_query* d_q = (const vieo_proj_query*)(D + o_in);
  if (h_points) {
    hipLaunchKernelGGL(k_sbp_project, dim3((nq + 255) / 256, 1), dim3(256), 0, st, (const vieo_last_frame_point*)(D + o_in),
                       (const int*)(D + o_nq), nq, (const vieo_sbp_camera*)(D + o_cam), (const vieo_sbp_rig*)nullptr, 1,
                       (vieo_proj_query*)(D + o_q));
    d_q = (const vieo_proj_query*)(D + o_q);
  }
  SbpArgs A;
  memset(&A, 0, sizeof(A));
  A.mode = mode;
  A.queries = d_q, A.nq = (const int*)(D + o_nq), A.q_cap = nq;
  A.keys = fr->d_kp.as<vieo_keypoint>(), A.desc = fr->d_desc.as<uint8_t>(), A.counts = fr->d_counts.as<int>();
  A.uright = h_uright ? (const float*)(D + o_ur) : fr->d_uright.as<float>();
  A.taken = h_taken ? D + o_taken : nullptr;
  A.key_cap = cap, A.img_first = 0, A.img_step = 0, A.n_cams = 1, A.cam_first = nullptr;
  memcpy(A.bounds[0], h_bounds, 16);
  A.nn_ratio = nn_ratio, A.check_ori = check_orientation;
  A.assign = (int*)(D + o_assign), A.nmatches = (int*)(D + o_nm);
  // the grid holds uright as well: one built from caller-supplied values is not kept
  SbpGridRef G{fr->g_start.as<int>(), fr->g_rec.as<float4>(), fr->g_ang.as<float>(), !h_uright && fr->grid_epoch == fr->epoch};
  (void)take_grid_keep();
  if ((rc = run_search(A, 1, st, false, &G)) != VIEO_OK) return rc;
  fr->grid_epoch = h_uright ? ~0ull : fr->epoch;
  VIEO_HIP_CHECK(hipMemcpyAsync(H + o_assign, D + o_assign, io_end - o_assign, hipMemcpyDeviceToHost, st));
  VIEO_HIP_CHECK(hipStreamSynchronize(st));
  memcpy(h_assign, H + o_assign, (size_t)n_keys * 4);
  memcpy(nmatches, H + o_nm, 4);
  if (*nmatches < 0) {
    set_error("search_by_projection: more than %d window candidates for one query", kCandCap);
    return VIEO_E_CAPACITY;
  }
  return VIEO_OK;
}

// ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono, th_far) (ORBmatcher.cc:1303-1467)
// as ONE call: the projection of the last frame's points and the search.  h_assign[vieo_orb_resident_keys(frame)].
int vieo_search_by_projection_last_frame_resident(vieo_orb* frame, const vieo_last_frame_point* h_points, int n_points,
                                                  const vieo_sbp_camera* h_cam, const float* h_uright, float nn_ratio,
                                                  int check_orientation, int32_t* h_assign, int32_t* nmatches) {
  if (n_points < 0 || !h_cam || !nmatches || (n_points > 0 && !h_points) || h_cam->nlevels < 1 || h_cam->nlevels > 16)
    return VIEO_E_INVALID;
  int rc = resident_common(frame, h_uright, "vieo_search_by_projection_last_frame_resident");
  if (rc != VIEO_OK) return rc;
  if (frame->res_n > 0 && !h_assign) return VIEO_E_INVALID;
  return resident_search(VIEO_SBP_LAST_FRAME, frame, h_points, h_cam, nullptr, n_points, h_uright, nullptr, h_cam->bounds, nn_ratio,
                         check_orientation, h_assign, nmatches);
}

// The search of the other two overloads on queries the caller built (local map: mode VIEO_SBP_LOCAL_MAP from
// MapPoint::GetTrackInfoRef(); relocalisation: VIEO_SBP_RELOC from vieo_sbp_project_keyframe).
int vieo_search_by_projection_resident(int mode, vieo_orb* frame, const vieo_proj_query* h_queries, int nq,
                                       const float* h_uright, const uint8_t* h_taken, const float* h_bounds, float nn_ratio,
                                       int check_orientation, int32_t* h_assign, int32_t* nmatches) {
  if (nq < 0 || !h_bounds || !nmatches || (nq > 0 && !h_queries) ||
      (mode != VIEO_SBP_LAST_FRAME && mode != VIEO_SBP_LOCAL_MAP && mode != VIEO_SBP_RELOC))
    return VIEO_E_INVALID;
  int rc = resident_common(frame, h_uright, "vieo_search_by_projection_resident");
  if (rc != VIEO_OK) return rc;
  if (frame->res_n > 0 && !h_assign) return VIEO_E_INVALID;
  return resident_search(mode, frame, nullptr, nullptr, h_queries, nq, h_uright, h_taken, h_bounds, nn_ratio, check_orientation,
                         h_assign, nmatches);
}

int vieo_fuse_search(const vieo_fuse_frame* h_frame, const vieo_keypoint* const* h_keys,
                     const float* const* h_uright, const uint8_t* const* h_desc, const int32_t* n_keys,
                     const vieo_fuse_point* h_points, int n_points, int32_t* h_best_idx, int32_t* h_best_dist) {
  if (!h_frame || !h_keys || !h_desc || !n_keys || n_points < 0 || (n_points > 0 && (!h_points || !h_best_idx || !h_best_dist)))
    return VIEO_E_INVALID;
  const vieo_frustum_frame& F = h_frame->base;
  const int nc = F.n_cams;
  if (nc < 1 || nc > 4 || !F.cams || F.n_levels <= 0 || F.n_levels > 16) {
    set_error("SearchByProjectionBase: n_cams = %d (1..4) with cameras, n_levels = %d (1..16)", nc, F.n_levels);
    return VIEO_E_INVALID;
  }
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  if (n_points == 0) return VIEO_OK;
  FuseDev fd;
  memset(&fd, 0, sizeof(fd));
  fd.F = *h_frame;
  fd.F.base.cams = nullptr;
  int cap = 1;
  for (int c = 0; c < nc; ++c) {
    if (!cam_from_abi(F.cams[c], fd.cams[c])) {
      set_error("SearchByProjectionBase: camera %d has an unknown model or coefficient count", c);
      return VIEO_E_INVALID;
    }
    if (n_keys[c] < 0 || n_keys[c] > kMaxKeys || (n_keys[c] > 0 && (!h_keys[c] || !h_desc[c]))) {
      set_error("SearchByProjectionBase: camera %d has %d keys (limit %d)", c, n_keys[c], kMaxKeys);
      return n_keys[c] > kMaxKeys ? VIEO_E_CAPACITY : VIEO_E_INVALID;
    }
    cap = std::max(cap, n_keys[c]);
  }
  static thread_local DevBuf dF, dP, dI, dD;
  SbpScratch& S = g_sbp;
#define ENS(b, n) \
  if ((rc = (b).ensure(n)) != VIEO_OK) return rc
  ENS(dF, sizeof(FuseDev));
  ENS(dP, (size_t)n_points * sizeof(vieo_fuse_point));
  ENS(dI, (size_t)n_points * nc * 4);
  ENS(dD, (size_t)n_points * nc * 4);
  ENS(S.keys, (size_t)cap * sizeof(vieo_keypoint));
  ENS(S.ur, (size_t)cap * 4);
  ENS(S.desc, (size_t)cap * 32);
  ENS(S.counts, 8);
  ENS(S.cursor, 4);
  ENS(S.cell_start, (size_t)(kGridCells + 1) * 4);
  ENS(S.cell_rec, (size_t)cap * 16);
  ENS(S.cell_ang, (size_t)cap * 4);
#undef ENS
  VIEO_HIP_CHECK(hipMemcpy(dF.p, &fd, sizeof(fd), hipMemcpyHostToDevice));
  VIEO_HIP_CHECK(hipMemcpy(dP.p, h_points, (size_t)n_points * sizeof(vieo_fuse_point), hipMemcpyHostToDevice));
  std::vector<float> mono;
  for (int c = 0; c < nc; ++c) {
    if (n_keys[c] > 0) {
      VIEO_HIP_CHECK(hipMemcpy(S.keys.p, h_keys[c], (size_t)n_keys[c] * sizeof(vieo_keypoint), hipMemcpyHostToDevice));
      VIEO_HIP_CHECK(hipMemcpy(S.desc.p, h_desc[c], (size_t)n_keys[c] * 32, hipMemcpyHostToDevice));
      const float* ur = h_uright ? h_uright[c] : nullptr;
      if (!ur) {
        mono.assign(n_keys[c], -1.f);
        ur = mono.data();
      }
      VIEO_HIP_CHECK(hipMemcpy(S.ur.p, ur, (size_t)n_keys[c] * 4, hipMemcpyHostToDevice));
    }
    const int cnt[2] = {n_keys[c], 0};
    VIEO_HIP_CHECK(hipMemcpy(S.counts.p, cnt, 8, hipMemcpyHostToDevice));
    SbpArgs A;
    memset(&A, 0, sizeof(A));
    A.keys = S.keys.as<vieo_keypoint>(), A.uright = S.ur.as<float>(), A.counts = S.counts.as<int>();
    A.key_cap = cap, A.img_first = 0, A.img_step = 0;
    A.n_cams = 1;
    for (int q = 0; q < 4; ++q) A.bounds[0][q] = F.bounds[c][q];
    A.cursor = S.cursor.as<int>();
    hipLaunchKernelGGL(k_sbp_grid<1024>, dim3(1), dim3(1024), 0, 0, A, S.cell_start.as<int>(), S.cell_rec.as<float4>(),
                       S.cell_ang.as<float>());
    hipLaunchKernelGGL(k_fuse_search, dim3((n_points + 3) / 4), dim3(256), 0, 0, dF.as<FuseDev>(), c,
                       S.cell_start.as<int>(), S.cell_rec.as<float4>(), S.desc.as<uint8_t>(),
                       dP.as<vieo_fuse_point>(), n_points, dI.as<int32_t>(), dD.as<int32_t>());
    VIEO_HIP_CHECK(hipGetLastError());
    VIEO_HIP_CHECK(hipDeviceSynchronize());  // the per-camera staging buffers are reused
  }
  VIEO_HIP_CHECK(hipMemcpy(h_best_idx, dI.p, (size_t)n_points * nc * 4, hipMemcpyDeviceToHost));
  VIEO_HIP_CHECK(hipMemcpy(h_best_dist, dD.p, (size_t)n_points * nc * 4, hipMemcpyDeviceToHost));
  return VIEO_OK;
}

}  // extern "C"
