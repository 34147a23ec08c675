// fisheye_stereo.hip -- Frame::ComputeStereoFishEyeMatches (reference src/Frame.cc:613-779), the stereo stage
// of the distorted multi-camera configurations (a10).
//
//   device   k_knn2 (matching.hip) for every camera pair; k_fe_pairs: Lowe ratio + the pair triangulation of
//            GeometricCamera::FillMatchesFromPair (common/camera_models/camera_base.h:408-574 ->
//            TriangulateMatches :199-285 -> Triangulate :576-608) for every query row, one lane per row -- the
//            triangulation is a pure function of the pair, so it is evaluated for all rows at once and the
//            order-dependent part only reads its verdicts; k_fe_groups: the all-camera re-triangulation of every
//            group when n_cams > 2 (Frame.cc:704-737).
//            k_fe_fill: the group bookkeeping of FillMatchesFromPair under USE_STRATEGY_MIN_DIST (common/config.h:12)
//            -- sequential in the reference (each match reads what the previous ones wrote) -- as a
//            speculative-parallel walk on one wavefront per frame; k_fe_finish: mvKeys / mDescriptors / vdepth_.
//   No host step: the counts are read on the device, a batch of rig frames is five launches.
// Eigen::JacobiSVD's last right singular vector (camera_base.h:599-600) is obtained by one-sided Jacobi
// rotations on the columns of A (FP64), which is branch-light and register resident for a 4-column matrix.
#include <algorithm>
#include <cmath>
#include <vector>

#include "ba_device.h"
#include "cam_unproject.h"

namespace vieo {

struct FeRig {
  int n_cams;
  CamD cam[4];
  double Rrc[4][9];   // rotation of Trc
  double Tcw[4][12];  // Trc^-1 (3x4), inverted in double like Twi[i].inverse()
  float th[2];        // the two parallax thresholds as FillMatchesFromPair receives them (float)
};

// right singular vector of the smallest singular value of A (M x 4), one-sided Jacobi; A is destroyed
template <int M>
__device__ __forceinline__ void null_vector4(double (&A)[M][4], double* x4) {
  double V[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) V[r][c] = r == c ? 1. : 0.;
  for (int sweep = 0; sweep < 40; ++sweep) {
    bool rotated = false;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        double a = 0, b = 0, g = 0;
#pragma unroll
        for (int r = 0; r < M; ++r) a += A[r][p] * A[r][p], b += A[r][q] * A[r][q], g += A[r][p] * A[r][q];
        if (g == 0 || fabs(g) <= 1e-15 * sqrt(a * b)) continue;
        rotated = true;
        const double zeta = (b - a) / (2 * g);
        const double t = (zeta >= 0 ? 1. : -1.) / (fabs(zeta) + sqrt(1 + zeta * zeta));
        const double cs = 1 / sqrt(1 + t * t), sn = cs * t;
#pragma unroll
        for (int r = 0; r < M; ++r) {
          const double u = A[r][p], v = A[r][q];
          A[r][p] = cs * u - sn * v, A[r][q] = sn * u + cs * v;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double u = V[r][p], v = V[r][q];
          V[r][p] = cs * u - sn * v, V[r][q] = sn * u + cs * v;
        }
      }
    if (!rotated) break;
  }
  double nb = INFINITY;
  x4[0] = x4[1] = x4[2] = x4[3] = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double n = 0;
#pragma unroll
    for (int r = 0; r < M; ++r) n += A[r][c] * A[r][c];
    if (n < nb) {
      nb = n;
#pragma unroll
      for (int r = 0; r < 4; ++r) x4[r] = V[r][c];
    }
  }
}

// GeometricCamera::TriangulateMatches over N cameras ci[] with key points kp[] (camera_base.h:199-285).
// gate[k]: whether the parallax test passes for threshold th[k]; the rest does not depend on the threshold.
// returns false for the reference's empty vector (apart from the parallax gate); czs = depths.
template <int N>
__device__ bool triangulate_matches(const FeRig& R, const int* ci, const float (*kp)[2], const float* sig,
                                    bool* gate, double* p3d, float* czs) {
  double nP[N][3];
#pragma unroll
  for (int i = 0; i < N; ++i) cam_unproject(R.cam[ci[i]], kp[i][0], kp[i][1], nP[i]);
  // "bret" of the reference: every pair has cos > th  =>  no usable parallax
  bool all_above[2] = {true, true};
#pragma unroll
  for (int i = 0; i < N - 1; ++i)
#pragma unroll
    for (int j = i + 1; j < N; ++j) {
      const double* Ri = R.Rrc[ci[i]];
      const double* Rj = R.Rrc[ci[j]];
      double w[3], v[3];
      for (int r = 0; r < 3; ++r) w[r] = Rj[r * 3] * nP[j][0] + Rj[r * 3 + 1] * nP[j][1] + Rj[r * 3 + 2] * nP[j][2];
      for (int r = 0; r < 3; ++r) v[r] = Ri[r] * w[0] + Ri[3 + r] * w[1] + Ri[6 + r] * w[2];
      const double dot = nP[i][0] * v[0] + nP[i][1] * v[1] + nP[i][2] * v[2];
      const double ni = sqrt(nP[i][0] * nP[i][0] + nP[i][1] * nP[i][1] + nP[i][2] * nP[i][2]);
      const double nj = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      const float cosr = (float)(dot / (ni * nj));
      if (cosr <= R.th[0]) all_above[0] = false;
      if (cosr <= R.th[1]) all_above[1] = false;
    }
  gate[0] = !(R.th[0] < 1.f && all_above[0]);
  gate[1] = !(R.th[1] < 1.f && all_above[1]);
  if (!gate[0] && !gate[1]) return false;
  double A[2 * N][4];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double* T = R.Tcw[ci[i]];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      A[2 * i][c] = nP[i][0] * T[8 + c] - T[c];
      A[2 * i + 1][c] = nP[i][1] * T[8 + c] - T[4 + c];
    }
  }
  double x4[4];
  null_vector4<2 * N>(A, x4);
  if (!x4[3]) return false;
  const double X[3] = {x4[0] / x4[3], x4[1] / x4[3], x4[2] / x4[3]};
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double* T = R.Tcw[ci[i]];
    czs[i] = (float)(T[8] * X[0] + T[9] * X[1] + T[10] * X[2] + T[11]);
    if (czs[i] <= 0) return false;
    double Pc[3], uv[2];
    for (int r = 0; r < 3; ++r) Pc[r] = (T[r * 4] * X[0] + T[r * 4 + 1] * X[1] + T[r * 4 + 2] * X[2]) + T[r * 4 + 3];
    cam_project(R.cam[ci[i]], Pc, uv, nullptr);
    const float e0 = (float)uv[0] - kp[i][0], e1 = (float)uv[1] - kp[i][1];
    if (e0 * e0 + e1 * e1 > 5.991f * sig[i]) return false;
  }
  p3d[0] = X[0], p3d[1] = X[1], p3d[2] = X[2];
  return true;
}

// The same for a run-time number of cameras n <= 4 (the groups of one wavefront have 2, 3 and 4 members: three
// instances of the template one after the other were the kernel's time).  The system is padded to 8 rows with zeros,
// which changes no bit: the Jacobi sums gain terms + 0 * 0 and a rotation leaves a zero row zero.
__device__ bool triangulate_matches_n(const FeRig& R, int n, const int* ci, const float (*kp)[2], const float* sig,
                                      bool* gate, double* p3d, float* czs) {
  double nP[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    nP[i][0] = nP[i][1] = nP[i][2] = 0;
    if (i < n) cam_unproject(R.cam[ci[i]], kp[i][0], kp[i][1], nP[i]);
  }
  bool all_above[2] = {true, true};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i + 1; j < 4; ++j)
      if (j < n) {
        const double* Ri = R.Rrc[ci[i]];
        const double* Rj = R.Rrc[ci[j]];
        double w[3], v[3];
        for (int r = 0; r < 3; ++r) w[r] = Rj[r * 3] * nP[j][0] + Rj[r * 3 + 1] * nP[j][1] + Rj[r * 3 + 2] * nP[j][2];
        for (int r = 0; r < 3; ++r) v[r] = Ri[r] * w[0] + Ri[3 + r] * w[1] + Ri[6 + r] * w[2];
        const double dot = nP[i][0] * v[0] + nP[i][1] * v[1] + nP[i][2] * v[2];
        const double ni = sqrt(nP[i][0] * nP[i][0] + nP[i][1] * nP[i][1] + nP[i][2] * nP[i][2]);
        const double nj = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const float cosr = (float)(dot / (ni * nj));
        if (cosr <= R.th[0]) all_above[0] = false;
        if (cosr <= R.th[1]) all_above[1] = false;
      }
  gate[0] = !(R.th[0] < 1.f && all_above[0]);
  gate[1] = !(R.th[1] < 1.f && all_above[1]);
  if (!gate[0] && !gate[1]) return false;
  double A[8][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double* T = R.Tcw[ci[i < n ? i : 0]];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      A[2 * i][c] = i < n ? nP[i][0] * T[8 + c] - T[c] : 0.0;
      A[2 * i + 1][c] = i < n ? nP[i][1] * T[8 + c] - T[4 + c] : 0.0;
    }
  }
  double x4[4];
  null_vector4<8>(A, x4);
  if (!x4[3]) return false;
  const double X[3] = {x4[0] / x4[3], x4[1] / x4[3], x4[2] / x4[3]};
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < n) {
      const double* T = R.Tcw[ci[i]];
      czs[i] = (float)(T[8] * X[0] + T[9] * X[1] + T[10] * X[2] + T[11]);
      if (czs[i] <= 0) return false;
      double Pc[3], uv[2];
      for (int r = 0; r < 3; ++r) Pc[r] = (T[r * 4] * X[0] + T[r * 4 + 1] * X[1] + T[r * 4 + 2] * X[2]) + T[r * 4 + 3];
      cam_project(R.cam[ci[i]], Pc, uv, nullptr);
      const float e0 = (float)uv[0] - kp[i][0], e1 = (float)uv[1] - kp[i][1];
      if (e0 * e0 + e1 * e1 > 5.991f * sig[i]) return false;
    }
  p3d[0] = X[0], p3d[1] = X[1], p3d[2] = X[2];
  return true;
}

struct FePairRec {  // verdict of one knn row
  int32_t idxj;     // absolute key index in camera j, -1: ratio test failed / fewer than two neighbours
  float dist;
  int32_t ok;       // bit k: FillMatchesFromPair's triangulation accepts the pair under threshold k
  int32_t pad;
  double p3d[3];
};

// One batch of camera-rig frames in HBM: inputs = the extractor's arrays ([frame][camera][cap]), everything else is
// written on the device.  The counts are read where they are needed, so the stage runs without a host round trip.
struct FeBatch {
  int part;  // k_fe_finish: 0 everything, 1 the concatenation only (keys, descriptors, ranges, uright), 2 the group part only
  const FeRig* rig;
  const float* level_sigma2;
  const double* Tcr;           // [n_cams][12]
  const vieo_keypoint* keys;   // [frame][cam][cap]
  const uint8_t* desc;         // [frame][cam][cap][32]
  const int32_t* counts;       // [frame][cam][2] = {n, num_mono}
  const int32_t* knn_idx;      // [frame][pair][cap][2]
  const int32_t* knn_dist;
  FePairRec* rec;              // [frame][pair][cap]
  uint32_t* brief;             // [frame][pair][cap]: idxj | dist << 16 | ok << 24 | accepted << 31
  unsigned long long* list;    // [frame][2][n_pairs * cap]: the rows the bookkeeping walks, in the reference's order
  int32_t* hdr;                // [frame][8]: n_groups, n_matches, which, status, good groups (n_cams > 2), rows, steps
  int32_t* group_idx;          // [frame][gcap][n_cams]
  uint8_t* group_good;         // [frame][gcap]
  double* group_p3d;           // [frame][gcap][3]
  int32_t* key_group_cam;      // [frame][cam][cap]
  vieo_keypoint* keys_cat;     // [frame][key_cap] mvKeys (camera-major)
  uint8_t* desc_cat;           // [frame][key_cap][32] mDescriptors
  int32_t* cam_first;          // [frame][n_cams + 1]
  int32_t* frame_counts;       // [frame][2] = {N, 0}: the frame's key count in the layout the tracking glue reads
  float* depth;                // [frame][key_cap] vdepth_
  float* uright;               // [frame][key_cap] vuright_ (-1)
  int32_t* key_group;          // [frame][key_cap] mapcamidx2idxs_ in mvKeys order
  int cap, gcap, n_cams, n_pairs, tries, key_cap;
  signed char pi[6], pj[6];
};

__device__ __forceinline__ int fe_count(const FeBatch& B, int f, int c) { return min(B.counts[((size_t)f * B.n_cams + c) * 2], B.cap); }
__device__ __forceinline__ int fe_mono(const FeBatch& B, int f, int c) { return B.counts[((size_t)f * B.n_cams + c) * 2 + 1]; }
__device__ __forceinline__ int fe_nq(const FeBatch& B, int f, int p) {  // Frame.cc:623
  const int ci = B.pi[p], cj = B.pj[p];
  const int ni = fe_count(B, f, ci), mi = fe_mono(B, f, ci), nj = fe_count(B, f, cj), mj = fe_mono(B, f, cj);
  return (mi >= ni || mj >= nj) ? 0 : ni - mi;
}

// grid (ceil(cap / 64), n_pairs, n_frames)
__global__ void __launch_bounds__(64)
k_fe_pairs(FeBatch B) {
  const int f = blockIdx.z, p = blockIdx.y, q = blockIdx.x * 64 + threadIdx.x;
  if (q >= fe_nq(B, f, p)) return;
  FePairRec r;
  r.idxj = -1, r.dist = 0, r.ok = 0, r.pad = 0, r.p3d[0] = r.p3d[1] = r.p3d[2] = 0;
  const size_t row = ((size_t)f * B.n_pairs + p) * B.cap + q;
  const int32_t* id = B.knn_idx + row * 2;
  const int32_t* dd = B.knn_dist + row * 2;
  uint32_t brief = 0;
  if (id[0] >= 0 && id[1] >= 0) {
    const float d0 = (float)dd[0], d1 = (float)dd[1];
    // Lowe ratio, Frame.cc:661-663 (float distance against double products)
    if ((double)d0 < (double)d1 * 0.7 || (d0 < 75.f && (double)d0 < (double)d1 * 0.9)) {
      const int ci[2] = {B.pi[p], B.pj[p]};
      const int ia = q + fe_mono(B, f, ci[0]), ib = id[0] + fe_mono(B, f, ci[1]);
      const vieo_keypoint ka = B.keys[((size_t)f * B.n_cams + ci[0]) * B.cap + ia], kb = B.keys[((size_t)f * B.n_cams + ci[1]) * B.cap + ib];
      const float kp[2][2] = {{ka.x, ka.y}, {kb.x, kb.y}};
      const float sig[2] = {B.level_sigma2[ka.octave], B.level_sigma2[kb.octave]};
      bool gate[2];
      float czs[2];
      r.idxj = ib, r.dist = d0;
      if (triangulate_matches<2>(*B.rig, ci, kp, sig, gate, r.p3d, czs) && czs[0] > 0.0001f && czs[1] > 0.0001f)
        r.ok = (gate[0] ? 1 : 0) | (gate[1] ? 2 : 0);
      brief = (uint32_t)ib | ((uint32_t)dd[0] << 16) | ((uint32_t)r.ok << 24) | 0x80000000u;
    }
  }
  B.rec[row] = r;
  B.brief[row] = brief;
}

// ---- the group tables of FillMatchesFromPair on the device (camera_base.h:408-574, USE_STRATEGY_MIN_DIST) -----------
// The reference walks the ratio-accepted knn rows in order (pair-major, query ascending) and every row reads what the
// earlier ones wrote: which group its two keys belong to, the members and last distances of those groups.  A row whose
// pair failed the triangulation changes nothing (it returns before any write), so only the accepted rows are walked.
// Speculative-parallel form: one wavefront takes the next 64 rows; every lane collects what its row could read or
// write -- its two keys, the groups they point to, and those groups' members in the two cameras (the keys an eviction
// would release) -- and marks each with its lane number (ds_min into owner tables); a lane that finds a smaller number
// on any of its marks depends on an earlier row of the batch.  The rows before the first such lane touch pairwise
// disjoint state, so they are applied in one step with the sequential result (new groups numbered by a prefix count
// over the creating lanes); the walk resumes at that lane, which is then lane 0 and always runs.  With the typical few
// per cent of rows that share a key the walk takes n / 64 + (#dependent rows) steps instead of n.  Tables in LDS:
// key -> group (int16), members (int16) and last distances (uint8: a ratio-accepted Hamming distance is < 180; 255 =
// infinity) per group, the owner words, the row a group's point came from.
static constexpr int kFeInf = 255;

static inline size_t fe_fill_lds(int nc, int cap, int gcap, int n_chunks) {
  return (size_t)4 * nc * cap + (size_t)4 * gcap + (size_t)2 * nc * cap + (size_t)2 * gcap * nc + (size_t)2 * gcap +
         (size_t)2 * (n_chunks + 2) + (size_t)gcap * nc + 64;
}

__global__ void __launch_bounds__(256)
k_fe_fill(FeBatch B) {
  extern __shared__ unsigned s_raw[];
  __shared__ int s_n, s_ng, s_nm, s_status, s_steps;
  const int f = blockIdx.x, nc = B.n_cams, cap = B.cap, gcap = B.gcap, tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cpp = (cap + 63) >> 6, n_chunks = B.n_pairs * cpp, nk = nc * cap;
  unsigned* s_kown = s_raw;                                       // [nc][cap]
  unsigned* s_gown = s_kown + nk;                                 // [gcap]
  short* s_k2g = (short*)(s_gown + gcap);                         // [nc][cap]
  short* s_gidx = s_k2g + ((nk + 1) & ~1);                        // [gcap][nc]
  unsigned short* s_gsrc = (unsigned short*)(s_gidx + ((gcap * nc + 1) & ~1));  // [gcap]
  unsigned short* s_ccnt = s_gsrc + ((gcap + 1) & ~1);            // [n_chunks + 1]
  unsigned char* s_gdist = (unsigned char*)(s_ccnt + ((n_chunks + 2) & ~1));    // [gcap][nc]
  int nq[6], mono[4];
  for (int p = 0; p < 6; p++) nq[p] = p < B.n_pairs ? fe_nq(B, f, p) : 0;
  for (int c = 0; c < 4; c++) mono[c] = c < nc ? fe_mono(B, f, c) : 0;
  const uint32_t* brief = B.brief + (size_t)f * B.n_pairs * cap;
  if (tid == 0) s_nm = 0, s_status = 0, s_steps = 0, s_ng = 0;
  for (int i = tid; i < gcap; i += 256) s_gown[i] = 0xFFFFFFFFu;
  int which = 0;
  for (int k = 0; k < B.tries; k++) {
    which = k;
    unsigned long long* L = B.list + ((size_t)f * 2 + k) * B.n_pairs * cap;
    __syncthreads();
    for (int i = tid; i < nk; i += 256) s_k2g[i] = -1, s_kown[i] = 0xFFFFFFFFu;
    // ---- the accepted rows, compacted in the reference's order
    auto row_word = [&](int c, int* src) -> unsigned {
      const int p = c / cpp, q = (c - p * cpp) * 64 + lane;
      *src = (p << 13) | q;
      unsigned v = 0;
      for (int pp = 0; pp < 6; pp++)  // (select chain: nq[] stays in registers)
        if (pp == p && q < nq[pp]) v = brief[(size_t)p * cap + q];
      return ((v >> 31) & (v >> (24 + k)) & 1u) ? v : 0u;
    };
    for (int c = wave; c < n_chunks; c += 4) {
      int src;
      const unsigned v = row_word(c, &src);
      const int cnt = __popcll(__ballot(v != 0));
      if (lane == 0) s_ccnt[c] = (unsigned short)cnt;
    }
    __syncthreads();
    if (wave == 0) {  // exclusive prefix over the chunk counts
      const int per = (n_chunks + 63) >> 6;
      int sum = 0;
      for (int i = 0; i < per; i++) {
        const int c = lane * per + i;
        if (c < n_chunks) sum += s_ccnt[c];
      }
      int inc = sum;
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
      }
      int run = inc - sum;
      for (int i = 0; i < per; i++) {
        const int c = lane * per + i;
        if (c < n_chunks) {
          const int t = s_ccnt[c];
          s_ccnt[c] = (unsigned short)run;
          run += t;
        }
      }
      if (lane == 63) s_n = inc;
    }
    __syncthreads();
    for (int c = wave; c < n_chunks; c += 4) {
      int src;
      const unsigned v = row_word(c, &src);
      const unsigned long long m = __ballot(v != 0);
      if (v) L[s_ccnt[c] + __popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)(unsigned)src << 32) | v;
    }
    __threadfence_block();
    __syncthreads();
    // ---- the walk
    if (wave == 0) {
      const int n = s_n;
      int pos = 0, ng = 0, nm = s_nm, steps = 0, status = 0;
      while (pos < n) {
        const bool have = pos + lane < n;
        const unsigned long long ent = have ? L[pos + lane] : 0ull;
        const unsigned v = (unsigned)ent, src = (unsigned)(ent >> 32);
        const int p = src >> 13, q = src & 8191;
        const int cami = B.pi[p], camj = B.pj[p];
        int mi = 0;
        for (int c = 0; c < 4; c++)
          if (c == cami) mi = mono[c];
        const int idxi = q + mi, idxj = v & 0xFFFF, dist = (v >> 16) & 0xFF;
        const int ki = cami * cap + idxi, kj = camj * cap + idxj;
        const int gi0 = have ? s_k2g[ki] : -1, gj = have ? s_k2g[kj] : -1;
        const int gi = (gi0 < 0 && gj >= 0) ? gj : gi0;  // iteri = iterj
        // what the row may touch: its keys, their groups, the members those groups hold in the two cameras
        int m0 = -1, m1 = -1, m2 = -1, m3 = -1;
        if (gi >= 0) {
          const int a = s_gidx[gi * nc + cami], b = s_gidx[gi * nc + camj];
          if (a >= 0) m0 = cami * cap + a;
          if (b >= 0) m1 = camj * cap + b;
        }
        if (gj >= 0 && gj != gi) {
          const int a = s_gidx[gj * nc + cami], b = s_gidx[gj * nc + camj];
          if (a >= 0) m2 = cami * cap + a;
          if (b >= 0) m3 = camj * cap + b;
        }
        if (have) {
          atomicMin(&s_kown[ki], (unsigned)lane), atomicMin(&s_kown[kj], (unsigned)lane);
          if (m0 >= 0) atomicMin(&s_kown[m0], (unsigned)lane);
          if (m1 >= 0) atomicMin(&s_kown[m1], (unsigned)lane);
          if (m2 >= 0) atomicMin(&s_kown[m2], (unsigned)lane);
          if (m3 >= 0) atomicMin(&s_kown[m3], (unsigned)lane);
          if (gi >= 0) atomicMin(&s_gown[gi], (unsigned)lane);
          if (gj >= 0) atomicMin(&s_gown[gj], (unsigned)lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bool dep = false;
        if (have) {
          const unsigned ul = (unsigned)lane;
          dep = s_kown[ki] < ul || s_kown[kj] < ul || (m0 >= 0 && s_kown[m0] < ul) || (m1 >= 0 && s_kown[m1] < ul) ||
                (m2 >= 0 && s_kown[m2] < ul) || (m3 >= 0 && s_kown[m3] < ul) || (gi >= 0 && s_gown[gi] < ul) ||
                (gj >= 0 && s_gown[gj] < ul);
        }
        const unsigned long long dm = __ballot(dep);
        const int first = dm ? __builtin_ctzll(dm) : 64;
        const int nexec = min(first, n - pos);
        __builtin_amdgcn_wave_barrier();
        if (have) {  // marks off again (every lane of the batch, run or not)
          s_kown[ki] = 0xFFFFFFFFu, s_kown[kj] = 0xFFFFFFFFu;
          if (m0 >= 0) s_kown[m0] = 0xFFFFFFFFu;
          if (m1 >= 0) s_kown[m1] = 0xFFFFFFFFu;
          if (m2 >= 0) s_kown[m2] = 0xFFFFFFFFu;
          if (m3 >= 0) s_kown[m3] = 0xFFFFFFFFu;
          if (gi >= 0) s_gown[gi] = 0xFFFFFFFFu;
          if (gj >= 0) s_gown[gj] = 0xFFFFFFFFu;
        }
        const bool active = lane < nexec;
        const bool creator = active && gi < 0;  // check0 = check1 = 1: neither key has a group
        const unsigned long long cm = __ballot(creator);
        const int n_new = __popcll(cm);
        if (ng + n_new > gcap) {
          status = 1;
          break;
        }
        bool success = false;
        if (creator) {
          const int g = ng + __popcll(cm & ((1ull << lane) - 1ull));
          for (int t = 0; t < nc; t++) s_gidx[g * nc + t] = -1, s_gdist[g * nc + t] = kFeInf;
          s_gidx[g * nc + cami] = (short)idxi, s_gidx[g * nc + camj] = (short)idxj;
          s_gdist[g * nc + cami] = (unsigned char)dist, s_gdist[g * nc + camj] = (unsigned char)dist;
          s_k2g[ki] = (short)g, s_k2g[kj] = (short)g;
          s_gsrc[g] = (unsigned short)src;
          success = true;
        } else if (active) {
          int g = gi, contradict = (gj >= 0 && gj != g) ? 2 : 0;
          if (contradict) {  // keep the group whose members were matched at the smaller mean distance
            int sum0 = 0, sum1 = 0, cnt0 = 0, cnt1 = 0;
            for (int t = 0; t < nc; t++) {
              if (s_gidx[g * nc + t] >= 0) sum0 += s_gdist[g * nc + t], ++cnt0;
              if (s_gidx[gj * nc + t] >= 0) sum1 += s_gdist[gj * nc + t], ++cnt1;
            }
            if (sum1 * cnt0 < sum0 * cnt1) g = gj, contradict = 1;
          }
          const int ixi = s_gidx[g * nc + cami], ixj = s_gidx[g * nc + camj];
          const int ldi = s_gdist[g * nc + cami], ldj = s_gdist[g * nc + camj];
          const int check0 = (ixi < 0 || (idxi != ixi && ldi > dist)) ? 2 : 0;
          const int check1 = (ixj < 0 || (idxj != ixj && ldj > dist)) ? 2 : 0;
          if (check0 || check1) {
            if (contradict) {
              const int gc = contradict == 1 ? gi : gj;
              if (idxi == s_gidx[gc * nc + cami]) s_k2g[ki] = -1, s_gdist[gc * nc + cami] = kFeInf, s_gidx[gc * nc + cami] = -1;
              if (idxj == s_gidx[gc * nc + camj]) s_k2g[kj] = -1, s_gdist[gc * nc + camj] = kFeInf, s_gidx[gc * nc + camj] = -1;
            }
            if (check0 == 2) {
              const int old = s_gidx[g * nc + cami];
              if (idxi != old) {
                if (old >= 0) s_k2g[cami * cap + old] = -1;
                if (s_k2g[ki] < 0) s_k2g[ki] = (short)g;  // map::emplace keeps an existing entry
                s_gidx[g * nc + cami] = (short)idxi;
              }
              s_gdist[g * nc + cami] = (unsigned char)dist;
            } else if (s_gdist[g * nc + cami] > dist)
              s_gdist[g * nc + cami] = (unsigned char)dist;
            if (check1 == 2) {
              const int old = s_gidx[g * nc + camj];
              if (idxj != old) {
                if (old >= 0) s_k2g[camj * cap + old] = -1;
                if (s_k2g[kj] < 0) s_k2g[kj] = (short)g;
                s_gidx[g * nc + camj] = (short)idxj;
              }
              s_gdist[g * nc + camj] = (unsigned char)dist;
            } else if (s_gdist[g * nc + camj] > dist)
              s_gdist[g * nc + camj] = (unsigned char)dist;
            s_gsrc[g] = (unsigned short)src;
            success = true;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        ng += n_new, nm += __popcll(__ballot(success)), pos += nexec, ++steps;
      }
      if (lane == 0) s_ng = ng, s_nm = nm, s_steps += steps, s_status = status;
    }
    __syncthreads();
    if (s_status || s_nm >= 30) break;  // Frame.cc:689-692 (nMatches is not reset between the two passes)
  }
  // ---- the tables out: members, goodmatches_ (>= 2 members, Frame.cc:695-701), the pair's point, key -> group
  const int ng = s_ng;
  for (int g = tid; g < ng; g += 256) {
    int cnt = 0;
    for (int t = 0; t < nc; t++) {
      const int ix = s_gidx[g * nc + t];
      B.group_idx[((size_t)f * gcap + g) * nc + t] = ix;
      cnt += ix >= 0;
    }
    B.group_good[(size_t)f * gcap + g] = cnt >= 2;
    const unsigned src = s_gsrc[g];
    const FePairRec& r = B.rec[((size_t)f * B.n_pairs + (src >> 13)) * cap + (src & 8191)];
    double* X = B.group_p3d + ((size_t)f * gcap + g) * 3;
    X[0] = r.p3d[0], X[1] = r.p3d[1], X[2] = r.p3d[2];
  }
  for (int i = tid; i < nk; i += 256) B.key_group_cam[(size_t)f * nk + i] = s_k2g[i];
  if (tid == 0) {
    int32_t* h = B.hdr + (size_t)f * 8;
    h[0] = ng, h[1] = s_nm, h[2] = which, h[3] = s_status, h[4] = 0, h[5] = s_n, h[6] = s_steps, h[7] = 0;
  }
}

// Frame.cc:704-737: every good group is triangulated again from all of its cameras (n_cams > 2); grid (ceil(gcap / 64), frames)
__global__ void __launch_bounds__(64)
k_fe_groups(FeBatch B) {
  const int f = blockIdx.y, g = blockIdx.x * 64 + threadIdx.x;
  int32_t* hdr = B.hdr + (size_t)f * 8;
  if (g >= hdr[0] || hdr[3]) return;
  uint8_t* good = B.group_good + (size_t)f * B.gcap;
  if (!good[g]) return;
  const int nc = B.n_cams, which = hdr[2];
  const int32_t* gidx = B.group_idx + ((size_t)f * B.gcap + g) * nc;
  int ci[4] = {0, 0, 0, 0}, n = 0;
  float kp[4][2], sig[4], czs[4] = {1, 1, 1, 1};
  for (int k = 0; k < nc; ++k) {
    const int ix = gidx[k];
    if (ix < 0) continue;
    const vieo_keypoint kk = B.keys[((size_t)f * nc + k) * B.cap + ix];
    // select chain instead of ci[n]: keeps the small arrays in registers
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (s == n) ci[s] = k, kp[s][0] = kk.x, kp[s][1] = kk.y, sig[s] = B.level_sigma2[kk.octave];
    ++n;
  }
  double X[3] = {0, 0, 0};
  bool gate[2] = {false, false}, ok = false;
  if (n >= 2 && n <= 4) ok = triangulate_matches_n(*B.rig, n, ci, kp, sig, gate, X, czs);
  ok = ok && gate[which];
#pragma unroll
  for (int s = 0; s < 4; ++s)
    if (s < n && czs[s] <= 0.0001f) ok = false;
  if (ok) {
    double* P = B.group_p3d + ((size_t)f * B.gcap + g) * 3;
    P[0] = X[0], P[1] = X[1], P[2] = X[2];
    atomicAdd(&hdr[4], 1);
  } else
    good[g] = 0;
}

// mvKeys / mDescriptors / vdepth_ / vuright_ of the frame (Frame.cc:742-764): the cameras' key lists one after the
// other, a key's depth in its own camera from its group's point.  grid (ceil(key_cap / 256), frames)
__global__ void __launch_bounds__(256)
k_fe_finish(FeBatch B) {
  const int f = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x, nc = B.n_cams;
  int first[5];
  first[0] = 0;
  for (int c = 0; c < 4; c++) first[c + 1] = first[c] + (c < nc ? fe_count(B, f, c) : 0);
  int32_t* hdr = B.hdr + (size_t)f * 8;
  const bool cat = B.part != 2, grp = B.part != 1;
  if (n == 0) {
    if (cat) {
      for (int c = 0; c <= nc; c++) B.cam_first[(size_t)f * (nc + 1) + c] = first[c];
      B.frame_counts[2 * (size_t)f] = min(first[nc], B.key_cap), B.frame_counts[2 * (size_t)f + 1] = 0;
    }
    if (grp && nc > 2) hdr[1] = hdr[4];  // Frame.cc:705: nMatches counts the re-triangulated groups
  }
  if (n >= first[nc] || n >= B.key_cap) return;
  int c = 0;
  for (int t = 1; t < 4; t++)
    if (t < nc && n >= first[t]) c = t;
  int k = n;
  for (int t = 0; t < 4; t++)
    if (t == c) k = n - first[t];
  const size_t src = ((size_t)f * nc + c) * B.cap + k, dst = (size_t)f * B.key_cap + n;
  if (cat) {
    B.keys_cat[dst] = B.keys[src];
    const uint4* d = (const uint4*)(B.desc + src * 32);
    uint4* o = (uint4*)(B.desc_cat + dst * 32);
    o[0] = d[0], o[1] = d[1];
    B.uright[dst] = -1.f;
  }
  if (!grp) return;
  const int g = hdr[3] ? -1 : B.key_group_cam[src];
  float z = -1;
  if (g >= 0 && B.group_good[(size_t)f * B.gcap + g]) {
    const double* T = B.Tcr + 12 * c;
    const double* X = B.group_p3d + ((size_t)f * B.gcap + g) * 3;
    z = (float)(T[8] * X[0] + T[9] * X[1] + T[10] * X[2] + T[11]);
  }
  B.key_group[dst] = g, B.depth[dst] = z;
}

static thread_local int32_t g_fe_last_steps[2] = {0, 0};

}  // namespace vieo

using namespace vieo;

namespace vieo {
int knn2_rig_launch(const uint8_t* d_desc, const int32_t* d_counts, int cap, int n_cams, int n_frames, int32_t* d_idx,
                    int32_t* d_dist, hipStream_t st);  // matching.hip
}

// rig constants in HBM + the scratch of the stage for up to max_frames frames
struct vieo_fisheye {
  int n_cams = 0, n_pairs = 0, cap = 0, gcap = 0, max_frames = 0, tries = 1, n_levels = 0;
  size_t lds = 0;
  DevBuf consts;  // FeRig | level sigmas | Tcr
  DevBuf idx, dist, rec, brief, list, kgc;
  size_t o_sig = 0, o_tcr = 0;
};

static const size_t kFeLdsMax = 156 * 1024;

extern "C" {

void vieo_fisheye_destroy(vieo_fisheye* h) {
  if (!h) return;
  for (DevBuf* b : {&h->consts, &h->idx, &h->dist, &h->rec, &h->brief, &h->list, &h->kgc}) b->release();
  delete h;
}

int vieo_fisheye_create(vieo_fisheye** out, const vieo_fisheye_params* P, int key_cap_per_camera, int max_frames) {
  if (!out || !P || key_cap_per_camera <= 0 || max_frames <= 0) return VIEO_E_INVALID;
  const int nc = P->n_cams;
  if (nc < 2 || nc > 4 || !P->cams || !P->Trc || !P->Tcr || !P->level_sigma2 || P->n_levels <= 0) {
    set_error("ComputeStereoFishEyeMatches: n_cams = %d (2..4) with cameras, Trc, Tcr and level sigmas", nc);
    return VIEO_E_INVALID;
  }
  if (key_cap_per_camera > 8191) {
    set_error("ComputeStereoFishEyeMatches: %d keys per camera, at most 8191", key_cap_per_camera);
    return VIEO_E_CAPACITY;
  }
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  FeRig R;
  memset(&R, 0, sizeof(R));
  R.n_cams = nc;
  for (int c = 0; c < nc; ++c) {
    if (!cam_from_abi(P->cams[c], R.cam[c])) {
      set_error("ComputeStereoFishEyeMatches: camera %d has an unknown model or coefficient count", c);
      return VIEO_E_INVALID;
    }
    const double* T = P->Trc + 12 * c;
    for (int r = 0; r < 3; ++r) {
      for (int q = 0; q < 3; ++q) R.Rrc[c][r * 3 + q] = T[r * 4 + q], R.Tcw[c][r * 4 + q] = T[q * 4 + r];
      R.Tcw[c][r * 4 + 3] = -(T[0 * 4 + r] * T[3] + T[1 * 4 + r] * T[7] + T[2 * 4 + r] * T[11]);
    }
  }
  // Frame.cc:636-644
  const float f_bar = (P->cams[0].fx + P->cams[0].fy) / 2.;
  double th[2] = {0.9998, 1. - 1e-6};
  if (P->th_far_pts > 0)
    for (int i = 0; i < 2; ++i) th[i] = std::min(1. - std::pow(P->bf / f_bar / P->th_far_pts, 2) / 2., th[i]);
  R.th[0] = (float)th[0], R.th[1] = (float)th[1];
  vieo_fisheye* h = new vieo_fisheye();
  h->n_cams = nc, h->n_pairs = nc * (nc - 1) / 2, h->cap = key_cap_per_camera, h->max_frames = max_frames;
  h->tries = th[1] == th[0] ? 1 : 2, h->n_levels = P->n_levels;
  // groups: as many as keys, or what the LDS tables hold
  const int n_chunks = h->n_pairs * ((h->cap + 63) / 64);
  h->gcap = nc * h->cap;
  if (const char* e = getenv("VIEO_FE_GCAP"))  // tests: a small table to reach the overflow status
    if (atoi(e) > 0) h->gcap = std::min(h->gcap, std::max(atoi(e), 64));
  while (h->gcap > 64 && fe_fill_lds(nc, h->cap, h->gcap, n_chunks) > kFeLdsMax) h->gcap -= 64;
  h->lds = fe_fill_lds(nc, h->cap, h->gcap, n_chunks);
  if (h->lds > kFeLdsMax) {
    set_error("ComputeStereoFishEyeMatches: %d cameras x %d keys do not fit the group tables", nc, h->cap);
    delete h;
    return VIEO_E_CAPACITY;
  }
  h->o_sig = (sizeof(FeRig) + 255) & ~(size_t)255;
  h->o_tcr = h->o_sig + (((size_t)P->n_levels * 4 + 255) & ~(size_t)255);
  const size_t cbytes = h->o_tcr + (size_t)nc * 12 * 8;
  const size_t rows = (size_t)max_frames * h->n_pairs * h->cap;
  std::vector<uint8_t> blk(cbytes, 0);
  memcpy(blk.data(), &R, sizeof(R));
  memcpy(blk.data() + h->o_sig, P->level_sigma2, (size_t)P->n_levels * 4);
  memcpy(blk.data() + h->o_tcr, P->Tcr, (size_t)nc * 12 * 8);
  bool ok = h->consts.ensure(cbytes) == VIEO_OK && h->idx.ensure(rows * 8) == VIEO_OK && h->dist.ensure(rows * 8) == VIEO_OK &&
            h->rec.ensure(rows * sizeof(FePairRec)) == VIEO_OK && h->brief.ensure(rows * 4) == VIEO_OK &&
            h->list.ensure(rows * 2 * 8) == VIEO_OK && h->kgc.ensure((size_t)max_frames * nc * h->cap * 4) == VIEO_OK;
  if (ok && hipMemcpy(h->consts.p, blk.data(), cbytes, hipMemcpyHostToDevice) != hipSuccess) ok = false;
  if (!ok) {
    set_error("vieo_fisheye_create: allocation failed");
    vieo_fisheye_destroy(h);
    return VIEO_E_HIP;
  }
  *out = h;
  return VIEO_OK;
}

int vieo_fisheye_group_capacity(const vieo_fisheye* h) { return h ? h->gcap : 0; }

int vieo_stereo_fisheye_match_batch_device(vieo_fisheye* h, const vieo_keypoint* d_keys, const uint8_t* d_desc,
                                           const int32_t* d_counts, int n_frames, vieo_keypoint* d_keys_cat,
                                           uint8_t* d_desc_cat, int32_t* d_cam_first, int32_t* d_frame_counts,
                                           float* d_depth, float* d_uright,
                                           int32_t* d_key_group, int32_t* d_group_idx, uint8_t* d_group_good,
                                           double* d_group_p3d, int32_t* d_hdr, void* stream) {
  return vieo_stereo_fisheye_match_batch_device_part(h, d_keys, d_desc, d_counts, n_frames, d_keys_cat, d_desc_cat, d_cam_first,
                                                     d_frame_counts, d_depth, d_uright, d_key_group, d_group_idx, d_group_good,
                                                     d_group_p3d, d_hdr, VIEO_FISHEYE_ALL, stream);
}

int vieo_stereo_fisheye_match_batch_device_part(vieo_fisheye* h, const vieo_keypoint* d_keys, const uint8_t* d_desc,
                                                const int32_t* d_counts, int n_frames, vieo_keypoint* d_keys_cat,
                                                uint8_t* d_desc_cat, int32_t* d_cam_first, int32_t* d_frame_counts,
                                                float* d_depth, float* d_uright, int32_t* d_key_group,
                                                int32_t* d_group_idx, uint8_t* d_group_good, double* d_group_p3d,
                                                int32_t* d_hdr, int part, void* stream) {
  if (part < VIEO_FISHEYE_ALL || part > VIEO_FISHEYE_GROUPS) return VIEO_E_INVALID;
  if (!h || !d_keys || !d_desc || !d_counts || n_frames <= 0 || n_frames > h->max_frames || !d_keys_cat || !d_desc_cat ||
      !d_cam_first || !d_frame_counts || !d_depth || !d_uright || !d_key_group || !d_group_idx || !d_group_good || !d_group_p3d || !d_hdr)
    return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int nc = h->n_cams, cap = h->cap;
  FeBatch B;
  memset(&B, 0, sizeof(B));
  B.rig = h->consts.as<FeRig>();
  B.level_sigma2 = (const float*)((uint8_t*)h->consts.p + h->o_sig);
  B.Tcr = (const double*)((uint8_t*)h->consts.p + h->o_tcr);
  B.keys = d_keys, B.desc = d_desc, B.counts = d_counts;
  B.knn_idx = h->idx.as<int32_t>(), B.knn_dist = h->dist.as<int32_t>();
  B.rec = h->rec.as<FePairRec>(), B.brief = h->brief.as<uint32_t>(), B.list = h->list.as<unsigned long long>();
  B.hdr = d_hdr, B.group_idx = d_group_idx, B.group_good = d_group_good, B.group_p3d = d_group_p3d;
  B.key_group_cam = h->kgc.as<int32_t>();
  B.keys_cat = d_keys_cat, B.desc_cat = d_desc_cat, B.cam_first = d_cam_first, B.depth = d_depth, B.uright = d_uright;
  B.frame_counts = d_frame_counts;
  B.key_group = d_key_group;
  B.cap = cap, B.gcap = h->gcap, B.n_cams = nc, B.n_pairs = h->n_pairs, B.tries = h->tries, B.key_cap = nc * cap;
  for (int i = 0, p = 0; i < nc - 1; ++i)
    for (int j = i + 1; j < nc; ++j, ++p) B.pi[p] = (signed char)i, B.pj[p] = (signed char)j;
  B.part = part;
  if (part != VIEO_FISHEYE_CONCAT) {
    // brute force between the key points of all image pairs (Frame.cc:618-628), the pairs' verdicts, the group tables
    if ((rc = knn2_rig_launch(d_desc, d_counts, cap, nc, n_frames, h->idx.as<int32_t>(), h->dist.as<int32_t>(), st)) != VIEO_OK) return rc;
    hipLaunchKernelGGL(k_fe_pairs, dim3((cap + 63) / 64, h->n_pairs, n_frames), dim3(64), 0, st, B);
    if (h->lds > 64 * 1024)
      VIEO_HIP_CHECK(hipFuncSetAttribute((const void*)k_fe_fill, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFeLdsMax));
    hipLaunchKernelGGL(k_fe_fill, dim3(n_frames), dim3(256), h->lds, st, B);
    if (nc > 2) hipLaunchKernelGGL(k_fe_groups, dim3((h->gcap + 63) / 64, n_frames), dim3(64), 0, st, B);
  }
  hipLaunchKernelGGL(k_fe_finish, dim3((B.key_cap + 255) / 256, n_frames), dim3(256), 0, st, B);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

// The host-pointer form: one frame up, the device stage, the tables back.
int vieo_stereo_fisheye_match(const vieo_fisheye_params* P, const vieo_keypoint* const* h_keys,
                              const uint8_t* const* h_desc, const int32_t* n_keys, const int32_t* num_mono,
                              int32_t group_capacity, float* h_depth, int32_t* h_key_group, int32_t* h_group_idx,
                              uint8_t* h_group_good, double* h_group_p3d, int32_t* n_groups, int32_t* n_matches) {
  if (!P || !h_keys || !h_desc || !n_keys || !num_mono || !h_depth || !h_key_group || !h_group_idx || !h_group_good ||
      !h_group_p3d || !n_groups || !n_matches || group_capacity < 0)
    return VIEO_E_INVALID;
  const int nc = P->n_cams;
  if (nc < 2 || nc > 4 || !P->cams || !P->Trc || !P->Tcr || !P->level_sigma2 || P->n_levels <= 0) {
    set_error("ComputeStereoFishEyeMatches: n_cams = %d (2..4) with cameras, Trc, Tcr and level sigmas", nc);
    return VIEO_E_INVALID;
  }
  int cap = 1, N = 0;
  for (int c = 0; c < nc; ++c) {
    if (n_keys[c] < 0 || num_mono[c] < 0 || (n_keys[c] > 0 && (!h_keys[c] || !h_desc[c]))) return VIEO_E_INVALID;
    cap = std::max(cap, n_keys[c]), N += n_keys[c];
    for (int k = 0; k < n_keys[c]; ++k)
      if (h_keys[c][k].octave < 0 || h_keys[c][k].octave >= P->n_levels) {
        set_error("ComputeStereoFishEyeMatches: key %d of camera %d has octave %d", k, c, h_keys[c][k].octave);
        return VIEO_E_INVALID;
      }
  }
  // the rig's handle and the staging blocks are kept per host thread (a sequence calls with one rig)
  static thread_local struct {
    vieo_fisheye* h = nullptr;
    std::vector<uint8_t> key;
    Staging S;
  } C;
  std::vector<uint8_t> key;
  auto put = [&](const void* p, size_t n) { key.insert(key.end(), (const uint8_t*)p, (const uint8_t*)p + n); };
  put(&nc, 4), put(&P->n_levels, 4), put(&P->bf, 4), put(&P->th_far_pts, 4), put(P->cams, sizeof(vieo_camera) * nc);
  put(P->Trc, 96 * nc), put(P->Tcr, 96 * nc), put(P->level_sigma2, 4 * P->n_levels);
  int cur_dev = 0;
  (void)hipGetDevice(&cur_dev);
  put(&cur_dev, 4);
  int rc;
  if (!C.h || C.key != key || C.h->cap < cap) {
    if (C.h) vieo_fisheye_destroy(C.h), C.h = nullptr;
    if ((rc = vieo_fisheye_create(&C.h, P, std::max(cap, 256), 1)) != VIEO_OK) return rc;
    C.key = key;
  }
  vieo_fisheye* h = C.h;
  cap = h->cap;
  Staging& S = C.S;
  S.reset();
  const int gcap = h->gcap, kc = nc * cap;
  std::vector<vieo_keypoint> keys((size_t)nc * cap);
  std::vector<uint8_t> desc((size_t)nc * cap * 32, 0);
  int32_t counts[8];
  memset(keys.data(), 0, keys.size() * sizeof(vieo_keypoint));
  for (int c = 0; c < nc; ++c) {
    counts[2 * c] = n_keys[c], counts[2 * c + 1] = num_mono[c];
    if (n_keys[c] > 0) {
      memcpy(&keys[(size_t)c * cap], h_keys[c], (size_t)n_keys[c] * sizeof(vieo_keypoint));
      memcpy(&desc[(size_t)c * cap * 32], h_desc[c], (size_t)n_keys[c] * 32);
    }
  }
  const size_t i_keys = S.in(keys.data(), keys.size() * sizeof(vieo_keypoint)), i_desc = S.in(desc.data(), desc.size());
  const size_t i_cnt = S.in(counts, sizeof(counts));
  const size_t o_kcat = S.out((size_t)kc * sizeof(vieo_keypoint)), o_dcat = S.out((size_t)kc * 32), o_first = S.out(5 * 4);
  const size_t o_fcnt = S.out(8);
  const size_t o_depth = S.out((size_t)kc * 4), o_ur = S.out((size_t)kc * 4), o_kg = S.out((size_t)kc * 4);
  const size_t o_gidx = S.out((size_t)gcap * nc * 4), o_good = S.out(gcap), o_p3d = S.out((size_t)gcap * 24), o_hdr = S.out(32);
  rc = S.upload(nullptr);
  if (rc == VIEO_OK)
    rc = vieo_stereo_fisheye_match_batch_device(h, S.d<vieo_keypoint>(i_keys), S.d<uint8_t>(i_desc), S.d<int32_t>(i_cnt), 1,
                                                S.d<vieo_keypoint>(o_kcat), S.d<uint8_t>(o_dcat), S.d<int32_t>(o_first), S.d<int32_t>(o_fcnt),
                                                S.d<float>(o_depth), S.d<float>(o_ur), S.d<int32_t>(o_kg), S.d<int32_t>(o_gidx),
                                                S.d<uint8_t>(o_good), S.d<double>(o_p3d), S.d<int32_t>(o_hdr), nullptr);
  if (rc == VIEO_OK) rc = S.download(o_depth, nullptr);
  if (rc == VIEO_OK) {
    const int32_t* hdr = (const int32_t*)S.h(o_hdr);
    const int ng = hdr[0];
    if (hdr[3] || ng > group_capacity) {
      set_error("ComputeStereoFishEyeMatches: %d groups, capacity %d", ng, hdr[3] ? gcap : group_capacity);
      rc = VIEO_E_CAPACITY;
    } else {
      *n_groups = ng, *n_matches = hdr[1];
      if (ng > 0) {
        memcpy(h_group_idx, S.h(o_gidx), (size_t)ng * nc * 4);
        memcpy(h_group_good, S.h(o_good), (size_t)ng);
        memcpy(h_group_p3d, S.h(o_p3d), (size_t)ng * 24);
      }
      if (N > 0) memcpy(h_depth, S.h(o_depth), (size_t)N * 4), memcpy(h_key_group, S.h(o_kg), (size_t)N * 4);
      g_fe_last_steps[0] = hdr[5], g_fe_last_steps[1] = hdr[6];
    }
  }
  return rc;
}

/* test tap: rows walked / wavefront steps taken by the last vieo_stereo_fisheye_match of this thread */
void vieo_fisheye_last_walk(int32_t* rows, int32_t* steps) {
  if (rows) *rows = g_fe_last_steps[0];
  if (steps) *steps = g_fe_last_steps[1];
}

}  // extern "C"
