// mappoint.hip -- the map-point steps either side of the hot path, batched over points:
//   k_in_frustum      Frame::isInFrustum + MapPoint::PredictScale   (reference src/Frame.cc:335-416, MapPoint.cc:491-509)
//   k_distinctive     MapPoint::ComputeDistinctiveDescriptors       (src/MapPoint.cc:314-378)
//   k_normal_depth    MapPoint::UpdateNormalAndDepth                (src/MapPoint.cc:424-480)
// Float arithmetic in the reference's association order (no contraction: -ffp-contract=off; hipcc's f32 divide
// and sqrt are correctly rounded).  HBM-bound, one pass over the points; one lane per point except the
// descriptor medians (one wavefront per point, the N x N distances in LDS).
#include <climits>

#include "ba_device.h"

namespace vieo {

struct FrustumDev {
  vieo_frustum_frame F;
  CamD cams[4];
};

// Frame::isInFrustum for one point (Frame.cc:335-416): the cameras that see it, in push order
__device__ __forceinline__ void frustum_eval(const vieo_frustum_frame& F, const CamD* cams, const vieo_frustum_point& P,
                                             vieo_track_info& T) {
  memset(&T, 0, sizeof(T));
  const float maxDistance = 1.2f * P.max_distance, minDistance = 0.8f * P.min_distance;
  const float* R = F.Rcrw;
  float Pcr[3];
  for (int r = 0; r < 3; ++r) Pcr[r] = (R[r * 3] * P.Xw[0] + R[r * 3 + 1] * P.Xw[1] + R[r * 3 + 2] * P.Xw[2]) + F.tcrw[r];
  float sum_depth = 0;
  int cnt = 0;
  for (int cami = 0; cami < F.n_cams; ++cami) {
    const float* Tc = F.Tcr[cami];
    float Pc[3], twc[3];
    for (int r = 0; r < 3; ++r)
      Pc[r] = (Tc[r * 4] * Pcr[0] + Tc[r * 4 + 1] * Pcr[1] + Tc[r * 4 + 2] * Pcr[2]) + Tc[r * 4 + 3];
    const float* t = F.trc[cami];
    for (int r = 0; r < 3; ++r) twc[r] = F.Ow[r] + (R[r] * t[0] + R[3 + r] * t[1] + R[6 + r] * t[2]);
    const float PcZ = Pc[2];
    if (PcZ < 0.0f) continue;
    const float invz = 1.0f / PcZ;
    float u, v;
    const CamD& C = cams[cami];
    if (!F.use_distort) {
      const float fx = (float)C.fx, fy = (float)C.fy, cx = (float)C.cx, cy = (float)C.cy;
      const float p0 = Pc[0] * invz, p1 = Pc[1] * invz;
      u = (fx * p0 + 0.f * p1) + cx * 1.f;
      v = (0.f * p0 + fy * p1) + cy * 1.f;
    } else {
      const double Pd[3] = {Pc[0], Pc[1], Pc[2]};
      double uv[2];
      cam_project(C, Pd, uv, nullptr);
      u = (float)uv[0], v = (float)uv[1];
    }
    const float* b = F.bounds[cami];
    if (u < b[0] || u > b[1]) continue;
    if (v < b[2] || v > b[3]) continue;
    const float PO[3] = {P.Xw[0] - twc[0], P.Xw[1] - twc[1], P.Xw[2] - twc[2]};
    const float dist3D = sqrtf(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    const float viewCos = (PO[0] * P.normal[0] + PO[1] * P.normal[1] + PO[2] * P.normal[2]) / dist3D;
    if (viewCos < F.viewing_cos_limit) continue;
    const float ratio = P.max_distance / dist3D;
    // logf as the correctly rounded value of the double logarithm (see oracle/mappoint.cc)
    int nscale = (int)ceilf((float)log((double)ratio) / F.log_scale_factor);
    if (nscale < 0)
      nscale = 0;
    else if (nscale >= F.n_levels)
      nscale = F.n_levels - 1;
    // select chain instead of T.u[cnt]: the record stays in registers
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k == cnt) {
        T.u[k] = u, T.v[k] = v, T.ur[k] = u - F.bf * invz;
        T.level[k] = nscale, T.viewcos[k] = viewCos, T.cam[k] = cami;
      }
    ++cnt;
    sum_depth += dist3D;
  }
  T.n = cnt;
  T.track_depth = cnt ? sum_depth / cnt : -1.f;
}

__global__ void __launch_bounds__(256)
k_in_frustum(const FrustumDev* __restrict__ fd, const vieo_frustum_point* __restrict__ pts, int n,
             vieo_track_info* __restrict__ out) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= n) return;
  const vieo_frustum_point P = pts[m];
  vieo_track_info T;
  frustum_eval(fd->F, fd->cams, P, T);
  out[m] = T;
}

// The head of Tracking::SearchLocalPoints (src/Tracking.cc:2308-2370) for a frame whose pose is still in HBM: the
// pose of the first PoseOptimization (its result when it succeeded, the frame's own estimate otherwise) -> Tcw in
// float as Frame::isInFrustum reads it -> isInFrustum of every candidate -> the window queries of
// SearchByProjection(Frame&, vector<MapPoint*>&) (ORBmatcher.cc:237-266), point-major with n_cams slots per point
// (slots of cameras that do not see the point carry flags = 0).  A candidate that aliases an entry of the frame's
// point table which a key already holds (mbTrackInView = false for points matched in the frame, Tracking.cc:2318-
// 2334) gets no query.  One lane per point.
// Batched form: grid y = frame; frame f reads frames[f] / results[f], candidates [f][p_cap], its own count counts[f]
// (null: n for every frame), writes queries [f][p_cap * n_cams], depths at track_depth + f * depth_stride, nq[f].
__global__ void __launch_bounds__(256)
k_track_local_queries(FrustumDev tmpl, const vieo_vio_frame* __restrict__ frame, const vieo_vio_result* __restrict__ result,
                      const vieo_frustum_point* __restrict__ pts, const uint8_t* __restrict__ desc,
                      const int32_t* __restrict__ alias, const uint8_t* __restrict__ held, int held_cap, int n, float th, float th_far,
                      const float* __restrict__ scale, vieo_proj_query* __restrict__ queries,
                      float* __restrict__ track_depth, int32_t* __restrict__ nq, int p_cap, const int32_t* __restrict__ counts,
                      size_t depth_stride) {
  __shared__ vieo_frustum_frame sF;
  {
    const size_t f = blockIdx.y;
    frame += f, result += f, nq += f;
    pts += f * p_cap, desc += f * p_cap * 32;
    if (alias) alias += f * p_cap, held += f * held_cap;
    queries += f * p_cap * tmpl.F.n_cams, track_depth += f * depth_stride;
    if (counts) n = min(counts[f], p_cap);
    if (blockIdx.x * 256 >= (unsigned)max(n, 1)) return;
  }
  if (threadIdx.x == 0) {
    sF = tmpl.F;
    const vieo_navstate& nav = result->base.status == 0 ? result->base.nav : frame->base.nav;
    // Tcw = Tcb * Twb^-1 in double, cast to float (Frame.cc:348-351 reads Tcw_ as float)
    double Rwb[9];
    {
      const double qw = nav.q[0], qx = nav.q[1], qy = nav.q[2], qz = nav.q[3];
      const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
      const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx;
      const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
      Rwb[0] = 1 - (tyy + tzz), Rwb[1] = txy - twz, Rwb[2] = txz + twy;
      Rwb[3] = txy + twz, Rwb[4] = 1 - (txx + tzz), Rwb[5] = tyz - twx;
      Rwb[6] = txz - twy, Rwb[7] = tyz + twx, Rwb[8] = 1 - (txx + tyy);
    }
    const double* Rcb = frame->base.Rcb;
    double Rcw[9], tcw[3];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) Rcw[r * 3 + c] = Rcb[r * 3] * Rwb[c * 3] + Rcb[r * 3 + 1] * Rwb[c * 3 + 1] + Rcb[r * 3 + 2] * Rwb[c * 3 + 2];
    for (int r = 0; r < 3; ++r)
      tcw[r] = frame->base.tcb[r] - (Rcw[r * 3] * nav.p[0] + Rcw[r * 3 + 1] * nav.p[1] + Rcw[r * 3 + 2] * nav.p[2]);
    for (int i = 0; i < 9; ++i) sF.Rcrw[i] = (float)Rcw[i];
    for (int r = 0; r < 3; ++r) {
      sF.tcrw[r] = (float)tcw[r];
      sF.Ow[r] = (float)(-(Rcw[r] * tcw[0] + Rcw[3 + r] * tcw[1] + Rcw[6 + r] * tcw[2]));
    }
    if (blockIdx.x == 0) *nq = n * tmpl.F.n_cams;
  }
  __syncthreads();
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= n) return;
  const vieo_frustum_point P = pts[m];
  vieo_track_info T;
  frustum_eval(sF, tmpl.cams, P, T);
  const int S = sF.n_cams;
  int cnt = T.n;
  if (th_far > 0.f && T.track_depth > th_far) cnt = 0;
  if (alias && alias[m] >= 0 && alias[m] < held_cap && held[alias[m]]) cnt = 0;
  track_depth[m] = T.track_depth;
  const uint4* d = (const uint4*)(desc + (size_t)m * 32);
  const uint4 d0 = d[0], d1 = d[1];
  for (int k = 0; k < S; ++k) {
    vieo_proj_query q;
    memset(&q, 0, sizeof(q));
    if (k < cnt) {
      float u = 0, v = 0, ur = 0, vc = 0;
      int lvl = 0, cam = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j == k) u = T.u[j], v = T.v[j], ur = T.ur[j], vc = T.viewcos[j], lvl = T.level[j], cam = T.cam[j];
      float r = vc > 0.998f ? 2.5f : 4.0f;  // RadiusByViewingCos, then the factor, then the level's scale
      if (th != 1.0f) r = r * th;
      q.u = u, q.v = v, q.ur = ur;
      q.radius = r * scale[lvl];
      q.level_min = lvl - 1, q.level_max = lvl, q.angle = 0.f;
      q.flags = 3 | (cam << 8);
      ((uint4*)q.desc)[0] = d0, ((uint4*)q.desc)[1] = d1;
    }
    queries[(size_t)m * S + k] = q;
  }
}

static const int kMaxObsLds = 128;      // rows of the N x N distance table a wavefront keeps in LDS
static const int kMaxObsPerPoint = 65535;  // bins are 16 bits wide

__device__ __forceinline__ int hamming256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
         __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// one wavefront per point: lane i owns rows i, i + 64, ...; the median of a row is its k-th smallest value
// (k = int(0.5 (N - 1))).  N <= cap: the distances as uint16 in LDS, the k-th smallest by bisection on the value
// (0..256) with counting.  N > cap (no limit in the reference, reachable with rigs and long sessions): the lane keeps
// a 257-bin histogram of its row in LDS instead and walks the cumulative count -- same value, no N x N table.
// slice: uint16 entries of LDS per wavefront (>= cap * cap, and >= 64 * 258 when a point of the batch needs bins).
__global__ void __launch_bounds__(256)
k_distinctive(const uint8_t* __restrict__ desc, const int32_t* __restrict__ first, int n, int32_t* __restrict__ best,
              int cap, int slice) {
  extern __shared__ unsigned short s_all[];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // wave-uniform
  const int p = blockIdx.x * 4 + wv;
  if (p >= n) return;
  const int N = first[p + 1] - first[p];
  if (N <= 0) {
    if (lane == 0) best[p] = -1;
    return;
  }
  unsigned short* sd = s_all + (size_t)wv * slice;
  const uint8_t* D = desc + (size_t)first[p] * 32;
  const int k = (int)(0.5 * (N - 1));
  int bm = INT_MAX, bi = INT_MAX;
  if (N <= cap) {
    for (int i = lane; i < N; i += 64) {
      const uint4 a0 = ((const uint4*)(D + (size_t)i * 32))[0], a1 = ((const uint4*)(D + (size_t)i * 32))[1];
      for (int j = 0; j < N; ++j) {
        const uint4 b0 = ((const uint4*)(D + (size_t)j * 32))[0], b1 = ((const uint4*)(D + (size_t)j * 32))[1];
        sd[i * N + j] = (unsigned short)hamming256(a0, a1, b0, b1);
      }
    }
    for (int i = lane; i < N; i += 64) {
      int lo = 0, hi = 256;  // smallest v with #{d <= v} >= k + 1
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        int c = 0;
        for (int j = 0; j < N; ++j) c += sd[i * N + j] <= mid;
        if (c >= k + 1)
          hi = mid;
        else
          lo = mid + 1;
      }
      if (lo < bm) bm = lo, bi = i;  // rows ascend within a lane: the first minimum stays
    }
  } else {
    unsigned short* h = sd + lane * 258;
    for (int i = lane; i < N; i += 64) {
      for (int v = 0; v < 257; ++v) h[v] = 0;
      const uint4 a0 = ((const uint4*)(D + (size_t)i * 32))[0], a1 = ((const uint4*)(D + (size_t)i * 32))[1];
      for (int j = 0; j < N; ++j) {
        const uint4 b0 = ((const uint4*)(D + (size_t)j * 32))[0], b1 = ((const uint4*)(D + (size_t)j * 32))[1];
        h[hamming256(a0, a1, b0, b1)]++;
      }
      int c = 0, v = 0;
      for (; v < 257; ++v) {
        c += h[v];
        if (c >= k + 1) break;
      }
      if (v < bm) bm = v, bi = i;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int m2 = __shfl_xor(bm, o), i2 = __shfl_xor(bi, o);
    if (m2 < bm || (m2 == bm && i2 < bi)) bm = m2, bi = i2;
  }
  if (lane == 0) best[p] = bi;
}

__global__ void __launch_bounds__(256)
k_normal_depth(const float* __restrict__ pts, const int32_t* __restrict__ first, const int32_t* __restrict__ obs_centre,
               const float* __restrict__ centres, const int32_t* __restrict__ ref_centre,
               const float* __restrict__ ref_scale, float scale_last, int n, float* __restrict__ normal,
               float* __restrict__ max_d, float* __restrict__ min_d) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const float Pos[3] = {pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]};
  float nrm[3] = {0, 0, 0};
  int cnt = 0;
  for (int i = first[p]; i < first[p + 1]; ++i) {
    const float* c = centres + 3 * (size_t)obs_centre[i];
    const float d[3] = {Pos[0] - c[0], Pos[1] - c[1], Pos[2] - c[2]};
    const float nn = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    for (int r = 0; r < 3; ++r) nrm[r] = nrm[r] + d[r] / nn;
    cnt++;
  }
  if (!cnt) {
    normal[3 * p] = normal[3 * p + 1] = normal[3 * p + 2] = 0, max_d[p] = min_d[p] = -1;
    return;
  }
  const float* rc = centres + 3 * (size_t)ref_centre[p];
  const float PC[3] = {Pos[0] - rc[0], Pos[1] - rc[1], Pos[2] - rc[2]};
  const float dist = sqrtf(PC[0] * PC[0] + PC[1] * PC[1] + PC[2] * PC[2]);
  for (int r = 0; r < 3; ++r) normal[3 * p + r] = nrm[r] / cnt;
  const float mx = dist * ref_scale[p];
  max_d[p] = mx;
  min_d[p] = mx / scale_last;
}

struct MpScratch {
  DevBuf a, b, c, d, e, f, g, h, i;
};
static thread_local MpScratch g_mp;

}  // namespace vieo

using namespace vieo;

extern "C" {

#define ENS(buf, bytes) \
  if ((rc = (buf).ensure(std::max<size_t>(bytes, 8))) != VIEO_OK) return rc
#define H2D(buf, src, bytes) \
  if ((bytes) > 0) VIEO_HIP_CHECK(hipMemcpy((buf).p, src, bytes, hipMemcpyHostToDevice))

int vieo_is_in_frustum_batch(const vieo_frustum_frame* h_frame, const vieo_frustum_point* h_points, int n_points,
                             vieo_track_info* h_info) {
  if (!h_frame || n_points < 0 || (n_points > 0 && (!h_points || !h_info))) return VIEO_E_INVALID;
  if (h_frame->n_cams < 1 || h_frame->n_cams > 4 || !h_frame->cams || h_frame->n_levels <= 0) {
    set_error("isInFrustum: n_cams = %d (1..4) with cameras and n_levels > 0", h_frame->n_cams);
    return VIEO_E_INVALID;
  }
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  if (n_points == 0) return VIEO_OK;
  FrustumDev fd;
  memset(&fd, 0, sizeof(fd));
  fd.F = *h_frame;
  fd.F.cams = nullptr;
  for (int c = 0; c < h_frame->n_cams; ++c)
    if (!cam_from_abi(h_frame->cams[c], fd.cams[c])) {
      set_error("isInFrustum: camera %d has an unknown model or coefficient count", c);
      return VIEO_E_INVALID;
    }
  MpScratch& S = g_mp;
  ENS(S.a, sizeof(fd));
  ENS(S.b, (size_t)n_points * sizeof(vieo_frustum_point));
  ENS(S.c, (size_t)n_points * sizeof(vieo_track_info));
  H2D(S.a, &fd, sizeof(fd));
  H2D(S.b, h_points, (size_t)n_points * sizeof(vieo_frustum_point));
  hipLaunchKernelGGL(k_in_frustum, dim3((n_points + 255) / 256), dim3(256), 0, 0, S.a.as<FrustumDev>(),
                     S.b.as<vieo_frustum_point>(), n_points, S.c.as<vieo_track_info>());
  VIEO_HIP_CHECK(hipGetLastError());
  VIEO_HIP_CHECK(hipMemcpy(h_info, S.c.p, (size_t)n_points * sizeof(vieo_track_info), hipMemcpyDeviceToHost));
  return VIEO_OK;
}

int vieo_track_local_queries_device(const vieo_frustum_frame* h_frame, const vieo_vio_frame* d_frame,
                                    const vieo_vio_result* d_result, const vieo_frustum_point* d_points,
                                    const uint8_t* d_desc, const int32_t* d_alias, const uint8_t* d_held, int held_cap,
                                    int n_points, float th, float th_far, const float* d_scale, vieo_proj_query* d_queries,
                                    float* d_track_depth, int32_t* d_nq, void* stream) {
  if (!h_frame || !d_frame || !d_result || n_points < 0 || !d_scale || !d_nq ||
      (n_points > 0 && (!d_points || !d_desc || !d_queries || !d_track_depth)) || (d_alias && (!d_held || held_cap <= 0)))
    return VIEO_E_INVALID;
  if (h_frame->n_cams < 1 || h_frame->n_cams > 4 || !h_frame->cams || h_frame->n_levels <= 0) {
    set_error("SearchLocalPoints: n_cams = %d (1..4) with cameras and n_levels > 0", h_frame->n_cams);
    return VIEO_E_INVALID;
  }
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  FrustumDev fd;
  memset(&fd, 0, sizeof(fd));
  fd.F = *h_frame;
  fd.F.cams = nullptr;
  for (int c = 0; c < h_frame->n_cams; ++c)
    if (!cam_from_abi(h_frame->cams[c], fd.cams[c])) {
      set_error("SearchLocalPoints: camera %d has an unknown model or coefficient count", c);
      return VIEO_E_INVALID;
    }
  hipLaunchKernelGGL(k_track_local_queries, dim3(std::max(1, (n_points + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fd,
                     d_frame, d_result, d_points, d_desc, d_alias, d_held, held_cap, n_points, th, th_far, d_scale, d_queries,
                     d_track_depth, d_nq, n_points, (const int32_t*)nullptr, (size_t)0);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

int vieo_track_local_queries_batch_device(const vieo_frustum_frame* h_frame, const vieo_vio_frame* d_frames,
                                          const vieo_vio_result* d_results, int n_frames, const vieo_frustum_point* d_points,
                                          const uint8_t* d_desc, const int32_t* d_alias, const int32_t* d_counts, int p_cap,
                                          const uint8_t* d_held, int held_cap, float th, float th_far, const float* d_scale,
                                          vieo_proj_query* d_queries, float* d_track_depth, size_t depth_stride,
                                          int32_t* d_nq, void* stream) {
  if (!h_frame || !d_frames || !d_results || n_frames <= 0 || p_cap <= 0 || !d_counts || !d_scale || !d_nq || !d_points ||
      !d_desc || !d_queries || !d_track_depth || (d_alias && (!d_held || held_cap <= 0)))
    return VIEO_E_INVALID;
  if (h_frame->n_cams < 1 || h_frame->n_cams > 4 || !h_frame->cams || h_frame->n_levels <= 0) {
    set_error("SearchLocalPoints: n_cams = %d (1..4) with cameras and n_levels > 0", h_frame->n_cams);
    return VIEO_E_INVALID;
  }
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  FrustumDev fd;
  memset(&fd, 0, sizeof(fd));
  fd.F = *h_frame;
  fd.F.cams = nullptr;
  for (int c = 0; c < h_frame->n_cams; ++c)
    if (!cam_from_abi(h_frame->cams[c], fd.cams[c])) {
      set_error("SearchLocalPoints: camera %d has an unknown model or coefficient count", c);
      return VIEO_E_INVALID;
    }
  hipLaunchKernelGGL(k_track_local_queries, dim3((p_cap + 255) / 256, n_frames), dim3(256), 0, (hipStream_t)stream, fd, d_frames,
                     d_results, d_points, d_desc, d_alias, d_held, held_cap, 0, th, th_far, d_scale, d_queries, d_track_depth,
                     d_nq, p_cap, d_counts, depth_stride);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

int vieo_distinctive_descriptors_batch(const uint8_t* h_descriptors, const int32_t* h_first, int n_points,
                                       int32_t* h_best) {
  if (n_points < 0 || (n_points > 0 && (!h_first || !h_best))) return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  if (n_points == 0) return VIEO_OK;
  const int total = h_first[n_points];
  if (total < 0 || (total > 0 && !h_descriptors)) return VIEO_E_INVALID;
  int n_max = 0;
  for (int p = 0; p < n_points; ++p) {
    const int N = h_first[p + 1] - h_first[p];
    if (N < 0) return VIEO_E_INVALID;
    if (N > kMaxObsPerPoint) {
      set_error("ComputeDistinctiveDescriptors: point %d has %d observations (limit %d)", p, N, kMaxObsPerPoint);
      return VIEO_E_CAPACITY;
    }
    n_max = std::max(n_max, N);
  }
  MpScratch& S = g_mp;
  ENS(S.a, (size_t)total * 32);
  ENS(S.b, (size_t)(n_points + 1) * 4);
  ENS(S.c, (size_t)n_points * 4);
  H2D(S.a, h_descriptors, (size_t)total * 32);
  H2D(S.b, h_first, (size_t)(n_points + 1) * 4);
  // LDS from the batch's largest point: its N x N table, or the per-lane bins when a point exceeds kMaxObsLds
  const int cap = std::min(n_max, kMaxObsLds);
  const int slice = std::max(cap * cap, n_max > kMaxObsLds ? 64 * 258 : 0);
  const size_t lds = (size_t)4 * slice * sizeof(unsigned short);  // <= 132 KB
  VIEO_HIP_CHECK(hipFuncSetAttribute((const void*)k_distinctive, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_distinctive, dim3((n_points + 3) / 4), dim3(256), lds, 0, S.a.as<uint8_t>(),
                     S.b.as<int32_t>(), n_points, S.c.as<int32_t>(), cap, slice);
  VIEO_HIP_CHECK(hipGetLastError());
  VIEO_HIP_CHECK(hipMemcpy(h_best, S.c.p, (size_t)n_points * 4, hipMemcpyDeviceToHost));
  return VIEO_OK;
}

int vieo_update_normal_and_depth_batch(const float* h_points, const int32_t* h_first, const int32_t* h_obs_centre,
                                       const float* h_centres, int n_centres, const int32_t* h_ref_centre,
                                       const float* h_ref_scale, float scale_last_level, int n_points,
                                       float* h_normal, float* h_max_distance, float* h_min_distance) {
  if (n_points < 0 || n_centres < 0 ||
      (n_points > 0 && (!h_points || !h_first || !h_ref_centre || !h_ref_scale || !h_normal || !h_max_distance ||
                        !h_min_distance)))
    return VIEO_E_INVALID;
  int rc = require_device();
  if (rc != VIEO_OK) return rc;
  if (n_points == 0) return VIEO_OK;
  const int total = h_first[n_points];
  if (total < 0 || (total > 0 && (!h_obs_centre || !h_centres))) return VIEO_E_INVALID;
  for (int i = 0; i < total; ++i)
    if (h_obs_centre[i] < 0 || h_obs_centre[i] >= n_centres) return VIEO_E_INVALID;
  for (int p = 0; p < n_points; ++p)
    if (h_first[p + 1] > h_first[p] && (h_ref_centre[p] < 0 || h_ref_centre[p] >= n_centres)) return VIEO_E_INVALID;
  MpScratch& S = g_mp;
  ENS(S.a, (size_t)n_points * 12);
  ENS(S.b, (size_t)(n_points + 1) * 4);
  ENS(S.c, (size_t)total * 4);
  ENS(S.d, (size_t)n_centres * 12);
  ENS(S.e, (size_t)n_points * 4);
  ENS(S.f, (size_t)n_points * 4);
  ENS(S.g, (size_t)n_points * 12);
  ENS(S.h, (size_t)n_points * 4);
  ENS(S.i, (size_t)n_points * 4);
  H2D(S.a, h_points, (size_t)n_points * 12);
  H2D(S.b, h_first, (size_t)(n_points + 1) * 4);
  H2D(S.c, h_obs_centre, (size_t)total * 4);
  H2D(S.d, h_centres, (size_t)n_centres * 12);
  H2D(S.e, h_ref_centre, (size_t)n_points * 4);
  H2D(S.f, h_ref_scale, (size_t)n_points * 4);
  hipLaunchKernelGGL(k_normal_depth, dim3((n_points + 255) / 256), dim3(256), 0, 0, S.a.as<float>(),
                     S.b.as<int32_t>(), S.c.as<int32_t>(), S.d.as<float>(), S.e.as<int32_t>(), S.f.as<float>(),
                     scale_last_level, n_points, S.g.as<float>(), S.h.as<float>(), S.i.as<float>());
  VIEO_HIP_CHECK(hipGetLastError());
  VIEO_HIP_CHECK(hipMemcpy(h_normal, S.g.p, (size_t)n_points * 12, hipMemcpyDeviceToHost));
  VIEO_HIP_CHECK(hipMemcpy(h_max_distance, S.h.p, (size_t)n_points * 4, hipMemcpyDeviceToHost));
  VIEO_HIP_CHECK(hipMemcpy(h_min_distance, S.i.p, (size_t)n_points * 4, hipMemcpyDeviceToHost));
  return VIEO_OK;
}

#undef ENS
#undef H2D

}  // extern "C"
