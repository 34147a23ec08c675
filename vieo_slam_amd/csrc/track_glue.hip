// track_glue.hip -- device-side glue of the per-frame replay (what Tracking.cc does between the
// hot-path calls), so a batch of frames can run extract -> stereo -> search -> pose optimisation
// without a host round trip.  Mirrors, on flattened arrays:
//   * Frame::mvpMapPoints bookkeeping after a search (AddMapPoint / EraseMapPointMatch results)
//   * the observation gathering at the top of PoseOptimization (Optimizer.cc:1704-1786,
//     Optimizer.h:406-490)
//   * "Discard outliers" after PoseOptimization (Tracking.cc:1903-1921) and the
//     Observations()>0 test of the next search (ORBmatcher.cc:289-291).
#include "common.h"

namespace vieo {

// mp_ref[f][key_cap]: index of the point held by keypoint i in the frame's point table, -1 none
// what a search's assignment makes of key i's entry (AddMapPoint / EraseMapPointMatch)
struct MergeArgs {
  const int* assign;  // null: no merge
  int point_offset, reset, query_div, q_cap;
  const vieo_last_frame_point* pts;
  const int* query_src;
};
__device__ __forceinline__ int merged_entry(const MergeArgs& M, int f, int i, int key_cap, int cur_in) {
  int cur = M.reset ? -1 : cur_in;
  const int a = M.assign[(size_t)f * key_cap + i];
  if (a >= 0) {
    const int q = M.query_src ? M.query_src[(size_t)f * M.q_cap + a] : a;  // compacted query list: back to (point, camera)
    int pi = M.query_div > 1 ? q / M.query_div : q;  // a rig's query (point i, camera c) is i * n_cams + c
    // a rig frame's map point is held by one key per camera: all of them stand for the first one's table entry
    if (M.pts) {
      const int rep = M.pts[(size_t)f * key_cap + pi].reserved[0];
      if (rep > 0) pi = rep - 1;
    }
    cur = M.point_offset + pi;
  } else if (a == VIEO_SBP_ERASED)
    cur = -1;
  return cur;
}
__global__ void __launch_bounds__(256)
k_track_merge_assign(const int* __restrict__ assign, int* __restrict__ mp_ref,
                     const int* __restrict__ counts, int key_cap, int img_first, int img_step,
                     int point_offset, int reset, int query_div, const vieo_last_frame_point* __restrict__ pts,
                     const int* __restrict__ query_src, int q_cap) {
  const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  const int img = img_first + f * img_step;
  if (i >= key_cap) return;
  int* m = mp_ref + (size_t)f * key_cap + i;
  if (i >= min(counts[2 * img], key_cap)) {
    *m = -1;
    return;
  }
  const MergeArgs M{assign, point_offset, reset, query_div, q_cap, pts, query_src};
  *m = merged_entry(M, f, i, key_cap, *m);
}

// one workgroup per frame: compact the held points into observations, in keypoint order
__global__ void __launch_bounds__(1024)
k_track_build_obs(int* mp_ref, MergeArgs MG, const float* __restrict__ point_xyz, int p_cap,
                  const vieo_keypoint* __restrict__ keys, const float* __restrict__ uright,
                  const int* __restrict__ counts, int key_cap, int img_first, int img_step,
                  const float* __restrict__ inv_sigma2, const float* __restrict__ point_depth, float close_depth,
                  vieo_pose_obs* __restrict__ obs, int* __restrict__ obs_key, uint8_t* frames_base,
                  size_t frame_stride, size_t nobs_offset, size_t obsbegin_offset,
                  const int* __restrict__ cam_first, int n_cams) {
  __shared__ int s_wsum[16];
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int img = img_first + f * img_step;
  const int N = min(counts[2 * img], key_cap);
  int* m = mp_ref + (size_t)f * key_cap;
  const vieo_keypoint* K = keys + (size_t)img * key_cap;
  const float* ur = uright + (size_t)f * key_cap;
  // A thread owns E consecutive keys, so the order of the edges is the order of the keys with ONE scan over the
  // workgroup: count, scan, write.  (1024 threads for a rig frame's thousands of keys, 256 otherwise.  Chunk after chunk of 256 keys -- a dependent load, three barriers each -- took 30 us
  // for the 6 000 keys of a 4-camera frame.)
  const int nt = (int)blockDim.x, E = (N + nt - 1) / nt, i_lo = tid * E, i_hi = min(i_lo + E, N);
  int cnt = 0;
  if (MG.assign) {  // k_track_merge_assign in the same launch: a thread's keys are its own, before and after
    for (int i = N + tid; i < key_cap; i += nt) m[i] = -1;
#pragma unroll 4
    for (int i = i_lo; i < i_hi; i++) {
      const int v = merged_entry(MG, f, i, key_cap, m[i]);
      m[i] = v;
      cnt += v >= 0 ? 1 : 0;
    }
  } else {
#pragma unroll 8
    for (int i = i_lo; i < i_hi; i++) cnt += m[i] >= 0 ? 1 : 0;
  }
  int inc = cnt;
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) s_wsum[wave] = inc;
  __syncthreads();
  int pos = inc - cnt;
  for (int w = 0; w < wave; w++) pos += s_wsum[w];
  const int* cf = cam_first ? cam_first + (size_t)f * (n_cams + 1) : nullptr;
#pragma unroll 4
  for (int i = i_lo; i < i_hi; i++) {
    const int mi = m[i];
    if (mi < 0) continue;
    const float* X = point_xyz + ((size_t)f * p_cap + mi) * 3;
    vieo_pose_obs o;
    o.Xw[0] = X[0], o.Xw[1] = X[1], o.Xw[2] = X[2];
    const vieo_keypoint k = K[i];
    o.u = k.x, o.v = k.y, o.ur = ur[i];
    o.inv_sigma2 = inv_sigma2[k.octave];
    // bit 0: the point was tracked at less than close_depth (the stereo chi2 gate of the visual-inertial
    // PoseOptimization, Optimizer.h:406-490 / mTrackDepth)
    o.flags = point_depth ? (point_depth[(size_t)f * p_cap + mi] < close_depth ? 1 : 0) : 0;
    if (cf) {  // bits 8..11: the key's camera (mapn2in_, Optimizer.h:424-426)
      int c = 0;
      for (int t = 1; t < n_cams; t++)
        if (i >= cf[t]) c = t;
      o.flags |= c << 8;
    }
    obs[(size_t)f * key_cap + pos] = o;
    obs_key[(size_t)f * key_cap + pos] = i;
    pos++;
  }
  if (tid == nt - 1) {  // (pos: all edges of the frame)
    uint8_t* fr = frames_base + (size_t)f * frame_stride;
    *(int*)(fr + nobs_offset) = pos;
    *(int*)(fr + obsbegin_offset) = f * key_cap;
  }
}

// held[f][p_cap]: 1 for every entry of the frame's point table that a key holds (after the outliers of the last
// PoseOptimization were dropped).  One workgroup per frame.
__global__ void __launch_bounds__(256)
k_track_mark_held(const int* __restrict__ mp_ref, const int* __restrict__ counts, int key_cap, int img_first,
                  int img_step, uint8_t* __restrict__ held, int p_cap) {
  const int f = blockIdx.x, tid = threadIdx.x;
  uint8_t* h = held + (size_t)f * p_cap;
  for (int i = tid; i < p_cap; i += 256) h[i] = 0;
  __syncthreads();
  const int N = min(counts[2 * (img_first + f * img_step)], key_cap);
  const int* m = mp_ref + (size_t)f * key_cap;
  for (int i = tid; i < N; i += 256)
    if (m[i] >= 0 && m[i] < p_cap) h[m[i]] = 1;
}

// after PoseOptimization: drop outlier matches, export the "claimed" flags of the next search,
// chain the optimised state into the next problem (nav block copied verbatim)
__global__ void __launch_bounds__(256)
k_track_after_pose(int* __restrict__ mp_ref, const int* __restrict__ obs_key,
                   const uint8_t* __restrict__ outlier, const uint8_t* frames_base,
                   size_t frame_stride, size_t nobs_offset, int key_cap,
                   const uint8_t* results_base, size_t result_stride, uint8_t* next_frames_base,
                   size_t next_frame_stride, uint8_t* __restrict__ taken, uint8_t* __restrict__ held, int p_cap,
                   const int* __restrict__ counts, int img_first, int img_step) {
  const int f = blockIdx.x, tid = threadIdx.x;
  const int n = *(const int*)(frames_base + (size_t)f * frame_stride + nobs_offset);
  int* m = mp_ref + (size_t)f * key_cap;
  for (int j = tid; j < n; j += 256)
    if (outlier[(size_t)f * key_cap + j]) m[obs_key[(size_t)f * key_cap + j]] = -1;
  __syncthreads();
  if (taken)
    for (int i = tid; i < key_cap; i += 256) taken[(size_t)f * key_cap + i] = m[i] >= 0 ? 1 : 0;
  if (next_frames_base) {
    const double* src = (const double*)(results_base + (size_t)f * result_stride);
    double* dst = (double*)(next_frames_base + (size_t)f * next_frame_stride);
    for (int i = tid; i < (int)(sizeof(vieo_navstate) / 8); i += 256) dst[i] = src[i];
  }
  if (held) {  // k_track_mark_held in the same launch (the one-call tracker: a launch less on its chain)
    uint8_t* h = held + (size_t)f * p_cap;
    for (int i = tid; i < p_cap; i += 256) h[i] = 0;
    __syncthreads();
    const int N = min(counts[2 * (img_first + f * img_step)], key_cap);
    for (int i = tid; i < N; i += 256)
      if (m[i] >= 0 && m[i] < p_cap) h[m[i]] = 1;
  }
}

// The valid queries of a frame moved to the front, order kept (the order is the order in which keys are claimed).  A rig
// frame's first search asks for every (last-frame key, camera) pair, most of which project outside their camera: the
// search kernels walk the list several times, so a list a fifth as long is worth one pass.  Two launches, a workgroup
// per 1024 queries in both (one workgroup per frame read the 4 MB of a 4-camera frame's 65 k records through ONE CU:
// 50 us): the chunks' counts, then every chunk moves its valid records behind those of the chunks before it.
__device__ __forceinline__ bool cq_valid(const uint4* in, int i, int n, uint4* r1) {
  if (i >= n) return false;
  *r1 = in[4 * (size_t)i + 1];
  return (r1->w & 1u) != 0;  // flags: the last word of the second quarter
}
__global__ void __launch_bounds__(1024)
k_track_compact_count(const vieo_proj_query* __restrict__ q_in, const int* __restrict__ nq_in, int q_cap, int* __restrict__ cnt) {
  __shared__ int s_w[16];
  const int f = blockIdx.y, c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = min(nq_in[f], q_cap);
  uint4 r1;
  const bool has = cq_valid((const uint4*)(q_in + (size_t)f * q_cap), c * 1024 + tid, n, &r1);
  const unsigned long long bal = __ballot(has);
  if (lane == 0) s_w[wave] = __popcll(bal);
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int w = 0; w < 16; w++) t += s_w[w];
    cnt[(size_t)f * gridDim.x + c] = t;
  }
}
__global__ void __launch_bounds__(1024)
k_track_compact_move(const vieo_proj_query* __restrict__ q_in, const int* __restrict__ nq_in, int q_cap, const int* __restrict__ cnt,
                     vieo_proj_query* __restrict__ q_out, int* __restrict__ src, int* __restrict__ nq_out) {
  __shared__ int s_w[16];
  __shared__ int s_base;
  const int f = blockIdx.y, c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = min(nq_in[f], q_cap);
  const uint4* in = (const uint4*)(q_in + (size_t)f * q_cap);
  uint4* out = (uint4*)(q_out + (size_t)f * q_cap);
  const int i = c * 1024 + tid;
  uint4 r0 = {0, 0, 0, 0}, r1 = r0, r2 = r0, r3 = r0;
  const bool has = cq_valid(in, i, n, &r1);
  if (has) r0 = in[4 * (size_t)i], r2 = in[4 * (size_t)i + 2], r3 = in[4 * (size_t)i + 3];
  const unsigned long long bal = __ballot(has);
  if (lane == 0) s_w[wave] = __popcll(bal);
  if (wave == 0) {  // the valid records of the chunks before this one
    int t = 0;
    for (int k = lane; k < c; k += 64) t += cnt[(size_t)f * gridDim.x + k];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (lane == 0) s_base = t;
  }
  __syncthreads();
  int off = s_base;
  for (int w = 0; w < wave; w++) off += s_w[w];
  if (has) {
    const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
    out[4 * (size_t)pos] = r0, out[4 * (size_t)pos + 1] = r1, out[4 * (size_t)pos + 2] = r2, out[4 * (size_t)pos + 3] = r3;
    src[(size_t)f * q_cap + pos] = i;
  }
  if (c == (int)gridDim.x - 1 && tid == 0) {
    int t = s_base;
    for (int w = 0; w < 16; w++) t += s_w[w];
    nq_out[f] = t;
  }
}

}  // namespace vieo

using namespace vieo;

extern "C" {

int vieo_track_merge_assign_batch_device(const int32_t* d_assign, int32_t* d_mp_ref,
                                         const int32_t* d_counts, int key_cap, int n_frames,
                                         int img_first, int img_step, int point_offset, int reset,
                                         void* stream) {
  if (!d_assign || !d_mp_ref || !d_counts || key_cap <= 0 || n_frames <= 0) return VIEO_E_INVALID;
  hipLaunchKernelGGL(k_track_merge_assign, dim3((key_cap + 255) / 256, n_frames), dim3(256), 0,
                     (hipStream_t)stream, d_assign, d_mp_ref, d_counts, key_cap, img_first, img_step,
                     point_offset, reset, 1, (const vieo_last_frame_point*)nullptr, (const int*)nullptr, 0);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

int vieo_track_merge_assign_rig_batch_device(const int32_t* d_assign, int32_t* d_mp_ref, const int32_t* d_counts,
                                             int key_cap, int n_frames, int img_first, int img_step, int point_offset,
                                             int reset, int query_div, const vieo_last_frame_point* d_same_point,
                                             const int32_t* d_query_src, int q_cap, void* stream) {
  if (!d_assign || !d_mp_ref || !d_counts || key_cap <= 0 || n_frames <= 0 || query_div < 1) return VIEO_E_INVALID;
  hipLaunchKernelGGL(k_track_merge_assign, dim3((key_cap + 255) / 256, n_frames), dim3(256), 0,
                     (hipStream_t)stream, d_assign, d_mp_ref, d_counts, key_cap, img_first, img_step,
                     point_offset, reset, query_div, d_same_point, d_query_src, q_cap);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

int vieo_track_compact_queries_batch_device(const vieo_proj_query* d_queries, const int32_t* d_nq, int q_cap, int n_frames,
                                            vieo_proj_query* d_queries_out, int32_t* d_src, int32_t* d_nq_out, void* stream) {
  if (!d_queries || !d_nq || q_cap <= 0 || n_frames <= 0 || !d_queries_out || !d_src || !d_nq_out) return VIEO_E_INVALID;
  static thread_local DevBuf counts;  // [frame][chunk] valid records of the chunk (this host thread's calls are ordered)
  const int chunks = (q_cap + 1023) / 1024;
  int rc = counts.ensure((size_t)n_frames * chunks * 4);
  if (rc != VIEO_OK) return rc;
  hipLaunchKernelGGL(k_track_compact_count, dim3(chunks, n_frames), dim3(1024), 0, (hipStream_t)stream, d_queries, d_nq, q_cap,
                     counts.as<int>());
  hipLaunchKernelGGL(k_track_compact_move, dim3(chunks, n_frames), dim3(1024), 0, (hipStream_t)stream, d_queries, d_nq, q_cap,
                     counts.as<int>(), d_queries_out, d_src, d_nq_out);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

int vieo_track_build_obs_batch_device(const int32_t* d_mp_ref, const float* d_point_xyz, int p_cap,
                                      const vieo_keypoint* d_keys, const float* d_uright,
                                      const int32_t* d_counts, int key_cap, int n_frames,
                                      int img_first, int img_step, const float* d_inv_sigma2,
                                      vieo_pose_obs* d_obs, int32_t* d_obs_key, void* d_frames,
                                      int frames_are_vio, void* stream) {
  if (!d_mp_ref || !d_point_xyz || !d_keys || !d_uright || !d_counts || !d_inv_sigma2 || !d_obs ||
      !d_obs_key || !d_frames || key_cap <= 0 || n_frames <= 0)
    return VIEO_E_INVALID;
  const size_t stride = frames_are_vio ? sizeof(vieo_vio_frame) : sizeof(vieo_pose_frame);
  const size_t base = frames_are_vio ? offsetof(vieo_vio_frame, base) : 0;
  hipLaunchKernelGGL(k_track_build_obs, dim3(n_frames), dim3(key_cap > 2048 ? 1024 : 256), 0, (hipStream_t)stream, const_cast<int32_t*>(d_mp_ref),
                     MergeArgs{nullptr, 0, 0, 1, 0, nullptr, nullptr}, d_point_xyz, p_cap, d_keys, d_uright, d_counts, key_cap, img_first, img_step,
                     d_inv_sigma2, (const float*)nullptr, 0.f, d_obs, d_obs_key, (uint8_t*)d_frames, stride,
                     base + offsetof(vieo_pose_frame, n_obs), base + offsetof(vieo_pose_frame, obs_begin),
                     (const int*)nullptr, 0);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

int vieo_track_build_obs_depth_batch_device(const int32_t* d_mp_ref, const float* d_point_xyz,
                                            const float* d_point_depth, float close_depth, int p_cap,
                                            const vieo_keypoint* d_keys, const float* d_uright,
                                            const int32_t* d_counts, int key_cap, int n_frames,
                                            int img_first, int img_step, const float* d_inv_sigma2,
                                            vieo_pose_obs* d_obs, int32_t* d_obs_key, void* d_frames,
                                            int frames_are_vio, void* stream) {
  if (!d_mp_ref || !d_point_xyz || !d_point_depth || !d_keys || !d_uright || !d_counts || !d_inv_sigma2 || !d_obs ||
      !d_obs_key || !d_frames || key_cap <= 0 || n_frames <= 0)
    return VIEO_E_INVALID;
  const size_t stride = frames_are_vio ? sizeof(vieo_vio_frame) : sizeof(vieo_pose_frame);
  const size_t base = frames_are_vio ? offsetof(vieo_vio_frame, base) : 0;
  hipLaunchKernelGGL(k_track_build_obs, dim3(n_frames), dim3(key_cap > 2048 ? 1024 : 256), 0, (hipStream_t)stream, const_cast<int32_t*>(d_mp_ref),
                     MergeArgs{nullptr, 0, 0, 1, 0, nullptr, nullptr}, d_point_xyz, p_cap, d_keys, d_uright, d_counts, key_cap, img_first, img_step,
                     d_inv_sigma2, d_point_depth, close_depth, d_obs, d_obs_key, (uint8_t*)d_frames, stride,
                     base + offsetof(vieo_pose_frame, n_obs), base + offsetof(vieo_pose_frame, obs_begin),
                     (const int*)nullptr, 0);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

int vieo_track_build_obs_rig_batch_device(const int32_t* d_mp_ref, const float* d_point_xyz,
                                          const float* d_point_depth, float close_depth, int p_cap,
                                          const vieo_keypoint* d_keys, const float* d_uright,
                                          const int32_t* d_counts, const int32_t* d_cam_first, int n_cams, int key_cap,
                                          int n_frames, const float* d_inv_sigma2, vieo_pose_obs* d_obs,
                                          int32_t* d_obs_key, void* d_frames, int frames_are_vio, void* stream) {
  if (!d_mp_ref || !d_point_xyz || !d_keys || !d_uright || !d_counts || !d_cam_first || n_cams < 1 || n_cams > 4 ||
      !d_inv_sigma2 || !d_obs || !d_obs_key || !d_frames || key_cap <= 0 || n_frames <= 0)
    return VIEO_E_INVALID;
  const size_t stride = frames_are_vio ? sizeof(vieo_vio_frame) : sizeof(vieo_pose_frame);
  const size_t base = frames_are_vio ? offsetof(vieo_vio_frame, base) : 0;
  hipLaunchKernelGGL(k_track_build_obs, dim3(n_frames), dim3(key_cap > 2048 ? 1024 : 256), 0, (hipStream_t)stream, const_cast<int32_t*>(d_mp_ref),
                     MergeArgs{nullptr, 0, 0, 1, 0, nullptr, nullptr}, d_point_xyz, p_cap, d_keys, d_uright, d_counts, key_cap, 0, 1,
                     d_inv_sigma2, d_point_depth, close_depth, d_obs, d_obs_key, (uint8_t*)d_frames, stride,
                     base + offsetof(vieo_pose_frame, n_obs), base + offsetof(vieo_pose_frame, obs_begin),
                     d_cam_first, n_cams);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

int vieo_track_merge_build_obs_batch_device(const int32_t* d_assign, int32_t* d_mp_ref, int point_offset, int reset, int query_div,
                                            const vieo_last_frame_point* d_same_point, const int32_t* d_query_src, int q_cap,
                                            const float* d_point_xyz, const float* d_point_depth, float close_depth, int p_cap,
                                            const vieo_keypoint* d_keys, const float* d_uright, const int32_t* d_counts,
                                            const int32_t* d_cam_first, int n_cams, int key_cap, int n_frames, int img_first,
                                            int img_step, const float* d_inv_sigma2, vieo_pose_obs* d_obs, int32_t* d_obs_key,
                                            void* d_frames, int frames_are_vio, void* stream) {
  if (!d_assign || !d_mp_ref || query_div < 1 || !d_point_xyz || !d_keys || !d_uright || !d_counts || (d_cam_first && (n_cams < 1 || n_cams > 4)) ||
      !d_inv_sigma2 || !d_obs || !d_obs_key || !d_frames || key_cap <= 0 || n_frames <= 0)
    return VIEO_E_INVALID;
  const size_t stride = frames_are_vio ? sizeof(vieo_vio_frame) : sizeof(vieo_pose_frame);
  const size_t base = frames_are_vio ? offsetof(vieo_vio_frame, base) : 0;
  hipLaunchKernelGGL(k_track_build_obs, dim3(n_frames), dim3(key_cap > 2048 ? 1024 : 256), 0, (hipStream_t)stream, d_mp_ref,
                     MergeArgs{d_assign, point_offset, reset, query_div, q_cap, d_same_point, d_query_src}, d_point_xyz, p_cap, d_keys,
                     d_uright, d_counts, key_cap, img_first, img_step, d_inv_sigma2, d_point_depth, close_depth, d_obs, d_obs_key,
                     (uint8_t*)d_frames, stride, base + offsetof(vieo_pose_frame, n_obs), base + offsetof(vieo_pose_frame, obs_begin),
                     d_cam_first, d_cam_first ? n_cams : 0);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

int vieo_track_mark_held_batch_device(const int32_t* d_mp_ref, const int32_t* d_counts, int key_cap,
                                      int n_frames, int img_first, int img_step, uint8_t* d_held, int p_cap,
                                      void* stream) {
  if (!d_mp_ref || !d_counts || !d_held || key_cap <= 0 || n_frames <= 0 || p_cap <= 0) return VIEO_E_INVALID;
  hipLaunchKernelGGL(k_track_mark_held, dim3(n_frames), dim3(256), 0, (hipStream_t)stream, d_mp_ref, d_counts,
                     key_cap, img_first, img_step, d_held, p_cap);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

int vieo_track_after_pose_batch_device(int32_t* d_mp_ref, const int32_t* d_obs_key,
                                       const uint8_t* d_outlier, const void* d_frames,
                                       const void* d_results, int frames_are_vio, int key_cap,
                                       int n_frames, void* d_next_frames, uint8_t* d_taken,
                                       void* stream) {
  if (!d_mp_ref || !d_obs_key || !d_outlier || !d_frames || !d_results || key_cap <= 0 || n_frames <= 0)
    return VIEO_E_INVALID;
  const size_t stride = frames_are_vio ? sizeof(vieo_vio_frame) : sizeof(vieo_pose_frame);
  const size_t rstride = frames_are_vio ? sizeof(vieo_vio_result) : sizeof(vieo_pose_result);
  hipLaunchKernelGGL(k_track_after_pose, dim3(n_frames), dim3(256), 0, (hipStream_t)stream, d_mp_ref,
                     d_obs_key, d_outlier, (const uint8_t*)d_frames, stride,
                     (frames_are_vio ? offsetof(vieo_vio_frame, base) : 0) + offsetof(vieo_pose_frame, n_obs),
                     key_cap, (const uint8_t*)d_results, rstride, (uint8_t*)d_next_frames, stride,
                     d_taken, (uint8_t*)nullptr, 0, (const int*)nullptr, 0, 0);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

int vieo_track_after_pose_held_batch_device(int32_t* d_mp_ref, const int32_t* d_obs_key, const uint8_t* d_outlier,
                                            const void* d_frames, const void* d_results, int frames_are_vio, int key_cap,
                                            int n_frames, void* d_next_frames, uint8_t* d_taken, const int32_t* d_counts,
                                            int img_first, int img_step, uint8_t* d_held, int p_cap, void* stream) {
  if (!d_mp_ref || !d_obs_key || !d_outlier || !d_frames || !d_results || key_cap <= 0 || n_frames <= 0 || !d_counts || !d_held ||
      p_cap <= 0)
    return VIEO_E_INVALID;
  const size_t stride = frames_are_vio ? sizeof(vieo_vio_frame) : sizeof(vieo_pose_frame);
  const size_t rstride = frames_are_vio ? sizeof(vieo_vio_result) : sizeof(vieo_pose_result);
  hipLaunchKernelGGL(k_track_after_pose, dim3(n_frames), dim3(256), 0, (hipStream_t)stream, d_mp_ref, d_obs_key, d_outlier,
                     (const uint8_t*)d_frames, stride,
                     (frames_are_vio ? offsetof(vieo_vio_frame, base) : 0) + offsetof(vieo_pose_frame, n_obs), key_cap,
                     (const uint8_t*)d_results, rstride, (uint8_t*)d_next_frames, stride, d_taken, d_held, p_cap, d_counts,
                     img_first, img_step);
  VIEO_HIP_CHECK(hipGetLastError());
  return VIEO_OK;
}

}  // extern "C"
