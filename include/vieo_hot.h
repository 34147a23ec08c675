/* vieo_hot.h -- C-ABI of the MI355X-native VIEO_SLAM hot path (libvieo_hot.so).
 *
 * The reference has no FFI layer: its boundary is three C++ classes.  Each entry
 * point below names the reference interface it replaces (file:line relative to the reference
 * root); INTEGRATION.md shows the C++ shim (VIEO_SLAM::ORBextractor / ORBmatcher / Optimizer with
 * the reference signatures) that forwards to these.  Plain pointers and sizes only; no C++ /
 * torch / OpenCV types.  Every function returns an int status: VIEO_OK (0) or a negative
 * VIEO_E_* code, unless stated otherwise.  Nothing here falls back to a CPU implementation: when
 * no gfx950 device (or no kernel image for it) is present, create() fails with VIEO_E_NO_DEVICE.
 *
 * Pointer naming: h_* = host memory, d_* = device (HBM) memory.
 */
#ifndef VIEO_HOT_H
#define VIEO_HOT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIEO_OK 0
#define VIEO_E_INVALID (-1)    /* bad argument */
#define VIEO_E_NO_DEVICE (-2)  /* no HIP device / kernels unavailable: there is NO CPU fallback */
#define VIEO_E_HIP (-3)        /* a HIP runtime call failed; see vieo_last_error() */
#define VIEO_E_CAPACITY (-4)   /* caller buffer too small; required size reported */
#define VIEO_E_EMPTY (-5)      /* empty image (reference: operator() returns -1) */

const char* vieo_last_error(void);
/* 1 if a usable gfx950 device is visible, 0 otherwise (never throws, never computes). */
int vieo_device_available(void);
/* One process per GPU: every host thread that calls into the library selects the process's GPU first
 * (the HIP current device is per thread; torch.cuda.set_device covers only the calling thread). */
int vieo_set_device(int device);
int vieo_get_device(void);
const char* vieo_version(void);

/* ---- device memory / stream helpers (so a C, C++ or ctypes host needs no other runtime) ---- */
int vieo_dev_malloc(void** d_ptr, size_t bytes);
int vieo_dev_free(void* d_ptr);
int vieo_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes);
int vieo_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes);
int vieo_device_synchronize(void);
/* HIP events on a given stream (NULL = default): kernel timing without any other runtime. */
int vieo_event_create(void** ev);
int vieo_event_destroy(void* ev);
int vieo_event_record(void* ev, void* stream);
int vieo_event_elapsed_ms(void* ev0, void* ev1, float* ms);
/* Streams (hipStream_t, non-blocking), pinned host memory and asynchronous copies: frames travel over PCIe on a copy
 * stream while the previous batch is processed; vieo_stream_wait_event orders the two streams. */
int vieo_stream_create(void** stream);
int vieo_stream_destroy(void* stream);
int vieo_stream_synchronize(void* stream);
int vieo_stream_wait_event(void* stream, void* ev);
int vieo_host_alloc_pinned(void** h_ptr, size_t bytes);
int vieo_host_free_pinned(void* h_ptr);
int vieo_memcpy_h2d_async(void* d_dst, const void* h_src, size_t bytes, void* stream);
int vieo_memcpy_d2h_async(void* h_dst, const void* d_src, size_t bytes, void* stream);

/* ---------------------------------------------------------------- ORB extractor ------------
 * Replaces VIEO_SLAM::ORBextractor (include/ORBextractor.h:27-80, src/ORBextractor.cc:391-1081).
 */

/* Same memory layout as cv::KeyPoint (28 bytes): the C++ shim memcpy's. */
typedef struct vieo_keypoint {
  float x, y;     /* level-0 pixel coordinates */
  float size;     /* 31 * scale[octave] truncated (ORBextractor.cc:787) */
  float angle;    /* degrees, cv::fastAtan2 of the intensity centroid */
  float response; /* FAST score */
  int32_t octave;
  int32_t class_id; /* always -1 */
} vieo_keypoint;

typedef struct vieo_orb vieo_orb; /* opaque */

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
 * (ORBextractor.cc:391-456).  Device buffers are sized lazily at the first extract call. */
int vieo_orb_create(vieo_orb** out, int nfeatures, float scale_factor, int nlevels,
                    int ini_th_fast, int min_th_fast);
void vieo_orb_destroy(vieo_orb* e);

/* Getters of include/ORBextractor.h:42-52 (tables are float[nlevels]). */
int vieo_orb_levels(const vieo_orb* e);
float vieo_orb_scale_factor(const vieo_orb* e);
int vieo_orb_scale_factors(const vieo_orb* e, float* h_out);
int vieo_orb_inv_scale_factors(const vieo_orb* e, float* h_out);
int vieo_orb_level_sigma2(const vieo_orb* e, float* h_out);
int vieo_orb_inv_level_sigma2(const vieo_orb* e, float* h_out);
int vieo_orb_features_per_level(const vieo_orb* e, int* h_out);
/* Upper bound on keypoints one image can return (nfeatures + 3*nlevels + slack). */
int vieo_orb_max_keypoints(const vieo_orb* e);

/* int ORBextractor::operator()(image, mask, keypoints, descriptors, pvLappingArea)
 * (ORBextractor.cc:968-1058), host buffers in/out, synchronous.
 *   h_image: 8-bit grey, `stride` bytes per row.        h_lapping: NULL or int[2].
 *   h_keypoints[capacity], h_descriptors[capacity*32].
 *   *n_keypoints = number written; *mono_index = the reference's return value (0 without
 *   lapping area).  Returns VIEO_E_EMPTY for an empty image (reference returns -1). */
int vieo_orb_extract(vieo_orb* e, const uint8_t* h_image, int width, int height, int stride,
                     const int* h_lapping, vieo_keypoint* h_keypoints, uint8_t* h_descriptors,
                     int capacity, int* n_keypoints, int* mono_index);

/* Batched, device-resident form of the same call: n_images frames of identical size, image i at
 * d_images + i*image_pitch_bytes.  Asynchronous on the extractor's stream; outputs stay in HBM:
 *   d_keypoints  [n_images][capacity]      d_descriptors [n_images][capacity][32]
 *   d_counts     [n_images][2] int32 = {n_keypoints, mono_index}
 * This is the form the frame-sharded throughput path (bench.py) uses. */
int vieo_orb_extract_batch_device(vieo_orb* e, const uint8_t* d_images, int n_images, int width,
                                  int height, int stride, size_t image_pitch_bytes,
                                  const int* h_lapping, vieo_keypoint* d_keypoints,
                                  uint8_t* d_descriptors, int capacity, int32_t* d_counts);
int vieo_orb_sync(vieo_orb* e);

/* mvImagePyramid (include/ORBextractor.h:54, read by Frame.cc:457,536-557): size and pixels of
 * pyramid level `level` of image `image_index` of the LAST call, valid until the next call.
 * with_border != 0 returns the (w+38)x(h+38) plane with the 19-px BORDER_REFLECT_101 frame that
 * ComputePyramid builds (ORBextractor.cc:1060-1081); the ROI origin is then at (+19,+19). */
int vieo_orb_level_size(const vieo_orb* e, int level, int* width, int* height);
int vieo_orb_get_level(vieo_orb* e, int image_index, int level, int with_border, uint8_t* h_dst,
                       int dst_stride);
/* Device view of the same plane (borderless): pointer + pitch, for device-side consumers
 * (the rectified stereo matcher reads it in place). */
int vieo_orb_level_device(vieo_orb* e, int image_index, int level, const uint8_t** d_ptr,
                          int* pitch);

/* Wall-clock-free timing of the kernels of the last batch call, measured with HIP events on the
 * extractor's own stream: milliseconds for {pyramid, fast, quadtree, blur, describe, total}. */
#define VIEO_ORB_NSTAGES 6
int vieo_orb_enable_timing(vieo_orb* e, int on); /* also resets the step counter */
int vieo_orb_last_stage_ms(vieo_orb* e, float* h_ms /*[VIEO_ORB_NSTAGES]*/);
/* The stamps of the last 64 batch calls are kept, so K steps can be timed with no host sync in
 * between: steps_back = 0 is the most recent call. */
int vieo_orb_timed_steps(vieo_orb* e);
int vieo_orb_stage_ms(vieo_orb* e, int steps_back, float* h_ms /*[VIEO_ORB_NSTAGES]*/);

/* ---------------------------------------------------------------- Hamming matching ---------
 * Replaces the descriptor searches of src/Frame.cc (stereo) on top of
 * ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1645-1667: Hamming distance of 32-byte rows).
 */

/* cv::BFMatcher(cv::NORM_HAMMING).knnMatch(query, train, matches, 2) as called by
 * Frame::ComputeStereoFishEyeMatches (src/Frame.cc:18,620-628): for every query row the two
 * nearest train rows, ascending distance, ties -> lower train index first.
 * idx/dist are [nq][2] int32; a missing neighbour is (-1, INT_MAX). */
int vieo_hamming_knn2(const uint8_t* h_query, int nq, const uint8_t* h_train, int nt,
                      int32_t* h_idx, int32_t* h_dist);
/* Batched device form over the output arrays of vieo_orb_extract_batch_device: pair p searches
 * image h_pairs[2p] (query) against image h_pairs[2p+1] (train), both restricted to rows
 * [mono_index, n) like the reference (descriptors.rowRange(num_mono, n)); h_counts is the host
 * copy of d_counts.  Results of pair p start at row p*capacity; indices are relative to the
 * train sub-matrix.  Asynchronous on `stream` (a hipStream_t, may be NULL). */
int vieo_hamming_knn2_batch_device(const uint8_t* d_descriptors, const int32_t* h_counts,
                                   int capacity, const int32_t* h_pairs, int n_pairs,
                                   int32_t* d_idx, int32_t* d_dist, void* stream);
/* The same search for a batch of camera-rig frames with the counts read ON THE DEVICE (no host copy, no synchronisation):
 * d_descriptors [n_frames][n_cams][capacity][32], d_counts [n_frames][n_cams][2] = {n, num_mono}; every camera pair
 * (i < j) of every frame, rows [num_mono, n) of camera i against those of camera j (src/Frame.cc:618-628); results of
 * (frame f, pair p) start at row (f * n_pairs + p) * capacity, n_pairs = n_cams (n_cams - 1) / 2.  One launch: a lane
 * owns a query row, the train rows pass through LDS in tiles of 256 (k_knn2, matching.hip). */
int vieo_hamming_knn2_rig_batch_device(const uint8_t* d_descriptors, const int32_t* d_counts, int capacity, int n_cams,
                                       int n_frames, int32_t* d_idx, int32_t* d_dist, void* stream);

/* void Frame::ComputeStereoFishEyeMatches(const float th_far_pts) (src/Frame.cc:613-779), the stereo stage of
 * the distorted multi-camera configurations: dense knn-2 between every camera pair (rows [num_mono, n)),
 * Lowe ratio 0.7 / (<75 and 0.9) (:661-663), camm::GeometricCamera::FillMatchesFromPair with
 * USE_STRATEGY_MIN_DIST (common/config.h:12; camera_base.h:408-574: UnProject, parallax gate, DLT triangulation
 * :576-608, positive depth, reprojection chi2 5.991*sigma2), the second pass with the far-point parallax
 * threshold when fewer than 30 matches survive, the all-camera re-triangulation of every group when
 * n_cams > 2 (:704-737), and vdepth_ of the concatenated key list (:742-764).
 *   device: everything -- the knn-2 searches, every pair / group triangulation and (round 4) the order-dependent group
 *   bookkeeping; this host-pointer form is one upload, the device stage below, one download.
 * Outputs: h_depth / h_key_group [sum n_keys] in mvKeys order (camera-major): depth in the key's own camera
 * (-1: none) and mapcamidx2idxs_ (-1: none); the groups mvidxsMatches (h_group_idx [n_groups][n_cams], -1:
 * none), goodmatches_, v3dpoints_ (reference-camera frame).  VIEO_E_CAPACITY when more than group_capacity
 * groups form. */
typedef struct vieo_fisheye_params {
  int32_t n_cams;   /* 2..4 */
  int32_t n_levels; /* entries of level_sigma2 */
  float bf;         /* stereoinfo_.baseline_bf_[1] */
  float th_far_pts; /* <= 0: the two parallax thresholds stay 0.9998 and 1 - 1e-6 */
  const struct vieo_camera* cams; /* model + float parameters of mpCameras[i] (Rcb / tcb unused here) */
  const double* Trc;         /* [n_cams][12] row-major 3x4 of mpCameras[i]->GetTrc().cast<double>() */
  const double* Tcr;         /* [n_cams][12] row-major 3x4 of mpCameras[i]->GetTcr().cast<double>() */
  const float* level_sigma2; /* scalepyrinfo_.vlevelsigma2_ */
} vieo_fisheye_params;

int vieo_stereo_fisheye_match(const vieo_fisheye_params* params, const vieo_keypoint* const* h_keys /*[n_cams]*/,
                              const uint8_t* const* h_descriptors /*[n_cams] rows of 32 bytes*/,
                              const int32_t* n_keys /*[n_cams]*/, const int32_t* num_mono /*[n_cams]*/,
                              int32_t group_capacity, float* h_depth, int32_t* h_key_group,
                              int32_t* h_group_idx, uint8_t* h_group_good, double* h_group_p3d,
                              int32_t* n_groups, int32_t* n_matches);

/* Device-resident form (round 4): the whole stage -- knn-2 of every camera pair, ratio test + pair triangulation, the
 * group tables of FillMatchesFromPair, the all-camera re-triangulation, mvKeys / mDescriptors / vdepth_ -- as five
 * launches on `stream` with no host round trip: the key counts are read on the device, and the order-dependent group
 * bookkeeping runs as a speculative-parallel walk on one wavefront per frame (rows that touch disjoint keys / groups
 * are applied 64 at a time with the sequential result; fisheye_stereo.hip).  A handle carries the rig's constants and
 * the scratch for up to max_frames frames of key_cap_per_camera (<= 8191) keys per camera.
 * Inputs are the extractor's batch arrays: d_keys / d_desc [n_frames][n_cams][key_cap_per_camera], d_counts
 * [n_frames][n_cams][2] = {n, num_mono} (vieo_orb_extract_batch_device with n_images = n_frames * n_cams).
 * Outputs, per frame: d_keys_cat / d_desc_cat / d_depth / d_uright / d_key_group [n_cams * key_cap_per_camera] in mvKeys
 * (camera-major) order, d_cam_first [n_cams + 1], d_frame_counts [2] = {N, 0} (the counts layout the vieo_track_* glue
 * reads); the groups d_group_idx [gcap][n_cams], d_group_good [gcap],
 * d_group_p3d [gcap][3] with gcap = vieo_fisheye_group_capacity(); d_hdr [8] = {n_groups, n_matches, threshold used
 * (0 / 1), status (1: more than gcap groups, the frame's tables are void), -, rows walked, wavefront steps, -}. */
typedef struct vieo_fisheye vieo_fisheye;
int vieo_fisheye_create(vieo_fisheye** out, const vieo_fisheye_params* params, int key_cap_per_camera, int max_frames);
void vieo_fisheye_destroy(vieo_fisheye* h);
int vieo_fisheye_group_capacity(const vieo_fisheye* h);
int vieo_stereo_fisheye_match_batch_device(vieo_fisheye* h, const vieo_keypoint* d_keys, const uint8_t* d_desc,
                                           const int32_t* d_counts, int n_frames, vieo_keypoint* d_keys_cat,
                                           uint8_t* d_desc_cat, int32_t* d_cam_first, int32_t* d_frame_counts,
                                           float* d_depth, float* d_uright, int32_t* d_key_group, int32_t* d_group_idx, uint8_t* d_group_good,
                                           double* d_group_p3d, int32_t* d_hdr, void* stream);
/* The same stage in two parts, for a caller that lets the tracking of a frame run beside its stereo bookkeeping: what
 * tracking reads of a rig frame -- the concatenated keys / descriptors, the cameras' ranges, uright = -1 -- does not
 * depend on the matches (VIEO_FISHEYE_CONCAT, one short kernel); the matches, the groups and the keys' depths
 * (VIEO_FISHEYE_GROUPS) are outputs of the frame only and may run on another stream.  CONCAT + GROUPS = ALL. */
#define VIEO_FISHEYE_ALL 0
#define VIEO_FISHEYE_CONCAT 1
#define VIEO_FISHEYE_GROUPS 2
int vieo_stereo_fisheye_match_batch_device_part(vieo_fisheye* h, const vieo_keypoint* d_keys, const uint8_t* d_desc,
                                           const int32_t* d_counts, int n_frames, vieo_keypoint* d_keys_cat,
                                           uint8_t* d_desc_cat, int32_t* d_cam_first, int32_t* d_frame_counts,
                                           float* d_depth, float* d_uright, int32_t* d_key_group, int32_t* d_group_idx, uint8_t* d_group_good,
                                           double* d_group_p3d, int32_t* d_hdr, int part, void* stream);
/* test tap: rows walked / wavefront steps of this thread's last vieo_stereo_fisheye_match */
void vieo_fisheye_last_walk(int32_t* rows, int32_t* steps);

/* int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, vMatchedPairs, bOnlyStereo)
 * (src/ORBmatcher.cc:896-1150; LocalMapping::CreateNewMapPoints, LocalMapping.cc:709): per shared vocabulary node,
 * every unmatched key of pKF1 against the unmatched keys of pKF2 -- Hamming <= TH_LOW (50), best per image of pKF2,
 * the epipole gate for two monocular keys, the epipolar constraint (GeometricCamera::epipolarConstrain,
 * camera_base.h:287-406, fundamental-matrix branch, 3.84 sigma2; distorted keys are un-projected first) -- then
 * FillMatchesFromPair (USE_STRATEGY_MIN_DIST) and the rotation-histogram filter.
 *   device: all gates of all (key1, key2) pairs of the shared nodes, for a batch of pKF2 at once;
 *   host (inside the library): the order-dependent part (skip keys already taken, best distance, group tables).
 * mFeatVec is the caller's (DBoW2 transform needs the vocabulary): nodes ascending, CSR over feature indices.
 * Both key frames of a pair are of one kind: n_cams == 0 (usedistort_ false: one undistorted pinhole camera, keys =
 * mvKeysUn) or n_cams 1..4 (a distorted rig: keys = mvKeys, camera-major). */
typedef struct vieo_tri_keyframe {
  double Tcw[12];              /* GetTcw() of the reference camera: row-major 3x4 */
  float fx, fy, cx, cy;        /* mpCameras[0] (pinhole), n_cams == 0 */
  int32_t n_keys, n_nodes;
  const vieo_keypoint* keys;   /* mvKeysUn / mvKeys (pt, angle, octave) */
  const uint8_t* descriptors;  /* mDescriptors, 32 bytes per key */
  const float* uright;         /* stereoinfo_.vuright_ (< 0: monocular key) */
  const uint8_t* has_mappoint; /* GetMapPoint(idx) != NULL */
  const uint32_t* node_id;     /* mFeatVec: node ids, ascending */
  const int32_t* node_first;   /* [n_nodes + 1] offsets into node_feat */
  const int32_t* node_feat;    /* feature indices of each node, in the vector's order */
  const float* scale_factor;   /* scalepyrinfo_.vscalefactor_ */
  const float* level_sigma2;   /* scalepyrinfo_.vlevelsigma2_ */
  int32_t n_levels, n_cams;
  const struct vieo_camera* cams; /* [n_cams] model + parameters of mpCameras[i] (Rcb / tcb unused here) */
  const double* Tcr;           /* [n_cams][12] mpCameras[i]->GetTcr() as row-major 3x4 */
  const double* Trc;           /* [n_cams][12] mpCameras[i]->GetTrc() */
  const uint8_t* key_cam;      /* [n_keys] get<0>(mapn2in_[idx]): the camera of each key */
} vieo_tri_keyframe;           /* 232 bytes */
/* kf2s[n_kf2]: the neighbours of kf1, each an independent call of the reference.  Outputs per neighbour p:
 * h_pairs[(p * pair_capacity + m) * pair_stride + c] = the key of camera c in match m of vMatchedPairs (its
 * order; cameras of pKF1 first, then those of pKF2; -1: none; pair_stride >= the number of cameras of the pair,
 * 2 for undistorted key frames), h_n_pairs[p] their count, h_n_matches[p] the reference's return value.
 * VIEO_E_CAPACITY when a neighbour has more matches than pair_capacity. */
int vieo_search_for_triangulation(const vieo_tri_keyframe* kf1, const vieo_tri_keyframe* kf2s, int n_kf2,
                                  int only_stereo, int check_orientation, int32_t pair_capacity, int32_t pair_stride,
                                  int32_t* h_pairs, int32_t* h_n_pairs, int32_t* h_n_matches);

/* void Frame::ComputeStereoMatches() (src/Frame.cc:451-611), rectified stereo: row-band Hamming
 * search (octave +-1, disparity window [0, bf/baseline]), 11 SADs of 11x11 patches on the
 * left key's pyramid level, parabola sub-pixel fit, rejection above 1.5*1.4*median SAD.
 * Outputs stereoinfo_.vuright_ / vdepth_ (-1 = no match).  `left`/`right` are the two
 * extractors that just processed the frame's images (their mvImagePyramid is read in place on
 * the device); `baseline` = stereoinfo_.baseline_bf_[0], `bf` = baseline_bf_[1]. */
int vieo_stereo_match_rectified(vieo_orb* left, vieo_orb* right, const vieo_keypoint* h_kpL,
                                const uint8_t* h_descL, int nL, const vieo_keypoint* h_kpR,
                                const uint8_t* h_descR, int nR, float baseline, float bf,
                                float* h_uright, float* h_depth);
/* Batched device form: the extractor's last batch holds 2*n_frames images, image 2f = left and
 * 2f+1 = right camera of frame f; keypoints/descriptors/counts are that batch's outputs.
 * d_uright / d_depth are [n_frames][capacity].  Asynchronous on the extractor's stream. */
int vieo_stereo_match_rectified_batch_device(vieo_orb* e, int n_frames,
                                             const vieo_keypoint* d_keypoints,
                                             const uint8_t* d_descriptors, const int32_t* d_counts,
                                             int capacity, float baseline, float bf,
                                             float* d_uright, float* d_depth);

/* camm::Camera of one physical camera as an EdgeReproject sees it (a20: common/camera_models/
 * camera_pinhole.h:70-106, camera_radtan.h:61-129, camera_kb8.h:68-157), with EdgeReproject::SetParams
 * (g2otypes.h:409-416) already applied to the extrinsics: Rcb = Rccr * Rcrb, tcb = Rccr * tcrb + tcr. */
#define VIEO_CAM_PINHOLE 0
#define VIEO_CAM_RADTAN 1
#define VIEO_CAM_KB8 2
typedef struct vieo_camera {
  int32_t model;   /* VIEO_CAM_* */
  int32_t num_k;   /* Radtan: number of radial coefficients (parameters.size() - 6), else unused */
  float fx, fy, cx, cy;
  float dist[8];   /* Radtan: k1..k_num_k, p1, p2;  KB8: k1..k4 */
  double Rcb[9], tcb[3];
} vieo_camera;

/* ---------------------------------------------------------------- projection search ---------
 * Replaces the tracking-side ORBmatcher::SearchByProjection overloads (src/ORBmatcher.cc:230-335
 * and :1303-1467) on flattened inputs.  The window query is FrameBase::GetFeaturesInArea
 * (src/FrameBase.cpp:95-141, 64x48 grid order: cell column, cell row, key index).
 */
typedef struct vieo_proj_query { /* one map point projected into one camera: 64 bytes */
  float u, v, ur;                /* projection; ur is compared with stereo keypoints only */
  float radius;                  /* window half-size in pixels (th * scale already applied) */
  int32_t level_min, level_max;  /* GetFeaturesInArea(minlevel, maxlevel); level_max < 0: no max */
  float angle;                   /* angle of the query's own keypoint (rotation check, a12) */
  int32_t flags;                 /* bit0: valid; bit1: the map point has Observations() > 0;
                                  * bits 8..11: camera of a rig frame (0 for the single-camera entries) */
  uint8_t desc[32];              /* MapPoint::GetDescriptor() */
} vieo_proj_query;

typedef struct vieo_last_frame_point { /* LastFrame.mvpMapPoints[i] flattened: 64 bytes */
  float Xw[3];       /* MapPoint::GetWorldPos() */
  int32_t octave;    /* LastFrame.mvKeys[i].octave */
  float angle;       /* LastFrame.mvKeys[i].angle */
  int32_t flags;     /* bit0: pMP != NULL && !LastFrame.mvbOutlier[i]; bit1: Observations() > 0 */
  int32_t reserved[2]; /* [0]: read by the frame tracker only -- 1 + the index of the FIRST key of LastFrame that holds the
                        * same MapPoint (a rig frame's point is held by one key per camera), 0: this key */
  uint8_t desc[32];
} vieo_last_frame_point;

typedef struct vieo_sbp_camera {
  double Tcw_cur[12], Tcw_last[12]; /* 3x4 row-major [R|t] of CurrentFrame / LastFrame */
  float fx, fy, cx, cy;
  float bounds[4];                  /* gridinfo_.minmax_xy_: min_x, max_x, min_y, max_y */
  float bf, baseline;               /* stereoinfo_.baseline_bf_[1], [0] */
  float th, th_far;                 /* window threshold; far-point cut (<= 0: off) */
  int32_t mono, nlevels;
  float scale[16];                  /* scalepyrinfo_.vscalefactor_ */
} vieo_sbp_camera;

#define VIEO_SBP_LAST_FRAME 0 /* accept best <= TH_HIGH, rotation histogram (ORBmatcher.cc:1303-1467) */
#define VIEO_SBP_LOCAL_MAP 1  /* best/second ratio test when same level (ORBmatcher.cc:230-335) */
#define VIEO_SBP_RELOC 2      /* a14 ORBmatcher.cc:1471-1606: SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th,
                                * ORBdist): keys holding ANY map point are skipped, no stereo gate, best only,
                                * accepted if dist <= ORBdist (passed in nn_ratio), rotation histogram as mode 0;
                                * queries = the key frame's map points (not in sAlreadyFound) projected with
                                * levels [L-1, L+1], radius th*scale[L], angle = pKF->mvKeys[i].angle */
#define VIEO_SBP_UNCHANGED (-1)
#define VIEO_SBP_ERASED (-2)

/* Projection part of SearchByProjection(Frame&, const Frame&, th, bMono, th_far)
 * (ORBmatcher.cc:1313-1378): last frame's valid map points -> window queries. */
int vieo_sbp_project_last_frame(const vieo_last_frame_point* h_points, int n,
                                const vieo_sbp_camera* h_cam, vieo_proj_query* h_queries);
/* The search + sequential assignment of both overloads.  h_taken[i] != 0 when keypoint i already
 * holds a map point with Observations() > 0.  h_assign[n_keys]: VIEO_SBP_UNCHANGED, VIEO_SBP_ERASED
 * (EraseMapPointMatch by the rotation check) or the index of the query whose map point
 * AddMapPoint() put there.  *nmatches = the reference's return value. */
int vieo_search_by_projection(int mode, const vieo_proj_query* h_queries, int nq,
                              const vieo_keypoint* h_keys, const float* h_uright,
                              const uint8_t* h_desc, const uint8_t* h_taken, int n_keys,
                              const float* h_bounds /*[4]*/, float nn_ratio, int check_orientation,
                              int32_t* h_assign, int32_t* nmatches);
/* Batched device form: frame f owns queries [f*q_cap, f*q_cap + d_nq[f]) and the key arrays of
 * image img_first + f*img_step of an extractor batch (d_counts as written by the extractor).
 * The search entries keep their scratch (candidate pool, cursors, grid CSRs) per HOST THREAD, not per stream: a thread
 * issues its searches on ONE stream, or synchronises between searches it issues on different streams. */
int vieo_search_by_projection_batch_device(int mode, const vieo_proj_query* d_queries,
                                           const int32_t* d_nq, int q_cap, int n_frames,
                                           const vieo_keypoint* d_keys, const float* d_uright,
                                           const uint8_t* d_desc, const uint8_t* d_taken,
                                           const int32_t* d_counts, int key_cap, int img_first,
                                           int img_step, const float* h_bounds, float nn_ratio,
                                           int check_orientation, int32_t* d_assign,
                                           int32_t* d_nmatches, void* stream);
int vieo_sbp_project_last_frame_batch_device(const vieo_last_frame_point* d_points,
                                             const int32_t* d_n, int p_cap, int n_frames,
                                             const vieo_sbp_camera* d_cams,
                                             vieo_proj_query* d_queries, void* stream);

/* ---- the same searches for frames of a camera rig (Frame::mpCameras.size() cameras, distorted or not) -------
 * The reference loops the cameras inside every overload: SearchByProjection(Frame&, const Frame&) projects each
 * last-frame point into every camera camj with mpCameras[camj]->GetTcr() and, when Frame::usedistort_, the camera
 * model's Project() (ORBmatcher.cc:1339-1366); the local-map overload walks the point's vtrack_cami_ list
 * (:257-266); the relocalisation overload loops cami (:1491-1543).  Every (point, camera) pair is one query, in
 * point-major order -- that is the order in which keys are claimed -- and its camera travels in bits 8..11 of
 * vieo_proj_query.flags.  The window query is GetFeaturesInArea(cami, ...) on that camera's 64 x 48 grid
 * (FrameBase.cpp:95-141); the frame's keys are mvKeys in camera-major order (Frame.cc:738-764), camera c owning
 * the keys [cam_first[c], cam_first[c + 1]).  One rotation histogram is shared by all cameras. */
typedef struct vieo_sbp_rig {
  int32_t n_cams;      /* 1..4 */
  int32_t use_distort; /* Frame::usedistort_: 1 -> mpCameras[c]->Project(), 0 -> K * (x/z, y/z, 1) in float */
  vieo_camera cams[4]; /* model + float parameters of mpCameras[c] (Rcb / tcb unused here) */
  double Tcr[4][12];   /* mpCameras[c]->GetTcr().cast<double>(), row-major 3x4 (identity for the reference camera) */
  double trc[4][3];    /* mpCameras[c]->GetTrc().translation().cast<double>() (camera centre, relocalisation variant) */
  float bounds[4][4];  /* gridinfo_.minmax_xy_[c]: min_x, max_x, min_y, max_y */
} vieo_sbp_rig;        /* 1160 bytes */

/* pKF->GetMapPointMatches()[i] of the relocalisation overload, flattened: 64 bytes, the layout of
 * vieo_last_frame_point with the two distances in place of `reserved`. */
typedef struct vieo_keyframe_point {
  float Xw[3];       /* MapPoint::GetWorldPos() */
  int32_t octave;    /* unused (the level is predicted from the distance) */
  float angle;       /* pKF->mvKeys[i].angle */
  int32_t flags;     /* bit0: pMP != NULL && !isBad() && not in sAlreadyFound; bit1: Observations() > 0 */
  float max_distance, min_distance; /* mfMaxDistance, mfMinDistance (the 1.2 / 0.8 factors are applied here) */
  uint8_t desc[32];
} vieo_keyframe_point;

/* ORBmatcher.cc:1313-1378 with the camera loop: h_queries[i * n_cams + camj].  h_cam supplies the poses, bf,
 * baseline, th, th_far, mono and the scale factors; its fx..cy / bounds are not read (h_rig has them). */
int vieo_sbp_project_last_frame_rig(const vieo_last_frame_point* h_points, int n, const vieo_sbp_camera* h_cam,
                                    const vieo_sbp_rig* h_rig, vieo_proj_query* h_queries /*[n * n_cams]*/);
/* Projection part of SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist, th_far)
 * (ORBmatcher.cc:1487-1543): Tcr, Project / K, IsInImage, the scale-invariance distances, MapPoint::PredictScale,
 * window th * scale[level], levels [L-1, L+1].  h_cam: Tcw_cur, th, th_far, nlevels, scale;
 * log_scale_factor = scalepyrinfo_.flogscalefactor_.  h_rig == NULL: the one rectified camera of h_cam. */
int vieo_sbp_project_keyframe(const vieo_keyframe_point* h_points, int n, const vieo_sbp_camera* h_cam,
                              const vieo_sbp_rig* h_rig, float log_scale_factor,
                              vieo_proj_query* h_queries /*[n * n_cams]*/);
/* vieo_search_by_projection on the camera-major key list of a rig frame.  h_bounds: [n_cams][4]. */
int vieo_search_by_projection_rig(int mode, const vieo_proj_query* h_queries, int nq, const vieo_keypoint* h_keys,
                                  const float* h_uright, const uint8_t* h_desc, const uint8_t* h_taken, int n_keys,
                                  const int32_t* h_cam_first /*[n_cams + 1]*/, const float* h_bounds, int n_cams,
                                  float nn_ratio, int check_orientation, int32_t* h_assign, int32_t* nmatches);
/* Batched device forms.  Frame f owns the queries [f * q_cap, f * q_cap + d_nq[f]) and the key arrays
 * [f * key_cap, f * key_cap + d_cam_first[f * (n_cams + 1) + n_cams]) (already concatenated camera-major);
 * d_rigs: one vieo_sbp_rig per frame; h_bounds [n_cams][4] shared by all frames. */
int vieo_sbp_project_last_frame_rig_batch_device(const vieo_last_frame_point* d_points, const int32_t* d_n,
                                                 int p_cap, int n_frames, const vieo_sbp_camera* d_cams,
                                                 const vieo_sbp_rig* d_rigs, int n_cams,
                                                 vieo_proj_query* d_queries /*[n_frames][p_cap * n_cams]*/,
                                                 void* stream);
int vieo_search_by_projection_rig_batch_device(int mode, const vieo_proj_query* d_queries, const int32_t* d_nq,
                                               int q_cap, int n_frames, const vieo_keypoint* d_keys,
                                               const float* d_uright, const uint8_t* d_desc,
                                               const uint8_t* d_taken, const int32_t* d_cam_first, int key_cap,
                                               const float* h_bounds, int n_cams, float nn_ratio,
                                               int check_orientation, int32_t* d_assign, int32_t* d_nmatches,
                                               void* stream);

/* The window grid of a search (Frame::mGrid as a CSR, built on the device by every *_batch_device search call) depends on
 * the frame's keys only.  vieo_sbp_keep_grid(1) tells the NEXT search call of this host thread that its key arrays hold what
 * the previous call's held (same device pointers, geometry and stream are checked, the contents are the caller's promise):
 * the grid is then not rebuilt -- TrackLocalMap's search after TrackWithIMU's in the one-call tracker.  One call only. */
int vieo_sbp_keep_grid(int on);

/* ---- the resident frame: the drop-in path's fast form (round 5) -------------------------------------------------
 * The reference's callers reach this library one class member at a time -- Frame::Frame -> ExtractORB per camera thread
 * (src/Frame.cc:259-278) -> ComputeStereoMatches (:451-611) -> ORBmatcher::SearchByProjection (src/Tracking.cc:296, :2599)
 * -> Optimizer::PoseOptimization (:321,333) -- and every member's signature hands over host objects.  Nothing in those
 * signatures says the data must travel: vieo_orb_extract LEAVES the image's keys, descriptors and pyramid in HBM, in the
 * extractor handle the Frame already points at (mpORBextractors[c]), and the entries below read them there.  A call
 * uploads only what the pointer graph forces (the last frame's map points, the window queries, the taken flags), runs on
 * the handle's stream and returns behind ONE synchronisation; the frame's window grid (Frame::mGrid as a CSR) is built by
 * its first search and kept for the later ones.  The shim decides per call with vieo_orb_holds whether the handle still
 * holds the frame it was given (a Frame copied long ago does not: the host-pointer entries above serve it).
 * Results are those of the host-pointer entries bit for bit (same kernels, same inputs). */
/* 1 when the handle's resident frame is the one these keys describe (count + first / last eight keys compared with what
 * the last vieo_orb_extract of this handle returned), else 0. */
int vieo_orb_holds(const vieo_orb* e, const vieo_keypoint* h_keys, int n_keys);
/* number of keys of the resident frame, -1: none */
int vieo_orb_resident_keys(const vieo_orb* e);
/* Frame::ComputeStereoMatches (src/Frame.cc:451-611) of the frame the two handles have just extracted.
 * h_uright / h_depth [vieo_orb_resident_keys(left)]; uright also stays in the left handle for the searches. */
int vieo_stereo_match_rectified_resident(vieo_orb* left, vieo_orb* right, float baseline, float bf,
                                         float* h_uright, float* h_depth);
/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono, th_far_pts)
 * (src/ORBmatcher.cc:1303-1467) as one call: vieo_sbp_project_last_frame + vieo_search_by_projection(VIEO_SBP_LAST_FRAME)
 * on the resident keys.  h_uright: NULL = the resident values of vieo_stereo_match_rectified_resident, else
 * stereoinfo_.vuright_ of the frame.  h_assign [vieo_orb_resident_keys(frame)] as in vieo_search_by_projection. */
int vieo_search_by_projection_last_frame_resident(vieo_orb* frame, const vieo_last_frame_point* h_points, int n_points,
                                                  const vieo_sbp_camera* h_cam, const float* h_uright, float nn_ratio,
                                                  int check_orientation, int32_t* h_assign, int32_t* nmatches);
/* vieo_search_by_projection on the resident keys: the local-map overload (src/ORBmatcher.cc:230-335, queries from
 * MapPoint::GetTrackInfoRef()) and the relocalisation overload (queries from vieo_sbp_project_keyframe). */
int vieo_search_by_projection_resident(int mode, vieo_orb* frame, const vieo_proj_query* h_queries, int nq,
                                       const float* h_uright, const uint8_t* h_taken, const float* h_bounds /*[4]*/,
                                       float nn_ratio, int check_orientation, int32_t* h_assign, int32_t* nmatches);

/* ---------------------------------------------------------------- pose optimisation --------
 * Replaces Optimizer::PoseOptimization (motion-only BA with fixed map points).  The host shim
 * flattens Frame / MapPoint objects into the POD structs below and writes the results back
 * (mvbOutlier, NavState, pose) exactly where the reference does (Optimizer.cc:1716,1824-1871).
 */

/* VIEO_SLAM::NavState (src/Odom/NavState.h:18-85): 22 doubles. */
typedef struct vieo_navstate {
  double p[3];  /* mpwb */
  double q[4];  /* mRwb as unit quaternion (w, x, y, z) */
  double v[3];  /* mvwb */
  double bg[3], ba[3], dbg[3], dba[3];
} vieo_navstate;


/* One 3D-2D correspondence = one EdgeReprojectPR / PRStereo (src/Odom/g2otypes.h:321-547). */
typedef struct vieo_pose_obs {
  float Xw[3];      /* MapPoint::mWorldPos, stored as float32 (include/MapPoint.h:52-53) */
  float u, v, ur;   /* mvKeysUn[i].pt and stereoinfo_.vuright_[i]; ur < 0 => monocular edge */
  float inv_sigma2; /* scalepyrinfo_.vinvlevelsigma2_[octave] */
  int32_t flags;    /* bit 0: close point (track_depth_ < max(10, ThDepth)), VIO variant only */
} vieo_pose_obs;    /* 32 bytes */

typedef struct vieo_enc_preint { /* EncPreIntegrator (src/Odom/OdomPreIntegrator.h:66-100) */
  double dt;          /* mdeltatij; 0 => no encoder edge */
  double delx[6];     /* mdelxEij: delta~Phi_ij (3), delta~p_ij (3) */
  double Sigma[36];   /* mSigmaEij, row-major */
} vieo_enc_preint;    /* 344 bytes */
/* The optional encoder edge of the pose optimisations (EdgeEncNavStatePR / PVR between the last frame and the
 * current one, Optimizer.cc:1650-1674, Optimizer.h:345-372): measurement, extrinsics, and the last frame's pose. */
typedef struct vieo_pose_enc {
  vieo_enc_preint enc;     /* pFrame->GetEncPreInt() */
  double qRbe[4], pbe[3];  /* Tbe = Frame::mTbc * Frame::mTce: rotation (w, x, y, z), translation */
  double p_last[3], q_last[4]; /* pLastF->GetNavStateRef(): mpwb, mRwb (w, x, y, z) -- the fixed vertex of a15 */
} vieo_pose_enc;           /* 456 bytes */

typedef struct vieo_pose_frame {
  vieo_navstate nav;     /* initial estimate (Frame::mNavState after UpdateNavStatePVRFromTcw) */
  double Rcb[9], tcb[3]; /* FrameBase::meigRcb (row-major) / meigtcb */
  float fx, fy, cx, cy;  /* pinhole intrinsics, float like camm::Camera::parameters_ */
  float bf;              /* stereoinfo_.baseline_bf_[1] */
  int32_t obs_begin;     /* first observation of this frame in the flat obs array */
  int32_t n_obs;
  int32_t n_cams;        /* 0: the rectified pinhole camera above; 1..4: `cams` (a20, Frame::usedistort_), and
                          * bits 8..11 of vieo_pose_obs.flags select the observation's camera (monocular edges) */
  const vieo_camera* cams; /* host pointer for the host entry points, device pointer for *_batch_device */
  const vieo_pose_enc* enc; /* NULL or enc->enc.dt == 0: no encoder edge; host / device pointer like `cams` */
} vieo_pose_frame;

/* The batched device forms cannot see n_cams from the host, so by default they launch both kernel instances
 * (rectified pinhole frames, camera-rig frames); each instance skips the other's frames.  A caller whose
 * frames are all of one kind says so once and saves the empty launch; a frame of the other kind then
 * comes back with status VIEO_E_INVALID instead of being optimised. */
#define VIEO_POSE_CAMS_AUTO 0
#define VIEO_POSE_CAMS_RECTIFIED 1 /* every frame has n_cams == 0 (Frame::usedistort_ false) */
#define VIEO_POSE_CAMS_RIG 2       /* every frame has n_cams > 0 */
int vieo_pose_set_camera_mode(int mode);
/* The same for the encoder edge of vieo_pose_optimization_vio_batch_device (EdgeEncNavStatePVR, Optimizer.h:345-363):
 * frames with base.enc->enc.dt != 0 run in their own kernel instance.  (The vision-only kernel branches at run
 * time and ignores this mode.) */
#define VIEO_POSE_ENC_AUTO 0
#define VIEO_POSE_ENC_NONE 1 /* no frame carries an encoder measurement (Frame::GetEncPreInt().mdeltatij == 0) */
#define VIEO_POSE_ENC_ALL 2  /* every frame carries one */
int vieo_pose_set_encoder_mode(int mode);

#define VIEO_POSE_OK 0
#define VIEO_POSE_TOO_FEW 1 /* < 3 correspondences: the reference returns 0 and leaves the pose */
typedef struct vieo_pose_result {
  vieo_navstate nav;  /* optimised state (p, q updated; other fields copied) */
  int32_t n_inliers;  /* return value of the reference: nInitialCorrespondences - nBad */
  int32_t status;     /* VIEO_POSE_* */
  int32_t lm_iterations; /* total LM iterations executed over the 4 rounds (diagnostic) */
  int32_t reserved;   /* visual-inertial variant: total lambda trials over the 4 rounds (diagnostic); else 0 */
} vieo_pose_result;

/* int Optimizer::PoseOptimization(Frame* pFrame, Frame* pLastF = NULL) (src/Optimizer.cc:1611-1874),
 * vision-only motion BA (the optional encoder edge to the last frame: `enc`): 4 rounds of optimize(10) from the
 * initial estimate, chi2 classification 5.991 / 7.815 after each round, Huber off after round 3.
 * h_outlier[n_obs] receives mvbOutlier of the matched keypoints. */
int vieo_pose_optimization(const vieo_pose_frame* h_frame, const vieo_pose_obs* h_obs,
                           uint8_t* h_outlier, vieo_pose_result* h_result);
/* Batched device form: one workgroup per frame, the whole 4x10 LM loop runs on the device.
 * d_obs / d_outlier are flat arrays indexed by obs_begin + i.  Asynchronous on `stream`. */
int vieo_pose_optimization_batch_device(const vieo_pose_frame* d_frames, int n_frames,
                                        const vieo_pose_obs* d_obs, uint8_t* d_outlier,
                                        vieo_pose_result* d_results, void* stream);

/* ---- visual-inertial variant --------------------------------------------------------------
 * template<class KeyFrame> int Optimizer::PoseOptimization(Frame*, KeyFrame* pLastKF,
 *     const cv::Mat& gw, bool bComputeMarg = false, bool bNoMPs = false)
 * (include/Optimizer.h:208-816), instantiated for Frame and KeyFrame by Tracking.cc:321,333,475,479.
 * Vertices: PVR_j (9) + Bias_j (6) free; PVR_i + Bias_i of the last (key)frame fixed unless it
 * carries a prior (mbPrior).  Edges: EdgeNavStatePVR (IMU pre-integration, g2otypes.h:703-884),
 * EdgeNavStateBias (g2otypes.cpp:14-34), EdgeNavStatePriorPVRBias (g2otypes.cpp:84-124),
 * EdgeReprojectPVR / PVRStereo per correspondence, and EdgeEncNavStatePVR (g2otypes.h:591-668, Optimizer.h:345-363)
 * when base.enc carries a measurement: between PVR_i and PVR_j (base.enc->p_last / q_last are not read here, the
 * last state is nav_last), Huber sqrt(12.592), information Sigma_E^-1; it also enters the marginal prior
 * (FillCovInv, Optimizer.h:195-204) and counts as the odometry edge that keeps the estimate between the rounds. */
typedef struct vieo_imu_preint { /* IMUPreIntegratorBase (src/Odom/OdomPreIntegrator.h:108-147) */
  double dt;                     /* mdeltatij; 0 => no IMU edge */
  double Rij[9];                 /* mRij, row-major */
  double vij[3], pij[3];         /* mvij, mpij */
  double JgR[9], Jgv[9], Jav[9], Jgp[9], Jap[9]; /* bias Jacobians, row-major */
  double Sigma[81];              /* mSigmaij, order (p, v, Phi), row-major */
} vieo_imu_preint;

typedef struct vieo_vio_frame {
  vieo_pose_frame base;     /* current frame: nav = nsj, extrinsics, camera, observations */
  vieo_navstate nav_last;   /* pLastKF->GetNavState() */
  vieo_navstate nav_prior;  /* pLastKF->mNavStatePrior   (used when last_has_prior) */
  double H_prior[225];      /* pLastKF->mMargCovInv, 15x15 row-major, order (p, v, Phi, bg, ba) */
  vieo_imu_preint imu;      /* pFrame->GetIMUPreInt() */
  double gw[3];             /* gravity in the world frame */
  double inv_sigma_bg2, inv_sigma_ba2; /* IMUDataBase::mInvSigmabg2 / mInvSigmaba2 */
  double dt_frames;         /* pFrame->ftimestamp_ - pLastKF->ftimestamp_ (used when imu.dt == 0) */
  float th_depth;           /* pFrame->mThDepth: close points use the 1.5x chi2 gate */
  int32_t last_has_prior;   /* pLastKF->mbPrior: last state is optimised too (30-dim system) */
  int32_t compute_marg;     /* bComputeMarg */
  int32_t no_mps;           /* bNoMPs */
} vieo_vio_frame;

typedef struct vieo_vio_result {
  vieo_pose_result base;   /* nav: p, q, v, dbg, dba updated */
  double H_marg[225];      /* pFrame->mMargCovInv when compute_marg (nav is then mNavStatePrior) */
  int32_t has_marg;        /* pFrame->mbPrior set */
  int32_t reserved;
} vieo_vio_result;

int vieo_pose_optimization_vio(const vieo_vio_frame* h_frame, const vieo_pose_obs* h_obs,
                               uint8_t* h_outlier, vieo_vio_result* h_result);
/* Rig frames (n_cams > 0) of a call with at most 4 frames -- the one-call tracker's case -- are optimised by 16
 * workgroups each: replicas that run the same optimisation and share the passes over the visual edges (thousands per
 * rig frame), exchanging partial sums through device memory; a frame below 700 edges is left to one of them.  The
 * result differs from the one-workgroup form only in the association order of those sums (1e-12 relative on the
 * pose); the status VIEO_E_HIP on a frame means a replica never arrived (never on a healthy device).
 * vieo_pose_set_replicas(0) keeps one workgroup per frame on this host thread (measurements; also the environment
 * variable VIEO_POSE_REPLICAS=0 at start-up), (1) restores the default.  Returns the previous setting. */
int vieo_pose_set_replicas(int on);
/* Capacity: a frame's edges are one bit per edge in a 64-bit mask per lane.  The host entry above always runs the
 * 256-thread instance (16384 observations per frame).  The batched device entry picks its instance from the batch
 * size -- more than 256 non-rig frames per call run one wavefront per frame (4096 observations per frame), smaller
 * batches and rig frames the 256-thread instance (16384) -- and reports a frame beyond the limit through that
 * frame's status (VIEO_E_CAPACITY), not through the return value. */
int vieo_pose_optimization_vio_batch_device(const vieo_vio_frame* d_frames, int n_frames,
                                            const vieo_pose_obs* d_obs, uint8_t* d_outlier,
                                            vieo_vio_result* d_results, void* stream);

/* ---------------------------------------------------------------- local bundle adjustment ---
 * void Optimizer::LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int Nlocal)
 * (src/Optimizer.cc:1876-2307), vision-only LBA without encoder edges.  The host shim collects the
 * window exactly as the reference does (:1880-1953) and flattens it:
 *   key frames  : local (free unless nid_ == 0) first, then the fixed observers;
 *   map points  : lLocalMapPoints order; XYZ float32 as stored by MapPoint;
 *   observations: in the reference's edge insertion order = per map point, per observing key frame
 *                 (so a point's edges are contiguous: obs must be sorted by mp).
 * BlockSolver_6_3 semantics: points are marginalised (Schur complement, block_solver.hpp:353-486),
 * LM as in g2o, optimize(its0=5) -> chi2 / depth outlier levels, Huber off -> optimize(its1=10).
 * Outputs: optimised NavStates of the free key frames, float32 points, and h_erase[n_obs] = 1 for
 * the observations the reference puts in vToErase (chi2 > 5.991 / 7.815 or non-positive depth). */
typedef struct vieo_lba_keyframe {
  vieo_navstate nav;
  int32_t fixed;
  int32_t reserved;
} vieo_lba_keyframe;

typedef struct vieo_lba_obs {
  int32_t kf, mp;   /* indices into the key-frame / point arrays */
  float u, v, ur;   /* ur < 0 => monocular edge */
  float inv_sigma2;
} vieo_lba_obs;

typedef struct vieo_lba_params {
  double Rcb[9], tcb[3];
  float fx, fy, cx, cy, bf;
  int32_t its0, its1;     /* 5 and 10 in the reference */
  int32_t n_cams;         /* 0: one rectified pinhole camera = the fields above (Frame::usedistort_ == false);
                           * 1..4: `cams`, and bits 24..27 of vieo_lba_obs.kf select the observation's camera
                           * (get<0>(pKFi->mapn2in_[idx])); distorted observations are monocular (ur < 0) */
  const vieo_camera* cams; /* host pointer, n_cams entries */
} vieo_lba_params;

#define VIEO_LBA_OK 0
#define VIEO_LBA_ABORTED 1       /* *stop was set (Optimizer.cc:2174-2186) */
#define VIEO_LBA_NO_FREE_POSE 2  /* Optimizer.cc:1993 */
typedef struct vieo_lba_result {
  int32_t status;
  int32_t n_erase;
  int32_t lm_iterations;
  int32_t lm_trials;
  double chi2_initial, chi2_final; /* robust chi2 at the first / after the last accepted step */
} vieo_lba_result;

/* Host-pointer entry (synchronous).  `stop` may be NULL; it is polled before the first
 * optimisation, between the two optimisations and between LM iterations. */
int vieo_local_bundle_adjustment(const vieo_lba_params* params, const vieo_lba_keyframe* h_kfs,
                                 int n_kf, const float* h_points, int n_mp,
                                 const vieo_lba_obs* h_obs, int n_obs, volatile const int* stop,
                                 vieo_navstate* h_navs_out, float* h_points_out, uint8_t* h_erase,
                                 vieo_lba_result* h_result);

/* Priority of the calling host thread's bundle-adjustment stream from its next call on: -1 lowest (the default, or
 * VIEO_LBA_PRIORITY: batches of windows beside a batched front end fill what it leaves free), 0 default priority (one
 * window beside a tracker: the LocalMapping thread of a sequential host -- examples/replay_common.hpp -- asks for this;
 * 1.03 against 1.08 ms per frame), 1 highest. */
int vieo_lba_set_stream_priority(int priority);

/* Several independent windows (one per map / per LocalMapping thread of a multi-session server)
 * advanced in lock step: every kernel launch covers all windows and the host reads one small
 * record per window and LM trial.  Each array argument has n_windows entries; per-window results
 * are identical to n_windows calls of vieo_local_bundle_adjustment.  `stop` is shared.
 * A batch advances at the pace of its slowest window: windows of very different size (ordinary windows of ten free key
 * frames and bLarge ones of twenty-five, which take another solve kernel and six Schur tiles instead of one) are better
 * given to separate calls from separate host threads -- 205 such windows: 18.9 ms as one mixed call, 13.6 ms as two. */
int vieo_local_bundle_adjustment_batch(int n_windows, const vieo_lba_params* const* params,
                                       const vieo_lba_keyframe* const* h_kfs, const int* n_kf,
                                       const float* const* h_points, const int* n_mp,
                                       const vieo_lba_obs* const* h_obs, const int* n_obs,
                                       volatile const int* stop, vieo_navstate* const* h_navs_out,
                                       float* const* h_points_out, uint8_t* const* h_erase,
                                       vieo_lba_result* h_results);

/* ---- LocalBundleAdjustmentNavStatePRV (a18) -----------------------------------------------------
 * void Optimizer::LocalBundleAdjustmentNavStatePRV(KeyFrame*, int Nlocal, bool* pbStopFlag, Map*,
 *   cv::Mat gw, bool bLarge, bool bRecInit, float th_dist_far) (src/Optimizer.cc:21-769), the local BA
 * of the visual-inertial configurations.  Per local key frame three vertices PR (6) + V (3) + Bias (6)
 * (g2otypes.h:287-288,553-567); between consecutive key frames an EdgeNavStatePRV (g2otypes.h:703-884,
 * residual order p, Phi, v) and an EdgeNavStateBias (g2otypes.cpp:14-34); the same EdgeReprojectPR /
 * PRStereo edges and marginalised points as the vision-only LBA; Chi2LargeSetLevel pre-pass
 * (optimizer_ba/g2o_graph_operator.h:23-40); user lambda init; divergence check (:660-666).
 * Key frames: local ones first, oldest to newest (= lLocalKeyFrames order), then the fixed observers;
 * the key frame before the window (pKFPrevLocal) is a fixed one whose full nav state is used.
 * Encoder edges (EdgeEncNavStatePR, g2otypes.h:591-668; Optimizer.cc:323-347): one per pair whose enc.dt != 0. */
typedef struct vieo_lba_imu_edge {
  int32_t kf_i, kf_j;  /* previous / current key frame of the pre-integration (indices) */
  double dt_kf;        /* pKF1->ftimestamp_ - pKF0->ftimestamp_, used when imu.dt == 0 */
  vieo_imu_preint imu; /* GetIMUPreInt() of kf_j; Sigma = mSigmaijPRV, order (p, Phi, v) */
  vieo_enc_preint enc; /* GetEncPreInt() of kf_j */
} vieo_lba_imu_edge;

typedef struct vieo_lba_vio_params {
  vieo_lba_params base;  /* extrinsics, camera, its0 / its1 = optit[0] / optit[1] (4 and 6; 2 and 2 if bLarge) */
  double gw[3];
  double inv_sigma_bg2, inv_sigma_ba2; /* IMUDataBase::mInvSigmabg2 / mInvSigmaba2 */
  double lambda_init;    /* setUserLambdaInit: 1e0, 1e-2 if bLarge (Optimizer.cc:131-138) */
  int32_t rec_init;      /* bRecInit: Huber kernels on the inertial edges of free key frames too */
  int32_t large;         /* bLarge: the divergence check is skipped */
  float th_dist_far;     /* th_dist_far (Optimizer.cc:395,454,513-517): a point none of whose monocular edges sees it
                          * closer than this has all its monocular edges excluded; <= 0 or inf: no such rule */
  int32_t reserved;
  double qRbe[4], pbe[3]; /* Tbe = Frame::mTbc * Frame::mTce: rotation (w, x, y, z) and translation (encoder edges) */
} vieo_lba_vio_params;

#define VIEO_LBA_DIVERGED 3 /* 2*err < err_end or NaN: returns without write-back (Optimizer.cc:660-666) */

/* h_close[n_mp]: 1 where pMP->GetTrackInfoRef().track_depth_ < thresh_depth_close (chi2 gate x1.5 for
 * monocular edges, Optimizer.cc:603-611,677-681).  h_navs_out[k] of a free key frame carries p, q, v,
 * dbg, dba of the optimised vertices (bg, ba unchanged); result.chi2_initial / chi2_final = err / err_end. */
int vieo_local_bundle_adjustment_vio(const vieo_lba_vio_params* params, const vieo_lba_keyframe* h_kfs,
                                     int n_kf, const float* h_points, const uint8_t* h_close, int n_mp,
                                     const vieo_lba_obs* h_obs, int n_obs,
                                     const vieo_lba_imu_edge* h_imu, int n_imu, volatile const int* stop,
                                     vieo_navstate* h_navs_out, float* h_points_out, uint8_t* h_erase,
                                     vieo_lba_result* h_result);

/* several windows in lock step, see vieo_local_bundle_adjustment_batch */
int vieo_local_bundle_adjustment_vio_batch(int n_windows, const vieo_lba_vio_params* const* params,
                                           const vieo_lba_keyframe* const* h_kfs, const int* n_kf,
                                           const float* const* h_points, const uint8_t* const* h_close,
                                           const int* n_mp, const vieo_lba_obs* const* h_obs, const int* n_obs,
                                           const vieo_lba_imu_edge* const* h_imu, const int* n_imu,
                                           volatile const int* stop, vieo_navstate* const* h_navs_out,
                                           float* const* h_points_out, uint8_t* const* h_erase,
                                           vieo_lba_result* h_results);

/* Encoder edges of the vision-only BAs: EdgeEncNavStatePR (g2otypes.h:591-668) between pKF1->GetPrevKeyFrame() and
 * pKF1 for every pair with GetEncPreInt().mdeltatij != 0 (LocalBundleAdjustment: Optimizer.cc:2008-2042, local key
 * frames only; BundleAdjustment with bEnc: :1401-1438).  Both key frames must be in the window (indices), the edges
 * chain the key frames (at most one into and one out of each).  Information Sigma_E^-1, x 1e-2 when kf_i is fixed;
 * Huber sqrt(12.592) -- always in the local BA (kept through both optimisations), iff `robust` in the full BA. */
typedef struct vieo_lba_enc_edge {
  int32_t kf_i, kf_j;  /* previous / current key frame (indices into the key-frame array) */
  vieo_enc_preint enc; /* GetEncPreInt() of kf_j */
} vieo_lba_enc_edge;   /* 352 bytes */
typedef struct vieo_lba_enc {
  int32_t n_edges, reserved;
  const vieo_lba_enc_edge* edges; /* host pointer */
  double qRbe[4], pbe[3];         /* Tbe = Frame::mTbc * Frame::mTce: rotation (w, x, y, z), translation */
} vieo_lba_enc;                   /* 72 bytes */
/* vieo_local_bundle_adjustment / vieo_bundle_adjustment with those edges; enc == NULL or n_edges == 0: identical to
 * the plain entries. */
int vieo_local_bundle_adjustment_enc(const vieo_lba_params* params, const vieo_lba_keyframe* h_kfs, int n_kf,
                                     const float* h_points, int n_mp, const vieo_lba_obs* h_obs, int n_obs,
                                     const vieo_lba_enc* enc, volatile const int* stop, vieo_navstate* h_navs_out,
                                     float* h_points_out, uint8_t* h_erase, vieo_lba_result* h_result);
/* lock-step batch: encs[w] may be NULL (a window without encoder edges); encs itself may be NULL */
int vieo_local_bundle_adjustment_batch_enc(int n_windows, const vieo_lba_params* const* params,
                                           const vieo_lba_keyframe* const* h_kfs, const int* n_kf,
                                           const float* const* h_points, const int* n_mp,
                                           const vieo_lba_obs* const* h_obs, const int* n_obs,
                                           const vieo_lba_enc* const* encs, volatile const int* stop,
                                           vieo_navstate* const* h_navs_out, float* const* h_points_out,
                                           uint8_t* const* h_erase, vieo_lba_result* h_results);
int vieo_bundle_adjustment_enc(const vieo_lba_params* params, int n_iterations, int robust,
                               const vieo_lba_keyframe* h_kfs, int n_kf, const float* h_points, int n_mp,
                               const vieo_lba_obs* h_obs, int n_obs, const vieo_lba_enc* enc,
                               volatile const int* stop, vieo_navstate* h_navs_out, float* h_points_out,
                               vieo_lba_result* h_result);

/* void Optimizer::BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust, bEnc = false)
 * (src/Optimizer.cc:1353-1609; GlobalBundleAdjustment :1346-1351 passes the whole map) and
 * int Optimizer::GlobalBundleAdjustmentNavStatePRV(pMap, gw, nIterations, pbStopFlag, nLoopKF, bRobust,
 * bScaleOpt = false, pimu_initiator = nullptr) (:771-1345) -- "full BA" -- on the flattened layout of the local
 * BAs: every key frame free except the flagged ones (nid_ == 0), ONE optimize(n_iterations) with g2o's own
 * initial lambda, Huber kernels (sqrt(5.99) / sqrt(7.815); sqrt(16.919) / sqrt(12.592) on the inertial / bias
 * edges) on every edge iff `robust`, no outlier classification, every point written back.  The visual-inertial
 * form weighs the inertial and bias edges leaving a fixed key frame by 1e-2 like the reference.  The reduced
 * system is factorised by one of three kernels according to its size n (lba.hip solver_class, the same for the local
 * BAs): n <= 159 k_lba_ldlt16 (one workgroup, whole triangle in LDS, blocked on the FP64 matrix cores), n <= 639
 * k_lba_ldltg (one workgroup, left-looking, the factor in L2 as 16 x 16 blocks), beyond that the tiled LDL^T over many
 * workgroups (k_big_*); up to 16320 unknowns (1088 visual-inertial key frames).
 * params->its0 / its1 (and lambda_init / rec_init / large of the VIO params) are ignored.  Not covered: the gravity
 * vertex of the IMU initialiser (pimu_initiator, SURVEY 2 row 14: out of scope).  bScaleOpt: the _scale entries
 * below.  Encoder edges: vieo_lba_imu_edge.enc in the
 * visual-inertial form, vieo_bundle_adjustment_enc (bEnc = true) in the vision-only one. */
int vieo_bundle_adjustment(const vieo_lba_params* params, int n_iterations, int robust,
                           const vieo_lba_keyframe* h_kfs, int n_kf, const float* h_points, int n_mp,
                           const vieo_lba_obs* h_obs, int n_obs, volatile const int* stop, vieo_navstate* h_navs_out,
                           float* h_points_out, vieo_lba_result* h_result);
int vieo_global_bundle_adjustment_vio(const vieo_lba_vio_params* params, int n_iterations, int robust,
                                      const vieo_lba_keyframe* h_kfs, int n_kf, const float* h_points, int n_mp,
                                      const vieo_lba_obs* h_obs, int n_obs, const vieo_lba_imu_edge* h_imu, int n_imu,
                                      volatile const int* stop, vieo_navstate* h_navs_out, float* h_points_out,
                                      vieo_lba_result* h_result);
/* The full BA as System::FinalGBA runs it (src/System.cc:24-33: bScaleOpt = true; Examples/Stereo/stereo_euroc.cc:334-351
 * "FullBA"), BASELINE configs[4].  scale_opt != 0 adds the VertexScale (src/Odom/g2otypes.h:292-311; estimate 1, id after
 * every key-frame vertex, src/Optimizer.cc:842-851) and turns every visual edge into the three-vertex EdgeReprojectPRS /
 * EdgeReprojectPRSStereo (g2otypes.h:321-541 with MODE_OPT_VAR == 1, typedefs :548-550; Optimizer.cc:1131-1137,
 * 1190-1196): the edge projects s * Xh, the point vertex stays unscaled.  Write-back as the reference's
 * (Optimizer.cc:1256-1335): points = (float)s * (float)Xh, key-frame states unscaled; *h_scale_out (may be NULL) = s.
 * With the scale vertex a map whose key frames are all fixed is still optimised (bdimPoses, :850).
 * scale_opt == 0: identical to vieo_global_bundle_adjustment_vio. */
int vieo_global_bundle_adjustment_vio_scale(const vieo_lba_vio_params* params, int n_iterations, int robust, int scale_opt,
                                            const vieo_lba_keyframe* h_kfs, int n_kf, const float* h_points, int n_mp,
                                            const vieo_lba_obs* h_obs, int n_obs, const vieo_lba_imu_edge* h_imu,
                                            int n_imu, volatile const int* stop, vieo_navstate* h_navs_out,
                                            float* h_points_out, vieo_lba_result* h_result, double* h_scale_out);

/* ---- one window over several GPUs ---------------------------------------------------
 * Landmark-sharded LocalBundleAdjustmentNavStatePRV: every rank passes ALL key frames and inertial
 * edges of a window but only ITS share of the points (and their observations).  Per LM trial the ranks
 * exchange exactly one thing, the sum of their visual contributions to the reduced pose system (Schur
 * product, H_pp, b_p), and once more three scalars (chi2 before / after, gain-ratio scale); everything
 * else -- inertial edges, the dense solve, the LM policy -- is replicated and therefore identical on all
 * ranks.  `allreduce(ctx, d_buf, n)` must return 0 after the IN-PLACE sum over all ranks of the n
 * doubles at device pointer d_buf has completed (RCCL: ncclAllReduce(sum, f64) + stream synchronise).
 * d_reduce_buf: device memory of at least vieo_lba_sharded_buffer_doubles(...) doubles.  Key-frame
 * outputs are identical on all ranks, h_points_out / h_erase cover the rank's own points.  No stop flag
 * (the ranks must take the same decisions). */
typedef int (*vieo_allreduce_sum_f64_fn)(void* ctx, double* d_buf, size_t n);
/* In-library RCCL (north_star: "RCCL all-reduce over xGMI on the pose-block Hessian"): librccl.so is loaded with dlopen at
 * first use.  Rank 0 draws the 128-byte unique id, the host carries it to the other ranks (torch.distributed broadcast,
 * MPI, ...), every rank creates its communicator on its own GPU.  Passing allreduce == NULL and ctx == that communicator to
 * the sharded entries makes the library issue ncclAllReduce(sum, f64) itself on its bundle-adjustment stream, between the
 * pack and the assemble kernels, with no host synchronisation in between. */
int vieo_rccl_available(void);
int vieo_rccl_unique_id(uint8_t* id128);
int vieo_rccl_comm_create(void** comm, const uint8_t* id128, int n_ranks, int rank);
int vieo_rccl_comm_destroy(void* comm);
size_t vieo_lba_sharded_buffer_doubles(int n_windows, const int* n_free_kf);

/* Measurement hook of the bundle-adjustment engine (no reference counterpart): with timing on, every kernel launch
 * of the local / full BA entries is bracketed by HIP events on the engine's own stream and folded into process-wide
 * totals per kernel class (vieo_lba_kernel_class_name: "lba.build", "lba.schur", ...).  enable(on) also clears the
 * totals; on == 2 counts the launches (and the Schur FLOPs) per class without recording events, for a timed region
 * that must not carry the events' markers on the stream.  schur_flops: dense FLOPs 2 np (np + 1) 3 n_mp of the windows the timed k_lba_schur launches worked on. */
void vieo_lba_enable_timing(int on);
int vieo_lba_kernel_classes(void);
const char* vieo_lba_kernel_class_name(int i);
void vieo_lba_kernel_times(double* ms /*[classes]*/, long long* launches /*[classes]*/, double* schur_flops);
int vieo_local_bundle_adjustment_vio_sharded(int n_windows, const vieo_lba_vio_params* const* params,
                                             const vieo_lba_keyframe* const* h_kfs, const int* n_kf,
                                             const float* const* h_points, const uint8_t* const* h_close,
                                             const int* n_mp, const vieo_lba_obs* const* h_obs, const int* n_obs,
                                             const vieo_lba_imu_edge* const* h_imu, const int* n_imu,
                                             double* d_reduce_buf, size_t reduce_cap_doubles,
                                             vieo_allreduce_sum_f64_fn allreduce, void* ctx,
                                             vieo_navstate* const* h_navs_out, float* const* h_points_out,
                                             uint8_t* const* h_erase, vieo_lba_result* h_results);
/* The same with pbStopFlag (src/Optimizer.cc:524-528,570-571,590-592): `stop` is this rank's flag (NULL: none).  A sharded
 * run never acts on its own copy: the ranks' requests are summed in the exchanges the run makes anyway (the agreement at
 * entry, the spare fourth scalar of every trial's exchange), so a flag raised on ONE rank aborts the call on ALL ranks at
 * the same trial -- nobody is left waiting in a collective. */
int vieo_local_bundle_adjustment_vio_sharded_stop(int n_windows, const vieo_lba_vio_params* const* params,
                                                  const vieo_lba_keyframe* const* h_kfs, const int* n_kf,
                                                  const float* const* h_points, const uint8_t* const* h_close,
                                                  const int* n_mp, const vieo_lba_obs* const* h_obs, const int* n_obs,
                                                  const vieo_lba_imu_edge* const* h_imu, const int* n_imu,
                                                  double* d_reduce_buf, size_t reduce_cap_doubles,
                                                  vieo_allreduce_sum_f64_fn allreduce, void* ctx, volatile const int* stop,
                                                  vieo_navstate* const* h_navs_out, float* const* h_points_out,
                                                  uint8_t* const* h_erase, vieo_lba_result* h_results);
/* The same exchange for the full BA (BASELINE configs[4]: "full BA with RCCL pose-Hessian all-reduce"): this rank's
 * landmark shard of GlobalBundleAdjustmentNavStatePRV; per LM trial one all-reduce of the packed reduced visual
 * system ((6n)(6n+1) + 42n doubles for n free key frames) and one of three scalars. */
int vieo_global_bundle_adjustment_vio_sharded(const vieo_lba_vio_params* params, int n_iterations, int robust,
                                              const vieo_lba_keyframe* h_kfs, int n_kf, const float* h_points,
                                              int n_mp, const vieo_lba_obs* h_obs, int n_obs,
                                              const vieo_lba_imu_edge* h_imu, int n_imu, double* d_reduce_buf,
                                              size_t reduce_cap_doubles, vieo_allreduce_sum_f64_fn allreduce,
                                              void* ctx, vieo_navstate* h_navs_out, float* h_points_out,
                                              vieo_lba_result* h_result);
/* ... and of System::FinalGBA's form (scale_opt, see vieo_global_bundle_adjustment_vio_scale): the scale vertex's row of
 * the reduced visual system and its H_ps / H_ss / b_s travel in the same all-reduce ((6n+1)(6n+2) + 48n + 2 doubles). */
int vieo_global_bundle_adjustment_vio_sharded_scale(const vieo_lba_vio_params* params, int n_iterations, int robust,
                                                    int scale_opt, const vieo_lba_keyframe* h_kfs, int n_kf,
                                                    const float* h_points, int n_mp, const vieo_lba_obs* h_obs, int n_obs,
                                                    const vieo_lba_imu_edge* h_imu, int n_imu, double* d_reduce_buf,
                                                    size_t reduce_cap_doubles, vieo_allreduce_sum_f64_fn allreduce,
                                                    void* ctx, vieo_navstate* h_navs_out, float* h_points_out,
                                                    vieo_lba_result* h_result, double* h_scale_out);

/* ---- replay glue (device-resident batches) -------------------------------------------------
 * What Tracking.cc does between the calls above, on flattened arrays, so a batch of frames runs
 * extract -> stereo -> search -> pose optimisation with no host round trip (bench.py):
 * keypoint -> point bookkeeping (Frame::mvpMapPoints), the observation gathering at the top of
 * PoseOptimization (Optimizer.cc:1704-1786), "Discard outliers" (Tracking.cc:1903-1921).
 * d_mp_ref[f][key_cap]: index into frame f's point table d_point_xyz[f][p_cap][3], -1 = none. */
int vieo_track_merge_assign_batch_device(const int32_t* d_assign, int32_t* d_mp_ref,
                                         const int32_t* d_counts, int key_cap, int n_frames,
                                         int img_first, int img_step, int point_offset, int reset,
                                         void* stream);
/* Fills d_obs[f][key_cap] / d_obs_key and writes n_obs, obs_begin (= f*key_cap) into the frame
 * structs (vieo_pose_frame or vieo_vio_frame array). */
int vieo_track_build_obs_batch_device(const int32_t* d_mp_ref, const float* d_point_xyz, int p_cap,
                                      const vieo_keypoint* d_keys, const float* d_uright,
                                      const int32_t* d_counts, int key_cap, int n_frames,
                                      int img_first, int img_step, const float* d_inv_sigma2,
                                      vieo_pose_obs* d_obs, int32_t* d_obs_key, void* d_frames,
                                      int frames_are_vio, void* stream);
/* Drops outlier matches from d_mp_ref, exports d_taken (may be NULL) for the next search and
 * copies the optimised NavState into d_next_frames[f].nav (may be NULL). */
int vieo_track_after_pose_batch_device(int32_t* d_mp_ref, const int32_t* d_obs_key,
                                       const uint8_t* d_outlier, const void* d_frames,
                                       const void* d_results, int frames_are_vio, int key_cap,
                                       int n_frames, void* d_next_frames, uint8_t* d_taken,
                                       void* stream);
/* ... and vieo_track_mark_held_batch_device (below) in the same launch: d_held[f][p_cap] from the entries left. */
int vieo_track_after_pose_held_batch_device(int32_t* d_mp_ref, const int32_t* d_obs_key, const uint8_t* d_outlier,
                                            const void* d_frames, const void* d_results, int frames_are_vio, int key_cap,
                                            int n_frames, void* d_next_frames, uint8_t* d_taken, const int32_t* d_counts,
                                            int img_first, int img_step, uint8_t* d_held, int p_cap, void* stream);
/* vieo_track_merge_assign[_rig]_batch_device + vieo_track_build_obs[_depth | _rig]_batch_device as ONE launch (the one-call
 * tracker's chain is a launch shorter per search): the assignment is merged into d_mp_ref key by key and the observations
 * are gathered from the merged entries.  d_same_point / d_query_src / d_point_depth / d_cam_first may be NULL (the plain
 * forms); same outputs as the two calls. */
int vieo_track_merge_build_obs_batch_device(const int32_t* d_assign, int32_t* d_mp_ref, int point_offset, int reset, int query_div,
                                            const vieo_last_frame_point* d_same_point, const int32_t* d_query_src, int q_cap,
                                            const float* d_point_xyz, const float* d_point_depth, float close_depth, int p_cap,
                                            const vieo_keypoint* d_keys, const float* d_uright, const int32_t* d_counts,
                                            const int32_t* d_cam_first, int n_cams, int key_cap, int n_frames, int img_first,
                                            int img_step, const float* d_inv_sigma2, vieo_pose_obs* d_obs, int32_t* d_obs_key,
                                            void* d_frames, int frames_are_vio, void* stream);
/* The same with the `close` bit of the observations (vieo_pose_obs.flags bit 0) taken from the depth at which the
 * point was tracked: d_point_depth[f][p_cap] < close_depth (Frame::mvpMapPoints[i]->mTrackDepth against
 * max(10, ThDepth), the stereo chi2 gate of the visual-inertial PoseOptimization, include/Optimizer.h:406-490). */
int vieo_track_build_obs_depth_batch_device(const int32_t* d_mp_ref, const float* d_point_xyz,
                                            const float* d_point_depth, float close_depth, int p_cap,
                                            const vieo_keypoint* d_keys, const float* d_uright,
                                            const int32_t* d_counts, int key_cap, int n_frames,
                                            int img_first, int img_step, const float* d_inv_sigma2,
                                            vieo_pose_obs* d_obs, int32_t* d_obs_key, void* d_frames,
                                            int frames_are_vio, void* stream);
/* The two glue steps for camera-rig frames: a search's query (point i, camera c) is i * query_div + c (query_div =
 * n_cams), and an observation carries its key's camera in bits 8..11 of vieo_pose_obs.flags (mapn2in_,
 * include/Optimizer.h:424-426) -- the camera found from d_cam_first [f][n_cams + 1]; keys [f][key_cap] camera-major,
 * d_counts [f][2] = {N, -} (d_frame_counts of vieo_stereo_fisheye_match_batch_device); d_point_depth may be NULL. */
int vieo_track_merge_assign_rig_batch_device(const int32_t* d_assign, int32_t* d_mp_ref, const int32_t* d_counts,
                                             int key_cap, int n_frames, int img_first, int img_step, int point_offset,
                                             int reset, int query_div,
                                             const vieo_last_frame_point* d_same_point /*[f][key_cap] or NULL: reserved[0]
                                             > 0 names (1 +) the first point-table entry of the same MapPoint*/,
                                             const int32_t* d_query_src /*[f][q_cap] or NULL: the searched list was
                                             compacted, entry a came from query d_query_src[a]*/, int q_cap, void* stream);
/* The valid queries (flags bit 0) of every frame moved to the front of d_queries_out in their order, d_src[f][k] = the
 * index the k-th one had, d_nq_out[f] = their number.  A rig frame's first search holds one query per (last-frame key,
 * camera) pair and most of them project outside their camera; the search walks the list several times. */
int vieo_track_compact_queries_batch_device(const vieo_proj_query* d_queries, const int32_t* d_nq, int q_cap, int n_frames,
                                            vieo_proj_query* d_queries_out, int32_t* d_src, int32_t* d_nq_out, void* stream);
int vieo_track_build_obs_rig_batch_device(const int32_t* d_mp_ref, const float* d_point_xyz,
                                          const float* d_point_depth, float close_depth, int p_cap,
                                          const vieo_keypoint* d_keys, const float* d_uright,
                                          const int32_t* d_counts, const int32_t* d_cam_first, int n_cams, int key_cap,
                                          int n_frames, const float* d_inv_sigma2, vieo_pose_obs* d_obs,
                                          int32_t* d_obs_key, void* d_frames, int frames_are_vio, void* stream);
/* d_held[f][p_cap] = 1 for the entries of frame f's point table that a key holds (the points
 * Tracking::SearchLocalPoints takes out of the local-map search, src/Tracking.cc:2318-2334). */
int vieo_track_mark_held_batch_device(const int32_t* d_mp_ref, const int32_t* d_counts, int key_cap,
                                      int n_frames, int img_first, int img_step, uint8_t* d_held, int p_cap,
                                      void* stream);
/* The extractor's own hipStream_t, so the calls above can be chained on it. */
void* vieo_orb_stream(vieo_orb* e);

/* ---- test taps (parity tests only; not part of the drop-in surface) ---- */
/* which: 1 = blurred level.  FAST candidates: int32 triplets (x, y, response) in
 * vToDistributeKeys order; level keys: vieo_keypoint in DistributeOctTree output order. */
int vieo_orb_tap_plane(vieo_orb* e, int image_index, int level, int which, uint8_t* h_dst,
                       int dst_stride);
int vieo_orb_tap_candidates(vieo_orb* e, int image_index, int level, int32_t* h_dst, int cap);
int vieo_orb_tap_level_keys(vieo_orb* e, int image_index, int level, vieo_keypoint* h_dst,
                            int cap);

/* ---------------------------------------------------------------- map-point side --------------
 * The per-point steps either side of SearchByProjection(local map) and of the local BA write-back. */

/* bool Frame::isInFrustum(MapPoint* pMP, float viewingCosLimit) (src/Frame.cc:335-416) for a batch of points:
 * per camera positive depth, image bounds, distance inside [0.8 mfMinDistance, 1.2 mfMaxDistance], viewing
 * cosine against the mean normal, MapPoint::PredictScale (src/MapPoint.cc:491-509).  All float, as the
 * reference computes it; Sophus SE3f products are taken through their 3x4 matrices. */
typedef struct vieo_frustum_frame {
  float Rcrw[9], tcrw[3]; /* Tcw_ rotation (row-major) and mtcw, cast to float (Frame.cc:348-351) */
  float Ow[3];            /* mOw */
  int32_t n_cams;         /* 1..4 */
  int32_t use_distort;    /* Frame::usedistort_: 0 -> K * (x/z, y/z, 1) in float, 1 -> camera Project() */
  const struct vieo_camera* cams; /* host pointer, n_cams entries (Rcb / tcb unused) */
  float Tcr[4][12];       /* mpCameras[i]->GetTcr().matrix3x4() row-major (identity for camera 0) */
  float trc[4][3];        /* mpCameras[i]->GetTrc().translation() */
  float bounds[4][4];     /* gridinfo_.minmax_xy_[i] = {min_x, max_x, min_y, max_y} */
  float bf;               /* stereoinfo_.baseline_bf_[1] */
  float log_scale_factor; /* scalepyrinfo_.flogscalefactor_ */
  int32_t n_levels;
  float viewing_cos_limit;
} vieo_frustum_frame;

typedef struct vieo_frustum_point {
  float Xw[3], normal[3];           /* GetWorldPos(), GetNormal() */
  float max_distance, min_distance; /* mfMaxDistance, mfMinDistance (the 1.2 / 0.8 factors are applied here) */
} vieo_frustum_point;               /* 32 bytes */

typedef struct vieo_track_info {    /* MapPoint::_TrackFastMatchInfo after the call */
  float u[4], v[4], ur[4], viewcos[4]; /* vtrack_proj_[0..2], vtrack_viewcos_ (entries 0..n-1, push order) */
  int32_t level[4], cam[4];         /* vtrack_scalelevel_, vtrack_cami_ */
  int32_t n;                        /* cameras that see the point; btrack_inview_ = n > 0 */
  float track_depth;                /* mean dist3D over those cameras; -1 when n == 0 (the member keeps its value) */
} vieo_track_info;                  /* 104 bytes */

int vieo_is_in_frustum_batch(const vieo_frustum_frame* h_frame, const vieo_frustum_point* h_points, int n_points,
                             vieo_track_info* h_info);
/* The head of Tracking::SearchLocalPoints (src/Tracking.cc:2308-2370) for ONE frame whose pose is still in HBM, so
 * that TrackWithIMU -> TrackLocalMapWithIMU runs as one chain of launches (no host round trip between the first
 * PoseOptimization and the local-map search): the pose is d_result->base.nav when d_result->base.status == 0, else
 * d_frame->base.nav, with d_frame->base.Rcb / tcb (h_frame's Rcrw / tcrw / Ow are ignored, the rest of h_frame is used
 * as in vieo_is_in_frustum_batch); isInFrustum of the n_points candidates; their window queries in the order of
 * SearchByProjection(Frame&, vector<MapPoint*>&, th, th_far) (src/ORBmatcher.cc:237-266): d_queries[p * n_cams + k]
 * is the k-th camera that sees candidate p (flags = 0 in the unused slots), *d_nq = n_points * n_cams.
 * d_alias[p] >= 0 (may be NULL): candidate p is the same map point as entry d_alias[p] of the frame's point table;
 * it gets no query when d_held[d_alias[p]] != 0 (vieo_track_mark_held_batch_device; d_held has held_cap entries,
 * larger indices count as not held).  d_track_depth[p] = mTrackDepth
 * (-1 when no camera sees the point).  d_scale: scalepyrinfo_.vscalefactor_ (n_levels floats). */
int vieo_track_local_queries_device(const vieo_frustum_frame* h_frame, const vieo_vio_frame* d_frame,
                                    const vieo_vio_result* d_result, const vieo_frustum_point* d_points,
                                    const uint8_t* d_desc, const int32_t* d_alias, const uint8_t* d_held, int held_cap,
                                    int n_points, float th, float th_far, const float* d_scale, vieo_proj_query* d_queries,
                                    float* d_track_depth, int32_t* d_nq, void* stream);

/* The same for a batch of frames (bench.py's batched step): frame f reads d_frames[f] / d_results[f], its candidates
 * d_points / d_desc / d_alias [f][p_cap] (d_counts[f] of them), d_held [f][held_cap]; writes d_queries [f][p_cap * n_cams],
 * d_track_depth + f * depth_stride, d_nq[f] = d_counts[f] * n_cams. */
int vieo_track_local_queries_batch_device(const vieo_frustum_frame* h_frame, const vieo_vio_frame* d_frames,
                                          const vieo_vio_result* d_results, int n_frames, const vieo_frustum_point* d_points,
                                          const uint8_t* d_desc, const int32_t* d_alias, const int32_t* d_counts, int p_cap,
                                          const uint8_t* d_held, int held_cap, float th, float th_far, const float* d_scale,
                                          vieo_proj_query* d_queries, float* d_track_depth, size_t depth_stride,
                                          int32_t* d_nq, void* stream);

/* void MapPoint::ComputeDistinctiveDescriptors() (src/MapPoint.cc:314-378) for a batch of points: point p owns
 * the descriptor rows [h_first[p], h_first[p + 1]) of h_descriptors (its observations in map order); h_best[p]
 * receives the row (relative to h_first[p]) with the least median Hamming distance to the others, first such
 * row on ties; -1 for a point without observations.  At most 128 observations per point. */
int vieo_distinctive_descriptors_batch(const uint8_t* h_descriptors, const int32_t* h_first, int n_points,
                                       int32_t* h_best);

/* void MapPoint::UpdateNormalAndDepth() (src/MapPoint.cc:424-480) for a batch of points: observation i of point
 * p looks from the camera centre h_centres[h_obs_centre[i]] (twc of the (key frame, camera) pair, float[3]);
 * outputs the mean viewing direction and mfMaxDistance / mfMinDistance from the reference key frame
 * (h_ref_centre[p] index into h_centres, h_ref_scale[p] = vscalefactor_[octave of the key in it]). */
int vieo_update_normal_and_depth_batch(const float* h_points /*[n][3]*/, const int32_t* h_first,
                                       const int32_t* h_obs_centre, const float* h_centres, int n_centres,
                                       const int32_t* h_ref_centre, const float* h_ref_scale,
                                       float scale_last_level, int n_points, float* h_normal /*[n][3]*/,
                                       float* h_max_distance, float* h_min_distance);

/* The search of ORBmatcher::SearchByProjectionBase (src/ORBmatcher.cc:26-227) -- what ORBmatcher::Fuse (:1152-1165,
 * LocalMapping::SearchInNeighbors) and the Sim3 / loop-closing callers run per map point: projection into every
 * camera of the key frame (positive depth, FrameBase::IsInImage, scale-invariance distances, optional 60 degree
 * viewing cone), MapPoint::PredictScale, FrameBase::GetFeaturesInArea (FrameBase.cpp:95-141) with radius
 * th_radius * scale[level], octave in [level-1, level], the chi2 gate on the reprojection error (7.8 stereo /
 * 5.99 mono, only when use_bf), best Hamming distance (first minimum in the reference's candidate order).
 * Per (point, camera): h_best_idx (index into that camera's keys, -1: none) and h_best_dist.  The thresholds on
 * the distance, KeyFrame::FuseMP and the only-one-match bookkeeping (:195-224) stay with the caller: they are
 * order-dependent mutations of the map.  A point whose skip_mask has bit 31 set is not searched at all
 * (isBad / IsInKeyFrame), bit c skips camera c (pvbAlreadyMatched1). */
typedef struct vieo_fuse_frame {
  vieo_frustum_frame base;     /* Rcrw, tcrw, Ow = GetCameraCenter(), cameras, bounds, bf, log scale factor, n_levels */
  float scale_factors[16];     /* scalepyrinfo_.vscalefactor_ */
  float inv_level_sigma2[16];  /* scalepyrinfo_.vinvlevelsigma2_ */
  float th_radius;
  int32_t check_viewing_angle; /* bCheckViewingAngle */
  int32_t use_bf;              /* pbf != nullptr: chi2 gate with the stereo coordinate */
  int32_t reserved;
} vieo_fuse_frame;

typedef struct vieo_fuse_point {
  float Xw[3], normal[3];
  float max_distance, min_distance; /* mfMaxDistance, mfMinDistance */
  uint8_t desc[32];                 /* GetDescriptor() */
  int32_t skip_mask;
  int32_t reserved;
} vieo_fuse_point;                  /* 72 bytes */

int vieo_fuse_search(const vieo_fuse_frame* h_frame, const vieo_keypoint* const* h_keys /*[n_cams]*/,
                     const float* const* h_uright /*[n_cams], may be NULL entries: all monocular*/,
                     const uint8_t* const* h_descriptors /*[n_cams]*/, const int32_t* n_keys /*[n_cams]*/,
                     const vieo_fuse_point* h_points, int n_points, int32_t* h_best_idx /*[n_points][n_cams]*/,
                     int32_t* h_best_dist /*[n_points][n_cams]*/);

/* ---------------------------------------------------------------- IMU pre-integration ----------
 * int IMUPreIntegratorBase::PreIntegration(timeStampi, timeStampj, bgi_bar, bai_bar, iterBegin, iterEnd, breset =
 * true) + update() (src/Odom/OdomPreIntegrator.h:226-506; mid-point samples, the partial intervals at both ends
 * interpolated as the reference does, USE_PREINT_EULA off) for a batch of intervals, one lane per interval:
 * delta R / v / p, the five bias Jacobians, Sigma in both orders (mSigmaij: p v Phi; mSigmaijPRV: p Phi v).
 * Interval k owns the samples [h_first[k], h_first[k + 1]) (time-ordered).  h_ti[k] > h_tj[k] is the reference's
 * backward order (map reuse, :241-262): the samples are walked from the end with negative steps.  h_status[k]: */
#define VIEO_PREINT_OK 0
#define VIEO_PREINT_EMPTY 1       /* no samples: PreIntegration() does nothing (outputs zeroed, dt = 0) */
#define VIEO_PREINT_GAP 2         /* |dt| > 1.5 s between samples: "CheckIMU", mdeltatij = 0, returns -1 */
#define VIEO_PREINT_UNSUPPORTED 3 /* (not returned any more: the backward order of map reuse, timeStampi > timeStampj, is built) */
typedef struct vieo_imu_sample {
  double t;            /* IMUData::mtm */
  double w[3], a[3];   /* mw, ma */
} vieo_imu_sample;     /* 56 bytes */
typedef struct vieo_imu_noise {
  double sigma_g[9], sigma_a[9]; /* IMUDataBase::mSigmag, mSigmaa (row-major) */
  double freq_ref;               /* IMUDataBase::mFreqRef */
  int32_t dt_cov_noise_fixed;    /* IMUDataBase::mdt_cov_noise_fixed */
  int32_t reserved;
} vieo_imu_noise;
int vieo_imu_preintegrate_batch(const vieo_imu_noise* noise, const vieo_imu_sample* h_samples,
                                const int32_t* h_first, const double* h_ti, const double* h_tj,
                                const double* h_bg /*[n][3]*/, const double* h_ba /*[n][3]*/, int n,
                                vieo_imu_preint* h_out, double* h_sigma_prv /*[n][81], may be NULL*/,
                                int32_t* h_status);

/* Device form of the same call (the chained frame below runs it on its own stream next to the extraction): every
 * array in HBM, asynchronous on `stream`; n < 1024 intervals get a wavefront each. */
int vieo_imu_preintegrate_batch_device(const vieo_imu_noise* d_noise, const vieo_imu_sample* d_samples,
                                       const int32_t* d_first, const double* d_ti, const double* d_tj,
                                       const double* d_bg, const double* d_ba, int n, vieo_imu_preint* d_out,
                                       double* d_sigma_prv, int32_t* d_status, void* stream);

/* ---------------------------------------------------------------- one frame's tracking as ONE call -----------
 * What Tracking::Track does for a stereo-inertial frame in the steady state -- Frame::Frame (ExtractORB x 2,
 * src/Frame.cc:259-320; ComputeStereoMatches :451-611), PreIntegration + PredictNavStateByIMU (src/Tracking.cc:385-451),
 * TrackWithIMU (:261-378: SearchByProjection(last frame) -> PoseOptimization), TrackLocalMapWithIMU (:453-488:
 * SearchLocalPoints :2308-2370 -> PoseOptimization(bComputeMarg)) -- as one chain of launches on one stream: ONE copy
 * up from pinned memory (two when the local map changed), the kernels of the entries above back to back with the
 * bookkeeping between them on the device (vieo_track_* glue), the IMU pre-integration on a second stream beside the
 * extraction, ONE copy back, ONE host synchronisation.  The caller keeps the reference's object graph (Frame /
 * KeyFrame / MapPoint / Map) and hands over flattened views of what the calls read; outputs point into pinned memory
 * owned by the tracker and stay valid until its next call.  Thread model: one tracker per tracking thread; other
 * threads (LocalMapping, LoopClosing) call the other entries concurrently on their own streams.
 * Three kinds of tracker (round 4): rectified stereo + IMU (BASELINE configs[1] / [2]; vieo_tracker_create); a distorted
 * camera rig of 2..4 cameras + IMU (the reference's default EuRoC_VIO_dist* set-up, configs[3] / [4];
 * vieo_tracker_create_rig): ExtractORB x n_cams with the KB8 lapping area, Frame::ComputeStereoFishEyeMatches
 * (Frame.cc:613-779) instead of the rectified matcher, the camera loop in both searches, the rig instances of the
 * optimiser; and rectified stereo WITHOUT the IMU (configs[0]; params.vision_only): Tracking::TrackWithMotionModel
 * (Tracking.cc:1844-1928) + TrackLocalMap (:1930-1945) with Optimizer::PoseOptimization(Frame*, Frame*)
 * (Optimizer.cc:1611-1874); the predicted pose (mVelocity * mLastFrame.Tcw, a 4x4 product) is the caller's. */
typedef struct vieo_tracker vieo_tracker;
typedef struct vieo_tracker_params {
  int32_t width, height;                /* image size; both cameras */
  int32_t n_features, n_levels, ini_th_fast, min_th_fast;
  float scale_factor;                   /* ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) */
  float fx, fy, cx, cy, bf, baseline;   /* rectified pinhole, stereoinfo_.baseline_bf_ */
  float th_depth;                       /* Frame::mThDepth */
  float th_last, th_local;              /* th of SearchByProjection(last frame) / (local map): Tracking.cc:296,2365 */
  float nn_last, nn_local;              /* ORBmatcher(nnratio): 0.9 / 0.8 */
  int32_t max_local_points;             /* capacity of the local-map candidate table */
  double Rcb[9], tcb[3];                /* camera <- body */
  double gw[3];                         /* gravity in the world frame */
  double inv_sigma_bg2, inv_sigma_ba2;  /* IMUDataBase::mInvSigmabg2 / mInvSigmaba2 */
  vieo_imu_noise noise;
  int32_t vision_only;                  /* 1: no IMU -- nav_ref of the input is the PREDICTED state of the frame (p, q; the
                                         * rest is carried along), imu / prior inputs are ignored, the two optimisations are
                                         * the vision-only ones and `first` / `second` hold their vieo_pose_result in .base */
  int32_t reserved;
} vieo_tracker_params;

/* The rig of a distorted multi-camera tracker (Frame::usedistort_, mpCameras).  params.fx..cy are not read (camera 0's
 * are used where the reference takes mpCameras[0]->toK()); params.Rcb / tcb: body -> reference camera; params.bf /
 * baseline: stereoinfo_.baseline_bf_. */
typedef struct vieo_tracker_rig {
  int32_t n_cams;                       /* 2..4 */
  int32_t use_lapping;                  /* KB8 cameras hand their lapping area to ExtractORB (Frame.cc:269-273) */
  int32_t lapping[2];                   /* GetvLappingArea() */
  float th_far_pts;                     /* mpLocalMapper->th_far_pts_ (<= 0: off) */
  float reserved;
  vieo_camera cams[4];                  /* model + parameters; Rcb / tcb: body -> camera c (pose optimisation) */
  double Trc[4][12], Tcr[4][12];        /* mpCameras[c]->GetTrc() / GetTcr() (Sophus::SE3<float>) cast to double, 3x4 */
} vieo_tracker_rig;

typedef struct vieo_track_input {
  const uint8_t *left, *right;          /* 8-bit grey images, `stride` bytes per row (or the tracker's own pinned
                                         * planes from vieo_tracker_image_buffers: then no host copy is made) */
  int32_t stride;
  int32_t n_imu;
  const vieo_imu_sample* imu;           /* the samples PreIntegration hands over for [t_ref, t_cur] */
  double t_ref, t_cur;
  vieo_navstate nav_ref;                /* the state the prediction starts from: the last key frame's after a map update,
                                         * else the last frame's (Tracking.cc:392-409) */
  vieo_navstate nav_last;               /* mLastFrame's state (Tcw of the projection search's LastFrame) */
  const vieo_navstate* nav_prior;       /* mNavStatePrior / mMargCovInv of the reference state, NULL: none (mbPrior) */
  const double* H_prior;                /* 15 x 15 row-major */
  int32_t n_last;
  const vieo_last_frame_point* last_points; /* mLastFrame.mvpMapPoints flattened (n_last = its key count) */
  const float* last_track_depth;        /* mTrackDepth of those points (inf: unknown) */
  int32_t n_local;                      /* local-map candidates (all points of the local key frames; the ones the frame
                                         * already holds are dropped on the device through local_alias) */
  int32_t local_version;                /* change it whenever the candidate arrays differ from the previous call's:
                                         * equal versions skip the re-upload (the local map changes per key frame) */
  const vieo_frustum_point* local_points;
  const uint8_t* local_desc;            /* [n_local][32] */
  const int32_t* local_alias;           /* candidate j is last_points[local_alias[j]] (-1: not in the last frame) */
  const uint8_t* images[4];             /* rig trackers: camera c's image (left / right are not read); n_last then counts
                                         * mLastFrame's keys in mvKeys (camera-major) order, up to vieo_tracker_key_capacity */
  /* Frame pipelining (a replay or any caller that has the NEXT frame's images in hand -- a camera
   * driver one frame ahead, a dataset player): the next frame's ExtractORB x 2 + ComputeStereoMatches are queued on a
   * third stream BEHIND this frame's stereo stage and run beside this frame's searches and optimisations (which keep one
   * CU busy; the extraction wants all of them for ~0.2 ms).  The NEXT call then finds its frame extracted: its
   * left / right are not read, and it says so with use_prefetched = 1 -- a call with use_prefetched = 0 discards a pending
   * prefetch.  Outputs are those of the unpipelined call bit for bit (same kernels on the same data).  NULL: off. */
  const uint8_t *next_left, *next_right; /* `stride` bytes per row like left / right */
  int32_t use_prefetched;
  /* ... and the next frame's pre-integration (IMU trackers): with the extraction out of the way it is the 75 us serial
   * chain at the head of the next call.  next_imu = the samples PreIntegration will hand over for [t_cur, next_t_cur],
   * i.e. with THIS frame as the next call's reference: they are integrated beside this frame's tail with the bias this
   * frame's prediction produced (bj_bar = bi_bar + dbi).  The next call uses the result when -- and only when -- its own
   * t_ref / t_cur / samples / nav_ref.bg / nav_ref.ba are bit for bit what was integrated (after a map update the
   * reference is the key frame: the comparison fails and the call integrates as usual).  NULL / 0: off. */
  int32_t next_n_imu;
  const vieo_imu_sample* next_imu;
  double next_t_cur;
  const uint8_t* next_images[4];        /* rig trackers: the next frame's camera images instead of next_left / next_right
                                         * (all of them or none); its extraction runs ahead, its stereo stage in its call */
  /* ... and when the NEXT call's reference will not be this frame: the caller is about to apply a local-BA write-back, so
   * the next prediction starts at the last key frame (Tracking.cc:392-409) and integrates every sample since -- 0.5 ms at
   * the head of that frame's chain.  The run-ahead integration can be told its reference: next_imu = the samples for
   * [next_t_ref, next_t_cur], next_ref_bias = that key frame's (bg, ba) as they will be AFTER the write-back.  Same rule:
   * the next call uses the result only if its own t_ref / t_cur / samples / nav_ref.bg / nav_ref.ba are bit for bit what was
   * integrated.  NULL: the reference is this frame (above). */
  const double* next_ref_bias;          /* [6] bg, ba */
  double next_t_ref;
} vieo_track_input;

#define VIEO_TRACK_OK 0
#define VIEO_TRACK_PREINT_FAILED 1      /* mdeltatij == 0 / CheckIMU: PredictNavStateByIMU returns false; extraction and
                                         * stereo outputs are valid, the tracking outputs are not */
#define VIEO_TRACK_LOST 2               /* fewer than 10 (vision-only: 20) matches after the widened search: TrackWithIMU /
                                         * TrackWithMotionModel return false before the optimisations (Tracking.cc:311,1878);
                                         * extraction, stereo and search outputs are valid, `first` / `second` are not.
                                         * (The inlier gates behind the optimisations -- nmatchesMap, mnMatchesInliers --
                                         * need MapPoint::Observations() and stay with the caller.) */
typedef struct vieo_track_output {
  int32_t status;                       /* VIEO_TRACK_* */
  int32_t n_keys, key_cap;              /* left image: N, and the offset that separates the two point tables */
  const vieo_keypoint* keys;            /* [n_keys] */
  const uint8_t* desc;                  /* [n_keys][32] */
  const float *uright, *depth;          /* ComputeStereoMatches */
  const int32_t* point_ref;             /* per key: -1 none; < key_cap: last_points[i]; else local_points[i - key_cap] */
  const uint8_t* outlier;               /* per key: mvbOutlier after the second optimisation */
  const float* local_track_depth;       /* [n_local] mTrackDepth of the candidates (isInFrustum) */
  int32_t n_matches_last, n_matches_local; /* return values of the two searches */
  int32_t widened;                      /* 1: the first search ran again with 2 * th_last (Tracking.cc:301-309) */
  vieo_navstate nav_pred;               /* PredictNavStateByIMU */
  vieo_imu_preint imu;                  /* the pre-integration [t_ref, t_cur] (Sigma in (p, v, Phi) order) */
  int32_t preint_status;                /* VIEO_PREINT_* */
  int32_t reserved;
  vieo_vio_result first, second;        /* TrackWithIMU's / TrackLocalMapWithIMU's PoseOptimization */
  float ms_gpu;                         /* HIP-event time of the chain, upload to download */
  float ms_host;                        /* wall time of the call */
  /* rig trackers: keys / desc / uright (-1) / depth / point_ref / outlier are in mvKeys order (camera-major) */
  int32_t cam_first[5];                 /* first key of camera c, [n_cams] = n_keys (mapn2in_) */
  int32_t mono_index[4];                /* num_mono of ExtractORB per camera */
  int32_t stereo_status;                /* != 0: more stereo groups than vieo_tracker_group_capacity, no depths */
  int32_t n_groups, n_stereo_matches;   /* ComputeStereoFishEyeMatches: mvidxsMatches.size(), nMatches */
  const int32_t* key_group;             /* [n_keys] mapcamidx2idxs_ (-1: none) */
  const int32_t* group_idx;             /* [n_groups][n_cams] mvidxsMatches */
  const uint8_t* group_good;            /* [n_groups] goodmatches_ */
  const double* group_p3d;              /* [n_groups][3] v3dpoints_ (reference-camera frame) */
} vieo_track_output;

int vieo_tracker_create(vieo_tracker** out, const vieo_tracker_params* params);
int vieo_tracker_create_rig(vieo_tracker** out, const vieo_tracker_params* params, const vieo_tracker_rig* rig /*NULL: rectified*/);
void vieo_tracker_destroy(vieo_tracker* t);
int vieo_tracker_key_capacity(const vieo_tracker* t);   /* keys of a frame at most (n_cams x the extractor's capacity) */
int vieo_tracker_group_capacity(const vieo_tracker* t); /* stereo groups of a rig frame at most */
/* pinned planes the caller may decode the next frame's images into (stride = width) */
int vieo_tracker_image_buffers(vieo_tracker* t, uint8_t** left, uint8_t** right);
int vieo_tracker_image_buffer(vieo_tracker* t, int image_index, uint8_t** plane);
int vieo_tracker_scale_factors(const vieo_tracker* t, float* h_out /*[n_levels]*/);
int vieo_track_frame(vieo_tracker* t, const vieo_track_input* in, vieo_track_output* out);
/* What the tracker knows about its two streams and its replicated optimisations.  A tracker runs its frame on two HIP
 * streams (pre-integration, a rig frame's stereo bookkeeping and the copies back beside the extraction); they overlap only
 * when the runtime serves them from different hardware queues, which it decides from the streams the PROCESS holds.  The
 * second stream is therefore chosen by measurement at creation, the measured figure is kept here, and the choice is made
 * again -- vieo_tracker_reprobe, or by the tracker itself when eight frames in a row take 30 % more GPU time than the
 * running median of the last 32 -- when streams created later have changed the mapping. */
typedef struct vieo_tracker_stats {
  float side_stream_ratio;        /* elapsed(two ~40 us spin kernels started together, one per stream) / elapsed(one):
                                   * ~1 = side by side, ~2 = one hardware queue; > 1.35: no better stream was to be had */
  int32_t side_stream_selections; /* times a second stream was chosen (1 = at creation only) */
  int32_t side_stream_checks;     /* times the ratio was measured again */
  int32_t replica_repeats;        /* rig frames whose optimisations were repeated on one workgroup because a replica
                                   * workgroup never became resident (vieo_pose_set_replicas) */
  float ms_gpu_median;            /* running median of vieo_track_output.ms_gpu (last 32 frames) */
  int32_t slow_frames_in_a_row;   /* frames in a row above 1.3 x that median */
  int32_t frames_prefetched;      /* frames extracted ahead (vieo_track_input.next_left / next_right) */
  int32_t preints_ahead_used;     /* calls that found their pre-integration done by the previous call (next_imu) */
} vieo_tracker_stats;
int vieo_tracker_get_stats(const vieo_tracker* t, vieo_tracker_stats* out);
int vieo_tracker_reprobe(vieo_tracker* t);
/* mvImagePyramid of the frame just tracked, lazily (only Frame::ComputeStereoMatches reads it, src/Frame.cc:457,536-557,
 * and that ran on the device): image 0 = left, 1 = right; see vieo_orb_get_level */
int vieo_tracker_get_level(vieo_tracker* t, int image_index, int level, int with_border, uint8_t* h_dst, int dst_stride);

/* Per-call form of the kernel-instance choice (vieo_pose_set_camera_mode / _encoder_mode above are per host thread
 * and kept for old callers): cams_mode VIEO_POSE_CAMS_*, enc_mode VIEO_POSE_ENC_*. */
int vieo_pose_optimization_vio_batch_device_ex(const vieo_vio_frame* d_frames, int n_frames, const vieo_pose_obs* d_obs,
                                               uint8_t* d_outlier, vieo_vio_result* d_results, int cams_mode,
                                               int enc_mode, void* stream);
int vieo_pose_optimization_batch_device_ex(const vieo_pose_frame* d_frames, int n_frames, const vieo_pose_obs* d_obs,
                                           uint8_t* d_outlier, vieo_pose_result* d_results, int cams_mode, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VIEO_HOT_H */
