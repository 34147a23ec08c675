// orb_internal.h -- device-visible geometry tables and the extractor handle, shared by the
// extractor kernels (orb_extractor.hip) and their device-side consumers (matching.hip).
#pragma once
#include <vector>

#include "common.h"

namespace vieo {

static const int kPatchSize = 31, kHalfPatch = 15, kEdge = 19;
static const int kMaxLevels = 16;
#ifndef VIEO_BLUR_TH
#define VIEO_BLUR_TH 96  // rows of a blur tile (even): 32 / 64 / 96 / 128 measured 1.18 / 1.02 / 0.96 / 1.00 ms per 1024 images (tools/ab_blur_th.sh)
#endif
static const int kBlurTW = 64, kBlurTH = VIEO_BLUR_TH;

struct LevelDesc {
  int w, h, pitch, off;      // plane geometry; off = byte offset in the per-image pyramid block
  int boff;                  // byte offset in the per-image blurred block
  int cell_begin, cell_end;  // range in the cell table
  int nfeat;                 // mnFeaturesPerLevel
  int regW, regH, nIni;      // DistributeOctTree region and root nodes
  float hX;
  int key_off, key_cap;  // candidate-key arena of this level inside the per-image arena
  int sel_off, ncap;     // selected-key arena / node capacity
  int xtab_off, ytab_off;
  float scale;
  int patch;
  int tile_begin, tile_end, tiles_x;
};

struct OrbParams {
  int nlevels, cell_cap, ncells, keys_per_image, sel_per_image, kp_cap;
  int umax[kHalfPatch + 1];
  LevelDesc lv[kMaxLevels];
};

struct CellDesc {
  short level, x0, y0, cw, ch, offx, offy, pad;
};

struct ImgSet {  // where the planes of a batch live
  const uint8_t* img0;
  int stride0;
  size_t img_pitch;
  uint8_t* pyr;
  size_t pyr_img;
  uint8_t* blur;
  size_t blur_img;
};

__device__ __forceinline__ const uint8_t* plane_ptr(const OrbParams& P, const ImgSet& I, int b,
                                                    int l, int* pitch) {
  if (l == 0) {
    *pitch = I.stride0;
    return I.img0 + (size_t)b * I.img_pitch;
  }
  *pitch = P.lv[l].pitch;
  return I.pyr + (size_t)b * I.pyr_img + P.lv[l].off;
}

struct BlurTile {
  short level, tx, ty, pad;
};

// HIP-event stamps around every stage of a batch call, on the extractor's own stream.  A ring of
// kTimingRing steps is kept so a bench can time K steps without a host sync in between.
static const int kTimingRing = 64;
struct Timing {
  bool on = false;
  hipEvent_t ev[kTimingRing][VIEO_ORB_NSTAGES] = {};
  long steps = 0;  // batch calls stamped since timing was enabled
};

int orb_create_with_priority(vieo_orb** out, int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th, int priority);

}  // namespace vieo

struct vieo_orb {  // global-scope tag declared in include/vieo_hot.h
  int nfeatures, nlevels, iniTh, minTh;
  double scaleFactor;  // ORBextractor.h:67 keeps the float argument in a double member
  std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
  std::vector<int> feats;
  int umax[vieo::kHalfPatch + 1];
  hipStream_t stream = nullptr;
  // geometry the buffers are currently sized for
  int w = 0, h = 0, B = 0;
  vieo::OrbParams P;
  std::vector<vieo::CellDesc> cells;
  std::vector<vieo::BlurTile> tiles;
  int tpitch = 0, fast_cand_cap = 0, tile_bytes = 0, score_bytes = 0, fast_lds = 0, qt_lds = 0, ncap_max = 0, scap_max = 0;
  int resize_pitch = 0, resize_lds = 0;
  int resize2_pitch = 0, resize2_l1_off = 0, resize2_lds = 0;  // k_resize2: source band pitch, offset of the level-l region, LDS bytes
  bool resize2_ok = false;                                     // the pairs' regions fit (else: one level per launch)
  size_t pyr_img = 0, blur_img = 0;
  vieo::DevBuf d_pyr, d_blur, d_cells, d_tiles, d_xtab, d_ytab, d_cell_keys, d_cell_counts, d_keys,
      d_kslot, d_kq, d_sel, d_sel_count, d_pattern, d_in, d_kp, d_desc, d_counts, d_tmp_kp,
      d_tmp_desc, d_tmp_counts, d_krec;  // d_krec: [image][kp_cap] (key, level) in output order
  int out_cap = 0;  // capacity of the internal (host-API) output buffers
  int last_B = 0;
  vieo::ImgSet last_imgs{};
  vieo::Timing tm;
  // ---- the resident frame (round 5).  vieo_orb_extract leaves the image's keys, descriptors and pyramid in HBM; the
  // *_resident entries (stereo matcher, projection searches) read them there instead of taking them back from the host:
  // Frame::Frame -> ComputeStereoMatches -> SearchByProjection x 2 of one frame upload only what the pointer graph forces
  // (last-frame points, window queries, taken flags).  `epoch` counts the host-API extractions of this handle.
  vieo::PinnedBuf h_in, h_res, h_io;  // pinned staging: the image up, counts | keys | descriptors back, small call blocks
  vieo::DevBuf d_io;                  // device twin of h_io
  unsigned long long epoch = 0;
  int res_n = -1, res_mono = 0;       // keys of the resident frame (-1: none)
  vieo_keypoint res_sample[16];       // its first 8 / last 8 keys: the identity test of vieo_orb_holds
  vieo::DevBuf d_uright, d_depth, d_sad;  // written by vieo_stereo_match_rectified_resident (uright: read by the searches)
  unsigned long long uright_epoch = ~0ull;
  std::vector<float> uright_host;  // what the resident matcher returned: a caller that passes these values back is resident
  vieo::DevBuf g_start, g_rec, g_ang;  // Frame::mGrid as a CSR, built by the frame's first resident search
  unsigned long long grid_epoch = ~0ull;
};

