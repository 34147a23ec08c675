// cabi_demo.cc -- a host program WITHOUT Python: links libvieo_hot.so and drives the hot path through the C-ABI of
// include/vieo_hot.h the way the C++ shims (shim/*.cc) do.  Built by __graft_entry__.build():
//   g++ -std=c++17 -Iinclude examples/cabi_demo.cc -o examples/cabi_demo -Lvieo_slam_amd -lvieo_hot -Wl,-rpath,$ORIGIN/../vieo_slam_amd
// Run on a gfx950 box: ./examples/cabi_demo  (prints one line per call, exit code 0 on success).
//
//   1. vieo_orb_extract on a synthetic 752x480 image (ORBextractor::operator(), ORBextractor.cc:968-1058)
//   2. vieo_hamming_knn2 of the descriptors against themselves (distance 0 to itself)
//   3. vieo_pose_optimization_vio on a noiseless synthetic frame: the perturbed pose returns to the truth
//      (Optimizer::PoseOptimization<KeyFrame>, Optimizer.h:208-816)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vieo_hot.h"

static unsigned rng_state = 12345u;
static double urand() {  // xorshift, deterministic
  rng_state ^= rng_state << 13, rng_state ^= rng_state >> 17, rng_state ^= rng_state << 5;
  return (rng_state & 0xFFFFFF) / double(0x1000000);
}

#define CHECK(call)                                                                     \
  do {                                                                                  \
    const int rc_ = (call);                                                             \
    if (rc_ != VIEO_OK) {                                                               \
      std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, vieo_last_error());      \
      return 1;                                                                         \
    }                                                                                   \
  } while (0)

int main() {
  if (!vieo_device_available()) {
    std::fprintf(stderr, "no gfx950 device: %s (there is no CPU fallback)\n", vieo_last_error());
    return 2;
  }
  std::printf("%s\n", vieo_version());
  // ---- 1. extractor: blocks of random grey levels give plenty of corners
  const int W = 752, H = 480;
  std::vector<uint8_t> img((size_t)W * H);
  for (int by = 0; by < H; by += 12)
    for (int bx = 0; bx < W; bx += 12) {
      const uint8_t g = (uint8_t)(20 + 215 * urand());
      for (int y = by; y < by + 12 && y < H; ++y)
        for (int x = bx; x < bx + 12 && x < W; ++x) img[(size_t)y * W + x] = g;
    }
  vieo_orb* ext = nullptr;
  CHECK(vieo_orb_create(&ext, 1200, 1.2f, 8, 20, 7));
  const int cap = vieo_orb_max_keypoints(ext);
  std::vector<vieo_keypoint> kps(cap);
  std::vector<uint8_t> desc((size_t)cap * 32);
  int n = 0, mono = 0;
  CHECK(vieo_orb_extract(ext, img.data(), W, H, W, nullptr, kps.data(), desc.data(), cap, &n, &mono));
  std::printf("vieo_orb_extract: %d keypoints, monoIndex %d, first (%.1f, %.1f) octave %d angle %.1f\n", n, mono,
              kps[0].x, kps[0].y, kps[0].octave, kps[0].angle);
  if (n < 600) return 3;
  // ---- 2. knn-2 against itself
  std::vector<int32_t> idx((size_t)n * 2), dist((size_t)n * 2);
  CHECK(vieo_hamming_knn2(desc.data(), n, desc.data(), n, idx.data(), dist.data()));
  int self = 0;
  for (int i = 0; i < n; ++i) self += dist[2 * i] == 0;
  std::printf("vieo_hamming_knn2: %d of %d rows find a zero-distance neighbour\n", self, n);
  if (self != n) return 4;
  vieo_orb_destroy(ext);
  // ---- 3. visual-inertial pose optimisation on a noiseless frame (no IMU measurement: dt = 0, bias edge only)
  vieo_vio_frame F;
  std::memset(&F, 0, sizeof(F));
  vieo_pose_frame& B = F.base;
  B.nav.q[0] = 1, F.nav_last.q[0] = 1, F.nav_prior.q[0] = 1;
  B.Rcb[0] = B.Rcb[4] = B.Rcb[8] = 1;  // camera = body
  B.fx = B.fy = 435.2f, B.cx = 367.45f, B.cy = 252.2f, B.bf = 47.9f;
  F.inv_sigma_bg2 = 1e6, F.inv_sigma_ba2 = 1e4, F.dt_frames = 0.05, F.th_depth = 35.f;
  F.gw[2] = -9.81;
  const int n_obs = 300;
  std::vector<vieo_pose_obs> obs(n_obs);
  for (int i = 0; i < n_obs; ++i) {
    const double X = 8 * urand() - 4, Y = 5 * urand() - 2.5, Z = 2 + 8 * urand();
    vieo_pose_obs& o = obs[i];
    o.Xw[0] = (float)X, o.Xw[1] = (float)Y, o.Xw[2] = (float)Z;
    // the observation is the projection of the FLOAT point from the true pose (identity)
    o.u = B.fx * (o.Xw[0] / o.Xw[2]) + B.cx, o.v = B.fy * (o.Xw[1] / o.Xw[2]) + B.cy;
    o.ur = (i % 3) ? o.u - B.bf / o.Xw[2] : -1.f;
    o.inv_sigma2 = 1.f, o.flags = 0;
  }
  B.n_obs = n_obs;
  B.nav.p[0] = 0.02, B.nav.p[1] = -0.015, B.nav.p[2] = 0.01;  // start 2.7 cm off
  std::vector<uint8_t> outl(n_obs);
  vieo_vio_result R;
  CHECK(vieo_pose_optimization_vio(&F, obs.data(), outl.data(), &R));
  const double err = std::sqrt(R.base.nav.p[0] * R.base.nav.p[0] + R.base.nav.p[1] * R.base.nav.p[1] +
                               R.base.nav.p[2] * R.base.nav.p[2]);
  std::printf("vieo_pose_optimization_vio: %d inliers of %d, %d LM iterations, |p - truth| = %.2e m\n",
              R.base.n_inliers, n_obs, R.base.lm_iterations, err);
  if (R.base.status != VIEO_POSE_OK || R.base.n_inliers < n_obs - 5 || err > 1e-4) return 5;
  std::printf("cabi_demo ok\n");
  return 0;
}
