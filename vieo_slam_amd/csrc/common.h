// common.h -- shared host-side helpers of libvieo_hot.so (gfx950 only; no CPU fallback paths).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/vieo_hot.h"

namespace vieo {

void set_error(const char* fmt, ...);

#define VIEO_HIP_CHECK(expr)                                                             \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      vieo::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
      return VIEO_E_HIP;                                                                 \
    }                                                                                    \
  } while (0)

static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }
static inline size_t align_up_sz(size_t v, size_t a) { return (v + a - 1) / a * a; }

// One growable device allocation.  It remembers the device it lives on: the thread_local scratch caches of the
// entry points are re-created when the calling thread has moved to another GPU (vieo_set_device).
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int dev = -1;
  int ensure(size_t bytes) {
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (p && dev != cur) {  // allocated on another device: free it there, start again here
      (void)hipSetDevice(dev);
      (void)hipFree(p);
      (void)hipSetDevice(cur);
      p = nullptr, cap = 0;
    }
    if (bytes <= cap) return VIEO_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    VIEO_HIP_CHECK(hipMalloc(&p, bytes));
    cap = bytes, dev = cur;
    return VIEO_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const {
    return (T*)p;
  }
};

// pinned host staging (grows by half again, never shrinks): one H2D / D2H per call instead of one per array
struct PinnedBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return VIEO_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr, cap = 0;
    bytes += bytes / 2;
    VIEO_HIP_CHECK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    cap = bytes;
    return VIEO_OK;
  }
};

int require_device();  // VIEO_OK or VIEO_E_NO_DEVICE
// which pose-optimisation kernels a *_batch_device call launches: bit 0 the rectified-pinhole instance,
// bit 1 the multi-camera-rig instance (vieo_pose_set_camera_mode)
int pose_rig_launches();
// the same for the visual-inertial kernel's encoder instances: bit 0 frames without an encoder measurement,
// bit 1 frames with one (vieo_pose_set_encoder_mode)
int pose_enc_launches();

}  // namespace vieo
