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
    // a buffer that GROWS gets half as much again (at most 64 MB of slack): a local-BA arena grows with its map, and an
    // exact fit was a hipFree (a device synchronisation) + hipMalloc on every call, 0.4 ms of a 3.6 ms window
    if (p) {
      const size_t slack = bytes / 2;
      bytes += slack < ((size_t)64 << 20) ? slack : ((size_t)64 << 20);
      (void)hipFree(p);
    }
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
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr, cap = 0;
  }
};

// Several host arrays -> one pinned block -> one device block: ONE asynchronous copy up and one back per call.  The
// host-pointer entry points of the small per-frame calls (one frame's search, one pose optimisation) spent a third of
// their time in five to nine synchronous copies of pageable memory.
struct Staging {
  PinnedBuf pin;
  DevBuf dev;
  struct Item {
    const void* src;
    size_t bytes, off;
  };
  Item items[16];
  int n = 0;
  size_t used = 0, in_end = 0;
  static size_t al(size_t v) { return (v + 255) & ~(size_t)255; }
  void reset() { n = 0, used = 0, in_end = 0; }
  size_t in(const void* src, size_t bytes) {  // inputs first, in call order
    const size_t off = used;
    items[n++] = Item{src, bytes, off};
    used += al(bytes);
    in_end = used;
    return off;
  }
  size_t out(size_t bytes) {  // device-written regions, after the inputs
    const size_t off = used;
    used += al(bytes);
    return off;
  }
  int upload(hipStream_t st) {
    int rc;
    if ((rc = pin.ensure(used)) != VIEO_OK || (rc = dev.ensure(used)) != VIEO_OK) return rc;
    for (int i = 0; i < n; i++)
      if (items[i].bytes) memcpy((uint8_t*)pin.p + items[i].off, items[i].src, items[i].bytes);
    if (in_end) VIEO_HIP_CHECK(hipMemcpyAsync(dev.p, pin.p, in_end, hipMemcpyHostToDevice, st));
    return VIEO_OK;
  }
  int download(size_t from, hipStream_t st) {  // [from, used) back to the pinned block, then wait
    if (used > from)
      VIEO_HIP_CHECK(hipMemcpyAsync((uint8_t*)pin.p + from, (uint8_t*)dev.p + from, used - from, hipMemcpyDeviceToHost, st));
    VIEO_HIP_CHECK(hipStreamSynchronize(st));
    return VIEO_OK;
  }
  template <class T>
  T* d(size_t off) const {
    return (T*)((uint8_t*)dev.p + off);
  }
  const void* h(size_t off) const { return (const uint8_t*)pin.p + off; }
};

int require_device();  // VIEO_OK or VIEO_E_NO_DEVICE
// which pose-optimisation kernels a *_batch_device call launches: bit 0 the rectified-pinhole instance,
// bit 1 the multi-camera-rig instance (vieo_pose_set_camera_mode)
int pose_rig_launches();
// the same for the visual-inertial kernel's encoder instances: bit 0 frames without an encoder measurement,
// bit 1 frames with one (vieo_pose_set_encoder_mode)
int pose_enc_launches();
int pose_launch_mask(int mode);  // VIEO_POSE_CAMS_* / VIEO_POSE_ENC_* -> bit mask of the kernel instances to launch

}  // namespace vieo
