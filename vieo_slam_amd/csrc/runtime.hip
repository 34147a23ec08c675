// runtime.hip -- error reporting, device probing and the small memory helpers of the C-ABI.
#include <atomic>
#include <mutex>

#include "common.h"

namespace vieo {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// The arch check is cached per device id: the per-step *_batch_device launches call this on the hot path.
int require_device() {
  static std::atomic<int> ok_dev[64];  // 0 unknown, 1 gfx950
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && ok_dev[dev].load(std::memory_order_relaxed) == 1)
    return VIEO_OK;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_error("no HIP device visible (%s); libvieo_hot has no CPU fallback",
              e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return VIEO_E_NO_DEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    set_error("cannot query HIP device");
    return VIEO_E_NO_DEVICE;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("device arch %s is not gfx950; kernels are built for gfx950 only", prop.gcnArchName);
    return VIEO_E_NO_DEVICE;
  }
  if (dev >= 0 && dev < 64) ok_dev[dev].store(1, std::memory_order_relaxed);
  return VIEO_OK;
}

// Per HOST THREAD (round 3): as process-wide values a LocalMapping-side call between another thread's set and reset
// silently skipped that thread's frames of the other kind.  New callers pass the modes per call (the _ex entries).
static thread_local int g_pose_cams_mode = VIEO_POSE_CAMS_AUTO;
int pose_launch_mask(int mode) { return mode == 1 ? 1 : mode == 2 ? 2 : 3; }
int pose_rig_launches() { return pose_launch_mask(g_pose_cams_mode); }
static thread_local int g_pose_enc_mode = VIEO_POSE_ENC_AUTO;
int pose_enc_launches() { return pose_launch_mask(g_pose_enc_mode); }

}  // namespace vieo

extern "C" {

const char* vieo_last_error(void) { return vieo::g_err; }
const char* vieo_version(void) { return "vieo_hot 0.1 (gfx950)"; }

int vieo_device_available(void) { return vieo::require_device() == VIEO_OK ? 1 : 0; }

// The HIP "current device" is a per-thread setting: a host thread that did not create the process's
// context (a LocalMapping-side worker) starts on device 0 and must select its GPU itself.
int vieo_set_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
    vieo::set_error("vieo_set_device: device %d of %d", device, n);
    return VIEO_E_INVALID;
  }
  VIEO_HIP_CHECK(hipSetDevice(device));
  return VIEO_OK;
}

int vieo_pose_set_camera_mode(int mode) {
  if (mode < VIEO_POSE_CAMS_AUTO || mode > VIEO_POSE_CAMS_RIG) return VIEO_E_INVALID;
  vieo::g_pose_cams_mode = mode;
  return VIEO_OK;
}

int vieo_pose_set_encoder_mode(int mode) {
  if (mode < VIEO_POSE_ENC_AUTO || mode > VIEO_POSE_ENC_ALL) return VIEO_E_INVALID;
  vieo::g_pose_enc_mode = mode;
  return VIEO_OK;
}

int vieo_get_device(void) {
  int dev = -1;
  return hipGetDevice(&dev) == hipSuccess ? dev : -1;
}

int vieo_dev_malloc(void** d_ptr, size_t bytes) {
  if (!d_ptr) return VIEO_E_INVALID;
  int rc = vieo::require_device();
  if (rc != VIEO_OK) return rc;
  VIEO_HIP_CHECK(hipMalloc(d_ptr, bytes ? bytes : 1));
  return VIEO_OK;
}
int vieo_dev_free(void* d_ptr) {
  if (d_ptr) VIEO_HIP_CHECK(hipFree(d_ptr));
  return VIEO_OK;
}
int vieo_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes) {
  VIEO_HIP_CHECK(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
  return VIEO_OK;
}
int vieo_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes) {
  VIEO_HIP_CHECK(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
  return VIEO_OK;
}
int vieo_event_create(void** ev) {
  if (!ev) return VIEO_E_INVALID;
  hipEvent_t e;
  VIEO_HIP_CHECK(hipEventCreate(&e));
  *ev = (void*)e;
  return VIEO_OK;
}
int vieo_event_destroy(void* ev) {
  if (ev) VIEO_HIP_CHECK(hipEventDestroy((hipEvent_t)ev));
  return VIEO_OK;
}
int vieo_event_record(void* ev, void* stream) {
  VIEO_HIP_CHECK(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
  return VIEO_OK;
}
int vieo_event_elapsed_ms(void* ev0, void* ev1, float* ms) {
  VIEO_HIP_CHECK(hipEventSynchronize((hipEvent_t)ev1));
  VIEO_HIP_CHECK(hipEventElapsedTime(ms, (hipEvent_t)ev0, (hipEvent_t)ev1));
  return VIEO_OK;
}
int vieo_device_synchronize(void) {
  VIEO_HIP_CHECK(hipDeviceSynchronize());
  return VIEO_OK;
}

// ---- streams, pinned host memory and asynchronous copies: what a host needs to feed frames over PCIe while the
// previous batch is being processed (bench.py's PCIe-inclusive leg, the single-stream replay)
int vieo_stream_create(void** stream) {
  if (!stream) return VIEO_E_INVALID;
  int rc = vieo::require_device();
  if (rc != VIEO_OK) return rc;
  hipStream_t s;
  VIEO_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = (void*)s;
  return VIEO_OK;
}
int vieo_stream_destroy(void* stream) {
  if (stream) VIEO_HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
  return VIEO_OK;
}
int vieo_stream_synchronize(void* stream) {
  VIEO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return VIEO_OK;
}
int vieo_stream_wait_event(void* stream, void* ev) {
  VIEO_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0));
  return VIEO_OK;
}
int vieo_host_alloc_pinned(void** h_ptr, size_t bytes) {
  if (!h_ptr) return VIEO_E_INVALID;
  int rc = vieo::require_device();
  if (rc != VIEO_OK) return rc;
  VIEO_HIP_CHECK(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault));
  return VIEO_OK;
}
int vieo_host_free_pinned(void* h_ptr) {
  if (h_ptr) VIEO_HIP_CHECK(hipHostFree(h_ptr));
  return VIEO_OK;
}
int vieo_memcpy_h2d_async(void* d_dst, const void* h_src, size_t bytes, void* stream) {
  VIEO_HIP_CHECK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return VIEO_OK;
}
int vieo_memcpy_d2h_async(void* h_dst, const void* d_src, size_t bytes, void* stream) {
  VIEO_HIP_CHECK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return VIEO_OK;
}

}  // extern "C"
