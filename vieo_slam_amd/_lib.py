"""ctypes binding of libvieo_hot.so (the C-ABI of include/vieo_hot.h).

There is no CPU fallback: if the shared object is missing, or no gfx950 device is visible when a
compute entry point is used, this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VIEO_LIB_PATH") or os.path.join(_HERE, "libvieo_hot.so")  # (the override: A/B runs of two builds)

VIEO_OK = 0
VIEO_E_INVALID, VIEO_E_NO_DEVICE, VIEO_E_HIP, VIEO_E_CAPACITY, VIEO_E_EMPTY = -1, -2, -3, -4, -5

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_f = ctypes.c_float
c_sz = ctypes.c_size_t
P = ctypes.POINTER

_SIGS = {
    "vieo_last_error": (ctypes.c_char_p, []),
    "vieo_device_available": (c_i, []),
    "vieo_set_device": (c_i, [c_i]),
    "vieo_pose_set_camera_mode": (c_i, [c_i]),
    "vieo_pose_set_replicas": (c_i, [c_i]),
    "vieo_search_for_triangulation": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p]),
    "vieo_pose_set_encoder_mode": (c_i, [c_i]),
    "vieo_is_in_frustum_batch": (c_i, [c_p, c_p, c_i, c_p]),
    "vieo_imu_preintegrate_batch": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p]),
    "vieo_fuse_search": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p]),
    "vieo_distinctive_descriptors_batch": (c_i, [c_p, c_p, c_i, c_p]),
    "vieo_update_normal_and_depth_batch": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_p, ctypes.c_float, c_i, c_p, c_p,
                                                 c_p]),
    "vieo_bundle_adjustment": (c_i, [c_p, c_i, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p]),
    "vieo_bundle_adjustment_enc": (c_i, [c_p, c_i, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p]),
    "vieo_local_bundle_adjustment_batch_enc": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                                      c_p]),
    "vieo_local_bundle_adjustment_enc": (c_i, [c_p, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "vieo_global_bundle_adjustment_vio": (c_i, [c_p, c_i, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p,
                                                c_p]),
    "vieo_global_bundle_adjustment_vio_scale": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p,
                                                      c_p, c_p, c_p]),
    "vieo_global_bundle_adjustment_vio_sharded_scale": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i,
                                                              c_p, ctypes.c_size_t, c_p, c_p, c_p, c_p, c_p, c_p]),
    "vieo_track_local_queries_batch_device": (c_i, [c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_f, c_f, c_p, c_p,
                                                    c_p, c_sz, c_p, c_p]),
    "vieo_pose_optimization_vio_batch_device_ex": (c_i, [c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_p]),
    "vieo_pose_optimization_batch_device_ex": (c_i, [c_p, c_i, c_p, c_p, c_p, c_i, c_p]),
    "vieo_get_device": (c_i, []),
    "vieo_version": (ctypes.c_char_p, []),
    "vieo_dev_malloc": (c_i, [P(c_p), c_sz]),
    "vieo_dev_free": (c_i, [c_p]),
    "vieo_memcpy_h2d": (c_i, [c_p, c_p, c_sz]),
    "vieo_memcpy_d2h": (c_i, [c_p, c_p, c_sz]),
    "vieo_device_synchronize": (c_i, []),
    "vieo_stream_create": (c_i, [P(c_p)]),
    "vieo_stream_destroy": (c_i, [c_p]),
    "vieo_stream_synchronize": (c_i, [c_p]),
    "vieo_stream_wait_event": (c_i, [c_p, c_p]),
    "vieo_host_alloc_pinned": (c_i, [P(c_p), c_sz]),
    "vieo_host_free_pinned": (c_i, [c_p]),
    "vieo_memcpy_h2d_async": (c_i, [c_p, c_p, c_sz, c_p]),
    "vieo_memcpy_d2h_async": (c_i, [c_p, c_p, c_sz, c_p]),
    "vieo_event_create": (c_i, [P(c_p)]),
    "vieo_event_destroy": (c_i, [c_p]),
    "vieo_event_record": (c_i, [c_p, c_p]),
    "vieo_event_elapsed_ms": (c_i, [c_p, c_p, P(c_f)]),
    "vieo_orb_create": (c_i, [P(c_p), c_i, c_f, c_i, c_i, c_i]),
    "vieo_orb_destroy": (None, [c_p]),
    "vieo_orb_levels": (c_i, [c_p]),
    "vieo_orb_scale_factor": (c_f, [c_p]),
    "vieo_orb_scale_factors": (c_i, [c_p, c_p]),
    "vieo_orb_inv_scale_factors": (c_i, [c_p, c_p]),
    "vieo_orb_level_sigma2": (c_i, [c_p, c_p]),
    "vieo_orb_inv_level_sigma2": (c_i, [c_p, c_p]),
    "vieo_orb_features_per_level": (c_i, [c_p, c_p]),
    "vieo_orb_max_keypoints": (c_i, [c_p]),
    "vieo_orb_extract": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_i, P(c_i), P(c_i)]),
    "vieo_orb_extract_batch_device": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_sz, c_p, c_p, c_p, c_i, c_p]),
    "vieo_orb_sync": (c_i, [c_p]),
    "vieo_orb_level_size": (c_i, [c_p, c_i, P(c_i), P(c_i)]),
    "vieo_orb_get_level": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i]),
    "vieo_orb_level_device": (c_i, [c_p, c_i, c_i, P(c_p), P(c_i)]),
    "vieo_orb_last_stage_ms": (c_i, [c_p, c_p]),
    "vieo_orb_enable_timing": (c_i, [c_p, c_i]),
    "vieo_orb_timed_steps": (c_i, [c_p]),
    "vieo_orb_stage_ms": (c_i, [c_p, c_i, c_p]),
    "vieo_hamming_knn2": (c_i, [c_p, c_i, c_p, c_i, c_p, c_p]),
    "vieo_stereo_fisheye_match": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "vieo_hamming_knn2_rig_batch_device": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    "vieo_fisheye_create": (c_i, [P(c_p), c_p, c_i, c_i]),
    "vieo_fisheye_destroy": (None, [c_p]),
    "vieo_fisheye_group_capacity": (c_i, [c_p]),
    "vieo_stereo_fisheye_match_batch_device": (c_i, [c_p, c_p, c_p, c_p, c_i] + [c_p] * 12),
    "vieo_stereo_fisheye_match_batch_device_part": (c_i, [c_p, c_p, c_p, c_p, c_i] + [c_p] * 11 + [c_i, c_p]),
    "vieo_track_merge_assign_rig_batch_device": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p]),
    "vieo_track_compact_queries_batch_device": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p]),
    "vieo_track_build_obs_rig_batch_device": (c_i, [c_p, c_p, c_p, c_f, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p,
                                                     c_p, c_i, c_p]),
    "vieo_fisheye_last_walk": (None, [c_p, c_p]),
    "vieo_hamming_knn2_batch_device": (c_i, [c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_p]),
    "vieo_stereo_match_rectified": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_f, c_f, c_p, c_p]),
    "vieo_stereo_match_rectified_batch_device": (c_i, [c_p, c_i, c_p, c_p, c_p, c_i, c_f, c_f, c_p, c_p]),
    "vieo_sbp_project_last_frame": (c_i, [c_p, c_i, c_p, c_p]),
    "vieo_sbp_project_last_frame_rig": (c_i, [c_p, c_i, c_p, c_p, c_p]),
    "vieo_sbp_project_keyframe": (c_i, [c_p, c_i, c_p, c_p, c_f, c_p]),
    "vieo_search_by_projection_rig": (c_i, [c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_f, c_i, c_p, c_p]),
    "vieo_sbp_keep_grid": (c_i, [c_i]),
    "vieo_orb_holds": (c_i, [c_p, c_p, c_i]),
    "vieo_orb_resident_keys": (c_i, [c_p]),
    "vieo_stereo_match_rectified_resident": (c_i, [c_p, c_p, c_f, c_f, c_p, c_p]),
    "vieo_search_by_projection_last_frame_resident": (c_i, [c_p, c_p, c_i, c_p, c_p, c_f, c_i, c_p, c_p]),
    "vieo_search_by_projection_resident": (c_i, [c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_f, c_i, c_p, c_p]),
    "vieo_sbp_project_last_frame_rig_batch_device": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p, c_i, c_p, c_p]),
    "vieo_search_by_projection_rig_batch_device": (c_i, [c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i,
                                                          c_f, c_i, c_p, c_p, c_p]),
    "vieo_search_by_projection": (c_i, [c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_f, c_i, c_p, c_p]),
    "vieo_search_by_projection_batch_device": (c_i, [c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_f, c_i, c_p, c_p, c_p]),
    "vieo_sbp_project_last_frame_batch_device": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p, c_p]),
    "vieo_pose_optimization": (c_i, [c_p, c_p, c_p, c_p]),
    "vieo_pose_optimization_batch_device": (c_i, [c_p, c_i, c_p, c_p, c_p, c_p]),
    "vieo_pose_optimization_vio": (c_i, [c_p, c_p, c_p, c_p]),
    "vieo_pose_optimization_vio_batch_device": (c_i, [c_p, c_i, c_p, c_p, c_p, c_p]),
    "vieo_track_merge_assign_batch_device": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "vieo_track_build_obs_batch_device": (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p]),
    "vieo_track_after_pose_batch_device": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    "vieo_track_build_obs_depth_batch_device": (c_i, [c_p, c_p, c_p, c_f, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p,
                                                       c_p, c_i, c_p]),
    "vieo_track_mark_held_batch_device": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "vieo_track_after_pose_held_batch_device": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_p, c_i, c_p]),
    "vieo_track_merge_build_obs_batch_device": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_f, c_i, c_p, c_p, c_p, c_p, c_i,
                                                       c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p]),
    "vieo_track_local_queries_device": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_f, c_p, c_p, c_p, c_p, c_p]),
    "vieo_orb_stream": (c_p, [c_p]),
    "vieo_local_bundle_adjustment": (c_i, [c_p, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p]),
    "vieo_local_bundle_adjustment_batch": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                                  c_p]),
    "vieo_local_bundle_adjustment_vio": (c_i, [c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_p,
                                                c_p, c_p]),
    "vieo_local_bundle_adjustment_vio_batch": (c_i, [c_i] + [c_p] * 15),
    "vieo_lba_sharded_buffer_doubles": (ctypes.c_size_t, [c_i, c_p]),
    "vieo_lba_enable_timing": (None, [c_i]),
    "vieo_lba_set_stream_priority": (c_i, [c_i]),
    "vieo_lba_kernel_classes": (c_i, []),
    "vieo_lba_kernel_class_name": (ctypes.c_char_p, [c_i]),
    "vieo_lba_kernel_times": (None, [c_p, c_p, c_p]),
    "vieo_rccl_available": (c_i, []),
    "vieo_rccl_unique_id": (c_i, [c_p]),
    "vieo_rccl_comm_create": (c_i, [P(c_p), c_p, c_i, c_i]),
    "vieo_rccl_comm_destroy": (c_i, [c_p]),
    "vieo_global_bundle_adjustment_vio_sharded": (c_i, [c_p, c_i, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p,
                                                        ctypes.c_size_t, c_p, c_p, c_p, c_p, c_p]),
    "vieo_local_bundle_adjustment_vio_sharded": (c_i, [c_i] + [c_p] * 10 + [c_p, ctypes.c_size_t, c_p, c_p] +
                                                 [c_p] * 4),
    "vieo_local_bundle_adjustment_vio_sharded_stop": (c_i, [c_i] + [c_p] * 10 + [c_p, ctypes.c_size_t, c_p, c_p, c_p] +
                                                      [c_p] * 4),
    "vieo_orb_tap_plane": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i]),
    "vieo_orb_tap_candidates": (c_i, [c_p, c_i, c_i, c_p, c_i]),
    "vieo_orb_tap_level_keys": (c_i, [c_p, c_i, c_i, c_p, c_i]),
    "vieo_imu_preintegrate_batch_device": (c_i, [c_p] * 7 + [c_i] + [c_p] * 4),
    "vieo_tracker_create": (c_i, [P(c_p), c_p]),
    "vieo_tracker_create_rig": (c_i, [P(c_p), c_p, c_p]),
    "vieo_tracker_key_capacity": (c_i, [c_p]),
    "vieo_tracker_group_capacity": (c_i, [c_p]),
    "vieo_tracker_image_buffer": (c_i, [c_p, c_i, P(c_p)]),
    "vieo_tracker_destroy": (None, [c_p]),
    "vieo_tracker_image_buffers": (c_i, [c_p, P(c_p), P(c_p)]),
    "vieo_tracker_scale_factors": (c_i, [c_p, c_p]),
    "vieo_track_frame": (c_i, [c_p, c_p, c_p]),
    "vieo_tracker_get_stats": (c_i, [c_p, c_p]),
    "vieo_tracker_reprobe": (c_i, [c_p]),
    "vieo_tracker_get_level": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i]),
}

_lib = None


class VieoError(RuntimeError):
    pass


def declared_symbols():
    """Every extern "C" symbol include/vieo_hot.h declares (used by the ABI export test)."""
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VieoError(
                "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != VIEO_OK:
        msg = lib().vieo_last_error().decode(errors="replace")
        raise VieoError("%s failed (%d): %s" % (what, rc, msg))


class DeviceBuffer:
    """Plain HBM allocation owned through the C-ABI (no torch needed)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = c_p()
        check(lib().vieo_dev_malloc(ctypes.byref(p), self.nbytes), "vieo_dev_malloc")
        self.ptr = p.value

    def upload(self, arr):
        import numpy as np
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        check(lib().vieo_memcpy_h2d(self.ptr, a.ctypes.data, a.nbytes), "h2d")

    def download(self, dtype, shape):
        import numpy as np
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        check(lib().vieo_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes), "d2h")
        return out

    def free(self):
        if self.ptr:
            lib().vieo_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
