"""Host-side mirror of VIEO_SLAM::ORBextractor (reference include/ORBextractor.h:27-80) on top of
the C-ABI.  Same constructor arguments, same call semantics (returns the reference's monoIndex),
same getters; numpy arrays stand in for cv::Mat / std::vector<cv::KeyPoint>.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import c_i, c_p, check, lib

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KEYPOINT_DTYPE.itemsize == 28

STAGES = ("pyramid", "fast", "quadtree", "blur", "describe", "total")


class ORBextractor:
    """ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)"""

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST):
        h = c_p()
        check(lib().vieo_orb_create(ctypes.byref(h), nfeatures, scaleFactor, nlevels, iniThFAST,
                                    minThFAST), "vieo_orb_create")
        self._h = h
        self.nlevels = nlevels

    def close(self):
        if getattr(self, "_h", None):
            lib().vieo_orb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- getters (include/ORBextractor.h:42-52)
    def GetLevels(self):
        return lib().vieo_orb_levels(self._h)

    def GetScaleFactor(self):
        return lib().vieo_orb_scale_factor(self._h)

    def _tab(self, fn, dtype=np.float32):
        out = np.zeros(self.nlevels, dtype)
        check(fn(self._h, out.ctypes.data))
        return out

    def GetScaleFactors(self):
        return self._tab(lib().vieo_orb_scale_factors)

    def GetInverseScaleFactors(self):
        return self._tab(lib().vieo_orb_inv_scale_factors)

    def GetScaleSigmaSquares(self):
        return self._tab(lib().vieo_orb_level_sigma2)

    def GetInverseScaleSigmaSquares(self):
        return self._tab(lib().vieo_orb_inv_level_sigma2)

    def features_per_level(self):
        return self._tab(lib().vieo_orb_features_per_level, np.int32)

    def max_keypoints(self):
        return lib().vieo_orb_max_keypoints(self._h)

    # ---- operator() (src/ORBextractor.cc:968-1058)
    def __call__(self, image, mask=None, pvLappingArea=None):
        """returns (monoIndex, keypoints[n] (KEYPOINT_DTYPE), descriptors[n,32] uint8).
        monoIndex is -1 for an empty image, as in the reference."""
        if image is None or image.size == 0:
            return -1, np.zeros(0, KEYPOINT_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2  # CV_8UC1 assert, ORBextractor.cc:973
        img = image if image.strides[1] == 1 and image.strides[0] >= image.shape[1] else np.ascontiguousarray(image)
        cap = self.max_keypoints()
        kps = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n, mono = c_i(), c_i()
        lap = None
        if pvLappingArea is not None:
            lap = (c_i * 2)(int(pvLappingArea[0]), int(pvLappingArea[1]))
        check(lib().vieo_orb_extract(self._h, img.ctypes.data, img.shape[1], img.shape[0],
                                     img.strides[0], lap, kps.ctypes.data, desc.ctypes.data, cap,
                                     ctypes.byref(n), ctypes.byref(mono)), "vieo_orb_extract")
        return mono.value, kps[:n.value].copy(), desc[:n.value].copy()

    # ---- the resident frame (include/vieo_hot.h): does the handle still hold these keys?
    def holds(self, keypoints):
        k = np.ascontiguousarray(keypoints)
        return bool(lib().vieo_orb_holds(self._h, k.ctypes.data, len(k)))

    def resident_keys(self):
        return lib().vieo_orb_resident_keys(self._h)

    # ---- mvImagePyramid (include/ORBextractor.h:54)
    def level_size(self, level):
        w, h = c_i(), c_i()
        check(lib().vieo_orb_level_size(self._h, level, ctypes.byref(w), ctypes.byref(h)))
        return w.value, h.value

    def image_pyramid(self, level, image_index=0, with_border=False):
        w, h = self.level_size(level)
        b = 38 if with_border else 0
        out = np.zeros((h + b, w + b), np.uint8)
        check(lib().vieo_orb_get_level(self._h, image_index, level, int(with_border),
                                       out.ctypes.data, out.strides[0]), "vieo_orb_get_level")
        return out

    # ---- batched device-resident form
    def extract_batch_device(self, d_images, n_images, width, height, stride, image_pitch,
                             d_keypoints, d_descriptors, capacity, d_counts, lapping=None):
        lap = None
        if lapping is not None:
            lap = (c_i * 2)(int(lapping[0]), int(lapping[1]))
        check(lib().vieo_orb_extract_batch_device(self._h, d_images, n_images, width, height,
                                                  stride, image_pitch, lap, d_keypoints,
                                                  d_descriptors, capacity, d_counts),
              "vieo_orb_extract_batch_device")

    def sync(self):
        check(lib().vieo_orb_sync(self._h))

    def enable_timing(self, on=True):
        check(lib().vieo_orb_enable_timing(self._h, int(on)))

    def last_stage_ms(self):
        ms = np.zeros(len(STAGES), np.float32)
        check(lib().vieo_orb_last_stage_ms(self._h, ms.ctypes.data), "last_stage_ms")
        return dict(zip(STAGES, ms.tolist()))

    def stage_ms_all(self):
        """per-stage milliseconds of every stamped batch call (oldest first)."""
        n = lib().vieo_orb_timed_steps(self._h)
        out = []
        for back in range(n - 1, -1, -1):
            ms = np.zeros(len(STAGES), np.float32)
            check(lib().vieo_orb_stage_ms(self._h, back, ms.ctypes.data), "stage_ms")
            out.append(dict(zip(STAGES, ms.tolist())))
        return out

    # ---- test taps
    def tap_blurred(self, level, image_index=0):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        check(lib().vieo_orb_tap_plane(self._h, image_index, level, 1, out.ctypes.data,
                                       out.strides[0]))
        return out

    def tap_candidates(self, level, image_index=0, cap=200000):
        out = np.zeros((cap, 3), np.int32)
        n = lib().vieo_orb_tap_candidates(self._h, image_index, level, out.ctypes.data, cap)
        if n < 0:
            check(n, "tap_candidates")
        return out[:n].copy()

    def tap_level_keys(self, level, image_index=0, cap=8192):
        out = np.zeros(cap, KEYPOINT_DTYPE)
        n = lib().vieo_orb_tap_level_keys(self._h, image_index, level, out.ctypes.data, cap)
        if n < 0:
            check(n, "tap_level_keys")
        return out[:n].copy()
