"""Host-side mirror of IMUPreIntegratorBase::PreIntegration (reference src/Odom/OdomPreIntegrator.h:226-506) for a
batch of intervals on the C-ABI."""
import numpy as np

from ._lib import check, lib
from .ba_types import IMU_PREINT_DTYPE

IMU_SAMPLE_DTYPE = np.dtype([("t", "<f8"), ("w", "<f8", 3), ("a", "<f8", 3)])
IMU_NOISE_DTYPE = np.dtype([("sigma_g", "<f8", 9), ("sigma_a", "<f8", 9), ("freq_ref", "<f8"),
                            ("dt_cov_noise_fixed", "<i4"), ("reserved", "<i4")])
assert IMU_SAMPLE_DTYPE.itemsize == 56 and IMU_NOISE_DTYPE.itemsize == 160
PREINT_OK, PREINT_EMPTY, PREINT_GAP, PREINT_UNSUPPORTED = 0, 1, 2, 3


def preint_call(fn, noise, sample_lists, ti, tj, bg, ba):
    """sample_lists: one IMU_SAMPLE_DTYPE array per interval.  returns (rc, IMU_PREINT_DTYPE[n],
    sigma_prv float64[n, 9, 9], status int32[n])."""
    n = len(sample_lists)
    nz = np.ascontiguousarray(noise, IMU_NOISE_DTYPE).reshape(1)
    first = np.concatenate([[0], np.cumsum([len(s) for s in sample_lists])]).astype(np.int32)
    flat = (np.concatenate([np.ascontiguousarray(s, IMU_SAMPLE_DTYPE) for s in sample_lists])
            if first[-1] else np.zeros(1, IMU_SAMPLE_DTYPE))
    ti, tj = np.ascontiguousarray(ti, np.float64), np.ascontiguousarray(tj, np.float64)
    bg, ba = np.ascontiguousarray(bg, np.float64).reshape(-1, 3), np.ascontiguousarray(ba, np.float64).reshape(-1, 3)
    out = np.zeros(max(n, 1), IMU_PREINT_DTYPE)
    prv = np.zeros((max(n, 1), 9, 9), np.float64)
    st = np.zeros(max(n, 1), np.int32)
    rc = fn(nz.ctypes.data, flat.ctypes.data, first.ctypes.data, ti.ctypes.data, tj.ctypes.data, bg.ctypes.data,
            ba.ctypes.data, n, out.ctypes.data, prv.ctypes.data, st.ctypes.data)
    return rc, out[:n], prv[:n], st[:n]


def imu_preintegrate(noise, sample_lists, ti, tj, bg, ba):
    rc, out, prv, st = preint_call(lib().vieo_imu_preintegrate_batch, noise, sample_lists, ti, tj, bg, ba)
    check(rc, "vieo_imu_preintegrate_batch")
    return out, prv, st
