"""numpy dtypes that mirror the POD structs of include/vieo_hot.h (pose optimisation / BA)."""
import numpy as np

NAVSTATE_DTYPE = np.dtype([("p", "<f8", 3), ("q", "<f8", 4), ("v", "<f8", 3), ("bg", "<f8", 3),
                           ("ba", "<f8", 3), ("dbg", "<f8", 3), ("dba", "<f8", 3)], align=True)
POSE_OBS_DTYPE = np.dtype([("Xw", "<f4", 3), ("u", "<f4"), ("v", "<f4"), ("ur", "<f4"),
                           ("inv_sigma2", "<f4"), ("flags", "<i4")], align=True)
POSE_FRAME_DTYPE = np.dtype([("nav", NAVSTATE_DTYPE), ("Rcb", "<f8", 9), ("tcb", "<f8", 3),
                             ("fx", "<f4"), ("fy", "<f4"), ("cx", "<f4"), ("cy", "<f4"),
                             ("bf", "<f4"), ("obs_begin", "<i4"), ("n_obs", "<i4"),
                             ("reserved", "<i4")], align=True)
POSE_RESULT_DTYPE = np.dtype([("nav", NAVSTATE_DTYPE), ("n_inliers", "<i4"), ("status", "<i4"),
                              ("lm_iterations", "<i4"), ("reserved", "<i4")], align=True)
assert NAVSTATE_DTYPE.itemsize == 176 and POSE_OBS_DTYPE.itemsize == 32
assert POSE_FRAME_DTYPE.itemsize == 304 and POSE_RESULT_DTYPE.itemsize == 192
