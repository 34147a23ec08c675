"""numpy dtypes that mirror the POD structs of include/vieo_hot.h (pose optimisation / BA)."""
import numpy as np

NAVSTATE_DTYPE = np.dtype([("p", "<f8", 3), ("q", "<f8", 4), ("v", "<f8", 3), ("bg", "<f8", 3),
                           ("ba", "<f8", 3), ("dbg", "<f8", 3), ("dba", "<f8", 3)], align=True)
POSE_OBS_DTYPE = np.dtype([("Xw", "<f4", 3), ("u", "<f4"), ("v", "<f4"), ("ur", "<f4"),
                           ("inv_sigma2", "<f4"), ("flags", "<i4")], align=True)
POSE_FRAME_DTYPE = np.dtype([("nav", NAVSTATE_DTYPE), ("Rcb", "<f8", 9), ("tcb", "<f8", 3),
                             ("fx", "<f4"), ("fy", "<f4"), ("cx", "<f4"), ("cy", "<f4"),
                             ("bf", "<f4"), ("obs_begin", "<i4"), ("n_obs", "<i4"),
                             ("n_cams", "<i4"), ("cams", "<u8"), ("enc", "<u8")],
                            align=True)  # cams / enc: pointers, see vieo_hot.h
ENC_PREINT_DTYPE = np.dtype([("dt", "<f8"), ("delx", "<f8", 6), ("Sigma", "<f8", 36)], align=True)
POSE_ENC_DTYPE = np.dtype([("enc", ENC_PREINT_DTYPE), ("qRbe", "<f8", 4), ("pbe", "<f8", 3), ("p_last", "<f8", 3),
                           ("q_last", "<f8", 4)], align=True)
POSE_RESULT_DTYPE = np.dtype([("nav", NAVSTATE_DTYPE), ("n_inliers", "<i4"), ("status", "<i4"),
                              ("lm_iterations", "<i4"), ("reserved", "<i4")], align=True)
assert NAVSTATE_DTYPE.itemsize == 176 and POSE_OBS_DTYPE.itemsize == 32
assert POSE_FRAME_DTYPE.itemsize == 320 and POSE_RESULT_DTYPE.itemsize == 192
assert ENC_PREINT_DTYPE.itemsize == 344 and POSE_ENC_DTYPE.itemsize == 456

PROJ_QUERY_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("ur", "<f4"), ("radius", "<f4"),
                             ("level_min", "<i4"), ("level_max", "<i4"), ("angle", "<f4"),
                             ("flags", "<i4"), ("desc", "u1", 32)], align=True)
LAST_FRAME_POINT_DTYPE = np.dtype([("Xw", "<f4", 3), ("octave", "<i4"), ("angle", "<f4"),
                                   ("flags", "<i4"), ("reserved", "<i4", 2), ("desc", "u1", 32)],
                                  align=True)
SBP_CAMERA_DTYPE = np.dtype([("Tcw_cur", "<f8", 12), ("Tcw_last", "<f8", 12), ("fx", "<f4"),
                             ("fy", "<f4"), ("cx", "<f4"), ("cy", "<f4"), ("bounds", "<f4", 4),
                             ("bf", "<f4"), ("baseline", "<f4"), ("th", "<f4"), ("th_far", "<f4"),
                             ("mono", "<i4"), ("nlevels", "<i4"), ("scale", "<f4", 16)], align=True)
assert PROJ_QUERY_DTYPE.itemsize == 64 and LAST_FRAME_POINT_DTYPE.itemsize == 64
assert SBP_CAMERA_DTYPE.itemsize == 312
KEYFRAME_POINT_DTYPE = np.dtype([("Xw", "<f4", 3), ("octave", "<i4"), ("angle", "<f4"), ("flags", "<i4"),
                                 ("max_distance", "<f4"), ("min_distance", "<f4"), ("desc", "u1", 32)], align=True)
assert KEYFRAME_POINT_DTYPE.itemsize == 64

IMU_PREINT_DTYPE = np.dtype([("dt", "<f8"), ("Rij", "<f8", 9), ("vij", "<f8", 3), ("pij", "<f8", 3),
                             ("JgR", "<f8", 9), ("Jgv", "<f8", 9), ("Jav", "<f8", 9),
                             ("Jgp", "<f8", 9), ("Jap", "<f8", 9), ("Sigma", "<f8", 81)], align=True)
VIO_FRAME_DTYPE = np.dtype([("base", POSE_FRAME_DTYPE), ("nav_last", NAVSTATE_DTYPE),
                            ("nav_prior", NAVSTATE_DTYPE), ("H_prior", "<f8", 225),
                            ("imu", IMU_PREINT_DTYPE), ("gw", "<f8", 3), ("inv_sigma_bg2", "<f8"),
                            ("inv_sigma_ba2", "<f8"), ("dt_frames", "<f8"), ("th_depth", "<f4"),
                            ("last_has_prior", "<i4"), ("compute_marg", "<i4"), ("no_mps", "<i4")],
                           align=True)
VIO_RESULT_DTYPE = np.dtype([("base", POSE_RESULT_DTYPE), ("H_marg", "<f8", 225),
                             ("has_marg", "<i4"), ("reserved", "<i4")], align=True)

LBA_KEYFRAME_DTYPE = np.dtype([("nav", NAVSTATE_DTYPE), ("fixed", "<i4"), ("reserved", "<i4")], align=True)
LBA_OBS_DTYPE = np.dtype([("kf", "<i4"), ("mp", "<i4"), ("u", "<f4"), ("v", "<f4"), ("ur", "<f4"),
                          ("inv_sigma2", "<f4")], align=True)
CAMERA_DTYPE = np.dtype([("model", "<i4"), ("num_k", "<i4"), ("fx", "<f4"), ("fy", "<f4"), ("cx", "<f4"),
                         ("cy", "<f4"), ("dist", "<f4", 8), ("Rcb", "<f8", 9), ("tcb", "<f8", 3)], align=True)
assert CAMERA_DTYPE.itemsize == 152
# vieo_sbp_rig: the cameras of a rig frame as the tracking-side projection searches use them
SBP_RIG_DTYPE = np.dtype([("n_cams", "<i4"), ("use_distort", "<i4"), ("cams", CAMERA_DTYPE, 4),
                          ("Tcr", "<f8", (4, 12)), ("trc", "<f8", (4, 3)), ("bounds", "<f4", (4, 4))], align=True)
assert SBP_RIG_DTYPE.itemsize == 1160
# "cams" is a host pointer (array.ctypes.data of a CAMERA_DTYPE array the caller keeps alive)
LBA_PARAMS_DTYPE = np.dtype([("Rcb", "<f8", 9), ("tcb", "<f8", 3), ("fx", "<f4"), ("fy", "<f4"),
                             ("cx", "<f4"), ("cy", "<f4"), ("bf", "<f4"), ("its0", "<i4"),
                             ("its1", "<i4"), ("n_cams", "<i4"), ("cams", "<u8")], align=True)
assert LBA_PARAMS_DTYPE.itemsize == 136
LBA_RESULT_DTYPE = np.dtype([("status", "<i4"), ("n_erase", "<i4"), ("lm_iterations", "<i4"),
                             ("lm_trials", "<i4"), ("chi2_initial", "<f8"), ("chi2_final", "<f8")],
                            align=True)

LBA_IMU_EDGE_DTYPE = np.dtype([("kf_i", "<i4"), ("kf_j", "<i4"), ("dt_kf", "<f8"), ("imu", IMU_PREINT_DTYPE),
                               ("enc", ENC_PREINT_DTYPE)], align=True)
LBA_VIO_PARAMS_DTYPE = np.dtype([("base", LBA_PARAMS_DTYPE), ("gw", "<f8", 3), ("inv_sigma_bg2", "<f8"),
                                 ("inv_sigma_ba2", "<f8"), ("lambda_init", "<f8"), ("rec_init", "<i4"),
                                 ("large", "<i4"), ("th_dist_far", "<f4"), ("reserved", "<i4"), ("qRbe", "<f8", 4),
                                 ("pbe", "<f8", 3)], align=True)
assert LBA_IMU_EDGE_DTYPE.itemsize == 1496 and LBA_VIO_PARAMS_DTYPE.itemsize == 256
# encoder edges of the vision-only BAs (vieo_lba_enc_edge / vieo_lba_enc)
LBA_ENC_EDGE_DTYPE = np.dtype([("kf_i", "<i4"), ("kf_j", "<i4"), ("enc", ENC_PREINT_DTYPE)], align=True)
LBA_ENC_DTYPE = np.dtype([("n_edges", "<i4"), ("reserved", "<i4"), ("edges", "<u8"), ("qRbe", "<f8", 4),
                          ("pbe", "<f8", 3)], align=True)
assert LBA_ENC_EDGE_DTYPE.itemsize == 352 and LBA_ENC_DTYPE.itemsize == 72

# vieo_fisheye_params: cams / Trc / Tcr / level_sigma2 are host pointers the caller keeps alive
FISHEYE_PARAMS_DTYPE = np.dtype([("n_cams", "<i4"), ("n_levels", "<i4"), ("bf", "<f4"), ("th_far_pts", "<f4"),
                                 ("cams", "<u8"), ("Trc", "<u8"), ("Tcr", "<u8"), ("level_sigma2", "<u8")],
                                align=True)
assert FISHEYE_PARAMS_DTYPE.itemsize == 48
