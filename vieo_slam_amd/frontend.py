"""Host glue between the hot-path stages, shared by tests and bench (the part Tracking.cc does
around the kernels): stereo unprojection into map points, query / observation assembly."""
import numpy as np

from .ba_types import LAST_FRAME_POINT_DTYPE, POSE_OBS_DTYPE, SBP_CAMERA_DTYPE


def pose_to_Tcw(Rwc, twc):
    Rcw = Rwc.T
    return np.hstack([Rcw, (-Rcw @ twc)[:, None]])


def unproject_stereo(keys, depth, K, Rwc, twc):
    """Frame::UnprojectStereo for keys with depth > 0 -> float32 world points (n,3), mask."""
    fx, fy, cx, cy = K
    ok = depth > 0
    z = np.where(ok, depth, 1.0).astype(np.float64)
    Xc = np.stack([(keys["x"] - cx) * z / fx, (keys["y"] - cy) * z / fy, z], 1)
    Xw = Xc @ Rwc.T + twc
    return Xw.astype(np.float32), ok


def make_last_frame_points(keys, desc, Xw, valid, observed=True):
    pts = np.zeros(len(keys), LAST_FRAME_POINT_DTYPE)
    pts["Xw"] = Xw
    pts["octave"] = keys["octave"]
    pts["angle"] = keys["angle"]
    pts["flags"] = valid.astype(np.int32) * (3 if observed else 1)
    pts["desc"] = desc
    return pts


def make_sbp_camera(Tcw_cur, Tcw_last, K, bounds, bf, baseline, th, scale, th_far=0.0, mono=False):
    cam = np.zeros(1, SBP_CAMERA_DTYPE)
    c = cam[0]
    c["Tcw_cur"] = np.asarray(Tcw_cur, np.float64).reshape(-1)
    c["Tcw_last"] = np.asarray(Tcw_last, np.float64).reshape(-1)
    c["fx"], c["fy"], c["cx"], c["cy"] = K
    c["bounds"] = bounds
    c["bf"], c["baseline"], c["th"], c["th_far"] = bf, baseline, th, th_far
    c["mono"], c["nlevels"] = int(mono), len(scale)
    c["scale"][:len(scale)] = scale
    return cam


def build_pose_obs(assign, query_Xw, keys, uright, inv_sigma2):
    """What PoseOptimization reads from the frame after a search: one observation per keypoint
    that holds a map point (Optimizer.cc:1704-1786).  returns (obs, key_index)."""
    idx = np.nonzero(assign >= 0)[0]
    obs = np.zeros(len(idx), POSE_OBS_DTYPE)
    obs["Xw"] = query_Xw[assign[idx]]
    obs["u"], obs["v"] = keys["x"][idx], keys["y"][idx]
    obs["ur"] = uright[idx]
    obs["inv_sigma2"] = inv_sigma2[keys["octave"][idx]]
    return obs, idx
