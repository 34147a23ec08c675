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


def queries_from_track_info(info, desc, th, scale, observed=None, th_far=0.0):
    """The head of SearchByProjection(Frame&, vector<MapPoint*>&, th, th_far) (ORBmatcher.cc:237-266): one window
    query per (map point in view, camera of vtrack_cami_), point-major.  info: TRACK_INFO_DTYPE[n] from
    Frame::isInFrustum, desc uint8[n, 32], scale = scalepyrinfo_.vscalefactor_ (float32).
    returns (queries PROJ_QUERY_DTYPE[m], point index of every query int32[m])."""
    from .ba_types import PROJ_QUERY_DTYPE
    info = np.asarray(info)
    n = len(info)
    kmax = info["level"].shape[1] if n else 0
    cnt = np.clip(info["n"].astype(np.int64), 0, kmax)
    if th_far > 0:
        cnt = np.where(info["track_depth"] > th_far, 0, cnt)
    # (point, slot) pairs, point-major
    sel = np.arange(kmax)[None, :] < cnt[:, None]
    owner, slot = np.nonzero(sel)
    lvl = info["level"][owner, slot].astype(np.int32)
    r = np.where(info["viewcos"][owner, slot] > np.float32(0.998), np.float32(2.5), np.float32(4.0)).astype(np.float32)
    if th != 1.0:  # RadiusByViewingCos, then the factor, then the level's scale: three float32 products
        r = (r * np.float32(th)).astype(np.float32)
    q = np.zeros(len(owner), PROJ_QUERY_DTYPE)
    q["u"], q["v"], q["ur"] = info["u"][owner, slot], info["v"][owner, slot], info["ur"][owner, slot]
    q["radius"] = (r * np.asarray(scale, np.float32)[lvl]).astype(np.float32)
    q["level_min"], q["level_max"], q["angle"] = lvl - 1, lvl, 0.0
    obs = np.ones(len(owner), bool) if observed is None else np.asarray(observed, bool)[owner]
    q["flags"] = 1 | np.where(obs, 2, 0) | (info["cam"][owner, slot].astype(np.int32) << 8)
    q["desc"] = np.asarray(desc)[owner]
    return q, owner.astype(np.int32)
