"""Host-side mirrors of the map-point steps next to the hot path, batched over points:
  is_in_frustum(frame, points)                Frame::isInFrustum                       (Frame.cc:335-416)
  compute_distinctive_descriptors(desc, first) MapPoint::ComputeDistinctiveDescriptors  (MapPoint.cc:314-378)
  update_normal_and_depth(...)                MapPoint::UpdateNormalAndDepth           (MapPoint.cc:424-480)
"""
import numpy as np

from ._lib import check, lib

FRUSTUM_FRAME_DTYPE = np.dtype([("Rcrw", "<f4", 9), ("tcrw", "<f4", 3), ("Ow", "<f4", 3), ("n_cams", "<i4"),
                                ("use_distort", "<i4"), ("cams", "<u8"), ("Tcr", "<f4", (4, 12)),
                                ("trc", "<f4", (4, 3)), ("bounds", "<f4", (4, 4)), ("bf", "<f4"),
                                ("log_scale_factor", "<f4"), ("n_levels", "<i4"), ("viewing_cos_limit", "<f4")],
                               align=True)
FRUSTUM_POINT_DTYPE = np.dtype([("Xw", "<f4", 3), ("normal", "<f4", 3), ("max_distance", "<f4"),
                                ("min_distance", "<f4")])
TRACK_INFO_DTYPE = np.dtype([("u", "<f4", 4), ("v", "<f4", 4), ("ur", "<f4", 4), ("viewcos", "<f4", 4),
                             ("level", "<i4", 4), ("cam", "<i4", 4), ("n", "<i4"), ("track_depth", "<f4")])
assert FRUSTUM_POINT_DTYPE.itemsize == 32 and TRACK_INFO_DTYPE.itemsize == 104
assert FRUSTUM_FRAME_DTYPE.itemsize == 400


def frustum_call(fn, frame, points):
    fr = np.ascontiguousarray(frame, FRUSTUM_FRAME_DTYPE).reshape(1)
    pts = np.ascontiguousarray(points, FRUSTUM_POINT_DTYPE)
    info = np.zeros(max(len(pts), 1), TRACK_INFO_DTYPE)
    rc = fn(fr.ctypes.data, pts.ctypes.data, len(pts), info.ctypes.data)
    return rc, info[:len(pts)]


def is_in_frustum(frame, points):
    """returns TRACK_INFO_DTYPE[n]: the _TrackFastMatchInfo of every point (n > 0 <=> in view)."""
    rc, info = frustum_call(lib().vieo_is_in_frustum_batch, frame, points)
    check(rc, "vieo_is_in_frustum_batch")
    return info


def distinctive_call(fn, descriptors, first):
    d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
    f = np.ascontiguousarray(first, np.int32)
    best = np.zeros(max(len(f) - 1, 1), np.int32)
    rc = fn(d.ctypes.data, f.ctypes.data, len(f) - 1, best.ctypes.data)
    return rc, best[:len(f) - 1]


def compute_distinctive_descriptors(descriptors, first):
    """descriptors uint8[total, 32], first int32[n + 1] (CSR): returns int32[n], the chosen row of every point."""
    rc, best = distinctive_call(lib().vieo_distinctive_descriptors_batch, descriptors, first)
    check(rc, "vieo_distinctive_descriptors_batch")
    return best


def normal_depth_call(fn, points, first, obs_centre, centres, ref_centre, ref_scale, scale_last_level, oracle=False):
    P = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
    f, oc = np.ascontiguousarray(first, np.int32), np.ascontiguousarray(obs_centre, np.int32)
    C = np.ascontiguousarray(centres, np.float32).reshape(-1, 3)
    rc_, rs = np.ascontiguousarray(ref_centre, np.int32), np.ascontiguousarray(ref_scale, np.float32)
    n = len(P)
    nrm, mx, mn = np.zeros((max(n, 1), 3), np.float32), np.zeros(max(n, 1), np.float32), np.zeros(max(n, 1), np.float32)
    import ctypes
    if oracle:
        rc = fn(P.ctypes.data, f.ctypes.data, oc.ctypes.data, C.ctypes.data, rc_.ctypes.data, rs.ctypes.data,
                ctypes.c_float(scale_last_level), n, nrm.ctypes.data, mx.ctypes.data, mn.ctypes.data)
    else:
        rc = fn(P.ctypes.data, f.ctypes.data, oc.ctypes.data, C.ctypes.data, len(C), rc_.ctypes.data,
                rs.ctypes.data, ctypes.c_float(scale_last_level), n, nrm.ctypes.data, mx.ctypes.data, mn.ctypes.data)
    return rc, nrm[:n], mx[:n], mn[:n]


def update_normal_and_depth(points, first, obs_centre, centres, ref_centre, ref_scale, scale_last_level):
    """returns (mNormalVector float32[n, 3], mfMaxDistance[n], mfMinDistance[n])."""
    rc, nrm, mx, mn = normal_depth_call(lib().vieo_update_normal_and_depth_batch, points, first, obs_centre,
                                        centres, ref_centre, ref_scale, scale_last_level)
    check(rc, "vieo_update_normal_and_depth_batch")
    return nrm, mx, mn


FUSE_FRAME_DTYPE = np.dtype([("base", FRUSTUM_FRAME_DTYPE), ("scale_factors", "<f4", 16),
                             ("inv_level_sigma2", "<f4", 16), ("th_radius", "<f4"), ("check_viewing_angle", "<i4"),
                             ("use_bf", "<i4"), ("reserved", "<i4")], align=True)
FUSE_POINT_DTYPE = np.dtype([("Xw", "<f4", 3), ("normal", "<f4", 3), ("max_distance", "<f4"),
                             ("min_distance", "<f4"), ("desc", "u1", 32), ("skip_mask", "<i4"), ("reserved", "<i4")])
assert FUSE_FRAME_DTYPE.itemsize == 544 and FUSE_POINT_DTYPE.itemsize == 72


def fuse_call(fn, frame, keys, uright, descs, points):
    """One SearchByProjectionBase search for `fn` (the C-ABI entry or the oracle's function of the same
    signature).  keys[c]: KEYPOINT_DTYPE[n_c], uright[c]: float32[n_c] or None, descs[c]: uint8[n_c, 32].
    returns (rc, best_idx int32[n, n_cams], best_dist int32[n, n_cams])."""
    import ctypes
    fr = np.ascontiguousarray(frame, FUSE_FRAME_DTYPE).reshape(1)
    nc = int(fr[0]["base"]["n_cams"])
    keys = [np.ascontiguousarray(k) for k in keys]
    descs = [np.ascontiguousarray(d, np.uint8).reshape(-1, 32) for d in descs]
    urs = [None if u is None else np.ascontiguousarray(u, np.float32) for u in uright]
    n_keys = np.array([len(k) for k in keys], np.int32)
    kp = (ctypes.c_void_p * nc)(*[k.ctypes.data for k in keys])
    dp = (ctypes.c_void_p * nc)(*[d.ctypes.data for d in descs])
    up = (ctypes.c_void_p * nc)(*[None if u is None else u.ctypes.data for u in urs])
    pts = np.ascontiguousarray(points, FUSE_POINT_DTYPE)
    bi = np.zeros((max(len(pts), 1), nc), np.int32)
    bd = np.zeros((max(len(pts), 1), nc), np.int32)
    rc = fn(fr.ctypes.data, ctypes.cast(kp, ctypes.c_void_p), ctypes.cast(up, ctypes.c_void_p),
            ctypes.cast(dp, ctypes.c_void_p), n_keys.ctypes.data, pts.ctypes.data, len(pts), bi.ctypes.data,
            bd.ctypes.data)
    return rc, bi[:len(pts)], bd[:len(pts)]


def fuse_search(frame, keys, uright, descs, points):
    """The search of ORBmatcher::SearchByProjectionBase / Fuse (ORBmatcher.cc:26-193): per (point, camera) the
    best key (index into that camera's keys, -1: none) and its Hamming distance."""
    rc, bi, bd = fuse_call(lib().vieo_fuse_search, frame, keys, uright, descs, points)
    check(rc, "vieo_fuse_search")
    return bi, bd
