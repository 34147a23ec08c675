"""Synthetic inputs for Frame::ComputeStereoFishEyeMatches (a10): key points + descriptors of one frame of a
distorted multi-camera rig (synth_ba.camera_rig) with known 3-D structure, so that the matcher has true answers.

A scene point seen by several cameras gives each of them a key (projection + noise at a random pyramid level)
whose descriptor is the point's 256-bit signature with a few flipped bits; distractor keys, near-duplicate
signatures (to drive the contradiction / replacement branches of FillMatchesFromPair) and points behind the
parallax threshold are mixed in.
"""
import numpy as np

from . import synth_ba
from .ba_types import FISHEYE_PARAMS_DTYPE
from .orb_extractor import KEYPOINT_DTYPE


def _inv(T):
    R, t = T[:3, :3], T[:3, 3]
    o = np.eye(4)
    o[:3, :3], o[:3, 3] = R.T, -R.T @ t
    return o


def make_fisheye_case(seed, rig="kb8", n_points=500, n_levels=8, th_far_pts=0.0, distractors=0.3,
                      duplicates=0.1, far_frac=0.1, num_mono=None, flip_bits=10, noise=0.3, z_max=10.0):
    """returns dict(params FISHEYE_PARAMS_DTYPE[1], keys [n_cams], descs [n_cams], num_mono int32[n_cams],
    truth: per camera the scene-point id of every key (-1: distractor) and `X` the points in the reference
    camera frame, keep: arrays the params record points into)."""
    rng = np.random.default_rng(seed)
    cams, (W, H), Tcr = synth_ba.camera_rig(rig, with_tcr=True)
    nc = len(cams)
    # Sophus::SE3<float> members: Trc as given, Tcr = Trc.inverse() (both float), then cast to double
    Trc_f = [np.asarray(_inv(T), np.float32).astype(np.float64) for T in Tcr]
    Tcr_f = [np.asarray(_inv(T), np.float32).astype(np.float64) for T in Trc_f]
    Trc = np.ascontiguousarray(np.stack([T[:3, :] for T in Trc_f]).reshape(nc, 12))
    Tcr_a = np.ascontiguousarray(np.stack([T[:3, :] for T in Tcr_f]).reshape(nc, 12))
    sigma2 = (np.float32(1.2) ** np.arange(n_levels, dtype=np.float32)) ** 2
    sigma2 = np.ascontiguousarray(sigma2, np.float32)
    # scene in the reference camera frame
    X = np.stack([rng.uniform(-5, 5, n_points), rng.uniform(-3, 3, n_points), rng.uniform(1.2, z_max, n_points)], 1)
    far = rng.random(n_points) < far_frac
    X[far] *= rng.uniform(20, 200, far.sum())[:, None]  # (almost) no parallax
    sig = rng.integers(0, 256, (n_points, 32), dtype=np.uint8)
    ndup = int(duplicates * n_points)
    for a, b in zip(rng.integers(0, n_points, ndup), rng.integers(0, n_points, ndup)):
        sig[a] = sig[b]  # different points with (nearly) the same appearance
    keys, descs, owner = [], [], []
    for c in range(nc):
        kk, dd, oo = [], [], []
        Rc, tc = np.linalg.inv(Trc_f[c])[:3, :3], np.linalg.inv(Trc_f[c])[:3, 3]
        for p in range(n_points):
            Pc = Rc @ X[p] + tc
            if Pc[2] < 0.2:
                continue
            th = np.arctan2(np.hypot(Pc[0], Pc[1]), Pc[2])
            if th > (1.2 if cams[c]["model"] == 2 else 0.75):
                continue
            u, v = synth_ba.project_camera(cams[c], Pc)
            if not (8 <= u < W - 8 and 8 <= v < H - 8):
                continue
            lvl = int(rng.integers(0, min(4, n_levels)))
            s = 1.2 ** lvl
            d = sig[p].copy()
            for b in rng.integers(0, 256, rng.integers(0, flip_bits + 1)):
                d[b >> 3] ^= np.uint8(1 << (b & 7))
            kk.append((u + rng.normal(0, noise) * s, v + rng.normal(0, noise) * s, 31 * s, 0, 20, lvl, -1))
            dd.append(d)
            oo.append(p)
        for _ in range(int(distractors * max(len(kk), 10))):
            kk.append((rng.uniform(8, W - 8), rng.uniform(8, H - 8), 31, 0, 20, int(rng.integers(0, min(4, n_levels))),
                       -1))
            dd.append(rng.integers(0, 256, 32, dtype=np.uint8))
            oo.append(-1)
        order = rng.permutation(len(kk))
        keys.append(np.array([kk[i] for i in order], KEYPOINT_DTYPE))
        descs.append(np.ascontiguousarray(np.stack([dd[i] for i in order]).astype(np.uint8)))
        owner.append(np.array([oo[i] for i in order], np.int64))
    mono = np.zeros(nc, np.int32) if num_mono is None else np.asarray(num_mono, np.int32)
    params = np.zeros(1, FISHEYE_PARAMS_DTYPE)
    P = params[0]
    P["n_cams"], P["n_levels"], P["bf"], P["th_far_pts"] = nc, n_levels, 0.11 * float(cams[0]["fx"]), th_far_pts
    P["cams"], P["Trc"], P["Tcr"], P["level_sigma2"] = (cams.ctypes.data, Trc.ctypes.data, Tcr_a.ctypes.data,
                                                        sigma2.ctypes.data)
    return dict(params=params, keys=keys, descs=descs, num_mono=mono, owner=owner, X=X, far=far,
                keep=(cams, Trc, Tcr_a, sigma2), cams=cams, Tcr=Tcr_a)


def rig_extrinsics(Tcr):
    """(Trc [n, 12], Tcr [n, 12]) row-major 3x4 doubles as the camera members hold them: Sophus::SE3<float> Trc as
    given, Tcr = Trc.inverse() (both float), cast to double afterwards."""
    Trc_f = [np.asarray(_inv(T), np.float32).astype(np.float64) for T in Tcr]
    Tcr_f = [np.asarray(_inv(T), np.float32).astype(np.float64) for T in Trc_f]
    n = len(Tcr)
    return (np.ascontiguousarray(np.stack([T[:3, :] for T in Trc_f]).reshape(n, 12)),
            np.ascontiguousarray(np.stack([T[:3, :] for T in Tcr_f]).reshape(n, 12)))


def make_sbp_rig(cams, Tcr, bounds, use_distort=True):
    """SBP_RIG_DTYPE[1] from a CAMERA_DTYPE array, the 4x4 Tcr list and per-camera bounds [n_cams][4].
    Tcr / Trc go through Sophus::SE3<float> like the camera members (cast to double afterwards)."""
    from .ba_types import SBP_RIG_DTYPE
    rig = np.zeros(1, SBP_RIG_DTYPE)
    R = rig[0]
    nc = len(cams)
    R["n_cams"], R["use_distort"] = nc, int(use_distort)
    R["cams"][:nc] = cams
    Trc_a, Tcr_a = rig_extrinsics(Tcr)
    for c in range(nc):
        R["Tcr"][c] = Tcr_a[c]
        R["trc"][c] = Trc_a[c].reshape(3, 4)[:, 3]
        R["bounds"][c] = bounds[c]
    return rig


def make_rig_tracking_case(seed, rig="kb8", n_cams=None, n_points=700, motion=(0.03, -0.01, 0.05), rot=(0.004, -0.006, 0.003),
                           n_levels=8, th=7.0, distractors=0.4, noise=0.6, flip_bits=24, observed=0.85, angle_noise=4.0):
    """One rig frame for the tracking-side projection searches (a12-a14 with the camera loop): world points with a
    signature, the current frame's keys in every camera (projection under the true current pose + noise, at the
    point's pyramid level +-1; distractors), the last frame's map points (LAST_FRAME_POINT_DTYPE) and the
    key-frame form (KEYFRAME_POINT_DTYPE), the poses (SBP_CAMERA_DTYPE) and the rig (SBP_RIG_DTYPE).
    Keys are concatenated camera-major like Frame::mvKeys (Frame.cc:738-764); vuright_ = -1 (:759)."""
    from . import frontend
    from .ba_types import KEYFRAME_POINT_DTYPE, LAST_FRAME_POINT_DTYPE
    rng = np.random.default_rng(seed)
    cams, (W, H), Tcr = synth_ba.camera_rig(rig, with_tcr=True, n_cams=n_cams)
    nc = len(cams)
    bounds = np.tile(np.array([0, W, 0, H], np.float32), (nc, 1))
    R = make_sbp_rig(cams, Tcr, bounds, True)
    scale = (np.float32(1.2) ** np.arange(n_levels, dtype=np.float32)).astype(np.float32)
    # world = the last frame's reference camera; the current reference camera moved a little
    Rcw = synth_ba.so3_exp(np.array(rot, float))
    tcw = -Rcw @ np.array(motion, float)
    Tc = np.hstack([Rcw, tcw[:, None]])
    Tl = np.hstack([np.eye(3), np.zeros((3, 1))])
    X = np.stack([rng.uniform(-6, 6, n_points), rng.uniform(-3.5, 3.5, n_points), rng.uniform(1.0, 12.0, n_points)], 1)
    X = X.astype(np.float32)
    sig = rng.integers(0, 256, (n_points, 32), dtype=np.uint8)
    lvl = rng.integers(0, n_levels - 1, n_points)
    ang = rng.uniform(0, 360, n_points).astype(np.float32)
    pts = np.zeros(n_points, LAST_FRAME_POINT_DTYPE)
    pts["Xw"], pts["octave"], pts["angle"], pts["desc"] = X, lvl, ang, sig
    valid = rng.random(n_points) < 0.92
    obs = rng.random(n_points) < observed
    pts["flags"] = valid.astype(np.int32) * (1 + 2 * obs.astype(np.int32))
    kfp = np.zeros(n_points, KEYFRAME_POINT_DTYPE)
    for f in ("Xw", "octave", "angle", "flags", "desc"):
        kfp[f] = pts[f]
    d0 = np.linalg.norm(X.astype(np.float64), axis=1)  # distance at which the point sits at level `lvl`
    kfp["max_distance"] = (d0 * 1.2 ** lvl).astype(np.float32)
    kfp["min_distance"] = (kfp["max_distance"] / np.float32(1.2 ** (n_levels - 1))).astype(np.float32)
    rot_common = rng.uniform(-3, 3)
    keys, descs = [], []
    for c in range(nc):
        Tcr_d = R[0]["Tcr"][c].reshape(3, 4)
        kk, dd = [], []
        for p in range(n_points):
            Pc = Tcr_d[:, :3] @ (Rcw @ X[p].astype(np.float64) + tcw) + Tcr_d[:, 3]
            if Pc[2] < 0.2 or rng.random() < 0.15:
                continue
            if np.arctan2(np.hypot(Pc[0], Pc[1]), Pc[2]) > (1.2 if cams[c]["model"] == 2 else 0.75):
                continue
            u, v = synth_ba.project_camera(cams[c], Pc)
            if not (4 <= u < W - 4 and 4 <= v < H - 4):
                continue
            l2 = int(np.clip(lvl[p] + rng.integers(-1, 2), 0, n_levels - 1))
            s = 1.2 ** l2
            d = sig[p].copy()
            for b in rng.integers(0, 256, rng.integers(0, flip_bits + 1)):
                d[b >> 3] ^= np.uint8(1 << (b & 7))
            a = (ang[p] - rot_common + rng.normal(0, angle_noise) + (rng.uniform(0, 360) if rng.random() < 0.06 else 0)) % 360
            kk.append((u + rng.normal(0, noise) * s, v + rng.normal(0, noise) * s, 31 * s, a, 20 + rng.integers(0, 60),
                       l2, -1))
            dd.append(d)
        for _ in range(int(distractors * max(len(kk), 10))):
            kk.append((rng.uniform(4, W - 4), rng.uniform(4, H - 4), 31, rng.uniform(0, 360), 20,
                       int(rng.integers(0, n_levels)), -1))
            dd.append(rng.integers(0, 256, 32, dtype=np.uint8))
        order = rng.permutation(len(kk))
        keys.append(np.array([kk[i] for i in order], KEYPOINT_DTYPE))
        descs.append(np.ascontiguousarray(np.stack([dd[i] for i in order]).astype(np.uint8)))
    cam_first = np.concatenate([[0], np.cumsum([len(k) for k in keys])]).astype(np.int32)
    allkeys = np.concatenate(keys)
    alldesc = np.ascontiguousarray(np.concatenate(descs))
    uright = np.full(len(allkeys), -1.0, np.float32)
    K0 = (float(cams[0]["fx"]), float(cams[0]["fy"]), float(cams[0]["cx"]), float(cams[0]["cy"]))
    bf = 0.11 * K0[0]
    cam = frontend.make_sbp_camera(Tc, Tl, K0, bounds[0], bf, bf / K0[0], th, scale)
    return dict(rig=R, cam=cam, pts=pts, kf_pts=kfp, keys=allkeys, desc=alldesc, uright=uright, cam_first=cam_first,
                bounds=bounds, scale=scale, cams=cams, cam_keys=keys, cam_descs=descs, X=X, Tcw=Tc,
                log_scale_factor=float(np.log(np.float32(1.2))))
