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
