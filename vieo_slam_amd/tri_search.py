"""Host-side mirror of ORBmatcher::SearchForTriangulation (reference src/ORBmatcher.cc:896-1150) over the C-ABI
(`vieo_search_for_triangulation`), plus the seeded key-frame pairs its tests run on."""
import ctypes

import numpy as np

from ._lib import check, lib
from .orb_extractor import KEYPOINT_DTYPE

TRI_KEYFRAME_DTYPE = np.dtype([("Tcw", "<f8", 12), ("fx", "<f4"), ("fy", "<f4"), ("cx", "<f4"), ("cy", "<f4"),
                               ("n_keys", "<i4"), ("n_nodes", "<i4"), ("keys", "<u8"), ("descriptors", "<u8"),
                               ("uright", "<u8"), ("has_mappoint", "<u8"), ("node_id", "<u8"), ("node_first", "<u8"),
                               ("node_feat", "<u8"), ("scale_factor", "<u8"), ("level_sigma2", "<u8"),
                               ("n_levels", "<i4"), ("n_cams", "<i4"), ("cams", "<u8"), ("Tcr", "<u8"), ("Trc", "<u8"),
                               ("key_cam", "<u8")], align=True)
assert TRI_KEYFRAME_DTYPE.itemsize == 232


class TriKeyFrame:
    """The arrays one key frame contributes (kept alive here) and its vieo_tri_keyframe record.
    feat_vec: list of (node id, [feature indices]) in ascending node order = DBoW2::FeatureVector.
    rig = (cams CAMERA_DTYPE[n], [Tcr 4x4 per camera], key_cam uint8[n_keys]) for a distorted rig (keys = mvKeys)."""

    def __init__(self, Tcw, K, keys, descriptors, uright, has_mappoint, feat_vec, scale_factor, level_sigma2, rig=None):
        self.keys = np.ascontiguousarray(keys, KEYPOINT_DTYPE)
        self.desc = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        self.uright = np.ascontiguousarray(uright, np.float32)
        self.has_mp = np.ascontiguousarray(has_mappoint, np.uint8)
        self.node_id = np.array([n for n, _ in feat_vec], np.uint32)
        self.node_first = np.zeros(len(feat_vec) + 1, np.int32)
        self.node_first[1:] = np.cumsum([len(f) for _, f in feat_vec])
        self.node_feat = np.array([i for _, f in feat_vec for i in f], np.int32)
        self.scale = np.ascontiguousarray(scale_factor, np.float32)
        self.sigma2 = np.ascontiguousarray(level_sigma2, np.float32)
        r = np.zeros(1, TRI_KEYFRAME_DTYPE)
        r[0]["Tcw"] = np.asarray(Tcw, float)[:3, :4].reshape(-1)
        r[0]["fx"], r[0]["fy"], r[0]["cx"], r[0]["cy"] = K
        r[0]["n_keys"], r[0]["n_nodes"], r[0]["n_levels"] = len(self.keys), len(self.node_id), len(self.scale)
        if rig is not None:
            cams, Tcr, key_cam = rig
            self.cams = np.ascontiguousarray(cams)
            # SE3<float> in the reference: the values handed over are the float-rounded ones
            self.Tcr = np.array([np.asarray(T, float)[:3, :4] for T in Tcr]).astype(np.float32).astype(np.float64)
            self.Trc = np.array([np.linalg.inv(np.asarray(T, float))[:3, :4] for T in Tcr]).astype(np.float32).astype(np.float64)
            self.key_cam = np.ascontiguousarray(key_cam, np.uint8)
            r[0]["n_cams"] = len(self.cams)
            for name, arr in (("cams", self.cams), ("Tcr", self.Tcr), ("Trc", self.Trc), ("key_cam", self.key_cam)):
                r[0][name] = arr.ctypes.data
        self.n_cams = 1 if rig is None else len(rig[0])
        for name, arr in (("keys", self.keys), ("descriptors", self.desc), ("uright", self.uright),
                          ("has_mappoint", self.has_mp), ("node_id", self.node_id), ("node_first", self.node_first),
                          ("node_feat", self.node_feat), ("scale_factor", self.scale), ("level_sigma2", self.sigma2)):
            r[0][name] = arr.ctypes.data
        self.rec = r


def tri_call(fn, kf1, kf2s, only_stereo=False, check_orientation=True, pair_capacity=None):
    """Marshals one batch for `fn` (the C-ABI entry or the oracle's function of the same signature).
    returns (rc, [(pairs int32[n, cameras of pKF1 + cameras of pKF2], nmatches)] per neighbour)."""
    recs = np.concatenate([k.rec for k in kf2s])
    cap = int(pair_capacity if pair_capacity is not None else max(2 * len(kf1.keys), 1))
    stride = kf1.n_cams + max(k.n_cams for k in kf2s)
    pairs = np.zeros((len(kf2s), cap, stride), np.int32)
    n_pairs, n_matches = np.zeros(len(kf2s), np.int32), np.zeros(len(kf2s), np.int32)
    rc = fn(kf1.rec.ctypes.data, recs.ctypes.data, len(kf2s), int(bool(only_stereo)), int(bool(check_orientation)), cap,
            stride, pairs.ctypes.data, n_pairs.ctypes.data, n_matches.ctypes.data)
    return rc, [(pairs[p, :min(int(n_pairs[p]), cap), :kf1.n_cams + k.n_cams].copy(), int(n_matches[p]))
                for p, k in enumerate(kf2s)]


def SearchForTriangulation(kf1, kf2s, bOnlyStereo=False, mbCheckOrientation=True, pair_capacity=None):
    """int ORBmatcher::SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo) for every neighbour pKF2 of
    kf2s at once (LocalMapping::CreateNewMapPoints, LocalMapping.cc:650-720).
    returns [(vMatchedPairs as int32[n, cameras] -- (idx1, idx2) for undistorted key frames, one key or -1 per
    camera of pKF1 then of pKF2 for rigs --, the reference's return value)]."""
    rc, out = tri_call(lib().vieo_search_for_triangulation, kf1, kf2s, bOnlyStereo, mbCheckOrientation, pair_capacity)
    check(rc, "vieo_search_for_triangulation")
    return out


# ---------------------------------------------------------------------------------------------- seeded scenes
def make_tri_scene(seed, n_points=900, n_neighbours=3, n_nodes=300, mapped_frac=0.4, stereo_frac=0.5, flip_bits=12,
                   pixel_noise=0.5, n_levels=8, node_noise=0.1, dup_frac=0.0, rig=None):
    """One key frame and `n_neighbours` others looking at the same cloud from poses a few decimetres apart.  Every
    point has a 256-bit descriptor; a view flips `flip_bits` random bits of it.  Vocabulary nodes: a random node
    per point, re-drawn in a view with probability `node_noise` (then the pair cannot be found, as with a real
    vocabulary).  returns (kf1, [kf2...], truth) with truth[p] = {(idx1, idx2): point} of the unmapped pairs that
    share a node.  dup_frac: that share of the points copies the descriptor of another point of its node (look-alikes:
    only the geometry gates and the order of the loops decide then).  rig = "radtan" | "kb8": distorted keys of a
    camera rig (synth_ba.camera_rig), every camera sees its own share of the points; truth then holds
    {(cam1, idx1, cam2, idx2): point}."""
    from . import synth_ba
    rng = np.random.default_rng(seed)
    K = (458.654, 457.296, 367.215, 248.375)
    W, H = 752, 480
    rig_c = synth_ba.camera_rig(rig, with_tcr=True) if rig else None
    if rig_c:
        W, H = rig_c[1]
    scale = 1.2 ** np.arange(n_levels)
    sigma2 = (scale * scale).astype(np.float32)
    X = np.stack([rng.uniform(-6, 6, n_points), rng.uniform(-4, 4, n_points), rng.uniform(4, 14, n_points)], 1)
    desc0 = rng.integers(0, 256, (n_points, 32), dtype=np.uint8)
    node0 = rng.integers(0, n_nodes, n_points)
    for i in np.flatnonzero(rng.random(n_points) < dup_frac):
        same = np.flatnonzero(node0 == node0[i])
        desc0[i] = desc0[rng.choice(same)]
    mapped = rng.random(n_points) < mapped_frac
    level = rng.integers(0, n_levels, n_points)

    def view(vseed, Rcw, tcw):
        r = np.random.default_rng(vseed)
        Xr = X @ Rcw.T + tcw  # reference-camera frame
        if rig_c is None:
            Xc = Xr
            u = K[0] * Xc[:, 0] / Xc[:, 2] + K[2]
            v = K[1] * Xc[:, 1] / Xc[:, 2] + K[3]
            vis = (Xc[:, 2] > 0.5) & (u > 10) & (u < W - 10) & (v > 10) & (v < H - 10) & (r.random(n_points) < 0.9)
            ids = np.flatnonzero(vis)
            r.shuffle(ids)
            kcam = None
        else:  # camera-major key order (mvKeys of a rig); a point may be a key of several cameras
            ids, us, vs, kcam, zs = [], [], [], [], []
            for ci, Tcr in enumerate(rig_c[2]):
                Xci = Xr @ Tcr[:3, :3].T + Tcr[:3, 3]
                sel = np.flatnonzero((Xci[:, 2] > 0.5) & (r.random(n_points) < 0.8))
                r.shuffle(sel)
                for m in sel:
                    uu, vv = synth_ba.project_camera(rig_c[0][ci], Xci[m])
                    if 10 < uu < W - 10 and 10 < vv < H - 10 and np.hypot(uu - W / 2, vv - H / 2) < 0.45 * W:
                        ids.append(m), us.append(uu), vs.append(vv), kcam.append(ci), zs.append(Xci[m, 2])
            ids = np.array(ids, int)
            u, v = np.zeros(n_points), np.zeros(n_points)
            Xc = np.zeros((n_points, 3))
        n = len(ids)
        keys = np.zeros(n, KEYPOINT_DTYPE)
        sc = scale[level[ids]]
        if rig_c is None:
            keys["x"] = u[ids] + r.normal(0, pixel_noise, n) * sc
            keys["y"] = v[ids] + r.normal(0, pixel_noise, n) * sc
            depth = Xc[ids, 2]
        else:
            keys["x"] = np.array(us) + r.normal(0, pixel_noise, n) * sc
            keys["y"] = np.array(vs) + r.normal(0, pixel_noise, n) * sc
            depth = np.array(zs)
        keys["octave"] = level[ids]
        keys["size"] = 31 * sc
        keys["angle"] = (37.0 + 3.0 * r.normal(0, 1, n) + 360.0 * (r.random(n) < 0.05) * r.random(n)) % 360.0
        keys["class_id"] = -1
        d = desc0[ids].copy()
        for i in range(n):
            bits = r.choice(256, flip_bits, replace=False)
            np.bitwise_xor.at(d[i], bits >> 3, (1 << (bits & 7)).astype(np.uint8))
        ur = np.where(r.random(n) < stereo_frac, np.abs(keys["x"] - 47.9 / depth), -1.0).astype(np.float32)
        node = node0[ids].copy()
        redo = r.random(n) < node_noise
        node[redo] = r.integers(0, n_nodes, int(redo.sum()))
        fv = [(int(nd), [int(i) for i in np.flatnonzero(node == nd)]) for nd in np.unique(node)]
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = Rcw, tcw
        kf = TriKeyFrame(T, K, keys, d, ur, mapped[ids], fv, scale.astype(np.float32), sigma2,
                         rig=None if rig_c is None else (rig_c[0], rig_c[2], kcam))
        return kf, ids, node

    def pose(r, spread):
        w = r.normal(0, 0.03, 3)
        th = np.linalg.norm(w)
        Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        R = np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th ** 2 * Kx @ Kx
        return R, r.normal(0, spread, 3) * np.array([1.0, 0.4, 0.3])

    R1, t1 = pose(rng, 0.05)
    kf1, ids1, node1 = view(seed * 100 + 1, R1, t1)
    kf2s, truth = [], []
    for p in range(n_neighbours):
        R2, t2 = pose(rng, 0.5)
        kf2, ids2, node2 = view(seed * 100 + 2 + p, R2, t2)
        tr = {}
        if rig_c is None:
            where2 = {int(pt): i for i, pt in enumerate(ids2)}
            for i1, pt in enumerate(ids1):
                i2 = where2.get(int(pt))
                if i2 is not None and not mapped[pt] and node1[i1] == node2[i2]:
                    tr[(i1, i2)] = int(pt)
        else:
            where2 = {}
            for i2, pt in enumerate(ids2):
                where2.setdefault(int(pt), []).append(i2)
            for i1, pt in enumerate(ids1):
                for i2 in where2.get(int(pt), []):
                    if not mapped[pt] and node1[i1] == node2[i2]:
                        tr[(int(kf1.key_cam[i1]), i1, int(kf2.key_cam[i2]), i2)] = int(pt)
        kf2s.append(kf2)
        truth.append(tr)
    return kf1, kf2s, truth
