"""Map-point steps next to the hot path: Frame::isInFrustum, ComputeDistinctiveDescriptors,
UpdateNormalAndDepth.  Oracle known-answer tests (CPU) and HIP-vs-oracle parity (GPU; float results bit-equal:
both sides evaluate the same float expressions in the same order)."""
import numpy as np
import pytest

from vieo_slam_amd import synth_ba
from vieo_slam_amd.map_point import FRUSTUM_FRAME_DTYPE, FRUSTUM_POINT_DTYPE


def _frame(rng, rig=None):
    """a frame at a random pose; rig = None: one rectified pinhole camera, else the distorted cameras of a rig"""
    F = np.zeros(1, FRUSTUM_FRAME_DTYPE)
    f = F[0]
    Rcw = synth_ba.quat_to_R(synth_ba.quat_from_rotvec(rng.normal(0, 0.4, 3)))
    tcw = rng.uniform(-1, 1, 3)
    f["Rcrw"], f["tcrw"], f["Ow"] = Rcw.reshape(-1), tcw, -Rcw.T @ tcw
    if rig is None:
        from vieo_slam_amd.ba_types import CAMERA_DTYPE
        cams = np.zeros(1, CAMERA_DTYPE)
        cams[0]["fx"], cams[0]["fy"], cams[0]["cx"], cams[0]["cy"] = synth_ba.FX, synth_ba.FY, synth_ba.CX, synth_ba.CY
        size, Tcr = (synth_ba.W, synth_ba.H), [np.eye(4)]
        f["use_distort"] = 0
    else:
        cams, size, Tcr = synth_ba.camera_rig(rig, with_tcr=True)
        f["use_distort"] = 1
    f["n_cams"], f["cams"] = len(cams), cams.ctypes.data
    for c, T in enumerate(Tcr):
        f["Tcr"][c] = T[:3, :].reshape(-1)
        f["trc"][c] = np.linalg.inv(T)[:3, 3]
        f["bounds"][c] = (0, size[0], 0, size[1])
    f["bf"], f["n_levels"], f["viewing_cos_limit"] = synth_ba.BF, 8, 0.5
    f["log_scale_factor"] = np.float32(np.log(np.float32(1.2)))
    return F, cams, Rcw, tcw


def _points(rng, Rcw, tcw, n):
    P = np.zeros(n, FRUSTUM_POINT_DTYPE)
    Xc = np.stack([rng.uniform(-8, 8, n), rng.uniform(-5, 5, n), rng.uniform(-2, 14, n)], 1)
    Xw = (Xc - tcw) @ Rcw  # Rcw^T (Xc - tcw)
    P["Xw"] = Xw
    Ow = -Rcw.T @ tcw
    d = np.linalg.norm(Xw - Ow, axis=1)
    nrm = (Xw - Ow) / d[:, None] + rng.normal(0, 0.5, (n, 3))
    P["normal"] = nrm / np.linalg.norm(nrm, axis=1)[:, None]
    lvl = rng.integers(0, 8, n)
    ref_dist = d * rng.uniform(0.4, 2.5, n)  # distance at which the point was created
    P["max_distance"] = ref_dist * 1.2 ** lvl
    P["min_distance"] = P["max_distance"] / np.float32(1.2 ** 7)
    return P


def _obs_problem(rng, n_points, n_centres=40, max_obs=20):
    pts = rng.uniform(-5, 5, (n_points, 3)).astype(np.float32)
    centres = rng.uniform(-6, 6, (n_centres, 3)).astype(np.float32)
    cnt = rng.integers(0, max_obs + 1, n_points)
    cnt[0] = 0
    first = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    obs_centre = rng.integers(0, n_centres, first[-1]).astype(np.int32)
    ref_centre = np.array([obs_centre[first[p]] if cnt[p] else 0 for p in range(n_points)], np.int32)
    ref_scale = (np.float32(1.2) ** rng.integers(0, 8, n_points)).astype(np.float32)
    return pts, first, obs_centre, centres, ref_centre, ref_scale


# ------------------------------------------------------------------ oracle (CPU)
def test_oracle_frustum_gates(oracle):
    rng = np.random.default_rng(1)
    F, cams, Rcw, tcw = _frame(rng)
    P = _points(rng, Rcw, tcw, 4000)
    info = oracle.is_in_frustum(F, P)
    Xc = P["Xw"].astype(np.float64) @ Rcw.T + tcw
    u = synth_ba.FX * Xc[:, 0] / Xc[:, 2] + synth_ba.CX
    v = synth_ba.FY * Xc[:, 1] / Xc[:, 2] + synth_ba.CY
    d = np.linalg.norm(P["Xw"] - F[0]["Ow"], axis=1)
    cosv = ((P["Xw"] - F[0]["Ow"]) * P["normal"]).sum(1) / d
    exp = ((Xc[:, 2] >= 0) & (u >= 0) & (u <= synth_ba.W) & (v >= 0) & (v <= synth_ba.H) &
           (d >= 0.8 * P["min_distance"]) & (d <= 1.2 * P["max_distance"]) & (cosv >= 0.5))
    near = ((np.abs(u) < 1e-2) | (np.abs(u - synth_ba.W) < 1e-2) | (np.abs(v) < 1e-2) | (np.abs(v - synth_ba.H) < 1e-2)
            | (np.abs(cosv - 0.5) < 1e-5) | (np.abs(d / (0.8 * P["min_distance"]) - 1) < 1e-5)
            | (np.abs(d / (1.2 * P["max_distance"]) - 1) < 1e-5))
    got = info["n"] > 0
    assert (got == exp)[~near].all() and 200 < got.sum() < 3000
    k = got & ~near
    assert np.allclose(info["u"][k, 0], u[k], atol=2e-3) and np.allclose(info["ur"][k, 0], u[k] - synth_ba.BF / Xc[k, 2], atol=2e-3)
    assert np.allclose(info["track_depth"][k], d[k], rtol=1e-6) and (info["track_depth"][~got] == -1).all()
    # PredictScale: ceil(log(max/d) / log 1.2) clamped to [0, 7]
    lv = np.clip(np.ceil(np.log(P["max_distance"][k].astype(np.float64) / d[k]) / np.log(1.2)), 0, 7)
    assert (np.abs(info["level"][k, 0] - lv) <= 1).all() and (info["level"][k, 0] == lv).mean() > 0.999


def test_oracle_distinctive_descriptor_is_the_medoid(oracle):
    rng = np.random.default_rng(2)
    first, rows, expect = [0], [], []
    for p in range(60):
        N = int(rng.integers(0, 40)) if p else 0
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        D = np.repeat(base[None], N, 0)
        for i in range(N):
            for b in rng.integers(0, 256, rng.integers(0, 80)):
                D[i, b >> 3] ^= np.uint8(1 << (b & 7))
        rows.append(D)
        first.append(first[-1] + N)
        if N:
            dist = np.unpackbits(D[:, None] ^ D[None], axis=2).sum(2)
            med = np.sort(dist, 1)[:, int(0.5 * (N - 1))]
            expect.append(int(np.argmin(med)))  # first minimum
        else:
            expect.append(-1)
    best = oracle.distinctive_descriptors(np.concatenate(rows), first)
    assert best.tolist() == expect


def test_oracle_normal_and_depth(oracle):
    rng = np.random.default_rng(3)
    pts, first, oc, centres, rc, rs = _obs_problem(rng, 300)
    nrm, mx, mn = oracle.update_normal_and_depth(pts, first, oc, centres, rc, rs, np.float32(1.2) ** 7)
    for p in (1, 7, 100, 299):
        if first[p + 1] == first[p]:
            continue
        d = pts[p] - centres[oc[first[p]:first[p + 1]]]
        e = (d / np.linalg.norm(d, axis=1)[:, None]).mean(0)
        assert np.allclose(nrm[p], e, atol=1e-5)
        dist = np.linalg.norm(pts[p] - centres[rc[p]])
        assert np.isclose(mx[p], dist * rs[p], rtol=1e-6) and np.isclose(mn[p], mx[p] / 1.2 ** 7, rtol=1e-6)
    assert mx[0] == -1 and (nrm[0] == 0).all()  # no observations: members untouched


# ------------------------------------------------------------------ parity (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize("rig", [None, "radtan", "kb8"])
def test_frustum_parity(oracle, rig):
    from vieo_slam_amd.map_point import is_in_frustum
    rng = np.random.default_rng(11)
    F, cams, Rcw, tcw = _frame(rng, rig)
    P = _points(rng, Rcw, tcw, 20000)
    o, h = oracle.is_in_frustum(F, P), is_in_frustum(F, P)
    assert o["n"].sum() > 1000
    # float expressions in the same order: bit-equal records.  Distorted rigs too: Radtan has no transcendental
    # function, KB8 one double atan2 whose last-bit differences (libm vs device) vanish in the float rounding of u, v
    for k in o.dtype.names:
        assert np.array_equal(o[k], h[k]), (rig, k, int((o[k] != h[k]).sum()))
    assert o.tobytes() == h.tobytes()
    if rig == "kb8":
        assert (o["n"] > 1).any()  # some points are seen by several cameras of the rig


@pytest.mark.gpu
def test_distinctive_parity(oracle):
    from vieo_slam_amd._lib import lib
    from vieo_slam_amd.map_point import compute_distinctive_descriptors, distinctive_call
    rng = np.random.default_rng(12)
    cnt = rng.integers(0, 129, 3000)
    cnt[:4] = (0, 1, 2, 128)
    first = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    base = rng.integers(0, 256, (len(cnt), 32), dtype=np.uint8)
    D = np.repeat(base, cnt, 0)
    flip = rng.random(D.shape) < 0.08  # near-duplicates: many equal medians, ties resolved by the first row
    D ^= (flip * (1 << rng.integers(0, 8, D.shape))).astype(np.uint8)
    o = oracle.distinctive_descriptors(D, first)
    h = compute_distinctive_descriptors(D, first)
    assert np.array_equal(o, h) and (h[cnt == 0] == -1).all()
    # more than 128 observations of one point (4-camera rigs, long sessions): no limit in the reference; the kernel
    # switches from the N x N table in LDS to per-lane histograms, and the other points of the batch are unaffected
    cnt2 = np.array([129, 5, 700, 0, 128, 301], np.int64)
    first2 = np.concatenate([[0], np.cumsum(cnt2)]).astype(np.int32)
    D2 = np.repeat(rng.integers(0, 256, (len(cnt2), 32), dtype=np.uint8), cnt2, 0)
    D2 ^= ((rng.random(D2.shape) < 0.1) * (1 << rng.integers(0, 8, D2.shape))).astype(np.uint8)
    assert np.array_equal(oracle.distinctive_descriptors(D2, first2), compute_distinctive_descriptors(D2, first2))
    # small batches ask for small LDS (the table of the largest point, not 128 KB)
    small = np.array([0, 3, 7], np.int32)
    assert np.array_equal(oracle.distinctive_descriptors(D2[:7], small), compute_distinctive_descriptors(D2[:7], small))


@pytest.mark.gpu
def test_normal_depth_parity(oracle):
    from vieo_slam_amd.map_point import update_normal_and_depth
    rng = np.random.default_rng(13)
    pts, first, oc, centres, rc, rs = _obs_problem(rng, 50000, n_centres=300, max_obs=30)
    s7 = float(np.float32(1.2) ** 7)
    o = oracle.update_normal_and_depth(pts, first, oc, centres, rc, rs, s7)
    h = update_normal_and_depth(pts, first, oc, centres, rc, rs, s7)
    for a, b in zip(o, h):
        assert a.tobytes() == b.tobytes()


# ------------------------------------------------------------------ SearchByProjectionBase / Fuse
def _fuse_case(rng, rig, n_points=3000, use_bf=True, check_angle=True):
    """a key frame whose keys are the (noisy) projections of part of the map points + distractors"""
    from vieo_slam_amd.map_point import FUSE_FRAME_DTYPE, FUSE_POINT_DTYPE
    from vieo_slam_amd.orb_extractor import KEYPOINT_DTYPE
    F0, cams, Rcw, tcw = _frame(rng, rig)
    FF = np.zeros(1, FUSE_FRAME_DTYPE)
    FF[0]["base"] = F0[0]
    sc = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    FF[0]["scale_factors"][:8], FF[0]["inv_level_sigma2"][:8] = sc, 1.0 / (sc * sc)
    FF[0]["th_radius"], FF[0]["check_viewing_angle"], FF[0]["use_bf"] = 3.0, int(check_angle), int(use_bf)
    P0 = _points(rng, Rcw, tcw, n_points)
    P = np.zeros(n_points, FUSE_POINT_DTYPE)
    for k in ("Xw", "normal", "max_distance", "min_distance"):
        P[k] = P0[k]
    P["desc"] = rng.integers(0, 256, (n_points, 32), dtype=np.uint8)
    P["skip_mask"] = np.where(rng.random(n_points) < 0.05, np.int32(-2 ** 31), 0) | np.where(rng.random(n_points) < 0.05, 1, 0)
    nc = len(cams)
    keys, urs, descs = [], [], []
    for c in range(nc):
        Tcr = FF[0]["base"]["Tcr"][c].reshape(3, 4).astype(np.float64)
        kk, uu, dd = [], [], []
        for m in range(n_points):
            Pc = Tcr[:, :3] @ (Rcw @ P["Xw"][m].astype(np.float64) + tcw) + Tcr[:, 3]
            if Pc[2] < 0.3 or rng.random() < 0.4:
                continue
            if rig is None:
                u, v = synth_ba.FX * Pc[0] / Pc[2] + synth_ba.CX, synth_ba.FY * Pc[1] / Pc[2] + synth_ba.CY
            else:
                if np.hypot(Pc[0], Pc[1]) / Pc[2] > (0.9 if cams[c]["model"] == 1 else 2.0):
                    continue
                u, v = synth_ba.project_camera(cams[c], Pc)
            b = FF[0]["base"]["bounds"][c]
            if not (b[0] + 1 < u < b[1] - 1 and b[2] + 1 < v < b[3] - 1):
                continue
            d3 = np.linalg.norm(P["Xw"][m] - FF[0]["base"]["Ow"])
            lvl = int(np.clip(np.ceil(np.log(P["max_distance"][m] / d3) / np.log(1.2)), 0, 7)) - int(rng.integers(0, 2))
            lvl = max(lvl, 0)
            s = 1.2 ** lvl
            d = P["desc"][m].copy()
            for bit in rng.integers(0, 256, rng.integers(0, 30)):
                d[bit >> 3] ^= np.uint8(1 << (bit & 7))
            ku, kv = u + rng.normal(0, 0.8) * s, v + rng.normal(0, 0.8) * s
            kk.append((ku, kv, 31 * s, 0, 20, lvl, -1))
            uu.append(ku - synth_ba.BF / Pc[2] + rng.normal(0, 0.8) * s if (rig is None and rng.random() < 0.7) else -1.0)
            dd.append(d)
        for _ in range(len(kk) // 2 + 5):  # distractors
            b = FF[0]["base"]["bounds"][c]
            kk.append((rng.uniform(b[0], b[1]), rng.uniform(b[2], b[3]), 31, 0, 20, int(rng.integers(0, 8)), -1))
            uu.append(-1.0)
            dd.append(rng.integers(0, 256, 32, dtype=np.uint8))
        order = rng.permutation(len(kk))[:4000]
        keys.append(np.array([kk[i] for i in order], KEYPOINT_DTYPE))
        urs.append(np.array([uu[i] for i in order], np.float32))
        descs.append(np.stack([dd[i] for i in order]).astype(np.uint8))
    return FF, keys, urs, descs, P, cams


def test_oracle_fuse_search_finds_the_projected_keys(oracle):
    rng = np.random.default_rng(21)
    FF, keys, urs, descs, P, cams = _fuse_case(rng, None, n_points=1500)
    bi, bd = oracle.fuse_search(FF, keys, urs, descs, P)
    found = bi[:, 0] >= 0
    assert 150 < found.sum() < 1200
    # a found key is inside the window and carries the point's signature (<= 30 flipped bits) far more often than not
    assert (bd[found, 0] <= 30).mean() > 0.85
    assert (bi[P["skip_mask"] < 0, 0] == -1).all() and (bi[(P["skip_mask"] & 1) != 0, 0] == -1).all()
    # without the chi2 gate / viewing cone at least as many points find a key
    FF2 = FF.copy()
    FF2[0]["use_bf"], FF2[0]["check_viewing_angle"] = 0, 0
    bi2, _ = oracle.fuse_search(FF2, keys, urs, descs, P)
    assert (bi2[:, 0] >= 0).sum() >= found.sum() and ((bi2[:, 0] >= 0) | ~found).all()


@pytest.mark.gpu
@pytest.mark.parametrize("rig,use_bf,angle", [(None, True, True), (None, False, False), ("radtan", True, True),
                                              ("kb8", True, False)])
def test_fuse_search_parity(oracle, rig, use_bf, angle):
    from vieo_slam_amd.map_point import fuse_search
    rng = np.random.default_rng(22)
    FF, keys, urs, descs, P, cams = _fuse_case(rng, rig, n_points=4000, use_bf=use_bf, check_angle=angle)
    oi, od = oracle.fuse_search(FF, keys, urs, descs, P)
    hi, hd = fuse_search(FF, keys, urs, descs, P)
    assert (oi >= 0).sum() > 300
    # index work: exact for distorted rigs too (see test_frustum_parity)
    assert np.array_equal(oi, hi) and np.array_equal(od, hd), (int((oi != hi).sum()), int((od != hd).sum()))


@pytest.mark.gpu
@pytest.mark.parametrize("rig", [None, "kb8"])
def test_track_local_queries_device(oracle, rig):
    """vieo_track_local_queries_device (the head of Tracking::SearchLocalPoints for a frame whose pose is still in
    HBM): pose taken from a PoseOptimization result on the device -> isInFrustum -> window queries, against
    isInFrustum of the oracle + the host's query construction on the same pose."""
    import ctypes
    from vieo_slam_amd import frontend
    from vieo_slam_amd._lib import DeviceBuffer, check, lib
    from vieo_slam_amd.ba_types import PROJ_QUERY_DTYPE, VIO_FRAME_DTYPE, VIO_RESULT_DTYPE
    rng = np.random.default_rng(21)
    F, cams, _, _ = _frame(rng, rig)
    # the pose as the optimiser leaves it: body state + camera extrinsics; Tcw = Tcb * Twb^-1
    q = synth_ba.quat_from_rotvec(rng.normal(0, 0.4, 3))
    p = rng.uniform(-1, 1, 3)
    Rcb = synth_ba.quat_to_R(synth_ba.quat_from_rotvec(rng.normal(0, 0.2, 3)))
    tcb = rng.uniform(-0.1, 0.1, 3)
    Rcw = Rcb @ synth_ba.quat_to_R(q).T
    tcw = tcb - Rcw @ p
    F[0]["Rcrw"], F[0]["tcrw"], F[0]["Ow"] = Rcw.reshape(-1), tcw, -Rcw.T @ tcw
    n = 6000
    P = _points(rng, Rcw, tcw, n)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    th = 2.0
    S = int(F[0]["n_cams"])
    for status in (0, 1):  # result usable / not: then the frame's own estimate is the pose
        fr, res = np.zeros(1, VIO_FRAME_DTYPE), np.zeros(1, VIO_RESULT_DTYPE)
        fr[0]["base"]["Rcb"], fr[0]["base"]["tcb"] = Rcb.reshape(-1), tcb
        good, junk = (res, fr) if status == 0 else (fr, res)
        good[0]["base"]["nav"]["q"], good[0]["base"]["nav"]["p"] = q, p
        junk[0]["base"]["nav"]["q"], junk[0]["base"]["nav"]["p"] = (1, 0, 0, 0), (50, 50, 50)
        res[0]["base"]["status"] = status
        alias = np.where(rng.random(n) < 0.3, rng.integers(0, 500, n), -1).astype(np.int32)
        held = (rng.random(500) < 0.5).astype(np.uint8)
        D = DeviceBuffer
        d_fr, d_res, d_P, d_desc = D(fr.nbytes), D(res.nbytes), D(P.nbytes), D(desc.nbytes)
        d_alias, d_held, d_scale = D(alias.nbytes), D(held.nbytes), D(64)
        d_q, d_dep, d_nq = D(64 * n * S), D(4 * n), D(16)
        for b, a in ((d_fr, fr), (d_res, res), (d_P, P), (d_desc, desc), (d_alias, alias), (d_held, held), (d_scale, scale)):
            b.upload(a)
        Fz = F.copy()
        Fz[0]["Rcrw"], Fz[0]["tcrw"], Fz[0]["Ow"] = 0, 0, 0  # ignored by the device entry
        check(lib().vieo_track_local_queries_device(Fz.ctypes.data, d_fr.ptr, d_res.ptr, d_P.ptr, d_desc.ptr, d_alias.ptr,
                                                    d_held.ptr, len(held), n, th, 0.0, d_scale.ptr, d_q.ptr, d_dep.ptr, d_nq.ptr, None))
        check(lib().vieo_device_synchronize())
        got = d_q.download(PROJ_QUERY_DTYPE, (n * S,))
        dep = d_dep.download(np.float32, (n,))
        assert int(d_nq.download(np.int32, (4,))[0]) == n * S
        info = oracle.is_in_frustum(F, P)
        assert np.array_equal(dep, info["track_depth"])
        excluded = (alias >= 0) & (held[np.maximum(alias, 0)] != 0)
        info_x = info.copy()
        info_x["n"][excluded] = 0
        ref, owner = frontend.queries_from_track_info(info_x, desc, th, scale)
        valid = np.nonzero(got["flags"] & 1)[0]
        assert len(ref) > 300 and len(valid) == len(ref)
        assert got[valid].tobytes() == ref.tobytes()
        assert np.array_equal(valid // S, owner)
        assert not got[(got["flags"] & 1) == 0].view(np.uint8).any()  # unused slots are zero


@pytest.mark.gpu
def test_track_local_queries_batch_equals_single_frame_calls():
    """vieo_track_local_queries_batch_device (bench.py's batched step): every frame of the batch gets exactly what the
    one-frame entry gives for its pose / candidates / held table, ragged candidate counts included."""
    from vieo_slam_amd._lib import DeviceBuffer, check, lib
    from vieo_slam_amd.ba_types import PROJ_QUERY_DTYPE, VIO_FRAME_DTYPE, VIO_RESULT_DTYPE
    rng = np.random.default_rng(31)
    F, cams, _, _ = _frame(rng, None)
    B, pcap, hcap = 3, 2500, 400
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    fr, res = np.zeros(B, VIO_FRAME_DTYPE), np.zeros(B, VIO_RESULT_DTYPE)
    counts = np.array([2500, 1700, 0], np.int32)
    P = np.zeros((B, pcap), _points(rng, np.eye(3), np.zeros(3), 1).dtype)
    desc = rng.integers(0, 256, (B, pcap, 32), dtype=np.uint8)
    alias = np.where(rng.random((B, pcap)) < 0.3, rng.integers(0, hcap, (B, pcap)), -1).astype(np.int32)
    held = (rng.random((B, hcap)) < 0.5).astype(np.uint8)
    for b in range(B):
        q = synth_ba.quat_from_rotvec(rng.normal(0, 0.4, 3))
        p = rng.uniform(-1, 1, 3)
        Rcb = synth_ba.quat_to_R(synth_ba.quat_from_rotvec(rng.normal(0, 0.2, 3)))
        tcb = rng.uniform(-0.1, 0.1, 3)
        Rcw = Rcb @ synth_ba.quat_to_R(q).T
        fr[b]["base"]["Rcb"], fr[b]["base"]["tcb"] = Rcb.reshape(-1), tcb
        res[b]["base"]["nav"]["q"], res[b]["base"]["nav"]["p"] = q, p
        P[b] = _points(rng, Rcw, tcb - Rcw @ p, pcap)
    D = DeviceBuffer
    bufs = {}
    for name, a in (("fr", fr), ("res", res), ("P", P), ("desc", desc), ("alias", alias), ("held", held), ("scale", scale),
                    ("counts", counts)):
        bufs[name] = D(a.nbytes)
        bufs[name].upload(a)
    d_q, d_dep, d_nq = D(64 * B * pcap), D(4 * B * pcap), D(4 * B)
    check(lib().vieo_track_local_queries_batch_device(F.ctypes.data, bufs["fr"].ptr, bufs["res"].ptr, B, bufs["P"].ptr,
                                                      bufs["desc"].ptr, bufs["alias"].ptr, bufs["counts"].ptr, pcap,
                                                      bufs["held"].ptr, hcap, 2.0, 0.0, bufs["scale"].ptr, d_q.ptr, d_dep.ptr,
                                                      pcap, d_nq.ptr, None))
    check(lib().vieo_device_synchronize())
    got_q = d_q.download(PROJ_QUERY_DTYPE, (B, pcap))
    got_d = d_dep.download(np.float32, (B, pcap))
    got_n = d_nq.download(np.int32, (B,))
    assert got_n.tolist() == counts.tolist()
    s_q, s_dep, s_nq = D(64 * pcap), D(4 * pcap), D(16)
    for b in range(B):
        n = int(counts[b])
        if n == 0:
            continue
        check(lib().vieo_track_local_queries_device(F.ctypes.data, bufs["fr"].ptr + b * fr.itemsize,
                                                    bufs["res"].ptr + b * res.itemsize, bufs["P"].ptr + b * pcap * P.itemsize,
                                                    bufs["desc"].ptr + b * pcap * 32, bufs["alias"].ptr + 4 * b * pcap,
                                                    bufs["held"].ptr + b * hcap, hcap, n, 2.0, 0.0, bufs["scale"].ptr,
                                                    s_q.ptr, s_dep.ptr, s_nq.ptr, None))
        check(lib().vieo_device_synchronize())
        assert s_q.download(PROJ_QUERY_DTYPE, (pcap,))[:n].tobytes() == got_q[b, :n].tobytes()
        assert np.array_equal(s_dep.download(np.float32, (pcap,))[:n], got_d[b, :n])
        assert (got_q[b, :n]["flags"] & 1).sum() > 100
