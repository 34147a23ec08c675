"""Tracking-side ORBmatcher::SearchByProjection on frames of a camera rig (SURVEY 8 a12-a14 with the camera loops
of ORBmatcher.cc:1339-1366, :257-266, :1491-1543): oracle known-answer tests (CPU) and HIP-vs-oracle parity (GPU,
queries byte-equal, assignments equal -- no tolerance) for 2- and 4-camera Radtan and KB8 rigs in all three modes."""
import numpy as np
import pytest

from vieo_slam_amd import frontend, synth_ba
from vieo_slam_amd import synth_fisheye as sf
from vieo_slam_amd.ba_types import SBP_RIG_DTYPE
from vieo_slam_amd.map_point import FRUSTUM_FRAME_DTYPE, FRUSTUM_POINT_DTYPE

RIGS = [("radtan", 2), ("radtan", 4), ("kb8", 2), ("kb8", 4)]


def _frustum_frame(C):
    """FRUSTUM_FRAME_DTYPE[1] of the case's current frame (float casts of the double poses, Frame.cc:348-351)."""
    F = np.zeros(1, FRUSTUM_FRAME_DTYPE)
    f, R = F[0], C["rig"][0]
    Tc = C["Tcw"]
    f["Rcrw"], f["tcrw"], f["Ow"] = Tc[:, :3].reshape(-1), Tc[:, 3], -Tc[:, :3].T @ Tc[:, 3]
    f["n_cams"], f["use_distort"], f["cams"] = R["n_cams"], 1, C["cams"].ctypes.data
    for c in range(int(R["n_cams"])):
        f["Tcr"][c], f["trc"][c], f["bounds"][c] = R["Tcr"][c], R["trc"][c], R["bounds"][c]
    f["bf"], f["n_levels"], f["viewing_cos_limit"] = C["cam"][0]["bf"], len(C["scale"]), 0.5
    f["log_scale_factor"] = np.float32(C["log_scale_factor"])
    return F


def _local_map_queries(oracle, C, th=3.0):
    P = np.zeros(len(C["kf_pts"]), FRUSTUM_POINT_DTYPE)
    P["Xw"] = C["kf_pts"]["Xw"]
    Ow = _frustum_frame(C)[0]["Ow"]
    n = C["kf_pts"]["Xw"] - Ow
    P["normal"] = n / np.linalg.norm(n, axis=1)[:, None]
    P["max_distance"], P["min_distance"] = C["kf_pts"]["max_distance"], C["kf_pts"]["min_distance"]
    info = oracle.is_in_frustum(_frustum_frame(C), P)
    obs = (C["pts"]["flags"] & 2) != 0
    return frontend.queries_from_track_info(info, C["kf_pts"]["desc"], th, C["scale"], obs)


# ------------------------------------------------------------------ oracle (CPU)
def test_oracle_rig_of_one_equals_single_camera(oracle):
    """a rig of one undistorted camera with identity Tcr is the single-camera search, bit for bit"""
    from tests.test_proj_search import BOUNDS, K, _scenario
    kl, dl, ur, pts, cam = _scenario(oracle, 1000)
    rig = np.zeros(1, SBP_RIG_DTYPE)
    R = rig[0]
    R["n_cams"], R["use_distort"] = 1, 0
    R["cams"][0]["fx"], R["cams"][0]["fy"], R["cams"][0]["cx"], R["cams"][0]["cy"] = K
    R["Tcr"][0] = np.eye(4)[:3].reshape(-1)
    R["bounds"][0] = BOUNDS
    q0 = oracle.sbp_project_last_frame(pts, cam)
    q1 = oracle.sbp_project_last_frame(pts, cam, rig)
    assert np.array_equal(q0.view(np.uint8), q1.view(np.uint8)) and ((q0["flags"] & 1) > 0).sum() > 300
    for mode in (0, 1, 2):
        n0, a0 = oracle.search_by_projection(mode, q0, kl, ur, dl, None, BOUNDS, nn_ratio=0.8 if mode < 2 else 80.0)
        n1, a1 = oracle.search_by_projection(mode, q1, kl, ur, dl, None, BOUNDS[None], nn_ratio=0.8 if mode < 2 else 80.0,
                                             cam_first=[0, len(kl)])
        assert n0 == n1 and np.array_equal(a0, a1) and n0 > 50


@pytest.mark.parametrize("rig,nc", RIGS)
def test_oracle_rig_projection_and_camera_separation(oracle, rig, nc):
    C = sf.make_rig_tracking_case(3, rig, nc, n_points=300)
    q = oracle.sbp_project_last_frame(C["pts"], C["cam"], C["rig"]).reshape(-1, nc)
    valid = (q["flags"] & 1) > 0
    assert valid.sum() > 150 * nc / 2
    assert np.array_equal(((q["flags"] >> 8) & 15)[valid], np.broadcast_to(np.arange(nc), q.shape)[valid])
    # the projections are those of the float64 python camera models
    Tc = C["Tcw"]
    for i in np.nonzero(valid.any(1))[0][:60]:
        for c in np.nonzero(valid[i])[0]:
            T = C["rig"][0]["Tcr"][c].reshape(3, 4)
            Pc = T[:, :3] @ (Tc[:, :3] @ C["pts"]["Xw"][i].astype(np.float64) + Tc[:, 3]) + T[:, 3]
            u, v = synth_ba.project_camera(C["cams"][c], Pc)
            assert abs(u - q["u"][i, c]) < 2e-3 and abs(v - q["v"][i, c]) < 2e-3
    # a key is only ever claimed by a query of its own camera
    qf = q.reshape(-1)
    n, a = oracle.search_by_projection(0, qf, C["keys"], C["uright"], C["desc"], None, C["bounds"],
                                       cam_first=C["cam_first"], check_ori=False)
    got = np.nonzero(a >= 0)[0]
    key_cam = np.searchsorted(C["cam_first"], got, side="right") - 1
    assert n == len(got) > 100 and np.array_equal((qf["flags"][a[got]] >> 8) & 15, key_cam)
    # removing camera 1's queries leaves the other cameras' matches as they were (claims never cross cameras)
    q2 = qf.copy()
    q2["flags"][((q2["flags"] >> 8) & 15) == 1] = 0
    n2, a2 = oracle.search_by_projection(0, q2, C["keys"], C["uright"], C["desc"], None, C["bounds"],
                                         cam_first=C["cam_first"], check_ori=False)
    other = key_cam != 1
    assert np.array_equal(a2[got[other]], a[got[other]]) and np.all(a2[got[~other]] == -1)


def test_oracle_rig_rotation_histogram_is_shared(oracle):
    """one histogram over all cameras (ORBmatcher.cc:1309-1311,1446-1464): the bins kept are those of the frame"""
    C = sf.make_rig_tracking_case(4, "kb8", 4, n_points=400)
    q = oracle.sbp_project_last_frame(C["pts"], C["cam"], C["rig"])
    n, a = oracle.search_by_projection(0, q, C["keys"], C["uright"], C["desc"], None, C["bounds"], cam_first=C["cam_first"])
    n0, a0 = oracle.search_by_projection(0, q, C["keys"], C["uright"], C["desc"], None, C["bounds"],
                                         cam_first=C["cam_first"], check_ori=False)
    erased = np.nonzero(a == -2)[0]
    assert len(erased) > 5 and n == n0 - len(erased) and np.array_equal(a[a >= 0], a0[a >= 0])
    rot = (q["angle"][a0[a0 >= 0]] - C["keys"]["angle"][a0 >= 0]) % 360
    bins = np.round(rot.astype(np.float32) * np.float32(1 / 30)).astype(int) % 30
    kept_bins = set(bins[np.isin(np.nonzero(a0 >= 0)[0], np.nonzero(a >= 0)[0])].tolist())
    assert len(kept_bins) <= 3


def test_oracle_keyframe_projection_levels(oracle):
    C = sf.make_rig_tracking_case(6, "radtan", 2, n_points=300)
    q = oracle.sbp_project_keyframe(C["kf_pts"], C["cam"], C["rig"], C["log_scale_factor"]).reshape(-1, 2)
    v = (q["flags"] & 1) > 0
    assert v.sum() > 200
    # the generator put every point at the distance of its level: predicted level = level (+-1 after the motion)
    lv = q["level_min"] + 1
    assert np.all(np.abs(lv[v] - np.broadcast_to(C["kf_pts"]["octave"][:, None], lv.shape)[v]) <= 1)
    assert np.all(q["level_max"][v] - q["level_min"][v] == 2)


# ------------------------------------------------------------------ GPU parity
def _m(nn=0.6, ori=True):
    from vieo_slam_amd.matching import ORBmatcher
    return ORBmatcher(nn, ori)


@pytest.mark.gpu
@pytest.mark.parametrize("rig,nc", RIGS)
@pytest.mark.parametrize("seed", [11, 12])
def test_gpu_rig_last_frame_parity(oracle, rig, nc, seed):
    motion = (0.03, -0.01, 0.05) if seed == 11 else (-0.02, 0.01, 0.3)  # the second one: bForward
    C = sf.make_rig_tracking_case(seed, rig, nc, n_points=900 if nc == 4 else 1200, motion=motion, th=7.0 + 8 * (seed % 2))
    oq = oracle.sbp_project_last_frame(C["pts"], C["cam"], C["rig"])
    m = _m()
    hq = m.project_last_frame(C["pts"], C["cam"], C["rig"])
    assert np.array_equal(oq.view(np.uint8), hq.view(np.uint8))
    rng = np.random.default_rng(seed)
    for taken in (None, (rng.random(len(C["keys"])) < 0.25).astype(np.uint8)):
        on, oa = oracle.search_by_projection(0, oq, C["keys"], C["uright"], C["desc"], taken, C["bounds"],
                                             cam_first=C["cam_first"])
        hn, ha = m.SearchByProjectionLastFrame(hq, C["keys"], C["uright"], C["desc"], taken, C["bounds"], C["cam_first"])
        assert on == hn and np.array_equal(oa, ha) and on > 200
    on, oa = oracle.search_by_projection(0, oq, C["keys"], C["uright"], C["desc"], None, C["bounds"],
                                         cam_first=C["cam_first"], check_ori=False)
    hn, ha = _m(0.6, False).SearchByProjectionLastFrame(hq, C["keys"], C["uright"], C["desc"], None, C["bounds"],
                                                        C["cam_first"])
    assert on == hn and np.array_equal(oa, ha)


@pytest.mark.gpu
@pytest.mark.parametrize("rig,nc", RIGS)
@pytest.mark.parametrize("mode", [0, 1])
def test_gpu_rig_assignment_forms_agree(oracle, rig, nc, mode, monkeypatch):
    """The order-dependent assignment of a rig frame, four ways through the library: one workgroup per camera with rounds
    flat over the candidate pairs (the default; the rotation histogram, which spans the cameras, is finished by the last
    workgroup to arrive), one workgroup per frame with a thread per query (VIEO_SBP_FLAT=0), the sequential replay alone,
    and the flat form giving up after one round.  Look-alike descriptors and wide windows make the claims depend on the
    order; all equal the oracle bit for bit."""
    C = sf.make_rig_tracking_case(31 + mode, rig, nc, n_points=900 if nc == 4 else 1200, th=20.0)
    rng = np.random.default_rng(310 + mode)
    fam = rng.integers(0, 256, (10, 32), dtype=np.uint8)
    desc = fam[rng.integers(0, 10, len(C["keys"]))].copy()
    flip = rng.integers(0, 256, len(desc))
    desc[np.arange(len(desc)), flip % 32] ^= (1 << (flip % 8)).astype(np.uint8)
    if mode == 0:
        pts = C["pts"].copy()
        pts["desc"] = fam[rng.integers(0, 10, len(pts))]
        pts["flags"] = np.where(rng.random(len(pts)) < 0.15, pts["flags"] & 1, pts["flags"])  # some without observations
        q = oracle.sbp_project_last_frame(pts, C["cam"], C["rig"])
        nn = 0.9
    else:
        q, _ = _local_map_queries(oracle, C, th=8.0)
        q["desc"] = fam[rng.integers(0, 10, len(q))]
        nn = 0.95
    taken = (rng.random(len(C["keys"])) < 0.1).astype(np.uint8)
    on, oa = oracle.search_by_projection(mode, q, C["keys"], C["uright"], desc, taken, C["bounds"], nn_ratio=nn,
                                         cam_first=C["cam_first"])
    assert on > 100
    m = _m(nn)
    call = m.SearchByProjectionLastFrame if mode == 0 else m.SearchByProjectionLocalMap
    for name, env in (("flat per camera", {}), ("thread per query", {"VIEO_SBP_FLAT": "0"}),
                      ("sequential", {"VIEO_SBP_ASSIGN": "seq"}), ("fallback", {"VIEO_SBP_MAX_ROUNDS": "1"})):
        for k in ("VIEO_SBP_FLAT", "VIEO_SBP_ASSIGN", "VIEO_SBP_MAX_ROUNDS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        hn, ha = call(q, C["keys"], C["uright"], desc, taken, C["bounds"], C["cam_first"])
        assert hn == on and np.array_equal(ha, oa), name
    for k in ("VIEO_SBP_FLAT", "VIEO_SBP_ASSIGN", "VIEO_SBP_MAX_ROUNDS"):
        monkeypatch.delenv(k, raising=False)


@pytest.mark.gpu
@pytest.mark.parametrize("rig,nc", RIGS)
def test_gpu_rig_local_map_parity(oracle, rig, nc):
    from vieo_slam_amd.map_point import is_in_frustum
    C = sf.make_rig_tracking_case(21, rig, nc, n_points=1000)
    q, owner = _local_map_queries(oracle, C)
    assert len(q) > 400 and len(np.unique((q["flags"] >> 8) & 15)) == nc
    rng = np.random.default_rng(21)
    for nn in (0.8, 0.6):
        for taken in (None, (rng.random(len(C["keys"])) < 0.3).astype(np.uint8)):
            on, oa = oracle.search_by_projection(1, q, C["keys"], C["uright"], C["desc"], taken, C["bounds"], nn_ratio=nn,
                                                 cam_first=C["cam_first"])
            hn, ha = _m(nn).SearchByProjectionLocalMap(q, C["keys"], C["uright"], C["desc"], taken, C["bounds"],
                                                       C["cam_first"])
            assert on == hn and np.array_equal(oa, ha) and on > 100


@pytest.mark.gpu
@pytest.mark.parametrize("rig,nc", RIGS)
def test_gpu_rig_reloc_parity(oracle, rig, nc):
    C = sf.make_rig_tracking_case(31, rig, nc, n_points=1000, th=10.0)
    oq = oracle.sbp_project_keyframe(C["kf_pts"], C["cam"], C["rig"], C["log_scale_factor"])
    m = _m()
    hq = m.project_keyframe(C["kf_pts"], C["cam"], C["rig"], C["log_scale_factor"])
    assert np.array_equal(oq.view(np.uint8), hq.view(np.uint8)) and ((oq["flags"] & 1) > 0).sum() > 500
    rng = np.random.default_rng(31)
    for orbdist, taken in ((100, None), (64, (rng.random(len(C["keys"])) < 0.4).astype(np.uint8))):
        on, oa = oracle.search_by_projection(2, oq, C["keys"], C["uright"], C["desc"], taken, C["bounds"],
                                             nn_ratio=float(orbdist), cam_first=C["cam_first"])
        hn, ha = m.SearchByProjectionKeyFrame(hq, C["keys"], C["uright"], C["desc"], taken, C["bounds"], orbdist,
                                              C["cam_first"])
        assert on == hn and np.array_equal(oa, ha) and on > 100


@pytest.mark.gpu
def test_gpu_keyframe_projection_single_camera(oracle):
    """vieo_sbp_project_keyframe without a rig: the rectified camera of the single-camera entries"""
    from tests.test_proj_search import BOUNDS, _scenario
    from vieo_slam_amd.ba_types import KEYFRAME_POINT_DTYPE
    kl, dl, ur, pts, cam = _scenario(oracle, 1003, th=10.0)
    kf = np.zeros(len(pts), KEYFRAME_POINT_DTYPE)
    for f in ("Xw", "angle", "flags", "desc"):
        kf[f] = pts[f]
    d = np.linalg.norm(pts["Xw"].astype(np.float64), axis=1)
    kf["max_distance"] = (d * 1.2 ** kl["octave"]).astype(np.float32)
    kf["min_distance"] = kf["max_distance"] / np.float32(1.2 ** 7)
    lsf = float(np.log(np.float32(1.2)))
    oq = oracle.sbp_project_keyframe(kf, cam, None, lsf)
    hq = _m().project_keyframe(kf, cam, None, lsf)
    assert np.array_equal(oq.view(np.uint8), hq.view(np.uint8)) and ((oq["flags"] & 1) > 0).sum() > 300
    on, oa = oracle.search_by_projection(2, oq, kl, ur, dl, None, BOUNDS, nn_ratio=100.0)
    hn, ha = _m().SearchByProjectionKeyFrame(hq, kl, ur, dl, None, BOUNDS, 100)
    assert on == hn and np.array_equal(oa, ha) and on > 100


@pytest.mark.gpu
def test_gpu_rig_search_rejects_bad_arguments():
    from vieo_slam_amd._lib import VieoError
    C = sf.make_rig_tracking_case(5, "radtan", 2, n_points=100)
    q = _m().project_last_frame(C["pts"], C["cam"], C["rig"])
    bad = C["cam_first"].copy()
    bad[-1] -= 1
    with pytest.raises(VieoError):
        _m().SearchByProjectionLastFrame(q, C["keys"], C["uright"], C["desc"], None, C["bounds"], bad)
    rig = C["rig"].copy()
    rig[0]["cams"][1]["model"] = 7
    with pytest.raises(VieoError):
        _m().project_last_frame(C["pts"], C["cam"], rig)
