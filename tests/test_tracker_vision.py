"""The one-call frame tracker WITHOUT the IMU (vieo_tracker_params.vision_only; BASELINE configs[0]: stereo, no IMU, 1000
features): Tracking::TrackWithMotionModel (src/Tracking.cc:1844-1928) + TrackLocalMap (:1930-1945) -- ExtractORB x 2 ->
ComputeStereoMatches -> SearchByProjection(last frame) -> Optimizer::PoseOptimization(Frame*, Frame*) -> discard outliers ->
SearchLocalPoints -> PoseOptimization -- as ONE call, against the same chain issued stage by stage through the
host-pointer entries and against that chain on the CPU oracle."""
import numpy as np
import pytest

from vieo_slam_amd import frontend, synth_ba
from vieo_slam_amd import synth_scene as sc
from vieo_slam_amd.ba_types import POSE_FRAME_DTYPE, POSE_OBS_DTYPE
from vieo_slam_amd.map_point import FRUSTUM_FRAME_DTYPE, FRUSTUM_POINT_DTYPE

K = (sc.FX, sc.FY, sc.CX, sc.CY)
BOUNDS = np.array([0, 752, 0, 480], np.float32)
NFEAT = 1000


class Hip:
    def __init__(self):
        from vieo_slam_amd.matching import ORBmatcher
        from vieo_slam_amd.orb_extractor import ORBextractor
        self.ex = [ORBextractor(NFEAT, 1.2, 8, 20, 7) for _ in range(2)]
        self.M = ORBmatcher

    def extract(self, i, img):
        return self.ex[i](img)[1:]

    def stereo(self, kl, dl, kr, dr):
        from vieo_slam_amd.matching import compute_stereo_matches
        return compute_stereo_matches(self.ex[0], self.ex[1], kl, dl, kr, dr, sc.BASELINE, sc.BF)

    def scale_factors(self):
        return self.ex[0].GetScaleFactors()

    def project(self, pts, cam):
        return self.M.project_last_frame(pts, cam)

    def search(self, mode, q, k, ur, d, taken, nn):
        return self.M(nn, True)._search(mode, q, k, ur, d, taken, BOUNDS)

    def pose(self, F, obs):
        from vieo_slam_amd.optimizer import Optimizer
        return Optimizer.PoseOptimization(F, obs)

    def in_frustum(self, F, P):
        from vieo_slam_amd.map_point import is_in_frustum
        return is_in_frustum(F, P)


class Orc:
    def __init__(self, oracle):
        self.o = oracle
        self.ex = [oracle.extractor(NFEAT) for _ in range(2)]

    def extract(self, i, img):
        return self.ex[i](img)[1:]

    def stereo(self, kl, dl, kr, dr):
        return self.o.stereo_match(self.ex[0], self.ex[1], kl, dl, kr, dr, sc.BASELINE, sc.BF)

    def scale_factors(self):
        return np.array(self.ex[0].scale_factors(), np.float32)

    def project(self, pts, cam):
        return self.o.sbp_project_last_frame(pts, cam)

    def search(self, mode, q, k, ur, d, taken, nn):
        return self.o.search_by_projection(mode, q, k, ur, d, taken, BOUNDS, nn_ratio=nn)

    def pose(self, F, obs):
        return self.o.pose_optimization(F, obs)

    def in_frustum(self, F, P):
        return self.o.is_in_frustum(F, P)


def _Tcw(nav, Tbc):
    Rwb = synth_ba.quat_to_R(nav["q"])
    return frontend.pose_to_Tcw(Rwb @ Tbc[:3, :3], nav["p"] + Rwb @ Tbc[:3, 3])


def _obs(keys, ur, mp_ref, Xw, inv_sigma2):
    idx = np.nonzero(mp_ref >= 0)[0]
    obs = np.zeros(len(idx), POSE_OBS_DTYPE)
    obs["Xw"] = Xw[mp_ref[idx]]
    obs["u"], obs["v"], obs["ur"] = keys["x"][idx], keys["y"][idx], ur[idx]
    obs["inv_sigma2"] = inv_sigma2[keys["octave"][idx]]
    return obs, idx


def staged(B, case, last, nav_pred, th_last=7.0, th_local=2.0):
    """TrackWithMotionModel + TrackLocalMap stage by stage; `last` = (keys0, points, Xw, P) of the previous frame."""
    k0, pts, Xw, P, sel = last
    L1, R1 = case["images1"]
    k1, d1 = B.extract(0, L1)
    k1r, d1r = B.extract(1, R1)
    ur1, dp1 = B.stereo(k1, d1, k1r, d1r)
    scf = B.scale_factors()
    inv_sigma2 = (np.float32(1.0) / (scf * scf)).astype(np.float32)
    Tbc = synth_ba.EUROC_TBC
    nav_i = case["vio"][0]["nav_last"]
    cam = frontend.make_sbp_camera(_Tcw(nav_pred, Tbc), _Tcw(nav_i, Tbc), K, BOUNDS, sc.BF, sc.BASELINE, th_last, scf)
    q1 = B.project(pts, cam)
    n1, a1 = B.search(0, q1, k1, ur1, d1, None, 0.9)
    mp_ref = np.where(a1 >= 0, a1, -1).astype(np.int64)
    F = np.zeros(1, POSE_FRAME_DTYPE)
    F[0] = case["vio"][0]["base"]
    F[0]["nav"] = nav_pred
    obs1, idx1 = _obs(k1, ur1, mp_ref, Xw, inv_sigma2)
    F[0]["n_obs"], F[0]["obs_begin"] = len(obs1), 0
    r1, o1 = B.pose(F, obs1)
    mp_ref[idx1[o1 != 0]] = -1
    nav1 = r1["nav"] if int(r1["status"]) == 0 else F[0]["nav"]
    Tcw1 = _Tcw(nav1, Tbc)
    FF = np.zeros(1, FRUSTUM_FRAME_DTYPE)
    f = FF[0]
    f["Rcrw"], f["tcrw"], f["Ow"] = Tcw1[:, :3].reshape(-1), Tcw1[:, 3], -Tcw1[:, :3].T @ Tcw1[:, 3]
    from vieo_slam_amd.ba_types import CAMERA_DTYPE
    cams = np.zeros(1, CAMERA_DTYPE)
    cams[0]["fx"], cams[0]["fy"], cams[0]["cx"], cams[0]["cy"] = K
    f["n_cams"], f["use_distort"], f["cams"] = 1, 0, cams.ctypes.data
    f["Tcr"][0] = np.eye(3, 4, dtype=np.float32).reshape(-1)
    f["bounds"][0] = BOUNDS
    f["bf"], f["n_levels"], f["viewing_cos_limit"] = sc.BF, 8, 0.5
    f["log_scale_factor"] = np.float32(np.log(np.float32(1.2)))
    in_frame = np.zeros(len(P), bool)
    in_frame[mp_ref[mp_ref >= 0]] = True
    cand = sel[~in_frame[sel]]  # the local map = the last frame's valid points
    info = B.in_frustum(FF, P[cand])
    q2, owner = frontend.queries_from_track_info(info, pts["desc"][cand], th_local, scf)
    taken = (mp_ref >= 0).astype(np.uint8)
    n2, a2 = B.search(1, q2, k1, ur1, d1, taken, 0.8)
    ok = a2 >= 0
    mp_ref[ok] = cand[owner[a2[ok]]]
    obs2, idx2 = _obs(k1, ur1, mp_ref, Xw, inv_sigma2)
    F2 = F.copy()
    F2[0]["nav"], F2[0]["n_obs"] = nav1, len(obs2)
    r2, o2 = B.pose(F2, obs2)
    return dict(k1=k1, d1=d1, ur1=ur1, dp1=dp1, n1=n1, n2=n2, mp_ref=mp_ref, r1=r1, r2=r2, o2=o2, idx2=idx2, keep=cams)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2])
def test_gpu_vision_tracker_equals_the_staged_chain_and_the_oracle(oracle, seed):
    from vieo_slam_amd import replay as rp
    from vieo_slam_amd.tracker import Tracker, euroc_params
    case = sc.make_tracking_case(seed)
    H = Hip()
    L0, R0 = case["images0"]
    k0, d0 = H.extract(0, L0)
    k0r, d0r = H.extract(1, R0)
    ur0, dp0 = H.stereo(k0, d0, k0r, d0r)
    Ri, pi, Rwc0, twc0 = case["pose0"]
    Xw, valid = frontend.unproject_stereo(k0, dp0, K, Rwc0, twc0)
    pts = frontend.make_last_frame_points(k0, d0, Xw, valid, True)
    # the local map = the last frame's points (each aliasing its key), as FRUSTUM points seen from the last camera centre
    P = np.zeros(len(k0), FRUSTUM_POINT_DTYPE)
    P["Xw"] = Xw
    d = Xw.astype(np.float64) - twc0
    dist = np.maximum(np.linalg.norm(d, axis=1), 1e-6)
    P["normal"] = (d / dist[:, None]).astype(np.float32)
    scf = H.scale_factors()
    P["max_distance"] = (dist * scf[k0["octave"]]).astype(np.float32)
    P["min_distance"] = P["max_distance"] / scf[7]
    sel = np.nonzero(valid)[0]
    rng = np.random.default_rng(seed)
    nav_pred = case["vio"][0]["base"]["nav"].copy()
    nav_pred["p"] += rng.normal(0, 0.01, 3)
    nav_pred["q"] = synth_ba.quat_mul(nav_pred["q"], synth_ba.quat_from_rotvec(rng.normal(0, 0.003, 3)))
    prm = euroc_params(max_local_points=len(sel) + 10)
    prm[0]["n_features"], prm[0]["vision_only"] = NFEAT, 1
    trk = Tracker(prm)
    nav_i = case["vio"][0]["nav_last"]
    L1, R1 = case["images1"]
    o, v = trk.track(L1, R1, np.zeros(0, np.dtype([("t", "<f8"), ("w", "<f8", 3), ("a", "<f8", 3)])), 0.0, 0.05, nav_pred, nav_i,
                     None, pts, np.full(len(pts), np.inf, np.float32), P[sel], pts["desc"][sel], sel.astype(np.int32), 1)
    assert int(o["status"]) == 0 and int(o["widened"]) == 0
    cap = int(o["key_cap"])
    tab = v["point_ref"].astype(np.int64)
    held = np.full(len(tab), -1, np.int64)
    a = (tab >= 0) & (tab < cap)
    held[a] = tab[a]
    held[tab >= cap] = sel[tab[tab >= cap] - cap]
    last = (k0, pts, Xw, P, sel)
    for name, B in (("hip", H), ("oracle", Orc(oracle))):
        ref = staged(B, case, last, nav_pred)
        assert np.array_equal(v["keys"].view(np.uint8), ref["k1"].view(np.uint8)) and np.array_equal(v["desc"], ref["d1"]), name
        assert np.array_equal(v["uright"].view(np.uint32), ref["ur1"].view(np.uint32)), name
        assert np.array_equal(v["depth"].view(np.uint32), ref["dp1"].view(np.uint32)), name
        assert int(o["n_matches_last"]) == ref["n1"] > 100 and int(o["n_matches_local"]) == ref["n2"], name
        assert np.array_equal(held, ref["mp_ref"]), name
        out2 = np.zeros(len(tab), np.uint8)
        out2[ref["idx2"]] = ref["o2"]
        assert np.array_equal(v["outlier"], out2), name
        for which, r in (("first", ref["r1"]), ("second", ref["r2"])):
            dt, dr = synth_ba.pose_error(o[which]["base"]["nav"], r["nav"])
            assert dt < 1e-4 and dr < 1e-4, (name, which, dt, dr)
            assert int(o[which]["base"]["n_inliers"]) == int(r["n_inliers"]), (name, which)
    gdt, gdr = synth_ba.pose_error(o["second"]["base"]["nav"], case["truth"])
    assert gdt < 2e-2 and gdr < 3e-3, (gdt, gdr)  # (vision only, the map = one frame's stereo depths)
    # fewer than 20 matches: TrackWithMotionModel gives up (status LOST) after the widened search
    far = nav_pred.copy()
    far["q"] = synth_ba.quat_mul(far["q"], synth_ba.quat_from_rotvec(np.array([np.pi, 0.0, 0.0])))  # looking away
    o3, _ = trk.track(L1, R1, np.zeros(0, np.dtype([("t", "<f8"), ("w", "<f8", 3), ("a", "<f8", 3)])), 0.0, 0.05, far, nav_i, None,
                      pts, np.full(len(pts), np.inf, np.float32), P[sel], pts["desc"][sel], sel.astype(np.int32), 1)
    assert int(o3["widened"]) == 1 and int(o3["status"]) == 2 and int(o3["n_matches_last"]) < 20
    print("vision tracker: %.2f ms in the call (GPU %.2f)" % (float(o["ms_host"]), float(o["ms_gpu"])))
    trk.close()
