"""End-to-end front-end replay on a geometrically consistent synthetic stereo-inertial pair:
extract x2 -> rectified stereo match -> SearchByProjection(last frame) -> PoseOptimization (VIO),
in the reference's call order.  CPU: the oracle chain recovers the ground-truth
pose.  GPU: every stage of the HIP chain equals the oracle chain and the final pose agrees to 1e-4."""
import numpy as np
import pytest

from vieo_slam_amd import frontend, synth_ba
from vieo_slam_amd import synth_scene as sc

K = (sc.FX, sc.FY, sc.CX, sc.CY)
BOUNDS = np.array([0, 752, 0, 480], np.float32)


class OracleBackend:
    def __init__(self, oracle):
        self.o = oracle
        self.ex = [oracle.extractor(1200) for _ in range(4)]

    def extract(self, i, img):
        return self.ex[i](img)[1:]

    def stereo(self, iL, iR, kl, dl, kr, dr):
        return self.o.stereo_match(self.ex[iL], self.ex[iR], kl, dl, kr, dr, sc.BASELINE, sc.BF)

    def scale_factors(self):
        return np.array(self.ex[0].scale_factors(), np.float32)

    def project(self, pts, cam):
        return self.o.sbp_project_last_frame(pts, cam)

    def search(self, q, k, ur, d):
        return self.o.search_by_projection(0, q, k, ur, d, None, BOUNDS)

    def pose_vio(self, F, obs):
        return self.o.pose_optimization_vio(F, obs)


class HipBackend:
    def __init__(self):
        from vieo_slam_amd.matching import ORBmatcher
        from vieo_slam_amd.orb_extractor import ORBextractor
        self.ex = [ORBextractor(1200, 1.2, 8, 20, 7) for _ in range(4)]
        self.m = ORBmatcher(0.9, True)

    def extract(self, i, img):
        return self.ex[i](img)[1:]

    def stereo(self, iL, iR, kl, dl, kr, dr):
        from vieo_slam_amd.matching import compute_stereo_matches
        return compute_stereo_matches(self.ex[iL], self.ex[iR], kl, dl, kr, dr, sc.BASELINE, sc.BF)

    def scale_factors(self):
        return self.ex[0].GetScaleFactors()

    def project(self, pts, cam):
        return self.m.project_last_frame(pts, cam)

    def search(self, q, k, ur, d):
        return self.m.SearchByProjectionLastFrame(q, k, ur, d, None, BOUNDS)

    def pose_vio(self, F, obs):
        from vieo_slam_amd.optimizer import Optimizer
        return Optimizer.PoseOptimizationVIO(F, obs)


def run_chain(B, case, seed):
    out = {}
    L0, R0 = case["images0"]
    L1, R1 = case["images1"]
    k0, d0 = B.extract(0, L0)
    k0r, d0r = B.extract(1, R0)
    ur0, dp0 = B.stereo(0, 1, k0, d0, k0r, d0r)
    k1, d1 = B.extract(2, L1)
    k1r, d1r = B.extract(3, R1)
    ur1, dp1 = B.stereo(2, 3, k1, d1, k1r, d1r)
    out.update(k0=k0, d0=d0, ur0=ur0, dp0=dp0, k1=k1, d1=d1, ur1=ur1, dp1=dp1)
    Ri, pi, Rwc0, twc0 = case["pose0"]
    Rj, pj, Rwc1, twc1 = case["pose1"]
    Xw, valid = frontend.unproject_stereo(k0, dp0, K, Rwc0, twc0)
    pts = frontend.make_last_frame_points(k0, d0, Xw, valid, True)
    scf = B.scale_factors()
    # predicted pose = truth + perturbation (what PredictNavStateByIMU would supply)
    rng = np.random.default_rng(seed)
    F = case["vio"].copy()
    F[0]["base"]["nav"]["p"] += rng.normal(0, 0.01, 3)
    F[0]["base"]["nav"]["q"] = synth_ba.quat_mul(F[0]["base"]["nav"]["q"],
                                                synth_ba.quat_from_rotvec(rng.normal(0, 0.003, 3)))
    Rwb_pred = synth_ba.quat_to_R(F[0]["base"]["nav"]["q"])
    Tbc = synth_ba.EUROC_TBC
    Rwc_pred = Rwb_pred @ Tbc[:3, :3]
    twc_pred = F[0]["base"]["nav"]["p"] + Rwb_pred @ Tbc[:3, 3]
    cam = frontend.make_sbp_camera(frontend.pose_to_Tcw(Rwc_pred, twc_pred),
                                   frontend.pose_to_Tcw(Rwc0, twc0), K, BOUNDS, sc.BF, sc.BASELINE,
                                   7.0, scf)
    q = B.project(pts, cam)
    n, assign = B.search(q, k1, ur1, d1)
    inv_sigma2 = (np.float32(1.0) / (scf * scf)).astype(np.float32)
    obs, idx = frontend.build_pose_obs(assign, Xw, k1, ur1, inv_sigma2)
    F[0]["base"]["n_obs"] = len(obs)
    F[0]["compute_marg"] = 1
    res, outl = B.pose_vio(F, obs)
    out.update(q=q, nmatch=n, assign=assign, obs=obs, res=res, outl=outl)
    return out


@pytest.mark.parametrize("seed", [1, 2])
def test_oracle_chain_recovers_ground_truth(oracle, seed):
    case = sc.make_tracking_case(seed)
    r = run_chain(OracleBackend(oracle), case, seed)
    assert (r["dp0"] > 0).sum() > 400 and r["nmatch"] > 200
    dt, dr = synth_ba.pose_error(r["res"]["base"]["nav"], case["truth"])
    assert dt < 2e-3 and dr < 1e-3, (dt, dr)
    assert r["res"]["base"]["n_inliers"] > 0.8 * len(r["obs"])
    # stereo depth agrees with the rendered depth
    x = np.clip(np.rint(r["k0"]["x"]).astype(int), 0, 751)
    y = np.clip(np.rint(r["k0"]["y"]).astype(int), 0, 479)
    ok = r["dp0"] > 0
    assert np.median(np.abs(r["dp0"][ok] - case["depth0"][y[ok], x[ok]])) < 0.4


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_hip_chain_matches_oracle_chain(oracle, seed):
    case = sc.make_tracking_case(seed)
    a = run_chain(OracleBackend(oracle), case, seed)
    b = run_chain(HipBackend(), case, seed)
    for name in ("k0", "d0", "k1", "d1"):
        assert np.array_equal(a[name].view(np.uint8), b[name].view(np.uint8)), name
    for name in ("ur0", "dp0", "ur1", "dp1"):
        assert np.array_equal(a[name].view(np.uint32), b[name].view(np.uint32)), name
    assert np.array_equal(a["q"].view(np.uint8), b["q"].view(np.uint8))
    assert a["nmatch"] == b["nmatch"] and np.array_equal(a["assign"], b["assign"])
    assert np.array_equal(a["obs"].view(np.uint8), b["obs"].view(np.uint8))
    dt, dr = synth_ba.pose_error(a["res"]["base"]["nav"], b["res"]["base"]["nav"])
    assert dt < 1e-4 and dr < 1e-4
    assert a["res"]["base"]["n_inliers"] == b["res"]["base"]["n_inliers"]
    assert np.array_equal(a["outl"], b["outl"])
    gdt, gdr = synth_ba.pose_error(b["res"]["base"]["nav"], case["truth"])
    assert gdt < 2e-3 and gdr < 1e-3
