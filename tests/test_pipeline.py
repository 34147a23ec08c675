"""GPU: the device-resident batched front-end replay (vieo_slam_amd/pipeline.py, what bench.py
times) against the same chain evaluated stage by stage with the CPU oracle."""
import numpy as np
import pytest

from vieo_slam_amd import synth_ba
from vieo_slam_amd import synth_scene as sc
from vieo_slam_amd.ba_types import POSE_OBS_DTYPE

pytestmark = pytest.mark.gpu
BOUNDS = np.array([0, 752, 0, 480], np.float32)


def _obs_from(mp_ref, xyz, keys, ur, inv_sigma2):
    idx = np.nonzero(mp_ref >= 0)[0]
    obs = np.zeros(len(idx), POSE_OBS_DTYPE)
    obs["Xw"] = xyz[mp_ref[idx]]
    obs["u"], obs["v"], obs["ur"] = keys["x"][idx], keys["y"][idx], ur[idx]
    obs["inv_sigma2"] = inv_sigma2[keys["octave"][idx]]
    return obs, idx


def test_batched_pipeline_matches_oracle_chain(oracle):
    from vieo_slam_amd.pipeline import FramePipeline, make_cases
    cases = make_cases(2, seed0=3)
    B = 5  # frames 2..4 are noisy replicas of the 2 base cases
    P = FramePipeline(cases, B, seed=7)
    P.step()
    P.step()  # a second pass over the resident batch must give the same answer
    R = P.results()
    cap = P.cap
    xyz = P.d_xyz.download(np.float32, (B, 2 * cap, 3))
    oL, oR = oracle.extractor(1200), oracle.extractor(1200)
    for b in range(B):
        _, k1, d1 = oL(P.imgs_host[b, 0])
        _, kr, dr = oR(P.imgs_host[b, 1])
        n = len(k1)
        assert R["counts"][2 * b, 0] == n
        assert np.array_equal(R["kps"][2 * b, :n].view(np.uint8), k1.view(np.uint8))
        ur, _ = oracle.stereo_match(oL, oR, k1, d1, kr, dr, sc.BASELINE, sc.BF)
        assert np.array_equal(ur.view(np.uint32), R["uright"][b, :n].view(np.uint32))
        npts = int((P.pts_host[b]["flags"] != 0).sum() + (P.pts_host[b]["flags"] == 0)[:0].sum())
        n0 = int(np.count_nonzero(np.any(P.pts_host[b]["desc"] != 0, axis=1)))
        pts = P.pts_host[b][:n0]
        cam = np.array([P.f1_host[b]])  # placeholder to keep names short
        q1 = oracle.sbp_project_last_frame(pts, np.array([_cam(P, b)]))
        _, a1 = oracle.search_by_projection(0, q1, k1, ur, d1, None, BOUNDS)
        mp = np.where(a1 >= 0, a1, -1)
        obs1, idx1 = _obs_from(mp, xyz[b], k1, ur, P.inv_sigma2)
        F1 = np.array([P.f1_host[b]])
        F1[0]["base"]["n_obs"] = len(obs1)
        r1, o1 = oracle.pose_optimization_vio(F1, obs1)
        mp[idx1[o1 != 0]] = -1
        taken = (mp >= 0).astype(np.uint8)
        _, a2 = oracle.search_by_projection(1, P.q2_host[b][:n0], k1, ur, d1, taken, BOUNDS, nn_ratio=0.8)
        mp = np.where(a2 >= 0, cap + a2, mp)
        obs2, idx2 = _obs_from(mp, xyz[b], k1, ur, P.inv_sigma2)
        F2 = F1.copy()
        F2[0]["base"]["nav"] = r1["base"]["nav"]
        F2[0]["base"]["n_obs"] = len(obs2)
        F2[0]["compute_marg"] = 1
        r2, o2 = oracle.pose_optimization_vio(F2, obs2)
        assert np.array_equal(R["mp_ref"][b, :n], mp)
        for name, ref, got in (("r1", r1, R["r1"][b]), ("r2", r2, R["r2"][b])):
            dt, dr = synth_ba.pose_error(ref["base"]["nav"], got["base"]["nav"])
            assert dt < 1e-4 and dr < 1e-4, (name, b, dt, dr)
            assert ref["base"]["n_inliers"] == got["base"]["n_inliers"], (name, b)
        Ho, Hh = r2["H_marg"].reshape(15, 15), R["r2"][b]["H_marg"].reshape(15, 15)
        assert np.allclose(Ho, Hh, rtol=1e-5, atol=1e-5 * np.abs(Ho).max())
        gdt, gdr = synth_ba.pose_error(R["r2"][b]["base"]["nav"], P.truth[b])
        assert gdt < 3e-3 and gdr < 2e-3, (b, gdt, gdr)
        assert R["r2"][b]["base"]["n_inliers"] > 150


def _cam(P, b):
    from vieo_slam_amd.ba_types import SBP_CAMERA_DTYPE
    return P.d_cams.download(SBP_CAMERA_DTYPE, (P.B,))[b]


def test_batched_pipeline_r3_local_map_stage_on_device(oracle):
    """The r3 workload of bench.py: non-planar scenes with low-contrast patches, and SearchLocalPoints' head inside the
    step -- isInFrustum + window queries of 4 k candidates per frame from the first optimisation's pose in HBM.  Checked
    against the same chain on the CPU oracle (tests/pipeline_check.py, the checker bench.py's parity_sample uses too):
    is_in_frustum at the oracle's first pose, the host's query construction, the second search, the second optimisation."""
    from tests.pipeline_check import R3FrameChecker
    from vieo_slam_amd.pipeline import FramePipeline, make_cases
    cases = make_cases(2, seed0=5, workload="r3")
    assert len(cases[0]["scene"].sheets) == 12
    B = 3
    P = FramePipeline(cases, B, seed=9, workload="r3")
    P.step()
    C = R3FrameChecker(P, P.results(), oracle)
    retried = 0
    for b in range(B):
        c = C.check(b)
        assert c["keys_equal"] and c["uright_equal"] and c["matches_equal"] and c["inliers_equal"], c
        assert c["max_se3_error"] < 1e-4 and c["n_local_queries"] > 500, c
        assert c["error_vs_truth"][0] < 5e-3 and c["error_vs_truth"][1] < 3e-3, c
        retried += c["retried"]
    assert retried > 0, "no cell needed minThFAST: the low-contrast patches are not doing their job"
