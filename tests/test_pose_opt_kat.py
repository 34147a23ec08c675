"""CPU known-answer tests pinning the pose-optimisation oracle (oracle/pose_opt.cc)."""
import numpy as np
import pytest

from vieo_slam_amd import synth_ba
from vieo_slam_amd.ba_types import POSE_FRAME_DTYPE, POSE_OBS_DTYPE


def test_so3_exp_log_roundtrip_and_small_angle(oracle):
    rng = np.random.default_rng(0)
    for scale in (1e-9, 1e-6, 0.9e-5, 1.1e-5, 1e-3, 0.5, 2.5, 3.1):
        w = rng.normal(0, 1, 3)
        w = w / np.linalg.norm(w) * scale
        q = oracle.so3_exp(w)
        assert abs(np.linalg.norm(q) - 1) < 1e-14
        ref = synth_ba.quat_from_rotvec(w)
        assert np.allclose(q, ref, atol=1e-12)
        assert np.allclose(oracle.so3_log(q), w, atol=1e-10 * max(1, scale / 1e-3))


def test_so3_right_jacobian_and_inverse(oracle):
    rng = np.random.default_rng(1)
    for scale in (1e-7, 1e-3, 0.3, 2.0):
        w = rng.normal(0, 1, 3) * scale
        Jr, Jri = oracle.so3_jr(w), oracle.so3_jr(w, inverse=True)
        assert np.allclose(Jr @ Jri, np.eye(3), atol=1e-9)
        # Exp(w + dw) ~= Exp(w) Exp(Jr dw)
        dw = rng.normal(0, 1, 3) * 1e-6
        lhs = synth_ba.quat_from_rotvec(w + dw)
        rhs = synth_ba.quat_mul(synth_ba.quat_from_rotvec(w), synth_ba.quat_from_rotvec(Jr @ dw))
        assert np.allclose(lhs, rhs, atol=1e-10) or np.allclose(lhs, -rhs, atol=1e-10)


def test_reprojection_jacobian_matches_central_differences(oracle):
    fr, obs, _ = synth_ba.make_pose_problem(3, n_obs=40, outlier_frac=0)
    for i in range(0, 40, 3):
        _, J = oracle.pose_edge_eval(fr, obs[i:i + 1])
        de = 2 if obs[i]["ur"] < 0 else 3
        for k in range(6):
            d = np.zeros(6)
            h = 1e-6
            d[k] = h
            ep, _ = oracle.pose_edge_eval(fr, obs[i:i + 1], d, want_jac=False)
            em, _ = oracle.pose_edge_eval(fr, obs[i:i + 1], -d, want_jac=False)
            num = (ep - em) / (2 * h)
            assert np.allclose(J[:de, k], num[:de], rtol=1e-4, atol=1e-4), (i, k, J[:de, k], num[:de])


def test_noiseless_scene_recovers_ground_truth(oracle):
    fr, obs, gt = synth_ba.make_pose_problem(5, n_obs=200, outlier_frac=0.0, noise=0.0)
    res, outl = oracle.pose_optimization(fr, obs)
    dt, dr = synth_ba.pose_error(res["nav"], gt)
    # residual floor = float32 rounding of the projection (~3e-5 px) and of the map points
    assert dt < 2e-5 and dr < 5e-6, (dt, dr)
    assert res["n_inliers"] == 200 and not outl.any()


@pytest.mark.parametrize("name", ["radtan", "kb8"])
def test_noiseless_rig_scene_recovers_ground_truth(oracle, name):
    """a20: distorted cameras of a rig; the projection used to make the data is the float64 python
    restatement (synth_ba.project_camera), independent of the oracle's C++ one."""
    rig = synth_ba.camera_rig(name)
    fr, obs, gt = synth_ba.make_pose_problem(9, n_obs=200, outlier_frac=0.0, noise=0.0, rig=rig)
    res, outl = oracle.pose_optimization(fr, obs)
    dt, dr = synth_ba.pose_error(res["nav"], gt)
    assert dt < 5e-5 and dr < 2e-5, (dt, dr)
    assert res["n_inliers"] == 200 and not outl.any()
    Fv, obsv, gtv = synth_ba.make_vio_problem(9, n_obs=200, outlier_frac=0.0, noise=0.0, rig=rig)
    resv, outlv = oracle.pose_optimization_vio(Fv, obsv)
    dt, dr = synth_ba.pose_error(resv["base"]["nav"], gtv)
    assert dt < 2e-3 and dr < 5e-4 and not outlv.any(), (dt, dr)


def test_outliers_are_rejected_and_pose_improves(oracle):
    for seed in range(4):
        fr, obs, gt = synth_ba.make_pose_problem(seed)
        res, outl = oracle.pose_optimization(fr, obs)
        e0 = synth_ba.pose_error(fr[0]["nav"], gt)
        e1 = synth_ba.pose_error(res["nav"], gt)
        assert e1[0] < 0.3 * e0[0] and e1[1] < 0.3 * e0[1]
        gross = gt["is_outlier"]
        assert outl[gross].mean() > 0.9  # planted gross outliers flagged
        assert res["n_inliers"] == len(obs) - int(outl.sum())


def test_too_few_correspondences_and_small_problem(oracle):
    fr, obs, _ = synth_ba.make_pose_problem(7, n_obs=2)
    res, outl = oracle.pose_optimization(fr, obs)
    assert res["n_inliers"] == 0 and res["status"] == 1
    assert np.array_equal(res["nav"]["p"], fr[0]["nav"]["p"])
    # fewer than 10 edges: only the first round runs (Optimizer.cc:1863-1865)
    fr, obs, _ = synth_ba.make_pose_problem(8, n_obs=8, outlier_frac=0)
    res, _ = oracle.pose_optimization(fr, obs)
    assert 0 < res["lm_iterations"] <= 10


def test_struct_sizes_match_header():
    assert POSE_OBS_DTYPE.itemsize == 32 and POSE_FRAME_DTYPE.itemsize == 320


def test_encoder_edge_constrains_a_weak_visual_problem(oracle):
    """Optimizer.cc:1650-1674: with few, noisy correspondences the EdgeEncNavStatePR to the last frame pulls the
    estimate towards the odometry; the edge counts in `edges().size() < 10`."""
    e0 = e1 = 0
    for seed in range(6):
        fr, obs, gt = synth_ba.make_pose_problem(seed, n_obs=30, noise=2.5, outlier_frac=0.0)
        r0, _ = oracle.pose_optimization(fr, obs)
        fr, obs, gt = synth_ba.make_pose_problem(seed, n_obs=30, noise=2.5, outlier_frac=0.0, enc=True)
        r1, _ = oracle.pose_optimization(fr, obs)
        e0 += synth_ba.pose_error(r0["nav"], gt)[0]
        e1 += synth_ba.pose_error(r1["nav"], gt)[0]
    assert e1 < 0.6 * e0
    fr, obs, gt = synth_ba.make_pose_problem(7, n_obs=9, outlier_frac=0.0)
    a, _ = oracle.pose_optimization(fr, obs)       # 9 edges: one round
    fr, obs, gt = synth_ba.make_pose_problem(7, n_obs=9, outlier_frac=0.0, enc=True)
    b, _ = oracle.pose_optimization(fr, obs)       # 10 edges with the encoder edge: four rounds
    assert b["lm_iterations"] > a["lm_iterations"]
