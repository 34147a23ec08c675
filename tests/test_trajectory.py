"""Trajectory file formats (System::Save*Trajectory*) and the ATE evaluation the reference shells out to
(SURVEY.md 8f-4): known-answer tests, CPU only."""
import numpy as np
import pytest

from vieo_slam_amd import synth_ba, trajectory
from vieo_slam_amd.ba_types import NAVSTATE_DTYPE


def _traj(rng, n=200):
    t = 1403636579.0 + 0.05 * np.arange(n)
    p = np.cumsum(rng.normal(0, 0.05, (n, 3)), 0)
    return t, p


def test_align_recovers_a_rigid_motion_exactly():
    rng = np.random.default_rng(0)
    _, p = _traj(rng)
    R = synth_ba.quat_to_R(synth_ba.quat_from_rotvec(np.array([0.3, -0.8, 1.1])))
    t = np.array([2.0, -1.0, 0.5])
    data = (R @ p.T).T + t
    rot, trans, err = trajectory.align(p.T, data.T)
    assert np.allclose(rot, R, atol=1e-12) and np.allclose(trans, t, atol=1e-12) and err.max() < 1e-12
    # a mirrored cloud must still give a proper rotation (the S = diag(1, 1, -1) branch)
    rot, _, _ = trajectory.align(p.T, (p * np.array([1, 1, -1])).T)
    assert np.isclose(np.linalg.det(rot), 1.0)
    # with a scale: rotation first, then s = <data, R model> / |model|^2
    rot, trans_s, err_s, trans, err, s = trajectory.align(p.T, (2.5 * (R @ p.T).T + t).T, True)
    assert np.isclose(s, 2.5) and err_s.max() < 1e-11 and err.max() > 1e-2


def test_associate_is_greedy_closest_first_and_one_to_one():
    a = [0.0, 1.0, 2.0, 3.0]
    b = [0.012, 0.992, 1.005, 2.5, 3.019]
    m = trajectory.associate(a, b, 0.0, 0.02)
    assert m == [(0.0, 0.012), (1.0, 1.005), (3.0, 3.019)]  # 0.992 loses against 1.005, 2.5 is too far
    assert trajectory.associate(a, b, 0.5, 0.02) == [(3.0, 2.5)]  # the offset is added to the second file's stamps
    assert trajectory.associate(a, [], 0.0, 0.02) == []


def test_navstate_file_round_trip_and_ate(tmp_path):
    rng = np.random.default_rng(1)
    stamps, p = _traj(rng, 300)
    navs = np.zeros(len(stamps), NAVSTATE_DTYPE)
    navs["p"] = p
    navs["q"] = [synth_ba.quat_from_rotvec(rng.normal(0, 0.5, 3)) for _ in stamps]
    navs["v"] = rng.normal(0, 1, (len(stamps), 3))
    navs["bg"], navs["dbg"] = 0.01, 0.002
    navs["ba"], navs["dba"] = -0.1, 0.03
    f = tmp_path / "KeyFrameTrajectoryIMU.txt"
    trajectory.write_trajectory_navstate(f, stamps, navs)
    first = open(f).readline().strip().split(" ")
    assert len(first) == 17 and all(len(c.split(".")[1]) == 9 for c in first)  # std::fixed << setprecision(9)
    rec = trajectory.read_trajectory(f)
    assert len(rec) == len(stamps)
    row = np.array(rec[float("%.9f" % stamps[7])])
    assert np.allclose(row[:3], p[7], atol=1e-9)
    assert np.allclose(row[3:7], navs["q"][7][[1, 2, 3, 0]], atol=1e-9)  # file order qx qy qz qw
    assert np.allclose(row[10:13], 0.012, atol=1e-9) and np.allclose(row[13:16], -0.07, atol=1e-9)  # b + db
    # ground truth = the same trajectory in another frame, sampled 4 ms later, with 1 cm noise
    R = synth_ba.quat_to_R(synth_ba.quat_from_rotvec(np.array([0.1, 0.2, -0.4])))
    noise = rng.normal(0, 0.01, p.shape)
    g = tmp_path / "groundtruth.txt"
    with open(g, "w") as fh:
        fh.write("# timestamp tx ty tz qx qy qz qw\n")
        for t, x in zip(stamps + 0.004, (R @ (p + noise).T).T + 3.0):
            fh.write("%.6f,%.9f,%.9f,%.9f,0,0,0,1\n" % (t, *x))
    r = trajectory.evaluate_ate(g, f)
    assert r["compared_pose_pairs"] == len(stamps)
    assert 0.012 < r["rmse"] < 0.022 and r["min"] >= 0 and r["max"] < 0.06  # |N(0, 1 cm)^3| has rms 1.73 cm
    assert np.allclose(r["rot"], R, atol=5e-3)
    assert trajectory.evaluate_ate(g, f, offset=0.2)["compared_pose_pairs"] < len(stamps)  # shifted out of 20 ms
    s = trajectory.evaluate_ate(g, f, with_scale=True)
    assert abs(s["scale"] - 1) < 5e-3 and s["rmse"] <= s["rmse_no_scale"] + 1e-12
    with pytest.raises(ValueError):
        trajectory.evaluate_ate(g, f, offset=100.0)


def test_tum_writers_and_gravity_alignment(tmp_path):
    stamps = [1.5, 2.25]
    twc = np.array([[1.0, 2.0, 3.0], [0.1234567891, -4.0, 5.0]])
    q = np.array([[0, 0, 0, 1.0], [0.5, 0.5, 0.5, 0.5]])
    f = tmp_path / "CameraTrajectory.txt"
    trajectory.write_trajectory_tum(f, stamps, twc, q, bg=np.zeros((2, 3)), ba=np.ones((2, 3)))
    cols = open(f).read().split("\n")[1].split(" ")
    assert cols[0] == "2.250000" and len(cols) == 14 and cols[1] == "%.9f" % np.float32(0.1234567891)
    trajectory.write_trajectory_tum(f, stamps, twc, q, keyframes=True)
    cols = open(f).read().split("\n")[0].split(" ")
    assert cols == ["1.500000", "1.0000000", "2.0000000", "3.0000000", "0.0000000", "0.0000000", "0.0000000", "1.0000000"]
    gw = np.array([0.3, -9.7, 1.2])
    RIw = trajectory.gravity_alignment(gw)
    assert np.allclose(RIw @ RIw.T, np.eye(3), atol=1e-12)
    assert np.allclose(RIw @ gw / np.linalg.norm(gw), [0, 0, 1], atol=1e-12)
    assert np.array_equal(trajectory.gravity_alignment([0, 0, 9.81]), np.eye(3))
