"""LocalBundleAdjustment (vision): CPU known-answer tests of the oracle, GPU parity of the HIP
Schur/MFMA path.  Tolerance 1e-4 on the key-frame poses (BASELINE.json)."""
import numpy as np
import pytest

from vieo_slam_amd import synth_ba


def _pose_errs(navs, truth, n_local):
    return np.array([synth_ba.pose_error(navs[k], dict(p=truth["p"][k], q=truth["q"][k]))
                     for k in range(n_local)])


def test_oracle_lba_improves_poses_and_rejects_outliers(oracle):
    params, kfs, pts, obs, gt = synth_ba.make_lba_problem(0)
    navs, pout, erase, res = oracle.local_ba(params, kfs, pts, obs)
    nl = int((kfs["fixed"] == 0).sum())
    e0 = _pose_errs(kfs["nav"], gt, nl)
    e1 = _pose_errs(navs, gt, nl)
    assert e1[:, 1].mean() < 0.3 * e0[:, 1].mean() and e1[:, 0].mean() < 0.7 * e0[:, 0].mean()
    assert res["chi2_final"] < 0.4 * res["chi2_initial"]
    assert 0.02 * len(obs) < res["n_erase"] < 0.15 * len(obs) and res["n_erase"] == int(erase.sum())
    # fixed key frames are returned untouched
    for k in range(nl, len(kfs)):
        assert np.array_equal(navs[k]["p"], kfs[k]["nav"]["p"]) and np.array_equal(navs[k]["q"], kfs[k]["nav"]["q"])
    # (points start 2 cm from truth, better than ~2 px stereo noise at 2-12 m supports: their
    # error is checked in the noiseless test)
    assert np.isfinite(pout).all()


def test_oracle_lba_noiseless_recovers_truth(oracle):
    params, kfs, pts, obs, gt = synth_ba.make_lba_problem(3, n_points=600, outlier_frac=0.0, noise=0.0,
                                                          stereo_frac=1.0)
    navs, pout, erase, res = oracle.local_ba(params, kfs, pts, obs)
    nl = int((kfs["fixed"] == 0).sum())
    e1 = _pose_errs(navs, gt, nl)
    assert e1[:, 0].max() < 2e-4 and e1[:, 1].max() < 5e-5
    assert res["n_erase"] == 0
    assert np.median(np.linalg.norm(pout - gt["X"], axis=1)) < 2e-3


def test_oracle_lba_edge_cases(oracle):
    params, kfs, pts, obs, gt = synth_ba.make_lba_problem(4, n_local=3, n_fixed=2, n_points=200)
    kfs2 = kfs.copy()
    kfs2["fixed"] = 1  # no free pose: returns silently (Optimizer.cc:1993)
    navs, pout, erase, res = oracle.local_ba(params, kfs2, pts, obs)
    assert res["status"] == 2 and res["lm_iterations"] == 0 and not erase.any()
    assert np.array_equal(pout, pts)
    stop = np.array([1], np.int32)  # abort flag already set
    navs, pout, erase, res = oracle.local_ba(params, kfs, pts, obs, stop=stop)
    assert res["status"] == 1 and res["lm_iterations"] == 0
    # first local key frame fixed (nid_ == 0 case)
    params, kfs, pts, obs, gt = synth_ba.make_lba_problem(5, n_local=4, n_fixed=0, n_points=300, first_fixed=True)
    navs, pout, erase, res = oracle.local_ba(params, kfs, pts, obs)
    assert np.array_equal(navs[0]["p"], kfs[0]["nav"]["p"]) and res["lm_iterations"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed,kw", [(0, {}), (1, {}), (2, dict(n_local=25, n_fixed=10, n_points=2500)),
                                     (6, dict(n_local=4, n_fixed=0, n_points=300, first_fixed=True)),
                                     (7, dict(n_local=3, n_fixed=2, n_points=150, stereo_frac=0.0)),
                                     # 30 key frames x 6 = 180 unknowns: the L2-resident one-workgroup solver with 6-dim blocks
                                     (8, dict(n_local=30, n_fixed=8, n_points=2500)),
                                     # the hand-over between the three solve kernels (lba.hip solver_class): 6-dim blocks,
                                     # 156 | 162 unknowns (k_lba_ldlt16 | k_lba_ldltg), 636 | 642 (k_lba_ldltg | tiled k_big_*)
                                     (9, dict(n_local=26, n_fixed=4, n_points=2000)),
                                     (10, dict(n_local=27, n_fixed=4, n_points=2000)),
                                     (11, dict(n_local=106, n_fixed=2, n_points=4000)),
                                     (12, dict(n_local=107, n_fixed=2, n_points=4000))])
def test_gpu_lba_parity(oracle, seed, kw):
    from vieo_slam_amd.optimizer import Optimizer
    params, kfs, pts, obs, gt = synth_ba.make_lba_problem(seed, **kw)
    on, op, oe, ores = oracle.local_ba(params, kfs, pts, obs)
    hn, hp, he, hres = Optimizer.LocalBundleAdjustment(params, kfs, pts, obs)
    assert hres["status"] == ores["status"] == 0
    for k in range(len(kfs)):
        dt, dr = synth_ba.pose_error(on[k], hn[k])
        assert dt < 1e-4 and dr < 1e-4, (k, dt, dr)
    # points seen only monocularly / at low parallax are weakly constrained along the ray:
    # tiny solver differences are amplified there, the poses (the contract) are not affected
    assert np.abs(op - hp).max() < 5e-2 and np.median(np.abs(op - hp)) < 2e-5
    assert (oe != he).mean() < 0.002  # chi2 gates on values that differ at 1e-9 relative
    assert abs(hres["chi2_final"] - ores["chi2_final"]) < 1e-6 * ores["chi2_final"] + 1e-3
    assert abs(hres["chi2_initial"] - ores["chi2_initial"]) < 1e-6 * ores["chi2_initial"]


@pytest.mark.gpu
def test_gpu_lba_edge_cases(oracle):
    from vieo_slam_amd.optimizer import Optimizer
    params, kfs, pts, obs, gt = synth_ba.make_lba_problem(4, n_local=3, n_fixed=2, n_points=200)
    kfs2 = kfs.copy()
    kfs2["fixed"] = 1
    hn, hp, he, hres = Optimizer.LocalBundleAdjustment(params, kfs2, pts, obs)
    assert hres["status"] == 2 and np.array_equal(hp, pts) and not he.any()
    hn, hp, he, hres = Optimizer.LocalBundleAdjustment(params, kfs, pts, obs, stop=np.array([1], np.int32))
    assert hres["status"] == 1 and hres["lm_iterations"] == 0


@pytest.mark.gpu
def test_gpu_lba_batch_lockstep_matches_oracle(oracle):
    """Windows of different sizes (and one with no free pose) advanced in lock step give, per window,
    what the reference's one-window-at-a-time call gives."""
    from vieo_slam_amd.optimizer import Optimizer
    cfgs = [(10, {}), (11, dict(n_local=3, n_fixed=2, n_points=150, stereo_frac=0.0)),
            (12, dict(n_local=12, n_fixed=5, n_points=1200)), (13, dict(n_local=4, n_fixed=0, n_points=300,
                                                                        first_fixed=True)), (14, {})]
    wins = []
    for seed, kw in cfgs:
        params, kfs, pts, obs, gt = synth_ba.make_lba_problem(seed, **kw)
        wins.append((params, kfs, pts, obs))
    frozen = wins[4][1].copy()
    frozen["fixed"] = 1
    wins[4] = (wins[4][0], frozen, wins[4][2], wins[4][3])
    outs = Optimizer.LocalBundleAdjustmentBatch(wins)
    for w, (params, kfs, pts, obs) in enumerate(wins):
        on, op, oe, ores = oracle.local_ba(params, kfs, pts, obs)
        hn, hp, he, hres = outs[w]
        assert hres["status"] == ores["status"], w
        if ores["status"] == 2:
            assert np.array_equal(hp, pts) and not he.any()
            continue
        for k in range(len(kfs)):
            dt, dr = synth_ba.pose_error(on[k], hn[k])
            assert dt < 1e-4 and dr < 1e-4, (w, k, dt, dr)
        assert np.abs(op - hp).max() < 5e-2 and np.median(np.abs(op - hp)) < 2e-5
        assert (oe != he).mean() < 0.002
        assert abs(hres["chi2_final"] - ores["chi2_final"]) < 1e-6 * ores["chi2_final"] + 1e-3
        assert abs(hres["chi2_initial"] - ores["chi2_initial"]) < 1e-6 * ores["chi2_initial"]
        assert hres["lm_trials"] == ores["lm_trials"], w


# ---- a20: Radtan / KB8 camera models, several cameras per key frame -------------------------------
@pytest.mark.parametrize("rig", ["radtan", "kb8"])
def test_oracle_camera_models_match_closed_form_and_differences(oracle, rig):
    """Project() of the distorted models against an independent float64 restatement, and its Jacobian
    against central differences (camera_radtan.h:61-129, camera_kb8.h:68-157)."""
    cams, (w, h) = synth_ba.camera_rig(rig)
    rng = np.random.default_rng(5)
    for cam in cams:
        for _ in range(50):
            z = rng.uniform(0.5, 10)
            P = np.array([rng.uniform(-0.7, 0.7) * z, rng.uniform(-0.5, 0.5) * z, z])
            uv, J = oracle.cam_project(cam, P)
            ref = synth_ba.project_camera(cam, P)
            assert abs(uv[0] - ref[0]) < 2e-4 and abs(uv[1] - ref[1]) < 2e-4  # float image point
            if rig == "radtan":
                # The reference's Radtan Jacobian is NOT the derivative of its projection: the loop that
                # accumulates d(fd)/d(r2) starts at i = 2 (camera_radtan.h:86-91), so with the usual two
                # radial coefficients the 2 x^2 d(fd)/d(r2) term is missing.  Parity means reproducing it:
                # check against the same expression written independently.
                fx, fy = float(cam["fx"]), float(cam["fy"])
                k1, k2, p1, p2 = [float(v) for v in cam["dist"][:4]]
                x, y = P[0] / z, P[1] / z
                r2 = x * x + y * y
                fd = 1 + k1 * r2 + k2 * r2 * r2
                du_dx = fx / z * (fd + 2 * (p1 * y + 3 * p2 * x))
                du_dy = fx / z * (2 * (p1 * x + p2 * y))
                dv_dy = fy / z * (fd + 2 * (p2 * x + 3 * p1 * y))
                dv_dx = du_dy * fy / fx
                ref_J = np.array([[du_dx, du_dy, -(x * du_dx + y * du_dy)], [dv_dx, dv_dy, -(x * dv_dx + y * dv_dy)]])
                assert np.allclose(J, ref_J, rtol=1e-9, atol=1e-9)
                continue
            for a in range(3):
                d = np.zeros(3)
                d[a] = 1e-6 * z
                num = (np.array(synth_ba.project_camera(cam, P + d)) - np.array(synth_ba.project_camera(cam, P - d))) / (2e-6 * z)
                assert np.allclose(num, J[:, a], rtol=2e-5, atol=1e-5), (a, num, J[:, a])
    if rig == "kb8":  # on the optical axis the model degenerates to the pinhole projection
        uv, J = oracle.cam_project(cams[0], np.array([0.0, 0.0, 2.0]))
        assert abs(uv[0] - cams[0]["cx"]) < 1e-4 and abs(J[0, 0] - cams[0]["fx"] / 2.0) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("rig,seed", [("radtan", 60), ("kb8", 61)])
def test_gpu_lba_distorted_rig_parity(oracle, rig, seed):
    from vieo_slam_amd.optimizer import Optimizer
    params, kfs, pts, obs, gt = synth_ba.make_lba_problem(seed, n_local=6, n_fixed=3, n_points=500, rig=rig)
    assert len(np.unique(obs["kf"] >> 24)) == len(gt["cams"])  # every camera contributes edges
    on, op, oe, ores = oracle.local_ba(params, kfs, pts, obs)
    hn, hp, he, hres = Optimizer.LocalBundleAdjustment(params, kfs, pts, obs)
    assert hres["status"] == ores["status"] == 0
    for k in range(len(kfs)):
        dt, dr = synth_ba.pose_error(on[k], hn[k])
        assert dt < 1e-4 and dr < 1e-4, (k, dt, dr)
    assert np.abs(op - hp).max() < 5e-2 and np.median(np.abs(op - hp)) < 5e-5
    assert (oe != he).mean() < 0.003
    assert abs(hres["chi2_final"] - ores["chi2_final"]) < 1e-6 * ores["chi2_final"] + 1e-3


def _enc_of(seed, kfs, gt, **kw):
    nl = int((kfs["fixed"] == 0).sum())
    return synth_ba.make_lba_enc(seed, gt, synth_ba.lba_enc_pairs(nl, len(kfs)), **kw)


def test_oracle_lba_encoder_edges(oracle):
    """a17: EdgeEncNavStatePR between consecutive key frames (Optimizer.cc:2008-2042).  An empty edge list changes
    nothing; with few, noisy, monocular-heavy observations the wheel odometry tightens the window; a key frame that
    lost every visual edge stays in the system through its encoder edges."""
    params, kfs, pts, obs, gt = synth_ba.make_lba_problem(20, n_local=6, n_fixed=3, n_points=120, noise=2.5,
                                                          stereo_frac=0.2, outlier_frac=0.0)
    a = oracle.local_ba(params, kfs, pts, obs)
    enc0, e0 = synth_ba.make_lba_enc(20, gt, [])
    b = oracle.local_ba(params, kfs, pts, obs, enc=enc0)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    enc, edges = _enc_of(20, kfs, gt)
    assert len(edges) == 6 and edges[0]["kf_i"] == len(kfs) - 1  # the key frame before the window is a fixed one
    c = oracle.local_ba(params, kfs, pts, obs, enc=enc)
    ea, ec = _pose_errs(a[0], gt, 6), _pose_errs(c[0], gt, 6)
    assert ec[:, 0].mean() < 0.7 * ea[:, 0].mean() and ec[:, 1].mean() < 0.7 * ea[:, 1].mean(), (ea.mean(0), ec.mean(0))
    assert c[3]["chi2_initial"] > a[3]["chi2_initial"]  # the pair edges count in the robust chi2
    # a local key frame without observations: only the encoder edges move it (without them it stays put)
    params, kfs, pts, obs, gt = synth_ba.make_lba_problem(21, n_local=6, n_fixed=3, n_points=300, pert_t=0.05,
                                                          pert_r_deg=2.0)
    enc, edges = _enc_of(21, kfs, gt)
    lone = 2
    obs2 = obs[(obs["kf"] & 0xFFFFFF) != lone]
    d0 = oracle.local_ba(params, kfs, pts, obs2)
    assert np.array_equal(d0[0][lone]["p"], kfs[lone]["nav"]["p"])
    d1 = oracle.local_ba(params, kfs, pts, obs2, enc=enc)
    e_before = synth_ba.pose_error(kfs[lone]["nav"], dict(p=gt["p"][lone], q=gt["q"][lone]))
    e_after = synth_ba.pose_error(d1[0][lone], dict(p=gt["p"][lone], q=gt["q"][lone]))
    assert e_after[0] < 0.5 * e_before[0] and e_after[1] < 0.5 * e_before[1], (e_before, e_after)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,kw", [(30, {}), (31, dict(n_local=6, n_fixed=3, n_points=120, noise=2.5, stereo_frac=0.2)),
                                     (32, dict(n_local=25, n_fixed=10, n_points=2500)),
                                     (33, dict(n_local=4, n_fixed=0, n_points=300, first_fixed=True))])
def test_gpu_lba_encoder_edges_parity(oracle, seed, kw):
    from vieo_slam_amd.optimizer import Optimizer
    params, kfs, pts, obs, gt = synth_ba.make_lba_problem(seed, **kw)
    enc, edges = _enc_of(seed, kfs, gt)
    on, op, oe, ores = oracle.local_ba(params, kfs, pts, obs, enc=enc)
    hn, hp, he, hres = Optimizer.LocalBundleAdjustment(params, kfs, pts, obs, enc=enc)
    assert hres["status"] == ores["status"] == 0
    for k in range(len(kfs)):
        dt, dr = synth_ba.pose_error(on[k], hn[k])
        assert dt < 1e-4 and dr < 1e-4, (k, dt, dr)
    assert (oe != he).mean() < 0.002
    assert abs(hres["chi2_final"] - ores["chi2_final"]) < 1e-6 * ores["chi2_final"] + 1e-3
    assert abs(hres["chi2_initial"] - ores["chi2_initial"]) < 1e-6 * ores["chi2_initial"]
    plain = Optimizer.LocalBundleAdjustment(params, kfs, pts, obs)
    assert not np.array_equal(plain[0]["p"], hn["p"])  # the edges are in the system
    none = Optimizer.LocalBundleAdjustment(params, kfs, pts, obs, enc=synth_ba.make_lba_enc(0, gt, [])[0])
    assert np.array_equal(none[0], plain[0]) and np.array_equal(none[2], plain[2])


@pytest.mark.gpu
def test_gpu_lba_encoder_only_key_frame(oracle):
    """A local key frame whose observations are all gone joins the reduced system through its encoder edges."""
    from vieo_slam_amd.optimizer import Optimizer
    params, kfs, pts, obs, gt = synth_ba.make_lba_problem(34, n_local=6, n_fixed=3, n_points=200)
    enc, edges = _enc_of(34, kfs, gt)
    obs2 = obs[(obs["kf"] & 0xFFFFFF) != 2]
    on, op, oe, ores = oracle.local_ba(params, kfs, pts, obs2, enc=enc)
    hn, hp, he, hres = Optimizer.LocalBundleAdjustment(params, kfs, pts, obs2, enc=enc)
    assert not np.array_equal(hn[2]["p"], kfs[2]["nav"]["p"])
    for k in range(len(kfs)):
        dt, dr = synth_ba.pose_error(on[k], hn[k])
        assert dt < 1e-4 and dr < 1e-4, (k, dt, dr)


@pytest.mark.gpu
def test_gpu_lba_batch_with_and_without_encoder_edges(oracle):
    """Lock-step batch in which only some windows carry encoder edges (encs[w] = NULL for the others)."""
    from vieo_slam_amd.optimizer import Optimizer
    wins, encs, keep = [], [], []
    for i in range(5):
        w = synth_ba.make_lba_problem(40 + i, n_local=4 + 3 * i, n_fixed=2 + i, n_points=300 + 200 * i)
        wins.append(w[:4])
        e = _enc_of(40 + i, w[1], w[4]) if i % 2 == 0 else None
        keep.append(e)
        encs.append(None if e is None else e[0])
    outs = Optimizer.LocalBundleAdjustmentBatch(wins, encs=encs)
    for w, (win, enc) in enumerate(zip(wins, encs)):
        on, op, oe, ores = oracle.local_ba(*win, enc=enc)
        hn, hp, he, hres = outs[w]
        for k in range(len(win[1])):
            dt, dr = synth_ba.pose_error(on[k], hn[k])
            assert dt < 1e-4 and dr < 1e-4, (w, k, dt, dr)
        assert hres["lm_trials"] == ores["lm_trials"] and (oe != he).mean() < 0.002
