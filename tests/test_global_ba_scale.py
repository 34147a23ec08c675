"""System::FinalGBA's full BA (BASELINE configs[4]): GlobalBundleAdjustmentNavStatePRV with bScaleOpt = true
(src/System.cc:24-33, src/Optimizer.cc:842-851,1131-1137,1190-1196,1256-1335): VertexScale (g2otypes.h:292-311) and
the three-vertex EdgeReprojectPRS / PRSStereo (g2otypes.h:321-541, MODE_OPT_VAR == 1, typedefs :548-550).
Oracle known-answer tests (CPU) and GPU parity (<= 1e-4 on SE(3), recovered scale included)."""
import numpy as np
import pytest

from vieo_slam_amd import synth_ba
from vieo_slam_amd.ba_types import LBA_OBS_DTYPE, NAVSTATE_DTYPE

TOL = 1e-4


def _pose_diff(a, b, n):
    dt = np.linalg.norm(a["p"][:n] - b["p"][:n], axis=1).max()
    dr = max(synth_ba.pose_error(a[k], b[k])[1] for k in range(n))
    return dt, dr


def _problem(seed, n_local=12, n_points=900, s0=1.0, **kw):
    """A visual-inertial full-BA problem whose map points are handed over divided by s0: the PRS edges see s * Xh, so
    the optimum has s * Xh = the true points (s and Xh share a gauge; their product is what the edges constrain)."""
    params, kfs, pts, close, obs, imu, gt = synth_ba.make_lba_vio_problem(seed, n_local=n_local, n_fixed=1,
                                                                          n_points=n_points,
                                                                          anchors=max(2, n_local // 6), **kw)
    return params, kfs, (pts / np.float32(s0)).astype(np.float32), obs, imu, gt


# ------------------------------------------------------------------ oracle (CPU)
@pytest.mark.parametrize("stereo", [False, True])
def test_oracle_prs_edge_jacobians_match_central_differences(oracle, stereo):
    """J_PR through the vertex's own retraction (p <- p + R dp, R <- R Exp(dphi)), J_Xh and J_s against central
    differences of the error, at a scale away from 1."""
    params, kfs, pts, close, obs, imu, gt = synth_ba.make_lba_vio_problem(3, n_local=3, n_fixed=1, n_points=60)
    rng = np.random.default_rng(5)
    ns = kfs["nav"][0].copy()
    checked = 0
    for o in obs[obs["kf"] == 0]:
        if (o["ur"] >= 0) != stereo:
            continue
        s = 1.0 + rng.uniform(-0.2, 0.2)
        Xh = pts[o["mp"]].astype(np.float64) / s + rng.normal(0, 0.01, 3)
        err, Jp, Jx, Js = oracle.lba_prs_edge_eval(params, ns, Xh, s, o)
        de = 3 if stereo else 2
        assert np.all(err[de:] == 0)
        h = 1e-3  # the projection is rounded to float (camera_base Project returns float pixels): 3e-5 px steps
        for k in range(6):
            d = np.zeros(15)
            d[k] = h
            ep = oracle.lba_prs_edge_eval(params, oracle.lba_navstate_inc(ns, d), Xh, s, o, jac=False)[0]
            em = oracle.lba_prs_edge_eval(params, oracle.lba_navstate_inc(ns, -d), Xh, s, o, jac=False)[0]
            assert np.allclose((ep - em)[:de] / (2 * h), Jp[:de, k], rtol=1e-3, atol=5e-2), (k, Jp[:de, k])
        for k in range(3):
            d = np.zeros(3)
            d[k] = h
            ep = oracle.lba_prs_edge_eval(params, ns, Xh + d, s, o, jac=False)[0]
            em = oracle.lba_prs_edge_eval(params, ns, Xh - d, s, o, jac=False)[0]
            assert np.allclose((ep - em)[:de] / (2 * h), Jx[:de, k], rtol=1e-3, atol=5e-2)
        ep = oracle.lba_prs_edge_eval(params, ns, Xh, s + h, o, jac=False)[0]
        em = oracle.lba_prs_edge_eval(params, ns, Xh, s - h, o, jac=False)[0]
        assert np.allclose((ep - em)[:de] / (2 * h), Js[:de], rtol=1e-3, atol=5e-2)
        # chain rule of the three-vertex edge: J_Xh = s * (Jproj Rcw), J_s = (Jproj Rcw) Xh
        assert np.allclose(Jx[:de] @ Xh / s, Js[:de], rtol=1e-12, atol=1e-12)
        checked += 1
        if checked == 8:
            break
    assert checked >= 4


def test_oracle_scale_one_edge_equals_plain_edge(oracle):
    """At s = 1 the PRS edge is the PR edge: same error and Jacobians as the local BA's linearisation, which the
    scale-free full BA keeps using."""
    params, kfs, pts, obs, imu, gt = _problem(4, n_local=4, n_points=200)
    a = oracle.global_ba_vio(params, kfs, pts, obs, imu, 1, True)
    b = oracle.global_ba_vio(params, kfs, pts, obs, imu, 1, True, scale_opt=False)
    assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]) and b[3] == 1.0


def test_oracle_scale_gba_recovers_scaled_map(oracle):
    """Noiseless scene, map handed over 8 % too small: the inertial edges pin the metric trajectory, the scale vertex
    (with the points) absorbs the factor -- s * Xh comes back as the true points and the key frames stay on truth."""
    s0 = 1.08
    params, kfs, pts, obs, imu, gt = _problem(6, n_local=16, n_points=1000, s0=s0, outlier_frac=0.0, noise=0.0,
                                              stereo_frac=0.5, imu_noise=0.0, pert_x=0.0)
    navs, pout, res, scale = oracle.global_ba_vio(params, kfs, pts, obs, imu, 30, False, scale_opt=True)
    assert res["status"] == 0 and res["chi2_final"] < 1e-6 * res["chi2_initial"]
    dp = np.linalg.norm(navs["p"][:16] - gt["p"][:16], axis=1)
    assert dp.max() < 1e-3, dp.max()
    assert np.abs(pout - gt["X"]).max() < 5e-3
    assert scale != 1.0 and abs(scale - 1.0) > 1e-3  # the vertex moved (how the factor splits between s and Xh is gauge)
    # without the scale vertex the same start needs the points alone to move: same optimum for s * Xh
    n2, p2, r2 = oracle.global_ba_vio(params, kfs, pts, obs, imu, 30, False)
    assert np.abs(p2 - gt["X"]).max() < 5e-3


def test_oracle_scale_gba_only_scale_free(oracle):
    """bdimPoses is true with the scale vertex even when every key frame is fixed (Optimizer.cc:850): the optimiser still
    runs over scale + points."""
    params, kfs, pts, obs, imu, gt = _problem(7, n_local=5, n_points=300, s0=1.05, outlier_frac=0.0, noise=0.0,
                                              imu_noise=0.0, pert_t=0.0, pert_r_deg=0.0, pert_v=0.0, pert_x=0.0)
    k2 = kfs.copy()
    k2["fixed"] = 1
    navs, pout, res, scale = oracle.global_ba_vio(params, k2, pts, obs, imu, 20, False, scale_opt=True)
    assert res["status"] == 0 and res["lm_iterations"] >= 1
    assert np.abs(pout - gt["X"]).max() < 1e-3
    assert navs["p"].tobytes() == kfs["nav"]["p"].tobytes()
    plain = oracle.global_ba_vio(params, k2, pts, obs, imu, 20, False)
    assert plain[2]["status"] == 2  # VIEO_LBA_NO_FREE_POSE without it


def test_oracle_scale_write_back_is_float_product(oracle):
    """SetWorldPos(scale * vPoint->estimate().cast<float>()) (Optimizer.cc:1321): the double scale meets a float vector, so
    the product is taken in float."""
    params, kfs, pts, obs, imu, gt = _problem(8, n_local=6, n_points=300, s0=1.03)
    navs, pout, res, scale = oracle.global_ba_vio(params, kfs, pts, obs, imu, 3, True, scale_opt=True)
    assert pout.dtype == np.float32 and res["lm_trials"] >= 3
    # the points' own move is small next to the factor: p_out / p_in ~ scale * (1 + small)
    r = np.median(pout[:, 2] / pts[:, 2])
    assert abs(r - scale) < 2e-2


# ------------------------------------------------------------------ parity (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_local,n_points,robust,iters,s0", [
    (51, 8, 900, True, 8, 1.05),      # 121 unknowns: blocked LDL^T in one workgroup
    (52, 10, 1200, False, 6, 0.95),   # 151 unknowns
    (53, 30, 3000, True, 6, 1.04),    # 451 unknowns: not divisible by the column-panel width -> tiled LDL^T
    (54, 50, 5000, True, 5, 1.02),    # 751 unknowns: tiled LDL^T, block-sparse BB with the dense scale row
])
def test_vio_gba_scale_parity(oracle, seed, n_local, n_points, robust, iters, s0):
    from vieo_slam_amd.optimizer import Optimizer
    params, kfs, pts, obs, imu, gt = _problem(seed, n_local=n_local, n_points=n_points, s0=s0)
    on, op, ores, osc = oracle.global_ba_vio(params, kfs, pts, obs, imu, iters, robust, scale_opt=True)
    hn, hp, hres, hsc = Optimizer.GlobalBundleAdjustmentNavStatePRV(params, kfs, pts, obs, imu, iters, robust,
                                                                    bScaleOpt=True)
    dt, dr = _pose_diff(on, hn, n_local)
    assert dt < TOL and dr < TOL, (dt, dr)
    assert abs(osc - hsc) < 1e-6 and abs(osc - 1.0) > 1e-4, (osc, hsc)
    assert np.linalg.norm(on["v"][:n_local] - hn["v"][:n_local], axis=1).max() < 1e-4
    assert np.abs(on["dbg"][:n_local] - hn["dbg"][:n_local]).max() < 1e-6
    assert np.abs(on["dba"][:n_local] - hn["dba"][:n_local]).max() < 1e-5
    assert np.abs(op - hp).max() < 1e-3
    assert hres["status"] == ores["status"] == 0 and hres["lm_trials"] == ores["lm_trials"]
    assert abs(hres["chi2_final"] - ores["chi2_final"]) <= 1e-6 * ores["chi2_final"]
    # the scale vertex changes the answer (the test is not vacuous)
    plain = Optimizer.GlobalBundleAdjustmentNavStatePRV(params, kfs, pts, obs, imu, iters, robust)
    assert not np.array_equal(plain[0]["p"], hn["p"])


@pytest.mark.gpu
def test_vio_gba_scale_off_is_the_plain_call(oracle):
    from vieo_slam_amd.optimizer import Optimizer
    params, kfs, pts, obs, imu, gt = _problem(55, n_local=12, n_points=1000)
    a = Optimizer.GlobalBundleAdjustmentNavStatePRV(params, kfs, pts, obs, imu, 5, True)
    b = Optimizer.GlobalBundleAdjustmentNavStatePRV(params, kfs, pts, obs, imu, 5, True, bScaleOpt=False)
    assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]) and b[3] == 1.0


@pytest.mark.gpu
def test_vio_gba_scale_rig_parity(oracle):
    """Distorted four-camera key frames (configs[4] is a dStereo rig): the PRS edges through the KB8 model."""
    from vieo_slam_amd.optimizer import Optimizer
    params, kfs, pts, close, obs, imu, gt = synth_ba.make_lba_vio_problem(56, n_local=10, n_fixed=1, n_points=900,
                                                                          anchors=3, rig="kb8")
    pts = (pts / np.float32(1.03)).astype(np.float32)
    on, op, ores, osc = oracle.global_ba_vio(params, kfs, pts, obs, imu, 6, True, scale_opt=True)
    hn, hp, hres, hsc = Optimizer.GlobalBundleAdjustmentNavStatePRV(params, kfs, pts, obs, imu, 6, True, bScaleOpt=True)
    dt, dr = _pose_diff(on, hn, 10)
    assert dt < TOL and dr < TOL, (dt, dr)
    assert abs(osc - hsc) < 1e-6 and np.abs(op - hp).max() < 1e-3
    assert hres["lm_trials"] == ores["lm_trials"]


@pytest.mark.gpu
def test_vio_gba_scale_all_key_frames_fixed(oracle):
    """bdimPoses stays true with the scale vertex (Optimizer.cc:850): scale + points are optimised against fixed key
    frames (a reduced system of ONE unknown)."""
    from vieo_slam_amd.optimizer import Optimizer
    params, kfs, pts, obs, imu, gt = _problem(57, n_local=5, n_points=300, s0=1.05)
    k2 = kfs.copy()
    k2["fixed"] = 1
    on, op, ores, osc = oracle.global_ba_vio(params, k2, pts, obs, imu, 8, True, scale_opt=True)
    hn, hp, hres, hsc = Optimizer.GlobalBundleAdjustmentNavStatePRV(params, k2, pts, obs, imu, 8, True, bScaleOpt=True)
    assert hres["status"] == ores["status"] == 0 and hres["lm_trials"] == ores["lm_trials"]
    assert abs(osc - hsc) < 1e-6 and np.abs(op - hp).max() < 1e-3
    assert hn["p"].tobytes() == k2["nav"]["p"].tobytes()


@pytest.mark.gpu
def test_vio_gba_scale_landmark_sharded_two_ranks_on_one_gpu(oracle):
    """BASELINE configs[4] as System::FinalGBA runs it, landmarks split over two 'ranks' (threads with their own
    reduction buffers; the callback sums through the host): the scale's row of the reduced system and its H_ps / H_ss /
    b_s travel in the same exchange."""
    import threading
    from vieo_slam_amd import sharding
    from vieo_slam_amd._lib import DeviceBuffer, check, lib
    from vieo_slam_amd.optimizer import Optimizer
    params, kfs, pts, close, obs, imu, gt = synth_ba.make_lba_vio_problem(58, n_local=44, n_fixed=1, n_points=4000,
                                                                          anchors=22, span=5)
    pts = (pts / np.float32(1.04)).astype(np.float32)
    win = (params, kfs, pts, close, obs, imu)
    on, op, ores, osc = oracle.global_ba_vio(params, kfs, pts, obs, imu, 4, True, scale_opt=True)
    world = 2
    shards = [sharding.shard_window(win, r, world) for r in range(world)]
    n = Optimizer.sharded_buffer_doubles([win])
    bufs = [DeviceBuffer(8 * n) for _ in range(world)]
    barrier = threading.Barrier(world)
    stage, results = [None] * world, [None] * world

    def make_cb(rank):
        def cb(offset, count):
            h = np.empty(count)
            check(lib().vieo_memcpy_d2h(h.ctypes.data, bufs[rank].ptr + 8 * offset, 8 * count))
            stage[rank] = h
            barrier.wait()
            total = stage[0] + stage[1]
            barrier.wait()
            check(lib().vieo_memcpy_h2d(bufs[rank].ptr + 8 * offset, total.ctypes.data, 8 * count))
            return 0
        return cb

    def run(rank):
        results[rank] = Optimizer.GlobalBundleAdjustmentNavStatePRVSharded(shards[rank][0], bufs[rank].ptr, n,
                                                                           make_cb(rank), 4, True, bScaleOpt=True)
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert all(r is not None for r in results)
    for rank in range(world):
        hn, hp, hr, hsc = results[rank]
        assert hr["status"] == 0 and hr["lm_trials"] == ores["lm_trials"]
        dt, dr = _pose_diff(on, hn, 44)
        assert dt < TOL and dr < TOL, (dt, dr)
        assert abs(hsc - osc) < 1e-6
        assert np.abs(op[shards[rank][1]] - hp).max() < 1e-3
    assert results[0][0].tobytes() == results[1][0].tobytes() and results[0][3] == results[1][3]


@pytest.mark.gpu
def test_vio_gba_scale_bench_problem_parity(oracle):
    """bench.py's `sharded_full_ba` problem (60 key frames, 6000 points before culling, System::FinalGBA's form with the
    scale vertex, map handed over at scale 1 / 1.03) UNSHARDED against the oracle -- the bench compares the sharded run
    with the unsharded HIP call only."""
    from vieo_slam_amd.optimizer import Optimizer
    win = synth_ba.make_lba_vio_problem(901, n_local=60, n_fixed=1, n_points=6000, anchors=30, span=5)[:6]
    params, kfs, pts, obs, imu = win[0], win[1], (win[2] / np.float32(1.03)).astype(np.float32), win[4], win[5]
    on, op, ores, osc = oracle.global_ba_vio(params, kfs, pts, obs, imu, 5, True, scale_opt=True)
    hn, hp, hres, hsc = Optimizer.GlobalBundleAdjustmentNavStatePRV(params, kfs, pts, obs, imu, 5, True, bScaleOpt=True)
    dt, dr = _pose_diff(on, hn, 60)
    assert dt < TOL and dr < TOL, (dt, dr)
    assert abs(osc - hsc) < 1e-6 and abs(hsc - 1.03) < 5e-3, (osc, hsc)
    assert hres["status"] == ores["status"] == 0 and hres["lm_trials"] == ores["lm_trials"]
    assert np.abs(op - hp).max() < 1e-3


@pytest.mark.gpu
def test_vio_gba_scale_400_key_frames_parity(oracle):
    """Full BA at BASELINE configs[4] scale -- 400 key frames = 6001 unknowns of the reduced system (tiled LDL^T,
    block-sparse Schur with the dense scale row) -- against the oracle on a thinned point set (4000 points, ~19 k
    observations; the oracle's dense solve is what takes the time: two iterations, about a minute of CPU)."""
    from vieo_slam_amd.optimizer import Optimizer
    params, kfs, pts, close, obs, imu, gt = synth_ba.make_lba_vio_problem(7, n_local=400, n_fixed=1, n_points=4000,
                                                                          anchors=200, span=5)
    pts = (pts / np.float32(1.02)).astype(np.float32)
    on, op, ores, osc = oracle.global_ba_vio(params, kfs, pts, obs, imu, 2, True, scale_opt=True)
    hn, hp, hres, hsc = Optimizer.GlobalBundleAdjustmentNavStatePRV(params, kfs, pts, obs, imu, 2, True, bScaleOpt=True)
    dt, dr = _pose_diff(on, hn, 400)
    assert dt < TOL and dr < TOL, (dt, dr)
    assert abs(osc - hsc) < 1e-6, (osc, hsc)
    assert hres["status"] == ores["status"] == 0 and hres["lm_trials"] == ores["lm_trials"]
    assert np.linalg.norm(on["v"][:400] - hn["v"][:400], axis=1).max() < 1e-4
