"""Full BA (BASELINE configs[4]): Optimizer::BundleAdjustment (Optimizer.cc:1353-1609) and
GlobalBundleAdjustmentNavStatePRV (:771-1345) on the local-BA engine; reduced systems of hundreds of unknowns
go through the tiled LDL^T.  Oracle known-answer tests (CPU) and GPU parity (<= 1e-4 on SE(3))."""
import numpy as np
import pytest

from vieo_slam_amd import synth_ba

TOL = 1e-4


def _pose_diff(a, b, n):
    dt = np.linalg.norm(a["p"][:n] - b["p"][:n], axis=1).max()
    dr = max(synth_ba.pose_error(a[k], b[k])[1] for k in range(n))
    return dt, dr


def _gt_err(navs, gt, n):
    dp = np.linalg.norm(navs["p"][:n] - gt["p"][:n], axis=1)
    dr = np.array([synth_ba.pose_error(navs[k], dict(p=navs[k]["p"], q=gt["q"][k]))[1] for k in range(n)])
    return dp, dr


# ------------------------------------------------------------------ oracle (CPU)
def test_oracle_vision_gba_noiseless_recovers_truth(oracle):
    P, kfs, pts, obs, gt = synth_ba.make_lba_problem(2, n_local=24, n_fixed=2, n_points=1500, anchors=5,
                                                     outlier_frac=0.0, noise=0.0, stereo_frac=1.0)
    navs, pout, res = oracle.bundle_adjustment(P, kfs, pts, obs, n_iterations=20, robust=False)
    dp, dr = _gt_err(navs, gt, 24)
    assert res["status"] == 0 and res["n_erase"] == 0
    assert dp.max() < 5e-4 and dr.max() < 1e-4, (dp.max(), dr.max())
    assert res["chi2_final"] < 1e-3 * res["chi2_initial"]


def test_oracle_gba_robust_flag_and_iteration_count(oracle):
    P, kfs, pts, obs, gt = synth_ba.make_lba_problem(3, n_local=12, n_fixed=1, n_points=600, anchors=3)
    a = oracle.bundle_adjustment(P, kfs, pts, obs, n_iterations=1, robust=True)
    b = oracle.bundle_adjustment(P, kfs, pts, obs, n_iterations=8, robust=True)
    c = oracle.bundle_adjustment(P, kfs, pts, obs, n_iterations=8, robust=False)
    assert a[2]["lm_iterations"] == 1 and 1 < b[2]["lm_iterations"] <= 8
    assert b[2]["chi2_final"] < a[2]["chi2_final"]
    # 3 % gross outliers: the Huber cost is far below the squared one, and the robust poses are closer to truth
    assert b[2]["chi2_final"] < 0.5 * c[2]["chi2_final"]
    assert _gt_err(b[0], gt, 12)[0].mean() < _gt_err(c[0], gt, 12)[0].mean()
    # its0 / its1 of the params play no role
    P2 = P.copy()
    P2[0]["its0"], P2[0]["its1"] = 1, 0
    d = oracle.bundle_adjustment(P2, kfs, pts, obs, n_iterations=8, robust=True)
    assert np.array_equal(d[0]["p"], b[0]["p"])


def test_oracle_vio_gba_noiseless_recovers_truth(oracle):
    params, kfs, pts, close, obs, imu, gt = synth_ba.make_lba_vio_problem(
        5, n_local=20, n_fixed=1, n_points=1200, anchors=4, outlier_frac=0.0, noise=0.0, stereo_frac=1.0,
        imu_noise=0.0)
    navs, pout, res = oracle.global_ba_vio(params, kfs, pts, obs, imu, n_iterations=20, robust=False)
    dp, dr = _gt_err(navs, gt, 20)
    assert res["status"] == 0
    assert dp.max() < 1e-3 and dr.max() < 2e-4, (dp.max(), dr.max())


# ------------------------------------------------------------------ parity (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_local,n_points,robust,iters", [
    (11, 30, 2500, True, 10),     # 180 unknowns: single-workgroup LDL^T
    (12, 100, 6000, True, 6),     # 600 unknowns: tiled LDL^T
    (13, 100, 6000, False, 6),
])
def test_vision_gba_parity(oracle, seed, n_local, n_points, robust, iters):
    from vieo_slam_amd.optimizer import Optimizer
    P, kfs, pts, obs, gt = synth_ba.make_lba_problem(seed, n_local=n_local, n_fixed=1, n_points=n_points,
                                                     anchors=max(2, n_local // 6))
    on, op, ores = oracle.bundle_adjustment(P, kfs, pts, obs, iters, robust)
    hn, hp, hres = Optimizer.BundleAdjustment(P, kfs, pts, obs, iters, robust)
    dt, dr = _pose_diff(on, hn, n_local)
    assert dt < TOL and dr < TOL, (dt, dr)
    assert np.abs(op - hp).max() < 1e-3
    assert hres["status"] == ores["status"] == 0 and hres["n_erase"] == 0
    assert abs(hres["chi2_final"] - ores["chi2_final"]) <= 1e-6 * ores["chi2_final"]
    e0, e1 = _gt_err(kfs["nav"], gt, n_local)[0].mean(), _gt_err(hn, gt, n_local)[0].mean()
    assert e1 < e0 or not robust  # without the kernel the 3 % gross outliers drag the poses


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_local,n_points,robust,iters", [
    (21, 12, 1500, True, 8),      # 180 unknowns
    (22, 50, 5000, True, 5),      # 750 unknowns: tiled LDL^T
    (23, 50, 5000, False, 5),
])
def test_vio_gba_parity(oracle, seed, n_local, n_points, robust, iters):
    from vieo_slam_amd.optimizer import Optimizer
    params, kfs, pts, close, obs, imu, gt = synth_ba.make_lba_vio_problem(
        seed, n_local=n_local, n_fixed=1, n_points=n_points, anchors=max(2, n_local // 6))
    on, op, ores = oracle.global_ba_vio(params, kfs, pts, obs, imu, iters, robust)
    hn, hp, hres = Optimizer.GlobalBundleAdjustmentNavStatePRV(params, kfs, pts, obs, imu, iters, robust)
    dt, dr = _pose_diff(on, hn, n_local)
    assert dt < TOL and dr < TOL, (dt, dr)
    assert np.linalg.norm(on["v"][:n_local] - hn["v"][:n_local], axis=1).max() < 1e-4
    assert np.abs(on["dbg"][:n_local] - hn["dbg"][:n_local]).max() < 1e-6
    assert np.abs(on["dba"][:n_local] - hn["dba"][:n_local]).max() < 1e-5
    assert np.abs(op - hp).max() < 1e-3
    assert hres["status"] == ores["status"] == 0


@pytest.mark.gpu
def test_gba_stop_flag_and_no_free_pose(oracle):
    from vieo_slam_amd.optimizer import Optimizer
    P, kfs, pts, obs, gt = synth_ba.make_lba_problem(31, n_local=8, n_fixed=1, n_points=400)
    hn, hp, hres = Optimizer.BundleAdjustment(P, kfs, pts, obs, 5, True, stop=np.ones(1, np.int32))
    assert hres["status"] == 1 and np.array_equal(hn["p"], kfs["nav"]["p"]) and np.array_equal(hp, pts)
    k2 = kfs.copy()
    k2["fixed"] = 1
    hn, hp, hres = Optimizer.BundleAdjustment(P, k2, pts, obs, 5, True)
    assert hres["status"] == 2


@pytest.mark.gpu
def test_vio_gba_landmark_sharded_two_ranks_on_one_gpu(oracle):
    """BASELINE configs[4] in small: the full BA of 44 key frames (660 unknowns, tiled LDL^T) with its landmarks
    split over two 'ranks' (threads with their own reduction buffers; the callback sums through the host)."""
    import threading
    from vieo_slam_amd import sharding
    from vieo_slam_amd._lib import DeviceBuffer, check, lib
    from vieo_slam_amd.optimizer import Optimizer
    params, kfs, pts, close, obs, imu, gt = synth_ba.make_lba_vio_problem(41, n_local=44, n_fixed=1, n_points=4000,
                                                                          anchors=22, span=5)
    win = (params, kfs, pts, close, obs, imu)
    on, op, ores = oracle.global_ba_vio(params, kfs, pts, obs, imu, 4, True)
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
                                                                           make_cb(rank), 4, True)
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert all(r is not None for r in results)
    for rank in range(world):
        hn, hp, hr = results[rank]
        assert hr["status"] == 0 and hr["lm_trials"] == ores["lm_trials"]
        dt, dr = _pose_diff(on, hn, 44)
        assert dt < TOL and dr < TOL, (dt, dr)
        assert np.abs(op[shards[rank][1]] - hp).max() < 1e-3
    assert results[0][0].tobytes() == results[1][0].tobytes()  # replicated solve: bit-identical key frames


@pytest.mark.gpu
def test_vio_gba_in_library_rccl_single_rank():
    """BASELINE configs[4]'s exchange step through the dlopen'ed RCCL (vieo_rccl_*): a one-rank communicator, the
    all-reduce of the reduced visual system issued by the library on the BA stream; equal to the unsharded call."""
    from vieo_slam_amd import sharding
    from vieo_slam_amd._lib import DeviceBuffer, lib
    from vieo_slam_amd.optimizer import Optimizer
    assert lib().vieo_rccl_available() == 1
    params, kfs, pts, close, obs, imu, gt = synth_ba.make_lba_vio_problem(43, n_local=36, n_fixed=1, n_points=3000,
                                                                          anchors=18, span=5)
    win = (params, kfs, pts, close, obs, imu)
    comm = sharding.RcclComm(0, 1)
    try:
        n = Optimizer.sharded_buffer_doubles([win])
        buf = DeviceBuffer(8 * n)
        sn, sp, sr = Optimizer.GlobalBundleAdjustmentNavStatePRVSharded(win, buf.ptr, n, None, 4, True, comm=comm.handle)
        pn, pp, pr = Optimizer.GlobalBundleAdjustmentNavStatePRV(params, kfs, pts, obs, imu, 4, True)
        assert sr["status"] == 0 and sr["lm_trials"] == pr["lm_trials"]
        assert sn.tobytes() == pn.tobytes() and np.array_equal(sp, pp)
    finally:
        comm.close()


def test_oracle_vision_gba_encoder_edges(oracle):
    """BundleAdjustment(bEnc = true) (Optimizer.cc:1401-1438): noiseless odometry between every pair leaves the
    noiseless optimum where it is; the kernel is on the pair edges iff bRobust."""
    P, kfs, pts, obs, gt = synth_ba.make_lba_problem(5, n_local=12, n_fixed=1, n_points=800, outlier_frac=0.0, noise=0.0,
                                                     stereo_frac=1.0, anchors=3)
    enc, edges = synth_ba.make_lba_enc(5, gt, synth_ba.lba_enc_pairs(12, 13), noise=0.0)
    assert len(edges) == 12
    navs, pout, res = oracle.bundle_adjustment(P, kfs, pts, obs, 10, True, enc=enc)
    dt, dr = _gt_err(navs, gt, 12)
    assert dt.max() < 5e-4 and dr.max() < 1e-4
    # a gross odometry error on one pair: with the kernel its pull is bounded
    edges[5]["enc"]["delx"][3:] += 0.5
    r1 = oracle.bundle_adjustment(P, kfs, pts, obs, 10, True, enc=enc)
    r0 = oracle.bundle_adjustment(P, kfs, pts, obs, 10, False, enc=enc)
    assert _gt_err(r1[0], gt, 12)[0].max() < _gt_err(r0[0], gt, 12)[0].max()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_local,n_points,robust,iters", [(21, 30, 2500, True, 8), (22, 100, 6000, False, 5)])
def test_vision_gba_encoder_edges_parity(oracle, seed, n_local, n_points, robust, iters):
    from vieo_slam_amd.optimizer import Optimizer
    P, kfs, pts, obs, gt = synth_ba.make_lba_problem(seed, n_local=n_local, n_fixed=1, n_points=n_points,
                                                     anchors=max(2, n_local // 6))
    enc, edges = synth_ba.make_lba_enc(seed, gt, synth_ba.lba_enc_pairs(n_local, n_local + 1))
    on, op, ores = oracle.bundle_adjustment(P, kfs, pts, obs, iters, robust, enc=enc)
    hn, hp, hres = Optimizer.BundleAdjustment(P, kfs, pts, obs, iters, robust, enc=enc)
    dt, dr = _pose_diff(on, hn, n_local)
    assert dt < TOL and dr < TOL, (dt, dr)
    assert hres["status"] == ores["status"] == 0
    assert abs(hres["chi2_final"] - ores["chi2_final"]) <= 1e-6 * ores["chi2_final"]
    plain = Optimizer.BundleAdjustment(P, kfs, pts, obs, iters, robust)
    assert not np.array_equal(plain[0]["p"], hn["p"])
