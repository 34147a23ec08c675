"""Full-BA timing (run on the GPU box): GlobalBundleAdjustmentNavStatePRV / BundleAdjustment on maps of growing
size, the HIP engine against the CPU oracle (single thread, as g2o runs in the reference)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from tests import oracle_lib
from vieo_slam_amd import synth_ba
from vieo_slam_amd.optimizer import Optimizer

orc = oracle_lib.load()
ITERS = 5
for n_local, n_points, cpu in ((50, 5000, True), (100, 10000, True), (200, 20000, True), (400, 40000, False),
                               (800, 80000, False)):
    t0 = time.time()
    params, kfs, pts, close, obs, imu, gt = synth_ba.make_lba_vio_problem(
        7, n_local=n_local, n_fixed=1, n_points=n_points, anchors=n_local // 2, span=5)
    tg = time.time() - t0
    Optimizer.GlobalBundleAdjustmentNavStatePRV(params, kfs, pts, obs, imu, 1, True)  # warm-up (allocations)
    t = time.time()
    hn, hp, hres = Optimizer.GlobalBundleAdjustmentNavStatePRV(params, kfs, pts, obs, imu, ITERS, True)
    dt = time.time() - t
    line = "VIO GBA  %3d KFs (%4d unknowns) %5d points %6d edges: HIP %.1f ms (%d trials)" % (
        n_local, 15 * n_local, len(pts), len(obs), dt * 1e3, hres["lm_trials"])
    if cpu:
        t = time.time()
        on, op, ores = orc.global_ba_vio(params, kfs, pts, obs, imu, ITERS, True)
        dc = time.time() - t
        line += "   oracle %.0f ms (x%.0f)   max |dp| %.1e" % (dc * 1e3, dc / dt, np.abs(on["p"] - hn["p"]).max())
    print(line + "   [gen %.0f s]" % tg, flush=True)
