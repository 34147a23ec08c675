"""LocalBundleAdjustmentNavStatePRV timing: single window vs lock-step batches (run on the GPU box)."""
import sys, time
sys.path.insert(0, ".")
import torch  # noqa: F401
from vieo_slam_amd import synth_ba
from vieo_slam_amd.optimizer import Optimizer

wins = [synth_ba.make_lba_vio_problem(100 + s, n_local=10, n_fixed=5, n_points=1000)[:6] for s in range(64)]
Optimizer.LocalBundleAdjustmentNavStatePRV(*wins[0])
t = time.time()
for i in range(8):
    r = Optimizer.LocalBundleAdjustmentNavStatePRV(*wins[i])
dt = (time.time() - t) / 8
print("single window: %.2f ms  (trials %d, n_obs %d)" % (dt * 1e3, r[3]["lm_trials"], len(wins[0][4])))
for W in (1, 4, 16, 64):
    Optimizer.LocalBundleAdjustmentNavStatePRVBatch(wins[:W])
    t = time.time()
    n = 3
    for i in range(n):
        Optimizer.LocalBundleAdjustmentNavStatePRVBatch(wins[:W])
    dt = (time.time() - t) / n
    print("batch W=%2d: %.2f ms per call, %.3f ms per window, %.0f windows/s" % (W, dt * 1e3, dt * 1e3 / W, W / dt))
