"""Host phases of one lock-step LocalBundleAdjustmentNavStatePRV batch (VIEO_LBA_TIMING=1 prints them)."""
import os, sys, time
os.environ["VIEO_LBA_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vieo_slam_amd import synth_ba
from vieo_slam_amd.optimizer import Optimizer
probs = [synth_ba.make_lba_vio_problem(500 + i, n_local=10, n_fixed=6, n_points=2000)[:6] for i in range(8)]
for N in (1, 103):
    wins = [probs[i % 8] for i in range(N)]
    Optimizer.LocalBundleAdjustmentNavStatePRVBatch(wins)
    t = time.perf_counter()
    for _ in range(3):
        r = Optimizer.LocalBundleAdjustmentNavStatePRVBatch(wins)
    print("%d windows: %.2f ms per call; trials %s" % (N, (time.perf_counter() - t) / 3 * 1e3, [int(x[3]["lm_trials"]) for x in r[:8]]))
