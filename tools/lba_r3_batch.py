"""The r3 bench step's local-BA share alone: N windows in lock step (every 4th bLarge), kernel-class times.
usage: lba_r3_batch.py <repo root> <windows> [mixed|small|large]"""
import sys, time
sys.path.insert(0, sys.argv[1])
from vieo_slam_amd import synth_ba
from vieo_slam_amd.optimizer import Optimizer
probs = []
for i in range(8):
    large = i % 4 == 3
    w = synth_ba.make_lba_vio_problem(500 + i, n_local=25 if large else 10, n_fixed=40, n_points=2000)[:6]
    w[0][0]["large"] = int(large)
    if large:
        w[0][0]["base"]["its0"], w[0][0]["base"]["its1"] = 2, 2
    probs.append(w)
N = int(sys.argv[2])
which = sys.argv[3] if len(sys.argv) > 3 else "mixed"
sel = [p for i, p in enumerate(probs) if which == "mixed" or (which == "large") == (i % 4 == 3)]
wins = [sel[i % len(sel)] for i in range(N)]
Optimizer.LocalBundleAdjustmentNavStatePRVBatch(wins)
Optimizer.enable_kernel_timing(True)
t = time.perf_counter()
for _ in range(3):
    r = Optimizer.LocalBundleAdjustmentNavStatePRVBatch(wins)
dt = (time.perf_counter() - t) / 3 * 1e3
k, fl = Optimizer.kernel_times()
print("%s, %d windows: %.2f ms per call; trials %s" % (which, N, dt, [int(x[3]["lm_trials"]) for x in r[:8]]))
print({n: (round(v["ms"] / 3, 3), v["launches"] // 3) for n, v in k.items()}, "schur TFLOP/s %.2f" % (fl / (k["lba.schur"]["ms"] * 1e-3) / 1e12))
