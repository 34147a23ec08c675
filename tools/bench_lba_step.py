"""The LocalBundleAdjustmentNavStatePRV share of one bench.py step (103 windows of the benchmark's shape), alone on
the GPU: one lock-step batch from one host thread, and split over two host threads as bench.py issues it."""
import sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, ".")
import torch  # noqa: F401
from vieo_slam_amd import synth_ba
from vieo_slam_amd.optimizer import Optimizer

probs = [synth_ba.make_lba_vio_problem(500 + i, n_local=10, n_fixed=6, n_points=2000)[:6] for i in range(8)]
N = 103
wins = [probs[i % 8] for i in range(N)]


def run(idx):
    return Optimizer.LocalBundleAdjustmentNavStatePRVBatch([wins[i] for i in idx])


for threads in (1, 2, 4):
    pool = ThreadPoolExecutor(max_workers=threads)
    chunks = [list(range(i, N, threads)) for i in range(threads)]
    list(pool.map(run, chunks * 2))
    t = time.perf_counter()
    reps = 4
    for _ in range(reps):
        list(pool.map(run, chunks))
    dt = (time.perf_counter() - t) / reps
    print("%d host thread(s): %.2f ms for %d windows (%.3f ms per window)" % (threads, dt * 1e3, N, dt * 1e3 / N))
