"""One lock-step visual-inertial LBA batch for rocprofv3 (W from argv)."""
import sys
sys.path.insert(0, ".")
import torch  # noqa: F401
from vieo_slam_amd import synth_ba
from vieo_slam_amd.optimizer import Optimizer
W = int(sys.argv[1]) if len(sys.argv) > 1 else 64
wins = [synth_ba.make_lba_vio_problem(100 + s, n_local=10, n_fixed=5, n_points=1000)[:6] for s in range(W)]
for i in range(3):
    Optimizer.LocalBundleAdjustmentNavStatePRVBatch(wins)
