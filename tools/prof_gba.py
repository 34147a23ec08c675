"""One GlobalBundleAdjustmentNavStatePRV of 400 key frames under rocprofv3 (run on the GPU box)."""
import sys
sys.path.insert(0, ".")
from vieo_slam_amd import synth_ba
from vieo_slam_amd.optimizer import Optimizer

n_local = int(sys.argv[1]) if len(sys.argv) > 1 else 400
params, kfs, pts, close, obs, imu, gt = synth_ba.make_lba_vio_problem(
    7, n_local=n_local, n_fixed=1, n_points=100 * n_local, anchors=n_local // 2, span=5)
Optimizer.GlobalBundleAdjustmentNavStatePRV(params, kfs, pts, obs, imu, 1, True)
hn, hp, hres = Optimizer.GlobalBundleAdjustmentNavStatePRV(params, kfs, pts, obs, imu, 5, True)
print(hres)
