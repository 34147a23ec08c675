#!/bin/bash
# kernel trace of the LocalBA share of a bench step alone on the GPU (103 windows in lock step, one host thread)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/lba1.py <<'PY'
import sys, time
sys.path.insert(0, sys.argv[1])
from vieo_slam_amd import synth_ba
from vieo_slam_amd.optimizer import Optimizer
probs = [synth_ba.make_lba_vio_problem(500 + i, n_local=10, n_fixed=6, n_points=2000)[:6] for i in range(8)]
N = int(sys.argv[2])
wins = [probs[i % 8] for i in range(N)]
Optimizer.LocalBundleAdjustmentNavStatePRVBatch(wins)
t = time.perf_counter()
for _ in range(3):
    r = Optimizer.LocalBundleAdjustmentNavStatePRVBatch(wins)
print("%d windows: %.2f ms per call; trials %s" % (N, (time.perf_counter() - t) / 3 * 1e3, [int(x[3]["lm_trials"]) for x in r[:8]]))
PY
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_lba_step -o out -- python /tmp/lba1.py $R ${1:-103} > $R/gpurun_out/prof_lba_step.log 2>&1
tail -2 $R/gpurun_out/prof_lba_step.log
python $R/tools/rocpd_summary.py $(find $R/gpurun_out/prof_lba_step -name "*.db" | head -1) $R/gpurun_out/prof_lba_step.md | head -30
