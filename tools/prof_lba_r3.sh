#!/bin/bash
# kernel trace of the r3 bench step's local-BA share ALONE on the GPU: 205 windows in lock step (154 ordinary: 10 local +
# 40 fixed key frames; 51 bLarge: 25 local), one host thread
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/lba3.py <<'PY'
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
PY
for which in mixed small large; do
  n=205; [ $which = small ] && n=154; [ $which = large ] && n=51
  python /tmp/lba3.py $R $n $which 2>&1 | grep -v "^$" | tail -2
done
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_lba_r3 -o out -- python /tmp/lba3.py $R 205 mixed > $R/gpurun_out/prof_lba_r3.log 2>&1
python $R/tools/rocpd_summary.py $(find $R/gpurun_out/prof_lba_r3 -name "*.db" | head -1) $R/gpurun_out/r3_lba_alone_205_windows.md | head -24
