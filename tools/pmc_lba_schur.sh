#!/bin/bash
# MFMA-busy counters of k_lba_schur in the 205-window local-BA batch of a bench step (workload r3: 40 fixed key frames, every 4th window bLarge) (own rocprofv3 --pmc pass,
# kernel trace only); writes gpurun_out/pmc_lba_schur.{txt,json}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/lba1.py <<'PY'
import sys
sys.path.insert(0, sys.argv[1])
from vieo_slam_amd import synth_ba
from vieo_slam_amd.optimizer import Optimizer
probs = []
for i in range(8):  # the r3 bench step's windows exactly (bench.py): 40 fixed key frames, every 4th window bLarge
    large = i % 4 == 3
    w = synth_ba.make_lba_vio_problem(500 + i, n_local=25 if large else 10, n_fixed=40, n_points=2000)[:6]
    w[0][0]["large"] = int(large)
    if large:
        w[0][0]["base"]["its0"], w[0][0]["base"]["its1"] = 2, 2
    probs.append(w)
wins = [probs[i % 8] for i in range(int(sys.argv[2]))]
for _ in range(2):
    Optimizer.LocalBundleAdjustmentNavStatePRVBatch(wins)
PY
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace \
  -d $R/gpurun_out/pmc_lba_schur -o out -- python /tmp/lba1.py $R ${1:-205} > $R/gpurun_out/pmc_lba_schur.log 2>&1
tail -3 $R/gpurun_out/pmc_lba_schur.log
db=$(find $R/gpurun_out/pmc_lba_schur -name "*.db" | head -1)
python $R/tools/rocpd_pmc.py $db > $R/gpurun_out/pmc_lba_schur_all.txt 2>&1
grep -A6 "k_lba_schur" $R/gpurun_out/pmc_lba_schur_all.txt | tee $R/gpurun_out/pmc_lba_schur.txt
python - $R/gpurun_out/pmc_lba_schur.txt $R/gpurun_out/pmc_lba_schur.json <<'PY'
import json, re, sys
v = {}
for line in open(sys.argv[1]):
    m = re.match(r"\s+(\S+)\s+avg (\S+)\s+\(n=(\d+)\)", line)
    if m:
        v[m.group(1)] = float(m.group(2))
out = {"source": "tools/pmc_lba_schur.sh: rocprofv3 --pmc (own pass), the r3 bench step's 205 visual-inertial windows in one lock-step call (mixed ordinary / bLarge), "
                 "averages per counter instance and launch", "counters": v}
# one SQ counter instance = one (XCD, shader engine) pair = 8 CUs = 32 SIMDs on this part; GRBM_GUI_ACTIVE = the
# launch's duration in clock cycles; SQ_VALU_MFMA_BUSY_CYCLES counts clock cycles summed over the instance's SIMDs
if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE"):
    out["mfma_busy_fraction_of_simd_time"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * v["GRBM_GUI_ACTIVE"])
    out["unit_note"] = "MFMA_BUSY / (32 SIMDs per counter instance x GRBM_GUI_ACTIVE cycles)"
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(out)
PY
