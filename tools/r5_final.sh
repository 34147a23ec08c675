#!/bin/bash
# the round's closing pass: GPU suite, smoke, default bench (timed), counters, traces
cd "$(dirname "$0")/.."
TAG=${1:-r5z}
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; grep -E "passed|failed|FAILED" gpurun_out/${TAG}_tests.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
SECONDS=0
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$? wall ${SECONDS}s"
timeout 1500 bash tools/pmc_round5.sh > gpurun_out/${TAG}_pmc.log 2>&1; head -1 gpurun_out/${TAG}_pmc.log
timeout 1200 bash tools/prof_round.sh $TAG > gpurun_out/${TAG}_prof.log 2>&1
# a second bench line with the fresh counters in place (traffic filled in)
cp gpurun_out/r5_pmc_extractor.json gpurun_out/r5_pmc_fast.json gpurun_out/r5_pmc_kernels.json profiles/
timeout 900 python bench.py --no-rig-legs --single-stream-frames 0 --no-cpu-baseline --no-multi-gpu-legs --no-pcie-leg --parity-sample 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('roofline with fresh counters:', r['roofline']['frac'], r['roofline']['traffic'])"
