#!/bin/bash
# the round's closing pass on the GPU box: GPU suite, smoke, counters (stamped with the sources' hashes, put in place for
# the bench), default bench (timed), traces.  Everything judged lands in gpurun_out/ (copied to profiles/ afterwards).
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
TAG=${1:-r6z}
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; grep -E "passed|failed|FAILED" gpurun_out/${TAG}_tests.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1500 bash tools/pmc_round6.sh > gpurun_out/${TAG}_pmc.log 2>&1
timeout 600 bash tools/pmc_pose.sh > gpurun_out/r6_pmc_pose.txt 2>&1
cp gpurun_out/r6_pmc_extractor.json gpurun_out/r6_pmc_kernels.json gpurun_out/r6_pmc_lba_schur.json gpurun_out/r6_pmc_pose.txt profiles/
python tools/finish_fast_profile.py gpurun_out/r6_pmc_fast.json profiles/r6_pmc_fast.json > /dev/null 2>&1; cp profiles/r6_pmc_fast.json gpurun_out/r6_pmc_fast_final.json
SECONDS=0
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$? wall ${SECONDS}s"
cp bench_detail.json gpurun_out/${TAG}_bench_detail.json
timeout 1200 bash tools/prof_round.sh $TAG > gpurun_out/${TAG}_prof.log 2>&1
timeout 300 bash tools/prof_pipelined_timeline.sh $TAG > /dev/null 2>&1
find gpurun_out -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
tail -c 3900 gpurun_out/${TAG}_bench.json
