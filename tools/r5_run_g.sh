#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time python -m pytest tests -x -q -m gpu > gpurun_out/r5g_gpu_suite.log 2>&1 ) 2> gpurun_out/r5g_suite.time
echo "suite rc=$?" >> gpurun_out/r5g_gpu_suite.log
( time python bench.py > gpurun_out/bench_r5g.log 2> gpurun_out/bench_r5g.err ) 2> gpurun_out/bench_r5g.time
tail -1 gpurun_out/bench_r5g.log > gpurun_out/r5g_bench_line.json
tail -n 3 gpurun_out/r5g_gpu_suite.log; cat gpurun_out/r5g_suite.time gpurun_out/bench_r5g.time
