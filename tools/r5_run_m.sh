#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 900 python bench.py > gpurun_out/r5x_bench.json 2> gpurun_out/r5x_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r5x_bench.err
