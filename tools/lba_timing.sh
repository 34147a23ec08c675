#!/bin/bash
# host phases of the single-window local BA in the C++ replay (VIEO_LBA_TIMING=1): staging, rounds, waiting, results
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
python tools/write_sequence.py /tmp/seq.vseq --frames 200 > /dev/null
VIEO_LBA_TIMING=1 ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --lba-lag 8 --prefetch 1 2>&1 | tail -12
