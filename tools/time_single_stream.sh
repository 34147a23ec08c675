#!/bin/bash
# quick single-stream timings (no profiler): the C++ replay, the 2- and 4-camera rig tracker
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
python $R/tools/write_sequence.py /tmp/seq.vseq --frames 60 > /dev/null
$R/examples/replay_main /tmp/seq.vseq --warmup 12 --quiet | tail -1
python $R/tools/run_rig_tracker.py kb8 4 1500 20 | tail -1
python $R/tools/run_rig_tracker.py radtan 2 1200 20 | tail -1
