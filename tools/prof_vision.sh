#!/bin/bash
# rocprofv3 kernel trace of the vision-only one-call tracker (tests/test_tracker_vision.py): gpurun_out/$1_vision.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r4}
export PYTHONPATH=$R
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_vision -o out -- python -m pytest $R/tests/test_tracker_vision.py -m gpu -x -q > $R/gpurun_out/prof_vision.log 2>&1
python $R/tools/rocpd_summary.py $(find $R/gpurun_out/prof_vision -name "*.db" | head -1) $R/gpurun_out/${TAG}_vision.md --merge-grids > /dev/null
tail -1 $R/gpurun_out/prof_vision.log
