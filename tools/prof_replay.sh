#!/bin/bash
# rocprofv3 kernel trace of the C++ sequence replay alone (60 frames, local BA beside tracking): gpurun_out/$1_cpp_replay.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r4}
export PYTHONPATH=$R
python $R/tools/write_sequence.py /tmp/seq.vseq --frames 60 > /dev/null
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_replay -o out -- $R/examples/replay_main /tmp/seq.vseq --warmup 12 --quiet --lba-lag 6 > $R/gpurun_out/prof_replay.log 2>&1
python $R/tools/rocpd_summary.py $(find $R/gpurun_out/prof_replay -name "*.db" | head -1) $R/gpurun_out/${TAG}_cpp_replay.md --merge-grids > /dev/null
tail -1 $R/gpurun_out/prof_replay.log
