#!/bin/bash
# kernel timeline of a few pipelined frames of the sequential replay (who runs beside whom): gpurun_out/$1_pipelined_timeline.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r5z}
python $R/tools/write_sequence.py /tmp/seq.vseq --frames 60 > /dev/null
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_pipe -o out -- $R/examples/replay_main /tmp/seq.vseq --quiet --lba-lag 8 --prefetch 1 > /dev/null 2>&1
python $R/tools/rocpd_timeline.py $(find $R/gpurun_out/prof_pipe -name "*.db" | head -1) k_track_adopt 5 120 > $R/gpurun_out/${TAG}_pipelined_timeline.txt 2>&1
head -3 $R/gpurun_out/${TAG}_pipelined_timeline.txt
