#!/bin/bash
# rocprofv3 kernel traces of the round's two measured paths, summaries under gpurun_out/ (copy to profiles/):
#   $1_full_step.md      python bench.py (batched step, workload r3), 5 timed steps
#   $1_cpp_replay.md     examples/replay_main over 60 frames (the single_stream leg's program)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_step -o out -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline \
  --single-stream-frames 0 --no-pcie-leg --no-multi-gpu-legs > $R/gpurun_out/prof_step.log 2>&1
python $R/tools/rocpd_summary.py $(find $R/gpurun_out/prof_step -name "*.db" | head -1) $R/gpurun_out/${TAG}_full_step.md > /dev/null
python $R/tools/write_sequence.py /tmp/seq.vseq --frames 60 > /dev/null
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_replay -o out -- $R/examples/replay_main /tmp/seq.vseq --warmup 12 --quiet > $R/gpurun_out/prof_replay.log 2>&1
python $R/tools/rocpd_summary.py $(find $R/gpurun_out/prof_replay -name "*.db" | head -1) $R/gpurun_out/${TAG}_cpp_replay.md > /dev/null
tail -1 $R/gpurun_out/prof_step.log | cut -c1-300
tail -1 $R/gpurun_out/prof_replay.log
