#!/bin/bash
# rocprofv3 kernel traces of the round's measured paths, summaries under gpurun_out/ (copy to profiles/):
#   $1_full_step.md      python bench.py (batched step, workload r3), 5 timed steps; rows per (kernel, grid) and the roofline
#                        lines of the dominant kernels recomputed from the trace alone
#   $1_cpp_replay.md     examples/replay_main over 60 frames with the local BA beside tracking and frame pipelining (the single_stream
#                        leg's program);  $1_dropin_replay.md: examples/dropin_replay, the same frames member by member
#   $1_rig_*.md          the one-call rig tracker (tools/prof_rig_tracker.sh)
#   $1_cpp_rig_*.md, $1_cpp_vision.md   examples/replay_modes on the camera-rig / vision-only sequences (60 frames, lag 8, pipelined)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r4}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_step -o out -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline \
  --single-stream-frames 0 --no-pcie-leg --no-multi-gpu-legs --no-rig-legs > $R/gpurun_out/prof_step.log 2>&1
# algorithmic bytes / FLOPs per BATCH launch: k_fast 1 117 367 B x 4096 images; k_knn2 is in the rig profile
python $R/tools/rocpd_summary.py $(find $R/gpurun_out/prof_step -name "*.db" | head -1) $R/gpurun_out/${TAG}_full_step.md --min-us 1.0 \
  --bytes k_fast=4576735232@1000000 > /dev/null
python $R/tools/write_sequence.py /tmp/seq.vseq --frames 60 > /dev/null
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_replay -o out -- $R/examples/replay_main /tmp/seq.vseq --warmup 12 --quiet --lba-lag 8 --prefetch 1 > $R/gpurun_out/prof_replay.log 2>&1
python $R/tools/rocpd_summary.py $(find $R/gpurun_out/prof_replay -name "*.db" | head -1) $R/gpurun_out/${TAG}_cpp_replay.md --merge-grids > /dev/null
# the same sequence member by member (the resident drop-in calls of shim/*.cc)
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dropin -o out -- $R/examples/dropin_replay /tmp/seq.vseq --warmup 12 --quiet --lba-lag 8 > $R/gpurun_out/prof_dropin.log 2>&1
python $R/tools/rocpd_summary.py $(find $R/gpurun_out/prof_dropin -name "*.db" | head -1) $R/gpurun_out/${TAG}_dropin_replay.md --merge-grids > /dev/null
bash $R/tools/prof_rig_tracker.sh $TAG > /dev/null 2>&1
# the other configurations' sequences as C++ programs (examples/replay_modes: rig / --vision), local BA beside tracking
export PYTHONPATH=$R
for cfg in "kb8 4 1500 5" "radtan 2 1200 3"; do
  set -- $cfg
  python $R/tools/write_sequence.py /tmp/rig_$1_$2.vseq --rig $1 --cams $2 --features $3 --seed $4 --frames 60 > /dev/null
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cpp_$1_$2 -o out -- $R/examples/replay_modes /tmp/rig_$1_$2.vseq --warmup 12 --quiet --lba-lag 8 --prefetch 1 > $R/gpurun_out/prof_cpp_$1_$2.log 2>&1
  python $R/tools/rocpd_summary.py $(find $R/gpurun_out/prof_cpp_$1_$2 -name "*.db" | head -1) $R/gpurun_out/${TAG}_cpp_rig_$1_$2.md --merge-grids > /dev/null
done
python $R/tools/write_sequence.py /tmp/seq2.vseq --seed 2 --frames 60 > /dev/null
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cpp_vision -o out -- $R/examples/replay_modes /tmp/seq2.vseq --vision --warmup 12 --quiet --lba-lag 8 --prefetch 1 > $R/gpurun_out/prof_cpp_vision.log 2>&1
python $R/tools/rocpd_summary.py $(find $R/gpurun_out/prof_cpp_vision -name "*.db" | head -1) $R/gpurun_out/${TAG}_cpp_vision.md --merge-grids > /dev/null
tail -n 1 $R/gpurun_out/prof_cpp_*.log | cut -c1-400
tail -1 $R/gpurun_out/prof_step.log | cut -c1-300
tail -1 $R/gpurun_out/prof_replay.log
