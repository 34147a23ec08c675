#!/bin/bash
cd "$(dirname "$0")/.."
python tools/write_sequence.py /tmp/seq.vseq --frames 30 > /dev/null
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['trackers'], 'fps', d['frames_per_s_all_trackers'], 'lat mean', d['ms_per_frame_latency_mean'], 'median', d['ms_per_frame_latency_median'], 'gpu', d['ms_track_gpu_mean'], 'call', d['ms_track_call_mean'])" "$1"; }
for nt in 2 4 8 16 32; do
  ./examples/replay_main /tmp/seq.vseq --trackers $nt --quiet --kf-every 100000 | q "no_lba"
  VIEO_TRACKER_PRIORITY=0 ./examples/replay_main /tmp/seq.vseq --trackers $nt --quiet --kf-every 100000 | q "no_lba_prio0"
done
GPU_MAX_HW_QUEUES=8 VIEO_TRACKER_PRIORITY=0 ./examples/replay_main /tmp/seq.vseq --trackers 8 --quiet --kf-every 100000 | q "no_lba_prio0_hwq8"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_conc2 -o out -- $GRAFT_REPO_ROOT/examples/replay_main /tmp/seq.vseq --trackers 2 --quiet --kf-every 100000 --frames 12 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $(find $GRAFT_REPO_ROOT/gpurun_out/prof_conc2 -name "*.db" | head -1) 2>/dev/null | tail -120 > $GRAFT_REPO_ROOT/gpurun_out/r5f_conc2_timeline.txt
