#!/bin/bash
# N concurrent trackers on one GPU (examples/replay_main --trackers N)
cd "$(dirname "$0")/.."
python tools/write_sequence.py /tmp/seq.vseq --frames 60 > /dev/null
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['trackers'], 'fps', d['frames_per_s_all_trackers'], 'lat mean', d['ms_per_frame_latency_mean'], 'median', d['ms_per_frame_latency_median'], 'gpu', d['ms_track_gpu_mean'], 'call', d['ms_track_call_mean'])" "$1"; }
for nt in 2 4 8 16; do
  ./examples/replay_main /tmp/seq.vseq --trackers $nt --quiet --lba-lag 6 | q "default"
  VIEO_TRACKER_PRIORITY=0 ./examples/replay_main /tmp/seq.vseq --trackers $nt --quiet --lba-lag 6 | q "prio0"
  GPU_MAX_HW_QUEUES=16 ./examples/replay_main /tmp/seq.vseq --trackers $nt --quiet --lba-lag 6 | q "hwq16"
  GPU_MAX_HW_QUEUES=16 VIEO_TRACKER_PRIORITY=0 ./examples/replay_main /tmp/seq.vseq --trackers $nt --quiet --lba-lag 6 | q "hwq16_prio0"
done
GPU_MAX_HW_QUEUES=32 VIEO_TRACKER_PRIORITY=0 ./examples/replay_main /tmp/seq.vseq --trackers 32 --quiet --lba-lag 6 | q "hwq32_prio0"
GPU_MAX_HW_QUEUES=8 VIEO_TRACKER_PRIORITY=0 ./examples/replay_main /tmp/seq.vseq --trackers 8 --quiet --lba-lag 0 | q "hwq8_prio0_inlineLBA"
