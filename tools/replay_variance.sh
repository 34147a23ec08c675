#!/bin/bash
# run-to-run spread of the sequential replay (six runs per write-back lag) with the caller's own time per frame
cd "$(dirname "$0")/.."
python tools/write_sequence.py /tmp/seq.vseq --frames 400 > /dev/null
for lag in 8 9; do for rep in 1 2 3 4 5 6; do timeout 120 ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --lba-lag $lag --prefetch 1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('lag$lag', r['ms_per_frame'], r['ms_per_frame_last_200'], r['ms_track_call'], r['ms_track_gpu'], r['ms_per_local_ba'], r['caller_ms_per_frame'], r['ms_per_frame_p99'])"; done; done
nproc; cat /proc/cpuinfo | grep "model name" | head -1
