#!/bin/bash
# the sequential replay with its LocalMapping thread's bundle-adjustment stream at the lowest / default / highest priority
cd "$(dirname "$0")/.."
python tools/write_sequence.py /tmp/seq.vseq --frames 400 > /dev/null
for p in -1 0 1; do for rep in 1 2 3; do VIEO_REPLAY_LBA_PRIORITY=$p timeout 120 ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --lba-lag 8 --prefetch 1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('lba_prio$p', r['ms_per_frame'], r['ms_per_frame_last_200'], r['ms_track_call'], r['ms_track_gpu'], r['ms_per_local_ba'], r['caller_ms_per_frame']['map_write_back'])"; done; done
