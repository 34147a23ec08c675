#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests/test_tracker.py tests/test_tracker_rig.py tests/test_tracker_vision.py tests/test_replay_modes.py -m gpu -x -q > /tmp/t.log 2>&1; grep -E "passed|failed|FAILED|Error" /tmp/t.log | tail -6
python tools/write_sequence.py /tmp/seq.vseq --frames 400 > /dev/null
for rep in 1 2 3 4 5 6; do timeout 120 ./examples/replay_main /tmp/seq.vseq /tmp/traj$rep.bin --warmup 16 --quiet --lba-lag 8 --prefetch 1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('lag8', r['ms_per_frame'], r['ms_per_frame_last_200'], r['ms_track_call'], r['ms_track_gpu'], r['ms_per_local_ba'], r['ms_per_frame_p99'])"; done
