#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests/test_orb_parity.py tests/test_matching_parity.py tests/test_tracker.py tests/test_resident_frame.py tests/test_dropin_replay.py -m gpu -x -q > /tmp/t.log 2>&1; grep -E "passed|failed|FAILED|Error" /tmp/t.log | tail -8
python tools/write_sequence.py /tmp/seq.vseq --frames 400 > /dev/null
for rep in 1 2 3; do timeout 120 ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --lba-lag 8 --prefetch 1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('one-call', r['ms_per_frame'], r['ms_per_frame_last_200'], r['ms_track_call'], r['ms_track_gpu'], r['ms_per_local_ba'])"; done
for rep in 1 2; do timeout 120 ./examples/dropin_replay /tmp/seq.vseq --warmup 16 --quiet --lba-lag 8 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('drop-in', r['ms_per_frame'], r['ms_per_frame_last_200'], r['stage_ms_per_frame'])"; done
for B in 2 8 4096; do echo "B=$B $(timeout 300 python tools/run_extract.py $B 4 2>&1 | head -2 | tr '\n' ' ')"; done
