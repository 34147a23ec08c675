#!/bin/bash
# the C++ replays after a change to examples/: their GPU tests + three timed runs of each
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests/test_replay.py tests/test_dropin_replay.py -m gpu -x -q > /tmp/t.log 2>&1; grep -E "passed|failed|FAILED|Error" /tmp/t.log | tail -4
python tools/write_sequence.py /tmp/seq.vseq --frames 400 > /dev/null
for exe in "replay_main /tmp/seq.vseq --prefetch 1" "dropin_replay /tmp/seq.vseq"; do for rep in 1 2 3; do timeout 120 ./examples/$exe --warmup 16 --quiet --lba-lag 8 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$exe', r['ms_per_frame'], r['ms_per_frame_last_200'], r['ms_per_local_ba'])"; done; done
