#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_replay.py tests/test_dropin_replay.py tests/test_replay_modes.py -m gpu -x -q > /tmp/t.log 2>&1; grep -E "passed|failed|FAILED" /tmp/t.log | tail -5
python tools/write_sequence.py /tmp/seq.vseq --frames 400 > /dev/null
for fr in 1 0; do for rep in 1 2; do VIEO_LBA_FAST_ROUNDS=$fr timeout 120 ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --lba-lag 8 --prefetch 1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('fast_rounds=$fr', r['ms_per_frame'], r['ms_per_frame_last_200'], r['ms_track_call'], r['ms_track_gpu'], r['ms_per_local_ba'], r['caller_ms_per_frame'])"; done; done
for fr in 1 0; do VIEO_LBA_FAST_ROUNDS=$fr timeout 120 ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --prefetch 1 --frames 200 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('inline fast_rounds=$fr', r['ms_per_frame'], r['ms_per_local_ba'])"; done
