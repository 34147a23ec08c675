#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests/test_lba.py tests/test_lba_vio.py tests/test_replay.py tests/test_dropin_replay.py tests/test_resident_frame.py tests/test_tracker.py tests/test_orb_parity.py tests/test_imu_preint.py -m gpu -x -q 2>&1 | tail -4
python tools/write_sequence.py /tmp/seq.vseq --frames 400 > /dev/null
for fl in 1 0; do for rep in 1 2; do VIEO_SYNC_FLAG=$fl timeout 120 ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --lba-lag 8 --prefetch 1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('one-call sync_flag=$fl', r['ms_per_frame'], r['ms_per_frame_last_200'], r['ms_track_call'], r['ms_track_gpu'], r['ms_per_local_ba'])"; done; done
for fl in 1 0; do for rep in 1 2; do VIEO_SYNC_FLAG=$fl timeout 120 ./examples/dropin_replay /tmp/seq.vseq --warmup 16 --quiet --lba-lag 8 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('drop-in sync_flag=$fl', r['ms_per_frame'], r['ms_per_frame_last_200'], r['stage_ms_per_frame'])"; done; done
