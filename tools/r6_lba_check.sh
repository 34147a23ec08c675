#!/bin/bash
# after a change to lba.hip: its parity tests, then timed replays: host loop / device policy, fused tail on / off
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
timeout 1500 python -m pytest tests/test_lba.py tests/test_lba_vio.py tests/test_global_ba.py tests/test_global_ba_scale.py tests/test_golden_ba.py -m gpu -x -q > /tmp/t.log 2>&1; grep -E "passed|failed|FAILED|Error" /tmp/t.log | tail -6
VIEO_LBA_DEVICE_POLICY=1 timeout 1500 python -m pytest tests/test_lba.py tests/test_lba_vio.py -m gpu -x -q > /tmp/t2.log 2>&1; echo "with VIEO_LBA_DEVICE_POLICY=1:"; grep -E "passed|failed|FAILED|Error" /tmp/t2.log | tail -6
python tools/write_sequence.py /tmp/seq.vseq --frames 400 > /dev/null
for cfg in "1 1" "1 0" "0 1" "0 0" "1 1" "0 1"; do set -- $cfg; VIEO_LBA_DEVICE_POLICY=$1 VIEO_LBA_FUSED_TAIL=$2 timeout 120 ./examples/replay_main /tmp/seq.vseq --prefetch 1 --warmup 16 --quiet --lba-lag 8 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('device_policy=$1 fused_tail=$2 replay_main ms_per_frame', r['ms_per_frame'], 'track call', r['ms_track_call'], 'lba', r['ms_per_local_ba'])"; done
