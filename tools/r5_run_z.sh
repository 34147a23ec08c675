#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests/test_lba.py tests/test_lba_vio.py -m gpu -q > /tmp/t.log 2>&1; grep -E "passed|failed|FAILED|Error" /tmp/t.log | tail -12
python tools/write_sequence.py /tmp/seq.vseq --frames 400 > /dev/null
for dp in 1 0; do for rep in 1 2; do VIEO_LBA_DEVICE_POLICY=$dp timeout 120 ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --lba-lag 8 --prefetch 1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('device_policy=$dp', r['ms_per_frame'], r['ms_per_frame_last_200'], r['ms_track_call'], r['ms_per_local_ba'], r['ate_rmse_vs_truth_m'], r['caller_ms_per_frame']['map_write_back'])"; done; done
for dp in 1 0; do VIEO_LBA_DEVICE_POLICY=$dp timeout 120 ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --prefetch 1 --frames 200 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('inline device_policy=$dp', r['ms_per_frame'], r['ms_per_local_ba'], r['ate_rmse_vs_truth_m'])"; done
