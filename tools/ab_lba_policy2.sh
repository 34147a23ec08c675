#!/bin/bash
# the device policy (rounds queued blind) again, now that a blind round is 5 launches (fused build, fused tail)
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
VIEO_LBA_DEVICE_POLICY=1 timeout 900 python -m pytest tests/test_lba.py tests/test_lba_vio.py -m gpu -x -q > /tmp/t2.log 2>&1; echo "with VIEO_LBA_DEVICE_POLICY=1:"; grep -E "passed|failed|FAILED|Error" /tmp/t2.log | tail -4
python tools/write_sequence.py /tmp/seq.vseq --frames 400 > /dev/null
for rep in 1 2 3; do
for v in 1 0; do
  VIEO_LBA_DEVICE_POLICY=$v timeout 120 ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --lba-lag 8 --prefetch 1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('device_policy=$v ms_per_frame', r['ms_per_frame'], 'lba', r['ms_per_local_ba'])"
done; done
