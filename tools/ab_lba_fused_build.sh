#!/bin/bash
# both halves of k_lba_build in one launch (VIEO_LBA_FUSED_BUILD=1, default for calls of <= 4 windows) against two launches
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
timeout 1500 python -m pytest tests/test_lba.py tests/test_lba_vio.py tests/test_global_ba.py tests/test_golden_ba.py -m gpu -x -q > /tmp/t.log 2>&1; grep -E "passed|failed|FAILED|Error" /tmp/t.log | tail -6
python tools/write_sequence.py /tmp/seq.vseq --frames 400 > /dev/null
for rep in 1 2 3; do
for v in 1 0; do
  VIEO_LBA_FUSED_BUILD=$v ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --lba-lag 8 --prefetch 1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('fused_build=$v ms_per_frame', r['ms_per_frame'], 'lba', r['ms_per_local_ba'])"
done; done
