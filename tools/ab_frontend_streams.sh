#!/bin/bash
# the batched bench step with the front end on 1 / 2 / 4 streams (with and without the local BA)
cd "$(dirname "$0")/.."
q() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(sys.argv[1], round(r['value'],1), round(r['ms_per_step'],3), 'fe', round(r['stage_ms_per_step_stream0']['total'],2), r['config'].get('hip_streams_per_gpu'))" "$1"; }
A="--no-cpu-baseline --no-pcie-leg --no-rig-legs --no-multi-gpu-legs --single-stream-frames 0 --parity-sample 0"
for s in 1 2 4; do timeout 600 python bench.py $A --streams $s 2>/dev/null | q "streams$s"; done
for s in 1 2; do timeout 600 python bench.py $A --streams $s --lba-every 0 2>/dev/null | q "streams$s-nolba"; done
