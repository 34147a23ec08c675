#!/bin/bash
# the batched bench step with the bundle-adjustment streams at the lowest / default / highest priority x 2 / 4 / 8 issuing threads
cd "$(dirname "$0")/.."
q() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(sys.argv[1], round(r['value'],1), round(r['ms_per_step'],3), 'fe', round(r['stage_ms_per_step_stream0']['total'],2), 'lba ms/window', round(r['config']['local_ba_ms_per_window_mean'],3))" "$1"; }
A="--no-cpu-baseline --no-pcie-leg --no-rig-legs --no-multi-gpu-legs --single-stream-frames 0 --parity-sample 0"
for p in -1 0 1; do for t in 4 8; do VIEO_LBA_PRIORITY=$p timeout 600 python bench.py $A --lba-threads $t 2>/dev/null | q "prio$p threads$t"; done; done
VIEO_LBA_PRIORITY=-1 timeout 600 python bench.py $A --lba-threads 2 2>/dev/null | q "prio-1 threads2"
VIEO_LBA_PRIORITY=-1 timeout 600 python bench.py $A --steps 30 2>/dev/null | q "prio-1 steps30"
