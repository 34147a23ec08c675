#!/bin/bash
cd "$(dirname "$0")/.."
q() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(sys.argv[1], round(r['value'],1), round(r['ms_per_step'],3), {k:round(v,3) for k,v in r['stage_ms_per_step_stream0'].items()})" "$1"; }
A="--no-cpu-baseline --no-pcie-leg --no-rig-legs --no-multi-gpu-legs --single-stream-frames 0 --parity-sample 0"
for w in 1 2; do VIEO_POSE_NARROW_WAVES=$w timeout 600 python bench.py $A 2>/dev/null | q "waves$w"; VIEO_POSE_NARROW_WAVES=$w timeout 600 python bench.py $A --lba-every 0 2>/dev/null | q "waves$w-nolba"; done
timeout 900 python -m pytest tests/test_pose_opt_vio_parity.py tests/test_pipeline.py -m gpu -x -q 2>&1 | tail -3
