B="python bench.py --no-cpu-baseline --single-stream-frames 0 --no-pcie-leg --no-multi-gpu-legs"
q() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value']), round(d['ms_per_step'],2), 'fe', round(d['stage_ms_per_step_stream0']['total'],2))
" $1 "$2"; }
$B > /tmp/o 2>/dev/null; q /tmp/o fe_normal
VIEO_ORB_PRIORITY=1 $B > /tmp/o 2>/dev/null; q /tmp/o fe_high
VIEO_ORB_PRIORITY=1 VIEO_LBA_PRIORITY=0 $B > /tmp/o 2>/dev/null; q /tmp/o fe_high_lba_normal
