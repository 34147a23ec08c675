B="python bench.py --no-cpu-baseline --single-stream-frames 0 --no-pcie-leg --no-multi-gpu-legs"
q() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value']), round(d['ms_per_step'],2), 'fe', round(d['stage_ms_per_step_stream0']['total'],2))
" $1 "$2"; }
$B > /tmp/o1 2>/dev/null; q /tmp/o1 base
$B --lba-batch 103 > /tmp/o2 2>/dev/null; q /tmp/o2 batch103
$B --lba-batch 52 --lba-threads 6 > /tmp/o3 2>/dev/null; q /tmp/o3 batch52_thr6
$B --lba-threads 3 > /tmp/o4 2>/dev/null; q /tmp/o4 thr3
$B --steps 20 > /tmp/o5 2>/dev/null; q /tmp/o5 steps20
$B --streams 2 > /tmp/o6 2>/dev/null; q /tmp/o6 streams2
