B="python bench.py --no-cpu-baseline --single-stream-frames 0 --no-pcie-leg --no-multi-gpu-legs"
q() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value']), round(d['ms_per_step'],2))
" $1 "$2"; }
$B > /tmp/o 2>/dev/null; q /tmp/o small_first
VIEO_BENCH_LBA_LARGE_FIRST=1 $B > /tmp/o 2>/dev/null; q /tmp/o large_first
$B --lba-threads 5 > /tmp/o 2>/dev/null; q /tmp/o thr5
$B --steps 20 > /tmp/o 2>/dev/null; q /tmp/o steps20
