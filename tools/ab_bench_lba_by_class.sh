B="python bench.py --no-cpu-baseline --single-stream-frames 0 --no-pcie-leg --no-multi-gpu-legs"
q() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value']), round(d['ms_per_step'],2), d['config']['workload'][-120:])
" $1 "$2"; }
$B > /tmp/o 2>/dev/null; q /tmp/o by_class
VIEO_BENCH_LBA_SMALL_CALLS=2 $B > /tmp/o 2>/dev/null; q /tmp/o by_class_small2
VIEO_BENCH_LBA_SMALL_CALLS=3 $B --lba-threads 6 > /tmp/o 2>/dev/null; q /tmp/o by_class_small3_thr6
$B --lba-threads 3 > /tmp/o 2>/dev/null; q /tmp/o by_class_thr3
$B --lba-mixed > /tmp/o 2>/dev/null; q /tmp/o mixed
