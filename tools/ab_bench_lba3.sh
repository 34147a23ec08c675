B="python bench.py --no-cpu-baseline --single-stream-frames 0 --no-pcie-leg --no-multi-gpu-legs"
q() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value']), round(d['ms_per_step'],2), 'fe', round(d['stage_ms_per_step_stream0']['total'],2))
" $1 "$2"; }
$B --lba-every 0 > /tmp/o1 2>/dev/null; q /tmp/o1 fe_only
VIEO_BENCH_SKIP_FE=1 $B > /tmp/o2 2>/dev/null; q /tmp/o2 lba_only_4thr
VIEO_BENCH_SKIP_FE=1 $B --lba-threads 8 > /tmp/o3 2>/dev/null; q /tmp/o3 lba_only_8thr
VIEO_BENCH_SKIP_FE=1 $B --lba-threads 2 > /tmp/o4 2>/dev/null; q /tmp/o4 lba_only_2thr
VIEO_BENCH_SKIP_FE=1 $B --lba-threads 1 > /tmp/o5 2>/dev/null; q /tmp/o5 lba_only_1thr
