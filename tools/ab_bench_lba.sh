B="python bench.py --no-cpu-baseline --single-stream-frames 0 --no-pcie-leg --no-multi-gpu-legs"
q() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value']), round(d['ms_per_step'],2), 'fe', round(d['stage_ms_per_step_stream0']['total'],2), 'fast', round(d['extractor_kernel_ms_per_step_stream0']['fast'],2))
" $1 "$2"; }
$B > /tmp/o1 2>/dev/null; q /tmp/o1 base
VIEO_BENCH_LBA_COUNT_ONLY=1 $B > /tmp/o2 2>/dev/null; q /tmp/o2 count_only
VIEO_BENCH_LBA_COUNT_ONLY=1 VIEO_LBA_PRIORITY=0 $B > /tmp/o3 2>/dev/null; q /tmp/o3 count_only_prio0
VIEO_BENCH_LBA_COUNT_ONLY=1 VIEO_LBA_CU_MASK=0,32 $B > /tmp/o4 2>/dev/null; q /tmp/o4 mask32
VIEO_BENCH_LBA_COUNT_ONLY=1 VIEO_LBA_CU_MASK=0,64 $B > /tmp/o5 2>/dev/null; q /tmp/o5 mask64
VIEO_BENCH_LBA_COUNT_ONLY=1 $B --lba-threads 8 > /tmp/o6 2>/dev/null; q /tmp/o6 thr8
VIEO_BENCH_LBA_COUNT_ONLY=1 $B --lba-threads 2 > /tmp/o7 2>/dev/null; q /tmp/o7 thr2
VIEO_BENCH_LBA_COUNT_ONLY=1 $B --lba-every 0 > /tmp/o8 2>/dev/null; q /tmp/o8 nolba
