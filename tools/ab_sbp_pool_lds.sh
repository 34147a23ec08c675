B="python bench.py --no-cpu-baseline --single-stream-frames 0 --no-pcie-leg --no-multi-gpu-legs --lba-every 0 --steps 6"
q() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms_per_step_stream0'].items()})
" $1 "$2"; }
for v in 1024 2048 3072 5120 8192; do VIEO_SBP_POOL_LDS=$v $B > /tmp/o 2>/dev/null; q /tmp/o pool_lds_$v; done
