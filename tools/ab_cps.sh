B="python bench.py --no-cpu-baseline --single-stream-frames 0 --no-pcie-leg --no-multi-gpu-legs"
python tools/write_sequence.py /tmp/seq.vseq --frames 100 > /dev/null
for c in 6 8 12 16; do
  export VIEO_LBA_CPS=$c
  v=$($B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2))")
  r=$(examples/replay_main /tmp/seq.vseq --warmup 12 --quiet | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_frame'], d['ms_per_local_ba'])")
  l=$(VIEO_BENCH_SKIP_FE=1 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2))")
  echo "CPS=$c bench $v | replay $r | lba-only $l"
done
