#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tools/micro/call_overhead > gpurun_out/r5b_call_overhead.txt 2>&1
tools/micro/call_overhead 400000 8192 >> gpurun_out/r5b_call_overhead.txt 2>&1
python tools/write_sequence.py /tmp/seq.vseq --frames 100 > /dev/null
for rep in 1 2 3 4 5 6; do ./examples/replay_main /tmp/seq.vseq --warmup 12 --quiet --lba-lag 6; done > gpurun_out/r5b_replay_main.log 2>&1
for rep in 1 2 3 4; do VIEO_TRACKER_PRIORITY=0 ./examples/replay_main /tmp/seq.vseq --warmup 12 --quiet --lba-lag 6; done > gpurun_out/r5b_replay_main_prio0.log 2>&1
for rep in 1 2 3; do ./examples/dropin_replay /tmp/seq.vseq --warmup 12 --quiet --lba-lag 6; done > gpurun_out/r5b_dropin.log 2>&1
python -m pytest tests -x -q -m gpu > gpurun_out/r5b_gpu_suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/r5b_gpu_suite.log
cat gpurun_out/r5b_call_overhead.txt
python - <<'PY'
import json
for f in ("r5b_replay_main","r5b_replay_main_prio0","r5b_dropin"):
    rows=[json.loads(l) for l in open("gpurun_out/%s.log"%f) if l.startswith("{")]
    print(f, [r["ms_per_frame"] for r in rows], [r.get("ms_track_gpu") for r in rows])
PY
tail -n 4 gpurun_out/r5b_gpu_suite.log
