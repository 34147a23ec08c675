#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_tracker.py tests/test_tracker_vision.py tests/test_tracker_rig.py -x -q -m gpu -s > gpurun_out/r5h_tracker.log 2>&1; echo "rc=$?" >> gpurun_out/r5h_tracker.log
python tools/write_sequence.py /tmp/seq.vseq --frames 200 > /dev/null
for pf in 0 1; do for rep in 1 2 3; do ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --lba-lag 6 --prefetch $pf; done; done > gpurun_out/r5h_replay.log 2>&1
tail -n 12 gpurun_out/r5h_tracker.log
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r5h_replay.log") if l.startswith("{")]
for r in rows: print(r["prefetch"], r["ms_per_frame"], r["ms_per_frame_last_200"], r["ms_track_call"], r["ms_track_gpu"], r["ms_per_local_ba"])
PY
