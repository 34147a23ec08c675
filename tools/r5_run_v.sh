#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_matching_parity.py tests/test_fisheye_stereo.py tests/test_tracker_rig.py -m gpu -x -q > /tmp/t.log 2>&1; grep -E "passed|failed|FAILED|Error" /tmp/t.log | tail -8
for m in 1 0; do VIEO_KNN2_MFMA=$m timeout 600 python - <<'PY'
import os, json, sys
sys.argv = ["bench.py"]
import bench
import torch
r = bench.rig_frontend_batch()
print("mfma=%s" % os.environ["VIEO_KNN2_MFMA"], None if r is None else (r["rig_frames_per_s"], r["stage_ms_per_step"]["stereo"], r["roofline_knn2"]["avg_launch_ms"], r["roofline_knn2"].get("mfma")))
PY
done
