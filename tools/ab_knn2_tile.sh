#!/bin/bash
# k_knn2_mfma with 32 / 64 train rows per barrier (-DVIEO_KNN2_TILE_ROWS=n): the rig batch's stereo stage and the kernel alone
cd $GRAFT_REPO_ROOT
for n in 32 64; do
  touch vieo_slam_amd/csrc/matching.hip
  VIEO_EXTRA_HIPCC_FLAGS="-DVIEO_KNN2_TILE_ROWS=$n" python -c "from vieo_slam_amd import build; build.build()" > /dev/null 2>&1
  timeout 600 python - <<PY
import sys
sys.argv = ["bench.py"]
import bench
r = bench.rig_frontend_batch()
print("tile_rows=$n", round(r["rig_frames_per_s"], 1), round(r["stage_ms_per_step"]["stereo"], 3), round(r["roofline_knn2"]["avg_launch_ms"], 4), round(r["roofline_knn2"]["mfma"]["frac"], 3))
PY
done
touch vieo_slam_amd/csrc/matching.hip
python -c "from vieo_slam_amd import build; build.build()" > /dev/null 2>&1
timeout 600 python -m pytest tests/test_matching_parity.py tests/test_fisheye_stereo.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
