#!/bin/bash
# k_describe_fused with parts compiled out (-DVIEO_FUSED_AB=mask, wrong results): where its time goes
cd $GRAFT_REPO_ROOT
for n in 0 1 2 3 4 7; do
  touch vieo_slam_amd/csrc/orb_extractor.hip
  VIEO_EXTRA_HIPCC_FLAGS="-DVIEO_FUSED_AB=$n" python -c "from vieo_slam_amd import build; build.build()" > /dev/null 2>&1
  echo "AB=$n $(timeout 300 python tools/run_extract.py 4096 4 2>&1 | head -1)"
done
touch vieo_slam_amd/csrc/orb_extractor.hip
python -c "from vieo_slam_amd import build; build.build()" > /dev/null 2>&1
