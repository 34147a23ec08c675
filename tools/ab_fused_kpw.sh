#!/bin/bash
# k_describe_fused with 1 / 2 / 4 / 8 key points per wavefront (-DVIEO_FUSED_KPW=n): extractor stage times per 4096 images + checksum
cd $GRAFT_REPO_ROOT
for n in 1 2 4 8; do
  touch vieo_slam_amd/csrc/orb_extractor.hip
  VIEO_EXTRA_HIPCC_FLAGS="-DVIEO_FUSED_KPW=$n" python -c "from vieo_slam_amd import build; build.build()" > /dev/null 2>&1
  echo "KPW=$n $(timeout 300 python tools/run_extract.py 4096 4 2>&1 | head -2 | tr '\n' ' ')"
done
touch vieo_slam_amd/csrc/orb_extractor.hip
python -c "from vieo_slam_amd import build; build.build()" > /dev/null 2>&1
timeout 600 python -m pytest tests/test_orb_parity.py -m gpu -x -q 2>&1 | tail -2
