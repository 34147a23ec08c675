#!/bin/bash
# k_blur tile height: -DVIEO_BLUR_TH=n, extractor stage times per 1024 images (results must stay bit-equal: same crc)
cd $GRAFT_REPO_ROOT
for n in 32 64 128 96; do
  touch vieo_slam_amd/csrc/orb_extractor.hip
  VIEO_EXTRA_HIPCC_FLAGS="-DVIEO_BLUR_TH=$n" python -c "from vieo_slam_amd import build; build.build()" > /dev/null 2>&1
  echo "TH=$n $(python tools/run_extract.py 1024 6 2>&1 | head -2 | tr '\n' ' ')"
done
