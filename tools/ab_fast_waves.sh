#!/bin/bash
# k_fast with 1 / 2 / 4 cells (wavefronts) per workgroup: -DVIEO_FAST_WAVES=n, extractor stage times per 1024 images
cd $GRAFT_REPO_ROOT
for n in 2 4 1; do
  touch vieo_slam_amd/csrc/orb_extractor.hip
  VIEO_EXTRA_HIPCC_FLAGS="-DVIEO_FAST_WAVES=$n" python -c "from vieo_slam_amd import build; build.build()" > /dev/null 2>&1
  echo "WAVES=$n $(python tools/run_extract.py 1024 6 2>&1 | head -2 | tr '\n' ' ')"
done
