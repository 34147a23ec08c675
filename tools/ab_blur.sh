#!/bin/bash
# k_blur with and without its arithmetic (-DVIEO_BLUR_AB=1: WRONG results, timing only): what the load -> LDS -> store
# skeleton of the kernel costs by itself.  Rebuilds orb_extractor.hip on the GPU box, restores the normal build.
cd $GRAFT_REPO_ROOT
for ab in 1 0; do
  touch vieo_slam_amd/csrc/orb_extractor.hip
  VIEO_EXTRA_HIPCC_FLAGS="-DVIEO_BLUR_AB=$ab" python -c "from vieo_slam_amd import build; build.build()" > /dev/null 2>&1
  echo "AB=$ab $(python tools/run_extract.py 1024 5 2>&1 | head -1)"
done
