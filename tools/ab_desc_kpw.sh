#!/bin/bash
# k_describe with 1 / 2 / 3 / 4 key points per wavefront: -DVIEO_DESC_KPW=n, extractor stage times per 1024 images
cd $GRAFT_REPO_ROOT
for n in 1 2 3 4; do
  touch vieo_slam_amd/csrc/orb_extractor.hip
  VIEO_EXTRA_HIPCC_FLAGS="-DVIEO_DESC_KPW=$n" python -c "from vieo_slam_amd import build; build.build()" > /dev/null 2>&1
  echo "KPW=$n $(python tools/run_extract.py 1024 6 2>&1 | head -2 | tr '\n' ' ')"
done
touch vieo_slam_amd/csrc/orb_extractor.hip
python -c "from vieo_slam_amd import build; build.build()" > /dev/null 2>&1
