#!/bin/bash
# sweeps of the extractor's tiling constants (build macros), stage times per 1024 images; crc must not change
cd $GRAFT_REPO_ROOT
for f in "-DVIEO_RESIZE_ROWS=16" "-DVIEO_RESIZE_ROWS=24" "-DVIEO_XCD_RUN=16" "-DVIEO_XCD_RUN=64" "-DVIEO_XCD_RUN=128" ""; do
  touch vieo_slam_amd/csrc/orb_extractor.hip
  VIEO_EXTRA_HIPCC_FLAGS="$f" python -c "from vieo_slam_amd import build; build.build()" > /dev/null 2>&1
  echo "[$f] $(python tools/run_extract.py 1024 6 2>&1 | head -2 | tr '\n' ' ')"
done
