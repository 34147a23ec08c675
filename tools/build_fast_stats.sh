#!/bin/bash
# libvieo_hot_faststats.so = the library with k_fast's per-cell counters compiled in (-DVIEO_FAST_STATS); use it through
# VIEO_LIB_PATH (tools/fast_stats.py)
set -e
cd "$(dirname "$0")/.."
python vieo_slam_amd/build.py > /dev/null
O=vieo_slam_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function \
  -Wno-unused-result -DVIEO_FAST_STATS -c vieo_slam_amd/csrc/orb_extractor.hip -o $O/orb_extractor_stats.o
objs=$(ls $O/*.hip.o | grep -v orb_extractor.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vieo_slam_amd/libvieo_hot_faststats.so $objs $O/orb_extractor_stats.o
echo vieo_slam_amd/libvieo_hot_faststats.so
