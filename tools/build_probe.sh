#!/bin/bash
# libvieo_hot_probe.so = the library with the s_memtime probes of k_pose_opt_vio compiled in (-DVIEO_POSE_PROBE);
# use it through VIEO_LIB_PATH (tools/probe_pose.py)
set -e
cd "$(dirname "$0")/.."
python vieo_slam_amd/build.py > /dev/null
O=vieo_slam_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function \
  -Wno-unused-result -DVIEO_POSE_PROBE -c vieo_slam_amd/csrc/pose_opt_vio.hip -o $O/pose_opt_vio_probe.o
objs=$(ls $O/*.hip.o | grep -v pose_opt_vio.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vieo_slam_amd/libvieo_hot_probe.so $objs $O/pose_opt_vio_probe.o
echo vieo_slam_amd/libvieo_hot_probe.so
