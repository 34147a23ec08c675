#!/bin/bash
# k_pose_opt_vio<256> with s_setprio (-DVIEO_POSE_SETPRIO=n) beside the local BA: the sequential replay's ms per frame
cd $GRAFT_REPO_ROOT
python tools/write_sequence.py /tmp/seq.vseq --frames 400 > /dev/null
run() { for rep in 1 2 3; do timeout 120 ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --lba-lag 8 --prefetch 1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$1', r['ms_per_frame'], r['ms_per_frame_last_200'], r['ms_track_call'], r['ms_track_gpu'], r['ms_per_local_ba'])"; done; }
run base
for n in 3 1; do
  touch vieo_slam_amd/csrc/pose_opt_vio.hip
  VIEO_EXTRA_HIPCC_FLAGS="-DVIEO_POSE_SETPRIO=$n" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  run setprio$n
done
touch vieo_slam_amd/csrc/pose_opt_vio.hip
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
