#!/bin/bash
# after a change to pose_opt_vio.hip: its parity tests, the phase probe (if the probe library is there), three timed replays
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
timeout 1200 python -m pytest tests/test_pose_opt_vio_parity.py tests/test_pose_opt_vio_kat.py tests/test_tracker.py -m gpu -x -q > /tmp/t.log 2>&1; grep -E "passed|failed|FAILED|Error" /tmp/t.log | tail -6
if [ -f vieo_slam_amd/libvieo_hot_probe.so ]; then VIEO_LIB_PATH=vieo_slam_amd/libvieo_hot_probe.so python tools/probe_pose.py 2>&1 | tail -24; fi
python tools/write_sequence.py /tmp/seq.vseq --frames 400 > /dev/null
for rep in 1 2 3; do timeout 120 ./examples/replay_main /tmp/seq.vseq --prefetch 1 --warmup 16 --quiet --lba-lag 8 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('replay_main ms_per_frame', r['ms_per_frame'], 'track call', r['ms_track_call'], 'gpu', r['ms_track_gpu'], 'lba', r['ms_per_local_ba'])"; done
