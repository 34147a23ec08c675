#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_tracker.py tests/test_tracker_rig.py -m gpu -x -q 2>&1 | tail -5
python tools/write_sequence.py /tmp/seq.vseq --frames 200 > /dev/null
q() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(sys.argv[1], r['prefetch'], r['ms_per_frame'], r['ms_track_call'], r['ms_track_gpu'], r['ms_per_local_ba'])" "$1"; }
for pf in 0 1; do for rep in 1 2 3; do timeout 120 ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --lba-lag 6 --prefetch $pf | q lag6; done; done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_pref2 -o out -- $GRAFT_REPO_ROOT/examples/replay_main /tmp/seq.vseq --quiet --lba-lag 6 --prefetch 1 --frames 30 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $(find $GRAFT_REPO_ROOT/gpurun_out/prof_pref2 -name "*.db" | head -1) k_track_adopt 2 60 > $GRAFT_REPO_ROOT/gpurun_out/r5j_pref_timeline.txt 2>&1
