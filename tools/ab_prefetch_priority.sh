#!/bin/bash
# frame pipelining with the prefetch stream at normal / low / high priority (VIEO_PREFETCH_PRIORITY) + a kernel timeline
cd "$(dirname "$0")/.."
python tools/write_sequence.py /tmp/seq.vseq --frames 200 > /dev/null
q() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(sys.argv[1], r['prefetch'], r['ms_per_frame'], r['ms_track_call'], r['ms_track_gpu'], r['ms_per_local_ba'])" "$1"; }
for pf in 0 1; do for rep in 1 2; do timeout 120 ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --lba-lag 6 --prefetch $pf | q prio0; done; done
for rep in 1 2; do VIEO_PREFETCH_PRIORITY=-1 timeout 120 ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --lba-lag 6 --prefetch 1 | q prio_low; done
for rep in 1 2; do VIEO_PREFETCH_PRIORITY=1 timeout 120 ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --lba-lag 6 --prefetch 1 | q prio_high; done
for rep in 1 2; do timeout 120 ./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --prefetch 1 | q inline_lba; done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_pref -o out -- $GRAFT_REPO_ROOT/examples/replay_main /tmp/seq.vseq --quiet --lba-lag 6 --prefetch 1 --frames 30 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $(find $GRAFT_REPO_ROOT/gpurun_out/prof_pref -name "*.db" | head -1) k_track_adopt 2 60 > $GRAFT_REPO_ROOT/gpurun_out/r5i_pref_timeline.txt 2>&1
