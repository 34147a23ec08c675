#!/bin/bash
# the key-frame reference's pre-integration run ahead (vieo_track_input.next_ref_bias): C++ replays with frame pipelining on
# (uses it) and off, trajectories must be byte-identical; the tracker tests
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_tracker.py tests/test_replay.py -m "gpu and not slow" -x -q > /tmp/t.log 2>&1; grep -E "passed|failed|FAILED|Error" /tmp/t.log | tail -4
python tools/write_sequence.py /tmp/seq.vseq --frames 400 > /dev/null
for rep in 1 2 3; do
  ./examples/replay_main /tmp/seq.vseq /tmp/t1.bin --warmup 16 --quiet --lba-lag 8 --prefetch 1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('prefetch=1 ms_per_frame', r['ms_per_frame'], 'track call', r['ms_track_call'], 'gpu', r['ms_track_gpu'], 'p99', r['ms_per_frame_p99'], 'lba', r['ms_per_local_ba'])"
done
./examples/replay_main /tmp/seq.vseq /tmp/t0.bin --warmup 16 --quiet --lba-lag 8 --prefetch 0 | cut -c1-120
cmp /tmp/t0.bin /tmp/t1.bin && echo "trajectories identical with and without pipelining"
python tools/write_sequence.py /tmp/rig4.vseq --rig kb8 --cams 4 --features 1500 --seed 5 --frames 100 > /dev/null
./examples/replay_modes /tmp/rig4.vseq /tmp/r1.bin --warmup 14 --quiet --lba-lag 8 --prefetch 1 | cut -c1-200
./examples/replay_modes /tmp/rig4.vseq /tmp/r0.bin --warmup 14 --quiet --lba-lag 8 --prefetch 0 | cut -c1-200
cmp /tmp/r0.bin /tmp/r1.bin && echo "rig trajectories identical with and without pipelining"
