#!/bin/bash
# after the chunked key-frame half of k_lba_build: the bundle-adjustment parity tests, the rig replays' tests, then the three
# C++ sequence replays' local-BA times (examples/replay_modes / replay_main, 60 frames, lag 8)
cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
timeout 1500 python -m pytest tests/test_lba.py tests/test_lba_vio.py tests/test_global_ba.py tests/test_global_ba_scale.py tests/test_golden_ba.py tests/test_sharding.py -m gpu -x -q > /tmp/t.log 2>&1; grep -E "passed|failed|FAILED|Error" /tmp/t.log | tail -6
timeout 1500 python -m pytest tests/test_replay_modes.py -m "gpu and not slow" -x -q > /tmp/t2.log 2>&1; grep -E "passed|failed|FAILED|Error" /tmp/t2.log | tail -6
python tools/write_sequence.py /tmp/rig4.vseq --rig kb8 --cams 4 --features 1500 --seed 5 --frames 100 > /dev/null
python tools/write_sequence.py /tmp/rig2.vseq --rig radtan --cams 2 --features 1200 --seed 3 --frames 100 > /dev/null
python tools/write_sequence.py /tmp/seq.vseq --frames 200 > /dev/null
for f in rig4 rig2; do ./examples/replay_modes /tmp/$f.vseq --warmup 14 --quiet --lba-lag 8 --prefetch 1 | cut -c1-330; done
./examples/replay_modes /tmp/seq.vseq --vision --warmup 14 --quiet --lba-lag 8 --prefetch 1 | cut -c1-330
./examples/replay_main /tmp/seq.vseq --warmup 16 --quiet --lba-lag 8 --prefetch 1 | cut -c1-330
