#!/bin/bash
# round 5, first GPU pass: the new tests first, then the whole GPU suite, then the drop-in replay's timings
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_resident_frame.py tests/test_dropin_replay.py -x -q -m gpu -s > gpurun_out/r5a_new_tests.log 2>&1
echo "new tests rc=$?" >> gpurun_out/r5a_new_tests.log
python -m pytest tests/test_tracker.py tests/test_tracker_rig.py tests/test_pose_opt_vio_parity.py -x -q -m gpu -s > gpurun_out/r5a_harden_tests.log 2>&1
echo "harden tests rc=$?" >> gpurun_out/r5a_harden_tests.log
python tools/write_sequence.py /tmp/seq.vseq --frames 100 > /dev/null
for res in 1 0; do for lag in 0 6; do
  for rep in 1 2 3; do ./examples/dropin_replay /tmp/seq.vseq --warmup 12 --quiet --resident $res --lba-lag $lag; done
done; done > gpurun_out/r5a_dropin.log 2>&1
for rep in 1 2 3; do ./examples/replay_main /tmp/seq.vseq --warmup 12 --quiet --lba-lag 6; done > gpurun_out/r5a_replay_main.log 2>&1
python -m pytest tests -x -q -m gpu > gpurun_out/r5a_gpu_suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/r5a_gpu_suite.log
tail -3 gpurun_out/r5a_new_tests.log gpurun_out/r5a_harden_tests.log gpurun_out/r5a_gpu_suite.log
cat gpurun_out/r5a_dropin.log
