#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests/test_replay_modes.py tests/test_tracker_rig.py tests/test_tracker.py -m gpu -x -q -s > /tmp/t.log 2>&1; grep -E "passed|failed|FAILED|Error|rig replay|vision-only replay" /tmp/t.log | tail -12
