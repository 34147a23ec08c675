#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_tracker.py -m gpu -x -q > /tmp/t.log 2>&1; grep -E "passed|failed|FAILED|Error|assert" /tmp/t.log | tail -6
