#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests/test_lba.py tests/test_lba_vio.py -m gpu -q > /tmp/t.log 2>&1; grep -E "passed|failed|FAILED|Error" /tmp/t.log | tail -12
