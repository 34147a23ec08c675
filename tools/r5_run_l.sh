#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_orb_parity.py -m gpu -x -q 2>&1 | tail -5
for f in 1 0; do for B in 4096 2; do echo "fused=$f B=$B"; VIEO_ORB_FUSED=$f timeout 300 python tools/run_extract.py $B 4; done; done
