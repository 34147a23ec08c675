#!/bin/bash
# Pinning recipe, one command.  Where OpenCV >= 4.5 (+ Eigen3 / Sophus for the BA
# hooks) exist: build the reference's own ORBextractor (and its g2o edge types) from VIEO_REFERENCE_ROOT into
# oracle/_ref/, run it and the restated oracle on the seeded cases, print the first differing stage, and -- with
# --write -- regenerate tests/golden/orb_golden.npz from the REFERENCE's outputs (the manifest then says "reference").
# In the authoring image and on the GPU box this stops at the cmake step with "OpenCV >= 4.5 not found".
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
REF="${VIEO_REFERENCE_ROOT:-/root/reference}"
WITH_BA="${VIEO_REF_WITH_BA:-OFF}"
mkdir -p "$ROOT/oracle/_ref/build"
cmake -S "$ROOT/oracle/ref_build" -B "$ROOT/oracle/_ref/build" -DVIEO_REFERENCE_ROOT="$REF" -DVIEO_REF_WITH_BA="$WITH_BA"
cmake --build "$ROOT/oracle/_ref/build" -j "$(nproc)"
make -s -C "$ROOT/oracle"
cd "$ROOT"
python tools/compare_with_reference.py "$@"
