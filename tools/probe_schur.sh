#!/bin/bash
# s_memtime-style probes inside k_lba_schur (-DVIEO_SCHUR_PROBE): cycles per phase of the chunk loop, summed over a
# workgroup's chunks, for a few workgroups of an ordinary (window 0) and a bLarge (window 3) window; first launch of the
# mixed 205-window batch.  Restores the normal build.
cd $GRAFT_REPO_ROOT
touch vieo_slam_amd/csrc/lba.hip
VIEO_EXTRA_HIPCC_FLAGS="-DVIEO_SCHUR_PROBE" python -c "from vieo_slam_amd import build; build.build()" > /dev/null 2>&1
python tools/lba_r3_batch.py $GRAFT_REPO_ROOT 205 mixed 2>&1 | grep "schur probe" | head -12
touch vieo_slam_amd/csrc/lba.hip
python -c "from vieo_slam_amd import build; build.build()" > /dev/null 2>&1
