#!/bin/bash
# timing A/B of k_lba_schur's phases (VIEO_SCHUR_AB builds give WRONG results; never ship them): rebuilds lba.hip on the
# GPU box per variant and prints the duration of the FIRST k_lba_schur launch (all 205 windows of the mixed batch active)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for ab in ${@:-0 1 2 4 6 7}; do
  touch vieo_slam_amd/csrc/lba.hip
  VIEO_EXTRA_HIPCC_FLAGS="-DVIEO_SCHUR_AB=$ab" python -c "from vieo_slam_amd import build; build.build(force=False)" > /dev/null 2>&1
  rm -rf /tmp/ab_prof
  (cd /tmp && VIEO_EXTRA_HIPCC_FLAGS="-DVIEO_SCHUR_AB=$ab" rocprofv3 --kernel-trace -d /tmp/ab_prof -o out -- python $GRAFT_REPO_ROOT/tools/lba_r3_batch.py $GRAFT_REPO_ROOT 205 mixed > /dev/null 2>&1)
  echo "AB=$ab $(python tools/rocpd_summary.py $(find /tmp/ab_prof -name '*.db' | head -1) /tmp/ab.md | grep k_lba_schur)"
done
