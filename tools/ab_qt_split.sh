#!/bin/bash
# k_quadtree as several launches over groups of levels (-DVIEO_QT_GROUPS={..}; {64} = one launch sized for level 0)
cd $GRAFT_REPO_ROOT
for n in "{64}" "{1,64}" "{1,3,64}" "{1,2,4,64}" "{1,2,3,4,5,6,7,64}"; do
  touch vieo_slam_amd/csrc/orb_extractor.hip
  VIEO_EXTRA_HIPCC_FLAGS="-DVIEO_QT_GROUPS=$n" python -c "from vieo_slam_amd import build; build.build()" > /dev/null 2>&1
  echo "GROUPS=$n $(python tools/run_extract.py 1024 6 2>&1 | head -2 | tr '\n' ' ')"
done
