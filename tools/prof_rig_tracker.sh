#!/bin/bash
# rocprofv3 kernel trace of the one-call rig tracker: $1 = tag, summaries under gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r4}
export PYTHONPATH=$R
for cfg in "kb8 4 1500" "radtan 2 1200"; do
  set -- $cfg
  name=rig_$1_$2
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$name -o out -- python $R/tools/run_rig_tracker.py $1 $2 $3 30 > $R/gpurun_out/prof_$name.log 2>&1
  python $R/tools/rocpd_summary.py $(find $R/gpurun_out/prof_$name -name "*.db" | head -1) $R/gpurun_out/${TAG}_$name.md > /dev/null
  tail -1 $R/gpurun_out/prof_$name.log
done
