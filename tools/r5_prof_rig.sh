#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/run_rig_sequence.py kb8 4 1500 40 5 > $R/gpurun_out/r5e_rig_seq.txt 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_rig_seq -o out -- python $R/tools/run_rig_sequence.py kb8 4 1500 40 5 > $R/gpurun_out/r5e_rig_seq_prof.log 2>&1
python $R/tools/rocpd_summary.py $(find $R/gpurun_out/prof_rig_seq -name "*.db" | head -1) > $R/gpurun_out/r5e_rig_seq_kernels.md 2>&1
head -30 $R/gpurun_out/r5e_rig_seq.txt; head -30 $R/gpurun_out/r5e_rig_seq_kernels.md
