#!/bin/bash
# kernel trace of the r3 bench step's local-BA share ALONE on the GPU: 205 windows in lock step (154 ordinary: 10 local +
# 40 fixed key frames; 51 bLarge: 25 local), one host thread
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/tools/lba_r3_batch.py /tmp/lba3.py
for which in mixed small large; do
  n=205; [ $which = small ] && n=154; [ $which = large ] && n=51
  python /tmp/lba3.py $R $n $which 2>&1 | grep -v "^$" | tail -2
done
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_lba_r3 -o out -- python /tmp/lba3.py $R 205 mixed > $R/gpurun_out/prof_lba_r3.log 2>&1
python $R/tools/rocpd_summary.py $(find $R/gpurun_out/prof_lba_r3 -name "*.db" | head -1) $R/gpurun_out/r3_lba_alone_205_windows.md | head -24
