#!/bin/bash
# Counter passes over the C++ sequence replay (60 frames) for k_pose_opt_vio<256>: instruction fetch (the kernel is 207 KB of
# code, one LM iteration touches ~84 KB of it: profiles/r6_pose_code_size.txt), issue and wait cycles.  Each pass is its own
# rocprofv3 --pmc run with --kernel-trace only (MI355X_MICROARCH.md).  Output: gpurun_out/$1_pmc_pose.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r6}
export PYTHONPATH=$R
python $R/tools/write_sequence.py /tmp/seq.vseq --frames 60 > /dev/null
run() {
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/pmc_pose_$name -o out -- $R/examples/replay_main /tmp/seq.vseq --warmup 12 --quiet --lba-lag 6 > $R/gpurun_out/pmc_pose_$name.log 2>&1
  python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pmc_pose_$name -name "*.db" | head -1) | grep -A12 "k_pose_opt_vio"
}
{
echo "# k_pose_opt_vio<256> in examples/replay_main (60 frames), averages per counter instance and launch; source sha16 $(sha256sum $R/vieo_slam_amd/csrc/pose_opt_vio.hip | cut -c1-16)"
run a SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run b SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run c SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
run d SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
} > $R/gpurun_out/${TAG}_pmc_pose.txt 2>&1
cat $R/gpurun_out/${TAG}_pmc_pose.txt
