#!/bin/bash
# SQ counter passes over the batched extractor (k_fast is the kernel of interest); results under gpurun_out/pmc_fast_*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-512}
V=${2:-0}
export VIEO_FAST_VARIANT=$V
run() {  # name counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/pmc_fast_$name -o out -- python $R/tools/run_extract.py $B 3 > $R/gpurun_out/pmc_fast_$name.log 2>&1
  db=$(find $R/gpurun_out/pmc_fast_$name -name "*.db" | head -1)
  python $R/tools/rocpd_pmc.py $db 2>&1 | grep -A12 "k_fast" | head -14
}
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA
run c SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_ACTIVE_INST_FLAT
run d GRBM_GUI_ACTIVE GRBM_COUNT
