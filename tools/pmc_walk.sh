#!/bin/bash
# Instruction-mix / stall counters of the walk kernel (diagnostic)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
CMD="python3 tools/walk_bench.py 20000 1000000"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcw_$i -- $CMD > /dev/null 2>&1
done
python3 tools/pmc_summary.py gpurun_out/pmcw_* | grep -E "k_walk_pairs|kernel,"
rm -rf gpurun_out/pmcw_*
