#!/bin/bash
# Instruction-mix / stall counters of the probe kernel on the all-hit mix (diagnostic): tools/pmc_k2.sh [env assignments...]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
CMD="python3 bench.py --steps 1 --warmup 0 --ref-reads 0 --cpu-seconds 0 --no-e2e --no-walk --mix-steps 2"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmck2_$i -- $CMD > /dev/null 2>&1
done
python3 tools/pmc_summary.py gpurun_out/pmck2_* | grep -E "k_probe|kernel,"
rm -rf gpurun_out/pmck2_*
