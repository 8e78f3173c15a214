#!/bin/bash
# Counters of the probe kernel on the all-hit mix (diagnostic):  tools/pmc_probe.sh <tag> [ENV=V ...]  ->  gpurun_out/<tag>_pmc_probe.txt
# One rocprofv3 --pmc pass per counter set, no trace domains; the program after `--` is python3 itself (no re-exec).
tag=${1:-pmc}; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
CMD="python3 tools/probe_bench.py --child --reads 0 --steps 2 --rounds 1"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "FETCH_SIZE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcp_${tag}_$i -- $CMD > gpurun_out/pmcp_${tag}_$i.log 2>&1
done
python3 tools/pmc_summary.py gpurun_out/pmcp_${tag}_* 2>/dev/null | grep -E "k_probe|k_surv|k_pair|kernel," > gpurun_out/${tag}_pmc_probe.txt
cat gpurun_out/${tag}_pmc_probe.txt
tail -3 gpurun_out/pmcp_${tag}_5.log gpurun_out/pmcp_${tag}_6.log | cut -c1-300
rm -rf gpurun_out/pmcp_${tag}_*
