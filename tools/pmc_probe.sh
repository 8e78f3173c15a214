#!/bin/bash
# Counters of the probe kernel on the all-hit mix (diagnostic):  tools/pmc_probe.sh <tag> [ENV=V ...]  ->  gpurun_out/<tag>_pmc_probe.txt
# One rocprofv3 --pmc pass per counter set, no trace domains; the program after `--` is python3 itself (no re-exec).
tag=${1:-pmc}; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export DBTK_LANES=1
for kv in "$@"; do export "$kv"; done
CMD="python3 tools/probe_bench.py --child --reads 0 --steps 2 --rounds 1"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" \
           "FETCH_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcp_${tag}_$i -- $CMD > gpurun_out/pmcp_${tag}_$i.log 2>&1
done
python3 tools/pmc_summary.py gpurun_out/pmcp_${tag}_* 2>/dev/null | grep -E "k_probe|kernel," > gpurun_out/${tag}_pmc_probe.txt
cat gpurun_out/${tag}_pmc_probe.txt
rm -rf gpurun_out/pmcp_${tag}_*
