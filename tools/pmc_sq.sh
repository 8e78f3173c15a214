#!/bin/bash
# Issue / wait counters of the probe kernels on the all-hit mix (diagnostic):  tools/pmc_sq.sh <tag> <mix-reads> [ENV=V ...]  ->  gpurun_out/<tag>_pmc_sq.txt
tag=${1:-sq}; reads=${2:-4000000}; shift; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export DBTK_LANES=1
for kv in "$@"; do export "$kv"; done
CMD="python3 tools/probe_bench.py --child --reads 0 --mix-reads $reads --steps 2 --rounds 1"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcs_${tag}_$i -- $CMD > gpurun_out/pmcs_${tag}_$i.log 2>&1
done
python3 tools/pmc_summary.py gpurun_out/pmcs_${tag}_* 2>/dev/null | grep -E "k_probe|kernel," > gpurun_out/${tag}_pmc_sq.txt
cat gpurun_out/${tag}_pmc_sq.txt
rm -rf gpurun_out/pmcs_${tag}_*
