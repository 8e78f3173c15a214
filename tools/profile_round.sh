#!/bin/bash
# The committed evidence of a round: tools/profile_round.sh r01h  ->  gpurun_out/<tag>_*  (copy to profiles/)
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 bench.py --extra-lanes > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.log
python3 bench.py --lanes 2 --cpu-seconds 0 > gpurun_out/${tag}_lanes2_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --cpu-seconds 0 --parity-pairs 0 > /dev/null 2>&1
cp $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc1_$tag -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc2_$tag -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/pmc3_$tag -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc1_$tag gpurun_out/pmc2_$tag gpurun_out/pmc3_$tag | grep -v "k_idx\|k_cls\|k_fill\|k_flt_insert\|rocclr" > gpurun_out/${tag}_pmc.csv
rm -rf gpurun_out/prof_$tag gpurun_out/pmc1_$tag gpurun_out/pmc2_$tag gpurun_out/pmc3_$tag
head -8 gpurun_out/${tag}_kernel_stats.csv; cat gpurun_out/${tag}_pmc.csv | head -40; tail -2 gpurun_out/${tag}_bench.log
