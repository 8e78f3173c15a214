#!/bin/bash
# The committed evidence of a round, in one GPU call:  tools/profile_round.sh r02  ->  gpurun_out/<tag>_*  (copy to profiles/)
#   1. PMC passes (HBM traffic, requests) on the headline command and on the all-hit mix -> profiles/<tag>_pmc.json (keyed by
#      the hash of the device sources; bench.py quotes `traffic` from it only when the hash matches)
#   2. rocprofv3 --kernel-trace --stats of the headline command (same launches as the bench line's timed region + warm-up)
#   3. the default bench.py line (with traffic from 1.), and the two-lane variant
# tools/profile_round.sh <tag> bench  runs step 3 alone (the profile steps take ~25 min of box time, the bench ~4)
tag=${1:-rXX}
what=${2:-all}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
HEAD="python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --ref-reads 0 --mix-reads 0 --no-e2e --sustain-seconds 0 --no-extra-lanes"
MIX="python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --ref-reads 0 --no-e2e --no-walk --mix-steps 2 --sustain-seconds 0 --no-extra-lanes"
if [ "$what" != bench ]; then
i=0
for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"; do
  i=$((i+1))
  [ $i -le 3 ] && rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcH${i}_$tag -- $HEAD > /dev/null 2>&1
  [ $i -ne 3 ] && rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcM${i}_$tag -- $MIX > /dev/null 2>&1
done
python3 tools/pmc_json.py gpurun_out/${tag}_pmc.json "rocprofv3 --pmc <set> -- $HEAD  |  mix: $MIX" gpurun_out/pmcH?_$tag --mix gpurun_out/pmcM?_$tag > /dev/null
cp gpurun_out/${tag}_pmc.json profiles/${tag}_pmc.json
STATS="python3 bench.py --cpu-seconds 0 --ref-reads 0 --mix-reads 0 --no-e2e --sustain-seconds 0 --no-extra-lanes"  # (one lane: a kernel's duration must not include its neighbour's)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- $STATS > gpurun_out/${tag}_stats_bench.json 2> /dev/null
cp $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profM_$tag -- $MIX --mix-steps 10 > /dev/null 2>&1
cp $(find gpurun_out/profM_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_mixes_kernel_stats.csv
rm -rf gpurun_out/prof_$tag gpurun_out/profM_$tag gpurun_out/pmc??_$tag
head -8 gpurun_out/${tag}_kernel_stats.csv; head -8 gpurun_out/${tag}_mixes_kernel_stats.csv
fi
[ "$what" = prof ] && exit 0
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.log
tail -14 gpurun_out/${tag}_bench.log
python3 - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["traffic"], d["roofline"].get("traffic_stale"))
PY
