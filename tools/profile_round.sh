#!/bin/bash
# The committed evidence of a round, in one GPU call:  tools/profile_round.sh r04a  ->  gpurun_out/<tag>_*  (copy to profiles/)
#   1. PMC passes (HBM traffic, requests, instructions) on the headline command, and on one command per extra mix (all-hit, walk at
#      k = 21, walk at k = 25: bench.py --only-mix NAME) -> profiles/<tag>_pmc.json (keyed by the hash of the device sources; bench.py
#      quotes `traffic` from it only when the hash matches), with the rocprofv3 --kernel-trace --stats durations of the same mix commands
#   2. rocprofv3 --kernel-trace --stats of the headline command (same launches as the bench line's timed region + warm-up)
#   3. the default bench.py line (with traffic from 1.)
# tools/profile_round.sh <tag> bench  runs step 3 alone;  <tag> prof  steps 1 and 2 alone
tag=${1:-rXX}
what=${2:-all}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
HEAD="python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --ref-reads 0 --mix-reads 0 --no-e2e --sustain-seconds 0 --no-extra-lanes"
MIX="python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --ref-reads 0 --no-e2e --sustain-seconds 0 --no-extra-lanes --mix-steps 3 --k25-parity-pairs 0"
SETS=("FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES")
if [ "$what" != bench ]; then
i=0
for set in "${SETS[@]}" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcH${i}_$tag -- $HEAD > /dev/null 2>&1
done
args=()
for mix in all_hit walk k25; do
  i=0
  for set in "${SETS[@]}" "WRITE_SIZE"; do  # (the probe stage's hit rows are WRITES: the fused kernel's whole point)
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcM_${mix}_${i}_$tag -- $MIX --only-mix $mix > /dev/null 2>&1
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmcM_${mix}_s_$tag -- $MIX --only-mix $mix > /dev/null 2>&1
  cp $(find gpurun_out/pmcM_${mix}_s_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_mix_${mix}_kernel_stats.csv
  args+=(--mix $mix gpurun_out/pmcM_${mix}_?_$tag)
done
python3 tools/pmc_json.py gpurun_out/${tag}_pmc.json "rocprofv3 --pmc <set> -- $HEAD  |  mixes: $MIX --only-mix <mix>" gpurun_out/pmcH?_$tag "${args[@]}" > /dev/null
cp gpurun_out/${tag}_pmc.json profiles/${tag}_pmc.json
STATS="python3 bench.py --cpu-seconds 0 --ref-reads 0 --mix-reads 0 --no-e2e --sustain-seconds 0 --no-extra-lanes"  # (one lane: a kernel's duration must not include its neighbour's)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- $STATS > gpurun_out/${tag}_stats_bench_line.json 2> /dev/null
cp $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_$tag gpurun_out/pmcH?_$tag gpurun_out/pmcM_*_$tag
head -8 gpurun_out/${tag}_kernel_stats.csv; for mix in all_hit walk k25; do echo "== $mix"; grep -E "walk|probe|pair|encode" gpurun_out/${tag}_mix_${mix}_kernel_stats.csv | cut -d, -f1-4 | head -12; done
fi
[ "$what" = prof ] && exit 0
# the ONE line the driver parses -> <tag>_bench_line.json; everything else (per-kernel tables, mixes, CLI legs) -> <tag>_bench.json
DBTK_BENCH_DETAIL=gpurun_out/${tag}_bench.json python3 bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.log
tail -14 gpurun_out/${tag}_bench.log
python3 - <<PY
import json
line = open("gpurun_out/${tag}_bench_line.json").read().strip().splitlines()[-1]
print("bench line:", len(line), "characters")
c = json.loads(line)
print({k: c[k] for k in ("value", "ms_per_step")}, c["roofline"], c.get("cpu_baseline", {}).get("value"))
d = json.load(open("gpurun_out/${tag}_bench.json"))
for k, v in (d.get("mixes") or {}).items():
    if isinstance(v, dict) and "ms_per_step" in v:
        print(k, round(v["ms_per_step"], 2), "ms/step", round(v["value"] / 1e6, 1), "M reads/s", {n: round(x["avg_ms"], 2) for n, x in v["roofline"]["kernels"].items() if ":" not in n})
PY
echo "produced (copy these to profiles/, nothing else): gpurun_out/${tag}_pmc.json gpurun_out/${tag}_kernel_stats.csv gpurun_out/${tag}_mix_*_kernel_stats.csv gpurun_out/${tag}_bench.json gpurun_out/${tag}_bench_line.json gpurun_out/${tag}_bench.log"
