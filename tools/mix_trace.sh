#!/bin/bash
# Per-launch durations of the probe-stage kernels of one bench mix (diagnostic):  tools/mix_trace.sh [mix=all_hit] [ENV=V ...]
mix=${1:-all_hit}; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/mixtrace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/mixtrace -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --ref-reads 0 --no-e2e --sustain-seconds 0 --no-extra-lanes --mix-steps 5 --k25-parity-pairs 0 --only-mix $mix > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/mixtrace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"]
    if any(t in n for t in ("k_probe", "k_loc_", "k_encode", "k_surv_", "k_pair", "k_walk")):
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:10.3f} ms  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:8.3f} ms  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8s}  {n[:80]}")
PY
rm -rf gpurun_out/mixtrace
