#!/bin/bash
# the walk mixes of bench.py alone (k = 21 at --mix-reads, k = 25 at --k25-reads), one lane:  tools/walk_mix.sh [ENV=V ...]
cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
DBTK_BENCH_DETAIL=gpurun_out/walk_mix_detail.json python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --ref-reads 0 --no-e2e --sustain-seconds 0 --no-extra-lanes --mix-steps 5 --k25-parity-pairs 20000 2>gpurun_out/walk_mix.err >/dev/null; python3 -c "
import sys, json
d = json.load(open('gpurun_out/walk_mix_detail.json'))
for k, v in d['mixes'].items():
    if isinstance(v, dict) and 'ms_per_step' in v:
        print(k, round(v['ms_per_step'], 2), 'ms/step', round(v['value'] / 1e6, 1), 'M reads/s', {n: round(x['avg_ms'], 2) for n, x in v['roofline']['kernels'].items() if ':' not in n}, {k_: v_ for k_, v_ in (v.get('path') or {}).items() if k_ in ('walk_items', 'walk_pairs', 'walk_rest', 'probe_rest', 'fused_done')})
"
tail -3 gpurun_out/walk_mix.err
