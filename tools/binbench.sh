#!/bin/bash
# one bench line per setting: name:ENV...:bench args (diagnostic)
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; envs=${rest%%:*}; args=${rest#*:}; [ "$args" = "$rest" ] && args=""
  env $envs python bench.py --cpu-seconds 0 --steps 20 $args > gpurun_out/bb_$name.json 2>/dev/null
  python - "$name" <<PY
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open("gpurun_out/bb_%s.json"%n).read().strip().splitlines()[-1])
    print(n, round(d["value"]/1e9,3), round(d["ms_per_step"],3), {k.replace("k_encode_subfilter: ",""):round(v["avg_ms"],4) for k,v in d["roofline"]["kernels"].items()})
except Exception as e: print(n, "ERR", e)
PY
done
