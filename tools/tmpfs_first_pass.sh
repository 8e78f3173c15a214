#!/bin/bash
# Diagnostic: how fast can ANY program read a file another process has just written into /dev/shm — the first time, and again?
# (bench.py's CLI legs read such a file; the command line's first pass over it runs at about half the rate of a later one: is that the
# command line, or the box?)   tools/tmpfs_first_pass.sh [GB=8] [readers=8]
gb=${1:-8}; nr=${2:-8}
f=/dev/shm/dbtk_fp_$$
python3 - <<PY
import numpy as np, time
t0 = time.time()
b = np.random.default_rng(1).integers(65, 85, 1 << 28, dtype=np.uint8)
with open("$f", "wb") as out:
    for _ in range($gb * 4):
        out.write(b)
print(f"wrote $gb GB in {time.time() - t0:.1f}s")
PY
sz=$(stat -c %s $f); per=$((sz / nr / 1048576))
for pass in 1 2 3; do
  t0=$(date +%s.%N)
  for i in $(seq 0 $((nr - 1))); do dd if=$f of=/dev/null bs=32M skip=$((i * per / 32)) count=$((per / 32)) status=none & done
  wait
  t1=$(date +%s.%N)
  echo "pass $pass: $nr readers, $(python3 -c "print(round($sz / 1e9 / ($t1 - $t0), 1))") GB/s"
done
rm -f $f
