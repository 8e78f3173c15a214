#!/usr/bin/env python3
"""Per-kernel mean / max per launch of the counters in rocprofv3 --pmc output directories (counter_collection.csv)."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            for pre in ("void ", "dbtk::"):
                k = k.replace(pre, "")
            acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
print("kernel,counter,launches,mean_per_launch,max_per_launch")
for (k, c), v in sorted(acc.items()):
    print(f"{k},{c},{len(v)},{sum(v) / len(v):.1f},{max(v):.1f}")
