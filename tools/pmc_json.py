#!/usr/bin/env python3
"""rocprofv3 --pmc output directories -> the JSON summary bench.py reads (profiles/<round>_pmc.json).

    tools/pmc_json.py OUT.json "COMMAND" headline_dir... [--mix mix_dir...]

Per kernel and counter: the mean per launch over the headline directories (`kernels`), and the largest launch over the
mix directories (`mix_all_hit`: the all-hit launches are the big ones of that command).  The summary is keyed by the hash
of the device sources it was collected for, so that bench.py never quotes traffic measured on other kernels.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402  (kernel_source_hash)

SKIP = ("k_idx", "k_cls", "k_fill", "k_flt_insert", "k_gr_insert", "k_mz_insert", "rocclr", "at::", "Cijk")


def collect(dirs):
    acc = defaultdict(list)
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dbtk::", "")
                k = re.sub(r"<.*", "", k)
                if any(s in k for s in SKIP):
                    continue
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    out, cmd, rest = sys.argv[1], sys.argv[2], sys.argv[3:]
    mix = []
    if "--mix" in rest:
        i = rest.index("--mix")
        rest, mix = rest[:i], rest[i + 1:]
    doc = dict(kernel_source_hash=bench.kernel_source_hash(), command=cmd,
               note="rocprofv3 --pmc, one pass per counter set, no trace domains; FETCH_SIZE / WRITE_SIZE are in KB (gfx950: a wide "
                    "coalesced read stream is counted at half its bytes: bench.py adds the correction for K1's read stream)",
               kernels={}, mix_all_hit={})
    for (k, c), v in sorted(collect(rest).items()):
        e = doc["kernels"].setdefault(k, {})
        name = c + "_KB" if c in ("FETCH_SIZE", "WRITE_SIZE") else c
        e[name] = sum(v) / len(v)
        e["launches"] = len(v)
    for (k, c), v in sorted(collect(mix).items()):
        e = doc["mix_all_hit"].setdefault(k, {})
        name = c + "_KB" if c in ("FETCH_SIZE", "WRITE_SIZE") else c
        e[name] = max(v)
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
