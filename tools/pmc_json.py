#!/usr/bin/env python3
"""rocprofv3 output directories -> the JSON summary bench.py reads (profiles/<round>_pmc.json).

    tools/pmc_json.py OUT.json "COMMAND" headline_pmc_dir... [--mix NAME pmc_or_stats_dir...]...

`kernels`: per kernel (template arguments stripped) and counter the mean per launch over the headline directories (the default
bench.py command: every launch is a step of config 2).  `mixes[NAME]`: one bench.py run per mix (--only-mix NAME: one small
headline step, then the mix's warm-up and timed steps): per kernel (FULL name, template arguments kept) the sum of every counter
over its launches, the launches, and from the --kernel-trace --stats pass of the same command its calls and average duration;
`steps` = launches of k_encode_subfilter in that run minus the one headline step.  bench.py turns these into per-step figures of a GROUP of kernels (the probe
stage = k_probe + k_probe_locus<...> + the item kernels; the walk = k_walk_fast* + k_walk_pairs).  The summary is keyed by the hash of
the device sources it was collected for, so that bench.py never quotes traffic measured on other kernels.
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

SKIP = ("k_idx", "k_cls", "k_fill", "k_flt_insert", "k_gr_insert", "k_mz_insert", "k_mz_fill", "k_grmz", "k_loc_count", "k_loc_scatter", "k_loc_place", "k_loc_verify",
        "k_gloc", "rocclr", "at::", "Cijk", "__amd_")


def short(name, keep_templates):
    k = name.split("(")[0].replace("void ", "").replace("dbtk::", "").strip()
    return k if keep_templates else re.sub(r"<.*", "", k)


def collect(dirs, keep_templates=False):
    acc = defaultdict(list)
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"], keep_templates)
                if any(s in k for s in SKIP):
                    continue
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    return acc


def stats(dirs):
    out = {}
    for d in dirs:
        for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Name"], True)
                if any(s in k for s in SKIP):
                    continue
                out[k] = dict(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]))
    return out


def main():
    out, cmd, rest = sys.argv[1], sys.argv[2], sys.argv[3:]
    groups, cur = {"": []}, ""
    i = 0
    while i < len(rest):
        if rest[i] == "--mix":
            cur = rest[i + 1]
            groups[cur] = []
            i += 2
            continue
        groups[cur].append(rest[i])
        i += 1
    doc = dict(kernel_source_hash=bench.kernel_source_hash(), command=cmd,
               note="rocprofv3 --pmc, one pass per counter set, no trace domains; FETCH_SIZE / WRITE_SIZE are in KB (gfx950: a wide "
                    "coalesced read stream is counted at half its bytes: bench.py adds the correction for K1's read stream)",
               kernels={}, mixes={})
    for (k, c), v in sorted(collect(groups[""]).items()):
        e = doc["kernels"].setdefault(k, {})
        name = c + "_KB" if c in ("FETCH_SIZE", "WRITE_SIZE") else c
        e[name] = sum(v) / len(v)
        e["launches"] = len(v)
    for mix, dirs in groups.items():
        if not mix:
            continue
        m = doc["mixes"].setdefault(mix, dict(kernels={}, steps=0))
        for (k, c), v in sorted(collect(dirs, True).items()):
            e = m["kernels"].setdefault(k, {})
            name = c + "_KB" if c in ("FETCH_SIZE", "WRITE_SIZE") else c
            e[name + ":sum"] = sum(v)
            e["launches"] = len(v)
        for k, s in stats(dirs).items():
            m["kernels"].setdefault(k, {}).update(calls=s["calls"], avg_ns=s["avg_ns"])
        # (the run is ONE headline step, then the mix's warm-up and timed steps, each one launch of k_encode_subfilter: the headline
        # step's launches of the same kernels stay in the sums — a 1.2-ms step against 20 - 35 ms ones)
        enc = m["kernels"].get("k_encode_subfilter", {})
        m["steps"] = max((enc.get("calls") or enc.get("launches") or 0) - 1, 0)
        m["headline_steps"] = 1
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps({k: (v if k != "mixes" else {m: dict(steps=x["steps"], kernels=len(x["kernels"])) for m, x in v.items()}) for k, v in doc.items() if k != "kernels"}, indent=1))


if __name__ == "__main__":
    main()
