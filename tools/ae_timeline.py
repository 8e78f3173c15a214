#!/usr/bin/env python3
"""Diagnostic: where the GPU's time goes during the command line's -ae run (rocprofv3 --kernel-trace): busy time (union of kernel
intervals over all streams) against the batch loop's wall time, the kernels' sums, and the longest idle gaps.
    python tools/ae_timeline.py [npairs=5000000] [extra CLI args ...]"""
import csv
import glob
import importlib
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("danbing-tk_amd")


def main():
    npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
    extra = sys.argv[2:]
    d = tempfile.mkdtemp(prefix="dbtk_aetl_", dir="/dev/shm")
    try:
        syn = pkg.Synth(nloci=80000)
        syn.graph()
        syn.write_files(os.path.join(d, "pan"))
        seq, _ = syn.reads(npairs, hit_frac=1.0, seed=2)
        syn.write_fasta(seq, npairs, os.path.join(d, "reads_hit.fa"))
        del seq
        cli = os.path.join(ROOT, "danbing-tk_amd", "bin", "danbing-tk")
        base = [cli, "-k", "21", "-kf", "4", "1", "-cth", "45", "--v13-threading", "-gc", "85", "3", "-fa", "reads_hit.fa", "-qs", "pan", "-o", "w", "-ae", "--aln-gz", "w.aln.gz"] + extra
        subprocess.run(base, cwd=d, capture_output=True)  # (warm: page cache)
        r = subprocess.run(base, cwd=d, capture_output=True, text=True, env=dict(os.environ, DBTK_VERBOSE="2"))
        ls = [l for l in r.stderr.splitlines() if l.startswith("block ") or l.startswith("alloc") or l.startswith("aligner") or l.startswith("batch loop") or l.startswith("ingest:") or l.startswith("device reader") or l.startswith("timeline")]
        print("\n".join(ls[:400]))
        noemit = [a for a in base if a not in ("-ae", "--aln-gz", "w.aln.gz")]
        r = subprocess.run(noemit, cwd=d, capture_output=True, text=True, env=dict(os.environ, DBTK_VERBOSE="1"))
        print("no emit:", "\n".join(l for l in r.stderr.splitlines() if l.startswith("ingest:") or l.startswith("device reader") or l.startswith("timeline")))
        for ch in [x for x in os.environ.get("AE_CHUNKS", "").split(",") if x]:  # DBTK_INGEST_CHUNK sweep (MB), with and without -ae
            for cmd, tag in ((base, "-ae"), (noemit, "no emit")):
                r = subprocess.run(cmd, cwd=d, capture_output=True, text=True, env=dict(os.environ, DBTK_INGEST_CHUNK=str(int(ch) << 20)))
                print("chunk", ch, "MB", tag, [l for l in r.stderr.splitlines() if l.startswith("ingest:")][0][:62])
        for n in [x for x in os.environ.get("AE_SWEEP", "").split(",") if x]:
            for rep in range(2):
                r = subprocess.run(base + ["--aln-aligners", n], cwd=d, capture_output=True, text=True)
                print("aligners", n, [l for l in r.stderr.splitlines() if l.startswith("ingest:")][0][:60])
        if os.environ.get("AE_NOPROF"):
            return
        pd = os.path.join(d, "prof")
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", pd, "--"] + base, cwd=d, capture_output=True, text=True, env=dict(os.environ, TMPDIR="/tmp", DBTK_VERBOSE="1"))
        print("\n".join(l for l in r.stderr.splitlines() if not l.startswith("Buffered") and not l.startswith("[rocprof")))
        f = glob.glob(pd + "/**/*kernel_trace.csv", recursive=True)[0]
        rows = sorted(csv.DictReader(open(f)), key=lambda x: int(x["Start_Timestamp"]))
        # the batch loop = from the first k_ing_ kernel to the last kernel
        ing = [i for i, x in enumerate(rows) if x["Kernel_Name"].startswith("k_ing_")]
        rows = rows[ing[0]:]
        t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(x["End_Timestamp"]) for x in rows)
        busy, cur_s, cur_e = 0, None, None
        gaps = []
        prev_name = ""
        for x in rows:
            s, e = int(x["Start_Timestamp"]), int(x["End_Timestamp"])
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                    gaps.append((s - cur_e, round((cur_e - t0) / 1e6, 1), prev_name, x["Kernel_Name"].split("(")[0][-28:]))
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
            if e >= cur_e:
                prev_name = x["Kernel_Name"].split("(")[0][-28:]
        busy += cur_e - cur_s
        sums = {}
        for x in rows:
            n = x["Kernel_Name"].split("(")[0][:36]
            sums[n] = sums.get(n, 0) + int(x["End_Timestamp"]) - int(x["Start_Timestamp"])
        print(f"span {(t1 - t0) / 1e6:.1f} ms, GPU busy (union over streams) {busy / 1e6:.1f} ms, sum of kernel durations {sum(sums.values()) / 1e6:.1f} ms, {len(rows)} kernels")
        for n, v in sorted(sums.items(), key=lambda t: -t[1])[:14]:
            print(f"   {n:38s} {v / 1e6:8.1f} ms")
        gaps.sort(reverse=True)
        print("longest idle gaps (us, at ms, kernel before, kernel after):")
        for g in gaps[:24]:
            print("   ", round(g[0] / 1e3), g[1:])
        print("total idle", round(sum(g[0] for g in gaps) / 1e6, 1), "ms in", len(gaps), "gaps")
        # busy fraction per 10-ms bin of the span
        nb = int((t1 - t0) / 1e7) + 1
        bins = [0] * nb
        for x in rows:
            s, e = int(x["Start_Timestamp"]) - t0, int(x["End_Timestamp"]) - t0
            b = int(s / 1e7)
            bins[b] += e - s
        print("sum of kernel time per 10-ms bin (ms):", [round(v / 1e6, 1) for v in bins])
        streams = {}
        for x in rows:
            k = x.get("Stream_Id", x.get("Queue_Id", "?"))
            streams[k] = streams.get(k, 0) + int(x["End_Timestamp"]) - int(x["Start_Timestamp"])
        print("kernel time per stream/queue (ms):", {k: round(v / 1e6, 1) for k, v in sorted(streams.items(), key=lambda t: -t[1])[:16]})
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
