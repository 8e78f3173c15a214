#!/usr/bin/env python3
"""Diagnostic: the command line's batch loop with the reader on the device (default for a regular file) and on the host
(--host-ingest), same files, same binary: the `ingest:` lines, wall times, and that the outputs are the same bytes.
    python tools/ingest_bench.py [--nloci 20000] [--reads 32000000] [--hit 0.02] [--kam] [--chunk BYTES]"""
import argparse
import importlib
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nloci", type=int, default=20000)
    ap.add_argument("--reads", type=int, default=32_000_000)
    ap.add_argument("--hit", type=float, default=0.02)
    ap.add_argument("--kam", action="store_true", help="with kam records on stdout (default: -ka)")
    ap.add_argument("--chunk", default=None)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--rocprof", default=None, help="directory for a rocprofv3 --kernel-trace --stats run of the device-reader command")
    ap.add_argument("--variants", nargs="*", default=[], help="extra device-reader runs with K=V[,K=V] in the environment")
    a = ap.parse_args()
    pkg = importlib.import_module("danbing-tk_amd")
    syn = pkg.Synth(nloci=a.nloci, k=21)
    d = tempfile.mkdtemp(prefix="dbtk_ing_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        t0 = time.time()
        syn.write_files(os.path.join(d, "pan"))
        npairs = a.reads // 2
        part = 4_000_000  # pairs generated at a time
        with open(os.path.join(d, "reads.fa"), "wb") as out:
            for p0 in range(0, npairs, part):
                n = min(part, npairs - p0)
                seq, _ = syn.reads(n, hit_frac=a.hit, seed=1, first_pair=p0)
                syn.write_fasta(seq, n, os.path.join(d, "part.fa"), rlen=150, first_pair=p0)
                with open(os.path.join(d, "part.fa"), "rb") as f:
                    shutil.copyfileobj(f, out, 64 << 20)
        os.unlink(os.path.join(d, "part.fa"))
        print(f"files: {a.nloci} loci, {a.reads} reads, {os.path.getsize(os.path.join(d, 'reads.fa')) / 1e9:.2f} GB of FASTA in {d}, {time.time() - t0:.0f}s", flush=True)
        cli = os.path.join(ROOT, "danbing-tk_amd", "bin", "danbing-tk")
        base = [cli, "-k", "21", "-kf", "4", "1", "-cth", "45"] + ([] if a.kam else ["-ka"]) + ["-fa", "reads.fa", "-qs", "pan"]
        env = dict(os.environ)
        if a.chunk:
            env["DBTK_INGEST_CHUNK"] = a.chunk
        outs = {}
        if os.environ.get("INGEST_BENCH_CAT"):  # how long the host takes to read the fresh file once, and again: is a first pass slow by itself?
            for i in range(2):
                t0 = time.time()
                subprocess.run(["dd", "if=" + os.path.join(d, "reads.fa"), "of=/dev/null", "bs=32M"], stderr=subprocess.DEVNULL)
                print(f"dd pass {i}: {time.time() - t0:.2f} s", flush=True)
        for rep in range(a.repeat):
            for tag, extra, ev in [("dev", [], {}), ("host", ["--host-ingest"], {})] + [("dev", [], dict(kv.split("=") for kv in v.split(","))) for v in a.variants]:
                t0 = time.time()
                if ev:
                    print("variant", ev)
                with open(os.path.join(d, tag + ".kam"), "wb") as so:
                    r = subprocess.run(base + ["-o", tag] + extra, cwd=d, stdout=so, stderr=subprocess.PIPE, env=dict(env, **ev))
                dt = time.time() - t0
                if r.returncode:
                    print(tag, "FAILED", r.stderr.decode()[-3000:])
                    return 1
                ing = [l for l in r.stderr.decode().splitlines() if l.startswith("ingest:")]
                print(f"{tag} (run {rep}): {dt:.1f} s wall;  {ing[0] if ing else ''}", flush=True)
                for l in r.stderr.decode().splitlines():
                    if l.startswith("device reader:"):
                        print("     ", l, flush=True)
                outs[tag] = [open(os.path.join(d, tag + e), "rb").read() for e in (".trkmc.ar", ".tr.summary.txt", ".kam")]
        if a.rocprof:
            os.makedirs(a.rocprof, exist_ok=True)
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.abspath(a.rocprof), "--"] + base + ["-o", "prof"],
                               cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(env, TMPDIR="/tmp"))
            print("rocprofv3 rc", r.returncode, [l for l in r.stderr.decode().splitlines() if l.startswith("ingest:")])
            for root, _, files in os.walk(a.rocprof):
                for f in files:
                    if f.endswith("kernel_stats.csv"):
                        print(open(os.path.join(root, f)).read()[:4000])
        print("outputs identical:", outs["dev"] == outs["host"], f"({len(outs['dev'][0])} + {len(outs['dev'][1])} + {len(outs['dev'][2])} bytes)")
        return 0 if outs["dev"] == outs["host"] else 1
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
