#!/usr/bin/env python3
"""Diagnostic: BASELINE config 5's bar on one GPU — the command line's batch loop on an all-hit FASTA with the graph walk, without
and with -ae --aln-gz, for several --aln-aligners / --emit-threads settings.
    python tools/emit_bench.py [nloci=80000] [npairs=2000000] [variant ...]      variant = "A,T" (aligners, emit threads)"""
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
    nloci = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
    npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
    variants = [tuple(int(x) for x in v.split(",")) for v in sys.argv[3:]] or [(4, 64)]
    d = tempfile.mkdtemp(prefix="dbtk_emit_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        syn = pkg.Synth(nloci=nloci)
        syn.graph()
        syn.write_files(os.path.join(d, "pan"))
        seq, _ = syn.reads(npairs, hit_frac=1.0, seed=2)
        syn.write_fasta(seq, npairs, os.path.join(d, "reads_hit.fa"))
        cli = os.path.join(ROOT, "danbing-tk_amd", "bin", "danbing-tk")
        base = [cli, "-k", "21", "-kf", "4", "1", "-cth", "45", "--v13-threading", "-gc", "85", "3", "-fa", "reads_hit.fa", "-qs", "pan", "-o", "w"]

        def run(extra, tag, env=None):
            r = subprocess.run(base + extra, cwd=d, capture_output=True, text=True, env=dict(os.environ, **(env or {})))
            ing = [l for l in r.stderr.splitlines() if l.startswith("ingest:")]
            em = [l for l in r.stderr.splitlines() if l.startswith("emit:")]
            print(f"{tag:28s} rc={r.returncode} {ing[0] if ing else r.stderr[-400:]}", flush=True)
            if em:
                print(f"{'':28s} {em[0]}", flush=True)
        for al in [int(v) for v in os.environ.get("EMIT_BENCH_DEV_ALIGNERS", "").split(",") if v]:  # (the device path with other --aln-aligners)
            run(["-ae", "--aln-gz", "w_dev.aln.gz", "--aln-aligners", str(al)], f"-ae --aln-gz (device) A={al}")
            run(["-ae", "--aln-gz", "w_dev.aln.gz", "--aln-aligners", str(al)], f"-ae --aln-gz (device) A={al} again")
        run([], "no emit")
        run([], "no emit (again)")
        run(["--host-ingest"], "no emit, host reader")
        run(["-ae", "--aln-gz", "w_dev.aln.gz"], "-ae --aln-gz (device)")
        run(["-ae", "--aln-gz", "w_dev.aln.gz"], "-ae --aln-gz (device, again)")
        for ev in [dict(kv.split("=") for kv in v.split(",")) for v in os.environ.get("EMIT_BENCH_ENVS", "").split(";") if v]:
            run([], f"no emit {ev}", ev)
            run(["-ae", "--aln-gz", "w_dev.aln.gz"], f"-ae --aln-gz (device) {ev}", ev)
        if os.environ.get("EMIT_BENCH_ROCPROF"):
            pd = os.path.abspath(os.environ["EMIT_BENCH_ROCPROF"])
            os.makedirs(pd, exist_ok=True)
            prof_extra = [] if os.environ.get("EMIT_BENCH_ROCPROF_NOEMIT") else ["-ae", "--aln-gz", "w_prof.aln.gz"]
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", pd, "--"] + base + prof_extra,
                               cwd=d, capture_output=True, text=True, env=dict(os.environ, TMPDIR="/tmp"))
            print("rocprofv3 rc", r.returncode, [l for l in r.stderr.splitlines() if l.startswith("ingest:")])
            for root, _, files in os.walk(pd):
                for f in files:
                    if f.endswith("kernel_stats.csv"):
                        print("\n".join(l[:150] for l in open(os.path.join(root, f)).read().splitlines()[:int(os.environ.get('EMIT_BENCH_ROCPROF_LINES', '22'))]))
        for al, th in variants:
            run(["-ae", "--aln-gz", "w.aln.gz", "--host-ingest", "--aln-aligners", str(al), "--emit-threads", str(th)], f"-ae --aln-gz host A={al} T={th}")
        a, b = os.path.join(d, "w_dev.aln.gz"), os.path.join(d, "w.aln.gz")
        if os.path.exists(a) and os.path.exists(b):
            import gzip
            import hashlib
            def dig(fn):
                h, n = hashlib.sha256(), 0
                with gzip.open(fn, "rb") as f:
                    while True:
                        blk = f.read(64 << 20)
                        if not blk:
                            break
                        h.update(blk); n += len(blk)
                return h.hexdigest()[:16], n
            da, db = dig(a), dig(b)
            print(f"zcat device stream == zcat host stream: {da == db} ({da[1]} bytes of text; .gz {os.path.getsize(a)} vs {os.path.getsize(b)} bytes)", flush=True)
            if da != db:
                with gzip.open(a, "rb") as fa, gzip.open(b, "rb") as fb:
                    nshow = 0
                    for ln, (x, y) in enumerate(zip(fa, fb)):
                        if x != y:
                            fx, fy = x.split(b"\t"), y.split(b"\t")
                            print(f"line {ln}: device {fx[:3] + fx[5:]}\n         host   {fy[:3] + fy[5:]}", flush=True)
                            nshow += 1
                            if nshow == 6:
                                break
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
