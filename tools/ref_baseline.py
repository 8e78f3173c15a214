#!/usr/bin/env python3
"""The reference's own pthread CPU path on the bench workload: oracle/_ref/danbing-tk (compiled from /root/reference by
oracle/Makefile) on the release-scale synthetic RPGG written out as the HEAD files it loads, and a FASTA sample of the
same read generator.  Timed from its "threads created" line to the end of its last batch line (stderr, stamped here).

    python tools/ref_baseline.py [--nloci 80000] [--reads 8000000] [--hit-frac 0.02] [-p 1 8 256]
"""
import argparse
import importlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_reference(ref, workdir, fasta, nproc, extra=(), kill_after_query=True):
    """-> dict(p, load_s, query_s, total_s, reads, returncode).  query_s: from "threads created" until the process has
    printed "parallel query completed" (the reference's own bracket, AQ.cpp:2576-2627), on this side's clock."""
    cmd = [ref, "-k", "21", "-kf", "4", "1", "-cth", "45", "-ka", "-p", str(nproc), *extra, "-fa", fasta, "-qs", "pan", "-o", f"ref_p{nproc}"]
    t0 = time.perf_counter()
    p = subprocess.Popen(cmd, cwd=workdir, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, bufsize=1)
    t_created = t_done = None
    reads = None
    for line in p.stderr:
        now = time.perf_counter()
        if line.startswith("threads created") or "threads created" in line:
            t_created = now
        elif "parallel query completed" in line:
            t_done = now
            if kill_after_query:  # its dumps and the teardown of its hash maps (tens of seconds at release scale) are not measured
                p.kill()
                break
        elif line.rstrip().endswith("reads processed in total."):
            reads = int(line.split()[0])
    p.wait()
    t1 = time.perf_counter()
    return dict(p=nproc, load_s=(t_created - t0) if t_created else None, query_s=(t_done - t_created) if (t_created and t_done) else None,
                total_s=t1 - t0, reads=reads, returncode=p.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nloci", type=int, default=80000)
    ap.add_argument("--reads", type=int, default=8_000_000)
    ap.add_argument("--hit-frac", type=float, default=0.02)
    ap.add_argument("-p", type=int, nargs="+", default=[1, 8, os.cpu_count() or 8])
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()
    pkg = importlib.import_module("danbing-tk_amd")
    ref = os.path.join(ROOT, "oracle", "_ref", "danbing-tk")
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    wd = tempfile.mkdtemp(prefix="dbtk_ref_", dir=base)
    try:
        t0 = time.time()
        syn = pkg.Synth(nloci=args.nloci, k=21, flank=700, seed=20250808)
        print(f"synth {time.time() - t0:.1f}s", file=sys.stderr)
        t0 = time.time()
        syn.write_files(os.path.join(wd, "pan"))
        print(f"files {time.time() - t0:.1f}s: " + ", ".join(f"{f} {os.path.getsize(os.path.join(wd, f)) / 1e6:.0f} MB" for f in sorted(os.listdir(wd))), file=sys.stderr)
        t0 = time.time()
        npairs = args.reads // 2
        seq, _ = syn.reads(npairs, hit_frac=args.hit_frac, seed=1)
        syn.write_fasta(seq, npairs, os.path.join(wd, "reads.fa"))
        print(f"reads + fasta {time.time() - t0:.1f}s ({os.path.getsize(os.path.join(wd, 'reads.fa')) / 1e6:.0f} MB)", file=sys.stderr)
        out = []
        for p in args.p:
            r = run_reference(ref, wd, "reads.fa", p)
            r["reads_per_s"] = (r["reads"] / r["query_s"]) if r["query_s"] else None
            print(json.dumps(r), flush=True)
            out.append(r)
    finally:
        if not args.keep:
            shutil.rmtree(wd, ignore_errors=True)


if __name__ == "__main__":
    main()
