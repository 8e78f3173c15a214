#!/usr/bin/env python3
"""Diagnostic: k_gz_member's time (rocprofv3 --kernel-trace --stats) and the .aln.gz size of the command line's -ae run for several builds
of the library (directories holding a libdbtk_hip.so, put in front with LD_LIBRARY_PATH; "" = the product's).
    python tools/gz_ab.py [npairs=2000000] [libdir ...]"""
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
    npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    libs = sys.argv[2:] or [""]
    d = tempfile.mkdtemp(prefix="dbtk_gzab_", dir="/dev/shm")
    try:
        syn = pkg.Synth(nloci=20000)
        syn.graph()
        syn.write_files(os.path.join(d, "pan"))
        seq, _ = syn.reads(npairs, hit_frac=1.0, seed=2)
        syn.write_fasta(seq, npairs, os.path.join(d, "reads_hit.fa"))
        cli = os.path.join(ROOT, "danbing-tk_amd", "bin", "danbing-tk")
        base = [cli, "-k", "21", "-kf", "4", "1", "-cth", "45", "--v13-threading", "-gc", "85", "3", "-fa", "reads_hit.fa", "-qs", "pan", "-o", "w", "-ae", "--aln-gz", "w.aln.gz"]
        for lib in libs:
            env = dict(os.environ, TMPDIR="/tmp")
            if lib:
                env["LD_LIBRARY_PATH"] = os.path.abspath(lib) + ":" + env.get("LD_LIBRARY_PATH", "")
            pd = os.path.join(d, "prof")
            shutil.rmtree(pd, ignore_errors=True)
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", pd, "--"] + base, cwd=d, capture_output=True, text=True, env=env)
            ing = [l for l in r.stderr.splitlines() if l.startswith("ingest:")]
            gzl = ""
            for root, _, files in os.walk(pd):
                for f in files:
                    if f.endswith("kernel_stats.csv"):
                        gzl = " | ".join(",".join(l.split(",")[:4]) for l in open(os.path.join(root, f)).read().splitlines() if "k_gz_member" in l or "k_walk_pairs" in l)
            print(f"{lib or 'product':28s} rc={r.returncode} {os.path.getsize(os.path.join(d, 'w.aln.gz'))} bytes; {ing[0][:60] if ing else ''}; {gzl}", flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
