#!/usr/bin/env python3
"""Diagnostic: the command line end to end at release scale — start-up (RPGG files, tables in HBM), the batch loop, the walk with the
device reader — each leg twice (the first pass over a freshly written tmpfs file is bound by the host's first touch of its pages).
    python tools/cli_e2e.py [nloci] [reads]        (writes into /dev/shm, removes it)"""
import importlib
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("danbing-tk_amd")


def main():
    nloci = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
    nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 8_000_000
    d = tempfile.mkdtemp(prefix="dbtk_e2e_", dir="/dev/shm")
    try:
        syn = pkg.Synth(nloci=nloci)
        syn.graph()
        syn.write_files(os.path.join(d, "pan"))
        seq, _ = syn.reads(nreads // 2, hit_frac=0.02, seed=1)
        syn.write_fasta(seq, nreads // 2, os.path.join(d, "reads.fa"))
        hs, _ = syn.reads(5_000_000, hit_frac=1.0, seed=2)
        syn.write_fasta(hs, 5_000_000, os.path.join(d, "reads_hit.fa"))
        del seq, hs
        syn.close()
        cli = os.path.join(ROOT, "danbing-tk_amd", "bin", "danbing-tk")
        legs = [("count", ["-k", "21", "-kf", "4", "1", "-cth", "45", "-ka", "-fa", "reads.fa", "-qs", "pan", "-o", "c"], {}),
                ("count again", ["-k", "21", "-kf", "4", "1", "-cth", "45", "-ka", "-fa", "reads.fa", "-qs", "pan", "-o", "c"], {}),
                ("walk", ["-k", "21", "-kf", "4", "1", "-cth", "45", "--v13-threading", "-gc", "85", "3", "-fa", "reads_hit.fa", "-qs", "pan", "-o", "w"], {}),
                ("walk again", ["-k", "21", "-kf", "4", "1", "-cth", "45", "--v13-threading", "-gc", "85", "3", "-fa", "reads_hit.fa", "-qs", "pan", "-o", "w"], {}),
                ("walk, blocks one by one", ["-k", "21", "-kf", "4", "1", "-cth", "45", "--v13-threading", "-gc", "85", "3", "-fa", "reads_hit.fa", "-qs", "pan", "-o", "w1"], {"DBTK_NO_MERGE": "1"})]
        for name, argv, env in legs:
            t0 = time.perf_counter()
            r = subprocess.run([cli] + argv, cwd=d, capture_output=True, text=True, env=dict(os.environ, DBTK_VERBOSE="1", **env))
            w = time.perf_counter() - t0
            keep = [l for l in r.stderr.splitlines() if l.startswith(("load:", "ingest:", "rpgg ", "device reader:", "timeline:", "tables: "))]
            print(f"== {name}: rc {r.returncode}, {w:.2f} s wall")
            if name == "count again" and os.environ.get("CLI_E2E_ALL"):  # (every timing line the library and the command line print under DBTK_VERBOSE)
                keep = r.stderr.splitlines()[:80]
            for l in keep:
                print("   " + l[:400])
        same = open(os.path.join(d, "w.trkmc.ar"), "rb").read() == open(os.path.join(d, "w1.trkmc.ar"), "rb").read()
        print("walk outputs identical with and without merged batches:", same)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
