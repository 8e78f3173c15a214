#!/usr/bin/env python3
"""End-to-end command line: this CLI (GPU) next to the reference binary (its pthread CPU path, -p = host cores) on the same
FASTA and the same on-disk RPGG.  Needs oracle/_ref (built in the container, travels to the GPU box) and a GPU.
    python tests/cli_throughput.py [nloci] [npairs] [hit_frac]
Prints reads/s of both and checks that OUT.trkmc.ar / stdout are byte-identical."""
import ctypes as C
import importlib
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("danbing-tk_amd")
import synth  # noqa: E402


def np_of(ptr, n, dt):
    return np.ctypeslib.as_array(ptr, shape=(int(n),)).view(dt) if n and bool(ptr) else np.zeros(0, dt)


def main():
    nloci = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
    hit = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02
    ncpu = os.cpu_count() or 8
    d = tempfile.mkdtemp(prefix="dbtk_cli_")
    syn = pkg.Synth(nloci=nloci)
    a = syn.arrays()
    nl = int(a.nloci)
    arr = dict(keys=np_of(a.keys, a.nkeys, np.uint64), vals=np_of(a.vals, a.nkeys, np.uint32), vv=np_of(a.vv, a.nvv, np.uint32))
    for c, ks in (("fl_cnt", "fl_ks"), ("tre_cnt", "tre_ks"), ("tr_cnt", "tr_ks")):
        arr[c] = np_of(getattr(a, c), nl, np.uint64)
        if len(arr[c]) == 0:
            arr[c] = np.zeros(nl, np.uint64)
        arr[ks] = np_of(getattr(a, ks), int(arr[c].sum()), np.uint64)
    synth.write_rpgg_files(arr, os.path.join(d, "pan"))
    seq, off = syn.reads(npairs, hit_frac=hit)
    rl = int(off[1] - off[0])
    assert (np.diff(off.astype(np.int64)) == rl).all()
    # interleaved FASTA: >i/1 \n seq \n >i/2 \n seq \n  (titles of fixed width so that the file is one reshape away)
    n = 2 * npairs
    ids = np.char.zfill(np.repeat(np.arange(npairs), 2).astype("U9"), 9).astype("S9")
    rec = np.empty((n, 1 + 9 + 2 + 1 + rl + 1), np.uint8)
    rec[:, 0] = ord(">")
    rec[:, 1:10] = np.frombuffer(ids.tobytes(), np.uint8).reshape(n, 9)
    rec[:, 10] = ord("/")
    rec[:, 11] = np.tile(np.array([ord("1"), ord("2")], np.uint8), npairs)
    rec[:, 12] = 10
    rec[:, 13:13 + rl] = seq[: n * rl].reshape(n, rl)
    rec[:, 13 + rl] = 10
    fa = os.path.join(d, "reads.fa")
    rec.tofile(fa)
    print(f"RPGG {nl} loci, {len(arr['keys'])} keys; {n} reads, {os.path.getsize(fa) / 1e6:.0f} MB FASTA in {d}", flush=True)
    res = {}
    for tag, exe, extra in (("hip", os.path.join(ROOT, "danbing-tk_amd", "bin", "danbing-tk"), []),
                            ("ref", synth.ref_tool("danbing-tk"), ["-p", str(ncpu)])):
        t0 = time.time()
        r = subprocess.run([exe, "-k", "21", "-kf", "4", "1", "-cth", "45"] + extra + ["-fa", "reads.fa", "-qs", "pan", "-o", tag], cwd=d,
                           stdout=open(os.path.join(d, tag + ".kam"), "wb"), stderr=subprocess.PIPE)
        dt = time.time() - t0
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        res[tag] = dt
        for l in r.stderr.decode().split("\n"):
            if l.startswith("ingest:") or "loaded" in l.lower():
                print("   ", l)
        print(f"{tag}: {dt:.2f} s wall (incl. RPGG load) = {n / dt / 1e6:.2f} M reads/s" + (f" on {ncpu} host threads" if tag == "ref" else " on 1 GPU"), flush=True)
    same = True
    for e in (".trkmc.ar", ".tr.summary.txt", ".kam"):
        x, y = open(os.path.join(d, "hip" + e), "rb").read(), open(os.path.join(d, "ref" + e), "rb").read()
        if e == ".kam":  # the reference's -p > 1 interleaves the batches' records in completion order: compare as multisets of lines
            x, y = sorted(x.split(b"\n")), sorted(y.split(b"\n"))
        print(f"  {e}: {'identical' if x == y else 'DIFFERENT'} ({len(x)} vs {len(y)})", flush=True)
        same &= x == y
    assert same


if __name__ == "__main__":
    main()
