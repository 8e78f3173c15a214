#!/usr/bin/env python3
"""Diagnostic: bench.py's genome-like mix on its own (the headline mix with 15 % of the background pairs carrying, in each mate, a 64-base
stretch of some locus over a sampled window), with the -DDBTK_STAMPS build: how many pairs reach the general resolve kernel and what
its phases cost.    python tools/genome_like.py [nloci=80000] [npairs=5000000]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch  # (before the library)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("danbing-tk_amd")
abi = pkg.abi


def main():
    nloci = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
    mp = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
    rlen = 150
    stamps = os.path.join(ROOT, "danbing-tk_amd", "libdbtk_hip_stamps.so")
    lib = pkg.Dbtk(stamps) if os.environ.get("DBTK_STAMPS_LIB") else pkg.Dbtk()
    syn = pkg.Synth(nloci=nloci)
    arr = syn.arrays()
    h = C.c_void_p()
    lib._chk(lib.L.dbtk_rpgg_from_arrays(C.byref(arr), C.byref(h)))
    g = pkg.Rpgg(lib, h)
    p = abi.default_params(ksize=21, n_filter=4, nm_filter=1, cthreshold=45, okam=0)
    seq, off = syn.reads(mp, rlen=rlen, hit_frac=0.02, seed=1)
    ah_seq, _ = syn.reads(mp, rlen=rlen, hit_frac=1.0, seed=2)
    rng = np.random.default_rng(7)
    gseq = seq[:2 * mp * rlen].copy()
    pick = np.nonzero(rng.random(mp) < 0.15)[0]
    src = rng.integers(0, mp, len(pick))
    cand = np.array([0, (rlen - 21 + 1) // 3, 2 * ((rlen - 21 + 1) // 3), rlen - 64])
    at = np.minimum(rng.choice(cand, len(pick)), rlen - 64)
    at2 = np.minimum(rng.choice(cand, len(pick)), rlen - 64)
    for q in range(64):
        gseq[2 * pick * rlen + at + q] = ah_seq[2 * src * rlen + 40 + q]
        gseq[(2 * pick + 1) * rlen + at2 + q] = ah_seq[(2 * src + 1) * rlen + 40 + q]
    d_g = torch.from_numpy(gseq).cuda()
    d_o = torch.from_numpy(off.view(np.int64)).cuda()
    ctx = lib.context(g, p)
    ctx.timers_enable(1)
    for _ in range(2):
        ctx.align_device(d_g.data_ptr(), d_o.data_ptr(), mp, rlen); ctx.synchronize()
    ctx.reset(); ctx.timers_reset()
    for _ in range(3):
        ctx.align_device(d_g.data_ptr(), d_o.data_ptr(), mp, rlen)
    ctx.synchronize()
    c = ctx.counters()
    print({k: round(v[0] / max(v[1], 1), 3) for k, v in ctx.kernel_times().items()})
    print("per step: survivors", c[abi.C_SURVIVORS] // 3, "kmerfiltered", c[abi.C_KMERFILTERED] // 3, "locusfiltered", c[abi.C_LOCUSFILTERED] // 3)
    if os.environ.get("DBTK_STAMPS_LIB"):
        st = np.zeros(48, np.uint64)
        lib.L.dbtk_debug_stamps.argtypes = [C.c_void_p, abi.u64p]
        lib.L.dbtk_debug_stamps(ctx.h, st.ctypes.data_as(abi.u64p))
        print("general kernel pairs (5 steps):", int(st[46]), " vote paths:", int(st[24]), int(st[25]), int(st[26]))
        names = ["ticket", "-", "-", "hit-buffer loads", "kfilter verdict", "gather", "rank sort", "dedup", "nml/single test", "vote fast", "vote general", "states", "assign_bits", "accumulate", "vote-order sort", "-"]
        tot = float(st[:16].sum())
        print({n: round(100 * float(v) / tot, 1) for n, v in zip(names, st[:16]) if v})
    ctx.close()


if __name__ == "__main__":
    main()
