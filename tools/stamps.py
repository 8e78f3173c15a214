#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of k_pair from the -DDBTK_STAMPS build
(make -C danbing-tk_amd/csrc stamps).  Read the SHARES, not the run time."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("danbing-tk_amd")
abi = pkg.abi
NAMES = ["ticket", "-", "-", "hit-buffer loads", "kfilter verdict", "gather", "rank sort", "dedup", "nml/single test",
         "vote fast", "vote general", "states", "assign_bits", "accumulate", "vote-order sort", "-",
         "K1 tile set-up", "K1 loads+pack", "K1 valid-window", "K1 sampled probes", "K1 verdict+append"] + ["-"] * 11


def main():
    nloci = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    hit = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02
    lib = pkg.Dbtk(os.path.join(ROOT, "danbing-tk_amd", "libdbtk_hip_stamps.so"))
    lib.L.dbtk_debug_stamps.argtypes = [C.c_void_p, abi.u64p]
    syn = pkg.Synth(nloci=nloci)
    a = syn.arrays()
    h = C.c_void_p()
    lib._chk(lib.L.dbtk_rpgg_from_arrays(C.byref(a), C.byref(h)))
    g = pkg.Rpgg(lib, h)
    p = abi.default_params(cthreshold=45, okam=0)
    p.diag = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # diagnostic knobs of the stamps build
    ctx = lib.context(g, p)
    seq, off = syn.reads(npairs, hit_frac=hit)
    for _ in range(3):
        ctx.align(seq, off)
    st = np.zeros(48, np.uint64)
    lib.L.dbtk_debug_stamps(ctx.h, st.ctypes.data_as(abi.u64p))
    r = ctx.counts()["counters"]
    print(f"survivors/step {r[abi.C_SURVIVORS] / 3:.0f}  kernels {ctx.kernel_times()}")
    tot = float(st[:16].sum())
    for n, v in zip(NAMES[:16], st[:16]):
        if v:
            print(f"{n:18s} {100 * float(v) / tot:6.2f} %   {float(v) / (r[abi.C_SURVIVORS]):10.0f} cycles/pair")
    print(f"vote paths: single-locus {st[24]}  all-equal-nml {st[25]}  introsort {st[26]}   mean nu {st[27] / max(1, st[24] + st[25] + st[26]):.1f}  mean n {st[28] / max(1, st[24] + st[25] + st[26]):.1f}")
    st[24:32] = 0
    tot = float(st[16:32].sum())
    ntiles = 3 * ((npairs + 15) // 16)
    for n, v in zip(NAMES[16:32], st[16:32]):
        if v:
            print(f"{n:18s} {100 * float(v) / tot:6.2f} %   {float(v) / ntiles:10.0f} cycles/tile")


    print(f"general kernel: {st[46]} pairs, {st[44]} serial-vote fallbacks, slowest pair {st[45]} cycles")
    print(f"  slowest pair: {int(st[47]) >> 32} cycles, {(int(st[47]) >> 12) & 0xFFFFF} loci-list words, {int(st[47]) & 0xFFF} distinct k-mers")
    print("  pair times (cycles, all 3 steps): " + ", ".join(f"<2^{12 + 2 * b}: {int(st[21 + b if b < 3 else 26 + b])}" for b in range(6)))
    st[44:47] = 0
    # (the lean probe kernel, dbtk_probe2.h; its stamps 16 and 18 share their words with K1's tile stamps above: with survivors to probe
    # the K1 lines for those two indexes are not K1's alone)
    k2 = {40: "K2 fetch+pack+windows+hashes", 16: "K2 minimizers+runs", 18: "K2 level 1 (buckets)", 41: "K2 level 2 (overflow)", 42: "K2 results", 43: "K2 loop"}
    nrows = 2.0 * r[abi.C_SURVIVORS]
    for i, n in k2.items():
        print(f"{n:30s} {float(st[i]) / nrows:10.0f} cycles/read")
    st[40:44] = 0
    u = st[32:48]
    names = {32: "usual: request+test", 33: "usual: states", 34: "usual: assign", 35: "usual: LDS histogram", 36: "usual: delivery", 37: "usual: count atomics", 39: "usual: loop/record"}
    tot = float(u.sum())
    for i, n in names.items():
        print(f"{n:22s} {100 * float(st[i]) / max(tot, 1):6.2f} %   {float(st[i]) / r[abi.C_SURVIVORS]:10.0f} cycles/pair")


if __name__ == "__main__":
    main()
