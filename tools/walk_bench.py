#!/usr/bin/env python3
"""Diagnostic: time of k_walk_pairs (threading = 2) on all-hit reads of the synthetic RPGG, with and without error correction.
    python tools/walk_bench.py [nloci] [npairs]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch  # (before the library: both bring a HIP runtime, torch's has to come first)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("danbing-tk_amd")
abi = pkg.abi


def main():
    nloci = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    stamps = os.environ.get('DBTK_STAMPS_LIB')
    lib = pkg.Dbtk(os.path.join(ROOT, 'danbing-tk_amd', 'libdbtk_hip_stamps.so')) if stamps else (pkg.Dbtk(os.environ['DBTK_LIB']) if os.environ.get('DBTK_LIB') else pkg.Dbtk())
    syn = pkg.Synth(nloci=nloci)
    syn.graph()
    a = syn.arrays()
    h = C.c_void_p()
    lib._chk(lib.L.dbtk_rpgg_from_arrays(C.byref(a), C.byref(h)))
    g = pkg.Rpgg(lib, h)
    seq, off = syn.reads(npairs, hit_frac=1.0, seed=2)
    d_seq = torch.from_numpy(seq).cuda()
    d_off = torch.from_numpy(off.view(np.int64)).cuda()
    torch.cuda.synchronize()
    variants = ((1, 85, 0, "-gc 85 3"), (1, 85, 2 | abi.ALN_TEXT, "-gc 85 3 -ae (text records on the device)"), (0, 85, 0, "-g 85 (no correction)"),
                (1, 130, 0, "-gc 130 3 (every error ends the read early)"))
    for corr, cth, aln, name in variants[:int(os.environ.get('WALK_BENCH_VARIANTS', '4'))]:
        p = abi.default_params(ksize=21, cthreshold=45, okam=0, threading=2, thread_cth=cth, correction=corr, maxncorrection=3)
        p.aln = aln
        p.diag = int(os.environ.get('DBTK_DIAG', '0'))
        ctx = lib.context(g, p)
        # (text records only exist on the synchronous entry point: host buffers in, the kernels' own timers are what is read here)
        step = (lambda: ctx.align(seq, off)) if aln else (lambda: ctx.align_device(d_seq.data_ptr(), d_off.data_ptr(), npairs, 150))
        for _ in range(2):
            step()
        ctx.synchronize()
        ctx.reset(); ctx.timers_reset()
        for _ in range(5):
            step()
        ctx.synchronize()
        kt = ctx.kernel_times()
        c = ctx.counters()
        print(f"{name:45s} k_walk_pairs {kt['k_walk_pairs'][0] / kt['k_walk_pairs'][1]:8.3f} ms for {c[abi.C_THREADING] // 5} walked reads "
              f"({c[abi.C_FEASIBLE] // 5} kept) -> {c[abi.C_THREADING] / 5 / (kt['k_walk_pairs'][0] / kt['k_walk_pairs'][1] * 1e-3) / 1e6:.1f} M reads/s")
        if stamps:
            st = np.zeros(48, np.uint64)
            lib.L.dbtk_debug_stamps.argtypes = [C.c_void_p, abi.u64p]
            lib.L.dbtk_debug_stamps(ctx.h, st.ctypes.data_as(abi.u64p))
            npair = float(c[abi.C_THREADING]) / 2
            names = ["prefetch issue", "stage (bytes arrive)", "probe issue", "probe finish", "walk mate 0", "walk mate 1", "count+results", "loop tail"]
            print("   cycles per pair: " + ", ".join(f"{n} {float(st[i]) / npair:.0f}" for i, n in enumerate(names)))
            sub = ["EC forward", "edit forward", "EC backward", "edit backward", "find_anchor", "slow walk: the rest"]
            print("   inside the slow walk, per pair: " + ", ".join(f"{n} {float(st[8 + i]) / npair:.0f}" for i, n in enumerate(sub)))
        ctx.close()


if __name__ == "__main__":
    main()
