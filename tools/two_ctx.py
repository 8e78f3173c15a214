#!/usr/bin/env python3
"""Diagnostic: do two contexts on one GPU (two streams, batches alternating) overlap each other's kernels?"""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("danbing-tk_amd")
abi = pkg.abi


def main():
    nloci = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
    npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 5000000
    lib = pkg.Dbtk()
    syn = pkg.Synth(nloci=nloci)
    a = syn.arrays()
    h = C.c_void_p()
    lib._chk(lib.L.dbtk_rpgg_from_arrays(C.byref(a), C.byref(h)))
    g = pkg.Rpgg(lib, h)
    seq, off = syn.reads(npairs, hit_frac=0.02)
    d_seq = torch.from_numpy(seq).to("cuda:0")
    d_off = torch.from_numpy(off.view(np.int64)).to("cuda:0")
    prm = abi.default_params(cthreshold=45, okam=0, n_filter=4, nm_filter=1)
    ctxs = [lib.context(g, prm) for _ in range(2)]
    for nctx in (1, 2):
        use = ctxs[:nctx]
        for c in use:
            c.align_device(d_seq.data_ptr(), d_off.data_ptr(), npairs, 150)
        for c in use:
            c.synchronize()
        steps = 20
        t0 = time.perf_counter()
        for i in range(steps):
            use[i % nctx].align_device(d_seq.data_ptr(), d_off.data_ptr(), npairs, 150)
        for c in use:
            c.synchronize()
        dt = time.perf_counter() - t0
        print(f"{nctx} context(s): {dt / steps * 1e3:.3f} ms/step, {2 * npairs * steps / dt / 1e9:.2f} G reads/s")


if __name__ == "__main__":
    main()
