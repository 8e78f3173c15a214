#!/usr/bin/env python3
"""Diagnostic: k_encode_subfilter time with and without its sampled probes (streaming + pack floor)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("danbing-tk_amd")
abi = pkg.abi


def main():
    nloci = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
    npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
    lib = pkg.Dbtk(os.path.join(ROOT, "danbing-tk_amd", "libdbtk_hip_stamps.so"))
    syn = pkg.Synth(nloci=nloci)
    a = syn.arrays()
    h = C.c_void_p()
    lib._chk(lib.L.dbtk_rpgg_from_arrays(C.byref(a), C.byref(h)))
    g = pkg.Rpgg(lib, h)
    seq, off = syn.reads(npairs, hit_frac=0.0)
    d_seq = torch.from_numpy(seq).to("cuda:0")
    d_off = torch.from_numpy(off.view(np.int64)).to("cuda:0")
    for name, knob in (("full", 0), ("no probes", 1), ("no streaming", 2), ("neither", 3)):
        prm = abi.default_params(cthreshold=45, okam=0, n_filter=4, nm_filter=1)
        prm.diag = knob
        ctx = lib.context(g, prm)
        for _ in range(2):
            ctx.align_device(d_seq.data_ptr(), d_off.data_ptr(), npairs, 150)
        ctx.synchronize()
        ctx.timers_reset()
        for _ in range(5):
            ctx.align_device(d_seq.data_ptr(), d_off.data_ptr(), npairs, 150)
        ctx.synchronize()
        kt = ctx.kernel_times()
        ms = kt["k_encode_subfilter"][0] / kt["k_encode_subfilter"][1]
        print(f"{name:24s} K1 {ms:.3f} ms for {npairs} pairs = {npairs * 300 / ms / 1e6:.0f} GB/s of read bytes; all: " + ", ".join(f"{k} {v[0] / 5:.2f}" for k, v in kt.items() if v[1]))
        ctx.close()


if __name__ == "__main__":
    main()
