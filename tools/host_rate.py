#!/usr/bin/env python3
"""The PCIe-inclusive rate: dbtk_align_batch (host buffers: validation + H2D copies + kernels) on the bench workload,
next to dbtk_align_batch_device (reads resident in HBM).   python tools/host_rate.py [nloci] [npairs]"""
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
    npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
    lib = pkg.Dbtk()
    syn = pkg.Synth(nloci=nloci)
    a = syn.arrays()
    h = C.c_void_p()
    lib._chk(lib.L.dbtk_rpgg_from_arrays(C.byref(a), C.byref(h)))
    g = pkg.Rpgg(lib, h)
    seq, off = syn.reads(npairs, hit_frac=0.02)
    p = abi.default_params(cthreshold=45, okam=0, n_filter=4, nm_filter=1)
    ctx = lib.context(g, p)
    for _ in range(2):
        ctx.align(seq, off)
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        ctx.align(seq, off)
    dt = (time.perf_counter() - t0) / n
    print(f"host buffers (pageable numpy arrays, {seq.nbytes / 1e9:.2f} GB of reads + {off.nbytes / 1e6:.0f} MB of offsets per batch): "
          f"{dt * 1e3:.1f} ms per batch of {2 * npairs} reads = {2 * npairs / dt / 1e6:.0f} M reads/s, {(seq.nbytes + off.nbytes) / dt / 1e9:.1f} GB/s over PCIe")
    d_seq = torch.from_numpy(seq).to("cuda:0")
    d_off = torch.from_numpy(off.view(np.int64)).to("cuda:0")
    for _ in range(2):
        ctx.align_device(d_seq.data_ptr(), d_off.data_ptr(), npairs, 150)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.align_device(d_seq.data_ptr(), d_off.data_ptr(), npairs, 150)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"resident in HBM: {dt * 1e3:.2f} ms per batch = {2 * npairs / dt / 1e6:.0f} M reads/s")


if __name__ == "__main__":
    main()
