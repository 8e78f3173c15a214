#!/usr/bin/env python3
"""Throughput of the danbing-tk-pred kernels on one MI355X (include/dbtk_pred.h): a synthetic cohort of ns samples over an
RPGG of ntr loci x kpl k-mers, a tenth of them invariant.  Prints the kernel times and their HBM rates (algorithmic bytes:
bias sums 4 B per (invariant k-mer, sample); correction 8 B per matrix entry, read + write).
    python tools/pred_bench.py [ns] [ntr] [kpl]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("danbing-tk_amd")


def main():
    ns = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    ntr = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    kpl = int(sys.argv[3]) if len(sys.argv) > 3 else 184
    rng = np.random.default_rng(1)
    nk = ntr * kpl
    nk_cum = (np.arange(1, ntr + 1) * kpl).astype(np.uint32)
    nikl = max(1, kpl // 10)
    iki = (np.arange(ntr)[:, None] * kpl + np.sort(rng.integers(0, kpl, (ntr, nikl)), axis=1)).astype(np.uint32).ravel()
    nik_cum = (np.arange(1, ntr + 1) * nikl).astype(np.uint32)
    ikmc = rng.integers(1, 4, len(iki)).astype(np.uint8)
    depths = rng.uniform(10, 50, ns).astype(np.float32)
    lib = pkg.Dbtk()
    P = pkg.Pred(lib, ns, nk_cum, nik_cum, iki, ikmc, nk=nk)
    t0 = time.time()
    B = 16
    for s0 in range(0, ns, B):
        n = min(B, ns - s0)
        P.load(s0, rng.integers(0, 200, (n, nk), dtype=np.uint64), depths[s0:s0 + n])
    t_load = time.time() - t0
    P.correct()
    P.correct()
    ms = P.times()
    bytes_bias, bytes_cor = 4.0 * len(iki) * ns, 8.0 * nk * ns
    print(f"cohort {ns} samples x {nk} k-mers ({4 * nk * ns / 1e9:.2f} GB matrix), {ntr} loci, {len(iki)} invariant k-mers; load (host RNG + PCIe) {t_load:.1f}s")
    print(f"k_pred_bias {ms[0]:.3f} ms = {bytes_bias / ms[0] / 1e6:.0f} GB/s   k_pred_bias_norm {ms[1]:.3f} ms   "
          f"k_pred_correct {ms[2]:.3f} ms = {bytes_cor / ms[2] / 1e6:.0f} GB/s ({bytes_cor / ms[2] / 1e6 / 8000:.1%} of 8 TB/s)")
    P.close()


if __name__ == "__main__":
    main()
