#!/usr/bin/env python3
"""A batch larger than the survivor chunk (2^23 pairs): one call of 12 M pairs with every pair surviving the subfilter would need two
chunks of hit buffers; here: 12 M pairs at the bench's 2 % (one chunk iteration does the work, the second exits) and 9 M pairs at 100 %
from the loci (both chunks work), each against the same reads in small calls.   python tools/big_batch_check.py"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("danbing-tk_amd")
abi = pkg.abi


def main():
    lib = pkg.Dbtk()
    syn = pkg.Synth(nloci=20000)
    a = syn.arrays()
    h = C.c_void_p()
    lib._chk(lib.L.dbtk_rpgg_from_arrays(C.byref(a), C.byref(h)))
    g = pkg.Rpgg(lib, h)
    p = abi.default_params(cthreshold=45, okam=0, n_filter=4, nm_filter=1)
    for npairs, hit in ((12_000_000, 0.02), (9_000_000, 1.0)):
        seq, off = syn.reads(npairs, hit_frac=hit)
        rlen = int(off[1] - off[0])
        res = []
        for cuts in ([0, npairs], [0, npairs // 3, npairs // 3 * 2, npairs]):
            ctx = lib.context(g, p)
            for x, y in zip(cuts[:-1], cuts[1:]):
                base = int(off[2 * x])
                ctx.align(seq[base:int(off[2 * y])], off[2 * x:2 * y + 1] - np.uint64(base))
            res.append(ctx.counts())
            ctx.close()
        same = all((res[0][k] == res[1][k]).all() for k in ("counts", "kmc", "nmapread", "counters"))
        c = res[0]["counters"]
        print(f"{npairs} pairs, {hit:.0%} from loci: survivors {int(c[abi.C_SURVIVORS])} (chunk = {1 << 23}), one call vs three calls: {'identical' if same else 'DIFFERENT'}", flush=True)
        assert same


if __name__ == "__main__":
    main()
