#!/usr/bin/env python3
"""Child process of test_gpu_parity.py::test_accumulator_as_torch_tensor_and_rccl: torch first, then the library."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import bind  # noqa: E402
import synth  # noqa: E402

abi = bind.abi


def main():
    prefix, k = sys.argv[1], int(sys.argv[2])
    dbtk = bind.pkg.Dbtk()
    g = dbtk.load(prefix, k)
    ctx = dbtk.context(g, abi.default_params(ksize=k, okam=0))
    loci = synth.make_loci(nloci=6, nhap=2, flank=300, seed=5)
    reads = synth.sim_reads(loci, npairs=400, seed=6)  # any reads do: the check is view == copy-out, before and after the reduce
    seq, off = reads.packed()
    ctx.align(seq, off)
    ctx.synchronize()
    want = ctx.counts()
    ptr, n = ctx.accum_buffer()

    class _Acc:
        __cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 2}
    t = torch.as_tensor(_Acc(), device="cuda:0")
    assert t.data_ptr() == ptr and t.numel() == n, "the tensor must alias the accumulator"
    host = t.cpu().numpy().view(np.uint64)
    assert (host[:g.ntrkmers] == want["counts"]).all()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        dist.all_reduce(t)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    got = ctx.counts()
    assert (got["counts"] == want["counts"]).all() and (got["kmc"] == want["kmc"]).all() and (got["counters"] == want["counters"]).all()
    print("ACCUM-OK")


if __name__ == "__main__":
    main()
