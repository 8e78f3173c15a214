"""Multi-GPU glue: reads shard across ranks with no data-path collective; the
per-rank accumulators (counts | kmc | nmapread | counters, one 64-bit buffer)
are summed ONCE at the end — the cross-process form of the reference's shared
atomics (src/aQueryFasta_thread.cpp:2146-2158) and counter merge (:1887-1895).
Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU test."""
from __future__ import annotations

import numpy as np


def shard(npairs: int, rank: int, world: int):
    """Contiguous, balanced [begin, end) of the pairs of a read set (strong scaling)."""
    base, rem = divmod(npairs, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def pack_accum(counts, kmc, nmapread, counters) -> np.ndarray:
    """Layout of dbtk_ctx_accum_buffer(): u64 counts | kmc | nmapread (widened) | counters."""
    return np.concatenate([counts.astype(np.uint64), kmc.astype(np.uint64), nmapread.astype(np.uint64),
                           counters.astype(np.uint64)])


def unpack_accum(buf: np.ndarray, ntr: int, nloci: int):
    buf = buf.view(np.uint64)
    return dict(counts=buf[:ntr], kmc=buf[ntr:ntr + nloci],
                nmapread=buf[ntr + nloci:ntr + 2 * nloci].astype(np.uint32),  # atomic_uint32_t in the reference: wraps
                counters=buf[ntr + 2 * nloci:])


def allreduce_accum(t, group=None):
    """Sum an int64 torch tensor holding the accumulator buffer over all ranks.
    Two's-complement int64 adds wrap exactly like the reference's uint64 adds."""
    import torch.distributed as dist
    if dist.is_initialized():  # (also with one rank: the same RCCL call, a copy onto itself)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
