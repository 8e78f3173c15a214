"""World-size-2 check of the multi-GPU glue on CPU (gloo): each rank runs the
kernel bodies (emulator) on its shard of the pairs, the accumulator buffers are
all-reduced, and the result must equal the oracle on the whole read set."""
import os
import subprocess
import sys
import tempfile

import numpy as np

import bind

ROOT = bind.ROOT

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import bind, cases
par = __import__("importlib").import_module("danbing-tk_amd.parallel")
abi = bind.abi
rank, world, tmp = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[2]
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["MASTER_PORT"], rank=rank, world_size=world)
c = cases.make_case("mixed", os.path.join(tmp, f"r{rank}"))
E = bind.Emu(); g = E.load(c.prefix, c.k); T = E.tables(g)
seq, off = c.reads.packed()
b, e = par.shard(c.reads.npairs, rank, world)
p = abi.default_params(ksize=c.k, cthreshold=45, okam=0)
sub = off[2 * b:2 * e + 1]
r = E.align(g, T, p, seq[int(sub[0]):int(sub[-1])], sub - sub[0])
acc = torch.from_numpy(par.pack_accum(r["counts"], r["kmc"], r["nmapread"], r["counters"]).view(np.int64).copy())
par.allreduce_accum(acc)
if rank == 0:
    np.save(os.path.join(tmp, "acc.npy"), acc.numpy())
dist.barrier(); dist.destroy_process_group()
'''


def test_two_rank_shard_and_allreduce_equals_oracle():
    import cases
    abi = bind.abi
    par = __import__("importlib").import_module("danbing-tk_amd.parallel")
    assert [par.shard(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    with tempfile.TemporaryDirectory() as tmp:
        script = os.path.join(tmp, "w.py")
        open(script, "w").write(WORKER)
        env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 2000))
        procs = [subprocess.Popen([sys.executable, script, ROOT, tmp], env=dict(env, RANK=str(r))) for r in range(2)]
        for p in procs:
            assert p.wait(timeout=600) == 0
        c = cases.make_case("mixed", os.path.join(tmp, "o"))
        O = bind.Oracle()
        go = O.load(c.prefix, c.k)
        seq, off = c.reads.packed()
        p = abi.default_params(ksize=c.k, cthreshold=45, okam=0)
        o = O.align(go, p, seq, off, trace=False)
        E = bind.Emu()
        g = E.load(c.prefix, c.k)
        co = np.zeros(g.ntrkmers, np.uint64)
        np.add.at(co, g.output_order().astype(np.int64), o["counts_file"])
        got = par.unpack_accum(np.load(os.path.join(tmp, "acc.npy")), g.ntrkmers, g.nloci)
        assert (got["counts"] == co).all() and (got["kmc"] == o["kmc"]).all() and (got["nmapread"] == o["nmapread"]).all()
        assert (got["counters"] == o["counters"]).all()
