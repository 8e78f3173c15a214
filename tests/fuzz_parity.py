#!/usr/bin/env python3
"""Randomised parity soak on the GPU: random RPGGs, read sets and parameters, the HIP path against the oracle
(records in trace mode, counts / totals / counters without).   python tests/fuzz_parity.py [nseeds] [first_seed] [k,k,...]
python tests/fuzz_parity.py walk [nseeds] [first_seed]: the same for the hot loop with threading = 2 (graph walk, exact
counts, walk results, -a / -ae records)."""
import importlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # (this file lives there: it drives the oracle, which only tests may do)
import bind  # noqa: E402
import cases  # noqa: E402
import synth  # noqa: E402

abi = bind.pkg.abi
KSET = []  # optional third argument: the k values to draw from (default: 17 / 21 / 25)


def walk_main(argv):
    """Random RPGGs with their graphs, read sets with errors of every kind, random walk parameters: dbtk_align_batch with
    threading = 2 against orc_align_walk (counts, kmc, nmapread, counters, walk results, alignment records)."""
    import test_walk
    n = int(argv[0]) if argv else 20
    s0 = int(argv[1]) if len(argv) > 1 else 5000
    dbtk, orc = bind.pkg.Dbtk(), bind.Oracle()
    bad = 0
    for seed in range(s0, s0 + n):
        rng = np.random.default_rng(seed)
        k = int(rng.choice(KSET if KSET else [21, 21, 25, 25, 17]))
        loci = synth.make_loci(nloci=int(rng.integers(3, 60)), nhap=int(rng.integers(1, 4)), flank=int(rng.integers(300, 700)), seed=seed,
                               shared_frac=float(rng.choice([0.0, 0.2, 0.6])), tr_min=int(rng.integers(40, 200)), tr_max=int(rng.integers(300, 1200)))
        rlen = int(rng.choice([150, 150, 100, 250, 80] if not KSET else [150, 250, 200, 100]))
        reads = synth.sim_reads(loci, npairs=int(rng.integers(100, 1500)), rlen=rlen, seed=seed + 7, sub=float(rng.choice([0.0, 0.005, 0.02, 0.05])),
                                indel=float(rng.choice([0.0, 0.002, 0.01])), nrate=float(rng.choice([0.0, 0.003])),
                                chimeric=float(rng.choice([0.0, 0.2])), background=float(rng.choice([0.0, 0.2])), frag=(max(300, rlen), max(320, rlen) + 250))
        nk = rlen - k + 1
        cth = min(int(rng.choice([45, 30, 10])), max(1, nk // 2))
        ps = dict(thread_cth=int(rng.integers(max(1, nk // 3), nk + 10)), correction=int(rng.integers(0, 2)), maxncorrection=int(rng.integers(0, 5)))
        aln = int(rng.integers(0, 3))
        with tempfile.TemporaryDirectory() as d:
            pref = os.path.join(d, "pan")
            synth.write_rpgg_files(synth.build_rpgg_arrays(loci, k), pref)
            synth.write_graph_file(synth.build_graph_arrays(loci, k), pref, binary=bool(seed & 1))
            go = orc.load(pref, k)
            orc.load_graph(go, pref + (".graph.umap" if seed & 1 else ".graph.kmers"))
            g = dbtk.load(pref, k, flags=abi.LOAD_GRAPH)
            p = abi.default_params(ksize=k, cthreshold=cth, threading=2, aln=aln, okam=0, **ps)
            seq, off = reads.packed()
            o = orc.align_walk(go, p, seq, off)
            ctx = dbtk.context(g, p)
            ctx.align(seq, off)
            r = ctx.counts()
            res, _, nres = ctx.walk_results(len(off))
            got_aln = ctx.aln_records()
            co = np.zeros(g.ntrkmers, np.uint64)
            np.add.at(co, g.output_order().astype(np.int64), o["counts_file"])
            ok = bool((co == r["counts"]).all() and (o["kmc"] == r["kmc"]).all() and (o["nmapread"] == r["nmapread"]).all()
                      and (o["counters"] == r["counters"]).all())
            m = o["nres"]
            ok &= nres == m and bind.walk_res_equal(res, o["res"], m, loci.nloci, every_mate=bool(aln)) >= 0
            exp, _ = test_walk.expected_aln(orc, o, reads, aln, loci.nloci)
            ok &= ([(h.pair, h.dst, t) for h, t in got_aln] == exp) if aln else (got_aln == [])
            ctx.close()
            orc.free(go)
            g.close()
        print(f"walk seed {seed}: k={k} rlen={rlen} pairs={reads.npairs} cth={cth} aln={aln} {ps} walked={m} -> {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
    print(f"{n - bad}/{n} walk seeds bit-exact")
    sys.exit(1 if bad else 0)


def main():
    global KSET
    if len(sys.argv) > 1 and "," in sys.argv[-1] or (len(sys.argv) > 3 and sys.argv[-1].isdigit() and sys.argv[1] != "walk") or (len(sys.argv) > 4 and sys.argv[-1].isdigit()):
        KSET = [int(v) for v in sys.argv.pop().split(",")]
    if len(sys.argv) > 1 and sys.argv[1] == "walk":
        return walk_main(sys.argv[2:])
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    dbtk, orc = bind.pkg.Dbtk(), bind.Oracle()
    bad = 0
    for seed in range(s0, s0 + n):
        rng = np.random.default_rng(seed)
        k = int(rng.choice(KSET if KSET else [21, 21, 21, 17, 25]))
        loci = synth.make_loci(nloci=int(rng.integers(3, 40)), nhap=int(rng.integers(1, 4)), flank=int(rng.integers(300, 700)), seed=seed,
                               shared_frac=float(rng.choice([0.0, 0.2, 0.6, 1.0])), tr_min=int(rng.integers(40, 200)), tr_max=int(rng.integers(300, 1200)))
        rlen = int(rng.choice([150, 150, 100, 250, 64] if not KSET else [150, 250, 200, 100]))
        reads = synth.sim_reads(loci, npairs=int(rng.integers(50, 900)), rlen=rlen, seed=seed + 7, sub=float(rng.choice([0.0, 0.005, 0.03])),
                                indel=float(rng.choice([0.0, 0.002])), nrate=float(rng.choice([0.0, 0.003, 0.02])),
                                chimeric=float(rng.choice([0.0, 0.3])), background=float(rng.choice([0.0, 0.3])), frag=(max(300, rlen), max(320, rlen) + 250),
                                splice=float(rng.choice([0.0, 0.0, 0.3])))  # (spliced pairs: the fused probe kernel resolves them ahead and must take them back)
        with tempfile.TemporaryDirectory() as d:
            pref = cases._np_rpgg(d, "f", loci, k)
            go, g = orc.load(pref, k, None), dbtk.load(pref, k, None)
            seq, off = reads.packed()
            cth = int(rng.choice([45, 30, 10, 60]))
            if rlen - k + 1 < cth:
                cth = max(1, (rlen - k + 1) // 2)
            kw = dict(cthreshold=cth, okam=int(rng.integers(0, 2)))
            if rng.random() < 0.3:
                kw.update(n_filter=int(rng.choice([2, 4, 8])), nm_filter=int(rng.choice([1, 2])))
            ok = True
            for trace, nokam in ((1, False), (0, False), (0, True)):  # (the last: no record buffer at all — the fused resolve of the locus-resident probe kernel)
                p = abi.default_params(ksize=k, trace=trace, **(dict(kw, okam=0) if nokam else kw))
                o = orc.align(go, p, seq, off, trace=bool(trace))
                ctx = dbtk.context(g, p)
                recs, nrec = ctx.align(seq, off)
                r = ctx.counts()
                co = np.zeros(g.ntrkmers, np.uint64)
                np.add.at(co, g.output_order().astype(np.int64), o["counts_file"])
                same = bool((co == r["counts"]).all() and (o["kmc"] == r["kmc"]).all() and (o["nmapread"] == r["nmapread"]).all()
                            and (o["counters"] == r["counters"]).all())
                if trace and same:
                    same = bind.recs_equal(o["recs"], recs, reads.npairs) < 0
                ok &= same
                ctx.close()
            orc.free(go)
            g.close()
        print(f"seed {seed}: k={k} rlen={rlen} pairs={reads.npairs} {kw} -> {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
    print(f"{n - bad}/{n} seeds bit-exact")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
