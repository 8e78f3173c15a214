"""The oracle (oracle/dbtk_oracle.c) against the real reference: function by
function through oracle/_ref/libdbtk_refharness.so (marker `ref`, needs the
compiled reference) and end to end against the committed golden fixtures the
reference binary produced (tests/golden/, always)."""
import os

import numpy as np
import pytest

import bind
import cases
import refio
import synth

abi = bind.abi
GOLDEN = cases.GOLDEN
needs_ref = pytest.mark.skipif(not synth.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.fixture(scope="module")
def O():
    return bind.Oracle()


@pytest.fixture(scope="module")
def R():
    return bind.RefHarness()


@needs_ref
def test_nurc_matches_reference(O, R):
    rng = np.random.default_rng(0)
    for k in (2, 5, 17, 21, 25, 31):
        for _ in range(500):
            x = int(rng.integers(0, 1 << 62)) & ((1 << (2 * k)) - 1)
            assert O.L.orc_nurc(x, k) == R.L.ref_nurc(x, k)


@needs_ref
def test_read2kmers_edges_matches_reference(O, R):
    rng = np.random.default_rng(1)
    alph = np.frombuffer(b"ACGTACGTACGTACGTNacgt", np.uint8)
    fixed = [b"", b"A", b"ACGT" * 5, b"ACGT" * 5 + b"A", b"N" * 40, b"A" * 100, b"ACGTN" * 30, b"acgt" * 30,
             b"ACGTACGTACGTACGTACGTAN" + b"C" * 30]
    for it in range(1500):
        if it < len(fixed):
            s = fixed[it]
        else:
            L = int(rng.integers(0, 260))
            s = alph[rng.integers(0, len(alph) if it % 3 == 0 else 4, L)].tobytes()
        for k in (5, 21, 25, 31):
            a, b = O.read2kmers_edges(s, k), R.read2kmers_edges(s, k)
            assert len(a[0]) == len(b[0]) and (a[0] == b[0]).all() and (a[1] == b[1]).all(), (s, k)


def killer(n):
    """median-of-3 killer sequence: drives introsort into its heapsort fallback."""
    k = n // 2
    a = [0] * n
    for i in range(1, k + 1):
        if i % 2:
            a[i - 1] = i
            a[i] = k + i
        a[k + i - 1] = 2 * i
    return np.array(a, np.uint64)


@needs_ref
def test_sort_index_is_gcc_std_sort(O, R):
    rng = np.random.default_rng(2)
    for it in range(3000):
        n = int(rng.integers(0, 600))
        hi = int(rng.choice([1, 2, 3, 5, 50, 100000]))
        d = rng.integers(0, hi, n).astype(np.uint64)
        if it % 7 == 0:
            d = np.sort(d)
        if it % 11 == 0:
            d = np.sort(d)[::-1].copy()
        assert (O.sort_index(d) == R.sort_index(d)).all(), (n, hi)
    for n in (1, 2, 16, 17, 33, 130, 260, 512):
        d = np.ones(n, np.uint64)
        assert (O.sort_index(d) == R.sort_index(d)).all()
    for n in (64, 128, 200, 260, 400, 512):
        d = killer(n)
        assert (O.sort_index(d) == R.sort_index(d)).all()


@needs_ref
def test_unordered_map_iteration_order(O, R):
    rng = np.random.default_rng(3)
    for it in range(150):
        n = int(rng.integers(0, 4000))
        keys = rng.integers(0, 1 << 42, n).astype(np.uint64)
        if it % 5 == 0 and n > 10:
            keys[n // 2] = keys[0]
        assert (O.umap_order(keys) == R.umap_order(keys)).all(), n


@needs_ref
@pytest.mark.parametrize("case", sorted(cases.CASES))
def test_pipeline_matches_reference_functions(case, O, R, tmp_path):
    """Every pair through the reference's own functions (harness) vs the oracle:
    counts, kmc, nmapread, counters and all per-pair record fields."""
    c = cases.make_case(case, str(tmp_path))
    go, gr = O.load(c.prefix, c.k, c.qc_file), R.load(c.prefix, c.qc_file)
    seq, off = c.reads.packed()
    for kw in c.param_sets:
        if kw.get("extract"):
            continue  # the harness glue has no extract mode
        p = abi.default_params(ksize=c.k, **kw)
        a, b = O.align(go, p, seq, off), R.align(gr, p, seq, off)
        for f in ("counts_file", "kmc", "nmapread"):
            assert (a[f] == b[f]).all(), f
        ca = a["counters"].copy()
        ca[abi.C_ALGO_PROBES:] = 0  # algorithmic-work counters are the oracle's own bookkeeping
        assert (ca == b["counters"]).all()
        d = bind.recs_equal(a["recs"], b["recs"], c.reads.npairs)
        assert d < 0, f"{bind.rec_str(a['recs'][d])}\n{bind.rec_str(b['recs'][d])}"
    O.free(go)
    R.free(gr)


@needs_ref
def test_qstring2qmask_matches_reference(O, R):
    rng = np.random.default_rng(4)
    for it in range(3000):
        n = int(rng.integers(45, 256))
        k = int(rng.choice([21, 25]))
        mode = it % 4
        q = rng.integers(33 + (25 if mode == 0 else 2), 33 + 41, n, dtype=np.uint8)
        if mode == 2:
            q[rng.integers(0, n, 3)] = 33 + 3
        if mode == 3:
            q[: int(rng.integers(0, n))] = 33 + 5
        qb = q.tobytes()
        for qth in (20, 10):
            assert (O.qmask(qb, qth, k) == R.qmask(qb, qth, k)).all(), (it, n, k, qth)


@needs_ref
@pytest.mark.parametrize("fastq", [False, True])
def test_bait_and_bubbles_match_reference_functions(O, R, tmp_path, fastq):
    """-b (bfilter_FPSv1 + qString2qMask) and -bu (countNovelEdges) through the reference's own functions."""
    loci = synth.make_loci(nloci=10, nhap=3, flank=500, seed=61, shared_frac=0.3)
    d = str(tmp_path)
    pref = synth.build_rpgg_with_reference(loci, d, k=21)
    reads = synth.sim_reads(loci, npairs=1500, seed=62, sub=0.01, indel=0.002, nrate=0.002, chimeric=0.3, background=0.1,
                            with_qual=fastq)
    bait = synth.make_bait_db(loci, reads, d)
    go, gr = O.load(pref, 21), R.load(pref)
    O.load_bait(go, bait)
    R.load_bait(gr, pref)
    seq, off = reads.packed()
    qual = np.frombuffer(b"".join(reads.quals), np.uint8).copy() if fastq else None
    for kw in (dict(bait=1, bubbles=1, cthreshold=45), dict(bait=1, cthreshold=20, okam=0), dict(bubbles=1, cthreshold=30)):
        p = abi.default_params(ksize=21, **kw)
        a = O.align_ex(go, p, seq, off, qual)
        b = R.align_ex(gr, p, seq, off, qual)
        for f in ("counts_file", "kmc", "nmapread"):
            assert (a[f] == b[f]).all(), f
        ca = a["counters"].copy()
        ca[abi.C_ALGO_PROBES:] = 0
        assert (ca == b["counters"]).all(), (ca, b["counters"])
        dd = bind.recs_equal(a["recs"], b["recs"], reads.npairs)
        assert dd < 0, f"{bind.rec_str(a['recs'][dd])}\n{bind.rec_str(b['recs'][dd])}"
        if p.bubbles:
            assert len(a["events"]) == len(b["events"]) and (a["events"] == b["events"]).all()
            assert len(a["events"]) > 100
        if p.bait:
            assert a["counters"][abi.C_BAITFILTERED] > 0


GOLD = {
    "g1_k21": dict(k=21, fastq=False, params=dict(cthreshold=45), qc=False),
    "g2_shared_fq": dict(k=21, fastq=True, params=dict(), qc=False),
    "g3_k25_qc": dict(k=25, fastq=False, params=dict(cthreshold=40, nm_tr=30, qc=1), qc=True),
}


def golden_inputs(name):
    spec = GOLD[name]
    d = os.path.join(GOLDEN, name)
    p = abi.default_params(ksize=spec["k"], **spec["params"])
    reads = refio.read_pairs(os.path.join(d, "reads.fq" if spec["fastq"] else "reads.fa"), spec["fastq"],
                             p.cthreshold + p.ksize - 1)
    return d, p, reads, (os.path.join(d, "qc.txt") if spec["qc"] else None)


def check_against_golden(d, res_counts_out, kmc, nmapread, counters, recs, reads, out_kmers_by_locus=None):
    """Compare results with the bytes the reference binary wrote."""
    ar = np.fromfile(os.path.join(d, "ref.trkmc.ar"), np.uint64)
    assert ar[0] == len(res_counts_out) and (ar[1:] == res_counts_out).all()
    summ = np.loadtxt(os.path.join(d, "ref.tr.summary.txt"), dtype=np.uint64, ndmin=2)
    assert (summ[:, 0] == nmapread).all() and (summ[:, 1] == kmc).all()
    tot = [int(l.split()[0]) for l in open(os.path.join(d, "ref.totals.txt"))]
    mine = [counters[i] for i in (abi.C_NREADS, abi.C_SUBFILTERED, abi.C_KMERFILTERED, abi.C_BAITFILTERED, abi.C_QUALFILTERED,
                                   abi.C_LOCUSFILTERED, abi.C_QCFILTERED, abi.C_THREADING, abi.C_FEASIBLE, abi.C_ASGN)]
    assert tot == [int(x) for x in mine]
    kam = refio.parse_kam(os.path.join(d, "ref.kam.txt"))
    mykam = [r for r in recs if r.stage == abi.STAGE_COUNTED]
    assert len(kam) == len(mykam)
    for line, r in zip(kam, mykam):
        assert line["dst"] == r.dst and line["dst0"] == -1 and line["src"] == "."
        assert line["len2"] == r.r2.ei - r.r2.si and line["len1"] == r.r1.ei - r.r1.si
        assert line["r2"] == refio.mate_fields(r.r2) and line["r1"] == refio.mate_fields(r.r1)
        assert line["annot2"] == refio.annot_str(r.r2.annot()) and line["annot1"] == refio.annot_str(r.r1.annot())
        assert line["title"] == reads.titles[r.pair][1:]
        assert line["seq2"] == reads.seqs[2 * r.pair + 1].decode() and line["seq1"] == reads.seqs[2 * r.pair].decode()


@pytest.mark.parametrize("name", sorted(GOLD))
def test_oracle_reproduces_reference_binary_outputs(name, O):
    """End to end: same files in, the reference binary's bytes out
    (.trkmc.ar incl. unordered_map order, summary, totals, kam fields)."""
    d, p, reads, qc = golden_inputs(name)
    g = O.load(os.path.join(d, "pan"), p.ksize, qc)
    seq, off = reads.packed()
    p.trace = 1
    a = O.align(g, p, seq, off)
    # output order from the oracle's own restatement of libstdc++'s hashtable
    tr = synth.read_rpgg_files(os.path.join(d, "pan"))
    out, i = [], 0
    for n in tr["tr_cnt"]:
        n = int(n)
        order = O.umap_order(tr["tr_ks"][i:i + n])
        out.append(a["counts_file"][i:i + n][order.astype(np.int64)])
        i += n
    check_against_golden(d, np.concatenate(out), a["kmc"], a["nmapread"], a["counters"], a["recs"], reads)
    # -on text: k-mer names in iteration order
    names = [int(l.split()[0]) for l in open(os.path.join(d, "refon.tr.kmers")) if l[0] != ">"]
    i, mine = 0, []
    for n in tr["tr_cnt"]:
        n = int(n)
        mine += [int(x) for x in tr["tr_ks"][i:i + n][O.umap_order(tr["tr_ks"][i:i + n]).astype(np.int64)]]
        i += n
    assert names == mine
    O.free(g)
