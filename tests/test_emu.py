"""The kernel bodies (danbing-tk_amd/csrc/dbtk_kernels.h — the code hipcc
compiles for gfx950) run on the test-only SPMD emulator and compared with the
oracle.  This is how the device logic is exercised where no GPU exists; the GPU
suite (test_gpu_parity.py) repeats the same cases on hardware."""
import os

import numpy as np
import pytest

import bind
import cases
from test_oracle import GOLD, check_against_golden, golden_inputs

abi = bind.abi


@pytest.fixture(scope="module")
def O():
    return bind.Oracle()


@pytest.fixture(scope="module")
def E():
    return bind.Emu()


@pytest.mark.parametrize("case", sorted(cases.CASES))
def test_kernel_bodies_match_oracle(case, O, E, tmp_path):
    c = cases.make_case(case, str(tmp_path))
    go = O.load(c.prefix, c.k, c.qc_file)
    g = E.load(c.prefix, c.k, c.qc_file)
    T = E.tables(g)
    order = g.output_order().astype(np.int64)
    seq, off = c.reads.packed()
    # the load-time check: index memberships == flank/TR sets  <=>  the class rides in the index slot
    assert E.consistent(T) == (0 if case == "inconsistent" else 1)
    E.probe_stats()
    E.locus_stats()
    E.path_stats()
    for i, kw in enumerate(c.param_sets):
        p = abi.default_params(ksize=c.k, trace=1, **kw)
        a = O.align(go, p, seq, off)
        co = np.zeros(g.ntrkmers, np.uint64)
        np.add.at(co, order, a["counts_file"])
        for mode in ((1, 0) if E.consistent(T) else (0,)):  # fused-aux path and the general class-table path
            E.set_consistent(T, mode)
            b = E.align(g, T, p, seq, off, grid_k1=1 + i, grid_pair=2 + 3 * i)
            assert (co == b["counts"]).all()
            assert (a["kmc"] == b["kmc"]).all() and (a["nmapread"] == b["nmapread"]).all()
            assert (a["counters"] == b["counters"]).all(), (a["counters"], b["counters"])
            d = bind.recs_equal(a["recs"], b["recs"], c.reads.npairs)
            assert d < 0, f"{bind.rec_str(a['recs'][d])}\n{bind.rec_str(b['recs'][d])}"
        E.set_consistent(T, 1 if case != "inconsistent" else 0)
        # without trace the usual pair (one locus, >= cth hits per mate) skips the sort and the vote: same counts and totals
        p2 = abi.default_params(ksize=c.k, **kw)
        a2 = O.align(go, p2, seq, off, trace=False)
        co2 = np.zeros(g.ntrkmers, np.uint64)
        np.add.at(co2, order, a2["counts_file"])
        # (the probe bodies that resolve pairs themselves: both — the default —, the locus-resident one alone, the lean one alone, neither)
        # ("lean": no locus-resident body at all — a batch with few survivors per locus — so that the lean body sees every pair,
        # those with k-mers shared between loci included)
        for fuse_bits in (None, "1", "2", "0", "lean"):
            os.environ.pop("EMU_FUSE", None)
            os.environ.pop("EMU_NO_LOCUS", None)
            if fuse_bits == "lean":
                os.environ["EMU_NO_LOCUS"] = "1"
            elif fuse_bits is not None:
                os.environ["EMU_FUSE"] = fuse_bits
            try:
                b2 = E.align(g, T, p2, seq, off, grid_k1=11, grid_pair=3)
            finally:
                os.environ.pop("EMU_FUSE", None)
                os.environ.pop("EMU_NO_LOCUS", None)
            assert (co2 == b2["counts"]).all(), fuse_bits
            assert (a2["kmc"] == b2["kmc"]).all() and (a2["nmapread"] == b2["nmapread"]).all(), fuse_bits
            assert (a2["counters"] == b2["counters"]).all(), (fuse_bits, a2["counters"], b2["counters"])
    general, lean, turned = E.probe_stats()  # the probe body the case is meant to exercise is the one that ran
    assert (lean > 0 and general == 0) if case in cases.LEAN_PROBE else (general > 0 and lean == 0), (general, lean)
    if case in ("shared", "spill"):
        assert turned > 0  # keys in the overflow table: level 2 of the lean body's look-ups is exercised
    cls, rest, left_out = E.locus_stats()
    print(f"{case}: locus-resident body {cls} pairs, lean body {rest}; {left_out} keys left out of the images")
    assert left_out == 0  # (hash and displace places every key at these loads)
    # (images are only built for an RPGG whose sets agree with its index; "spill" has too few survivors per locus for a list in locus order)
    if case in cases.LEAN_PROBE and case not in ("inconsistent", "spill"):
        assert sum(cls) > 0 and rest > 0  # both the image path and the hand-over to the global tables ran
    if case in ("shared", "k25"):
        assert cls[0 if case == "shared" else 1] > 0  # different classes of workgroup are exercised
    ps = E.path_stats()
    print(f"{case}: fused resolve: {ps['fused_done']} pairs ({ps['fused_shared']} with k-mers shared between loci), {ps['fused_redone']} taken back")
    if case == "shared":
        assert ps["fused_shared"] > 0  # pairs with shared k-mers decided by the k-mers unique to the locus: no sort, no vote
    if case in ("clean", "shared", "qc", "spliced", "k25"):
        assert ps["fused_done"] > 0   # the no-trace, no-record runs resolve the usual pairs inside the locus-resident probe body
        assert ps["lean_done"] > 0    # ... and inside the lean one
    if case == "spliced":
        assert ps["fused_redone"] > 20  # ... and take back the ones that have a k-mer of the index outside the image
    if case in ("shared", "clean"):
        # the lean body alone (no locus-resident body: a batch with few survivors per locus): it resolves its usual pairs itself, and in
        # "shared" also pairs with k-mers shared between loci (the class of such a k-mer at the locus from the class table)
        os.environ["EMU_NO_LOCUS"] = "1"
        try:
            E.align(g, T, abi.default_params(ksize=c.k, **{**c.param_sets[0], "okam": 0}), seq, off, grid_k1=9, grid_pair=3)
        finally:
            os.environ.pop("EMU_NO_LOCUS", None)
        ps = E.path_stats()
        print(f"{case}: lean body alone: {ps['lean_done']} pairs resolved, {ps['fused_shared']} with shared k-mers")
        assert ps["lean_done"] > 0 and ps["fused_done"] == 0
        if case == "shared":
            assert ps["fused_shared"] > 0
    if case == "shared":
        # the vv words of the VOTE (find_matching_locus) are a path statistic, not a counter: the reference votes on every pair, the kernels
        # prove most pairs' outcome without one (the fused probe bodies, and — round 6 — body_pair itself).  In trace mode every vote is held
        # (the records want its partial sums) and the two figures are equal; without, the device's is smaller — and the counter
        # DBTK_C_ALGO_VV (fillstats' words) is the oracle's on every path (asserted above with the counters).
        for tr, same in ((1, True), (0, False)):
            p3 = abi.default_params(ksize=c.k, **{**c.param_sets[0], "okam": 0, "trace": tr})
            a3 = O.align(go, p3, seq, off, trace=bool(tr))
            E.path_stats()
            E.align(g, T, p3, seq, off, grid_k1=2, grid_pair=3)
            ps = E.path_stats()
            fuse_bits = "trace" if tr else "no trace"
            assert a3["vote_vv"] > 0 and (ps["vote_vv"] == a3["vote_vv"] if same else ps["vote_vv"] < a3["vote_vv"]), (fuse_bits, ps["vote_vv"], a3["vote_vv"])
    E.L.emu_tables_free(T)
    O.free(go)
    g.close()


@pytest.mark.skipif(not __import__("synth").have_ref(), reason="needs oracle/_ref (ktools serialize-bt builds the bait DB)")
@pytest.mark.parametrize("fastq", [False, True])
def test_bait_and_bubble_kernels_match_oracle(O, E, tmp_path, fastq):
    import synth
    loci = synth.make_loci(nloci=10, nhap=3, flank=500, seed=61, shared_frac=0.3)
    d = str(tmp_path)
    pref = synth.build_rpgg_with_reference(loci, d, k=21)
    reads = synth.sim_reads(loci, npairs=700, seed=62, sub=0.01, indel=0.002, nrate=0.002, chimeric=0.3, background=0.1, with_qual=fastq)
    bait = synth.make_bait_db(loci, reads, d)
    go = O.load(pref, 21)
    O.load_bait(go, bait)
    g = E.load(pref, 21, bait_file=bait)
    T = E.tables(g)
    order = g.output_order().astype(np.int64)
    seq, off = reads.packed()
    qual = np.frombuffer(b"".join(reads.quals), np.uint8).copy() if fastq else None
    for kw in (dict(bait=1, bubbles=1, cthreshold=45), dict(bait=1, cthreshold=20, okam=0), dict(bubbles=1, cthreshold=30, simmode=2)):
        p = abi.default_params(ksize=21, trace=1, **kw)
        a = O.align_ex(go, p, seq, off, qual)
        b = E.align_ex(g, T, p, seq, off, qual)
        co = np.zeros(g.ntrkmers, np.uint64)
        np.add.at(co, order, a["counts_file"])
        assert (co == b["counts"]).all() and (a["kmc"] == b["kmc"]).all() and (a["nmapread"] == b["nmapread"]).all()
        assert (a["counters"] == b["counters"]).all(), (a["counters"], b["counters"])
        dd = bind.recs_equal(a["recs"], b["recs"], reads.npairs)
        assert dd < 0, f"{bind.rec_str(a['recs'][dd])}\n{bind.rec_str(b['recs'][dd])}"
        if p.bubbles:
            assert len(a["events"]) == len(b["events"]) and (a["events"] == b["events"]).all()
    E.L.emu_tables_free(T)
    O.free(go)
    g.close()


def test_device_sort_is_gcc_std_sort(E):
    """dbtk_sort.h (index form and packed form) against this host's std::sort, incl. the heapsort fallback."""
    assert E.selftest_sort(3, 20000) == 0


def test_wave_sort_is_gcc_std_sort(E):
    """The wave-parallel introsort of the general resolve kernel vs this host's std::sort (ties, sorted, reversed, killer sequences)."""
    assert E.selftest_wavesort(5, 4000) == 0


def test_wave_vote_is_find_matching_locus(E):
    """The scan form of the general vote vs the literal find_matching_locus loop, on tie-heavy random event streams."""
    assert E.selftest_vote(11, 30000) == 0


def test_assign_bits_equals_literal_scan(E):
    """The mask form of assignTRkmc used by the kernels vs the literal restatement of AQ.cpp:1470-1555."""
    assert E.selftest_assign(7, 400000) == 0


@pytest.mark.parametrize("name", sorted(GOLD))
def test_kernel_bodies_reproduce_reference_binary(name, E):
    d, p, reads, qc = golden_inputs(name)
    import os
    g = E.load(os.path.join(d, "pan"), p.ksize, qc)
    T = E.tables(g)
    seq, off = reads.packed()
    p.trace = 1
    b = E.align(g, T, p, seq, off)
    check_against_golden(d, b["counts"], b["kmc"], b["nmapread"], b["counters"], b["recs"], reads)
    E.L.emu_tables_free(T)
    g.close()


@pytest.mark.parametrize("name", sorted(GOLD))
def test_encode_stage_lazy_sampling(name, E, monkeypatch):
    """body_encode_subfilter<true> (k_encode_subfilter_lazy, the form the launcher picks when most of the batch before passed subfilter: a
    mate's first sample alone, the other three only if that one is not in the index) against the bytes the reference binary wrote; the
    emulator also checks the sort keys it hands over against body_surv_key's."""
    monkeypatch.setenv("EMU_K1_LAZY", "1")
    d, p, reads, qc = golden_inputs(name)
    g = E.load(os.path.join(d, "pan"), p.ksize, qc)
    T = E.tables(g)
    seq, off = reads.packed()
    p.trace = 1
    for grid in (3, 9):
        b = E.align(g, T, p, seq, off, grid_k1=grid)
        check_against_golden(d, b["counts"], b["kmc"], b["nmapread"], b["counters"], b["recs"], reads)
    E.L.emu_tables_free(T)
    g.close()


def test_wave_formatters_are_writeCigar_and_writeAnnot(E):
    """The whole-wave CIGAR / annotation formatters of the -a / -ae walk (dbtk_walk.h: wave_fmt_*) vs the one-lane scans that restate
    writeCigar / writeAnnot (AQ.cpp:1683-1740; pinned to the reference's strings in tests/test_walk.py): runs across the 64-entry chunks,
    D / I stretches of both parities, the "1" + type a token starting at the last entry gets from the code behind writeCigar's loop."""
    assert E.selftest_fmt(3, 20000) == 0
