"""GPU parity: the HIP path (through the C-ABI of libdbtk_hip.so) against the
oracle on the same seeded inputs.  Integer outputs must be bit-exact: counts in
OUT.trkmc.ar order, kmc, nmapread, the reference's counters, and every field
of every per-pair record."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import bind
import synth
from cases import CASES, make_case

abi = bind.abi
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dbtk():
    return bind.pkg.Dbtk()


@pytest.fixture(scope="module")
def oracle():
    return bind.Oracle()


def compare(o, g, order, ntr, npairs, recs=True):
    co = np.zeros(ntr, np.uint64)
    np.add.at(co, order.astype(np.int64), o["counts_file"])
    assert (co == g["counts"]).all(), f"{int((co != g['counts']).sum())} k-mer counts differ"
    assert (o["kmc"] == g["kmc"]).all()
    assert (o["nmapread"] == g["nmapread"]).all()
    assert (o["counters"] == g["counters"]).all(), (o["counters"], g["counters"])
    if recs:
        d = bind.recs_equal(o["recs"], g["recs"], npairs)
        assert d < 0, f"record {d}:\n  oracle {bind.rec_str(o['recs'][d])}\n  hip    {bind.rec_str(g['recs'][d])}"


@pytest.mark.parametrize("case", sorted(CASES))
def test_hip_matches_oracle(case, dbtk, oracle, tmp_path):
    c = make_case(case, str(tmp_path))
    go = oracle.load(c.prefix, c.k, c.qc_file)
    g = dbtk.load(c.prefix, c.k, c.qc_file)
    order = g.output_order()
    seq, off = c.reads.packed()
    for kw in c.param_sets:
        p = abi.default_params(ksize=c.k, trace=1, **kw)
        o = oracle.align(go, p, seq, off)
        ctx = dbtk.context(g, p)
        recs, nrec = ctx.align(seq, off)
        res = ctx.counts()
        res["recs"] = recs
        assert nrec == c.reads.npairs
        compare(o, res, order, g.ntrkmers, c.reads.npairs)
        # kam mode (no trace): only counted pairs, in pair order
        p2 = abi.default_params(ksize=c.k, **kw)
        if p2.okam or p2.extract:
            ctx2 = dbtk.context(g, p2)
            recs2, n2 = ctx2.align(seq, off)
            want = [r for r in o["recs"] if r.stage in (abi.STAGE_COUNTED, abi.STAGE_EXTRACT)]
            assert n2 == len(want)
            sz = C.sizeof(abi.PairRec)
            for i, r in enumerate(want):
                w = abi.PairRec.from_buffer_copy(bytes(r)[:sz])
                w.nm1 = w.nm2 = 0  # the vote's partial sums are reported in trace mode only
                assert bytes(recs2[i]) == bytes(w)
            # without trace the usual pair skips the sort and the vote: same counts, same totals
            res2 = ctx2.counts()
            compare(o, res2, order, g.ntrkmers, 0, recs=False)
            ctx2.close()
        p3 = abi.default_params(ksize=c.k, **dict(kw, okam=0))
        ctx3 = dbtk.context(g, p3)
        ctx3.align(seq, off)  # no record buffer at all: the record-free kernel variant
        compare(oracle.align(go, p3, seq, off, trace=False), ctx3.counts(), order, g.ntrkmers, 0, recs=False)
        ctx3.close()
        ctx.close()
    oracle.free(go)
    g.close()


def test_index_image_sidecar_round_trip(dbtk, oracle, tmp_path, monkeypatch):
    """SURVEY 8 f2, second half: the GPU-layout index (the per-locus images the probe kernel keeps in LDS) written to
    PREF.dbtk.idx and loaded from it: same results whether the images were built, loaded, or rebuilt because the file is
    of another RPGG / damaged / of another layout version; the file is only written when asked for."""
    monkeypatch.setenv("DBTK_LOCUS_ALWAYS", "1")
    c = make_case("shared", str(tmp_path))
    go = oracle.load(c.prefix, c.k, c.qc_file)
    seq, off = c.reads.packed()
    p = abi.default_params(ksize=c.k, trace=1, **c.param_sets[0])
    o = oracle.align(go, p, seq, off)
    side = c.prefix + ".dbtk.idx"

    def run(mode=None, expect_cache=None):
        g = dbtk.load(c.prefix, c.k, c.qc_file)
        if mode is not None:
            g.set_index_cache(side, mode)
        ctx = dbtk.context(g, p)
        tb = ctx.table_bytes()
        assert tb["index_images"] > 0 and tb["total"] >= tb["index"] + tb["index_images"]
        if expect_cache is not None:
            assert tb["index_images:from_cache"] == expect_cache
        recs, nrec = ctx.align(seq, off)
        res = ctx.counts()
        res["recs"] = recs
        compare(o, res, g.output_order(), g.ntrkmers, c.reads.npairs)
        ctx.close(); g.close()

    run(expect_cache=0)                      # default mode 1: nothing to load, and nothing is written
    assert not os.path.exists(side)
    run(mode=2, expect_cache=0)              # built and written
    assert os.path.getsize(side) > 1000
    run(expect_cache=1)                      # the default now finds it
    good = open(side, "rb").read()
    blob = bytearray(good)
    for at in (len(blob) // 2, len(blob) - 9, 200):  # one flipped bit in an image (a tag, a pay word, a displacement byte), in the directory
        blob[at] ^= 0x20
        open(side, "wb").write(bytes(blob))
        run(expect_cache=0)                  # the checksum does not match: rebuilt, never used
        blob[at] ^= 0x20
    # a file that is whole (its checksum holds) but is not THIS index's: the images of an RPGG whose TR k-mers have other counters.  The
    # fingerprint is defeated on purpose (the one of this RPGG copied in); what refuses the file is the entry-by-entry look-up in the index.
    import struct
    hdr_fmt = "<8sIIQQQQQQIIQ"
    hdr = list(struct.unpack_from(hdr_fmt, good, 0))
    assert hdr[0] == b"DBTKIDX\x01" and hdr[10] == struct.calcsize(hdr_fmt)
    body = bytearray(good[hdr[10]:])
    nloci = hdr[3]
    dirs = [struct.unpack_from("<IIII", body, 16 * l) for l in range(nloci)]  # {off16, bytes, lgnb, trbeg}
    first = next(d for d in dirs if d[1])
    arena0 = 16 * nloci + 16 * first[0]
    # swap the pay words of two TR entries of the first image that differ (another counter for the same k-mer), keep the file's checksum right
    lg = struct.unpack_from("<I", body, arena0)[0]
    pays = [(b * 32 + 16 + 4 * s_, struct.unpack_from("<I", body, arena0 + 16 + b * 32 + 16 + 4 * s_)[0]) for b in range(1 << lg) for s_ in range(4)]
    tr = [(o_, v) for o_, v in pays if v != 0xFFFFFFFE and (v >> 21) & 7 == 1]
    (o1, v1), (o2, v2) = tr[0], next(t for t in tr[1:] if (t[1] & 0x1FFFFF) != (tr[0][1] & 0x1FFFFF))
    struct.pack_into("<I", body, arena0 + 16 + o1, (v1 & ~0x1FFFFF) | (v2 & 0x1FFFFF))
    struct.pack_into("<I", body, arena0 + 16 + o2, (v2 & ~0x1FFFFF) | (v1 & 0x1FFFFF))

    def csum(words, base=0):
        m = (1 << 64) - 1
        tot = 0
        for i, w in enumerate(words):
            v = w ^ (((i + 1) * 0xD6E8FEB86659FD93) & m)
            v ^= v >> 32; v = (v * 0x9E3779B97F4A7C15) & m; v ^= v >> 29
            tot = (tot + v) & m
        return tot
    dirw = struct.unpack_from(f"<{2 * nloci}Q", body, 0)
    arw = struct.unpack_from(f"<{hdr[6] // 8}Q", body, 16 * nloci)
    hdr[11] = (csum(arw) + ((csum(dirw) * 0x9E3779B97F4A7C15) & ((1 << 64) - 1)) + hdr[6]) & ((1 << 64) - 1)
    forged = struct.pack(hdr_fmt, *hdr) + bytes(body)
    assert struct.unpack_from(hdr_fmt, good, 0)[11] == (csum(struct.unpack_from(f"<{hdr[6] // 8}Q", good, hdr[10] + 16 * nloci)) + ((csum(dirw) * 0x9E3779B97F4A7C15) & ((1 << 64) - 1)) + hdr[6]) & ((1 << 64) - 1)
    open(side, "wb").write(forged)
    run(expect_cache=0)                      # checksum fine, fingerprint fine, two counters swapped: refused by the look-up in the index
    open(side, "wb").write(good)
    run(expect_cache=1)
    blob = bytearray(good)
    blob[33] ^= 0xFF                         # the fingerprint: not this RPGG's file
    open(side, "wb").write(bytes(blob))
    run(expect_cache=0)
    open(side, "wb").write(bytes(blob[:100]))  # truncated
    run(expect_cache=0)
    run(mode=2, expect_cache=0)              # rewritten
    run(expect_cache=1)
    c2 = make_case("clean", str(tmp_path / "other"))   # another RPGG pointed at this file: rebuilt, not trusted
    g2 = dbtk.load(c2.prefix, c2.k, c2.qc_file)
    g2.set_index_cache(side, 1)
    ctx2 = dbtk.context(g2, abi.default_params(ksize=c2.k))
    assert ctx2.table_bytes()["index_images:from_cache"] == 0
    ctx2.close(); g2.close()
    oracle.free(go)


@pytest.mark.skipif(not synth.have_ref(), reason="needs oracle/_ref (ktools serialize-bt builds the bait DB)")
@pytest.mark.parametrize("fastq", [False, True])
def test_bait_and_bubble_gates_match_oracle(dbtk, oracle, tmp_path, fastq):
    """-b (bfilter_FPSv1 + qString2qMask) and -bu (countNovelEdges) through the C-ABI."""
    loci = synth.make_loci(nloci=10, nhap=3, flank=500, seed=61, shared_frac=0.3)
    d = str(tmp_path)
    pref = synth.build_rpgg_with_reference(loci, d, k=21)
    reads = synth.sim_reads(loci, npairs=1500, seed=62, sub=0.01, indel=0.002, nrate=0.002, chimeric=0.3, background=0.1, with_qual=fastq)
    bait = synth.make_bait_db(loci, reads, d)
    go = oracle.load(pref, 21)
    oracle.load_bait(go, bait)
    g = dbtk.load(pref, 21, bait_file=bait)
    seq, off = reads.packed()
    qual = np.frombuffer(b"".join(reads.quals), np.uint8).copy() if fastq else None
    for kw in (dict(bait=1, bubbles=1, cthreshold=45), dict(bait=1, cthreshold=20, okam=0), dict(bubbles=1, cthreshold=30, simmode=2)):
        p = abi.default_params(ksize=21, trace=1, **kw)
        o = oracle.align_ex(go, p, seq, off, qual)
        ctx = dbtk.context(g, p)
        recs, nrec = ctx.align(seq, off, qual=qual)
        res = ctx.counts()
        res["recs"] = recs
        compare(o, res, g.output_order(), g.ntrkmers, reads.npairs)
        if p.bait:
            assert res["counters"][abi.C_BAITFILTERED] > 0
        if p.bubbles:
            ctx.write_bubbles(str(tmp_path / "o"))
            a = np.fromfile(str(tmp_path / "o.bub.kmdb"), np.uint64)
            nl = int(a[0]); nk = int(a[1 + nl])
            assert nl == g.nloci and a[2 + nl] == 8 and len(a) == 3 + nl + 2 * nk and (a[3 + nl + nk:] >= 5).all()
            # every dumped (locus, edge) has exactly the oracle's count
            ev = o["events"]
            want = {}
            for l, e in zip(ev["locus"], ev["edge"]):
                want[(int(l), int(e))] = want.get((int(l), int(e)), 0) + 1
            want = {k: v for k, v in want.items() if v >= 5}
            got, i = {}, 0
            for l in range(nl):
                for _ in range(int(a[1 + l])):
                    got[(l, int(a[3 + nl + i]))] = int(a[3 + nl + nk + i]); i += 1
            assert got == want
        ctx.close()
    oracle.free(go)
    g.close()


def test_batches_accumulate_and_split_invariance(dbtk, oracle, tmp_path):
    """Batch boundaries do not change results (AQ.cpp:1918-1976: all effects additive)."""
    c = make_case("mixed", str(tmp_path))
    g = dbtk.load(c.prefix, c.k)
    go = oracle.load(c.prefix, c.k)
    seq, off = c.reads.packed()
    p = abi.default_params(ksize=c.k, cthreshold=45, okam=0)
    o = oracle.align(go, p, seq, off)
    ctx = dbtk.context(g, p)
    n = c.reads.npairs
    cuts = [0, 1, 7, n // 3, n // 3, n - 1, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        ctx.align(seq, off[2 * a:2 * b + 1])
    res = ctx.counts()
    compare(o, res, g.output_order(), g.ntrkmers, n, recs=False)
    ctx.close()


def test_empty_and_degenerate_batches(dbtk, tmp_path):
    c = make_case("clean", str(tmp_path))
    g = dbtk.load(c.prefix, c.k)
    p = abi.default_params(ksize=c.k, trace=1)
    ctx = dbtk.context(g, p)
    recs, n = ctx.align(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert n == 0
    # empty reads, reads shorter than k, all-N reads
    reads = synth.Reads()
    reads.seqs = [b"", b"", b"ACGT", b"ACGTACGTAC", b"N" * 150, b"A" * 150, b"A" * 150, b"N" * 150]
    reads.titles = ["a", "b", "c", "d"]
    seq, off = reads.packed()
    recs, n = ctx.align(seq, off)
    assert n == 4 and all(recs[i].stage == abi.STAGE_SHORT for i in range(4))
    res = ctx.counts()
    assert res["counters"][abi.C_NSHORT] == 4 and res["counters"][abi.C_NREADS] == 8
    assert res["counts"].sum() == 0
    with pytest.raises(bind.pkg.DbtkError) as e:
        ctx.align(np.frombuffer(b"A" * 300 + b"C" * 10, np.uint8), np.array([0, 300, 310], np.uint64))
    assert e.value.status == abi.ERR_READ_TOO_LONG
    ctx.close()


def test_accumulator_as_torch_tensor_and_rccl(tmp_path):
    """bench.py reduces the accumulator in place: a torch tensor over the context's device buffer through RCCL (1 rank).
    In a child process, torch imported first as in bench.py (its HIP runtime must be the one libdbtk_hip.so binds to)."""
    import subprocess
    import sys
    c = make_case("clean", str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_torch_accum.py"), c.prefix, str(c.k)],
                       capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517"))
    assert r.returncode == 0 and "ACCUM-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("lanes", [1, 2, 3])
def test_device_entry_alternating_lanes(dbtk, oracle, tmp_path, lanes, monkeypatch):
    """dbtk_align_batch_device (reads resident in HBM, asynchronous): five batches, alternating between the context's two
    streams when DBTK_LANES=2, must add up to five times the oracle's single-batch result."""
    monkeypatch.setenv("DBTK_LANES", str(lanes))
    c = make_case("mixed", str(tmp_path))
    go = oracle.load(c.prefix, c.k, c.qc_file)
    g = dbtk.load(c.prefix, c.k, c.qc_file)
    seq, off = c.reads.packed()
    p = abi.default_params(ksize=c.k, **dict(c.param_sets[0], okam=0))
    o = oracle.align(go, p, seq, off, trace=False)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    d_seq, d_off = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d_seq), len(seq) + 64) == 0 and hip.hipMalloc(C.byref(d_off), off.nbytes) == 0
    assert hip.hipMemcpy(d_seq, seq.ctypes.data_as(C.c_void_p), len(seq), 1) == 0
    assert hip.hipMemcpy(d_off, off.ctypes.data_as(C.c_void_p), off.nbytes, 1) == 0
    ctx = dbtk.context(g, p)
    maxlen = int(np.diff(off.astype(np.int64)).max())
    for _ in range(5):
        ctx.align_device(d_seq.value, d_off.value, c.reads.npairs, maxlen)
    ctx.synchronize()
    r = ctx.counts()
    co = np.zeros(g.ntrkmers, np.uint64)
    np.add.at(co, g.output_order().astype(np.int64), o["counts_file"])
    assert (5 * co == r["counts"]).all()
    assert (5 * o["kmc"] == r["kmc"]).all() and (5 * o["nmapread"].astype(np.uint64) == r["nmapread"]).all()
    assert (5 * o["counters"] == r["counters"]).all(), (o["counters"], r["counters"])
    ctx.close()
    hip.hipFree(d_seq); hip.hipFree(d_off)
    oracle.free(go)
    g.close()



def dense_slice_checks(monkeypatch, dbtk, oracle, syn, arrs, g, orc_g, k):
    """the locus-resident kernels (dbtk_locus.h: k_probe_locus; dbtk_walkfast.h: k_walk_fast_locus) against the ORACLE on
    the release-scale RPGG.  A uniform draw over 80 000 loci gives ~1 pair per locus and never reaches them; here ~1 400 loci of every
    image class (every locus of the largest class) get 64 pairs each, 6 % of the pairs chimeric / foreign.  Device counters assert that
    the locus path took >= 90 % of the pairs (87 % at k = 25) and every class; counts, kmc, nmapread, all counters, the trace records, the walk results and
    the -ae text are the oracle's."""
    loci, classes = bind.dense_loci(arrs, k, per_lg=650, seed=11 + k)
    assert len(loci) >= 600 and 2 in classes and 1 in classes and (k > 22 or 0 in classes), (len(loci), classes)
    n = 64 * len(loci)
    seq, off = syn.reads_loci(n, loci, odd_frac=0.06, seed=100 + k)
    monkeypatch.setenv("DBTK_LOCUS_ALWAYS", "1")  # (read at context creation: every batch is sorted and offered to the locus path)
    order, ntr = g.output_order(), g.ntrkmers
    base = dict(ksize=k, n_filter=4, nm_filter=1, cthreshold=45, okam=0)
    # 1. counting (the resolve kernels' regime), then with trace records (every pair through the general resolve kernel)
    for trace in (0, 1):
        p = abi.default_params(trace=trace, **base)
        o = oracle.align(orc_g, p, seq, off, trace=bool(trace))
        ctx = dbtk.context(g, p)
        recs, nrec = ctx.align(seq, off)
        res = ctx.counts()
        res["recs"] = recs
        compare(o, res, order, ntr, n, recs=bool(trace))
        ps = ctx.path_stats()
        surv = int(res["counters"][abi.C_SURVIVORS])
        assert surv > 0.99 * n
        # (6 % of the pairs are not their list locus'; a pair more than 96 of whose positions miss the image is handed to the lean kernel too:
        # at k = 25 an error costs 25 positions — measured: 6.9 % of the pairs at k = 21, 10.1 % at k = 25)
        assert surv - ps["probe_rest"] >= (0.9 if k < 23 else 0.87) * surv, (ps, surv)
        for c in classes:
            assert ps["probe_items"][c] > 0 and ps["probe_pairs"][c] >= 16 * ps["probe_items"][c], (c, ps)
        ctx.close()
    # 2. the graph walk (-gc 85 3): counts, counters, walk results; then -ae text on a part of the slice
    pw = abi.default_params(threading=abi.THREADING_V13, thread_cth=85, correction=1, maxncorrection=3, **base)
    ow = oracle.align_walk(orc_g, pw, seq, off, with_recs=False)
    cw = dbtk.context(g, pw)
    cw.align(seq, off)
    r = cw.counts()
    co = np.zeros(ntr, np.uint64)
    np.add.at(co, order.astype(np.int64), ow["counts_file"])
    assert (co == r["counts"]).all() and co.sum() > 0 and (ow["counters"] == r["counters"]).all(), (ow["counters"], r["counters"])
    wres, _, nres = cw.walk_results(n)
    assert nres == ow["nres"] and bind.walk_res_equal(wres, ow["res"], nres, g.nloci, every_mate=False) > 0
    ps = cw.path_stats()
    walked = sum(ps["walk_pairs"])
    assert walked >= 0.85 * nres and all(ps["walk_items"][c] > 0 for c in classes), (ps, nres)
    cw.close()
    n2 = min(n, 12_000)
    pa = abi.default_params(threading=abi.THREADING_V13, thread_cth=85, correction=1, maxncorrection=3, aln=2 | abi.ALN_TEXT, **base)
    pa_o = abi.default_params(threading=abi.THREADING_V13, thread_cth=85, correction=1, maxncorrection=3, aln=2, **base)
    s2, o2 = seq[:int(off[2 * n2])], off[:2 * n2 + 1]
    oa = oracle.align_walk(orc_g, pa_o, s2, o2)
    exp = []
    for i in range(oa["nres"]):
        w = oa["res"][i]
        if w.dst == g.nloci:
            continue
        c1, a1 = oracle.cigar_annot(oa["trecs"][2 * i])
        c2, a2 = oracle.cigar_annot(oa["trecs"][2 * i + 1])
        exp.append((w.pair, w.dst, f"{c2}\t{a2}\t{c1}\t{a1}"))
    ca = dbtk.context(g, pa)
    ca.align(s2, o2)
    assert ca.aln_text(n2) == exp and len(exp) > n2 // 2
    ra = ca.counts()
    co = np.zeros(ntr, np.uint64)
    np.add.at(co, order.astype(np.int64), oa["counts_file"])
    assert (co == ra["counts"]).all() and (oa["counters"] == ra["counters"]).all()
    ca.close()
    monkeypatch.delenv("DBTK_LOCUS_ALWAYS")


def test_release_scale_properties(dbtk, oracle, monkeypatch):
    """BASELINE config 2 at its full size: the bench's release-scale synthetic RPGG (80 000 loci, 1.4e8 index keys) and
    10 M reads (2 % of the pairs from the loci).  Size-independent properties — additivity over batch splits, the
    counters' conservation laws, sum of counts == counted increments — plus the oracle itself on a 100 000-pair slice of
    that batch, on a 100 000-pair ALL-HIT slice (every pair from a locus: the probe / resolve kernels' regime), and on an
    all-hit slice through the graph walk (threading = 2, -gc 85 3)."""
    syn = bind.pkg.Synth(nloci=80000)
    syn.graph()
    arrs = syn.arrays()
    h = C.c_void_p()
    dbtk._chk(dbtk.L.dbtk_rpgg_from_arrays(C.byref(arrs), C.byref(h)))
    g = bind.pkg.Rpgg(dbtk, h)
    npairs = 5_000_000
    seq, off = syn.reads(npairs, hit_frac=0.02)
    p = abi.default_params(ksize=21, n_filter=4, nm_filter=1, cthreshold=45, okam=0)
    whole = dbtk.context(g, p)
    whole.align(seq, off)
    w = whole.counts()
    parts = dbtk.context(g, p)
    cuts = [0, 1, 4097, 2_500_000, 2_500_016, npairs]
    for a, b in zip(cuts[:-1], cuts[1:]):
        base = int(off[2 * a])
        parts.align(seq[base:int(off[2 * b])], off[2 * a:2 * b + 1] - np.uint64(base))
    q = parts.counts()
    for k in ("counts", "kmc", "nmapread", "counters"):
        assert (w[k] == q[k]).all(), k
    parts.close()
    ctr = w["counters"]
    assert ctr[abi.C_NREADS] == 2 * npairs
    assert int(w["counts"].sum()) == int(ctr[abi.C_ALGO_INC])          # every increment lands in exactly one counter
    assert int(w["nmapread"].astype(np.uint64).sum()) == int(ctr[abi.C_ASGN])  # reads counted per locus == reads assigned
    # the funnel: every read is short, subfiltered, k-mer-filtered, locus-filtered, QC-filtered or enters threading
    funnel = 2 * int(ctr[abi.C_NSHORT]) + sum(int(ctr[i]) for i in (abi.C_SUBFILTERED, abi.C_KMERFILTERED, abi.C_LOCUSFILTERED, abi.C_QCFILTERED, abi.C_THREADING))
    assert funnel == 2 * npairs, (funnel, ctr)
    assert ctr[abi.C_ASGN] <= ctr[abi.C_FEASIBLE] <= ctr[abi.C_THREADING]
    # the oracle on a slice of the same batch
    n = 100_000
    orc_g = oracle.from_arrays(arrs)
    o = oracle.align(orc_g, p, seq[:int(off[2 * n])], off[:2 * n + 1], trace=False)
    whole.reset()
    whole.align(seq[:int(off[2 * n])], off[:2 * n + 1])
    compare(o, whole.counts(), g.output_order(), g.ntrkmers, 0, recs=False)
    # all-hit slice: every pair from a locus
    hseq, hoff = syn.reads(n, hit_frac=1.0, seed=2)
    o = oracle.align(orc_g, p, hseq, hoff, trace=False)
    assert o["counters"][abi.C_THREADING] > n and o["counters"][abi.C_ASGN] > n // 20
    whole.reset()
    whole.align(hseq, hoff)
    compare(o, whole.counts(), g.output_order(), g.ntrkmers, 0, recs=False)
    whole.close()
    # the same all-hit slice through the graph walk (config 4's flags on this RPGG)
    n2 = 30_000
    pw = abi.default_params(ksize=21, n_filter=4, nm_filter=1, cthreshold=45, okam=0, threading=abi.THREADING_V13, thread_cth=85,
                            correction=1, maxncorrection=3)
    ow = oracle.align_walk(orc_g, pw, hseq[:int(hoff[2 * n2])], hoff[:2 * n2 + 1], with_recs=False)
    cw = dbtk.context(g, pw)
    cw.align(hseq[:int(hoff[2 * n2])], hoff[:2 * n2 + 1])
    r = cw.counts()
    co = np.zeros(g.ntrkmers, np.uint64)
    np.add.at(co, g.output_order().astype(np.int64), ow["counts_file"])
    assert (co == r["counts"]).all() and co.sum() > 0 and (ow["counters"] == r["counters"]).all()
    res, _, nres = cw.walk_results(n2)
    assert nres == ow["nres"] and bind.walk_res_equal(res, ow["res"], nres, g.nloci, every_mate=False) > 0
    cw.close()
    dense_slice_checks(monkeypatch, dbtk, oracle, syn, arrs, g, orc_g, 21)
    oracle.free(orc_g)
    g.close()


def test_release_scale_walk_k25(dbtk, oracle, monkeypatch):
    """BASELINE config 4 as stated: a release-scale synthetic RPGG at k = 25 (pipeline/k25.json:5), all-hit reads through the graph
    walk with -gc 85 3: counts, all counters and the walk results of a 30 000-pair slice against the oracle; the whole 1 M-pair
    batch through size-independent properties (additivity over a split, counted increments == sum of counts)."""
    syn = bind.pkg.Synth(nloci=80000, k=25)
    syn.graph()
    arrs = syn.arrays()
    h = C.c_void_p()
    dbtk._chk(dbtk.L.dbtk_rpgg_from_arrays(C.byref(arrs), C.byref(h)))
    g = bind.pkg.Rpgg(dbtk, h)
    pw = abi.default_params(ksize=25, n_filter=4, nm_filter=1, cthreshold=45, okam=0, threading=abi.THREADING_V13, thread_cth=85,
                            correction=1, maxncorrection=3)
    npairs = 1_000_000
    seq, off = syn.reads(npairs, hit_frac=1.0, seed=3)
    whole = dbtk.context(g, pw)
    whole.align(seq, off)
    w = whole.counts()
    ctr = w["counters"]
    assert ctr[abi.C_THREADING] > npairs and ctr[abi.C_FEASIBLE] > ctr[abi.C_THREADING] // 2
    assert int(w["counts"].sum()) == int(ctr[abi.C_ALGO_INC])
    parts = dbtk.context(g, pw)
    for a, b in ((0, 333_333), (333_333, npairs)):
        base = int(off[2 * a])
        parts.align(seq[base:int(off[2 * b])], off[2 * a:2 * b + 1] - np.uint64(base))
    q = parts.counts()
    for k_ in ("counts", "kmc", "nmapread", "counters"):
        assert (w[k_] == q[k_]).all(), k_
    parts.close()
    n2 = 30_000
    orc_g = oracle.from_arrays(arrs)
    ow = oracle.align_walk(orc_g, pw, seq[:int(off[2 * n2])], off[:2 * n2 + 1], with_recs=False)
    whole.reset()
    whole.align(seq[:int(off[2 * n2])], off[:2 * n2 + 1])
    r = whole.counts()
    co = np.zeros(g.ntrkmers, np.uint64)
    np.add.at(co, g.output_order().astype(np.int64), ow["counts_file"])
    assert (co == r["counts"]).all() and co.sum() > 0 and (ow["counters"] == r["counters"]).all()
    res, _, nres = whole.walk_results(n2)
    assert nres == ow["nres"] and bind.walk_res_equal(res, ow["res"], nres, g.nloci, every_mate=False) > 0
    whole.close()
    dense_slice_checks(monkeypatch, dbtk, oracle, syn, arrs, g, orc_g, 25)
    oracle.free(orc_g)
    g.close()


def test_legacy_v13_rpgg_end_to_end(dbtk, oracle):
    """The reference's own v1.3 fixture (tests/golden/legacy_v13) loaded by dbtk_rpgg_load, reads stitched from its k-mers:
    the HIP path against the oracle on the same flat arrays."""
    import test_abi
    g = dbtk.load(os.path.join(test_abi.LEGACY, "pan"), 21)
    a = test_abi._view_arrays(g)
    go = oracle.from_arrays(g.view())
    rng = np.random.default_rng(3)
    pool = np.concatenate([a["tr_ks"], a["fl_ks"]])

    def dec(km):
        return "".join("ACGT"[(int(km) >> (2 * (20 - i))) & 3] for i in range(21))
    reads = synth.Reads()
    for p in range(300):
        for _ in range(2):
            s = "".join(dec(pool[rng.integers(len(pool))]) for _ in range(8))[:150]
            if rng.random() < 0.3:
                s = s[:70] + "".join(rng.choice(list("ACGT"), 80))
            reads.seqs.append(s.encode())
        reads.titles.append(f"r{p}")
    seq, off = reads.packed()
    for kw in (dict(cthreshold=2, trace=1), dict(cthreshold=2, okam=0), dict(cthreshold=5)):
        p = abi.default_params(ksize=21, **kw)
        o = oracle.align(go, p, seq, off, trace=bool(kw.get("trace")))
        ctx = dbtk.context(g, p)
        recs, nrec = ctx.align(seq, off)
        res = ctx.counts()
        res["recs"] = recs
        compare(o, res, g.output_order(), g.ntrkmers, reads.npairs, recs=bool(kw.get("trace")))
        ctx.close()
    oracle.free(go)
    g.close()


def test_leaked_context_cannot_lend_its_tables(dbtk, oracle, tmp_path):
    """A context that is never freed keeps its share of the device tables alive.  If its RPGG handle is freed all the same and the
    next handle lands on the same address, the new context must build its own tables (they are keyed by the handle's id, not its
    address): results are those of the NEW RPGG.  Also: DBTK_MZ=0 (no minimizer-grouped copy: the general probe kernel on a
    geometry the lean one usually takes) is read when a handle's tables are first built."""
    p = abi.default_params(ksize=21, cthreshold=45, trace=1)
    c1 = make_case("clean", str(tmp_path))
    g1 = dbtk.load(c1.prefix, c1.k, c1.qc_file)
    leaked = dbtk.context(g1, p)  # (never closed)
    addr = g1.h.value
    g1.close()
    c2 = make_case("shared", str(tmp_path))
    go = oracle.load(c2.prefix, c2.k, c2.qc_file)
    seq, off = c2.reads.packed()
    o = oracle.align(go, p, seq, off)
    same_address = False
    for attempt in range(6):
        if attempt == 3:
            os.environ["DBTK_MZ"] = "0"
        g2 = dbtk.load(c2.prefix, c2.k, c2.qc_file)
        same_address |= g2.h.value == addr
        ctx = dbtk.context(g2, p)
        recs, nrec = ctx.align(seq, off)
        res = ctx.counts()
        res["recs"] = recs
        compare(o, res, g2.output_order(), g2.ntrkmers, c2.reads.npairs, recs=True)
        ctx.close()
        g2.close()
    os.environ.pop("DBTK_MZ", None)
    oracle.free(go)
    del leaked


def test_survivor_chunks_on_device(dbtk, oracle, tmp_path, monkeypatch):
    """The K2 -> K3 hit buffers hold DBTK_SURV_CAP survivors; with a tiny cap one batch runs as many chunk iterations
    (the path batches of more than 8 M pairs take): same results, records included."""
    monkeypatch.setenv("DBTK_SURV_CAP", "37")
    c = make_case("shared", str(tmp_path))
    go = oracle.load(c.prefix, c.k, c.qc_file)
    g = dbtk.load(c.prefix, c.k, c.qc_file)
    seq, off = c.reads.packed()
    for kw in (dict(trace=1), dict(okam=0), dict()):
        p = abi.default_params(ksize=c.k, **dict(c.param_sets[0], **kw))
        o = oracle.align(go, p, seq, off, trace=True)
        ctx = dbtk.context(g, p)
        recs, nrec = ctx.align(seq, off)
        res = ctx.counts()
        if kw.get("trace"):
            res["recs"] = recs
            compare(o, res, g.output_order(), g.ntrkmers, c.reads.npairs)
        else:
            oo = oracle.align(go, p, seq, off, trace=False)
            compare(oo, res, g.output_order(), g.ntrkmers, 0, recs=False)
        ctx.close()
    oracle.free(go)
    g.close()


def test_randomised_parity_soak():
    """tests/fuzz_parity.py: random RPGGs (k 17/21/25, shared flanks), read sets (64-250 bp, substitutions, indels, N, chimeras,
    background) and parameters; trace records and counts against the oracle.  (600 seeds were run when this was added.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_parity.py"), "16", "900"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-3000:]


def test_randomised_walk_parity_soak():
    """tests/fuzz_parity.py walk: random RPGGs with their graphs (text / .umap loaders alternating), read sets with errors of every
    kind, random thread_cth / correction / maxncorrection / -a / -ae; counts, totals, walk results and alignment records against
    the oracle.  (40 further seeds, 5000-5039, were run when this was added.)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_parity.py"), "walk", "10", "6000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-3000:]


def test_device_entry_reports_a_read_longer_than_promised(dbtk, tmp_path):
    """dbtk_align_batch_device trusts max_read_len (it sizes the rows of the hit buffers).  A longer read must not write past its
    row: the probe kernel clamps it and raises the sticky error word, which dbtk_ctx_synchronize reports once — also when
    further batches ran in between."""
    c = make_case("mixed", str(tmp_path))
    g = dbtk.load(c.prefix, c.k, c.qc_file)
    seq, off = c.reads.packed()
    p = abi.default_params(ksize=c.k, **dict(c.param_sets[0], okam=0))
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    d_seq, d_off = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d_seq), len(seq) + 64) == 0 and hip.hipMalloc(C.byref(d_off), off.nbytes) == 0
    assert hip.hipMemcpy(d_seq, seq.ctypes.data_as(C.c_void_p), len(seq), 1) == 0
    assert hip.hipMemcpy(d_off, off.ctypes.data_as(C.c_void_p), off.nbytes, 1) == 0
    maxlen = int(np.diff(off.astype(np.int64)).max())
    ctx = dbtk.context(g, p)
    ctx.align_device(d_seq.value, d_off.value, c.reads.npairs, 100)      # the reads are 150 bases
    for _ in range(3):
        ctx.align_device(d_seq.value, d_off.value, c.reads.npairs, maxlen)
    with pytest.raises(bind.pkg.DbtkError) as e:
        ctx.synchronize()
    assert e.value.status == abi.ERR_READ_TOO_LONG
    ctx.synchronize()  # reported once
    ctx.close()
    hip.hipFree(d_seq); hip.hipFree(d_off)
    g.close()


def test_contexts_share_device_tables(dbtk, oracle, tmp_path):
    """Contexts created for the same (RPGG handle, device) share one set of HBM tables (built by the first, freed with the
    last): a second and third context cost only their accumulators and scratch, and still give the oracle's result — also
    after the first context has been closed."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

    def free_bytes():
        f, t = C.c_size_t(0), C.c_size_t(0)
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value
    syn = bind.pkg.Synth(nloci=4000)
    arrs = syn.arrays()
    h = C.c_void_p()
    dbtk._chk(dbtk.L.dbtk_rpgg_from_arrays(C.byref(arrs), C.byref(h)))
    g = bind.pkg.Rpgg(dbtk, h)
    p = abi.default_params(ksize=21, cthreshold=45, okam=0)
    seq, off = syn.reads(20000, hit_frac=0.5, seed=4)
    # Twice through everything first, with as many contexts alive as below: what the HIP runtime allocates at a kernel's launch — code
    # objects at the first, and scratch for the kernels that spill (75 MB more for the fused probe kernel, taken at the SECOND cycle of
    # contexts and kept from then on: probing free memory over four such cycles shows cycles 2, 3, 4 ... identical to the byte) — is not the
    # library's to give back, and must not count below.  A leak of the library's own would show in every cycle, so also in the measured one.
    for _ in range(2):
        warm = [dbtk.context(g, p) for _ in range(3)]
        for c0 in warm:
            c0.align(seq, off)
        for c0 in warm:
            c0.close()
    m0 = free_bytes()
    c1 = dbtk.context(g, p)
    m1 = free_bytes()
    c2 = dbtk.context(g, p)
    c3 = dbtk.context(g, abi.default_params(ksize=21, cthreshold=30, okam=0))
    m3 = free_bytes()
    first, extra = m0 - m1, (m1 - m3) / 2
    assert first > 250e6 and extra < 0.25 * first, (first, extra)   # tables ~0.5 GB here; a further context: accumulators + vote scratch
    go = oracle.from_arrays(arrs)
    o = oracle.align(go, p, seq, off, trace=False)
    c1.close()                                                        # the tables outlive their builder
    c2.align(seq, off)
    compare(o, c2.counts(), g.output_order(), g.ntrkmers, 0, recs=False)
    c2.close(); c3.close()
    assert free_bytes() >= m0 - 64e6                                 # and go with the last context
    oracle.free(go)
    g.close()


def test_two_gpu_allreduce_equals_single_gpu(dbtk, oracle):
    """The one exchange of the path on real hardware: two contexts on two GPUs take halves of the pairs, dbtk_allreduce (RCCL
    over xGMI) sums their accumulators, and both then hold the single-GPU / oracle result.  Skipped on a 1-GPU box."""
    hip = C.CDLL("libamdhip64.so")
    n = C.c_int(0)
    if hip.hipGetDeviceCount(C.byref(n)) != 0 or n.value < 2:
        pytest.skip("needs two GPUs")
    syn = bind.pkg.Synth(nloci=3000)
    arrs = syn.arrays()
    h = C.c_void_p()
    dbtk._chk(dbtk.L.dbtk_rpgg_from_arrays(C.byref(arrs), C.byref(h)))
    g = bind.pkg.Rpgg(dbtk, h)
    p = abi.default_params(ksize=21, cthreshold=45, okam=0)
    npairs = 60000
    seq, off = syn.reads(npairs, hit_frac=0.5, seed=6)
    go = oracle.from_arrays(arrs)
    o = oracle.align(go, p, seq, off, trace=False)
    ctxs = [dbtk.context(g, p, device=d) for d in range(2)]
    half = npairs // 2
    ctxs[0].align(seq[:int(off[2 * half])], off[:2 * half + 1])
    ctxs[1].align(seq[int(off[2 * half]):], off[2 * half:] - off[2 * half])
    dbtk.allreduce(ctxs)
    for c in ctxs:
        compare(o, c.counts(), g.output_order(), g.ntrkmers, 0, recs=False)
        c.close()
    oracle.free(go)
    g.close()


@pytest.mark.gpu
def test_bench_collective_path_on_one_gpu():
    """bench.py exactly as the driver starts it for N > 1 — under torch.distributed.run, one rank per GPU, RCCL process group,
    barriers, the MAX over ranks and the all-reduce of the accumulators — in a world of one (DBTK_BENCH_FORCE_DIST=1), so the
    code path of the 8-GPU run executes on every 1-GPU box: one rank seen, counts bit-exact against the oracle."""
    import json
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
           str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--nloci", "3000", "--reads", "400000",
           "--mix-reads", "0", "--no-e2e", "--ref-reads", "0", "--cpu-seconds", "4", "--parity-pairs", "30000"]
    import tempfile
    detail = os.path.join(tempfile.mkdtemp(prefix="dbtk_bench_"), "detail.json")
    env = dict(os.environ, DBTK_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), DBTK_BENCH_DETAIL=detail)
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert len(line) < 6000 and r.stdout.rstrip().endswith(line)  # the driver parses the LAST line out of an 8 000-character tail of stdout
    d = json.loads(line)
    full = json.load(open(detail))  # per-kernel tables and everything else: the side file
    assert full["roofline"]["kernels"] and full["value"] == pytest.approx(d["value"], rel=1e-3)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in d["cpu_baseline"], key
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["parity"] and d["parity"]["bit_exact"] and d["parity"]["pairs"] >= 30000
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """The N > 1 code of bench.py with two REAL ranks on a one-GPU box (DBTK_BENCH_ALL_ON_DEVICE0=1: both ranks on device 0, gloo
    with host staging in place of RCCL, which refuses two ranks on one device): each rank its own shard of the read set, barriers,
    MAX over ranks, the all-reduce of the accumulators — and the reduced counts equal the oracle's over BOTH shards x steps."""
    import json
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--nloci", "2000", "--reads", "120000",
           "--hit-frac", "0.5", "--mix-reads", "0", "--no-e2e", "--ref-reads", "0", "--cpu-seconds", "0"]
    import tempfile
    env = dict(os.environ, DBTK_BENCH_ALL_ON_DEVICE0="1", DBTK_BENCH_REDUCE_CHECK="1", DBTK_BENCH_DETAIL=os.path.join(tempfile.mkdtemp(prefix="dbtk_bench_"), "detail.json"))
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert len(d["per_rank_ms_per_step"]) == 2 and max(d["per_rank_ms_per_step"]) == pytest.approx(d["ms_per_step"], rel=1e-3)  # (every rank's own time: a first multi-GPU run is diagnosable from the line)
    assert d["reduce_check"] == dict(ranks=2, steps=2, pairs_per_rank=60000, bit_exact=True)


def test_checked_launches_mode(tmp_path):
    """DBTK_SYNC_LAUNCHES=1 (what the debug build, `make -C danbing-tk_amd/csrc debug`, hard-wires): every launch is waited
    for and checked by name.  Same results; run in a child process because the switch is read once per process."""
    code = """
import sys, ctypes as C
sys.path.insert(0, %r)
import bind
import numpy as np
abi = bind.abi
dbtk = bind.pkg.Dbtk()
orc = bind.Oracle()
syn = bind.pkg.Synth(nloci=300)
syn.graph()
arrs = syn.arrays()
h = C.c_void_p()
dbtk._chk(dbtk.L.dbtk_rpgg_from_arrays(C.byref(arrs), C.byref(h)))
g = bind.pkg.Rpgg(dbtk, h)
seq, off = syn.reads(3000, hit_frac=0.6, seed=9)
for p in (abi.default_params(ksize=21, cthreshold=45, okam=0),
          abi.default_params(ksize=21, cthreshold=45, okam=0, threading=abi.THREADING_V13, thread_cth=85, correction=1, maxncorrection=3)):
    ctx = dbtk.context(g, p)
    ctx.align(seq, off)
    r = ctx.counts()
    go = orc.from_arrays(arrs)
    if p.threading == abi.THREADING_V13:
        o = orc.align_walk(go, p, seq, off)
    else:
        o = orc.align(go, p, seq, off, trace=False)
    co = np.zeros(g.ntrkmers, np.uint64)
    np.add.at(co, g.output_order().astype(np.int64), o["counts_file"])
    assert (co == r["counts"]).all(), "counts differ"
    assert (o["kmc"] == r["kmc"]).all() and (o["nmapread"] == r["nmapread"]).all()
    assert (o["counters"] == r["counters"]).all(), (o["counters"], r["counters"])
    ctx.close()
print("checked-launches ok")
""" % os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, DBTK_SYNC_LAUNCHES="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "checked-launches ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"DBTK_K1_LAZY": "1"}, {"DBTK_K1_LAZY": "1", "DBTK_LOCUS_ALWAYS": "1"}, {"DBTK_K1_XCD": "0", "DBTK_NO_K1_KEYS": "1"}])
def test_encode_kernel_forms_forced(env):
    """The encode kernel's forms the launcher otherwise picks by a hint — k_encode_subfilter_lazy (a batch that hits), the sort keys handed over
    or looked up again, the tiles per XCD or by plain stride — forced by their environment switches (read once per process: a child process),
    each on a few random cases of tests/fuzz_parity.py against the oracle."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "fuzz_parity.py"), "8", "424200"], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
    assert r.returncode == 0 and "8/8 seeds bit-exact" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
