"""Graph-threading walk: isThreadFeasible and everything under it.

  oracle (oracle/dbtk_oracle_walk.c)  ==  the reference itself (oracle/_ref/libdbtk_refharness.so: ref_thread)
  emulated kernel body (dbtk_walk.h)   ==  oracle                                  [CPU suite]
  k_walk_reads on the GPU (dbtk_thread_batch through the C-ABI) == oracle          [-m gpu]

bit-exact on every output field: return code, cg.ni, the corrected k-mers, every edit operation (type, read base,
graph base), every k-mer annotation, and the strings writeCigar / writeAnnot print.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from collections import Counter

import sys

import numpy as np
import pytest

_HERE = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.dirname(_HERE), _HERE):  # (also run as a script: the per-k worker of the oracle-vs-reference test)
    if _p not in sys.path:
        sys.path.insert(0, _p)

import bind  # noqa: E402
import synth  # noqa: E402

abi = bind.abi


# ------------------------------------------------------------------ read sets --
def _mutate_at(s: bytes, pos: int, kind: str, rng) -> bytes:
    b = bytearray(s)
    alt = lambda c: bytes([x for x in b"ACGT" if x != c])[int(rng.integers(0, 3))]
    if kind == "X":
        b[pos] = alt(b[pos])
    elif kind == "XX":
        b[pos] = alt(b[pos]); b[pos + 1] = alt(b[pos + 1])
    elif kind == "I":
        b.insert(pos, b"ACGT"[int(rng.integers(0, 4))]); b = b[:len(s)]
    elif kind == "II":
        b.insert(pos, b"ACGT"[int(rng.integers(0, 4))]); b.insert(pos, b"ACGT"[int(rng.integers(0, 4))]); b = b[:len(s)]
    elif kind == "D":
        del b[pos]; b.append(b"ACGT"[int(rng.integers(0, 4))])
    elif kind == "DD":
        del b[pos:pos + 2]; b += bytes(b"ACGT"[int(x)] for x in rng.integers(0, 4, 2))
    elif kind == "XI":
        b[pos] = alt(b[pos]); b.insert(pos + 1, b"ACGT"[int(rng.integers(0, 4))]); b = b[:len(s)]
    elif kind == "XD":
        b[pos] = alt(b[pos]); del b[pos + 1]; b.append(b"ACGT"[int(rng.integers(0, 4))])
    elif kind == "N":
        b[pos] = ord("N")
    return bytes(b)


def make_reads(loci, k, n, seed, sub=0.0, indel=0.0, nrate=0.0, targeted=0, wrong_locus=0.0, rlen=150):
    """(reads, locus per read).  `targeted` reads carry exactly one edit of each class at a chosen place:
    near the start (leading gap -> backward correction), in the middle, near the end (tail skip)."""
    rng = np.random.default_rng(seed)
    sim = synth.sim_reads(loci, npairs=(n + 1) // 2, seed=seed, sub=sub, indel=indel, nrate=nrate, rlen=rlen)
    seqs, loc = [], []
    for i, s in enumerate(sim.seqs[:n]):
        l = int(re.match(r"r\d+:l(\d+)", sim.titles[i // 2]).group(1))
        if len(s) < k:
            continue
        if rng.random() < wrong_locus:
            l = int(rng.integers(0, loci.nloci))
        seqs.append(s); loc.append(l)
    kinds = ["X", "XX", "I", "II", "D", "DD", "XI", "XD", "N"]
    clean = synth.sim_reads(loci, npairs=(targeted + 1) // 2, seed=seed + 1000, rlen=rlen)
    for i, s in enumerate(clean.seqs[:targeted]):
        l = int(re.match(r"r\d+:l(\d+)", clean.titles[i // 2]).group(1))
        kind = kinds[i % len(kinds)]
        where = (i // len(kinds)) % 4
        pos = [int(rng.integers(0, 12)), int(rng.integers(12, 40)), int(rng.integers(40, rlen - 40)), int(rng.integers(rlen - 30, rlen - 3))][where]
        s2 = _mutate_at(s, pos, kind, rng)
        if (i // (4 * len(kinds))) % 2:  # a second edit elsewhere
            s2 = _mutate_at(s2, int(rng.integers(5, rlen - 5)), kinds[int(rng.integers(0, len(kinds)))], rng)
        seqs.append(s2); loc.append(l)
    return seqs, loc


def pack(seqs):
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    return np.frombuffer(b"".join(seqs), np.uint8).copy(), off


class WalkCase:
    def __init__(self, tmp, name, k, seed, nloci=8, motif=(3, 30), with_ref=True):
        self.k = k
        self.loci = synth.make_loci(nloci=nloci, nhap=3, seed=seed, motif_min=motif[0], motif_max=motif[1])
        d = os.path.join(tmp, name)
        os.makedirs(d, exist_ok=True)
        if with_ref and synth.have_ref():
            self.prefix = synth.build_rpgg_with_reference(self.loci, d, k=k)  # fa2kmers -g writes pan.graph.kmers too
        else:
            self.prefix = os.path.join(d, "pan")
            synth.write_rpgg_files(synth.build_rpgg_arrays(self.loci, k), self.prefix)
            synth.write_graph_file(synth.build_graph_arrays(self.loci, k), self.prefix)


PARAM_SETS = [
    dict(thread_cth=85, correction=1, maxncorrection=3),   # BASELINE config 4: -gc 85 3
    dict(thread_cth=100, correction=1, maxncorrection=4),  # the reference's defaults with -gc
    dict(thread_cth=60, correction=0, maxncorrection=4),   # -g: no correction
    dict(thread_cth=120, correction=1, maxncorrection=1),  # budgets that run out
]


def edit_classes(cigar: str) -> Counter:
    c = Counter()
    for m in re.finditer(r"(?:X[ACGT]|D[ACGT*]|I)+", cigar):
        ops = re.findall(r"X[ACGT]|D[ACGT*]|I", m.group(0))
        c["".join(o[0] for o in ops)] += 1
    return c


def same_rec(a, b):
    return bytes(a) == bytes(b)


def describe(name, i, ro, orec, r2, rec2, O):
    return (f"{name} read {i}: oracle ret={ro} other ret={r2} nkm {orec.nkm}/{rec2.nkm} nes {orec.nes}/{rec2.nes} "
            f"ni {orec.ni}/{rec2.ni} flags {orec.flags}/{rec2.flags}\n  {O.cigar_annot(orec)}\n  {O.cigar_annot(rec2)}")


# ------------------------------------------------------------ oracle vs ref --
def oracle_vs_reference(tmp, k, seed):
    """>= 100 k reads per k in {21, 25}: error rates 0 / 0.5 % / 2 % / 5 %, N, targeted edits of all 8 classes at the read's
    start / middle / end, reads walked through the wrong locus, four parameter sets; threadCheck's [!] flags are counted."""
    O = bind.Oracle()
    H = bind.RefHarness()
    case = WalkCase(tmp, f"w{k}", k, seed)
    oh = O.load(case.prefix, k)
    O.load_graph(oh, case.prefix + ".graph.kmers")
    nper = 13000 if k != 17 else 2500
    classes, rets, flagged, total = Counter(), Counter(), 0, 0
    for pi, ps in enumerate(PARAM_SETS):
        p = abi.default_params(ksize=k, **ps)
        H.set_params(p)
        if pi == 0:
            h = H.load(case.prefix)
            H.load_graph(h, case.prefix + ".graph.kmers")
        for si, (sub, indel, nrate) in enumerate([(0.0, 0.0, 0.0), (0.005, 0.001, 0.0), (0.02, 0.005, 0.002), (0.05, 0.02, 0.0)]):
            seqs, loc = make_reads(case.loci, k, nper if pi == 0 else nper // 3, seed=100 * pi + si + k, sub=sub, indel=indel, nrate=nrate,
                                   targeted=720 if si == 0 else 0, wrong_locus=0.03)
            for i, s in enumerate(seqs):
                rr, rrec, cig, ann, fl = H.thread(h, loc[i], s, p, tc=True)
                ro, orec = O.thread(oh, loc[i], s, p)
                assert rr >= 0, "the reference asserted on a clean graph"
                assert ro == rr and same_rec(orec, rrec), describe("oracle vs reference", i, ro, orec, rr, rrec, O)
                assert O.cigar_annot(orec) == (cig, ann)
                classes.update(edit_classes(cig))
                rets[rr] += 1
                flagged += fl
                total += 1
    if k != 17:
        assert total >= 100_000
    for cls in ("X", "D", "I", "XX", "XD", "XI", "DD", "II"):  # all 8 edit classes occurred
        assert classes[cls] > 0, (cls, classes)
    assert rets[0] > 0 and rets[1] > 0 and rets[2] > 0
    print(f"k={k}: {total} reads, ret {dict(rets)}, edit tracts {dict(classes)}, {flagged} flagged by threadCheck")


# ---- the hot loop with threading = 2 (v1.3 call sites): pair mode ---------------------------------------------
PAIR_IDX = [abi.C_NREADS, abi.C_SUBFILTERED, abi.C_KMERFILTERED, abi.C_LOCUSFILTERED, abi.C_QCFILTERED, abi.C_THREADING,
            abi.C_FEASIBLE, abi.C_NSHORT, abi.C_NHASH0, abi.C_NHASH1, abi.C_ALGO_INC]


def pair_reads(case, seed, npairs=1500, rlen=150):
    return synth.sim_reads(case.loci, npairs=npairs, sub=0.02, indel=0.004, seed=seed, nrate=0.002, chimeric=0.2, background=0.2, short=0.03, rlen=rlen)


def expected_aln(O, o, reads, aln, nloci):
    """(pair, dst, "cigar2 annot2 cigar1 annot1") per emitted pair + the text writeAlignments prints, from the oracle's results."""
    recs, lines = [], []
    for i in range(o["nres"]):
        w = o["res"][i]
        if aln == 2 and w.dst == nloci:
            continue
        c1, a1 = O.cigar_annot(o["trecs"][2 * i])
        c2, a2 = O.cigar_annot(o["trecs"][2 * i + 1])
        recs.append((w.pair, w.dst, f"{c2}\t{a2}\t{c1}\t{a1}"))
        lines.append(bind.aln_line(O, o["res"], o["trecs"], i, reads.titles[w.pair], reads.seqs[2 * w.pair], reads.seqs[2 * w.pair + 1], nloci))
    return recs, "".join(lines)


def glue_oracle_vs_reference(tmp, k, seed):
    """orc_align_walk against the reference's functions driven through the commented-out v1.3 call sites
    (oracle/ref_harness.cpp: ref_align_v13): walk results, thread records, exact counts, counters, and the -a / -ae text."""
    O = bind.Oracle()
    H = bind.RefHarness()
    case = WalkCase(tmp, f"p{k}", k, seed)
    oh = O.load(case.prefix, k); O.load_graph(oh, case.prefix + ".graph.kmers")
    h = H.load(case.prefix); H.load_graph(h, case.prefix + ".graph.kmers")
    for aln, ps in ((1, PARAM_SETS[0]), (2, PARAM_SETS[1]), (2, PARAM_SETS[3])):
        p = abi.default_params(ksize=k, cthreshold=45, threading=2, aln=aln, okam=0, **ps)
        reads = pair_reads(case, seed=5 + aln + ps["thread_cth"])
        seq, off = reads.packed()
        r = H.align_v13(h, p, seq, off, reads.titles)
        o = O.align_walk(oh, p, seq, off)
        n = o["nres"]
        assert n == r["nres"] and n > 500
        assert bytes(r["res"])[:8 * n] == bytes(o["res"])[:8 * n]
        sz = C.sizeof(abi.ThreadRec)
        assert bytes(r["trecs"])[:2 * n * sz] == bytes(o["trecs"])[:2 * n * sz]
        assert (r["counts_file"] == o["counts_file"]).all() and o["counts_file"].sum() > 0
        assert (r["counters"][PAIR_IDX] == o["counters"][PAIR_IDX]).all()
        _, text = expected_aln(O, o, reads, aln, case.loci.nloci)
        assert text == r["aln"]
    print(f"glue k={k}: ok")


@pytest.mark.skipif(not synth.have_ref(), reason="needs oracle/_ref (make -C oracle ref where /root/reference exists)")
@pytest.mark.parametrize("k,seed", [(21, 3), (25, 4), (17, 5)])
def test_oracle_walk_equals_reference(tmp_path, k, seed):
    """One process per k: the reference keeps `static` constants derived from the global ksize inside
    edit_kmers_backward (AQ.cpp:653-654), so a loaded copy of it can only walk with the k of its first correction."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__), str(tmp_path), str(k), str(seed)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = [l for l in r.stdout.strip().splitlines() if l.startswith(("k=", "glue"))]
    assert len(out) == 2 and out[1].endswith("ok")
    print("\n".join(out))


LEGACY = os.path.join(_HERE, "golden", "legacy_v13")


def _product_graph(g):
    """graphDB of a product handle from its flat view: per locus (nodes ascending, masks OR-ed per node)."""
    v = g.view()
    nl = int(v.nloci)
    cnt = np.ctypeslib.as_array(v.gr_cnt, shape=(nl,)).astype(np.int64)
    n = int(cnt.sum())
    ks = np.ctypeslib.as_array(v.gr_ks, shape=(n,)).copy()
    ms = np.ctypeslib.as_array(v.gr_ms, shape=(n,)).copy()
    out, i = [], 0
    for c in cnt:
        d = {}
        for k_, m_ in zip(ks[i:i + c].tolist(), ms[i:i + c].tolist()):
            d[k_] = d.get(k_, 0) | m_
        i += int(c)
        out.append((np.array(sorted(d), np.uint64), np.array([d[k_] for k_ in sorted(d)], np.uint8)))
    return out


def _legacy_prefix_with(tmp, graph_ext):
    """The reference's v1.3 fixture with only ONE of its two graph files next to it (the loaders prefer the text form)."""
    import shutil
    d = os.path.join(tmp, "leg_" + graph_ext.replace(".", "_"))
    os.makedirs(d, exist_ok=True)
    for f in ("pan.kmerDBi.umap", "pan.kmerDBi.vv", "pan.tr.kmers", "pan.ntr.kmers", "pan.graph." + graph_ext):
        shutil.copy(os.path.join(LEGACY, f), os.path.join(d, f))
    return os.path.join(d, "pan")


def test_reference_graph_fixture_loads_the_same_everywhere(tmp_path):
    """The reference's own v1.3 graph fixture (test/QC/input/pan.graph.umap and its text twin pan.graph.kmers, byte copies under
    tests/golden/legacy_v13): the binary loader == the text loader == the reference's readGraphKmers
    (src/aQueryFasta_thread.h:550-575, through oracle/_ref), node for node and mask for mask, in the oracle AND through
    dbtk_rpgg_load(DBTK_LOAD_GRAPH)."""
    umap, text = os.path.join(LEGACY, "pan.graph.umap"), os.path.join(LEGACY, "pan.graph.kmers")
    assert os.path.getsize(umap) == 26998  # 8 (nloci) + 8 (n) + 2998 x 9
    raw = open(umap, "rb").read()
    assert int.from_bytes(raw[:8], "little") == 1 and int.from_bytes(raw[8:16], "little") == 2998
    lib = bind.pkg.Dbtk()
    O = bind.Oracle()
    graphs = {}
    for ext in ("umap", "kmers"):
        pref = _legacy_prefix_with(str(tmp_path), ext)
        g = lib.load(pref, 21, flags=abi.LOAD_GRAPH)  # product loader (csrc/dbtk_rpgg.cpp: read_graph_umap / read_graph_text)
        graphs["product " + ext] = _product_graph(g)[0]
        oh = O.from_arrays(g.view())                  # oracle loader (oracle/dbtk_oracle.c: orc_rpgg_load_graph) on the same index
        O.load_graph(oh, pref + ".graph." + ext)
        graphs["oracle " + ext] = O.graph_dump(oh, 0)
        O.free(oh)
        g.close()
    if synth.have_ref():  # the reference's own loader, on HEAD files of the same k-mer sets (ktools serialize) + the text graph
        import shutil
        import subprocess
        d = os.path.join(str(tmp_path), "head")
        os.makedirs(d)
        shutil.copy(os.path.join(LEGACY, "pan.tr.kmers"), os.path.join(d, "pan.tr.kmers"))
        shutil.copy(os.path.join(LEGACY, "pan.ntr.kmers"), os.path.join(d, "pan.fl.kmers"))
        open(os.path.join(d, "pan.tre.kmers"), "w").write(">0\n")
        assert subprocess.run([synth.ref_tool("ktools"), "serialize", "pan"], cwd=d, capture_output=True).returncode == 0
        R = bind.RefHarness()
        rh = R.load(os.path.join(d, "pan"))
        R.load_graph(rh, text)
        graphs["reference readGraphKmers"] = R.graph_dump(rh, 0)
        R.free(rh)
    first = graphs["oracle umap"]
    assert len(first[0]) == 2998 and (np.diff(first[0].astype(np.int64)) > 0).all()
    for name, (ks, ms) in graphs.items():
        assert len(ks) == len(first[0]) and (ks == first[0]).all() and (ms == first[1]).all(), name
    # and the file says the same: node i at bytes 16 + 9 i, its mask in the ninth byte
    nodes = np.frombuffer(raw[16:], np.uint8).reshape(2998, 9)
    fk = nodes[:, :8].copy().view(np.uint64).ravel()
    order = np.argsort(fk)
    assert (fk[order] == first[0]).all() and (nodes[:, 8][order] == first[1]).all()


def _reads_from_graph(ks, ms, k, n, rlen, seed, sub=0.0):
    """Reads spelled by random walks through a graph (node = non-canonical k-mer, bit b of its mask = successor by base b);
    a walk that reaches a node without successors starts over."""
    rng = np.random.default_rng(seed)
    mask = dict(zip(ks.tolist(), ms.tolist()))
    nodes = [x for x in ks.tolist() if mask[x]]
    kmask = (1 << (2 * k)) - 1
    reads = []
    while len(reads) < n:
        node = nodes[int(rng.integers(len(nodes)))]
        s = ["ACGT"[(node >> (2 * (k - 1 - i))) & 3] for i in range(k)]
        while len(s) < rlen:
            m = mask.get(node, 0)
            if not m:
                break
            b = int(rng.choice([i for i in range(4) if m >> i & 1]))
            node = ((node << 2) | b) & kmask
            s.append("ACGT"[b])
        if len(s) < rlen:
            continue
        r = bytearray("".join(s).encode())
        for i in np.nonzero(rng.random(rlen) < sub)[0]:
            r[i] = ord("ACGT"[(b"ACGT".index(r[i]) + 1 + int(rng.integers(3))) % 4])
        reads.append(bytes(r))
    return reads


def test_oracle_walks_the_reference_graph_fixture_like_the_reference(tmp_path):
    """Reads stitched from the reference's own graph fixture, walked by the oracle and by the reference's isThreadFeasible on the
    graph each loaded for itself."""
    if not synth.have_ref():
        pytest.skip("needs oracle/_ref")
    import shutil
    import subprocess
    d = os.path.join(str(tmp_path), "head")
    os.makedirs(d)
    shutil.copy(os.path.join(LEGACY, "pan.tr.kmers"), os.path.join(d, "pan.tr.kmers"))
    shutil.copy(os.path.join(LEGACY, "pan.ntr.kmers"), os.path.join(d, "pan.fl.kmers"))
    open(os.path.join(d, "pan.tre.kmers"), "w").write(">0\n")
    assert subprocess.run([synth.ref_tool("ktools"), "serialize", "pan"], cwd=d, capture_output=True).returncode == 0
    O, R = bind.Oracle(), bind.RefHarness()
    oh = O.load(os.path.join(d, "pan"), 21); O.load_graph(oh, os.path.join(LEGACY, "pan.graph.umap"))
    rh = R.load(os.path.join(d, "pan")); R.load_graph(rh, os.path.join(LEGACY, "pan.graph.kmers"))
    ks, ms = O.graph_dump(oh, 0)
    p = abi.default_params(ksize=21, thread_cth=85, correction=1, maxncorrection=3)
    R.set_params(p)
    rets = Counter()
    for sub in (0.0, 0.01, 0.03):
        for s in _reads_from_graph(ks, ms, 21, 150, 150, seed=int(100 * sub) + 7, sub=sub):
            ro, orec = O.thread(oh, 0, s, p)
            rr, rrec, _, _, _ = R.thread(rh, 0, s, p)
            assert ro == rr and same_rec(orec, rrec)
            rets[ro] += 1
    assert rets[1] > 0 and rets[2] > 0
    O.free(oh); R.free(rh)


# --------------------------------------------------- emulated kernel vs oracle --
@pytest.mark.parametrize("k,seed", [(21, 3), (25, 4)])
def test_emulated_walk_kernel_equals_oracle(tmp_path, k, seed):
    O = bind.Oracle()
    E = bind.Emu()
    case = WalkCase(str(tmp_path), f"e{k}", k, seed)
    oh = O.load(case.prefix, k)
    O.load_graph(oh, case.prefix + ".graph.kmers")
    g = E.load(case.prefix, k, flags=abi.LOAD_GRAPH)
    tb = E.tables(g)
    n = 0
    for pi, ps in enumerate(PARAM_SETS):
        p = abi.default_params(ksize=k, threading=2, **ps)
        for si, (sub, indel, nrate) in enumerate([(0.0, 0.0, 0.0), (0.01, 0.003, 0.002), (0.04, 0.02, 0.0)]):
            seqs, loc = make_reads(case.loci, k, 260 if pi else 700, seed=100 * pi + si + k, sub=sub, indel=indel, nrate=nrate,
                                   targeted=216 if (si == 0 and pi < 2) else 0, wrong_locus=0.03)
            buf, off = pack(seqs)
            recs = E.thread(g, tb, p, buf, off, np.array(loc, np.uint32), grid=3)
            for i, s in enumerate(seqs):
                ro, orec = O.thread(oh, loc[i], s, p)
                assert recs[i].ret == ro and same_rec(recs[i], orec), describe("emulated kernel vs oracle", i, ro, orec, recs[i].ret, recs[i], O)
                n += 1
    assert n > 3000


def check_pair_mode(run, O, oh, case, k, nloci, rlen=150, sets=((1, 0), (2, 1), (0, 2), (2, 3)), npairs=1500, lean=True):
    """run(p, seq, off) -> dict(counts (OUT.trkmc.ar order), counters, res, nres, aln, order): compared with the oracle."""
    for aln, psi in sets:
        ps = PARAM_SETS[psi]
        p = abi.default_params(ksize=k, cthreshold=45, threading=2, aln=aln, okam=0, **ps)
        reads = pair_reads(case, seed=5 + aln + ps["thread_cth"], npairs=npairs, rlen=rlen)
        seq, off = reads.packed()
        o = O.align_walk(oh, p, seq, off)
        g = run(p, seq, off)
        co = np.zeros(len(g["counts"]), np.uint64)
        np.add.at(co, g["order"].astype(np.int64), o["counts_file"])
        assert (co == g["counts"]).all() and co.sum() > 0
        assert (o["counters"] == g["counters"]).all(), (o["counters"], g["counters"])
        n = o["nres"]
        assert g["nres"] == n
        skipped = bind.walk_res_equal(g["res"], o["res"], n, nloci, every_mate=bool(aln))
        assert skipped >= 0 and (aln or skipped > 0 or not lean)  # (without -a / -ae the lean first kernel decides most pairs; the other one stops at a pair's first threading mate)
        exp, _ = expected_aln(O, o, reads, aln, nloci)
        if aln:
            assert [(h.pair, h.dst, t) for h, t in g["aln"]] == exp
            # the same records in text form (params.aln | DBTK_ALN_TEXT: writeCigar / writeAnnot run by the kernel)
            p.aln = aln | abi.ALN_TEXT
            gt = run(p, seq, off)
            assert gt["txt"] == exp and (co == gt["counts"]).all() and (o["counters"] == gt["counters"]).all()
        else:
            assert g["aln"] == []


@pytest.mark.parametrize("k,seed,locus_ec", [(21, 3, False), (25, 4, False), (17, 5, False), (21, 3, True), (25, 4, True)])
def test_emulated_pair_mode_equals_oracle(tmp_path, monkeypatch, k, seed, locus_ec):
    """K1..K3 + the walk kernel's pair mode on the emulated lanes: exact counts, all counters, walk results, -a / -ae records.
    locus_ec: with DBTK_WALK_LOCUS_EC=1 (the error-correcting walk of the lean kernel's undecided pairs with the graph image in LDS: built in
    round 5, measured slower, kept opt-in)."""
    if locus_ec:
        monkeypatch.setenv("DBTK_WALK_LOCUS_EC", "1")
    O = bind.Oracle()
    E = bind.Emu()
    case = WalkCase(str(tmp_path), f"ep{k}", k, seed)
    oh = O.load(case.prefix, k); O.load_graph(oh, case.prefix + ".graph.kmers")
    g = E.load(case.prefix, k, flags=abi.LOAD_GRAPH)
    tb = E.tables(g)
    order = g.output_order()

    def run(p, seq, off):
        e = E.align(g, tb, p, seq, off)
        res, nres = E.walk_results(len(off))
        return dict(counts=e["counts"], counters=e["counters"], res=res, nres=nres, aln=E.aln_records(), order=order, txt=E.aln_text(len(off) // 2))
    E.walk_locus_stats()
    E.path_stats()
    check_pair_mode(run, O, oh, case, k, case.loci.nloci)
    ec = E.path_stats()["walk_locus_ec"]
    print(f"k={k}: error-correcting walk with the graph image in LDS: {ec} pairs")
    assert (ec > 100) if locus_ec else ec == 0
    img, plain = E.walk_locus_stats()
    print(f"k={k}: lean walk body: {img} pairs with the graph image in LDS, {plain} from the global tables")
    assert (img > 0 and plain > 0) if k != 17 else img == 0  # (graph images exist where the minimizer-grouped tables do)


# read lengths either side of what a half-wave of the lean walk kernel covers (walkfast_npl, dbtk_walkfast.h): with the minimizer-grouped
# graph table 32 NPL + m - 1 (m = 15 at k = 21: 110 and 174), without it 32 NPL + k - 1 (116, 180).  ADVICE r3: at 113, 116, 175 and 180
# bases the lean kernel was chosen although its lanes do not hold the m-mers of the last windows, and TR k-mers went uncounted.
LIMIT_RLENS = [110, 111, 116, 174, 175, 180]


@pytest.mark.parametrize("rlen", LIMIT_RLENS)
def test_emulated_pair_mode_at_the_lean_kernels_length_limits(tmp_path, rlen):
    O = bind.Oracle()
    E = bind.Emu()
    k = 21
    case = WalkCase(str(tmp_path), f"el{rlen}", k, 3)
    oh = O.load(case.prefix, k); O.load_graph(oh, case.prefix + ".graph.kmers")
    g = E.load(case.prefix, k, flags=abi.LOAD_GRAPH)
    tb = E.tables(g)
    order = g.output_order()

    def run(p, seq, off):
        e = E.align(g, tb, p, seq, off)
        res, nres = E.walk_results(len(off))
        return dict(counts=e["counts"], counters=e["counters"], res=res, nres=nres, aln=E.aln_records(), order=order, txt=E.aln_text(len(off) // 2))
    check_pair_mode(run, O, oh, case, k, case.loci.nloci, rlen=rlen, sets=((0, 0),), npairs=600, lean=rlen <= 174)


def test_oracle_reproduces_golden_g5():
    """The committed vectors of tests/golden/g5_walk_k25 (made by tests/golden/make_golden_walk.py from the reference's own
    functions): alignment records, exact counts in OUT.trkmc.ar order, totals."""
    import refio
    O = bind.Oracle()
    d = os.path.join(_HERE, "golden", "g5_walk_k25")
    pref = os.path.join(d, "pan")
    oh = O.load(pref, 25); O.load_graph(oh, pref + ".graph.kmers")
    tr = synth.read_rpgg_files(pref)
    for tag, aln, tcth, maxc in (("refae", 2, 85, 3), ("refa", 1, 100, 4), ("refg", 0, 85, 3)):
        p = abi.default_params(ksize=25, cthreshold=45, threading=2, aln=aln, okam=0, thread_cth=tcth, correction=1, maxncorrection=maxc)
        rd = refio.read_pairs(os.path.join(d, "reads.fa"), False, p.cthreshold + p.ksize - 1)
        seq, off = rd.packed()
        o = O.align_walk(oh, p, seq, off)
        _, text = expected_aln(O, o, rd, aln, len(tr["tr_cnt"]))
        assert (text if aln else "") == open(os.path.join(d, tag + ".aln.txt")).read()
        out, i = [], 0
        for n in tr["tr_cnt"]:
            n = int(n)
            out.append(o["counts_file"][i:i + n][O.umap_order(tr["tr_ks"][i:i + n]).astype(np.int64)])
            i += n
        ar = np.fromfile(os.path.join(d, tag + ".trkmc.ar"), np.uint64)
        assert ar[0] == len(ar) - 1 and (ar[1:] == np.concatenate(out)).all() and ar[1:].sum() > 0
        tot = [int(l.split()[0]) for l in open(os.path.join(d, tag + ".totals.txt"))]
        c = o["counters"]
        assert tot == [int(x) for x in (c[abi.C_NREADS], c[abi.C_SUBFILTERED], c[abi.C_KMERFILTERED], 0, 0, c[abi.C_LOCUSFILTERED],
                                        c[abi.C_QCFILTERED], c[abi.C_THREADING], c[abi.C_FEASIBLE], 0)]


# ------------------------------------------------------------ GPU vs oracle --
@pytest.mark.gpu
@pytest.mark.parametrize("k,seed,locus_ec", [(21, 3, False), (25, 4, False), (17, 5, False), (21, 3, True), (25, 4, True)])  # (k = 17: no minimizer-grouped tables: the lean walk kernel's single look-ups, the general probe kernel)
def test_gpu_pair_mode_equals_oracle(tmp_path, monkeypatch, k, seed, locus_ec):
    """BASELINE config 4's path (k = 25, -gc 85 3) and config 5's (-ae) through the C-ABI: dbtk_align_batch with
    threading = 2, then dbtk_ctx_counts / dbtk_ctx_walk_results / dbtk_ctx_aln_records, against the oracle.
    locus_ec: the opt-in k_walk_pairs_locus (DBTK_WALK_LOCUS_EC=1, read when the first walking batch is launched: a fresh process per value
    would be exact; within one process the first test to launch a walk decides — so this case runs in a subprocess)."""
    if locus_ec:
        import subprocess, sys
        env = dict(os.environ, DBTK_WALK_LOCUS_EC="1", DBTK_LOCUS_ALWAYS="1")
        r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-x", "-q", "-m", "gpu", "-k", f"test_gpu_pair_mode_equals_oracle and {k}-{seed}-False", "-p", "no:cacheprovider"],
                           env=env, capture_output=True, text=True)
        assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        return
    O = bind.Oracle()
    D = bind.pkg.Dbtk()
    case = WalkCase(str(tmp_path), f"gp{k}", k, seed)
    oh = O.load(case.prefix, k); O.load_graph(oh, case.prefix + ".graph.kmers")
    g = D.load(case.prefix, k, flags=abi.LOAD_GRAPH)
    order = g.output_order()

    def run(p, seq, off):
        ctx = D.context(g, p, device=0)
        ctx.align(seq, off)
        r = ctx.counts()
        res, _, nres = ctx.walk_results(len(off))
        out = dict(counts=r["counts"], counters=r["counters"], res=res, nres=nres, aln=ctx.aln_records(), order=order, txt=ctx.aln_text(len(off) // 2))
        ctx.close()
        return out
    check_pair_mode(run, O, oh, case, k, case.loci.nloci)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", ["0", "5"])
def test_gpu_pair_mode_with_few_graph_info_rows(tmp_path, monkeypatch, rows):
    """The lean walk kernels hand the graph info of a passed-on pair's positions to the error-correction kernel as long as
    there are rows for it (WalkArgs::slow_info); a pair beyond them is looked up again there.  With no rows, and with five (both
    ways inside one batch), the results are the oracle's."""
    monkeypatch.setenv("DBTK_WALK_INFO_ROWS", rows)
    O = bind.Oracle()
    D = bind.pkg.Dbtk()
    case = WalkCase(str(tmp_path), f"gi{rows}", 21, 3)
    oh = O.load(case.prefix, 21); O.load_graph(oh, case.prefix + ".graph.kmers")
    g = D.load(case.prefix, 21, flags=abi.LOAD_GRAPH)
    order = g.output_order()

    def run(p, seq, off):
        ctx = D.context(g, p, device=0)
        ctx.align(seq, off)
        r = ctx.counts()
        res, _, nres = ctx.walk_results(len(off))
        out = dict(counts=r["counts"], counters=r["counters"], res=res, nres=nres, aln=ctx.aln_records(), order=order, txt=ctx.aln_text(len(off) // 2))
        ctx.close()
        return out
    check_pair_mode(run, O, oh, case, 21, case.loci.nloci)


@pytest.mark.gpu
@pytest.mark.parametrize("k,rlen", [(21, r) for r in LIMIT_RLENS] + [(25, 110), (25, 113), (25, 175), (25, 184), (17, 112), (17, 176)])
def test_gpu_pair_mode_at_the_lean_kernels_length_limits(tmp_path, k, rlen):
    O = bind.Oracle()
    D = bind.pkg.Dbtk()
    case = WalkCase(str(tmp_path), f"gl{k}_{rlen}", k, 3)
    oh = O.load(case.prefix, k); O.load_graph(oh, case.prefix + ".graph.kmers")
    g = D.load(case.prefix, k, flags=abi.LOAD_GRAPH)
    order = g.output_order()

    def run(p, seq, off):
        ctx = D.context(g, p, device=0)
        ctx.align(seq, off)
        r = ctx.counts()
        res, _, nres = ctx.walk_results(len(off))
        out = dict(counts=r["counts"], counters=r["counters"], res=res, nres=nres, aln=ctx.aln_records(), order=order, txt=ctx.aln_text(len(off) // 2))
        ctx.close()
        return out
    check_pair_mode(run, O, oh, case, k, case.loci.nloci, rlen=rlen, sets=((0, 0), (2, 1)), npairs=1200,
                    lean=rlen <= (174 if k != 17 else 176))


@pytest.mark.gpu
@pytest.mark.parametrize("k,seed", [(21, 3), (25, 4)])
def test_gpu_walk_equals_oracle(tmp_path, k, seed):
    """The HIP kernel through the C-ABI (dbtk_thread_batch) against the oracle: the same read sets as the
    oracle-vs-reference test (>= 100 k reads per k), every output field."""
    O = bind.Oracle()
    D = bind.pkg.Dbtk()
    case = WalkCase(str(tmp_path), f"g{k}", k, seed)
    oh = O.load(case.prefix, k)
    O.load_graph(oh, case.prefix + ".graph.kmers")
    g = D.load(case.prefix, k, flags=abi.LOAD_GRAPH)
    total, rets = 0, Counter()
    for pi, ps in enumerate(PARAM_SETS):
        p = abi.default_params(ksize=k, threading=2, **ps)
        ctx = D.context(g, p, device=0)
        for si, (sub, indel, nrate) in enumerate([(0.0, 0.0, 0.0), (0.005, 0.001, 0.0), (0.02, 0.005, 0.002), (0.05, 0.02, 0.0)]):
            seqs, loc = make_reads(case.loci, k, 13000 if pi == 0 else 4334, seed=100 * pi + si + k, sub=sub, indel=indel, nrate=nrate,
                                   targeted=720 if si == 0 else 0, wrong_locus=0.03)
            buf, off = pack(seqs)
            recs = ctx.thread(buf, off, np.array(loc, np.uint32))
            for i, s in enumerate(seqs):
                ro, orec = O.thread(oh, loc[i], s, p)
                assert recs[i].ret == ro and same_rec(recs[i], orec), describe("GPU vs oracle", i, ro, orec, recs[i].ret, recs[i], O)
                rets[ro] += 1
                total += 1
        ctx.close()
    assert total >= 100_000
    assert rets[0] > 0 and rets[1] > 0 and rets[2] > 0


@pytest.mark.gpu
def test_gpu_walks_the_reference_graph_fixture(tmp_path):
    """The reference's v1.3 fixture (pan.kmerDBi.umap/.vv, pan.tr.kmers, pan.ntr.kmers, pan.graph.umap) loaded by dbtk_rpgg_load with
    DBTK_LOAD_GRAPH; reads stitched from that graph walked by k_walk_reads (dbtk_thread_batch) and by the oracle."""
    O = bind.Oracle()
    D = bind.pkg.Dbtk()
    for ext in ("umap", "kmers"):
        pref = _legacy_prefix_with(str(tmp_path), ext)
        g = D.load(pref, 21, flags=abi.LOAD_GRAPH)
        oh = O.from_arrays(g.view())
        O.load_graph(oh, pref + ".graph." + ext)
        ks, ms = O.graph_dump(oh, 0)
        for ps in PARAM_SETS[:2]:
            p = abi.default_params(ksize=21, threading=2, **ps)
            ctx = D.context(g, p, device=0)
            rets = Counter()
            for sub in (0.0, 0.01, 0.03):
                seqs = _reads_from_graph(ks, ms, 21, 400, 150, seed=int(100 * sub) + 11, sub=sub)
                buf, off = pack(seqs)
                recs = ctx.thread(buf, off, np.zeros(len(seqs), np.uint32))
                for i, s_ in enumerate(seqs):
                    ro, orec = O.thread(oh, 0, s_, p)
                    assert recs[i].ret == ro and same_rec(recs[i], orec), describe("GPU vs oracle", i, ro, orec, recs[i].ret, recs[i], O)
                    rets[ro] += 1
            assert rets[1] > 0 and rets[2] > 0
            ctx.close()
        O.free(oh)
        g.close()


if __name__ == "__main__":  # worker of test_oracle_walk_equals_reference
    oracle_vs_reference(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
    glue_oracle_vs_reference(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))


@pytest.mark.gpu
@pytest.mark.parametrize("aln", [1, 2])
def test_gpu_device_writer_lines_equal_the_oracles(tmp_path, aln):
    """dbtk_ingest_aln_lines through the C-ABI: the raw bytes of an interleaved FASTA in, writeAlignments' lines out (AQ.cpp:1742-1759),
    assembled on the device from the parsed block and the walk's text records — as text and as gzip members — against the lines
    built from the oracle's alignment strings (which the reference's own writeCigar / writeAnnot pin)."""
    import gzip
    k = 21
    O = bind.Oracle()
    D = bind.pkg.Dbtk()
    case = WalkCase(str(tmp_path), f"dw{aln}", k, 3)
    oh = O.load(case.prefix, k); O.load_graph(oh, case.prefix + ".graph.kmers")
    g = D.load(case.prefix, k, flags=abi.LOAD_GRAPH)
    p = abi.default_params(ksize=k, cthreshold=45, threading=2, aln=aln, okam=0, **PARAM_SETS[1])
    reads = pair_reads(case, seed=77)
    minread = 45 + k - 1
    keep = [q for q in range(reads.npairs) if len(reads.seqs[2 * q]) >= minread and len(reads.seqs[2 * q + 1]) >= minread]
    sub = synth.Reads()
    for q in keep:
        sub.seqs += [reads.seqs[2 * q], reads.seqs[2 * q + 1]]
        sub.titles.append(reads.titles[q])
    seq, off = sub.packed()
    o = O.align_walk(oh, p, seq, off)
    exp, _ = expected_aln(O, o, sub, aln, case.loci.nloci)
    want = b"".join(b".\t%d\t>%s\t%s\t%s\t%s\n" % (dst, sub.titles[pr].encode(), sub.seqs[2 * pr + 1], sub.seqs[2 * pr], t.encode()) for pr, dst, t in exp)
    assert want.count(b"\n") > 100
    data = b"".join(b">" + reads.titles[q].encode() + b"/2\n" + reads.seqs[2 * q + 1] + b"\n>" + reads.titles[q].encode() + b"/1\n" + reads.seqs[2 * q] + b"\n"
                    for q in range(reads.npairs))
    p.aln = aln | abi.ALN_TEXT
    for gz, chunk in ((False, 1 << 22), (True, 1 << 22), (True, 70000)):
        ctx = D.context(g, p, device=0)
        ing = bind.pkg.Ingest(ctx, False, minread, chunk, nslots=3, with_spans=True)
        got, nl = b"", 0
        nblocks = (len(data) + chunk - 1) // chunk
        for j in range(nblocks):  # (one block at a time: the lines are those of the block aligned last)
            slot = ing.submit(data[j * chunk:(j + 1) * chunk], j == nblocks - 1)
            info = ing.wait(slot)
            assert info.flags == 0
            ing.align(slot, info, sync=True)
            b, n, tb = ing.aln_lines(slot, gz=gz)
            got += b; nl += n
            assert gz or tb == len(b)
        ing.close(); ctx.close()
        assert (gzip.decompress(got) if gz else got) == want and nl == want.count(b"\n"), (gz, chunk)
