"""Seeded synthetic RPGG + read generator for the parity tests (numpy only).

Nothing here comes from the reference: loci are random flank + motif-repeat
sequences, haplotypes differ by copy number and point variation, reads are
150 bp PE tiles with optional substitutions / indels / N / lowercase, chimeric
pairs, background pairs and short reads.  RPGG *files* are produced from the
haplotype FASTAs by the reference's own tools (oracle/_ref/fa2kmers + ktools
serialize; recipe of SURVEY.md 8c) so the inputs of every parity test are in
the reference's on-disk formats.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from dataclasses import dataclass, field

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    COMP[a] = b


def ref_tool(name: str) -> str:
    p = os.path.join(REFDIR, name)
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p} missing: run `make -C oracle ref` where /root/reference exists")
    return p


def have_ref() -> bool:
    if os.environ.get("DBTK_NO_REF"):  # (make asan: the compiled reference is not sanitizer-clean and not ours to fix)
        return False
    return all(os.path.exists(os.path.join(REFDIR, t)) for t in ("danbing-tk", "ktools", "fa2kmers", "libdbtk_refharness.so"))


def revcomp(s: np.ndarray) -> np.ndarray:
    return COMP[s[::-1]]


@dataclass
class Loci:
    flank: int
    haps: list  # haps[h][l] = uint8 array: flank + TR + flank
    nloci: int = 0
    nhap: int = 0


def make_loci(nloci=3, nhap=3, flank=500, tr_min=60, tr_max=900, seed=1, shared_frac=0.0, motif_min=5, motif_max=40,
              variation=0.15) -> Loci:
    """Random loci.  With shared_frac > 0 a locus copies 300 bp of its left
    flank from the previous locus (k-mers shared by loci -> odd `val` / vv)."""
    rng = np.random.default_rng(seed)
    haps = [[] for _ in range(nhap)]
    prev_lf = None
    for l in range(nloci):
        lf = BASES[rng.integers(0, 4, flank)]
        rf = BASES[rng.integers(0, 4, flank)]
        if prev_lf is not None and rng.random() < shared_frac:
            n = min(300, flank)
            lf[flank - n:] = prev_lf[flank - n:]
        prev_lf = lf
        motif = BASES[rng.integers(0, 4, rng.integers(motif_min, motif_max + 1))]
        trlen = int(np.exp(rng.uniform(np.log(tr_min), np.log(tr_max))))
        ncopy = max(2, trlen // len(motif))
        for h in range(nhap):
            nc = max(1, ncopy + int(rng.integers(-3, 4)))
            copies = []
            for _ in range(nc):
                m = motif.copy()
                if rng.random() < variation:
                    m[rng.integers(0, len(m))] = BASES[rng.integers(0, 4)]
                copies.append(m)
            tr = np.concatenate(copies)
            haps[h].append(np.concatenate([lf, tr, rf]))
    return Loci(flank=flank, haps=haps, nloci=nloci, nhap=nhap)


def write_hap_fastas(loci: Loci, outdir: str) -> list:
    os.makedirs(outdir, exist_ok=True)
    fns = []
    for h, hap in enumerate(loci.haps):
        fn = os.path.join(outdir, f"h{h}.fa")
        with open(fn, "wb") as f:
            for l, s in enumerate(hap):
                f.write(b">locus%d\n" % l)
                f.write(s.tobytes() + b"\n")
        fns.append(fn)
    return fns


def build_rpgg_with_reference(loci: Loci, outdir: str, k=21, name="pan") -> str:
    """SURVEY.md 8c recipe: fa2kmers (tr/fl/graph), fa2kmers -tr -k k+1 (tre),
    ktools serialize.  Returns the RPGG prefix."""
    fas = write_hap_fastas(loci, outdir)
    pref = os.path.join(outdir, name)
    fs = str(loci.flank)
    run = lambda *a: subprocess.run(list(a), check=True, cwd=outdir, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    run(ref_tool("fa2kmers"), "-g", "-k", str(k), "-fsi", fs, "-fso", fs, "-on", name, "-fa", str(len(fas)), *fas)
    run(ref_tool("fa2kmers"), "-tr", "-k", str(k + 1), "-fsi", fs, "-fso", fs, "-on", name + "e", "-fa", str(len(fas)), *fas)
    shutil.move(os.path.join(outdir, name + "e.tr.kmers"), pref + ".tre.kmers")
    run(ref_tool("ktools"), "serialize", name)
    return pref


@dataclass
class Reads:
    seqs: list = field(default_factory=list)    # list of bytes; read 2p, 2p+1 = pair p
    titles: list = field(default_factory=list)  # one per pair (no /1 /2)
    quals: list = field(default_factory=list)

    def packed(self):
        off = np.zeros(len(self.seqs) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(s) for s in self.seqs], dtype=np.uint64)
        buf = np.frombuffer(b"".join(self.seqs), dtype=np.uint8).copy() if self.seqs else np.zeros(0, np.uint8)
        return buf, off

    @property
    def npairs(self):
        return len(self.seqs) // 2


def _mutate(rng, s: np.ndarray, sub, indel, nrate, lower) -> np.ndarray:
    s = s.copy()
    if sub > 0:
        m = rng.random(len(s)) < sub
        s[m] = BASES[rng.integers(0, 4, int(m.sum()))]
    if indel > 0 and rng.random() < indel * len(s):
        p = int(rng.integers(1, len(s) - 1))
        if rng.random() < 0.5:
            s = np.concatenate([s[:p], BASES[rng.integers(0, 4, 1)], s[p:-1]])
        else:
            s = np.concatenate([s[:p], s[p + 1:], BASES[rng.integers(0, 4, 1)]])
    if nrate > 0:
        m = rng.random(len(s)) < nrate
        s[m] = ord("N")
    if lower > 0 and rng.random() < lower:
        p = int(rng.integers(0, len(s)))
        s[p] = s[p] | 0x20
    return s


def sim_reads(loci: Loci, npairs=1000, rlen=150, frag=(300, 500), seed=2, sub=0.0, indel=0.0, nrate=0.0, lower=0.0,
              chimeric=0.0, background=0.0, short=0.0, with_qual=False, splice=0.0) -> Reads:
    """Pairs in the orientation the reference's reader hands to the hot loop:
    seqs[2p] and seqs[2p+1] are simply the two mates."""
    rng = np.random.default_rng(seed)
    out = Reads()
    for p in range(npairs):
        u = rng.random()
        if u < background:
            a = BASES[rng.integers(0, 4, rlen)]
            b = BASES[rng.integers(0, 4, rlen)]
            name = "bg"
        else:
            def mate_from(l, h):
                s = loci.haps[h][l]
                fl = int(rng.integers(frag[0], frag[1] + 1))
                fl = min(fl, len(s))
                beg = int(rng.integers(0, len(s) - fl + 1))
                return s[beg:beg + rlen], revcomp(s[beg + fl - rlen:beg + fl])
            l = int(rng.integers(0, loci.nloci)); h = int(rng.integers(0, loci.nhap))
            a, b = mate_from(l, h)
            name = f"l{l}h{h}"
            if rng.random() < chimeric:
                l2 = int(rng.integers(0, loci.nloci)); h2 = int(rng.integers(0, loci.nhap))
                _, b = mate_from(l2, h2)
                name += f"x{l2}"
            if splice and rng.random() < splice:
                # the tail of one mate comes from ANOTHER locus: a pair that is its locus' by every count (cth k-mers per mate), with
                # a stretch of k-mers that are in the index but not this locus' (the fused probe kernel must take such a pair back)
                l2 = int(rng.integers(0, loci.nloci)); h2 = int(rng.integers(0, loci.nhap))
                s2 = loci.haps[h2][l2]
                n2 = int(rng.integers(28, 50))
                at = int(rng.integers(0, len(s2) - n2 + 1))
                b = np.concatenate([b[:len(b) - n2], s2[at:at + n2]])
            if rng.random() < 0.5:
                a, b = b, a
        a = _mutate(rng, a, sub, indel, nrate, lower)
        b = _mutate(rng, b, sub, indel, nrate, lower)
        if rng.random() < short:
            n = int(rng.integers(0, 70))
            if rng.random() < 0.5:
                a = a[:n]
            else:
                b = b[:n]
        out.seqs += [a.tobytes(), b.tobytes()]
        out.titles.append(f"r{p}:{name}")
        if with_qual:
            def q(n):  # mostly Q30-40, ~2 % low-quality bases, occasionally a low-quality tail
                v = rng.integers(33 + 30, 33 + 41, n, dtype=np.uint8)
                v[rng.random(n) < 0.02] = 33 + int(rng.integers(2, 15))
                if n and rng.random() < 0.2:
                    v[int(rng.integers(n // 2, n)):] = 33 + 8
                return bytes(v)
            out.quals += [q(len(a)), q(len(b))]
    return out


def write_fasta(reads: Reads, fn: str, fastq=False, interleave=True):
    """Mates are written so that the reference's reader (AQ.cpp:1918-1976)
    rebuilds seqs[2p] = reads.seqs[2p]: it makes `seq1` the LATER record of a
    title and `seq2` the parked one, so mate 2p+1 is written first."""
    with open(fn, "wb") as f:
        for p in range(reads.npairs):
            t = reads.titles[p].encode()
            for which, tag in ((2 * p + 1, b"/2"), (2 * p, b"/1")):
                if fastq:
                    f.write(b"@" + t + tag + b"\n" + reads.seqs[which] + b"\n+\n" + reads.quals[which] + b"\n")
                else:
                    f.write(b">" + t + tag + b"\n" + reads.seqs[which] + b"\n")


def read_rpgg_files(pref: str):
    """Flat arrays of the HEAD on-disk RPGG (layouts: SURVEY.md 2.3)."""
    with open(pref + ".kmers.dbi", "rb") as f:
        nk = int(np.frombuffer(f.read(8), np.uint64)[0])
        keys = np.frombuffer(f.read(8 * nk), np.uint64).copy()
        vals = np.frombuffer(f.read(4 * nk), np.uint32).copy()
        nvv = int(np.frombuffer(f.read(8), np.uint64)[0])
        vv = np.frombuffer(f.read(4 * nvv), np.uint32).copy()

    def kdb(fn):
        with open(fn, "rb") as f:
            nl = int(np.frombuffer(f.read(8), np.uint64)[0])
            cnt = np.frombuffer(f.read(8 * nl), np.uint64).copy()
            n = int(np.frombuffer(f.read(8), np.uint64)[0])
            ks = np.frombuffer(f.read(8 * n), np.uint64).copy()
        return cnt, ks

    fl_cnt, fl_ks = kdb(pref + ".fl.kdb")
    tre_cnt, tre_ks = kdb(pref + ".tre.kdb")
    tr_cnt, tr_ks = [], []
    with open(pref + ".tr.kmers") as f:
        for line in f:
            if line[0] == ">":
                tr_cnt.append(0)
            else:
                tr_ks.append(int(line.split()[0]))
                tr_cnt[-1] += 1
    return dict(keys=keys, vals=vals, vv=vv, fl_cnt=fl_cnt, fl_ks=fl_ks, tre_cnt=tre_cnt, tre_ks=tre_ks,
                tr_cnt=np.array(tr_cnt, np.uint64), tr_ks=np.array(tr_ks, np.uint64), nloci=len(tr_cnt))


# --------------------------------------------------------------------------
# Array-level RPGG builder (numpy).  Used for cases the reference's tools are
# too slow or too big for (hundreds of loci sharing k-mers); oracle and product
# both consume the SAME arrays, so this builder only has to be self-consistent,
# not identical to fa2kmers/ktools.
CODE = np.full(256, -1, dtype=np.int64)
for _i, _c in enumerate(b"ACGT"):
    CODE[_c] = _i


def canon_kmers(seq: np.ndarray, k: int):
    """(canonical k-mers, valid mask) for every window of seq (uint8 ASCII)."""
    c = CODE[seq]
    n = len(seq) - k + 1
    if n <= 0:
        return np.zeros(0, np.uint64), np.zeros(0, bool)
    win = np.lib.stride_tricks.sliding_window_view(c, k)
    valid = (win >= 0).all(axis=1)
    w = np.where(win < 0, 0, win).astype(np.uint64)
    sh = (2 * np.arange(k - 1, -1, -1)).astype(np.uint64)
    fw = (w << sh).sum(axis=1, dtype=np.uint64)
    rc = ((np.uint64(3) - w[:, ::-1]) << sh).sum(axis=1, dtype=np.uint64)
    return np.minimum(fw, rc), valid


def _uniq_in_order(a):
    _, idx = np.unique(a, return_index=True)
    return a[np.sort(idx)]


def build_rpgg_arrays(loci: Loci, k=21):
    """Flat RPGG (the layout of read_rpgg_files) straight from the haplotypes."""
    fs = loci.flank
    tr_cnt, tr_ks, fl_cnt, fl_ks, tre_cnt, tre_ks = [], [], [], [], [], []
    for l in range(loci.nloci):
        tr, fl, tre = [], [], []
        for h in range(loci.nhap):
            s = loci.haps[h][l]
            ks, ok = canon_kmers(s, k)
            n = len(ks)
            pos = np.arange(n)
            in_tr = (pos >= fs) & (pos <= len(s) - fs - k)
            tr.append(ks[ok & in_tr])
            fl.append(ks[ok & ~in_tr])
            es, eok = canon_kmers(s, k + 1)
            epos = np.arange(len(es))
            tre.append(es[eok & (epos >= fs) & (epos <= len(s) - fs - k - 1)])
        tr = _uniq_in_order(np.concatenate(tr))
        fl = _uniq_in_order(np.concatenate(fl))
        tre = _uniq_in_order(np.concatenate(tre))
        tr_cnt.append(len(tr)); tr_ks.append(tr)
        fl_cnt.append(len(fl)); fl_ks.append(fl)
        tre_cnt.append(len(tre)); tre_ks.append(tre)
    # index: k-mer -> locus<<1 | vv offset<<1|1 (readKmerIndex + serialize semantics: tr loci first, then fl)
    allk = np.concatenate(tr_ks + fl_ks)
    alll = np.concatenate([np.full(n, l, np.uint32) for l, n in enumerate(tr_cnt)] +
                          [np.full(n, l, np.uint32) for l, n in enumerate(fl_cnt)])
    order = np.argsort(allk, kind="stable")
    sk, sl = allk[order], alll[order]
    starts = np.r_[0, np.nonzero(sk[1:] != sk[:-1])[0] + 1, len(sk)]
    keys, vals, vv = [], [], []
    for a, b in zip(starts[:-1], starts[1:]):
        ls = _uniq_in_order(sl[a:b])
        keys.append(sk[a])
        if len(ls) == 1:
            vals.append(int(ls[0]) << 1)
        else:
            vals.append((len(vv) << 1) | 1)
            vv.append(len(ls))
            vv.extend(int(x) for x in ls)
    rng = np.random.default_rng(12345)
    perm = rng.permutation(len(keys))  # the on-disk key order is arbitrary (hash-map iteration order)
    return dict(keys=np.array(keys, np.uint64)[perm], vals=np.array(vals, np.uint32)[perm], vv=np.array(vv, np.uint32),
                fl_cnt=np.array(fl_cnt, np.uint64), fl_ks=np.concatenate(fl_ks).astype(np.uint64),
                tre_cnt=np.array(tre_cnt, np.uint64), tre_ks=np.concatenate(tre_ks).astype(np.uint64),
                tr_cnt=np.array(tr_cnt, np.uint64), tr_ks=np.concatenate(tr_ks).astype(np.uint64), nloci=loci.nloci)


def fw_kmers(seq: np.ndarray, k: int) -> np.ndarray:
    """Non-canonical k-mers of every window of an all-ACGT sequence."""
    c = CODE[seq].astype(np.uint64)
    win = np.lib.stride_tricks.sliding_window_view(c, k)
    sh = (2 * np.arange(k - 1, -1, -1)).astype(np.uint64)
    return (win << sh).sum(axis=1, dtype=np.uint64)


def build_graph_arrays(loci: Loci, k=21):
    """graphDB as buildKmerGraph makes it (src/aQueryFasta_thread.h:215-243: both strands of every haplotype,
    out-edge bit per successor base, no self loops, the last k-mer of a strand present with whatever mask it has).
    Returns (gr_cnt, gr_ks, gr_ms), nodes sorted within a locus."""
    cnt, ks, ms = [], [], []
    for l in range(loci.nloci):
        g = {}
        for h in range(loci.nhap):
            s = loci.haps[h][l]
            for strand in (s, revcomp(s)):
                f = fw_kmers(strand, k)
                nxt = CODE[strand[k:]]
                for i in range(len(f) - 1):
                    a, b = int(f[i]), int(f[i + 1])
                    g[a] = g.get(a, 0) | ((1 << int(nxt[i])) if a != b else 0)
                g[int(f[-1])] = g.get(int(f[-1]), 0)
        nodes = sorted(g)
        cnt.append(len(nodes)); ks += nodes; ms += [g[n] for n in nodes]
    return np.array(cnt, np.uint64), np.array(ks, np.uint64), np.array(ms, np.uint8)


def write_graph_file(gr, pref: str, binary=False):
    """PREF.graph.kmers (text, what fa2kmers -g writes) or the v1.3 binary PREF.graph.umap."""
    cnt, ks, ms = gr
    if binary:
        with open(pref + ".graph.umap", "wb") as f:
            f.write(np.uint64(len(cnt)).tobytes())
            i = 0
            for n in cnt:
                n = int(n)
                f.write(np.uint64(n).tobytes())
                rec = np.zeros(n, dtype=[("k", "<u8"), ("m", "u1")])
                rec["k"] = ks[i:i + n]; rec["m"] = ms[i:i + n]
                f.write(rec.tobytes())
                i += n
        return
    with open(pref + ".graph.kmers", "w") as f:
        i = 0
        for l, n in enumerate(cnt):
            f.write(f">{l}\n")
            for j in range(i, i + int(n)):
                f.write(f"{int(ks[j])}\t{int(ms[j])}\n")
            i += int(n)


def write_rpgg_files(arr, pref: str):
    """HEAD on-disk formats from flat arrays."""
    with open(pref + ".kmers.dbi", "wb") as f:
        f.write(np.uint64(len(arr["keys"])).tobytes() + arr["keys"].tobytes() + arr["vals"].tobytes())
        f.write(np.uint64(len(arr["vv"])).tobytes() + arr["vv"].tobytes())
    for tag, c, ks in (("fl", "fl_cnt", "fl_ks"), ("tre", "tre_cnt", "tre_ks")):
        with open(f"{pref}.{tag}.kdb", "wb") as f:
            f.write(np.uint64(len(arr[c])).tobytes() + arr[c].tobytes() + np.uint64(len(arr[ks])).tobytes() + arr[ks].tobytes())
    with open(pref + ".tr.kmers", "w") as f:
        i = 0
        for l, n in enumerate(arr["tr_cnt"]):
            f.write(f">{l}\n")
            for km in arr["tr_ks"][i:i + int(n)]:
                f.write(f"{int(km)}\t0\n")
            i += int(n)


def make_bait_db(loci: Loci, reads: Reads, outdir: str, k=21, seed=5, per_locus=40, name="pan"):
    """A bait DB (PREF.bt.kmdb) for the -b gate, serialised by the reference's own `ktools serialize-bt`
    from a text file: for every locus a few k-mers taken from the READS with (min, max) thresholds —
    (255, 0) "any occurrence is a false positive", (2, 9) "at least twice", (0, 1) "at most once"."""
    rng = np.random.default_rng(seed)
    pool = []
    for s in reads.seqs[: 2 * 400]:
        ks, ok = canon_kmers(np.frombuffer(s, np.uint8), k)
        pool.append(ks[ok])
    pool = np.unique(np.concatenate(pool)) if pool else np.zeros(0, np.uint64)
    fn = os.path.join(outdir, name + ".bait.txt")
    with open(fn, "w") as f:
        for l in range(loci.nloci):
            f.write(f">{l}\n")
            if len(pool) == 0:
                continue
            for km in rng.choice(pool, size=min(per_locus, len(pool)), replace=False):
                mi, ma = [(255, 0), (2, 9), (0, 1), (1, 1)][int(rng.integers(0, 4))]
                f.write(f"{int(km)}\t{mi}\t{ma}\n")
    subprocess.run([ref_tool("ktools"), "serialize-bt", fn, str(loci.nloci), os.path.join(outdir, name)], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.join(outdir, name + ".bt.kmdb")
