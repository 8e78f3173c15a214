"""Named parity cases shared by the CPU (emulator) and GPU suites."""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@dataclass
class Case:
    prefix: str
    k: int
    reads: synth.Reads
    param_sets: list
    qc_file: str | None = None


def _rpgg(tmp, name, loci, k):
    """RPGG files for a case: the reference's own tools when oracle/_ref is
    there (this container, and the GPU box: the static binaries travel),
    otherwise the numpy builder."""
    d = os.path.join(tmp, name)
    os.makedirs(d, exist_ok=True)
    if synth.have_ref():
        return synth.build_rpgg_with_reference(loci, d, k=k)
    pref = os.path.join(d, "pan")
    synth.write_rpgg_files(synth.build_rpgg_arrays(loci, k), pref)
    return pref


def _np_rpgg(tmp, name, loci, k):
    d = os.path.join(tmp, name)
    os.makedirs(d, exist_ok=True)
    pref = os.path.join(d, "pan")
    synth.write_rpgg_files(synth.build_rpgg_arrays(loci, k), pref)
    return pref


def case_clean(tmp):
    loci = synth.make_loci(nloci=12, nhap=3, flank=500, seed=5)
    return Case(_rpgg(tmp, "clean", loci, 21), 21, synth.sim_reads(loci, npairs=600, seed=11),
                [dict(cthreshold=45, okam=1), dict(cthreshold=10, okam=0)])


def case_mixed(tmp):
    loci = synth.make_loci(nloci=12, nhap=3, flank=500, seed=5)
    reads = synth.sim_reads(loci, npairs=900, seed=12, sub=0.01, indel=0.002, nrate=0.003, lower=0.05, chimeric=0.3,
                            background=0.2, short=0.05)
    return Case(_rpgg(tmp, "mixed", loci, 21), 21, reads, [dict(cthreshold=45, okam=1), dict(cthreshold=10, okam=1)])


def case_shared(tmp):
    """Loci sharing flank k-mers (odd val / vv) + chimeric pairs: the unstable-sort tie order decides."""
    loci = synth.make_loci(nloci=30, nhap=2, flank=500, seed=7, shared_frac=0.8)
    reads = synth.sim_reads(loci, npairs=900, seed=13, sub=0.005, chimeric=0.5)
    return Case(_rpgg(tmp, "shared", loci, 21), 21, reads, [dict(cthreshold=45, okam=1), dict(cthreshold=20, okam=0)])


def case_k25(tmp):
    loci = synth.make_loci(nloci=10, nhap=3, flank=500, seed=9, shared_frac=0.3)
    reads = synth.sim_reads(loci, npairs=500, seed=14, sub=0.01, nrate=0.002, chimeric=0.2, background=0.1)
    return Case(_rpgg(tmp, "k25", loci, 25), 25, reads, [dict(cthreshold=45, okam=1), dict(cthreshold=45, okam=0)])


def case_k31(tmp):
    """Largest k (62-bit k-mers): the sort falls back from the composite-key bitonic network to two-word keys."""
    loci = synth.make_loci(nloci=8, nhap=2, flank=500, seed=10, shared_frac=0.3)
    reads = synth.sim_reads(loci, npairs=400, seed=19, sub=0.004, chimeric=0.2, background=0.1)
    return Case(_np_rpgg(tmp, "k31", loci, 31), 31, reads, [dict(cthreshold=40, okam=1)])


def case_k31long(tmp):
    """k = 31 with 250 bp reads: a lane owns four positions, and their windows (31 + 3 bases) do not fit the one 32-base word
    the probe kernel usually shifts them out of; k = 30 / 29 sit on the boundary.  Includes N (the validity path)."""
    loci = synth.make_loci(nloci=8, nhap=2, flank=600, seed=31, shared_frac=0.3, tr_min=200, tr_max=1200)
    reads = synth.sim_reads(loci, npairs=300, rlen=250, frag=(300, 600), seed=32, sub=0.004, nrate=0.001, chimeric=0.1, background=0.1)
    return Case(_np_rpgg(tmp, "k31long", loci, 31), 31, reads, [dict(cthreshold=60, okam=1)])


def case_k30long(tmp):
    loci = synth.make_loci(nloci=6, nhap=2, flank=600, seed=33, shared_frac=0.3, tr_min=200, tr_max=900)
    reads = synth.sim_reads(loci, npairs=250, rlen=200, frag=(300, 600), seed=34, sub=0.004, chimeric=0.1, background=0.1)
    return Case(_np_rpgg(tmp, "k30long", loci, 30), 30, reads, [dict(cthreshold=50, okam=1)])


def case_qc(tmp):
    loci = synth.make_loci(nloci=12, nhap=3, flank=500, seed=5)
    pref = _rpgg(tmp, "qc", loci, 21)
    qc = os.path.join(tmp, "qc", "qc.txt")
    with open(qc, "w") as f:
        f.write("".join("01"[i % 3 != 0] for i in range(12)))
    return Case(pref, 21, synth.sim_reads(loci, npairs=500, seed=15, sub=0.005), [dict(cthreshold=45, okam=1, qc=1), dict(cthreshold=45, okam=0, qc=1)], qc)


def case_lengths(tmp):
    """Read lengths from below k to DBTK_MAX_READ_LEN (4 k-mer positions per lane)."""
    loci = synth.make_loci(nloci=8, nhap=2, flank=500, seed=21, tr_min=200, tr_max=900)
    rng = np.random.default_rng(3)
    reads = synth.Reads()
    for L in (250, 256, 101, 66, 36, 151):
        r = synth.sim_reads(loci, npairs=80, rlen=L, seed=int(rng.integers(1 << 30)), sub=0.004, frag=(max(300, L), 520))
        reads.seqs += r.seqs
        reads.titles += [f"L{L}_{t}" for t in r.titles]
    return Case(_rpgg(tmp, "lengths", loci, 21), 21, reads, [dict(cthreshold=45, okam=1), dict(cthreshold=10, okam=1)])


def case_short21(tmp):
    """Reads of at most 110 bases, k = 21: the probe kernel's three-positions-per-lane form (window of 7 m-mers);
    lengths down to k + 15, N bases (the exact-validity path of that form)."""
    loci = synth.make_loci(nloci=8, nhap=2, flank=500, seed=23, tr_min=200, tr_max=900)
    rng = np.random.default_rng(4)
    reads = synth.Reads()
    for L in (100, 110, 66, 36):
        r = synth.sim_reads(loci, npairs=120, rlen=L, seed=int(rng.integers(1 << 30)), sub=0.004, nrate=0.002, frag=(max(300, L), 520))
        reads.seqs += r.seqs
        reads.titles += [f"L{L}_{t}" for t in r.titles]
    return Case(_rpgg(tmp, "short21", loci, 21), 21, reads, [dict(cthreshold=30, okam=1), dict(cthreshold=10, okam=0)])


def case_short25(tmp):
    """The same form at k = 25 (window of 11 m-mers)."""
    loci = synth.make_loci(nloci=8, nhap=2, flank=500, seed=24, shared_frac=0.3, tr_min=200, tr_max=900)
    reads = synth.sim_reads(loci, npairs=400, rlen=100, seed=25, sub=0.006, nrate=0.001, chimeric=0.2, background=0.1)
    return Case(_np_rpgg(tmp, "short25", loci, 25), 25, reads, [dict(cthreshold=30, okam=1), dict(cthreshold=30, okam=0)])


def case_kf(tmp):
    loci = synth.make_loci(nloci=12, nhap=3, flank=500, seed=5)
    reads = synth.sim_reads(loci, npairs=600, seed=16, sub=0.03, background=0.3, nrate=0.002)
    return Case(_rpgg(tmp, "kf", loci, 21), 21, reads,
                [dict(cthreshold=45, okam=0, n_filter=0, nm_filter=0), dict(cthreshold=45, okam=1, n_filter=8, nm_filter=2),
                 dict(cthreshold=30, okam=0, n_filter=2, nm_filter=1), dict(cthreshold=45, okam=1, extract=2)])


def case_spliced(tmp):
    """Pairs that are their locus' by every count but carry a stretch of another locus' k-mers (the last 28-49 bases of a mate): the
    locus-resident probe kernel resolves a pair AHEAD of the look-ups in the plain index (dbtk_locus.h: FUSE) and must take these back."""
    loci = synth.make_loci(nloci=10, nhap=3, flank=500, seed=51)
    reads = synth.sim_reads(loci, npairs=800, seed=52, sub=0.003, splice=0.4)
    return Case(_rpgg(tmp, "spliced", loci, 21), 21, reads, [dict(cthreshold=45, okam=0), dict(cthreshold=30, okam=0), dict(cthreshold=60, okam=0)])


def case_spill(tmp):
    """450 loci sharing 300 bp of flank: one pair touches > LLIMIT loci, the
    per-pair locus map spills from LDS to the stamped HBM array."""
    loci = synth.make_loci(nloci=450, nhap=1, flank=400, seed=31, shared_frac=1.0, tr_min=60, tr_max=120)
    reads = synth.sim_reads(loci, npairs=300, seed=17, sub=0.002, frag=(300, 420), chimeric=0.2)
    return Case(_np_rpgg(tmp, "spill", loci, 21), 21, reads, [dict(cthreshold=30, okam=1), dict(cthreshold=45, okam=0)])


def case_inconsistent(tmp):
    """Index and flank/TR sets that do NOT describe the same memberships (files mixed from different
    builds): the load-time check must notice and the kernels must use the general class-table path."""
    loci = synth.make_loci(nloci=10, nhap=2, flank=500, seed=41, shared_frac=0.4)
    arr = synth.build_rpgg_arrays(loci, 21)
    rng = np.random.default_rng(5)
    keep = rng.random(len(arr["keys"])) > 0.1          # 10 % of the index keys dropped
    arr["keys"], arr["vals"] = arr["keys"][keep], arr["vals"][keep]
    drop = rng.random(len(arr["fl_ks"])) < 0.05          # and 5 % of the flank k-mers dropped from fl.kdb
    cnt, out, i = [], [], 0
    for n in arr["fl_cnt"]:
        seg = arr["fl_ks"][i:i + int(n)][~drop[i:i + int(n)]]
        out.append(seg); cnt.append(len(seg)); i += int(n)
    arr["fl_ks"], arr["fl_cnt"] = np.concatenate(out), np.array(cnt, np.uint64)
    d = os.path.join(tmp, "inconsistent")
    os.makedirs(d, exist_ok=True)
    synth.write_rpgg_files(arr, os.path.join(d, "pan"))
    reads = synth.sim_reads(loci, npairs=500, seed=18, sub=0.004, chimeric=0.2)
    return Case(os.path.join(d, "pan"), 21, reads, [dict(cthreshold=30, okam=1), dict(cthreshold=45, okam=0)])


CASES = dict(clean=case_clean, mixed=case_mixed, shared=case_shared, k25=case_k25, k31=case_k31, k31long=case_k31long, k30long=case_k30long,
             qc=case_qc, lengths=case_lengths, short21=case_short21, short25=case_short25,
             kf=case_kf, spill=case_spill, inconsistent=case_inconsistent, spliced=case_spliced)


# cases whose geometry the lean probe kernel (dbtk_probe2.h) takes; the others run the general one
LEAN_PROBE = {"clean", "mixed", "shared", "k25", "qc", "short21", "short25", "kf", "spill", "inconsistent", "spliced"}


def make_case(name, tmp) -> Case:
    return CASES[name](tmp)
