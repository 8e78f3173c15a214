"""danbing-tk-pred: the oracle restatement of src/pred.h (CPU) and the HIP path against it (GPU).

Floating point (IEEE float32 like the reference's Eigen arrays).  Tolerances, as written in include/dbtk_pred.h:
  raw matrix        bit-exact (one conversion + one division per entry)
  Bias, corrected   relative 2e-6: the per-sample sums are taken in the same order, the mean over the samples is a
                    pairwise tree on the GPU, numpy's pairwise sum in the oracle and a packet reduction in Eigen
PARITY UNPINNED against the reference itself: pred.cpp needs Eigen, which this image lacks."""
import ctypes as C
import os
import re
import struct
import subprocess
import sys

import numpy as np
import pytest

import bind

sys.path.insert(0, os.path.join(bind.ROOT, "oracle"))
import pred_oracle as PO  # noqa: E402

pkg = bind.pkg
RTOL = 2e-6


def make_cohort(seed, ns=37, ntr=50, max_k=160):
    """Counts of ns samples over ntr loci: some loci without k-mers, some without invariant k-mers, one zero-depth column."""
    rng = np.random.default_rng(seed)
    nks = rng.integers(0, max_k, ntr)
    nks[rng.integers(0, ntr, 3)] = 0                      # loci without k-mers
    nk_cum = np.cumsum(nks).astype(np.uint32)
    nk = int(nk_cum[-1])
    iki, ikmc, nik_cum = [], [], []
    for t in range(ntr):
        si = int(nk_cum[t - 1]) if t else 0
        n = int(nks[t])
        m = 0 if (n == 0 or t % 7 == 3) else int(rng.integers(1, max(2, n // 3)))   # loci without invariant k-mers
        sel = np.sort(rng.choice(n, m, replace=False)) + si if m else np.zeros(0, np.int64)
        iki += list(sel)
        ikmc += list(rng.integers(1, 5, m))
        nik_cum.append(len(iki))
    depths = rng.uniform(8, 60, ns).astype(np.float32)
    lam = rng.uniform(0.5, 3.0, nk)
    lam[np.asarray(iki, np.int64)] = np.asarray(ikmc, np.float64)
    locus_bias = rng.uniform(0.7, 1.4, (ns, ntr))
    kbias = np.repeat(locus_bias, nks, axis=1) if nk else np.zeros((ns, 0))
    counts = rng.poisson(lam[None, :] * depths[:, None] * kbias).astype(np.uint64)
    counts[:, rng.integers(0, max(nk, 1), 5) % max(nk, 1)] += np.uint64(1) << np.uint64(40)   # counts beyond float32's integers
    meta = dict(nk=nk, nik=len(iki), ntr=ntr, nk_cum=nk_cum, nik_cum=np.asarray(nik_cum, np.uint32), iki=np.asarray(iki, np.uint32),
                ikmc=np.asarray(ikmc, np.uint8))
    return meta, counts, depths


def close(a, b, rtol=RTOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    fin = np.isfinite(a) & np.isfinite(b)
    same_kind = (np.isnan(a) == np.isnan(b)).all() and (np.isposinf(a) == np.isposinf(b)).all() and (np.isneginf(a) == np.isneginf(b)).all()
    return bool(same_kind and (np.abs(a[fin] - b[fin]) <= rtol * np.maximum(np.abs(a[fin]), np.abs(b[fin]))).all())


# ------------------------------------------------------------------- CPU ---
def test_oracle_against_the_formulas_in_float64():
    """The restatement against pred.h's formulas evaluated in float64 on the float32 raw matrix (a numerics check of the
    oracle itself; a few float32 roundings apart), and its handling of the loci the reference skips."""
    meta, counts, depths = make_cohort(1)
    raw = PO.raw_matrix(counts, depths)
    assert raw.dtype == np.float32 and raw.shape == (meta["nk"], len(depths))
    assert (raw == (counts.astype(np.float32) / depths[:, None]).T).all()
    cor, bias = PO.bias_correction(raw, meta)
    r64 = raw.astype(np.float64)
    nskip = 0
    for t in range(meta["ntr"]):
        si = int(meta["nk_cum"][t - 1]) if t else 0
        ei = int(meta["nk_cum"][t])
        isi = int(meta["nik_cum"][t - 1]) if t else 0
        iei = int(meta["nik_cum"][t])
        if si == ei or isi == iei:
            nskip += 1
            assert (bias[t] == 0).all() and (cor[si:ei] == raw[si:ei]).all()
            continue
        B = r64[meta["iki"][isi:iei]] / meta["ikmc"][isi:iei, None].astype(np.float64)
        b = B.mean(axis=0)
        b /= b.mean()
        assert close(bias[t], b, 1e-5)
        assert close(cor[si:ei], r64[si:ei] / b[None, :], 1e-5)
    assert nskip >= 5


def test_file_formats_round_trip(tmp_path):
    meta, counts, depths = make_cohort(2, ns=5, ntr=9)
    fn = str(tmp_path / "ikmer.meta")
    PO.write_ikmer_meta(fn, meta["nk"], meta["nk_cum"], meta["nik_cum"], meta["iki"], meta["ikmc"])
    m2 = PO.read_ikmer_meta(fn)
    for k in ("nk", "nik", "ntr"):
        assert m2[k] == meta[k]
    for k in ("nk_cum", "nik_cum", "iki", "ikmc"):
        assert (m2[k] == meta[k]).all()
    raw = PO.raw_matrix(counts, depths)
    b = PO.matrix_bytes(raw)
    assert struct.unpack("<II", b[:8]) == (5, meta["nk"]) and len(b) == 8 + 4 * 5 * meta["nk"]
    _, bias = PO.bias_correction(raw, meta)
    tsv = PO.bias_tsv(bias)
    rows = tsv.split("\n")
    assert len(rows) == 5 and all(len(r.split("\t")) == 9 for r in rows) and not tsv.endswith("\n")


def test_library_exports_the_pred_abi():
    hdr = open(os.path.join(bind.ROOT, "include", "dbtk_pred.h")).read()
    declared = {s for s in re.findall(r"\b(dbtk_pred_[a-z_0-9]+)\s*\(", hdr)}
    assert declared == set(pkg.EXPORTS_PRED), declared ^ set(pkg.EXPORTS_PRED)
    lib = pkg.Dbtk()
    for s in declared:
        assert hasattr(lib.L, s), s


# ------------------------------------------------------------------- GPU ---
@pytest.mark.gpu
@pytest.mark.parametrize("seed,ns,ntr", [(3, 37, 50), (4, 300, 20), (5, 1, 8), (6, 64, 400)])
def test_hip_pred_matches_oracle(seed, ns, ntr):
    meta, counts, depths = make_cohort(seed, ns=ns, ntr=ntr)
    lib = pkg.Dbtk()
    P = pkg.Pred(lib, ns, meta["nk_cum"], meta["nik_cum"], meta["iki"], meta["ikmc"], nk=meta["nk"])
    for s0 in range(0, ns, 23):                                   # ragged transfers
        P.load(s0, counts[s0:s0 + 23], depths[s0:s0 + 23])
    raw_o = PO.raw_matrix(counts, depths)
    raw = P.matrix()
    assert raw.tobytes() == raw_o.tobytes()                       # bit-exact
    P.correct()
    cor_o, bias_o = PO.bias_correction(raw_o, meta)
    assert close(P.bias(), bias_o) and close(P.matrix(), cor_o)
    assert (P.bias()[bias_o == 0] == 0).all()                     # skipped loci stay 0
    P.correct()                                                   # (idempotent in structure: runs again on the corrected matrix)
    cor2_o, _ = PO.bias_correction(cor_o, meta)
    assert close(P.matrix(), cor2_o, 1e-5)
    P.close()


@pytest.mark.gpu
def test_pred_command_line(tmp_path):
    """bin/danbing-tk-pred on files: the reference's arguments and output layouts (pred.cpp:14-84, pred.h:236-258)."""
    meta, counts, depths = make_cohort(7, ns=19, ntr=30)
    d = str(tmp_path)
    PO.write_ikmer_meta(os.path.join(d, "ikmer.meta"), meta["nk"], meta["nk_cum"], meta["nik_cum"], meta["iki"], meta["ikmc"])
    with open(os.path.join(d, "trkmers.meta.txt"), "w") as f:
        for s in range(19):
            fn = os.path.join(d, f"s{s}.trkmc.ar")
            with open(fn, "wb") as g:
                g.write(struct.pack("<Q", meta["nk"]) + counts[s].tobytes())
            f.write(f"{fn}\t{float(depths[s])!r}\n")
    exe = os.path.join(bind.ROOT, "danbing-tk_amd", "bin", "danbing-tk-pred")
    r = subprocess.run([exe, os.path.join(d, "trkmers.meta.txt"), os.path.join(d, "ikmer.meta"), os.path.join(d, "raw.gt"), os.path.join(d, "cor.gt"),
                        os.path.join(d, "bias.tsv")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw_o = PO.raw_matrix(counts, depths)
    cor_o, bias_o = PO.bias_correction(raw_o, meta)
    assert open(os.path.join(d, "raw.gt"), "rb").read() == PO.matrix_bytes(raw_o)
    cb = open(os.path.join(d, "cor.gt"), "rb").read()
    assert cb[:8] == PO.matrix_bytes(cor_o)[:8]
    assert close(np.frombuffer(cb[8:], np.float32), cor_o.ravel())
    tsv = open(os.path.join(d, "bias.tsv")).read()
    got = np.array([[float(x) for x in row.split("\t")] for row in tsv.split("\n")])
    assert got.shape == (19, 30) and close(got, bias_o.T, 2e-5) and not tsv.endswith("\n")   # (%g keeps 6 digits)
    # a count file of another RPGG build: the reference asserts (exit 134)
    with open(os.path.join(d, "s0.trkmc.ar"), "wb") as g:
        g.write(struct.pack("<Q", meta["nk"] + 1) + counts[0].tobytes() + b"\0" * 8)
    r = subprocess.run([exe, os.path.join(d, "trkmers.meta.txt"), os.path.join(d, "ikmer.meta"), os.path.join(d, "raw.gt"), os.path.join(d, "cor.gt"),
                        os.path.join(d, "bias.tsv")], capture_output=True, text=True)
    assert r.returncode == 134 and "nk" in r.stderr
    assert subprocess.run([exe], capture_output=True, text=True).returncode == 0   # usage, like the reference
