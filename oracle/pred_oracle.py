"""TEST INFRASTRUCTURE — the CPU restatement of `danbing-tk-pred` (the reference's src/pred.h), used by tests/ only.

PARITY UNPINNED: the reference's pred.cpp needs Eigen and the submodule directory is empty in /root/reference
(.gitmodules:1-3), so the reference cannot be compiled here and this restatement was not checked against a run of it.
It follows the source statement by statement instead:

  load_eachBinGT   pred.h:166-186   u64 counts of every sample -> float32 (Eigen cast<float>())
  norm_rd          pred.h:204-209   gt(sample, kmer) = count / read depth (float32 division)
  bias_correction  pred.h:212-233   per locus with k-mers and invariant k-mers:
                                      B(s, j) = gt(s, iki[j]) / ikmc[j];  bias(s) = mean_j B(s, j);  bias /= mean_s bias(s)
                                      gt(s, si:ei) /= bias(s);  Bias(s, tri) = bias(s)
  save_matrix      pred.h:236-258   u32 rows, u32 cols, column-major float32 | TSV with the stream's default precision

All arithmetic is numpy float32.  The row sums are accumulated in k-mer order (one float32 add per invariant k-mer, the
order of Eigen's scalar reduction); the mean over the samples is numpy's pairwise float32 sum (Eigen uses packet sums:
the last bits may differ, which is why the GPU tests compare the corrected values with a relative tolerance).
"""
import struct

import numpy as np


def read_ikmer_meta(fn):
    """read_ikmer, pred.h:64-126."""
    with open(fn, "rb") as f:
        nk, nik, ntr = struct.unpack("<QQQ", f.read(24))
        nk_cum = np.frombuffer(f.read(4 * ntr), "<u4").copy()
        nik_cum = np.frombuffer(f.read(4 * ntr), "<u4").copy()
        rec = np.frombuffer(f.read(5 * nik), dtype=[("ki", "<u4"), ("kc", "u1")])
    return dict(nk=nk, nik=nik, ntr=ntr, nk_cum=nk_cum, nik_cum=nik_cum, iki=rec["ki"].copy(), ikmc=rec["kc"].copy())


def write_ikmer_meta(fn, nk, nk_cum, nik_cum, iki, ikmc):
    with open(fn, "wb") as f:
        f.write(struct.pack("<QQQ", nk, len(iki), len(nk_cum)))
        f.write(np.asarray(nk_cum, "<u4").tobytes())
        f.write(np.asarray(nik_cum, "<u4").tobytes())
        rec = np.zeros(len(iki), dtype=[("ki", "<u4"), ("kc", "u1")])
        rec["ki"] = iki
        rec["kc"] = ikmc
        f.write(rec.tobytes())


def raw_matrix(counts, depths):
    """counts[ns][nk] (u64), depths[ns] -> gt[nk][ns] float32 (= the ns x nk column-major matrix save_matrix writes)."""
    c = np.asarray(counts, np.uint64).astype(np.float32)            # cast<float>()
    d = np.asarray(depths, np.float32)
    return np.ascontiguousarray((c / d[:, None]).T)                  # rowwise() / rd, transposed


def bias_correction(gt, meta):
    """gt[nk][ns] float32 (modified copy returned) -> (corrected gt, Bias[ntr][ns]); skipped loci keep Bias = 0."""
    gt = gt.copy()
    nk, ns = gt.shape
    ntr = int(meta["ntr"])
    bias_all = np.zeros((ntr, ns), np.float32)
    for tri in range(ntr):
        si = int(meta["nk_cum"][tri - 1]) if tri else 0
        ei = int(meta["nk_cum"][tri])
        isi = int(meta["nik_cum"][tri - 1]) if tri else 0
        iei = int(meta["nik_cum"][tri])
        if si == ei or isi == iei:
            continue
        acc = np.zeros(ns, np.float32)
        for j in range(isi, iei):                                    # B.rowwise().mean(): sum in k-mer order, then / n
            acc = acc + gt[int(meta["iki"][j])] / np.float32(meta["ikmc"][j])
        bias = acc / np.float32(iei - isi)
        with np.errstate(divide="ignore", invalid="ignore"):
            bias = bias / (bias.sum(dtype=np.float32) / np.float32(ns))  # bias /= bias.mean()
            gt[si:ei] = gt[si:ei] / bias[None, :]
        bias_all[tri] = bias
    return gt, bias_all


def matrix_bytes(gt):
    """save_matrix (binary): low 4 bytes of rows (ns) and columns (nk), then the column-major data."""
    nk, ns = gt.shape
    return struct.pack("<II", ns & 0xFFFFFFFF, nk & 0xFFFFFFFF) + np.ascontiguousarray(gt, np.float32).tobytes()


def bias_tsv(bias_all):
    """save_matrix (tsv_format): rows = samples, columns = loci, '%g', no final newline."""
    ntr, ns = bias_all.shape
    return "\n".join("\t".join("%g" % float(bias_all[t, s]) for t in range(ntr)) for s in range(ns))
