#!/usr/bin/env python3
"""Generates tests/golden/g5_walk_k25: the v1.3 threading path (BASELINE configs 4 and 5: k = 25, -gc 85 3, -a / -ae).

The mounted reference keeps the threading call sites of `danbing-tk` in comments (src/aQueryFasta_thread.cpp:2072-2088,
2189-2194, 2232-2248), so its binary cannot produce these outputs.  The expected bytes therefore come from the
reference's own FUNCTIONS (isThreadFeasible, noncaVec2CaUmap, writeAlignments, ... compiled from /root/reference into
oracle/_ref/libdbtk_refharness.so) driven through exactly those commented lines by oracle/ref_harness.cpp:ref_align_v13.
A fixture is data only: RPGG files incl. pan.graph.kmers (the reference's fa2kmers -g + ktools serialize), a read
file, the expected stdout (alignment records), OUT.trkmc.ar / OUT.tr.summary.txt and the totals.  Run from the repo root
where /root/reference exists:

    python tests/golden/make_golden_walk.py
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import bind  # noqa: E402
import refio  # noqa: E402
import synth  # noqa: E402

abi = bind.abi
NAME, K = "g5_walk_k25", 25
TOTALS = ("{} reads processed in total.\n{} reads removed by subsampled kmer-filter.\n{} reads removed by kmer-filter.\n"
          "{} reads removed by bait locus.\n{} reads removed by qual filter.\n{} reads removed during locus assignment.\n"
          "{} reads removed by QC filter.\n{} reads entered threading step.\n{} reads passsed threading.\n"
          "{} reads assigned to TR region.\n")


def main():
    d = os.path.join(HERE, NAME)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    loci = synth.make_loci(nloci=4, nhap=2, flank=300, seed=105, tr_max=400, motif_min=3, motif_max=25)
    synth.build_rpgg_with_reference(loci, d, k=K)
    reads = synth.sim_reads(loci, npairs=260, seed=205, sub=0.012, indel=0.004, nrate=0.002, chimeric=0.1, background=0.15, frag=(200, 320))
    synth.write_fasta(reads, os.path.join(d, "reads.fa"))
    for f in os.listdir(d):
        if f.startswith("pan.") and f.split(".", 1)[1] not in ("tr.kmers", "kmers.dbi", "fl.kdb", "tre.kdb", "graph.kmers"):
            os.remove(os.path.join(d, f))
        elif f.startswith("h") and f.endswith(".fa"):
            os.remove(os.path.join(d, f))
    H, O = bind.RefHarness(), bind.Oracle()
    pref = os.path.join(d, "pan")
    h = H.load(pref)
    H.load_graph(h, pref + ".graph.kmers")
    tr = synth.read_rpgg_files(pref)
    cmds = []
    for tag, flags, aln, tcth, maxc in (("refae", ["-gc", "85", "3", "-ae"], 2, 85, 3), ("refa", ["-gc", "100", "-a"], 1, 100, 4),
                                        ("refg", ["-gc", "85", "3"], 0, 85, 3)):
        p = abi.default_params(ksize=K, cthreshold=45, threading=2, aln=aln, okam=0, thread_cth=tcth, correction=1, maxncorrection=maxc)
        # the reads as the reference's reader hands them to the hot loop (pairing + minimum length, AQ.cpp:1918-1976)
        rd = refio.read_pairs(os.path.join(d, "reads.fa"), False, p.cthreshold + p.ksize - 1)
        seq, off = rd.packed()
        r = H.align_v13(h, p, seq, off, rd.titles)
        open(os.path.join(d, tag + ".aln.txt"), "w").write(r["aln"])
        out, i = [], 0
        for n in tr["tr_cnt"]:  # OUT.trkmc.ar: per locus in unordered_map iteration order (dumpTRKmers, AQ.h:968-973)
            n = int(n)
            out.append(r["counts_file"][i:i + n][H.umap_order(tr["tr_ks"][i:i + n]).astype(np.int64)])
            i += n
        out = np.concatenate(out)
        with open(os.path.join(d, tag + ".trkmc.ar"), "wb") as f:
            f.write(np.uint64(len(out)).tobytes() + out.tobytes())
        with open(os.path.join(d, tag + ".tr.summary.txt"), "w") as f:  # nmapread / kmc are not touched on the threading path
            f.write("0\t0\n" * loci.nloci)
        c = r["counters"]
        nreads = 2 * rd.npairs
        open(os.path.join(d, tag + ".totals.txt"), "w").write(TOTALS.format(
            nreads, c[abi.C_SUBFILTERED], c[abi.C_KMERFILTERED], 0, 0, c[abi.C_LOCUSFILTERED], c[abi.C_QCFILTERED], c[abi.C_THREADING],
            c[abi.C_FEASIBLE], 0))
        cmds.append("danbing-tk --v13-threading " + " ".join(flags) + f" -ka -k {K} -cth 45 -fa reads.fa -qs pan -p 1 -o {tag} > {tag}.aln.txt")
        print(tag, r["nres"], "walked pairs,", r["aln"].count("\n"), "records,", int(out.sum()), "counts")
    open(os.path.join(d, "cmd.txt"), "w").write("\n".join(cmds) + "\n")
    sz = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
    print(NAME, sorted(os.listdir(d)), f"{sz / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
