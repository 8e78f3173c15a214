#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/<name>/.

Inputs are made by tests/synth.py (seeded); the RPGG files come from the
reference's own builders (oracle/_ref/fa2kmers, ktools serialize); the expected
outputs are what the reference binary oracle/_ref/danbing-tk writes for them at
-p 1.  A fixture is data only: RPGG files, a read file, and the reference's
output bytes (+ the command line in cmd.txt).  Needs /root/reference compiled
into oracle/_ref (make -C oracle ref); run from the repo root:

    python tests/golden/make_golden.py
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402

SETS = {
    # name: (loci kwargs, k, read kwargs, fastq, extra flags, qc mask)
    "g1_k21": (dict(nloci=6, nhap=3, flank=500, seed=101, tr_max=600), 21,
               dict(npairs=350, seed=201, sub=0.004, indel=0.001, nrate=0.001, background=0.1, short=0.03), False,
               ["-cth", "45", "-kf", "4", "1"], None),
    "g2_shared_fq": (dict(nloci=8, nhap=2, flank=500, seed=102, shared_frac=0.9, tr_max=500), 21,
                     dict(npairs=350, seed=202, sub=0.003, chimeric=0.5, lower=0.03, with_qual=True), True, [], None),
    "g3_k25_qc": (dict(nloci=6, nhap=3, flank=500, seed=103, shared_frac=0.3, tr_max=600), 25,
                  dict(npairs=300, seed=203, sub=0.006, chimeric=0.2, background=0.1), False,
                  ["-cth", "40", "-c", "30"], "101101"),
}


def bait_bubble_set(dtk):
    """g4: -bu and -b.  Reads come from a haplotype that is NOT in the RPGG (a few substitutions per
    locus), so the same novel (k+1)-mers recur and pass dumpBubbles' count >= 5 threshold."""
    import numpy as np
    name = "g4_bait_bubbles"
    d = os.path.join(HERE, name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    loci = synth.make_loci(nloci=6, nhap=2, flank=500, seed=104, tr_max=500)
    synth.build_rpgg_with_reference(loci, d, k=21)
    rng = np.random.default_rng(7)
    mut = synth.Loci(flank=loci.flank, haps=[[s.copy() for s in loci.haps[0]]], nloci=loci.nloci, nhap=1)
    for s in mut.haps[0]:
        for p in rng.integers(loci.flank + 5, len(s) - loci.flank - 5, 3):
            s[p] = synth.BASES[(int(np.where(synth.BASES == s[p])[0][0]) + 1) % 4]
    reads = synth.sim_reads(mut, npairs=900, seed=204, sub=0.002, with_qual=True)
    synth.make_bait_db(loci, reads, d, per_locus=25)
    synth.write_fasta(reads, os.path.join(d, "reads.fq"), fastq=True)
    fa = synth.Reads(seqs=reads.seqs, titles=reads.titles)
    synth.write_fasta(fa, os.path.join(d, "reads.fa"))
    for f in os.listdir(d):
        if f.startswith("pan.") and f.split(".", 1)[1] not in ("tr.kmers", "kmers.dbi", "fl.kdb", "tre.kdb", "bt.kmdb"):
            os.remove(os.path.join(d, f))
        elif f.startswith("h") and f.endswith(".fa"):
            os.remove(os.path.join(d, f))
    cmds = []
    for tag, args in (("refbu", ["-bu", "-k", "21", "-cth", "45", "-fa", "reads.fa", "-qs", "pan", "-p", "1"]),
                      ("refbt", ["-b", "-k", "21", "-cth", "45", "-fa", "reads.fa", "-qs", "pan", "-p", "1"]),
                      ("refbq", ["-bu", "-b", "-qth", "25", "-k", "21", "-cth", "30", "-fq", "reads.fq", "-qs", "pan", "-p", "1"]),
                      # -tb: OUT.btk.kmdb (with -b: FASTA and FASTQ; without -b: the empty tracker)
                      ("reftb", ["-b", "-tb", "-k", "21", "-cth", "45", "-fa", "reads.fa", "-qs", "pan", "-p", "1"]),
                      ("reftq", ["-tb", "-b", "-qth", "25", "-k", "21", "-cth", "30", "-fq", "reads.fq", "-qs", "pan", "-p", "1"]),
                      ("reftn", ["-tb", "-k", "21", "-cth", "45", "-ka", "-fa", "reads.fa", "-qs", "pan", "-p", "1"])):
        cmd = ["danbing-tk"] + args + ["-o", tag]
        with open(os.path.join(d, tag + ".kam.txt"), "wb") as so, open(os.path.join(d, tag + ".stderr.txt"), "wb") as se:
            subprocess.run([dtk] + cmd[1:], cwd=d, check=True, stdout=so, stderr=se)
        keep = [l for l in open(os.path.join(d, tag + ".stderr.txt"), errors="replace") if l[:1].isdigit() and " reads " in l]
        open(os.path.join(d, tag + ".totals.txt"), "w").writelines(keep)
        os.remove(os.path.join(d, tag + ".stderr.txt"))
        cmds.append(" ".join(cmd) + f" > {tag}.kam.txt")
    open(os.path.join(d, "cmd.txt"), "w").write("\n".join(cmds) + "\n")
    sz = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
    print(name, sorted(os.listdir(d)), f"{sz / 1024:.0f} KiB")
    for tag in ("refbu", "refbq"):
        import numpy as np
        a = np.fromfile(os.path.join(d, tag + ".bub.kmdb"), np.uint64)
        print("  ", tag, "bubble k-mers kept:", int(a[1 + int(a[0])]))


def main():
    dtk = synth.ref_tool("danbing-tk")
    bait_bubble_set(dtk)
    for name, (lk, k, rk, fastq, flags, qc) in SETS.items():
        d = os.path.join(HERE, name)
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        loci = synth.make_loci(**lk)
        synth.build_rpgg_with_reference(loci, d, k=k)
        for f in os.listdir(d):  # keep only what the aligner loads
            if not f.startswith("pan.") or f.split(".", 1)[1] not in ("tr.kmers", "kmers.dbi", "fl.kdb", "tre.kdb"):
                os.remove(os.path.join(d, f))
        reads = synth.sim_reads(loci, **rk)
        rfile = "reads.fq" if fastq else "reads.fa"
        synth.write_fasta(reads, os.path.join(d, rfile), fastq=fastq)
        base = ["-k", str(k)]
        if qc:
            with open(os.path.join(d, "qc.txt"), "w") as f:
                f.write(qc)
            base += ["-qc", "qc.txt"]
        base += flags + ["-fq" if fastq else "-fa", rfile, "-qs", "pan", "-p", "1"]
        cmds = []
        # 1) counts + summary + kam on stdout
        cmd = ["danbing-tk"] + base + ["-o", "ref"]
        with open(os.path.join(d, "ref.kam.txt"), "wb") as so, open(os.path.join(d, "ref.stderr.txt"), "wb") as se:
            subprocess.run([dtk] + cmd[1:], cwd=d, check=True, stdout=so, stderr=se)
        cmds.append(" ".join(cmd) + " > ref.kam.txt")
        # 2) -on: named text output (k-mer order pins the unordered_map iteration order)
        cmd = ["danbing-tk", "-ka"] + base + ["-on", "refon"]
        subprocess.run([dtk] + cmd[1:], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        os.remove(os.path.join(d, "refon.trkmc.ar"))
        cmds.append(" ".join(cmd))
        # 3) -e 2: extracted pairs with the assigned locus appended to the title
        cmd = ["danbing-tk", "-e", "2"] + base
        with open(os.path.join(d, "ref.extract.txt"), "wb") as so:
            subprocess.run([dtk] + cmd[1:], cwd=d, check=True, stdout=so, stderr=subprocess.DEVNULL)
        cmds.append(" ".join(cmd) + " > ref.extract.txt")
        with open(os.path.join(d, "cmd.txt"), "w") as f:
            f.write("\n".join(cmds) + "\n")
        # stderr has timings: keep only the totals block
        keep = [l for l in open(os.path.join(d, "ref.stderr.txt"), errors="replace")
                if l[:1].isdigit() and " reads " in l]
        with open(os.path.join(d, "ref.totals.txt"), "w") as f:
            f.writelines(keep)
        os.remove(os.path.join(d, "ref.stderr.txt"))
        sz = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
        print(name, sorted(os.listdir(d)), f"{sz / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
