"""The reference-compatible command line (danbing-tk_amd/bin/danbing-tk): argv
behaviour on CPU, and on a GPU the same commands as tests/golden/*/cmd.txt must
reproduce the reference binary's stdout and output files byte for byte."""
import os
import shutil
import signal
import subprocess

import pytest

import bind
import cases
import synth
from test_oracle import GOLD

ROOT = bind.ROOT
CLI = os.environ.get("DBTK_CLI", os.path.join(ROOT, "danbing-tk_amd", "bin", "danbing-tk"))  # (make asan points it at the sanitized build)
GOLDEN = cases.GOLDEN


def run(args, cwd=None, **kw):
    return subprocess.run([CLI] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)


def test_usage_and_invalid_option():
    r = run([])
    assert r.returncode == 0 and b"Usage: danbing-tk" in r.stderr and r.stdout == b""   # AQ.cpp:2288-2333
    r = run(["-nope"])
    assert r.returncode == -signal.SIGABRT and b"invalid option: -nope" in r.stderr       # AQ.cpp:2424-2427 (`throw;`)


def test_missing_files_abort(tmp_path):
    r = run(["-qs", str(tmp_path / "nothing")])
    assert r.returncode == -signal.SIGABRT                                               # assert(trFile), AQ.cpp:2391
    d = os.path.join(GOLDEN, "g1_k21")
    r = run(["-fa", str(tmp_path / "no.fa"), "-qs", "pan"], cwd=d)
    assert r.returncode == -signal.SIGABRT                                               # assert(fastxFile), AQ.cpp:2412


def test_output_file_is_truncated_at_parse_time(tmp_path):
    out = tmp_path / "o"
    (tmp_path / "o.trkmc.ar").write_bytes(b"stale")
    run(["-o", str(out), "-nope"])
    assert (tmp_path / "o.trkmc.ar").read_bytes() == b""                                 # AQ.cpp:2417


def test_refuses_to_run_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    d = os.path.join(GOLDEN, "g1_k21")
    r = run(["-k", "21", "-fa", "reads.fa", "-qs", "pan", "-o", str(tmp_path / "o")], cwd=d)
    assert r.returncode == -signal.SIGABRT and b"no CPU execution path" in r.stderr


def golden_cmds(name):
    d = os.path.join(GOLDEN, name)
    return d, [l.strip() for l in open(os.path.join(d, "cmd.txt")) if l.strip()]


GOLDEN_SETS = sorted(d for d in os.listdir(GOLDEN) if os.path.isfile(os.path.join(GOLDEN, d, "cmd.txt")))


@pytest.mark.gpu
@pytest.mark.parametrize("name", GOLDEN_SETS)
def test_cli_reproduces_reference_binary(name, tmp_path):
    """Every command of the set's cmd.txt (the exact commands the reference binary was run with): same stdout,
    and every output file the reference wrote for that command's prefix is reproduced byte for byte
    (.trkmc.ar, .tr.summary.txt, -on .tr.kmers, .bub.kmdb), as is the totals block of stderr."""
    d, cmds = golden_cmds(name)
    w = str(tmp_path / "w")
    os.makedirs(w)
    ref_outputs = set()
    for line in cmds:
        a = line.split(" > ")[0].split()
        tag = a[a.index("-o") + 1] if "-o" in a else (a[a.index("-on") + 1] if "-on" in a else None)
        if tag:
            ref_outputs.add(tag)
    for f in os.listdir(d):  # inputs only
        if f.split(".")[0] not in ref_outputs:
            shutil.copy(os.path.join(d, f), w)
    for line in cmds:
        parts = line.split(" > ")
        args = parts[0].split()[1:]
        r = run(args, cwd=w)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        if len(parts) == 2:
            assert r.stdout == open(os.path.join(d, parts[1]), "rb").read(), f"stdout of `{line}` differs"
        tag = args[args.index("-o") + 1] if "-o" in args else (args[args.index("-on") + 1] if "-on" in args else None)
        if not tag:
            continue
        for f in sorted(os.listdir(d)):
            if not f.startswith(tag + ".") or f.endswith((".kam.txt", ".extract.txt", ".aln.txt")):
                continue
            if f.endswith(".totals.txt"):
                mine = [l + "\n" for l in r.stderr.decode().split("\n") if l[:1].isdigit() and " reads " in l]
                assert mine == open(os.path.join(d, f)).readlines(), f
            else:
                assert open(os.path.join(w, f), "rb").read() == open(os.path.join(d, f), "rb").read(), f
        if "-on" in args:
            assert os.path.getsize(os.path.join(w, tag + ".trkmc.ar")) == 0  # -on leaves the truncated .trkmc.ar empty


@pytest.mark.gpu
def test_cli_two_gpus_on_one_device(tmp_path):
    """`--gpus 2` on a one-GPU box (DBTK_DEVICE_MAP=0,0): two contexts, the input cut into two ranges each with its own device
    reader, the cross-range pairing of what the ranges leave over, the merge of the two contexts' accumulators on the host — the
    output files are the single-GPU run's (= the reference binary's) byte for byte, stdout the same lines."""
    name = GOLDEN_SETS[0]
    d, cmds = golden_cmds(name)
    w = str(tmp_path / "w")
    os.makedirs(w)
    for f in os.listdir(d):
        shutil.copy(os.path.join(d, f), w)
    ran = 0
    for line in cmds:
        parts = line.split(" > ")
        args = parts[0].split()[1:]
        tag = args[args.index("-o") + 1] if "-o" in args else None
        if not tag or "-tb" in args:
            continue
        for extra_env in (dict(DBTK_SHARD_MIN="0"), dict(DBTK_SHARD_MIN="0", DBTK_INGEST_CHUNK="20000")):
            r = run(args + ["--gpus", "2"], cwd=w, env=dict(os.environ, DBTK_DEVICE_MAP="0,0", **extra_env))
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            if len(parts) == 2:
                assert sorted(r.stdout.splitlines()) == sorted(open(os.path.join(d, parts[1]), "rb").read().splitlines())
            for f in sorted(os.listdir(d)):
                if f.startswith(tag + ".") and f.endswith((".trkmc.ar", ".tr.summary.txt")):
                    assert open(os.path.join(w, f), "rb").read() == open(os.path.join(d, f), "rb").read(), f
        ran += 1
    assert ran > 0


@pytest.mark.gpu
def test_cli_index_sidecar(tmp_path):
    """--write-idx-cache leaves PREF.dbtk.idx next to the RPGG; the next run loads the GPU-layout index from it (the `tables:` line
    says so) and prints the same bytes as the golden run — which never sees a sidecar, and writes none."""
    name = GOLDEN_SETS[0]
    d, cmds = golden_cmds(name)
    w = str(tmp_path / "w")
    os.makedirs(w)
    for f in os.listdir(d):
        shutil.copy(os.path.join(d, f), w)
    line = next(l for l in cmds if " > " in l)
    args, want = line.split(" > ")[0].split()[1:], open(os.path.join(d, line.split(" > ")[1]), "rb").read()
    r0 = run(args, cwd=w)
    assert r0.returncode == 0 and r0.stdout == want and not [f for f in os.listdir(w) if f.endswith(".dbtk.idx")]
    r1 = run(args + ["--write-idx-cache"], cwd=w)
    side = [f for f in os.listdir(w) if f.endswith(".dbtk.idx")]
    assert r1.returncode == 0 and r1.stdout == want and len(side) == 1, r1.stderr.decode()[-1500:]
    assert b"tables:" in r1.stderr and b"from the sidecar" not in r1.stderr
    r2 = run(args, cwd=w)
    assert r2.returncode == 0 and r2.stdout == want and b"images from the sidecar" in r2.stderr, r2.stderr.decode()[-1500:]


@pytest.mark.gpu
@pytest.mark.skipif(not synth.have_ref(), reason="oracle/_ref not built")
def test_cli_vs_reference_binary_live(tmp_path):
    """A fresh mid-size case, reference binary and this CLI run side by side (FASTQ, mates not adjacent)."""
    loci = synth.make_loci(nloci=60, nhap=3, flank=500, seed=77, shared_frac=0.4)
    d = str(tmp_path)
    synth.build_rpgg_with_reference(loci, d, k=21)
    reads = synth.sim_reads(loci, npairs=6000, seed=78, sub=0.006, indel=0.001, nrate=0.002, chimeric=0.3, background=0.2, short=0.02,
                            with_qual=True)
    # interleave: all /2 mates first, then all /1 mates (the reader pairs by title across the whole file)
    with open(os.path.join(d, "r.fq"), "wb") as f:
        for which, tag in ((1, b"/2"), (0, b"/1")):
            for p in range(reads.npairs):
                f.write(b"@" + reads.titles[p].encode() + tag + b"\n" + reads.seqs[2 * p + which] + b"\n+\n" + reads.quals[2 * p + which] + b"\n")
    # mates adjacent (the parser's held-record path), many small batches, with singletons, a triple and a late straggler mixed in
    with open(os.path.join(d, "r.fa"), "wb") as f:
        for p in range(reads.npairs):
            t = b">" + reads.titles[p].encode()
            if p % 97 == 5:
                f.write(t + b"/1\n" + reads.seqs[2 * p] + b"\n")                      # singleton: stays parked
                continue
            f.write(t + b"/1\n" + reads.seqs[2 * p] + b"\n" + t + b"/2\n" + reads.seqs[2 * p + 1] + b"\n")
            if p % 131 == 7:
                f.write(t + b"/1\n" + reads.seqs[2 * p] + b"\n")                      # third record of a title: parked again
            if p % 211 == 9 and p > 300:
                q = p - 300 + (5 - (p - 300) % 97) % 97                                  # a singleton seen long ago finds its mate
                if q % 97 == 5 and q < p:
                    f.write(b">" + reads.titles[q].encode() + b"/2\n" + reads.seqs[2 * q + 1] + b"\n")
    # interleaved FASTQ (the ingest pairs such blocks in its parallel stage), with an orphan a third of the way in whose mate
    # arrives at two thirds: in between something is parked and every block goes record by record
    with open(os.path.join(d, "r2.fq"), "wb") as f:
        rec = lambda p, w: b"@" + reads.titles[p].encode() + (b"/1", b"/2")[w] + b"\n" + reads.seqs[2 * p + w] + b"\n+\n" + reads.quals[2 * p + w] + b"\n"
        for p in range(reads.npairs):
            if p == reads.npairs // 3:
                f.write(rec(0, 0))
            if p == 2 * reads.npairs // 3:
                f.write(rec(0, 1))
            if p:
                f.write(rec(p, 0) + rec(p, 1))
    for flags in (["-cth", "45"], ["-cth", "20", "-kf", "8", "2", "-r", "0.01"], ["-e", "1"], ["-gc", "85", "3"], ["-cth", "30", "-r", "0.004", "FA"],
                  ["-cth", "30", "-r", "0.004", "FA", "BLK"], ["-cth", "45", "-r", "0.003", "FQ2", "BLK"], ["-cth", "45", "FQ2"]):
        outs = []
        env = dict(os.environ)
        if flags[-1] == "BLK":  # many small input blocks: pre-paired and record-by-record blocks alternate, batches end inside blocks
            flags = flags[:-1]
            env["DBTK_INGEST_BLOCK"] = "30000"
        fa = flags[-1] == "FA"
        fq2 = flags[-1] == "FQ2"
        if fa or fq2:
            flags = flags[:-1]
        for exe, tag in ((synth.ref_tool("danbing-tk"), "ref"), (CLI, "hip")):
            a = ["-k", "21"] + flags + (["-fa", "r.fa"] if fa else ["-fq", "r2.fq" if fq2 else "r.fq"]) + ["-qs", "pan", "-o", tag]
            r = subprocess.run([exe] + a, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            assert r.returncode == 0, r.stderr.decode()[-1000:]
            outs.append(r.stdout)
        assert outs[0] == outs[1], f"stdout differs for {flags}"
        if "-e" not in flags:
            for ext in (".trkmc.ar", ".tr.summary.txt"):
                assert open(os.path.join(d, "ref" + ext), "rb").read() == open(os.path.join(d, "hip" + ext), "rb").read(), (flags, ext)
    # -t N: PREF.tr.trimN.kmers in place of PREF.tr.kmers (AQ.cpp:2352, 2389): a trimmed file (every fifth TR k-mer dropped, one locus
    # emptied) -> fewer counters, k-mers of the index that are in no set any more; the reference binary on the same files
    with open(os.path.join(d, "pan.tr.kmers")) as f, open(os.path.join(d, "pan.tr.trim7.kmers"), "w") as g:
        nk, locus = 0, -1
        for line in f:
            if line[0] == ">":
                locus += 1
                g.write(line)
            else:
                nk += 1
                if nk % 5 and locus != 3:
                    g.write(line)
    outs = []
    for exe, tag in ((synth.ref_tool("danbing-tk"), "reft"), (CLI, "hipt")):
        r = subprocess.run([exe, "-k", "21", "-cth", "45", "-t", "7", "-fa", "r.fa", "-qs", "pan", "-o", tag], cwd=d, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-1000:]
        outs.append(r.stdout)
    assert outs[0] == outs[1] and len(outs[0]) > 0
    for ext in (".trkmc.ar", ".tr.summary.txt"):
        assert open(os.path.join(d, "reft" + ext), "rb").read() == open(os.path.join(d, "hipt" + ext), "rb").read(), ("-t 7", ext)
    assert os.path.getsize(os.path.join(d, "hipt.trkmc.ar")) < os.path.getsize(os.path.join(d, "hip.trkmc.ar"))
    # the pipeline form of the reference's README (`samtools fasta ... | danbing-tk -fa /dev/stdin`): a pipe, not a seekable file
    with open(os.path.join(d, "r.fa"), "rb") as f:
        r = subprocess.run([CLI, "-k", "21", "-cth", "30", "-fa", "/dev/stdin", "-qs", "pan", "-o", "pipe"], cwd=d, stdin=f,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    r2 = subprocess.run([CLI, "-k", "21", "-cth", "30", "-fa", "r.fa", "-qs", "pan", "-o", "file"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r2.returncode == 0 and r.stdout == r2.stdout
    assert open(os.path.join(d, "pipe.trkmc.ar"), "rb").read() == open(os.path.join(d, "file.trkmc.ar"), "rb").read()
    # four ingest pipelines with a context each on the one GPU (--ingest-shards): the same counts and the same kam lines (as
    # a set: the ranges' batches leave in completion order, like the reference's with -p > 1)
    r3 = subprocess.run([CLI, "-k", "21", "-cth", "30", "--ingest-shards", "4", "-fa", "r.fa", "-qs", "pan", "-o", "sh4"], cwd=d,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, DBTK_SHARD_MIN="0"))
    assert r3.returncode == 0, r3.stderr.decode()[-1000:]
    assert b"cross-range pairing" in r3.stderr
    assert open(os.path.join(d, "sh4.trkmc.ar"), "rb").read() == open(os.path.join(d, "file.trkmc.ar"), "rb").read()
    assert open(os.path.join(d, "sh4.tr.summary.txt"), "rb").read() == open(os.path.join(d, "file.tr.summary.txt"), "rb").read()
    assert sorted(r3.stdout.splitlines()) == sorted(r2.stdout.splitlines())


@pytest.mark.gpu
@pytest.mark.skipif(not synth.have_ref(), reason="oracle/_ref not built")
def test_cli_accepts_v13_rpgg(tmp_path):
    """`-qs PREF` with only the v1.3 files (PREF.kmerDBi.umap/.vv, PREF.ntr.kmers, PREF.tr.kmers — the reference's own fixture):
    same stdout and output files as with the HEAD files that `ktools serialize` makes from the same k-mer sets, and as the
    reference binary on those."""
    import shutil
    import numpy as np
    leg = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "legacy_v13")
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "v13"))
    os.makedirs(os.path.join(d, "head"))
    for f in ("pan.tr.kmers", "pan.ntr.kmers", "pan.kmerDBi.umap", "pan.kmerDBi.vv"):
        shutil.copy(os.path.join(leg, f), os.path.join(d, "v13", f))
    shutil.copy(os.path.join(leg, "pan.tr.kmers"), os.path.join(d, "head", "pan.tr.kmers"))
    shutil.copy(os.path.join(leg, "pan.ntr.kmers"), os.path.join(d, "head", "pan.fl.kmers"))
    open(os.path.join(d, "head", "pan.tre.kmers"), "w").write(">0\n")
    assert subprocess.run([synth.ref_tool("ktools"), "serialize", "pan"], cwd=os.path.join(d, "head")).returncode == 0
    ks = [int(l.split()[0]) for f in ("pan.tr.kmers", "pan.ntr.kmers") for l in open(os.path.join(leg, f)) if l[0] != ">"]
    rng = np.random.default_rng(8)

    def dec(km):
        return "".join("ACGT"[(km >> (2 * (20 - i))) & 3] for i in range(21))
    with open(os.path.join(d, "r.fa"), "w") as f:
        for p in range(400):
            for m in (1, 2):
                f.write(f">r{p}/{m}\n" + "".join(dec(ks[rng.integers(len(ks))]) for _ in range(8))[:150] + "\n")
    outs = {}
    for tag, exe, sub in (("v13", CLI, "v13"), ("head", CLI, "head"), ("ref", synth.ref_tool("danbing-tk"), "head")):
        r = subprocess.run([exe, "-k", "21", "-cth", "3", "-fa", os.path.join(d, "r.fa"), "-qs", "pan", "-o", tag], cwd=os.path.join(d, sub),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-1000:]
        outs[tag] = (r.stdout, open(os.path.join(d, sub, tag + ".trkmc.ar"), "rb").read(), open(os.path.join(d, sub, tag + ".tr.summary.txt"), "rb").read())
    assert outs["v13"] == outs["head"] == outs["ref"]
    assert len(outs["ref"][0]) > 0


@pytest.mark.skipif(not synth.have_ref(), reason="oracle/_ref not built")
def test_ktools_serialize_binary(tmp_path):
    """bin/ktools serialize PREF == the reference's ktools serialize PREF, byte for byte (host only: runs without a GPU)."""
    import shutil
    loci = synth.make_loci(nloci=7, nhap=3, flank=300, seed=23, shared_frac=0.5)
    d = str(tmp_path / "ref")
    os.makedirs(d)
    pref = synth.build_rpgg_with_reference(loci, d, k=21)
    d2 = str(tmp_path / "mine")
    os.makedirs(d2)
    for ext in (".tr.kmers", ".fl.kmers", ".tre.kmers"):
        shutil.copy(pref + ext, os.path.join(d2, "pan" + ext))
    kt = os.path.join(os.path.dirname(CLI), "ktools")
    r = subprocess.run([kt, "serialize", "pan"], cwd=d2, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    for ext in (".kmers.dbi", ".fl.kdb", ".tre.kdb"):
        assert open(pref + ext, "rb").read() == open(os.path.join(d2, "pan" + ext), "rb").read(), ext
    assert subprocess.run([kt], stderr=subprocess.PIPE).returncode == 0


@pytest.mark.gpu
def test_cli_aln_gz_is_the_stdout_stream(tmp_path):
    """--aln-gz FILE (BASELINE config 5: gzip on the host, overlapped): `zcat FILE` is what stdout carries without it —
    the golden -ae records of g5 — whatever the number of deflate threads and chunks."""
    import gzip
    d, _ = golden_cmds("g5_walk_k25")
    w = str(tmp_path / "w")
    shutil.copytree(d, w)
    want = open(os.path.join(d, "refae.aln.txt"), "rb").read()
    # (a regular interleaved file: the lines are assembled and compressed on the GPU, dbtk_gz.h; --host-ingest / another --gz-level: zlib on
    # the host's emit pool)
    for extra in ([], ["--emit-threads", "3", "--gz-level", "1"], ["--host-ingest"], ["--gz-level", "4"]):
        # (several blocks, several gzip members each; three slots: the carried-over bytes of block i + 2 land in front of the device block that
        # block i's lines are still being assembled from unless the submission holds back)
        env = dict(os.environ, DBTK_INGEST_CHUNK="60000", DBTK_INGEST_SLOTS="3") if not extra else dict(os.environ)
        r = subprocess.run([CLI, "--v13-threading", "-gc", "85", "3", "-ae", "-ka", "-k", "25", "-cth", "45", "-fa", "reads.fa", "-qs", "pan", "-o", "gz",
                            "--aln-gz", "out.aln.gz"] + extra, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert r.stdout == b""
        assert gzip.open(os.path.join(w, "out.aln.gz"), "rb").read() == want
        assert open(os.path.join(w, "gz.trkmc.ar"), "rb").read() == open(os.path.join(d, "refae.trkmc.ar"), "rb").read()
        em = [l for l in r.stderr.decode().splitlines() if l.startswith("emit:")][0]
        on_device = int(em.split("; ")[-1].split()[0])
        assert (on_device == len(want)) == (extra in ([], ["--emit-threads", "3", "--gz-level", "1"])), em
    # without records (the counting run of config 4) the command line MERGES its parsed blocks into larger batches on the device: whatever
    # the cut (every block its own batch, batches of >= 90 pairs, one batch, blocks aligned one by one as before), the counts are the golden ones
    for tag, env in (("m1", dict(DBTK_MERGE_PAIRS="1")), ("m_all", dict(DBTK_MERGE_PAIRS="1000000")), ("mfile", {}), ("mno", dict(DBTK_NO_MERGE="1"))):
        r = subprocess.run([CLI, "--v13-threading", "-gc", "85", "3", "-ka", "-k", "25", "-cth", "45", "-fa", "reads.fa", "-qs", "pan", "-o", tag], cwd=w,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, DBTK_INGEST_CHUNK="8000", DBTK_INGEST_SLOTS="3", **env))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert open(os.path.join(w, tag + ".trkmc.ar"), "rb").read() == open(os.path.join(d, "refae.trkmc.ar"), "rb").read(), tag
        paths = [l for l in r.stderr.decode().splitlines() if l.startswith("kernel paths:")][0].split()
        on_locus_path = int(paths[4]) + int(paths[paths.index("walk") + 1])
        # one merged batch (65 pairs per locus): enough for the kernels that keep a locus in LDS; the 8 000-byte blocks (6 pairs per locus) one by one: never
        assert (on_locus_path > 0) == (tag == "m_all"), (tag, " ".join(paths))
    # the README's own command line for the v1.3 contract (README.md:38-39: no flag of ours), switched by the environment
    r = subprocess.run([CLI, "-gc", "85", "3", "-ae", "-ka", "-k", "25", "-cth", "45", "-fa", "reads.fa", "-qs", "pan", "-o", "envsw"], cwd=w,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, DBTK_V13_THREADING="1"))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout == want
    assert open(os.path.join(w, "envsw.trkmc.ar"), "rb").read() == open(os.path.join(d, "refae.trkmc.ar"), "rb").read()


@pytest.mark.gpu
def test_cli_readme_pipe_takes_the_device_reader(tmp_path):
    """/root/reference/README.md:38-39: `samtools fasta -n ... | danbing-tk -gc 85 3 -ae ... -fa /dev/stdin | gzip`.  Reads arriving
    through a PIPE are parsed, paired, walked and formatted on the device like those of a file (stderr says so), stdout is the golden
    -ae stream of g5, whatever the chunking; and a pipe whose records stop being interleaved halfway is handed over to the host reader
    at exactly that byte (the bytes already taken from the pipe are replayed to it): same stdout and counts as from the file."""
    d, _ = golden_cmds("g5_walk_k25")
    w = str(tmp_path / "w")
    shutil.copytree(d, w)
    want = open(os.path.join(d, "refae.aln.txt"), "rb").read()
    base = ["-gc", "85", "3", "-ae", "-ka", "-k", "25", "-cth", "45", "-fa", "/dev/stdin", "-qs", "pan", "-o"]
    data = open(os.path.join(w, "reads.fa"), "rb").read()
    for tag, chunk in (("p0", None), ("p1", "60000"), ("p2", "4096")):
        env = dict(os.environ, DBTK_V13_THREADING="1")
        if chunk:
            env.update(DBTK_INGEST_CHUNK=chunk, DBTK_INGEST_SLOTS="3")
        # (a real pipe: `cat` feeds it)
        cat = subprocess.Popen(["cat", "reads.fa"], cwd=w, stdout=subprocess.PIPE)
        r = subprocess.run([CLI] + base + [tag], cwd=w, stdin=cat.stdout, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        cat.wait()
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert r.stdout == want
        assert open(os.path.join(w, tag + ".trkmc.ar"), "rb").read() == open(os.path.join(d, "refae.trkmc.ar"), "rb").read()
        dev = [l for l in r.stderr.decode().splitlines() if l.startswith("device reader:")]
        assert dev and "takes over" not in r.stderr.decode(), r.stderr.decode()[-1500:]
    # a chunk boundary exactly at the end of the input (the look-ahead read finds the pipe closed), and an empty pipe
    r = subprocess.run([CLI] + base + ["pe"], cwd=w, input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, DBTK_V13_THREADING="1", DBTK_INGEST_CHUNK=str(len(data))))
    assert r.returncode == 0 and r.stdout == want, r.stderr.decode()[-1500:]
    r = subprocess.run([CLI] + base + ["pz"], cwd=w, input=b"", stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, DBTK_V13_THREADING="1"))
    assert r.returncode == 0 and r.stdout == b"", r.stderr.decode()[-1500:]
    # not interleaved from the middle on: records 2q, 2q + 1 of the second half swapped with their neighbours' so that mates are two apart
    recs = [l for l in data.split(b">") if l]
    half = (len(recs) // 4) * 2
    tail = recs[half:]
    mixed = recs[:half]
    for q in range(0, len(tail) - 3, 4):
        mixed += [tail[q], tail[q + 2], tail[q + 1], tail[q + 3]]
    mixed += tail[len(tail) - len(tail) % 4:]
    blob = b"".join(b">" + x for x in mixed)
    open(os.path.join(w, "mixed.fa"), "wb").write(blob)
    rf = subprocess.run([CLI] + base[:-5] + ["-fa", "mixed.fa", "-qs", "pan", "-o", "mf", "--host-ingest"], cwd=w, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                        env=dict(os.environ, DBTK_V13_THREADING="1"))
    assert rf.returncode == 0, rf.stderr.decode()[-1500:]
    for chunk in ("60000", "5000"):
        rp = subprocess.run([CLI] + base + ["mp"], cwd=w, input=blob, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                            env=dict(os.environ, DBTK_V13_THREADING="1", DBTK_INGEST_CHUNK=chunk, DBTK_INGEST_SLOTS="3"))
        assert rp.returncode == 0, rp.stderr.decode()[-1500:]
        assert b"the host reader takes over" in rp.stderr
        assert sorted(rp.stdout.splitlines()) == sorted(rf.stdout.splitlines())
        assert open(os.path.join(w, "mp.trkmc.ar"), "rb").read() == open(os.path.join(w, "mf.trkmc.ar"), "rb").read()
    # a pair whose TITLES are 1.1 MB in the middle of the input: what lies behind the last whole pair of a block outgrows the carry-over
    # room (DBTK_ING_CARRY alone): that block is aligned and RELEASED before the host reader takes over behind it — and the pipe's reader
    # must not refill its slot meanwhile (the replayed bytes start in that very buffer)
    t = b"x" * 1_100_000
    big = recs[:half] + [t + b"/1\n" + recs[half].split(b"\n", 1)[1], t + b"/2\n" + recs[half + 1].split(b"\n", 1)[1]] + recs[half + 2:]
    blob = b"".join(b">" + x for x in big)
    open(os.path.join(w, "big.fa"), "wb").write(blob)
    rf = subprocess.run([CLI] + base[:-5] + ["-fa", "big.fa", "-qs", "pan", "-o", "bf", "--host-ingest"], cwd=w, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                        env=dict(os.environ, DBTK_V13_THREADING="1"))
    assert rf.returncode == 0 and len(rf.stdout) > 1_100_000, rf.stderr.decode()[-1500:]  # (the long title is printed)
    for slots in ("2", "3"):
        cat = subprocess.Popen(["cat", "big.fa"], cwd=w, stdout=subprocess.PIPE)
        rp = subprocess.run([CLI] + base + ["bp"], cwd=w, stdin=cat.stdout, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                            env=dict(os.environ, DBTK_V13_THREADING="1", DBTK_INGEST_CHUNK="65536", DBTK_INGEST_SLOTS=slots))
        cat.wait()
        assert rp.returncode == 0, rp.stderr.decode()[-1500:]
        assert b"the host reader takes over" in rp.stderr
        assert sorted(rp.stdout.splitlines()) == sorted(rf.stdout.splitlines())
        assert open(os.path.join(w, "bp.trkmc.ar"), "rb").read() == open(os.path.join(w, "bf.trkmc.ar"), "rb").read()


@pytest.mark.gpu
def test_cli_simulation_mode_alignments(tmp_path):
    """-s 1 with -a / -ae under the v1.3 contract (writeAlignments prints sams[i].src, AQ.cpp:1744-1745; in simulation mode -ae keeps a
    walked pair when its source OR its destination is a locus, AQ.cpp:2241-2247): the -a lines are the plain run's with the source
    locus in the first column; the -ae lines are those of them whose source or destination is a locus."""
    d, _ = golden_cmds("g5_walk_k25")
    w = str(tmp_path / "w")
    shutil.copytree(d, w)
    nloci = sum(1 for l in open(os.path.join(w, "pan.tr.kmers")) if l.startswith(">"))
    recs = [r for r in open(os.path.join(w, "reads.fa"), "rb").read().split(b">") if r]
    out, srcs = [], {}
    for i in range(0, len(recs) - 1, 2):  # titles >LOCUS.i (simmode 1, AQ.cpp:478-489); every fifth pair "not from a locus"
        src = nloci if (i // 2) % 5 == 0 else (i // 2) % nloci
        t = b"%d.p%d" % (src, i // 2)
        srcs[">" + t.decode()] = src
        for r in recs[i:i + 2]:
            out.append(b">" + t + b"\n" + r.split(b"\n", 1)[1])
    open(os.path.join(w, "sim.fa"), "wb").write(b"".join(out))
    base = ["--v13-threading", "-gc", "85", "3", "-ka", "-k", "25", "-cth", "45", "-qs", "pan"]

    def lines(extra, fa):
        r = run(base + extra + ["-fa", fa], cwd=w)
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        return [l.split("\t") for l in r.stdout.decode().splitlines()]
    plain = lines(["-a", "-o", "a0"], "sim.fa")
    sim_a = lines(["-a", "-s", "1", "-o", "a1"], "sim.fa")
    assert len(plain) == len(sim_a) > 100
    assert [[str(srcs[l[2]])] + l[1:] for l in plain] == sim_a
    sim_ae = lines(["-ae", "-s", "1", "-o", "a2"], "sim.fa")
    assert sim_ae == [l for l in sim_a if int(l[0]) != nloci or int(l[1]) != nloci]
    assert 0 < len(sim_ae) < len(sim_a) or all(int(l[1]) != nloci for l in sim_a)


# ---- the ingest (reader, splitters, pairing) on the CPU: `--parse-only` reports what the pairing stage handed on --------
def _digest(reads, fastq):
    """pairs, bases and the order-independent digest dbtk_cli.cpp computes with --parse-only, from tests/refio.py's restatement
    of the reference's reader (AQ.cpp:1918-1976)."""
    M = (1 << 64) - 1

    def fnv(h, b):
        for c in b:
            h = ((h ^ c) * 0x100000001B3) & M
        return ((h ^ 0xFF) * 0x100000001B3) & M
    dg = 0
    for p in range(reads.npairs):
        h = fnv(0xCBF29CE484222325, reads.titles[p].encode())
        h = fnv(fnv(h, reads.seqs[2 * p]), reads.seqs[2 * p + 1])
        if fastq:
            h = fnv(fnv(h, reads.quals[2 * p]), reads.quals[2 * p + 1])
        dg = (dg + h) & M
    return reads.npairs, sum(len(s) for s in reads.seqs), dg


def _write_records(fn, recs, fastq):
    with open(fn, "wb") as f:
        for t, s, q in recs:
            f.write(t + b"\n" + s + b"\n" + ((b"+\n" + q + b"\n") if fastq else b""))


@pytest.mark.parametrize("fastq", [False, True])
def test_ingest_pairs_like_the_reference_reader(tmp_path, fastq):
    """Interleaved mates, mates far apart, singletons, a title seen three and four times, short reads, /1 /2 suffixes, a
    missing final newline — through block sizes from a few records to one block: the pairs handed to the aligner are
    exactly those of the reference's reader (as restated in tests/refio.py)."""
    import numpy as np
    import refio
    rng = np.random.default_rng(11 + fastq)
    mk = lambda n: bytes(rng.choice(list(b"ACGTN"), n, p=[.24, .24, .24, .24, .04]).astype(np.uint8))
    q = lambda n: bytes(rng.integers(35, 74, n).astype(np.uint8)) if fastq else b""
    lead = b"@" if fastq else b">"
    recs, late = [], []
    for i in range(900):
        t = lead + b"r%d" % i
        n1, n2 = (int(rng.integers(20, 151)) for _ in range(2))
        a, b = (t + b"/1", mk(n1), q(n1)), (t + b"/2", mk(n2), q(n2))
        u = rng.random()
        if u < 0.6:
            recs += [a, b]                      # adjacent
        elif u < 0.8:
            recs.append(a); late.append(b)      # mate arrives much later
        elif u < 0.9:
            recs.append(a)                      # singleton
        elif u < 0.95:
            recs += [a, b, (t + b"/1", mk(80), q(80))]            # a triple
        else:
            recs += [a, b, (t + b"/1", mk(90), q(90)), (t, mk(95), q(95))]  # four of a kind
        if len(late) > 40 and rng.random() < 0.3:
            rng.shuffle(late)
            recs += late
            late = []
    recs += late
    fn = str(tmp_path / ("reads.fq" if fastq else "reads.fa"))
    _write_records(fn, recs, fastq)
    with open(fn, "rb+") as f:  # no newline after the last line
        f.seek(-1, 2)
        f.truncate()
    d = os.path.join(GOLDEN, "g1_k21")
    for cth in (10, 45):
        want = _digest(refio.read_pairs(fn, fastq, cth + 21 - 1), fastq)
        for blk in ("64", "300", "5000", None):
            env = dict(os.environ)
            if blk:
                env["DBTK_INGEST_BLOCK"] = blk
            r = run(["--parse-only", "-k", "21", "-cth", str(cth), "-fq" if fastq else "-fa", fn, "-qs", "pan", "-o", str(tmp_path / "o")], cwd=d, env=env)
            assert r.returncode == 0, r.stderr.decode()[-1500:]
            got = tuple(int(x) for x in r.stdout.decode().split()[1:])
            assert got == want, (blk, cth, got, want)


@pytest.mark.parametrize("fastq", [False, True])
def test_sharded_ingest_equals_single_reader(tmp_path, fastq):
    """--gpus N on a seekable file: N byte ranges cut at record boundaries, each with its own reader / pairing pipeline,
    what a range could not pair is paired across ranges at the end.  The pairs handed to the aligners (titles occurring at
    most twice: adjacent mates, mates far apart — also across the cuts —, singletons, short reads) are the single reader's."""
    import numpy as np
    import refio
    rng = np.random.default_rng(21 + fastq)
    mk = lambda n: bytes(rng.choice(list(b"ACGT"), n).astype(np.uint8))
    q = lambda n: bytes(rng.integers(35, 74, n).astype(np.uint8)) if fastq else b""
    lead = b"@" if fastq else b">"
    recs, late = [], []
    for i in range(3000):
        t = lead + b"r%d" % i
        n1, n2 = (int(rng.integers(25, 151)) for _ in range(2))
        a, b = (t + b"/1", mk(n1), (b"@" + q(n1)[1:]) if fastq and i % 7 == 0 else q(n1)), (t + b"/2", mk(n2), q(n2))  # (some quality lines begin with '@')
        u = rng.random()
        if u < 0.7:
            recs += [a, b]
        elif u < 0.9:
            recs.append(a); late.append(b)
        else:
            recs.append(a)
        if len(late) > 300 and rng.random() < 0.05:
            rng.shuffle(late)
            recs += late
            late = []
    recs += late
    fn = str(tmp_path / ("reads.fq" if fastq else "reads.fa"))
    _write_records(fn, recs, fastq)
    d = os.path.join(GOLDEN, "g1_k21")
    want = _digest(refio.read_pairs(fn, fastq, 45 + 21 - 1), fastq)
    for gpus in ("1", "2", "5"):
        r = run(["--parse-only", "--gpus", gpus, "-k", "21", "-cth", "45", "-fq" if fastq else "-fa", fn, "-qs", "pan", "-o", str(tmp_path / "o")], cwd=d,
                env=dict(os.environ, DBTK_SHARD_MIN="0"))
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        got = tuple(int(x) for x in r.stdout.decode().split()[1:])
        assert got == want, (gpus, got, want)
        if gpus != "1":
            assert b"cross-range pairing" in r.stderr
    # more ranges than GPUs (--ingest-shards): the same pairs again
    r = run(["--parse-only", "--gpus", "2", "--ingest-shards", "7", "-k", "21", "-cth", "45", "-fq" if fastq else "-fa", fn, "-qs", "pan", "-o", str(tmp_path / "o")],
            cwd=d, env=dict(os.environ, DBTK_SHARD_MIN="0"))
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    assert tuple(int(x) for x in r.stdout.decode().split()[1:]) == want


@pytest.mark.gpu
def test_cli_device_reader_equals_host_reader(tmp_path):
    """The reader on the device (dbtk_ingest_*; default for a regular file) against the host reader (--host-ingest) of the same
    binary, itself pinned to the reference's reader above: same stdout, same output files — interleaved FASTA in many small
    blocks (records and titles straddle chunks), without a final newline, FASTQ extraction, no-record mode (asynchronous
    blocks), and an input that stops being interleaved halfway (the host reader takes over at that byte)."""
    loci = synth.make_loci(nloci=40, nhap=2, flank=500, seed=91, shared_frac=0.3)
    d = str(tmp_path)
    arr = synth.build_rpgg_arrays(loci, 21)
    synth.write_rpgg_files(arr, os.path.join(d, "pan"))
    reads = synth.sim_reads(loci, npairs=5000, seed=92, sub=0.006, indel=0.001, nrate=0.002, chimeric=0.2, background=0.2, short=0.03, with_qual=True)
    synth.write_fasta(reads, os.path.join(d, "r.fa"))
    synth.write_fasta(reads, os.path.join(d, "r.fq"), fastq=True)
    data = open(os.path.join(d, "r.fa"), "rb").read()
    open(os.path.join(d, "nonl.fa"), "wb").write(data[:-1])
    recs = data.split(b">")[1:]
    half = sum(len(r) + 1 for r in recs[:5000])
    open(os.path.join(d, "half.fa"), "wb").write(data[:half] + b">orphan/1\n" + reads.seqs[0] + b"\n" + data[half:])
    open(os.path.join(d, "odd.fa"), "wb").write(data + b">last/1\n" + reads.seqs[2] + b"\n")  # (ends inside a pair: the tail is the host reader's)
    for flags, fn, chunk in ((["-cth", "45"], "r.fa", "20000"), (["-cth", "45"], "nonl.fa", "50000"), (["-cth", "45", "-ka"], "r.fa", "8192"),
                             (["-cth", "30", "-e", "1"], "r.fq", "30000"), (["-cth", "30", "-kf", "8", "2"], "r.fq", None), (["-cth", "45"], "half.fa", "65536"), (["-cth", "45"], "odd.fa", "40000")):
        outs = []
        for tag, extra in (("host", ["--host-ingest"]), ("dev", [])):
            env = dict(os.environ)
            if chunk:
                env["DBTK_INGEST_CHUNK"] = chunk
            a = ["-k", "21"] + flags + ["-fq" if fn.endswith(".fq") else "-fa", fn, "-qs", "pan", "-o", tag] + extra
            r = subprocess.run([CLI] + a, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            assert r.returncode == 0, r.stderr.decode()[-1500:]
            outs.append(r)
        assert outs[0].stdout == outs[1].stdout, (flags, fn)
        ing = [l for l in outs[1].stderr.decode().splitlines() if l.startswith("ingest:")][0]
        ndev = int(ing.split("device reader: ")[1].split()[0])
        total = int(ing.split(" s for ")[1].split()[0])
        if fn == "half.fa":
            assert 0 < ndev < total and b"the host reader takes over" in outs[1].stderr
        elif fn == "odd.fa":
            assert ndev == total > 0 and b"the host reader takes over" in outs[1].stderr  # (every pair on the device; the lone record is parked by the host reader)
        else:
            assert ndev == total > 0, ing
        assert [l for l in outs[0].stderr.decode().splitlines() if l[:1].isdigit() and " reads " in l] == \
               [l for l in outs[1].stderr.decode().splitlines() if l[:1].isdigit() and " reads " in l]
        if "-e" not in flags:
            for ext in (".trkmc.ar", ".tr.summary.txt"):
                assert open(os.path.join(d, "host" + ext), "rb").read() == open(os.path.join(d, "dev" + ext), "rb").read(), (flags, fn, ext)
    # three byte ranges, each with its own device reader and context on the one GPU: the cuts fall on pair boundaries, so every
    # range is interleaved from its first byte and nothing is left to the cross-range pairing
    env = dict(os.environ, DBTK_SHARD_MIN="0", DBTK_INGEST_CHUNK="100000")
    r3 = subprocess.run([CLI, "-k", "21", "-cth", "45", "--ingest-shards", "3", "-fa", "r.fa", "-qs", "pan", "-o", "sh3"], cwd=d, stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, env=env)
    assert r3.returncode == 0, r3.stderr.decode()[-1500:]
    r1 = subprocess.run([CLI, "-k", "21", "-cth", "45", "-fa", "r.fa", "-qs", "pan", "-o", "one", "--host-ingest"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r1.returncode == 0
    ing = [l for l in r3.stderr.decode().splitlines() if l.startswith("ingest:")][0]
    assert int(ing.split("device reader: ")[1].split()[0]) == int(ing.split(" s for ")[1].split()[0]) > 0, ing
    assert b"cross-range pairing: 0 reads" in r3.stderr
    assert sorted(r3.stdout.splitlines()) == sorted(r1.stdout.splitlines())
    for ext in (".trkmc.ar", ".tr.summary.txt"):
        assert open(os.path.join(d, "sh3" + ext), "rb").read() == open(os.path.join(d, "one" + ext), "rb").read(), ext
