"""The reader on the device (danbing-tk_amd/csrc/dbtk_ingest.h; include/dbtk.h: dbtk_ingest_*): record boundaries, mate pairing
and the batch arrays made by kernels from raw FASTA / FASTQ bytes.  Reference semantics: src/aQueryFasta_thread.cpp:1918-1976
(getline title / seq [/ + / qual]; prunePEinfo :455-462; a record is parked under its title until a record with the same title
arrives, which then is seqs[2p] and the parked one seqs[2p + 1]; pairs with a read shorter than Cthreshold + k - 1 are dropped).
CPU: the kernel bodies on the emulated lanes against a plain-Python restatement of that reader.  GPU: the library's
dbtk_ingest_* against the host-buffer entry point on the same reads (tests/test_gpu_parity.py style), and the command line with
and without the device reader (tests/test_cli.py)."""
import os
import random

import pytest

import bind

ACGT = b"ACGT"


def ref_reader(data: bytes, fastq: bool, min_read: int):
    """The reference's reader on a whole input: [(title, seq of the completing record, seq of the parked one, qual, qual)] and
    what stayed parked.  std::getline semantics: a last line without a newline is a line; nothing after the last newline is none."""
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    L = 4 if fastq else 2
    parked, out = {}, []
    i = 0
    while i < len(lines):
        rec = lines[i:i + L] + [b""] * (L - len(lines[i:i + L]))  # a failed getline leaves an empty string
        i += L
        title, seq, qual = rec[0], rec[1], (rec[3] if fastq else b"")
        if len(title) >= 2 and title[-2:] in (b"/1", b"/2"):
            title = title[:-2]
        if title in parked:
            s2, q2 = parked.pop(title)
            if len(seq) >= min_read and len(s2) >= min_read:
                out.append((title, seq, s2, qual, q2))
        else:
            parked[title] = (seq, qual)
    return out, parked


def make_input(rng, npairs, fastq, suffix=True, short_every=0, newline_at_end=True, qual_short_every=0):
    recs = []
    for p in range(npairs):
        name = (b"@" if fastq else b">") + b"read" + str(rng.randrange(10 ** rng.randrange(1, 9))).encode() + b"_" + str(p).encode()
        for m in (1, 2):
            n = rng.randrange(60, 152)
            if short_every and p % short_every == 3 and m == 2:
                n = rng.randrange(0, 40)
            seq = bytes(rng.choice(ACGT + b"N") for _ in range(n))
            t = name + (b"/" + str(m).encode() if suffix else b"")
            if fastq:
                qn = n - 3 if (qual_short_every and p % qual_short_every == 1 and n > 3) else n
                q = bytes(rng.choice(b"@>+!#5ABCI") for _ in range(qn))
                recs.append(t + b"\n" + seq + b"\n+\n" + q + b"\n")
            else:
                recs.append(t + b"\n" + seq + b"\n")
    data = b"".join(recs)
    return data if newline_at_end else data[:-1]


def padded(q, s):
    return (q + b"!" * len(s))[:len(s)]


@pytest.fixture(scope="module")
def emu():
    return bind.Emu()


@pytest.mark.parametrize("fastq", [False, True])
@pytest.mark.parametrize("chunk", [257, 1000, 4096, 50000])
def test_emulated_reader_equals_the_reference_reader_on_interleaved_input(emu, fastq, chunk):
    rng = random.Random(chunk * 2 + fastq)
    data = make_input(rng, 150, fastq, suffix=chunk != 1000, short_every=7, newline_at_end=chunk != 4096, qual_short_every=5)
    want, parked = ref_reader(data, fastq, 55)
    assert not parked and len(want) < 150
    H, got, used = emu.ingest(data, fastq, 55, chunk, head=1024 if fastq else 512)
    assert all(h["flags"] == 0 for h in H), H
    assert used == len(data) and sum(h["npairs"] for h in H) == 150 and sum(h["nkept"] for h in H) == len(want)
    assert len(H) == (len(data) + chunk - 1) // chunk
    assert [g[:3] for g in got] == [w[:3] for w in want]
    if fastq:  # what the bait filter's quality mask reads: the quality string cut or '!'-padded to its read's length
        assert [(g[3], g[4]) for g in got] == [(padded(w[3], w[1]), padded(w[4], w[2])) for w in want]
    assert max(h["maxlen"] for h in H) == max(max(len(w[1]), len(w[2])) for w in want)


def test_block_with_a_singleton_is_flagged_and_the_host_reader_can_take_over(emu):
    rng = random.Random(5)
    a, b = make_input(rng, 40, False), make_input(rng, 40, False)
    data = a + b">lonely\nACGTACGTAC\n" + b
    H, got, _ = emu.ingest(data, False, 20, 2000, head=512)
    assert H[-1]["flags"] & 1 and all(h["flags"] == 0 for h in H[:-1])
    # every pair before the flagged block was delivered, and the flagged block starts at a record the host reader can start from
    first = sum(min(2000, len(data) - i * 2000) for i in range(len(H) - 1)) - (512 - H[-1]["base"])
    before, parked = ref_reader(data[:first], False, 20)
    assert not parked and [g[:3] for g in got] == [w[:3] for w in before]
    rest, _ = ref_reader(data[first:], False, 20)
    assert before + rest == ref_reader(data, False, 20)[0]


def test_mates_that_are_not_adjacent_are_flagged(emu):
    rng = random.Random(6)
    data = make_input(rng, 30, True)
    recs = data.split(b"\n")
    recs = [b"\n".join(recs[i:i + 4]) + b"\n" for i in range(0, len(recs) - 1, 4)]
    recs[10], recs[13] = recs[13], recs[10]
    H, got, _ = emu.ingest(b"".join(recs), True, 20, 1 << 20, head=512)
    assert len(H) == 1 and H[0]["flags"] & 1 and not got


def test_input_that_ends_inside_a_pair_is_flagged_after_its_whole_pairs(emu):
    rng = random.Random(7)
    data = make_input(rng, 25, False)
    cutoff = data.rindex(b">")  # the last record is missing
    H, got, _ = emu.ingest(data[:cutoff], False, 20, 3000, head=512)
    assert H[-1]["flags"] == 8 and H[-1]["carry"] > 0
    want, parked = ref_reader(data[:cutoff], False, 20)
    assert len(parked) == 1 and [g[:3] for g in got] == [w[:3] for w in want]


def test_line_table_overflow_and_empty_input(emu):
    H, got, _ = emu.ingest(b"\n" * 3000, False, 20, 4096, head=256, line_cap=1000)
    assert H[0]["flags"] & 2 and not got
    H, got, _ = emu.ingest(b"", False, 20, 4096)
    assert len(H) == 1 and H[0]["flags"] == 0 and H[0]["npairs"] == 0 and not got
    # a record longer than the carry-over room
    H, got, _ = emu.ingest(b">a\n" + b"A" * 3000 + b"\n>a\n" + b"C" * 3000 + b"\n", False, 20, 1024, head=256)
    assert H[0]["flags"] & 4



def test_emulated_gzip_members_gunzip_to_the_text(emu):
    """dbtk_gz.h: the -a / -ae text as gzip members made by the kernel body (LZ77 tokens per lane, literal / length and distance
    histograms, minimum-redundancy code lengths limited to 15 bits, canonical codes, lanes writing at scanned bit offsets, CRC-32
    combined from the lanes' spans): the system's zlib
    (which checks every member's CRC-32 and length) must return the text — alignment-like lines, all byte values, one symbol,
    incompressible bytes, a Fibonacci-skewed histogram (Huffman depths past 15), sizes around the member and span boundaries."""
    import gzip
    import zlib
    rng = random.Random(11)

    def line():
        return (b".\t%d\t>read%d\t" % (rng.randrange(80000), rng.randrange(10 ** 8)) + bytes(rng.choice(ACGT) for _ in range(150)) + b"\t" +
                bytes(rng.choice(ACGT) for _ in range(150)) + b"\t150=\t150=\t148=1X1=\t149.1*\n")
    text = b"".join(line() for _ in range(700))
    fib, a, b = b"", 1, 1
    for sym in range(24):  # frequencies 1, 1, 2, 3, 5, ...: the unrestricted code is 23 bits deep
        fib += bytes([65 + sym]) * a
        a, b = b, a + b
    shuffled = bytearray(fib)
    rng.shuffle(shuffled)
    cases = [text[:n] for n in (1, 5, 1000, 1023, 1024, 1025, 65535, 65536, 65537, 200000)] + [bytes(range(256)) * 300, b"A" * 70000,
             bytes(rng.randrange(256) for _ in range(100000)), b"AB" * 40000 + b"C", bytes(shuffled)]
    for t in cases:
        z = emu.gz(t, grid=3)
        assert gzip.decompress(z) == t, len(t)
        assert z.count(b"\x1f\x8b\x08") >= (len(t) + 65535) // 65536
    assert emu.gz(b"") == b""
    z = emu.gz(text[:200000])
    assert len(z) < 1.1 * len(zlib.compress(text[:200000], 1))  # DNA-dominated text with random titles: within 10 % of zlib's level 1
    # Round 6 (LZ77 in the kernel body: per-lane greedy matches of >= 4 bytes at distances <= 16 KB): lines as a sequencer names them —
    # titles that share a prefix with their neighbours', CIGARs and annotations mostly those of the line before — come out SMALLER than
    # zlib's level 1 makes them (the DNA itself does not compress beyond its 2 bits per base either way)
    def line2(i):
        return (b".\t%d\tA00123:45:HXXX:1:1101:%d:%d\t" % (rng.randrange(80000), 1000 + i // 3, 5000 + rng.randrange(900)) +
                bytes(rng.choice(ACGT) for _ in range(150)) + b"\t" + bytes(rng.choice(ACGT) for _ in range(150)) +
                (b"\t150=\t130=\t150=\t130=\n" if rng.random() < 0.7 else b"\t70=XA79=\t50=21.59=\t150=\t130=\n"))
    text2 = b"".join(line2(i) for i in range(600))
    z2 = emu.gz(text2)
    assert gzip.decompress(z2) == text2
    assert len(z2) < len(zlib.compress(text2, 1)), (len(z2), len(zlib.compress(text2, 1)))
    # (the switch DBTK_GZ_LZ=0 of the library: literals only — a valid, larger stream)
    os.environ["EMU_GZ_LZ"] = "0"
    try:
        z0 = emu.gz(text2)
    finally:
        os.environ.pop("EMU_GZ_LZ", None)
    assert gzip.decompress(z0) == text2 and len(z0) > len(z2)
    # long matches, matches that end exactly at span / member boundaries, a repeat longer than the longest match (258)
    for t in (b"ACGTTGCA" * 20000, (b"x" * 300 + b"\n") * 500, text2[:1024] * 70, bytes(rng.randrange(65, 69) for _ in range(1024)) * 65):
        assert gzip.decompress(emu.gz(t, grid=2)) == t, len(t)


# ---------------------------------------------------------------------------------------------------------------- GPU
def _fasta_of(reads, fastq=False, drop_newline=False):
    out = []
    for p in range(reads.npairs):
        t = reads.titles[p].encode()
        for which, tag in ((2 * p + 1, b"/2"), (2 * p, b"/1")):  # (the reader makes the LATER record of a title seqs[2p])
            if fastq:
                out.append(b"@" + t + tag + b"\n" + reads.seqs[which] + b"\n+\n" + reads.quals[which] + b"\n")
            else:
                out.append(b">" + t + tag + b"\n" + reads.seqs[which] + b"\n")
    data = b"".join(out)
    return data[:-1] if drop_newline else data


@pytest.mark.gpu
@pytest.mark.parametrize("name,chunk", [("mixed", 1 << 20), ("mixed", 20000), ("lengths", 9000), ("short25", 4096)])
def test_device_reader_feeds_the_hot_path_like_host_buffers(name, chunk, tmp_path):
    """dbtk_ingest_* on the raw bytes of an interleaved FASTA against dbtk_align_batch on the same reads from host arrays:
    same records for every pair (trace mode), same accumulators; the spans name the same titles and reads."""
    import cases
    import numpy as np
    abi = bind.abi
    c = cases.make_case(name, str(tmp_path))
    dbtk = bind.pkg.Dbtk()
    g = dbtk.load(c.prefix, c.k)
    seq, off = c.reads.packed()
    cth = 30
    minread = cth + c.k - 1
    p = abi.default_params(ksize=c.k, cthreshold=cth, trace=1)
    # host buffers: only the pairs the reader keeps
    keep = [q for q in range(c.reads.npairs) if len(c.reads.seqs[2 * q]) >= minread and len(c.reads.seqs[2 * q + 1]) >= minread]
    import synth
    sub = synth.Reads()
    for q in keep:
        sub.seqs += [c.reads.seqs[2 * q], c.reads.seqs[2 * q + 1]]
        sub.titles.append(c.reads.titles[q])
    hs, ho = sub.packed()
    ctx = dbtk.context(g, p)
    recs_h, n_h = ctx.align(hs, ho)
    want = ctx.counts()
    ctx.close()
    data = _fasta_of(c.reads, drop_newline=chunk == 20000)
    ctx = dbtk.context(g, p)
    if chunk in (20000, 9000):  # the slots' pinned buffers made ahead (dbtk_ingest_reserve_host): two of the three, the third on first use
        dbtk.reserve_host(0, chunk, 2)
    ing = bind.pkg.Ingest(ctx, False, minread, chunk, nslots=3)
    got_recs, titles, q0 = [], [], 0
    pos, pending = 0, []
    nblocks = max(1, (len(data) + chunk - 1) // chunk)
    for j in range(nblocks):  # two blocks in flight
        pending.append(ing.submit(data[pos:pos + chunk], j == nblocks - 1))
        pos += chunk
        if len(pending) == 2 or j == nblocks - 1:
            while pending and (len(pending) == 2 or j == nblocks - 1):
                slot = pending.pop(0)
                info = ing.wait(slot)
                assert info.flags == 0
                recs, n = ing.align(slot, info, sync=True)
                assert n == info.nkept
                for r in recs[:n] if n else []:
                    rr = abi.PairRec.from_buffer_copy(r)
                    rr.pair += q0
                    got_recs.append(rr)
                for t, s0, s1, _, _ in ing.spans(slot, info):
                    q = len(titles)
                    assert t == b">" + sub.titles[q].encode() and s0 == sub.seqs[2 * q] and s1 == sub.seqs[2 * q + 1]
                    titles.append(t)
                q0 += info.nkept
    ing.close()
    got = ctx.counts()
    ctx.close()
    assert q0 == len(keep) == n_h
    arr = (abi.PairRec * max(len(got_recs), 1))(*got_recs)
    assert bind.recs_equal(recs_h, arr, n_h) < 0
    for k_ in ("counts", "kmc", "nmapread", "counters"):
        assert (want[k_] == got[k_]).all(), k_


@pytest.mark.gpu
def test_device_reader_asynchronous_blocks_and_flags(tmp_path):
    """sync = 0 (no records: the blocks' kernels overlap the next blocks' copies): same accumulators; a singleton in the
    input flags its block, which refuses to be aligned, and first_byte is where a host reader must continue."""
    import cases
    abi = bind.abi
    c = cases.make_case("mixed", str(tmp_path))
    dbtk = bind.pkg.Dbtk()
    g = dbtk.load(c.prefix, c.k)
    p = abi.default_params(ksize=c.k, cthreshold=45, okam=0)
    seq, off = c.reads.packed()
    ctx = dbtk.context(g, p)
    ctx.align(seq, off)
    want = ctx.counts()
    ctx.close()
    data = _fasta_of(c.reads)
    chunk = 16384
    ctx = dbtk.context(g, p)
    ing = bind.pkg.Ingest(ctx, False, 0, chunk, nslots=4, with_spans=False)
    nblocks = (len(data) + chunk - 1) // chunk
    slots = []
    for j in range(nblocks):
        slots.append(ing.submit(data[j * chunk:(j + 1) * chunk], j == nblocks - 1))
        if len(slots) == 3 or j == nblocks - 1:
            for s in (slots if j == nblocks - 1 else slots[:1]):
                info = ing.wait(s)
                assert info.flags == 0
                ing.align(s, info, sync=False)
            slots = [] if j == nblocks - 1 else slots[1:]
    ing.close()
    got = ctx.counts()
    ctx.close()
    for k_ in ("counts", "kmc", "nmapread", "counters"):
        assert (want[k_] == got[k_]).all(), k_
    # the same blocks MERGED into larger batches on the device (dbtk_ingest_align_merged): batches of >= 150 pairs cut wherever the blocks
    # end, one batch for everything, and a walking context (v1.3 threading: the merged batch goes through the walk kernels)
    for min_pairs, lanes in ((150, "2"), (10 ** 9, "1"), (1, "2")):
        os.environ["DBTK_LANES"] = lanes
        try:
            ctx = dbtk.context(g, p)
        finally:
            del os.environ["DBTK_LANES"]
        ing = bind.pkg.Ingest(ctx, False, 0, chunk, nslots=4, with_spans=False)
        slots = []
        for j in range(nblocks):
            slots.append(ing.submit(data[j * chunk:(j + 1) * chunk], j == nblocks - 1))
            if len(slots) == 3 or j == nblocks - 1:
                for s in (slots if j == nblocks - 1 else slots[:1]):
                    info = ing.wait(s)
                    assert info.flags == 0
                    ing.align_merged(s, min_pairs)
                slots = [] if j == nblocks - 1 else slots[1:]
        ing.align_merged(None, 0, flush=True)
        ing.align_merged(None, 0, flush=True)  # (nothing left: a no-op)
        got = ctx.counts()
        ing.close()
        ctx.close()
        for k_ in ("counts", "kmc", "nmapread", "counters"):
            assert (want[k_] == got[k_]).all(), (k_, min_pairs)
    # dbtk_ctx_reset discards pairs that were appended to the merged batch and not yet aligned: a flush after the reset finds
    # nothing, and a fresh run on the same context gives the run's own counts
    ctx = dbtk.context(g, p)
    ing = bind.pkg.Ingest(ctx, False, 0, chunk, nslots=4, with_spans=False)
    s0 = ing.submit(data[:chunk], nblocks == 1)
    info = ing.wait(s0)
    assert info.flags == 0 and info.nkept > 0
    ing.align_merged(s0, 10 ** 9)  # appended, not aligned
    ctx.reset()
    ing.align_merged(None, 0, flush=True)
    z = ctx.counts()
    assert not z["counts"].any() and not z["kmc"].any() and not z["nmapread"].any() and not z["counters"].any()
    ing.close()
    ctx.align(seq, off)
    got = ctx.counts()
    ctx.close()
    for k_ in ("counts", "kmc", "nmapread", "counters"):
        assert (want[k_] == got[k_]).all(), (k_, "after reset")
    # a singleton after 100 pairs
    recs = data.split(b">")[1:]
    cut = sum(len(r) + 1 for r in recs[:200])
    bad = data[:cut] + b">orphan\nACGTACGTACGTACGTACGTAGCATCAGCATCGACGACTAGCACGACTAGCATCAGCAT\n" + data[cut:]
    ctx = dbtk.context(g, p)
    ing = bind.pkg.Ingest(ctx, False, 0, 1 << 20, nslots=2)
    info = ing.wait(ing.submit(bad, True))
    assert info.flags & abi.ING_DIRTY and info.first_byte == 0
    with pytest.raises(bind.pkg.DbtkError):
        ing.align(0, info)
    ing.close()
    ing = bind.pkg.Ingest(ctx, False, 0, 8192, nslots=2)
    first = None
    for j in range((len(bad) + 8191) // 8192):
        info = ing.wait(ing.submit(bad[j * 8192:(j + 1) * 8192], (j + 1) * 8192 >= len(bad)))
        if info.flags:
            first = info.first_byte
            break
        ing.align(j % 2, info, sync=False)
    ing.close()
    ctx.close()
    assert first is not None and 0 < first <= cut and bad[first:first + 1] == b">" and bad[:first].count(b">") % 2 == 0
