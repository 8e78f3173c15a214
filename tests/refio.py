"""Test helpers that restate the reference's READER (not part of the product).

read_pairs() pairs mates the way CountWords' critical section A does
(src/aQueryFasta_thread.cpp:1918-1976): title without /1 /2, first-seen mate is
parked, the pair is (seq1 = later record, seq2 = parked record), pairs with a
mate shorter than Cth + k - 1 are dropped, leftovers are ignored."""
from __future__ import annotations

import numpy as np

import synth


def read_pairs(fn, fastq, minlen):
    reads = synth.Reads()
    parked = {}
    with open(fn, "rb") as f:
        lines = f.read().split(b"\n")
    step = 4 if fastq else 2
    for i in range(0, len(lines) - (step - 1), step):
        title, seq = lines[i], lines[i + 1]
        qual = lines[i + 3] if fastq else b""
        if len(title) >= 2 and title[-2:] in (b"/1", b"/2"):
            title = title[:-2]
        if title in parked:
            s2, q2 = parked.pop(title)
            if len(seq) < minlen or len(s2) < minlen:
                continue
            reads.seqs += [seq, s2]
            reads.quals += [qual, q2]
            reads.titles.append(title.decode())
        else:
            parked[title] = (seq, qual)
    return reads


def parse_kam(fn):
    """Fields of the reference's kam lines (writeKmerAssignments, AQ.cpp:1646-1681)."""
    out = []
    for line in open(fn):
        f = line.rstrip("\n").split("\t")
        r2 = f[6].split(":")
        r1 = f[7].split(":")
        out.append(dict(src=f[0], dst=int(f[1]), dst0=int(f[2]), len2=int(f[3]), len1=int(f[4]), r2=r2, r1=r1,
                        annot2=f[8], annot1=f[9], title=f[10], seq2=f[11], seq1=f[13]))
    return out


def annot_str(states):
    """km_asgn_t::annot2str_ (AQ.cpp:121-138): run-length of '*' '.' '='."""
    if not states:
        return "*"
    chs = "*.="
    s, ct, a0 = [], 1, states[0]
    for a1 in states[1:]:
        if a1 != a0:
            s.append(f"{ct}{chs[a0]}")
            ct = 1
        else:
            ct += 1
        a0 = a1
    s.append(f"{ct}{chs[a0]}")
    return "".join(s)


def mate_fields(m):
    na = lambda v: "." if v == -1 else str(v)
    return [str(m.kf), str(m.hf), str(m.bf), str(m.qf), str(m.af), str(m.rm), "0", "0", na(m.si), na(m.nt), na(m.bs), na(m.ti)]
