"""The rule by which the fused probe kernels decide a pair with k-mers shared between loci WITHOUT the vote (dbtk_locus.h, dbtk_probe2.h):
if every found k-mer names locus L (the unique ones by their index value, the shared ones by having L in their list), at least one is
unique to L, and each mate has >= cth found positions, then find_matching_locus + countHit (src/aQueryFasta_thread.cpp:331-453) end with
top.idx == L and accept — whatever order std::sort left the k-mers of equal list length in.  Checked here on a literal model of those
functions over random instances and random tie orders (CPU; the kernels themselves are checked against the oracle by the emulator and GPU
suites)."""
import numpy as np


def vote(entries, cth):
    """entries: list of (loci list, d1, d2) in visiting order.  Returns (top idx, fc, rc).  AQ.cpp:331-422 literally."""
    NAN = -1
    top, second = [NAN, 0, 0], [NAN, 0, 0]
    h1, h2 = {}, {}
    n = len(entries)
    dupsum = sum(d1 + d2 for _, d1, d2 in entries)
    remain = [0] * n
    remain[0] = dupsum - entries[0][1] - entries[0][2]
    for i in range(1, n - 1):
        remain[i] = remain[i - 1] - entries[i][1] - entries[i][2]

    def updatetop2(cf, ind, cr):
        if cf + cr > top[1] + top[2]:
            if top[0] != ind:
                second[:] = top
                top[0] = ind
            top[1], top[2] = cf, cr
        elif cf + cr > second[1] + second[2]:
            if second[0] != ind:
                second[0] = ind
            second[1], second[2] = cf, cr

    for i, (loci, d1, d2) in enumerate(entries):
        for l in loci:
            h1[l] = h1.get(l, 0) + d1
            h2[l] = h2.get(l, 0) + d2
            updatetop2(h1[l], l, h2[l])
        if not ((top[1] + top[2] - second[1] - second[2]) < remain[i]):
            j = i
            while (top[1] < cth and cth - top[1] <= remain[j]) or (top[2] < cth and cth - top[2] <= remain[j]):
                j += 1
                if j >= n:
                    break
                lj, e1, e2 = entries[j]
                if top[0] in lj:
                    top[1] += e1
                    top[2] += e2
            break
    return tuple(top)


def test_order_cannot_matter_when_the_rule_holds():
    rng = np.random.default_rng(20251004)
    tried = 0
    for it in range(4000):
        cth = int(rng.integers(1, 60))
        L = int(rng.integers(0, 8))
        nu = int(rng.integers(1, 4)) if rng.random() < 0.5 else int(rng.integers(1, 120))  # k-mers unique to L (often very few)
        ns = int(rng.integers(0, 200))                                                      # shared ones, L somewhere in their lists
        ent = []
        for _ in range(nu):
            d = [(1, 0), (0, 1), (1, 1), (2, 0), (0, 3)][int(rng.integers(0, 5))]
            ent.append(([L], d[0], d[1]))
        for _ in range(ns):
            others = [int(x) for x in rng.permutation([l for l in range(8) if l != L])[:int(rng.integers(1, 6))]]
            lst = others[:]
            lst.insert(int(rng.integers(0, len(lst) + 1)), L)
            d = [(1, 0), (0, 1), (1, 1), (5, 0), (0, 7), (9, 9)][int(rng.integers(0, 6))]
            ent.append((lst, d[0], d[1]))
        D1, D2 = sum(e[1] for e in ent), sum(e[2] for e in ent)
        if D1 < cth or D2 < cth:
            continue
        tried += 1
        for _ in range(6):  # random order among equal list lengths (what std::sort leaves unspecified), ascending by length
            keys = np.array([len(e[0]) for e in ent], float) + rng.random(len(ent)) * 0.5
            order = np.argsort(keys, kind="stable")
            idx, fc, rc = vote([ent[i] for i in order], cth)
            assert idx == L and fc >= cth and rc >= cth, (it, cth, L, nu, ns, idx, fc, rc)
    assert tried > 1000


def test_one_kept_mate_below_two_cth_cannot_be_accepted():
    """The other short cut of the lean probe kernel (dbtk_probe2.h: p2_resolve_gone with one mate cleared by kfilter): the vote runs on the
    kept mate's list alone, so one strand of `top` stays 0 (test1 fails) and top.fc + top.rc is at most the kept mate's found positions
    (test2 fails when they are fewer than 2 cth) — for ANY lists, any order."""
    rng = np.random.default_rng(7)
    for it in range(3000):
        cth = int(rng.integers(1, 60))
        strand = int(rng.integers(0, 2))
        ent, D = [], 0
        while True:
            d = int(rng.integers(1, 4))
            if D + d >= 2 * cth:
                break
            lst = [int(x) for x in rng.permutation(8)[:int(rng.integers(1, 6))]]
            ent.append((lst, d if strand == 0 else 0, 0 if strand == 0 else d))
            D += d
            if rng.random() < 0.02:
                break
        if not ent:
            continue
        keys = np.array([len(e[0]) for e in ent], float) + rng.random(len(ent)) * 0.5
        idx, fc, rc = vote([ent[i] for i in np.argsort(keys, kind="stable")], cth)
        assert not ((fc >= cth and rc >= cth) or fc + rc >= 2 * cth), (it, cth, D, fc, rc)


def test_the_rule_needs_its_premises():
    """Without a k-mer unique to L another locus of the lists may lead: the rule does not apply (the kernels send such a pair to the vote)."""
    ent = [([1, 0], 1, 1)] * 50  # every k-mer shared, locus 1 first in the lists
    idx, fc, rc = vote(ent, 10)
    assert idx == 1
