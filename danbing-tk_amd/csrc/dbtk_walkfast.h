// dbtk_walkfast.h — the lean first kernel of the pair-mode graph walk (split off dbtk_walk.h: it shares the probe kernel's bucket
// machinery, dbtk_probe2.h, and is included behind it by dbtk_kernels.h).
#ifndef DBTK_WALKFAST_H_
#define DBTK_WALKFAST_H_

namespace dbtk {

// The minimizer-grouped copy of the graph table: every (canonical k-mer, locus) entry of the hashed table (GrSlot) in the 128-byte bucket
// of the k-mer's minimizer, key = the k-mer, payload = {locus, info}; a k-mer that is a node at several loci has several entries.  Keys a
// full bucket turns away (MZ_TURNED on key 7) are found in the hashed table, which stays.
struct GrMzBuildArgs {
    const GrSlot* gr;
    uint64_t nslots;
    MzBucket* mz;
    uint32_t mask, ksize, m;
};
template <class X>
DBTK_HD void body_grmz_insert(X& x, const GrMzBuildArgs& a) {
    for (uint64_t i = (uint64_t)x.bid() * x.nthreads() + x.tid(); i < a.nslots; i += (uint64_t)x.nblocks() * x.nthreads()) {
        const GrSlot s = a.gr[i];
        if (s.kmer == NAN64) continue;
        MzBucket* b = a.mz + mz_bucket(mz_of_kmer(s.kmer, a.ksize, a.m), a.mask);
        bool placed = false;
        for (int j = 0; j < 8 && !placed; ++j)
            if (x.atomic_cas(&b->key[j], MZ_EMPTY, s.kmer) == MZ_EMPTY) { b->pl[j].val = (uint32_t)(s.li >> 32); b->pl[j].aux = (uint32_t)s.li; placed = true; }
        if (!placed) x.atomic_or(&b->key[7], MZ_TURNED);
    }
}

// ---- pair mode, first kernel: the pairs one of whose mates threads through the graph as it stands.
// isThreadFeasible on a read whose first k-mer is a node and whose every next k-mer is an out-edge of the one before returns 1
// without skipping or correcting anything (walk_read's ballot loop above; AQ.cpp:1114-1260, 1259) — most reads.  The v1.3 call
// site keeps a pair when EITHER mate is feasible and then counts the uncorrected k-mers of BOTH (AQ.cpp:2082-2087, 2189-2194):
// so as soon as one mate threads cleanly the pair is decided and counted here, whatever the walk of the other mate would find
// (its return code is reported as WALK_NOT_EVALUATED).  Pairs with no such mate — and everything when alignment records or
// thread records are wanted — go to body_walk_pairs, the kernel that carries the error-correction machinery (and its 200+
// registers); this one is the probe kernel's shape: one wave per pair, one mate per half-wave, NPL consecutive positions per
// lane as shifts of one 32-base word, one 16-byte graph-table load per position, the feasibility of a mate one ballot.
#ifndef DBTK_WF_R
#define DBTK_WF_R 2
#endif
constexpr uint32_t WF_R = DBTK_WF_R;  // places of the passed-on list a wave reserves at a time (what it does not use stays WALK_NO_ENTRY, an idle turn of the other kernel: 8 at a time cost that one 1 ms per 10 M reads, 1 at a time costs this one its atomics)
// NPL of the lean kernel for a batch whose longest read has max_read_len bases (0: the kernel does not apply).  A half-wave's lanes hold the
// k-mers of positions 0 .. 32 NPL - 1 — and, with the minimizer-grouped table, only the m-mers of those base positions: the window of the last
// k-mer reaches m-mer len - m, so the bound is 32 NPL + m - 1 there (the probe kernel's), not 32 NPL + k - 1 (longer reads took the
// other mate's m-mers for their tail windows and lost TR k-mers without an error).  One definition for launch_batch and the emulator.
DBTK_HD int walkfast_npl(uint32_t max_read_len, uint32_t k, bool grmz) {
    if (k + 4 > 32) return 0;
    const uint32_t ext = grmz ? mz_m_for_k(k) - 1 : k - 1;
    if (max_read_len <= 32 * 3 + ext && k + 2 <= 32) return 3;
    if (max_read_len <= 32 * 5 + ext) return 5;
    return 0;
}
template <int NPL>
struct __attribute__((aligned(16))) WalkFastSmemT {
    uint32_t pk[2][20];       // 2-bit stream of each mate from its 4-byte-aligned start
    uint32_t rb[64 * NPL];    // bucket of every run of the pair (the minimizer-grouped copy of the graph table)
    uint4 stg[P2_RCH][P2_ROW];  // the buckets of a chunk of runs
    uint4 cache[P2_CACHE];    // {k-mer, info, locus} of single look-ups already made (info 0: no node): the wave works through the reads of a locus
    uint8_t tok[P2_CACHE];    // which lane writes an entry when several want to in one step
};
// The verdict on one pair from the graph nodes of its positions (gi[j]: info of the canonical k-mer at the pair's locus, 0: no node),
// shared by the two forms of the lean kernel (nodes from the global tables / from the locus' image in LDS): the step test, the exact
// counting of a kept pair, its text record, or the hand-over to the other kernel.
struct WfState {
    uint64_t c_feas = 0, c_inc = 0;
    uint32_t rbase = 0, rleft = 0;  // places of the passed-on list this wave has reserved and not used yet
    uint32_t txt_base = 0, txt_left = 0;  // text records (-a / -ae): arena bytes are taken TXT_CHUNK at a time, as in body_walk_pairs
};
// A pair goes on to body_walk_pairs: its place in the list into the next reserved entry of the passed-on list — and, when the graph
// info of all its positions is known (gi: this lane's NPL positions), that too: row e of slow_info for entry e, so that the other
// kernel starts from what this one has looked up (260 graph-table probe sequences per pair otherwise, a quarter of its time).
template <int NPL, class X>
DBTK_HD void wf_hand_over(X& x, const WalkArgs& a, WfState& S, uint32_t i, const uint32_t* gi) {
    const int lane = x.lane();
    if (!S.rleft) {
        uint32_t b = 0;
        if (lane == 0) b = x.atomic_add(a.nslow, WF_R);
        S.rbase = x.bcast(b, 0);
        S.rleft = WF_R;
    }
    const uint32_t e = S.rbase + (WF_R - S.rleft);
    --S.rleft;
    const bool info = gi && a.slow_info && e < a.info_cap && a.info_stride == 32u * NPL;
    if (lane == 0) a.slow_list[e] = i | (info ? WALK_HAS_INFO : 0u);
    if (info) {
        uint32_t* row = a.slow_info + ((size_t)e * 2 + ((uint32_t)lane >> 5)) * (32u * NPL) + ((uint32_t)lane & 31u) * NPL;
#pragma unroll
        for (int j = 0; j < NPL; ++j) row[j] = gi[j];
    }
}
template <class X>
DBTK_HD void wf_finish(X& x, const WalkArgs& a, WfState& S) {  // the reserved places nothing went into
    if ((uint32_t)x.lane() < S.rleft) a.slow_list[S.rbase + (WF_R - S.rleft) + (uint32_t)x.lane()] = WALK_NO_ENTRY;
    S.rleft = 0;
}
template <int NPL, class X>
DBTK_HD void wf_decide(X& x, const WalkArgs& a, uint32_t i, uint32_t dst, uint32_t len, uint32_t nk, uint64_t badm, const uint64_t (&fw)[NPL],
                       const uint64_t (&cn)[NPL], const uint32_t (&gi)[NPL], const bool (&act)[NPL], WfState& S, bool texting,
                       uint32_t* lcnt = nullptr, uint32_t lcap = 0) {
    const int lane = x.lane();
    const uint32_t hl = (uint32_t)lane & 31u, half = (uint32_t)lane >> 5, p0 = hl * NPL;
    const DevTables& T = a.T;
    uint64_t& c_feas = S.c_feas; uint64_t& c_inc = S.c_inc;
    uint32_t& txt_base = S.txt_base; uint32_t& txt_left = S.txt_left;
    // oriented info of every position (as w_info), then the step test of walk_read: position p continues the walk iff the
    // k-mer before it is a node with an out-edge labelled by p's last base (and is not p's k-mer itself: a homopolymer)
    uint32_t go[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const bool isf = fw[j] == cn[j];
        const uint32_t fa = gi[j] & 0x1Fu, fb = (gi[j] >> GR_OPP) & 0x1Fu;
        go[j] = isf ? fa : fb;
    }
    uint32_t pg = x.shfl_up1(go[NPL - 1]);
    uint32_t plo = x.shfl_up1((uint32_t)fw[NPL - 1]), phi = x.shfl_up1((uint32_t)(fw[NPL - 1] >> 32));
    bool fail = false;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const uint64_t pv = ((uint64_t)phi << 32) | plo;
        if (act[j]) {
            if (p0 + j == 0) fail |= !(go[j] & GR_HAS);  // the anchor is the first k-mer
            else fail |= !((pg & GR_HAS) && ((pg >> (uint32_t)(fw[j] & 3)) & 1) && fw[j] != pv);
        }
        pg = go[j]; plo = (uint32_t)fw[j]; phi = (uint32_t)(fw[j] >> 32);
    }
    const uint64_t failm = x.ballot(fail) | badm;
    const uint64_t nkm = x.ballot(nk > 0);
    const bool clean0 = (nkm & 1) && !(failm & 0xFFFFFFFFull), clean1 = ((nkm >> 32) & 1) && !(failm >> 32);
    // (-a / -ae: the record holds both mates' alignments, so only a pair BOTH of whose mates thread cleanly is finished here — its
    // strings are "len=" and the run lengths of the TR flags of its k-mers; any other pair goes on to the kernel that aligns)
    if ((texting ? (clean0 && clean1) : (clean0 || clean1)) && !badm) {  // (uniform) the pair is kept: count the uncorrected k-mers of both mates (AQ.cpp:2189-2194)
        // (a pair with a non-ACGT byte in either mate is passed on even when its other mate threads: the valid k-mers of the
        // mate with the N count too, and which of its windows are valid is the other kernel's business)
        c_feas += 2;
        const uint32_t tb = T.trbeg[dst];
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const bool hit = act[j] && (gi[j] & GR_TR);
            // (the locus-resident form counts into the workgroup's LDS copy of the locus' counters, flushed once per item)
            if (hit) { const uint32_t sl = gi[j] >> GR_SLOT_SHIFT; if (sl < lcap) x.lds_add(&lcnt[sl], 1u); else x.atomic_add(&a.counts[tb + sl], 1ull); }
            c_inc += (uint64_t)__builtin_popcountll(x.ballot(hit));
        }
        if (lane == 0) {
            a.walk_dst[i] = dst;
            a.walk_ret[i] = (uint32_t)(uint8_t)(clean0 ? 1 : WALK_NOT_EVALUATED) | ((uint32_t)(uint8_t)(clean1 ? 1 : WALK_NOT_EVALUATED) << 8);
        }
        if (texting) {
            // writeAnnot on cg.tr of a clean walk (one '=' / '.' per k-mer: is it a TR k-mer of the locus) = the run lengths of the
            // flags; writeCigar on its cg.es (len matches) = "len=".  A run is printed by the lane of its last position: the flag of
            // the position after it comes from the lane's own next position or, across lanes, from the ballot of first positions;
            // where the run began is a max-scan over the positions where the flag changes (each mate's own scale, so that mate 1's
            // scan never sees mate 0's).
            bool f[NPL];
#pragma unroll
            for (int j = 0; j < NPL; ++j) f[j] = (gi[j] & GR_TR) != 0;
            const uint64_t f0m = x.ballot(f[0]);
            const bool fnext = lane < 63 && ((f0m >> (lane + 1)) & 1);  // (lane 31's successor is mate 1's first: never looked at, p0 + NPL - 1 >= nk - 1 there or inactive)
            const uint32_t scale = half ? 4096u : 0u;
            uint32_t chg = 0;  // (index + 1 on the mate's scale) of the last position of this lane that starts a run
            bool fp = x.shfl_up1(f[NPL - 1] ? 1u : 0u) != 0;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                if (act[j] && (p0 + j == 0 || f[j] != fp)) chg = scale + p0 + j + 1;
                fp = f[j];
            }
            const uint32_t inc = x.wave_scan_max(chg);
            uint32_t before = x.shfl_up1(inc);  // the last run start in the lanes before this one
            if (lane == 0 || before < scale + 1) before = scale + 1;  // (a mate's position 0 always starts a run; lane 32 must not see mate 0)
            uint32_t rl[NPL], ol[NPL], tot = 0;
            uint32_t st = before;
            fp = x.shfl_up1(f[NPL - 1] ? 1u : 0u) != 0;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                if (act[j] && (p0 + j == 0 || f[j] != fp)) st = scale + p0 + j + 1;
                fp = f[j];
                const bool nx = j + 1 < NPL ? f[j + 1] : fnext;
                const bool end = act[j] && (p0 + j + 1 == nk || nx != f[j]);
                rl[j] = end ? (scale + p0 + j + 1) - st + 1 : 0u;
                ol[j] = end ? (rl[j] >= 100 ? 3u : rl[j] >= 10 ? 2u : 1u) + 1u : 0u;
                tot += ol[j];
            }
            const uint32_t ex = x.wave_excl_scan(tot), la = x.half_sum(tot);
            const uint32_t la0 = x.bcast(la, 0), la1 = x.bcast(la, 32);
            const uint32_t len0 = x.bcast(len, 0), len1 = x.bcast(len, 32);  // (each half holds its own mate's length)
            const uint32_t lc0 = (len0 >= 100 ? 3u : len0 >= 10 ? 2u : 1u) + 1u, lc1 = (len1 >= 100 ? 3u : len1 >= 10 ? 2u : 1u) + 1u;
            const uint32_t tlen = lc1 + 1 + la1 + 1 + lc0 + 1 + la0, need = (8 + tlen + 3) & ~3u;
            if (need > txt_left) {
                uint32_t b = 0;
                if (lane == 0) b = txt_carve(x, a);
                txt_base = x.bcast(b, 0);
                txt_left = TXT_CHUNK;
            }
            if ((uint64_t)txt_base + need <= a.txt_cap) {
                uint8_t* r = a.txt + txt_base;
                if (lane == 0) {  // header, mate 1's CIGAR, the three tabs, mate 0's CIGAR
                    reinterpret_cast<uint32_t*>(r)[0] = dst;
                    reinterpret_cast<uint32_t*>(r)[1] = tlen;
                    a.txt_idx[a.surv[i]] = txt_base;
                    uint32_t q = w_fmt_int(r, 8, (int)len1);
                    r[q] = '='; r[q + 1] = '\t';
                    r[8 + lc1 + 1 + la1] = '\t';
                    q = w_fmt_int(r, 8 + lc1 + 1 + la1 + 1, (int)len0);
                    r[q] = '='; r[q + 1] = '\t';
                }
                uint32_t o = half ? 8 + lc1 + 1 + (ex - la0) : 8 + lc1 + 1 + la1 + 1 + lc0 + 1 + ex;
#pragma unroll
                for (int j = 0; j < NPL; ++j)
                    if (ol[j]) { const uint32_t q = w_fmt_int(r, o, (int)rl[j]); r[q] = f[j] ? '=' : '.'; o += ol[j]; }
            } else if (lane == 0 && a.errflag) *a.errflag = DBTK_ERR_OVERFLOW;
            txt_base += need; txt_left -= need;
        }
    } else if (lcnt && a.pend_locus) {
        // the locus-resident form: the pair stays with its locus — body_walk_pairs_locus (dbtk_walkfast.h, below) walks it with the error
        // correction's graph look-ups answered from the same image in LDS
        if (lane == 0) a.walk_ret[i] = WALK_PENDING;
    } else {
        // the row the error-correcting kernel gets: the info ORIENTED as the read has the k-mer (bits 0-9), so that it need not
        // reverse-complement 260 k-mers again to find out (round 6: two revcomp2 per position were a twentieth of its instructions)
        uint32_t og[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const bool isf = fw[j] == cn[j];
            const uint32_t fa = gi[j] & 0x1Fu, fb = (gi[j] >> GR_OPP) & 0x1Fu;
            og[j] = (isf ? (fa | (fb << GR_OPP)) : (fb | (fa << GR_OPP))) | (gi[j] & ~0x3FFu);
        }
        wf_hand_over<NPL>(x, a, S, i, badm ? nullptr : og);
    }
}

// WN = k - m + 1 m-mers per window when the minimizer-grouped copy of the graph table exists (T.grmz), else unused
template <int NPL, int WN, class X>
DBTK_HD void body_walk_fast(X& x, const WalkArgs& a) {
    typedef WalkFastSmemT<NPL> SM;
    SM& sm = *x.template smem<SM>();
    const int lane = x.lane();
    const uint32_t hl = (uint32_t)lane & 31u, half = (uint32_t)lane >> 5;
    const DevTables& T = a.T;
    const uint32_t k = T.ksize;
    uint64_t* const ctr = a.ctr_rep ? a.ctr_rep + (size_t)(x.bid() & (W_CTR_REP - 1)) * W_CTR_STRIDE : a.counters;
    const uint32_t nsurv = a.sel ? *a.nsel : *a.nsurv;  // pairs this kernel takes: the whole list, or the places a.sel names
    if (a.sel && a.pstats && x.bid() == 0 && lane == 0 && nsurv) x.atomic_add(&a.pstats[13], (uint64_t)nsurv);
    // a contiguous range of the (locus-ordered) survivor list per wave: the waves running side by side count into different loci
    const uint32_t per = (nsurv + x.nblocks() - 1) / x.nblocks();
    const uint64_t lo64 = (uint64_t)x.bid() * per;
    const uint32_t first = lo64 < nsurv ? (uint32_t)lo64 : nsurv, hi = lo64 + per < nsurv ? (uint32_t)(lo64 + per) : nsurv;
    const uint32_t lmax = 32u * NPL + (T.grmz ? mz_m_for_k(k) : k) - 1;  // (walkfast_npl: the launcher promised no read is longer)
    const uint64_t kmask = (1ull << (2 * k)) - 1;
    const uint32_t p0 = hl * NPL;
    for (uint32_t e = (uint32_t)lane; e < (uint32_t)P2_CACHE; e += 64) sm.cache[e] = uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0xFFFFFFFFu};
    WfState S;
    const bool texting = a.txt && (a.P.aln & 3) != 0;
    // three-deep fetch pipeline as in the probe kernel: bytes of pair i + 1, offsets of pair i + 2, (pair, locus) of pair i + 3; the
    // place of pair i + 4 (i + 4 itself, or what a.sel says) is read an iteration before its list entries
    auto clampi = [&](uint32_t i) { return i < hi ? i : (first < hi ? first : 0u); };
    auto place_of = [&](uint32_t i) -> uint32_t { const uint32_t ic = clampi(i); return a.sel ? (nsurv ? a.sel[ic] : 0u) : ic; };
    uint32_t rw0 = 0, rw1 = 0;
    uint64_t o0C = 0, o1C = 0, o0B = 0, o1B = 0;
    uint32_t pairA = 0, dstA = NAN32, dstB = NAN32, dstC = NAN32, plN = 0;
    auto fetch_bytes = [&](uint64_t o0, uint64_t o1) {
        uint32_t len = (uint32_t)(o1 - o0);
        if (len > lmax) len = lmax;
        const uint64_t a0 = o0 & ~3ull;
        const uint32_t nw = ((uint32_t)(o0 - a0) + len + 3) >> 2;
        rw0 = *reinterpret_cast<const uint32_t*>(a.seq + (2 * hl < nw ? a0 + 8ull * hl : 0ull));
        rw1 = *reinterpret_cast<const uint32_t*>(a.seq + (2 * hl + 1 < nw ? a0 + 8ull * hl + 4 : 0ull));
    };
    auto fetch_offsets = [&](uint32_t pair) {
        const uint64_t r = 2 * (uint64_t)pair + half;
        o0B = a.off[r]; o1B = a.off[r + 1];
    };
    if (first < hi) {
        const uint32_t q0 = place_of(first), q1 = place_of(first + 1), q2 = place_of(first + 2);
        fetch_offsets(x.uni(a.surv[q0]));
        dstC = a.walk_dst[q0];
        o0C = o0B; o1C = o1B;
        fetch_bytes(o0C, o1C);
        if (first + 1 < hi) { fetch_offsets(x.uni(a.surv[q1])); dstB = a.walk_dst[q1]; }
        pairA = a.surv[q2]; dstA = a.walk_dst[q2];
        plN = place_of(first + 3);
    }
    for (uint32_t ii = first; ii < hi; ++ii) {
        const uint64_t o0 = o0C, o1 = o1C;
        uint32_t len = (uint32_t)(o1 - o0);
        if (len > lmax) { if (a.errflag) *a.errflag = DBTK_ERR_READ_TOO_LONG; len = lmax; }
        const uint64_t a0 = o0 & ~3ull;
        const uint32_t rsh = (uint32_t)(o0 - a0);
        const uint32_t d0 = rw0, d1 = rw1;
        const uint32_t dst = x.uni(dstC);
        const uint32_t i = x.uni(place_of(ii));  // the pair's place in the list (read again here: cheaper than a register through the pipeline)
        {   // advance the pipeline
            const bool hasB = ii + 1 < hi, hasA = ii + 2 < hi;
            o0C = hasB ? o0B : 0ull; o1C = hasB ? o1B : 0ull;
            fetch_bytes(o0C, o1C);
            dstC = dstB; dstB = dstA;
            fetch_offsets(hasA ? x.uni(pairA) : x.uni(pairA) * 0u);
            pairA = a.surv[plN]; dstA = a.walk_dst[plN];
            plN = place_of(ii + 4);
        }
        if (dst == NAN32) continue;  // the pair never reached threading (uniform)
        x.sync();
        uint32_t bad = 0;
        {
            const uint32_t c0 = pack4_b2(d0, &bad), c1 = pack4_b2(d1, &bad);
            reinterpret_cast<uint16_t*>(sm.pk[half])[hl ^ 1u] = (uint16_t)(((c0 >> 8) & 0xFF00u) | ((c1 >> 16) & 0xFFu));
        }
        const uint32_t nk = len >= k ? len - k + 1 : 0;
        // a mate with a non-ACGT byte does not thread cleanly (a k-mer with an N is no node); bytes of the lane's dwords outside
        // the read are the neighbouring reads': counting them only sends a mate to the other kernel for nothing
        const uint64_t badm = x.ballot(bad != 0 && 8 * hl < rsh + len);
        x.sync();
        const uint64_t W = window_fw_clean(sm.pk[half], rsh + p0, 32);
        const uint64_t RW = revcomp2(W, 32);
        uint64_t fw[NPL], cn[NPL];
        uint32_t gi[NPL];
        uint64_t at[NPL];
        bool act[NPL], open[NPL];
        uint4 q[NPL];
        // The graph nodes of every position.  One 16-byte request per position into the hashed graph table was what this kernel cost
        // (260 per pair); with the minimizer-grouped copy of the table (T.grmz: the probe kernel's remedy, dbtk_probe2.h) the positions
        // of a run share a 128-byte bucket that 8 lanes fetch once, and only what a full bucket turned away is still looked up singly.
        auto single = [&]() {  // open[j]: (canonical k-mer, locus) in the hashed table
            bool anyo = false;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                if (open[j]) { at[j] = hash_cls(cn[j], dst, T.gr_shift); q[j] = reinterpret_cast<const uint4*>(T.gr)[at[j]]; }
                anyo |= open[j];
            }
            while (x.ballot(anyo)) {  // (a look-up rarely needs a second slot)
                anyo = false;
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    if (!open[j]) continue;
                    if (q[j].x == (uint32_t)cn[j] && q[j].y == (uint32_t)(cn[j] >> 32) && q[j].w == dst) { gi[j] = q[j].z; open[j] = false; }
                    else if ((q[j].x & q[j].y) == 0xFFFFFFFFu) open[j] = false;  // empty slot: not in the table
                    else { at[j] = (at[j] + 1) & T.gr_mask; q[j] = reinterpret_cast<const uint4*>(T.gr)[at[j]]; }
                    anyo |= open[j];
                }
            }
        };
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            fw[j] = (W >> (2 * (32 - k - j))) & kmask;
            const uint64_t rc = (RW >> (2 * j)) & kmask;
            cn[j] = fw[j] <= rc ? fw[j] : rc;
            act[j] = p0 + j < nk;
            at[j] = 0; gi[j] = 0; open[j] = false;
            q[j] = uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u};
        }
        if (badm) {  // (uniform) a pair with a non-ACGT byte is the other kernel's whatever its mates do
            wf_hand_over<NPL>(x, a, S, i, nullptr);
            continue;
        }
        if (T.grmz) {
            const uint32_t m = mz_m_for_k(k), mmask = (uint32_t)((1ull << (2 * m)) - 1), nmm = len >= m ? len - m + 1 : 0;
            uint32_t f[NPL + WN - 1], mz[NPL], bk[NPL], rid[NPL];
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const uint32_t fwm = (uint32_t)(W >> (2 * (32 - m - j))) & mmask, rcm = (uint32_t)(RW >> (2 * j)) & mmask;
                f[j] = p0 + j < nmm ? mmer_hash2(fwm, rcm) : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int t = NPL; t < NPL + WN - 1; ++t) f[t] = x.shfl_down1(f[t - NPL]);
            sliding_min<NPL, WN>(f, mz);
            uint32_t cnt = 0;
            bool st[NPL];
#pragma unroll
            for (int j = 0; j < NPL; ++j) bk[j] = act[j] ? mz_bucket(mz[j] >> 4, (uint32_t)T.grmz_mask) : 0xFFFFFFFFu;
            uint32_t prev = x.shfl_up1(bk[NPL - 1]);
            if (hl == 0) prev = 0xFFFFFFFEu;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                st[j] = act[j] && bk[j] != prev;
                cnt += st[j] ? 1u : 0u;
                prev = bk[j];
            }
            uint32_t r = x.wave_excl_scan(cnt);
            const uint32_t nruns = x.bcast(r + cnt, 63);
            x.sync();
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                if (st[j]) { sm.rb[r] = bk[j]; ++r; }
                rid[j] = r - 1;
            }
            x.sync();
            const uint32_t fq8 = (uint32_t)lane >> 3, part = (uint32_t)lane & 7u;
            p2_v4u qq[P2_RCH / 8];
            p2_fetch_runs(sm.rb, T.grmz, nruns, 0u, fq8, part, 0xFFFFFFFFu, qq);
            for (uint32_t r0 = 0; r0 < nruns; r0 += P2_RCH) {
                x.sync();
#pragma unroll
                for (int u = 0; u < P2_RCH / 8; ++u) *reinterpret_cast<p2_v4u*>(&sm.stg[8 * u + fq8][part]) = qq[u];
                if (r0 + P2_RCH < nruns) p2_fetch_runs(sm.rb, T.grmz, nruns, r0 + P2_RCH, fq8, part, 0xFFFFFFFFu, qq);
                x.sync();
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    const uint32_t row = rid[j] - r0;
                    if (act[j] && row < (uint32_t)P2_RCH) {
                        const uint4* rp = sm.stg[row];
                        uint32_t mm = 0, w7 = 0;  // slots whose key is this k-mer (one per locus that has the node)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const uint4 kk = rp[g];
                            const uint64_t k0 = ((uint64_t)kk.y << 32) | kk.x, k1 = ((uint64_t)kk.w << 32) | kk.z;
                            if (k0 == cn[j]) mm |= 1u << (2 * g);
                            if ((g == 3 ? (k1 & ~MZ_TURNED) : k1) == cn[j]) mm |= 1u << (2 * g + 1);
                            if (g == 3) w7 = kk.w;
                        }
                        bool found = false;
                        while (mm) {
                            const uint32_t sl = (uint32_t)__builtin_ctz(mm);
                            mm &= mm - 1;
                            const uint32_t* pl = reinterpret_cast<const uint32_t*>(rp + 4) + 2 * sl;
                            if (pl[0] == dst) { gi[j] = pl[1]; found = true; }
                        }
                        open[j] = !found && (w7 >> 31) != 0;  // the bucket turned keys away: ask the hashed table
                    }
                }
            }
            // what the buckets turned away: the wave's cache first (the k-mers of a tandem repeat come back all along a read and in
            // every read of its locus), then the hashed table; what was looked up goes into the cache (the absent ones too)
            bool anyp = false;
#pragma unroll
            for (int j = 0; j < NPL; ++j) anyp |= open[j];
            if (x.ballot(anyp)) {
                uint32_t ce[NPL];
                bool glob[NPL];
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    ce[j] = (ovf_hash(cn[j]) >> 20) & (P2_CACHE - 1);
                    glob[j] = false;
                    if (open[j]) {
                        const uint4 c = sm.cache[ce[j]];
                        if (c.x == (uint32_t)cn[j] && c.y == (uint32_t)(cn[j] >> 32) && c.w == dst) { gi[j] = c.z; open[j] = false; }
                        else glob[j] = true;
                    }
                }
                single();
#pragma unroll
                for (int j = 0; j < NPL; ++j) if (glob[j]) sm.tok[ce[j]] = (uint8_t)lane;
                x.sync();
#pragma unroll
                for (int j = 0; j < NPL; ++j)
                    if (glob[j] && sm.tok[ce[j]] == (uint8_t)lane) sm.cache[ce[j]] = uint4{(uint32_t)cn[j], (uint32_t)(cn[j] >> 32), gi[j], dst};
                x.sync();
            }
        } else {
#pragma unroll
            for (int j = 0; j < NPL; ++j) open[j] = act[j];
            single();
        }
        wf_decide<NPL>(x, a, i, dst, len, nk, badm, fw, cn, gi, act, S, texting);
    }
    wf_finish(x, a, S);
    if (lane == 0) {
        if (S.c_feas) x.atomic_add(&ctr[DBTK_C_FEASIBLE], S.c_feas);
        if (S.c_inc) x.atomic_add(&ctr[DBTK_C_ALGO_INC], S.c_inc);
    }
}

// ---- the lean kernel with the locus' graph nodes resident in LDS (the probe kernel's remedy, dbtk_locus.h): a workgroup takes ITEMS
// = (locus, up to LOC_CH consecutive pairs of its segment of the list), copies the locus' graph image into LDS and its waves get the
// node of every position from there: one LDS bucket per position instead of a 128-byte line per run of positions plus the single
// look-ups.  A pair whose destLocus is not the item's locus (the list is in the order of the pairs' first index hit, the vote may have
// chosen another locus) goes to the other kernel like every pair this one cannot decide.
template <int NPL, int NW>
struct __attribute__((aligned(16))) WalkFastLocWaveSmemT {
    uint32_t pk[2][20];
};
constexpr uint32_t WFL_CNT = 2048;  // counters of the item's locus kept in LDS (a locus with more TR k-mers counts the rest directly)
template <int NPL, int NW, int IMGB>
struct __attribute__((aligned(16))) WalkFastLocSmemT {
    uint4 img[IMGB / 16];
    uint32_t cnt[WFL_CNT];
    WalkFastLocWaveSmemT<NPL, NW> w[NW];
};
template <int NPL, int NW, int IMGB, class X>
DBTK_HD void body_walk_fast_locus(X& x, const WalkArgs& a, const LocRunArgs& r) {
    typedef WalkFastLocSmemT<NPL, NW, IMGB> SM;
    constexpr int IPT = ((IMGB - (int)LOC_HDR) / 16 + NW * 64 - 1) / (NW * 64);
    SM& smb = *x.template smem<SM>();
    const int lane = x.lane();
    const uint32_t wave = (uint32_t)x.tid() >> 6;
    WalkFastLocWaveSmemT<NPL, NW>& sm = smb.w[wave];
    const uint32_t* bks = reinterpret_cast<const uint32_t*>(smb.img);
    const uint32_t hl = (uint32_t)lane & 31u, half = (uint32_t)lane >> 5;
    const DevTables& T = a.T;
    const uint32_t k = T.ksize;
    uint64_t* const ctr = a.ctr_rep ? a.ctr_rep + (size_t)(x.bid() & (W_CTR_REP - 1)) * W_CTR_STRIDE : a.counters;
    const uint32_t ifirst = r.starts[x.bid()], nitems = r.starts[x.bid() + 1];  // this workgroup's items: [ifirst, nitems) (body_loc_split)
    const uint32_t lmax = 32u * NPL + k - 1;
    const uint64_t kmask = (1ull << (2 * k)) - 1;
    const uint32_t p0 = hl * NPL;
    constexpr uint32_t S = 1;
    WfState W;
    const bool texting = a.txt && (a.P.aln & 3) != 0;
    // this wave's pairs, item after item (as body_probe_locus): the fetch pipeline runs along that sequence across items
    struct Cur { uint32_t it; uint32_t i, end; };
    auto desc = [&](uint32_t it) -> uint4 { return r.items[it < nitems ? it : 0u]; };
    auto seek = [&](uint32_t it) -> Cur {
        for (;;) {
            if (it >= nitems) return Cur{it, 0u, 0u};
            const uint4 d = desc(it);
            if (d.y + wave < d.z) return Cur{it, d.y + wave, d.z};
            it += S;
        }
    };
    auto next = [&](const Cur& c) -> Cur {
        if (c.it >= nitems) return c;
        if (c.i + NW < c.end) return Cur{c.it, c.i + NW, c.end};
        return seek(c.it + S);
    };
    uint32_t rw0 = 0, rw1 = 0;
    uint64_t o0C = 0, o1C = 0, o0B = 0, o1B = 0;
    uint32_t pairA = 0, dstA = NAN32, dstB = NAN32, dstC = NAN32;
    auto fetch_bytes = [&](uint64_t o0, uint64_t o1) {
        uint32_t len = (uint32_t)(o1 - o0);
        if (len > lmax) len = lmax;
        const uint64_t a0 = o0 & ~3ull;
        const uint32_t nw = ((uint32_t)(o0 - a0) + len + 3) >> 2;
        rw0 = *reinterpret_cast<const uint32_t*>(a.seq + (2 * hl < nw ? a0 + 8ull * hl : 0ull));
        rw1 = *reinterpret_cast<const uint32_t*>(a.seq + (2 * hl + 1 < nw ? a0 + 8ull * hl + 4 : 0ull));
    };
    auto fetch_offsets = [&](uint32_t pair) {
        const uint64_t rr = 2 * (uint64_t)pair + half;
        o0B = a.off[rr]; o1B = a.off[rr + 1];
    };
    auto at_of = [&](const Cur& c) { return c.it < nitems ? c.i : 0u; };
    Cur cC = seek(ifirst), cB = next(cC), cA = next(cB);
    if (cC.it < nitems) {
        fetch_offsets(x.uni(a.surv[at_of(cC)]));
        dstC = a.walk_dst[at_of(cC)];
        o0C = o0B; o1C = o1B;
        fetch_bytes(o0C, o1C);
        if (cB.it < nitems) { fetch_offsets(x.uni(a.surv[at_of(cB)])); dstB = a.walk_dst[at_of(cB)]; }
        pairA = a.surv[at_of(cA)]; dstA = a.walk_dst[at_of(cA)];
    }
    uint4 d1 = desc(ifirst), d2 = desc(ifirst + S);
    LocusDir ld1 = r.dir[x.uni(d1.x) < T.nloci ? x.uni(d1.x) : 0u];
    for (uint32_t e = (uint32_t)x.tid(); e < WFL_CNT; e += (uint32_t)x.nthreads()) smb.cnt[e] = 0;
    uint32_t trb_prev = 0;
    // the item's counts: LDS -> the locus' counters (one add per counter the item's pairs touched, not one per k-mer of every pair)
    auto flush_counts = [&]() {
        for (uint32_t e = (uint32_t)x.tid(); e < WFL_CNT; e += (uint32_t)x.nthreads()) {
            const uint32_t v = smb.cnt[e];
            if (v) { x.atomic_add(&a.counts[trb_prev + e], (uint64_t)v); smb.cnt[e] = 0; }
        }
    };
    for (uint32_t item = ifirst; item < nitems; item += S) {
        const uint4 d = d1;
        const LocusDir ld = ld1;
        const uint32_t locus = x.uni(d.x), lgnb = x.uni(ld.lgnb), trb = x.uni(ld.trbeg);
        x.bsync();  // every wave is done with the image of the item before, and with its counts
        flush_counts();
        trb_prev = trb;
        {
            const p2_v4u* src = reinterpret_cast<const p2_v4u*>(r.arena + 16ull * ld.off16 + LOC_HDR);
            const uint32_t n16 = (ld.bytes - LOC_HDR) / 16;
            p2_v4u t[IPT];
#pragma unroll
            for (int u = 0; u < IPT; ++u) {
                const uint32_t o = (uint32_t)x.tid() + (uint32_t)u * NW * 64;
                t[u] = src[o < n16 ? o : 0u];
            }
#pragma unroll
            for (int u = 0; u < IPT; ++u) {
                const uint32_t o = (uint32_t)x.tid() + (uint32_t)u * NW * 64;
                if (o < n16) *reinterpret_cast<p2_v4u*>(&smb.img[o]) = t[u];
            }
        }
        d1 = d2; d2 = desc(item + 2 * S);
        ld1 = r.dir[x.uni(d1.x) < T.nloci ? x.uni(d1.x) : 0u];
        x.bsync();
        const uint8_t* dsp = reinterpret_cast<const uint8_t*>(bks + (8u << lgnb));
        while (cC.it == item) {
            const uint32_t i = cC.i;  // place in the list
            const uint64_t o0 = o0C, o1 = o1C;
            uint32_t len = (uint32_t)(o1 - o0);
            if (len > lmax) { if (a.errflag) *a.errflag = DBTK_ERR_READ_TOO_LONG; len = lmax; }
            const uint64_t a0 = o0 & ~3ull;
            const uint32_t rsh = (uint32_t)(o0 - a0);
            const uint32_t dw0 = rw0, dw1 = rw1;
            const uint32_t dst = x.uni(dstC);
            {   // advance the pipeline
                cC = cB; cB = cA; cA = next(cA);
                const bool hasC = cC.it < nitems, hasB = cB.it < nitems;
                o0C = hasC ? o0B : 0ull; o1C = hasC ? o1B : 0ull;
                fetch_bytes(o0C, o1C);
                dstC = dstB; dstB = dstA;
                fetch_offsets(hasB ? x.uni(pairA) : x.uni(pairA) * 0u);
                pairA = a.surv[at_of(cA)]; dstA = a.walk_dst[at_of(cA)];
            }
            if (dst == NAN32) continue;  // the pair never reached threading (uniform)
            if (dst != locus) {          // (uniform, rare) voted to another locus than the list's order has it under: the other kernel's
                wf_hand_over<NPL>(x, a, W, i, nullptr);
                continue;
            }
            x.sync();
            uint32_t bad = 0;
            {
                const uint32_t c0 = pack4_b2(dw0, &bad), c1 = pack4_b2(dw1, &bad);
                reinterpret_cast<uint16_t*>(sm.pk[half])[hl ^ 1u] = (uint16_t)(((c0 >> 8) & 0xFF00u) | ((c1 >> 16) & 0xFFu));
            }
            const uint32_t nk = len >= k ? len - k + 1 : 0;
            const uint64_t badm = x.ballot(bad != 0 && 8 * hl < rsh + len);
            x.sync();
            if (badm) {  // (uniform) a pair with a non-ACGT byte is the other kernel's whatever its mates do
                if (a.pend_locus) { if (lane == 0) a.walk_ret[i] = WALK_PENDING; }
                else wf_hand_over<NPL>(x, a, W, i, nullptr);
                continue;
            }
            const uint64_t Wd = window_fw_clean(sm.pk[half], rsh + p0, 32);
            const uint64_t RW = revcomp2(Wd, 32);
            uint64_t fw[NPL], cn[NPL];
            uint32_t gi[NPL], bo[NPL];
            bool act[NPL];
            uint4 tg[NPL];
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                fw[j] = (Wd >> (2 * (32 - k - j))) & kmask;
                const uint64_t rc = (RW >> (2 * j)) & kmask;
                cn[j] = fw[j] <= rc ? fw[j] : rc;
                act[j] = p0 + j < nk;
                bo[j] = dsp[loc_group((uint32_t)cn[j], (uint32_t)(cn[j] >> 32), lgnb)];
            }
#pragma unroll
            for (int j = 0; j < NPL; ++j) bo[j] = 8 * ((loc_base((uint32_t)cn[j], (uint32_t)(cn[j] >> 32), lgnb) + bo[j]) & ((1u << lgnb) - 1));
#pragma unroll
            for (int j = 0; j < NPL; ++j) tg[j] = *reinterpret_cast<const uint4*>(bks + bo[j]);
            bool any[NPL];
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const uint32_t lo = (uint32_t)cn[j];
                uint32_t sl = 0;
                any[j] = false;
                if (tg[j].x == lo) { sl = 0; any[j] = true; }
                if (tg[j].y == lo) { sl = 1; any[j] = true; }
                if (tg[j].z == lo) { sl = 2; any[j] = true; }
                if (tg[j].w == lo) { sl = 3; any[j] = true; }
                gi[j] = bks[bo[j] + 4 + sl];
            }
            bool slow = false;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const uint32_t extra = (uint32_t)(cn[j] >> 32) >> lgnb, p = gi[j];
                const bool ok = any[j] && (p >> 24) == extra && p != LOC_EMPTY;
                slow |= act[j] && !ok && any[j];
                gi[j] = !act[j] ? 0u : ok ? p : any[j] ? 0xFFFFFFFFu : 0u;
            }
            if (x.ballot(slow)) {  // (rare) a tag matched but not its entry: every slot of the bucket
#pragma unroll
                for (int j = 0; j < NPL; ++j)
                    if (gi[j] == 0xFFFFFFFFu) {
                        const uint32_t lo = (uint32_t)cn[j], extra = (uint32_t)(cn[j] >> 32) >> lgnb;
                        const uint32_t* bk = bks + bo[j];
                        uint32_t p = 0;
                        for (int s2 = 0; s2 < 4; ++s2) {
                            const uint32_t q = bk[4 + s2];
                            if (bk[s2] == lo && (q >> 24) == extra && q != LOC_EMPTY) p = q;
                        }
                        gi[j] = p;
                    }
            }
            // the image's pay word -> the graph table's info word: the flags as they are, the counter back at GR_SLOT_SHIFT (the extra tag bits dropped)
#pragma unroll
            for (int j = 0; j < NPL; ++j) gi[j] &= 0x00FFFFFFu;
            wf_decide<NPL>(x, a, i, dst, len, nk, badm, fw, cn, gi, act, W, texting, smb.cnt, WFL_CNT);
        }
    }
    x.bsync();
    flush_counts();
    wf_finish(x, a, W);
    if (lane == 0) {
        if (W.c_feas) x.atomic_add(&ctr[DBTK_C_FEASIBLE], W.c_feas);
        if (W.c_inc) x.atomic_add(&ctr[DBTK_C_ALGO_INC], W.c_inc);
    }
}

// ---- the error-correcting walk with the locus' graph nodes resident in LDS (round 5).  What body_walk_pairs costs is the chain of
// dependent graph look-ups of a correction (walk_ec: the successor cube's two levels, the hypotheses' lock-step steps, the survivors'
// extensions; then the rewritten window): eight round trips to the 8.6-GB table per correction, 45 000 cycles of a dirty mate's ~60 000
// (tools/walk_bench.py with the stamps build).  The lean kernel's locus-resident form has the item's image in LDS already when it finds a
// pair it cannot decide; this kernel goes over the SAME items, copies the image in again, and walks the pairs that kernel marked
// (walk_ret == WALK_PENDING) with every look-up answered from LDS (DevTables::gimg: gr_lookup takes the image).  A workgroup of NW waves
// per item, each wave with its two WalkSmem; a wave takes every NW-th marked pair of the item.
template <int NW, int IMGB>
struct __attribute__((aligned(16))) WalkPairsLocSmemT {
    uint4 img[IMGB / 16];
    WalkConst c;  // a.T with the image named in it (and a.P): in LDS, because the walk's out-of-line routines take the tables by reference — a
                  // copy on the stack would be scratch memory, a round trip to HBM for every field they read
    WalkSmem w[NW][2];
};
template <int NW, int IMGB, class X>
DBTK_HD void body_walk_pairs_locus(X& x, const WalkArgs& a, const LocRunArgs& r) {
    typedef WalkPairsLocSmemT<NW, IMGB> SM;
    constexpr int IPT = ((IMGB - (int)LOC_HDR) / 16 + NW * 64 - 1) / (NW * 64);
    SM& smb = *x.template smem<SM>();
    const int lane = x.lane();
    const uint32_t wave = (uint32_t)x.tid() >> 6;
    WalkSmem* const smm = smb.w[wave];
#ifdef DBTK_STAMPS
    if (lane == 0) for (int m_ = 0; m_ < 2; ++m_) for (int i_ = 0; i_ < 8; ++i_) smm[m_].dst_[i_] = 0;
#endif
    if (x.tid() == 0) { smb.c.T = a.T; smb.c.P = a.P; smb.c.T.gimg = reinterpret_cast<const uint32_t*>(smb.img); smb.c.T.gimg_lgnb = 0; }
    x.bsync();
    uint64_t* const ctr = a.ctr_rep ? a.ctr_rep + (size_t)(x.bid() & (W_CTR_REP - 1)) * W_CTR_STRIDE : a.counters;
    const uint32_t ifirst = r.starts[x.bid()], nitems = r.starts[x.bid() + 1];
    auto desc = [&](uint32_t it) -> uint4 { return r.items[it < nitems ? it : 0u]; };
    auto uni64 = [&](uint64_t v) { return ((uint64_t)x.uni((uint32_t)(v >> 32)) << 32) | x.uni((uint32_t)v); };
    auto clampl = [&](uint64_t o0, uint64_t o1) { const uint64_t l = o1 - o0; return (uint32_t)(l > (uint64_t)MAXL ? (uint64_t)MAXL : l); };
    WalkPairAcc A;
    uint32_t nwalked = 0;
#ifdef DBTK_STAMPS
    A.wlast = x.clock();
#endif
    uint4 d1 = desc(ifirst), d2 = desc(ifirst + 1);
    LocusDir ld1 = r.dir[x.uni(d1.x) < a.T.nloci ? x.uni(d1.x) : 0u];
    // (the marks of the next item's pairs are fetched an item ahead, like its descriptor: a wave that finds none of its own in an item
    // goes straight to the barrier)
    auto marks = [&](const uint4& d) -> uint32_t {
        const uint32_t n = d.z - d.y;
        return (uint32_t)lane < n ? a.walk_ret[d.y + (uint32_t)lane] : 0u;
    };
    uint32_t mk1 = ifirst < nitems ? marks(d1) : 0u;
    for (uint32_t item = ifirst; item < nitems; ++item) {
        const uint4 d = d1;
        const LocusDir ld = ld1;
        const uint32_t locus = x.uni(d.x);
        const uint64_t pending = x.ballot(mk1 == WALK_PENDING);  // (the same in every wave of the workgroup)
        d1 = d2; d2 = desc(item + 2);
        ld1 = r.dir[x.uni(d1.x) < a.T.nloci ? x.uni(d1.x) : 0u];
        mk1 = item + 1 < nitems ? marks(d1) : 0u;
        if (!pending) continue;  // (uniform over the workgroup: no barrier is skipped by some waves only)
        x.bsync();  // every wave is done with the image of the item before
        {
            const p2_v4u* src = reinterpret_cast<const p2_v4u*>(r.arena + 16ull * ld.off16 + LOC_HDR);
            const uint32_t n16 = (ld.bytes - LOC_HDR) / 16;
            p2_v4u t[IPT];
#pragma unroll
            for (int u = 0; u < IPT; ++u) {
                const uint32_t o = (uint32_t)x.tid() + (uint32_t)u * NW * 64;
                t[u] = src[o < n16 ? o : 0u];
            }
#pragma unroll
            for (int u = 0; u < IPT; ++u) {
                const uint32_t o = (uint32_t)x.tid() + (uint32_t)u * NW * 64;
                if (o < n16) *reinterpret_cast<p2_v4u*>(&smb.img[o]) = t[u];
            }
        }
        if (x.tid() == 0) smb.c.T.gimg_lgnb = ld.lgnb;  // (between the item's two barriers)
        // this wave's pairs: every NW-th marked one.  Lane j holds the j-th of them: place, pair index, offsets (the two dependent loads
        // of all of them in flight together)
        uint32_t my_t = 0, my_pair = 0;
        uint64_t my_o[3] = {0, 0, 0};
        uint32_t nmine = 0;
        {
            uint64_t m = pending;
            uint32_t rank = 0;
            while (m) {
                const uint32_t b = (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                if (rank % NW == wave) { if ((uint32_t)lane == nmine) my_t = d.y + b; ++nmine; }
                ++rank;
            }
        }
        if ((uint32_t)lane < nmine) {
            my_pair = a.surv[my_t];
            my_o[0] = a.off[2 * (uint64_t)my_pair]; my_o[1] = a.off[2 * (uint64_t)my_pair + 1]; my_o[2] = a.off[2 * (uint64_t)my_pair + 2];
        }
        x.bsync();  // the image is in LDS
        uint64_t oA[3] = {0, 0, 0}, oB[3];
        uint32_t wA[2][2] = {{0, 0}, {0, 0}}, wB[2][2];
        auto take = [&](uint32_t j, uint64_t (&o)[3], uint32_t (&w)[2][2]) {  // offsets and raw words of this wave's j-th pair (loads issued, not waited for)
            for (int q = 0; q < 3; ++q) o[q] = ((uint64_t)x.bcast((uint32_t)(my_o[q] >> 32), (int)j) << 32) | x.bcast((uint32_t)my_o[q], (int)j);
            walk_raw_words(a.seq, o[0], clampl(o[0], o[1]), lane, w[0]);
            walk_raw_words(a.seq, o[1], clampl(o[1], o[2]), lane, w[1]);
        };
        if (nmine) take(0, oA, wA);
        for (uint32_t j = 0; j < nmine; ++j) {
            if (j + 1 < nmine) take(j + 1, oB, wB);
            const uint32_t t = x.bcast(my_t, (int)j), pair = x.bcast(my_pair, (int)j);
            walk_pair(x, smm, a, smb.c, t, locus, pair, NAN32, oA, wA, A);
            ++nwalked;
            for (int q = 0; q < 3; ++q) oA[q] = oB[q];
            for (int m = 0; m < 2; ++m) { wA[m][0] = wB[m][0]; wA[m][1] = wB[m][1]; }
        }
    }
    if (a.aln && lane == 0)  // the slots of the last chunk that were not used
        for (; A.slot_used < ALN_CHUNK; ++A.slot_used) {
            const uint32_t slot = A.slot_base + A.slot_used;
            if (slot < a.aln_max) reinterpret_cast<dbtk_aln_hdr_t*>(a.aln + (size_t)slot * a.aln_stride)->pair = NAN32;
        }
#ifdef DBTK_STAMPS
    if (lane == 0 && a.dbg) {
        for (int i_ = 0; i_ < 8; ++i_) if (A.wst[i_]) x.atomic_add(&a.dbg[i_], A.wst[i_]);
        for (int m_ = 0; m_ < 2; ++m_) for (int i_ = 0; i_ < 8; ++i_) if (smm[m_].dst_[i_]) x.atomic_add(&a.dbg[8 + i_], smm[m_].dst_[i_]);
    }
#endif
    if (lane == 0) {
        if (A.c_feas) x.atomic_add(&ctr[DBTK_C_FEASIBLE], A.c_feas);
        if (A.c_inc) x.atomic_add(&ctr[DBTK_C_ALGO_INC], A.c_inc);
        if (a.pstats && nwalked) x.atomic_add(&a.pstats[18], (uint64_t)nwalked);
    }
}

}  // namespace dbtk
#endif
