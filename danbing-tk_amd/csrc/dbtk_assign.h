// dbtk_assign.h — assignTRkmc's state machine on 256-bit masks.
//
// The reference scans the per-position states as[i] in {0 unknown, 1 flank,
// 2 TR} once, left to right (src/aQueryFasta_thread.cpp:1477-1555).  On the GPU
// the states of a mate arrive as wave ballots: K = positions with a known state
// (as != 0), R = positions in state TR (as == 2), 64 positions per word.  The
// scan's result is a function of a handful of features of those masks — the
// first known state, the first two state transitions among known positions, and
// the unknown tract that ends exactly at each transition — which are extracted
// here with log-step fill-forward and bit scans by ONE lane in a few hundred
// operations instead of a 130-step dependent loop.  assign_scan (the literal
// restatement, dbtk_kernels.h) stays as the reference form; tests/emu checks the
// two against each other on random and exhaustive inputs.
#ifndef DBTK_ASSIGN_H_
#define DBTK_ASSIGN_H_

#include "dbtk_tables.h"

namespace dbtk {

struct MateState {  // km_asgn_read_t fields the state machine writes, AQ.cpp:93-108
    int si, ei, nt, bs, ti, si_, ei_, af, rm;
};

struct Bits256 {
    uint64_t w[4];
};

DBTK_HD Bits256 b_shl(const Bits256& a, int d) {  // bit i -> bit i + d
    Bits256 r;
    const int ws = d >> 6, bs = d & 63;
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        uint64_t v = 0;
        if (i - ws >= 0) {
            v = a.w[i - ws] << bs;
            if (bs && i - ws - 1 >= 0) v |= a.w[i - ws - 1] >> (64 - bs);
        }
        r.w[i] = v;
    }
    return r;
}
DBTK_HD int b_popc(const Bits256& a) {
    return __builtin_popcountll(a.w[0]) + __builtin_popcountll(a.w[1]) + __builtin_popcountll(a.w[2]) + __builtin_popcountll(a.w[3]);
}
// word i of a, selected without indexing the array by a run-time value (that would move it from registers to scratch memory)
DBTK_HD uint64_t b_word(const Bits256& a, int i) { return i == 0 ? a.w[0] : (i == 1 ? a.w[1] : (i == 2 ? a.w[2] : a.w[3])); }
DBTK_HD bool b_test(const Bits256& a, int i) { return (b_word(a, i >> 6) >> (i & 63)) & 1; }
DBTK_HD int b_first(const Bits256& a) {  // lowest set bit, 256 if none
    int r = 256;
#pragma unroll
    for (int i = 3; i >= 0; --i)
        if (a.w[i]) r = 64 * i + __builtin_ctzll(a.w[i]);
    return r;
}
DBTK_HD int b_last_below(const Bits256& a, int n) {  // highest set bit < n, -1 if none
    int r = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint64_t v = a.w[i];
        const int lo = 64 * i;
        if (n <= lo) v = 0;
        else if (n < lo + 64) v &= (1ull << (n - lo)) - 1;
        if (v) r = lo + 63 - __builtin_clzll(v);
    }
    return r;
}
DBTK_HD Bits256 b_clear_lowest(const Bits256& a) {
    Bits256 r = a;
    bool done = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool here = !done && r.w[i] != 0;
        if (here) r.w[i] &= r.w[i] - 1;
        done = done || here;
    }
    return r;
}

// Transition mask by log-step fill-forward on the masks alone (used by the self-test and as the
// definition the wave-scan form in k_pair must reproduce): T[i] = position i is known, a known
// position exists before it, and the last such position has the other state.
DBTK_HD Bits256 transitions_from_masks(const Bits256& K, const Bits256& R) {
    // Fv[i] = TR bit of the last known position <= i, Fm[i] = such a position exists
    Bits256 Fv = R, Fm = K;
#pragma unroll
    for (int d = 1; d < 256; d <<= 1) {
        const Bits256 sv = b_shl(Fv, d), sm = b_shl(Fm, d);
#pragma unroll
        for (int i = 0; i < 4; ++i) { Fv.w[i] |= sv.w[i] & ~Fm.w[i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { Fm.w[i] |= sm.w[i]; }
    }
    const Bits256 pv = b_shl(Fv, 1), pm = b_shl(Fm, 1);
    Bits256 T;
#pragma unroll
    for (int i = 0; i < 4; ++i) T.w[i] = K.w[i] & pm.w[i] & (R.w[i] ^ pv.w[i]);
    return T;
}

// The same on one 64-position word with a carry from the words before it, by carry propagation instead of fill steps:
// adding F << 1 (F = known flank positions) to ~K lets a carry run from just after every flank position through the
// unknown positions (ones in ~K) and stop on the next known one — so (~K + (F << 1)) & K marks the known positions whose
// last known predecessor is flank; likewise with R for TR.  A transition is a TR position after a flank one or a flank
// position after a TR one.  Two injections never meet: a run starts after a known position and ends at the next known one.
// `carry`: bit 0 / bit 1 = a flank / TR run is still open at the end of the previous word (a carry out of bit 63, or the
// injection of a flank / TR position at bit 63).  K and R are wave ballots, so all of this is scalar-unit work.
DBTK_HD uint64_t transitions_word(uint64_t K, uint64_t R, uint32_t& carry) {
    const uint64_t F = K & ~R, nK = ~K;
    const uint64_t aF = (F << 1) | (carry & 1), aR = (R << 1) | ((carry >> 1) & 1);
    const uint64_t sF = nK + aF, sR = nK + aR;
    const uint32_t cF = (uint32_t)(sF < nK) | (uint32_t)(F >> 63), cR = (uint32_t)(sR < nK) | (uint32_t)(R >> 63);
    carry = cF | (cR << 1);
    return (sF & K & R) | (sR & K & F);
}

// K, R: masks over positions [0, nk); R subset of K; T: the transition mask.  ntr: number of TR
// states reduced to uint8_t (AQ.cpp:1454).  r.rm on entry = the mate is already removed.
DBTK_HD void assign_masks(const Bits256& K, const Bits256& R, const Bits256& T, int nk, uint32_t ntr, const dbtk_params_t& P,
                          MateState& r) {
    if (r.rm) { r.nt = -1; r.bs = -1; r.ti = -1; return; }
    const int ntot = b_popc(T);
    const int i0 = b_first(K);
    r.bs = (i0 < 256) ? (b_test(R, i0) ? 2 : 1) : 0;
    const int maxnt = (int)P.max_nt;
    // the scan's early returns, in the order it meets them (only what each case needs is computed)
    const int ti1 = ntot >= 1 ? b_first(T) : 256;
    if (ntot >= 1 && maxnt >= 1) r.ti = ti1;  // ti1 is recorded after the nt > MAX_NT test of that step
    if (ntot >= 2 && maxnt >= 2 && r.bs == 2) { r.nt = 2; r.af = 1; r.rm = 1; return; }  // TR-flank-TR
    if (ntot > maxnt) { r.nt = maxnt + 1; r.af = 1; r.rm = 1; return; }
    r.nt = ntot;
    if (ntot == 0) {
        if (r.bs != 2) { r.af = 1; r.rm = 1; return; }
        r.si = 0; r.ei = nk; r.si_ = 0; r.ei_ = nk;
        return;
    }
    // unknown tract that ends exactly at a transition: [last known before + 1, transition)
    int si1 = -1, ei1 = -1;
    if (ti1 > 0 && !b_test(K, ti1 - 1)) { si1 = b_last_below(K, ti1) + 1; ei1 = ti1; }
    if (ntot == 1) {
        if (r.bs == 1) {
            r.si = si1 >= 0 ? (si1 + ei1) / 2 : ti1; r.ei = nk;
            r.si_ = si1 >= 0 ? ei1 : ti1; r.ei_ = nk;
        } else {
            r.si = 0; r.ei = si1 >= 0 ? (si1 + ei1) / 2 : ti1;
            r.si_ = 0; r.ei_ = si1 >= 0 ? si1 : ti1;
        }
        return;
    }
    if (ntr < P.nm_tr) { r.af = 1; r.rm = 1; return; }
    const int ti2 = b_first(b_clear_lowest(T));
    int si2 = -1, ei2 = -1;
    if (ti2 > 0 && !b_test(K, ti2 - 1)) { si2 = b_last_below(K, ti2) + 1; ei2 = ti2; }
    r.si = (si1 >= 0 ? (si1 + ei1) / 2 : ti1);
    r.ei = (si2 >= 0 ? (si2 + ei2) / 2 : ti2);
    r.si_ = ei1 >= 0 ? ei1 : ti1;
    r.ei_ = si2 >= 0 ? si2 : ti2;
}

DBTK_HD void assign_bits(Bits256 K, Bits256 R, int nk, uint32_t ntr, const dbtk_params_t& P, MateState& r) {
    const Bits256 T = transitions_from_masks(K, R);
    assign_masks(K, R, T, nk, ntr, P, r);
}

// assignTRkmc's scan (AQ.cpp:1477-1555; the literal form is assign_scan, the mask form assign_masks) for the two mates at once, a
// mate per half-wave, lane hl of a half holding the states of the NPL consecutive positions hl * NPL ..: kn[j] = the position's
// k-mer is known at the locus (flank or TR), tr[j] = it is a TR k-mer.  What the scan's result depends on — the first known state,
// the number of state changes between consecutive known positions, the first two of them and the last known position before each
// — comes from one half-wave scan (the last known position before the lane's) and five half-wave reductions.  Returns the mate's
// verdict in every lane of its half: rm (the mate is removed: af) and ei - si (0 for a removed mate).
template <int NPL, class X>
DBTK_HD void assign_halves(X& x, const bool (&kn)[NPL], const bool (&tr)[NPL], uint32_t p0, uint32_t nk, const dbtk_params_t& P, bool& rm_out, uint32_t& span_out) {
    const uint32_t hl = (uint32_t)x.lane() & 31u;
    // (position + 1) << 1 | TR of the lane's last known position, 0: none
    uint32_t lastk = 0, ntr_l = 0, firstk = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        if (kn[j]) { lastk = ((p0 + j + 1) << 1) | (tr[j] ? 1u : 0u); if (firstk == 0xFFFFFFFFu) firstk = ((p0 + j) << 1) | (tr[j] ? 1u : 0u); }
        ntr_l += tr[j] ? 1u : 0u;
    }
    uint32_t prev = x.shfl_up1(x.half_scan_max(lastk));  // ... of the lanes before this one
    if (hl == 0) prev = 0;
    uint32_t cnt = 0, k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;  // state changes at the lane's positions; the first two as position << 8 | (last known position before + 1)
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        if (kn[j]) {
            const uint32_t t = tr[j] ? 1u : 0u;
            if (prev && (prev & 1u) != t) {
                const uint32_t key = ((p0 + j) << 8) | (prev >> 1);
                if (cnt == 0) k1 = key; else if (cnt == 1) k2 = key;
                ++cnt;
            }
            prev = ((p0 + j + 1) << 1) | t;
        }
    }
    const uint32_t ntot = x.half_sum(cnt), ntr = x.half_sum(ntr_l) & 0xFFu;  // uint8_t ntr, AQ.cpp:1454
    const uint32_t fk = ~x.half_max(~firstk);
    const uint32_t g1 = ~x.half_max(~k1);
    const uint32_t g2 = ~x.half_max(~(k1 == g1 ? k2 : k1));
    const uint32_t bs = fk == 0xFFFFFFFFu ? 0u : (fk & 1u) ? 2u : 1u;
    const uint32_t maxnt = P.max_nt;
    // where the TR segment begins / ends at a state change: the middle of the unknown tract that ends exactly there, else the position itself
    const uint32_t t1 = g1 >> 8, q1 = g1 & 0xFFu, t2 = g2 >> 8, q2 = g2 & 0xFFu;
    const uint32_t e1 = q1 < t1 ? (q1 + t1) / 2 : t1, e2 = q2 < t2 ? (q2 + t2) / 2 : t2;
    bool rm = false;
    uint32_t span = 0;
    if (ntot >= 2 && maxnt >= 2 && bs == 2) rm = true;  // TR-flank-TR
    else if (ntot > maxnt) rm = true;
    else if (ntot == 0) { if (bs != 2) rm = true; else span = nk; }
    else if (ntot == 1) span = bs == 1 ? nk - e1 : e1;
    else if (ntr < P.nm_tr) rm = true;
    else span = e2 - e1;
    rm_out = rm; span_out = span;
}

}  // namespace dbtk
#endif
