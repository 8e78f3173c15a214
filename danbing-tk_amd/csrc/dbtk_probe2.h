// dbtk_probe2.h — the probe kernel (K2) in its lean form: kfilter's look-ups (src/aQueryFasta_thread.cpp:190-224, the
// `kmerDBi.find(kmers[i])` of every position of both mates) through the minimizer-grouped copy of the index
// (dbtk_tables.h: MzBucket, MzSlot).  Included by dbtk_kernels.h (needs BatchArgs); instantiated with DevX on the GPU and
// with the coroutine lanes of tests/emu on the host.
//
// What bounds this kernel is the number of 128-byte lines it asks the memory system for (dbtk_tables.h: a request is a line
// asked for by one load instruction), so it is built to ask for every line once:
//   * one wave per surviving PAIR, one mate per half-wave: lane l of a half owns the NPL consecutive positions l * NPL ..
//     l * NPL + NPL - 1 of its mate (NPL = 5: reads up to 32 * 5 + m - 1 bases); everything a position needs is a shift of
//     one 32-base word W held by its lane and of W's reverse complement: the canonical k-mer, and the hashed canonical
//     m-mer of every BASE (each hashed once, handed to the two lanes before by DPP), whose sliding minimum is the position's minimizer;
//   * a RUN of positions with one minimizer shares a bucket: run starts are found with one wave scan, and the bucket of
//     every run is fetched ONCE — 8 lanes x 16 bytes, 8 runs per load instruction, the next chunk of runs in flight while
//     this one is searched — into LDS, where every position compares its k-mer with the 8 keys of its run's bucket;
//   * the keys a full bucket turned away (tandem repeats: many k-mers around few m-mers) are in the overflow table, one
//     16-byte load per look-up — and the same few k-mers come back all along a repeat and in every read of its locus, so the
//     wave keeps the results of its overflow look-ups in a small LDS cache: the survivor list is in locus order
//     (body_surv_*, below) and a wave works through a CONTIGUOUS range of it, the reads of one locus one after the other;
//   * results: 4 bytes per position (`aux`), + 4 (`val`) for a read whose found k-mers are not all unique to one locus,
//     transposed through LDS so that they leave as 16-byte stores; the per-read header (found positions, one index value or
//     not) comes from three half-wave reductions.
#ifndef DBTK_PROBE2_H_
#define DBTK_PROBE2_H_

namespace dbtk {

#ifndef DBTK_P2_RCH
#define DBTK_P2_RCH 40
#endif
constexpr int P2_RCH = DBTK_P2_RCH;  // runs (buckets) staged in LDS at a time (a pair of 150-bp reads has ~66: two chunks)
constexpr int P2_ROW = 9;     // 16-byte granules per staged bucket: 8 + 1 of padding (rows on different LDS banks)
constexpr int P2_CACHE = 128; // entries of the wave's cache of overflow look-ups
// (FUSE) what the resolve part needs of the kernel's arguments, read from LDS where it is used: held in scalar registers across the
// look-ups they cost the kernel its prefetched buckets (the register allocator spilt those: a spilled load is waited for where it is issued)
struct P2FuseArgs {
    uint64_t* counts; uint64_t* nmapread; uint64_t* kmc;
    uint32_t* gen_list; uint32_t* ngen; uint32_t* walk_dst;
    const uint8_t* qc;
    const ClsSlot* cls; uint64_t cls_mask; uint32_t cls_shift;  // the class table: (k-mer, locus) -> flank / counter (the shared-k-mer shortcut)
    uint64_t* pstats;
    uint64_t* ctr;     // this block's row of the counter replicas (counters_of), or the counters themselves
    uint32_t ctr_rep;  // ... which of the two: a row has CTR_STRIDE - DBTK_C_COUNT spare words, where the path statistics of this kernel go
                       // (thousands of blocks adding to ONE word serialize: 18 ns each)
    dbtk_params_t P;
};
template <int NPL>
struct __attribute__((aligned(16))) Probe2SmemT {
    uint32_t pk[2][20];              // 2-bit stream of each mate from its 4-byte-aligned start: 16 words (+ slack)
    uint16_t vd[2][40];              // validity bits of the same bases (only for a pair with a non-ACGT byte)
    uint32_t rb[64 * NPL];           // bucket of every run of the pair
    union {                          // two phases share one region
        uint4 stg[P2_RCH][P2_ROW];                                      // the buckets of a chunk of runs
        uint32_t res[2][2][32 * NPL];                                   // [mate][aux | val][position]: results on their way to 16-byte stores
    };
    uint4 cache[P2_CACHE];           // {k-mer, val, aux} of overflow look-ups already made (val = NOHIT: not in the index)
    uint8_t tok[P2_CACHE];           // which lane writes an entry when several want to in one step
    uint32_t gbuf[64];               // (FUSE) pairs for the general resolve kernel, not yet appended to its list
    uint32_t fc[12];                 // (FUSE) this wave's sums for the counters (kept here, not in scalar registers: the kernel has none to spare)
    P2FuseArgs fa;                   // (FUSE)
};
// (FUSE) the counters of the pair's locus that its TR k-mers fall on, staged in LDS where the buckets were (two 16-bit increments per
// word), and flushed with lanes on consecutive counters: one memory-side read-modify-write per touched line instead of one per k-mer
// (body_pair_usual has the measurements)
constexpr uint32_t P2_HWIN = 2048;
// spare words of a counter replica row (CTR_STRIDE = 32 words, DBTK_C_COUNT = 24 counters): the lean kernel's path statistics
constexpr uint32_t P2_REP_DONE = 24, P2_REP_CLS = 25, P2_REP_INC = 26, P2_REP_SHARED = 27;
#ifndef DBTK_P2_SHARED_MAX
#define DBTK_P2_SHARED_MAX 48  /* probe + resolve + body_pair of a WGS-like step: 16 / 32 / 48 / 96 measured 0.477 / 0.473 / 0.455 / 0.469 ms in round 5; round 6 (relaxed rule): 48: 0.459, no cap (320): 0.474 — a class-table look-up per shared position costs the lean kernel more than the pair costs body_pair */
#endif
constexpr uint32_t P2_SHARED_MAX = DBTK_P2_SHARED_MAX;  // shared positions of a pair the shortcut takes: each costs a look-up in the class table
static_assert(P2_HWIN / 2 * 4 <= sizeof(uint4) * P2_RCH * P2_ROW, "the counter window fits where the buckets were");

// out[j] = min(f[j .. j + WN - 1]), j < NPL, sharing the part common to all windows (WN >= NPL)
template <int NPL, int WN>
DBTK_HD void sliding_min(const uint32_t (&f)[NPL + WN - 1], uint32_t (&out)[NPL]) {
    static_assert(WN >= NPL, "the windows of one lane share at least one element");
    uint32_t c = f[NPL - 1];
#pragma unroll
    for (int t = NPL; t < WN; ++t) c = f[t] < c ? f[t] : c;
    uint32_t suf[NPL];  // suf[j] = min f[j .. NPL - 2]
    suf[NPL - 1] = 0xFFFFFFFFu;
#pragma unroll
    for (int j = NPL - 2; j >= 0; --j) suf[j] = f[j] < suf[j + 1] ? f[j] : suf[j + 1];
    uint32_t pre = 0xFFFFFFFFu;  // min f[WN .. WN + j - 1]
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        uint32_t v = suf[j] < c ? suf[j] : c;
        out[j] = pre < v ? pre : v;
        if (j + 1 < NPL) pre = f[WN + j] < pre ? f[WN + j] : pre;
    }
}

// The 16-byte part `part` of the buckets of runs r0 + 8u + fq8, u = 0 .. P2_RCH / 8 - 1: straight-line loads, no branch around them
// (the array must stay in registers and the loads in flight); a run past the end re-reads bucket 0, one hot line.
#ifdef DBTK_STAMPS
#define P2_DIAG_MASK(a) (((a).P.diag & 512) ? 127u : 0xFFFFFFFFu)  /* diagnostic: every bucket in the first 16 KB of the table (cache hits) */
#else
#define P2_DIAG_MASK(a) 0xFFFFFFFFu
#endif
typedef uint32_t p2_v4u __attribute__((vector_size(16)));  // (a native vector, not HIP's uint4 class: an array of these that lives across
                                                           // a loop's back edge stays in registers; an array of uint4 went to scratch memory)
DBTK_HD void p2_fetch_runs(const uint32_t* rb, const MzBucket* mz, uint32_t nruns, uint32_t r0, uint32_t fq8, uint32_t part, uint32_t bmask,
                           p2_v4u (&q)[P2_RCH / 8]) {
#pragma unroll
    for (int u = 0; u < P2_RCH / 8; ++u) {
        const uint32_t run = r0 + 8 * u + fq8;
        const uint32_t b = rb[run < nruns ? run : 0u] & bmask;
        q[u] = reinterpret_cast<const p2_v4u*>(mz + (run < nruns ? (size_t)b : (size_t)0))[part];
    }
}

// (FUSE) the two ways the lean probe kernel finishes a pair itself, as functions of their own: inlined, their registers cost the look-ups
// above them the prefetched buckets (see P2FuseArgs)
template <int NPL> struct P2Rv { uint64_t v[NPL]; };  // look-up results of the lane's positions: val | aux << 32
enum { P2C_QC, P2C_KF, P2C_THR, P2C_FEAS, P2C_ASGN, P2C_CLS, P2C_INC, P2C_NHASH1, P2C_DONE, P2C_VV, P2C_SHARED, P2C_HF };  // sm.fc: a wave's share of a batch (< 2^32)
// A mate kfilter clears takes its hit list with it (AQ.cpp:190-228).  gmask = which mates it clears (bit 0: mate 0, bit 1: mate 1).
// Both: nothing is left to vote on (a background pair that got through subfilter on a shared repeat).  ONE (round 6): countHit runs on the
// other mate's list alone, and cannot accept when that mate has fewer than 2 cth found positions — whatever the vote's order: the cleared
// mate's strand of `top` stays 0, so test1 fails, and top.fc + top.rc is at most the kept mate's found positions, so test2 fails
// (AQ.cpp:439-451) — the kept mate is "locus filtered" (hf), nothing else happens to the pair.  (A read that shares a 64-base stretch with
// a locus and one more base by chance: 45 found positions of 130.  In a genome-like batch these pairs were 150 000 sorts and votes per step.)
// What kfilter looked up before it gave up on a mate: the positions up to the (nk - cth + 1)-th miss; none for a mate with nk < cth; all of
// a kept mate's.  What fillstats reads of vv for the kept mate: one word per DISTINCT found k-mer whose index value is a list.
// The k-mers are re-made from the pair's 2-bit stream (sm.pk: every byte ACGT — the caller's condition when gmask != 3).
template <int NPL, class SM, class X>
DBTK_HD_NOINLINE void p2_resolve_gone(X& x, SM& sm_, P2Rv<NPL> rv, uint32_t nk, uint32_t gmask, uint32_t rsh, uint32_t k) {
    SM& sm = DBTK_LDS_REF(SM, sm_);  // (dbtk_tables.h: an out-of-line routine must be told that its reference is LDS)
    const int lane = x.lane();
    const uint32_t hl = (uint32_t)lane & 31u, half = (uint32_t)lane >> 5, p0 = hl * NPL, cth = sm.fa.P.cthreshold;
    const bool mygone = ((gmask >> half) & 1u) != 0;
    uint32_t mc = 0;
#pragma unroll
    for (int j = 0; j < NPL; ++j) mc += (p0 + j < nk && (uint32_t)rv.v[j] == NOHIT) ? 1u : 0u;
    uint32_t ex = x.wave_excl_scan(mc);
    const uint32_t ex32 = x.bcast(ex, 32);
    if (half) ex -= ex32;
    uint32_t upto = 0;  // abort position + 1
    if (mygone && nk >= cth) {
        const uint32_t target = nk - cth + 1;
#pragma unroll
        for (int j = 0; j < NPL; ++j)
            if (p0 + j < nk && (uint32_t)rv.v[j] == NOHIT) { ++ex; if (ex == target) upto = p0 + j + 1; }
    }
    if (!mygone) upto = nk;  // (a kept mate: kfilter ran over all of it)
    upto = x.half_max(upto);
    const uint32_t looked = x.bcast(upto, 0) + x.bcast(upto, 32);
    uint32_t nvv = 0;
    if (gmask != 3u) {
        bool sh[NPL];
        bool any = false;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            sh[j] = !mygone && p0 + j < nk && (uint32_t)rv.v[j] != NOHIT && ((uint32_t)rv.v[j] & 1u);
            any |= sh[j];
        }
        if (x.ballot(any)) {  // (rare in such a pair) the distinct ones: handed round the wave, each compared with the ones after it
            const uint64_t kmask = (1ull << (2 * k)) - 1;
            const uint64_t W = window_fw_clean(sm.pk[half], rsh + p0, 32), RW = revcomp2(W, 32);
            uint64_t km[NPL];
            uint32_t mine = 0, dup = 0;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const uint64_t fw = (W >> (2 * (32 - k - j))) & kmask, rc = (RW >> (2 * j)) & kmask;
                km[j] = fw < rc ? fw : rc;
                mine += sh[j] ? 1u : 0u;
            }
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                uint64_t mj = x.ballot(sh[j]);
                while (mj) {
                    const int src = (int)__builtin_ctzll(mj);
                    mj &= mj - 1;
                    const uint64_t ke = ((uint64_t)x.bcast((uint32_t)(km[j] >> 32), src) << 32) | x.bcast((uint32_t)km[j], src);
#pragma unroll
                    for (int j2 = j; j2 < NPL; ++j2)
                        if (sh[j2] && km[j2] == ke && (j2 > j || lane > src)) dup |= 1u << j2;
                }
            }
            nvv = x.wave_sum(mine) - x.wave_sum((uint32_t)__builtin_popcount(dup));
        }
    }
    if (lane == 0) {
        sm.fc[P2C_NHASH1] += looked;
        sm.fc[P2C_KF] += (gmask & 1u) + (gmask >> 1);
        if (gmask != 3u) { sm.fc[P2C_HF] += 1; sm.fc[P2C_VV] += nvv; }
    }
}
// the usual pair: both mates pass kfilter, every found k-mer unique to the locus v0 >> 1 — countHit needs no sort and no vote
// (body_pair_usual has the argument); nks = k-mers of both mates, t = the pair's place in the survivor list
template <int NPL, class SM, class X>
DBTK_HD_NOINLINE void p2_resolve_usual(X& x, SM& sm_, P2Rv<NPL> rv, uint32_t nk, uint32_t v0, uint32_t nks, uint32_t t) {
    SM& sm = DBTK_LDS_REF(SM, sm_);
    const int lane = x.lane();
    const uint32_t hl = (uint32_t)lane & 31u, half = (uint32_t)lane >> 5, p0 = hl * NPL;
    const P2FuseArgs& fa = sm.fa;
    auto tally = [&](int w, uint32_t v) { if (lane == 0) sm.fc[w] += v; };
    tally(P2C_DONE, 1);
    tally(P2C_NHASH1, nks);  // kfilter ran over both mates in full
    const uint32_t dst = v0 >> 1;
#if defined(DBTK_STAMPS) || defined(DBTK_LF_DIAG)
    if (fa.P.diag & 64) return;  // diagnostic: verdict only
#endif
    if (fa.P.qc && fa.qc && !fa.qc[dst]) { tally(P2C_QC, 2); return; }  // AQ.cpp:2059-2062
    if (fa.P.threading) {  // AQ.cpp:2070-2090; v1.3 (threading = 2): the walk kernels take the pair from here
        tally(P2C_THR, 2);
        if (fa.P.threading == DBTK_THREADING_V13 && lane == 0) fa.walk_dst[t] = dst;
        return;
    }
    tally(P2C_THR, 2); tally(P2C_FEAS, 2);
    if (fa.P.extract) return;  // AQ.cpp:2094-2099
    tally(P2C_CLS, nks);
    // assignTRkmc (AQ.cpp:2138-2144): the class of a found k-mer at the one locus rides with its index value
    bool kn[NPL], tr[NPL];
    uint32_t anyk = 0, anyt = 0;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const uint32_t ax = (uint32_t)(rv.v[j] >> 32);
        kn[j] = p0 + j < nk && (uint32_t)rv.v[j] != NOHIT && ax != CLS_NONE;
        tr[j] = kn[j] && ax != CLS_FLANK;
        anyk |= kn[j] && !tr[j] ? 1u : 0u; anyt |= tr[j] ? 1u : 0u;
    }
    // a mate without a TR k-mer is removed (no state change, first state flank: AQ.cpp:1531-1534), one all of whose known
    // k-mers are TR k-mers is the TR segment from end to end: only a mate with both needs the scan
    const uint64_t bt = x.ballot(anyt != 0), bf = x.ballot(anyk != 0);
    const bool t0 = (uint32_t)bt != 0, t1 = (bt >> 32) != 0, f0 = (uint32_t)bf != 0, f1 = (bf >> 32) != 0;
    bool rm = true;
    uint32_t span = 0;
    if ((t0 && f0) || (t1 && f1)) assign_halves<NPL>(x, kn, tr, p0, nk, fa.P, rm, span);
    else { const bool tmate = half ? t1 : t0; rm = !tmate; span = tmate ? nk : 0u; }
    const uint32_t rm0 = x.bcast(rm ? 1u : 0u, 0), rm1 = x.bcast(rm ? 1u : 0u, 32);
    if (rm0 && rm1) return;
    // accumulate (AQ.cpp:2145-2158)
    tally(P2C_ASGN, 2 - rm0 - rm1);
#if defined(DBTK_STAMPS) || defined(DBTK_LF_DIAG)
    if (fa.P.diag & 128) return;  // diagnostic: no counting at all
#endif
    uint32_t mn = 0xFFFFFFFFu, mx = 0;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const uint32_t ax = (uint32_t)(rv.v[j] >> 32);
        if (tr[j] && !rm) { mn = ax < mn ? ax : mn; mx = ax > mx ? ax : mx; }
    }
    const uint32_t hmn2 = ~x.half_max(~mn), hmx2 = x.half_max(mx);
    const uint32_t mnA = x.bcast(hmn2, 0), mnB = x.bcast(hmn2, 32), mxA = x.bcast(hmx2, 0), mxB = x.bcast(hmx2, 32);
    const uint32_t base = mnA < mnB ? mnA : mnB, top = mxA > mxB ? mxA : mxB;
    const uint32_t win = top >= base ? (top - base + 1 < P2_HWIN ? top - base + 1 : P2_HWIN) : 0u;
    uint32_t* hist = reinterpret_cast<uint32_t*>(sm.stg);
    for (uint32_t w = (uint32_t)lane; w < (win + 1) / 2; w += 64) hist[w] = 0;
    x.sync();
    uint32_t ninc = 0;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const bool hit = tr[j] && !rm;
        if (hit) {
            const uint32_t ax = (uint32_t)(rv.v[j] >> 32), o = ax - base;
            if (o < win) x.lds_add(&hist[o >> 1], 1u << (16 * (o & 1)));
            else x.atomic_add(&fa.counts[ax], 1ull);
        }
        ninc += (uint32_t)__builtin_popcountll(x.ballot(hit));
    }
    tally(P2C_INC, ninc);
    x.sync();
    const uint32_t span0 = x.bcast(span, 0), span1 = x.bcast(span, 32);
#if defined(DBTK_STAMPS) || defined(DBTK_LF_DIAG)
    if (!(fa.P.diag & 16))  // diagnostic: no per-locus atomics
#endif
    if (lane == 0) {
        x.atomic_add(&fa.nmapread[dst], (uint64_t)(2 - rm0 - rm1));
        x.atomic_add(&fa.kmc[dst], (uint64_t)span0 + span1);
    }
#if defined(DBTK_STAMPS) || defined(DBTK_LF_DIAG)
    if (!(fa.P.diag & 4))  // diagnostic: no count atomics
#endif
    for (uint32_t i = (uint32_t)lane; i < win; i += 64) {
        const uint32_t v = (hist[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
        if (v) x.atomic_add(&fa.counts[base + i], (uint64_t)v);
    }
}


// A pair with k-mers SHARED between loci is still locus L's — whatever order fillstats' unstable sort leaves the k-mers in — when the
// k-mers unique to L alone decide (the argument is body_probe_locus', dbtk_locus.h: every found k-mer unique to L or shared with L in
// its list, each mate cth unique ones, the unique ones at least as many as the shared ones): the caller has checked the counts, this
// function asks the class table for every shared k-mer's class AT L (with a consistent RPGG "(k-mer, L) is in flankDB[L] / trKmers[L]"
// is "L is in the k-mer's vv list") and, if all have one, resolves the pair as a usual pair of L with those classes.  What the shared
// k-mers still cost the reference is one vv word each in fillstats (their DISTINCT number: DBTK_C_ALGO_VV, AQ.cpp:311-316).
// The k-mers are re-made from the pair's 2-bit stream (sm.pk: every byte ACGT, the caller's condition); rsh = where the mate starts in it.
template <int NPL, class SM, class X>
DBTK_HD_NOINLINE bool p2_resolve_shared(X& x, SM& sm_, P2Rv<NPL> rv, uint32_t nk, uint32_t rsh, uint32_t k, uint32_t L, uint32_t nshared, uint32_t nks, uint32_t t) {
    SM& sm = DBTK_LDS_REF(SM, sm_);
    const int lane = x.lane();
    const uint32_t hl = (uint32_t)lane & 31u, half = (uint32_t)lane >> 5, p0 = hl * NPL;
    const P2FuseArgs& fa = sm.fa;
    const uint64_t kmask = (1ull << (2 * k)) - 1;
    const uint64_t W = window_fw_clean(sm.pk[half], rsh + p0, 32), RW = revcomp2(W, 32);
    uint64_t km[NPL];
    bool sh[NPL], bad = false;
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const uint64_t fw = (W >> (2 * (32 - k - j))) & kmask, rc = (RW >> (2 * j)) & kmask;
        km[j] = fw < rc ? fw : rc;
        sh[j] = p0 + j < nk && (uint32_t)rv.v[j] != NOHIT && ((uint32_t)rv.v[j] & 1u);
        mine += sh[j] ? 1u : 0u;
    }
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        if (sh[j]) {
            const uint32_t c = kl_lookup(fa.cls, fa.cls_mask, fa.cls_shift, km[j], L);
            bad |= c == CLS_NONE;
            rv.v[j] = (uint64_t)(2u * L) | ((uint64_t)c << 32);  // (from here on: a k-mer of L with that class)
        }
    }
    if (x.ballot(bad)) return false;
    // distinct shared k-mers of the pair (the mates overlap, a repeat repeats): every one is handed round the wave from its lane's
    // registers and compared with the ones after it in the order (j, lane): of equal ones all but the first are marked
    uint32_t dup = 0;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        uint64_t mj = x.ballot(sh[j]);
        while (mj) {
            const int src = (int)__builtin_ctzll(mj);
            mj &= mj - 1;
            const uint64_t ke = ((uint64_t)x.bcast((uint32_t)(km[j] >> 32), src) << 32) | x.bcast((uint32_t)km[j], src);
#pragma unroll
            for (int j2 = j; j2 < NPL; ++j2)
                if (sh[j2] && km[j2] == ke && (j2 > j || lane > src)) dup |= 1u << j2;
        }
    }
    const uint32_t nvv = nshared - x.wave_sum((uint32_t)__builtin_popcount(dup));
    if (lane == 0) { sm.fc[P2C_VV] += nvv; sm.fc[P2C_SHARED] += 1; }
    p2_resolve_usual<NPL>(x, sm, rv, nk, 2u * L, nks, t);
    return true;
}

// SEL: the kernel takes the pairs a.sel lists (what the locus-resident kernel leaves) instead of the whole chunk — a compile-time
// switch, so that the form without a list (every WGS-like batch) carries none of its registers or branches
// FUSE: the kernel also RESOLVES the pairs whose look-ups decide them (what body_pair_usual, dbtk_kernels.h, does from the hit rows:
// the usual pair — both mates pass kfilter, every found k-mer unique to one and the same locus — through QC gate, assignTRkmc
// (assign_halves, dbtk_assign.h: the lanes hold the positions' states already) and the count increments; and the pair kfilter removes
// altogether): such a pair writes no hit rows and no second kernel reads them.  Every other pair writes its rows as before and goes on
// the general resolve kernel's list.  The launcher's conditions are body_pair_usual's: consistent RPGG, no records, no trace / -b / -bu.
template <int NPL, int WN, bool SEL, bool FUSE, class X>
DBTK_HD void body_probe2(X& x, const BatchArgs& a) {
    typedef Probe2SmemT<NPL> SM;
    SM& sm = *x.template smem<SM>();
    const int lane = x.lane();
    const uint32_t hl = (uint32_t)lane & 31u, half = (uint32_t)lane >> 5;
    const DevTables& T = a.T;
    const uint32_t k = T.ksize, m = T.mz_m;  // k - m + 1 == WN (the launcher's condition)
    const uint32_t ns = *a.nsurv;
    const uint32_t tend = ns - a.t0 < a.tcap ? ns : a.t0 + a.tcap;
    const uint32_t npc = ns > a.t0 ? tend - a.t0 : 0;  // pairs of this chunk of the list; pair i -> hit-buffer rows 2i, 2i + 1
    const uint32_t npr = SEL ? *a.nsel : npc;         // ... of which this kernel takes all, or the ones listed in a.sel (dbtk_locus.h: the rest)
    if (SEL && a.pstats && x.bid() == 0 && lane == 0 && npr) x.atomic_add(&a.pstats[6], (uint64_t)npr);
    // this wave's pairs: a contiguous range of the (locus-ordered) list
    const uint32_t per = (npr + x.nblocks() - 1) / x.nblocks();
    const uint64_t lo64 = (uint64_t)x.bid() * per;
    const uint32_t first = lo64 < npr ? (uint32_t)lo64 : npr, hi = lo64 + per < npr ? (uint32_t)(lo64 + per) : npr;
    const uint32_t lmax = 32u * NPL + m - 1;  // bases the lanes of a half cover (the launcher promised no read is longer)
    const uint64_t kmask = (1ull << (2 * k)) - 1;
    const uint32_t mmask = (uint32_t)((1ull << (2 * m)) - 1);
    const uint32_t p0 = hl * NPL;
    // (FUSE) per-wave sums, flushed once at the end
    if (FUSE && lane < 12) sm.fc[lane] = 0;
    if (FUSE && lane == 0) {
        sm.fa.counts = a.counts; sm.fa.nmapread = a.nmapread; sm.fa.kmc = a.kmc; sm.fa.gen_list = a.gen_list; sm.fa.ngen = a.ngen;
        sm.fa.walk_dst = a.walk_dst; sm.fa.qc = a.T.qc; sm.fa.pstats = a.pstats; sm.fa.P = a.P;
        sm.fa.ctr = counters_of(x, a); sm.fa.ctr_rep = a.ctr_rep ? 1u : 0u;
        sm.fa.cls = a.T.cls; sm.fa.cls_mask = a.T.cls_mask; sm.fa.cls_shift = a.T.cls_shift;
    }
    const P2FuseArgs& fa = sm.fa;
    uint32_t ngb = 0;
    auto flush_gen = [&]() {
        x.sync();
        uint32_t base = 0;
        if (lane == 0) base = x.atomic_add(fa.ngen, ngb);
        base = x.bcast(base, 0);
        if ((uint32_t)lane < ngb) fa.gen_list[base + lane] = sm.gbuf[lane];
        x.sync();
        ngb = 0;
    };
    for (uint32_t e = (uint32_t)lane; e < (uint32_t)P2_CACHE; e += 64) sm.cache[e] = uint4{0xFFFFFFFFu, 0x3FFFFFFFu, 0u, 0u};  // MZ_EMPTY
    // Three-deep fetch pipeline, all loads unconditional (clamped to something valid) so that they stay in flight:
    // while pair i is looked up, the bytes of pair i + 1 are on their way into registers, the offsets of pair i + 2
    // are being fetched, and the survivor entry of pair i + 3.
    // (place of the range's pair i in the chunk: i itself, or what the list says; its survivor entry is read an iteration after the
    // place, so that neither load is waited for where it is issued)
    auto place_of = [&](uint32_t i) -> uint32_t {
        const uint32_t ic = i < hi ? i : (first < hi ? first : 0u);
        return SEL ? (npr ? a.sel[ic] : 0u) : ic;
    };
    uint32_t rw0 = 0, rw1 = 0;  // pair i: dwords 2 hl and 2 hl + 1 of the mate, from its 4-byte-aligned start
    uint64_t o0C = 0, o1C = 0;  //         its offsets
    uint64_t o0B = 0, o1B = 0;  // pair i + 1: offsets (in flight)
    uint32_t pairA = 0;         // pair i + 2: survivor entry (in flight)
    uint32_t plN = 0;           // pair i + 3: its place (in flight)
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
        fetch_offsets(x.uni(a.surv[a.t0 + place_of(first)]));
        o0C = o0B; o1C = o1B;
        fetch_bytes(o0C, o1C);
        if (first + 1 < hi) fetch_offsets(x.uni(a.surv[a.t0 + place_of(first + 1)]));
        pairA = a.surv[a.t0 + place_of(first + 2)];
        plN = place_of(first + 3);
    }
    DBTK_STAMP_DECL
    for (uint32_t i = first; i < hi; ++i) {
        DBTK_STAMP(43);  // loop
        const uint64_t o0 = o0C, o1 = o1C;
        uint32_t len = (uint32_t)(o1 - o0);
        if (len > lmax) { *a.errflag = DBTK_ERR_READ_TOO_LONG; len = lmax; }
        const uint64_t a0 = o0 & ~3ull;
        const uint32_t rsh = (uint32_t)(o0 - a0);
        const uint32_t d0 = rw0, d1 = rw1;
        const uint32_t place_v = place_of(i);  // (read again here, needed with the results: cheaper than a register through the pipeline)
        {   // advance the pipeline
            const bool hasB = i + 1 < hi, hasA = i + 2 < hi;
            o0C = hasB ? o0B : 0ull; o1C = hasB ? o1B : 0ull;
            fetch_bytes(o0C, o1C);
            fetch_offsets(hasA ? x.uni(pairA) : x.uni(pairA) * 0u);
            pairA = a.surv[a.t0 + plN];
            plN = place_of(i + 4);
        }
        x.sync();  // the previous pair's LDS is dead
        uint32_t bad = 0;
        {   // 2-bit pack, eight bases per lane; entry hl of the big-endian 16-bit stream sits at index hl ^ 1 of the words
            const uint32_t c0 = pack4_b2(d0, &bad), c1 = pack4_b2(d1, &bad);
            reinterpret_cast<uint16_t*>(sm.pk[half])[hl ^ 1u] = (uint16_t)(((c0 >> 8) & 0xFF00u) | ((c1 >> 16) & 0xFFu));
        }
        const uint32_t nk = len >= k ? len - k + 1 : 0, nmm = len >= m ? len - m + 1 : 0;
        // Every byte of both mates ACGT (the usual pair): all windows are valid.  (Bytes of the lane's dwords outside the read
        // are the neighbouring reads': a non-ACGT byte there only sends this pair down the exact path for nothing.)
        const bool clean = x.ballot(bad != 0 && 8 * hl < rsh + len) == 0;
        uint64_t rv[NPL];  // look-up result per position: val | aux << 32, low word NOHIT: not in the index
        if (clean) {
            x.sync();
            const uint64_t W = window_fw_clean(sm.pk[half], rsh + p0, 32);  // the 32 bases from position p0
            const uint64_t RW = revcomp2(W, 32);                            // base t of that window at bits 2t
            // hashed canonical m-mer at the lane's own base positions; a position's window reaches into the next two lanes' m-mers:
            // they come over by DPP (wave_shl), not through LDS.  (Lane 31's neighbours are the other mate's: only windows past the
            // end of the read reach them, and those positions are not looked up.)
            uint32_t f[NPL + WN - 1];
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const uint32_t fwm = (uint32_t)(W >> (2 * (32 - m - j))) & mmask, rcm = (uint32_t)(RW >> (2 * j)) & mmask;
                f[j] = p0 + j < nmm ? mmer_hash2(fwm, rcm) : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int t = NPL; t < NPL + WN - 1; ++t) f[t] = x.shfl_down1(f[t - NPL]);
            // canonical k-mers (the m-mer hashes make their round trip through LDS meanwhile)
            uint64_t km[NPL];
            bool act[NPL];
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const uint64_t fw = (W >> (2 * (32 - k - j))) & kmask, rc = (RW >> (2 * j)) & kmask;
                km[j] = fw < rc ? fw : rc;
                act[j] = p0 + j < nk;
                rv[j] = (uint64_t)NOHIT;
            }
            DBTK_STAMP(40);  // fetch pipeline, pack, windows, m-mer hashes
            uint32_t mz[NPL], bk[NPL];
            sliding_min<NPL, WN>(f, mz);
            // runs: a position opens one when its bucket differs from the previous position's (a mate's first position always does)
            uint32_t cnt = 0;
            bool st[NPL];
#pragma unroll
            for (int j = 0; j < NPL; ++j) bk[j] = act[j] ? mz_bucket(mz[j] >> 4, (uint32_t)T.mz_mask) : 0xFFFFFFFFu;
            uint32_t prev = x.shfl_up1(bk[NPL - 1]);
            if (hl == 0) prev = 0xFFFFFFFEu;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                st[j] = act[j] && bk[j] != prev;
                cnt += st[j] ? 1u : 0u;
                prev = bk[j];
            }
            uint32_t r = x.wave_excl_scan(cnt);            // runs opened by the lanes before this one (mate 0's come first)
            const uint32_t nruns = x.bcast(r + cnt, 63);   // (wave-uniform)
            uint32_t rid[NPL];
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                if (st[j]) { sm.rb[r] = bk[j]; ++r; }
                rid[j] = r - 1;  // (meaningful for active positions only: a non-start continues the run before it)
            }
            x.sync();
            DBTK_STAMP(16);  // minimizers + runs
            // the buckets, P2_RCH runs at a time: 8 lanes x 16 bytes per bucket, 8 buckets per load instruction; the loads of the
            // next chunk are issued before this one is searched
            const uint32_t fq8 = (uint32_t)lane >> 3, part = (uint32_t)lane & 7u;
            p2_v4u q[P2_RCH / 8];
            p2_fetch_runs(sm.rb, T.mz, nruns, 0u, fq8, part, P2_DIAG_MASK(a), q);
            bool pend[NPL];
#pragma unroll
            for (int j = 0; j < NPL; ++j) pend[j] = false;
            for (uint32_t r0 = 0; r0 < nruns; r0 += P2_RCH) {
                x.sync();  // (the previous chunk is dead)
#pragma unroll
                for (int u = 0; u < P2_RCH / 8; ++u) *reinterpret_cast<p2_v4u*>(&sm.stg[8 * u + fq8][part]) = q[u];
                if (r0 + P2_RCH < nruns)  // (uniform) the next chunk's loads go out before this one is searched
                    p2_fetch_runs(sm.rb, T.mz, nruns, r0 + P2_RCH, fq8, part, P2_DIAG_MASK(a), q);
                x.sync();
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    const uint32_t row = rid[j] - r0;
                    const bool mine = act[j] && row < (uint32_t)P2_RCH;
                    if (mine) {  // (under the execution mask: the lanes whose rows are in the other chunk cost the LDS nothing: 4.90 -> 4.70 ms)
                        const uint4* rp = sm.stg[row];
                        int hit = -1;
                        uint32_t w7 = 0;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {  // (64-bit compares; only key 7 can carry the flag bit)
                            const uint4 kk = rp[g];
                            const uint64_t k0 = ((uint64_t)kk.y << 32) | kk.x, k1 = ((uint64_t)kk.w << 32) | kk.z;
                            if (k0 == km[j]) hit = 2 * g;
                            if ((g == 3 ? (k1 & ~MZ_TURNED) : k1) == km[j]) hit = 2 * g + 1;
                            if (g == 3) w7 = kk.w;
                        }
                        const uint32_t* pl = reinterpret_cast<const uint32_t*>(rp + 4) + 2 * (hit >= 0 ? hit : 0);
                        const uint64_t pv = (uint64_t)pl[0] | ((uint64_t)pl[1] << 32);
                        if (hit >= 0) rv[j] = pv;
                        else pend[j] = (w7 >> 31) != 0;  // the bucket turned keys away: ask the overflow table
                    }
                }
            }
            DBTK_STAMP(18);  // level 1: buckets fetched, staged, searched
            bool anyp = false;
#pragma unroll
            for (int j = 0; j < NPL; ++j) anyp |= pend[j];
#ifdef DBTK_STAMPS
            if (a.P.diag & 256) anyp = false;  // diagnostic: no level-2 look-ups (wrong results)
#endif
            if (x.ballot(anyp)) {
                // the wave's cache first: the overflow keys are those of tandem repeats, and they come back
                uint32_t oh[NPL], oi[NPL];
                bool glob[NPL];
                anyp = false;
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    oh[j] = ovf_hash(km[j]);
                    oi[j] = oh[j] & (uint32_t)T.ovf_mask;
                    glob[j] = false;
                    if (pend[j]) {
                        const uint4 c = sm.cache[(oh[j] >> 20) & (P2_CACHE - 1)];
                        if (c.x == (uint32_t)km[j] && c.y == (uint32_t)(km[j] >> 32)) { rv[j] = ((uint64_t)c.w << 32) | c.z; pend[j] = false; }
                        else glob[j] = true;
                    }
                    anyp |= pend[j];
                }
                if (x.ballot(anyp)) {
                    // (the table is at most a sixteenth full: a look-up that needs a second slot, and with it a second round trip, is one in thirty)
                    uint4 q2[NPL];
                    do {
#pragma unroll
                        for (int j = 0; j < NPL; ++j) q2[j] = reinterpret_cast<const uint4*>(T.ovf)[pend[j] ? oi[j] : 0u];
                        anyp = false;
#pragma unroll
                        for (int j = 0; j < NPL; ++j) {
                            if (!pend[j]) continue;
                            const uint64_t key = ((uint64_t)q2[j].y << 32) | q2[j].x;
                            if (key == km[j]) { rv[j] = ((uint64_t)q2[j].w << 32) | q2[j].z; pend[j] = false; }
                            else if (key == MZ_EMPTY) pend[j] = false;  // (rv stays NOHIT)
                            else oi[j] = (oi[j] + 1) & (uint32_t)T.ovf_mask;
                            anyp |= pend[j];
                        }
                    } while (x.ballot(anyp));
                    // what was looked up goes into the cache (the absent ones too); of the lanes that want one entry in this
                    // step, the one whose token is left standing writes it
#pragma unroll
                    for (int j = 0; j < NPL; ++j) if (glob[j]) sm.tok[(oh[j] >> 20) & (P2_CACHE - 1)] = (uint8_t)lane;
                    x.sync();
#pragma unroll
                    for (int j = 0; j < NPL; ++j) {  // (two of a lane's own positions on one entry: both pass, the later write stands)
                        const uint32_t ce = (oh[j] >> 20) & (P2_CACHE - 1);
                        if (glob[j] && sm.tok[ce] == (uint8_t)lane)
                            sm.cache[ce] = uint4{(uint32_t)km[j], (uint32_t)(km[j] >> 32), (uint32_t)rv[j], (uint32_t)(rv[j] >> 32)};
                    }
                    x.sync();
                }
            }
            DBTK_STAMP(41);  // level 2
        } else {
            // (rare) a non-ACGT byte somewhere in the pair: exact validity bits, every position through the plain index
            uint32_t vb = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t sidx = 8 * hl + e, by = ((e < 4 ? d0 : d1) >> (8 * (e & 3))) & 0xFFu;
                const bool ok = sidx >= rsh && sidx < rsh + len && (by == 'A' || by == 'C' || by == 'G' || by == 'T');
                vb |= (ok ? 1u : 0u) << (7 - e);
            }
            reinterpret_cast<uint8_t*>(sm.vd[half])[hl ^ 1u] = (uint8_t)vb;
            x.sync();
#pragma nounroll
            for (int j = 0; j < NPL; ++j) {  // (rolled: this path must not cost the hot one registers)
                uint64_t r1 = (uint64_t)NOHIT;
                if (p0 + j < nk) {
                    const uint64_t kq = window_kmer(sm.pk[half], sm.vd[half], rsh + p0 + j, k, nullptr, nullptr);
                    if (kq != NAN64) r1 = idx_lookup64(T, kq);
                }
                sm.res[half][0][p0 + j] = (uint32_t)r1 != NOHIT ? (uint32_t)(r1 >> 32) : AUX_MISS;  // (parked in the results region: it does
                sm.res[half][1][p0 + j] = (uint32_t)r1;                                              // not overlap pk / vd)
            }
            x.sync();
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const uint32_t v = sm.res[half][1][p0 + j];
                rv[j] = (uint64_t)v | ((uint64_t)sm.res[half][0][p0 + j] << 32);
            }
        }
        {   // the read's results: found positions, and whether they are all unique to one and the same locus
            x.sync();  // (the staged buckets / validity bits are dead: the region now takes the results)
            uint32_t cnt = 0, vmx = 0, vmn = 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const uint32_t v = p0 + j < nk ? (uint32_t)rv[j] : NOHIT;
                const bool fnd = v != NOHIT;
                cnt += fnd ? 1u : 0u;
                vmx = fnd && v > vmx ? v : vmx;
                vmn = fnd && v < vmn ? v : vmn;
                if (!FUSE) {  // (FUSE: only once the pair is known to need its rows — the region may become the counter window)
                    sm.res[half][0][p0 + j] = fnd ? (uint32_t)(rv[j] >> 32) : AUX_MISS;
                    sm.res[half][1][p0 + j] = v;
                }
            }
            const uint32_t nh = x.half_sum(cnt), hmx = x.half_max(vmx), hmn = ~x.half_max(~vmn);
            const bool uniform = T.consistent && nh && hmx == hmn && !(hmx & 1u);
            bool done = false;
            if (FUSE) {
                const uint32_t cth = fa.P.cthreshold;
                P2Rv<NPL> rvs;
#pragma unroll
                for (int j = 0; j < NPL; ++j) rvs.v[j] = rv[j];
                const uint32_t nk0 = x.bcast(nk, 0), nk1 = x.bcast(nk, 32), nh0 = x.bcast(nh, 0), nh1 = x.bcast(nh, 32);
                const uint32_t v0 = x.bcast(hmx, 0), v1 = x.bcast(hmx, 32);
                const bool u0 = x.bcast(uniform ? 1u : 0u, 0) != 0, u1 = x.bcast(uniform ? 1u : 0u, 32) != 0;
                // kfilter (AQ.cpp:190-228) clears a mate with fewer than cth k-mers, or at its (nk - cth + 1)-th miss: iff it has fewer than cth found positions
                const bool gone0 = nk0 < cth || nh0 < cth, gone1 = nk1 < cth || nh1 < cth;
                if (gone0 && gone1) {
                    p2_resolve_gone<NPL>(x, sm, rvs, nk, 3u, rsh, k);
                    done = true;
                } else if (clean && gone0 != gone1 && (gone0 ? nh1 : nh0) < 2 * cth) {
                    p2_resolve_gone<NPL>(x, sm, rvs, nk, gone0 ? 1u : 2u, rsh, k);  // (one mate cleared, the other cannot carry the pair alone)
                    done = true;
                } else if (!gone0 && !gone1 && nh0 && nh1 && u0 && u1 && v0 == v1) {
                    p2_resolve_usual<NPL>(x, sm, rvs, nk, v0, nk0 + nk1, a.t0 + place_v);
                    done = true;
                } else {
                    if (clean && !gone0 && !gone1) {
                        // k-mers shared between loci among the found ones: the pair may still be decided by the k-mers unique to one locus
                        uint32_t cu = 0, cs = 0, umx = 0, umn = 0xFFFFFFFFu;
#pragma unroll
                        for (int j = 0; j < NPL; ++j) {
                            const uint32_t v = p0 + j < nk ? (uint32_t)rv[j] : NOHIT;
                            const bool fnd = v != NOHIT, un = fnd && !(v & 1u);
                            cu += un ? 1u : 0u; cs += fnd && !un ? 1u : 0u;
                            umx = un && v > umx ? v : umx;
                            umn = un && v < umn ? v : umn;
                        }
                        const uint32_t hu = x.half_sum(cu), hs = x.half_sum(cs), hux = x.half_max(umx), hun = ~x.half_max(~umn);
                        const uint32_t u0n = x.bcast(hu, 0), u1n = x.bcast(hu, 32), ns = x.bcast(hs, 0) + x.bcast(hs, 32);
                        const uint32_t x0 = x.bcast(hux, 0), x1 = x.bcast(hux, 32), n0 = x.bcast(hun, 0), n1 = x.bcast(hun, 32);
                        // (round 6: one k-mer unique to the locus is enough, and each mate needs cth FOUND positions — which "not gone" says —
                        // not cth unique ones: body_probe_locus, dbtk_locus.h, has the argument)
                        const uint32_t wx = x0 > x1 ? x0 : x1, wn = n0 < n1 ? n0 : n1;  // (a mate without a unique k-mer: 0 / ~0)
                        // (the cap on the shared positions — a class-table look-up each — only where body_pair is the cheaper place for the
                        // pair: a WGS-like batch, !SEL.  The list the locus-resident kernel leaves, SEL, holds the pairs that reach a hundred
                        // positions into a flank shared with a neighbour locus: 1.3 % of an all-hit batch, 2.2 ms of body_pair's introsort per step)
                        if (T.consistent && ns && (SEL || ns <= P2_SHARED_MAX) && u0n + u1n >= 1 && wx == wn)
                            done = p2_resolve_shared<NPL>(x, sm, rvs, nk, rsh, k, wx >> 1, ns, nk0 + nk1, a.t0 + place_v);
                    }
                    if (!done) {  // the general resolve kernel redoes this pair from its rows
                        if (lane == 0) sm.gbuf[ngb] = a.t0 + place_v;
                        if (++ngb == 64) flush_gen();
                    }
                }
            }
            if (!done) {
                if (FUSE) {
#pragma unroll
                    for (int j = 0; j < NPL; ++j) {
                        const uint32_t v = p0 + j < nk ? (uint32_t)rv[j] : NOHIT;
                        sm.res[half][0][p0 + j] = v != NOHIT ? (uint32_t)(rv[j] >> 32) : AUX_MISS;
                        sm.res[half][1][p0 + j] = v;
                    }
                }
                const uint32_t row = 2 * x.uni(place_v) + half;
                if (hl == 0) {
                    a.hithdr[row] = (uint64_t)(nh ? hmx : NOHIT) | ((uint64_t)nh << 32) | (uniform ? HDR_UNIFORM : 0ull);
                    a.hitnk[row] = nk;
                    a.hitoff[row] = o0;
                }
                x.sync();
                uint4* outa = reinterpret_cast<uint4*>(a.hitaux + (size_t)row * a.nkp);
                uint4* outv = reinterpret_cast<uint4*>(a.hitval + (size_t)row * a.nkp);
#pragma unroll
                for (int c = 0; c < (32 * NPL + 127) / 128; ++c) {
                    const uint32_t i4 = 32u * c + hl;
                    if (4 * i4 < nk && i4 < 8u * NPL) {
                        outa[i4] = reinterpret_cast<const uint4*>(sm.res[half][0])[i4];
                        if (!uniform) outv[i4] = reinterpret_cast<const uint4*>(sm.res[half][1])[i4];
                    }
                }
            }
        }
        DBTK_STAMP(42);  // results
    }
    if (FUSE) {
        if (ngb) flush_gen();
        x.sync();
        if (lane == 0) {
            uint64_t* const ctr = fa.ctr;
            const uint32_t* f = sm.fc;
            if (f[P2C_QC]) x.atomic_add(&ctr[DBTK_C_QCFILTERED], (uint64_t)f[P2C_QC]);
            if (f[P2C_KF]) x.atomic_add(&ctr[DBTK_C_KMERFILTERED], (uint64_t)f[P2C_KF]);
            if (f[P2C_HF]) x.atomic_add(&ctr[DBTK_C_LOCUSFILTERED], (uint64_t)f[P2C_HF]);
            if (f[P2C_THR]) x.atomic_add(&ctr[DBTK_C_THREADING], (uint64_t)f[P2C_THR]);
            if (f[P2C_FEAS]) x.atomic_add(&ctr[DBTK_C_FEASIBLE], (uint64_t)f[P2C_FEAS]);
            if (f[P2C_ASGN]) x.atomic_add(&ctr[DBTK_C_ASGN], (uint64_t)f[P2C_ASGN]);
            if (f[P2C_CLS]) x.atomic_add(&ctr[DBTK_C_ALGO_CLS], (uint64_t)f[P2C_CLS]);
            if (f[P2C_INC]) x.atomic_add(&ctr[DBTK_C_ALGO_INC], (uint64_t)f[P2C_INC]);
            if (f[P2C_NHASH1]) { x.atomic_add(&ctr[DBTK_C_NHASH1], (uint64_t)f[P2C_NHASH1]); x.atomic_add(&ctr[DBTK_C_ALGO_PROBES], (uint64_t)f[P2C_NHASH1]); }
            if (f[P2C_VV]) x.atomic_add(&ctr[DBTK_C_ALGO_VV], (uint64_t)f[P2C_VV]);
            if (fa.ctr_rep) {  // (folded into the path statistics with the counters: k_fold_counters)
                if (f[P2C_DONE]) x.atomic_add(&ctr[P2_REP_DONE], (uint64_t)f[P2C_DONE]);
                if (f[P2C_CLS]) x.atomic_add(&ctr[P2_REP_CLS], (uint64_t)f[P2C_CLS]);
                if (f[P2C_INC]) x.atomic_add(&ctr[P2_REP_INC], (uint64_t)f[P2C_INC]);
                if (f[P2C_SHARED]) x.atomic_add(&ctr[P2_REP_SHARED], (uint64_t)f[P2C_SHARED]);
            } else if (fa.pstats) {
                if (f[P2C_DONE]) x.atomic_add(&fa.pstats[20], (uint64_t)f[P2C_DONE]);
                if (f[P2C_CLS]) x.atomic_add(&fa.pstats[16], (uint64_t)f[P2C_CLS]);
                if (f[P2C_INC]) x.atomic_add(&fa.pstats[17], (uint64_t)f[P2C_INC]);
                if (f[P2C_SHARED]) x.atomic_add(&fa.pstats[19], (uint64_t)f[P2C_SHARED]);
            }
        }
    }
    DBTK_STAMP_FLUSH;
}

// ---- builders.  Level 1 from the finished plain index (keys unique, val | aux with them): pass 0 claims slots (a bucket's
// slots fill in order) and counts the keys turned away; pass 1 (after the host has sized the overflow table from that count)
// files those keys in level 2.
struct MzBuildArgs {
    const IdxBucket* idx;
    uint64_t nslots;    // 4 per IdxBucket
    MzBucket* mz;
    uint32_t mz_mask;   // buckets - 1
    MzSlot* ovf;        // pass 1 only
    uint32_t ovf_mask;
    uint32_t ksize, m, pass;
    uint64_t* nturned;  // pass 0: += keys that found their bucket full
};
template <class X>
DBTK_HD void body_mz_insert(X& x, const MzBuildArgs& a) {
    uint64_t turned = 0;
    for (uint64_t i = (uint64_t)x.bid() * x.nthreads() + x.tid(); i < a.nslots; i += (uint64_t)x.nblocks() * x.nthreads()) {
        uint64_t key = a.idx[i >> 2].key[i & 3];
        if (key == NAN64) continue;
        key &= ~IDX_OVF;
        const uint64_t va = a.idx[i >> 2].val[i & 3];
        MzBucket* b = a.mz + mz_bucket(mz_of_kmer(key, a.ksize, a.m), a.mz_mask);
        if (a.pass == 0) {
            bool placed = false;
            for (int j = 0; j < 8 && !placed; ++j)
                if (x.atomic_cas(&b->key[j], MZ_EMPTY, key) == MZ_EMPTY) { b->pl[j].val = (uint32_t)va; b->pl[j].aux = (uint32_t)(va >> 32); placed = true; }
            if (!placed) { x.atomic_or(&b->key[7], MZ_TURNED); ++turned; }
        } else {
            bool there = false;
            for (int j = 0; j < 8; ++j) there |= (b->key[j] & ~MZ_TURNED) == key;
            if (!there) {
                uint32_t o = ovf_hash(key) & a.ovf_mask;
                while (x.atomic_cas(&a.ovf[o].key, MZ_EMPTY, key) != MZ_EMPTY) o = (o + 1) & a.ovf_mask;
                a.ovf[o].val = (uint32_t)va; a.ovf[o].aux = (uint32_t)(va >> 32);
            }
        }
    }
    if (turned) x.atomic_add(a.nturned, turned);
}
// every 16-byte slot of a level-1 / level-2 table free (a bucket = 4 slots' worth of keys, then 4 of payload: the payload words
// are overwritten with the pattern too, harmlessly)
template <class X>
DBTK_HD void body_mz_fill(X& x, uint64_t* t, uint64_t nwords, int level1) {
    for (uint64_t i = (uint64_t)x.bid() * x.nthreads() + x.tid(); i < nwords; i += (uint64_t)x.nblocks() * x.nthreads())
        t[i] = level1 ? ((i & 15) < 8 ? MZ_EMPTY : 0ull) : ((i & 1) ? 0ull : MZ_EMPTY);
}

// ---- the survivor list in locus order (a counting sort between the encode stage and the probe kernel).  Pairs are
// independent and every effect of one is an integer add (src/aQueryFasta_thread.cpp:2146-2158), so the order in which the
// survivors are taken changes no result; it decides how often a bucket of the tables is fetched from HBM.
struct SurvSortArgs {
    DevTables T;
    dbtk_params_t P;
    const uint8_t* seq;
    const uint64_t* off;
    const uint32_t* surv;    // the encode stage's list (in whatever order its waves appended)
    const uint32_t* nsurv;
    uint32_t* key;           // [nsurv]: locus of the pair's first sampled k-mer found in the index (nloci: none)
    uint32_t* hist;          // [nloci + 2 + SCAN_BLOCKS]: zero before body_surv_key; pairs per key, then (body_surv_scan) first place of each key
    uint32_t* sorted;        // [nsurv]: the list in key order
    uint32_t* flag;          // 1: sorted by key; 0: too few survivors for the order to matter (fewer than SORT_MIN_PER_LOCUS per locus:
                             // a WGS batch) — `sorted` is then a plain copy of the list and the key / histogram work is skipped
    uint32_t* rank;          // [nsurv]: a survivor's place among those of its key (what body_surv_key's counting atomic returned)
    uint32_t* starts;        // [nloci + 1]: first place of each key (body_surv_scan); hist[] then holds each key's END, which is what the locus lists read
    uint32_t have_keys;      // the encode stage has written key[] (BatchArgs::skey): body_surv_key only counts them
    uint32_t sort_min;       // survivors per locus from which the list is sorted (SORT_MIN_PER_LOCUS; 0 under DBTK_LOCUS_ALWAYS: tests of a dense
                             // slice — many pairs on few loci — of a large RPGG)
};
constexpr uint32_t SORT_MIN_PER_LOCUS = 4;
// canonical k-mer of the k bytes at p; NAN64 if one of them is not A, C, G or T
// The k bytes come in with three 16-byte loads from the 16-byte-aligned chunk below them (a load per byte is a request per byte: a line
// asked for by one instruction is one request, dbtk_tables.h — 21 requests per survivor made this the most expensive of the sort's three
// kernels) and are shifted into place; `seq` is readable up to the next multiple of 16 past its last read (dbtk.h): no load goes past the
// chunk that holds the k-mer's last byte.
DBTK_HD uint64_t kmer_of_bytes(const uint8_t* seq, uint64_t pos, uint32_t k) {
    const uint64_t b0 = pos & ~15ull;
    const uint32_t sh = (uint32_t)(pos & 15);
    const uint4* c = reinterpret_cast<const uint4*>(seq + b0);
    const uint32_t i1 = sh + k > 16 ? 1u : 0u, i2 = sh + k > 32 ? 2u : i1;  // (never past the chunk that holds the k-mer's last byte)
    const uint4 c0 = c[0], c1 = c[i1], c2 = c[i2];
    const uint64_t q[6] = {((uint64_t)c0.y << 32) | c0.x, ((uint64_t)c0.w << 32) | c0.z, ((uint64_t)c1.y << 32) | c1.x,
                           ((uint64_t)c1.w << 32) | c1.z, ((uint64_t)c2.y << 32) | c2.x, ((uint64_t)c2.w << 32) | c2.z};
    uint64_t r[5], w[4];
#pragma unroll
    for (int t = 0; t < 5; ++t) r[t] = (sh & 8) ? q[t + 1] : q[t];
    const uint32_t s = 8 * (sh & 7);
#pragma unroll
    for (int t = 0; t < 4; ++t) w[t] = s ? (r[t] >> s) | (r[t + 1] << (64 - s)) : r[t];
    uint64_t fw = 0, rc = 0;
    bool ok = true;
#pragma unroll
    for (uint32_t i = 0; i < 32; ++i) {
        if (i < k) {
            const uint32_t ch = (uint32_t)(w[i >> 3] >> (8 * (i & 7))) & 0xFFu, code = ((ch >> 1) ^ (ch >> 2)) & 3u;
            ok = ok && (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T');
            fw = (fw << 2) | code;
            rc = (rc >> 2) | ((uint64_t)(3u - code) << (2 * (k - 1)));
        }
    }
    return ok ? (fw < rc ? fw : rc) : NAN64;
}
// one lane per survivor: its key, counted
template <class X>
DBTK_HD void body_surv_key(X& x, const SurvSortArgs& a) {
    const uint32_t n = *a.nsurv, k = a.T.ksize, nloci = a.T.nloci;
    const bool on = (uint64_t)n >= (uint64_t)a.sort_min * nloci;
    if (x.bid() == 0 && x.tid() == 0) *a.flag = on ? 1u : 0u;
    if (!on) return;
    const uint32_t NF = (a.P.n_filter && a.P.nm_filter) ? a.P.n_filter : 4u;  // subfilter's sampled positions (AQ.cpp:172-188); four without it
    for (uint32_t t = x.bid() * (uint32_t)x.nthreads() + (uint32_t)x.tid(); t < n; t += x.nblocks() * (uint32_t)x.nthreads()) {
        if (a.have_keys) { a.rank[t] = x.atomic_add(&a.hist[a.key[t]], 1u); continue; }
        const uint64_t r = 2 * (uint64_t)a.surv[t];
        const uint64_t o0 = a.off[r];
        const uint32_t len = (uint32_t)(a.off[r + 1] - o0);
        uint32_t key = nloci;
        if (len >= k) {
            // The key is the locus of the first sampled k-mer that is in the index.  (Asking for two sampled k-mers to agree — to keep reads that
            // merely share a stretch with a locus out of its segment — was measured and dropped: the second look-up costs the sort 0.25 ms per
            // 10 M all-hit reads and 1.3 ms in the genome-like mix, and the locus kernel hands such pairs back by itself, dbtk_locus.h.)
            const uint32_t L = len - k + 1, S = NF > 1 ? L / (NF - 1) : 0;
            for (uint32_t sidx = 0; sidx < NF && key == nloci; ++sidx) {
                const uint32_t pos = sidx != NF - 1 ? sidx * S : L - 1;
                const uint64_t km = pos < L ? kmer_of_bytes(a.seq, o0 + pos, k) : NAN64;
                if (km == NAN64) continue;
                const uint32_t v = idx_lookup(a.T, km);
                if (v != NOHIT) key = (v & 1u) ? a.T.vv[(v >> 1) + 1] : v >> 1;
            }
        }
        if (key > nloci) key = nloci;
        a.key[t] = key;
        a.rank[t] = x.atomic_add(&a.hist[key], 1u);
    }
}
// hist[] (pairs per key) -> starts[] = exclusive prefix sums (the first place of each key) and hist[] = inclusive ones (its end), in two steps of SCAN_BLOCKS one-wave blocks: the sum of every
// block's chunk, then each block scans its chunk from the sum of the chunks before it
constexpr uint32_t SCAN_BLOCKS = 256;
template <class X>
DBTK_HD void body_surv_scan(X& x, const SurvSortArgs& a, int step) {
    if (!*a.flag) return;
    const uint32_t n = a.T.nloci + 1, lane = (uint32_t)x.lane(), b = x.bid();
    const uint32_t per = ((n + SCAN_BLOCKS - 1) / SCAN_BLOCKS + 63) & ~63u;  // entries per block (whole waves)
    const uint32_t lo = b * per, hi = lo + per < n ? lo + per : n;
    uint32_t* part = a.hist + n + 1;  // [SCAN_BLOCKS] behind the histogram
    if (step == 0) {
        uint32_t s = 0;
        for (uint32_t i = lo + lane; i < hi; i += 64) s += a.hist[i];
        s = x.wave_sum(s);
        if (lane == 0) part[b] = s;
        return;
    }
    uint32_t base = 0;
    for (uint32_t i = lane; i < b; i += 64) base += part[i];
    base = x.wave_sum(base);
    for (uint32_t i0 = lo; i0 < hi; i0 += 64) {
        const uint32_t v = i0 + lane < hi ? a.hist[i0 + lane] : 0u;
        const uint32_t ex = x.wave_excl_scan(v);
        if (i0 + lane < hi) { a.starts[i0 + lane] = base + ex; a.hist[i0 + lane] = base + ex + v; }
        base += x.wave_sum(v);
    }
}
// one lane per survivor: into its place — its key's first place + its rank among that key's survivors
template <class X>
DBTK_HD void body_surv_scatter(X& x, const SurvSortArgs& a) {
    const uint32_t n = *a.nsurv;
    const bool on = *a.flag != 0;
    for (uint32_t t = x.bid() * (uint32_t)x.nthreads() + (uint32_t)x.tid(); t < n; t += x.nblocks() * (uint32_t)x.nthreads())
        a.sorted[on ? a.starts[a.key[t]] + a.rank[t] : t] = a.surv[t];  // (round 6: no second atomic per survivor — its rank came back from the first)
}

}  // namespace dbtk
#endif
