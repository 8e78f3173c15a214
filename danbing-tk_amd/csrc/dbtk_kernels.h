// dbtk_kernels.h — SPMD bodies of the align hot path.
//
// Each body is a template over an execution context X.  On the GPU X is DevX
// (dbtk_hip.hip: threadIdx/blockIdx, __syncthreads, wave64 ballot / DPP scans,
// global and LDS atomics).  tests/emu instantiates the SAME bodies with a
// coroutine-lane context so that the device logic is checked against the oracle
// on a machine without a GPU; that harness is test infrastructure and is not
// part of, linked into, or reachable from libdbtk_hip.so.
//
// X provides: tid() nthreads() bid() nblocks() lane() sync() ballot(bool)
//   wave_sum(u32) wave_min(u32) wave_excl_scan(u32) wave_scan_max(u32) bcast(u32, srclane)
//   wave_scan_lastnz(u32): inclusive scan with op(a, b) = b ? b : a;  shfl_up1(u32): value of lane - 1
//   quad_perm<P0,P1,P2,P3>(u32): lane 4q+j reads lane 4q+Pj (DPP);  shfl_xor64<E>(in[E], out[E], lanemask)
//   uni(u32: value known to be wave-uniform)
//   atomic_add(u64*,u64) atomic_add(u32*,u32)->old atomic_cas(u64*,exp,des)->old
//   atomic_max(u64*,u64) atomic_or(u64*,u64) lds_add(u32*,u32)->old lds_or(u32*,u32)
//   smem<T>()
// Every kernel runs one wavefront per block, so sync() only orders the wave's own LDS traffic.
//
// Kernels:
//   body_idx_insert / body_idx_finalize / body_idx_aux / body_flt_insert / body_cls_insert   table build in HBM
//   body_encode_subfilter   K1:  2-bit encode + subfilter over tiles of 16 pairs (presence filter, then bucket probes)
//   body_probe              K2:  kfilter's look-ups, one wave per surviving read, results to the hit buffers
//   body_pair_usual         K3a: the usual pair (one locus, enough hits): no sort, no vote; passes the rest on
//   body_pair               K3b: the general resolve: dedup -> introsort -> vote -> assign -> count (+ bait, bubbles)
#ifndef DBTK_KERNELS_H_
#define DBTK_KERNELS_H_

#include "dbtk_assign.h"
#include "dbtk_sort.h"
#include "dbtk_tables.h"
#include "dbtk_walk.h"

namespace dbtk {

// ------------------------------------------------------------------ build --
struct IdxBuildArgs {
    IdxBucket* bkt;
    uint64_t mask;   // buckets - 1
    uint32_t shift;
    const uint64_t* keys;
    const uint32_t* vals;
    uint64_t n;
};

template <class X>
DBTK_HD void body_idx_insert(X& x, const IdxBuildArgs& a) {
    for (uint64_t i = (uint64_t)x.bid() * x.nthreads() + x.tid(); i < a.n; i += (uint64_t)x.nblocks() * x.nthreads()) {
        const uint64_t key = a.keys[i];
        uint64_t b = hash_idx(key, a.shift);
        for (;;) {
            bool placed = false;
            for (int j = 0; j < 4 && !placed; ++j) {  // slots are claimed in order, so the occupied ones form a prefix
                const uint64_t prev = x.atomic_cas(&a.bkt[b].key[j], NAN64, key);
                if (prev == NAN64 || (prev & ~IDX_OVF) == key) {
                    x.atomic_max(&a.bkt[b].val[j], (i << 32) | a.vals[i]);  // kmerDBi[key] = val: last one wins
                    placed = true;
                }
            }
            if (placed) break;
            x.atomic_or(&a.bkt[b].key[3], IDX_OVF);  // full: lookups of absent keys must go on from here
            b = (b + 1) & a.mask;
        }
    }
}
struct FltBuildArgs {
    uint64_t* words;
    uint32_t logw, ksize;  // 2^logw words; k
    const uint64_t* keys;
    uint64_t n;
};
template <class X>
DBTK_HD void body_flt_insert(X& x, const FltBuildArgs& a) {
    for (uint64_t i = (uint64_t)x.bid() * x.nthreads() + x.tid(); i < a.n; i += (uint64_t)x.nblocks() * x.nthreads()) {
        const uint64_t m = kmix(a.keys[i], a.ksize);
        x.atomic_or(&a.words[flt_word(m, a.ksize, a.logw)], flt_bits(m));
    }
}
template <class X>
DBTK_HD void body_idx_finalize(X& x, IdxBucket* bkt, uint64_t nslots) {
    for (uint64_t i = (uint64_t)x.bid() * x.nthreads() + x.tid(); i < nslots; i += (uint64_t)x.nblocks() * x.nthreads())
        bkt[i >> 2].val[i & 3] &= 0xFFFFFFFFull;
}

// After both tables exist: (1) check that the index and the per-locus flank/TR
// sets describe the same (k-mer, locus) memberships — `ktools serialize` builds
// the index from exactly those sets (src/kmertools.cpp:232-258), but the four
// files are loaded independently, so this is verified, not assumed; (2) store the
// class of every single-locus k-mer next to its val.  stats[0] += memberships in
// the index, stats[1] += memberships missing from the class table.
struct IdxAuxArgs {
    IdxBucket* bkt;
    uint64_t nslots;
    DevTables T;  // cls, vv valid
    uint64_t* stats;
};
template <class X>
DBTK_HD void body_idx_aux(X& x, const IdxAuxArgs& a) {
    uint64_t nmemb = 0, nmiss = 0;
    for (uint64_t i = (uint64_t)x.bid() * x.nthreads() + x.tid(); i < a.nslots; i += (uint64_t)x.nblocks() * x.nthreads()) {
        uint64_t key = a.bkt[i >> 2].key[i & 3];
        if (key == NAN64) continue;
        key &= ~IDX_OVF;
        const uint32_t v = (uint32_t)a.bkt[i >> 2].val[i & 3];
        if (v & 1) {
            const uint32_t n = a.T.vv[v >> 1];
            for (uint32_t j = 0; j < n; ++j) {
                ++nmemb;
                if (cls_lookup(a.T, key, a.T.vv[(v >> 1) + 1 + j]) == CLS_NONE) ++nmiss;
            }
        } else {
            ++nmemb;
            const uint32_t c = cls_lookup(a.T, key, v >> 1);
            if (c == CLS_NONE) ++nmiss;
            a.bkt[i >> 2].val[i & 3] = (uint64_t)v | ((uint64_t)c << 32);
        }
    }
    if (nmemb) x.atomic_add(&a.stats[0], nmemb);
    if (nmiss) x.atomic_add(&a.stats[1], nmiss);
}

struct ClsBuildArgs {
    ClsSlot* slots;
    uint64_t mask;
    uint32_t shift;
    const uint64_t* ks;       // k-mers, loci concatenated
    const uint64_t* beg;      // nloci+1 prefix offsets into ks
    uint32_t nloci;
    const uint64_t* outslot;  // TR pass: slot in OUT.trkmc.ar order; nullptr = flank pass
    uint64_t n;
    uint64_t* nentries;       // += number of distinct (k-mer, locus) entries created
};

template <class X>
DBTK_HD void body_cls_insert(X& x, const ClsBuildArgs& a) {
    for (uint64_t i = (uint64_t)x.bid() * x.nthreads() + x.tid(); i < a.n; i += (uint64_t)x.nblocks() * x.nthreads()) {
        uint32_t lo = 0, hi = a.nloci;  // locus of entry i: beg[l] <= i < beg[l+1]
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (a.beg[mid] <= i) lo = mid; else hi = mid;
        }
        const uint32_t locus = lo;
        const uint64_t kmer = a.ks[i];
        const uint32_t cls = a.outslot ? (uint32_t)a.outslot[i] : CLS_FLANK;
        const uint64_t lc = ((uint64_t)locus << 32) | cls;
        uint64_t s = hash_cls(kmer, locus, a.shift);
        for (;;) {
            const uint64_t prev = x.atomic_cas(&a.slots[s].kmer, NAN64, kmer);
            if (prev == NAN64 || prev == kmer) {
                const uint64_t plc = x.atomic_cas(&a.slots[s].lc, ~0ull, lc);
                if (plc == ~0ull) { x.atomic_add(a.nentries, 1ull); break; }  // slot is ours
                if ((uint32_t)(plc >> 32) == locus) {  // same (k-mer, locus): flank overrides TR
                    if (cls == CLS_FLANK) x.atomic_or(&a.slots[s].lc, 0xFFFFFFFFull);
                    break;
                }
            }
            s = (s + 1) & a.mask;
        }
    }
}

// ------------------------------------------------------------- batch args --
struct BatchArgs {
    DevTables T;
    dbtk_params_t P;
    const uint8_t* seq;   // reads back to back; read r = [off[r], off[r+1])
    const uint64_t* off;  // 2*npairs + 1
    uint64_t seq_len;     // bytes readable at seq
    uint64_t npairs;
    uint32_t* surv;       // K1 -> pair kernel: indices of pairs that passed subfilter
    uint32_t* nsurv;
    uint64_t* counts;     // OUT.trkmc.ar order
    uint64_t* kmc;
    uint64_t* nmapread;   // widened; low 32 bits are the reference's uint32 counter
    uint64_t* counters;   // DBTK_C_*
    uint64_t* ctr_rep;    // nullptr, or CTR_REP replicas of the counters, CTR_STRIDE words apart (see counters_of)
    dbtk_pair_rec_t* recs;  // nullptr, or rec_cap records
    uint32_t* nrec;         // kam mode: compaction counter (may exceed rec_cap)
    uint32_t rec_cap;
    uint32_t* errflag;
    uint64_t* vote_scratch;  // a pool of rows of nloci+1 stamped hit words (see vote): taken by a workgroup for the one pair that needs it
    uint32_t* vote_epoch;    // per row
    uint64_t* vote_busy;     // per row: taken
    uint32_t vote_rows;
    uint64_t* dbg;           // diagnostic build only (-DDBTK_STAMPS): per-phase cycle sums of k_pair
    // K2 -> K3, per (survivor, mate) row of nkp positions: the index results (val, aux), and per row the number of
    // positions and the read's offset.  The canonical k-mers do NOT travel: only the general resolve kernel needs them, for
    // an eighth of the pairs, and re-encodes them from the read.
    // The index results of a row, split: `hitaux` (class at the locus, AUX_MISS where the k-mer is not in the index) is always
    // written; `hitval` only for a read whose found k-mers are NOT all unique to one and the same locus — for the others
    // (most reads from a locus) the one index value, the number of found positions and the flag sit in `hithdr`
    // (val | found << 32 | 1 << 63), the usual-pair kernel reads 4 bytes per position and decides from the two headers.
    uint32_t* hitaux;
    uint32_t* hitval;
    uint64_t* hithdr;        // [2 * tcap]
    uint32_t* hitnk;         // [2 * tcap]
    uint64_t* hitoff;        // [2 * tcap]: where the read starts in seq (the general resolve kernel re-encodes its k-mers from there)
    uint32_t nkp;            // positions reserved per read in the hit buffers (multiple of 64)
    uint32_t* gen_list;      // K3a -> K3b: survivors (t) that need the general resolve kernel; nullptr: K3b takes every survivor
    uint32_t* ngen;
    uint32_t pair_base;      // index of this sub-batch's first pair inside the caller's batch (records)
    const uint8_t* qual;     // base qualities (same offsets as seq), nullptr for FASTA; only read when P.bait
    uint64_t* edgebuf;       // -bu: K2 -> K3 canonical (k+1)-mers [survivor][mate][nkp]
    uint64_t* qmaskbuf;      // -b with qualities: K2 -> K3 k-mer quality masks [survivor][mate][4 x u64]
    struct BubEvent* events; // -bu: novel-edge log
    uint32_t* nevents;
    uint32_t events_cap;
    uint32_t t0, tcap;       // K2/K3 work on survivors [t0, min(t0 + tcap, *nsurv)): the hit buffer holds tcap pairs
    uint32_t* walk_dst;      // threading = 2 (v1.3): per survivor, destLocus of a pair that reaches threading (else left NAN32);
                             // the walk kernel (dbtk_walk.h: body_walk_pairs) takes it from there
    const uint32_t* sel;     // the lean probe kernel only: nullptr, or the chunk-relative indices of the pairs it is to look up (the pairs the
    const uint32_t* nsel;    //   locus-resident kernel, dbtk_locus.h, does not take) and their number
    uint64_t* pstats;        // nullptr, or the context's path statistics (dbtk.h: dbtk_ctx_path_stats)
    uint32_t* sortflag;      // nullptr, or "the survivor list is in locus order" (dbtk_probe2.h: SurvSortArgs::flag): the encode stage clears it
                             // (body_surv_key, when it runs, decides), so that a batch that is not sorted needs no memset between two kernels
    uint32_t* hint_out;      // nullptr, or the host's pinned hint words (launch_batch: sort_hint, locus_hint): [0] survivors, [1] pairs the lean probe kernel took, [6] sort flag of this
                             // batch, written by the general resolve kernel (a copy engine's round between two kernels cost 16 us of a 1.2-ms step)
    uint32_t* skey;          // nullptr, or [npairs]: the encode stage also writes, beside each survivor, the sort key body_surv_key would look up again — the
                             // locus of the pair's first sampled k-mer of mate 1 that is in the index (it has just found it: the table line is in its L1)
    uint32_t k1_xcd;         // the encode stage cuts its tiles into one contiguous range per XCD (body_encode_subfilter); 0 = plain stride (DBTK_K1_XCD=0, the emulator)
    uint32_t vzero;          // always 0: `lane * vzero` makes an address look lane-dependent, so that a load whose value is only
                             // needed an iteration later is not turned into scalars (and waited for) right where it is issued
};

// Thousands of waves each flush a handful of counters at their end; atomics on ONE address serialize at the memory side
// (tens of nanoseconds each), which showed up as a fixed cost per resident wave.  On the device the waves therefore add
// into CTR_REP replicas on separate cache lines, and a tiny kernel folds the replicas into the real counters after the batch.
constexpr uint32_t CTR_REP = 256, CTR_STRIDE = 32;
static_assert(CTR_REP == W_CTR_REP && CTR_STRIDE == W_CTR_STRIDE, "dbtk_walk.h addresses the same counter replicas");
constexpr uint32_t AUX_MISS = 0xFFFFFFFDu;  // hitaux: the position's k-mer is not in the index (next to CLS_NONE, CLS_FLANK)
constexpr uint64_t HDR_UNIFORM = 1ull << 63;
template <class X>
DBTK_HD uint64_t* counters_of(X& x, const BatchArgs& a) {
    return a.ctr_rep ? a.ctr_rep + (size_t)(x.bid() & (CTR_REP - 1)) * CTR_STRIDE : a.counters;
}

// In-kernel stamps (cdna_hip_programming.md 7): only in the separate diagnostic
// library built with -DDBTK_STAMPS; the product build compiles them away.
#ifdef DBTK_STAMPS
#define DBTK_STAMP_DECL uint64_t st_acc[48] = {0}; uint64_t st_last = x.clock();
#define DBTK_STAMP(i) do { const uint64_t now_ = x.clock(); st_acc[i] += now_ - st_last; st_last = now_; } while (0)
#define DBTK_STAMP_FLUSH do { if (lane == 0 && a.dbg) for (int i_ = 0; i_ < 48; ++i_) if (st_acc[i_]) x.atomic_add(&a.dbg[i_], st_acc[i_]); } while (0)
#else
#define DBTK_STAMP_DECL
#if defined(DBTK_MARKS) && defined(__HIP_DEVICE_COMPILE__)
#define DBTK_STAMP(i) asm volatile("; MARK " #i)  // (ISA reading aid: the phase boundaries as comments in the -S output)
#else
#define DBTK_STAMP(i) do { } while (0)
#endif
#define DBTK_STAMP_FLUSH do { } while (0)
#endif

// ----------------------------------------------------------------- records --
DBTK_HD void mate_rec_init(dbtk_mate_rec_t* m) {
    m->si = -1; m->ei = -1; m->si_ = -1; m->ei_ = -1; m->nt = 0; m->bs = 0; m->ti = -1;
    m->kf = 0; m->hf = 0; m->bf = 0; m->qf = 0; m->af = 0; m->rm = 0;
    m->nk = 0;
}
// Record of a pair that stops before the pair kernel (trace mode only).
DBTK_HD void write_early_rec(dbtk_pair_rec_t* r, uint32_t pair, uint32_t stage, uint32_t nloci) {
    r->pair = pair; r->stage = stage; r->dst = nloci; r->dst0 = NAN32; r->nm1 = 0; r->nm2 = 0;
    mate_rec_init(&r->r1);
    mate_rec_init(&r->r2);
    for (int i = 0; i < MAXL / 4; ++i) { r->r1.as2[i] = 0; r->r2.as2[i] = 0; }
}

template <int P0, int P1, int P2, int P3, class X>
DBTK_HD uint64_t quad_perm64(X& x, uint64_t v) {
    const uint32_t lo = x.template quad_perm<P0, P1, P2, P3>((uint32_t)v), hi = x.template quad_perm<P0, P1, P2, P3>((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
// The 16 bytes of bucket b that lane part `part` (0..3) of a cooperating lane group reads: keys 0,1 | keys 2,3 | values 0,1 | values 2,3.
DBTK_HD void bucket_part(const IdxBucket* idx, uint64_t b, uint32_t part, uint64_t* a0, uint64_t* a1) {
    const uint4 q = reinterpret_cast<const uint4*>(idx + b)[part];
    *a0 = ((uint64_t)q.y << 32) | q.x;
    *a1 = ((uint64_t)q.w << 32) | q.z;
}

// ======================================================================= K1 =
// encode + subfilter (read2kmers_edges' validity + subfilter,
// src/aQueryFasta_thread.h:274-311, src/aQueryFasta_thread.cpp:172-188, 2035-2051).
// One wavefront per block, no block barriers: a wave takes tiles of K1_TP = 16
// consecutive pairs (their bytes are contiguous in the batch), fetches them with
// coalesced 16-byte loads, packs them to 2 bits + 1 validity bit per base in its
// LDS slice, and probes the N_FILTER sampled windows of mate 1 with 4 lanes per
// pair (sample s on lane 4*pair + s%4); the group verdicts come from one ballot.
// Mate 2 is probed only for pairs whose mate 1 passed — the probes the reference
// performs.  The wave is software-pipelined over its tiles: while tile t is being
// probed (one HBM round trip per probe, see IdxBucket) the bytes and read offsets
// of tile t+1 are already in flight into registers and the tile geometry of t+2
// into SGPRs, so the streaming and the random traffic overlap inside every wave.
constexpr int K1_NT = 64;
constexpr int K1_TP = 16;                              // pairs per tile
constexpr int K1_CH = K1_TP * 2 * MAXL / 16 + 4;       // 16-base chunks per tile (+ slack)
constexpr int K1_SBF = 240;                            // survivors buffered per wave before one atomic appends them (round 6: 48 -> 240: in a batch that hits,
                                                       // 5 M survivors were 104 000 atomics on ONE word, 14 ns each = the kernel's 1.5 ms; 496: no further gain)
constexpr int K1_PF = 5;                               // chunks per lane fetched ahead (320 chunks: a tile of 150 bp pairs has <= 301)
struct K1Smem {
    uint32_t pk[K1_CH];
    uint16_t vd[K1_CH];
    uint16_t rb[2 * K1_TP], rl[2 * K1_TP];  // per read of the tile: first base (relative to the tile's A0), length
    uint32_t sbuf[K1_SBF + K1_TP];          // survivors not yet appended to the global list
    uint32_t kbuf[K1_SBF + K1_TP];          // ... and their sort keys (BatchArgs::skey)
    uint32_t kgrp[K1_TP];                   // sort key of each pair of the tile in work
};


// true iff bases [b, b+len) of the stream contain a run of >= k valid bases
DBTK_HD bool any_valid_window(const uint16_t* vd, uint32_t b, uint32_t len, uint32_t k) {
    if (len < k) return false;
    uint32_t nvalid = 0, run = 0;
    bool found = false;
    for (uint32_t c = b >> 4; c <= (b + len - 1) >> 4; ++c) {
        uint32_t bits = vd[c];
        const uint32_t c0 = c << 4;
        if (c0 < b) bits &= 0xFFFFu >> (b - c0);                        // drop bases before b
        if (c0 + 16 > b + len) bits &= 0xFFFFu << (c0 + 16 - (b + len));  // and after the read
        nvalid += (uint32_t)__builtin_popcount(bits);
    }
    if (nvalid == len) return true;  // no N: the common case
    if (nvalid < k) return false;
    for (uint32_t i = 0; i < len && !found; ++i) {
        const uint32_t p = b + i;
        const bool v = (vd[p >> 4] >> (15 - (p & 15))) & 1;
        run = v ? run + 1 : 0;
        found = run >= k;
    }
    return found;
}

// The last, partial 16 bytes of the batch at g as four little-endian words; bytes past the end read as 0.
// Rare (one chunk per batch): kept as a rolled loop off the hot path.
DBTK_HD void load_tail_chunk(const uint8_t* seq, uint64_t seq_len, uint64_t g, uint32_t w[4]) {
    w[0] = w[1] = w[2] = w[3] = 0;
#pragma nounroll
    for (uint32_t b = 0; b < 16 && g + b < seq_len; ++b) w[b >> 2] |= (uint32_t)seq[g + b] << (8 * (b & 3));
}

// LAZY: the form for a batch that hits (k_encode_subfilter_lazy; the launcher's hint: more than half of the batch before passed) — see "LAZY" below.
// A template parameter, not an argument: the form every WGS-like batch runs must not carry the second pass' registers.
template <bool LAZY = false, class X>
DBTK_HD void body_encode_subfilter(X& x, const BatchArgs& a) {
    uint64_t* const ctr = counters_of(x, a);
    using SM = K1Smem;
    SM& sm = *x.template smem<SM>();
    const uint32_t lane = (uint32_t)x.lane();
    const uint32_t k = a.P.ksize, NF = a.P.n_filter, NM = a.P.nm_filter;
    const bool dosub = NF && NM;
    const uint32_t grp = lane >> 2, sub = lane & 3;  // pair of the tile this lane works for, sample phase
    uint32_t c_short = 0, c_sub = 0, c_nhash = 0, c_probe = 0, c_surv = 0;  // per-lane partial sums (a wave sees < 2^32 pairs)
    uint64_t c_bases = 0;
    if (lane == 0 && x.bid() == 0) {
        x.atomic_add(&ctr[DBTK_C_NREADS], 2 * a.npairs);  // nReads, AQ.cpp:1977
        if (a.sortflag) *a.sortflag = 0u;
    }
    const uint64_t ntiles = (a.npairs + K1_TP - 1) / K1_TP;
    const int lane_ = (int)lane; (void)lane_;
    // Bytes readable at a.seq.  The device entry point does not know the batch's length on the host (~0): the contract there is
    // "readable up to the end of the last read rounded up to 16" (dbtk.h), so that is what the fetch guards work with.
    const uint64_t seq_len = a.seq_len != ~0ull ? a.seq_len : ((a.off[2 * a.npairs] + 15) & ~15ull);
    DBTK_STAMP_DECL
    // Tile geometry comes from two offsets; they are requested two tiles ahead and only turned into
    // (A0, nch) when their tile becomes the one being fetched, so nothing waits on them.
    auto tile_p0 = [&](uint64_t tile) { return tile * K1_TP; };
    auto tile_np = [&](uint64_t tile) { const uint64_t p0 = tile * K1_TP; return (uint32_t)((a.npairs - p0 < (uint64_t)K1_TP) ? a.npairs - p0 : K1_TP); };
    uint32_t w[K1_PF][4];        // this lane's chunks lane, lane+64, ... of the tile in flight
    uint64_t ro0 = 0, ro1 = 0;   // lane r < 2*np: offsets of read r of the tile in flight
    // Straight-line loads only (no per-load branches, or the compiler waits for each one in turn), addressed as a
    // wave-uniform base plus a 32-bit lane offset.  Chunks past the tile are simply read (they are the next tile's bytes;
    // the pack step never looks at them); a chunk that would cross the end of the batch reads the base instead and the
    // pack step replaces it.  Precondition (kept by the host side): 16 bytes are readable at a.seq.
    auto fetch = [&](uint64_t p0, uint32_t np, uint64_t A0, uint32_t nchf) {  // nchf: chunks of the tile (the lanes past them re-read its first chunk:
        const bool inb = A0 + 16 <= seq_len;                                 // the fixed-size fetch asked for 2.4 lines of the NEXT tile per tile, round 6)
#ifdef DBTK_STAMPS
        const uint8_t* base = a.seq + ((inb && !(a.P.diag & 2)) ? A0 : 0ull);  // (knob 2: no streaming traffic)
#else
        const uint8_t* base = a.seq + (inb ? A0 : 0ull);
#endif
        const uint64_t rem64 = inb ? seq_len - A0 : 0ull;
        const uint32_t rem = rem64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)rem64;  // bytes readable from base (saturated)
#pragma unroll
        for (int j = 0; j < K1_PF; ++j) {
            const uint32_t cj = lane + 64u * j, o = 16u * cj;
            const uint4 q = *reinterpret_cast<const uint4*>(base + ((o + 16u <= rem && cj < nchf) ? o : 0u));
            w[j][0] = q.x; w[j][1] = q.y; w[j][2] = q.z; w[j][3] = q.w;
        }
        const uint32_t r = lane < 2 * np ? lane : 2 * np - 1;
        ro0 = a.off[2 * p0 + r]; ro1 = a.off[2 * p0 + r + 1];
    };
    uint32_t nsb = 0;  // survivors waiting in sm.sbuf (wave-uniform)
    auto flush_survivors = [&]() {
        x.sync();
        uint32_t base = 0;
        if (lane == 0) base = x.atomic_add(a.nsurv, nsb);
        base = x.bcast(base, 0);
        for (uint32_t i = lane; i < nsb; i += K1_NT) a.surv[base + i] = sm.sbuf[i];
        if (a.skey) for (uint32_t i = lane; i < nsb; i += K1_NT) a.skey[base + i] = sm.kbuf[i];
        x.sync();
    };
    // Which tiles a wave takes.  Workgroup b runs on XCD b % 8 (round robin), and every XCD has an L2 of its own: with tile = b, b + nblocks, ...
    // the 128-byte line two neighbouring tiles share, and the few chunks a tile's fixed-size fetch reads past its end, were asked for twice —
    // by two XCDs.  So the tiles are cut into eight contiguous ranges, one per XCD, and the XCD's workgroups stride through their range:
    // neighbours in the batch are worked on at about the same time under the same L2: 0.654 -> 0.649 ms per 10 M reads (a contiguous run
    // of tiles per wave instead: 0.656, no gain).  (Fewer than 8 workgroups: the plain stride.)
    const uint64_t nbk = x.nblocks();
    const bool xcd_map = nbk >= 8 && a.k1_xcd;
    const uint64_t xcd = x.bid() & 7u, t8 = (ntiles + 7) / 8;
    const uint64_t rlo = xcd_map ? (xcd * t8 < ntiles ? xcd * t8 : ntiles) : 0;
    const uint64_t rhi = xcd_map ? (rlo + t8 < ntiles ? rlo + t8 : ntiles) : ntiles;  // this wave's tiles: rlo + its place among its XCD's workgroups, + stride, ... < rhi
    const uint64_t stride = xcd_map ? (nbk - xcd + 7) / 8 : nbk;
    uint64_t tile = xcd_map ? rlo + (x.bid() >> 3) : x.bid();
    uint64_t cB0 = 0, cB1 = 0;  // first/last offset of the current tile (wave-uniform)
    uint64_t nB0 = 0, nB1 = 0;  // ... of the next one: in flight in vector registers, made uniform only when their tile comes up
    const uint64_t lz = (uint64_t)lane * a.vzero;  // 0
    auto uni64 = [&](uint64_t v) { return ((uint64_t)x.uni((uint32_t)(v >> 32)) << 32) | x.uni((uint32_t)v); };
    if (tile < rhi) {
        cB0 = a.off[2 * tile_p0(tile)]; cB1 = a.off[2 * (tile_p0(tile) + tile_np(tile))];
        fetch(tile_p0(tile), tile_np(tile), cB0 & ~15ull, (uint32_t)((cB1 - (cB0 & ~15ull) + 15) >> 4));
    }
    if (tile + stride < rhi) { nB0 = a.off[2 * tile_p0(tile + stride) + lz]; nB1 = a.off[2 * (tile_p0(tile + stride) + tile_np(tile + stride)) + lz]; }
    for (; tile < rhi; tile += stride) {
        const uint64_t p0 = tile_p0(tile), A0 = cB0 & ~15ull;
        const uint32_t np = tile_np(tile), nch = (uint32_t)((cB1 - A0 + 15) >> 4);
        const bool toolong = nch + 3 > (uint32_t)K1_CH;  // a read longer than DBTK_MAX_READ_LEN slipped through
        if (toolong && lane == 0) *a.errflag = DBTK_ERR_READ_TOO_LONG;
        x.sync();  // the previous tile's LDS is dead
        DBTK_STAMP(16);  // tile set-up
        // A: pack the tile.  Fast form: 2-bit codes only, plus "some byte is not ACGT" per lane.
        uint32_t bad = 0;
        const bool tile_tail = A0 + 16ull * nch > seq_len;  // only the batch's last tile can hold the partial last chunk
        if (!toolong) {
#pragma unroll
            for (int j = 0; j < K1_PF; ++j) {
                const uint32_t c = lane + 64u * j;
                if (c < nch + 3) {  // (chunks nch .. nch+2 are padding: whatever was fetched will do)
                    if (tile_tail && c < nch && A0 + 16ull * c + 16 > seq_len) {  // (through a temporary: w must stay in registers)
                        uint32_t t[4];
                        load_tail_chunk(a.seq, seq_len, A0 + 16ull * c, t);
                        w[j][0] = t[0]; w[j][1] = t[1]; w[j][2] = t[2]; w[j][3] = t[3];
                    }
                    uint32_t b = 0;
                    sm.pk[c] = pack16_fast(w[j], &b);
                    if (c < nch) bad |= b;
                }
            }
            for (uint32_t c = lane + 64u * K1_PF; c < nch + 3; c += K1_NT) {  // reads longer than 150 bp: the rest of the tile, exact form
                uint32_t t[4] = {0, 0, 0, 0}, vd = 0;
                const uint64_t gb = A0 + 16ull * c;
                if (c < nch) {
                    if (gb + 16 <= seq_len) { const uint4 q = *reinterpret_cast<const uint4*>(a.seq + gb); t[0] = q.x; t[1] = q.y; t[2] = q.z; t[3] = q.w; }
                    else load_tail_chunk(a.seq, seq_len, gb, t);
                }
                sm.pk[c] = pack16(t, &vd);
                sm.vd[c] = (uint16_t)vd;
                if (c < nch && vd != 0xFFFFu) bad = 1;
            }
        }
        // Every base of the tile is ACGT (the usual case): every window is valid and no validity bits are needed.
        // Otherwise (an N somewhere, or the zero padding after the batch's last read) compute them exactly.
        const bool clean = x.ballot(bad != 0) == 0;
        if (!clean && !toolong) {
#pragma unroll
            for (int j = 0; j < K1_PF; ++j) {
                const uint32_t c = lane + 64u * j;
                if (c < nch + 3) {
                    uint32_t vd = 0;
                    (void)pack16(w[j], &vd);
                    sm.vd[c] = (uint16_t)vd;
                }
            }
        }
        const uint64_t o0 = ro0, o1 = ro1;
        // the next tile's bytes, and the geometry of the one after, start their trip now
        const uint64_t t1 = tile + stride, t2 = t1 + stride;
        cB0 = uni64(nB0); cB1 = uni64(nB1);  // (requested one iteration ago)
        {   // unconditional (a dummy fetch of nothing past the last tile) so that the loads land straight in w/ro
            const bool h1 = t1 < rhi, h2 = t2 < rhi;
            const uint64_t A1 = cB0 & ~15ull;
            fetch(h1 ? tile_p0(t1) : 0, h1 ? tile_np(t1) : 1u, h1 ? A1 : 0ull, h1 ? (uint32_t)((cB1 - A1 + 15) >> 4) : 0u);
            nB0 = a.off[(h2 ? 2 * tile_p0(t2) : 0) + lz]; nB1 = a.off[(h2 ? 2 * (tile_p0(t2) + tile_np(t2)) : 0) + lz];
        }
        if (toolong) continue;
        DBTK_STAMP(17);  // pack + issue of the next fetch
        // B: per-read geometry and "has a valid window" (caks.size() != 0, AQ.cpp:2037): lane r < 2*np owns read r
        bool myany = false;
        if (lane < 2 * np) {
            const uint32_t len = (uint32_t)(o1 - o0);
            sm.rb[lane] = (uint16_t)(o0 - A0);
            sm.rl[lane] = (uint16_t)len;
            c_bases += len;
        }
        x.sync();
        if (lane < 2 * np) myany = clean ? (uint32_t)(o1 - o0) >= k : any_valid_window(sm.vd, sm.rb[lane], sm.rl[lane], k);
        const uint64_t anym = x.ballot(myany);                       // bit r: read r has a window
        const bool gvalid = grp < np;
        const bool gany = gvalid && ((anym >> (2 * grp)) & 3) == 3;  // both mates of my pair
        DBTK_STAMP(18);  // valid-window test
        // C/D: sampled probes; hm[mate] = hit mask of the sampled positions of my pair (same on its 4 lanes)
        uint32_t hm[2] = {0, 0};
        if (dosub) {
            for (int mate = 0; mate < 2; ++mate) {
                const bool go = gany && (mate == 0 || (uint32_t)__builtin_popcount(hm[0]) >= NM);
                if (mate == 1 && x.ballot(go) == 0) break;  // nobody's mate 1 passed: the usual case
                uint32_t bpos = 0, L = 1, S = 0;
                if (go) {
                    bpos = sm.rb[2 * grp + mate];
                    L = (uint32_t)sm.rl[2 * grp + mate] - k + 1;
                    S = L / (NF - 1);
                }
                for (uint32_t s0 = 0; s0 < NF; s0 += 4) {
                    const uint32_t sidx = s0 + sub;
                    uint64_t km_all = NAN64;
                    if (go && sidx < NF) {
                        const uint32_t pos = (sidx != NF - 1) ? sidx * S : L - 1;
                        km_all = clean ? window_kmer_clean(sm.pk, bpos + pos, k) : window_kmer(sm.pk, sm.vd, bpos + pos, k, nullptr, nullptr);
                    }
#ifdef DBTK_STAMPS
                    if (a.P.diag & 1) km_all = NAN64;  // diagnostic: no probes
#endif
                    // LAZY: sample 0 alone first — filter word, table line —
                    // and the other three only for a mate whose sample 0 is not in the index.  In a batch that hits, the first sample is
                    // there nine times in ten and the other three filter words were requests for nothing (8 + 2 per pair, now 2 + 2: the
                    // kernel runs at the chip's request ceiling).  A WGS-like batch would pay a second dependent round trip per tile: never lazy there.
                    const bool lazy = LAZY && NM == 1 && a.T.flt && NF <= 4;
                    uint32_t hits = 0;
                    for (uint32_t ps = 0; ps < (lazy ? 2u : 1u); ++ps) {
                    uint64_t km = km_all;
                    if (lazy && ((ps == 0) != (sub == 0))) km = NAN64;  // (pass 0: sample 0; pass 1: the others ...
                    if (lazy && hits) km = NAN64;                        //  ... of a mate still without a hit)
                    const uint64_t hmix = hash_mix(km);
                    if (a.T.flt && km != NAN64) {  // presence filter: "no" is final, and needs no HBM line
                        const uint64_t fm = kmix(km, k), fb = flt_bits(fm);
                        if ((a.T.flt[flt_word(fm, k, a.T.flt_logw)] & fb) != fb) km = NAN64;
                    }
                    const uint32_t hb = km != NAN64 ? (uint32_t)(hmix >> a.T.idx_shift) : 0u;
                    // Which of the samples the filter let through are looked up in the table.  subfilter stops at its NM-th hit
                    // (AQ.cpp:176-180), and everything after is never read from hm below: with NM == 1 — the default — only the FIRST
                    // sample that may be in the index is looked up, and the next one only if that was one of the filter's false
                    // positives.  A pair from a locus then costs one table line per mate instead of four (all-hit batches: the
                    // encode kernel's table reads fall to a quarter); a background pair costs what it did (its samples end at the
                    // filter), and one tile in twenty-five makes a second turn.
                    const bool seq = NM == 1 && a.T.flt;  // (without the filter every sample "may be": all at once, as before)
                    uint32_t mq = (uint32_t)(x.ballot(km != NAN64) >> (lane & ~3u)) & 0xFu;  // my pair's samples still to be looked up
                    if (seq && hm[mate]) mq = 0;  // (NF > 4: a hit among the first four samples has ended the loop)
                    for (;;) {
                    const uint32_t act = seq ? (mq & (0u - mq)) : mq;  // this turn's samples
                    const uint64_t kmt = ((act >> sub) & 1u) ? km : NAN64;
                    // Two rounds; in round r the lane pairs {0,1} and {2,3} of the group look up samples 2r and 2r+1:
                    // each lane reads two of the bucket's four keys, so one load instruction fetches the keys of 32
                    // probes and a probe is ONE request to the memory system.
                    uint32_t hits_t = 0;
                    uint64_t kq[2], k0[2], k1[2];
                    uint32_t bq[2];
                    kq[0] = quad_perm64<0, 0, 1, 1>(x, kmt); kq[1] = quad_perm64<2, 2, 3, 3>(x, kmt);
                    {
                        const uint32_t hbt = kmt != NAN64 ? hb : 0u;
                        bq[0] = x.template quad_perm<0, 0, 1, 1>(hbt); bq[1] = x.template quad_perm<2, 2, 3, 3>(hbt);
                    }
#pragma unroll
                    for (int r = 0; r < 2; ++r)  // both rounds' loads in flight together: unconditional (bucket 0 for a lane without a probe)
                        bucket_part(a.T.idx, bq[r], sub & 1, &k0[r], &k1[r]);
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        bool open = kq[r] != NAN64, hitl = false;
                        uint64_t hmask;
                        for (;;) {
                            bool again = false;
                            if (open) {
                                hitl = k0[r] == kq[r] || (k1[r] & ~IDX_OVF) == kq[r];
                                again = (sub & 1) && k1[r] != NAN64 && (k1[r] & IDX_OVF);  // key[3]: full, and an insert walked past
                            }
                            hmask = x.ballot(hitl);
                            const uint64_t amask = x.ballot(again);
                            open = open && ((hmask >> (lane & ~1u)) & 3) == 0 && ((amask >> (lane | 1u)) & 1);
                            if (x.ballot(open) == 0) break;
                            if (open) {  // rare: on to the next bucket
                                bq[r] = (bq[r] + 1) & (uint32_t)a.T.idx_mask;
                                bucket_part(a.T.idx, bq[r], sub & 1, &k0[r], &k1[r]);
                            }
                        }
                        const uint32_t g4 = (uint32_t)(hmask >> (lane & ~3u)) & 0xF;
                        hits_t |= ((g4 & 3) ? 1u : 0u) << (2 * r) | ((g4 >> 2) ? 1u : 0u) << (2 * r + 1);
                        // (the sort key: mate 1's first sample found in the index — in this form one sample per pair is looked up per turn, in
                        // sample order, so the lane that holds the matching key holds THE sample: its value is 32 bytes further in the line)
                        if (a.skey && seq && mate == 0 && hitl) {
                            const uint32_t slot = 2 * (sub & 1) + (k0[r] == kq[r] ? 0u : 1u);
                            const uint32_t v = (uint32_t)a.T.idx[bq[r]].val[slot];
                            uint32_t key = (v & 1u) ? a.T.vv[(v >> 1) + 1] : v >> 1;
                            sm.kgrp[grp] = key > a.T.nloci ? a.T.nloci : key;
                        }
                    }
                    hits |= hits_t;
                    mq &= ~act;
                    if (!seq || hits_t) mq = 0;  // all looked up / the first hit found: this pair is done
                    if (x.ballot(mq != 0) == 0) break;  // (a false positive of the filter somewhere in the tile: its pair's next sample)
                    }
                    if (!lazy || x.ballot(go && !hits) == 0) break;  // (lazy: a second pass only for a tile with a mate still undecided)
                    }
                    hm[mate] |= hits << s0;
                }
            }
        }
        DBTK_STAMP(19);  // sampled probes
        // E: verdicts (computed on every lane of a pair's group, counted on its first)
        bool pass = false;
        {
            const bool lead = gvalid && sub == 0;
            uint32_t stage = 0xFFFFFFFFu;
            if (!gany) {
                stage = DBTK_STAGE_SHORT;
                c_short += lead;
            } else if (dosub) {
                // subfilter's loop (AQ.cpp:176-180) stops at the NM-th hit, at sample p: p+1 probes, ++nhash p times;
                // without NM hits it runs over all NF samples
                uint32_t nhash = 0, nprobe = 0;
                bool rej = false;
                for (int mate = 0; mate < 2 && !rej; ++mate) {
                    uint32_t r = hm[mate];
                    for (uint32_t i = 1; i < NM; ++i) r &= r - 1;
                    const uint32_t pth = r ? (uint32_t)__builtin_ctz(r) : NF;
                    nhash += pth; nprobe += r ? pth + 1 : NF;
                    rej = r == 0;
                }
                if (lead) { c_nhash += nhash; c_probe += nprobe; }
                if (rej) { stage = DBTK_STAGE_SUBFILTER; c_sub += lead ? 2 : 0; }
            }
            pass = lead && stage == 0xFFFFFFFFu;
            if (lead && !pass && a.P.trace && a.recs) {
                const uint32_t pair = (uint32_t)(p0 + grp) + a.pair_base;
                write_early_rec(&a.recs[pair], pair, stage, a.T.nloci);
            }
        }
        const uint64_t pm = x.ballot(pass);
        if (pm) {  // survivors go to the wave's LDS buffer (pair order inside the tile); one atomic per K1_SBF of them
            if (pass) {
                const uint32_t at = nsb + (uint32_t)__builtin_popcountll(pm & ((1ull << lane) - 1));
                sm.sbuf[at] = (uint32_t)(p0 + grp);
                if (a.skey) sm.kbuf[at] = sm.kgrp[grp];
                ++c_surv;
            }
            nsb += (uint32_t)__builtin_popcountll(pm);
            if (nsb >= (uint32_t)K1_SBF) { flush_survivors(); nsb = 0; }
        }
        DBTK_STAMP(20);  // verdict + survivor append
    }
    if (nsb) flush_survivors();
    {
        const int lane = lane_; (void)lane;
        DBTK_STAMP_FLUSH;
    }
    // flush: wave sums, one atomic each
    const uint32_t s_short = x.wave_sum((uint32_t)c_short), s_sub = x.wave_sum((uint32_t)c_sub), s_surv = x.wave_sum((uint32_t)c_surv);
    const uint32_t s_nhash = x.wave_sum((uint32_t)c_nhash), s_probe = x.wave_sum((uint32_t)c_probe);
    const uint32_t b_lo = x.wave_sum((uint32_t)(c_bases & 0xFFFFF)), b_hi = x.wave_sum((uint32_t)(c_bases >> 20));
    if (lane == 0) {
        if (s_short) x.atomic_add(&ctr[DBTK_C_NSHORT], (uint64_t)s_short);
        if (s_sub) x.atomic_add(&ctr[DBTK_C_SUBFILTERED], (uint64_t)s_sub);
        if (s_nhash) x.atomic_add(&ctr[DBTK_C_NHASH0], (uint64_t)s_nhash);
        if (s_probe) x.atomic_add(&ctr[DBTK_C_ALGO_PROBES], (uint64_t)s_probe);
        if (s_surv) x.atomic_add(&ctr[DBTK_C_SURVIVORS], (uint64_t)s_surv);
        const uint64_t bases = (uint64_t)b_lo + ((uint64_t)b_hi << 20);
        if (bases) x.atomic_add(&ctr[DBTK_C_BASES], bases);
    }
}


// ================================================================= K2 .. K4 =
// One wavefront (64-thread block) per surviving pair.
constexpr int NSLOT = NKMAX / 64;  // upper bound of k-mer positions per lane per mate
constexpr int NBKT = 256;          // buckets of the hit-list sort (top 8 bits of the k-mer: monotone in the key)
// LDS of the resolve kernel, sized by NS = 64-position slots per read (3 for 150 bp reads): NH hit
// entries per pair, LC slots in the per-pair locus map of the vote (which spills to vote_scratch).
template <int NS>
struct PairSmemT {
    static constexpr int NK = 64 * NS, NH = 128 * NS, LC = NS >= 3 ? 512 : 256;  // LC >= NH: the map doubles as sort scratch
    uint32_t hval[2][NK];      // index val by read position (NOHIT = not in the index)
    union {
        struct { uint64_t skey[NH]; uint16_t sinfo[NH]; } s;                      // hit list being sorted
        struct { uint32_t nml[NH]; uint32_t lkey[LC]; uint16_t ord[NH]; } v;       // vote phase
    } u;
    union {
        struct {
            uint32_t uval[NH];  // unique k-mers in ascending key order: index val
            uint32_t dd[NH];    // PE_KMC dup: count in mate 0 | count in mate 1 << 16
            uint32_t lhit[LC];
            uint16_t poff[NH];  // vote: where a multi-locus k-mer's loci sit in the LDS pool (0xFFFF: read vv in HBM)
        } a;
        struct {                // bucket sort of the hit list (before any of the above is live)
            uint64_t tmpk[NH];
            uint16_t tmpi[NH];
            uint32_t cnt[NBKT];
            uint16_t bst[NBKT];
        } b;
    } w;
    int stack[3 * 40];
    // re-encoding of the pair's reads (deliver): raw bytes from each read's 4-byte-aligned start, then 2-bit packed + validity
    uint32_t raw[2][MAXL / 4 + 6];
    uint32_t pk[2][MAXL / 16 + 4];
    uint16_t vd[2][MAXL / 16 + 4];
    int32_t res[8];            // vote result
    uint32_t evd[2 * NH];      // the vote's loci pool (lists of the multi-locus k-mers), then top - second per event
};

// assignTRkmc's scan, literally (src/aQueryFasta_thread.cpp:1470-1555), over the
// states as[0..nk); ntr = number of TR states, already reduced to uint8_t.  The
// kernels use assign_bits (dbtk_assign.h); this form is what it is checked against.
DBTK_HD void assign_scan(const uint8_t* as, int nk, uint32_t ntr, const dbtk_params_t& P, MateState& r) {
    int s = 0, s_ = 0, s__ = 0;
    int ti2 = -1, si1 = -1, ei1 = -1, si2 = -1, ei2 = -1;
    if (r.rm) { r.nt = -1; r.bs = -1; r.ti = -1; return; }
    for (int i = 0; i < nk; ++i) {
        s = as[i];
        if (s && s__) {
            if (s != s__) {
                ++r.nt;
                if ((uint64_t)(int64_t)r.nt > P.max_nt) { r.af = 1; r.rm = 1; return; }
                if (r.nt == 1) {
                    r.ti = i;
                    if (s_) { si1 = -1; ei1 = -1; }
                } else if (r.nt == 2) {
                    if (r.bs == 2) { r.af = 1; r.rm = 1; return; }
                    ti2 = i;
                    if (s_) { si2 = -1; ei2 = -1; }
                }
            }
        }
        if (!r.bs) { if (s) r.bs = s; }
        if (!s) {
            if (r.nt == 0) { if (!s_) ++ei1; else { si1 = i; ei1 = i + 1; } }
            if (r.nt == 1) { if (!s_) ++ei2; else { si2 = i; ei2 = i + 1; } }
        }
        s_ = s;
        if (s) s__ = s;
    }
    const int ti1 = r.ti;
    if (r.nt == 0) {
        if (r.bs != 2) { r.af = 1; r.rm = 1; return; }
        r.si = 0; r.ei = nk; r.si_ = 0; r.ei_ = nk;
    } else if (r.nt == 1) {
        if (r.bs == 1) {
            r.si = si1 >= 0 ? (si1 + ei1) / 2 : ti1; r.ei = nk;
            r.si_ = si1 >= 0 ? ei1 : ti1; r.ei_ = nk;
        } else {
            r.si = 0; r.ei = si1 >= 0 ? (si1 + ei1) / 2 : ti1;
            r.si_ = 0; r.ei_ = si1 >= 0 ? si1 : ti1;
        }
    } else {
        if (ntr < P.nm_tr) { r.af = 1; r.rm = 1; return; }
        r.si = (si1 >= 0 ? (si1 + ei1) / 2 : ti1);
        r.ei = (si2 >= 0 ? (si2 + ei2) / 2 : ti2);
        r.si_ = ei1 >= 0 ? ei1 : ti1;
        r.ei_ = si2 >= 0 ? si2 : ti2;
    }
}

// Running per-locus hit counts of find_matching_locus (hits1/hits2,
// src/aQueryFasta_thread.cpp:373-375): a small open-addressed map in LDS; when a
// pair touches more than LLIMIT loci it migrates to this block's stamped array
// in HBM (vote_scratch: epoch<<32 | h2<<16 | h1 per locus), which replaces the
// reference's two std::fill(nloci+1) per pair (AQ.cpp:433-434).
struct HitMap {
    uint32_t* lkey; uint32_t* lhit; uint32_t n;
    uint64_t* g; uint32_t epoch; bool spilled;
    uint32_t cap, shift, limit;  // LDS map: cap = 2^(32 - shift) slots, migrate to HBM beyond `limit` loci
};
DBTK_HD uint32_t hitmap_add(HitMap& m, uint32_t locus, uint32_t add) {  // returns the new h1 | h2<<16
    if (!m.spilled) {
        uint32_t i = (locus * 0x9E3779B1u) >> m.shift;
        for (;;) {
            if (m.lkey[i] == locus) { m.lhit[i] += add; return m.lhit[i]; }
            if (m.lkey[i] == NAN32) break;
            i = (i + 1) & (m.cap - 1);
        }
        if (m.n < m.limit) { m.lkey[i] = locus; m.lhit[i] = add; ++m.n; return add; }
        for (uint32_t j = 0; j < m.cap; ++j)  // migrate
            if (m.lkey[j] != NAN32) DBTK_COH_STORE(&m.g[m.lkey[j]], ((uint64_t)m.epoch << 32) | m.lhit[j]);
        m.spilled = true;
    }
    // (the row was last written by whichever workgroup held it before: its words are read and written at the device-coherent level)
    const uint64_t w = DBTK_COH_LOAD(&m.g[locus]);
    const uint32_t cur = ((uint32_t)(w >> 32) == m.epoch) ? (uint32_t)w : 0u;
    const uint32_t nw = cur + add;
    DBTK_COH_STORE(&m.g[locus], ((uint64_t)m.epoch << 32) | nw);
    return nw;
}

struct Asgn { uint64_t idx, fc, rc; };  // asgn_t, AQ.cpp:146-149
DBTK_HD void updatetop2(uint64_t cf, uint32_t ind, uint64_t cr, Asgn& top, Asgn& second) {  // AQ.cpp:331-347
    if (cf + cr > top.fc + top.rc) {
        if (top.idx != ind) { second = top; top.idx = ind; }
        top.fc = cf; top.rc = cr;
    } else if (cf + cr > second.fc + second.rc) {
        if (second.idx != ind) second.idx = ind;
        second.fc = cf; second.rc = cr;
    }
}
DBTK_HD bool get_acm1(uint64_t fc, uint64_t rc, uint64_t rem, uint64_t cth) {  // AQ.cpp:354-357
    return (fc < cth && cth - fc <= rem) || (rc < cth && cth - rc <= rem);
}

// find_matching_locus (src/aQueryFasta_thread.cpp:364-422) on the permuted
// unique list, general form (any number of loci per k-mer); lane 0 only.
// Loci of multi-locus k-mers come from `pool` (LDS, filled in parallel beforehand) at poff[u]
// when they fit there (poff[u] != 0xFFFF), else straight from vv in HBM.
DBTK_HD void vote(const DevTables& T, const uint16_t* ord, const uint32_t* uval, const uint32_t* dd, int nu, uint32_t cth,
                  HitMap& hm, Asgn& top, uint64_t& nvvw, const uint32_t* nml, const uint32_t* pool, const uint16_t* poff) {
    Asgn second{NAN32, 0, 0};
    top = Asgn{NAN32, 0, 0};
    uint64_t total = 0;
    for (int i = 0; i < nu; ++i) total += (dd[i] & 0xFF) + ((dd[i] >> 16) & 0xFF);
    uint64_t rem = total;  // remain[i] = sum of dups after i (countRemain, AQ.cpp:298-306)
    for (int i = 0; i < nu; ++i) {
        const uint32_t u = ord[i], vi = uval[u];
        const uint32_t d1 = dd[u] & 0xFF, d2 = (dd[u] >> 16) & 0xFF;
        const uint32_t add = d1 | (d2 << 16);
        rem -= d1 + d2;
        if (vi & 1) {
            const uint32_t n = nml[u];
            const bool inl = poff[u] != 0xFFFFu;
            nvvw += 1 + n;
            for (uint32_t j = 0; j < n; ++j) {
                const uint32_t locus = inl ? pool[poff[u] + j] : T.vv[(vi >> 1) + 1 + j];
                const uint32_t h = hitmap_add(hm, locus, add);
                updatetop2(h & 0xFFFF, locus, h >> 16, top, second);
            }
        } else {
            const uint32_t locus = vi >> 1;
            const uint32_t h = hitmap_add(hm, locus, add);
            updatetop2(h & 0xFFFF, locus, h >> 16, top, second);
        }
        if (!((top.fc + top.rc - second.fc - second.rc) < rem)) {  // !get_acm2
            int j = i;
            uint64_t remj = rem;
            while (get_acm1(top.fc, top.rc, remj, cth)) {
                if (++j >= nu) break;
                const uint32_t uj = ord[j], vj = uval[uj];
                const uint32_t e1 = dd[uj] & 0xFF, e2 = (dd[uj] >> 16) & 0xFF;
                remj -= e1 + e2;
                if (vj & 1) {
                    const uint32_t n = nml[uj];
                    const bool inl = poff[uj] != 0xFFFFu;
                    nvvw += 1;
                    for (uint32_t q = 0; q < n; ++q) {
                        nvvw += 1;
                        const uint32_t locus = inl ? pool[poff[uj] + q] : T.vv[(vj >> 1) + 1 + q];
                        if (locus == top.idx) { top.fc += e1; top.rc += e2; break; }
                    }
                } else if ((vj >> 1) == top.idx) {
                    top.fc += e1; top.rc += e2;
                }
            }
            break;
        }
    }
}

// The same vote when every unique k-mer maps to ONE and the same locus (all
// `val` equal and even — the common case of a read pair from a non-shared
// region).  Then `second` never gets a locus, top's sums are the prefix sums
// S_i of the permuted dups, the early stop is the first i with 2*S_i >= total,
// and the get_acm1 loop ends at the first j >= i whose prefix sums fail the
// test: three wave scans instead of a serial loop.  Lane l owns permuted
// entries [EPL*l, EPL*l + EPL).  Returns fc | rc << 16 (same on every lane).
template <int EPL, class X>
DBTK_HD uint32_t vote_single_locus(X& x, const uint16_t* perm_row, const uint32_t* dd, uint32_t nu, uint32_t cth) {
    const uint32_t lane = (uint32_t)x.lane(), b0 = EPL * lane;
    uint32_t pre[EPL], run = 0;
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const uint32_t i = b0 + j;
        if (i < nu) run += dd[perm_row[i]] & 0x00FF00FFu;  // uint8_t counts, AQ.cpp:42
        pre[j] = run;
    }
    const uint32_t excl = x.wave_excl_scan(run);
    const uint32_t tot2 = x.wave_sum(run);
    const uint32_t total = (tot2 & 0xFFFF) + (tot2 >> 16);
    uint32_t firstA = 0xFFFFFFFFu;
#pragma unroll
    for (int j = EPL - 1; j >= 0; --j) {
        const uint32_t i = b0 + j, p = excl + pre[j], S = (p & 0xFFFF) + (p >> 16);
        if (i < nu && 2 * S >= total) firstA = i;
    }
    const uint64_t mA = x.ballot(firstA != 0xFFFFFFFFu);
    const uint32_t istar = x.bcast(firstA, mA ? (int)__builtin_ctzll(mA) : 0);
    uint32_t firstB = 0xFFFFFFFFu, pB = 0;
#pragma unroll
    for (int j = EPL - 1; j >= 0; --j) {
        const uint32_t i = b0 + j, p = excl + pre[j], f = p & 0xFFFF, r = p >> 16;
        if (i < nu && i >= istar && !get_acm1(f, r, total - f - r, cth)) { firstB = i; pB = p; }
    }
    const uint64_t mB = x.ballot(firstB != 0xFFFFFFFFu);
    const uint32_t res = x.bcast(pB, mB ? (int)__builtin_ctzll(mB) : 0);
    return mB ? res : tot2;  // loop ran off the end: every k-mer was added
}

// find_matching_locus (src/aQueryFasta_thread.cpp:364-422) in its general form — any number of loci per k-mer — by the
// whole wave.  The serial loop is a stream of EVENTS (k-mer i of the vote order, locus q of its list): each event adds
// the k-mer's dups to its locus' running counts h and offers (h, locus) to updatetop2.  Stated without the loop:
//   * h of an event = sum of the adds of the events of the same locus up to it          (masked prefix sums per locus);
//   * top's sum after event e = M_e, the prefix maximum of s = h.f + h.r; e is a RECORD iff s_e > M_{e-1}; top's locus
//     and counts are those of the last record (updatetop2's first branch fires exactly on records);
//   * second's sum only ever grows: a record that changes top's locus sets it to M_{e-1} (the old top, >= second), a
//     non-record raises it to s_e if larger, a record of the same locus leaves it.  So it is the prefix maximum of
//     c_e = M_{e-1} | 0 | s_e for those three kinds — no per-locus bookkeeping;
//   * the early stop is the first k-mer whose last event has M - second >= remain (AQ.cpp:405, !get_acm2);
//   * from there on only top grows, by the dups of the following k-mers that list top's locus, until get_acm1 fails or
//     the k-mers run out — prefix sums again, as in vote_single_locus.
// Every step is a wave scan or a ballot; nothing is serial.  Arrays (LDS): ord/uval/dd/nml/poff/pool as for vote();
// ev_loc, ev_h, evd: ECAP words each of scratch; ev_loc and ev_h MAY overlay the input arrays (those are read into registers
// before the first event is written).  Returns false, having written nothing, when the events do not fit (more than ECAP)
// or a loci list is not in the LDS pool: the caller then runs vote().
// Steps (k-mers in vote order) are owned chunk-wise: step 64*q + lane sits in slot q of lane `lane`.
template <int EPL, int ECAP, class X>
DBTK_HD bool vote_parallel(X& x, const uint16_t* ord, const uint32_t* uval, const uint32_t* dd, const uint32_t* nml, const uint16_t* poff,
                           const uint32_t* pool, uint32_t* ev_loc, uint32_t* ev_h, uint32_t* evd, uint32_t nu, uint32_t cth, Asgn& top,
                           uint64_t& nvvw) {
    const uint32_t lane = (uint32_t)x.lane();
    constexpr uint32_t LAST = 1u << 31, LOCM = LAST - 1;
    if (nu == 0) return false;
    // ---- per step: dups, list length, where its events start, remain after it
    uint32_t sd[EPL], sn[EPL], sval[EPL], spo[EPL], soff[EPL], srem[EPL];
    uint32_t cd = 0, cn = 0;
    bool bad = false;
#pragma unroll
    for (int q = 0; q < EPL; ++q) {
        const uint32_t i = 64 * q + lane;
        sd[q] = 0; sn[q] = 0; sval[q] = 0; spo[q] = 0;
        if (i < nu) {
            const uint32_t u = ord[i];
            sval[q] = uval[u];
            sd[q] = dd[u] & 0x00FF00FFu;  // uint8_t counts, AQ.cpp:42
            sn[q] = (sval[q] & 1) ? nml[u] : 1u;
            if (sval[q] & 1) { spo[q] = poff[u]; bad |= spo[q] == 0xFFFFu; }
        }
        const uint32_t ds = (sd[q] & 0xFFFF) + (sd[q] >> 16);
        srem[q] = cd + x.wave_excl_scan(ds) + ds;  // inclusive prefix of the dups, for now
        soff[q] = cn + x.wave_excl_scan(sn[q]);
        cd += x.wave_sum(ds);
        cn += x.wave_sum(sn[q]);
    }
    const uint32_t total = cd, E = cn;
    if (x.ballot(bad) != 0 || E > (uint32_t)ECAP) return false;
#pragma unroll
    for (int q = 0; q < EPL; ++q) srem[q] = total - srem[q];
    x.sync();
    // ---- events: locus (| LAST on a k-mer's last event), and the k-mer's dups where h will be
#pragma unroll
    for (int q = 0; q < EPL; ++q)
        for (uint32_t j = 0; j < sn[q]; ++j) {
            const uint32_t loc = (sval[q] & 1) ? pool[spo[q] + j] : (sval[q] >> 1);
            ev_loc[soff[q] + j] = loc | (j + 1 == sn[q] ? LAST : 0u);
            ev_h[soff[q] + j] = sd[q];
        }
    x.sync();
    const uint32_t nch = (E + 63) >> 6;
    // ---- h: per distinct locus (ascending), a masked running sum over the events
    {
        bool started = false;
        uint32_t prev = 0;
        for (;;) {
            uint32_t mymin = 0xFFFFFFFFu;
            for (uint32_t e = lane; e < E; e += 64) {
                const uint32_t l = ev_loc[e] & LOCM;
                if ((!started || l > prev) && l < mymin) mymin = l;
            }
            const uint32_t L = x.wave_min(mymin);
            if (L == 0xFFFFFFFFu) break;
            uint32_t carry = 0;
            for (uint32_t c = 0; c < nch; ++c) {
                const uint32_t e = 64 * c + lane;
                const bool m = e < E && (ev_loc[e] & LOCM) == L;
                if (x.ballot(m) == 0) continue;
                const uint32_t v = m ? ev_h[e] : 0u;
                const uint32_t incl = carry + x.wave_excl_scan(v) + v;
                if (m) ev_h[e] = incl;
                carry = x.bcast(incl, 63);
            }
            prev = L;
            started = true;
        }
    }
    x.sync();
    // ---- top / second along the events: evd[e] = (M_e - second_e) | record_e << 31
    {
        uint32_t cM = 0, cS2 = 0, cTop = 0;  // carries: prefix max of s, of c, last record's locus + 1
        for (uint32_t c = 0; c < nch; ++c) {
            const uint32_t e = 64 * c + lane;
            const bool in = e < E;
            const uint32_t h = in ? ev_h[e] : 0u, s = (h & 0xFFFF) + (h >> 16), loc = in ? (ev_loc[e] & LOCM) : 0u;
            const uint32_t im = x.wave_scan_max(s);
            uint32_t em = x.shfl_up1(im);  // exclusive prefix max
            if (lane == 0) em = 0;
            if (em < cM) em = cM;
            const bool rec = in && s > em;
            const uint32_t il = x.wave_scan_lastnz(rec ? loc + 1 : 0u);
            uint32_t pl = x.shfl_up1(il);  // top's locus + 1 before this event
            if (lane == 0) pl = 0;
            if (pl == 0) pl = cTop;
            const uint32_t cc = !in ? 0u : (rec ? (pl != loc + 1 ? em : 0u) : s);
            uint32_t s2 = x.wave_scan_max(cc);
            if (s2 < cS2) s2 = cS2;
            const uint32_t M = s > em ? s : em;
            if (in) evd[e] = (M - s2) | (rec ? LAST : 0u);
            const uint32_t lm = x.bcast(M, 63), ls2 = x.bcast(s2, 63), ll = x.bcast(il, 63);
            cM = lm; cS2 = ls2;
            if (ll) cTop = ll;
        }
    }
    x.sync();
    // ---- the early stop: first k-mer with M - second >= remain at its last event
    uint32_t istar = nu - 1;
    {
        bool found = false;
#pragma unroll
        for (int q = 0; q < EPL; ++q) {
            const uint32_t i = 64 * q + lane;
            const bool t = i < nu && (evd[soff[q] + sn[q] - 1] & LOCM) >= srem[q];
            const uint64_t mk = x.ballot(t);
            if (!found && mk) { istar = 64 * q + (uint32_t)__builtin_ctzll(mk); found = true; }
        }
    }
    // top at that point: the last record at or before the k-mer's last event
    const uint32_t qs = istar >> 6, ls = istar & 63;
    uint32_t estar = 0;
#pragma unroll
    for (int q = 0; q < EPL; ++q)
        if ((uint32_t)q == qs) estar = x.bcast(soff[q] + sn[q] - 1, (int)ls);
    uint32_t er = 0;
    for (uint32_t c = 0; c <= (estar >> 6); ++c) {
        const uint32_t e = 64 * c + lane;
        const uint64_t mk = x.ballot(e <= estar && (evd[e] & LAST));
        if (mk) er = 64 * c + 63 - (uint32_t)__builtin_clzll(mk);
    }
    const uint32_t tloc = x.uni(ev_loc[er]) & LOCM;
    uint32_t th = x.uni(ev_h[er]);
    // ---- countHit's traffic so far, and the tail: k-mers after istar that list top's locus
    uint32_t vvw = 0, g[EPL], w[EPL];
#pragma unroll
    for (int q = 0; q < EPL; ++q) {
        const uint32_t i = 64 * q + lane;
        const bool multi = sval[q] & 1;
        if (i < nu && i <= istar && multi) vvw += 1 + sn[q];
        g[q] = 0; w[q] = 0;
        if (i < nu && i > istar) {
            uint32_t pos = sn[q];
            for (uint32_t j = 0; j < sn[q]; ++j)
                if ((ev_loc[soff[q] + j] & LOCM) == tloc) { pos = j; break; }
            if (pos < sn[q]) g[q] = sd[q];
            if (multi) w[q] = 1 + (pos < sn[q] ? pos + 1 : sn[q]);
        }
    }
    nvvw += x.wave_sum(vvw);
    // get_acm1 at istar with top as it stands, then after each further k-mer; stop at the first failure
    {
        const uint32_t f0 = th & 0xFFFF, r0 = th >> 16;
        uint32_t rem_star = 0;
#pragma unroll
        for (int q = 0; q < EPL; ++q)
            if ((uint32_t)q == qs) rem_star = x.bcast(srem[q], (int)ls);
        if (get_acm1(f0, r0, rem_star, cth)) {
            uint32_t cg = th, cw = 0;
            bool done = false;
#pragma unroll
            for (int q = 0; q < EPL; ++q) {
                if (done || 64u * q >= nu || 64u * q + 63 <= istar) continue;  // (uniform)
                const uint32_t i = 64 * q + lane;
                const uint32_t ig = cg + x.wave_excl_scan(g[q]) + g[q], iw = cw + x.wave_excl_scan(w[q]) + w[q];
                const bool stop = i < nu && i > istar && !get_acm1(ig & 0xFFFF, ig >> 16, srem[q], cth);
                const uint64_t mk = x.ballot(stop);
                if (mk) {
                    const int l = (int)__builtin_ctzll(mk);
                    th = x.bcast(ig, l);
                    nvvw += x.bcast(iw, l);
                    done = true;
                } else {  // through the chunk (or to the last k-mer)
                    const uint32_t lastl = (nu - 64 * q < 64) ? nu - 64 * q - 1 : 63;
                    cg = x.bcast(ig, (int)lastl); cw = x.bcast(iw, (int)lastl);
                }
            }
            if (!done) { th = cg; nvvw += cw; }  // ++j >= nu: every remaining k-mer was looked at
        }
    }
    top.idx = tloc; top.fc = th & 0xFFFF; top.rc = th >> 16;
    return true;
}

#ifndef DBTK_K2_NB
#define DBTK_K2_NB 4  // tuning knob of the probe kernel (variant builds: -DDBTK_K2_NB=n)
#endif
// ======================================================================= K2 =
// kfilter's look-ups (src/aQueryFasta_thread.cpp:204-209, 215-220) as a kernel of their own: one wavefront per
// surviving READ, rows taken at a fixed stride through a three-deep fetch pipeline (pair index -> offsets -> bytes).
// The read is packed to 2 bits/base in LDS, every lane extracts its <= 4 canonical k-mer windows
// (read2kmers_edges, AQ.h:274-311) and their home buckets, and the positions are then looked up 16 at a time: the four
// lanes of a quad read the four 16-byte parts of one bucket, so a look-up is one request for one 64-byte line.  The
// results {val, aux} are collected in LDS and leave once per read, coalesced, for the hit buffers that the resolve
// kernels (K3a / K3b) consume (BatchArgs: hitaux / hitval / hithdr / hitnk / hitoff).
// Every position is looked up; what the reference would NOT have probed (after kfilter's abort) is discounted from
// nhash1 by K3.
struct BubEvent {  // one novel read (k+1)-mer (countNovelEdges, AQ.cpp:1559-1567)
    uint32_t pair, mate, pos, locus;
    uint64_t edge;
};
struct ProbeSmem {
    uint32_t raw[72];
    uint32_t pk[20];
    uint16_t vd[20];
    uint32_t qraw[72];   // base qualities of the read (only with -b and qualities)
    uint32_t qmask[8];
    // look-up list of the plain index
    uint64_t km[NKMAX];  // canonical k-mer per list entry (NAN64: window not valid)
    uint32_t hb[NKMAX];  // its home bucket
    uint64_t rva[NKMAX]; // look-up result per POSITION: val | aux << 32 (val = NOHIT: not in the index)
};

// qString2qMask (src/aQueryFasta_thread.h:1038-1071), statement by statement, on the quality bytes
// q[0..nq): bit i of out = k-mer i passes.  Its bounds compare the BASE index with the number of
// K-MERS, so the scan stops early near the read end: reproduced as is.  One lane, only with -b -fq.
DBTK_HD void qmask_scan(const uint8_t* q, int nq, int qth, int ksize, uint32_t* out) {
    const int nk = nq - ksize + 1;
    if (nk <= 0) return;
    int qi = 0, ki = 0;
    while ((int)q[qi] - 33 < qth) { ++qi; ++ki; if (qi >= nk) return; }
    while (qi < nk) {
        bool pass = true;
        for (int qj = qi; qi < qj + ksize; ++qi) {
            if ((int)q[qi] - 33 < qth) {
                pass = false;
                ki = qi;
                while ((int)q[qi] - 33 < qth) { ++qi; ++ki; if (qi >= nk) return; }
                break;
            }
        }
        if (pass) {
            out[ki >> 5] |= 1u << (ki & 31);
            ++ki;
            if (qi >= nk) return;
            while ((int)q[qi] - 33 >= qth) {
                out[ki >> 5] |= 1u << (ki & 31);
                ++qi; ++ki;
                if (qi >= nk) return;
            }
            ki = qi;
            while ((int)q[qi] - 33 < qth) { ++qi; ++ki; if (qi >= nk) return; }
        }
    }
}

// The probe kernel in its general form: any k <= 31, reads up to 64 * NS + k - 1 bases, -bu edges, -b quality masks; every
// position is looked up in the plain index.  The lean form for the usual geometry is body_probe2 (dbtk_probe2.h).
template <int NS, class X>
DBTK_HD void body_probe(X& x, const BatchArgs& a) {
    constexpr int NSLOT = NS;  // 64-position slots per read (shadows the global upper bound)
    ProbeSmem& sm = *x.template smem<ProbeSmem>();
    const int lane = x.lane();
    const DevTables& T = a.T;
    const uint32_t k = T.ksize;
    const uint32_t ns = *a.nsurv;
    const uint32_t tend = ns - a.t0 < a.tcap ? ns : a.t0 + a.tcap;  // (ns >= t0 checked by the caller loop below)
    const uint32_t nitems = ns > a.t0 ? 2 * (tend - a.t0) : 0;
    // Rows are taken at a fixed stride and fetched through a three-deep pipeline, so that a wave is (almost) always
    // probing: while row `it` is looked up, the bytes of row it+S are in flight into registers, the offsets of row it+2S
    // are being fetched, and the pair index of row it+3S.  All of these loads are unconditional (indices clamped to
    // something valid) so that the compiler leaves them in flight.  Precondition kept by the host side: the batch
    // buffer is readable up to seq_len rounded up to a multiple of 16.
    const uint32_t S = x.nblocks();
    auto surv_of = [&](uint32_t row) { return a.surv[a.t0 + ((row < nitems ? row : 0u) >> 1)]; };
    uint32_t rw0 = 0, rw1 = 0;   // row `it`: dwords lane and 64 + lane of the read, from its 4-byte-aligned start
    uint64_t o0C = 0, o1C = 0;   //           its offsets
    uint64_t o0B = 0, o1B = 0;   // row it + S: offsets (in flight)
    uint32_t pairA = 0;          // row it + 2S: pair index (in flight)
    auto fetch_bytes = [&](uint64_t o0, uint64_t o1) {
        uint32_t len = (uint32_t)(o1 - o0);
        const uint32_t lmax = (uint32_t)MAXL < a.nkp + k - 1 ? (uint32_t)MAXL : a.nkp + k - 1;
        if (len > lmax) len = lmax;
        const uint64_t a0 = o0 & ~3ull;
        const uint32_t nw = ((uint32_t)(o0 - a0) + len + 3) >> 2;
        rw0 = *reinterpret_cast<const uint32_t*>(a.seq + ((uint32_t)lane < nw ? a0 + 4ull * lane : 0ull));
        rw1 = *reinterpret_cast<const uint32_t*>(a.seq + (64u + lane < nw ? a0 + 4ull * (64 + lane) : 0ull));
    };
    auto fetch_offsets = [&](uint32_t pair, uint32_t row) {
        const uint64_t r = 2 * (uint64_t)pair + (row & 1);
        o0B = a.off[r]; o1B = a.off[r + 1];
    };
    {
        const uint32_t it0 = x.bid();
        if (it0 < nitems) {
            fetch_offsets(x.uni(surv_of(it0)), it0);
            o0C = o0B; o1C = o1B;
            fetch_bytes(o0C, o1C);
        }
        if (it0 + S < nitems) fetch_offsets(x.uni(surv_of(it0 + S)), it0 + S);
        pairA = surv_of(it0 + 2 * S);
    }
    DBTK_STAMP_DECL
    for (uint32_t it = x.bid(); it < nitems; it += S) {  // `it` = hit-buffer row of (survivor, mate)
        DBTK_STAMP(43);  // loop / extras of the previous row
        const uint64_t o0 = o0C, o1 = o1C;
        uint32_t len = (uint32_t)(o1 - o0);
        // stay inside LDS and inside the row of the hit buffers (sized from the caller's max_read_len: a longer read breaks the contract)
        const uint32_t lmax = (uint32_t)MAXL < a.nkp + k - 1 ? (uint32_t)MAXL : a.nkp + k - 1;
        if (len > lmax) { *a.errflag = DBTK_ERR_READ_TOO_LONG; len = lmax; }
        const uint64_t a0 = o0 & ~3ull;
        const uint32_t rsh = (uint32_t)(o0 - a0), nw = (rsh + len + 3) >> 2;
        x.sync();  // previous read's LDS is dead
        if ((uint32_t)lane < nw) sm.raw[lane] = rw0;
        if (64u + lane < nw) sm.raw[64 + lane] = rw1;
        {   // advance the pipeline: bytes of row it+S, offsets of row it+2S, pair index of row it+3S
            const bool hasB = it + S < nitems, hasA = it + 2 * S < nitems;
            o0C = hasB ? o0B : 0ull; o1C = hasB ? o1B : 0ull;
            fetch_bytes(o0C, o1C);
            fetch_offsets(hasA ? x.uni(pairA) : x.uni(pairA) * 0u, hasA ? it + 2 * S : 0u);
            pairA = surv_of(it + 3 * S);
        }
        if (lane < 4) sm.raw[nw + lane] = 0;
        x.sync();
        uint32_t bad = 0;  // some base of the read is not ACGT (bytes past the read do not count)
        {   // 2-bit pack, four bases per lane: lane l packs bytes 4l .. 4l + 3 of the read into byte l of the big-endian stream
            const uint32_t B = rsh + 4u * lane, j = B >> 2, r8 = 8 * (B & 3);
            uint32_t v = 0, inmask = 0;
            if (4u * lane < len) {
                const uint32_t lo = sm.raw[j], hi = sm.raw[j + 1];
                v = r8 ? ((lo >> r8) | (hi << (32 - r8))) : lo;
                const uint32_t left = len - 4u * lane;  // bytes of this word inside the read
                inmask = left < 4 ? (1u << (8 * left)) - 1 : 0xFFFFFFFFu;
                v &= inmask;
            }
            uint32_t b = 0;
            const uint32_t r4 = pack4_b2(v, &b);
            bad = b & inmask;
            reinterpret_cast<uint8_t*>(sm.pk)[4 * (lane >> 2) + 3 - (lane & 3)] = (uint8_t)(r4 >> 16);
            if (lane < 4) sm.pk[16 + lane] = 0;
        }
        // every base ACGT (the usual read): all windows are valid and no validity bits are needed; else compute them exactly
        const bool clean = x.ballot(bad != 0) == 0;
        DBTK_STAMP(40);  // fetch pipeline, raw to LDS, pack
        if (!clean && lane < 16) {  // (rare) validity bits, 16 bases per lane
            const int c = lane;
            const uint32_t B = rsh + 16 * c, j = B >> 2, r8 = 8 * (B & 3);
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (16u * c + 4 * q < len) {
                    const uint32_t lo = sm.raw[j + q], hi = sm.raw[j + q + 1];
                    uint32_t v = r8 ? ((lo >> r8) | (hi << (32 - r8))) : lo;
                    const uint32_t left = len - (16 * c + 4 * q);
                    v &= left < 4 ? (1u << (8 * left)) - 1 : 0xFFFFFFFFu;
                    w[q] = v;
                }
            }
            uint32_t vd;
            (void)pack16(w, &vd);
            sm.vd[lane] = (uint16_t)vd;  // bytes past the read are 0 -> invalid
            if (lane < 4) sm.vd[16 + lane] = 0;
        }
        x.sync();
        const uint32_t nk = len >= k ? len - k + 1 : 0;
        // Lane l owns the npl CONSECUTIVE positions l * npl .. l * npl + npl - 1: its windows are shifts of one 32-base word, its
        // minimizers share all but npl - 1 of their m-mers, and its results are neighbours in the hit buffers.
        const uint32_t npl = (nk + 63) >> 6;
        const uint32_t p0 = (uint32_t)lane * npl;
        const bool fastw = clean && k + npl - 1 <= 32;
        uint64_t W = 0, RW = 0;  // the 32 bases from p0, and their reverse complement (base t of the window at bits 2t of RW)
        if (fastw && p0 < nk) { W = window_fw_clean(sm.pk, p0, 32); RW = revcomp2(W, 32); }
        const uint64_t kmask = (1ull << (2 * k)) - 1;
        uint64_t km[NSLOT];
        bool open[NSLOT];
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            const uint32_t i = p0 + j;
            km[j] = NAN64; open[j] = false;
            if ((uint32_t)j < npl && i < nk) {
                if (fastw) {
                    const uint64_t fw = (W >> (2 * (32 - k - j))) & kmask, rc = (RW >> (2 * j)) & kmask;
                    km[j] = fw < rc ? fw : rc;
                } else km[j] = clean ? window_kmer_clean(sm.pk, i, k) : window_kmer(sm.pk, sm.vd, i, k, nullptr, nullptr);
                open[j] = km[j] != NAN64;
            }
        }
        uint32_t* outa = a.hitaux + (size_t)it * a.nkp;
        uint32_t* outv = a.hitval + (size_t)it * a.nkp;
        if (lane == 0) { a.hitnk[it] = nk; a.hitoff[it] = o0; }
        // ---- look-ups in the plain index
        // Stage (k-mer, home bucket) per position in LDS, then look the positions up 16 at a time: the four lanes of a
        // quad read the four 16-byte parts of one bucket (keys 0,1 | keys 2,3 | values 0,1 | values 2,3), so a lookup
        // is one request for one 64-byte line and ends in its home bucket unless that bucket is full AND overflowed.
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            const uint32_t i = p0 + j;
            if ((uint32_t)j < npl && i < nk) { sm.km[i] = km[j]; sm.hb[i] = open[j] ? (uint32_t)hash_idx(km[j], T.idx_shift) : 0u; }
        }
        x.sync();
        const uint32_t nl = nk;  // entries of the look-up list in sm.km / sm.hb: one per position
        const uint32_t sub = lane & 3, qd = lane >> 2;
        // One evaluation of a quad's bucket parts against its k-mer: the two key lanes compare, the value lanes' words reach
        // them by DPP; the hit lane stores the result, lane 0 of the quad stores the miss, and `more` says that the quad must
        // look into the next bucket (full and overflowed, no match).  Returns the wave's ballot of `more`.
        auto settle = [&](bool active, uint32_t pos, uint64_t kq, uint64_t p0_, uint64_t p1_, bool& more) -> uint64_t {
            const uint64_t v0 = quad_perm64<2, 3, 2, 3>(x, p0_), v1 = quad_perm64<2, 3, 2, 3>(x, p1_);
            const bool live = active && kq != NAN64, keyl = sub < 2;
            const bool m0 = live && keyl && p0_ == kq, m1 = live && keyl && (p1_ & ~IDX_OVF) == kq;
            const bool ovf = live && sub == 1 && p1_ != NAN64 && (p1_ & IDX_OVF);
            const uint64_t hmask = x.ballot(m0 || m1), omask = x.ballot(ovf);
            const bool qhit = ((hmask >> (lane & ~3u)) & 0xF) != 0;
            more = live && !qhit && ((omask >> ((lane & ~3u) + 1)) & 1);
            // results are collected in LDS and leave for HBM once per read, coalesced (a store between two look-up rounds
            // would sit in front of the next round's loads: vector-memory operations complete in issue order)
            if (active && pos < nk) {
                if (m0 || m1) sm.rva[pos] = m0 ? v0 : v1;
                else if (sub == 0 && !qhit && !more) sm.rva[pos] = (uint64_t)NOHIT;
            }
            return x.ballot(more);
        };
        constexpr int NB = DBTK_K2_NB;  // buckets in flight per lane
        for (uint32_t i0 = 0; i0 < nl; i0 += 16 * NB) {
            uint32_t ii[NB], bq[NB];
            uint64_t kq[NB], a0[NB], a1[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                ii[u] = i0 + 16 * u + qd;
                // straight-line loads (LDS and HBM alike): a closed lane reads position 0 / bucket 0 and ignores it, so that
                // the NB bucket loads are all in flight together (a load under a branch makes the compiler wait for it)
                const uint32_t ic = ii[u] < nl ? ii[u] : 0u;
                kq[u] = sm.km[ic]; bq[u] = sm.hb[ic];
                if (ii[u] >= nk) { kq[u] = NAN64; bq[u] = 0; }
#ifdef DBTK_STAMPS
                if (a.P.diag & 64) bq[u] &= 1023;  // diagnostic: every look-up in the first 64 KB of the table (cache hits)
#endif
                bucket_part(T.idx, bq[u], sub, &a0[u], &a1[u]);  // (bq = 0 for a position without a k-mer)
            }
            uint64_t anymore = 0;
            bool more[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                more[u] = false;
                if (i0 + 16 * u < nl) anymore |= settle(true, ii[u], kq[u], a0[u], a1[u], more[u]);  // (uniform condition)
            }
            if (anymore) {  // rare (about one look-up in a thousand): walk on, bucket by bucket
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    bool mo = more[u];
                    uint32_t b = bq[u];
                    while (x.ballot(mo)) {
                        b = (b + 1) & (uint32_t)T.idx_mask;
                        uint64_t q0 = 0, q1 = 0;
                        if (mo) bucket_part(T.idx, b, sub, &q0, &q1);
                        bool m2 = false;
                        (void)settle(mo, ii[u], kq[u], q0, q1, m2);
                        mo = m2;
                    }
                }
            }
        }
        DBTK_STAMP(41);  // look-ups in the plain index
        x.sync();
        {   // the read's results: found positions, and whether they are all unique to one and the same locus
            uint64_t rv[NSLOT];
            uint32_t nh = 0, v0 = NOHIT;
            bool vdiff = false;
#pragma unroll
            for (int j = 0; j < NSLOT; ++j) {
                const uint32_t i = p0 + j;
                const bool in = (uint32_t)j < npl && i < nk;
                rv[j] = in ? sm.rva[i] : (uint64_t)NOHIT;
                const uint32_t v = (uint32_t)rv[j];
                const uint64_t hmk = x.ballot(v != NOHIT);
                nh += (uint32_t)__builtin_popcountll(hmk);
                if (v0 == NOHIT && hmk) v0 = x.bcast(v, (int)__builtin_ctzll(hmk));  // (any found value: it only counts when all are equal)
                vdiff |= v != NOHIT && v != v0;
            }
            const bool uniform = T.consistent && nh && !(v0 & 1) && x.ballot(vdiff) == 0;
            if (lane == 0) a.hithdr[it] = (uint64_t)v0 | ((uint64_t)nh << 32) | (uniform ? HDR_UNIFORM : 0ull);
#pragma unroll
            for (int j = 0; j < NSLOT; ++j) {
                const uint32_t i = p0 + j;
                if ((uint32_t)j < npl && i < nk) {
                    const uint32_t v = (uint32_t)rv[j];
                    outa[i] = v != NOHIT ? (uint32_t)(rv[j] >> 32) : AUX_MISS;
                    if (!uniform) outv[i] = v;
                }
            }
        }
        DBTK_STAMP(42);  // look-ups + result stores
        if (a.edgebuf) {  // -bu: canonical (k+1)-mers = read2kmers_edges' `edges` (AQ.h:290-295): window of k+1 bases
            uint64_t* eo = a.edgebuf + (size_t)it * a.nkp;
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {
                const uint32_t i = 64 * s + lane;
                if ((uint32_t)s < npl && i + 1 < nk) eo[i] = clean ? window_kmer_clean(sm.pk, i, k + 1) : window_kmer(sm.pk, sm.vd, i, k + 1, nullptr, nullptr);
            }
        }
        if (a.qmaskbuf) {  // -b with qualities
            x.sync();
            for (uint32_t w = lane; w < nw; w += 64) sm.qraw[w] = *reinterpret_cast<const uint32_t*>(a.qual + a0 + 4ull * w);
            if (lane < 8) sm.qmask[lane] = 0;
            x.sync();
            if (lane == 0) qmask_scan(reinterpret_cast<const uint8_t*>(sm.qraw) + rsh, (int)len, (int)a.P.qth, (int)k, sm.qmask);
            x.sync();
            if (lane < 4) a.qmaskbuf[(size_t)it * 4 + lane] = (uint64_t)sm.qmask[2 * lane] | ((uint64_t)sm.qmask[2 * lane + 1] << 32);
        }
    }
    DBTK_STAMP_FLUSH;
}

}  // namespace dbtk
#include "dbtk_probe2.h"
#include "dbtk_locus.h"
#include "dbtk_walkfast.h"
namespace dbtk {

// ---- one pair's record (kam: AQ.cpp:2169-2175; trace: every pair).  Trace records are indexed by pair, the others compacted.
template <class X>
DBTK_HD void emit_pair_record(X& x, const BatchArgs& a, int lane, uint32_t pair, uint32_t stage, uint32_t dst, uint32_t dst0, int nm1,
                              int nm2, const MateState (&ms)[2], const int (&kf)[2], const int (&hf)[2], const int (&bf)[2],
                              const int (&af)[2], const int (&rm)[2], const uint32_t (&nas)[2], const uint64_t (&Kw)[2][4],
                              const uint64_t (&Rw)[2][4]) {
    uint32_t at = pair;
    if (!a.P.trace) {
        if (lane == 0) at = x.atomic_add(a.nrec, 1u);
        at = x.bcast(at, 0);
    }
    if (at >= a.rec_cap) return;
    dbtk_pair_rec_t* r = &a.recs[at];
    if (lane == 0) {
        r->pair = pair; r->stage = stage; r->dst = dst; r->dst0 = dst0;
        r->nm1 = a.P.trace ? nm1 : 0; r->nm2 = a.P.trace ? nm2 : 0;
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        dbtk_mate_rec_t* mr = m ? &r->r2 : &r->r1;
        if (lane == 0) {
            const MateState& st = ms[m];
            mr->si = (int16_t)st.si; mr->ei = (int16_t)st.ei; mr->si_ = (int16_t)st.si_; mr->ei_ = (int16_t)st.ei_;
            mr->nt = (int16_t)st.nt; mr->bs = (int16_t)st.bs; mr->ti = (int16_t)st.ti;
            mr->kf = (uint8_t)kf[m]; mr->hf = (uint8_t)hf[m]; mr->bf = (uint8_t)bf[m]; mr->qf = 0;
            mr->af = (uint8_t)af[m]; mr->rm = (uint8_t)rm[m];
            mr->nk = (uint16_t)nas[m];
        }
        // as2 byte `lane` = states of positions 4*lane .. 4*lane+3 (0 '*', 1 '.', 2 '=')
        const int wq = lane >> 4, sh = 4 * (lane & 15);
        uint64_t kq = Kw[m][0], rq = Rw[m][0];
        if (wq == 1) { kq = Kw[m][1]; rq = Rw[m][1]; }
        if (wq == 2) { kq = Kw[m][2]; rq = Rw[m][2]; }
        if (wq == 3) { kq = Kw[m][3]; rq = Rw[m][3]; }
        const uint32_t k4 = (uint32_t)(kq >> sh) & 0xF, r4 = (uint32_t)(rq >> sh) & 0xF;
        uint8_t b = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t st = ((k4 >> q) & 1) + ((r4 >> q) & 1);  // known: 1, known and TR: 2
            if ((uint32_t)(4 * lane + q) < nas[m]) b |= (uint8_t)(st << (2 * q));
        }
        mr->as2[lane] = b;
    }
}

// ====================================================================== K3a =
// The usual pair: both mates pass kfilter, every k-mer found in the index is unique to ONE and the same locus, and
// each mate has at least cth found positions.  Then countHit needs no sorting and no vote: with a single candidate
// locus `second` stays empty and top.(fc, rc) are prefix sums of the per-k-mer dups in vote order; find_matching_locus
// keeps adding while a strand is below cth and can still reach it (get_acm1, AQ.cpp:354-357), which with totals
// D1, D2 >= cth (D = the mate's found positions = the sum of its dups) can only end with fc >= cth and rc >= cth —
// accepted by test1 (AQ.cpp:439-451) whatever the order was.  What is left is a streaming pass: 8 bytes per position
// from the probe kernel, ballots, the scalar state machine, the count atomics.  No LDS staging, few registers, so many
// waves per SIMD; pairs are taken at a fixed stride (the work per pair is uniform) with the next pair's loads in flight.
// Any other pair is passed on to the general resolve kernel untouched.  Needs: consistent RPGG (the class of a k-mer
// rides with its index value), no trace, no -b, no -bu — otherwise the host does not launch it.
constexpr uint32_t HWIN = 2048;  // counters of the pair's locus staged in LDS (loci with more TR k-mers: the rest goes straight to HBM)
struct UsualSmem {
    uint32_t gbuf[64];         // passed-on survivors not yet appended to gen_list
    uint32_t hist[HWIN / 2];   // two 16-bit increments per word (a pair adds < 2^16 to any counter)
};

// SEL: the kernel takes the pairs a.sel lists — with the fused locus-resident probe kernel (dbtk_locus.h: FUSE) the pairs that one
// did not take: it resolves its usual pairs itself and hands the others straight to the general kernel
template <int NS, bool RECS, bool SEL, class X>
DBTK_HD void body_pair_usual(X& x, const BatchArgs& a) {
    uint64_t* const ctr = counters_of(x, a);
    constexpr int NSLOT = NS;
    UsualSmem& sm = *x.template smem<UsualSmem>();
    const int lane = x.lane();
    const DevTables& T = a.T;
    const uint32_t cth = a.P.cthreshold, nloci = T.nloci;
    const bool okam = a.P.okam != 0;
    uint64_t c_qc = 0, c_thr = 0, c_feas = 0, c_asgn = 0, c_cls = 0, c_inc = 0, c_nhash1 = 0, c_kf = 0;
    const uint32_t nsurv = *a.nsurv;
    const uint64_t tlim64 = (uint64_t)a.t0 + a.tcap;
    const uint32_t tlim = nsurv < tlim64 ? nsurv : (uint32_t)tlim64;
    const uint32_t nslp = a.nkp >> 6;
    // The survivor list is in locus order (dbtk_probe2.h: body_surv_*).  A wave takes a CONTIGUOUS range of it, so that the waves
    // running at the same time are a range apart — on different loci: fifty pairs of one locus resolved side by side would send
    // all their count atomics to the same few counters at once (atomics on one address serialize).
    // (SEL: the same split over the entries of the list; q = entry, t = a.t0 + a.sel[q] its place in the survivor list)
    const uint32_t nit = SEL ? *a.nsel : (tlim > a.t0 ? tlim - a.t0 : 0u), per = (nit + x.nblocks() - 1) / x.nblocks();
    const uint64_t tb64 = (uint64_t)(SEL ? 0u : a.t0) + (uint64_t)x.bid() * per;
    const uint32_t tend = SEL ? nit : tlim;
    const uint32_t tbeg = tb64 < tend ? (uint32_t)tb64 : tend, tfin = tb64 + per < tend ? (uint32_t)(tb64 + per) : tend;
    auto place = [&](uint32_t q) -> uint32_t { return SEL ? a.t0 + a.sel[q < tfin ? q : (tbeg < tfin ? tbeg : 0u)] : q; };
    uint32_t ngb = 0;
    auto flush_gen = [&]() {
        x.sync();
        uint32_t base = 0;
        if (lane == 0) base = x.atomic_add(a.ngen, ngb);
        base = x.bcast(base, 0);
        if ((uint32_t)lane < ngb) a.gen_list[base + lane] = sm.gbuf[lane];
        x.sync();
        ngb = 0;
    };
    uint32_t nx[2][NSLOT];
    uint32_t nxnk[2] = {0, 0}, nxpair = 0;
    uint64_t nxhdr[2] = {0, 0};
    auto request = [&](uint32_t tt) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {  // straight-line loads (a slot the rows do not have re-reads the last one; deliver() masks it)
                const uint32_t sc = (uint32_t)s < nslp ? (uint32_t)s : nslp - 1;
                nx[m][s] = a.hitaux[((size_t)2 * (tt - a.t0) + m) * a.nkp + 64 * sc + lane];
            }
            nxnk[m] = a.hitnk[2 * (tt - a.t0) + m];
            nxhdr[m] = a.hithdr[2 * (tt - a.t0) + m];
        }
        nxpair = a.surv[tt];
    };
    // The loop is arranged around one hardware fact: vector-memory operations complete in issue order, so waiting for
    // a load also waits for every atomic issued before it, and a counter atomic is a slow read-modify-write at the
    // memory side.  Hence: the next pair's loads are issued at the top of an iteration and taken delivery of (into
    // registers) just BEFORE this pair's atomics and record stores go out; those then have a whole iteration to finish.
    uint32_t pair = 0, nkm[2] = {0, 0}, nsl = 0;
    uint32_t ha[2][NSLOT];     // class of the position's k-mer at the pair's locus, AUX_MISS: not in the index
    uint32_t hval[2] = {NOHIT, NOHIT}, hfound[2] = {0, 0};
    bool huni[2] = {false, false};
    auto deliver = [&]() {
        pair = x.uni(nxpair) + a.pair_base;
        nkm[0] = x.uni(nxnk[0]); nkm[1] = x.uni(nxnk[1]);
        nsl = ((nkm[0] > nkm[1] ? nkm[0] : nkm[1]) + 63) >> 6;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const uint32_t hlo = x.uni((uint32_t)nxhdr[m]), hhi = x.uni((uint32_t)(nxhdr[m] >> 32));
            hval[m] = hlo; hfound[m] = hhi & 0x7FFFFFFFu; huni[m] = (hhi >> 31) != 0;
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {
                const uint32_t i = 64 * s + lane;
                const bool in = (uint32_t)s < nsl && i < nkm[m];
                ha[m][s] = in ? nx[m][s] : AUX_MISS;
            }
        }
    };
    DBTK_STAMP_DECL
    uint32_t q = tbeg;
    uint32_t t = tbeg < tfin ? x.uni(place(tbeg)) : 0u, tn = tbeg < tfin ? place(tbeg + 1) : 0u;  // this entry's place and the next one's (SEL: loaded an iteration ahead)
    if (q < tfin) { request(t); deliver(); }
    for (; q < tfin; ++q, t = x.uni(tn), tn = place(q + 1)) {
        DBTK_STAMP(39);  // loop overhead / record of the previous pair
#ifdef DBTK_STAMPS
        if (!(a.P.diag & 8))  // diagnostic: no loads after the first pair (every pair re-resolves the same data)
#endif
        request(q + 1 < tfin ? x.uni(tn) : t);  // (past the end: a harmless reload, so that the loads stay straight-line)
        const uint32_t pair_cur = pair;
        // kfilter (AQ.cpp:190-228) aborts a mate at its (nk - cth + 1)-th miss, i.e. iff it has fewer than cth found positions
        // (the probe kernel counted the found positions of each read and checked that they share one even index value)
        const uint32_t nhit[2] = {hfound[0], hfound[1]}, v0 = hval[0];
        const bool usual = nkm[0] >= cth && nkm[1] >= cth && nhit[0] >= cth && nhit[1] >= cth && nhit[0] && nhit[1] && huni[0] && huni[1] &&
                           hval[0] == hval[1];
        // The pair kfilter removes altogether (AQ.cpp:190-228: a mate with fewer than cth k-mers, or whose misses pass nk - cth, is
        // cleared; both cleared: nothing is left to vote on) — a background pair that got through subfilter on a shared repeat.  It
        // is decided from the same data: the look-ups kfilter made before it gave up are the positions up to the (nk - cth + 1)-th
        // miss.  Nothing else happens to such a pair, so it ends here instead of in the general kernel.
        const bool gone0 = nkm[0] < cth || nhit[0] < cth, gone1 = nkm[1] < cth || nhit[1] < cth;
        if (gone0 && gone1) {
            const uint32_t nslk = ((nkm[0] > nkm[1] ? nkm[0] : nkm[1]) + 63) >> 6;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                if (nkm[m] < cth) { c_kf += 1; continue; }  // kf = 1 without a look-up (AQ.cpp:196-199)
                const uint32_t nk = nkm[m], maxns = nk - cth;
                uint32_t cum_miss = 0, abort_at = nk;
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) {
                    if ((uint32_t)s >= nslk) continue;
                    const uint32_t i = 64 * s + lane;
                    const uint64_t missmask = x.ballot(i < nk && ha[m][s] == AUX_MISS);
                    const uint32_t nm = (uint32_t)__builtin_popcountll(missmask);
                    if (abort_at == nk && cum_miss + nm > maxns) {
                        uint64_t mm = missmask;  // the (maxns + 1 - cum_miss)-th set bit
                        for (uint32_t r = maxns - cum_miss; r > 0; --r) mm &= mm - 1;
                        abort_at = 64 * s + (uint32_t)__builtin_ctzll(mm);
                    }
                    cum_miss += nm;
                }
                c_nhash1 += abort_at + 1;  // (found < cth: the abort point exists)
                c_kf += 1;
            }
            deliver();
            continue;
        }
        DBTK_STAMP(32);  // request + usual test
        uint32_t dst0 = NAN32, dst = nloci, stage = DBTK_STAGE_LOCUS;
        MateState ms[2];
        for (int m = 0; m < 2; ++m) ms[m] = MateState{-1, -1, 0, 0, -1, -1, -1, 0, 0};
        uint32_t nas[2] = {0, 0};
        uint64_t Kw[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, Rw[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        int kf[2] = {0, 0}, hf[2] = {0, 0}, bf[2] = {0, 0}, af[2] = {0, 0}, rm[2] = {0, 0};
        uint32_t base = 0, win = 0;
        if (usual) {
            c_nhash1 += nkm[0] + nkm[1];  // kfilter ran over both mates in full
            dst0 = v0 >> 1;
            dst = dst0;
            const uint32_t trb0 = T.trbeg[dst0], trb1 = T.trbeg[dst0 + 1];  // the locus' counter range
            if (a.P.qc && T.qc && !T.qc[dst]) {  // AQ.cpp:2059-2062
                c_qc += 2;
                stage = DBTK_STAGE_QC;
            } else if (a.P.threading) {  // AQ.cpp:2070-2090: `alned` stays false at HEAD, nothing else happens;
                c_thr += 2;              // v1.3 (threading = 2): the walk kernel takes the pair from here
                if (a.P.threading == DBTK_THREADING_V13) { stage = DBTK_STAGE_THREADING; if (lane == 0) a.walk_dst[t] = dst; }
            } else if (a.P.extract) {  // AQ.cpp:2094-2099
                c_thr += 2; c_feas += 2;
                stage = DBTK_STAGE_EXTRACT;
            } else {
                c_thr += 2; c_feas += 2;
                // assignTRkmc (AQ.cpp:2138-2144): the class of a found k-mer at the one locus rides with its index value
                uint32_t ntr[2] = {0, 0};
                uint64_t Tw[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    nas[m] = nkm[m];
                    c_cls += nkm[m];
                    uint32_t mytr = 0, carry = 0;
#pragma unroll
                    for (int s = 0; s < NSLOT; ++s) {
                        if ((uint32_t)s >= nsl) continue;
                        const uint32_t c = ha[m][s] != AUX_MISS ? ha[m][s] : CLS_NONE;
                        const uint64_t kb = x.ballot(c != CLS_NONE), rb = x.ballot(c != CLS_NONE && c != CLS_FLANK);
                        Kw[m][s] = kb; Rw[m][s] = rb;
                        mytr += (uint32_t)__builtin_popcountll(rb);
                        Tw[m][s] = transitions_word(kb, rb, carry);
                    }
                    ntr[m] = mytr & 0xFF;  // uint8_t ntr, AQ.cpp:1454
                }
                DBTK_STAMP(33);  // states
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    Bits256 K, R, Tm;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { K.w[q] = Kw[m][q]; R.w[q] = Rw[m][q]; Tm.w[q] = Tw[m][q]; }
                    assign_masks(K, R, Tm, (int)nas[m], ntr[m], a.P, ms[m]);
                    af[m] = ms[m].af; rm[m] = ms[m].rm;
                }
                DBTK_STAMP(34);  // assign
                // accumulate (AQ.cpp:2145-2158)
                if (rm[0] && rm[1]) { dst = nloci; stage = DBTK_STAGE_ASGN; }
                else {
                    stage = DBTK_STAGE_COUNTED;
                    c_asgn += (uint64_t)(2 - rm[0] - rm[1]);
                    // The increments of one pair all fall in its locus' contiguous counter range, but scattered over it
                    // (slot order is a hash order) and often repeated (tandem repeats).  A random 8-byte atomic costs a
                    // 64-byte read-modify-write at the memory side (~24 G/s chip-wide, tools/atomics.hip), so the pair's
                    // increments are first summed in an LDS window over the range and then flushed with lanes on
                    // consecutive counters: one transaction per touched 64-byte line instead of one per k-mer.
                    base = x.uni(trb0);
                    win = x.uni(trb1) - base < HWIN ? x.uni(trb1) - base : HWIN;
                    for (uint32_t w = lane; w < (win + 1) / 2; w += 64) sm.hist[w] = 0;
                    x.sync();
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        if (rm[m]) continue;
#pragma unroll
                        for (int s = 0; s < NSLOT; ++s) {
                            if ((uint32_t)s >= nsl) continue;
                            const uint32_t c = ha[m][s] != AUX_MISS ? ha[m][s] : CLS_NONE;
                            if (c != CLS_NONE && c != CLS_FLANK) {
                                const uint32_t o = c - base;
                                if (o < win) x.lds_add(&sm.hist[o >> 1], 1u << (16 * (o & 1)));
                                else x.atomic_add(&a.counts[c], 1ull);
                            }
                            c_inc += (uint64_t)__builtin_popcountll(Rw[m][s]);
                        }
                    }
                    x.sync();
                }
            }
        }
        DBTK_STAMP(35);  // LDS histogram
        deliver();  // the next pair's probe results, before anything of this pair goes out to memory
        DBTK_STAMP(36);  // delivery of the next pair
        if (!usual) {  // the general kernel redoes this pair from its probe results
            if (lane == 0) sm.gbuf[ngb] = t;
            if (++ngb == 64) flush_gen();
            continue;
        }
        if (stage == DBTK_STAGE_COUNTED) {
#ifdef DBTK_STAMPS
            if (!(a.P.diag & 16))  // diagnostic: no per-locus atomics
#endif
            if (lane == 0) {
                x.atomic_add(&a.nmapread[dst], (uint64_t)(2 - rm[0] - rm[1]));
                x.atomic_add(&a.kmc[dst], (uint64_t)(int64_t)((ms[0].ei - ms[0].si) + (ms[1].ei - ms[1].si)));
            }
#ifdef DBTK_STAMPS
            if (a.P.diag & 4) win = 0;  // diagnostic: no count atomics
#endif
            for (uint32_t i = lane; i < win; i += 64) {
                const uint32_t v = (sm.hist[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
                if (v) x.atomic_add(&a.counts[base + i], (uint64_t)v);
            }
            x.sync();
        }
        DBTK_STAMP(37);  // count atomics
        const bool want = RECS && a.recs && ((okam && stage == DBTK_STAGE_COUNTED) || (okam && a.P.simmode && stage == DBTK_STAGE_ASGN) ||
                                             (a.P.extract && stage == DBTK_STAGE_EXTRACT));
        if (want) emit_pair_record(x, a, lane, pair_cur, stage, dst, dst0, 0, 0, ms, kf, hf, bf, af, rm, nas, Kw, Rw);
    }
    if (ngb) flush_gen();
    DBTK_STAMP_FLUSH;
    if (lane == 0) {
        if (c_qc) x.atomic_add(&ctr[DBTK_C_QCFILTERED], c_qc);
        if (c_kf) x.atomic_add(&ctr[DBTK_C_KMERFILTERED], c_kf);
        if (c_thr) x.atomic_add(&ctr[DBTK_C_THREADING], c_thr);
        if (c_feas) x.atomic_add(&ctr[DBTK_C_FEASIBLE], c_feas);
        if (c_asgn) x.atomic_add(&ctr[DBTK_C_ASGN], c_asgn);
        if (c_cls) x.atomic_add(&ctr[DBTK_C_ALGO_CLS], c_cls);
        if (c_inc) x.atomic_add(&ctr[DBTK_C_ALGO_INC], c_inc);
        if (c_nhash1) {
            x.atomic_add(&ctr[DBTK_C_NHASH1], c_nhash1);
            x.atomic_add(&ctr[DBTK_C_ALGO_PROBES], c_nhash1);
        }
    }
}

// ====================================================================== K3b =
template <int NS, bool RECS, class X>
DBTK_HD void body_pair(X& x, const BatchArgs& a) {
    uint64_t* const ctr = counters_of(x, a);
    typedef PairSmemT<NS> Smem;
    constexpr int NSLOT = NS, NHMAX = Smem::NH, LCAP = Smem::LC;  // shadow the global upper bounds
    constexpr int EPL = 2 * NS;  // hit-list entries per lane
    Smem& sm = *x.template smem<Smem>();
    const int lane = x.lane();
    const DevTables& T = a.T;
    const uint32_t k = T.ksize, cth = a.P.cthreshold, nloci = T.nloci;
    const bool okam = a.P.okam != 0;
    // per-block counters, flushed once at the end
    uint64_t c_kf = 0, c_hf = 0, c_qc = 0, c_thr = 0, c_feas = 0, c_asgn = 0, c_nhash1 = 0, c_vv = 0, c_cls = 0, c_inc = 0, c_bait = 0;
    uint64_t c_vote = 0;  // vv words the vote read (a path statistic, DBTK_PS_VOTE_VV: which pairs are voted on at all depends on the path)
    DBTK_STAMP_DECL
    const uint32_t nsurv = *a.nsurv;
    const uint32_t nslp = a.nkp >> 6;  // slots the hit buffers reserve per read
    if (a.hint_out && x.bid() == 0 && lane == 0) {  // (a hint for the host's next launches: never a matter of results)
        volatile uint32_t* h = a.hint_out;
        h[0] = nsurv; h[6] = a.sortflag ? *a.sortflag : 0u;
        h[1] = a.nsel ? *a.nsel : nsurv;  // pairs the lean probe kernel took
    }

    // Work items: every survivor of the chunk, or (after the usual-pair kernel) the ones it passed on.
    // Software pipeline, three deep, so that nothing is waited for in the iteration that requested it: the ticket of
    // item i+2 (atomic), the survivor index of item i+1 (list lookup), the probe results of item i+1 (hit buffers).
    const bool listmode = a.gen_list != nullptr;
    const uint32_t nitems = listmode ? *a.ngen : (nsurv > a.t0 ? ((nsurv - a.t0 < a.tcap) ? nsurv - a.t0 : a.tcap) : 0u);
    constexpr uint32_t NOITEM = 0xFFFFFFFFu;
    if (nitems == 0) return;  // (an empty chunk must not touch the ticket: thousands of atomics on one address serialize)
    // A contiguous range of the items per wave (the list is in locus order, or — after the usual-pair kernel — in runs of it:
    // waves side by side then work on different loci, see body_pair_usual).  (A shared ticket counter balanced the varying work
    // per item better, but 20 000 atomics on one address serialize at ~18 ns each: they, not the work, were the kernel's 0.2 ms.)
    const uint32_t qper = (nitems + x.nblocks() - 1) / x.nblocks();
    const uint64_t qb64 = (uint64_t)x.bid() * qper;
    const uint32_t qend = qb64 + qper < nitems ? (uint32_t)(qb64 + qper) : nitems;
    uint32_t myq = qb64 < nitems ? (uint32_t)qb64 : nitems;
    auto take = [&]() { myq += 1; return myq; };
    auto lookup = [&](uint32_t q) -> uint32_t {  // (an unconditional load: one issued under a branch is waited for on the spot)
        const uint32_t qq = q < qend ? q : 0u;
        const uint32_t v = listmode ? a.gen_list[qq] : a.t0 + qq;
        return q < qend ? v : NOITEM;
    };
    // Pipeline, so that nothing is waited for in the iteration that requested it (all loads straight-line, wave-uniform
    // values kept in vector registers until they are used — see `vzero`):
    //   item i+3: its survivor index (list lookup)            -> tD
    //   item i+2: its reads' offsets and position counts      -> ofC, nkC
    //   item i+1: its probe results, its pair index, its bytes -> nx*, taken delivery of at the end of item i
    const uint32_t lzz = (uint32_t)lane * a.vzero;  // 0
    uint32_t nxv[2][NSLOT], nxa[2][NSLOT];
    uint64_t nxhdr[2] = {0, 0};
    uint32_t nxpair = 0, nxrw[2][2] = {{0, 0}, {0, 0}};
    uint64_t nxo0[2] = {0, 0};   // where the bytes in flight start, and how many positions they make
    uint32_t nxnk[2] = {0, 0};
    uint64_t ofC[2] = {0, 0}, hdC[2] = {0, 0};    // item i+2 (in flight)
    uint32_t nkC[2] = {0, 0};
    // "no item" (past the end of the wave's range) re-reads the rows of a pair that HAS rows, so that the loads stay straight-line: with a
    // list, its first entry — since the fused probe kernel (dbtk_locus.h) a pair that is not listed may have written none, and its
    // read offset would be whatever the buffer held
    const uint32_t tsafe = listmode ? x.uni(a.gen_list[0]) : a.t0;
    auto rowof = [&](uint32_t tt, int m) { return (size_t)2 * ((tt != NOITEM ? tt : tsafe) - a.t0) + m; };
    auto fetch_meta = [&](uint32_t tt) {
#pragma unroll
        for (int m = 0; m < 2; ++m) { ofC[m] = a.hitoff[rowof(tt, m) + lzz]; nkC[m] = a.hitnk[rowof(tt, m) + lzz]; hdC[m] = a.hithdr[rowof(tt, m) + lzz]; }
    };
    // probe results + bytes of item tt, whose offsets / position counts are o0[], nk[]
    auto request = [&](uint32_t tt, const uint64_t o0[2], const uint32_t nk[2], const uint64_t hd[2]) {
        const uint32_t tc = tt != NOITEM ? tt : tsafe;  // (what comes back for "no item" is never used: the loop ends first)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {  // straight-line loads: no item -> row 0, a slot the rows do not have -> the last one
                const uint32_t sc = (uint32_t)s < nslp ? (uint32_t)s : nslp - 1;
                nxa[m][s] = a.hitaux[rowof(tc, m) * a.nkp + 64 * sc + lane];
                nxv[m][s] = a.hitval[rowof(tc, m) * a.nkp + 64 * sc + lane];  // (stale for a read whose header says "uniform")
            }
            nxhdr[m] = hd[m];
            const uint32_t len = nk[m] ? nk[m] + k - 1 : 0;  // (<= MAXL: the probe kernel clamps)
            const uint64_t a0 = o0[m] & ~3ull;
            const uint32_t nw = ((uint32_t)(o0[m] - a0) + len + 3) >> 2;
            nxrw[m][0] = *reinterpret_cast<const uint32_t*>(a.seq + ((uint32_t)lane < nw ? a0 + 4ull * lane : 0ull));
            nxrw[m][1] = *reinterpret_cast<const uint32_t*>(a.seq + (64u + lane < nw ? a0 + 4ull * (64 + lane) : 0ull));
            nxo0[m] = o0[m]; nxnk[m] = nk[m];
        }
        nxpair = a.surv[tc];
    };
    // The current item's probe results live in km / hv / ha (+ sm.hval) from the moment they are taken delivery of: at the
    // end of the previous item, just BEFORE that item's count atomics went out — a wait for a load also waits for every
    // older store or atomic (vector memory completes in issue order), and a counter atomic is a slow memory-side RMW.
    uint64_t km[2][NSLOT];
    uint32_t hv[2][NSLOT], ha[2][NSLOT];
    uint32_t nkm[2] = {0, 0}, pair_cur = 0;
    auto deliver = [&]() {
        pair_cur = x.uni(nxpair) + a.pair_base;
        uint32_t rsh[2], len[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            nkm[m] = nxnk[m];  // positions of the read (the probe kernel clamps reads to MAXL)
            len[m] = nkm[m] ? nkm[m] + k - 1 : 0;
            rsh[m] = (uint32_t)(nxo0[m] & 3);
        }
        const uint32_t nslN = ((nkm[0] > nkm[1] ? nkm[0] : nkm[1]) + 63) >> 6;
        // the k-mers: raw bytes -> LDS, 16 lanes per mate pack 16 bases each (bytes past the read are 0 = invalid), windows
        x.sync();
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const uint32_t nw = (rsh[m] + len[m] + 3) >> 2;
            if ((uint32_t)lane < nw) sm.raw[m][lane] = nxrw[m][0];
            if (64u + lane < nw) sm.raw[m][64 + lane] = nxrw[m][1];
            if (lane < 4) sm.raw[m][nw + lane] = 0;
        }
        x.sync();
        if (lane < 32) {
            const int m = (int)(lane >> 4);
            const uint32_t c = lane & 15, B = rsh[m] + 16 * c, j = B >> 2, r8 = 8 * (B & 3);
            uint32_t w[4] = {0, 0, 0, 0}, vdb = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (16u * c + 4 * q < len[m]) {
                    const uint32_t lo = sm.raw[m][j + q], hi = sm.raw[m][j + q + 1];
                    const uint32_t v = r8 ? ((lo >> r8) | (hi << (32 - r8))) : lo;
                    const uint32_t left = len[m] - (16 * c + 4 * q);  // bytes of this word inside the read
                    w[q] = v & (left < 4 ? (1u << (8 * left)) - 1 : 0xFFFFFFFFu);
                }
            sm.pk[m][c] = pack16(w, &vdb);
            sm.vd[m][c] = (uint16_t)vdb;
            if (c < 4) { sm.pk[m][16 + c] = 0; sm.vd[m][16 + c] = 0; }
        }
        x.sync();
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {
                const uint32_t i = 64 * s + lane;
                km[m][s] = NAN64; hv[m][s] = NOHIT; ha[m][s] = 0;
                if ((uint32_t)s < nslN && i < nkm[m]) {
                    km[m][s] = window_kmer(sm.pk[m], sm.vd[m], i, k, nullptr, nullptr);
                    const bool found = nxa[m][s] != AUX_MISS;
                    // a read whose found k-mers all carry one index value has it in its header, not per position
                    hv[m][s] = (nxhdr[m] & HDR_UNIFORM) ? (found ? (uint32_t)nxhdr[m] : NOHIT) : nxv[m][s];
                    ha[m][s] = found ? nxa[m][s] : 0u;
                    sm.hval[m][i] = hv[m][s];  // (read by the dedup only: dead by the time the next item is delivered)
                }
            }
    };
    auto uni64 = [&](uint64_t v) { return ((uint64_t)x.uni((uint32_t)(v >> 32)) << 32) | x.uni((uint32_t)v); };
    uint32_t t = x.uni(lookup(myq));
    {   // prologue: item 0 through all its stages
        fetch_meta(t);
        const uint64_t o0[2] = {uni64(ofC[0]), uni64(ofC[1])}, hd[2] = {uni64(hdC[0]), uni64(hdC[1])};
        const uint32_t nk[2] = {x.uni(nkC[0]), x.uni(nkC[1])};
        request(t, o0, nk, hd);
        deliver();
    }
    uint32_t tB = x.uni(lookup(take()));        // item 1 (value)
    fetch_meta(tB);                             //         its offsets (in flight)
    uint32_t tD = lookup(take() + lzz);         // item 2 (in flight)

#ifdef DBTK_STAMPS
    uint64_t pair_t0_ = x.clock();
    uint32_t diag_need = 0, diag_nu = 0;
#endif
    for (;;) {
        if (t == NOITEM) break;
        const uint32_t pair = pair_cur;
        bool delivered = false;
        x.sync();  // previous pair's LDS is dead from here on
        DBTK_STAMP(0);  // ticket

        // ---- P3: the probe kernel's results for both reads are in km / hv / ha
        int kf[2], rm[2], hf[2] = {0, 0}, af[2] = {0, 0}, bf[2] = {0, 0};
        uint32_t nhit[2] = {0, 0};  // positions of the mate found in the index
        kf[0] = nkm[0] < cth; kf[1] = nkm[1] < cth;
        rm[0] = kf[0]; rm[1] = kf[1];
        const bool both_short = rm[0] && rm[1];
        const uint32_t nsl = ((nkm[0] > nkm[1] ? nkm[0] : nkm[1]) + 63) >> 6;  // slots in use (3 for 150 bp reads)
        // the pipeline moves on: data of the next item, offsets of the one after, survivor index of the third
        const uint32_t tnext = tB;
        {
            const uint64_t o0[2] = {uni64(ofC[0]), uni64(ofC[1])}, hd[2] = {uni64(hdC[0]), uni64(hdC[1])};
            const uint32_t nk[2] = {x.uni(nkC[0]), x.uni(nkC[1])};
            request(tnext, o0, nk, hd);
        }
        tB = x.uni(tD);
        fetch_meta(tB);
        tD = lookup(take() + lzz);
        DBTK_STAMP(3);  // hit-buffer loads
        if (!both_short) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                if (rm[m]) continue;
                const uint32_t nk = nkm[m], maxns = nk - cth;
                uint32_t cum_miss = 0, abort_at = nk;  // abort_at: position of the (maxns+1)-th miss
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) {
                    if ((uint32_t)s >= nsl) continue;
                    const uint32_t i = 64 * s + lane;
                    const uint64_t missmask = x.ballot(i < nk && hv[m][s] == NOHIT);
                    const uint32_t nm = (uint32_t)__builtin_popcountll(missmask);
                    if (abort_at == nk && cum_miss + nm > maxns) {
                        uint64_t mm = missmask;  // the (maxns + 1 - cum_miss)-th set bit
                        for (uint32_t r = maxns - cum_miss; r > 0; --r) mm &= mm - 1;
                        abort_at = 64 * s + (uint32_t)__builtin_ctzll(mm);
                    }
                    cum_miss += nm;
                }
                nhit[m] = nk - cum_miss;
                if (abort_at != nk) { kf[m] = 1; rm[m] = 1; c_nhash1 += abort_at + 1; }  // its.clear(); kf = 1
                else c_nhash1 += nk;
            }
        }
        c_kf += (uint64_t)(kf[0] + kf[1]);
        DBTK_STAMP(4);  // kfilter verdicts

        uint32_t stage = DBTK_STAGE_KFILTER, dst = nloci, dst0 = NAN32;
        int nm1 = 0, nm2 = 0;
        MateState ms[2];
        for (int m = 0; m < 2; ++m) ms[m] = MateState{-1, -1, 0, 0, -1, -1, -1, 0, 0};
        uint32_t nas[2] = {0, 0};
        uint64_t Kw[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, Rw[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};

        if (!(rm[0] && rm[1])) {
            // ---- The usual pair: both mates kept, every hit k-mer unique to ONE and the same locus, and at least
            // cth hit positions in each mate.  Then countHit's outcome needs no sorting at all: with a single
            // candidate locus `second` stays empty and top.(fc, rc) are prefix sums of the per-k-mer dups in vote
            // order; find_matching_locus keeps adding while a strand is below cth and can still reach it
            // (get_acm1, AQ.cpp:354-357), which with totals D1, D2 >= cth (D = the mate's hit positions, the sum of
            // its dups) can only end with fc >= cth and rc >= cth — accepted by test1 (AQ.cpp:439-451) whatever
            // the order was.  The partial sums themselves (nm1/nm2) are only reported in trace mode, which takes
            // the general path below.
            bool usual = false;
            uint32_t v0 = NOHIT;
            if (!a.P.trace && !rm[0] && !rm[1] && nhit[0] >= cth && nhit[1] >= cth && nhit[0] && nhit[1]) {
                bool vdiff = false;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int s = 0; s < NSLOT; ++s) {
                        if ((uint32_t)s >= nsl) continue;
                        const uint32_t i = 64 * s + lane;
                        const bool hit = i < nkm[m] && hv[m][s] != NOHIT;
                        if (v0 == NOHIT) {
                            const uint64_t hmk = x.ballot(hit);
                            if (hmk) v0 = x.bcast(hv[m][s], (int)__builtin_ctzll(hmk));
                        }
                        vdiff |= hit && hv[m][s] != v0;
                    }
                usual = !(v0 & 1) && x.ballot(vdiff) == 0;
            }
            if (usual) {
                dst0 = v0 >> 1;
                nm1 = (int)nhit[0]; nm2 = (int)nhit[1];  // >= cth each: accepted below
            } else {
            // ---- P4: gather the hit lists (its1 ++ its2 with the orient bit, AQ.cpp:263-266)
            uint32_t n = 0;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                if (rm[m]) continue;
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) {
                    if ((uint32_t)s >= nsl) continue;
                    const uint32_t i = 64 * s + lane;
                    const bool hit = i < nkm[m] && hv[m][s] != NOHIT;
                    const uint64_t hm = x.ballot(hit);
                    if (hit) {
                        const uint32_t at = n + (uint32_t)__builtin_popcountll(hm & ((1ull << lane) - 1));
                        sm.u.s.skey[at] = km[m][s];
                        sm.u.s.sinfo[at] = (uint16_t)((m << 8) | i);
                    }
                    n += (uint32_t)__builtin_popcountll(hm);
                }
            }
            x.sync();
            DBTK_STAMP(5);  // gather
            // ---- P5: sort the hit list by key, then run-length encode into unique k-mers + PE_KMC dups
            // (AQ.cpp:268-295).  Order-preserving bucket sort in LDS: bucket = the k-mer's top 8 bits
            // (monotone in the key), slots claimed with LDS atomics, exact rank inside the (tiny) bucket by
            // comparing (key, info) with its other members.  O(n) instead of a comparison network.
            {
                const uint32_t bsh = 2 * k > 8 ? 2 * k - 8 : 0;
                const int nown = (int)((n + 63 - lane) / 64);  // entries lane, lane+64, ... owned by this lane
                uint64_t myk[EPL]; uint32_t myi[EPL], mys[EPL];
                for (uint32_t i = lane; i < (uint32_t)NBKT; i += 64) sm.w.b.cnt[i] = 0;
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    myk[j] = 0; myi[j] = 0; mys[j] = 0;
                    if (j < nown) { myk[j] = sm.u.s.skey[lane + 64 * j]; myi[j] = sm.u.s.sinfo[lane + 64 * j]; }
                }
                x.sync();
#pragma unroll
                for (int j = 0; j < EPL; ++j)
                    if (j < nown) mys[j] = x.lds_add(&sm.w.b.cnt[(uint32_t)(myk[j] >> bsh) & (NBKT - 1)], 1u);  // slot in bucket
                x.sync();
                {   // bucket starts: lane owns buckets [4 * lane, 4 * lane + 4)
                    uint32_t c4[4], sum = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { c4[q] = sm.w.b.cnt[4 * lane + q]; sum += c4[q]; }
                    uint32_t at = x.wave_excl_scan(sum);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { sm.w.b.bst[4 * lane + q] = (uint16_t)at; at += c4[q]; }
                }
                x.sync();
#pragma unroll
                for (int j = 0; j < EPL; ++j)
                    if (j < nown) {
                        const uint32_t b = (uint32_t)(myk[j] >> bsh) & (NBKT - 1);
                        const uint32_t at = sm.w.b.bst[b] + mys[j];
                        sm.w.b.tmpk[at] = myk[j];
                        sm.w.b.tmpi[at] = (uint16_t)myi[j];
                    }
                x.sync();
#pragma unroll
                for (int j = 0; j < EPL; ++j)
                    if (j < nown) {
                        const uint32_t b = (uint32_t)(myk[j] >> bsh) & (NBKT - 1);
                        const uint32_t b0 = sm.w.b.bst[b], bn = sm.w.b.cnt[b];
                        uint32_t rk = 0;
                        for (uint32_t q = 0; q < bn; ++q) {
                            const uint64_t ok = sm.w.b.tmpk[b0 + q];
                            const uint32_t oi = sm.w.b.tmpi[b0 + q];
                            rk += (ok < myk[j]) || (ok == myk[j] && oi < myi[j]);
                        }
                        sm.u.s.skey[b0 + rk] = myk[j];
                        sm.u.s.sinfo[b0 + rk] = (uint16_t)myi[j];
                    }
                x.sync();
                for (uint32_t i = lane; i < (uint32_t)NHMAX; i += 64) sm.w.a.dd[i] = 0;
                x.sync();
            }
            DBTK_STAMP(6);  // sort
            uint32_t nu;
            {
                // contiguous ownership: lane owns sorted entries [EPL*lane, EPL*lane + EPL)
                const uint32_t b0 = EPL * (uint32_t)lane;
                uint32_t heads = 0;
                const uint64_t prevk = (b0 > 0 && b0 <= n) ? sm.u.s.skey[b0 - 1] : 0;
                uint64_t ks[EPL];
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    const uint32_t r = b0 + j;
                    ks[j] = r < n ? sm.u.s.skey[r] : 0;
                    if (r < n && (r == 0 || ks[j] != (j ? ks[j - 1] : prevk))) ++heads;
                }
                uint32_t uidx = x.wave_excl_scan(heads);  // unique index of my first head
                nu = x.wave_sum(heads);
#pragma unroll
                for (int j = 0; j < EPL; ++j) {  // a non-head entry belongs to the most recent head at or before it
                    const uint32_t r = b0 + j;
                    if (r < n) {
                        const bool head = (r == 0 || ks[j] != (j ? ks[j - 1] : prevk));
                        if (head) ++uidx;
                        const uint32_t u = uidx - 1;
                        const uint16_t info = sm.u.s.sinfo[r];
                        if (head) sm.w.a.uval[u] = sm.hval[info >> 8][info & 0xFF];
                        x.lds_add(&sm.w.a.dd[u], (info >> 8) ? 0x10000u : 1u);
                    }
                }
            }
            x.sync();
            DBTK_STAMP(7);  // dedup
            // ---- P6: loci per unique k-mer (AQ.cpp:311-317); skey/sinfo are dead now
            bool alleq, single;
            {
                const uint32_t v0 = nu ? x.uni(sm.w.a.uval[0]) : 0;
                uint32_t odd = 0;
                bool vdiff = false;
                for (uint32_t u = lane; u < nu; u += 64) {
                    const uint32_t v = sm.w.a.uval[u];
                    odd += v & 1;
                    vdiff |= v != v0;
                }
                const uint32_t nodd = x.wave_sum(odd);
                c_vv += nodd;
                single = nu && nodd == 0 && x.ballot(vdiff) == 0;  // one locus, every k-mer unique to it
                alleq = single;
                if (!single) {
                    for (uint32_t u = lane; u < nu; u += 64) {
                        const uint32_t v = sm.w.a.uval[u];
                        sm.u.v.nml[u] = (v & 1) ? T.vv[v >> 1] : 1u;
                    }
                    for (uint32_t i = lane; i < (uint32_t)LCAP; i += 64) sm.u.v.lkey[i] = NAN32;
                    x.sync();
                    {   // loci lists of the multi-locus k-mers into the LDS pool, all lanes
                        // loading in parallel: the vote itself then runs without touching HBM
                        const uint32_t b0 = EPL * (uint32_t)lane;
                        uint32_t need = 0;
#pragma unroll
                        for (int j = 0; j < EPL; ++j) if (b0 + j < nu && (sm.w.a.uval[b0 + j] & 1)) need += sm.u.v.nml[b0 + j];
                        uint32_t at = x.wave_excl_scan(need);
#ifdef DBTK_STAMPS
                        diag_need = x.wave_sum(need);
#endif
                        uint32_t* pool = sm.evd;  // (the parallel vote reads the pool before it writes evd over it)
#pragma unroll
                        for (int j = 0; j < EPL; ++j) {
                            const uint32_t u = b0 + j;
                            if (u < nu) {
                                const uint32_t v = sm.w.a.uval[u], nn = (v & 1) ? sm.u.v.nml[u] : 0;
                                if (nn && at + nn <= 2u * NHMAX) {
                                    sm.w.a.poff[u] = (uint16_t)at;
                                    for (uint32_t q = 0; q < nn; ++q) pool[at + q] = T.vv[(v >> 1) + 1 + q];
                                } else sm.w.a.poff[u] = 0xFFFFu;
                                at += nn;
                            }
                        }
                        x.sync();
                    }
                    const uint32_t n0 = nu ? x.uni(sm.u.v.nml[0]) : 0;
                    bool diff = false;
                    for (uint32_t u = lane; u < nu; u += 64) diff |= sm.u.v.nml[u] != n0;
                    alleq = x.ballot(diff) == 0;
                }
            }
            DBTK_STAMP(8);  // nml / single-locus test
            // ---- (round 6) the pair whose vote cannot go any other way: the fused probe kernels' rule (dbtk_locus.h has the argument,
            // tests/test_vote_rule.py checks it) for the pairs they do not take — more shared positions than the lean kernel looks classes
            // up for, lists too long for a locus' image.  Every found k-mer names ONE locus L (the unique ones by their value, every
            // multi-locus one by holding L in its list: the lists are in the LDS pool already), at least one is unique to L, both mates
            // pass kfilter: L is `top` whatever order std::sort leaves equal keys in, and countHit accepts it.  No sort, no vote.
            // (nm1 / nm2, the vote's partial sums, do depend on the order: trace mode votes.)
            bool decided = false;
            uint32_t Ldec = 0;
            if (!single && !a.P.trace && !rm[0] && !rm[1] && nhit[0] >= cth && nhit[1] >= cth && nhit[0] && nhit[1]) {
                uint32_t mx = 0, mn = 0xFFFFFFFFu;
                for (uint32_t u = lane; u < nu; u += 64) {
                    const uint32_t v = sm.w.a.uval[u];
                    if (!(v & 1u)) { mx = v > mx ? v : mx; mn = v < mn ? v : mn; }
                }
                const uint32_t wmx = ~x.wave_min(~mx), wmn = x.wave_min(mn);
                if (wmn != 0xFFFFFFFFu && wmx == wmn) {
                    Ldec = wmx >> 1;
                    const uint32_t* pool = sm.evd;
                    const uint32_t b0 = EPL * (uint32_t)lane;
                    bool bad = false;
#pragma unroll
                    for (int j = 0; j < EPL; ++j) {
                        const uint32_t u = b0 + j;
                        if (u < nu && (sm.w.a.uval[u] & 1u)) {
                            const uint32_t po = sm.w.a.poff[u], nn = sm.u.v.nml[u];
                            bool has = false;
                            if (po != 0xFFFFu) for (uint32_t q = 0; q < nn; ++q) has |= pool[po + q] == Ldec;
                            bad |= !has;  // (a list that did not fit the pool: not decided here)
                        }
                    }
                    decided = x.ballot(bad) == 0;
                }
            }
            // ---- P7 + P8: the permutation std::sort applies (AQ.cpp:320-327) and the vote
            const uint16_t* perm_row = T.permtab + (size_t)nu * (nu ? nu - 1 : 0) / 2;  // introsort of nu equal keys
            if (single) {
                const uint32_t fr = vote_single_locus<EPL>(x, perm_row, sm.w.a.dd, nu, cth);
                dst0 = x.uni(sm.w.a.uval[0]) >> 1;
                nm1 = (int)(fr & 0xFFFF); nm2 = (int)(fr >> 16);
            } else if (decided) {
                dst0 = Ldec;
                nm1 = (int)nhit[0]; nm2 = (int)nhit[1];  // (>= cth each: accepted below)
            } else {
                if (alleq) { for (uint32_t i = lane; i < nu; i += 64) sm.u.v.ord[i] = perm_row[i]; }
                else {
                    // exact std::sort on packed (nml << 9 | index) words held in the (not yet used) locus map
                    bool big = false;
                    for (uint32_t u = lane; u < nu; u += 64) big |= sm.u.v.nml[u] >= (1u << 23);
                    const bool packed = x.ballot(big) == 0;
                    if (packed) {
                        for (uint32_t u = lane; u < nu; u += 64) sm.u.v.lkey[u] = (sm.u.v.nml[u] << 9) | u;
                        x.sync();
                        // the whole wave sorts; lhit (idle until the vote) is its position scratch and then its output
                        uint16_t* const pos16 = reinterpret_cast<uint16_t*>(sm.w.a.lhit);
                        wave_gcc_sort_packed(x, sm.u.v.lkey, (int)nu, pos16, pos16 + NHMAX, sm.w.a.lhit, sm.stack);
                        for (uint32_t i = lane; i < nu; i += 64) sm.u.v.ord[i] = (uint16_t)(sm.w.a.lhit[i] & 0x1FF);
                        x.sync();
                        for (uint32_t i = lane; i < (uint32_t)LCAP; i += 64) sm.u.v.lkey[i] = NAN32;
                    } else if (lane == 0) gcc_sort_index(sm.u.v.ord, (int)nu, sm.u.v.nml, sm.stack);
                }
                x.sync();
                DBTK_STAMP(14);  // the sort of the vote order
                Asgn ptop{NAN32, 0, 0};
                uint64_t pvvw = 0;
                // event scratch: 2 * NHMAX words each (a k-mer has one event per locus; the loci lists come from the pool of 2 * NHMAX);
                // two of the three overlay the vote's input arrays, which are in registers by the time the events are written
                static_assert(sizeof(sm.u) >= 8 * NHMAX && sizeof(sm.w) >= 8 * NHMAX, "event scratch does not fit");
                if (vote_parallel<NHMAX / 64, 2 * NHMAX>(x, sm.u.v.ord, sm.w.a.uval, sm.w.a.dd, sm.u.v.nml, sm.w.a.poff, sm.evd,
                                                        reinterpret_cast<uint32_t*>(&sm.u), reinterpret_cast<uint32_t*>(&sm.w), sm.evd, nu, cth,
                                                        ptop, pvvw)) {
                    c_vote += pvvw;
                    dst0 = (uint32_t)ptop.idx;
                    nm1 = (int)ptop.fc; nm2 = (int)ptop.rc;
                } else {  // events do not fit in LDS: the literal loop on one lane
#ifdef DBTK_STAMPS
                    st_acc[44] += 1;
#endif
                    if (lane == 0) {
                        for (uint32_t u = 0; u < nu; ++u) sm.w.a.dd[u] &= 0x00FF00FFu;  // PE_KMC counts are uint8_t
                        // a row of the spill pool for this pair (as many rows as workgroups can be resident, so one is always free; the grid
                        // has several times as many workgroups): whoever held it before left its epoch
                        uint32_t row = x.bid() % a.vote_rows;
                        while (x.atomic_cas(&a.vote_busy[row], 0ull, 1ull) != 0ull) row = row + 1 < a.vote_rows ? row + 1 : 0u;
                        const uint32_t ep = DBTK_COH_LOAD(&a.vote_epoch[row]) + 1;
                        HitMap hmap{sm.u.v.lkey, sm.w.a.lhit, 0, a.vote_scratch + (size_t)row * ((size_t)nloci + 1), ep, false,
                                    (uint32_t)LCAP, LCAP == 512 ? 23u : 24u, (uint32_t)(LCAP * 3 / 4)};
                        Asgn top;
                        uint64_t nvvw = 0;
                        vote(T, sm.u.v.ord, sm.w.a.uval, sm.w.a.dd, (int)nu, cth, hmap, top, nvvw, sm.u.v.nml, sm.evd, sm.w.a.poff);
                        c_vote += nvvw;
                        if (hmap.spilled) DBTK_COH_STORE(&a.vote_epoch[row], ep);
                        DBTK_COH_RELEASE();
                        DBTK_COH_STORE(&a.vote_busy[row], 0ull);
                        sm.res[0] = (int32_t)(uint32_t)top.idx;
                        sm.res[1] = (int32_t)top.fc;
                        sm.res[2] = (int32_t)top.rc;
                    }
                    x.sync();
                    dst0 = x.uni((uint32_t)sm.res[0]);
                    nm1 = (int)x.uni((uint32_t)sm.res[1]); nm2 = (int)x.uni((uint32_t)sm.res[2]);
                }
            }
            DBTK_STAMP(single ? 9 : 10);  // vote: fast / general
#ifdef DBTK_STAMPS
            st_acc[single ? 24 : (alleq ? 25 : 26)] += 1;  // pairs per vote path
            st_acc[27] += nu; st_acc[28] += n;
            diag_nu = nu;
#endif
            }  // general path
            {  // countHit's accept test, AQ.cpp:439-451
                const uint64_t fc = (uint64_t)(uint32_t)nm1, rc = (uint64_t)(uint32_t)nm2;
                const bool test1 = fc >= cth && rc >= cth, test2 = (fc + rc) >= 2ull * cth;
                if ((test1 || test2) && dst0 != NAN32) dst = dst0;
                else { hf[0] = 1 & !rm[0]; hf[1] = 1 & !rm[1]; rm[0] = 1; rm[1] = 1; dst = nloci; }
            }
            c_hf += (uint64_t)(hf[0] + hf[1]);
            stage = DBTK_STAGE_LOCUS;
            if (dst != nloci) {
                if (a.P.qc && T.qc && !T.qc[dst]) {  // AQ.cpp:2059-2062
                    c_qc += (uint64_t)(2 - rm[0] - rm[1]);
                    stage = DBTK_STAGE_QC;
                } else if (a.P.threading) {  // AQ.cpp:2070-2090: `alned` stays false at HEAD, nothing else happens;
                    c_thr += 2;              // v1.3 (threading = 2): the walk kernel takes the pair from here
                    if (a.P.threading == DBTK_THREADING_V13) { stage = DBTK_STAGE_THREADING; if (lane == 0) a.walk_dst[t] = dst; }
                } else if (a.P.extract) {  // AQ.cpp:2094-2099
                    c_thr += 2; c_feas += 2;
                    stage = DBTK_STAGE_EXTRACT;
                } else {
                    c_thr += 2; c_feas += 2;
                    if (a.P.bait) {
                        // ---- bait gate (bfilter_FPSv1, AQ.cpp:1377-1419; call site 2111-2126): a mate is flagged when a
                        // k-mer of baitDB[destLocus] occurs in it (uint8_t count, over quality-passing positions when the
                        // input has qualities) outside [min, max]; either flag removes both mates.
                        int bfl[2] = {0, 0};
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            uint64_t qm[4] = {~0ull, ~0ull, ~0ull, ~0ull};
                            if (a.qmaskbuf) {
#pragma unroll
                                for (int q = 0; q < 4; ++q) qm[q] = a.qmaskbuf[((size_t)2 * (t - a.t0) + m) * 4 + q];
                            }
                            uint32_t th[NSLOT];
                            bool on[NSLOT];
#pragma unroll
                            for (int s = 0; s < NSLOT; ++s) {
                                const uint32_t i = 64 * s + lane;
                                on[s] = i < nkm[m] && km[m][s] != NAN64 && ((qm[s] >> lane) & 1);
                                th[s] = on[s] ? kl_lookup(T.bait, T.bait_mask, T.bait_shift, km[m][s], dst) : CLS_NONE;
                            }
#pragma unroll
                            for (int s = 0; s < NSLOT; ++s) {
                                uint64_t pend = x.ballot(th[s] != CLS_NONE);
                                while (pend) {  // one bait k-mer at a time: its multiplicity among the mate's counted positions
                                    const int src = (int)__builtin_ctzll(pend);
                                    const uint64_t K = ((uint64_t)x.bcast((uint32_t)(km[m][s] >> 32), src) << 32) | x.bcast((uint32_t)km[m][s], src);
                                    const uint32_t thr = x.bcast(th[s], src);
                                    uint32_t cnt = 0;
#pragma unroll
                                    for (int s2 = 0; s2 < NSLOT; ++s2) cnt += (uint32_t)__builtin_popcountll(x.ballot(on[s2] && km[m][s2] == K));
                                    cnt &= 0xFF;  // kc8_t counts in uint8_t
                                    const uint32_t mi = (thr >> 8) & 0xFF, ma = thr & 0xFF;
                                    if (cnt < mi || cnt > ma) bfl[m] = 1;
                                    pend &= ~x.ballot(th[s] != CLS_NONE && km[m][s] == K);
                                }
                            }
                        }
                        bf[0] = bfl[0]; bf[1] = bfl[1];
                        if (bf[0] || bf[1]) {
                            c_bait += (uint64_t)((bf[0] & !rm[0]) + (bf[1] & !rm[1]));
                            rm[0] = 1; rm[1] = 1;
                        }
                    }
                    // ---- P10: assignTRkmc against the DBs of destLocus0 (AQ.cpp:2138-2144).  State of a
                    // position: flank 1 beats TR 2 (AQ.cpp:1467-1468).  With a consistent RPGG the class of a
                    // single-locus k-mer rides in the index slot (`aux`): no second probe.
                    uint32_t slot[2][NSLOT], ntr[2] = {0, 0};
                    uint64_t Tw[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};  // transitions between known states
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        ms[m].rm = rm[m];
                        if (!okam && rm[m]) continue;
                        nas[m] = nkm[m];
                        c_cls += nkm[m];
                        uint32_t mytr = 0, carry = 0;  // carry: last known state before this word
#pragma unroll
                        for (int s = 0; s < NSLOT; ++s) {
                            slot[m][s] = CLS_NONE;
                            if ((uint32_t)s >= nsl) continue;
                            const uint32_t i = 64 * s + lane;
                            uint32_t c = CLS_NONE;
                            if (i < nkm[m] && km[m][s] != NAN64) {
                                const uint32_t v = hv[m][s];
                                if (T.consistent && v == NOHIT) c = CLS_NONE;
                                else if (T.consistent && !(v & 1)) c = ((v >> 1) == dst0) ? ha[m][s] : CLS_NONE;
                                else c = cls_lookup(T, km[m][s], dst0);
                            }
                            slot[m][s] = c;
                            const uint32_t st = (c == CLS_NONE) ? 0u : (c == CLS_FLANK ? 1u : 2u);
                            const uint64_t kb = x.ballot(st != 0), rb = x.ballot(st == 2);
                            Kw[m][s] = kb; Rw[m][s] = rb;
                            mytr += (uint32_t)__builtin_popcountll(rb);
                            Tw[m][s] = transitions_word(kb, rb, carry);  // known position whose last known predecessor has the other state
                        }
                        ntr[m] = mytr & 0xFF;  // uint8_t ntr, AQ.cpp:1454 (mytr is wave-uniform)
                    }
                    DBTK_STAMP(11);  // states
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        if (!nas[m]) continue;
                        // K / R / T are wave-uniform ballots: the rest of the state machine is scalar work
                        Bits256 K, R, Tm;
#pragma unroll
                        for (int q = 0; q < 4; ++q) { K.w[q] = Kw[m][q]; R.w[q] = Rw[m][q]; Tm.w[q] = Tw[m][q]; }
                        assign_masks(K, R, Tm, (int)nas[m], ntr[m], a.P, ms[m]);
                        af[m] = ms[m].af; rm[m] = ms[m].rm;
                    }
                    DBTK_STAMP(12);  // assign_bits
                    // ---- the next item's probe results, before anything of this one goes out to memory
                    deliver();
                    delivered = true;
                    // ---- P11: accumulate (AQ.cpp:2145-2158)
                    if (rm[0] && rm[1]) { dst = nloci; stage = (bf[0] || bf[1]) ? DBTK_STAGE_BAIT : DBTK_STAGE_ASGN; }
                    else {
                        stage = DBTK_STAGE_COUNTED;
                        const int nmap = 2 - rm[0] - rm[1];
                        if (lane == 0) {
                            x.atomic_add(&a.nmapread[dst], (uint64_t)nmap);
                            x.atomic_add(&a.kmc[dst], (uint64_t)(int64_t)((ms[0].ei - ms[0].si) + (ms[1].ei - ms[1].si)));
                        }
                        c_asgn += (uint64_t)nmap;
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            if (rm[m]) continue;
#pragma unroll
                            for (int s = 0; s < NSLOT; ++s) {
                                const uint32_t c = slot[m][s];
                                if (c != CLS_NONE && c != CLS_FLANK) x.atomic_add(&a.counts[c], 1ull);
                                c_inc += (uint64_t)__builtin_popcountll(Rw[m][s]);
                            }
                        }
                        if (a.P.bubbles && a.edgebuf) {
                            // ---- novel edges of the kept mates (countNovelEdges, AQ.cpp:1559-1567; call site 2161-2166):
                            // read (k+1)-mers at positions [si_, ei_ - 1) that are not in trEdgeDB[destLocus]
#pragma unroll
                            for (int m = 0; m < 2; ++m) {
                                if (rm[m]) continue;
                                const uint64_t* ein = a.edgebuf + ((size_t)2 * (t - a.t0) + m) * a.nkp;
                                const int lo = ms[m].si_, hi = ms[m].ei_ - 1;
#pragma unroll
                                for (int s = 0; s < NSLOT; ++s) {
                                    const int i = 64 * s + lane;
                                    if (i >= lo && i < hi) {
                                        const uint64_t e = ein[i];
                                        if (e != NAN64 && kl_lookup(T.tre, T.tre_mask, T.tre_shift, e, dst) == CLS_NONE) {
                                            const uint32_t at = x.atomic_add(a.nevents, 1u);
                                            if (at < a.events_cap) a.events[at] = BubEvent{pair, (uint32_t)m, (uint32_t)i, dst, e};
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
        DBTK_STAMP(13);  // accumulate
        // ---- P12: record (kam: AQ.cpp:2169-2175; trace: every pair)
        const bool want = RECS && a.recs && (a.P.trace || (okam && stage == DBTK_STAGE_COUNTED) ||
                                     (okam && a.P.simmode && (stage == DBTK_STAGE_ASGN || stage == DBTK_STAGE_BAIT)) ||
                                     (a.P.extract && stage == DBTK_STAGE_EXTRACT) || (a.P.trackbait && stage == DBTK_STAGE_BAIT));
        if (!delivered) deliver();
        if (want) emit_pair_record(x, a, lane, pair, stage, dst, dst0, nm1, nm2, ms, kf, hf, bf, af, rm, nas, Kw, Rw);
#ifdef DBTK_STAMPS
        {
            const uint64_t now_ = x.clock();
            if (lane == 0 && a.dbg) {
                x.atomic_max(&a.dbg[45], now_ - pair_t0_);
                // the slowest pair with its size: cycles << 32 | events of multi-locus k-mers << 12 | distinct k-mers
                x.atomic_max(&a.dbg[47], ((now_ - pair_t0_) << 32) | ((uint64_t)(diag_need > 0xFFFFFu ? 0xFFFFFu : diag_need) << 12) | (diag_nu & 0xFFF));
                // histogram of pair times: 2^b cycles
                const uint64_t dcy = now_ - pair_t0_;
                uint32_t bkt = 0;
                while (bkt < 5 && (dcy >> (12 + 2 * bkt))) ++bkt;
                x.atomic_add(&a.dbg[bkt < 3 ? 21 + bkt : 26 + bkt], 1ull);
            }
            diag_need = 0; diag_nu = 0;
            pair_t0_ = now_;
            st_acc[46] += 1;
        }
#endif
        t = tnext;
    }
    DBTK_STAMP_FLUSH;
    if (lane == 0) {
        if (c_kf) x.atomic_add(&ctr[DBTK_C_KMERFILTERED], c_kf);
        if (c_hf) x.atomic_add(&ctr[DBTK_C_LOCUSFILTERED], c_hf);
        if (c_qc) x.atomic_add(&ctr[DBTK_C_QCFILTERED], c_qc);
        if (c_thr) x.atomic_add(&ctr[DBTK_C_THREADING], c_thr);
        if (c_feas) x.atomic_add(&ctr[DBTK_C_FEASIBLE], c_feas);
        if (c_asgn) x.atomic_add(&ctr[DBTK_C_ASGN], c_asgn);
        if (c_bait) x.atomic_add(&ctr[DBTK_C_BAITFILTERED], c_bait);
        if (c_vv) x.atomic_add(&ctr[DBTK_C_ALGO_VV], c_vv);
        if (c_vote && a.pstats) x.atomic_add(&a.pstats[DBTK_PS_VOTE_VV], c_vote);
        if (c_vv && a.pstats) x.atomic_add(&a.pstats[DBTK_PS_PAIR_VV], c_vv);
        if (c_cls) x.atomic_add(&ctr[DBTK_C_ALGO_CLS], c_cls);
        if (c_inc) x.atomic_add(&ctr[DBTK_C_ALGO_INC], c_inc);
        if (c_nhash1) {
            x.atomic_add(&ctr[DBTK_C_NHASH1], c_nhash1);
            x.atomic_add(&ctr[DBTK_C_ALGO_PROBES], c_nhash1);
        }
    }
}

}  // namespace dbtk
#endif
