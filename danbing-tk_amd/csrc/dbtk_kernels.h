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
//   wave_sum(u32) wave_excl_scan(u32) bcast(u32, srclane)
//   atomic_add(u64*,u64) atomic_add(u32*,u32)->old atomic_cas(u64*,exp,des)->old
//   atomic_max(u64*,u64) atomic_or(u64*,u64) lds_add(u32*,u32)->old lds_or(u32*,u32)
//   smem<T>()
//
// Kernels:
//   body_idx_insert / body_idx_finalize / body_cls_insert  table build in HBM
//   body_encode_subfilter   K1: 2-bit encode + subfilter, 256-thread blocks, 64-pair tiles
//   body_pair               K2..K4: one wavefront per surviving pair:
//                           kfilter probe -> dedup -> introsort -> vote -> assign -> count
#ifndef DBTK_KERNELS_H_
#define DBTK_KERNELS_H_

#include "dbtk_sort.h"
#include "dbtk_tables.h"

namespace dbtk {

// ------------------------------------------------------------------ build --
struct IdxBuildArgs {
    IdxSlot* slots;
    uint64_t mask;
    uint32_t shift;
    const uint64_t* keys;
    const uint32_t* vals;
    uint64_t n;
};

template <class X>
DBTK_HD void body_idx_insert(X& x, const IdxBuildArgs& a) {
    for (uint64_t i = (uint64_t)x.bid() * x.nthreads() + x.tid(); i < a.n; i += (uint64_t)x.nblocks() * x.nthreads()) {
        const uint64_t key = a.keys[i];
        uint64_t s = hash_idx(key, a.shift);
        for (;;) {
            const uint64_t prev = x.atomic_cas(&a.slots[s].key, NAN64, key);
            if (prev == NAN64 || prev == key) {
                x.atomic_max(&a.slots[s].val, (i << 32) | a.vals[i]);  // kmerDBi[key] = val: last one wins
                break;
            }
            s = (s + 1) & a.mask;
        }
    }
}
template <class X>
DBTK_HD void body_idx_finalize(X& x, IdxSlot* slots, uint64_t cap) {
    for (uint64_t i = (uint64_t)x.bid() * x.nthreads() + x.tid(); i < cap; i += (uint64_t)x.nblocks() * x.nthreads())
        slots[i].val &= 0xFFFFFFFFull;
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
                if (plc == ~0ull) break;               // slot is ours
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
    uint32_t* ticket;     // pair kernel work counter
    uint64_t* counts;     // OUT.trkmc.ar order
    uint64_t* kmc;
    uint64_t* nmapread;   // widened; low 32 bits are the reference's uint32 counter
    uint64_t* counters;   // DBTK_C_*
    dbtk_pair_rec_t* recs;  // nullptr, or rec_cap records
    uint32_t* nrec;         // kam mode: compaction counter (may exceed rec_cap)
    uint32_t rec_cap;
    uint32_t* errflag;
    uint64_t* vote_scratch;  // per block: nloci+1 stamped hit words (see vote)
    uint32_t* vote_epoch;    // per block
};

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

// ======================================================================= K1 =
// encode + subfilter (read2kmers_edges' validity + subfilter,
// src/aQueryFasta_thread.h:274-311, src/aQueryFasta_thread.cpp:172-188, 2035-2051).
// A 256-thread block takes tiles of K1_TP consecutive pairs: the tile's bytes
// are contiguous in the batch, so they are fetched with 16-byte coalesced loads
// and packed straight into LDS (2 bits + 1 validity bit per base); the
// N_FILTER sampled windows of mate 1 are probed by 4 lanes per pair, mate 2 only
// for pairs whose mate 1 passed — the same probes the reference performs.
constexpr int K1_NT = 256;
constexpr int K1_TP = 64;                              // pairs per tile
constexpr int K1_CH = K1_TP * 2 * MAXL / 16 + 4;       // 16-base chunks per tile (+ slack)
struct K1Smem {
    uint32_t pk[K1_CH];
    uint16_t vd[K1_CH];
    uint32_t bpos[2 * K1_TP];  // stream position of each read's first base
    uint16_t rlen[2 * K1_TP];
    uint8_t any[2 * K1_TP];    // read has at least one valid k-mer window
    uint32_t hm1[K1_TP], hm2[K1_TP];  // hit masks of the sampled positions
    uint32_t cnt[8];           // block counters: nshort, nsub, nhash0, nprobe, nsurv, nbases
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

template <class X>
DBTK_HD void body_encode_subfilter(X& x, const BatchArgs& a) {
    K1Smem& sm = *x.template smem<K1Smem>();
    const int tid = x.tid();
    const uint32_t k = a.P.ksize, NF = a.P.n_filter, NM = a.P.nm_filter;
    const bool dosub = NF && NM;
    if (tid < 8) sm.cnt[tid] = 0;
    if (tid == 0 && x.bid() == 0) x.atomic_add(&a.counters[DBTK_C_NREADS], 2 * a.npairs);  // nReads, AQ.cpp:1977
    x.sync();
    for (uint64_t tile = x.bid(); tile * K1_TP < a.npairs; tile += x.nblocks()) {
        const uint64_t p0 = tile * K1_TP;
        const uint32_t np = (uint32_t)((a.npairs - p0 < (uint64_t)K1_TP) ? a.npairs - p0 : K1_TP);
        const uint64_t B0 = a.off[2 * p0], B1 = a.off[2 * (p0 + np)];
        const uint64_t A0 = B0 & ~15ull;
        const uint32_t nch = (uint32_t)((B1 - A0 + 15) >> 4);
        if (nch + 3 > (uint32_t)K1_CH) {  // a read longer than DBTK_MAX_READ_LEN slipped through
            if (tid == 0) *a.errflag = DBTK_ERR_READ_TOO_LONG;
            continue;
        }
        // A: pack the tile
        for (uint32_t c = tid; c < nch + 3; c += K1_NT) {
            uint32_t pk = 0, vd = 0;
            if (c < nch) {
                const uint64_t g = A0 + 16ull * c;
                uint32_t w[4];
                if (g + 16 <= a.seq_len) {
                    const uint4 q = *reinterpret_cast<const uint4*>(a.seq + g);
                    w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
                } else {
                    for (int j = 0; j < 4; ++j) {
                        w[j] = 0;
                        for (int b = 0; b < 4; ++b)
                            if (g + 4 * j + b < a.seq_len) w[j] |= (uint32_t)a.seq[g + 4 * j + b] << (8 * b);
                    }
                }
                pk = pack16(w, &vd);
            }
            sm.pk[c] = pk;
            sm.vd[c] = (uint16_t)vd;
        }
        if (tid < K1_TP) { sm.hm1[tid] = 0; sm.hm2[tid] = 0; }
        x.sync();
        // B: per-read geometry and "has a valid window" (caks.size() != 0, AQ.cpp:2037)
        if ((uint32_t)tid < 2 * np) {
            const uint64_t o0 = a.off[2 * p0 + tid], o1 = a.off[2 * p0 + tid + 1];
            const uint32_t len = (uint32_t)(o1 - o0), b = (uint32_t)(o0 - A0);
            sm.bpos[tid] = b;
            sm.rlen[tid] = (uint16_t)len;
            x.lds_add(&sm.cnt[5], len);
            sm.any[tid] = any_valid_window(sm.vd, b, len, k);
        }
        x.sync();
        // C/D: sampled probes, 4 lanes per pair; mate index 0 = seq1 first
        for (int mate = 0; mate < 2 && dosub; ++mate) {
            const uint32_t j = (uint32_t)tid >> 2;
            if (j < np && sm.any[2 * j] && sm.any[2 * j + 1]) {
                bool go = true;
                if (mate == 1) go = (uint32_t)__builtin_popcount(sm.hm1[j]) >= NM;
                if (go) {
                    const uint32_t r = 2 * j + mate;
                    const uint32_t L = sm.rlen[r] - k + 1, S = L / (NF - 1);
                    for (uint32_t s = (uint32_t)tid & 3; s < NF; s += 4) {
                        const uint32_t pos = (s != NF - 1) ? s * S : L - 1;
                        const uint64_t km = window_kmer(sm.pk, sm.vd, sm.bpos[r] + pos, k, nullptr, nullptr);
                        if (km != NAN64 && idx_lookup(a.T, km) != NOHIT) x.lds_or(mate ? &sm.hm2[j] : &sm.hm1[j], 1u << s);
                    }
                }
            }
            x.sync();
        }
        // E: verdicts
        if ((uint32_t)tid < np) {
            const uint32_t j = tid;
            const uint32_t pair = (uint32_t)(p0 + j);
            uint32_t stage = 0xFFFFFFFFu;
            if (!sm.any[2 * j] || !sm.any[2 * j + 1]) {
                stage = DBTK_STAGE_SHORT;
                x.lds_add(&sm.cnt[0], 1);
            } else if (dosub) {
                uint32_t nhash = 0, nprobe = 0, h = 0;
                bool brk = false;
                for (uint32_t i = 0; i < NF; ++i) {  // AQ.cpp:176-180: ++nhash only when the loop continues
                    h += (sm.hm1[j] >> i) & 1; ++nprobe;
                    if (h >= NM) { brk = true; break; }
                    ++nhash;
                }
                bool rej = !brk;
                if (!rej) {
                    h = 0; brk = false;
                    for (uint32_t i = 0; i < NF; ++i) {
                        h += (sm.hm2[j] >> i) & 1; ++nprobe;
                        if (h >= NM) { brk = true; break; }
                        ++nhash;
                    }
                    rej = !brk;
                }
                x.lds_add(&sm.cnt[2], nhash);
                x.lds_add(&sm.cnt[3], nprobe);
                if (rej) { stage = DBTK_STAGE_SUBFILTER; x.lds_add(&sm.cnt[1], 2); }
            }
            if (stage == 0xFFFFFFFFu) {
                const uint32_t at = x.atomic_add(a.nsurv, 1u);
                a.surv[at] = pair;
                x.lds_add(&sm.cnt[4], 1);
            } else if (a.P.trace && a.recs) {
                write_early_rec(&a.recs[pair], pair, stage, a.T.nloci);
            }
        }
        x.sync();
    }
    if (tid == 0) {
        if (sm.cnt[0]) x.atomic_add(&a.counters[DBTK_C_NSHORT], (uint64_t)sm.cnt[0]);
        if (sm.cnt[1]) x.atomic_add(&a.counters[DBTK_C_SUBFILTERED], (uint64_t)sm.cnt[1]);
        if (sm.cnt[2]) x.atomic_add(&a.counters[DBTK_C_NHASH0], (uint64_t)sm.cnt[2]);
        if (sm.cnt[3]) x.atomic_add(&a.counters[DBTK_C_ALGO_PROBES], (uint64_t)sm.cnt[3]);
        if (sm.cnt[4]) x.atomic_add(&a.counters[DBTK_C_SURVIVORS], (uint64_t)sm.cnt[4]);
        if (sm.cnt[5]) x.atomic_add(&a.counters[DBTK_C_BASES], (uint64_t)sm.cnt[5]);
    }
}

// ================================================================= K2 .. K4 =
// One wavefront (64-thread block) per surviving pair.
constexpr int LCAP = 512;        // per-pair locus map in LDS (vote); spills to vote_scratch
constexpr int LLIMIT = 384;
struct PairSmem {
    uint32_t pk[2][20];
    uint16_t vd[2][20];
    uint64_t kmer[2][NKMAX];   // canonical k-mers by read position (caks1 / caks2)
    uint32_t hval[2][NKMAX];   // index val by position; later the class/slot of the position
    union {
        struct { uint64_t skey[NHMAX]; uint16_t sinfo[NHMAX]; } s;                     // hit list being sorted
        struct { uint32_t nml[NHMAX]; uint32_t lkey[LCAP]; uint16_t ord[NHMAX]; } v;    // vote phase
    } u;
    uint32_t uval[NHMAX];      // unique k-mers in ascending key order: index val
    uint32_t dd[NHMAX];        // PE_KMC dup: count in mate 0 | count in mate 1 << 16
    uint32_t lhit[LCAP];
    uint8_t as[2][NKMAX];
    int stack[3 * 40];
    // scalars
    uint32_t len[2], nk[2];
    int32_t res[16];           // vote result: tri, tri0, fc, rc ...
    int32_t mres[2][12];       // per-mate assign results
};

struct MateState {  // km_asgn_read_t fields the state machine writes, AQ.cpp:93-108
    int si, ei, nt, bs, ti, si_, ei_, af, rm;
};

// assignTRkmc's scan (src/aQueryFasta_thread.cpp:1470-1555) over the states
// as[0..nk); ntr = number of TR states, already reduced to uint8_t.
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
};
DBTK_HD uint32_t hitmap_add(HitMap& m, uint32_t locus, uint32_t add) {  // returns the new h1 | h2<<16
    if (!m.spilled) {
        uint32_t i = (locus * 0x9E3779B1u) >> 23;  // LCAP = 512
        for (;;) {
            if (m.lkey[i] == locus) { m.lhit[i] += add; return m.lhit[i]; }
            if (m.lkey[i] == NAN32) break;
            i = (i + 1) & (LCAP - 1);
        }
        if (m.n < (uint32_t)LLIMIT) { m.lkey[i] = locus; m.lhit[i] = add; ++m.n; return add; }
        for (int j = 0; j < LCAP; ++j)  // migrate
            if (m.lkey[j] != NAN32) m.g[m.lkey[j]] = ((uint64_t)m.epoch << 32) | m.lhit[j];
        m.spilled = true;
    }
    const uint64_t w = m.g[locus];
    const uint32_t cur = ((uint32_t)(w >> 32) == m.epoch) ? (uint32_t)w : 0u;
    const uint32_t nw = cur + add;
    m.g[locus] = ((uint64_t)m.epoch << 32) | nw;
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

// find_matching_locus + the accept test of countHit
// (src/aQueryFasta_thread.cpp:364-422, 436-451) on the permuted unique list.
DBTK_HD void vote(const DevTables& T, const uint16_t* ord, const uint32_t* uval, const uint32_t* dd, int nu, uint32_t cth,
                  HitMap& hm, Asgn& top, uint64_t& nvvw) {
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
            const uint32_t n = T.vv[vi >> 1];
            nvvw += 1 + n;
            for (uint32_t j = 0; j < n; ++j) {
                const uint32_t locus = T.vv[(vi >> 1) + 1 + j];
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
            while ((top.fc < cth && cth - top.fc <= remj) || (top.rc < cth && cth - top.rc <= remj)) {  // get_acm1
                if (++j >= nu) break;
                const uint32_t uj = ord[j], vj = uval[uj];
                const uint32_t e1 = dd[uj] & 0xFF, e2 = (dd[uj] >> 16) & 0xFF;
                remj -= e1 + e2;
                if (vj & 1) {
                    const uint32_t n = T.vv[vj >> 1];
                    nvvw += 1;
                    for (uint32_t q = 0; q < n; ++q) {
                        nvvw += 1;
                        if (T.vv[(vj >> 1) + 1 + q] == top.idx) { top.fc += e1; top.rc += e2; break; }
                    }
                } else if ((vj >> 1) == top.idx) {
                    top.fc += e1; top.rc += e2;
                }
            }
            break;
        }
    }
}

template <class X>
DBTK_HD void body_pair(X& x, const BatchArgs& a) {
    PairSmem& sm = *x.template smem<PairSmem>();
    const int lane = x.lane();
    const DevTables& T = a.T;
    const uint32_t k = T.ksize, cth = a.P.cthreshold, nloci = T.nloci;
    const bool okam = a.P.okam != 0;
    // per-block counters, flushed once at the end
    uint64_t c_kf = 0, c_hf = 0, c_qc = 0, c_thr = 0, c_feas = 0, c_asgn = 0, c_nhash1 = 0, c_vv = 0, c_cls = 0, c_inc = 0;

    for (;;) {
        uint32_t t = 0;
        if (lane == 0) t = x.atomic_add(a.ticket, 1u);
        t = x.bcast(t, 0);
        if (t >= *a.nsurv) break;
        const uint32_t pair = a.surv[t];
        x.sync();  // previous pair's LDS is dead from here on

        // ---- P0: load + pack both reads (lanes 0..15 mate 0, 16..31 mate 1)
        if (lane < 32) {
            const int m = lane >> 4, c = lane & 15;
            const uint64_t o0 = a.off[2 * (uint64_t)pair + m], o1 = a.off[2 * (uint64_t)pair + m + 1];
            uint32_t len = (uint32_t)(o1 - o0);
            if (len > (uint32_t)MAXL) { *a.errflag = DBTK_ERR_READ_TOO_LONG; len = MAXL; }  // stay inside LDS
            uint32_t w[4] = {0, 0, 0, 0};
            for (int b = 0; b < 16; ++b) {
                const uint32_t p = 16 * c + b;
                if (p < len) w[b >> 2] |= (uint32_t)a.seq[o0 + p] << (8 * (b & 3));
            }
            uint32_t vd;
            sm.pk[m][c] = pack16(w, &vd);
            sm.vd[m][c] = (uint16_t)vd;  // bytes past the read are 0 -> invalid
            if (c < 4) { sm.pk[m][16 + c] = 0; sm.vd[m][16 + c] = 0; }
            if (c == 0) { sm.len[m] = len; sm.nk[m] = len >= k ? len - k + 1 : 0; }
        }
        x.sync();
        const uint32_t nk0 = sm.nk[0], nk1 = sm.nk[1];

        // ---- P1: canonical k-mers by position (read2kmers_edges, AQ.h:274-311)
        for (int m = 0; m < 2; ++m) {
            const uint32_t nk = m ? nk1 : nk0;
            for (uint32_t i = lane; i < nk; i += 64) sm.kmer[m][i] = window_kmer(sm.pk[m], sm.vd[m], i, k, nullptr, nullptr);
        }
        x.sync();

        // ---- P3: kfilter (AQ.cpp:190-224): probe every position; a mate fails
        // when its misses exceed nk - Cth; nhash1 counts probes up to the abort.
        int kf[2], rm[2], hf[2] = {0, 0}, af[2] = {0, 0};
        uint32_t nhit[2] = {0, 0};
        kf[0] = nk0 < cth; kf[1] = nk1 < cth;
        rm[0] = kf[0]; rm[1] = kf[1];
        if (!(rm[0] && rm[1])) {
            for (int m = 0; m < 2; ++m) {
                if (rm[m]) continue;
                const uint32_t nk = m ? nk1 : nk0;
                const uint32_t maxns = nk - cth;
                uint32_t cum_miss = 0, hits = 0, abort_at = nk;  // abort_at: position of the (maxns+1)-th miss
                for (uint32_t base = 0; base < nk; base += 64) {
                    const uint32_t i = base + lane;
                    uint32_t v = NOHIT;
                    if (i < nk) {
                        const uint64_t km = sm.kmer[m][i];
                        if (km != NAN64) v = idx_lookup(T, km);
                        sm.hval[m][i] = v;
                    }
                    const uint64_t missmask = x.ballot(i < nk && v == NOHIT);
                    const uint32_t nm = (uint32_t)__builtin_popcountll(missmask);
                    if (abort_at == nk && cum_miss + nm > maxns) {
                        uint64_t mm = missmask;  // the (maxns + 1 - cum_miss)-th set bit
                        for (uint32_t r = maxns - cum_miss; r > 0; --r) mm &= mm - 1;
                        abort_at = base + (uint32_t)__builtin_ctzll(mm);
                    }
                    cum_miss += nm;
                    hits += (uint32_t)__builtin_popcountll(x.ballot(i < nk && v != NOHIT));
                }
                if (abort_at != nk) {  // its.clear(); kf = 1
                    kf[m] = 1; rm[m] = 1;
                    c_nhash1 += abort_at + 1;
                } else {
                    nhit[m] = hits;
                    c_nhash1 += nk;
                }
            }
        }
        c_kf += (uint64_t)(kf[0] + kf[1]);
        x.sync();

        uint32_t stage = DBTK_STAGE_KFILTER, dst = nloci, dst0 = NAN32;
        int nm1 = 0, nm2 = 0;
        MateState ms[2];
        for (int m = 0; m < 2; ++m) ms[m] = MateState{-1, -1, 0, 0, -1, -1, -1, 0, 0};
        uint32_t nas[2] = {0, 0};

        if (!(rm[0] && rm[1])) {
            // ---- P4: gather the hit lists (its1 ++ its2 with the orient bit, AQ.cpp:263-266)
            uint32_t n = 0;
            for (int m = 0; m < 2; ++m) {
                if (rm[m]) continue;
                const uint32_t nk = m ? nk1 : nk0;
                for (uint32_t base = 0; base < nk; base += 64) {
                    const uint32_t i = base + lane;
                    const bool hit = i < nk && sm.hval[m][i] != NOHIT;
                    const uint64_t hm = x.ballot(hit);
                    if (hit) {
                        const uint32_t at = n + (uint32_t)__builtin_popcountll(hm & ((1ull << lane) - 1));
                        sm.u.s.skey[at] = sm.kmer[m][i];
                        sm.u.s.sinfo[at] = (uint16_t)((m << 8) | i);
                    }
                    n += (uint32_t)__builtin_popcountll(hm);
                }
            }
            x.sync();
            // ---- P5: sort by key (rank sort; (key, info) is a strict total order),
            // then run-length encode into unique k-mers + PE_KMC dups (AQ.cpp:268-295)
            {
                uint64_t myk[8]; uint16_t myi[8]; uint32_t rk[8];
                const int nown = (int)((n + 63 - lane) / 64);  // entries lane, lane+64, ...
                for (int j = 0; j < 8; ++j) {
                    rk[j] = 0;
                    if (j < nown) { myk[j] = sm.u.s.skey[lane + 64 * j]; myi[j] = sm.u.s.sinfo[lane + 64 * j]; }
                    else { myk[j] = 0; myi[j] = 0; }
                }
                for (uint32_t f = 0; f < n; ++f) {
                    const uint64_t fk = sm.u.s.skey[f];
                    const uint16_t fi = sm.u.s.sinfo[f];
                    for (int j = 0; j < 8; ++j) rk[j] += (fk < myk[j]) || (fk == myk[j] && fi < myi[j]);
                }
                x.sync();
                for (int j = 0; j < 8; ++j)
                    if (j < nown) { sm.u.s.skey[rk[j]] = myk[j]; sm.u.s.sinfo[rk[j]] = myi[j]; }
                for (uint32_t i = lane; i < (uint32_t)NHMAX; i += 64) sm.dd[i] = 0;
                x.sync();
            }
            uint32_t nu;
            {
                // contiguous ownership: lane owns sorted entries [8*lane, 8*lane+8)
                const uint32_t b0 = 8 * (uint32_t)lane;
                uint32_t heads = 0;
                uint64_t prevk = (b0 > 0 && b0 <= n) ? sm.u.s.skey[b0 - 1] : 0;
                uint64_t ks[8];
                for (int j = 0; j < 8; ++j) {
                    const uint32_t r = b0 + j;
                    ks[j] = r < n ? sm.u.s.skey[r] : 0;
                    if (r < n && (r == 0 || ks[j] != (j ? ks[j - 1] : prevk))) ++heads;
                }
                uint32_t uidx = x.wave_excl_scan(heads);  // unique index of my first head
                nu = x.wave_sum(heads);
                // a non-head entry belongs to the most recent head at or before it
                for (int j = 0; j < 8; ++j) {
                    const uint32_t r = b0 + j;
                    if (r >= n) break;
                    const bool head = (r == 0 || ks[j] != (j ? ks[j - 1] : prevk));
                    if (head) ++uidx;
                    const uint32_t u = uidx - 1;
                    const uint16_t info = sm.u.s.sinfo[r];
                    if (head) sm.uval[u] = sm.hval[info >> 8][info & 0xFF];
                    x.lds_add(&sm.dd[u], (info >> 8) ? 0x10000u : 1u);
                }
            }
            x.sync();
            // ---- P6: number of mapped loci per unique k-mer (AQ.cpp:311-317); skey/sinfo are dead now
            bool alleq = true;
            for (uint32_t u = lane; u < nu; u += 64) {
                const uint32_t v = sm.uval[u];
                const uint32_t c = (v & 1) ? T.vv[v >> 1] : 1u;
                sm.u.v.nml[u] = c;
            }
            {
                uint32_t odd = 0;
                for (uint32_t u = lane; u < nu; u += 64) odd += sm.uval[u] & 1;
                c_vv += x.wave_sum(odd);
            }
            for (uint32_t i = lane; i < (uint32_t)LCAP; i += 64) sm.u.v.lkey[i] = NAN32;
            x.sync();
            {
                const uint32_t n0 = sm.u.v.nml[0];
                bool diff = false;
                for (uint32_t u = lane; u < nu; u += 64) diff |= sm.u.v.nml[u] != n0;
                alleq = x.ballot(diff) == 0;
            }
            // ---- P7: the permutation std::sort applies (AQ.cpp:320-327)
            if (alleq) {
                const uint16_t* row = T.permtab + (size_t)nu * (nu - 1) / 2;
                for (uint32_t i = lane; i < nu; i += 64) sm.u.v.ord[i] = row[i];
            } else if (lane == 0) {
                gcc_sort_index(sm.u.v.ord, (int)nu, sm.u.v.nml, sm.stack);
            }
            x.sync();
            // ---- P8: vote (lane 0)
            if (lane == 0) {
                // PE_KMC counts are uint8_t in the reference: reduce them here
                for (uint32_t u = 0; u < nu; ++u) sm.dd[u] &= 0x00FF00FFu;
                const uint32_t ep = a.vote_epoch[x.bid()] + 1;
                HitMap hmap{sm.u.v.lkey, sm.lhit, 0, a.vote_scratch + (size_t)x.bid() * ((size_t)nloci + 1), ep, false};
                Asgn top;
                uint64_t nvvw = 0;
                vote(T, sm.u.v.ord, sm.uval, sm.dd, (int)nu, cth, hmap, top, nvvw);
                c_vv += nvvw;
                if (hmap.spilled) a.vote_epoch[x.bid()] = ep;
                sm.res[0] = (int32_t)(uint32_t)top.idx;
                sm.res[1] = (int32_t)top.fc;
                sm.res[2] = (int32_t)top.rc;
            }
            x.sync();
            dst0 = (uint32_t)sm.res[0];
            nm1 = sm.res[1]; nm2 = sm.res[2];
            {
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
                } else if (a.P.extract) {  // AQ.cpp:2094-2099
                    c_thr += 2; c_feas += 2;
                    stage = DBTK_STAGE_EXTRACT;
                } else {
                    c_thr += 2; c_feas += 2;
                    // ---- P10: assignTRkmc against the DBs of destLocus0 (AQ.cpp:2138-2144)
                    uint32_t ntr[2] = {0, 0};
                    for (int m = 0; m < 2; ++m) {
                        ms[m].rm = rm[m];
                        if (!okam && rm[m]) continue;
                        const uint32_t nk = m ? nk1 : nk0;
                        nas[m] = nk;
                        c_cls += nk;
                        uint32_t mytr = 0;
                        for (uint32_t i = lane; i < nk; i += 64) {
                            const uint64_t km = sm.kmer[m][i];
                            uint32_t c = CLS_NONE;
                            if (km != NAN64) c = cls_lookup(T, km, dst0);
                            const uint8_t s = (c == CLS_FLANK) ? 1 : (c == CLS_NONE ? 0 : 2);
                            sm.as[m][i] = s;
                            sm.hval[m][i] = c;
                            mytr += s == 2;
                        }
                        ntr[m] = x.wave_sum(mytr) & 0xFF;  // uint8_t ntr, AQ.cpp:1454
                    }
                    x.sync();
                    if (lane < 2 && nas[lane]) {
                        MateState r = ms[lane];
                        assign_scan(sm.as[lane], (int)nas[lane], ntr[lane], a.P, r);
                        int32_t* o = sm.mres[lane];
                        o[0] = r.si; o[1] = r.ei; o[2] = r.nt; o[3] = r.bs; o[4] = r.ti; o[5] = r.si_; o[6] = r.ei_; o[7] = r.af; o[8] = r.rm;
                    }
                    x.sync();
                    for (int m = 0; m < 2; ++m) {
                        if (!nas[m]) continue;
                        const int32_t* o = sm.mres[m];
                        ms[m] = MateState{o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8]};
                        af[m] = ms[m].af; rm[m] = ms[m].rm;
                    }
                    // ---- P11: accumulate (AQ.cpp:2145-2158)
                    if (rm[0] && rm[1]) { dst = nloci; stage = DBTK_STAGE_ASGN; }
                    else {
                        stage = DBTK_STAGE_COUNTED;
                        const int nmap = 2 - rm[0] - rm[1];
                        if (lane == 0) {
                            x.atomic_add(&a.nmapread[dst], (uint64_t)nmap);
                            x.atomic_add(&a.kmc[dst], (uint64_t)(int64_t)((ms[0].ei - ms[0].si) + (ms[1].ei - ms[1].si)));
                        }
                        c_asgn += (uint64_t)nmap;
                        for (int m = 0; m < 2; ++m) {
                            if (rm[m]) continue;
                            uint32_t myinc = 0;
                            for (uint32_t i = lane; i < nas[m]; i += 64)
                                if (sm.as[m][i] == 2) { x.atomic_add(&a.counts[sm.hval[m][i]], 1ull); ++myinc; }
                            c_inc += x.wave_sum(myinc);
                        }
                    }
                }
            }
        }
        // ---- P12: record (kam: AQ.cpp:2169-2175; trace: every pair)
        const bool want = a.recs && (a.P.trace || (okam && stage == DBTK_STAGE_COUNTED) ||
                                     (a.P.extract && stage == DBTK_STAGE_EXTRACT));
        if (want) {
            uint32_t at = pair;
            if (!a.P.trace) {
                if (lane == 0) at = x.atomic_add(a.nrec, 1u);
                at = x.bcast(at, 0);
            }
            if (at < a.rec_cap) {
                dbtk_pair_rec_t* r = &a.recs[at];
                if (lane == 0) {
                    r->pair = pair; r->stage = stage; r->dst = dst; r->dst0 = dst0; r->nm1 = nm1; r->nm2 = nm2;
                }
                if (lane < 2) {
                    dbtk_mate_rec_t* mr = lane ? &r->r2 : &r->r1;
                    const MateState& s = ms[lane];
                    mr->si = (int16_t)s.si; mr->ei = (int16_t)s.ei; mr->si_ = (int16_t)s.si_; mr->ei_ = (int16_t)s.ei_;
                    mr->nt = (int16_t)s.nt; mr->bs = (int16_t)s.bs; mr->ti = (int16_t)s.ti;
                    mr->kf = (uint8_t)kf[lane]; mr->hf = (uint8_t)hf[lane]; mr->bf = 0; mr->qf = 0;
                    mr->af = (uint8_t)af[lane]; mr->rm = (uint8_t)rm[lane];
                    mr->nk = (uint16_t)nas[lane];
                }
                for (int m = 0; m < 2; ++m) {
                    dbtk_mate_rec_t* mr = m ? &r->r2 : &r->r1;
                    if (lane < MAXL / 4) {
                        uint8_t b = 0;
                        for (int q = 0; q < 4; ++q) {
                            const uint32_t i = 4 * lane + q;
                            if (i < nas[m]) b |= (uint8_t)((sm.as[m][i] & 3) << (2 * q));
                        }
                        mr->as2[lane] = b;
                    }
                }
            }
        }
    }
    if (lane == 0) {
        if (c_kf) x.atomic_add(&a.counters[DBTK_C_KMERFILTERED], c_kf);
        if (c_hf) x.atomic_add(&a.counters[DBTK_C_LOCUSFILTERED], c_hf);
        if (c_qc) x.atomic_add(&a.counters[DBTK_C_QCFILTERED], c_qc);
        if (c_thr) x.atomic_add(&a.counters[DBTK_C_THREADING], c_thr);
        if (c_feas) x.atomic_add(&a.counters[DBTK_C_FEASIBLE], c_feas);
        if (c_asgn) x.atomic_add(&a.counters[DBTK_C_ASGN], c_asgn);
        if (c_vv) x.atomic_add(&a.counters[DBTK_C_ALGO_VV], c_vv);
        if (c_cls) x.atomic_add(&a.counters[DBTK_C_ALGO_CLS], c_cls);
        if (c_inc) x.atomic_add(&a.counters[DBTK_C_ALGO_INC], c_inc);
        if (c_nhash1) {
            x.atomic_add(&a.counters[DBTK_C_NHASH1], c_nhash1);
            x.atomic_add(&a.counters[DBTK_C_ALGO_PROBES], c_nhash1);
        }
    }
}

}  // namespace dbtk
#endif
