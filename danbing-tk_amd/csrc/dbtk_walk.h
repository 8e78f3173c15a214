// dbtk_walk.h — the graph-threading walk of the v1.3 contract as a wave64 kernel body.
//
// = isThreadFeasible and everything under it (src/aQueryFasta_thread.cpp:1114-1260: find_anchor :878-888,
// errorCorrection_forward :898-1089, errorCorrection_backward :1091-1106, thread_ext_t::get_edit :627-647,
// edit_kmers_forward :828-862, edit_kmers_backward :649-825, annot_gap :1108-1111) plus the call-site glue the
// reference keeps in comments (:2072-2088 both mates walked, pair kept if either is feasible; :2189-2194 "exact" counting).
//
// One wavefront walks one read; the walk is a sequential state machine in the reference, here it is laid out over
// the wave like this:
//   * the read's non-canonical k-mers (read2kmers keepN, AQ.h:246-271) are extracted by the lanes and ALL looked up
//     in the graph table up front (dbtk_tables.h: one probe per k-mer answers node / out-edges of both strands /
//     TR k-mer / counter), so the common step of the walk — "is k-mer ki a successor of k-mer ki-1" — is a bit test
//     on data already in LDS, and whole runs of matching k-mers are accepted with one ballot;
//   * what is left is handled one event at a time with wave-uniform control flow: N, homopolymer, re-anchoring
//     (find_anchor is a ballot over the next 64 positions), and error correction;
//   * errorCorrection_forward's eight edit classes are 62 independent hypotheses (4 + 16 + 4 + 16 + 1 + 4 + 1 + 16):
//     one per lane, all extended in lockstep, one graph probe per live lane and step; the lane numbering IS
//     get_edit's scan order, so the winner is a wave maximum of (score, 255 - lane); the reachability cube
//     (graph_triplet_t, AQ.cpp:865-875) is 20 probes in two rounds held as 4 + 16 four-bit masks;
//   * the array surgery of edit_kmers_forward / _backward (vector::insert / erase on k-mers, annotations and edit
//     operations) is done by all lanes as block shifts; only the edit-tract merging of edit_kmers_backward
//     (AQ.cpp:712-822), a few dozen dependent byte operations, runs on lane 0.
// The k-mer / annotation / edit arrays are fixed-capacity LDS arrays (DBTK_THREAD_CAP entries: every inserted k-mer
// costs at least MSC = 5 extended ones).
//
// Quirks of the reference that results depend on and that are kept: the unsigned wrap of `kmers[ki] - oldnt + nt0`
// when the k-mer is NAN64 (AQ.cpp:930), getNextNucs leaving the next-base set stale when the node is absent
// (AQ.cpp:547-557), `nkmers` frozen at the read's original size (AQ.cpp:1126, 1182), `++ncorrection` on top of the
// edits after a mid-read backward correction (AQ.cpp:1204), nskip / ncorrection wrapping as size_t.  Where the
// reference asserts (a successor named by an edge mask is missing, AQ.cpp:528-532) the walk ends with
// DBTK_THREAD_F_MISSING_NODE and ret = -1.
#ifndef DBTK_WALK_H_
#define DBTK_WALK_H_

#include "dbtk_tables.h"

namespace dbtk {

constexpr int WCAP = DBTK_THREAD_CAP;
constexpr uint32_t W_MSC = 5;  // min score for thread extension, AQ.cpp:1120
constexpr uint32_t W_CTR_REP = 256, W_CTR_STRIDE = 32;  // == CTR_REP / CTR_STRIDE of dbtk_kernels.h (counter replicas)

// ------------------------------------------------------------------ build --
struct GrBuildArgs {
    GrSlot* slots;
    uint64_t mask;
    uint32_t shift, ksize;
    const uint64_t* ks;       // graph pass: nodes (non-canonical); TR pass: the locus' TR k-mers (file order)
    const uint8_t* ms;        // graph pass: out-edge masks; nullptr = TR pass
    const uint64_t* beg;      // nloci + 1 prefix offsets into ks
    uint32_t nloci;
    const uint64_t* outslot;  // TR pass: OUT.trkmc.ar slot of entry i
    const uint32_t* trbeg;    // TR pass: first slot of each locus
    uint64_t n;
    uint64_t* nentries;       // += distinct (k-mer, locus) entries
};
template <class X>
DBTK_HD void body_gr_insert(X& x, const GrBuildArgs& a) {
    for (uint64_t i = (uint64_t)x.bid() * x.nthreads() + x.tid(); i < a.n; i += (uint64_t)x.nblocks() * x.nthreads()) {
        uint32_t lo = 0, hi = a.nloci;  // locus of entry i: beg[l] <= i < beg[l+1]
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (a.beg[mid] <= i) lo = mid; else hi = mid;
        }
        const uint32_t locus = lo;
        uint64_t canon;
        uint32_t bits;
        if (a.ms) {  // kmerDB[idx][kmer] |= c (readGraphKmers, AQ.h:571): repeated nodes OR
            const uint64_t node = a.ks[i], rc = revcomp2(node, a.ksize);
            const bool isf = node <= rc;
            canon = isf ? node : rc;
            const uint32_t b = (a.ms[i] & 0xFu) | GR_HAS;
            bits = isf ? b : b << GR_OPP;
        } else {
            canon = a.ks[i];
            bits = GR_TR | (((uint32_t)a.outslot[i] - a.trbeg[locus]) << GR_SLOT_SHIFT);
        }
        const uint64_t li = ((uint64_t)locus << 32) | bits;
        uint64_t s = hash_cls(canon, locus, a.shift);
        for (;;) {
            const uint64_t prev = x.atomic_cas(&a.slots[s].kmer, NAN64, canon);
            if (prev == NAN64 || prev == canon) {
                const uint64_t pl = x.atomic_cas(&a.slots[s].li, ~0ull, li);
                if (pl == ~0ull) { x.atomic_add(a.nentries, 1ull); break; }
                if ((uint32_t)(pl >> 32) == locus) { x.atomic_or(&a.slots[s].li, (uint64_t)bits); break; }
            }
            s = (s + 1) & a.mask;
        }
    }
}

// -------------------------------------------------------------- the walk --
struct WalkArgs {
    DevTables T;
    dbtk_params_t P;
    const uint8_t* seq;   // reads back to back; read r = [off[r], off[r+1])
    const uint64_t* off;
    // pair mode (the hot path, threading = 2): survivors [0, *nsurv) of the batch; walk_dst[t] = destLocus of survivor t
    // after countHit and the QC gate, NAN32 = the pair never reached threading
    const uint32_t* surv;
    const uint32_t* nsurv;
    uint32_t* walk_dst;      // in: destLocus; out: destLocus after threading (nloci when neither mate is feasible)
    uint32_t* walk_ret;      // out, per survivor: ret of mate 0 | ret of mate 1 << 8 (as int8)
    uint64_t* counts;        // OUT.trkmc.ar order
    uint64_t* counters;      // DBTK_C_*
    uint64_t* ctr_rep;       // nullptr or replicas (dbtk_kernels.h: counters_of)
    // function mode (tests, dbtk_thread_batch): read r walked against read_locus[r]
    const uint32_t* read_locus;
    uint32_t nreads;
    dbtk_thread_rec_t* trecs;  // function mode: one per read.  pair mode: nullptr, or two per survivor (trace / -a records)
    uint64_t* noncak;          // function mode: nullptr or nreads x MAXL uncorrected k-mers
    uint32_t* errflag;
    // -a / -ae (AQ.cpp:2232-2240): compact alignment records, one per emitted pair (layout: dbtk.h, dbtk_aln_hdr_t)
    uint8_t* aln;              // nullptr: none
    uint32_t aln_stride, aln_cap, aln_max;  // bytes per record, entries per packed array, records the buffer holds
    uint32_t* naln;            // slots handed out (in chunks of ALN_CHUNK per wave)
    // -a / -ae in TEXT form (params.aln & DBTK_ALN_TEXT): writeCigar / writeAnnot run on the device; a record = {u32 dst, u16 len, u16 0}
    // + "cigar2 \t annot2 \t cigar1 \t annot1" (len bytes, padded to 4), packed into an arena the waves carve TXT_CHUNK bytes at a
    // time; txt_idx[pair] = the record's byte offset (NAN32: the pair has none).  A few tens of bytes per pair instead of the
    // fixed-size arrays above: what the command line fetches.
    uint8_t* txt;              // nullptr: none
    uint32_t* txt_idx;         // [pairs of the batch], all NAN32 at launch
    uint32_t* ntxt;            // arena cursor (bytes handed out)
    uint32_t txt_cap;          // arena bytes
    uint64_t* dbg;             // diagnostic build only (-DDBTK_STAMPS): per-phase cycle sums
    // pair mode in two kernels (body_walk_fast, then body_walk_pairs on what it passed on): survivors t the fast kernel could
    // not decide.  body_walk_fast appends to it; body_walk_pairs takes its items from it (nullptr: every survivor).
    uint32_t* slow_list;       // entries: place in the list | WALK_HAS_INFO; WALK_NO_ENTRY: a reserved place nothing was put in
    uint32_t* nslow;
    // what the fast kernel already knows about a pair it passes on: the graph table's info word of every position of both mates
    // (row e of slow_info belongs to entry e of slow_list: [2][info_stride] words), so that body_walk_pairs need not look 260 k-mers up again
    uint32_t* slow_info;       // nullptr: none
    uint32_t info_cap;         // rows
    uint32_t info_stride;      // words per mate (32 * NPL of the fast kernel that ran)
    // body_walk_fast only: nullptr, or the places in the list of the pairs it is to take (the pairs of loci without a graph image:
    // what body_walk_fast_locus leaves) and their number
    const uint32_t* sel;
    const uint32_t* nsel;
    uint64_t* pstats;          // nullptr, or the context's path statistics (dbtk.h: dbtk_ctx_path_stats)
    uint32_t pend_locus;       // the lean kernel's locus-resident form leaves the pairs it cannot decide to body_walk_pairs_locus (marked in walk_ret) instead of the list
};
constexpr uint32_t WALK_HAS_INFO = 0x80000000u, WALK_NO_ENTRY = 0xFFFFFFFFu;
constexpr uint32_t WALK_PENDING = 0x80000000u;  // walk_ret of a pair the locus-resident lean kernel left to body_walk_pairs_locus (walk_ret is zeroed per batch)
constexpr int8_t WALK_NOT_EVALUATED = -2;  // walk_ret of a mate whose walk nothing needed: its pair was kept by the other mate (dbtk.h)
#ifdef DBTK_STAMPS
#define W_STAMP_DECL uint64_t wst[8] = {0}; uint64_t wlast = x.clock();
#define W_STAMP(i) do { const uint64_t now_ = x.clock(); wst[i] += now_ - wlast; wlast = now_; } while (0)
#define W_STAMP_FLUSH do { if (lane == 0 && a.dbg) for (int i_ = 0; i_ < 8; ++i_) if (wst[i_]) x.atomic_add(&a.dbg[i_], wst[i_]); } while (0)
// inside the out-of-line routines: sums kept in LDS (WalkSmem::dst_), lane 0 only
#define WS_BEGIN() do { if (x.lane() == 0) sm.dlast = x.clock(); } while (0)
#define WS(i) do { if (x.lane() == 0) { const uint64_t now_ = x.clock(); sm.dst_[i] += now_ - sm.dlast; sm.dlast = now_; } } while (0)
#else
#define WS_BEGIN() do { } while (0)
#define WS(i) do { } while (0)
#define W_STAMP_DECL
#define W_STAMP(i) do { } while (0)
#define W_STAMP_FLUSH do { } while (0)
#endif
constexpr uint32_t ALN_CHUNK = 16;
constexpr uint32_t TXT_CHUNK = 16384;  // arena bytes a wave takes at a time (one atomic per ~200 records)
constexpr uint64_t TXT_ARENA_MAX = 0xF0000000ull;  // largest arena: the 32-bit cursor keeps 256 MB of headroom past it (below)
constexpr int WTXT = 2 * (WCAP + 8);    // the longest text of one CIGAR / annotation: two characters per entry
DBTK_HD uint8_t aln_pack(uint8_t t, uint8_t g) {  // edit_t (t, g) in one byte: dbtk.h DBTK_ALN_*
    const uint32_t tc = t == '*' ? 0u : t == '=' ? 1u : t == 'X' ? 2u : t == 'D' ? 3u : t == 'I' ? 4u : 7u;
    const uint32_t gc = g == 0 ? 0u : g == 'A' ? 1u : g == 'C' ? 2u : g == 'G' ? 3u : g == 'T' ? 4u : 5u;
    return (uint8_t)(tc | (gc << 3));
}

// Lane 0 takes the next TXT_CHUNK bytes of the text arena.  A full arena is sticky: once the cursor has reached the capacity no wave
// adds to it any more (the cursor is read first), so it cannot wrap past 2^32 and hand out bytes that already hold records — between the
// read and the add at most one chunk per resident wave goes on top, far less than the headroom TXT_ARENA_MAX leaves.  The base
// returned for a full arena fails the caller's bounds test; the error word tells the host.
template <class X>
DBTK_HD uint32_t txt_carve(X& x, const WalkArgs& a) {
    if (x.atomic_or32(a.ntxt, 0u) >= a.txt_cap) {
        if (a.errflag) *a.errflag = DBTK_ERR_OVERFLOW;
        return a.txt_cap;
    }
    return x.atomic_add(a.ntxt, TXT_CHUNK);
}

struct WalkState {
    int ki, ni, nkm, nes, ntr;
    uint64_t nskip, ncorr;
    uint32_t flags;
};
// What the walk's out-of-line routines share with their callers goes through LDS, never through the stack: a local whose address is
// handed to a routine that is not inlined (a WalkState, an `int* out`, the kernel's argument struct behind `const DevTables&`) lives in
// scratch memory — a store and a load through L2 / HBM per access, in exactly the phases of the walk that wait for memory anyway
// (960 B of private segment per lane, 1.7 times as many write requests as read requests).
enum { ST_NI, ST_NES, ST_WID, ST_SCORE, ST_FLAGS, ST_KI, ST_NM, ST_ND, ST_NINS, ST_N };
struct WalkSmem {
    uint32_t raw[72];
    uint32_t pk[20];
    uint16_t vd[20];
    uint64_t km[WCAP + 8];    // the walked k-mers: non-canonical, NAN64 where the read has none
    uint16_t gi[WCAP + 8];    // graph info of km[i] as oriented: bits 0-3 out-edges, bit 4 is a node; bits 5-9 the same for its
                              // reverse complement; bit 10 its canonical form is a TR k-mer of the locus
    uint8_t tr[WCAP + 8];     // cg.tr
    uint8_t es_t[WCAP + 8], es_r[WCAP + 8], es_g[WCAP + 8];  // cg.es
    uint32_t slot[NKMAX];     // counter of the UNcorrected k-mer at each position of this mate (NAN32: not a TR k-mer)
    uint8_t bases[48];        // edit_kmers_*: bases to roll in / leading bases
    uint8_t cube[24];         // errorCorrection_forward: m1[4], m2[16]
    uint8_t scr[128];         // edit_kmers_backward: the read / graph bases of an edit tract
    uint32_t txl[2];          // lengths of the text form of the alignment: this mate's CIGAR and annotation, which are written OVER km[]
                              // (txc() / txa(): the walked k-mers are dead by then — walk_store, which wants them, runs first)
    DBTK_HD uint8_t* txc() { return reinterpret_cast<uint8_t*>(km); }
    DBTK_HD uint8_t* txa() { return reinterpret_cast<uint8_t*>(km) + WTXT; }
    DBTK_HD const uint8_t* txc() const { return reinterpret_cast<const uint8_t*>(km); }
    DBTK_HD const uint8_t* txa() const { return reinterpret_cast<const uint8_t*>(km) + WTXT; }
    int32_t st[ST_N + 1];     // the out-of-line routines' scalar results (ST_*), written by lane 0
    uint64_t st64[2];
    WalkState ws;             // the walk's state across a call of an out-of-line routine (ws_put / ws_get)
#ifdef DBTK_STAMPS
    uint64_t dst_[8], dlast;  // diagnostic: cycle sums of the slow walk's parts
#endif
};

DBTK_HD uint64_t w_roll(uint64_t kmer, uint64_t rmask, uint64_t b) { return ((kmer & rmask) << 2) + b; }
DBTK_HD uint8_t w_comp_char(uint8_t c) {  // baseComplement on a base letter, AQ.h:71-87
    return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 127;
}
DBTK_HD uint32_t w_code(uint8_t c) { return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : 3u; }

// graph info of a non-canonical k-mer at the locus, oriented (see WalkSmem::gi); *slot gets its counter or NAN32
DBTK_HD uint32_t w_info(const DevTables& T, uint32_t locus, uint64_t fw, uint32_t k, uint32_t* slot) {
    if (slot) *slot = NAN32;
    if (fw == NAN64) return 0;
    const uint64_t rc = revcomp2(fw, k);
    const bool isf = fw <= rc;
    const uint32_t info = gr_lookup(T, isf ? fw : rc, locus);
    if (slot && (info & GR_TR)) *slot = hbm_load32(&T.trbeg[locus]) + (info >> GR_SLOT_SHIFT);
    const uint32_t a = info & 0x1Fu, b = (info >> GR_OPP) & 0x1Fu;
    return (isf ? (a | (b << GR_OPP)) : (b | (a << GR_OPP))) | (info & GR_TR);
}

// Edit classes in get_edit's scan order (AQ.cpp:627-647) = lane numbering of the hypotheses:
//   0-3 1X(nt0)   4-7 1D(nt0)   8 1I   9 + 13 nt0 + 3 nt1 + {0: 2X, 1: X+D, 2: 2D}   9 + 13 nt0 + 12: X+I(nt0)   61: 2I
enum { H_1X, H_1D, H_1I, H_2X, H_XD, H_2D, H_XI, H_2I, H_NONE };
DBTK_HD void w_hyp(int id, int* type, uint32_t* nt0, uint32_t* nt1) {
    *nt0 = 0; *nt1 = 0;
    if (id < 4) { *type = H_1X; *nt0 = (uint32_t)id; }
    else if (id < 8) { *type = H_1D; *nt0 = (uint32_t)id - 4; }
    else if (id == 8) *type = H_1I;
    else if (id < 61) {
        const int r = id - 9, q = r % 13;
        *nt0 = (uint32_t)(r / 13);
        if (q == 12) *type = H_XI;
        else { *nt1 = (uint32_t)(q / 3); *type = q % 3 == 0 ? H_2X : q % 3 == 1 ? H_XD : H_2D; }
    } else if (id == 61) *type = H_2I;
    else *type = H_NONE;
}

// The tables and parameters as the out-of-line routines see them: a copy in LDS, one per workgroup (body_walk_*: made once at the start).
struct WalkConst {
    DevTables T;
    dbtk_params_t P;
};
template <class X>
DBTK_HD void ws_put(X& x, WalkSmem& sm, const WalkState& S) {
    x.sync();
    if (x.lane() == 0) sm.ws = S;
    x.sync();
}
template <class X>
DBTK_HD void ws_get(X& x, const WalkSmem& sm, WalkState& S) {
    x.sync();
    S.ki = (int)x.uni((uint32_t)sm.ws.ki); S.ni = (int)x.uni((uint32_t)sm.ws.ni); S.nkm = (int)x.uni((uint32_t)sm.ws.nkm);
    S.nes = (int)x.uni((uint32_t)sm.ws.nes); S.ntr = (int)x.uni((uint32_t)sm.ws.ntr); S.flags = x.uni(sm.ws.flags);
    const uint64_t a = sm.ws.nskip, b = sm.ws.ncorr;
    S.nskip = ((uint64_t)x.uni((uint32_t)(a >> 32)) << 32) | x.uni((uint32_t)a);
    S.ncorr = ((uint64_t)x.uni((uint32_t)(b >> 32)) << 32) | x.uni((uint32_t)b);
}

// errorCorrection_forward (AQ.cpp:898-1089) at index ki of the walk's k-mers, or — backward = true —
// errorCorrection_backward (AQ.cpp:1091-1106): the same on the reverse-complemented prefix before the anchor ki1.
// Returns skip; on success sm.st[ST_WID] = the winning hypothesis, sm.st[ST_SCORE] its score; sm.st[ST_FLAGS] = DBTK_THREAD_F_* raised.
template <class X>
DBTK_HD_NOINLINE bool walk_ec(X& x, WalkSmem& sm_, const DevTables& T_, uint32_t locus, bool backward, int ki, int nkm, uint32_t mes) {
    WalkSmem& sm = DBTK_LDS_REF(WalkSmem, sm_);
    const DevTables& T = DBTK_LDS_REF(const DevTables, T_);
    const int lane = x.lane();
    const uint32_t k = T.ksize;
    const uint64_t rmask = (1ull << 2 * (k - 1)) - 1;
    // V(j) = kmers[kiV + j] of the array the reference corrects on: forward the walk's own k-mers at ki + j;
    // backward kmers_rc[1 + j] = RC(kmers[ki - 1 - j]), kmers_rc[0] = RC(node)
    auto V = [&](int j) -> uint64_t {
        if (!backward) return sm.km[ki + j];
        const uint64_t v = sm.km[ki - 1 - j];
        return v == NAN64 ? NAN64 : revcomp2(v, k);
    };
    const int n = backward ? ki : nkm - ki;  // nkmers - ki of the corrected array
    const int ngood = (int)k + 2 < n ? (int)k + 2 : n;
    // the node whose successors seed the hypotheses, and its out-edges
    const uint32_t gS = backward ? ((uint32_t)sm.gi[ki] >> GR_OPP) & 0x1Fu : (uint32_t)sm.gi[ki - 1] & 0x1Fu;
    const uint64_t S = backward ? revcomp2(sm.km[ki], k) : sm.km[ki - 1];
    auto fail_missing = [&]() { x.sync(); if (lane == 0) { sm.st[ST_FLAGS] = (int32_t)DBTK_THREAD_F_MISSING_NODE; sm.st[ST_WID] = 0; sm.st[ST_SCORE] = 0; } x.sync(); return true; };
    if (!(gS & GR_HAS)) return fail_missing();  // getOutNodes asserts
    const uint32_t mask0 = gS & 0xFu;
    // reachability: m1[nt0] = out-edges of successor nt0, m2[nt0][nt1] = out-edges of its successor nt1
    x.sync();
    bool missing = false;
    {
        uint32_t m = 0;
        if (lane < 4 && ((mask0 >> lane) & 1)) {
            const uint32_t g = w_info(T, locus, w_roll(S, rmask, (uint64_t)lane), k, nullptr);
            if (!(g & GR_HAS)) missing = true;
            m = g & 0xFu;
        }
        if (lane < 4) sm.cube[lane] = (uint8_t)m;
    }
    x.sync();
    {
        uint32_t m = 0;
        if (lane >= 4 && lane < 20) {
            const uint32_t nt0 = (uint32_t)(lane - 4) >> 2, nt1 = (uint32_t)(lane - 4) & 3;
            if (((mask0 >> nt0) & 1) && ((sm.cube[nt0] >> nt1) & 1)) {
                const uint32_t g = w_info(T, locus, w_roll(w_roll(S, rmask, nt0), rmask, nt1), k, nullptr);
                if (!(g & GR_HAS)) missing = true;
                m = g & 0xFu;
            }
            sm.cube[lane] = (uint8_t)m;
        }
    }
    x.sync();
    if (x.ballot(missing)) return fail_missing();
    uint32_t nts1 = 0, nts2 = 0, nn1[4];  // nn1[nt0] = get_nnts(nt0): nt1 with some nt2 behind it
    for (uint32_t a = 0; a < 4; ++a) {
        nts1 |= sm.cube[a];
        nn1[a] = 0;
        for (uint32_t b = 0; b < 4; ++b) {
            const uint32_t m2 = sm.cube[4 + 4 * a + b];
            nts2 |= m2;
            if (m2) nn1[a] |= 1u << b;
        }
    }
    auto good = [&](int j) { return j < ngood && V(j) != NAN64; };
    const uint64_t v0 = V(0);
    const uint64_t oldnt = v0 % 4;
    const bool g0 = good(0), g1 = good(1), g2 = good(2);
    const uint32_t b0 = (uint32_t)(v0 % 4), b1 = g1 ? (uint32_t)(V(1) % 4) : 0, b2 = g2 ? (uint32_t)(V(2) % 4) : 0;
    const bool c1X = g1 && ((nts1 >> b1) & 1);
    const bool c2X = !c1X && g2 && mes >= 2 && ((nts2 >> b2) & 1);
    const bool cXI = g2 && mes >= 2 && ((nts1 >> b2) & 1);
    const bool cXD = g1 && mes >= 2 && ((nts2 >> b1) & 1);
    const bool c1I = g1 && ((mask0 >> b1) & 1);
    const bool c1D = g0 && ((nts1 >> b0) & 1);
    const bool c2I = g2 && mes >= 2 && ((mask0 >> b2) & 1);
    const bool c2D = g0 && mes >= 2 && ((nts2 >> b0) & 1);
    // this lane's hypothesis
    int type; uint32_t nt0, nt1;
    w_hyp(lane, &type, &nt0, &nt1);
    const uint64_t base0 = v0 - oldnt + nt0;  // corrected read kmer (wraps on NAN64 like the reference)
    const uint64_t prevk = backward ? revcomp2(sm.km[ki], k) : sm.km[ki - 1];  // kmers[ki - 1] of the corrected array
    const bool two = type == H_2X || type == H_XD || type == H_2D;
    const bool seeded = ((mask0 >> nt0) & 1) && (!two || ((nn1[nt0] >> nt1) & 1));
    bool alive = false;
    uint64_t cr = 0;
    uint32_t nn = 0;
    int j0 = 0, jl = 0;
    const int lim0 = (int)k < n ? (int)k : n, lim1 = (int)k + 1 < n ? (int)k + 1 : n, lim2 = (int)k + 2 < n ? (int)k + 2 : n;
    switch (type) {
        case H_1X: alive = c1X && seeded; cr = base0; nn = nn1[nt0]; j0 = 1; jl = lim1; break;
        case H_1D: alive = c1D && seeded; cr = base0; nn = nn1[nt0]; j0 = 0; jl = lim0; break;
        case H_1I: alive = c1I; cr = prevk; nn = mask0; j0 = 1; jl = lim1; break;
        case H_2X: alive = c2X && seeded; cr = w_roll(base0, rmask, nt1); nn = sm.cube[4 + 4 * nt0 + nt1]; j0 = 2; jl = lim2; break;
        case H_XD: alive = cXD && seeded; cr = w_roll(base0, rmask, nt1); nn = sm.cube[4 + 4 * nt0 + nt1]; j0 = 1; jl = lim1; break;
        case H_2D: alive = c2D && seeded; cr = w_roll(base0, rmask, nt1); nn = sm.cube[4 + 4 * nt0 + nt1]; j0 = 0; jl = lim0; break;
        case H_XI: alive = cXI && seeded; cr = base0; nn = nn1[nt0]; j0 = 2; jl = lim2; break;
        case H_2I: alive = c2I; cr = prevk; nn = mask0; j0 = 2; jl = lim2; break;
        default: break;
    }
    uint32_t cnt = 0;
    // The nodes a hypothesis visits do not depend on what the graph answers — only how far it gets does.  So: the first
    // W_PH1 steps of all hypotheses in lockstep (one probe per live lane and step: most hypotheses die there), then every
    // survivor's remaining steps AT ONCE — lane t rolls the survivor's k-mer t + 1 bases on and probes that node, and
    // the sequential rule "extend while the next base is in the current next-base set; an existing node replaces the set,
    // an absent one leaves it" (getNextNucs) becomes a last-present-mask scan over the lanes and one ballot.
    x.sync();
    if (lane < 48) { uint8_t b = 4; if (lane < lim2) { const uint64_t vj = V(lane); if (vj != NAN64) b = (uint8_t)(vj % 4); } sm.bases[lane] = b; }
    x.sync();
    constexpr int W_PH1 = 3;
    for (int j = 0; j < lim2 && j < W_PH1; ++j) {
        if (!x.ballot(alive && j < jl)) break;
        const uint32_t b = sm.bases[j];
        const bool act = alive && j >= j0 && j < jl;
        if (act) {
            if (b > 3) alive = false;  // good[j] is false
            else {
                cr = w_roll(cr, rmask, b);
                if ((nn >> b) & 1) {
                    ++cnt;
                    const uint32_t g = w_info(T, locus, cr, k, nullptr);  // getNextNucs: only an existing node replaces the set
                    if (g & GR_HAS) nn = g & 0xFu;
                } else alive = false;
            }
        }
    }
    for (;;) {
        const uint64_t sv = x.ballot(alive && W_PH1 < jl);
        if (!sv) break;
        const int L = (int)__builtin_ctzll(sv);
        const uint64_t crL = ((uint64_t)x.bcast((uint32_t)(cr >> 32), L) << 32) | x.bcast((uint32_t)cr, L);
        const uint32_t nnL = x.bcast(nn, L);
        const int jlL = (int)x.bcast((uint32_t)jl, L);
        const int j = W_PH1 + lane;  // this lane's step of the survivor
        bool valid = j < jlL;
        uint64_t node = crL;
        if (valid) {
            for (int q = W_PH1; q <= j; ++q) {
                const uint32_t b = sm.bases[q];
                if (b > 3) { valid = false; break; }  // a k-mer without all its bases ends the extension (good[] false)
                node = w_roll(node, rmask, b);
            }
        }
        uint32_t pm = 0;
        if (valid) { const uint32_t g = w_info(T, locus, node, k, nullptr); if (g & GR_HAS) pm = 0x10u | (g & 0xFu); }
        const uint32_t sc = x.wave_scan_lastnz(pm);  // the latest existing node's out-edges at or before this step
        uint32_t before = x.shfl_up1(sc);
        if (lane == 0) before = 0;
        const uint32_t nnb = before ? (before & 0xFu) : nnL;  // the next-base set this step is tested against
        const bool ok = valid && ((nnb >> (node % 4)) & 1);
        const uint64_t bad = x.ballot(!ok);
        const uint32_t adv = bad ? (uint32_t)__builtin_ctzll(bad) : 64u;
        if (lane == L) { cnt += adv; alive = false; }
    }
    // get_edit: the strictly best score in scan order; one edit needs >= MSC extended k-mers, two edits >= 2 MSC and mes > 1
    const bool twoed = type == H_2X || type == H_XD || type == H_2D || type == H_XI || type == H_2I;
    const bool valid = type != H_NONE && cnt >= (twoed ? 2 * W_MSC : W_MSC) && (!twoed || mes > 1);
    const uint32_t key = valid ? (cnt << 8) | (255u - (uint32_t)lane) : 0u;
    const uint32_t best = ~x.wave_min(~key);
    x.sync();
    if (lane == 0) { sm.st[ST_SCORE] = (int32_t)(best >> 8); sm.st[ST_WID] = (int32_t)(255u - (best & 255u)); sm.st[ST_FLAGS] = 0; }
    x.sync();
    return best == 0;
}

// the edits of a hypothesis: t[], g[] (graph base letters), returns how many
DBTK_HD uint8_t w_letter(uint32_t nt) { return (uint8_t)(0x54474341u >> (8 * nt)); }  // alphabet[nt]: 'A' 'C' 'G' 'T'
DBTK_HD int w_edits(int id, uint8_t t[2], uint8_t g[2]) {
    int type; uint32_t nt0, nt1;
    w_hyp(id, &type, &nt0, &nt1);
    t[0] = t[1] = 0; g[0] = g[1] = 0;
    switch (type) {
        case H_1X: t[0] = 'X'; g[0] = w_letter(nt0); return 1;
        case H_1D: t[0] = 'D'; g[0] = w_letter(nt0); return 1;
        case H_1I: t[0] = 'I'; return 1;
        case H_2X: t[0] = 'X'; g[0] = w_letter(nt0); t[1] = 'X'; g[1] = w_letter(nt1); return 2;
        case H_XD: t[0] = 'X'; g[0] = w_letter(nt0); t[1] = 'D'; g[1] = w_letter(nt1); return 2;
        case H_2D: t[0] = 'D'; g[0] = w_letter(nt0); t[1] = 'D'; g[1] = w_letter(nt1); return 2;
        case H_XI: t[0] = 'X'; g[0] = w_letter(nt0); t[1] = 'I'; return 2;
        case H_2I: t[0] = 'I'; t[1] = 'I'; return 2;
        default: return 0;
    }
}

// Block shift of the walk's arrays by the whole wave: elements [from, n) move by `d` (+: towards the end).
template <class X, class E>
DBTK_HD void w_shift(X& x, E* a, int from, int n, int d) {
    const int lane = x.lane();
    constexpr int R = (WCAP + 63) / 64;
    E v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { const int i = from + 64 * r + lane; if (i < n) v[r] = a[i]; }
    x.sync();
#pragma unroll
    for (int r = 0; r < R; ++r) { const int i = from + 64 * r + lane; if (i < n && i + d >= 0 && i + d < WCAP) a[i + d] = v[r]; }
    x.sync();
}

// graph info of km[lo, hi) again (they were rewritten)
template <class X>
DBTK_HD void walk_refresh(X& x, WalkSmem& sm, const DevTables& T, uint32_t locus, const WalkState& S, int lo, int hi) {
    const int lane = x.lane();
    const uint32_t k = T.ksize;
    if (lo < 0) lo = 0;
    if (hi > S.nkm) hi = S.nkm;
    x.sync();
    for (int i = lo + lane; i < hi; i += 64) sm.gi[i] = (uint16_t)w_info(T, locus, sm.km[i], k, nullptr);
    x.sync();
}

// find_anchor, AQ.cpp:878-888
template <class X>
DBTK_HD bool walk_find_anchor(X& x, WalkSmem& sm, uint32_t k, WalkState& S) {
    const int lane = x.lane();
    for (;;) {
        const int p = S.ki + lane;
        const bool is = p < S.nkm && (sm.gi[p] & GR_HAS);
        const uint64_t m = x.ballot(is);
        const int lim = S.nkm - S.ki < 64 ? S.nkm - S.ki : 64;
        const int adv = m ? __builtin_ctzll(m) : lim;
        S.nskip += (uint64_t)adv; S.ni += adv; S.ki += adv;
        if (m) break;
        if (S.ki >= S.nkm) return false;
    }
    x.sync();
    if (lane == 0) sm.tr[S.ki] = (sm.gi[S.ki] & GR_TR) ? '=' : '.';
    if (lane < (int)k && sm.es_t[S.ni + lane] == '*') sm.es_t[S.ni + lane] = '=';
    x.sync();
    return true;
}

// edit_kmers_forward, AQ.cpp:828-862
template <class X>
DBTK_HD_NOINLINE void walk_edit_forward(X& x, WalkSmem& sm_, const DevTables& T_, uint32_t locus, int wid, uint32_t score) {
    WalkSmem& sm = DBTK_LDS_REF(WalkSmem, sm_);
    const DevTables& T = DBTK_LDS_REF(const DevTables, T_);
    const int lane = x.lane();
    WalkState S;  // in: sm.ws; out: sm.ws
    ws_get(x, sm, S);
    const uint32_t k = T.ksize;
    const uint64_t rmask = (1ull << 2 * (k - 1)) - 1;
    auto refresh = [&](int lo, int hi) { walk_refresh(x, sm, T, locus, S, lo, hi); };
    uint8_t et[2], eg[2];
    const int ne = w_edits(wid, et, eg);
    int nm = 0, nd = 0, nins = 0;
#pragma unroll
    for (int e = 0; e < 2; ++e) if (e < ne) { nm += et[e] == 'X'; nd += et[e] == 'D'; nins += et[e] == 'I'; }  // (fixed bounds: et / eg stay in registers)
    const int ki0 = S.ki, dlen = nd - nins, dpos = nm + nd;
    const int n0 = S.nkm - ki0;
    // the bases to roll in after kmers[ki0 - 1]: the graph bases of the X / D edits, then the read's own bases from
    // old position ki0 + nm + nins on, while they are good and the rewritten k-mer stays inside min(size, ki + k)
    x.sync();
    if (lane < 40) {
        uint8_t b = 4;
        int q = 0;
#pragma unroll
        for (int e = 0; e < 2; ++e) if (e < ne && et[e] != 'I') { if (lane == q) b = (uint8_t)w_code(eg[e]); ++q; }
        if (lane >= dpos) {
            const int o = nm + nins + (lane - dpos);  // old offset from ki0
            if (o < n0 && sm.km[ki0 + o] != NAN64) b = (uint8_t)(sm.km[ki0 + o] % 4);
        }
        sm.bases[lane] = b;
    }
    x.sync();
    if (dlen) {
        const int from = ki0 + nm + nins;
        w_shift(x, sm.km, from, S.nkm, dlen);
        w_shift(x, sm.gi, from, S.nkm, dlen);
        S.nkm += dlen;
        if (S.nkm > WCAP) { S.flags |= DBTK_THREAD_F_OVERFLOW; S.nkm = WCAP; }
    }
    const int ki = ki0 + dpos;
    int nb = dpos;  // rewritten k-mers: the corrected ones, then the extended ones
    {
        const int lim = S.nkm < ki + (int)k ? S.nkm : ki + (int)k;
        for (int i = ki; i < lim; ++i) { if (sm.bases[nb] > 3) break; ++nb; }
    }
    {
        uint64_t v = sm.km[ki0 - 1];
        if (lane < nb) {
            for (int q = 0; q <= lane; ++q) v = w_roll(v, rmask, sm.bases[q]);
        }
        x.sync();
        if (lane < nb) sm.km[ki0 + lane] = v;
    }
    refresh(ki0, ki0 + nb);
    if (dlen) {  // cg.tr.resize(size + dlen, '*')
        const int nn = S.ntr + dlen;
        for (int i = S.ntr + lane; i < nn && i < WCAP; i += 64) sm.tr[i] = '*';
        S.ntr = nn > WCAP ? WCAP : nn;
    }
    if (nd) {  // cg.es.insert(begin + ni + k - 1 + nm, edit_t('D', 0, '*')) x nd
        const int at = S.ni + (int)k - 1 + nm;
        w_shift(x, sm.es_t, at, S.nes, nd);
        w_shift(x, sm.es_r, at, S.nes, nd);
        w_shift(x, sm.es_g, at, S.nes, nd);
        if (lane < nd) { sm.es_t[at + lane] = 'D'; sm.es_r[at + lane] = 0; sm.es_g[at + lane] = '*'; }
        S.nes += nd;
        if (S.nes > WCAP) { S.flags |= DBTK_THREAD_F_OVERFLOW; S.nes = WCAP; }
    }
    x.sync();
    for (int i = lane; i < dpos + (int)score; i += 64) sm.tr[ki0 + i] = (sm.gi[ki0 + i] & GR_TR) ? '=' : '.';
    if (lane < ne) { sm.es_t[S.ni + (int)k - 1 + lane] = lane == 0 ? et[0] : et[1]; sm.es_g[S.ni + (int)k - 1 + lane] = lane == 0 ? eg[0] : eg[1]; }
    for (int i = lane; i < (int)score; i += 64) sm.es_t[S.ni + ne + (int)k - 1 + i] = '=';
    x.sync();
    S.ni += ne + (int)score - 1;
    S.ki = ki + (int)score - 1;  // the last edited kmer
    S.ncorr += (uint64_t)ne;
    ws_put(x, sm, S);
}

// edit_kmers_backward, AQ.cpp:649-825, for the anchor at ki; the state comes and goes in sm.ws; the anchor's new place and the edits'
// nm / nd / ni are handed back in sm.st[ST_KI, ST_NM, ST_ND, ST_NINS]
template <class X>
DBTK_HD_NOINLINE void walk_edit_backward(X& x, WalkSmem& sm_, const DevTables& T_, uint32_t locus, int wid, uint32_t score, int ki) {
    WalkSmem& sm = DBTK_LDS_REF(WalkSmem, sm_);
    const DevTables& T = DBTK_LDS_REF(const DevTables, T_);
    const int lane = x.lane();
    WalkState S;
    ws_get(x, sm, S);
    const uint32_t k = T.ksize;
    auto refresh = [&](int lo, int hi) { walk_refresh(x, sm, T, locus, S, lo, hi); };
    uint8_t et[2], eg[2];
    const int ne = w_edits(wid, et, eg);
    int nm = 0, nd = 0, nins = 0;
#pragma unroll
    for (int e = 0; e < 2; ++e) if (e < ne) { nm += et[e] == 'X'; nd += et[e] == 'D'; nins += et[e] == 'I'; }  // (fixed bounds: et / eg stay in registers)
    const int dlen = nd - nins;
    S.ni += nd;
    if (dlen > 0) {  // kmers.insert(begin + ki, 0) / cg.tr.insert(begin + ki, '*'), dlen times
        w_shift(x, sm.km, ki, S.nkm, dlen);
        w_shift(x, sm.gi, ki, S.nkm, dlen);
        w_shift(x, sm.tr, ki, S.ntr, dlen);
        if (lane < dlen) { sm.km[ki + lane] = 0; sm.gi[ki + lane] = 0; sm.tr[ki + lane] = '*'; }
    } else if (dlen < 0) {  // erase [ki + dlen, ki)
        w_shift(x, sm.km, ki, S.nkm, dlen);
        w_shift(x, sm.gi, ki, S.nkm, dlen);
        w_shift(x, sm.tr, ki, S.ntr, dlen);
    }
    S.nkm += dlen; S.ntr += dlen;
    if (S.nkm > WCAP || S.ntr > WCAP) { S.flags |= DBTK_THREAD_F_OVERFLOW; if (S.nkm > WCAP) S.nkm = WCAP; if (S.ntr > WCAP) S.ntr = WCAP; }
    ki += dlen;
    x.sync();
    // corrected kmers (X / D edits, walking down from the anchor), then extended ones while the old k-mers were good:
    // kmers[i - 1] = (kmers[i] >> 2) + leading base, the base being the complement of the edit's graph base or
    // the old k-mer's own
    const int ncor = nm + nd;
    const int ki_ = ki - ncor;
    int next = 0;  // extended
    {
        int lo = ki_ - (int)k; if (lo < 0) lo = 0;
        for (int i = ki_; i > lo; --i) { if (sm.km[i - 1] == NAN64) break; ++next; }
    }
    if (lane < 40) {
        uint8_t b = 0;
        int q = 0;
#pragma unroll
        for (int e = 0; e < 2; ++e) if (e < ne && et[e] != 'I') { if (lane == q) b = (uint8_t)(3 - w_code(eg[e])); ++q; }
        if (lane >= ncor && lane < ncor + next) b = (uint8_t)(sm.km[ki_ - 1 - (lane - ncor)] >> (2 * (k - 1)));
        sm.bases[lane] = b;
    }
    x.sync();
    {
        uint64_t v = sm.km[ki];
        if (lane < ncor + next) {
            for (int q = 0; q <= lane; ++q) v = (v >> 2) + ((uint64_t)sm.bases[q] << (2 * (k - 1)));
        }
        x.sync();
        if (lane < ncor + next) sm.km[ki - 1 - lane] = v;
    }
    refresh(ki - ncor - next, ki);
    // the rest is sequential byte work on a few entries: lane 0
    if (lane == 0) {
        uint64_t nstar = 0;
        const int lb = ki - nm - nd - (int)score;
        for (int i = ki - 1; i >= lb; --i) {
            if (sm.tr[i] == '*') ++nstar;
            sm.tr[i] = (sm.gi[i] & GR_TR) ? '=' : '.';
        }
        nstar -= (uint64_t)(nm + nd);
        uint64_t nskip = S.nskip - nstar, ncorr = S.ncorr + (uint64_t)ne;
        int ni = S.ni, nes = S.nes;
        auto es_ins = [&](int at) {
            if (nes >= WCAP) return;
            for (int i = nes; i > at; --i) { sm.es_t[i] = sm.es_t[i - 1]; sm.es_r[i] = sm.es_r[i - 1]; sm.es_g[i] = sm.es_g[i - 1]; }
            sm.es_t[at] = 'D'; sm.es_r[at] = 0; sm.es_g[at] = '*';
            ++nes;
        };
        auto es_del = [&](int at) {
            for (int i = at; i + 1 < nes; ++i) { sm.es_t[i] = sm.es_t[i + 1]; sm.es_r[i] = sm.es_r[i + 1]; sm.es_g[i] = sm.es_g[i + 1]; }
            --nes;
        };
        auto t_at = [&](int i) -> uint8_t { return (i < 0 || i >= nes) ? (uint8_t)0 : sm.es_t[i]; };
        int ins_seen = 0;  // insertions already recorded in front of the entry: they hold a place in the edit string, none in the trace
        const int tr_pos = ki - dlen;
        for (int i = 0; i < tr_pos + ins_seen; ++i) if (sm.es_t[i] == 'I') ++ins_seen;
        int es_pos = tr_pos + ins_seen - 1;  // the entry of the edit string that belongs to this place of the trace
        for (int i = 0; i < ne; ++i, --es_pos) {  // the step's edits go into the edit string, newest first
            const uint8_t ti = i == 0 ? et[0] : et[1], gi_ = i == 0 ? eg[0] : eg[1];
            if (ti == 'D') { ++es_pos; es_ins(es_pos); }
            if (sm.es_t[es_pos] == 'D') {
                if (ti == 'I') { es_del(es_pos); --ni; }  // an insertion meeting a pending deletion: the two annihilate
                else sm.es_g[es_pos] = w_comp_char(gi_);
            } else {
                while (sm.es_t[es_pos] == 'I') --es_pos;
                sm.es_t[es_pos] = ti;
                sm.es_g[es_pos] = gi_ ? w_comp_char(gi_) : (uint8_t)0;
            }
        }
        int e0 = es_pos + 1, e1 = e0;
        for (uint32_t i = 0; i < score; ++i, --es_pos) {  // the positions the correction let the walk pass become matches
            const uint8_t c = sm.es_t[es_pos];
            if (c == '=') { }
            else if (c == '*') sm.es_t[es_pos] = '=';
            else break;
        }
        {   // widen [e0, e1) to the whole run of adjacent edits
            uint8_t c = t_at(e1);
            while (c == 'X' || c == 'D' || c == 'I') { ++e1; c = t_at(e1); }
            c = t_at(e0 - 1);
            while (c == 'X' || c == 'D' || c == 'I') { --e0; c = t_at(e0 - 1); }
        }
        // adjacent edits are rewritten in their shortest form: compare the read's and the graph's bases over the run
        int nets = e1 - e0, nr = 0, ng = 0;
        for (int i = e0; i < e1; ++i) { nr += sm.es_r[i] != 0; ng += sm.es_g[i] != 0; }
        auto rnt = [&](int q) -> uint8_t { for (int i = e0; i < e1; ++i) if (sm.es_r[i] && q-- == 0) return sm.es_r[i]; return 0; };
        auto gnt = [&](int q) -> uint8_t { for (int i = e0; i < e1; ++i) if (sm.es_g[i] && q-- == 0) return sm.es_g[i]; return 0; };
        if (nr == ng) {
            bool no_edit = true;
            for (int i = 0; i < nr; ++i) if (rnt(i) != gnt(i)) { no_edit = false; break; }
            if (no_edit) {  // same bases on both sides: the run was no edit at all
                int slid = 0;
                for (int i = e0; i < e1; ++i) {
                    if (sm.es_t[i + slid] == 'D') { es_del(i + slid); --slid; }
                    else { sm.es_t[i + slid] = '='; sm.es_g[i + slid] = 0; }
                }
                ni += slid;
                ncorr -= (uint64_t)(e1 - e0);
                nskip -= (uint64_t)(e1 - e0);
            } else if (nets != nr) {  // D + I (same position) -> X: the tract shrinks
                uint8_t* const rn = sm.scr; uint8_t* const gn = sm.scr + 64;  // (LDS: a private array indexed like this would be scratch memory)
                for (int i = 0; i < nr && i < 64; ++i) { rn[i] = rnt(i); gn[i] = gnt(i); }
                int slid = 0;
                const int shrink_by = nr - nets;
                int j = 0, kk = 0;
                for (int i = e0; i < e1; ++i) {
                    if (sm.es_t[i + slid] == 'D' && slid != shrink_by) { es_del(i + slid); --slid; }
                    else {
                        if (rn[kk & 63] == gn[kk & 63]) { sm.es_t[i + slid] = '='; sm.es_g[i + slid] = 0; }
                        else { sm.es_t[i + slid] = 'X'; sm.es_g[i + slid] = gn[j & 63]; }
                        ++j; ++kk;
                    }
                }
                ni += slid;
                ncorr += (uint64_t)(int64_t)slid;
                nskip += (uint64_t)(int64_t)slid;
            } else {  // match / mismatch only
                for (int i = 0; i < nr; ++i) {
                    if (sm.es_r[e0 + i] && sm.es_r[e0 + i] == sm.es_g[e0 + i]) {  // (here every entry has both bases) edit reverted
                        sm.es_t[e0 + i] = '='; sm.es_g[e0 + i] = 0;
                        --ncorr; --nskip;
                    }
                }
            }
        } else {
            for (int i = 0; i < nets; ++i) {
                if (sm.es_r[e0 + i] == sm.es_g[e0 + i]) { sm.es_t[e0 + i] = '='; sm.es_g[e0 + i] = 0; --ncorr; --nskip; }
            }
        }
        sm.st[0] = ni; sm.st[1] = nes;
        sm.st64[0] = nskip; sm.st64[1] = ncorr;
    }
    x.sync();
    S.ni = (int)x.uni((uint32_t)sm.st[0]); S.nes = (int)x.uni((uint32_t)sm.st[1]);
    {
        const uint64_t a = sm.st64[0], b = sm.st64[1];
        S.nskip = ((uint64_t)x.uni((uint32_t)(a >> 32)) << 32) | x.uni((uint32_t)a);
        S.ncorr = ((uint64_t)x.uni((uint32_t)(b >> 32)) << 32) | x.uni((uint32_t)b);
    }
    x.sync();
    if (S.nes > WCAP) { S.flags |= DBTK_THREAD_F_OVERFLOW; S.nes = WCAP; }
    if (lane == 0) { sm.st[ST_KI] = ki; sm.st[ST_NM] = nm; sm.st[ST_ND] = nd; sm.st[ST_NINS] = nins; }
    ws_put(x, sm, S);
}

// isThreadFeasible in full (the read's arrays in sm, the state in sm.ws): from its beginning (phase 0), or picking the main loop up at
// S.ki (phase 1) where walk_read's inlined common path met its first event.  Inlined since round 6: with the state, the tables and every
// result of the routines below it passing through LDS, nothing of the walk is on the stack any more and the kernel has no private
// segment at all (as a separate function it kept two callee-saved registers there: -DDBTK_WALK_SLOW_OUTLINE, tools/_variants).
#ifdef DBTK_WALK_SLOW_OUTLINE
#define DBTK_WALK_SLOW_FN DBTK_HD_NOINLINE
#else
#define DBTK_WALK_SLOW_FN DBTK_HD
#endif
template <class X>
DBTK_WALK_SLOW_FN int walk_slow(X& x, WalkSmem& sm_, const DevTables& T_, const dbtk_params_t& P_, uint32_t locus, int len, int phase) {
    WalkSmem& sm = DBTK_LDS_REF(WalkSmem, sm_);
    const DevTables& T = DBTK_LDS_REF(const DevTables, T_);
    const dbtk_params_t& P = DBTK_LDS_REF(const dbtk_params_t, P_);
    const int lane = x.lane();
    const uint32_t k = T.ksize;
    const uint64_t rmask = (1ull << 2 * (k - 1)) - 1;
    const bool correction = P.correction != 0;
    const uint64_t maxc = P.maxncorrection;
    const uint64_t nkmers = (uint64_t)(len - (int)k + 1);  // frozen (AQ.cpp:1126)
    const uint64_t maxnskip = nkmers >= P.thread_cth ? nkmers - P.thread_cth : 0;

    // The walk's state lives in registers (in: sm.ws, out: sm.ws).  The rare, large pieces (error correction, the k-mer / CIGAR surgery) are
    // separate functions so that the common path stays small; the state crosses those calls in sm.ws too (ws_put / ws_get): a reference to
    // S would pin it in scratch memory and turn every `S.ki += run` of the common path into a load and a store.
    WalkState S;
    ws_get(x, sm, S);
    WS_BEGIN();
    auto find_anchor = [&]() { WS(5); const bool r_ = walk_find_anchor(x, sm, k, S); WS(4); return r_; };
    auto edit_forward = [&](int wid, uint32_t score) { WS(5); ws_put(x, sm, S); walk_edit_forward(x, sm, T, locus, wid, score); ws_get(x, sm, S); WS(1); };
    // (anchor = -1: the walk's own position S.ki, which the routine moves; else an earlier anchor, handed back)
    auto edit_backward = [&](int wid, uint32_t score, int anchor, int& onm, int& ond, int& oni) -> int {
        WS(5);
        ws_put(x, sm, S);
        walk_edit_backward(x, sm, T, locus, wid, score, anchor < 0 ? S.ki : anchor);
        ws_get(x, sm, S);
        const int tki = (int)x.uni((uint32_t)sm.st[ST_KI]);
        onm = (int)x.uni((uint32_t)sm.st[ST_NM]); ond = (int)x.uni((uint32_t)sm.st[ST_ND]); oni = (int)x.uni((uint32_t)sm.st[ST_NINS]);
        WS(3);
        if (anchor < 0) S.ki = tki;
        return tki;
    };
    auto ec = [&](bool backward, int ki, uint32_t mes, int& wid, uint32_t& wscore) {
        WS(5);
        const bool skip = walk_ec(x, sm, T, locus, backward, ki, S.nkm, mes);
        WS(backward ? 2 : 0);
        wid = (int)x.uni((uint32_t)sm.st[ST_WID]); wscore = x.uni((uint32_t)sm.st[ST_SCORE]); S.flags |= x.uni((uint32_t)sm.st[ST_FLAGS]);
        return skip;
    };
    // every way out of the routine leaves the state in sm.ws
#define W_RETURN(v) do { const int r__ = (v); ws_put(x, sm, S); return r__; } while (0)

#define W_FAIL_CHECK() do { if (S.flags & (DBTK_THREAD_F_MISSING_NODE | DBTK_THREAD_F_OVERFLOW)) W_RETURN(-1); } while (0)
    if (phase == 0) {
    if (!find_anchor()) W_RETURN(0);
    if (S.ki > 0 && correction && S.ncorr < maxc && (uint32_t)S.ki >= W_MSC + 1) {  // leading unaligned kmers: backward first
        const uint32_t mes = (uint32_t)S.ki >= 2 * W_MSC + 2 ? 2 : 1;
        int wid; uint32_t score;
        const bool skip = ec(true, S.ki, mes, wid, score);
        W_FAIL_CHECK();
        if (!skip) { int a, b, c; (void)edit_backward(wid, score, -1, a, b, c); W_FAIL_CHECK(); }
    }
    ++S.ki; ++S.ni;
    }
    while (S.ki < S.nkm) {
        // runs of plain matches: k-mer p continues k-mer p - 1 and is one of its successors in the graph
        {
            const int p = S.ki + lane;
            bool ism = false;
            if (p < S.nkm) {
                const uint64_t kv = sm.km[p], pv = sm.km[p - 1];
                const uint32_t g = sm.gi[p - 1];
                ism = kv != NAN64 && pv != NAN64 && kv != pv && (g & GR_HAS) && ((g >> (kv % 4)) & 1) && kv == w_roll(pv, rmask, kv % 4);
            }
            const uint64_t m = x.ballot(ism);
            const int run = ~m ? __builtin_ctzll(~m) : 64;
            if (run) {
                if (lane < run) {
                    sm.tr[p] = (sm.gi[p] & GR_TR) ? '=' : '.';
                    sm.es_t[S.ni + (int)k - 1 + lane] = '=';
                }
                S.ki += run; S.ni += run;
                continue;
            }
        }
        x.sync();
        const uint64_t kv = sm.km[S.ki], pv = sm.km[S.ki - 1];
        if (kv == NAN64 || kv == pv) {  // "N" in read / homopolymer run
            if (lane == 0) { sm.tr[S.ki] = '*'; sm.es_t[S.ni + (int)k - 1] = '*'; }
            ++S.nskip;
            if (S.nskip > maxnskip) W_RETURN(0);
            ++S.ki; ++S.ni;
            continue;
        }
        if (pv == NAN64) {  // triggered after passing 'N'
            if (!find_anchor()) break;
            if (S.nskip > maxnskip) W_RETURN(0);
            ++S.ki; ++S.ni;
            continue;
        }
        if (!(sm.gi[S.ki - 1] & GR_HAS)) { S.flags |= DBTK_THREAD_F_MISSING_NODE; W_RETURN(-1); }  // getOutNodes(node) asserts
        // read kmer has no matching node in the graph, try error correction
        if ((uint64_t)S.ki + W_MSC >= nkmers) {  // not enough info
            S.nskip += nkmers - (uint64_t)S.ki;
            W_RETURN(S.nskip <= maxnskip ? (S.ncorr ? 2 : 1) : 0);
        }
        if (correction && S.ncorr < maxc) {
            uint32_t mes = (uint32_t)(S.nkm - S.ki) >= 2 * W_MSC + 2 ? 2 : 1;
            int wid; uint32_t score;
            bool skip = ec(false, S.ki, mes, wid, score);
            W_FAIL_CHECK();
            if (!skip) {  // passed forward correction
                uint8_t et[2], eg[2];
                S.nskip += (uint64_t)w_edits(wid, et, eg);
                if (S.nskip > maxnskip) W_RETURN(0);
                edit_forward(wid, score);
                W_FAIL_CHECK();
            } else {
                if (!find_anchor()) break;
                mes = 2;  // always have enough info to make 2 edits
                skip = ec(true, S.ki, mes, wid, score);
                W_FAIL_CHECK();
                if (!skip) {  // passed reverse correction
                    int nm, nd, nins;
                    (void)edit_backward(wid, score, -1, nm, nd, nins);
                    W_FAIL_CHECK();
                    ++S.ncorr;
                    uint64_t ki64 = (uint64_t)S.ki;
                    uint64_t reach = ki64 - (uint64_t)nm - (uint64_t)nd;
                    uint64_t gap = (k < reach ? k : reach) - score;
                    uint64_t ki0 = ki64, ki1 = ki64;
                    uint32_t sc = score;
                    while (!skip && gap) {  // forward and backward threads not fully patched
                        ki0 = ki1;
                        ki1 = ki0 - (uint64_t)nm - (uint64_t)nd - sc;
                        mes = ki1 >= 2 * W_MSC + 2 ? 2 : 1;
                        if (ki1 < W_MSC + 1) break;
                        if (!(sm.gi[ki1] & GR_HAS)) { S.flags |= DBTK_THREAD_F_MISSING_NODE; W_RETURN(-1); }  // assert(g.count(node_))
                        skip = ec(true, (int)ki1, mes, wid, sc);
                        W_FAIL_CHECK();
                        if (!skip) {
                            const int k1 = edit_backward(wid, sc, (int)ki1, nm, nd, nins);
                            W_FAIL_CHECK();
                            ki1 = (uint64_t)k1;
                            S.ki += nd - nins;
                            reach = ki1 - (uint64_t)nm - (uint64_t)nd;
                            gap = (k < reach ? k : reach) - sc;
                        }
                    }
                    if (gap) {  // annot_gap, AQ.cpp:1108-1111
                        x.sync();
                        for (uint64_t i = lane; i < gap; i += 64) { const int64_t q = (int64_t)ki1 - 1 - (int64_t)i; if (q >= 0 && q < WCAP) sm.tr[q] = '*'; }
                        x.sync();
                        S.nskip -= gap;
                    }
                    if (S.nskip > maxnskip) W_RETURN(0);
                }
                if (skip) {  // either initial or iterative backward correction failed
                    if (!find_anchor()) break;
                    if (S.nskip > maxnskip) W_RETURN(0);
                }
            }
        } else {
            if (!find_anchor()) break;
            if (S.nskip > maxnskip) W_RETURN(0);
        }
        ++S.ki; ++S.ni;
    }
    WS(5);
    W_RETURN((S.nskip <= maxnskip && S.ncorr <= maxc) ? (S.ncorr ? 2 : 1) : 0);
#undef W_FAIL_CHECK
#undef W_RETURN
}

// One read through isThreadFeasible.  The read's arrays must be in sm (walk_load).  Returns ret (wave-uniform).
// Inlined: the common path — the read's first k-mer is a node (find_anchor, AQ.cpp:878-888, stops at once) and every
// further k-mer continues its predecessor along an edge of the graph (AQ.cpp:1167-1180), 64 positions per ballot.  The
// first position that is anything else (N, homopolymer, no such edge, ...) hands the walk to walk_slow.
template <class X>
DBTK_HD int walk_read(X& x, WalkSmem& sm, const DevTables& T, const dbtk_params_t& P, uint32_t locus, int len, WalkState& S) {
    const int lane = x.lane();
    const uint32_t k = T.ksize;
    const uint64_t rmask = (1ull << 2 * (k - 1)) - 1;
    S.ki = 0; S.ni = 0; S.nkm = len - (int)k + 1; S.nes = len; S.ntr = S.nkm; S.nskip = 0; S.ncorr = 0; S.flags = 0;
    int phase = 0;
    if (sm.gi[0] & GR_HAS) {  // the anchor is the first k-mer
        if (lane == 0) sm.tr[0] = (sm.gi[0] & GR_TR) ? '=' : '.';
        if (lane < (int)k) sm.es_t[lane] = '=';  // (all '*' so far)
        x.sync();
        S.ki = 1; S.ni = 1;
        phase = 1;
        while (S.ki < S.nkm) {
            const int p = S.ki + lane;
            bool ism = false;
            if (p < S.nkm) {
                const uint64_t kv = sm.km[p], pv = sm.km[p - 1];
                const uint32_t g = sm.gi[p - 1];
                ism = kv != NAN64 && pv != NAN64 && kv != pv && (g & GR_HAS) && ((g >> (kv % 4)) & 1) && kv == w_roll(pv, rmask, kv % 4);
            }
            const uint64_t m = x.ballot(ism);
            const int run = ~m ? __builtin_ctzll(~m) : 64;
            if (!run) break;
            if (lane < run) {
                sm.tr[p] = (sm.gi[p] & GR_TR) ? '=' : '.';
                sm.es_t[S.ni + (int)k - 1 + lane] = '=';
            }
            S.ki += run; S.ni += run;
        }
        if (S.ki >= S.nkm) { x.sync(); return 1; }  // nothing skipped, nothing corrected: feasible (AQ.cpp:1259)
    }
    ws_put(x, sm, S);  // (through LDS: S itself stays in registers on the common path, and nothing of it is ever on the stack)
    const int ret = walk_slow(x, sm, T, P, locus, len, phase);
    ws_get(x, sm, S);
    return ret;
}

// ---- a read into the walk's LDS arrays, in stages so that the pair kernel can keep the memory round trips of the next read
// (its bytes) and of both mates (their graph look-ups) in flight together:
//   walk_raw_words   the read's bytes as 4-byte words from its 4-byte-aligned start, two per lane (loads only)
//   walk_stage       words -> LDS, 2-bit pack + validity, cg.init (AQ.cpp:62-67)
//   walk_probe_issue the read's non-canonical k-mers (read2kmers keepN, AQ.h:246-271) and the FIRST slot of every k-mer's
//                    probe sequence in the graph table (loads only: up to four per lane, all in flight)
//   walk_probe_finish  the look-ups resolved (a look-up rarely needs a second slot) -> k-mers, graph info, counters in LDS
constexpr int W_R = NKMAX / 64;
struct WalkProbe {  // what stays in registers while the look-ups are in flight: canonical k-mer and the first slot read
    uint64_t cn[W_R];
    GrSlot first[W_R];
};
DBTK_HD void walk_raw_words(const uint8_t* seq, uint64_t o0, uint32_t len, int lane, uint32_t w[2]) {
    const uint64_t a0 = o0 & ~3ull;
    const uint32_t nw = ((uint32_t)(o0 - a0) + len + 3) >> 2;
    w[0] = *reinterpret_cast<const uint32_t*>(seq + ((uint32_t)lane < nw ? a0 + 4ull * lane : a0));
    w[1] = *reinterpret_cast<const uint32_t*>(seq + (64u + lane < nw ? a0 + 4ull * (64 + lane) : a0));
}
template <class X>
DBTK_HD void walk_stage(X& x, WalkSmem& sm, const uint32_t w[2], uint64_t o0, uint32_t len) {
    const int lane = x.lane();
    const uint32_t rsh = (uint32_t)(o0 & 3), nw = (rsh + len + 3) >> 2;
    x.sync();
    if ((uint32_t)lane < nw) sm.raw[lane] = w[0];
    if (64u + lane < nw) sm.raw[64 + lane] = w[1];
    if (lane < 4) sm.raw[nw + lane] = 0;
    x.sync();
    if (lane < 16) {
        uint32_t q4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t v = 0;
            if (16u * lane + 4 * q < len) {
                const uint32_t B = rsh + 16 * lane + 4 * q, j = B >> 2, r8 = 8 * (B & 3);
                const uint32_t lo = sm.raw[j], hi = sm.raw[j + 1];
                v = r8 ? ((lo >> r8) | (hi << (32 - r8))) : lo;
                const uint32_t left = len - (16 * lane + 4 * q);
                if (left < 4) v &= (1u << (8 * left)) - 1;
            }
            q4[q] = v;
        }
        uint32_t vd;
        sm.pk[lane] = pack16(q4, &vd);
        sm.vd[lane] = (uint16_t)vd;  // bytes past the read are 0 -> invalid
        if (lane < 4) { sm.pk[16 + lane] = 0; sm.vd[16 + lane] = 0; }
    }
    for (int i = lane; i < (int)len; i += 64) {
        const uint32_t B = rsh + (uint32_t)i;
        sm.es_t[i] = '*';
        sm.es_r[i] = (uint8_t)(sm.raw[B >> 2] >> (8 * (B & 3)));
        sm.es_g[i] = 0;
    }
    x.sync();
}
template <class X>
DBTK_HD void walk_probe_issue(X& x, WalkSmem& sm, const DevTables& T, uint32_t locus, uint32_t len, WalkProbe& P, uint64_t* noncak) {
    const int lane = x.lane();
    const uint32_t k = T.ksize;
    const int nk = len >= k ? (int)(len - k + 1) : 0;
#pragma unroll
    for (int r = 0; r < W_R; ++r) {
        const int i = 64 * r + lane;
        uint64_t fw = NAN64, idx = 0;
        P.cn[r] = NAN64;
        if (i < nk) {
            uint64_t f;
            if (window_kmer(sm.pk, sm.vd, (uint32_t)i, k, &f, nullptr) != NAN64) fw = f;
            sm.km[i] = fw;
            sm.tr[i] = '*';
            if (noncak) noncak[i] = fw;
        }
        if (fw != NAN64) {
            const uint64_t rc = revcomp2(fw, k);
            P.cn[r] = fw <= rc ? fw : rc;
            idx = hash_cls(P.cn[r], locus, T.gr_shift);
        }
        if (!T.gimg) P.first[r] = gr_slot_load(&T.gr[idx]);  // (a position without a k-mer reads slot 0 and ignores it; with the locus' image in LDS: nothing to issue)
    }
}
template <class X>
DBTK_HD void walk_probe_finish(X& x, WalkSmem& sm, const DevTables& T, uint32_t locus, uint32_t len, const WalkProbe& P, uint32_t* slot) {
    const int lane = x.lane();
    const uint32_t k = T.ksize;
    const int nk = len >= k ? (int)(len - k + 1) : 0;
    const uint32_t trb = slot ? hbm_load32(&T.trbeg[locus]) : 0u;  // (once per read: inside the loop below it was a memory round trip per 64 positions)
#pragma unroll
    for (int r = 0; r < W_R; ++r) {
        const int i = 64 * r + lane;
        uint32_t info = 0;
        if (P.cn[r] != NAN64 && T.gimg) info = gimg_find(T.gimg, T.gimg_lgnb, P.cn[r]);
        else if (P.cn[r] != NAN64) {
            GrSlot sl = P.first[r];
            uint64_t j = hash_cls(P.cn[r], locus, T.gr_shift);
            for (;;) {
                if (sl.kmer == P.cn[r] && (uint32_t)(sl.li >> 32) == locus) { info = (uint32_t)sl.li; break; }
                if (sl.kmer == NAN64) break;
                j = (j + 1) & T.gr_mask;
                sl = gr_slot_load(&T.gr[j]);
            }
        }
        if (i < nk) {
            const bool isf = sm.km[i] == P.cn[r];  // the k-mer as read is its canonical form
            const uint32_t a = info & 0x1Fu, b = (info >> GR_OPP) & 0x1Fu;
            sm.gi[i] = (uint16_t)((isf ? (a | (b << GR_OPP)) : (b | (a << GR_OPP))) | (info & GR_TR));
        }
        if (slot) slot[i] = (info & GR_TR) ? trb + (info >> GR_SLOT_SHIFT) : NAN32;
    }
    x.sync();
}
// The same two stages for a pair the fast kernel passed on WITH what it had looked up (WalkArgs::slow_info): the k-mers as
// walk_probe_issue makes them and one coalesced row of info words instead of a graph-table probe sequence per position.
struct WalkInfoRow { uint32_t v[W_R]; };
template <class X>
DBTK_HD void walk_info_issue(X& x, WalkSmem& sm, const DevTables& T, uint32_t len, const uint32_t* row, uint32_t stride, WalkInfoRow& R) {
    const int lane = x.lane();
    const uint32_t k = T.ksize;
    const int nk = len >= k ? (int)(len - k + 1) : 0;
#pragma unroll
    for (int r = 0; r < W_R; ++r) {
        const int i = 64 * r + lane;
        if (i < nk) {
            uint64_t fw = NAN64, f;
            if (window_kmer(sm.pk, sm.vd, (uint32_t)i, k, &f, nullptr) != NAN64) fw = f;
            sm.km[i] = fw;
            sm.tr[i] = '*';
        }
        R.v[r] = row[(uint32_t)i < stride ? i : 0];
    }
}
template <class X>
DBTK_HD void walk_info_finish(X& x, WalkSmem& sm, const DevTables& T, uint32_t locus, uint32_t len, uint32_t stride, const WalkInfoRow& R, uint32_t* slot) {
    const int lane = x.lane();
    const uint32_t k = T.ksize;
    const int nk = len >= k ? (int)(len - k + 1) : 0;
    const uint32_t trb = slot ? hbm_load32(&T.trbeg[locus]) : 0u;  // (once per read)
#pragma unroll
    for (int r = 0; r < W_R; ++r) {
        const int i = 64 * r + lane;
        uint32_t info = 0;
        if (i < nk) {
            const uint64_t fw = sm.km[i];
            const bool has = fw != NAN64 && (uint32_t)i < stride;
            if (has) info = R.v[r];
            sm.gi[i] = (uint16_t)(info & 0x7FFu);  // (the row holds the info oriented as the read has the k-mer: wf_decide, dbtk_walkfast.h)
        }
        if (slot) slot[i] = (info & GR_TR) ? trb + (info >> GR_SLOT_SHIFT) : NAN32;
    }
    x.sync();
}
// All stages for one read (function mode).  Returns the read's length (clamped to the arrays, flagged).
template <class X>
DBTK_HD int walk_load(X& x, WalkSmem& sm, const DevTables& T, const uint8_t* seq, uint64_t o0, uint64_t o1, uint32_t locus,
                      uint32_t* slot, uint64_t* noncak, uint32_t* errflag) {
    uint32_t len = (uint32_t)(o1 - o0);
    if (len > (uint32_t)MAXL) { if (x.lane() == 0 && errflag) *errflag = DBTK_ERR_READ_TOO_LONG; len = MAXL; }
    uint32_t w[2];
    walk_raw_words(seq, o0, len, x.lane(), w);
    walk_stage(x, sm, w, o0, len);
    WalkProbe P;
    walk_probe_issue(x, sm, T, locus, len, P, noncak);
    walk_probe_finish(x, sm, T, locus, len, P, slot);
    return (int)len;
}

// What the walk left in LDS -> a thread record in HBM.
template <class X>
DBTK_HD_NOINLINE void walk_store(X& x, const WalkSmem& sm_, int ret, dbtk_thread_rec_t* o) {
    const WalkSmem& sm = DBTK_LDS_REF(const WalkSmem, sm_);
    const int lane = x.lane();
    WalkState S;  // (sm.ws: put there by the caller)
    ws_get(x, sm, S);
    x.sync();
    if (lane == 0) {
        o->ret = ret; o->ni = S.ni; o->nkm = (uint32_t)S.nkm; o->nes = (uint32_t)S.nes; o->ntr = (uint32_t)S.ntr; o->flags = S.flags;
    }
    for (int i = lane; i < WCAP; i += 64) {
        o->es_t[i] = i < S.nes ? sm.es_t[i] : (uint8_t)0;
        o->es_r[i] = i < S.nes ? sm.es_r[i] : (uint8_t)0;
        o->es_g[i] = i < S.nes ? sm.es_g[i] : (uint8_t)0;
        o->tr[i] = i < S.ntr ? sm.tr[i] : (uint8_t)0;
        o->kmers[i] = i < S.nkm ? sm.km[i] : 0ull;
    }
    x.sync();
}

// What the walk left in LDS -> mate m's half of a compact alignment record.
template <class X>
DBTK_HD_NOINLINE void walk_store_aln(X& x, const WalkSmem& sm_, int ret, uint8_t* rec, uint32_t cap, int m) {
    const WalkSmem& sm = DBTK_LDS_REF(const WalkSmem, sm_);
    const int lane = x.lane();
    WalkState S;
    ws_get(x, sm, S);
    dbtk_aln_hdr_t* h = reinterpret_cast<dbtk_aln_hdr_t*>(rec);
    uint8_t* es = rec + sizeof(dbtk_aln_hdr_t) + (size_t)(2 * m) * cap;
    uint8_t* tr = es + cap;
    x.sync();
    if (lane == 0) {
        if (m == 0) { h->ret1 = (int8_t)ret; h->nes1 = (uint16_t)S.nes; h->ntr1 = (uint16_t)S.ntr; }
        else { h->ret2 = (int8_t)ret; h->nes2 = (uint16_t)S.nes; h->ntr2 = (uint16_t)S.ntr; }
    }
    for (int i = lane; i < (int)cap; i += 64) {
        es[i] = i < S.nes ? aln_pack(sm.es_t[i], sm.es_g[i]) : (uint8_t)0;
        tr[i] = i < S.ntr ? sm.tr[i] : (uint8_t)0;
    }
    x.sync();
}

// the LDS of a wave of body_walk_pairs: one set of arrays per mate + the tables / parameters
struct WalkPairSmem {
    WalkSmem w[2];
    WalkConst c;
};
struct WalkReadSmem {  // ... of body_walk_reads
    WalkSmem w;
    WalkConst c;
};
template <class X>
DBTK_HD void walk_const_init(X& x, WalkConst& c, const WalkArgs& a) {
    if (x.lane() == 0) { c.T = a.T; c.P = a.P; }
    x.sync();
}
// Function mode: read r against read_locus[r] (one wave per read, reads at a fixed stride).
template <class X>
DBTK_HD void body_walk_reads(X& x, const WalkArgs& a) {
    WalkReadSmem& smb = *x.template smem<WalkReadSmem>();
    WalkSmem& sm = smb.w;
    walk_const_init(x, smb.c, a);
    const DevTables& T_ = smb.c.T;
    const uint32_t k = a.T.ksize;
    for (uint32_t r = x.bid(); r < a.nreads; r += x.nblocks()) {
        const uint64_t o0 = a.off[r], o1 = a.off[r + 1];
        const uint32_t locus = a.read_locus[r];
        WalkState S;
        int ret = -1;
        const int len = walk_load(x, sm, T_, a.seq, o0, o1, locus, nullptr, a.noncak ? a.noncak + (size_t)r * MAXL : nullptr, a.errflag);
        // the reference indexes kmers[0] of an empty vector when the read has no valid k-mer (the hot path never
        // hands such a read to the walk: both mates have one, AQ.cpp:2037)
        bool any = false;
        for (int i = x.lane(); i < len - (int)k + 1; i += 64) any |= sm.km[i] != NAN64;
        if (len >= (int)k && locus < a.T.nloci && x.ballot(any)) ret = walk_read(x, sm, T_, smb.c.P, locus, len, S);
        else { S.ki = 0; S.ni = 0; S.nkm = 0; S.nes = 0; S.ntr = 0; S.nskip = 0; S.ncorr = 0; S.flags = DBTK_THREAD_F_OVERFLOW; }
        if (S.flags & (DBTK_THREAD_F_MISSING_NODE | DBTK_THREAD_F_OVERFLOW)) ret = -1;
        ws_put(x, sm, S);
        walk_store(x, sm, ret, &a.trecs[r]);
    }
}

// writeCigar (src/aQueryFasta_thread.cpp:1683-1722) on cg.es as the walk left it (types t[], graph bases g[]): one lane, the
// same scan as dbtk_aln_format's (dbtk_rpgg.cpp) on the packed record.  Returns the text's length (<= 2 sz).
DBTK_HD uint32_t w_fmt_int(uint8_t* out, uint32_t n, int v) {
    if (v >= 100) out[n++] = (uint8_t)('0' + v / 100);
    if (v >= 10) out[n++] = (uint8_t)('0' + (v / 10) % 10);
    out[n++] = (uint8_t)('0' + v % 10);
    return n;
}
DBTK_HD uint32_t w_fmt_cigar(const uint8_t* t, const uint8_t* g, int sz, uint8_t* out) {
    uint32_t n = 0;
    if (!sz) { out[n++] = '*'; return n; }
    auto T = [&](int i) -> uint8_t { const uint8_t c = t[i]; return (c == '*' || c == '=' || c == 'X' || c == 'D' || c == 'I') ? c : (uint8_t)'?'; };
    auto G = [&](int i) -> uint8_t { const uint8_t c = g[i]; return c == 0 ? (uint8_t)0 : (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : (uint8_t)'*'; };
    int ct = 1;
    uint8_t t0 = T(0), g0 = G(0), t1 = 0, g1 = 0;
    for (int i = 1; i < sz; ++i) {
        t1 = T(i); g1 = G(i);
        if (t0 == '=' || t0 == '*') {
            while (t1 == t0) {
                ++ct; ++i;
                if (i == sz) break;
                t1 = T(i); g1 = G(i);
            }
            n = w_fmt_int(out, n, ct); out[n++] = t0;
        } else if (t0 == 'X') { out[n++] = 'X'; out[n++] = g0; }
        else if (t0 == 'D') {
            if (t1 == 'I') { out[n++] = 'X'; out[n++] = g0; ++i; }  // ins + del printed as a mismatch
            else { out[n++] = 'D'; out[n++] = g0; }
        } else if (t0 == 'I') {
            if (t1 == 'D') { out[n++] = 'X'; out[n++] = g1; ++i; }
            else out[n++] = 'I';
        } else out[n++] = t0;
        if (i == sz) return n;
        ct = 1;
        t0 = T(i); g0 = G(i);
    }
    n = w_fmt_int(out, n, ct); out[n++] = t0;
    return n;
}
// writeAnnot (AQ.cpp:1724-1740) on cg.tr
DBTK_HD uint32_t w_fmt_annot(const uint8_t* tr, int sz, uint8_t* out) {
    uint32_t n = 0;
    if (!sz) { out[n++] = '*'; return n; }
    int ct = 1;
    uint8_t c0 = tr[0];
    for (int i = 1; i < sz; ++i) {
        if (c0 == '=' || c0 == '.' || c0 == '*') {
            while (tr[i] == c0) { ++ct; ++i; if (i == sz) break; }
            n = w_fmt_int(out, n, ct); out[n++] = c0;
        } else out[n++] = c0;
        if (i == sz) return n;
        ct = 1;
        c0 = tr[i];
    }
    n = w_fmt_int(out, n, ct); out[n++] = c0;
    return n;
}
// The same two scans by the whole wave, 64 entries at a time (one lane per string took a hundred dependent LDS round trips per read
// and made the -a / -ae walk 2.6 times the plain one).  What a lane prints depends on its neighbours only through three things, all of
// them scans: the start of the run an '=' / '*' (annotation: '=' / '.' / '*') entry belongs to, the parity of a D / I entry's position in
// its maximal alternating D-I stretch (writeCigar pairs them greedily from the left: the odd ones are swallowed by the entry before
// them), and where in the text the lane's token goes.  A token is printed by the lane of its LAST entry.  writeCigar's quirk is kept:
// a token that starts at the very last entry is printed by the code behind its loop, as "1" + type, whatever the type.
template <class X>
DBTK_HD uint32_t wave_fmt_cigar(X& x, const uint8_t* t, const uint8_t* g, int sz, uint8_t* out) {
    const int lane = x.lane();
    if (!sz) { if (lane == 0) out[0] = '*'; return 1; }
    auto T = [&](int i) -> uint32_t { if (i < 0 || i >= sz) return 0u; const uint8_t c = t[i]; return (c == '*' || c == '=' || c == 'X' || c == 'D' || c == 'I') ? c : (uint32_t)'?'; };
    auto G = [&](int i) -> uint32_t { if (i < 0 || i >= sz) return 0u; const uint8_t c = g[i]; return c == 0 ? 0u : (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : (uint32_t)'*'; };
    uint32_t base = 0, run0 = 0, alt0 = 0;  // text written so far; (index + 1) of the last run start / alternating-stretch start before this chunk
    for (int c0 = 0; c0 < sz; c0 += 64) {
        const int i = c0 + lane;
        const bool in = i < sz;
        const uint32_t tp = T(i - 1), tc = T(i), tn = T(i + 1);
        const bool runny = tc == '=' || tc == '*', di = tc == 'D' || tc == 'I';
        const bool rstart = in && runny && tp != tc;
        const bool astart = in && di && !((tp == 'D' || tp == 'I') && tp != tc);
        const uint32_t rs = x.wave_scan_max(rstart ? (uint32_t)i + 1 : 0u), as = x.wave_scan_max(astart ? (uint32_t)i + 1 : 0u);
        const uint32_t rbeg = (rs ? rs : run0) - 1, abeg = (as ? as : alt0) - 1;  // (meaningful for runny / di entries: they always have a start)
        const bool odd = di && (((uint32_t)i - abeg) & 1u);
        const bool paired = di && !odd && (tn == 'D' || tn == 'I') && tn != tc;  // swallows the next entry
        const bool rend = runny && tn != tc;
        const bool lastsolo = in && i == sz - 1 && !odd && !(runny && !rstart);  // a token that starts at the last entry: "1" + type
        uint32_t n = 0, ct = 0;
        if (in && !odd) {
            if (lastsolo) n = 2;
            else if (runny) { if (rend) { ct = (uint32_t)i - rbeg + 1; n = (ct >= 100 ? 3u : ct >= 10 ? 2u : 1u) + 1; } }
            else if (tc == 'X' || tc == 'D') n = 2;
            else if (tc == 'I') n = paired ? 2 : 1;
            else n = 1;
        }
        const uint32_t o = base + x.wave_excl_scan(n);
        if (n) {
            if (lastsolo) { out[o] = '1'; out[o + 1] = (uint8_t)tc; }
            else if (runny) { uint32_t q = w_fmt_int(out, o, (int)ct); out[q] = (uint8_t)tc; }
            else if (tc == 'X') { out[o] = 'X'; out[o + 1] = (uint8_t)G(i); }
            else if (tc == 'D') { out[o] = paired ? 'X' : 'D'; out[o + 1] = (uint8_t)G(i); }
            else if (tc == 'I') { if (paired) { out[o] = 'X'; out[o + 1] = (uint8_t)G(i + 1); } else out[o] = 'I'; }
            else out[o] = (uint8_t)tc;
        }
        base += x.wave_sum(n);
        const uint32_t r63 = x.bcast(rs, 63), a63 = x.bcast(as, 63);
        if (r63) run0 = r63;
        if (a63) alt0 = a63;
    }
    return base;
}
template <class X>
DBTK_HD uint32_t wave_fmt_annot(X& x, const uint8_t* tr, int sz, uint8_t* out) {
    const int lane = x.lane();
    if (!sz) { if (lane == 0) out[0] = '*'; return 1; }
    auto C = [&](int i) -> uint32_t { return (i < 0 || i >= sz) ? 0x100u : (uint32_t)tr[i]; };
    uint32_t base = 0, run0 = 0;
    for (int c0 = 0; c0 < sz; c0 += 64) {
        const int i = c0 + lane;
        const bool in = i < sz;
        const uint32_t cp = C(i - 1), cc = C(i), cn = C(i + 1);
        const bool runny = cc == '=' || cc == '.' || cc == '*';
        const bool rstart = in && runny && cp != cc;
        const uint32_t rs = x.wave_scan_max(rstart ? (uint32_t)i + 1 : 0u);
        const uint32_t rbeg = (rs ? rs : run0) - 1;
        const bool lastsolo = in && i == sz - 1 && !(runny && !rstart);
        uint32_t n = 0, ct = 0;
        if (in) {
            if (lastsolo) n = 2;
            else if (runny) { if (cn != cc) { ct = (uint32_t)i - rbeg + 1; n = (ct >= 100 ? 3u : ct >= 10 ? 2u : 1u) + 1; } }
            else n = 1;
        }
        const uint32_t o = base + x.wave_excl_scan(n);
        if (n) {
            if (lastsolo) { out[o] = '1'; out[o + 1] = (uint8_t)cc; }
            else if (runny) { uint32_t q = w_fmt_int(out, o, (int)ct); out[q] = (uint8_t)cc; }
            else out[o] = (uint8_t)cc;
        }
        base += x.wave_sum(n);
        const uint32_t r63 = x.bcast(rs, 63);
        if (r63) run0 = r63;
    }
    return base;
}
// this mate's two strings into its LDS text buffers
template <class X>
DBTK_HD_NOINLINE void walk_format_text(X& x, WalkSmem& sm_, uint32_t cap) {
    WalkSmem& sm = DBTK_LDS_REF(WalkSmem, sm_);
    WalkState S;
    ws_get(x, sm, S);
    x.sync();
    const int nes = S.nes < (int)cap ? S.nes : (int)cap, ntr = S.ntr < (int)cap ? S.ntr : (int)cap;  // (as dbtk_aln_format clamps)
    static_assert(2 * WTXT <= (int)sizeof(sm.km), "the text buffers lie over the walked k-mers");
    const uint32_t lc = wave_fmt_cigar(x, sm.es_t, sm.es_g, nes, sm.txc());
    const uint32_t la = wave_fmt_annot(x, sm.tr, ntr, sm.txa());
    if (x.lane() == 0) { sm.txl[0] = lc; sm.txl[1] = la; }
    x.sync();
}

// Pair mode, the v1.3 call-site glue (AQ.cpp:2072-2088, 2090-2092, 2189-2194): both mates of an assigned pair are
// walked through graphDB[destLocus]; if either walk is feasible the pair is kept (nFeasibleReads += 2) and every
// uncorrected k-mer of both mates that is a TR k-mer of the locus is counted ("exact" mode: the canonical multiset of
// the reads' k-mers added to trKmers, i.e. one increment per position); else destLocus = nloci.
// One pair of the pair-mode walk: both mates staged, their graph look-ups (or the fast kernel's row of them), isThreadFeasible x 2 with the
// call site's short cut, records, exact counting, results.  Shared by body_walk_pairs (graph nodes from the global table, T = a.T) and
// body_walk_pairs_locus (T = a copy of a.T that names the locus' graph image in LDS).
struct WalkPairAcc {
    uint64_t c_feas = 0, c_inc = 0;
    uint32_t slot_base = 0, slot_used = ALN_CHUNK;  // alignment records: slots are taken ALN_CHUNK at a time (one atomic per chunk)
    uint32_t txt_base = 0, txt_left = 0;             // text records: arena bytes are taken TXT_CHUNK at a time
#ifdef DBTK_STAMPS
    uint64_t wst[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wlast = 0;
#endif
};
template <class X>
DBTK_HD void walk_pair(X& x, WalkSmem* smm, const WalkArgs& a, const WalkConst& C, uint32_t t, uint32_t dst, uint32_t pair, uint32_t inf,
                       const uint64_t (&oA)[3], const uint32_t (&wA)[2][2], WalkPairAcc& A) {
    const int lane = x.lane();
    const DevTables& T = C.T;  // (the copy in LDS: the walk's out-of-line routines take the tables and the parameters by reference)
#ifdef DBTK_STAMPS
    uint64_t (&wst)[8] = A.wst;
    uint64_t& wlast = A.wlast;
#endif
        int ret[2] = {0, 0};
        uint8_t* arec = nullptr;
        if (a.aln) {
            if (A.slot_used == ALN_CHUNK) {
                uint32_t b = 0;
                if (lane == 0) b = x.atomic_add(a.naln, ALN_CHUNK);
                A.slot_base = x.bcast(b, 0);
                A.slot_used = 0;
            }
            const uint32_t slot = A.slot_base + A.slot_used++;
            if (slot < a.aln_max) arec = a.aln + (size_t)slot * a.aln_stride;
            else if (lane == 0 && a.errflag) *a.errflag = DBTK_ERR_OVERFLOW;
        }
        uint32_t len[2];
        len[0] = (uint32_t)(oA[1] - oA[0]); len[1] = (uint32_t)(oA[2] - oA[1]);
        for (int m = 0; m < 2; ++m) if (len[m] > (uint32_t)MAXL) { if (lane == 0 && a.errflag) *a.errflag = DBTK_ERR_READ_TOO_LONG; len[m] = MAXL; }
        // both mates: bytes -> LDS -> k-mers, and the graph look-ups of both in flight together
        W_STAMP(0);  // prefetch issue + record slot
        walk_stage(x, smm[0], wA[0], oA[0], len[0]);
        walk_stage(x, smm[1], wA[1], oA[1], len[1]);
        W_STAMP(1);  // bytes (arrive) -> LDS, pack, cg.init
        if (inf != NAN32) {  // (uniform) the fast kernel looked every position up already: its row instead of the graph table
            WalkInfoRow R0, R1;
            const uint32_t* row = a.slow_info + (size_t)inf * 2 * a.info_stride;
            walk_info_issue(x, smm[0], T, len[0], row, a.info_stride, R0);
            walk_info_issue(x, smm[1], T, len[1], row + a.info_stride, a.info_stride, R1);
            W_STAMP(2);
            walk_info_finish(x, smm[0], T, dst, len[0], a.info_stride, R0, smm[0].slot);
            walk_info_finish(x, smm[1], T, dst, len[1], a.info_stride, R1, smm[1].slot);
        } else {
            WalkProbe P0, P1;
            walk_probe_issue(x, smm[0], T, dst, len[0], P0, nullptr);
            walk_probe_issue(x, smm[1], T, dst, len[1], P1, nullptr);
            W_STAMP(2);  // k-mers + first-slot loads issued
            walk_probe_finish(x, smm[0], T, dst, len[0], P0, smm[0].slot);
            walk_probe_finish(x, smm[1], T, dst, len[1], P1, smm[1].slot);
        }
        W_STAMP(3);  // look-ups resolved -> LDS
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            // The call site keeps a pair when EITHER mate is feasible and then counts the uncorrected k-mers of both (AQ.cpp:2082-2087,
            // 2189-2194): once mate 0 has threaded, nothing but a record (thread records, -a / -ae) needs mate 1's walk — the most
            // expensive thing this kernel does, and every pair on its list has two mates that did not thread cleanly.
            if (m == 1 && ret[0] && !a.trecs && !arec && !a.txt) { ret[1] = WALK_NOT_EVALUATED; W_STAMP(5); continue; }
            WalkState S;
            if (a.P.diag & 1) { ret[m] = 1; S.flags = 0; S.nes = S.ntr = S.nkm = 0; S.ni = 0; S.ki = 0; S.nskip = S.ncorr = 0; }  // diagnostic: no walk
            else ret[m] = walk_read(x, smm[m], T, C.P, dst, (int)len[m], S);
            if (S.flags & (DBTK_THREAD_F_MISSING_NODE | DBTK_THREAD_F_OVERFLOW)) { ret[m] = 0; if (lane == 0 && a.errflag) *a.errflag = DBTK_ERR_FORMAT; }
            if (a.trecs || arec || a.txt) ws_put(x, smm[m], S);  // (the record writers are out of line: the state goes to them through LDS)
            if (a.trecs) walk_store(x, smm[m], ret[m], &a.trecs[2 * (size_t)t + m]);
            if (arec) walk_store_aln(x, smm[m], ret[m], arec, a.aln_cap, m);
            if (a.txt) walk_format_text(x, smm[m], a.aln_cap);
            W_STAMP(4 + m);  // the walk of mate m
        }
        const bool alned = ret[0] || ret[1];
        if (arec && lane == 0) {  // -a: every walked pair; -ae: only the kept ones (AQ.cpp:2234)
            dbtk_aln_hdr_t* h = reinterpret_cast<dbtk_aln_hdr_t*>(arec);
            h->pair = (a.P.aln == 2 && !alned) ? NAN32 : pair;
            h->dst = alned ? dst : T.nloci;
            h->pad[0] = h->pad[1] = 0;
        }
        if (a.txt && (a.P.aln & 3) != 0 && ((a.P.aln & 3) == 1 || alned)) {  // -a: every walked pair; -ae: only the kept ones
            // "cigar2 \t annot2 \t cigar1 \t annot1" (writeAlignments' order, AQ.cpp:1751-1757) behind an 8-byte header
            const uint32_t lc1 = smm[0].txl[0], la1 = smm[0].txl[1], lc2 = smm[1].txl[0], la2 = smm[1].txl[1];
            const uint32_t len = lc2 + 1 + la2 + 1 + lc1 + 1 + la1, need = (8 + len + 3) & ~3u;
            if (need > A.txt_left) {  // (what is left of the old chunk stays unused)
                uint32_t b = 0;
                if (lane == 0) b = txt_carve(x, a);
                A.txt_base = x.bcast(b, 0);
                A.txt_left = TXT_CHUNK;
            }
            if ((uint64_t)A.txt_base + need <= a.txt_cap) {
                uint8_t* r = a.txt + A.txt_base;
                if (lane == 0) {
                    reinterpret_cast<uint32_t*>(r)[0] = alned ? dst : T.nloci;
                    reinterpret_cast<uint32_t*>(r)[1] = len;
                    a.txt_idx[pair] = A.txt_base;
                }
                for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
                    uint8_t c;
                    if (i < lc2) c = smm[1].txc()[i];
                    else if (i == lc2) c = '\t';
                    else if (i < lc2 + 1 + la2) c = smm[1].txa()[i - lc2 - 1];
                    else if (i == lc2 + 1 + la2) c = '\t';
                    else if (i < lc2 + la2 + 2 + lc1) c = smm[0].txc()[i - lc2 - la2 - 2];
                    else if (i == lc2 + la2 + 2 + lc1) c = '\t';
                    else c = smm[0].txa()[i - lc2 - la2 - lc1 - 3];
                    r[8 + i] = c;
                }
            } else if (lane == 0 && a.errflag) *a.errflag = DBTK_ERR_OVERFLOW;
            A.txt_base += need; A.txt_left -= need;
        }
        x.sync();
        if (alned) {
            A.c_feas += 2;
#pragma unroll
            for (int m = 0; m < 2; ++m)
                for (int i = lane; i < NKMAX; i += 64) {
                    const uint32_t s = smm[m].slot[i];
                    const bool hit = s != NAN32;
                    if (hit && !(a.P.diag & 2)) x.atomic_add(&a.counts[s], 1ull);  // (diagnostic 2: no count atomics)
                    A.c_inc += (uint64_t)__builtin_popcountll(x.ballot(hit));
                }
        }
        if (lane == 0) {
            a.walk_dst[t] = alned ? dst : T.nloci;
            a.walk_ret[t] = ((uint32_t)ret[0] & 0xFFu) | (((uint32_t)ret[1] & 0xFFu) << 8);
        }
        x.sync();
        W_STAMP(6);  // counting + results
}

template <class X>
DBTK_HD void body_walk_pairs(X& x, const WalkArgs& a) {
    WalkPairSmem& smb = *x.template smem<WalkPairSmem>();
    WalkSmem* const smm = smb.w;  // one set of arrays per mate
    walk_const_init(x, smb.c, a);
#ifdef DBTK_STAMPS
    if (x.lane() == 0) for (int m_ = 0; m_ < 2; ++m_) for (int i_ = 0; i_ < 8; ++i_) smm[m_].dst_[i_] = 0;
#endif
    const int lane = x.lane();
    uint64_t* const ctr = a.ctr_rep ? a.ctr_rep + (size_t)(x.bid() & (W_CTR_REP - 1)) * W_CTR_STRIDE : a.counters;
    const uint32_t nsurv = a.slow_list ? *a.nslow : *a.nsurv;  // items: the fast kernel's leftovers, or every survivor
    if (nsurv == 0) return;  // (the clamped prefetches below read entry 0 of the survivor list)
    const uint32_t S_ = x.nblocks();
    WalkPairAcc A;
    // A pair's data hangs on a chain of dependent loads: survivor -> (destLocus, pair index) -> the reads' offsets -> their bytes
    // -> their k-mers' graph look-ups.  What bounds this kernel is round trips per wave, so the chain is software-pipelined
    // over the wave's items (t, t + S, ...): while item i is walked, the bytes of item i + 1, the offsets of item i + 2 and
    // the (destLocus, pair) of item i + 3 are in flight; all of these loads are unconditional (clamped indices).
    const uint32_t ncap = *a.nsurv;  // (an entry of the list is a place below this, or WALK_NO_ENTRY)
    auto meta = [&](uint32_t i, uint32_t* dst, uint32_t* pair, uint32_t* tt, uint32_t* inf) {
        const uint32_t ic = i < nsurv ? i : 0u;
        const uint32_t e = a.slow_list ? a.slow_list[ic] : ic;  // (one more link of the chain in list mode: a dependent load)
        const bool none = a.slow_list && e == WALK_NO_ENTRY;
        const uint32_t tc0 = a.slow_list ? e & ~WALK_HAS_INFO : e, tc = !none && tc0 < ncap ? tc0 : 0u;
        const uint32_t d = a.walk_dst[tc];
        *pair = a.surv[tc];
        *dst = i < nsurv && !none ? d : NAN32;
        *tt = tc;
        *inf = a.slow_list && a.slow_info && !none && (e & WALK_HAS_INFO) && ic < a.info_cap ? ic : NAN32;
    };
    auto offs = [&](uint32_t pair, uint64_t o[3]) {
        o[0] = a.off[2 * (uint64_t)pair]; o[1] = a.off[2 * (uint64_t)pair + 1]; o[2] = a.off[2 * (uint64_t)pair + 2];
    };
    auto clampl = [&](uint64_t o0, uint64_t o1) { const uint64_t l = o1 - o0; return (uint32_t)(l > (uint64_t)MAXL ? (uint64_t)MAXL : l); };
    auto raws = [&](const uint64_t o[3], uint32_t w[2][2]) {
        walk_raw_words(a.seq, o[0], clampl(o[0], o[1]), lane, w[0]);
        walk_raw_words(a.seq, o[1], clampl(o[1], o[2]), lane, w[1]);
    };
    auto uni64 = [&](uint64_t v) { return ((uint64_t)x.uni((uint32_t)(v >> 32)) << 32) | x.uni((uint32_t)v); };
    uint32_t tA = x.bid();
    uint32_t dstA, pairA, dstB, pairB, dstC, pairC, dstD, pairD, ttA, ttB, ttC, ttD, infA, infB, infC, infD;
    uint64_t oA[3], oB[3], oC[3];
    uint32_t wA[2][2], wB[2][2];
    meta(tA, &dstA, &pairA, &ttA, &infA); meta(tA + S_, &dstB, &pairB, &ttB, &infB); meta(tA + 2 * S_, &dstC, &pairC, &ttC, &infC);
    dstA = x.uni(dstA); pairA = x.uni(pairA); dstB = x.uni(dstB); pairB = x.uni(pairB); ttA = x.uni(ttA); ttB = x.uni(ttB); infA = x.uni(infA); infB = x.uni(infB);
    offs(pairA, oA); offs(pairB, oB);
    for (int q = 0; q < 3; ++q) { oA[q] = uni64(oA[q]); oB[q] = uni64(oB[q]); }
    raws(oA, wA);
#ifdef DBTK_STAMPS
    uint64_t (&wst)[8] = A.wst;
    uint64_t& wlast = A.wlast;
    wlast = x.clock();
#endif
    for (; tA < nsurv; tA += S_) {
        W_STAMP(7);  // loop tail / pipeline rotation
        // the next items' loads, before anything of this item is waited for
        raws(oB, wB);
        dstC = x.uni(dstC); pairC = x.uni(pairC); ttC = x.uni(ttC); infC = x.uni(infC);
        offs(pairC, oC);
        meta(tA + 3 * S_, &dstD, &pairD, &ttD, &infD);
        const uint32_t t = ttA, dst = dstA, pair = pairA, inf = infA;
        if (dst != NAN32) walk_pair(x, smm, a, smb.c, t, dst, pair, inf, oA, wA, A);
        // the pipeline moves on
        dstA = dstB; pairA = pairB; ttA = ttB; infA = infB;
        for (int q = 0; q < 3; ++q) { oA[q] = oB[q]; oB[q] = uni64(oC[q]); }
        for (int m = 0; m < 2; ++m) { wA[m][0] = wB[m][0]; wA[m][1] = wB[m][1]; }
        dstB = dstC; pairB = pairC; ttB = ttC; infB = infC;
        dstC = dstD; pairC = pairD; ttC = ttD; infC = infD;
    }
    if (a.aln && lane == 0)  // the slots of the last chunk that were not used
        for (; A.slot_used < ALN_CHUNK; ++A.slot_used) {
            const uint32_t slot = A.slot_base + A.slot_used;
            if (slot < a.aln_max) reinterpret_cast<dbtk_aln_hdr_t*>(a.aln + (size_t)slot * a.aln_stride)->pair = NAN32;
        }
    W_STAMP_FLUSH;
#ifdef DBTK_STAMPS
    if (lane == 0 && a.dbg) for (int m_ = 0; m_ < 2; ++m_) for (int i_ = 0; i_ < 8; ++i_) if (smm[m_].dst_[i_]) x.atomic_add(&a.dbg[8 + i_], smm[m_].dst_[i_]);
#endif
    if (lane == 0) {
        if (A.c_feas) x.atomic_add(&ctr[DBTK_C_FEASIBLE], A.c_feas);
        if (A.c_inc) x.atomic_add(&ctr[DBTK_C_ALGO_INC], A.c_inc);
    }
}

}  // namespace dbtk
#endif
