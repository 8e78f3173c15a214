// dbtk_tables.h — HBM-resident tables of the RPGG and the per-k-mer primitives.
//
// Everything here is written once and compiled twice: by hipcc for gfx950
// (the product) and by the host compiler for tests/emu (a test-only SPMD
// emulator that runs the same kernel bodies on coroutine lanes, so the device
// logic can be checked against the oracle without a GPU).  There is no CPU
// execution path in the product library.
#ifndef DBTK_TABLES_H_
#define DBTK_TABLES_H_

#include <stdint.h>

#include "../../include/dbtk.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DBTK_HD __host__ __device__ __forceinline__
#ifdef DBTK_WALK_INLINE  /* diagnostic build: the walk's out-of-line routines inlined (tools/_variants) */
#define DBTK_HD_NOINLINE __host__ __device__ __forceinline__
#else
#define DBTK_HD_NOINLINE __host__ __device__ __noinline__
#endif
#else
#define DBTK_HD inline
#define DBTK_HD_NOINLINE inline
struct uint4 { uint32_t x, y, z, w; };  // host (tests/emu) stand-in for HIP's vector type
#endif

// Loads / stores that go to the device-coherent level (past the CU's own L1): for the few words one workgroup leaves for
// whichever workgroup comes next inside the same kernel (dbtk_kernels.h: the pooled vote-spill rows).
#if defined(__HIP_DEVICE_COMPILE__)
#define DBTK_COH_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define DBTK_COH_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define DBTK_COH_RELEASE() __atomic_thread_fence(__ATOMIC_RELEASE)
#else
#define DBTK_COH_LOAD(p) (*(p))
#define DBTK_COH_STORE(p, v) (*(p) = (v))
#define DBTK_COH_RELEASE() ((void)0)
#endif

namespace dbtk {

constexpr uint64_t NAN64 = DBTK_NAN64;
constexpr uint32_t NAN32 = DBTK_NAN32;
constexpr uint32_t NOHIT = 0xFFFFFFFFu;      // hval[] marker: position not in the index
constexpr uint32_t CLS_FLANK = 0xFFFFFFFFu;  // class-table value: flank k-mer of the locus
constexpr uint32_t CLS_NONE = 0xFFFFFFFEu;   // lookup result: (k-mer, locus) not in flank/TR sets
constexpr int MAXL = DBTK_MAX_READ_LEN;      // bases per read
constexpr int NKMAX = 256;                   // k-mer positions per mate (4 per lane)
constexpr int NHMAX = 2 * NKMAX;             // hits per pair

// ---- k-mer index: PREF.kmers.dbi (src/aQueryFasta_thread.h:654-673) as a
// bucketed open-addressed table in HBM.  A bucket is one 64-byte line: four
// keys, then the four values (SoA, so the encode kernel — which only needs
// "present or not" — reads 32 bytes).  A key lives in the first bucket, in probe
// order from its home bucket, that had a free slot; slots of a bucket fill in
// order 0..3, so key[3] == NAN64 <=> the bucket has a free slot, and bit 63 of
// key[3] (IDX_OVF; k <= 31 keeps canonical k-mers below 2^62) records that some
// insert walked past this full bucket.  A lookup therefore ends in its home
// bucket — one HBM line, one round trip — unless that bucket is full AND has
// overflowed.  `val` keeps the reference's encoding: even -> locus = val>>1,
// odd -> vv[val>>1] = n followed by n loci.  During the build `val` temporarily
// holds (file index << 32 | val) so that the LAST assignment of a duplicated key
// wins, as operator[] does.
struct __attribute__((aligned(64))) IdxBucket {
    uint64_t key[4];  // NAN64 = empty
    uint64_t val[4];  // low 32 bits: val.  High 32 bits (`aux`): for a k-mer unique to one locus (even val) its
                      // class at that locus — CLS_FLANK or its OUT.trkmc.ar slot — so that assignTRkmc needs no second probe
};
constexpr uint64_t IDX_OVF = 1ull << 63;

// ---- class table: for locus l, k-mer km: flank (PREF.fl.kdb) beats TR
// (PREF.tr.kmers) exactly as assignTRkmc tests them
// (src/aQueryFasta_thread.cpp:1466-1468).  lc = locus << 32 | cls, where cls is
// CLS_FLANK or the k-mer's slot in the OUT.trkmc.ar order.
struct ClsSlot {
    uint64_t kmer;  // NAN64 = empty
    uint64_t lc;    // ~0 = not yet written (build only)
};

// ---- the probe kernel's own copy of the index: two levels.
//
// Level 1 is grouped by MINIMIZER.  The ~130 k-mers of a read overlap, so consecutive ones share most of their m-mers: every
// k-mer is filed under its minimizer (the canonical m-mer with the smallest hash among its k - m + 1 m-mers; the same for a
// k-mer and its reverse complement), one 128-byte bucket per minimizer: 8 keys, then their 8 (val, aux) pairs.  The
// consecutive positions of a read that share a minimizer (a RUN: ~4 positions, ~33 runs per 150-bp read) find their keys in
// ONE bucket — one 128-byte line, fetched once per run by 8 lanes and searched in LDS.  A bucket that is full turns further
// keys away and says so (MZ_TURNED on its last key): those keys — the heavy minimizers of tandem repeats, whose hundreds of
// variant k-mers share a handful of m-mers — go to level 2.
//
// Level 2 (the overflow table) holds exactly the keys level 1 turned away, in 16-byte slots {key, val, aux}, open-addressed
// by a hash of the k-mer, linear probing, at most a quarter full: a look-up that does not find its key in a bucket marked
// MZ_TURNED goes there.  Two levels, never a chain.
//
// What this layout is built around (tools/pend.hip, MI355X): the chip's L1s pass on ~40-48 G requests per second however
// small they are, where a request is one 128-byte line asked for by one load INSTRUCTION — lanes of the same instruction
// that want the same line share a request, the same lane asking for the same line with its next instruction pays again,
// even while the first is still on its way.  So every line is asked for exactly once, by adjacent lanes of one instruction.
struct __attribute__((aligned(128))) MzBucket {
    uint64_t key[8];                       // canonical k-mers (< 2^62); MZ_EMPTY = free; slots fill in order; key[7] may carry MZ_TURNED
    struct { uint32_t val, aux; } pl[8];   // index value; class of a single-locus k-mer at its locus (IdxBucket::val's high word)
};
struct __attribute__((aligned(16))) MzSlot {  // level 2
    uint64_t key;  // MZ_EMPTY = free
    uint32_t val, aux;
};
constexpr uint64_t MZ_TURNED = 1ull << 63;
constexpr uint64_t MZ_EMPTY = 0x3FFFFFFFFFFFFFFFull;  // poly-T at k = 31, never canonical (poly-A is): matches no k-mer, carries no flag

struct DevTables;
struct LocusDir;
// ---- graph table (v1.3 threading): graphDB[locus] (GraphType = unordered_map<node, out-edge mask>,
// src/aQueryFasta_thread.h:32, loader :550-575) and trKmers[locus] folded into ONE open-addressed table keyed by
// (canonical k-mer, locus), so that a single probe answers everything the walk asks about a k-mer at a locus:
// is the k-mer (as given, non-canonical) a node and what are its out-edges, the same for its reverse complement
// (errorCorrection_backward walks the other strand, AQ.cpp:1091-1106), is its canonical form a TR k-mer
// (cg.tr annotation, AQ.cpp:886) and which OUT.trkmc.ar counter is it ("exact" counting, AQ.cpp:2189-2194).
// li = locus << 32 | info; info: bits 0-3 out-edge mask of the canonical form taken as a node (bit b: successor
// ((node & rmask) << 2) | b exists), bit 4 that node exists; bits 5-8 / bit 9 the same for the reverse-complement
// form; bit 10 TR k-mer of the locus; bits 11-31 its counter relative to the locus' first one.
struct GrSlot {
    uint64_t kmer;  // NAN64 = empty
    uint64_t li;    // ~0 = not yet written (build only)
};
constexpr uint32_t GR_HAS = 1u << 4, GR_OPP = 5, GR_TR = 1u << 10, GR_SLOT_SHIFT = 11;

struct DevTables {
    const IdxBucket* idx;
    uint64_t idx_mask;   // buckets - 1 (power of two)
    uint32_t idx_shift;  // 64 - log2(buckets)
    const uint32_t* vv;
    const ClsSlot* cls;
    uint64_t cls_mask;
    uint32_t cls_shift;
    const uint8_t* qc;        // nullptr or nloci bytes
    const uint16_t* permtab;  // introsort permutation of n equal keys, n = 1..NHMAX, row n at n(n-1)/2
    const uint64_t* flt;      // presence filter words (nullptr: none), 2^flt_logw of them
    uint32_t flt_logw;
    const uint32_t* trbeg;    // nloci + 1: first OUT.trkmc.ar slot of each locus (its TR k-mers' counters are contiguous)
    uint32_t nloci;
    uint32_t ksize;
    uint32_t consistent;  // index memberships == flank/TR sets (verified on the GPU at load): `aux` may be used
    // optional gates: (canonical (k+1)-mer, locus) set of PREF.tre.kdb for -bu; (k-mer, locus) -> (min << 8 | max) of
    // PREF.bt.kmdb for -b.  Same slot layout and probing as the class table.
    const ClsSlot* tre; uint64_t tre_mask; uint32_t tre_shift;
    const ClsSlot* bait; uint64_t bait_mask; uint32_t bait_shift;
    const GrSlot* gr; uint64_t gr_mask; uint32_t gr_shift;  // nullptr: no graph loaded
    const MzBucket* grmz; uint64_t grmz_mask;                 // its minimizer-grouped copy (dbtk_walkfast.h), nullptr: none
    const MzBucket* mz; uint64_t mz_mask; uint32_t mz_m;  // level 1 (mz_mask = buckets - 1 <= 2^28 - 1); nullptr: the probe kernel looks up the plain index
    const MzSlot* ovf; uint64_t ovf_mask;               // level 2 (ovf_mask = slots - 1 <= 2^32 - 1)
    const LocusDir* ldir; const uint8_t* limg;          // per-locus images of the index (dbtk_locus.h), nullptr: none
    const LocusDir* gldir; const uint8_t* glimg;        // ... and of the graph table (the lean walk kernel), nullptr: none
    // the graph image of ONE locus, resident in LDS (the buckets, behind the image's header): set by a kernel that works locus by locus
    // (dbtk_walk.h: body_walk_pairs_locus) in its own copy of this struct; gr_lookup then answers from it.  nullptr: the global table
    const uint32_t* gimg; uint32_t gimg_lgnb;
};

// ---- the per-locus images' hashes (dbtk_locus.h: an image is 2^lgnb buckets of 4 tags + 4 pay words, then a displacement byte per group)
// bucket before the displacement: the low bits of hi, mixed with everything else of the key (its low word AND the bits of hi above the
// bucket number, which the entry stores: k-mers that differ in their first bases only must not all want one bucket)
DBTK_HD uint32_t loc_base(uint32_t lo, uint32_t hi, uint32_t lgnb) { return hi ^ (((lo ^ ((hi >> lgnb) * 0x85EBCA6Bu)) * 0x9E3779B1u) >> 15); }
// a key's group: as many groups as buckets; a function of the low word and of the bits of hi the entry stores (not of the bits the bucket
// number implies): the k-mers of a tandem repeat that share their last 16 bases and differ in their first few must not all be one group
DBTK_HD uint32_t loc_group(uint32_t lo, uint32_t hi, uint32_t lgnb) { const uint32_t v = lo ^ ((hi >> lgnb) * 0x9E3779B1u); return ((v ^ (v >> 15)) * 0x85EBCA6Bu) >> (32 - lgnb); }
constexpr uint32_t GIMG_EMPTY = 0xFFFFFFFEu;  // pay of a free slot
// (canonical k-mer) -> info word of a GRAPH image (pay = extra << 24 | counter << 11 | the 11 flag bits); 0 when absent
DBTK_HD uint32_t gimg_find(const uint32_t* bks, uint32_t lgnb, uint64_t canon) {
    const uint32_t lo = (uint32_t)canon, hi = (uint32_t)(canon >> 32);
    const uint8_t* disp = reinterpret_cast<const uint8_t*>(bks + (8u << lgnb));
    const uint32_t* bk = bks + 8 * ((loc_base(lo, hi, lgnb) + disp[loc_group(lo, hi, lgnb)]) & ((1u << lgnb) - 1));
    const uint32_t extra = hi >> lgnb;
    uint32_t r = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const uint32_t p = bk[4 + s];
        if (bk[s] == lo && (p >> 24) == extra && p != GIMG_EMPTY) r = p & 0x00FFFFFFu;
    }
    return r;
}

DBTK_HD uint64_t hash_mix(uint64_t key) {
    key ^= key >> 29;
    return key * 0x9E3779B97F4A7C15ull;
}
DBTK_HD uint64_t hash_idx(uint64_t key, uint32_t shift) { return hash_mix(key) >> shift; }
// ---- presence filter in front of the index (encode stage only).  98 % of the sampled k-mers of WGS reads are not in
// the index, and each such probe costs a 64-byte line of HBM traffic next to the read stream it competes with.  A
// word-blocked Bloom filter of ~4 bits per key (128 MB at release scale) answers them: "no" is final, "maybe" (the keys
// themselves plus a few % of the others) goes on to the table, so the answer is exact.  Three bits of one 64-bit word per key.
// The filter is addressed through a mix of the 2k-bit k-mer onto 2k bits (odd multiplications modulo 2^2k and xor-shifts
// by k): the top bits pick the word, the low 18 bits its three bit positions.
constexpr uint64_t KMIX_C1 = 0x9E3779B97F4A7C15ull, KMIX_C2 = 0xD6E8FEB86659FD93ull;
DBTK_HD uint64_t kmix(uint64_t km, uint32_t k) {  // k <= 31
    const uint64_t M = (1ull << (2 * k)) - 1;
    uint64_t m = (km * KMIX_C1) & M;
    m ^= m >> k;
    m = (m * KMIX_C2) & M;
    return m ^ (m >> k);
}
// word of a mixed value in a filter of 2^logw words: its top bits
DBTK_HD uint64_t flt_word(uint64_t m, uint32_t k, uint32_t logw) { return logw <= 2 * k ? m >> (2 * k - logw) : m; }
DBTK_HD uint64_t flt_bits(uint64_t m) { return (1ull << (m & 63)) | (1ull << ((m >> 6) & 63)) | (1ull << ((m >> 12) & 63)); }
DBTK_HD uint64_t hash_cls(uint64_t kmer, uint32_t locus, uint32_t shift) {
    uint64_t x = kmer ^ ((uint64_t)locus * 0xD6E8FEB86659FD93ull);
    x ^= x >> 31;
    return (x * 0x9E3779B97F4A7C15ull) >> shift;
}

// The four keys of a bucket: two 16-byte loads.
DBTK_HD void bucket_keys(const IdxBucket* b, uint64_t k[4]) {
    const uint4 lo = reinterpret_cast<const uint4*>(b)[0], hi = reinterpret_cast<const uint4*>(b)[1];
    k[0] = ((uint64_t)lo.y << 32) | lo.x; k[1] = ((uint64_t)lo.w << 32) | lo.z;
    k[2] = ((uint64_t)hi.y << 32) | hi.x; k[3] = ((uint64_t)hi.w << 32) | hi.z;
}
// Where `key` stands in a bucket: 0..3 = its slot, BKT_MISS = not in the table, BKT_NEXT = look in the next bucket.
constexpr int BKT_MISS = 4, BKT_NEXT = 5;
DBTK_HD int bucket_find(const uint64_t k[4], uint64_t key) {
    int r = (k[3] != NAN64 && (k[3] & IDX_OVF)) ? BKT_NEXT : BKT_MISS;
    if (k[3] != NAN64 && (k[3] & ~IDX_OVF) == key) r = 3;
    if (k[2] == key) r = 2;
    if (k[1] == key) r = 1;
    if (k[0] == key) r = 0;
    return r;
}

// kmerDBi.find(kmer): returns val | aux << 32, low word NOHIT on a miss.
DBTK_HD uint64_t idx_lookup64_raw(const IdxBucket* idx, uint64_t idx_mask, uint32_t idx_shift, uint64_t key) {
    uint64_t b = hash_idx(key, idx_shift);
    for (;;) {
        uint64_t k[4];
        bucket_keys(&idx[b], k);
        const int r = bucket_find(k, key);
        if (r < 4) return idx[b].val[r];
        if (r == BKT_MISS) return (uint64_t)NOHIT;
        b = (b + 1) & idx_mask;
    }
}
DBTK_HD uint64_t idx_lookup64(const DevTables& T, uint64_t key) { return idx_lookup64_raw(T.idx, T.idx_mask, T.idx_shift, key); }
DBTK_HD uint32_t idx_lookup(const DevTables& T, uint64_t key) { return (uint32_t)idx_lookup64(T, key); }
// kmerDBi.count(kmer): the keys only
DBTK_HD bool idx_contains(const DevTables& T, uint64_t key) {
    uint64_t b = hash_idx(key, T.idx_shift);
    for (;;) {
        uint64_t k[4];
        bucket_keys(&T.idx[b], k);
        const int r = bucket_find(k, key);
        if (r < 4) return true;
        if (r == BKT_MISS) return false;
        b = (b + 1) & T.idx_mask;
    }
}
// flankDB[locus].count(km) / trKmers[locus].find(km) in one probe.
DBTK_HD uint32_t cls_lookup(const DevTables& T, uint64_t kmer, uint32_t locus) {
    uint64_t i = hash_cls(kmer, locus, T.cls_shift);
    for (;;) {
        const ClsSlot s = T.cls[i];
        if (s.kmer == kmer && (uint32_t)(s.lc >> 32) == locus) return (uint32_t)s.lc;
        if (s.kmer == NAN64) return CLS_NONE;
        i = (i + 1) & T.cls_mask;
    }
}

// generic (key, locus) probe of a ClsSlot table; CLS_NONE when absent
DBTK_HD uint32_t kl_lookup(const ClsSlot* tab, uint64_t mask, uint32_t shift, uint64_t key, uint32_t locus) {
    uint64_t i = hash_cls(key, locus, shift);
    for (;;) {
        const ClsSlot s = tab[i];
        if (s.kmer == key && (uint32_t)(s.lc >> 32) == locus) return (uint32_t)s.lc;
        if (s.kmer == NAN64) return CLS_NONE;
        i = (i + 1) & mask;
    }
}

// An object that is KNOWN to live in LDS, handed to a routine that is not inlined: the routine's parameter is a generic pointer, and when
// several kernels call it with different LDS objects nothing tells the compiler otherwise — every access becomes a FLAT instruction
// (slower than a ds_ one, and waited for with vmcnt(0) lgkmcnt(0) both).  Casting the reference to the LDS address space and back lets
// the address-space inference of the routine's own body see it: ds_read / ds_write again.  (Host build, tests/emu: the reference itself.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DBTK_NO_LDS_REF)
template <class TYPE> __device__ __forceinline__ TYPE& dbtk_lds_ref(TYPE& r) {
    __builtin_assume(__builtin_amdgcn_is_shared((const void*)&r));
    return r;
}
#define DBTK_LDS_REF(TYPE, ref) dbtk_lds_ref<TYPE>(ref)
#else
#define DBTK_LDS_REF(TYPE, ref) (ref)
#endif

// Loads through a pointer that is KNOWN to point into HBM although the compiler cannot see it (a DevTables copied into LDS hands its
// pointers out as generic ones, and a generic load is a FLAT instruction: it counts on the LDS and the vector-memory counters both, may
// return out of order with either kind, and so every wait for one is `s_waitcnt vmcnt(0) lgkmcnt(0)` — it drains whatever else is in
// flight, the prefetched next items included).  With the address space stated they are plain global loads.
#if defined(__HIP_DEVICE_COMPILE__)
typedef uint32_t dbtk_v4u_ __attribute__((vector_size(16)));
DBTK_HD GrSlot gr_slot_load(const GrSlot* p) {
    const dbtk_v4u_ v = *(const __attribute__((address_space(1))) dbtk_v4u_*)p;
    GrSlot s;
    s.kmer = (uint64_t)v[0] | ((uint64_t)v[1] << 32);
    s.li = (uint64_t)v[2] | ((uint64_t)v[3] << 32);
    return s;
}
DBTK_HD uint32_t hbm_load32(const uint32_t* p) { return *(const __attribute__((address_space(1))) uint32_t*)p; }
#else
DBTK_HD GrSlot gr_slot_load(const GrSlot* p) { return *p; }
DBTK_HD uint32_t hbm_load32(const uint32_t* p) { return *p; }
#endif

// (canonical k-mer, locus) -> info of the graph table; 0 when absent (a stored info is never 0)
DBTK_HD uint32_t gr_lookup(const DevTables& T, uint64_t canon, uint32_t locus) {
    if (T.gimg) return gimg_find(T.gimg, T.gimg_lgnb, canon);  // (the locus' own image, in LDS: the caller works on this locus only)
    uint64_t i = hash_cls(canon, locus, T.gr_shift);
    for (;;) {
        const GrSlot s = gr_slot_load(&T.gr[i]);
        if (s.kmer == canon && (uint32_t)(s.li >> 32) == locus) return (uint32_t)s.li;
        if (s.kmer == NAN64) return 0;
        i = (i + 1) & T.gr_mask;
    }
}

// ---- 2-bit packing (alphabet / baseNumConversion, src/aQueryFasta_thread.h:52-69:
// ONLY the bytes 'A','C','G','T' are bases).  A word holds 16 bases, the first
// base in the two most significant bits, so that a k-mer window is a funnel
// shift and compares like the reference's big-endian encoding
// (encodeSeq, src/aQueryFasta_thread.h:126-132).
DBTK_HD uint32_t pack4(uint32_t x, uint32_t* valid4) {
    // codes: ((c>>1) ^ (c>>2)) & 3 maps A,C,G,T -> 0,1,2,3
    const uint32_t t = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
    // expected byte for each code, compared with the input byte
    uint32_t e = 0;
    e |= (0x54474341u >> (8 * (t & 3))) & 0xFFu;
    e |= ((0x54474341u >> (8 * ((t >> 8) & 3))) & 0xFFu) << 8;
    e |= ((0x54474341u >> (8 * ((t >> 16) & 3))) & 0xFFu) << 16;
    e |= ((0x54474341u >> (8 * ((t >> 24) & 3))) & 0xFFu) << 24;
    const uint32_t d = x ^ e;
    // 0x80 in every byte of d that is zero
    const uint32_t z = ~(((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d | 0x7F7F7F7Fu);
    const uint32_t m = (z >> 7) & 0x01010101u;
    *valid4 = (m * 0x08040201u) >> 24 & 0xFu;  // byte 0 -> bit 3
    return (t * 0x40100401u) >> 24;            // byte 0 -> bits 7:6
}
// 16 ASCII bytes (little-endian words w[0..3]) -> 32-bit packed + 16 validity bits.
DBTK_HD uint32_t pack16(const uint32_t w[4], uint32_t* valid16) {
    uint32_t v0, v1, v2, v3;
    const uint32_t p0 = pack4(w[0], &v0), p1 = pack4(w[1], &v1), p2 = pack4(w[2], &v2), p3 = pack4(w[3], &v3);
    *valid16 = (v0 << 12) | (v1 << 8) | (v2 << 4) | v3;
    return (p0 << 24) | (p1 << 16) | (p2 << 8) | p3;
}

// ---- the same packing on the streaming path of the encode kernel, where instruction count is the bound:
// byte-select and 24-bit multiply instead of shifts/masks, and validity only as "any byte of the chunk is not
// ACGT" (nonzero *bad).  Exact validity bits are recomputed with pack16 for the rare tile that has such a byte.
DBTK_HD uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel) {  // v_perm_b32: selector byte 0..3 -> lo, 4..7 -> hi
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    const uint64_t src = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) r |= (uint32_t)((src >> (8 * ((sel >> (8 * i)) & 7))) & 0xFF) << (8 * i);
    return r;
#endif
}
// 4 ASCII bases -> their 2-bit codes gathered in BYTE 2 of the result (base 0 in bits 23:22); other bytes are scratch
DBTK_HD uint32_t pack4_b2(uint32_t x, uint32_t* bad) {
    const uint32_t t = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
    *bad |= x ^ byte_perm(0u, 0x54474341u, t);  // expected byte of each code vs the input byte
    // (t & 0xFFFFFF) * (2^22 + 2^12 + 2^2): codes 0,1,2 land at bits 23:22, 21:20, 19:18 (no carries); code 3 comes down from bits 25:24
    return ((t & 0x00FFFFFFu) * 0x00401004u) | (t >> 8);
}
DBTK_HD uint32_t pack16_fast(const uint32_t w[4], uint32_t* bad) {
    const uint32_t r0 = pack4_b2(w[0], bad), r1 = pack4_b2(w[1], bad), r2 = pack4_b2(w[2], bad), r3 = pack4_b2(w[3], bad);
    const uint32_t hi = byte_perm(r0, r1, 0x06020000u), lo = byte_perm(r2, r3, 0x00000602u);
    return byte_perm(hi, lo, 0x07060100u);
}

// Reverse complement of a k-mer held in the low 2k bits (getNuRC,
// src/aQueryFasta_thread.h:165-178): reverse the 2-bit symbols, complement.
DBTK_HD uint64_t revcomp2(uint64_t x, uint32_t k) {
    x = __builtin_bswap64(x);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    return (~x) >> (64 - 2 * k);
}

// Hash of a canonical m-mer (m <= 16) that orders the m-mers of a k-mer; the k-mer's minimizer is the smallest.  One
// multiplication (the probe kernel hashes every base of every read): the xor-shift in front spreads the high bases over the
// word, the one behind brings the product's well-mixed high bits down into the bits the bucket number is taken from.
DBTK_HD uint32_t mmer_hash2(uint32_t fw, uint32_t rc) {  // the two strands given
    uint32_t x = (fw < rc ? fw : rc) ^ 0x2545F491u;      // (poly-A must not hash to the smallest value)
    x ^= x >> 14; x *= 0x9E3779B1u; x ^= x >> 15;
    return x;
}
DBTK_HD uint32_t mmer_hash(uint64_t fw, uint32_t m) { return mmer_hash2((uint32_t)fw, (uint32_t)revcomp2(fw, m)); }
// Minimizer of a k-mer (either orientation: the hash is of the canonical m-mer) as the 28 bits its bucket is computed from.
DBTK_HD uint32_t mz_of_kmer(uint64_t kmer, uint32_t k, uint32_t m) {
    const uint64_t mm = (1ull << 2 * m) - 1;
    uint32_t best = 0xFFFFFFFFu;
    for (uint32_t i = 0; i + m <= k; ++i) {
        const uint32_t h = mmer_hash((kmer >> (2 * (k - m - i))) & mm, m);
        best = h < best ? h : best;
    }
    return best >> 4;
}
// Bucket of a minimizer.  The smallest of several hashes is a small number: its high bits are biased towards zero, its low
// bits are not, so the high bits are folded onto unbiased ones before the mask (no multiplication: once per position).
DBTK_HD uint32_t mz_bucket(uint32_t mz28, uint32_t mask) { return (mz28 ^ (mz28 << 9)) & mask; }
// Slot of a k-mer in the overflow table (before the mask).
DBTK_HD uint32_t ovf_hash(uint64_t kmer) {
    uint32_t h = (uint32_t)kmer ^ ((uint32_t)(kmer >> 32) * 0x9E3779B1u);
    h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13;
    return h;
}
// The minimizer length used for k, and whether the probe kernel's fast form (dbtk_probe2.h) exists for it: it is
// instantiated for windows of k - m + 1 = 7 (k = 19 .. 22) and 11 (k = 23 .. 26) m-mers.  0: no level-1 / level-2 tables.
DBTK_HD uint32_t mz_m_for_k(uint32_t k) { return (k >= 19 && k <= 22) ? k - 6 : (k >= 23 && k <= 26) ? k - 10 : 0u; }

// Window of k bases starting at base `b` of a packed stream (pk: 16 bases per
// word, vd: 16 validity bits per entry, both big-endian; two zero entries of
// padding must follow the data).  Returns the canonical k-mer
// (min(fw, rc), src/aQueryFasta_thread.h:288) or NAN64 when any base of the
// window is not ACGT.  *fw_out / *rc_out get the two strands (for edges).
DBTK_HD uint64_t window_kmer(const uint32_t* pk, const uint16_t* vd, uint32_t b, uint32_t k, uint64_t* fw_out,
                             uint64_t* rc_out) {
    const uint32_t w = b >> 4, o = b & 15;
    const uint64_t hi = ((uint64_t)pk[w] << 32) | pk[w + 1];
    const uint64_t lo = (uint64_t)pk[w + 2] << 32;
    const uint64_t x = o ? ((hi << (2 * o)) | (lo >> (64 - 2 * o))) : hi;
    const uint64_t fw = x >> (64 - 2 * k);
    const uint64_t v48 = ((uint64_t)vd[w] << 32) | ((uint64_t)vd[w + 1] << 16) | vd[w + 2];
    const uint64_t vk = (v48 << (16 + o)) >> (64 - k);  // the k validity bits of the window
    const uint64_t rc = revcomp2(fw, k);
    if (fw_out) *fw_out = fw;
    if (rc_out) *rc_out = rc;
    if (vk != ((1ull << k) - 1)) return NAN64;
    return fw < rc ? fw : rc;
}

// Forward strand only of the same window (every base known to be valid).
DBTK_HD uint64_t window_fw_clean(const uint32_t* pk, uint32_t b, uint32_t k) {
    const uint32_t w = b >> 4, o = b & 15;
    const uint64_t hi = ((uint64_t)pk[w] << 32) | pk[w + 1];
    const uint64_t lo = (uint64_t)pk[w + 2] << 32;
    const uint64_t x = o ? ((hi << (2 * o)) | (lo >> (64 - 2 * o))) : hi;
    return x >> (64 - 2 * k);
}

// The same window when every base of the stream is known to be valid (no validity words are read).
DBTK_HD uint64_t window_kmer_clean(const uint32_t* pk, uint32_t b, uint32_t k) {
    const uint32_t w = b >> 4, o = b & 15;
    const uint64_t hi = ((uint64_t)pk[w] << 32) | pk[w + 1];
    const uint64_t lo = (uint64_t)pk[w + 2] << 32;
    const uint64_t x = o ? ((hi << (2 * o)) | (lo >> (64 - 2 * o))) : hi;
    const uint64_t fw = x >> (64 - 2 * k);
    const uint64_t rc = revcomp2(fw, k);
    return fw < rc ? fw : rc;
}

}  // namespace dbtk
#endif
