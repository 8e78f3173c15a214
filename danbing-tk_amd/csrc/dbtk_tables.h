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
#define DBTK_HD_NOINLINE __host__ __device__ __noinline__
#else
#define DBTK_HD inline
#define DBTK_HD_NOINLINE
struct uint4 { uint32_t x, y, z, w; };  // host (tests/emu) stand-in for HIP's vector type
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

// ---- k-mer index: PREF.kmers.dbi (src/aQueryFasta_thread.h:654-673) as an
// open-addressed, linear-probed table of 16-byte slots.  `val` keeps the
// reference's encoding: even -> locus = val>>1, odd -> vv[val>>1] = n followed
// by n loci.  During the build `val` temporarily holds (file index << 32 | val)
// so that the LAST assignment of a duplicated key wins, as operator[] does.
struct IdxSlot {
    uint64_t key;  // NAN64 = empty
    uint64_t val;  // low 32 bits: val.  High 32 bits (`aux`): for a k-mer unique to one locus (even val) its
                   // class at that locus — CLS_FLANK or its OUT.trkmc.ar slot — so that assignTRkmc needs no second probe
};

// ---- class table: for locus l, k-mer km: flank (PREF.fl.kdb) beats TR
// (PREF.tr.kmers) exactly as assignTRkmc tests them
// (src/aQueryFasta_thread.cpp:1466-1468).  lc = locus << 32 | cls, where cls is
// CLS_FLANK or the k-mer's slot in the OUT.trkmc.ar order.
struct ClsSlot {
    uint64_t kmer;  // NAN64 = empty
    uint64_t lc;    // ~0 = not yet written (build only)
};

struct DevTables {
    const IdxSlot* idx;
    uint64_t idx_mask;   // capacity - 1 (power of two)
    uint32_t idx_shift;  // 64 - log2(capacity)
    const uint32_t* vv;
    const ClsSlot* cls;
    uint64_t cls_mask;
    uint32_t cls_shift;
    const uint8_t* qc;        // nullptr or nloci bytes
    const uint16_t* permtab;  // introsort permutation of n equal keys, n = 1..NHMAX, row n at n(n-1)/2
    uint32_t nloci;
    uint32_t ksize;
    uint32_t consistent;  // index memberships == flank/TR sets (verified on the GPU at load): `aux` may be used
    // optional gates: (canonical (k+1)-mer, locus) set of PREF.tre.kdb for -bu; (k-mer, locus) -> (min << 8 | max) of
    // PREF.bt.kmdb for -b.  Same slot layout and probing as the class table.
    const ClsSlot* tre; uint64_t tre_mask; uint32_t tre_shift;
    const ClsSlot* bait; uint64_t bait_mask; uint32_t bait_shift;
};

DBTK_HD uint64_t hash_idx(uint64_t key, uint32_t shift) {
    key ^= key >> 29;
    return (key * 0x9E3779B97F4A7C15ull) >> shift;
}
DBTK_HD uint64_t hash_cls(uint64_t kmer, uint32_t locus, uint32_t shift) {
    uint64_t x = kmer ^ ((uint64_t)locus * 0xD6E8FEB86659FD93ull);
    x ^= x >> 31;
    return (x * 0x9E3779B97F4A7C15ull) >> shift;
}

// kmerDBi.find(kmer): returns val or NOHIT.
DBTK_HD uint32_t idx_lookup(const DevTables& T, uint64_t key) {
    uint64_t i = hash_idx(key, T.idx_shift);
    for (;;) {
        const IdxSlot s = T.idx[i];
        if (s.key == key) return (uint32_t)s.val;
        if (s.key == NAN64) return NOHIT;
        i = (i + 1) & T.idx_mask;
    }
}

// same, returning val | aux << 32 (low word NOHIT on a miss)
DBTK_HD uint64_t idx_lookup64(const DevTables& T, uint64_t key) {
    uint64_t i = hash_idx(key, T.idx_shift);
    for (;;) {
        const IdxSlot s = T.idx[i];
        if (s.key == key) return s.val;
        if (s.key == NAN64) return (uint64_t)NOHIT;
        i = (i + 1) & T.idx_mask;
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

// Reverse complement of a k-mer held in the low 2k bits (getNuRC,
// src/aQueryFasta_thread.h:165-178): reverse the 2-bit symbols, complement.
DBTK_HD uint64_t revcomp2(uint64_t x, uint32_t k) {
    x = __builtin_bswap64(x);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    return (~x) >> (64 - 2 * k);
}

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

}  // namespace dbtk
#endif
