// dbtk_gz.h — the -a / -ae writer on the device: writeAlignments' lines (src/aQueryFasta_thread.cpp:1742-1759:
// `src dst title seq2 seq1 cigar2 annot2 cigar1 annot1`) assembled from the block the device reader parsed (dbtk_ingest.h)
// and the text records the walk kernel wrote (dbtk_walk.h: WTXT), then gzip-compressed, so that what crosses PCIe and what
// the host handles is the compressed stream only.  (The host cores of a GPU box are few next to what deflate costs: zlib at
// level 1 needs 1.25 CPU-seconds per million reads)
//
//   body_aln_len     one lane per kept pair: the length of its line (0: no record)
//   (scan)           ing_scan_step (dbtk_ingest.h) -> where each line starts
//   body_aln_write   one wave per pair: the line, byte by byte from the block, the record arena and the decimal of dst
//   body_gz_member   one wave per GZ_MEMBER bytes of text -> one gzip member (RFC 1952) holding one deflate block (RFC 1951)
//                    with dynamic Huffman codes.  Round 6: with string matching (LZ77).  The member's text is staged in LDS; every
//                    lane tokenizes its 1-KB span greedily against a hash table of its own (4-byte hashes of the span and of the
//                    512 bytes before it: the line before, whose title, CIGARs and annotations the line mostly repeats), matches of
//                    4 .. 258 bytes at distances up to 32 KB, never across the end of the span; literal / length and distance
//                    histograms in LDS; optimal code lengths by the in-place minimum-redundancy algorithm of Moffat & Katajainen
//                    (1995) on the sorted frequencies, limited to 15 bits by the usual Kraft-sum repair; canonical codes; every
//                    lane encodes its span's tokens at the bit offset a wave scan gives it; CRC-32 per span, combined by the
//                    zero-byte operator.  `zcat` of the members in order is the text.
//   body_gz_pack     the members moved back to back (one copy to the host)
// Instantiated with DevX on the GPU and with the coroutine lanes of tests/emu on the host (where zlib checks the stream).
#ifndef DBTK_GZ_H_
#define DBTK_GZ_H_

#include "dbtk_ingest.h"

namespace dbtk {

constexpr uint32_t GZ_MEMBER = 65536;   // bytes of text per gzip member: 64 lanes x GZ_SPAN
constexpr uint32_t GZ_SPAN = 1024;      // bytes one lane encodes
constexpr uint32_t GZ_STRIDE = GZ_MEMBER + GZ_MEMBER / 8 + 1024;  // room per member: an optimal prefix code spends < 9 bits per byte, + headers
constexpr uint32_t GZ_NLL = 286, GZ_ND = 30;  // literal / length codes, distance codes (RFC 1951 3.2.5)
constexpr uint32_t GZ_HDR_BITS = 3 + 5 + 5 + 4 + 19 * 3 + (GZ_NLL + GZ_ND) * 4;  // block header: BFINAL/BTYPE, HLIT, HDIST, HCLEN, 19 x 3, 316 lengths as 4-bit codes
constexpr uint32_t GZ_HASH = 128;       // entries of a lane's hash table (last position of a 4-byte hash)
constexpr uint32_t GZ_SEED = 512;       // bytes before a lane's span that are hashed too (never emitted: the lane before owns them)
constexpr uint32_t GZ_MAXM = 48;        // matches a lane records in its span (more: literals)
constexpr uint32_t GZ_MINM = 4, GZ_MAXLEN = 258, GZ_MAXDIST = 32768;

struct AlnLineArgs {
    const uint8_t* raw;                 // the slot's block (dbtk_ingest.h)
    const dbtk_ingest_span_t* spans;    // its kept pairs
    const IngestHdr* hdr;               // ... and how many
    const uint8_t* txt;                 // the walk's text arena: record = {u32 dst, u32 len} + len bytes
    const uint32_t* txt_idx;            // per pair: offset of its record, NAN32: none
    uint32_t* len;                      // [pairs + 1 + ING_SCAN_BLOCKS] line lengths -> line starts (exclusive scan in place)
    uint8_t* text;                      // the lines, back to back
    uint64_t* total;                    // [0] bytes of text, [1] lines
};
struct GzArgs {
    const uint8_t* text;
    const uint64_t* total;              // [0]: bytes of text
    uint8_t* out;                       // member c at out + c * GZ_STRIDE
    uint32_t* out_len;                  // [members + 1 + ING_SCAN_BLOCKS] bytes of member c -> (scan) where it starts in `packed`
    const uint32_t* crc_tab;            // [256] CRC-32 table, then [32]: the operator "append GZ_SPAN zero bytes" (column b = image of bit b)
    uint8_t* packed;
    uint64_t* packed_total;
    uint32_t lz;                        // 1: with string matching (LZ77); 0: literals only (DBTK_GZ_LZ=0: a third of the kernel's time, a 30 % larger stream)
};
struct GzSmem {
    uint32_t hist[GZ_NLL + 2];          // literal / length symbols
    uint32_t dhist[GZ_ND + 2];          // distance symbols
    uint16_t code[GZ_NLL + 2];          // bit-reversed canonical code of a symbol
    uint16_t dcode[GZ_ND + 2];
    uint8_t len[GZ_NLL + 2];
    uint8_t dlen[GZ_ND + 2];
    uint32_t sfreq[GZ_NLL + 2];         // (lane 0) present symbols sorted by frequency: frequency / work array of the length algorithm
    uint16_t ssym[GZ_NLL + 2];          //          ... and which symbol
    uint32_t lcrc[64];
    uint32_t crc_all;
    uint32_t crct[256 + 32];            // the CRC tables (a.crc_tab), staged
    uint16_t head[64][GZ_HASH];         // per lane: last position (in the member) of each 4-byte hash, 0xFFFF: none
    uint32_t mt[64][GZ_MAXM];           // per lane: its matches, (position in the span) | (length - 3) << 10 | (distance - 1) << 18 ... see gz_pack_match
    uint8_t nm[64];
};  // (32 KB: five members per CU at once.  Staging the member's 64 KB of text here as well — tried first — left ONE wave per CU and made the
    // kernel twenty times slower than the lanes reading their spans from global memory, whose lines stay in L1 while a lane walks them)
// a match as one word: position in the lane's span (10 bits), length - 3 (8 bits), distance - 1 in 14 bits: distances up to 16 384
// (a line is ~400 bytes: what a line repeats of the line before is well inside)
constexpr uint32_t GZ_DMAX = 16384;
DBTK_HD uint32_t gz_pack_match(uint32_t at, uint32_t l, uint32_t d) { return at | ((l - 3) << 10) | ((d - 1) << 18); }
// RFC 1951 3.2.5: length 3 .. 258 -> symbol 257 .. 285 + extra bits; distance 1 .. 32768 -> symbol 0 .. 29 + extra bits
DBTK_HD void gz_len_sym(uint32_t len, uint32_t* sym, uint32_t* eb, uint32_t* ev) {
    const uint32_t l = len - 3;
    if (len == 258) { *sym = 285; *eb = 0; *ev = 0; }
    else if (l < 8) { *sym = 257 + l; *eb = 0; *ev = 0; }
    else {
        const uint32_t e = (31u - (uint32_t)__builtin_clz(l)) - 2;
        *sym = 257 + 4 * e + 4 + ((l >> e) - 4); *eb = e; *ev = l & ((1u << e) - 1);
    }
}
DBTK_HD void gz_dist_sym(uint32_t dist, uint32_t* sym, uint32_t* eb, uint32_t* ev) {
    const uint32_t d = dist - 1;
    if (d < 4) { *sym = d; *eb = 0; *ev = 0; }
    else {
        const uint32_t e = (31u - (uint32_t)__builtin_clz(d)) - 1;
        *sym = 2 * e + 2 + ((d >> e) & 1u); *eb = e; *ev = d & ((1u << e) - 1);
    }
}

DBTK_HD uint32_t dec_digits(uint32_t v) { uint32_t n = 1; while (v >= 10) { v /= 10; ++n; } return n; }

template <class X>
DBTK_HD void body_aln_len(X& x, const AlnLineArgs& a) {
    const uint32_t nk = a.hdr->nkept;
    for (uint32_t q = x.bid() * (uint32_t)x.nthreads() + (uint32_t)x.tid(); q < nk; q += x.nblocks() * (uint32_t)x.nthreads()) {
        const uint32_t o = a.txt_idx[q];
        uint32_t n = 0;
        if (o != NAN32) {
            const uint32_t* rec = reinterpret_cast<const uint32_t*>(a.txt + o);
            const dbtk_ingest_span_t& S = a.spans[q];
            n = 2 + dec_digits(rec[0]) + 1 + S.title_len + 1 + S.seq_len[1] + 1 + S.seq_len[0] + 1 + rec[1] + 1;
        }
        a.len[q] = n;
    }
}
template <class X>
DBTK_HD void body_aln_scan(X& x, const AlnLineArgs& a, int step) {  // ING_SCAN_BLOCKS blocks of 64
    const uint32_t nk = a.hdr->nkept;
    const uint32_t all = ing_scan_step(x, a.len, nk, a.len + nk + 1, step);
    if (step == 1 && x.bid() == 0 && x.lane() == 0) { a.total[0] = all; a.len[nk] = all; }
}
template <class X>
DBTK_HD void body_aln_write(X& x, const AlnLineArgs& a) {  // one wave per pair
    const uint32_t nk = a.hdr->nkept, lane = (uint32_t)x.lane();
    for (uint32_t q = x.bid(); q < nk; q += x.nblocks()) {
        const uint32_t o = a.txt_idx[q];
        if (o == NAN32) continue;
        const uint32_t* rec = reinterpret_cast<const uint32_t*>(a.txt + o);
        const uint32_t dst = rec[0], rl = rec[1];
        const dbtk_ingest_span_t& S = a.spans[q];
        uint8_t* w = a.text + a.len[q];
        const uint32_t nd = dec_digits(dst);
        if (lane == 0) {  // srcLocus is -1 outside simulation mode: "."
            w[0] = '.'; w[1] = '\t';
            uint32_t v = dst;
            for (uint32_t i = nd; i-- > 0; v /= 10) w[2 + i] = (uint8_t)('0' + v % 10);
            w[2 + nd] = '\t';
        }
        w += 3 + nd;
        for (uint32_t i = lane; i < S.title_len; i += 64) w[i] = a.raw[S.title + i];
        if (lane == 0) w[S.title_len] = '\t';
        w += S.title_len + 1;
        for (uint32_t i = lane; i < S.seq_len[1]; i += 64) w[i] = a.raw[S.seq[1] + i];
        if (lane == 0) w[S.seq_len[1]] = '\t';
        w += S.seq_len[1] + 1;
        for (uint32_t i = lane; i < S.seq_len[0]; i += 64) w[i] = a.raw[S.seq[0] + i];
        if (lane == 0) w[S.seq_len[0]] = '\t';
        w += S.seq_len[0] + 1;
        const uint8_t* rt = reinterpret_cast<const uint8_t*>(rec + 2);
        for (uint32_t i = lane; i < rl; i += 64) w[i] = rt[i];
        if (lane == 0) w[rl] = '\n';
    }
    // (the number of lines: the pairs with a record)
    uint32_t c = 0;
    for (uint32_t q = x.bid() * 64 + lane; q < nk; q += x.nblocks() * 64) c += a.txt_idx[q] != NAN32 ? 1u : 0u;
    c = x.wave_sum(c);
    if (lane == 0 && c) x.atomic_add(&a.total[1], (uint64_t)c);
}

DBTK_HD uint32_t bitrev16(uint32_t v, uint32_t n) {  // the low n bits of v, reversed
    uint32_t r = 0;
    for (uint32_t i = 0; i < n; ++i) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}
// bits appended to a little-endian bit stream held as 32-bit words (pre-zeroed); `shared` words (the first and the last one a
// lane touches may also be a neighbour's) are or-ed atomically
template <class X>
struct BitOut {
    X& x; uint32_t* w; uint64_t acc; uint32_t nb; uint64_t word; bool first;
    DBTK_HD BitOut(X& x_, uint8_t* base, uint64_t bitpos) : x(x_), w(reinterpret_cast<uint32_t*>(base)), acc(0), nb((uint32_t)(bitpos & 31)), word(bitpos >> 5), first(true) {}
    DBTK_HD void put(uint32_t v, uint32_t n) {
        acc |= (uint64_t)v << nb;
        nb += n;
        if (nb >= 32) {
            if (first) { x.atomic_or32(&w[word], (uint32_t)acc); first = false; } else w[word] = (uint32_t)acc;
            ++word; acc >>= 32; nb -= 32;
        }
    }
    DBTK_HD void finish() { if (nb) x.atomic_or32(&w[word], (uint32_t)acc); }
};

// The tables body_gz_member reads: t[0 .. 256) the CRC-32 table (reflected polynomial 0xEDB88320), t[256 + b] the image of bit b under
// "clock the register through GZ_SPAN zero bytes".  (Host side; 288 words.)
inline void gz_tables(uint32_t* t) {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        t[i] = c;
    }
    for (uint32_t b = 0; b < 32; ++b) {
        uint32_t c = 1u << b;
        for (uint32_t z = 0; z < GZ_SPAN; ++z) c = t[c & 0xFFu] ^ (c >> 8);
        t[256 + b] = c;
    }
}

// Code lengths (<= 15 bits) and canonical codes (stored bit-reversed: Huffman codes enter the stream most significant bit first) of one
// alphabet from its histogram, by ONE lane.  A used alphabet gets a complete code (a single used symbol gets a second leaf); an unused one
// (no distance code at all: a member without matches) all zeros, which inflate accepts as long as no such code is used.
template <class SM>
DBTK_HD void gz_build_code(SM& sm, const uint32_t* hist, uint32_t nsym, uint8_t* len, uint16_t* code) {
    for (uint32_t s2 = 0; s2 < nsym; ++s2) { len[s2] = 0; code[s2] = 0; }
    // present symbols, sorted by frequency (ascending; ties by symbol: insertion sort, a few tens of symbols)
    uint32_t ns = 0;
    for (uint32_t s2 = 0; s2 < nsym; ++s2) {
        const uint32_t f = hist[s2];
        if (!f) continue;
        uint32_t j = ns++;
        while (j > 0 && sm.sfreq[j - 1] > f) { sm.sfreq[j] = sm.sfreq[j - 1]; sm.ssym[j] = sm.ssym[j - 1]; --j; }
        sm.sfreq[j] = f; sm.ssym[j] = (uint16_t)s2;
    }
    if (ns == 0) return;
    if (ns == 1) { len[sm.ssym[0]] = 1; len[sm.ssym[0] == 0 ? 1 : 0] = 1; }  // (a complete code needs two leaves)
    else {
        // Moffat & Katajainen, in place on sfreq[0 .. ns): frequencies -> code lengths
        uint32_t* A = sm.sfreq;
        const int nn = (int)ns;
        A[0] += A[1];
        int root = 0, leaf = 2, next;
        for (next = 1; next < nn - 1; ++next) {
            if (leaf >= nn || A[root] < A[leaf]) { A[next] = A[root]; A[root++] = (uint32_t)next; } else A[next] = A[leaf++];
            if (leaf >= nn || (root < next && A[root] < A[leaf])) { A[next] += A[root]; A[root++] = (uint32_t)next; } else A[next] += A[leaf++];
        }
        A[nn - 2] = 0;
        for (next = nn - 3; next >= 0; --next) A[next] = A[A[next]] + 1;
        int avbl = 1, used = 0, dpth = 0;
        root = nn - 2; next = nn - 1;
        while (avbl > 0) {
            while (root >= 0 && (int)A[root] == dpth) { ++used; --root; }
            while (avbl > used) { A[next--] = (uint32_t)dpth; --avbl; }
            avbl = 2 * used; ++dpth; used = 0;
        }
        // at most 15 bits: codes longer than that are counted as 15, then the Kraft sum is repaired by lengthening the
        // longest shorter codes (the lengths are handed out again in frequency order)
        uint32_t cnt[33];
        for (int i = 0; i <= 32; ++i) cnt[i] = 0;
        for (int i = 0; i < nn; ++i) ++cnt[A[i] > 32 ? 32 : A[i]];
        for (int i = 16; i <= 32; ++i) cnt[15] += cnt[i];
        uint32_t kraft = 0;
        for (int i = 15; i > 0; --i) kraft += cnt[i] << (15 - i);
        while (kraft != (1u << 15)) {
            --cnt[15];
            for (int i = 14; i > 0; --i) if (cnt[i]) { --cnt[i]; cnt[i + 1] += 2; break; }
            --kraft;
        }
        int j = nn - 1;  // the most frequent symbol gets the shortest length
        for (int l = 1; l <= 15; ++l) for (uint32_t t = 0; t < cnt[l]; ++t) len[sm.ssym[j--]] = (uint8_t)l;
    }
    // canonical codes (RFC 1951 3.2.2)
    uint32_t blc[16], nxt[16];
    for (int i = 0; i < 16; ++i) blc[i] = 0;
    for (uint32_t s2 = 0; s2 < nsym; ++s2) ++blc[len[s2]];
    blc[0] = 0;
    uint32_t cd = 0;
    for (int i = 1; i < 16; ++i) { cd = (cd + blc[i - 1]) << 1; nxt[i] = cd; }
    for (uint32_t s2 = 0; s2 < nsym; ++s2) { const uint32_t l = len[s2]; if (l) code[s2] = (uint16_t)bitrev16(nxt[l]++, l); }
}

template <class X>
DBTK_HD void body_gz_member(X& x, const GzArgs& a) {  // one wave per member; `out` is zero before
    GzSmem& sm = *x.template smem<GzSmem>();
    const uint64_t total = a.total[0];
    const uint32_t nmem = (uint32_t)((total + GZ_MEMBER - 1) / GZ_MEMBER), lane = (uint32_t)x.lane();
    for (uint32_t e = lane; e < 256 + 32; e += 64) sm.crct[e] = a.crc_tab[e];
    for (uint32_t c = x.bid(); c < nmem; c += x.nblocks()) {
        const uint8_t* src = a.text + (uint64_t)c * GZ_MEMBER;
        const uint8_t* const tx = src;  // the member's text, read where it lies
        const uint32_t n = (uint32_t)(total - (uint64_t)c * GZ_MEMBER < GZ_MEMBER ? total - (uint64_t)c * GZ_MEMBER : GZ_MEMBER);
        const uint32_t lo = lane * GZ_SPAN < n ? lane * GZ_SPAN : n, hi = lo + GZ_SPAN < n ? lo + GZ_SPAN : n;
        uint8_t* out = a.out + (uint64_t)c * GZ_STRIDE;
        x.sync();
        for (uint32_t s2 = lane; s2 < GZ_NLL + 2; s2 += 64) sm.hist[s2] = 0;
        if (lane < GZ_ND + 2) sm.dhist[lane] = 0;
        for (uint32_t h = 0; h < GZ_HASH; ++h) sm.head[lane][h] = 0xFFFFu;
        x.sync();
        // ---- pass 1: CRC of the lane's span; its tokens (greedy: the longest match at the last position with the same 4-byte hash)
        auto load32 = [&](uint32_t i) -> uint32_t {  // four bytes of the text at any offset (an unaligned load: fine in global memory)
            uint32_t v;
            __builtin_memcpy(&v, tx + i, 4);
            return v;
        };
        // all four bytes of a word are A, C, G or T
        auto is_dna4 = [&](uint32_t w) -> bool {
            bool ok = true;
            for (uint32_t q = 0; q < 4; ++q) { const uint32_t b = (w >> (8 * q)) & 0xFFu; ok = ok && (b == 'A' || b == 'C' || b == 'G' || b == 'T'); }
            return ok;
        };
        // 6 bits of hash + 1 bit "four bases": GZ_HASH entries, half of them for each kind of word
        auto hash_of = [&](uint32_t w) -> uint32_t { return ((w * 2654435761u) >> 26) | (is_dna4(w) ? 64u : 0u); };
        static_assert(GZ_HASH == 128, "hash_of makes 6 + 1 bits");
        // what `nb` bytes of a word would cost as literals, in bits: a base ~2, any other byte of these lines ~6
        auto worth_of = [&](uint32_t w, uint32_t nb) -> uint32_t {
            uint32_t t = 0;
            for (uint32_t q = 0; q < nb; ++q) { const uint32_t b = (w >> (8 * q)) & 0xFFu; t += (b == 'A' || b == 'C' || b == 'G' || b == 'T') ? 2u : 6u; }
            return t;
        };
        uint32_t crc = 0xFFFFFFFFu;
        {
            uint32_t i = lo;
            for (; i + 4 <= hi; i += 4) {
                const uint32_t w = load32(i);
#pragma unroll
                for (int q = 0; q < 4; ++q) crc = sm.crct[(crc ^ (w >> (8 * q))) & 0xFFu] ^ (crc >> 8);
            }
            for (; i < hi; ++i) crc = sm.crct[(crc ^ tx[i]) & 0xFFu] ^ (crc >> 8);
        }
        sm.lcrc[lane] = crc ^ 0xFFFFFFFFu;
        if (lo < hi) {  // the bytes before the span (the lane before owns them): hashed, not emitted
            const uint32_t s0 = lo > GZ_SEED ? lo - GZ_SEED : 0u;
            for (uint32_t i = s0; i < lo && i + 4 <= n; ++i) sm.head[lane][hash_of(load32(i))] = (uint16_t)i;
        }
        // The tokenizer is a state machine that every lane steps once per turn — look at a position (a literal, or the start of a match), or
        // extend the match it is in by up to four bytes — so that no lane waits for another lane's long match (the first form, a loop per
        // match inside a loop per token, made the wave wait for its longest match at every token: 8.8 ms per 34-MB block instead of ~1).
        uint32_t nmatch = 0;
        {
            uint32_t i = lo, mcand = 0, ml = 0, worth = 0;
            bool ext = false;
            while (x.ballot(i < hi)) {
                if (i < hi) {
                    if (!ext) {
                        uint32_t lit = 0x100u;
                        if (i + GZ_MINM <= hi) {
                            const uint32_t w = load32(i);
                            // Half of a lane's table is for words of four bases, half for everything else: the 300 bases of a line
                            // would otherwise have overwritten, by the time the next line comes, nine in ten of the entries its title,
                            // numbers, CIGARs and annotations left — the very strings the next line repeats
                            const uint32_t h = hash_of(w);
                            const uint32_t cand = sm.head[lane][h];
                            sm.head[lane][h] = (uint16_t)i;
                            if (a.lz && cand != 0xFFFFu && cand < i && i - cand <= GZ_DMAX && nmatch < GZ_MAXM && load32(cand) == w) {
                                ext = true; mcand = cand; ml = 4; worth = worth_of(w, 4);
                            } else lit = w & 0xFFu;
                        } else lit = tx[i];
                        if (lit != 0x100u) { x.lds_add(&sm.hist[lit], 1u); ++i; }
                    } else {
                        const uint32_t cap = hi - i < GZ_MAXLEN ? hi - i : GZ_MAXLEN;  // a token stays inside the lane's span
                        bool done = ml >= cap;
                        if (!done) {
                            if (ml + 4 <= cap) {
                                const uint32_t wa = load32(mcand + ml), wb = load32(i + ml), df = wa ^ wb;
                                const uint32_t eq = df ? (uint32_t)__builtin_ctz(df) >> 3 : 4u;
                                worth += worth_of(wb, eq);
                                ml += eq;
                                done = eq < 4 || ml >= cap;
                            } else {  // the last one to three bytes the span has room for
                                while (ml < cap && tx[mcand + ml] == tx[i + ml]) { worth += worth_of(tx[i + ml], 1); ++ml; }
                                done = true;
                            }
                        }
                        if (done) {
                            // worth it?  A match costs ~20 bits (length code, distance code, ~7 extra bits): random 4- and 5-base repeats —
                            // DNA is full of them — are cheaper as literals
                            if (worth >= 24) {
                                uint32_t ls, le, lv, ds, de, dv;
                                gz_len_sym(ml, &ls, &le, &lv);
                                gz_dist_sym(i - mcand, &ds, &de, &dv);
                                x.lds_add(&sm.hist[ls], 1u);
                                x.lds_add(&sm.dhist[ds], 1u);
                                sm.mt[lane][nmatch++] = gz_pack_match(i - lo, ml, i - mcand);
                                i += ml;
                            } else { x.lds_add(&sm.hist[tx[i]], 1u); ++i; }
                            ext = false;
                        }
                    }
                }
            }
        }
        sm.nm[lane] = (uint8_t)nmatch;
        x.sync();
        if (lane == 0) {
            sm.hist[256] = 1;  // end of block
            gz_build_code(sm, sm.hist, GZ_NLL, sm.len, sm.code);
            gz_build_code(sm, sm.dhist, GZ_ND, sm.dlen, sm.dcode);
            // CRC-32 of the member from the spans': crc(AB) = shift(crc(A), |B|) ^ crc(B), shift = the register clocked through |B| zero bytes
            uint32_t crc_all = 0;
            for (uint32_t l = 0; l < 64; ++l) {
                const uint32_t l0 = l * GZ_SPAN < n ? l * GZ_SPAN : n, l1 = l0 + GZ_SPAN < n ? l0 + GZ_SPAN : n, ln = l1 - l0;
                if (!ln) break;
                if (ln == GZ_SPAN) {
                    uint32_t r = 0;
                    for (uint32_t b = 0; b < 32; ++b) if ((crc_all >> b) & 1u) r ^= sm.crct[256 + b];
                    crc_all = r;
                } else for (uint32_t z = 0; z < ln; ++z) crc_all = sm.crct[crc_all & 0xFFu] ^ (crc_all >> 8);
                crc_all ^= sm.lcrc[l];
            }
            sm.crc_all = crc_all;
            // gzip header (RFC 1952): magic, deflate, no flags, no mtime, no extra flags, OS unknown
            out[0] = 0x1f; out[1] = 0x8b; out[2] = 8; out[3] = 0; out[4] = out[5] = out[6] = out[7] = 0; out[8] = 0; out[9] = 0xff;
        }
        x.sync();
        // ---- pass 2: where the lane's codes go: the bits of its tokens
        uint32_t bits = 0;
        {
            uint32_t mi = 0, nxt = nmatch ? (sm.mt[lane][0] & 0x3FFu) : 0xFFFFFFFFu;
            for (uint32_t i = lo; i < hi;) {
                if (i - lo == nxt) {
                    const uint32_t m = sm.mt[lane][mi++], ml = ((m >> 10) & 0xFFu) + 3, md = (m >> 18) + 1;
                    uint32_t ls, le, lv, ds, de, dv;
                    gz_len_sym(ml, &ls, &le, &lv);
                    gz_dist_sym(md, &ds, &de, &dv);
                    bits += (uint32_t)sm.len[ls] + le + (uint32_t)sm.dlen[ds] + de;
                    i += ml;
                    nxt = mi < nmatch ? (sm.mt[lane][mi] & 0x3FFu) : 0xFFFFFFFFu;
                } else { bits += sm.len[tx[i]]; ++i; }
            }
        }
        const uint32_t before = x.wave_excl_scan(bits), allbits = x.wave_sum(bits);
        const uint64_t bit0 = 80 + GZ_HDR_BITS;  // after the 10 header bytes and the block header
        if (lane == 0) {  // the block header: final block, dynamic codes; 286 literal/length codes, 30 distance codes, 19 code-length codes of
            BitOut<X> h(x, out, 80);  // which 0 .. 15 are 4 bits long (code = value) and 16, 17, 18 unused; then every length as such a code
            h.put(1u | (2u << 1), 3); h.put(GZ_NLL - 257, 5); h.put(GZ_ND - 1, 5); h.put(15, 4);
            for (int i = 0; i < 19; ++i) h.put(i < 3 ? 0u : 4u, 3);
            for (uint32_t s2 = 0; s2 < GZ_NLL; ++s2) h.put(bitrev16(sm.len[s2], 4), 4);
            for (uint32_t s2 = 0; s2 < GZ_ND; ++s2) h.put(bitrev16(sm.dlen[s2], 4), 4);
            h.finish();
        }
        {   // ---- pass 3: the codes
            BitOut<X> o(x, out, bit0 + before);
            uint32_t mi = 0, nxt = nmatch ? (sm.mt[lane][0] & 0x3FFu) : 0xFFFFFFFFu;
            for (uint32_t i = lo; i < hi;) {
                if (i - lo == nxt) {
                    const uint32_t m = sm.mt[lane][mi++], ml = ((m >> 10) & 0xFFu) + 3, md = (m >> 18) + 1;
                    uint32_t ls, le, lv, ds, de, dv;
                    gz_len_sym(ml, &ls, &le, &lv);
                    gz_dist_sym(md, &ds, &de, &dv);
                    o.put(sm.code[ls], sm.len[ls]);
                    if (le) o.put(lv, le);        // (extra bits: as they are, least significant bit first)
                    o.put(sm.dcode[ds], sm.dlen[ds]);
                    if (de) o.put(dv, de);
                    i += ml;
                    nxt = mi < nmatch ? (sm.mt[lane][mi] & 0x3FFu) : 0xFFFFFFFFu;
                } else { const uint32_t b = tx[i]; o.put(sm.code[b], sm.len[b]); ++i; }
            }
            if (lane == 63) {  // (lane 63's span ends the member, empty or not): end of block, up to the next byte, CRC-32, length (RFC 1952)
                o.put(sm.code[256], sm.len[256]);
                const uint64_t endbit = bit0 + allbits + sm.len[256];
                const uint32_t pad = (uint32_t)((8 - (endbit & 7)) & 7);
                if (pad) o.put(0, pad);
                o.put(sm.crc_all & 0xFFFFu, 16); o.put(sm.crc_all >> 16, 16);
                o.put(n & 0xFFFFu, 16); o.put(n >> 16, 16);
                a.out_len[c] = (uint32_t)((endbit + pad) / 8) + 8;
            }
            o.finish();
        }
    }
}
template <class X>
DBTK_HD void body_gz_scan(X& x, const GzArgs& a, int step) {  // ING_SCAN_BLOCKS blocks of 64
    const uint32_t nmem = (uint32_t)((a.total[0] + GZ_MEMBER - 1) / GZ_MEMBER);
    const uint32_t all = ing_scan_step(x, a.out_len, nmem, a.out_len + nmem + 1, step);
    if (step == 1 && x.bid() == 0 && x.lane() == 0) { a.packed_total[0] = all; a.out_len[nmem] = all; }
}
template <class X>
DBTK_HD void body_gz_pack(X& x, const GzArgs& a) {  // one wave per member
    const uint32_t nmem = (uint32_t)((a.total[0] + GZ_MEMBER - 1) / GZ_MEMBER), lane = (uint32_t)x.lane();
    for (uint32_t c = x.bid(); c < nmem; c += x.nblocks()) {
        const uint32_t o = a.out_len[c], n = a.out_len[c + 1] - o;
        const uint8_t* s = a.out + (uint64_t)c * GZ_STRIDE;
        for (uint32_t i = lane; i < n; i += 64) a.packed[o + i] = s[i];
    }
}

}  // namespace dbtk
#endif
