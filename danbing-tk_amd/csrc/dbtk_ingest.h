// dbtk_ingest.h — the reader of the batch loop on the device: record boundaries, on-the-fly mate pairing and the batch
// arrays of a block of raw FASTA / FASTQ bytes (src/aQueryFasta_thread.cpp:1918-1976, critical section A: `getline(title);
// getline(seq); [getline(+); getline(qual);]`, prunePEinfo :455-462, park-by-title pairing, the minimal read size :1940-1943).
//
// The host only moves bytes: it reads the file in fixed-size chunks into pinned buffers and copies them to the device.
// A record is L = 2 (FASTA) or 4 (FASTQ) lines, so where records start follows from COUNTING newlines from the start
// of the file — no heuristic on '>' / '@':
//   body_ing_count  newlines per 4-KB tile of the block                            (one wave per tile, 16-byte loads)
//   body_ing_scan   tile counts -> first line number of every tile, line total     (two steps of ING_SCAN_BLOCKS waves)
//   body_ing_lines  the byte offset of every newline, in line order
//   body_ing_pairs  per pair of consecutive records: titles equal after prunePEinfo?  both reads long enough?
//   body_ing_place  kept pairs -> their rank and the offsets of their reads in the batch's flat sequence array (scan)
//   body_ing_gather the reads (and, FASTQ with the bait filter, their qualities) copied into the batch arrays; as in the
//                   reference seqs[2p] is the record that COMPLETED the pair (the second in the file), seqs[2p + 1] the parked one
//   body_ing_carry  what follows the last whole pair goes in front of the next block's bytes (device to device)
// The fast path is the interleaved file (`samtools fasta -n` of a name-collated BAM: mates adjacent).  Whatever else a
// block holds — a pair of neighbours with different titles, an odd record, more lines than the line table holds — raises
// a flag, the block is NOT aligned, and the caller goes on from the block's first byte with the host reader, whose
// park-by-title map gives the reference's outcome for any input (dbtk_cli.cpp).  Up to that byte nothing was parked, so
// the host reader starts in exactly the state the reference's would be in.
// Instantiated with DevX on the GPU and with the coroutine lanes of tests/emu on the host.
#ifndef DBTK_INGEST_H_
#define DBTK_INGEST_H_

#include "../../include/dbtk.h"
#include "dbtk_tables.h"

namespace dbtk {

constexpr uint32_t ING_TILE = 4096;        // bytes one wave scans per step: 64 lanes x 16 bytes x 4
constexpr uint32_t ING_SCAN_BLOCKS = 256;  // waves of the two-step scans
enum : uint32_t {
    ING_F_DIRTY = 1,      // two neighbouring records with different titles: not an interleaved block
    ING_F_LINES_OVF = 2,  // more lines than the line table holds
    ING_F_CARRY_OVF = 4,  // the bytes after the last whole pair do not fit in front of the next block
    ING_F_TAIL = 8,       // last block of the input and it does not end with a whole pair
};
struct IngestHdr {  // what the host reads back per block (64 bytes)
    uint32_t base;        // first byte of the block in the slot's buffer: head - the bytes carried over from the block before
    uint32_t nlines;      // newlines in [base, end)
    uint32_t npairs;      // whole pairs of records
    uint32_t nkept;       // pairs whose two reads are both at least min_read long (the others are dropped, AQ.cpp:1940-1943)
    uint32_t flags;       // ING_F_*
    uint32_t cut;         // first byte after the last whole pair
    uint32_t carry;       // end - cut
    uint32_t pad;
    uint64_t flat_bytes;  // bytes of the kept reads
    uint64_t maxlen;      // the longest kept read
    uint64_t pad2[3];
};
struct IngestArgs {
    const uint8_t* raw;       // the slot's buffer (16-byte aligned, readable up to the next multiple of 16 past `end`)
    const uint32_t* base_in;  // where the block starts (written by the previous block's body_ing_carry; `head` for the first block)
    uint32_t end;             // head + bytes of this chunk
    uint32_t L;               // lines per record
    uint32_t min_read;        // Cthreshold + k - 1
    uint32_t last;            // the input ends with this block
    uint32_t* tile_cnt;       // [ntiles + 1 + ING_SCAN_BLOCKS]
    uint32_t* nlpos;          // [line_cap]
    uint32_t line_cap;
    uint32_t* pk;             // [line_cap / (2 L) + 1 + 2 * ING_SCAN_BLOCKS] per pair: 1 << 31 | len(parked) << 15 | len(completing); 0: dropped
    uint32_t* kept;           // [pairs] kept pair -> pair of the block
    uint64_t* off;            // [2 * pairs + 1]
    uint8_t* flat;            // the kept reads, back to back
    uint8_t* qual;            // optional: their qualities, same offsets ('!' where a quality string is shorter than its read)
    dbtk_ingest_span_t* spans;  // optional, per kept pair: where its title, reads and qualities lie in the block (for the host's writers)
    IngestHdr* hdr;
    uint8_t* next_raw;        // the next block's buffer
    uint32_t* base_out;       // ... and where its bytes will start
    uint32_t head;            // room in front of a chunk for the carried-over bytes
};

// bit i of the result: byte i of the 16 at raw + o is a newline and o + i lies in [base, end)
DBTK_HD uint32_t ing_nl_mask(const uint8_t* raw, uint32_t o, uint32_t base, uint32_t end) {
    if (o >= end || o + 16 <= base) return 0;
    const uint64_t* p = reinterpret_cast<const uint64_t*>(raw + o);
    uint32_t m = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint64_t x = p[h] ^ 0x0A0A0A0A0A0A0A0Aull;
        const uint64_t z = ~(((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x | 0x7F7F7F7F7F7F7F7Full);  // 0x80 in every zero byte, exactly
        m |= (uint32_t)(((z >> 7) * 0x0102040810204080ull) >> 56) << (8 * h);
    }
    const uint32_t lo = base > o ? base - o : 0, hi = end - o < 16 ? end - o : 16;
    return m & ((1u << hi) - 1) & ~((1u << lo) - 1);
}

template <class X>
DBTK_HD void body_ing_count(X& x, const IngestArgs& a) {
    const uint32_t base = *a.base_in, end = a.end, lane = (uint32_t)x.lane();
    const uint32_t ntiles = (end + ING_TILE - 1) / ING_TILE;
    for (uint32_t t = x.bid(); t < ntiles; t += x.nblocks()) {
        uint32_t c = 0;
#pragma unroll
        for (uint32_t it = 0; it < ING_TILE / 1024; ++it) c += (uint32_t)__builtin_popcount(ing_nl_mask(a.raw, t * ING_TILE + it * 1024 + lane * 16, base, end));
        c = x.wave_sum(c);
        if (lane == 0) a.tile_cnt[t] = c;
    }
}

// v[0 .. n) -> exclusive prefix sums in place, in two steps of ING_SCAN_BLOCKS one-wave blocks (part[]: ING_SCAN_BLOCKS words);
// returns (step 1) the total
template <class X>
DBTK_HD uint32_t ing_scan_step(X& x, uint32_t* v, uint32_t n, uint32_t* part, int step) {
    const uint32_t lane = (uint32_t)x.lane(), b = x.bid();
    const uint32_t per = ((n + ING_SCAN_BLOCKS - 1) / ING_SCAN_BLOCKS + 63) & ~63u;
    const uint32_t lo = b * per < n ? b * per : n, hi = lo + per < n ? lo + per : n;
    if (step == 0) {
        uint32_t s = 0;
        for (uint32_t i = lo + lane; i < hi; i += 64) s += v[i];
        s = x.wave_sum(s);
        if (lane == 0) part[b] = s;
        return 0;
    }
    uint32_t before = 0, all = 0;
    for (uint32_t i = lane; i < ING_SCAN_BLOCKS; i += 64) { const uint32_t p = part[i]; all += p; before += i < b ? p : 0u; }
    before = x.wave_sum(before);
    all = x.wave_sum(all);
    for (uint32_t i0 = lo; i0 < hi; i0 += 64) {
        const uint32_t w = i0 + lane < hi ? v[i0 + lane] : 0u;
        const uint32_t ex = x.wave_excl_scan(w);
        if (i0 + lane < hi) v[i0 + lane] = before + ex;
        before += x.wave_sum(w);
    }
    return all;
}
template <class X>
DBTK_HD void body_ing_scan(X& x, const IngestArgs& a, int step) {  // launched with ING_SCAN_BLOCKS blocks of 64
    const uint32_t ntiles = (a.end + ING_TILE - 1) / ING_TILE;
    const uint32_t total = ing_scan_step(x, a.tile_cnt, ntiles, a.tile_cnt + ntiles + 1, step);
    if (step == 1 && x.bid() == 0 && x.lane() == 0) {
        IngestHdr& h = *a.hdr;
        h.base = *a.base_in;
        h.nlines = total;
        if (total > a.line_cap) { h.flags |= ING_F_LINES_OVF; h.npairs = 0; }
        else h.npairs = total / (2 * a.L);
    }
}

template <class X>
DBTK_HD void body_ing_lines(X& x, const IngestArgs& a) {
    const uint32_t base = *a.base_in, end = a.end, lane = (uint32_t)x.lane();
    const uint32_t ntiles = (end + ING_TILE - 1) / ING_TILE;
    for (uint32_t t = x.bid(); t < ntiles; t += x.nblocks()) {
        uint32_t run = a.tile_cnt[t];
#pragma unroll
        for (uint32_t it = 0; it < ING_TILE / 1024; ++it) {
            const uint32_t o = t * ING_TILE + it * 1024 + lane * 16;
            uint32_t m = ing_nl_mask(a.raw, o, base, end);
            const uint32_t c = (uint32_t)__builtin_popcount(m);
            uint32_t idx = run + x.wave_excl_scan(c);
            while (m) {
                const uint32_t bit = (uint32_t)__builtin_ctz(m);
                m &= m - 1;
                if (idx < a.line_cap) a.nlpos[idx] = o + bit;
                ++idx;
            }
            run += x.wave_sum(c);
        }
    }
}

// the four (offset, length) spans of record r of the block: title (pruned), sequence, quality
struct IngRec { uint32_t t, tn, s, sn, q, qn; };
DBTK_HD IngRec ing_record(const IngestArgs& a, uint32_t base, uint32_t r) {
    const uint32_t l = r * a.L;
    IngRec R;
    R.t = l ? a.nlpos[l - 1] + 1 : base;
    const uint32_t te = a.nlpos[l];
    R.tn = te - R.t;
    R.s = te + 1;
    R.sn = a.nlpos[l + 1] - R.s;
    R.q = R.s; R.qn = 0;
    if (a.L == 4) { R.q = a.nlpos[l + 2] + 1; R.qn = a.nlpos[l + 3] - R.q; }
    // prunePEinfo, AQ.cpp:455-462
    if (R.tn >= 2 && a.raw[R.t + R.tn - 2] == '/' && (a.raw[R.t + R.tn - 1] == '1' || a.raw[R.t + R.tn - 1] == '2')) R.tn -= 2;
    return R;
}

template <class X>
DBTK_HD void body_ing_pairs(X& x, const IngestArgs& a) {  // one lane per pair
    const uint32_t np = a.hdr->npairs, base = a.hdr->base;
    uint32_t dirty = 0, longest = 0;
    for (uint32_t p = x.bid() * (uint32_t)x.nthreads() + (uint32_t)x.tid(); p < np; p += x.nblocks() * (uint32_t)x.nthreads()) {
        const IngRec P = ing_record(a, base, 2 * p), C = ing_record(a, base, 2 * p + 1);  // the parked record, the one completing the pair
        bool same = P.tn == C.tn;
        for (uint32_t i = 0; same && i < P.tn; ++i) same = a.raw[P.t + i] == a.raw[C.t + i];
        if (!same) dirty = 1;
        const bool keep = P.sn >= a.min_read && C.sn >= a.min_read;
        // (the packed lengths are 15 bits each: a kept read of 32 767 bases or more cannot be placed by body_ing_place, whose offsets
        // would disagree with what body_ing_gather copies — the block goes to the host reader instead)
        if (keep && (P.sn >= 0x7FFFu || C.sn >= 0x7FFFu)) dirty = 1;
        a.pk[p] = keep ? 0x80000000u | ((P.sn < 0x7FFFu ? P.sn : 0x7FFFu) << 15) | (C.sn < 0x7FFFu ? C.sn : 0x7FFFu) : 0u;
        if (keep) { longest = P.sn > longest ? P.sn : longest; longest = C.sn > longest ? C.sn : longest; }
    }
    if (dirty) x.atomic_or32(&a.hdr->flags, ING_F_DIRTY);
    if (longest > 0) x.atomic_max(&a.hdr->maxlen, (uint64_t)longest);
}

// kept pairs -> rank and read offsets (two scans at once: pairs and bytes); launched with ING_SCAN_BLOCKS blocks of 64
template <class X>
DBTK_HD void body_ing_place(X& x, const IngestArgs& a, int step) {
    const uint32_t n = a.hdr->npairs, lane = (uint32_t)x.lane(), b = x.bid();
    const uint32_t pcap = a.line_cap / (2 * a.L) + 1;
    uint32_t* partc = a.pk + pcap;
    uint32_t* partb = partc + ING_SCAN_BLOCKS;
    const uint32_t per = ((n + ING_SCAN_BLOCKS - 1) / ING_SCAN_BLOCKS + 63) & ~63u;
    const uint32_t lo = b * per < n ? b * per : n, hi = lo + per < n ? lo + per : n;
    auto bytes_of = [](uint32_t v) { return v ? ((v >> 15) & 0x7FFFu) + (v & 0x7FFFu) : 0u; };
    if (step == 0) {
        uint32_t sc = 0, sb = 0;
        for (uint32_t i = lo + lane; i < hi; i += 64) { const uint32_t v = a.pk[i]; sc += v >> 31; sb += bytes_of(v); }
        sc = x.wave_sum(sc); sb = x.wave_sum(sb);
        if (lane == 0) { partc[b] = sc; partb[b] = sb; }
        return;
    }
    uint32_t bc = 0, bb = 0, allc = 0, allb = 0;
    for (uint32_t i = lane; i < ING_SCAN_BLOCKS; i += 64) {
        const uint32_t c = partc[i], y = partb[i];
        allc += c; allb += y;
        if (i < b) { bc += c; bb += y; }
    }
    bc = x.wave_sum(bc); bb = x.wave_sum(bb); allc = x.wave_sum(allc); allb = x.wave_sum(allb);
    for (uint32_t i0 = lo; i0 < hi; i0 += 64) {
        const uint32_t v = i0 + lane < hi ? a.pk[i0 + lane] : 0u;
        const uint32_t k = v >> 31, y = bytes_of(v);
        const uint32_t q = bc + x.wave_excl_scan(k), o = bb + x.wave_excl_scan(y);
        if (k) {
            a.kept[q] = i0 + lane;
            a.off[2 * (uint64_t)q] = o;                       // the completing record's read first
            a.off[2 * (uint64_t)q + 1] = o + (v & 0x7FFFu);   // then the parked one's
        }
        bc += x.wave_sum(k); bb += x.wave_sum(y);
    }
    if (b == 0 && lane == 0) {
        a.off[2 * (uint64_t)allc] = allb;
        a.hdr->nkept = allc;
        a.hdr->flat_bytes = allb;
    }
}

template <class X>
DBTK_HD void body_ing_gather(X& x, const IngestArgs& a) {  // one wave per kept pair
    const uint32_t nk = a.hdr->nkept, base = a.hdr->base, lane = (uint32_t)x.lane();
    for (uint32_t q = x.bid(); q < nk; q += x.nblocks()) {
        const uint32_t p = a.kept[q];
        const IngRec P = ing_record(a, base, 2 * p), C = ing_record(a, base, 2 * p + 1);
        const uint64_t o0 = a.off[2 * (uint64_t)q], o1 = a.off[2 * (uint64_t)q + 1];
        for (uint32_t i = lane; i < C.sn; i += 64) a.flat[o0 + i] = a.raw[C.s + i];
        for (uint32_t i = lane; i < P.sn; i += 64) a.flat[o1 + i] = a.raw[P.s + i];
        if (a.qual) {
            for (uint32_t i = lane; i < C.sn; i += 64) a.qual[o0 + i] = i < C.qn ? a.raw[C.q + i] : (uint8_t)'!';
            for (uint32_t i = lane; i < P.sn; i += 64) a.qual[o1 + i] = i < P.qn ? a.raw[P.q + i] : (uint8_t)'!';
        }
        if (a.spans && lane == 0) {
            dbtk_ingest_span_t& S = a.spans[q];
            S.title = C.t; S.title_len = C.tn;
            S.seq[0] = C.s; S.seq_len[0] = C.sn; S.qual[0] = C.q; S.qual_len[0] = C.qn;
            S.seq[1] = P.s; S.seq_len[1] = P.sn; S.qual[1] = P.q; S.qual_len[1] = P.qn;
        }
    }
}

template <class X>
DBTK_HD void body_ing_carry(X& x, const IngestArgs& a) {  // one block
    IngestHdr& h = *a.hdr;
    const uint32_t base = h.base;
    const uint32_t cut = h.npairs ? a.nlpos[h.npairs * 2 * a.L - 1] + 1 : base;
    const uint32_t carry = a.end - cut;
    const bool fits = carry <= a.head;
    if (fits && !a.last && a.next_raw)
        for (uint32_t i = (uint32_t)x.tid(); i < carry; i += (uint32_t)x.nthreads()) a.next_raw[a.head - carry + i] = a.raw[cut + i];
    if (x.tid() == 0) {
        h.cut = cut;
        h.carry = carry;
        if (!fits) h.flags |= ING_F_CARRY_OVF;
        if (a.last && carry) h.flags |= ING_F_TAIL;
        if (a.base_out) *a.base_out = fits ? a.head - carry : a.head;
    }
}

}  // namespace dbtk
#endif
