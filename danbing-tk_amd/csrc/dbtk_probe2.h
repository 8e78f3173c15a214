// dbtk_probe2.h — the probe kernel (K2) in its lean form: kfilter's look-ups (src/aQueryFasta_thread.cpp:190-224, the
// `kmerDBi.find(kmers[i])` of every position of both mates) through the offset-slotted, minimizer-grouped copy of the index
// (dbtk_tables.h: MzSlot).  Included by dbtk_kernels.h (needs BatchArgs); instantiated with DevX on the GPU and with the
// coroutine lanes of tests/emu on the host.
//
// One wave per surviving PAIR, one mate per half-wave: lane l of a half owns the NPL consecutive positions l * NPL ..
// l * NPL + NPL - 1 of its mate (NPL = 5: reads up to 32 * 5 + m - 1 bases).  Everything a position needs is a shift of
// one 32-base word W held by its lane and of W's reverse complement:
//   raw bytes (two dwords per lane, prefetched two pairs ahead)  ->  2 bits per base, 16 bits per lane, into LDS
//   W, RW                      <- three LDS words
//   canonical k-mer            <- W >> .., RW >> ..                                  (per position)
//   hashed canonical m-mer     <- W >> .., RW >> ..  -> LDS                          (per BASE: each is hashed once)
//   minimizer of every window  <- NPL + WN - 1 LDS words, sliding minimum in registers, twice: first and last smallest
//                                 hash (the canonical form of a window is the read's strand or the other one, and the
//                                 builder's tie rule is stated on the canonical form)
//   bucket, slot               <- the minimizer's hash, its offset in the canonical k-mer
//   ONE 16-byte load per position (the positions sharing a minimizer read one 128-byte line), ONE comparison
//   a slot that turned keys away and holds another key: the overflow table, one 16-byte load per probe
//   results: 4 bytes per position (`aux`), + 4 (`val`) for a read whose found k-mers are not all unique to one locus,
//            transposed through LDS so that they leave as 16-byte stores.
// No run detection, no bucket staging, no search loop, no scalar loop over positions: the unrolled body is the same for
// every read, and the per-read header (found positions, one index value or not) comes from three half-wave reductions.
#ifndef DBTK_PROBE2_H_
#define DBTK_PROBE2_H_

namespace dbtk {

template <int NPL>
struct __attribute__((aligned(16))) Probe2SmemT {
    uint32_t pk[2][20];              // 2-bit stream of each mate from its 4-byte-aligned start: 16 words (+ slack)
    uint16_t vd[2][40];              // validity bits of the same bases (only for a pair with a non-ACGT byte)
    uint32_t hm[2][32 * NPL + 16];   // hashed canonical m-mer by base position; the last 16 stay 0xFFFFFFFF
    uint32_t res[2][2][32 * NPL];    // [mate][aux | val][position]: the results on their way to 16-byte stores
};

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

template <int NPL, int WN, class X>
DBTK_HD void body_probe2(X& x, const BatchArgs& a) {
    static_assert(NPL + WN - 1 <= 16, "window offsets travel in four bits");
    typedef Probe2SmemT<NPL> SM;
    SM& sm = *x.template smem<SM>();
    const int lane = x.lane();
    const uint32_t hl = (uint32_t)lane & 31u, half = (uint32_t)lane >> 5;
    const DevTables& T = a.T;
    const uint32_t k = T.ksize, m = T.mz_m;  // k - m + 1 == WN (the launcher's condition)
    const uint32_t ns = *a.nsurv;
    const uint32_t tend = ns - a.t0 < a.tcap ? ns : a.t0 + a.tcap;
    const uint32_t npr = ns > a.t0 ? tend - a.t0 : 0;  // pairs of this chunk; pair i -> hit-buffer rows 2i, 2i + 1
    // Which pairs this wave takes.  The survivor list is sorted by locus (body_surv_*, below): each of the chip's 8 XCDs — blocks
    // b, b + 8, b + 16, ... share one — takes one contiguous eighth of it, its waves striding through that eighth together, so the
    // reads of a locus are looked up at about the same time on ONE XCD and find each other's buckets in its L2 (a bucket is
    // asked for by ~4 reads per batch of all-hit reads; unsorted, each of them fetched it from HBM).
    const uint32_t nblk = x.nblocks();
    const bool xcd = (nblk & 7u) == 0;
    const uint32_t S = xcd ? nblk >> 3 : nblk;                                  // stride of this wave's pairs
    const uint32_t chunk = xcd ? (npr + 7) >> 3 : npr;
    const uint32_t lo = xcd ? (x.bid() & 7u) * chunk : 0u;
    const uint32_t hi = lo + chunk < npr ? lo + chunk : npr;                    // this wave's pairs: lo + w, lo + w + S, ... < hi
    const uint32_t first = lo + (xcd ? x.bid() >> 3 : x.bid());
    const uint32_t lmax = 32u * NPL + m - 1;  // bases the lanes of a half cover (the launcher promised no read is longer)
    const uint64_t kmask = (1ull << (2 * k)) - 1;
    const uint32_t mmask = (uint32_t)((1ull << (2 * m)) - 1);
    const uint32_t p0 = hl * NPL;
    if (hl < 16) sm.hm[half][32 * NPL + hl] = 0xFFFFFFFFu;
    // Three-deep fetch pipeline, all loads unconditional (clamped to something valid) so that they stay in flight:
    // while pair i is looked up, the bytes of pair i + S are on their way into registers, the offsets of pair i + 2S
    // are being fetched, and the survivor entry of pair i + 3S.
    auto surv_of = [&](uint32_t i) { return a.surv[a.t0 + (i < hi ? i : 0u)]; };
    uint32_t rw0 = 0, rw1 = 0;  // pair i: dwords 2 hl and 2 hl + 1 of the mate, from its 4-byte-aligned start
    uint64_t o0C = 0, o1C = 0;  //         its offsets
    uint64_t o0B = 0, o1B = 0;  // pair i + S: offsets (in flight)
    uint32_t pairA = 0;         // pair i + 2S: survivor entry (in flight)
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
    {
        const uint32_t i0 = first;
        if (i0 < hi) {
            fetch_offsets(x.uni(surv_of(i0)));
            o0C = o0B; o1C = o1B;
            fetch_bytes(o0C, o1C);
        }
        if (i0 + S < hi) fetch_offsets(x.uni(surv_of(i0 + S)));
        pairA = surv_of(i0 + 2 * S);
    }
    DBTK_STAMP_DECL
    for (uint32_t i = first; i < hi; i += S) {
        DBTK_STAMP(43);  // loop
        const uint64_t o0 = o0C, o1 = o1C;
        uint32_t len = (uint32_t)(o1 - o0);
        if (len > lmax) { *a.errflag = DBTK_ERR_READ_TOO_LONG; len = lmax; }
        const uint64_t a0 = o0 & ~3ull;
        const uint32_t rsh = (uint32_t)(o0 - a0);
        const uint32_t d0 = rw0, d1 = rw1;
        {   // advance the pipeline
            const bool hasB = i + S < hi, hasA = i + 2 * S < hi;
            o0C = hasB ? o0B : 0ull; o1C = hasB ? o1B : 0ull;
            fetch_bytes(o0C, o1C);
            fetch_offsets(hasA ? x.uni(pairA) : x.uni(pairA) * 0u);
            pairA = surv_of(i + 3 * S);
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
            // hashed canonical m-mer at the lane's own base positions
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const uint32_t fwm = (uint32_t)(W >> (2 * (32 - m - j))) & mmask, rcm = (uint32_t)(RW >> (2 * j)) & mmask;
                sm.hm[half][p0 + j] = p0 + j < nmm ? mmer_hash2(fwm, rcm) : 0xFFFFFFFFu;
            }
            // canonical k-mers (the m-mer hashes make their round trip through LDS meanwhile)
            uint64_t km[NPL];
            bool isfw[NPL];
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const uint64_t fw = (W >> (2 * (32 - k - j))) & kmask, rc = (RW >> (2 * j)) & kmask;
                isfw[j] = fw < rc;
                km[j] = isfw[j] ? fw : rc;
            }
            DBTK_STAMP(40);  // fetch pipeline, pack, windows, m-mer hashes
            x.sync();
            uint32_t f[NPL + WN - 1], g[NPL + WN - 1], mf[NPL], ml[NPL];
#pragma unroll
            for (int t = 0; t < NPL + WN - 1; ++t) {
                f[t] = mz_order(sm.hm[half][p0 + t], (uint32_t)t);  // smallest hash, FIRST one among equals
                g[t] = f[t] ^ 15u;                                   //                LAST one among equals
            }
            sliding_min<NPL, WN>(f, mf);
            sliding_min<NPL, WN>(g, ml);
            DBTK_STAMP(16);  // minimizers
            uint4 q[NPL];
            bool act[NPL];
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                act[j] = p0 + j < nk;
                // strand of the read canonical: the builder's first offset is the first one here, window offset t - j;
                // other strand: its offsets run backwards, the builder's first is the last one here, WN - 1 - (t - j)
                const uint32_t sel = isfw[j] ? mf[j] : ml[j];
                const uint32_t off = (sel & 15u) + (isfw[j] ? (uint32_t)-j : (uint32_t)(j + WN - 16));
                const uint32_t b = mz_bucket(sel >> 4, (uint32_t)T.mz_mask);
                size_t at = act[j] ? (size_t)b * MZ_SLOTS + mz_slot(off) : 0;
#ifdef DBTK_STAMPS
                if (a.P.diag & 512) at &= 1023;  // diagnostic: every level-1 look-up in the first 16 KB of the table (cache hits)
#endif
                q[j] = reinterpret_cast<const uint4*>(T.mz)[at];
            }
            bool pend[NPL];
            bool anyp = false;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const bool hit = act[j] && q[j].x == (uint32_t)km[j] && (q[j].y & 0x7FFFFFFFu) == (uint32_t)(km[j] >> 32);
                pend[j] = act[j] && !hit && (q[j].y >> 31);  // the slot turned keys away: ask the overflow table
                rv[j] = hit ? ((uint64_t)q[j].w << 32) | q[j].z : (uint64_t)NOHIT;
                anyp |= pend[j];
            }
            DBTK_STAMP(18);  // level 1
#ifdef DBTK_STAMPS
            if (a.P.diag & 256) anyp = false;  // diagnostic: no level-2 look-ups (wrong results)
#endif
            if (x.ballot(anyp)) {
                uint32_t oi[NPL];
#pragma unroll
                for (int j = 0; j < NPL; ++j) oi[j] = ovf_hash(km[j]) & (uint32_t)T.ovf_mask;
                do {
#pragma unroll
                    for (int j = 0; j < NPL; ++j) q[j] = reinterpret_cast<const uint4*>(T.ovf)[pend[j] ? oi[j] : 0u];
                    anyp = false;
#pragma unroll
                    for (int j = 0; j < NPL; ++j) {
                        if (!pend[j]) continue;
                        const uint64_t key = ((uint64_t)q[j].y << 32) | q[j].x;
                        if (key == km[j]) { rv[j] = ((uint64_t)q[j].w << 32) | q[j].z; pend[j] = false; }
                        else if (key == MZ_EMPTY) pend[j] = false;  // (rv stays NOHIT)
                        else oi[j] = (oi[j] + 1) & (uint32_t)T.ovf_mask;
                        anyp |= pend[j];
                    }
                } while (x.ballot(anyp));
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
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                rv[j] = (uint64_t)NOHIT;
                if (p0 + j < nk) {
                    const uint64_t kq = window_kmer(sm.pk[half], sm.vd[half], rsh + p0 + j, k, nullptr, nullptr);
                    if (kq != NAN64) rv[j] = idx_lookup64(T, kq);
                }
            }
        }
        {   // the read's results: found positions, and whether they are all unique to one and the same locus
            uint32_t cnt = 0, vmx = 0, vmn = 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const uint32_t v = p0 + j < nk ? (uint32_t)rv[j] : NOHIT;
                const bool fnd = v != NOHIT;
                cnt += fnd ? 1u : 0u;
                vmx = fnd && v > vmx ? v : vmx;
                vmn = fnd && v < vmn ? v : vmn;
                sm.res[half][0][p0 + j] = fnd ? (uint32_t)(rv[j] >> 32) : AUX_MISS;
                sm.res[half][1][p0 + j] = v;
            }
            const uint32_t nh = x.half_sum(cnt), hmx = x.half_max(vmx), hmn = ~x.half_max(~vmn);
            const bool uniform = T.consistent && nh && hmx == hmn && !(hmx & 1u);
            const uint32_t row = 2 * i + half;
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
        DBTK_STAMP(42);  // results
    }
    DBTK_STAMP_FLUSH;
}

// ---- builders.  Level 1 from the finished plain index (keys unique, val | aux with them): pass 0 claims slots and counts
// the keys turned away; pass 1 (after the host has sized the overflow table from that count) files those keys in level 2.
struct MzBuildArgs {
    const IdxBucket* idx;
    uint64_t nslots;    // 4 per IdxBucket
    MzSlot* mz;
    uint32_t mz_mask;   // buckets - 1
    MzSlot* ovf;        // pass 1 only
    uint32_t ovf_mask;
    uint32_t ksize, m, pass;
    uint64_t* nturned;  // pass 0: += keys that found their slot taken
};
template <class X>
DBTK_HD void body_mz_insert(X& x, const MzBuildArgs& a) {
    uint64_t turned = 0;
    for (uint64_t i = (uint64_t)x.bid() * x.nthreads() + x.tid(); i < a.nslots; i += (uint64_t)x.nblocks() * x.nthreads()) {
        uint64_t key = a.idx[i >> 2].key[i & 3];
        if (key == NAN64) continue;
        key &= ~IDX_OVF;
        const uint64_t va = a.idx[i >> 2].val[i & 3];
        uint32_t mz28, off;
        mz_of_kmer(key, a.ksize, a.m, &mz28, &off);
        MzSlot* s = a.mz + ((size_t)mz_bucket(mz28, a.mz_mask) * MZ_SLOTS + mz_slot(off));
        if (a.pass == 0) {
            if (x.atomic_cas(&s->key, MZ_EMPTY, key) == MZ_EMPTY) { s->val = (uint32_t)va; s->aux = (uint32_t)(va >> 32); }
            else { x.atomic_or(&s->key, MZ_TURNED); ++turned; }
        } else if ((s->key & ~MZ_TURNED) != key) {
            uint32_t o = ovf_hash(key) & a.ovf_mask;
            while (x.atomic_cas(&a.ovf[o].key, MZ_EMPTY, key) != MZ_EMPTY) o = (o + 1) & a.ovf_mask;
            a.ovf[o].val = (uint32_t)va; a.ovf[o].aux = (uint32_t)(va >> 32);
        }
    }
    if (turned) x.atomic_add(a.nturned, turned);
}
// every slot of a level-1 / level-2 table free
template <class X>
DBTK_HD void body_mz_fill(X& x, MzSlot* t, uint64_t n) {
    for (uint64_t i = (uint64_t)x.bid() * x.nthreads() + x.tid(); i < n; i += (uint64_t)x.nblocks() * x.nthreads()) {
        t[i].key = MZ_EMPTY; t[i].val = 0; t[i].aux = 0;
    }
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
};
// canonical k-mer of the k bytes at p; NAN64 if one of them is not A, C, G or T
DBTK_HD uint64_t kmer_of_bytes(const uint8_t* p, uint32_t k) {
    uint64_t fw = 0, rc = 0;
    bool ok = true;
    for (uint32_t i = 0; i < k; ++i) {
        const uint32_t c = p[i], code = ((c >> 1) ^ (c >> 2)) & 3u;
        ok = ok && (c == 'A' || c == 'C' || c == 'G' || c == 'T');
        fw = (fw << 2) | code;
        rc = (rc >> 2) | ((uint64_t)(3u - code) << (2 * (k - 1)));
    }
    return ok ? (fw < rc ? fw : rc) : NAN64;
}
// one lane per survivor: its key, counted
template <class X>
DBTK_HD void body_surv_key(X& x, const SurvSortArgs& a) {
    const uint32_t n = *a.nsurv, k = a.T.ksize, nloci = a.T.nloci;
    const uint32_t NF = (a.P.n_filter && a.P.nm_filter) ? a.P.n_filter : 4u;  // subfilter's sampled positions (AQ.cpp:172-188); four without it
    for (uint32_t t = x.bid() * (uint32_t)x.nthreads() + (uint32_t)x.tid(); t < n; t += x.nblocks() * (uint32_t)x.nthreads()) {
        const uint64_t r = 2 * (uint64_t)a.surv[t];
        const uint64_t o0 = a.off[r];
        const uint32_t len = (uint32_t)(a.off[r + 1] - o0);
        uint32_t key = nloci;
        if (len >= k) {
            const uint32_t L = len - k + 1, S = NF > 1 ? L / (NF - 1) : 0;
            for (uint32_t sidx = 0; sidx < NF && key == nloci; ++sidx) {
                const uint32_t pos = sidx != NF - 1 ? sidx * S : L - 1;
                const uint64_t km = pos < L ? kmer_of_bytes(a.seq + o0 + pos, k) : NAN64;
                if (km == NAN64) continue;
                const uint32_t v = idx_lookup(a.T, km);
                if (v != NOHIT) key = (v & 1u) ? a.T.vv[(v >> 1) + 1] : v >> 1;
            }
        }
        if (key > nloci) key = nloci;
        a.key[t] = key;
        x.atomic_add(&a.hist[key], 1u);
    }
}
// hist[] -> exclusive prefix sums (the first place of each key), in two steps of SCAN_BLOCKS one-wave blocks: the sum of every
// block's chunk, then each block scans its chunk from the sum of the chunks before it
constexpr uint32_t SCAN_BLOCKS = 256;
template <class X>
DBTK_HD void body_surv_scan(X& x, const SurvSortArgs& a, int step) {
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
        if (i0 + lane < hi) a.hist[i0 + lane] = base + ex;
        base += x.wave_sum(v);
    }
}
// one lane per survivor: into its key's next place
template <class X>
DBTK_HD void body_surv_scatter(X& x, const SurvSortArgs& a) {
    const uint32_t n = *a.nsurv;
    for (uint32_t t = x.bid() * (uint32_t)x.nthreads() + (uint32_t)x.tid(); t < n; t += x.nblocks() * (uint32_t)x.nthreads())
        a.sorted[x.atomic_add(&a.hist[a.key[t]], 1u)] = a.surv[t];
}

}  // namespace dbtk
#endif
