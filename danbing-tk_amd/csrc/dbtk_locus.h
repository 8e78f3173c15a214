// dbtk_locus.h — the probe kernel with the locus' own k-mers resident in LDS (K2L): kfilter's look-ups
// (src/aQueryFasta_thread.cpp:190-224, `kmerDBi.find(kmers[i])` for every position of both mates) for the pairs of a batch that
// come from loci, answered from a per-locus image of the index instead of the global tables.
//
// What bounds the probe kernel on this chip is the number of 128-byte lines it asks the memory system for (dbtk_tables.h: ~45 G
// requests per second whatever their size), and the minimizer-grouped global table still costs ~33 lines per read.  But the survivor
// list is in locus order (dbtk_probe2.h: body_surv_*), the k-mers of a read from locus l are — but for sequencing errors and the
// rare shared k-mer — k-mers of locus l, and a locus has ~2 000 of them: 16 KB.  So every locus gets an IMAGE: its part of the
// index (every key whose index value names the locus, with what the global look-up would return for it) as a small bucketed hash
// table, stored in HBM exactly as it will sit in LDS.  A workgroup takes ITEMS = (locus, up to LOC_CH consecutive pairs of its
// segment of the list), copies the locus' image into LDS with coalesced 16-byte loads (one pass over ~34 KB for ~64 pairs: ~4 lines
// per pair) and its waves look every position of their pairs up there: 260 LDS probes per pair instead of 70 global lines.  Only what
// the image cannot answer goes to the global index: positions whose k-mer is not in the image (a k-mer with a sequencing error —
// which must be PROVEN absent from the whole index, so it is looked up there —, a k-mer of another locus) and the keys shared
// between loci (their index value is a `vv` list: marked in the image, fetched from the index).  The result per position is what
// the global tables would have given, bit for bit: an image is a partition of the index, never a second opinion.
//
// The image (LOC_HDR bytes of header, then nb = 2^lgnb buckets of 32 bytes, then nb displacement bytes):
//   bucket  = tag[4], pay[4].  tag = low 32 bits of the canonical k-mer.  The bucket number is (hi ^ h(lo)) & (nb - 1) with hi the
//             k-mer's bits from 32 up, so given the bucket and lo the low lgnb bits of hi are implied; the bits of hi above them
//             (at most 8: 2k - 32 - lgnb <= 8 is the launcher's condition) sit in pay[31:24]: tag + bucket + those bits = the key.
//   pay     = extra << 24 | LOC_MULTI | LOC_FLANK | LOC_TR | slot.  A k-mer unique to the locus is its FLANK k-mer or its TR k-mer
//             number `slot` (counter trbeg[l] + slot: IdxBucket::val's high word); LOC_MULTI: the key's index value is a vv list
//             (or the RPGG's sets disagree about it): ask the index.  A free slot holds LOC_EMPTY (FLANK and TR both set, which
//             no entry has).
// Hash and displace: a look-up must be exactly ONE bucket for every lane (a second probe that one lane in twenty needs is a second
// probe every wave pays for), so no bucket may overflow — and at the loads that keep an image small (up to 0.8) a plain hash overflows
// all the time.  The keys are therefore placed group by group: a key's GROUP is a second hash of its low word, every group has a
// displacement byte, and the bucket of a key is (hi ^ h(lo)) + disp[group] (still a bijection between (bucket, low word) and the key).
// The builder (one thread per locus, its keys sorted by group) gives the large groups their displacement first and tries 0 .. 255
// until every key of the group finds a free slot.  A group for which none works is LEFT OUT of the image (never seen at these loads;
// counted): "not in the image" never means "not in the index", the index answers for it like for any other k-mer the image does not hold.
// Built on the GPU from the finished plain index (body_loc_count -> sizes on the host -> body_loc_scatter -> body_loc_place);
// written to / read from the sidecar file PREF.dbtk.idx by the host (dbtk_hip.hip).
#ifndef DBTK_LOCUS_H_
#define DBTK_LOCUS_H_

namespace dbtk {

constexpr uint32_t LOC_MULTI = 1u << 23, LOC_FLANK = 1u << 22, LOC_TR = 1u << 21, LOC_SLOT = (1u << 21) - 1;
constexpr uint32_t LOC_EMPTY = 0xFFFFFFFEu;  // pay of a free slot (FLANK and TR both set)
constexpr uint32_t LOC_MISS = 0xFFFFFFFFu;   // look-up result: the k-mer is not in the image
constexpr uint32_t LOC_HDR = 16;             // header bytes: lgnb, keys left out, trbeg[l], locus
constexpr uint32_t LOC_CH = 64;              // pairs per item
constexpr uint32_t LOC_MIN_PAIRS = 16;       // a locus with fewer pairs in the chunk is not worth fetching its image for (16 - 34 KB: the lines of a few pairs'
                                             // global look-ups — and a locus with few pairs is more often one that reads merely resemble)
constexpr uint32_t LOC_LG_MIN = 5, LOC_LG_MAX = 11;
struct LocusDir {
    uint32_t off16;  // of the image in the arena, in units of 16 bytes
    uint32_t bytes;  // of the image (a multiple of 16); 0: the locus has none (too large, or built badly): its pairs take the global path
    uint32_t lgnb;
    uint32_t trbeg;  // the locus' first counter (T.trbeg[l]: here so that the kernel needs no further load for it)
};
// smallest lgnb that leaves at most 8 bits of the key unaccounted for
DBTK_HD uint32_t loc_lg_min(uint32_t k) { return 2 * k > 40 + LOC_LG_MIN ? 2 * k - 40 : LOC_LG_MIN; }
// image of a locus with n keys: the smallest with a load of at most 0.8
DBTK_HD uint32_t loc_lgnb_for(uint64_t nkeys, uint32_t k) {
    uint32_t lg = loc_lg_min(k);
    while ((16ull << lg) < 5 * nkeys) ++lg;
    return lg;
}
DBTK_HD uint32_t loc_image_bytes(uint32_t lgnb) { return LOC_HDR + (32u << lgnb) + (1u << lgnb > 16u ? 1u << lgnb : 16u); }
// (loc_base / loc_group — the image's bucket and group hashes — live in dbtk_tables.h: the graph look-up of the walk kernels uses them too)
static_assert(LOC_EMPTY == GIMG_EMPTY, "one free-slot marker for both kinds of image");

// ------------------------------------------------------------------ build --
struct LocBuildArgs {
    const IdxBucket* idx;
    uint64_t nslots;        // 4 per IdxBucket
    const uint32_t* vv;
    const uint32_t* trbeg;
    uint32_t nloci, ksize;
    uint32_t* cnt;          // [nloci] pass 0: keys per locus
    const LocusDir* dir;    // later passes
    uint8_t* arena;
    uint32_t* bad;          // [nloci] != 0: the image could not be built (a TR k-mer's number does not fit the slot field, ...)
    const uint64_t* ebeg;   // [nloci + 1] first entry of each locus in the entry arrays
    uint32_t* ecur;         // [nloci] entries gathered so far
    uint64_t* ekey; uint32_t* epay;   // entries as gathered
    uint64_t* skey; uint32_t* spay;   // ... and sorted by group
    uint16_t* gscr;         // per locus gstride scratch words (2 * buckets + 2)
    uint32_t gstride;
    uint64_t* nleft;        // += keys left out of their image
    const GrSlot* gr;       // the graph images' source: the hashed graph table
    uint64_t gr_nslots;
    const ClsSlot* cls;     // the class table (flank / TR sets by (k-mer, locus)): the class AT THIS LOCUS of a key shared between loci
    uint64_t cls_mask;      //   rides in its image entry next to LOC_MULTI (nullptr: such entries carry LOC_MULTI alone)
    uint32_t cls_shift;
    uint64_t idx_mask;      // body_loc_verify: the plain index as a look-up table (buckets - 1, 64 - log2(buckets))
    uint32_t idx_shift;
    uint32_t* vcnt;         // [nloci] body_loc_verify: entries of the image that ARE the index's + the keys the image says it left out
};
// keys per locus
template <class X>
DBTK_HD void body_loc_count(X& x, const LocBuildArgs& a) {
    for (uint64_t i = (uint64_t)x.bid() * x.nthreads() + x.tid(); i < a.nslots; i += (uint64_t)x.nblocks() * x.nthreads()) {
        const uint64_t key = a.idx[i >> 2].key[i & 3];
        if (key == NAN64) continue;
        const uint32_t v = (uint32_t)a.idx[i >> 2].val[i & 3];
        if (v & 1) {
            const uint32_t n = a.vv[v >> 1];
            for (uint32_t j = 0; j < n; ++j) { const uint32_t l = a.vv[(v >> 1) + 1 + j]; if (l < a.nloci) x.atomic_add(&a.cnt[l], 1u); }
        } else if ((v >> 1) < a.nloci) x.atomic_add(&a.cnt[v >> 1], 1u);
    }
}
// the (key, pay) of every membership, gathered per locus: ent[ebeg[l] ..)
// pay of a key whose index value is a vv list (shared between loci), in the image of locus l of that list: LOC_MULTI, and — round 5 —
// the key's class at l (flankDB[l] / trKmers[l]: what assignTRkmc asks, AQ.cpp:1466-1468) when the class table knows it: the fused probe
// kernel then resolves a pair with such k-mers without the index (dbtk_locus.h: body_probe_locus)
DBTK_HD uint32_t loc_multi_pay(const LocBuildArgs& a, uint64_t key, uint32_t l) {
    if (!a.cls) return LOC_MULTI;
    const uint32_t c = kl_lookup(a.cls, a.cls_mask, a.cls_shift, key, l);
    if (c == CLS_NONE) return LOC_MULTI;
    if (c == CLS_FLANK) return LOC_MULTI | LOC_FLANK;
    const uint32_t slot = c - a.trbeg[l];
    if (c < a.trbeg[l] || slot > LOC_SLOT) return LOC_MULTI;
    return LOC_MULTI | LOC_TR | slot;
}
template <class X>
DBTK_HD void loc_scatter_one(X& x, const LocBuildArgs& a, uint64_t key, uint32_t l, uint32_t val, uint32_t aux) {
    if (l >= a.nloci || !a.dir[l].bytes) return;
    uint32_t pay;
    if (val & 1) pay = loc_multi_pay(a, key, l);
    else if (aux == CLS_NONE) pay = LOC_MULTI;
    else if (aux == CLS_FLANK) pay = LOC_FLANK;
    else {
        const uint32_t slot = aux - a.trbeg[l];
        if (aux < a.trbeg[l] || slot > LOC_SLOT) { a.bad[l] = 1; return; }
        pay = LOC_TR | slot;
    }
    const uint64_t at = a.ebeg[l] + x.atomic_add(&a.ecur[l], 1u);
    if (at >= a.ebeg[l + 1]) { a.bad[l] = 1; return; }
    a.ekey[at] = key; a.epay[at] = pay;
}
template <class X>
DBTK_HD void body_loc_scatter(X& x, const LocBuildArgs& a) {
    for (uint64_t i = (uint64_t)x.bid() * x.nthreads() + x.tid(); i < a.nslots; i += (uint64_t)x.nblocks() * x.nthreads()) {
        uint64_t key = a.idx[i >> 2].key[i & 3];
        if (key == NAN64) continue;
        key &= ~IDX_OVF;
        const uint64_t va = a.idx[i >> 2].val[i & 3];
        const uint32_t v = (uint32_t)va, aux = (uint32_t)(va >> 32);
        if (v & 1) {
            const uint32_t n = a.vv[v >> 1];
            for (uint32_t j = 0; j < n; ++j) loc_scatter_one(x, a, key, a.vv[(v >> 1) + 1 + j], v, aux);
        } else loc_scatter_one(x, a, key, v >> 1, v, aux);
    }
}
// ---- the same images for the GRAPH (the lean walk kernel, dbtk_walkfast.h): every (canonical k-mer, locus) entry of the hashed graph
// table (dbtk_tables.h: GrSlot) in its locus' image, pay = extra << 24 | counter << 11 | the entry's 11 flag bits (out-edges and
// "is a node" of both strands, "is a TR k-mer").  13 bits of counter: a locus with 8 191 TR k-mers or more has no graph image.
constexpr uint32_t GLOC_SLOT_MAX = (1u << 13) - 1;
template <class X>
DBTK_HD void body_gloc_count(X& x, const LocBuildArgs& a) {
    for (uint64_t i = (uint64_t)x.bid() * x.nthreads() + x.tid(); i < a.gr_nslots; i += (uint64_t)x.nblocks() * x.nthreads()) {
        const GrSlot s = a.gr[i];
        if (s.kmer == NAN64 || s.li == ~0ull) continue;
        const uint32_t l = (uint32_t)(s.li >> 32);
        if (l < a.nloci) x.atomic_add(&a.cnt[l], 1u);
    }
}
template <class X>
DBTK_HD void body_gloc_scatter(X& x, const LocBuildArgs& a) {
    for (uint64_t i = (uint64_t)x.bid() * x.nthreads() + x.tid(); i < a.gr_nslots; i += (uint64_t)x.nblocks() * x.nthreads()) {
        const GrSlot s = a.gr[i];
        if (s.kmer == NAN64 || s.li == ~0ull) continue;
        const uint32_t l = (uint32_t)(s.li >> 32), info = (uint32_t)s.li;
        if (l >= a.nloci || !a.dir[l].bytes) continue;
        const uint32_t slot = info >> GR_SLOT_SHIFT;
        if (slot >= GLOC_SLOT_MAX) { a.bad[l] = 1; continue; }
        const uint64_t at = a.ebeg[l] + x.atomic_add(&a.ecur[l], 1u);
        if (at >= a.ebeg[l + 1]) { a.bad[l] = 1; continue; }
        a.ekey[at] = s.kmer; a.epay[at] = (info & ((1u << GR_SLOT_SHIFT) - 1)) | (slot << GR_SLOT_SHIFT);
    }
}

// One thread per locus: its keys sorted by group (a counting sort through the locus' scratch words), then group after group — the
// largest first — the smallest displacement with which every key of the group finds a free slot.
template <class X>
DBTK_HD void body_loc_place(X& x, const LocBuildArgs& a) {
    for (uint32_t l = x.bid() * (uint32_t)x.nthreads() + (uint32_t)x.tid(); l < a.nloci; l += x.nblocks() * (uint32_t)x.nthreads()) {
        const LocusDir d = a.dir[l];
        if (!d.bytes) continue;
        uint32_t* w = reinterpret_cast<uint32_t*>(a.arena + 16ull * d.off16);
        const uint32_t nb = 1u << d.lgnb, mask = nb - 1;
        uint32_t* bks = w + LOC_HDR / 4;
        uint8_t* disp = reinterpret_cast<uint8_t*>(bks + 8 * nb);
        w[0] = d.lgnb; w[1] = 0; w[2] = d.trbeg; w[3] = l;
        for (uint32_t i = 0; i < 8 * nb; ++i) bks[i] = (i & 7) >= 4 ? LOC_EMPTY : 0u;
        for (uint32_t i = 0; i < (nb > 16 ? nb : 16u); ++i) disp[i] = 0;
        const uint64_t e0 = a.ebeg[l];
        const uint32_t n = a.ecur[l] < (uint32_t)(a.ebeg[l + 1] - e0) ? a.ecur[l] : (uint32_t)(a.ebeg[l + 1] - e0);
        uint16_t* gcnt = a.gscr + (size_t)l * a.gstride;   // keys per group, then (after the prefix sums) the groups' first places
        uint16_t* gpos = gcnt + nb + 1;                    // next free place of each group during the scatter
        if (n > 0xFFF0u) { a.bad[l] = 1; continue; }
        for (uint32_t g = 0; g <= nb; ++g) gcnt[g] = 0;
        for (uint32_t i = 0; i < n; ++i) ++gcnt[loc_group((uint32_t)a.ekey[e0 + i], (uint32_t)(a.ekey[e0 + i] >> 32), d.lgnb)];
        uint32_t run = 0;
        for (uint32_t g = 0; g < nb; ++g) { const uint32_t c = gcnt[g]; gcnt[g] = (uint16_t)run; gpos[g] = (uint16_t)run; run += c; }
        gcnt[nb] = (uint16_t)run;
        for (uint32_t i = 0; i < n; ++i) {
            const uint64_t key = a.ekey[e0 + i];
            const uint32_t at = gpos[loc_group((uint32_t)key, (uint32_t)(key >> 32), d.lgnb)]++;
            a.skey[e0 + at] = key; a.spay[e0 + at] = a.epay[e0 + i];
        }
        uint32_t left = 0, maxsz = 0;
        for (uint32_t g = 0; g < nb; ++g) { const uint32_t sz = gcnt[g + 1] - gcnt[g]; maxsz = sz > maxsz ? sz : maxsz; }
        for (uint32_t cls = maxsz; cls >= 1; --cls)  // the largest groups first, while the buckets are empty
            for (uint32_t g = 0; g < nb; ++g) {
                const uint32_t f = gcnt[g], sz = gcnt[g + 1] - f;
                if (sz != cls) continue;
                uint32_t dd = 0;
                for (; dd < 256; ++dd) {  // every key of the group a free slot? (keys of the group that share a bucket need as many)
                    bool ok = true;
                    for (uint32_t t = 0; t < sz && ok; ++t) {
                        const uint64_t key = a.skey[e0 + f + t];
                        const uint32_t b = (loc_base((uint32_t)key, (uint32_t)(key >> 32), d.lgnb) + dd) & mask;
                        uint32_t need = 1;
                        for (uint32_t u = 0; u < t; ++u) {
                            const uint64_t k2 = a.skey[e0 + f + u];
                            if (((loc_base((uint32_t)k2, (uint32_t)(k2 >> 32), d.lgnb) + dd) & mask) == b) ++need;
                        }
                        uint32_t fr = 0;
                        for (int s2 = 0; s2 < 4; ++s2) fr += bks[8 * b + 4 + s2] == LOC_EMPTY ? 1u : 0u;
                        ok = fr >= need;
                    }
                    if (ok) break;
                }
                if (dd == 256) {
#if defined(DBTK_LOC_DEBUG) && !defined(__HIPCC__)
                    fprintf(stderr, "locus %u lgnb %u n %u: group %u size %u not placed:", l, d.lgnb, n, g, sz);
                    for (uint32_t t = 0; t < sz && t < 12; ++t) fprintf(stderr, " %llx", (unsigned long long)a.skey[e0 + f + t]);
                    fprintf(stderr, "\n");
#endif
                    left += sz; continue;  // (the group stays out of the image: the index answers for its keys)
                }
                disp[g] = (uint8_t)dd;
                for (uint32_t t = 0; t < sz; ++t) {
                    const uint64_t key = a.skey[e0 + f + t];
                    const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
                    const uint32_t b = (loc_base(lo, hi, d.lgnb) + dd) & mask, extra = hi >> d.lgnb;
                    if (extra > 0xFF) { a.bad[l] = 1; break; }
                    for (int s2 = 0; s2 < 4; ++s2)
                        if (bks[8 * b + 4 + s2] == LOC_EMPTY) { bks[8 * b + s2] = lo; bks[8 * b + 4 + s2] = a.spay[e0 + f + t] | (extra << 24); break; }
                }
            }
        w[1] = left;
        if (left) x.atomic_add(a.nleft, (uint64_t)left);
    }
}

// The same placement by a WAVE per locus, the image being built in LDS and copied out once (round 5: one thread per locus took 0.19 s
// of a 0.64-s table build at release scale — every word of the image and of the scratch arrays a memory round trip of its own).  What
// differs from body_loc_place is the order of a group's keys — ascending by key instead of by where the gather left them, so that the
// image is the same bit for bit from run to run whatever order the atomics of the passes before ran in — and with it at most which of a
// bucket's four slots a key takes; the greedy order of the groups (largest first, then by number) and the rule for a group's displacement
// (the smallest with which every key finds a free slot) are the same.  LGMAX: the launch takes the loci whose images have at most
// 2^LGMAX buckets and more than 2^LGLOW (three launches by image class, like the kernels that use the images).
template <int LGMAX>
struct LocPlaceSmemT {
    uint32_t bks[8u << LGMAX];     // the image's buckets as they will lie in the arena
    uint32_t gcnt[(1u << LGMAX) + 1];  // keys per group, then the groups' first places
    uint32_t gpos[1u << LGMAX];    // next free place of each group during the scatter
    uint8_t disp[1u << LGMAX];
    uint64_t gk[64];               // the keys of the group being placed
    uint32_t gp[64];               // ... and their pay words
};
template <int LGMAX, int LGLOW, class X>
DBTK_HD void body_loc_place_wave(X& x, const LocBuildArgs& a) {
    typedef LocPlaceSmemT<LGMAX> SM;
    SM& sm = *x.template smem<SM>();
    const uint32_t lane = (uint32_t)x.lane();
    for (uint32_t l = x.bid(); l < a.nloci; l += x.nblocks()) {
        const LocusDir d = a.dir[l];
        if (!d.bytes || d.lgnb > (uint32_t)LGMAX || (LGLOW >= 0 && d.lgnb <= (uint32_t)LGLOW)) continue;  // (uniform: one locus per wave)
        uint32_t* w = reinterpret_cast<uint32_t*>(a.arena + 16ull * d.off16);
        const uint32_t nb = 1u << d.lgnb, mask = nb - 1;
        const uint64_t e0 = a.ebeg[l];
        const uint32_t n = a.ecur[l] < (uint32_t)(a.ebeg[l + 1] - e0) ? a.ecur[l] : (uint32_t)(a.ebeg[l + 1] - e0);
        if (n > 0xFFF0u) { if (lane == 0) a.bad[l] = 1; continue; }
        x.sync();  // (the locus before is out of LDS)
        for (uint32_t i = lane; i < 8 * nb; i += 64) sm.bks[i] = (i & 7) >= 4 ? LOC_EMPTY : 0u;
        for (uint32_t i = lane; i < nb; i += 64) { sm.disp[i] = 0; sm.gcnt[i] = 0; }
        if (lane == 0) sm.gcnt[nb] = 0;
        x.sync();
        // keys per group; the groups' first places (exclusive prefix sums, 64 groups at a time)
        for (uint32_t i = lane; i < n; i += 64) { const uint64_t key = a.ekey[e0 + i]; x.lds_add(&sm.gcnt[loc_group((uint32_t)key, (uint32_t)(key >> 32), d.lgnb)], 1u); }
        x.sync();
        uint32_t run = 0, maxsz = 0;
        for (uint32_t g0 = 0; g0 < nb; g0 += 64) {
            const uint32_t c = g0 + lane < nb ? sm.gcnt[g0 + lane] : 0u;
            const uint32_t ex = x.wave_excl_scan(c);
            x.sync();
            if (g0 + lane < nb) { sm.gcnt[g0 + lane] = run + ex; sm.gpos[g0 + lane] = run + ex; }
            run += x.bcast(ex + c, 63);
            maxsz = c > maxsz ? c : maxsz;
        }
        if (lane == 0) sm.gcnt[nb] = run;
        maxsz = ~x.wave_min(~maxsz);
        x.sync();
        // the keys by group, in whatever order the lanes get there (skey / spay) ...
        for (uint32_t i = lane; i < n; i += 64) {
            const uint64_t key = a.ekey[e0 + i];
            const uint32_t at = x.lds_add(&sm.gpos[loc_group((uint32_t)key, (uint32_t)(key >> 32), d.lgnb)], 1u);
            a.skey[e0 + at] = key; a.spay[e0 + at] = a.epay[e0 + i];
        }
        x.sync();
        // ... and within a group ascending by key (back into ekey / epay: the gathered order is not needed any more): a key's place is
        // the number of smaller keys in its group (the keys of a locus are distinct)
        for (uint32_t i = lane; i < n; i += 64) {
            const uint64_t key = a.skey[e0 + i];
            const uint32_t g = loc_group((uint32_t)key, (uint32_t)(key >> 32), d.lgnb), f = sm.gcnt[g], e = sm.gcnt[g + 1];
            uint32_t rank = 0;
            for (uint32_t j = f; j < e; ++j) rank += a.skey[e0 + j] < key ? 1u : 0u;
            a.ekey[e0 + f + rank] = key; a.epay[e0 + f + rank] = a.spay[e0 + i];
        }
        x.sync();
        uint32_t left = 0, bad = 0;
        for (uint32_t cls = maxsz; cls >= 1; --cls)  // the largest groups first, while the buckets are empty
            for (uint32_t g0 = 0; g0 < nb; g0 += 64) {
                const uint32_t gl = g0 + lane;
                uint64_t todo = x.ballot(gl < nb && sm.gcnt[gl < nb ? gl + 1 : 0] - sm.gcnt[gl < nb ? gl : 0] == cls);
                while (todo) {
                    const uint32_t g = g0 + (uint32_t)__builtin_ctzll(todo);
                    todo &= todo - 1;
                    const uint32_t f = sm.gcnt[g], sz = cls;
                    if (sz > 64) { left += sz; continue; }  // (no group comes near that: the index answers for its keys)
                    x.sync();
                    if (lane < sz) { sm.gk[lane] = a.ekey[e0 + f + lane]; sm.gp[lane] = a.epay[e0 + f + lane]; }
                    x.sync();
                    uint32_t dd = 256;
                    for (uint32_t d0 = 0; d0 < 256 && dd == 256; d0 += 64) {  // 64 displacements at a time: every key of the group a free slot?
                        const uint32_t dt = d0 + lane;
                        bool ok = true;
                        for (uint32_t t = 0; t < sz && ok; ++t) {
                            const uint64_t key = sm.gk[t];
                            const uint32_t b = (loc_base((uint32_t)key, (uint32_t)(key >> 32), d.lgnb) + dt) & mask;
                            uint32_t need = 1;
                            for (uint32_t u = 0; u < t; ++u) {
                                const uint64_t k2 = sm.gk[u];
                                if (((loc_base((uint32_t)k2, (uint32_t)(k2 >> 32), d.lgnb) + dt) & mask) == b) ++need;
                            }
                            uint32_t fr = 0;
                            for (int s2 = 0; s2 < 4; ++s2) fr += sm.bks[8 * b + 4 + s2] == LOC_EMPTY ? 1u : 0u;
                            ok = fr >= need;
                        }
                        const uint64_t okm = x.ballot(ok);
                        if (okm) dd = d0 + (uint32_t)__builtin_ctzll(okm);
                    }
                    if (dd == 256) { left += sz; continue; }  // (the group stays out of the image: the index answers for its keys)
                    if (lane == 0) {
                        sm.disp[g] = (uint8_t)dd;
                        for (uint32_t t = 0; t < sz; ++t) {
                            const uint64_t key = sm.gk[t];
                            const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
                            const uint32_t b = (loc_base(lo, hi, d.lgnb) + dd) & mask, extra = hi >> d.lgnb;
                            if (extra > 0xFF) { bad = 1; break; }
                            for (int s2 = 0; s2 < 4; ++s2)
                                if (sm.bks[8 * b + 4 + s2] == LOC_EMPTY) { sm.bks[8 * b + s2] = lo; sm.bks[8 * b + 4 + s2] = sm.gp[t] | (extra << 24); break; }
                        }
                    }
                }
            }
        x.sync();
        // the image, out of LDS: header, buckets (16-byte pieces), displacement bytes
        uint32_t* bks = w + LOC_HDR / 4;
        uint8_t* disp = reinterpret_cast<uint8_t*>(bks + 8 * nb);
        if (lane == 0) {
            w[0] = d.lgnb; w[1] = left; w[2] = d.trbeg; w[3] = l;
            if (bad) a.bad[l] = 1;
            if (left) x.atomic_add(a.nleft, (uint64_t)left);
        }
        for (uint32_t i = lane; i < 2 * nb; i += 64) reinterpret_cast<uint4*>(bks)[i] = reinterpret_cast<const uint4*>(sm.bks)[i];
        for (uint32_t i = lane; i < (nb > 16 ? nb : 16u) / 4; i += 64) reinterpret_cast<uint32_t*>(disp)[i] = i < nb / 4 ? reinterpret_cast<const uint32_t*>(sm.disp)[i] : 0u;
    }
}

// An image read from the sidecar file, checked against the index it claims to be a partition of (structure alone lets a
// stale or bit-flipped image send TR k-mers to the wrong counter): every entry's KEY is rebuilt from (bucket, tag, extra bits,
// displacement) and looked up in the plain index as built from the RPGG; the entry must be what body_loc_scatter would have made of the
// index's answer — the locus named by the value (or in its vv list), the same class, the same counter — and vcnt[l] = such entries + the
// keys the image says it left out must equal the locus' key count (body_loc_count: the host compares), so nothing is missing either.
template <class X>
DBTK_HD void body_loc_verify(X& x, const LocBuildArgs& a) {
    for (uint32_t l = x.bid(); l < a.nloci; l += x.nblocks()) {
        const LocusDir d = a.dir[l];
        if (!d.bytes) continue;
        const uint32_t* w = reinterpret_cast<const uint32_t*>(a.arena + 16ull * d.off16);
        const uint32_t ntr = a.trbeg[l + 1] - a.trbeg[l];
        const uint32_t nb = 1u << d.lgnb, mask = nb - 1;
        const uint8_t* disp = reinterpret_cast<const uint8_t*>(w + LOC_HDR / 4 + 8 * nb);
        bool bad = d.trbeg != a.trbeg[l] || w[0] != d.lgnb || w[3] != l;
        if (x.tid() == 0 && w[1]) x.atomic_add(&a.vcnt[l], w[1]);
        const uint32_t nround = (4 * nb + (uint32_t)x.nthreads() - 1) / (uint32_t)x.nthreads() * (uint32_t)x.nthreads();
        for (uint32_t i = (uint32_t)x.tid(); i < nround; i += (uint32_t)x.nthreads()) {
            bool good = false;
            if (i < 4 * nb) {
                const uint32_t b = i >> 2, s2 = i & 3;
                const uint32_t lo = w[LOC_HDR / 4 + 8 * b + s2], p = w[LOC_HDR / 4 + 8 * b + 4 + s2];
                if (p != LOC_EMPTY) {
                    const uint32_t cls = p & (LOC_MULTI | LOC_FLANK | LOC_TR), extra = p >> 24;
                    const uint32_t hx = extra << d.lgnb;  // (the group and the hash see hi only through hi >> lgnb)
                    // bucket = (loc_base(lo, hi) + disp) & mask and loc_base = hi ^ f(lo, hi >> lgnb): the low bits of hi follow
                    const uint32_t hlow = (((b - disp[loc_group(lo, hx, d.lgnb)]) & mask) ^ loc_base(lo, hx, d.lgnb)) & mask;
                    const uint64_t key = ((uint64_t)(hx | hlow) << 32) | lo;
                    const uint64_t va = a.idx ? idx_lookup64_raw(a.idx, a.idx_mask, a.idx_shift, key) : (uint64_t)NOHIT;
                    const uint32_t v = (uint32_t)va, aux = (uint32_t)(va >> 32);
                    bool ok = v != NOHIT && (cls & (LOC_FLANK | LOC_TR)) != (LOC_FLANK | LOC_TR) && cls != 0;
                    if (ok) {
                        bool named = false;
                        if (v & 1) { const uint32_t n = a.vv[v >> 1]; for (uint32_t j = 0; j < n; ++j) named |= a.vv[(v >> 1) + 1 + j] == l; }
                        else named = (v >> 1) == l;
                        uint32_t want;
                        if (v & 1) want = loc_multi_pay(a, key, l);
                        else if (aux == CLS_NONE) want = LOC_MULTI;
                        else if (aux == CLS_FLANK) want = LOC_FLANK;
                        else want = LOC_TR | ((aux - a.trbeg[l]) & LOC_SLOT);
                        ok = named && (p & 0x00FFFFFFu) == want && (!(cls & LOC_TR) || ((p & LOC_SLOT) < ntr && ((v & 1) || aux >= a.trbeg[l])));
                    }
                    good = ok;
                    bad |= !ok;
                }
            }
            const uint64_t m = x.ballot(good);
            if (m && x.lane() == 0) x.atomic_add(&a.vcnt[l], (uint32_t)__builtin_popcountll(m));
        }
        if (bad) a.bad[l] = 1;
    }
}
// Order-independent 64-bit checksum of n 8-byte words (the sidecar's arena as it lies in HBM): sum of mixed (word, position) values.
DBTK_HD uint64_t csum_term(uint64_t w, uint64_t i) {
    uint64_t v = w ^ ((i + 1) * 0xD6E8FEB86659FD93ull);
    v ^= v >> 32; v *= 0x9E3779B97F4A7C15ull; v ^= v >> 29;
    return v;
}
template <class X>
DBTK_HD void body_csum(X& x, const uint64_t* p, uint64_t n, uint64_t* out) {
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)x.bid() * x.nthreads() + x.tid(); i < n; i += (uint64_t)x.nblocks() * x.nthreads()) acc += csum_term(p[i], i);
    if (acc) x.atomic_add(out, acc);
}

// ------------------------------------------------------------------ look-up --
// pay of `km` in the image at `img` (LDS on the device), LOC_MISS when it is not there.  The plain form: every slot of the home
// bucket.  (The kernel's unrolled fast path below does the same with its loads issued together.)
DBTK_HD uint32_t loc_find(const uint32_t* img, uint64_t km) {
    const uint32_t lgnb = img[0], lo = (uint32_t)km, hi = (uint32_t)(km >> 32);
    const uint8_t* disp = reinterpret_cast<const uint8_t*>(img + LOC_HDR / 4 + (8u << lgnb));
    const uint32_t* bk = img + LOC_HDR / 4 + 8 * ((loc_base(lo, hi, lgnb) + disp[loc_group(lo, hi, lgnb)]) & ((1u << lgnb) - 1));
    const uint32_t extra = hi >> lgnb;
    for (int s = 0; s < 4; ++s) {
        const uint32_t p = bk[4 + s];
        if (bk[s] == lo && (p >> 24) == extra && (p & (LOC_FLANK | LOC_TR)) != (LOC_FLANK | LOC_TR)) return p;
    }
    return LOC_MISS;
}

// ------------------------------------------------------------------ items --
// After the survivor sort hist[l] is where the segment of key l ends in the sorted list (body_surv_scatter advanced every key's
// first place past its last pair): the segment of l is [hist[l - 1], hist[l]).  One thread per key: the part of its segment inside
// the chunk [t0, tend) of the list, cut into items of LOC_CH pairs, for the class of workgroup its image fits (the smaller the image, the more workgroups a CU holds).
struct LocItemArgs {
    const uint32_t* hist;     // [nloci + 1] segment ends (key nloci: no locus)
    const uint32_t* nsurv;
    const uint32_t* flag;     // the list is in locus order
    const LocusDir* dir;
    uint32_t nloci, t0, tcap;
    uint32_t cap_bytes[3];    // largest image of each class (0: the class is not in use)
    uint4* items[3];          // {locus, first, end, 0}
    uint32_t* nitems;         // [3], then [3] = entries of `rest`
    uint32_t item_cap;
    uint32_t* rest;           // chunk-relative indices of the pairs no item covers
};
DBTK_HD int loc_class(const LocItemArgs& a, uint32_t l) {
    if (l >= a.nloci) return -1;
    const uint32_t b = a.dir[l].bytes;
    return !b ? -1 : b <= a.cap_bytes[0] ? 0 : b <= a.cap_bytes[1] ? 1 : b <= a.cap_bytes[2] ? 2 : -1;
}
template <class X>
DBTK_HD void body_loc_items(X& x, const LocItemArgs& a) {
    if (!*a.flag) return;
    const uint32_t ns = *a.nsurv;
    const uint32_t tend = ns - a.t0 < a.tcap ? ns : a.t0 + a.tcap;
    // (a wave reserves the places of its 64 keys' items with one atomic per class: 80 000 returning atomics on one counter took 0.6 ms)
    const uint32_t lround = (a.nloci + 63) & ~63u;
    for (uint32_t l = x.bid() * (uint32_t)x.nthreads() + (uint32_t)x.tid(); l < lround; l += x.nblocks() * (uint32_t)x.nthreads()) {
        const int c = loc_class(a, l);
        uint32_t lo = 0, hi = 0, n = 0;
        if (c >= 0) {
            lo = l ? a.hist[l - 1] : 0u; hi = a.hist[l];
            if (lo < a.t0) lo = a.t0;
            if (hi > tend) hi = tend;
            if (lo < hi && hi - lo >= LOC_MIN_PAIRS) n = (hi - lo + LOC_CH - 1) / LOC_CH;
        }
        uint32_t base = 0;
        for (int cc = 0; cc < 3; ++cc) {
            const uint32_t mine = c == cc ? n : 0u;
            const uint32_t ex = x.wave_excl_scan(mine);
            const uint32_t tot = x.bcast(ex + mine, 63);
            if (!tot) continue;  // (uniform)
            uint32_t b = 0;
            if (x.lane() == 0) b = x.atomic_add(&a.nitems[cc], tot);
            b = x.bcast(b, 0);
            if (c == cc) base = b + ex;
        }
        const uint32_t per = n ? (hi - lo + n - 1) / n : 0u;  // (items of equal size: 70 pairs are 35 + 35, not 64 + 6 — every item pays for the image)
        for (uint32_t j = 0; j < n && base + j < a.item_cap; ++j) {
            const uint32_t f = lo + j * per;
            a.items[c][base + j] = uint4{l, f, f + per < hi ? f + per : hi, 0u};
        }
    }
}
// the pairs of the chunk whose key has no image (or no key at all): their indices, for the global-table kernel.  One thread per pair;
// a wave appends its pairs together, in list order.  With the list not in locus order (a batch with few survivors per locus): all of them.
template <class X>
DBTK_HD void body_loc_rest(X& x, const LocItemArgs& a) {
    const uint32_t ns = *a.nsurv;
    const uint32_t tend = ns - a.t0 < a.tcap ? ns : a.t0 + a.tcap;
    const uint32_t n = ns > a.t0 ? tend - a.t0 : 0;
    const bool sorted = *a.flag != 0;
    const uint32_t nround = (n + 63) & ~63u;
    for (uint32_t i = x.bid() * (uint32_t)x.nthreads() + (uint32_t)x.tid(); i < nround; i += x.nblocks() * (uint32_t)x.nthreads()) {
        bool mine = i < n;
        if (mine && sorted) {
            const uint32_t t = a.t0 + i;
            uint32_t lo = 0, hi = a.nloci + 1;  // first key whose segment ends behind t
            while (lo < hi) {
                const uint32_t mid = lo + (hi - lo) / 2;
                if (a.hist[mid] > t) hi = mid; else lo = mid + 1;
            }
            mine = loc_class(a, lo) < 0;
            if (!mine) {  // (as body_loc_items decides: the segment's part inside the chunk)
                uint32_t s0 = lo ? a.hist[lo - 1] : 0u, s1 = a.hist[lo];
                if (s0 < a.t0) s0 = a.t0;
                if (s1 > tend) s1 = tend;
                mine = s1 - s0 < LOC_MIN_PAIRS;
            }
        }
        const uint64_t m = x.ballot(mine);
        if (!m) continue;
        uint32_t base = 0;
        if (x.lane() == 0) base = x.atomic_add(&a.nitems[3], (uint32_t)__builtin_popcountll(m));
        base = x.bcast(base, 0);
        if (mine) a.rest[base + (uint32_t)__builtin_popcountll(m & ((1ull << x.lane()) - 1))] = i;
    }
}

// ------------------------------------------------------------------ split --
// Which items each workgroup of a class takes: a contiguous range of the class' item list, the ranges cut where the running sum of the
// items' weights (pairs + a fixed part for the image) passes the multiples of total / workgroups.  (Item i, i + S, i + 2 S ... per
// workgroup left the slowest of ~1000 workgroups 10-15 % behind the mean: 40-80 items of 16-64 pairs each do not average out.)
// One workgroup per class; starts[c][j] .. starts[c][j + 1] are workgroup j's items.
struct LocSplitArgs {
    const uint4* items[3];
    const uint32_t* nitems;  // [3]
    uint32_t item_cap;
    uint32_t nblk[3];        // workgroups of the class' launch (0: class not launched)
    uint32_t wfix[3];        // an item's fixed cost, in pairs
    uint32_t* starts[3];     // [nblk + 1]
    uint64_t* stats;         // nullptr, or the context's path statistics (dbtk.h: dbtk_ctx_path_stats): [c] += items of class c, [3 + c] += their pairs
};
struct LocSplitSmem { uint32_t wt[64]; };
template <class X>
DBTK_HD void body_loc_split(X& x, const LocSplitArgs& a) {
    LocSplitSmem& sm = *x.template smem<LocSplitSmem>();
    const uint32_t c = x.bid();
    if (c >= 3 || !a.nblk[c]) return;
    const uint32_t B = a.nblk[c], NT = (uint32_t)x.nthreads(), tid = (uint32_t)x.tid(), nwv = NT / 64, wave = tid >> 6;
    const uint32_t n = a.nitems[c] < a.item_cap ? a.nitems[c] : a.item_cap;
    const uint4* it = a.items[c];
    uint32_t* st = a.starts[c];
    const uint32_t lo = (uint32_t)((uint64_t)n * tid / NT), hi = (uint32_t)((uint64_t)n * (tid + 1) / NT);
    // (eight items' loads in flight at a time: one dependent load after the other made this kernel 80 us)
    auto weights8 = [&](uint32_t i0, uint32_t (&w)[8]) {
        uint4 d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) d[u] = it[i0 + u < hi ? i0 + u : lo];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = i0 + u < hi ? d[u].z - d[u].y + a.wfix[c] : 0u;
    };
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i += 8) {
        uint32_t w[8];
        weights8(i, w);
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += w[u];
    }
    const uint32_t ex = x.wave_excl_scan(sum);
    if (x.lane() == 63) sm.wt[wave] = ex + sum;
    x.bsync();
    uint32_t base = 0, P = 0;
    for (uint32_t w = 0; w < nwv; ++w) { const uint32_t v = sm.wt[w]; if (w < wave) base += v; P += v; }
    if (a.stats && tid == 0 && n) { x.atomic_add(&a.stats[c], (uint64_t)n); x.atomic_add(&a.stats[3 + c], (uint64_t)P - (uint64_t)n * a.wfix[c]); }
    if (!P) {  // no items: every workgroup's range is empty
        for (uint32_t j = tid; j <= B; j += NT) st[j] = 0;
        return;
    }
    if (tid == 0) st[0] = 0;
    uint64_t e = (uint64_t)base + ex;  // weight before item i
    uint64_t j = e * B / P + 1;        // the first cut point j * P / B behind e
    for (uint32_t i = lo; i < hi; i += 8) {
        uint32_t w[8];
        weights8(i, w);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i + u >= hi) break;
            e += w[u];
            // the workgroups j whose cut point lies in (weight before the item, weight with it]: they start behind this item
            for (; j <= B && j * P <= e * B; ++j) st[j] = i + u + 1;
        }
    }
}

// ------------------------------------------------------------------ the kernel --
// A workgroup of NW waves per item.  Per item a wave owns the pairs q = wave, wave + NW, ... (at most LOC_CH / NW) and works in
// three phases, so that the latency of what goes to the global index is paid once per item and not once per pair:
//   1  per pair: 2-bit pack, canonical k-mers, one bucket of the image per position (LDS); the positions the image answers go to
//      the hit buffers at once (`aux`, 16-byte stores: the values of such positions are all 2 * locus and need not travel);
//      every other position's k-mer is QUEUED in LDS and stands as "not in the index" for now;
//   2  the queue against the plain index, 16 look-ups per step, several steps in flight (the four lanes of a quad read the four
//      16-byte parts of one bucket: one request per look-up); a k-mer found there (one the image left out, a k-mer of another locus,
//      a key shared between loci) is PATCHED into its row of the hit buffers, and the row's statistics follow;
//   3  the per-read headers (found positions, the one index value or not); the rare read whose found k-mers do not all have one
//      index value gets its value row (every position the image answered: 2 * locus; the patched ones already hold theirs).
//
// FUSE (round 5): the kernel also RESOLVES the usual pair — both mates pass kfilter, every found k-mer unique to this one locus
// (dbtk_kernels.h: body_pair_usual; countHit's shortcut AQ.cpp:354-357, 439-451, the QC gate :2059-2062, assignTRkmc :1450-1556,
// accumulate :2146-2158) — from the pay words it has just read from the image, instead of writing 4 bytes per position to the hit
// buffers for a second kernel to read back: such a pair writes NO row.  Whether a pair is usual is known only when the queue has
// been looked up (a k-mer the image does not hold may still be another locus'), and that happens once per item; so the pair is
// resolved AHEAD of it, as if every queued k-mer were absent from the index (a k-mer with a sequencing error: nearly always so):
// its TR k-mers are counted at once into an LDS copy of the locus' counters (two 16-bit counts per word, flushed once per item: one
// atomic per touched counter instead of one per k-mer of every pair), what else it adds (kmc, nmapread, the counters, walk_dst)
// waits in a row of LDS.  After the look-ups a row none of whose queued k-mers was found is COMMITTED; one that had a k-mer found
// (rare) is taken back: the pair is looked up again, its counts subtracted, and it goes the general way — rows in the hit buffers
// and an entry of gen_list for body_pair, like every pair with a shared k-mer or too few hits.  assignTRkmc's scan runs on the two
// half-waves at once (a mate each) as wave scans over the lanes' NPL consecutive positions — no scalar state machine, no ballots.
#ifndef DBTK_LOC_Q
#define DBTK_LOC_Q 128
#endif
constexpr int LOC_Q = DBTK_LOC_Q;              // queue entries per wave (the test emulator builds with a short queue: look-ups in the middle of a pair)
constexpr int LOC_ROWS = 2 * (int)LOC_CH / 4;  // rows (mates) of one wave's pairs of an item, for workgroups of at least 4 waves
constexpr uint32_t LOC_FCNT = 2048;            // counters of the item's locus kept in LDS (a locus with more TR k-mers counts the rest directly)
// what a pair resolved ahead of the look-ups still has to add
struct LocSpecRow {
    uint32_t i;      // the pair's place in the chunk
    uint32_t flags;  // bit 0: one of its queued k-mers WAS in the index (set by the look-ups); bits 1-2: mate removed by assignTRkmc; bit 3: the
                     // pair was resolved ahead (0: it wrote rows); bits 4-6: stage
    uint32_t kmc;    // (ei1 - si1) + (ei2 - si2)
    uint32_t nks;    // k-mers of both mates
    uint32_t inc;    // count increments
    uint32_t pad;    // vv words of fillstats (pairs with k-mers shared between loci)
};
constexpr uint32_t LSP_FAILED = 1u, LSP_RM0 = 2u, LSP_RM1 = 4u, LSP_SPEC = 8u, LSP_STAGE_SHIFT = 4;
constexpr uint32_t LSP_COUNT = 0, LSP_QC = 1, LSP_THREAD_HEAD = 2, LSP_THREAD_V13 = 3, LSP_EXTRACT = 4;
template <int NPL, bool FUSE>
struct __attribute__((aligned(16))) LocWaveSmemT {
    uint32_t pk[2][20];             // 2-bit stream of each mate from its 4-byte-aligned start
    uint16_t vd[2][40];             // validity bits (only for a pair with a non-ACGT byte)
    uint32_t res[2][32 * NPL];      // [mate][position]: aux of the pair being looked up, on its way to 16-byte stores
    uint64_t qkm[LOC_Q];            // queue: k-mers the image could not answer
    uint32_t qcode[LOC_Q];          //   row << 8 | position
    uint32_t stat[LOC_ROWS][4];     // per row (local pair << 1 | mate): found positions, largest and smallest index value, hit-buffer row
    uint32_t patched[LOC_ROWS][NPL];  // per row: positions whose value came from the index (bit p of word p / 32)
    LocSpecRow spec[FUSE ? LOC_ROWS / 2 : 1];  // per pair of the wave in the item (rows 2q, 2q + 1): what a pair resolved ahead of the look-ups still has to add
    uint32_t ctr[12];               // (FUSE) this wave's share of the counters, flushed at the end of the kernel
};
template <int NPL, int NW, int IMGB, bool FUSE>
struct __attribute__((aligned(16))) LocSmemT {
    uint4 img[IMGB / 16];
    uint32_t tally[4];  // of the item: pairs looked up so far, pairs of them handed back (LOC_BAIL); [2], [3]: reads assigned / kmc of the item's committed pairs
    uint32_t fcnt[FUSE ? LOC_FCNT / 2 : 4];  // the item's count increments, two 16-bit counts per word (an item adds < 2^16 to any counter: 64 pairs x 260)
    LocWaveSmemT<NPL, FUSE> w[NW];
};
struct LocRunArgs {
    const LocusDir* dir;
    const uint8_t* arena;
    const uint4* items;
    const uint32_t* nitems;
    uint32_t* rest;       // the probe kernel only: the list of the pairs left to the global-table kernel, which this kernel appends to:
    uint32_t* nrest;      //   a pair most of whose k-mers are NOT its locus' (below) is better off there
    const uint32_t* starts;  // [workgroups + 1] the workgroups' ranges of the item list (body_loc_split)
};
// A pair is handed back when more than this many of its positions miss the image: a read pair that merely touches the locus (a
// stretch of genome that resembles it: it passed subfilter on one k-mer) would send nearly all its positions to the plain index one
// by one, where the global-table kernel fetches a bucket per RUN of positions.  (A pair from the locus with an error or two: ~20-40.)
constexpr uint32_t LOC_BAIL = 96;

// (assign_halves — assignTRkmc for the two mates of a pair at once, a mate per half-wave — is in dbtk_assign.h)

template <int NPL, int NW, int IMGB, bool FUSE, class X>
DBTK_HD void body_probe_locus(X& x, const BatchArgs& a, const LocRunArgs& r) {
    typedef LocSmemT<NPL, NW, IMGB, FUSE> SM;
    static_assert(NW >= 4 && 2 * ((int)LOC_CH / NW) <= LOC_ROWS, "rows of a wave's pairs of one item");
    static_assert((uint32_t)LOC_CH * 2 * 32 * NPL < 65536, "an item's increments of one counter fit 16 bits");
    constexpr int IPT = ((IMGB - (int)LOC_HDR) / 16 + NW * 64 - 1) / (NW * 64);  // 16-byte pieces of an image per thread
    SM& smb = *x.template smem<SM>();
    const int lane = x.lane();
    const uint32_t wave = (uint32_t)x.tid() >> 6;
    LocWaveSmemT<NPL, FUSE>& sm = smb.w[wave];
    const uint32_t* bks = reinterpret_cast<const uint32_t*>(smb.img);  // (the image's buckets; its header stays in HBM: the directory says as much)
    const uint32_t hl = (uint32_t)lane & 31u, half = (uint32_t)lane >> 5;
    const DevTables& T = a.T;
    const uint32_t k = T.ksize, cth = a.P.cthreshold;
    const uint32_t ifirst = r.starts[x.bid()], nitems = r.starts[x.bid() + 1];  // this workgroup's items: [ifirst, nitems)
    const uint32_t lmax = 32u * NPL + k - 1;  // bases the lanes of a half cover (the launcher promised no read is longer)
    const uint64_t kmask = (1ull << (2 * k)) - 1;
    const uint32_t p0 = hl * NPL;
    const uint32_t sub = (uint32_t)lane & 3u, qd = (uint32_t)lane >> 2;
    constexpr uint32_t S = 1;
    // This wave's pairs, item after item: pair q of an item is the wave's when q % NW == wave.  The fetch pipeline runs along that
    // sequence ACROSS items (while the workgroup waits for an image, the reads of its first pairs are already on their way).  All loads
    // unconditional, clamped to something valid, so that they stay in flight.
    struct Cur { uint32_t it; uint32_t i, end; };  // item, pair (index into the sorted list), the item's end
    auto desc = [&](uint32_t it) -> uint4 { return r.items[it < nitems ? it : 0u]; };
    auto seek = [&](uint32_t it) -> Cur {  // first pair of the wave at or after item `it`
        for (;;) {
            if (it >= nitems) return Cur{it, 0u, 0u};
            const uint4 d = desc(it);
            if (d.y + wave < d.z) return Cur{it, d.y + wave, d.z};
            it += S;
        }
    };
    auto next = [&](const Cur& c) -> Cur {
        if (c.it >= nitems) return c;
        if (c.i + NW < c.end) return Cur{c.it, c.i + NW, c.end};
        return seek(c.it + S);
    };
    uint32_t rw0 = 0, rw1 = 0;
    uint64_t o0C = 0, o1C = 0, o0B = 0, o1B = 0;
    uint32_t pairA = 0;
    auto fetch_bytes = [&](uint64_t o0, uint64_t o1) {
        uint32_t len = (uint32_t)(o1 - o0);
        if (len > lmax) len = lmax;
        const uint64_t a0 = o0 & ~3ull;
        const uint32_t nw = ((uint32_t)(o0 - a0) + len + 3) >> 2;
        rw0 = *reinterpret_cast<const uint32_t*>(a.seq + (2 * hl < nw ? a0 + 8ull * hl : 0ull));
        rw1 = *reinterpret_cast<const uint32_t*>(a.seq + (2 * hl + 1 < nw ? a0 + 8ull * hl + 4 : 0ull));
    };
    auto fetch_offsets = [&](uint32_t pair) {
        const uint64_t rr = 2 * (uint64_t)pair + half;
        o0B = a.off[rr]; o1B = a.off[rr + 1];
    };
    auto surv_at = [&](const Cur& c) { return a.surv[c.it < nitems ? c.i : a.t0]; };
    Cur cC = seek(ifirst), cB = next(cC), cA = next(cB);  // pair being looked up; the one whose offsets / whose list entry are in flight
    if (cC.it < nitems) {
        fetch_offsets(x.uni(surv_at(cC)));
        o0C = o0B; o1C = o1B;
        fetch_bytes(o0C, o1C);
        if (cB.it < nitems) fetch_offsets(x.uni(surv_at(cB)));
        pairA = surv_at(cA);
    }
    if (FUSE) {
        for (uint32_t e = (uint32_t)x.tid(); e < LOC_FCNT / 2; e += (uint32_t)x.nthreads()) smb.fcnt[e] = 0;
        if (lane < 12) sm.ctr[lane] = 0;
    }
    if (x.tid() < 4) smb.tally[x.tid()] = 0;
    // ---- phase 2: the queue against the plain index; what is found is patched into the hit buffers
    uint32_t qn = 0;  // queue entries (uniform)
    auto settle = [&](bool active, uint32_t code, uint64_t kq, uint64_t q0, uint64_t q1, bool& more) -> uint64_t {
        const uint64_t v0 = quad_perm64<2, 3, 2, 3>(x, q0), v1 = quad_perm64<2, 3, 2, 3>(x, q1);
        const bool keyl = sub < 2;
        const bool m0 = active && keyl && q0 == kq, m1 = active && keyl && (q1 & ~IDX_OVF) == kq;
        const bool ovf = active && sub == 1 && q1 != NAN64 && (q1 & IDX_OVF);
        const uint64_t hmask = x.ballot(m0 || m1), omask = x.ballot(ovf);
        const bool qhit = ((hmask >> (lane & ~3u)) & 0xF) != 0;
        more = active && !qhit && ((omask >> ((lane & ~3u) + 1)) & 1);
        if (m0 || m1) {
            const uint32_t row = code >> 8;
            if (FUSE && (sm.spec[row >> 1].flags & LSP_SPEC)) x.lds_or(&sm.spec[row >> 1].flags, LSP_FAILED);  // the pair was NOT usual after all: taken back below
            else {
                const uint64_t v = m0 ? v0 : v1;
                const uint32_t pos = code & 0xFFu, val = (uint32_t)v;
                const size_t at = (size_t)sm.stat[row][3] * a.nkp + pos;
                a.hitaux[at] = (uint32_t)(v >> 32);
                a.hitval[at] = val;
                x.lds_add(&sm.stat[row][0], 1u);
                x.lds_max(&sm.stat[row][1], val);
                x.lds_min(&sm.stat[row][2], val);
                x.lds_or(&sm.patched[row][pos >> 5], 1u << (pos & 31));
            }
        }
        return x.ballot(more);
    };
    auto flush = [&]() {
        x.sync();
        constexpr int NB = 4;
        for (uint32_t i0 = 0; i0 < qn; i0 += 16 * NB) {
            uint32_t bq[NB], cd[NB];
            uint64_t kq[NB], q0[NB], q1[NB];
            bool actv[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const uint32_t ii = i0 + 16 * u + qd;
                actv[u] = ii < qn;
                const uint32_t ic = actv[u] ? ii : 0u;
                kq[u] = sm.qkm[ic]; cd[u] = sm.qcode[ic];
                bq[u] = actv[u] ? (uint32_t)hash_idx(kq[u], T.idx_shift) : 0u;
                bucket_part(T.idx, bq[u], sub, &q0[u], &q1[u]);
            }
            uint64_t anymore = 0;
            bool more[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                more[u] = false;
                if (i0 + 16 * u < qn) anymore |= settle(actv[u], cd[u], kq[u], q0[u], q1[u], more[u]);
            }
            if (anymore) {  // (about one look-up in a thousand: the next bucket)
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    bool mo = more[u];
                    uint32_t b = bq[u];
                    while (x.ballot(mo)) {
                        b = (b + 1) & (uint32_t)T.idx_mask;
                        uint64_t n0 = 0, n1 = 0;
                        if (mo) bucket_part(T.idx, b, sub, &n0, &n1);
                        bool m2 = false;
                        (void)settle(mo, cd[u], kq[u], n0, n1, m2);
                        mo = m2;
                    }
                }
            }
        }
        qn = 0;
        x.sync();
    };
    // (the descriptor and the directory entry of the workgroup's next items are fetched an item ahead: three dependent round trips
    // in front of every image otherwise)
    uint4 d1 = desc(ifirst), d2 = desc(ifirst + S);
    LocusDir ld1 = r.dir[x.uni(d1.x) < T.nloci ? x.uni(d1.x) : 0u];
    uint32_t trb_prev = 0, locus_prev = 0;
    // the item's counts: LDS -> the locus' counters, kmc and nmapread (one add per counter the item's pairs touched)
    auto flush_counts = [&]() {
        if (!FUSE) return;
        for (uint32_t e = (uint32_t)x.tid(); e < LOC_FCNT / 2; e += (uint32_t)x.nthreads()) {
            const uint32_t v = smb.fcnt[e];
            if (v) {
                if (v & 0xFFFFu) x.atomic_add(&a.counts[trb_prev + 2 * e], (uint64_t)(v & 0xFFFFu));
                if (v >> 16) x.atomic_add(&a.counts[trb_prev + 2 * e + 1], (uint64_t)(v >> 16));
                smb.fcnt[e] = 0;
            }
        }
        if (x.tid() == 0) {  // (between two barriers: the waves add to these when they commit their pairs of an item)
            const uint32_t na = smb.tally[2], km = smb.tally[3];
            if (na) x.atomic_add(&a.nmapread[locus_prev], (uint64_t)na);
            if (km) x.atomic_add(&a.kmc[locus_prev], (uint64_t)km);
            smb.tally[2] = 0; smb.tally[3] = 0;
        }
    };
    DBTK_STAMP_DECL
    for (uint32_t item = ifirst; item < nitems; item += S) {
        DBTK_STAMP(42);  // headers of the item before
        const uint4 d = d1;
        const LocusDir ld = ld1;
        const uint32_t locus = x.uni(d.x), lgnb = x.uni(ld.lgnb), trb = x.uni(ld.trbeg);
        const uint8_t* dsp = reinterpret_cast<const uint8_t*>(bks + (8u << lgnb));
        x.bsync();  // every wave is done with the image of the item before, and with its counts
        flush_counts();
        trb_prev = trb; locus_prev = locus;
        {
            const p2_v4u* src = reinterpret_cast<const p2_v4u*>(r.arena + 16ull * ld.off16 + LOC_HDR);
            const uint32_t n16 = (ld.bytes - LOC_HDR) / 16;
            p2_v4u t[IPT];
#pragma unroll
            for (int u = 0; u < IPT; ++u) {
                const uint32_t o = (uint32_t)x.tid() + (uint32_t)u * NW * 64;
                t[u] = src[o < n16 ? o : 0u];
            }
#pragma unroll
            for (int u = 0; u < IPT; ++u) {
                const uint32_t o = (uint32_t)x.tid() + (uint32_t)u * NW * 64;
                if (o < n16) *reinterpret_cast<p2_v4u*>(&smb.img[o]) = t[u];
            }
        }
        d1 = d2; d2 = desc(item + 2 * S);
        ld1 = r.dir[x.uni(d1.x) < T.nloci ? x.uni(d1.x) : 0u];
        // the QC gate of the item's locus (AQ.cpp:2059-2062)
        const bool qc_off = FUSE && a.P.qc && T.qc && !T.qc[locus];
        if (x.tid() < 2) smb.tally[x.tid()] = 0;
        x.bsync();
        uint32_t nrow = 0;   // rows of this wave in the item so far (uniform): rows 2q, 2q + 1 = the wave's q-th pair of the item
        uint64_t redo = 0;   // pairs (q) resolved ahead that have to be taken back (uniform)
        DBTK_STAMP(43);  // barriers, image into LDS, the next one's loads issued
        for (;;) {
            bool again = false;  // this turn takes a resolved pair back
            uint32_t i, dw0, dw1, rmbits = 0;
            uint64_t o0, o1;
            if (cC.it == item) {
                i = cC.i - a.t0;  // pair of the chunk: hit-buffer rows 2i, 2i + 1
                o0 = o0C; o1 = o1C;
                dw0 = rw0; dw1 = rw1;
                // advance the pipeline
                cC = cB; cB = cA; cA = next(cA);
                const bool hasC = cC.it < nitems, hasB = cB.it < nitems;
                o0C = hasC ? o0B : 0ull; o1C = hasC ? o1B : 0ull;
                fetch_bytes(o0C, o1C);
                fetch_offsets(hasB ? x.uni(pairA) : x.uni(pairA) * 0u);
                pairA = surv_at(cA);
            } else {
                // ---- the wave's pairs of this item are through: what the index says about the queued k-mers, then the rows' headers,
                // the general kernel's list, and the pairs resolved ahead: committed, or marked to be taken back
                if (qn) flush();
                DBTK_STAMP(41);  // the queue against the index
                x.sync();
                if (nrow) {
                    bool genv = false, isrow = false, fail = false;
                    uint32_t hrow = 0, asg = 0, km = 0;
                    const uint32_t np = nrow >> 1;
                    if ((uint32_t)lane < nrow) isrow = !FUSE || !(sm.spec[lane >> 1].flags & LSP_SPEC);
                    if (isrow) {
                        const uint32_t nh = sm.stat[lane][0], vmx = sm.stat[lane][1], vmn = sm.stat[lane][2];
                        hrow = sm.stat[lane][3];
#if !defined(__HIPCC__) && defined(DBTK_LOC_DEBUG)
                        if (hrow >= 2 * a.tcap) { fprintf(stderr, "bad hrow %u lane %d nrow %u flags %x item %u wave %u\n", hrow, lane, nrow, sm.spec[lane >> 1].flags, item, wave); abort(); }
#endif
                        const bool uniform = T.consistent && nh && vmx == vmn && !(vmx & 1u);
                        a.hithdr[hrow] = (uint64_t)(nh ? vmx : NOHIT) | ((uint64_t)nh << 32) | (uniform ? HDR_UNIFORM : 0ull);
                        genv = !uniform;
                    }
                    uint64_t gm = x.ballot(genv);
                    while (gm) {  // (rare) a read whose found k-mers do not share one index value: its value row
                        const uint32_t rr = (uint32_t)__builtin_ctzll(gm);
                        gm &= gm - 1;
                        const uint32_t hr = x.bcast(hrow, (int)rr), nk = a.hitnk[hr];
                        for (uint32_t pos = (uint32_t)lane; pos < ((nk + 3) & ~3u) && pos < 32u * NPL; pos += 64) {
                            const size_t at = (size_t)hr * a.nkp + pos;
                            if (!((sm.patched[rr][pos >> 5] >> (pos & 31)) & 1)) a.hitval[at] = a.hitaux[at] == AUX_MISS ? NOHIT : locus << 1;
                        }
                    }
                    if (FUSE) {
                        // lane q: the wave's q-th pair of the item.  One that wrote rows is the general kernel's (body_pair decides it from
                        // its rows); one resolved ahead is committed unless a queued k-mer of it was found in the index
                        bool gen = false;
                        if ((uint32_t)lane < np) {
                            const LocSpecRow sr = sm.spec[lane];
                            gen = !(sr.flags & LSP_SPEC);
                            fail = !gen && (sr.flags & LSP_FAILED) != 0;
                            if (!gen && !fail) {
                                const uint32_t stage = (sr.flags >> LSP_STAGE_SHIFT) & 7u;
                                const uint32_t nrm = ((sr.flags >> 1) & 1u) + ((sr.flags >> 2) & 1u);
                                // [0] qc, [1] threading, [2] feasible, [3] asgn, [4] cls, [5] inc, [6] nhash1, [7] pairs, [8] pairs taken back
                                x.lds_add(&sm.ctr[6], sr.nks);
                                x.lds_add(&sm.ctr[7], 1u);
                                if (sr.pad) { x.lds_add(&sm.ctr[9], sr.pad); x.lds_add(&sm.ctr[10], 1u); }
                                if (stage == LSP_QC) x.lds_add(&sm.ctr[0], 2u);
                                else {
                                    x.lds_add(&sm.ctr[1], 2u);
                                    if (stage == LSP_THREAD_V13) a.walk_dst[a.t0 + sr.i] = locus;
                                    if (stage == LSP_EXTRACT) x.lds_add(&sm.ctr[2], 2u);
                                    if (stage == LSP_COUNT) {
                                        x.lds_add(&sm.ctr[2], 2u);
                                        x.lds_add(&sm.ctr[4], sr.nks);
                                        if (nrm < 2) { asg = 2 - nrm; km = sr.kmc; x.lds_add(&sm.ctr[3], asg); x.lds_add(&sm.ctr[5], sr.inc); }
                                    }
                                }
                            }
                        }
                        const uint64_t gmask = x.ballot(gen);
                        if (gmask) {
                            uint32_t base = 0;
                            if (lane == 0) base = x.atomic_add(a.ngen, (uint32_t)__builtin_popcountll(gmask));
                            base = x.bcast(base, 0);
                            if (gen) a.gen_list[base + (uint32_t)__builtin_popcountll(gmask & ((1ull << lane) - 1))] = a.t0 + (sm.stat[2 * lane][3] >> 1);
                        }
                        if (x.ballot(asg != 0)) {
                            const uint32_t sa = x.wave_sum(asg), sk = x.wave_sum(km);
                            if (lane == 0) { x.lds_add(&smb.tally[2], sa); x.lds_add(&smb.tally[3], sk); }
                        }
                        redo |= x.ballot(fail);  // (none in a turn that takes pairs back: those write rows)
                    }
                    nrow = 0;
                }
                if (!FUSE || !redo) break;
                // ---- (rare) a pair resolved ahead that has a k-mer of the index outside the image: its counts are subtracted and it
                // goes the general way, like a pair with a shared k-mer (lowest q first)
                const uint32_t rr = (uint32_t)__builtin_ctzll(redo);
                i = x.uni(sm.spec[rr].i);
                rmbits = x.uni(sm.spec[rr].flags);
                x.sync();
                redo &= redo - 1;  // (this turn's rows are 0 and 1, its entry spec[0]: of a pair already committed, or this very one)
                again = true;
                const uint64_t rd = 2 * (uint64_t)x.uni(a.surv[a.t0 + i]) + half;
                o0 = a.off[rd]; o1 = a.off[rd + 1];
                {
                    uint32_t len = (uint32_t)(o1 - o0);
                    if (len > lmax) len = lmax;
                    const uint64_t a0 = o0 & ~3ull;
                    const uint32_t nw = ((uint32_t)(o0 - a0) + len + 3) >> 2;
                    dw0 = *reinterpret_cast<const uint32_t*>(a.seq + (2 * hl < nw ? a0 + 8ull * hl : 0ull));
                    dw1 = *reinterpret_cast<const uint32_t*>(a.seq + (2 * hl + 1 < nw ? a0 + 8ull * hl + 4 : 0ull));
                }
                if (lane == 0) x.lds_add(&sm.ctr[8], 1u);
            }
            uint32_t len = (uint32_t)(o1 - o0);
            if (len > lmax) { *a.errflag = DBTK_ERR_READ_TOO_LONG; len = lmax; }
            const uint64_t a0 = o0 & ~3ull;
            const uint32_t rsh = (uint32_t)(o0 - a0);
            x.sync();  // the previous pair's LDS is dead
            if (r.rest && !again) {
                // an item most of whose pairs so far were handed back is one of reads that merely resemble the locus: the rest of it goes
                // the same way at once (which pairs that catches depends on the waves' timing; the results do not)
                const uint32_t done = x.bcast(smb.tally[0], 0), gone = x.bcast(smb.tally[1], 0);  // (lane 0's reading: the other waves are adding)
                if (done >= (uint32_t)NW && 4 * gone >= 3 * done) {
                    if (lane == 0) r.rest[x.atomic_add(r.nrest, 1u)] = i;
                    continue;
                }
            }
            uint32_t bad = 0;
            {
                const uint32_t c0 = pack4_b2(dw0, &bad), c1 = pack4_b2(dw1, &bad);
                reinterpret_cast<uint16_t*>(sm.pk[half])[hl ^ 1u] = (uint16_t)(((c0 >> 8) & 0xFF00u) | ((c1 >> 16) & 0xFFu));
            }
            const uint32_t nk = len >= k ? len - k + 1 : 0;
            const bool clean = x.ballot(bad != 0 && 8 * hl < rsh + len) == 0;
            const uint32_t row = nrow + half, hrow = 2 * i + half;
            DBTK_STAMP(40);  // fetch pipeline, pack
            uint64_t km[NPL];
            bool pend[NPL];
            uint32_t npend = 0, nres = 0;
            uint32_t stbits = 0;  // (FUSE) bit j: the lane's j-th position is known at the locus (flank or TR), bit 8 + j: it is a TR k-mer,
                                  // bit 16 + j: its k-mer is SHARED between loci (known all the same: the image carries its class here)
            bool multi = false;
            if (clean) {
                x.sync();
                const uint64_t W = window_fw_clean(sm.pk[half], rsh + p0, 32);
                const uint64_t RW = revcomp2(W, 32);
                uint32_t bo[NPL];
                uint4 tg[NPL];
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    const uint64_t fw = (W >> (2 * (32 - k - j))) & kmask, rc = (RW >> (2 * j)) & kmask;
                    km[j] = fw < rc ? fw : rc;
                    bo[j] = dsp[loc_group((uint32_t)km[j], (uint32_t)(km[j] >> 32), lgnb)];  // (the groups' displacement bytes: reads issued together)
                }
#pragma unroll
                for (int j = 0; j < NPL; ++j) bo[j] = 8 * ((loc_base((uint32_t)km[j], (uint32_t)(km[j] >> 32), lgnb) + bo[j]) & ((1u << lgnb) - 1));
#pragma unroll
                for (int j = 0; j < NPL; ++j) tg[j] = *reinterpret_cast<const uint4*>(bks + bo[j]);  // (the tags of the lane's positions: reads issued together)
                // the slot whose tag matches (the last of them, should two match), then ITS pay word: one more 4-byte read instead of the
                // bucket's four (registers)
                uint32_t pay[NPL];
                bool any[NPL];
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    const uint32_t lo = (uint32_t)km[j];
                    uint32_t sl = 0;
                    any[j] = false;
                    if (tg[j].x == lo) { sl = 0; any[j] = true; }
                    if (tg[j].y == lo) { sl = 1; any[j] = true; }
                    if (tg[j].z == lo) { sl = 2; any[j] = true; }
                    if (tg[j].w == lo) { sl = 3; any[j] = true; }
                    pay[j] = bks[bo[j] + 4 + sl];
                }
                bool slow = false;
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    const uint32_t extra = (uint32_t)(km[j] >> 32) >> lgnb, p = pay[j];
                    const bool act = p0 + j < nk;
                    const bool ok = any[j] && (p >> 24) == extra && (p & (LOC_FLANK | LOC_TR)) != (LOC_FLANK | LOC_TR);
                    const bool agn = act && !ok && any[j];  // a tag matched but not its entry (another slot of the bucket may): the plain search
                    slow |= agn;
                    pay[j] = !act ? LOC_EMPTY : ok ? p : agn ? LOC_EMPTY - 1 : LOC_MISS;
                }
                if (x.ballot(slow)) {  // (rare)
#pragma unroll
                    for (int j = 0; j < NPL; ++j)
                        if (pay[j] == LOC_EMPTY - 1) {
                            const uint32_t lo = (uint32_t)km[j], extra = (uint32_t)(km[j] >> 32) >> lgnb;
                            const uint32_t* bk = bks + bo[j];
                            uint32_t p = LOC_MISS;
                            for (int s2 = 0; s2 < 4; ++s2) {
                                const uint32_t q = bk[4 + s2];
                                if (bk[s2] == lo && (q >> 24) == extra && (q & (LOC_FLANK | LOC_TR)) != (LOC_FLANK | LOC_TR)) p = q;
                            }
                            pay[j] = p;
                        }
                }
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    const bool act = p0 + j < nk;
                    const bool inimg = act && pay[j] != LOC_MISS;
                    const bool shared = inimg && (pay[j] & LOC_MULTI);                       // a key shared between loci: its index value is a vv list
                    const bool classed = shared && (pay[j] & (LOC_FLANK | LOC_TR)) != 0;     // ... whose class at THIS locus the image carries
                    pend[j] = act && (pay[j] == LOC_MISS || shared);
                    multi |= shared && !(FUSE && classed);
                    npend += pend[j] ? 1u : 0u;
                    const bool fnd = act && !pend[j];
                    nres += fnd ? 1u : 0u;
                    const bool known = fnd || (FUSE && classed);  // the position's class at the locus is known from the image
                    sm.res[half][p0 + j] = !known ? AUX_MISS : (pay[j] & LOC_FLANK) ? CLS_FLANK : trb + (pay[j] & LOC_SLOT);
                    if (FUSE) stbits |= (known ? 1u : 0u) << j | (known && (pay[j] & LOC_TR) ? 1u : 0u) << (8 + j) | (classed ? 1u : 0u) << (16 + j);
                }
            } else {
                // (rare) a non-ACGT byte somewhere in the pair: exact validity bits; a position with an invalid window is no k-mer,
                // every other one is looked up in the index
                uint32_t vb = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t sidx = 8 * hl + e, by = ((e < 4 ? dw0 : dw1) >> (8 * (e & 3))) & 0xFFu;
                    const bool ok = sidx >= rsh && sidx < rsh + len && (by == 'A' || by == 'C' || by == 'G' || by == 'T');
                    vb |= (ok ? 1u : 0u) << (7 - e);
                }
                reinterpret_cast<uint8_t*>(sm.vd[half])[hl ^ 1u] = (uint8_t)vb;
                x.sync();
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    km[j] = p0 + j < nk ? window_kmer(sm.pk[half], sm.vd[half], rsh + p0 + j, k, nullptr, nullptr) : NAN64;
                    pend[j] = km[j] != NAN64;
                    npend += pend[j] ? 1u : 0u;
                    sm.res[half][p0 + j] = AUX_MISS;
                }
            }
            const uint32_t pex = x.wave_excl_scan(npend);
            const uint32_t ptot = x.bcast(pex + npend, 63);
            if (!again) {
                if (lane == 0 && r.rest) { x.lds_add(&smb.tally[0], 1u); if (ptot > LOC_BAIL) x.lds_add(&smb.tally[1], 1u); }
                if (ptot > LOC_BAIL && r.rest) {  // (uniform) not this locus' pair after all: the global-table kernel's
                    if (lane == 0) r.rest[x.atomic_add(r.nrest, 1u)] = i;
                    DBTK_STAMP(16);
                    continue;
                }
            }
            DBTK_STAMP(16);  // k-mers, image look-ups
            const uint32_t nh = x.half_sum(nres);
            // ---- the usual pair, resolved here (FUSE): both mates with cth k-mers and cth of them found, all of them this locus' alone
            // (AQ.cpp:190-228: kfilter keeps such a mate; 354-357, 439-451: countHit then needs no vote) — provided none of the queued
            // k-mers turns out to be in the index
            bool spec = false;
            uint32_t nshared = 0;  // (uniform) positions of the pair whose k-mer is shared between loci, when the pair is resolved here all the same
            if (FUSE) {
                const uint32_t nk0 = x.bcast(nk, 0), nk1 = x.bcast(nk, 32), nh0 = x.bcast(nh, 0), nh1 = x.bcast(nh, 32);
                spec = clean && !again && x.ballot(multi) == 0 && nk0 >= cth && nk1 >= cth;
                // A pair with k-mers SHARED between loci is still this locus' — whatever order fillstats' unstable sort leaves the k-mers
                // in — when (i) every found k-mer names the locus (the unique ones by their value, the shared ones by being in its image:
                // their vv lists hold it), (ii) at least one of them is unique to it and (iii) each mate has cth found positions, shared
                // ones included.  std::sort orders by the number of loci (AQ.cpp:320-327: only the order among EQUAL keys is
                // unspecified), so the k-mers unique to the locus come first: after the first of them the locus leads every other one
                // by at least that k-mer's count, and every later k-mer adds to the locus what it adds to any other — inside a shared
                // k-mer's list another locus may pass it for the moment (updatetop2, AQ.cpp:331-349), by the end of the list the locus
                // is `top` again, and find_matching_locus only looks at `top` between k-mers (AQ.cpp:383).  Wherever its first loop
                // breaks, the second (AQ.cpp:396-418) goes on adding the k-mers that hold top.idx — all of them — while a strand is
                // below cth and the remaining counts could still lift it there: with D_m >= cth found positions in mate m, "fc < cth
                // and cth - fc > remain" would mean D_1 - fc <= remain < cth - fc, i.e. D_1 < cth.  So it ends with fc >= cth and
                // rc >= cth, and countHit accepts the locus (AQ.cpp:439-451).  The partial sums nm1 / nm2 do depend on the order: trace
                // mode never comes here.  (Round 5 asked for cth UNIQUE positions per mate and no more shared than unique ones, at most
                // 96: that left 1.3 % of an all-hit batch — pairs reaching into a flank shared with a neighbour — to body_pair's introsort.)
                // What the shared k-mers still cost the reference is one vv word each in fillstats (their distinct number:
                // DBTK_C_ALGO_VV, below).
                const uint32_t sh = x.half_sum((uint32_t)__builtin_popcount(stbits >> 16));
                const uint32_t s0 = x.bcast(sh, 0), s1 = x.bcast(sh, 32);
                if (spec) {
                    if (s0 + s1 == 0) spec = nh0 >= cth && nh1 >= cth && nh0 && nh1;
                    else {
                        spec = nh0 + s0 >= cth && nh1 + s1 >= cth && nh0 + nh1 >= 1;
                        if (spec) nshared = s0 + s1;
                    }
                }
                if (lane == 0) { sm.spec[nrow >> 1].i = i; sm.spec[nrow >> 1].flags = spec ? LSP_SPEC : 0u; }  // (before anything of the pair is queued: the look-ups ask)
            }
            uint32_t pexq = pex, ptotq = ptot;  // places in the queue: without the shared k-mers when the pair is resolved here
            uint32_t nvv = 0;                   // (uniform) vv words fillstats reads for the pair: one per DISTINCT shared k-mer (AQ.cpp:311-316)
            if (FUSE && nshared) {
                // distinct shared k-mers of the pair (the mates overlap, a repeat repeats): every one is handed round the wave from its
                // lane's registers and compared with the ones after it — in the order (j, lane) — so that of equal ones all but the first
                // are marked; no LDS, no cap on their number
                uint32_t dupb = 0;
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    uint64_t mj = x.ballot(((stbits >> (16 + j)) & 1u) != 0);
                    while (mj) {
                        const int src = (int)__builtin_ctzll(mj);
                        mj &= mj - 1;
                        const uint64_t ke = ((uint64_t)x.bcast((uint32_t)(km[j] >> 32), src) << 32) | x.bcast((uint32_t)km[j], src);
#pragma unroll
                        for (int j2 = j; j2 < NPL; ++j2)
                            if (((stbits >> (16 + j2)) & 1u) && km[j2] == ke && (j2 > j || lane > src)) dupb |= 1u << j2;
                    }
                }
                nvv = nshared - x.wave_sum((uint32_t)__builtin_popcount(dupb));
            }
            if (FUSE && nshared) {
#pragma unroll
                for (int j = 0; j < NPL; ++j) if ((stbits >> (16 + j)) & 1u) pend[j] = false;
                uint32_t np2 = 0;
#pragma unroll
                for (int j = 0; j < NPL; ++j) np2 += pend[j] ? 1u : 0u;
                pexq = x.wave_excl_scan(np2);
                ptotq = x.bcast(pexq + np2, 63);
            }

#if !defined(__HIPCC__) && defined(DBTK_LOC_DEBUG)
            if (hl == 0) fprintf(stderr, "item %u wave %u lane %d pair %u nrow %u spec %d again %d clean %d nh %u nk %u\n", item, wave, lane, i, nrow, (int)spec, (int)again, (int)clean, nh, nk);
#endif
            if (FUSE && again) {
                // the counts this pair added when it was resolved ahead are subtracted (the same positions: the same image, the same read)
                const bool rm = ((rmbits >> (1 + half)) & 1u) != 0 || ((rmbits >> LSP_STAGE_SHIFT) & 7u) != LSP_COUNT;
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    const uint32_t ax = sm.res[half][p0 + j];
                    if (!rm && p0 + j < nk && ax != AUX_MISS && ax != CLS_FLANK) {
                        const uint32_t sl = ax - trb;
                        if (sl < LOC_FCNT) x.lds_add(&smb.fcnt[sl >> 1], 0u - (1u << (16 * (sl & 1))));
                        else x.atomic_add(&a.counts[ax], ~0ull);
                    }
                }
            }
            if (FUSE && !spec && (x.ballot((stbits >> 16) != 0))) {
                // the pair goes the general way: a shared k-mer's class is then the index's business (its row gets what the look-up says)
#pragma unroll
                for (int j = 0; j < NPL; ++j) if ((stbits >> (16 + j)) & 1u) sm.res[half][p0 + j] = AUX_MISS;
            }
            if (!spec) {   // the row as the image answers it: its statistics, its aux words (16-byte stores)
                if (hl == 0) {
                    sm.stat[row][0] = nh; sm.stat[row][1] = nh ? locus << 1 : 0u; sm.stat[row][2] = nh ? locus << 1 : 0xFFFFFFFFu; sm.stat[row][3] = hrow;
                    a.hitnk[hrow] = nk;
                    a.hitoff[hrow] = o0;
                }
                if (hl < (uint32_t)NPL) sm.patched[row][hl] = 0u;
                x.sync();
                uint4* outa = reinterpret_cast<uint4*>(a.hitaux + (size_t)hrow * a.nkp);
#pragma unroll
                for (int c = 0; c < (32 * NPL + 127) / 128; ++c) {
                    const uint32_t i4 = 32u * c + hl;
                    if (4 * i4 < nk && i4 < 8u * NPL) outa[i4] = reinterpret_cast<const uint4*>(sm.res[half])[i4];
                }
            } else x.sync();
            {   // what the image does not answer into the queue (a queue that cannot take them all is looked up first)
                uint32_t done = 0;  // entries of this pair already queued (uniform)
                while (done < ptotq) {
                    if (qn == (uint32_t)LOC_Q) flush();
                    const uint32_t take = ptotq - done < (uint32_t)LOC_Q - qn ? ptotq - done : (uint32_t)LOC_Q - qn;
                    uint32_t at = pexq;
#pragma unroll
                    for (int j = 0; j < NPL; ++j)
                        if (pend[j]) {
                            if (at >= done && at < done + take) {
                                sm.qkm[qn + at - done] = km[j];
                                sm.qcode[qn + at - done] = (row << 8) | (p0 + j);
                            }
                            ++at;
                        }
                    qn += take; done += take;
                }
            }
            DBTK_STAMP(18);  // row statistics + stores, queue
            if (FUSE && spec) {
                // states of the lane's positions at the locus: known (flank or TR), TR
                bool kn[NPL], tr[NPL];
#pragma unroll
                for (int j = 0; j < NPL; ++j) { kn[j] = (stbits >> j) & 1u; tr[j] = (stbits >> (8 + j)) & 1u; }
                uint32_t stage = LSP_COUNT;
                if (qc_off) stage = LSP_QC;
                else if (a.P.threading) stage = a.P.threading == DBTK_THREADING_V13 ? LSP_THREAD_V13 : LSP_THREAD_HEAD;
                else if (a.P.extract) stage = LSP_EXTRACT;
                bool rm = true;
                uint32_t span = 0;
                if (stage == LSP_COUNT) {
                    // a mate without a TR k-mer is removed (no state change, first state flank: AQ.cpp:1531-1534), one all of whose known
                    // k-mers are TR k-mers is the TR segment from end to end: only a mate with both needs the scan
                    const uint64_t anytr = x.ballot((stbits >> 8) != 0), anyfl = x.ballot((stbits & 0xFFu & ~(stbits >> 8)) != 0);
                    const bool t0 = (uint32_t)anytr != 0, t1 = (anytr >> 32) != 0, f0 = (uint32_t)anyfl != 0, f1 = (anyfl >> 32) != 0;
                    if ((t0 && f0) || (t1 && f1)) assign_halves<NPL>(x, kn, tr, p0, nk, a.P, rm, span);
                    else { const bool tm = half ? t1 : t0; rm = !tm; span = tm ? nk : 0u; }
                }
                const uint32_t rm0 = x.bcast(rm ? 1u : 0u, 0), rm1 = x.bcast(rm ? 1u : 0u, 32);
                uint32_t ninc = 0;
                if (!(rm0 && rm1)) {  // (a pair both of whose mates are removed adds nothing: AQ.cpp:2145)
#pragma unroll
                    for (int j = 0; j < NPL; ++j) {
                        const bool hit = tr[j] && !rm;
                        if (hit) {
                            const uint32_t axj = sm.res[half][p0 + j], sl = axj - trb;  // (the aux words of the look-up are still in LDS)
                            if (sl < LOC_FCNT) x.lds_add(&smb.fcnt[sl >> 1], 1u << (16 * (sl & 1)));
                            else x.atomic_add(&a.counts[axj], 1ull);
                        }
                        ninc += (uint32_t)__builtin_popcountll(x.ballot(hit));
                    }
                }
                const uint32_t span0 = x.bcast(span, 0), span1 = x.bcast(span, 32), nks = x.bcast(nk, 0) + x.bcast(nk, 32);
                if (lane == 0) {
                    LocSpecRow& sr = sm.spec[nrow >> 1];
                    x.lds_or(&sr.flags, (rm0 ? LSP_RM0 : 0u) | (rm1 ? LSP_RM1 : 0u) | (stage << LSP_STAGE_SHIFT));  // (or-ed: a look-up in the middle of the pair may have marked it already)
                    sr.kmc = span0 + span1; sr.nks = nks; sr.inc = ninc; sr.pad = nvv;
                }
            }
            nrow += 2;
            DBTK_STAMP(1);  // (FUSE) assignTRkmc + counts
        }
    }
    if (FUSE) {
        x.bsync();
        flush_counts();
        x.sync();
        if (lane == 0) {
            uint64_t* const ctr = counters_of(x, a);
            const uint32_t c0 = sm.ctr[0], c1 = sm.ctr[1], c2 = sm.ctr[2], c3 = sm.ctr[3], c4 = sm.ctr[4], c5 = sm.ctr[5], c6 = sm.ctr[6], c7 = sm.ctr[7];
            if (c0) x.atomic_add(&ctr[DBTK_C_QCFILTERED], (uint64_t)c0);
            if (c1) x.atomic_add(&ctr[DBTK_C_THREADING], (uint64_t)c1);
            if (c2) x.atomic_add(&ctr[DBTK_C_FEASIBLE], (uint64_t)c2);
            if (c3) x.atomic_add(&ctr[DBTK_C_ASGN], (uint64_t)c3);
            if (c4) x.atomic_add(&ctr[DBTK_C_ALGO_CLS], (uint64_t)c4);
            if (c5) x.atomic_add(&ctr[DBTK_C_ALGO_INC], (uint64_t)c5);
            if (c6) { x.atomic_add(&ctr[DBTK_C_NHASH1], (uint64_t)c6); x.atomic_add(&ctr[DBTK_C_ALGO_PROBES], (uint64_t)c6); }
            if (sm.ctr[9]) x.atomic_add(&ctr[DBTK_C_ALGO_VV], (uint64_t)sm.ctr[9]);
            if (a.pstats && sm.ctr[10]) x.atomic_add(&a.pstats[19], (uint64_t)sm.ctr[10]);
            if (a.pstats && c7) x.atomic_add(&a.pstats[14], (uint64_t)c7);
            if (a.pstats && sm.ctr[8]) x.atomic_add(&a.pstats[15], (uint64_t)sm.ctr[8]);
            if (a.pstats && c4) x.atomic_add(&a.pstats[16], (uint64_t)c4);
            if (a.pstats && c5) x.atomic_add(&a.pstats[17], (uint64_t)c5);
        }
    }
    DBTK_STAMP_FLUSH;
}

}  // namespace dbtk
#endif
