// dbtk_synth.cpp — seeded, multithreaded workload generator (host C++).
//
// The release RPGG of the reference is not available offline, so the
// benchmark configs of BASELINE.json are driven by a "release-scale" synthetic
// RPGG: 80 000 loci, 700 bp flanks, TR length log-uniform
// 50-5000 bp built from a random 8-60 bp motif with 30 % per-copy point
// variation, 2-4 haplotypes per locus differing by +-3 copies, 2 % of loci
// sharing 300 bp of flank with their neighbour  ->  ~1.3e8 index keys,
// ~1.5e7 TR k-mers.  Reads are 150 bp pairs tiled from the haplotypes with
// substitutions (0.1-0.5 % per pair) and rare indels, mixed with uniform
// random background pairs.  Everything is a pure function of (seed, index), so
// any thread count gives the same bytes.
//
// The output is the flat, file-equivalent RPGG of include/dbtk.h
// (dbtk_rpgg_arrays_t): what `fa2kmers` + `ktools serialize` would have put in
// PREF.tr.kmers / .kmers.dbi / .fl.kdb for these haplotypes.  Not part of the
// reference's interface; used by bench.py and the scale tests.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <stdio.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dbtk.h"

namespace {

struct Rng {  // splitmix64 stream
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};
inline uint64_t mix(uint64_t a, uint64_t b) {
    Rng r(a * 0xD6E8FEB86659FD93ull + b);
    r.next();
    return r.next();
}
const char ACGT[4] = {'A', 'C', 'G', 'T'};

struct Synth {
    uint32_t k = 21, flank = 700, nloci = 0;
    uint64_t seed = 0;
    // haplotype sequences
    std::vector<uint64_t> hap_beg;   // per (locus, hap) -> offset into seq; size = total haps + 1
    std::vector<uint32_t> locus_hap0;  // nloci+1: first hap index of each locus
    std::vector<uint8_t> seq;
    // flat RPGG
    std::vector<uint64_t> keys; std::vector<uint32_t> vals, vv;
    std::vector<uint64_t> fl_cnt, fl_ks, tr_cnt, tr_ks, tre_cnt;
    std::vector<uint64_t> gr_cnt, gr_ks;  // graphDB (dbtk_synth_graph): nodes of both strands, sorted within a locus
    std::vector<uint8_t> gr_ms;
};

template <class F>
void parallel_for(uint64_t n, unsigned nth, F f) {
    std::atomic<uint64_t> next(0);
    const uint64_t chunk = std::max<uint64_t>(1, n / (nth * 16ull));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nth; ++t)
        th.emplace_back([&, t]() {
            for (;;) {
                const uint64_t b = next.fetch_add(chunk);
                if (b >= n) break;
                f(b, std::min(n, b + chunk), t);
            }
        });
    for (auto& x : th) x.join();
}

void gen_flanks(const Synth& s, uint32_t l, std::vector<uint8_t>& lf, std::vector<uint8_t>& rf) {
    Rng r(mix(s.seed, 2ull * l));
    lf.resize(s.flank); rf.resize(s.flank);
    for (auto& c : lf) c = ACGT[r.below(4)];
    for (auto& c : rf) c = ACGT[r.below(4)];
}

// haplotypes of locus l (each: left flank + TR + right flank)
void gen_locus(const Synth& s, uint32_t l, std::vector<std::vector<uint8_t>>& haps) {
    std::vector<uint8_t> lf, rf, plf, prf;
    gen_flanks(s, l, lf, rf);
    Rng r(mix(s.seed, 2ull * l + 1));
    if (l > 0 && r.unit() < 0.02) {  // neighbour sharing 300 bp of flank -> odd val / vv
        gen_flanks(s, l - 1, plf, prf);
        const uint32_t n = std::min<uint32_t>(300, s.flank);
        memcpy(lf.data() + s.flank - n, plf.data() + s.flank - n, n);
    }
    const uint32_t mlen = 8 + r.below(53);
    std::vector<uint8_t> motif(mlen);
    for (auto& c : motif) c = ACGT[r.below(4)];
    const double trlen = exp(log(50.0) + r.unit() * (log(5000.0) - log(50.0)));
    const int ncopy = std::max(2, (int)(trlen / mlen));
    const uint32_t nhap = 2 + r.below(3);
    haps.assign(nhap, {});
    for (uint32_t h = 0; h < nhap; ++h) {
        const int nc = std::max(1, ncopy + (int)r.below(7) - 3);
        auto& v = haps[h];
        v.reserve(2 * s.flank + (size_t)nc * mlen);
        v.insert(v.end(), lf.begin(), lf.end());
        for (int c = 0; c < nc; ++c) {
            const size_t at = v.size();
            v.insert(v.end(), motif.begin(), motif.end());
            if (r.unit() < 0.30) v[at + r.below(mlen)] = ACGT[r.below(4)];
        }
        v.insert(v.end(), rf.begin(), rf.end());
    }
}

inline uint64_t revcomp2(uint64_t x, uint32_t k) {
    x = __builtin_bswap64(x);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    return (~x) >> (64 - 2 * k);
}
inline int code(uint8_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }

struct KP { uint64_t km; uint32_t pos; };

// unique k-mers of a class in first-occurrence order
void uniq_first_order(std::vector<KP>& v, std::vector<uint64_t>& out) {
    std::sort(v.begin(), v.end(), [](const KP& a, const KP& b) { return a.km != b.km ? a.km < b.km : a.pos < b.pos; });
    size_t n = 0;
    for (size_t i = 0; i < v.size(); ++i)
        if (i == 0 || v[i].km != v[i - 1].km) v[n++] = v[i];
    v.resize(n);
    std::sort(v.begin(), v.end(), [](const KP& a, const KP& b) { return a.pos < b.pos; });
    out.resize(n);
    for (size_t i = 0; i < n; ++i) out[i] = v[i].km;
}

struct KL { uint64_t km; uint32_t locus; uint32_t src; };  // src 0 = tr.kmers, 1 = fl.kmers (serialize reads tr first)

}  // namespace

extern "C" {

void* dbtk_synth_create(uint32_t nloci, uint32_t ksize, uint32_t flank, uint64_t seed, uint32_t nthreads) {
    Synth* s = new Synth;
    s->k = ksize; s->flank = flank; s->nloci = nloci; s->seed = seed;
    const unsigned nth = nthreads ? nthreads : std::max(1u, std::thread::hardware_concurrency());
    const uint32_t k = ksize;
    std::vector<std::vector<uint64_t>> trk(nloci), flk(nloci);
    std::vector<std::vector<std::vector<uint8_t>>> all(nloci);
    parallel_for(nloci, nth, [&](uint64_t b, uint64_t e, unsigned) {
        std::vector<KP> tr, fl;
        for (uint64_t l = b; l < e; ++l) {
            auto& haps = all[l];
            gen_locus(*s, (uint32_t)l, haps);
            tr.clear(); fl.clear();
            uint32_t base = 0;
            for (auto& h : haps) {
                const uint32_t n = (uint32_t)h.size();
                uint64_t fw = 0;
                uint32_t run = 0;
                const uint64_t mask = (k < 32) ? ((1ull << (2 * k)) - 1) : ~0ull;
                for (uint32_t i = 0; i < n; ++i) {
                    const int c = code(h[i]);
                    if (c < 0) { run = 0; continue; }
                    fw = ((fw << 2) | (uint64_t)c) & mask;
                    if (++run < k) continue;
                    const uint32_t p = i + 1 - k;  // window start
                    const uint64_t rc = revcomp2(fw, k);
                    const uint64_t ca = fw < rc ? fw : rc;
                    // fa2kmers -fsi F -fso F: TR k-mers start in [F, len-F-k], the rest are flank
                    const bool in_tr = p >= s->flank && p + k + s->flank <= n;
                    (in_tr ? tr : fl).push_back(KP{ca, base + p});
                }
                base += n;
            }
            uniq_first_order(tr, trk[l]);
            uniq_first_order(fl, flk[l]);
        }
    });
    // haplotype store
    s->locus_hap0.assign(nloci + 1, 0);
    for (uint32_t l = 0; l < nloci; ++l) s->locus_hap0[l + 1] = s->locus_hap0[l] + (uint32_t)all[l].size();
    s->hap_beg.assign(s->locus_hap0[nloci] + 1, 0);
    for (uint32_t l = 0; l < nloci; ++l)
        for (size_t h = 0; h < all[l].size(); ++h) s->hap_beg[s->locus_hap0[l] + h + 1] = all[l][h].size();
    for (size_t i = 1; i < s->hap_beg.size(); ++i) s->hap_beg[i] += s->hap_beg[i - 1];
    s->seq.resize(s->hap_beg.back());
    parallel_for(nloci, nth, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t l = b; l < e; ++l) {
            for (size_t h = 0; h < all[l].size(); ++h)
                memcpy(s->seq.data() + s->hap_beg[s->locus_hap0[l] + h], all[l][h].data(), all[l][h].size());
            std::vector<std::vector<uint8_t>>().swap(all[l]);
        }
    });
    // per-locus arrays
    s->tr_cnt.resize(nloci); s->fl_cnt.resize(nloci); s->tre_cnt.assign(nloci, 0);
    std::vector<uint64_t> trb(nloci + 1, 0), flb(nloci + 1, 0);
    for (uint32_t l = 0; l < nloci; ++l) {
        s->tr_cnt[l] = trk[l].size(); s->fl_cnt[l] = flk[l].size();
        trb[l + 1] = trb[l] + trk[l].size(); flb[l + 1] = flb[l] + flk[l].size();
    }
    s->tr_ks.resize(trb[nloci]); s->fl_ks.resize(flb[nloci]);
    // global index: partition (k-mer, locus, src) by hash, group per k-mer
    const uint32_t NP = 1024;
    std::vector<std::vector<KL>> part(NP);
    {
        std::vector<std::vector<std::vector<KL>>> local(nth, std::vector<std::vector<KL>>(NP));
        parallel_for(nloci, nth, [&](uint64_t b, uint64_t e, unsigned t) {
            for (uint64_t l = b; l < e; ++l) {
                memcpy(s->tr_ks.data() + trb[l], trk[l].data(), trk[l].size() * 8);
                memcpy(s->fl_ks.data() + flb[l], flk[l].data(), flk[l].size() * 8);
                for (uint64_t km : trk[l]) local[t][(km * 0x9E3779B97F4A7C15ull) >> 54].push_back(KL{km, (uint32_t)l, 0});
                for (uint64_t km : flk[l]) local[t][(km * 0x9E3779B97F4A7C15ull) >> 54].push_back(KL{km, (uint32_t)l, 1});
                std::vector<uint64_t>().swap(trk[l]);
                std::vector<uint64_t>().swap(flk[l]);
            }
        });
        parallel_for(NP, nth, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t p = b; p < e; ++p) {
                size_t n = 0;
                for (unsigned t = 0; t < nth; ++t) n += local[t][p].size();
                part[p].reserve(n);
                for (unsigned t = 0; t < nth; ++t) {
                    part[p].insert(part[p].end(), local[t][p].begin(), local[t][p].end());
                    std::vector<KL>().swap(local[t][p]);
                }
            }
        });
    }
    std::vector<std::vector<uint64_t>> pkeys(NP);
    std::vector<std::vector<uint32_t>> pvals(NP), pvv(NP);
    parallel_for(NP, nth, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t p = b; p < e; ++p) {
            auto& v = part[p];
            // readKmerIndex order: all of tr.kmers (locus ascending), then fl.kmers (src/kmerIO.hpp:47-78)
            std::sort(v.begin(), v.end(), [](const KL& a, const KL& c) {
                if (a.km != c.km) return a.km < c.km;
                if (a.src != c.src) return a.src < c.src;
                return a.locus < c.locus;
            });
            std::vector<uint32_t> loci;
            for (size_t i = 0; i < v.size();) {
                size_t j = i;
                loci.clear();
                for (; j < v.size() && v[j].km == v[i].km; ++j)
                    if (std::find(loci.begin(), loci.end(), v[j].locus) == loci.end()) loci.push_back(v[j].locus);
                pkeys[p].push_back(v[i].km);
                if (loci.size() == 1) pvals[p].push_back(loci[0] << 1);
                else {
                    pvals[p].push_back((uint32_t)((pvv[p].size() << 1) | 1));  // local offset, fixed below
                    pvv[p].push_back((uint32_t)loci.size());
                    pvv[p].insert(pvv[p].end(), loci.begin(), loci.end());
                }
                i = j;
            }
            std::vector<KL>().swap(v);
        }
    });
    std::vector<uint64_t> kb(NP + 1, 0), vb(NP + 1, 0);
    for (uint32_t p = 0; p < NP; ++p) { kb[p + 1] = kb[p] + pkeys[p].size(); vb[p + 1] = vb[p] + pvv[p].size(); }
    s->keys.resize(kb[NP]); s->vals.resize(kb[NP]); s->vv.resize(vb[NP]);
    parallel_for(NP, nth, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t p = b; p < e; ++p) {
            memcpy(s->keys.data() + kb[p], pkeys[p].data(), pkeys[p].size() * 8);
            for (size_t i = 0; i < pvals[p].size(); ++i) {
                uint32_t v = pvals[p][i];
                if (v & 1) v = (uint32_t)((((uint64_t)(v >> 1) + vb[p]) << 1) | 1);
                s->vals[kb[p] + i] = v;
            }
            if (!pvv[p].empty()) memcpy(s->vv.data() + vb[p], pvv[p].data(), pvv[p].size() * 4);
        }
    });
    return s;
}

void dbtk_synth_free(void* h) { delete (Synth*)h; }

void dbtk_synth_arrays(void* h, dbtk_rpgg_arrays_t* a) {
    Synth* s = (Synth*)h;
    memset(a, 0, sizeof(*a));
    a->ksize = s->k; a->nloci = s->nloci;
    a->nkeys = s->keys.size(); a->keys = s->keys.data(); a->vals = s->vals.data();
    a->nvv = s->vv.size(); a->vv = s->vv.data();
    a->fl_cnt = s->fl_cnt.data(); a->fl_ks = s->fl_ks.data();
    a->tr_cnt = s->tr_cnt.data(); a->tr_ks = s->tr_ks.data();
    a->tre_cnt = nullptr; a->tre_ks = nullptr;  // TR edges only matter to -bu
    if (!s->gr_cnt.empty()) { a->gr_cnt = s->gr_cnt.data(); a->gr_ks = s->gr_ks.data(); a->gr_ms = s->gr_ms.data(); }
}

// graphDB of the haplotypes, as `fa2kmers -g` builds it (buildKmerGraph, src/aQueryFasta_thread.h:215-243): every k-mer of
// both strands of every haplotype is a node; bit b of its mask: the successor ((node & rmask) << 2) | b follows it
// somewhere; no self loops.  After this call dbtk_synth_arrays also hands out gr_cnt / gr_ks / gr_ms.
void dbtk_synth_graph(void* h, uint32_t nthreads) {
    Synth* s = (Synth*)h;
    if (!s->gr_cnt.empty()) return;
    const unsigned nth = nthreads ? nthreads : std::max(1u, std::thread::hardware_concurrency());
    const uint32_t k = s->k, nloci = s->nloci;
    const uint64_t mask = (k < 32) ? ((1ull << (2 * k)) - 1) : ~0ull;
    std::vector<std::vector<uint64_t>> nk(nloci);
    std::vector<std::vector<uint8_t>> nm(nloci);
    parallel_for(nloci, nth, [&](uint64_t b, uint64_t e, unsigned) {
        std::vector<std::pair<uint64_t, uint8_t>> v;
        std::vector<uint8_t> rcs;
        for (uint64_t l = b; l < e; ++l) {
            v.clear();
            for (uint32_t hi = s->locus_hap0[l]; hi < s->locus_hap0[l + 1]; ++hi) {
                const uint8_t* hs = s->seq.data() + s->hap_beg[hi];
                const uint32_t hl = (uint32_t)(s->hap_beg[hi + 1] - s->hap_beg[hi]);
                rcs.resize(hl);
                for (uint32_t i = 0; i < hl; ++i) { const uint8_t c = hs[hl - 1 - i]; rcs[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; }
                for (int strand = 0; strand < 2; ++strand) {
                    const uint8_t* q = strand ? rcs.data() : hs;
                    uint64_t fw = 0;
                    for (uint32_t i = 0; i < hl; ++i) {
                        fw = ((fw << 2) | (uint64_t)code(q[i])) & mask;
                        if (i + 1 < k) continue;
                        uint8_t m = 0;
                        if (i + 1 < hl) {
                            const uint64_t nx = ((fw << 2) | (uint64_t)code(q[i + 1])) & mask;
                            if (nx != fw) m = (uint8_t)(1u << code(q[i + 1]));
                        }
                        v.emplace_back(fw, m);
                    }
                }
            }
            std::sort(v.begin(), v.end());
            for (size_t i = 0; i < v.size(); ++i) {
                if (!nk[l].empty() && nk[l].back() == v[i].first) nm[l].back() |= v[i].second;
                else { nk[l].push_back(v[i].first); nm[l].push_back(v[i].second); }
            }
        }
    });
    s->gr_cnt.resize(nloci);
    std::vector<uint64_t> gb(nloci + 1, 0);
    for (uint32_t l = 0; l < nloci; ++l) { s->gr_cnt[l] = nk[l].size(); gb[l + 1] = gb[l] + nk[l].size(); }
    s->gr_ks.resize(gb[nloci]); s->gr_ms.resize(gb[nloci]);
    parallel_for(nloci, nth, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t l = b; l < e; ++l) {
            if (!nk[l].empty()) { memcpy(s->gr_ks.data() + gb[l], nk[l].data(), nk[l].size() * 8); memcpy(s->gr_ms.data() + gb[l], nm[l].data(), nm[l].size()); }
            std::vector<uint64_t>().swap(nk[l]);
            std::vector<uint8_t>().swap(nm[l]);
        }
    });
}

// The RPGG as the files the reference's `danbing-tk` loads (HEAD formats): PREF.tr.kmers (text),
// PREF.kmers.dbi, PREF.fl.kdb, PREF.tre.kdb (no edges: only -bu reads them), and PREF.graph.umap once the graph exists — for bench.py's reference-binary
// baseline, which must see the same RPGG as the GPU.  Returns 0 on success.
int dbtk_synth_write_files(void* h, const char* prefix) {
    Synth* s = (Synth*)h;
    const std::string pref(prefix);
    auto put = [](FILE* f, const void* p, size_t n) { return n == 0 || fwrite(p, 1, n, f) == n; };
    {
        FILE* f = fopen((pref + ".tr.kmers").c_str(), "wb");
        if (!f) return -1;
        std::vector<char> buf;
        buf.reserve(1 << 24);
        uint64_t i = 0;
        char line[64];
        for (uint32_t l = 0; l < s->nloci; ++l) {
            int n = snprintf(line, sizeof line, ">%u\n", l);
            buf.insert(buf.end(), line, line + n);
            for (uint64_t j = 0; j < s->tr_cnt[l]; ++j, ++i) {
                n = snprintf(line, sizeof line, "%llu\t0\n", (unsigned long long)s->tr_ks[i]);
                buf.insert(buf.end(), line, line + n);
            }
            if (buf.size() > (1u << 24) - 4096) { if (!put(f, buf.data(), buf.size())) { fclose(f); return -1; } buf.clear(); }
        }
        const bool ok = put(f, buf.data(), buf.size());
        if (fclose(f) || !ok) return -1;
    }
    {
        FILE* f = fopen((pref + ".kmers.dbi").c_str(), "wb");
        if (!f) return -1;
        const uint64_t nk = s->keys.size(), nvv = s->vv.size();
        const bool ok = put(f, &nk, 8) && put(f, s->keys.data(), nk * 8) && put(f, s->vals.data(), nk * 4) && put(f, &nvv, 8) && put(f, s->vv.data(), nvv * 4);
        if (fclose(f) || !ok) return -1;
    }
    for (int which = 0; which < 2; ++which) {
        FILE* f = fopen((pref + (which ? ".tre.kdb" : ".fl.kdb")).c_str(), "wb");
        if (!f) return -1;
        const uint64_t nl = s->nloci, nk = which ? 0 : s->fl_ks.size();
        const std::vector<uint64_t>& cnt = which ? s->tre_cnt : s->fl_cnt;
        const bool ok = put(f, &nl, 8) && put(f, cnt.data(), nl * 8) && put(f, &nk, 8) && (which || put(f, s->fl_ks.data(), nk * 8));
        if (fclose(f) || !ok) return -1;
    }
    if (!s->gr_cnt.empty()) {  // the graph (after dbtk_synth_graph), in the v1.3 PREF.graph.umap layout: nloci, then per locus n and n x (u64 node, u8 out-edge mask)
        FILE* f = fopen((pref + ".graph.umap").c_str(), "wb");
        if (!f) return -1;
        const uint64_t nl = s->nloci;
        bool ok = put(f, &nl, 8);
        std::vector<uint8_t> rec;
        uint64_t at = 0;
        for (uint32_t l = 0; l < s->nloci && ok; ++l) {
            const uint64_t n = s->gr_cnt[l];
            rec.resize(n * 9);
            for (uint64_t i = 0; i < n; ++i) { memcpy(&rec[9 * i], &s->gr_ks[at + i], 8); rec[9 * i + 8] = s->gr_ms[at + i]; }
            at += n;
            ok = put(f, &n, 8) && put(f, rec.data(), rec.size());
        }
        if (fclose(f) || !ok) return -1;
    }
    return 0;
}

// Interleaved 2-line FASTA (">p<index>/1", "/2") of npairs pairs laid out as dbtk_synth_reads makes them: the input file of
// the reference binary for bench.py's CPU baseline.  Returns 0 on success.
int dbtk_synth_write_fasta(const uint8_t* reads, uint64_t npairs, uint32_t rlen, uint64_t first_pair, const char* fn) {
    FILE* f = fopen(fn, "wb");
    if (!f) return -1;
    std::vector<char> buf;
    buf.reserve(1 << 24);
    char t[48];
    bool ok = true;
    for (uint64_t p = 0; p < npairs && ok; ++p) {
        for (int m = 0; m < 2; ++m) {
            const int n = snprintf(t, sizeof t, ">p%llu/%d\n", (unsigned long long)(first_pair + p), m + 1);
            buf.insert(buf.end(), t, t + n);
            const uint8_t* r = reads + (2 * p + m) * (uint64_t)rlen;
            buf.insert(buf.end(), r, r + rlen);
            buf.push_back('\n');
        }
        if (buf.size() > (1u << 24) - 1024) { ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size(); buf.clear(); }
    }
    if (ok && !buf.empty()) ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    if (fclose(f)) ok = false;
    return ok ? 0 : -1;
}

uint64_t dbtk_synth_nbases(void* h) { return ((Synth*)h)->seq.size(); }

// npairs pairs of rlen-bp reads into out[npairs * 2 * rlen]; read r at r*rlen.
// hit_frac of the pairs come from the loci, the rest are uniform random.
void dbtk_synth_reads(void* h, uint64_t npairs, uint64_t first_pair, uint32_t rlen, double hit_frac, uint64_t seed,
                      uint8_t* out, uint32_t nthreads) {
    Synth* s = (Synth*)h;
    const unsigned nth = nthreads ? nthreads : std::max(1u, std::thread::hardware_concurrency());
    const uint32_t nhaps = s->locus_hap0[s->nloci];
    parallel_for(npairs, nth, [&](uint64_t b, uint64_t e, unsigned) {
        std::vector<uint8_t> frag;
        for (uint64_t p = b; p < e; ++p) {
            Rng r(mix(seed ^ 0xA5A5A5A5ull, first_pair + p));
            uint8_t* m1 = out + (2 * p) * (uint64_t)rlen;
            uint8_t* m2 = m1 + rlen;
            if (r.unit() >= hit_frac) {
                for (uint32_t i = 0; i < rlen; ++i) m1[i] = ACGT[r.below(4)];
                for (uint32_t i = 0; i < rlen; ++i) m2[i] = ACGT[r.below(4)];
                continue;
            }
            const uint32_t hi = r.below(nhaps);
            const uint8_t* hs = s->seq.data() + s->hap_beg[hi];
            const uint32_t hl = (uint32_t)(s->hap_beg[hi + 1] - s->hap_beg[hi]);
            uint32_t fl = 300 + r.below(201);
            if (fl > hl) fl = hl;
            if (fl < rlen) fl = rlen;
            const uint32_t beg = r.below(hl - fl + 1);
            memcpy(m1, hs + beg, rlen);
            for (uint32_t i = 0; i < rlen; ++i) {  // reverse complement of the fragment's end
                const uint8_t c = hs[beg + fl - 1 - i];
                m2[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
            }
            const double sub = 0.001 + 0.004 * r.unit();
            for (int m = 0; m < 2; ++m) {
                uint8_t* x = m ? m2 : m1;
                for (uint32_t i = 0; i < rlen; ++i)
                    if (r.unit() < sub) x[i] = ACGT[r.below(4)];
                if (r.unit() < 0.0001 * rlen) {  // one indel, length kept
                    const uint32_t at = 1 + r.below(rlen - 2);
                    if (r.below(2)) { memmove(x + at + 1, x + at, rlen - at - 1); x[at] = ACGT[r.below(4)]; }
                    else { memmove(x + at, x + at + 1, rlen - at - 1); x[rlen - 1] = ACGT[r.below(4)]; }
                }
            }
            if (r.below(2)) for (uint32_t i = 0; i < rlen; ++i) std::swap(m1[i], m2[i]);
        }
    });
}

// The same read model, but every pair is drawn from a locus of a given LIST (pair p from loci[(first_pair + p) % nsel], any of its
// haplotypes): a dense slice — many pairs per locus, the regime of the locus-resident kernels (dbtk_locus.h) — on an RPGG of any size.
// odd_frac of the pairs are not what the list says: half of them CHIMERIC (mate 1 from the listed locus, mate 2 from any locus of the
// RPGG), half FOREIGN (both mates from any locus).
void dbtk_synth_reads_loci(void* h, uint64_t npairs, uint64_t first_pair, uint32_t rlen, const uint32_t* loci, uint32_t nsel, double odd_frac,
                           uint64_t seed, uint8_t* out, uint32_t nthreads) {
    Synth* s = (Synth*)h;
    const unsigned nth = nthreads ? nthreads : std::max(1u, std::thread::hardware_concurrency());
    if (!nsel) return;
    parallel_for(npairs, nth, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t p = b; p < e; ++p) {
            Rng r(mix(seed ^ 0x5A5A5A5A5Aull, first_pair + p));
            uint8_t* m1 = out + (2 * p) * (uint64_t)rlen;
            uint8_t* m2 = m1 + rlen;
            uint32_t l1 = loci[(first_pair + p) % nsel] % s->nloci, l2 = l1;
            if (r.unit() < odd_frac) {
                l2 = r.below(s->nloci);
                if (r.below(2)) l1 = l2;  // foreign: both mates from the other locus; else chimeric
            }
            auto frag = [&](uint32_t l, const uint8_t*& hs, uint32_t& hl, uint32_t& fl, uint32_t& beg) {
                const uint32_t h0 = s->locus_hap0[l], nh = s->locus_hap0[l + 1] - h0;
                const uint32_t hi = h0 + r.below(nh ? nh : 1);
                hs = s->seq.data() + s->hap_beg[hi];
                hl = (uint32_t)(s->hap_beg[hi + 1] - s->hap_beg[hi]);
                fl = 300 + r.below(201);
                if (fl > hl) fl = hl;
                if (fl < rlen) fl = rlen;
                beg = r.below(hl - fl + 1);
            };
            const uint8_t *hs1, *hs2;
            uint32_t hl1, fl1, beg1, hl2, fl2, beg2;
            frag(l1, hs1, hl1, fl1, beg1);
            if (l2 != l1) frag(l2, hs2, hl2, fl2, beg2); else { hs2 = hs1; hl2 = hl1; fl2 = fl1; beg2 = beg1; }
            memcpy(m1, hs1 + beg1, rlen);
            for (uint32_t i = 0; i < rlen; ++i) {
                const uint8_t c = hs2[beg2 + fl2 - 1 - i];
                m2[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
            }
            const double sub = 0.001 + 0.004 * r.unit();
            for (int m = 0; m < 2; ++m) {
                uint8_t* x = m ? m2 : m1;
                for (uint32_t i = 0; i < rlen; ++i)
                    if (r.unit() < sub) x[i] = ACGT[r.below(4)];
                if (r.unit() < 0.0001 * rlen) {
                    const uint32_t at = 1 + r.below(rlen - 2);
                    if (r.below(2)) { memmove(x + at + 1, x + at, rlen - at - 1); x[at] = ACGT[r.below(4)]; }
                    else { memmove(x + at, x + at + 1, rlen - at - 1); x[rlen - 1] = ACGT[r.below(4)]; }
                }
            }
            if (r.below(2)) for (uint32_t i = 0; i < rlen; ++i) std::swap(m1[i], m2[i]);
        }
    });
}

}  // extern "C"
