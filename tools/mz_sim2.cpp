// Diagnostic (host only): requests (distinct 128-byte lines per load instruction) per read of the probe kernel when the
// survivor list is in locus order and a wave works through a contiguous range of it with a small software cache of
// look-up results.   g++ -O3 -std=c++17 -I. -o tools/mz_sim2 tools/mz_sim2.cpp -ldl -lpthread ; tools/mz_sim2 [nloci=800] [npairs=20000]
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <unordered_map>
#include <vector>
#include "danbing-tk_amd/csrc/dbtk_tables.h"
using namespace dbtk;
// (the table header no longer returns the minimizer's offset — the final layout does not use it; the designs simulated here do)
static inline void mz_of_kmer_off(uint64_t kmer, uint32_t k, uint32_t m, uint32_t* h28, uint32_t* off) {
    const uint64_t mm = (1ull << 2 * m) - 1;
    uint32_t best = 0xFFFFFFFFu, bo = 0;
    for (uint32_t i = 0; i + m <= k; ++i) {
        const uint32_t h = dbtk::mmer_hash((kmer >> (2 * (k - m - i))) & mm, m);
        if (h < best) { best = h; bo = i; }
    }
    *h28 = best >> 4; *off = bo;
}
int main(int argc, char** argv) {
    const uint32_t nloci = argc > 1 ? atoi(argv[1]) : 800, K = 21, M = 15;
    const uint64_t npairs = argc > 2 ? atoll(argv[2]) : 20000;
    void* so = dlopen("danbing-tk_amd/libdbtk_synth.so", RTLD_NOW);
    auto create = (void* (*)(uint32_t, uint32_t, uint32_t, uint64_t, uint32_t))dlsym(so, "dbtk_synth_create");
    auto arrays = (void (*)(void*, dbtk_rpgg_arrays_t*))dlsym(so, "dbtk_synth_arrays");
    auto reads = (void (*)(void*, uint64_t, uint64_t, uint32_t, double, uint64_t, uint8_t*, uint32_t))dlsym(so, "dbtk_synth_reads");
    void* s = create(nloci, K, 700, 20250808, 0);
    dbtk_rpgg_arrays_t a; arrays(s, &a);
    uint64_t nb = 1; while (nb < a.nkeys * 6 / 8 + 8) nb <<= 1;
    struct Bk { uint64_t key[8]; uint8_t n, turned, tslot; };
    std::vector<Bk> TA(nb), TB(nb);
    for (auto* T : {&TA, &TB}) for (auto& b : *T) { for (int i = 0; i < 8; ++i) b.key[i] = NAN64; b.n = b.turned = b.tslot = 0; }
    std::unordered_map<uint64_t, uint32_t> val;
    for (uint64_t i = 0; i < a.nkeys; ++i) {
        const uint64_t km = a.keys[i]; val[km] = a.vals[i];
        uint32_t h, o; mz_of_kmer_off(km, K, M, &h, &o);
        const uint64_t b = mz_bucket(h, (uint32_t)(nb - 1)); const uint32_t sl = o & 7;
        { Bk& B = TA[b]; if (B.n < 8) B.key[B.n++] = km; else B.turned = 1; }
        { Bk& B = TB[b]; if (B.key[sl] == NAN64) B.key[sl] = km; else B.tslot |= 1 << sl; }
    }
    std::vector<uint8_t> seq(npairs * 300);
    reads(s, npairs, 0, 150, 1.0, 1, seq.data(), 0);
    auto kmers_of = [&](const uint8_t* rd, std::vector<uint64_t>& out) {
        out.assign(130, NAN64); uint64_t fw = 0; int valid = 0;
        for (int i = 0; i < 150; ++i) {
            int c = rd[i] == 'A' ? 0 : rd[i] == 'C' ? 1 : rd[i] == 'G' ? 2 : rd[i] == 'T' ? 3 : -1;
            if (c < 0) { valid = 0; continue; }
            fw = ((fw << 2) | c) & ((1ull << 2 * K) - 1); ++valid;
            if (valid >= (int)K) { const uint64_t rc = revcomp2(fw, K); out[i - K + 1] = fw < rc ? fw : rc; }
        }
    };
    // locus key of every pair (first sampled k-mer of mate 1 found in the index), then locus order
    std::vector<std::pair<uint32_t, uint32_t>> order;
    std::vector<uint64_t> km;
    for (uint64_t p = 0; p < npairs; ++p) {
        kmers_of(&seq[p * 300], km);
        uint32_t key = nloci;
        for (int pos : {0, 43, 86, 129}) { auto it = val.find(km[pos]); if (it != val.end()) { key = it->second & 1 ? nloci + 1 : it->second >> 1; break; } }
        order.emplace_back(key, (uint32_t)p);
    }
    for (int sorted = 0; sorted < 2; ++sorted) {
        if (sorted) std::sort(order.begin(), order.end());
        for (int cache_entries : {0, 256, 512, 1024}) for (int policy = 1; policy < 3; ++policy) {  // 1: any-slot with a cache of level-2 results, 2: any-slot with a cache of ALL results  // policy 0: offset-slotted (B), 1: any-slot (A)
            const uint64_t per_wave = 244;  // pairs of a wave's contiguous range
            uint64_t r1 = 0, r2 = 0, nreads = 0, l2look = 0, l2hit = 0;
            std::vector<uint64_t> cache(std::max(cache_entries, 1), NAN64);
            for (uint64_t q = 0; q < npairs; ++q) {
                if (q % per_wave == 0) std::fill(cache.begin(), cache.end(), NAN64);
                for (int mate = 0; mate < 2; ++mate) {
                    kmers_of(&seq[(uint64_t)order[q].second * 300 + 150 * mate], km);
                    ++nreads;
                    for (int ins = 0; ins < 3; ++ins) {  // 64 positions per load instruction
                        std::vector<uint64_t> l1, l2;
                        for (int pos = 64 * ins; pos < std::min(130, 64 * ins + 64); ++pos) {
                            if (km[pos] == NAN64) continue;
                            if (policy == 2 && cache_entries) {
                                uint64_t& e = cache[ovf_hash(km[pos] * 0x9E3779B97F4A7C15ull >> 7) % cache_entries];
                                ++l2look;
                                if (e == km[pos]) { ++l2hit; continue; }
                                e = km[pos];
                            }
                            uint32_t h, o; mz_of_kmer_off(km[pos], K, M, &h, &o);
                            const uint64_t b = mz_bucket(h, (uint32_t)(nb - 1)); const uint32_t sl = o & 7;
                            l1.push_back(b);
                            bool pend;
                            if (policy == 0) { const Bk& B = TB[b]; pend = B.key[sl] != km[pos] && (B.tslot >> sl & 1); }
                            else { const Bk& B = TA[b]; bool f = false; for (int e = 0; e < 8; ++e) f |= B.key[e] == km[pos]; pend = !f && B.turned; }
                            if (!pend) continue;
                            if (policy != 2) ++l2look;
                            if (cache_entries && policy != 2) {
                                uint64_t& e = cache[ovf_hash(km[pos] * 0x9E3779B97F4A7C15ull >> 7) % cache_entries];
                                if (e == km[pos]) { ++l2hit; continue; }
                                e = km[pos];
                            }
                            l2.push_back(ovf_hash(km[pos]) >> 3);
                        }
                        for (auto* v : {&l1, &l2}) { std::sort(v->begin(), v->end()); v->erase(std::unique(v->begin(), v->end()), v->end()); }
                        r1 += l1.size(); r2 += l2.size();
                    }
                }
            }
            printf("%s  %s  cache %4d entries: requests per read: level 1 %.1f  level 2 %.1f  (level-2 look-ups %.1f per read, cache hits %.0f %%)\n", sorted ? "locus order " : "random order",
                   policy == 2 ? "any-slot, unified cache" : policy ? "any-slot, level-2 cache" : "offset-slotted", cache_entries, (double)r1 / nreads, (double)r2 / nreads, (double)l2look / nreads, 100.0 * l2hit / std::max<uint64_t>(l2look, 1));
        }
    }
    return 0;
}
