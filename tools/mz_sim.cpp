// Diagnostic (host only): how a minimizer-grouped copy of the index answers the look-ups of all-hit reads under different
// in-bucket placement policies.  g++ -O3 -std=c++17 -I. -o tools/mz_sim tools/mz_sim.cpp -ldl -lpthread
//   usage: tools/mz_sim [nloci=8000] [npairs=100000] [k=21] [m=15] [per=6]
// Policies:  A  any free slot of 8 (round 2's MzBucket)
//            B  slot = offset of the minimizer inside the canonical k-mer (one key per (minimizer, offset))
//            C  B + slot 7 (k = 21 leaves it unused) as a shared spare
//            D  B, else any free slot (bucket flag "has displaced keys": a look-up that misses its slot scans the bucket)
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

static uint32_t K, M, WN;
static void mz_of(uint64_t km, uint32_t* h32, uint32_t* off) { mz_of_kmer_off(km, K, M, h32, off); }
static uint64_t bkt(uint32_t mz28, uint32_t shift) { return dbtk::mz_bucket(mz28, (uint32_t)((1ull << (64 - shift)) - 1)); }

int main(int argc, char** argv) {
    const uint32_t nloci = argc > 1 ? atoi(argv[1]) : 8000;
    const uint64_t npairs = argc > 2 ? atoll(argv[2]) : 100000;
    K = argc > 3 ? atoi(argv[3]) : 21; M = argc > 4 ? atoi(argv[4]) : 15; WN = K - M + 1;
    const uint64_t per = argc > 5 ? atoll(argv[5]) : 6;
    void* so = dlopen("danbing-tk_amd/libdbtk_synth.so", RTLD_NOW);
    if (!so) { fprintf(stderr, "%s\n", dlerror()); return 1; }
    auto create = (void* (*)(uint32_t, uint32_t, uint32_t, uint64_t, uint32_t))dlsym(so, "dbtk_synth_create");
    auto arrays = (void (*)(void*, dbtk_rpgg_arrays_t*))dlsym(so, "dbtk_synth_arrays");
    auto reads = (void (*)(void*, uint64_t, uint64_t, uint32_t, double, uint64_t, uint8_t*, uint32_t))dlsym(so, "dbtk_synth_reads");
    void* s = create(nloci, K, 700, 20250808, 0);
    dbtk_rpgg_arrays_t a; arrays(s, &a);
    fprintf(stderr, "nkeys %lu\n", (unsigned long)a.nkeys);
    uint64_t nb = 1; while (nb < a.nkeys * per / 8 + 8) nb <<= 1;
    const uint32_t shift = 64 - __builtin_ctzll(nb);
    struct Bk { uint64_t key[8]; uint8_t n; uint8_t turned; uint8_t displaced; uint8_t tslot; };
    std::vector<Bk> TA(nb), TB(nb), TC(nb), TD(nb);
    // W ways per offset slot, bucket count scaled so that the memory stays that of nb 8-slot buckets
    struct BW { uint64_t key[8][4]; uint8_t n[8]; uint8_t turned; };
    std::vector<BW> T2(nb / 2), T4(nb / 4);
    for (auto* T : {&T2, &T4}) for (auto& b : *T) { memset(&b, 0, sizeof(b)); }
    for (auto* T : {&TA, &TB, &TC, &TD}) for (auto& b : *T) { for (int i = 0; i < 8; ++i) b.key[i] = NAN64; b.n = b.turned = b.displaced = b.tslot = 0; }
    uint64_t offhist[32] = {0};
    for (uint64_t i = 0; i < a.nkeys; ++i) {
        const uint64_t km = a.keys[i];
        uint32_t h, o; mz_of(km, &h, &o);
        const uint64_t b = bkt(h, shift);
        const uint32_t sl = o & 7;
        ++offhist[o];
        { BW& B = T2[b & (nb / 2 - 1)]; if (B.n[sl] < 2) B.key[sl][B.n[sl]++] = km; else B.turned |= 1 << sl; }
        { BW& B = T4[b & (nb / 4 - 1)]; if (B.n[sl] < 4) B.key[sl][B.n[sl]++] = km; else B.turned |= 1 << sl; }
        { Bk& B = TA[b]; if (B.n < 8) B.key[B.n++] = km; else B.turned = 1; }
        { Bk& B = TB[b]; if (B.key[sl] == NAN64) B.key[sl] = km; else B.tslot |= 1 << sl; }
        { Bk& B = TC[b]; if (B.key[sl] == NAN64) B.key[sl] = km; else if (B.key[7] == NAN64) { B.key[7] = km; B.displaced |= 1 << sl; } else B.tslot |= 1 << sl; }
        { Bk& B = TD[b]; if (B.key[sl] == NAN64) B.key[sl] = km; else { int f = -1; for (int q = 7; q >= 0; --q) if (B.key[q] == NAN64) { f = q; break; } if (f >= 0) { B.key[f] = km; B.displaced = 1; } else B.turned = 1; } }
    }
    printf("offset histogram of keys:"); for (uint32_t i = 0; i < WN; ++i) printf(" %.3f", (double)offhist[i] / a.nkeys); printf("\n");
    // reads
    std::vector<uint8_t> seq(npairs * 300);
    reads(s, npairs, 0, 150, 1.0, 1, seq.data(), 0);
    // request model: one request per distinct (load instruction, 128-byte line)
    uint64_t req1[3] = {0, 0, 0}, req2[3] = {0, 0, 0}, req2A[3] = {0, 0, 0};
    uint64_t omask = 1; { uint64_t nt = 0; for (auto& b : TB) nt += __builtin_popcount(b.tslot); while (omask < 4 * nt + 8) omask <<= 1; omask -= 1; }
    uint64_t nlook = 0, runs = 0;
    uint64_t A_home = 0, A_ovf = 0, A_miss = 0;
    uint64_t B_home = 0, B_ovf = 0, B_miss = 0;
    uint64_t C_home = 0, C_spare = 0, C_ovf = 0, C_miss = 0;
    uint64_t W2_home = 0, W2_ovf = 0, W4_home = 0, W4_ovf = 0, runs2 = 0, runs4 = 0; uint64_t pb2 = ~0ull, pb4 = ~0ull;
    uint64_t D_home = 0, D_scan = 0, D_ovf = 0, D_miss = 0, D_scanreads = 0, B_ovfreads = 0, A_ovfreads = 0, nreads = 0;
    for (uint64_t r = 0; r < 2 * npairs; ++r) {
        const uint8_t* rd = &seq[r * 150];
        uint64_t fw = 0; int valid = 0; uint64_t prevb = ~0ull;
        std::vector<uint64_t> L1(150, ~0ull), L2(150, ~0ull), L2A(150, ~0ull);  // per position: level-1 line, level-2 line (policy B / A), ~0: none
        bool dscan = false, bovf = false, aovf = false;
        for (int i = 0; i < 150; ++i) {
            int c = rd[i] == 'A' ? 0 : rd[i] == 'C' ? 1 : rd[i] == 'G' ? 2 : rd[i] == 'T' ? 3 : -1;
            if (c < 0) { valid = 0; continue; }
            fw = ((fw << 2) | c) & ((1ull << 2 * K) - 1); ++valid;
            if (valid < (int)K) continue;
            const uint64_t rc = revcomp2(fw, K), km = fw < rc ? fw : rc;
            uint32_t h, o; mz_of(km, &h, &o);
            const uint64_t b = bkt(h, shift);
            const uint32_t sl = o & 7;
            ++nlook; if (b != prevb) ++runs; prevb = b;
            { const int pos = i - (int)K + 1; L1[pos] = b;
              const Bk& B = TB[b]; if (B.key[sl] != km && (B.tslot >> sl & 1)) L2[pos] = ((uint64_t)ovf_hash(km) & omask) >> 3;
              const Bk& A = TA[b]; bool f = false; for (int q = 0; q < 8; ++q) f |= A.key[q] == km; if (!f && A.turned) L2A[pos] = hash_idx(km, 64 - 27); }
            { const uint64_t b2 = b & (nb / 2 - 1); const BW& B = T2[b2]; bool f = false; for (int q = 0; q < B.n[sl]; ++q) f |= B.key[sl][q] == km; if (f) ++W2_home; else if (B.turned >> sl & 1) ++W2_ovf; if (b2 != pb2) ++runs2; pb2 = b2; }
            { const uint64_t b4 = b & (nb / 4 - 1); const BW& B = T4[b4]; bool f = false; for (int q = 0; q < B.n[sl]; ++q) f |= B.key[sl][q] == km; if (f) ++W4_home; else if (B.turned >> sl & 1) ++W4_ovf; if (b4 != pb4) ++runs4; pb4 = b4; }
            { const Bk& B = TA[b]; bool f = false; for (int q = 0; q < 8; ++q) f |= B.key[q] == km; if (f) ++A_home; else if (B.turned) { ++A_ovf; aovf = true; } else ++A_miss; }
            { const Bk& B = TB[b]; if (B.key[sl] == km) ++B_home; else if (B.tslot >> sl & 1) { ++B_ovf; bovf = true; } else ++B_miss; }
            { const Bk& B = TC[b]; if (B.key[sl] == km) ++C_home; else if ((B.displaced >> sl & 1) && B.key[7] == km) ++C_spare; else if (B.tslot >> sl & 1) ++C_ovf; else ++C_miss; }
            { const Bk& B = TD[b]; if (B.key[sl] == km) ++D_home; else if (B.displaced) { bool f = false; for (int q = 0; q < 8; ++q) f |= B.key[q] == km; dscan = true; if (f) ++D_scan; else if (B.turned) ++D_ovf; else ++D_miss; } else if (B.turned) ++D_ovf; else ++D_miss; }
        }
        for (int model = 0; model < 3; ++model) {
            // model 0: lane l owns positions 5l .. 5l+4, instruction j takes position 5l + j; 1: instruction r takes positions 32r .. 32r+31; 2: 64r .. 64r+63
            const int ninstr = model == 0 ? 5 : model == 1 ? 5 : 3;
            for (int ins = 0; ins < ninstr; ++ins) {
                std::vector<uint64_t> a, b2, b2a;
                for (int pos = 0; pos < 130; ++pos) {
                    const bool mine = model == 0 ? pos % 5 == ins : model == 1 ? pos / 32 == ins : pos / 64 == ins;
                    if (!mine) continue;
                    if (L1[pos] != ~0ull) a.push_back(L1[pos]);
                    if (L2[pos] != ~0ull) b2.push_back(L2[pos]);
                    if (L2A[pos] != ~0ull) b2a.push_back(L2A[pos]);
                }
                for (auto* v : {&a, &b2, &b2a}) { std::sort(v->begin(), v->end()); v->erase(std::unique(v->begin(), v->end()), v->end()); }
                req1[model] += a.size(); req2[model] += b2.size(); req2A[model] += b2a.size();
            }
        }
        ++nreads; D_scanreads += dscan; B_ovfreads += bovf; A_ovfreads += aovf;
    }
    const double n = (double)nlook;
    printf("look-ups %lu, runs per read %.1f\n", (unsigned long)nlook, (double)runs / (2 * npairs));
    for (int model = 0; model < 3; ++model)
        printf("requests per read, layout %d (%s): level 1 %.1f   level 2 offset-slotted %.1f   level 2 any-slot %.1f\n", model,
               model == 0 ? "5 consecutive positions per lane" : model == 1 ? "32 positions per instruction" : "64 positions per instruction",
               (double)req1[model] / nreads, (double)req2[model] / nreads, (double)req2A[model] / nreads);
    printf("A any-slot     : home %.4f  plain-index %.4f  miss %.4f   reads with a plain-index look-up %.3f\n", A_home / n, A_ovf / n, A_miss / n, (double)A_ovfreads / nreads);
    printf("B offset slot  : home %.4f  plain-index %.4f  miss %.4f   reads with a plain-index look-up %.3f\n", B_home / n, B_ovf / n, B_miss / n, (double)B_ovfreads / nreads);
    printf("B2 offset, 2 ways (256-B buckets, half as many): home %.4f  level 2 %.4f\n", W2_home / n, W2_ovf / n);
    printf("B4 offset, 4 ways (512-B buckets, a quarter as many): home %.4f  level 2 %.4f\n", W4_home / n, W4_ovf / n);
    printf("C offset+spare : home %.4f  spare %.4f  plain-index %.4f  miss %.4f\n", C_home / n, C_spare / n, C_ovf / n, C_miss / n);
    printf("D offset/any   : home %.4f  scan %.4f  plain-index %.4f  miss %.4f   reads that scan %.3f\n", D_home / n, D_scan / n, D_ovf / n, D_miss / n, (double)D_scanreads / nreads);
    return 0;
}
