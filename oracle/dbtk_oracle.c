/*
 * dbtk_oracle.c — CPU oracle (plain C) for the `danbing-tk align` hot path.
 *
 * TEST INFRASTRUCTURE ONLY — see dbtk_oracle.h.  Every function restates the
 * reference routine named beside it (file:line into /root/reference/src);
 * "AQ.cpp" = aQueryFasta_thread.cpp, "AQ.h" = aQueryFasta_thread.h,
 * "BIO" = binaryKmerIO.hpp, "KIO" = kmerIO.hpp.  The data structures are the
 * oracle's own (flat open-addressed index, sorted per-locus arrays); only the
 * arithmetic and the control flow follow the reference.
 *
 * Two libstdc++ behaviours are part of the reference's results and are
 * restated here from GCC 11's headers:
 *   std::sort            bits/stl_algo.h:1800-1957, bits/stl_heap.h
 *   unordered_map order  bits/hashtable.h:1888-1912,2010-2031,2380-2415 and
 *                        _Prime_rehash_policy (src/c++11/hashtable_c++0x.cc);
 *                        its prime table is read from libstdc++.so at run time.
 */
#define _GNU_SOURCE
#include "dbtk_oracle.h"

#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NAN64 DBTK_NAN64
#define NAN32 DBTK_NAN32

/* ------------------------------------------------------------------ tables */
/* baseNumConversion / alphabet, AQ.h:52-69: only 'A','C','G','T' are bases. */
static inline int base_code(uint8_t c) {
    switch (c) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return -1;
    }
}

/* getNuRC, AQ.h:165-178 (byteRC LUT at AQ.h:88-114 == reverse the 2-bit
 * symbols of a byte and complement them). */
static inline uint8_t byte_rc(uint8_t b) {
    uint8_t r = (uint8_t)(((b & 3u) << 6) | ((b & 0xCu) << 2) | ((b & 0x30u) >> 2) | ((b & 0xC0u) >> 6));
    return (uint8_t)~r;
}

uint64_t orc_nurc(uint64_t num, uint32_t k) {
    uint64_t rc = 0;
    while (k >= 4) {
        rc <<= 8;
        rc += byte_rc((uint8_t)(num & 0xff));
        num >>= 8;
        k -= 4;
    }
    if (k > 0) {
        rc <<= (k << 1);
        rc += (uint64_t)(byte_rc((uint8_t)num) >> ((4 - k) << 1));
    }
    return rc;
}

/* getNextKmer, AQ.h:134-153.  Returns rlen when no further window exists. */
static uint64_t next_kmer(uint64_t* kmer, uint64_t beg, const uint8_t* read, uint64_t rlen, uint32_t k) {
    if (beg + k > rlen) return rlen;
    uint64_t validlen = 0;
    while (validlen != k) {
        if (beg + k > rlen) return rlen;
        if (base_code(read[beg + validlen]) < 0) {
            beg = beg + validlen + 1;
            validlen = 0;
        } else {
            validlen += 1;
        }
    }
    uint64_t v = 0; /* encodeSeq, AQ.h:126-132 */
    for (uint64_t i = beg; i < beg + k; ++i) v = (v << 2) + (uint64_t)base_code(read[i]);
    *kmer = v;
    return beg;
}

/* read2kmers_edges, AQ.h:274-311.  kmers must hold rlen-k+1 and edges rlen-k
 * entries.  Returns the size of the kmers vector: 0 (vectors left empty) or
 * rlen-k+1. */
uint64_t orc_read2kmers_edges(const uint8_t* read, uint64_t rlen, uint32_t k, uint64_t* kmers, uint64_t* edges) {
    const uint64_t mask = (1ULL << 2 * (k - 1)) - 1;
    uint64_t kmer = 0, rckmer, kmer_ = NAN64, rckmer_ = NAN64;
    uint64_t beg = next_kmer(&kmer, 0, read, rlen, k);
    if (beg == rlen) return 0;
    for (uint64_t i = 0; i < rlen - k + 1; ++i) kmers[i] = NAN64;
    for (uint64_t i = 0; i + k < rlen; ++i) edges[i] = NAN64;
    rckmer = orc_nurc(kmer, k);
    for (uint64_t i = beg; i < rlen - k + 1; ++i) {
        kmers[i] = kmer < rckmer ? kmer : rckmer;
        if (kmer_ != NAN64) {
            uint64_t edge = (kmer_ << 2) + (kmer % 4);
            uint64_t rcedge = (rckmer << 2) + (rckmer_ % 4);
            edges[i - 1] = edge < rcedge ? edge : rcedge;
        }
        /* read[i+k] at i+k == rlen is the string's terminating '\0' */
        int c = (i + k < rlen) ? base_code(read[i + k]) : -1;
        if (c < 0) {
            uint64_t nbeg = next_kmer(&kmer, i + k + 1, read, rlen, k);
            if (nbeg == rlen) return rlen - k + 1;
            rckmer = orc_nurc(kmer, k);
            i = nbeg - 1;
            kmer_ = NAN64;
            rckmer_ = NAN64;
        } else {
            kmer_ = kmer;
            rckmer_ = rckmer;
            kmer = ((kmer & mask) << 2) + (uint64_t)c;
            rckmer = (rckmer >> 2) + (((uint64_t)(3 - c) & mask) << (2 * (k - 1)));
        }
    }
    return rlen - k + 1;
}

/* --------------------------------------------------------- std::sort (GCC) */
/* Comparator of getSortedIndex(vector<uint64_t>&, ...), AQ.cpp:247-250. */
typedef struct { const uint64_t* d; } cmp_t;
static inline int lt(const cmp_t* c, uint64_t a, uint64_t b) { return c->d[a] < c->d[b]; }

static void s_unguarded_linear_insert(uint64_t* last, const cmp_t* c) {
    uint64_t val = *last;
    uint64_t* next = last - 1;
    while (lt(c, val, *next)) { *last = *next; last = next; --next; }
    *last = val;
}
static void s_insertion_sort(uint64_t* first, uint64_t* last, const cmp_t* c) {
    if (first == last) return;
    for (uint64_t* i = first + 1; i != last; ++i) {
        if (lt(c, *i, *first)) {
            uint64_t val = *i;
            memmove(first + 1, first, (size_t)(i - first) * sizeof(uint64_t));
            *first = val;
        } else {
            s_unguarded_linear_insert(i, c);
        }
    }
}
static void s_push_heap(uint64_t* first, long hole, long top, uint64_t value, const cmp_t* c) {
    long parent = (hole - 1) / 2;
    while (hole > top && lt(c, first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}
static void s_adjust_heap(uint64_t* first, long hole, long len, uint64_t value, const cmp_t* c) {
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (lt(c, first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    s_push_heap(first, hole, top, value, c);
}
static void s_heapsort(uint64_t* first, uint64_t* last, const cmp_t* c) { /* __partial_sort(first,last,last) */
    long len = last - first;
    if (len >= 2) { /* __make_heap */
        long parent = (len - 2) / 2;
        for (;;) {
            uint64_t v = first[parent];
            s_adjust_heap(first, parent, len, v, c);
            if (parent == 0) break;
            parent--;
        }
    }
    while (last - first > 1) { /* __sort_heap */
        --last;
        uint64_t v = *last;
        *last = *first;
        s_adjust_heap(first, 0, last - first, v, c);
    }
}
static void s_move_median_to_first(uint64_t* result, uint64_t* a, uint64_t* b, uint64_t* cc, const cmp_t* c) {
    uint64_t t;
#define ISWAP(x, y) do { t = *(x); *(x) = *(y); *(y) = t; } while (0)
    if (lt(c, *a, *b)) {
        if (lt(c, *b, *cc)) ISWAP(result, b);
        else if (lt(c, *a, *cc)) ISWAP(result, cc);
        else ISWAP(result, a);
    } else if (lt(c, *a, *cc)) ISWAP(result, a);
    else if (lt(c, *b, *cc)) ISWAP(result, cc);
    else ISWAP(result, b);
}
static uint64_t* s_unguarded_partition(uint64_t* first, uint64_t* last, uint64_t* pivot, const cmp_t* c) {
    uint64_t t;
    for (;;) {
        while (lt(c, *first, *pivot)) ++first;
        --last;
        while (lt(c, *pivot, *last)) --last;
        if (!(first < last)) return first;
        ISWAP(first, last);
        ++first;
    }
}
static void s_introsort_loop(uint64_t* first, uint64_t* last, long depth, const cmp_t* c) {
    while (last - first > 16) {
        if (depth == 0) { s_heapsort(first, last, c); return; }
        --depth;
        uint64_t* mid = first + (last - first) / 2;
        s_move_median_to_first(first, first + 1, mid, last - 1, c);
        uint64_t* cut = s_unguarded_partition(first + 1, last, first, c);
        s_introsort_loop(cut, last, depth, c);
        last = cut;
    }
}
void orc_sort_index(const uint64_t* data, uint64_t n, uint64_t* idx) {
    for (uint64_t i = 0; i < n; ++i) idx[i] = i; /* std::iota */
    if (n == 0) return;
    cmp_t c = { data };
    long lg = 63 - __builtin_clzll(n); /* std::__lg */
    s_introsort_loop(idx, idx + n, lg * 2, &c);
    if (n > 16) { /* __final_insertion_sort */
        s_insertion_sort(idx, idx + 16, &c);
        for (uint64_t* i = idx + 16; i != idx + n; ++i) s_unguarded_linear_insert(i, &c);
    } else {
        s_insertion_sort(idx, idx + n, &c);
    }
}

/* ------------------------------------------- std::unordered_map iteration */
static const unsigned long* prime_list(void) {
    static const unsigned long* pl = NULL;
    if (!pl) {
        void* h = dlopen("libstdc++.so.6", RTLD_NOW | RTLD_GLOBAL);
        if (h) pl = (const unsigned long*)dlsym(h, "_ZNSt8__detail12__prime_listE");
    }
    return pl;
}
typedef struct { size_t next_resize; } rehash_policy_t; /* max_load_factor == 1.0f */
static size_t pol_next_bkt(rehash_policy_t* p, size_t n) {
    static const unsigned char fast_bkt[] = { 2, 2, 2, 3, 5, 5, 7, 7, 11, 11, 11, 11, 13, 13 };
    if (n < sizeof(fast_bkt)) {
        if (n == 0) return 1;
        p->next_resize = (size_t)floor(fast_bkt[n] * 1.0);
        return fast_bkt[n];
    }
    const unsigned long* pl = prime_list();
    const size_t n_primes = 256 + 48; /* 64-bit table, without the sentinel */
    const unsigned long* lo = pl + 6;
    const unsigned long* last_prime = pl + n_primes - 1;
    const unsigned long* hi = last_prime;
    while (lo < hi) { /* std::lower_bound */
        const unsigned long* mid = lo + (hi - lo) / 2;
        if (*mid < n) lo = mid + 1; else hi = mid;
    }
    if (lo == last_prime) p->next_resize = (size_t)-1;
    else p->next_resize = (size_t)floor(*lo * 1.0);
    return *lo;
}
static int pol_need_rehash(rehash_policy_t* p, size_t n_bkt, size_t n_elt, size_t n_ins, size_t* out) {
    if (n_elt + n_ins > p->next_resize) {
        size_t a = n_elt + n_ins, b = p->next_resize ? 0 : 11;
        double min_bkts = (double)(a > b ? a : b) / 1.0;
        if (min_bkts >= (double)n_bkt) {
            size_t x = (size_t)floor(min_bkts) + 1, y = n_bkt * 2;
            *out = pol_next_bkt(p, x > y ? x : y);
            return 1;
        }
        p->next_resize = (size_t)floor(n_bkt * 1.0);
        return 0;
    }
    return 0;
}
int orc_umap_order(const uint64_t* keys, uint64_t n, uint64_t* order) {
    if (!prime_list()) return -1;
    /* nodes: singly linked list through nxt[]; index n == _M_before_begin */
    const int64_t BB = (int64_t)n;
    int64_t* nxt = (int64_t*)malloc((n + 1) * sizeof(int64_t));
    size_t nb = 1;
    int64_t* bkt = (int64_t*)malloc(sizeof(int64_t)); /* "before" node of each bucket, -1 = empty */
    bkt[0] = -1;
    nxt[BB] = -1;
    rehash_policy_t pol = { 0 };
    size_t cnt = 0;
    for (uint64_t e = 0; e < n; ++e) {
        /* operator[]: find first (duplicate keys do not insert) */
        int dup = 0;
        {
            size_t b = keys[e] % nb;
            if (bkt[b] >= 0) {
                int64_t prev = bkt[b];
                for (int64_t p = nxt[prev]; p >= 0; p = nxt[p]) {
                    if (keys[p] == keys[e]) { dup = 1; break; }
                    if (nxt[p] < 0 || keys[nxt[p]] % nb != b) break;
                }
            }
        }
        if (dup) { nxt[e] = -2; continue; }
        size_t newnb;
        if (pol_need_rehash(&pol, nb, cnt, 1, &newnb)) { /* _M_rehash_aux(unique), hashtable.h:2380 */
            int64_t* nbk = (int64_t*)malloc(newnb * sizeof(int64_t));
            for (size_t i = 0; i < newnb; ++i) nbk[i] = -1;
            int64_t p = nxt[BB];
            nxt[BB] = -1;
            size_t bbegin = 0;
            while (p >= 0) {
                int64_t next = nxt[p];
                size_t b = keys[p] % newnb;
                if (nbk[b] < 0) {
                    nxt[p] = nxt[BB];
                    nxt[BB] = p;
                    nbk[b] = BB;
                    if (nxt[p] >= 0) nbk[bbegin] = p;
                    bbegin = b;
                } else {
                    nxt[p] = nxt[nbk[b]];
                    nxt[nbk[b]] = p;
                }
                p = next;
            }
            free(bkt);
            bkt = nbk;
            nb = newnb;
        }
        size_t b = keys[e] % nb; /* _M_insert_bucket_begin, hashtable.h:1888 */
        if (bkt[b] >= 0) {
            nxt[e] = nxt[bkt[b]];
            nxt[bkt[b]] = (int64_t)e;
        } else {
            nxt[e] = nxt[BB];
            nxt[BB] = (int64_t)e;
            if (nxt[e] >= 0) bkt[keys[nxt[e]] % nb] = (int64_t)e;
            bkt[b] = BB;
        }
        ++cnt;
    }
    uint64_t j = 0;
    for (int64_t p = nxt[BB]; p >= 0; p = nxt[p]) order[j++] = (uint64_t)p;
    for (; j < n; ++j) order[j] = NAN64; /* duplicates in the input: fewer elements than keys */
    free(nxt);
    free(bkt);
    return 0;
}

/* ------------------------------------------------------------------- RPGG */
typedef struct { uint64_t k, i; } ki_t;
struct orc_rpgg {
    uint32_t k;
    uint64_t nloci;
    /* kmerDBi (AQ.h:654-673): open-addressed, last assignment wins like operator[] */
    uint64_t cap;      /* power of two */
    uint64_t* hkeys;   /* NAN64 = empty */
    uint32_t* hvals;
    uint64_t nvv;
    uint32_t* vv;
    /* flankDB[l] / trKmers[l] as sorted arrays (AQ.h:675-698, 469-480) */
    uint64_t* fl_beg;  /* nloci+1 */
    uint64_t* fl_ks;   /* sorted within locus */
    uint64_t* tr_beg;  /* nloci+1 */
    uint64_t* tr_cnt;  /* nloci (as given) */
    uint64_t* tr_ks_file; /* file order */
    uint64_t* tr_ks;   /* sorted within locus */
    uint64_t* tr_fi;   /* file-order index (global) of tr_ks[i] */
    uint64_t ntr;
    uint64_t* tre_beg;
    uint64_t* tre_ks;  /* sorted within locus, may be NULL */
    uint8_t* qc;
    /* baitDB[l] (AQ.h:542-547): k-mers sorted within locus + thresholds */
    uint64_t* bt_beg;
    ki_t* bt;          /* .k = k-mer, .i = (min << 8 | max) */
    /* graphDB[l] (AQ.h:32, 550-575): nodes sorted within locus + out-edge masks */
    uint64_t* gr_beg;
    uint64_t* gr_ks;
    uint8_t* gr_ms;
};

static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
static inline int64_t idx_find(const orc_rpgg_t* g, uint64_t key) {
    if (key == NAN64) return -1;
    uint64_t m = g->cap - 1, i = mix64(key) & m;
    while (g->hkeys[i] != NAN64) {
        if (g->hkeys[i] == key) return (int64_t)i;
        i = (i + 1) & m;
    }
    return -1;
}
static int cmp_u64(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : x > y;
}
static int cmp_ki(const void* a, const void* b) {
    const ki_t* x = (const ki_t*)a; const ki_t* y = (const ki_t*)b;
    if (x->k != y->k) return x->k < y->k ? -1 : 1;
    return x->i < y->i ? -1 : x->i > y->i;
}
static int bs_has(const uint64_t* a, uint64_t lo, uint64_t hi, uint64_t key, uint64_t* pos) {
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    if (pos) *pos = lo;
    return 0;
}

orc_rpgg_t* orc_rpgg_from_arrays(const dbtk_rpgg_arrays_t* a) {
    orc_rpgg_t* g = (orc_rpgg_t*)calloc(1, sizeof(*g));
    g->k = a->ksize;
    g->nloci = a->nloci;
    uint64_t cap = 16;
    while (cap < a->nkeys * 2 + 2) cap <<= 1;
    g->cap = cap;
    g->hkeys = (uint64_t*)malloc(cap * 8);
    g->hvals = (uint32_t*)malloc(cap * 4);
    memset(g->hkeys, 0xff, cap * 8);
    for (uint64_t i = 0; i < a->nkeys; ++i) { /* kmerDBi[key] = val, AQ.h:671 */
        uint64_t key = a->keys[i], m = cap - 1, j = mix64(key) & m;
        while (g->hkeys[j] != NAN64 && g->hkeys[j] != key) j = (j + 1) & m;
        g->hkeys[j] = key;
        g->hvals[j] = a->vals[i];
    }
    g->nvv = a->nvv;
    g->vv = (uint32_t*)malloc((a->nvv + 1) * 4);
    if (a->nvv) memcpy(g->vv, a->vv, a->nvv * 4);
    /* flank sets */
    g->fl_beg = (uint64_t*)malloc((a->nloci + 1) * 8);
    g->fl_beg[0] = 0;
    for (uint64_t l = 0; l < a->nloci; ++l) g->fl_beg[l + 1] = g->fl_beg[l] + a->fl_cnt[l];
    uint64_t nfl = g->fl_beg[a->nloci];
    g->fl_ks = (uint64_t*)malloc((nfl + 1) * 8);
    if (nfl) memcpy(g->fl_ks, a->fl_ks, nfl * 8);
    for (uint64_t l = 0; l < a->nloci; ++l)
        qsort(g->fl_ks + g->fl_beg[l], (size_t)a->fl_cnt[l], 8, cmp_u64);
    /* TR maps */
    g->tr_beg = (uint64_t*)malloc((a->nloci + 1) * 8);
    g->tr_cnt = (uint64_t*)malloc((a->nloci + 1) * 8);
    g->tr_beg[0] = 0;
    for (uint64_t l = 0; l < a->nloci; ++l) { g->tr_cnt[l] = a->tr_cnt[l]; g->tr_beg[l + 1] = g->tr_beg[l] + a->tr_cnt[l]; }
    g->ntr = g->tr_beg[a->nloci];
    g->tr_ks_file = (uint64_t*)malloc((g->ntr + 1) * 8);
    g->tr_ks = (uint64_t*)malloc((g->ntr + 1) * 8);
    g->tr_fi = (uint64_t*)malloc((g->ntr + 1) * 8);
    if (g->ntr) memcpy(g->tr_ks_file, a->tr_ks, g->ntr * 8);
    {
        uint64_t mx = 0;
        for (uint64_t l = 0; l < a->nloci; ++l) if (a->tr_cnt[l] > mx) mx = a->tr_cnt[l];
        ki_t* tmp = (ki_t*)malloc((mx + 1) * sizeof(ki_t));
        for (uint64_t l = 0; l < a->nloci; ++l) {
            uint64_t b = g->tr_beg[l], n = a->tr_cnt[l];
            for (uint64_t i = 0; i < n; ++i) { tmp[i].k = a->tr_ks[b + i]; tmp[i].i = b + i; }
            qsort(tmp, (size_t)n, sizeof(ki_t), cmp_ki);
            for (uint64_t i = 0; i < n; ++i) { g->tr_ks[b + i] = tmp[i].k; g->tr_fi[b + i] = tmp[i].i; }
        }
        free(tmp);
    }
    if (a->tre_cnt) {
        g->tre_beg = (uint64_t*)malloc((a->nloci + 1) * 8);
        g->tre_beg[0] = 0;
        for (uint64_t l = 0; l < a->nloci; ++l) g->tre_beg[l + 1] = g->tre_beg[l] + a->tre_cnt[l];
        uint64_t ne = g->tre_beg[a->nloci];
        g->tre_ks = (uint64_t*)malloc((ne + 1) * 8);
        if (ne) memcpy(g->tre_ks, a->tre_ks, ne * 8);
        for (uint64_t l = 0; l < a->nloci; ++l)
            qsort(g->tre_ks + g->tre_beg[l], (size_t)a->tre_cnt[l], 8, cmp_u64);
    }
    if (a->qc) {
        g->qc = (uint8_t*)malloc(a->nloci + 1);
        memcpy(g->qc, a->qc, a->nloci);
    }
    if (a->gr_cnt) orc_rpgg_set_graph(g, a->gr_cnt, a->gr_ks, a->gr_ms);
    return g;
}

static void* read_all(FILE* f, size_t n) {
    void* p = malloc(n ? n : 1);
    if (n && fread(p, 1, n, f) != n) { free(p); return NULL; }
    return p;
}
/* readBinaryKmerSetDB, AQ.h:675-698 */
static int read_kdb(const char* fn, uint64_t nloci_expect, uint64_t** cnt, uint64_t** ks) {
    FILE* f = fopen(fn, "rb");
    if (!f) return -1;
    uint64_t nloci, nk;
    if (fread(&nloci, 8, 1, f) != 1 || nloci != nloci_expect) { fclose(f); return -1; }
    *cnt = (uint64_t*)read_all(f, nloci * 8);
    if (!*cnt || fread(&nk, 8, 1, f) != 1) { fclose(f); return -1; }
    *ks = (uint64_t*)read_all(f, nk * 8);
    fclose(f);
    return *ks ? 0 : -1;
}

orc_rpgg_t* orc_rpgg_load(const char* prefix, uint32_t ksize, const char* qc_file) {
    char fn[4096];
    dbtk_rpgg_arrays_t a;
    memset(&a, 0, sizeof(a));
    a.ksize = ksize;
    /* countLoci KIO:33-45 + readKmersWithZeroCount AQ.h:469-480: first field of each line (stoul) */
    snprintf(fn, sizeof fn, "%s.tr.kmers", prefix);
    FILE* f = fopen(fn, "r");
    if (!f) return NULL;
    size_t capk = 1024, capl = 64, nk = 0, nl = 0;
    uint64_t* trk = (uint64_t*)malloc(capk * 8);
    uint64_t* trc = (uint64_t*)malloc(capl * 8);
    char* line = NULL; size_t lcap = 0; ssize_t len;
    while ((len = getline(&line, &lcap, f)) >= 0) {
        if (line[0] == '>') {
            if (nl == capl) { capl *= 2; trc = (uint64_t*)realloc(trc, capl * 8); }
            trc[nl++] = 0;
        } else {
            if (nk == capk) { capk *= 2; trk = (uint64_t*)realloc(trk, capk * 8); }
            trk[nk++] = strtoull(line, NULL, 10);
            trc[nl - 1]++;
        }
    }
    free(line);
    fclose(f);
    a.nloci = nl; a.tr_cnt = trc; a.tr_ks = trk;
    /* readBinaryIndex AQ.h:654-673 */
    snprintf(fn, sizeof fn, "%s.kmers.dbi", prefix);
    f = fopen(fn, "rb");
    if (!f) return NULL;
    uint64_t nkeys, nvv;
    if (fread(&nkeys, 8, 1, f) != 1) return NULL;
    uint64_t* keys = (uint64_t*)read_all(f, nkeys * 8);
    uint32_t* vals = (uint32_t*)read_all(f, nkeys * 4);
    if (!keys || !vals || fread(&nvv, 8, 1, f) != 1) return NULL;
    uint32_t* vv = (uint32_t*)read_all(f, nvv * 4);
    fclose(f);
    if (!vv) return NULL;
    a.nkeys = nkeys; a.keys = keys; a.vals = vals; a.nvv = nvv; a.vv = vv;
    uint64_t *flc = NULL, *flk = NULL, *trec = NULL, *trek = NULL;
    snprintf(fn, sizeof fn, "%s.fl.kdb", prefix);
    if (read_kdb(fn, nl, &flc, &flk)) return NULL;
    snprintf(fn, sizeof fn, "%s.tre.kdb", prefix);
    if (read_kdb(fn, nl, &trec, &trek) == 0) { a.tre_cnt = trec; a.tre_ks = trek; }
    a.fl_cnt = flc; a.fl_ks = flk;
    uint8_t* qc = NULL;
    if (qc_file) { /* readQCFile KIO:111-120 */
        f = fopen(qc_file, "rb");
        if (!f) return NULL;
        qc = (uint8_t*)read_all(f, nl);
        fclose(f);
        if (!qc) return NULL;
        for (uint64_t i = 0; i < nl; ++i) qc[i] = (uint8_t)(qc[i] - 48);
        a.qc = qc;
    }
    orc_rpgg_t* g = orc_rpgg_from_arrays(&a);
    free(trk); free(trc); free(keys); free(vals); free(vv); free(flc); free(flk); free(trec); free(trek); free(qc);
    return g;
}

void orc_rpgg_set_bait(orc_rpgg_t* g, const uint64_t* bt_cnt, const uint64_t* bt_ks, const uint16_t* bt_vs) {
    free(g->bt_beg); free(g->bt);
    g->bt_beg = (uint64_t*)malloc((g->nloci + 1) * 8);
    g->bt_beg[0] = 0;
    for (uint64_t l = 0; l < g->nloci; ++l) g->bt_beg[l + 1] = g->bt_beg[l] + bt_cnt[l];
    const uint64_t n = g->bt_beg[g->nloci];
    g->bt = (ki_t*)malloc((n + 1) * sizeof(ki_t));
    for (uint64_t i = 0; i < n; ++i) { g->bt[i].k = bt_ks[i]; g->bt[i].i = bt_vs[i]; }
    for (uint64_t l = 0; l < g->nloci; ++l) qsort(g->bt + g->bt_beg[l], (size_t)bt_cnt[l], sizeof(ki_t), cmp_ki);
}
/* deserializeKmapDB, BIO:70-98: u64 nloci | u64 cnt[nloci] | u64 nk | u64 sizeofval | u64 ks[nk] | u16 vs[nk] */
int orc_rpgg_load_bait(orc_rpgg_t* g, const char* fn) {
    FILE* f = fopen(fn, "rb");
    if (!f) return -1;
    uint64_t nl, nk, szv;
    if (fread(&nl, 8, 1, f) != 1 || nl != g->nloci) { fclose(f); return -1; }
    uint64_t* cnt = (uint64_t*)read_all(f, nl * 8);
    if (!cnt || fread(&nk, 8, 1, f) != 1 || fread(&szv, 8, 1, f) != 1 || szv != 2) { fclose(f); return -1; }
    uint64_t* ks = (uint64_t*)read_all(f, nk * 8);
    uint16_t* vs = (uint16_t*)read_all(f, nk * 2);
    fclose(f);
    if (!ks || !vs) return -1;
    orc_rpgg_set_bait(g, cnt, ks, vs);
    free(cnt); free(ks); free(vs);
    return 0;
}

/* graphDB: kmerDB[idx][kmer] |= c (readGraphKmers, AQ.h:550-575): duplicates of a node OR their masks */
typedef struct { uint64_t k; uint8_t m; } km_t;
static int cmp_km(const void* a, const void* b) {
    const km_t* x = (const km_t*)a; const km_t* y = (const km_t*)b;
    return x->k < y->k ? -1 : x->k > y->k;
}
void orc_rpgg_set_graph(orc_rpgg_t* g, const uint64_t* gr_cnt, const uint64_t* gr_ks, const uint8_t* gr_ms) {
    free(g->gr_beg); free(g->gr_ks); free(g->gr_ms);
    uint64_t n = 0, mx = 0;
    for (uint64_t l = 0; l < g->nloci; ++l) { n += gr_cnt[l]; if (gr_cnt[l] > mx) mx = gr_cnt[l]; }
    g->gr_beg = (uint64_t*)malloc((g->nloci + 1) * 8);
    g->gr_ks = (uint64_t*)malloc((n + 1) * 8);
    g->gr_ms = (uint8_t*)malloc(n + 1);
    km_t* tmp = (km_t*)malloc((mx + 1) * sizeof(km_t));
    uint64_t src = 0, dst = 0;
    for (uint64_t l = 0; l < g->nloci; ++l) {
        const uint64_t c = gr_cnt[l];
        for (uint64_t i = 0; i < c; ++i) { tmp[i].k = gr_ks[src + i]; tmp[i].m = gr_ms[src + i]; }
        qsort(tmp, (size_t)c, sizeof(km_t), cmp_km);
        g->gr_beg[l] = dst;
        for (uint64_t i = 0; i < c; ++i) {
            if (dst > g->gr_beg[l] && g->gr_ks[dst - 1] == tmp[i].k) g->gr_ms[dst - 1] |= tmp[i].m;
            else { g->gr_ks[dst] = tmp[i].k; g->gr_ms[dst] = tmp[i].m; ++dst; }
        }
        src += c;
    }
    g->gr_beg[g->nloci] = dst;
    free(tmp);
}
/* PREF.graph.kmers (text, AQ.h:550-575) or, when the name ends in ".umap", the v1.3 binary
 * `u64 nloci | per locus: u64 n | n x (u64 node, u8 mask)`. */
int orc_rpgg_load_graph(orc_rpgg_t* g, const char* fn) {
    const size_t fl = strlen(fn);
    const int binary = fl > 5 && strcmp(fn + fl - 5, ".umap") == 0;
    FILE* f = fopen(fn, binary ? "rb" : "r");
    if (!f) return -1;
    uint64_t* cnt = (uint64_t*)calloc(g->nloci + 1, 8);
    uint64_t cap = 1 << 16, n = 0;
    uint64_t* ks = (uint64_t*)malloc(cap * 8);
    uint8_t* ms = (uint8_t*)malloc(cap);
    int rc = 0;
#define GR_PUSH(K, M) do { if (n == cap) { cap *= 2; ks = (uint64_t*)realloc(ks, cap * 8); ms = (uint8_t*)realloc(ms, cap); } ks[n] = (K); ms[n] = (M); ++n; } while (0)
    if (binary) {
        uint64_t nl;
        if (fread(&nl, 8, 1, f) != 1 || nl != g->nloci) rc = -1;
        for (uint64_t l = 0; !rc && l < nl; ++l) {
            uint64_t c;
            if (fread(&c, 8, 1, f) != 1) { rc = -1; break; }
            cnt[l] = c;
            for (uint64_t i = 0; i < c; ++i) {
                uint64_t node; uint8_t m;
                if (fread(&node, 8, 1, f) != 1 || fread(&m, 1, 1, f) != 1) { rc = -1; break; }
                GR_PUSH(node, m);
            }
        }
    } else {
        char line[256];
        int64_t l = -1;
        while (fgets(line, sizeof line, f)) {
            if (line[0] == '>') { ++l; continue; }
            if (l < 0 || (uint64_t)l >= g->nloci) { rc = -1; break; }
            char* tab = NULL;
            const uint64_t node = strtoull(line, &tab, 10);
            const uint64_t m = (tab && *tab == '\t') ? strtoull(tab + 1, NULL, 10) : 0;
            GR_PUSH(node, (uint8_t)m);
            ++cnt[l];
        }
    }
#undef GR_PUSH
    fclose(f);
    if (!rc) orc_rpgg_set_graph(g, cnt, ks, ms);
    free(cnt); free(ks); free(ms);
    return rc;
}
int orc_rpgg_has_graph(const orc_rpgg_t* g) { return g->gr_beg != NULL; }
/* graphDB[locus] as (node, mask), ascending nodes: what the loaders made of a file (test accessor) */
uint64_t orc_rpgg_graph_dump(const orc_rpgg_t* g, uint32_t locus, uint64_t* ks, uint8_t* ms, uint64_t cap) {
    if (!g->gr_beg || locus >= g->nloci) return 0;
    const uint64_t b = g->gr_beg[locus], n = g->gr_beg[locus + 1] - b;
    for (uint64_t i = 0; i < n && i < cap; ++i) { ks[i] = g->gr_ks[b + i]; ms[i] = g->gr_ms[b + i]; }
    return n;
}

void orc_rpgg_free(orc_rpgg_t* g) {
    if (!g) return;
    free(g->bt_beg); free(g->bt);
    free(g->gr_beg); free(g->gr_ks); free(g->gr_ms);
    free(g->hkeys); free(g->hvals); free(g->vv); free(g->fl_beg); free(g->fl_ks); free(g->tr_beg);
    free(g->tr_cnt); free(g->tr_ks_file); free(g->tr_ks); free(g->tr_fi); free(g->tre_beg); free(g->tre_ks); free(g->qc);
    free(g);
}
uint64_t orc_rpgg_nloci(const orc_rpgg_t* g) { return g->nloci; }
uint64_t orc_rpgg_ntrkmers(const orc_rpgg_t* g) { return g->ntr; }
const uint64_t* orc_rpgg_tr_cnt(const orc_rpgg_t* g) { return g->tr_cnt; }
const uint64_t* orc_rpgg_tr_ks(const orc_rpgg_t* g) { return g->tr_ks_file; }

static inline int fl_has(const orc_rpgg_t* g, uint64_t locus, uint64_t km) {
    uint64_t pos;
    bs_has(g->fl_ks, g->fl_beg[locus], g->fl_beg[locus + 1], km, &pos);
    return pos < g->fl_beg[locus + 1] && g->fl_ks[pos] == km;
}
static inline int64_t tr_find(const orc_rpgg_t* g, uint64_t locus, uint64_t km) {
    uint64_t pos;
    bs_has(g->tr_ks, g->tr_beg[locus], g->tr_beg[locus + 1], km, &pos);
    if (pos < g->tr_beg[locus + 1] && g->tr_ks[pos] == km) {
        /* duplicate lines in tr.kmers collapse onto ONE map node (operator[]):
         * use the first file occurrence as its identity */
        return (int64_t)g->tr_fi[pos];
    }
    return -1;
}

#include "dbtk_oracle_walk.c" /* the graph walk (isThreadFeasible & co.), same translation unit */

/* -------------------------------------------------------------- hot path */
typedef struct { uint64_t key; uint32_t val; } hit_t; /* what an index iterator dereferences to */
typedef struct { uint8_t first, second; } pe_kmc_t;   /* PE_KMC, AQ.cpp:42 */
typedef struct { uint64_t idx, fc, rc; } asgn_t;      /* AQ.cpp:146-149 */

/* subfilter, AQ.cpp:172-188.  *nprobe counts lookups actually performed. */
static int subfilter(const orc_rpgg_t* g, const dbtk_params_t* p, const uint64_t* k1, uint64_t L1,
                     const uint64_t* k2, uint64_t L2, uint64_t* nhash, uint64_t* nprobe) {
    uint64_t NF = p->n_filter, NM = p->nm_filter;
    uint64_t S1 = L1 / (NF - 1), S2 = L2 / (NF - 1);
    uint64_t h1 = 0, h2 = 0;
    for (uint64_t i = 0; i < NF; ++i, ++*nhash) {
        uint64_t i1 = (i != NF - 1 ? i * S1 : L1 - 1);
        h1 += idx_find(g, k1[i1]) >= 0;
        ++*nprobe;
        if (h1 >= NM) break;
    }
    if (h1 < NM) return 1;
    for (uint64_t i = 0; i < NF; ++i, ++*nhash) {
        uint64_t i2 = (i != NF - 1 ? i * S2 : L2 - 1);
        h2 += idx_find(g, k2[i2]) >= 0;
        ++*nprobe;
        if (h2 >= NM) break;
    }
    return h2 < NM;
}

/* kfilter, AQ.cpp:190-224 (one mate). */
static void kfilter_mate(const orc_rpgg_t* g, const uint64_t* ks, uint64_t nk, uint32_t cth, hit_t* its,
                         uint64_t* nits, uint64_t* nhash, int* kf, int* rm) {
    const uint64_t MAX_NS = nk - cth;
    uint64_t ns = 0, si = 0;
    for (; si < nk; ++si) {
        ++*nhash;
        int64_t s = idx_find(g, ks[si]);
        if (s < 0) { ++ns; if (ns > MAX_NS) { *nits = 0; break; } }
        else { its[*nits].key = g->hkeys[s]; its[*nits].val = g->hvals[s]; ++*nits; }
    }
    *kf = (si != nk);
    *rm |= *kf;
}

typedef struct { uint64_t key; uint32_t val; uint8_t orient; } hit_o_t;
static int cmp_hit_o(const void* a, const void* b) {
    const hit_o_t* x = (const hit_o_t*)a; const hit_o_t* y = (const hit_o_t*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return (int)x->orient - (int)y->orient;
}

/* updatetop2, AQ.cpp:331-347 */
static inline void updatetop2(uint64_t cf, uint32_t ind, uint64_t cr, asgn_t* top, asgn_t* second) {
    if (cf + cr > top->fc + top->rc) {
        if (top->idx != ind) { *second = *top; top->idx = ind; }
        top->fc = cf; top->rc = cr;
    } else if (cf + cr > second->fc + second->rc) {
        if (second->idx != ind) second->idx = ind;
        second->fc = cf; second->rc = cr;
    }
}

typedef struct {
    uint32_t *hits1, *hits2; /* nloci+1 each, AQ.cpp:1858 */
    uint32_t* touched; uint64_t ntouched, touched_cap;
} hits_t;
static inline void touch(hits_t* h, uint32_t locus) {
    if (h->hits1[locus] == 0 && h->hits2[locus] == 0) {
        if (h->ntouched == h->touched_cap) {
            h->touched_cap = h->touched_cap ? h->touched_cap * 2 : 1024;
            h->touched = (uint32_t*)realloc(h->touched, h->touched_cap * 4);
        }
        h->touched[h->ntouched++] = locus;
    }
}

/* vv words find_matching_locus read in every vote since the last orc_vote_vv_reset(): the reference votes on every pair, in the order
 * std::sort left the k-mers in (AQ.cpp:364-422).  Kept beside the counters, not among them: the product proves most pairs' outcome without
 * the vote and reads none of these words for them (include/dbtk.h: DBTK_PS_VOTE_VV is its own, smaller, figure).  Not thread-safe: the
 * tests run one oracle call at a time. */
static uint64_t orc_vote_vv_words = 0;
uint64_t orc_vote_vv(void) { return orc_vote_vv_words; }
void orc_vote_vv_reset(void) { orc_vote_vv_words = 0; }

/* countHit, AQ.cpp:424-453 = fillstats (308-329) + find_matching_locus (364-422). */
static uint64_t count_hit(const orc_rpgg_t* g, hit_t* its1, uint64_t n1, hit_t* its2, uint64_t n2, hits_t* H,
                          uint32_t cth, uint64_t* tri0, int* nm1, int* nm2, int* hf1, int* hf2, int* rm1, int* rm2,
                          uint64_t* nvvw, uint64_t* nvote) {
    /* countDupRemove, AQ.cpp:257-296 */
    uint64_t n = n1 + n2;
    hit_o_t* all = (hit_o_t*)malloc((n + 1) * sizeof(hit_o_t));
    for (uint64_t i = 0; i < n1; ++i) { all[i].key = its1[i].key; all[i].val = its1[i].val; all[i].orient = 0; }
    for (uint64_t i = 0; i < n2; ++i) { all[n1 + i].key = its2[i].key; all[n1 + i].val = its2[i].val; all[n1 + i].orient = 1; }
    qsort(all, (size_t)n, sizeof(hit_o_t), cmp_hit_o); /* ties merge below: any key sort is equivalent */
    uint64_t nu = 0;
    hit_t* u = (hit_t*)malloc((n + 1) * sizeof(hit_t));
    pe_kmc_t* dup = (pe_kmc_t*)malloc((n + 1) * sizeof(pe_kmc_t));
    {
        pe_kmc_t pe = { 0, 0 };
        u[0].key = all[0].key; u[0].val = all[0].val; nu = 1;
        if (all[0].orient) ++pe.second; else ++pe.first;
        for (uint64_t i = 1; i < n; ++i) {
            if (all[i].key != u[nu - 1].key) {
                dup[nu - 1] = pe;
                pe.first = 0; pe.second = 0;
                u[nu].key = all[i].key; u[nu].val = all[i].val; ++nu;
            }
            if (all[i].orient) ++pe.second; else ++pe.first;
        }
        dup[nu - 1] = pe;
    }
    /* fillstats, AQ.cpp:311-328 */
    uint64_t* nml = (uint64_t*)malloc(nu * 8);
    uint64_t* ord = (uint64_t*)malloc(nu * 8);
    for (uint64_t i = 0; i < nu; ++i) {
        nml[i] = (u[i].val % 2) ? g->vv[u[i].val >> 1] : 1;
        *nvvw += u[i].val % 2; /* one vv word read */
    }
    orc_sort_index(nml, nu, ord);
    hit_t* su = (hit_t*)malloc(nu * sizeof(hit_t));
    pe_kmc_t* sdup = (pe_kmc_t*)malloc(nu * sizeof(pe_kmc_t));
    for (uint64_t i = 0; i < nu; ++i) { su[i] = u[ord[i]]; sdup[i] = dup[ord[i]]; }
    /* countRemain, AQ.cpp:298-306 */
    uint64_t* remain = (uint64_t*)calloc(nu, 8);
    {
        int dupsum = 0; /* std::accumulate(..., 0, ...): the accumulator is an int */
        for (uint64_t i = 0; i < nu; ++i) dupsum = (int)((uint64_t)(int64_t)dupsum + sdup[i].first + sdup[i].second);
        remain[0] = (uint64_t)(int64_t)dupsum - sdup[0].first - sdup[0].second;
        for (uint64_t i = 1; i + 1 < nu; ++i) remain[i] = remain[i - 1] - sdup[i].first - sdup[i].second;
    }
    /* find_matching_locus, AQ.cpp:364-422 */
    asgn_t top = { NAN32, 0, 0 }, second = { NAN32, 0, 0 };
    for (uint64_t i = 0; i < nu; ++i) {
        uint32_t vi = su[i].val;
        if (vi % 2) {
            uint64_t j0 = (vi >> 1) + 1, j1 = j0 + g->vv[vi >> 1];
            *nvote += 1 + (j1 - j0);
            for (; j0 < j1; ++j0) {
                uint32_t locus = g->vv[j0];
                touch(H, locus);
                H->hits1[locus] += sdup[i].first;
                H->hits2[locus] += sdup[i].second;
                updatetop2(H->hits1[locus], locus, H->hits2[locus], &top, &second);
            }
        } else {
            uint32_t locus = vi >> 1;
            touch(H, locus);
            H->hits1[locus] += sdup[i].first;
            H->hits2[locus] += sdup[i].second;
            updatetop2(H->hits1[locus], locus, H->hits2[locus], &top, &second);
        }
        if (!((top.fc + top.rc - second.fc - second.rc) < remain[i])) { /* !get_acm2, AQ.cpp:359-362 */
            uint64_t j = i;
            /* get_acm1, AQ.cpp:354-357 */
            while ((top.fc < cth && cth - top.fc <= remain[j]) || (top.rc < cth && cth - top.rc <= remain[j])) {
                if (++j >= nu) break;
                uint32_t vj = su[j].val;
                if (vj % 2) {
                    uint64_t j0 = (vj >> 1) + 1, j1 = j0 + g->vv[vj >> 1];
                    *nvote += 1;
                    for (; j0 < j1; ++j0) {
                        *nvote += 1;
                        if (g->vv[j0] == top.idx) { top.fc += sdup[j].first; top.rc += sdup[j].second; break; }
                    }
                } else if ((vj >> 1) == top.idx) {
                    top.fc += sdup[j].first; top.rc += sdup[j].second;
                }
            }
            break;
        }
    }
    for (uint64_t i = 0; i < H->ntouched; ++i) { H->hits1[H->touched[i]] = 0; H->hits2[H->touched[i]] = 0; } /* == std::fill, AQ.cpp:433-434 */
    H->ntouched = 0;
    free(all); free(u); free(dup); free(nml); free(ord); free(su); free(sdup); free(remain);

    *tri0 = top.idx;
    *nm1 = (int)top.fc;
    *nm2 = (int)top.rc;
    int test1 = (top.fc >= cth && top.rc >= cth);
    int test2 = (top.fc + top.rc) >= 2ull * cth;
    if ((test1 || test2) && top.idx != NAN32) return top.idx;
    *hf1 = 1 & !*rm1;
    *hf2 = 1 & !*rm2;
    *rm1 = 1;
    *rm2 = 1;
    return g->nloci;
}

typedef struct {
    int kf, hf, bf, qf, af, rm;
    int si, ei, nt, bs, ti, si_, ei_;
    int nas;              /* as.size() */
    int as[DBTK_MAX_READ_LEN];
    int64_t its[DBTK_MAX_READ_LEN]; /* file-order index of the TR k-mer node, -1 = end() */
} kmr_t; /* km_asgn_read_t, AQ.cpp:93-108 */

static void kmr_init(kmr_t* r) {
    memset(r, 0, sizeof(*r));
    r->si = -1; r->ei = -1; r->nt = 0; r->bs = 0; r->ti = -1; r->si_ = -1; r->ei_ = -1;
}

/* assignTRkmc, AQ.cpp:1450-1556 */
static void assign_trkmc(const orc_rpgg_t* g, const dbtk_params_t* p, const uint64_t* kmers, int nk, uint64_t locus,
                         kmr_t* r, int* af, int* rm, int okam) {
    if (!okam && *rm) return;
    uint8_t ntr = 0;
    int s = 0, s_ = 0, s__ = 0;
    int ti2 = -1, si1 = -1, ei1 = -1, si2 = -1, ei2 = -1;
    r->nas = nk;
    for (int i = 0; i < nk; ++i) {
        uint64_t km = kmers[i];
        r->as[i] = 0;
        r->its[i] = (km == NAN64) ? -1 : tr_find(g, locus, km);
        if (km != NAN64 && fl_has(g, locus, km)) r->as[i] = 1;
        else if (r->its[i] >= 0) { r->as[i] = 2; ++ntr; }
    }
    if (*rm) { r->nt = -1; r->bs = -1; r->ti = -1; return; }
    for (int i = 0; i < nk; ++i) {
        s = r->as[i];
        if (s && s__) {
            if (s != s__) {
                ++r->nt;
                if ((uint64_t)(int64_t)r->nt > p->max_nt) { *af = 1; *rm = 1; return; }
                if (r->nt == 1) {
                    r->ti = i;
                    if (s_) { si1 = -1; ei1 = -1; }
                } else if (r->nt == 2) {
                    if (r->bs == 2) { *af = 1; *rm = 1; return; }
                    ti2 = i;
                    if (s_) { si2 = -1; ei2 = -1; }
                }
            }
        }
        if (!r->bs) { if (s) r->bs = s; }
        if (!s) {
            if (r->nt == 0) { if (!s_) ++ei1; else { si1 = i; ei1 = i + 1; } }
            if (r->nt == 1) { if (!s_) ++ei2; else { si2 = i; ei2 = i + 1; } }
        }
        s_ = s;
        if (s) s__ = s;
    }
    int ti1 = r->ti;
    if (r->nt == 0) {
        if (r->bs != 2) { *af = 1; *rm = 1; return; }
        r->si = 0; r->ei = nk; r->si_ = 0; r->ei_ = nk;
    } else if (r->nt == 1) {
        if (r->bs == 1) {
            r->si = si1 >= 0 ? (si1 + ei1) / 2 : ti1;
            r->ei = nk;
            r->si_ = si1 >= 0 ? ei1 : ti1;
            r->ei_ = nk;
        } else {
            r->si = 0;
            r->ei = si1 >= 0 ? (si1 + ei1) / 2 : ti1;
            r->si_ = 0;
            r->ei_ = si1 >= 0 ? si1 : ti1;
        }
    } else {
        if (ntr < p->nm_tr) { *af = 1; *rm = 1; return; }
        r->si = (si1 >= 0 ? (si1 + ei1) / 2 : ti1);
        r->ei = (si2 >= 0 ? (si2 + ei2) / 2 : ti2);
        r->si_ = ei1 >= 0 ? ei1 : ti1;
        r->ei_ = si2 >= 0 ? si2 : ti2;
    }
}

/* qString2qMask, AQ.h:1038-1071, statement by statement (its bounds compare the BASE index with the
 * number of K-MERS, so the scan stops early near the read end: reproduced as is). */
void orc_qstring2qmask(const uint8_t* qual, int nq, int qth, int ksize, uint8_t* qkm) {
    int nk = nq - ksize + 1;
    int qi = 0, ki = 0;
    int* qscore = (int*)malloc((size_t)(nq > 0 ? nq : 1) * sizeof(int));
    for (int i = 0; i < nq; ++i) qscore[i] = (int)qual[i] - 33;
    for (int i = 0; i < nk; ++i) qkm[i] = 0;
    while (qscore[qi] < qth) { ++qi; ++ki; if (qi >= nk) { free(qscore); return; } }
    while (qi < nk) {
        int pass = 1;
        for (int qj = qi; qi < qj + ksize; ++qi) {
            if (qscore[qi] < qth) {
                pass = 0;
                ki = qi;
                while (qscore[qi] < qth) { ++qi; ++ki; if (qi >= nk) { free(qscore); return; } }
                break;
            }
        }
        if (pass) {
            qkm[ki] = 1;
            ++ki;
            if (qi >= nk) { free(qscore); return; }
            while (qscore[qi] >= qth) {
                qkm[ki] = 1;
                ++qi; ++ki;
                if (qi >= nk) { free(qscore); return; }
            }
            ki = qi;
            while (qscore[qi] < qth) { ++qi; ++ki; if (qi >= nk) { free(qscore); return; } }
        }
    }
    free(qscore);
}

/* bfilter_FPSv1, AQ.cpp:1377-1419 (both overloads; km == NULL: every k-mer counts).  The k-mer
 * multiset is counted in uint8_t (kc8_t); any k-mer of the bait DB whose count is outside
 * [min, max] flags the mate.  NAN64 entries are keys like any other (never in the DB). */
static void bfilter(const orc_rpgg_t* g, uint64_t locus, const uint64_t* ks, int nk, const uint8_t* km, int* bf) {
    if (!nk) return;
    uint64_t* tmp = (uint64_t*)malloc((size_t)nk * 8);
    int n = 0;
    for (int i = 0; i < nk; ++i) if (!km || km[i]) tmp[n++] = ks[i];
    qsort(tmp, (size_t)n, 8, cmp_u64);
    for (int i = 0; i < n;) {
        int j = i;
        while (j < n && tmp[j] == tmp[i]) ++j;
        const uint8_t cnt = (uint8_t)(j - i);
        uint64_t lo = g->bt_beg[locus], hi = g->bt_beg[locus + 1];
        while (lo < hi) { uint64_t mid = lo + (hi - lo) / 2; if (g->bt[mid].k < tmp[i]) lo = mid + 1; else hi = mid; }
        if (lo < g->bt_beg[locus + 1] && g->bt[lo].k == tmp[i]) {
            const uint16_t th = (uint16_t)g->bt[lo].i;
            const uint8_t mi = (uint8_t)(th >> 8), ma = (uint8_t)(th & 0xff);
            if (cnt < mi || cnt > ma) { *bf = 1; break; }
        }
        i = j;
    }
    free(tmp);
}

static int tre_has(const orc_rpgg_t* g, uint64_t locus, uint64_t e) {
    uint64_t pos;
    bs_has(g->tre_ks, g->tre_beg[locus], g->tre_beg[locus + 1], e, &pos);
    return pos < g->tre_beg[locus + 1] && g->tre_ks[pos] == e;
}

static void fill_mate_rec(dbtk_mate_rec_t* m, const kmr_t* r) {
    memset(m, 0, sizeof(*m));
    m->si = (int16_t)r->si; m->ei = (int16_t)r->ei; m->si_ = (int16_t)r->si_; m->ei_ = (int16_t)r->ei_;
    m->nt = (int16_t)r->nt; m->bs = (int16_t)r->bs; m->ti = (int16_t)r->ti;
    m->kf = (uint8_t)r->kf; m->hf = (uint8_t)r->hf; m->bf = (uint8_t)r->bf; m->qf = (uint8_t)r->qf;
    m->af = (uint8_t)r->af; m->rm = (uint8_t)r->rm;
    m->nk = (uint16_t)r->nas;
    for (int i = 0; i < r->nas; ++i) m->as2[i >> 2] |= (uint8_t)((r->as[i] & 3) << (2 * (i & 3)));
}

int orc_align(const orc_rpgg_t* g, const dbtk_params_t* p, const uint8_t* seq, const uint64_t* off, uint64_t npairs,
              uint64_t* counts, uint64_t* kmc, uint32_t* nmapread, uint64_t* C, dbtk_pair_rec_t* recs) {
    uint64_t nev = 0;
    return orc_align_ex(g, p, seq, off, NULL, npairs, counts, kmc, nmapread, C, recs, NULL, 0, &nev);
}

int orc_align_ex(const orc_rpgg_t* g, const dbtk_params_t* p, const uint8_t* seq, const uint64_t* off, const uint8_t* qual,
                 uint64_t npairs, uint64_t* counts, uint64_t* kmc, uint32_t* nmapread, uint64_t* C, dbtk_pair_rec_t* recs,
                 orc_bub_event_t* ev, uint64_t evcap, uint64_t* nev) {
    return orc_align_walk(g, p, seq, off, qual, npairs, counts, kmc, nmapread, C, recs, ev, evcap, nev, NULL);
}

int orc_align_walk(const orc_rpgg_t* g, const dbtk_params_t* p, const uint8_t* seq, const uint64_t* off, const uint8_t* qual,
                   uint64_t npairs, uint64_t* counts, uint64_t* kmc, uint32_t* nmapread, uint64_t* C, dbtk_pair_rec_t* recs,
                   orc_bub_event_t* ev, uint64_t evcap, uint64_t* nev, orc_walk_out_t* walk) {
    uint64_t nBait = 0;
    if (walk) walk->n = 0;
    *nev = 0;
    if (p->bait && !g->bt_beg) return DBTK_ERR_ARG;
    if (p->bubbles && !g->tre_beg) return DBTK_ERR_ARG;
    const uint32_t k = p->ksize;
    const uint64_t nloci = g->nloci;
    const int okam = (int)p->okam;
    hits_t H;
    memset(&H, 0, sizeof(H));
    H.hits1 = (uint32_t*)calloc(nloci + 1, 4);
    H.hits2 = (uint32_t*)calloc(nloci + 1, 4);
    static const size_t ML = DBTK_MAX_READ_LEN;
    uint64_t *caks1 = malloc(ML * 8), *caks2 = malloc(ML * 8), *caes1 = malloc(ML * 8), *caes2 = malloc(ML * 8);
    hit_t *its1 = malloc(ML * sizeof(hit_t)), *its2 = malloc(ML * sizeof(hit_t));
    uint64_t nShort = 0, nSub = 0, nKf = 0, nLocus = 0, nQC = 0, nThr = 0, nFeas = 0, nAsgn = 0, nhash0 = 0, nhash1 = 0, nprobe = 0;
    uint64_t nvvw = 0, nvote = 0, ncls = 0, ninc = 0, nsurv = 0, nbases = 0;

    for (uint64_t pi = 0; pi < npairs; ++pi) {
        const uint8_t* s1 = seq + off[2 * pi];     uint64_t l1 = off[2 * pi + 1] - off[2 * pi];
        const uint8_t* s2 = seq + off[2 * pi + 1]; uint64_t l2 = off[2 * pi + 2] - off[2 * pi + 1];
        if (l1 > ML || l2 > ML) return DBTK_ERR_READ_TOO_LONG;
        nbases += l1 + l2;
        dbtk_pair_rec_t* rec = recs ? &recs[pi] : NULL;
        kmr_t r1, r2;
        kmr_init(&r1); kmr_init(&r2);
        int rm1 = 0, rm2 = 0, kf1 = 0, kf2 = 0, hf1 = 0, hf2 = 0, af1 = 0, af2 = 0, nm1 = 0, nm2 = 0, bf1 = 0, bf2 = 0;
        uint64_t destLocus = nloci, destLocus0 = NAN32;
        uint32_t stage;
        /* AQ.cpp:2035-2044 */
        uint64_t nk1 = orc_read2kmers_edges(s1, l1, k, caks1, caes1);
        uint64_t nk2 = orc_read2kmers_edges(s2, l2, k, caks2, caes2);
        if (!nk1 || !nk2) { ++nShort; stage = DBTK_STAGE_SHORT; goto emit; }
        /* AQ.cpp:2045-2051 */
        if (p->n_filter && p->nm_filter) {
            if (subfilter(g, p, caks1, nk1, caks2, nk2, &nhash0, &nprobe)) { nSub += 2; stage = DBTK_STAGE_SUBFILTER; goto emit; }
        }
        ++nsurv;
        /* kfilter, AQ.cpp:190-200, 2052-2054 */
        {
            uint64_t n1 = 0, n2 = 0, h1before = nhash1;
            kf1 = nk1 < p->cthreshold; kf2 = nk2 < p->cthreshold;
            rm1 |= kf1; rm2 |= kf2;
            if (!(rm1 && rm2)) {
                if (!rm1) kfilter_mate(g, caks1, nk1, p->cthreshold, its1, &n1, &nhash1, &kf1, &rm1);
                if (!rm2) kfilter_mate(g, caks2, nk2, p->cthreshold, its2, &n2, &nhash1, &kf2, &rm2);
            }
            nprobe += nhash1 - h1before;
            nKf += (uint64_t)(kf1 + kf2);
            if (rm1 && rm2) { stage = DBTK_STAGE_KFILTER; goto emit; }
            /* AQ.cpp:2056-2062 */
            destLocus = count_hit(g, its1, n1, its2, n2, &H, p->cthreshold, &destLocus0, &nm1, &nm2, &hf1, &hf2, &rm1, &rm2, &nvvw, &nvote);
            nLocus += (uint64_t)(hf1 + hf2);
            if (destLocus == nloci) { stage = DBTK_STAGE_LOCUS; goto emit; }
            if (p->qc && g->qc && !g->qc[destLocus]) { nQC += (uint64_t)(2 - rm1 - rm2); stage = DBTK_STAGE_QC; goto emit; }
        }
        nThr += 2;  /* AQ.cpp:2070 */
        if (p->threading == DBTK_THREADING_V13) {
            /* The v1.3 call sites the reference keeps in comments.  AQ.cpp:2072-2088: both mates are walked through
             * graphDB[destLocus] (sam.init1/2 + isThreadFeasible); the pair is kept if either walk is feasible, else
             * destLocus = nloci.  AQ.cpp:2090-2092: nFeasibleReads += 2.  AQ.cpp:2189-2194 (countMode 0, "exact"): the
             * canonical multiset of the UNcorrected k-mers of both mates (noncaVec2CaUmap, AQ.h:392-399) is added to
             * the locus' TR k-mers.  AQ.cpp:2232-2240: with -a every walked pair yields an alignment record, with -ae
             * only the kept ones. */
            if (!g->gr_beg) return DBTK_ERR_ARG;
            dbtk_thread_rec_t* t1 = NULL; dbtk_thread_rec_t* t2 = NULL;
            if (walk && walk->trecs && walk->n < walk->cap) { t1 = &walk->trecs[2 * walk->n]; t2 = t1 + 1; }
            uint64_t nonca1[DBTK_MAX_READ_LEN], nonca2[DBTK_MAX_READ_LEN];
            const uint64_t locus0 = destLocus;
            const int alned0 = orc_thread(g, destLocus, s1, l1, k, p->thread_cth, (int)p->correction, p->maxncorrection, t1, nonca1);
            const int alned1 = orc_thread(g, destLocus, s2, l2, k, p->thread_cth, (int)p->correction, p->maxncorrection, t2, nonca2);
            if (alned0 < 0 || alned1 < 0) return DBTK_ERR_FORMAT; /* the reference would have asserted (unclean graph) */
            stage = DBTK_STAGE_THREADING;
            if (alned0 || alned1) {
                nFeas += 2;
                for (int m = 0; m < 2; ++m) {
                    const uint64_t* nk = m ? nonca2 : nonca1;
                    const uint64_t n = m ? nk2 : nk1;
                    for (uint64_t i = 0; i < n; ++i) {
                        if (nk[i] == NAN64) continue;
                        const uint64_t rc = orc_nurc(nk[i], k);
                        const int64_t it = tr_find(g, destLocus, nk[i] <= rc ? nk[i] : rc);
                        if (it >= 0) { ++counts[it]; ++ninc; }
                    }
                }
            } else destLocus = nloci; /* removed by threading */
            if (walk) {
                if (walk->n < walk->cap && walk->res) {
                    dbtk_walk_res_t* w = &walk->res[walk->n];
                    w->pair = (uint32_t)pi; w->dst = (uint32_t)destLocus; w->ret1 = (int8_t)alned0; w->ret2 = (int8_t)alned1; w->pad[0] = w->pad[1] = 0;
                }
                ++walk->n;
            }
            destLocus = locus0; /* the pair record keeps the locus the pair was walked through */
            goto emit;
        }
        if (p->threading) { stage = DBTK_STAGE_LOCUS; goto emit; } /* AQ.cpp:2072-2090: `alned` stays false at HEAD */
        nFeas += 2; /* AQ.cpp:2092 */
        if (p->extract) { stage = DBTK_STAGE_EXTRACT; goto emit; } /* AQ.cpp:2094-2099 */
        if (p->bait) { /* AQ.cpp:2104-2126 */
            uint8_t q1[DBTK_MAX_READ_LEN], q2[DBTK_MAX_READ_LEN];
            const uint8_t *m1 = NULL, *m2 = NULL;
            if (qual) {
                orc_qstring2qmask(qual + off[2 * pi], (int)l1, (int)p->qth, (int)k, q1);
                orc_qstring2qmask(qual + off[2 * pi + 1], (int)l2, (int)p->qth, (int)k, q2);
                m1 = q1; m2 = q2;
            }
            bfilter(g, destLocus, caks1, (int)nk1, m1, &bf1);
            bfilter(g, destLocus, caks2, (int)nk2, m2, &bf2);
            if (bf1 || bf2) {
                nBait += (uint64_t)((bf1 & !rm1) + (bf2 & !rm2));
                rm1 = 1; rm2 = 1;
                destLocus = nloci;
            }
        }
        /* AQ.cpp:2138-2158 */
        if (okam || !rm1 || !rm2) {
            if (okam || !rm1) ncls += nk1;
            if (okam || !rm2) ncls += nk2;
            assign_trkmc(g, p, caks1, (int)nk1, destLocus0, &r1, &af1, &rm1, okam);
            assign_trkmc(g, p, caks2, (int)nk2, destLocus0, &r2, &af2, &rm2, okam);
        }
        if (rm1 && rm2) { destLocus = nloci; stage = (bf1 || bf2) ? DBTK_STAGE_BAIT : DBTK_STAGE_ASGN; }
        else {
            int n = 2 - rm1 - rm2;
            nmapread[destLocus] += (uint32_t)n;
            nAsgn += (uint64_t)n;
            kmc[destLocus] += (uint64_t)(int64_t)((r1.ei - r1.si) + (r2.ei - r2.si));
            if (!rm1) for (int i = 0; i < r1.nas; ++i) if (r1.as[i] == 2) { ++counts[r1.its[i]]; ++ninc; }
            if (!rm2) for (int i = 0; i < r2.nas; ++i) if (r2.as[i] == 2) { ++counts[r2.its[i]]; ++ninc; }
            if (p->bubbles) { /* countNovelEdges, AQ.cpp:1559-1567, for the kept mates (AQ.cpp:2161-2166) */
                for (int m = 0; m < 2; ++m) {
                    if (m ? rm2 : rm1) continue;
                    const kmr_t* r = m ? &r2 : &r1;
                    const uint64_t* es = m ? caes2 : caes1;
                    for (int i = r->si_; i < r->ei_ - 1; ++i) {
                        const uint64_t e = es[i];
                        if (e == NAN64) continue;
                        if (!tre_has(g, destLocus, e)) {
                            if (ev && *nev < evcap) { ev[*nev].pair = (uint32_t)pi; ev[*nev].mate = (uint32_t)m; ev[*nev].pos = (uint32_t)i; ev[*nev].locus = (uint32_t)destLocus; ev[*nev].edge = e; }
                            ++*nev;
                        }
                    }
                }
            }
            stage = DBTK_STAGE_COUNTED;
        }
    emit:
        if (rec) {
            r1.kf = kf1; r1.hf = hf1; r1.af = af1; r1.rm = rm1; r1.bf = bf1;
            r2.kf = kf2; r2.hf = hf2; r2.af = af2; r2.rm = rm2; r2.bf = bf2;
            rec->pair = (uint32_t)pi;
            rec->stage = stage;
            rec->dst = (uint32_t)destLocus;
            rec->dst0 = (uint32_t)destLocus0;
            rec->nm1 = nm1; rec->nm2 = nm2;
            fill_mate_rec(&rec->r1, &r1);
            fill_mate_rec(&rec->r2, &r2);
        }
    }
    C[DBTK_C_NREADS] += 2 * npairs;
    C[DBTK_C_NSHORT] += nShort;
    C[DBTK_C_SUBFILTERED] += nSub;
    C[DBTK_C_KMERFILTERED] += nKf;
    C[DBTK_C_LOCUSFILTERED] += nLocus;
    C[DBTK_C_QCFILTERED] += nQC;
    C[DBTK_C_BAITFILTERED] += nBait;
    C[DBTK_C_THREADING] += nThr;
    C[DBTK_C_FEASIBLE] += nFeas;
    C[DBTK_C_ASGN] += nAsgn;
    C[DBTK_C_NHASH0] += nhash0;
    C[DBTK_C_NHASH1] += nhash1;
    C[DBTK_C_ALGO_PROBES] += nprobe;
    C[DBTK_C_ALGO_VV] += nvvw;
    orc_vote_vv_words += nvote;
    C[DBTK_C_ALGO_CLS] += ncls;
    C[DBTK_C_ALGO_INC] += ninc;
    C[DBTK_C_SURVIVORS] += nsurv;
    C[DBTK_C_BASES] += nbases;
    free(H.hits1); free(H.hits2); free(H.touched);
    free(caks1); free(caks2); free(caes1); free(caes2); free(its1); free(its2);
    return 0;
}
