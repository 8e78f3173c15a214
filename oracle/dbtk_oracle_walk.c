/*
 * dbtk_oracle_walk.c — CPU oracle (plain C) for the graph-threading walk of the
 * v1.3 contract: isThreadFeasible and everything under it.
 *
 * TEST INFRASTRUCTURE ONLY — see dbtk_oracle.h.  #included by dbtk_oracle.c (one
 * translation unit).  Every routine restates the reference routine named beside
 * it ("AQ.cpp" = /root/reference/src/aQueryFasta_thread.cpp, "AQ.h" = .../aQueryFasta_thread.h);
 * the containers are the oracle's own (fixed-capacity arrays instead of std::vector,
 * sorted per-locus arrays instead of unordered_map), the arithmetic and the control
 * flow follow the reference statement by statement, including its quirks:
 *   - the unsigned wrap-arounds of `kmers[ki] - oldnt + nt0` on a NAN64 k-mer (AQ.cpp:930),
 *   - getNextNucs keeping a stale next-base set when the node is absent (AQ.cpp:547-557),
 *   - `++ncorrection` after a mid-read backward correction on top of the edits (AQ.cpp:1204),
 *   - nskip / ncorrection being unsigned and compared after they wrapped.
 * Where the reference reads a vector out of bounds the value never decides anything
 * (it is and-ed with a `good[]` flag that is false there): those reads are skipped.
 * Where it asserts (a successor named by an edge mask is missing from the graph,
 * AQ.cpp:528-532) the walk stops and reports DBTK_THREAD_F_MISSING_NODE.
 *
 * Parity pinning: against the reference itself through oracle/_ref/libdbtk_refharness.so
 * (ref_thread), tests/test_oracle_walk.py.
 */

#define WCAP DBTK_THREAD_CAP
#define MSC 5u /* min score for thread extension, AQ.cpp:1120 */

typedef struct { uint8_t t, r, g; } edit_t; /* AQ.cpp:46-51 */
typedef struct {                            /* cigar_t, AQ.cpp:53-68 */
    int ni;
    edit_t es[WCAP + 8];
    int nes;
    char tr[WCAP + 8];
    int ntr;
} cigar_t;
typedef struct { /* the state of one walk */
    const orc_rpgg_t* g;
    uint64_t locus;
    uint32_t k;
    uint64_t rmask;
    uint64_t maxncorrection;
    uint64_t kmers[WCAP + 8];
    int nkm;
    cigar_t cg;
    uint32_t flags;
} walk_t;

static const char W_ALPHA[4] = {'A', 'C', 'G', 'T'};
static inline uint8_t w_comp_char(uint8_t c) { /* baseComplement on a base letter, AQ.h:71-87 */
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 127; }
}

/* ---- containers ------------------------------------------------------------*/
static void km_insert(walk_t* w, int at, uint64_t v) {
    if (w->nkm >= (int)WCAP) { w->flags |= DBTK_THREAD_F_OVERFLOW; return; }
    memmove(&w->kmers[at + 1], &w->kmers[at], (size_t)(w->nkm - at) * 8);
    w->kmers[at] = v; ++w->nkm;
}
static void km_erase(walk_t* w, int from, int to) { /* [from, to) */
    memmove(&w->kmers[from], &w->kmers[to], (size_t)(w->nkm - to) * 8);
    w->nkm -= to - from;
}
static void tr_insert(cigar_t* c, int at, char v, walk_t* w) {
    if (c->ntr >= (int)WCAP) { w->flags |= DBTK_THREAD_F_OVERFLOW; return; }
    memmove(&c->tr[at + 1], &c->tr[at], (size_t)(c->ntr - at));
    c->tr[at] = v; ++c->ntr;
}
static void tr_erase(cigar_t* c, int from, int to) {
    memmove(&c->tr[from], &c->tr[to], (size_t)(c->ntr - to));
    c->ntr -= to - from;
}
static void es_insert(cigar_t* c, int at, edit_t e, walk_t* w) {
    if (c->nes >= (int)WCAP) { w->flags |= DBTK_THREAD_F_OVERFLOW; return; }
    memmove(&c->es[at + 1], &c->es[at], (size_t)(c->nes - at) * sizeof(edit_t));
    c->es[at] = e; ++c->nes;
}
static void es_erase(cigar_t* c, int at) {
    memmove(&c->es[at], &c->es[at + 1], (size_t)(c->nes - at - 1) * sizeof(edit_t));
    --c->nes;
}
/* cg.es[i].t where the reference may index one before the vector (AQ.cpp:735: the bytes there are the
 * allocator's size field, never an edit letter) */
static inline uint8_t es_t_at(const cigar_t* c, int i) { return (i < 0 || i >= c->nes) ? 0 : c->es[i].t; }

/* ---- graph access ------------------------------------------------------------*/
/* GraphType = unordered_map<node, uint8 mask> per locus (AQ.h:32); here: sorted (node, mask) arrays */
static inline int gr_find(const orc_rpgg_t* g, uint64_t locus, uint64_t node, uint8_t* mask) {
    uint64_t lo = g->gr_beg[locus], hi = g->gr_beg[locus + 1];
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (g->gr_ks[mid] < node) lo = mid + 1; else hi = mid;
    }
    if (lo < g->gr_beg[locus + 1] && g->gr_ks[lo] == node) { if (mask) *mask = g->gr_ms[lo]; return 1; }
    return 0;
}
static inline int w_is_tr(const walk_t* w, uint64_t kmer) { /* trKmers.count(toCaKmer(kmer, ksize)), AQ.h:180-183 */
    uint64_t rc = orc_nurc(kmer, w->k);
    return tr_find(w->g, w->locus, kmer < rc ? kmer : rc) >= 0;
}
/* getOutNodes, AQ.cpp:526-540.  nnts is OR-ed.  Returns 0 where the reference asserts. */
static int get_out_nodes(walk_t* w, uint64_t node, uint64_t* nnds, int* nn, int nnts[4]) {
    uint8_t bits;
    if (!gr_find(w->g, w->locus, node, &bits)) { w->flags |= DBTK_THREAD_F_MISSING_NODE; return 0; }
    uint64_t nnd = (node & w->rmask) << 2;
    for (uint64_t i = 0; i < 4; ++i) {
        if (bits % 2) nnds[(*nn)++] = nnd + i;
        nnts[i] |= bits % 2;
        bits >>= 1;
    }
    return 1;
}
/* getNextNucs, AQ.cpp:547-557: assignment, and only when the node exists */
static void get_next_nucs(walk_t* w, uint64_t node, int nnts[4]) {
    uint8_t bits;
    if (gr_find(w->g, w->locus, node, &bits)) {
        for (int i = 0; i < 4; ++i) { nnts[i] = bits % 2; bits >>= 1; }
    }
}

/* ---- thread_ext_t, AQ.cpp:596-863 ----------------------------------------------*/
typedef struct {
    int rv;
    uint64_t nem1[4], nem2[16], nemi[4], nemd[16], ned1[4], ned2[16], nei1, nei2;
    uint64_t mes, ms1, ms2, score, nrk, nm, nd, ni;
    int dt_km, dt_ki, dt_nti;
    edit_t edits[2];
    int nedits;
} txt_t;
static void txt_init(txt_t* t, uint64_t msc, uint64_t mes, int rv) { /* AQ.cpp:619-624 */
    memset(t, 0, sizeof(*t));
    t->ms1 = 1 * msc; t->ms2 = 2 * msc; t->mes = mes; t->rv = rv;
}
static inline void set1(txt_t* t, char a, char ga) { t->nedits = 1; t->edits[0] = (edit_t){(uint8_t)a, 0, (uint8_t)ga}; }
static inline void set2(txt_t* t, char a, char ga, char b, char gb) {
    t->nedits = 2; t->edits[0] = (edit_t){(uint8_t)a, 0, (uint8_t)ga}; t->edits[1] = (edit_t){(uint8_t)b, 0, (uint8_t)gb};
}
/* get_edit, AQ.cpp:627-647: priority mismatch > del > ins, 1 edit > 2 edits; strictly better scores replace */
static int get_edit(txt_t* t) {
    for (int i = 0; i < 4; ++i) if (t->nem1[i] > t->score && t->nem1[i] >= t->ms1) { t->score = t->nem1[i]; set1(t, 'X', W_ALPHA[i]); }
    for (int i = 0; i < 4; ++i) if (t->ned1[i] > t->score && t->ned1[i] >= t->ms1) { t->score = t->ned1[i]; set1(t, 'D', W_ALPHA[i]); }
    if (t->nei1 > t->score && t->nei1 >= t->ms1) { t->score = t->nei1; set1(t, 'I', 0); }
    if (t->mes > 1) {
        for (int i = 0; i < 4; ++i) {
            for (int j = 0; j < 4; ++j) {
                uint64_t sm2 = t->nem2[i * 4 + j], smd = t->nemd[i * 4 + j], sd2 = t->ned2[i * 4 + j];
                if (sm2 > t->score && sm2 >= t->ms2) { t->score = sm2; set2(t, 'X', W_ALPHA[i], 'X', W_ALPHA[j]); }
                if (smd > t->score && smd >= t->ms2) { t->score = smd; set2(t, 'X', W_ALPHA[i], 'D', W_ALPHA[j]); }
                if (sd2 > t->score && sd2 >= t->ms2) { t->score = sd2; set2(t, 'D', W_ALPHA[i], 'D', W_ALPHA[j]); }
            }
            if (t->nemi[i] > t->score && t->nemi[i] >= t->ms2) { t->score = t->nemi[i]; set2(t, 'X', W_ALPHA[i], 'I', 0); }
        }
        if (t->nei2 > t->score && t->nei2 >= t->ms2) { t->score = t->nei2; set2(t, 'I', 0, 'I', 0); }
    }
    return t->score > 0;
}

/* edit_kmers_forward, AQ.cpp:828-862.  kmers / cg of the walk; *ki advances to the last extended k-mer. */
static void edit_kmers_forward(walk_t* w, txt_t* t, uint64_t* pki, uint64_t* ncorrection) {
    cigar_t* cg = &w->cg;
    uint64_t ki = *pki;
    const uint32_t k = w->k;
    const int n0 = w->nkm - (int)ki;
    uint8_t good[WCAP + 8];
    uint64_t nts[WCAP + 8];
    for (int i = 0; i < n0; ++i) { good[i] = w->kmers[ki + i] != NAN64; nts[i] = w->kmers[ki + i] % 4; }
    for (int e = 0; e < t->nedits; ++e) {
        const edit_t ed = t->edits[e];
        if (ed.t == 'X')      { w->kmers[ki] = ((w->kmers[ki - 1] & w->rmask) << 2) + (uint64_t)base_code(ed.g); ++ki; ++t->nm; }
        else if (ed.t == 'D') { km_insert(w, (int)ki, 0); w->kmers[ki] = ((w->kmers[ki - 1] & w->rmask) << 2) + (uint64_t)base_code(ed.g); ++ki; ++t->nd; }
        else if (ed.t == 'I') { km_erase(w, (int)ki, (int)ki + 1); ++t->ni; }
    }
    t->dt_nti = (int)(t->nm + t->ni);
    t->dt_ki = (int)(t->nm + t->nd);
    t->dt_km = (int)t->nd - (int)t->ni;
    {
        uint64_t lim = (uint64_t)w->nkm < ki + k ? (uint64_t)w->nkm : ki + k;
        for (uint64_t i = ki; i < lim; ++i) {
            if (t->dt_nti >= n0 || !good[t->dt_nti]) break;
            w->kmers[i] = ((w->kmers[i - 1] & w->rmask) << 2) + nts[t->dt_nti++];
        }
    }
    if (t->dt_km) { /* cg.tr.resize(size + dt_km, '*') */
        int nn = cg->ntr + t->dt_km;
        if (nn > (int)WCAP) { w->flags |= DBTK_THREAD_F_OVERFLOW; nn = (int)WCAP; }
        for (int i = cg->ntr; i < nn; ++i) cg->tr[i] = '*';
        cg->ntr = nn;
    }
    for (uint64_t i = 0; i < t->nd; ++i) es_insert(cg, cg->ni + (int)k - 1 + (int)t->nm, (edit_t){'D', 0, '*'}, w);
    {
        const int ki_ = (int)ki - t->dt_ki;
        for (int i = 0; i < t->dt_ki + (int)t->score; ++i) cg->tr[ki_ + i] = w_is_tr(w, w->kmers[ki_ + i]) ? '=' : '.';
    }
    for (int i = 0; i < t->nedits; ++i, ++cg->ni) {
        edit_t* e0 = &cg->es[cg->ni + (int)k - 1];
        e0->t = t->edits[i].t;
        e0->g = t->edits[i].g;
    }
    for (uint64_t i = 0; i < t->score; ++i, ++cg->ni) cg->es[cg->ni + (int)k - 1].t = '=';
    --cg->ni;
    ki += t->score - 1; /* shift to the last edited kmer */
    *ncorrection += (uint64_t)t->nedits;
    *pki = ki;
}

/* edit_kmers_backward, AQ.cpp:649-825.  Edits found on the reverse strand are applied to the k-mers before
 * the anchor *pki; then adjacent edits inside one "edit tract" are merged or cancelled. */
static void edit_kmers_backward(walk_t* w, txt_t* t, uint64_t* pki, uint64_t* ncorrection, uint64_t* nskip) {
    cigar_t* cg = &w->cg;
    uint64_t ki = *pki;
    const uint32_t k = w->k;
    const int n0 = (int)ki;
    uint8_t good[WCAP + 8];
    uint64_t nts[WCAP + 8]; /* leading nucleotides in kmers */
    const uint64_t lmask = 3ULL << 2 * (k - 1);
    const uint64_t lbase = 1ULL << 2 * (k - 1);
    for (int i = 0; i < n0; ++i) { good[i] = w->kmers[i] != NAN64; nts[i] = w->kmers[i] & lmask; }
    for (int e = 0; e < t->nedits; ++e) {
        if (t->edits[e].t == 'X') ++t->nm;
        else if (t->edits[e].t == 'D') ++t->nd;
        else if (t->edits[e].t == 'I') ++t->ni;
    }
    t->dt_km = (int)t->nd - (int)t->ni;
    t->dt_nti = -(int)(t->nm + t->ni);
    cg->ni += (int)t->nd;
    if (t->dt_km > 0) {
        for (int i = 0; i < t->dt_km; ++i) { km_insert(w, (int)ki, 0); tr_insert(cg, (int)ki, '*', w); }
    } else if (t->dt_km < 0) {
        km_erase(w, (int)ki + t->dt_km, (int)ki);
        tr_erase(cg, (int)ki + t->dt_km, (int)ki);
    }
    ki += (uint64_t)(int64_t)t->dt_km;
    /* corrected kmers */
    int ki_ = (int)ki;
    for (int e = 0; e < t->nedits; ++e) {
        const edit_t ed = t->edits[e];
        if (ed.t == 'X' || ed.t == 'D') {
            w->kmers[ki_ - 1] = (w->kmers[ki_] >> 2) + (uint64_t)(3 - base_code(ed.g)) * lbase;
            --ki_;
        }
    }
    /* extended kmers */
    {
        int lo = ki_ - (int)k; if (lo < 0) lo = 0;
        for (int i = ki_; i > lo; --i) {
            if (!good[i - 1]) break;
            w->kmers[i - 1] = (w->kmers[i] >> 2) + nts[i - 1];
        }
    }
    {
        const int lb = (int)ki - (int)t->nm - (int)t->nd - (int)t->score;
        for (int i = (int)ki - 1; i >= lb; --i) {
            if (cg->tr[i] == '*') ++t->nrk;
            cg->tr[i] = w_is_tr(w, w->kmers[i]) ? '=' : '.';
        }
    }
    t->nrk -= (t->nm + t->nd);
    *nskip -= t->nrk;
    *ncorrection += (uint64_t)t->nedits;
    /* CIGAR */
    {
        int cni = 0; /* cumulative # of ins */
        const int nti_ = (int)ki - t->dt_km;
        for (int i = 0; i < nti_ + cni; ++i) if (cg->es[i].t == 'I') ++cni;
        int nti = nti_ + cni - 1; /* convert cg.tr index (ki) to cg.es index (nti) */
        int e0, e1;               /* start, end of edit_tract */
        /* CIGAR of edits */
        for (int i = 0; i < t->nedits; ++i, --nti) {
            const edit_t ed1 = t->edits[i];
            if (ed1.t == 'D') {
                ++nti;
                es_insert(cg, nti, (edit_t){'D', 0, '*'}, w);
            }
            edit_t* ed0 = &cg->es[nti];
            if (ed0->t == 'D') {
                if (ed1.t == 'I') { es_erase(cg, nti); --cg->ni; } /* delete edit immediately */
                else              { ed0->g = w_comp_char(ed1.g); }
            } else {
                while (cg->es[nti].t == 'I') --nti;
                ed0 = &cg->es[nti];
                ed0->t = ed1.t;
                ed0->g = ed1.g ? w_comp_char(ed1.g) : 0;
            }
        }
        e0 = nti + 1;
        e1 = e0;
        /* CIGAR of extended alignment */
        for (uint64_t i = 0; i < t->score; ++i, --nti) {
            edit_t* e = &cg->es[nti];
            if (e->t == '=') { }
            else if (e->t == '*') e->t = '=';
            else break;
        }
        { /* find edit_tract */
            uint8_t c;
            c = es_t_at(cg, e1);
            while (c == 'X' || c == 'D' || c == 'I') { ++e1; c = es_t_at(cg, e1); }
            c = es_t_at(cg, e0 - 1);
            while (c == 'X' || c == 'D' || c == 'I') { --e0; c = es_t_at(cg, e0 - 1); }
        }
        /* merge edits if possible */
        uint8_t rnts[WCAP], gnts[WCAP];
        int nets = 0, nr = 0, ng = 0; /* ets.size(), rnts.size(), gnts.size() */
        for (int i = e0; i < e1; ++i) {
            const edit_t e = cg->es[i];
            ++nets;
            if (e.r) rnts[nr++] = e.r;
            if (e.g) gnts[ng++] = e.g;
        }
        if (nr == ng) {
            int no_edit = 1;
            for (int i = 0; i < nr; ++i) if (rnts[i] != gnts[i]) { no_edit = 0; break; }
            if (no_edit) { /* edits canceled out */
                int dt_es = 0;
                for (int i = e0; i < e1; ++i) {
                    const uint8_t c = cg->es[i + dt_es].t;
                    if (c == 'D') { es_erase(cg, i + dt_es); --dt_es; }
                    else { edit_t* e = &cg->es[i + dt_es]; e->t = '='; e->g = 0; }
                }
                cg->ni += dt_es;
                *ncorrection -= (uint64_t)(e1 - e0);
                *nskip -= (uint64_t)(e1 - e0);
            } else {
                if (nets != nr) { /* D+I (same position) -> X: the tract shrinks */
                    int dt_es = 0;
                    const int dt_es_ = nr - nets;
                    int i = e0, j = 0, kk = 0;
                    for (; i < e1; ++i) {
                        const uint8_t c = cg->es[i + dt_es].t;
                        if (c == 'D' && dt_es != dt_es_) { es_erase(cg, i + dt_es); --dt_es; }
                        else {
                            edit_t* e = &cg->es[i + dt_es];
                            if (rnts[kk] == gnts[kk]) { e->t = '='; e->g = 0; }
                            else { e->t = 'X'; e->g = gnts[j]; }
                            ++j; ++kk;
                        }
                    }
                    /* assert(dt_es == dt_es_) */
                    cg->ni += dt_es;
                    *ncorrection += (uint64_t)(int64_t)dt_es;
                    *nskip += (uint64_t)(int64_t)dt_es;
                } else { /* match/mismatch only */
                    for (int i = 0; i < nr; ++i) {
                        if (rnts[i] == gnts[i]) { /* edit reverted */
                            edit_t* e = &cg->es[e0 + i];
                            e->t = '='; e->g = 0;
                            --*ncorrection; --*nskip;
                        }
                    }
                }
            }
        } else { /* rnts.size() != gnts.size() */
            for (int i = 0; i < nets; ++i) {
                edit_t* e = &cg->es[e0 + i];
                if (e->r == e->g) { e->t = '='; e->g = 0; --*ncorrection; --*nskip; }
            }
        }
    }
    *pki = ki;
}

/* find_anchor, AQ.cpp:878-888: the first k-mer at/after *ki that is a node; every one passed is a skip */
static int find_anchor(walk_t* w, uint64_t* nskip, uint64_t* ki, uint64_t* node) {
    cigar_t* cg = &w->cg;
    while (!gr_find(w->g, w->locus, w->kmers[*ki], NULL)) {
        ++*nskip;
        ++cg->ni;
        if (++*ki >= (uint64_t)w->nkm) return 0;
    }
    *node = w->kmers[*ki];
    cg->tr[*ki] = w_is_tr(w, *node) ? '=' : '.';
    for (int i = cg->ni; i < cg->ni + (int)w->k; ++i) if (cg->es[i].t == '*') cg->es[i].t = '=';
    return 1;
}

/* errorCorrection_forward, AQ.cpp:898-1089, on an arbitrary k-mer array (the walk's own, or the
 * reverse-complemented prefix built by errorCorrection_backward).  Returns skip (1 = no edit found),
 * -1 where the reference asserts. */
static int ec_forward(walk_t* w, const uint64_t* nnds, int nn, const uint64_t* kmers, uint64_t nkmers, uint64_t ki,
                      const int nts0[4], txt_t* txt, uint64_t mes) {
    const uint32_t k = w->k;
    const uint64_t rmask = w->rmask;
    int nts1[4] = {0}, nts2[4] = {0};
    uint8_t mat[64] = {0}; /* graph_triplet_t, AQ.cpp:865-875 */
    const uint64_t oldnt = kmers[ki] % 4;
    for (int a = 0; a < nn; ++a) {
        const uint64_t node_i = nnds[a], nt0 = node_i % 4;
        uint64_t n1[4]; int nn1 = 0;
        if (!get_out_nodes(w, node_i, n1, &nn1, nts1)) return -1;
        for (int b = 0; b < nn1; ++b) {
            const uint64_t node_ip1 = n1[b], nt1 = node_ip1 % 4;
            uint64_t n2[4]; int nn2 = 0;
            if (!get_out_nodes(w, node_ip1, n2, &nn2, nts2)) return -1;
            for (int c = 0; c < nn2; ++c) mat[nt0 * 16 + nt1 * 4 + n2[c] % 4] = 1;
        }
    }
#define NNTS_1(i, out) do { for (int j_ = 0; j_ < 4; ++j_) for (int k_ = 0; k_ < 4; ++k_) (out)[j_] |= mat[(i) * 16 + j_ * 4 + k_]; } while (0)
#define NNTS_2(i, j, out) do { for (int k_ = 0; k_ < 4; ++k_) (out)[k_] |= mat[(i) * 16 + (j) * 4 + k_]; } while (0)
#define MINU(a, b) ((a) < (b) ? (a) : (b))
    uint8_t good[40] = {0};
    for (uint64_t i = 0; i < MINU((uint64_t)k + 2, nkmers - ki); ++i) good[i] = kmers[ki + i] != NAN64;
    /* One mismatch: match at ki+1 position */
    if (good[1] && nts1[kmers[ki + 1] % 4]) {
        for (uint64_t nt0 = 0; nt0 < 4; ++nt0) {
            if (!nts0[nt0]) continue;
            uint64_t crkmer = kmers[ki] - oldnt + nt0; /* corrected read kmer */
            int nnts[4] = {0};
            NNTS_1(nt0, nnts);
            for (uint64_t j = 1; j < MINU((uint64_t)k + 1, nkmers - ki); ++j) {
                if (!good[j]) break;
                crkmer = ((crkmer & rmask) << 2) + kmers[ki + j] % 4;
                if (nnts[crkmer % 4]) { ++txt->nem1[nt0]; get_next_nucs(w, crkmer, nnts); } else break;
            }
        }
    }
    /* Two mismatches: match at ki+2 position */
    else if (good[2] && mes >= 2 && nts2[kmers[ki + 2] % 4]) {
        for (uint64_t nt0 = 0; nt0 < 4; ++nt0) {
            if (!nts0[nt0]) continue;
            const uint64_t crkmer0 = kmers[ki] - oldnt + nt0;
            int nnt0[4] = {0};
            NNTS_1(nt0, nnt0);
            for (uint64_t nt1 = 0; nt1 < 4; ++nt1) {
                if (!nnt0[nt1]) continue;
                uint64_t crkmer1 = ((crkmer0 & rmask) << 2) + nt1;
                int nnt1[4] = {0};
                NNTS_2(nt0, nt1, nnt1);
                for (uint64_t j = 2; j < MINU((uint64_t)k + 2, nkmers - ki); ++j) {
                    if (!good[j]) break;
                    crkmer1 = ((crkmer1 & rmask) << 2) + kmers[ki + j] % 4;
                    if (nnt1[crkmer1 % 4]) { ++txt->nem2[nt0 * 4 + nt1]; get_next_nucs(w, crkmer1, nnt1); } else break;
                }
            }
        }
    }
    /* 1 substitution + 1 insertion */
    if (good[2] && mes >= 2 && nts1[kmers[ki + 2] % 4]) {
        for (uint64_t nt0 = 0; nt0 < 4; ++nt0) {
            if (!nts0[nt0]) continue;
            uint64_t crkmer = kmers[ki] - oldnt + nt0;
            int nnt0[4] = {0};
            NNTS_1(nt0, nnt0);
            for (uint64_t j = 2; j < MINU((uint64_t)k + 2, nkmers - ki); ++j) {
                if (!good[j]) break;
                crkmer = ((crkmer & rmask) << 2) + kmers[ki + j] % 4;
                if (nnt0[crkmer % 4]) { ++txt->nemi[nt0]; get_next_nucs(w, crkmer, nnt0); } else break;
            }
        }
    }
    /* 1 substitution + 1 deletion */
    if (good[1] && mes >= 2 && nts2[kmers[ki + 1] % 4]) {
        for (uint64_t nt0 = 0; nt0 < 4; ++nt0) {
            if (!nts0[nt0]) continue;
            const uint64_t crkmer0 = kmers[ki] - oldnt + nt0;
            int nnt0[4] = {0};
            NNTS_1(nt0, nnt0);
            for (uint64_t nt1 = 0; nt1 < 4; ++nt1) {
                if (!nnt0[nt1]) continue;
                uint64_t crkmer1 = ((crkmer0 & rmask) << 2) + nt1;
                int nnt1[4] = {0};
                NNTS_2(nt0, nt1, nnt1);
                for (uint64_t j = 1; j < MINU((uint64_t)k + 1, nkmers - ki); ++j) {
                    if (!good[j]) break;
                    crkmer1 = ((crkmer1 & rmask) << 2) + kmers[ki + j] % 4;
                    if (nnt1[crkmer1 % 4]) { ++txt->nemd[nt0 * 4 + nt1]; get_next_nucs(w, crkmer1, nnt1); } else break;
                }
            }
        }
    }
    /* 1 insertion */
    if (good[1] && nts0[kmers[ki + 1] % 4]) {
        uint64_t crkmer = kmers[ki - 1];
        int nnt0[4] = {nts0[0], nts0[1], nts0[2], nts0[3]};
        for (uint64_t j = 1; j < MINU((uint64_t)k + 1, nkmers - ki); ++j) {
            if (!good[j]) break;
            crkmer = ((crkmer & rmask) << 2) + kmers[ki + j] % 4;
            if (nnt0[crkmer % 4]) { ++txt->nei1; get_next_nucs(w, crkmer, nnt0); } else break;
        }
    }
    /* 1 deletion */
    if (good[0] && nts1[kmers[ki + 0] % 4]) {
        for (uint64_t nt0 = 0; nt0 < 4; ++nt0) {
            if (!nts0[nt0]) continue;
            uint64_t crkmer = kmers[ki] - oldnt + nt0;
            int nnt0[4] = {0};
            NNTS_1(nt0, nnt0);
            for (uint64_t j = 0; j < MINU((uint64_t)k, nkmers - ki); ++j) {
                if (!good[j]) break;
                crkmer = ((crkmer & rmask) << 2) + kmers[ki + j] % 4;
                if (nnt0[crkmer % 4]) { ++txt->ned1[nt0]; get_next_nucs(w, crkmer, nnt0); } else break;
            }
        }
    }
    /* 2 insertions */
    if (good[2] && mes >= 2 && nts0[kmers[ki + 2] % 4]) {
        uint64_t crkmer = kmers[ki - 1];
        int nnt0[4] = {nts0[0], nts0[1], nts0[2], nts0[3]};
        for (uint64_t j = 2; j < MINU((uint64_t)k + 2, nkmers - ki); ++j) {
            if (!good[j]) break;
            crkmer = ((crkmer & rmask) << 2) + kmers[ki + j] % 4;
            if (nnt0[crkmer % 4]) { ++txt->nei2; get_next_nucs(w, crkmer, nnt0); } else break;
        }
    }
    /* 2 deletions */
    if (good[0] && mes >= 2 && nts2[kmers[ki + 0] % 4]) {
        for (uint64_t nt0 = 0; nt0 < 4; ++nt0) {
            if (!nts0[nt0]) continue;
            const uint64_t crkmer0 = kmers[ki] - oldnt + nt0;
            int nnt0[4] = {0};
            NNTS_1(nt0, nnt0);
            for (uint64_t nt1 = 0; nt1 < 4; ++nt1) {
                if (!nnt0[nt1]) continue;
                uint64_t crkmer1 = ((crkmer0 & rmask) << 2) + nt1;
                int nnt1[4] = {0};
                NNTS_2(nt0, nt1, nnt1);
                for (uint64_t j = 0; j < MINU((uint64_t)k, nkmers - ki); ++j) {
                    if (!good[j]) break;
                    crkmer1 = ((crkmer1 & rmask) << 2) + kmers[ki + j] % 4;
                    if (nnt1[crkmer1 % 4]) { ++txt->ned2[nt0 * 4 + nt1]; get_next_nucs(w, crkmer1, nnt1); } else break;
                }
            }
        }
    }
#undef NNTS_1
#undef NNTS_2
    return !get_edit(txt); /* longer edits are treated with path-skipping and re-anchoring */
}

/* errorCorrection_backward, AQ.cpp:1091-1106: reverse-complement the prefix before the anchor and correct forward there */
static int ec_backward(walk_t* w, uint64_t node, uint64_t ki, txt_t* txt, uint64_t mes) {
    int nts0_rc[4] = {0};
    uint64_t nnds_rc[4]; int nn = 0;
    uint64_t kmers_rc[WCAP + 8];
    const uint64_t node_rc = orc_nurc(node, w->k);
    if (!get_out_nodes(w, node_rc, nnds_rc, &nn, nts0_rc)) return -1;
    kmers_rc[0] = node_rc; /* kmers_rc[1] is the first kmer that requires correction */
    int64_t j; uint64_t kk;
    for (j = (int64_t)ki - 1, kk = 1; j >= 0; --j, ++kk) kmers_rc[kk] = w->kmers[j] != NAN64 ? orc_nurc(w->kmers[j], w->k) : NAN64;
    return ec_forward(w, nnds_rc, nn, kmers_rc, ki + 1, 1, nts0_rc, txt, mes);
}

/* annot_gap, AQ.cpp:1108-1111 */
static void annot_gap(walk_t* w, uint64_t gap, uint64_t ki, uint64_t* nskip) {
    for (uint64_t i = 0; i < gap; ++i) w->cg.tr[--ki] = '*';
    *nskip -= gap;
}

/* read2kmers(..., canonical = false, keepN = true), AQ.h:246-271.  Returns kmers.size(): 0 or rlen-k+1. */
uint64_t orc_read2kmers_nonca(const uint8_t* read, uint64_t rlen, uint32_t k, uint64_t* kmers) {
    const uint64_t mask = (1ULL << 2 * (k - 1)) - 1;
    uint64_t kmer = 0;
    uint64_t beg = next_kmer(&kmer, 0, read, rlen, k);
    if (beg == rlen) return 0;
    for (uint64_t i = 0; i < rlen - k + 1; ++i) kmers[i] = NAN64;
    for (uint64_t i = beg; i < rlen - k + 1; ++i) {
        kmers[i] = kmer;
        int c = (i + k < rlen) ? base_code(read[i + k]) : -1;
        if (c < 0) {
            uint64_t nbeg = next_kmer(&kmer, i + k + 1, read, rlen, k);
            if (nbeg == rlen) return rlen - k + 1;
            i = nbeg - 1;
        } else {
            kmer = ((kmer & mask) << 2) + (uint64_t)c;
        }
    }
    return rlen - k + 1;
}

#define W_RET(v) do { ret = (v); goto done; } while (0)
/* isThreadFeasible, AQ.cpp:1114-1260, preceded by cigar_t::init (AQ.cpp:62-67) as at the call site (AQ.cpp:2073-2076).
 * 0: not feasible, 1: feasible w/o correction, 2: feasible w/ correction; -1: the reference would have asserted. */
int orc_thread(const orc_rpgg_t* g, uint64_t locus, const uint8_t* seq, uint64_t len, uint32_t k, uint32_t thread_cth, int correction,
               uint32_t maxncorrection, dbtk_thread_rec_t* out, uint64_t* noncakmers) {
    walk_t* w = (walk_t*)calloc(1, sizeof(walk_t));
    cigar_t* cg = &w->cg;
    int ret = 0;
    w->g = g; w->locus = locus; w->k = k; w->rmask = (1ULL << 2 * (k - 1)) - 1; w->maxncorrection = maxncorrection;
    /* cg.init(seq) */
    cg->nes = (int)len;
    for (int i = 0; i < (int)len; ++i) { cg->es[i].t = '*'; cg->es[i].r = seq[i]; cg->es[i].g = 0; }
    cg->ntr = (int)len - (int)k + 1;
    for (int i = 0; i < cg->ntr; ++i) cg->tr[i] = '*';

    w->nkm = (int)orc_read2kmers_nonca(seq, len, k, w->kmers);
    if (noncakmers) memcpy(noncakmers, w->kmers, (size_t)w->nkm * 8);
    if (w->nkm == 0) { w->flags |= DBTK_THREAD_F_OVERFLOW; W_RET(-1); } /* the reference indexes kmers[0] of an empty vector */

    const uint64_t maxnskip = ((uint64_t)w->nkm >= thread_cth ? (uint64_t)w->nkm - thread_cth : 0);
    uint64_t ki = 0, nskip = 0, ncorrection = 0;
    uint64_t node = w->kmers[0];
    const uint64_t nkmers = (uint64_t)w->nkm; /* NOT updated when the vector grows or shrinks (AQ.cpp:1126, 1182) */
    uint64_t mes;

    if (!find_anchor(w, &nskip, &ki, &node)) W_RET(0);
    if (ki > 0 && correction && ncorrection < maxncorrection) { /* leading unaligned kmers: backward alignment first */
        if (ki >= MSC + 1) {
            mes = (ki >= 2 * MSC + 2) ? 2 : 1;
            txt_t txtr; txt_init(&txtr, MSC, mes, 1);
            int skip = ec_backward(w, node, ki, &txtr, mes);
            if (skip < 0) W_RET(-1);
            if (!skip) edit_kmers_backward(w, &txtr, &ki, &ncorrection, &nskip);
        }
    }

    for (ki = ki + 1, cg->ni = cg->ni + 1; ki < (uint64_t)w->nkm; ++ki, ++cg->ni) {
        if (w->flags & DBTK_THREAD_F_OVERFLOW) W_RET(-1);
        if (w->kmers[ki] == NAN64) { /* "N" in read */
            cg->tr[ki] = '*';
            cg->es[cg->ni + (int)k - 1].t = '*';
            ++nskip;
            if (nskip > maxnskip) W_RET(0);
            continue;
        }
        if (w->kmers[ki] == w->kmers[ki - 1]) { /* skip homopolymer run */
            cg->tr[ki] = '*';
            cg->es[cg->ni + (int)k - 1].t = '*';
            ++nskip;
            if (nskip > maxnskip) W_RET(0);
            continue;
        }
        if (w->kmers[ki - 1] == NAN64) { /* triggered after passing 'N' */
            if (!find_anchor(w, &nskip, &ki, &node)) break;
            if (nskip > maxnskip) W_RET(0);
            continue;
        }

        int skip = 1;
        int nts0[4] = {0};
        uint64_t nnds[4]; int nn = 0;
        if (!get_out_nodes(w, node, nnds, &nn, nts0)) W_RET(-1);
        for (int a = 0; a < nn; ++a) {
            if (w->kmers[ki] == nnds[a]) { /* matching node found */
                node = nnds[a];
                skip = 0;
                cg->tr[ki] = w_is_tr(w, w->kmers[ki]) ? '=' : '.';
                cg->es[cg->ni + (int)k - 1].t = '=';
                break;
            }
        }
        if (!skip) continue;
        /* read kmer has no matching node in the graph, try error correction */
        if (ki + MSC >= nkmers) { /* not enough info */
            nskip += (nkmers - ki);
            W_RET(nskip <= maxnskip ? (ncorrection ? 2 : 1) : 0);
        }
        if (correction && ncorrection < maxncorrection) {
            mes = ((uint64_t)w->nkm - ki >= 2 * MSC + 2) ? 2 : 1;
            txt_t txtf; txt_init(&txtf, MSC, mes, 0);
            skip = ec_forward(w, nnds, nn, w->kmers, (uint64_t)w->nkm, ki, nts0, &txtf, mes);
            if (skip < 0) W_RET(-1);
            if (!skip) { /* passed forward correction */
                nskip += (uint64_t)txtf.nedits;
                if (nskip > maxnskip) W_RET(0);
                edit_kmers_forward(w, &txtf, &ki, &ncorrection);
                node = w->kmers[ki];
            } else {
                uint64_t gap;
                if (!find_anchor(w, &nskip, &ki, &node)) break;
                mes = 2; /* always have enough info to make 2 edits */
                txt_t txtr; txt_init(&txtr, MSC, mes, 1);
                skip = ec_backward(w, node, ki, &txtr, mes);
                if (skip < 0) W_RET(-1);
                if (!skip) { /* passed reverse correction */
                    edit_kmers_backward(w, &txtr, &ki, &ncorrection, &nskip);
                    ++ncorrection;
                    gap = MINU((uint64_t)k, ki - txtr.nm - txtr.nd) - txtr.score;
                    uint64_t ki0 = ki, ki1 = ki;
                    while (!skip && gap) { /* forward and backward threads not fully patched */
                        ki0 = ki1;
                        ki1 = ki0 - txtr.nm - txtr.nd - txtr.score;
                        mes = (ki1 >= 2 * MSC + 2) ? 2 : 1;
                        if (ki1 < MSC + 1) break;
                        txt_init(&txtr, MSC, mes, 1);
                        const uint64_t node_ = w->kmers[ki1];
                        /* assert(g.count(node_)) */
                        if (!gr_find(g, locus, node_, NULL)) { w->flags |= DBTK_THREAD_F_MISSING_NODE; W_RET(-1); }
                        skip = ec_backward(w, node_, ki1, &txtr, mes);
                        if (skip < 0) W_RET(-1);
                        if (!skip) {
                            edit_kmers_backward(w, &txtr, &ki1, &ncorrection, &nskip);
                            ki += txtr.nd - txtr.ni;
                            gap = MINU((uint64_t)k, ki1 - txtr.nm - txtr.nd) - txtr.score;
                        }
                    }
                    if (gap) annot_gap(w, gap, ki1, &nskip);
                    if (nskip > maxnskip) W_RET(0);
                }
                if (skip) { /* either initial or iterative backward correction failed */
                    if (!find_anchor(w, &nskip, &ki, &node)) break;
                    if (nskip > maxnskip) W_RET(0);
                    continue;
                }
            }
        } else {
            if (!find_anchor(w, &nskip, &ki, &node)) break;
            if (nskip > maxnskip) W_RET(0);
            continue;
        }
    }
    ret = (nskip <= maxnskip && ncorrection <= maxncorrection ? (ncorrection ? 2 : 1) : 0);
done:
    if (w->flags & DBTK_THREAD_F_OVERFLOW) ret = -1;
    if (out) {
        memset(out, 0, sizeof(*out));
        out->ret = ret; out->ni = cg->ni; out->nkm = (uint32_t)w->nkm; out->nes = (uint32_t)cg->nes; out->ntr = (uint32_t)cg->ntr;
        out->flags = w->flags;
        for (int i = 0; i < w->nkm && i < (int)WCAP; ++i) out->kmers[i] = w->kmers[i];
        for (int i = 0; i < cg->nes && i < (int)WCAP; ++i) { out->es_t[i] = cg->es[i].t; out->es_r[i] = cg->es[i].r; out->es_g[i] = cg->es[i].g; }
        for (int i = 0; i < cg->ntr && i < (int)WCAP; ++i) out->tr[i] = (uint8_t)cg->tr[i];
    }
    free(w);
    return ret;
}
#undef W_RET
#undef MINU

/* ---- emit: writeCigar / writeAnnot, AQ.cpp:1683-1740 ------------------------------------------------*/
/* Both print to std::cout in the reference; here they append to buf (capacity cap) and return the new length. */
static size_t w_put(char* buf, size_t cap, size_t n, const char* s) {
    for (; *s; ++s) { if (n + 1 < cap) buf[n] = *s; ++n; }
    if (cap) buf[n < cap ? n : cap - 1] = 0;
    return n;
}
static size_t w_putc(char* buf, size_t cap, size_t n, char c) { char s[2] = {c, 0}; return w_put(buf, cap, n, s); }
static size_t w_puti(char* buf, size_t cap, size_t n, int v) { char s[16]; snprintf(s, sizeof s, "%d", v); return w_put(buf, cap, n, s); }

size_t orc_write_annot(const dbtk_thread_rec_t* r, char* buf, size_t cap) { /* writeAnnot, AQ.cpp:1683-1700 */
    size_t n = 0;
    const uint8_t* tr = r->tr;
    const int sz = (int)r->ntr;
    if (cap) buf[0] = 0;
    if (!sz) return w_putc(buf, cap, n, '*');
    int ct = 1;
    uint8_t c0 = tr[0];
    for (int i = 1; i < sz; ++i) {
        if (c0 == '=' || c0 == '.' || c0 == '*') {
            while (tr[i] == c0) { ++ct; ++i; if (i == sz) break; }
            n = w_puti(buf, cap, n, ct); n = w_putc(buf, cap, n, (char)c0);
        } else n = w_putc(buf, cap, n, (char)c0);
        if (i == sz) return n;
        ct = 1;
        c0 = tr[i];
    }
    n = w_puti(buf, cap, n, ct); n = w_putc(buf, cap, n, (char)c0);
    return n;
}
size_t orc_write_cigar(const dbtk_thread_rec_t* r, char* buf, size_t cap) { /* writeCigar, AQ.cpp:1702-1740 */
    size_t n = 0;
    const int sz = (int)r->nes;
    if (cap) buf[0] = 0;
    if (!sz) return w_putc(buf, cap, n, '*');
    int ct = 1;
    uint8_t t0 = r->es_t[0], g0 = r->es_g[0], t1, g1;
    for (int i = 1; i < sz; ++i) {
        t1 = r->es_t[i]; g1 = r->es_g[i];
        if (t0 == '=' || t0 == '.' || t0 == '*') {
            while (t1 == t0) {
                ++ct; ++i;
                if (i == sz) break;
                t1 = r->es_t[i]; g1 = r->es_g[i];
            }
            n = w_puti(buf, cap, n, ct); n = w_putc(buf, cap, n, (char)t0);
        } else if (t0 == 'X') {
            n = w_putc(buf, cap, n, 'X'); n = w_putc(buf, cap, n, (char)g0);
        } else if (t0 == 'D') {
            if (t1 == 'I') { n = w_putc(buf, cap, n, 'X'); n = w_putc(buf, cap, n, (char)g0); ++i; } /* ins + del = mismatch */
            else { n = w_putc(buf, cap, n, 'D'); n = w_putc(buf, cap, n, (char)g0); }
        } else if (t0 == 'I') {
            if (t1 == 'D') { n = w_putc(buf, cap, n, 'X'); n = w_putc(buf, cap, n, (char)g1); ++i; }
            else n = w_putc(buf, cap, n, 'I');
        } else n = w_putc(buf, cap, n, (char)t0);
        if (i == sz) return n;
        ct = 1;
        t0 = r->es_t[i]; g0 = r->es_g[i];
    }
    n = w_puti(buf, cap, n, ct); n = w_putc(buf, cap, n, (char)t0);
    return n;
}

/* One record of writeAlignments (AQ.cpp:1742-1759): `src dst title seq2 seq1 cigar2 annot2 cigar1 annot1\n` with src '.'
 * outside simulation mode; r1 / r2 = the thread records of seq1 (read 2p) / seq2 (read 2p+1). */
size_t orc_write_alignment(int64_t src, uint32_t dst, const char* title, const uint8_t* seq1, uint64_t l1, const uint8_t* seq2, uint64_t l2,
                           const dbtk_thread_rec_t* r1, const dbtk_thread_rec_t* r2, char* buf, size_t cap) {
    size_t n = 0;
    char tmp[4096];
    if (cap) buf[0] = 0;
    if (src < 0) n = w_putc(buf, cap, n, '.'); else n = w_puti(buf, cap, n, (int)src);
    n = w_putc(buf, cap, n, '\t');
    n = w_puti(buf, cap, n, (int)dst);
    n = w_putc(buf, cap, n, '\t');
    n = w_put(buf, cap, n, title);
    n = w_putc(buf, cap, n, '\t');
    for (uint64_t i = 0; i < l2; ++i) n = w_putc(buf, cap, n, (char)seq2[i]);
    n = w_putc(buf, cap, n, '\t');
    for (uint64_t i = 0; i < l1; ++i) n = w_putc(buf, cap, n, (char)seq1[i]);
    n = w_putc(buf, cap, n, '\t');
    orc_write_cigar(r2, tmp, sizeof tmp); n = w_put(buf, cap, n, tmp); n = w_putc(buf, cap, n, '\t');
    orc_write_annot(r2, tmp, sizeof tmp); n = w_put(buf, cap, n, tmp); n = w_putc(buf, cap, n, '\t');
    orc_write_cigar(r1, tmp, sizeof tmp); n = w_put(buf, cap, n, tmp); n = w_putc(buf, cap, n, '\t');
    orc_write_annot(r1, tmp, sizeof tmp); n = w_put(buf, cap, n, tmp); n = w_putc(buf, cap, n, '\n');
    return n;
}
