/*
 * ref_harness.cpp — function-level access to the REAL reference.
 *
 * TEST INFRASTRUCTURE ONLY.  This translation unit contains no reference
 * code: it #includes the reference's own source where it lies under
 * /root/reference (the include path is given by oracle/Makefile) with `main`
 * renamed, and exports thin extern "C" shims around the reference's functions
 * so the tests can pin the C oracle (dbtk_oracle.c) and the HIP path against
 * them one routine at a time.  Output: oracle/_ref/libdbtk_refharness.so
 * (git-ignored; it travels to the GPU box with the snapshot).
 *
 * ref_pair_*() drives the reference's functions in the order CountWords does
 * (src/aQueryFasta_thread.cpp:2035-2158); only that glue is restated here,
 * every computation is the reference's.
 */
/* The reference prints its records to std::cout (writeCigar, writeAnnot, writeAlignments ...).  The harness hands
 * them back as strings, so `cout` inside the reference's translation unit is a stream the harness owns. */
#include <iostream>
#include <sstream>
namespace std { static ostream dbtk_harness_cout(nullptr); }  /* (in std so that the reference's `using std::cout;` still parses) */
#define cout dbtk_harness_cout
#define main danbing_tk_reference_main
#include "aQueryFasta_thread.cpp"
#undef main
#undef cout

#include "../include/dbtk.h"

#include <setjmp.h>
#include <signal.h>

namespace {
/* The reference asserts (abort) on an unclean graph (AQ.cpp:528-532) and inside threadCheck; the harness turns
 * such an abort into a return code instead of losing the test process. */
sigjmp_buf g_abort_env;
volatile sig_atomic_t g_abort_armed = 0;
void on_abort(int) { if (g_abort_armed) { g_abort_armed = 0; siglongjmp(g_abort_env, 1); } }
struct AbortTrap {
    struct sigaction old;
    AbortTrap() { struct sigaction sa; memset(&sa, 0, sizeof(sa)); sa.sa_handler = on_abort; sa.sa_flags = SA_NODEFER; sigaction(SIGABRT, &sa, &old); }
    ~AbortTrap() { g_abort_armed = 0; sigaction(SIGABRT, &old, nullptr); }
};
/* std::cout captured while writeCigar / writeAnnot / writeAlignments run (they print, AQ.cpp:1683-1759) */
struct CoutCapture {
    std::ostringstream ss; std::streambuf* old;
    CoutCapture() { old = std::dbtk_harness_cout.rdbuf(ss.rdbuf()); std::dbtk_harness_cout.clear(); }
    ~CoutCapture() { std::dbtk_harness_cout.rdbuf(old); }
};
void fill_thread_rec(dbtk_thread_rec_t* o, int ret, cigar_t& cg, vector<uint64_t>& kmers) {
    memset(o, 0, sizeof(*o));
    o->ret = ret; o->ni = cg.ni;
    o->nkm = kmers.size(); o->nes = cg.es.size(); o->ntr = cg.tr.size();
    if (o->nkm > DBTK_THREAD_CAP || o->nes > DBTK_THREAD_CAP || o->ntr > DBTK_THREAD_CAP) { o->flags |= DBTK_THREAD_F_OVERFLOW; }
    for (size_t i = 0; i < kmers.size() && i < DBTK_THREAD_CAP; ++i) o->kmers[i] = kmers[i];
    for (size_t i = 0; i < cg.es.size() && i < DBTK_THREAD_CAP; ++i) { o->es_t[i] = cg.es[i].t; o->es_r[i] = cg.es[i].r; o->es_g[i] = cg.es[i].g; }
    for (size_t i = 0; i < cg.tr.size() && i < DBTK_THREAD_CAP; ++i) o->tr[i] = cg.tr[i];
}

struct RefDB {
    uint64_t nloci = 0;
    kmerIndex_uint32_umap kmerDBi;
    vector<uint32_t> kmerDBi_vv;
    kset_db_t flankDB, trEdgeDB;
    vector<kmer_aCount_umap> trKmerDB;
    vector<uint8_t> qcFilter;
    vector<uint32_t> hits1, hits2;
    /* key -> file-order index of PREF.tr.kmers, per locus */
    vector<unordered_map<uint64_t, uint64_t>> fileIndex;
    uint64_t ntr = 0;
    bait_fps_db_t baitDB;
    vector<GraphType> graphDB;   /* readGraphKmers(PREF.graph.kmers), AQ.h:550-575 */
};
struct BubEvent { uint32_t pair, mate, pos, locus; uint64_t edge; };  // == orc_bub_event_t

void fill_mate(dbtk_mate_rec_t& m, km_asgn_read_t& r) {
    memset(&m, 0, sizeof(m));
    m.si = r.si; m.ei = r.ei; m.si_ = r.si_; m.ei_ = r.ei_; m.nt = r.nt; m.bs = r.bs; m.ti = r.ti;
    m.kf = r.kf; m.hf = r.hf; m.bf = r.bf; m.qf = r.qf; m.af = r.af; m.rm = r.rm;
    m.nk = r.as.size();
    for (size_t i = 0; i < r.as.size(); ++i) m.as2[i >> 2] |= (uint8_t)((r.as[i] & 3) << (2 * (i & 3)));
}
}  // namespace

extern "C" {

void ref_set_params(uint64_t k, uint64_t nfilter, uint64_t nmfilter, uint64_t max_nt, uint64_t nm_tr) {
    ksize = k;
    rmask = (1ULL << 2 * (ksize - 1)) - 1;
    N_FILTER = nfilter;
    NM_FILTER = nmfilter;
    MAX_NT = max_nt;
    NM_TR = nm_tr;
}

uint64_t ref_nurc(uint64_t kmer, uint64_t k) { return getNuRC(kmer, k); }

uint64_t ref_read2kmers_edges(const char* read, uint64_t rlen, uint64_t k, uint64_t* kmers, uint64_t* edges) {
    string s(read, rlen);
    vector<size_t> ks, es;
    read2kmers_edges(ks, es, s, k);
    for (size_t i = 0; i < ks.size(); ++i) kmers[i] = ks[i];
    for (size_t i = 0; i < es.size(); ++i) edges[i] = es[i];
    return ks.size();
}

void ref_sort_index(const uint64_t* data, uint64_t n, uint64_t* idx) {
    vector<uint64_t> d(data, data + n), ind(n);
    getSortedIndex(d, ind);
    for (uint64_t i = 0; i < n; ++i) idx[i] = ind[i];
}

/* iteration order of kmer_aCount_umap after `db[key] = 0` in input order
 * (readKmersWithZeroCount, src/aQueryFasta_thread.h:469-480) */
void ref_umap_order(const uint64_t* keys, uint64_t n, uint64_t* order) {
    kmer_aCount_umap m;
    unordered_map<uint64_t, uint64_t> first;
    for (uint64_t i = 0; i < n; ++i) {
        m[keys[i]] = 0;
        if (!first.count(keys[i])) first[keys[i]] = i;
    }
    uint64_t j = 0;
    for (auto& p : m) order[j++] = first[p.first];
    for (; j < n; ++j) order[j] = (uint64_t)-1;
}

void* ref_db_load(const char* prefix, const char* qc_file) {
    RefDB* db = new RefDB;
    string pref(prefix);
    db->nloci = countLoci(pref + ".tr.kmers");
    db->trKmerDB = vector<kmer_aCount_umap>(db->nloci);
    readBinaryIndex(db->kmerDBi, db->kmerDBi_vv, pref);
    readBinaryKmerSetDB(db->flankDB, pref + ".fl");
    readBinaryKmerSetDB(db->trEdgeDB, pref + ".tre");
    readKmersWithZeroCount(db->trKmerDB, pref + ".tr.kmers");
    db->qcFilter.resize(db->nloci);
    if (qc_file) readQCFile(db->qcFilter, string(qc_file));
    db->hits1.assign(db->nloci + 1, 0);
    db->hits2.assign(db->nloci + 1, 0);
    db->fileIndex.resize(db->nloci);
    {
        ifstream f(pref + ".tr.kmers");
        string line;
        size_t idx = -1;
        uint64_t fi = 0;
        while (getline(f, line)) {
            if (line[0] == '>') { ++idx; }
            else {
                uint64_t km = stoul(line);
                if (!db->fileIndex[idx].count(km)) db->fileIndex[idx][km] = fi;
                ++fi;
            }
        }
        db->ntr = fi;
    }
    return db;
}
void ref_db_free(void* h) { delete (RefDB*)h; }
/* graph loader of the v1.3 contract: readGraphKmers (src/aQueryFasta_thread.h:550-575) on PREF.graph.kmers */
void ref_db_load_graph(void* h, const char* graph_kmers_file) {
    RefDB* db = (RefDB*)h;
    db->graphDB = vector<GraphType>(db->nloci);
    std::streambuf* old = std::cerr.rdbuf(nullptr);
    readGraphKmers(db->graphDB, string(graph_kmers_file));
    std::cerr.rdbuf(old);
    std::cerr.clear();
}
/* graphDB[locus] as readGraphKmers left it: (node, mask) in ascending node order (the map itself is unordered) */
uint64_t ref_graph_dump(void* h, uint32_t locus, uint64_t* ks, uint8_t* ms, uint64_t cap) {
    RefDB* db = (RefDB*)h;
    if (locus >= db->graphDB.size()) return 0;
    std::vector<std::pair<uint64_t, uint8_t>> v(db->graphDB[locus].begin(), db->graphDB[locus].end());
    std::sort(v.begin(), v.end());
    for (size_t i = 0; i < v.size() && i < cap; ++i) { ks[i] = v[i].first; ms[i] = v[i].second; }
    return v.size();
}
void ref_set_thread_params(uint64_t maxncorr, int verb) { maxncorrection = maxncorr; verbosity = verb; }

/* isThreadFeasible (src/aQueryFasta_thread.cpp:1114-1260) for one read against graphDB[locus] / trKmerDB[locus],
 * called the way the v1.3 call site does (AQ.cpp:2072-2075): cigar_t::init, then the walk; tc = also threadCheck
 * (AQ.cpp:1276-1342, -gcc) when the walk is feasible.  Returns the walk's code, or -1 if the reference aborted
 * (assert).  cigar/annot receive what writeCigar(cg.es) / writeAnnot(cg.tr) print (AQ.cpp:1683-1740), NUL-terminated;
 * *flagged = threadCheck wrote a "[!]" line; noncak[nk] = the uncorrected k-mers (read2kmers keepN). */
int ref_thread(void* h, uint32_t locus, const char* s, uint64_t len, uint32_t thread_cth, int correction, int tc,
               dbtk_thread_rec_t* out, uint64_t* noncak, char* cigar, char* annot, uint64_t strcap, int* flagged) {
    RefDB& db = *(RefDB*)h;
    AbortTrap trap;
    string seq(s, len);
    vector<uint64_t> noncakmers, akmers;
    cigar_t cg;
    log_t log;
    int ret = -1;
    if (flagged) *flagged = 0;
    std::streambuf* olderr = std::cerr.rdbuf(nullptr);
    g_abort_armed = 1;
    if (sigsetjmp(g_abort_env, 1) == 0) {
        cg.init(seq);
        ret = isThreadFeasible(db.graphDB[locus], seq, noncakmers, akmers, thread_cth, correction != 0, cg, db.trKmerDB[locus], log);
        if (out) fill_thread_rec(out, ret, cg, akmers);
        if (noncak) for (size_t i = 0; i < noncakmers.size(); ++i) noncak[i] = noncakmers[i];
        if (cigar) {
            CoutCapture cap;
            writeCigar(cg.es);
            snprintf(cigar, strcap, "%s", cap.ss.str().c_str());
        }
        if (annot) {
            CoutCapture cap;
            writeAnnot(cg.tr);
            snprintf(annot, strcap, "%s", cap.ss.str().c_str());
        }
        if (tc && ret) {
            threadCheck(db.graphDB[locus], seq, akmers, cg, log);
            if (flagged && log.m.str().find("[!]") != string::npos) *flagged = 1;
        }
        g_abort_armed = 0;
    } else {
        ret = -1;
        if (out) { out->ret = -1; }
    }
    std::cerr.rdbuf(olderr);
    std::cerr.clear();
    return ret;
}
/* readBinaryBaitDB reads PREF.bt.kmdb (src/aQueryFasta_thread.h:542-547) */
void ref_db_load_bait(void* h, const char* prefix) { readBinaryBaitDB(((RefDB*)h)->baitDB, string(prefix)); }
void ref_qstring2qmask(const char* qual, int nq, int qth_, int k, uint8_t* mask) {
    string q(qual, nq);
    vector<bool> m;
    qString2qMask(q, qth_, k, m);
    for (size_t i = 0; i < m.size(); ++i) mask[i] = m[i];
}
uint64_t ref_db_nloci(void* h) { return ((RefDB*)h)->nloci; }
uint64_t ref_db_ntr(void* h) { return ((RefDB*)h)->ntr; }

/* One pair through the reference's live path.  Accumulates like orc_align. */
void ref_pair(void* h, const char* s1, uint64_t l1, const char* s2, uint64_t l2, uint32_t Cth, int okam, int qc,
              uint64_t* counts_fileorder, uint64_t* kmc, uint32_t* nmapread, uint64_t* C, dbtk_pair_rec_t* rec,
              uint32_t pair_index, const char* q1 = nullptr, const char* q2 = nullptr, int bait = 0, int bubbles_on = 0,
              BubEvent* ev = nullptr, uint64_t evcap = 0, uint64_t* nev = nullptr) {
    RefDB& db = *(RefDB*)h;
    const uint64_t nloci = db.nloci;
    uint16_t Cthreshold = Cth;
    string seq1(s1, l1), seq2(s2, l2);
    vector<uint64_t> caks1, caks2, caes1, caes2;
    vector<kmerIndex_uint32_umap::iterator> its1, its2;
    vector<PE_KMC> dup;
    log_t log;
    int rm1 = 0, rm2 = 0, kf1 = 0, kf2 = 0, hf1 = 0, hf2 = 0, bf1 = 0, bf2 = 0, qf1 = 0, qf2 = 0, af1 = 0, af2 = 0;
    int qn1 = 0, qn2 = 0, qm1 = 0, qm2 = 0, nm1 = 0, nm2 = 0;
    uint64_t destLocus = nloci, destLocus0 = NAN32;
    uint64_t nhash0 = 0, nhash1 = 0;
    uint32_t stage = 0;
    km_asgn_t kam;
    C[DBTK_C_NREADS] += 2;

    read2kmers_edges(caks1, caes1, seq1, ksize);
    read2kmers_edges(caks2, caes2, seq2, ksize);
    do {
        if (not caks1.size() or not caks2.size()) { C[DBTK_C_NSHORT] += 1; stage = DBTK_STAGE_SHORT; break; }
        if (N_FILTER and NM_FILTER) {
            bool sub = subfilter(caks1, caks2, db.kmerDBi, nhash0);
            if (sub) { C[DBTK_C_SUBFILTERED] += 2; stage = DBTK_STAGE_SUBFILTER; break; }
        }
        kfilter(caks1, caks2, its1, its2, db.kmerDBi, Cthreshold, nhash1, kf1, kf2, rm1, rm2);
        C[DBTK_C_KMERFILTERED] += kf1 + kf2;
        if (rm1 and rm2) { stage = DBTK_STAGE_KFILTER; break; }
        destLocus = countHit(db.kmerDBi_vv, its1, its2, db.hits1, db.hits2, dup, nloci, Cthreshold, log, destLocus0, nm1, nm2,
                             hf1, hf2, rm1, rm2);
        C[DBTK_C_LOCUSFILTERED] += hf1 + hf2;
        if (destLocus == nloci) { stage = DBTK_STAGE_LOCUS; break; }
        if (qc and not db.qcFilter[destLocus]) { C[DBTK_C_QCFILTERED] += 2 - rm1 - rm2; stage = DBTK_STAGE_QC; break; }
        C[DBTK_C_THREADING] += 2;
        C[DBTK_C_FEASIBLE] += 2;
        if (bait) {  // AQ.cpp:2101-2126
            vector<bool> qkm1, qkm2;
            bt_tracker_t tkr;
            auto& baitdb = db.baitDB[destLocus];
            if (q1) {
                string qs1(q1, l1), qs2(q2, l2);
                qString2qMask(qs1, qth, ksize, qkm1);
                qString2qMask(qs2, qth, ksize, qkm2);
                bfilter_FPSv1(baitdb, caks1, qkm1, bf1, false, tkr, destLocus);
                bfilter_FPSv1(baitdb, caks2, qkm2, bf2, false, tkr, destLocus);
            } else {
                bfilter_FPSv1(baitdb, caks1, bf1, false, tkr, destLocus);
                bfilter_FPSv1(baitdb, caks2, bf2, false, tkr, destLocus);
            }
            if (bf1 or bf2) {
                C[DBTK_C_BAITFILTERED] += (bf1 & !rm1) + (bf2 & !rm2);
                rm1 = 1;
                rm2 = 1;
                destLocus = nloci;
            }
        }
        vector<kmer_aCount_umap::iterator> kits1, kits2;
        if (okam or not rm1 or not rm2) {
            kmer_aCount_umap& trKmers = db.trKmerDB[destLocus0];
            unordered_set<uint64_t>& flKmers = db.flankDB[destLocus0];
            assignTRkmc(caks1, trKmers, flKmers, kits1, kam.r1, af1, rm1, okam);
            assignTRkmc(caks2, trKmers, flKmers, kits2, kam.r2, af2, rm2, okam);
        }
        if (rm1 and rm2) { destLocus = nloci; stage = (bf1 or bf2) ? DBTK_STAGE_BAIT : DBTK_STAGE_ASGN; }
        else {
            int n = 2 - rm1 - rm2;
            nmapread[destLocus] += n;
            C[DBTK_C_ASGN] += n;
            kmc[destLocus] += (kam.r1.ei - kam.r1.si) + (kam.r2.ei - kam.r2.si);
            auto& as1 = kam.r1.as;
            auto& as2 = kam.r2.as;
            auto& fidx = db.fileIndex[destLocus0];
            if (not rm1) { for (int i = 0; i < (int)as1.size(); ++i) { if (as1[i] == 2) { ++counts_fileorder[fidx[kits1[i]->first]]; } } }
            if (not rm2) { for (int i = 0; i < (int)as2.size(); ++i) { if (as2[i] == 2) { ++counts_fileorder[fidx[kits2[i]->first]]; } } }
            if (bubbles_on) {  // AQ.cpp:2161-2166 with countNovelEdges' loop (AQ.cpp:1559-1567) spelled out to keep the order
                unordered_set<uint64_t>& tres = db.trEdgeDB[destLocus];
                for (int m = 0; m < 2; ++m) {
                    if (m ? rm2 : rm1) continue;
                    km_asgn_read_t& r = m ? kam.r2 : kam.r1;
                    vector<uint64_t>& es = m ? caes2 : caes1;
                    for (int i = r.si_; i < r.ei_ - 1; ++i) {
                        auto e = es[i];
                        if (e == NAN64) { continue; }
                        if (tres.count(e) == 0) {
                            if (ev && *nev < evcap) ev[*nev] = BubEvent{pair_index, (uint32_t)m, (uint32_t)i, (uint32_t)destLocus, e};
                            ++*nev;
                        }
                    }
                }
            }
            stage = DBTK_STAGE_COUNTED;
        }
    } while (0);
    C[DBTK_C_NHASH0] += nhash0;
    C[DBTK_C_NHASH1] += nhash1;
    if (rec) {
        kam.r1.assign(kf1, hf1, bf1, qf1, af1, rm1, qm1, qn1);
        kam.r2.assign(kf2, hf2, bf2, qf2, af2, rm2, qm2, qn2);
        rec->pair = pair_index;
        rec->stage = stage;
        rec->dst = destLocus;
        rec->dst0 = destLocus0;
        rec->nm1 = nm1;
        rec->nm2 = nm2;
        fill_mate(rec->r1, kam.r1);
        fill_mate(rec->r2, kam.r2);
    }
}

void ref_align(void* h, const char* seq, const uint64_t* off, uint64_t npairs, uint32_t Cth, int okam, int qc,
               uint64_t* counts_fileorder, uint64_t* kmc, uint32_t* nmapread, uint64_t* C, dbtk_pair_rec_t* recs) {
    for (uint64_t p = 0; p < npairs; ++p) {
        ref_pair(h, seq + off[2 * p], off[2 * p + 1] - off[2 * p], seq + off[2 * p + 1], off[2 * p + 2] - off[2 * p + 1], Cth, okam,
                 qc, counts_fileorder, kmc, nmapread, C, recs ? recs + p : nullptr, (uint32_t)p);
    }
}

void ref_align_ex(void* h, const char* seq, const uint64_t* off, const char* qual, uint64_t npairs, uint32_t Cth, int okam, int qc,
                  uint32_t qth_, int bait, int bubbles_on, uint64_t* counts_fileorder, uint64_t* kmc, uint32_t* nmapread, uint64_t* C,
                  dbtk_pair_rec_t* recs, BubEvent* ev, uint64_t evcap, uint64_t* nev) {
    qth = qth_;
    *nev = 0;
    for (uint64_t p = 0; p < npairs; ++p) {
        ref_pair(h, seq + off[2 * p], off[2 * p + 1] - off[2 * p], seq + off[2 * p + 1], off[2 * p + 2] - off[2 * p + 1], Cth, okam,
                 qc, counts_fileorder, kmc, nmapread, C, recs ? recs + p : nullptr, (uint32_t)p,
                 qual ? qual + off[2 * p] : nullptr, qual ? qual + off[2 * p + 1] : nullptr, bait, bubbles_on, ev, evcap, nev);
    }
}

/* The hot loop with the v1.3 threading call sites live.  Everything up to the QC gate is ref_pair's sequence
 * (AQ.cpp:2035-2062); then the lines the reference keeps in comments are executed as written there:
 *   AQ.cpp:2072-2088  sam.init1/2 + isThreadFeasible on both mates (+ threadCheck with tc), alned = alned0 || alned1,
 *                     noncaVec2CaUmap of both mates' uncorrected k-mers, else destLocus = nloci
 *   AQ.cpp:2090-2092  nFeasibleReads += 2
 *   AQ.cpp:2189-2194  countMode 0 ("exact"): trKmers[p.first] += p.second for the k-mers found there
 *   AQ.cpp:2232-2240  -a: every walked pair -> sams; -ae: only pairs with destLocus != nloci
 * and the batch's records are printed by the reference's writeAlignments (AQ.cpp:1742-1759).
 * res / trecs: per walked pair (pair order); aln_text receives the lines; titles = '\n'-separated, one per pair. */
int64_t ref_align_v13(void* h, const char* seq, const uint64_t* off, const char* titles_blob, uint64_t npairs, uint32_t Cth, int qc,
                      uint32_t thread_cth, int correction, int tc, int aln, int aln_minimal, uint64_t* counts_fileorder, uint64_t* C,
                      dbtk_walk_res_t* res, dbtk_thread_rec_t* trecs, uint64_t rescap, uint64_t* nres, char* aln_text, uint64_t aln_cap) {
    RefDB& db = *(RefDB*)h;
    AbortTrap trap;
    const uint64_t nloci = db.nloci;
    uint16_t Cthreshold = Cth;
    vector<string> seqs, titles;
    {
        const char* t = titles_blob;
        for (uint64_t p = 0; p < npairs; ++p) {
            const char* e = t ? strchr(t, '\n') : nullptr;
            titles.push_back(t ? (e ? string(t, e) : string(t)) : string("r"));
            if (t) t = e ? e + 1 : nullptr;
            seqs.push_back(string(seq + off[2 * p], off[2 * p + 1] - off[2 * p]));
            seqs.push_back(string(seq + off[2 * p + 1], off[2 * p + 2] - off[2 * p + 1]));
        }
    }
    vector<uint64_t> alnindices;
    vector<sam_t> sams;
    *nres = 0;
    std::streambuf* olderr = std::cerr.rdbuf(nullptr);
    g_abort_armed = 1;
    int64_t rc = 0;
    if (sigsetjmp(g_abort_env, 1) == 0) {
        uint64_t seqi = 0;
        while (seqi < 2 * npairs) {
            vector<uint64_t> caks1, caks2, caes1, caes2;
            vector<kmerIndex_uint32_umap::iterator> its1, its2;
            vector<PE_KMC> dup;
            log_t log;
            int rm1 = 0, rm2 = 0, kf1 = 0, kf2 = 0, hf1 = 0, hf2 = 0, nm1 = 0, nm2 = 0;
            uint64_t destLocus, destLocus0 = NAN32, nhash0 = 0, nhash1 = 0;
            string* seq1 = &seqs[seqi];
            string* seq2 = &seqs[seqi + 1];
            seqi += 2;
            C[DBTK_C_NREADS] += 2;
            read2kmers_edges(caks1, caes1, *seq1, ksize);
            read2kmers_edges(caks2, caes2, *seq2, ksize);
            if (not caks1.size() or not caks2.size()) { C[DBTK_C_NSHORT] += 1; continue; }
            if (N_FILTER and NM_FILTER) {
                const bool sub = subfilter(caks1, caks2, db.kmerDBi, nhash0);
                C[DBTK_C_NHASH0] += nhash0;
                if (sub) { C[DBTK_C_SUBFILTERED] += 2; continue; }
            }
            kfilter(caks1, caks2, its1, its2, db.kmerDBi, Cthreshold, nhash1, kf1, kf2, rm1, rm2);
            C[DBTK_C_NHASH1] += nhash1;
            C[DBTK_C_KMERFILTERED] += kf1 + kf2;
            if (rm1 and rm2) { continue; }
            destLocus = countHit(db.kmerDBi_vv, its1, its2, db.hits1, db.hits2, dup, nloci, Cthreshold, log, destLocus0, nm1, nm2, hf1, hf2, rm1, rm2);
            C[DBTK_C_LOCUSFILTERED] += hf1 + hf2;
            if (destLocus == nloci) { continue; }
            if (qc and not db.qcFilter[destLocus]) { C[DBTK_C_QCFILTERED] += 2 - rm1 - rm2; continue; }

            bool alned = false;
            int alned0 = 0, alned1 = 0;
            sam_t sam;
            GraphType& gf = db.graphDB[destLocus];
            C[DBTK_C_THREADING] += 2;
            kmerCount_umap cakmers;
            vector<uint64_t> noncakmers0, noncakmers1, akmers0, akmers1;
            const uint64_t walked = destLocus;
            {   // if (threading) {            AQ.cpp:2072-2088
                sam.init1(*seq1);
                alned0 = isThreadFeasible(gf, *seq1, noncakmers0, akmers0, thread_cth, correction != 0, sam.r1, db.trKmerDB[destLocus], log);
                sam.init2(*seq2);
                alned1 = isThreadFeasible(gf, *seq2, noncakmers1, akmers1, thread_cth, correction != 0, sam.r2, db.trKmerDB[destLocus], log);
                if (*nres < rescap && trecs) {  // (what the walk left, before threadCheck may mark cg.tr)
                    fill_thread_rec(&trecs[2 * *nres], alned0, sam.r1, akmers0);
                    fill_thread_rec(&trecs[2 * *nres + 1], alned1, sam.r2, akmers1);
                }
                if (tc) {
                    if (alned0) { threadCheck(gf, *seq1, akmers0, sam.r1, log); }
                    if (alned1) { threadCheck(gf, *seq2, akmers1, sam.r2, log); }
                }
                if (alned0 or alned1) {
                    alned = true;
                    noncaVec2CaUmap(noncakmers0, cakmers, ksize);
                    noncaVec2CaUmap(noncakmers1, cakmers, ksize);
                }
                else { destLocus = nloci; } // removed by threading
            }
            if (alned) {                    // AQ.cpp:2090-2092, 2189-2194
                C[DBTK_C_FEASIBLE] += 2;
                kmer_aCount_umap& trKmers = db.trKmerDB[walked];
                auto& fidx = db.fileIndex[walked];
                for (auto& p : cakmers) {
                    auto it = trKmers.find(p.first);
                    if (it != trKmers.end()) { counts_fileorder[fidx[p.first]] += p.second; C[DBTK_C_ALGO_INC] += p.second; }
                }
            }
            if (*nres < rescap && res) {
                dbtk_walk_res_t& w = res[*nres];
                w.pair = (uint32_t)(seqi / 2 - 1); w.dst = (uint32_t)destLocus; w.ret1 = (int8_t)alned0; w.ret2 = (int8_t)alned1; w.pad[0] = w.pad[1] = 0;
            }
            ++*nres;
            if (aln) {                      // AQ.cpp:2232-2240 (not simmode)
                if ((aln_minimal and destLocus != nloci) or (not aln_minimal)) {
                    alnindices.push_back(seqi);
                    sam.src = -1;           // srcLocus = -1 outside simulation mode (AQ.cpp:2000)
                    sam.dst = destLocus;
                    sams.push_back(sam);
                }
            }
        }
        if (aln && aln_text) {
            CoutCapture cap;
            writeAlignments(seqs, titles, alnindices, sams);
            const string out = cap.ss.str();
            rc = (int64_t)out.size();
            if (out.size() < aln_cap) memcpy(aln_text, out.data(), out.size());
            if (aln_cap) aln_text[out.size() < aln_cap ? out.size() : aln_cap - 1] = 0;
        }
        g_abort_armed = 0;
    } else {
        rc = -1;
    }
    std::cerr.rdbuf(olderr);
    std::cerr.clear();
    return rc;
}

}  // extern "C"
