/*
 * dbtk_oracle.h — CPU oracle for the `danbing-tk align` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm (ChaissonLab/danbing-tk, src/aQueryFasta_thread.cpp) used as the
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Nothing in the product (danbing-tk_amd/) may include, link or call it.
 *
 * Parity pinning: the reference holds no golden vectors for this path
 * (SURVEY.md 4, 8c).  The oracle is pinned against the reference itself,
 * compiled from /root/reference into oracle/_ref/ (Makefile in this
 * directory): function by function through oracle/_ref/libdbtk_refharness.so
 * and end to end against oracle/_ref/danbing-tk, and against the fixtures
 * those produced under tests/golden/.
 */
#ifndef DBTK_ORACLE_H_
#define DBTK_ORACLE_H_

#include "../include/dbtk.h" /* struct layouts only (params, records, flat RPGG) */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_rpgg orc_rpgg_t;

orc_rpgg_t* orc_rpgg_from_arrays(const dbtk_rpgg_arrays_t* a);
/* Reads PREF.kmers.dbi / PREF.fl.kdb / PREF.tre.kdb / PREF.tr.kmers (+qc). */
orc_rpgg_t* orc_rpgg_load(const char* prefix, uint32_t ksize, const char* qc_file);
void        orc_rpgg_free(orc_rpgg_t* g);
uint64_t    orc_rpgg_nloci(const orc_rpgg_t* g);
uint64_t    orc_rpgg_ntrkmers(const orc_rpgg_t* g);
const uint64_t* orc_rpgg_tr_cnt(const orc_rpgg_t* g);
const uint64_t* orc_rpgg_tr_ks(const orc_rpgg_t* g);

/* The hot loop over a batch (src/aQueryFasta_thread.cpp:2002-2249).
 * counts_fileorder[ntrkmers] is indexed like PREF.tr.kmers (file order), NOT
 * like OUT.trkmc.ar; all outputs ACCUMULATE.  recs: NULL, or npairs records
 * (one per pair, `trace` semantics of include/dbtk.h). */
/* vv words read by the votes (find_matching_locus) of every orc_align* call since the last reset: see dbtk_oracle.c */
uint64_t orc_vote_vv(void);
void orc_vote_vv_reset(void);
int orc_align(const orc_rpgg_t* g, const dbtk_params_t* p, const uint8_t* seq,
              const uint64_t* off, uint64_t npairs, uint64_t* counts_fileorder, uint64_t* kmc,
              uint32_t* nmapread, uint64_t* counters, dbtk_pair_rec_t* recs);

/* Novel-edge events of -bu in processing order (countNovelEdges, AQ.cpp:1559-1567): pair, mate
 * (0 = seq1), position, destLocus, canonical (k+1)-mer. */
typedef struct orc_bub_event { uint32_t pair, mate, pos, locus; uint64_t edge; } orc_bub_event_t;

/* Same loop with the optional gates: qual (NULL for FASTA; same offsets as seq) feeds the bait
 * filter's quality mask (-b with -fq); bubble events are appended to ev[0..cap) and *nev gets
 * the number produced (ev may be NULL when !p->bubbles). */
int orc_align_ex(const orc_rpgg_t* g, const dbtk_params_t* p, const uint8_t* seq, const uint64_t* off,
                 const uint8_t* qual, uint64_t npairs, uint64_t* counts_fileorder, uint64_t* kmc, uint32_t* nmapread,
                 uint64_t* counters, dbtk_pair_rec_t* recs, orc_bub_event_t* ev, uint64_t cap, uint64_t* nev);
/* The same loop with threading = 2 (the v1.3 call sites, AQ.cpp:2072-2088, 2189-2194): every pair that reaches threading
 * appends a result (pair order) and, when trecs != NULL, its two thread records.  n = results produced (may exceed cap). */
typedef struct orc_walk_out { dbtk_walk_res_t* res; dbtk_thread_rec_t* trecs; uint64_t cap, n; } orc_walk_out_t;
int orc_align_walk(const orc_rpgg_t* g, const dbtk_params_t* p, const uint8_t* seq, const uint64_t* off,
                   const uint8_t* qual, uint64_t npairs, uint64_t* counts_fileorder, uint64_t* kmc, uint32_t* nmapread,
                   uint64_t* counters, dbtk_pair_rec_t* recs, orc_bub_event_t* ev, uint64_t cap, uint64_t* nev, orc_walk_out_t* walk);
/* Attach a bait DB (PREF.bt.kmdb layout: per-locus counts, k-mers, (min<<8|max)). */
void orc_rpgg_set_bait(orc_rpgg_t* g, const uint64_t* bt_cnt, const uint64_t* bt_ks, const uint16_t* bt_vs);
int  orc_rpgg_load_bait(orc_rpgg_t* g, const char* bait_file);
/* qString2qMask, AQ.h:1038-1071: mask[i] = 1 iff k-mer i passes (mask has nq-k+1 bytes, pre-zeroed by the callee). */
void orc_qstring2qmask(const uint8_t* qual, int nq, int qth, int ksize, uint8_t* mask);

/* graphDB for the v1.3 threading path: PREF.graph.kmers (text) or PREF.graph.umap (v1.3 binary). */
void orc_rpgg_set_graph(orc_rpgg_t* g, const uint64_t* gr_cnt, const uint64_t* gr_ks, const uint8_t* gr_ms);
int  orc_rpgg_load_graph(orc_rpgg_t* g, const char* graph_file);
int  orc_rpgg_has_graph(const orc_rpgg_t* g);
uint64_t orc_rpgg_graph_dump(const orc_rpgg_t* g, uint32_t locus, uint64_t* ks, uint8_t* ms, uint64_t cap);
/* isThreadFeasible (AQ.cpp:1114-1260) after cigar_t::init, for one read against graphDB[locus] / trKmers[locus].
 * Returns 0 / 1 / 2 like the reference, -1 where the reference would assert (flags in out).  out and
 * noncakmers (>= len entries: the uncorrected k-mers of read2kmers(canonical=false, keepN=true)) may be NULL. */
int  orc_thread(const orc_rpgg_t* g, uint64_t locus, const uint8_t* seq, uint64_t len, uint32_t k, uint32_t thread_cth,
                int correction, uint32_t maxncorrection, dbtk_thread_rec_t* out, uint64_t* noncakmers);
uint64_t orc_read2kmers_nonca(const uint8_t* read, uint64_t rlen, uint32_t k, uint64_t* kmers); /* AQ.h:246-271 */
/* writeCigar / writeAnnot (AQ.cpp:1683-1740) into buf; return the length the text needs. */
size_t orc_write_cigar(const dbtk_thread_rec_t* r, char* buf, size_t cap);
size_t orc_write_annot(const dbtk_thread_rec_t* r, char* buf, size_t cap);
/* one line of writeAlignments (AQ.cpp:1742-1759); src < 0 prints '.' */
size_t orc_write_alignment(int64_t src, uint32_t dst, const char* title, const uint8_t* seq1, uint64_t l1, const uint8_t* seq2, uint64_t l2,
                           const dbtk_thread_rec_t* r1, const dbtk_thread_rec_t* r2, char* buf, size_t cap);

/* Function-level restatements (for pinning against the reference harness). */
uint64_t orc_nurc(uint64_t kmer, uint32_t k);                               /* getNuRC  AQ.h:165-178 */
uint64_t orc_read2kmers_edges(const uint8_t* read, uint64_t rlen, uint32_t k,
                              uint64_t* kmers, uint64_t* edges);            /* AQ.h:274-311; returns kmers.size() */
void     orc_sort_index(const uint64_t* data, uint64_t n, uint64_t* idx);   /* getSortedIndex AQ.cpp:247-250 == GCC std::sort */
/* libstdc++ std::unordered_map<size_t,...> iteration order after inserting
 * keys[0..n) with operator[] (AQ.h:469-480 then AQ.h:929-936 / BIO:36-46):
 * order[j] = index into keys of the j-th element visited. Returns 0 on success. */
int      orc_umap_order(const uint64_t* keys, uint64_t n, uint64_t* order);

#ifdef __cplusplus
}
#endif
#endif
