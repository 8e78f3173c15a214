/*
 * dbtk.h — C-ABI of the MI355X-native `danbing-tk align` hot path.
 *
 * The reference (ChaissonLab/danbing-tk) has no plugin/FFI interface: its
 * boundary is the process (argv, RPGG files, FASTA/FASTQ, stdout, output
 * files) and, inside the process, the worker seam `CountWords<T>(void*)`
 * (src/aQueryFasta_thread.cpp:1802-2283).  The entry points below sit exactly
 * on that seam: what the worker does between its two critical sections
 * (src/aQueryFasta_thread.cpp:1988-2249) for one batch of read pairs, plus the
 * loaders (src/aQueryFasta_thread.cpp:2459-2500) and the dumps
 * (src/aQueryFasta_thread.cpp:2631-2656) either side of it.
 *
 * Conventions: every function returns a dbtk_status_t (0 = ok); no exceptions
 * cross the boundary; all buffers are caller-owned unless stated; handles are
 * opaque.  A dbtk_rpgg_t is immutable after load and may be shared by any
 * number of contexts; a dbtk_ctx_t belongs to one GPU and one host thread at a
 * time.  There is no CPU fallback: if no HIP device is usable, ctx creation
 * fails with DBTK_ERR_NO_DEVICE.
 */
#ifndef DBTK_H_
#define DBTK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DBTK_ABI_VERSION 8u  /* v8: dbtk_ingest_reserve_host; v7: DBTK_C_ALGO_VV = the vv words fillstats reads (the vote's words: DBTK_PS_VOTE_VV); DBTK_PS_PAIR_VV */

/* Reads longer than this are rejected (DBTK_ERR_READ_TOO_LONG).  The
 * reference's per-read k-mer multiplicity is a uint8_t pair (`PE_KMC`,
 * src/aQueryFasta_thread.cpp:42: "not compatible with reads longer than
 * 255 bp"); 256 is one wavefront x 4 k-mer positions per lane. */
#define DBTK_MAX_READ_LEN 256u
#define DBTK_NAN64 0xFFFFFFFFFFFFFFFFull
#define DBTK_NAN32 0xFFFFFFFFu

typedef enum dbtk_status {
    DBTK_OK = 0,
    DBTK_ERR_ARG = 1,            /* null / inconsistent argument            */
    DBTK_ERR_IO = 2,             /* missing or truncated file (reference: assert -> abort) */
    DBTK_ERR_FORMAT = 3,         /* file does not parse                      */
    DBTK_ERR_NO_DEVICE = 4,      /* no usable HIP device: there is no CPU path */
    DBTK_ERR_HIP = 5,            /* a HIP runtime call failed (see dbtk_last_error) */
    DBTK_ERR_READ_TOO_LONG = 6,
    DBTK_ERR_NOMEM = 7,
    DBTK_ERR_UNSUPPORTED = 8,    /* flag combination the reference accepts but HEAD leaves dead */
    DBTK_ERR_OVERFLOW = 9        /* record buffer too small; nothing was lost, call again larger */
} dbtk_status_t;

/* ---- parameters: the reference's globals + Counts flags ------------------
 * src/aQueryFasta_thread.cpp:26-34 (globals), 2336-2339 (defaults),
 * 2344-2429 (flags). */
typedef struct dbtk_params {
    uint32_t ksize;        /* -k   [21]  (2..31: edges need 2(k+1) <= 64 bits) */
    uint32_t n_filter;     /* -kf N  [4]  N_FILTER  */
    uint32_t nm_filter;    /* -kf M  [1]  NM_FILTER */
    uint32_t cthreshold;   /* -cth [10]  (uint16_t in the reference) */
    uint32_t nm_tr;        /* -c   [40]  NM_TR      */
    uint32_t max_nt;       /*      [2]   MAX_NT (not a CLI flag) */
    uint32_t qth;          /* -qth [20]  bait quality threshold */
    uint32_t okam;         /* !-ka [1]   emit kmer-assignment records */
    uint32_t qc;           /* -qc        per-locus QC mask present in the RPGG handle */
    uint32_t bait;         /* -b         bait filter (FPSv1); needs the bait DB in the RPGG handle */
    uint32_t bubbles;      /* -bu        count novel (k+1)-mers; needs PREF.tre.kdb; host-buffer batches only */
    uint32_t extract;      /* -e 1|2     extract mode: locus assignment only */
    uint32_t trace;        /* test hook: emit a record for EVERY pair, not only kam ones */
    uint32_t threading;    /* 0 off.  1 = -g/-gc/-gcc as the mounted HEAD runs them: the call sites are commented out
                              (AQ.cpp:2072-2088), so assigned pairs are only counted as "entered threading".
                              2 = the v1.3 contract those comments spell out (DBTK_THREADING_V13, CLI: --v13-threading):
                              both mates are walked through graphDB[destLocus] with isThreadFeasible (AQ.cpp:1114-1260),
                              the pair is kept if either walk is feasible, TR k-mers are counted in "exact" mode
                              (AQ.cpp:2072-2088, 2189-2194); needs the graph in the RPGG handle (DBTK_LOAD_GRAPH) */
    uint32_t simmode;      /* -s 1|2: also emit records of pairs whose two mates assignTRkmc rejected (AQ.cpp:2169) */
    uint32_t thread_cth;   /* -g/-gc/-gcc N [100]  minimal number of walked k-mers per read (AQ.cpp:2338, 1122) */
    uint32_t maxncorrection; /* -gc/-gcc N M [4]   corrections allowed per read (AQ.cpp:33) */
    uint32_t correction;   /* -gc/-gcc: error correction on (plain -g walks without it) */
    uint32_t aln;          /* 0; 1 = -a: an alignment record for every walked pair; 2 = -ae: only for kept pairs (AQ.cpp:2232-2248);
                              | DBTK_ALN_TEXT: the records in text form (dbtk_ctx_aln_text) instead of the arrays of dbtk_ctx_aln_records */
    uint32_t trackbait;    /* -tb: per locus, count the bait k-mer that made bfilter_FPSv1 flag a mate (AQ.cpp:1391,1414);
                              with -b, host-buffer batches only; such pairs then also yield records (stage DBTK_STAGE_BAIT) */
    uint32_t diag;         /* 0.  Diagnostic knobs of the profiling tools (tools/k1_limits.py): results are wrong when set */
} dbtk_params_t;

/* ---- RPGG in flat (file-equivalent) form ----------------------------------
 * Exactly the arrays the reference's loaders read:
 *   keys/vals/vv      PREF.kmers.dbi   readBinaryIndex      src/aQueryFasta_thread.h:654-673
 *   fl_cnt/fl_ks      PREF.fl.kdb      readBinaryKmerSetDB  src/aQueryFasta_thread.h:675-698
 *   tre_cnt/tre_ks    PREF.tre.kdb     (same reader)
 *   tr_cnt/tr_ks      PREF.tr.kmers    readKmersWithZeroCount src/aQueryFasta_thread.h:469-480
 *                     (k-mers in FILE order, per locus)
 *   qc                -qc FILE         readQCFile           src/kmerIO.hpp:111-120 (0/1 per locus, may be NULL)
 *   bt_*              PREF.bt.kmdb     readBinaryBaitDB     src/aQueryFasta_thread.h:542-547 (may be NULL)
 *   gr_*              PREF.graph.kmers readGraphKmers       src/aQueryFasta_thread.h:550-575 (may be NULL)
 */
typedef struct dbtk_rpgg_arrays {
    uint32_t ksize;
    uint64_t nloci;
    uint64_t nkeys;  const uint64_t* keys;  const uint32_t* vals;
    uint64_t nvv;    const uint32_t* vv;
    const uint64_t* fl_cnt;  const uint64_t* fl_ks;    /* fl_cnt[nloci]; fl_ks[sum]   */
    const uint64_t* tre_cnt; const uint64_t* tre_ks;   /* may be NULL when !bubbles   */
    const uint64_t* tr_cnt;  const uint64_t* tr_ks;    /* file order                  */
    const uint8_t*  qc;                                 /* NULL or nloci bytes 0/1     */
    const uint64_t* bt_cnt;  const uint64_t* bt_ks;  const uint16_t* bt_vs; /* NULL or bait DB */
    /* graphDB (v1.3 threading): per locus the de Bruijn nodes (non-canonical k-mers of both strands) and their
     * out-edge masks, bit b = successor ((node & rmask) << 2) | b exists (T/G/C/A = bits 3..0).
     * PREF.graph.kmers (text, readGraphKmers src/aQueryFasta_thread.h:550-575) or PREF.graph.umap (v1.3 binary).
     * NULL when no graph is loaded. */
    const uint64_t* gr_cnt;  const uint64_t* gr_ks;  const uint8_t* gr_ms;
} dbtk_rpgg_arrays_t;

typedef struct dbtk_rpgg dbtk_rpgg_t;
typedef struct dbtk_ctx  dbtk_ctx_t;

#define DBTK_THREADING_HEAD 1u
#define DBTK_THREADING_V13  2u

/* ---- graph walk: what isThreadFeasible leaves behind for one read ------------
 * (src/aQueryFasta_thread.cpp:1114-1260; cigar_t / edit_t at :46-68).  The walk may insert k-mers and
 * edit operations (deletions), so the arrays have room for DBTK_THREAD_CAP entries:
 * every inserted k-mer is paid for with >= 5 extended ones (MSC, AQ.cpp:1120), so a read of
 * DBTK_MAX_READ_LEN bases cannot grow past 5/4 of its size. */
#define DBTK_THREAD_CAP 384u
typedef struct dbtk_thread_rec {
    int32_t  ret;      /* 0 infeasible, 1 feasible, 2 feasible with corrections */
    int32_t  ni;       /* cg.ni */
    uint32_t nkm;      /* kmers.size(): the walked (corrected) k-mers, non-canonical, NAN64 where the read had none */
    uint32_t nes;      /* cg.es.size(): one edit per read base + one per deletion */
    uint32_t ntr;      /* cg.tr.size(): one annotation per walked k-mer */
    uint32_t flags;    /* DBTK_THREAD_F_* */
    uint8_t  es_t[DBTK_THREAD_CAP];  /* edit_t::t  '*' '=' 'X' 'D' 'I' */
    uint8_t  es_r[DBTK_THREAD_CAP];  /* edit_t::r  read base, 0 for an inserted deletion */
    uint8_t  es_g[DBTK_THREAD_CAP];  /* edit_t::g  graph base ('A','C','G','T'), 0 when none */
    uint8_t  tr[DBTK_THREAD_CAP];    /* '*' unaligned, '.' flank, '=' TR */
    uint64_t kmers[DBTK_THREAD_CAP];
} dbtk_thread_rec_t;
#define DBTK_THREAD_F_OVERFLOW 1u    /* the walk outgrew DBTK_THREAD_CAP (never at <= DBTK_MAX_READ_LEN) */
#define DBTK_THREAD_F_MISSING_NODE 2u /* the reference would assert: a successor named by an edge mask is not in the graph (AQ.cpp:528-532) */

/* ---- per-pair record -------------------------------------------------------
 * Fields of `km_asgn_t` / `km_asgn_read_t` (src/aQueryFasta_thread.cpp:93-144)
 * as they stand when the reference pushes a kam record
 * (src/aQueryFasta_thread.cpp:2169-2175).  `as` is 2 bits per k-mer position
 * (0 '*', 1 '.', 2 '='), position i in bits [2*(i%4), +2) of byte i/4. */
typedef struct dbtk_mate_rec {
    int16_t si, ei, si_, ei_, nt, bs, ti;   /* -1 = unset, as in the reference */
    uint8_t kf, hf, bf, qf, af, rm;         /* filter flags */
    uint16_t nk;                            /* size of `as` (0 when assignTRkmc did not run) */
    uint8_t as2[DBTK_MAX_READ_LEN / 4];
} dbtk_mate_rec_t;

enum {
    DBTK_STAGE_SHORT = 0,     /* no valid k-mer window in a mate      AQ.cpp:2037 */
    DBTK_STAGE_SUBFILTER = 1, /* rejected by subfilter                AQ.cpp:2046 */
    DBTK_STAGE_KFILTER = 2,   /* both mates rejected by kfilter       AQ.cpp:2054 */
    DBTK_STAGE_LOCUS = 3,     /* countHit found no locus              AQ.cpp:2058 */
    DBTK_STAGE_QC = 4,        /* locus fails QC mask                  AQ.cpp:2059 */
    DBTK_STAGE_BAIT = 5,      /* bait filter removed the pair         AQ.cpp:2120 */
    DBTK_STAGE_ASGN = 6,      /* both mates rejected by assignTRkmc   AQ.cpp:2145 */
    DBTK_STAGE_COUNTED = 7,   /* counts were accumulated              AQ.cpp:2146 */
    DBTK_STAGE_EXTRACT = 8,   /* -e: pair assigned, reported only     AQ.cpp:2094 */
    DBTK_STAGE_THREADING = 9  /* threading = 2: the pair reached the graph walk (AQ.cpp:2070); what the walk decided is in
                                 dbtk_ctx_walk_results */
};

typedef struct dbtk_pair_rec {
    uint32_t pair;        /* index of the pair inside the batch */
    uint32_t stage;       /* DBTK_STAGE_* where the pair ended  */
    uint32_t dst;         /* destLocus (== nloci when unassigned) */
    uint32_t dst0;        /* destLocus0 = top.idx (DBTK_NAN32 before countHit) */
    int32_t  nm1, nm2;    /* top.fc / top.rc (partial sums, AQ.cpp:436-438): trace mode only, 0 otherwise (not part of any output) */
    dbtk_mate_rec_t r1, r2;
} dbtk_pair_rec_t;

/* Global counters, in the order the reference prints them
 * (src/aQueryFasta_thread.cpp:2617-2626), then the per-batch extras
 * (src/aQueryFasta_thread.cpp:2266-2277). */
enum {
    DBTK_C_NREADS = 0, DBTK_C_SUBFILTERED, DBTK_C_KMERFILTERED, DBTK_C_BAITFILTERED,
    DBTK_C_QUALFILTERED, DBTK_C_LOCUSFILTERED, DBTK_C_QCFILTERED, DBTK_C_THREADING,
    DBTK_C_FEASIBLE, DBTK_C_ASGN, DBTK_C_NSHORT, DBTK_C_NHASH0, DBTK_C_NHASH1,
    /* algorithmic work of SURVEY.md 8(d): B = 2L + 12 P + 4 V + 8 A + 16 I per pair */
    DBTK_C_ALGO_PROBES,   /* P: index lookups the reference algorithm performs (subfilter + kfilter) */
    DBTK_C_ALGO_VV,       /* V: uint32 words of vv read by fillstats (one per distinct k-mer whose index value is a list, AQ.cpp:311-316).  The
                           * list words find_matching_locus goes on to read depend on the order std::sort leaves equal keys in, and a pair
                           * whose outcome is proven without the vote reads none: they are the path statistic DBTK_PS_VOTE_VV, not a counter */
    DBTK_C_ALGO_CLS,      /* A: k-mers classified by assignTRkmc (one flank/TR lookup each) */
    DBTK_C_ALGO_INC,      /* I: TR k-mer count increments */
    DBTK_C_SURVIVORS,     /* pairs that passed subfilter (entered kfilter) */
    DBTK_C_BASES,         /* bases of all reads handed to the hot loop (sum of L) */
    DBTK_C_COUNT = 24
};

/* ---- RPGG -----------------------------------------------------------------*/
/* Load PREF.{tr.kmers,kmers.dbi,fl.kdb,tre.kdb} (+ optional qc / bait files,
 * NULL to skip).  Replaces src/aQueryFasta_thread.cpp:2459,2490-2500. */
#define DBTK_LOAD_INDEX_ONLY 1u  /* -e (extract) mode: only PREF.tr.kmers + PREF.kmers.dbi are read (AQ.cpp:2484-2488) */
#define DBTK_LOAD_GRAPH 2u       /* also graphDB for threading = 2: PREF.graph.kmers (text, readGraphKmers AQ.h:550-575), or,
                                    when that file is absent, the v1.3 binary PREF.graph.umap */
dbtk_status_t dbtk_rpgg_load(const char* prefix, uint32_t ksize, const char* qc_file,
                             const char* bait_file, uint32_t flags, dbtk_rpgg_t** out);
/* The same with the TR k-mer file named explicitly (NULL: PREF.tr.kmers): `-t N` reads PREF.tr.trimN.kmers in its place —
 * locus count, trKmerDB and with it the OUT.trkmc.ar order (src/aQueryFasta_thread.cpp:2352, 2389, 2459, 2493); every other
 * file still comes from PREF. */
dbtk_status_t dbtk_rpgg_load_tr(const char* prefix, const char* tr_kmers_file, uint32_t ksize, const char* qc_file,
                                const char* bait_file, uint32_t flags, dbtk_rpgg_t** out);
/* Same handle from caller arrays (copied). */
dbtk_status_t dbtk_rpgg_from_arrays(const dbtk_rpgg_arrays_t* a, dbtk_rpgg_t** out);
void          dbtk_rpgg_free(dbtk_rpgg_t* h);
/* Process-unique id of the handle, never reused (> 0).  The HBM tables built from a handle are cached per (id, device) and
 * shared by the contexts created from it; the id — not the handle's address, which the allocator hands out again — is the key,
 * so tables can never be taken for another RPGG's.  The handle must still outlive its contexts. */
uint64_t      dbtk_rpgg_uid(const dbtk_rpgg_t* h);
/* The sidecar of the GPU-layout index: the per-locus images of PREF.kmers.dbi the probe kernel keeps in LDS (what `ktools serialize`
 * is to the reference's loaders, src/kmertools.cpp:221-345, one step further: the layout the GPU reads, built on the GPU in
 * dbtk_ctx_create).  mode 0: no file; 1: load `path` when it is there and was built from this RPGG (a fingerprint of the
 * handle's arrays and the layout version are checked, every image is verified on the device; anything else: built afresh);
 * 2: the same, and (re)write the file after a build.  dbtk_rpgg_load sets path = PREF.dbtk.idx and the mode from
 * DBTK_IDX_CACHE (default 1); a handle made from arrays has no file until this call names one. */
dbtk_status_t dbtk_rpgg_set_index_cache(dbtk_rpgg_t* h, const char* path, int mode);
uint64_t      dbtk_rpgg_nloci(const dbtk_rpgg_t* h);
uint64_t      dbtk_rpgg_ntrkmers(const dbtk_rpgg_t* h);  /* == length of the counts vector */
uint64_t      dbtk_rpgg_nkeys(const dbtk_rpgg_t* h);
/* Flat view of the loaded arrays (valid while the handle lives). */
dbtk_status_t dbtk_rpgg_view(const dbtk_rpgg_t* h, dbtk_rpgg_arrays_t* out);
/* Output order: out_slot[i] = position in OUT.trkmc.ar of the i-th k-mer of
 * PREF.tr.kmers (file order, loci concatenated).  This is the iteration order
 * of the reference's per-locus std::unordered_map (src/binaryKmerIO.hpp:31-51,
 * src/aQueryFasta_thread.h:926-937). */
dbtk_status_t dbtk_rpgg_output_order(const dbtk_rpgg_t* h, uint64_t* out_slot);

/* ---- context (one per GPU) ------------------------------------------------*/
void          dbtk_params_default(dbtk_params_t* p);
/* Optional: start the HIP runtime on `device_id` (its first call costs a few tenths of a second: library load, device context).  A caller
 * that still has files to parse — dbtk_rpgg_load — may do this on a thread of its own meanwhile; dbtk_ctx_create does it otherwise. */
dbtk_status_t dbtk_device_warmup(int device_id);
dbtk_status_t dbtk_ctx_create(const dbtk_rpgg_t* h, const dbtk_params_t* p, int device_id,
                              dbtk_ctx_t** out);
void          dbtk_ctx_free(dbtk_ctx_t* ctx);

/* One batch of complete read pairs: the body of the hot loop
 * (src/aQueryFasta_thread.cpp:2002-2249).  Read 2p is `seq1`, read 2p+1 is
 * `seq2` of pair p (the reference's seqs[seqi], seqs[seqi+1]); read r occupies
 * seq_bytes[seq_offsets[r] .. seq_offsets[r+1]).  qual_bytes (same offsets) is
 * NULL for FASTA.  Counts accumulate inside the context.  Up to rec_cap
 * records are written to `recs` in pair order; *nrec receives how many the
 * batch produced (DBTK_ERR_OVERFLOW if > rec_cap; counts are still complete).
 * recs may be NULL when the parameters produce no records. */
dbtk_status_t dbtk_align_batch(dbtk_ctx_t* ctx, const uint8_t* seq_bytes,
                               const uint64_t* seq_offsets, const uint8_t* qual_bytes,
                               uint64_t npairs, dbtk_pair_rec_t* recs, uint64_t rec_cap,
                               uint64_t* nrec);

/* ---- graph walk (threading = 2) ------------------------------------------------------------
 * Function-level entry: read r = seq_bytes[seq_offsets[r] .. seq_offsets[r+1]) is walked through graphDB[loci[r]] with
 * the context's thread_cth / correction / maxncorrection: cigar_t::init + isThreadFeasible
 * (src/aQueryFasta_thread.cpp:62-67, 1114-1260, as called at :2073-2076).  recs[nreads] receives what the walk left
 * behind; ret = -1 (with a flag) where the reference would have asserted, and for a read without any valid k-mer
 * (which the hot path never hands to the walk).  Host buffers; needs the graph in the RPGG handle (DBTK_LOAD_GRAPH). */
dbtk_status_t dbtk_thread_batch(dbtk_ctx_t* ctx, const uint8_t* seq_bytes, const uint64_t* seq_offsets,
                                const uint32_t* loci, uint64_t nreads, dbtk_thread_rec_t* recs);
/* What threading did to the pairs of the last dbtk_align_batch call that reached it (AQ.cpp:2070-2088), in pair order:
 * destLocus afterwards (nloci = neither mate's walk was feasible) and isThreadFeasible's two return codes
 * (ret1: seq1 = read 2p, ret2: seq2 = read 2p+1).  trecs (NULL, or 2 * cap entries: seq1's then seq2's record of
 * every result) needs params.trace or params.aln.  *n = number of results (DBTK_ERR_OVERFLOW if > cap). */
/* ret1 / ret2 = isThreadFeasible's return codes (0 infeasible, 1 feasible, 2 feasible after correction), or
 * DBTK_WALK_NOT_EVALUATED: the pair was kept by its OTHER mate, which threads through the graph (as it stands, or after correction), and
 * nothing asked for this mate's alignment (no -a / -ae records, no params.trace) — the call site only ever uses `alned0 || alned1` (AQ.cpp:2082-2087).
 * A mate that is not walked is not checked either: a condition only its walk would raise through the sticky error word (a walk longer than
 * DBTK_THREAD_CAP entries) is not reported for it in this mode; with records or trace both mates are walked and checked. */
#define DBTK_WALK_NOT_EVALUATED (-2)
typedef struct dbtk_walk_res { uint32_t pair, dst; int8_t ret1, ret2; uint8_t pad[2]; } dbtk_walk_res_t;
dbtk_status_t dbtk_ctx_walk_results(dbtk_ctx_t* ctx, dbtk_walk_res_t* res, dbtk_thread_rec_t* trecs, uint64_t cap, uint64_t* n);

/* -a / -ae (params.aln, threading = 2): the alignment records of the last dbtk_align_batch call, in pair order —
 * every pair that was walked (-a) or only those kept (-ae), AQ.cpp:2232-2240.  A record is a header followed by four
 * byte arrays of `cap` entries each: es1[cap] tr1[cap] es2[cap] tr2[cap] (1 = seq1 = read 2p, 2 = seq2 = read 2p+1);
 * es = cg.es with an edit in one byte: type (bits 0-2: 0 '*', 1 '=', 2 'X', 3 'D', 4 'I') | graph base << 3 (0 none,
 * 1 'A', 2 'C', 3 'G', 4 'T', 5 '*'), tr = cg.tr as characters.  dbtk_aln_format prints a record's four strings the way
 * writeAlignments does (writeCigar / writeAnnot, AQ.cpp:1683-1740): "cigar2 \t annot2 \t cigar1 \t annot1". */
typedef struct dbtk_aln_hdr {
    uint32_t pair, dst;       /* pair index inside the batch; destLocus after threading (nloci: removed by threading) */
    int8_t ret1, ret2;        /* isThreadFeasible's return codes */
    uint8_t pad[2];
    uint16_t nes1, ntr1, nes2, ntr2;
    uint32_t pad2;
} dbtk_aln_hdr_t;
/* buf receives *nrec records of *stride bytes each.  DBTK_ERR_OVERFLOW if buf_bytes is too small for every record the
 * batch may hold: nothing is copied, and *nrec (an upper bound of the record count) and *stride say what is needed. */
dbtk_status_t dbtk_ctx_aln_records(dbtk_ctx_t* ctx, void* buf, uint64_t buf_bytes, uint64_t* nrec, uint32_t* stride, uint32_t* cap);
/* Returns the length of the text (without the terminating NUL it also writes when it fits). */
size_t dbtk_aln_format(const void* rec, uint32_t cap, char* out, size_t out_cap);
/* params.aln | DBTK_ALN_TEXT: writeCigar / writeAnnot (AQ.cpp:1683-1740) run on the device, and what comes back is the text itself —
 * a few tens of bytes per pair instead of a fixed-size record.  idx[p] (p < idx_cap, the batch's pairs) = byte offset of pair p's
 * record in `arena`, or DBTK_NAN32 when it has none (never walked; -ae: not kept).  A record = {uint32 dst (nloci: removed by
 * threading), uint32 len} followed by len bytes "cigar2 \t annot2 \t cigar1 \t annot1" (not NUL-terminated), 4-byte aligned.
 * *arena_used = bytes of the arena in use; DBTK_ERR_OVERFLOW (nothing copied) when arena_cap is smaller. */
#define DBTK_ALN_TEXT 4u
dbtk_status_t dbtk_ctx_aln_text(dbtk_ctx_t* ctx, uint32_t* idx, uint64_t idx_cap, void* arena, uint64_t arena_cap, uint64_t* arena_used);

/* Device-resident variant used when the reads already sit in HBM (bench, or a
 * caller that overlaps its own H2D copies): d_seq / d_offsets are device
 * pointers with the same meaning, max_read_len the longest read in the batch.
 * d_seq must be 16-byte aligned and readable up to the end of the last read
 * rounded up to a multiple of 16 (any hipMalloc'd buffer is).  Asynchronous on
 * the context's streams (successive batches go round DBTK_LANES of them);
 * records are not produced.  A read longer than max_read_len (or than
 * DBTK_MAX_READ_LEN) is not validated on the host here: the device truncates it
 * to what was promised and raises an error word that the next
 * dbtk_ctx_synchronize / dbtk_ctx_counts / dbtk_allreduce returns, once, as
 * DBTK_ERR_READ_TOO_LONG; the accumulators are tainted from then on
 * (dbtk_ctx_reset clears both).
 * Stream ordering: the kernels run on the context's own streams, which are NOT
 * ordered against the stream that produced d_seq / d_offsets.  The inputs must
 * be complete before the call (synchronize the producing stream, or make it
 * wait on an event first) and must stay untouched until dbtk_ctx_synchronize
 * (or dbtk_ctx_counts / dbtk_allreduce) has returned. */
dbtk_status_t dbtk_align_batch_device(dbtk_ctx_t* ctx, const void* d_seq, const void* d_offsets,
                                      uint64_t npairs, uint32_t max_read_len);
dbtk_status_t dbtk_ctx_synchronize(dbtk_ctx_t* ctx);

/* ---- raw-bytes ingest: the reader of the batch loop on the device ---------------------------------------------------
 * Replaces critical section A of the worker (src/aQueryFasta_thread.cpp:1918-1976: getline title / seq [/ + / qual],
 * prunePEinfo :455-462, on-the-fly mate pairing, the minimal read size :1940-1943) for interleaved input.  The caller only
 * moves bytes: it reads the input in chunks of at most chunk_bytes, in file order, into the pinned buffer of the next
 * slot (slots are used round robin) and submits it; record boundaries (2 lines per FASTA record, 4 per FASTQ, counted
 * from the first submitted byte), pairing and the batch arrays are made by kernels, and what follows a chunk's last whole
 * pair is carried over to the next chunk on the device.
 *   submit(slot, nbytes, last)   asynchronous: host-to-device copy + the parse kernels of the block
 *   wait(slot, &info)            blocks until the block is parsed; info says what it held
 *   align(slot, ctx, ...)        the parsed pairs through the hot path (as dbtk_align_batch_device when sync = 0, as
 *                                dbtk_align_batch — records, dbtk_ctx_aln_text — when sync = 1)
 * Calls for successive blocks must be made in order by one thread; submit may run ahead of wait by up to nslots - 1
 * blocks.  A slot's buffer may be refilled once align (or wait, for a block that is not aligned) has returned and the
 * caller no longer needs its bytes (dbtk_ingest_spans point into it).
 * Parsing block i puts its carried-over bytes in front of slot (i + 1) % nslots's DEVICE block: submit block i only when the caller is
 * done with dbtk_ingest_aln_lines of that slot's previous block (block i + 1 - nslots).
 * info.flags != 0: the block is not (only) a run of adjacent mates.  DBTK_ING_DIRTY / DBTK_ING_LINES: nothing of it may be
 * aligned — continue at input offset info.first_byte with a host reader (up to there every record was paired, so nothing
 * is parked: the state the reference's reader would be in).  DBTK_ING_CARRY / DBTK_ING_TAIL alone: align it, then continue
 * at info.cut_byte with a host reader.  After a flagged block the ingest object accepts no further blocks. */
typedef struct dbtk_ingest dbtk_ingest_t;
#define DBTK_ING_DIRTY 1u   /* neighbouring records with different titles (or an odd record in between) */
#define DBTK_ING_LINES 2u   /* more lines than the line table holds (lines shorter than 8 bytes on average) */
#define DBTK_ING_CARRY 4u   /* the bytes after the last whole pair exceed the carry-over room (1 MB) */
#define DBTK_ING_TAIL  8u   /* the input does not end with a whole pair */
typedef struct dbtk_ingest_info {
    uint32_t flags, npairs, nkept, max_read_len;  /* pairs of records in the block; pairs kept (both reads >= min_read_size) */
    uint64_t first_byte, cut_byte;                /* input offsets (first submitted byte = 0) of the block's first record and
                                                     of the first byte after its last whole pair */
    uint64_t seq_bytes;
} dbtk_ingest_info_t;
/* kept pair q of a block: where its (pruned) title, reads and quality strings lie in the slot's block buffer
 * (dbtk_ingest_block); index 0 = read 2q (the record that completed the pair), 1 = read 2q + 1 (the parked one). */
typedef struct dbtk_ingest_span { uint32_t title, title_len, seq[2], seq_len[2], qual[2], qual_len[2]; } dbtk_ingest_span_t;
/* with_spans: dbtk_ingest_spans will be called (the per-pair spans are then made with every block). */
/* Optional (ABI v8): pin, ahead of dbtk_ingest_create and on any thread, the host buffers an ingest of this shape will use — `nslots`
 * chunk buffers for `chunk_bytes` (0: none) and, if `lines_bytes` != 0, as many buffers of `lines_bytes` for the -a / -ae lines of a block
 * (dbtk_ingest_aln_lines).  Pinning runs at a few GB/s and stalls every other allocation of the process meanwhile: a caller that still has
 * files to parse (dbtk_rpgg_load) does it beside that; what is not there when an ingest asks is pinned on first use.  The buffers belong to
 * the process: an ingest takes them, dbtk_ingest_free hands them back for the next one.  Replaces nothing in the reference (its reader
 * std::getline()s into std::strings, src/aQueryFasta_thread.cpp:1918-1976). */
dbtk_status_t dbtk_ingest_reserve_host(int device_id, uint64_t chunk_bytes, uint32_t nslots, uint64_t lines_bytes);
dbtk_status_t dbtk_ingest_create(dbtk_ctx_t* ctx, uint32_t fastq, uint32_t min_read_size, uint64_t chunk_bytes, uint32_t nslots,
                                 uint32_t with_spans, dbtk_ingest_t** out);
void          dbtk_ingest_free(dbtk_ingest_t* ing);
void*         dbtk_ingest_chunk_buffer(dbtk_ingest_t* ing, uint32_t slot);  /* pinned, chunk_bytes: read the chunk into it */
const void*   dbtk_ingest_block(dbtk_ingest_t* ing, uint32_t slot);         /* what the spans are offsets into (valid after wait) */
dbtk_status_t dbtk_ingest_submit(dbtk_ingest_t* ing, uint32_t slot, uint64_t nbytes, int last);
dbtk_status_t dbtk_ingest_wait(dbtk_ingest_t* ing, uint32_t slot, dbtk_ingest_info_t* info);
/* ctx: NULL = the ingest's own context; another context of the same RPGG on the same device lets several threads run the blocks of one
 * ingest through the hot path side by side (each thread its own context; sync = 1: its records, its text arena). */
dbtk_status_t dbtk_ingest_align(dbtk_ingest_t* ing, uint32_t slot, dbtk_ctx_t* ctx, int sync, dbtk_pair_rec_t* recs, uint64_t rec_cap, uint64_t* nrec);
/* The same without records, blocks MERGED into larger batches: the block of `slot` is appended (device to device) to the context's merged
 * batch, which is aligned once it holds min_pairs pairs, or at once with flush != 0 (slot = ~0u: nothing to append, flush only — call it
 * when the input has ended).  The reference cuts its input into batches of 300 000 reads per thread whatever they hold
 * (src/aQueryFasta_thread.cpp:1918-1976); how the pairs are cut changes no result (every effect of a pair is an integer add), but the
 * kernels that keep a locus' k-mers in LDS need batches with many pairs per locus.  Asynchronous like dbtk_ingest_align with sync = 0; the
 * slot may be submitted again as soon as the call returns.  Not with trace, -bu, or -b on FASTQ (those need a block's own records). */
dbtk_status_t dbtk_ingest_align_merged(dbtk_ingest_t* ing, uint32_t slot, dbtk_ctx_t* ctx, uint64_t min_pairs, int flush);
dbtk_status_t dbtk_ingest_spans(dbtk_ingest_t* ing, uint32_t slot, dbtk_ingest_span_t* spans, uint64_t cap);
/* -a / -ae with the device reader (params.aln | DBTK_ALN_TEXT): writeAlignments' lines (src/aQueryFasta_thread.cpp:1742-1759:
 * `. dst title seq2 seq1 cigar2 annot2 cigar1 annot1`) of the block aligned last with sync = 1, in pair order, assembled on the device
 * from the block's bytes and the walk's text records.  gz = 0: the text; gz != 0: gzip members (RFC 1952; 64 KB of text each, one
 * deflate block with a dynamic Huffman code over the literals, no string matching) whose concatenation `zcat`s to the text.
 * *data points at *nbytes bytes in a pinned buffer of the slot, valid until the slot's next submit.  *nlines = pairs with a record;
 * *text_bytes = bytes of the text. */
dbtk_status_t dbtk_ingest_aln_lines(dbtk_ingest_t* ing, uint32_t slot, dbtk_ctx_t* ctx /* the one that aligned the block; NULL: the ingest's */, int gz,
                                    const void** data, uint64_t* nbytes, uint64_t* nlines, uint64_t* text_bytes);

/* Copy the accumulated results to the host.  counts[ntrkmers] is in
 * OUT.trkmc.ar order; kmc[nloci]; nmapread[nloci]; counters[DBTK_C_COUNT].
 * Any pointer may be NULL. */
dbtk_status_t dbtk_ctx_counts(dbtk_ctx_t* ctx, uint64_t* counts, uint64_t* kmc,
                              uint32_t* nmapread, uint64_t* counters);
/* Device addresses + lengths of the same accumulators, for the one RCCL
 * all-reduce that replaces the reference's shared-memory atomics
 * (src/aQueryFasta_thread.cpp:2146-2158, 1887-1895) across GPUs.  All four live
 * in ONE contiguous uint64 buffer: *d_base, *n_u64 (nmapread widened to u64).
 * The buffer holds the results of every batch launched so far only after dbtk_ctx_synchronize (which also folds the kernels' private
 * counter replicas into counters[]): a device-wide or stream synchronize of the caller's own is NOT enough. */
dbtk_status_t dbtk_ctx_accum_buffer(dbtk_ctx_t* ctx, void** d_base, uint64_t* n_u64);
dbtk_status_t dbtk_ctx_reset(dbtk_ctx_t* ctx);

/* In-process multi-GPU reduce (one context per GPU, RCCL over xGMI). */
dbtk_status_t dbtk_allreduce(dbtk_ctx_t** ctxs, int n);

/* Per-kernel device time accumulated since dbtk_ctx_timers_reset (HIP events
 * recorded on the context's stream around every launch): total_ms[i] over
 * launches[i] launches of kernel names[i] (static strings).  Synchronises the
 * stream.  Returns how many kernels were filled (<= cap). */
int  dbtk_ctx_kernel_times(dbtk_ctx_t* ctx, const char** names, double* total_ms, uint64_t* launches, int cap);
/* HBM bytes of the context's RPGG tables (shared by the contexts of one handle on one device): names[i] (static strings) and
 * bytes[i]; returns how many were filled (<= cap).  The entry "index_images:from_cache" is 1 when the images came from the sidecar. */
int  dbtk_ctx_table_bytes(dbtk_ctx_t* ctx, const char** names, uint64_t* bytes, int cap);
/* Which kernels took how many pairs since the context was created / reset (diagnostic; never part of the results — the hot loop of
 * src/aQueryFasta_thread.cpp:2002-2249 has one path, this library several that must agree).  out[DBTK_PS_*]; returns words filled. */
#define DBTK_PATH_STATS 24u
#define DBTK_PS_PROBE_ITEMS   0u  /* [3] work items (locus, <= 64 pairs) of the locus-resident probe kernel, per class of image size */
#define DBTK_PS_PROBE_PAIRS   3u  /* [3] pairs in those items */
#define DBTK_PS_PROBE_REST    6u  /* pairs the lean probe kernel took from the list the locus path left (incl. pairs handed back) */
#define DBTK_PS_WALK_ITEMS    7u  /* [3] the same for the locus-resident form of the lean walk kernel */
#define DBTK_PS_WALK_PAIRS   10u  /* [3] */
#define DBTK_PS_WALK_REST    13u  /* pairs its global-table form took */
#define DBTK_PS_FUSED_DONE   14u  /* pairs the locus-resident probe kernel resolved itself (countHit shortcut + assignTRkmc + accumulate) */
#define DBTK_PS_FUSED_REDONE 15u  /* ... of the pairs it had resolved ahead of the global look-ups, those it had to take back */
#define DBTK_PS_FUSED_CLS    16u  /* of DBTK_C_ALGO_CLS / DBTK_C_ALGO_INC, the part that kernel did (its algorithmic bytes: 8 A + 16 I) */
#define DBTK_PS_FUSED_INC    17u
#define DBTK_PS_FUSED_SHARED 19u /* ... of the pairs resolved there, those with k-mers shared between loci (decided by the k-mers unique to the locus) */
#define DBTK_PS_LEAN_DONE    20u /* pairs the LEAN probe kernel resolved itself (the usual pair, and the pair kfilter removes altogether are not counted
                                  * here: only pairs resolved through assignTRkmc / QC / threading hand-over); its share of CLS / INC is in FUSED_CLS / _INC */
#define DBTK_PS_VOTE_VV      21u /* vv words read by the votes that were held (find_matching_locus, AQ.cpp:364-422: body_pair's pairs only; <= what the
                                  * reference reads for the same batch, which votes on every pair) */
#define DBTK_PS_PAIR_VV      22u /* of DBTK_C_ALGO_VV (fillstats' words), the part of the pairs body_pair handled: the rest was accounted by the fused probe
                                  * kernels, which read no vv word at all (bench.py prices each kernel with its own) */
#define DBTK_PS_WALK_LOCUS_EC 18u /* pairs the error-correcting walk took with the locus' graph image in LDS (k_walk_pairs_locus) */
int  dbtk_ctx_path_stats(dbtk_ctx_t* ctx, uint64_t* out, int cap);
void dbtk_ctx_timers_reset(dbtk_ctx_t* ctx);
void dbtk_ctx_timers_enable(dbtk_ctx_t* ctx, int on);  /* default 1: event records around every kernel of every batch (~30 us per batch);
                                                          * 0 = none; n > 1 = only around the kernels of every n-th batch (sampling) */

/* -tb: OUT.btk.kmdb = dumpBaitKmerHits (src/aQueryFasta_thread.h:1010-1012): the per-locus (bait k-mer -> times it was
 * the first violated one) maps in the reference's iteration order, serializeKmapDB layout with 8-byte values. */
dbtk_status_t dbtk_ctx_write_bait_hits(dbtk_ctx_t* ctx, const char* out_prefix);
dbtk_status_t dbtk_ctx_merge_bait_hits(dbtk_ctx_t* dst, dbtk_ctx_t* src);

/* -bu: the per-locus novel (k+1)-mer counts accumulated by dbtk_align_batch (host path) -> OUT.bub.kmdb
 * (dumpBubbles, src/aQueryFasta_thread.h:1006-1008: entries with count >= 5).  merge: fold another GPU's DB in. */
dbtk_status_t dbtk_ctx_write_bubbles(dbtk_ctx_t* ctx, const char* out_prefix);
dbtk_status_t dbtk_ctx_merge_bubbles(dbtk_ctx_t* dst, dbtk_ctx_t* src);

/* ---- dumps: src/aQueryFasta_thread.cpp:2631-2641 --------------------------*/
/* `ktools serialize PREF` (src/kmertools.cpp:221-345): PREF.tr.kmers + PREF.fl.kmers
 * -> PREF.kmers.dbi (readKmerIndex, src/kmerIO.hpp:47-78), PREF.fl.kmers ->
 * PREF.fl.kdb, PREF.tre.kmers -> PREF.tre.kdb (flattenKsetDB / serializeKsetDB,
 * src/binaryKmerIO.hpp:116-139), byte for byte (the orders inside the files are
 * libstdc++ unordered_map / unordered_set iteration orders, reproduced with the
 * same containers filled in the same order).  Host only, no GPU needed. */
dbtk_status_t dbtk_rpgg_serialize(const char* prefix);

/* with_names = 0: OUT.trkmc.ar + OUT.tr.summary.txt; 1: OUT.tr.kmers (-on). */
dbtk_status_t dbtk_write_outputs(const dbtk_rpgg_t* h, const uint64_t* counts, const uint64_t* kmc,
                                 const uint32_t* nmapread, const char* out_prefix, int with_names);

const char* dbtk_last_error(void);
uint32_t    dbtk_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DBTK_H_ */
