/* dbtk_pred.h — C-ABI of the `danbing-tk-pred` step: read-depth normalisation and invariant-k-mer
 * bias correction of a cohort's OUT.trkmc.ar count vectors, on one MI355X.
 *
 * Replaces, in /root/reference:  src/pred.cpp:52-82 (main's compute part) and src/pred.h:166-233
 *   load_eachBinGT  pred.h:166-186   counts (u64, one file per sample) -> float matrix
 *   norm_rd         pred.h:204-209   gt(sample, kmer) = count / read depth of the sample
 *   bias_correction pred.h:212-233   per locus: bias(sample) = mean_j gt(sample, ikmer_j) / ikmc_j, divided by its mean
 *                                    over the samples; the locus' k-mer columns are divided by it; Bias(sample, locus) kept
 *   save_matrix     pred.h:236-258   layouts of the three outputs
 *
 * Arithmetic: IEEE float32, as the reference's Eigen::ArrayXXf.  The raw matrix (one conversion and one division per entry)
 * is bit-exact; the sums of a bias are taken in k-mer order per sample (Eigen's scalar order); the mean over the samples is
 * a pairwise tree here and a packet reduction in Eigen, so the corrected matrix and Bias agree to float32 rounding
 * (tests: relative 2e-6), not bit for bit.  PARITY UNPINNED: the reference's pred.cpp needs Eigen, which this image
 * lacks (.gitmodules: the submodule directory is empty), so the oracle (oracle/pred_oracle.py) restates pred.h and could
 * not be checked against a run of the reference.
 *
 * All entry points return dbtk_status_t (dbtk.h); dbtk_last_error() holds the message.  No CPU path: dbtk_pred_create
 * fails with DBTK_ERR_NO_DEVICE without a HIP device.
 */
#ifndef DBTK_PRED_H_
#define DBTK_PRED_H_

#include <stdint.h>

#include "dbtk.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dbtk_pred dbtk_pred_t;

/* The invariant-k-mer metadata of an RPGG build (read_ikmer, pred.h:64-126: `ikmer.meta`): per locus the CUMULATIVE number
 * of k-mers (nk_cum[ntr], nk_cum[ntr-1] == nk) and of invariant k-mers (nik_cum[ntr], nik_cum[ntr-1] == nik); per invariant
 * k-mer its column (iki[nik] < nk) and its expected count (ikmc[nik]).  The matrix G[nk][ns] (float32, 4*nk*ns bytes) lives
 * in HBM. */
dbtk_status_t dbtk_pred_create(int device_id, uint64_t ns, uint64_t nk, uint64_t ntr, const uint32_t* nk_cum, const uint32_t* nik_cum,
                               uint64_t nik, const uint32_t* iki, const uint8_t* ikmc, dbtk_pred_t** out);
void dbtk_pred_free(dbtk_pred_t* p);

/* Reads `ikmer.meta` (little endian: u64 nk, u64 nik, u64 ntr, u32 nk_cum[ntr], u32 nik_cum[ntr], nik x {u32 ki, u8 kc}) and
 * creates the handle for ns samples. */
dbtk_status_t dbtk_pred_create_from_file(int device_id, uint64_t ns, const char* ikmer_meta, dbtk_pred_t** out);
uint64_t dbtk_pred_nk(const dbtk_pred_t* p);
uint64_t dbtk_pred_ntr(const dbtk_pred_t* p);

/* Samples first_sample .. first_sample + n - 1: counts[i * nk + k] = count of k-mer k in sample i (the body of its
 * OUT.trkmc.ar), read_depth[i] its depth.  load_eachBinGT + norm_rd for these columns: G[k][s] = (float)count / depth. */
dbtk_status_t dbtk_pred_load_samples(dbtk_pred_t* p, uint64_t first_sample, uint64_t n, const uint64_t* counts, const float* read_depth);

/* bias_correction (pred.h:212-233) on the loaded matrix, in place; fills the bias matrix.  A locus without k-mers or
 * without invariant k-mers is left alone (the reference `continue`s and leaves its Bias column uninitialised: 0 here). */
dbtk_status_t dbtk_pred_correct(dbtk_pred_t* p);

/* The matrix as save_matrix lays it out (pred.h:236-249): ns x nk, column-major = nk runs of ns floats (before
 * dbtk_pred_correct: the raw genotype matrix, after: the corrected one).  out holds ns * nk floats. */
dbtk_status_t dbtk_pred_matrix(dbtk_pred_t* p, float* out);
/* Bias, ns x ntr column-major (Bias(s, tri) at tri * ns + s). */
dbtk_status_t dbtk_pred_bias(dbtk_pred_t* p, float* out);

/* Kernel times of the last dbtk_pred_correct in milliseconds: bias sums, bias normalisation, the correcting pass. */
dbtk_status_t dbtk_pred_times(dbtk_pred_t* p, float ms[3]);

#ifdef __cplusplus
}
#endif
#endif
