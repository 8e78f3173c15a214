// danbing-tk-pred on the GPU (include/dbtk_pred.h): the cohort's genotype matrix G[nk][ns] (float32) stays in HBM; three
// streaming kernels restate src/pred.h:204-233 of the reference.  HBM-bound float column work: no MFMA, nothing to tile
// but the transposition of the per-sample count vectors into the sample-minor matrix.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/dbtk_pred.h"
#include "dbtk_internal.h"

using namespace dbtk;

#define PCHK(call)                                                                                    \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            set_error(std::string(#call) + ": " + hipGetErrorString(e_));                             \
            return DBTK_ERR_HIP;                                                                      \
        }                                                                                             \
    } while (0)

// ---- load_eachBinGT + norm_rd (pred.h:166-186, 204-209): counts[i][k] (u64, sample-major as the files are) ->
// G[k][first + i] = (float)count / depth[i].  One wave per tile of 64 k-mers x 32 samples: the counts are read along k
// (coalesced), turned in LDS, and written along the samples (runs of 128 bytes).
constexpr int PT_K = 64, PT_S = 32;
__global__ void __launch_bounds__(64) k_pred_load(const uint64_t* __restrict__ counts, const float* __restrict__ depth, float* __restrict__ G,
                                                  uint64_t nk, uint64_t ns, uint64_t first, uint32_t n) {
    __shared__ float tile[PT_K][PT_S + 1];
    const int lane = threadIdx.x;
    const uint64_t k0 = (uint64_t)blockIdx.x * PT_K;
    for (uint32_t i0 = blockIdx.y * PT_S; i0 < n; i0 += gridDim.y * PT_S) {
        const uint32_t ni = n - i0 < (uint32_t)PT_S ? n - i0 : (uint32_t)PT_S;
        for (uint32_t i = 0; i < ni; ++i) {
            const uint64_t k = k0 + lane;
            // uint64 -> float and the division are each one correctly rounded IEEE operation, as Eigen's cast<float>() and operator/
            tile[lane][i] = k < nk ? (float)counts[(uint64_t)(i0 + i) * nk + k] / depth[i0 + i] : 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int r = lane / PT_S; r < PT_K; r += 64 / PT_S) {  // two rows per pass: lanes 0-31 / 32-63 along the samples
            const uint32_t i = lane % PT_S;
            if (k0 + r < nk && i < ni) G[(k0 + r) * ns + first + i0 + i] = tile[r][i];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- bias_correction, first half (pred.h:217-228): B(s, j) = gt(s, iki[j]) / ikmc[j]; bias(s) = B.rowwise().mean(): the sum
// over the locus' invariant k-mers in their order, per sample, then / n.  One lane per sample: row iki[j] of G is read
// coalesced, the adds of a lane are sequential (the order Eigen's scalar reduction takes).
__global__ void __launch_bounds__(64) k_pred_bias(const float* __restrict__ G, const uint32_t* __restrict__ nk_cum, const uint32_t* __restrict__ nik_cum,
                                                  const uint32_t* __restrict__ iki, const float* __restrict__ ikmc, float* __restrict__ bias, uint64_t ns) {
    const uint32_t tri = blockIdx.x;
    const uint64_t s = (uint64_t)blockIdx.y * 64 + threadIdx.x;
    const uint32_t si = tri ? nk_cum[tri - 1] : 0u, ei = nk_cum[tri], isi = tri ? nik_cum[tri - 1] : 0u, iei = nik_cum[tri];
    if (si == ei || isi == iei || s >= ns) return;
    float acc = 0.f;
    for (uint32_t j = isi; j < iei; ++j) acc += G[(uint64_t)iki[j] * ns + s] / ikmc[j];
    bias[(uint64_t)tri * ns + s] = acc / (float)(iei - isi);
}
// second half (pred.h:229-231): bias /= bias.mean() over the samples.  One block per locus; the mean is a pairwise tree.
__global__ void __launch_bounds__(256) k_pred_bias_norm(const uint32_t* __restrict__ nk_cum, const uint32_t* __restrict__ nik_cum, float* __restrict__ bias, uint64_t ns) {
    __shared__ float part[256];
    const uint32_t tri = blockIdx.x;
    const uint32_t si = tri ? nk_cum[tri - 1] : 0u, ei = nk_cum[tri], isi = tri ? nik_cum[tri - 1] : 0u, iei = nik_cum[tri];
    if (si == ei || isi == iei) return;
    float* b = bias + (uint64_t)tri * ns;
    float acc = 0.f;
    for (uint64_t s = threadIdx.x; s < ns; s += 256) acc += b[s];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    const float mean = part[0] / (float)ns;
    for (uint64_t s = threadIdx.x; s < ns; s += 256) b[s] = b[s] / mean;
}
// the correcting pass (pred.h:230): every k-mer column of the locus divided by the locus' bias, sample by sample.  G is
// streamed once, read and written in place: a wave takes PR_ROWS consecutive rows (a row = one k-mer, ns floats) and walks
// the samples 64 at a time, so that PR_ROWS independent loads are in flight per lane; loc[k] = the row's locus (NOLOC: left alone).
constexpr int PR_ROWS = 8;
constexpr uint32_t NOLOC = 0xFFFFFFFFu;
__global__ void __launch_bounds__(256) k_pred_correct(float* __restrict__ G, const uint32_t* __restrict__ loc, const float* __restrict__ bias, uint64_t nk, uint64_t ns) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t r0 = ((uint64_t)blockIdx.x * 4 + w) * PR_ROWS;
    if (r0 >= nk) return;
    uint32_t tri[PR_ROWS];
#pragma unroll
    for (int r = 0; r < PR_ROWS; ++r) tri[r] = r0 + r < nk ? loc[r0 + r] : NOLOC;
    for (uint64_t s = lane; s < ns; s += 64) {
        float g[PR_ROWS], b[PR_ROWS];
#pragma unroll
        for (int r = 0; r < PR_ROWS; ++r) {
            const bool on = tri[r] != NOLOC;
            g[r] = on ? G[(r0 + r) * ns + s] : 0.f;
            b[r] = on ? bias[(uint64_t)tri[r] * ns + s] : 1.f;
        }
#pragma unroll
        for (int r = 0; r < PR_ROWS; ++r) if (tri[r] != NOLOC) G[(r0 + r) * ns + s] = g[r] / b[r];
    }
}

struct dbtk_pred {
    int device = 0;
    uint64_t ns = 0, nk = 0, ntr = 0, nik = 0;
    float* d_G = nullptr;
    float* d_bias = nullptr;
    uint32_t *d_nk = nullptr, *d_nik = nullptr, *d_iki = nullptr, *d_loc = nullptr;  // d_loc[k]: locus of k-mer k, NOLOC where bias_correction skips it
    float* d_ikmc = nullptr;
    uint64_t* d_counts = nullptr; float* d_depth = nullptr; uint64_t stage_cap = 0;  // staging of dbtk_pred_load_samples
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    float ms[3] = {0, 0, 0};
};

extern "C" {

static dbtk_status_t dbtk_pred_create_impl(int device_id, uint64_t ns, uint64_t nk, uint64_t ntr, const uint32_t* nk_cum, const uint32_t* nik_cum,
                               uint64_t nik, const uint32_t* iki, const uint8_t* ikmc, dbtk_pred_t** out) {
    if (!out || !nk_cum || !nik_cum || (nik && (!iki || !ikmc))) { set_error("null argument"); return DBTK_ERR_ARG; }
    *out = nullptr;
    if (!ns || !nk || !ntr) { set_error("empty cohort / RPGG"); return DBTK_ERR_ARG; }
    if (nk > 0xFFFFFFFFull || ntr > 0xFFFFFFFFull) { set_error("ikmer.meta holds 32-bit k-mer indices"); return DBTK_ERR_ARG; }
    for (uint64_t t = 0; t < ntr; ++t) {
        const uint32_t a = t ? nk_cum[t - 1] : 0u, b = nk_cum[t], c = t ? nik_cum[t - 1] : 0u, d = nik_cum[t];
        if (b < a || b > nk || d < c || d > nik) { set_error("ikmer.meta: the cumulative counts must not decrease or pass the totals"); return DBTK_ERR_FORMAT; }
    }
    for (uint64_t j = 0; j < nik; ++j) if (iki[j] >= nk) { set_error("ikmer.meta: invariant k-mer index out of range"); return DBTK_ERR_FORMAT; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device (the library has no CPU path)"); return DBTK_ERR_NO_DEVICE; }
    if (device_id < 0 || device_id >= ndev) { set_error("device_id out of range"); return DBTK_ERR_ARG; }
    PCHK(hipSetDevice(device_id));
    std::vector<float> kc(nik);
    for (uint64_t j = 0; j < nik; ++j) kc[j] = (float)ikmc[j];
    std::vector<uint32_t> loc(nk, NOLOC);  // (k-mers past the last locus' cumulative count belong to no locus, like in the reference's loop)
    for (uint64_t t = 0; t < ntr; ++t) {
        const uint32_t a = t ? nk_cum[t - 1] : 0u, b = nk_cum[t], c = t ? nik_cum[t - 1] : 0u, d = nik_cum[t];
        if (a == b || c == d) continue;
        for (uint32_t k = a; k < b; ++k) loc[k] = (uint32_t)t;
    }
    dbtk_pred* p = new dbtk_pred;  // (after the host-side vectors: nothing below throws)
    p->device = device_id; p->ns = ns; p->nk = nk; p->ntr = ntr; p->nik = nik;
    dbtk_status_t st = DBTK_OK;
    auto fail = [&](hipError_t e, const char* what) { set_error(std::string(what) + ": " + hipGetErrorString(e)); st = DBTK_ERR_HIP; };
    hipError_t e;
    if ((e = hipStreamCreate(&p->stream)) != hipSuccess) fail(e, "hipStreamCreate");
    for (int i = 0; i < 4 && !st; ++i) if ((e = hipEventCreate(&p->ev[i])) != hipSuccess) fail(e, "hipEventCreate");
    if (!st && (e = hipMalloc(&p->d_G, nk * ns * sizeof(float))) != hipSuccess) fail(e, "hipMalloc (genotype matrix)");
    if (!st && (e = hipMalloc(&p->d_bias, ntr * ns * sizeof(float))) != hipSuccess) fail(e, "hipMalloc (bias matrix)");
    if (!st && (e = hipMalloc(&p->d_nk, ntr * 4)) != hipSuccess) fail(e, "hipMalloc");
    if (!st && (e = hipMalloc(&p->d_nik, ntr * 4)) != hipSuccess) fail(e, "hipMalloc");
    if (!st && (e = hipMalloc(&p->d_iki, (nik + 1) * 4)) != hipSuccess) fail(e, "hipMalloc");
    if (!st && (e = hipMalloc(&p->d_loc, nk * 4)) != hipSuccess) fail(e, "hipMalloc");
    if (!st && (e = hipMemcpyAsync(p->d_loc, loc.data(), nk * 4, hipMemcpyHostToDevice, p->stream)) != hipSuccess) fail(e, "hipMemcpy");
    if (!st && (e = hipMalloc(&p->d_ikmc, (nik + 1) * 4)) != hipSuccess) fail(e, "hipMalloc");
    if (!st && (e = hipMemsetAsync(p->d_G, 0, nk * ns * sizeof(float), p->stream)) != hipSuccess) fail(e, "hipMemset");
    if (!st && (e = hipMemsetAsync(p->d_bias, 0, ntr * ns * sizeof(float), p->stream)) != hipSuccess) fail(e, "hipMemset");
    if (!st && (e = hipMemcpyAsync(p->d_nk, nk_cum, ntr * 4, hipMemcpyHostToDevice, p->stream)) != hipSuccess) fail(e, "hipMemcpy");
    if (!st && (e = hipMemcpyAsync(p->d_nik, nik_cum, ntr * 4, hipMemcpyHostToDevice, p->stream)) != hipSuccess) fail(e, "hipMemcpy");
    if (!st && nik && (e = hipMemcpyAsync(p->d_iki, iki, nik * 4, hipMemcpyHostToDevice, p->stream)) != hipSuccess) fail(e, "hipMemcpy");
    if (!st && nik && (e = hipMemcpyAsync(p->d_ikmc, kc.data(), nik * 4, hipMemcpyHostToDevice, p->stream)) != hipSuccess) fail(e, "hipMemcpy");
    if (!st && (e = hipStreamSynchronize(p->stream)) != hipSuccess) fail(e, "hipStreamSynchronize");
    if (st) { dbtk_pred_free(p); return st; }
    *out = p;
    return DBTK_OK;
}

void dbtk_pred_free(dbtk_pred_t* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    void* ptrs[] = {p->d_G, p->d_bias, p->d_nk, p->d_nik, p->d_iki, p->d_loc, p->d_ikmc, p->d_counts, p->d_depth};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    for (auto& e : p->ev) if (e) (void)hipEventDestroy(e);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

static dbtk_status_t dbtk_pred_create_from_file_impl(int device_id, uint64_t ns, const char* ikmer_meta, dbtk_pred_t** out) {
    if (!ikmer_meta || !out) { set_error("null argument"); return DBTK_ERR_ARG; }
    *out = nullptr;
    FILE* f = fopen(ikmer_meta, "rb");
    if (!f) { set_error(std::string("cannot open ") + ikmer_meta); return DBTK_ERR_IO; }
    uint64_t hdr[3];
    dbtk_status_t st = DBTK_OK;
    std::vector<uint32_t> nkc, nikc, iki;
    std::vector<uint8_t> kc;
    if (fread(hdr, 8, 3, f) != 3) { set_error(std::string("truncated ") + ikmer_meta); st = DBTK_ERR_IO; }
    if (!st && (hdr[0] > 0xFFFFFFFFull || hdr[1] > hdr[0] || hdr[2] > 0xFFFFFFFFull)) { set_error(std::string(ikmer_meta) + ": implausible header"); st = DBTK_ERR_FORMAT; }
    if (!st) {
        const uint64_t nik = hdr[1], ntr = hdr[2];
        nkc.resize(ntr); nikc.resize(ntr); iki.resize(nik); kc.resize(nik);
        std::vector<uint8_t> rec(nik * 5);
        if (fread(nkc.data(), 4, ntr, f) != ntr || fread(nikc.data(), 4, ntr, f) != ntr || (nik && fread(rec.data(), 5, nik, f) != nik)) {
            set_error(std::string("truncated ") + ikmer_meta); st = DBTK_ERR_IO;
        }
        for (uint64_t j = 0; j < nik && !st; ++j) { memcpy(&iki[j], &rec[5 * j], 4); kc[j] = rec[5 * j + 4]; }
    }
    fclose(f);
    if (st) return st;
    return dbtk_pred_create(device_id, ns, hdr[0], hdr[2], nkc.data(), nikc.data(), hdr[1], iki.data(), kc.data(), out);
}

uint64_t dbtk_pred_nk(const dbtk_pred_t* p) { return p ? p->nk : 0; }
uint64_t dbtk_pred_ntr(const dbtk_pred_t* p) { return p ? p->ntr : 0; }

static dbtk_status_t dbtk_pred_load_samples_impl(dbtk_pred_t* p, uint64_t first_sample, uint64_t n, const uint64_t* counts, const float* read_depth) {
    if (!p || !counts || !read_depth) { set_error("null argument"); return DBTK_ERR_ARG; }
    if (first_sample + n > p->ns || n > 0xFFFFFFFFull) { set_error("sample range outside the cohort"); return DBTK_ERR_ARG; }
    if (!n) return DBTK_OK;
    PCHK(hipSetDevice(p->device));
    if (n > p->stage_cap) {
        if (p->d_counts) PCHK(hipFree(p->d_counts));
        if (p->d_depth) PCHK(hipFree(p->d_depth));
        p->d_counts = nullptr; p->d_depth = nullptr; p->stage_cap = 0;
        PCHK(hipMalloc(&p->d_counts, n * p->nk * 8));
        PCHK(hipMalloc(&p->d_depth, n * 4));
        p->stage_cap = n;
    }
    PCHK(hipMemcpyAsync(p->d_counts, counts, n * p->nk * 8, hipMemcpyHostToDevice, p->stream));
    PCHK(hipMemcpyAsync(p->d_depth, read_depth, n * 4, hipMemcpyHostToDevice, p->stream));
    const uint64_t kt = (p->nk + PT_K - 1) / PT_K;
    if (kt > 0x7FFFFFFFull) { set_error("too many k-mers for one launch"); return DBTK_ERR_ARG; }
    const uint32_t gy = (uint32_t)std::min<uint64_t>((n + PT_S - 1) / PT_S, 64);
    hipLaunchKernelGGL(k_pred_load, dim3((uint32_t)kt, gy), dim3(64), 0, p->stream, p->d_counts, p->d_depth, p->d_G, p->nk, p->ns, first_sample, (uint32_t)n);
    PCHK(hipGetLastError());
    PCHK(hipStreamSynchronize(p->stream));  // (the caller's buffers are free again)
    return DBTK_OK;
}

dbtk_status_t dbtk_pred_correct(dbtk_pred_t* p) {
    if (!p) { set_error("null argument"); return DBTK_ERR_ARG; }
    PCHK(hipSetDevice(p->device));
    hipStream_t s = p->stream;
    PCHK(hipMemsetAsync(p->d_bias, 0, p->ntr * p->ns * sizeof(float), s));
    PCHK(hipEventRecord(p->ev[0], s));
    hipLaunchKernelGGL(k_pred_bias, dim3((uint32_t)p->ntr, (uint32_t)((p->ns + 63) / 64)), dim3(64), 0, s, p->d_G, p->d_nk, p->d_nik, p->d_iki, p->d_ikmc, p->d_bias, p->ns);
    PCHK(hipGetLastError());
    PCHK(hipEventRecord(p->ev[1], s));
    hipLaunchKernelGGL(k_pred_bias_norm, dim3((uint32_t)p->ntr), dim3(256), 0, s, p->d_nk, p->d_nik, p->d_bias, p->ns);
    PCHK(hipGetLastError());
    PCHK(hipEventRecord(p->ev[2], s));
    const uint64_t nb = (p->nk + 4 * PR_ROWS - 1) / (4 * PR_ROWS);
    hipLaunchKernelGGL(k_pred_correct, dim3((uint32_t)nb), dim3(256), 0, s, p->d_G, p->d_loc, p->d_bias, p->nk, p->ns);
    PCHK(hipGetLastError());
    PCHK(hipEventRecord(p->ev[3], s));
    PCHK(hipStreamSynchronize(s));
    for (int i = 0; i < 3; ++i) PCHK(hipEventElapsedTime(&p->ms[i], p->ev[i], p->ev[i + 1]));
    return DBTK_OK;
}

dbtk_status_t dbtk_pred_matrix(dbtk_pred_t* p, float* out) {
    if (!p || !out) { set_error("null argument"); return DBTK_ERR_ARG; }
    PCHK(hipSetDevice(p->device));
    PCHK(hipMemcpy(out, p->d_G, p->nk * p->ns * sizeof(float), hipMemcpyDeviceToHost));
    return DBTK_OK;
}
dbtk_status_t dbtk_pred_bias(dbtk_pred_t* p, float* out) {
    if (!p || !out) { set_error("null argument"); return DBTK_ERR_ARG; }
    PCHK(hipSetDevice(p->device));
    PCHK(hipMemcpy(out, p->d_bias, p->ntr * p->ns * sizeof(float), hipMemcpyDeviceToHost));
    return DBTK_OK;
}
dbtk_status_t dbtk_pred_times(dbtk_pred_t* p, float ms[3]) {
    if (!p || !ms) { set_error("null argument"); return DBTK_ERR_ARG; }
    for (int i = 0; i < 3; ++i) ms[i] = p->ms[i];
    return DBTK_OK;
}

// ---- the entry points above that parse files or allocate host memory, behind the exception barrier (dbtk_internal.h: guarded)
dbtk_status_t dbtk_pred_create(int device_id, uint64_t ns, uint64_t nk, uint64_t ntr, const uint32_t* nk_cum, const uint32_t* nik_cum,
                               uint64_t nik, const uint32_t* iki, const uint8_t* ikmc, dbtk_pred_t** out) {
    return dbtk::guarded([&] { return dbtk_pred_create_impl(device_id, ns, nk, ntr, nk_cum, nik_cum, nik, iki, ikmc, out); });
}
dbtk_status_t dbtk_pred_create_from_file(int device_id, uint64_t ns, const char* ikmer_meta, dbtk_pred_t** out) {
    return dbtk::guarded([&] { return dbtk_pred_create_from_file_impl(device_id, ns, ikmer_meta, out); });
}
dbtk_status_t dbtk_pred_load_samples(dbtk_pred_t* p, uint64_t first_sample, uint64_t n, const uint64_t* counts, const float* read_depth) {
    return dbtk::guarded([&] { return dbtk_pred_load_samples_impl(p, first_sample, n, counts, read_depth); });
}

}  // extern "C"
