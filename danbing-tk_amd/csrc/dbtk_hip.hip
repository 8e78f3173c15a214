// dbtk_hip.hip — gfx950 kernels + per-GPU context of the align hot path.
//
// The kernel bodies live in dbtk_kernels.h; this file binds them to the
// hardware (DevX: wave64 ballots, DPP scans and quad permutes, wavefront-scope
// fences, global + LDS atomics), owns the HBM-resident tables and accumulators, and implements the
// device half of include/dbtk.h.  There is no host execution path: without a
// HIP device dbtk_ctx_create fails with DBTK_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and enum values only: the library itself is dlopen()ed by dbtk_allreduce
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <deque>
#include <map>
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "dbtk_devx.h"
#include "dbtk_internal.h"
#include "dbtk_kernels.h"
#include "dbtk_ingest.h"
#include "dbtk_gz.h"

using namespace dbtk;

// DBTK_VERBOSE >= 2: every device / pinned allocation or release that takes longer than 0.2 ms says so (which line, how many bytes)
static const bool g_alloc_trace = getenv("DBTK_VERBOSE") && atoi(getenv("DBTK_VERBOSE")) >= 2;
template <class F>
static inline hipError_t alloc_traced(const char* what, size_t bytes, int line, F f) {
    if (!g_alloc_trace) return f();
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t r = f();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ms > 0.2) fprintf(stderr, "alloc: %s of %zu bytes at line %d: %.2f ms\n", what, bytes, line, ms);
    return r;
}
#define hipMalloc(p, n) alloc_traced("hipMalloc", (size_t)(n), __LINE__, [&] { return (hipMalloc)((void**)(p), (size_t)(n)); })
#define hipHostMalloc(p, n, f) alloc_traced("hipHostMalloc", (size_t)(n), __LINE__, [&] { return (hipHostMalloc)((void**)(p), (size_t)(n), (f)); })
#define hipFree(p) alloc_traced("hipFree", 0, __LINE__, [&] { return (hipFree)((void*)(p)); })
#define hipHostFree(p) alloc_traced("hipHostFree", 0, __LINE__, [&] { return (hipHostFree)((void*)(p)); })

// --------------------------------------------------------------- kernels ---
__global__ void __launch_bounds__(256) k_fill_idx(IdxBucket* b, uint64_t nslots) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += (uint64_t)gridDim.x * blockDim.x) {
        b[i >> 2].key[i & 3] = NAN64;
        b[i >> 2].val[i & 3] = 0;
    }
}
__global__ void __launch_bounds__(256) k_idx_insert(IdxBuildArgs a) { DevX x{nullptr}; body_idx_insert(x, a); }
__global__ void __launch_bounds__(256) k_flt_insert(FltBuildArgs a) { DevX x{nullptr}; body_flt_insert(x, a); }
__global__ void __launch_bounds__(256) k_idx_finalize(IdxBucket* b, uint64_t nslots) { DevX x{nullptr}; body_idx_finalize(x, b, nslots); }
__global__ void __launch_bounds__(256) k_cls_insert(ClsBuildArgs a) { DevX x{nullptr}; body_cls_insert(x, a); }
__global__ void __launch_bounds__(256) k_idx_aux(IdxAuxArgs a) { DevX x{nullptr}; body_idx_aux(x, a); }

// dst[i] += src[i] (dbtk_allreduce: contexts that share a device)
__global__ void __launch_bounds__(256) k_accum_add(uint64_t* dst, const uint64_t* src, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] += src[i];
}
// counters[c] += sum of its replicas; replicas back to zero
// (words DBTK_C_COUNT .. of a replica row: path statistics of the lean probe kernel, dbtk_probe2.h: P2_REP_*)
__global__ void __launch_bounds__(64) k_fold_counters(uint64_t* counters, uint64_t* rep, uint64_t* pstats) {
    const uint32_t c = threadIdx.x;
    static_assert(P2_REP_SHARED < CTR_STRIDE && P2_REP_DONE >= DBTK_C_COUNT, "spare words of a replica row");
    if (c >= DBTK_C_COUNT) {
        const uint32_t to = c == P2_REP_DONE ? 20u : c == P2_REP_CLS ? 16u : c == P2_REP_INC ? 17u : c == P2_REP_SHARED ? 19u : 0xFFFFFFFFu;
        if (to == 0xFFFFFFFFu || !pstats) return;
        uint64_t s = 0;
        for (uint32_t r = 0; r < CTR_REP; ++r) s += atomicExch(reinterpret_cast<unsigned long long*>(&rep[(size_t)r * CTR_STRIDE + c]), 0ull);
        pstats[to] += s;
        return;
    }
    uint64_t s = 0;
    for (uint32_t r = 0; r < CTR_REP; ++r) s += atomicExch(reinterpret_cast<unsigned long long*>(&rep[(size_t)r * CTR_STRIDE + c]), 0ull);
    if (s) atomicAdd(reinterpret_cast<unsigned long long*>(&counters[c]), (unsigned long long)s);  // (the other lane may be folding too)
}
__global__ void __launch_bounds__(K1_NT, 4) k_encode_subfilter(BatchArgs a) {
    __shared__ __attribute__((aligned(16))) K1Smem sm;
    DevX x{&sm};
    body_encode_subfilter(x, a);
}
// (the form for a batch that hits: a mate's first sample alone before the other three, body_encode_subfilter<true>)
__global__ void __launch_bounds__(K1_NT, 4) k_encode_subfilter_lazy(BatchArgs a) {
    __shared__ __attribute__((aligned(16))) K1Smem sm;
    DevX x{&sm};
    body_encode_subfilter<true>(x, a);
}
// K2 / K3 are instantiated per NS = 64-position slots a read needs (2: <= 128 positions, 3: 150 bp reads, 4: up to 256 bp)
#ifndef DBTK_K2_WPE
#define DBTK_K2_WPE 4  // waves per SIMD the general probe kernel's registers are budgeted for
#endif
template <int NS> __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DBTK_K2_WPE, 8))) k_probe_general(BatchArgs a) {
    __shared__ __attribute__((aligned(16))) ProbeSmem sm;
    DevX x{&sm};
    body_probe<NS>(x, a);
}
// the lean probe kernel (dbtk_probe2.h): NPL positions per lane, windows of WN m-mers
#ifndef DBTK_P2_WPE
#define DBTK_P2_WPE 4  // waves per SIMD its registers are budgeted for (its LDS allows 16 waves per CU)
#endif
template <int NPL, int WN, bool SEL = false, bool FUSE = false> __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DBTK_P2_WPE, 8))) k_probe(BatchArgs a) {
    __shared__ Probe2SmemT<NPL> sm;
    DevX x{&sm};
    body_probe2<NPL, WN, SEL, FUSE>(x, a);
}
template <int NS, bool RECS, bool SEL = false> __global__ void __launch_bounds__(64) k_pair_usual(BatchArgs a) {
    __shared__ UsualSmem sm;
    DevX x{&sm};
    body_pair_usual<NS, RECS, SEL>(x, a);
}
template <int NS, bool RECS> __global__ void __launch_bounds__(64, NS <= 3 ? 2 : 1) k_pair(BatchArgs a) {
    __shared__ __attribute__((aligned(16))) PairSmemT<NS> sm;
    DevX x{&sm};
    body_pair<NS, RECS>(x, a);
}

// per-locus images of the index and the probe kernel that keeps one in LDS (dbtk_locus.h)
__global__ void __launch_bounds__(256) k_loc_count(LocBuildArgs a) { DevX x{nullptr}; body_loc_count(x, a); }
__global__ void __launch_bounds__(256) k_loc_scatter(LocBuildArgs a) { DevX x{nullptr}; body_loc_scatter(x, a); }
__global__ void __launch_bounds__(64) k_loc_place(LocBuildArgs a) { DevX x{nullptr}; body_loc_place(x, a); }  // (DBTK_LOC_PLACE_THREAD=1: the thread-per-locus form)
template <int LGMAX, int LGLOW> __global__ void __launch_bounds__(64) k_loc_place_wave(LocBuildArgs a) {
    __shared__ LocPlaceSmemT<LGMAX> sm;
    DevX x{&sm};
    body_loc_place_wave<LGMAX, LGLOW>(x, a);
}

__global__ void __launch_bounds__(256) k_loc_verify(LocBuildArgs a) { DevX x{nullptr}; body_loc_verify(x, a); }
// offsets of a parsed block appended behind the merged batch's: dst[r] = src[r] + add, r = 0 .. n - 1
__global__ void __launch_bounds__(256) k_off_rebase(uint64_t* dst, const uint64_t* src, uint64_t n, uint64_t add) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[i] + add;
}
__global__ void __launch_bounds__(256) k_csum(const uint64_t* p, uint64_t n, uint64_t* out) { DevX x{nullptr}; body_csum(x, p, n, out); }
__global__ void __launch_bounds__(256) k_gloc_count(LocBuildArgs a) { DevX x{nullptr}; body_gloc_count(x, a); }
__global__ void __launch_bounds__(256) k_gloc_scatter(LocBuildArgs a) { DevX x{nullptr}; body_gloc_scatter(x, a); }
__global__ void __launch_bounds__(256) k_loc_items(LocItemArgs a) { DevX x{nullptr}; body_loc_items(x, a); }
__global__ void __launch_bounds__(1024) k_loc_split(LocSplitArgs a) { __shared__ LocSplitSmem sm; DevX x{&sm}; body_loc_split(x, a); }
__global__ void __launch_bounds__(256) k_loc_rest(LocItemArgs a) { DevX x{nullptr}; body_loc_rest(x, a); }
// three classes of workgroup by the size of the image (loc_image_bytes of 512, 1024, 2048 buckets): the smaller the image, the fewer
// waves share it and the more workgroups a CU holds (4 x 4, 2 x 8, 1 x 16 waves)
constexpr int LOC_IMGB_XS = LOC_HDR + (32 << 9) + (1 << 9), LOC_IMGB_S = LOC_HDR + (32 << 10) + (1 << 10), LOC_IMGB_L = LOC_HDR + (32 << LOC_LG_MAX) + (1 << LOC_LG_MAX);
constexpr int LOC_NW_XS = 4, LOC_NW_S = 8, LOC_NW_L = 16;
// FUSE: the kernel also resolves the usual pairs of its items (dbtk_locus.h)
template <int NPL, int NW, int IMGB, bool FUSE = false> __global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4, 8))) k_probe_locus(BatchArgs a, LocRunArgs r) {
    __shared__ LocSmemT<NPL, NW, IMGB, FUSE> sm;
    DevX x{&sm};
    body_probe_locus<NPL, NW, IMGB, FUSE>(x, a, r);
}
__global__ void __launch_bounds__(256) k_mz_insert(MzBuildArgs a) { DevX x{nullptr}; body_mz_insert(x, a); }
__global__ void __launch_bounds__(256) k_mz_fill(uint64_t* t, uint64_t nwords, int level1) { DevX x{nullptr}; body_mz_fill(x, t, nwords, level1); }
__global__ void __launch_bounds__(256) k_surv_key(SurvSortArgs a) { DevX x{nullptr}; body_surv_key(x, a); }
__global__ void __launch_bounds__(64) k_surv_scan(SurvSortArgs a, int step) { DevX x{nullptr}; body_surv_scan(x, a, step); }
__global__ void __launch_bounds__(256) k_surv_scatter(SurvSortArgs a) { DevX x{nullptr}; body_surv_scatter(x, a); }
__global__ void __launch_bounds__(256) k_gr_insert(GrBuildArgs a) { DevX x{nullptr}; body_gr_insert(x, a); }
// the reader on the device (dbtk_ingest.h): record boundaries, mate pairing and the batch arrays of a block of raw bytes
__global__ void __launch_bounds__(64) k_ing_count(IngestArgs a) { DevX x{nullptr}; body_ing_count(x, a); }
__global__ void __launch_bounds__(64) k_ing_scan(IngestArgs a, int step) { DevX x{nullptr}; body_ing_scan(x, a, step); }
__global__ void __launch_bounds__(64) k_ing_lines(IngestArgs a) { DevX x{nullptr}; body_ing_lines(x, a); }
__global__ void __launch_bounds__(256) k_ing_pairs(IngestArgs a) { DevX x{nullptr}; body_ing_pairs(x, a); }
__global__ void __launch_bounds__(64) k_ing_place(IngestArgs a, int step) { DevX x{nullptr}; body_ing_place(x, a, step); }
__global__ void __launch_bounds__(64) k_ing_gather(IngestArgs a) { DevX x{nullptr}; body_ing_gather(x, a); }
__global__ void __launch_bounds__(256) k_ing_carry(IngestArgs a) { DevX x{nullptr}; body_ing_carry(x, a); }
// the -a / -ae writer on the device (dbtk_gz.h): the lines of a parsed and walked block, and their gzip members
__global__ void __launch_bounds__(256) k_aln_len(AlnLineArgs a) { DevX x{nullptr}; body_aln_len(x, a); }
__global__ void __launch_bounds__(64) k_aln_scan(AlnLineArgs a, int step) { DevX x{nullptr}; body_aln_scan(x, a, step); }
__global__ void __launch_bounds__(64) k_aln_write(AlnLineArgs a) { DevX x{nullptr}; body_aln_write(x, a); }
__global__ void __launch_bounds__(64) k_gz_member(GzArgs a) {
    __shared__ GzSmem sm;
    DevX x{&sm};
    body_gz_member(x, a);
}
__global__ void __launch_bounds__(64) k_gz_scan(GzArgs a, int step) { DevX x{nullptr}; body_gz_scan(x, a, step); }
__global__ void __launch_bounds__(64) k_gz_pack(GzArgs a) { DevX x{nullptr}; body_gz_pack(x, a); }
// the graph walk (dbtk_walk.h): one wave per read (function mode) / per pair (the hot path with threading = 2)
#ifndef DBTK_WALK_WAVES
#define DBTK_WALK_WAVES 2  // waves per SIMD the pair kernel is compiled for (register budget 512 / this): 3, 4 and 5 measured no faster
#endif
__global__ void __launch_bounds__(64) k_walk_reads(WalkArgs a) {
    __shared__ __attribute__((aligned(16))) WalkReadSmem sm;
    DevX x{&sm};
    body_walk_reads(x, a);
}
// pair mode, first kernel (dbtk_walk.h: body_walk_fast): decides and counts the pairs one of whose mates threads cleanly
#ifndef DBTK_WF_WPE
#define DBTK_WF_WPE 4
#endif
template <int NPL, int WN> __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DBTK_WF_WPE, 8))) k_walk_fast(WalkArgs a) {
    __shared__ WalkFastSmemT<NPL> sm;
    DevX x{&sm};
    body_walk_fast<NPL, WN>(x, a);
}
__global__ void __launch_bounds__(256) k_grmz_insert(GrMzBuildArgs a) { DevX x{nullptr}; body_grmz_insert(x, a); }
// ... and its form with the locus' graph image in LDS (dbtk_walkfast.h: body_walk_fast_locus; classes of workgroup as k_probe_locus)
template <int NPL, int NW, int IMGB> __global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4, 8))) k_walk_fast_locus(WalkArgs a, LocRunArgs r) {
    __shared__ WalkFastLocSmemT<NPL, NW, IMGB> sm;
    DevX x{&sm};
    body_walk_fast_locus<NPL, NW, IMGB>(x, a, r);
}
// the error-correcting walk for the pairs the lean kernel's locus-resident form could not decide: same items, the graph image in LDS again
// (dbtk_walkfast.h: body_walk_pairs_locus); two waves per workgroup, each with its two WalkSmem
constexpr int WPL_NW = 2;
template <int IMGB> __global__ void __launch_bounds__(WPL_NW * 64) k_walk_pairs_locus(WalkArgs a, LocRunArgs r) {
    __shared__ WalkPairsLocSmemT<WPL_NW, IMGB> sm;
    DevX x{&sm};
    body_walk_pairs_locus<WPL_NW, IMGB>(x, a, r);
}
__global__ void __launch_bounds__(64, DBTK_WALK_WAVES) k_walk_pairs(WalkArgs a) {
    __shared__ __attribute__((aligned(16))) WalkPairSmem sm;  // one set of arrays per mate + the tables / parameters
    DevX x{&sm};
    body_walk_pairs(x, a);
}

// ---------------------------------------------------------------- context --
// Every kernel launch is checked: a launch-configuration error at once (hipGetLastError is a host-side query) and — in a
// debug build (-DDBTK_DEBUG_LAUNCH) or with DBTK_SYNC_LAUNCHES=1 in the environment — the kernel's own faults too, by waiting
// for it, so that an error carries the name of the kernel that raised it instead of surfacing at the next synchronisation.
#ifndef DBTK_DEBUG_LAUNCH
#define DBTK_DEBUG_LAUNCH 0
#endif
static bool sync_launches() {
    static const bool v = [] { const char* e = getenv("DBTK_SYNC_LAUNCHES"); return (e && atoi(e) != 0) || DBTK_DEBUG_LAUNCH; }();
    return v;
}
// DBTK_TRACE_LAUNCHES=1 (diagnostic): every launch is announced on stderr, waited for, and acknowledged — the last name before a GPU
// memory fault is the kernel that raised it.  DBTK_SYNC_ONLY=<substring>: only the kernels whose name contains it are waited for.
static int trace_launches() {
    static const int v = [] { const char* e = getenv("DBTK_TRACE_LAUNCHES"); return e ? atoi(e) : 0; }();
    return v;
}
static bool sync_only(const char* name) {
    static const char* pat = getenv("DBTK_SYNC_ONLY");
    return pat && *pat && strstr(name, pat) != nullptr;
}
#define LAUNCH(kern, grid, block, stream, ...)                                                        \
    do {                                                                                              \
        if (trace_launches()) fprintf(stderr, "[launch] %s grid %u block %u\n", #kern, (unsigned)dim3(grid).x, (unsigned)dim3(block).x); \
        hipLaunchKernelGGL(kern, grid, block, 0, stream, __VA_ARGS__);                                \
        hipError_t e_ = hipGetLastError();                                                            \
        if (e_ == hipSuccess && (sync_launches() || trace_launches() || sync_only(#kern))) e_ = hipStreamSynchronize(stream); \
        if (trace_launches()) fprintf(stderr, "[launch] %s done: %s\n", #kern, hipGetErrorString(e_)); \
        if (e_ != hipSuccess) {                                                                       \
            set_error(std::string("kernel " #kern ": ") + hipGetErrorString(e_));                     \
            return DBTK_ERR_HIP;                                                                      \
        }                                                                                             \
    } while (0)

#define HIPCHK(call)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            set_error(std::string(#call) + ": " + hipGetErrorString(e_));                             \
            return DBTK_ERR_HIP;                                                                      \
        }                                                                                             \
    } while (0)

namespace {
// Per-kernel timing: a pool of HIP event pairs recorded on the context's stream
// around every launch; folded into (total ms, launches) when the pool fills or
// when the caller asks.
constexpr int NKERN = 6;       // k_encode_subfilter, k_probe, k_pair_usual, k_pair, k_walk_pairs, k_surv_sort (its three kernels)
constexpr int EVPOOL = 128;    // launches in flight before a fold
struct Timed {
    const char* name;
    hipEvent_t beg[EVPOOL], end[EVPOOL];
    int used;
    double total_ms;
    uint64_t launches;
};
}  // namespace


// The HBM tables of an RPGG on a device, shared by every context created for that (handle, device): the index and its
// minimizer-grouped copy, the presence filter, the class table, vv, ... are built by the first context and freed with the
// last one (two contexts on one GPU used to hold two copies: 26 GB each at release scale).  The optional tables (graph,
// TR edges, bait) are added by the first context that needs them.
struct TableShare {
    int refs = 0;
    IdxBucket* d_idx = nullptr; uint64_t* d_flt = nullptr; uint64_t flt_words = 0; uint32_t* d_trbeg = nullptr; ClsSlot* d_cls = nullptr;
    MzBucket* d_mz = nullptr; MzSlot* d_ovf = nullptr; GrSlot* d_gr = nullptr; MzBucket* d_grmz = nullptr; uint32_t* d_vv = nullptr; uint8_t* d_qc = nullptr; uint16_t* d_perm = nullptr;
    ClsSlot* d_tre = nullptr; ClsSlot* d_bait = nullptr;
    LocusDir* d_ldir = nullptr; uint8_t* d_limg = nullptr; LocusDir* d_gldir = nullptr; uint8_t* d_glimg = nullptr;
    uint64_t bytes[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // HBM bytes per table (dbtk_ctx_table_bytes)
    DevTables T;
    uint32_t consistent = 0;
};

constexpr uint32_t TK_INLINE = 8;  // words in front of d_small (same allocation): the chunk counters of a batch of up to three chunks
struct dbtk_ctx {
    TableShare* share = nullptr;
    const dbtk_rpgg* g = nullptr;
    uint64_t g_uid = 0;           // g->uid, kept apart: releasing the tables must not touch a handle the caller may have freed first
    dbtk_params_t P;
    int device = 0;
    hipStream_t stream = nullptr;
    DevTables T;
    // device allocations
    IdxBucket* d_idx = nullptr;
    uint64_t* d_flt = nullptr; uint64_t flt_words = 0;
    uint32_t* d_trbeg = nullptr;
    ClsSlot* d_cls = nullptr;
    MzBucket* d_mz = nullptr;     // the probe kernel's minimizer-grouped copy of the index (level 1)
    MzSlot* d_ovf = nullptr;      //   ... and its overflow table (level 2)
    GrSlot* d_gr = nullptr;       // graph table (threading = 2), nullptr when the handle holds no graph
    MzBucket* d_grmz = nullptr;   //   ... and its minimizer-grouped copy (the lean walk kernel)
    LocusDir* d_ldir = nullptr;   // per-locus images of the index (dbtk_locus.h): directory,
    uint8_t* d_limg = nullptr;    //   ... and the images
    LocusDir* d_gldir = nullptr;  // per-locus images of the graph table (walking contexts)
    uint8_t* d_glimg = nullptr;
    uint64_t glimg_bytes = 0;
    int wfl_blocks[6] = {0, 0, 0, 0, 0, 0};  // workgroups of k_walk_fast_locus<3 | 5, class 0 | 1 | 2>
    uint64_t limg_bytes = 0, loc_nimg = 0, loc_left_out = 0;
    bool loc_from_cache = false;
    uint64_t tb_idx = 0, tb_flt = 0, tb_cls = 0, tb_mz = 0, tb_ovf = 0, tb_gr = 0, tb_grmz = 0;  // bytes of the tables this context built
    int loc_blocks[6] = {0, 0, 0, 0, 0, 0};  // workgroups of k_probe_locus<3 | 5, class 0 | 1 | 2>
    int locf_blocks[6] = {0, 0, 0, 0, 0, 0}; // ... of its fused form (more LDS per workgroup)
    uint32_t* h_sortflag = nullptr;  // pinned: survivors [0] and sort flag [6] of the batch before (a hint: see launch_batch)
    uint32_t* d_vv = nullptr;
    uint8_t* d_qc = nullptr;
    uint16_t* d_perm = nullptr;
    uint64_t* d_accum = nullptr;  // counts | kmc | nmapread | counters
    uint64_t* d_ctr = nullptr;    // counter replicas (folded into d_accum's counters before anything reads them: sync_all)
    bool fold_pending = false;    // batches have been launched since the replicas were last folded
    uint8_t* m_flat = nullptr; uint64_t m_flat_cap = 0;   // the current lane's merged blocks (dbtk_ingest_align_merged; Lane::m_*)
    uint64_t* m_off = nullptr; uint64_t m_off_cap = 0;
    uint64_t m_bytes = 0, m_pairs = 0; uint32_t m_maxlen = 0;
    uint64_t* d_pstats = nullptr; // path statistics (dbtk.h: dbtk_ctx_path_stats): which kernels took how many pairs; never part of the results
    uint64_t n_accum = 0, ntr = 0;
    uint32_t* d_small = nullptr;  // nsurv, novf, nrec, errflag
    uint32_t* d_surv = nullptr; uint64_t surv_cap = 0;  // the encode stage's survivor list | the list in locus order | keys | histogram
    uint32_t* d_sorted = nullptr;                       // (inside d_surv) the sorted list of the last batch: what K2, K3 and the walk index
    uint8_t* d_seq = nullptr; uint64_t seq_cap = 0;
    uint64_t* d_off = nullptr; uint64_t off_cap = 0;
    dbtk_pair_rec_t* d_recs = nullptr; uint64_t rec_cap = 0;
    uint64_t* d_hitva = nullptr; uint64_t hitva_cap = 0;   // aux words of every row, then val words (4 + 4 bytes per position)
    uint32_t* d_hitnk = nullptr; uint64_t hitnk_cap = 0;
    uint64_t* d_hitoff = nullptr; uint64_t hitoff_cap = 0;
    uint32_t* d_gen = nullptr; uint64_t gen_cap = 0;        // K3a -> K3b
    uint32_t* d_tickets = nullptr; uint64_t tickets_cap = 0;
    uint32_t* d_walk = nullptr; uint64_t walk_cap = 0;       // threading = 2, per survivor: destLocus [cap], then the walk's return codes [cap]
    dbtk_thread_rec_t* d_trecs = nullptr; uint64_t trecs_cap = 0;  // thread records (function mode; pair mode with trace / -a)
    uint32_t* d_loci = nullptr; uint64_t loci_cap = 0;       // function mode: locus per read
    uint8_t* d_aln = nullptr; uint64_t aln_bytes = 0;        // -a / -ae: compact alignment records of the last host-buffer batch
    uint8_t* d_txt = nullptr; uint64_t txt_bytes = 0;        // -a / -ae | DBTK_ALN_TEXT: the text arena of the last host-buffer batch
    uint32_t* d_txtidx = nullptr; uint64_t txtidx_cap = 0;   //   ... and its per-pair index
    uint64_t txt_cap = 0;                                    //   bytes of the arena the last batch could use
    uint32_t aln_stride = 0, aln_cap = 0; uint64_t aln_max = 0;
    uint8_t* h_aln = nullptr; size_t h_aln_bytes = 0;  // pinned staging of dbtk_ctx_aln_records
    uint64_t last_walk_npairs = 0; bool last_walk_recs = false;  // what dbtk_ctx_walk_results may fetch
    int walk_blocks = 0, walkfast_blocks = 0;
    // optional gates
    ClsSlot* d_tre = nullptr; ClsSlot* d_bait = nullptr;
    uint8_t* d_qual = nullptr; uint64_t qual_cap = 0;
    uint64_t* d_edge = nullptr; uint64_t edge_cap = 0;
    uint64_t* d_qmask = nullptr; uint64_t qmask_cap = 0;
    BubEvent* d_events = nullptr; uint64_t events_cap = 0;
    uint32_t* d_nevents = nullptr;
    // bubbleDB (bubble_db_t: per locus unordered_map<size_t, uint32_t>, src/aQueryFasta_thread.h:41), filled batch by
    // batch exactly like accumBubbles (AQ.cpp:1599-1606) so that the dump order matches the reference's
    std::vector<std::unordered_map<size_t, uint32_t>> bubbleDB;
    // -tb: btTK (bt_tracker_db_t, src/aQueryFasta_thread.h:44) and a host copy of baitDB for the replay
    std::vector<std::unordered_map<uint64_t, uint64_t>> btTK;
    std::vector<std::unordered_map<uint64_t, uint16_t>> baitDB_host;
    std::vector<dbtk_pair_rec_t> own_recs;  // record buffer when the caller passes none but -tb needs the bait-stage records
    uint64_t mz_turned = 0;       // keys in the overflow table
    int k1_blocks = 0, probe_wpc = 32, probe2_wpc[4] = {8, 8, 8, 8}, probe2f_wpc[4] = {8, 8, 8, 8};
    bool timers_on = true;
    uint32_t timers_every = 1;  // event records around the kernels of every n-th batch (8 records cost ~30 us per batch)
    uint64_t batch_no = 0;
    uint64_t* d_vote = nullptr;
    uint32_t* d_epoch = nullptr;
    int pair_blocks[3] = {0, 0, 0}, usual_blocks[3] = {0, 0, 0}, num_cu = 0, max_pair_blocks = 0;
    int vote_rows = 0;  // rows of the vote-spill pool: the workgroups of the general resolve kernel that can be resident at once
    uint32_t consistent = 0;
    Timed timed[NKERN];
    // Second lane of the device-resident entry point: successive batches alternate between two streams, each with its own
    // per-batch scratch, so that one batch's VALU-bound encode kernel overlaps the other's request-bound probe kernel
    // (tables and accumulators are shared; the adds are atomic).  `alt` holds the lane that is not current.
    struct Lane {
        hipStream_t stream = nullptr;
        uint32_t* d_small = nullptr;
        uint32_t* d_surv = nullptr; uint64_t surv_cap = 0; uint32_t* d_sorted = nullptr;
        uint64_t* d_hitva = nullptr; uint64_t hitva_cap = 0;   // aux words of every row, then val words (4 + 4 bytes per position)
        uint32_t* d_hitnk = nullptr; uint64_t hitnk_cap = 0;
        uint64_t* d_hitoff = nullptr; uint64_t hitoff_cap = 0;
        uint32_t* d_gen = nullptr; uint64_t gen_cap = 0;
        uint32_t* d_tickets = nullptr; uint64_t tickets_cap = 0;
        uint32_t* d_walk = nullptr; uint64_t walk_cap = 0;
        uint64_t* d_vote = nullptr;
        uint32_t* d_epoch = nullptr;
        // blocks of the device reader merged into one batch (dbtk_ingest_align_merged): reads back to back, offsets
        uint8_t* m_flat = nullptr; uint64_t m_flat_cap = 0;
        uint64_t* m_off = nullptr; uint64_t m_off_cap = 0;
        uint64_t m_bytes = 0, m_pairs = 0; uint32_t m_maxlen = 0;
    } alt;
    std::deque<Lane> parked;  // lanes beyond the second (DBTK_LANES > 2): switch_lane goes round all of them
    bool two_lanes = false;   // more than one lane
};

namespace {

uint64_t pow2_at_least(uint64_t n) {
    uint64_t c = 1024;
    while (c < n) c <<= 1;
    return c;
}
uint32_t log2u(uint64_t c) { return 63u - (uint32_t)__builtin_clzll(c); }

std::mutex g_share_m;
std::map<std::pair<uint64_t, int>, TableShare*> g_shares;  // (handle's uid, device): see dbtk_rpgg::uid

void release_share(dbtk_ctx* c) {
    std::lock_guard<std::mutex> l(g_share_m);
    TableShare* sh = c->share;
    if (!sh) {  // the context never got as far as sharing: the tables (if any) are its own
        void* own[] = {c->d_flt, c->d_trbeg, c->d_idx, c->d_cls, c->d_vv, c->d_qc, c->d_perm, c->d_tre, c->d_bait, c->d_gr, c->d_grmz, c->d_mz, c->d_ovf, c->d_ldir, c->d_limg, c->d_gldir, c->d_glimg};
        for (void* p : own) if (p) (void)hipFree(p);
        return;
    }
    if (--sh->refs > 0) return;
    void* ptrs[] = {sh->d_flt, sh->d_trbeg, sh->d_idx, sh->d_cls, sh->d_vv, sh->d_qc, sh->d_perm, sh->d_tre, sh->d_bait, sh->d_gr, sh->d_grmz, sh->d_mz, sh->d_ovf, sh->d_ldir, sh->d_limg, sh->d_gldir, sh->d_glimg};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    g_shares.erase(std::make_pair(c->g_uid, c->device));
    delete sh;
}

void free_ctx(dbtk_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    release_share(c);
    for (int i = 0; i < NKERN; ++i)
        for (int j = 0; j < EVPOOL; ++j) {
            if (c->timed[i].beg[j]) (void)hipEventDestroy(c->timed[i].beg[j]);
            if (c->timed[i].end[j]) (void)hipEventDestroy(c->timed[i].end[j]);
        }
    void* ptrs[] = {c->m_flat, c->m_off, c->d_ctr, c->d_pstats, c->d_accum, c->d_small ? c->d_small - TK_INLINE : nullptr, c->d_surv,
                    c->d_seq, c->d_off, c->d_recs, c->d_vote, c->d_epoch, c->d_hitva, c->d_hitnk, c->d_hitoff, c->d_gen, c->d_tickets,
                    c->d_qual, c->d_edge, c->d_qmask, c->d_events, c->d_nevents, c->d_walk, c->d_trecs, c->d_loci, c->d_aln, c->d_txt, c->d_txtidx};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (c->h_aln) (void)hipHostFree(c->h_aln);
    if (c->h_sortflag) (void)hipHostFree(c->h_sortflag);
    std::vector<dbtk_ctx::Lane*> others{&c->alt};
    for (auto& l : c->parked) others.push_back(&l);
    for (dbtk_ctx::Lane* l : others) {
        void* aptrs[] = {l->d_small ? l->d_small - TK_INLINE : nullptr, l->d_surv, l->d_hitva, l->d_hitnk, l->d_hitoff, l->d_gen, l->d_tickets, l->d_vote, l->d_epoch, l->d_walk, l->m_flat, l->m_off};
        for (void* p : aptrs) if (p) (void)hipFree(p);
        if (l != &c->alt && l->stream) (void)hipStreamDestroy(l->stream);
    }
    if (c->alt.stream) (void)hipStreamDestroy(c->alt.stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

dbtk_status_t build_locus_images(dbtk_ctx* c);
dbtk_status_t build_graph_images(dbtk_ctx* c);
dbtk_status_t build_tables(dbtk_ctx* c) {
    const dbtk_rpgg* g = c->g;
    const uint64_t nloci = g->nloci;
    hipStream_t s = c->stream;
    // (DBTK_VERBOSE: where the start-up goes)
    const bool verbose = getenv("DBTK_VERBOSE") != nullptr;
    auto wallclk = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tick_t = wallclk();
    auto tick = [&](const char* what) { if (!verbose) return; (void)hipStreamSynchronize(s); const double n = wallclk(); fprintf(stderr, "tables: %s %.3f s\n", what, n - tick_t); tick_t = n; };
    // ---- index
    const uint64_t nkeys = g->keys.size();
    // slots = 4 per bucket; DBTK_IDX_SPARSITY (default 2) = minimum slots per key before rounding up to a power of two:
    // at 2..4 slots per key (8.6 GB at release scale; round 3 kept 4..8: 17 GB) one bucket in fifty is full, and the look-ups that
    // then go on to the next one cost the probe kernels 2 % (tools/footprint_sweep.sh)
    uint64_t sparsity = 2;
    if (const char* e = getenv("DBTK_IDX_SPARSITY")) { const long v = atol(e); if (v >= 2 && v <= 64) sparsity = (uint64_t)v; }
    const uint64_t icap = pow2_at_least(sparsity * nkeys + 8), nbkt = icap / 4;
    HIPCHK(hipMalloc(&c->d_idx, nbkt * sizeof(IdxBucket)));
    c->tb_idx = nbkt * sizeof(IdxBucket);
    LAUNCH(k_fill_idx, dim3(2048), dim3(256), s, c->d_idx, icap);
    if (nkeys) {
        uint64_t* dk = nullptr; uint32_t* dv = nullptr;
        HIPCHK(hipMalloc(&dk, nkeys * 8));
        HIPCHK(hipMalloc(&dv, nkeys * 4));
        HIPCHK(hipMemcpyAsync(dk, g->keys.data(), nkeys * 8, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(dv, g->vals.data(), nkeys * 4, hipMemcpyHostToDevice, s));
        IdxBuildArgs a{c->d_idx, nbkt - 1, 64 - log2u(nbkt), dk, dv, nkeys};
        LAUNCH(k_idx_insert, dim3(2048), dim3(256), s, a);
        {   // presence filter: DBTK_FILTER_BPK bits per key (default 4, 0 = none), rounded up to a power of two of words
            uint64_t bpk = 4;
            if (const char* e = getenv("DBTK_FILTER_BPK")) { const long v = atol(e); if (v >= 0 && v <= 64) bpk = (uint64_t)v; }
            if (bpk) {
                c->flt_words = pow2_at_least((bpk * nkeys + 63) / 64);
                HIPCHK(hipMalloc(&c->d_flt, c->flt_words * 8));
                c->tb_flt = c->flt_words * 8;
                HIPCHK(hipMemsetAsync(c->d_flt, 0, c->flt_words * 8, s));
                FltBuildArgs fa{c->d_flt, log2u(c->flt_words), g->ksize, dk, nkeys};
                LAUNCH(k_flt_insert, dim3(2048), dim3(256), s, fa);
            }
        }
        LAUNCH(k_idx_finalize, dim3(2048), dim3(256), s, c->d_idx, icap);
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipFree(dk));
        HIPCHK(hipFree(dv));
    }
    tick("index + presence filter (H2D of keys and values, insert, finalize)");
    // ---- vv (never empty on the device: odd vals index it)
    HIPCHK(hipMalloc(&c->d_vv, (g->vv.size() + 1) * 4));
    if (!g->vv.empty()) HIPCHK(hipMemcpyAsync(c->d_vv, g->vv.data(), g->vv.size() * 4, hipMemcpyHostToDevice, s));
    // ---- class table: TR pass first, then flank (flank overrides)
    const uint64_t ntrf = g->tr_ks.size(), nfl = g->fl_ks.size();
    uint64_t cls_pct = 130;  // slots per entry, in percent, before rounding up to a power of two (DBTK_CLS_SPARSITY_PCT): only k-mers shared between loci are looked up here
    if (const char* e = getenv("DBTK_CLS_SPARSITY_PCT")) { const long v = atol(e); if (v >= 110 && v <= 1600) cls_pct = (uint64_t)v; }
    const uint64_t ccap = pow2_at_least((ntrf + nfl) * cls_pct / 100 + 2);
    HIPCHK(hipMalloc(&c->d_cls, ccap * sizeof(ClsSlot)));
    c->tb_cls = ccap * sizeof(ClsSlot);
    HIPCHK(hipMemsetAsync(c->d_cls, 0xFF, ccap * sizeof(ClsSlot), s));
    uint64_t* dstats = nullptr;  // [0] index memberships, [1] of them missing from the class table, [2] class entries
    HIPCHK(hipMalloc(&dstats, 3 * 8));
    HIPCHK(hipMemsetAsync(dstats, 0, 3 * 8, s));
    {
        std::vector<uint64_t> beg(nloci + 1, 0);
        uint64_t *dks = nullptr, *dbeg = nullptr, *dslot = nullptr;
        const uint64_t nmax = ntrf > nfl ? ntrf : nfl;
        HIPCHK(hipMalloc(&dks, (nmax + 1) * 8));
        HIPCHK(hipMalloc(&dslot, (ntrf + 1) * 8));
        HIPCHK(hipMalloc(&dbeg, (nloci + 1) * 8));
        for (uint64_t l = 0; l < nloci; ++l) beg[l + 1] = beg[l] + g->tr_cnt[l];
        if (ntrf) {
            HIPCHK(hipMemcpyAsync(dks, g->tr_ks.data(), ntrf * 8, hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(dslot, g->out_slot.data(), ntrf * 8, hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(dbeg, beg.data(), (nloci + 1) * 8, hipMemcpyHostToDevice, s));
            ClsBuildArgs a{c->d_cls, ccap - 1, 64 - log2u(ccap), dks, dbeg, (uint32_t)nloci, dslot, ntrf, dstats + 2};
            LAUNCH(k_cls_insert, dim3(2048), dim3(256), s, a);
            HIPCHK(hipStreamSynchronize(s));
        }
        for (uint64_t l = 0; l < nloci; ++l) beg[l + 1] = beg[l] + g->fl_cnt[l];
        if (nfl) {
            HIPCHK(hipMemcpyAsync(dks, g->fl_ks.data(), nfl * 8, hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(dbeg, beg.data(), (nloci + 1) * 8, hipMemcpyHostToDevice, s));
            ClsBuildArgs a{c->d_cls, ccap - 1, 64 - log2u(ccap), dks, dbeg, (uint32_t)nloci, nullptr, nfl, dstats + 2};
            LAUNCH(k_cls_insert, dim3(2048), dim3(256), s, a);
            HIPCHK(hipStreamSynchronize(s));
        }
        HIPCHK(hipFree(dks));
        HIPCHK(hipFree(dslot));
        HIPCHK(hipFree(dbeg));
    }
    tick("class table (H2D of TR and flank k-mers, insert)");
    // ---- QC mask
    if (!g->qc.empty()) {
        HIPCHK(hipMalloc(&c->d_qc, nloci));
        HIPCHK(hipMemcpyAsync(c->d_qc, g->qc.data(), nloci, hipMemcpyHostToDevice, s));
    }
    // ---- introsort permutation of n equal keys (the common case: every k-mer unique to one locus)
    {
        std::vector<uint16_t> perm((size_t)NHMAX * (NHMAX + 1) / 2 + 1);
        std::vector<uint32_t> key(NHMAX, 1);
        int stack[3 * 40];
        for (int n = 1; n <= NHMAX; ++n) gcc_sort_index(perm.data() + (size_t)n * (n - 1) / 2, n, key.data(), stack);
        HIPCHK(hipMalloc(&c->d_perm, perm.size() * 2));
        HIPCHK(hipMemcpyAsync(c->d_perm, perm.data(), perm.size() * 2, hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    {   // first output slot of each locus
        std::vector<uint32_t> tb(nloci + 1, 0);
        for (uint64_t l = 0; l <= nloci; ++l) tb[l] = (uint32_t)g->out_beg[l];
        HIPCHK(hipMalloc(&c->d_trbeg, (nloci + 1) * 4));
        HIPCHK(hipMemcpyAsync(c->d_trbeg, tb.data(), (nloci + 1) * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    DevTables& T = c->T;
    memset(&T, 0, sizeof(T));  // optional tables (tre, bait, gr, mz) stay null unless built
    T.trbeg = c->d_trbeg;
    T.flt = c->d_flt; T.flt_logw = c->flt_words ? log2u(c->flt_words) : 0;
    T.idx = c->d_idx; T.idx_mask = nbkt - 1; T.idx_shift = 64 - log2u(nbkt);
    T.vv = c->d_vv;
    T.cls = c->d_cls; T.cls_mask = ccap - 1; T.cls_shift = 64 - log2u(ccap);
    T.qc = c->d_qc;
    T.permtab = c->d_perm;
    T.nloci = (uint32_t)nloci;
    T.ksize = g->ksize;
    T.consistent = 0;
    {   // class of single-locus k-mers into the index slots + index-vs-sets consistency verdict
        IdxAuxArgs a{c->d_idx, icap, T, dstats};
        LAUNCH(k_idx_aux, dim3(2048), dim3(256), s, a);
        uint64_t st[3] = {0, 0, 0};
        HIPCHK(hipMemcpyAsync(st, dstats, sizeof(st), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipFree(dstats));
        T.consistent = (st[1] == 0 && st[0] == st[2]) ? 1u : 0u;
        c->consistent = T.consistent;
    }
    tick("permutation table, counters' first slots, index consistency + classes");
    {   // the probe kernel's minimizer-grouped copy of the index, for the values of k its lean form exists for
        // (DBTK_MZ=0: do without: the general probe kernel then looks every position up in the plain index)
        bool on = true;
        if (const char* e = getenv("DBTK_MZ")) on = atoi(e) != 0;
        const uint32_t m = mz_m_for_k(g->ksize);
        if (on && m && nkeys) {
            uint64_t per = 3;  // buckets per 8 keys (8.6 GB at release scale; 6 made no measurable difference: the dense batches are the locus-resident kernel's now)
            if (const char* e = getenv("DBTK_MZ_SPARSITY")) { const long v = atol(e); if (v >= 1 && v <= 64) per = (uint64_t)v; }
            uint64_t nb = pow2_at_least(nkeys * per / 8 + 8);
            if (nb > (1ull << 28)) nb = 1ull << 28;  // the bucket number comes out of 28 bits of the minimizer's hash
            HIPCHK(hipMalloc(&c->d_mz, nb * sizeof(MzBucket)));
            c->tb_mz = nb * sizeof(MzBucket);
            LAUNCH(k_mz_fill, dim3(2048), dim3(256), s, reinterpret_cast<uint64_t*>(c->d_mz), nb * 16, 1);
            uint64_t* dn = nullptr;
            HIPCHK(hipMalloc(&dn, 8));
            HIPCHK(hipMemsetAsync(dn, 0, 8, s));
            MzBuildArgs a{c->d_idx, icap, c->d_mz, (uint32_t)(nb - 1), nullptr, 0, g->ksize, m, 0, dn};
            LAUNCH(k_mz_insert, dim3(2048), dim3(256), s, a);  // level 1; counts the keys it turns away
            uint64_t nturned = 0;
            HIPCHK(hipMemcpyAsync(&nturned, dn, 8, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            HIPCHK(hipFree(dn));
            uint64_t ovf_sp = 8;  // level 2: those keys, at most an eighth full (DBTK_OVF_SPARSITY; a sixteenth: twice the bytes for 1 % of the lean kernel's time)
            if (const char* e = getenv("DBTK_OVF_SPARSITY")) { const long v = atol(e); if (v >= 2 && v <= 64) ovf_sp = (uint64_t)v; }
            uint64_t ocap = pow2_at_least(ovf_sp * nturned + 8);
            if (ocap > (1ull << 32)) { set_error("overflow table of the probe kernel: more than 2^28 keys turned away by full buckets"); return DBTK_ERR_UNSUPPORTED; }
            HIPCHK(hipMalloc(&c->d_ovf, ocap * sizeof(MzSlot)));
            c->tb_ovf = ocap * sizeof(MzSlot);
            LAUNCH(k_mz_fill, dim3(2048), dim3(256), s, reinterpret_cast<uint64_t*>(c->d_ovf), ocap * 2, 0);
            a.ovf = c->d_ovf; a.ovf_mask = (uint32_t)(ocap - 1); a.pass = 1;
            LAUNCH(k_mz_insert, dim3(2048), dim3(256), s, a);
            HIPCHK(hipStreamSynchronize(s));
            HIPCHK(hipGetLastError());
            T.mz = c->d_mz; T.mz_mask = nb - 1; T.mz_m = m;
            T.ovf = c->d_ovf; T.ovf_mask = ocap - 1;
            c->mz_turned = nturned;
        }
    }
    tick("index by minimizer + overflow table");
    const dbtk_status_t li = build_locus_images(c);
    tick("index images");
    return li;
}

// ---- the sidecar of the per-locus images (dbtk.h: dbtk_rpgg_set_index_cache): header, directory, arena — the bytes as they lie in HBM
struct LocCacheHdr {
    char magic[8];           // "DBTKIDX\1"
    uint32_t version, ksize;
    uint64_t nloci, nkeys, fingerprint, arena_bytes, nimg, left_out;
    uint32_t lg_max, hdr_bytes;
    uint64_t checksum;       // of the directory and the arena (csum_term over their 8-byte words): a torn, mixed or bit-flipped file is not used
};
constexpr uint32_t LOC_CACHE_VERSION = 5;  // (bumped with every change of the image layout or of its hashes)
// what the images were built from: the handle's arrays, sampled (a different RPGG, another -t N order, a changed file: another value)
static uint64_t rpgg_fingerprint(const dbtk_rpgg* g) {
    uint64_t h = 0xCBF29CE484222325ull;
    auto mixin = [&](uint64_t v) { h = (h ^ v) * 0x100000001B3ull; h ^= h >> 29; };
    mixin(g->ksize); mixin(g->nloci); mixin(g->keys.size()); mixin(g->vv.size()); mixin(g->tr_ks.size()); mixin(g->fl_ks.size());
    auto sample = [&](const auto& v) { const size_t n = v.size(), st = n / 65536 + 1; for (size_t i = 0; i < n; i += st) mixin((uint64_t)v[i]); if (n) mixin((uint64_t)v[n - 1]); };
    sample(g->keys); sample(g->vals); sample(g->vv); sample(g->tr_ks); sample(g->fl_ks); sample(g->out_slot); sample(g->out_beg); sample(g->tr_cnt); sample(g->fl_cnt);
    return h;
}
// checksum of the sidecar's payload: the directory (host) and the arena as it lies in HBM (device)
static dbtk_status_t locus_cache_checksum(dbtk_ctx* c, const std::vector<LocusDir>& dir, const uint8_t* d_arena, uint64_t arena_bytes, uint64_t* out) {
    static_assert(sizeof(LocusDir) == 16, "directory entries are two checksum words");
    uint64_t hd = 0;
    const uint64_t* dw = reinterpret_cast<const uint64_t*>(dir.data());
    for (uint64_t i = 0; i < 2 * dir.size(); ++i) hd += csum_term(dw[i], i);
    uint64_t* dsum = nullptr;
    HIPCHK(hipMalloc(&dsum, 8));
    HIPCHK(hipMemsetAsync(dsum, 0, 8, c->stream));
    LAUNCH(k_csum, dim3(2048), dim3(256), c->stream, reinterpret_cast<const uint64_t*>(d_arena), arena_bytes / 8, dsum);
    uint64_t ha = 0;
    HIPCHK(hipMemcpyAsync(&ha, dsum, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipFree(dsum));
    *out = ha + hd * 0x9E3779B97F4A7C15ull + arena_bytes;
    return DBTK_OK;
}
// DBTK_OK: the images are in HBM, from the file; DBTK_ERR_FORMAT: no file, or not one of this RPGG / this layout / this build: the caller builds
static dbtk_status_t load_locus_cache(dbtk_ctx* c, uint64_t fp) {
    const dbtk_rpgg* g = c->g;
    const uint64_t nloci = g->nloci;
    FILE* f = fopen(g->idx_cache.c_str(), "rb");
    if (!f) return DBTK_ERR_FORMAT;
    struct Closer { FILE* f; ~Closer() { if (f) fclose(f); } } closer{f};
    LocCacheHdr h;
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "DBTKIDX\1", 8) != 0 || h.version != LOC_CACHE_VERSION || h.hdr_bytes != sizeof(h) ||
        h.ksize != g->ksize || h.nloci != nloci || h.nkeys != g->keys.size() || h.fingerprint != fp || h.lg_max != LOC_LG_MAX ||
        h.arena_bytes > (16ull << 32) || (h.arena_bytes & 15)) return DBTK_ERR_FORMAT;
    std::vector<LocusDir> dir(nloci);
    if (fread(dir.data(), sizeof(LocusDir), nloci, f) != nloci) return DBTK_ERR_FORMAT;
    uint64_t nimg = 0;
    for (uint64_t l = 0; l < nloci; ++l) {
        const LocusDir& d = dir[l];
        if (!d.bytes) continue;
        if (d.lgnb < loc_lg_min(g->ksize) || d.lgnb > LOC_LG_MAX || d.bytes != loc_image_bytes(d.lgnb) || 16ull * d.off16 + d.bytes > h.arena_bytes ||
            d.trbeg != (uint32_t)g->out_beg[l]) return DBTK_ERR_FORMAT;
        ++nimg;
    }
    if (!nimg) return DBTK_ERR_FORMAT;
    hipStream_t s = c->stream;
    HIPCHK(hipMalloc(&c->d_ldir, nloci * sizeof(LocusDir)));
    HIPCHK(hipMalloc(&c->d_limg, h.arena_bytes + 16));
    {   // the arena, through a pinned buffer, 64 MB at a time
        const uint64_t CH = 64ull << 20;
        uint8_t* pin = nullptr;
        HIPCHK(hipHostMalloc((void**)&pin, CH, hipHostMallocDefault));
        bool ok = true;
        for (uint64_t at = 0; at < h.arena_bytes && ok; at += CH) {
            const uint64_t n = std::min<uint64_t>(CH, h.arena_bytes - at);
            ok = fread(pin, 1, n, f) == n && hipMemcpy(c->d_limg + at, pin, n, hipMemcpyHostToDevice) == hipSuccess;
        }
        (void)hipHostFree(pin);
        if (!ok) { (void)hipFree(c->d_ldir); (void)hipFree(c->d_limg); c->d_ldir = nullptr; c->d_limg = nullptr; return DBTK_ERR_FORMAT; }
    }
    auto drop = [&]() { (void)hipFree(c->d_ldir); (void)hipFree(c->d_limg); c->d_ldir = nullptr; c->d_limg = nullptr; return DBTK_ERR_FORMAT; };
    {   // the bytes are the ones that were written (a torn write, two writers' blocks mixed, a flipped bit: not used)
        uint64_t sum = 0;
        const dbtk_status_t cs = locus_cache_checksum(c, dir, c->d_limg, h.arena_bytes, &sum);
        if (cs) { (void)drop(); return cs; }
        if (sum != h.checksum) return drop();
    }
    HIPCHK(hipMemcpyAsync(c->d_ldir, dir.data(), nloci * sizeof(LocusDir), hipMemcpyHostToDevice, s));
    // ... and they are THIS index's images: every entry looked up in the plain index just built from the RPGG, the entries of a locus
    // counted against its keys in the index (a file of another RPGG of the same sizes, one rebuilt in place: not used)
    uint32_t *dbad = nullptr, *dvcnt = nullptr, *dcnt = nullptr;
    HIPCHK(hipMalloc(&dbad, nloci * 4));
    HIPCHK(hipMalloc(&dvcnt, nloci * 4));
    HIPCHK(hipMalloc(&dcnt, nloci * 4));
    HIPCHK(hipMemsetAsync(dbad, 0, nloci * 4, s));
    HIPCHK(hipMemsetAsync(dvcnt, 0, nloci * 4, s));
    HIPCHK(hipMemsetAsync(dcnt, 0, nloci * 4, s));
    LocBuildArgs a;
    memset(&a, 0, sizeof(a));
    a.idx = c->d_idx; a.nslots = (c->T.idx_mask + 1) * 4; a.idx_mask = c->T.idx_mask; a.idx_shift = c->T.idx_shift; a.vv = c->d_vv;
    a.cls = c->T.cls; a.cls_mask = c->T.cls_mask; a.cls_shift = c->T.cls_shift;
    a.trbeg = c->d_trbeg; a.nloci = (uint32_t)nloci; a.ksize = g->ksize; a.dir = c->d_ldir; a.arena = c->d_limg; a.bad = dbad; a.vcnt = dvcnt; a.cnt = dcnt;
    LAUNCH(k_loc_count, dim3(2048), dim3(256), s, a);
    LAUNCH(k_loc_verify, dim3(4096), dim3(256), s, a);
    std::vector<uint32_t> bad(nloci), vcnt(nloci), cnt(nloci);
    HIPCHK(hipMemcpyAsync(bad.data(), dbad, nloci * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(vcnt.data(), dvcnt, nloci * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(cnt.data(), dcnt, nloci * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipFree(dbad)); HIPCHK(hipFree(dvcnt)); HIPCHK(hipFree(dcnt));
    for (uint64_t l = 0; l < nloci; ++l)
        if (bad[l] || (dir[l].bytes && vcnt[l] != cnt[l])) return drop();  // nothing of such a file is used
    c->limg_bytes = h.arena_bytes; c->loc_nimg = nimg; c->loc_left_out = h.left_out; c->loc_from_cache = true;
    c->T.ldir = c->d_ldir; c->T.limg = c->d_limg;
    if (getenv("DBTK_VERBOSE")) fprintf(stderr, "locus images: %llu loci, %.1f MB, from %s\n", (unsigned long long)nimg, h.arena_bytes / 1e6, g->idx_cache.c_str());
    return DBTK_OK;
}
// (best effort: a sidecar that cannot be written is not an error of the run)
static void write_locus_cache(dbtk_ctx* c, uint64_t fp, const std::vector<LocusDir>& dir) {
    const dbtk_rpgg* g = c->g;
    // (a name of its own per writer: two jobs on one RPGG prefix must not write into one file)
    std::random_device rd;
    const std::string tmp = g->idx_cache + ".tmp" + std::to_string((unsigned long long)getpid()) + "." + std::to_string((unsigned long long)g->uid) + "." + std::to_string((unsigned long long)rd());
    uint64_t sum = 0;
    if (locus_cache_checksum(c, dir, c->d_limg, (c->limg_bytes + 15) & ~15ull, &sum)) return;
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return;
    LocCacheHdr h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "DBTKIDX\1", 8);
    h.version = LOC_CACHE_VERSION; h.ksize = g->ksize; h.nloci = g->nloci; h.nkeys = g->keys.size(); h.fingerprint = fp;
    h.arena_bytes = (c->limg_bytes + 15) & ~15ull; h.nimg = c->loc_nimg; h.left_out = c->loc_left_out; h.lg_max = LOC_LG_MAX; h.hdr_bytes = sizeof(h); h.checksum = sum;
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(dir.data(), sizeof(LocusDir), dir.size(), f) == dir.size();
    const uint64_t CH = 64ull << 20;
    uint8_t* pin = nullptr;
    if (ok && hipHostMalloc((void**)&pin, CH, hipHostMallocDefault) == hipSuccess) {
        for (uint64_t at = 0; at < h.arena_bytes && ok; at += CH) {
            const uint64_t n = std::min<uint64_t>(CH, h.arena_bytes - at);
            ok = hipMemcpy(pin, c->d_limg + at, n, hipMemcpyDeviceToHost) == hipSuccess && fwrite(pin, 1, n, f) == n;
        }
        (void)hipHostFree(pin);
    } else ok = false;
    if (ok && (fflush(f) != 0 || fsync(fileno(f)) != 0)) ok = false;  // (on the disk before it takes the name)
    if (fclose(f) != 0) ok = false;
    if (!ok || rename(tmp.c_str(), g->idx_cache.c_str()) != 0) remove(tmp.c_str());
}

// Per-locus images of the index (dbtk_locus.h), from the finished plain index: keys per locus -> image sizes (host) -> empty images
// -> every (key, locus) membership into its locus' image.  DBTK_LOCUS=0: do without (the global tables answer every look-up).
// the images' placement: a wave per locus, three launches by image class (the LDS a block needs: 19 / 37 / 83 KB)
static dbtk_status_t launch_loc_place(const LocBuildArgs& a, uint64_t nloci, hipStream_t s) {
    static const bool per_thread = [] { const char* e = getenv("DBTK_LOC_PLACE_THREAD"); return e && atoi(e) != 0; }();
    if (per_thread) { LAUNCH(k_loc_place, dim3((uint32_t)((nloci + 63) / 64)), dim3(64), s, a); return DBTK_OK; }
    const dim3 g((uint32_t)std::min<uint64_t>(nloci, 8192));
    LAUNCH((k_loc_place_wave<9, -1>), g, dim3(64), s, a);
    LAUNCH((k_loc_place_wave<10, 9>), g, dim3(64), s, a);
    LAUNCH((k_loc_place_wave<11, 10>), g, dim3(64), s, a);
    return DBTK_OK;
}

dbtk_status_t build_locus_images(dbtk_ctx* c) {
    const dbtk_rpgg* g = c->g;
    hipStream_t s = c->stream;
    const uint64_t nloci = g->nloci;
    if (const char* e = getenv("DBTK_LOCUS")) if (!atoi(e)) return DBTK_OK;
    if (!nloci || g->keys.empty() || loc_lg_min(g->ksize) > LOC_LG_MAX || !c->T.consistent) return DBTK_OK;
    const uint64_t fp = rpgg_fingerprint(g);
    if (g->idx_cache_mode >= 1 && !g->idx_cache.empty()) {
        const dbtk_status_t lc = load_locus_cache(c, fp);
        if (lc != DBTK_ERR_FORMAT) return lc;  // loaded (DBTK_OK), or a device error; DBTK_ERR_FORMAT: no usable file: build
    }
    static const bool verbose = getenv("DBTK_VERBOSE") != nullptr;
    double tk0 = 0;
    auto tk = [&](const char* what) {  // (DBTK_VERBOSE: where the build's time goes)
        if (!verbose) return;
        (void)hipStreamSynchronize(s);
        timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
        const double now = t.tv_sec + 1e-9 * t.tv_nsec;
        if (tk0 != 0) fprintf(stderr, "  images: %s %.3f s\n", what, now - tk0);
        tk0 = now;
    };
    tk("");
    uint32_t *dcnt = nullptr, *dbad = nullptr;
    HIPCHK(hipMalloc(&dcnt, nloci * 4));
    HIPCHK(hipMalloc(&dbad, nloci * 4));
    HIPCHK(hipMemsetAsync(dcnt, 0, nloci * 4, s));
    HIPCHK(hipMemsetAsync(dbad, 0, nloci * 4, s));
    LocBuildArgs a;
    memset(&a, 0, sizeof(a));
    a.idx = c->d_idx; a.nslots = (c->T.idx_mask + 1) * 4; a.vv = c->d_vv; a.trbeg = c->d_trbeg; a.nloci = (uint32_t)nloci; a.ksize = g->ksize;
    a.cls = c->T.cls; a.cls_mask = c->T.cls_mask; a.cls_shift = c->T.cls_shift;
    a.cnt = dcnt; a.bad = dbad;
    LAUNCH(k_loc_count, dim3(2048), dim3(256), s, a);
    std::vector<uint32_t> cnt(nloci), bad(nloci);
    HIPCHK(hipMemcpyAsync(cnt.data(), dcnt, nloci * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    tk("count");
    std::vector<LocusDir> dir(nloci);
    std::vector<uint64_t> ebeg(nloci + 1, 0);
    uint64_t at = 0, nimg = 0;
    for (uint64_t l = 0; l < nloci; ++l) {
        const uint32_t lg = loc_lgnb_for(cnt[l], g->ksize);
        dir[l] = LocusDir{(uint32_t)(at / 16), 0u, lg, (uint32_t)g->out_beg[l]};
        ebeg[l + 1] = ebeg[l];
        if (!cnt[l] || lg > LOC_LG_MAX || cnt[l] > 0xFFF0u || at + loc_image_bytes(lg) > (16ull << 32)) continue;
        dir[l].bytes = loc_image_bytes(lg);
        at += dir[l].bytes;
        ebeg[l + 1] += cnt[l];
        ++nimg;
    }
    if (!nimg) { HIPCHK(hipFree(dcnt)); HIPCHK(hipFree(dbad)); return DBTK_OK; }
    const uint64_t nent = ebeg[nloci];
    const uint32_t gstride = 2 * (1u << LOC_LG_MAX) + 2;
    uint64_t *debeg = nullptr, *dekey = nullptr, *dskey = nullptr, *dnleft = nullptr;
    uint32_t *depay = nullptr, *dspay = nullptr;
    uint16_t* dgscr = nullptr;
    HIPCHK(hipMalloc(&c->d_ldir, nloci * sizeof(LocusDir)));
    HIPCHK(hipMalloc(&c->d_limg, at + 16));
    HIPCHK(hipMalloc(&debeg, (nloci + 1) * 8));
    HIPCHK(hipMalloc(&dekey, (nent + 1) * 8)); HIPCHK(hipMalloc(&dskey, (nent + 1) * 8));
    HIPCHK(hipMalloc(&depay, (nent + 1) * 4)); HIPCHK(hipMalloc(&dspay, (nent + 1) * 4));
    HIPCHK(hipMalloc(&dgscr, nloci * (uint64_t)gstride * 2));
    HIPCHK(hipMalloc(&dnleft, 8));
    HIPCHK(hipMemsetAsync(dnleft, 0, 8, s));
    HIPCHK(hipMemsetAsync(dcnt, 0, nloci * 4, s));  // (now the gather cursors)
    HIPCHK(hipMemcpyAsync(c->d_ldir, dir.data(), nloci * sizeof(LocusDir), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(debeg, ebeg.data(), (nloci + 1) * 8, hipMemcpyHostToDevice, s));
    a.dir = c->d_ldir; a.arena = c->d_limg; a.ebeg = debeg; a.ecur = dcnt; a.ekey = dekey; a.epay = depay; a.skey = dskey; a.spay = dspay;
    a.gscr = dgscr; a.gstride = gstride; a.nleft = dnleft;
    tk("directory + allocations");
    LAUNCH(k_loc_scatter, dim3(2048), dim3(256), s, a);
    tk("scatter");
    { const dbtk_status_t stp = launch_loc_place(a, nloci, s); if (stp) return stp; }
    tk("place");
    uint64_t nleft = 0;
    HIPCHK(hipMemcpyAsync(bad.data(), dbad, nloci * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&nleft, dnleft, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    uint64_t nbad = 0;
    for (uint64_t l = 0; l < nloci; ++l) if (bad[l] && dir[l].bytes) { dir[l].bytes = 0; ++nbad; }
    if (nbad) HIPCHK(hipMemcpy(c->d_ldir, dir.data(), nloci * sizeof(LocusDir), hipMemcpyHostToDevice));
    HIPCHK(hipFree(debeg)); HIPCHK(hipFree(dekey)); HIPCHK(hipFree(dskey)); HIPCHK(hipFree(depay)); HIPCHK(hipFree(dspay)); HIPCHK(hipFree(dgscr)); HIPCHK(hipFree(dnleft));
    tk("read-back + frees");
    c->loc_left_out = nleft;
    HIPCHK(hipFree(dcnt)); HIPCHK(hipFree(dbad));
    c->limg_bytes = at; c->loc_nimg = nimg - nbad;
    c->T.ldir = c->d_ldir; c->T.limg = c->d_limg;
    if (g->idx_cache_mode == 2 && !g->idx_cache.empty()) write_locus_cache(c, fp, dir);
    if (getenv("DBTK_VERBOSE")) fprintf(stderr, "locus images: %llu of %llu loci, %.1f MB, %llu keys left out\n", (unsigned long long)c->loc_nimg, (unsigned long long)nloci, at / 1e6, (unsigned long long)c->loc_left_out);
    return DBTK_OK;
}

// The same images for the graph table (the lean walk kernel's; dbtk_locus.h: body_gloc_*), from the finished hashed table.  No sidecar.
dbtk_status_t build_graph_images(dbtk_ctx* c) {
    const dbtk_rpgg* g = c->g;
    hipStream_t s = c->stream;
    const uint64_t nloci = g->nloci;
    if (const char* e = getenv("DBTK_LOCUS")) if (!atoi(e)) return DBTK_OK;
    if (!nloci || !c->d_gr || loc_lg_min(g->ksize) > LOC_LG_MAX) return DBTK_OK;
    uint32_t *dcnt = nullptr, *dbad = nullptr;
    HIPCHK(hipMalloc(&dcnt, nloci * 4));
    HIPCHK(hipMalloc(&dbad, nloci * 4));
    HIPCHK(hipMemsetAsync(dcnt, 0, nloci * 4, s));
    HIPCHK(hipMemsetAsync(dbad, 0, nloci * 4, s));
    LocBuildArgs a;
    memset(&a, 0, sizeof(a));
    a.gr = c->d_gr; a.gr_nslots = c->T.gr_mask + 1; a.trbeg = c->d_trbeg; a.nloci = (uint32_t)nloci; a.ksize = g->ksize;
    a.cnt = dcnt; a.bad = dbad;
    LAUNCH(k_gloc_count, dim3(2048), dim3(256), s, a);
    std::vector<uint32_t> cnt(nloci), bad(nloci);
    HIPCHK(hipMemcpyAsync(cnt.data(), dcnt, nloci * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    std::vector<LocusDir> dir(nloci);
    std::vector<uint64_t> ebeg(nloci + 1, 0);
    uint64_t at = 0, nimg = 0;
    for (uint64_t l = 0; l < nloci; ++l) {
        const uint32_t lg = loc_lgnb_for(cnt[l], g->ksize);
        dir[l] = LocusDir{(uint32_t)(at / 16), 0u, lg, (uint32_t)g->out_beg[l]};
        ebeg[l + 1] = ebeg[l];
        if (!cnt[l] || lg > LOC_LG_MAX || cnt[l] > 0xFFF0u || at + loc_image_bytes(lg) > (16ull << 32) || g->out_beg[l + 1] - g->out_beg[l] >= GLOC_SLOT_MAX) continue;
        dir[l].bytes = loc_image_bytes(lg);
        at += dir[l].bytes;
        ebeg[l + 1] += cnt[l];
        ++nimg;
    }
    if (!nimg) { HIPCHK(hipFree(dcnt)); HIPCHK(hipFree(dbad)); return DBTK_OK; }
    const uint64_t nent = ebeg[nloci];
    const uint32_t gstride = 2 * (1u << LOC_LG_MAX) + 2;
    uint64_t *debeg = nullptr, *dekey = nullptr, *dskey = nullptr, *dnleft = nullptr;
    uint32_t *depay = nullptr, *dspay = nullptr;
    uint16_t* dgscr = nullptr;
    HIPCHK(hipMalloc(&c->d_gldir, nloci * sizeof(LocusDir)));
    HIPCHK(hipMalloc(&c->d_glimg, at + 16));
    HIPCHK(hipMalloc(&debeg, (nloci + 1) * 8));
    HIPCHK(hipMalloc(&dekey, (nent + 1) * 8)); HIPCHK(hipMalloc(&dskey, (nent + 1) * 8));
    HIPCHK(hipMalloc(&depay, (nent + 1) * 4)); HIPCHK(hipMalloc(&dspay, (nent + 1) * 4));
    HIPCHK(hipMalloc(&dgscr, nloci * (uint64_t)gstride * 2));
    HIPCHK(hipMalloc(&dnleft, 8));
    HIPCHK(hipMemsetAsync(dnleft, 0, 8, s));
    HIPCHK(hipMemsetAsync(dcnt, 0, nloci * 4, s));
    HIPCHK(hipMemcpyAsync(c->d_gldir, dir.data(), nloci * sizeof(LocusDir), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(debeg, ebeg.data(), (nloci + 1) * 8, hipMemcpyHostToDevice, s));
    a.dir = c->d_gldir; a.arena = c->d_glimg; a.ebeg = debeg; a.ecur = dcnt; a.ekey = dekey; a.epay = depay; a.skey = dskey; a.spay = dspay;
    a.gscr = dgscr; a.gstride = gstride; a.nleft = dnleft;
    LAUNCH(k_gloc_scatter, dim3(2048), dim3(256), s, a);
    { const dbtk_status_t stp = launch_loc_place(a, nloci, s); if (stp) return stp; }
    uint64_t nleft = 0;
    HIPCHK(hipMemcpyAsync(bad.data(), dbad, nloci * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&nleft, dnleft, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    uint64_t nbad = 0;
    for (uint64_t l = 0; l < nloci; ++l) if ((bad[l] || false) && dir[l].bytes) { dir[l].bytes = 0; ++nbad; }
    // (a group left out of a GRAPH image would read as "no node": unlike the index images there is no second look-up behind a miss, so
    // such an image is not used at all)
    if (nleft) { for (uint64_t l = 0; l < nloci; ++l) dir[l].bytes = 0; nbad = nimg; }
    if (nbad) HIPCHK(hipMemcpy(c->d_gldir, dir.data(), nloci * sizeof(LocusDir), hipMemcpyHostToDevice));
    HIPCHK(hipFree(debeg)); HIPCHK(hipFree(dekey)); HIPCHK(hipFree(dskey)); HIPCHK(hipFree(depay)); HIPCHK(hipFree(dspay)); HIPCHK(hipFree(dgscr)); HIPCHK(hipFree(dnleft));
    HIPCHK(hipFree(dcnt)); HIPCHK(hipFree(dbad));
    c->glimg_bytes = at;
    c->T.gldir = c->d_gldir; c->T.glimg = c->d_glimg;
    if (getenv("DBTK_VERBOSE")) fprintf(stderr, "graph images: %llu of %llu loci, %.1f MB\n", (unsigned long long)(nimg - nbad), (unsigned long long)nloci, at / 1e6);
    return DBTK_OK;
}

// Graph table (dbtk_tables.h: GrSlot) from graphDB's flat arrays + the TR k-mers: graph pass, then TR pass.
dbtk_status_t build_graph_table(dbtk_ctx* c) {
    const dbtk_rpgg* g = c->g;
    hipStream_t s = c->stream;
    const uint64_t nloci = g->nloci, ngr = g->gr_ks.size(), ntrf = g->tr_ks.size();
    for (uint64_t l = 0; l < nloci; ++l)
        if (g->out_beg[l + 1] - g->out_beg[l] >= (1ull << (32 - GR_SLOT_SHIFT))) { set_error("a locus has more than 2^21 TR k-mers: graph table slot field too small"); return DBTK_ERR_UNSUPPORTED; }
    const uint64_t cap = pow2_at_least(ngr + ngr / 2 + 2 * ntrf + 2);  // > the entries whatever the files hold; the two strands of a node share one, so the load is ~0.2-0.35
    HIPCHK(hipMalloc(&c->d_gr, cap * sizeof(GrSlot)));
    c->tb_gr = cap * sizeof(GrSlot);
    HIPCHK(hipMemsetAsync(c->d_gr, 0xFF, cap * sizeof(GrSlot), s));
    std::vector<uint64_t> beg(nloci + 1, 0);
    uint64_t *dks = nullptr, *dbeg = nullptr, *dslot = nullptr, *dn = nullptr;
    uint8_t* dms = nullptr;
    const uint64_t nmax = ngr > ntrf ? ngr : ntrf;
    HIPCHK(hipMalloc(&dks, (nmax + 1) * 8));
    HIPCHK(hipMalloc(&dms, ngr + 1));
    HIPCHK(hipMalloc(&dslot, (ntrf + 1) * 8));
    HIPCHK(hipMalloc(&dbeg, (nloci + 1) * 8));
    HIPCHK(hipMalloc(&dn, 8));
    HIPCHK(hipMemsetAsync(dn, 0, 8, s));
    GrBuildArgs a{c->d_gr, cap - 1, 64 - log2u(cap), g->ksize, dks, dms, dbeg, (uint32_t)nloci, nullptr, c->d_trbeg, ngr, dn};
    if (ngr) {
        for (uint64_t l = 0; l < nloci; ++l) beg[l + 1] = beg[l] + g->gr_cnt[l];
        HIPCHK(hipMemcpyAsync(dks, g->gr_ks.data(), ngr * 8, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(dms, g->gr_ms.data(), ngr, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(dbeg, beg.data(), (nloci + 1) * 8, hipMemcpyHostToDevice, s));
        LAUNCH(k_gr_insert, dim3(2048), dim3(256), s, a);
        HIPCHK(hipStreamSynchronize(s));
    }
    if (ntrf) {
        for (uint64_t l = 0; l < nloci; ++l) beg[l + 1] = beg[l] + g->tr_cnt[l];
        HIPCHK(hipMemcpyAsync(dks, g->tr_ks.data(), ntrf * 8, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(dslot, g->out_slot.data(), ntrf * 8, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(dbeg, beg.data(), (nloci + 1) * 8, hipMemcpyHostToDevice, s));
        a.ms = nullptr; a.outslot = dslot; a.n = ntrf;
        LAUNCH(k_gr_insert, dim3(2048), dim3(256), s, a);
        HIPCHK(hipStreamSynchronize(s));
    }
    HIPCHK(hipGetLastError());
    uint64_t nent = 0;  // entries of the table: the two strands of a node share one, most TR k-mers are nodes already
    HIPCHK(hipMemcpy(&nent, dn, 8, hipMemcpyDeviceToHost));
    if (nent == 0 || nent > ngr + ntrf) nent = ngr + ntrf;
    HIPCHK(hipFree(dks)); HIPCHK(hipFree(dms)); HIPCHK(hipFree(dslot)); HIPCHK(hipFree(dbeg)); HIPCHK(hipFree(dn));
    c->T.gr = c->d_gr; c->T.gr_mask = cap - 1; c->T.gr_shift = 64 - log2u(cap);
    {   // the lean walk kernel's minimizer-grouped copy of the table, for the values of k its bucket form exists for (DBTK_MZ=0: do without)
        bool on = true;
        if (const char* e = getenv("DBTK_MZ")) on = atoi(e) != 0;
        const uint32_t m = mz_m_for_k(g->ksize);
        if (on && m) {
            uint64_t gper = 3;  // buckets per 8 entries, as the index's copy (DBTK_GRMZ_SPARSITY; 6 was 17 GB at release scale for +0.6 % on the global lean walk kernel, which only takes what the locus-resident one leaves)
            if (const char* e = getenv("DBTK_GRMZ_SPARSITY")) { const long v = atol(e); if (v >= 1 && v <= 64) gper = (uint64_t)v; }
            uint64_t nb = pow2_at_least(nent * gper / 8 + 8);
            if (nb > (1ull << 28)) nb = 1ull << 28;
            HIPCHK(hipMalloc(&c->d_grmz, nb * sizeof(MzBucket)));
            c->tb_grmz = nb * sizeof(MzBucket);
            LAUNCH(k_mz_fill, dim3(2048), dim3(256), s, reinterpret_cast<uint64_t*>(c->d_grmz), nb * 16, 1);
            GrMzBuildArgs ga{c->d_gr, cap, c->d_grmz, (uint32_t)(nb - 1), g->ksize, m};
            LAUNCH(k_grmz_insert, dim3(2048), dim3(256), s, ga);
            HIPCHK(hipStreamSynchronize(s));
            c->T.grmz = c->d_grmz; c->T.grmz_mask = nb - 1;
        }
    }
    return build_graph_images(c);
}

// (key, locus) -> value table from per-locus arrays (tre edges: value unused; bait: min << 8 | max)
dbtk_status_t build_kl_table(dbtk_ctx* c, const std::vector<uint64_t>& cnt, const std::vector<uint64_t>& ks,
                             const std::vector<uint16_t>* vals, ClsSlot** out, uint64_t* mask, uint32_t* shift) {
    hipStream_t s = c->stream;
    const uint64_t nloci = c->g->nloci, n = ks.size();
    const uint64_t cap = pow2_at_least(2 * n + 2);
    HIPCHK(hipMalloc(out, cap * sizeof(ClsSlot)));
    HIPCHK(hipMemsetAsync(*out, 0xFF, cap * sizeof(ClsSlot), s));
    *mask = cap - 1;
    *shift = 64 - log2u(cap);
    if (!n) { HIPCHK(hipStreamSynchronize(s)); return DBTK_OK; }
    std::vector<uint64_t> beg(nloci + 1, 0), v64(n, 0);
    for (uint64_t l = 0; l < nloci; ++l) beg[l + 1] = beg[l] + cnt[l];
    if (vals) for (uint64_t i = 0; i < n; ++i) v64[i] = (*vals)[i];
    uint64_t *dks = nullptr, *dbeg = nullptr, *dval = nullptr, *dn = nullptr;
    HIPCHK(hipMalloc(&dks, n * 8));
    HIPCHK(hipMalloc(&dval, n * 8));
    HIPCHK(hipMalloc(&dbeg, (nloci + 1) * 8));
    HIPCHK(hipMalloc(&dn, 8));
    HIPCHK(hipMemsetAsync(dn, 0, 8, s));
    HIPCHK(hipMemcpyAsync(dks, ks.data(), n * 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dval, v64.data(), n * 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dbeg, beg.data(), (nloci + 1) * 8, hipMemcpyHostToDevice, s));
    ClsBuildArgs a{*out, cap - 1, 64 - log2u(cap), dks, dbeg, (uint32_t)nloci, dval, n, dn};
    LAUNCH(k_cls_insert, dim3(1024), dim3(256), s, a);
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipFree(dks)); HIPCHK(hipFree(dval)); HIPCHK(hipFree(dbeg)); HIPCHK(hipFree(dn));
    return DBTK_OK;
}

template <class T>
dbtk_status_t ensure(T** p, uint64_t* cap, uint64_t need) {
    if (need <= *cap && *p) return DBTK_OK;
    if (*p) HIPCHK(hipFree(*p));
    *p = nullptr;
    const uint64_t ncap = need + need / 4 + 64;
    HIPCHK(hipMalloc(p, ncap * sizeof(T)));
    *cap = ncap;
    return DBTK_OK;
}

// fold the recorded event pairs of one kernel into its totals (stream must be idle)
void fold_timer(Timed& t) {
    for (int j = 0; j < t.used; ++j) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.beg[j], t.end[j]) == hipSuccess) { t.total_ms += ms; ++t.launches; }
    }
    t.used = 0;
}
void switch_lane(dbtk_ctx* c) {
    std::swap(c->stream, c->alt.stream); std::swap(c->d_small, c->alt.d_small);
    std::swap(c->d_surv, c->alt.d_surv); std::swap(c->surv_cap, c->alt.surv_cap); std::swap(c->d_sorted, c->alt.d_sorted);
    std::swap(c->d_hitva, c->alt.d_hitva); std::swap(c->hitva_cap, c->alt.hitva_cap);
    std::swap(c->d_hitnk, c->alt.d_hitnk); std::swap(c->hitnk_cap, c->alt.hitnk_cap);
    std::swap(c->d_hitoff, c->alt.d_hitoff); std::swap(c->hitoff_cap, c->alt.hitoff_cap);
    std::swap(c->d_gen, c->alt.d_gen); std::swap(c->gen_cap, c->alt.gen_cap);
    std::swap(c->d_tickets, c->alt.d_tickets); std::swap(c->tickets_cap, c->alt.tickets_cap);
    std::swap(c->d_walk, c->alt.d_walk); std::swap(c->walk_cap, c->alt.walk_cap);
    std::swap(c->d_vote, c->alt.d_vote); std::swap(c->d_epoch, c->alt.d_epoch);
    std::swap(c->m_flat, c->alt.m_flat); std::swap(c->m_flat_cap, c->alt.m_flat_cap); std::swap(c->m_off, c->alt.m_off); std::swap(c->m_off_cap, c->alt.m_off_cap);
    std::swap(c->m_bytes, c->alt.m_bytes); std::swap(c->m_pairs, c->alt.m_pairs); std::swap(c->m_maxlen, c->alt.m_maxlen);
    if (!c->parked.empty()) {  // round robin: the lane just left goes to the back of the queue, the longest-parked one is next
        c->parked.push_back(c->alt);
        c->alt = c->parked.front();
        c->parked.pop_front();
    }
}
// Every reader of the accumulators comes through here (counts, the all-reduce, reset, the caller of dbtk_ctx_accum_buffer after
// dbtk_ctx_synchronize): the counter replicas are folded into the counters once, now, instead of at the end of every batch (a launch of
// one wave + its gap: 14 us of a 1.2-ms step).
hipError_t sync_all(dbtk_ctx* c) {
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && c->alt.stream) e = hipStreamSynchronize(c->alt.stream);
    for (auto& l : c->parked) if (e == hipSuccess && l.stream) e = hipStreamSynchronize(l.stream);
    if (e == hipSuccess && c->fold_pending && c->d_ctr && c->d_accum) {
        hipLaunchKernelGGL(k_fold_counters, dim3(1), dim3(64), 0, c->stream, c->d_accum + c->ntr + 2 * (uint64_t)c->g->nloci, c->d_ctr, c->d_pstats);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) c->fold_pending = false;
    }
    return e;
}
dbtk_status_t timed_slot(dbtk_ctx* c, int k, int* slot) {
    Timed& t = c->timed[k];
    if (t.used == EVPOOL) {
        HIPCHK(sync_all(c));
        for (int i = 0; i < NKERN; ++i) fold_timer(c->timed[i]);
    }
    *slot = t.used++;
    return DBTK_OK;
}

// K1 over the whole batch, then K2 -> K3 over chunks of the survivor list (the K2 -> K3 hit
// buffer holds SURV_CAP pairs).  How many survivors there are is known only on the device, so
// ceil(npairs / SURV_CAP) chunk iterations are enqueued and the kernels of a chunk past the end
// of the list exit at once; nothing waits for the host.
constexpr uint64_t SURV_CAP = 1ull << 23;  // at most 51 GB of hit buffers at 150 bp (allocated for the batch size actually seen): one chunk for batches of up to 8 M pairs
constexpr uint32_t SMALL_WORDS = 64;  // d_small: nsurv, (unused), nrec, errflag (sticky until reported); stamps at +32

dbtk_status_t launch_batch(dbtk_ctx* c, const uint8_t* d_seq, const uint64_t* d_off, uint64_t seq_len, uint64_t npairs,
                           uint32_t max_read_len, dbtk_pair_rec_t* d_recs, uint32_t rec_cap, const uint8_t* d_qual = nullptr,
                           dbtk_thread_rec_t* walk_trecs = nullptr, bool walk_aln = false, bool walk_txt = false) {
    hipStream_t s = c->stream;
    if (npairs >= 0xFFFFFFFFull) { set_error("batch too large (pair index is 32-bit)"); return DBTK_ERR_ARG; }
    if (npairs == 0) return DBTK_OK;
    const uint32_t k = c->g->ksize;
    const uint32_t nkmax = max_read_len >= k ? max_read_len - k + 1 : 1;
    const uint32_t nkp = 64 * ((nkmax + 63) / 64);
    const uint32_t ns = nkp / 64;
    uint64_t cap = SURV_CAP;
    if (const char* e = getenv("DBTK_SURV_CAP")) { const long long v = atoll(e); if (v > 0) cap = (uint64_t)v; }  // (tests: force several chunks)
    const uint64_t tcap = npairs < cap ? npairs : cap;
    const uint64_t nchunks = (npairs + tcap - 1) / tcap;
    const uint64_t nloci = c->g->nloci;
    // behind the survivor lists: the work items of the locus-resident probe kernel (two classes of workgroup) and the list of the
    // pairs it leaves to the global-table kernel (dbtk_locus.h)
    const uint64_t surv_words = (3 * (npairs + 1) + nloci + 2 + SCAN_BLOCKS + 3) & ~3ull;
    const uint64_t item_cap = npairs / LOC_CH + nloci + 2;  // (of a chunk for the probe kernel, of the whole list for the walk's)
    uint32_t split_blk = 0;  // most workgroups any locus-resident launch has: three lists of workgroup ranges behind the rest list
    for (int q = 0; q < 6; ++q) split_blk = std::max(split_blk, (uint32_t)std::max(std::max(c->loc_blocks[q], c->locf_blocks[q]), c->wfl_blocks[q]));
    const uint64_t split_at = surv_words + 3 * 4 * item_cap + npairs + 4;
    dbtk_status_t st = ensure(&c->d_surv, &c->surv_cap, split_at + 3 * ((uint64_t)split_blk + 1));
    if (st) return st;
    if ((st = ensure(&c->d_hitoff, &c->hitoff_cap, tcap * 4))) return st;  // offsets, then headers
    if ((st = ensure(&c->d_hitva, &c->hitva_cap, tcap * 2 * nkp))) return st;
    if ((st = ensure(&c->d_hitnk, &c->hitnk_cap, tcap * 2))) return st;
    if ((st = ensure(&c->d_gen, &c->gen_cap, tcap))) return st;
    // per chunk: a ticket counter and a passed-on counter — for up to three chunks in the TK_INLINE words in front of d_small, so that they
    // and nsurv / novf / nrec are zeroed by ONE memset
    const bool tk_inline = 2 * (nchunks + 1) <= TK_INLINE;
    if (!tk_inline && (st = ensure(&c->d_tickets, &c->tickets_cap, 2 * (nchunks + 1)))) return st;
    uint32_t* const tickets = tk_inline ? c->d_small - 2 * (nchunks + 1) : c->d_tickets;
    const bool walking = c->P.threading == DBTK_THREADING_V13;
    uint64_t slow_cap = 0, info_rows = 0;
    if (walking) {
        // destLocus | return codes | the fast kernel's passed-on list (its waves reserve WF_R places at a time) | the graph info of the
        // passed-on pairs' positions (rows for an eighth of the pairs — with text records, where every pair with a dirty mate is passed
        // on, for half of them; a pair beyond that is looked up again by the other kernel)
        uint64_t wf_waves = 4ull * (uint64_t)c->walkfast_blocks;
        for (int q = 0; q < 3; ++q) wf_waves += (uint64_t)std::max(c->wfl_blocks[q], c->wfl_blocks[3 + q]) * (q == 0 ? LOC_NW_XS : q == 1 ? LOC_NW_S : LOC_NW_L);
        slow_cap = npairs + WF_R * wf_waves + 8;
        info_rows = std::min<uint64_t>(slow_cap, std::max<uint64_t>(walk_txt ? npairs / 2 : npairs / 8, 4096));
        if (const char* e = getenv("DBTK_WALK_INFO_ROWS")) info_rows = std::min<uint64_t>(slow_cap, (uint64_t)std::max<long>(atol(e), 0));  // diagnostic / tests: 0 = none, a few = both ways in one batch
        if ((st = ensure(&c->d_walk, &c->walk_cap, 2 * npairs + slow_cap + info_rows * 2 * 160))) return st;
        HIPCHK(hipMemsetAsync(c->d_walk, 0xFF, npairs * sizeof(uint32_t), s));  // NAN32: the pair does not reach threading
        HIPCHK(hipMemsetAsync(c->d_walk + npairs, 0, npairs * sizeof(uint32_t), s));  // walk_ret: no pair is marked WALK_PENDING
    }
    // nsurv, novf, nrec; the error word (3) stays until it has been reported
    if (tk_inline) HIPCHK(hipMemsetAsync(tickets, 0, (2 * (nchunks + 1) + 3) * sizeof(uint32_t), s));
    else {
        HIPCHK(hipMemsetAsync(c->d_small, 0, 3 * sizeof(uint32_t), s));
        HIPCHK(hipMemsetAsync(tickets, 0, 2 * (nchunks + 1) * sizeof(uint32_t), s));
    }
    if (c->P.bubbles) {
        if ((st = ensure(&c->d_edge, &c->edge_cap, tcap * 2 * nkp))) return st;
        // every position of every kept mate could be novel; bounded so that the log stays < 6.4 GB
        const uint64_t ecap = std::min<uint64_t>(npairs * 2 * nkmax, 1ull << 28);
        if ((st = ensure(&c->d_events, &c->events_cap, ecap))) return st;
        HIPCHK(hipMemsetAsync(c->d_nevents, 0, sizeof(uint32_t), s));
    }
    if (c->P.bait && d_qual && (st = ensure(&c->d_qmask, &c->qmask_cap, tcap * 2 * 4))) return st;
    BatchArgs a;
    memset(&a, 0, sizeof(a));
    { static const bool k1x = !(getenv("DBTK_K1_XCD") && atoi(getenv("DBTK_K1_XCD")) == 0); a.k1_xcd = k1x ? 1u : 0u; }
    a.T = c->T; a.P = c->P; a.P.aln &= 3u;
    a.seq = d_seq; a.off = d_off; a.seq_len = seq_len; a.npairs = npairs;
    a.surv = c->d_surv; a.nsurv = c->d_small + 0; a.nrec = c->d_small + 2; a.errflag = c->d_small + 3;
    a.counts = c->d_accum;
    a.kmc = c->d_accum + c->ntr;
    a.nmapread = a.kmc + c->g->nloci;
    a.counters = a.nmapread + c->g->nloci;
    a.ctr_rep = c->d_ctr;
    a.pstats = c->d_pstats;
    a.sortflag = c->d_small + 6; a.hint_out = c->h_sortflag;
    a.recs = d_recs; a.rec_cap = rec_cap;
    a.vote_scratch = c->d_vote; a.vote_epoch = c->d_epoch; a.vote_rows = (uint32_t)c->vote_rows;
    a.vote_busy = reinterpret_cast<uint64_t*>(c->d_epoch + ((c->vote_rows + 1) & ~1));  // (behind the epochs, 8-byte aligned)
    a.walk_dst = walking ? c->d_walk : nullptr;
    a.hitaux = reinterpret_cast<uint32_t*>(c->d_hitva); a.hitval = a.hitaux + tcap * 2 * nkp; a.hitnk = c->d_hitnk; a.hitoff = c->d_hitoff; a.hithdr = c->d_hitoff + tcap * 2; a.nkp = nkp; a.pair_base = 0; a.tcap = (uint32_t)tcap;
    // the usual-pair kernel takes the pairs it can finish and passes the rest on; it needs the class of a k-mer next to its
    // index value (consistent RPGG) and does not do the trace, bait or bubble work
    const bool usual = c->T.consistent && !c->P.trace && !c->P.bait && !c->P.bubbles;
    // ... and with no record buffer the locus-resident probe kernel resolves the usual pairs of its items itself (dbtk_locus.h: FUSE;
    // DBTK_FUSE=0: the two-kernel form, for measurements)
    // (DBTK_FUSE: bit 0 the locus-resident kernel's fusion, bit 1 the lean kernel's; default both)
    static const int fuse_on = [] { const char* e = getenv("DBTK_FUSE"); return e ? atoi(e) : 3; }();
    const bool fuse = usual && !d_recs && (fuse_on & 1);
    const bool fuse_lean = usual && !d_recs && (fuse_on & 2);
    if (c->P.bubbles) { a.edgebuf = c->d_edge; a.events = c->d_events; a.nevents = c->d_nevents; a.events_cap = (uint32_t)std::min<uint64_t>(c->events_cap, 0xFFFFFFFFull); }
    if (c->P.bait && d_qual) { a.qual = d_qual; a.qmaskbuf = c->d_qmask; }
#ifdef DBTK_STAMPS
    a.dbg = reinterpret_cast<uint64_t*>(c->d_small + 32);
#endif
    const uint64_t ntiles = (npairs + K1_TP - 1) / K1_TP;
    int e = 0;
    const bool tm = c->timers_on && (c->batch_no++ % c->timers_every) == 0;
    auto rec_beg = [&](int kslot) -> dbtk_status_t { if (!tm) return DBTK_OK; dbtk_status_t r = timed_slot(c, kslot, &e); if (r) return r; HIPCHK(hipEventRecord(c->timed[kslot].beg[e], s)); return DBTK_OK; };
    auto rec_end = [&](int kslot) -> dbtk_status_t { if (tm) HIPCHK(hipEventRecord(c->timed[kslot].end[e], s)); return DBTK_OK; };
    // The survivor list in locus order (dbtk_probe2.h: body_surv_*): what every later kernel indexes.  A batch with few survivors per locus (a
    // WGS batch) is not sorted — the four kernels then only find that out and copy the list — so when the batch BEFORE had fewer than half
    // the survivors from which a list is sorted, they are not launched at all and the encode stage's own list is used (20 us of a 1.2-ms
    // step).  A hint like the one below: the order of the list is never a matter of results.
    bool sort_hint = true, locus_hint = true;
    uint32_t surv_hint = 0xFFFFFFFFu;  // pairs the lean probe kernel took in the batch before (no hint: "many")
    if (c->h_sortflag) {  // (both hints from ONE reading of the pinned words: the copy of the batch before may land at any time)
        volatile uint32_t* hh = c->h_sortflag;
        const uint32_t prev_surv = hh[0], prev_flag = hh[6];
        surv_hint = hh[1];  // (pairs the lean probe kernel took)
        sort_hint = prev_flag != 0 || 2 * (uint64_t)prev_surv >= (uint64_t)SORT_MIN_PER_LOCUS * nloci;
        locus_hint = prev_flag != 0 && (uint64_t)prev_surv >= (uint64_t)LOC_MIN_PAIRS * nloci;
    }
    {   // the encode stage
    const uint32_t g1 = (uint32_t)(ntiles < (uint64_t)c->k1_blocks ? ntiles : (uint64_t)c->k1_blocks);  // resident waves
    // (the sort key of every survivor comes with it when the list is going to be sorted: the encode stage has the table line of the pair's
    // first sampled k-mer in hand, body_surv_key would look it up again — 12 M requests per 10 M all-hit reads.  Its one-sample-per-turn
    // form only: NM = 1 with the presence filter, the default.)
    const bool k1_keys = sort_hint && c->P.n_filter && c->P.nm_filter == 1 && c->T.flt && !getenv("DBTK_NO_K1_KEYS");
    a.skey = k1_keys ? c->d_surv + 2 * (npairs + 1) : nullptr;  // (= SurvSortArgs::key below)
    bool k1_lazy = false;
    {   // (lazy sampling when more than half of the batch before passed subfilter — a hint like the others; DBTK_K1_LAZY=0 / 1: never / always)
        static const int lazy_env = getenv("DBTK_K1_LAZY") ? atoi(getenv("DBTK_K1_LAZY")) : -1;
        bool lazy = lazy_env == 1;
        if (lazy_env < 0 && c->h_sortflag) { const uint32_t prev_surv = ((volatile uint32_t*)c->h_sortflag)[0]; lazy = prev_surv != 0xFFFFFFFFu && 2 * (uint64_t)prev_surv >= npairs; }
        k1_lazy = lazy;
    }
    if ((st = rec_beg(0))) return st;
    if (k1_lazy) LAUNCH(k_encode_subfilter_lazy, dim3(g1), dim3(K1_NT), s, a);
    else LAUNCH(k_encode_subfilter, dim3(g1), dim3(K1_NT), s, a);
    if ((st = rec_end(0))) return st;
    }
    if (sort_hint) {
        SurvSortArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.T = c->T; sa.P = c->P; sa.seq = d_seq; sa.off = d_off; sa.surv = c->d_surv; sa.nsurv = c->d_small + 0;
        sa.sorted = c->d_surv + (npairs + 1); sa.key = sa.sorted + (npairs + 1); sa.hist = sa.key + (npairs + 1); sa.flag = c->d_small + 6;
        sa.sort_min = c->h_sortflag ? SORT_MIN_PER_LOCUS : 0u;  // (no hint word = DBTK_LOCUS_ALWAYS: every batch sorted, every batch through the locus path)
        sa.have_keys = a.skey ? 1u : 0u;
        // (two arrays that live only between these kernels, in regions the locus lists take over afterwards: the items' and the rest list's)
        sa.starts = c->d_surv + surv_words; sa.rank = c->d_surv + surv_words + 3 * 4 * item_cap;
        HIPCHK(hipMemsetAsync(sa.hist, 0, (nloci + 2) * sizeof(uint32_t), s));
        const uint32_t gs = (uint32_t)std::min<uint64_t>((npairs + 255) / 256, (uint64_t)c->num_cu * 8);
        if ((st = rec_beg(5))) return st;
        LAUNCH(k_surv_key, dim3(gs), dim3(256), s, sa);
        LAUNCH(k_surv_scan, dim3(SCAN_BLOCKS), dim3(64), s, sa, 0);
        LAUNCH(k_surv_scan, dim3(SCAN_BLOCKS), dim3(64), s, sa, 1);
        LAUNCH(k_surv_scatter, dim3(gs), dim3(256), s, sa);
        if ((st = rec_end(5))) return st;
        c->d_sorted = sa.sorted;
        a.surv = sa.sorted;
    } else c->d_sorted = c->d_surv;  // (the flag says "not in locus order": the encode stage cleared it)
    // Does the locus-resident probe kernel have anything to do?  Only in a batch with many survivors per locus (list in locus order, loci
    // with LOC_MIN_PAIRS pairs and more); a WGS-like batch (one survivor per locus) would pay its empty launches for nothing (40 us on a
    // 1.2-ms step).  Which kind a batch is, is known on the device only — so the survivor count and the sort flag of the batch BEFORE
    // come back through pinned words and decide for this one.  A hint, never a matter of results: the lean kernel looks up whatever
    // the locus path does not take.
    // (the words are written by the general resolve kernel, BatchArgs::hint_out: pinned memory is the device's to write)
    bool lean_fused = false;
    for (uint64_t ch = 0; ch < nchunks; ++ch) {
        a.t0 = (uint32_t)(ch * tcap);
        if (tm) { if ((st = timed_slot(c, 1, &e))) return st; HIPCHK(hipEventRecord(c->timed[1].beg[e], s)); }
        {
            // the lean form (dbtk_probe2.h) where its conditions hold: the tables exist for this k, no read longer than its
            // lanes cover, none of the optional per-read extras (-bu edges, -b quality masks); otherwise the general form
            const dim3 gpr(c->num_cu * c->probe_wpc);
            const uint32_t wn = a.T.mz ? k - a.T.mz_m + 1 : 0;
            const int npl = !a.T.mz || a.edgebuf || a.qmaskbuf || (wn != 7 && wn != 11) ? 0 : (max_read_len <= 32 * 3 + a.T.mz_m - 1 ? 3 : max_read_len <= 32 * 5 + a.T.mz_m - 1 ? 5 : 0);
            a.sel = nullptr; a.nsel = nullptr;
            // (the general resolve kernel's list: the usual-pair kernel appends to it — and, fused, the locus-resident probe kernel)
            a.gen_list = usual ? c->d_gen : nullptr; a.ngen = usual ? tickets + (nchunks + 1) + ch : nullptr;
            if (npl && a.T.ldir && locus_hint) {
                // The pairs of loci that have an image: the locus-resident kernel (dbtk_locus.h), image in LDS, one item = one locus'
                // next LOC_CH pairs of the list; the lean kernel then takes what is left (loci without an image, pairs without a
                // locus — and everything when the batch has too few survivors per locus for the list to be in locus order).
                LocItemArgs ia;
                memset(&ia, 0, sizeof(ia));
                ia.hist = c->d_surv + 3 * (npairs + 1); ia.nsurv = c->d_small + 0; ia.flag = c->d_small + 6; ia.dir = a.T.ldir;
                ia.nloci = (uint32_t)nloci; ia.t0 = a.t0; ia.tcap = (uint32_t)tcap;
                static const int loc_classes = [] { const char* e = getenv("DBTK_LOC_CLASSES"); return e ? atoi(e) : 7; }();  // diagnostic: bit c = class c in use
                ia.cap_bytes[0] = (loc_classes & 1) ? LOC_IMGB_XS : 0; ia.cap_bytes[1] = (loc_classes & 2) ? LOC_IMGB_S : 0; ia.cap_bytes[2] = (loc_classes & 4) ? LOC_IMGB_L : 0;
                for (int q = 0; q < 3; ++q) ia.items[q] = reinterpret_cast<uint4*>(c->d_surv + surv_words) + q * item_cap;
                ia.nitems = c->d_small + 8; ia.item_cap = (uint32_t)item_cap;
                ia.rest = c->d_surv + surv_words + 3 * 4 * item_cap;
                HIPCHK(hipMemsetAsync(c->d_small + 8, 0, 4 * sizeof(uint32_t), s));
                LAUNCH(k_loc_items, dim3((uint32_t)((nloci + 255) / 256)), dim3(256), s, ia);
                LAUNCH(k_loc_rest, dim3((uint32_t)std::min<uint64_t>((tcap + 255) / 256, (uint64_t)c->num_cu * 8)), dim3(256), s, ia);
                LocSplitArgs sp;
                memset(&sp, 0, sizeof(sp));
                for (int q = 0; q < 3; ++q) {
                    sp.items[q] = ia.items[q]; sp.starts[q] = c->d_surv + split_at + q * ((uint64_t)split_blk + 1);
                    sp.nblk[q] = (uint32_t)(fuse ? c->locf_blocks : c->loc_blocks)[(npl == 3 ? 0 : 3) + q];
                    sp.wfix[q] = 4u << q;  // (an image of the class costs about as much as that many pairs)
                }
                sp.nitems = c->d_small + 8; sp.item_cap = (uint32_t)item_cap; sp.stats = c->d_pstats;
                LAUNCH(k_loc_split, dim3(3), dim3(1024), s, sp);
                LocRunArgs r0{a.T.ldir, a.T.limg, ia.items[0], c->d_small + 8, ia.rest, c->d_small + 11, sp.starts[0]}, r1{a.T.ldir, a.T.limg, ia.items[1], c->d_small + 9, ia.rest, c->d_small + 11, sp.starts[1]},
                    r2{a.T.ldir, a.T.limg, ia.items[2], c->d_small + 10, ia.rest, c->d_small + 11, sp.starts[2]};
                if (fuse && npl == 3) {
                    LAUNCH((k_probe_locus<3, LOC_NW_XS, LOC_IMGB_XS, true>), dim3(c->locf_blocks[0]), dim3(LOC_NW_XS * 64), s, a, r0);
                    LAUNCH((k_probe_locus<3, LOC_NW_S, LOC_IMGB_S, true>), dim3(c->locf_blocks[1]), dim3(LOC_NW_S * 64), s, a, r1);
                    LAUNCH((k_probe_locus<3, LOC_NW_L, LOC_IMGB_L, true>), dim3(c->locf_blocks[2]), dim3(LOC_NW_L * 64), s, a, r2);
                } else if (fuse) {
                    LAUNCH((k_probe_locus<5, LOC_NW_XS, LOC_IMGB_XS, true>), dim3(c->locf_blocks[3]), dim3(LOC_NW_XS * 64), s, a, r0);
                    LAUNCH((k_probe_locus<5, LOC_NW_S, LOC_IMGB_S, true>), dim3(c->locf_blocks[4]), dim3(LOC_NW_S * 64), s, a, r1);
                    LAUNCH((k_probe_locus<5, LOC_NW_L, LOC_IMGB_L, true>), dim3(c->locf_blocks[5]), dim3(LOC_NW_L * 64), s, a, r2);
                } else if (npl == 3) {
                    LAUNCH((k_probe_locus<3, LOC_NW_XS, LOC_IMGB_XS>), dim3(c->loc_blocks[0]), dim3(LOC_NW_XS * 64), s, a, r0);
                    LAUNCH((k_probe_locus<3, LOC_NW_S, LOC_IMGB_S>), dim3(c->loc_blocks[1]), dim3(LOC_NW_S * 64), s, a, r1);
                    LAUNCH((k_probe_locus<3, LOC_NW_L, LOC_IMGB_L>), dim3(c->loc_blocks[2]), dim3(LOC_NW_L * 64), s, a, r2);
                } else {
                    LAUNCH((k_probe_locus<5, LOC_NW_XS, LOC_IMGB_XS>), dim3(c->loc_blocks[3]), dim3(LOC_NW_XS * 64), s, a, r0);
                    LAUNCH((k_probe_locus<5, LOC_NW_S, LOC_IMGB_S>), dim3(c->loc_blocks[4]), dim3(LOC_NW_S * 64), s, a, r1);
                    LAUNCH((k_probe_locus<5, LOC_NW_L, LOC_IMGB_L>), dim3(c->loc_blocks[5]), dim3(LOC_NW_L * 64), s, a, r2);
                }
                a.sel = ia.rest; a.nsel = c->d_small + 11;
            }
            // (a wave of the lean form works through one contiguous range of the list: as many waves as are resident at once)
            // four forms: with / without the list the locus path left (SEL), resolving the pairs its look-ups decide itself or not (FUSE)
            lean_fused = fuse_lean && npl != 0 && (!a.sel || fuse);  // (pairs an UNFUSED locus-resident kernel took wait in their rows for the usual-pair kernel)
            if (npl) {
                const int vi = (npl == 3 ? 0 : 2) + (wn == 7 ? 0 : 1);
                // blocks: eight rounds of the resident waves for a batch that leaves this kernel many pairs (see dbtk_ctx_create) — but a WGS-like
                // batch has 100 000 survivors, three per block then, and what a block does once (its cache, its share of the counters, its entry in
                // the general kernel's list) is then most of it (fused form, 10 M reads: 0.93 ms with 128 blocks per CU, 0.35 with 16).  So: whole
                // rounds of the resident waves, ~32 pairs per block, by what the kernel took in the batch before (a hint, like the others: never
                // a matter of results).
                const int wpc_max = (lean_fused ? c->probe2f_wpc : c->probe2_wpc)[vi];
                int wpc = wpc_max;
                static const bool wpc_forced = getenv("DBTK_PROBE_WPC") != nullptr;
                if (!wpc_forced) {
                    const uint64_t resident = std::max<uint64_t>((uint64_t)wpc_max / 8, 1), per_round = resident * c->num_cu * 32;
                    wpc = (int)(resident * std::min<uint64_t>(8, std::max<uint64_t>(1, ((uint64_t)surv_hint + per_round - 1) / per_round)));
                }
                const dim3 g2((uint32_t)c->num_cu * (uint32_t)wpc);
#define DBTK_P2_LAUNCH(NPL_, WN_) \
                do { if (a.sel && lean_fused) LAUNCH((k_probe<NPL_, WN_, true, true>), g2, dim3(64), s, a); \
                     else if (a.sel) LAUNCH((k_probe<NPL_, WN_, true, false>), g2, dim3(64), s, a); \
                     else if (lean_fused) LAUNCH((k_probe<NPL_, WN_, false, true>), g2, dim3(64), s, a); \
                     else LAUNCH((k_probe<NPL_, WN_, false, false>), g2, dim3(64), s, a); } while (0)
                if (vi == 0) DBTK_P2_LAUNCH(3, 7); else if (vi == 1) DBTK_P2_LAUNCH(3, 11); else if (vi == 2) DBTK_P2_LAUNCH(5, 7); else DBTK_P2_LAUNCH(5, 11);
#undef DBTK_P2_LAUNCH
            }
            else if (ns <= 2) LAUNCH((k_probe_general<2>), gpr, dim3(64), s, a);
            else if (ns == 3) LAUNCH((k_probe_general<3>), gpr, dim3(64), s, a);
            else LAUNCH((k_probe_general<4>), gpr, dim3(64), s, a);
        }
        if (tm) HIPCHK(hipEventRecord(c->timed[1].end[e], s));
        // RECS = false: no record buffer (-ka without -e): record emission is compiled out
        const int nsi = ns <= 2 ? 0 : (ns == 3 ? 1 : 2);
        if (usual && !lean_fused) {  // (the fused lean probe kernel has resolved what this kernel would, and filled the general kernel's list)
            const dim3 gu(c->usual_blocks[nsi]);
            if (tm) { if ((st = timed_slot(c, 2, &e))) return st; HIPCHK(hipEventRecord(c->timed[2].beg[e], s)); }
            if (d_recs) {
                if (nsi == 0) LAUNCH((k_pair_usual<2, true>), gu, dim3(64), s, a);
                else if (nsi == 1) LAUNCH((k_pair_usual<3, true>), gu, dim3(64), s, a);
                else LAUNCH((k_pair_usual<4, true>), gu, dim3(64), s, a);
            } else if (fuse && a.sel) {  // (the locus-resident kernel has resolved its pairs itself: what is left is the list it did not take)
                if (nsi == 0) LAUNCH((k_pair_usual<2, false, true>), gu, dim3(64), s, a);
                else if (nsi == 1) LAUNCH((k_pair_usual<3, false, true>), gu, dim3(64), s, a);
                else LAUNCH((k_pair_usual<4, false, true>), gu, dim3(64), s, a);
            } else {
                if (nsi == 0) LAUNCH((k_pair_usual<2, false>), gu, dim3(64), s, a);
                else if (nsi == 1) LAUNCH((k_pair_usual<3, false>), gu, dim3(64), s, a);
                else LAUNCH((k_pair_usual<4, false>), gu, dim3(64), s, a);
            }
            if (tm) HIPCHK(hipEventRecord(c->timed[2].end[e], s));
        }
        if (tm) { if ((st = timed_slot(c, 3, &e))) return st; HIPCHK(hipEventRecord(c->timed[3].beg[e], s)); }
        const dim3 gp(c->pair_blocks[nsi]);
        if (d_recs) {
            if (nsi == 0) LAUNCH((k_pair<2, true>), gp, dim3(64), s, a);
            else if (nsi == 1) LAUNCH((k_pair<3, true>), gp, dim3(64), s, a);
            else LAUNCH((k_pair<4, true>), gp, dim3(64), s, a);
        } else {
            if (nsi == 0) LAUNCH((k_pair<2, false>), gp, dim3(64), s, a);
            else if (nsi == 1) LAUNCH((k_pair<3, false>), gp, dim3(64), s, a);
            else LAUNCH((k_pair<4, false>), gp, dim3(64), s, a);
        }
        if (tm) HIPCHK(hipEventRecord(c->timed[3].end[e], s));
    }
    if (walking) {  // the graph walk over every pair that reached threading, both mates (AQ.cpp:2072-2088), exact counting (:2189-2194)
        WalkArgs w;
        memset(&w, 0, sizeof(w));
        w.T = c->T; w.P = c->P; w.P.aln &= 3u; w.seq = d_seq; w.off = d_off;
        w.surv = c->d_sorted; w.nsurv = c->d_small + 0;
        w.walk_dst = c->d_walk; w.walk_ret = c->d_walk + npairs;
        w.counts = a.counts; w.counters = a.counters; w.ctr_rep = c->d_ctr; w.pstats = c->d_pstats;
        w.trecs = walk_trecs; w.errflag = c->d_small + 3;
#ifdef DBTK_STAMPS
        w.dbg = reinterpret_cast<uint64_t*>(c->d_small + 32);
#endif
        if (walk_txt) {  // text records: an arena sized for the worst case (two characters per entry, four strings), carved by the waves
            const uint32_t acap = std::min<uint32_t>(DBTK_THREAD_CAP, (max_read_len + max_read_len / 4 + 8 + 7) & ~7u);
            c->aln_cap = acap;
            uint64_t wfl_waves = 0;
            for (int q = 0; q < 3; ++q) wfl_waves += (uint64_t)std::max(c->wfl_blocks[q], c->wfl_blocks[3 + q]) * (q == 0 ? LOC_NW_XS : q == 1 ? LOC_NW_S : LOC_NW_L);
            const uint64_t want = std::min<uint64_t>(npairs * (uint64_t)(8 + 8 * acap + 8) + (uint64_t)TXT_CHUNK * (c->walk_blocks + c->walkfast_blocks + wfl_waves + 2), TXT_ARENA_MAX);  // (a chunk per wave that writes: every wave of the walk kernels' launches)
            if (want > c->txt_bytes) {
                if (c->d_txt) HIPCHK(hipFree(c->d_txt));
                c->d_txt = nullptr; c->txt_bytes = 0;
                HIPCHK(hipMalloc(&c->d_txt, want));
                c->txt_bytes = want;
            }
            if ((st = ensure(&c->d_txtidx, &c->txtidx_cap, npairs))) return st;
            HIPCHK(hipMemsetAsync(c->d_txtidx, 0xFF, npairs * sizeof(uint32_t), s));
            HIPCHK(hipMemsetAsync(c->d_small + 7, 0, 4, s));
            c->txt_cap = want;
            w.txt = c->d_txt; w.txt_idx = c->d_txtidx; w.ntxt = c->d_small + 7; w.txt_cap = (uint32_t)want; w.aln_cap = acap;
        }
        if (walk_aln) {
            // a record holds, per mate, cg.es and cg.tr: the read's bases plus what deletions can add (dbtk.h: DBTK_THREAD_CAP)
            const uint32_t acap = std::min<uint32_t>(DBTK_THREAD_CAP, (max_read_len + max_read_len / 4 + 8 + 7) & ~7u);
            c->aln_cap = acap;
            c->aln_stride = (uint32_t)sizeof(dbtk_aln_hdr_t) + 4 * acap;
            c->aln_max = npairs + (uint64_t)ALN_CHUNK * c->walk_blocks;
            if (c->aln_max * c->aln_stride > c->aln_bytes) {
                if (c->d_aln) HIPCHK(hipFree(c->d_aln));
                c->d_aln = nullptr; c->aln_bytes = 0;
                HIPCHK(hipMalloc(&c->d_aln, c->aln_max * c->aln_stride));
                c->aln_bytes = c->aln_max * c->aln_stride;
            }
            HIPCHK(hipMemsetAsync(c->d_small + 4, 0, 4, s));
            w.aln = c->d_aln; w.aln_stride = c->aln_stride; w.aln_cap = acap; w.aln_max = (uint32_t)std::min<uint64_t>(c->aln_max, 0xFFFFFFFFull);
            w.naln = c->d_small + 4;
        }
        // Two kernels when nothing needs the alignment arrays of every mate (no array records, no thread records): the lean one
        // decides and counts the pairs one of whose mates threads cleanly — with text records (-a / -ae): the pairs BOTH of whose
        // mates do, and writes their records itself —, the one with the error-correction machinery takes the rest from its list.  (The walk's kernels are timed together: "k_walk_pairs".)
        const int wnpl = (walk_aln || walk_trecs) ? 0 : walkfast_npl(max_read_len, k, w.T.grmz != nullptr);
        if (tm) { if ((st = timed_slot(c, 4, &e))) return st; HIPCHK(hipEventRecord(c->timed[4].beg[e], s)); }
        if (wnpl) {
            w.slow_list = c->d_walk + 2 * npairs; w.nslow = c->d_small + 5;
            if (info_rows) { w.slow_info = c->d_walk + 2 * npairs + slow_cap; w.info_cap = (uint32_t)std::min<uint64_t>(info_rows, 0x7FFFFFFFull); w.info_stride = 32u * (uint32_t)wnpl; }
            HIPCHK(hipMemsetAsync(c->d_small + 5, 0, 4, s));
            // (four ranges per resident wave even out their different costs: 19.7 -> 18.7 ms per 4 M reads; not with text records, where
            // every block that writes takes a chunk of the arena)
            if (w.T.gldir && locus_hint) {
                // the pairs of loci with a graph image: the lean kernel's locus-resident form (dbtk_walkfast.h), over the whole list;
                // the plain form then takes what is left
                LocItemArgs ia;
                memset(&ia, 0, sizeof(ia));
                ia.hist = c->d_surv + 3 * (npairs + 1); ia.nsurv = c->d_small + 0; ia.flag = c->d_small + 6; ia.dir = w.T.gldir;
                ia.nloci = (uint32_t)nloci; ia.t0 = 0; ia.tcap = (uint32_t)npairs;
                ia.cap_bytes[0] = LOC_IMGB_XS; ia.cap_bytes[1] = LOC_IMGB_S; ia.cap_bytes[2] = LOC_IMGB_L;
                for (int q = 0; q < 3; ++q) ia.items[q] = reinterpret_cast<uint4*>(c->d_surv + surv_words) + q * item_cap;
                ia.nitems = c->d_small + 8; ia.item_cap = (uint32_t)item_cap;
                ia.rest = c->d_surv + surv_words + 3 * 4 * item_cap;
                HIPCHK(hipMemsetAsync(c->d_small + 8, 0, 4 * sizeof(uint32_t), s));
                LAUNCH(k_loc_items, dim3((uint32_t)((nloci + 255) / 256)), dim3(256), s, ia);
                LAUNCH(k_loc_rest, dim3((uint32_t)std::min<uint64_t>((npairs + 255) / 256, (uint64_t)c->num_cu * 8)), dim3(256), s, ia);
                LocSplitArgs sp;
                memset(&sp, 0, sizeof(sp));
                for (int q = 0; q < 3; ++q) {
                    sp.items[q] = ia.items[q]; sp.starts[q] = c->d_surv + split_at + q * ((uint64_t)split_blk + 1);
                    sp.nblk[q] = (uint32_t)c->wfl_blocks[(wnpl == 3 ? 0 : 3) + q];
                    sp.wfix[q] = 4u << q;
                }
                sp.nitems = c->d_small + 8; sp.item_cap = (uint32_t)item_cap; sp.stats = c->d_pstats + 7;
                LAUNCH(k_loc_split, dim3(3), dim3(1024), s, sp);
                LocRunArgs r0{w.T.gldir, w.T.glimg, ia.items[0], c->d_small + 8, nullptr, nullptr, sp.starts[0]}, r1{w.T.gldir, w.T.glimg, ia.items[1], c->d_small + 9, nullptr, nullptr, sp.starts[1]},
                    r2{w.T.gldir, w.T.glimg, ia.items[2], c->d_small + 10, nullptr, nullptr, sp.starts[2]};
                // (DBTK_WALK_LOCUS_EC=1: the pairs these cannot decide stay with their locus — marked in walk_ret, walked by k_walk_pairs_locus
                // over the same items with the error correction's graph look-ups answered from LDS.  Built and bit-exact in round 5, and
                // MEASURED SLOWER than the global-table kernel (walk kernels 23.0 against 16.2 ms per 10 M reads: DESIGN 4.2) — the walk is not
                // bound by its look-ups' round trips — so it is off unless asked for; the pairs go on the other kernel's list)
                static const bool locus_ec = [] { const char* e = getenv("DBTK_WALK_LOCUS_EC"); return e && atoi(e) != 0; }();
                w.pend_locus = locus_ec ? 1u : 0u;
                const int wq = wnpl == 3 ? 0 : 3;
                if (wnpl == 3) {
                    LAUNCH((k_walk_fast_locus<3, LOC_NW_XS, LOC_IMGB_XS>), dim3(c->wfl_blocks[0]), dim3(LOC_NW_XS * 64), s, w, r0);
                    LAUNCH((k_walk_fast_locus<3, LOC_NW_S, LOC_IMGB_S>), dim3(c->wfl_blocks[1]), dim3(LOC_NW_S * 64), s, w, r1);
                    LAUNCH((k_walk_fast_locus<3, LOC_NW_L, LOC_IMGB_L>), dim3(c->wfl_blocks[2]), dim3(LOC_NW_L * 64), s, w, r2);
                } else {
                    LAUNCH((k_walk_fast_locus<5, LOC_NW_XS, LOC_IMGB_XS>), dim3(c->wfl_blocks[3]), dim3(LOC_NW_XS * 64), s, w, r0);
                    LAUNCH((k_walk_fast_locus<5, LOC_NW_S, LOC_IMGB_S>), dim3(c->wfl_blocks[4]), dim3(LOC_NW_S * 64), s, w, r1);
                    LAUNCH((k_walk_fast_locus<5, LOC_NW_L, LOC_IMGB_L>), dim3(c->wfl_blocks[5]), dim3(LOC_NW_L * 64), s, w, r2);
                }
                if (locus_ec) {  // (the same grids: the workgroups' ranges of the item lists are the ones k_loc_split made for those launches)
                    LAUNCH((k_walk_pairs_locus<LOC_IMGB_XS>), dim3(c->wfl_blocks[wq + 0]), dim3(WPL_NW * 64), s, w, r0);
                    LAUNCH((k_walk_pairs_locus<LOC_IMGB_S>), dim3(c->wfl_blocks[wq + 1]), dim3(WPL_NW * 64), s, w, r1);
                    LAUNCH((k_walk_pairs_locus<LOC_IMGB_L>), dim3(c->wfl_blocks[wq + 2]), dim3(WPL_NW * 64), s, w, r2);
                }
                w.pend_locus = 0;
                w.sel = ia.rest; w.nsel = c->d_small + 11;
            }
            const dim3 gf(walk_txt ? c->walkfast_blocks : 4 * c->walkfast_blocks);
            const bool w11 = w.T.grmz && k - mz_m_for_k(k) + 1 == 11;  // (windows of 7 m-mers for k = 19 .. 22, of 11 for k = 23 .. 26)
            if (wnpl == 3) { if (w11) LAUNCH((k_walk_fast<3, 11>), gf, dim3(64), s, w); else LAUNCH((k_walk_fast<3, 7>), gf, dim3(64), s, w); }
            else { if (w11) LAUNCH((k_walk_fast<5, 11>), gf, dim3(64), s, w); else LAUNCH((k_walk_fast<5, 7>), gf, dim3(64), s, w); }
        }
        LAUNCH(k_walk_pairs, dim3(c->walk_blocks), dim3(64), s, w);
        if (tm) HIPCHK(hipEventRecord(c->timed[4].end[e], s));
    }
    c->fold_pending = true;  // (the counter replicas: folded by sync_all)
    HIPCHK(hipGetLastError());
    return DBTK_OK;
}

}  // namespace

extern "C" {

int dbtk_ctx_path_stats(dbtk_ctx_t* c, uint64_t* out, int cap) {
    if (!c || !out || cap <= 0) return 0;
    if (hipSetDevice(c->device) != hipSuccess || sync_all(c) != hipSuccess) return 0;
    uint64_t v[DBTK_PATH_STATS];
    if (hipMemcpy(v, c->d_pstats, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    const int n = cap < (int)DBTK_PATH_STATS ? cap : (int)DBTK_PATH_STATS;
    for (int i = 0; i < n; ++i) out[i] = v[i];
    return n;
}

int dbtk_ctx_table_bytes(dbtk_ctx_t* c, const char** names, uint64_t* bytes, int cap) {
    if (!c || !names || !bytes || !c->share) return 0;
    static const char* nm[13] = {"index", "presence_filter", "class_table", "index_by_minimizer", "index_overflow", "graph", "graph_by_minimizer",
                                 "index_images", "index_images:from_cache", "vv+qc+perm+trbeg", "gates(tre,bait)", "total", "graph_images"};
    int n = 0;
    for (int i = 0; i < 13 && n < cap; ++i) { names[n] = nm[i]; bytes[n] = c->share->bytes[i]; ++n; }
    return n;
}

static dbtk_status_t dbtk_ctx_create_impl(const dbtk_rpgg_t* h, const dbtk_params_t* p, int device_id, dbtk_ctx_t** out) {
    if (!h || !p || !out) { set_error("null argument"); return DBTK_ERR_ARG; }
    *out = nullptr;
    if (p->ksize != h->ksize) { set_error("params.ksize differs from the RPGG's k"); return DBTK_ERR_ARG; }
    if (p->n_filter == 1) { set_error("-kf 1 M divides by zero in the reference (subfilter); refusing"); return DBTK_ERR_ARG; }
    if (p->n_filter > 32) { set_error("-kf N: N > 32 unsupported"); return DBTK_ERR_UNSUPPORTED; }
    if (p->bait && h->bt_cnt.empty()) { set_error("params.bait set but the RPGG handle has no bait DB"); return DBTK_ERR_ARG; }
    if (p->trackbait && !p->bait) { set_error("params.trackbait needs params.bait"); return DBTK_ERR_ARG; }
    if (p->bubbles && (h->tre_cnt.empty() || p->extract)) { set_error("params.bubbles needs PREF.tre.kdb (and is not an extract-mode flag)"); return DBTK_ERR_ARG; }
    if (p->qc && h->qc.empty()) { set_error("params.qc set but the RPGG handle has no QC mask"); return DBTK_ERR_ARG; }
    if ((p->aln & 3u) == 3u || (p->aln & ~7u)) { set_error("params.aln: 0, 1 (-a) or 2 (-ae), optionally | DBTK_ALN_TEXT"); return DBTK_ERR_ARG; }
    if (p->threading > DBTK_THREADING_V13) { set_error("params.threading: 0, 1 (HEAD) or 2 (v1.3)"); return DBTK_ERR_ARG; }
    if (p->threading == DBTK_THREADING_V13 && h->gr_cnt.empty()) { set_error("params.threading = 2 needs the graph in the RPGG handle (DBTK_LOAD_GRAPH)"); return DBTK_ERR_ARG; }
    if (p->threading == DBTK_THREADING_V13 && p->extract) { set_error("threading = 2 with -e is not supported"); return DBTK_ERR_UNSUPPORTED; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device: this library has no CPU execution path");
        return DBTK_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= ndev) { set_error("device_id out of range"); return DBTK_ERR_ARG; }
    HIPCHK(hipSetDevice(device_id));
    dbtk_ctx* c = new dbtk_ctx;
    memset(c->timed, 0, sizeof(c->timed));
    c->g = h; c->g_uid = h->uid; c->P = *p; c->device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { free_ctx(c); set_error("hipGetDeviceProperties failed"); return DBTK_ERR_HIP; }
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_encode_subfilter, K1_NT, 0) != hipSuccess || nb <= 0) nb = 4;
        if (const char* ev = getenv("DBTK_ENC_WPC")) { const int v = atoi(ev); if (v > 0) nb = v; }  // diagnostic: encode waves per CU (more than resident: several rounds)
        c->k1_blocks = c->num_cu * nb;
        if (const char* ev = getenv("DBTK_PROBE_WPC")) { const int v = atoi(ev); if (v > 0) c->probe_wpc = v; }  // probe waves per CU
        {
            const void* k2[4] = {(const void*)k_probe<3, 7>, (const void*)k_probe<3, 11>, (const void*)k_probe<5, 7>, (const void*)k_probe<5, 11>};
            for (int i = 0; i < 4; ++i) {
                nb = 0;
                // blocks per CU: eight times what the occupancy query says is resident.  A wave works through one contiguous range of
                // the locus-ordered list, so ranges an eighth as long even out their different costs (5.42 -> 5.09 ms per all-hit launch);
                // and a grid of EXACTLY the resident waves is a cliff: the query counts LDS in finer granules than the hardware
                // allocates, and one block too many per CU runs alone in a second round at twice the time (measured with a 10.7-KB variant
                // of this kernel: 15 per CU by the query, 14 in fact: 8.7 ms against 6.1).
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k2[i], 64, 0) != hipSuccess || nb <= 0) nb = 8;
                nb *= 8;
                if (const char* ev = getenv("DBTK_PROBE_WPC")) { const int v = atoi(ev); if (v > 0) nb = v; }
                c->probe2_wpc[i] = nb;
            }
            const void* k2f[4] = {(const void*)k_probe<3, 7, false, true>, (const void*)k_probe<3, 11, false, true>, (const void*)k_probe<5, 7, false, true>, (const void*)k_probe<5, 11, false, true>};
            for (int i = 0; i < 4; ++i) {  // (the fused forms: their own register count)
                nb = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k2f[i], 64, 0) != hipSuccess || nb <= 0) nb = 8;
                nb *= 8;
                if (const char* ev = getenv("DBTK_PROBE_WPC")) { const int v = atoi(ev); if (v > 0) nb = v; }
                c->probe2f_wpc[i] = nb;
            }
            if (getenv("DBTK_VERBOSE")) fprintf(stderr, "k_probe<5, 7>: %d blocks per CU\n", c->probe2_wpc[2]);
            {   // the locus-resident kernel: as many workgroups as are resident (an item is short: the workgroups take them round robin)
                const void* kl[6] = {(const void*)k_probe_locus<3, LOC_NW_XS, LOC_IMGB_XS>, (const void*)k_probe_locus<3, LOC_NW_S, LOC_IMGB_S>, (const void*)k_probe_locus<3, LOC_NW_L, LOC_IMGB_L>,
                                     (const void*)k_probe_locus<5, LOC_NW_XS, LOC_IMGB_XS>, (const void*)k_probe_locus<5, LOC_NW_S, LOC_IMGB_S>, (const void*)k_probe_locus<5, LOC_NW_L, LOC_IMGB_L>};
                const int nwv[3] = {LOC_NW_XS, LOC_NW_S, LOC_NW_L};
                for (int i = 0; i < 6; ++i) {
                    nb = 0;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kl[i], nwv[i % 3] * 64, 0) != hipSuccess || nb <= 0) nb = 1;
                    nb *= 2;  // (twice as many as are resident: each takes a contiguous, weight-balanced range of the items, k_loc_split, and the ones that start late even out what the weights do not see)
                    if (const char* ev = getenv("DBTK_LOC_BPC")) { const int v = atoi(ev); if (v > 0) nb = v; }
                    c->loc_blocks[i] = c->num_cu * nb;
                    if (getenv("DBTK_VERBOSE")) fprintf(stderr, "k_probe_locus[%d]: %d workgroups per CU\n", i, nb);
                }
                const void* kf[6] = {(const void*)k_probe_locus<3, LOC_NW_XS, LOC_IMGB_XS, true>, (const void*)k_probe_locus<3, LOC_NW_S, LOC_IMGB_S, true>, (const void*)k_probe_locus<3, LOC_NW_L, LOC_IMGB_L, true>,
                                     (const void*)k_probe_locus<5, LOC_NW_XS, LOC_IMGB_XS, true>, (const void*)k_probe_locus<5, LOC_NW_S, LOC_IMGB_S, true>, (const void*)k_probe_locus<5, LOC_NW_L, LOC_IMGB_L, true>};
                for (int i = 0; i < 6; ++i) {
                    nb = 0;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kf[i], nwv[i % 3] * 64, 0) != hipSuccess || nb <= 0) nb = 1;
                    nb *= 2;
                    if (const char* ev = getenv("DBTK_LOC_BPC")) { const int v = atoi(ev); if (v > 0) nb = v; }
                    c->locf_blocks[i] = c->num_cu * nb;
                    if (getenv("DBTK_VERBOSE")) fprintf(stderr, "k_probe_locus[%d] (fused): %d workgroups per CU\n", i, nb);
                }
            }
        }
        // resident waves of each resolve-kernel instance (one vote-spill scratch row per resident wave)
        const void* kp[3] = {(const void*)k_pair<2, true>, (const void*)k_pair<3, true>, (const void*)k_pair<4, true>};
        for (int i = 0; i < 3; ++i) {
            nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kp[i], 64, 0) != hipSuccess || nb <= 0) nb = 8;
            if (const char* e = getenv("DBTK_PAIR_WPC")) { const int v = atoi(e); if (v > 0) nb = v; }  // diagnostic: blocks per CU (the vote scratch follows)
            if (getenv("DBTK_VERBOSE")) fprintf(stderr, "k_pair<%d>: %d waves per CU\n", i + 2, nb);
            c->vote_rows = std::max(c->vote_rows, c->num_cu * nb);
            if (!getenv("DBTK_PAIR_WPC")) nb *= 4;  // (ranges a quarter as long, as for the other range kernels: 0.179 -> 0.147 ms in the headline)
            c->pair_blocks[i] = c->num_cu * nb;
            c->max_pair_blocks = std::max(c->max_pair_blocks, c->pair_blocks[i]);
            const void* ku[3] = {(const void*)k_pair_usual<2, true>, (const void*)k_pair_usual<3, true>, (const void*)k_pair_usual<4, true>};
            nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ku[i], 64, 0) != hipSuccess || nb <= 0) nb = 16;
            nb *= 4;  // (ranges a quarter as long, as for the probe kernel: 2.66 -> 1.93 ms per all-hit launch; eight times: 1.78, but the
                      // genome-like mix, where most survivors leave at once, then pays for the blocks' own start-up: 0.26 -> 0.40 ms)
            if (const char* e = getenv("DBTK_USUAL_WPC")) { const int v = atoi(e); if (v > 0) nb = v; }  // diagnostic: blocks per CU
            c->usual_blocks[i] = c->num_cu * nb;
        }
    }
    dbtk_status_t st = DBTK_OK;
    do {
        if (hipStreamCreate(&c->stream) != hipSuccess) { set_error("hipStreamCreate failed"); st = DBTK_ERR_HIP; break; }
        c->timed[0].name = "k_encode_subfilter";
        c->timed[1].name = "k_probe";
        c->timed[2].name = "k_pair_usual";
        c->timed[3].name = "k_pair";
        c->timed[4].name = "k_walk_pairs";
        c->timed[5].name = "k_surv_sort";
        {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_walk_pairs, 64, 0) != hipSuccess || nb <= 0) nb = 8;
            if (const char* e = getenv("DBTK_WALK_WPC")) { const int v = atoi(e); if (v > 0) nb = v; }
            c->walk_blocks = c->num_cu * nb;
            nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_walk_fast<5, 7>, 64, 0) != hipSuccess || nb <= 0) nb = 16;
            c->walkfast_blocks = c->num_cu * nb;
            const void* kw[6] = {(const void*)k_walk_fast_locus<3, LOC_NW_XS, LOC_IMGB_XS>, (const void*)k_walk_fast_locus<3, LOC_NW_S, LOC_IMGB_S>, (const void*)k_walk_fast_locus<3, LOC_NW_L, LOC_IMGB_L>,
                                 (const void*)k_walk_fast_locus<5, LOC_NW_XS, LOC_IMGB_XS>, (const void*)k_walk_fast_locus<5, LOC_NW_S, LOC_IMGB_S>, (const void*)k_walk_fast_locus<5, LOC_NW_L, LOC_IMGB_L>};
            const int nwv[3] = {LOC_NW_XS, LOC_NW_S, LOC_NW_L};
            for (int i = 0; i < 6; ++i) {
                nb = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kw[i], nwv[i % 3] * 64, 0) != hipSuccess || nb <= 0) nb = 1;
                nb *= 2;  // (as for k_probe_locus: contiguous weight-balanced ranges, twice the resident workgroups)
                if (const char* ev = getenv("DBTK_WFL_BPC")) { const int v = atoi(ev); if (v > 0) nb = v; }
                c->wfl_blocks[i] = c->num_cu * nb;
                if (getenv("DBTK_VERBOSE")) fprintf(stderr, "k_walk_fast_locus[%d]: %d workgroups per CU\n", i, nb);
            }
        }
        for (int i = 0; i < NKERN && !st; ++i)
            for (int j = 0; j < EVPOOL && !st; ++j)
                if (hipEventCreate(&c->timed[i].beg[j]) != hipSuccess || hipEventCreate(&c->timed[i].end[j]) != hipSuccess) { set_error("hipEventCreate failed"); st = DBTK_ERR_HIP; }
        if (st) break;
        // what the context owns itself FIRST, the shared tables after it: a context created on a side thread while another one builds the
        // tables (the command line's further aligner contexts for -a / -ae) has its buffers by the time the tables are there
        if (p->trackbait) {
            c->btTK.resize(h->nloci);
            c->baitDB_host.resize(h->nloci);
            uint64_t i = 0;
            for (uint64_t l = 0; l < h->nloci; ++l)
                for (uint64_t j = 0; j < h->bt_cnt[l]; ++j, ++i) c->baitDB_host[l][h->bt_ks[i]] = h->bt_vs[i];
        }
        if (p->bubbles) {
            c->bubbleDB.resize(h->nloci);
            if (hipMalloc(&c->d_nevents, 4) != hipSuccess) { set_error("hipMalloc nevents"); st = DBTK_ERR_HIP; break; }
        }
        c->ntr = h->out_kmer.size();
        c->n_accum = c->ntr + 2 * h->nloci + DBTK_C_COUNT;
        auto chk = [&](hipError_t e, const char* what) { if (e != hipSuccess && !st) { set_error(std::string(what) + ": " + hipGetErrorString(e)); st = DBTK_ERR_HIP; } };
        chk(hipMalloc(&c->d_accum, c->n_accum * 8), "hipMalloc accum");
        chk(hipMalloc(&c->d_ctr, (size_t)CTR_REP * CTR_STRIDE * 8), "hipMalloc counter replicas");
        if (!st) chk(hipMemsetAsync(c->d_ctr, 0, (size_t)CTR_REP * CTR_STRIDE * 8, c->stream), "memset");
        chk(hipMalloc(&c->d_pstats, DBTK_PATH_STATS * 8), "hipMalloc path statistics");
        if (!st) chk(hipMemsetAsync(c->d_pstats, 0, DBTK_PATH_STATS * 8, c->stream), "memset");
        chk(hipMalloc(&c->d_small, 4 * (TK_INLINE + SMALL_WORDS) + 48 * 8), "hipMalloc small");
        if (c->d_small) c->d_small += TK_INLINE;
        if (!getenv("DBTK_LOCUS_ALWAYS")) {  // (DBTK_LOCUS_ALWAYS=1: every batch launches the locus path: tests of small batches)
            chk(hipHostMalloc((void**)&c->h_sortflag, 64, hipHostMallocDefault), "hipHostMalloc");
            if (c->h_sortflag) { c->h_sortflag[0] = 0xFFFFFFFFu; c->h_sortflag[1] = 0xFFFFFFFFu; c->h_sortflag[6] = 1u; }
        }
        if (!st) chk(hipMemsetAsync(c->d_small - TK_INLINE, 0, 4 * (TK_INLINE + SMALL_WORDS) + 48 * 8, c->stream), "memset");
        chk(hipMalloc(&c->d_vote, (size_t)c->vote_rows * (h->nloci + 1) * 8), "hipMalloc vote scratch");
        chk(hipMalloc(&c->d_epoch, (size_t)c->vote_rows * 16), "hipMalloc epoch");
        if (st) break;
        chk(hipMemsetAsync(c->d_accum, 0, c->n_accum * 8, c->stream), "memset");
        chk(hipMemsetAsync(c->d_vote, 0, (size_t)c->vote_rows * (h->nloci + 1) * 8, c->stream), "memset");
        chk(hipMemsetAsync(c->d_epoch, 0, (size_t)c->vote_rows * 16, c->stream), "memset");
        // further lanes (not with -bu: its event log is replayed batch by batch on the host; not in the stamps build)
        int nlanes = 1;
#ifndef DBTK_STAMPS
        {   // DBTK_LANES=1..3 (default 2; a third lane gains another 2 %, a fourth fell apart: 25 ms/step)
            const char* e = getenv("DBTK_LANES");
            nlanes = (e && atoi(e) >= 1 && atoi(e) <= 3) ? atoi(e) : 2;
            if (p->bubbles) nlanes = 1;
            c->two_lanes = nlanes > 1;
        }
#endif
        for (int li = 1; li < nlanes; ++li) {
            dbtk_ctx::Lane l;
            chk(hipStreamCreate(&l.stream), "hipStreamCreate");
            chk(hipMalloc(&l.d_small, 4 * (TK_INLINE + SMALL_WORDS) + 48 * 8), "hipMalloc small");
            if (l.d_small) l.d_small += TK_INLINE;
            chk(hipMalloc(&l.d_vote, (size_t)c->vote_rows * (h->nloci + 1) * 8), "hipMalloc vote scratch");
            chk(hipMalloc(&l.d_epoch, (size_t)c->vote_rows * 16), "hipMalloc epoch");
            if (li == 1) c->alt = l; else c->parked.push_back(l);
            if (st) break;
            chk(hipMemsetAsync(l.d_small - TK_INLINE, 0, 4 * (TK_INLINE + SMALL_WORDS) + 48 * 8, c->stream), "memset");
            chk(hipMemsetAsync(l.d_vote, 0, (size_t)c->vote_rows * (h->nloci + 1) * 8, c->stream), "memset");
            chk(hipMemsetAsync(l.d_epoch, 0, (size_t)c->vote_rows * 16, c->stream), "memset");
        }
        if (st) break;
        {   // the tables: shared per (handle, device)
            std::lock_guard<std::mutex> lk(g_share_m);
            const auto key = std::make_pair(h->uid, device_id);
            auto it = g_shares.find(key);
            TableShare* sh = it != g_shares.end() ? it->second : nullptr;
            auto to_share = [&](TableShare* t) {
                t->d_idx = c->d_idx; t->d_flt = c->d_flt; t->flt_words = c->flt_words; t->d_trbeg = c->d_trbeg; t->d_cls = c->d_cls; t->d_mz = c->d_mz; t->d_ovf = c->d_ovf;
                t->d_gr = c->d_gr; t->d_grmz = c->d_grmz; t->d_ldir = c->d_ldir; t->d_limg = c->d_limg; t->d_gldir = c->d_gldir; t->d_glimg = c->d_glimg; t->d_vv = c->d_vv; t->d_qc = c->d_qc; t->d_perm = c->d_perm; t->d_tre = c->d_tre; t->d_bait = c->d_bait;
                t->T = c->T; t->consistent = c->consistent;
                // HBM bytes per table: what this context built is added to what the share already holds
                const uint64_t mine[9] = {c->tb_idx, c->tb_flt, c->tb_cls, c->tb_mz, c->tb_ovf, c->tb_gr, c->tb_grmz, c->limg_bytes, c->loc_from_cache ? 1u : 0u};
                for (int i = 0; i < 9; ++i) if (mine[i]) t->bytes[i] = mine[i];
                if (c->glimg_bytes) t->bytes[12] = c->glimg_bytes;
                t->bytes[9] = (h->vv.size() + 1) * 4 + (h->qc.empty() ? 0 : h->nloci) + ((size_t)NHMAX * (NHMAX + 1) / 2 + 1) * 2 + (h->nloci + 1) * 4 + (c->d_ldir ? h->nloci * sizeof(LocusDir) : 0);
                t->bytes[10] = (c->d_tre ? (c->T.tre_mask + 1) * sizeof(ClsSlot) : 0) + (c->d_bait ? (c->T.bait_mask + 1) * sizeof(ClsSlot) : 0);
                t->bytes[11] = 0;
                for (int i = 0; i < 13; ++i) if (i != 8 && i != 11) t->bytes[11] += t->bytes[i];
            };
            if (!sh) {
                if ((st = build_tables(c))) break;
                sh = new TableShare;
                to_share(sh);
                g_shares[key] = sh;
            } else {
                c->d_idx = sh->d_idx; c->d_flt = sh->d_flt; c->flt_words = sh->flt_words; c->d_trbeg = sh->d_trbeg; c->d_cls = sh->d_cls; c->d_mz = sh->d_mz; c->d_ovf = sh->d_ovf;
                c->d_gr = sh->d_gr; c->d_grmz = sh->d_grmz; c->d_ldir = sh->d_ldir; c->d_limg = sh->d_limg; c->d_gldir = sh->d_gldir; c->d_glimg = sh->d_glimg; c->d_vv = sh->d_vv; c->d_qc = sh->d_qc; c->d_perm = sh->d_perm; c->d_tre = sh->d_tre; c->d_bait = sh->d_bait;
                c->T = sh->T; c->consistent = sh->consistent;
            }
            c->share = sh;
            ++sh->refs;
            // optional tables, by the first context that needs them (only a walking context pays for the graph table)
            if (p->threading == DBTK_THREADING_V13 && !c->d_gr) { if ((st = build_graph_table(c))) break; }
            if (p->bubbles && !c->d_tre) {
                if ((st = build_kl_table(c, h->tre_cnt, h->tre_ks, nullptr, &c->d_tre, &c->T.tre_mask, &c->T.tre_shift))) break;
                c->T.tre = c->d_tre;
            }
            if (p->bait && !c->d_bait) {
                if ((st = build_kl_table(c, h->bt_cnt, h->bt_ks, &h->bt_vs, &c->d_bait, &c->T.bait_mask, &c->T.bait_shift))) break;
                c->T.bait = c->d_bait;
            }
            to_share(sh);
        }
        if (st) break;
        chk(hipStreamSynchronize(c->stream), "sync");
    } while (0);
    if (st) { free_ctx(c); return st; }
    *out = c;
    return DBTK_OK;
}

void dbtk_ctx_free(dbtk_ctx_t* ctx) { free_ctx(ctx); }

// A batch whose reads are in device memory (d_seq, d_off; d_qual or null), run to completion, with what the host-buffer
// entry point hands back: records in pair order, the -bu replay, the -tb replay (which needs the reads on the host: seq / off /
// qual, else null).  Shared by dbtk_align_batch and dbtk_ingest_align.
static dbtk_status_t run_batch_sync(dbtk_ctx_t* c, const uint8_t* d_seq, const uint64_t* d_off, const uint8_t* d_qual, uint64_t nbytes,
                                    uint64_t npairs, uint32_t maxlen, const uint8_t* seq, const uint64_t* off, const uint8_t* qual,
                                    dbtk_pair_rec_t* recs, uint64_t rec_cap, uint64_t* nrec) {
    dbtk_status_t st;
    hipStream_t s = c->stream;
    bool want_recs = recs && rec_cap && (c->P.trace || c->P.okam || c->P.extract);
    if (c->P.trackbait && !want_recs) {  // the replay needs the bait-stage records even when the caller wants none
        c->own_recs.resize(npairs);
        recs = c->own_recs.data();
        rec_cap = npairs;
        want_recs = true;
    }
    uint64_t dcap = 0;
    if (want_recs) {
        dcap = c->P.trace ? npairs : (rec_cap < npairs ? rec_cap : npairs);
        if (c->P.trace && rec_cap < npairs) { set_error("trace mode needs rec_cap >= npairs"); return DBTK_ERR_ARG; }
        if ((st = ensure(&c->d_recs, &c->rec_cap, dcap))) return st;
    }
    if (c->P.trackbait && !seq) { set_error("-tb needs the reads in host buffers"); return DBTK_ERR_UNSUPPORTED; }
    const bool walk_recs = c->P.threading == DBTK_THREADING_V13 && c->P.trace;
    if (walk_recs && (st = ensure(&c->d_trecs, &c->trecs_cap, 2 * npairs))) return st;
    c->last_walk_npairs = c->P.threading == DBTK_THREADING_V13 ? npairs : 0;
    c->last_walk_recs = walk_recs;
    const bool walk_txt = c->P.threading == DBTK_THREADING_V13 && (c->P.aln & 3u) && (c->P.aln & DBTK_ALN_TEXT);
    const bool walk_aln = c->P.threading == DBTK_THREADING_V13 && (c->P.aln & 3u) && !walk_txt;
    if (!walk_aln) c->aln_max = 0;
    if (!walk_txt) c->txt_cap = 0;
    if ((st = launch_batch(c, d_seq, d_off, nbytes, npairs, maxlen, want_recs ? c->d_recs : nullptr, (uint32_t)dcap,
                           d_qual, walk_recs ? c->d_trecs : nullptr, walk_aln, walk_txt))) return st;
    uint32_t small[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(small, c->d_small, sizeof(small), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (small[3]) {
        (void)hipMemsetAsync(c->d_small + 3, 0, 4, s);
        set_error("device reported an over-long read");
        return (dbtk_status_t)small[3];
    }
    if (c->P.bubbles) {
        // Replay the batch's novel edges the way the reference accumulates them: the worker's per-batch
        // `bubbles[destLocus][edge]` is filled in pair order, mate 1 before mate 2, positions ascending
        // (AQ.cpp:2161-2166), then merged map by map into bubbleDB (accumBubbles, AQ.cpp:1599-1606).
        uint32_t nev = 0;
        HIPCHK(hipMemcpy(&nev, c->d_nevents, 4, hipMemcpyDeviceToHost));
        if (nev > c->events_cap) { set_error("bubble event log overflow"); return DBTK_ERR_OVERFLOW; }
        std::vector<BubEvent> ev(nev);
        if (nev) HIPCHK(hipMemcpy(ev.data(), c->d_events, (size_t)nev * sizeof(BubEvent), hipMemcpyDeviceToHost));
        std::sort(ev.begin(), ev.end(), [](const BubEvent& x, const BubEvent& y) {
            if (x.pair != y.pair) return x.pair < y.pair;
            if (x.mate != y.mate) return x.mate < y.mate;
            return x.pos < y.pos;
        });
        std::unordered_map<uint64_t, std::unordered_map<size_t, uint32_t>> bubbles;  // bubbles_t, AQ.cpp:43
        for (const BubEvent& e : ev) ++bubbles[e.locus][(size_t)e.edge];
        for (auto& pl : bubbles) {
            auto& bu_o = c->bubbleDB[pl.first];
            for (auto& q : pl.second) bu_o[q.first] += q.second;
        }
    }
    if (want_recs) {
        const uint64_t produced = c->P.trace ? npairs : small[2];
        if (nrec) *nrec = produced;
        const uint64_t ncopy = produced < dcap ? produced : dcap;
        if (ncopy) HIPCHK(hipMemcpy(recs, c->d_recs, ncopy * sizeof(dbtk_pair_rec_t), hipMemcpyDeviceToHost));
        if (!c->P.trace) {
            // compaction order is arbitrary on the device; the reference's order within a batch is pair order
            std::sort(recs, recs + ncopy, [](const dbtk_pair_rec_t& x, const dbtk_pair_rec_t& y) { return x.pair < y.pair; });
            if (produced > dcap) { set_error("record buffer too small"); return DBTK_ERR_OVERFLOW; }
        }
        if (c->P.trackbait) {
            // ---- -tb: for every mate the bait filter flagged, the k-mer that made bfilter_FPSv1 return — the FIRST violated one
            // in the iteration order of its kc8_t (unordered_map<uint64_t, uint8_t>, AQ.cpp:1380-1394 / 1401-1417), found by
            // filling the same container in the same order.  Per batch into tkr (AQ.cpp:1993), then accumBaitKmerHits (1608-1616).
            std::unordered_map<uint32_t, std::unordered_map<uint64_t, uint64_t>> tkr;
            const uint32_t k = c->g->ksize;
            std::vector<uint64_t> ks;
            for (uint64_t i = 0; i < ncopy; ++i) {
                const dbtk_pair_rec_t& r = recs[i];
                if (r.stage != DBTK_STAGE_BAIT) continue;
                for (int m = 0; m < 2; ++m) {
                    if (!(m ? r.r2.bf : r.r1.bf)) continue;
                    const uint64_t o0 = off[2 * (uint64_t)r.pair + m], o1 = off[2 * (uint64_t)r.pair + m + 1];
                    const uint8_t* rd = seq + o0;
                    const uint64_t len = o1 - o0;
                    if (len < k) continue;
                    // read2kmers_edges' k-mers: canonical k-mer per position, NAN64 where the window has a non-ACGT base
                    ks.assign(len - k + 1, NAN64);
                    uint64_t fw = 0, rc = 0, run = 0;
                    const uint64_t mask = (k < 32) ? ((1ull << (2 * k)) - 1) : ~0ull;
                    for (uint64_t b = 0; b < len; ++b) {
                        uint64_t code;
                        switch (rd[b]) { case 'A': code = 0; break; case 'C': code = 1; break; case 'G': code = 2; break; case 'T': code = 3; break; default: code = 4; }
                        if (code == 4) { run = 0; fw = rc = 0; continue; }
                        fw = ((fw << 2) | code) & mask;
                        rc = (rc >> 2) | ((3 - code) << (2 * (k - 1)));
                        if (++run >= k) ks[b + 1 - k] = fw < rc ? fw : rc;
                    }
                    uint32_t qm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    const bool fq = qual != nullptr;
                    if (fq) qmask_scan(qual + o0, (int)len, (int)c->P.qth, (int)k, qm);
                    std::unordered_map<uint64_t, uint8_t> kc;  // kc8_t
                    for (uint64_t p = 0; p < ks.size(); ++p)
                        if (!fq || ((qm[p >> 5] >> (p & 31)) & 1)) ++kc[ks[p]];
                    const auto& baitdb = c->baitDB_host[r.dst0];
                    for (auto& pc : kc) {
                        auto it = baitdb.find(pc.first);
                        if (it == baitdb.end()) continue;
                        const uint8_t mi = (uint8_t)(it->second >> 8), ma = (uint8_t)(it->second & 0xff);
                        if (pc.second < mi || pc.second > ma) { ++tkr[r.dst0][pc.first]; break; }
                    }
                }
            }
            for (auto& p1 : tkr) {
                auto& t = c->btTK[p1.first];
                for (auto& p2 : p1.second) t[p2.first] += p2.second;
            }
        }
    }
    return DBTK_OK;
}

static dbtk_status_t dbtk_align_batch_impl(dbtk_ctx_t* c, const uint8_t* seq, const uint64_t* off, const uint8_t* qual,
                               uint64_t npairs, dbtk_pair_rec_t* recs, uint64_t rec_cap, uint64_t* nrec) {
    if (!c || !off || (!seq && npairs)) { set_error("null argument"); return DBTK_ERR_ARG; }
    if (nrec) *nrec = 0;
    HIPCHK(hipSetDevice(c->device));
    const uint64_t nreads = 2 * npairs;
    for (uint64_t r = 0; r < nreads; ++r) {
        if (off[r + 1] < off[r]) { set_error("seq_offsets not monotone"); return DBTK_ERR_ARG; }
        if (off[r + 1] - off[r] > DBTK_MAX_READ_LEN) {
            set_error("read longer than DBTK_MAX_READ_LEN (256): the reference's PE_KMC is uint8_t, src/aQueryFasta_thread.cpp:42");
            return DBTK_ERR_READ_TOO_LONG;
        }
    }
    if (npairs == 0) return DBTK_OK;
    const uint64_t base = off[0], nbytes = off[nreads] - base;
    dbtk_status_t st;
    if ((st = ensure(&c->d_seq, &c->seq_cap, nbytes + 32))) return st;
    if ((st = ensure(&c->d_off, &c->off_cap, nreads + 1))) return st;
    hipStream_t s = c->stream;
    if (nbytes) HIPCHK(hipMemcpyAsync(c->d_seq, seq + base, nbytes, hipMemcpyHostToDevice, s));
    if (base == 0) {
        HIPCHK(hipMemcpyAsync(c->d_off, off, (nreads + 1) * 8, hipMemcpyHostToDevice, s));
    } else {
        std::vector<uint64_t> o2(nreads + 1);
        for (uint64_t r = 0; r <= nreads; ++r) o2[r] = off[r] - base;
        HIPCHK(hipMemcpy(c->d_off, o2.data(), (nreads + 1) * 8, hipMemcpyHostToDevice));
    }
    uint32_t maxlen = 0;
    for (uint64_t r = 0; r < nreads; ++r) maxlen = std::max<uint32_t>(maxlen, (uint32_t)(off[r + 1] - off[r]));
    const bool use_qual = c->P.bait && qual;  // base qualities only matter to the bait filter (-b with -fq)
    if (use_qual) {
        if ((st = ensure(&c->d_qual, &c->qual_cap, nbytes + 32))) return st;
        if (nbytes) HIPCHK(hipMemcpyAsync(c->d_qual, qual + base, nbytes, hipMemcpyHostToDevice, s));
    }
    return run_batch_sync(c, c->d_seq, c->d_off, use_qual ? c->d_qual : nullptr, nbytes, npairs, maxlen, seq, off, qual, recs, rec_cap, nrec);
}

// Function-level entry of the graph walk (tests, and callers that want isThreadFeasible alone).
static dbtk_status_t dbtk_thread_batch_impl(dbtk_ctx_t* c, const uint8_t* seq, const uint64_t* off, const uint32_t* loci, uint64_t nreads,
                                dbtk_thread_rec_t* recs) {
    if (!c || !off || !loci || !recs || (!seq && nreads)) { set_error("null argument"); return DBTK_ERR_ARG; }
    if (!c->d_gr) { set_error("the RPGG handle holds no graph (DBTK_LOAD_GRAPH)"); return DBTK_ERR_ARG; }
    if (nreads >= 0x7FFFFFFFull) { set_error("too many reads"); return DBTK_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    for (uint64_t r = 0; r < nreads; ++r) {
        if (off[r + 1] < off[r]) { set_error("seq_offsets not monotone"); return DBTK_ERR_ARG; }
        if (off[r + 1] - off[r] > DBTK_MAX_READ_LEN) { set_error("read longer than DBTK_MAX_READ_LEN"); return DBTK_ERR_READ_TOO_LONG; }
        if (loci[r] >= c->g->nloci) { set_error("locus out of range"); return DBTK_ERR_ARG; }
    }
    if (!nreads) return DBTK_OK;
    const uint64_t base = off[0], nbytes = off[nreads] - base;
    dbtk_status_t st;
    if ((st = ensure(&c->d_seq, &c->seq_cap, nbytes + 32))) return st;
    if ((st = ensure(&c->d_off, &c->off_cap, nreads + 1))) return st;
    if ((st = ensure(&c->d_loci, &c->loci_cap, nreads))) return st;
    if ((st = ensure(&c->d_trecs, &c->trecs_cap, nreads))) return st;
    hipStream_t s = c->stream;
    std::vector<uint64_t> o2(nreads + 1);
    for (uint64_t r = 0; r <= nreads; ++r) o2[r] = off[r] - base;
    if (nbytes) HIPCHK(hipMemcpyAsync(c->d_seq, seq + base, nbytes, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_off, o2.data(), (nreads + 1) * 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->d_loci, loci, nreads * 4, hipMemcpyHostToDevice, s));
    WalkArgs w;
    memset(&w, 0, sizeof(w));
    w.T = c->T; w.P = c->P; w.seq = c->d_seq; w.off = c->d_off;
    w.read_locus = c->d_loci; w.nreads = (uint32_t)nreads; w.trecs = c->d_trecs; w.errflag = c->d_small + 3;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(nreads, (uint64_t)c->walk_blocks);
    LAUNCH(k_walk_reads, dim3(grid), dim3(64), s, w);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(recs, c->d_trecs, nreads * sizeof(dbtk_thread_rec_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return DBTK_OK;
}

// What the walk decided for the pairs of the last host-buffer batch that reached threading, in pair order.
static dbtk_status_t dbtk_ctx_walk_results_impl(dbtk_ctx_t* c, dbtk_walk_res_t* res, dbtk_thread_rec_t* trecs, uint64_t cap, uint64_t* n) {
    if (!c || !n) { set_error("null argument"); return DBTK_ERR_ARG; }
    *n = 0;
    if (!c->last_walk_npairs) return DBTK_OK;
    if (trecs && !c->last_walk_recs) { set_error("thread records are only kept with params.trace"); return DBTK_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    uint32_t nsurv = 0;
    HIPCHK(hipMemcpy(&nsurv, c->d_small, 4, hipMemcpyDeviceToHost));
    const uint64_t np = c->last_walk_npairs;
    std::vector<uint32_t> dst(nsurv), ret(nsurv), surv(nsurv);
    if (nsurv) {
        HIPCHK(hipMemcpy(dst.data(), c->d_walk, (size_t)nsurv * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ret.data(), c->d_walk + np, (size_t)nsurv * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(surv.data(), c->d_sorted, (size_t)nsurv * 4, hipMemcpyDeviceToHost));
    }
    std::vector<std::pair<uint32_t, uint32_t>> walked;  // (pair, t)
    for (uint32_t t = 0; t < nsurv; ++t) if (dst[t] != NAN32) walked.emplace_back(surv[t], t);
    std::sort(walked.begin(), walked.end());
    *n = walked.size();
    if (walked.size() > cap) { set_error("result buffer too small"); return DBTK_ERR_OVERFLOW; }
    for (size_t i = 0; i < walked.size(); ++i) {
        const uint32_t t = walked[i].second;
        if (res) { res[i].pair = walked[i].first; res[i].dst = dst[t]; res[i].ret1 = (int8_t)(ret[t] & 0xFF); res[i].ret2 = (int8_t)((ret[t] >> 8) & 0xFF); res[i].pad[0] = res[i].pad[1] = 0; }
        if (trecs) HIPCHK(hipMemcpy(&trecs[2 * i], &c->d_trecs[2 * (size_t)t], 2 * sizeof(dbtk_thread_rec_t), hipMemcpyDeviceToHost));
    }
    return DBTK_OK;
}

// -a / -ae: the compact alignment records of the last host-buffer batch, invalid slots dropped, pair order.
static dbtk_status_t dbtk_ctx_aln_records_impl(dbtk_ctx_t* c, void* buf, uint64_t buf_bytes, uint64_t* nrec, uint32_t* stride, uint32_t* cap) {
    if (!c || !nrec || !stride || !cap) { set_error("null argument"); return DBTK_ERR_ARG; }
    *nrec = 0; *stride = c->aln_stride; *cap = c->aln_cap;
    if (!c->aln_max || !c->d_aln) return DBTK_OK;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    uint32_t nslots = 0;
    HIPCHK(hipMemcpy(&nslots, c->d_small + 4, 4, hipMemcpyDeviceToHost));
    if (nslots > c->aln_max) { set_error("alignment record buffer overflow"); return DBTK_ERR_OVERFLOW; }
    const size_t st = c->aln_stride;
    // too small for every slot handed out: say what is needed (an upper bound: some slots of a wave's last chunk stay empty), copy nothing
    if ((uint64_t)nslots * st > buf_bytes || (!buf && nslots)) { *nrec = nslots; set_error("alignment record buffer too small"); return DBTK_ERR_OVERFLOW; }
    if (!nslots) return DBTK_OK;
    if ((size_t)nslots * st > c->h_aln_bytes) {  // pinned staging, grown as needed and kept: the copy runs at link speed
        if (c->h_aln) HIPCHK(hipHostFree(c->h_aln));
        c->h_aln = nullptr; c->h_aln_bytes = 0;
        const size_t want = std::max((size_t)nslots * st, (size_t)c->aln_max * st);
        HIPCHK(hipHostMalloc((void**)&c->h_aln, want, hipHostMallocDefault));
        c->h_aln_bytes = want;
    }
    const uint8_t* raw = c->h_aln;
    HIPCHK(hipMemcpy(c->h_aln, c->d_aln, (size_t)nslots * st, hipMemcpyDeviceToHost));
    std::vector<std::pair<uint32_t, uint32_t>> order;  // (pair, slot)
    order.reserve(nslots);
    for (uint32_t i = 0; i < nslots; ++i) {
        const dbtk_aln_hdr_t* h = reinterpret_cast<const dbtk_aln_hdr_t*>(raw + (size_t)i * st);
        if (h->pair != NAN32) order.emplace_back(h->pair, i);
    }
    std::sort(order.begin(), order.end());
    *nrec = order.size();
    // gather into pair order; a record only as far as its arrays are filled (the tail of a slot is never read back by anyone)
    const size_t n = order.size();
    const unsigned nt = (unsigned)std::min<size_t>(std::max<size_t>(1, n / 16384), std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency())));
    auto gather = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) memcpy((uint8_t*)buf + i * st, raw + (size_t)order[i].second * st, st);
    };
    if (nt <= 1) gather(0, n);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t) th.emplace_back(gather, n * t / nt, n * (t + 1) / nt);
        for (auto& x : th) x.join();
    }
    return DBTK_OK;
}

// -a / -ae in text form: the per-pair index and the arena of the last host-buffer batch.
static dbtk_status_t dbtk_ctx_aln_text_impl(dbtk_ctx_t* c, uint32_t* idx, uint64_t idx_cap, void* arena, uint64_t arena_cap, uint64_t* used) {
    if (!c || !used) { set_error("null argument"); return DBTK_ERR_ARG; }
    *used = 0;
    if (!c->txt_cap || !c->d_txt) { for (uint64_t i = 0; idx && i < idx_cap; ++i) idx[i] = NAN32; return DBTK_OK; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    uint32_t cur = 0;
    HIPCHK(hipMemcpy(&cur, c->d_small + 7, 4, hipMemcpyDeviceToHost));
    if (cur > c->txt_cap) { set_error("alignment text arena overflow"); return DBTK_ERR_OVERFLOW; }
    *used = cur;
    if (cur > arena_cap || (!arena && cur)) { set_error("alignment text buffer too small"); return DBTK_ERR_OVERFLOW; }
    const uint64_t np = std::min<uint64_t>(idx_cap, c->last_walk_npairs);
    if (idx && np) HIPCHK(hipMemcpy(idx, c->d_txtidx, np * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint64_t i = np; idx && i < idx_cap; ++i) idx[i] = NAN32;
    if (cur) HIPCHK(hipMemcpy(arena, c->d_txt, cur, hipMemcpyDeviceToHost));
    return DBTK_OK;
}

dbtk_status_t dbtk_align_batch_device(dbtk_ctx_t* c, const void* d_seq, const void* d_offsets, uint64_t npairs,
                                      uint32_t max_read_len) {
    if (!c || !d_seq || !d_offsets) { set_error("null argument"); return DBTK_ERR_ARG; }
    if (max_read_len > DBTK_MAX_READ_LEN) { set_error("max_read_len > DBTK_MAX_READ_LEN"); return DBTK_ERR_READ_TOO_LONG; }
    if (((uintptr_t)d_seq & 15) != 0) { set_error("d_seq must be 16-byte aligned"); return DBTK_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    // d_seq is 16-byte aligned and device allocations are page-granular, so the aligned 16-byte
    // chunk holding the last base is always readable: no byte-wise tail needed (seq_len = max).
    if (c->two_lanes) switch_lane(c);  // successive batches alternate between the two streams
    return launch_batch(c, (const uint8_t*)d_seq, (const uint64_t*)d_offsets, ~0ull, npairs, max_read_len, nullptr, 0);
}

// The sticky error words of all lanes (a read longer than promised was truncated on the device: the accumulators are
// tainted from then on).  Reported once, then cleared.
static dbtk_status_t take_error_words(dbtk_ctx_t* c) {
    uint32_t err = 0, err2 = 0;
    HIPCHK(hipMemcpy(&err, c->d_small + 3, 4, hipMemcpyDeviceToHost));
    if (c->alt.d_small) HIPCHK(hipMemcpy(&err2, c->alt.d_small + 3, 4, hipMemcpyDeviceToHost));
    for (auto& l : c->parked) {
        uint32_t e3 = 0;
        if (l.d_small) HIPCHK(hipMemcpy(&e3, l.d_small + 3, 4, hipMemcpyDeviceToHost));
        if (e3) err2 = e3;
    }
    if (err || err2) {
        (void)hipMemset(c->d_small + 3, 0, 4);
        if (c->alt.d_small) (void)hipMemset(c->alt.d_small + 3, 0, 4);
        for (auto& l : c->parked) if (l.d_small) (void)hipMemset(l.d_small + 3, 0, 4);
        const uint32_t e = err ? err : err2;
        set_error(e == DBTK_ERR_READ_TOO_LONG ? "device reported an over-long read (the accumulators include truncated reads)"
                                              : "device reported an error during the batch");
        return (dbtk_status_t)e;
    }
    return DBTK_OK;
}

dbtk_status_t dbtk_ctx_synchronize(dbtk_ctx_t* c) {
    if (!c) { set_error("null argument"); return DBTK_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_all(c));
    return take_error_words(c);
}

dbtk_status_t dbtk_ctx_counts(dbtk_ctx_t* c, uint64_t* counts, uint64_t* kmc, uint32_t* nmapread, uint64_t* counters) {
    if (!c) { set_error("null argument"); return DBTK_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_all(c));
    { const dbtk_status_t es = take_error_words(c); if (es) return es; }  // never hand out tainted accumulators silently
    const uint64_t nloci = c->g->nloci;
    if (counts && c->ntr) HIPCHK(hipMemcpy(counts, c->d_accum, c->ntr * 8, hipMemcpyDeviceToHost));
    if (kmc && nloci) HIPCHK(hipMemcpy(kmc, c->d_accum + c->ntr, nloci * 8, hipMemcpyDeviceToHost));
    if (nmapread && nloci) {
        std::vector<uint64_t> w(nloci);
        HIPCHK(hipMemcpy(w.data(), c->d_accum + c->ntr + nloci, nloci * 8, hipMemcpyDeviceToHost));
        for (uint64_t l = 0; l < nloci; ++l) nmapread[l] = (uint32_t)w[l];  // atomic_uint32_t in the reference
    }
    if (counters) HIPCHK(hipMemcpy(counters, c->d_accum + c->ntr + 2 * nloci, DBTK_C_COUNT * 8, hipMemcpyDeviceToHost));
    return DBTK_OK;
}

dbtk_status_t dbtk_ctx_accum_buffer(dbtk_ctx_t* c, void** d_base, uint64_t* n_u64) {
    if (!c || !d_base || !n_u64) { set_error("null argument"); return DBTK_ERR_ARG; }
    *d_base = c->d_accum;
    *n_u64 = c->n_accum;
    return DBTK_OK;
}

dbtk_status_t dbtk_ctx_reset(dbtk_ctx_t* c) {
    if (!c) { set_error("null argument"); return DBTK_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(sync_all(c));
    HIPCHK(hipMemsetAsync(c->d_accum, 0, c->n_accum * 8, c->stream));
    HIPCHK(hipMemsetAsync(c->d_pstats, 0, DBTK_PATH_STATS * 8, c->stream));
    HIPCHK(hipMemsetAsync(c->d_small + 3, 0, 4, c->stream));  // a stale error word would be reported against the next run
    if (c->alt.d_small) HIPCHK(hipMemsetAsync(c->alt.d_small + 3, 0, 4, c->stream));
    for (auto& l : c->parked) if (l.d_small) HIPCHK(hipMemsetAsync(l.d_small + 3, 0, 4, c->stream));
    // pairs appended by dbtk_ingest_align_merged and not yet aligned belong to the run that is being discarded: they must not be
    // aligned into the fresh accumulators by the next append or flush
    c->m_bytes = 0; c->m_pairs = 0; c->m_maxlen = 0;
    c->alt.m_bytes = 0; c->alt.m_pairs = 0; c->alt.m_maxlen = 0;
    for (auto& l : c->parked) { l.m_bytes = 0; l.m_pairs = 0; l.m_maxlen = 0; }
    HIPCHK(sync_all(c));
    return DBTK_OK;
}

// OUT.bub.kmdb: dumpBubbles -> dumpKmerMapDB("bub", ..., th = 5) (src/aQueryFasta_thread.h:999-1008):
// flattenKmapDB keeps the entries with count >= 5 in map iteration order (src/binaryKmerIO.hpp:31-51), then
// serializeKmapDB writes u64 nloci | u64 index[nloci] | u64 nk | u64 sizeof(val) = 8 | u64 ks[nk] | u64 vs[nk].
static dbtk_status_t dbtk_ctx_write_bubbles_impl(dbtk_ctx_t* c, const char* out_prefix) {
    if (!c || !out_prefix) { set_error("null argument"); return DBTK_ERR_ARG; }
    if (!c->P.bubbles) { set_error("context was not created with params.bubbles"); return DBTK_ERR_ARG; }
    const uint64_t nloci = c->g->nloci;
    std::vector<uint64_t> index(nloci), ks, vs;
    for (uint64_t l = 0; l < nloci; ++l) {
        uint64_t kept = 0;
        for (auto& p : c->bubbleDB[l])
            if ((int)p.second >= 5) { ks.push_back(p.first); vs.push_back(p.second); ++kept; }
        index[l] = kept;
    }
    const std::string fn = std::string(out_prefix) + ".bub.kmdb";
    FILE* f = fopen(fn.c_str(), "wb");
    if (!f) { set_error("cannot create " + fn); return DBTK_ERR_IO; }
    const uint64_t nk = ks.size(), szv = 8;
    bool ok = fwrite(&nloci, 8, 1, f) == 1 && (nloci == 0 || fwrite(index.data(), 8, nloci, f) == nloci) && fwrite(&nk, 8, 1, f) == 1 &&
              fwrite(&szv, 8, 1, f) == 1 && (nk == 0 || (fwrite(ks.data(), 8, nk, f) == nk && fwrite(vs.data(), 8, nk, f) == nk));
    fclose(f);
    if (!ok) { set_error("write failed: " + fn); return DBTK_ERR_IO; }
    return DBTK_OK;
}

// OUT.btk.kmdb: dumpBaitKmerHits -> dumpKmerMapDB("btk", ..., th = 0) (src/aQueryFasta_thread.h:998-1012): every entry, map iteration
// order, serializeKmapDB layout u64 nloci | u64 index[nloci] | u64 nk | u64 sizeof(val) = 8 | u64 ks[nk] | u64 vs[nk].
static dbtk_status_t dbtk_ctx_write_bait_hits_impl(dbtk_ctx_t* c, const char* out_prefix) {
    if (!c || !out_prefix) { set_error("null argument"); return DBTK_ERR_ARG; }
    if (!c->P.trackbait) { set_error("context was not created with params.trackbait"); return DBTK_ERR_ARG; }
    const uint64_t nloci = c->g->nloci;
    std::vector<uint64_t> index(nloci), ks, vs;
    for (uint64_t l = 0; l < nloci; ++l) {
        for (auto& p : c->btTK[l]) { ks.push_back(p.first); vs.push_back(p.second); }
        index[l] = c->btTK[l].size();
    }
    const std::string fn = std::string(out_prefix) + ".btk.kmdb";
    FILE* f = fopen(fn.c_str(), "wb");
    if (!f) { set_error("cannot create " + fn); return DBTK_ERR_IO; }
    const uint64_t nk = ks.size(), szv = 8;
    bool ok = fwrite(&nloci, 8, 1, f) == 1 && (nloci == 0 || fwrite(index.data(), 8, nloci, f) == nloci) && fwrite(&nk, 8, 1, f) == 1 &&
              fwrite(&szv, 8, 1, f) == 1 && (nk == 0 || (fwrite(ks.data(), 8, nk, f) == nk && fwrite(vs.data(), 8, nk, f) == nk));
    fclose(f);
    if (!ok) { set_error("write failed: " + fn); return DBTK_ERR_IO; }
    return DBTK_OK;
}
dbtk_status_t dbtk_ctx_merge_bait_hits(dbtk_ctx_t* dst, dbtk_ctx_t* src) {
    if (!dst || !src || dst->btTK.size() != src->btTK.size()) { set_error("bad argument"); return DBTK_ERR_ARG; }
    for (size_t l = 0; l < src->btTK.size(); ++l)
        for (auto& q : src->btTK[l]) dst->btTK[l][q.first] += q.second;
    return DBTK_OK;
}

// Multi-GPU: fold src's bubble DB into dst's (after the run; the reference's order is unspecified for -p > 1).
dbtk_status_t dbtk_ctx_merge_bubbles(dbtk_ctx_t* dst, dbtk_ctx_t* src) {
    if (!dst || !src || dst->bubbleDB.size() != src->bubbleDB.size()) { set_error("bad argument"); return DBTK_ERR_ARG; }
    for (size_t l = 0; l < src->bubbleDB.size(); ++l)
        for (auto& q : src->bubbleDB[l]) dst->bubbleDB[l][q.first] += q.second;
    return DBTK_OK;
}

int dbtk_ctx_kernel_times(dbtk_ctx_t* c, const char** names, double* total_ms, uint64_t* launches, int cap) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)sync_all(c);
    int n = 0;
    for (int i = 0; i < NKERN && n < cap; ++i) {
        fold_timer(c->timed[i]);
        names[n] = c->timed[i].name;
        total_ms[n] = c->timed[i].total_ms;
        launches[n] = c->timed[i].launches;
        ++n;
    }
    return n;
}

#ifdef DBTK_STAMPS
// diagnostic build only: the 16 per-phase cycle sums of k_pair since context creation
int dbtk_debug_stamps(dbtk_ctx_t* c, uint64_t* out16) {
    (void)hipSetDevice(c->device);
    (void)sync_all(c);
    return (int)hipMemcpy(out16, c->d_small + 32, 48 * 8, hipMemcpyDeviceToHost);
}
#endif

void dbtk_ctx_timers_enable(dbtk_ctx_t* c, int on) {
    if (!c) return;
    c->timers_on = on > 0;
    c->timers_every = on > 1 ? (uint32_t)on : 1u;
    c->batch_no = 0;
}

void dbtk_ctx_timers_reset(dbtk_ctx_t* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)sync_all(c);
    for (int i = 0; i < NKERN; ++i) { c->timed[i].used = 0; c->timed[i].total_ms = 0; c->timed[i].launches = 0; }
}

// One RCCL all-reduce (sum, uint64) over the accumulator buffers of n contexts
// in this process: the cross-GPU form of the reference's shared atomics
// (src/aQueryFasta_thread.cpp:2146-2158, 1887-1895).  librccl is loaded on first
// use so that single-GPU runs do not pay for it.
dbtk_status_t dbtk_allreduce(dbtk_ctx_t** ctxs, int n) {
    if (!ctxs || n <= 0) { set_error("null argument"); return DBTK_ERR_ARG; }
    if (n == 1) return DBTK_OK;
    // RCCL is bound at run time (dlopen: a one-GPU caller needs no librccl), but its TYPES come from <rccl/rccl.h>: the function
    // pointer types are decltype()s of the header's declarations and the enum values the header's own, so a changed signature or
    // renumbered ncclDataType_t is a compile error here, not a silently wrong sum on the first 8-GPU run.
    typedef ncclComm_t comm_t;
    typedef decltype(&ncclCommInitAll) commInitAll_t;
    typedef decltype(&ncclAllReduce) allReduce_t;
    typedef decltype(&ncclGroupStart) group_t;
    typedef decltype(&ncclCommDestroy) commDestroy_t;
    static_assert(sizeof(uint64_t) == 8, "the accumulators are summed as ncclUint64");
    for (int i = 0; i < n; ++i) {
        if (ctxs[i]->n_accum != ctxs[0]->n_accum) { set_error("contexts belong to different RPGGs"); return DBTK_ERR_ARG; }
        HIPCHK(hipSetDevice(ctxs[i]->device));
        HIPCHK(sync_all(ctxs[i]));
        const dbtk_status_t es = take_error_words(ctxs[i]);  // do not spread tainted accumulators over the other GPUs
        if (es) return es;
    }
    // Contexts that share a device (several ingest pipelines per GPU; a one-GPU box standing in for several) are summed on that device
    // first, into the first of them; RCCL then runs over one context per DISTINCT device (it refuses a device twice); at the end
    // every context of a device gets its representative's sum.
    std::vector<int> rep;  // index of the first context of each distinct device
    std::vector<int> rep_of(n, 0);
    for (int i = 0; i < n; ++i) {
        int r = -1;
        for (size_t q = 0; q < rep.size(); ++q) if (ctxs[rep[q]]->device == ctxs[i]->device) r = (int)q;
        if (r < 0) { rep.push_back(i); r = (int)rep.size() - 1; }
        rep_of[i] = rep[r];
    }
    for (int i = 0; i < n; ++i)
        if (rep_of[i] != i) {
            dbtk_ctx* d = ctxs[rep_of[i]];
            HIPCHK(hipSetDevice(d->device));
            LAUNCH(k_accum_add, dim3(1024), dim3(256), d->stream, d->d_accum, ctxs[i]->d_accum, d->n_accum);
            HIPCHK(hipStreamSynchronize(d->stream));
        }
    const int nd = (int)rep.size();
    int rc = 0;
    if (nd > 1) {
        static void* lib = nullptr;
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) { set_error(std::string("cannot load librccl: ") + dlerror()); return DBTK_ERR_HIP; }
        auto commInitAll = (commInitAll_t)dlsym(lib, "ncclCommInitAll");
        auto allReduce = (allReduce_t)dlsym(lib, "ncclAllReduce");
        auto groupStart = (group_t)dlsym(lib, "ncclGroupStart");
        auto groupEnd = (group_t)dlsym(lib, "ncclGroupEnd");
        auto commDestroy = (commDestroy_t)dlsym(lib, "ncclCommDestroy");
        if (!commInitAll || !allReduce || !groupStart || !groupEnd || !commDestroy) { set_error("librccl lacks the expected symbols"); return DBTK_ERR_HIP; }
        std::vector<comm_t> comms(nd);
        std::vector<int> devs(nd);
        for (int q = 0; q < nd; ++q) devs[q] = ctxs[rep[q]]->device;
        if (commInitAll(comms.data(), nd, devs.data()) != ncclSuccess) { set_error("ncclCommInitAll failed"); return DBTK_ERR_HIP; }
        rc = (int)groupStart();
        for (int q = 0; q < nd && !rc; ++q) {
            dbtk_ctx* c = ctxs[rep[q]];
            (void)hipSetDevice(c->device);
            rc = (int)allReduce(c->d_accum, c->d_accum, c->n_accum, ncclUint64, ncclSum, comms[q], c->stream);
        }
        rc |= (int)groupEnd();
        for (int q = 0; q < nd; ++q) {
            (void)hipSetDevice(ctxs[rep[q]]->device);
            (void)hipStreamSynchronize(ctxs[rep[q]]->stream);
            commDestroy(comms[q]);
        }
    }
    for (int i = 0; i < n && !rc; ++i)
        if (rep_of[i] != i) {
            HIPCHK(hipSetDevice(ctxs[i]->device));
            HIPCHK(hipMemcpy(ctxs[i]->d_accum, ctxs[rep_of[i]]->d_accum, ctxs[i]->n_accum * 8, hipMemcpyDeviceToDevice));
        }
    if (rc) { set_error("ncclAllReduce failed"); return DBTK_ERR_HIP; }
    return DBTK_OK;
}

// ---- raw-bytes ingest (include/dbtk.h: dbtk_ingest_*; kernels in dbtk_ingest.h) ------------------------------------------
struct dbtk_ingest {
    dbtk_ctx* c = nullptr;
    uint32_t L = 2, min_read = 0, nslots = 0, head = 1u << 20, line_cap = 0, pair_cap = 0;
    uint64_t chunk = 0;
    bool with_spans = false, with_qual = false, dead = false;
    hipStream_t stream = nullptr;       // the parse kernels, block after block
    hipStream_t copy_stream = nullptr;  // the host-to-device copies: block i + 1's runs while block i is parsed
    uint64_t raw_bytes = 0;
    std::unique_ptr<std::mutex[]> slot_m;  // (pinning a slot's host buffer)
    uint32_t* d_basew = nullptr;  // per slot: where its block starts (written by the block before)
    IngestHdr* d_hdr = nullptr;   // per slot
    IngestHdr* h_hdr = nullptr;   // pinned, per slot
    uint64_t submitted = 0;       // input bytes submitted so far
    uint64_t nsubmitted = 0, nwaited = 0;
    int last_byte = '\n';
    // -a / -ae lines (dbtk_ingest_aln_lines): device buffers per aligning context (several contexts may work on different slots at once)
    struct LinesBuf {
        uint32_t* d_linelen = nullptr; uint64_t linelen_cap = 0;
        uint8_t* d_text = nullptr; uint64_t text_cap = 0;
        uint8_t* d_gzout = nullptr; uint64_t gzout_cap = 0;
        uint8_t* d_packed = nullptr; uint64_t packed_cap = 0;
        uint32_t* d_outlen = nullptr; uint64_t outlen_cap = 0;
        uint64_t* d_totals = nullptr;   // text bytes, lines, packed bytes
        uint64_t* h_totals = nullptr;   // pinned
    };
    std::map<dbtk_ctx*, LinesBuf> lines;
    std::mutex lines_m;
    uint32_t* d_crctab = nullptr;
    std::vector<uint8_t> carry_host;  // host copy of the bytes the last waited block handed on (the spans of the next block reach into them)
    struct Slot {
        uint8_t* h_raw = nullptr; uint8_t* d_raw = nullptr;
        uint32_t* d_tile = nullptr; uint32_t* d_nlpos = nullptr; uint32_t* d_pk = nullptr; uint32_t* d_kept = nullptr;
        uint64_t* d_off = nullptr; uint8_t* d_flat = nullptr; uint8_t* d_qual = nullptr; dbtk_ingest_span_t* d_spans = nullptr;
        hipEvent_t parsed = nullptr, aligned = nullptr, copied = nullptr;
        uint8_t* h_lines = nullptr; uint64_t h_lines_cap = 0;  // pinned: the block's -a / -ae lines as the caller's writer sees them
        bool has_aligned = false, waited = false, pending = false;
        uint32_t end = 0; uint64_t file_off = 0; int last = 0;
        IngestHdr hdr;
    };
    std::vector<Slot> slots;
    // slot -> dbtk_ingest_aln_lines calls in progress on it: k_aln_write reads the slot's raw block (its carried-over head included) on an
    // aligning context's stream, while a submit copies new bytes into that buffer on the copy stream and the block before it carries its tail
    // into the buffer's head on the parse stream.  aln_lines returns only when its kernels are done, so "in progress" is exact; a submit that
    // would overwrite bytes still being read is refused instead of corrupting the -a / -ae lines.
    std::unique_ptr<std::atomic<int>[]> lines_busy;
};

// Pinned host buffers made ahead of the ingest that uses them (dbtk_ingest_reserve_host): pinning runs at ~4 GB/s and holds a lock of the
// runtime that every other allocation of the process waits for, so a caller with something else to do first (the command line: parsing the
// RPGG files) pins beside that.  A buffer goes to the smallest pooled one that holds it; what the pool cannot serve is pinned on first use
// as before; an ingest hands its buffers back when it is freed.
namespace {
struct PinnedPool {
    std::mutex m;
    std::vector<std::pair<uint8_t*, uint64_t>> free;
} g_pinned;
uint8_t* pinned_take(uint64_t need, uint64_t* cap) {
    std::lock_guard<std::mutex> l(g_pinned.m);
    size_t best = ~(size_t)0;
    for (size_t i = 0; i < g_pinned.free.size(); ++i)
        if (g_pinned.free[i].second >= need && (best == ~(size_t)0 || g_pinned.free[i].second < g_pinned.free[best].second)) best = i;
    if (best == ~(size_t)0) return nullptr;
    uint8_t* p = g_pinned.free[best].first;
    *cap = g_pinned.free[best].second;
    g_pinned.free.erase(g_pinned.free.begin() + (long)best);
    return p;
}
void pinned_give(uint8_t* p, uint64_t cap) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> l(g_pinned.m);
        uint64_t held = 0;
        for (auto& b : g_pinned.free) held += b.second;
        // (the pool is a hand-over between a reservation and an ingest, and between one ingest and the next: not a store without bounds —
        // a process that makes ingests of many shapes gets its memory back beyond 128 buffers or 4 GB)
        if (g_pinned.free.size() < 128 && held + cap <= (4ull << 30)) { g_pinned.free.emplace_back(p, cap); return; }
    }
    (void)(hipHostFree)(p);
}
inline uint32_t ingest_head_bytes(uint64_t chunk_bytes) { return (uint32_t)std::min<uint64_t>(1u << 20, (chunk_bytes + 15) & ~15ull); }
inline uint64_t ingest_raw_bytes(uint64_t chunk_bytes) { return ((uint64_t)ingest_head_bytes(chunk_bytes) + chunk_bytes + 1 + 63) & ~63ull; }
}  // namespace

static void ingest_free_impl(dbtk_ingest* g) {
    if (!g) return;
    if (g->c) (void)hipSetDevice(g->c->device);
    if (g->copy_stream) (void)hipStreamSynchronize(g->copy_stream);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    if (g->c) (void)sync_all(g->c);
    for (auto& S : g->slots) {
        pinned_give(S.h_raw, g->raw_bytes);
        pinned_give(S.h_lines, S.h_lines_cap);
        void* ptrs[] = {S.d_raw, S.d_tile, S.d_nlpos, S.d_pk, S.d_kept, S.d_off, S.d_flat, S.d_qual, S.d_spans};
        for (void* p : ptrs) if (p) (void)hipFree(p);
        if (S.parsed) (void)hipEventDestroy(S.parsed);
        if (S.aligned) (void)hipEventDestroy(S.aligned);
        if (S.copied) (void)hipEventDestroy(S.copied);
    }
    for (auto& kv : g->lines) {
        dbtk_ingest::LinesBuf& B = kv.second;
        void* ptrs[] = {B.d_linelen, B.d_text, B.d_gzout, B.d_packed, B.d_outlen, B.d_totals};
        for (void* p : ptrs) if (p) (void)hipFree(p);
        if (B.h_totals) (void)hipHostFree(B.h_totals);
    }
    if (g->d_crctab) (void)hipFree(g->d_crctab);
    if (g->d_basew) (void)hipFree(g->d_basew);
    if (g->d_hdr) (void)hipFree(g->d_hdr);
    if (g->h_hdr) (void)hipHostFree(g->h_hdr);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    if (g->copy_stream) (void)hipStreamDestroy(g->copy_stream);
    delete g;
}

static dbtk_status_t dbtk_ingest_create_impl(dbtk_ctx_t* c, uint32_t fastq, uint32_t min_read_size, uint64_t chunk_bytes, uint32_t nslots,
                                             uint32_t with_spans, dbtk_ingest_t** out) {
    if (!c || !out) { set_error("null argument"); return DBTK_ERR_ARG; }
    *out = nullptr;
    if (nslots < 2 || nslots > 64) { set_error("dbtk_ingest_create: 2 .. 64 slots"); return DBTK_ERR_ARG; }
    if (chunk_bytes < 4096 || chunk_bytes > (1ull << 30)) { set_error("dbtk_ingest_create: chunk_bytes must be 4 KB .. 1 GB"); return DBTK_ERR_ARG; }
    if (c->P.trackbait) { set_error("dbtk_ingest: -tb replays its batches on the host and needs the reads in host buffers"); return DBTK_ERR_UNSUPPORTED; }
    HIPCHK(hipSetDevice(c->device));
    dbtk_ingest* g = new dbtk_ingest;
    g->c = c; g->L = fastq ? 4 : 2; g->min_read = min_read_size; g->nslots = nslots; g->chunk = chunk_bytes;
    g->head = ingest_head_bytes(chunk_bytes);
    g->with_spans = with_spans != 0;
    g->with_qual = fastq && c->P.bait;  // base qualities only matter to the bait filter
    const uint64_t raw_bytes = ingest_raw_bytes(chunk_bytes);
    g->raw_bytes = raw_bytes;
    g->line_cap = (uint32_t)(raw_bytes / 8 + 64);
    g->pair_cap = g->line_cap / (2 * g->L) + 1;
    const uint64_t ntiles = raw_bytes / ING_TILE + 1;
    g->slots.resize(nslots);
    g->slot_m.reset(new std::mutex[nslots]);
    g->lines_busy.reset(new std::atomic<int>[nslots]);
    for (uint32_t i = 0; i < nslots; ++i) g->lines_busy[i].store(0);
    dbtk_status_t st = DBTK_OK;
    auto chk = [&](hipError_t e, const char* what) { if (e != hipSuccess && !st) { set_error(std::string(what) + ": " + hipGetErrorString(e)); st = DBTK_ERR_HIP; } };
    chk(hipStreamCreate(&g->stream), "hipStreamCreate");
    chk(hipStreamCreate(&g->copy_stream), "hipStreamCreate");
    chk(hipMalloc(&g->d_basew, nslots * 4), "hipMalloc");
    chk(hipMalloc(&g->d_hdr, nslots * sizeof(IngestHdr)), "hipMalloc");
    chk(hipHostMalloc((void**)&g->h_hdr, nslots * sizeof(IngestHdr), hipHostMallocDefault), "hipHostMalloc");
    for (auto& S : g->slots) {
        chk(hipMalloc(&S.d_raw, raw_bytes), "hipMalloc (block)");  // (the pinned host buffer: on first use, by the thread that fills it — dbtk_ingest_chunk_buffer)
        chk(hipMalloc(&S.d_tile, (ntiles + 1 + ING_SCAN_BLOCKS) * 4), "hipMalloc");
        chk(hipMalloc(&S.d_nlpos, (uint64_t)g->line_cap * 4), "hipMalloc (line table)");
        chk(hipMalloc(&S.d_pk, ((uint64_t)g->pair_cap + 2 * ING_SCAN_BLOCKS) * 4), "hipMalloc");
        chk(hipMalloc(&S.d_kept, (uint64_t)g->pair_cap * 4), "hipMalloc");
        chk(hipMalloc(&S.d_off, (2 * (uint64_t)g->pair_cap + 1) * 8), "hipMalloc");
        chk(hipMalloc(&S.d_flat, raw_bytes + 64), "hipMalloc (reads)");
        if (g->with_qual) chk(hipMalloc(&S.d_qual, raw_bytes + 64), "hipMalloc (qualities)");
        if (g->with_spans) chk(hipMalloc(&S.d_spans, (uint64_t)g->pair_cap * sizeof(dbtk_ingest_span_t)), "hipMalloc (spans)");
        chk(hipEventCreateWithFlags(&S.parsed, hipEventDisableTiming), "hipEventCreate");
        chk(hipEventCreateWithFlags(&S.aligned, hipEventDisableTiming), "hipEventCreate");
        chk(hipEventCreateWithFlags(&S.copied, hipEventDisableTiming), "hipEventCreate");
        if (st) break;
    }
    if (st) { ingest_free_impl(g); return st; }
    *out = g;
    return DBTK_OK;
}

static dbtk_status_t dbtk_ingest_submit_impl(dbtk_ingest_t* g, uint32_t slot, uint64_t nbytes, int last) {
    if (!g) { set_error("null argument"); return DBTK_ERR_ARG; }
    if (g->dead) { set_error("dbtk_ingest: a flagged block ended this ingest"); return DBTK_ERR_ARG; }
    if (slot != g->nsubmitted % g->nslots) { set_error("dbtk_ingest_submit: slots are used round robin"); return DBTK_ERR_ARG; }
    if (nbytes > g->chunk) { set_error("dbtk_ingest_submit: more than chunk_bytes"); return DBTK_ERR_ARG; }
    dbtk_ingest::Slot& S = g->slots[slot];
    if (S.pending) { set_error("dbtk_ingest_submit: the slot's previous block has not been waited for"); return DBTK_ERR_ARG; }
    if (!S.h_raw && !dbtk_ingest_chunk_buffer(g, slot)) { set_error("hipHostMalloc (chunk buffer) failed"); return DBTK_ERR_NOMEM; }
    if (g->lines_busy[slot].load() || g->lines_busy[(slot + 1) % g->nslots].load()) {
        set_error("dbtk_ingest_submit: dbtk_ingest_aln_lines is still reading this slot's block (or the next slot's, whose head this block's carry-over overwrites)");
        return DBTK_ERR_ARG;
    }
    dbtk_ctx* c = g->c;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = g->stream;
    uint64_t n = nbytes;
    if (nbytes) g->last_byte = S.h_raw[g->head + nbytes - 1];
    if (last && g->last_byte != '\n') { S.h_raw[g->head + n++] = '\n'; g->last_byte = '\n'; }  // std::getline: the last line need not end with a newline
    if (S.has_aligned) HIPCHK(hipStreamWaitEvent(s, S.aligned, 0));  // the align kernels of the slot's previous block still read its arrays
    // the copy on its own stream (the slot's previous block has been waited for: nothing reads its bytes any more); the kernels wait for it
    if (n) HIPCHK(hipMemcpyAsync(S.d_raw + g->head, S.h_raw + g->head, n, hipMemcpyHostToDevice, g->copy_stream));
    HIPCHK(hipEventRecord(S.copied, g->copy_stream));
    HIPCHK(hipStreamWaitEvent(s, S.copied, 0));
    if (g->nsubmitted == 0) HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(g->d_basew + slot), (int)g->head, 1, s));
    HIPCHK(hipMemsetAsync(g->d_hdr + slot, 0, sizeof(IngestHdr), s));
    const uint32_t next = (slot + 1) % g->nslots;
    IngestArgs a;
    memset(&a, 0, sizeof(a));
    a.raw = S.d_raw; a.base_in = g->d_basew + slot; a.end = g->head + (uint32_t)n; a.L = g->L; a.min_read = g->min_read; a.last = last ? 1u : 0u;
    a.tile_cnt = S.d_tile; a.nlpos = S.d_nlpos; a.line_cap = g->line_cap; a.pk = S.d_pk; a.kept = S.d_kept; a.off = S.d_off;
    a.flat = S.d_flat; a.qual = S.d_qual; a.spans = S.d_spans; a.hdr = g->d_hdr + slot;
    a.next_raw = g->slots[next].d_raw; a.base_out = g->d_basew + next; a.head = g->head;
    const uint32_t ntiles = (a.end + ING_TILE - 1) / ING_TILE;
    const uint32_t gt = std::min<uint32_t>(ntiles, (uint32_t)c->num_cu * 16);
    LAUNCH(k_ing_count, dim3(gt), dim3(64), s, a);
    LAUNCH(k_ing_scan, dim3(ING_SCAN_BLOCKS), dim3(64), s, a, 0);
    LAUNCH(k_ing_scan, dim3(ING_SCAN_BLOCKS), dim3(64), s, a, 1);
    LAUNCH(k_ing_lines, dim3(gt), dim3(64), s, a);
    LAUNCH(k_ing_pairs, dim3(c->num_cu * 4), dim3(256), s, a);
    LAUNCH(k_ing_place, dim3(ING_SCAN_BLOCKS), dim3(64), s, a, 0);
    LAUNCH(k_ing_place, dim3(ING_SCAN_BLOCKS), dim3(64), s, a, 1);
    LAUNCH(k_ing_gather, dim3(c->num_cu * 32), dim3(64), s, a);
    LAUNCH(k_ing_carry, dim3(1), dim3(256), s, a);
    HIPCHK(hipMemcpyAsync(g->h_hdr + slot, g->d_hdr + slot, sizeof(IngestHdr), hipMemcpyDeviceToHost, s));
    HIPCHK(hipEventRecord(S.parsed, s));
    S.end = a.end; S.file_off = g->submitted; S.last = last; S.pending = true; S.waited = false; S.has_aligned = false;
    g->submitted += nbytes;
    ++g->nsubmitted;
    return DBTK_OK;
}

static dbtk_status_t dbtk_ingest_wait_impl(dbtk_ingest_t* g, uint32_t slot, dbtk_ingest_info_t* info) {
    if (!g || !info) { set_error("null argument"); return DBTK_ERR_ARG; }
    if (slot >= g->nslots || slot != g->nwaited % g->nslots || !g->slots[slot].pending) { set_error("dbtk_ingest_wait: blocks are waited for in the order they were submitted"); return DBTK_ERR_ARG; }
    dbtk_ingest::Slot& S = g->slots[slot];
    HIPCHK(hipSetDevice(g->c->device));
    HIPCHK(hipEventSynchronize(S.parsed));
    S.hdr = g->h_hdr[slot];
    S.pending = false; S.waited = true;
    ++g->nwaited;
    const IngestHdr& h = S.hdr;
    // the host's copy of the block gets the bytes the block before handed on (the device copied its own)
    if (!g->carry_host.empty() && g->carry_host.size() == g->head - h.base) memcpy(S.h_raw + h.base, g->carry_host.data(), g->carry_host.size());
    g->carry_host.clear();
    uint32_t flags = h.flags;
    if (h.cut >= h.base && h.cut <= S.end && h.carry <= g->head && !S.last) g->carry_host.assign(S.h_raw + h.cut, S.h_raw + S.end);
    if (flags) g->dead = true;
    info->flags = flags; info->npairs = h.npairs; info->nkept = h.nkept; info->max_read_len = (uint32_t)h.maxlen;
    info->first_byte = S.file_off - (g->head - h.base);
    info->cut_byte = info->first_byte + (h.cut - h.base);
    info->seq_bytes = h.flat_bytes;
    return DBTK_OK;
}

static dbtk_status_t ingest_ctx(dbtk_ingest_t* g, dbtk_ctx_t* cx, dbtk_ctx** out) {
    *out = cx ? cx : g->c;
    if (cx && (cx->device != g->c->device || cx->g != g->c->g)) { set_error("dbtk_ingest: the context must be on the ingest's device and of its RPGG"); return DBTK_ERR_ARG; }
    return DBTK_OK;
}
static dbtk_status_t dbtk_ingest_align_impl(dbtk_ingest_t* g, uint32_t slot, dbtk_ctx_t* cx, int sync, dbtk_pair_rec_t* recs, uint64_t rec_cap, uint64_t* nrec) {
    if (!g || slot >= g->nslots) { set_error("null argument"); return DBTK_ERR_ARG; }
    if (nrec) *nrec = 0;
    dbtk_ingest::Slot& S = g->slots[slot];
    if (!S.waited) { set_error("dbtk_ingest_align: wait for the block first"); return DBTK_ERR_ARG; }
    const IngestHdr& h = S.hdr;
    if (h.flags & (ING_F_DIRTY | ING_F_LINES_OVF)) { set_error("dbtk_ingest_align: the block is not a run of adjacent mates; continue with a host reader at info.first_byte"); return DBTK_ERR_ARG; }
    if (h.nkept == 0) return DBTK_OK;
    if (h.maxlen > DBTK_MAX_READ_LEN) {
        set_error("read longer than DBTK_MAX_READ_LEN (256): the reference's PE_KMC is uint8_t, src/aQueryFasta_thread.cpp:42");
        return DBTK_ERR_READ_TOO_LONG;
    }
    dbtk_ctx* c = nullptr;
    { const dbtk_status_t sc = ingest_ctx(g, cx, &c); if (sc) return sc; }
    HIPCHK(hipSetDevice(c->device));
    if (sync) return run_batch_sync(c, S.d_flat, S.d_off, S.d_qual, ~0ull, h.nkept, (uint32_t)h.maxlen, nullptr, nullptr, nullptr, recs, rec_cap, nrec);
    if (c->P.bubbles) { set_error("dbtk_ingest_align: -bu is replayed batch by batch on the host: sync = 1"); return DBTK_ERR_ARG; }
    if (c->two_lanes) switch_lane(c);
    const dbtk_status_t st = launch_batch(c, S.d_flat, S.d_off, ~0ull, h.nkept, (uint32_t)h.maxlen, nullptr, 0, S.d_qual);
    if (st) return st;
    HIPCHK(hipEventRecord(S.aligned, c->stream));
    S.has_aligned = true;
    return DBTK_OK;
}

// The parsed block of `slot` appended to the context's MERGED batch (device-to-device: the reads back to back behind the ones already
// there, the offsets rebased), which is aligned once it holds min_pairs pairs — or now, with flush.  A 32-MB block is ~100 000 pairs;
// the locus-resident kernels (dbtk_locus.h) want a batch with LOC_MIN_PAIRS pairs per locus: at release scale two million.  The
// reference batches 300 000 reads per thread whatever the content (src/aQueryFasta_thread.cpp:1918-1976); pairs are independent and
// every effect of one is an integer add, so how the pairs are cut into batches changes no result.  No records, no -b qualities,
// no -bu (those go block by block through dbtk_ingest_align).  slot = ~0u: nothing to append (flush only).
static dbtk_status_t dbtk_ingest_align_merged_impl(dbtk_ingest_t* g, uint32_t slot, dbtk_ctx_t* cx, uint64_t min_pairs, int flush) {
    if (!g || (slot != ~0u && slot >= g->nslots)) { set_error("null argument"); return DBTK_ERR_ARG; }
    dbtk_ctx* c = nullptr;
    { const dbtk_status_t sc = ingest_ctx(g, cx, &c); if (sc) return sc; }
    HIPCHK(hipSetDevice(c->device));
    if (c->P.bubbles || c->P.trace || (c->P.bait && g->with_qual)) { set_error("dbtk_ingest_align_merged: records, -bu and -b with qualities go block by block (dbtk_ingest_align)"); return DBTK_ERR_ARG; }
    hipStream_t s = c->stream;
    if (slot != ~0u) {
        dbtk_ingest::Slot& S = g->slots[slot];
        if (!S.waited) { set_error("dbtk_ingest_align_merged: wait for the block first"); return DBTK_ERR_ARG; }
        const IngestHdr& h = S.hdr;
        if (h.flags & (ING_F_DIRTY | ING_F_LINES_OVF)) { set_error("dbtk_ingest_align_merged: the block is not a run of adjacent mates; continue with a host reader at info.first_byte"); return DBTK_ERR_ARG; }
        if (h.maxlen > DBTK_MAX_READ_LEN) {
            set_error("read longer than DBTK_MAX_READ_LEN (256): the reference's PE_KMC is uint8_t, src/aQueryFasta_thread.cpp:42");
            return DBTK_ERR_READ_TOO_LONG;
        }
        if (h.nkept) {
            const uint64_t nb = h.flat_bytes, nr = 2ull * h.nkept;
            // room: sized ONCE, for a whole merged batch — min_pairs pairs of this block's longest read plus the block that takes it past
            // that (min_pairs itself is bounded: 2^26 pairs, 20 GB of 150-bp reads) — and grown, copying what is there, only when
            // later blocks hold longer reads.  An allocation that fails (beside 28 - 47 GB of tables) is not the end of the run: the
            // pairs merged so far are aligned now, and the block starts a new batch in the room there is.
            const uint64_t mp = std::min<uint64_t>(std::max<uint64_t>(min_pairs, 1), 1ull << 26);
            if (c->m_bytes + nb + 64 > c->m_flat_cap || 2 * c->m_pairs + nr + 1 > c->m_off_cap) {
                const uint64_t want_f = std::max<uint64_t>({(c->m_bytes + nb + 64) * 5 / 4, 64ull << 20, mp * 2 * h.maxlen + nb + 64});
                const uint64_t want_o = std::max<uint64_t>({(2 * c->m_pairs + nr + 1) * 5 / 4, 1ull << 20, 2 * mp + nr + 1});
                uint8_t* nf = nullptr; uint64_t* no = nullptr;
                const bool grow_f = c->m_bytes + nb + 64 > c->m_flat_cap, grow_o = 2 * c->m_pairs + nr + 1 > c->m_off_cap;
                bool ok = (!grow_f || hipMalloc(&nf, want_f) == hipSuccess) && (!grow_o || hipMalloc(&no, want_o * 8) == hipSuccess);
                if (!ok) {
                    (void)hipGetLastError();
                    if (nf) (void)hipFree(nf);
                    if (no) (void)hipFree(no);
                    nf = nullptr; no = nullptr;
                    if (c->m_pairs) {  // align what is there, wait for it, start over in the same buffers
                        const dbtk_status_t st = launch_batch(c, c->m_flat, c->m_off, ~0ull, c->m_pairs, c->m_maxlen, nullptr, 0, nullptr);
                        if (st) return st;
                        HIPCHK(hipStreamSynchronize(s));
                        c->m_bytes = 0; c->m_pairs = 0; c->m_maxlen = 0;
                    }
                    if (nb + 64 > c->m_flat_cap) { if (c->m_flat) HIPCHK(hipFree(c->m_flat)); c->m_flat = nullptr; c->m_flat_cap = 0; HIPCHK(hipMalloc(&nf, nb + 64)); c->m_flat = nf; c->m_flat_cap = nb + 64; }
                    if (nr + 1 > c->m_off_cap) { if (c->m_off) HIPCHK(hipFree(c->m_off)); c->m_off = nullptr; c->m_off_cap = 0; HIPCHK(hipMalloc(&no, (nr + 1) * 8)); c->m_off = no; c->m_off_cap = nr + 1; }
                } else {
                    if (nf && c->m_bytes) HIPCHK(hipMemcpyAsync(nf, c->m_flat, c->m_bytes, hipMemcpyDeviceToDevice, s));
                    if (no && c->m_pairs) HIPCHK(hipMemcpyAsync(no, c->m_off, (2 * c->m_pairs + 1) * 8, hipMemcpyDeviceToDevice, s));
                    if ((nf && c->m_flat) || (no && c->m_off)) HIPCHK(hipStreamSynchronize(s));
                    if (nf) { if (c->m_flat) HIPCHK(hipFree(c->m_flat)); c->m_flat = nf; c->m_flat_cap = want_f; }
                    if (no) { if (c->m_off) HIPCHK(hipFree(c->m_off)); c->m_off = no; c->m_off_cap = want_o; }
                }
            }
            HIPCHK(hipMemcpyAsync(c->m_flat + c->m_bytes, S.d_flat, nb, hipMemcpyDeviceToDevice, s));
            LAUNCH(k_off_rebase, dim3((uint32_t)std::min<uint64_t>((nr + 256) / 256, 1024)), dim3(256), s, c->m_off + 2 * c->m_pairs, S.d_off, nr + 1, c->m_bytes);
            HIPCHK(hipEventRecord(S.aligned, s));  // (the slot's arrays are free again once they have been copied)
            S.has_aligned = true;
            c->m_bytes += nb; c->m_pairs += h.nkept;
            c->m_maxlen = std::max<uint32_t>(c->m_maxlen, (uint32_t)h.maxlen);
        }
    }
    if (c->m_pairs && (flush || c->m_pairs >= min_pairs)) {
        const dbtk_status_t st = launch_batch(c, c->m_flat, c->m_off, ~0ull, c->m_pairs, c->m_maxlen, nullptr, 0, nullptr);
        if (st) return st;
        c->m_bytes = 0; c->m_pairs = 0; c->m_maxlen = 0;
        // (the next blocks are merged into the OTHER lane's buffers, on its stream: this lane's kernels are reading these)
        if (c->two_lanes) switch_lane(c);
    }
    return DBTK_OK;
}

static dbtk_status_t dbtk_ingest_spans_impl(dbtk_ingest_t* g, uint32_t slot, dbtk_ingest_span_t* spans, uint64_t cap) {
    if (!g || slot >= g->nslots || (!spans && cap)) { set_error("null argument"); return DBTK_ERR_ARG; }
    dbtk_ingest::Slot& S = g->slots[slot];
    if (!g->with_spans) { set_error("dbtk_ingest_spans: the ingest was created without spans"); return DBTK_ERR_ARG; }
    if (!S.waited) { set_error("dbtk_ingest_spans: wait for the block first"); return DBTK_ERR_ARG; }
    if (cap < S.hdr.nkept) { set_error("dbtk_ingest_spans: buffer too small"); return DBTK_ERR_OVERFLOW; }
    HIPCHK(hipSetDevice(g->c->device));
    if (S.hdr.nkept) HIPCHK(hipMemcpy(spans, S.d_spans, (size_t)S.hdr.nkept * sizeof(dbtk_ingest_span_t), hipMemcpyDeviceToHost));
    return DBTK_OK;
}

// -a / -ae with the device reader: the lines of the block the context aligned last (dbtk_gz.h)
static dbtk_status_t dbtk_ingest_aln_lines_impl(dbtk_ingest_t* g, uint32_t slot, dbtk_ctx_t* cx, int gz, const void** data, uint64_t* nbytes, uint64_t* nlines,
                                                uint64_t* text_bytes) {
    if (!g || slot >= g->nslots || !nbytes || !data) { set_error("null argument"); return DBTK_ERR_ARG; }
    *nbytes = 0; *data = nullptr;
    if (nlines) *nlines = 0;
    if (text_bytes) *text_bytes = 0;
    dbtk_ingest::Slot& S = g->slots[slot];
    dbtk_ctx* c = nullptr;
    { const dbtk_status_t sc = ingest_ctx(g, cx, &c); if (sc) return sc; }
    if (!g->with_spans || !S.waited) { set_error("dbtk_ingest_aln_lines: needs an ingest created with spans and a block that has been waited for"); return DBTK_ERR_ARG; }
    const uint32_t nk = S.hdr.nkept;
    if (!nk || !c->txt_cap || !c->d_txt || c->last_walk_npairs != nk) return DBTK_OK;  // no text records: no lines
    struct Busy { std::atomic<int>& b; Busy(std::atomic<int>& x) : b(x) { ++b; } ~Busy() { --b; } } busy(g->lines_busy[slot]);
    if (S.pending) { set_error("dbtk_ingest_aln_lines: the slot has been submitted again"); return DBTK_ERR_ARG; }
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    dbtk_ingest::LinesBuf* Bp;
    {
        std::lock_guard<std::mutex> l(g->lines_m);
        Bp = &g->lines[c];
        if (gz && !g->d_crctab) {
            uint32_t tab[288];
            gz_tables(tab);
            HIPCHK(hipMalloc(&g->d_crctab, sizeof(tab)));
            HIPCHK(hipMemcpy(g->d_crctab, tab, sizeof(tab), hipMemcpyHostToDevice));
        }
    }
    dbtk_ingest::LinesBuf& B = *Bp;
    auto grow = [&](auto** p, uint64_t* have, uint64_t want, size_t elem) -> dbtk_status_t {
        if (want <= *have) return DBTK_OK;
        HIPCHK(hipStreamSynchronize(s));
        if (*p) HIPCHK(hipFree(*p));
        *p = nullptr; *have = 0;
        HIPCHK(hipMalloc((void**)p, (want + want / 4) * elem));
        *have = want + want / 4;
        return DBTK_OK;
    };
    HIPCHK(hipStreamSynchronize(s));
    uint32_t cur = 0;
    HIPCHK(hipMemcpy(&cur, c->d_small + 7, 4, hipMemcpyDeviceToHost));
    if (cur > c->txt_cap) { set_error("alignment text arena overflow"); return DBTK_ERR_OVERFLOW; }
    const uint64_t tcap = (uint64_t)(S.end - S.hdr.base) + cur + 24ull * nk + 64;  // every line: bytes of the block + its record + separators and dst
    const uint64_t nmem = tcap / GZ_MEMBER + 1;
    dbtk_status_t st;
    if ((st = grow(&B.d_linelen, &B.linelen_cap, (uint64_t)nk + 2 + ING_SCAN_BLOCKS, 4))) return st;
    if ((st = grow(&B.d_text, &B.text_cap, tcap, 1))) return st;
    if (!B.d_totals) { HIPCHK(hipMalloc(&B.d_totals, 4 * 8)); HIPCHK(hipHostMalloc((void**)&B.h_totals, 4 * 8, hipHostMallocDefault)); }
    HIPCHK(hipMemsetAsync(B.d_totals, 0, 4 * 8, s));
    AlnLineArgs la;
    memset(&la, 0, sizeof(la));
    la.raw = S.d_raw; la.spans = S.d_spans; la.hdr = g->d_hdr + slot; la.txt = c->d_txt; la.txt_idx = c->d_txtidx;
    la.len = B.d_linelen; la.text = B.d_text; la.total = B.d_totals;
    LAUNCH(k_aln_len, dim3(c->num_cu * 2), dim3(256), s, la);
    LAUNCH(k_aln_scan, dim3(ING_SCAN_BLOCKS), dim3(64), s, la, 0);
    LAUNCH(k_aln_scan, dim3(ING_SCAN_BLOCKS), dim3(64), s, la, 1);
    LAUNCH(k_aln_write, dim3(c->num_cu * 32), dim3(64), s, la);
    if (gz) {
        if ((st = grow(&B.d_gzout, &B.gzout_cap, (nmem + 1) * GZ_STRIDE, 1))) return st;
        if ((st = grow(&B.d_packed, &B.packed_cap, (nmem + 1) * GZ_STRIDE, 1))) return st;
        if ((st = grow(&B.d_outlen, &B.outlen_cap, nmem + 4 + ING_SCAN_BLOCKS, 4))) return st;
        HIPCHK(hipMemsetAsync(B.d_gzout, 0, nmem * GZ_STRIDE, s));
        GzArgs ga;
        memset(&ga, 0, sizeof(ga));
        ga.text = B.d_text; ga.total = B.d_totals; ga.out = B.d_gzout; ga.out_len = B.d_outlen; ga.crc_tab = g->d_crctab;
        { static const bool lz = !(getenv("DBTK_GZ_LZ") && atoi(getenv("DBTK_GZ_LZ")) == 0); ga.lz = lz ? 1u : 0u; }
        ga.packed = B.d_packed; ga.packed_total = B.d_totals + 2;
        const uint32_t gm = (uint32_t)std::min<uint64_t>(nmem, (uint64_t)c->num_cu * 16);
        LAUNCH(k_gz_member, dim3(gm), dim3(64), s, ga);
        LAUNCH(k_gz_scan, dim3(ING_SCAN_BLOCKS), dim3(64), s, ga, 0);
        LAUNCH(k_gz_scan, dim3(ING_SCAN_BLOCKS), dim3(64), s, ga, 1);
        LAUNCH(k_gz_pack, dim3(gm), dim3(64), s, ga);
    }
    HIPCHK(hipMemcpyAsync(B.h_totals, B.d_totals, 3 * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (B.h_totals[0] > tcap) { set_error("alignment lines overran their buffer"); return DBTK_ERR_OVERFLOW; }
    const uint64_t n = gz ? B.h_totals[2] : B.h_totals[0];
    if (n > S.h_lines_cap) {  // (pinned: the copy runs at the link's speed and the writer reads it where it lands)
        pinned_give(S.h_lines, S.h_lines_cap);
        S.h_lines = nullptr; S.h_lines_cap = 0;
        uint64_t cap = 0;
        uint8_t* pl = pinned_take(n, &cap);
        if (pl && cap > 4 * n + (8u << 20)) { pinned_give(pl, cap); pl = nullptr; }  // (not a chunk buffer for a few lines)
        if (!pl) { cap = n + n / 4 + 4096; HIPCHK(hipHostMalloc((void**)&pl, cap, hipHostMallocPortable)); }
        S.h_lines = pl; S.h_lines_cap = cap;
    }
    if (n) {
        HIPCHK(hipMemcpyAsync(S.h_lines, gz ? B.d_packed : B.d_text, n, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    *data = S.h_lines; *nbytes = n;
    if (nlines) *nlines = B.h_totals[1];
    if (text_bytes) *text_bytes = B.h_totals[0];
    return DBTK_OK;
}

// ---- the entry points above that parse files or allocate host memory, behind the exception barrier (dbtk_internal.h: guarded)
dbtk_status_t dbtk_device_warmup(int device_id) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device: this library has no CPU execution path"); return DBTK_ERR_NO_DEVICE; }
    if (device_id < 0 || device_id >= n) { set_error("no such device"); return DBTK_ERR_ARG; }
    HIPCHK(hipSetDevice(device_id));
    HIPCHK(hipFree(nullptr));  // (forces the device context)
    return DBTK_OK;
}

dbtk_status_t dbtk_ingest_reserve_host(int device_id, uint64_t chunk_bytes, uint32_t nslots, uint64_t lines_bytes) {
    return dbtk::guarded([&]() -> dbtk_status_t {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device: this library has no CPU execution path"); return DBTK_ERR_NO_DEVICE; }
        if (device_id < 0 || device_id >= n) { set_error("no such device"); return DBTK_ERR_ARG; }
        if (nslots > 64) { set_error("dbtk_ingest_reserve_host: at most 64 slots"); return DBTK_ERR_ARG; }
        HIPCHK(hipSetDevice(device_id));
        const uint64_t raw = chunk_bytes ? ingest_raw_bytes(chunk_bytes) : 0;
        for (uint32_t i = 0; i < nslots; ++i) {  // (chunk and lines buffer of a slot together: the first blocks find both)
            if (raw) { uint8_t* p = nullptr; HIPCHK(hipHostMalloc((void**)&p, raw, hipHostMallocPortable)); pinned_give(p, raw); }
            if (lines_bytes) { uint8_t* p = nullptr; HIPCHK(hipHostMalloc((void**)&p, lines_bytes, hipHostMallocPortable)); pinned_give(p, lines_bytes); }
        }
        return DBTK_OK;
    });
}

dbtk_status_t dbtk_ctx_create(const dbtk_rpgg_t* h, const dbtk_params_t* p, int device_id, dbtk_ctx_t** out) {
    return dbtk::guarded([&] { return dbtk_ctx_create_impl(h, p, device_id, out); });
}
dbtk_status_t dbtk_align_batch(dbtk_ctx_t* c, const uint8_t* seq, const uint64_t* off, const uint8_t* qual,
                               uint64_t npairs, dbtk_pair_rec_t* recs, uint64_t rec_cap, uint64_t* nrec) {
    return dbtk::guarded([&] { return dbtk_align_batch_impl(c, seq, off, qual, npairs, recs, rec_cap, nrec); });
}
dbtk_status_t dbtk_thread_batch(dbtk_ctx_t* c, const uint8_t* seq, const uint64_t* off, const uint32_t* loci, uint64_t nreads,
                                dbtk_thread_rec_t* recs) {
    return dbtk::guarded([&] { return dbtk_thread_batch_impl(c, seq, off, loci, nreads, recs); });
}
dbtk_status_t dbtk_ctx_walk_results(dbtk_ctx_t* c, dbtk_walk_res_t* res, dbtk_thread_rec_t* trecs, uint64_t cap, uint64_t* n) {
    return dbtk::guarded([&] { return dbtk_ctx_walk_results_impl(c, res, trecs, cap, n); });
}
dbtk_status_t dbtk_ctx_aln_records(dbtk_ctx_t* c, void* buf, uint64_t buf_bytes, uint64_t* nrec, uint32_t* stride, uint32_t* cap) {
    return dbtk::guarded([&] { return dbtk_ctx_aln_records_impl(c, buf, buf_bytes, nrec, stride, cap); });
}
dbtk_status_t dbtk_ctx_aln_text(dbtk_ctx_t* c, uint32_t* idx, uint64_t idx_cap, void* arena, uint64_t arena_cap, uint64_t* arena_used) {
    return dbtk::guarded([&] { return dbtk_ctx_aln_text_impl(c, idx, idx_cap, arena, arena_cap, arena_used); });
}
dbtk_status_t dbtk_ctx_write_bubbles(dbtk_ctx_t* c, const char* out_prefix) {
    return dbtk::guarded([&] { return dbtk_ctx_write_bubbles_impl(c, out_prefix); });
}
dbtk_status_t dbtk_ctx_write_bait_hits(dbtk_ctx_t* c, const char* out_prefix) {
    return dbtk::guarded([&] { return dbtk_ctx_write_bait_hits_impl(c, out_prefix); });
}

dbtk_status_t dbtk_ingest_create(dbtk_ctx_t* c, uint32_t fastq, uint32_t min_read_size, uint64_t chunk_bytes, uint32_t nslots, uint32_t with_spans,
                                 dbtk_ingest_t** out) {
    return dbtk::guarded([&] { return dbtk_ingest_create_impl(c, fastq, min_read_size, chunk_bytes, nslots, with_spans, out); });
}
void dbtk_ingest_free(dbtk_ingest_t* g) { ingest_free_impl(g); }
void* dbtk_ingest_chunk_buffer(dbtk_ingest_t* g, uint32_t slot) {
    if (!g || slot >= g->nslots) return nullptr;
    dbtk_ingest::Slot& S = g->slots[slot];
    if (!S.h_raw) {  // pinned on first use, by the caller's reader thread (several slots at once: the pinning of one overlaps the reading into another)
        std::lock_guard<std::mutex> l(g->slot_m[slot]);
        if (!S.h_raw) {
            uint64_t cap = 0;
            uint8_t* p = pinned_take(g->raw_bytes, &cap);
            if (p && cap != g->raw_bytes) { pinned_give(p, cap); p = nullptr; }  // (a slot's buffer goes back to the pool as raw_bytes: exact fits only)
            if (!p && (hipSetDevice(g->c->device) != hipSuccess || hipHostMalloc((void**)&p, g->raw_bytes, hipHostMallocPortable) != hipSuccess)) return nullptr;
            S.h_raw = p;
        }
    }
    return S.h_raw + g->head;
}
const void* dbtk_ingest_block(dbtk_ingest_t* g, uint32_t slot) { return g && slot < g->nslots ? g->slots[slot].h_raw : nullptr; }
dbtk_status_t dbtk_ingest_submit(dbtk_ingest_t* g, uint32_t slot, uint64_t nbytes, int last) {
    return dbtk::guarded([&] { return dbtk_ingest_submit_impl(g, slot, nbytes, last); });
}
dbtk_status_t dbtk_ingest_wait(dbtk_ingest_t* g, uint32_t slot, dbtk_ingest_info_t* info) {
    return dbtk::guarded([&] { return dbtk_ingest_wait_impl(g, slot, info); });
}
dbtk_status_t dbtk_ingest_align(dbtk_ingest_t* g, uint32_t slot, dbtk_ctx_t* cx, int sync, dbtk_pair_rec_t* recs, uint64_t rec_cap, uint64_t* nrec) {
    return dbtk::guarded([&] { return dbtk_ingest_align_impl(g, slot, cx, sync, recs, rec_cap, nrec); });
}
dbtk_status_t dbtk_ingest_align_merged(dbtk_ingest_t* g, uint32_t slot, dbtk_ctx_t* cx, uint64_t min_pairs, int flush) {
    return dbtk::guarded([&] { return dbtk_ingest_align_merged_impl(g, slot, cx, min_pairs, flush); });
}
dbtk_status_t dbtk_ingest_spans(dbtk_ingest_t* g, uint32_t slot, dbtk_ingest_span_t* spans, uint64_t cap) {
    return dbtk::guarded([&] { return dbtk_ingest_spans_impl(g, slot, spans, cap); });
}

dbtk_status_t dbtk_ingest_aln_lines(dbtk_ingest_t* g, uint32_t slot, dbtk_ctx_t* cx, int gz, const void** data, uint64_t* nbytes, uint64_t* nlines,
                                    uint64_t* text_bytes) {
    return dbtk::guarded([&] { return dbtk_ingest_aln_lines_impl(g, slot, cx, gz, data, nbytes, nlines, text_bytes); });
}

}  // extern "C"
