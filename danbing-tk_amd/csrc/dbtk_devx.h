// dbtk_devx.h — the execution context the kernel bodies (dbtk_kernels.h, dbtk_walk.h) are instantiated with on the GPU:
// threadIdx / blockIdx, wave64 ballots, DPP scans and quad permutes, wavefront-scope fences, global + LDS atomics.
// (tests/emu instantiates the same bodies with a coroutine-lane context instead.)
#ifndef DBTK_DEVX_H_
#define DBTK_DEVX_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

// ------------------------------------------------------------------ DevX ---
struct DevX {
    void* sm;
    __device__ int tid() const { return (int)threadIdx.x; }
    __device__ int nthreads() const { return (int)blockDim.x; }
    __device__ uint32_t bid() const { return blockIdx.x; }
    __device__ uint32_t nblocks() const { return gridDim.x; }
    __device__ int lane() const { return (int)(threadIdx.x & 63); }
    // Every kernel here runs ONE wavefront per block, so "sync" only has to order this wave's own
    // LDS traffic: a wavefront-scope fence (compiler ordering; the LDS executes a wave's operations
    // in order) instead of __syncthreads(), whose s_waitcnt vmcnt(0) would drain the prefetched loads
    // and the fire-and-forget count atomics at every phase boundary.
    __device__ void sync() const {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // workgroup barrier, for the kernels that do run several waves per block (dbtk_locus.h: the waves of a block share an LDS image)
    __device__ void bsync() const { __syncthreads(); }
    __device__ uint64_t ballot(bool p) const { return __ballot(p); }
    // wave-uniform results are returned through readfirstlane/readlane so that the compiler keeps
    // them (and every loop bound, length and flag derived from them) in SGPRs with scalar branches
    __device__ uint32_t uni(uint32_t v) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
    // Wave-wide inclusive scans in six DPP steps (row_shr 1, 2, 4, 8 inside the rows of 16 lanes, then row_bcast 15 and 31
    // across the rows) instead of six ds_bpermute round trips through the LDS crossbar.  OP(cur, earlier) must be associative;
    // lanes without a source keep the identity 0.
    template <class OP> __device__ static uint32_t dpp_scan(uint32_t v, OP op) {
        v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));  // row_shr:1
        v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));  // row_shr:2
        v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));  // row_shr:4
        v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));  // row_shr:8
        v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1 and 3
        v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));  // row_bcast:31 into rows 2 and 3
        return v;
    }
    // The same scan inside each HALF of the wave (lanes 0-31, 32-63): without the last step lane 31 / lane 63 hold their half's total.
    template <class OP> __device__ static uint32_t dpp_scan_half(uint32_t v, OP op) {
        v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
        v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
        v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
        v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
        v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
        return v;
    }
    struct OpAdd { __device__ uint32_t operator()(uint32_t c, uint32_t e) const { return c + e; } };
    struct OpMax { __device__ uint32_t operator()(uint32_t c, uint32_t e) const { return c > e ? c : e; } };
    struct OpLastNz { __device__ uint32_t operator()(uint32_t c, uint32_t e) const { return c ? c : e; } };
    __device__ uint32_t wave_sum(uint32_t v) const { return (uint32_t)__builtin_amdgcn_readlane((int)dpp_scan(v, OpAdd{}), 63); }
    __device__ uint32_t wave_min(uint32_t v) const {  // as a max-scan of the complement
        return ~(uint32_t)__builtin_amdgcn_readlane((int)dpp_scan(~v, OpMax{}), 63);
    }
    // sum / maximum over the lane's half of the wave (the probe kernel works on one mate per half)
    __device__ uint32_t half_sum(uint32_t v) const {
        const uint32_t s = dpp_scan_half(v, OpAdd{});
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)s, 31), hi = (uint32_t)__builtin_amdgcn_readlane((int)s, 63);
        return (threadIdx.x & 32) ? hi : lo;
    }
    __device__ uint32_t half_max(uint32_t v) const {
        const uint32_t s = dpp_scan_half(v, OpMax{});
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)s, 31), hi = (uint32_t)__builtin_amdgcn_readlane((int)s, 63);
        return (threadIdx.x & 32) ? hi : lo;
    }
    __device__ uint32_t wave_scan_max(uint32_t v) const { return dpp_scan(v, OpMax{}); }  // inclusive
    __device__ uint32_t half_scan_max(uint32_t v) const { return dpp_scan_half(v, OpMax{}); }  // inclusive, inside each half of the wave
    __device__ uint32_t wave_excl_scan(uint32_t v) const { return dpp_scan(v, OpAdd{}) - v; }
    template <int E> __device__ void shfl_xor64(const uint64_t (&in)[E], uint64_t (&out)[E], int mask) const {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)in[j], mask, 64);
            const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(in[j] >> 32), mask, 64);
            out[j] = ((uint64_t)hi << 32) | lo;
        }
    }
    __device__ uint32_t wave_scan_lastnz(uint32_t v) const { return dpp_scan(v, OpLastNz{}); }  // inclusive scan, op(a, b) = b ? b : a
    __device__ uint32_t shfl_up1(uint32_t v) const {  // value of lane - 1 (lane 0 keeps its own): wave_shr:1
        return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false);
    }
    __device__ uint32_t shfl_down1(uint32_t v) const {  // value of lane + 1 (lane 63 keeps its own): wave_shl:1
        return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130, 0xf, 0xf, false);
    }
    // lane 4q+j reads lane 4q+Pj (DPP quad_perm: no LDS traffic).  Every lane of the wave must be active at the call.
    template <int P0, int P1, int P2, int P3> __device__ uint32_t quad_perm(uint32_t v) const {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xF, 0xF, true);
    }
    __device__ uint32_t bcast(uint32_t v, int src) const {  // src must be wave-uniform
        return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(src));
    }
    __device__ void atomic_add(uint64_t* p, uint64_t v) const { atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v); }
    __device__ uint32_t atomic_add(uint32_t* p, uint32_t v) const { return atomicAdd(p, v); }
    __device__ uint64_t atomic_cas(uint64_t* p, uint64_t e, uint64_t d) const {
        return atomicCAS(reinterpret_cast<unsigned long long*>(p), (unsigned long long)e, (unsigned long long)d);
    }
    __device__ uint32_t atomic_cas32(uint32_t* p, uint32_t e, uint32_t d) const { return atomicCAS(p, e, d); }
    __device__ void atomic_max(uint64_t* p, uint64_t v) const { atomicMax(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v); }
    __device__ void atomic_or(uint64_t* p, uint64_t v) const { atomicOr(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v); }
    __device__ uint32_t atomic_or32(uint32_t* p, uint32_t v) const { return atomicOr(p, v); }
    __device__ uint64_t clock() const { return (uint64_t)clock64(); }
    __device__ uint32_t lds_add(uint32_t* p, uint32_t v) const { return atomicAdd(p, v); }
    __device__ void lds_or(uint32_t* p, uint32_t v) const { atomicOr(p, v); }
    __device__ void lds_max(uint32_t* p, uint32_t v) const { atomicMax(p, v); }
    __device__ void lds_min(uint32_t* p, uint32_t v) const { atomicMin(p, v); }
    template <class T> __device__ T* smem() const { return reinterpret_cast<T*>(sm); }
};


#endif
