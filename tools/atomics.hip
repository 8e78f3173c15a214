// Diagnostic microbenchmark: throughput of fire-and-forget random atomic adds on a counts-like array.
// hipcc --offload-arch=gfx950 -O3 -o atomics tools/atomics.hip && ./atomics
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

// mode 0: u64 atomics, random; 1: u32 atomics, random; 2: u64, 64 lanes of a wave hit 64 consecutive counters (one line group);
// 3: u64 random but each wave's 64 addresses fall in one 32 KB window
template <int MODE> __global__ void k(uint64_t* c64, uint32_t* c32, uint64_t n, uint64_t per_thread) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint64_t j = 0; j < per_thread; ++j) {
        const uint64_t wave = tid >> 6, lane = tid & 63;
        uint64_t i;
        if (MODE == 2) i = (mix(wave * 1315423911ull + j) % (n - 64)) + lane;
        else if (MODE == 3) i = ((mix(wave * 1315423911ull + j) % (n - 4096)) & ~4095ull) + (mix(tid * 7 + j) & 4095);
        else i = mix(tid * 0x9E3779B97F4A7C15ull + j) % n;
        if (MODE == 1) atomicAdd(&c32[i], 1u); else atomicAdd((unsigned long long*)&c64[i], 1ull);
    }
}

int main() {
    const uint64_t n = 31u << 20;  // counters (the release-scale graph has 31 M TR k-mers)
    uint64_t* c64; uint32_t* c32;
    hipMalloc(&c64, n * 8); hipMalloc(&c32, n * 4);
    hipMemset(c64, 0, n * 8); hipMemset(c32, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const uint64_t threads = 256ull * 2048, per = 40;  // 21 M atomics
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            switch (mode) {
                case 0: k<0><<<threads / 256, 256>>>(c64, c32, n, per); break;
                case 1: k<1><<<threads / 256, 256>>>(c64, c32, n, per); break;
                case 2: k<2><<<threads / 256, 256>>>(c64, c32, n, per); break;
                default: k<3><<<threads / 256, 256>>>(c64, c32, n, per); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("mode %d: %.3f ms for %.1f M atomics = %.1f G atomics/s\n", mode, ms, threads * per / 1e6, threads * per / ms / 1e6);
        }
    }
    return 0;
}
