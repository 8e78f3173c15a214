// Diagnostic microbenchmark: does the cache policy of a load change what a random 8-byte probe costs the fabric?
// Random 8-B loads (the encode kernel's filter words) from a 128 MB and a 16 GB table, 8 in flight per lane, with the
// gfx950 cache-policy bits: none | sc0 | sc1 | sc0 sc1 | nt | sc0 sc1 nt.   hipcc --offload-arch=gfx950 -O3 -o tools/probe_gran tools/probe_gran.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define LOADV(POL)                                                                                          \
    __device__ inline void load8_##POL(const uint64_t* t, const uint64_t idx[8], uint64_t v[8]) {           \
        for (int j = 0; j < 8; ++j) asm volatile("global_load_dwordx2 %0, %1, off " POLSTR_##POL : "=v"(v[j]) : "v"(t + idx[j]) : "memory"); \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                    \
    }
#define POLSTR_none ""
#define POLSTR_sc0 "sc0"
#define POLSTR_sc1 "sc1"
#define POLSTR_sc01 "sc0 sc1"
#define POLSTR_nt "nt"
#define POLSTR_all "sc0 sc1 nt"
LOADV(none) LOADV(sc0) LOADV(sc1) LOADV(sc01) LOADV(nt) LOADV(all)

template <int POL>
__global__ void probe(const uint64_t* t, uint64_t mask, int steps, uint64_t* out) {
    uint64_t h[8], v[8], idx[8];
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int j = 0; j < 8; ++j) h[j] = (gid * 8 + j + 1) * 0x9E3779B97F4A7C15ull;
    uint64_t acc = 0;
    for (int s = 0; s < steps; ++s) {
        for (int j = 0; j < 8; ++j) idx[j] = (h[j] >> 20) & mask;
        if (POL == 0) load8_none(t, idx, v);
        else if (POL == 1) load8_sc0(t, idx, v);
        else if (POL == 2) load8_sc1(t, idx, v);
        else if (POL == 3) load8_sc01(t, idx, v);
        else if (POL == 4) load8_nt(t, idx, v);
        else load8_all(t, idx, v);
        for (int j = 0; j < 8; ++j) { acc += v[j]; h[j] = (h[j] ^ v[j]) * 0x9E3779B97F4A7C15ull + s; }
    }
    out[gid] = acc;
}
template <int POL>
static void run(const char* name, const uint64_t* t, uint64_t n, uint64_t* out, int cus) {
    const int blocks = cus * 16, steps = 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<POL>, dim3(blocks), dim3(64), 0, 0, t, n - 1, steps, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<POL>, dim3(blocks), dim3(64), 0, 0, t, n - 1, steps, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double loads = (double)blocks * 64 * steps * 8;
    printf("  %-12s %7.2f G loads/s\n", name, loads / ms / 1e6);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    uint64_t* out; hipMalloc(&out, 64ull << 20);
    for (int lg : {24, 31}) {
        const uint64_t n = 1ull << lg;  // 8-byte words
        uint64_t* t; if (hipMalloc(&t, n * 8) != hipSuccess) { printf("alloc fail\n"); break; }
        hipMemset(t, 1, n * 8);
        printf("table %.0f MB, random 8-byte loads, 16 waves per CU, 8 in flight per lane\n", n * 8 / 1048576.0);
        run<0>("(default)", t, n, out, p.multiProcessorCount);
        run<1>("sc0", t, n, out, p.multiProcessorCount);
        run<2>("sc1", t, n, out, p.multiProcessorCount);
        run<3>("sc0 sc1", t, n, out, p.multiProcessorCount);
        run<4>("nt", t, n, out, p.multiProcessorCount);
        run<5>("sc0 sc1 nt", t, n, out, p.multiProcessorCount);
        hipFree(t);
    }
    return 0;
}
