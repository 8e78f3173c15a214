// Diagnostic microbenchmark: latency and throughput of dependent random 16-B
// loads over tables of different sizes (TLB / HBM behaviour of index probes).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
struct Slot { uint64_t a, b; };
__global__ void chase(const Slot* t, uint64_t mask, int steps, int per_lane, uint64_t* out, uint64_t* cyc) {
    uint64_t h[8];
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int j = 0; j < 8; ++j) h[j] = (gid * 8 + j) * 0x9E3779B97F4A7C15ull;
    uint64_t acc = 0;
    const uint64_t t0 = clock64();
    for (int s = 0; s < steps; ++s) {
        Slot q[8];
        for (int j = 0; j < 8; ++j) if (j < per_lane) q[j] = t[(h[j] >> 20) & mask];
        for (int j = 0; j < 8; ++j) if (j < per_lane) { acc += q[j].b; h[j] = (h[j] ^ q[j].a) * 0x9E3779B97F4A7C15ull + s; }
    }
    const uint64_t t1 = clock64();
    out[gid] = acc;
    if (threadIdx.x == 0) atomicAdd((unsigned long long*)cyc, (unsigned long long)(t1 - t0));
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d clock %d kHz\n", p.multiProcessorCount, p.clockRate);
    uint64_t *out, *cyc; hipMalloc(&out, 8ull << 20); hipMalloc(&cyc, 8);
    for (int lg = 22; lg <= 30; lg += 2) {
        const uint64_t n = 1ull << lg;  // slots of 16 B
        Slot* t; if (hipMalloc(&t, n * 16) != hipSuccess) { printf("alloc fail\n"); break; }
        hipMemset(t, 1, n * 16);
        for (int wpc : {1, 8, 16, 32}) for (int pl : {1, 8}) {
            const int blocks = p.multiProcessorCount * wpc, steps = 64;
            hipMemset(cyc, 0, 8);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, t, n - 1, steps, pl, out, cyc);  // warm
            hipMemset(cyc, 0, 8);
            hipEventRecord(e0);
            hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, t, n - 1, steps, pl, out, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            uint64_t c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double loads = (double)blocks * 64 * steps * pl;
            printf("table %6.0f MB  waves/CU %2d  loads/lane in flight %d : %8.0f cycles/step  %7.2f Gloads/s  %7.1f GB/s(64B)\n",
                   n * 16 / 1048576.0, wpc, pl, (double)c / blocks / steps, loads / ms / 1e6, loads * 64 / ms / 1e6);
        }
        hipFree(t);
    }
    return 0;
}
