// Diagnostic microbenchmark: scalar-ALU issue rate per CU (how many SALU instructions per cycle a CU retires with
// 1..8 waves per SIMD), next to the vector-ALU rate.  hipcc --offload-arch=gfx950 -O3 -o salu tools/salu.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int KIND> __global__ void __launch_bounds__(64) k(uint32_t* out, int iters) {
    uint32_t s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    uint32_t v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                asm volatile("s_add_u32 %0, %0, %4\n s_xor_b32 %1, %1, %0\n s_add_u32 %2, %2, %5\n s_xor_b32 %3, %3, %2"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(s1), "s"(s3) : "scc");
        } else if (KIND == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %1, %1, %0\n v_add_u32 %2, %2, %3\n v_xor_b32 %3, %3, %2"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        } else {  // 64-bit scalar shifts / logic, as the mask code uses
            uint64_t a = ((uint64_t)s0 << 32) | s1, b = ((uint64_t)s2 << 32) | s3;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                asm volatile("s_lshl_b64 %0, %0, 1\n s_andn2_b64 %1, %1, %0\n s_or_b64 %0, %0, %1\n s_lshl_b64 %1, %1, 3"
                             : "+s"(a), "+s"(b) : : "scc");
            s0 = (uint32_t)a; s1 = (uint32_t)b;
        }
    }
    if (s0 + s1 + s2 + s3 + v0 + v1 + v2 + v3 == 0x12345) out[0] = 1;
}

int main() {
    uint32_t* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount, iters = 20000;
    const double ghz = p.clockRate / 1e6;
    for (int kind = 0; kind < 3; ++kind)
        for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD
            const int blocks = ncu * 4 * wps;
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) k<0><<<blocks, 64>>>(out, iters); else if (kind == 1) k<1><<<blocks, 64>>>(out, iters); else k<2><<<blocks, 64>>>(out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double instr_per_cu = (double)blocks / ncu * iters * 64.0;
            printf("%s  %d waves/SIMD: %.3f ms, %.2f instr/cycle/CU at %.2f GHz (%.2f per SIMD)\n", kind == 0 ? "SALU32" : kind == 1 ? "VALU  " : "SALU64",
                   wps, ms, instr_per_cu / (ms * 1e-3 * ghz * 1e9), ghz, instr_per_cu / (ms * 1e-3 * ghz * 1e9) / 4);
        }
    return 0;
}
