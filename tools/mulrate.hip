// tools/mulrate.hip — what an integer multiply / a 64-bit shift costs a SIMD on MI355X (the locus-resident
// probe kernel is bound by its vector instructions).  Waves of independent lanes run unrolled loops of ONE kind of instruction on 8
// independent chains; the rate per CU and cycle says how many issue cycles an instruction of that kind takes.
//   hipcc --offload-arch=gfx950 -O2 -o tools/mulrate tools/mulrate.hip && tools/mulrate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITER = 4096, CH = 8;
template <int KIND> __global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
    uint32_t v[CH];
    uint64_t w[CH];
    for (int c = 0; c < CH; ++c) { v[c] = seed + threadIdx.x * 7919u + c * 104729u; w[c] = ((uint64_t)v[c] << 32) | (v[c] * 3u); }
    const uint32_t m = seed | 1u, sh = (seed & 15u) + 1u;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (KIND == 0) v[c] = v[c] * m;                                   // v_mul_lo_u32
            if (KIND == 1) v[c] = __umul24(v[c], m);                          // v_mul_u32_u24
            if (KIND == 2) v[c] = (v[c] ^ m) + sh;                            // two simple 32-bit ops (v_xor + v_add, or one v_xad)
            if (KIND == 3) w[c] = (w[c] >> sh) ^ w[c];                        // v_lshrrev_b64 + 2 v_xor
            if (KIND == 4) v[c] = __builtin_amdgcn_perm(v[c], m, 0x02010003u); // v_perm_b32
            if (KIND == 5) v[c] = __umulhi(v[c], m);                          // v_mul_hi_u32
            if (KIND == 6) v[c] = __builtin_amdgcn_alignbit(v[c], m, sh);     // v_alignbit_b32
            if (KIND == 7) v[c] = __umul24(v[c], m) + sh;                     // v_mad_u32_u24
        }
    }
    uint32_t r = 0;
    for (int c = 0; c < CH; ++c) r ^= v[c] ^ (uint32_t)w[c] ^ (uint32_t)(w[c] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND> int run(const char* name, uint32_t* d, int ncu, double clk_ghz, int ops_per) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    const int blocks = ncu * 8;  // 8 x 4 waves per CU = 8 waves per SIMD
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, a, b));
    const double winst = (double)blocks * 4 * ITER * CH;  // wave-level statements
    const double cyc_per_simd = ms * 1e-3 * clk_ghz * 1e9;
    const double per = cyc_per_simd / (winst / (ncu * 4.0));
    printf("%-34s %8.3f ms  %6.2f SIMD cycles per statement (%d instruction%s)\n", name, ms, per, ops_per, ops_per > 1 ? "s" : "");
    return 0;
}
int main() {
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    const double ghz = p.clockRate / 1e6;
    printf("%s: %d CUs, %.2f GHz\n", p.name, ncu, ghz);
    uint32_t* d = nullptr;
    CHK(hipMalloc(&d, (size_t)ncu * 8 * 256 * 4));
    run<2>("xor + add (32-bit)", d, ncu, ghz, 2);
    run<0>("v_mul_lo_u32", d, ncu, ghz, 1);
    run<1>("v_mul_u32_u24", d, ncu, ghz, 1);
    run<7>("v_mad_u32_u24", d, ncu, ghz, 1);
    run<5>("v_mul_hi_u32", d, ncu, ghz, 1);
    run<3>("v_lshrrev_b64 + 2 xor", d, ncu, ghz, 3);
    run<4>("v_perm_b32", d, ncu, ghz, 1);
    run<6>("v_alignbit_b32", d, ncu, ghz, 1);
    return 0;
}
