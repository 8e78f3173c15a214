// Diagnostic microbenchmark: what does it cost to ask for the SAME 128-byte line more than once while it is still on its way?
// Random lines of a 16 GB table, 16-byte loads, 5 load instructions in flight per lane, 16 waves per CU:
//   A  every lane its own line per instruction                      (64 lines per instruction, each asked for once)
//   B  every lane ONE line, its five loads five 16-byte parts of it (64 lines per 5 instructions, each asked for 5 times in a row)
//   C  quads share a line (4 parts), five different lines           (16 lines per instruction, asked for once by 4 adjacent lanes)
//   D  groups of 8 lanes share a line (the whole 128 bytes)         (8 lines per instruction)
//   E  as B, but the five parts are asked for by five DIFFERENT lanes of the same instruction, lanes l, l+13, l+26, ... (scattered over the wave)
//   hipcc --offload-arch=gfx950 -O3 -o tools/pend tools/pend.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int V>
__global__ void __launch_bounds__(64) k(const uint4* t, uint64_t mask, int steps, uint64_t* out) {
    const uint32_t lane = threadIdx.x;
    uint64_t h = ((uint64_t)blockIdx.x * 64 + lane + 1) * 0x9E3779B97F4A7C15ull;
    uint64_t acc = 0;
    for (int s = 0; s < steps; ++s) {
        uint4 q[5];
        uint64_t line[5];
        uint32_t part[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            uint64_t hh = (h + (uint64_t)j * 0xD6E8FEB86659FD93ull) * 0x9E3779B97F4A7C15ull;
            if (V == 0) { line[j] = (hh >> 20) & mask; part[j] = (hh >> 10) & 7; }
            if (V == 1) { uint64_t h0 = h * 0x9E3779B97F4A7C15ull; line[j] = (h0 >> 20) & mask; part[j] = (j + lane) & 7; }
            if (V == 2) {  // the quad's line: derived from the quad leader's hash
                const uint32_t lo = __builtin_amdgcn_readfirstlane(0); (void)lo;
                uint64_t hq = __shfl((unsigned long long)hh, lane & ~3u, 64);
                line[j] = (hq >> 20) & mask; part[j] = (lane & 3) + 4 * ((hq >> 8) & 1);
            }
            if (V == 3) { uint64_t hq = __shfl((unsigned long long)hh, lane & ~7u, 64); line[j] = (hq >> 20) & mask; part[j] = lane & 7; }
            if (V == 4) {  // 5 lanes spread over the wave share a line within ONE instruction: lane groups {l, l+13, l+26, l+39, l+52}
                uint64_t hq = __shfl((unsigned long long)hh, lane % 13, 64);
                line[j] = (hq >> 20) & mask; part[j] = (lane / 13 + j) & 7;
            }
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) q[j] = t[line[j] * 8 + part[j]];
#pragma unroll
        for (int j = 0; j < 5; ++j) acc += q[j].x + q[j].w;
        h = (h ^ acc) * 0x9E3779B97F4A7C15ull + s;
    }
    out[(uint64_t)blockIdx.x * 64 + lane] = acc;
}

template <int V>
static void run(const char* name, double lines_per_step_per_wave, const uint4* t, uint64_t nlines, uint64_t* out, int cus, int wpc) {
    const int blocks = cus * wpc, steps = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, t, nlines - 1, steps, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, t, nlines - 1, steps, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double loads = (double)blocks * 64 * steps * 5, lines = (double)blocks * steps * lines_per_step_per_wave;
    printf("  %-58s %2d waves/CU: %8.3f ms  %7.2f G loads/s  %7.2f G lines/s\n", name, wpc, ms, loads / ms / 1e6, lines / ms / 1e6);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const uint64_t nlines = 1ull << 27;  // 16 GB
    uint4* t; if (hipMalloc(&t, nlines * 128) != hipSuccess) { printf("alloc fail\n"); return 1; }
    hipMemset(t, 1, nlines * 128);
    uint64_t* out; hipMalloc(&out, 64ull << 20);
    for (int wpc : {8, 16, 32}) {
        run<0>("A  own line per lane and instruction", 320, t, nlines, out, p.multiProcessorCount, wpc);
        run<1>("B  one line per lane, five parts in five instructions", 64, t, nlines, out, p.multiProcessorCount, wpc);
        run<2>("C  a line per quad and instruction", 80, t, nlines, out, p.multiProcessorCount, wpc);
        run<3>("D  a line per 8 lanes and instruction", 40, t, nlines, out, p.multiProcessorCount, wpc);
        run<4>("E  a line per 5 scattered lanes of one instruction", 65, t, nlines, out, p.multiProcessorCount, wpc);
    }
    return 0;
}
