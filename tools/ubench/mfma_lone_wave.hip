// Microbenchmark (GPU-box aid): the MFMA issue rate of ONE wave on its SIMD, `v_mfma_f32_32x32x16_bf16`, by number of independent
// accumulators and by accumulator file (arch VGPRs, what hipcc picks for kernels compiled for 2 waves per SIMD, vs AccVGPRs through
// inline asm), measured in shader cycles with s_memtime around 256 MFMAs; 1 and 2 waves per SIMD, one workgroup per CU.
// Question behind it (profiles/r05_p9us2_probe_variants.log): a lone wave of the split-bf16 conv kernels never issues faster than one
// MFMA per ~48 cycles, while two waves together reach one per ~33.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_lone_wave.hip -o ubench_bin/mfma_lone_wave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool AGPR, int NOPS>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* cyc, unsigned seed) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    unsigned h = (t * 2654435761u) ^ seed;
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return (h & 0x007f007fu) | 0x3f003f00u; };
    u32x4 a[2], b[2];
    for (int s = 0; s < 2; ++s) for (int q = 0; q < 4; ++q) { a[s][q] = rnd(); b[s][q] = rnd(); }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
#pragma unroll
        for (int rep = 0; rep < 32 / NACC; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if constexpr (AGPR) {
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[rep & 1]), "v"(b[i & 1]));
                } else {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[rep & 1]), __builtin_bit_cast(bf16x8, b[i & 1]), acc[i], 0, 0, 0);
                }
                if constexpr (NOPS > 0) asm volatile("s_nop %0" :: "n"(NOPS - 1));
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if constexpr (AGPR) asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[t] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 7) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int NACC, bool AGPR, int NOPS = 0>
void run(const char* label, float* out, unsigned long long* cyc, int threads) {
    unsigned long long h[8] = {0};
    hipMemset(cyc, 0, sizeof(h));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<NACC, AGPR, NOPS>), dim3(256), dim3(threads), 0, 0, out, cyc, 777u + rep);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s %d wave(s)/SIMD:", label, threads / 256);
    for (int w = 0; w < threads / 64; ++w) printf(" %5.1f", h[w] / 256.0);
    printf("  cycles per MFMA (per wave; the SIMD issues 1 / that x waves)\n");
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    for (int threads : {256, 512}) {
        run<1, false>("VGPR acc, 1 accumulator (dependent chain)", out, cyc, threads);
        run<2, false>("VGPR acc, 2 accumulators", out, cyc, threads);
        run<4, false>("VGPR acc, 4 accumulators", out, cyc, threads);
        run<8, false>("VGPR acc, 8 accumulators", out, cyc, threads);
        run<1, true>("AGPR acc, 1 accumulator (dependent chain)", out, cyc, threads);
        run<2, true>("AGPR acc, 2 accumulators", out, cyc, threads);
        run<4, true>("AGPR acc, 4 accumulators", out, cyc, threads);
        run<8, true>("AGPR acc, 8 accumulators", out, cyc, threads);
        run<4, false, 1>("VGPR acc, 4 accumulators + s_nop 0 each", out, cyc, threads);
        run<4, false, 4>("VGPR acc, 4 accumulators + s_nop 3 each", out, cyc, threads);
    }
    return 0;
}
