// Microbenchmark (GPU-box aid): does the ISSUE ORDER of v_mfma_f32_32x32x16_bf16 over a wave's accumulators matter to the
// matrix pipe?  72 MFMAs per loop trip in every variant, program order pinned with sched_barrier (the scheduler otherwise
// round-robins independent MFMAs, which made a first version of this benchmark measure its own loop overhead), random bf16
// operands, 1 and 2 waves per SIMD.  Result (profiles/r04_mfma_bf16_chain.log): 2.15-2.25 PF = 86-90 % of 2.5 PF for a fully
// dependent chain, round robin over 2 / 4 / 9 accumulators and six-in-a-row alike -- the order is not a lever; the split
// kernels' product-major order (igemm_p9s.h) and W9S's tap-major order (igemm_w9s.h) both issue at the pipe's rate.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_bf16_chain.hip -o ubench_bin/mfma_bf16_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int RUN>   // RUN consecutive MFMAs on one accumulator before moving to the next of NACC; 72 MFMAs per loop trip
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned seed) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    unsigned h = (t * 2654435761u) ^ seed;
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return (h & 0x007f007fu) | 0x3f003f00u; };   // two bf16 in [0.5, 1)
    u32x4 ar[3], br[3];
    for (int s = 0; s < 3; ++s) for (int q = 0; q < 4; ++q) { ar[s][q] = rnd(); br[s][q] = rnd(); }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    constexpr int REPS = 72 / (NACC * RUN);
    static_assert(REPS * NACC * RUN == 72, "72 MFMAs per trip");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < REPS; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
#pragma unroll
                for (int u = 0; u < RUN; ++u) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ar[u % 3]), __builtin_bit_cast(bf16x8, br[(u + i) % 3]), acc[i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);      // program order = issue order (the scheduler would round-robin them)
                }
            }
        const u32x4 tmp = ar[0]; ar[0] = ar[1]; ar[1] = ar[2]; ar[2] = br[0]; br[0] = br[1]; br[1] = br[2]; br[2] = tmp;
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[t] = s;
}

template <int NACC, int RUN>
void run(const char* label, float* out, int threads, int total_mfma) {
    const int blocks = 256 * 2;
    const int iters = total_mfma / 72;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, RUN>), dim3(blocks), dim3(threads), 0, 0, out, iters, 777u + rep);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = (double)blocks * (threads / 64) * iters * 72 * 2.0 * 32 * 32 * 16;
    printf("%-52s %d waves/SIMD  %8.2f ms  %7.1f TFLOP/s (%.1f %% of 2500)\n", label, threads / 256, best, flops / best / 1e9, flops / best / 1e9 / 25.0);
}

int main() {
    float* out; hipMalloc(&out, (size_t)512 * 512 * 4);
    const int N = 72 * 3000;
    for (int threads : {256, 512}) {
        run<1, 72>("1 accumulator (fully dependent chain)", out, threads, N);
        run<2, 1>("2 accumulators, round robin", out, threads, N);
        run<4, 1>("4 accumulators, round robin (P9S order)", out, threads, N);
        run<9, 1>("9 accumulators, round robin", out, threads, N);
        run<2, 6>("2 accumulators, 6 in a row on each", out, threads, N);
        run<4, 6>("4 accumulators, 6 in a row on each", out, threads, N);
        run<4, 18>("4 accumulators, 18 in a row on each", out, threads, N);
    }
    return 0;
}
