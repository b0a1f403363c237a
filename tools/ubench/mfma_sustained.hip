// Microbenchmark (GPU-box aid): the fp32-MFMA rate MI355X SUSTAINS under its power budget, as a function of operand
// data.  MI355X_MICROARCH.md "DVFS give-back": the chip clocks to its power budget; zero-filled operands ran +19 % in
// TF/s over random ones at identical instruction counts.  A bare v_mfma_f32_32x32x2_f32 stream (4 independent
// accumulators per wave, operands in registers, nothing else) is the ceiling ANY fp32-MFMA kernel can reach on real
// data; this prints it for zero / random operands over ~1 s launches, so the convolution kernels' rates can be read
// against the sustained ceiling rather than against the 2.4 GHz spec peak (157.3 TF).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_sustained.hip -o ubench_bin/mfma_sustained
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(float* out, int iters, float scale, unsigned seed) {
    const int t = threadIdx.x + blockIdx.x * 256;
    unsigned h = (t * 2654435761u) ^ seed;
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return scale * ((h >> 8) * (1.0f / 8388608.0f) - 1.0f); };
    float a0 = rnd(), a1 = rnd(), b0 = rnd(), b1 = rnd();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
        for (int u = 0; u < 8; ++u) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        // rotate the operands so that the data changes every few MFMAs like a real K loop (cheap: 4 VALU / 32 MFMA)
        const float tmp = a0; a0 = a1; a1 = b0; b0 = b1; b1 = tmp;
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[t] = s;
}

int main(int argc, char** argv) {
    const int blocks = 256 * 3;
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = argc > 1 ? atoi(argv[1]) : 60000;
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            const float scale = mode == 0 ? 0.f : (mode == 1 ? 1e-3f : 1.0f);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, scale, 12345u + rep);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * 4 * iters * 32.0 * 2.0 * 32 * 32 * 2;   // waves * mfma * 2*32*32*2
            printf("rep %d operands %-22s %8.1f ms  %7.2f TFLOP/s  (%.1f %% of 157.3; implied clock %.2f GHz)\n", rep,
                   mode == 0 ? "all zero" : (mode == 1 ? "random, |x| < 1e-3" : "random, |x| < 1"), ms, flops / ms / 1e9,
                   flops / ms / 1e9 / 157.3 * 100, flops / ms / 1e9 / 157.3 * 2.4);
        }
    return 0;
}
