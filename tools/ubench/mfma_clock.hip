// Microbenchmark (GPU-box aid): what does s_memtime count while the matrix pipes are saturated for milliseconds?  Every SIMD of the
// chip runs `iters` x 256 back-to-back v_mfma_f32_32x32x16_bf16 (4 accumulators, 1 or 2 waves per SIMD, random or zero operands);
// one wave stamps s_memtime (readcyclecounter) and s_memrealtime (constant 100 MHz) around the run:
//   ticks per MFMA per SIMD  -- 32 if s_memtime counts shader cycles and the pipe is never idle;
//   s_memtime frequency       -- delta(memtime) / delta(realtime) x 100 MHz.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_clock.hip -o ubench_bin/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* st, int iters, unsigned seed, int zero) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    unsigned h = (t * 2654435761u) ^ seed;
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return zero ? 0u : ((h & 0x807f807fu) | 0x3f003f00u); };
    u32x4 a[2], b[2];
    for (int s = 0; s < 2; ++s) for (int q = 0; q < 4; ++q) { a[s][q] = rnd(); b[s][q] = rnd(); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long c_mid0 = 0, c_mid1 = 0;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (it == iters - 2) c_mid0 = __builtin_readcyclecounter();
#pragma unroll
        for (int rep = 0; rep < 64; ++rep)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[rep & 1]), __builtin_bit_cast(bf16x8, b[i & 1]), acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        if (it == iters - 2) c_mid1 = __builtin_readcyclecounter();
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[t] = s;
    if (threadIdx.x == 0 && blockIdx.x == 7) { st[0] = c1 - c0; st[1] = r1 - r0; st[2] = c_mid1 - c_mid0; }
}
int main() {
    float* out; unsigned long long* st;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int zero : {0, 1})
        for (int threads : {256, 512})
            for (int iters : {8, 512, 8192}) {
                unsigned long long h[3];
                float ms = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, st, iters, 777u + rep, zero);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
                const double mf = (double)iters * 256;            // MFMAs per wave
                const double waves = threads / 256;
                const double tf = 256.0 * (threads / 64) * mf * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
                printf("%s operands, %d wave/SIMD, %6.0f MFMAs per wave, %7.3f ms: wave 0 %6.1f ticks/MFMA (late loop trip %6.1f), s_memtime runs at %6.1f MHz, %6.0f TF\n",
                       zero ? "zero  " : "random", threads / 256, mf, ms, h[0] / mf, h[2] / 256.0, h[0] / (h[1] / 100.0), tf);
                (void)waves;
            }
    return 0;
}
