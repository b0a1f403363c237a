// Standalone harness for the iconv forward kernel (igemm_p9us2.h) at the step's largest shape
// (8 x [256 skip + 256 up + 1] -> 256 @256^2): random inputs and weight bits (timing only, results unchecked), HIP-event time,
// and -- built with -DP9S_TRACE -- the per-step cycle stamps of S stage 2 (waves 0, 1, 4, 5 of one workgroup).
// Compiles in seconds, so stream variants (-DP9US2_xxx probes) can be A/B'd without rebuilding the library:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -Ijperceiver_amd/csrc [-DP9S_TRACE] tools/ubench/p9us2_bench.hip -o ubench_bin/p9us2_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "igemm_p9us2.h"
struct FwdEpi {
    typedef size_t St;
    float* y; const float* bias; int Cout, OHW, act;
    __device__ __forceinline__ St col(int p) const { int img = p / OHW; return (size_t)img * Cout * OHW + (p - img * OHW); }
    __device__ __forceinline__ void put(St base, int m, float v) const { if (bias) v += bias[m]; y[base + (size_t)m * OHW] = jp_act(v, act); }
};
__global__ void fill(unsigned* p, size_t n, unsigned seed, int as_float) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        if (as_float) { float f = ((h & 0xffffff) / 16777216.0f - 0.5f) * 4.f; p[i] = __float_as_uint(f); }
        else p[i] = (h & 0x807f807fu) | 0x3c003c00u;     // two bf16 of magnitude ~2^-7 .. 2^-6, random sign / mantissa
    }
}
__global__ void checksum(const float* p, size_t n, unsigned long long* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    unsigned long long s = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) s += (unsigned long long)__float_as_uint(p[i]) * (i % 1000003 + 1);
    atomicAdd(out, s);
}
int main(int argc, char** argv) {
    const int N = 8, H = 256, W = 256, C0 = 256, C1 = 256, C2 = 1, M = 256, MT = M / 128;
    const int reps = argc > 1 ? atoi(argv[1]) : 6;
    const size_t n0 = (size_t)N * C0 * H * W, n1 = (size_t)N * C1 * (H / 2) * (W / 2), n2 = (size_t)N * C2 * H * W, ny = (size_t)N * M * H * W;
    const size_t SB = JP_NS * 2 * 128 * 16;
    const size_t wbytes = (size_t)MT * (C0 / 16) * 9 * SB + 4 * (size_t)MT * (C1 / 16) * 4 * SB + (size_t)MT * 9 * SB + SB;
    float *x0, *x1, *x2, *y, *bias; unsigned* wp;
    hipMalloc(&x0, n0 * 4); hipMalloc(&x1, n1 * 4); hipMalloc(&x2, n2 * 4); hipMalloc(&y, ny * 4); hipMalloc(&wp, wbytes); hipMalloc(&bias, M * 4);
    fill<<<4096, 256>>>((unsigned*)x0, n0, 1u, 1); fill<<<4096, 256>>>((unsigned*)x1, n1, 2u, 1); fill<<<4096, 256>>>((unsigned*)x2, n2, 3u, 1);
    fill<<<4096, 256>>>(wp, wbytes / 4, 4u, 0); fill<<<16, 256>>>((unsigned*)bias, M, 5u, 1);
    FwdEpi e{y, bias, M, H * W, 2};
    float* am; hipMalloc(&am, JP_AMAX_SLOT * 4); hipMemset(am, 0, JP_AMAX_SLOT * 4);      // largest |x| of the sources (JP_NS == 2): the fill is in [-2, 2)
    { const float two = 2.f; hipMemcpy(am, &two, 4, hipMemcpyHostToDevice); }
    { const float hdr[4] = {1.f, 1.f, 0.f, 0.f}; hipMemcpy(wp, hdr, 16, hipMemcpyHostToDevice); }     // pack header {scale, 1 / scale} (JP_NS == 2)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flops = (JP_NS == 2 ? 3.0 : 6.0) * 2.0 * M * (double)N * H * W * (9.0 * C0 + 4.0 * C1 + 9.0 * 16);
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((jp_igemm_p9us2_kernel<FwdEpi>), dim3(N * (H / 4) * (W / 64), MT, 1), dim3(512), 0, 0, wp, x0, x1, x2, e, M, C0, C1, C2, H, W, am);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2) printf("%s %.3f ms  %.0f TF executed\n", argv[0], ms, flops / ms / 1e9);
    }
    if (hipGetLastError() != hipSuccess) { printf("launch error\n"); return 1; }
    {
        unsigned long long* cs; hipMalloc(&cs, 8); hipMemset(cs, 0, 8);
        checksum<<<2048, 256>>>(y, ny, cs);
        unsigned long long h; hipMemcpy(&h, cs, 8, hipMemcpyDeviceToHost);
        printf("checksum %llx\n", h);
    }
#ifdef P9S_TRACE
    unsigned long long t[64];
    hipMemcpyFromSymbol(t, HIP_SYMBOL(jp_p9s_trace), sizeof(t));
    unsigned long long t0 = ~0ull; for (int i = 0; i < 64; ++i) if (t[i] && t[i] < t0) t0 = t[i];
    const int ws[4] = {0, 1, 4, 5};
    for (int s = 0; s < 4; ++s) {
        printf("  w%d |", ws[s]);
        for (int i = 0; i < 16; ++i) printf(" %6lld", t[s * 16 + i] ? (long long)(t[s * 16 + i] - t0) : -1);
        printf("\n       ");
        for (int i = 0; i < 16; ++i) printf(" %6lld", (i && t[s * 16 + i]) ? (long long)(t[s * 16 + i] - t[s * 16 + i - 1]) : 0);
        printf("\n");
    }
#endif
    return 0;
}
