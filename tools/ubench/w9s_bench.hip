// Standalone harness for the 3x3 weight-gradient kernel jp_wgrad_w9s_kernel<4, true, 1, 1> (igemm_w9s.h) at the step's by-time
// dominant shape (8 x 256 -> 256 reflect @256^2; grid and split-K plan as conv.hip's w9_plan gives them): random inputs, HIP-event
// time and a checksum of the partial sums (the round-4 stream, removed since, gave the same checksum: profiles/r05_w9s_ab.log).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -Ijperceiver_amd/csrc tools/ubench/w9s_bench.hip -o ubench_bin/w9s_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "igemm_w9s.h"
__global__ void fill(float* p, size_t n, unsigned seed) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((h & 0xffffff) / 16777216.0f - 0.5f) * 4.f;
    }
}
__global__ void checksum(const float* p, size_t n, unsigned long long* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    unsigned long long s = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) s += (unsigned long long)__float_as_uint(p[i]) * (i % 1000003 + 1);
    atomicAdd(out, s);
}
int main(int argc, char** argv) {
    const int N = 8, H = 256, W = 256, C = 256, Cout = 256;
    const int reps = argc > 1 ? atoi(argv[1]) : 6;
    const size_t n = (size_t)N * C * H * W;
    const int ntiles = N * (H / 4) * (W / 32), splits = 32, tps = (ntiles + splits - 1) / splits;
    const size_t nws = (size_t)splits * Cout * 9 * C;
    float *dy, *x, *ws; unsigned long long* cs;
    hipMalloc(&dy, n * 4); hipMalloc(&x, n * 4); hipMalloc(&ws, nws * 4); hipMalloc(&cs, 8);
    fill<<<4096, 256>>>(dy, n, 1u); fill<<<4096, 256>>>(x, n, 2u);
    hipMemset(ws, 0, nws * 4); hipMemset(cs, 0, 8);
    float* am; hipMalloc(&am, 8);                       // largest magnitudes of dY / X for the fp16 split build (-DJP_NS=2): the fill is in [-2, 2)
    { const float h[2] = {2.f, 2.f}; hipMemcpy(am, h, 8, hipMemcpyHostToDevice); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flops = (JP_NS == 2 ? 3.0 : 6.0) * 2.0 * Cout * 9.0 * C * (double)N * H * W;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((jp_wgrad_w9s_kernel<4, true, 1, 1>), dim3(C / 32, Cout / 256, splits), dim3(512), 0, 0, dy, x, ws, Cout, C, C, H, W,
                           ntiles, tps, (int)(n * 4), (int)(n * 4), am, am + 1);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2) printf("%s NS=%d %.3f ms  %.0f TF executed  %.0f TF fp32-equivalent\n", argv[0], JP_NS, ms, flops / ms / 1e9, flops / (JP_NS == 2 ? 3 : 6) / ms / 1e9);
    }
    checksum<<<1024, 256>>>(ws, nws, cs);
    unsigned long long h; hipMemcpy(&h, cs, 8, hipMemcpyDeviceToHost);
    printf("checksum %llx  %s\n", h, hipGetLastError() == hipSuccess ? "ok" : "LAUNCH ERROR");
    return 0;
}
