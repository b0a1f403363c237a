// Standalone harness for the P9S patch kernels (igemm_p9s.h): the 3x3 wide tile kernel and the 1x1 kernel at 8 x 256 -> 256 @256^2,
// random inputs and random weight-fragment bits (timing only), HIP-event times.  Build with -DJP_NS=3 (three bf16 splits, six products) or
// -DJP_NS=2 (two fp16 splits, three products) to A/B the two arithmetic schemes on the same kernel body:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -Ijperceiver_amd/csrc -DJP_NS=2 tools/ubench/p9s_bench.hip -o ubench_bin/p9s_ns2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "igemm_p9s.h"
#ifndef KGS3
#define KGS3 1      // 16-channel groups per stage of the 3x3 kernels (the library: 1)
#endif
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
        else p[i] = (h & 0x83ff83ffu) | 0x20002000u;     // two 16-bit values, finite as bf16 and as fp16, random sign / mantissa
    }
}
int main(int argc, char** argv) {
    const int N = 8, H = 256, W = 256, C = 256, M = 256;
    const int reps = argc > 1 ? atoi(argv[1]) : 6;
    const size_t nx = (size_t)N * C * H * W, ny = (size_t)N * M * H * W;
    const size_t wbytes = ((size_t)(C / 16) * 9 + 2) * JP_NS * 2 * 256 * 16 + 16;
    float *x, *y, *bias; unsigned* wp;
    hipMalloc(&x, nx * 4); hipMalloc(&y, ny * 4); hipMalloc(&wp, wbytes); hipMalloc(&bias, M * 4);
    fill<<<4096, 256>>>((unsigned*)x, nx, 1u, 1); fill<<<4096, 256>>>(wp, wbytes / 4, 4u, 0);
    { const float hdr[4] = {1.f, 1.f, 0.f, 0.f}; hipMemcpy(wp, hdr, 16, hipMemcpyHostToDevice); }     // pack header {scale, 1 / scale} (JP_NS == 2) fill<<<16, 256>>>((unsigned*)bias, M, 5u, 1);
    FwdEpi e{y, bias, M, H * W, 2};
    float* am; hipMalloc(&am, JP_AMAX_SLOT * 4); hipMemset(am, 0, JP_AMAX_SLOT * 4);      // largest |x| (JP_NS == 2): the fill is in [-2, 2)
    { const float two = 2.f; hipMemcpy(am, &two, 4, hipMemcpyHostToDevice); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 3; ++which) {
        const double macs = (double)M * N * H * W * C * (which == 2 ? 1 : 9);
        for (int r = 0; r < reps; ++r) {
            hipEventRecord(e0);
            if (which == 0)
                hipLaunchKernelGGL((jp_igemm_p9s_wide_kernel<4, 2, true, false, FwdEpi, 9, KGS3>), dim3(N * (H / 8) * (W / 32), 1, 1), dim3(512), 0, 0, wp, x, e, M, C, C / (16 * KGS3), H, W, 0, am);
            else if (which == 1)
                hipLaunchKernelGGL((jp_igemm_p9s_kernel<4, 2, 2, false, false, FwdEpi, 9, KGS3>), dim3(N * (H / 4) * (W / 32), 1, 1), dim3(512), 0, 0, wp, x, e, M, C, C / (16 * KGS3), H, W, 0, am);
            else
                hipLaunchKernelGGL((jp_igemm_p9s_kernel<4, 2, 2, false, false, FwdEpi, 1, 2>), dim3(N * (H / 4) * (W / 32), 1, 1), dim3(512), 0, 0, wp, x, e, M, C, C / 32, H, W, 0, am);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r >= 2) printf("NS=%d %s %.3f ms  %.0f TF fp32-equivalent  %.0f TF executed\n", JP_NS, which == 0 ? "3x3 wide" : which == 1 ? "3x3     " : "1x1     ", ms,
                               2.0 * macs / ms / 1e9, 2.0 * macs * (JP_NS == 2 ? 3 : 6) / ms / 1e9);
        }
    }
    if (hipGetLastError() != hipSuccess) { printf("launch error\n"); return 1; }
    return 0;
}
