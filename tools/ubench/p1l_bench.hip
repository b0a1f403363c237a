// Standalone harness for the persistent 1x1 kernel (igemm_p1l.h) at the CRP shape (8 x 256 -> 256 @256^2, no bias): random input and
// weight bits, HIP-event time, and a bit-for-bit comparison with the patch kernel it replaces (jp_igemm_p9s_kernel<4, 2, 2, .., 1, 2>,
// same pack, same products in the same order).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -Ijperceiver_amd/csrc tools/ubench/p1l_bench.hip -o ubench_bin/p1l_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "igemm_p1l.h"
#if JP_NS != 3
#error "the persistent 1x1 kernel reads the three-plane bf16 pack: build this harness with -DJP_NS=3"
#endif
struct FwdEpi {
    typedef size_t St;
    float* y; const float* bias; int Cout, OHW, act;
    __device__ __forceinline__ St col(int p) const { int img = p / OHW; return (size_t)img * Cout * OHW + (p - img * OHW); }
    __device__ __forceinline__ void put(St base, int m, float v) const { if (bias) v += bias[m]; y[base + (size_t)m * OHW] = jp_act(v, act); }
    __device__ __forceinline__ void put4(St base, int m, float4 v) const {
        const float b = bias ? bias[m] : 0.f;
        *reinterpret_cast<float4*>(y + base + (size_t)m * OHW) =
            make_float4(jp_act(v.x + b, act), jp_act(v.y + b, act), jp_act(v.z + b, act), jp_act(v.w + b, act));
    }
};
__global__ void fill(unsigned* p, size_t n, unsigned seed, int as_float) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        if (as_float) { float f = ((h & 0xffffff) / 16777216.0f - 0.5f) * 4.f; p[i] = __float_as_uint(f); }
        else p[i] = (h & 0x807f807fu) | 0x3c003c00u;
    }
}
__global__ void diff(const float* a, const float* b, size_t n, unsigned long long* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    unsigned long long d = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) d += __float_as_uint(a[i]) != __float_as_uint(b[i]);
    if (d) atomicAdd(out, d);
}
int main(int argc, char** argv) {
    const int N = 8, H = argc > 2 ? atoi(argv[2]) : 256, W = H, C = 256, M = 256, NST = C / 32;
    const int reps = argc > 1 ? atoi(argv[1]) : 6;
    const int G = argc > 3 ? atoi(argv[3]) : 256;
    const size_t nx = (size_t)N * C * H * W, ny = (size_t)N * M * H * W;
    const size_t SB = 3 * 2 * 256 * 16, wbytes = ((size_t)NST * 2 + 1) * SB * (M / 256);
    float *x, *y0, *y1; unsigned* wp; unsigned long long* nd;
    hipMalloc(&x, nx * 4); hipMalloc(&y0, ny * 4); hipMalloc(&y1, ny * 4); hipMalloc(&wp, wbytes); hipMalloc(&nd, 8);
    fill<<<4096, 256>>>((unsigned*)x, nx, 1u, 1); fill<<<4096, 256>>>(wp, wbytes / 4, 4u, 0);
    hipMemset(y0, 0, ny * 4); hipMemset(y1, 0xff, ny * 4); hipMemset(nd, 0, 8);
    FwdEpi e0{y0, nullptr, M, H * W, 0}, e1{y1, nullptr, M, H * W, 0};
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const double flops = 6.0 * 2.0 * M * (double)N * H * W * C;
    const int ntiles = N * (H / 4) * (W / 32), tpw = (ntiles + G - 1) / G;
    for (int r = 0; r < reps; ++r) {
        float ms0, ms1;
        hipEventRecord(a);
        hipLaunchKernelGGL((jp_igemm_p9s_kernel<4, 2, 2, false, false, FwdEpi, 1, 2>), dim3(ntiles, M / 256, 1), dim3(512), 0, 0, wp, x, e0, M, C, NST, H, W, 0, (const float*)nullptr);   // (JP_NS == 3 build only: no operand scales)
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms0, a, b);
        hipEventRecord(a);
        hipLaunchKernelGGL((jp_conv1x1_p1l_kernel<FwdEpi>), dim3((ntiles + tpw - 1) / tpw, M / 256, 1), dim3(512), 0, 0, wp, x, e1, M, C, NST, H, W, ntiles, tpw, (int)(nx * 4));
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms1, a, b);
        if (r >= 2) printf("patch kernel %.3f ms %.0f TF | P1L (%d workgroups x %d tiles) %.3f ms %.0f TF\n", ms0, flops / ms0 / 1e9, (ntiles + tpw - 1) / tpw, tpw, ms1, flops / ms1 / 1e9);
    }
#ifdef P1L_TRACE
    {
        unsigned long long tr[96];
        hipMemcpyFromSymbol(tr, HIP_SYMBOL(jp_p1l_trace), sizeof(tr));
        const char* nm[4] = {"issued", "copies", "barrier", "next"};
        for (int w = 0; w < 2; ++w) {
            printf("wave %d: stamps per stage = MFMAs issued | copies landed | barrier passed | next stage begins (cycles since the previous stamp)\n", 4 * w);
            for (int i = 1; i < 48; ++i) printf("%s%6lld%s", i % 4 == 1 ? "   " : " ", (long long)(tr[w * 48 + i] - tr[w * 48 + i - 1]), i % 4 == 0 ? "\n" : "");
            printf("\n");
            (void)nm;
        }
    }
#endif
    diff<<<2048, 256>>>(y0, y1, ny, nd);
    unsigned long long d; hipMemcpy(&d, nd, 8, hipMemcpyDeviceToHost);
    printf("%s: %llu of %zu outputs differ  (%s)\n", d ? "MISMATCH" : "bit-identical", d, ny, hipGetLastError() == hipSuccess ? "ok" : "LAUNCH ERROR");
    return d != 0;
}
