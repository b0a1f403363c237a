// Microbenchmark (GPU-box aid): what does rocprofv3's FETCH_SIZE report per byte actually fetched, by access pattern?
// MI355X_MICROARCH.md (HBM): FETCH_SIZE counts 64 B per 128-B request for wide coalesced streaming reads (x2 correction);
// "other access widths are uncalibrated".  The CGT warp kernels read 4-byte bilinear corners -- VERDICT r03 asks for the factor of
// that pattern before their over-fetch is quoted.  Every kernel below reads a 1.5 GiB source (past the 256 MB Infinity Cache) a
// known number of times; run under `rocprofv3 --pmc FETCH_SIZE` and divide (tools/pmc_dump.py prints the per-kernel counter):
//   copy16   16 B / lane coalesced streaming read                         unique bytes = S
//   read4     4 B / lane coalesced streaming read                         unique bytes = S
//   corners1d 4 corners (y,x) (y,x+1) (y+1,x) (y+1,x+1), 1-D pixel order  unique bytes = S  (2048 consecutive pixels per workgroup: the round-3 warp mapping)
//   corners2d the same corners, 8-row x 256-column tiles per workgroup    unique bytes = S  (the round-4 mapping)
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/fetch_calib.hip -o ubench_bin/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void copy16(const float4* __restrict__ s, float* __restrict__ out, long n4) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = s[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void read4(const float* __restrict__ s, float* __restrict__ out, long n) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc += s[i];
    if (acc == 123.456f) out[0] = acc;
}
// images of H x W floats, NI of them
__global__ void corners1d(const float* __restrict__ s, float* __restrict__ out, int H, int W) {
    const float* c = s + (size_t)blockIdx.y * H * W;
    float acc = 0.f;
    for (int it = 0; it < 8; ++it) {
        const int p = (blockIdx.x * 8 + it) * 256 + threadIdx.x;
        if (p >= H * W) break;
        const int y = p / W, x = p - y * W, x1 = min(x + 1, W - 1), y1 = min(y + 1, H - 1);
        acc += c[y * W + x] + c[y * W + x1] + c[y1 * W + x] + c[y1 * W + x1];
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void corners2d(const float* __restrict__ s, float* __restrict__ out, int H, int W) {
    const float* c = s + (size_t)blockIdx.y * H * W;
    const int x = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (int it = 0; it < 8; ++it) {
        const int y = blockIdx.z * 8 + it;
        if (y >= H || x >= W) break;
        const int x1 = min(x + 1, W - 1), y1 = min(y + 1, H - 1);
        acc += c[y * W + x] + c[y * W + x1] + c[y1 * W + x] + c[y1 * W + x1];
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    const int H = 1024, W = 1024, NI = 384;                 // 384 images x 4 MiB = 1.5 GiB
    const long n = (long)NI * H * W;
    float *s, *out;
    if (hipMalloc(&s, n * 4) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(s, 0, n * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto timed = [&](const char* name, auto launch) {
        launch();
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-10s unique bytes %.3f GB  %.3f ms  %.2f TB/s\n", name, n * 4 / 1e9, ms, n * 4 / ms / 1e9);
    };
    timed("copy16", [&] { hipLaunchKernelGGL(copy16, dim3(8192), dim3(256), 0, 0, (const float4*)s, out, n / 4); });
    timed("read4", [&] { hipLaunchKernelGGL(read4, dim3(8192), dim3(256), 0, 0, s, out, n); });
    timed("corners1d", [&] { hipLaunchKernelGGL(corners1d, dim3(H * W / 2048, NI), dim3(256), 0, 0, s, out, H, W); });
    timed("corners2d", [&] { hipLaunchKernelGGL(corners2d, dim3(W / 256, NI, H / 8), dim3(256), 0, 0, s, out, H, W); });
    return 0;
}
