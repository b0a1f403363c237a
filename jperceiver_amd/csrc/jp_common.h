// Shared device/host helpers for libjperceiver_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define JP_OK 0
#define JP_EBADARG (-1)

extern "C" void jp_set_last_error(const char* msg);
// opt-in per-kernel profiler (capi.cpp): no-ops unless jp_profile_begin() opened a profile
void jp_prof_before(const char* tag, double flops, hipStream_t st);
void jp_prof_after(hipStream_t st);

#define JP_CHECK_ARG(cond, msg)                         \
    do {                                                \
        if (!(cond)) {                                  \
            jp_set_last_error(msg);                     \
            return JP_EBADARG;                          \
        }                                               \
    } while (0)

#define JP_LAUNCH_CHECK()                               \
    do {                                                \
        hipError_t e_ = hipGetLastError();              \
        if (e_ != hipSuccess) {                         \
            jp_set_last_error(hipGetErrorString(e_));   \
            return (int)e_;                             \
        }                                               \
        return JP_OK;                                   \
    } while (0)

#define JP_HIP(call)                                    \
    do {                                                \
        hipError_t e_ = (call);                         \
        if (e_ != hipSuccess) {                         \
            jp_set_last_error(hipGetErrorString(e_));   \
            return (int)e_;                             \
        }                                               \
    } while (0)

static inline int jp_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// activation codes shared by conv epilogues and jp_act_bwd
enum { JP_ACT_NONE = 0, JP_ACT_RELU = 1, JP_ACT_LEAKY = 2, JP_ACT_SIGMOID = 3 };
enum { JP_PAD_ZERO = 0, JP_PAD_REFLECT = 1 };

__device__ __forceinline__ float jp_act(float v, int act) {
    switch (act) {
        case JP_ACT_RELU: return v > 0.f ? v : 0.f;
        case JP_ACT_LEAKY: return v > 0.f ? v : 0.01f * v;
        case JP_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
        default: return v;
    }
}

// 64-lane wavefront sum (all lanes get the total)
__device__ __forceinline__ float jp_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double jp_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); result valid in thread 0
__device__ __forceinline__ double jp_block_sum_d(double v, double* sm /*>=4*/) {
    v = jp_wave_sum_d(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += sm[i];
    }
    __syncthreads();
    return r;
}

// reflect index for ReflectionPad (pad < n): -1 -> 1, n -> n-2
__device__ __forceinline__ int jp_reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}
