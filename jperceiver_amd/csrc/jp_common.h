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

// ---- operand scales of the fp16 split kernels (scale.hip): "largest ordinary magnitude" of a tensor as the bit pattern of a float.
// Inf / NaN and finite magnitudes of 2^100 and more do not take part (they overflow fp16 under the scale of the rest: NaN outputs).
__host__ __device__ __forceinline__ unsigned jp_amag(unsigned bits) {
    const unsigned u = bits & 0x7fffffffu;
    return u >= (227u << 23) ? 0u : u;
}
// the same filter on a value, for producers that keep a running float maximum PER ELEMENT (a lane that filtered only its final maximum
// would drop every ordinary value it saw together with one Inf): |v|, or 0 when v is NaN / Inf / >= 2^100
__device__ __forceinline__ float jp_fmag(float v) {
    const float a = fabsf(v);
    return a < 0x1p100f ? a : 0.f;
}
// A magnitude SLOT is JP_AMAX_WAYS words, JP_AMAX_STRIDE words (one 64-byte line) apart: producers spread their atomicMax over the ways
// (a million same-address device-scope atomics cost 11 ns each -- measured: they made bn_apply 18 x slower; a "skip if the slot already
// holds as much" check needs a device-scope load per workgroup, which doubled the kernel's time), consumers take the maximum of the ways.
constexpr int JP_AMAX_WAYS = 32, JP_AMAX_STRIDE = 16, JP_AMAX_SLOT = JP_AMAX_WAYS * JP_AMAX_STRIDE;     // 512 floats = 2 KB
__device__ __forceinline__ unsigned jp_amax_way() {
    return ((blockIdx.x + 5u * blockIdx.y + 11u * blockIdx.z + (threadIdx.x >> 6)) & (JP_AMAX_WAYS - 1)) * JP_AMAX_STRIDE;
}
// consumer side: the slot's value (uniform over the wave)
__device__ __forceinline__ float jp_slot_amax(const float* __restrict__ slot) {
    const int lane = threadIdx.x & 63;
    unsigned m = lane < JP_AMAX_WAYS ? __float_as_uint(slot[lane * JP_AMAX_STRIDE]) : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    return __uint_as_float(m);
}
// producer side: every wave folds its lanes' running maximum `mx` (of |stored value|) and commits it with one non-returning atomicMax
// (nothing waits for it).  `out` == nullptr: nothing asked for it.
__device__ __forceinline__ void jp_wave_amax_commit(float mx, unsigned* out) {
    if (!out) return;
    unsigned m = jp_amag(__float_as_uint(mx));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out + jp_amax_way(), m);
}
// the same for a whole workgroup of <= 16 waves (one atomic per workgroup; every thread of the workgroup must call it)
__device__ __forceinline__ void jp_block_amax_commit(float mx, unsigned* out) {
    if (!out) return;                                   // (uniform: a kernel argument)
    __shared__ unsigned jp_bam_[16];
    unsigned m = jp_amag(__float_as_uint(mx));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) jp_bam_[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 1; i < nw; ++i) m = max(m, jp_bam_[i]);
        if (m) atomicMax(out + jp_amax_way(), m);
    }
}
// reflect index for ReflectionPad (pad < n): -1 -> 1, n -> n-2
__device__ __forceinline__ int jp_reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}
