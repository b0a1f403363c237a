// Largest-magnitude reductions for the operand scales of the fp16 split kernels (scale.h).
#include "scale.h"
#include "igemm_p9s.h"
#include <cstdint>

namespace {
constexpr int TPB = 256;
// |x| of finite floats orders like its bit pattern.  Inf / NaN elements (and finite ones of 2^100 and more) do not take part: the scale
// comes from the largest ordinary magnitude, so a non-finite input makes exactly the outputs that read it NaN (s * Inf = Inf,
// Inf - fp16(Inf) = NaN) and leaves every other output as it would be without it.
__device__ __forceinline__ unsigned amag(float v) { return jp_amag(__float_as_uint(v)); }
__global__ __launch_bounds__(TPB) void amax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out) {
    unsigned m = 0;
    const long tid = (long)blockIdx.x * TPB + threadIdx.x, nth = (long)gridDim.x * TPB;
    const long head = min(n, (long)((16 - ((uintptr_t)x & 15)) & 15) >> 2);     // scalars in front of the first 16-byte boundary
    const float4* x4 = reinterpret_cast<const float4*>(x + head);
    const long n4 = (n - head) >> 2;
    for (long i = tid; i < n4; i += nth) {
        const float4 v = x4[i];
        m = max(max(m, amag(v.x)), amag(v.y));
        m = max(max(m, amag(v.z)), amag(v.w));
    }
    if (tid < head) m = max(m, amag(x[tid]));
    const long tail = head + (n4 << 2);
    if (tail + tid < n) m = max(m, amag(x[tail + tid]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    __shared__ unsigned sm[TPB / 64];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < TPB / 64; ++i) m = max(m, sm[i]);
        if (m) atomicMax(out + jp_amax_way(), m);       // a maximum: the result does not depend on the order the blocks arrive in
    }
}

int launch_amax(const float* x, long n, float* out, hipStream_t st, bool zero) {
    if (zero) {
        hipError_t e = hipMemsetAsync(out, 0, JP_AMAX_SLOT * sizeof(float), st);
        if (e != hipSuccess) { jp_set_last_error(hipGetErrorString(e)); return (int)e; }
    }
    if (n > 0) {
        const int blocks = (int)std::min<long>((n + TPB * 16 - 1) / (TPB * 16), 2048);
        hipLaunchKernelGGL(amax_kernel, dim3(blocks), dim3(TPB), 0, st, x, n, reinterpret_cast<unsigned*>(out));
    }
    return JP_OK;
}

// max of up to three slots into `dst` (way by way): the iconv kernels read ONE magnitude for their three channel segments
__global__ __launch_bounds__(64) void amax_fold_kernel(unsigned* __restrict__ dst, const unsigned* __restrict__ s0,
                                                       const unsigned* __restrict__ s1, const unsigned* __restrict__ s2, int keep) {
    const int i = threadIdx.x;
    if (i >= JP_AMAX_WAYS) return;
    unsigned m = keep ? dst[i * JP_AMAX_STRIDE] : 0u;
    if (s0) m = max(m, s0[i * JP_AMAX_STRIDE]);
    if (s1) m = max(m, s1[i * JP_AMAX_STRIDE]);
    if (s2) m = max(m, s2[i * JP_AMAX_STRIDE]);
    dst[i * JP_AMAX_STRIDE] = m;
}

float* next_ws_slot(const JpCall& c) {
    if (!c.ax || !c.ax->ws || c.ax->ws_used >= JP_AMAX_WS_SLOTS) {
        jp_set_last_error("conv: an operand's magnitude was not passed (amax_x / amax_dy) and amax_ws is NULL or used up -- pass "
                          "jp_conv2d_amax_ws_floats() floats of scratch");
        return nullptr;
    }
    return c.ax->ws + (size_t)(c.ax->ws_used++) * JP_AMAX_SLOT;
}
}  // namespace

const float* jp_amax_of(const float* x, long n, const JpCall& c) {
    if (c.ax)
        if (const float* h = c.ax->find(x)) return h;
    float* s = next_ws_slot(c);
    if (!s) return nullptr;
    if (launch_amax(x, n, s, c.st, true) != JP_OK) return nullptr;
    c.ax->know(x, s);                   // (the weight-gradient entry points ask for dY once per source segment)
    return s;
}
const float* jp_amax_of3(const float* x0, long n0, const float* x1, long n1, const float* x2, long n2, const JpCall& c) {
    const float* xs[3] = {x0, x1, x2};
    const long ns[3] = {n0, n1, n2};
    const float* known[3] = {nullptr, nullptr, nullptr};
    int live = 0, last = -1, nknown = 0;
    for (int i = 0; i < 3; ++i)
        if (xs[i] && ns[i] > 0) {
            ++live;
            last = i;
            if (c.ax && (known[i] = c.ax->find(xs[i]))) ++nknown;
        }
    if (live == 1) return jp_amax_of(xs[last], ns[last], c);
    float* s = next_ws_slot(c);
    if (!s) return nullptr;
    bool zero = true;                   // segments without a slot of their own are reduced into s, the known ones folded in behind them
    for (int i = 0; i < 3; ++i) {
        if (!xs[i] || ns[i] <= 0 || known[i]) continue;
        if (launch_amax(xs[i], ns[i], s, c.st, zero) != JP_OK) return nullptr;
        zero = false;
    }
    if (nknown || zero)
        hipLaunchKernelGGL(amax_fold_kernel, dim3(1), dim3(64), 0, c.st, reinterpret_cast<unsigned*>(s),
                           reinterpret_cast<const unsigned*>(known[0]), reinterpret_cast<const unsigned*>(known[1]),
                           reinterpret_cast<const unsigned*>(known[2]), zero ? 0 : 1);
    return s;
}

// ---- C ABI (include/jperceiver_hip.h)
extern "C" int jp_amax_slot_floats(void) { return JP_AMAX_SLOT; }
extern "C" int jp_amax(const float* x, long n, float* out, void* stream) {
    JP_CHECK_ARG(out && (x || n == 0) && n >= 0, "amax: bad arguments");
    const int rc = launch_amax(x, n, out, static_cast<hipStream_t>(stream), true);
    if (rc != JP_OK) return rc;
    JP_LAUNCH_CHECK();
}
// as jp_amax, but *out is not zeroed first: it must hold 0 (or an earlier maximum to extend) -- callers that hand out slots of a
// buffer they zero once save a memset per reduction
extern "C" int jp_amax_into(const float* x, long n, float* out, void* stream) {
    JP_CHECK_ARG(out && (x || n == 0) && n >= 0, "amax_into: bad arguments");
    const int rc = launch_amax(x, n, out, static_cast<hipStream_t>(stream), false);
    if (rc != JP_OK) return rc;
    JP_LAUNCH_CHECK();
}
extern "C" int jp_conv2d_amax_ws_floats(void) { return JP_AMAX_WS_SLOTS * JP_AMAX_SLOT; }
