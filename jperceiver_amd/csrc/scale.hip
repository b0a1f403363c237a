// Largest-magnitude reductions for the operand scales of the fp16 split kernels (scale.h).
#include "scale.h"
#include "igemm_p9s.h"
#include <atomic>
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

// slots of library-launched reductions: a ring per device, one slot per call.  A slot is reused RING calls later -- several training
// steps of launches; the host cannot run that far ahead of the device (the step reads its loss back), and a captured graph owns the
// slots it was captured with for as long as launches outside it number fewer than RING between two replays of the same node.
constexpr int RING = 1 << 12, MAXDEV = 16;      // slots of JP_AMAX_SLOT floats (8 MB per device)
float* g_ring[MAXDEV] = {};
std::atomic<unsigned> g_next{0};
float* next_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
    if (!g_ring[dev]) {
        float* p = nullptr;
        if (hipMalloc(&p, (size_t)RING * JP_AMAX_SLOT * sizeof(float)) != hipSuccess) return nullptr;
        g_ring[dev] = p;
    }
    return g_ring[dev] + (size_t)(g_next.fetch_add(1) % RING) * JP_AMAX_SLOT;
}

thread_local unsigned* g_amax_out = nullptr;       // jp_amax_out: pending request; taken by the next supporting entry point
thread_local int g_amax_out_done = 0;
struct Hint { const float* t; const float* a; };
constexpr int MAXH = 8;
thread_local Hint g_hint[MAXH];
thread_local int g_nh = 0;
const float* find_hint(const float* x) {
    for (int i = 0; i < g_nh; ++i)
        if (g_hint[i].t == x) return g_hint[i].a;
    return nullptr;
}
}  // namespace

unsigned* jp_take_amax_out() {
    unsigned* p = g_amax_out;
    if (p) { g_amax_out = nullptr; g_amax_out_done = 1; }
    return p;
}

const float* jp_amax_of(const float* x, long n, hipStream_t st) {
    if (const float* h = find_hint(x)) return h;
    float* s = next_slot();
    if (!s) { jp_set_last_error("amax: no device slot"); return nullptr; }
    if (launch_amax(x, n, s, st, true) != JP_OK) return nullptr;
    return s;
}
const float* jp_amax_of3(const float* x0, long n0, const float* x1, long n1, const float* x2, long n2, hipStream_t st) {
    const float* xs[3] = {x0, x1, x2};
    const long ns[3] = {n0, n1, n2};
    int live = 0, last = -1;
    for (int i = 0; i < 3; ++i)
        if (xs[i] && ns[i] > 0) { ++live; last = i; }
    if (live == 1) return jp_amax_of(xs[last], ns[last], st);
    float* s = next_slot();
    if (!s) { jp_set_last_error("amax: no device slot"); return nullptr; }
    bool zero = true;
    for (int i = 0; i < 3; ++i) {
        if (!xs[i] || ns[i] <= 0) continue;
        if (launch_amax(xs[i], ns[i], s, st, zero) != JP_OK) return nullptr;      // (a hinted segment is simply reduced again: rare, small)
        zero = false;
    }
    if (zero && launch_amax(nullptr, 0, s, st, true) != JP_OK) return nullptr;
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
// "the next entry point that can, folds max |what it writes| into *slot" (max with what the slot holds: pre-zeroed by the caller).
// Supporting entry points: jp_conv2d_fwd* when a patch kernel runs the layer (y), jp_bn_train_fwd (y), jp_bn_train_bwd (dx),
// jp_act_bwd / jp_act_bwd_bias (dx).  jp_amax_out_done() -> 1 if the request was taken since jp_amax_out, and drops it otherwise:
// call it right after the entry point the request was meant for.
extern "C" int jp_amax_out(float* slot) {
    JP_CHECK_ARG(slot, "amax_out: null pointer");
    g_amax_out = reinterpret_cast<unsigned*>(slot);
    g_amax_out_done = 0;
    return JP_OK;
}
extern "C" int jp_amax_out_done(void) {
    const int d = g_amax_out_done;
    g_amax_out = nullptr;
    g_amax_out_done = 0;
    return d;
}
extern "C" int jp_amax_hint(const float* tensor, const float* amax) {
    JP_CHECK_ARG(tensor && amax, "amax_hint: null pointer");
    JP_CHECK_ARG(g_nh < MAXH, "amax_hint: more than 8 hints pending (jp_amax_hint_clear after the call they are for)");
    g_hint[g_nh++] = Hint{tensor, amax};
    return JP_OK;
}
extern "C" int jp_amax_hint_clear(void) {
    g_nh = 0;
    return JP_OK;
}
