// Optimizer side of the step over ONE flat fp32 parameter arena (and matching grad / Adam-moment
// arenas): global L2 grad norm, clip-by-norm (max 35) + Adam(lr 1e-4, betas .9/.999, eps 1e-8, wd 0)
// fused in a single streaming pass that reads the clip coefficient from device memory (no host
// sync).  Replaces mmcv OptimizerHook.clip_grads + torch.optim.Adam.step
// (mono/core/utils/dist_utils.py:58-60, config optimizer/optimizer_config).  Also the counter-based
// RNG used for train-mode Dropout masks and the automask noise (depth_decoder.py:13,52-53; net.py:163).
#include "jp_common.h"
#include <algorithm>
#include <cmath>

namespace {

constexpr int TPB = 256;

// Deterministic global-norm reduction (two stages, no atomics): stage 1 writes one double per block -- the block's
// grid-stride share of g, reduced in a fixed order -- stage 2 sums any number of such partials in a fixed order.
// Every rank of a data-parallel job therefore derives bit-identical clip coefficients from its (bit-identical)
// all-reduced gradients, run after run (the cross-block atomicAdd this replaces depended on block completion order).
constexpr int SUMSQ_BLOCKS = 256;

__global__ __launch_bounds__(TPB) void sumsq_partial_kernel(const float* __restrict__ g, double* __restrict__ partials,
                                                            long n) {
    __shared__ double sm[4];
    double s = 0.0;
    float fs = 0.f;
    int run = 0;
    const long n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long)gridDim.x * TPB) {
        const float4 v = g4[i];
        fs += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        if (++run == 16) { s += fs; fs = 0.f; run = 0; }
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) fs += g[i] * g[i];
    s += fs;
    s = jp_block_sum_d(s, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ __launch_bounds__(TPB) void sum_doubles_kernel(const double* __restrict__ in, double* __restrict__ out, int count) {
    __shared__ double sm[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += TPB) s += in[i];
    s = jp_block_sum_d(s, sm);
    if (threadIdx.x == 0) out[0] = s;
}

// p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps), with g scaled by grad_scale * min(1, max_norm/(norm+1e-6))
__global__ __launch_bounds__(TPB) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n,
                                                   const double* __restrict__ normsq, float grad_scale,
                                                   float max_norm, float lr, float b1, float b2, float omb1, float omb2,
                                                   float eps, float bc1, float bc2, const float* __restrict__ state) {
    if (state) { lr = state[0]; bc1 = state[1]; bc2 = state[2]; }      // captured step: this step's scalars live in device memory
    float coef = grad_scale;
    if (normsq && max_norm > 0.f) {
        const float nrm = (float)sqrt(*normsq) * grad_scale;
        const float c = max_norm / (nrm + 1e-6f);
        if (c < 1.f) coef *= c;
    }
    const float step = lr / bc1, isb2 = 1.f / sqrtf(bc2);
    const long n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
#define JP_ADAM1(P, G, M, V)                                   \
    {                                                          \
        const float gs_ = (G) * coef;                          \
        (M) = b1 * (M) + omb1 * gs_;                           \
        (V) = b2 * (V) + omb2 * gs_ * gs_;                      \
        (P) -= step * (M) / (sqrtf(V) * isb2 + eps);           \
    }
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long)gridDim.x * TPB) {
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const float4 gg = g4[i];
        JP_ADAM1(pp.x, gg.x, mm.x, vv.x) JP_ADAM1(pp.y, gg.y, mm.y, vv.y)
        JP_ADAM1(pp.z, gg.z, mm.z, vv.z) JP_ADAM1(pp.w, gg.w, mm.w, vv.w)
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB)
        JP_ADAM1(p[i], g[i], m[i], v[i])
#undef JP_ADAM1
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

// keep-mask in {0,1}: P(keep) = 1-p
__global__ __launch_bounds__(TPB) void rng_mask_kernel(float* __restrict__ out, long n, uint64_t seed, float p,
                                                       const long* __restrict__ base) {
    if (base) seed += (uint64_t)base[0];            // captured step: seed = this step's base (device memory) + the call's offset
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
        const uint64_t h = mix64((uint64_t)i ^ mix64(seed));
        const float u = (float)(h >> 40) * (1.f / 16777216.f);
        out[i] = u >= p ? 1.f : 0.f;
    }
}

__global__ __launch_bounds__(TPB) void rng_normal_kernel(float* __restrict__ out, long n, uint64_t seed,
                                                         const long* __restrict__ base) {
    if (base) seed += (uint64_t)base[0];
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
        const uint64_t h1 = mix64((uint64_t)i ^ mix64(seed));
        const uint64_t h2 = mix64(h1 ^ 0xD1B54A32D192ED03ULL);
        const float u1 = ((float)(h1 >> 40) + 1.f) * (1.f / 16777216.f);   // (0,1]
        const float u2 = (float)(h2 >> 40) * (1.f / 16777216.f);
        out[i] = sqrtf(-2.f * __logf(u1)) * __cosf(6.28318530718f * u2);
    }
}

__global__ __launch_bounds__(TPB) void fill_kernel(float* __restrict__ out, long n, float v) {
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) out[i] = v;
}

inline int blocks_for(long n) { return (int)std::min<long>((n + TPB * 4 - 1) / (TPB * 4), 4096); }

}  // namespace

#define JP_ST hipStream_t st = (hipStream_t)stream

// Stage 1 of the global gradient norm over one bucket g[0..n): writes jp_sumsq_blocks() doubles to `partials`.
extern "C" int jp_sumsq_blocks(void) { return SUMSQ_BLOCKS; }

extern "C" int jp_grad_sumsq_partials(const float* g, double* partials, long n, void* stream) {
    JP_CHECK_ARG(g && partials && n > 0 && ((uintptr_t)g & 15) == 0, "grad_sumsq_partials: bad args (16-B aligned bucket required)");
    JP_ST;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SUMSQ_BLOCKS), dim3(TPB), 0, st, g, partials, n);
    JP_LAUNCH_CHECK();
}

// Stage 2: out[0] = sum of `count` doubles in a fixed order (all buckets' partials -> the squared global norm).
extern "C" int jp_sum_doubles(const double* in, double* out, int count, void* stream) {
    JP_CHECK_ARG(in && out && count > 0, "sum_doubles: bad args");
    JP_ST;
    hipLaunchKernelGGL(sum_doubles_kernel, dim3(1), dim3(TPB), 0, st, in, out, count);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_adam_clip_step(float* p, const float* g, float* m, float* v, long n, const double* normsq,
                                 float grad_scale, float max_norm, double lr, double beta1, double beta2,
                                 double eps, int step, void* stream) {
    JP_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adam_clip_step: bad args");
    JP_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_clip_step: arenas must be 16-B aligned");
    JP_ST;
    // bias corrections in double on the host, as torch.optim.Adam does with python floats
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n)), dim3(TPB), 0, st, p, g, m, v, n, normsq, grad_scale, max_norm,
                       (float)lr, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)bc1,
                       (float)bc2, (const float*)nullptr);
    JP_LAUNCH_CHECK();
}

// The same pass with the step's scalars read from DEVICE memory: state = {lr, 1 - beta1^step, 1 - beta2^step} (fp32, the
// values jp_adam_clip_step derives on the host).  Nothing that changes from step to step is a kernel argument, so the launch
// can sit in a captured hipGraph of the whole training step (apis/trainer.py CapturedStep); the host refreshes `state` with
// one small H2D copy before each replay.
extern "C" int jp_adam_clip_step_dev(float* p, const float* g, float* m, float* v, long n, const double* normsq,
                                     float grad_scale, float max_norm, double beta1, double beta2, double eps,
                                     const float* state, void* stream) {
    JP_CHECK_ARG(p && g && m && v && state && n > 0, "adam_clip_step_dev: bad args");
    JP_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_clip_step_dev: arenas must be 16-B aligned");
    JP_ST;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n)), dim3(TPB), 0, st, p, g, m, v, n, normsq, grad_scale, max_norm,
                       0.f, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, 1.f, 1.f, state);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_rng_keep_mask(float* out, long n, uint64_t seed, float p_drop, void* stream) {
    JP_CHECK_ARG(out && n > 0, "rng_keep_mask: bad args");
    JP_ST;
    hipLaunchKernelGGL(rng_mask_kernel, dim3(blocks_for(n)), dim3(TPB), 0, st, out, n, seed, p_drop, (const long*)nullptr);
    JP_LAUNCH_CHECK();
}

// seed = base[0] + offset (mod 2^64), base in device memory: the captured-step forms of the two generators (same streams as
// jp_rng_keep_mask / jp_rng_normal called with that seed)
extern "C" int jp_rng_keep_mask_dev(float* out, long n, const long* base, long offset, float p_drop, void* stream) {
    JP_CHECK_ARG(out && base && n > 0, "rng_keep_mask_dev: bad args");
    JP_ST;
    hipLaunchKernelGGL(rng_mask_kernel, dim3(blocks_for(n)), dim3(TPB), 0, st, out, n, (uint64_t)offset, p_drop, base);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_rng_normal_dev(float* out, long n, const long* base, long offset, void* stream) {
    JP_CHECK_ARG(out && base && n > 0, "rng_normal_dev: bad args");
    JP_ST;
    hipLaunchKernelGGL(rng_normal_kernel, dim3(blocks_for(n)), dim3(TPB), 0, st, out, n, (uint64_t)offset, base);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_rng_normal(float* out, long n, uint64_t seed, void* stream) {
    JP_CHECK_ARG(out && n > 0, "rng_normal: bad args");
    JP_ST;
    hipLaunchKernelGGL(rng_normal_kernel, dim3(blocks_for(n)), dim3(TPB), 0, st, out, n, seed, (const long*)nullptr);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_fill(float* out, long n, float v, void* stream) {
    JP_CHECK_ARG(out && n > 0, "fill: bad args");
    JP_ST;
    hipLaunchKernelGGL(fill_kernel, dim3(blocks_for(n)), dim3(TPB), 0, st, out, n, v);
    JP_LAUNCH_CHECK();
}
