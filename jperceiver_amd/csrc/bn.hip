// Train-mode BatchNorm2d (+ fused residual add and ReLU) forward/backward and the per-channel
// reductions they need.  Replaces nn.BatchNorm2d at resnet.py:21-24,41-45,92 and
// layout_model.py:146,152 (batch statistics, biased var for normalisation, unbiased var into
// running_var with momentum 0.1, eps 1e-5).  All kernels are HBM-bound streaming passes:
// NCHW fp32, one (channel, slice) per workgroup, float4 loads when HW % 4 == 0.
#include "jp_common.h"
#include <algorithm>

namespace {

constexpr int TPB = 256;

// grid (C, S): partial sum / sum of squares of channel c over a slice of the N*HW elements,
// fp32 per-thread accumulation over short runs, double for the block and cross-block combination.
__global__ __launch_bounds__(TPB) void bn_stats_kernel(const float* __restrict__ x, double* __restrict__ sums,
                                                       int C, int HW, int CH, int chunk) {
    __shared__ double sm[4];
    const int c = blockIdx.x;
    const int n = blockIdx.y / CH, ck = blockIdx.y - n * CH;
    const int beg = ck * chunk, end = min(HW, beg + chunk);
    const float* xp = x + ((size_t)n * C + c) * HW;
    double s = 0.0, q = 0.0;
    float fs = 0.f, fq = 0.f;
    int run = 0;
    if (((HW | chunk) & 3) == 0) {      // 16 B per lane: planes and chunks are float4-aligned
        const float4* x4 = reinterpret_cast<const float4*>(xp);
        for (int i = (beg >> 2) + threadIdx.x; i < (end >> 2); i += TPB) {
            const float4 v = x4[i];
            fs += (v.x + v.y) + (v.z + v.w);
            fq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            if (++run == 8) { s += fs; q += fq; fs = fq = 0.f; run = 0; }
        }
    } else {
        for (int i = beg + threadIdx.x; i < end; i += TPB) {
            const float v = xp[i];
            fs += v;
            fq += v * v;
            if (++run == 32) { s += fs; q += fq; fs = fq = 0.f; run = 0; }
        }
    }
    s += fs; q += fq;
    s = jp_block_sum_d(s, sm);
    q = jp_block_sum_d(q, sm);
    if (threadIdx.x == 0) {   // per-workgroup partials, summed in a fixed order by the consumer: no memset, no atomics
        sums[(size_t)(2 * c) * gridDim.y + blockIdx.y] = s;
        sums[(size_t)(2 * c + 1) * gridDim.y + blockIdx.y] = q;
    }
}

// One wave of every apply workgroup folds its channel's partial sums (fixed order -> every workgroup of a channel gets
// the same bits): mean / invstd, and -- by image 0's first workgroup only -- the saved statistics for backward and the
// running-stat momentum update (applied n_updates times: the reference evaluates the layout branch twice per iteration,
// SURVEY.md N4).  Replaces a separate C-thread "finalize" launch per layer (120 launches, 1.8 ms per step).
__device__ __forceinline__ void bn_channel_stats(const double* __restrict__ sums, int c, int S, double count, float eps,
                                                 double* __restrict__ tot /* LDS[2] */, float* mean_out, float* invstd_out,
                                                 double* var_out) {
    if (threadIdx.x < 64) {
        double a = 0.0, b = 0.0;
        for (int i = threadIdx.x; i < S; i += 64) {
            a += sums[(size_t)(2 * c) * S + i];
            b += sums[(size_t)(2 * c + 1) * S + i];
        }
        a = jp_wave_sum_d(a);
        b = jp_wave_sum_d(b);
        if (threadIdx.x == 0) { tot[0] = a; tot[1] = b; }
    }
    __syncthreads();
    const double m = tot[0] / count;
    double var = tot[1] / count - m * m;
    if (var < 0.0) var = 0.0;
    *mean_out = (float)m;
    *invstd_out = (float)(1.0 / sqrt(var + (double)eps));
    *var_out = var;
}

// The normalised value is ALWAYS formed as fmaf(x, sc, sh) with sc = invstd*gamma, sh = beta - mean*sc: the backward
// kernels re-evaluate exactly this expression to recover the ReLU mask of residual-free layers instead of reading y.
__device__ __forceinline__ float bn_affine(float x, float sc, float sh) { return fmaf(x, sc, sh); }
__device__ __forceinline__ float bn_shift(float beta, float mean, float sc) { return fmaf(-mean, sc, beta); }

__device__ __forceinline__ void bn_apply_body(const float* __restrict__ x, const float* __restrict__ residual,
                                              float* __restrict__ y, float sc, float sh, size_t base, int HW, int relu) {
    if ((HW & 3) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(x + base);
        const float4* r4 = residual ? reinterpret_cast<const float4*>(residual + base) : nullptr;
        float4* y4 = reinterpret_cast<float4*>(y + base);
        for (int i = blockIdx.x * TPB + threadIdx.x; i < (HW >> 2); i += gridDim.x * TPB) {
            const float4 a = x4[i];
            float4 v = make_float4(bn_affine(a.x, sc, sh), bn_affine(a.y, sc, sh), bn_affine(a.z, sc, sh), bn_affine(a.w, sc, sh));
            if (r4) { const float4 r = r4[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            y4[i] = v;
        }
    } else {
        for (int i = blockIdx.x * TPB + threadIdx.x; i < HW; i += gridDim.x * TPB) {
            float v = bn_affine(x[base + i], sc, sh);
            if (residual) v += residual[base + i];
            if (relu) v = fmaxf(v, 0.f);
            y[base + i] = v;
        }
    }
}

// y = relu?( (x-mean)*invstd*gamma + beta (+ residual) ), statistics from the stats kernel's partial sums
__global__ __launch_bounds__(TPB) void bn_apply_kernel(const float* __restrict__ x, const double* __restrict__ sums,
                                                       float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                       float* __restrict__ running_mean, float* __restrict__ running_var,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       const float* __restrict__ residual, float* __restrict__ y,
                                                       int C, int HW, int relu, double count, float momentum, float eps,
                                                       int n_updates, int S) {
    __shared__ double tot[2];
    const int nc = blockIdx.y;  // n*C + c
    const int c = nc % C;
    float mean, invstd;
    double var;
    bn_channel_stats(sums, c, S, count, eps, tot, &mean, &invstd, &var);
    if (nc < C && blockIdx.x == 0 && threadIdx.x == 0) {
        save_mean[c] = mean;
        save_invstd[c] = invstd;
        if (running_mean) {
            const float unb = (float)(count > 1.0 ? var * count / (count - 1.0) : var);
            float rm = running_mean[c], rv = running_var[c];
            for (int i = 0; i < n_updates; ++i) {
                rm = (1.f - momentum) * rm + momentum * mean;
                rv = (1.f - momentum) * rv + momentum * unb;
            }
            running_mean[c] = rm;
            running_var[c] = rv;
        }
    }
    const float sc = invstd * gamma[c];
    const float sh = bn_shift(beta[c], mean, sc);
    bn_apply_body(x, residual, y, sc, sh, (size_t)nc * HW, HW, relu);
}

// eval mode: y = relu?( (x-running_mean)/sqrt(running_var+eps)*gamma + beta (+ residual) )
__global__ __launch_bounds__(TPB) void bn_eval_kernel(const float* __restrict__ x, const float* __restrict__ rm,
                                                      const float* __restrict__ rv, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta,
                                                      const float* __restrict__ residual, float* __restrict__ y,
                                                      int C, int HW, float eps, int relu) {
    const int nc = blockIdx.y;
    const int c = nc % C;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    const float sh = bn_shift(beta[c], rm[c], sc);
    bn_apply_body(x, residual, y, sc, sh, (size_t)nc * HW, HW, relu);
}

// backward reduction: per channel sum(dyr), sum(dyr * xhat) with dyr = dy * (y > 0) when relu.
// y == nullptr with relu: residual-free layer, the mask is recomputed as bn_affine(x) > 0 (one tensor read less).
__global__ __launch_bounds__(TPB) void bn_bwd_reduce_kernel(const float* __restrict__ dy,
                                                            const float* __restrict__ x,
                                                            const float* __restrict__ y,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            double* __restrict__ sums, int C, int HW, int CH,
                                                            int chunk, int relu) {
    __shared__ double sm[4];
    const int c = blockIdx.x;
    const int n = blockIdx.y / CH, ck = blockIdx.y - n * CH;
    const int beg = ck * chunk, end = min(HW, beg + chunk);
    const size_t base = ((size_t)n * C + c) * HW;
    const float mu = mean[c], is = invstd[c];
    const float sc = is * gamma[c], sh = bn_shift(beta[c], mu, sc);
    double s = 0.0, q = 0.0;
    float fs = 0.f, fq = 0.f;
    int run = 0;
#define JP_BN_RED1(G, X, Yv)                                              \
    {                                                                     \
        float g_ = (G);                                                   \
        if (relu && !((y ? (Yv) : bn_affine((X), sc, sh)) > 0.f)) g_ = 0.f; \
        fs += g_;                                                         \
        fq += g_ * ((X) - mu) * is;                                       \
    }
    if (((HW | chunk) & 3) == 0) {
        const float4* d4 = reinterpret_cast<const float4*>(dy + base);
        const float4* x4 = reinterpret_cast<const float4*>(x + base);
        const float4* y4 = y ? reinterpret_cast<const float4*>(y + base) : nullptr;
        for (int i = (beg >> 2) + threadIdx.x; i < (end >> 2); i += TPB) {
            const float4 d = d4[i], a = x4[i];
            const float4 yy = y4 ? y4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            JP_BN_RED1(d.x, a.x, yy.x) JP_BN_RED1(d.y, a.y, yy.y) JP_BN_RED1(d.z, a.z, yy.z) JP_BN_RED1(d.w, a.w, yy.w)
            if (++run == 8) { s += fs; q += fq; fs = fq = 0.f; run = 0; }
        }
    } else {
        for (int i = beg + threadIdx.x; i < end; i += TPB) {
            const size_t o = base + i;
            JP_BN_RED1(dy[o], x[o], y[o])
            if (++run == 32) { s += fs; q += fq; fs = fq = 0.f; run = 0; }
        }
    }
#undef JP_BN_RED1
    s += fs; q += fq;
    s = jp_block_sum_d(s, sm);
    q = jp_block_sum_d(q, sm);
    if (threadIdx.x == 0) {
        sums[(size_t)(2 * c) * gridDim.y + blockIdx.y] = s;
        sums[(size_t)(2 * c + 1) * gridDim.y + blockIdx.y] = q;
    }
}

// dx = gamma*invstd * (dyr - sum_dy/cnt - xhat*sum_dy_xhat/cnt);  dres = dyr;  dgamma/dbeta by block (0,c-row 0)
__global__ __launch_bounds__(TPB) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ y,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const double* __restrict__ sums, float* __restrict__ dx,
                                                           float* __restrict__ dres, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int C, int HW, double count,
                                                           int relu, int acc_param_grads, int S) {
    const int nc = blockIdx.y;
    const int c = nc % C;
    const float mu = mean[c], is = invstd[c], g = gamma[c];
    const float sc = is * g, sh = bn_shift(beta[c], mu, sc);
    __shared__ double tot[2];
    if (threadIdx.x < 64) {   // one wave sums the partials (fixed order), the rest of the workgroup waits
        double a = 0.0, b = 0.0;
        for (int i = threadIdx.x; i < S; i += 64) {
            a += sums[(size_t)(2 * c) * S + i];
            b += sums[(size_t)(2 * c + 1) * S + i];
        }
        a = jp_wave_sum_d(a);
        b = jp_wave_sum_d(b);
        if (threadIdx.x == 0) { tot[0] = a; tot[1] = b; }
    }
    __syncthreads();
    const double t1 = tot[0], t2 = tot[1];
    const float k1 = (float)(t1 / count), k2 = (float)(t2 / count);
    if (nc < C && blockIdx.x == 0 && threadIdx.x == 0) {  // image 0 owns the parameter gradients
        const float dg = (float)t2, db = (float)t1;
        dgamma[c] = acc_param_grads ? dgamma[c] + dg : dg;
        dbeta[c] = acc_param_grads ? dbeta[c] + db : db;
    }
    const size_t base = (size_t)nc * HW;
    const float gi = g * is;
#define JP_BN_APP1(D, X, Yv, OUT, RES)                                     \
    {                                                                      \
        float d_ = (D);                                                    \
        if (relu && !((y ? (Yv) : bn_affine((X), sc, sh)) > 0.f)) d_ = 0.f; \
        RES = d_;                                                          \
        OUT = gi * (d_ - k1 - ((X) - mu) * is * k2);                       \
    }
    if ((HW & 3) == 0) {
        const float4* d4 = reinterpret_cast<const float4*>(dy + base);
        const float4* x4 = reinterpret_cast<const float4*>(x + base);
        const float4* y4 = y ? reinterpret_cast<const float4*>(y + base) : nullptr;
        float4* o4 = reinterpret_cast<float4*>(dx + base);
        float4* r4 = dres ? reinterpret_cast<float4*>(dres + base) : nullptr;
        for (int i = blockIdx.x * TPB + threadIdx.x; i < (HW >> 2); i += gridDim.x * TPB) {
            const float4 d = d4[i], a = x4[i];
            const float4 yy = y4 ? y4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 o, r;
            JP_BN_APP1(d.x, a.x, yy.x, o.x, r.x) JP_BN_APP1(d.y, a.y, yy.y, o.y, r.y)
            JP_BN_APP1(d.z, a.z, yy.z, o.z, r.z) JP_BN_APP1(d.w, a.w, yy.w, o.w, r.w)
            o4[i] = o;
            if (r4) r4[i] = r;
        }
    } else {
        for (int i = blockIdx.x * TPB + threadIdx.x; i < HW; i += gridDim.x * TPB) {
            float o, r;
            JP_BN_APP1(dy[base + i], x[base + i], y[base + i], o, r)
            dx[base + i] = o;
            if (dres) dres[base + i] = r;
        }
    }
#undef JP_BN_APP1
}

// generic per-channel sum over (N, HW): out[c] (+)= sum  — conv bias gradients
__global__ __launch_bounds__(TPB) void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          int C, int HW, int CH, int chunk) {
    __shared__ double sm[4];
    const int c = blockIdx.x;
    const int n = blockIdx.y / CH, ck = blockIdx.y - n * CH;
    const int beg = ck * chunk, end = min(HW, beg + chunk);
    const float* xp = x + ((size_t)n * C + c) * HW;
    double s = 0.0;
    float fs = 0.f;
    int run = 0;
    for (int i = beg + threadIdx.x; i < end; i += TPB) {
        fs += xp[i];
        if (++run == 32) { s += fs; fs = 0.f; run = 0; }
    }
    s += fs;
    s = jp_block_sum_d(s, sm);
    if (threadIdx.x == 0) atomicAdd(&out[c], (float)s);
}

// chunks per image so that ~2k workgroups stream the tensor, each at least 2048 elements
void chunking(int N, int C, int HW, int* CH, int* chunk) {
    int ch = std::max(1, 2048 / std::max(1, N * C));
    ch = std::min(ch, std::max(1, HW / 2048));
    *chunk = (HW + ch - 1) / ch;
    *CH = (HW + *chunk - 1) / *chunk;
}

}  // namespace

// ws: jp_bn_ws_doubles(N, C, HW) doubles of caller-owned scratch (per-workgroup partial sums, reduced in a fixed order:
// no memset, no atomics, bit-reproducible statistics).  Saves mean/invstd for backward.
extern "C" long jp_bn_ws_doubles(int N, int C, int HW) {
    int CH, chunk;
    chunking(N, C, HW, &CH, &chunk);
    return 2L * C * N * CH;
}

extern "C" int jp_bn_train_fwd(const float* x, const float* gamma, const float* beta, const float* residual,
                               float* y, float* running_mean, float* running_var, float* save_mean,
                               float* save_invstd, double* ws, int N, int C, int HW, float momentum, float eps,
                               int relu, int n_updates, void* stream) {
    JP_CHECK_ARG(x && gamma && beta && y && save_mean && save_invstd && ws, "bn_train_fwd: null pointer");
    JP_CHECK_ARG(N > 0 && C > 0 && HW > 0, "bn_train_fwd: bad dims");
    hipStream_t st = (hipStream_t)stream;
    int CH, chunk;
    chunking(N, C, HW, &CH, &chunk);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C, N * CH), dim3(TPB), 0, st, x, ws, C, HW, CH, chunk);
    const int gx = std::min(jp_cdiv(HW, 4 * TPB), 64);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(gx, N * C), dim3(TPB), 0, st, x, ws, save_mean, save_invstd, running_mean,
                       running_var, gamma, beta, residual, y, C, HW, relu, (double)N * HW, momentum, eps, n_updates, N * CH);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_bn_train_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* beta,
                               const float* save_mean, const float* save_invstd, float* dx, float* dres,
                               float* dgamma, float* dbeta, double* ws, int N, int C, int HW, int relu,
                               int acc_param_grads, void* stream) {
    JP_CHECK_ARG(dy && x && gamma && beta && save_mean && save_invstd && dx && dgamma && dbeta && ws, "bn_train_bwd: null pointer");
    // y == NULL with relu: legal only for layers WITHOUT a residual input (the mask is recomputed from x)
    hipStream_t st = (hipStream_t)stream;
    int CH, chunk;
    chunking(N, C, HW, &CH, &chunk);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, N * CH), dim3(TPB), 0, st, dy, x, y, save_mean, save_invstd, gamma,
                       beta, ws, C, HW, CH, chunk, relu);
    const int gx = std::min(jp_cdiv(HW, 4 * TPB), 64);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(gx, N * C), dim3(TPB), 0, st, dy, x, y, save_mean, save_invstd, gamma,
                       beta, ws, dx, dres, dgamma, dbeta, C, HW, (double)N * HW, relu, acc_param_grads, N * CH);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_channel_sum(const float* x, float* out, int N, int C, int HW, int accumulate, void* stream) {
    JP_CHECK_ARG(x && out && N > 0 && C > 0 && HW > 0, "channel_sum: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate) JP_HIP(hipMemsetAsync(out, 0, sizeof(float) * C, st));
    int CH, chunk;
    chunking(N, C, HW, &CH, &chunk);
    hipLaunchKernelGGL(channel_sum_kernel, dim3(C, N * CH), dim3(TPB), 0, st, x, out, C, HW, CH, chunk);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_bn_eval_fwd(const float* x, const float* gamma, const float* beta, const float* running_mean,
                              const float* running_var, const float* residual, float* y, int N, int C, int HW,
                              float eps, int relu, void* stream) {
    JP_CHECK_ARG(x && gamma && beta && running_mean && running_var && y && N > 0 && C > 0 && HW > 0, "bn_eval_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    const int gx = std::min(jp_cdiv(HW, 4 * TPB), 64);
    hipLaunchKernelGGL(bn_eval_kernel, dim3(gx, N * C), dim3(TPB), 0, st, x, running_mean, running_var, gamma, beta,
                       residual, y, C, HW, eps, relu);
    JP_LAUNCH_CHECK();
}
