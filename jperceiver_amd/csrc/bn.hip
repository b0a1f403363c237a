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

// Statistics that arrive as partial sums from the producing convolution's epilogue (igemm_p9s.h; conv_stats[(c * 2 + {0, 1}) * parts + p],
// fp32 sums over <= 128 pixels each): one workgroup per channel folds parts [p0, p0 + np) in double, in a fixed order, into the layout
// bn_apply reads with S = 1.  Replaces bn_stats_kernel's pass over the tensor (round 6).
__global__ __launch_bounds__(TPB) void bn_stats_fold_kernel(const float* __restrict__ conv_stats, double* __restrict__ sums, int parts,
                                                            int p0, int np) {
    __shared__ double sm[4];
    const int c = blockIdx.x;
    const float* a = conv_stats + ((size_t)c * 2 + 0) * parts + p0;
    const float* b = conv_stats + ((size_t)c * 2 + 1) * parts + p0;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < np; i += TPB) { s += (double)a[i]; q += (double)b[i]; }
    s = jp_block_sum_d(s, sm);
    q = jp_block_sum_d(q, sm);
    if (threadIdx.x == 0) { sums[2 * c] = s; sums[2 * c + 1] = q; }
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
                                              float* __restrict__ y, float sc, float sh, size_t base, int HW, int relu,
                                              unsigned* __restrict__ amax = nullptr) {
    float mx = 0.f;                      // largest |y| this thread wrote (-> amax_y)
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
            mx = fmaxf(fmaxf(mx, fmaxf(jp_fmag(v.x), jp_fmag(v.y))), fmaxf(jp_fmag(v.z), jp_fmag(v.w)));
        }
    } else {
        for (int i = blockIdx.x * TPB + threadIdx.x; i < HW; i += gridDim.x * TPB) {
            float v = bn_affine(x[base + i], sc, sh);
            if (residual) v += residual[base + i];
            if (relu) v = fmaxf(v, 0.f);
            y[base + i] = v;
            mx = fmaxf(mx, jp_fmag(v));
        }
    }
    jp_block_amax_commit(mx, amax);
}

// y = relu?( (x-mean)*invstd*gamma + beta (+ residual) ), statistics from the stats kernel's partial sums
__global__ __launch_bounds__(TPB) void bn_apply_kernel(const float* __restrict__ x, const double* __restrict__ sums,
                                                       float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                       float* __restrict__ running_mean, float* __restrict__ running_var,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       const float* __restrict__ residual, float* __restrict__ y,
                                                       int C, int HW, int relu, double count, float momentum, float eps,
                                                       int n_updates, int S, unsigned* __restrict__ amax) {
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
    bn_apply_body(x, residual, y, sc, sh, (size_t)nc * HW, HW, relu, amax);
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
                                                           int relu, int acc_param_grads, int S, unsigned* __restrict__ amax) {
    float mx = 0.f;                      // largest |dx| this thread wrote (-> amax_dx)
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
            mx = fmaxf(fmaxf(mx, fmaxf(jp_fmag(o.x), jp_fmag(o.y))), fmaxf(jp_fmag(o.z), jp_fmag(o.w)));
            if (r4) r4[i] = r;
        }
    } else {
        for (int i = blockIdx.x * TPB + threadIdx.x; i < HW; i += gridDim.x * TPB) {
            float o, r;
            JP_BN_APP1(dy[base + i], x[base + i], y[base + i], o, r)
            dx[base + i] = o;
            mx = fmaxf(mx, jp_fmag(o));
            if (dres) dres[base + i] = r;
        }
    }
    jp_block_amax_commit(mx, amax);
#undef JP_BN_APP1
}

// generic per-channel sum over (N, HW): out[c] (+)= sum  — conv bias gradients
__global__ __launch_bounds__(TPB) void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          int C, int HW, int CH, int chunk, float* __restrict__ part) {
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
    if (threadIdx.x == 0) {
        if (part) part[(size_t)c * gridDim.y + blockIdx.y] = (float)s;      // fixed-order fold below: bit-reproducible
        else atomicAdd(&out[c], (float)s);
    }
}
// out[c] += part[c][0] + part[c][1] + ... in that order (one thread per channel)
__global__ void channel_fold_kernel(const float* __restrict__ part, float* __restrict__ out, int C, int S) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int i = 0; i < S; ++i) s += part[(size_t)c * S + i];
    out[c] += s;
}


// ------------------------------------------------------------------------------------------------------------------------
// ResNet stem tail fused: BatchNorm2d (train) + ReLU + MaxPool2d(3, stride 2, pad 1)  (resnet.py:92-94; the decoders never
// read the normalised 64-channel half-resolution map, only its pooled version).  Rounds 1-3 wrote it (bn_apply), read it
// (max-pool) and, in backward, wrote and re-read its gradient (max-pool backward -> bn reduce + apply): at 8 x 64 x 512^2
// that is 0.54 GB per pass.  Here the forward pools relu(fmaf(x, sc, sh)) straight off the convolution output and the
// backward gathers the pooled gradient through the argmax bytes inside the BatchNorm reduce / apply kernels.
// Results: bit-identical to bn_apply + maxpool_fwd_t_kernel<3, 2> in forward (same expression, same first-maximum tie rule);
// the backward sums the same terms in another order.
constexpr int SP_TW = 64, SP_TH = 16;            // pooled tile per workgroup (4 outputs per thread)

__global__ __launch_bounds__(TPB) void bn_pool_fwd_kernel(const float* __restrict__ x, const double* __restrict__ sums,
                                                          float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ y, uint8_t* __restrict__ idx, int C, int H, int W,
                                                          int OH, int OW, double count, float momentum, float eps, int n_updates,
                                                          int S, unsigned* __restrict__ amax) {
    constexpr int PW = (SP_TW - 1) * 2 + 3, PH = (SP_TH - 1) * 2 + 3;
    __shared__ float tile[PW * PH];
    __shared__ double tot[2];
    const int nc = blockIdx.z, c = nc % C;
    float mean, invstd;
    double var;
    bn_channel_stats(sums, c, S, count, eps, tot, &mean, &invstd, &var);
    if (nc < C && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
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
    const float sc = invstd * gamma[c], sh = bn_shift(beta[c], mean, sc);
    const float* xp = x + (size_t)nc * H * W;
    const int ox0 = blockIdx.x * SP_TW, oy0 = blockIdx.y * SP_TH;
    const int ix0 = ox0 * 2 - 1, iy0 = oy0 * 2 - 1;
    for (int i = threadIdx.x; i < PW * PH; i += TPB) {
        const int ly = i / PW, lx = i - ly * PW;
        const int iy = iy0 + ly, ix = ix0 + lx;
        tile[i] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? fmaxf(bn_affine(xp[iy * W + ix], sc, sh), 0.f) : -INFINITY;
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, q = threadIdx.x >> 6;
    float best[4];
    int bi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { best[j] = -INFINITY; bi[j] = 0; }
    const float* base = tile + (q * 4 * 2) * PW + tx * 2;
    // separable first-maximum scan of pointwise.hip's maxpool_fwd_t_kernel<3, 2> (a NaN always takes over)
#pragma unroll
    for (int r = 0; r < 3 * 2 + 3; ++r) {
        float rv = base[r * PW];
        int rk = 0;
#pragma unroll
        for (int kx = 1; kx < 3; ++kx) {
            const float v = base[r * PW + kx];
            const bool take = v > rv || v != v;
            rv = take ? v : rv;
            rk = take ? kx : rk;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ky = r - j * 2;
            if (ky < 0 || ky >= 3) continue;
            const bool take = ky == 0 || rv > best[j] || rv != rv;
            best[j] = take ? rv : best[j];
            bi[j] = take ? ky * 3 + rk : bi[j];
        }
    }
    const int ox = ox0 + tx;
    float mx = 0.f;                      // largest pooled value this thread wrote (-> amax_y: the first residual block's operand scale)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int oy = oy0 + q * 4 + j;
        if (ox < OW && oy < OH) {
            y[(size_t)nc * OH * OW + oy * OW + ox] = best[j];
            idx[(size_t)nc * OH * OW + oy * OW + ox] = (uint8_t)bi[j];
            mx = fmaxf(mx, jp_fmag(best[j]));
        }
    }
    jp_wave_amax_commit(mx, amax);
}

// backward reduction over the POOLED gradient: sum(g), sum(g * xhat) at the argmax positions whose relu is open
__global__ __launch_bounds__(TPB) void bn_pool_bwd_reduce_kernel(const float* __restrict__ dpool, const uint8_t* __restrict__ idx,
                                                                 const float* __restrict__ x, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, double* __restrict__ sums, int C,
                                                                 int H, int W, int OH, int OW, int CH, int chunk) {
    __shared__ double sm[4];
    const int c = blockIdx.x;
    const int n = blockIdx.y / CH, ck = blockIdx.y - n * CH;
    const int OHW = OH * OW;
    const int beg = ck * chunk, end = min(OHW, beg + chunk);
    const size_t nc = (size_t)n * C + c;
    const float* xp = x + nc * H * W;
    const float mu = mean[c], is = invstd[c];
    const float sc = is * gamma[c], sh = bn_shift(beta[c], mu, sc);
    double s = 0.0, q = 0.0;
    float fs = 0.f, fq = 0.f;
    int run = 0;
    for (int o = beg + threadIdx.x; o < end; o += TPB) {
        const int oy = o / OW, ox = o - oy * OW;
        const int k = idx[nc * OHW + o];
        const int iy = 2 * oy - 1 + k / 3, ix = 2 * ox - 1 + k % 3;
        const float v = xp[iy * W + ix];          // the argmax of a window is always inside the map
        float g = dpool[nc * OHW + o];
        if (!(bn_affine(v, sc, sh) > 0.f)) g = 0.f;
        fs += g;
        fq += g * (v - mu) * is;
        if (++run == 32) { s += fs; q += fq; fs = fq = 0.f; run = 0; }
    }
    s += fs; q += fq;
    s = jp_block_sum_d(s, sm);
    q = jp_block_sum_d(q, sm);
    if (threadIdx.x == 0) {
        sums[(size_t)(2 * c) * gridDim.y + blockIdx.y] = s;
        sums[(size_t)(2 * c + 1) * gridDim.y + blockIdx.y] = q;
    }
}

// one workgroup per channel: fold the partials (fixed order) -> k[c] = {sum_dy / count, sum_dy_xhat / count}, dgamma, dbeta
__global__ __launch_bounds__(64) void bn_pool_bwd_fold_kernel(const double* __restrict__ sums, float* __restrict__ k12,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, int S,
                                                              double count, int acc_param_grads) {
    const int c = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < S; i += 64) {
        a += sums[(size_t)(2 * c) * S + i];
        b += sums[(size_t)(2 * c + 1) * S + i];
    }
    a = jp_wave_sum_d(a);
    b = jp_wave_sum_d(b);
    if (threadIdx.x == 0) {
        k12[2 * c] = (float)(a / count);
        k12[2 * c + 1] = (float)(b / count);
        dgamma[c] = acc_param_grads ? dgamma[c] + (float)b : (float)b;
        dbeta[c] = acc_param_grads ? dbeta[c] + (float)a : (float)a;
    }
}

// dx of the convolution output: the pooled gradient gathered through the argmax bytes (2x2 input cell <- the four outputs
// that can select it, pointwise.hip maxpool3s2_bwd_kernel), relu mask recomputed from x, BatchNorm input gradient on the fly
__global__ __launch_bounds__(TPB) void bn_pool_bwd_apply_kernel(const float* __restrict__ dpool, const uint8_t* __restrict__ idx,
                                                                const float* __restrict__ x, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ k12,
                                                                float* __restrict__ dx, long units, int C, int OH, int OW, int RB,
                                                                int bands) {
    const long u = (long)blockIdx.x * TPB + threadIdx.x;
    if (u >= units) return;
    const int OW2 = OW >> 1, W = 2 * OW;
    const int q = (int)(u % OW2);
    const long v = u / OW2;
    const int b = (int)(v % bands);
    const long nc = v / bands;
    const int c = (int)(nc % C);
    const float mu = mean[c], is = invstd[c], gm = gamma[c];
    const float sc = is * gm, sh = bn_shift(beta[c], mu, sc), gi = gm * is;
    const float k1 = k12[2 * c], k2 = k12[2 * c + 1];
    const int i0 = b * RB, i1 = min(OH, i0 + RB);
    const float* dp = dpool + nc * (long)OH * OW + 2 * q;
    const uint8_t* ip = idx + nc * (long)OH * OW + 2 * q;
    const float* xp = x + nc * 4L * OH * OW + 4 * q;
    float* op = dx + nc * 4L * OH * OW + 4 * q;
    const bool right = 2 * q + 2 < OW;
    float g[2][3];
    int k[2][3];
    auto load = [&](int slot, int i) {
        if (i < OH) {
            const float2 a = *reinterpret_cast<const float2*>(dp + (long)i * OW);
            const uchar2 cc = *reinterpret_cast<const uchar2*>(ip + (long)i * OW);
            g[slot][0] = a.x; g[slot][1] = a.y; k[slot][0] = cc.x; k[slot][1] = cc.y;
            g[slot][2] = right ? dp[(long)i * OW + 2] : 0.f;
            k[slot][2] = right ? (int)ip[(long)i * OW + 2] : -1;
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) { g[slot][j] = 0.f; k[slot][j] = -1; }
        }
    };
    auto fin = [&](float d, float xv) {
        if (!(bn_affine(xv, sc, sh) > 0.f)) d = 0.f;
        return gi * (d - k1 - (xv - mu) * is * k2);
    };
    load(0, i0);
    for (int i = i0; i < i1; ++i) {
        load(1, i + 1);
        const float4 x0 = *reinterpret_cast<const float4*>(xp + (long)(2 * i) * W);
        const float4 x1 = *reinterpret_cast<const float4*>(xp + (long)(2 * i + 1) * W);
        float4 r0, r1;
        r0.x = k[0][0] == 4 ? g[0][0] : 0.f;
        r0.y = (k[0][0] == 5 ? g[0][0] : 0.f) + (k[0][1] == 3 ? g[0][1] : 0.f);
        r0.z = k[0][1] == 4 ? g[0][1] : 0.f;
        r0.w = (k[0][1] == 5 ? g[0][1] : 0.f) + (k[0][2] == 3 ? g[0][2] : 0.f);
        r1.x = (k[0][0] == 7 ? g[0][0] : 0.f) + (k[1][0] == 1 ? g[1][0] : 0.f);
        r1.y = (k[0][0] == 8 ? g[0][0] : 0.f) + (k[0][1] == 6 ? g[0][1] : 0.f) + (k[1][0] == 2 ? g[1][0] : 0.f) + (k[1][1] == 0 ? g[1][1] : 0.f);
        r1.z = (k[0][1] == 7 ? g[0][1] : 0.f) + (k[1][1] == 1 ? g[1][1] : 0.f);
        r1.w = (k[0][1] == 8 ? g[0][1] : 0.f) + (k[0][2] == 6 ? g[0][2] : 0.f) + (k[1][1] == 2 ? g[1][1] : 0.f) + (k[1][2] == 0 ? g[1][2] : 0.f);
        *reinterpret_cast<float4*>(op + (long)(2 * i) * W) = make_float4(fin(r0.x, x0.x), fin(r0.y, x0.y), fin(r0.z, x0.z), fin(r0.w, x0.w));
        *reinterpret_cast<float4*>(op + (long)(2 * i + 1) * W) = make_float4(fin(r1.x, x1.x), fin(r1.y, x1.y), fin(r1.z, x1.z), fin(r1.w, x1.w));
#pragma unroll
        for (int j = 0; j < 3; ++j) { g[0][j] = g[1][j]; k[0][j] = k[1][j]; }
    }
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
                               int relu, int n_updates, float* amax_y, const float* conv_stats, int conv_parts, void* stream) {
    // amax_y: optional magnitude slot (jp_amax_slot_floats floats, see the header); the apply kernel folds max |y| into it.
    // conv_stats / conv_parts: the partial sums the producing convolution left (jp_conv2d_fwd* bn_stats, *bn_stats_parts > 0) -- the
    // statistics are then folded from them instead of read off x (NULL / 0: the statistics pass over x runs)
    JP_CHECK_ARG(x && gamma && beta && y && save_mean && save_invstd && ws, "bn_train_fwd: null pointer");
    JP_CHECK_ARG(N > 0 && C > 0 && HW > 0, "bn_train_fwd: bad dims");
    hipStream_t st = (hipStream_t)stream;
    int CH, chunk;
    chunking(N, C, HW, &CH, &chunk);
    int S = N * CH;
    if (conv_stats && conv_parts > 0) {
        hipLaunchKernelGGL(bn_stats_fold_kernel, dim3(C), dim3(TPB), 0, st, conv_stats, ws, conv_parts, 0, conv_parts);
        S = 1;          // one folded (sum, sum of squares) pair per channel
    } else {
        hipLaunchKernelGGL(bn_stats_kernel, dim3(C, N * CH), dim3(TPB), 0, st, x, ws, C, HW, CH, chunk);
    }
    const int gx = std::min(jp_cdiv(HW, 4 * TPB), 64);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(gx, N * C), dim3(TPB), 0, st, x, ws, save_mean, save_invstd, running_mean,
                       running_var, gamma, beta, residual, y, C, HW, relu, (double)N * HW, momentum, eps, n_updates, S,
                       reinterpret_cast<unsigned*>(amax_y));
    JP_LAUNCH_CHECK();
}

extern "C" int jp_bn_train_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* beta,
                               const float* save_mean, const float* save_invstd, float* dx, float* dres,
                               float* dgamma, float* dbeta, double* ws, int N, int C, int HW, int relu,
                               int acc_param_grads, float* amax_dx, void* stream) {
    JP_CHECK_ARG(dy && x && gamma && beta && save_mean && save_invstd && dx && dgamma && dbeta && ws, "bn_train_bwd: null pointer");
    // y == NULL with relu: legal only for layers WITHOUT a residual input (the mask is recomputed from x)
    hipStream_t st = (hipStream_t)stream;
    int CH, chunk;
    chunking(N, C, HW, &CH, &chunk);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, N * CH), dim3(TPB), 0, st, dy, x, y, save_mean, save_invstd, gamma,
                       beta, ws, C, HW, CH, chunk, relu);
    const int gx = std::min(jp_cdiv(HW, 4 * TPB), 64);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(gx, N * C), dim3(TPB), 0, st, dy, x, y, save_mean, save_invstd, gamma,
                       beta, ws, dx, dres, dgamma, dbeta, C, HW, (double)N * HW, relu, acc_param_grads, N * CH,
                       reinterpret_cast<unsigned*>(amax_dx));
    JP_LAUNCH_CHECK();
}

// BatchNorm2d (train) + ReLU + MaxPool2d(3, 2, 1) in one pass over the convolution output x (N, C, H, W), H and W multiples of 4:
// pooled (N, C, H/2, W/2) + the argmax byte (0..8 = ky * 3 + kx) per pooled element; statistics / running-stat update exactly
// as jp_bn_train_fwd.  ws = jp_bn_ws_doubles(N, C, H * W) doubles.
extern "C" int jp_bn_relu_pool_fwd(const float* x, const float* gamma, const float* beta, float* pooled, uint8_t* idx,
                                   float* running_mean, float* running_var, float* save_mean, float* save_invstd, double* ws,
                                   int N, int C, int H, int W, float momentum, float eps, int n_updates, float* amax_y,
                                   void* stream) {
    JP_CHECK_ARG(x && gamma && beta && pooled && idx && save_mean && save_invstd && ws, "bn_relu_pool_fwd: null pointer");
    JP_CHECK_ARG(N > 0 && C > 0 && H >= 4 && W >= 4 && H % 4 == 0 && W % 4 == 0, "bn_relu_pool_fwd: H and W must be multiples of 4");
    hipStream_t st = (hipStream_t)stream;
    const int HW = H * W, OH = H / 2, OW = W / 2;
    int CH, chunk;
    chunking(N, C, HW, &CH, &chunk);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C, N * CH), dim3(TPB), 0, st, x, ws, C, HW, CH, chunk);
    hipLaunchKernelGGL(bn_pool_fwd_kernel, dim3(jp_cdiv(OW, SP_TW), jp_cdiv(OH, SP_TH), N * C), dim3(TPB), 0, st, x, ws, save_mean,
                       save_invstd, running_mean, running_var, gamma, beta, pooled, idx, C, H, W, OH, OW, (double)N * HW, momentum,
                       eps, n_updates, N * CH, reinterpret_cast<unsigned*>(amax_y));
    JP_LAUNCH_CHECK();
}

// backward of the above: dpool (N, C, H/2, W/2) -> dx (N, C, H, W), dgamma / dbeta (= or +=).  ws = jp_bn_ws_doubles(N, C, H*W/4)
// doubles followed by 2*C floats (use jp_bn_relu_pool_bwd_ws_doubles).
extern "C" long jp_bn_relu_pool_bwd_ws_doubles(int N, int C, int H, int W) {
    return jp_bn_ws_doubles(N, C, (H / 2) * (W / 2)) + C;
}
extern "C" int jp_bn_relu_pool_bwd(const float* dpool, const uint8_t* idx, const float* x, const float* gamma, const float* beta,
                                   const float* save_mean, const float* save_invstd, float* dx, float* dgamma, float* dbeta,
                                   double* ws, int N, int C, int H, int W, int acc_param_grads, void* stream) {
    JP_CHECK_ARG(dpool && idx && x && gamma && beta && save_mean && save_invstd && dx && dgamma && dbeta && ws,
                 "bn_relu_pool_bwd: null pointer");
    JP_CHECK_ARG(N > 0 && C > 0 && H >= 4 && W >= 4 && H % 4 == 0 && W % 4 == 0, "bn_relu_pool_bwd: H and W must be multiples of 4");
    hipStream_t st = (hipStream_t)stream;
    const int OH = H / 2, OW = W / 2;
    int CH, chunk;
    chunking(N, C, OH * OW, &CH, &chunk);
    float* k12 = reinterpret_cast<float*>(ws + 2L * C * N * CH);
    hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel, dim3(C, N * CH), dim3(TPB), 0, st, dpool, idx, x, save_mean, save_invstd, gamma, beta,
                       ws, C, H, W, OH, OW, CH, chunk);
    hipLaunchKernelGGL(bn_pool_bwd_fold_kernel, dim3(C), dim3(64), 0, st, ws, k12, dgamma, dbeta, N * CH, (double)N * H * W,
                       acc_param_grads);
    const int RB = OH >= 128 ? 32 : 8, bands = jp_cdiv(OH, RB);
    const long units = (long)N * C * bands * (OW / 2);
    hipLaunchKernelGGL(bn_pool_bwd_apply_kernel, dim3((unsigned)jp_cdiv(units, (long)TPB)), dim3(TPB), 0, st, dpool, idx, x, save_mean,
                       save_invstd, gamma, beta, k12, dx, units, C, OH, OW, RB, bands);
    JP_LAUNCH_CHECK();
}

// ws: optional scratch of jp_channel_sum_ws_floats(N, C, HW) floats -- the per-workgroup partial sums are then folded in a fixed order
// (bit-reproducible); NULL: they meet in float atomics (run-dependent last bits)
extern "C" long jp_channel_sum_ws_floats(int N, int C, int HW) {
    int CH, chunk;
    chunking(N, C, HW, &CH, &chunk);
    return (long)C * N * CH;
}
extern "C" int jp_channel_sum(const float* x, float* out, int N, int C, int HW, int accumulate, float* ws, void* stream) {
    JP_CHECK_ARG(x && out && N > 0 && C > 0 && HW > 0, "channel_sum: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate) JP_HIP(hipMemsetAsync(out, 0, sizeof(float) * C, st));
    int CH, chunk;
    chunking(N, C, HW, &CH, &chunk);
    hipLaunchKernelGGL(channel_sum_kernel, dim3(C, N * CH), dim3(TPB), 0, st, x, out, C, HW, CH, chunk, ws);
    if (ws) hipLaunchKernelGGL(channel_fold_kernel, dim3(jp_cdiv(C, 64)), dim3(64), 0, st, ws, out, C, N * CH);
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
