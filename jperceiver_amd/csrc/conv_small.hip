// Direct (non-MFMA) 3x3 stride-1 convolution for heads with very few output channels: the sigmoid
// disparity heads Conv3x3(256 -> 1) (depth_decoder.py:36-39, fed by the fused nearest-2x upsample) and
// the BEV logits head Conv3x3(16 -> 2) (layout_model.py:154).  With Cout <= 4 an MFMA tile would be
// >= 94 % padding; here every thread owns one output pixel (forward) or a strip of pixels of one input
// channel (wgrad), filter taps are wave-uniform LDS broadcasts, and the kernels are bound by L1/L2
// streaming of the input (each input element is touched 9x from cache, once from HBM).
#include "jp_common.h"
#include <algorithm>

namespace {

constexpr int TPB = 256;
constexpr int MAXCO = 4;

struct SrcSeg {   // one channel segment of the (virtually concatenated / upsampled) input
    const float* p;
    int C, sh;    // channels, 1 = stored at half resolution
};
struct Src3s {
    SrcSeg s[3];
    int nseg, H, W;
};

// 9 gather offsets of output pixel (y, x) inside one channel plane of a segment with shift sh
__device__ __forceinline__ void tap_offsets(int y, int x, int H, int W, int sh, int reflect, int off[9]) {
    const int w = W >> sh;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        int iy = y - 1 + t / 3, ix = x - 1 + t % 3;
        if (reflect) {
            iy = jp_reflect(iy, H);
            ix = jp_reflect(ix, W);
            off[t] = (iy >> sh) * w + (ix >> sh);
        } else {
            off[t] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? (iy >> sh) * w + (ix >> sh) : -1;
        }
    }
}

// y[img][co][pix] = act(bias[co] + sum_{ci,t} w[co][ci][t] * x[ci][pix + t])
template <int CO>
__global__ __launch_bounds__(TPB) void conv_small_fwd_kernel(Src3s src, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             int Cin, int act, int reflect, int accumulate) {
    extern __shared__ float ws[];   // [CO][Cin*9]
    for (int i = threadIdx.x; i < CO * Cin * 9; i += TPB) ws[i] = w[i];
    __syncthreads();
    const int img = blockIdx.y;
    const int HW = src.H * src.W;
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= HW) return;
    const int yy = p / src.W, xx = p - yy * src.W;
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = bias ? bias[c] : 0.f;
    int cbase = 0;
#pragma unroll
    for (int sg = 0; sg < 3; ++sg) {   // fully unrolled: constant indices into the kernel-argument struct
        if (sg >= src.nseg) break;
        const SrcSeg seg = src.s[sg];
        int off[9];
        tap_offsets(yy, xx, src.H, src.W, seg.sh, reflect, off);
        const int plane = (src.H >> seg.sh) * (src.W >> seg.sh);
        const float* xp = seg.p + (size_t)img * seg.C * plane;
        for (int ci = 0; ci < seg.C; ++ci) {
            const float* q = xp + (size_t)ci * plane;
            const float* wc = ws + (cbase + ci) * 9;
            float v[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) v[t] = off[t] >= 0 ? q[off[t]] : 0.f;
#pragma unroll
            for (int c = 0; c < CO; ++c)
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[c] = fmaf(wc[c * Cin * 9 + t], v[t], acc[c]);
        }
        cbase += seg.C;
    }
#pragma unroll
    for (int c = 0; c < CO; ++c) {
        float* q = y + ((size_t)img * CO + c) * HW + p;
        const float v = jp_act(acc[c], act);
        *q = accumulate ? *q + v : v;
    }
}

// dw[co][ci][t] += sum_{img, pix} dy[img][co][pix] * x[img][ci][pix + t];  grid (pixel chunks, Cin, batch)
template <int CO>
__global__ __launch_bounds__(TPB) void conv_small_wgrad_kernel(Src3s src, const float* __restrict__ dy,
                                                               float* __restrict__ dw, int Cin, int chunk,
                                                               int reflect) {
    __shared__ float red[4][CO * 9];
    const int ci = blockIdx.y, img = blockIdx.z;
    const int HW = src.H * src.W;
    // segment of this input channel (block-uniform)
    const int e0 = src.s[0].C, e1 = e0 + (src.nseg > 1 ? src.s[1].C : 0);
    const int c0 = ci < e0 ? 0 : (ci < e1 ? e0 : e1);
    const SrcSeg seg = ci < e0 ? src.s[0] : (ci < e1 ? src.s[1] : src.s[2]);
    const int plane = (src.H >> seg.sh) * (src.W >> seg.sh);
    const float* q = seg.p + ((size_t)img * seg.C + (ci - c0)) * plane;
    float acc[CO * 9];
#pragma unroll
    for (int i = 0; i < CO * 9; ++i) acc[i] = 0.f;
    const int beg = blockIdx.x * chunk, end = min(HW, beg + chunk);
    for (int p = beg + threadIdx.x; p < end; p += TPB) {
        const int yy = p / src.W, xx = p - yy * src.W;
        int off[9];
        tap_offsets(yy, xx, src.H, src.W, seg.sh, reflect, off);
        float v[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) v[t] = off[t] >= 0 ? q[off[t]] : 0.f;
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            const float g = dy[((size_t)img * CO + c) * HW + p];
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[c * 9 + t] = fmaf(g, v[t], acc[c * 9 + t]);
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < CO * 9; ++i) {
        const float s = jp_wave_sum(acc[i]);
        if (lane == 0) red[wv][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < CO * 9) {
        const int i = threadIdx.x;
        const float s = red[0][i] + red[1][i] + red[2][i] + red[3][i];
        const int c = i / 9, t = i - c * 9;
        atomicAdd(dw + ((size_t)c * Cin + ci) * 9 + t, s);
    }
}

// Single full-resolution source: LDS-tiled wgrad.  grid (64-row bands, Cin, batch); a workgroup walks the 16x64 tiles
// of its band, stages tile + halo once (padding resolved while staging) and every thread owns 4 vertically adjacent
// pixels, so a staged row is read once from LDS for all the taps that use it; one reduction per workgroup at the end.
template <int CO>
__global__ __launch_bounds__(TPB) void conv_small_wgrad_tiled_kernel(const float* __restrict__ x,
                                                                     const float* __restrict__ dy,
                                                                     float* __restrict__ dw, int Cin, int H, int W,
                                                                     int reflect) {
    constexpr int TW = 64, TH = 16, PW = TW + 2, PH = TH + 2, BAND = 64;
    __shared__ float tile[PH * PW];
    __shared__ float red[4][CO * 9];
    const int ci = blockIdx.y, img = blockIdx.z;
    const int HW = H * W;
    const float* xp = x + ((size_t)img * Cin + ci) * HW;
    const float* dp = dy + (size_t)img * CO * HW;
    const int tx = threadIdx.x & 63, q = threadIdx.x >> 6;
    float acc[CO * 9];
#pragma unroll
    for (int i = 0; i < CO * 9; ++i) acc[i] = 0.f;
    const int yend = min(H, (int)(blockIdx.x + 1) * BAND);
    for (int ty0 = blockIdx.x * BAND; ty0 < yend; ty0 += TH) {
        for (int tx0 = 0; tx0 < W; tx0 += TW) {
            __syncthreads();
            for (int i = threadIdx.x; i < PH * PW; i += TPB) {
                const int ly = i / PW, lx = i - ly * PW;
                int iy = ty0 - 1 + ly, ix = tx0 - 1 + lx;
                float v = 0.f;
                if (reflect) {
                    iy = jp_reflect(min(iy, H), H);
                    ix = jp_reflect(min(ix, W), W);
                    v = xp[iy * W + ix];
                } else if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    v = xp[iy * W + ix];
                }
                tile[i] = v;
            }
            __syncthreads();
            const int xo = tx0 + tx;
            float g[4][CO];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int yo = ty0 + q * 4 + j;
                const bool ok = yo < H && xo < W;
#pragma unroll
                for (int c = 0; c < CO; ++c) g[j][c] = ok ? dp[(size_t)c * HW + yo * W + xo] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const float* row = tile + (q * 4 + r) * PW + tx;
                const float v0 = row[0], v1 = row[1], v2 = row[2];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ky = r - j;
                    if (ky < 0 || ky > 2) continue;
#pragma unroll
                    for (int c = 0; c < CO; ++c) {
                        acc[c * 9 + ky * 3 + 0] = fmaf(g[j][c], v0, acc[c * 9 + ky * 3 + 0]);
                        acc[c * 9 + ky * 3 + 1] = fmaf(g[j][c], v1, acc[c * 9 + ky * 3 + 1]);
                        acc[c * 9 + ky * 3 + 2] = fmaf(g[j][c], v2, acc[c * 9 + ky * 3 + 2]);
                    }
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < CO * 9; ++i) {
        const float s = jp_wave_sum(acc[i]);
        if (lane == 0) red[wv][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < CO * 9) {
        const int i = threadIdx.x;
        const float s = red[0][i] + red[1][i] + red[2][i] + red[3][i];
        const int c = i / 9, t = i - c * 9;
        atomicAdd(dw + ((size_t)c * Cin + ci) * 9 + t, s);
    }
}

Src3s make_src(const float* x0, int c0, int up0, const float* x1, int c1, int up1, const float* x2, int c2, int up2,
               int H, int W) {
    Src3s s;
    s.nseg = 0;
    s.H = H; s.W = W;
    const float* ps[3] = {x0, x1, x2};
    const int cs[3] = {c0, c1, c2}, us[3] = {up0, up1, up2};
    for (int i = 0; i < 3; ++i)
        if (cs[i] > 0) { s.s[s.nseg].p = ps[i]; s.s[s.nseg].C = cs[i]; s.s[s.nseg].sh = us[i]; ++s.nseg; }
    for (int i = s.nseg; i < 3; ++i) s.s[i] = s.s[0];
    return s;
}

}  // namespace

// internal entry points used by conv.hip's dispatcher (not part of the public ABI)
int jp_conv_small_fwd(const float* x0, int c0, int up0, const float* x1, int c1, int up1, const float* x2, int c2, int up2,
                      const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int act, int reflect,
                      hipStream_t st, int accumulate) {
    const int Cin = c0 + c1 + c2;
    const Src3s src = make_src(x0, c0, up0, x1, c1, up1, x2, c2, up2, H, W);
    const dim3 grid(jp_cdiv(H * W, TPB), N);
    const size_t lds = sizeof(float) * (size_t)Cout * Cin * 9;
#define JP_GO(CO) hipLaunchKernelGGL((conv_small_fwd_kernel<CO>), grid, dim3(TPB), lds, st, src, w, bias, y, Cin, act, reflect, accumulate)
    switch (Cout) {
        case 1: JP_GO(1); break;
        case 2: JP_GO(2); break;
        case 3: JP_GO(3); break;
        default: JP_GO(4); break;
    }
#undef JP_GO
    return 0;
}

int jp_conv_small_wgrad(const float* x0, int c0, int up0, const float* x1, int c1, int up1, const float* x2, int c2,
                        int up2, const float* dy, float* dw, int N, int H, int W, int Cout, int reflect,
                        hipStream_t st) {
    const int Cin = c0 + c1 + c2;
    const Src3s src = make_src(x0, c0, up0, x1, c1, up1, x2, c2, up2, H, W);
    if (src.nseg == 1 && src.s[0].sh == 0 && H >= 2 && W >= 2) {   // single full-resolution source: LDS-tiled kernel
        const dim3 gt(jp_cdiv(H, 64), Cin, N);
#define JP_GT(CO) hipLaunchKernelGGL((conv_small_wgrad_tiled_kernel<CO>), gt, dim3(TPB), 0, st, src.s[0].p, dy, dw, Cin, H, W, reflect)
        switch (Cout) {
            case 1: JP_GT(1); break;
            case 2: JP_GT(2); break;
            case 3: JP_GT(3); break;
            default: JP_GT(4); break;
        }
#undef JP_GT
        return 0;
    }
    // chunks so that ~2k blocks stream the input, >= 4k pixels each
    int chunks = std::max(1, 2048 / std::max(1, Cin * N));
    chunks = std::min(chunks, std::max(1, H * W / 4096));
    const int chunk = jp_cdiv(jp_cdiv(H * W, chunks), TPB) * TPB;
    const dim3 grid(jp_cdiv(H * W, chunk), Cin, N);
#define JP_GO(CO) hipLaunchKernelGGL((conv_small_wgrad_kernel<CO>), grid, dim3(TPB), 0, st, src, dy, dw, Cin, chunk, reflect)
    switch (Cout) {
        case 1: JP_GO(1); break;
        case 2: JP_GO(2); break;
        case 3: JP_GO(3); break;
        default: JP_GO(4); break;
    }
#undef JP_GO
    return 0;
}
