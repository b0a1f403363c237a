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
                                                               int reflect, float* __restrict__ part) {
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
        // part: [workgroup (img, x-block)][ci][CO * 9] partial sums, folded in a fixed order by conv_small_wgrad_fold_kernel
        if (part) part[(((size_t)blockIdx.z * gridDim.x + blockIdx.x) * Cin + ci) * (CO * 9) + i] = s;
        else atomicAdd(dw + ((size_t)c * Cin + ci) * 9 + t, s);
    }
}

// Single full-resolution source: LDS-tiled wgrad.  grid (64-row bands, Cin, batch); a workgroup walks the 16x64 tiles
// of its band, stages tile + halo once (padding resolved while staging) and every thread owns 4 vertically adjacent
// pixels, so a staged row is read once from LDS for all the taps that use it; one reduction per workgroup at the end.
template <int CO>
__global__ __launch_bounds__(TPB) void conv_small_wgrad_tiled_kernel(const float* __restrict__ x,
                                                                     const float* __restrict__ dy,
                                                                     float* __restrict__ dw, int Cin, int H, int W,
                                                                     int reflect, float* __restrict__ part) {
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
        // part: [workgroup (img, x-block)][ci][CO * 9] partial sums, folded in a fixed order by conv_small_wgrad_fold_kernel
        if (part) part[(((size_t)blockIdx.z * gridDim.x + blockIdx.x) * Cin + ci) * (CO * 9) + i] = s;
        else atomicAdd(dw + ((size_t)c * Cin + ci) * 9 + t, s);
    }
}

// dw[c][ci][t] += sum over the workgroups (in index order) of part[wg][ci][c * 9 + t]
__global__ void conv_small_wgrad_fold_kernel(const float* __restrict__ part, float* __restrict__ dw, int Cin, int CO, int nwg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cin * CO * 9) return;
    const int ci = i / (CO * 9), r = i - ci * (CO * 9), c = r / 9, t = r - c * 9;
    float s = 0.f;
    for (int b = 0; b < nwg; ++b) s += part[((size_t)b * Cin + ci) * (CO * 9) + r];
    dw[((size_t)c * Cin + ci) * 9 + t] += s;
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

// ------------------------------------------------------------------------------------------------------------------
// Disparity heads Conv3x3(up2x(x)) -> 1 channel (depth_decoder.py:36-39), upsample-aware: an output pixel (2i+a, 2j+b)
// sees a 2x2 patch of the half-resolution x (reflection padding of up(x) == edge clamp on x), so all three kernels
// work on the half-resolution grid with the 16 pre-summed (class, slot) weights
//   W'[(a,b),(r,s)][c] = sum_{dy in Dy(a,r), dx in Dx(b,s)} w[c][dy][dx],  Dy(0,.) = {0},{1,2};  Dy(1,.) = {0,1},{2}.
__device__ __forceinline__ void up_tap_range(int a, int r, int& lo, int& hi) {
    lo = a ? (r ? 2 : 0) : (r ? 1 : 0);
    hi = a ? (r ? 2 : 1) : (r ? 2 : 0);
}
// wq[q][c], q = a*8 + b*4 + r*2 + s
__device__ __forceinline__ void up_build_weights(const float* __restrict__ w, float* wq, int C) {
    for (int i = threadIdx.x; i < 16 * C; i += blockDim.x) {
        const int c = i % C, q = i / C;
        int y0, y1, x0, x1;
        up_tap_range(q >> 3, (q >> 1) & 1, y0, y1);
        up_tap_range((q >> 2) & 1, q & 1, x0, x1);
        float v = 0.f;
        for (int dy = y0; dy <= y1; ++dy)
            for (int dx = x0; dx <= x1; ++dx) v += w[c * 9 + dy * 3 + dx];
        wq[i] = v;
    }
}
// D[q] = sum of the dY entries that reach x(i, j) through (class, slot) q: the regular one (i', j') = (i+1-a-r, j+1-b-s)
// plus the clamp-folded ones on the 4 boundary lines.  dyp = one (2h x 2w) plane.
__device__ __forceinline__ void up_gather_D(const float* __restrict__ dyp, int i, int j, int h, int w, float D[16]) {
    const int W = 2 * w;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int a = q >> 3, b = (q >> 2) & 1, r = (q >> 1) & 1, s = q & 1;
        const int ri = i + 1 - a - r, rj = j + 1 - b - s;
        const int ei = (i == 0 && a == 0 && r == 0) ? 0 : ((i == h - 1 && a == 1 && r == 1) ? h - 1 : -1);
        const int ej = (j == 0 && b == 0 && s == 0) ? 0 : ((j == w - 1 && b == 1 && s == 1) ? w - 1 : -1);
        const bool rv = (unsigned)ri < (unsigned)h, cv = (unsigned)rj < (unsigned)w;
        float v = 0.f;
        if (rv && cv) v += dyp[(2 * ri + a) * W + 2 * rj + b];
        if (ei >= 0 && cv) v += dyp[(2 * ei + a) * W + 2 * rj + b];
        if (rv && ej >= 0) v += dyp[(2 * ri + a) * W + 2 * ej + b];
        if (ei >= 0 && ej >= 0) v += dyp[(2 * ei + a) * W + 2 * ej + b];
        D[q] = v;
    }
}

// forward: one thread per half-resolution pixel -> its 2x2 output block
__global__ __launch_bounds__(TPB) void up_head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int C,
                                                          int h, int wd, int act) {
    extern __shared__ float wq[];   // [16][C]
    up_build_weights(w, wq, C);
    __syncthreads();
    const int img = blockIdx.y, hw = h * wd;
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= hw) return;
    const int i = p / wd, j = p - i * wd;
    const int r0 = max(i - 1, 0) * wd, r1 = i * wd, r2 = min(i + 1, h - 1) * wd;
    const int c0 = max(j - 1, 0), c2 = min(j + 1, wd - 1);
    float o00 = 0.f, o01 = 0.f, o10 = 0.f, o11 = 0.f;
    const float* xp = x + (size_t)img * C * hw;
    for (int c = 0; c < C; ++c) {
        const float* q = xp + (size_t)c * hw;
        const float v00 = q[r0 + c0], v01 = q[r0 + j], v02 = q[r0 + c2];
        const float v10 = q[r1 + c0], v11 = q[r1 + j], v12 = q[r1 + c2];
        const float v20 = q[r2 + c0], v21 = q[r2 + j], v22 = q[r2 + c2];
        const float* wc = wq + c;
        // class (a,b) reads v[a+r][b+s]
        o00 += wc[0 * C] * v00 + wc[1 * C] * v01 + wc[2 * C] * v10 + wc[3 * C] * v11;
        o01 += wc[4 * C] * v01 + wc[5 * C] * v02 + wc[6 * C] * v11 + wc[7 * C] * v12;
        o10 += wc[8 * C] * v10 + wc[9 * C] * v11 + wc[10 * C] * v20 + wc[11 * C] * v21;
        o11 += wc[12 * C] * v11 + wc[13 * C] * v12 + wc[14 * C] * v21 + wc[15 * C] * v22;
    }
    const float bv = bias ? bias[0] : 0.f;
    float* yo = y + (size_t)img * 4 * hw + (size_t)(2 * i) * (2 * wd) + 2 * j;
    *reinterpret_cast<float2*>(yo) = make_float2(jp_act(o00 + bv, act), jp_act(o01 + bv, act));
    *reinterpret_cast<float2*>(yo + 2 * wd) = make_float2(jp_act(o10 + bv, act), jp_act(o11 + bv, act));
}

// forward, two horizontally adjacent half-resolution pixels (j, j+1; j even) per thread: a 3 x 4 window per channel (three
// aligned float2 + two scalars per row pair instead of 2 x 9 scalars), the 16 slot weights of a channel as four broadcast
// 16-byte LDS reads shared by both pixels.  wq[c][16] here.
__global__ __launch_bounds__(TPB) void up_head_fwd2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int C,
                                                           int h, int wd, int act) {
    extern __shared__ float wq[];   // [16][C] built, then read transposed below
    up_build_weights(w, wq, C);
    __syncthreads();
    const int img = blockIdx.y, hw = h * wd, w2 = wd >> 1;
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= h * w2) return;
    const int i = p / w2, j = 2 * (p - i * w2);
    const int r0 = max(i - 1, 0) * wd, r1 = i * wd, r2 = min(i + 1, h - 1) * wd;
    const int cl = max(j - 1, 0), cr = min(j + 2, wd - 1);
    float o[2][4];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) o[k][q] = 0.f;
    const float* xp = x + (size_t)img * C * hw;
    // (a thread's channel loop is a chain of dependent-latency loads: four channels in flight)
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
        const float* q = xp + (size_t)c * hw;
        float v[3][4];
        const int rr[3] = {r0, r1, r2};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float2 m = *reinterpret_cast<const float2*>(q + rr[r] + j);
            v[r][0] = q[rr[r] + cl]; v[r][1] = m.x; v[r][2] = m.y; v[r][3] = q[rr[r] + cr];
        }
        float wc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) wc[t] = wq[t * C + c];
        // pixel k reads columns k .. k+2 of the window; class (a, b), slot (r, s) reads v[a + r][k + b + s]
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            o[k][0] += wc[0] * v[0][k] + wc[1] * v[0][k + 1] + wc[2] * v[1][k] + wc[3] * v[1][k + 1];
            o[k][1] += wc[4] * v[0][k + 1] + wc[5] * v[0][k + 2] + wc[6] * v[1][k + 1] + wc[7] * v[1][k + 2];
            o[k][2] += wc[8] * v[1][k] + wc[9] * v[1][k + 1] + wc[10] * v[2][k] + wc[11] * v[2][k + 1];
            o[k][3] += wc[12] * v[1][k + 1] + wc[13] * v[1][k + 2] + wc[14] * v[2][k + 1] + wc[15] * v[2][k + 2];
        }
    }
    const float bv = bias ? bias[0] : 0.f;
    float* yo = y + (size_t)img * 4 * hw + (size_t)(2 * i) * (2 * wd) + 2 * j;
    *reinterpret_cast<float4*>(yo) = make_float4(jp_act(o[0][0] + bv, act), jp_act(o[0][1] + bv, act), jp_act(o[1][0] + bv, act),
                                                 jp_act(o[1][1] + bv, act));
    *reinterpret_cast<float4*>(yo + 2 * wd) = make_float4(jp_act(o[0][2] + bv, act), jp_act(o[0][3] + bv, act),
                                                          jp_act(o[1][2] + bv, act), jp_act(o[1][3] + bv, act));
}

// the same with the channel loop split over the workgroup's 4 waves (64 pixel pairs per workgroup): the small maps
template <int groups>
__global__ __launch_bounds__(TPB) void up_head_fwd2s_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int C,
                                                           int h, int wd, int act) {
    extern __shared__ float wq[];   // [16][C]
    __shared__ float red[3][64][8];
    up_build_weights(w, wq, C);
    __syncthreads();
    // 64 pixel pairs per workgroup; wave cg takes a quarter of the channels (a thread's channel loop is a chain of dependent-latency
    // loads: four chains of C/4 instead of one of C, four times the waves on the small maps), partial sums meet in LDS
    const int img = blockIdx.y, hw = h * wd, w2 = wd >> 1;
    // (groups = 1 on the large maps: 256 pairs per workgroup, one chain -- there the kernel is throughput-bound and four times the
    // workgroups only rebuild the slot weights four times as often: 0.39 vs 0.50 ms at 8 x 256 x 256^2)
    constexpr int ppb = groups == 4 ? 64 : TPB;
    const int cg = threadIdx.x / ppb;
    const int p = blockIdx.x * ppb + (threadIdx.x % ppb);
    const bool live = p < h * w2;
    const int pc = live ? p : 0;
    const int i = pc / w2, j = 2 * (pc - i * w2);
    const int r0 = max(i - 1, 0) * wd, r1 = i * wd, r2 = min(i + 1, h - 1) * wd;
    const int cl = max(j - 1, 0), cr = min(j + 2, wd - 1);
    float o[2][4];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) o[k][q] = 0.f;
    const float* xp = x + (size_t)img * C * hw;
    const int cq = (C + groups - 1) / groups, cbeg = cg * cq, cend = min(C, cbeg + cq);
#pragma unroll 4
    for (int c = cbeg; c < cend; ++c) {
        const float* q = xp + (size_t)c * hw;
        float v[3][4];
        const int rr[3] = {r0, r1, r2};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float2 m = *reinterpret_cast<const float2*>(q + rr[r] + j);
            v[r][0] = q[rr[r] + cl]; v[r][1] = m.x; v[r][2] = m.y; v[r][3] = q[rr[r] + cr];
        }
        float wc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) wc[t] = wq[t * C + c];
        // pixel k reads columns k .. k+2 of the window; class (a, b), slot (r, s) reads v[a + r][k + b + s]
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            o[k][0] += wc[0] * v[0][k] + wc[1] * v[0][k + 1] + wc[2] * v[1][k] + wc[3] * v[1][k + 1];
            o[k][1] += wc[4] * v[0][k + 1] + wc[5] * v[0][k + 2] + wc[6] * v[1][k + 1] + wc[7] * v[1][k + 2];
            o[k][2] += wc[8] * v[1][k] + wc[9] * v[1][k + 1] + wc[10] * v[2][k] + wc[11] * v[2][k + 1];
            o[k][3] += wc[12] * v[1][k + 1] + wc[13] * v[1][k + 2] + wc[14] * v[2][k + 1] + wc[15] * v[2][k + 2];
        }
    }
    const int lp = threadIdx.x & 63;
    if (groups == 1) {
        if (!live) return;
    } else {
    if (cg > 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) red[cg - 1][lp][k] = o[k >> 2][k & 3];
    }
    __syncthreads();
    if (cg > 0 || !live) return;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k >> 2][k & 3] += red[g][lp][k];
    }
    const float bv = bias ? bias[0] : 0.f;
    float* yo = y + (size_t)img * 4 * hw + (size_t)(2 * i) * (2 * wd) + 2 * j;
    *reinterpret_cast<float4*>(yo) = make_float4(jp_act(o[0][0] + bv, act), jp_act(o[0][1] + bv, act), jp_act(o[1][0] + bv, act),
                                                 jp_act(o[1][1] + bv, act));
    *reinterpret_cast<float4*>(yo + 2 * wd) = make_float4(jp_act(o[0][2] + bv, act), jp_act(o[0][3] + bv, act),
                                                          jp_act(o[1][2] + bv, act), jp_act(o[1][3] + bv, act));
}

// dgrad: dx[c][i][j] (= | +=) sum_q W'[q][c] * D_q(i, j), straight at half resolution
__global__ __launch_bounds__(TPB) void up_head_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                            float* __restrict__ dx, int C, int h, int wd,
                                                            int accumulate) {
    extern __shared__ float wq[];
    up_build_weights(w, wq, C);
    __syncthreads();
    const int img = blockIdx.y, hw = h * wd;
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= hw) return;
    float D[16];
    up_gather_D(dy + (size_t)img * 4 * hw, p / wd, p % wd, h, wd, D);
    float* dp = dx + (size_t)img * C * hw + p;
    // four channels per iteration, the read-modify-write loads issued together (the compiler cannot move a load above the previous
    // channel's store on its own: the 256-iteration chain of dependent-latency accesses was the kernel's whole run time)
    int c = 0;
    for (; c + 4 <= C; c += 4) {
        float old[4], v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) old[k] = accumulate ? dp[(size_t)(c + k) * hw] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) v[k] = fmaf(wq[q * C + c + k], D[q], v[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) dp[(size_t)(c + k) * hw] = old[k] + v[k];
    }
    for (; c < C; ++c) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v = fmaf(wq[q * C + c], D[q], v);
        float* o = dp + (size_t)c * hw;
        *o = accumulate ? *o + v : v;
    }
}

// wgrad: dW'[q][c] = sum_{img, i, j} D_q(i, j) * x[c][i][j], folded into dw[c][tap] on the way out.
// grid (C/8, pixel bands, batch); 8 channels x 16 (class, slot) accumulators per thread.
constexpr int UP_CB = 8;
__global__ __launch_bounds__(TPB) void up_head_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ dw, int C, int h, int wd, int band) {
    __shared__ float red[4][UP_CB * 16];
    const int c0 = blockIdx.x * UP_CB, img = blockIdx.z, hw = h * wd;
    const float* xp = x + ((size_t)img * C + c0) * hw;
    const float* dyp = dy + (size_t)img * 4 * hw;
    float acc[UP_CB * 16];
#pragma unroll
    for (int i = 0; i < UP_CB * 16; ++i) acc[i] = 0.f;
    const int pend = min(hw, (int)(blockIdx.y + 1) * band);
    for (int p = blockIdx.y * band + threadIdx.x; p < pend; p += TPB) {
        float D[16];
        up_gather_D(dyp, p / wd, p % wd, h, wd, D);
#pragma unroll
        for (int k = 0; k < UP_CB; ++k) {
            const float xv = (c0 + k < C) ? xp[(size_t)k * hw + p] : 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[k * 16 + q] = fmaf(xv, D[q], acc[k * 16 + q]);
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < UP_CB * 16; ++i) {
        const float s = jp_wave_sum(acc[i]);
        if (lane == 0) red[wv][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < UP_CB * 16) {
        const int k = threadIdx.x >> 4, q = threadIdx.x & 15;
        if (c0 + k < C) {
            const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
            int y0, y1, x0, x1;
            up_tap_range(q >> 3, (q >> 1) & 1, y0, y1);
            up_tap_range((q >> 2) & 1, q & 1, x0, x1);
            for (int ty = y0; ty <= y1; ++ty)
                for (int tx = x0; tx <= x1; ++tx) atomicAdd(dw + (size_t)(c0 + k) * 9 + ty * 3 + tx, s);
        }
    }
}

// wgrad with the 16 gathered dY sums D_q(i, j) PRECOMPUTED (up_head_D_kernel -> D[img][q][h*w]): the gather (16 values x up to 4
// loads + boundary logic) is done once per pixel instead of once per pixel and 8-channel block (C/8 = 32 times).
__global__ __launch_bounds__(TPB) void up_head_D_kernel(const float* __restrict__ dy, float* __restrict__ D, int h, int wd) {
    const int img = blockIdx.y, hw = h * wd;
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= hw) return;
    float d[16];
    up_gather_D(dy + (size_t)img * 4 * hw, p / wd, p % wd, h, wd, d);
#pragma unroll
    for (int q = 0; q < 16; ++q) D[((size_t)img * 16 + q) * hw + p] = d[q];
}
constexpr int UPD_CB = 8;
// part != nullptr: the workgroup's 16 (class, slot) sums per channel go to part[(img * bands + band)][c][16] and
// up_head_wgrad_fold_kernel adds them into the 9 taps in a fixed order (bit-reproducible); nullptr: float atomics on dw
__global__ __launch_bounds__(TPB) void up_head_wgrad_D_kernel(const float* __restrict__ x, const float* __restrict__ D,
                                                              float* __restrict__ dw, int C, int hw, int band,
                                                              float* __restrict__ part) {
    __shared__ float red[4][UPD_CB * 16];
    const int c0 = blockIdx.x * UPD_CB, img = blockIdx.z;
    const float* xp = x + ((size_t)img * C + c0) * hw;
    const float* dp = D + (size_t)img * 16 * hw;
    float acc[UPD_CB * 16];
#pragma unroll
    for (int i = 0; i < UPD_CB * 16; ++i) acc[i] = 0.f;
    const int pend = min(hw, (int)(blockIdx.y + 1) * band);
    for (int p = blockIdx.y * band + threadIdx.x; p < pend; p += TPB) {
        float d[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) d[q] = dp[(size_t)q * hw + p];
#pragma unroll
        for (int k = 0; k < UPD_CB; ++k) {
            const float xv = (c0 + k < C) ? xp[(size_t)k * hw + p] : 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[k * 16 + q] = fmaf(xv, d[q], acc[k * 16 + q]);
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < UPD_CB * 16; ++i) {
        const float s = jp_wave_sum(acc[i]);
        if (lane == 0) red[wv][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < UPD_CB * 16) {
        const int k = threadIdx.x >> 4, q = threadIdx.x & 15;
        if (c0 + k < C) {
            const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
            if (part) {
                part[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * C + c0 + k) * 16 + q] = s;
            } else {
                int y0, y1, x0, x1;
                up_tap_range(q >> 3, (q >> 1) & 1, y0, y1);
                up_tap_range((q >> 2) & 1, q & 1, x0, x1);
                for (int ty = y0; ty <= y1; ++ty)
                    for (int tx = x0; tx <= x1; ++tx) atomicAdd(dw + (size_t)(c0 + k) * 9 + ty * 3 + tx, s);
            }
        }
    }
}
// dw[c][tap] += sum over the (class, slot) sums q that reach the tap, over the workgroups in index order
__global__ void up_head_wgrad_fold_kernel(const float* __restrict__ part, float* __restrict__ dw, int C, int nblk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * 9) return;
    const int c = i / 9, t = i - c * 9, ty = t / 3, tx = t - ty * 3;
    float s = 0.f;
    for (int q = 0; q < 16; ++q) {
        int y0, y1, x0, x1;
        up_tap_range(q >> 3, (q >> 1) & 1, y0, y1);
        up_tap_range((q >> 2) & 1, q & 1, x0, x1);
        if (ty < y0 || ty > y1 || tx < x0 || tx > x1) continue;
        for (int b = 0; b < nblk; ++b) s += part[((size_t)b * C + c) * 16 + q];
    }
    dw[i] += s;
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

static void small_wgrad_grid(int Cin, int N, int H, int W, bool tiled, dim3* grid, int* chunk) {
    if (tiled) { *grid = dim3(jp_cdiv(H, 64), Cin, N); *chunk = 0; return; }
    // chunks so that ~2k blocks stream the input, >= 4k pixels each
    int chunks = std::max(1, 2048 / std::max(1, Cin * N));
    chunks = std::min(chunks, std::max(1, H * W / 4096));
    *chunk = jp_cdiv(jp_cdiv(H * W, chunks), TPB) * TPB;
    *grid = dim3(jp_cdiv(H * W, *chunk), Cin, N);
}
// scratch with which the workgroups' partial sums are folded in a fixed order (bit-reproducible) instead of meeting in float atomics
long jp_conv_small_wgrad_ws_floats(int N, int Cin, int H, int W, int Cout, int single_full_res) {
    dim3 g;
    int chunk;
    small_wgrad_grid(Cin, N, H, W, single_full_res && H >= 2 && W >= 2, &g, &chunk);
    return (long)g.x * g.z * Cin * std::min(std::max(Cout, 1), 4) * 9;
}
int jp_conv_small_wgrad(const float* x0, int c0, int up0, const float* x1, int c1, int up1, const float* x2, int c2,
                        int up2, const float* dy, float* dw, int N, int H, int W, int Cout, int reflect,
                        hipStream_t st, float* ws, long ws_floats) {
    const int Cin = c0 + c1 + c2;
    const Src3s src = make_src(x0, c0, up0, x1, c1, up1, x2, c2, up2, H, W);
    const bool tiled = src.nseg == 1 && src.s[0].sh == 0 && H >= 2 && W >= 2;   // single full-resolution source: LDS-tiled kernel
    const int CO = Cout <= 3 ? std::max(Cout, 1) : 4;
    dim3 grid;
    int chunk;
    small_wgrad_grid(Cin, N, H, W, tiled, &grid, &chunk);
    float* part = (ws && ws_floats >= (long)grid.x * grid.z * Cin * CO * 9) ? ws : nullptr;
    if (tiled) {
#define JP_GT(COv) hipLaunchKernelGGL((conv_small_wgrad_tiled_kernel<COv>), grid, dim3(TPB), 0, st, src.s[0].p, dy, dw, Cin, H, W, reflect, part)
        switch (Cout) {
            case 1: JP_GT(1); break;
            case 2: JP_GT(2); break;
            case 3: JP_GT(3); break;
            default: JP_GT(4); break;
        }
#undef JP_GT
    } else {
#define JP_GO(COv) hipLaunchKernelGGL((conv_small_wgrad_kernel<COv>), grid, dim3(TPB), 0, st, src, dy, dw, Cin, chunk, reflect, part)
        switch (Cout) {
            case 1: JP_GO(1); break;
            case 2: JP_GO(2); break;
            case 3: JP_GO(3); break;
            default: JP_GO(4); break;
        }
#undef JP_GO
    }
    if (part) hipLaunchKernelGGL(conv_small_wgrad_fold_kernel, dim3(jp_cdiv(Cin * CO * 9, 256)), dim3(256), 0, st, part, dw, Cin, CO,
                                 (int)(grid.x * grid.z));
    return 0;
}

// upsample-aware disparity head (Cout = 1, one source read through the nearest-2x upsample, reflection padding);
// x: (N, C, h, w), y / dy: (N, 1, 2h, 2w)
int jp_up_head_fwd(const float* x, const float* w, const float* bias, float* y, int N, int C, int h, int wd, int act,
                   hipStream_t st) {
    if (wd % 4 == 0) {       // two pixels per thread (aligned float2 / float4 accesses)
        if ((long)N * h * (wd / 2) <= 8L * 2048)      // small maps: a thread's 256-channel chain of dependent-latency loads split over 4 waves
            hipLaunchKernelGGL(up_head_fwd2s_kernel<4>, dim3(jp_cdiv(h * (wd / 2), 64), N), dim3(TPB), sizeof(float) * 16 * C, st, x, w,
                               bias, y, C, h, wd, act);
        else
            hipLaunchKernelGGL(up_head_fwd2_kernel, dim3(jp_cdiv(h * (wd / 2), TPB), N), dim3(TPB), sizeof(float) * 16 * C, st, x, w,
                               bias, y, C, h, wd, act);
        return 0;
    }
    hipLaunchKernelGGL(up_head_fwd_kernel, dim3(jp_cdiv(h * wd, TPB), N), dim3(TPB), sizeof(float) * 16 * C, st, x, w, bias, y,
                       C, h, wd, act);
    return 0;
}
int jp_up_head_dgrad(const float* dy, const float* w, float* dx, int N, int C, int h, int wd, int accumulate,
                     hipStream_t st) {
    hipLaunchKernelGGL(up_head_dgrad_kernel, dim3(jp_cdiv(h * wd, TPB), N), dim3(TPB), sizeof(float) * 16 * C, st, dy, w, dx, C,
                       h, wd, accumulate);
    return 0;
}
long jp_up_head_wgrad_ws_floats(int N, int C, int h, int wd) {
    const int hw = h * wd;
    const int bands = std::max(1, std::min(hw / (TPB * 16), 16));
    const int band = jp_cdiv(jp_cdiv(hw, bands), TPB) * TPB;
    return 16L * N * hw + 16L * C * jp_cdiv(hw, band) * N;
}
int jp_up_head_wgrad(const float* x, const float* dy, float* dw, int N, int C, int h, int wd, hipStream_t st, float* ws,
                     long ws_floats) {
    const int hw = h * wd;
    if (ws && ws_floats >= 16L * N * hw) {       // caller scratch for the gathered dY sums: gather once, stream afterwards
        hipLaunchKernelGGL(up_head_D_kernel, dim3(jp_cdiv(hw, TPB), N), dim3(TPB), 0, st, dy, ws, h, wd);
        const int bands = std::max(1, std::min(hw / (TPB * 16), 16));
        const int band = jp_cdiv(jp_cdiv(hw, bands), TPB) * TPB;
        const int nb = jp_cdiv(hw, band);
        // room behind the D planes for the workgroups' partial sums -> fixed-order fold (jp_up_head_wgrad_ws_floats asks for it)
        float* part = ws_floats >= 16L * N * hw + 16L * C * nb * N ? ws + 16L * N * hw : nullptr;
        hipLaunchKernelGGL(up_head_wgrad_D_kernel, dim3(jp_cdiv(C, UPD_CB), nb, N), dim3(TPB), 0, st, x, ws, dw, C, hw, band, part);
        if (part) hipLaunchKernelGGL(up_head_wgrad_fold_kernel, dim3(jp_cdiv(C * 9, 256)), dim3(256), 0, st, part, dw, C, nb * N);
        return 0;
    }
    const int bands = std::max(1, std::min(hw / (TPB * 16), 16));     // (64 bands of 4 iterations measured slower: 361 vs 278 us)
    const int band = jp_cdiv(jp_cdiv(hw, bands), TPB) * TPB;
    hipLaunchKernelGGL(up_head_wgrad_kernel, dim3(jp_cdiv(C, UP_CB), jp_cdiv(hw, band), N), dim3(TPB), 0, st, x, dy, dw, C, h, wd,
                       band);
    return 0;
}
