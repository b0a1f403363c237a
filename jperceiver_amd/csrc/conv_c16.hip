// Direct (VALU, exact fp32) 3x3 stride-1 zero-pad convolutions for the 16 / 32-channel layers of the BEV decoder at 128^2 / 256^2
// (layout_model.py:138-153: upconv(32 -> 32 on the 2x-upsampled map), conv(32 -> 16), upconv(16 -> 16)).  With 16 rows an MFMA
// tile of the implicit-GEMM engine is 75-87 % padding (measured: forward 196 us, dgrad 210 us, wgrad 515 us for 2.4 GFLOP at
// 8 x 16 x 256^2); here
//   forward / dgrad: a thread owns a 2x2 block of output pixels and ALL output channels (4*CO accumulators), per input channel it
//     loads the 4x4 window once and takes the 16 weights of a tap as four broadcast 16-byte LDS reads;  dgrad is the same kernel
//     on dY with the weights read transposed / flipped, and for an upsampled source the 2x2 block is summed into its
//     half-resolution pixel on the way out;
//   wgrad: a thread owns (co, ci) pairs with their 9 tap accumulators; a workgroup stages a (4 rows x 64 columns) tile of dY and
//     the haloed tile of X in LDS and slides a 3x3 register window along the rows (1 dY + 3 X LDS reads per 9 FMAs).
#include "jp_common.h"
#include <algorithm>

namespace {

constexpr int TPB = 256;

// logical input X (N, CI, H, W) -- stored at half resolution when UPIN -- -> Y (N, CO, H, W), or, SUMOUT, the 2x2 sums at half
// resolution.  TRANS = false: w is (CO, CI, 3, 3);  TRANS = true (dgrad: X = dY): w is (CI, CO, 3, 3) and taps are mirrored.
template <int CI, int CO, bool UPIN, bool SUMOUT, bool TRANS>
__global__ __launch_bounds__(TPB) void c16_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y, int H, int W, int act,
                                                       int accumulate) {
    __shared__ __attribute__((aligned(16))) float wl[CI * 9 * CO];     // [ci][tap][co]
    for (int i = threadIdx.x; i < CI * 9 * CO; i += TPB) {
        const int co = i % CO, tap = (i / CO) % 9, ci = i / (9 * CO);
        wl[i] = TRANS ? w[((size_t)ci * CO + co) * 9 + (8 - tap)] : w[((size_t)co * CI + ci) * 9 + tap];
    }
    __syncthreads();
    const int img = blockIdx.z;
    const int y0 = blockIdx.y * 32 + (threadIdx.x >> 4) * 2, x0 = blockIdx.x * 32 + (threadIdx.x & 15) * 2;
    const int sh = UPIN ? 1 : 0, hs = H >> sh, ws = W >> sh;
    const float* xp = x + (size_t)img * CI * hs * ws;
    // window offsets (4 x 4, zero padding) inside one channel plane; -1 = outside
    int off[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int yy = y0 - 1 + a, xx = x0 - 1 + b;
            off[a][b] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? (yy >> sh) * ws + (xx >> sh) : -1;
        }
    float acc[4][CO];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[p][c] = 0.f;
#pragma unroll 1
    for (int ci = 0; ci < CI; ++ci) {
        const float* q = xp + (size_t)ci * hs * ws;
        float v[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) v[a][b] = off[a][b] >= 0 ? q[off[a][b]] : 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ty = tap / 3, tx = tap % 3;
            const float4* wr = reinterpret_cast<const float4*>(wl + (ci * 9 + tap) * CO);
#pragma unroll
            for (int c4 = 0; c4 < CO / 4; ++c4) {
                const float4 w4 = wr[c4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float xv = v[(p >> 1) + ty][(p & 1) + tx];
                    acc[p][4 * c4 + 0] = fmaf(w4.x, xv, acc[p][4 * c4 + 0]);
                    acc[p][4 * c4 + 1] = fmaf(w4.y, xv, acc[p][4 * c4 + 1]);
                    acc[p][4 * c4 + 2] = fmaf(w4.z, xv, acc[p][4 * c4 + 2]);
                    acc[p][4 * c4 + 3] = fmaf(w4.w, xv, acc[p][4 * c4 + 3]);
                }
            }
        }
    }
    if (SUMOUT) {
        const int h2 = H >> 1, w2 = W >> 1;
        float* yo = y + (size_t)img * CO * h2 * w2 + (size_t)(y0 >> 1) * w2 + (x0 >> 1);
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            const float s = (acc[0][c] + acc[1][c]) + (acc[2][c] + acc[3][c]);
            float* o = yo + (size_t)c * h2 * w2;
            *o = accumulate ? *o + s : s;
        }
    } else {
        float* yo = y + (size_t)img * CO * H * W + (size_t)y0 * W + x0;
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            const float bv = bias ? bias[c] : 0.f;
            float2* o0 = reinterpret_cast<float2*>(yo + (size_t)c * H * W);
            float2* o1 = reinterpret_cast<float2*>(yo + (size_t)c * H * W + W);
            float2 r0 = make_float2(jp_act(acc[0][c] + bv, act), jp_act(acc[1][c] + bv, act));
            float2 r1 = make_float2(jp_act(acc[2][c] + bv, act), jp_act(acc[3][c] + bv, act));
            if (accumulate) { const float2 a0 = *o0, a1 = *o1; r0.x += a0.x; r0.y += a0.y; r1.x += a1.x; r1.y += a1.y; }
            *o0 = r0;
            *o1 = r1;
        }
    }
}

// dw[co][ci][tap] += sum over pixels of dy[co][y][x] * X[ci][y + ty - 1][x + tx - 1]
constexpr int WG_R = 4, WG_C = 64;
// PART: every workgroup writes its partial sums to its own slice of caller scratch (`dw` = scratch + blockIdx.x * CO*CI*9) and
// c16_wgrad_fold_kernel adds the slices in block order -- bit-reproducible run to run; without scratch the workgroups merge
// with fp32 atomics (order = arrival order).
template <int CI, int CO, bool UPIN, bool PART>
__global__ __launch_bounds__(TPB) void c16_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                        int H, int W, int ntiles) {
    constexpr int P = CI * CO / TPB;                       // (co, ci) pairs per thread
    static_assert(CI * CO % TPB == 0, "pairs per thread");
    constexpr int DP = WG_C + 1, XP = WG_C + 3;            // row pitches (floats): bank-conflict-free channel strides
    __shared__ float dT[CO * WG_R * DP];
    __shared__ float xT[CI * (WG_R + 2) * XP];
    const int sh = UPIN ? 1 : 0, hs = H >> sh, ws = W >> sh;
    const int tiles_x = W / WG_C, tiles_img = tiles_x * (H / WG_R);
    float acc[P][9];
#pragma unroll
    for (int k = 0; k < P; ++k)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[k][t] = 0.f;
    for (int T = blockIdx.x; T < ntiles; T += gridDim.x) {
        const int img = T / tiles_img, r_ = T - img * tiles_img;
        const int y0 = (r_ / tiles_x) * WG_R, x0 = (r_ % tiles_x) * WG_C;
        __syncthreads();
        const float* dp = dy + (size_t)img * CO * H * W;
        for (int i = threadIdx.x; i < CO * WG_R * WG_C; i += TPB) {
            const int c = i % WG_C, r = (i / WG_C) % WG_R, co = i / (WG_C * WG_R);
            dT[(co * WG_R + r) * DP + c] = dp[((size_t)co * H + y0 + r) * W + x0 + c];
        }
        const float* xp = x + (size_t)img * CI * hs * ws;
        for (int i = threadIdx.x; i < CI * (WG_R + 2) * (WG_C + 2); i += TPB) {
            const int c = i % (WG_C + 2), r = (i / (WG_C + 2)) % (WG_R + 2), ci = i / ((WG_C + 2) * (WG_R + 2));
            const int yy = y0 - 1 + r, xx = x0 - 1 + c;
            const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            xT[(ci * (WG_R + 2) + r) * XP + c] = in ? xp[((size_t)ci * hs + (yy >> sh)) * ws + (xx >> sh)] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int q = threadIdx.x + TPB * k, co = q / CI, ci = q - co * CI;
            const float* dr = dT + co * WG_R * DP;
            const float* xr = xT + ci * (WG_R + 2) * XP;
#pragma unroll 1
            for (int r = 0; r < WG_R; ++r) {
                float w0[3], w1[3], w2[3];                   // window columns c-1 .. c+1 of rows r-1, r, r+1 (tile coordinates + 1)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    w0[t] = xr[(r + t) * XP + 0];
                    w1[t] = xr[(r + t) * XP + 1];
                }
#pragma unroll 4
                for (int c = 0; c < WG_C; ++c) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) w2[t] = xr[(r + t) * XP + c + 2];
                    const float d = dr[r * DP + c];
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        acc[k][3 * t + 0] = fmaf(d, w0[t], acc[k][3 * t + 0]);
                        acc[k][3 * t + 1] = fmaf(d, w1[t], acc[k][3 * t + 1]);
                        acc[k][3 * t + 2] = fmaf(d, w2[t], acc[k][3 * t + 2]);
                    }
#pragma unroll
                    for (int t = 0; t < 3; ++t) { w0[t] = w1[t]; w1[t] = w2[t]; }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const int q = threadIdx.x + TPB * k;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (PART) dw[(size_t)blockIdx.x * (CO * CI * 9) + (size_t)q * 9 + t] = acc[k][t];
            else atomicAdd(dw + (size_t)q * 9 + t, acc[k][t]);
        }
    }
}

__global__ void c16_wgrad_fold_kernel(const float* __restrict__ part, float* __restrict__ dw, int n, int nblocks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int b = 0; b < nblocks; ++b) s += part[(size_t)b * n + i];
    dw[i] += s;
}

template <bool UPIN, bool SUMOUT, bool TRANS>
int launch_conv(const float* x, const float* w, const float* bias, float* y, int N, int CI, int CO, int H, int W, int act, int accumulate,
                hipStream_t st) {
    const dim3 grid(W / 32, H / 32, N);
#define JP_GO(A, B)                                                                                                          \
    hipLaunchKernelGGL((c16_conv_kernel<A, B, UPIN, SUMOUT, TRANS>), grid, dim3(TPB), 0, st, x, w, bias, y, H, W, act, accumulate)
    if (CI == 16 && CO == 16) JP_GO(16, 16);
    else if (CI == 32 && CO == 32) JP_GO(32, 32);
    else if (CI == 32 && CO == 16) JP_GO(32, 16);
    else if (CI == 16 && CO == 32) JP_GO(16, 32);
    else return 1;
#undef JP_GO
    return 0;
}

}  // namespace

// internal entry points used by conv.hip's dispatcher (not part of the public ABI)
bool jp_c16_ok(int Cin, int Cout, int KH, int stride, int pad, int pad_mode, int H, int W) {
    static const bool on = [] { const char* e = getenv("JP_C16"); return !(e && e[0] == '0'); }();
    // 16 -> 16 only: measured at 8 x 32 x 128^2 the 32-channel instantiations lose to the implicit-GEMM engine (forward 32 -> 32
    // 0.085 vs 0.062 ms: 128 accumulators = one wave per SIMD; wgrad 0.32 vs 0.09 ms: four (co, ci) pairs per thread, LDS-bound)
    return on && Cin == 16 && Cout == 16 && KH == 3 && stride == 1 && pad == 1 && pad_mode == 0 &&
           H % 32 == 0 && W % 64 == 0 && H >= 64;
}
// forward: x (N, Cin, H >> up, W >> up) -> y (N, Cout, H, W)
int jp_c16_fwd(const float* x, int up, const float* w, const float* bias, float* y, int N, int Cin, int Cout, int H, int W, int act,
               hipStream_t st) {
    return up ? launch_conv<true, false, false>(x, w, bias, y, N, Cin, Cout, H, W, act, 0, st)
              : launch_conv<false, false, false>(x, w, bias, y, N, Cin, Cout, H, W, act, 0, st);
}
// dgrad: dy (N, Cout, H, W) -> dx (N, Cin, H >> up, W >> up)  (= or +=)
int jp_c16_dgrad(const float* dy, const float* w, float* dx, int up, int N, int Cin, int Cout, int H, int W, int accumulate,
                 hipStream_t st) {
    return up ? launch_conv<false, true, true>(dy, w, nullptr, dx, N, Cout, Cin, H, W, 0, accumulate, st)
              : launch_conv<false, false, true>(dy, w, nullptr, dx, N, Cout, Cin, H, W, 0, accumulate, st);
}
// wgrad: dw (Cout, Cin, 3, 3) += ...
long jp_c16_wgrad_ws_floats(int N, int Cin, int Cout, int H, int W) {
    return (long)std::min(N * (H / WG_R) * (W / WG_C), 512) * Cout * Cin * 9;
}
int jp_c16_wgrad(const float* x, int up, const float* dy, float* dw, int N, int Cin, int Cout, int H, int W, hipStream_t st,
                 float* ws, long ws_floats) {
    const int ntiles = N * (H / WG_R) * (W / WG_C);
    const dim3 grid(std::min(ntiles, 512));
    const int nw = Cout * Cin * 9;
    const bool part = ws && ws_floats >= (long)grid.x * nw;       // fixed-order merge through caller scratch
    float* out = part ? ws : dw;
#define JP_GO1(A, B, U, P) hipLaunchKernelGGL((c16_wgrad_kernel<A, B, U, P>), grid, dim3(TPB), 0, st, x, dy, out, H, W, ntiles)
#define JP_GO(A, B)                                                                     \
    {                                                                                   \
        if (up) { if (part) JP_GO1(A, B, true, true); else JP_GO1(A, B, true, false); } \
        else { if (part) JP_GO1(A, B, false, true); else JP_GO1(A, B, false, false); }  \
    }
    if (Cin == 16 && Cout == 16) JP_GO(16, 16)
    else if (Cin == 32 && Cout == 32) JP_GO(32, 32)
    else if (Cin == 32 && Cout == 16) JP_GO(32, 16)
    else if (Cin == 16 && Cout == 32) JP_GO(16, 32)
    else return 1;
#undef JP_GO
#undef JP_GO1
    if (part) hipLaunchKernelGGL(c16_wgrad_fold_kernel, dim3((nw + 255) / 256), dim3(256), 0, st, ws, dw, nw, (int)grid.x);
    return 0;
}
