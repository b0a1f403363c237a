// The CGT view-synthesis + photometric loss path of the JPerceiver train step, forward and backward:
//   pose      : Rodrigues + (T*R | R^T*T(-t)) + P = K*T        (net.py:704-756, layers.py:74)
//   cgt warp  : bilinear-up(disp) -> depth -> backproject -> project -> grid_sample(bilinear, border,
//               align_corners=False)                           (net.py:690-702, layers.py:57-61,73-82)
//   ssim + l1 : 0.85*mean_c SSIM + 0.15*mean_c sqrt((t-p)^2+1e-6)  (layers.py:97-107, net.py:84-92)
//   min-reproj: min/argmin over identity(+noise) and warped candidates, mean  (net.py:159-175)
// All HBM-bound.  SSIM forward does its 3x3 window sums in registers: a wave owns a 64-pixel-wide
// column strip, horizontal taps come from neighbouring lanes (DPP/ds_bpermute wave shuffles), the
// vertical taps from a 3-row sliding window, so every pixel is read once (+2/ROWS halo rows).
#include "jp_common.h"
#include <algorithm>

namespace {

constexpr int TPB = 256;
constexpr float SSIM_C1 = 0.0001f, SSIM_C2 = 0.0009f;

// ------------------------------------------------------------------------------ pose
__device__ void rodrigues(const float v[3], float R[9]) {
    const float th = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float inv = 1.f / (th + 1e-7f);
    const float x = v[0] * inv, y = v[1] * inv, z = v[2] * inv;
    const float ca = cosf(th), sa = sinf(th), C = 1.f - ca;
    R[0] = x * x * C + ca;     R[1] = x * y * C - z * sa; R[2] = z * x * C + y * sa;
    R[3] = x * y * C + z * sa; R[4] = y * y * C + ca;     R[5] = y * z * C - x * sa;
    R[6] = z * x * C - y * sa; R[7] = y * z * C + x * sa; R[8] = z * z * C + ca;
}

// one thread per batch element.  aa/tr: (B,3).  T: (B,4,4) cam_T_cam.  P: (B,3,4) = (K*T)[:3]
__global__ void pose_fwd_kernel(const float* __restrict__ aa, const float* __restrict__ tr,
                                const float* __restrict__ K, float* __restrict__ T, float* __restrict__ P, int B,
                                int invert) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float v[3] = {aa[3 * b], aa[3 * b + 1], aa[3 * b + 2]};
    float t[3] = {tr[3 * b], tr[3 * b + 1], tr[3 * b + 2]};
    float R[9], M[16];
    rodrigues(v, R);
    for (int i = 0; i < 16; ++i) M[i] = 0.f;
    M[15] = 1.f;
    if (!invert) {  // M = T(t) * R
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) M[4 * i + j] = R[3 * i + j];
            M[4 * i + 3] = t[i];
        }
    } else {        // M = R^T * T(-t)
        for (int i = 0; i < 3; ++i) {
            float s = 0.f;
            for (int j = 0; j < 3; ++j) {
                M[4 * i + j] = R[3 * j + i];
                s += R[3 * j + i] * (-t[j]);
            }
            M[4 * i + 3] = s;
        }
    }
    for (int i = 0; i < 16; ++i) T[16 * b + i] = M[i];
    const float* Kb = K + 16 * b;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
            for (int k = 0; k < 4; ++k) s += Kb[4 * i + k] * M[4 * k + j];
            P[12 * b + 4 * i + j] = s;
        }
}

// dP (B,3,4) as doubles accumulated by the warp backward -> d axisangle, d translation.
// Evaluated in DOUBLE (round 6): one thread per image, so it costs nothing, and the float version was the weakest link of the pose
// gradient -- for the small rotations of a pose network (|axisangle| ~ 1e-3) the chain through th = |v|, a = v / th, 1 - cos th
// cancels catastrophically in fp32 (1 - cos(3e-3) = 4.5e-6 carries ~1 % of float rounding, da * inv and dth * v / th nearly cancel):
// tools/debug/two_step_referee.py measured the pose networks' gradients 8-10 % from the float64 oracle after one Adam step where the
// fp32 CPU oracle sat at 0.4 %.  The forward keeps the reference's fp32 formulas (net.py:704-756).
__global__ void pose_bwd_kernel(const double* __restrict__ dP, const float* __restrict__ aa,
                                const float* __restrict__ tr, const float* __restrict__ K, float* __restrict__ daa,
                                float* __restrict__ dtr, int B, int invert, int accumulate) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* Kb = K + 16 * b;
    double dM[12];  // rows 0..2 of K[:3,:]^T dP  (row 3 of M is constant)
    for (int k = 0; k < 3; ++k)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int i = 0; i < 3; ++i) s += (double)Kb[4 * i + k] * dP[12 * b + 4 * i + j];
            dM[4 * k + j] = s;
        }
    const double v[3] = {aa[3 * b], aa[3 * b + 1], aa[3 * b + 2]};
    const double t[3] = {tr[3 * b], tr[3 * b + 1], tr[3 * b + 2]};
    const double th = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const double inv = 1.0 / (th + 1e-7);
    const double a[3] = {v[0] * inv, v[1] * inv, v[2] * inv};
    const double ca = cos(th), sa = sin(th), C = 1.0 - ca;
    const double R[9] = {a[0] * a[0] * C + ca,        a[0] * a[1] * C - a[2] * sa, a[2] * a[0] * C + a[1] * sa,
                         a[0] * a[1] * C + a[2] * sa, a[1] * a[1] * C + ca,        a[1] * a[2] * C - a[0] * sa,
                         a[2] * a[0] * C - a[1] * sa, a[1] * a[2] * C + a[0] * sa, a[2] * a[2] * C + ca};
    double G[9], dt[3];
    if (!invert) {
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) G[3 * i + j] = dM[4 * i + j];
            dt[i] = dM[4 * i + 3];
        }
    } else {
        const double g3[3] = {dM[3], dM[7], dM[11]};
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int i = 0; i < 3; ++i) {
                G[3 * j + i] = dM[4 * i + j] - g3[i] * t[j];
                s += R[3 * j + i] * g3[i];
            }
            dt[j] = -s;
        }
    }
    double aGa = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) aGa += a[i] * G[3 * i + j] * a[j];
    const double dca = G[0] + G[4] + G[8] - aGa;
    const double dsa = -a[2] * G[1] + a[1] * G[2] + a[2] * G[3] - a[0] * G[5] - a[1] * G[6] + a[0] * G[7];
    double da[3] = {sa * (-G[5] + G[7]), sa * (G[2] - G[6]), sa * (-G[1] + G[3])};
    for (int i = 0; i < 3; ++i) {
        double s = 0.0;
        for (int j = 0; j < 3; ++j) s += (G[3 * i + j] + G[3 * j + i]) * a[j];
        da[i] += C * s;
    }
    double dth = -sa * dca + ca * dsa;
    dth -= (da[0] * v[0] + da[1] * v[1] + da[2] * v[2]) * inv * inv;
    for (int i = 0; i < 3; ++i) {
        const float g = (float)(da[i] * inv + (th > 0.0 ? dth * v[i] / th : 0.0));
        daa[3 * b + i] = accumulate ? daa[3 * b + i] + g : g;
        dtr[3 * b + i] = accumulate ? dtr[3 * b + i] + (float)dt[i] : (float)dt[i];
    }
}

// ------------------------------------------------------------------------------ CGT warp
__device__ __forceinline__ void bil_src(int o, float scale, int in, int& i0, int& i1, float& w1) {
    float src = ((float)o + 0.5f) * scale - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    w1 = src - (float)i0;
}

struct WarpGeo {   // per-pixel forward geometry, recomputed identically in backward
    float depth, rx, ry, rz;       // depth and K^-1 ray
    float px, py, pz;              // projected homogeneous point (pz before +eps)
    float ix, iy;                  // clipped source coordinates
    float mx, my;                  // d(clipped)/d(unclipped): 0 when clipped (PyTorch clip_coordinates_set_grad)
};

__device__ __forceinline__ WarpGeo warp_geo(const float* __restrict__ disp, int hs, int ws, const float* iK,
                                            const float* P, int y, int x, int H, int W, float min_disp,
                                            float max_disp, float ssy, float ssx) {
    WarpGeo g;
    int y0, y1, x0, x1;
    float wy, wx;
    bil_src(y, ssy, hs, y0, y1, wy);
    bil_src(x, ssx, ws, x0, x1, wx);
    const float d = (1.f - wy) * ((1.f - wx) * disp[y0 * ws + x0] + wx * disp[y0 * ws + x1]) +
                    wy * ((1.f - wx) * disp[y1 * ws + x0] + wx * disp[y1 * ws + x1]);
    g.depth = 1.f / (min_disp + (max_disp - min_disp) * d);
    const float fx = (float)x, fy = (float)y;
    g.rx = iK[0] * fx + iK[1] * fy + iK[2];
    g.ry = iK[4] * fx + iK[5] * fy + iK[6];
    g.rz = iK[8] * fx + iK[9] * fy + iK[10];
    const float X = g.depth * g.rx, Y = g.depth * g.ry, Z = g.depth * g.rz;
    g.px = P[0] * X + P[1] * Y + P[2] * Z + P[3];
    g.py = P[4] * X + P[5] * Y + P[6] * Z + P[7];
    g.pz = P[8] * X + P[9] * Y + P[10] * Z + P[11];
    const float zi = g.pz + 1e-7f;
    float gx = (g.px / zi / (float)(W - 1) - 0.5f) * 2.f;
    float gy = (g.py / zi / (float)(H - 1) - 0.5f) * 2.f;
    float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
    float iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    g.mx = (ix > 0.f && ix < (float)(W - 1)) ? 1.f : 0.f;   // clip_coordinates_set_grad; NaN -> 0 too
    g.my = (iy > 0.f && iy < (float)(H - 1)) ? 1.f : 0.f;
    g.ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
    g.iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
    if (!(ix == ix)) g.ix = 0.f;
    if (!(iy == iy)) g.iy = 0.f;
    return g;
}

// grid (blocks over H*W, B).  pred (B,3,H,W)
constexpr int CGT_FROWS = 4;
__global__ __launch_bounds__(TPB) void cgt_warp_fwd_kernel(const float* __restrict__ disp, int hs, int ws,
                                                           const float* __restrict__ invK,
                                                           const float* __restrict__ Pm,
                                                           const float* __restrict__ color,
                                                           float* __restrict__ pred, int H, int W, float min_disp,
                                                           float max_disp) {
    // 2-D tile (CGT_FROWS rows x 256 columns per workgroup, a thread walks one column down): consecutive rows share a source
    // row of their bilinear corners in L1 (see cgt_warp_bwd_kernel)
    const int b = blockIdx.y;
    const int x = blockIdx.x * TPB + threadIdx.x;
    if (x >= W) return;
    const size_t HW = (size_t)H * W;
    const float* c = color + (size_t)b * 3 * HW;
#pragma unroll
    for (int it = 0; it < CGT_FROWS; ++it) {
        const int y = blockIdx.z * CGT_FROWS + it;
        if (y >= H) break;
        const int p = y * W + x;
        const WarpGeo g = warp_geo(disp + (size_t)b * hs * ws, hs, ws, invK + 16 * b, Pm + 12 * b, y, x, H, W, min_disp,
                                   max_disp, (float)hs / (float)H, (float)ws / (float)W);
        const int x0 = (int)floorf(g.ix), y0 = (int)floorf(g.iy);
        const float tx = g.ix - (float)x0, ty = g.iy - (float)y0;
        const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);   // weight of an OOB corner is 0
        const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float* cc = c + ch * HW;
            pred[((size_t)b * 3 + ch) * HW + p] =
                cc[y0 * W + x0] * w00 + cc[y0 * W + x1] * w01 + cc[y1 * W + x0] * w10 + cc[y1 * W + x1] * w11;
        }
    }
}

// dpred (B,3,H,W) -> ddisp_up (B,1,H,W) (accumulated over source frames) and dP (B,12) doubles
constexpr int CGT_PPT = 8;
__global__ __launch_bounds__(TPB) void cgt_warp_bwd_kernel(const float* __restrict__ dpred,
                                                           const float* __restrict__ disp, int hs, int ws,
                                                           const float* __restrict__ invK,
                                                           const float* __restrict__ Pm,
                                                           const float* __restrict__ color,
                                                           float* __restrict__ ddisp_up, double* __restrict__ dP,
                                                           int H, int W, float min_disp, float max_disp,
                                                           int accumulate) {
    __shared__ float red[12][4];
    const int b = blockIdx.y;
    float dPl[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) dPl[i] = 0.f;
    // CGT_PPT pixels per thread: the 12 wave reductions + double atomics of dP are paid once per 2048 pixels.  (A 2-D tile --
    // CGT_PPT rows x 256 columns, a thread walking one column down, as cgt_warp_fwd_kernel does -- cuts the source-row
    // re-fetch of this 1-D order from 1.5x to 1.09x on a smooth flow (tools/ubench/fetch_calib.hip), but with the seven
    // tensor streams of the backward it measured SLOWER on the step's data: 107 -> 123 us per call, profiles/r04_photo_ab.log.)
    for (int it = 0; it < CGT_PPT; ++it) {
        const int p = (blockIdx.x * CGT_PPT + it) * TPB + threadIdx.x;
        if (p >= H * W) break;
        const int y = p / W, x = p - y * W;
        const float* Pb = Pm + 12 * b;
        const WarpGeo g = warp_geo(disp + (size_t)b * hs * ws, hs, ws, invK + 16 * b, Pb, y, x, H, W, min_disp,
                                   max_disp, (float)hs / (float)H, (float)ws / (float)W);
        const int x0 = (int)floorf(g.ix), y0 = (int)floorf(g.iy);
        const float tx = g.ix - (float)x0, ty = g.iy - (float)y0;
        const bool xin = x0 + 1 <= W - 1, yin = y0 + 1 <= H - 1;
        const int x1 = xin ? x0 + 1 : x0, y1 = yin ? y0 + 1 : y0;
        const size_t HW = (size_t)H * W;
        const float* c = color + (size_t)b * 3 * HW;
        float gix = 0.f, giy = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float* cc = c + ch * HW;
            const float go = dpred[((size_t)b * 3 + ch) * HW + p];
            const float v00 = cc[y0 * W + x0];
            const float v01 = xin ? cc[y0 * W + x1] : 0.f;
            const float v10 = yin ? cc[y1 * W + x0] : 0.f;
            const float v11 = (xin && yin) ? cc[y1 * W + x1] : 0.f;
            // PyTorch grid_sampler_2d_backward: OOB corners contribute nothing
            gix += go * (-(v00 * (1.f - ty)) + v01 * (1.f - ty) - v10 * ty + v11 * ty);
            giy += go * (-(v00 * (1.f - tx)) - v01 * tx + v10 * (1.f - tx) + v11 * tx);
        }
        // unnormalise (W/2), border clip mask, then Project's normalisation 2/(W-1)
        const float du = gix * g.mx * (0.5f * (float)W) * (2.f / (float)(W - 1));
        const float dv = giy * g.my * (0.5f * (float)H) * (2.f / (float)(H - 1));
        const float zi = g.pz + 1e-7f;
        const float dpx = du / zi, dpy = dv / zi;
        const float dpz = -(du * g.px + dv * g.py) / (zi * zi);
        const float X = g.depth * g.rx, Y = g.depth * g.ry, Z = g.depth * g.rz;
        dPl[0] += dpx * X; dPl[1] += dpx * Y; dPl[2] += dpx * Z; dPl[3] += dpx;
        dPl[4] += dpy * X; dPl[5] += dpy * Y; dPl[6] += dpy * Z; dPl[7] += dpy;
        dPl[8] += dpz * X; dPl[9] += dpz * Y; dPl[10] += dpz * Z; dPl[11] += dpz;
        const float dX = Pb[0] * dpx + Pb[4] * dpy + Pb[8] * dpz;
        const float dY = Pb[1] * dpx + Pb[5] * dpy + Pb[9] * dpz;
        const float dZ = Pb[2] * dpx + Pb[6] * dpy + Pb[10] * dpz;
        const float ddepth = dX * g.rx + dY * g.ry + dZ * g.rz;
        const float dd = -(max_disp - min_disp) * g.depth * g.depth * ddepth;
        float* q = ddisp_up + (size_t)b * HW + p;
        *q = accumulate ? *q + dd : dd;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const float s = jp_wave_sum(dPl[i]);
        if (lane == 0) red[i][wv] = s;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const double s = (double)red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        atomicAdd(&dP[12 * b + threadIdx.x], s);
    }
}

// ------------------------------------------------------------------------------ SSIM + L1 forward
// block = 4 waves; wave w of block (bx, by, b) owns columns [64*(4*bx+w), +64) and rows [ROWS*by, +ROWS)
constexpr int SSIM_ROWS = 4;

struct RowSums { float x, y, xx, yy, xy; };

// One row of the 3x3 window sums at column x.  The three column indices (x-1, x, x+1 with ReflectionPad2d(1) resolved)
// are per-lane constants, so a row costs six independent, coalesced loads and no branch: the row loop below is
// straight-line code whose loads the compiler can keep in flight (the earlier version took the neighbours from
// adjacent lanes and reloaded at strip edges inside per-row branches -- one dependent load step per row: 1.6 TB/s).
__device__ __forceinline__ RowSums ssim_hsum(const float* __restrict__ xr, const float* __restrict__ yr, int il, int ic,
                                             int ir, float& xc, float& yc) {
    const float xl = xr[il], xrn = xr[ir], yl = yr[il], yrn = yr[ir];
    xc = xr[ic];
    yc = yr[ic];
    RowSums s;
    s.x = xl + xc + xrn;
    s.y = yl + yc + yrn;
    s.xx = xl * xl + xc * xc + xrn * xrn;
    s.yy = yl * yl + yc * yc + yrn * yrn;
    s.xy = xl * yl + xc * yc + xrn * yrn;
    return s;
}

__global__ __launch_bounds__(TPB) void ssim_l1_fwd_kernel(const float* __restrict__ pred,
                                                          const float* __restrict__ target,
                                                          float* __restrict__ out, int H, int W) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = (blockIdx.x * 4 + wv) * 64 + lane;
    if ((blockIdx.x * 4 + wv) * 64 >= W) return;   // whole wave out of range (wave-uniform)
    const int b = blockIdx.z;
    const int ybeg = blockIdx.y * SSIM_ROWS, yend = min(H, ybeg + SSIM_ROWS);
    const size_t HW = (size_t)H * W;
    const int ic = min(x, W - 1), il = jp_reflect(ic - 1, W), ir = jp_reflect(ic + 1, W);
    // row offsets of the window rows -1 .. SSIM_ROWS (reflection at the image border; rows past the strip end are
    // clamped to a valid row and their results dropped)
    float accS[SSIM_ROWS], accL[SSIM_ROWS];
#pragma unroll
    for (int r = 0; r < SSIM_ROWS; ++r) accS[r] = accL[r] = 0.f;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float* xp = pred + ((size_t)b * 3 + ch) * HW;
        const float* yp = target + ((size_t)b * 3 + ch) * HW;
        RowSums r0, r1, r2;
        float xc, yc, xc1, yc1;
        {
            const size_t oa = (size_t)jp_reflect(ybeg - 1, H) * W, ob = (size_t)ybeg * W;
            r0 = ssim_hsum(xp + oa, yp + oa, il, ic, ir, xc, yc);
            r1 = ssim_hsum(xp + ob, yp + ob, il, ic, ir, xc1, yc1);
        }
#pragma unroll
        for (int r = 0; r < SSIM_ROWS; ++r) {
            const int y = min(ybeg + r, H - 1);
            const size_t on = (size_t)jp_reflect(y + 1, H) * W;
            float xc2, yc2;
            r2 = ssim_hsum(xp + on, yp + on, il, ic, ir, xc2, yc2);
            const float k = 1.f / 9.f;
            const float mx = (r0.x + r1.x + r2.x) * k, my = (r0.y + r1.y + r2.y) * k;
            const float sx = (r0.xx + r1.xx + r2.xx) * k - mx * mx;
            const float sy = (r0.yy + r1.yy + r2.yy) * k - my * my;
            const float sxy = (r0.xy + r1.xy + r2.xy) * k - mx * my;
            const float n = (2.f * mx * my + SSIM_C1) * (2.f * sxy + SSIM_C2);
            const float d = (mx * mx + my * my + SSIM_C1) * (sx + sy + SSIM_C2);
            accS[r] += fminf(fmaxf((1.f - n / d) * 0.5f, 0.f), 1.f);
            const float df = yc1 - xc1, v = df * df + 1e-6f;
            accL[r] += v * rsqrtf(v);                  // sqrt(v), v >= 1e-6: one v_rsq_f32 instead of the IEEE sqrt sequence
            r0 = r1; r1 = r2; xc1 = xc2; yc1 = yc2;
        }
    }
    if (x < W) {
#pragma unroll
        for (int r = 0; r < SSIM_ROWS; ++r) {
            const int y = ybeg + r;
            if (y < yend) out[(size_t)b * HW + (size_t)y * W + x] = 0.85f * (accS[r] / 3.f) + 0.15f * (accL[r] / 3.f);
        }
    }
}

// ------------------------------------------------------------------------------ SSIM + L1 backward
// dpred = d( sum_q g_q * reproj(q) ) / d pred, g_q = gscale * [min_index(q) == cand] (or 1 if no index).
// Two nested 3x3 stencils: the SSIM value of window centre q depends on the 3x3 neighbourhood of q (reflection padded),
// so pixel r collects  SA + x_r*SB + y_r*SG  with S* = sum over the (in-image) centres q around r of w(r,q) * (A,B,G)_q --
// w counts how often the reflection padding shows r to window q (twice when q sits on the border and r next to it).
// Row-streaming form (like the forward): a wave owns 64 coefficient columns c0-1 .. c0+62 and marches down the rows.
// Per row it loads the six input values of its column (per-lane constant, reflection-resolved column indices: no
// branches), forms the window sums from a 3-row register window, the coefficients (A, B, G), their weighted horizontal
// 3-sums through wave shuffles, and -- two rows later -- the output row from a 3-row window of those.  Lanes 1..62
// produce outputs (neighbouring strips overlap by two columns).  No LDS, no barrier; the earlier LDS-tile version ran
// three barrier-separated phases per channel on 32x16 tiles (1.4 TB/s).
constexpr int SSIMB_ROWS = 8;

__global__ __launch_bounds__(TPB) void ssim_l1_bwd_kernel(const float* __restrict__ pred,
                                                          const float* __restrict__ target,
                                                          const int64_t* __restrict__ min_index, int cand,
                                                          const float* __restrict__ gout, float gscale,
                                                          float* __restrict__ dpred, int H, int W) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int strip = blockIdx.x * 4 + wv;
    if (strip * 62 >= W) return;                             // wave-uniform
    const int cq = strip * 62 - 1 + lane;                    // this lane's coefficient / output column
    const bool col_in = cq >= 0 && cq < W;
    const bool col_out = col_in && lane >= 1 && lane <= 62;
    const int ic = min(max(cq, 0), W - 1), il = jp_reflect(ic - 1, W), ir = jp_reflect(ic + 1, W);
    const float wl = (cq == 1) ? 2.f : 1.f, wr = (cq == W - 2) ? 2.f : 1.f;       // weights of the columns cq-1 / cq+1
    const int b = blockIdx.z;
    const int ybeg = blockIdx.y * SSIMB_ROWS, yend = min(H, ybeg + SSIMB_ROWS);
    const size_t HW = (size_t)H * W;
    const float gs = gscale * (gout ? gout[0] : 1.f);
    const int* mi32 = min_index ? reinterpret_cast<const int*>(min_index + (size_t)b * HW) : nullptr;   // low dwords
#pragma unroll 1
    for (int ch = 0; ch < 3; ++ch) {
        const float* xp = pred + ((size_t)b * 3 + ch) * HW;
        const float* yp = target + ((size_t)b * 3 + ch) * HW;
        float* dp = dpred + ((size_t)b * 3 + ch) * HW;
        RowSums r0, r1, r2;
        float xc1, yc1, xc2, yc2;                         // centre values of the window rows
        {
            float xd, yd;
            const size_t oa = (size_t)jp_reflect(min(max(ybeg - 2, -1), H), H) * W;
            const size_t ob = (size_t)jp_reflect(ybeg - 1, H) * W;
            r0 = ssim_hsum(xp + oa, yp + oa, il, ic, ir, xd, yd);
            r1 = ssim_hsum(xp + ob, yp + ob, il, ic, ir, xc1, yc1);
        }
        float hA0 = 0.f, hB0 = 0.f, hG0 = 0.f, hA1 = 0.f, hB1 = 0.f, hG1 = 0.f;
        float xo = 0.f, yo = 0.f, go = 0.f;               // centre values / mask of the row that is output next
        // software prefetch: the six values of the NEXT step's row (and its mask word) are requested one step ahead
        auto rowoff = [&](int t) -> size_t { return (size_t)jp_reflect(min(ybeg + t, H), H) * W; };   // row q+1 of step t
        auto qrow = [&](int t) -> int { return min(max(ybeg - 1 + t, 0), H - 1); };
        size_t on = rowoff(0);
        float nxl = xp[on + il], nxc = xp[on + ic], nxr = xp[on + ir], nyl = yp[on + il], nyc = yp[on + ic], nyr = yp[on + ir];
        int nmi = mi32 ? mi32[2 * ((size_t)qrow(0) * W + ic)] : cand;
        // a real loop (not unrolled): unrolled, the compiler hoists every row's loads to the top and needs 246 VGPRs;
        // rolled, the body keeps ~70 and eight waves per SIMD hide what the one-step prefetch does not
#pragma unroll 1
        for (int t = 0; t < SSIMB_ROWS + 2; ++t) {
            const int q = ybeg - 1 + t;                    // coefficient row of this step (wave-uniform)
            const float xl = nxl, xr = nxr, yl = nyl, yr = nyr;
            xc2 = nxc; yc2 = nyc;
            const int miv = nmi;
            if (t + 1 < SSIMB_ROWS + 2) {
                on = rowoff(t + 1);
                nxl = xp[on + il]; nxc = xp[on + ic]; nxr = xp[on + ir];
                nyl = yp[on + il]; nyc = yp[on + ic]; nyr = yp[on + ir];
                if (mi32) nmi = mi32[2 * ((size_t)qrow(t + 1) * W + ic)];
            }
            r2.x = xl + xc2 + xr;
            r2.y = yl + yc2 + yr;
            r2.xx = xl * xl + xc2 * xc2 + xr * xr;
            r2.yy = yl * yl + yc2 * yc2 + yr * yr;
            r2.xy = xl * yl + xc2 * yc2 + xr * yr;
            // ---- coefficients of window centre (q, cq)
            const bool q_in = q >= 0 && q < H && col_in;
            const float g = (q_in && miv == cand) ? gs : 0.f;
            const float k = 1.f / 9.f;
            const float mx = (r0.x + r1.x + r2.x) * k, my = (r0.y + r1.y + r2.y) * k;
            const float vx = (r0.xx + r1.xx + r2.xx) * k - mx * mx, vy = (r0.yy + r1.yy + r2.yy) * k - my * my;
            const float cxy = (r0.xy + r1.xy + r2.xy) * k - mx * my;
            const float n1 = 2.f * mx * my + SSIM_C1, n2 = 2.f * cxy + SSIM_C2;
            const float d1 = mx * mx + my * my + SSIM_C1, d2 = vx + vy + SSIM_C2;
            const float n = n1 * n2, d = d1 * d2;
            // one IEEE division per window (the reciprocal of d); the five quotients of the formulas are products with it
            const float invd = 1.f / d, nd = n * invd;
            const float val = (1.f - nd) * 0.5f;
            const float gk = (val >= 0.f && val <= 1.f) ? g * (2.f / 9.f) : 0.f;
            const float G = gk * n1 * invd;
            const float Bc = -gk * nd * d1 * invd;
            const float A = gk * (my * (n2 - n1) * invd - nd * mx * (d2 - d1) * invd);
            // ---- weighted horizontal 3-sums (columns outside the image carry zeros: gk == 0 there)
            const float hA2 = wl * __shfl_up(A, 1, 64) + A + wr * __shfl_down(A, 1, 64);
            const float hB2 = wl * __shfl_up(Bc, 1, 64) + Bc + wr * __shfl_down(Bc, 1, 64);
            const float hG2 = wl * __shfl_up(G, 1, 64) + G + wr * __shfl_down(G, 1, 64);
            // ---- output row ry = q - 1 from the coefficient rows q-2 (h*0), q-1 (h*1), q (h*2)
            if (t >= 2) {
                const int ry = q - 1;
                if (ry < yend && col_out) {
                    const float wu = (ry == 1) ? 2.f : 1.f, wd = (ry == H - 2) ? 2.f : 1.f;
                    const float SA = wu * hA0 + hA1 + wd * hA2;
                    const float SB = wu * hB0 + hB1 + wd * hB2;
                    const float SG = wu * hG0 + hG1 + wd * hG2;
                    const float df = xo - yo;
                    const float l1 = go * df * rsqrtf(df * df + 1e-6f);
                    dp[(size_t)ry * W + cq] = -0.5f * (0.85f / 3.f) * (SA + xo * SB + yo * SG) + (0.15f / 3.f) * l1;
                }
            }
            // the row whose coefficients were just formed (q) is output next step: keep its centre values and mask
            xo = xc1; yo = yc1; go = g;
            hA0 = hA1; hB0 = hB1; hG0 = hG1;
            hA1 = hA2; hB1 = hB2; hG1 = hG2;
            r0 = r1; r1 = r2;
            xc1 = xc2; yc1 = yc2;
        }
    }
}

// ------------------------------------------------------------------------------ min-reprojection
// cands c0..c3 (B,1,H,W) (null when absent); noise n0,n1 scaled by 1e-5 is added to c0,c1 (the identity
// candidates, net.py:163).  Writes argmin (int64, reference dtype) and accumulates sum(min) as double.
__global__ __launch_bounds__(TPB) void minreproj_kernel(const float* __restrict__ c0, const float* __restrict__ c1,
                                                        const float* __restrict__ c2, const float* __restrict__ c3,
                                                        const float* __restrict__ n0, const float* __restrict__ n1,
                                                        int64_t* __restrict__ idx, double* __restrict__ acc,
                                                        long total) {
    __shared__ double sm[4];
    double s = 0.0;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
        float best = c0[i] + (n0 ? n0[i] * 1e-5f : 0.f);
        int bi = 0;
        if (c1) { const float v = c1[i] + (n1 ? n1[i] * 1e-5f : 0.f); if (v < best) { best = v; bi = 1; } }
        if (c2) { const float v = c2[i]; if (v < best) { best = v; bi = 2; } }
        if (c3) { const float v = c3[i]; if (v < best) { best = v; bi = 3; } }
        idx[i] = bi;
        s += (double)best;
    }
    s = jp_block_sum_d(s, sm);
    if (threadIdx.x == 0) atomicAdd(acc, s);
}

__global__ void scalar_finalize_kernel(const double* __restrict__ in, float* __restrict__ out, int n,
                                       double scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)(in[i] * scale);
}

// ------------------------------------------------------------------------------ standalone §8b callables
// SSIM()(x, y) (layers.py:85-107): per-channel (B,C,H,W) loss map clamp((1 - SSIM_n/SSIM_d)/2, 0, 1) with
// ReflectionPad2d(1) + AvgPool2d(3,1).  The train step uses the fused SSIM+L1 kernels above; this is the
// module's public forward.  One thread per pixel, taps through the L1/L2 (not a hot path).
__global__ __launch_bounds__(TPB) void ssim_map_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                       float* __restrict__ out, long total, int H, int W) {
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
        const int px = (int)(i % W);
        const long t = i / W;
        const int py = (int)(t % H);
        const long plane = (t / H) * (long)H * W;
        float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = jp_reflect(py + dy, H);
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = jp_reflect(px + dx, W);
                const float a = x[plane + (long)yy * W + xx], b = y[plane + (long)yy * W + xx];
                sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
            }
        }
        const float k = 1.f / 9.f;
        const float mx = sx * k, my = sy * k;
        const float vx = sxx * k - mx * mx, vy = syy * k - my * my, cxy = sxy * k - mx * my;
        const float n = (2.f * mx * my + SSIM_C1) * (2.f * cxy + SSIM_C2);
        const float d = (mx * mx + my * my + SSIM_C1) * (vx + vy + SSIM_C2);
        out[i] = fminf(fmaxf((1.f - n / d) * 0.5f, 0.f), 1.f);
    }
}

// Backproject(B,H,W)(depth, inv_K) (layers.py:41-61): cam_points (B,4,H*W) = [depth * invK[:3,:3] @ (x, y, 1); 1]
__global__ __launch_bounds__(TPB) void backproject_kernel(const float* __restrict__ depth, const float* __restrict__ invK,
                                                          float* __restrict__ out, int B, int H, int W) {
    const long HW = (long)H * W;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < B * HW; i += (long)gridDim.x * TPB) {
        const int b = (int)(i / HW);
        const long p = i - b * HW;
        const float fx = (float)(p % W), fy = (float)(p / W);
        const float* k = invK + 16 * b;
        const float d = depth[i];
        float* o = out + (long)b * 4 * HW + p;
        o[0] = d * (k[0] * fx + k[1] * fy + k[2]);
        o[HW] = d * (k[4] * fx + k[5] * fy + k[6]);
        o[2 * HW] = d * (k[8] * fx + k[9] * fy + k[10]);
        o[3 * HW] = 1.f;
    }
}

// Project(B,H,W)(points, K, T) (layers.py:64-82): pix (B,H,W,2) = ((P @ points)[:2] / (z + eps)) / (W-1, H-1), then
// (x - 0.5) * 2 with P = (K @ T)[:3]
__global__ __launch_bounds__(TPB) void project_kernel(const float* __restrict__ pts, const float* __restrict__ K,
                                                      const float* __restrict__ T, float* __restrict__ out, int B, int H,
                                                      int W, float eps) {
    const long HW = (long)H * W;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < B * HW; i += (long)gridDim.x * TPB) {
        const int b = (int)(i / HW);
        const long p = i - b * HW;
        const float* k = K + 16 * b;
        const float* t = T + 16 * b;
        float P[12];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                P[4 * r + c] = k[4 * r] * t[c] + k[4 * r + 1] * t[4 + c] + k[4 * r + 2] * t[8 + c] + k[4 * r + 3] * t[12 + c];
        const float* q = pts + (long)b * 4 * HW + p;
        const float X = q[0], Y = q[HW], Z = q[2 * HW], Wc = q[3 * HW];
        const float cx = P[0] * X + P[1] * Y + P[2] * Z + P[3] * Wc;
        const float cy = P[4] * X + P[5] * Y + P[6] * Z + P[7] * Wc;
        const float cz = P[8] * X + P[9] * Y + P[10] * Z + P[11] * Wc;
        const float inv = 1.f / (cz + eps);
        out[2 * i] = (cx * inv / (float)(W - 1) - 0.5f) * 2.f;
        out[2 * i + 1] = (cy * inv / (float)(H - 1) - 0.5f) * 2.f;
    }
}

}  // namespace

#define JP_ST hipStream_t st = (hipStream_t)stream

extern "C" int jp_pose_fwd(const float* axisangle, const float* translation, const float* K, float* T, float* P,
                           int B, int invert, void* stream) {
    JP_CHECK_ARG(axisangle && translation && K && T && P && B > 0, "pose_fwd: bad args");
    JP_ST;
    hipLaunchKernelGGL(pose_fwd_kernel, dim3(jp_cdiv(B, 64)), dim3(64), 0, st, axisangle, translation, K, T, P, B, invert);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_pose_bwd(const double* dP, const float* axisangle, const float* translation, const float* K,
                           float* daxisangle, float* dtranslation, int B, int invert, int accumulate, void* stream) {
    JP_CHECK_ARG(dP && axisangle && translation && K && daxisangle && dtranslation && B > 0, "pose_bwd: bad args");
    JP_ST;
    hipLaunchKernelGGL(pose_bwd_kernel, dim3(jp_cdiv(B, 64)), dim3(64), 0, st, dP, axisangle, translation, K,
                       daxisangle, dtranslation, B, invert, accumulate);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_cgt_warp_fwd(const float* disp, int hs, int ws, const float* invK, const float* P,
                               const float* color, float* pred, int B, int H, int W, float min_depth,
                               float max_depth, void* stream) {
    JP_CHECK_ARG(disp && invK && P && color && pred && B > 0 && H > 1 && W > 1, "cgt_warp_fwd: bad args");
    JP_ST;
    hipLaunchKernelGGL(cgt_warp_fwd_kernel, dim3(jp_cdiv(W, TPB), B, jp_cdiv(H, CGT_FROWS)), dim3(TPB), 0, st, disp, hs, ws, invK, P,
                       color, pred, H, W, 1.f / max_depth, 1.f / min_depth);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_cgt_warp_bwd(const float* dpred, const float* disp, int hs, int ws, const float* invK,
                               const float* P, const float* color, float* ddisp_up, double* dP, int B, int H, int W,
                               float min_depth, float max_depth, int accumulate, void* stream) {
    JP_CHECK_ARG(dpred && disp && invK && P && color && ddisp_up && dP && B > 0, "cgt_warp_bwd: bad args");
    JP_ST;
    hipLaunchKernelGGL(cgt_warp_bwd_kernel, dim3(jp_cdiv((long)H * W, TPB * CGT_PPT), B), dim3(TPB), 0, st, dpred, disp, hs, ws,
                       invK, P, color, ddisp_up, dP, H, W, 1.f / max_depth, 1.f / min_depth, accumulate);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_ssim_l1_fwd(const float* pred, const float* target, float* out, int B, int H, int W,
                              void* stream) {
    JP_CHECK_ARG(pred && target && out && B > 0 && H >= 2 && W >= 2, "ssim_l1_fwd: bad args");
    JP_ST;
    dim3 grid(jp_cdiv(W, 256), jp_cdiv(H, SSIM_ROWS), B);
    hipLaunchKernelGGL(ssim_l1_fwd_kernel, grid, dim3(TPB), 0, st, pred, target, out, H, W);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_ssim_l1_bwd(const float* pred, const float* target, const int64_t* min_index, int cand,
                              const float* gout, float gscale, float* dpred, int B, int H, int W, void* stream) {
    JP_CHECK_ARG(pred && target && dpred && B > 0 && H >= 2 && W >= 2, "ssim_l1_bwd: bad args");
    JP_ST;
    dim3 grid(jp_cdiv(jp_cdiv(W, 62), 4), jp_cdiv(H, SSIMB_ROWS), B);
    hipLaunchKernelGGL(ssim_l1_bwd_kernel, grid, dim3(TPB), 0, st, pred, target, min_index, cand, gout, gscale, dpred,
                       H, W);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_minreproj_fwd(const float* c0, const float* c1, const float* c2, const float* c3,
                                const float* noise0, const float* noise1, int64_t* min_index, double* acc,
                                long total, void* stream) {
    JP_CHECK_ARG(c0 && min_index && acc && total > 0, "minreproj_fwd: bad args");
    JP_ST;
    JP_HIP(hipMemsetAsync(acc, 0, sizeof(double), st));
    hipLaunchKernelGGL(minreproj_kernel, dim3((int)std::min<long>((total + TPB - 1) / TPB, 2048)), dim3(TPB), 0, st,
                       c0, c1, c2, c3, noise0, noise1, min_index, acc, total);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_scalar_finalize(const double* in, float* out, int n, double scale, void* stream) {
    JP_CHECK_ARG(in && out && n > 0, "scalar_finalize: bad args");
    JP_ST;
    hipLaunchKernelGGL(scalar_finalize_kernel, dim3(jp_cdiv(n, 64)), dim3(64), 0, st, in, out, n, scale);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_ssim_map(const float* x, const float* y, float* out, int NC, int H, int W, void* stream) {
    JP_CHECK_ARG(x && y && out && NC > 0 && H >= 2 && W >= 2, "ssim_map: bad args");
    JP_ST;
    const long total = (long)NC * H * W;
    hipLaunchKernelGGL(ssim_map_kernel, dim3((int)std::min<long>((total + TPB - 1) / TPB, 65535)), dim3(TPB), 0, st, x, y, out,
                       total, H, W);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_backproject(const float* depth, const float* invK, float* out, int B, int H, int W, void* stream) {
    JP_CHECK_ARG(depth && invK && out && B > 0 && H > 0 && W > 0, "backproject: bad args");
    JP_ST;
    const long total = (long)B * H * W;
    hipLaunchKernelGGL(backproject_kernel, dim3((int)std::min<long>((total + TPB - 1) / TPB, 65535)), dim3(TPB), 0, st, depth,
                       invK, out, B, H, W);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_project(const float* points, const float* K, const float* T, float* out, int B, int H, int W, float eps,
                          void* stream) {
    JP_CHECK_ARG(points && K && T && out && B > 0 && H > 1 && W > 1, "project: bad args");
    JP_ST;
    const long total = (long)B * H * W;
    hipLaunchKernelGGL(project_kernel, dim3((int)std::min<long>((total + TPB - 1) / TPB, 65535)), dim3(TPB), 0, st, points, K,
                       T, out, B, H, W, eps);
    JP_LAUNCH_CHECK();
}
