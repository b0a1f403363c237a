// "W7": weight gradient of the 7x7 stride-2 pad-3 stem convolutions (3 or 6 input channels -> 64; resnet.py conv1 of the
// depth / layout / pose encoders) in the W9 style (igemm_w9.h).  GEMM view: M = 64 output channels, N = 49*CIN (tap, ci)
// columns, K = N*OH*OW output pixels.  The generic engine gathered the 147-wide B operand element by element through a
// slot table (40 TF, gather-bound).  Here a workgroup walks 4x32 output-pixel tiles: the (13 x 69 x CIN) input patch is
// staged once per tile, a lane owns ONE (tap, ci) column for the whole kernel -- its patch offset is a constant folded
// into the lane's LDS base, the pixel offset of a k-step is a compile-time immediate -- and dY comes straight from global
// memory as 16-byte fragments.  A wave owns both 32-channel row blocks of one 32-column block (2 MFMAs per ds_read_b32);
// 10 waves = NBLK column blocks x KG K groups (CIN = 3: 5 x 2, CIN = 6: 10 x 1).  Every workgroup is one K slice
// (grid = splits); partials ws[slice][64][NBLK*32] are folded by w7_reduce_kernel.
// Preconditions (host-checked): H, W even, (W/2) % 32 == 0, (H/2) % 4 == 0, Cout == 64.
#pragma once
#include "igemm_w9.h"

template <int CIN>
__global__ __launch_bounds__(640) void jp_wgrad_w7_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          float* __restrict__ ws, int H, int W, int ntiles,
                                                          int tiles_per_split, int dy_bytes) {
    constexpr int NREAL = 49 * CIN, NBLK = (NREAL + 31) / 32, KG = 10 / NBLK, NT = 640;
    static_assert(NBLK * KG == 10, "10 waves");
    constexpr int TR = 4, TRG = TR / KG, PRH = 2 * TR + 5, PCW = 69, PITCH = 70;
    constexpr int NEL = CIN * PRH * PCW, NLOAD = (NEL + NT - 1) / NT;
    constexpr int QT = TRG * 4;                                      // quads (8 output pixels) per tile and K group
    __shared__ float patch[CIN * PRH * PITCH];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nb = wave % NBLK, kg = wave / NBLK;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int OH = H / 2, OW = W / 2;
    const int zs = blockIdx.x;
    const int T0 = zs * tiles_per_split, T1 = min(ntiles, T0 + tiles_per_split);
    const int tiles_x = OW / 32, tiles_img = tiles_x * (OH / TR);
    const int OHW = OH * OW;

    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, dy_bytes, 0x00020000);
    int arow[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) arow[a] = ((a * 32 + l31) * OHW + 4 * lhi) * 4;
    auto tile_org = [&](int T, int& img, int& y0, int& x0) {
        const int Tc = min(T, ntiles - 1);
        img = Tc / tiles_img;
        const int r = Tc - img * tiles_img;
        y0 = (r / tiles_x) * TR;
        x0 = (r % tiles_x) * 32;
    };
    float ra[2][2][4];
    auto aload = [&](int slot, int tbase, int qd) {        // quad qd (0..QT-1) of the tile at dY element offset tbase
        const int o = __builtin_amdgcn_readfirstlane((tbase + (qd / 4) * OW + 8 * (qd % 4)) * 4);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const jp_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(drs, arow[a], o, 0);
            const jp_f32x4 v = __builtin_bit_cast(jp_f32x4, u);
#pragma unroll
            for (int j = 0; j < 4; ++j) ra[slot][a][j] = v[j];
        }
    };

    // ---- patch staging: element e = t + NT*q -> (ci, row, col) of the (2*TR+5) x 69 input window, zero outside the image
    float rb[NLOAD];
    auto gload = [&](int T) {
        int img, y0, x0;
        tile_org(T, img, y0, x0);
        const float* xc = x + (long)img * CIN * H * W;
        const int iy0 = 2 * y0 - 3, ix0 = 2 * x0 - 3;
#pragma unroll
        for (int q = 0; q < NLOAD; ++q) {
            const int e = t + NT * q;
            const int ci = e / (PRH * PCW), rem = e - ci * (PRH * PCW);
            const int row = rem / PCW, col = rem - row * PCW;
            const int yy = iy0 + row, xx = ix0 + col;
            const bool ok = e < NEL && yy >= 0 && yy < H && xx >= 0 && xx < W;
            rb[q] = ok ? xc[((long)ci * H + yy) * W + xx] : 0.f;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int q = 0; q < NLOAD; ++q) {
            const int e = t + NT * q;
            const int ci = e / (PRH * PCW), rem = e - ci * (PRH * PCW);
            const int row = rem / PCW, col = rem - row * PCW;
            if (e < NEL) patch[(ci * PRH + row) * PITCH + col] = rb[q];
        }
    };

    jp_f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    // this lane's column n = (tap, ci), tap-major; output pixel (r, c) of the tile reads input (2r + ty, 2c + tx) of the patch
    const int n = min(nb * 32 + l31, NREAL - 1);
    const int tap = n / CIN, ci = n - tap * CIN, ty = tap / 7, tx = tap - ty * 7;
    const float* bp = patch + (ci * PRH + ty + 2 * kg * TRG) * PITCH + tx + 8 * lhi;

    if (T0 < T1) {
        int img, y0, x0;
        tile_org(T0, img, y0, x0);
        int tb = (img * 64) * OHW + (y0 + kg * TRG) * OW + x0;
        aload(0, tb, 0);
        gload(T0);
        for (int T = T0; T < T1; ++T) {
            lstore();
            __syncthreads();
            gload(T + 1);
            tile_org(T + 1, img, y0, x0);
            const int tbn = (img * 64) * OHW + (y0 + kg * TRG) * OW + x0;
            auto boff = [&](int s) -> int {
                const int qd = s / 4, j = s % 4;
                return 2 * (qd / 4) * PITCH + 16 * (qd % 4) + 2 * j;
            };
            float b = bp[boff(0)];
#pragma unroll
            for (int s = 0; s < 4 * QT; ++s) {
                const int qd = s / 4, j = s % 4;
                if (j == 0) {
                    const int qa = qd + 1;
                    if (qa < QT) aload(qa % 2, tb, qa);
                    else aload(qa % 2, tbn, qa - QT);
                }
                const int sn = s + 1 < 4 * QT ? s + 1 : s;
                const float nb_ = bp[boff(sn)];
                const float a0 = ra[qd % 2][0][j], a1 = ra[qd % 2][1][j];
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                b = nb_;
            }
            tb = tbn;
            __syncthreads();
        }
    }

    // ---- partial tile -> ws[zs*KG + kg][m][nb*32 + l31]; C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    constexpr int NP = NBLK * 32;
    float* wz = ws + (long)(zs * KG + kg) * 64 * NP + nb * 32 + l31;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) wz[(a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * NP] = acc[a][r];
}

// dw[m][ci][tap] += sum_s ws[s][m][n = tap*CIN + ci]: 64 outputs x 16 slice lanes per workgroup
template <int CIN>
__global__ __launch_bounds__(1024) void w7_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int slices) {
    constexpr int NREAL = 49 * CIN, NP = (NREAL + 31) / 32 * 32, TOTAL = 64 * NP;
    __shared__ float part[16][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (i < TOTAL)
        for (int k = ty; k < slices; k += 16) s += ws[(size_t)k * TOTAL + i];
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < TOTAL) {
#pragma unroll
        for (int k = 1; k < 16; ++k) s += part[k][tx];
        const int m = i / NP, n = i - m * NP;
        if (n < NREAL) {
            const int tap = n / CIN, ci = n - tap * CIN;
            dw[(m * CIN + ci) * 49 + tap] += s;
        }
    }
}
