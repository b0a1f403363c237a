// "W9" patch kernel: WEIGHT GRADIENT of a 3x3 stride-1 pad-1 convolution on the fp32 MFMA pipe,
//     dW[co][ci][ty][tx] = sum over pixels p of  dY[co][p] * Xpad[ci][p + (ty-1, tx-1)]
// i.e. per tap a GEMM with M = Cout, N = Cin and K = N*H*W pixels, both operands pixel-contiguous in HBM.
//
// The generic engine (igemm.h) transposes BOTH operands through LDS for every 32-pixel K chunk and re-gathers the
// shifted input for each of the 9 taps (2 barriers per 16 k-steps, 1 LDS store + 1 ds_read per operand value).  Here a
// workgroup walks a range of 2-D pixel tiles (TR rows x 32 columns) and per tile
//   * stages the (TR+2) x 34 input PATCH of its 32*NB input channels ONCE, transposed to [pixel][channel] (padding --
//     zero or reflection -- resolved while staging): all 9 taps read their MFMA B fragments from it at compile-time
//     LDS offsets, one ds_read_b32 each, no address arithmetic;
//   * takes the dY operand STRAIGHT FROM GLOBAL MEMORY in fragment form: the K index inside an MFMA is free as long as
//     A and B agree, so k-step j of a "quad" pairs pixel (8q + j) [lanes 0-31] with pixel (8q + 4 + j) [lanes 32-63]
//     and a lane's four k-steps are ONE aligned global_load_dwordx4 of its channel row (prefetched W9_AHEAD quads ahead
//     in a register ring; rows are shared by the 3*NB waves of a wave row through L1/L2).
// A wave owns 2 blocks of 32 output channels x the three taps (ty, 0..2) x one block of 32 input channels: six 32x32
// accumulators, 6 MFMAs per k-step for 3 LDS reads (0.5 reads per MFMA instead of 1) and no barrier inside a tile.
// Workgroup = KG K groups x MW wave rows (64 output channels each) x 3 tap rows x NB input-channel blocks.  KG = 2 (layers
// with <= 64 output channels, whose output tile only feeds 6 waves): two wave groups share the staged patch, take two
// rows of the pixel tile each and write their own partial slice (2*zs + kg) -- 12 waves per CU either way.
//
// Output: split-K partial sums ws[split*KG + kg][m][n = tap*Cm + ci] (plain stores; wgrad_reduce_kernel folds them into dW).
// Preconditions (host-checked): W % 32 == 0, H % TR == 0, Cm % (32*NB) == 0.
#pragma once
#include "igemm.h"

constexpr int W9_TR = 4;              // pixel-tile rows
constexpr int W9_AHEAD = 1;           // dY quads prefetched ahead (register ring of W9_AHEAD + 1 quads x 2 blocks x 4)

typedef float jp_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned jp_u32x4 __attribute__((ext_vector_type(4)));

template <int MW, int NB, int KG, bool REFLECT>
__global__ __launch_bounds__(64 * KG * MW * 3 * NB, (KG * MW * 3 * NB) / 4) void jp_wgrad_w9_kernel(
        const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ ws, int Cout, int Cx, int Cm,
        int H, int W, int ntiles, int tiles_per_split, int dy_bytes) {
    constexpr int NWAVE = KG * MW * 3 * NB, NT = 64 * NWAVE;
    constexpr int TR = W9_TR, PR = TR + 2, PC = 34;
    constexpr int NC = 32 * NB, LDB = NC + 1;                       // patch: [PR*PC pixels][LDB]
    constexpr int ROWS = NC * PR;                                    // (channel, patch row) rows of 32 centre columns
    constexpr int NHALO = (2 * ROWS + NT - 1) / NT;
    constexpr int TRG = TR / KG;                                     // tile rows of one K group
    constexpr int QT = TRG * 4;                                      // quads (of 8 pixels) per tile and K group
    static_assert(TR % KG == 0, "K groups split the tile rows");
    constexpr int RING = W9_AHEAD + 1;
    static_assert(QT % RING == 0, "ring slots must line up across tiles");
    __shared__ float patch[PR * PC * LDB];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int cb = wave % NB, tyw = (wave / NB) % 3, wr = (wave / (3 * NB)) % MW, kg = wave / (3 * NB * MW);
    const int l31 = lane & 31, lhi = lane >> 5;

    // ---- (n tile, m tile, K slice) of this workgroup; every XCD owns whole K slices (all tiles of a slice read the same
    // pixels of both operands: they share them through ONE L2), see jp_igemm_kernel
    int mt, nt, zs;
    {
        const int gx = gridDim.x, gy = gridDim.y, T = gx * gy, SG = gridDim.z & ~7;
        const int L3 = blockIdx.x + blockIdx.y * gx + blockIdx.z * T;
        int tile;
        if (L3 < SG * T) {
            const int idx = L3 >> 3;
            zs = (idx / T) * 8 + (L3 & 7);
            tile = idx % T;
        } else {
            const int r = L3 - SG * T;
            zs = SG + r / T;
            tile = r % T;
        }
        mt = tile % gy;
        nt = tile / gy;
    }
    const int m0 = mt * 64 * MW, c0 = nt * NC;
    const int T0 = zs * tiles_per_split, T1 = min(ntiles, T0 + tiles_per_split);
    const int tiles_x = W / 32, tiles_img = tiles_x * (H / TR);
    const long HW = (long)H * W;

    // ---- dY fragment rows of this lane: channel m0 + wr*64 + a*32 + l31 (clamped; rows >= Cout are dropped in the epilogue)
    // buffer addressing (SGPR resource over dY + per-lane byte offset + scalar tile/quad offset): a load costs no VALU
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, dy_bytes, 0x00020000);
    int arow[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) arow[a] = (min(m0 + wr * 64 + a * 32 + l31, Cout - 1) * (int)HW + 4 * lhi) * 4;
    auto tile_org = [&](int T, int& img, int& y0, int& x0) {
        const int Tc = min(T, ntiles - 1);
        img = Tc / tiles_img;
        const int r = Tc - img * tiles_img;
        y0 = (r / tiles_x) * TR;
        x0 = (r % tiles_x) * 32;
    };
    float ra[RING][2][4];
    auto aload = [&](int slot, int tbase, int qd) {        // quad qd (0..QT-1) of the tile at dY element offset tbase
        const int o = __builtin_amdgcn_readfirstlane((tbase + (qd / 4) * W + 8 * (qd % 4)) * 4);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const jp_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(drs, arow[a], o, 0);
            const jp_f32x4 v = __builtin_bit_cast(jp_f32x4, u);
#pragma unroll
            for (int j = 0; j < 4; ++j) ra[slot][a][j] = v[j];
        }
    };

    // ---- patch staging map: the NT/32 half-waves split as PR patch rows x HPR half-waves per row; a half-wave loads the 32
    // centre columns of channels c = hsub + HPR*r (r = 0 .. NROW-1): one row predicate and one base address per thread
    constexpr int HPR = NT / 32 / PR;
    static_assert(HPR * PR * 32 == NT && NC % HPR == 0, "half-waves must tile the patch rows");
    constexpr int NROWS = NC / HPR;
    const int hw = t >> 5, l32 = t & 31;
    const int prw = hw / HPR, hsub = hw % HPR;
    float rb[NROWS], rh[NHALO];
    auto gload = [&](int T) {
        int img, y0, x0;
        tile_org(T, img, y0, x0);
        const float* xc = x + ((long)img * Cx + c0) * HW;
        int yw = y0 - 1 + prw;
        if (REFLECT) yw = jp_reflect(yw, H);
        const bool okw = yw >= 0 && yw < H;
        const float* xr0 = xc + (long)hsub * HW + (long)(okw ? yw : 0) * W + x0 + l32;
#pragma unroll
        for (int r = 0; r < NROWS; ++r) rb[r] = okw ? xr0[(long)(HPR * r) * HW] : 0.f;
        int xl = x0 - 1, xr = x0 + 32;
        if (REFLECT) { xl = jp_reflect(xl, W); xr = jp_reflect(xr, W); }
#pragma unroll
        for (int q = 0; q < NHALO; ++q) {
            const int e = t + NT * q;
            const int side = e & 1, rho = e >> 1;
            const int pr = rho / NC, c = rho % NC;
            int yy = y0 - 1 + pr;
            if (REFLECT) yy = jp_reflect(yy, H);
            const int xx = side ? xr : xl;
            const bool ok = e < 2 * ROWS && yy >= 0 && yy < H && xx >= 0 && xx < W;
            rh[q] = ok ? xc[(long)c * HW + (long)yy * W + xx] : 0.f;
        }
    };
    auto lstore = [&]() {
        float* pd = patch + (prw * PC + 1 + l32) * LDB + hsub;
#pragma unroll
        for (int r = 0; r < NROWS; ++r) pd[HPR * r] = rb[r];
#pragma unroll
        for (int q = 0; q < NHALO; ++q) {
            const int e = t + NT * q;
            const int side = e & 1, rho = e >> 1;
            const int pr = rho / NC, c = rho % NC;
            if (e < 2 * ROWS) patch[(pr * PC + (side ? 33 : 0)) * LDB + c] = rh[q];
        }
    };

    jp_f32x16 acc[2][3];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][j][r] = 0.f;

    // B fragment of k-step (tr, q, j), tap (tyw, tx): pixel (tr + tyw, 8q + 4*lhi + j + tx) of the patch
    const float* bp = patch + (((tyw + kg * TRG) * PC) + 4 * lhi) * LDB + cb * 32 + l31;

    if (T0 < T1) {
        int img, y0, x0;
        tile_org(T0, img, y0, x0);
        int tb = (img * Cout) * (int)HW + (y0 + kg * TRG) * W + x0;      // dY element offset of this K group's rows (channel 0)
#pragma unroll
        for (int d = 0; d < W9_AHEAD; ++d) aload(d, tb, d);
        gload(T0);
        for (int T = T0; T < T1; ++T) {
            lstore();
            __syncthreads();
            gload(T + 1);                                           // next tile's patch: in flight during the MFMAs below
            tile_org(T + 1, img, y0, x0);
            const int tbn = (img * Cout) * (int)HW + (y0 + kg * TRG) * W + x0;
            // B fragments are read one k-step ahead of the MFMAs that use them; all offsets are compile-time (the 64
            // k-steps of a tile are fully unrolled)
            auto boff = [&](int s) -> int {
                const int qd = s / 4, j = s % 4;
                return ((qd / 4) * PC + 8 * (qd % 4) + j) * LDB;
            };
            float b0 = bp[boff(0)], b1 = bp[boff(0) + LDB], b2 = bp[boff(0) + 2 * LDB];
#pragma unroll
            for (int s = 0; s < 4 * QT; ++s) {
                const int qd = s / 4, j = s % 4;
                if (j == 0) {
                    // ring: quad qd lives in slot qd % RING; quad qd + AHEAD (of this or the next tile) is requested now
                    const int qa = qd + W9_AHEAD;
                    if (qa < QT) aload(qa % RING, tb, qa);
                    else aload(qa % RING, tbn, qa - QT);
                }
                const int sn = s + 1 < 4 * QT ? s + 1 : s;
                const float nb0 = bp[boff(sn)], nb1 = bp[boff(sn) + LDB], nb2 = bp[boff(sn) + 2 * LDB];
                const float a0 = ra[qd % RING][0][j], a1 = ra[qd % RING][1][j];
                __builtin_amdgcn_sched_barrier(0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b2, acc[0][2], 0, 0, 0);
                acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b2, acc[1][2], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                b0 = nb0; b1 = nb1; b2 = nb2;
            }
            tb = tbn;
            __syncthreads();
        }
    }

    // ---- partial tile -> ws[zs][m][tap*Cm + ci]; C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const long Np = 9L * Cm;
    float* wz = ws + (long)(zs * KG + kg) * Cout * Np;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const long n = (long)(tyw * 3 + j) * Cm + c0 + cb * 32 + l31;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (m < Cout) wz[(long)m * Np + n] = acc[a][j][r];
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// "W1": the same scheme for 1x1 stride-1 weight gradients (dW[co][ci] = sum_p dY[co][p] * X[ci][p]; the CRP / reduce
// layers).  No taps, so a staged input value is only reused across output channels: the workgroup therefore spans 256
// output channels (4 wave rows) x 128 input channels (2 wave columns of 2 blocks): 8 waves, 2x2 accumulators each, a
// 4x32-pixel tile of the 128 input channels transposed to [pixel][channel] in LDS (no halo), dY fragments from global.
// Output: ws[split][m][ci] (wgrad_reduce4_kernel with one "tap").  Preconditions: Cout % 256 == 0 is NOT required (rows
// are clamped / masked), Cm % 128 == 0, W % 32 == 0, H % 4 == 0.
__global__ __launch_bounds__(512, 2) void jp_wgrad_w1_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            float* __restrict__ ws, int Cout, int Cx, int Cm, int H, int W,
                                                            int ntiles, int tiles_per_split, int dy_bytes) {
    constexpr int NT = 512, TR = 4, NC = 128, LDB = NC + 1, QT = TR * 4;
    constexpr int NROWS = NC * TR * 32 / NT;                          // 32 staged values per thread and tile
    __shared__ float patch[TR * 32 * LDB];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 1, cbw = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    int mt, nt, zs;
    {   // every XCD owns whole K slices, see jp_wgrad_w9_kernel
        const int gx = gridDim.x, gy = gridDim.y, T = gx * gy, SG = gridDim.z & ~7;
        const int L3 = blockIdx.x + blockIdx.y * gx + blockIdx.z * T;
        int tile;
        if (L3 < SG * T) {
            const int idx = L3 >> 3;
            zs = (idx / T) * 8 + (L3 & 7);
            tile = idx % T;
        } else {
            const int r = L3 - SG * T;
            zs = SG + r / T;
            tile = r % T;
        }
        mt = tile % gy;
        nt = tile / gy;
    }
    const int m0 = mt * 256, c0 = nt * NC;
    const int T0 = zs * tiles_per_split, T1 = min(ntiles, T0 + tiles_per_split);
    const int tiles_x = W / 32, tiles_img = tiles_x * (H / TR);
    const long HW = (long)H * W;
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, dy_bytes, 0x00020000);
    int arow[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) arow[a] = (min(m0 + wr * 64 + a * 32 + l31, Cout - 1) * (int)HW + 4 * lhi) * 4;
    auto tile_org = [&](int T, int& img, int& y0, int& x0) {
        const int Tc = min(T, ntiles - 1);
        img = Tc / tiles_img;
        const int r = Tc - img * tiles_img;
        y0 = (r / tiles_x) * TR;
        x0 = (r % tiles_x) * 32;
    };
    float ra[2][2][4];
    auto aload = [&](int slot, int tbase, int qd) {
        const int o = __builtin_amdgcn_readfirstlane((tbase + (qd / 4) * W + 8 * (qd % 4)) * 4);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const jp_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(drs, arow[a], o, 0);
            const jp_f32x4 v = __builtin_bit_cast(jp_f32x4, u);
#pragma unroll
            for (int j = 0; j < 4; ++j) ra[slot][a][j] = v[j];
        }
    };
    // staging: half-wave hw takes tile row hw/4 and channels (hw%4) + 4*r, lanes along the 32 columns
    const int hw = t >> 5, l32 = t & 31;
    const int prw = hw >> 2, hsub = hw & 3;
    float rb[NROWS];
    auto gload = [&](int T) {
        int img, y0, x0;
        tile_org(T, img, y0, x0);
        const float* xr0 = x + ((long)img * Cx + c0 + hsub) * HW + (long)(y0 + prw) * W + x0 + l32;
#pragma unroll
        for (int r = 0; r < NROWS; ++r) rb[r] = xr0[(long)(4 * r) * HW];
    };
    auto lstore = [&]() {
        float* pd = patch + (prw * 32 + l32) * LDB + hsub;
#pragma unroll
        for (int r = 0; r < NROWS; ++r) pd[4 * r] = rb[r];
    };
    jp_f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][j][r] = 0.f;
    const float* bp = patch + (4 * lhi) * LDB + cbw * 64 + l31;
    if (T0 < T1) {
        int img, y0, x0;
        tile_org(T0, img, y0, x0);
        int tb = (img * Cout) * (int)HW + y0 * W + x0;
        aload(0, tb, 0);
        gload(T0);
        for (int T = T0; T < T1; ++T) {
            lstore();
            __syncthreads();
            gload(T + 1);
            tile_org(T + 1, img, y0, x0);
            const int tbn = (img * Cout) * (int)HW + y0 * W + x0;
            auto boff = [&](int s) -> int {
                const int qd = s / 4, j = s % 4;
                return ((qd / 4) * 32 + 8 * (qd % 4) + j) * LDB;
            };
            float b0 = bp[boff(0)], b1 = bp[boff(0) + 32];
#pragma unroll
            for (int s = 0; s < 4 * QT; ++s) {
                const int qd = s / 4, j = s % 4;
                if (j == 0) {
                    const int qa = qd + 1;
                    if (qa < QT) aload(qa % 2, tb, qa);
                    else aload(qa % 2, tbn, qa - QT);
                }
                const int sn = s + 1 < 4 * QT ? s + 1 : s;
                const float nb0 = bp[boff(sn)], nb1 = bp[boff(sn) + 32];
                const float a0 = ra[qd % 2][0][j], a1 = ra[qd % 2][1][j];
                __builtin_amdgcn_sched_barrier(0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                b0 = nb0; b1 = nb1;
            }
            tb = tbn;
            __syncthreads();
        }
    }
    float* wz = ws + (long)zs * Cout * Cm;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const long n = c0 + cbw * 64 + j * 32 + l31;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (m < Cout) wz[(long)m * Cm + n] = acc[a][j][r];
            }
    }
}
