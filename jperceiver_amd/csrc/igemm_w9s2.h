// "W9S2" patch kernel: WEIGHT GRADIENT of a 3x3 STRIDE-2 pad-1 (zero padding) convolution on the bf16 matrix pipe -- the
// ResNet-18 stage transitions (resnet.py:6-8 conv3x3 with stride 2: 64 -> 128 at 256^2, 128 -> 256 at 128^2, 256 -> 512 at 64^2),
// the stride-2 twin of igemm_w9s.h and with the same arithmetic (igemm_p9s.h: fp32 in / out / accumulate, every fp32 product
// formed from six bf16 products of exact three-way splits).  Until round 4 these ran on the generic exact-fp32 engine
// (WgradAS / WgradBUS<3>, ~90 TF).
//     dW[co][ci][ty][tx] = sum over output pixels (oy, ox) of  dY[co][oy][ox] * Xpad[ci][2 oy + ty - 1][2 ox + tx - 1]
// Per tap a GEMM with M = Cout, N = Cin, K = output pixels; an MFMA's K group is 16 consecutive output pixels of one row.
//   A (dY) straight from global memory, split in registers (as W9S).
//   B (X): the (2 TR + 1) x 65 input patch of a TR x 32 output-pixel tile is staged DE-INTERLEAVED into its four parity planes
//     [row parity a][column parity b][i][j] (patch pixel (2 i + a, 2 j + b)), each value split once, 64 bytes per pixel
//     (32 channels) per split.  Tap (ty, tx) of output pixel (py, px) is patch pixel (2 py + ty, 2 px + tx) = plane
//     (ty & 1, tx & 1) pixel (py + (ty >> 1), px + (tx >> 1)): inside a plane the 16 pixels of a K group are CONSECUTIVE again,
//     so the same LDS transpose reads (`ds_read_b64_tr_b16`) at compile-time offsets serve every tap.  Channel quads are
//     XOR-swizzled with (plane column & 7) exactly as in W9S.
// Workgroup: 8 waves, 32 input channels x (KG = 1: 256 | KG = 2: 128) output channels; with KG = 2 the two wave groups take
// the two pixel rows of the tile and write their own partial slice.  A wave owns all 9 taps of its 32 x 32 block.
// Output: split-K partial sums ws[slice][m][tap*Cm + ci] (wgrad_reduce4_kernel folds them), as W9S.
// Preconditions (host-checked): H, W even; OW = W/2 a multiple of 32; OH = H/2 a multiple of 2; Cm % 32 == 0; tensors < 2 GiB.
#pragma once
#include "igemm_w9s.h"

template <int KG>
__global__ __launch_bounds__(512, 2) void jp_wgrad_w9s2_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              float* __restrict__ ws, int Cout, int Cx, int Cm, int H, int W,
                                                              int ntiles, int tiles_per_split, int dy_bytes, int x_bytes,
                                                              const float* __restrict__ gam, const float* __restrict__ xam) {
    constexpr int NS = JP_NS;
    float gsc = 1.f, xsc = 1.f, osc = 1.f;       // JP_NS == 2: operand scales, see jp_wgrad_w9s_kernel
    if constexpr (NS == 2) {
        const int kg_ = __builtin_amdgcn_readfirstlane(jp_scale_exp(jp_slot_amax(gam))), kx_ = __builtin_amdgcn_readfirstlane(jp_scale_exp(jp_slot_amax(xam)));
        gsc = jp_exp2i(kg_);
        xsc = jp_exp2i(kx_);
        osc = jp_exp2i(-kg_) * jp_exp2i(-kx_);
    }
    constexpr int NT = 512, TR = 2;
    constexpr int PRX = 2 * TR + 1, PCX = 65;      // patch rows / columns in input pixels
    constexpr int PRP = TR + 1, PCP = 34;          // rows / row pitch of one parity plane
    constexpr int PSL = PRP * PCP;                 // slots per plane
    constexpr int SPL = 4 * PSL * 64;              // bytes per split (32 channels = 64 bytes per pixel)
    constexpr int ITEMS = PRX * PCX * 8, NQ = (ITEMS + NT - 1) / NT;   // (patch pixel, channel quad) items, rounds per thread
    constexpr int KGR = TR * 2;                    // K groups (16 output pixels) per tile
    constexpr int MB = 8 / KG, KGW = KGR / KG;     // 32-channel output blocks per M tile; K groups per wave and tile
    static_assert(KG == 1 || KG == 2, "one or two K groups");
    __shared__ __attribute__((aligned(16))) unsigned char patch[NS * SPL];

    const int OH = H >> 1, OW = W >> 1;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ab = wave % MB, kg = wave / MB;
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
    const int m0 = mt * 32 * MB, c0 = nt * 32;
    const int T0 = zs * tiles_per_split, T1 = min(ntiles, T0 + tiles_per_split);
    const int tiles_x = OW / 32, tiles_img = tiles_x * (OH / TR);
    const long HW = (long)H * W;
    const int OHW = OH * OW;
    auto tile_org = [&](int T, int& img, int& y0, int& x0) {      // output-pixel origin of tile T
        const int Tc = min(T, ntiles - 1);
        img = Tc / tiles_img;
        const int r = Tc - img * tiles_img;
        y0 = (r / tiles_x) * TR;
        x0 = (r % tiles_x) * 32;
    };

    // ---- A: dY rows of this lane (channel clamped; rows >= Cout are dropped in the epilogue), 8 pixels per K group
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, dy_bytes, 0x00020000);
    const int arow = (min(m0 + ab * 32 + l31, Cout - 1) * OHW + 8 * lhi) * 4;
    jp_u32x4 araw[2][2];
    auto aload = [&](int slot, int tbase, int g) {          // K group g of the tile whose dY element offset (channel 0) is tbase
        const int so = __builtin_amdgcn_readfirstlane((tbase + (g / 2) * OW + 16 * (g % 2)) * 4);
        araw[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(drs, arow, so, 0);
        araw[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(drs, arow + 16, so, 0);
    };

    // ---- B: per-lane byte bases of the transpose reads (igemm_w9s.h): plane column = 16*(K-group half) + (tx >> 1) + 4*rd +
    // 8*lhi + r  ->  swizzle mask ((tx >> 1) + 4*rd + r) & 7
    const int rr = (lane & 15) >> 2, Qq = 4 * ((lane >> 4) & 1) + (lane & 3);
    int bbase[2][2];
#pragma unroll
    for (int txs = 0; txs < 2; ++txs)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd)
            bbase[txs][rd] = (kg * (KGW / 2) * PCP + 8 * lhi + rr) * 64 + ((Qq ^ ((txs + 4 * rd + rr) & 7)) * 8);
    auto bread = [&](int ty, int tx, int g, int s) -> jp_u32x4 {
        const int plane = (ty & 1) * 2 + (tx & 1);
        const int imm = s * SPL + (plane * PSL + (g / 2 + (ty >> 1)) * PCP + 16 * (g % 2) + (tx >> 1)) * 64;
        const jp_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (jp_s16x4 __attribute__((address_space(3)))*)(patch + bbase[tx >> 1][0] + imm));
        const jp_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (jp_s16x4 __attribute__((address_space(3)))*)(patch + bbase[tx >> 1][1] + imm + 256));
        const jp_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(jp_u32x4, v);
    };

    // ---- staging: item e = t + NT*q -> (patch column, patch row, channel quad); lanes run along the patch columns (a wave's
    // 64 lanes read 64 consecutive input pixels of one channel: coalesced), each item = 4 channels of one pixel -> three
    // 8-byte LDS words in the pixel's parity plane
    float rv[NQ][4];
    int ipos[NQ], ilds[NQ];              // (patch column - 1) | (patch row - 1) << 16;  LDS byte offset, -1: no item
    unsigned iq[NQ];                     // byte offset of the item's first channel inside the (image, channel block) slab
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = t + NT * q;
        const int pcol = e % PCX, rest = e / PCX, prow = rest % PRX, Qd = rest / PRX;
        const int plane = (prow & 1) * 2 + (pcol & 1), pi = prow >> 1, pj = pcol >> 1;
        ipos[q] = ((pcol - 1) & 0xffff) | ((prow - 1) << 16);
        ilds[q] = e < ITEMS ? (plane * PSL + pi * PCP + pj) * 64 + ((Qd ^ (pj & 7)) * 8) : -1;
        iq[q] = (unsigned)(4 * Qd) * (unsigned)HW * 4u;
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
    auto gload = [&](int T) {
        int img, y0, x0;
        tile_org(T, img, y0, x0);
        const long slab = ((long)img * Cx + c0) * HW * 4;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int yy = 2 * y0 + (ipos[q] >> 16), xx = 2 * x0 + (int)(short)(ipos[q] & 0xffff);
            const bool ok = ilds[q] >= 0 && yy >= 0 && yy < H && xx >= 0 && xx < W;
            const unsigned lo = ok ? iq[q] + (unsigned)(yy * W + xx) * 4u : 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ub = __builtin_amdgcn_readfirstlane((int)(slab + (long)k * HW * 4));
                const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, lo, ub, 0));
                rv[q][k] = ok ? v : 0.f;
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (ilds[q] < 0) continue;
            unsigned a[3], b[3];
            jp_split_ns(rv[q][0], rv[q][1], xsc, a);
            jp_split_ns(rv[q][2], rv[q][3], xsc, b);
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u2*>(patch + ilds[q]) = u2{a[0], b[0]};
            *reinterpret_cast<u2*>(patch + SPL + ilds[q]) = u2{a[1], b[1]};
            if constexpr (NS == 3) *reinterpret_cast<u2*>(patch + 2 * SPL + ilds[q]) = u2{a[2], b[2]};
        }
    };

    jp_f32x16 acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    if (T0 < T1) {
        int img, y0, x0;
        tile_org(T0, img, y0, x0);
        int tb = (img * Cout) * OHW + y0 * OW + x0;
        aload(0, tb, kg * KGW);
        gload(T0);
        for (int T = T0; T < T1; ++T) {
            lstore();
            __syncthreads();
            gload(T + 1);                                           // next tile's patch: in flight during the MFMAs below
            tile_org(T + 1, img, y0, x0);
            const int tbn = (img * Cout) * OHW + y0 * OW + x0;
#pragma unroll
            for (int gi = 0; gi < KGW; ++gi) {
                const int g = gi;                                   // (the kg part of the K group sits in the read bases)
                if (gi + 1 < KGW) aload((gi + 1) & 1, tb, kg * KGW + gi + 1);
                else aload((gi + 1) & 1, tbn, kg * KGW);
                jp_u32x4 sa[3];
                {
                    const jp_u32x4 lo = araw[gi & 1][0], hi = araw[gi & 1][1];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        unsigned sq[3];
                        jp_split_ns(__uint_as_float(lo[2 * k]), __uint_as_float(lo[2 * k + 1]), gsc, sq);
                        sa[0][k] = sq[0]; sa[1][k] = sq[1]; sa[2][k] = sq[2];
                        jp_split_ns(__uint_as_float(hi[2 * k]), __uint_as_float(hi[2 * k + 1]), gsc, sq);
                        sa[0][2 + k] = sq[0]; sa[1][2 + k] = sq[1]; sa[2][2 + k] = sq[2];
                    }
                }
                jp_u32x4 bq[2][3];
#pragma unroll
                for (int s_ = 0; s_ < NS; ++s_) bq[0][s_] = bread(0, 0, g, s_);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    if (tap + 1 < 9) {
#pragma unroll
                        for (int s_ = 0; s_ < NS; ++s_) bq[(tap + 1) & 1][s_] = bread((tap + 1) / 3, (tap + 1) % 3, g, s_);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#define JP_W9S2_MFMA(SA_, SB_) acc[tap] = jp_mfma_bf16_sw<false>(sa[SA_], bq[tap & 1][SB_], acc[tap])
                    JP_SPLIT_PRODUCTS(JP_W9S2_MFMA);
#undef JP_W9S2_MFMA
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            tb = tbn;
            __syncthreads();
        }
    }

    // ---- partial tile -> ws[slice][m][tap*Cm + ci]; C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const long Np = 9L * Cm;
    float* wz = ws + (long)(zs * KG + kg) * Cout * Np;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const long n = (long)tap * Cm + c0 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + ab * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (m < Cout) wz[(long)m * Np + n] = NS == 2 ? acc[tap][r] * osc : acc[tap][r];
        }
    }
}
