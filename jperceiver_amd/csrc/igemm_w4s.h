// "W4S" patch kernel: WEIGHT GRADIENT of the UPSAMPLED segment of an iconv layer in parity-class form, on the bf16 matrix
// pipe (split products, igemm_p9s.h; machinery of igemm_w9s.h).  With y = Conv3x3_reflect(cat(.., up2x(x), ..)):
//     dW'_{(a,b)}[(r,s)][co][c] = sum over half-resolution pixels (i', j') of
//                                 dY[co][2i'+a][2j'+b] * X[c][clamp(i'-1+a+r)][clamp(j'-1+b+s)]
// (16 (class, slot) GEMMs with M = co, N = c, K = half-resolution pixels; wgrad_fold_parity_kernel then adds every (class,
// slot) plane into the 3x3 taps it contains).  One MFMA K group = 16 consecutive j' of one row i'.
//   A (dY): a lane's 8 pixels of BOTH column classes b = 0, 1 are 16 consecutive full-resolution floats of row 2i'+a: four
//     16-byte buffer loads, de-interleaved and split in registers (two A operands per K group).
//   B (X): the (TR+2) x 34 half-resolution patch (edge clamp resolved while staging, every value split once) in the W9S layout
//     [split][32-channel block][patch pixel][32 channels], read with the transpose read ds_read_b64_tr_b16.  Slot (r, s) of
//     class (a, b) is the patch shifted by (a + r, b + s): only 6 distinct fragments serve the 8 (b, r, s) accumulators.
// Workgroup: 8 waves = 4 blocks of 32 output x 2 blocks of 32 input channels, ONE row class a per workgroup (grid.z = 2 x
// K splits); a wave owns the 8 accumulators (b, r, s) of its block pair: 48 MFMAs per K group for 4 dY loads + 36 reads.
// Output: ws[split][co][q*Cx + c], q = ((a*2 + b)*4 + r*2 + s)  (the layout WgradAP/BP + WgradEpiWS produce).
// Preconditions (host-checked): w2 % 32 == 0, h2 % TR == 0, Cx % 64 == 0.
#pragma once
#include "igemm_w9s.h"

template <int TR>
__global__ __launch_bounds__(512, 2) void jp_wgrad_w4s_kernel(const float* __restrict__ dy, const float* __restrict__ xh,
                                                             float* __restrict__ ws, int Cout, int Cx, int h2, int w2,
                                                             int ntiles, int tiles_per_split, int dy_bytes,
                                                             const float* __restrict__ gam, const float* __restrict__ xam) {
    constexpr int NS = JP_NS;
    float gsc = 1.f, xsc = 1.f, osc = 1.f;       // JP_NS == 2: operand scales, see jp_wgrad_w9s_kernel
    if constexpr (NS == 2) {
        const int kg_ = __builtin_amdgcn_readfirstlane(jp_scale_exp(jp_slot_amax(gam))), kx_ = __builtin_amdgcn_readfirstlane(jp_scale_exp(jp_slot_amax(xam)));
        gsc = jp_exp2i(kg_);
        xsc = jp_exp2i(kx_);
        osc = jp_exp2i(-kg_) * jp_exp2i(-kx_);
    }
    constexpr int NT = 512, PR = TR + 2, PC = 34;
    constexpr int SLOTS = PR * PC, CBP = SLOTS * 64, SPL = 2 * CBP;
    constexpr int ITEMS = SLOTS * 16, NQ = (ITEMS + NT - 1) / NT;
    constexpr int KGR = TR * 2;
    static_assert(2 * SPL + ((TR + 1) * PC + 18) * 64 + 256 < 65536, "transpose-read immediates must fit 16 bits");
    __shared__ __attribute__((aligned(16))) unsigned char patch[NS * SPL];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ab = wave & 3, cb = wave >> 2;
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
    const int ca = zs & 1, split = zs >> 1;                   // row class a of this workgroup, K split
    const int m0 = mt * 128, c0 = nt * 64;
    const int T0 = split * tiles_per_split, T1 = min(ntiles, T0 + tiles_per_split);
    const int tiles_x = w2 / 32, tiles_img = tiles_x * (h2 / TR);
    const int H = 2 * h2, W = 2 * w2;
    const long HW = (long)H * W, hw2 = (long)h2 * w2;
    auto tile_org = [&](int T, int& img, int& i0, int& j0) {
        const int Tc = min(T, ntiles - 1);
        img = Tc / tiles_img;
        const int r = Tc - img * tiles_img;
        i0 = (r / tiles_x) * TR;
        j0 = (r % tiles_x) * 32;
    };

    // ---- A: 16 consecutive full-resolution floats of row 2i'+a per lane and K group (both column classes)
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, dy_bytes, 0x00020000);
    const int arow = (min(m0 + ab * 32 + l31, Cout - 1) * (int)HW + 16 * lhi) * 4;
    jp_u32x4 araw[2][4];
    auto aload = [&](int slot, int tbase, int g) {            // tbase: dY element offset of (img, channel 0, row 2*i0, column 2*j0)
        const int so = __builtin_amdgcn_readfirstlane((tbase + (2 * (g / 2) + ca) * W + 32 * (g % 2)) * 4);
#pragma unroll
        for (int v = 0; v < 4; ++v) araw[slot][v] = __builtin_amdgcn_raw_buffer_load_b128(drs, arow + 16 * v, so, 0);
    };

    // ---- B: transpose-read bases as in W9S, tx = b + s in 0..2, ty = a + r (a through a wave-uniform byte offset)
    const int rr = (lane & 15) >> 2, Qq = 4 * ((lane >> 4) & 1) + (lane & 3);
    int bbase[3][2];
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd)
            bbase[tx][rd] = cb * CBP + (ca * PC + 8 * lhi + rr) * 64 + ((Qq ^ ((tx + 4 * rd + rr) & 7)) * 8);
    auto bread = [&](int r, int tx, int g, int s) -> jp_u32x4 {
        const int imm = s * SPL + ((g / 2 + r) * PC + 16 * (g % 2) + tx) * 64;
        const jp_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (jp_s16x4 __attribute__((address_space(3)))*)(patch + bbase[tx][0] + imm));
        const jp_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (jp_s16x4 __attribute__((address_space(3)))*)(patch + bbase[tx][1] + imm + 256));
        const jp_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(jp_u32x4, v);
    };

    // ---- staging (edge clamp): item e = t + NT*q -> (patch column, patch row, channel quad Qd)
    float rv[NQ][4];
    auto gload = [&](int T) {
        int img, i0, j0;
        tile_org(T, img, i0, j0);
        const float* xc = xh + ((long)img * Cx + c0) * hw2;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = t + NT * q;
            const int pcol = e % PC, rest = e / PC, prow = rest % PR, Qd = rest / PR;
            const int yy = min(max(i0 - 1 + prow, 0), h2 - 1), xx = min(max(j0 - 1 + pcol, 0), w2 - 1);
            const bool ok = e < ITEMS;
            const float* p = xc + (long)(4 * Qd) * hw2 + (ok ? (long)yy * w2 + xx : 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) rv[q][k] = ok ? p[(long)k * hw2] : 0.f;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = t + NT * q;
            if (e >= ITEMS) continue;
            const int pcol = e % PC, rest = e / PC, prow = rest % PR, Qd = rest / PR;
            const int off = (Qd >> 3) * CBP + (prow * PC + pcol) * 64 + (((Qd & 7) ^ (pcol & 7)) * 8);
            unsigned a[3], b[3];
            jp_split_ns(rv[q][0], rv[q][1], xsc, a);
            jp_split_ns(rv[q][2], rv[q][3], xsc, b);
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u2*>(patch + off) = u2{a[0], b[0]};
            *reinterpret_cast<u2*>(patch + SPL + off) = u2{a[1], b[1]};
            if constexpr (NS == 3) *reinterpret_cast<u2*>(patch + 2 * SPL + off) = u2{a[2], b[2]};
        }
    };

    jp_f32x16 acc[2][2][2];                                    // [b][r][s]
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j >> 2][(j >> 1) & 1][j & 1][r] = 0.f;

    if (T0 < T1) {
        int img, i0, j0;
        tile_org(T0, img, i0, j0);
        int tb = (img * Cout) * (int)HW + 2 * i0 * W + 2 * j0;
        aload(0, tb, 0);
        gload(T0);
        for (int T = T0; T < T1; ++T) {
            lstore();
            __syncthreads();
            gload(T + 1);
            tile_org(T + 1, img, i0, j0);
            const int tbn = (img * Cout) * (int)HW + 2 * i0 * W + 2 * j0;
#pragma unroll
            for (int g = 0; g < KGR; ++g) {
                if (g + 1 < KGR) aload((g + 1) & 1, tb, g + 1);
                else aload((g + 1) & 1, tbn, 0);
                // de-interleave the 16 floats by column class and split: sa[b][split]
                jp_u32x4 sa[2][3];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const jp_u32x4 v = araw[g & 1][k];           // full-res columns 4k .. 4k+3: classes b = 0, 1, 0, 1
                    unsigned sq[3];
                    jp_split_ns(__uint_as_float(v[0]), __uint_as_float(v[2]), gsc, sq);
                    sa[0][0][k] = sq[0]; sa[0][1][k] = sq[1]; sa[0][2][k] = sq[2];
                    jp_split_ns(__uint_as_float(v[1]), __uint_as_float(v[3]), gsc, sq);
                    sa[1][0][k] = sq[0]; sa[1][1][k] = sq[1]; sa[1][2][k] = sq[2];
                }
                // the 6 distinct B fragments (r, tx = b + s), read one ahead of their MFMAs
                jp_u32x4 bq[2][3];
#pragma unroll
                for (int s_ = 0; s_ < NS; ++s_) bq[0][s_] = bread(0, 0, g, s_);
#pragma unroll
                for (int f = 0; f < 6; ++f) {
                    const int r = f / 3, tx = f % 3;
                    if (f + 1 < 6) {
#pragma unroll
                        for (int s_ = 0; s_ < NS; ++s_) bq[(f + 1) & 1][s_] = bread((f + 1) / 3, (f + 1) % 3, g, s_);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int s = tx - b;
                        if (s < 0 || s > 1) continue;
                        jp_f32x16& c = acc[b][r][s];
#define JP_W4S_MFMA(SA_, SB_) c = jp_mfma_bf16_sw<false>(sa[b][SA_], bq[f & 1][SB_], c)
                        JP_SPLIT_PRODUCTS(JP_W4S_MFMA);
#undef JP_W4S_MFMA
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            tb = tbn;
            __syncthreads();
        }
    }

    // ---- partial tile -> ws[split][m][q*Cx + ci]; C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const long Np = 16L * Cx;
    float* wz = ws + (long)split * Cout * Np;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int b = j >> 2, r_ = (j >> 1) & 1, s = j & 1;
        const long n = (long)(((ca * 2 + b) * 4) + r_ * 2 + s) * Cx + c0 + cb * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + ab * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (m < Cout) wz[(long)m * Np + n] = NS == 2 ? acc[b][r_][s][r] * osc : acc[b][r_][s][r];
        }
    }
}
