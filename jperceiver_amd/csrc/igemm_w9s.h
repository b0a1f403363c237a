// "W9S" patch kernel: WEIGHT GRADIENT of a 3x3 stride-1 pad-1 convolution with every fp32 product formed on the BF16 matrix
// pipe from three-way splits of both operands (igemm_p9s.h has the arithmetic argument: fp32 in / out / accumulate,
// fp32-equivalent accuracy, 6 MFMAs of 32 cycles instead of 8 of 64).
//     dW[co][ci][ty][tx] = sum over pixels p of  dY[co][p] * Xpad[ci][p + (ty-1, tx-1)]
// Per tap a GEMM with M = Cout, N = Cin and K = pixels; one `v_mfma_f32_32x32x16_bf16` takes 16 consecutive pixels of one
// image row as its K group, a lane holding the 8 pixels 8*(lane>>5) .. +7 of its row (A: output channel) / column (B: input
// channel).
//   A (dY) comes STRAIGHT FROM GLOBAL MEMORY: a lane's 8 pixels are two aligned 16-byte buffer loads of its channel row
//     (one K group ahead); they are split in registers (~44 VALU operations per 54 MFMAs, hidden behind the other wave of
//     the SIMD).
//   B (X) is staged once per pixel tile -- padding resolved, each value split ONCE -- as bf16 [split][32-channel block]
//     [patch pixel][32 channels] (64 bytes per pixel), and read with the LDS TRANSPOSE read `ds_read_b64_tr_b16`: 16 lanes
//     fetch a [4 pixels][16 channels] block and each receives the 4 pixels of ITS channel, so a tap shift is just another
//     patch pixel (an immediate offset) and no element is ever re-laid out per tap.  The 8-byte channel quads of a pixel
//     row are XOR-swizzled with (patch column & 7): conflict-free transpose reads, 2-way conflicts on the (rare) writes.
// Workgroup: 8 waves = 4 blocks of 32 output channels x 2 blocks of 32 input channels; a wave owns ALL 9 taps of its
// (32 x 32) block pair: nine 32x32 accumulators, 54 MFMAs per K group for 2 dY loads and 54 transpose reads.
// Output: split-K partial sums ws[split][m][tap*Cm + ci] (wgrad_reduce kernels fold them into dW), as igemm_w9.h.
// Preconditions (host-checked): W % 32 == 0, H % TR == 0, Cm % 64 == 0.
#pragma once
#include "igemm_p9s.h"

typedef short jp_s16x4 __attribute__((ext_vector_type(4)));
typedef short jp_s16x8 __attribute__((ext_vector_type(8)));

// KG = 1: M tile = 128 output channels (4 blocks).  KG = 2 (layers with <= 64 output channels): M tile = 64 channels and the
// two wave groups take different pixel rows of the tile (K groups), writing their own partial slice 2*split + kg.
// W9S_DB (round 4): the patch is double-buffered in LDS.  The split + LDS store of tile T + 1's patch used to sit between two
// barriers with no MFMA in flight (timing probe: staging = 12.6 % of the kernel, profiles/r04_w9s_probes.log); now it is issued
// item by item between the taps of tile T's LAST K group, underneath its MFMAs, and a tile needs ONE barrier.
#ifndef W9S_DB
#define W9S_DB 1
#endif
constexpr int W9S_LA = 2;  // taps of look-ahead of the transpose reads in the three-product build (1: as the six-product stream; < 1 % either way, the switch is gone)
// NCB = 32-channel INPUT blocks per workgroup: 2 (128 output x 64 input channels, rounds 3) or 1 (256 output x 32 input channels,
// round 4).  The kernel is VALU-issue-sensitive (timing probes, profiles/r04_w9s_probes.log: dropping the 44-VALU dY split per K
// group is worth 8 %, dropping the patch staging 12.6 %) and the patch staging -- address arithmetic + split of every staged X
// value -- is its largest VALU item; with one input block per workgroup a staged value serves 256 output channels instead of 128,
// i.e. half the staging instructions per MFMA, and no dY row is split twice.
template <int TR, bool REFLECT, int KG = 1, int NCB = 2>
__global__ __launch_bounds__(512, 2) void jp_wgrad_w9s_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             float* __restrict__ ws, int Cout, int Cx, int Cm, int H, int W,
                                                             int ntiles, int tiles_per_split, int dy_bytes, int x_bytes,
                                                             const float* __restrict__ gam, const float* __restrict__ xam) {
    constexpr int NS = JP_NS;
    // JP_NS == 2: power-of-two scales of dY and X from their largest magnitudes (scale.hip); the sums are scaled back on the way out
    float gsc = 1.f, xsc = 1.f, osc = 1.f;
    if constexpr (NS == 2) {
        const int kg_ = __builtin_amdgcn_readfirstlane(jp_scale_exp(jp_slot_amax(gam))), kx_ = __builtin_amdgcn_readfirstlane(jp_scale_exp(jp_slot_amax(xam)));
        gsc = jp_exp2i(kg_);
        xsc = jp_exp2i(kx_);
        osc = jp_exp2i(-kg_) * jp_exp2i(-kx_);
    }
    constexpr int NT = 512, PR = TR + 2, PC = 34;
    constexpr int SLOTS = PR * PC;                 // patch pixels
    constexpr int CBP = SLOTS * 64;                // bytes per (split, channel block) plane
    constexpr int SPL = NCB * CBP;                 // bytes per split
    constexpr int ITEMS = SLOTS * 8 * NCB, NQ = (ITEMS + NT - 1) / NT;  // (pixel, channel quad) items, rounds per thread
    constexpr int KGR = TR * 2;                    // K groups (16 pixels) per tile
    constexpr int MB = 8 / (KG * NCB), KGW = KGR / KG;  // 32-channel output blocks per M tile; K groups per wave and tile
    static_assert(KG == 1 || KG == 2, "one or two K groups");
    static_assert(NCB == 1 || NCB == 2, "one or two input-channel blocks");
    static_assert(2 * SPL + ((TR + 1) * PC + 18) * 64 + 256 < 65536, "transpose-read immediates must fit 16 bits");
    static_assert(KGW % 2 == 0, "a wave's K groups per tile: whole pixel rows, and an even count (operand / fragment set parities)");
    constexpr bool DB = W9S_DB != 0;
    constexpr int BUFB = NS * SPL;                 // bytes per patch buffer
    __shared__ __attribute__((aligned(16))) unsigned char patch[(DB ? 2 : 1) * BUFB];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ab = wave % MB, cb = (wave / MB) % NCB, kg = wave / (NCB * MB);
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
    const int m0 = mt * 32 * MB, c0 = nt * 32 * NCB;
    const int T0 = zs * tiles_per_split, T1 = min(ntiles, T0 + tiles_per_split);
    const int tiles_x = W / 32, tiles_img = tiles_x * (H / TR);
    const long HW = (long)H * W;
    auto tile_org = [&](int T, int& img, int& y0, int& x0) {
        const int Tc = min(T, ntiles - 1);
        img = Tc / tiles_img;
        const int r = Tc - img * tiles_img;
        y0 = (r / tiles_x) * TR;
        x0 = (r % tiles_x) * 32;
    };

    // ---- A: dY rows of this lane (channel clamped; rows >= Cout are dropped in the epilogue), 8 pixels per K group
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, dy_bytes, 0x00020000);
    const int arow = (min(m0 + ab * 32 + l31, Cout - 1) * (int)HW + 8 * lhi) * 4;
    jp_u32x4 araw[2][2];
    auto aload = [&](int slot, int tbase, int g) {          // K group g of the tile whose dY element offset (channel 0) is tbase
        const int so = __builtin_amdgcn_readfirstlane((tbase + (g / 2) * W + 16 * (g % 2)) * 4);
        araw[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(drs, arow, so, 0);
        araw[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(drs, arow + 16, so, 0);
    };

    // ---- B: per-lane byte bases of the transpose reads.  Lane (g16 = lane>>4, r = (lane&15)>>2, q = lane&3) addresses pixel row
    // r of the [4 pixels][16 channels] block of channel half g16 & 1; its channel quad Q = 4*(g16&1) + q sits at position
    // Q ^ (patch column & 7), patch column = 16*(K-group half) + tx + 4*rd + 8*lhi + r  ->  mask (tx + 4*rd + r) & 7
    const int rr = (lane & 15) >> 2, Qq = 4 * ((lane >> 4) & 1) + (lane & 3);
    int bbase[3][2];
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd)
            bbase[tx][rd] = cb * CBP + (kg * (KGW / 2) * PC + 8 * lhi + rr) * 64 + ((Qq ^ ((tx + 4 * rd + rr) & 7)) * 8);
    int rbuf = 0;                                  // byte offset of the buffer the MFMAs read (0 | BUFB), toggled per tile
    // ---- staging: item e = t + NT*q -> (patch column, patch row, channel quad Qd of the 64 channels); lanes run along
    // the patch columns (coalesced loads), each item = 4 channels of one pixel -> three 8-byte LDS words
    // Everything about an item that does not depend on the tile is decoded ONCE (round 4): its patch position (packed), its LDS
    // byte offset and its channel-quad offset; a tile then costs an item two adds, the border handling and one multiply-add,
    // and its four channel loads are buffer loads (SGPR resource of the whole X tensor + one per-lane 32-bit offset + a
    // wave-uniform byte offset per channel: no 64-bit address arithmetic in vector registers).
    float rv[NQ][4];
    int ipos[NQ], ilds[NQ];              // (patch column - 1) | (patch row - 1) << 16;  LDS byte offset, -1: no item
    unsigned iq[NQ];                     // byte offset of the item's first channel inside the (image, channel block) slab
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = t + NT * q;
        const int pcol = e % PC, rest = e / PC, prow = rest % PR, Qd = rest / PR;
        ipos[q] = ((pcol - 1) & 0xffff) | ((prow - 1) << 16);
        ilds[q] = e < ITEMS ? (Qd >> 3) * CBP + (prow * PC + pcol) * 64 + (((Qd & 7) ^ (pcol & 7)) * 8) : -1;
        iq[q] = (unsigned)(4 * Qd) * (unsigned)HW * 4u;
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
    auto gload = [&](int T) {
        int img, y0, x0;
        tile_org(T, img, y0, x0);
        const long slab = ((long)img * Cx + c0) * HW * 4;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            int yy = y0 + (ipos[q] >> 16), xx = x0 + (int)(short)(ipos[q] & 0xffff);
            if (REFLECT) { yy = jp_reflect(yy, H); xx = jp_reflect(xx, W); }
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
    auto lstore1 = [&](int q, int wbuf) {           // item q of the staged patch -> buffer at byte offset wbuf
        if (ilds[q] < 0) return;
        const int off = wbuf + ilds[q];
        unsigned a[3], b[3];
        jp_split_ns(rv[q][0], rv[q][1], xsc, a);
        jp_split_ns(rv[q][2], rv[q][3], xsc, b);
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u2*>(patch + off) = u2{a[0], b[0]};
        *reinterpret_cast<u2*>(patch + SPL + off) = u2{a[1], b[1]};
        if constexpr (NS == 3) *reinterpret_cast<u2*>(patch + 2 * SPL + off) = u2{a[2], b[2]};
    };
    auto lstore = [&](int wbuf) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) lstore1(q, wbuf);
    };

    jp_f32x16 acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // Round 5: the instruction stream is laid out for an in-order wave that shares its SIMD's matrix pipe with ONE partner which the
    // arbiter serves strictly by age (tools/ubench/mfma_lone_wave.hip: of two waves with MFMAs ready the older issues all of its
    // own first).  A wave that stops issuing MFMAs to run a burst of other work leaves the pipe to its partner only if the partner
    // has operands ready, so nothing here is a burst any more:
    //   * dY of K group g + 2 is requested at the start of group g (the raw registers of group g are dead: it was split during
    //     group g - 1) and group g + 1 is split into the OTHER operand set in twelve pieces of <= 5 VALU instructions, one behind
    //     every fourth MFMA of group g -- the 44-VALU block that used to open every K group is gone;
    //   * the six transpose reads of tap t + 1 go out one behind each of the six MFMAs of tap t;
    //   * the next tile's patch items keep their place behind the taps of the tile's last K group (W9S_DB).
    // Same products in the same order: results are bit-identical to the round-4 stream.
    jp_u32x4 sa[2][NS];                                  // [K group parity][split]: 16-bit operand sets
    constexpr int NPIECE = 4 * NS;                       // pieces of one K group's dY split: (pair, phase)
    constexpr int NPROD = NS == 2 ? 3 : 6;               // matrix products per fp32 product
    float sr[2];                                         // residuals of the pair being split (between two pieces)
    // piece c = 3 * pair + phase of the split of raw set `rs` into operand set `ds`; pair p = floats 2p, 2p + 1 of the 8 pixels
    auto split_piece = [&](int rs, int ds, int c) {
        const int pr = c / NS, ph = c % NS;
        const jp_u32x4& src = araw[rs][pr >> 1];
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        const int k = 2 * (pr >> 1) + (pr & 1);          // word of the operand registers: lo pairs 0, 1 -> 0, 1; hi pairs -> 2, 3
        if constexpr (NS == 2) {                         // fp16 two-way split of gsc * dY (jp_split2h, in two pieces)
            if (ph == 0) {
                const float x_ = __uint_as_float(src[2 * (pr & 1)]) * gsc, y_ = __uint_as_float(src[2 * (pr & 1) + 1]) * gsc;
                const h2 a = __builtin_convertvector(f2{x_, y_}, h2);
                const f2 af = __builtin_convertvector(a, f2);
                sr[0] = x_ - af[0];
                sr[1] = y_ - af[1];
                sa[ds][0][k] = __builtin_bit_cast(unsigned, a);
            } else {
                sa[ds][1][k] = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{sr[0], sr[1]}, h2));
            }
        } else if (ph == 0) {
            const float x_ = __uint_as_float(src[2 * (pr & 1)]), y_ = __uint_as_float(src[2 * (pr & 1) + 1]);
            const unsigned h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{x_, y_}, bf2));
            sr[0] = x_ - __uint_as_float(h0 << 16);
            sr[1] = y_ - __uint_as_float(h0 & 0xffff0000u);
            sa[ds][0][k] = h0;
        } else if (ph == 1) {
            const unsigned h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{sr[0], sr[1]}, bf2));
            sr[0] -= __uint_as_float(h1 << 16);
            sr[1] -= __uint_as_float(h1 & 0xffff0000u);
            sa[ds][1][k] = h1;
        } else {
            sa[ds][NS - 1][k] = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{sr[0], sr[1]}, bf2));
        }
    };
    typedef short jp_s16x4_ __attribute__((ext_vector_type(4)));
    // transpose-read results [fragment set][split][pixel half]; a tap's fragments are requested LA taps ahead.  Six products per tap
    // (192 pipe cycles) cover the LDS latency from one tap ahead; three products (96 cycles) do not: two ahead (W9S_LA), three sets
    constexpr int LA = NS == 2 ? W9S_LA : 1, NSET = LA + 1;
    jp_s16x4_ bh[NSET][NS][2];
    auto bread_half = [&](int ty, int tx, int g, int s, int h) -> jp_s16x4_ {
        const int imm = s * SPL + ((g / 2 + ty) * PC + 16 * (g % 2) + tx) * 64;
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (jp_s16x4 __attribute__((address_space(3)))*)(patch + (bbase[tx][h] + rbuf) + imm + 256 * h));
    };
    auto bfrag = [&](int par, int s) -> jp_u32x4 {
        return __builtin_bit_cast(jp_u32x4, __builtin_shufflevector(bh[par][s][0], bh[par][s][1], 0, 1, 2, 3, 4, 5, 6, 7));
    };
    if (T0 < T1) {
        int img, y0, x0;
        tile_org(T0, img, y0, x0);
        int tb = (img * Cout) * (int)HW + y0 * W + x0;
        tile_org(T0 + 1, img, y0, x0);
        int tbn = (img * Cout) * (int)HW + y0 * W + x0;
        aload(0, tb, kg * KGW);
        if (KGW > 1) aload(1, tb, kg * KGW + 1); else aload(1, tbn, kg * KGW);
        gload(T0);
        if (DB) lstore(0);
#pragma unroll
        for (int c = 0; c < NPIECE; ++c) split_piece(0, 0, c);
        for (int T = T0; T < T1; ++T) {
            if (!DB) lstore(0);
            __syncthreads();                                        // DB: buffer rbuf is complete, the other one is free
            gload(T + 1);                                            // next tile's patch: in flight during the MFMAs below
            tile_org(T + 2, img, y0, x0);
            const int tbnn = (img * Cout) * (int)HW + y0 * W + x0;
#pragma unroll
            for (int q0 = 0; q0 < LA; ++q0)
#pragma unroll
                for (int s_ = 0; s_ < NS; ++s_)
#pragma unroll
                    for (int h = 0; h < 2; ++h) bh[q0 % NSET][s_][h] = bread_half(q0 / 3, q0 % 3, 0, s_, h);
#pragma unroll
            for (int gi = 0; gi < KGW; ++gi) {
                const int g = gi, cur = gi & 1;
                // dY of K group gi + 2 (this tile, the next one or the one after it) -> the raw set group gi used
                if (gi + 2 < KGW) aload(cur, tb, kg * KGW + gi + 2);
                else if (gi + 2 - KGW < KGW) aload(cur, tbn, kg * KGW + gi + 2 - KGW);
                else aload(cur, tbnn, kg * KGW + gi + 2 - 2 * KGW);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int tq = 9 * gi + tap;                     // tap counter of the tile: fragment set tq % NSET, requested at tap tq - LA
                    const int tp = tq % NSET, tn = (tq + LA) % NSET, gn = (tq + LA) / 9, tapn = (tq + LA) % 9;
#pragma unroll
                    for (int m = 0; m < NPROD; ++m) {
                        // the products with split index sum <= NS - 1, smallest terms first: (2,0) (1,1) (0,2) (1,0) (0,1) (0,0) | (1,0) (0,1) (0,0)
                        const int sa_ = NS == 3 ? ((m == 0) ? 2 : ((m == 1 || m == 3) ? 1 : 0)) : (m == 0 ? 1 : 0);
                        const int sb = NS == 3 ? ((m == 0 || m == 3 || m == 5) ? 0 : ((m == 1 || m == 4) ? 1 : 2)) : (m == 1 ? 1 : 0);
                        acc[tap] = jp_mfma_bf16_sw<false>(sa[cur][sa_], bfrag(tp, sb), acc[tap]);
                        // behind the MFMAs of a tap: the 2 * NS transpose reads of the next tap (of the next K group's first tap behind the
                        // last one), one per MFMA (six products) or two behind the first (three products)
#pragma unroll
                        for (int rd = 0; rd < 2 * NS; ++rd) {
                            if ((NS == 3 ? rd : (rd < 2 ? 0 : rd - 1)) != m) continue;
                            if (gn < KGW) bh[tn][rd >> 1][rd & 1] = bread_half(tapn / 3, tapn % 3, gn, rd >> 1, rd & 1);
                        }
                        // ... and, spread over the K group, the pieces of the next K group's dY split
                        {
                            const int slot = NPROD * tap + m;
                            constexpr int EVERY = NS == 3 ? 4 : 3;
                            if (slot >= 2 && (slot - 2) % EVERY == 0 && (slot - 2) / EVERY < NPIECE) split_piece(cur ^ 1, cur ^ 1, (slot - 2) / EVERY);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // DB: one item of the NEXT tile's patch is split and stored into the other buffer behind each of the first
                    // NQ taps of the tile's last K group
                    if (DB && gi == KGW - 1 && tap < NQ) {
                        lstore1(tap, rbuf ^ BUFB);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            tb = tbn;
            tbn = tbnn;
            if (DB) rbuf ^= BUFB;
            else __syncthreads();
        }
    }
    // ---- partial tile -> ws[zs][m][tap*Cm + ci]; C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const long Np = 9L * Cm;
    float* wz = ws + (long)(zs * KG + kg) * Cout * Np;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const long n = (long)tap * Cm + c0 + cb * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + ab * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (m < Cout) wz[(long)m * Np + n] = NS == 2 ? acc[tap][r] * osc : acc[tap][r];
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// "W1S": the 1x1 stride-1 weight gradient (dW[co][ci] = sum_p dY[co][p] * X[ci][p]; the CRP / reduce layers) on the bf16
// pipe.  No tap shifts, so BOTH operands are pixel-contiguous with aligned 8-pixel groups: X is staged in its natural
// [channel][pixel] layout as three bf16 planes (a thread converts 8 consecutive pixels of one channel: two 16-byte loads,
// three 16-byte LDS writes; row pitch 272 B -> conflict-free b128 fragment reads), dY comes straight from global memory and
// is split in registers like in W9S.  Workgroup = 8 waves x (32 output channels) = 256 output channels x 128 input channels
// (the W1 tile and split-K plan); a wave reads its four 32-channel B blocks for every K group: 12 ds_read_b128 + 2 dY
// loads per 24 MFMAs.  Output: ws[split][m][ci] (wgrad_reduce4_kernel with one "tap").
// Preconditions: Cm % 128 == 0, W % 32 == 0, H % 4 == 0 (rows >= Cout are clamped / masked).
__global__ __launch_bounds__(512, 2) void jp_wgrad_w1s_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             float* __restrict__ ws, int Cout, int Cx, int Cm, int H, int W,
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
    constexpr int NT = 512, TR = 4, NC = 128, P = TR * 32;
    constexpr int PITCH = P * 2 + 16;                 // bytes per channel row of one split plane
    constexpr int SPL = NC * PITCH;                   // bytes per split plane
    constexpr int NQ = NC * (P / 8) / NT;             // (channel, pixel octet) items per thread: 4
    constexpr int KGR = TR * 2;
    __shared__ __attribute__((aligned(16))) unsigned char patch[NS * SPL];
    const int t = threadIdx.x, lane = t & 63;
    const int ab = __builtin_amdgcn_readfirstlane(t >> 6);
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
    auto tile_org = [&](int T, int& img, int& y0, int& x0) {
        const int Tc = min(T, ntiles - 1);
        img = Tc / tiles_img;
        const int r = Tc - img * tiles_img;
        y0 = (r / tiles_x) * TR;
        x0 = (r % tiles_x) * 32;
    };
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, dy_bytes, 0x00020000);
    const int arow = (min(m0 + ab * 32 + l31, Cout - 1) * (int)HW + 8 * lhi) * 4;
    jp_u32x4 araw[2][2];
    auto aload = [&](int slot, int tbase, int g) {
        const int so = __builtin_amdgcn_readfirstlane((tbase + (g / 2) * W + 16 * (g % 2)) * 4);
        araw[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(drs, arow, so, 0);
        araw[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(drs, arow + 16, so, 0);
    };
    // staging: item e = t + NT*q -> (octet o = e & 15 of the tile's 128 pixels: tile row o >> 2, columns 8*(o & 3) .. +7; channel
    // e >> 4): lanes run along the octets of a channel row -> 16-byte global loads, contiguous 16-byte LDS writes
    jp_u32x4 rx[NQ][2];
    auto gload = [&](int T) {
        int img, y0, x0;
        tile_org(T, img, y0, x0);
        const float* xc = x + ((long)img * Cx + c0) * HW + (long)y0 * W + x0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = t + NT * q, o = e & 15, c = e >> 4;
            const jp_u32x4* p = reinterpret_cast<const jp_u32x4*>(xc + (long)c * HW + (o >> 2) * W + 8 * (o & 3));
            rx[q][0] = p[0];
            rx[q][1] = p[1];
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = t + NT * q, o = e & 15, c = e >> 4;
            jp_u32x4 w0, w1, w2;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                unsigned sq[3];
                jp_split_ns(__uint_as_float(rx[q][0][2 * k]), __uint_as_float(rx[q][0][2 * k + 1]), xsc, sq);
                w0[k] = sq[0]; w1[k] = sq[1]; w2[k] = sq[2];
                jp_split_ns(__uint_as_float(rx[q][1][2 * k]), __uint_as_float(rx[q][1][2 * k + 1]), xsc, sq);
                w0[2 + k] = sq[0]; w1[2 + k] = sq[1]; w2[2 + k] = sq[2];
            }
            unsigned char* d = patch + c * PITCH + o * 16;
            *reinterpret_cast<jp_u32x4*>(d) = w0;
            *reinterpret_cast<jp_u32x4*>(d + SPL) = w1;
            if constexpr (NS == 3) *reinterpret_cast<jp_u32x4*>(d + 2 * SPL) = w2;
        }
    };
    jp_f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // B fragment of K group g (tile row g/2, columns 16*(g%2) + 8*lhi .. +7 = octet 4*(g/2) + 2*(g%2) + lhi), block j, split s
    const unsigned char* bp = patch + l31 * PITCH + lhi * 16;
    auto bread = [&](int g, int j, int s) -> jp_u32x4 {
        return *reinterpret_cast<const jp_u32x4*>(bp + s * SPL + j * 32 * PITCH + (4 * (g / 2) + 2 * (g % 2)) * 16);
    };
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
#pragma unroll
            for (int g = 0; g < KGR; ++g) {
                if (g + 1 < KGR) aload((g + 1) & 1, tb, g + 1);
                else aload((g + 1) & 1, tbn, 0);
                jp_u32x4 sa[3];
                {
                    const jp_u32x4 lo = araw[g & 1][0], hi = araw[g & 1][1];
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
                for (int s_ = 0; s_ < NS; ++s_) bq[0][s_] = bread(g, 0, s_);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j + 1 < 4) {
#pragma unroll
                        for (int s_ = 0; s_ < NS; ++s_) bq[(j + 1) & 1][s_] = bread(g, j + 1, s_);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#define JP_W1S_MFMA(SA_, SB_) acc[j] = jp_mfma_bf16_sw<false>(sa[SA_], bq[j & 1][SB_], acc[j])
                    JP_SPLIT_PRODUCTS(JP_W1S_MFMA);
#undef JP_W1S_MFMA
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            tb = tbn;
            __syncthreads();
        }
    }
    float* wz = ws + (long)zs * Cout * Cm;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long n = c0 + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + ab * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (m < Cout) wz[(long)m * Cm + n] = NS == 2 ? acc[j][r] * osc : acc[j][r];
        }
    }
}
