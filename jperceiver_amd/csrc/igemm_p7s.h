// "P7S" stem kernel: FORWARD of the 7x7 stride-2 pad-3 (zero) stem convolutions (CIN = 3: resnet.py:90, CIN = 6: the pose
// encoder's pair input) on the bf16 matrix pipe (split products, igemm_p9s.h).  GEMM M = 64 output channels, N = output
// pixels, K = (c, ky, kx) with ky and kx padded 7 -> 8: one 16-wide MFMA K step = TWO tap rows (c, 2v), (c, 2v+1) x 8 kx, so a
// lane's 8 k-values are 8 CONSECUTIVE input floats x[c][2oy+ky-3][2ox-3 .. 2ox+4] (the pad taps meet zero weights).
//   B: the whole (2*8+6) x 70 input patch of an (8 rows x 32 columns) output tile -- all channels, the entire K -- is staged
//      ONCE per tile as bf16 triples, rows of 35 dwords (bf16 pairs); the operand of output column l is dwords l .. l+3 of the
//      row (four conflict-free 4-byte reads: the start is only 4-byte aligned), a tap row ky / the next channel is an offset.
//   A: weights split once per step (PACK_SPLIT7): [step][split][k-half][64 rows] x 16 B, streamed from L2 one step ahead.
// Workgroup = 4 waves = one tile; wave wn owns output rows 2wn, 2wn+1 of the tile and all 64 channels (2 x 2 accumulators).
// Preconditions (host-checked): OH % 8 == 0, OW % 32 == 0, H == 2*OH, W == 2*OW, M <= 64.
#pragma once
#include "igemm_p9s.h"

#ifndef P7S_OCC
#define P7S_OCC 2
#endif
template <int CIN, class Epi>
__global__ __launch_bounds__(256, P7S_OCC) void jp_igemm_p7s_kernel(const unsigned* __restrict__ wp, const float* __restrict__ x, Epi epi,
                                                              int M, int OH, int OW, int ntiles, const float* __restrict__ xam) {
    constexpr int NS = JP_NS;
    float xsc = 1.f, osc = 1.f;
    if constexpr (NS == 2) {    // operand scales, see jp_igemm_p9s_body (the pack's header: PACK_SPLIT7)
        const int kx = __builtin_amdgcn_readfirstlane(jp_scale_exp(jp_slot_amax(xam)));
        xsc = jp_exp2i(kx);
        osc = jp_exp2i(-kx) * __uint_as_float(__builtin_amdgcn_readfirstlane(wp[1]));
        wp += JP_PACK_HDR;
    }
    constexpr int NT = 256, NJ = 2, TR = 8;
    constexpr int STEPS = 4 * CIN;                                 // two of the 8 (padded) tap rows of a channel per step
    constexpr int PRW = 2 * TR + 6, PDW = 35, PITCH = 36;          // patch rows per channel, dwords per row, row pitch (dwords)
    constexpr int SPLW = CIN * PRW * PITCH;                        // dwords per split plane
    constexpr int RPP = NT / PITCH, KP = (PRW + RPP - 1) / RPP;    // patch rows per staging pass (7), passes per channel (4)
    constexpr int NQ = CIN * KP;
    constexpr int SBYTES = NS * 2 * 64 * 16;
    __shared__ unsigned patch[NS * SPLW];

    const int t = threadIdx.x, lane = t & 63;
    const int wn = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int H = 2 * OH, W = 2 * OW;
    const long HW = (long)H * W;
    const int tiles_x = OW / 32, tiles_img = tiles_x * (OH / TR);
    auto tile_org = [&](int T, int& img, int& i0, int& j0) {
        const int Tc = min(T, ntiles - 1);
        img = Tc / tiles_img;
        const int r = Tc - img * tiles_img;
        i0 = (r / tiles_x) * TR;
        j0 = (r % tiles_x) * 32;
    };

    // ---- staging: thread (q2 = t % 36: dword of a patch row = input columns 2*j0 - 3 + 2*q2 + {0, 1}; r0 = t / 36 < 7) takes patch
    // rows r0 + 7k of every channel: global and LDS addresses are ONE lane offset each plus wave-uniform / compile-time terms
    // (buffer loads: SGPR resource of the tile's image, out-of-range offset = zero fill)
    const int q2 = t % PITCH, r0 = t / PITCH;
    const bool sthread = q2 < PDW && r0 < RPP;
    float rv[NQ][2];
    auto gload = [&](int T) {
        int img, i0, j0;
        tile_org(T, img, i0, j0);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (long)img * CIN * HW), 0,
                                                                             (int)(CIN * HW * 4), 0x00020000);
        const int yb = 2 * i0 - 3 + r0, xx = 2 * j0 - 3 + 2 * q2;
        const bool c0ok = sthread && xx >= 0 && xx < W, c1ok = sthread && xx + 1 >= 0 && xx + 1 < W;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int yy = yb + RPP * k;
            const bool rok = r0 + RPP * k < PRW && yy >= 0 && yy < H;
            // the LANE offset must itself be inside the resource (the range check sees it before the scalar offset is added)
            const unsigned lo = (unsigned)(yy * W + xx) * 4u;
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const int ub = __builtin_amdgcn_readfirstlane((int)(c * HW * 4));
                rv[c * KP + k][0] = jp_gather(xrs, (rok && c0ok) ? lo : 0x7ffffff0u, ub);
                rv[c * KP + k][1] = jp_gather(xrs, (rok && c1ok) ? lo + 4u : 0x7ffffff0u, ub);
            }
        }
    };
    unsigned* const lbase = patch + r0 * PITCH + q2;
    auto lstore = [&]() {
        if (!sthread) return;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            if (r0 + RPP * k >= PRW) continue;
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                unsigned sq[3];
                jp_split_ns(rv[c * KP + k][0], rv[c * KP + k][1], xsc, sq);
                unsigned* o = lbase + (c * PRW + RPP * k) * PITCH;
                o[0] = sq[0];
                o[SPLW] = sq[1];
                if constexpr (NS == 3) o[2 * SPLW] = sq[2];
            }
        }
    };

    // ---- weights: [step][split][k-half][64 rows] x 16 B
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(wp), 0, (STEPS + 1) * SBYTES, 0x00020000);
    const int avo = (lhi * 64 + l31) * 16;
    jp_u32x4 ra[2][2][NS];
    auto aload = [&](int slot, int step) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < NS; ++s)
                ra[slot][i][s] = __builtin_amdgcn_raw_buffer_load_b128(wrs, avo + i * 512 + s * (2 * 64 * 16), step * SBYTES, 0);
    };
    // ---- B: step u = (channel c = u / 4, tap rows ky = 2*(u % 4) + lhi); LDS dword of that patch row for output row j of this wave:
    // (c*PRW + 2*(2wn + j) + ky) * PITCH + l31 -- one lane base, everything else compile-time
    const unsigned* bpl = patch + (4 * wn + lhi) * PITCH + l31;
    jp_u32x4 rb[2][NJ][NS];
    auto bload = [&](int slot, int u) {
        const int d0 = ((u / 4) * PRW + 2 * (u % 4)) * PITCH;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const unsigned* r = bpl + d0 + s * SPLW + (2 * j) * PITCH;
                rb[slot][j][s] = jp_u32x4{r[0], r[1], r[2], r[3]};
            }
    };

    // one tile per workgroup (co-resident workgroups hide each other's staging; a persistent tile loop made the compiler hoist
    // the epilogue's 32 row offsets across it and spill)
    const int T = blockIdx.x;
    jp_f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    aload(0, 0);
    gload(T);
    lstore();
    __syncthreads();
    bload(0, 0);
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
        if (u + 1 < STEPS) { aload((u + 1) & 1, u + 1); bload((u + 1) & 1, u + 1); }
        __builtin_amdgcn_sched_barrier(0);
#define JP_P7S_MFMA(SA_, SB_)                                                                                            \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j)                         \
        acc[i][j] = jp_mfma_bf16_sw<false>(ra[u & 1][i][SA_], rb[u & 1][j][SB_], acc[i][j])
        JP_SPLIT_PRODUCTS(JP_P7S_MFMA);
#undef JP_P7S_MFMA
        __builtin_amdgcn_sched_barrier(0);
    }
    int img, i0, j0;
    tile_org(T, img, i0, j0);
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int p = img * (OH * OW) + (i0 + 2 * wn + j) * OW + j0 + l31;
        const typename Epi::St se = epi.col(p);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (m < M) epi.put(se, m, NS == 2 ? acc[i][j][r] * osc : acc[i][j][r]);
            }
        }
    }
}
