// "P9SD" patch kernel: dgrad of the UPSAMPLED segment of an iconv layer, directly at half resolution, on the bf16 matrix pipe
// (split products, igemm_p9s.h).  With y = Conv3x3_reflect(cat(.., up2x(x), ..)) an output pixel (2i'+a, 2j'+b) of parity class
// (a, b) sees a 2x2 patch of x through 4 pre-summed weight slots W'_{ab}[r][s] (conv.hip, "parity-class form"), hence
//     dX[c][i][j] = sum_{class (a,b), slot (r,s), co}  W'_{ab}[r][s][co][c] * dY[co][2i + 2 - a - 2r][2j + 2 - b - 2s]
// (zero where the dY index leaves the image; the edge-clamp fold is DgradUPBorderB's pass): a 16-"tap" correlation over the
// FULL-resolution dY, sampled at stride 2.  GEMM M = c, N = half-resolution pixels, K = (16-channel stage of co, 16 (class,
// slot) steps).  Tiling as P9S: workgroup = 2x2 waves, 128 rows x (4 half-res rows x 32 columns); per stage the
// (2*4+2) x 66 full-resolution dY patch is staged once, split into bf16 triples, with its columns DE-INTERLEAVED by parity
// ([even | odd]) so that the 32 stride-2 pixels of a fragment are 32 consecutive 16-byte words; a step's B operand is one
// ds_read_b128 at a compile-time offset, the weights (PACK_SPLITUPD) stream from L2 one step ahead.
// Preconditions (host-checked): C rows in tiles of 128 (rows >= M masked), Cout padded to 16 in the pack, h2 % 4 == 0, w2 % 32 == 0.
#pragma once
#include "igemm_p9s.h"

template <class Epi>
__global__ __launch_bounds__(256, 2) void jp_igemm_p9sd_kernel(const unsigned* __restrict__ wp, const float* __restrict__ dy,
                                                               Epi epi, int M, int Cout, int NST, int h2, int w2,
                                                               const float* __restrict__ xam) {
    constexpr int NS = JP_NS;
    float xsc = 1.f, osc = 1.f;
    if constexpr (NS == 2) {    // operand scales, see jp_igemm_p9s_body (the pack's header: PACK_SPLITUPD)
        const int kx = __builtin_amdgcn_readfirstlane(jp_scale_exp(jp_slot_amax(xam)));
        xsc = jp_exp2i(kx);
        osc = jp_exp2i(-kx) * __uint_as_float(__builtin_amdgcn_readfirstlane(wp[1]));
        wp += JP_PACK_HDR;
    }
    constexpr int NT = 256, WN = 2, NJ = 2, TR = WN * NJ;
    constexpr int PR = 2 * TR + 2, COLS = 66, PHALF = 34, PITS = 2 * PHALF;
    constexpr int PLS = PR * PITS;                           // 16-byte words per (split, k-half) plane
    constexpr int ITEMS = 2 * PR * COLS, NQ = (ITEMS + NT - 1) / NT;
    constexpr int STEPS = 16, BMT = 128;
    constexpr int SBYTES = NS * 2 * BMT * 16;
    __shared__ jp_u32x4 patch[NS * 2 * PLS];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    int mt, nt;
    {   // XCD band order, see jp_igemm_kernel
        const int gx = gridDim.x, gy = gridDim.y, G = gx & ~7;
        const int L = blockIdx.x + blockIdx.y * gx;
        if (L < G * gy) {
            const int j = L >> 3;
            mt = j % gy;
            nt = (L & 7) * (G >> 3) + j / gy;
        } else {
            const int i = L - G * gy;
            mt = i % gy;
            nt = G + i / gy;
        }
    }
    const int H = 2 * h2, W = 2 * w2;
    const int tiles_x = w2 / 32, tiles_y = h2 / TR;
    const int img = nt / (tiles_x * tiles_y), tr_ = nt - img * (tiles_x * tiles_y);
    const int i0 = (tr_ / tiles_x) * TR, j0 = (tr_ % tiles_x) * 32;
    const int m0 = mt * BMT;
    const long HW = (long)H * W;
    const float* xin = dy + (long)img * Cout * HW;

    // ---- staging map: item e = t + NT*q -> (patch column 0..65 <-> full-res column 2*j0 - 1 + col, patch row, k-half)
    unsigned soff[NQ];                                       // byte offset inside the image, bit 0 set = zero
    int loff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = t + NT * q;
        const int col = e % COLS, rp = e / COLS, pr = rp % PR, kh = rp / PR;
        const int yy = 2 * i0 - 1 + pr, xx = 2 * j0 - 1 + col;
        const bool ok = e < ITEMS && yy >= 0 && yy < H && xx >= 0 && xx < W;
        soff[q] = ok ? (unsigned)(kh * 8 * HW + (long)yy * W + xx) * 4u : 1u;
        loff[q] = e < ITEMS ? (kh * PR + pr) * PITS + (col & 1) * PHALF + (col >> 1) : -1;
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0, (int)((long)Cout * HW * 4), 0x00020000);
    float rv[NQ][8];
    auto gload = [&](int stage) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ch = stage * 16 + k;                   // + 8 per k-half through soff; channels >= Cout read as zero (bounds)
            const int ub = __builtin_amdgcn_readfirstlane((int)((long)ch * HW * 4));
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float v = jp_gather(xrs, soff[q] & ~1u, ub);
                rv[q][k] = (soff[q] & 1u) ? 0.f : v;
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (loff[q] < 0) continue;
            jp_u32x4 w0, w1, w2_;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned sq[3];
                jp_split_ns(rv[q][2 * k], rv[q][2 * k + 1], xsc, sq);
                w0[k] = sq[0]; w1[k] = sq[1]; w2_[k] = sq[2];
            }
            patch[loff[q]] = w0;
            patch[2 * PLS + loff[q]] = w1;
            if constexpr (NS == 3) patch[4 * PLS + loff[q]] = w2_;
        }
    };

    jp_f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const long tile_bytes = ((long)NST * STEPS + P9S_AHEAD) * SBYTES;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(wp)) + (long)mt * tile_bytes, 0, (int)tile_bytes, 0x00020000);
    const int avo = (lhi * BMT + wm * 64 + l31) * 16;
    jp_u32x4 ra[2][2][NS];
    auto aload = [&](int slot, int step_bytes) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < NS; ++s)
                ra[slot][i][s] = __builtin_amdgcn_raw_buffer_load_b128(wrs, avo + i * 512 + s * (2 * BMT * 16), step_bytes, 0);
    };
    aload(0, 0);
    // B fragment of step q = (a, b, r, s), pixel row j of the wave: patch row 2*(wn*NJ + j) + 3 - a - 2r, position
    // (b == 0 ? PHALF : 0) + l31 + 1 - s
    const jp_u32x4* bp = patch + lhi * PLS + (2 * wn * NJ) * PITS + l31;
    jp_u32x4 rb[2][NJ][NS];
    auto bload = [&](int slot, int u) {
        const int a = u >> 3, b = (u >> 2) & 1, r = (u >> 1) & 1, s = u & 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int sp = 0; sp < NS; ++sp)
                rb[slot][j][sp] = bp[sp * 2 * PLS + (2 * j + 3 - a - 2 * r) * PITS + (b == 0 ? PHALF : 0) + 1 - s];
    };
#define JP_P9SD_MFMA(SA_, SB_)                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j)                         \
        acc[i][j] = jp_mfma_bf16_sw<false>(ra[u & 1][i][SA_], rb[u & 1][j][SB_], acc[i][j])
    gload(0);
    for (int stage = 0; stage < NST; ++stage) {
        lstore();
        __syncthreads();
        if (stage + 1 < NST) gload(stage + 1);
        const int ab = __builtin_amdgcn_readfirstlane(stage * STEPS * SBYTES);
        bload(0, 0);
#pragma unroll
        for (int u = 0; u < STEPS; ++u) {
            aload((u + 1) & 1, ab + (u + 1) * SBYTES);      // 16 steps per stage: the ring parity is the step parity
            if (u + 1 < STEPS) bload((u + 1) & 1, u + 1);
            __builtin_amdgcn_sched_barrier(0);
            JP_SPLIT_PRODUCTS(JP_P9SD_MFMA);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
#undef JP_P9SD_MFMA

    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int p = img * (h2 * w2) + (i0 + wn * NJ + j) * w2 + j0 + l31;
        const typename Epi::St se = epi.col(p);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (m < M) epi.put(se, m, NS == 2 ? acc[i][j][r] * osc : acc[i][j][r]);
            }
        }
    }
}
