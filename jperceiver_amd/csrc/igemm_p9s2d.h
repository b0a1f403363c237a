// "P9S2D" patch kernel: DGRAD of a 3x3 STRIDE-2 pad-1 convolution in parity-class form, on the bf16 matrix pipe (split
// products, igemm_p9s.h).  With y[oy][ox] = sum W[ky][kx] x[2oy+ky-1][2ox+kx-1], an input pixel (2i+a, 2j+b) of parity class
// (a, b) only meets the taps with ky = 1 (a = 0) or ky in {0, 2} (a = 1), same for the columns:
//     dX[c][2i+a][2j+b] = sum over co and the class's (1+a)(1+b) taps of  W[co][c][ky][kx] * dY[co][i + (ky==0)][j + (kx==0)]
// (zero where the dY index leaves the map).  9 taps over 4 classes, no zero products (the generic engine's stride-2 dgrad
// executes 16/9 of them).  A workgroup is CLASS-UNIFORM: it owns one class of a (TR x 32) tile of half-resolution pixels,
// stages the (TR+1) x 33 dY patch once per 16-channel stage of co (split into bf16 triples, P9S layout) and runs the class's
// 1, 2 or 4 steps on it; consecutive workgroups are the 4 classes of the same tile (they share the patch in L2).  Weights:
// the P9S dgrad pack (PACK_SPLIT, natural tap order), of which a class reads only its taps' steps.
// Preconditions (host-checked): h2 % (WN*NJ) == 0, w2 % 32 == 0, Cout % 16 == 0; M (= Cin) rows in tiles of 64*WM.
#pragma once
#include "igemm_p9s.h"

template <int WM, int WN, int NJ, class Epi>
__global__ __launch_bounds__(64 * WM * WN, 2) void jp_igemm_p9s2d_kernel(const unsigned* __restrict__ wp, const float* __restrict__ dy,
                                                                      Epi epi, int M, int C, int NST, int h2, int w2,
                                                                      const float* __restrict__ xam) {
    constexpr int NS = JP_NS;
    float xsc = 1.f, osc = 1.f;
    if constexpr (NS == 2) {    // operand scales, see jp_igemm_p9s_body
        const int kx = __builtin_amdgcn_readfirstlane(jp_scale_exp(jp_slot_amax(xam)));
        xsc = jp_exp2i(kx);
        osc = jp_exp2i(-kx) * __uint_as_float(__builtin_amdgcn_readfirstlane(wp[1]));
        wp += JP_PACK_HDR;
    }
    constexpr int NT = 64 * WM * WN;
    constexpr int TR = WN * NJ, PR = TR + 1, COLS = 33;
    constexpr int PLANE = PR * COLS;                          // 16-byte words per (split, k-half)
    constexpr int ITEMS = 2 * PLANE, NQ = (ITEMS + NT - 1) / NT;
    constexpr int BMT = 64 * WM;
    constexpr int SBYTES = NS * 2 * BMT * 16;                 // bytes per weight step
    __shared__ jp_u32x4 patch[NS * 2 * PLANE];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    int mt, ntc;
    {   // XCD band order, see jp_igemm_kernel
        const int gx = gridDim.x, gy = gridDim.y, G = gx & ~7;
        const int L = blockIdx.x + blockIdx.y * gx;
        if (L < G * gy) {
            const int j = L >> 3;
            mt = j % gy;
            ntc = (L & 7) * (G >> 3) + j / gy;
        } else {
            const int i = L - G * gy;
            mt = i % gy;
            ntc = G + i / gy;
        }
    }
    const int cls = 3 - (ntc & 3), nt = ntc >> 2;             // the 4-tap class first
    const int tiles_x = w2 / 32, tiles_y = h2 / TR;
    const int img = nt / (tiles_x * tiles_y), tr_ = nt - img * (tiles_x * tiles_y);
    const int i0 = (tr_ / tiles_x) * TR, j0 = (tr_ % tiles_x) * 32;
    const int m0 = mt * BMT;
    const long hw2 = (long)h2 * w2;
    const float* xin = dy + (long)img * C * hw2;

    unsigned soff[NQ];                                       // byte offset inside the image, bit 0 set = zero
    int loff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = t + NT * q;
        const int col = e % COLS, rp = e / COLS, pr = rp % PR, kh = rp / PR;
        const int yy = i0 + pr, xx = j0 + col;
        const bool ok = e < ITEMS && yy < h2 && xx < w2;
        soff[q] = ok ? (unsigned)(kh * 8 * hw2 + (long)yy * w2 + xx) * 4u : 1u;
        loff[q] = e < ITEMS ? (kh * PR + pr) * COLS + col : -1;
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0, (int)((long)C * hw2 * 4), 0x00020000);
    float rv[NQ][8];
    auto gload = [&](int stage) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ub = __builtin_amdgcn_readfirstlane((int)(((long)stage * 16 + k) * hw2 * 4));
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
                unsigned sp[3];
                jp_split_ns(rv[q][2 * k], rv[q][2 * k + 1], xsc, sp);
                w0[k] = sp[0]; w1[k] = sp[1]; w2_[k] = sp[2];
            }
            patch[loff[q]] = w0;
            patch[2 * PLANE + loff[q]] = w1;
            if constexpr (NS == 3) patch[4 * PLANE + loff[q]] = w2_;
        }
    };

    jp_f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const long tile_bytes = ((long)NST * 9 + P9S_AHEAD) * SBYTES;
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
    const jp_u32x4* bp = patch + (lhi * PR + wn * NJ) * COLS + l31;
    jp_u32x4 rb[2][NJ][NS];

    // one class: CA, CB compile-time.  Step v of a stage <-> tap (ky, kx); ring slot of a step = (steps issued so far) & 1
    auto run_class = [&](auto ca_tag, auto cb_tag) {
        constexpr int CA = decltype(ca_tag)::value, CB = decltype(cb_tag)::value;
        constexpr int T = (1 + CA) * (1 + CB);
        auto tap_of = [](int v) constexpr {
            const int vy = CB ? v / 2 : v, vx = CB ? v % 2 : 0;
            const int ky = CA ? 2 * vy : 1, kx = CB ? 2 * vx : 1;
            return ky * 3 + kx;
        };
        auto bload = [&](int slot, int v) {
            const int tap = tap_of(v);
            const int dr = (tap / 3 == 0) ? 1 : 0, dc = (tap % 3 == 0) ? 1 : 0;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int s = 0; s < NS; ++s) rb[slot][j][s] = bp[s * 2 * PLANE + (j + dr) * COLS + dc];
        };
        auto run_stage = [&](auto par_tag, int stage) {
            constexpr int PAR = decltype(par_tag)::value;
            lstore();
            __syncthreads();
            if (stage + 1 < NST) gload(stage + 1);
            const int ab = __builtin_amdgcn_readfirstlane(stage * 9 * SBYTES);
            bload(PAR & 1, 0);
#pragma unroll
            for (int v = 0; v < T; ++v) {
                // weights of the next step (of this stage, or the class's first one of the next stage; after the last
                // stage a valid step is re-read and dropped)
                aload((PAR + v + 1) & 1, v + 1 < T ? ab + tap_of(v + 1) * SBYTES
                                                   : (stage + 1 < NST ? ab + (9 + tap_of(0)) * SBYTES : tap_of(0) * SBYTES));
                if (v + 1 < T) bload((PAR + v + 1) & 1, v + 1);
                __builtin_amdgcn_sched_barrier(0);
#define JP_P9S2D_MFMA(SA_, SB_)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j)                         \
        acc[i][j] = jp_mfma_bf16_sw<false>(ra[(PAR + v) & 1][i][SA_], rb[(PAR + v) & 1][j][SB_], acc[i][j])
                JP_SPLIT_PRODUCTS(JP_P9S2D_MFMA);
#undef JP_P9S2D_MFMA
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        };
        aload(0, tap_of(0) * SBYTES);
        gload(0);
        for (int stage = 0; stage < NST; stage += 2) {
            run_stage(std::integral_constant<int, 0>{}, stage);
            if (stage + 1 < NST) run_stage(std::integral_constant<int, (T & 1)>{}, stage + 1);
        }
        // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        const int W = 2 * w2;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int p = img * (int)(4 * hw2) + (2 * (i0 + wn * NJ + j) + CA) * W + 2 * (j0 + l31) + CB;
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
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    switch (cls) {
        case 0: run_class(I0{}, I0{}); break;
        case 1: run_class(I0{}, I1{}); break;
        case 2: run_class(I1{}, I0{}); break;
        default: run_class(I1{}, I1{}); break;
    }
}
