// "P9U" patch kernel: the decoder's iconv layers  y = act(Conv3x3_reflect(cat(skip, up2x(x), disp)) + b)
// (depth_decoder.py:76-77) on the fp32 MFMA pipe, in the parity-class form of conv.hip (an output pixel (2i+a, 2j+b)
// sees only a 2x2 patch of x through its 9 taps, against weights pre-summed per class (a, b): 4 "slots" instead of 9 taps
// on the upsampled segment) with the P9 machinery: operands staged once per 32-channel stage for all taps / slots, the
// weights streamed from L2 in MFMA fragment order as 16-byte buffer loads, no address arithmetic in the k loop.
//
// Workgroup: 8 waves, output tile 128 channels x (4 rows x 64 columns).  Wave (wm, class (py, px)) owns 64 channels x
// the tile's 2 x 32 pixels of its parity class (rows py, py+2; columns px, px+2, ...) -- a wave is class-uniform, so the
// upsampled stages read a class-specific weight stream while the skip / disparity stages share one.
//   S stage (32 skip channels, 9 taps, 144 k-steps): full-resolution patch 6 x 66, columns stored de-interleaved by
//       parity ([even cols | odd cols]) so that a fragment's 32 same-parity pixels are 32 consecutive words;
//   U stage (32 upsampled channels, 4 slots, 64 k-steps): low-resolution patch 4 x 34 (edge clamp == reflection of up(x));
//   D stage (the disparity channel, padded to 8: 9 taps x 4 k-pairs = 36 k-steps).
// Weight pack (PACK_FRAGSEG, conv.hip): [S: M tile][U: class][M tile][D: M tile] streams of quads (4 k-steps x 2 parities
// x 128 rows x 4 floats = 4 KB); step order inside a stage: (tap | slot, k-pair).
// Preconditions (host-checked): Cout % 128 == 0, C0 % 32 == 0, C1 % 32 == 0, C2 <= 8, H % 4 == 0, W % 64 == 0.
#pragma once
#include "igemm_p9.h"

template <class Epi>
__global__ __launch_bounds__(512, 2) void jp_igemm_p9u_kernel(const float* __restrict__ wp, const float* __restrict__ x0,
                                                              const float* __restrict__ x1, const float* __restrict__ x2,
                                                              Epi epi, int M, int C0, int C1, int C2, int H, int W) {
    constexpr int PRS = 6, PITS = 68, PHALF = 34;           // S / D patch: 6 rows x [33 even | pad | 33 odd | pad]
    constexpr int PRU = 4, PITU = 36;                       // U patch: 4 rows x [halo | 32 | halo] + 2 pad
    constexpr int QBYTES = 2 * 128 * 16;                    // bytes per quad
    __shared__ float patch[32 * PRS * PITS];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, py = (wave >> 1) & 1, px = wave & 1, cls = wave & 3;
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
    const int tiles_x = W / 64, tiles_y = H / 4;
    const int img = nt / (tiles_x * tiles_y), tr_ = nt - img * (tiles_x * tiles_y);
    const int y0 = (tr_ / tiles_x) * 4, x0c = (tr_ % tiles_x) * 64;
    const int MT = M / 128;
    const long HW = (long)H * W;
    const int h2 = H / 2, w2 = W / 2;
    const int NS0 = C0 / 32, NS1 = C1 / 32;

    // ---- weight streams (byte offsets into one buffer resource)
    const int TS = NS0 * 36 * QBYTES, TU = NS1 * 16 * QBYTES, TD = 9 * QBYTES;
    const int offS = mt * TS, offU = MT * TS + (cls * MT + mt) * TU, offD = MT * TS + 4 * MT * TU + mt * TD;
    const int wbytes = MT * TS + 4 * MT * TU + MT * TD;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, wbytes, 0x00020000);
    const int avo = (lhi * 128 + wm * 64 + l31) * 16;
    float ra[2][2][4];
    auto aload = [&](int slot, int byte_off) {
        const int so = __builtin_amdgcn_readfirstlane(byte_off);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const jp_p9_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(wrs, avo + 512 * i, so, 0);
            const jp_p9_f32x4 v = __builtin_bit_cast(jp_p9_f32x4, u);
#pragma unroll
            for (int j = 0; j < 4; ++j) ra[slot][i][j] = v[j];
        }
    };

    // ---- staging registers (union of the three stage kinds)
    float rb[24], rh;
    const float* xs = x0 + (long)img * C0 * HW;             // skip segment, channel c at xs + c*HW
    const float* xu = x1 + (long)img * C1 * h2 * w2;
    const float* xd = x2 ? x2 + (long)img * C2 * HW : nullptr;
    int xl = jp_reflect(x0c - 1, W), xr = jp_reflect(x0c + 64, W);
    // S / D patch: wave w takes channels w + 8*(r%4) of patch row r/4 (24 rows); lanes run along the 64 centre columns
    auto gloadS = [&](const float* xc, int nch) {           // xc: first channel of the stage; nch valid channels
#pragma unroll
        for (int r = 0; r < 24; ++r) {
            const int pr = r / 4, c = wave + 8 * (r % 4);
            const int yy = jp_reflect(y0 - 1 + pr, H);
            rb[r] = c < nch ? xc[(long)c * HW + (long)yy * W + x0c + lane] : 0.f;
        }
        rh = 0.f;
        if (t < 2 * 32 * PRS) {
            const int side = t & 1, rho = t >> 1, pr = rho / 32, c = rho % 32;
            const int yy = jp_reflect(y0 - 1 + pr, H);
            if (c < nch) rh = xc[(long)c * HW + (long)yy * W + (side ? xr : xl)];
        }
    };
    auto lstoreS = [&]() {
        const int pos = ((lane + 1) & 1) * PHALF + ((lane + 1) >> 1);
#pragma unroll
        for (int r = 0; r < 24; ++r) {
            const int pr = r / 4, c = wave + 8 * (r % 4);
            patch[(c * PRS + pr) * PITS + pos] = rb[r];
        }
        if (t < 2 * 32 * PRS) {
            const int side = t & 1, rho = t >> 1, pr = rho / 32, c = rho % 32;
            patch[(c * PRS + pr) * PITS + (side ? PHALF + 32 : 0)] = rh;     // column 65 (odd half, index 32) / column 0
        }
    };
    // U patch: half-wave hw takes channels hw + 16*(r%2) of patch row r/2 (8 rows); lanes run along the 32 centre columns
    const int hw = t >> 5, l32 = t & 31;
    const int i0 = y0 / 2, j0 = x0c / 2;
    auto gloadU = [&](const float* xc) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int pr = r / 2, c = hw + 16 * (r % 2);
            const int ii = min(max(i0 - 1 + pr, 0), h2 - 1);
            rb[r] = xc[((long)c * h2 + ii) * w2 + j0 + l32];
        }
        rh = 0.f;
        if (t < 2 * 32 * PRU) {
            const int side = t & 1, rho = t >> 1, pr = rho / 32, c = rho % 32;
            const int ii = min(max(i0 - 1 + pr, 0), h2 - 1);
            const int jj = side ? min(j0 + 32, w2 - 1) : max(j0 - 1, 0);
            rh = xc[((long)c * h2 + ii) * w2 + jj];
        }
    };
    auto lstoreU = [&]() {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int pr = r / 2, c = hw + 16 * (r % 2);
            patch[(c * PRU + pr) * PITU + 1 + l32] = rb[r];
        }
        if (t < 2 * 32 * PRU) {
            const int side = t & 1, rho = t >> 1, pr = rho / 32, c = rho % 32;
            patch[(c * PRU + pr) * PITU + (side ? 33 : 0)] = rh;
        }
    };

    jp_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- B fragment bases.  S / D: pixel (row py + 2j + ty, column 2*l31 + px + tx) -> de-interleaved position
    // ((px+tx)&1)*34 + l31 + ((px+tx)>>1): taps tx = 0, 2 share a base (+0 / +1), tap tx = 1 has its own.
    const float* bsA = patch + (lhi * PRS + py) * PITS + l31 + px * PHALF;                 // u = px (+2 -> +1)
    const float* bsB = patch + (lhi * PRS + py) * PITS + l31 + (px ? 1 : PHALF);           // u = px + 1
    const float* bu = patch + (lhi * PRU + py) * PITU + l31 + px;

    // one stage: KP k-pairs per tap, T taps (9) or slots (4); quad stream at byte offset `cur`, the first quad of the
    // stage that follows at `nxt` (prefetched by the last quad)
    auto stageS = [&](auto kp_tag, int cur, int nxt) {
        constexpr int KP = decltype(kp_tag)::value, STEPS = 9 * KP, QS = STEPS / 4;
        auto bo = [&](int q, int j) -> int {
            const int tap = q / KP, s = q % KP, ty = tap / 3;
            return ((2 * s) * PRS + 2 * j + ty) * PITS;
        };
        auto rd = [&](int q, int j) -> float {
            const int tx = (q / KP) % 3;
            return tx == 1 ? bsB[bo(q, j)] : bsA[bo(q, j) + (tx == 2 ? 1 : 0)];
        };
        float b0 = rd(0, 0), b1 = rd(0, 1);
#pragma unroll
        for (int q = 0; q < STEPS; ++q) {
            const int qn = q + 1 < STEPS ? q + 1 : q;
            const float nb0 = rd(qn, 0), nb1 = rd(qn, 1);
            const int Q = q / 4, j4 = q % 4;
            if (j4 == 0) aload((Q + 1) % 2, Q + 1 < QS ? cur + (Q + 1) * QBYTES : nxt);
            const float a0 = ra[Q % 2][0][j4], a1 = ra[Q % 2][1][j4];
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            b0 = nb0; b1 = nb1;
        }
        static_assert(QS % 2 == 0 || QS == 9, "ring parity");
    };
    auto stageU = [&](int cur, int nxt) {
        constexpr int STEPS = 64, QS = 16;
        auto bo = [&](int q, int j) -> int {
            const int slot = q / 16, s = q % 16, r = slot >> 1, sx = slot & 1;
            return ((2 * s) * PRU + j + r) * PITU + sx;
        };
        float b0 = bu[bo(0, 0)], b1 = bu[bo(0, 1)];
#pragma unroll
        for (int q = 0; q < STEPS; ++q) {
            const int qn = q + 1 < STEPS ? q + 1 : q;
            const float nb0 = bu[bo(qn, 0)], nb1 = bu[bo(qn, 1)];
            const int Q = q / 4, j4 = q % 4;
            if (j4 == 0) aload((Q + 1) % 2, Q + 1 < QS ? cur + (Q + 1) * QBYTES : nxt);
            const float a0 = ra[Q % 2][0][j4], a1 = ra[Q % 2][1][j4];
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            b0 = nb0; b1 = nb1;
        }
    };

    // ---- stage sequence: S x NS0, U x NS1, D x (C2 ? 1 : 0).  Every stage starts with its quad 0 in ring slot 0 (the
    // S / U stages have an even number of quads; the 9-quad D stage comes last).
    aload(0, NS0 ? offS : offU);
    if (NS0) gloadS(xs, 32); else gloadU(xu);
    for (int st = 0; st < NS0; ++st) {
        lstoreS();
        __syncthreads();
        if (st + 1 < NS0) gloadS(xs + (long)(st + 1) * 32 * HW, 32);
        else if (NS1) gloadU(xu);
        else if (C2) gloadS(xd, C2);
        const int cur = offS + st * 36 * QBYTES;
        stageS(std::integral_constant<int, 16>{}, cur, st + 1 < NS0 ? cur + 36 * QBYTES : (NS1 ? offU : offD));
        __syncthreads();
    }
    for (int st = 0; st < NS1; ++st) {
        lstoreU();
        __syncthreads();
        if (st + 1 < NS1) gloadU(xu + (long)(st + 1) * 32 * h2 * w2);
        else if (C2) gloadS(xd, C2);
        const int cur = offU + st * 16 * QBYTES;
        stageU(cur, st + 1 < NS1 ? cur + 16 * QBYTES : offD);
        __syncthreads();
    }
    if (C2) {
        lstoreS();
        __syncthreads();
        stageS(std::integral_constant<int, 4>{}, offD, offD);
    }

    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int m0 = mt * 128;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = img * (int)HW + (y0 + py + 2 * j) * W + x0c + 2 * l31 + px;
        const typename Epi::St se = epi.col(p);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (m < M) epi.put(se, m, acc[i][j][r]);
            }
        }
    }
}
