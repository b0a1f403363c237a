// "P9US" patch kernel: the decoder's iconv layers  y = act(Conv3x3_reflect(cat(skip, up2x(x), disp)) + b)
// (depth_decoder.py:76-77) in the parity-class form of igemm_p9u.h, with every fp32 product formed on the BF16 matrix pipe
// from three-way splits of both operands (igemm_p9s.h has the arithmetic argument: fp32-equivalent accuracy, 6 MFMAs of 32
// cycles instead of 8 of 64).
//
// Workgroup: 8 waves, output tile 128 channels x (4 rows x 64 columns).  Wave (wm, class (py, px)) owns 64 channels x the
// tile's 2 x 32 pixels of its parity class (rows py, py+2; columns px, px+2, ...): a wave is class-uniform, so the
// upsampled stages read a class-specific weight stream while the skip / disparity stages share one.  Stages of 16 channels:
//   S (skip, 9 taps = 9 steps): full-resolution patch 6 x 66, columns stored de-interleaved by parity ([even | odd]) so that
//       a fragment's 32 same-parity pixels are 32 consecutive 16-byte words;
//   U (upsampled, 4 slots = 4 steps): low-resolution patch 4 x 34 (edge clamp == reflection of up(x));
//   D (the disparity channel(s), padded to 16: 9 steps).
// LDS word = 8 channels of one pixel as bf16, [split][k-half][patch row][position]; weights (PACK_SPLITSEG, conv.hip):
// streams [S: M tile][U: class][M tile][D: M tile] of steps, a step = [split][k-half][128 rows] x 16 bytes, one step ahead.
// Preconditions (host-checked): Cout % 128 == 0, C0 % 32 == 0, C1 % 16 == 0, C2 <= 8, H % 4 == 0, W % 64 == 0.
#pragma once
#include "igemm_p9s.h"

// NJ = pixel rows per parity class and wave: 2 (tile 4 rows x 64 columns, rounds 3) or 4 (8 rows x 64 columns, round 4: a wave's
// weight fragments serve twice the pixels and the patches carry 10 / 6 rows for 8 / 4 instead of 6 / 4 for 4 / 2)
template <class Epi, int NJ = 2, bool DB = false>
__global__ __launch_bounds__(512, 2) void jp_igemm_p9us_kernel(const unsigned* __restrict__ wp, const float* __restrict__ x0,
                                                               const float* __restrict__ x1, const float* __restrict__ x2,
                                                               Epi epi, int M, int C0, int C1, int C2, int H, int W) {
    constexpr int NT = 512;
    constexpr int TR = 2 * NJ;                                             // output rows per tile
    constexpr int PRS = TR + 2, PHALF = 34, PITS = 2 * PHALF, COLS_S = 66; // S / D patch rows x [33 even | pad | 33 odd | pad]
    constexpr int PRU = NJ + 2, PITU = 34;                                 // U patch: NJ + 2 rows x [halo | 32 | halo]
    constexpr int PLS = PRS * PITS, PLU = PRU * PITU;                      // 16-byte words per (split, k-half) plane
    constexpr int ITS = 2 * PRS * COLS_S, NQS = (ITS + NT - 1) / NT;       // 792 items -> 2 rounds
    constexpr int ITU = 2 * PRU * PITU;                                    // 272 / 408 items -> 1 round
    static_assert(ITU <= NT, "one staging round for the low-resolution patch");
    constexpr int SBYTES = 3 * 2 * 128 * 16;                               // bytes per weight step
    // P9US_DB (round 4, as P9S_DB in igemm_p9s.h): two patch buffers; the NEXT stage's patch (whatever its kind) is split and stored
    // into the other one underneath this stage's MFMAs and a stage ends in one barrier.  Needs C0 % 32 == 0 and C1 % 32 == 0 (the
    // buffer of a stage is a compile-time parity); the host launches the single-buffer instantiation otherwise.
    constexpr int BUFW = 3 * 2 * PLS;
    __shared__ jp_u32x4 patch[(DB ? 2 : 1) * BUFW];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, py = (wave >> 1) & 1, px = wave & 1, cls = wave & 3;
    const int l31 = lane & 31, lhi = lane >> 5;
#ifdef P9S_PRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
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
#ifdef P9S_TRACE   // debug build: cycle stamps of wave 0 of one workgroup (tools/debug/p9us_trace.py), see igemm_p9s.h
    unsigned long long trc_[40];
#if defined(P9US_TRACE_WAVES) || defined(P9US_TRACE_STEPS)   // every wave's lane 0 stamps S stages 1 and 2: [wave*8 + {start, stored, after barrier 1, issued}] x 2
    const bool tr_on = nt == 1000 && mt == 0 && lane == 0;
#else
    const bool tr_on = nt == 1000 && mt == 0 && t == 0;
#endif
    int tr_n = 0;
    for (int i = 0; i < 40; ++i) trc_[i] = 0;
#define JP_UTR() do { if (tr_on && tr_n < 40) trc_[tr_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define JP_UTR() do { } while (0)
#endif
    JP_UTR();
    const int tiles_x = W / 64, tiles_y = H / TR;
    const int img = nt / (tiles_x * tiles_y), tr_ = nt - img * (tiles_x * tiles_y);
    const int y0 = (tr_ / tiles_x) * TR, x0c = (tr_ % tiles_x) * 64;
    const int MT = M / 128;
    const long HW = (long)H * W;
    const int h2 = H / 2, w2 = W / 2;
    const int NS0 = C0 / 16, NS1 = C1 / 16;

    // ---- weight streams (byte offsets into one buffer resource)
    const int TS = NS0 * 9 * SBYTES, TU = NS1 * 4 * SBYTES, TD = 9 * SBYTES;
    const int offS = mt * TS, offU = MT * TS + (cls * MT + mt) * TU, offD = MT * TS + 4 * MT * TU + mt * TD;
    const int wbytes = MT * TS + 4 * MT * TU + MT * TD + SBYTES;          // + one step of slack for the last prefetch
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(wp), 0, wbytes, 0x00020000);
    const int avo = (lhi * 128 + wm * 64 + l31) * 16;
    jp_u32x4 ra[2][2][3];
    auto aload = [&](int slot, int byte_off) {
#ifdef P9US_PROBE_W   // timing probe (wrong results): every step re-reads the stream's first bytes -- weight latency / bandwidth out of the picture
        const int so = __builtin_amdgcn_readfirstlane(byte_off & 0);
#else
        const int so = __builtin_amdgcn_readfirstlane(byte_off);
#endif
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 3; ++s)
#ifdef P9S_PROBE_AHALF   // timing probe (wrong results): half the weight-stream bytes through the vector memory pipe
                ra[slot][i][s] = i ? ra[slot][0][s] : __builtin_amdgcn_raw_buffer_load_b128(wrs, avo + s * (2 * 128 * 16), so, 0);
#else
                ra[slot][i][s] = __builtin_amdgcn_raw_buffer_load_b128(wrs, avo + i * 512 + s * (2 * 128 * 16), so, 0);
#endif
    };

    // ---- staging registers (union of the three stage kinds): item = 8 channels of one patch pixel
    float rv[NQS][8];
    const float* xs = x0 + (long)img * C0 * HW;
    const float* xu = x1 + (long)img * C1 * h2 * w2;
    const float* xd = x2 ? x2 + (long)img * C2 * HW : nullptr;
    // S / D: item e -> (column 0..65, patch row, k-half); reflection resolved here
    // (addresses = wave-uniform channel base (SGPRs, one per k) + a per-lane 32-bit offset that never changes: nothing for the
    // compiler to hoist into 64-bit address registers)
    unsigned sS[NQS];
    int lS[NQS];
#pragma unroll
    for (int q = 0; q < NQS; ++q) {
        const int e = t + NT * q;
        const int col = e % COLS_S, rp = e / COLS_S, pr = rp % PRS, kh = rp / PRS;
        const int yy = jp_reflect(y0 - 1 + pr, H), xx = jp_reflect(x0c - 1 + col, W);
        sS[q] = e < ITS ? (unsigned)(kh * 8 * HW + (long)yy * W + xx) : 0u;
        lS[q] = e < ITS ? (kh * PRS + pr) * PITS + (col & 1) * PHALF + (col >> 1) : -1;
    }
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xs), 0, (int)((long)C0 * HW * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xu), 0, (int)((long)C1 * h2 * w2 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xd ? xd : xs), 0, (int)((long)(xd ? C2 : 1) * HW * 4), 0x00020000);
    auto gloadS = [&](const __amdgpu_buffer_rsrc_t& rs, int ch0, int nch) {     // ch0: first channel of the stage; nch valid channels
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            // the scalar offset must stay wave-uniform (a lane-dependent one makes the compiler wrap every load in a
            // waterfall loop): channel k of the k-half 0 exists iff k < nch, lanes of k-half 1 only when nch == 16
            const int ub = __builtin_amdgcn_readfirstlane(k < nch ? (int)((long)(ch0 + k) * HW * 4) : 0);
#pragma unroll
            for (int q = 0; q < NQS; ++q) {
                const int kh8 = (t + NT * q) / (COLS_S * PRS) * 8;
                const bool ok = lS[q] >= 0 && kh8 + k < nch;
                const float v = jp_gather(rs, ok ? sS[q] * 4u : 0u, ub);
                rv[q][k] = ok ? v : 0.f;
            }
        }
    };
    // U: item t -> (column 0..33, patch row, k-half) of the low-resolution map; edge clamp
    const int i0 = y0 / 2, j0 = x0c / 2;
    unsigned sU;
    int lU;
    {
        const int col = t % PITU, rp = t / PITU, pr = rp % PRU, kh = rp / PRU;
        const int ii = min(max(i0 - 1 + pr, 0), h2 - 1), jj = min(max(j0 - 1 + col, 0), w2 - 1);
        sU = t < ITU ? (unsigned)((long)kh * 8 * h2 * w2 + (long)ii * w2 + jj) : 0u;
        lU = t < ITU ? (kh * PRU + pr) * PITU + col : -1;
    }
    auto gloadU = [&](int ch0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ub = __builtin_amdgcn_readfirstlane((int)((long)(ch0 + k) * h2 * w2 * 4));
            const float v = jp_gather(rsU, sU * 4u, ub);
            rv[0][k] = lU >= 0 ? v : 0.f;
        }
    };
    auto store_item = [&](jp_u32x4* patch, int q, int loff, int plane) {
        jp_u32x4 w0, w1, w2_;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned a, b, c;
            jp_split3(rv[q][2 * k], rv[q][2 * k + 1], a, b, c);
            w0[k] = a; w1[k] = b; w2_[k] = c;
        }
        patch[loff] = w0;
        patch[2 * plane + loff] = w1;
        patch[4 * plane + loff] = w2_;
    };
    auto lstoreS = [&](int buf = 0) {
#pragma unroll
        for (int q = 0; q < NQS; ++q)
            if (lS[q] >= 0) store_item(patch + buf * BUFW, q, lS[q], PLS);
    };
    auto lstoreU = [&](int buf = 0) {
        if (lU >= 0) store_item(patch + buf * BUFW, 0, lU, PLU);
    };
    // stage k of the whole sequence [S x NS0][U x NS1][D x (C2 ? 1 : 0)]
    const int NSTG = NS0 + NS1 + (C2 ? 1 : 0);
    auto gload_stage = [&](int k) {
        if (k < NS0) gloadS(rsS, k * 16, 16);
        else if (k < NS0 + NS1) gloadU((k - NS0) * 16);
        else gloadS(rsD, 0, C2);
    };
    auto lstore_stage = [&](int k, int buf) {
        if (k >= NS0 && k < NS0 + NS1) lstoreU(buf); else lstoreS(buf);
    };

    jp_f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- B fragment bases (16-byte words).  S / D: pixel (row py + 2j + ty, patch column 2*l31 + px + tx) -> de-interleaved
    // position ((px+tx)&1)*PHALF + l31 + ((px+tx)>>1): taps tx = 0, 2 share a base (+0 / +1), tap tx = 1 has its own.
    const jp_u32x4* bsA = patch + (lhi * PRS + py) * PITS + l31 + px * PHALF;                 // u = px (+2 -> +1)
    const jp_u32x4* bsB = patch + (lhi * PRS + py) * PITS + l31 + (px ? 1 : PHALF);           // u = px + 1
    const jp_u32x4* bu = patch + (lhi * PRU + py) * PITU + l31 + px;
    // B fragments of pixel row j, [split]: each half (j = 0, 1) is re-read just in time -- row j of the NEXT use is requested
    // while the 12 MFMAs of the other row run (24 registers instead of a 48-register double buffer)
    jp_u32x4 rb[NJ][3];
    auto breadS = [&](int buf, int j, int tap) {
#ifdef P9S_PROBE_NOB     // timing probe (wrong results): B fragments read once per stage
        if (tap) return;
#endif
        const int ty = tap / 3, tx = tap % 3;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int o = buf * BUFW + s * 2 * PLS + (2 * j + ty) * PITS;
            rb[j][s] = tx == 1 ? bsB[o] : bsA[o + (tx == 2 ? 1 : 0)];
        }
    };
    auto breadU = [&](int buf, int j, int sl) {
#ifdef P9S_PROBE_NOB
        if (sl) return;
#endif
        const int r = sl >> 1, sx = sl & 1;
#pragma unroll
        for (int s = 0; s < 3; ++s) rb[j][s] = bu[buf * BUFW + s * 2 * PLU + (j + r) * PITU + sx];
    };
#define JP_P9US_MFMA(J_, SA_, SB_)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                          \
        acc[i][J_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(jp_bf16x8, ra[(PAR + u) & 1][i][SA_]),      \
                                                             __builtin_bit_cast(jp_bf16x8, rb[J_][SB_]), acc[i][J_], 0, 0, 0)
#define JP_P9US_ROW(J_)                                                                                                    \
    JP_P9US_MFMA(J_, 2, 0); JP_P9US_MFMA(J_, 1, 1); JP_P9US_MFMA(J_, 0, 2);                                                \
    JP_P9US_MFMA(J_, 1, 0); JP_P9US_MFMA(J_, 0, 1); JP_P9US_MFMA(J_, 0, 0)
    // one stage of T steps (9 taps or 4 slots); weights of step u live in ring slot (PAR + u) & 1; `cur` = byte offset of the
    // stage's first step, `nxt` = first step of the stage that follows (prefetched by the last step)
    // (DB: `k` = the stage's index in the whole sequence, BUF = its patch buffer = k & 1)
    auto run_stage = [&](auto par_tag, auto up_tag, int cur, int nxt, auto buf_tag, int k) {
        constexpr int PAR = decltype(par_tag)::value;
        constexpr bool UP = decltype(up_tag)::value;
        constexpr int T = UP ? 4 : 9;
        constexpr int BUF = DB ? decltype(buf_tag)::value : 0;
        constexpr int LSU = T - 3;                       // the step behind whose MFMAs the next stage's patch is stored
        if (UP) breadU(BUF, 0, 0); else breadS(BUF, 0, 0);
#pragma unroll
        for (int u = 0; u < T; ++u) {
#ifdef P9US_TRACE_STEPS
            if (!UP && k == 2) JP_UTR();
#endif
            aload((PAR + u + 1) & 1, u + 1 < T ? cur + (u + 1) * SBYTES : nxt);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                // the fragments of the row after this one are requested before this row's 12 MFMAs issue; the last row of a
                // step requests row 0 of the next step (rb[0] is free again by then)
                if (j + 1 < NJ) { if (UP) breadU(BUF, j + 1, u); else breadS(BUF, j + 1, u); }
                else if (u + 1 < T) { if (UP) breadU(BUF, 0, u + 1); else breadS(BUF, 0, u + 1); }
                __builtin_amdgcn_sched_barrier(0);
                JP_P9US_ROW(j);                          // the six products with split index sum <= 2, smallest terms first
                __builtin_amdgcn_sched_barrier(0);
            }
            if (DB && u == LSU && k + 1 < NSTG) {
                lstore_stage(k + 1, BUF ^ 1);
                if (k + 2 < NSTG) gload_stage(k + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using UPF = std::false_type;
    using UPT = std::true_type;

    // ---- stage sequence: S x NS0 (NS0 even: the ring parity is 0 again after every pair), U x NS1, D x (C2 ? 1 : 0)
    aload(0, NS0 ? offS : offU);
    if (NS0) gloadS(rsS, 0, 16); else gloadU(0);
    if (DB) {
        lstore_stage(0, 0);
        if (1 < NSTG) gload_stage(1);
        __syncthreads();
    }
    auto s_stage = [&](auto par_tag, int st) {
#ifdef P9US_TRACE_STEPS
        if (st == 2) JP_UTR();
#elif defined(P9US_TRACE_WAVES)
        if (st == 1 || st == 2) JP_UTR();
#else
        if (st < 4 || st + 2 >= NS0) JP_UTR();                    // S stages 0..3 and the last two: start
#endif
        if (!DB) {
#ifdef P9S_PROBE_NOSTG   // timing probe (wrong results): no staging after the first stage (the barriers stay)
            if (st == 0) lstoreS();
#else
            lstoreS();
#endif
#ifdef P9US_TRACE_STEPS
            if (st == 2) JP_UTR();
            __syncthreads();
            if (st == 2) JP_UTR();
#elif defined(P9US_TRACE_WAVES)
            if (st == 1 || st == 2) JP_UTR();
            __syncthreads();
            if (st == 1 || st == 2) JP_UTR();
#else
            if (st < 4) JP_UTR();                                 // ... input arrived, split and stored
            __syncthreads();
#endif
#ifndef P9S_PROBE_NOSTG
            if (st + 1 < NS0) gloadS(rsS, (st + 1) * 16, 16);
            else if (NS1) gloadU(0);
            else if (C2) gloadS(rsD, 0, C2);
#endif
        }
        const int cur = offS + st * 9 * SBYTES;
        const int nxt = st + 1 < NS0 ? cur + 9 * SBYTES : (NS1 ? offU : offD);
        run_stage(par_tag, UPF{}, cur, nxt, par_tag, st);          // NS0 even: stage st's buffer = st & 1 = its ring parity tag
#ifdef P9US_TRACE_STEPS
        if (st == 2) JP_UTR();
        __syncthreads();
        if (st == 2) JP_UTR();
        return;
#elif defined(P9US_TRACE_WAVES)
        if (st == 1 || st == 2) JP_UTR();
#else
        if (st < 4) JP_UTR();                                     // ... step loop issued
#endif
        __syncthreads();
    };
    for (int st = 0; st < NS0; st += 2) {         // 9 steps per stage: the ring parity alternates, a stage pair restores it
        s_stage(P0{}, st);
        s_stage(P1{}, st + 1);
    }
    auto u_stage = [&](auto buf_tag, int st) {
#if !defined(P9US_TRACE_WAVES) && !defined(P9US_TRACE_STEPS)
        if (st < 4) JP_UTR();                                     // U stages 0..3: start
#endif
        if (!DB) {
#ifndef P9S_PROBE_NOSTG
            lstoreU();
#endif
#if !defined(P9US_TRACE_WAVES) && !defined(P9US_TRACE_STEPS)
            if (st < 4) JP_UTR();
#endif
            __syncthreads();
#ifndef P9S_PROBE_NOSTG
            if (st + 1 < NS1) gloadU((st + 1) * 16);
            else if (C2) gloadS(rsD, 0, C2);
#endif
        }
        const int cur = offU + st * 4 * SBYTES;
        run_stage(P0{}, UPT{}, cur, st + 1 < NS1 ? cur + 4 * SBYTES : offD, buf_tag, NS0 + st);
#if !defined(P9US_TRACE_WAVES) && !defined(P9US_TRACE_STEPS)
        if (st < 4) JP_UTR();
#endif
        __syncthreads();
    };
    if (DB) {                                     // NS0, NS1 even (host-checked): U stage st sits in buffer st & 1
        for (int st = 0; st < NS1; st += 2) {
            u_stage(P0{}, st);
            u_stage(P1{}, st + 1);
        }
    } else {
        for (int st = 0; st < NS1; ++st) u_stage(P0{}, st);
    }
    if (C2) {
        if (!DB) {
            lstoreS();
            __syncthreads();
        }
        run_stage(P0{}, UPF{}, offD, offD + 9 * SBYTES, P0{}, NS0 + NS1);
    }
#undef JP_P9US_MFMA
#undef JP_P9US_ROW
#if !defined(P9US_TRACE_WAVES) && !defined(P9US_TRACE_STEPS)
    JP_UTR();                                                     // K loop done
#endif

    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int m0 = mt * 128;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
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
#if !defined(P9US_TRACE_WAVES) && !defined(P9US_TRACE_STEPS)
    JP_UTR();                                                     // epilogue done
#endif
#ifdef P9S_TRACE
#ifdef P9US_TRACE_STEPS      // waves 0, 1 (older) and 4, 5 (younger): 16 stamps each
    if (tr_on && (wave & 3) < 2)
        for (int i = 0; i < 16; ++i) jp_p9s_trace[((wave >> 2) * 2 + (wave & 3)) * 16 + i] = trc_[1 + i];
#elif defined(P9US_TRACE_WAVES)
    if (tr_on)
        for (int i = 0; i < 8; ++i) jp_p9s_trace[wave * 8 + i] = trc_[1 + i];
#else
    if (tr_on)
        for (int i = 0; i < 40; ++i) jp_p9s_trace[i] = trc_[i];
#endif
#endif
#undef JP_UTR
}
