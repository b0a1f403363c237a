// "P9US2" patch kernel: the decoder's iconv layers  y = act(Conv3x3_reflect(cat(skip, up2x(x), disp)) + b)  (depth_decoder.py:76-77)
// in the parity-class form of igemm_p9u.h, every fp32 product as six bf16 MFMA products of exact three-way splits (igemm_p9s.h has the
// arithmetic argument).  Round 5 re-laid the round-3 kernel's instruction stream for how a CDNA4 SIMD issues; the tiling is unchanged:
//
// Workgroup: 8 waves, output tile 128 channels x (4 rows x 64 columns).  Wave (wm, class (py, px)) owns 64 channels x the tile's
// 2 x 32 pixels of its parity class (rows py, py+2; columns px, px+2, ...): a wave is class-uniform, so the upsampled stages read a
// class-specific weight stream while the skip / disparity stages share one.  Stages of 16 channels:
//   S (skip, 9 taps = 9 steps): full-resolution patch 6 x 66, columns stored de-interleaved by parity ([even | odd]) so that a
//       fragment's 32 same-parity pixels are 32 consecutive 16-byte words;
//   U (upsampled, 4 slots = 4 steps): low-resolution patch 4 x 34 (edge clamp == reflection of up(x));
//   D (the disparity channel(s), padded to 16: 9 steps).
// LDS word = 8 channels of one pixel as bf16, [split][k-half][patch row][position]; weights (PACK_SPLITSEG, conv.hip): streams
// [S: M tile][U: class][M tile][D: M tile] of steps, a step = [split][k-half][128 rows] x 16 bytes, one step ahead.
//
// What the round-5 step trace of the round-3 stream showed (profiles/r05_p9us_steps_before.log, cycle stamps per 24-MFMA step):
//   * ONE wave never issued faster than ~44 cycles per MFMA (1 050 cycles per step, with or without a partner on its SIMD),
//     although the matrix pipe takes a `32x32x16` every 32: a step began with a burst of 6 weight loads + 3 LDS reads (+ waits), and
//     an in-order wave issues nothing else while those go out;
//   * the two waves of a SIMD do not share the pipe evenly: the older one (waves 0-3) ran its 216 MFMAs of a stage at that rate while
//     the younger got ~1 slot in 3, then the younger ran alone at the same 44 cycles and the older waited at the barrier.
// So this kernel (a) puts every operand request into the shadow of an MFMA pair -- after every two MFMAs (64 pipe cycles) at most
// one memory instruction and a few VALU are issued, pinned with sched_barrier, never a burst inside the step loop -- and (b) gives
// the two halves of the workgroup different places for the staging of the NEXT stage's patch (double-buffered in LDS): the younger
// half splits + stores it and requests the stage after that at the START of a stage, when the older half owns the pipe anyway; the
// older half spreads split + store over the MFMA gaps of rows 1-5 and requests its gathers at the END of the stage, when it is about
// to wait for the younger half at the barrier (they land during that wait instead of holding up the in-order vmcnt queue in front of
// the next weight loads).  Matrix beside memory on every SIMD, one barrier per stage.
//
// Results are bit-identical to the round-3 stream's (same products, same order per accumulator).
// Measured and not kept (profiles/r05_p9us2_ldsa_ab.log): the S stages' weight fragments through LDS -- LDS-DMA copies of three-step
// chunks issued by the older half, one barrier per chunk, ds_read_b128 fragments shared by the four waves of a channel half (a quarter
// of the L2 -> CU weight bytes) -- gives the same results and the same time (3.87-3.95 vs 3.89-3.91 ms): what the L2 stream costs in
// power the extra LDS traffic, copies and barriers cost again.
// Preconditions (host-checked): Cout % 128 == 0, C0 % 32 == 0, C1 % 32 == 0 (a stage's patch buffer is its index parity),
// C2 <= 8, H % 4 == 0, W % 64 == 0.
#pragma once
#include "igemm_p9s.h"

template <class Epi>
__global__ __launch_bounds__(512, 2) void jp_igemm_p9us2_kernel(const unsigned* __restrict__ wp, const float* __restrict__ x0,
                                                                const float* __restrict__ x1, const float* __restrict__ x2,
                                                                Epi epi, int M, int C0, int C1, int C2, int H, int W,
                                                                const float* __restrict__ xam) {
    constexpr int NS = JP_NS;
    // JP_NS == 2 (two fp16 splits, three products): ONE scale for the three sources -- they meet in one accumulator -- from the largest
    // magnitude over all of them (*xam, jp_amax_of3), the weights' from the pack header of the first segment (all segments carry the same)
    float xsc = 1.f, osc = 1.f;
    if constexpr (NS == 2) {
        const int kx = __builtin_amdgcn_readfirstlane(jp_scale_exp(jp_slot_amax(xam)));
        xsc = jp_exp2i(kx);
        osc = jp_exp2i(-kx) * __uint_as_float(__builtin_amdgcn_readfirstlane(wp[1]));
        wp += JP_PACK_HDR;
    }
    constexpr int NT = 512, NJ = 2;
    constexpr int TR = 2 * NJ;
    constexpr int PRS = TR + 2, PHALF = 34, PITS = 2 * PHALF, COLS_S = 66; // S / D patch rows x [33 even | pad | 33 odd | pad]
    constexpr int PRU = NJ + 2, PITU = 34;                                 // U patch: NJ + 2 rows x [halo | 32 | halo]
    constexpr int PLS = PRS * PITS, PLU = PRU * PITU;                      // 16-byte words per (split, k-half) plane
    constexpr int ITS = 2 * PRS * COLS_S, NQS = (ITS + NT - 1) / NT;       // 792 items -> 2 rounds
    constexpr int ITU = 2 * PRU * PITU;                                    // 272 items -> 1 round
    static_assert(ITU <= NT && NQS == 2, "staging rounds");
    constexpr int SBYTES = NS * 2 * 128 * 16;                              // bytes per weight step
    constexpr int BUFW = NS * 2 * PLS;
    __shared__ jp_u32x4 patch[2 * BUFW];

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
#ifdef P9S_TRACE   // debug build: per-step cycle stamps of S stage 2, waves 0, 1, 4, 5 (tools/debug/p9us_trace_steps.py)
    unsigned long long trc_[16];
    const bool tr_on = nt == 1000 && mt == 0 && lane == 0;
    int tr_n = 0;
    for (int i = 0; i < 16; ++i) trc_[i] = 0;
#define JP_UTR() do { if (tr_on && tr_n < 16) trc_[tr_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define JP_UTR() do { } while (0)
#endif
    const int tiles_x = W / 64, tiles_y = H / TR;
    const int img = nt / (tiles_x * tiles_y), tr_ = nt - img * (tiles_x * tiles_y);
    const int y0 = (tr_ / tiles_x) * TR, x0c = (tr_ % tiles_x) * 64;
    const int MT = M / 128;
    const long HW = (long)H * W;
    const int h2 = H / 2, w2 = W / 2;
    const int NS0 = C0 / 16, NS1 = C1 / 16;
    const int NSTG = NS0 + NS1 + (C2 ? 1 : 0);

    // ---- weight streams (byte offsets into one buffer resource)
    const int TS = NS0 * 9 * SBYTES, TU = NS1 * 4 * SBYTES, TD = 9 * SBYTES;
    const int offS = mt * TS, offU = MT * TS + (cls * MT + mt) * TU, offD = MT * TS + 4 * MT * TU + mt * TD;
    const int wbytes = MT * TS + 4 * MT * TU + MT * TD + SBYTES;          // + one step of slack for the last prefetch
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(wp), 0, wbytes, 0x00020000);
    const int avo = (lhi * 128 + wm * 64 + l31) * 16;
    jp_u32x4 ra[2][2][NS];
    // one of a step's 2 * NS weight fragments: a = NS i + s
    auto aload1 = [&](int slot, int so, int a) {
        const int i = a / NS, s = a % NS;
        ra[slot][i][s] = __builtin_amdgcn_raw_buffer_load_b128(wrs, avo + i * 512 + s * (2 * 128 * 16), so, 0);
    };

    // ---- staging registers (union of the three stage kinds): item = 8 channels of one patch pixel
    float rv[NQS][8];
    const float* xs = x0 + (long)img * C0 * HW;
    const float* xu = x1 + (long)img * C1 * h2 * w2;
    const float* xd = x2 ? x2 + (long)img * C2 * HW : nullptr;
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
    // FULL (all 16 channels of the stage exist: the S stages): the loaded value goes to its register as it is -- a thread without an
    // item never stores it -- so nothing forces a wait for the data where the loads are issued
    auto gloadS = [&](auto full_tag, const __amdgpu_buffer_rsrc_t& rs, int ch0, int nch) {     // ch0: first channel; nch valid channels
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            // (the scalar offset must stay wave-uniform: a lane-dependent one wraps every load in a waterfall loop)
            const int ub = __builtin_amdgcn_readfirstlane(FULL || k < nch ? (int)((long)(ch0 + k) * HW * 4) : 0);
#pragma unroll
            for (int q = 0; q < NQS; ++q) {
                if (FULL) {
                    rv[q][k] = jp_gather(rs, sS[q] * 4u, ub);
                } else {
                    const int kh8 = (t + NT * q) / (COLS_S * PRS) * 8;
                    const bool ok = lS[q] >= 0 && kh8 + k < nch;
                    const float v = jp_gather(rs, ok ? sS[q] * 4u : 0u, ub);
                    rv[q][k] = ok ? v : 0.f;
                }
            }
        }
    };
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
            rv[0][k] = jp_gather(rsU, sU * 4u, ub);
        }
    };
    // split + store of one item, in seven pieces (four channel pairs, three 16-byte words) so that a piece fits an MFMA gap
    jp_u32x4 w0, w1, w2_;
    auto split_pair = [&](int q, int kp) {
        unsigned sq[3];
        jp_split_ns(rv[q][2 * kp], rv[q][2 * kp + 1], xsc, sq);
        w0[kp] = sq[0]; w1[kp] = sq[1]; w2_[kp] = sq[2];
    };
    auto store_word = [&](jp_u32x4* pb, int loff, int plane, int s) {
        if (loff >= 0) pb[2 * s * plane + loff] = s == 0 ? w0 : (s == 1 ? w1 : w2_);
    };
    // piece c of the staging of an S / D stage (2 NPC pieces) or a U stage (NPC pieces) into patch buffer `buf`
    constexpr int NPC = 4 + NS;                      // pieces per item: four channel pairs, NS 16-byte words
    auto piece = [&](bool up, int buf, int c) {
        const int q = c / NPC, r = c % NPC;
        if (up && q) return;
        if (r < 4) split_pair(q, r);
        else store_word(patch + buf * BUFW, up ? lU : lS[q], up ? PLU : PLS, r - 4);
    };
    auto lstore_all = [&](bool up, int buf) {
#pragma unroll
        for (int c = 0; c < 2 * NPC; ++c) piece(up, buf, c);
    };
    // stage k of the whole sequence [S x NS0][U x NS1][D x (C2 ? 1 : 0)]
    auto gload_stage = [&](int k) {
        if (k < NS0) gloadS(std::true_type{}, rsS, k * 16, 16);
        else if (k < NS0 + NS1) gloadU((k - NS0) * 16);
        else gloadS(std::false_type{}, rsD, 0, C2);
    };
    auto lstore_stage = [&](int k, int buf) { lstore_all(k >= NS0 && k < NS0 + NS1, buf); };

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
    // B fragments of a step: [row][split 1, 2] and -- split 0 is wanted by the step's first and last product -- [step parity][row] for split 0
    jp_u32x4 rb0[2][NJ], rb12[NJ][NS - 1];
    // split s of the B fragments of pixel row j for tap / slot tp (sb: the step's parity, selects the split-0 buffer)
    auto bread1 = [&](bool up, int buf, int j, int tp, int s, int sb) {
        jp_u32x4 v;
        if (up) {
            const int r = tp >> 1, sx = tp & 1;
            v = bu[buf * BUFW + s * 2 * PLU + (j + r) * PITU + sx];
        } else {
            const int ty = tp / 3, tx = tp % 3;
            const int o = buf * BUFW + s * 2 * PLS + (2 * j + ty) * PITS;
            v = tx == 1 ? bsB[o] : bsA[o + (tx == 2 ? 1 : 0)];
        }
        if (s == 0) rb0[sb][j] = v; else rb12[j][s - 1] = v;
    };
#define JP_P9US2_PAIR(J_, SA_, SB_)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                          \
        acc[i][J_] = jp_mfma_bf16_sw<false>(ra[(PAR + u) & 1][i][SA_],                                                     \
                                            (SB_) == 0 ? rb0[(PAR + u) & 1][J_] : rb12[J_][(SB_) == 0 ? 0 : (SB_) - 1], acc[i][J_])

    // One stage of T steps (9 taps or 4 slots) x 2 pixel rows x 6 MFMA pairs.  After pair g of row (u, j):
    //   g = 0..2: split g of the B fragments of the NEXT row (the other row's registers: its MFMAs were issued before this row began);
    //   row 0, every g: one sixth of the NEXT step's weight fragments (ring slot (PAR + u + 1) & 1);
    //   row 1, g = 3..5 (older half, steady state): piece 3u + g - 3 of the next stage's patch.
    // ROLE 0 = older half (waves 0-3), 1 = younger half.  NK: kind of the stage that follows (0 S / D, 1 U, 2 none staged here).
    auto run_stage = [&](auto role_tag, auto up_tag, auto par_tag, auto buf_tag, auto nk_tag, int cur, int nxt, int k) __attribute__((always_inline)) {
        constexpr int ROLE = decltype(role_tag)::value;
        constexpr bool UP = decltype(up_tag)::value;
        constexpr int PAR = decltype(par_tag)::value, BUF = decltype(buf_tag)::value, NK = decltype(nk_tag)::value;
        constexpr int T = UP ? 4 : 9;
#ifndef P9US2_NOSTG   // (timing probes, wrong results: P9US2_NOSTG / _NOA / _NOBR drop the staging / weight loads / LDS reads)
        if (ROLE == 1 && k + 1 < NSTG) {                 // younger half: the next stage's patch now, underneath the older half's MFMAs
            lstore_stage(k + 1, BUF ^ 1);
            if (k + 2 < NSTG) gload_stage(k + 2);
        }
#endif
#ifdef P9S_TRACE
        if (!UP && k == 2) JP_UTR();
#endif
        // the first step's fragments: the only LDS reads of a stage that nothing hides (the patch was complete at the barrier)
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < NJ; ++j) bread1(UP, BUF, j, 0, s, PAR & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < T; ++u) {
#ifdef P9S_TRACE
            if (!UP && k == 2) JP_UTR();
#endif
            const int so = __builtin_amdgcn_readfirstlane(u + 1 < T ? cur + (u + 1) * SBYTES : nxt);
            // 6 products x 2 rows x 2 channel blocks: consecutive MFMAs walk the wave's FOUR accumulators, so an accumulator is
            // touched every fourth MFMA (a lone wave that alternates two of them issues one MFMA per ~48 cycles, not per 32:
            // profiles/r05_p9us2_probe_variants.log).  q = 2 * product + row indexes the 12 pairs; after pair q:
            //   q 0..5   the next step's weights, split 2 first (the order the products want them), a full step ahead;
            //   q 0, 1   the next step's split-0 B fragments (other parity buffer);  q 4, 5 / 8, 9: its split-2 / split-1 ones, as
            //            soon as the last product that reads the current ones has issued;
            //   q 7, 10, 11 (older half, steady state): piece 3u + {0, 1, 2} of the next stage's patch.
            if constexpr (NS == 3) {
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const int p_ = q >> 1, j = q & 1;
                // the six products with split index sum <= 2, smallest terms first
                if (p_ == 0) { JP_P9US2_PAIR(j, (NS - 1), 0); }
                else if (p_ == 1) { JP_P9US2_PAIR(j, 1, 1); }
                else if (p_ == 2) { JP_P9US2_PAIR(j, 0, (NS - 1)); }
                else if (p_ == 3) { JP_P9US2_PAIR(j, 1, 0); }
                else if (p_ == 4) { JP_P9US2_PAIR(j, 0, 1); }
                else { JP_P9US2_PAIR(j, 0, 0); }
#ifndef P9US2_NOA
                if (q < 6) aload1((PAR + u + 1) & 1, so, 3 * (q & 1) + 2 - (q >> 1));
#endif
#ifndef P9US2_NOBR
                if (u + 1 < T) {
                    if (q < 2) bread1(UP, BUF, j, u + 1, 0, (PAR + u + 1) & 1);
                    else if (q == 4 || q == 5) bread1(UP, BUF, j, u + 1, NS - 1, 0);
                    else if (q == 8 || q == 9) bread1(UP, BUF, j, u + 1, 1, 0);
                }
#endif
#ifndef P9US2_NOSTG
                if (ROLE == 0 && NK != 2 && (q == 7 || q >= 10)) {
                    const int c = 3 * u + (q == 7 ? 0 : q - 9);
                    if (c < (NK == 1 ? 7 : 14)) piece(NK == 1, BUF ^ 1, c);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            } else {
            // two fp16 splits, three products (1,0) (0,1) (0,0): q = 2 * product + row indexes 6 pairs; after pair q:
            //   q 0..3   the next step's weights, split 1 first;   q 0, 1: its split-0 B fragments (other parity buffer);  q 4, 5: its split-1
            //            ones (the last product that reads the current ones, (0,1), has issued);
            //   q 3, 5 (older half, steady state): piece 2u + {0, 1} of the next stage's patch.
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int p_ = q >> 1, j = q & 1;
                if (p_ == 0) { JP_P9US2_PAIR(j, 1, 0); }
                else if (p_ == 1) { JP_P9US2_PAIR(j, 0, 1); }
                else { JP_P9US2_PAIR(j, 0, 0); }
#ifndef P9US2_NOA
                if (q < 4) aload1((PAR + u + 1) & 1, so, 2 * (q & 1) + 1 - (q >> 1));
#endif
#ifndef P9US2_NOBR
                if (u + 1 < T) {
                    if (q < 2) bread1(UP, BUF, j, u + 1, 0, (PAR + u + 1) & 1);
                    else if (q >= 4) bread1(UP, BUF, j, u + 1, 1, 0);
                }
#endif
#ifndef P9US2_NOSTG
                if (ROLE == 0 && NK != 2 && (q == 3 || q == 5)) {
                    const int c = 2 * u + (q == 3 ? 0 : 1);
                    if (c < (NK == 1 ? NPC : 2 * NPC)) piece(NK == 1, BUF ^ 1, c);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            }
        }
#ifdef P9S_TRACE
        if (!UP && k == 2) JP_UTR();
#endif
#ifndef P9US2_NOSTG
        if (ROLE == 0) {                                 // older half: about to wait for the younger one -- its gathers fly meanwhile
            if (NK == 2 && k + 1 < NSTG) lstore_stage(k + 1, BUF ^ 1);
            if (k + 2 < NSTG) gload_stage(k + 2);
        }
#endif
        __syncthreads();
#ifdef P9S_TRACE
        if (!UP && k == 2) JP_UTR();
#endif
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using UPF = std::false_type;
    using UPT = std::true_type;

    // ---- prologue: weights of the first step, patch of stage 0 stored, stage 1 requested
    {
        const int so = __builtin_amdgcn_readfirstlane(NS0 ? offS : offU);
#pragma unroll
        for (int a = 0; a < 2 * NS; ++a) aload1(0, so, a);
    }
    gload_stage(0);
    lstore_stage(0, 0);
    if (1 < NSTG) gload_stage(1);
    __syncthreads();

    // ---- stage sequence: S x NS0, U x NS1 (both even: a stage's patch buffer and -- 9 steps per S stage -- its ring parity are
    // its index parity), D x (C2 ? 1 : 0).  The last stage of a kind stages its successor (another kind) in one piece; the last
    // pair of a kind is peeled out of the loop (straight-line code: no variant branch inside a loop body), and everything is
    // force-inlined -- with this many stage bodies the inliner otherwise gives up and the register arrays end up in scratch.
    auto k_loop = [&](auto role) __attribute__((always_inline)) {
        int st = 0;
        for (; st + 2 < NS0; st += 2) {
            const int cur = offS + st * 9 * SBYTES;
            run_stage(role, UPF{}, I0{}, I0{}, I0{}, cur, cur + 9 * SBYTES, st);
            run_stage(role, UPF{}, I1{}, I1{}, I0{}, cur + 9 * SBYTES, cur + 18 * SBYTES, st + 1);
        }
        if (NS0) {
            const int cur = offS + st * 9 * SBYTES;
            run_stage(role, UPF{}, I0{}, I0{}, I0{}, cur, cur + 9 * SBYTES, st);
            run_stage(role, UPF{}, I1{}, I1{}, I2{}, cur + 9 * SBYTES, NS1 ? offU : offD, st + 1);
        }
        for (st = 0; st + 2 < NS1; st += 2) {
            const int cur = offU + st * 4 * SBYTES;
            run_stage(role, UPT{}, I0{}, I0{}, I1{}, cur, cur + 4 * SBYTES, NS0 + st);
            run_stage(role, UPT{}, I0{}, I1{}, I1{}, cur + 4 * SBYTES, cur + 8 * SBYTES, NS0 + st + 1);
        }
        if (NS1) {
            const int cur = offU + st * 4 * SBYTES;
            run_stage(role, UPT{}, I0{}, I0{}, I1{}, cur, cur + 4 * SBYTES, NS0 + st);
            run_stage(role, UPT{}, I0{}, I1{}, I2{}, cur + 4 * SBYTES, offD, NS0 + st + 1);
        }
        if (C2) run_stage(role, UPF{}, I0{}, I0{}, I2{}, offD, offD + 9 * SBYTES, NS0 + NS1);
    };
    if (wave < 4) k_loop(I0{}); else k_loop(I1{});
#undef JP_P9US2_PAIR

    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int m0 = mt * 128;
    float omx = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int p = img * (int)HW + (y0 + py + 2 * j) * W + x0c + 2 * l31 + px;
        const typename Epi::St se = epi.col(p);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const float v = NS == 2 ? acc[i][j][r] * osc : acc[i][j][r];
                if constexpr (jp_has_amax<Epi>::value) { if (m < M) omx = fmaxf(omx, jp_fmag(epi.put_get(se, m, v))); }
                else { if (m < M) epi.put(se, m, v); }
            }
        }
    }
    if constexpr (jp_has_amax<Epi>::value) jp_wave_amax_commit(omx, epi.amax);
#ifdef P9S_TRACE
    if (tr_on && (wave & 3) < 2)
        for (int i = 0; i < 16; ++i) jp_p9s_trace[((wave >> 2) * 2 + (wave & 3)) * 16 + i] = trc_[i];
#endif
#undef JP_UTR
}
