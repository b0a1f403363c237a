// "P1L" kernel (round 5): the 1x1 stride-1 convolutions with 256-row banks (CRP / reduce layers of the depth decoder,
// layers.py:147-167, 184-199: forward, and dgrad through the transposed bank) on the bf16 matrix pipe -- fp32 in / out /
// accumulate, every product as six bf16 MFMA products of exact three-way operand splits (igemm_p9s.h) -- as a PERSISTENT
// kernel whose weight fragments come through LDS.
//
// Why a kernel of its own.  In the patch kernel (igemm_p9s.h, TAPS = 1) a 256 x 128-pixel tile is 384 MFMAs per wave for a
// 128 KB gather and a 128 KB store; PMC: matrix pipe busy 0.40-0.43 (3x3 layers 0.75).  Two structural reasons, both measured:
//   * vector-memory loads return IN ORDER (one vmcnt counter): every stage's input gather -- an HBM access, ~2.5 us -- sits in
//     front of the next steps' weight-fragment loads, so each 16-channel step's `s_waitcnt` for its L2-resident weights is a wait
//     for HBM.  With 2 steps per stage nothing covers it (profiles/r04_p1_trace.log: ~2 300 MFMA-free cycles per 3 100-cycle stage).
//     Here the weights of a whole stage are copied global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, the pack's fragment
//     order verbatim) one stage ahead and reach the MFMAs through ds_read_b128: the step loop waits on LDS only.  The copies are
//     issued by waves 0-3, the gathers (and the split + LDS store of what they bring) by waves 4-7: neither half's in-order queue
//     mixes the two, the gathers are consumed 2.5 stages after their request.
//   * one workgroup per CU, one tile per workgroup: prologue (first gather: 20 %) and epilogue (15 %) of every tile are exposed.
//     Here a workgroup walks a range of tiles as ONE sequence of stages: the next tile's first stages are requested, split and
//     stored underneath the current tile's last ones.
// Also: weight fragments through LDS are shared by the two waves of a channel block (half the L2 -> CU bytes; the L2 weight stream
// is the largest power item of these kernels: profiles/r05_p9us2_harness_pmc.log, "noa"), every LDS read / copy / gather / split
// piece sits in the shadow of an MFMA pair (igemm_p9us2.h), and the four accumulators are walked round robin.
// Tile: 8 waves = 4 channel blocks (wm) x 2 pixel-row pairs (wn); wave = 64 channels x 2 rows x 32 pixels (same as the patch
// kernel: same products in the same order, bit-identical results).  LDS: 2 x 24 KB patch + 2 x 48 KB weights = 144 KB.
// Preconditions (host-checked): M % 256 == 0, C % 128 == 0 (stages of 32 channels, in fours), H % 4 == 0, W % 32 == 0, N*C*H*W*4 < 2^31; pack = PACK_SPLIT with
// 256-row tiles and KGS = 2 (conv.hip: p9s_ws_floats).
#pragma once
#include "igemm_p9s.h"
#ifdef P1L_TRACE
__device__ unsigned long long jp_p1l_trace[96];
#endif
#ifndef P1L_PF
#define P1L_PF 2          // stages of input gathers in flight (2 or 3; 3 spills 12 registers and measured no faster)
#endif

template <class Epi>
__global__ __launch_bounds__(512, 2) void jp_conv1x1_p1l_kernel(const unsigned* __restrict__ wp, const float* __restrict__ x, Epi epi,
                                                               int M, int C, int NST, int H, int W, int ntiles, int tiles_per_wg,
                                                               int x_bytes) {
    constexpr int BMT = 256, PR = 4, COLS = 32, KH = 4;
    constexpr int PLANE = PR * COLS;                          // 16-byte words per (split, k-half)
    constexpr int BUFW = 3 * KH * PLANE;                      // words per patch buffer (24 KB)
    constexpr int SBYTES = 3 * 2 * BMT * 16;                  // bytes per weight step (24 KB); 2 steps per stage
    constexpr int ABYTES = 2 * SBYTES;
    // four DISTINCT LDS objects: the compiler orders an LDS read behind every LDS-DMA copy it cannot prove disjoint from it
    // (vmcnt wait in front of the ds_read); reads of one weight buffer must not wait for the copy into the other
    __shared__ __attribute__((aligned(16))) unsigned char abuf0[ABYTES], abuf1[ABYTES];
    __shared__ jp_u32x4 patch0[BUFW], patch1[BUFW];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
#ifdef P1L_TRACE          // debug build: cycle stamps of waves 0 and 4 of workgroup 7 (tools/ubench/p1l_bench.hip prints them)
    unsigned long long trc_[48];
    int tr_n = 0;
    const bool tr_on = blockIdx.x == 7 && blockIdx.y == 0 && lane == 0 && (wave & 3) == 0;
#define JP_LTR() do { if (tr_on && tr_n < 48) trc_[tr_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define JP_LTR() do { } while (0)
#endif
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int mt = blockIdx.y;
    const int T0 = blockIdx.x * tiles_per_wg, T1 = min(ntiles, T0 + tiles_per_wg);
    if (T0 >= T1) return;
    const int tiles_x = W / 32, tiles_img = tiles_x * (H / PR);
    const long HW = (long)H * W;
    const int m0 = mt * BMT;

    // ---- division of the memory work between the two halves of the workgroup (the arbiter serves the matrix pipe strictly by age:
    // waves 0-3 compute first, waves 4-7 starve meanwhile and compute while the others wait at the barrier):
    //   waves 0-3: the weight copies (LDS-DMA) -- their vmcnt queue holds nothing but L2-resident copies, so the one wait per stage
    //              in front of the barrier (`vmcnt(0)`) is a wait for a copy requested a whole stage earlier;
    //   waves 4-7: the input -- gather (HBM), split, LDS store; each of their 256 threads owns TWO items (k-halves kh, kh + 2) of the
    //              stage's 512, three stages of gathers in flight (HBM needs ~12 MB in flight chip-wide to stream at full rate:
    //              16 KB per stage and CU x 3 x 256 CUs), and every wait in their queue is a wait for data they are about to split.
    // Vector loads return in order: with copies and gathers in ONE wave's queue (the first version of this kernel) the wait for a copy
    // was a wait for every gather in front of it.
    const int tt = t & 255;
    const int g_col = tt & 31, g_pr = (tt >> 5) & 3, g_kh = tt >> 7;                 // item q: k-half g_kh + 2 q
    const unsigned g_lane = (unsigned)(g_kh * 8 * HW + g_pr * W + g_col) * 4u;
    const int g_lds = (g_kh * PR + g_pr) * COLS + g_col;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
    auto tile_base = [&](int T) __attribute__((always_inline)) -> int {                      // byte offset of (image, channel 0, first pixel) of tile T
        const int img = T / tiles_img, r = T - img * tiles_img;
        return (int)((((long)img * C) * HW + (long)(r / tiles_x) * PR * W + (r % tiles_x) * 32) * 4);
    };
    float rv[4][2][8];                                        // [set][item][channel]: up to three stages in flight + the one being split
    auto gather1 = [&](int set, int tb, int s, int n) __attribute__((always_inline)) {       // load n (0..15) = item n >> 3, channel n & 7 of stage s of the tile at tb
        const int q = n >> 3, k = n & 7;
        const int ub = __builtin_amdgcn_readfirstlane(tb + (int)(((long)s * 32 + q * 16 + k) * HW * 4));
        rv[set][q][k] = jp_gather(xrs, g_lane, ub);
    };
    auto piece = [&](int set, int buf, int c) __attribute__((always_inline)) {               // channel pair c & 3 of item c >> 2: split, three 4-byte LDS stores
        const int q = c >> 2, r = c & 3;
        unsigned a_, b_, c_;
        jp_split3(rv[set][q][2 * r], rv[set][q][2 * r + 1], a_, b_, c_);
        unsigned* const d = reinterpret_cast<unsigned*>((buf ? patch1 : patch0) + q * 2 * PLANE + g_lds) + r;
        d[0] = a_;
        d[KH * PLANE * 4] = b_;
        d[2 * KH * PLANE * 4] = c_;
    };

    // ---- weights: LDS-DMA of stage s (of any tile: the same 48 KB) -> weight buffer `buf`; wave w < 4 copies chunks 12 w .. 12 w + 11
    const long tile_wbytes = ((long)NST * 2 + P9S_AHEAD) * SBYTES;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(wp)) + (long)mt * tile_wbytes, 0, (int)tile_wbytes, 0x00020000);
    auto dma1 = [&](int buf, int s, int c) __attribute__((always_inline)) {                  // chunk c (0..11) of this wave
        const int ch = (wave & 3) * 12 + c;
        const int so = __builtin_amdgcn_readfirstlane(s * ABYTES + ch * 1024);
        unsigned char* dst = (buf ? abuf1 : abuf0) + ch * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)dst, 16, lane * 16, so, 0, 0);
    };
    // fragment reads: A [step][split][k-half][row] x 16 B, B [split][k-half][patch row][column] x 16 B
    const int a_lane = (lhi * BMT + wm * 64 + l31) * 16;
    const int b_lane = (lhi * PR + wn * 2) * COLS + l31;
    // Fragment registers: splits 1 and 2 of both operands are single-buffered -- a product order of (a2 b0) (a1 b1) (a0 b2) (a1 b0)
    // (a0 b1) (a0 b0) frees a2 after product 1, b2 after 3, a1 after 4, b1 after 5, and the next step's copy is requested right then --
    // only split 0 (wanted by the first and the last product) is double-buffered by step parity: 64 registers instead of 96, which is
    // what makes room for the second accumulator set.
    jp_u32x4 fa0[2][2], fb0[2][2], fa1[2], fa2[2], fb1[2], fb2[2];
    auto afrag = [&](int par, int i, int s_) __attribute__((always_inline)) -> jp_u32x4& { return s_ == 0 ? fa0[par][i] : (s_ == 1 ? fa1[i] : fa2[i]); };
    auto bfrag = [&](int par, int j, int s_) __attribute__((always_inline)) -> jp_u32x4& { return s_ == 0 ? fb0[par][j] : (s_ == 1 ? fb1[j] : fb2[j]); };
    auto aread1 = [&](int par, int buf, int u, int i, int s_) __attribute__((always_inline)) {
        afrag(par, i, s_) = *reinterpret_cast<const jp_u32x4*>((buf ? abuf1 : abuf0) + a_lane + u * SBYTES + s_ * (2 * BMT * 16) + i * 512);
    };
    auto bread1 = [&](int par, int buf, int u, int j, int s_) __attribute__((always_inline)) {
        bfrag(par, j, s_) = (buf ? patch1 : patch0)[b_lane + s_ * KH * PLANE + (u * 2 * PR + j) * COLS];
    };
    // fragment f (0..11) of a step, in the order its registers come free during the step before it
    auto fread = [&](int par, int buf, int u, int f) __attribute__((always_inline)) {
        const int x_ = f & 1;
        switch (f >> 1) {
            case 0: aread1(par, buf, u, x_, 0); break;
            case 1: bread1(par, buf, u, x_, 0); break;
            case 2: aread1(par, buf, u, x_, 2); break;
            case 3: bread1(par, buf, u, x_, 2); break;
            case 4: aread1(par, buf, u, x_, 1); break;
            default: bread1(par, buf, u, x_, 1); break;
        }
    };

    jp_f32x16 acc[2][2];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();
    // The sums leave as 64 scalar stores per lane right behind the tile's last barrier (13 000-23 000 cycles: the write-back of 128 KB
    // per CU is paced by HBM).  Measured and rejected (profiles/r05_p1l_*.log): transposed accumulators + 16-byte stores (a lane's 16
    // bytes are an eighth of a line, a wave's store then touches 32 lines: no gain); a second accumulator set with the stores spread
    // over the next tile's stages (0.50 -> 0.83 ms: stores and loads share the in-order vmcnt queue, so every wait for a gather or a
    // weight copy became a wait for the HBM acknowledgement of the stores in front of it).
    constexpr bool VEC = false;
    auto store_tile = [&](int T) __attribute__((always_inline)) {
        const int img = T / tiles_img, r_ = T - img * tiles_img;
        const int y0 = (r_ / tiles_x) * PR, x0 = (r_ % tiles_x) * 32;
        // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = img * (int)HW + (y0 + wn * 2 + j) * W + x0 + l31;
            const typename Epi::St se = epi.col(p);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    // (the row's address offset does not depend on the tile: without this the compiler hoists all 32 of them -- 64
                    // registers -- out of the tile loop and spills the MFMA loop's operands instead)
                    asm volatile("" : "+v"(m));
                    if (m < M) epi.put(se, m, acc[i][j][r]);
                }
        }
    };
#define JP_P1L_PAIR(SP_, J_, SA_, SB_)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) acc[i][J_] = jp_mfma_bf16_sw<VEC>(afrag(SP_, i, SA_), bfrag(SP_, J_, SB_), acc[i][J_])

    // One stage s of the tile at tb (LDS buffers BUF = s & 1, gather register sets by s & 3; NST % 4 == 0); (tb2, s2) = the stage THREE
    // further on (HBM needs ~12 MB in flight chip-wide to stream at full rate: 16 KB per stage and CU x 3 stages x 256 CUs; with two
    // stages in flight the first version of this kernel ran at half of that), s1 = the next stage:
    //   step 0, behind pair q: A / B fragment q of step 1 (LDS reads);  q 0..5: weight copy chunk q of stage s1 (requested
    //           before any gather of this stage);  q 6..11: gather channels 0..5 of stage (tb2, s2);
    //   step 1, behind pair q: q 0, 1: gather channels 6, 7;  q 3..9: piece q - 3 of the NEXT stage's patch (its gather was requested
    //           two and a half stages ago);
    //   then (stage_end): wait for the copy (vmcnt(8): the 8 newest loads -- this stage's gathers -- may stay in flight), barrier, the
    //   first step's fragments of the next stage (12 LDS reads), and behind a tile's last stage its 64 stores per lane.  Past the workgroup's last stage the requests repeat the
    //   last stage (clamped by the caller): harmless, and the stream has no branches.
    auto run_stage = [&](auto role_tag, auto k_tag, int s1, int tb2, int s2) __attribute__((always_inline)) {
        constexpr int ROLE = decltype(role_tag)::value;                // 0: waves 0-3 (weight copies), 1: waves 4-7 (input)
        constexpr int K = decltype(k_tag)::value, BUF = K & 1;         // K = stage index mod 4
        constexpr int SET_R = (K + 1) & 3, SET_W = (K + P1L_PF) & 3;        // gather sets: the next stage's (split now), stage + 3's (requested now)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const int p_ = q >> 1, j = q & 1;
                // the six products with split index sum <= 2, smallest terms first; rows alternate: four accumulators round robin
                if (p_ == 0) { JP_P1L_PAIR(u, j, 2, 0); }
                else if (p_ == 1) { JP_P1L_PAIR(u, j, 1, 1); }
                else if (p_ == 2) { JP_P1L_PAIR(u, j, 0, 2); }
                else if (p_ == 3) { JP_P1L_PAIR(u, j, 1, 0); }
                else if (p_ == 4) { JP_P1L_PAIR(u, j, 0, 1); }
                else { JP_P1L_PAIR(u, j, 0, 0); }
                if (u == 0) fread(1, BUF, 1, q);                      // fragments of step 1, each as its registers come free
                if (ROLE == 0) {
                    if (u == 0) dma1(BUF ^ 1, s1, q);                  // the next stage's weights: 12 x 1 KB per wave
                } else {
                    // step 0, q 4..11: the eight pieces of the next stage's patch (its gathers were requested a stage and a half or more
                    // ago); step 1, q 0..7: the 16 loads of the stage P1L_PF further on (after the pieces: the in-order queue makes a piece's
                    // wait cover every load issued before it)
                    const int n = 12 * u + q;
                    if (n >= 4 && n < 12) piece(SET_R, BUF ^ 1, n - 4);
                    else if (n >= 12 && n < 20) { gather1(SET_W, tb2, s2, 2 * (n - 12)); gather1(SET_W, tb2, s2, 2 * (n - 12) + 1); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // `done_tile` >= 0: the stage was a tile's last.  Its sums leave AFTER the barrier, behind the next stage's fragment requests: the
    // stores are then younger than every load in flight (a store burst in front of the wait below would make it a wait for the
    // gathers issued half a stage ago -- HBM -- and for the stores themselves), and the LDS reads fly underneath them.
    auto stage_end = [&](auto role_tag, auto k_tag, int done_tile) __attribute__((always_inline)) {
        constexpr int BUF = decltype(k_tag)::value & 1;
        // the copying half: its copies of the next stage's weights have landed.  LDS-DMA completion is not something the compiler's
        // wait-count insertion knows this barrier needs, so the wait is explicit: vmcnt(0), expcnt / lgkmcnt unconstrained
        JP_LTR();                                                 // (trace: step loop issued)
        if (decltype(role_tag)::value == 0) __builtin_amdgcn_s_waitcnt(0x0F70);
        JP_LTR();                                                 // (trace: copies landed)
        __syncthreads();
        JP_LTR();                                                 // (trace: barrier passed)
#pragma unroll
        for (int f = 0; f < 12; ++f) fread(0, BUF ^ 1, 0, f);
        __builtin_amdgcn_sched_barrier(0);
        if (done_tile >= 0) {
            store_tile(done_tile);
            zero_acc();
        }
        JP_LTR();                                                 // (trace: next stage's MFMAs begin)
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    // (Every workgroup walks equal tiles at the same rate, so all CUs reach their tile ends together: a chip-wide burst of stores.
    // A start-up skew of 0 / 8 / 16 / 24 thousand cycles between the four quarters of every XCD's workgroups was measured: no
    // difference, profiles/r05_p1l_skew.log.)
    // ---- prologue: weights of stage 0 (waves 0-3); patch of stage 0, gathers of stages 1 and 2 (waves 4-7)
    int tb = tile_base(T0);
    if (wave < 4) {
#pragma unroll
        for (int c = 0; c < 12; ++c) dma1(0, 0, c);
        __builtin_amdgcn_s_waitcnt(0x0F70);
    } else {
#pragma unroll
        for (int n = 0; n < 16; ++n) gather1(0, tb, 0, n);
#pragma unroll
        for (int c = 0; c < 8; ++c) piece(0, 0, c);
#pragma unroll
        for (int n = 0; n < 16; ++n) gather1(1, tb, 1, n);
        if (P1L_PF == 3) {
#pragma unroll
            for (int n = 0; n < 16; ++n) gather1(2, tb, 2, n);
        }
    }
    __syncthreads();
#pragma unroll
    for (int f = 0; f < 12; ++f) fread(0, 0, 0, f);
    __builtin_amdgcn_sched_barrier(0);
    auto tiles_loop = [&](auto role) __attribute__((always_inline)) {
        for (int T = T0; T < T1; ++T) {
            const int tbn = tile_base(min(T + 1, T1 - 1));
            const bool last = T + 1 == T1;
            for (int s = 0; s < NST; s += 4) {
                // (stage three further on: this tile's, or one of the next tile's first three; past the end: the last stage again)
                const bool wrap = s + 4 >= NST;
                auto far = [&](int d, int& tbf, int& sf) __attribute__((always_inline)) {    // stage s + d + P1L_PF
                    const int f = s + d + P1L_PF;
                    if (f < NST) { tbf = tb; sf = f; }
                    else if (!last) { tbf = tbn; sf = f - NST; }
                    else { tbf = tb; sf = NST - 1; }
                };
                int tbf, sf;
                far(0, tbf, sf); run_stage(role, I0{}, s + 1, tbf, sf); stage_end(role, I0{}, -1);
                far(1, tbf, sf); run_stage(role, I1{}, s + 2, tbf, sf); stage_end(role, I1{}, -1);
                far(2, tbf, sf); run_stage(role, I2{}, s + 3, tbf, sf); stage_end(role, I2{}, -1);
                far(3, tbf, sf); run_stage(role, I3{}, wrap ? 0 : s + 4, tbf, sf); stage_end(role, I3{}, wrap ? T : -1);
            }
            tb = tbn;
        }
    };
    if (wave < 4) tiles_loop(I0{}); else tiles_loop(I1{});
#ifdef P1L_TRACE
    if (tr_on)
        for (int i = 0; i < 48; ++i) jp_p1l_trace[(wave >> 2) * 48 + i] = trc_[i];
#endif
#undef JP_LTR
#undef JP_P1L_PAIR
}
