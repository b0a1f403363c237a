// "P9S" patch kernel: the P9 convolution (3x3 stride-1 pad-1 forward / dgrad main pass, and 1x1) with every fp32 product
// formed on the 16-bit matrix pipe from splits of both operands -- fp32 in, fp32 out, fp32 accumulate, fp32-grade accuracy.
// Two arithmetic schemes share the kernel bodies (JP_NS below): since the second half of round 5 the default is TWO fp16 splits of
// power-of-two-scaled operands and THREE products (jp_split2h; DESIGN 4.6b has the error model and the measurements) -- the
// six-product kernels had reached the power wall of the matrix pipe, so the only way on was fewer products.  What
// follows describes the round-3 scheme (JP_NS == 3: three bf16 splits, six products, exact operands), which -DJP_NS=3 still builds
// and the 7x7 stem / stride-2 weight-gradient kernels still use.
//
// Why.  gfx950 has no TF32/xf32 path and its fp32 MFMA runs at the fp32 VECTOR rate (157 TF), 1/16 of the bf16 MFMA rate
// (2.5 PF dense).  An fp32 value v is the sum of three bf16 values to within 2^-25 |v|:
//     v0 = bf16(v)   r1 = v - v0 (exact)   v1 = bf16(r1)   r2 = r1 - v1 (exact)   v2 = bf16(r2)          (round to nearest)
// and a product a*b = sum_{s,t} a_s * b_t.  Every a_s * b_t is exact in fp32 (8 x 8 significand bits) and the bf16 MFMA
// accumulates in fp32, so issuing the six terms with s + t <= 2
//     a0 b0   a0 b1   a1 b0   a0 b2   a1 b1   a2 b0
// leaves out a1 b2 + a2 b1 + a2 b2 <= (2^-24 + 2^-24 + 2^-32) |a b|: the same size as the ONE rounding an fp32 FMA applies
// to the running sum it adds the product to.  The sum over the 16 products of one MFMA is formed inside the matrix unit and
// rounded once into the fp32 accumulator -- 6 roundings per 16 reduction elements instead of the FMA chain's 16.
// tests/test_kernels_gpu.py::test_split_product_accuracy_vs_float64 holds the kernel to "no further from the float64
// result than the exact-fp32 P9 kernel is".  6 MFMAs of 32 cycles replace 8 of 64: 2.67x less matrix-pipe time.
//
// Structure (same tiling ideas as igemm_p9.h): a workgroup owns TR = WN*NJ pixel rows x 32 columns and 64*WM output
// channels.  Per stage of CS = 16*KGS reduction channels the (TR+2) x 34 input patch is staged ONCE: 8 channels of one
// pixel are gathered by one thread (8 coalesced dword loads), split, and written as three 16-byte LDS words
//     patch[split][k-half (8 channels)][patch row][column]  = 8 bf16
// so that a lane's B operand of `v_mfma_f32_32x32x16_bf16` (pixel = lane & 31, 8 consecutive k = channels of k-half
// lane >> 5) for ANY tap is one aligned ds_read_b128 at a compile-time offset.  The weights are split once per step by the
// pack (PACK_SPLIT, conv.hip) into MFMA fragment order [M tile][step = (stage, tap, 16-channel group)][split][k-half][row]
// x 16 bytes and stream from L2 with one buffer_load_dwordx4 per (row block, split) per step, one step ahead.
// Per step a wave issues 6 weight loads + 3*NJ LDS reads for 12*NJ MFMAs; a workgroup meets 2 barriers per stage.
#pragma once
#include "igemm.h"

typedef __bf16 jp_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 jp_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned jp_u32x4 __attribute__((ext_vector_type(4)));

// JP_NS = operand splits per fp32 value.  3: three bf16 splits, six products (exact operands).  2: two fp16 splits of the operand scaled by
// a power of two (the caller's, from the tensor's largest magnitude), three products a0 b0 + a0 b1 + a1 b0 -- see jp_split2h below.
#ifndef JP_NS
#define JP_NS 2
#endif

// three-way bf16 split of a pair of floats -> packed words {lo = x, hi = y} of split 0, 1, 2 (round to nearest even)
__device__ __forceinline__ void jp_split3(float x, float y, unsigned& s0, unsigned& s1, unsigned& s2) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v = {x, y};
    const unsigned h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
    v[0] = x - __uint_as_float(h0 << 16);
    v[1] = y - __uint_as_float(h0 & 0xffff0000u);
    const unsigned h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
    v[0] -= __uint_as_float(h1 << 16);
    v[1] -= __uint_as_float(h1 & 0xffff0000u);
    const unsigned h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
    s0 = h0; s1 = h1; s2 = h2;
}

// two-way fp16 split of a pair of floats scaled by the power of two `sc` (sc * largest magnitude of the tensor in (2^15, 65504]):
//     h0 = fp16(sc x)    h1 = fp16(sc x - h0)          (both round to nearest; the subtraction is exact)
// sc x = h0 + h1 to within 2^-23 |sc x| while h1 is a normal fp16, i.e. for every element within 2^-17 of the tensor's largest; below
// that h1 is subnormal and the error is 2^-25 ABSOLUTE = 2^-40 of the largest magnitude.  The three products a0 b0 + a0 b1 + a1 b0 are
// exact in fp32 and leave out a1 b1 <= 2^-22 |a b|: measured against float64 on layer-shaped data this is 3-4x BELOW the rounding error
// of the fp32 accumulation itself (tools/split_study.py), so the result is as far from float64 as an all-fp32 kernel's, to within ~5 %.
__device__ __forceinline__ void jp_split2h(float x, float y, float sc, unsigned& s0, unsigned& s1) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v = {x * sc, y * sc};
    const h2 a = __builtin_convertvector(v, h2);
    const f2 af = __builtin_convertvector(a, f2);
    f2 r = {v[0] - af[0], v[1] - af[1]};
    const h2 b = __builtin_convertvector(r, h2);
    s0 = __builtin_bit_cast(unsigned, a);
    s1 = __builtin_bit_cast(unsigned, b);
}

// exponent k of the power-of-two operand scale 2^k that puts a tensor's largest magnitude `amax` into [2^14, 2^15) (fp16 tops out at
// 65504); 0 for an all-zero tensor.  jp_amag (below) is what the reductions feed on: Inf / NaN and finite magnitudes of 2^100 and more
// do not take part in `amax` -- they overflow fp16 under the scale of the rest and are treated like Inf (outputs that read them: NaN).
// k is clamped to 126 (tensors whose largest magnitude is below 2^-112 are scaled by 2^126: still >= 22 significant bits down to 2^-126).
__device__ __forceinline__ int jp_scale_exp(float amax) {
    const unsigned u = __float_as_uint(amax) & 0x7fffffffu;
    const int k = 14 - ((int)(u >> 23) - 127);
    return u == 0 ? 0 : min(126, k);
}
__device__ __forceinline__ float jp_exp2i(int k) { return __uint_as_float((unsigned)(127 + k) << 23); }
// products of a K step, smallest terms first: M_(split of A, split of B)
#if JP_NS == 2
#define JP_SPLIT_PRODUCTS(M_) M_(1, 0); M_(0, 1); M_(0, 0)
#else
#define JP_SPLIT_PRODUCTS(M_) M_(2, 0); M_(1, 1); M_(0, 2); M_(1, 0); M_(0, 1); M_(0, 0)
#endif
constexpr int JP_PACK_HDR = JP_NS == 2 ? 4 : 0;     // words in front of a split pack: {scale, 1 / scale, 0, 0} of the weights
// split a pair of floats into the JP_NS packed 16-bit pairs the matrix pipe consumes (s[2] unused for JP_NS == 2)
__device__ __forceinline__ void jp_split_ns(float x, float y, float sc, unsigned (&s)[3]) {
#if JP_NS == 2
    jp_split2h(x, y, sc, s[0], s[1]);
    s[2] = 0;
#else
    jp_split3(x, y, s[0], s[1], s[2]);
#endif
}

// gather load: SGPR buffer resource + per-lane byte offset (a loop-invariant 32-bit VGPR) + wave-uniform byte offset (SGPR):
// no 64-bit per-lane address arithmetic for the compiler to hoist out of the stage loop (it spilled dozens of pointer pairs)
__device__ __forceinline__ float jp_gather(const __amdgpu_buffer_rsrc_t& r, unsigned lane_bytes, int uniform_bytes) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, lane_bytes, uniform_bytes, 0));
}

#ifdef P9S_TRACE    // debug build only: cycle stamps of wave 0 of one workgroup (tools/debug/p1_trace.py)
__device__ unsigned long long jp_p9s_trace[64];
#define JP_TR(i_) do { if (TAPS == 1 && tr_on) trc_[(i_)] = __builtin_readcyclecounter(); } while (0)
#else
#define JP_TR(i_) do { } while (0)
#endif
// Epilogues that offer put4(St first_pixel, int m, float4 v) -- four CONSECUTIVE pixels of output channel m -- get the
// TRANSPOSED accumulator tile (round 4): the MFMA operands are swapped (pixels as rows, channels as columns; the products and
// the k order are the same, so the sums are bit-identical), a lane then holds 4 consecutive pixels of ONE channel per register
// quad and the tile leaves as 16-byte stores -- a quarter of the store instructions of the scalar epilogue, which cost a
// 256 x 128 tile 11 300 cycles of nothing but store issue (profiles/r04_p1_trace.log).
template <class E, class = void>
struct jp_has_put4 : std::false_type {};
template <class E>
struct jp_has_put4<E, std::void_t<decltype(&E::put4)>> : std::true_type {};
// Epilogues with an `amax` member (FwdEpi) can report the largest magnitude of what they store (put_get / put4_get return it):
// the next convolution's operand scale without another pass over the tensor (jp_amax_out, scale.hip).
template <class E, class = void>
struct jp_has_amax : std::false_type {};
template <class E>
struct jp_has_amax<E, std::void_t<decltype(&E::put_get)>> : std::true_type {};
// Epilogues with a `stats` member (FwdEpi): BatchNorm statistics of the stored tile (sum, sum of squares per channel) as partials
template <class E, class = void>
struct jp_has_stats : std::false_type {};
template <class E>
struct jp_has_stats<E, std::void_t<decltype(std::declval<E&>().stats)>> : std::true_type {};
template <bool SWAP>
__device__ __forceinline__ jp_f32x16 jp_mfma_bf16_sw(jp_u32x4 a, jp_u32x4 b, jp_f32x16 c) {
#if JP_NS == 2
    if constexpr (SWAP)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(jp_f16x8, b), __builtin_bit_cast(jp_f16x8, a), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(jp_f16x8, a), __builtin_bit_cast(jp_f16x8, b), c, 0, 0, 0);
#else
    if constexpr (SWAP)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(jp_bf16x8, b), __builtin_bit_cast(jp_bf16x8, a), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(jp_bf16x8, a), __builtin_bit_cast(jp_bf16x8, b), c, 0, 0, 0);
#endif
}

constexpr int P9S_AHEAD = 2;          // steps of slack behind every M tile's weight stream in the pack: the deepest prefetch of any kernel
constexpr int P9S_AH3 = 2;            // steps of weight prefetch of the 3x3 kernels in the three-product build (1 vs 2: < 1 %, profiles/r05_fp16x2_lookahead_ab.log; the switch is gone)

// WM x WN waves; wave (wm, wn) owns channels [64 wm, +64) of the M tile and pixel rows [NJ wn, +NJ) of the tile.
// TAPS = 9 (3x3, one-pixel halo) or 1 (1x1).  KGS = 16-channel groups per stage.
#ifndef P9S_DB
#define P9S_DB 1           // double-buffered patch, one barrier per stage (round 4); 0: the two-barrier stage of round 3
#endif
// XS = input stride (1x1 only): output pixel (y, x) reads input pixel (XS*y, XS*x) of an (XS*H) x (XS*W) map.
// MASK (small maps, igemm_p9sm kernels below): H / W need not be multiples of the tile -- partial tiles stage zeros outside
// the map and store only pixels inside it -- and the workgroup runs the stage range [s_begin, s_end) of the reduction only
// (split-K over grid.z for launches whose tile grid cannot fill the chip; partial sums through a slice epilogue).
// ROWB: B fragments are re-read row by row just in time (12*NJ registers) instead of double-buffered per step (24*NJ): what
// lets a wave hold NJ = 4 pixel rows (8 accumulators) in 230-250 registers -- the 1x1 "wide" tiles of round 4, which halve the
// L2 weight-stream traffic per MFMA (a wave's A fragments serve 4 pixel rows instead of 2).
template <int WM, int WN, int NJ, bool REFLECT, bool REV, class Epi, int TAPS, int KGS, int XS, bool MASK = false,
          bool ROWB = false>
__device__ __forceinline__ void jp_igemm_p9s_body(
    const unsigned* __restrict__ wp, const float* __restrict__ x, Epi epi, int M, int C, int NST, int H, int W, int mt_off,
    const float* __restrict__ xam, int s_begin = 0, int s_end = -1) {
    if (!MASK) { s_begin = 0; s_end = NST; }
    constexpr int NS = JP_NS;
    // JP_NS == 2: operand scales.  Input: from its largest magnitude *xam (jp_amax_of, scale.hip); weights: the pack's header
    float xsc = 1.f, osc = 1.f;
    if constexpr (NS == 2) {
        const int kx = __builtin_amdgcn_readfirstlane(jp_scale_exp(jp_slot_amax(xam)));
        xsc = jp_exp2i(kx);
        osc = jp_exp2i(-kx) * __uint_as_float(__builtin_amdgcn_readfirstlane(wp[1]));
        wp += JP_PACK_HDR;
    }
    constexpr int NT = 64 * WM * WN;
    static_assert(TAPS == 9 || TAPS == 1, "3x3 or 1x1");
    static_assert(XS == 1 || TAPS == 1, "strided input: 1x1 only");
    constexpr int HALO = TAPS == 9 ? 1 : 0;
    constexpr int TR = WN * NJ, PR = TR + 2 * HALO, COLS = 32 + 2 * HALO;
    constexpr int KH = 2 * KGS;                               // k-halves (8 channels each) per stage
    constexpr int CS = 16 * KGS;
    constexpr int PLANE = PR * COLS;                          // 16-byte words per (split, k-half)
    // (round 5: gathering / splitting / storing the 1x1 kernels' patch by waves 4-7 alone -- so that the older half's in-order vmcnt queue
    // holds weight loads only -- was measured: 0.561-0.581 -> 0.584-0.599 ms, profiles/r05_p1_halfstg_ab.log; not kept)
    constexpr int ITEMS = KH * PLANE, NQ = (ITEMS + NT - 1) / NT;
    constexpr int STEPS = TAPS * KGS;                         // (tap, group) steps per stage
    constexpr int BMT = 64 * WM;
    constexpr int LSU = STEPS >= 3 ? STEPS - 3 : 0;           // P9S_DB: the step behind whose MFMAs the next stage's patch is stored
    // transposed accumulators + 16-byte stores: the 4-wave 3x3 kernels only.  Same box, alone (profiles/r04_vec4_ab.log): <1,4> wide
    // 64->64 @256^2 0.225 / 0.216 -> 0.208 / 0.204 ms, <2,2> wide 128->128 @128^2 0.176 / 0.172 -> 0.166 / 0.167; the 8-wave 3x3
    // kernels do not care (2.469 -> 2.480) and the 1x1 kernels, whose tile is mostly stores, LOSE 2-6 % (a 16-byte store per lane
    // covers an eighth of a 128-byte line; the scalar stores of 32 adjacent lanes cover it whole).
    constexpr bool VEC = jp_has_put4<Epi>::value && !MASK && TAPS == 9 && WM * WN <= 4;
    // P9S_DB (round 4): the patch is double-buffered.  A stage used to be [split + LDS store, barrier, MFMAs, barrier]: the cycle
    // stamps of a 1x1 tile (profiles/r04_p1_trace.log) show ~2 300 cycles per stage with no MFMA in flight (drain, barrier, store,
    // barrier) next to 3 100 (1x1) / 13 800 (3x3) cycles of MFMA issue.  Now stage s + 1's patch is split and stored into the other
    // buffer underneath the MFMAs of stage s (its loads were requested a whole stage earlier) and a stage ends in ONE barrier.
    constexpr int BUFW = NS * KH * PLANE;                     // 16-byte words per patch buffer
    // (not where two buffers would cost the 4-wave kernels their second workgroup per CU: the 16x32-pixel tiles of <1, 4> wide)
    constexpr bool DB = P9S_DB != 0 && (WM * WN == 8 || 2 * BUFW * 16 <= 80 * 1024);
    __shared__ jp_u32x4 patch[(DB ? 2 : 1) * BUFW];

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
#ifdef P9S_TRACE
    unsigned long long trc_[40];
    const bool tr_on = nt == 2000 && mt == 0 && t == 0;
    for (int i = 0; i < 40; ++i) trc_[i] = 0;
#endif
    JP_TR(0);
    const int tiles_x = MASK ? (W + 31) / 32 : W / 32, tiles_y = MASK ? (H + TR - 1) / TR : H / TR;
    const int img = nt / (tiles_x * tiles_y), tr_ = nt - img * (tiles_x * tiles_y);
    const int y0 = (tr_ / tiles_x) * TR, x0 = (tr_ % tiles_x) * 32;
    const int m0 = mt * BMT;
    const long HW = (long)H * W, HWI = HW * XS * XS;
    const float* xin = x + (long)img * C * HWI;

    // ---- staging map: item e = t + NT*q -> (k-half, patch row, column); source offset relative to the stage's first
    // channel (or -1: zero), LDS word index
    unsigned soff[NQ];                                       // byte offset inside the image, bit 0 set = zero (padding / no item)
    int loff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = t + NT * q;
        const int col = e % COLS, rp = e / COLS, pr = rp % PR, kh = rp / PR;
        int yy = y0 - HALO + pr, xx = x0 - HALO + col;
        // partial tiles: positions beyond the one-pixel ring around the map feed only masked outputs -- zero, never reflected
        const bool ring = !MASK || (yy <= H && xx <= W);
        if (REFLECT) { yy = jp_reflect(yy, H); xx = jp_reflect(xx, W); }
        const bool ok = e < ITEMS && ring && yy >= 0 && yy < H && xx >= 0 && xx < W;
        soff[q] = ok ? (unsigned)(kh * 8 * HWI + (long)(yy * XS) * (W * XS) + xx * XS) * 4u : 1u;
        loff[q] = e < ITEMS ? (kh * PR + pr) * COLS + col : -1;
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0, (int)((long)C * HWI * 4), 0x00020000);
    constexpr int PF = 1;                                     // stages of the input in flight (2 was measured for the 1x1 kernels: no change,
                                                              // profiles/r04_p9s_pf_ab.log)
    float rv_[PF][NQ][8];
    auto gload = [&](int slot, int stage) {
        float (&rv)[NQ][8] = rv_[slot];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ub = __builtin_amdgcn_readfirstlane((int)(((long)stage * CS + k) * HWI * 4));
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float v = jp_gather(xrs, soff[q] & ~1u, ub);
                rv[q][k] = (soff[q] & 1u) ? 0.f : v;
            }
        }
    };
    auto lstore = [&](int buf, int slot) {
        float (&rv)[NQ][8] = rv_[slot];
        jp_u32x4* patch_ = patch + buf * BUFW;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (loff[q] < 0) continue;
            jp_u32x4 w0, w1, w2;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned sp[3];
                jp_split_ns(rv[q][2 * k], rv[q][2 * k + 1], xsc, sp);
                w0[k] = sp[0]; w1[k] = sp[1]; w2[k] = sp[2];
            }
            patch_[loff[q]] = w0;
            patch_[KH * PLANE + loff[q]] = w1;
            if constexpr (NS == 3) patch_[2 * KH * PLANE + loff[q]] = w2;
        }
    };

    jp_f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- weight stream of this M tile: step u (global over stages) = [split][k-half][row] x 16 B; lane (l31, lhi) of row
    // block i reads [s][lhi][wm*64 + i*32 + l31].  SGPR buffer resource + constant per-lane offset + scalar step offset.
    constexpr int SBYTES = NS * 2 * BMT * 16;                 // bytes per step
    // steps of weight prefetch (register ring of AH + 1 slots).  One step ahead was a whole step of 12-24 MFMAs x 2 waves when a product
    // cost six MFMAs; with three it is 400-800 cycles, less than a loaded L2 round trip: the 3x3 kernels (9 steps per stage: the ring slot
    // of a step is its index mod 3 in every stage) request two steps ahead
    constexpr int AH = (NS == 2 && TAPS == 9 && KGS == 1) ? P9S_AH3 : 1;
    constexpr int RING = AH + 1;
    static_assert(AH <= P9S_AHEAD && (RING == 2 || STEPS % RING == 0), "ring slot of a step must not depend on the stage");
    const long tile_bytes = ((long)NST * STEPS + P9S_AHEAD) * SBYTES;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(wp)) + (long)(mt + mt_off) * tile_bytes, 0, (int)tile_bytes, 0x00020000);
    const int avo = (lhi * BMT + wm * 64 + l31) * 16;
    jp_u32x4 ra[RING][2][NS];
    auto aload = [&](int slot, int step_bytes) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < NS; ++s)
                ra[slot][i][s] = __builtin_amdgcn_raw_buffer_load_b128(wrs, avo + i * 512 + s * (2 * BMT * 16), step_bytes, 0);
    };
#pragma unroll
    for (int d = 0; d < AH; ++d) aload(d, (s_begin * STEPS + d) * SBYTES);
    const jp_u32x4* bp = patch + (lhi * PR + wn * NJ) * COLS + l31;

    if constexpr (ROWB) {
        // B fragments of pixel row j, [split]: each row is re-read just in time -- row j of the next use is requested while the
        // MFMAs of the other rows run (12*NJ registers instead of a double buffer of 24*NJ)
        jp_u32x4 rb[NJ][NS];
        auto bload = [&](int buf, int j, int u) {
            const int tap = u / KGS, kg = u % KGS;
            const int dy = TAPS == 1 ? 0 : (REV ? 2 - tap / 3 : tap / 3), dx = TAPS == 1 ? 0 : (REV ? 2 - tap % 3 : tap % 3);
#pragma unroll
            for (int s = 0; s < NS; ++s) rb[j][s] = bp[buf * BUFW + s * KH * PLANE + (kg * 2 * PR + j + dy) * COLS + dx];
        };
#define JP_P9S_MFMA_ROW(J_, SA_, SB_)                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                        \
            acc[i][J_] = jp_mfma_bf16_sw<VEC>(ra[(PAR * STEPS + u) % RING][i][SA_], rb[J_][SB_], acc[i][J_])
        auto run_stage = [&](auto par_tag, auto buf_tag, int stage) {
            constexpr int PAR = decltype(par_tag)::value, BUF = DB ? decltype(buf_tag)::value : 0;
            if (!DB) {
                lstore(0, 0);
                __syncthreads();
                if (stage + 1 < s_end) gload(0, stage + 1);
            }
            const int ab = __builtin_amdgcn_readfirstlane(stage * STEPS * SBYTES);
            bload(BUF, 0, 0);
#pragma unroll
            for (int u = 0; u < STEPS; ++u) {
                aload((PAR * STEPS + u + AH) % RING, ab + (u + AH) * SBYTES);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (j + 1 < NJ) bload(BUF, j + 1, u);
                    __builtin_amdgcn_sched_barrier(0);
#define JP_P9S_ROWJ(SA_, SB_) JP_P9S_MFMA_ROW(j, SA_, SB_)
                    JP_SPLIT_PRODUCTS(JP_P9S_ROWJ);
#undef JP_P9S_ROWJ
                    __builtin_amdgcn_sched_barrier(0);
                    if (j + 1 == NJ && u + 1 < STEPS) bload(BUF, 0, u + 1);
                }
                if (DB && u == LSU && stage + 1 < s_end) {          // next stage's patch -> the other buffer, under the MFMAs just issued
                    lstore(BUF ^ 1, (BUF + 1) % PF);
                    if (stage + 1 + PF < s_end) gload((BUF + 1) % PF, stage + 1 + PF);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
        };
        static_assert(RING == 2 || STEPS % RING == 0, "two stage parities <-> two ring phases (or none)");
        gload(0, s_begin);
        if (DB) {
            lstore(0, 0);
#pragma unroll
            for (int d = 1; d <= PF; ++d)
                if (s_begin + d < s_end) gload(d % PF, s_begin + d);
            __syncthreads();
        }
        for (int stage = s_begin; stage < s_end; stage += 2) {
            run_stage(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, stage);
            if (stage + 1 < s_end) run_stage(std::integral_constant<int, (STEPS & 1)>{}, std::integral_constant<int, 1>{}, stage + 1);
        }
    } else {
        // B fragments of step u: [j][split], compile-time LDS offsets (the step loop is fully unrolled)
        jp_u32x4 rb[2][NJ][NS];
        auto bload = [&](int buf, int slot, int u) {
            const int tap = u / KGS, kg = u % KGS;
            const int dy = TAPS == 1 ? 0 : (REV ? 2 - tap / 3 : tap / 3), dx = TAPS == 1 ? 0 : (REV ? 2 - tap % 3 : tap % 3);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int s = 0; s < NS; ++s) rb[slot][j][s] = bp[buf * BUFW + s * KH * PLANE + (kg * 2 * PR + j + dy) * COLS + dx];
        };
#define JP_P9S_MFMA(SA_, SB_)                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j)                         \
            acc[i][j] = jp_mfma_bf16_sw<VEC>(ra[(PAR * STEPS + u) % RING][i][SA_], rb[u & 1][j][SB_], acc[i][j])

        // one stage; PAR = stage parity (compile-time): the weight ring slot of step u is (global step) % RING and STEPS may be odd
        auto run_stage = [&](auto par_tag, auto buf_tag, int stage) {
            constexpr int PAR = decltype(par_tag)::value, BUF = DB ? decltype(buf_tag)::value : 0;
            JP_TR(2 + 4 * (stage & 7));
            if (!DB) {
                lstore(0, 0);
                JP_TR(3 + 4 * (stage & 7));
                __syncthreads();
                JP_TR(4 + 4 * (stage & 7));
                if (stage + 1 < s_end) gload(0, stage + 1);         // next stage's patch: in flight during the MFMAs below
            }
            const int ab = __builtin_amdgcn_readfirstlane(stage * STEPS * SBYTES);
            bload(BUF, 0, 0);
#pragma unroll
            for (int u = 0; u < STEPS; ++u) {
                // operands of step u + 1 are requested before the MFMAs of step u issue: weights of step u + AHEAD (the stream
                // continues into the next stage; the pack carries AHEAD steps of slack), B fragments of step u + 1
                aload((PAR * STEPS + u + AH) % RING, ab + (u + AH) * SBYTES);
                if (u + 1 < STEPS) bload(BUF, (u + 1) & 1, u + 1);
                __builtin_amdgcn_sched_barrier(0);
                // the six products with split index sum <= 2, smallest terms first; consecutive MFMAs go to different accumulators
                // (the issue order does not matter to the matrix pipe: tools/ubench/mfma_bf16_chain.hip measures 86-90 % of 2.5 PF for
                // dependent chains, round robin and six-in-a-row alike -- and the compiler's scheduler interleaves them anyway)
                JP_SPLIT_PRODUCTS(JP_P9S_MFMA);
                __builtin_amdgcn_sched_barrier(0);
                if (DB && u == LSU && stage + 1 < s_end) {          // next stage's patch -> the other buffer, under the MFMAs just issued
                    JP_TR(3 + 4 * (stage & 7));
                    lstore(BUF ^ 1, (BUF + 1) % PF);
                    if (stage + 1 + PF < s_end) gload((BUF + 1) % PF, stage + 1 + PF);
                    JP_TR(4 + 4 * (stage & 7));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            JP_TR(5 + 4 * (stage & 7));
            __syncthreads();
        };
        static_assert(RING == 2 || STEPS % RING == 0, "two stage parities <-> two ring phases (or none)");
        gload(0, s_begin);
        JP_TR(1);
        if (DB) {
            lstore(0, 0);
#pragma unroll
            for (int d = 1; d <= PF; ++d)
                if (s_begin + d < s_end) gload(d % PF, s_begin + d);
            __syncthreads();
        }
        for (int stage = s_begin; stage < s_end; stage += 2) {
            run_stage(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, stage);
            if (stage + 1 < s_end) run_stage(std::integral_constant<int, (STEPS & 1)>{}, std::integral_constant<int, 1>{}, stage + 1);
        }
    }
    JP_TR(34);
#undef JP_P9S_MFMA
#undef JP_P9S_MFMA_ROW

    if constexpr (NS == 2) {        // undo the operands' power-of-two scales (exact)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= osc;
    }
    if constexpr (VEC && jp_has_stats<Epi>::value) {
        // BatchNorm statistics of what this wave is about to store (convolutions that feed a train-mode BatchNorm: no bias, no
        // activation, so the stored value IS the accumulator): lane l31 owns channel m, its 16 * NJ registers are that channel's
        // pixels of this half (lhi) of the tile.  stats[(m * 2 + {0, 1}) * parts + part], part = pixel tile * WN + pixel-row wave:
        // every (channel, part) is written exactly once -- no atomics, folded in a fixed order by bn.hip's bn_stats_fold_kernel.
        if (epi.stats) {
            const size_t parts = (size_t)gridDim.x * WN, part = (size_t)nt * WN + wn;     // (nt: this workgroup's pixel tile after the XCD remap)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const float v = acc[i][j][r]; s1 += v; s2 = fmaf(v, v, s2); }
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                const int m = m0 + wm * 64 + i * 32 + l31;
                if (lhi == 0 && m < M) {
                    epi.stats[((size_t)m * 2 + 0) * parts + part] = s1;
                    epi.stats[((size_t)m * 2 + 1) * parts + part] = s2;
                }
            }
        }
    }
    float omx = 0.f;                     // largest |stored value| of this lane (epilogues that report it)
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if constexpr (VEC) {
        // transposed tile: col = output channel, row = pixel -> register quad k holds pixels 8k + 4*lhi + {0..3} of the row
        static_assert(std::is_integral<typename Epi::St>::value, "put4: the column state is an element offset");
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int p = img * (int)HW + (y0 + wn * NJ + j) * W + x0 + 4 * lhi;
            const typename Epi::St se = epi.col(p);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = m0 + wm * 64 + i * 32 + l31;
                if (m >= M) continue;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 v4 = make_float4(acc[i][j][4 * k], acc[i][j][4 * k + 1], acc[i][j][4 * k + 2], acc[i][j][4 * k + 3]);
                    if constexpr (jp_has_amax<Epi>::value) omx = fmaxf(omx, epi.put4_get(se + 8 * k, m, v4));
                    else epi.put4(se + 8 * k, m, v4);
                }
            }
        }
    } else {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (MASK && (y0 + wn * NJ + j >= H || x0 + l31 >= W)) continue;
        const int p = img * (int)HW + (y0 + wn * NJ + j) * W + x0 + l31;
        const typename Epi::St se = epi.col(p);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if constexpr (jp_has_amax<Epi>::value) { if (m < M) omx = fmaxf(omx, jp_fmag(epi.put_get(se, m, acc[i][j][r]))); }
                else { if (m < M) epi.put(se, m, acc[i][j][r]); }
            }
        }
    }
    }
    if constexpr (jp_has_amax<Epi>::value) jp_wave_amax_commit(omx, epi.amax);
#ifdef P9S_TRACE
    JP_TR(35);
    if (TAPS == 1 && tr_on)
        for (int i = 0; i < 40; ++i) jp_p9s_trace[i] = trc_[i];
#endif
}

constexpr bool P9S_OCC1 = true;       // the 8-wave 1x1 kernels with <= 128 registers (two workgroups per CU) in the three-product build (0.467 -> 0.385 ms alone, step unchanged: settled, the switch is gone)
template <int WM, int WN, int NJ, bool REFLECT, bool REV, class Epi, int TAPS, int KGS>
__global__ __launch_bounds__(64 * WM * WN, (P9S_OCC1 && JP_NS == 2 && TAPS == 1 && WM * WN == 8) ? 4 : (NJ <= 2 ? 2 : 1)) void jp_igemm_p9s_kernel(
    const unsigned* __restrict__ wp, const float* __restrict__ x, Epi epi, int M, int C, int NST, int H, int W, int mt_off,
    const float* __restrict__ xam) {
    jp_igemm_p9s_body<WM, WN, NJ, REFLECT, REV, Epi, TAPS, KGS, 1, false, (P9S_OCC1 && JP_NS == 2 && TAPS == 1 && WM * WN == 8)>(
        wp, x, epi, M, C, NST, H, W, mt_off, xam);
}
// "wide" tiles (round 4): NJ = 4 pixel rows per wave (8 rows x 32 columns per workgroup with WN = 2), B fragments re-read row
// by row (ROWB): a wave's weight fragments serve twice the pixels, i.e. half the L2 -> CU weight-stream bytes per MFMA, and a 3x3
// patch carries 10 rows for 8 instead of 6 for 4
template <int WM, int WN, bool REFLECT, bool REV, class Epi, int TAPS, int KGS>
__global__ __launch_bounds__(64 * WM * WN, 2) void jp_igemm_p9s_wide_kernel(
    const unsigned* __restrict__ wp, const float* __restrict__ x, Epi epi, int M, int C, int NST, int H, int W, int mt_off,
    const float* __restrict__ xam) {
    jp_igemm_p9s_body<WM, WN, 4, REFLECT, REV, Epi, TAPS, KGS, 1, false, true>(wp, x, epi, M, C, NST, H, W, mt_off, xam);
}
// small maps (pose encoder 24x80 .. 6x20, BEV 32x32 .. 8x8): masked partial tiles + split-K over grid.z; the epilogue's
// `slice` member receives blockIdx.z (conv_p9sm.hip)
template <int WM, int WN, int NJ, bool REFLECT, bool REV, class Epi, int TAPS, int KGS>
__global__ __launch_bounds__(64 * WM * WN, 2) void jp_igemm_p9sm_kernel(
    const unsigned* __restrict__ wp, const float* __restrict__ x, Epi epi, int M, int C, int NST, int H, int W, int sps,
    const float* __restrict__ xam) {
    const int s_begin = blockIdx.z * sps;
    epi.slice = blockIdx.z;
    jp_igemm_p9s_body<WM, WN, NJ, REFLECT, REV, Epi, TAPS, KGS, 1, true>(wp, x, epi, M, C, NST, H, W, 0, xam, s_begin,
                                                                          min(NST, s_begin + sps));
}
// 1x1 stride-2 (the ResNet downsample branches): same tiles, the staging gather reads every second input pixel
template <int WM, int WN, int NJ, class Epi>
__global__ __launch_bounds__(64 * WM * WN, NJ <= 2 ? 2 : 1) void jp_igemm_p9s_x2_kernel(
    const unsigned* __restrict__ wp, const float* __restrict__ x, Epi epi, int M, int C, int NST, int H, int W, int mt_off,
    const float* __restrict__ xam) {
    jp_igemm_p9s_body<WM, WN, NJ, false, false, Epi, 1, 2, 2>(wp, x, epi, M, C, NST, H, W, mt_off, xam);
}
