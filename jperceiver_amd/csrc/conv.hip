// conv2d forward / dgrad / wgrad for the JPerceiver train step as instances of the fp32-MFMA
// implicit-GEMM engine (igemm.h).  Replaces every nn.Conv2d / ReflectionPad2d+Conv2d call site of
// the reference hot path (resnet.py:6-13,91; layers.py:147-167; depth_decoder.py:15-39;
// pose_decoder.py:9-12; layout_model.py:31-47,138-153; CrossViewTransformer.py:30-42).
//
// Layout: activations NCHW fp32, weights [Cout][Cin][KH][KW] fp32 (the reference's state-dict
// layout, so checkpoints stay interchangeable).
//   forward : M = Cout,  N = batch*OH*OW pixels,  K = taps*Cin    (B gathered with zero/reflect
//             padding, stride, and optional fused nearest-2x-upsample + channel-concat sources)
//   dgrad   : M = Cin,   N = batch*H*W  pixels,   K = taps*Cout   (B gathers dY; the adjoint of
//             reflection padding is folded into the gather, no workspace)
//   wgrad   : M = Cout,  N = taps*Cin,            K = batch*OH*OW (split-K; partial tiles reduced
//             through caller scratch, or fp32 atomics without it)
//
// Loader families (DESIGN.md §4.1 has the measurements behind each):
//   "T"  tap-major fast path: K ordered (channel chunk of 32, tap, channel); tap decode, bounds / reflection and source
//        selection once per chunk; scalar-base addressing (wave-uniform 64-bit base + per-lane 32-bit offset); weights
//        re-packed per launch to [tap][row][channel_pad32] (L2-resident) so the A loads are coalesced.
//   "R3" row-tile kernel for 3x3 stride-1 layers whose pixel tile is one image-row segment: the three dx taps share one
//        staged row segment (forward and dgrad main pass).
//   "P"  parity-class kernels for inputs read through the fused nearest-2x upsample (iconv layers, disparity heads):
//        4 pre-summed weight slots per output parity instead of 9 taps, in forward, dgrad and wgrad.
//   "S2" parity-class dgrad of the 3x3 stride-2 convs; "C" whole-tap K chunks for the 3/6-channel stems.
//   "G"  generic (channel-major K, per-element decode): whatever the fast paths do not cover.
#include "igemm_p9.h"
#include "igemm_p9s.h"
#include "conv_p9sm.h"
#include "igemm_w9s.h"
#include "igemm_w9s2.h"
#include "igemm_p9us2.h"
#if JP_NS == 3
#include "igemm_p1l.h"      // the persistent 1x1 kernel exists in the six-product build only (opt-in there, JP_P1L=1)
#endif
#include "scale.h"
constexpr double JP_NPROD = JP_NS == 2 ? 3.0 : 6.0;     // matrix-pipe products per fp32 product of the P9S-family kernels
#include "igemm_p9sd.h"
#include "igemm_w4s.h"
#include "igemm_p9s2d.h"
#include "igemm_p9s2f.h"
#include "igemm_p7s.h"
#include "igemm_w9.h"
#include "igemm_p9u.h"
#include "igemm_w7.h"
#include <algorithm>
#include <cstdlib>

namespace {

// Masked-out gathers read this word instead of branching around the load: the gather stays a straight run of
// unconditional global loads (no exec-mask save/restore per element).
__device__ float jp_zero_word[4] = {0.f, 0.f, 0.f, 0.f};

// ------------------------------------------------------------------------------------------------------------------
// Weight packing.  The MFMA kernels read their A operand (the weights) from a packed copy whose layout depends on the
// kernel family (5 layouts below).  A pack is a pure function of the weight tensor, so it only has to be redone when
// the weights change -- once per training step, after the optimizer.  Every pack goes through do_pack(), which
//   * launches it right away (ws_state 0 of the conv entry points: the caller's scratch holds nothing yet), and
//   * appends a 64-byte job descriptor to the caller's host buffer while a recording is open (jp_pack_record_begin).
// The host keeps the descriptors of all layers in one device table and refreshes EVERY pack of the model with ONE
// launch of jp_pack_replay per step (conv entry points then run with ws_state 1 = "scratch already packed"): ~310
// tiny launches per step become one.
enum { PACK_TAP = 0, PACK_ROWMAJOR = 1, PACK_SEG = 2, PACK_UP_DGRAD = 3, PACK_FLIP = 4, PACK_FRAG = 5, PACK_FRAGSEG = 6, PACK_SPLIT = 7, PACK_SPLITSEG = 8, PACK_SPLITUPD = 9, PACK_SPLIT7 = 10 };
struct JpPackJob {          // 64 bytes, mirrored by jperceiver_amd/ops.py (struct layout "PPqqi6i")
    const float* w;
    float* wp;
    long total, begin;      // elements of this pack; prefix offset inside a replay table
    int mode, p[6];
};
static_assert(sizeof(JpPackJob) == 64, "JpPackJob layout");
__host__ __device__ inline int jp_cdiv_d(int a, int b) { return (a + b - 1) / b; }

// taps a (class, slot) pair of the parity-class form stands for: dy in [y0, y1], dx in [x0, x1]
__device__ __forceinline__ float pack_slot_sum(const float* wc, int pl) {
    const int a = pl >> 3, b = (pl >> 2) & 1, r = (pl >> 1) & 1, sx = pl & 1;
    // Dy(a, r): a=0 -> {0} / {1,2};  a=1 -> {0,1} / {2}
    const int y0 = a ? (r ? 2 : 0) : (r ? 1 : 0), y1 = a ? (r ? 2 : 1) : (r ? 2 : 0);
    const int x0 = b ? (sx ? 2 : 0) : (sx ? 1 : 0), x1 = b ? (sx ? 2 : 1) : (sx ? 2 : 0);
    float v = 0.f;
    for (int dy = y0; dy <= y1; ++dy)
        for (int dx = x0; dx <= x1; ++dx) v += wc[dy * 3 + dx];
    return v;
}

__device__ __forceinline__ float pack_elem(int mode, const float* __restrict__ w, long i, const int* p, float sc = 1.f) {
    switch (mode) {
        case PACK_TAP: {       // p = Cout, Cin, KHW, Cp, for_dgrad: wp[tap][row][Cp], rows = Cout (fwd) / Cin (dgrad)
            const int Cout = p[0], Cin = p[1], KHW = p[2], Cp = p[3], for_dgrad = p[4];
            const int rows = for_dgrad ? Cin : Cout, red = for_dgrad ? Cout : Cin;
            const int c = (int)(i % Cp);
            const long t = i / Cp;
            const int r = (int)(t % rows), tap = (int)(t / rows);
            if (c >= red) return 0.f;
            const int co = for_dgrad ? c : r, ci = for_dgrad ? r : c;
            return w[((size_t)co * Cin + ci) * KHW + tap];
        }
        case PACK_ROWMAJOR: {  // p = Cout, Cin, KHW, CP, Kp: wp[m][k = tap*CP + c] (zero padded to Kp), stem convs
            const int Cin = p[1], KHW = p[2], CP = p[3], Kp = p[4];
            const int k = (int)(i % Kp), m = (int)(i / Kp);
            const int tap = k / CP, c = k - tap * CP;
            return (tap < KHW && c < Cin) ? w[((size_t)m * Cin + c) * KHW + tap] : 0.f;
        }
        case PACK_SEG: {       // p = Cout, Cin, c_off, C, Cp, up: one channel segment [c_off, c_off + C); full resolution
                               // -> wp[tap][co][Cp]; upsampled -> wp[class][slot][co][Cp] (pre-summed taps)
            const int Cout = p[0], Cin = p[1], c_off = p[2], C = p[3], Cp = p[4], up = p[5];
            const int c = (int)(i % Cp);
            const long t = i / Cp;
            const int co = (int)(t % Cout), pl = (int)(t / Cout);
            if (c >= C) return 0.f;
            const float* wc = w + ((size_t)co * Cin + c_off + c) * 9;
            return up ? pack_slot_sum(wc, pl) : wc[pl];
        }
        case PACK_UP_DGRAD: {  // p = Cout, Cin, c_off, Cx, Cp: wpT[(class, slot)][c][co_pad] (dgrad A of the upsampled segment)
            const int Cout = p[0], Cin = p[1], c_off = p[2], Cx = p[3], Cp = p[4];
            const int co = (int)(i % Cp);
            const long t = i / Cp;
            const int c = (int)(t % Cx), pl = (int)(t / Cx);
            if (co >= Cout) return 0.f;
            return pack_slot_sum(w + ((size_t)co * Cin + c_off + c) * 9, pl);
        }
        case PACK_FRAG: {      // p = Cout, Cin, for_dgrad, BMT, KHW (9 or 1): MFMA fragment order of the P9 / P1 kernel (igemm_p9.h):
                               // wp[M tile][quad Q (+ slack quads)][k parity][row in tile][j] = W(row, reduction channel
                               // chunk*32 + 2s + parity, tap) for k-step 4Q + j = (chunk*KHW + tap)*16 + s: a lane's four
                               // k-steps of a quad are one 16-byte load; zero beyond the last step / row / channel
            const int Cout = p[0], Cin = p[1], for_dgrad = p[2], BMT = p[3], KHW = p[4];
            const int rows = for_dgrad ? Cin : Cout, red = for_dgrad ? Cout : Cin;
            const long per_tile = ((long)((red + 31) / 32) * KHW * 4 + P9_QAHEAD + 1) * 8 * BMT;
            const int mt = (int)(i / per_tile);
            long t = i - (long)mt * per_tile;
            const int j = (int)(t & 3); t >>= 2;
            const int m = mt * BMT + (int)(t % BMT);
            t /= BMT;
            const int par = (int)(t & 1); t >>= 1;          // t = quad
            const long step = t * 4 + j;                    // global k-step
            const int s_ = (int)(step & 15);
            const long tc = step >> 4;                      // chunk*KHW + tap
            const int tap = (int)(tc % KHW);
            const int c = (int)(tc / KHW) * 32 + 2 * s_ + par;
            if (m >= rows || c >= red) return 0.f;
            const int co = for_dgrad ? c : m, ci = for_dgrad ? m : c;
            return w[((size_t)co * Cin + ci) * KHW + tap];
        }
        case PACK_SPLIT: {     // p = Cout, Cin, for_dgrad, BMT, KHW (9 or 1), KGS: split weights (JP_NS planes; i counts behind the pack header) in the fragment order
                               // of the P9S kernel (igemm_p9s.h): wp[M tile][step = (stage, tap, 16-channel group) (+ slack)]
                               // [split][k-half][row][4 words]; word w4 = the bf16 pair of reduction channels
                               // stage*16*KGS + group*16 + khalf*8 + 2*w4 + {0, 1} (low half = the even channel)
            const int Cout = p[0], Cin = p[1], for_dgrad = p[2], BMT = p[3], KHW = p[4], KGS = p[5];
            const int rows = for_dgrad ? Cin : Cout, red = for_dgrad ? Cout : Cin;
            const long nsteps = (long)(red / (16 * KGS)) * KHW * KGS;
            const long per_tile = (nsteps + P9S_AHEAD) * (JP_NS * 8) * BMT;
            const int mt = (int)(i / per_tile);
            long t = i - (long)mt * per_tile;
            const int w4 = (int)(t & 3); t >>= 2;
            const int m = mt * BMT + (int)(t % BMT);
            t /= BMT;
            const int khalf = (int)(t & 1); t >>= 1;
            const int sp = (int)(t % JP_NS);
            const long U = t / JP_NS;
            if (U >= nsteps || m >= rows) return 0.f;
            const int stage = (int)(U / (KHW * KGS)), u = (int)(U % (KHW * KGS));
            const int tap = u / KGS, kg = u % KGS;
            const int c = stage * 16 * KGS + kg * 16 + khalf * 8 + 2 * w4;
            float v[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int cc = c + k;
                const int co = for_dgrad ? cc : m, ci = for_dgrad ? m : cc;
                v[k] = cc < red ? w[((size_t)co * Cin + ci) * KHW + tap] : 0.f;
            }
            unsigned sq[3];
            jp_split_ns(v[0], v[1], sc, sq);
            return __uint_as_float(sp == 0 ? sq[0] : (sp == 1 ? sq[1] : sq[2]));
        }
        case PACK_SPLIT7: {    // p = Cout (<= 64), Cin: 7x7 stem weights as bf16 three-way splits in the fragment order of the P7S kernel
                               // (igemm_p7s.h): [step u = (c, tap-row pair v)][split][k-half][64 rows][4 words]; word w4 of k-half h =
                               // the taps (ky = 2v + h, kx = 2*w4 + {0, 1}) of channel c = u / 4 (ky, kx = 7: zero padding)
            const int Cout = p[0], Cin = p[1];
            const int w4 = (int)(i & 3), m = (int)((i >> 2) & 63), khalf = (int)((i >> 8) & 1);
            const long t = i >> 9;
            const int sp = (int)(t % JP_NS);
            const long U = t / JP_NS;
            const int c = (int)(U >> 2), ky = 2 * (int)(U & 3) + khalf;
            float v[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int kx = 2 * w4 + k;
                v[k] = (c < Cin && m < Cout && ky < 7 && kx < 7) ? w[(((size_t)m * Cin + c) * 7 + ky) * 7 + kx] : 0.f;
            }
            unsigned sq[3];
            jp_split_ns(v[0], v[1], sc, sq);
            return __uint_as_float(sp == 0 ? sq[0] : (sp == 1 ? sq[1] : sq[2]));
        }
        case PACK_SPLITUPD: {  // p = Cout, Cin, c_off, Cx: dgrad weights of the upsampled iconv segment as bf16 three-way splits in the
                               // fragment order of the P9SD kernel (igemm_p9sd.h): [M tile of 128 rows c][step = (16-channel stage
                               // of co, (class, slot) q)][split][k-half][row][4 words]; value = the pre-summed slot weight W'_q[co][c]
            const int Cout = p[0], Cin = p[1], c_off = p[2], Cx = p[3];
            const long nsteps = (long)((Cout + 31) / 32 * 2) * 16;
            const long per_tile = (nsteps + P9S_AHEAD) * (JP_NS * 1024);
            const int mt = (int)(i / per_tile);
            long t = i - (long)mt * per_tile;
            const int w4 = (int)(t & 3); t >>= 2;
            const int c = mt * 128 + (int)(t & 127); t >>= 7;
            const int khalf = (int)(t & 1); t >>= 1;
            const int sp = (int)(t % JP_NS);
            const long U = t / JP_NS;
            if (U >= nsteps || c >= Cx) return 0.f;
            const int stage = (int)(U >> 4), q = (int)(U & 15);
            const int co = stage * 16 + khalf * 8 + 2 * w4;
            float v[2];
#pragma unroll
            for (int k = 0; k < 2; ++k)
                v[k] = co + k < Cout ? pack_slot_sum(w + ((size_t)(co + k) * Cin + c_off + c) * 9, q) : 0.f;
            unsigned sq[3];
            jp_split_ns(v[0], v[1], sc, sq);
            return __uint_as_float(sp == 0 ? sq[0] : (sp == 1 ? sq[1] : sq[2]));
        }
        case PACK_SPLITSEG: {  // p = Cout, Cin, c_off, C, up, hdr_back: one channel segment of an iconv bank as JP_NS-way splits (the scale header
                               // of the bank sits hdr_back words in front of this segment, JP_NS == 2) in the
                               // fragment order of the P9US2 kernel (igemm_p9us2.h): [class (up only)][M tile of 128][step = (16-channel
                               // stage, tap | slot)][split][k-half][row 128][4 words]; word w4 = channels stage*16 + khalf*8 + 2*w4 + {0,1}
            const int Cout = p[0], Cin = p[1], c_off = p[2], C = p[3], up = p[4];
            const int T = up ? 4 : 9, MT = Cout / 128;
            const long tile = (long)((C + 15) / 16) * T * (JP_NS * 1024);
            const int cm = (int)(i / tile);
            if (cm >= (up ? 4 : 1) * MT) return 0.f;            // slack words behind the last stream
            long t = i - (long)cm * tile;
            const int cls = cm / MT, mt = cm - cls * MT;
            const int w4 = (int)(t & 3); t >>= 2;
            const int row = (int)(t & 127); t >>= 7;
            const int khalf = (int)(t & 1); t >>= 1;
            const int sp = (int)(t % JP_NS);
            const int U = (int)(t / JP_NS);
            const int stage = U / T, tap = U - stage * T;
            const int c = stage * 16 + khalf * 8 + 2 * w4, co = mt * 128 + row;
            float v[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float* wc = w + ((size_t)co * Cin + c_off + c + k) * 9;
                v[k] = (c + k < C && co < Cout) ? (up ? pack_slot_sum(wc, cls * 4 + tap) : wc[tap]) : 0.f;
            }
            unsigned sq[3];
            jp_split_ns(v[0], v[1], sc, sq);
            return __uint_as_float(sp == 0 ? sq[0] : (sp == 1 ? sq[1] : sq[2]));
        }
        case PACK_FRAGSEG: {   // p = Cout, Cin, c_off, C, KP, up: one channel segment of an iconv bank in the fragment order of
                               // the P9U kernel (igemm_p9u.h): [class (up only)][M tile of 128][quad][k parity][row][4],
                               // k-step 4*quad + j = (stage, tap | slot, k-pair s) with 2*KP channels per stage
            const int Cout = p[0], Cin = p[1], c_off = p[2], C = p[3], KP = p[4], up = p[5];
            const int T = up ? 4 : 9, MT = Cout / 128;
            const long tile = (long)((C + 2 * KP - 1) / (2 * KP)) * T * KP / 4 * 1024;
            const int cm = (int)(i / tile);
            long t = i - (long)cm * tile;
            const int cls = cm / MT, mt = cm - cls * MT;
            const int j = (int)(t & 3); t >>= 2;
            const int row = (int)(t & 127); t >>= 7;
            const int par = (int)(t & 1); t >>= 1;          // t = quad
            const long step = t * 4 + j;
            const int stage = (int)(step / (T * KP)), rem = (int)(step - (long)stage * (T * KP));
            const int tap = rem / KP, s_ = rem - tap * KP;
            const int c = stage * 2 * KP + 2 * s_ + par, co = mt * 128 + row;
            if (c >= C || co >= Cout) return 0.f;
            const float* wc = w + ((size_t)co * Cin + c_off + c) * 9;
            return up ? pack_slot_sum(wc, cls * 4 + tap) : wc[tap];
        }
        default: {             // PACK_FLIP, p = Cout, Cin, c_off, C: wf[c][co][t] = w[co][c_off + c][8 - t]
            const int Cout = p[0], Cin = p[1], c_off = p[2];
            const int t = (int)(i % 9), co = (int)((i / 9) % Cout), c = (int)(i / (9 * (long)Cout));
            return w[((size_t)co * Cin + c_off + c) * 9 + 8 - t];
        }
    }
}

// Split packs of the fp16 two-way scheme (JP_NS == 2) start with a header {s, 1 / s, 0, 0}: s = the power of two that puts the weight
// tensor's largest magnitude into [2^14, 2^15) (jp_scale_exp); the fragments hold the splits of s * w.  pack_scale_kernel writes the
// headers (one workgroup per job) BEFORE the pack kernels of the same stream read them.
__device__ __forceinline__ int pack_hdr(int mode) { return (mode == PACK_SPLIT || mode == PACK_SPLITUPD || mode == PACK_SPLIT7) ? JP_PACK_HDR : 0; }
// the weight scale a split pack's elements are multiplied by: PACK_SPLIT's own header; PACK_SPLITSEG: the bank's header, p[5] words back
__device__ __forceinline__ float pack_scale_of(const JpPackJob& j) {
    if (!JP_PACK_HDR) return 1.f;
    if (j.mode == PACK_SPLIT || j.mode == PACK_SPLITUPD || j.mode == PACK_SPLIT7) return j.wp[0];
    if (j.mode == PACK_SPLITSEG) return *(j.wp - j.p[5]);
    return 1.f;
}
// Three small launches: zero the reduction word (header word 2) -- PSL workgroups per job reduce a slice of the weight tensor each into it
// (a maximum: order-independent) -- one thread per job turns it into {s, 1 / s}.  (One workgroup per job, as first written, took 2 ms per
// step on the 512 x 512 x 9 tensors: profiles/r05_fp16x2_first_kernel_stats.md.)
constexpr int PSL = 32;
__device__ __forceinline__ float* pack_scale_hdr(const JpPackJob& j, long* n) {
    const bool seg = j.mode == PACK_SPLITSEG && j.p[2] == 0;          // the bank's first segment owns the bank's header
    const bool upd = j.mode == PACK_SPLITUPD;                         // (pre-summed slot weights of the whole tensor: x 4 bound, like seg)
    const bool s7 = j.mode == PACK_SPLIT7;
    if (!JP_PACK_HDR || !(j.mode == PACK_SPLIT || seg || upd || s7)) return nullptr;
    *n = (long)j.p[0] * j.p[1] * ((seg || upd) ? 9 : (s7 ? 49 : j.p[4]));
    return seg ? j.wp - j.p[5] : j.wp;
}
__device__ __forceinline__ void pack_scale_zero(const JpPackJob& j) {
    long n;
    float* hdr = pack_scale_hdr(j, &n);
    if (hdr && threadIdx.x == 0) reinterpret_cast<unsigned*>(hdr)[2] = 0u;
}
__device__ __forceinline__ void pack_scale_reduce(const JpPackJob& j, int slice) {
    long n;
    float* hdr = pack_scale_hdr(j, &n);
    if (!hdr) return;
    const long per = (n + PSL - 1) / PSL, beg = slice * per, end = min(n, beg + per);
    unsigned m = 0;
    for (long i = beg + threadIdx.x; i < end; i += 256) m = max(m, jp_amag(__float_as_uint(j.w[i])));    // largest ordinary magnitude (scale.hip)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(reinterpret_cast<unsigned*>(hdr) + 2, m);
}
__device__ __forceinline__ void pack_scale_finish(const JpPackJob& j) {
    long n;
    float* hdr = pack_scale_hdr(j, &n);
    if (!hdr || threadIdx.x) return;
    // (iconv banks: the upsampled segment's elements are sums of up to four taps -- pack_slot_sum -- and all segments share one scale)
    const int k = jp_scale_exp(hdr[2] * ((j.mode == PACK_SPLITSEG || j.mode == PACK_SPLITUPD) ? 4.f : 1.f));
    hdr[0] = jp_exp2i(k);
    hdr[1] = jp_exp2i(-k);
    hdr[2] = hdr[3] = 0.f;
}
__global__ __launch_bounds__(64) void pack_scale_zero_one_kernel(JpPackJob job) { pack_scale_zero(job); }
__global__ __launch_bounds__(256) void pack_scale_reduce_one_kernel(JpPackJob job) { pack_scale_reduce(job, blockIdx.x); }
__global__ __launch_bounds__(64) void pack_scale_finish_one_kernel(JpPackJob job) { pack_scale_finish(job); }
__global__ __launch_bounds__(64) void pack_scale_zero_kernel(const JpPackJob* __restrict__ jobs, int njobs) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q < njobs) { long n; float* hdr = pack_scale_hdr(jobs[q], &n); if (hdr) reinterpret_cast<unsigned*>(hdr)[2] = 0u; }
}
__global__ __launch_bounds__(256) void pack_scale_reduce_kernel(const JpPackJob* __restrict__ jobs, int njobs) {
    pack_scale_reduce(jobs[blockIdx.y], blockIdx.x);
}
__global__ __launch_bounds__(64) void pack_scale_finish_kernel(const JpPackJob* __restrict__ jobs, int njobs) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q < njobs) {
        long n;
        float* hdr = pack_scale_hdr(jobs[q], &n);
        if (hdr) {
            const int k = jp_scale_exp(hdr[2] * ((jobs[q].mode == PACK_SPLITSEG || jobs[q].mode == PACK_SPLITUPD) ? 4.f : 1.f));
            hdr[0] = jp_exp2i(k);
            hdr[1] = jp_exp2i(-k);
            hdr[2] = hdr[3] = 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void pack_one_kernel(JpPackJob job) {
    const int hdr = pack_hdr(job.mode);
    const float sc = pack_scale_of(job);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < job.total - hdr; i += (long)gridDim.x * 256)
        job.wp[hdr + i] = pack_elem(job.mode, job.w, i, job.p, sc);
}

// every pack of the model in one launch.  The concatenated element range is walked in groups of FOUR consecutive elements
// (the host aligns every job's begin to a multiple of 4, jp_pack_replay contract): one binary search over the prefix
// offsets per group, and for the fragment-order packs -- most of the elements -- one index decode per group (the four
// elements are the four k-steps j of one quad: same tile, row, parity and tap) and one 16-byte store.
__global__ __launch_bounds__(256) void pack_replay_kernel(const JpPackJob* __restrict__ jobs, int njobs, long total) {
    const long groups = (total + 3) >> 2;
    for (long g4 = (long)blockIdx.x * 256 + threadIdx.x; g4 < groups; g4 += (long)gridDim.x * 256) {
        const long g = g4 << 2;
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].begin <= g) lo = mid; else hi = mid - 1;
        }
        const JpPackJob& j = jobs[lo];
        const long i = g - j.begin;
        if (i >= j.total) continue;                         // alignment padding between two jobs
        if (j.mode == PACK_SPLIT || j.mode == PACK_SPLITSEG) continue;      // pack_split_replay_kernel's
        if (j.mode == PACK_FRAG && i + 4 <= j.total) {
            const int* p = j.p;
            const int Cout = p[0], Cin = p[1], for_dgrad = p[2], BMT = p[3], KHW = p[4];
            const int rows = for_dgrad ? Cin : Cout, red = for_dgrad ? Cout : Cin;
            const long per_tile = ((long)((red + 31) / 32) * KHW * 4 + P9_QAHEAD + 1) * 8 * BMT;
            const int mt = (int)(i / per_tile);
            long t = (i - (long)mt * per_tile) >> 2;
            const int m = mt * BMT + (int)(t % BMT);
            t /= BMT;
            const int par = (int)(t & 1); t >>= 1;          // t = quad
            const long step = t * 4;                        // k-steps step .. step+3 share (chunk, tap)
            const int s0 = (int)(step & 15);
            const long tc = step >> 4;
            const int tap = (int)(tc % KHW), c0 = (int)(tc / KHW) * 32 + 2 * s0 + par;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < rows) {
                const size_t cs = for_dgrad ? (size_t)Cin * KHW : (size_t)KHW;      // stride of the reduction channel in w
                const float* wb = j.w + (for_dgrad ? (size_t)m * KHW : (size_t)m * Cin * KHW) + tap;
                if (c0 < red) v.x = wb[(size_t)c0 * cs];
                if (c0 + 2 < red) v.y = wb[(size_t)(c0 + 2) * cs];
                if (c0 + 4 < red) v.z = wb[(size_t)(c0 + 4) * cs];
                if (c0 + 6 < red) v.w = wb[(size_t)(c0 + 6) * cs];
            }
            *reinterpret_cast<float4*>(j.wp + i) = v;
        } else {
            // (split packs with a scale header -- PACK_SPLITUPD, PACK_SPLIT7: the header words belong to the pack_scale kernels, the
            // elements count from behind it and are the splits of scale * w)
            const int hdr = pack_hdr(j.mode);
            const float sc = hdr ? pack_scale_of(j) : 1.f;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i + k >= hdr && i + k < j.total) j.wp[i + k] = pack_elem(j.mode, j.w, i + k - hdr, j.p, sc);
        }
    }
}

// The split packs (PACK_SPLIT / PACK_SPLITSEG, most of the replayed bytes since round 3) through LDS: a work item = (job, M tile,
// 16*KGS-channel stage, chunk of 64 rows).  Its weights -- 64 rows x CS channels x KHW taps, contiguous runs of the weight tensor in
// either orientation -- are read COALESCED into LDS, then every (step, k-half, row) triple is split once and written as three
// 16-byte words; the 64 rows of a (step, split, k-half) are one contiguous 1 KB run of the pack.  (The generic replay decoded
// every 4-byte word on its own and read its two weights with a KHW-float stride: 1.3 ms per step at the head of the step.)
__device__ __forceinline__ int split_job_items(const JpPackJob& j) {
    const int* p = j.p;
    if (j.mode == PACK_SPLIT) {
        const int rows = p[2] ? p[1] : p[0], red = p[2] ? p[0] : p[1];
        return jp_cdiv_d(rows, p[3]) * (red / (16 * p[5])) * (p[3] / 64);
    }
    if (j.mode == PACK_SPLITSEG) return (p[0] / 128) * ((p[3] + 15) / 16) * 2;
    return 0;
}
__global__ __launch_bounds__(256) void pack_split_replay_kernel(const JpPackJob* __restrict__ jobs, int njobs) {
    __shared__ float tile[64 * 16 * 9 + 64];           // [row][channel][tap] (PACK_SPLIT fwd / SEG) or [channel][row][tap] (dgrad)
    constexpr int MAXJ = 2048;
    __shared__ int pre[MAXJ + 1];                       // exclusive prefix of the jobs' work-item counts
    const int t = threadIdx.x;
    for (int q = t; q < njobs && q < MAXJ; q += 256) pre[q + 1] = split_job_items(jobs[q]);
    if (t == 0) pre[0] = 0;
    __syncthreads();
    if (t == 0)
        for (int q = 1; q <= njobs && q <= MAXJ; ++q) pre[q] += pre[q - 1];
    __syncthreads();
    const int nj = min(njobs, MAXJ), total_items = pre[nj];
    for (int item = blockIdx.x; item < total_items; item += gridDim.x) {
        __syncthreads();                                // the tile of the previous item is no longer read
        int lo = 0, hi = nj - 1;                        // last job whose prefix <= item
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (pre[mid] <= item) lo = mid; else hi = mid - 1;
        }
        const int jj = lo, loc = item - pre[lo];
        const JpPackJob& j = jobs[jj];
        const int* p = j.p;
        unsigned* out = reinterpret_cast<unsigned*>(j.wp) + pack_hdr(j.mode);
        if (j.mode == PACK_SPLIT) {
            const float wsc = JP_PACK_HDR ? j.wp[0] : 1.f;
            const int Cout = p[0], Cin = p[1], for_dgrad = p[2], BMT = p[3], KHW = p[4], KGS = p[5];
            const int rows = for_dgrad ? Cin : Cout, red = for_dgrad ? Cout : Cin;
            const int CS = 16 * KGS, nst = red / CS, rch = BMT / 64;
            const int rc = loc % rch, stage = (loc / rch) % nst, mt = loc / (rch * nst);
            const int m0 = mt * BMT + rc * 64, c0 = stage * CS;
            const int RUN = CS * KHW;                   // floats of one row (fwd) -- or 64*KHW of one channel (dgrad)
            if (!for_dgrad) {
                for (int e = t; e < 64 * RUN; e += 256) {
                    const int r = e / RUN, o = e - r * RUN;
                    tile[e] = (m0 + r < rows) ? j.w[((size_t)(m0 + r) * Cin + c0) * KHW + o] : 0.f;
                }
            } else {
                const int RUNd = 64 * KHW;
                for (int e = t; e < CS * RUNd; e += 256) {
                    const int c = e / RUNd, o = e - c * RUNd;            // o = r*KHW + tap
                    tile[e] = (m0 + o / KHW < rows) ? j.w[((size_t)(c0 + c) * Cin + m0) * KHW + o] : 0.f;
                }
            }
            __syncthreads();
            const long nsteps = (long)nst * KHW * KGS;
            const long per_tile = (nsteps + P9S_AHEAD) * (JP_NS * 8) * BMT;
            const int trip = KHW * KGS * 2 * 64;        // (tap, group, k-half, row) triples of this item
            for (int e = t; e < trip; e += 256) {
                const int r = e & 63, kh = (e >> 6) & 1, u = e >> 7;      // u = tap*KGS + kg
                const int tap = u / KGS, kg = u - tap * KGS;
                jp_u32x4 w0, w1, w2;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = kg * 16 + kh * 8 + 2 * k;
                    const float a = for_dgrad ? tile[(c * 64 + r) * KHW + tap] : tile[(r * CS + c) * KHW + tap];
                    const float b = for_dgrad ? tile[((c + 1) * 64 + r) * KHW + tap] : tile[(r * CS + c + 1) * KHW + tap];
                    unsigned sq[3];
                    jp_split_ns(a, b, wsc, sq);
                    w0[k] = sq[0]; w1[k] = sq[1]; w2[k] = sq[2];
                }
                const long U = (long)stage * KHW * KGS + u;
                unsigned* q = out + (long)mt * per_tile + ((U * JP_NS) * 2 + kh) * (long)BMT * 4 + (long)(rc * 64 + r) * 4;
                *reinterpret_cast<jp_u32x4*>(q) = w0;
                *reinterpret_cast<jp_u32x4*>(q + 2L * BMT * 4) = w1;
                if (JP_NS == 3) *reinterpret_cast<jp_u32x4*>(q + 4L * BMT * 4) = w2;
            }
        } else {            // PACK_SPLITSEG: p = Cout, Cin, c_off, C, up
            const int Cout = p[0], Cin = p[1], c_off = p[2], C = p[3], up = p[4];
            const int T = up ? 4 : 9, MT = Cout / 128, nst = (C + 15) / 16;
            const int rc = loc & 1, stage = (loc >> 1) % nst, mt = (loc >> 1) / nst;
            const int m0 = mt * 128 + rc * 64, c0 = stage * 16;
            for (int e = t; e < 64 * 144; e += 256) {
                const int r = e / 144, o = e - r * 144, c = o / 9;
                tile[e] = (c0 + c < C && m0 + r < Cout) ? j.w[((size_t)(m0 + r) * Cin + c_off + c0) * 9 + o] : 0.f;
            }
            __syncthreads();
            const long tl = (long)nst * T * (JP_NS * 1024);
            const int ncls = up ? 4 : 1, trip = ncls * T * 2 * 64;
            const float wsc = pack_scale_of(j);
            for (int e = t; e < trip; e += 256) {
                const int r = e & 63, kh = (e >> 6) & 1, ct = e >> 7, tap = ct % T, cls = ct / T;
                jp_u32x4 w0, w1, w2;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = kh * 8 + 2 * k;
                    const float* wa = tile + (r * 16 + c) * 9;
                    const float a = up ? pack_slot_sum(wa, cls * 4 + tap) : wa[tap];
                    const float b = up ? pack_slot_sum(wa + 9, cls * 4 + tap) : wa[9 + tap];
                    unsigned sq[3];
                    jp_split_ns(a, b, wsc, sq);
                    w0[k] = sq[0]; w1[k] = sq[1]; w2[k] = sq[2];
                }
                const long U = (long)stage * T + tap;
                unsigned* q = out + (long)(cls * MT + mt) * tl + ((U * JP_NS) * 2 + kh) * 512L + (long)(rc * 64 + r) * 4;
                *reinterpret_cast<jp_u32x4*>(q) = w0;
                *reinterpret_cast<jp_u32x4*>(q + 1024) = w1;
                if (JP_NS == 3) *reinterpret_cast<jp_u32x4*>(q + 2048) = w2;
            }
        }
    }
}

thread_local JpPackJob* g_pack_rec = nullptr;
thread_local int g_pack_rec_n = 0, g_pack_rec_cap = 0;

void do_pack(int mode, const float* w, float* wp, long total, int p0, int p1, int p2, int p3, int p4, int p5, hipStream_t st) {
    JpPackJob j{w, wp, total, 0, mode, {p0, p1, p2, p3, p4, p5}};
    if (g_pack_rec) {
        if (g_pack_rec_n < g_pack_rec_cap) g_pack_rec[g_pack_rec_n] = j;
        ++g_pack_rec_n;      // counted even when the buffer is full: jp_pack_record_end reports the overflow
    }
    if (JP_PACK_HDR && (mode == PACK_SPLIT || mode == PACK_SPLITUPD || mode == PACK_SPLIT7 || (mode == PACK_SPLITSEG && p2 == 0))) {
        hipLaunchKernelGGL(pack_scale_zero_one_kernel, dim3(1), dim3(64), 0, st, j);
        hipLaunchKernelGGL(pack_scale_reduce_one_kernel, dim3(PSL), dim3(256), 0, st, j);
        hipLaunchKernelGGL(pack_scale_finish_one_kernel, dim3(1), dim3(64), 0, st, j);
    }
    hipLaunchKernelGGL(pack_one_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, st, j);
}

void pack_weights(const float* w, float* wp, int Cout, int Cin, int KHW, int Cp, int for_dgrad, hipStream_t st) {
    do_pack(PACK_TAP, w, wp, (long)KHW * (for_dgrad ? Cin : Cout) * Cp, Cout, Cin, KHW, Cp, for_dgrad, 0, st);
}


struct Src3 {  // input as up to 3 channel segments, each optionally stored at half resolution
    const float *p0, *p1, *p2;
    int e0, e1, e2;     // cumulative channel ends
    int s0, s1, s2;     // 1 = segment stored at half resolution (fused nearest 2x upsample)
    int H, W;           // logical spatial size seen by the convolution
    __device__ __forceinline__ float at(int img, int ci, int iy, int ix) const {
        // scalar selects only: runtime-indexed member arrays would be spilled to scratch
        const bool a = ci < e0, b = ci < e1;
        const float* p = a ? p0 : (b ? p1 : p2);
        const int c0 = a ? 0 : (b ? e0 : e1);
        const int Cs = a ? e0 : (b ? e1 - e0 : e2 - e1);
        const int sh = a ? s0 : (b ? s1 : s2);
        const int h = H >> sh, w = W >> sh;
        return p[((size_t)(img * Cs + (ci - c0)) * h + (iy >> sh)) * w + (ix >> sh)];
    }
};

struct PixSt {  // decoded output pixel of a conv
    int img, iy0, ix0, valid;
};

__device__ __forceinline__ PixSt conv_pix(int p, int Npix, int OH, int OW, int stride, int pad) {
    PixSt st;
    st.valid = p < Npix;
    int ohw = OH * OW;
    st.img = p / ohw;
    int pix = p - st.img * ohw;
    int oy = pix / OW, ox = pix - oy * OW;
    st.iy0 = oy * stride - pad;
    st.ix0 = ox * stride - pad;
    return st;
}

// =============================================================== generic (channel-major K) loaders
struct FwdA {  // A[m=co][k] = W[co*K + k]
    static constexpr bool ALONG_K = true;
    typedef int St;
    const float* w;
    int M, K;
    __device__ __forceinline__ void init(St&, int, int) const {}
    __device__ __forceinline__ void fix(St& st, int k) const { st = k; }
    __device__ __forceinline__ float get(St k, int m, int) const { return (m < M && k < K) ? w[(size_t)m * K + k] : 0.f; }
};

template <int KH>
__device__ __forceinline__ float conv_gather(const Src3& src, const PixSt& st, int k, int reflect) {
    int ci = k / (KH * KH);
    int tap = k - ci * (KH * KH);
    int dy = tap / KH, dx = tap - dy * KH;
    int iy = st.iy0 + dy, ix = st.ix0 + dx;
    if (reflect) {
        iy = jp_reflect(iy, src.H);
        ix = jp_reflect(ix, src.W);
    } else if ((unsigned)iy >= (unsigned)src.H || (unsigned)ix >= (unsigned)src.W) {
        return 0.f;
    }
    return src.at(st.img, ci, iy, ix);
}

struct PixKSt {
    PixSt px;
    int kc;
};

template <int KH>
struct FwdB {  // B[k=(ci,dy,dx)][n=pixel]
    static constexpr bool ALONG_K = false;
    typedef PixKSt St;
    Src3 src;
    int K, Npix, OH, OW, stride, pad, reflect;
    __device__ __forceinline__ void init(St& st, int p) const { st = St{conv_pix(p, Npix, OH, OW, stride, pad), 0}; }
    __device__ __forceinline__ void chunk(St& st, int kc) const { st.kc = kc; }
    __device__ __forceinline__ float get(const St& st, int kl, int) const {
        const int k = st.kc + kl;
        if (!st.px.valid || k >= K) return 0.f;
        return conv_gather<KH>(src, st.px, k, reflect);
    }
};

struct FwdEpi {  // y[img][co][pix] = act(acc + bias[co])
    typedef size_t St;
    float* y;
    const float* bias;
    int Cout, OHW, act;
    unsigned* amax = nullptr;           // != nullptr: the patch kernels report max |y| here (the entry point's amax_y)
    float* stats = nullptr;             // != nullptr: BatchNorm statistics partials of y (the entry point's bn_stats), igemm_p9s.h epilogue
    __device__ __forceinline__ St col(int p) const {
        int img = p / OHW;
        return (size_t)img * Cout * OHW + (p - img * OHW);
    }
    __device__ __forceinline__ void put(St base, int m, float v) const {
        if (bias) v += bias[m];
        y[base + (size_t)m * OHW] = jp_act(v, act);
    }
    __device__ __forceinline__ float put_get(St base, int m, float v) const {       // put, returning the stored value
        if (bias) v += bias[m];
        const float r = jp_act(v, act);
        y[base + (size_t)m * OHW] = r;
        return r;
    }
    __device__ __forceinline__ float put4_get(St base, int m, float4 v) const {     // put4, returning the largest stored magnitude
        const float b = bias ? bias[m] : 0.f;
        const float4 r = make_float4(jp_act(v.x + b, act), jp_act(v.y + b, act), jp_act(v.z + b, act), jp_act(v.w + b, act));
        *reinterpret_cast<float4*>(y + base + (size_t)m * OHW) = r;
        return fmaxf(fmaxf(jp_fmag(r.x), jp_fmag(r.y)), fmaxf(jp_fmag(r.z), jp_fmag(r.w)));
    }
    // four consecutive pixels of channel m (16-byte aligned: the patch kernels' tiles start at multiples of 32 pixels)
    __device__ __forceinline__ void put4(St base, int m, float4 v) const {
        const float b = bias ? bias[m] : 0.f;
        *reinterpret_cast<float4*>(y + base + (size_t)m * OHW) =
            make_float4(jp_act(v.x + b, act), jp_act(v.y + b, act), jp_act(v.z + b, act), jp_act(v.w + b, act));
    }
};

struct AtomicEpi {  // out[img][m][pix] += acc : split-K forward / dgrad on grids too small to fill the chip
    typedef size_t St;
    float* out;
    int C, HW;
    __device__ __forceinline__ St col(int p) const {
        int img = p / HW;
        return (size_t)img * C * HW + (p - img * HW);
    }
    __device__ __forceinline__ void put(St base, int m, float v) const { atomicAdd(out + base + (size_t)m * HW, v); }
};

__global__ void bias_act_nchw_kernel(float* __restrict__ y, const float* __restrict__ bias, long total, int C, int HW,
                                     int act) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)((i / HW) % C);
        y[i] = jp_act(y[i] + (bias ? bias[c] : 0.f), act);
    }
}

// deterministic small-grid split-K: y[img][m][pix] = act(bias[m] + sum_s part[s][m][n]) in a fixed order
__global__ void slice_reduce_nchw_kernel(const float* __restrict__ part, float* __restrict__ y,
                                         const float* __restrict__ bias, int M, int Npix, int HW, int splits, int act,
                                         int accumulate) {
    const long total = (long)M * Npix;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / Npix), n = (int)(i - (long)m * Npix);
        float s = bias ? bias[m] : 0.f;
        for (int k = 0; k < splits; ++k) s += part[(size_t)k * total + i];
        const int img = n / HW;
        float* q = y + ((size_t)img * M + m) * HW + (n - img * HW);
        *q = accumulate ? *q + jp_act(s, act) : jp_act(s, act);
    }
}

struct MKSt {
    int m, kc;
};

template <int KH>
struct DgradA {  // A[m=ci][k=(co,dy,dx)] = W[co][ci][dy][dx]
    static constexpr bool ALONG_K = false;
    typedef MKSt St;
    const float* w;
    int Cin, K;
    __device__ __forceinline__ void init(St& st, int m) const { st = St{m, 0}; }
    __device__ __forceinline__ void chunk(St& st, int kc) const { st.kc = kc; }
    __device__ __forceinline__ float get(const St& st, int kl, int) const {
        const int k = st.kc + kl;
        if (st.m >= Cin || k >= K) return 0.f;
        int co = k / (KH * KH);
        int tap = k - co * (KH * KH);
        return w[((size_t)co * Cin + st.m) * (KH * KH) + tap];
    }
};

struct InPixSt {
    int img, y, x, valid;
};

// offsets (within one dY channel plane) of the dY entries that touched input pixel (y,x) through tap (ty,tx):
// zero padding: at most one; ReflectionPad2d(1) + 3x3 stride 1: padded row py in [0,H+1] maps to input row
// reflect(py-1); input row y is hit by py=y+1 and additionally by py=0 when y==1 and py=H+1 when y==H-2;
// the dY row is py - ty and must lie in [0,H) -> at most 2 rows x 2 cols.
struct DyOffs {
    int o0, o1, o2, o3;
};
__device__ __forceinline__ DyOffs dy_offsets(int y, int x, int ty, int tx, int H, int W, int OH, int OW, int stride,
                                             int pad, int reflect) {
    DyOffs d{-1, -1, -1, -1};
    if (!reflect) {
        int ny = y + pad - ty, nx = x + pad - tx;
        if (ny < 0 || nx < 0) return d;
        int oy = ny / stride, ox = nx / stride;
        if (oy * stride != ny || ox * stride != nx || oy >= OH || ox >= OW) return d;
        d.o0 = oy * OW + ox;
        return d;
    }
    int r0 = y + 1 - ty, r1 = -1, c0 = x + 1 - tx, c1 = -1;
    if (r0 < 0 || r0 >= H) r0 = -1;
    if (y == 1 && ty == 0) r1 = 0;
    if (y == H - 2 && ty == 2) r1 = H - 1;
    if (c0 < 0 || c0 >= W) c0 = -1;
    if (x == 1 && tx == 0) c1 = 0;
    if (x == W - 2 && tx == 2) c1 = W - 1;
    if (r0 >= 0 && c0 >= 0) d.o0 = r0 * OW + c0;
    if (r0 >= 0 && c1 >= 0) d.o1 = r0 * OW + c1;
    if (r1 >= 0 && c0 >= 0) d.o2 = r1 * OW + c0;
    if (r1 >= 0 && c1 >= 0) d.o3 = r1 * OW + c1;
    return d;
}
__device__ __forceinline__ float dy_sum(const float* q, const DyOffs& d) {
    float s = 0.f;
    if (d.o0 >= 0) s += q[d.o0];
    if (d.o1 >= 0) s += q[d.o1];
    if (d.o2 >= 0) s += q[d.o2];
    if (d.o3 >= 0) s += q[d.o3];
    return s;
}

struct InPixKSt {
    InPixSt px;
    int kc;
};

__device__ __forceinline__ InPixSt in_pix(int p, int Npix, int H, int W) {
    InPixSt px;
    px.valid = p < Npix;
    int hw = H * W;
    px.img = p / hw;
    int pix = p - px.img * hw;
    px.y = pix / W;
    px.x = pix - px.y * W;
    return px;
}

template <int KH>
struct DgradB {  // B[k=(co,dy,dx)][n=input pixel]
    static constexpr bool ALONG_K = false;
    typedef InPixKSt St;
    const float* dy;
    int K, Npix, H, W, Cout, OH, OW, stride, pad, reflect;
    __device__ __forceinline__ void init(St& st, int p) const { st = St{in_pix(p, Npix, H, W), 0}; }
    __device__ __forceinline__ void chunk(St& st, int kc) const { st.kc = kc; }
    __device__ __forceinline__ float get(const St& st, int kl, int) const {
        const int k = st.kc + kl;
        if (!st.px.valid || k >= K) return 0.f;
        int co = k / (KH * KH);
        int tap = k - co * (KH * KH);
        int ty = tap / KH, tx = tap - ty * KH;
        const float* d = dy + (size_t)(st.px.img * Cout + co) * OH * OW;
        return dy_sum(d, dy_offsets(st.px.y, st.px.x, ty, tx, H, W, OH, OW, stride, pad, reflect));
    }
};

struct DgradEpi {  // dx[img][ci][pix] (= or +=) acc
    typedef size_t St;
    float* dx;
    int Cin, HW, accumulate;
    unsigned* amax = nullptr;           // != nullptr: the patch kernels report max |dx as stored| here (the entry point's amax_dx)
    __device__ __forceinline__ St col(int p) const {
        int img = p / HW;
        return (size_t)img * Cin * HW + (p - img * HW);
    }
    __device__ __forceinline__ void put(St base, int m, float v) const {
        float* q = dx + base + (size_t)m * HW;
        *q = accumulate ? (*q + v) : v;
    }
    __device__ __forceinline__ void put4(St base, int m, float4 v) const {
        float4* q = reinterpret_cast<float4*>(dx + base + (size_t)m * HW);
        if (accumulate) { const float4 o = *q; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *q = v;
    }
    __device__ __forceinline__ float put_get(St base, int m, float v) const {       // put, returning the stored value
        float* q = dx + base + (size_t)m * HW;
        const float r = accumulate ? (*q + v) : v;
        *q = r;
        return r;
    }
    __device__ __forceinline__ float put4_get(St base, int m, float4 v) const {     // put4, returning the largest stored magnitude
        float4* q = reinterpret_cast<float4*>(dx + base + (size_t)m * HW);
        if (accumulate) { const float4 o = *q; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *q = v;
        return fmaxf(fmaxf(jp_fmag(v.x), jp_fmag(v.y)), fmaxf(jp_fmag(v.z), jp_fmag(v.w)));
    }
};

struct WgradASt {
    size_t base;
    int valid;
};
struct WgradA {  // A[m=co][k=pixel] = dY[img][co][pix]
    static constexpr bool ALONG_K = true;
    typedef WgradASt St;
    const float* dy;
    int Cout, Npix, OHW;
    __device__ __forceinline__ void init(St&, int, int) const {}
    __device__ __forceinline__ void fix(St& st, int p) const {
        st.valid = p < Npix;
        int img = p / OHW;
        st.base = (size_t)img * Cout * OHW + (p - img * OHW);
    }
    __device__ __forceinline__ float get(const St& st, int m, int) const {
        return (st.valid && m < Cout) ? dy[st.base + (size_t)m * OHW] : 0.f;
    }
};

template <int KH>
struct WgradB {  // B[k=pixel][n=(ci,dy,dx)] = xpad[img][ci][oy*s+dy][ox*s+dx]
    static constexpr bool ALONG_K = true;
    typedef PixSt St;
    Src3 src;
    int Kw, Npix, OH, OW, stride, pad, reflect;
    __device__ __forceinline__ void init(St&, int, int) const {}
    __device__ __forceinline__ void fix(St& st, int p) const { st = conv_pix(p, Npix, OH, OW, stride, pad); }
    __device__ __forceinline__ float get(const St& st, int j, int) const {
        if (!st.valid || j >= Kw) return 0.f;
        return conv_gather<KH>(src, st, j, reflect);
    }
};

struct WgradEpi {  // dw[co][j] += acc   (split-K partials meet in L2 atomics)
    typedef int St;
    float* dw;
    int Kw;
    __device__ __forceinline__ St col(int n) const { return n; }
    __device__ __forceinline__ void put(St j, int m, float v) const { atomicAdd(dw + (size_t)m * Kw + j, v); }
};

// =============================================================== tap-major ("T") fast-path loaders
// packed weights wp[tap][row][Cp] (row = co for forward, ci for dgrad; Cp = reduction channels padded to 32)
struct PackASt {
    const float* base;   // wave-uniform: (tap, channel chunk) corner of the packed weights
    unsigned voff;       // per-lane byte offset: (row within the 8-row group, channel within the chunk)
};
struct PackA {  // A[m=row][k=(tap,c)] = wp[(tap*M + m)*Cp + c]
    static constexpr bool ALONG_K = true;
    static constexpr bool SPLIT = true;
    typedef PackASt St;
    const float* wp;
    int M, Kp, Cp, khw;
    __device__ __forceinline__ void init(St& st, int, int) const {
        st.base = wp;
        st.voff = ((threadIdx.x >> 5) * Cp + (threadIdx.x & 31)) * 4u;
    }
    __device__ __forceinline__ void fix(St& st, int k) const {
        // K order = (channel chunk of 32, tap, channel in chunk): the 9 taps of one channel chunk are consecutive
        // K-chunks, so the gathered input rows are re-read while still hot in L1/L2 instead of from the fabric
        const int q = __builtin_amdgcn_readfirstlane(k >> 5);
        const int cc = q / khw, tap = q - cc * khw;
        st.base = wp + (size_t)tap * M * Cp + cc * 32;
    }
    // rows >= M are not masked: they read the next tap's rows (or the scratch slack after the last tap, see
    // jp_conv2d_ws_floats) and the epilogue never stores them.  Kp is a multiple of KC.
    __device__ __forceinline__ float get_u(const St& st, int m_u, int) const {
        const float* rp = st.base + (size_t)m_u * Cp;
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rp) + st.voff);
    }
};

struct FwdBTSt {
    int img_rel, iy0, ix0;   // per-lane: image relative to the block's first image, top-left tap coordinate
    int img0;                // uniform: image of the block's first pixel
    const float* rowp;       // uniform: plane (segment, img0, first channel of the chunk)
    unsigned voff;           // per-lane byte offset of (relative image, tap position) from rowp
    int hw, nm1;             // uniform: plane size of the segment; valid channels in this chunk - 1
    int ok;                  // per-lane, zero padding only: the tap lies inside the image
};

template <int KH, bool REFLECT>
struct FwdBT {  // B[k=(tap,ci)][n=pixel]
    static constexpr bool ALONG_K = false;
    static constexpr bool POST = true;
    typedef FwdBTSt St;
    Src3 src;
    int Cp, Npix, OH, OW, stride, pad;
    __device__ __forceinline__ void init(St& st, int p, int p0) const {
        const int ohw = OH * OW;
        p = min(p, Npix - 1);          // columns >= N are computed on a duplicate pixel and never stored
        const int img = p / ohw, pix = p - img * ohw;
        const int oy = pix / OW, ox = pix - oy * OW;
        st.img0 = min(p0, Npix - 1) / ohw;
        st.img_rel = img - st.img0;
        st.iy0 = oy * stride - pad;
        st.ix0 = ox * stride - pad;
        st.rowp = src.p0;
        st.voff = 0;
        st.hw = 0;
        st.nm1 = 0;
        st.ok = 1;
    }
    __device__ __forceinline__ void chunk(St& st, int kc) const {
        const int q = kc >> 5;                 // chunk index = channel_chunk * taps + tap
        const int cc = q / (KH * KH), tap = q - cc * (KH * KH);
        const int ci0 = cc << 5;
        const int dy = tap / KH, dx = tap - dy * KH;
        int iy = st.iy0 + dy, ix = st.ix0 + dx;
        if (REFLECT) {
            iy = jp_reflect(iy, src.H);
            ix = jp_reflect(ix, src.W);
        } else {
            st.ok = (unsigned)iy < (unsigned)src.H && (unsigned)ix < (unsigned)src.W;
            iy = min(max(iy, 0), src.H - 1);
            ix = min(max(ix, 0), src.W - 1);
        }
        const bool a = ci0 < src.e0, b = ci0 < src.e1;   // segment ends are multiples of 32 (host-checked)
        const float* p = a ? src.p0 : (b ? src.p1 : src.p2);
        const int c0 = a ? 0 : (b ? src.e0 : src.e1);
        const int ce = a ? src.e0 : (b ? src.e1 : src.e2);
        const int sh = a ? src.s0 : (b ? src.s1 : src.s2);
        const int h = src.H >> sh, w = src.W >> sh, Cs = ce - c0;
        st.hw = h * w;
        st.voff = (unsigned)((st.img_rel * Cs * h + (iy >> sh)) * w + (ix >> sh)) * 4u;
        st.rowp = p + (size_t)(st.img0 * Cs + (ci0 - c0)) * st.hw;
        st.nm1 = min(32, ce - ci0) - 1;       // rows beyond are clamped: their packed weights are zero
    }
    __device__ __forceinline__ float get(const St& st, int kl, int) const {   // kl is wave-uniform
        const float* rp = st.rowp + (size_t)min(kl, st.nm1) * st.hw;
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rp) + st.voff);
    }
    static constexpr bool ALL_OK = true;
    __device__ __forceinline__ bool all_ok(const St& st) const { return REFLECT || __all(st.ok); }
    __device__ __forceinline__ float post(const St& st, float v, int) const { return (REFLECT || st.ok) ? v : 0.f; }
};

struct DgradBTSt {
    int img_rel, y, x, img0;
    const float* rowp;   // uniform: dY plane (img0, first output channel of the chunk)
    unsigned voff;       // per-lane byte offset of (relative image, tap position)
    int nm1, ok;
};

// main dgrad gather: the single dY entry reached through tap (ty,tx) by the *direct* (non-folded) path.
// With reflection padding the extra folded-in entries of the 4 border-adjacent lines are added by a second,
// tiny launch (DgradBorderB below), which keeps this hot loop as lean as the forward gather.
template <int KH, bool S1 = false>   // S1: stride 1 (no per-lane integer divisions in the chunk prologue)
struct DgradBT {  // B[k=(tap,co)][n=input pixel]
    static constexpr bool ALONG_K = false;
    static constexpr bool POST = true;
    typedef DgradBTSt St;
    const float* dy;
    int Cp, Npix, H, W, Cout, OH, OW, stride, pad, reflect;
    __device__ __forceinline__ void init(St& st, int p, int p0) const {
        const int hw = H * W;
        p = min(p, Npix - 1);
        const int img = p / hw, pix = p - img * hw;
        st.y = pix / W;
        st.x = pix - st.y * W;
        st.img0 = min(p0, Npix - 1) / hw;
        st.img_rel = img - st.img0;
        st.rowp = dy;
        st.voff = 0;
        st.nm1 = 0;
        st.ok = 0;
    }
    __device__ __forceinline__ void chunk(St& st, int kc) const {
        const int q = kc >> 5;
        const int cc = q / (KH * KH), tap = q - cc * (KH * KH);
        const int co0 = cc << 5;
        const int ty = tap / KH, tx = tap - ty * KH;
        int oy, ox, ok;
        if (reflect) {   // 3x3 stride 1 pad 1: direct path = padded row y+1, dY row y+1-ty
            oy = st.y + 1 - ty;
            ox = st.x + 1 - tx;
            ok = 1;
        } else {
            const int ny = st.y + pad - ty, nx = st.x + pad - tx;
            if (S1) {
                oy = ny;
                ox = nx;
                ok = 1;        // range-checked below
            } else {
                oy = ny / stride;
                ox = nx / stride;
                ok = ny >= 0 && nx >= 0 && oy * stride == ny && ox * stride == nx;
            }
        }
        st.ok = ok && (unsigned)oy < (unsigned)OH && (unsigned)ox < (unsigned)OW;
        oy = min(max(oy, 0), OH - 1);
        ox = min(max(ox, 0), OW - 1);
        const int ohw = OH * OW;
        st.voff = (unsigned)(st.img_rel * Cout * ohw + oy * OW + ox) * 4u;
        st.rowp = dy + (size_t)(st.img0 * Cout + co0) * ohw;
        st.nm1 = min(32, Cout - co0) - 1;
    }
    __device__ __forceinline__ float get(const St& st, int kl, int) const {   // kl is wave-uniform
        const float* rp = st.rowp + (size_t)min(kl, st.nm1) * (OH * OW);
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rp) + st.voff);
    }
    static constexpr bool ALL_OK = true;
    __device__ __forceinline__ bool all_ok(const St& st) const { return __all(st.ok); }
    __device__ __forceinline__ float post(const St& st, float v, int) const { return st.ok ? v : 0.f; }
};

// ---- B loader of the row-tile kernel (igemm.h jp_igemm_r3_kernel): 3x3 stride 1 pad 1, the pixel tile is BN consecutive
// pixels of ONE image row.  REV = dgrad (row y+1-ty of dY, taps mirrored, zero fill -- the reflection fold is the border
// pass's job); otherwise forward with reflection or zero padding.  Sources must be stored at full resolution.
struct FwdBR3St {
    int img, y, x0;          // uniform: the tile's image, row, first column
    const float* rowp;       // uniform: (segment, image, first channel of the chunk, source row, x0)
    int hw, nm1, rowok;      // uniform
};
template <bool REFLECT, bool REV, int BN>
struct FwdBR3 {
    static constexpr bool REVERSE = REV;
    typedef FwdBR3St St;
    Src3 src;
    int H, W;
    __device__ __forceinline__ void init(St& st, int n0) const {
        const int hw = H * W;
        st.img = n0 / hw;
        const int rem = n0 - st.img * hw;
        st.y = rem / W;
        st.x0 = rem - st.y * W;
        st.rowp = src.p0;
        st.hw = hw;
        st.nm1 = 0;
        st.rowok = 0;
    }
    __device__ __forceinline__ void row(St& st, int kc) const {
        const int q = kc >> 5;
        const int cc = q / 9, tap = q - cc * 9;
        const int dy = tap / 3;
        int iy = REV ? st.y + 1 - dy : st.y - 1 + dy;
        if (REFLECT) {
            iy = jp_reflect(iy, H);
            st.rowok = 1;
        } else {
            st.rowok = (unsigned)iy < (unsigned)H;
            iy = min(max(iy, 0), H - 1);
        }
        const int ci0 = cc << 5;
        const bool a = ci0 < src.e0, b = ci0 < src.e1;
        const float* p = a ? src.p0 : (b ? src.p1 : src.p2);
        const int c0 = a ? 0 : (b ? src.e0 : src.e1);
        const int ce = a ? src.e0 : (b ? src.e1 : src.e2);
        st.rowp = p + ((size_t)(st.img * (ce - c0) + (ci0 - c0)) * H + iy) * W + st.x0;
        st.nm1 = min(32, ce - ci0) - 1;
    }
    __device__ __forceinline__ float get(const St& st, int kl, int n_l) const {    // kl wave-uniform, n_l = lane's column
        const float* rp = st.rowp + (size_t)min(kl, st.nm1) * st.hw;
        const float v = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rp) + (unsigned)n_l * 4u);
        return st.rowok ? v : 0.f;
    }
    __device__ __forceinline__ float halo(const St& st, int kl, int side) const {   // column x0-1 / x0+BN
        int x = side ? st.x0 + BN : st.x0 - 1;
        bool ok = st.rowok;
        if (REFLECT) x = jp_reflect(x, W);
        else { ok = ok && (unsigned)x < (unsigned)W; x = min(max(x, 0), W - 1); }
        const float v = st.rowp[(size_t)min(kl, st.nm1) * st.hw + (x - st.x0)];
        return ok ? v : 0.f;
    }
};

// ---- Few input channels (the 7x7 stride-2 ResNet stems: 3 image channels, 6 for the pose pair): K = (tap, c) with the
// channels padded to CP = 4 or 8, so a K chunk of 32 holds 32/CP whole taps; one per-lane offset per tap per chunk,
// every element a scalar-base load (the generic path decodes (c, dy, dx) per element).  64x256 tiles only (Cout <= 64).
struct PackARowSt {
    const float* base;
    unsigned voff;
};
struct PackARow {  // A[m][k] = wp[m*Kp + k]  (row-major, K zero-padded to Kp)
    static constexpr bool ALONG_K = true;
    static constexpr bool SPLIT = true;
    typedef PackARowSt St;
    const float* wp;
    int Kp;
    __device__ __forceinline__ void init(St& st, int, int) const {
        st.base = wp;
        st.voff = ((threadIdx.x >> 5) * Kp + (threadIdx.x & 31)) * 4u;
    }
    __device__ __forceinline__ void fix(St& st, int k) const { st.base = wp + __builtin_amdgcn_readfirstlane(k & ~31); }
    __device__ __forceinline__ float get_u(const St& st, int m_u, int) const {
        const float* rp = st.base + (size_t)m_u * Kp;
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rp) + st.voff);
    }
};
template <int TPC>
struct FwdBCSt {
    int img_rel, iy0, ix0, img0;
    unsigned voff[TPC];
    unsigned ok;
};
template <int KH, int CP>
struct FwdBC {  // B[k=(tap, c)][n=pixel], zero padding, single full-resolution source
    static constexpr bool ALONG_K = false;
    static constexpr bool POST = true;
    static constexpr int TPC = 32 / CP;
    typedef FwdBCSt<TPC> St;
    const float* x;
    int Cin, H, W, Npix, OH, OW, stride, pad;
    __device__ __forceinline__ void init(St& st, int p, int p0) const {
        const int ohw = OH * OW;
        p = min(p, Npix - 1);
        const int img = p / ohw, pix = p - img * ohw;
        const int oy = pix / OW, ox = pix - oy * OW;
        st.img0 = min(p0, Npix - 1) / ohw;
        st.img_rel = img - st.img0;
        st.iy0 = oy * stride - pad;
        st.ix0 = ox * stride - pad;
        st.ok = 0;
#pragma unroll
        for (int s = 0; s < TPC; ++s) st.voff[s] = 0;
    }
    __device__ __forceinline__ void chunk(St& st, int kc) const {
        const int tap0 = (kc >> 5) * TPC;
        unsigned ok = 0;
#pragma unroll
        for (int s = 0; s < TPC; ++s) {
            const int tap = min(tap0 + s, KH * KH - 1);     // taps past the filter: zero weights
            const int dy = tap / KH, dx = tap - dy * KH;
            int iy = st.iy0 + dy, ix = st.ix0 + dx;
            ok |= (unsigned)((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) << s;
            iy = min(max(iy, 0), H - 1);
            ix = min(max(ix, 0), W - 1);
            st.voff[s] = (unsigned)((st.img_rel * Cin * H + iy) * W + ix) * 4u;
        }
        st.ok = ok;
    }
    // 64x256 tile: the k row of slot r is r itself -> tap slot r/CP, channel r%CP at compile time
    __device__ __forceinline__ float get(const St& st, int, int r) const {
        const float* rp = x + ((size_t)st.img0 * Cin + min(r % CP, Cin - 1)) * H * W;
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rp) + st.voff[r / CP]);
    }
    static constexpr bool ALL_OK = true;
    __device__ __forceinline__ bool all_ok(const St& st) const { return __all(st.ok == ((1u << TPC) - 1u)); }
    __device__ __forceinline__ float post(const St& st, float v, int r) const { return ((st.ok >> (r / CP)) & 1u) ? v : 0.f; }
};

// ---- Upsample-aware 3x3 reflect conv (iconv layers: cat(reduce, up2x(x), disp), depth_decoder.py:68,76-77).
// For the nearest-2x upsampled segment U = up(X) an output pixel (2i+a, 2j+b) sees only a 2x2 patch of X through its
// 9 taps: rows {i-1+a, i+a} x cols {j-1+b, j+b} (reflection padding of U == edge clamp on X).  Output pixels are
// enumerated parity-class-major n = (class (a,b), img, i, j) so a tile is class-uniform; the K loop then runs 9 taps
// over the full-resolution segments but only 4 slots (r, s) over the upsampled one, against weights pre-summed per
// class:  W'_{ab}[r][s] = sum_{dy in Dy(a,r)} sum_{dx in Dx(b,s)} W[dy][dx],  Dy(0,.) = {0},{1,2}; Dy(1,.) = {0,1},{2}.
// 27.7 % fewer MFMA FLOPs on the 513-channel layers.
struct ParSeg {       // K-chunk layout of the three channel segments
    int q0, q1;       // cumulative chunk counts after segment 0 / 1
    int t0, t1, t2;   // taps (9) or slots (4) per channel chunk of each segment
    int o0, o1, o2;   // float offset of each segment's packed weights in ws
    int p0, p1, p2;   // padded channel count (row length) of each segment
};
struct PackAPSt {
    const float* base;
    unsigned rg, cl, Cp;   // row group (t>>5), channel in chunk (t&31), row length of the current segment
    int cls;
};
struct PackAP {   // A[m=co][k] for the segment-piecewise K order; up segments hold [class][slot][co][Cp]
    static constexpr bool ALONG_K = true;
    static constexpr bool SPLIT = true;
    static constexpr bool WANTS_TILE = true;
    typedef PackAPSt St;
    const float* wp;
    ParSeg ps;
    int M, Nc;
    __device__ __forceinline__ void init(St& st, int, int, int, int n0) const {
        st.base = wp;
        st.rg = threadIdx.x >> 5;
        st.cl = threadIdx.x & 31;
        st.Cp = 32;
        st.cls = n0 / Nc;
    }
    __device__ __forceinline__ void fix(St& st, int k) const {
        const int q = __builtin_amdgcn_readfirstlane(k >> 5);
        const bool a = q < ps.q0, b = q < ps.q1;
        const int ql = a ? q : (b ? q - ps.q0 : q - ps.q1);
        const int T = a ? ps.t0 : (b ? ps.t1 : ps.t2);
        const int off = a ? ps.o0 : (b ? ps.o1 : ps.o2);
        const int Cp = a ? ps.p0 : (b ? ps.p1 : ps.p2);
        const int cc = ql / T, t = ql - cc * T;
        const int plane = T == 4 ? st.cls * 4 + t : t;      // [class][slot] or [tap]
        st.base = wp + off + (size_t)plane * M * Cp + cc * 32;
        st.Cp = Cp;
    }
    __device__ __forceinline__ float get_u(const St& st, int m_u, int) const {
        const float* rp = st.base + (size_t)m_u * st.Cp;
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rp) + (st.rg * st.Cp + st.cl) * 4u);
    }
};

struct FwdBPSt {
    int img_rel, i, j, img0, a, b;
    const float* rowp;
    unsigned voff;
    int hw, nm1;
};
struct FwdBP {  // B[k][n=(class, img, i, j)], 3x3 stride 1 reflection pad
    static constexpr bool ALONG_K = false;
    static constexpr bool POST = true;
    typedef FwdBPSt St;
    Src3 src;
    ParSeg ps;
    int Nc, h2, w2;    // pixels per class, half-resolution size
    __device__ __forceinline__ void init(St& st, int p, int p0) const {
        const int cls = p0 / Nc;
        st.a = cls >> 1;
        st.b = cls & 1;
        const int hw2 = h2 * w2;
        const int q = min(p - cls * Nc, Nc - 1), q0 = p0 - cls * Nc;
        const int img = q / hw2, pix = q - img * hw2;
        st.i = pix / w2;
        st.j = pix - st.i * w2;
        st.img0 = q0 / hw2;
        st.img_rel = img - st.img0;
        st.rowp = src.p0;
        st.voff = 0;
        st.hw = 0;
        st.nm1 = 0;
    }
    __device__ __forceinline__ void chunk(St& st, int kc) const {
        const int q = kc >> 5;
        const bool sa = q < ps.q0, sb = q < ps.q1;
        const int ql = sa ? q : (sb ? q - ps.q0 : q - ps.q1);
        const int T = sa ? ps.t0 : (sb ? ps.t1 : ps.t2);
        const float* p = sa ? src.p0 : (sb ? src.p1 : src.p2);
        const int Cs = sa ? src.e0 : (sb ? src.e1 - src.e0 : src.e2 - src.e1);
        const int cc = ql / T, t = ql - cc * T;
        int yy, xx, h, w;
        if (T == 4) {      // upsampled segment: 2x2 patch of the half-resolution tensor, edge-clamped
            h = h2; w = w2;
            yy = min(max(st.i - 1 + st.a + (t >> 1), 0), h2 - 1);
            xx = min(max(st.j - 1 + st.b + (t & 1), 0), w2 - 1);
        } else {           // full-resolution segment: the usual 9 reflected taps
            h = 2 * h2; w = 2 * w2;
            const int dy = t / 3, dx = t - dy * 3;
            yy = jp_reflect(2 * st.i + st.a - 1 + dy, h);
            xx = jp_reflect(2 * st.j + st.b - 1 + dx, w);
        }
        st.hw = h * w;
        st.voff = (unsigned)((st.img_rel * Cs * h + yy) * w + xx) * 4u;
        st.rowp = p + (size_t)(st.img0 * Cs + cc * 32) * st.hw;
        st.nm1 = min(32, Cs - cc * 32) - 1;
    }
    __device__ __forceinline__ float get(const St& st, int kl, int) const {
        const float* rp = st.rowp + (size_t)min(kl, st.nm1) * st.hw;
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rp) + st.voff);
    }
    static constexpr bool ALL_OK = true;
    __device__ __forceinline__ bool all_ok(const St&) const { return true; }
    __device__ __forceinline__ float post(const St&, float v, int) const { return v; }
};
struct FwdEpiP {  // y[img][co][2i+a][2j+b] = act(acc + bias[co])
    typedef size_t St;
    float* y;
    const float* bias;
    int Cout, Nc, h2, w2, act;
    __device__ __forceinline__ St col(int n) const {
        const int cls = n / Nc, q = n - cls * Nc;
        const int hw2 = h2 * w2;
        const int img = q / hw2, pix = q - img * hw2;
        const int i = pix / w2, j = pix - i * w2;
        return (size_t)img * Cout * (4 * hw2) + (size_t)(2 * i + (cls >> 1)) * (2 * w2) + 2 * j + (cls & 1);
    }
    __device__ __forceinline__ void put(St base, int m, float v) const {
        if (bias) v += bias[m];
        y[base + (size_t)m * (4 * h2 * w2)] = jp_act(v, act);
    }
};


// ---- wgrad of the upsampled segment in the same parity-class form: dW'_{class}[slot][co][c] = sum over the class's
// output pixels of dY[co][2i+a][2j+b] * X[c][clamp(i-1+a+r)][clamp(j-1+b+s)]  (GEMM M=co, N=(class, slot, c), K = class
// pixels; 4/9 of the plain wgrad's MFMA work), then dW[dy][dx] += the 4 (class, slot) pairs that contain the tap.
// Preconditions: W/2 % 32 == 0 (a K chunk is 32 consecutive j of one row), Cx % 128 == 0, Cout % 8 == 0.
struct WgradAPSt {
    const float* base;
    unsigned voff;
    int a, b;
};
struct WgradAP {  // A[m=co][k=(img,i,j) of the tile's class] = dY[img][co][2i+a][2j+b]
    static constexpr bool ALONG_K = true;
    static constexpr bool SPLIT = true;
    static constexpr bool WANTS_TILE = true;
    typedef WgradAPSt St;
    const float* dy;
    int Cout, h2, w2, ncls;    // ncls = columns per class = 4 * Cx
    __device__ __forceinline__ void init(St& st, int, int, int, int n0) const {
        const int cls = n0 / ncls;
        st.a = cls >> 1;
        st.b = cls & 1;
        st.base = dy;
        st.voff = ((threadIdx.x >> 5) * (4 * h2 * w2) + 2 * (threadIdx.x & 31)) * 4u;
    }
    __device__ __forceinline__ void fix(St& st, int p) const {
        const int p0 = __builtin_amdgcn_readfirstlane(p & ~31);
        const int hw2 = h2 * w2;
        const int img = p0 / hw2, rem = p0 - img * hw2;
        const int i = rem / w2, j0 = rem - i * w2;
        st.base = dy + (size_t)img * Cout * (4 * hw2) + (size_t)(2 * i + st.a) * (2 * w2) + 2 * j0 + st.b;
    }
    __device__ __forceinline__ float get_u(const St& st, int m_u, int) const {
        const float* rp = st.base + (size_t)min(m_u, Cout - 8) * (4 * h2 * w2);
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rp) + st.voff);
    }
};
struct WgradBPSt {
    const float* base;
    unsigned voff;
    int a, b, r, s, c0;
};
struct WgradBP {  // B[k=(img,i,j)][n=(class, slot, c)] = X[img][c][clamp(i-1+a+r)][clamp(j-1+b+s)]
    static constexpr bool ALONG_K = true;
    static constexpr bool SPLIT = true;
    static constexpr bool WANTS_TILE = true;
    typedef WgradBPSt St;
    const float* x;     // half-resolution source (N, Cx, h2, w2)
    int Cx, h2, w2;
    __device__ __forceinline__ void init(St& st, int, int, int, int n0) const {
        const int q = n0 / Cx;           // (class, slot)
        st.c0 = n0 - q * Cx;
        const int cls = q >> 2, slot = q & 3;
        st.a = cls >> 1; st.b = cls & 1; st.r = slot >> 1; st.s = slot & 1;
        st.base = x;
        st.voff = 0;
    }
    __device__ __forceinline__ void fix(St& st, int p) const {
        const int p0 = __builtin_amdgcn_readfirstlane(p & ~31);
        const int hw2 = h2 * w2;
        const int img = p0 / hw2, rem = p0 - img * hw2;
        const int i = rem / w2;
        const int j = p - img * hw2 - i * w2;       // per lane
        const int xi = min(max(i - 1 + st.a + st.r, 0), h2 - 1);
        const int xj = min(max(j - 1 + st.b + st.s, 0), w2 - 1);
        st.voff = (unsigned)((threadIdx.x >> 5) * hw2 + xj) * 4u;
        st.base = x + (size_t)(img * Cx + st.c0) * hw2 + (size_t)xi * w2;
    }
    __device__ __forceinline__ float get_u(const St& st, int, int r) const {
        const float* rp = st.base + (size_t)(8 * r) * (h2 * w2);
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rp) + st.voff);
    }
};
// dw[m][c_off + c][tap] += sum over splits and over the 4 (class, slot) pairs containing the tap of ws[k][m][n]
__global__ void wgrad_fold_parity_kernel(const float* __restrict__ ws, float* __restrict__ dw, int M, int Cx, int splits,
                                         int c_off, int Ctot) {
    const long total = (long)M * Cx * 9, plane = (long)M * 16 * Cx;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cx);
        const long t = i / Cx;
        const int tap = (int)(t % 9), m = (int)(t / 9);
        const int dy = tap / 3, dx = tap - dy * 3;
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int r = a ? (dy == 2) : (dy >= 1);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int sx = b ? (dx == 2) : (dx >= 1);
                const long n = (long)(((a * 2 + b) * 4 + r * 2 + sx)) * Cx + c;
                for (int k = 0; k < splits; ++k) s += ws[(size_t)k * plane + (size_t)m * 16 * Cx + n];
            }
        }
        dw[((size_t)m * Ctot + c_off + c) * 9 + tap] += s;
    }
}

// ---- dgrad of the upsampled segment, directly at half resolution:
//   dX[c][i][j] = sum_{class (a,b), slot (r,s), co} W'_{ab}[r][s][co][c] * dY[co][2i'+a][2j'+b],
//   (i', j') = (i+1-a-r, j+1-b-s) where it exists (GEMM M=c, N=half-res pixels, K=(co chunk, 16 (class,slot), co); 4/9 of
//   the full-resolution dgrad + 2x2 sum it replaces).  The edge clamp of the forward adds, on the 4 boundary lines, the
//   entries (i'=0 via a=0,r=0 | i'=h-1 via a=1,r=1, same for columns): a split-K border pass like the reflection one.
struct DgradUPSt {
    int img_rel, i, j, img0;
    const float* rowp;
    unsigned voff;
    int nm1, ok;
};
struct DgradUPB {
    static constexpr bool ALONG_K = false;
    static constexpr bool POST = true;
    typedef DgradUPSt St;
    const float* dy;
    int Npix2, h2, w2, Cout;
    __device__ __forceinline__ void init(St& st, int p, int p0) const {
        const int hw2 = h2 * w2;
        p = min(p, Npix2 - 1);
        const int img = p / hw2, pix = p - img * hw2;
        st.i = pix / w2;
        st.j = pix - st.i * w2;
        st.img0 = min(p0, Npix2 - 1) / hw2;
        st.img_rel = img - st.img0;
        st.rowp = dy;
        st.voff = 0;
        st.nm1 = 0;
        st.ok = 0;
    }
    __device__ __forceinline__ void chunk(St& st, int kc) const {
        const int qk = kc >> 5;
        const int cc = qk >> 4, q = qk & 15;
        const int a = q >> 3, b = (q >> 2) & 1, r = (q >> 1) & 1, s = q & 1;
        const int co0 = cc << 5;
        int ip = st.i + 1 - a - r, jp = st.j + 1 - b - s;
        st.ok = (unsigned)ip < (unsigned)h2 && (unsigned)jp < (unsigned)w2;
        ip = min(max(ip, 0), h2 - 1);
        jp = min(max(jp, 0), w2 - 1);
        const int HW = 4 * h2 * w2;
        st.voff = (unsigned)(st.img_rel * Cout * HW + (2 * ip + a) * (2 * w2) + 2 * jp + b) * 4u;
        st.rowp = dy + (size_t)(st.img0 * Cout + co0) * HW;
        st.nm1 = min(32, Cout - co0) - 1;
    }
    __device__ __forceinline__ float get(const St& st, int kl, int) const {
        const float* rp = st.rowp + (size_t)min(kl, st.nm1) * (4 * h2 * w2);
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rp) + st.voff);
    }
    static constexpr bool ALL_OK = true;
    __device__ __forceinline__ bool all_ok(const St& st) const { return __all(st.ok); }
    __device__ __forceinline__ float post(const St& st, float v, int) const { return st.ok ? v : 0.f; }
};

// boundary pixels of the half-resolution map: rows 0 / h-1, columns 0 / w-1 (duplicates masked)
__device__ __forceinline__ InPixSt edge_pix(int b, int Nb, int h, int w) {
    InPixSt px;
    const int per = 2 * w + 2 * h;
    px.valid = b < Nb;
    px.img = b / per;
    const int i = b - px.img * per;
    int y, x;
    bool dup = false;
    if (i < w) { y = 0; x = i; }
    else if (i < 2 * w) { y = h - 1; x = i - w; dup = (h == 1); }
    else if (i < 2 * w + h) { y = i - 2 * w; x = 0; dup = (y == 0 || y == h - 1); }
    else { y = i - 2 * w - h; x = w - 1; dup = (y == 0 || y == h - 1) || (w == 1); }
    px.y = y; px.x = x;
    if (dup) px.valid = 0;
    return px;
}
struct DgradUPBorderSt {
    InPixSt px;
    const float* p;
    int o1, o2, o3, n;
};
struct DgradUPBorderB {   // only the clamp-folded entries: (extra row, any column) and (regular row, extra column)
    static constexpr bool ALONG_K = false;
    typedef DgradUPBorderSt St;
    const float* dy;
    int Nb, h2, w2, Cout;
    __device__ __forceinline__ void init(St& st, int b) const { st = St{edge_pix(b, Nb, h2, w2), nullptr, -1, -1, -1, 0}; }
    __device__ __forceinline__ void chunk(St& st, int kc) const {
        const int qk = kc >> 5;
        const int cc = qk >> 4, q = qk & 15;
        const int a = q >> 3, b = (q >> 2) & 1, r = (q >> 1) & 1, s = q & 1;
        const int co0 = cc << 5;
        st.n = 0;
        if (!st.px.valid || co0 >= Cout) return;
        const int i = st.px.y, j = st.px.x;
        const int er = (i == 0 && a == 0 && r == 0) ? 0 : ((i == h2 - 1 && a == 1 && r == 1) ? h2 - 1 : -1);
        const int ec = (j == 0 && b == 0 && s == 0) ? 0 : ((j == w2 - 1 && b == 1 && s == 1) ? w2 - 1 : -1);
        if (er < 0 && ec < 0) return;
        const int rr = i + 1 - a - r, rc = j + 1 - b - s;
        const bool rrv = (unsigned)rr < (unsigned)h2, rcv = (unsigned)rc < (unsigned)w2;
        const int W = 2 * w2;
        st.o1 = (er >= 0 && rcv) ? (2 * er + a) * W + 2 * rc + b : -1;
        st.o2 = (rrv && ec >= 0) ? (2 * rr + a) * W + 2 * ec + b : -1;
        st.o3 = (er >= 0 && ec >= 0) ? (2 * er + a) * W + 2 * ec + b : -1;
        st.p = dy + (size_t)(st.px.img * Cout + co0) * (4 * h2 * w2);
        st.n = min(32, Cout - co0);
    }
    __device__ __forceinline__ float get(const St& st, int kl, int) const {
        if (kl >= st.n) return 0.f;
        const float* q = st.p + (size_t)kl * (4 * h2 * w2);
        float v = 0.f;
        if (st.o1 >= 0) v += q[st.o1];
        if (st.o2 >= 0) v += q[st.o2];
        if (st.o3 >= 0) v += q[st.o3];
        return v;
    }
    // slots (a, b, r, s) that fold something in for some pixel of the tile (see chunk()); the rest are stepped over
    static constexpr bool SKIP = true;
    __device__ __forceinline__ unsigned tile_mask(const St& st) const {
        __shared__ unsigned tm;
        if (threadIdx.x == 0) tm = 0;
        __syncthreads();
        unsigned mine = 0;
        if (st.px.valid) {
            const int i = st.px.y, j = st.px.x;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int a = q >> 3, b = (q >> 2) & 1, r = (q >> 1) & 1, s2 = q & 1;
                const bool er = (i == 0 && a == 0 && r == 0) || (i == h2 - 1 && a == 1 && r == 1);
                const bool ec = (j == 0 && b == 0 && s2 == 0) || (j == w2 - 1 && b == 1 && s2 == 1);
                if (er || ec) mine |= 1u << q;
            }
        }
        if (mine) atomicOr(&tm, mine);
        __syncthreads();
        return (unsigned)__builtin_amdgcn_readfirstlane((int)tm);
    }
    __device__ __forceinline__ bool skip(unsigned m, int kc) const { return !((m >> ((kc >> 5) & 15)) & 1u); }
};
struct DgradEdgeEpi {  // dx[img][c][y][x] += acc for the boundary pixel b of a (h, w) map
    typedef long St;
    float* dx;
    int C, h, w, Nb, split;
    __device__ __forceinline__ St col(int b) const {
        const InPixSt px = edge_pix(b, Nb, h, w);
        return px.valid ? (long)px.img * C * h * w + px.y * w + px.x : -1;
    }
    __device__ __forceinline__ void put(St base, int m, float v) const {
        if (base < 0) return;
        float* q = dx + base + (size_t)m * h * w;
        if (split) atomicAdd(q, v);       // split-K partials meet here
        else *q += v;                     // one workgroup owns the pixel: plain read-modify-write
    }
};

// ---- 3x3 stride-2 pad-1 dgrad (ResNet downsampling convs), parity-class form.  An input pixel (y, x) is reached only
// through taps with ty = y+1 (mod 2), tx = x+1 (mod 2): 1, 2, 2 or 4 of the 9.  Input pixels are enumerated class-major
// n = (class (py,px), img, i, j) with y = 2i+py, x = 2j+px, so a pixel tile is class-uniform and the K loop runs over
// 4 tap slots (r, s) instead of 9 taps; slots a class does not have are masked.  (H, W even; class size % 256 == 0.)
struct PackAS2St {
    const float* base;
    unsigned voff;
    int py, px;
};
struct PackAS2 {   // A[m=ci][k=(cc, slot, c)] = wp[(tap(slot, class)*M + m)*Cp + c]
    static constexpr bool ALONG_K = true;
    static constexpr bool SPLIT = true;
    static constexpr bool WANTS_TILE = true;
    typedef PackAS2St St;
    const float* wp;
    int M, Cp, Nc;   // Nc = pixels per class
    __device__ __forceinline__ void init(St& st, int, int, int, int n0) const {
        st.base = wp;
        st.voff = ((threadIdx.x >> 5) * Cp + (threadIdx.x & 31)) * 4u;
        const int cls = n0 / Nc;
        st.py = cls >> 1;
        st.px = cls & 1;
    }
    __device__ __forceinline__ void fix(St& st, int k) const {
        const int q = __builtin_amdgcn_readfirstlane(k >> 5);
        const int cc = q >> 2, r = (q >> 1) & 1, c = q & 1;
        const int ty = st.py ? 2 * r : 1, tx = st.px ? 2 * c : 1;   // masked slots read a valid (unused) tap
        st.base = wp + (size_t)(ty * 3 + tx) * M * Cp + cc * 32;
    }
    __device__ __forceinline__ float get_u(const St& st, int m_u, int) const {
        const float* rp = st.base + (size_t)m_u * Cp;
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rp) + st.voff);
    }
};

struct DgradS2St {
    int img_rel, i, j, img0, py, px;
    const float* rowp;
    unsigned voff;
    int nm1, ok;
};
struct DgradS2B {  // B[k=(cc, slot, co)][n=(class, img, i, j)]
    static constexpr bool ALONG_K = false;
    static constexpr bool POST = true;
    typedef DgradS2St St;
    const float* dy;
    int Cp, Nc, H2, W2, Cout, OH, OW;
    __device__ __forceinline__ void init(St& st, int p, int p0) const {
        const int cls = p0 / Nc;                 // tile-uniform
        st.py = cls >> 1;
        st.px = cls & 1;
        const int hw2 = H2 * W2;
        const int q = min(p - cls * Nc, Nc - 1), q0 = p0 - cls * Nc;
        const int img = q / hw2, pix = q - img * hw2;
        st.i = pix / W2;
        st.j = pix - st.i * W2;
        st.img0 = q0 / hw2;
        st.img_rel = img - st.img0;
        st.rowp = dy;
        st.voff = 0;
        st.nm1 = 0;
        st.ok = 0;
    }
    __device__ __forceinline__ void chunk(St& st, int kc) const {
        const int q = kc >> 5;
        const int cc = q >> 2, r = (q >> 1) & 1, c = q & 1;
        const int co0 = cc << 5;
        // odd rows: taps 0 and 2 -> dY rows i+1 and i; even rows: tap 1 -> dY row i (slot r = 1 does not exist)
        int oy = st.i + st.py - r, ox = st.j + st.px - c;
        st.ok = (st.py || r == 0) && (st.px || c == 0) && oy < OH && ox < OW;
        oy = min(max(oy, 0), OH - 1);
        ox = min(max(ox, 0), OW - 1);
        const int ohw = OH * OW;
        st.voff = (unsigned)(st.img_rel * Cout * ohw + oy * OW + ox) * 4u;
        st.rowp = dy + (size_t)(st.img0 * Cout + co0) * ohw;
        st.nm1 = min(32, Cout - co0) - 1;
    }
    __device__ __forceinline__ float get(const St& st, int kl, int) const {
        const float* rp = st.rowp + (size_t)min(kl, st.nm1) * (OH * OW);
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rp) + st.voff);
    }
    static constexpr bool ALL_OK = true;
    __device__ __forceinline__ bool all_ok(const St& st) const { return __all(st.ok); }
    __device__ __forceinline__ float post(const St& st, float v, int) const { return st.ok ? v : 0.f; }
};
struct DgradS2Epi {  // dx[img][ci][2i+py][2j+px] (= or +=) acc
    typedef size_t St;
    float* dx;
    int Cin, Nc, H2, W2, accumulate;
    __device__ __forceinline__ St col(int n) const {
        const int cls = n / Nc, q = n - cls * Nc;
        const int hw2 = H2 * W2;
        const int img = q / hw2, pix = q - img * hw2;
        const int i = pix / W2, j = pix - i * W2;
        return (size_t)img * Cin * (4 * hw2) + (size_t)(2 * i + (cls >> 1)) * (2 * W2) + 2 * j + (cls & 1);
    }
    __device__ __forceinline__ void put(St base, int m, float v) const {
        float* q = dx + base + (size_t)m * (4 * H2 * W2);
        *q = accumulate ? (*q + v) : v;
    }
};

struct DgradS2PointEpi {  // dgrad of a 1x1 stride-2 conv: dx[img][ci][2i][2j] (= or +=) acc, the other three pixels of the 2x2 cell = 0
    typedef size_t St;
    float* dx;
    int Cin, ohw, OW, accumulate;
    __device__ __forceinline__ St col(int p) const {
        const int img = p / ohw, pix = p - img * ohw;
        const int i = pix / OW, j = pix - i * OW;
        return (size_t)img * Cin * (4 * ohw) + (size_t)(2 * i) * (2 * OW) + 2 * j;
    }
    __device__ __forceinline__ void put(St base, int m, float v) const {
        float* q = dx + base + (size_t)m * (4 * ohw);
        if (accumulate) { *q += v; return; }
        *reinterpret_cast<float2*>(q) = make_float2(v, 0.f);
        *reinterpret_cast<float2*>(q + 2 * OW) = make_float2(0.f, 0.f);
    }
};

// Border pass of the reflection-pad adjoint.  N enumerates the 2W+2H border-adjacent pixels of every image
// (rows 1 and H-2, columns 1 and W-2; duplicates masked); the gather returns only the folded-in ("extra")
// dY entries: row extras r1 = 0 (y==1, ty==0) / H-1 (y==H-2, ty==2), column extras likewise.
struct DgradBorderSt {
    InPixSt px;
    int img0;
    const float* rowp;
    unsigned v1, v2, v3;
    float m1, m2, m3;
    int nm1;
};
__device__ __forceinline__ InPixSt border_pix(int b, int Nb, int H, int W) {
    InPixSt px;
    const int per = 2 * W + 2 * H;
    px.valid = b < Nb;
    px.img = b / per;
    const int i = b - px.img * per;
    int y, x;
    bool dup = false;
    if (i < W) { y = 1; x = i; }
    else if (i < 2 * W) { y = H - 2; x = i - W; dup = (H - 2 == 1); }
    else if (i < 2 * W + H) { y = i - 2 * W; x = 1; dup = (y == 1 || y == H - 2); }
    else { y = i - 2 * W - H; x = W - 2; dup = (y == 1 || y == H - 2) || (W - 2 == 1); }
    px.y = y; px.x = x;
    if (dup) px.valid = 0;
    return px;
}

template <int KH>
struct DgradBorderB {   // scalar-base: uniform dY plane pointer + three per-lane byte offsets with 0/1 weights, no branches
    static constexpr bool ALONG_K = false;
    static constexpr bool POST = true;
    typedef DgradBorderSt St;
    const float* dy;
    int Cp, Nb, H, W, Cout;
    __device__ __forceinline__ void init(St& st, int b, int b0) const {
        st.px = border_pix(min(b, Nb - 1), Nb, H, W);
        if (b >= Nb) st.px.valid = 0;
        st.img0 = border_pix(min(b0, Nb - 1), Nb, H, W).img;
        st.rowp = dy;
        st.v1 = st.v2 = st.v3 = 0;
        st.m1 = st.m2 = st.m3 = 0.f;
        st.nm1 = 0;
    }
    __device__ __forceinline__ void chunk(St& st, int kc) const {
        const int q = kc >> 5;
        const int cc = q / (KH * KH), tap = q - cc * (KH * KH);
        const int co0 = cc << 5;
        const int ty = tap / KH, tx = tap - ty * KH;
        const DyOffs d = dy_offsets(st.px.y, st.px.x, ty, tx, H, W, H, W, 1, 1, 1);
        const bool ok = st.px.valid;
        const unsigned base = (unsigned)((st.px.img - st.img0) * Cout * H * W);
        st.m1 = (ok && d.o1 >= 0) ? 1.f : 0.f;
        st.m2 = (ok && d.o2 >= 0) ? 1.f : 0.f;
        st.m3 = (ok && d.o3 >= 0) ? 1.f : 0.f;
        st.v1 = (base + (unsigned)max(d.o1, 0)) * 4u;
        st.v2 = (base + (unsigned)max(d.o2, 0)) * 4u;
        st.v3 = (base + (unsigned)max(d.o3, 0)) * 4u;
        st.rowp = dy + (size_t)(st.img0 * Cout + co0) * H * W;
        st.nm1 = min(32, Cout - co0) - 1;
    }
    __device__ __forceinline__ float get(const St& st, int kl, int) const {
        const char* rp = reinterpret_cast<const char*>(st.rowp + (size_t)min(kl, st.nm1) * H * W);
        const float a = *reinterpret_cast<const float*>(rp + st.v1);
        const float b = *reinterpret_cast<const float*>(rp + st.v2);
        const float c = *reinterpret_cast<const float*>(rp + st.v3);
        return fmaf(st.m1, a, fmaf(st.m2, b, st.m3 * c));
    }
    __device__ __forceinline__ float post(const St&, float v, int) const { return v; }
    // Only the taps that fold something in are non-zero: (ty, tx) with a row extra at ty (y == 1 / H-2) or a column
    // extra at tx (x == 1 / W-2).  A pixel tile lies along one edge, so it needs 3 of the 9 taps (5 next to a corner):
    // the engine steps over the other K chunks.
    static constexpr bool SKIP = true;
    __device__ __forceinline__ unsigned tile_mask(const St& st) const {
        __shared__ unsigned tm;
        if (threadIdx.x == 0) tm = 0;
        __syncthreads();
        unsigned mine = 0;
        if (st.px.valid) {
            const unsigned ry = (st.px.y == 1 ? 1u : 0u) | (st.px.y == H - 2 ? 4u : 0u);
            const unsigned cx = (st.px.x == 1 ? 1u : 0u) | (st.px.x == W - 2 ? 4u : 0u);
#pragma unroll
            for (int ty = 0; ty < KH; ++ty)
#pragma unroll
                for (int tx = 0; tx < KH; ++tx)
                    if (((ry >> ty) | (cx >> tx)) & 1u) mine |= 1u << (ty * KH + tx);
        }
        if (mine) atomicOr(&tm, mine);
        __syncthreads();
        return (unsigned)__builtin_amdgcn_readfirstlane((int)tm);
    }
    __device__ __forceinline__ bool skip(unsigned m, int kc) const { return !((m >> ((kc >> 5) % (KH * KH))) & 1u); }
};

// Split policy of the border passes.  Their cost is the scattered read-modify-write of the epilogue (one 4-byte access per
// 64-byte sector, atomics when K is split: 12.6 M atomics = 0.24 ms for a 256-channel 256x256 layer with 6 splits), not
// the K loop -- so K is split only when the pass would otherwise be a few dozen workgroups.
static inline int border_splits(long btiles, int chunks) {
    if (btiles >= 48) return 1;
    return (int)std::max<long>(1, std::min<long>(jp_cdiv(96, btiles), chunks / 4));
}

struct DgradBorderEpi {  // dx[img][ci][y][x] += acc for the border pixel b
    typedef long St;
    float* dx;
    int Cin, H, W, Nb, split;
    __device__ __forceinline__ St col(int b) const {
        const InPixSt px = border_pix(b, Nb, H, W);
        return px.valid ? (long)px.img * Cin * H * W + px.y * W + px.x : -1;
    }
    __device__ __forceinline__ void put(St base, int m, float v) const {
        if (base < 0) return;
        float* q = dx + base + (size_t)m * H * W;
        if (split) atomicAdd(q, v);       // split-K partials meet here
        else *q += v;                     // one workgroup owns the pixel: plain read-modify-write
    }
};

// border pass through scratch: the K loop is split into slices that store part[slice][ci][b] (coalesced along the border pixels),
// then ONE small pass folds the slices into dx in a fixed order -- more workgroups on a launch that has only a few dozen tiles,
// the scattered read-modify-write done once, no atomics
// amax != nullptr: max |final value| of the pixels this pass touches goes to the slot the main pass reported into -- every element of dx
// is then covered by one of the two (an upper bound: the main pass's value of a border pixel is in there as well)
__global__ void border_add_kernel(const float* __restrict__ part, float* __restrict__ dx, int Cin, int Nb, int H, int W,
                                  int slices, unsigned* __restrict__ amax) {
    const long total = (long)Cin * Nb;
    float mx = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / Nb), b = (int)(i - (long)m * Nb);
        const InPixSt px = border_pix(b, Nb, H, W);
        if (!px.valid) continue;
        float s = 0.f;
        for (int k = 0; k < slices; ++k) s += part[(size_t)k * total + i];
        float* q = dx + ((size_t)px.img * Cin + m) * H * W + px.y * W + px.x;
        const float r = *q + s;
        *q = r;
        mx = fmaxf(mx, jp_fmag(r));
    }
    jp_wave_amax_commit(mx, amax);
}

// the same fold for the edge pass of an upsampled segment (boundary pixels of the half-resolution map, edge_pix)
__global__ void edge_add_kernel(const float* __restrict__ part, float* __restrict__ dx, int C, int Nb, int h, int w, int slices) {
    const long total = (long)C * Nb;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / Nb), b = (int)(i - (long)m * Nb);
        const InPixSt px = edge_pix(b, Nb, h, w);
        if (!px.valid) continue;
        float s = 0.f;
        for (int k = 0; k < slices; ++k) s += part[(size_t)k * total + i];
        dx[((size_t)px.img * C + m) * h * w + px.y * w + px.x] += s;
    }
}

template <int KH>
struct WgradBT {  // B[k=pixel][n=(tap,ci)], any source layout (per-element decode)
    static constexpr bool ALONG_K = true;
    typedef PixSt St;
    Src3 src;
    int Np, Cp, Cin, Npix, OH, OW, stride, pad, reflect;
    unsigned magic;   // floor(2^32 / Cp) + 1: n / Cp == umulhi(n, magic) for n < 2^16
    __device__ __forceinline__ void init(St&, int, int) const {}
    __device__ __forceinline__ void fix(St& st, int p) const { st = conv_pix(p, Npix, OH, OW, stride, pad); }
    __device__ __forceinline__ float get(const St& st, int n, int) const {
        if (!st.valid || n >= Np) return 0.f;
        const int tap = (Cp == 1 ? n : (int)__umulhi((unsigned)n, magic));
        const int ci = n - tap * Cp;
        if (ci >= Cin) return 0.f;
        const int dy = tap / KH, dx = tap - dy * KH;
        int iy = st.iy0 + dy, ix = st.ix0 + dx;
        if (reflect) {
            iy = jp_reflect(iy, src.H);
            ix = jp_reflect(ix, src.W);
        } else if ((unsigned)iy >= (unsigned)src.H || (unsigned)ix >= (unsigned)src.W) {
            return 0.f;
        }
        return src.at(st.img, ci, iy, ix);
    }
};

// single full-resolution source: the (tap, ci) decode of every slot is done ONCE per thread (the N index of a
// slot never changes over the K loop), leaving ~8 integer ops per gathered element.
struct WgradB1St {
    PixSt px;
    const float* base;   // x + img*Cin*H*W
    int off[32];         // ci*H*W, or -1 for padded / out-of-range slots
    int dd[32];          // dy | dx << 8
};

template <int KH>
struct WgradBT1 {
    static constexpr bool ALONG_K = true;
    typedef WgradB1St St;
    const float* x;      // already offset to the first channel of the sub-range
    int Np, Cp, Cin, Ctot, H, W, Npix, OH, OW, stride, pad, reflect;   // Cin = channels in this sub-range
    unsigned magic;
    __device__ __forceinline__ void init(St& st, int n_first, int step) const {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int n = n_first + step * r;
            const int tap = (Cp == 1 ? n : (int)__umulhi((unsigned)n, magic));
            const int ci = n - tap * Cp;
            const int dy = tap / KH, dx = tap - dy * KH;
            st.off[r] = (n < Np && ci < Cin) ? ci * H * W : -1;
            st.dd[r] = dy | (dx << 8);
        }
    }
    __device__ __forceinline__ void fix(St& st, int p) const {
        st.px = conv_pix(p, Npix, OH, OW, stride, pad);
        st.base = x + (size_t)st.px.img * Ctot * H * W;
    }
    __device__ __forceinline__ float get(const St& st, int, int r) const {
        if (!st.px.valid || st.off[r] < 0) return 0.f;
        int iy = st.px.iy0 + (st.dd[r] & 255), ix = st.px.ix0 + (st.dd[r] >> 8);
        if (reflect) {
            iy = jp_reflect(iy, H);
            ix = jp_reflect(ix, W);
        } else if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) {
            return 0.f;
        }
        return st.base[st.off[r] + iy * W + ix];
    }
};

// Channel counts that are multiples of the N tile (128): every workgroup's N tile lies inside ONE filter tap, so
// the tap shift, bounds / reflection and the source pointer are per-chunk work and each element is one strided load
// (same cost as the forward gather, ~100 fewer VGPRs than the table version -> 3 waves/SIMD).
struct WgradBUSt {
    const float* q;   // x element (first slot's channel) at this chunk's pixel shifted by the tile's tap
    int dy, dx, c0, ok, step;
};
template <int KH>
struct WgradBU {
    static constexpr bool ALONG_K = true;
    typedef WgradBUSt St;
    const float* x;
    int Cpp, Cin, Ctot, H, W, Npix, OH, OW, stride, pad, reflect;
    __device__ __forceinline__ void init(St& st, int n_first, int step) const {
        st.step = step;
        const int tap = n_first / Cpp;          // Cpp % tile == 0: identical for every slot of the workgroup
        st.c0 = n_first - tap * Cpp;
        st.dy = tap / KH;
        st.dx = tap - st.dy * KH;
        st.q = nullptr;
        st.ok = 0;
    }
    __device__ __forceinline__ void fix(St& st, int p) const {
        const PixSt px = conv_pix(p, Npix, OH, OW, stride, pad);
        int iy = px.iy0 + st.dy, ix = px.ix0 + st.dx;
        st.ok = px.valid;
        if (reflect) {
            iy = jp_reflect(iy, H);
            ix = jp_reflect(ix, W);
        } else if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) {
            st.ok = 0;
        }
        if (st.ok) st.q = x + ((size_t)(px.img * Ctot + st.c0) * H + iy) * W + ix;
    }
    __device__ __forceinline__ float get(const St& st, int, int r) const {
        // slot r holds channel c0 + step*r (step = slot stride of the lanes-along-K mapping)
        return (st.ok && st.c0 + st.step * r < Cin) ? st.q[(size_t)(st.step * r) * H * W] : 0.f;
    }
};

// ---- scalar-base wgrad loaders (igemm.h SPLIT protocol).  Preconditions (host-checked): OH*OW % 32 == 0, so a
// K chunk of 32 pixels never straddles two images (the image and the chunk's first pixel are wave-uniform), and
// Cout % 8 == 0, so the 8-row slot groups can be clamped with scalar arithmetic.
struct WgradASSt {
    const float* base;   // uniform: dY (image of the chunk, channel 0, first pixel of the chunk)
    unsigned voff;       // per-lane: (row within the slot group, pixel within the chunk), constant
};
struct WgradAS {  // A[m=co][k=pixel] = dY[img][co][pix]
    static constexpr bool ALONG_K = true;
    static constexpr bool SPLIT = true;
    typedef WgradASSt St;
    const float* dy;
    int Cout, Npix, OHW, P;   // P = pixels per image padded to 32: K = (img, padded pixel), the padding is masked on B
    __device__ __forceinline__ void init(St& st, int, int) const {
        st.base = dy;
        st.voff = ((threadIdx.x >> 5) * OHW + (threadIdx.x & 31)) * 4u;
    }
    __device__ __forceinline__ void fix(St& st, int p) const {
        const int p0 = __builtin_amdgcn_readfirstlane(p & ~31);
        const int img = p0 / P, pix0 = p0 - img * P;
        st.base = dy + (size_t)img * Cout * OHW + pix0;
        // lanes in the per-image K padding re-read the last pixel (their B operand is zeroed)
        st.voff = ((threadIdx.x >> 5) * OHW + min((int)(threadIdx.x & 31), OHW - 1 - pix0)) * 4u;
    }
    __device__ __forceinline__ float get_u(const St& st, int m_u, int) const {
        const float* rp = st.base + (size_t)min(m_u, Cout - 8) * OHW;   // rows >= Cout: duplicates, never stored
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rp) + st.voff);
    }
};

// uniform-tap B gather (Cin % 128 == 0, 128-wide N tile -> one filter tap per workgroup), scalar-base form
struct WgradBUSSt {
    int dy, dx, c0;      // uniform: the tile's tap and first channel
    const float* base;   // uniform: x (image of the chunk, channel c0)
    unsigned voff;       // per-lane: (channel within the slot group, tap-shifted pixel)
    int ok;
};
template <int KH, bool REFLECT>
struct WgradBUS {
    static constexpr bool ALONG_K = true;
    static constexpr bool SPLIT = true;
    static constexpr bool POST = true;
    typedef WgradBUSSt St;
    const float* x;
    int Cpp, Ctot, H, W, OH, OW, stride, pad, P;
    __device__ __forceinline__ void init(St& st, int n_first, int) const {
        const int n0 = __builtin_amdgcn_readfirstlane(n_first - (int)(threadIdx.x >> 5));
        const int tap = n0 / Cpp;
        st.c0 = n0 - tap * Cpp;
        st.dy = tap / KH;
        st.dx = tap - st.dy * KH;
        st.base = x;
        st.voff = 0;
        st.ok = 1;
    }
    __device__ __forceinline__ void fix(St& st, int p) const {
        const int ohw = OH * OW;
        const int p0 = __builtin_amdgcn_readfirstlane(p & ~31);
        const int img = p0 / P;                         // uniform
        const int pr = p - img * P;
        const int pix = min(pr, ohw - 1);
        const int oy = pix / OW, ox = pix - oy * OW;
        int iy = oy * stride - pad + st.dy, ix = ox * stride - pad + st.dx;
        st.ok = pr < ohw;                               // per-image K padding
        if (REFLECT) {
            iy = jp_reflect(iy, H);
            ix = jp_reflect(ix, W);
        } else {
            st.ok = st.ok && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            iy = min(max(iy, 0), H - 1);
            ix = min(max(ix, 0), W - 1);
        }
        st.voff = (unsigned)(((threadIdx.x >> 5) * H + iy) * W + ix) * 4u;
        st.base = x + (size_t)(img * Ctot + st.c0) * H * W;
    }
    __device__ __forceinline__ float get_u(const St& st, int, int r) const {   // slot r = channel c0 + 8r (+ lane part)
        const float* rp = st.base + (size_t)(8 * r) * H * W;
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rp) + st.voff);
    }
    static constexpr bool ALL_OK = true;
    __device__ __forceinline__ bool all_ok(const St& st) const { return __all(st.ok); }
    __device__ __forceinline__ float post(const St& st, float v, int) const { return st.ok ? v : 0.f; }
};

// CPT-channel inputs (CPT = 16, 32 or 64): a CPT*TPT wide N tile holds TPT whole filter taps, and slot group r (8
// columns) belongs to tap slot 8r/CPT at compile time -> one per-lane offset per tap per chunk, every element one
// scalar-base load.
template <int TPT>
struct WgradBMSSt {
    int tap0;            // uniform: first tap of the tile
    const float* base;   // uniform: x (image of the chunk, first channel of the sub-range)
    unsigned voff[TPT];  // per-lane: (channel within the slot group, pixel shifted by tap slot s)
    unsigned ok;         // per-lane bit s: tap slot s lies inside the image (zero padding)
};
template <int KH, bool REFLECT, int TPT, int CPT>
struct WgradBMS {
    static constexpr bool ALONG_K = true;
    static constexpr bool SPLIT = true;
    static constexpr bool POST = true;
    typedef WgradBMSSt<TPT> St;
    const float* x;      // already offset to the first channel of the CPT-channel sub-range
    int Ctot, H, W, OH, OW, stride, pad, P;
    __device__ __forceinline__ void init(St& st, int n_first, int) const {
        const int n0 = __builtin_amdgcn_readfirstlane(n_first - (int)(threadIdx.x >> 5));
        st.tap0 = n0 / CPT;
        st.base = x;
        st.ok = ~0u;
#pragma unroll
        for (int s = 0; s < TPT; ++s) st.voff[s] = 0;
    }
    __device__ __forceinline__ void fix(St& st, int p) const {
        const int ohw = OH * OW;
        const int p0 = __builtin_amdgcn_readfirstlane(p & ~31);
        const int img = p0 / P;                         // uniform
        const int pr = p - img * P;
        const int pix = min(pr, ohw - 1);
        const int oy = pix / OW, ox = pix - oy * OW;
        const int by = oy * stride - pad, bx = ox * stride - pad;
        unsigned ok = 0;
#pragma unroll
        for (int s = 0; s < TPT; ++s) {
            const int tap = min(st.tap0 + s, KH * KH - 1);   // taps past the filter: duplicate columns, never stored
            const int dy = tap / KH, dx = tap - dy * KH;
            int iy = by + dy, ix = bx + dx;
            if (REFLECT) {
                ok |= 1u << s;
                iy = jp_reflect(iy, H);
                ix = jp_reflect(ix, W);
            } else {
                ok |= (unsigned)((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) << s;
                iy = min(max(iy, 0), H - 1);
                ix = min(max(ix, 0), W - 1);
            }
            st.voff[s] = (unsigned)(((threadIdx.x >> 5) * H + iy) * W + ix) * 4u;
        }
        st.ok = pr < ohw ? ok : 0u;                     // per-image K padding
        st.base = x + (size_t)img * Ctot * H * W;
    }
    __device__ __forceinline__ float get_u(const St& st, int, int r) const {
        const float* rp = st.base + (size_t)((8 * r) % CPT) * H * W;
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rp) + st.voff[(8 * r) / CPT]);
    }
    static constexpr bool ALL_OK = true;
    __device__ __forceinline__ bool all_ok(const St& st) const { return __all(st.ok == ((1u << TPT) - 1u)); }
    __device__ __forceinline__ float post(const St& st, float v, int r) const {
        return ((st.ok >> ((8 * r) / CPT)) & 1u) ? v : 0.f;
    }
};

struct WgradEpiT {  // dw[co][c_off + ci][tap] += acc for n = tap*Cp + ci
    typedef int St;
    float* dw;
    int Cp, Cin, KHW, c_off, Ctot;
    unsigned magic;
    __device__ __forceinline__ St col(int n) const {
        const int tap = (Cp == 1 ? n : (int)__umulhi((unsigned)n, magic));
        const int ci = n - tap * Cp;
        return ci < Cin ? (c_off + ci) * KHW + tap : -1;
    }
    __device__ __forceinline__ void put(St j, int m, float v) const {
        if (j >= 0) atomicAdd(dw + (size_t)m * Ctot * KHW + j, v);
    }
};

struct WgradEpiWS {  // split-K partial tiles as plain stores into caller scratch ws[split][m][n]; wgrad_reduce sums them
    typedef int St;
    float* ws;
    int M, Np;
    __device__ __forceinline__ St col(int n) const { return n; }
    static constexpr bool WANTS_SLICE = true;
    __device__ __forceinline__ void put(St n, int m, float v, int slice) const { ws[((size_t)slice * M + m) * Np + n] = v; }
};

// dw[m][c_off + ci][tap] += sum_s ws[s][m][n = tap*Cp + ci]
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int M, int Np, int splits,
                                    int Cp, int Cin, int KHW, int c_off, int Ctot) {
    const long total = (long)M * Np;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / Np), n = (int)(i - (long)m * Np);
        const int tap = n / Cp, ci = n - tap * Cp;
        if (ci >= Cin) continue;
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += ws[(size_t)k * total + i];
        dw[((size_t)m * Ctot + c_off + ci) * KHW + tap] += s;
    }
}

// W9 partials (Np % 4 == 0, Cp == Cin): 4 consecutive n per thread as one 16-byte load per slice, SL slice lanes per output
// quad (blockDim = 64 x SL) so that few-output / many-slice layers still put enough loads in flight
template <int SL>
__global__ __launch_bounds__(64 * SL) void wgrad_reduce4_kernel(const float* __restrict__ ws, float* __restrict__ dw, int M,
                                                               int Np, int splits, int Cp, int KHW, int c_off, int Ctot) {
    __shared__ float4 part[SL][64];
    const long total4 = (long)M * Np / 4;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long i4 = (long)blockIdx.x * 64 + tx;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 < total4) {
        const float4* p = reinterpret_cast<const float4*>(ws) + i4;
#pragma unroll 4
        for (int k = ty; k < splits; k += SL) {
            const float4 v = p[(long)k * total4];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    if (SL > 1) {
        part[ty][tx] = s;
        __syncthreads();
        if (ty != 0) return;
#pragma unroll
        for (int k = 1; k < SL; ++k) { const float4 v = part[k][tx]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    }
    if (i4 < total4) {
        const long i = i4 * 4;
        const int m = (int)(i / Np), n = (int)(i - (long)m * Np);
        const int tap = n / Cp, ci = n - tap * Cp;          // the 4 values share the tap (Cp % 4 == 0)
        float* q = dw + ((size_t)m * Ctot + c_off + ci) * KHW + tap;
        q[0] += s.x; q[KHW] += s.y; q[2 * KHW] += s.z; q[3 * KHW] += s.w;
    }
}

// many slices, few outputs (narrow layers: 16x144 outputs x ~700 slices): 64 outputs x 16 slice lanes per workgroup
__global__ __launch_bounds__(1024) void wgrad_reduce_wide_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                                 int M, int Np, int splits, int Cp, int Cin, int KHW,
                                                                 int c_off, int Ctot) {
    __shared__ float part[16][65];
    const long total = (long)M * Np;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + tx;
    float s = 0.f;
    if (i < total)
        for (int k = ty; k < splits; k += 16) s += ws[(size_t)k * total + i];
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < total) {
#pragma unroll
        for (int k = 1; k < 16; ++k) s += part[k][tx];
        const int m = (int)(i / Np), n = (int)(i - (long)m * Np);
        const int tap = n / Cp, ci = n - tap * Cp;
        if (ci < Cin) dw[((size_t)m * Ctot + c_off + ci) * KHW + tap] += s;
    }
}

// wp[tap][row][Cp]: forward rows = co (src W[co][ci][tap]); dgrad rows = ci, reduction = co
constexpr int KC = 32;

// split-K plan of a wgrad GEMM (M x Np, K = npix): workgroups run in rounds of `slots` = 256 CUs x per_cu; a split
// costs its K chunks plus an epilogue -- ~14 chunk-times when the partial tile is merged with device-scope atomics,
// ~3 when it is stored to scratch and summed by wgrad_reduce_kernel (whose pass over splits*M*Np floats is charged
// too).  Returns the split count minimising rounds * (chunks per split + epilogue).
struct WgradPlan {
    int splits, kps;
    bool use_ws;
    long ws_need;
};
inline WgradPlan wgrad_plan(int M, int Np, long npix, int BM, int BN, int per_cu, long ws_floats) {
    const long tiles = (long)jp_cdiv(M, BM) * jp_cdiv(Np, BN), chunks = jp_cdiv(npix, KC), slots = 256L * per_cu;
    WgradPlan best{1, (int)(chunks * KC), false, 0};
    double bc = 1e30;
    for (long sp = 1; sp <= std::max<long>(1, chunks / 4) && sp <= 4096; ++sp) {
        const long per = jp_cdiv(chunks, sp), need = sp * (long)M * Np;
        const double rounds = (double)jp_cdiv(sp * tiles, slots);
        const double c_at = rounds * (per + 14.0);
        if (c_at < bc * 0.999) { bc = c_at; best = WgradPlan{(int)sp, (int)(per * KC), false, 0}; }
        if (sp > 1 && need <= ws_floats) {
            const double c_ws = rounds * (per + 3.0) + (need * 4.0 / 3.0e12 + 6e-6) / 7e-6;
            if (c_ws < bc * 0.999) { bc = c_ws; best = WgradPlan{(int)sp, (int)(per * KC), true, need}; }
        }
    }
    best.splits = jp_cdiv(npix, best.kps);
    return best;
}

// IL: gather interleaved with the MFMAs (measured +6 % on wgrad, -6 % on fwd/dgrad -> wgrad only)
template <bool IL, int WM, int WN, class A, class B, class E>
void launch(A a, B b, E e, int M, int N, int K, int splits, int kps, hipStream_t st) {
    dim3 grid(jp_cdiv(N, 64 * WN), jp_cdiv(M, 64 * WM), splits);
    jp_prof_before(__PRETTY_FUNCTION__, 2.0 * M * (double)N * K, st);
    hipLaunchKernelGGL((jp_igemm_kernel<WM, WN, KC, A, B, E, false, IL>), grid, dim3(64 * WM * WN), 0, st, a, b, e, M, N, K, kps);
    jp_prof_after(st);
}

template <bool IL = false, class A, class B, class E>
void launch_auto(A a, B b, E e, int M, int N, int K, int splits, int kps, hipStream_t st) {
    if (M <= 64) launch<IL, 1, 4>(a, b, e, M, N, K, splits, kps, st);
    else if (N <= 64) launch<IL, 4, 1>(a, b, e, M, N, K, splits, kps, st);
    else launch<IL, 2, 2>(a, b, e, M, N, K, splits, kps, st);
}

template <int WM, int WN, class A, class B, class E>
void launch_r3(A a, B b, E e, int M, int N, int K, hipStream_t st) {
    dim3 grid(N / (64 * WN), jp_cdiv(M, 64 * WM), 1);
    jp_prof_before(__PRETTY_FUNCTION__, 2.0 * M * (double)N * K, st);
    hipLaunchKernelGGL((jp_igemm_r3_kernel<WM, WN, KC, A, B, E>), grid, dim3(256), 0, st, a, b, e, M, N, K);
    jp_prof_after(st);
}

// P9 patch kernel (igemm_p9.h): 3x3 stride 1 pad 1, single full-resolution source.  JP_P9=0 in the environment turns
// it off (A/B measurements); the pack needs p9_ws_floats() floats of scratch.
inline bool p9_enabled() {
    static const int on = [] { const char* e = getenv("JP_P9"); return e ? atoi(e) : 1; }();
    return on != 0;
}
inline bool p9u_enabled() {
    static const int on = [] { const char* e = getenv("JP_P9U"); return e ? atoi(e) : 1; }();
    return on != 0;
}
template <class E>
const char* p9u_tag() { return __PRETTY_FUNCTION__; }
inline bool p9us_enabled() {
    static const int on = [] { const char* e = getenv("JP_P9US"); return e ? atoi(e) : 1; }();
    return on != 0;
}
template <class E>
const char* p9us2_tag() { return __PRETTY_FUNCTION__; }
// P9SD (igemm_p9sd.h): dgrad of the upsampled iconv segment at half resolution on the bf16 pipe; JP_P9SD=0 keeps DgradUPB
inline bool p9sd_enabled() {
    static const int on = [] { const char* e = getenv("JP_P9SD"); return e ? atoi(e) : 1; }();
    return on != 0;
}
inline long p9sd_floats(int rows, int Cout) { return (long)jp_cdiv(rows, 128) * ((long)((Cout + 31) / 32 * 2) * 16 + P9S_AHEAD) * (JP_NS * 1024) + JP_PACK_HDR; }
template <class E>
const char* p9sd_tag() { return __PRETTY_FUNCTION__; }
// P9S2D (igemm_p9s2d.h): class-uniform split-bf16 dgrad of the 3x3 stride-2 layers; JP_P9S2=0 keeps the generic DgradS2B form
inline bool p9s2_enabled() {
    static const bool on = [] { const char* e = getenv("JP_P9S2"); return !(e && e[0] == '0'); }();
    return on;
}
// per-kernel switches below JP_P9S2 (A/B and bisection): JP_P9S2D / JP_P9S2F / JP_P1S2 / JP_P7S = 0 keep the generic engine
#define JP_ENV_ON(NAME) ([] { static const bool on = [] { const char* e = getenv(NAME); return !(e && e[0] == '0'); }(); return on; }())
template <int WM, int WN, class E>
const char* p9s2d_tag() { return __PRETTY_FUNCTION__; }
template <class E>
const char* p9s2f_tag() { return __PRETTY_FUNCTION__; }
template <int CIN, class E>
const char* p7s_tag() { return __PRETTY_FUNCTION__; }
// channels per M tile of a bank with `rows` rows: 64 x (8x32 px), 128 x (4x32 px), or -- 3x3 banks whose row count is a
// multiple of 256 -- 256 x (4x32 px) on 8 waves (JP_P9_M256=0 turns that variant off)
inline bool p9_m256() {
    static const int on = [] { const char* e = getenv("JP_P9_M256"); return e ? atoi(e) : 1; }();
    return on != 0;
}
// `ptiles` = N * (H/4) * (W/32) pixel tiles of the launch (the conv entry points know it when they pack: a layer's pack is
// keyed by its shape on the host side); the 8-wave variant needs >= 256 workgroups or it leaves CUs empty
// (512->512 @32x32: 141 -> 92 TF), where it has them it is 2-4 % faster (256->256 @128x128: 141 -> 146 TF)
// (1x1 layers keep the 4x32-pixel tiles: 8x32 ones and 128-row 4-wave tiles were measured in round 4, profiles/r04_p1_tile_ab.log,
// r04_p1_tile3_ab.log; 4-wave 256-row workgroups, two per CU, in round 5: -7..-9 % alone, +0.3 ms in the step, profiles/r05_x_ab.log)
inline bool p9_wide256(int rows, long ptiles) { return p9_m256() && rows % 256 == 0 && ptiles * (rows / 256) >= 256; }
inline int p9_bmt(int rows, int khw = 9, long ptiles = 0) {
    if (rows <= 64) return 64;
    return p9_wide256(rows, ptiles) ? 256 : 128;
}
inline long p9_ptiles(int N, int H, int W) { return (long)N * (H / 4) * (W / 32); }
inline long dgrad_tap_floats(int Cin, int Cout, int KH) { return ((long)(KH * KH + 16) * Cin + 512 + 64) * ((Cout + 31) / 32 * 32); }
inline long p9_ws_floats(int rows, int red, int khw = 9) {
    const int bmt = rows <= 64 ? 64 : 128;          // the 256-row tiling of the same bank never needs more
    return ((long)jp_cdiv(red, 32) * khw * 4 + P9_QAHEAD + 1) * 8 * bmt * jp_cdiv(rows, bmt);
}
// P9S (igemm_p9s.h): the same tiles with every fp32 product formed on the bf16 matrix pipe from three-way splits of both
// operands (6 MFMAs of 32 cycles instead of 8 of 64; fp32-equivalent accuracy).  JP_P9S=0 keeps the exact-fp32 P9 kernel.
inline bool p9s_enabled() {
    static const int on = [] { const char* e = getenv("JP_P9S"); return e ? atoi(e) : 1; }();
    return on != 0;
}
inline int p9s_kgs(int khw) { return khw == 9 ? 1 : 2; }       // 16-channel groups per stage
inline long p9s_ws_floats(int rows, int red, int khw = 9) {
    const int bmt = rows <= 64 ? 64 : 128, kgs = p9s_kgs(khw);
    return ((long)(red / (16 * kgs)) * khw * kgs + P9S_AHEAD) * (JP_NS * 8) * bmt * jp_cdiv(rows, bmt) + JP_PACK_HDR;
}
// scratch that holds either pack of a bank
inline long p9_alloc_floats(int rows, int red, int khw = 9) { return std::max(p9_ws_floats(rows, red, khw), p9s_ws_floats(rows, (red + 31) / 32 * 32, khw)); }
// fragment-order pack of a bank for the patch kernels: fp32 (PACK_FRAG) or bf16 splits (PACK_SPLIT)
inline void pack_p9(const float* w, float* wp, int Cout, int Cin, int for_dgrad, int bmt, int khw, hipStream_t st) {
    const int rows = for_dgrad ? Cin : Cout, red = for_dgrad ? Cout : Cin;
    if (p9s_enabled()) do_pack(PACK_SPLIT, w, wp, p9s_ws_floats(rows, red, khw), Cout, Cin, for_dgrad, bmt, khw, p9s_kgs(khw), st);
    else do_pack(PACK_FRAG, w, wp, p9_ws_floats(rows, red, khw), Cout, Cin, for_dgrad, bmt, khw, 0, st);
}
// JP_P1: 1 = every eligible 1x1 layer, 0 = none, unset = only banks that get the 256-channel 8-wave tiles (there the
// patch kernel wins: 256->256 @256x256 forward 101 -> 107 TF, dgrad 106 -> 113 TF; with 128-channel tiles it does not)
inline int p1_mode() {
    static const int m = [] { const char* e = getenv("JP_P1"); return e ? atoi(e) : 2; }();
    return m;
}
inline bool p9_ok(int rows, int red, int N, int H, int W, int khw = 9) {
    const int tr = rows <= 64 ? 8 : 4;
    const bool p1 = p1_mode() == 1 || (p1_mode() == 2 && p9_wide256(rows, p9_ptiles(N, H, W)));
    return p9_enabled() && (khw == 9 || p1) && rows >= 32 && red >= 32 && red % (khw == 1 ? 64 : 32) == 0 && W % 32 == 0 && H % tr == 0 &&
           (long)jp_cdiv(rows, p9_bmt(rows)) * N * (H / tr) * (W / 32) >= 192;
}
template <int WM, int WN, bool REFLECT, bool REV, class E, int TAPS = 9>
const char* p9_tag() { return __PRETTY_FUNCTION__; }       // profiler tag naming the instantiation
template <int WM, int WN, bool REFLECT, bool REV, class E, int TAPS = 9>
const char* p9s_tag() { return __PRETTY_FUNCTION__; }
template <int WM, int WN, bool REFLECT, bool REV, class E, int TAPS>
const char* p9sw_tag() { return __PRETTY_FUNCTION__; }
// JP_P9_TILE (3x3 layers): 0 = 4x32-pixel tiles (rounds 2-3), 1 = 8x32-pixel wide tiles for 256-row banks, 2 = also for
// 128-row tiles (jp_igemm_p9s_wide_kernel<2, 2, ...>, 4 waves), 3 (default) = also 16x32-pixel tiles for 64-row banks (<1, 4, ...>:
// 64->64 @256^2 forward / dgrad 0.229 / 0.233 -> 0.220 / 0.209 ms).  Whole step, same box, three runs each (r04_tile_step_ab.log):
// 85.61 ms (0), 84.77 (2), 84.69 (3).
// Same box (profiles/r04_p9_tile_ab.log): 256->256 reflect @256^2 forward / dgrad 2.652 / 2.631 -> 2.513 / 2.496 ms (1 477-1 487 TF
// executed = 0.59 of 2.5 PF), @128^2 0.717 / 0.677 -> 0.678 / 0.640; 128->128 @128^2 0.197 / 0.193 -> 0.189 / 0.183.
inline int p9_tile() {
    static const int m = [] { const char* e = getenv("JP_P9_TILE"); return e ? atoi(e) : 3; }();
    return m;
}
// P1L (igemm_p1l.h, round 5): the 256-row 1x1 layers as a persistent kernel with LDS-resident weight stages -- OPT-IN (JP_P1L=1).
// Alone it is 8-9 % faster than the patch kernel on the @256^2 layers (0.556 / 0.501 -> 0.512 / 0.456 ms forward / dgrad, same pack,
// bit-identical results: profiles/r05_p1l_conv_bench.log); in the overlapped step it LOSES: same-box pairs 82.4 -> 83.2 ms with it in
// both directions, 83.2 -> 83.7 ms forward-only (profiles/r05_p1l_step_ab.log, r05_p1l_fwd_only_step_ab.log) -- a persistent workgroup
// with 144 KB of LDS and 250 registers per lane owns its CU for the whole launch, while the patch kernel's short-lived workgroups
// (49 KB, ~90 registers) let the side streams' kernels in beside them.  tests/test_kernels_gpu.py::test_p1l_persistent_1x1 runs it.
template <class E>
const char* p1l_tag() { return __PRETTY_FUNCTION__; }
inline bool p1l_enabled() {
    static const bool on = [] { const char* e = getenv("JP_P1L"); return JP_NS == 3 && e && e[0] == '1'; }();     // (written for the three-plane bf16 pack)
    return on;
}
inline int jp_num_cus() {
    static const int n = [] {
        int dev = 0, v = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        return v > 0 ? v : 256;
    }();
    return n;
}
template <bool REFLECT, bool REV, class E, int TAPS>
void launch_p9s(const float* wp, const float* x, E e, int rows, int red, int N, int H, int W, const JpCall& st, int mt_off, int bmt) {
    constexpr int KGS = TAPS == 9 ? 1 : 2;
    const int NST = red / (16 * KGS);
    const unsigned* wq = reinterpret_cast<const unsigned*>(wp);
    const float* xam = JP_NS == 2 ? jp_amax_of(x, (long)N * red * H * W, st) : nullptr;
    if constexpr (JP_NS == 2 && jp_has_amax<E>::value) e.amax = jp_take_amax_out(st);       // (every kernel below reports it)
    if constexpr (jp_has_stats<E>::value) {
        // BatchNorm statistics in the epilogue: the 4-wave 3x3 kernels only (transposed accumulators: a lane owns one channel, the
        // sums over its pixels are plain register adds + one cross-half shuffle); partials per channel = pixel tiles x pixel-row waves
        e.stats = nullptr;
        if (TAPS == 9 && bmt != 256 && mt_off == 0 && st.ax && st.ax->stats) {
            const int mode = p9_tile();
            long tiles;
            if (bmt == 64) tiles = (mode >= 3 && H % 16 == 0 && (long)N * (H / 16) * (W / 32) >= 256) ? (long)N * (H / 16) * (W / 32) : (long)N * (H / 8) * (W / 32);
            else tiles = (mode >= 2 && H % 8 == 0 && (long)N * (H / 8) * (W / 32) * jp_cdiv(rows, bmt) >= 256) ? (long)N * (H / 8) * (W / 32) : (long)N * (H / 4) * (W / 32);
            e.stats = st.ax->stats;
            st.ax->stats_parts = (int)(tiles * (bmt == 64 ? 4 : 2));
        }
    }
#if JP_NS == 3
    if constexpr (TAPS == 1) {
        const long ntiles = (long)N * (H / 4) * (W / 32), xb = (long)N * red * H * W * 4;
        if (bmt == 256 && mt_off == 0 && p1l_enabled() && rows % 256 == 0 && red % 128 == 0 && H % 4 == 0 && W % 32 == 0 && xb < (1L << 31) &&
            ntiles >= 8L * jp_num_cus()) {     // (4 tiles per workgroup, the @128^2 layers: no gain over the patch kernel, profiles/r05_p1l_conv_bench.log)
            const int G = jp_num_cus(), tpw = jp_cdiv(ntiles, G);
            jp_prof_before(p1l_tag<E>(), JP_NPROD * 2.0 * rows * (double)N * H * W * red, st);
            hipLaunchKernelGGL((jp_conv1x1_p1l_kernel<E>), dim3(jp_cdiv(ntiles, tpw), rows / 256, 1), dim3(512), 0, st, wq, x, e, rows, red, NST,
                                   H, W, (int)ntiles, tpw, (int)xb);
            jp_prof_after(st);
            return;
        }
    }
#endif
    if constexpr (TAPS == 9) {
        // wide tiles (8 rows x 32 columns per workgroup, NJ = 4): only where they keep every CU busy (3x3 layers only)
        const int mode = p9_tile();
        if (TAPS == 9 && mode >= 3 && bmt == 64 && H % 16 == 0 && (long)N * (H / 16) * (W / 32) >= 256) {
            jp_prof_before(p9sw_tag<1, 4, REFLECT, REV, E, TAPS>(), JP_NPROD * 2.0 * rows * (double)N * H * W * TAPS * red, st);
            hipLaunchKernelGGL((jp_igemm_p9s_wide_kernel<1, 4, REFLECT, REV, E, TAPS, KGS>), dim3(N * (H / 16) * (W / 32), 1, 1), dim3(256), 0,
                               st, wq, x, e, rows, red, NST, H, W, mt_off, xam);
            jp_prof_after(st);
            return;
        }
        const bool want = (mode >= 1 && bmt == 256) || (mode >= 2 && bmt == 128);
        if (want && H % 8 == 0 && (long)N * (H / 8) * (W / 32) * jp_cdiv(rows, bmt) >= 256) {
            jp_prof_before(bmt == 256 ? p9sw_tag<4, 2, REFLECT, REV, E, TAPS>() : p9sw_tag<2, 2, REFLECT, REV, E, TAPS>(),
                           JP_NPROD * 2.0 * rows * (double)N * H * W * TAPS * red, st);
            dim3 grid(N * (H / 8) * (W / 32), jp_cdiv(rows, bmt), 1);
            if (bmt == 256)
                hipLaunchKernelGGL((jp_igemm_p9s_wide_kernel<4, 2, REFLECT, REV, E, TAPS, KGS>), grid, dim3(512), 0, st, wq, x, e, rows, red, NST, H, W, mt_off, xam);
            else
                hipLaunchKernelGGL((jp_igemm_p9s_wide_kernel<2, 2, REFLECT, REV, E, TAPS, KGS>), grid, dim3(256), 0, st, wq, x, e, rows, red, NST, H, W, mt_off, xam);
            jp_prof_after(st);
            return;
        }
    }
    // executed FLOPs: 6 bf16 MFMA products per fp32 product
    jp_prof_before(bmt == 64 ? p9s_tag<1, 4, REFLECT, REV, E, TAPS>() : (bmt == 256 ? p9s_tag<4, 2, REFLECT, REV, E, TAPS>() : p9s_tag<2, 2, REFLECT, REV, E, TAPS>()),
                   JP_NPROD * 2.0 * rows * (double)N * H * W * TAPS * red, st);
    if (bmt == 64) {
        dim3 grid(N * (H / 8) * (W / 32), 1, 1);
        hipLaunchKernelGGL((jp_igemm_p9s_kernel<1, 4, 2, REFLECT, REV, E, TAPS, KGS>), grid, dim3(256), 0, st, wq, x, e, rows, red, NST, H, W, mt_off, xam);
    } else if (bmt == 256) {
        dim3 grid(N * (H / 4) * (W / 32), jp_cdiv(rows, 256), 1);
        hipLaunchKernelGGL((jp_igemm_p9s_kernel<4, 2, 2, REFLECT, REV, E, TAPS, KGS>), grid, dim3(512), 0, st, wq, x, e, rows, red, NST, H, W, mt_off, xam);
    } else {
        dim3 grid(N * (H / 4) * (W / 32), jp_cdiv(rows, 128), 1);
        hipLaunchKernelGGL((jp_igemm_p9s_kernel<2, 2, 2, REFLECT, REV, E, TAPS, KGS>), grid, dim3(256), 0, st, wq, x, e, rows, red, NST, H, W, mt_off, xam);
    }
    jp_prof_after(st);
}
// 1x1 stride-2 (ResNet downsample branches) on the same tiles: forward = P9S with a stride-2 staging gather (igemm_p9s.h, XS = 2);
// dgrad = the stride-1 kernel over the half-resolution grid with a scattering epilogue (DgradS2PointEpi)
template <int WM, int WN, class E>
const char* p9sx2_tag() { return __PRETTY_FUNCTION__; }
inline bool p1s2_ok(int rows, int red, int N, int OH, int OW) {
    const int tr = rows <= 64 ? 8 : 4;
    return p9s_enabled() && p9s2_enabled() && JP_ENV_ON("JP_P1S2") && rows >= 32 && red >= 64 && red % 64 == 0 && OW % 32 == 0 && OH % tr == 0 &&
           (long)N * red * OH * OW * 16 < (1L << 31) &&
           (long)jp_cdiv(rows, p9_bmt(rows, 1, p9_ptiles(N, OH, OW))) * N * (OH / tr) * (OW / 32) >= 192;
}
template <class E>
void launch_p1s2(const float* wp, const float* x, E e, int rows, int red, int N, int OH, int OW, const JpCall& st) {
    const int bmt = p9_bmt(rows, 1, p9_ptiles(N, OH, OW)), NST = red / 32;
    const unsigned* wq = reinterpret_cast<const unsigned*>(wp);
    const float* xam = JP_NS == 2 ? jp_amax_of(x, (long)N * red * (2L * OH) * (2L * OW), st) : nullptr;
    jp_prof_before(bmt == 64 ? p9sx2_tag<1, 4, E>() : (bmt == 256 ? p9sx2_tag<4, 2, E>() : p9sx2_tag<2, 2, E>()),
                   JP_NPROD * 2.0 * rows * (double)N * OH * OW * red, st);
    if (bmt == 64)
        hipLaunchKernelGGL((jp_igemm_p9s_x2_kernel<1, 4, 2, E>), dim3(N * (OH / 8) * (OW / 32), 1, 1), dim3(256), 0, st, wq, x, e, rows, red, NST, OH, OW, 0, xam);
    else if (bmt == 256)
        hipLaunchKernelGGL((jp_igemm_p9s_x2_kernel<4, 2, 2, E>), dim3(N * (OH / 4) * (OW / 32), jp_cdiv(rows, 256), 1), dim3(512), 0, st, wq, x, e, rows, red, NST, OH, OW, 0, xam);
    else
        hipLaunchKernelGGL((jp_igemm_p9s_x2_kernel<2, 2, 2, E>), dim3(N * (OH / 4) * (OW / 32), jp_cdiv(rows, 128), 1), dim3(256), 0, st, wq, x, e, rows, red, NST, OH, OW, 0, xam);
    jp_prof_after(st);
}
template <bool REFLECT, bool REV, class E>
void launch_p9(const float* wp, const float* x, E e, int rows, int red, int N, int H, int W, const JpCall& st, int mt_off = 0,
               int bank_rows = -1) {
    const int NCH = jp_cdiv(red, 32);
    const int bmt = p9_bmt(bank_rows < 0 ? rows : bank_rows, 9, p9_ptiles(N, H, W));   // the M tile of the PACK (the bank's rows)
    if (p9s_enabled()) { launch_p9s<REFLECT, REV, E, 9>(wp, x, e, rows, red, N, H, W, st, mt_off, bmt); return; }
    jp_prof_before(bmt == 64 ? p9_tag<1, 4, REFLECT, REV, E>() : (bmt == 256 ? p9_tag<4, 2, REFLECT, REV, E>() : p9_tag<2, 2, REFLECT, REV, E>()),
                   2.0 * rows * (double)N * H * W * 9.0 * red, st);
    if (bmt == 64) {
        dim3 grid(N * (H / 8) * (W / 32), 1, 1);
        hipLaunchKernelGGL((jp_igemm_p9_kernel<1, 4, REFLECT, REV, E>), grid, dim3(256), 0, st, wp, x, e, rows, red, NCH, H, W, mt_off);
    } else if (bmt == 256) {
        dim3 grid(N * (H / 4) * (W / 32), jp_cdiv(rows, 256), 1);
        hipLaunchKernelGGL((jp_igemm_p9_kernel<4, 2, REFLECT, REV, E>), grid, dim3(512), 0, st, wp, x, e, rows, red, NCH, H, W, mt_off);
    } else {
        dim3 grid(N * (H / 4) * (W / 32), jp_cdiv(rows, 128), 1);
        hipLaunchKernelGGL((jp_igemm_p9_kernel<2, 2, REFLECT, REV, E>), grid, dim3(256), 0, st, wp, x, e, rows, red, NCH, H, W, mt_off);
    }
    jp_prof_after(st);
}
// 1x1 stride-1 convolution / its dgrad through the same kernel (TAPS = 1, two channel chunks per stage)
template <class E>
void launch_p1(const float* wp, const float* x, E e, int rows, int red, int N, int H, int W, const JpCall& st) {
    const int NST = jp_cdiv(red, 64);
    const int bmt = p9_bmt(rows, 1, p9_ptiles(N, H, W));
    if (p9s_enabled()) { launch_p9s<false, false, E, 1>(wp, x, e, rows, red, N, H, W, st, 0, bmt); return; }
    jp_prof_before(bmt == 64 ? p9_tag<1, 4, false, false, E, 1>() : (bmt == 256 ? p9_tag<4, 2, false, false, E, 1>() : p9_tag<2, 2, false, false, E, 1>()),
                   2.0 * rows * (double)N * H * W * red, st);
    if (bmt == 64) {
        dim3 grid(N * (H / 8) * (W / 32), 1, 1);
        hipLaunchKernelGGL((jp_igemm_p9_kernel<1, 4, false, false, E, 1, 2>), grid, dim3(256), 0, st, wp, x, e, rows, red, NST, H, W, 0);
    } else if (bmt == 256) {
        dim3 grid(N * (H / 4) * (W / 32), rows / 256, 1);
        hipLaunchKernelGGL((jp_igemm_p9_kernel<4, 2, false, false, E, 1, 2>), grid, dim3(512), 0, st, wp, x, e, rows, red, NST, H, W, 0);
    } else {
        dim3 grid(N * (H / 4) * (W / 32), jp_cdiv(rows, 128), 1);
        hipLaunchKernelGGL((jp_igemm_p9_kernel<2, 2, false, false, E, 1, 2>), grid, dim3(256), 0, st, wp, x, e, rows, red, NST, H, W, 0);
    }
    jp_prof_after(st);
}

// ---- W9 patch wgrad (igemm_w9.h): 3x3 s1 p1, single full-resolution source, >= 128 output channels, input channels a
// multiple of 64.  One 12-wave workgroup per CU: the K (pixel-tile) range is split so that ~256 workgroups exist.
static bool w9_enabled() {
    static const bool on = [] { const char* e = getenv("JP_W9"); return !(e && e[0] == '0'); }();
    return on;
}
// W9S (igemm_w9s.h): the wide variant (> 64 output channels) with split-bf16 products on the bf16 matrix pipe; JP_W9S=0
// keeps the exact-fp32 W9 kernel.  Its pixel tiles are W9S_TR rows high.
constexpr int W9S_TR = 2;
constexpr int W9S_TRN = 4;     // narrow variant (<= 64 output channels): two K groups of 2 rows each
// pixel-tile height of the 256x32-channel variant (NCB = 1): JP_W9S_TR1 = 2 | 4
static inline int w9s_tr1() {
    static const int v = [] { const char* e = getenv("JP_W9S_TR1"); return e ? atoi(e) : 4; }();
    return v == 2 ? 2 : 4;      // 4 (default): 1.5x instead of 2x patch re-staging, half the barriers: 2.738 -> 2.649 ms @256^2 (r04_w9s_ab.log)
}
static bool w9s_enabled() {
    static const bool on = [] { const char* e = getenv("JP_W9S"); return !(e && e[0] == '0'); }();
    return on;
}
struct W9Plan { int splits, tps, ntiles, slices, narrow, split_mfma, ncb1, tr; long need; };
// W9S with 256 output x 32 input channels per workgroup (igemm_w9s.h NCB = 1) for layers whose output channels fill 256-row tiles;
// JP_W9S_NCB=2 keeps the 128 x 64 tiles of round 3
static inline bool w9s_ncb1(int Cout, int narrow) {
    static const bool on = [] { const char* e = getenv("JP_W9S_NCB"); return !(e && e[0] == '2'); }();
    return on && !narrow && Cout % 256 == 0;
}
static inline bool w9_plan(int N, int Cm, int H, int W, int Cout, int KH, int stride, int pad, long ws_floats, W9Plan* p) {
    if (!w9_enabled() || KH != 3 || stride != 1 || pad != 1 || W % 32 || Cm < 64 || Cm % 64 || Cout < 48 ||
        (long)N * Cout * H * W * 4 >= (1L << 31))
        return false;
    const int narrow = Cout <= 64;                    // KG = 2: two K groups per workgroup, two slices per split
    p->split_mfma = w9s_enabled();
    p->ncb1 = p->split_mfma && w9s_ncb1(Cout, narrow);
    // pixel-tile height of the kernel that will run; the 256 x 32 variant falls back from 4-row to 2-row tiles on maps whose height is
    // not a multiple of 4 (round 6: the 10 x 32 and 20 x 64 maps of the 1024 x 320 shape used to drop to the exact-fp32 engine here)
    p->tr = !p->split_mfma ? W9_TR : (narrow ? W9S_TRN : (p->ncb1 ? ((w9s_tr1() == 4 && H % 4 == 0) ? 4 : 2) : W9S_TR));
    if (H % p->tr) return false;
    const int ntiles = N * (H / p->tr) * (W / 32), kg = narrow ? 2 : 1;
    const long out_tiles = p->ncb1 ? (long)(Cm / 32) * (Cout / 256) : (long)(Cm / 64) * jp_cdiv(Cout, narrow ? 64 : 128);
    const long per = (long)Cout * 9 * Cm;
    static const long wgs = [] { const char* e = getenv("JP_W9_WGS"); return e ? atol(e) : 256L; }();
    long sp = std::max<long>(1, std::min<long>(wgs / std::max<long>(1, out_tiles), ntiles / 2));
    sp = std::min<long>(sp, ws_floats / (per * kg));
    if (sp < 1 || ntiles < 8) return false;
    const int tps = (int)jp_cdiv(ntiles, sp);
    p->splits = jp_cdiv(ntiles, tps);
    p->tps = tps;
    p->ntiles = ntiles;
    p->narrow = narrow;
    p->slices = p->splits * kg;
    p->need = (long)p->slices * per;
    return true;
}
// W9S2 (igemm_w9s2.h): the 3x3 stride-2 zero-pad weight gradient of the ResNet stage transitions on the bf16 pipe; JP_W9S2=0 (or
// JP_W9S=0) keeps the exact-fp32 generic engine.
struct W9S2Plan { int splits, tps, ntiles, slices, kg; long need; };
static inline bool w9s2_plan(int N, int Cm, int H, int W, int Cout, int KH, int stride, int pad, int pad_mode, long ws_floats,
                             W9S2Plan* p) {
    static const bool on = [] { const char* e = getenv("JP_W9S2"); return !(e && e[0] == '0'); }();
    if (!on || !w9s_enabled() || KH != 3 || stride != 2 || pad != 1 || pad_mode == JP_PAD_REFLECT || (H & 1) || (W & 1) ||
        (W / 2) % 32 || (H / 2) % 2 || Cm < 32 || Cm % 32 || !(Cout == 128 || Cout % 256 == 0) ||
        (long)N * Cout * (H / 2) * (W / 2) * 4 >= (1L << 31) || (long)N * Cm * H * W * 4 >= (1L << 31))
        return false;
    const int kg = Cout == 128 ? 2 : 1;
    const int ntiles = N * (H / 4) * (W / 64);
    const long out_tiles = (long)(Cm / 32) * (kg == 2 ? 1 : Cout / 256);
    const long per = (long)Cout * 9 * Cm;
    long sp = std::max<long>(1, std::min<long>(256 / std::max<long>(1, out_tiles), ntiles / 4));
    sp = std::min<long>(sp, std::min<long>(ws_floats, 16L << 20) / (per * kg));     // partial slices: at most 64 MB written + folded
    if (sp < 1 || ntiles < 8) return false;
    p->tps = (int)jp_cdiv(ntiles, sp);
    p->splits = jp_cdiv(ntiles, p->tps);
    p->ntiles = ntiles;
    p->kg = kg;
    p->slices = p->splits * kg;
    p->need = (long)p->slices * per;
    return true;
}
template <int KG>
const char* w9s2_tag() { return __PRETTY_FUNCTION__; }
static void launch_w9s2(const float* dy, const float* x, float* ws, int N, int Cx, int Cm, int H, int W, int Cout,
                        const W9S2Plan& p, const JpCall& st) {
    const int dyb = (int)((long)N * Cout * (H / 2) * (W / 2) * 4), xb = (int)((long)N * Cx * H * W * 4);
    const float* gam = JP_NS == 2 ? jp_amax_of(dy, (long)N * Cout * (H / 2) * (W / 2), st) : nullptr;
    const float* xam = JP_NS == 2 ? jp_amax_of(x, (long)N * Cx * H * W, st) : nullptr;
    const double fl = JP_NPROD * 2.0 * Cout * 9.0 * Cm * (double)N * (H / 2) * (W / 2);
    if (p.kg == 2) {
        jp_prof_before(w9s2_tag<2>(), fl, st);
        hipLaunchKernelGGL((jp_wgrad_w9s2_kernel<2>), dim3(Cm / 32, 1, p.splits), dim3(512), 0, st, dy, x, ws, Cout, Cx, Cm, H, W,
                           p.ntiles, p.tps, dyb, xb, gam, xam);
    } else {
        jp_prof_before(w9s2_tag<1>(), fl, st);
        hipLaunchKernelGGL((jp_wgrad_w9s2_kernel<1>), dim3(Cm / 32, Cout / 256, p.splits), dim3(512), 0, st, dy, x, ws, Cout, Cx, Cm,
                           H, W, p.ntiles, p.tps, dyb, xb, gam, xam);
    }
    jp_prof_after(st);
}
template <int MW, int KG, bool REFLECT>
const char* w9_tag() { return __PRETTY_FUNCTION__; }
template <int TR, bool REFLECT, int KG, int NCB = 2>
const char* w9s_tag() { return __PRETTY_FUNCTION__; }
template <bool REFLECT>
static void launch_w9(const float* dy, const float* x, float* ws, int N, int Cx, int Cm, int H, int W, int Cout,
                      const W9Plan& p, const JpCall& st) {
    if (p.split_mfma) {
        // executed FLOPs: 6 bf16 MFMA products per fp32 product
        const int dyb = (int)((long)N * Cout * H * W * 4);
        const int xb = (int)std::min<long>((long)N * Cx * H * W * 4, 0x7fffffffL);       // (w9_plan: the X tensor is < 2 GiB too)
        const float* gam = JP_NS == 2 ? jp_amax_of(dy, (long)N * Cout * H * W, st) : nullptr;
        const float* xam = JP_NS == 2 ? jp_amax_of(x, (long)N * Cx * H * W, st) : nullptr;
        if (p.narrow) {
            jp_prof_before(w9s_tag<W9S_TRN, REFLECT, 2>(), JP_NPROD * 2.0 * Cout * 9.0 * Cm * (double)N * H * W, st);
            dim3 grid(Cm / 64, jp_cdiv(Cout, 64), p.splits);
            hipLaunchKernelGGL((jp_wgrad_w9s_kernel<W9S_TRN, REFLECT, 2>), grid, dim3(512), 0, st, dy, x, ws, Cout, Cx, Cm, H, W,
                               p.ntiles, p.tps, dyb, xb, gam, xam);
        } else if (p.ncb1) {
            dim3 grid(Cm / 32, Cout / 256, p.splits);
            if (p.tr == 4) {
                jp_prof_before(w9s_tag<4, REFLECT, 1, 1>(), JP_NPROD * 2.0 * Cout * 9.0 * Cm * (double)N * H * W, st);
                hipLaunchKernelGGL((jp_wgrad_w9s_kernel<4, REFLECT, 1, 1>), grid, dim3(512), 0, st, dy, x, ws, Cout, Cx, Cm, H, W,
                                   p.ntiles, p.tps, dyb, xb, gam, xam);
            } else {
                jp_prof_before(w9s_tag<W9S_TR, REFLECT, 1, 1>(), JP_NPROD * 2.0 * Cout * 9.0 * Cm * (double)N * H * W, st);
                hipLaunchKernelGGL((jp_wgrad_w9s_kernel<W9S_TR, REFLECT, 1, 1>), grid, dim3(512), 0, st, dy, x, ws, Cout, Cx, Cm, H, W,
                                   p.ntiles, p.tps, dyb, xb, gam, xam);
            }
        } else {
            jp_prof_before(w9s_tag<W9S_TR, REFLECT, 1>(), JP_NPROD * 2.0 * Cout * 9.0 * Cm * (double)N * H * W, st);
            dim3 grid(Cm / 64, jp_cdiv(Cout, 128), p.splits);
            hipLaunchKernelGGL((jp_wgrad_w9s_kernel<W9S_TR, REFLECT, 1>), grid, dim3(512), 0, st, dy, x, ws, Cout, Cx, Cm, H, W,
                               p.ntiles, p.tps, dyb, xb, gam, xam);
        }
        jp_prof_after(st);
        return;
    }
    jp_prof_before(p.narrow ? w9_tag<1, 2, REFLECT>() : w9_tag<2, 1, REFLECT>(), 2.0 * Cout * 9.0 * Cm * (double)N * H * W, st);
    const int dyb = (int)((long)N * Cout * H * W * 4);          // < 2^31 (w9_plan): dY is addressed through a buffer resource
    if (p.narrow) {
        dim3 grid(Cm / 64, jp_cdiv(Cout, 64), p.splits);
        hipLaunchKernelGGL((jp_wgrad_w9_kernel<1, 2, 2, REFLECT>), grid, dim3(768), 0, st, dy, x, ws, Cout, Cx, Cm, H, W,
                           p.ntiles, p.tps, dyb);
    } else {
        dim3 grid(Cm / 64, jp_cdiv(Cout, 128), p.splits);
        hipLaunchKernelGGL((jp_wgrad_w9_kernel<2, 2, 1, REFLECT>), grid, dim3(768), 0, st, dy, x, ws, Cout, Cx, Cm, H, W,
                           p.ntiles, p.tps, dyb);
    }
    jp_prof_after(st);
}

// ---- W7 stem wgrad (igemm_w7.h): 7x7 stride 2 pad 3, 3 or 6 input channels -> 64
struct W7Plan { int splits, tps, ntiles, slices; long need; };
static inline bool w7_plan(int N, int Cin, int H, int W, int Cout, int KH, int stride, int pad, long ws_floats, W7Plan* p) {
    if (!w9_enabled() || KH != 7 || stride != 2 || pad != 3 || (Cin != 3 && Cin != 6) || Cout != 64 || H % 2 || W % 2 ||
        (W / 2) % 32 || (H / 2) % 4 || (long)N * 64 * (H / 2) * (W / 2) * 4 >= (1L << 31) || (long)N * Cin * H * W >= (1L << 31))
        return false;
    const int ntiles = N * (H / 8) * (W / 64), kg = Cin == 3 ? 2 : 1, np = (49 * Cin + 31) / 32 * 32;
    long sp = std::max<long>(1, std::min<long>(512, ntiles / 4));
    sp = std::min<long>(sp, ws_floats / (64L * np * kg));
    if (sp < 1 || ntiles < 16) return false;
    p->tps = (int)jp_cdiv(ntiles, sp);
    p->splits = jp_cdiv(ntiles, p->tps);
    p->ntiles = ntiles;
    p->slices = p->splits * kg;
    p->need = (long)p->slices * 64 * np;
    return true;
}
template <int CIN>
const char* w7_tag() { return __PRETTY_FUNCTION__; }
template <int CIN>
static void launch_w7(const float* dy, const float* x, float* dw, float* ws, int N, int H, int W, const W7Plan& p,
                      const JpCall& st) {
    jp_prof_before(w7_tag<CIN>(), 2.0 * 64 * 49.0 * CIN * (double)N * (H / 2) * (W / 2), st);
    hipLaunchKernelGGL((jp_wgrad_w7_kernel<CIN>), dim3(p.splits), dim3(640), 0, st, dy, x, ws, H, W, p.ntiles, p.tps,
                       (int)((long)N * 64 * (H / 2) * (W / 2) * 4));
    jp_prof_after(st);
    constexpr int NP = (49 * CIN + 31) / 32 * 32;
    hipLaunchKernelGGL((w7_reduce_kernel<CIN>), dim3(64 * NP / 64), dim3(1024), 0, st, ws, dw, p.slices);
}

// ---- W1 (igemm_w9.h): 1x1 stride-1 wgrad, single full-resolution source, 128-aligned input channels
struct W1Plan { int splits, tps, ntiles; long need; };
static inline bool w1_plan(int N, int Cm, int H, int W, int Cout, int KH, int stride, int pad, long ws_floats, W1Plan* p) {
    static const bool on = [] { const char* e = getenv("JP_W1"); return !(e && e[0] == '0'); }();
    if (!on || !w9_enabled() || KH != 1 || stride != 1 || pad != 0 || W % 32 || H % 4 || Cm < 128 || Cm % 128 || Cout < 192 ||
        (long)N * Cout * H * W * 4 >= (1L << 31))
        return false;
    const int ntiles = N * (H / 4) * (W / 32);
    const long out_tiles = (long)(Cm / 128) * jp_cdiv(Cout, 256), per = (long)Cout * Cm;
    static const long wgs = [] { const char* e = getenv("JP_W1_WGS"); return e ? atol(e) : 256L; }();
    long sp = std::max<long>(1, std::min<long>(wgs / std::max<long>(1, out_tiles), ntiles / 4));
    sp = std::min<long>(sp, ws_floats / per);
    if (sp < 1 || ntiles < 16) return false;
    p->tps = (int)jp_cdiv(ntiles, sp);
    p->splits = jp_cdiv(ntiles, p->tps);
    p->ntiles = ntiles;
    p->need = (long)p->splits * per;
    return true;
}
static const char* w1_tag() { return "const char *w1_tag() [K = 1]"; }
static const char* w1s_tag() { return "const char *w1s_tag() [K = 1]"; }

Src3 make_src(const float* x0, int c0, int up0, const float* x1, int c1, int up1, const float* x2, int c2,
              int up2, int H, int W) {
    Src3 s;
    s.p0 = x0; s.p1 = x1 ? x1 : x0; s.p2 = x2 ? x2 : x0;
    s.e0 = c0; s.e1 = c0 + c1; s.e2 = c0 + c1 + c2;
    s.s0 = up0; s.s1 = up1; s.s2 = up2;
    s.H = H; s.W = W;
    return s;
}

inline int pad32(int c) { return (c + 31) / 32 * 32; }
// split-K factor for forward / dgrad launches whose tile grid cannot fill 256 CUs (pose encoder, 32x32 .. 6x20 maps)
inline int small_grid_splits(int M, long N, int Kp) {
    const int bm = M <= 64 ? 64 : (N <= 64 ? 256 : 128), bn = M <= 64 ? 256 : (N <= 64 ? 64 : 128);
    const long tiles = (long)jp_cdiv(M, bm) * jp_cdiv(N, bn);
    if (tiles >= 192) return 1;
    const int chunks = Kp / KC;
    int sp = (int)std::min<long>(jp_cdiv(512, tiles), chunks / 8);
    return std::max(1, sp);
}
inline bool seg_aligned(int c0, int c1, int c2) {   // every segment end except the last is a multiple of 32
    if (c1 == 0 && c2 == 0) return true;
    if (c0 % 32) return false;
    if (c2 != 0 && (c0 + c1) % 32) return false;
    return true;
}

}  // namespace

// conv_small.hip: direct kernels for <= 4 output channels (disparity / BEV logits heads)
int jp_conv_small_fwd(const float* x0, int c0, int up0, const float* x1, int c1, int up1, const float* x2, int c2, int up2,
                      const float* w, const float* bias, float* y, int N, int H, int W, int Cout, int act, int reflect,
                      hipStream_t st, int accumulate = 0);
long jp_conv_small_wgrad_ws_floats(int N, int Cin, int H, int W, int Cout, int single_full_res);
int jp_conv_small_wgrad(const float* x0, int c0, int up0, const float* x1, int c1, int up1, const float* x2, int c2, int up2,
                        const float* dy, float* dw, int N, int H, int W, int Cout, int reflect, hipStream_t st, float* ws,
                        long ws_floats);
// conv_c16.hip: direct kernels for the 16 / 32-channel 3x3 layers of the BEV decoder
bool jp_c16_ok(int Cin, int Cout, int KH, int stride, int pad, int pad_mode, int H, int W);
int jp_c16_fwd(const float* x, int up, const float* w, const float* bias, float* y, int N, int Cin, int Cout, int H, int W, int act,
               hipStream_t st);
int jp_c16_dgrad(const float* dy, const float* w, float* dx, int up, int N, int Cin, int Cout, int H, int W, int accumulate,
                 hipStream_t st);
int jp_c16_wgrad(const float* x, int up, const float* dy, float* dw, int N, int Cin, int Cout, int H, int W, hipStream_t st,
                 float* ws, long ws_floats);
long jp_c16_wgrad_ws_floats(int N, int Cin, int Cout, int H, int W);
int jp_up_head_fwd(const float* x, const float* w, const float* bias, float* y, int N, int C, int h, int wd, int act,
                   hipStream_t st);
int jp_up_head_dgrad(const float* dy, const float* w, float* dx, int N, int C, int h, int wd, int accumulate, hipStream_t st);
long jp_up_head_wgrad_ws_floats(int N, int C, int h, int wd);
int jp_up_head_wgrad(const float* x, const float* dy, float* dw, int N, int C, int h, int wd, hipStream_t st, float* ws,
                     long ws_floats);
// one-channel head on a single nearest-2x-upsampled source (the disparity heads): upsample-aware direct kernels
static inline bool up_head(int c0, int up0, int c1, int c2, int Cout, int KH, int stride, int pad, int pad_mode, int H, int W) {
    return Cout == 1 && c1 == 0 && c2 == 0 && up0 && c0 >= 8 && c0 <= 768 && KH == 3 && stride == 1 && pad == 1 &&
           pad_mode == JP_PAD_REFLECT && H % 2 == 0 && W % 2 == 0 && H >= 4 && W >= 4;
}
extern "C" int jp_conv2d_up_head_ok(int c0, int up0, int c1, int c2, int Cout, int KH, int stride, int pad, int pad_mode, int H,
                                    int W) {
    return up_head(c0, up0, c1, c2, Cout, KH, stride, pad, pad_mode, H, W) ? 1 : 0;
}
static inline bool small_head(int Cin, int Cout, int KH, int stride, int pad) {
    return Cout <= 4 && KH == 3 && stride == 1 && pad == 1 && (long)Cout * Cin * 9 * 4 <= 48 * 1024;
}

#define JP_KH_SWITCH(KHV, ...)                                  \
    switch (KHV) {                                              \
        case 1: { constexpr int KH_ = 1; __VA_ARGS__; } break;  \
        case 3: { constexpr int KH_ = 3; __VA_ARGS__; } break;  \
        case 7: { constexpr int KH_ = 7; __VA_ARGS__; } break;  \
        default: jp_set_last_error("conv: kernel size must be 1, 3 or 7"); return JP_EBADARG; \
    }

// floats of caller-owned scratch for the packed-weight fast path (0 = the generic path will be used).
// which: 0 forward, 1 dgrad, 2 wgrad
// does the small-map split-bf16 path (conv_p9sm.hip) take this stride-1 3x3 / 1x1 GEMM (rows x red over N x H x W pixels)?
// JP_P9SM = 2 (default): every eligible layer the regular patch kernels do not run (also the 1x1 layers with 64 / 128-row
// banks, which those leave to the generic engine: reduce 128->256 @128^2 dgrad 0.116 -> 0.066 ms, 64->256 @256^2 0.217 -> 0.187);
// 1: only what they reject for their tile shape or grid size (maps that are not multiples of the pixel tile, < 192 workgroups);
// 0: off.
static bool p9sm_wanted(int rows, int red, int N, int H, int W, int KH, int stride, int pad, JpP9smPlan* sm) {
    static const int mode = [] { const char* e = getenv("JP_P9SM"); return e ? atoi(e) : 2; }();
    if (!mode || stride != 1 || !((KH == 3 && pad == 1) || (KH == 1 && pad == 0))) return false;
    if (p9_ok(rows, red, N, H, W, KH * KH)) return false;
    if (!jp_p9sm_plan(rows, red, N, H, W, KH * KH, sm)) return false;
    const bool odd_shape = W % 32 != 0 || H % sm->tr != 0;
    return mode >= 2 || odd_shape || sm->splits > 1 ||
           (long)jp_cdiv(rows, sm->bmt) * N * jp_cdiv(H, sm->tr) * jp_cdiv(W, 32) < 192;
}

// floats of `bn_stats` scratch for a forward call with OH x OW output maps: 2 sums x Cout x (at most one partial per 2 x 32 output pixels)
extern "C" long jp_conv2d_fwd_bn_stats_floats(int N, int OH, int OW, int Cout) {
    return 2L * Cout * N * jp_cdiv(OH, 2) * jp_cdiv(OW, 32);
}
extern "C" long jp_conv2d_ws_floats(int Cin, int Cout, int KH, int which) {
    // + 256 rows of slack: the A gather of the last M tile reads (never uses) up to 255 rows past the last tap
    // forward: up to 16 weight planes per channel (parity-class path of fused-upsample segments), 3 padded segments
    if (which == 0 && Cin <= 8) return (long)(Cout + 256) * pad32(KH * KH * 8);   // row-major pack of the stem path
    if (which == 0) return Cin >= 16 ? std::max(std::max(((long)std::max(KH * KH, 16) * Cout + 256) * (pad32(Cin) + 96),
                                                         // P9US pack: <= 16 steps per 16 channels (4 classes x 4 slots) + D + slack
                                                         KH == 3 ? (long)jp_cdiv(Cout, 128) * ((long)jp_cdiv(Cin, 16) * 16 + 10) * 3072 : 0L),
                                                KH == 3 ? p9_alloc_floats(Cout, pad32(Cin)) : (KH == 1 ? p9_alloc_floats(Cout, pad32(Cin), 1) : 0L)) : 0;
    // dgrad: [tap][ci][Cp] + slack, plus 16 planes [class,slot][c][Cp] + slack for jp_conv2d_dgrad_src3's upsampled
    // segment, plus the fragment-order pack of the P9 main pass behind them
    if (which == 1) return Cout >= 16 ? dgrad_tap_floats(Cin, Cout, KH) + (KH == 3 ? p9_alloc_floats(Cin, pad32(Cout)) + p9sd_floats(Cin, Cout) : (KH == 1 ? p9_alloc_floats(Cin, pad32(Cout), 1) : 0L)) : 0;
    return 0;
}

extern "C" int jp_conv2d_fwd_src3(const float* x0, int c0, int up0, const float* x1, int c1, int up1,
                                  const float* x2, int c2, int up2, const float* w, const float* bias, float* y,
                                  int N, int H, int W, int Cout, int KH, int stride, int pad, int pad_mode, int act,
                                  float* ws, int ws_state, float* split_ws, const float* amax_x0, const float* amax_x1,
                                  const float* amax_x2, float* amax_y, int* amax_y_done, float* amax_ws, float* bn_stats,
                                  int* bn_stats_parts, void* stream) {
    // ws_state: 0 = pack the weights into ws now; 1 = ws already holds this layer's pack (refreshed by jp_pack_replay)
    // amax_*: operand magnitudes of the fp16 split kernels, see the header (all may be NULL when amax_ws is given)
    // bn_stats: optional scratch of jp_conv2d_fwd_bn_stats_floats floats for a convolution that feeds a train-mode BatchNorm (no bias, no
    // activation): the 4-wave 3x3 patch kernels leave per-channel partial sums of y and y^2 there and *bn_stats_parts (host int) = the
    // partials per channel, to be handed to jp_bn_train_fwd instead of its own pass over y; 0 = the kernel that ran does not (BatchNorm
    // then reads y itself, as before)
    if (amax_y_done) *amax_y_done = 0;
    if (bn_stats_parts) *bn_stats_parts = 0;
    JP_CHECK_ARG(x0 && w && y, "conv2d_fwd: null pointer");
    JP_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cout > 0 && c0 > 0 && stride >= 1, "conv2d_fwd: bad dims");
    JP_CHECK_ARG(!(pad_mode == JP_PAD_REFLECT && (pad >= H || pad >= W)), "conv2d_fwd: reflect pad >= size");
    const int Cin = c0 + c1 + c2;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KH) / stride + 1;
    const long npix = (long)N * OH * OW;
    JP_CHECK_ARG(npix < (1L << 31) && (long)N * Cin * H * W < (1L << 31) * 2, "conv2d_fwd: tensor too large");
    // (a multi-source call always needs the scratch: the kernel reads ONE magnitude, the largest of the sources', folded into amax_ws)
    JP_CHECK_ARG(JP_NS != 2 || amax_ws || (amax_x0 && !c1 && !c2),
                 "conv2d_fwd"": every operand magnitude (amax_*) or amax_ws (jp_conv2d_amax_ws_floats floats of scratch) must be given");
    JpAmaxCtx ax;
    ax.know(x0, amax_x0);
    ax.know(x1, amax_x1);
    ax.know(x2, amax_x2);
    ax.ws = amax_ws;
    ax.out = reinterpret_cast<unsigned*>(amax_y);
    if (bn_stats && bn_stats_parts && !bias && act == JP_ACT_NONE) ax.stats = bn_stats;
    const JpAmaxDone done_flag{amax_y_done, &ax, bn_stats_parts};
    const JpCall st((hipStream_t)stream, &ax);
    FwdEpi e{y, bias, Cout, OH * OW, act};
    if (up_head(c0, up0, c1, c2, Cout, KH, stride, pad, pad_mode, H, W)) {
        jp_up_head_fwd(x0, w, bias, y, N, c0, H / 2, W / 2, act, st);
        JP_LAUNCH_CHECK();
    }
    if (small_head(Cin, Cout, KH, stride, pad)) {
        jp_conv_small_fwd(x0, c0, up0, x1, c1, up1, x2, c2, up2, w, bias, y, N, H, W, Cout, act,
                          pad_mode == JP_PAD_REFLECT, st);
        JP_LAUNCH_CHECK();
    }
    if (c1 == 0 && c2 == 0 && jp_c16_ok(c0, Cout, KH, stride, pad, pad_mode, H, W)) {
        jp_c16_fwd(x0, up0, w, bias, y, N, c0, Cout, H, W, act, st);
        JP_LAUNCH_CHECK();
    }
    const Src3 src = make_src(x0, c0, up0, x1, c1, up1, x2, c2, up2, H, W);
    if (ws && Cin <= 8 && Cout <= 64 && c1 == 0 && c2 == 0 && !up0 && pad_mode != JP_PAD_REFLECT && (KH == 7 || KH == 3) &&
        npix / 256 >= 192 && (long)2 * Cin * H * W * 4 < (1L << 32)) {
        if (KH == 7 && stride == 2 && pad == 3 && (Cin == 3 || Cin == 6) && p9s_enabled() && p9s2_enabled() && JP_ENV_ON("JP_P7S") && H == 2 * OH && W == 2 * OW &&
            OH % 8 == 0 && OW % 32 == 0 && (long)Cin * H * W * 4 < (1L << 31)) {
            // P7S stem kernel (igemm_p7s.h): split-bf16 products, the whole K of a tile staged once
            const long tot = (4L * Cin + 1) * (JP_NS * 512) + JP_PACK_HDR;
            if (!ws_state) do_pack(PACK_SPLIT7, w, ws, tot, Cout, Cin, 0, 0, 0, 0, st);
            const int ntiles = N * (OH / 8) * (OW / 32);
            const unsigned* wq = reinterpret_cast<const unsigned*>(ws);
            const float* xam = JP_NS == 2 ? jp_amax_of(x0, (long)N * Cin * H * W, st) : nullptr;
            jp_prof_before(Cin == 3 ? p7s_tag<3, FwdEpi>() : p7s_tag<6, FwdEpi>(), JP_NPROD * 2.0 * 64 * (double)npix * 64.0 * Cin, st);
            if (Cin == 3) hipLaunchKernelGGL((jp_igemm_p7s_kernel<3, FwdEpi>), dim3(ntiles), dim3(256), 0, st, wq, x0, e, Cout, OH, OW, ntiles, xam);
            else hipLaunchKernelGGL((jp_igemm_p7s_kernel<6, FwdEpi>), dim3(ntiles), dim3(256), 0, st, wq, x0, e, Cout, OH, OW, ntiles, xam);
            jp_prof_after(st);
            JP_LAUNCH_CHECK();
        }
        // stem convs: whole taps per K chunk
        const int CP = Cin <= 4 ? 4 : 8;
        const int Kp = jp_cdiv(KH * KH * CP, KC) * KC;
        if (!ws_state) do_pack(PACK_ROWMAJOR, w, ws, (long)Cout * Kp, Cout, Cin, KH * KH, CP, Kp, 0, st);
        PackARow a{ws, Kp};
#define JP_BC(KHv, CPv)                                                          \
    {                                                                            \
        FwdBC<KHv, CPv> b{x0, Cin, H, W, (int)npix, OH, OW, stride, pad};        \
        launch<false, 1, 4>(a, b, e, Cout, (int)npix, Kp, 1, Kp, st);            \
    }
        if (KH == 7 && CP == 4) JP_BC(7, 4)
        else if (KH == 7) JP_BC(7, 8)
        else if (CP == 4) JP_BC(3, 4)
        else JP_BC(3, 8)
#undef JP_BC
        JP_LAUNCH_CHECK();
    }
    // P9U patch kernel for the iconv layers: cat(skip (full res), up2x(x), disparity channel) -> Cout, reflection pad
    if (ws && p9_enabled() && KH == 3 && stride == 1 && pad == 1 && pad_mode == JP_PAD_REFLECT && c0 >= 32 && c0 % 32 == 0 &&
        !up0 && c1 >= 32 && c1 % 32 == 0 && up1 && c2 <= 8 && !(c2 && up2) && Cout % 128 == 0 && H % 4 == 0 && W % 64 == 0 &&
        (long)(Cout / 128) * N * (H / 4) * (W / 64) >= 128 && p9u_enabled()) {
        const int MT = Cout / 128;
        if (p9us_enabled()) {
            // P9US2: the same tiles on the bf16 matrix pipe (three-way split products, igemm_p9us2.h)
            constexpr long SF = JP_NS * 1024;         // floats per weight step
            const long fS = (long)MT * (c0 / 16) * 9 * SF, fU = 4L * MT * (c1 / 16) * 4 * SF, fD = c2 ? (long)MT * 9 * SF : 0;
            constexpr int HD = JP_PACK_HDR;           // one scale header in front of the bank (JP_NS == 2); every segment knows how far back
            if (!ws_state) {
                do_pack(PACK_SPLITSEG, w, ws + HD, fS, Cout, Cin, 0, c0, 0, HD, st);
                do_pack(PACK_SPLITSEG, w, ws + HD + fS, fU, Cout, Cin, c0, c1, 1, (int)(HD + fS), st);
                if (c2) do_pack(PACK_SPLITSEG, w, ws + HD + fS + fU, fD, Cout, Cin, c0 + c1, c2, 0, (int)(HD + fS + fU), st);
                // (the kernel's last weight prefetch reads one step past the streams: inside the scratch, never used)
            }
            // P9US2 (igemm_p9us2.h, round 5): every operand request inside an MFMA pair's shadow, the next patch staged by the half of
            // the workgroup that is off the pipe.  (The round-3 stream, its 8-row tiles and its double buffer are gone: logs in
            // profiles/r04_p9us_*.log, r05_p9us2_*.log.)
            const float* xam = JP_NS == 2 ? jp_amax_of3(x0, (long)N * c0 * H * W, x1, (long)N * c1 * (H / 2) * (W / 2), x2,
                                                        c2 ? (long)N * c2 * H * W : 0L, st) : nullptr;
            e.amax = jp_take_amax_out(st);
            jp_prof_before(p9us2_tag<FwdEpi>(), JP_NPROD * 2.0 * Cout * (double)npix * (9.0 * c0 + 4.0 * c1 + 9.0 * (c2 ? 16 : 0)), st);
            hipLaunchKernelGGL((jp_igemm_p9us2_kernel<FwdEpi>), dim3(N * (H / 4) * (W / 64), MT, 1), dim3(512), 0, st,
                               reinterpret_cast<const unsigned*>(ws), x0, x1, x2, e, Cout, c0, c1, c2, H, W, xam);
            jp_prof_after(st);
            JP_LAUNCH_CHECK();
        }
        const long fS = (long)MT * (c0 / 32) * 36 * 1024, fU = 4L * MT * (c1 / 32) * 16 * 1024, fD = (long)MT * 9 * 1024;
        if (!ws_state) {
            do_pack(PACK_FRAGSEG, w, ws, fS, Cout, Cin, 0, c0, 16, 0, st);
            do_pack(PACK_FRAGSEG, w, ws + fS, fU, Cout, Cin, c0, c1, 16, 1, st);
            if (c2) do_pack(PACK_FRAGSEG, w, ws + fS + fU, fD, Cout, Cin, c0 + c1, c2, 4, 0, st);
        }
        jp_prof_before(p9u_tag<FwdEpi>(), 2.0 * Cout * (double)npix * (9.0 * c0 + 4.0 * c1 + 9.0 * c2), st);
        dim3 grid(N * (H / 4) * (W / 64), MT, 1);
        hipLaunchKernelGGL((jp_igemm_p9u_kernel<FwdEpi>), grid, dim3(512), 0, st, ws, x0, x1, x2, e, Cout, c0, c1, c2, H, W);
        jp_prof_after(st);
        JP_LAUNCH_CHECK();
    }
    {   // upsample-aware parity-class path (iconv layers)
        const long Ncl = (long)N * (H / 2) * (W / 2);
        const bool any_up = (c0 && up0) || (c1 && up1) || (c2 && up2);
        if (ws && any_up && KH == 3 && stride == 1 && pad == 1 && pad_mode == JP_PAD_REFLECT && H % 2 == 0 && W % 2 == 0 &&
            Ncl % 256 == 0 && seg_aligned(c0, c1, c2) && Cin >= 32 && (long)jp_cdiv(Cout, 128) * (npix / 128) >= 192) {
            const int cs[3] = {c0, c1, c2}, us[3] = {up0, up1, up2};
            int T[3], P[3], O[3], nch[3];
            long off = 0;
            for (int i = 0; i < 3; ++i) {
                T[i] = us[i] ? 4 : 9;
                P[i] = pad32(cs[i]);
                nch[i] = P[i] / 32;
                O[i] = (int)off;
                if (cs[i]) {
                    const long tot = (long)(us[i] ? 16 : 9) * Cout * P[i];
                    if (!ws_state) do_pack(PACK_SEG, w, ws + off, tot, Cout, Cin, (i == 0 ? 0 : (i == 1 ? c0 : c0 + c1)), cs[i], P[i], us[i], st);
                    off += tot;
                }
            }
            ParSeg ps{nch[0] * T[0], nch[0] * T[0] + nch[1] * T[1], T[0], T[1], T[2], O[0], O[1], O[2], P[0], P[1], P[2]};
            const int Q = ps.q1 + nch[2] * T[2];
            PackAP a{ws, ps, Cout, (int)Ncl};
            FwdBP b{src, ps, (int)Ncl, H / 2, W / 2};
            FwdEpiP ep{y, bias, Cout, (int)Ncl, H / 2, W / 2, act};
            launch_auto(a, b, ep, Cout, (int)npix, Q * KC, 1, Q * KC, st);
            JP_LAUNCH_CHECK();
        }
    }
    if (ws && Cin >= 16 && seg_aligned(c0, c1, c2)) {   // tap-major fast path (16-channel inputs: half-empty K chunks)
        const int Cp = pad32(Cin), Kp = KH * KH * Cp;
        const bool use_p9 = KH == 3 && stride == 1 && pad == 1 && !((c0 && up0) || (c1 && up1) || (c2 && up2)) && c1 == 0 &&
                            c2 == 0 && p9_ok(Cout, Cin, N, H, W);
        const bool use_p1 = KH == 1 && stride == 1 && pad == 0 && !((c0 && up0) || (c1 && up1) || (c2 && up2)) && c1 == 0 &&
                            c2 == 0 && p9_ok(Cout, Cin, N, H, W, 1);
        if (KH == 1 && stride == 2 && pad == 0 && c1 == 0 && c2 == 0 && !up0 && H % 2 == 0 && W % 2 == 0 &&
            p1s2_ok(Cout, Cin, N, OH, OW)) {
            if (!ws_state) pack_p9(w, ws, Cout, Cin, 0, p9_bmt(Cout, 1, p9_ptiles(N, OH, OW)), 1, st);
            launch_p1s2(ws, x0, e, Cout, Cin, N, OH, OW, st);
            JP_LAUNCH_CHECK();
        }
        if (use_p1) {      // 1x1: weights stream in fragment order, two channel chunks of the pixel tile staged per barrier pair
            if (!ws_state) pack_p9(w, ws, Cout, Cin, 0, p9_bmt(Cout, 1, p9_ptiles(N, H, W)), 1, st);
            launch_p1(ws, x0, e, Cout, Cin, N, H, W, st);
            JP_LAUNCH_CHECK();
        }
        if (KH == 3 && stride == 2 && pad == 1 && pad_mode != JP_PAD_REFLECT && c1 == 0 && c2 == 0 && !up0 && p9s_enabled() &&
            p9s2_enabled() && JP_ENV_ON("JP_P9S2F") && H == 2 * OH && W == 2 * OW && OH % 4 == 0 && OW % 32 == 0 && Cin % 16 == 0 && Cin >= 32 && Cout > 64 &&
            (long)N * Cin * H * W * 4 < (1L << 31) && (long)jp_cdiv(Cout, 128) * N * (OH / 4) * (OW / 32) >= 192) {
            // 3x3 stride 2: P9S2F patch kernel (igemm_p9s2f.h) on the P9S forward pack (instead of the tap-major one)
            if (!ws_state) pack_p9(w, ws, Cout, Cin, 0, 128, 9, st);
            const float* xam = JP_NS == 2 ? jp_amax_of(x0, (long)N * Cin * H * W, st) : nullptr;
            jp_prof_before(p9s2f_tag<FwdEpi>(), JP_NPROD * 2.0 * Cout * (double)npix * 9.0 * Cin, st);
            hipLaunchKernelGGL((jp_igemm_p9s2f_kernel<FwdEpi>), dim3(N * (OH / 4) * (OW / 32), jp_cdiv(Cout, 128), 1), dim3(256), 0, st,
                               reinterpret_cast<const unsigned*>(ws), x0, e, Cout, Cin, Cin / 16, OH, OW, xam);
            jp_prof_after(st);
            JP_LAUNCH_CHECK();
        }
        if (use_p9) {
            // P9 patch kernel: the input patch of a 4x32 pixel tile is staged once per channel chunk for all 9 taps,
            // weights stream from L2 in MFMA fragment order (igemm_p9.h); its pack takes the place of the tap-major one
            if (!ws_state) pack_p9(w, ws, Cout, Cin, 0, p9_bmt(Cout, 9, p9_ptiles(N, H, W)), 9, st);
            if (pad_mode == JP_PAD_REFLECT) launch_p9<true, false>(ws, x0, e, Cout, Cin, N, H, W, st);
            else launch_p9<false, false>(ws, x0, e, Cout, Cin, N, H, W, st);
            JP_LAUNCH_CHECK();
        }
        {   // small maps (partial tiles, split-K over the channel stages) on the split-bf16 patch kernel (conv_p9sm.hip)
            JpP9smPlan sm;
            if (c1 == 0 && c2 == 0 && !up0 && p9sm_wanted(Cout, Cin, N, H, W, KH, stride, pad, &sm) && (sm.splits == 1 || split_ws)) {
                if (!ws_state) pack_p9(w, ws, Cout, Cin, 0, sm.bmt, KH * KH, st);
                jp_p9sm_launch(sm, ws, x0, y, bias, act, 0, split_ws, Cout, Cin, N, H, W, KH * KH, pad_mode == JP_PAD_REFLECT, 0, st);
                if (sm.splits > 1) {
                    const long total = npix * Cout;
                    hipLaunchKernelGGL(slice_reduce_nchw_kernel, dim3((int)std::min<long>((total + 255) / 256, 2048)), dim3(256), 0,
                                       st, split_ws, y, bias, Cout, (int)npix, OH * OW, sm.splits, act, 0);
                }
                JP_LAUNCH_CHECK();
            }
        }
        if (!ws_state) pack_weights(w, ws, Cout, Cin, KH * KH, Cp, 0, st);
        PackA a{ws, Cout, Kp, Cp, KH * KH};
        const int sp = small_grid_splits(Cout, npix, Kp);
        if (sp > 1) {
            const int kps = jp_cdiv(jp_cdiv(Kp, sp), KC) * KC;
            if (split_ws) {   // per-slice partial tiles + fixed-order reduction: bit-reproducible forward
                WgradEpiWS es{split_ws, Cout, (int)npix};
                JP_KH_SWITCH(KH, {
                    if (pad_mode == JP_PAD_REFLECT) {
                        FwdBT<KH_, true> b{src, Cp, (int)npix, OH, OW, stride, pad};
                        launch_auto(a, b, es, Cout, (int)npix, Kp, jp_cdiv(Kp, kps), kps, st);
                    } else {
                        FwdBT<KH_, false> b{src, Cp, (int)npix, OH, OW, stride, pad};
                        launch_auto(a, b, es, Cout, (int)npix, Kp, jp_cdiv(Kp, kps), kps, st);
                    }
                });
                const long total = npix * Cout;
                hipLaunchKernelGGL(slice_reduce_nchw_kernel, dim3((int)std::min<long>((total + 255) / 256, 2048)), dim3(256), 0,
                                   st, split_ws, y, bias, Cout, (int)npix, OH * OW, jp_cdiv(Kp, kps), act, 0);
                JP_LAUNCH_CHECK();
            }
            JP_HIP(hipMemsetAsync(y, 0, sizeof(float) * (size_t)npix * Cout, st));
            AtomicEpi ea{y, Cout, OH * OW};
            JP_KH_SWITCH(KH, {
                if (pad_mode == JP_PAD_REFLECT) {
                    FwdBT<KH_, true> b{src, Cp, (int)npix, OH, OW, stride, pad};
                    launch_auto(a, b, ea, Cout, (int)npix, Kp, jp_cdiv(Kp, kps), kps, st);
                } else {
                    FwdBT<KH_, false> b{src, Cp, (int)npix, OH, OW, stride, pad};
                    launch_auto(a, b, ea, Cout, (int)npix, Kp, jp_cdiv(Kp, kps), kps, st);
                }
            });
            if (bias || act != JP_ACT_NONE) {
                const long total = npix * Cout;
                hipLaunchKernelGGL(bias_act_nchw_kernel, dim3((int)std::min<long>((total + 255) / 256, 2048)), dim3(256), 0, st,
                                   y, bias, total, Cout, OH * OW, act);
            }
            JP_LAUNCH_CHECK();
        }
        const bool no_up = !((c0 && up0) || (c1 && up1) || (c2 && up2));
        const int bn3 = Cout <= 64 ? 256 : 128;
        if (KH == 3 && stride == 1 && pad == 1 && no_up && Cin >= 32 && W % bn3 == 0 && npix > 64) {
            // row-tile kernel: the three dx taps share one staged input row segment
            if (pad_mode == JP_PAD_REFLECT) {
                if (Cout <= 64) { FwdBR3<true, false, 256> b{src, H, W}; launch_r3<1, 4>(a, b, e, Cout, (int)npix, Kp, st); }
                else { FwdBR3<true, false, 128> b{src, H, W}; launch_r3<2, 2>(a, b, e, Cout, (int)npix, Kp, st); }
            } else {
                if (Cout <= 64) { FwdBR3<false, false, 256> b{src, H, W}; launch_r3<1, 4>(a, b, e, Cout, (int)npix, Kp, st); }
                else { FwdBR3<false, false, 128> b{src, H, W}; launch_r3<2, 2>(a, b, e, Cout, (int)npix, Kp, st); }
            }
            JP_LAUNCH_CHECK();
        }
        JP_KH_SWITCH(KH, {
            if (pad_mode == JP_PAD_REFLECT) {
                FwdBT<KH_, true> b{src, Cp, (int)npix, OH, OW, stride, pad};
                launch_auto(a, b, e, Cout, (int)npix, Kp, 1, Kp, st);
            } else {
                FwdBT<KH_, false> b{src, Cp, (int)npix, OH, OW, stride, pad};
                launch_auto(a, b, e, Cout, (int)npix, Kp, 1, Kp, st);
            }
        });
    } else {
        const int K = Cin * KH * KH;
        FwdA a{w, Cout, K};
        JP_KH_SWITCH(KH, {
            FwdB<KH_> b{src, K, (int)npix, OH, OW, stride, pad, pad_mode == JP_PAD_REFLECT};
            launch_auto(a, b, e, Cout, (int)npix, K, 1, jp_cdiv(K, KC) * KC, st);
        });
    }
    JP_LAUNCH_CHECK();
}

extern "C" int jp_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H,
                             int W, int Cout, int KH, int stride, int pad, int pad_mode, int act, float* ws,
                             int ws_state, float* split_ws, const float* amax_x, float* amax_y, int* amax_y_done, float* amax_ws,
                             float* bn_stats, int* bn_stats_parts, void* stream) {
    return jp_conv2d_fwd_src3(x, Cin, 0, nullptr, 0, 0, nullptr, 0, 0, w, bias, y, N, H, W, Cout, KH, stride, pad,
                              pad_mode, act, ws, ws_state, split_ws, amax_x, nullptr, nullptr, amax_y, amax_y_done, amax_ws, bn_stats,
                              bn_stats_parts, stream);
}

// same for jp_conv2d_dgrad's `split_ws`
extern "C" long jp_conv2d_dgrad_split_floats(int N, int Cin, int H, int W, int Cout, int KH, int stride, int pad) {
    if (Cout < 16) return 0;
    const long npix = (long)N * H * W;
    const int Kp = KH * KH * pad32(Cout);
    // 3x3 stride 2: the parity-class forms need no split; maps they do not take (N * H/2 * W/2 not a multiple of 256: the 8 x 8 maps of
    // the 256^2 test shapes) run the generic loader, whose small-grid K slices must not meet in atomics (round 6: that was the first
    // run-dependent sum of the backward at those shapes, tools/debug/first_divergence.py)
    if (KH == 3 && stride == 2 && pad == 1 && H % 2 == 0 && W % 2 == 0 && ((long)N * (H / 2) * (W / 2)) % 256 == 0) return 0;
    // reflection layers (the pad mode is not an argument here: every 3x3 stride-1 pad-1 layer gets it): <= 4 slices of the
    // border pass, part[slice][Cin][N * (2H + 2W)]
    long border = (KH == 3 && stride == 1 && pad == 1) ? 4L * Cin * N * (2L * H + 2L * W) : 0;
    JpP9smPlan sm;
    if (p9sm_wanted(Cin, Cout, N, H, W, KH, stride, pad, &sm)) border = std::max(border, sm.part_floats);
    const int sp = small_grid_splits(Cin, npix, Kp);
    if (sp <= 1) return border;
    const int kps = jp_cdiv(jp_cdiv(Kp, sp), KC) * KC;
    return std::max(border, (long)jp_cdiv(Kp, kps) * Cin * npix);
}

// floats of optional caller scratch (`split_ws`) for the split-K forward of layers whose tile grid cannot fill the
// chip: with it the K slices are reduced in a fixed order (bit-reproducible); without it they meet in atomics.
extern "C" long jp_conv2d_fwd_split_floats(int N, int Cin, int H, int W, int Cout, int KH, int stride, int pad) {
    if (Cin < 16) return 0;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KH) / stride + 1;
    const long npix = (long)N * OH * OW;
    const int Kp = KH * KH * pad32(Cin);
    JpP9smPlan sm;
    const long smf = p9sm_wanted(Cout, Cin, N, H, W, KH, stride, pad, &sm) ? sm.part_floats : 0;
    const int sp = small_grid_splits(Cout, npix, Kp);
    if (sp <= 1) return smf;
    const int kps = jp_cdiv(jp_cdiv(Kp, sp), KC) * KC;
    return std::max(smf, (long)jp_cdiv(Kp, kps) * Cout * npix);
}

// reflection border pass of a dgrad (rows of `a` = the C input channels of this call): through caller scratch when there is some
// (K slices stored coalesced, one fixed-order fold into dx), else the read-modify-write epilogue
static void border_pass(const PackA& a, const float* dy, float* dx, int C, int Cp, int Kp, int Cout, int N, int H, int W,
                        float* split_ws, const JpCall& st) {
    const int Nb = N * (2 * W + 2 * H);
    DgradBorderB<3> bb{dy, Cp, Nb, H, W, Cout};
    const long btiles = (long)jp_cdiv(C, C <= 64 ? 64 : 128) * jp_cdiv(Nb, C <= 64 ? 256 : 128);
    static const bool via_ws = [] { const char* e_ = getenv("JP_BORDER_WS"); return !(e_ && e_[0] == '0'); }();
    const int chunks = Kp / KC;
    const int wsl = (int)std::max<long>(1, std::min<long>(std::min<long>(4, chunks / 9), jp_cdiv(512, btiles)));
    if (via_ws && split_ws && wsl > 1) {      // (also the one-row launches -- the disparity channel: its slices met in float atomics until round 6)
        const int wkps = jp_cdiv(jp_cdiv(Kp, wsl), KC) * KC, nsl = jp_cdiv(Kp, wkps);
        WgradEpiWS es{split_ws, C, Nb};
        launch_auto(a, bb, es, C, Nb, Kp, nsl, wkps, st);
        const long total = (long)C * Nb;
        hipLaunchKernelGGL(border_add_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, st, split_ws, dx, C, Nb, H,
                           W, nsl, (st.ax && st.ax->out_taken) ? st.ax->out : nullptr);
        return;
    }
    if (st.ax) st.ax->out_taken = false;       // (the read-modify-write epilogue below does not report: the caller reduces dx itself)
    DgradBorderEpi be{dx, C, H, W, Nb, 0};
    const int bsp = border_splits(btiles, Kp / KC);
    const int bkps = jp_cdiv(jp_cdiv(Kp, bsp), KC) * KC;
    be.split = jp_cdiv(Kp, bkps) > 1;
    launch_auto(a, bb, be, C, Nb, Kp, jp_cdiv(Kp, bkps), bkps, st);
}

extern "C" int jp_conv2d_dgrad(const float* dy, const float* w, float* dx, int N, int Cin, int H, int W, int Cout,
                               int KH, int stride, int pad, int pad_mode, int accumulate, float* ws, int ws_state,
                               float* split_ws, const float* amax_dy, float* amax_dx, int* amax_dx_done, float* amax_ws,
                               void* stream) {
    if (amax_dx_done) *amax_dx_done = 0;
    JP_CHECK_ARG(dy && w && dx, "conv2d_dgrad: null pointer");
    JP_CHECK_ARG(!(pad_mode == JP_PAD_REFLECT && !(KH == 3 && stride == 1 && pad == 1 && H >= 2 && W >= 2)),
                 "conv2d_dgrad: reflect mode supports 3x3 stride 1 pad 1 only");
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KH) / stride + 1;
    const long npix = (long)N * H * W;
    JP_CHECK_ARG(npix < (1L << 31), "conv2d_dgrad: tensor too large");
    JP_CHECK_ARG(JP_NS != 2 || amax_ws || amax_dy, "conv2d_dgrad"": every operand magnitude (amax_*) or amax_ws (jp_conv2d_amax_ws_floats floats of scratch) must be given");
    JpAmaxCtx ax;
    ax.know(dy, amax_dy);
    ax.ws = amax_ws;
    // amax_dx: folded in by the patch kernels' epilogue (+ the border fold of reflection layers); a layer with a short row tail is
    // finished by a second launch that does not report, so the slot is not offered there
    if (!(Cin > 128 && Cin % 128 <= 16 && Cin % 128 != 0)) ax.out = reinterpret_cast<unsigned*>(amax_dx);
    const JpAmaxDone done_flag{amax_dx_done, &ax};
    const JpCall st((hipStream_t)stream, &ax);
    if (jp_c16_ok(Cin, Cout, KH, stride, pad, pad_mode, H, W)) {
        jp_c16_dgrad(dy, w, dx, 0, N, Cin, Cout, H, W, accumulate, st);
        JP_LAUNCH_CHECK();
    }
    DgradEpi e{dx, Cin, H * W, accumulate};
    if (ws && Cout >= 16) {
        const int Cp = pad32(Cout), Kp = KH * KH * Cp;
        const int sp = small_grid_splits(Cin, npix, Kp);
        // zero-padded 3x3 stride-1 layers that the P9 main pass covers completely (no row tail, no reflection border pass)
        // never read the tap-major pack: it is neither built nor replayed for them
        const bool p9_only = KH == 3 && stride == 1 && pad == 1 && pad_mode != JP_PAD_REFLECT && sp <= 1 &&
                             !(Cin > 128 && Cin % 128 <= 16 && Cin % 128 != 0) && p9_ok(Cin, Cout, N, H, W);
        const long Nc = (long)N * (H / 2) * (W / 2);
        // 3x3 stride-2 layers: class-uniform split-bf16 kernel on the half-resolution grid (its own fragment-order pack only)
        const bool s2_form = KH == 3 && stride == 2 && pad == 1 && pad_mode != JP_PAD_REFLECT && H % 2 == 0 && W % 2 == 0 &&
                             2 * OH == H && 2 * OW == W;
        if (s2_form && p9s_enabled() && p9s2_enabled() && JP_ENV_ON("JP_P9S2D") && Cin >= 32 && Cout % 16 == 0 && OW % 32 == 0 &&
            OH % (Cin <= 64 ? 8 : 4) == 0 && (long)N * Cout * OH * OW * 4 < (1L << 31) &&
            (long)jp_cdiv(Cin, Cin <= 64 ? 64 : 128) * N * (OH / (Cin <= 64 ? 8 : 4)) * (OW / 32) * 4 >= 192) {
            float* wfr = ws + dgrad_tap_floats(Cin, Cout, KH);
            const int bmt = Cin <= 64 ? 64 : 128;
            if (!ws_state) pack_p9(w, wfr, Cout, Cin, 1, bmt, 9, st);
            const unsigned* wq = reinterpret_cast<const unsigned*>(wfr);
            const int NST = Cout / 16;
            const float* xam = JP_NS == 2 ? jp_amax_of(dy, (long)N * Cout * OH * OW, st) : nullptr;
            jp_prof_before(bmt == 64 ? p9s2d_tag<1, 4, DgradEpi>() : p9s2d_tag<2, 2, DgradEpi>(),
                           JP_NPROD * 2.0 * Cin * (double)N * OH * OW * 9.0 * Cout, st);
            if (bmt == 64)
                hipLaunchKernelGGL((jp_igemm_p9s2d_kernel<1, 4, 2, DgradEpi>), dim3(4 * N * (OH / 8) * (OW / 32), 1, 1), dim3(256), 0, st,
                                   wq, dy, e, Cin, Cout, NST, OH, OW, xam);
            else
                hipLaunchKernelGGL((jp_igemm_p9s2d_kernel<2, 2, 2, DgradEpi>), dim3(4 * N * (OH / 4) * (OW / 32), jp_cdiv(Cin, 128), 1),
                                   dim3(256), 0, st, wq, dy, e, Cin, Cout, NST, OH, OW, xam);
            jp_prof_after(st);
            JP_LAUNCH_CHECK();
        }
        if (KH == 1 && stride == 2 && pad == 0 && H % 2 == 0 && W % 2 == 0 && 2 * OH == H && 2 * OW == W && p1s2_ok(Cin, Cout, N, OH, OW)) {
            // 1x1 stride 2: the stride-1 1x1 kernel over the half-resolution dY, scattering epilogue
            float* wfr = ws + dgrad_tap_floats(Cin, Cout, KH);
            const int bmt = p9_bmt(Cin, 1, p9_ptiles(N, OH, OW));
            if (!ws_state) pack_p9(w, wfr, Cout, Cin, 1, bmt, 1, st);
            DgradS2PointEpi ep{dx, Cin, OH * OW, OW, accumulate};
            launch_p9s<false, false, DgradS2PointEpi, 1>(wfr, dy, ep, Cin, Cout, N, OH, OW, st, 0, bmt);
            JP_LAUNCH_CHECK();
        }
        if (!ws_state && !p9_only) pack_weights(w, ws, Cout, Cin, KH * KH, Cp, 1, st);
        PackA a{ws, Cin, Kp, Cp, KH * KH};
        if (KH == 3 && stride == 2 && pad == 1 && pad_mode != JP_PAD_REFLECT && H % 2 == 0 && W % 2 == 0 && Nc % 256 == 0 &&
            2 * OH == H && 2 * OW == W) {
            // parity-class form: 4 tap slots instead of 9 taps per input pixel
            PackAS2 a2{ws, Cin, Cp, (int)Nc};
            DgradS2B b2{dy, Cp, (int)Nc, H / 2, W / 2, Cout, OH, OW};
            DgradS2Epi e2{dx, Cin, (int)Nc, H / 2, W / 2, accumulate};
            launch_auto(a2, b2, e2, Cin, (int)npix, 4 * Cp, 1, 4 * Cp, st);
            JP_LAUNCH_CHECK();
        }
        JpP9smPlan sm;
        const int sm_tail = (Cin > 128 && Cin % 128 <= 16) ? Cin % 128 : 0;
        if (sm_tail == 0 && !(sp <= 1 && p9_only) && p9sm_wanted(Cin, Cout, N, H, W, KH, stride, pad, &sm) && (sm.splits == 1 || split_ws)) {
            // small maps: main pass on the split-bf16 patch kernel (taps mirrored, zero fill), then the reflection fold as usual
            float* wfr = ws + dgrad_tap_floats(Cin, Cout, KH);
            if (!ws_state) pack_p9(w, wfr, Cout, Cin, 1, sm.bmt, KH * KH, st);
            jp_p9sm_launch(sm, wfr, dy, dx, nullptr, JP_ACT_NONE, accumulate, split_ws, Cin, Cout, N, H, W, KH * KH, 0, 1, st);
            if (sm.splits > 1) {
                const long total = npix * Cin;
                hipLaunchKernelGGL(slice_reduce_nchw_kernel, dim3((int)std::min<long>((total + 255) / 256, 2048)), dim3(256), 0, st,
                                   split_ws, dx, (const float*)nullptr, Cin, (int)npix, H * W, sm.splits, JP_ACT_NONE, accumulate);
            }
        } else
        if (sp > 1 && split_ws) {   // small grids: K slices to scratch, fixed-order reduction (no memset, no atomics)
            const int kps = jp_cdiv(jp_cdiv(Kp, sp), KC) * KC;
            WgradEpiWS es{split_ws, Cin, (int)npix};
            JP_KH_SWITCH(KH, {
                DgradBT<KH_> b{dy, Cp, (int)npix, H, W, Cout, OH, OW, stride, pad, pad_mode == JP_PAD_REFLECT};
                launch_auto(a, b, es, Cin, (int)npix, Kp, jp_cdiv(Kp, kps), kps, st);
            });
            const long total = npix * Cin;
            hipLaunchKernelGGL(slice_reduce_nchw_kernel, dim3((int)std::min<long>((total + 255) / 256, 2048)), dim3(256), 0, st,
                               split_ws, dx, (const float*)nullptr, Cin, (int)npix, H * W, jp_cdiv(Kp, kps), JP_ACT_NONE, accumulate);
        } else if (sp > 1) {
            const int kps = jp_cdiv(jp_cdiv(Kp, sp), KC) * KC;
            if (!accumulate) JP_HIP(hipMemsetAsync(dx, 0, sizeof(float) * (size_t)npix * Cin, st));
            AtomicEpi ea{dx, Cin, H * W};
            JP_KH_SWITCH(KH, {
                DgradBT<KH_> b{dy, Cp, (int)npix, H, W, Cout, OH, OW, stride, pad, pad_mode == JP_PAD_REFLECT};
                launch_auto(a, b, ea, Cin, (int)npix, Kp, jp_cdiv(Kp, kps), kps, st);
            });
        } else {
            if (stride == 1) {
                JP_KH_SWITCH(KH, {
                    DgradBT<KH_, true> b{dy, Cp, (int)npix, H, W, Cout, OH, OW, stride, pad, pad_mode == JP_PAD_REFLECT};
                    const int tail = (Cin > 128 && Cin % 128 <= 16) ? Cin % 128 : 0;
                    const int Mm = Cin - tail;
                    const int bn3 = Mm <= 64 ? 256 : 128;
                    if (KH == 1 && pad == 0 && tail == 0 && p9_ok(Cin, Cout, N, H, W, 1)) {
                        float* wfr = ws + dgrad_tap_floats(Cin, Cout, KH);
                        if (!ws_state) pack_p9(w, wfr, Cout, Cin, 1, p9_bmt(Cin, 1, p9_ptiles(N, H, W)), 1, st);
                        launch_p1(wfr, dy, e, Cin, Cout, N, H, W, st);
                    } else
                    if (KH == 3 && pad == 1 && (tail == 0 || Mm > 64) && p9_ok(Mm, Cout, N, H, W)) {
                        // P9 patch kernel on dY (taps mirrored, zero fill; the reflection fold stays with the border pass);
                        // fragment-order pack behind the tap-major one (which the border pass still reads).  With a short
                        // row tail (513 = 4*128 + 1) the kernel runs the full 128-row tiles of the same pack, the tail
                        // its own launch below.
                        float* wfr = ws + dgrad_tap_floats(Cin, Cout, KH);
                        if (!ws_state) pack_p9(w, wfr, Cout, Cin, 1, p9_bmt(Cin, 9, p9_ptiles(N, H, W)), 9, st);
                        launch_p9<false, true>(wfr, dy, e, Mm, Cout, N, H, W, st, 0, Cin);
                    } else
                    if (KH == 3 && pad == 1 && Cout >= 32 && W % bn3 == 0 && npix > 64) {
                        // row-tile kernel on dY (taps mirrored, zero fill; the reflection fold stays with the border pass)
                        const Src3 sdy = make_src(dy, Cout, 0, nullptr, 0, 0, nullptr, 0, 0, H, W);
                        if (Mm <= 64) { FwdBR3<false, true, 256> b3{sdy, H, W}; launch_r3<1, 4>(a, b3, e, Mm, (int)npix, Kp, st); }
                        else { FwdBR3<false, true, 128> b3{sdy, H, W}; launch_r3<2, 2>(a, b3, e, Mm, (int)npix, Kp, st); }
                    } else
                    launch_auto(a, b, e, Mm, (int)npix, Kp, 1, Kp, st);
                    if (tail) {
                        PackA at{ws + (size_t)Mm * Cp, Cin, Kp, Cp, KH * KH};
                        DgradEpi et{dx + (size_t)Mm * H * W, Cin, H * W, accumulate};
                        launch_auto(at, b, et, tail, (int)npix, Kp, 1, Kp, st);
                    }
                });
            } else
            JP_KH_SWITCH(KH, {
                DgradBT<KH_> b{dy, Cp, (int)npix, H, W, Cout, OH, OW, stride, pad, pad_mode == JP_PAD_REFLECT};
                // a short row tail (513 = 4*128 + 1 input channels of the iconv layers) runs as its own 64-row launch
                // instead of a fifth, almost empty 128-row tile column
                const int tail = (Cin > 128 && Cin % 128 <= 16) ? Cin % 128 : 0;
                const int Mm = Cin - tail;
                launch_auto(a, b, e, Mm, (int)npix, Kp, 1, Kp, st);
                if (tail) {
                    PackA at{ws + (size_t)Mm * Cp, Cin, Kp, Cp, KH * KH};
                    DgradEpi et{dx + (size_t)Mm * H * W, Cin, H * W, accumulate};
                    launch_auto(at, b, et, tail, (int)npix, Kp, 1, Kp, st);
                }
            });
        }
        if (pad_mode == JP_PAD_REFLECT)     // fold the reflected ring back in (border-adjacent lines only)
            border_pass(a, dy, dx, Cin, Cp, Kp, Cout, N, H, W, split_ws, st);
    } else {
        const int K = Cout * KH * KH;
        JP_KH_SWITCH(KH, {
            DgradA<KH_> a{w, Cin, K};
            DgradB<KH_> b{dy, K, (int)npix, H, W, Cout, OH, OW, stride, pad, pad_mode == JP_PAD_REFLECT};
            launch_auto(a, b, e, Cin, (int)npix, K, 1, jp_cdiv(K, KC) * KC, st);
        });
    }
    JP_LAUNCH_CHECK();
}


// per-source dgrad of a conv whose input is the channel concat of up to 3 sources, one of them read through the fused
// nearest-2x upsample: gradients go straight into the sources' own buffers (dx_s: (N, c_s, H, W), or (N, c_s, H/2, W/2)
// for the upsampled one; NULL = not needed; acc_s: add instead of overwrite) -- no concat-sized gradient tensor.
static bool dgrad_segments_ok(int c0, int up0, int c1, int up1, int c2, int up2, int N, int H, int W, int Cout, int KH,
                              int stride, int pad, int pad_mode) {
    const int cs[3] = {c0, c1, c2}, us[3] = {up0, up1, up2};
    int nup = 0;
    for (int i = 0; i < 3; ++i)
        if (cs[i] && us[i]) { ++nup; if (cs[i] < 32) return false; }
    return nup == 1 && KH == 3 && stride == 1 && pad == 1 && pad_mode == JP_PAD_REFLECT && H % 2 == 0 && W % 2 == 0 && H >= 4 &&
           W >= 4 && Cout >= 16 && (long)N * H * W < (1L << 31) && (long)N * (H / 2) * (W / 2) / 128 >= 96;
}
extern "C" int jp_conv2d_dgrad_src3_ok(int c0, int up0, int c1, int up1, int c2, int up2, int N, int H, int W, int Cout,
                                       int KH, int stride, int pad, int pad_mode) {
    if (up_head(c0, up0, c1, c2, Cout, KH, stride, pad, pad_mode, H, W)) return 1;
    return dgrad_segments_ok(c0, up0, c1, up1, c2, up2, N, H, W, Cout, KH, stride, pad, pad_mode) ? 1 : 0;
}
// scratch for jp_conv2d_dgrad_src3's `split_ws`: <= 4 K slices of the widest full-resolution segment's border pass
extern "C" long jp_conv2d_dgrad_src3_split_floats(int c0, int up0, int c1, int up1, int c2, int up2, int N, int H, int W) {
    const int cm = std::max(std::max(up0 ? 0 : c0, up1 ? 0 : c1), up2 ? 0 : c2);
    const int cu = std::max(std::max(up0 ? c0 : 0, up1 ? c1 : 0), up2 ? c2 : 0);
    // <= 4 slices of a full-resolution segment's border pass, <= 8 slices of the upsampled segment's edge pass (N * (H + W) pixels)
    return std::max(4L * cm * N * (2L * H + 2L * W), 8L * cu * N * ((long)H + W));
}

extern "C" int jp_conv2d_dgrad_src3(const float* dy, const float* w, float* dx0, int c0, int up0, int acc0, float* dx1,
                                    int c1, int up1, int acc1, float* dx2, int c2, int up2, int acc2, int N, int H, int W,
                                    int Cout, int KH, int stride, int pad, int pad_mode, float* ws, int ws_state,
                                    float* split_ws, const float* amax_dy, float* amax_dx0, int* amax_dx0_done, float* amax_ws,
                                    void* stream) {
    if (amax_dx0_done) *amax_dx0_done = 0;
    // split_ws: optional scratch of jp_conv2d_dgrad_src3_split_floats floats for the border passes (NULL: read-modify-write epilogue)
    JP_CHECK_ARG(dy && w, "conv2d_dgrad_src3: null pointer");
    if (up_head(c0, up0, c1, c2, Cout, KH, stride, pad, pad_mode, H, W)) {
        if (dx0) jp_up_head_dgrad(dy, w, dx0, N, c0, H / 2, W / 2, acc0, (hipStream_t)stream);
        JP_LAUNCH_CHECK();
    }
    JP_CHECK_ARG(ws != nullptr, "conv2d_dgrad_src3: null scratch");
    JP_CHECK_ARG(dgrad_segments_ok(c0, up0, c1, up1, c2, up2, N, H, W, Cout, KH, stride, pad, pad_mode),
                 "conv2d_dgrad_src3: shape not supported (check jp_conv2d_dgrad_src3_ok)");
    JP_CHECK_ARG(JP_NS != 2 || amax_ws || amax_dy, "conv2d_dgrad_src3"": every operand magnitude (amax_*) or amax_ws (jp_conv2d_amax_ws_floats floats of scratch) must be given");
    JpAmaxCtx ax;
    ax.know(dy, amax_dy);
    ax.ws = amax_ws;
    const JpCall st((hipStream_t)stream, &ax);
    const int Cin = c0 + c1 + c2, Cp = pad32(Cout), Kp = 9 * Cp;
    const long npix = (long)N * H * W;
    float* dxs[3] = {dx0, dx1, dx2};
    const int cs[3] = {c0, c1, c2}, us[3] = {up0, up1, up2}, accs[3] = {acc0, acc1, acc2};
    if (!ws_state) pack_weights(w, ws, Cout, Cin, 9, Cp, 1, st);  // [tap][ci][Cp] for the full-resolution segments
    float* wsT = ws + ((size_t)9 * Cin + 256) * Cp;               // [16][Cx][Cp] for the upsampled one
    const int cx_up = up0 ? c0 : (up1 ? c1 : c2);
    DgradBT<3, true> b{dy, Cp, (int)npix, H, W, Cout, H, W, 1, 1, 1};
    int coff = 0;
    bool frag_packed = false;
    for (int sidx = 0; sidx < 3; ++sidx) {
        const int C = cs[sidx];
        if (!C) continue;
        float* dx = dxs[sidx];
        // amax_dx0: max |dx0| out of the first source's passes (main patch kernel + border fold), if it is a full-resolution one
        ax.out = (sidx == 0 && !us[0]) ? reinterpret_cast<unsigned*>(amax_dx0) : nullptr;
        ax.out_taken = false;
        if (dx && !us[sidx]) {
            PackA a{ws + (size_t)coff * Cp, Cin, Kp, Cp, 9};
            if (C <= 4 && (long)C * Cout * 9 * 4 <= 48 * 1024) {
                // a few channels (the disparity channel): direct zero-pad correlation of dY with the flipped taps
                // instead of a 64-row MFMA tile; the reflection fold still comes from the border pass below
                float* wf = wsT + (16L * cx_up + 256) * Cp;     // behind the upsampled segment's pack (+ its slack)
                if (!ws_state) do_pack(PACK_FLIP, w, wf, (long)C * Cout * 9, Cout, Cin, coff, C, 0, 0, st);
                jp_conv_small_fwd(dy, Cout, 0, nullptr, 0, 0, nullptr, 0, 0, wf, nullptr, dx, N, H, W, C, JP_ACT_NONE, 0, st,
                                  accs[sidx]);
            } else {
                DgradEpi e{dx, C, H * W, accs[sidx]};
                const int bn3 = C <= 64 ? 256 : 128;
                if (Cin > 64 && C > 64 && coff % p9_bmt(Cin, 9, p9_ptiles(N, H, W)) == 0 && p9_ok(C, Cout, N, H, W)) {
                    // P9 patch kernel on this segment's 128-row tiles of the whole bank's fragment-order pack
                    float* wfr = ws + dgrad_tap_floats(Cin, Cout, 3);
                    if (!ws_state && !frag_packed) {
                        pack_p9(w, wfr, Cout, Cin, 1, p9_bmt(Cin, 9, p9_ptiles(N, H, W)), 9, st);
                        frag_packed = true;
                    }
                    launch_p9<false, true>(wfr, dy, e, C, Cout, N, H, W, st, coff / p9_bmt(Cin, 9, p9_ptiles(N, H, W)), Cin);
                } else
                if (W % bn3 == 0 && Cout >= 32) {     // row-tile kernel, see jp_conv2d_dgrad
                    const Src3 sdy = make_src(dy, Cout, 0, nullptr, 0, 0, nullptr, 0, 0, H, W);
                    if (C <= 64) { FwdBR3<false, true, 256> b3{sdy, H, W}; launch_r3<1, 4>(a, b3, e, C, (int)npix, Kp, st); }
                    else { FwdBR3<false, true, 128> b3{sdy, H, W}; launch_r3<2, 2>(a, b3, e, C, (int)npix, Kp, st); }
                } else {
                    launch_auto(a, b, e, C, (int)npix, Kp, 1, Kp, st);
                }
            }
            border_pass(a, dy, dx, C, Cp, Kp, Cout, N, H, W, split_ws, st);
            if (sidx == 0 && amax_dx0_done) *amax_dx0_done = ax.out_taken ? 1 : 0;
        } else if (dx) {
            const int h2 = H / 2, w2 = W / 2, KpU = 16 * Cp;
            const long tot = 16L * C * Cp, np2 = (long)N * h2 * w2;
            if (!ws_state) do_pack(PACK_UP_DGRAD, w, wsT, tot, Cout, Cin, coff, C, Cp, 0, st);
            PackA a{wsT, C, KpU, Cp, 16};
            DgradUPB bu{dy, (int)np2, h2, w2, Cout};
            DgradEpi e{dx, C, h2 * w2, accs[sidx]};
            if (p9sd_enabled() && h2 % 4 == 0 && w2 % 32 == 0 && Cout % 16 == 0 && (long)N * Cout * H * W * 4 < (1L << 31) &&
                (long)jp_cdiv(C, 128) * N * (h2 / 4) * (w2 / 32) >= 192) {
                // split-product patch kernel over the full-resolution dY (igemm_p9sd.h); its pack sits behind the P9 one
                float* wsd = ws + dgrad_tap_floats(Cin, Cout, 3) + p9_alloc_floats(Cin, Cp);
                if (!ws_state) do_pack(PACK_SPLITUPD, w, wsd, p9sd_floats(C, Cout), Cout, Cin, coff, C, 0, 0, st);
                const float* xam = JP_NS == 2 ? jp_amax_of(dy, (long)N * Cout * H * W, st) : nullptr;
                jp_prof_before(p9sd_tag<DgradEpi>(), JP_NPROD * 2.0 * C * (double)np2 * 16.0 * Cp, st);
                dim3 grid(N * (h2 / 4) * (w2 / 32), jp_cdiv(C, 128), 1);
                hipLaunchKernelGGL((jp_igemm_p9sd_kernel<DgradEpi>), grid, dim3(256), 0, st, reinterpret_cast<const unsigned*>(wsd), dy, e,
                                   C, Cout, Cp / 16, h2, w2, xam);
                jp_prof_after(st);
            } else
            launch_auto(a, bu, e, C, (int)np2, KpU, 1, KpU, st);
            const int Nb = N * (2 * w2 + 2 * h2);
            DgradUPBorderB bb{dy, Nb, h2, w2, Cout};
            DgradEdgeEpi be{dx, C, h2, w2, Nb, 0};
            const long btiles = (long)jp_cdiv(C, C <= 64 ? 64 : 128) * jp_cdiv(Nb, C <= 64 ? 256 : 128);
            // 16 slots x Cout: a long K loop of branchy gathers on a few dozen tiles -- up to 4 K slices (atomic epilogue)
            const int bsp = (int)std::max<long>(border_splits(btiles, KpU / KC), std::min<long>(4, jp_cdiv(256, btiles)));
            const int bkps = jp_cdiv(jp_cdiv(KpU, bsp), KC) * KC;
            be.split = jp_cdiv(KpU, bkps) > 1;
            if (be.split && split_ws) {
                // K slices to scratch, then one fixed-order fold into dx (round 6: the slices met in float atomics -- run-dependent last
                // bits in the gradient of the upsampled segment, which the whole coarser decoder level inherits)
                const int nsl = std::min(jp_cdiv(KpU, bkps), 8);
                const int wkps = jp_cdiv(jp_cdiv(KpU, nsl), KC) * KC, ns2 = jp_cdiv(KpU, wkps);
                WgradEpiWS es{split_ws, C, Nb};
                launch_auto(a, bb, es, C, Nb, KpU, ns2, wkps, st);
                const long total = (long)C * Nb;
                hipLaunchKernelGGL(edge_add_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, st, split_ws, dx, C,
                                   Nb, h2, w2, ns2);
            } else
            launch_auto(a, bb, be, C, Nb, KpU, jp_cdiv(KpU, bkps), bkps, st);
        }
        coff += C;
    }
    JP_LAUNCH_CHECK();
}

// wgrad of the (virtually concatenated) sources into the dw columns [dw_coff, dw_coff + c0+c1+c2) of a filter bank with
// dw_ctot input channels.  Sub-range calls (dw_coff > 0 or fewer channels than dw_ctot) must be single-source.
static int wgrad_impl(const float* x0, int c0, int up0, const float* x1, int c1, int up1, const float* x2, int c2, int up2,
                      const float* dy, float* dw, int N, int H, int W, int Cout, int KH, int stride, int pad,
                      int pad_mode, float* ws, long ws_floats, const JpCall& st, int dw_ctot, int dw_coff) {
    const int Cin = c0 + c1 + c2;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KH) / stride + 1;
    const long npix = (long)N * OH * OW;
    JP_CHECK_ARG(npix < (1L << 31), "conv2d_wgrad: tensor too large");
    const int Kw = Cin * KH * KH;
    const bool whole = dw_coff == 0 && dw_ctot == Cin;
    if (whole && up_head(c0, up0, c1, c2, Cout, KH, stride, pad, pad_mode, H, W)) {
        jp_up_head_wgrad(x0, dy, dw, N, c0, H / 2, W / 2, st, ws, ws_floats);
        JP_LAUNCH_CHECK();
    }
    if (whole && c1 == 0 && c2 == 0 && jp_c16_ok(c0, Cout, KH, stride, pad, pad_mode, H, W)) {
        jp_c16_wgrad(x0, up0, dy, dw, N, c0, Cout, H, W, st, ws, ws ? ws_floats : 0);
        JP_LAUNCH_CHECK();
    }
    if (whole && small_head(Cin, Cout, KH, stride, pad)) {
        jp_conv_small_wgrad(x0, c0, up0, x1, c1, up1, x2, c2, up2, dy, dw, N, H, W, Cout, pad_mode == JP_PAD_REFLECT, st, ws,
                            ws ? ws_floats : 0);
        JP_LAUNCH_CHECK();
    }
    WgradA a{dy, Cout, (int)npix, OH * OW};
    const Src3 src = make_src(x0, c0, up0, x1, c1, up1, x2, c2, up2, H, W);
    if (!ws) ws_floats = 0;
    // scalar-base loaders: K = (image, pixel padded to a multiple of 32) so a K chunk never straddles two images
    const int Ppad = pad32(OH * OW);
    const long kext = (long)N * Ppad;
    auto plan = [&](int Np, int* splits, int* kps, int per_cu = 2) {   // atomic-epilogue paths
        const bool narrow = Cout <= 64;
        const WgradPlan p = wgrad_plan(Cout, Np, npix, narrow ? 64 : 128, narrow ? 256 : 128, per_cu, 0);
        *splits = p.splits;
        *kps = p.kps;
    };
    // scalar-base paths: scratch-reduced split-K when the caller provided scratch and the plan prefers it
    auto go = [&](auto wm, auto wn, auto a_, auto b_, const WgradEpiT& e, int Np) {
        constexpr int WM = decltype(wm)::value, WN = decltype(wn)::value;
        const WgradPlan p = wgrad_plan(Cout, Np, kext, 64 * WM, 64 * WN, 3, ws_floats);
        if (p.use_ws) {
            WgradEpiWS ew{ws, Cout, Np};
            launch<true, WM, WN>(a_, b_, ew, Cout, Np, (int)kext, p.splits, p.kps, st);
            const long total = (long)Cout * Np;
            if (p.splits >= 32 && total <= (1L << 16))
                hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3((int)((total + 63) / 64)), dim3(1024), 0, st, ws, dw, Cout,
                                   Np, p.splits, e.Cp, e.Cin, e.KHW, e.c_off, e.Ctot);
            else
                hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, st,
                                   ws, dw, Cout, Np, p.splits, e.Cp, e.Cin, e.KHW, e.c_off, e.Ctot);
        } else {
            launch<true, WM, WN>(a_, b_, e, Cout, Np, (int)kext, p.splits, p.kps, st);
        }
    };
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>;
    const bool single = (c1 == 0 && c2 == 0 && up0 == 0 && (long)Cin * H * W < (1L << 31));
    int splits, kps;
    // table path over a channel sub-range [cb, cb+cn) of a single full-resolution source
    auto run_table = [&](int cb, int cn) -> int {
        const int Cp = cn, Np = KH * KH * Cp;    // the slot tables need no channel padding
        const unsigned magic = (unsigned)((1ULL << 32) / (unsigned)Cp) + 1u;
        plan(Np, &splits, &kps);
        WgradEpiT e{dw, Cp, cn, KH * KH, dw_coff + cb, dw_ctot, magic};
        // split-K partial tiles through the caller's scratch when it is large enough (fixed-order fold: bit-reproducible); the
        // atomic epilogue otherwise.  (The scratch is free again here: a main pass that used it has been folded on this stream.)
        const bool via_ws = splits > 1 && ws && (long)splits * Cout * Np <= ws_floats;
        JP_KH_SWITCH(KH, {
            WgradBT1<KH_> b{x0 + (size_t)cb * H * W, Np, Cp, cn, Cin, H, W, (int)npix, OH, OW, stride, pad,
                            pad_mode == JP_PAD_REFLECT, magic};
            if (via_ws) {
                WgradEpiWS ew{ws, Cout, Np};
                launch_auto<true>(a, b, ew, Cout, Np, (int)npix, splits, kps, st);
            } else {
                launch_auto<true>(a, b, e, Cout, Np, (int)npix, splits, kps, st);
            }
        });
        if (via_ws) {
            const long total = (long)Cout * Np;
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, st, ws, dw, Cout,
                               Np, splits, e.Cp, e.Cin, e.KHW, e.c_off, e.Ctot);
        }
        return 0;
    };
    JP_CHECK_ARG(whole || single, "conv2d_wgrad: internal sub-range call must be single-source");
    W1Plan w1;
    if (single && ws && w1_plan(N, Cin, H, W, Cout, KH, stride, pad, ws_floats, &w1)) {
        if (w9s_enabled()) {    // W1S: the same tile and split-K plan on the bf16 pipe (igemm_w9s.h)
            const float* gam = JP_NS == 2 ? jp_amax_of(dy, (long)N * Cout * H * W, st) : nullptr;
            const float* xam = JP_NS == 2 ? jp_amax_of(x0, (long)N * Cin * H * W, st) : nullptr;
            jp_prof_before(w1s_tag(), JP_NPROD * 2.0 * Cout * (double)Cin * N * H * W, st);
            hipLaunchKernelGGL(jp_wgrad_w1s_kernel, dim3(Cin / 128, jp_cdiv(Cout, 256), w1.splits), dim3(512), 0, st, dy, x0, ws, Cout,
                               Cin, Cin, H, W, w1.ntiles, w1.tps, (int)((long)N * Cout * H * W * 4), gam, xam);
        } else {
            jp_prof_before(w1_tag(), 2.0 * Cout * (double)Cin * N * H * W, st);
            hipLaunchKernelGGL(jp_wgrad_w1_kernel, dim3(Cin / 128, jp_cdiv(Cout, 256), w1.splits), dim3(512), 0, st, dy, x0, ws, Cout,
                               Cin, Cin, H, W, w1.ntiles, w1.tps, (int)((long)N * Cout * H * W * 4));
        }
        jp_prof_after(st);
        const long total = (long)Cout * Cin;
        const int nblk = (int)((total / 4 + 63) / 64);
        if (w1.splits >= 64 && nblk < 2048)
            hipLaunchKernelGGL(wgrad_reduce4_kernel<8>, dim3(nblk), dim3(512), 0, st, ws, dw, Cout, Cin, w1.splits, Cin, 1, dw_coff,
                               dw_ctot);
        else
            hipLaunchKernelGGL(wgrad_reduce4_kernel<1>, dim3(nblk), dim3(64), 0, st, ws, dw, Cout, Cin, w1.splits, Cin, 1, dw_coff,
                               dw_ctot);
        JP_LAUNCH_CHECK();
    }
    W7Plan w7;
    if (single && whole && ws && w7_plan(N, Cin, H, W, Cout, KH, stride, pad, ws_floats, &w7)) {
        if (Cin == 3) launch_w7<3>(dy, x0, dw, ws, N, H, W, w7, st);
        else launch_w7<6>(dy, x0, dw, ws, N, H, W, w7, st);
        JP_LAUNCH_CHECK();
    }
    W9S2Plan w92;
    if (single && whole && ws && w9s2_plan(N, Cin, H, W, Cout, KH, stride, pad, pad_mode, ws_floats, &w92)) {
        launch_w9s2(dy, x0, ws, N, Cin, Cin, H, W, Cout, w92, st);
        const long total = (long)Cout * 9 * Cin;
        const int nblk = (int)((total / 4 + 63) / 64);
        if (w92.slices >= 64 && nblk < 2048)
            hipLaunchKernelGGL(wgrad_reduce4_kernel<8>, dim3(nblk), dim3(512), 0, st, ws, dw, Cout, 9 * Cin, w92.slices, Cin, 9,
                               dw_coff, dw_ctot);
        else
            hipLaunchKernelGGL(wgrad_reduce4_kernel<1>, dim3(nblk), dim3(64), 0, st, ws, dw, Cout, 9 * Cin, w92.slices, Cin, 9,
                               dw_coff, dw_ctot);
        JP_LAUNCH_CHECK();
    }
    W9Plan w9;
    // W9 patch kernel on the 64-aligned channels (+ a table pass for a short channel tail, e.g. 513 = 512 + 1): input patch
    // staged once per pixel tile for all 9 taps, dY fragments straight from global memory
    const int c9 = Cin / 64 * 64, tail9 = Cin - c9;
    if (single && ws && (tail9 == 0 || (tail9 <= 32 && c9 >= 128)) && (long)N * Cin * H * W * 4 < (1L << 31) &&
        w9_plan(N, c9, H, W, Cout, KH, stride, pad, ws_floats, &w9)) {
        if (pad_mode == JP_PAD_REFLECT) launch_w9<true>(dy, x0, ws, N, Cin, c9, H, W, Cout, w9, st);
        else launch_w9<false>(dy, x0, ws, N, Cin, c9, H, W, Cout, w9, st);
        const long total = (long)Cout * 9 * c9;
        const int nblk = (int)((total / 4 + 63) / 64);
        if (w9.slices >= 64 && nblk < 2048)          // few outputs, many slices: 8 slice lanes per output quad
            hipLaunchKernelGGL(wgrad_reduce4_kernel<8>, dim3(nblk), dim3(512), 0, st, ws, dw, Cout, 9 * c9, w9.slices, c9, 9,
                               dw_coff, dw_ctot);
        else
            hipLaunchKernelGGL(wgrad_reduce4_kernel<1>, dim3(nblk), dim3(64), 0, st, ws, dw, Cout, 9 * c9, w9.slices, c9, 9,
                               dw_coff, dw_ctot);
        if (tail9) {
            const int rc = run_table(c9, tail9);
            if (rc) return rc;
        }
    } else
    if (!single && Cin < 32) {          // generic (channel-major) path: multi-source inputs with few channels
        WgradEpi e{dw, Kw};
        plan(Kw, &splits, &kps);
        JP_KH_SWITCH(KH, {
            WgradB<KH_> b{src, Kw, (int)npix, OH, OW, stride, pad, pad_mode == JP_PAD_REFLECT};
            launch_auto<true>(a, b, e, Cout, Kw, (int)npix, splits, kps, st);
        });
    } else if (!whole && Cin < 16) {    // a few channels of a wider filter bank (the disparity channel of the iconv input)
        const int rc = run_table(0, Cin);
        if (rc) return rc;
    } else if (single && Cout > 64 && Cin >= 128 && (Cin % 128 == 0 || Cin % 128 <= 32)) {
        // uniform-tap path on the 128-aligned part (+ a table pass for a short channel tail, e.g. 513 = 512 + 1)
        const int Cm = Cin / 128 * 128, tail = Cin - Cm;
        const int Np = KH * KH * Cm;
        const unsigned magic = (unsigned)((1ULL << 32) / (unsigned)Cm) + 1u;
        const bool scalar_ok = Cout % 8 == 0 && (long)8 * H * W * 4 < (1L << 31);
        WgradEpiT e{dw, Cm, Cm, KH * KH, dw_coff, dw_ctot, magic};
        if (scalar_ok) {   // scalar-base loaders
            WgradAS as{dy, Cout, (int)kext, OH * OW, Ppad};
            JP_KH_SWITCH(KH, {
                if (pad_mode == JP_PAD_REFLECT) {
                    WgradBUS<KH_, true> b{x0, Cm, Cin, H, W, OH, OW, stride, pad, Ppad};
                    go(I2{}, I2{}, as, b, e, Np);
                } else {
                    WgradBUS<KH_, false> b{x0, Cm, Cin, H, W, OH, OW, stride, pad, Ppad};
                    go(I2{}, I2{}, as, b, e, Np);
                }
            });
        } else {
            plan(Np, &splits, &kps);
            JP_KH_SWITCH(KH, {
                WgradBU<KH_> b{x0, Cm, Cm, Cin, H, W, (int)npix, OH, OW, stride, pad, pad_mode == JP_PAD_REFLECT};
                launch<true, 2, 2>(a, b, e, Cout, Np, (int)npix, splits, kps, st);
            });
        }
        if (tail) {
            const int rc = run_table(Cm, tail);
            if (rc) return rc;
        }
    } else if (single && (Cin == 64 || ((Cin == 16 || Cin == 32) && Cout <= 64)) && Cout % 8 == 0 &&
               (long)8 * H * W * 4 < (1L << 31)) {
        // 16 / 32 / 64 input channels (ResNet stem + layer1, BEV decoder): whole taps per N tile, scalar-base loaders
        const int Np = KH * KH * Cin;
        const unsigned magic = (unsigned)((1ULL << 32) / (unsigned)Cin) + 1u;
        WgradEpiT e{dw, Cin, Cin, KH * KH, dw_coff, dw_ctot, magic};
        WgradAS as{dy, Cout, (int)kext, OH * OW, Ppad};
#define JP_BMS(REFL, WMv, WNv, TPTv, CPTv)                                              \
    {                                                                                   \
        WgradBMS<KH_, REFL, TPTv, CPTv> b{x0, Cin, H, W, OH, OW, stride, pad, Ppad};    \
        go(std::integral_constant<int, WMv>{}, std::integral_constant<int, WNv>{}, as, b, e, Np); \
    }
        JP_KH_SWITCH(KH, {
            if (pad_mode == JP_PAD_REFLECT) {
                if (Cin == 16) JP_BMS(true, 1, 4, 16, 16)
                else if (Cin == 32) JP_BMS(true, 1, 4, 8, 32)
                else if (Cout <= 64) JP_BMS(true, 1, 4, 4, 64)
                else JP_BMS(true, 2, 2, 2, 64)
            } else {
                if (Cin == 16) JP_BMS(false, 1, 4, 16, 16)
                else if (Cin == 32) JP_BMS(false, 1, 4, 8, 32)
                else if (Cout <= 64) JP_BMS(false, 1, 4, 4, 64)
                else JP_BMS(false, 2, 2, 2, 64)
            }
        });
#undef JP_BMS
    } else if (single) {
        const int rc = run_table(0, Cin);
        if (rc) return rc;
    } else {
        const int Cp = pad32(Cin), Np = KH * KH * Cp;
        const unsigned magic = (unsigned)((1ULL << 32) / (unsigned)Cp) + 1u;
        plan(Np, &splits, &kps);
        WgradEpiT e{dw, Cp, Cin, KH * KH, 0, Cin, magic};
        JP_KH_SWITCH(KH, {
            WgradBT<KH_> b{src, Np, Cp, Cin, (int)npix, OH, OW, stride, pad, pad_mode == JP_PAD_REFLECT, magic};
            launch_auto<true>(a, b, e, Cout, Np, (int)npix, splits, kps, st);
        });
    }
    JP_LAUNCH_CHECK();
}


// ---- W4S (igemm_w4s.h): split-bf16 parity-class wgrad of the upsampled iconv segment; JP_W9S=0 keeps the generic
// WgradAP/WgradBP instantiation.  Same partial-sum layout ws[split][co][16*Cx], folded by wgrad_fold_parity_kernel.
constexpr int W4S_TR = 2;
struct W4SPlan { int splits, tps, ntiles; long need; };
static inline bool w4s_plan(int N, int Cx, int h2, int w2, int Cout, long ws_floats, W4SPlan* p) {
    if (!w9s_enabled() || Cx % 64 || w2 % 32 || h2 % W4S_TR || (long)N * Cout * h2 * w2 * 16 >= (1L << 31)) return false;
    const int ntiles = N * (h2 / W4S_TR) * (w2 / 32);
    const long out_tiles = 2L * (Cx / 64) * jp_cdiv(Cout, 128), per = (long)Cout * 16 * Cx;
    long sp = std::max<long>(1, std::min<long>(jp_cdiv(256, out_tiles), ntiles / 2));
    sp = std::min<long>(sp, ws_floats / per);
    if (sp < 1 || ntiles < 8) return false;
    p->tps = (int)jp_cdiv(ntiles, sp);
    p->splits = jp_cdiv(ntiles, p->tps);
    p->ntiles = ntiles;
    p->need = (long)p->splits * per;
    return true;
}
template <int TR>
const char* w4s_tag() { return __PRETTY_FUNCTION__; }

// one channel segment is eligible for the per-segment wgrad (no materialised concat): full-resolution segments run the
// single-source paths on their own tensor, the upsampled one the parity-class kernels
static bool wgrad_segments_ok(int c0, int up0, int c1, int up1, int c2, int up2, int N, int H, int W, int Cout, int KH,
                              int stride, int pad, int pad_mode) {
    const int cs[3] = {c0, c1, c2}, us[3] = {up0, up1, up2};
    int nseg = 0, nup = 0;
    for (int i = 0; i < 3; ++i) {
        if (!cs[i]) continue;
        ++nseg;
        if (us[i]) {
            ++nup;
            if (cs[i] % 128 || Cout % 8 || Cout <= 64) return false;
        }
    }
    if (nseg < 2 || nup != 1) return false;
    return KH == 3 && stride == 1 && pad == 1 && pad_mode == JP_PAD_REFLECT && H % 2 == 0 && (W / 2) % 32 == 0 && W % 2 == 0 &&
           (long)N * H * W < (1L << 31);
}

extern "C" int jp_conv2d_wgrad_src3(const float* x0, int c0, int up0, const float* x1, int c1, int up1,
                                    const float* x2, int c2, int up2, const float* dy, float* dw, int N, int H, int W,
                                    int Cout, int KH, int stride, int pad, int pad_mode, int accumulate,
                                    float* ws, long ws_floats, const float* amax_x0, const float* amax_x1, const float* amax_x2,
                                    const float* amax_dy, float* amax_ws, void* stream) {
    JP_CHECK_ARG(x0 && dy && dw, "conv2d_wgrad: null pointer");
    JP_CHECK_ARG(JP_NS != 2 || amax_ws || (amax_dy && amax_x0 && (!c1 || amax_x1) && (!c2 || amax_x2)),
                 "conv2d_wgrad"": every operand magnitude (amax_*) or amax_ws (jp_conv2d_amax_ws_floats floats of scratch) must be given");
    JpAmaxCtx ax;
    ax.know(x0, amax_x0);
    ax.know(x1, amax_x1);
    ax.know(x2, amax_x2);
    ax.know(dy, amax_dy);
    ax.ws = amax_ws;
    const JpCall st((hipStream_t)stream, &ax);
    const int Cin = c0 + c1 + c2;
    if (!accumulate) JP_HIP(hipMemsetAsync(dw, 0, sizeof(float) * (size_t)Cout * Cin * KH * KH, st));
    if (ws && wgrad_segments_ok(c0, up0, c1, up1, c2, up2, N, H, W, Cout, KH, stride, pad, pad_mode)) {
        const float* xs[3] = {x0, x1, x2};
        const int cs[3] = {c0, c1, c2}, us[3] = {up0, up1, up2};
        int coff = 0;
        for (int i = 0; i < 3; ++i) {
            if (!cs[i]) continue;
            if (!us[i]) {
                const int rc = wgrad_impl(xs[i], cs[i], 0, nullptr, 0, 0, nullptr, 0, 0, dy, dw, N, H, W, Cout, KH, stride, pad,
                                          pad_mode, ws, ws_floats, st, Cin, coff);
                if (rc) return rc;
            } else {
                const int h2 = H / 2, w2 = W / 2, Cx = cs[i], Np = 16 * Cx;
                const long Ncl = (long)N * h2 * w2;
                W4SPlan q;
                if (w4s_plan(N, Cx, h2, w2, Cout, ws_floats, &q)) {
                    // executed FLOPs: 6 bf16 MFMA products per fp32 product, 16 (class, slot) GEMMs over the half-res pixels
                    const float* gam = JP_NS == 2 ? jp_amax_of(dy, (long)N * Cout * H * W, st) : nullptr;
                    const float* xam = JP_NS == 2 ? jp_amax_of(xs[i], (long)N * Cx * h2 * w2, st) : nullptr;
                    jp_prof_before(w4s_tag<W4S_TR>(), JP_NPROD * 2.0 * Cout * 16.0 * Cx * (double)Ncl, st);
                    hipLaunchKernelGGL((jp_wgrad_w4s_kernel<W4S_TR>), dim3(Cx / 64, jp_cdiv(Cout, 128), 2 * q.splits), dim3(512), 0,
                                       st, dy, xs[i], ws, Cout, Cx, h2, w2, q.ntiles, q.tps, (int)((long)N * Cout * H * W * 4), gam, xam);
                    jp_prof_after(st);
                    const long total = (long)Cout * Cx * 9;
                    hipLaunchKernelGGL(wgrad_fold_parity_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0,
                                       st, ws, dw, Cout, Cx, q.splits, coff, Cin);
                    coff += cs[i];
                    continue;
                }
                WgradPlan p = wgrad_plan(Cout, Np, Ncl, 128, 128, 3, ws_floats);
                if (!p.use_ws) {    // this path always reduces through scratch: take the largest split that fits
                    long sp = std::max<long>(1, std::min<long>(ws_floats / ((long)Cout * Np), jp_cdiv(Ncl, KC) / 4));
                    JP_CHECK_ARG(ws_floats >= (long)Cout * Np, "conv2d_wgrad: scratch too small (jp_conv2d_wgrad_src3_ws_floats)");
                    sp = std::min<long>(sp, 64);
                    p.kps = (int)(jp_cdiv(jp_cdiv(Ncl, KC), sp) * KC);
                    p.splits = jp_cdiv(Ncl, p.kps);
                }
                WgradAP a{dy, Cout, h2, w2, 4 * Cx};
                WgradBP b{xs[i], Cx, h2, w2};
                WgradEpiWS ew{ws, Cout, Np};
                launch<true, 2, 2>(a, b, ew, Cout, Np, (int)Ncl, p.splits, p.kps, st);
                const long total = (long)Cout * Cx * 9;
                hipLaunchKernelGGL(wgrad_fold_parity_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, st,
                                   ws, dw, Cout, Cx, p.splits, coff, Cin);
            }
            coff += cs[i];
        }
        JP_LAUNCH_CHECK();
    }
    return wgrad_impl(x0, c0, up0, x1, c1, up1, x2, c2, up2, dy, dw, N, H, W, Cout, KH, stride, pad, pad_mode, ws, ws_floats, st,
                      Cin, 0);
}

// scratch floats for jp_conv2d_wgrad_src3 on a multi-source input (0 = use the single-source query / no benefit)
extern "C" long jp_conv2d_wgrad_src3_ws_floats(int c0, int up0, int c1, int up1, int c2, int up2, int N, int H, int W,
                                               int Cout, int KH, int stride, int pad, int pad_mode) {
    if (c1 == 0 && c2 == 0 && jp_c16_ok(c0, Cout, KH, stride, pad, pad_mode, H, W)) return jp_c16_wgrad_ws_floats(N, c0, Cout, H, W);
    // disparity head on an upsampled source: the gathered dY sums + the workgroups' partial sums (fixed-order fold)
    if (up_head(c0, up0, c1, c2, Cout, KH, stride, pad, pad_mode, H, W)) return jp_up_head_wgrad_ws_floats(N, c0, H / 2, W / 2);
    if (small_head(c0 + c1 + c2, Cout, KH, stride, pad))
        return jp_conv_small_wgrad_ws_floats(N, c0 + c1 + c2, H, W, Cout, c1 == 0 && c2 == 0 && !up0);
    if (!wgrad_segments_ok(c0, up0, c1, up1, c2, up2, N, H, W, Cout, KH, stride, pad, pad_mode)) return 0;
    const int cs[3] = {c0, c1, c2}, us[3] = {up0, up1, up2};
    const long cap = 48L << 20;
    long need = 0;
    for (int i = 0; i < 3; ++i) {
        if (!cs[i]) continue;
        if (us[i]) {
            const WgradPlan p = wgrad_plan(Cout, 16 * cs[i], (long)N * (H / 2) * (W / 2), 128, 128, 3, cap);
            need = std::max(need, p.use_ws ? p.ws_need : (long)Cout * 16 * cs[i]);
            W4SPlan q;
            if (w4s_plan(N, cs[i], H / 2, W / 2, Cout, cap, &q)) need = std::max(need, q.need);
        } else if (cs[i] >= 16) {
            const int Np = KH * KH * (cs[i] >= 64 ? cs[i] / 64 * 64 : cs[i]);
            const WgradPlan p = wgrad_plan(Cout, Np, (long)N * H * W, 128, 128, 3, cap);
            if (p.use_ws) need = std::max(need, p.ws_need);
        }
    }
    return need;
}

extern "C" int jp_conv2d_wgrad(const float* x, const float* dy, float* dw, int N, int Cin, int H, int W, int Cout,
                               int KH, int stride, int pad, int pad_mode, int accumulate, float* ws, long ws_floats,
                               const float* amax_x, const float* amax_dy, float* amax_ws, void* stream) {
    return jp_conv2d_wgrad_src3(x, Cin, 0, nullptr, 0, 0, nullptr, 0, 0, dy, dw, N, H, W, Cout, KH, stride, pad,
                                pad_mode, accumulate, ws, ws_floats, amax_x, nullptr, nullptr, amax_dy, amax_ws, stream);
}

// floats of optional caller scratch with which jp_conv2d_wgrad[_src3] merges its split-K partial tiles through a
// reduction pass instead of device-scope atomics (0 = no benefit for this shape).  Capped at 32 Mi floats.
extern "C" long jp_conv2d_wgrad_ws_floats(int N, int Cin, int H, int W, int Cout, int KH, int stride, int pad) {
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KH) / stride + 1;
    const long npix = (long)N * OH * OW, cap = 32L << 20;
    if (jp_c16_ok(Cin, Cout, KH, stride, pad, 0, H, W)) return jp_c16_wgrad_ws_floats(N, Cin, Cout, H, W);
    if (small_head(Cin, Cout, KH, stride, pad)) return jp_conv_small_wgrad_ws_floats(N, Cin, H, W, Cout, 1);
    W7Plan w7;
    if (w7_plan(N, Cin, H, W, Cout, KH, stride, pad, cap, &w7)) return w7.need;
    W1Plan w1;
    const long need1 = w1_plan(N, Cin, H, W, Cout, KH, stride, pad, cap, &w1) ? w1.need : 0;
    // the slot-table passes (odd channel counts, the 1-channel tail of the 513-channel iconv banks, Cout % 8 != 0): scratch for their
    // split-K partial tiles so that they are folded in a fixed order instead of meeting in atomics (round 6)
    auto table_need = [&](int cn) -> long {
        const int Npt = KH * KH * cn;
        const bool nrw = Cout <= 64;
        const WgradPlan pt = wgrad_plan(Cout, Npt, npix, nrw ? 64 : 128, nrw ? 256 : 128, 2, 0);
        const long n = (long)pt.splits * Cout * Npt;
        return (pt.splits > 1 && n <= cap) ? n : 0;
    };
    const int tail_c = (Cout > 64 && Cin >= 128 && Cin % 128 != 0 && Cin % 128 <= 32) ? Cin % 128 : 0;
    const long need_t = tail_c ? table_need(tail_c) : table_need(Cin);
    if (Cout % 8 != 0 || Cin < 16) return need_t;
    const int Np = KH * KH * (Cin >= 64 ? Cin / 64 * 64 : Cin);
    const bool narrow = Cout <= 64 && Cin <= 64;
    const WgradPlan p = wgrad_plan(Cout, Np, (long)N * pad32(OH * OW), narrow ? 64 : 128, narrow ? 256 : 128, 3, cap);
    W9Plan w9;
    const long need9 = w9_plan(N, Cin / 64 * 64, H, W, Cout, KH, stride, pad, cap, &w9) ? w9.need : 0;
    W9S2Plan w92;
    const long need92 = w9s2_plan(N, Cin, H, W, Cout, KH, stride, pad, 0, cap, &w92) ? w92.need : 0;
    return std::max(std::max(std::max(std::max(p.use_ws ? p.ws_need : 0, need9), need1), need92), need_t);
}

// ---- weight-pack recording / replay (see do_pack).  `host_jobs`: caller-owned HOST buffer of max_jobs 64-byte records.
extern "C" int jp_pack_job_bytes(void) { return (int)sizeof(JpPackJob); }

extern "C" int jp_pack_record_begin(void* host_jobs, int max_jobs) {
    JP_CHECK_ARG(host_jobs && max_jobs > 0 && !g_pack_rec, "pack_record_begin: bad args or a recording is already open");
    g_pack_rec = (JpPackJob*)host_jobs;
    g_pack_rec_n = 0;
    g_pack_rec_cap = max_jobs;
    return JP_OK;
}

// -> number of packs the conv entry points issued on this thread since jp_pack_record_begin (> max_jobs: overflow)
extern "C" int jp_pack_record_end(void) {
    g_pack_rec = nullptr;
    return g_pack_rec_n;
}

// jobs: DEVICE copy of `njobs` records whose `begin` fields hold the exclusive prefix sum of `total`; total_elems = the sum.
// 1 if a recorded job of this mode is re-packed by the LDS-staged split-pack kernel (its elements are skipped by the generic kernel):
// a caller that puts those jobs BEHIND the others in the table can tell jp_pack_replay where the generic kernel's range ends
extern "C" int jp_pack_mode_is_split(int mode) { return (mode == PACK_SPLIT || mode == PACK_SPLITSEG) ? 1 : 0; }
// generic_elems: elements (begin of the first split-pack job) the generic kernel has to walk when the split-pack jobs are the tail of
// the table; <= 0 or >= total_elems: the whole range (round 6: the generic kernel spent most of its 0.36 ms at the head of every step
// deciding, four elements at a time, that a range belongs to the other kernel)
extern "C" int jp_pack_replay(const void* jobs, int njobs, long total_elems, long generic_elems, void* stream) {
    JP_CHECK_ARG(jobs && njobs > 0 && total_elems > 0, "pack_replay: bad args");
    const long gen = (generic_elems > 0 && generic_elems < total_elems) ? generic_elems : total_elems;
    if (JP_PACK_HDR)    // headers (weight scales) of the fp16 split packs first: the pack kernels below read them
    {
        hipLaunchKernelGGL(pack_scale_zero_kernel, dim3((njobs + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const JpPackJob*)jobs, njobs);
        hipLaunchKernelGGL(pack_scale_reduce_kernel, dim3(PSL, njobs), dim3(256), 0, (hipStream_t)stream, (const JpPackJob*)jobs, njobs);
        hipLaunchKernelGGL(pack_scale_finish_kernel, dim3((njobs + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const JpPackJob*)jobs, njobs);
    }
    hipLaunchKernelGGL(pack_replay_kernel, dim3((int)std::min<long>((gen + 4095) / 4096, 16384)), dim3(256), 0,
                       (hipStream_t)stream, (const JpPackJob*)jobs, njobs, gen);
    // the split-bf16 packs of the table: LDS-staged, grid-stride over their work items (a block without an item returns)
    // (its job-prefix table lives in LDS: 2048 jobs per launch -- a model has a few hundred; longer tables go in slices)
    for (int off = 0; off < njobs; off += 2048)
        hipLaunchKernelGGL(pack_split_replay_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, (const JpPackJob*)jobs + off,
                           std::min(2048, njobs - off));
    JP_LAUNCH_CHECK();
}

#ifdef P9S_TRACE   // debug build only (not part of the ABI): cycle stamps of one 1x1 workgroup, tools/debug/p1_trace.py
extern "C" int dbg_p9s_trace(unsigned long long* host40) {
    return (int)hipMemcpyFromSymbol(host40, HIP_SYMBOL(jp_p9s_trace), 64 * sizeof(unsigned long long));   // host buffer: 64 entries
}
#endif
