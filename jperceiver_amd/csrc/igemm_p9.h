// "P9" patch kernel: 3x3 stride-1 pad-1 convolution (forward, and the dgrad main pass as the same correlation with
// reversed taps) on the fp32 MFMA pipe with NINE-fold reuse of the staged input.
//
// The generic engine (igemm.h) gathers a fresh 32 x BN operand tile for every (channel chunk, tap): each input element
// is fetched from L2 and written to LDS 9 times per M tile (3 times in the row-tile kernel).  Here a workgroup owns a
// 2-D pixel tile of TR rows x 32 columns; per 32-channel chunk it stages the (TR+2) x 34 input PATCH once -- padding
// (zero or reflection) resolved while staging -- and all 9 taps read their MFMA B fragments from that patch at a
// constant LDS offset (dy*PITCH + dx).  The weights (A operand) do not go through LDS at all: they are pre-packed in
// MFMA fragment order [M tile][quad of 4 k-steps][k parity][row in tile][4] (PACK_FRAG, conv.hip: a workgroup's weight
// stream is one contiguous run, 1 KB per k-step), so a lane's A values for FOUR k-steps are ONE 16-byte global load (L1/L2-resident: every workgroup of an M tile walks the same stream), prefetched a fixed
// number of k-steps ahead into a register ring.  Per 64 MFMAs a wave therefore issues 32 coalesced weight loads,
// 2 x 16 ds_read_b32 and ~3 patch loads / LDS stores (vs 32 gathered loads + 32 LDS stores before), and a workgroup
// meets 2 barriers per 576 MFMAs per wave instead of per 64.
//
// Geometry: 4 waves; wave (wm, wn) owns rows {2wn, 2wn+1} of the pixel tile (j = 0, 1; 32 pixels = one MFMA N
// block each) and 64 output channels (i = 0, 1): the same 2x2 grid of 32x32 accumulators as igemm.h.
//   WM=2, WN=2: 128 channels x (4 rows x 32 cols);   WM=1, WN=4: 64 channels x (8 rows x 32 cols)
// Preconditions (host-checked): W % 32 == 0, H % (2*WN) == 0, reduction channels padded to 32 in the pack.
// mt_off: first M tile of the pack this launch works on (a channel segment of a wider filter bank: rows are relative).
#pragma once
#include "igemm.h"

constexpr int P9_PITCH = 36;          // patch row pitch in floats: [halo | 32 pixels | halo] + 2 pad
constexpr int P9_QAHEAD = 1;          // quads (4 k-steps) of weight prefetch: register ring of (P9_QAHEAD + 1) x 2 x float4
typedef float jp_p9_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned jp_p9_u32x4 __attribute__((ext_vector_type(4)));

// TAPS = 9: 3x3 (patch with a one-pixel halo).  TAPS = 1: 1x1 convolution ("P1": no halo; CPB = 2 channel chunks are
// staged per barrier pair so that the barrier density stays at 2 per 128 MFMAs).
template <int WM, int WN, bool REFLECT, bool REV, class Epi, int TAPS = 9, int CPB = 1>
__global__ __launch_bounds__(64 * WM * WN, 2) void jp_igemm_p9_kernel(const float* __restrict__ wp, const float* __restrict__ x,
                                                          Epi epi, int M, int C, int NCH, int H, int W, int mt_off) {
    static_assert(WM * WN == 4 || (WM == 4 && WN == 2), "4 waves, or 8 waves = 256 channels x 4 rows");
    constexpr int NT = 64 * WM * WN, NHW = NT / 32;         // threads, half-waves per block
    static_assert(TAPS == 9 || TAPS == 1, "3x3 or 1x1");
    constexpr int HALO = TAPS == 9 ? 1 : 0;
    constexpr int CS = 32 * CPB;                            // channels staged per barrier pair
    constexpr int TR = 2 * WN, PR = TR + 2 * HALO;          // tile rows, patch rows
    constexpr int NROW = CS * PR / NHW;                     // patch rows (c, pr) per NHW-row group of the block: loads per thread
    static_assert(CS % NHW == 0, "channel groups per patch row");
    constexpr int STEPS = TAPS * 16 * CPB;                  // k-steps (of 2) per stage
    __shared__ float patch[CS * PR * P9_PITCH];

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
    const int tiles_x = W / 32, tiles_y = H / TR;
    const int img = nt / (tiles_x * tiles_y), tr_ = nt - img * (tiles_x * tiles_y);
    const int y0 = (tr_ / tiles_x) * TR, x0 = (tr_ % tiles_x) * 32;
    const int m0 = mt * 64 * WM;
    const long HW = (long)H * W;
    const float* xin = x + (long)img * C * HW;              // channel c of this image at xin + c*HW

    // ---- patch staging map.  Patch row rho' = pr*CS + c; the 8 half-waves of the block take rows rho' = 8*r + w8
    // (r = 0 .. NROW-1), lanes run along the 32 centre columns.  pr = r/(CS/8) is compile-time, c = 8*(r%(CS/8)) + w8.
    const int w8 = t >> 5, l32 = t & 31;
    long rowoff[PR];                                        // uniform: source row offset of patch row pr, or -1 (zero row)
#pragma unroll
    for (int pr = 0; pr < PR; ++pr) {
        int yy = y0 - HALO + pr;
        if (REFLECT) yy = jp_reflect(yy, H);
        rowoff[pr] = (yy >= 0 && yy < H) ? (long)yy * W : -1;
    }
    const long lane_off = (long)w8 * HW + x0 + l32;        // + NHW*(r%(CS/NHW))*HW + rowoff[pr] + chunk*32*HW
    // halo columns: element e = t + 256*q < 64*PR: side = e & 1, rho' = e >> 1
    int xl = x0 - 1, xr = x0 + 32;
    if (REFLECT) { xl = jp_reflect(xl, W); xr = jp_reflect(xr, W); }
    const bool okl = xl >= 0, okr = xr < W;

    constexpr int NHALO = HALO ? (2 * CS * PR + NT - 1) / NT : 0;
    float rb[NROW], rh[NHALO > 0 ? NHALO : 1];
    // (Measured and rejected: spreading the next patch's loads one by one over the current chunk's k-steps instead of
    // issuing them as one burst behind the barrier -- 128 -> 83 TF.)
    auto gload_row = [&](const float* xc, int r) {
        const int pr = r / (CS / NHW);
        const long ro = rowoff[pr];
        rb[r] = ro >= 0 ? xc[lane_off + (long)(NHW * (r % (CS / NHW))) * HW + ro] : 0.f;
    };
    auto gload_halo = [&](const float* xc, int q) {
        const int e = t + NT * q;
        float v = 0.f;
        if (e < 2 * CS * PR) {
            const int side = e & 1, rp = e >> 1, pr = rp / CS, c = rp % CS;
            int yy = y0 - 1 + pr;
            if (REFLECT) yy = jp_reflect(yy, H);
            const bool ok = yy >= 0 && yy < H && (side ? okr : okl);
            if (ok) v = xc[(long)c * HW + (long)yy * W + (side ? xr : xl)];
        }
        rh[q] = v;
    };
    auto gload = [&](int ch) {
        const float* xc = xin + (long)ch * CS * HW;
#pragma unroll
        for (int r = 0; r < NROW; ++r) gload_row(xc, r);
#pragma unroll
        for (int q = 0; q < NHALO; ++q) gload_halo(xc, q);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
            const int pr = r / (CS / NHW), c = NHW * (r % (CS / NHW)) + w8;
            patch[(c * PR + pr) * P9_PITCH + 1 + l32] = rb[r];
        }
#pragma unroll
        for (int q = 0; q < NHALO; ++q) {
            const int e = t + NT * q;
            if (e < 2 * CS * PR) {
                const int side = e & 1, rp = e >> 1, pr = rp / CS, c = rp % CS;
                patch[(c * PR + pr) * P9_PITCH + (side ? 33 : 0)] = rh[q];
            }
        }
    };

    jp_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- weight stream of this M tile: quad Q (4 k-steps, global over stages) = 2 x BMT float4 at wt + Q*2*BMT; lane
    // (l31, lhi) reads [lhi][wm*64 + i*32 + l31] -- ONE global_load_dwordx4 per row block per 4 k-steps: a wave-uniform
    // base that advances by a compile-time 2*BMT per quad + one per-lane offset register
    constexpr int BMT = 64 * WM, QS = STEPS / 4, RING = P9_QAHEAD + 1;
    static_assert(STEPS % 4 == 0 && QS % RING == 0, "ring slots must line up across stages");
    // buffer addressing: SGPR resource of this M tile's stream + per-lane byte offset (constant) + scalar quad offset, so a
    // load costs no VALU and no 64-bit address registers
    constexpr int QBYTES = 2 * BMT * 16;                       // bytes per quad
    const long tile_bytes = ((long)NCH * QS + P9_QAHEAD + 1) * QBYTES;                  // NCH = number of STAGES here
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(wp)) + (long)(mt + mt_off) * tile_bytes, 0, (int)tile_bytes, 0x00020000);
    const int avo = (lhi * BMT + wm * 64 + l31) * 16;
    float ra[RING][2][4];
    auto aload = [&](int slot, int quad_bytes) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const jp_p9_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(wrs, avo + 512 * i, quad_bytes, 0);
            const jp_p9_f32x4 v = __builtin_bit_cast(jp_p9_f32x4, u);
#pragma unroll
            for (int j = 0; j < 4; ++j) ra[slot][i][j] = v[j];
        }
    };
#pragma unroll
    for (int d = 0; d < P9_QAHEAD; ++d) aload(d, d * QBYTES);
    const float* bp = patch + (lhi * PR + 2 * wn) * P9_PITCH + l31;

    gload(0);
    for (int ch = 0; ch < NCH; ++ch) {
        lstore();
        __syncthreads();
        if (ch + 1 < NCH) gload(ch + 1);                   // next stage's patch: in flight during the MFMAs below
        const int aq = __builtin_amdgcn_readfirstlane(ch * QS * QBYTES);
        // B fragments are read one k-step ahead of the MFMAs that use them (offsets are compile-time: the loop over the
        // 144 k-steps of the chunk is fully unrolled)
        auto boff = [&](int q) -> int {
            const int cc = q / (TAPS * 16), tap = (q / 16) % TAPS, s = q % 16;       // pack order: chunk, tap, k-pair
            const int dy = TAPS == 1 ? 0 : (REV ? 2 - tap / 3 : tap / 3), dx = TAPS == 1 ? 1 : (REV ? 2 - tap % 3 : tap % 3);
            return ((cc * 32 + 2 * s) * PR + dy) * P9_PITCH + dx;
        };
        float b0 = bp[boff(0)], b1 = bp[boff(0) + P9_PITCH];
#pragma unroll
        for (int q = 0; q < STEPS; ++q) {
            const int qn = q + 1 < STEPS ? q + 1 : q;
            const float nb0 = bp[boff(qn)], nb1 = bp[boff(qn) + P9_PITCH];
            const int Q = q / 4, j = q % 4;
            if (j == 0) {
                // refill the ring slot of quad Q - 1 with the weights of quad Q + QAHEAD (the stream continues into the next
                // stage; the pack carries QAHEAD quads of slack past the end)
                aload((Q + P9_QAHEAD) % RING, aq + (Q + P9_QAHEAD) * QBYTES);
            }
            const float a0 = ra[Q % RING][0][j], a1 = ra[Q % RING][1][j];
            // keep the software pipeline as written (loads of quad Q+QAHEAD / step q+1 issue before the MFMAs of step q; the
            // scheduler must not hoist the whole unrolled chunk's loads to the front: 300+ live registers)
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            b0 = nb0; b1 = nb1;
        }
        __syncthreads();
    }

    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = img * (int)HW + (y0 + 2 * wn + j) * W + x0 + l31;
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
