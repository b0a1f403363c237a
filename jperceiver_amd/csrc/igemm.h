// One LDS-staged fp32-MFMA implicit-GEMM engine for gfx950:  C[m][n] (+)= sum_k A(m,k) * B(k,n)
// with pluggable gather-loaders for A and B and a pluggable epilogue.  conv2d forward, dgrad and
// wgrad and the 1x1 convs are all instances of this kernel (conv.hip).
//
// Geometry: 4 or 8 wave64 per workgroup.  Every wave owns a 64x64 output tile as 2x2
// v_mfma_f32_32x32x2_f32 accumulators (exact fp32, 64 FLOP/clk/SIMD = the fp32 peak, guide
// cdna_hip_programming.md §3).  Block tile = (64*WM) x (64*WN), WM*WN == 4.  The K loop is chunked
// by KC; each chunk is gathered global->registers (prefetched one chunk ahead, so the loads fly
// under the previous chunk's MFMAs), stored to LDS as As[k][m] / Bs[k][n] (row pad 1 -> both the
// lane-along-k and lane-along-mn store patterns and the MFMA fragment reads are bank-conflict
// free) and consumed with one ds_read_b32 per operand per MFMA.
//
// Loader protocol (all state lives in registers; the functors are read-only kernel arguments):
//   ALONG_K  loader (lanes run along k):  init(st, first_mn, mn_step)  once      -> per-slot tables
//                                          fix(st, k)                   per chunk -> decode this thread's k
//                                          get(st, mn, r)               per element (r = compile-time slot)
//   ALONG_MN loader (lanes run along m/n): init(st, mn)                  once      -> decode this thread's m/n
//                                          chunk(st, kc)                 per chunk -> chunk-uniform part
//                                          get(st, kl, r)                per element (kl = k - kc)
//
// Scalar-base loaders (SPLIT / POST below): a loader may split every address into a wave-uniform 64-bit base
// (SGPRs, advanced with SALU) plus a per-lane 32-bit byte offset that is constant over a chunk, so a gathered
// element costs one global_load_dword (saddr form) and zero VALU -- the VALU port is what the MFMA stream
// shares.  ALONG_K loaders flag it with SPLIT and implement get_u(st, uniform_mn, r); ALONG_MN loaders get a
// wave-uniform k row for free (tile widths are multiples of 64).  POST loaders mask a loaded value when it is
// stored to LDS (zero padding) instead of branching around the load.
#pragma once
#include "jp_common.h"
#include <type_traits>

template <class L, class = void> struct jp_has_post : std::false_type {};
template <class L> struct jp_has_post<L, std::void_t<decltype(L::POST)>> : std::true_type {};
template <class E, class = void> struct jp_epi_wants_slice : std::false_type {};   // put(st, m, v, k_slice)
template <class E> struct jp_epi_wants_slice<E, std::void_t<decltype(E::WANTS_SLICE)>> : std::true_type {};
template <class L, class = void> struct jp_has_all_ok : std::false_type {};   // bool all_ok(st): wave-uniform "nothing to mask"
template <class L> struct jp_has_all_ok<L, std::void_t<decltype(L::ALL_OK)>> : std::true_type {};
template <class L, class = void> struct jp_wants_tile : std::false_type {};   // init(st, first, step, m0, n0)
template <class L> struct jp_wants_tile<L, std::void_t<decltype(L::WANTS_TILE)>> : std::true_type {};
template <class L, class = void> struct jp_has_skip : std::false_type {};    // unsigned tile_mask(st); bool skip(mask, kc)
template <class L> struct jp_has_skip<L, std::void_t<decltype(L::SKIP)>> : std::true_type {};
template <class L, class = void> struct jp_has_split : std::false_type {};
template <class L> struct jp_has_split<L, std::void_t<decltype(L::SPLIT)>> : std::true_type {};

typedef float jp_f32x16 __attribute__((ext_vector_type(16)));

template <int WM, int WN, int KC, class ALoad, class BLoad, class Epi, bool DB = false, bool IL = false>
__global__ __launch_bounds__(64 * WM * WN) void jp_igemm_kernel(ALoad al, BLoad bl, Epi epi, int M, int N, int K,
                                                               int k_per_split) {
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves per block");
    constexpr int NT = 64 * WM * WN;             // threads per workgroup
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int LDA = BM + 1, LDB = BN + 1;
    constexpr int NA = BM * KC / NT, NB = BN * KC / NT;  // elements per thread per chunk
    // DB: two LDS stages -> the next chunk is stored while the current one is consumed, one barrier per chunk
    __shared__ float As[(DB ? 2 : 1) * KC * LDA];
    __shared__ float Bs[(DB ? 2 : 1) * KC * LDB];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs in dispatch order (x fastest), each with
    // its own 4 MiB L2.  Remap so that (a) every XCD owns a contiguous band of N tiles -- vertically adjacent pixel
    // tiles, which share two of their three input rows, then hit the same L2 instead of each pulling the rows over
    // the fabric -- and (b) the gridDim.y M-tiles of one N-tile are consecutive dispatches on that XCD and share
    // the gathered B operand.
    // (c) split-K launches (gridDim.z > 1): every XCD owns whole K slices -- all (m, n) tiles of a slice read the same
    // pixel range of both operands in lock-step, so they share it through one L2 instead of 8.
    int mt = blockIdx.y, nt = blockIdx.x, zs = blockIdx.z;
    {
        const int gx = gridDim.x, gy = gridDim.y;
        const int L = blockIdx.x + blockIdx.y * gx;
        if (gridDim.z > 1) {
            const int T = gx * gy, SG = gridDim.z & ~7;
            const int L3 = L + blockIdx.z * T;
            int tile;
            if (L3 < SG * T) {
                const int idx = L3 >> 3;
                zs = (idx / T) * 8 + (L3 & 7);
                tile = idx % T;
            } else {
                const int r = L3 - SG * T;
                zs = SG + r / T;
                tile = r % T;
            }
            mt = tile % gy;
            nt = tile / gy;
        } else {
            const int G = gx & ~7;
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
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const int kbeg = zs * k_per_split;
    const int kend = min(K, kbeg + k_per_split);

    // ---- loader thread mappings
    // ALONG_K  : kk = t % KC, mn = t / KC + (256/KC) * r
    // ALONG_MN : mn = t % B,  kk = t / B + (256/B) * r      (B <= 256)
    constexpr int A_ROWS = ALoad::ALONG_K ? (NT / KC) : (NT / BM);
    constexpr int B_ROWS = BLoad::ALONG_K ? (NT / KC) : (NT / BN);
    static_assert(BM <= NT && BN <= NT, "lanes-along-m/n mapping needs tile width <= threads");
    const int a_fix_l = ALoad::ALONG_K ? (t % KC) : (t % BM);
    const int a_var_l = ALoad::ALONG_K ? (t / KC) : __builtin_amdgcn_readfirstlane(t / BM);   // wave-uniform
    const int b_fix_l = BLoad::ALONG_K ? (t % KC) : (t % BN);
    const int b_var_l = BLoad::ALONG_K ? (t / KC) : __builtin_amdgcn_readfirstlane(t / BN);
    typename ALoad::St sa;
    typename BLoad::St sb;
    if constexpr (ALoad::ALONG_K) {
        if constexpr (jp_wants_tile<ALoad>::value) al.init(sa, m0 + a_var_l, A_ROWS, m0, n0);
        else al.init(sa, m0 + a_var_l, A_ROWS);
    } else if constexpr (jp_has_post<ALoad>::value) al.init(sa, m0 + a_fix_l, m0);
    else al.init(sa, m0 + a_fix_l);
    if constexpr (BLoad::ALONG_K) {
        if constexpr (jp_wants_tile<BLoad>::value) bl.init(sb, n0 + b_var_l, B_ROWS, m0, n0);
        else bl.init(sb, n0 + b_var_l, B_ROWS);
    } else if constexpr (jp_has_post<BLoad>::value) bl.init(sb, n0 + b_fix_l, n0);
    else bl.init(sb, n0 + b_fix_l);

    float ra[NA], rb[NB];
    auto getA = [&](int r) -> float {
        if constexpr (ALoad::ALONG_K) {
            if constexpr (jp_has_split<ALoad>::value) return al.get_u(sa, m0 + A_ROWS * r, r);
            else return al.get(sa, m0 + a_var_l + A_ROWS * r, r);
        } else {
            return al.get(sa, a_var_l + A_ROWS * r, r);
        }
    };
    auto getB = [&](int r) -> float {
        if constexpr (BLoad::ALONG_K) {
            if constexpr (jp_has_split<BLoad>::value) return bl.get_u(sb, n0 + B_ROWS * r, r);
            else return bl.get(sb, n0 + b_var_l + B_ROWS * r, r);
        } else {
            return bl.get(sb, b_var_l + B_ROWS * r, r);
        }
    };
    auto gload = [&](int kc) {
        if constexpr (ALoad::ALONG_K) al.fix(sa, kc + a_fix_l);
        else al.chunk(sa, kc);
#pragma unroll
        for (int r = 0; r < NA; ++r) ra[r] = getA(r);
        if constexpr (BLoad::ALONG_K) bl.fix(sb, kc + b_fix_l);
        else bl.chunk(sb, kc);
#pragma unroll
        for (int r = 0; r < NB; ++r) rb[r] = getB(r);
    };
    auto lstore = [&](int stage) {
        float* Ad = As + stage * (KC * LDA);
        float* Bd = Bs + stage * (KC * LDB);
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            float v = ra[r];
            if constexpr (jp_has_post<ALoad>::value) v = al.post(sa, v, r);
            if (ALoad::ALONG_K) Ad[a_fix_l * LDA + a_var_l + A_ROWS * r] = v;
            else Ad[(a_var_l + A_ROWS * r) * LDA + a_fix_l] = v;
        }
        // interior waves (every lane's taps inside the image) skip the per-element mask selects: one scalar branch
        bool maskB = jp_has_post<BLoad>::value;
        if constexpr (jp_has_all_ok<BLoad>::value) maskB = !bl.all_ok(sb);
        if (maskB) {
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                float v = rb[r];
                if constexpr (jp_has_post<BLoad>::value) v = bl.post(sb, v, r);
                if (BLoad::ALONG_K) Bd[b_fix_l * LDB + b_var_l + B_ROWS * r] = v;
                else Bd[(b_var_l + B_ROWS * r) * LDB + b_fix_l] = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                if (BLoad::ALONG_K) Bd[b_fix_l * LDB + b_var_l + B_ROWS * r] = rb[r];
                else Bd[(b_var_l + B_ROWS * r) * LDB + b_fix_l] = rb[r];
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

    const int l31 = lane & 31, lhi = lane >> 5;
    const float* ap = As + lhi * LDA + wm * 64 + l31;
    const float* bp = Bs + lhi * LDB + wn * 64 + l31;
    // a wave whose whole 64x64 sub-tile lies outside the problem (N = 147 in a 256-wide tile: the 7x7 stem wgrad) only
    // stages data and meets the barriers; its MFMA slots go to the other workgroups resident on the SIMD
    const bool wave_live = (m0 + wm * 64 < M) && (n0 + wn * 64 < N);

    auto compute = [&](int stage) {
        if (!wave_live) return;
        const float* aq = ap + stage * (KC * LDA);
        const float* bq = bp + stage * (KC * LDB);
        // operands of step kk+2 are read from LDS before the MFMAs of step kk issue (the scheduling barriers keep
        // the compiler from re-serialising read -> wait -> MFMA), so the LDS latency hides under the matrix pipe
        float a0 = aq[0], a1 = aq[32], b0 = bq[0], b1 = bq[32];
#pragma unroll 4
        for (int kk = 0; kk < KC; kk += 2) {
            const int kn = kk + 2 < KC ? kk + 2 : kk;   // the last step re-reads its own operands (unused)
            const float na0 = aq[kn * LDA], na1 = aq[kn * LDA + 32];
            const float nb0 = bq[kn * LDB], nb1 = bq[kn * LDB + 32];
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
    };
    if constexpr (DB) {
        if (kbeg < kend) {
            gload(kbeg);
            lstore(0);
        }
        __syncthreads();
        int cur = 0;
        for (int kc = kbeg; kc < kend; kc += KC) {
            const bool more = kc + KC < kend;
            if (more) gload(kc + KC);   // in flight during the MFMAs below
            compute(cur);
            if (more) lstore(cur ^ 1);  // the other stage was last read before the previous barrier
            __syncthreads();
            cur ^= 1;
        }
    } else if constexpr (IL) {
        // interleaved variant: the next chunk's gather (address VALU + global loads) is spread between the MFMAs of
        // the current chunk, so the wave keeps feeding the matrix pipe while it does its integer work
        constexpr int STEPS = KC / 2;
        if (kbeg < kend) gload(kbeg);
        for (int kc = kbeg; kc < kend; kc += KC) {
            lstore(0);
            __syncthreads();
            const bool more = kc + KC < kend;
            const int kn = kc + KC;
            if (more) {
                if constexpr (ALoad::ALONG_K) al.fix(sa, kn + a_fix_l); else al.chunk(sa, kn);
                if constexpr (BLoad::ALONG_K) bl.fix(sb, kn + b_fix_l); else bl.chunk(sb, kn);
            }
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                if (more) {
#pragma unroll
                    for (int r = s * NA / STEPS; r < (s + 1) * NA / STEPS; ++r) ra[r] = getA(r);
#pragma unroll
                    for (int r = s * NB / STEPS; r < (s + 1) * NB / STEPS; ++r) rb[r] = getB(r);
                }
                const int kk = 2 * s;
                if (wave_live) {
                    const float a0 = ap[kk * LDA], a1 = ap[kk * LDA + 32];
                    const float b0 = bp[kk * LDB], b1 = bp[kk * LDB + 32];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    } else {
        // SKIP loaders name K chunks that are zero for the whole pixel tile (workgroup-uniform): those are stepped over
        unsigned tmask = 0;
        if constexpr (jp_has_skip<BLoad>::value) tmask = bl.tile_mask(sb);
        auto adv = [&](int kc) {
            if constexpr (jp_has_skip<BLoad>::value)
                while (kc < kend && bl.skip(tmask, kc)) kc += KC;
            return kc;
        };
        int kc = adv(kbeg);
        if (kc < kend) gload(kc);
        while (kc < kend) {
            lstore(0);
            __syncthreads();
            const int kn = adv(kc + KC);
            if (kn < kend) gload(kn);  // in flight during the MFMAs below
            compute(0);
            __syncthreads();
            kc = kn;
        }
    }

    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + l31;
        if (n >= N) continue;
        const typename Epi::St se = epi.col(n);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (m < M) {
                    if constexpr (jp_epi_wants_slice<Epi>::value) epi.put(se, m, acc[i][j][r], zs);
                    else epi.put(se, m, acc[i][j][r]);
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Row-tile variant for 3x3 stride-1 pad-1 convolutions whose pixel tile is one segment of an image row (W % BN == 0):
// K order (channel chunk, dy, dx, c), and the three dx taps of one (chunk, dy) read the SAME input row segment shifted
// by one pixel.  The segment (+ one halo pixel on each side) is staged in LDS once per (chunk, dy) and the MFMA B
// fragments of tap dx are read at column offset dx (forward) / 2 - dx (dgrad): a third of the B gathers and B LDS
// stores of the generic kernel, all of them aligned, with wave-uniform row pointers and a per-lane offset that is
// constant for the whole kernel.
// BLoad protocol:  init(st, n0) | row(st, kc) per (chunk, dy) | get(st, kl) centre element of k row kl |
//                  halo(st, kl, side) column -1 / BN of k row kl | REVERSE: tap dx reads column offset 2 - dx
template <int WM, int WN, int KC, class ALoad, class BLoad, class Epi>
__global__ __launch_bounds__(64 * WM * WN) void jp_igemm_r3_kernel(ALoad al, BLoad bl, Epi epi, int M, int N, int K) {
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(ALoad::ALONG_K && jp_has_split<ALoad>::value, "A: packed weights, scalar-base");
    constexpr int NT = 256, BM = 64 * WM, BN = 64 * WN;
    constexpr int LDA = BM + 1, LDB = BN + 3;            // B rows: [halo | BN pixels | halo], odd stride
    constexpr int NA = BM * KC / NT, NB = BN * KC / NT;
    constexpr int A_ROWS = NT / KC, B_ROWS = NT / BN;
    __shared__ float As[KC * LDA];
    __shared__ float Bs[KC * LDB];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
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
    const int m0 = mt * BM, n0 = nt * BN;
    const int a_fix_l = t % KC, a_var_l = t / KC;
    const int b_fix_l = t % BN, b_var_l = __builtin_amdgcn_readfirstlane(t / BN);
    typename ALoad::St sa;
    typename BLoad::St sb;
    al.init(sa, m0 + a_var_l, A_ROWS);
    bl.init(sb, n0);
    float ra[NA], rb[NB], rh = 0.f;
    auto gload = [&](int kc) {
        al.fix(sa, kc + a_fix_l);
#pragma unroll
        for (int r = 0; r < NA; ++r) ra[r] = al.get_u(sa, m0 + A_ROWS * r, r);
        if ((kc / KC) % 3 == 0) {      // first dx tap of a (chunk, dy): fetch the row segment
            bl.row(sb, kc);
#pragma unroll
            for (int r = 0; r < NB; ++r) rb[r] = bl.get(sb, b_var_l + B_ROWS * r, b_fix_l);
            if (wave == 0) rh = bl.halo(sb, lane >> 1, lane & 1);
        }
    };
    auto lstore = [&](int kc) {
#pragma unroll
        for (int r = 0; r < NA; ++r) As[a_fix_l * LDA + a_var_l + A_ROWS * r] = ra[r];
        if ((kc / KC) % 3 == 0) {
#pragma unroll
            for (int r = 0; r < NB; ++r) Bs[(b_var_l + B_ROWS * r) * LDB + 1 + b_fix_l] = rb[r];
            if (wave == 0) Bs[(lane >> 1) * LDB + ((lane & 1) ? BN + 1 : 0)] = rh;
        }
    };
    jp_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int l31 = lane & 31, lhi = lane >> 5;
    const float* ap = As + lhi * LDA + wm * 64 + l31;
    const float* bp = Bs + lhi * LDB + wn * 64 + l31;
    gload(0);
    for (int kc = 0; kc < K; kc += KC) {
        lstore(kc);
        __syncthreads();
        if (kc + KC < K) gload(kc + KC);
        const int dx = (kc / KC) % 3;
        const float* bq = bp + (BLoad::REVERSE ? 2 - dx : dx);
        float a0 = ap[0], a1 = ap[32], b0 = bq[0], b1 = bq[32];
#pragma unroll 4
        for (int kk = 0; kk < KC; kk += 2) {
            const int kn = kk + 2 < KC ? kk + 2 : kk;
            const float na0 = ap[kn * LDA], na1 = ap[kn * LDA + 32];
            const float nb0 = bq[kn * LDB], nb1 = bq[kn * LDB + 32];
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + l31;
        if (n >= N) continue;
        const typename Epi::St se = epi.col(n);
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
