// Small-map launches of the split-bf16 patch kernel (igemm_p9s.h, MASK = true): 3x3 stride-1 pad-1 forward / dgrad main pass
// and 1x1 layers whose maps are not multiples of the 4x32 / 8x32 pixel tile (pose encoder: 24x80, 12x40, 6x20 at 16 images) or
// whose tile grid cannot fill 256 CUs (BEV / layout heads: 32x32 ... 8x8 at 8 images).  Until round 4 these ran on the generic
// exact-fp32 engine (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 MFMA rate) with split-K; here
//   * partial tiles are masked (zeros staged outside the map, stores only inside), and
//   * the reduction (16-channel stages) is split over grid.z when there are fewer than ~192 tiles; every slice writes its
//     partial tile to caller scratch part[slice][m][pixel] and the caller folds the slices in a fixed order
//     (slice_reduce_nchw_kernel, conv.hip) -- bit-reproducible, no atomics.
// Same weight pack as P9S (PACK_SPLIT fragment order, pack_p9 in conv.hip).  JP_P9SM=0 turns the path off.
#include <algorithm>
#include <cstdlib>

#include "igemm_p9s.h"
#include "conv_p9sm.h"
#include "scale.h"

namespace {

struct SmFwdEpi {  // y[img][co][pix] = act(acc + bias[co])
    typedef size_t St;
    float* y;
    const float* bias;
    int Cout, OHW, act, slice;
    __device__ __forceinline__ St col(int p) const {
        const int img = p / OHW;
        return (size_t)img * Cout * OHW + (p - img * OHW);
    }
    __device__ __forceinline__ void put(St base, int m, float v) const {
        if (bias) v += bias[m];
        y[base + (size_t)m * OHW] = jp_act(v, act);
    }
};
struct SmDgradEpi {  // dx[img][ci][pix] (= or +=) acc
    typedef size_t St;
    float* dx;
    int Cin, HW, accumulate, slice;
    __device__ __forceinline__ St col(int p) const {
        const int img = p / HW;
        return (size_t)img * Cin * HW + (p - img * HW);
    }
    __device__ __forceinline__ void put(St base, int m, float v) const {
        float* q = dx + base + (size_t)m * HW;
        *q = accumulate ? (*q + v) : v;
    }
};
struct SmSliceEpi {  // part[slice][m][n = img * HW + pix] = acc
    typedef int St;
    float* ws;
    int M, Np, slice;
    __device__ __forceinline__ St col(int p) const { return p; }
    __device__ __forceinline__ void put(St n, int m, float v) const { ws[((size_t)slice * M + m) * Np + n] = v; }
};

template <int WM, int WN, bool REFLECT, bool REV, class E, int TAPS>
const char* p9sm_tag() { return __PRETTY_FUNCTION__; }

template <bool REFLECT, bool REV, class E, int TAPS>
void launch(const JpP9smPlan& p, const float* wp, const float* x, E e, int rows, int red, int N, int H, int W, const JpCall& st) {
    constexpr int KGS = TAPS == 9 ? 1 : 2;
    const unsigned* wq = reinterpret_cast<const unsigned*>(wp);
    // executed FLOPs (6 bf16 products per fp32 product) of the tiles as launched, padding included
    const double px = (double)N * jp_cdiv(H, p.tr) * p.tr * jp_cdiv(W, 32) * 32;
    const float* xam = JP_NS == 2 ? jp_amax_of(x, (long)N * red * H * W, st) : nullptr;
    jp_prof_before(p.bmt == 64 ? p9sm_tag<1, 4, REFLECT, REV, E, TAPS>() : p9sm_tag<2, 2, REFLECT, REV, E, TAPS>(),
                   (JP_NS == 2 ? 3.0 : 6.0) * 2.0 * rows * px * TAPS * red, st);
    const dim3 grid(N * jp_cdiv(H, p.tr) * jp_cdiv(W, 32), jp_cdiv(rows, p.bmt), p.splits);
    if (p.bmt == 64)
        hipLaunchKernelGGL((jp_igemm_p9sm_kernel<1, 4, 2, REFLECT, REV, E, TAPS, KGS>), grid, dim3(256), 0, st, wq, x, e, rows, red, p.nst,
                           H, W, p.sps, xam);
    else
        hipLaunchKernelGGL((jp_igemm_p9sm_kernel<2, 2, 2, REFLECT, REV, E, TAPS, KGS>), grid, dim3(256), 0, st, wq, x, e, rows, red, p.nst,
                           H, W, p.sps, xam);
    jp_prof_after(st);
}

template <bool REFLECT, bool REV, int TAPS>
void launch_epi(const JpP9smPlan& p, const float* wp, const float* x, float* out, const float* bias, int act, int accumulate,
                float* part, int rows, int red, int N, int H, int W, const JpCall& st) {
    if (p.splits > 1) {
        SmSliceEpi e{part, rows, N * H * W, 0};
        launch<REFLECT, REV, SmSliceEpi, TAPS>(p, wp, x, e, rows, red, N, H, W, st);
    } else if (REV) {
        SmDgradEpi e{out, rows, H * W, accumulate, 0};
        launch<REFLECT, REV, SmDgradEpi, TAPS>(p, wp, x, e, rows, red, N, H, W, st);
    } else {
        SmFwdEpi e{out, bias, rows, H * W, act, 0};
        launch<REFLECT, REV, SmFwdEpi, TAPS>(p, wp, x, e, rows, red, N, H, W, st);
    }
}

}  // namespace

bool jp_p9sm_plan(int rows, int red, int N, int H, int W, int khw, JpP9smPlan* p) {
    static const bool on = [] {
        const char* e = getenv("JP_P9SM");
        const char* s = getenv("JP_P9S");
        return !(e && e[0] == '0') && !(s && s[0] == '0');
    }();
    const int kgs = khw == 9 ? 1 : 2;
    if (!on || (khw != 9 && khw != 1) || rows < 32 || red < 32 || red % (16 * kgs) != 0 || H < 2 || W < 2) return false;
    if ((long)red * H * W * 4 >= (1L << 31) || (long)N * H * W >= (1L << 30)) return false;
    p->bmt = rows <= 64 ? 64 : 128;
    p->tr = rows <= 64 ? 8 : 4;
    p->nst = red / (16 * kgs);
    const long ptiles = (long)N * jp_cdiv(H, p->tr) * jp_cdiv(W, 32);
    const long tiles = ptiles * jp_cdiv(rows, p->bmt);
    // at least a third of the staged / multiplied pixels must be real ones (6x20 maps on 8x32 tiles: 47 %)
    if (3L * N * H * W < ptiles * p->tr * 32) return false;
    int sp = 1;
    if (tiles < 192) sp = (int)std::max<long>(1, std::min<long>(std::min<long>(jp_cdiv(384, tiles), p->nst / 2), 16));
    p->sps = jp_cdiv(p->nst, sp);
    p->splits = jp_cdiv(p->nst, p->sps);
    p->part_floats = p->splits > 1 ? (long)p->splits * rows * N * H * W : 0;
    return true;
}

// forward (rev = 0: y = act(conv(x) + bias)) or dgrad main pass (rev = 1: taps mirrored, zero fill, dx (= or +=)); with
// plan.splits > 1 the partial tiles go to `part` (plan.part_floats floats) and the CALLER folds them.
void jp_p9sm_launch(const JpP9smPlan& p, const float* wp, const float* x, float* out, const float* bias, int act, int accumulate,
                    float* part, int rows, int red, int N, int H, int W, int khw, int reflect, int rev, const JpCall& st) {
    if (khw == 1 && rev && p.splits == 1) {
        SmDgradEpi e{out, rows, H * W, accumulate, 0};
        launch<false, true, SmDgradEpi, 1>(p, wp, x, e, rows, red, N, H, W, st);
    } else if (khw == 1) launch_epi<false, false, 1>(p, wp, x, out, bias, act, accumulate, part, rows, red, N, H, W, st);
    else if (rev) launch_epi<false, true, 9>(p, wp, x, out, bias, act, accumulate, part, rows, red, N, H, W, st);
    else if (reflect) launch_epi<true, false, 9>(p, wp, x, out, bias, act, accumulate, part, rows, red, N, H, W, st);
    else launch_epi<false, false, 9>(p, wp, x, out, bias, act, accumulate, part, rows, red, N, H, W, st);
}

// 2: the patch kernels of this build form fp32 products from two fp16 splits per operand (three products, operand scales from
// jp_amax / scale.hip); 3: from three bf16 splits (six products).  The host mirror uses it to decide whether to hand amax hints.
extern "C" int jp_split_scheme(void) { return JP_NS; }
