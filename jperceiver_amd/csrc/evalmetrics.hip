// Evaluation metrics of the reference as GPU reductions:
//   layout : confusion counts of argmax(topview logits) vs the BEV label  -> mean_IU / mean_precision
//            (mono/core/evaluation/pixel_error.py:59-118, eval_hooks.py:181-199)
//   depth  : Garg-crop + range mask, median scaling, clamp, the 7 error metrics
//            (pixel_error.py:27-40, eval_hooks.py:147-179)
// All host logic (class presence rules, averaging) stays in jperceiver_amd/core/evaluation.py.
#include "jp_common.h"
#include <algorithm>

namespace {
constexpr int TPB = 256;

// counts[b][2*p + t] += 1 for every pixel with prediction p = argmax(logits[b][:, pix]) (ties -> 0, like torch.argmax's
// first maximum) and label t = (gt != 0)
__global__ __launch_bounds__(TPB) void confusion2_kernel(const float* __restrict__ logits, const float* __restrict__ gt,
                                                         double* __restrict__ counts, int HW) {
    __shared__ double sm[4];
    const int b = blockIdx.y;
    const float* l0 = logits + (size_t)b * 2 * HW;
    const float* l1 = l0 + HW;
    const float* g = gt + (size_t)b * HW;
    int c[4] = {0, 0, 0, 0};
    for (int i = blockIdx.x * TPB + threadIdx.x; i < HW; i += gridDim.x * TPB) {
        const int p = l1[i] > l0[i] ? 1 : 0, t = g[i] != 0.f ? 1 : 0;
        c[2 * p + t] += 1;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double s = jp_block_sum_d((double)c[k], sm);
        if (threadIdx.x == 0 && s != 0.0) atomicAdd(&counts[b * 4 + k], s);   // integer-valued doubles: order-independent
    }
}

// pred_depth = 1 / resized scaled disparity; valid = MIN < gt < MAX inside the Garg crop (eval_hooks.py:160-166)
__global__ __launch_bounds__(TPB) void depth_prepare_kernel(const float* __restrict__ disp_resized, const float* __restrict__ gt,
                                                            float* __restrict__ pred, uint8_t* __restrict__ valid, int H, int W,
                                                            int y0, int y1, int x0, int x1, float dmin, float dmax) {
    for (int i = blockIdx.x * TPB + threadIdx.x; i < H * W; i += gridDim.x * TPB) {
        const int y = i / W, x = i - y * W;
        const float g = gt[i];
        pred[i] = 1.f / disp_resized[i];
        valid[i] = (g > dmin && g < dmax && y >= y0 && y < y1 && x >= x0 && x < x1) ? 1 : 0;
    }
}

__device__ __forceinline__ unsigned ord_key(float f) {      // monotone float -> uint map
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_val(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// np.median over the valid elements: out[0] = count, out[1] = median (mean of the two central order statistics for an
// even count).  One workgroup, bit-wise bisection on the ordered key: 32 counting passes per order statistic.
__global__ __launch_bounds__(1024) void masked_median_kernel(const float* __restrict__ x, const uint8_t* __restrict__ valid,
                                                             int n, float* __restrict__ out) {
    __shared__ unsigned cnt;
    __shared__ unsigned total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    unsigned mine = 0;
    for (int i = threadIdx.x; i < n; i += 1024) mine += valid[i];
    atomicAdd(&total, mine);
    __syncthreads();
    const unsigned m = total;
    if (m == 0) {
        if (threadIdx.x == 0) { out[0] = 0.f; out[1] = __uint_as_float(0x7fc00000u); }
        return;
    }
    float res[2];
    const unsigned ks[2] = {(m - 1) / 2, m / 2};
    for (int which = 0; which < 2; ++which) {
        if (which == 1 && ks[1] == ks[0]) { res[1] = res[0]; break; }
        unsigned prefix = 0;                    // smallest key K with #(key <= K) >= k + 1, built from the top bit down
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned cand = prefix | ((1u << bit) - 1u);      // largest key with this prefix and a 0 at `bit`
            __syncthreads();
            if (threadIdx.x == 0) cnt = 0;
            __syncthreads();
            unsigned c = 0;
            for (int i = threadIdx.x; i < n; i += 1024)
                if (valid[i] && ord_key(x[i]) <= cand) ++c;
            atomicAdd(&cnt, c);
            __syncthreads();
            if (cnt < ks[which] + 1) prefix |= 1u << bit;
        }
        res[which] = ord_val(prefix);
    }
    if (threadIdx.x == 0) { out[0] = (float)m; out[1] = 0.5f * (res[0] + res[1]); }
}

// sums[0..6] = a1, a2, a3 counts, sum (gt-p)^2, sum (log gt - log p)^2, sum |gt-p|/gt, sum (gt-p)^2/gt; sums[7] = count
// with p = clamp(pred * ratio, dmin, dmax), ratio = med[1] (gt) / med[3] (pred) unless fixed_scale > 0
__global__ __launch_bounds__(TPB) void depth_errors_kernel(const float* __restrict__ gt, const float* __restrict__ pred,
                                                           const uint8_t* __restrict__ valid, int n,
                                                           const float* __restrict__ med_gt, const float* __restrict__ med_pred,
                                                           float fixed_scale, float dmin, float dmax, double* __restrict__ sums) {
    __shared__ double sm[4];
    const float ratio = fixed_scale > 0.f ? fixed_scale : med_gt[1] / med_pred[1];
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
        if (!valid[i]) continue;
        const float g = gt[i];
        const float p = fminf(fmaxf(pred[i] * ratio, dmin), dmax);
        const float th = fmaxf(g / p, p / g), d = g - p, lg = logf(g) - logf(p);
        acc[0] += th < 1.25f;
        acc[1] += th < 1.25f * 1.25f;
        acc[2] += th < 1.25f * 1.25f * 1.25f;
        acc[3] += (double)d * d;
        acc[4] += (double)lg * lg;
        acc[5] += fabsf(d) / g;
        acc[6] += (double)d * d / g;
        acc[7] += 1.0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double s = jp_block_sum_d(acc[k], sm);
        if (threadIdx.x == 0) atomicAdd(&sums[k], s);
    }
}
}  // namespace

#define JP_ST hipStream_t st = (hipStream_t)stream

extern "C" int jp_confusion2(const float* logits, const float* gt, double* counts, int B, int HW, void* stream) {
    JP_CHECK_ARG(logits && gt && counts && B > 0 && HW > 0, "confusion2: bad args");
    JP_ST;
    JP_HIP(hipMemsetAsync(counts, 0, sizeof(double) * 4 * B, st));
    hipLaunchKernelGGL(confusion2_kernel, dim3(std::min(jp_cdiv(HW, TPB), 64), B), dim3(TPB), 0, st, logits, gt, counts, HW);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_depth_eval_prepare(const float* disp_resized, const float* gt, float* pred, uint8_t* valid, int H, int W,
                                     int y0, int y1, int x0, int x1, float dmin, float dmax, void* stream) {
    JP_CHECK_ARG(disp_resized && gt && pred && valid && H > 0 && W > 0, "depth_eval_prepare: bad args");
    JP_ST;
    hipLaunchKernelGGL(depth_prepare_kernel, dim3(std::min(jp_cdiv(H * W, TPB), 2048)), dim3(TPB), 0, st, disp_resized, gt, pred,
                       valid, H, W, y0, y1, x0, x1, dmin, dmax);
    JP_LAUNCH_CHECK();
}

// out: 2 floats {count, median}
extern "C" int jp_masked_median(const float* x, const uint8_t* valid, int n, float* out, void* stream) {
    JP_CHECK_ARG(x && valid && out && n > 0, "masked_median: bad args");
    JP_ST;
    hipLaunchKernelGGL(masked_median_kernel, dim3(1), dim3(1024), 0, st, x, valid, n, out);
    JP_LAUNCH_CHECK();
}

// sums: 8 doubles (zeroed here)
extern "C" int jp_depth_errors(const float* gt, const float* pred, const uint8_t* valid, int n, const float* med_gt,
                               const float* med_pred, float fixed_scale, float dmin, float dmax, double* sums, void* stream) {
    JP_CHECK_ARG(gt && pred && valid && sums && n > 0 && (fixed_scale > 0.f || (med_gt && med_pred)), "depth_errors: bad args");
    JP_ST;
    JP_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 8, st));
    hipLaunchKernelGGL(depth_errors_kernel, dim3(std::min(jp_cdiv(n, TPB), 1024)), dim3(TPB), 0, st, gt, pred, valid, n, med_gt,
                       med_pred, fixed_scale, dmin, dmax, sums);
    JP_LAUNCH_CHECK();
}
