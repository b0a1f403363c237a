// Device-side input pipeline: what MonoDataset.preprocess does with PIL on 24 host workers per GPU
// (mono/datasets/mono_dataset.py:126-171,417-431) runs here on the raw uint8 frames after ONE pinned async upload:
//   resample   : PIL Image.resize(..., ANTIALIAS == LANCZOS) on 8-bit images, bit-exact: Pillow's two-pass separable
//                convolution with 22-bit fixed-point coefficients and a uint8 intermediate (libImaging/Resample.c; the
//                coefficient tables are built on the host exactly like precompute_coeffs / normalize_coeffs_8bpc)
//   to_tensor  : HWC uint8 -> CHW float / 255 (fused into the vertical pass)
//   ColorJitter: brightness / contrast / saturation / hue in torchvision's tensor arithmetic, in a random order
//   topview    : L conversion, binarise, NEAREST resize to H/4, {0,1} float  (process_topview)
// Third-party semantics (Pillow, torchvision) are restated from their published algorithms; Pillow IS importable in
// the build image, so the resampler is pinned against it (tests/golden/preprocess.npz).
#include "jp_common.h"
#include <algorithm>

namespace {
constexpr int TPB = 256;
constexpr int PBITS = 22;    // PRECISION_BITS = 32 - 8 - 2

__device__ __forceinline__ int clip8(int v) {
    v >>= PBITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal: src (N*H rows, W, C) u8 -> dst (N*H rows, OW, C) u8
__global__ __launch_bounds__(TPB) void resample_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk,
                                                         long rows, int W, int OW, int C, int ksize) {
    const long total = rows * OW * C;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
        const int c = (int)(i % C);
        const long t = i / C;
        const int ox = (int)(t % OW);
        const long row = t / OW;
        const int xmin = bounds[2 * ox], xmax = bounds[2 * ox + 1];
        const int* k = kk + (long)ox * ksize;
        const uint8_t* sp = src + (row * W + xmin) * C + c;
        int ss = 1 << (PBITS - 1);
        for (int x = 0; x < xmax; ++x) ss += (int)sp[(long)x * C] * k[x];
        dst[i] = (uint8_t)clip8(ss);
    }
}

// vertical: src (N, H, OW, C) u8 -> out_u8 (N, OH, OW, C) and / or out_f (N, C, OH, OW) = value / 255
__global__ __launch_bounds__(TPB) void resample_v_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ out_u8,
                                                         float* __restrict__ out_f, const int* __restrict__ bounds,
                                                         const int* __restrict__ kk, int N, int H, int OH, int OW, int C,
                                                         int ksize) {
    const long total = (long)N * OH * OW * C;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
        const int c = (int)(i % C);
        long t = i / C;
        const int ox = (int)(t % OW);
        t /= OW;
        const int oy = (int)(t % OH);
        const int n = (int)(t / OH);
        const int ymin = bounds[2 * oy], ymax = bounds[2 * oy + 1];
        const int* k = kk + (long)oy * ksize;
        const uint8_t* sp = src + (((long)n * H + ymin) * OW + ox) * C + c;
        int ss = 1 << (PBITS - 1);
        for (int y = 0; y < ymax; ++y) ss += (int)sp[(long)y * OW * C] * k[y];
        const int v = clip8(ss);
        if (out_u8) out_u8[i] = (uint8_t)v;
        if (out_f) out_f[(((long)n * C + c) * OH + oy) * OW + ox] = (float)v / 255.f;      // transforms.ToTensor
    }
}

// HWC u8 -> CHW float / 255 (ToTensor without a resize)
__global__ __launch_bounds__(TPB) void to_tensor_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int N, int H,
                                                        int W, int C) {
    const long total = (long)N * H * W * C;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
        const int c = (int)(i % C);
        long t = i / C;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        dst[(((long)n * C + c) * H + y) * W + x] = (float)src[i] / 255.f;
    }
}

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float gray_of(float r, float g, float b) { return 0.2989f * r + 0.587f * g + 0.114f * b; }

// per-image mean of the grayscale image (torchvision adjust_contrast): sums[n] = sum gray
__global__ __launch_bounds__(TPB) void gray_sum_kernel(const float* __restrict__ x, double* __restrict__ sums, int HW) {
    __shared__ double sm[4];
    const int n = blockIdx.y;
    const float* r = x + (size_t)n * 3 * HW;
    double s = 0.0;
    for (int i = blockIdx.x * TPB + threadIdx.x; i < HW; i += gridDim.x * TPB) s += gray_of(r[i], r[HW + i], r[2 * HW + i]);
    s = jp_block_sum_d(s, sm);
    if (threadIdx.x == 0) atomicAdd(&sums[n], s);
}

// one ColorJitter op on (N,3,H,W) floats in place.  op: 0 brightness, 1 contrast (sums = per-image gray sums),
// 2 saturation, 3 hue.  torchvision.transforms.functional (tensor path): _blend(img, other, f) = clamp(f*img + (1-f)*other)
__global__ __launch_bounds__(TPB) void color_jitter_kernel(float* __restrict__ x, const double* __restrict__ sums, int N, int HW,
                                                           int op, float f) {
    const long total = (long)N * HW;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
        const int n = (int)(i / HW), p = (int)(i - (long)n * HW);
        float* px = x + (size_t)n * 3 * HW + p;
        float r = px[0], g = px[HW], b = px[2 * HW];
        if (op == 0) {
            r = clamp01(r * f); g = clamp01(g * f); b = clamp01(b * f);
        } else if (op == 1) {
            const float m = (float)(sums[n] / (double)HW) * (1.f - f);
            r = clamp01(f * r + m); g = clamp01(f * g + m); b = clamp01(f * b + m);
        } else if (op == 2) {
            const float gr = gray_of(r, g, b) * (1.f - f);
            r = clamp01(f * r + gr); g = clamp01(f * g + gr); b = clamp01(f * b + gr);
        } else {
            // _rgb2hsv
            const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
            const bool eqc = maxc == minc;
            const float cr = maxc - minc;
            const float s = cr / (eqc ? 1.f : maxc);
            const float crd = eqc ? 1.f : cr;
            const float rc = (maxc - r) / crd, gc = (maxc - g) / crd, bc = (maxc - b) / crd;
            const float hr = (maxc == r) ? (bc - gc) : 0.f;
            const float hg = ((maxc == g) && (maxc != r)) ? (2.f + rc - bc) : 0.f;
            const float hb = ((maxc != g) && (maxc != r)) ? (4.f + gc - rc) : 0.f;
            float h = fmodf(hr + hg + hb, 6.f) / 6.f + 1.f;          // torch.fmod keeps the dividend's sign
            h = fmodf(h, 1.f);
            h = fmodf(h + f, 1.f);
            if (h < 0.f) h += 1.f;                                    // python-style % on the shifted hue
            // _hsv2rgb
            const float v = maxc;
            const float h6 = h * 6.f;
            const float fl = floorf(h6);
            const float ff = h6 - fl;
            const int i6 = ((int)fl) % 6;
            const float p_ = clamp01(v * (1.f - s)), q_ = clamp01(v * (1.f - s * ff)), t_ = clamp01(v * (1.f - s * (1.f - ff)));
            switch (i6) {
                case 0: r = v; g = t_; b = p_; break;
                case 1: r = q_; g = v; b = p_; break;
                case 2: r = p_; g = v; b = t_; break;
                case 3: r = p_; g = q_; b = v; break;
                case 4: r = t_; g = p_; b = v; break;
                default: r = v; g = p_; b = q_; break;
            }
        }
        px[0] = r; px[HW] = g; px[2 * HW] = b;
    }
}

// BEV label: src (N, h, w, C) u8 (C = 1 or 3) -> dst (N, 1, S, S) float {0,1}: L = ITU-R 601 luma, NEAREST pick at
// floor((o + 0.5) * scale), then `>= thresh` (convert("1") without dithering: exact for the binary label images) or
// `== 255` (process_topview_both)
__global__ __launch_bounds__(TPB) void topview_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int N, int h, int w,
                                                      int C, int S, int exact255) {
    const long total = (long)N * S * S;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
        const int ox = (int)(i % S);
        const long t = i / S;
        const int oy = (int)(t % S), n = (int)(t / S);
        const int sx = min(w - 1, (int)(((double)ox + 0.5) * (double)w / (double)S));
        const int sy = min(h - 1, (int)(((double)oy + 0.5) * (double)h / (double)S));
        const uint8_t* p = src + (((long)n * h + sy) * w + sx) * C;
        const int L = C == 1 ? (int)p[0] : (int)(((unsigned)p[0] * 19595u + (unsigned)p[1] * 38470u + (unsigned)p[2] * 7471u + 0x8000u) >> 16);
        dst[i] = exact255 ? (L == 255 ? 1.f : 0.f) : (L >= 128 ? 1.f : 0.f);
    }
}

inline int blocks_for(long n) { return (int)std::min<long>((n + TPB - 1) / TPB, 1 << 16); }
}  // namespace

#define JP_ST hipStream_t st = (hipStream_t)stream

// bounds: (out, 2) int32 {first tap, tap count}; kk: (out, ksize) int32 fixed-point coefficients (device copies of the
// host tables).  axis 0: horizontal, src (rows, W, C) -> dst (rows, OW, C)
extern "C" int jp_resample_h_u8(const uint8_t* src, uint8_t* dst, const int* bounds, const int* kk, long rows, int W, int OW,
                                int C, int ksize, void* stream) {
    JP_CHECK_ARG(src && dst && bounds && kk && rows > 0 && W > 0 && OW > 0 && C > 0 && ksize > 0, "resample_h_u8: bad args");
    JP_ST;
    hipLaunchKernelGGL(resample_h_kernel, dim3(blocks_for(rows * OW * C)), dim3(TPB), 0, st, src, dst, bounds, kk, rows, W, OW, C,
                       ksize);
    JP_LAUNCH_CHECK();
}

// vertical pass; writes uint8 HWC (out_u8) and / or float CHW / 255 (out_f = transforms.ToTensor of the resized image)
extern "C" int jp_resample_v_u8(const uint8_t* src, uint8_t* out_u8, float* out_f, const int* bounds, const int* kk, int N,
                                int H, int OH, int OW, int C, int ksize, void* stream) {
    JP_CHECK_ARG(src && (out_u8 || out_f) && bounds && kk && N > 0 && H > 0 && OH > 0 && OW > 0 && C > 0 && ksize > 0,
                 "resample_v_u8: bad args");
    JP_ST;
    hipLaunchKernelGGL(resample_v_kernel, dim3(blocks_for((long)N * OH * OW * C)), dim3(TPB), 0, st, src, out_u8, out_f, bounds,
                       kk, N, H, OH, OW, C, ksize);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_u8_to_tensor(const uint8_t* src, float* dst, int N, int H, int W, int C, void* stream) {
    JP_CHECK_ARG(src && dst && N > 0 && H > 0 && W > 0 && C > 0, "u8_to_tensor: bad args");
    JP_ST;
    hipLaunchKernelGGL(to_tensor_kernel, dim3(blocks_for((long)N * H * W * C)), dim3(TPB), 0, st, src, dst, N, H, W, C);
    JP_LAUNCH_CHECK();
}

// sums: N doubles of caller scratch (only read / written for op == 1)
extern "C" int jp_color_jitter_op(float* x, double* sums, int N, int HW, int op, float factor, void* stream) {
    JP_CHECK_ARG(x && N > 0 && HW > 0 && op >= 0 && op <= 3 && (op != 1 || sums), "color_jitter_op: bad args");
    JP_ST;
    if (op == 1) {
        JP_HIP(hipMemsetAsync(sums, 0, sizeof(double) * N, st));
        hipLaunchKernelGGL(gray_sum_kernel, dim3(std::min(jp_cdiv(HW, TPB), 256), N), dim3(TPB), 0, st, x, sums, HW);
    }
    hipLaunchKernelGGL(color_jitter_kernel, dim3(blocks_for((long)N * HW)), dim3(TPB), 0, st, x, sums, N, HW, op, factor);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_topview_u8(const uint8_t* src, float* dst, int N, int h, int w, int C, int S, int exact255, void* stream) {
    JP_CHECK_ARG(src && dst && N > 0 && h > 0 && w > 0 && (C == 1 || C == 3) && S > 0, "topview_u8: bad args");
    JP_ST;
    hipLaunchKernelGGL(topview_kernel, dim3(blocks_for((long)N * S * S)), dim3(TPB), 0, st, src, dst, N, h, w, C, S, exact255);
    JP_LAUNCH_CHECK();
}
