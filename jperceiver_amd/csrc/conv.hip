// conv2d forward / dgrad / wgrad for the JPerceiver train step as instances of the fp32-MFMA
// implicit-GEMM engine (igemm.h).  Replaces every nn.Conv2d / ReflectionPad2d+Conv2d call site of
// the reference hot path (resnet.py:6-13,91; layers.py:147-167; depth_decoder.py:15-39;
// pose_decoder.py:9-12; layout_model.py:31-47,138-153; CrossViewTransformer.py:30-42).
//
// Layout: activations NCHW fp32, weights [Cout][Cin][KH][KW] fp32 (the reference's state-dict
// layout, so checkpoints stay interchangeable).
//   forward : M = Cout,  N = batch*OH*OW pixels,  K = Cin*KH*KW   (B gathered with zero/reflect
//             padding, stride, and optional fused nearest-2x-upsample + channel-concat sources)
//   dgrad   : M = Cin,   N = batch*H*W  pixels,   K = Cout*KH*KW  (B gathers dY; the adjoint of
//             reflection padding is folded into the gather, no workspace)
//   wgrad   : M = Cout,  N = Cin*KH*KW,           K = batch*OH*OW (split-K, fp32 atomics)
#include "igemm.h"
#include <algorithm>

namespace {

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

// ---------------------------------------------------------------- forward loaders
struct PixSt {  // decoded output pixel of a conv (shared by the fwd-B and wgrad-B gathers)
    int img, iy0, ix0, valid;
};

struct FwdA {  // A[m=co][k] = W[co*K + k]
    static constexpr bool ALONG_K = true;
    typedef int St;
    const float* w;
    int M, K;
    __device__ __forceinline__ St fix(int k) const { return k; }
    __device__ __forceinline__ float get(St k, int m) const { return (m < M && k < K) ? w[(size_t)m * K + k] : 0.f; }
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

template <int KH>
struct FwdB {  // B[k=(ci,dy,dx)][n=pixel]
    static constexpr bool ALONG_K = false;
    typedef PixSt St;
    Src3 src;
    int K, Npix, OH, OW, stride, pad, reflect;
    __device__ __forceinline__ St fix(int p) const { return conv_pix(p, Npix, OH, OW, stride, pad); }
    __device__ __forceinline__ float get(const St& st, int k) const {
        if (!st.valid || k >= K) return 0.f;
        return conv_gather<KH>(src, st, k, reflect);
    }
};

struct FwdEpi {  // y[img][co][pix] = act(acc + bias[co])
    typedef size_t St;
    float* y;
    const float* bias;
    int Cout, OHW, act;
    __device__ __forceinline__ St col(int p) const {
        int img = p / OHW;
        return (size_t)img * Cout * OHW + (p - img * OHW);
    }
    __device__ __forceinline__ void put(St base, int m, float v) const {
        if (bias) v += bias[m];
        y[base + (size_t)m * OHW] = jp_act(v, act);
    }
};

// ---------------------------------------------------------------- dgrad loaders
template <int KH>
struct DgradA {  // A[m=ci][k=(co,dy,dx)] = W[co][ci][dy][dx]
    static constexpr bool ALONG_K = false;
    typedef int St;
    const float* w;
    int Cin, K;
    __device__ __forceinline__ St fix(int m) const { return m; }
    __device__ __forceinline__ float get(St ci, int k) const {
        if (ci >= Cin || k >= K) return 0.f;
        int co = k / (KH * KH);
        int tap = k - co * (KH * KH);
        return w[((size_t)co * Cin + ci) * (KH * KH) + tap];
    }
};

struct InPixSt {
    int img, y, x, valid;
};

template <int KH>
struct DgradB {  // B[k=(co,dy,dx)][n=input pixel] = sum of dY entries that used x[pixel] through tap
    static constexpr bool ALONG_K = false;
    typedef InPixSt St;
    const float* dy;
    int K, Npix, H, W, Cout, OH, OW, stride, pad, reflect;
    __device__ __forceinline__ St fix(int p) const {
        St st;
        st.valid = p < Npix;
        int hw = H * W;
        st.img = p / hw;
        int pix = p - st.img * hw;
        st.y = pix / W;
        st.x = pix - st.y * W;
        return st;
    }
    __device__ __forceinline__ float get(const St& st, int k) const {
        if (!st.valid || k >= K) return 0.f;
        int co = k / (KH * KH);
        int tap = k - co * (KH * KH);
        int ty = tap / KH, tx = tap - ty * KH;
        const float* d = dy + (size_t)(st.img * Cout + co) * OH * OW;
        const int y = st.y, x = st.x;
        if (!reflect) {
            int ny = y + pad - ty, nx = x + pad - tx;
            if (ny < 0 || nx < 0) return 0.f;
            int oy = ny / stride, ox = nx / stride;
            if (oy * stride != ny || ox * stride != nx || oy >= OH || ox >= OW) return 0.f;
            return d[oy * OW + ox];
        }
        // ReflectionPad2d(1) + 3x3 stride-1 conv (OH==H, OW==W).  Padded row py in [0,H+1] maps to
        // input row reflect(py-1); input row y is hit by py=y+1, and additionally by py=0 when y==1
        // and by py=H+1 when y==H-2.  dY row = py - ty must lie in [0,H).
        int r0 = y + 1 - ty, r1 = -1, r2 = -1, c0 = x + 1 - tx, c1 = -1, c2 = -1;
        if (r0 < 0 || r0 >= H) r0 = -1;
        if (y == 1 && ty == 0) r1 = 0;
        if (y == H - 2 && ty == 2) r2 = H - 1;
        if (c0 < 0 || c0 >= W) c0 = -1;
        if (x == 1 && tx == 0) c1 = 0;
        if (x == W - 2 && tx == 2) c2 = W - 1;
        float s = 0.f;
#define JP_ROWSUM(r)                                  \
        if (r >= 0) {                                     \
            if (c0 >= 0) s += d[r * OW + c0];             \
            if (c1 >= 0) s += d[r * OW + c1];             \
            if (c2 >= 0) s += d[r * OW + c2];             \
        }
        JP_ROWSUM(r0) JP_ROWSUM(r1) JP_ROWSUM(r2)
#undef JP_ROWSUM
        return s;
    }
};

struct DgradEpi {  // dx[img][ci][pix] (= or +=) acc
    typedef size_t St;
    float* dx;
    int Cin, HW, accumulate;
    __device__ __forceinline__ St col(int p) const {
        int img = p / HW;
        return (size_t)img * Cin * HW + (p - img * HW);
    }
    __device__ __forceinline__ void put(St base, int m, float v) const {
        float* q = dx + base + (size_t)m * HW;
        *q = accumulate ? (*q + v) : v;
    }
};

// ---------------------------------------------------------------- wgrad loaders
struct WgradASt {
    size_t base;
    int valid;
};
struct WgradA {  // A[m=co][k=pixel] = dY[img][co][pix]
    static constexpr bool ALONG_K = true;
    typedef WgradASt St;
    const float* dy;
    int Cout, Npix, OHW;
    __device__ __forceinline__ St fix(int p) const {
        St st;
        st.valid = p < Npix;
        int img = p / OHW;
        st.base = (size_t)img * Cout * OHW + (p - img * OHW);
        return st;
    }
    __device__ __forceinline__ float get(const St& st, int m) const {
        return (st.valid && m < Cout) ? dy[st.base + (size_t)m * OHW] : 0.f;
    }
};

template <int KH>
struct WgradB {  // B[k=pixel][n=(ci,dy,dx)] = xpad[img][ci][oy*s+dy][ox*s+dx]
    static constexpr bool ALONG_K = true;
    typedef PixSt St;
    Src3 src;
    int Kw, Npix, OH, OW, stride, pad, reflect;
    __device__ __forceinline__ St fix(int p) const { return conv_pix(p, Npix, OH, OW, stride, pad); }
    __device__ __forceinline__ float get(const St& st, int j) const {
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

constexpr int KC = 32;

template <int WM, int WN, class A, class B, class E>
void launch(A a, B b, E e, int M, int N, int K, int splits, int kps, hipStream_t st) {
    dim3 grid(jp_cdiv(N, 64 * WN), jp_cdiv(M, 64 * WM), splits);
    hipLaunchKernelGGL((jp_igemm_kernel<WM, WN, KC, A, B, E>), grid, dim3(256), 0, st, a, b, e, M, N, K, kps);
}

template <class A, class B, class E>
void launch_auto(A a, B b, E e, int M, int N, int K, int splits, int kps, hipStream_t st) {
    if (M <= 64) launch<1, 4>(a, b, e, M, N, K, splits, kps, st);
    else if (N <= 64) launch<4, 1>(a, b, e, M, N, K, splits, kps, st);
    else launch<2, 2>(a, b, e, M, N, K, splits, kps, st);
}

Src3 make_src(const float* x0, int c0, int up0, const float* x1, int c1, int up1, const float* x2, int c2,
              int up2, int H, int W) {
    Src3 s;
    s.p0 = x0; s.p1 = x1 ? x1 : x0; s.p2 = x2 ? x2 : x0;
    s.e0 = c0; s.e1 = c0 + c1; s.e2 = c0 + c1 + c2;
    s.s0 = up0; s.s1 = up1; s.s2 = up2;
    s.H = H; s.W = W;
    return s;
}

}  // namespace

#define JP_KH_SWITCH(KHV, ...)                                  \
    switch (KHV) {                                              \
        case 1: { constexpr int KH_ = 1; __VA_ARGS__; } break;  \
        case 3: { constexpr int KH_ = 3; __VA_ARGS__; } break;  \
        case 7: { constexpr int KH_ = 7; __VA_ARGS__; } break;  \
        default: jp_set_last_error("conv: kernel size must be 1, 3 or 7"); return JP_EBADARG; \
    }

extern "C" int jp_conv2d_fwd_src3(const float* x0, int c0, int up0, const float* x1, int c1, int up1,
                                  const float* x2, int c2, int up2, const float* w, const float* bias, float* y,
                                  int N, int H, int W, int Cout, int KH, int stride, int pad, int pad_mode, int act,
                                  void* stream) {
    JP_CHECK_ARG(x0 && w && y, "conv2d_fwd: null pointer");
    JP_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cout > 0 && c0 > 0 && stride >= 1, "conv2d_fwd: bad dims");
    JP_CHECK_ARG(!(pad_mode == JP_PAD_REFLECT && (pad >= H || pad >= W)), "conv2d_fwd: reflect pad >= size");
    const int Cin = c0 + c1 + c2;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KH) / stride + 1;
    const long npix = (long)N * OH * OW;
    JP_CHECK_ARG(npix < (1L << 31) && (long)N * Cin * H * W < (1L << 31) * 2, "conv2d_fwd: tensor too large");
    const int K = Cin * KH * KH;
    hipStream_t st = (hipStream_t)stream;
    FwdA a{w, Cout, K};
    FwdEpi e{y, bias, Cout, OH * OW, act};
    JP_KH_SWITCH(KH, {
        FwdB<KH_> b{make_src(x0, c0, up0, x1, c1, up1, x2, c2, up2, H, W), K, (int)npix, OH, OW, stride, pad,
                    pad_mode == JP_PAD_REFLECT};
        launch_auto(a, b, e, Cout, (int)npix, K, 1, jp_cdiv(K, KC) * KC, st);
    });
    JP_LAUNCH_CHECK();
}

extern "C" int jp_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H,
                             int W, int Cout, int KH, int stride, int pad, int pad_mode, int act, void* stream) {
    return jp_conv2d_fwd_src3(x, Cin, 0, nullptr, 0, 0, nullptr, 0, 0, w, bias, y, N, H, W, Cout, KH, stride, pad,
                              pad_mode, act, stream);
}

extern "C" int jp_conv2d_dgrad(const float* dy, const float* w, float* dx, int N, int Cin, int H, int W, int Cout,
                               int KH, int stride, int pad, int pad_mode, int accumulate, void* stream) {
    JP_CHECK_ARG(dy && w && dx, "conv2d_dgrad: null pointer");
    JP_CHECK_ARG(!(pad_mode == JP_PAD_REFLECT && !(KH == 3 && stride == 1 && pad == 1 && H >= 2 && W >= 2)),
                 "conv2d_dgrad: reflect mode supports 3x3 stride 1 pad 1 only");
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KH) / stride + 1;
    const long npix = (long)N * H * W;
    JP_CHECK_ARG(npix < (1L << 31), "conv2d_dgrad: tensor too large");
    const int K = Cout * KH * KH;
    hipStream_t st = (hipStream_t)stream;
    DgradEpi e{dx, Cin, H * W, accumulate};
    JP_KH_SWITCH(KH, {
        DgradA<KH_> a{w, Cin, K};
        DgradB<KH_> b{dy, K, (int)npix, H, W, Cout, OH, OW, stride, pad, pad_mode == JP_PAD_REFLECT};
        launch_auto(a, b, e, Cin, (int)npix, K, 1, jp_cdiv(K, KC) * KC, st);
    });
    JP_LAUNCH_CHECK();
}

extern "C" int jp_conv2d_wgrad_src3(const float* x0, int c0, int up0, const float* x1, int c1, int up1,
                                    const float* x2, int c2, int up2, const float* dy, float* dw, int N, int H, int W,
                                    int Cout, int KH, int stride, int pad, int pad_mode, int accumulate,
                                    void* stream) {
    JP_CHECK_ARG(x0 && dy && dw, "conv2d_wgrad: null pointer");
    const int Cin = c0 + c1 + c2;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KH) / stride + 1;
    const long npix = (long)N * OH * OW;
    JP_CHECK_ARG(npix < (1L << 31), "conv2d_wgrad: tensor too large");
    const int Kw = Cin * KH * KH;
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate) JP_HIP(hipMemsetAsync(dw, 0, sizeof(float) * (size_t)Cout * Kw, st));
    // split-K so that ~2k workgroups are in flight (256 CUs x 8 XCD-interleaved)
    const int tiles = jp_cdiv(Cout, Cout <= 64 ? 64 : 128) * jp_cdiv(Kw, Cout <= 64 ? 256 : 128);
    int splits = (int)std::min<long>(std::max(1, 2048 / std::max(1, tiles)), jp_cdiv(npix, 4 * KC));
    int kps = jp_cdiv(jp_cdiv(npix, splits), KC) * KC;
    splits = jp_cdiv(npix, kps);
    WgradA a{dy, Cout, (int)npix, OH * OW};
    WgradEpi e{dw, Kw};
    JP_KH_SWITCH(KH, {
        WgradB<KH_> b{make_src(x0, c0, up0, x1, c1, up1, x2, c2, up2, H, W), Kw, (int)npix, OH, OW, stride, pad,
                      pad_mode == JP_PAD_REFLECT};
        launch_auto(a, b, e, Cout, Kw, (int)npix, splits, kps, st);
    });
    JP_LAUNCH_CHECK();
}

extern "C" int jp_conv2d_wgrad(const float* x, const float* dy, float* dw, int N, int Cin, int H, int W, int Cout,
                               int KH, int stride, int pad, int pad_mode, int accumulate, void* stream) {
    return jp_conv2d_wgrad_src3(x, Cin, 0, nullptr, 0, 0, nullptr, 0, 0, dy, dw, N, H, W, Cout, KH, stride, pad,
                                pad_mode, accumulate, stream);
}
