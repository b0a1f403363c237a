// Tiny dense pieces of the BEV-layout branch: the CVP MLP (CycledViewProjection.py:33-38,54-67:
// Linear(64,64)+ReLU twice over (B*128) rows) and the CCT attention algebra
// (CrossViewTransformer.py:53-65,77-88: K^T Q bmm, column max/argmax, gather, and the per-channel
// 8x8 matrix product).  All of it is < 3 MMAC per image, i.e. launch-latency class work: one
// strided-batched LDS-tiled fp32 GEMM plus index kernels, no MFMA (tiles are 8..64 wide).
#include "jp_common.h"
#include <algorithm>

namespace {

constexpr int TS = 16;

// C[b] = alpha * op(A[b]) * op(B[b]) + beta * C[b];  row-major, op = transpose when tA/tB.
// A is (M x K) after op, B is (K x N) after op.
__global__ __launch_bounds__(TS * TS) void gemm_sb_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                          float* __restrict__ C, int M, int N, int K, int lda,
                                                          int ldb, int ldc, long sA, long sB, long sC, int tA,
                                                          int tB, float alpha, float beta) {
    __shared__ float As[TS][TS + 1];
    __shared__ float Bs[TS][TS + 1];
    const int b = blockIdx.z;
    A += b * sA; B += b * sB; C += b * sC;
    const int tx = threadIdx.x % TS, ty = threadIdx.x / TS;
    const int row = blockIdx.y * TS + ty, col = blockIdx.x * TS + tx;
    float acc = 0.f;
    for (int k0 = 0; k0 < K; k0 += TS) {
        const int ka = k0 + tx, kb = k0 + ty;
        As[ty][tx] = (row < M && ka < K) ? (tA ? A[(size_t)ka * lda + row] : A[(size_t)row * lda + ka]) : 0.f;
        Bs[ty][tx] = (kb < K && col < N) ? (tB ? B[(size_t)col * ldb + kb] : B[(size_t)kb * ldb + col]) : 0.f;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TS; ++k) acc = fmaf(As[ty][k], Bs[k][tx], acc);
        __syncthreads();
    }
    if (row < M && col < N) {
        float* c = C + (size_t)row * ldc + col;
        *c = alpha * acc + (beta != 0.f ? beta * *c : 0.f);
    }
}

// y[r][c] = act(y[r][c] + bias[c])
__global__ void bias_act_rows_kernel(float* __restrict__ y, const float* __restrict__ bias, long total, int N,
                                     int act) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        y[i] = jp_act(y[i] + (bias ? bias[i % N] : 0.f), act);
}

// out[c] (+)= sum_r x[r][c]: 64 columns x 16 row lanes per workgroup (fixed summation order: deterministic), so a tall
// matrix (the linear layers' bias gradient: M = B*C rows) is not one thread's serial loop over M rows
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int N,
                                                      int accumulate) {
    __shared__ float part[16][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (c < N)
        for (int r = ty; r < M; r += 16) s += x[(size_t)r * N + c];
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < N) {
#pragma unroll
        for (int k = 1; k < 16; ++k) s += part[k][tx];
        out[c] = accumulate ? out[c] + s : s;
    }
}

// E (B, R, Cn): val[b][j] = max_i E[b][i][j], arg = first argmax (torch.max(dim=1) on CPU/GPU returns
// an index of a maximal element; ties are measure-zero on real activations)
__global__ void colmax_kernel(const float* __restrict__ E, float* __restrict__ val, int64_t* __restrict__ arg,
                              int B, int R, int Cn) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * Cn) return;
    const int b = t / Cn, j = t % Cn;
    const float* e = E + (size_t)b * R * Cn + j;
    float best = e[0];
    int bi = 0;
    for (int i = 1; i < R; ++i) {
        const float v = e[(size_t)i * Cn];
        if (v > best) { best = v; bi = i; }
    }
    val[t] = best;
    if (arg) arg[t] = bi;
}

// dE = 0 except dE[b][arg[b][j]][j] = dval[b][j]
__global__ void colmax_bwd_kernel(const float* __restrict__ dval, const int64_t* __restrict__ arg,
                                  float* __restrict__ dE, int B, int R, int Cn) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * R * Cn) return;
    const int j = (int)(t % Cn);
    const int i = (int)((t / Cn) % R);
    const int b = (int)(t / ((long)R * Cn));
    dE[t] = (arg[b * Cn + j] == i) ? dval[b * Cn + j] : 0.f;
}

// T[b][c][j] = V[b][c][arg[b][j]]   (feature_selection, CrossViewTransformer.py:14-24)
__global__ void gather_cols_kernel(const float* __restrict__ V, const int64_t* __restrict__ arg,
                                   float* __restrict__ T, int B, int C, int Nn) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * C * Nn) return;
    const int j = (int)(t % Nn);
    const long bc = t / Nn;
    const int b = (int)(bc / C);
    T[t] = V[bc * Nn + arg[b * Nn + j]];
}

// dV[b][c][i] = sum_{j: arg[b][j]==i} dT[b][c][j]
__global__ void gather_cols_bwd_kernel(const float* __restrict__ dT, const int64_t* __restrict__ arg,
                                       float* __restrict__ dV, int B, int C, int Nn) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * C * Nn) return;
    const int i = (int)(t % Nn);
    const long bc = t / Nn;
    const int b = (int)(bc / C);
    float s = 0.f;
    for (int j = 0; j < Nn; ++j)
        if (arg[b * Nn + j] == i) s += dT[bc * Nn + j];
    dV[t] = s;
}

// spatial mean of (B, C, HW) -> (B, C) scaled (PoseDecoder out.mean(3).mean(2)*0.01, pose_decoder.py:22-23)
__global__ void spatial_mean_kernel(const float* __restrict__ x, float* __restrict__ out, int BC, int HW,
                                    float scale) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= BC) return;
    float s = 0.f;
    for (int i = 0; i < HW; ++i) s += x[(size_t)t * HW + i];
    out[t] = s * scale / (float)HW;
}
__global__ void spatial_mean_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, long total, int HW,
                                        float scale) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    dx[t] = dout[t / HW] * scale / (float)HW;
}

// out[b,c,i,j] = sum_k attn[b,i,k] * V[b,c,k,j]   — the broadcast `attn @ proj_value_depth` of
// CrossViewTransformer.py:88 ((B,1,n,n) @ (B,C,n,n): a true n x n matrix product per channel)
__global__ void bcast_matmul_fwd_kernel(const float* __restrict__ attn, const float* __restrict__ V,
                                        float* __restrict__ out, long total, int C, int n) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int j = (int)(t % n);
    const int i = (int)((t / n) % n);
    const long bc = t / ((long)n * n);
    const long b = bc / C;
    const float* a = attn + (b * n + i) * n;
    const float* v = V + bc * n * n + j;
    float s = 0.f;
    for (int k = 0; k < n; ++k) s += a[k] * v[(long)k * n];
    out[t] = s;
}
// dV[b,c,k,j] = sum_i attn[b,i,k] * dOut[b,c,i,j]
__global__ void bcast_matmul_bwd_v_kernel(const float* __restrict__ attn, const float* __restrict__ dOut,
                                          float* __restrict__ dV, long total, int C, int n) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int j = (int)(t % n);
    const int k = (int)((t / n) % n);
    const long bc = t / ((long)n * n);
    const long b = bc / C;
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += attn[(b * n + i) * n + k] * dOut[(bc * n + i) * n + j];
    dV[t] = s;
}
// dattn[b,i,k] = sum_c sum_j dOut[b,c,i,j] * V[b,c,k,j]
__global__ void bcast_matmul_bwd_a_kernel(const float* __restrict__ dOut, const float* __restrict__ V,
                                          float* __restrict__ dattn, long total, int C, int n) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int k = (int)(t % n);
    const int i = (int)((t / n) % n);
    const long b = t / ((long)n * n);
    float s = 0.f;
    for (int c = 0; c < C; ++c) {
        const float* d = dOut + ((b * C + c) * n + i) * n;
        const float* v = V + ((b * C + c) * n + k) * n;
        for (int j = 0; j < n; ++j) s += d[j] * v[j];
    }
    dattn[t] = s;
}

// out = scale * acc[0] / acc[1]   (masked mean; 0/0 -> NaN exactly like torch.mean of an empty selection)
__global__ void ratio_finalize_kernel(const double* __restrict__ acc, float* __restrict__ out, double scale) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(scale * acc[0] / acc[1]);
}

}  // namespace

#define JP_ST hipStream_t st = (hipStream_t)stream

extern "C" int jp_gemm_strided_batched(const float* A, const float* B, float* C, int M, int N, int K, int lda,
                                       int ldb, int ldc, long strideA, long strideB, long strideC, int batch,
                                       int transA, int transB, float alpha, float beta, void* stream) {
    JP_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0 && batch < 65536, "gemm: bad args");
    JP_ST;
    dim3 grid(jp_cdiv(N, TS), jp_cdiv(M, TS), batch);
    hipLaunchKernelGGL(gemm_sb_kernel, grid, dim3(TS * TS), 0, st, A, B, C, M, N, K, lda, ldb, ldc, strideA, strideB,
                       strideC, transA, transB, alpha, beta);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_bias_act_rows(float* y, const float* bias, int M, int N, int act, void* stream) {
    JP_CHECK_ARG(y && M > 0 && N > 0, "bias_act_rows: bad args");
    JP_ST;
    const long total = (long)M * N;
    hipLaunchKernelGGL(bias_act_rows_kernel, dim3((int)std::min<long>((total + 255) / 256, 65535)), dim3(256), 0, st,
                       y, bias, total, N, act);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_colsum(const float* x, float* out, int M, int N, int accumulate, void* stream) {
    JP_CHECK_ARG(x && out && M > 0 && N > 0, "colsum: bad args");
    JP_ST;
    hipLaunchKernelGGL(colsum_kernel, dim3(jp_cdiv(N, 64)), dim3(1024), 0, st, x, out, M, N, accumulate);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_colmax(const float* E, float* val, int64_t* arg, int B, int R, int Cn, void* stream) {
    JP_CHECK_ARG(E && val && B > 0 && R > 0 && Cn > 0, "colmax: bad args");
    JP_ST;
    hipLaunchKernelGGL(colmax_kernel, dim3(jp_cdiv(B * Cn, 64)), dim3(64), 0, st, E, val, arg, B, R, Cn);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_colmax_bwd(const float* dval, const int64_t* arg, float* dE, int B, int R, int Cn, void* stream) {
    JP_CHECK_ARG(dval && arg && dE, "colmax_bwd: bad args");
    JP_ST;
    hipLaunchKernelGGL(colmax_bwd_kernel, dim3(jp_cdiv((long)B * R * Cn, 256)), dim3(256), 0, st, dval, arg, dE, B, R, Cn);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_gather_cols(const float* V, const int64_t* arg, float* T, int B, int C, int Nn, void* stream) {
    JP_CHECK_ARG(V && arg && T, "gather_cols: bad args");
    JP_ST;
    hipLaunchKernelGGL(gather_cols_kernel, dim3(jp_cdiv((long)B * C * Nn, 256)), dim3(256), 0, st, V, arg, T, B, C, Nn);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_gather_cols_bwd(const float* dT, const int64_t* arg, float* dV, int B, int C, int Nn,
                                  void* stream) {
    JP_CHECK_ARG(dT && arg && dV, "gather_cols_bwd: bad args");
    JP_ST;
    hipLaunchKernelGGL(gather_cols_bwd_kernel, dim3(jp_cdiv((long)B * C * Nn, 256)), dim3(256), 0, st, dT, arg, dV, B, C,
                       Nn);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_spatial_mean(const float* x, float* out, int BC, int HW, float scale, void* stream) {
    JP_CHECK_ARG(x && out && BC > 0 && HW > 0, "spatial_mean: bad args");
    JP_ST;
    hipLaunchKernelGGL(spatial_mean_kernel, dim3(jp_cdiv(BC, 64)), dim3(64), 0, st, x, out, BC, HW, scale);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_spatial_mean_bwd(const float* dout, float* dx, int BC, int HW, float scale, void* stream) {
    JP_CHECK_ARG(dout && dx && BC > 0 && HW > 0, "spatial_mean_bwd: bad args");
    JP_ST;
    const long total = (long)BC * HW;
    hipLaunchKernelGGL(spatial_mean_bwd_kernel, dim3(jp_cdiv(total, 256)), dim3(256), 0, st, dout, dx, total, HW, scale);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_bcast_matmul_fwd(const float* attn, const float* V, float* out, int B, int C, int n, void* stream) {
    JP_CHECK_ARG(attn && V && out && B > 0 && C > 0 && n > 0, "bcast_matmul_fwd: bad args");
    JP_ST;
    const long total = (long)B * C * n * n;
    hipLaunchKernelGGL(bcast_matmul_fwd_kernel, dim3(jp_cdiv(total, 256)), dim3(256), 0, st, attn, V, out, total, C, n);
    JP_LAUNCH_CHECK();
}

extern "C" int jp_bcast_matmul_bwd(const float* attn, const float* V, const float* dOut, float* dattn, float* dV,
                                   int B, int C, int n, void* stream) {
    JP_CHECK_ARG(attn && V && dOut && B > 0 && C > 0 && n > 0, "bcast_matmul_bwd: bad args");
    JP_ST;
    if (dV) {
        const long total = (long)B * C * n * n;
        hipLaunchKernelGGL(bcast_matmul_bwd_v_kernel, dim3(jp_cdiv(total, 256)), dim3(256), 0, st, attn, dOut, dV, total, C, n);
    }
    if (dattn) {
        const long total = (long)B * n * n;
        hipLaunchKernelGGL(bcast_matmul_bwd_a_kernel, dim3(jp_cdiv(total, 64)), dim3(64), 0, st, dOut, V, dattn, total, C, n);
    }
    JP_LAUNCH_CHECK();
}

extern "C" int jp_ratio_finalize(const double* acc, float* out, double scale, void* stream) {
    JP_CHECK_ARG(acc && out, "ratio_finalize: bad args");
    JP_ST;
    hipLaunchKernelGGL(ratio_finalize_kernel, dim3(1), dim3(64), 0, st, acc, out, scale);
    JP_LAUNCH_CHECK();
}
