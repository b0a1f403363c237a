// Microbenchmark (GPU-box aid): what fraction of the 157.3 TF fp32-MFMA peak does a bare
// v_mfma_f32_32x32x2_f32 stream sustain on MI355X, alone / with the LDS operand reads / with barriers?
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_peak.hip -o gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 4/5: the full igemm loop shape -- 32 global dword loads per thread per chunk prefetched into registers
// under the MFMAs, stored to LDS between two barriers.  4 = every block streams its own rows (L2/HBM), 5 = all
// blocks re-read the same 32 KB (L1/L2-hot).
template <int MODE>
__global__ __launch_bounds__(256) void kg(float* out, const float* __restrict__ src, int iters, size_t span) {
    __shared__ float As[32 * 129], Bs[32 * 129];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* ap = As + (lane >> 5) * 129 + (wave >> 1) * 64 + (lane & 31);
    const float* bp = Bs + (lane >> 5) * 129 + (wave & 1) * 64 + (lane & 31);
    float ra[16], rb[16];
    float4 ra4[4];
    auto gload = [&](int it) {
        const float* q = src + (MODE == 4 ? ((size_t)blockIdx.x * 8192 + (size_t)it * 8192 * 1024) % span : 0);
        if (MODE == 7) {   // A operand: 4 x dwordx4 per thread (pre-packed weights), B: 16 dword gathers
#pragma unroll
            for (int r = 0; r < 4; ++r) ra4[r] = reinterpret_cast<const float4*>(q)[r * 256 + t];
#pragma unroll
            for (int r = 0; r < 16; ++r) rb[r] = q[4096 + r * 256 + t];
        } else if (MODE == 8) {   // B row tile shared by the 3 dx taps: B is gathered on every third chunk only
#pragma unroll
            for (int r = 0; r < 16; ++r) ra[r] = q[r * 256 + t];
            if (it % 3 == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rb[r] = q[4096 + r * 256 + t];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { ra[r] = q[r * 256 + t]; rb[r] = q[4096 + r * 256 + t]; }
        }
    };
    gload(0);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 7) {
#pragma unroll
            for (int r = 0; r < 4; ++r) reinterpret_cast<float4*>(As)[r * 256 + t] = ra4[r];
#pragma unroll
            for (int r = 0; r < 16; ++r) Bs[(r * 2 + (t >> 7)) * 129 + (t & 127)] = rb[r];
        } else if (MODE == 8) {
#pragma unroll
            for (int r = 0; r < 16; ++r) As[(r * 2 + (t >> 7)) * 129 + (t & 127)] = ra[r];
            if (it % 3 == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Bs[(r * 2 + (t >> 7)) * 129 + (t & 127)] = rb[r];
            }
        } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) { As[(r * 2 + (t >> 7)) * 129 + (t & 127)] = ra[r]; Bs[(r * 2 + (t >> 7)) * 129 + (t & 127)] = rb[r]; }
        }
        __syncthreads();
        gload(it + 1);
        float a0 = ap[0], a1 = ap[32], b0 = bp[0], b1 = bp[32];
#pragma unroll 4
        for (int kk = 0; kk < 32; kk += 2) {
            const int kn = kk + 2 < 32 ? kk + 2 : kk;
            const float na0 = ap[kn * 129], na1 = ap[kn * 129 + 32], nb0 = bp[kn * 129], nb1 = bp[kn * 129 + 32];
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
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + t] = s + (MODE == 7 ? ra4[0].x : ra[0]) + rb[0];
}

// MODE 6: same loop, but the staging is LDS-direct (global_load_lds_dword, no VGPR round trip, no ds_write), double
// buffered in LDS: chunk i+1 lands in the other stage while chunk i feeds the MFMAs; one barrier per chunk.
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;
template <int HOT>
__global__ __launch_bounds__(256) void kl(float* out, const float* __restrict__ src, int iters, size_t span) {
    __shared__ float As[2][32 * 129], Bs[2][32 * 129];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto gload = [&](int it, int stage) {
        const float* q = src + (HOT ? 0 : ((size_t)blockIdx.x * 8192 + (size_t)it * 8192 * 1024) % span);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* da = &As[stage][(r * 2 + (wave >> 1)) * 129 + (wave & 1) * 64];
            float* db = &Bs[stage][(r * 2 + (wave >> 1)) * 129 + (wave & 1) * 64];
            __builtin_amdgcn_global_load_lds((glb_ptr)(q + r * 256 + t), (lds_ptr)da, 4, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr)(q + 4096 + r * 256 + t), (lds_ptr)db, 4, 0, 0);
        }
    };
    gload(0, 0);
    for (int it = 0; it < iters; ++it) {
        const int st = it & 1;
        __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): this stage has landed
        __syncthreads();
        gload(it + 1, st ^ 1);
        const float* ap = As[st] + (lane >> 5) * 129 + (wave >> 1) * 64 + (lane & 31);
        const float* bp = Bs[st] + (lane >> 5) * 129 + (wave & 1) * 64 + (lane & 31);
        float a0 = ap[0], a1 = ap[32], b0 = bp[0], b1 = bp[32];
#pragma unroll 4
        for (int kk = 0; kk < 32; kk += 2) {
            const int kn = kk + 2 < 32 ? kk + 2 : kk;
            const float na0 = ap[kn * 129], na1 = ap[kn * 129 + 32], nb0 = bp[kn * 129], nb1 = bp[kn * 129 + 32];
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + t] = s;
}
template <int HOT>
void runl(const char* name, int blocks, int iters, float* out, const float* src, size_t span) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kl<HOT>, dim3(blocks), dim3(256), 0, 0, out, src, 4, span);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kl<HOT>, dim3(blocks), dim3(256), 0, 0, out, src, iters, span);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 64 * 4096.0;
    printf("%-40s blocks=%5d (%.0f/CU)  %.3f ms  %.1f TF  (%.1f%% of 157.3)\n", name, blocks, blocks / 256.0, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 157.3 * 100);
}

template <int MODE>
void rung(const char* name, int blocks, int iters, float* out, const float* src, size_t span) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kg<MODE>, dim3(blocks), dim3(256), 0, 0, out, src, 4, span);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kg<MODE>, dim3(blocks), dim3(256), 0, 0, out, src, iters, span);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 64 * 4096.0;
    printf("%-40s blocks=%5d (%.0f/CU)  %.3f ms  %.1f TF  (%.1f%% of 157.3)\n", name, blocks, blocks / 256.0, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 157.3 * 100);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    __shared__ float As[32 * 129], Bs[32 * 129];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 32 * 129; i += 256) { As[i] = seed * (i & 7); Bs[i] = seed * (i & 3); }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* ap = As + (lane >> 5) * 129 + (wave >> 1) * 64 + (lane & 31);
    const float* bp = Bs + (lane >> 5) * 129 + (wave & 1) * 64 + (lane & 31);
    float a0 = seed, a1 = seed + 1, b0 = seed + 2, b1 = seed + 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2) {
            if (MODE >= 1) {
                a0 = ap[kk * 129]; a1 = ap[kk * 129 + 32];
                b0 = bp[kk * 129]; b1 = bp[kk * 129 + 32];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (MODE >= 2) {
            __syncthreads();
            if (MODE >= 3) {   // 32 LDS writes per thread per chunk like the igemm staging
#pragma unroll
                for (int r = 0; r < 16; ++r) { As[(r * 2 + (t >> 7)) * 129 + (t & 127)] = a0 + r; Bs[(r * 2 + (t >> 7)) * 129 + (t & 127)] = b0 + r; }
            }
            __syncthreads();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + t] = s;
}

template <int MODE>
void run(const char* name, int blocks, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 4, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 64 * 4096.0;
    printf("%-28s blocks=%5d (%.0f/CU)  %.3f ms  %.1f TF  (%.1f%% of 157.3)\n", name, blocks, blocks / 256.0, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 157.3 * 100);
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4 * 4);
    for (int bpc = 1; bpc <= 3; ++bpc) {
        run<0>("pure mfma", 256 * bpc, 2000, out);
        run<1>("mfma + lds operand reads", 256 * bpc, 2000, out);
        run<2>("  + 2 barriers / 64 mfma", 256 * bpc, 2000, out);
        run<3>("  + 32 lds writes / chunk", 256 * bpc, 2000, out);
    }
    run<1>("mfma + lds reads, 12 waves of blocks", 256 * 3 * 4, 500, out);
    const size_t span = (size_t)1 << 28;   // 1 GiB of floats
    float* src; hipMalloc(&src, (span + (1 << 24)) * 4); hipMemset(src, 0, (span + (1 << 24)) * 4);
    for (int bpc = 1; bpc <= 3; ++bpc) {
        rung<5>("igemm loop, L1/L2-hot source", 256 * bpc, 1000, out, src, span);
        rung<4>("igemm loop, streaming source", 256 * bpc, 1000, out, src, span);
        rung<7>("igemm loop hot, A as 4x dwordx4+b128", 256 * bpc, 1000, out, src, span);
        rung<8>("igemm loop hot, B every 3rd chunk", 256 * bpc, 999, out, src, span);
    }
    for (int bpc = 1; bpc <= 2; ++bpc) {
        runl<1>("lds-direct loop, hot source", 256 * bpc, 1000, out, src, span);
        runl<0>("lds-direct loop, streaming source", 256 * bpc, 1000, out, src, span);
    }
    rung<5>("igemm loop hot, 12 rounds of blocks", 256 * 3 * 12, 72, out, src, span);
    rung<4>("igemm loop streaming, 12 rounds", 256 * 3 * 12, 72, out, src, span);
    return 0;
}
