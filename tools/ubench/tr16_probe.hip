// Probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read) with per-lane addresses: prints, for every lane and result
// element, the LDS element index the value came from.  Used to pin the B-fragment addressing of the split wgrad kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef short s4 __attribute__((ext_vector_type(4)));

__global__ void probe(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int a;      // element (2-byte) address
    if (mode == 0) a = l * 4;                                          // contiguous 8 B per lane
    else {      // [pixel][32 ch] rows of 32 elements (64 B): lane (g = l>>4, r = (l&15)>>2, q = l&3)
        const int g = l >> 4, r = (l & 15) >> 2, q = l & 3;
        a = ((g >> 1) * 8 + r) * 32 + (g & 1) * 16 + q * 4;
    }
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(lds + a));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

int main() {
    unsigned short* d;
    hipMalloc(&d, 64 * 4 * 2);
    std::vector<unsigned short> h(256);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int j = 0; j < 4; ++j) {
                if (mode == 0) printf(" %4d", h[l * 4 + j]);
                else printf(" (px %2d, ch %2d)", h[l * 4 + j] / 32, h[l * 4 + j] % 32);
            }
            printf("\n");
        }
    }
    return 0;
}
