// Probe: how much vector-ALU issue is left to a second wave of a SIMD while the first runs a dependent fp32 MFMA chain.
// Workgroups of 8 waves (two per SIMD): waves 0..3 run MFMAs (MODE 1: 16x16x4, MODE 2: 32x32x2, MODE 0: idle), waves 4..7 run a
// chain of independent v_fma_f32.  Reports the VALU waves' time with and without the MFMA waves beside them.
// build: hipcc -O3 --offload-arch=gfx950 coissue.hip -o coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    if (wave < 4) {
        if (MODE == 1) {
            f32x4 c = {0, 0, 0, 0};
            for (int i = 0; i < iters; i++)
#pragma unroll
                for (int u = 0; u < 16; u++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
            s = c[0];
        } else if (MODE == 2) {
            f32x16 c = (f32x16)(0.0f);
            for (int i = 0; i < iters / 2; i++)
#pragma unroll
                for (int u = 0; u < 16; u++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
            s = c[0];
        }
    } else {
        float x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f);
                x4 = fmaf(x4, 1.0001f, 0.5f); x5 = fmaf(x5, 1.0001f, 0.5f); x6 = fmaf(x6, 1.0001f, 0.5f); x7 = fmaf(x7, 1.0001f, 0.5f);
            }
        }
        s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[MODE * 2 + (threadIdx.x == 256)] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    (void)hipMallocManaged(&cyc, 8 * sizeof(unsigned long long));
    const int iters = 4000;
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    const double nfma = iters * 16.0 * 8.0;
    printf("VALU wave alone:            %.2f cycles per v_fma_f32\n", cyc[1] / nfma);
    printf("beside a 16x16x4 f32 chain: %.2f cycles per v_fma_f32 (MFMA wave: %.1f cycles per MFMA)\n", cyc[3] / nfma, cyc[2] / (iters * 16.0));
    printf("beside a 32x32x2 f32 chain: %.2f cycles per v_fma_f32 (MFMA wave: %.1f cycles per MFMA)\n", cyc[5] / nfma, cyc[4] / (iters / 2 * 16.0));
    return 0;
}
