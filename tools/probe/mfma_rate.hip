// Probe: issue cost of fp32 MFMA shapes on gfx950, dependent chain vs 4 independent chains, one wave per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o mfma_rate_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f32x16 d0 = (f32x16)(0.0f), d1 = d0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {          // 16x16x4 dependent
#pragma unroll
            for (int u = 0; u < 16; u++) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        } else if (MODE == 1) {   // 16x16x4, 4 independent accumulators
#pragma unroll
            for (int u = 0; u < 4; u++) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
            }
        } else if (MODE == 2) {   // 32x32x2 dependent
#pragma unroll
            for (int u = 0; u < 16; u++) d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
        } else {                  // 32x32x2, 2 independent
#pragma unroll
            for (int u = 0; u < 8; u++) {
                d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d1, 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4 * sizeof(float));
    hipMallocManaged(&cyc, 8 * sizeof(unsigned long long));
    const int iters = 20000;
    const char* names[4] = {"16x16x4 dependent", "16x16x4 4 independent", "32x32x2 dependent", "32x32x2 2 independent"};
    for (int wpb = 1; wpb <= 8; wpb *= 2) {          // waves per workgroup: 4 = one per SIMD, 8 = two per SIMD
        float ms[4];
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(M) hipLaunchKernelGGL(k<M>, dim3(256), dim3(64 * wpb), 0, 0, out, cyc, iters); hipDeviceSynchronize(); \
        hipEventRecord(e0, 0); hipLaunchKernelGGL(k<M>, dim3(256), dim3(64 * wpb), 0, 0, out, cyc, iters); hipEventRecord(e1, 0); \
        hipEventSynchronize(e1); hipEventElapsedTime(&ms[M], e0, e1);
        RUN(0) RUN(1) RUN(2) RUN(3)
        for (int m = 0; m < 4; m++)
            printf("waves/WG %d  %-24s %.1f ticks per MFMA (wave 0 clock)  kernel %.3f ms = %.1f ns per MFMA per wave\n", wpb, names[m],
                   (double)cyc[m] / (iters * 16.0), ms[m], ms[m] * 1e6 / (iters * 16.0));
    }
    return 0;
}
