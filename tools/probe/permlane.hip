// probe: semantics of v_permlane32_swap on gfx950 (prints which source lane each output lane sees)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* c) {
    int l = threadIdx.x;
    float x = (float)l;
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    c[l] = __uint_as_float(r[0]);
    c[64 + l] = __uint_as_float(r[1]);
}
int main() {
    float* d; float h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("r0: lane0=%g lane1=%g lane31=%g lane32=%g lane33=%g lane63=%g\n", h[0], h[1], h[31], h[32], h[33], h[63]);
    printf("r1: lane0=%g lane1=%g lane31=%g lane32=%g lane33=%g lane63=%g\n", h[64], h[65], h[95], h[96], h[97], h[127]);
    return 0;
}
