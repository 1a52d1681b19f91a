// Probe: a cheaper exact three-way bf16 split for gfx950 — round-to-nearest planes from v_cvt_pk_bf16_f32, remainders from
// v_dot2c_f32_bf16 (r = x + (-1) * h.lo + 0 * h.hi): 28 vector instructions per eight values instead of the 44 of the truncating
// and/sub/perm form in csrc/split_bf16.h.  Checks, per candidate: (1) h + m + l == x exactly (float64) over random floats of every
// exponent, (2) what happens at the edges (subnormal parts, near FLT_MAX, Inf/NaN next to a finite neighbour), (3) vector-ALU
// throughput of both forms (s_memtime around 256 dependent-free splits per lane).
// build: hipcc -O3 --offload-arch=gfx950 -I ../../climateparameterizations.jl_amd/csrc split_rne.hip -o split_rne_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "split_bf16.h"

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ Bf3 bf3_split8_rne(const float* x) {
    Bf3 o;
    bf16x2 lo_one, hi_one;                                  // (-1, 0) and (0, -1): picks one half of a packed pair
    lo_one[0] = (__bf16)(-1.0f); lo_one[1] = (__bf16)0.0f;
    hi_one[0] = (__bf16)0.0f; hi_one[1] = (__bf16)(-1.0f);
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const float a = x[2 * p], b = x[2 * p + 1];
        bf16x2 h; h[0] = (__bf16)a; h[1] = (__bf16)b;
        const float ra = __builtin_amdgcn_fdot2_f32_bf16(h, lo_one, a, false), rb = __builtin_amdgcn_fdot2_f32_bf16(h, hi_one, b, false);
        bf16x2 m; m[0] = (__bf16)ra; m[1] = (__bf16)rb;
        const float la = __builtin_amdgcn_fdot2_f32_bf16(m, lo_one, ra, false), lb = __builtin_amdgcn_fdot2_f32_bf16(m, hi_one, rb, false);
        bf16x2 l; l[0] = (__bf16)la; l[1] = (__bf16)lb;
        o.h[p] = __builtin_bit_cast(unsigned, h);
        o.m[p] = __builtin_bit_cast(unsigned, m);
        o.l[p] = __builtin_bit_cast(unsigned, l);
    }
    return o;
}

template <bool RNE>
__global__ void k_planes(const float* x, float* planes, int n) {     // planes [n][3]: h, m, l of each value as floats
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i >= n) return;
    float v[8];
    for (int e = 0; e < 8; e++) v[e] = x[i + e];
    const Bf3 s = RNE ? bf3_split8_rne(v) : bf3_split8(v);
    for (int e = 0; e < 8; e++) {
        const unsigned sh = (e & 1) ? 0xffff0000u : 0u;
        auto get = [&](unsigned w) { return __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16)); };
        (void)sh;
        planes[(size_t)(i + e) * 3 + 0] = get(s.h[e >> 1]);
        planes[(size_t)(i + e) * 3 + 1] = get(s.m[e >> 1]);
        planes[(size_t)(i + e) * 3 + 2] = get(s.l[e >> 1]);
    }
}

template <bool RNE>
__global__ void __launch_bounds__(256) k_rate(const float* x, unsigned* out, unsigned long long* cyc, int reps) {
    float v[8];
    for (int e = 0; e < 8; e++) v[e] = x[threadIdx.x * 8 + e];
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) {
        const Bf3 s = RNE ? bf3_split8_rne(v) : bf3_split8(v);
        for (int p = 0; p < 4; p++) acc ^= s.h[p] ^ s.m[p] ^ s.l[p];
        for (int e = 0; e < 8; e++) v[e] = __uint_as_float(__float_as_uint(v[e]) ^ (acc & 0x7fffu));      // next input depends on this result: the loop is not hoisted
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int n = 1 << 20;
    std::vector<float> x(n);
    std::mt19937_64 g(20261005);
    for (int i = 0; i < n; i++) {                          // every exponent of the normal range, random sign and mantissa
        unsigned u = (unsigned)(g() & 0x807fffffu) | ((unsigned)(1 + g() % 253) << 23);
        memcpy(&x[i], &u, 4);
    }
    float *dx, *dp;
    hipMalloc(&dx, n * 4); hipMalloc(&dp, (size_t)n * 12);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<float> pl((size_t)n * 3);
    for (int rne = 0; rne < 2; rne++) {
        if (rne) k_planes<true><<<n / 8 / 256, 256>>>(dx, dp, n); else k_planes<false><<<n / 8 / 256, 256>>>(dx, dp, n);
        hipMemcpy(pl.data(), dp, (size_t)n * 12, hipMemcpyDeviceToHost);
        long bad = 0, notbf = 0; double worst = 0; int worst_i = -1;
        double mmax = 0, lmax = 0;
        for (int i = 0; i < n; i++) {
            const double s = (double)pl[3 * i] + (double)pl[3 * i + 1] + (double)pl[3 * i + 2];
            const double e = fabs(s - (double)x[i]) / fabs((double)x[i]);
            if (e != 0.0) { bad++; if (e > worst) { worst = e; worst_i = i; } }
            for (int q = 0; q < 3; q++) { unsigned u; memcpy(&u, &pl[3 * i + q], 4); if (u & 0xffffu) notbf++; }
            mmax = fmax(mmax, fabs(pl[3 * i + 1] / x[i])); lmax = fmax(lmax, fabs(pl[3 * i + 2] / x[i]));
        }
        printf("%s: %d random normal floats (exponents 1..253): h + m + l != x for %ld (worst relative %.3e at x = %.9g), planes not bf16: %ld, max |m/x| = 2^%.2f, max |l/x| = 2^%.2f\n",
               rne ? "RNE + dot2c " : "truncating   ", n, bad, worst, worst_i >= 0 ? x[worst_i] : 0.0f, notbf, log2(mmax), log2(lmax));
    }
    // edges: eight values per call, shown with their planes under both forms
    const float edge[16] = {ldexpf(1.2345678f, -100), ldexpf(1.2345678f, -110), ldexpf(1.2345678f, -118), ldexpf(1.2345678f, -126), ldexpf(1.2345678f, -130), 3.3e38f, 3.4e38f, 1.0f,
                            INFINITY, 1.2345678f, 1.2345678f, NAN, -INFINITY, 0.0f, -0.0f, 1.17549435e-38f};
    hipMemcpy(dx, edge, sizeof edge, hipMemcpyHostToDevice);
    for (int rne = 0; rne < 2; rne++) {
        if (rne) k_planes<true><<<1, 2>>>(dx, dp, 16); else k_planes<false><<<1, 2>>>(dx, dp, 16);
        hipMemcpy(pl.data(), dp, 16 * 12, hipMemcpyDeviceToHost);
        printf("%s edges (x | h m l | h+m+l-x relative):\n", rne ? "RNE + dot2c" : "truncating");
        for (int i = 0; i < 16; i++) {
            const double s = (double)pl[3 * i] + (double)pl[3 * i + 1] + (double)pl[3 * i + 2];
            printf("   %-16.9g | %-14.7g %-14.7g %-14.7g | %.3e\n", edge[i], pl[3 * i], pl[3 * i + 1], pl[3 * i + 2], edge[i] != 0 ? fabs(s - edge[i]) / fabs(edge[i]) : fabs(s));
        }
    }
    // throughput
    unsigned* dout; unsigned long long* dc;
    hipMalloc(&dout, 4096 * 256 * 4); hipMalloc(&dc, 4096 * 8);      // (sized for the largest launch below)
    const int reps = 256;
    for (int rne = 0; rne < 2; rne++) {
        for (int w = 0; w < 2; w++) {
            if (rne) k_rate<true><<<1024, 256>>>(dx, dout, dc, reps); else k_rate<false><<<1024, 256>>>(dx, dout, dc, reps);
            hipDeviceSynchronize();
        }
        std::vector<unsigned long long> c(1024);
        hipMemcpy(c.data(), dc, 1024 * 8, hipMemcpyDeviceToHost);
        double m = 0; for (auto v : c) m += (double)v; m /= 1024;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (rne) k_rate<true><<<4096, 256>>>(dx, dout, dc, reps); else k_rate<false><<<4096, 256>>>(dx, dout, dc, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.0f clock-counter ticks per wave for %d splits of eight (one wave per SIMD: 256-thread blocks, 1024 blocks); 4096 blocks: %.3f ms\n",
               rne ? "RNE + dot2c" : "truncating ", m, reps, ms);
    }
    return 0;
}
