// Probe: what the exact three-way bf16 split (csrc/split_bf16.h) and v_mfma_f32_32x32x16_bf16 do at the edges of the float32 range, beside
// v_mfma_f32_32x32x2_f32 (VERDICT r3 task 1e).  One wave, one 32 x 32 x 16 product C = A B with A = a everywhere in column 0 (zero elsewhere)
// and B = b in row 0, so C[i][j] = a b exactly, for a list of (a, b):
//   * bf16 MFMA fed a SUBNORMAL bf16 operand directly (is it flushed on input?) and producing a subnormal f32 result (flushed on output?);
//   * the split of a float whose m / l parts fall below FLT_MIN (|x| < 2^-110: l subnormal; |x| < 2^-118: m subnormal too);
//   * Inf and NaN operands through the split (x - x_h = Inf - Inf = NaN: the l/m planes carry NaN).
// build: hipcc -O3 --offload-arch=gfx950 -I ../../climateparameterizations.jl_amd/csrc split_edge.hip -o split_edge_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "split_bf16.h"

__global__ void __launch_bounds__(64) k_fp32(float a, float b, float* C) {
    const int lane = threadIdx.x, kh = lane >> 5;
    sp_f32x16 acc = (sp_f32x16)(0.0f);
    for (int k = 0; k < 16; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32((k + kh) == 0 ? a : 0.0f, (k + kh) == 0 ? b : 0.0f, acc, 0, 0, 0);
    if (lane == 0) C[0] = acc[0];
}

// the product through the six-product split
__global__ void __launch_bounds__(64) k_split(float a, float b, float* C) {
    const int lane = threadIdx.x, kh = lane >> 5;
    float a8[8], b8[8];
    for (int i = 0; i < 8; i++) { a8[i] = (8 * kh + i) == 0 ? a : 0.0f; b8[i] = (8 * kh + i) == 0 ? b : 0.0f; }
    const Bf3 A = bf3_split8(a8), B = bf3_split8(b8);
    sp_f32x16 acc = (sp_f32x16)(0.0f);
    acc = mfma_bf3(A, B, acc);
    if (lane == 0) {
        C[0] = acc[0];
        // the planes of a, as floats
        C[1] = __uint_as_float(A.h[0] << 16);
        C[2] = __uint_as_float(A.m[0] << 16);
        C[3] = __uint_as_float(A.l[0] << 16);
    }
}

// one bf16 MFMA on raw bf16 bit patterns (a_bits x b_bits in k = 0)
__global__ void __launch_bounds__(64) k_raw(unsigned a_bits, unsigned b_bits, float* C) {
    const int lane = threadIdx.x, kh = lane >> 5;
    u32x4 A = (u32x4)(0u), B = (u32x4)(0u);
    if (kh == 0) { A[0] = a_bits & 0xffffu; B[0] = b_bits & 0xffffu; }
    sp_f32x16 acc = (sp_f32x16)(0.0f);
    acc = mfma_bf(A, B, acc);
    if (lane == 0) C[0] = acc[0];
}

static unsigned bf16_bits(float x) { unsigned u; memcpy(&u, &x, 4); return u >> 16; }

int main() {
    float* d;
    hipMalloc(&d, 16 * sizeof(float));
    float h[4];
    printf("1. v_mfma_f32_32x32x16_bf16 on raw bf16 operands (k = 0 only): subnormal inputs and outputs\n");
    struct { float a, b; const char* what; } raw[] = {
        {ldexpf(1.0f, -130), ldexpf(1.0f, 20), "subnormal bf16 input 2^-130 x 2^20 (exact product 2^-110, normal)"},
        {ldexpf(1.0f, -133), ldexpf(1.0f, 100), "smallest bf16 subnormal 2^-133 x 2^100 (exact 2^-33)"},
        {ldexpf(1.0f, -100), ldexpf(1.0f, -30), "normal inputs 2^-100 x 2^-30 (exact 2^-130: subnormal f32 RESULT)"},
        {ldexpf(1.0f, -100), ldexpf(1.0f, -60), "normal inputs 2^-100 x 2^-60 (exact 2^-160: below the f32 subnormal range)"},
    };
    for (auto& r : raw) {
        hipLaunchKernelGGL(k_raw, dim3(1), dim3(64), 0, 0, bf16_bits(r.a), bf16_bits(r.b), d);
        hipMemcpy(h, d, 4, hipMemcpyDeviceToHost);
        printf("   %-80s -> %.9g (exact %.9g)%s\n", r.what, h[0], (double)r.a * (double)r.b, h[0] == 0.0f && (double)r.a * r.b != 0 ? "   FLUSHED" : "");
    }
    printf("2. a x b through the exact three-way split (six products) beside v_mfma_f32_32x32x2_f32; planes of a shown\n");
    struct { float a, b; } cases[] = {
        {1.2345678f, 0.87654321f},
        {1.2345678f * ldexpf(1.0f, -100), 0.87654321f},                  // l plane of a at 2^-116: normal
        {1.2345678f * ldexpf(1.0f, -108), 0.87654321f},                  // |a| ~ 2^-108 < 2^16 FLT_MIN: l plane subnormal
        {1.2345678f * ldexpf(1.0f, -116), 0.87654321f},                  // m plane at the edge, l subnormal
        {1.2345678f * ldexpf(1.0f, -124), 0.87654321f},                  // m and l subnormal
        {1.2345678f * ldexpf(1.0f, -126), 0.87654321f * ldexpf(1.0f, 10)},   // a itself at FLT_MIN
        {1.2345678f * ldexpf(1.0f, -130), 0.87654321f * ldexpf(1.0f, 10)},   // a subnormal
        {1.2345678f * ldexpf(1.0f, -60), 0.87654321f * ldexpf(1.0f, -60)},   // product 2^-120: near FLT_MIN
        {1.2345678f * ldexpf(1.0f, -64), 0.87654321f * ldexpf(1.0f, -64)},   // product 2^-128: subnormal result
        {INFINITY, 0.5f}, {-INFINITY, 0.5f}, {NAN, 0.5f}, {INFINITY, 0.0f}, {3.0e38f, 3.0e38f},
    };
    printf("   %-16s %-16s | %-16s %-16s | %-14s %-14s %-14s | rel. difference\n", "a", "b", "f32 MFMA", "split", "a_h", "a_m", "a_l");
    for (auto& c : cases) {
        float f32;
        hipLaunchKernelGGL(k_fp32, dim3(1), dim3(64), 0, 0, c.a, c.b, d);
        hipMemcpy(&f32, d, 4, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, c.a, c.b, d);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        const double ex = (double)c.a * (double)c.b;
        printf("   %-16.9g %-16.9g | %-16.9g %-16.9g | %-14.6g %-14.6g %-14.6g | f32 %.2e, split %.2e\n", c.a, c.b, f32, h[0], h[1], h[2], h[3],
               std::isfinite(ex) && ex != 0 ? fabs(f32 - ex) / fabs(ex) : 0.0, std::isfinite(ex) && ex != 0 ? fabs(h[0] - ex) / fabs(ex) : 0.0);
    }
    hipFree(d);
    return 0;
}
