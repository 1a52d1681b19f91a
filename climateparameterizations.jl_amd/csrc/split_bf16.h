// split_bf16.h — fp32 matrix products on the bf16 matrix pipe by EXACT three-way operand splitting (COLNDE_MATRIX_BF16X3_EXACT, the default matrix arithmetic).
//
// A float has 24 significant bits = three bf16 (8 bits each): x = x_h + x_m + x_l exactly, by truncation (x_h = the top half of the
// word, r = x - x_h is exact, x_m = the top half of r, x_l = r - x_m has at most 8 significant bits).  A product a b is then the nine
// products of the parts, each EXACT in fp32 (8 x 8 bits), accumulated in fp32 by the bf16 MFMA; the six of them down to 2^-16 are kept
// (hh, hm, mh, hl, lh, mm), the three dropped ones are below 2^-23 |a b| together — the size of ONE fp32 rounding of the product,
// which the fp32 MFMA chain commits at every accumulation anyway.
#pragma once
#include <hip/hip_runtime.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float sp_f32x16 __attribute__((ext_vector_type(16)));
struct Bf3 { u32x4 h, m, l; };     // eight consecutive k of one operand row / column: three planes of packed bf16 pairs (element 2p in the low half)

__device__ __forceinline__ Bf3 bf3_split8(const float* x) {
    Bf3 o;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const float a = x[2 * p], b = x[2 * p + 1];
        const float ra = a - __uint_as_float(__float_as_uint(a) & 0xffff0000u), rb = b - __uint_as_float(__float_as_uint(b) & 0xffff0000u);
        const float la = ra - __uint_as_float(__float_as_uint(ra) & 0xffff0000u), lb = rb - __uint_as_float(__float_as_uint(rb) & 0xffff0000u);
        o.h[p] = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);      // (hi16(b) << 16) | hi16(a)
        o.m[p] = __builtin_amdgcn_perm(__float_as_uint(rb), __float_as_uint(ra), 0x07060302u);
        o.l[p] = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
    }
    return o;
}

// one pair of the eight: elements 2 p, 2 p + 1 of the three planes (for callers that spread a split between other work)
__device__ __forceinline__ void bf3_split_pair(float a, float b, int p, Bf3& o) {
    const float ra = a - __uint_as_float(__float_as_uint(a) & 0xffff0000u), rb = b - __uint_as_float(__float_as_uint(b) & 0xffff0000u);
    const float la = ra - __uint_as_float(__float_as_uint(ra) & 0xffff0000u), lb = rb - __uint_as_float(__float_as_uint(rb) & 0xffff0000u);
    o.h[p] = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
    o.m[p] = __builtin_amdgcn_perm(__float_as_uint(rb), __float_as_uint(ra), 0x07060302u);
    o.l[p] = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
}

__device__ __forceinline__ sp_f32x16 mfma_bf(u32x4 a, u32x4 b, sp_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// c += A B over one 16-deep k-block, fp32-grade: smallest products first
__device__ __forceinline__ sp_f32x16 mfma_bf3(const Bf3& a, const Bf3& b, sp_f32x16 c) {
    c = mfma_bf(a.m, b.m, c);
    c = mfma_bf(a.l, b.h, c);
    c = mfma_bf(a.h, b.l, c);
    c = mfma_bf(a.m, b.h, c);
    c = mfma_bf(a.h, b.m, c);
    c = mfma_bf(a.h, b.h, c);
    return c;
}
