// engine_regtile.hip — register-resident tile engine for the static wind-mixing shape (gfx950 / CDNA4 only).
//
// One wavefront owns 32 columns.  Every quantity of a column lives in the MFMA 32x32 accumulator ("D") layout:
// lane (j = lane & 31, h = lane >> 5) holds, in element r of a 16-float tile, row  rho(r, h) = (r&3) + 8(r>>2) + 4h
// of column j.  With v_mfma_f32_32x32x2_f32 the B operand of k-step r is exactly element r of such a tile (K pair =
// rows rho(r,0), rho(r,1)), so a layer's output feeds the next layer with NO lane movement, no LDS round trip and no
// barrier: state [u;v;T] -> Dense(96,50) -> Dense(50,20) -> Dense(20,31) -> fluxes on faces, all in registers.
// The A operand (weights) is read from one compact LDS image (plain row-major matrices with odd row strides) as
// `per-lane base + compile-time immediate`: one ds_read_b32 per 64-cycle MFMA.  Vertical neighbours (the D^f / D^c
// stencils) are the previous/next tile element, except every fourth level where they sit in the partner lane (lane^32).
//
// Row placement (chosen here, free because K order and output-row order of a GEMM are arbitrary):
//   layer-1 outputs: the 3x50 features are stacked into 5 tiles; register G = 16*tile + r carries features
//                    (2g, 2g+1) of net n, with n = G / 25, g = G % 25 (G >= 75: padding, never consumed);
//   layer-2 outputs: registers r < 10 carry features (2r, 2r+1), the rest is padding;
//   layer-3 outputs: row = face index (row 0 = bottom boundary face, where the NN contributes nothing).
//
// Reference arithmetic: wind_mixing/src/NDE_training.jl:46-165 (NDE, predict_flux, predict_NDE).
#include <cstdlib>
#include "engine_regtile.h"
#include "split_bf16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// diagnostic build only (-DCOLNDE_STAMPS): per-phase cycle sums of wave 0 of workgroup 0
#ifdef COLNDE_STAMPS
__device__ unsigned long long g_rt_stamps[16];
#define RT_STAMP_DECL unsigned long long rs_t0 = 0, rs_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long rs_kk = __builtin_amdgcn_s_memtime(), rs_rr = __builtin_amdgcn_s_memrealtime()
#define RT_STAMP_BEGIN() do { __builtin_amdgcn_sched_barrier(0); rs_t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define RT_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); rs_acc[i] += t_ - rs_t0; rs_t0 = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
// slots 8, 9: the whole kernel in shader ticks and in 100 MHz reference ticks (their ratio is the clock the kernel ran at)
#define RT_STAMP_FLUSH() do { if (blockIdx.x == 0 && threadIdx.x == 0) { for (int q_ = 0; q_ < 8; q_++) g_rt_stamps[q_] = rs_acc[q_]; g_rt_stamps[8] = __builtin_amdgcn_s_memtime() - rs_kk; g_rt_stamps[9] = __builtin_amdgcn_s_memrealtime() - rs_rr; } } while (0)
#else
#define RT_STAMP_DECL
#define RT_STAMP_BEGIN()
#define RT_STAMP(i)
#define RT_STAMP_FLUSH()
#endif
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

#define RHO0(r) (((r) & 3) + 8 * ((r) >> 2))
// (Tried and removed, see DESIGN.md §6: "parking" cotangents in an L2-resident scratch buffer to free registers — slower and
// 115 GB of extra traffic; prefetching the stage input one stage ahead — spilled under exposed waits.)
#ifndef RT16_WAVES
#define RT16_WAVES 8      // wavefronts per workgroup of the 16-column forward kernel (two per SIMD)
#endif
#ifndef RT_ADJ_CH
#define RT_ADJ_CH 8       // A-operand prefetch depth (k-steps) of the adjoint kernel's layer-1 chains
#endif
#ifndef RT_ZRICH
#define RT_ZRICH 0            // 1 (A/B build, VERDICT r3 task 3): the forward kernel tapes the layer-1 activation VALUES and DERIVATIVES (act(z1), act'(z1)) instead of
#endif                        // z1 — twice the bytes of this tape (+50 GB written and read at the bench size), no layer-1 activation pairs in the adjoint's stage loop
#ifndef RT_Z2TAPE
#define RT_Z2TAPE 0           // 1 (A/B build, round 5): the forward kernel tapes the layer-2 pre-activations too (3 nets x 3 groups x 64 lanes x 4 behind the layer-1 record: +9.2 KB per
#endif                        // 32-column stage, +19 GB at the bench size) and the adjoint drops its layer-2 recomputation (75 of its 336 f32 MFMAs per stage).  Measured SLOWER: forward
                              // 27.5 -> 29.4-30.0 ms, adjoint 56.2 -> 61.9 ms (Z2 fetched one net ahead: scratch 536 -> 684 B) / 64.5 ms (fetched at the net's top: 668 B and the
                              // latency exposed) — the recomputation's MFMAs ride under the vector phases, its replacement's registers do not exist (profiles/r05_ab_adjoint_variants.log)
#define RT_TAPEZ_HALF (21 * 256)
#define RT_Z2OFF ((RT_ZRICH ? 2 : 1) * RT_TAPEZ_HALF)   // floats per (tile, step, stage) of the layer-1 tape: 3 nets x 7 groups x 64 lanes x 4 (RT_ZRICH: values, then derivatives)
#define RT_TAPEZ (RT_Z2OFF + (RT_Z2TAPE ? 9 * 256 : 0)) // ... and of the whole pre-activation record (layer 2 behind layer 1: net n's register r < 10 = features 2r, 2r+1 is element r & 3 of group 3n + (r >> 2))
#define RT_TAPE2 (20 * 256)   // ... of the layer-1 delta tape: 20 groups x 64 lanes x 4, the three nets' 25 registers stacked (G = 25 n + g)

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// exchange between the two lane halves with v_permlane32_swap (one VALU op, no LDS round trip): lanes h = 0 receive
// `from_hi` as held by their partner lane + 32, lanes h = 1 receive `from_lo` as held by their partner lane - 32.
// (The instruction swaps vdst[32:63] with src0[0:31]: result 0 = {vdst.lo, src0.lo}, result 1 = {vdst.hi, src0.hi};
// probed on gfx950 by tools/probe/permlane.hip.)  The two values go in as separate operands on purpose: selecting the
// value to send per lane first (`h ? T[i] : T[j]`) is folded by LLVM into a divergent dynamic index into the register
// tile, which lowers to a 16-long v_cmp/v_cndmask chain per exchange.
__device__ __forceinline__ float xchg32(float from_hi, float from_lo, int h) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(from_hi), __float_as_uint(from_lo), false, false);
    return __uint_as_float(h ? r[0] : r[1]);
}
__device__ __forceinline__ float swap32(float x, int h) { return xchg32(x, x, h); }

// tile value one level below (rho - 1); level -1 reads `below`
__device__ __forceinline__ f32x16 shift_down(const f32x16 T, int h, float below) {
    f32x16 o;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        // row 8g (h = 0) takes T[4g-1] of the upper half; row 8g+4 (h = 1) takes T[4g+3] of the lower half
        const float recv = xchg32(g > 0 ? T[(4 * g + 15) & 15] : 0.0f, T[4 * g + 3], h);
        o[4 * g] = (h == 0 && g == 0) ? below : recv;
        o[4 * g + 1] = T[4 * g];
        o[4 * g + 2] = T[4 * g + 1];
        o[4 * g + 3] = T[4 * g + 2];
    }
    return o;
}

// tile value one level above (rho + 1); level 32 reads `above`
__device__ __forceinline__ f32x16 shift_up(const f32x16 T, int h, float above) {
    f32x16 o;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        // row 8g+3 (h = 0) takes T[4g] of the upper half; row 8g+7 (h = 1) takes T[4g+4] of the lower half
        const float recv = xchg32(T[4 * g], g < 3 ? T[(4 * g + 4) & 15] : 0.0f, h);
        o[4 * g + 3] = (h == 1 && g == 3) ? above : recv;
        o[4 * g] = T[4 * g + 1];
        o[4 * g + 1] = T[4 * g + 2];
        o[4 * g + 2] = T[4 * g + 3];
    }
    return o;
}

template <int ACT>
__device__ __forceinline__ float rt_act(float z) {
    if (ACT == COLNDE_ACT_RELU) return fmaxf(z, 0.0f);
    if (ACT == COLNDE_ACT_MISH) {
        const float e = __expf(__builtin_amdgcn_fmed3f(z, -3.0e38f, 20.0f));
        const float n = e * (e + 2.0f);
        return z * fast_div(n, n + 2.0f);
    }
    if (ACT == COLNDE_ACT_SWISH) return fast_div(z, 1.0f + __expf(-z));
    if (ACT == COLNDE_ACT_TANH) { const float e = __expf(2.0f * fminf(fmaxf(z, -15.0f), 15.0f)); return 1.0f - fast_div(2.0f, 1.0f + e); }
    if (ACT == COLNDE_ACT_LEAKYRELU) return z > 0.0f ? z : 0.01f * z;
    return z;
}

template <int ACT>
__device__ __forceinline__ f32x16 act_tile(f32x16 z) {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; r++) o[r] = rt_act<ACT>(z[r]);
    return o;
}

// per-lane A-operand bases into the LDS weight image
struct RtBases {
    int a1[5];   // layer 1, stacked tiles: row of tile mt for this lane's output row
    int a2;      // layer 2
    int a3;      // layer 3
};

__device__ __forceinline__ RtBases rt_bases(int lane) {
    const int i = lane & 31, h = lane >> 5;
    const int r_i = (i & 3) + 4 * (i >> 3), h_i = (i >> 2) & 1;
    RtBases b;
#pragma unroll
    for (int mt = 0; mt < 5; mt++) {
        const int G = min(mt * 16 + r_i, 74);                       // padding rows read a valid row; their output is never used
        const int row = (G / 25) * 50 + 2 * (G % 25) + h_i;
        b.a1[mt] = RT_W1C + row * RT_LD1 + 4 * h;
    }
    b.a2 = RT_W2C + ((r_i < 10) ? 2 * r_i + h_i : 0) * RT_LD2 + h;
    b.a3 = RT_W3C + (i - 1) * RT_LD3 + h;                           // output row i = face i <- NN output i-1 (row 0 masked later)
    return b;
}

// acc += sum_{s<N} A_s * B_s on the matrix pipe, with the A operands (one ds_read_b32 each) fetched one chunk of CH
// k-steps ahead of their MFMAs.  The sched_barrier mask lets ALU work (e.g. the previous tile's activations) be
// scheduled into the MFMA shadow but pins the DS reads to their region: left alone, the scheduler hoists hundreds of
// operand reads and spills.
#define RT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0x00F)
#define RT_SCHED_HARD() __builtin_amdgcn_sched_barrier(0)      // nothing crosses: the operand fetches of a software pipeline stay AHEAD of the MFMAs that cover them

template <int N, int CH, class AF, class BF, class FF>
__device__ __forceinline__ f32x16 rt_chain_fill(const float* wl, f32x16 acc, AF aidx, BF bval, FF fill) {
    float a[2][CH];
#pragma unroll
    for (int u = 0; u < CH; u++)
        if (u < N) a[0][u] = wl[aidx(u)];
    RT_SCHED_FENCE();
#pragma unroll
    for (int c = 0; c * CH < N; c++) {
#pragma unroll
        for (int u = 0; u < CH; u++)
            if ((c + 1) * CH + u < N) a[(c + 1) & 1][u] = wl[aidx((c + 1) * CH + u)];
        fill(c);        // independent VALU work: scheduled into the gaps between this chunk's dependent MFMAs
#pragma unroll
        for (int u = 0; u < CH; u++)
            if (c * CH + u < N) acc = mfma32(a[c & 1][u], bval(c * CH + u), acc);
        RT_SCHED_FENCE();
    }
    return acc;
}

template <int N, int CH, class AF, class BF>
__device__ __forceinline__ f32x16 rt_chain(const float* wl, f32x16 acc, AF aidx, BF bval) {
    return rt_chain_fill<N, CH>(wl, acc, aidx, bval, [](int) {});
}

// The three MLPs on the stage input X (3 tiles u, v, T) -> face-flux tiles O (3 tiles), all in registers.
// The activations of a finished tile are issued as "filler" inside the next chain's chunks (MFMA shadow).
template <int ACT>
__device__ __forceinline__ void rt_mlp_forward(const float* wl, const RtBases& b, const f32x16 (&X)[3], int h, f32x16 (&O)[3]) {
    f32x16 A1[5];      // holds the raw pre-activation of a tile until the next chain's filler activates it in place
#pragma unroll
    for (int mt = 0; mt < 5; mt++) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int G = mt * 16 + r;
            acc[r] = wl[RT_B1C + (G < 75 ? (G / 25) * 50 + 2 * (G % 25) : 150 + 2 * (G - 75)) + h];
        }
        const int base = b.a1[mt];
        A1[mt] = rt_chain_fill<48, 8>(wl, acc, [=](int s) { return base + (s >> 4) * 32 + RHO0(s & 15); },
                                      [&](int s) { return X[s >> 4][s & 15]; },
                                      [&](int c) {
                                          if (mt > 0) {
#pragma unroll
                                              for (int r = 0; r < 16; r++)
                                                  if (r / 3 == c) A1[mt > 0 ? mt - 1 : 0][r] = rt_act<ACT>(A1[mt > 0 ? mt - 1 : 0][r]);
                                          }
                                      });
    }
    f32x16 Z2[3];
#pragma unroll
    for (int n = 0; n < 3; n++) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = r < 10 ? wl[RT_B2C + n * 20 + 2 * r + h] : 0.0f;
        const int base2 = b.a2 + n * 20 * RT_LD2;
        // net 0 reads layer-1 registers G < 25 only, so tile 4 (G >= 64) can still be activated under its chain;
        // nets 1, 2 activate the previous net's layer-2 outputs
        Z2[n] = rt_chain_fill<25, 5>(wl, acc, [=](int s) { return base2 + 2 * s; },
                                     [&](int s) { return A1[(25 * n + s) >> 4][(25 * n + s) & 15]; },
                                     [&](int c) {
                                         if (n == 0) {
#pragma unroll
                                             for (int r = 0; r < 16; r++)
                                                 if (r / 4 == c) A1[4][r] = rt_act<ACT>(A1[4][r]);
                                         } else {
#pragma unroll
                                             for (int r = 0; r < 10; r++)
                                                 if (r / 2 == c) Z2[n > 0 ? n - 1 : 0][r] = rt_act<ACT>(Z2[n > 0 ? n - 1 : 0][r]);
                                         }
                                     });
    }
#pragma unroll
    for (int r = 0; r < 10; r++) Z2[2][r] = rt_act<ACT>(Z2[2][r]);
#pragma unroll
    for (int n = 0; n < 3; n++) {
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; r++) o[r] = wl[RT_B3C + n * 32 + RHO0(r) + 4 * h];
        const int base3 = b.a3 + n * 31 * RT_LD3;
        O[n] = rt_chain<10, 10>(wl, o, [=](int s) { return base3 + 2 * s; }, [&](int s) { return Z2[n][s]; });
    }
}

struct RtBC { float b[3], t[3]; };   // scaled bottom / top fluxes of (uw, vw, wT) for this lane's column

// K = RHS(X) given the NN face fluxes O  (predict_flux :104-147, predict_NDE :160-162; MPP / CA branches, zero_weights)
__device__ __forceinline__ void rt_physics_forward(const DevModel& m, const f32x16 (&X)[3], const f32x16 (&O)[3],
                                                   const RtBC& bc, int h, f32x16 (&K)[3]) {
    const float Nz = 32.0f;
    f32x16 F[3];
    if (m.mpp || m.ca) {
        const f32x16 Ud = shift_down(X[0], h, 0.0f), Vd = shift_down(X[1], h, 0.0f), Td = shift_down(X[2], h, 0.0f);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const bool in = !(r == 0 && h == 0);                   // face rho(r,h) >= 1 (face 0 is the bottom boundary)
            const float gu = in ? (X[0][r] - Ud[r]) * Nz : 0.0f;
            const float gv = in ? (X[1][r] - Vd[r]) * Nz : 0.0f;
            const float gT = in ? (X[2][r] - Td[r]) * Nz : 0.0f;
            float f0 = in ? O[0][r] : 0.0f, f1 = in ? O[1][r] : 0.0f, f2 = in ? O[2][r] : 0.0f;
            if (!m.zero_w && !in) { f0 = bc.b[0]; f1 = bc.b[1]; f2 = bc.b[2]; }
            if (m.mpp) {
                if (in) {
                    const float a1 = m.sig_u * (gu + m.eps), a2 = m.sig_v * (gv + m.eps);
                    const float Ri = fast_div(m.B * (gT + m.eps), a1 * a1 + a2 * a2);
                    const float e = __expf(2.0f * fminf(fmaxf((Ri - m.Ric) * m.inv_dRi, -15.0f), 15.0f));
                    const float th = 1.0f - fast_div(2.0f, 1.0f + e);
                    const float nu = m.nu0 + m.nu_minus * (1.0f - th) * 0.5f;
                    f0 -= m.cs[0] * nu * gu;
                    f1 -= m.cs[1] * nu * gv;
                    f2 -= m.cs[2] * (nu * m.inv_Pr) * gT;
                } else if (m.zero_w) {
                    f0 += bc.b[0] - m.s0[0];
                    f1 += bc.b[1] - m.s0[1];
                    f2 += bc.b[2] - m.s0[2];
                }
            } else if (in) {
                f2 -= m.cs[2] * m.kappa * fminf(0.0f, gT);
            }
            F[0][r] = f0; F[1][r] = f1; F[2][r] = f2;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const bool in = !(r == 0 && h == 0);
            F[0][r] = in ? O[0][r] : bc.b[0];
            F[1][r] = in ? O[1][r] : bc.b[1];
            F[2][r] = in ? O[2][r] : bc.b[2];
        }
    }
    // top boundary face (face Nz): zero_weights: 0 - (-(BC - s(0))) ; else the BC itself
    float top[3];
#pragma unroll
    for (int k = 0; k < 3; k++) top[k] = m.zero_w ? bc.t[k] - m.s0[k] : bc.t[k];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const f32x16 Fu = shift_up(F[k], h, top[k]);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float v = -m.A[k] * (Fu[r] - F[k][r]);
            if (k == 0) v += m.cor_u * (m.sig_v * X[1][r] + m.mu_v);
            if (k == 1) v -= m.cor_v * (m.sig_u * X[0][r] + m.mu_u);
            K[k][r] = v;
        }
    }
}

__device__ __forceinline__ float rt_top_flux(const DevModel& m, float bc5, float t) {
    if (!m.diurnal) return bc5;
    const float wq = bc5 * __sinf(6.283185307179586f / 86400.0f * (t * m.tau)) / m.alpha_g;
    return (wq - m.mu_wT) / m.sig_wT;
}

extern __shared__ __attribute__((aligned(16))) float rt_smem[];

// ------------------------------------------------------------------------------------------------
// weight image
// ------------------------------------------------------------------------------------------------
__global__ void rt_pack_kernel(DevModel m, const float* __restrict__ w, float* __restrict__ img) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < RT_IMG_FLOATS; e += gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (e < RT_W2C) {
            const int row = e / RT_LD1, c = e - row * RT_LD1;
            if (row < 150 && c < 96) { const int n = row / 50, f = row - n * 50; v = w[n * m.net_size + m.w_off[0] + c * 50 + f]; }
        } else if (e < RT_W3C) {
            const int q = e - RT_W2C, row = q / RT_LD2, c = q - row * RT_LD2;
            if (row < 60 && c < 50) { const int n = row / 20, f = row - n * 20; v = w[n * m.net_size + m.w_off[1] + c * 20 + f]; }
        } else if (e < RT_B1C) {
            const int q = e - RT_W3C, row = q / RT_LD3, c = q - row * RT_LD3;
            if (row < 93 && c < 20) { const int n = row / 31, f = row - n * 31; v = w[n * m.net_size + m.w_off[2] + c * 31 + f]; }
        } else if (e < RT_B2C) {
            const int q = e - RT_B1C;
            if (q < 150) { const int n = q / 50, f = q - n * 50; v = w[n * m.net_size + m.b_off[0] + f]; }
        } else if (e < RT_B3C) {
            const int q = e - RT_B2C;
            if (q < 60) { const int n = q / 20, f = q - n * 20; v = w[n * m.net_size + m.b_off[1] + f]; }
        } else {
            const int q = e - RT_B3C, n = q / 32, face = q - n * 32;
            if (face >= 1) v = w[n * m.net_size + m.b_off[2] + face - 1];
        }
        img[e] = v;
    }
}


// The forward nets as A operands of v_mfma_f32_16x16x32_bf16 (COLNDE_MATRIX_BF16X3_EXACT), made from the fp32 image: group G, plane p, lane (i = lane & 15,
// kg = lane >> 4), element e = the weight that multiplies the B element e of lane (column, kg) — see rt16_forward_kernel<ACT, true>:
//   layer 1, G = 3 t + q            : row of quad Q = 4 t + (i & 3), feature 4 (Q % 13) + (i >> 2) of net Q / 13; input 32 q + level(kg, e),
//                                     level = e < 4 ? 4 kg + e : 16 + 4 kg + (e - 4)          (the two D tiles of a 32-level variable)
//   layer 2, G = 30 + 4 n + 2 u + c : output 4 (4 u + (i & 3)) + (i >> 2) (< 20) of net n; input feature 4 (8 c + e) + kg (< 50)
//   layer 3, G = 42 + 2 n + v       : face 16 v + i (>= 1) of net n; input feature 4 e + kg (e < 5)
// everything else is zero.  The three planes are the exact truncation split of the fp32 weight.
__global__ void rt_pack_split_kernel(const float* __restrict__ img, unsigned* __restrict__ simg) {
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < RT_SIMG_GROUPS * 64 * 4; x += gridDim.x * blockDim.x) {
        const int G = x >> 8, lane = (x >> 2) & 63, pr = x & 3;           // pr: the pair of elements (2 pr, 2 pr + 1)
        const int i = lane & 15, kg = lane >> 4, g_i = i >> 2, r_i = i & 3;
        float v[2];
#pragma unroll
        for (int z = 0; z < 2; z++) {
            const int e = 2 * pr + z;
            float w = 0.0f;
            if (G < 30) {
                const int t = G / 3, q = G - 3 * t, Q = 4 * t + r_i, f = 4 * (Q % 13) + g_i;
                const int lev = e < 4 ? 4 * kg + e : 16 + 4 * kg + (e - 4);
                if (Q < 39 && f < 50) w = img[RT_W1C + ((Q / 13) * 50 + f) * RT_LD1 + 32 * q + lev];
            } else if (G < 42) {
                const int y = G - 30, n = y >> 2, u = (y >> 1) & 1, c = y & 1;
                const int Q2 = 4 * u + r_i, qq = 8 * c + e, f = 4 * qq + kg;
                if (Q2 < 5 && qq < 13 && f < 50) w = img[RT_W2C + (n * 20 + 4 * Q2 + g_i) * RT_LD2 + f];
            } else {
                const int y = G - 42, n = y >> 1, vv = y & 1, face = 16 * vv + i;
                if (face >= 1 && e < 5) w = img[RT_W3C + (n * 31 + face - 1) * RT_LD3 + 4 * e + kg];
            }
            v[z] = w;
        }
        const float r0 = v[0] - __uint_as_float(__float_as_uint(v[0]) & 0xffff0000u), r1 = v[1] - __uint_as_float(__float_as_uint(v[1]) & 0xffff0000u);
        const float l0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u), l1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
        unsigned* o = simg + (size_t)G * 3 * 256 + lane * 4 + pr;
        o[0] = (__float_as_uint(v[1]) & 0xffff0000u) | (__float_as_uint(v[0]) >> 16);
        o[256] = (__float_as_uint(r1) & 0xffff0000u) | (__float_as_uint(r0) >> 16);
        o[512] = (__float_as_uint(l1) & 0xffff0000u) | (__float_as_uint(l0) >> 16);
    }
}

// ... and for the net-split adjoint's W1_n^T products (RT_NSA_*: rt16sh_adjoint_kernel<ACT, RICH, false, true>): group G = (n * 2 + kb) * 6 + tile, lane (i, kq),
// element e <-> W1_n[feature 4 (8 kb + e) + kq][x index 16 tile + i] (zero beyond feature 49); planes h, m interleaved per group, plane l behind them
__global__ void rt_pack_split_nsadj_kernel(const float* __restrict__ img, unsigned* __restrict__ simg) {
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < 36 * 256; x += gridDim.x * blockDim.x) {
        const int G = x >> 8, lane = (x >> 2) & 63, pr = x & 3;
        const int n = G / 12, kb = (G / 6) & 1, tile = G % 6, i = lane & 15, kq = lane >> 4;
        float v[2];
#pragma unroll
        for (int z = 0; z < 2; z++) {
            const int f = 4 * (8 * kb + 2 * pr + z) + kq;
            v[z] = f < 50 ? img[RT_W1C + (n * 50 + f) * RT_LD1 + 16 * tile + i] : 0.0f;
        }
        const float r0 = v[0] - __uint_as_float(__float_as_uint(v[0]) & 0xffff0000u), r1 = v[1] - __uint_as_float(__float_as_uint(v[1]) & 0xffff0000u);
        const float l0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u), l1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
        simg[(size_t)(G * 2 + 0) * 256 + lane * 4 + pr] = (__float_as_uint(v[1]) & 0xffff0000u) | (__float_as_uint(v[0]) >> 16);
        simg[(size_t)(G * 2 + 1) * 256 + lane * 4 + pr] = (__float_as_uint(r1) & 0xffff0000u) | (__float_as_uint(r0) >> 16);
        simg[RT_NSA_L + (size_t)G * 256 + lane * 4 + pr] = (__float_as_uint(l1) & 0xffff0000u) | (__float_as_uint(l0) >> 16);
    }
}

// ... and for the net-split forward kernel (RT_SIMG2_*): layer 1 per net (tiles of that kernel: quad Q = 4 t + (i & 3) of the net's 13, feature
// 4 Q + (i >> 2)), layer 2 as above, and the 8 live lanes of each net's fourth layer-1 tile (quad 12: features 48, 49 -> rows i = 0, 4)
__global__ void rt_pack_split_ns_kernel(const float* __restrict__ img, unsigned* __restrict__ simg) {
    auto put = [&](unsigned* o, int plane_stride, float v0, float v1) {
        const float r0 = v0 - __uint_as_float(__float_as_uint(v0) & 0xffff0000u), r1 = v1 - __uint_as_float(__float_as_uint(v1) & 0xffff0000u);
        const float l0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u), l1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
        o[0] = (__float_as_uint(v1) & 0xffff0000u) | (__float_as_uint(v0) >> 16);
        o[plane_stride] = (__float_as_uint(r1) & 0xffff0000u) | (__float_as_uint(r0) >> 16);
        o[2 * plane_stride] = (__float_as_uint(l1) & 0xffff0000u) | (__float_as_uint(l0) >> 16);
    };
    const int n_full = 39 * 256, n_t3 = 9 * 8 * 4;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < n_full + n_t3 + 4; x += gridDim.x * blockDim.x) {
        if (x >= n_full + n_t3) { simg[RT_SIMG2_ZERO + x - n_full - n_t3] = 0u; continue; }
        float v[2];
        if (x < n_full) {
            const int G = x >> 8, lane = (x >> 2) & 63, pr = x & 3;
            const int i = lane & 15, kg = lane >> 4, g_i = i >> 2, r_i = i & 3;
#pragma unroll
            for (int z = 0; z < 2; z++) {
                const int e = 2 * pr + z;
                float w = 0.0f;
                if (G < 27) {
                    const int n = G / 9, t = (G / 3) % 3, q = G % 3, f = 4 * (4 * t + r_i) + g_i;
                    const int lev = e < 4 ? 4 * kg + e : 16 + 4 * kg + (e - 4);
                    w = img[RT_W1C + (n * 50 + f) * RT_LD1 + 32 * q + lev];
                } else {
                    const int y = G - 27, n = y >> 2, u = (y >> 1) & 1, c = y & 1;
                    const int Q2 = 4 * u + r_i, qq = 8 * c + e, f = 4 * qq + kg;
                    if (Q2 < 5 && qq < 13 && f < 50) w = img[RT_W2C + (n * 20 + 4 * Q2 + g_i) * RT_LD2 + f];
                }
                v[z] = w;
            }
            put(simg + (size_t)G * 768 + lane * 4 + pr, 256, v[0], v[1]);
        } else {
            const int y = x - n_full, blk = y >> 5, l8 = (y >> 2) & 7, pr = y & 3;      // blk = n * 3 + q; l8 = g_i * 4 + kg
            const int n = blk / 3, q = blk % 3, g_i = l8 >> 2, kg = l8 & 3;
#pragma unroll
            for (int z = 0; z < 2; z++) {
                const int e = 2 * pr + z, lev = e < 4 ? 4 * kg + e : 16 + 4 * kg + (e - 4);
                v[z] = img[RT_W1C + (n * 50 + 48 + g_i) * RT_LD1 + 32 * q + lev];
            }
            put(simg + RT_SIMG2_T3 + (size_t)blk * 3 * 32 + l8 * 4 + pr, 32, v[0], v[1]);
        }
    }
}

// ... and the regtile adjoint's W1^T operands (RT_ASIMG_*)
__global__ void rt_pack_split_adj_kernel(const float* __restrict__ img, unsigned* __restrict__ simg) {
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < 27 * 256 + 576; x += gridDim.x * blockDim.x) {
        if (x >= 27 * 256) {
            const int y = x - 27 * 256, n = y / 192, kh = (y / 96) & 1, c = y % 96;
            reinterpret_cast<float*>(simg)[RT_ASIMG_LEFT + y] = img[RT_W1C + (n * 50 + 48 + kh) * RT_LD1 + c];
            continue;
        }
        const int G = x >> 8, lane = (x >> 2) & 63, pr = x & 3;
        const int n = G / 9, c = (G / 3) % 3, q = G % 3, m = lane & 31, kh = lane >> 5;
        const float v0 = img[RT_W1C + (n * 50 + 2 * (8 * c + 2 * pr) + kh) * RT_LD1 + 32 * q + m];
        const float v1 = img[RT_W1C + (n * 50 + 2 * (8 * c + 2 * pr + 1) + kh) * RT_LD1 + 32 * q + m];
        const float r0 = v0 - __uint_as_float(__float_as_uint(v0) & 0xffff0000u), r1 = v1 - __uint_as_float(__float_as_uint(v1) & 0xffff0000u);
        const float l0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u), l1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
        simg[(size_t)(G * 2) * 256 + lane * 4 + pr] = (__float_as_uint(v1) & 0xffff0000u) | (__float_as_uint(v0) >> 16);
        simg[(size_t)(G * 2 + 1) * 256 + lane * 4 + pr] = (__float_as_uint(r1) & 0xffff0000u) | (__float_as_uint(r0) >> 16);
        simg[RT_ASIMG_L + (size_t)G * 256 + lane * 4 + pr] = (__float_as_uint(l1) & 0xffff0000u) | (__float_as_uint(l0) >> 16);
    }
}

// ------------------------------------------------------------------------------------------------
// forward solve (classical RK4); stage inputs -> tape in register-image order
//   tape[((tile*n_steps + step)*4 + stage)*3072 + (q*4 + g)*256 + lane*4 + e] = X[q][4g + e]
// ------------------------------------------------------------------------------------------------
template <int ACT>
__global__ void __launch_bounds__(256)
rt_forward_kernel(DevModel m, const float* __restrict__ wimg, const float* __restrict__ x0, const float* __restrict__ bcs,
                  const float* __restrict__ save_times, int n_save, int substeps, float* __restrict__ sol,
                  float* __restrict__ tape, int n_col) {
    float* wl = rt_smem;
    for (int e = threadIdx.x; e < RT_IMG_FLOATS; e += blockDim.x) wl[e] = wimg[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x * RT_WAVES + wave;
    if (tile * RT_COLS >= n_col) return;                    // no barrier after this point: waves are independent
    const int col = tile * RT_COLS + j;
    const bool valid = col < n_col;
    const int colc = min(col, n_col - 1);
    const RtBases bases = rt_bases(lane);
    RtBC bc;
    float bc5;
    {
        const float* bp = bcs + (size_t)colc * 6;
        bc.b[0] = bp[0]; bc.t[0] = bp[1]; bc.b[1] = bp[2]; bc.t[1] = bp[3]; bc.b[2] = bp[4]; bc5 = bp[5];
        bc.t[2] = bc5;
    }
    f32x16 Xn[3];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const f32x4v v = *reinterpret_cast<const f32x4v*>(x0 + (size_t)colc * 96 + q * 32 + 8 * g + 4 * h);
            Xn[q][4 * g] = v[0]; Xn[q][4 * g + 1] = v[1]; Xn[q][4 * g + 2] = v[2]; Xn[q][4 * g + 3] = v[3];
            if (sol && valid) *reinterpret_cast<f32x4v*>(sol + ((size_t)col * n_save) * 96 + q * 32 + 8 * g + 4 * h) = v;
        }
    const int n_steps = (n_save - 1) * substeps;
    float* tp = tape ? tape + (size_t)tile * n_steps * 4 * 3072 : nullptr;
    int step = 0;
    for (int iv = 0; iv < n_save - 1; iv++) {
        const float t0 = save_times[iv];
        const float dt = (save_times[iv + 1] - t0) / (float)substeps;
        for (int s = 0; s < substeps; s++, step++) {
            const float ts = t0 + (float)s * dt;
            f32x16 Xs[3], Ks[3], Kacc[3];
#pragma unroll
            for (int q = 0; q < 3; q++) { Xs[q] = Xn[q]; Kacc[q] = Xn[q] - Xn[q]; }
#pragma nounroll
            for (int st = 0; st < 4; st++) {
                const float ca = st == 0 ? 0.0f : (st == 3 ? 1.0f : 0.5f);
                const float cb = (st == 0 || st == 3) ? 1.0f / 6.0f : 1.0f / 3.0f;
                if (st > 0) {
#pragma unroll
                    for (int q = 0; q < 3; q++) Xs[q] = Xn[q] + (ca * dt) * Ks[q];
                }
                if (tp) {
                    float* o = tp + ((size_t)step * 4 + st) * 3072 + lane * 4;
#pragma unroll
                    for (int q = 0; q < 3; q++)
#pragma unroll
                        for (int g = 0; g < 4; g++) {
                            f32x4v v = {Xs[q][4 * g], Xs[q][4 * g + 1], Xs[q][4 * g + 2], Xs[q][4 * g + 3]};
                            *reinterpret_cast<f32x4v*>(o + (q * 4 + g) * 256) = v;
                        }
                }
                bc.t[2] = rt_top_flux(m, bc5, ts + ca * dt);
                f32x16 O[3];
                rt_mlp_forward<ACT>(wl, bases, Xs, h, O);
                rt_physics_forward(m, Xs, O, bc, h, Ks);
#pragma unroll
                for (int q = 0; q < 3; q++) Kacc[q] += cb * Ks[q];
            }
#pragma unroll
            for (int q = 0; q < 3; q++) Xn[q] += dt * Kacc[q];
            if (s == substeps - 1 && sol && valid) {
#pragma unroll
                for (int q = 0; q < 3; q++)
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        f32x4v v = {Xn[q][4 * g], Xn[q][4 * g + 1], Xn[q][4 * g + 2], Xn[q][4 * g + 3]};
                        *reinterpret_cast<f32x4v*>(sol + ((size_t)col * n_save + iv + 1) * 96 + q * 32 + 8 * g + 4 * h) = v;
                    }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// adjoint (back-propagation through the RK4 stages), register-resident
// ------------------------------------------------------------------------------------------------
template <int ACT>
__device__ __forceinline__ float rt_act_grad(float z) {
    if (ACT == COLNDE_ACT_RELU) return z > 0.0f ? 1.0f : 0.0f;
    if (ACT == COLNDE_ACT_MISH) {
        const float e = __expf(__builtin_amdgcn_fmed3f(z, -3.0e38f, 20.0f));
        const float n = e * (e + 2.0f);
        const float t = fast_div(n, n + 2.0f);
        const float sg = fast_div(e, 1.0f + e);
        return t + z * (1.0f - t * t) * sg;
    }
    if (ACT == COLNDE_ACT_SWISH) { const float sg = fast_div(1.0f, 1.0f + __expf(-z)); return sg + z * sg * (1.0f - sg); }
    if (ACT == COLNDE_ACT_TANH) { const float e = __expf(2.0f * fminf(fmaxf(z, -15.0f), 15.0f)); const float t = 1.0f - fast_div(2.0f, 1.0f + e); return 1.0f - t * t; }
    if (ACT == COLNDE_ACT_LEAKYRELU) return z > 0.0f ? 1.0f : 0.01f;
    return 1.0f;
}

// activation and its derivative in one evaluation: one exponential and one reciprocal.
// mish(z) = z n / (n + 2) with n = e (e + 2), e = exp(z);  mish'(z) = e w / (n + 2)^2 with
// w = 4 (z + 1) + 4 e^2 + e^3 + e (4 z + 6) = p + e (n + 2 e + p + 2), p = 4 z + 4.
template <int ACT>
__device__ __forceinline__ void rt_act_pair(float z, float& a, float& d) {
    if (ACT == COLNDE_ACT_MISH) {
        const float e = __expf(__builtin_amdgcn_fmed3f(z, -3.0e38f, 20.0f));
        const float n = e * (e + 2.0f);
        const float q = n + 2.0f;
        const float r = __builtin_amdgcn_rcpf(q);
        const float p = 4.0f * z + 4.0f;
        const float w = fmaf(e, fmaf(2.0f, e, q) + p, p);          // n + 2 e + p + 2 = (q + 2 e) + p
        a = z * (n * r);
        d = (e * r) * (w * r);
    } else if (ACT == COLNDE_ACT_SWISH) {
        const float sg = fast_div(1.0f, 1.0f + __expf(-z));
        a = z * sg;
        d = sg + a * (1.0f - sg);
    } else {
        a = rt_act<ACT>(z);
        d = rt_act_grad<ACT>(z);
    }
}

// Two activation value/derivative pairs at once on packed f32 arithmetic (v_pk_mul / v_pk_add / v_pk_fma: two lanes' worth of the same
// instruction per issue slot).  With one wave per SIMD every vector instruction of these phases is exposed, and scalar v_fma runs the
// vector unit at half its rate: 13 packed + 6 scalar (clamp, exp, rcp) instructions for two pairs instead of 34.
template <int ACT>
__device__ __forceinline__ void rt_act_pair2(f32x2v z, f32x2v& a, f32x2v& d) {
    if (ACT == COLNDE_ACT_MISH) {
        f32x2v zc;
        zc.x = __builtin_amdgcn_fmed3f(z.x, -3.0e38f, 20.0f);
        zc.y = __builtin_amdgcn_fmed3f(z.y, -3.0e38f, 20.0f);
        const f32x2v t = zc * 1.4426950408889634f;
        f32x2v e;
        e.x = __builtin_amdgcn_exp2f(t.x);
        e.y = __builtin_amdgcn_exp2f(t.y);
        const f32x2v n = e * (e + 2.0f);
        const f32x2v q = n + 2.0f;
        f32x2v r;
        r.x = __builtin_amdgcn_rcpf(q.x);
        r.y = __builtin_amdgcn_rcpf(q.y);
        const f32x2v p = z * 4.0f + 4.0f;
        const f32x2v w = e * ((e * 2.0f + q) + p) + p;
        a = z * (n * r);
        d = (e * r) * (w * r);
    } else {
        float a0, d0, a1, d1;
        rt_act_pair<ACT>(z.x, a0, d0);
        rt_act_pair<ACT>(z.y, a1, d1);
        a.x = a0; a.y = a1; d.x = d0; d.y = d1;
    }
}

// activation VALUE only (forward kernels), four elements on packed arithmetic
template <int ACT>
__device__ __forceinline__ f32x4v rt_act4(f32x4v z) {
    if (ACT == COLNDE_ACT_MISH) {
        f32x2v z0 = {z[0], z[1]}, z1 = {z[2], z[3]}, c0, c1, e0, e1, r0, r1;
        c0.x = __builtin_amdgcn_fmed3f(z0.x, -3.0e38f, 20.0f); c0.y = __builtin_amdgcn_fmed3f(z0.y, -3.0e38f, 20.0f);
        c1.x = __builtin_amdgcn_fmed3f(z1.x, -3.0e38f, 20.0f); c1.y = __builtin_amdgcn_fmed3f(z1.y, -3.0e38f, 20.0f);
        const f32x2v t0 = c0 * 1.4426950408889634f, t1 = c1 * 1.4426950408889634f;
        e0.x = __builtin_amdgcn_exp2f(t0.x); e0.y = __builtin_amdgcn_exp2f(t0.y);
        e1.x = __builtin_amdgcn_exp2f(t1.x); e1.y = __builtin_amdgcn_exp2f(t1.y);
        const f32x2v n0 = e0 * (e0 + 2.0f), n1 = e1 * (e1 + 2.0f);
        const f32x2v q0 = n0 + 2.0f, q1 = n1 + 2.0f;
        r0.x = __builtin_amdgcn_rcpf(q0.x); r0.y = __builtin_amdgcn_rcpf(q0.y);
        r1.x = __builtin_amdgcn_rcpf(q1.x); r1.y = __builtin_amdgcn_rcpf(q1.y);
        const f32x2v a0 = z0 * (n0 * r0), a1 = z1 * (n1 * r1);
        return (f32x4v){a0.x, a0.y, a1.x, a1.y};
    }
    return (f32x4v){rt_act<ACT>(z[0]), rt_act<ACT>(z[1]), rt_act<ACT>(z[2]), rt_act<ACT>(z[3])};
}

// four at once: two packed flows side by side, so that the results of the transcendental unit are not consumed by the very next instruction
template <int ACT>
__device__ __forceinline__ void rt_act_pair4(f32x2v z0, f32x2v z1, f32x2v& a0, f32x2v& d0, f32x2v& a1, f32x2v& d1) {
    if (ACT == COLNDE_ACT_MISH) {
        f32x2v c0, c1, e0, e1, r0, r1;
        c0.x = __builtin_amdgcn_fmed3f(z0.x, -3.0e38f, 20.0f); c0.y = __builtin_amdgcn_fmed3f(z0.y, -3.0e38f, 20.0f);
        c1.x = __builtin_amdgcn_fmed3f(z1.x, -3.0e38f, 20.0f); c1.y = __builtin_amdgcn_fmed3f(z1.y, -3.0e38f, 20.0f);
        const f32x2v t0 = c0 * 1.4426950408889634f, t1 = c1 * 1.4426950408889634f;
        e0.x = __builtin_amdgcn_exp2f(t0.x); e0.y = __builtin_amdgcn_exp2f(t0.y);
        e1.x = __builtin_amdgcn_exp2f(t1.x); e1.y = __builtin_amdgcn_exp2f(t1.y);
        const f32x2v p0 = z0 * 4.0f + 4.0f, p1 = z1 * 4.0f + 4.0f;
        const f32x2v n0 = e0 * (e0 + 2.0f), n1 = e1 * (e1 + 2.0f);
        const f32x2v q0 = n0 + 2.0f, q1 = n1 + 2.0f;
        r0.x = __builtin_amdgcn_rcpf(q0.x); r0.y = __builtin_amdgcn_rcpf(q0.y);
        r1.x = __builtin_amdgcn_rcpf(q1.x); r1.y = __builtin_amdgcn_rcpf(q1.y);
        const f32x2v w0 = e0 * ((e0 * 2.0f + q0) + p0) + p0, w1 = e1 * ((e1 * 2.0f + q1) + p1) + p1;
        a0 = z0 * (n0 * r0); a1 = z1 * (n1 * r1);
        d0 = (e0 * r0) * (w0 * r0); d1 = (e1 * r1) * (w1 * r1);
    } else {
        rt_act_pair2<ACT>(z0, a0, d0);
        rt_act_pair2<ACT>(z1, a1, d1);
    }
}
template <int ACT>
__device__ __forceinline__ void rt_act_pair4_at(f32x16 (&A)[2], f32x16 (&D)[2], int G) {     // registers G .. G+3 (G % 4 == 0)
    f32x2v z0 = {A[G >> 4][G & 15], A[G >> 4][(G & 15) + 1]}, z1 = {A[G >> 4][(G & 15) + 2], A[G >> 4][(G & 15) + 3]}, a0, d0, a1, d1;
    rt_act_pair4<ACT>(z0, z1, a0, d0, a1, d1);
    A[G >> 4][G & 15] = a0.x; A[G >> 4][(G & 15) + 1] = a0.y; A[G >> 4][(G & 15) + 2] = a1.x; A[G >> 4][(G & 15) + 3] = a1.y;
    D[G >> 4][G & 15] = d0.x; D[G >> 4][(G & 15) + 1] = d0.y; D[G >> 4][(G & 15) + 2] = d1.x; D[G >> 4][(G & 15) + 3] = d1.y;
}

// registers G', G'+1 (G' even: the same tile, adjacent registers) of a two-tile layer-1 block
template <int ACT>
__device__ __forceinline__ void rt_act_pair2_at(f32x16 (&A)[2], f32x16 (&D)[2], int G) {
    f32x2v z = {A[G >> 4][G & 15], A[G >> 4][(G & 15) + 1]}, av, dv;
    rt_act_pair2<ACT>(z, av, dv);
    A[G >> 4][G & 15] = av.x; A[G >> 4][(G & 15) + 1] = av.y;
    D[G >> 4][G & 15] = dv.x; D[G >> 4][(G & 15) + 1] = dv.y;
}

// in place on register G' of a two-tile layer-1 block: A <- act(A), D <- act'(A)
template <int ACT>
__device__ __forceinline__ void rt_act_pair_at(f32x16 (&A)[2], f32x16 (&D)[2], int G) {
    float av, dv;
    rt_act_pair<ACT>(A[G >> 4][G & 15], av, dv);
    A[G >> 4][G & 15] = av;
    D[G >> 4][G & 15] = dv;
}

// Pullback of rt_physics_forward.  On entry kd holds the stage cotangent k̄; on exit it holds dO = the cotangent of the NN
// face fluxes (0 on face 0) and xb the physics part of the state cotangent (flux-divergence transpose + Coriolis).
// Ordered so that k̄ is consumed in place (Coriolis first, then F̄ overwrites k̄): this phase is the kernel's
// register-pressure peak.
__device__ __forceinline__ void rt_physics_vjp(const DevModel& m, const f32x16 (&X)[3], f32x16 (&kd)[3], int h, f32x16 (&xb)[3]) {
    const float Nz = 32.0f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        xb[0][r] = -m.cor_v * m.sig_u * kd[1][r];
        xb[1][r] = m.cor_u * m.sig_v * kd[0][r];
        xb[2][r] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const f32x16 dn = shift_down(kd[k], h, 0.0f);
#pragma unroll
        for (int r = 0; r < 16; r++) kd[k][r] = (r == 0 && h == 0) ? 0.0f : m.A[k] * (kd[k][r] - dn[r]);
    }
    if (!m.mpp && !m.ca) return;
    f32x16 gb[3];
    if (m.mpp) {
        const f32x16 Ud = shift_down(X[0], h, 0.0f), Vd = shift_down(X[1], h, 0.0f), Td = shift_down(X[2], h, 0.0f);
        // Same arithmetic as rt_physics_forward's diffusivity and its derivative, with every uniform factor folded into a handful
        // of wave-uniform constants (this phase has no MFMA beside it: each vector instruction is fully exposed).
        // level differences d = X[r] - X[r-1];  gu = Nz d_u etc.;  a1 = σ_u (gu + ε);  S2 = a1² + a2²;  Ri = B (gT + ε) / S2
        const float cU = m.sig_u * Nz, sU = m.sig_u * m.eps, cV = m.sig_v * Nz, sV = m.sig_v * m.eps, cB = m.B * Nz, sB = m.B * m.eps;
        // tanh((Ri - Riᶜ)/ΔRi) = 1 - 2 / (1 + exp2(kE Ri + oE)), argument clamped to ±30 log2(e)
        const float L2E = 1.4426950408889634f;
        const float kE = 2.0f * m.inv_dRi * L2E, oE = -2.0f * m.Ric * m.inv_dRi * L2E, cE = 30.0f * L2E;
        const float nA = -0.5f * m.nu_minus, nB = m.nu0 + 0.5f * m.nu_minus;                    // ν = nB + nA tanh
        const float m0 = -m.cs[0], m1 = -m.cs[1], m2 = -m.cs[2] * m.inv_Pr;                      // g_k = k̄_k ν m_k  (D = -k̄)
        const float n0 = m0 * Nz * m.c_rib, n1 = m1 * Nz * m.c_rib, n2 = m2 * Nz * m.c_rib;      // c_rib Σ D_k c_k g_k = Σ k̄_k d_k n_k
        const float q0 = -2.0f * m.sig_u, q1 = -2.0f * m.sig_v;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const bool in = !(r == 0 && h == 0);
            const float dU = X[0][r] - Ud[r], dV = X[1][r] - Vd[r], dT = X[2][r] - Td[r];
            const float a1 = fmaf(dU, cU, sU), a2 = fmaf(dV, cV, sV);
            const float rS = __builtin_amdgcn_rcpf(fmaf(a2, a2, a1 * a1));
            const float Ri = fmaf(dT, cB, sB) * rS;
            const float e = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(fmaf(Ri, kE, oE), -cE, cE));
            const float th = fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
            const float nu = fmaf(th, nA, nB);
            const float t0 = kd[0][r] * nu, t1 = kd[1][r] * nu, t2 = kd[2][r] * nu;
            float nub = (kd[0][r] * dU) * n0;
            nub = fmaf(kd[1][r] * dV, n1, nub);
            nub = fmaf(kd[2][r] * dT, n2, nub);
            const float w = nub * (fmaf(-th, th, 1.0f) * rS);                                  // c_rib (1 - tanh²) Σ… / S2
            const float qq = w * Ri;
            const float g2 = fmaf(w, m.B, t2 * m2);
            const float g0 = fmaf(qq, a1 * q0, t0 * m0);
            const float g1 = fmaf(qq, a2 * q1, t1 * m1);
            gb[0][r] = in ? g0 : 0.0f;
            gb[1][r] = in ? g1 : 0.0f;
            gb[2][r] = in ? g2 : 0.0f;
        }
    } else {
        const f32x16 Td = shift_down(X[2], h, 0.0f);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const bool in = !(r == 0 && h == 0);
            const float gT = (X[2][r] - Td[r]) * Nz;
            gb[0][r] = 0.0f;
            gb[1][r] = 0.0f;
            gb[2][r] = (in && gT < 0.0f) ? -kd[2][r] * m.cs[2] * m.kappa : 0.0f;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const f32x16 gu_ = shift_up(gb[k], h, 0.0f);
#pragma unroll
        for (int r = 0; r < 16; r++) xb[k][r] += (gb[k][r] - gu_[r]) * Nz;
    }
}

// Cache policy of the tape streams (written once, read once, 100+ GB per step).  Measured in one process: non-temporal LDS-DMA in the dW1 kernel
// 19.92 -> 19.66 ms, non-temporal delta-tape stores in the adjoint 64.1 -> 63.8 ms; the forward kernel's tape stores showed nothing either way
// and stay default.
#define RT_DMA_AUX 2                                   // aux of global_load_lds: 2 = nt
#ifdef RT_PLAIN_TAPE_STORES      // A/B aid (round 3: in fc32 non-temporal tape STORES were the slower choice; here they are not — profiles/r03l_ab_regtile_nt.log)
#define RT_NT_STORE4(p, v) (*reinterpret_cast<f32x4v*>(p) = (f32x4v)(v))
#define RT_NT_STORE1(p, v) (*(float*)(p) = (float)(v))
#else
#define RT_NT_STORE4(p, v) __builtin_nontemporal_store((f32x4v)(v), reinterpret_cast<f32x4v*>(p))
#define RT_NT_STORE1(p, v) __builtin_nontemporal_store((float)(v), (float*)(p))
#endif
#define RT_TAPE_LOAD4(p) __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(p))   // the adjoint's stage-input and Z1 tape loads: 63.75 -> 63.48 ms
#define RT_TB (32 * 36)   // floats of a wave's transposition tile
// λ in LDS, wave-private: element e of a lane at [e / 4][lane][e % 4] — four consecutive elements are one 16-byte access
#define RT_LAM(e, lane) ((((e) >> 2) * 64 + (lane)) * 4 + ((e) & 3))
// LDS transposition buffer (one 32 x 36 float tile per wave): a D-layout tile (lane = column) becomes the MFMA operand of
// a product contracted over the 32 columns (lane = row, k-step s = columns 2s, 2s+1)
__device__ __forceinline__ f32x16 rt_transpose(float* tb, const f32x16 T, int wbase, int rbase) {
    // tile stored [column][row] with a 36-float column stride: the four registers 4a .. 4a+3 of a lane are rows 8a + 4h .. + 3 of its column —
    // one 16-byte write each; the operand of k-step s, lane (row m, kh), is tile[(2 s + kh)][m]
#pragma unroll
    for (int a = 0; a < 4; a++)
        *reinterpret_cast<f32x4v*>(tb + wbase + 8 * a) = (f32x4v){T[4 * a], T[4 * a + 1], T[4 * a + 2], T[4 * a + 3]};
    f32x16 o;
#pragma unroll
    for (int s = 0; s < 16; s++) o[s] = tb[rbase + 72 * s];
    return o;
}

__device__ __forceinline__ f32x16 rt_outer(f32x16 acc, const f32x16 TA, const f32x16 TB) {
#pragma unroll
    for (int s = 0; s < 16; s++) acc = mfma32(TA[s], TB[s], acc);
    return acc;
}

// A/B build only (-DRT_OUTER_SPLIT, VERDICT r4 task 4b): the same outer product on the bf16 pipe — both transposed tiles split exactly per 16-deep k-block
// (k-block b = registers 8b .. 8b+7 of both operands: the two kernels' K orders agree), six bf16 products per block.  TA's split is passed in so that the
// two layer-1 tiles of dW2 share it.
struct Bf3x2 { Bf3 b[2]; };
__device__ __forceinline__ Bf3x2 rt_split_tile(const f32x16 T) {
    Bf3x2 o;
#pragma unroll
    for (int b = 0; b < 2; b++) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = T[8 * b + u];
        o.b[b] = bf3_split8(v);
    }
    return o;
}
__device__ __forceinline__ f32x16 rt_outer_split(f32x16 acc, const Bf3x2& A, const Bf3x2& B) {
    acc = mfma_bf3(A.b[0], B.b[0], acc);
    acc = mfma_bf3(A.b[1], B.b[1], acc);
    return acc;
}

__device__ __forceinline__ float rt_sum16(const f32x16 v) {
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r++) s += v[r];
    return s;
}

// which nets own registers of layer-1 tile mt (register G = 16 mt + r belongs to net G / 25)
#define RT_NET_LO(mt) (((mt) * 16) / 25)
#define RT_NET_HI(mt) ((((mt) * 16 + 15) < 75 ? ((mt) * 16 + 15) : 74) / 25)
// index of the (net, tile) weight-gradient accumulator: combos (0,0) (0,1) (1,1) (1,2) (1,3) (2,3) (2,4)
__host__ __device__ constexpr int rt_combo(int n, int mt) { return n == 0 ? mt : (n == 1 ? 1 + mt : 2 + mt); }

// tape of stage inputs (written by rt_forward_kernel):  [tile][step][stage][12 groups][64 lanes][4]
// tape2 of layer-1 deltas (written here, read by rt_dw1_kernel): [tile][step][stage][20 groups][64 lanes][4]; element e of
// group grp is stacked register G = 4 grp + e = 25 n + g (features 2g, 2g+1 of net n; G >= 75: never written, never used)
// SPLIT (COLNDE_MATRIX_BF16X3_EXACT; with the Z1 tape only): the W1^T products — 225 of the stage's 552 fp32 MFMAs — on v_mfma_f32_32x32x16_bf16 from exact three-way
// operand splits (csrc/split_bf16.h): the h and m planes of W1^T take the fp32 W1's place in LDS, the l planes come from L2 into registers, the delta
// registers are split per 16-deep k-block; features 48, 49 of a net (register 24) keep one fp32 k-step.
template <int ACT, bool ZT, bool SPLIT = false>
__global__ void __launch_bounds__(256)
rt_adjoint_kernel(DevModel m, const float* __restrict__ wimg, const float* __restrict__ bcs,
                  const float* __restrict__ save_times, int n_save, int substeps, const float* __restrict__ sol,
                  const float* __restrict__ truth, const float* __restrict__ tape, float* __restrict__ tape2,
                  const float* __restrict__ tapez /* layer-1 pre-activations taped by the forward kernel (ZT) */,
                  LossWeights lw, float* __restrict__ slab, int n_col) {
    static_assert(!SPLIT || ZT, "the split W1^T operands replace the fp32 W1 in LDS: no layer-1 recomputation");
    float* wl = rt_smem;
    if constexpr (SPLIT) {
        // LDS: the h, m planes of W1^T and the fp32 rows of features 48, 49 take the fp32 W1's place (14,400 of its 14,552 floats); the fp32 image from
        // W2 on stays where it is, so wl[RT_W2C ...] etc. are unchanged
        static_assert(RT_ASIMG_HM_WORDS + 576 <= RT_W2C, "the split W1^T operands must fit the fp32 W1's place");
        const u32x4* src = reinterpret_cast<const u32x4*>(wimg + RT_ASIMG_OFF);
        for (int e = threadIdx.x; e < (RT_ASIMG_HM_WORDS + 576) / 4; e += blockDim.x) reinterpret_cast<u32x4*>(rt_smem)[e] = src[e];
        for (int e = RT_W2C + threadIdx.x; e < RT_IMG_FLOATS; e += blockDim.x) wl[e] = wimg[e];
    } else {
        for (int e = threadIdx.x; e < RT_IMG_FLOATS; e += blockDim.x) wl[e] = wimg[e];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x * RT_WAVES + wave;
    if (tile * RT_COLS >= n_col) return;
    float* lam = rt_smem + ((RT_IMG_FLOATS + 3) & ~3) + wave * (3072 + RT_TB);   // λ: [48][64] floats, wave-private (16-byte aligned base)
    float* tb = lam + 3072;                                            // transposition tile [32 columns][36]
    const int col = tile * RT_COLS + j;
    const bool valid = col < n_col;
    const int i_ = lane & 31, r_i = (i_ & 3) + 4 * (i_ >> 3), h_i = (i_ >> 2) & 1;
    // per-net layer-1 tiles (n, t): register G' = 16 t + r carries features (2G', 2G'+1) of net n (G' >= 25: padding)
    int a1n[2];
#pragma unroll
    for (int t = 0; t < 2; t++) a1n[t] = RT_W1C + (2 * min(t * 16 + r_i, 24) + h_i) * RT_LD1 + 4 * h;
    const int a2b = RT_W2C + ((r_i < 10) ? 2 * r_i + h_i : 0) * RT_LD2 + h;
    // transposed-product bases
    const int f2c = (r_i < 10) ? 2 * r_i + h_i : 20;                                   // a2 feature of output row i_ (20 = zero column)
    const int b3T = RT_W3C + (4 * h - 1) * RT_LD3 + f2c;
    int b2T[2];
#pragma unroll
    for (int t = 0; t < 2; t++) b2T[t] = RT_W2C + h * RT_LD2 + ((t * 16 + r_i < 25) ? 2 * (t * 16 + r_i) + h_i : 50);   // 50 = zero column
    const int b1T = RT_W1C + h * RT_LD1 + i_;
    const int wbase = j * 36 + 4 * h, rbase = h * 36 + i_;

    f32x16 gW3[3], gW2[3][2];
#pragma unroll
    for (int q = 0; q < 3; q++) { gW3[q] = (f32x16)(0.0f); gW2[q][0] = (f32x16)(0.0f); gW2[q][1] = (f32x16)(0.0f); }
    float b2acc[3] = {0, 0, 0}, b3acc[3] = {0, 0, 0};
    float sums[6] = {0, 0, 0, 0, 0, 0};
    RT_STAMP_DECL;
    f32x16 xb[3], X[3];
#pragma unroll
    for (int q = 0; q < 3; q++) xb[q] = (f32x16)(0.0f);
    f32x16 xbs[3];
#pragma unroll
    for (int q = 0; q < 3; q++) xbs[q] = (f32x16)(0.0f);
#pragma unroll
    for (int e = 0; e < 48; e++) lam[RT_LAM(e, lane)] = 0.0f;

    const int n_steps = (n_save - 1) * substeps;
    const float* tp = tape + (size_t)tile * n_steps * 4 * 3072 + lane * 4;
    float* tp2 = tape2 + (size_t)tile * n_steps * 4 * RT_TAPE2 + lane * 4;
    const float* tpz = ZT ? tapez + (size_t)tile * n_steps * 4 * RT_TAPEZ + lane * 4 : nullptr;

    // loss injection at save point n: λ += ∂loss/∂sol[:, n]; also the six raw sums of squares
    auto inject = [&](int n, bool add) {
#pragma unroll
        for (int q = 0; q < 3; q++) {
            f32x16 d;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const size_t o = ((size_t)min(col, n_col - 1) * n_save + n) * 96 + q * 32 + 8 * g + 4 * h;
                const f32x4v a = *reinterpret_cast<const f32x4v*>(sol + o);
                const f32x4v b = *reinterpret_cast<const f32x4v*>(truth + o);
#pragma unroll
                for (int e = 0; e < 4; e++) d[4 * g + e] = valid ? a[e] - b[e] : 0.0f;
            }
            const f32x16 dd = shift_down(d, h, 0.0f);
            f32x16 gg;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                gg[r] = (r == 0 && h == 0) ? 0.0f : (d[r] - dd[r]) * 32.0f;
                sums[q] += d[r] * d[r];
                sums[3 + q] += gg[r] * gg[r];
            }
            if (add) {
                const f32x16 gu_ = shift_up(gg, h, 0.0f);
#pragma unroll
                for (int r = 0; r < 16; r++)
                    lam[RT_LAM(q * 16 + r, lane)] += 2.0f * lw.w[q] * d[r] + 2.0f * lw.w[3 + q] * 32.0f * (gg[r] - gu_[r]);
            }
        }
    };
    inject(0, false);

    // taped layer-1 pre-activations of net n at (step, stage): registers G' < 25 of Z (padding registers untouched)
    // (RT_ZRICH: Z receives the taped activation values and Dz their derivatives; nothing is left to evaluate)
    auto load_z1 = [&](int step, int st, int n, f32x16 (&Z)[2], f32x16 (&Dz)[2]) {
        size_t rec = (size_t)step * 4 + st;
#ifdef RT_ABL_Z1_CACHED   // (ablation probe, results wrong: every stage reads the tile's FIRST Z1 record — the loads stay, the 50.7 GB of HBM reads go; VERDICT r4 task 4a)
        { int z_ = 0; asm volatile("" : "+v"(z_)); rec = (size_t)z_; }      // (opaque, so that the loads are not hoisted out of the time loop)
#endif
        const float* srcz = tpz + rec * RT_TAPEZ + n * 7 * 256;
#pragma unroll
        for (int grp = 0; grp < 7; grp++) {
            const f32x4v v = RT_TAPE_LOAD4(srcz + grp * 256);
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (4 * grp + e < 25) Z[grp >> 2][(grp & 3) * 4 + e] = v[e];
        }
        if (RT_ZRICH) {
#pragma unroll
            for (int grp = 0; grp < 7; grp++) {
                const f32x4v v = RT_TAPE_LOAD4(srcz + RT_TAPEZ_HALF + grp * 256);
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (4 * grp + e < 25) Dz[grp >> 2][(grp & 3) * 4 + e] = v[e];
            }
        }
    };

    // taped layer-2 pre-activations of net n (RT_Z2TAPE): registers r < 10 of Z (the rest zero)
    auto load_z2 = [&](int step, int st, int n, f32x16& Z) {
        const float* srcz = tpz + ((size_t)step * 4 + st) * RT_TAPEZ + RT_Z2OFF + n * 3 * 256;
#pragma unroll
        for (int grp = 0; grp < 3; grp++) {
            const f32x4v v = RT_TAPE_LOAD4(srcz + grp * 256);
#pragma unroll
            for (int e = 0; e < 4; e++) Z[4 * grp + e] = (4 * grp + e < 10) ? v[e] : 0.0f;
        }
#pragma unroll
        for (int r = 12; r < 16; r++) Z[r] = 0.0f;
    };

    // stage input (taped by the forward kernel)
    auto load_x = [&](int step, int st) {
        const float* src = tp + ((size_t)step * 4 + st) * 3072;
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x4v v = RT_TAPE_LOAD4(src + (q * 4 + g) * 256);
                X[q][4 * g] = v[0]; X[q][4 * g + 1] = v[1]; X[q][4 * g + 2] = v[2]; X[q][4 * g + 3] = v[3];
            }
    };

    for (int iv = n_save - 2; iv >= 0; iv--) {
        const float dt = (save_times[iv + 1] - save_times[iv]) / (float)substeps;
        inject(iv + 1, true);
        for (int s = substeps - 1; s >= 0; s--) {
            const int step = iv * substeps + s;
#pragma nounroll
            for (int st = 3; st >= 0; st--) {
                const float cwl = (st == 0 || st == 3) ? dt / 6.0f : dt / 3.0f;
                const float cwx = st == 3 ? 0.0f : (st == 2 ? dt : 0.5f * dt);
                RT_STAMP_BEGIN();
                f32x16 A1[2], D1[2];        // net n: act(z1), act'(z1) (then dZ1); register G' = 16 t + r <-> features 2G', 2G'+1
                f32x16 A1n[2], D1n[2];      // net n + 1, in flight under net n's W1^T products (ZT)
#pragma unroll
                for (int r = 9; r < 16; r++) { A1[1][r] = 0.0f; D1[1][r] = 0.0f; A1n[1][r] = 0.0f; D1n[1][r] = 0.0f; }
                load_x(step, st);
                // (1) stage cotangent and the physics pullback: dO = cotangent of the NN fluxes, xb = physics part of x̄
                f32x16 dOk[3];
                {
                    f32x16 kb[3];
#pragma unroll
                    for (int q = 0; q < 3; q++)
#pragma unroll
                        for (int r = 0; r < 16; r++) kb[q][r] = cwl * lam[RT_LAM(q * 16 + r, lane)] + cwx * xb[q][r];
                    rt_physics_vjp(m, X, kb, h, xb);       // kb now holds dO
#pragma unroll
                    for (int q = 0; q < 3; q++) dOk[q] = kb[q];
                }
                RT_STAMP(0);
                float* dst = tp2 + ((size_t)step * 4 + st) * RT_TAPE2;
                float carry[2] = {0.0f, 0.0f};
                // the nets are handled one after the other so that only one net's hidden state is live at a time
#pragma unroll
                for (int n = 0; n < 3; n++) {
                    const f32x16 dOn = dOk[n];
                    // (2) hidden layer 1 of net n: activation A1 (feeds layer 2 and the dW2 products) and derivative D1 (feeds dZ1),
                    //     evaluated once, together.  With the Z1 tape (ZT) nets 1 and 2 arrive already activated: their taped
                    //     pre-activations were fetched and activated in the shadow of the previous net's W1^T products (6).
                    if (ZT) {
                        if (n == 0) {
                            load_z1(step, st, 0, A1, D1);      // (requested BEFORE the physics pullback instead: 55.9 -> 57.0 ms, round 5)
#ifdef COLNDE_STAMPS
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            RT_STAMP(7);            // diagnostic build only: the exposed latency of net 0's Z1 loads
#endif
                            if (!RT_ZRICH) {
#pragma unroll
                                for (int G = 0; G < 24; G += 4) rt_act_pair4_at<ACT>(A1, D1, G);
                                rt_act_pair_at<ACT>(A1, D1, 24);
                            }
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < 2; t++) {
                            f32x16 acc;
#pragma unroll
                            for (int r = 0; r < 16; r++) acc[r] = (t * 16 + r < 25) ? wl[RT_B1C + n * 50 + 2 * (t * 16 + r) + h] : 0.0f;
                            const int base = a1n[t] + n * 50 * RT_LD1;
                            acc = rt_chain<48, RT_ADJ_CH>(wl, acc, [=](int k) { return base + (k >> 4) * 32 + RHO0(k & 15); },
                                                          [&](int k) { return X[k >> 4][k & 15]; });
#pragma unroll
                            for (int r = 0; r < 16; r++) {
                                float av = 0.0f, dv = 0.0f;
                                if (t * 16 + r < 25) rt_act_pair<ACT>(acc[r], av, dv);
                                A1[t][r] = av;
                                D1[t][r] = dv;
                            }
                        }
                    }
                    RT_STAMP(1);
                    f32x16 Z2;
                    if (ZT && RT_Z2TAPE) load_z2(step, st, n, Z2);      // taped by the forward kernel: in flight under the dO transposition and the W3^T chain below
                    else {
                        f32x16 acc;
#pragma unroll
                        for (int r = 0; r < 16; r++) acc[r] = r < 10 ? wl[RT_B2C + n * 20 + 2 * r + h] : 0.0f;
                        const int base2 = a2b + n * 20 * RT_LD2;
                        Z2 = rt_chain<25, 5>(wl, acc, [=](int k) { return base2 + 2 * k; },
                                             [&](int k) { return A1[k >> 4][k & 15]; });
                    }
                    RT_STAMP(2);
                    // (3) layer 3: weight/bias gradient, then dZ2 = (W3^T dO) .* act'(Z2)
                    //     (evaluating the layer-2 activation pairs inside the W3^T chain instead was measured slower: 94.8 vs 89.7 ms; W3^T first, so that the taped
                    //      Z2 has the chain to land under: more spills, slower — round 5)
                    {
#ifndef RT_ABL_NO_OUTER     // (ablation probe, results wrong: what the dW2 / dW3 outer products and their five LDS transpositions per net cost the adjoint — DESIGN section 0.6)
                        const f32x16 TA = rt_transpose(tb, dOn, wbase, rbase);
                        b3acc[n] += rt_sum16(TA);
#endif
                        f32x16 A2;
#pragma unroll
                        for (int r = 0; r < 16; r += 4) {
                            f32x2v a0 = {0.0f, 0.0f}, d0 = {0.0f, 0.0f}, a1 = {0.0f, 0.0f}, d1 = {0.0f, 0.0f};
                            if (r < 8) rt_act_pair4<ACT>((f32x2v){Z2[r], Z2[r + 1]}, (f32x2v){Z2[r + 2], Z2[r + 3]}, a0, d0, a1, d1);
                            else if (r == 8) rt_act_pair2<ACT>((f32x2v){Z2[8], Z2[9]}, a0, d0);
                            A2[r] = a0.x; A2[r + 1] = a0.y; A2[r + 2] = a1.x; A2[r + 3] = a1.y;
                            Z2[r] = d0.x; Z2[r + 1] = d0.y; Z2[r + 2] = d1.x; Z2[r + 3] = d1.y;     // Z2 now holds act'(z2)
                        }
#ifndef RT_ABL_NO_OUTER
                        const f32x16 TB = rt_transpose(tb, A2, wbase, rbase);
#ifdef RT_OUTER_SPLIT
                        if constexpr (SPLIT) gW3[n] = rt_outer_split(gW3[n], rt_split_tile(TA), rt_split_tile(TB));
                        else
#endif
                        gW3[n] = rt_outer(gW3[n], TA, TB);
#else
                        gW3[n][0] += A2[0];
#endif
                    }
                    {
                        const int base = b3T + n * 31 * RT_LD3;
                        const f32x16 da = rt_chain<16, 8>(wl, (f32x16)(0.0f), [=](int k) { return base + RHO0(k) * RT_LD3; },
                                                          [&](int k) { return dOn[k]; });
#pragma unroll
                        for (int r = 0; r < 16; r++) Z2[r] = da[r] * Z2[r];
                    }
                    RT_STAMP(3);
                    // fetch net n + 1's taped pre-activations now: dW2, W2^T and the first W1^T chunk cover the HBM latency
                    if (ZT && n < 2) load_z1(step, st, n + 1, A1n, D1n);
                    // (SPLIT) the nine l-plane operands of this net's W1^T products: from L2, issued before the dW2 outer products (one phase earlier than needed: 56.5 -> 55.6 ms)
#ifndef RT_LPIPE
#define RT_LPIPE 1        // round 5: the l planes fetched one k-block (three operands) ahead inside the W1^T products instead of all nine before dW2: 12-24 live registers instead of 36, scratch 536 -> 500 B, adjoint 56.2 -> 55.6 ms (0: A/B build)
#endif
                    u32x4 Lr[9];
                    const u32x4* lg = nullptr;
                    if constexpr (SPLIT) {
                        // (an opaque lane index per net: left loop-invariant, the 27 loads are hoisted out of the time loop and their 108 registers spilled)
                        int lz = lane;
                        asm volatile("" : "+v"(lz));
                        lg = reinterpret_cast<const u32x4*>(wimg + RT_ASIMG_OFF + RT_ASIMG_L) + n * 9 * 64 + lz;
#pragma unroll
                        for (int u = 0; u < (RT_LPIPE ? 3 : 9); u++) Lr[u] = lg[u * 64];
                    }
                    // (4) layer 2: weight/bias gradient
#ifdef RT_ABL_NO_OUTER
                    gW2[n][0][0] += Z2[0] * A1[0][0]; gW2[n][1][0] += Z2[1] * A1[1][0];
#else
                    {
                        const f32x16 TA = rt_transpose(tb, Z2, wbase, rbase);
                        b2acc[n] += rt_sum16(TA);
#ifdef RT_OUTER_SPLIT
                        Bf3x2 TAs;
                        if constexpr (SPLIT) TAs = rt_split_tile(TA);
#endif
#pragma unroll
                        for (int t = 0; t < 2; t++) {
                            const f32x16 TB = rt_transpose(tb, A1[t], wbase, rbase);
#ifdef RT_OUTER_SPLIT
                            if constexpr (SPLIT) gW2[n][t] = rt_outer_split(gW2[n][t], TAs, rt_split_tile(TB));
                            else
#endif
                            gW2[n][t] = rt_outer(gW2[n][t], TA, TB);
                        }
                    }
#endif
                    RT_STAMP(4);
                    // (5) dZ1 = (W2^T dZ2) .* act'(Z1), in place; taped for the streaming dW1 kernel
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        const int base = b2T[t] + n * 20 * RT_LD2;
                        const f32x16 da = rt_chain<10, 10>(wl, (f32x16)(0.0f), [=](int k) { return base + 2 * k * RT_LD2; },
                                                           [&](int k) { return Z2[k]; });
#pragma unroll
                        for (int r = 0; r < 16; r++) D1[t][r] = da[r] * D1[t][r];     // dZ1 in place of act'(z1)
                    }
                    auto store_delta = [&] {
                    // delta tape: register g of net n is stacked register G = 25 n + g, element G & 3 of 16-byte group G >> 2.  Whole groups
                    // leave with one store; the groups that straddle a net boundary (6: G 24 | 25..27, 12: G 48, 49 | 50, 51) wait in
                    // `carry` for the next net's first registers, so a stage writes 19 16-byte stores and no scattered dwords
#pragma unroll
                    for (int g = 0; g < 25; g++) {
                        const int G = 25 * n + g;
                        if ((G & 3) == 0 && g + 3 < 25) {
                            const f32x4v v = {D1[g >> 4][g & 15], D1[(g + 1) >> 4][(g + 1) & 15], D1[(g + 2) >> 4][(g + 2) & 15],
                                              D1[(g + 3) >> 4][(g + 3) & 15]};
                            RT_NT_STORE4(dst + (G >> 2) * 256, v);
                        }
                    }
                    if (n == 0) carry[0] = D1[1][8];                                                   // G 24
                    if (n == 1) {
                        RT_NT_STORE4(dst + 6 * 256, ((f32x4v){carry[0], D1[0][0], D1[0][1], D1[0][2]}));   // G 24 | 25, 26, 27
                        carry[0] = D1[1][7];                                                           // G 48
                        carry[1] = D1[1][8];                                                           // G 49
                    }
                    if (n == 2) {
                        RT_NT_STORE4(dst + 12 * 256, ((f32x4v){carry[0], carry[1], D1[0][0], D1[0][1]})); // G 48, 49 | 50, 51
                        RT_NT_STORE4(dst + 18 * 256, ((f32x4v){D1[1][6], D1[1][7], D1[1][8], 0.0f}));     // G 72, 73, 74, pad
                    }
                    };
#ifndef RT_ABL_NO_TAPE2     // (ablation probe, results wrong: what the delta tape's stores cost the adjoint — DESIGN §6)
                    if constexpr (!SPLIT) store_delta();       // (SPLIT: behind the W1^T products — vmcnt is in order, and the l-plane loads must not queue behind these stores)
#endif
                    RT_STAMP(5);
                    // (6) x̄ += W1_n^T dZ1_n
                    if constexpr (SPLIT) {
                        const u32x4* hm = reinterpret_cast<const u32x4*>(rt_smem) + lane;
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            if (RT_LPIPE && c < 2) {
#pragma unroll
                                for (int u = 0; u < 3; u++) Lr[3 * (c + 1) + u] = lg[(3 * (c + 1) + u) * 64];
                            }
                            float d8[8];
#pragma unroll
                            for (int u = 0; u < 8; u++) d8[u] = D1[(8 * c + u) >> 4][(8 * c + u) & 15];
                            const Bf3 Bc = bf3_split8(d8);
#pragma unroll
                            for (int q = 0; q < 3; q++) {
                                const int G = n * 9 + c * 3 + q, slot = c * 3 + q;
                                const u32x4 Ah = hm[(G * 2) * 64], Am = hm[(G * 2 + 1) * 64];
                                // net n + 1's 25 activation pairs, spread over slots 2 .. 8 (as in the fp32 chain: not under the first chunks)
                                if (ZT && !RT_ZRICH && n < 2 && slot >= 2) {
                                    if (slot < 8) rt_act_pair4_at<ACT>(A1n, D1n, 4 * (slot - 2));
                                    else rt_act_pair_at<ACT>(A1n, D1n, 24);
                                }
                                xb[q] = mfma_bf(Am, Bc.m, xb[q]);
                                xb[q] = mfma_bf(Ah, Bc.l, xb[q]);
                                xb[q] = mfma_bf(Am, Bc.h, xb[q]);
                                xb[q] = mfma_bf(Ah, Bc.m, xb[q]);
                                xb[q] = mfma_bf(Ah, Bc.h, xb[q]);
                                xb[q] = mfma_bf(Lr[slot], Bc.h, xb[q]);          // the operand from L2 last
                                RT_SCHED_FENCE();
                            }
                        }
                        // register 24 (features 48, 49): one fp32 k-step per state tile
                        const float* left = rt_smem + RT_ASIMG_LEFT + (n * 2 + h) * 96 + i_;
#pragma unroll
                        for (int q = 0; q < 3; q++) xb[q] = mfma32(left[q * 32], D1[1][8], xb[q]);
                        store_delta();
                    } else
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        const int base = b1T + q * 32 + n * 50 * RT_LD1;
                        xb[q] = rt_chain_fill<25, 5>(wl, xb[q], [=](int g) { return base + 2 * g * RT_LD1; },
                                                     [&](int g) { return D1[g >> 4][g & 15]; },
                                                     [&](int c) {
                                                         // net n + 1's 25 activation pairs, spread over the 15 chunks
                                                         if (ZT && !RT_ZRICH && n < 2 && (q > 0 || c > 0)) {
#pragma unroll
                                                             for (int G = 0; G < 24; G += 4)
                                                                 if ((G >> 2) * 2 + 2 == q * 5 + c)
                                                                     rt_act_pair4_at<ACT>(A1n, D1n, G);
                                                             if (q * 5 + c == 14) rt_act_pair_at<ACT>(A1n, D1n, 24);
                                                         }
                                                     });
                    }
                    if (ZT && n < 2) {
#pragma unroll
                        for (int t = 0; t < 2; t++) { A1[t] = A1n[t]; D1[t] = D1n[t]; }
                    }
                }
                RT_STAMP(6);
#pragma unroll
                for (int q = 0; q < 3; q++) xbs[q] += xb[q];
            }
            // λ_n = λ_{n+1} + x̄_1 + x̄_2 + x̄_3 + x̄_4
#pragma unroll
            for (int q = 0; q < 3; q++) {
#pragma unroll
                for (int r = 0; r < 16; r++) lam[RT_LAM(q * 16 + r, lane)] += xbs[q][r];
                xbs[q] = (f32x16)(0.0f);
            }
        }
    }

#ifndef COLNDE_STAMPS_FWD
    RT_STAMP_FLUSH();
#endif
    // ---- flush this wave's partial gradients (row `tile` of the slab; dW1/db1 come from rt_dw1_kernel) ----
    float* out = slab + (size_t)tile * (m.n_params + 8);
    const int r_j = (j & 3) + 4 * (j >> 3), h_j = (j >> 2) & 1;     // this lane as a column index n' of a D tile
#pragma unroll
    for (int n = 0; n < 3; n++) {
        // dW3_n[out = face-1][in = f2]: D[m = face rho(r,h)][n' = a2 row j]
        if (r_j < 10) {
            const int f2 = 2 * r_j + h_j;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int face = RHO0(r) + 4 * h;
                if (face >= 1) out[n * m.net_size + m.w_off[2] + f2 * 31 + face - 1] = gW3[n][r];
            }
        }
#pragma unroll
        for (int t = 0; t < 2; t++) {
            // dW2_n[out = f2 = 2r+h][in = f1]: D[m = z2 row rho(r,h)][n' = row j of net n's layer-1 tile t]
            const int Gp = t * 16 + r_j;
            if (Gp < 25) {
                const int f1 = 2 * Gp + h_j;
#pragma unroll
                for (int r = 0; r < 10; r++) out[n * m.net_size + m.w_off[1] + f1 * 20 + 2 * r + h] = gW2[n][t][r];
            }
        }
        // biases: lanes (i, 0) and (i, 1) hold the even / odd column halves of row i_
        const float s2 = b2acc[n] + swap32(b2acc[n], h), s3 = b3acc[n] + swap32(b3acc[n], h);
        if (h == 0) {
            if (r_i < 10) out[n * m.net_size + m.b_off[1] + 2 * r_i + h_i] = s2;
            if (i_ >= 1) out[n * m.net_size + m.b_off[2] + i_ - 1] = s3;
        }
    }
#pragma unroll
    for (int q = 0; q < 6; q++) {
        float v = sums[q];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) out[m.n_params + q] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// dW1 / db1: a streaming GEMM over the two tapes, contracted over (column, step, stage)
//   dW1[n*50+f][c] = sum dZ1[(n,f)][col] * X[c][col]
// Each wave walks items (tile, step, stage) and accumulates all 15 (layer-1 tile, state tile) products in registers; its
// partial result is one slab row.  An item's two tape records are register images of the kernels that wrote them
// (lane = column, registers = rows); the products contract over columns, so their MFMA operands are the transposed tiles.
// The transposition costs nothing here: the record goes HBM -> LDS by LDS-DMA (global_load_lds, 16 bytes per lane, no
// registers), and because the DMA's SOURCE address is per lane while its destination is lane-linear, each 1-KB
// wave-instruction gathers the 16-byte pieces (4 consecutive rows of one column) of 8 columns so that they land as a
// column-major tile [col][row] — the operand of k-step s, lane (row m, kh), is then the conflict-free read
// tile[(2 s + kh) * 32 + m].  The unit of the pipeline is a QUARTER item (8 columns of all 8 tiles: 8 DMA instructions,
// 8 KB, 60 MFMAs): a ring of four quarter buffers per wave, DMA issued three quarters ahead (counted vmcnt), operands
// read one quarter ahead into the other of two register sets — so neither the HBM latency nor the LDS reads are exposed.
// ------------------------------------------------------------------------------------------------
#define RT_DW1_LDS (4 * 2048)      // floats per wave: ring of four quarter-item buffers [8 tiles][8 cols][32 rows]
__global__ void __launch_bounds__(256)
rt_dw1_kernel(DevModel m, const float* __restrict__ tape, const float* __restrict__ tape2, long n_items,
              float* __restrict__ slab_rows) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    float* buf = rt_smem + wave * RT_DW1_LDS;
    const long gw = (long)blockIdx.x * RT_WAVES + wave, GW = (long)gridDim.x * RT_WAVES;
    f32x16 gW1[5][3];
#pragma unroll
    for (int mt = 0; mt < 5; mt++)
#pragma unroll
        for (int q = 0; q < 3; q++) gW1[mt][q] = (f32x16)(0.0f);
    float b1acc[5] = {0, 0, 0, 0, 0};
    const int NQ = gw < n_items ? 4 * (int)((n_items - gw + GW - 1) / GW) : 0;       // this wave's quarters
    // DMA piece i of tile T: lane L fetches rows 4 k .. 4 k + 3 (k = L & 7) of column 8 i + (L >> 3), which the record holds in
    // group 4 T + (k >> 1), lane (k & 1) * 32 + column
    const int soff = ((lane >> 1) & 3) * 256 + ((lane & 1) * 32 + (lane >> 3)) * 4;
    // quarter Q of this wave into ring slot `ring`; past the end the last quarter is fetched again (into a slot nobody reads), so that
    // the counted waits below always see the same number of DMAs in flight
    auto issue = [&](int Q, int ring) {
        const int Qc = Q < NQ ? Q : NQ - 1;
        const size_t item = (size_t)(gw + (long)(Qc >> 2) * GW);
        const float* sx = tape + item * 3072 + soff + (Qc & 3) * 32;
        const float* sz = tape2 + item * RT_TAPE2 + soff + (Qc & 3) * 32;
#pragma unroll
        for (int T = 0; T < 3; T++) __builtin_amdgcn_global_load_lds(sx + T * 1024, buf + ring * 2048 + T * 256, 16, 0, RT_DMA_AUX);
#pragma unroll
        for (int T = 0; T < 5; T++) __builtin_amdgcn_global_load_lds(sz + T * 1024, buf + ring * 2048 + (3 + T) * 256, 16, 0, RT_DMA_AUX);
    };
    const float* rd = buf + h * 32 + j;
    float op[2][8][4];                         // operand sets: [set][tile][k-step]: tile[(2 s + kh) * 32 + m]
    auto read_set = [&](int set, int ring) {
#pragma unroll
        for (int T = 0; T < 8; T++)
#pragma unroll
            for (int s = 0; s < 4; s++) op[set][T][s] = rd[ring * 2048 + T * 256 + 64 * s];
    };
    if (NQ > 0) {
        issue(0, 0);
        issue(1, 1);
        issue(2, 2);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        read_set(0, 0);
    }
    for (int Qb = 0; Qb < NQ; Qb += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int cur = u & 1;
            // first delta tile of this quarter; behind it the DMA three quarters ahead and the operand reads one quarter ahead
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int q = 0; q < 3; q++) gW1[0][q] = mfma32(op[cur][3][s], op[cur][q][s], gW1[0][q]);
            b1acc[0] += (op[cur][3][0] + op[cur][3][1]) + (op[cur][3][2] + op[cur][3][3]);
            __builtin_amdgcn_sched_barrier(0);
            issue(Qb + u + 3, (u + 3) & 3);
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");       // quarter Qb + u + 1 has landed (two younger ones in flight)
            read_set(cur ^ 1, (u + 1) & 3);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 1; mt < 5; mt++) {
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int q = 0; q < 3; q++) gW1[mt][q] = mfma32(op[cur][3 + mt][s], op[cur][q][s], gW1[mt][q]);
                b1acc[mt] += (op[cur][3 + mt][0] + op[cur][3 + mt][1]) + (op[cur][3 + mt][2] + op[cur][3 + mt][3]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* out = slab_rows + (size_t)gw * (m.n_params + 8);
    // D[m = layer-1 row rho(r,h) of tile mt][n' = state feature 32 q + j]
#pragma unroll
    for (int mt = 0; mt < 5; mt++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int G = mt * 16 + r;
            if (G < 75) {
                const int n = G / 25, f = 2 * (G % 25) + h;
#pragma unroll
                for (int q = 0; q < 3; q++) out[n * m.net_size + m.w_off[0] + (q * 32 + j) * 50 + f] = gW1[mt][q][r];
            }
        }
        const float s1 = b1acc[mt] + swap32(b1acc[mt], h);
        const int r_j = (j & 3) + 4 * (j >> 3), h_j = (j >> 2) & 1;
        const int G = mt * 16 + r_j;
        if (h == 0 && G < 75) out[(G / 25) * m.net_size + m.b_off[0] + 2 * (G % 25) + h_j] = s1;
    }
}

// ------------------------------------------------------------------------------------------------
// dW1 / db1 on the bf16 matrix pipe with EXACT three-way operand splitting (COLNDE_MATRIX_BF16X3_EXACT, the default)
//
// A float has 24 significant bits = three bf16 (8 bits each): x = x_h + x_m + x_l exactly, by truncation (x_h = the top half of the
// word, r = x - x_h is exact, x_m = the top half of r, x_l = r - x_m has at most 8 significant bits).  A product a b is then the nine
// products of the parts, each EXACT in fp32 (8 x 8 bits), accumulated in fp32 by v_mfma_f32_32x32x16_bf16; the six of them down to
// 2^-16 are kept (hh, hm, mh, hl, lh, mm), the three dropped ones are below 2^-23 |a b| together — the size of ONE fp32 rounding of
// the product, which the fp32 MFMA chain commits at every accumulation anyway.  Six 32-cycle instructions cover 16 k-steps that cost
// eight 64-cycle v_mfma_f32_32x32x2_f32: 192 cycles instead of 512.
//
// Same tapes, same LDS-DMA gather and the same [col][row] tiles as rt_dw1_kernel.  The unit is HALF an item (16 columns = one
// k-block): lane (row j, kh) contracts k = 8 kh + c with column c of quarter 2 H + kh, i.e. it reads eight floats of ONE quarter
// buffer (conflict-free, stride 32) per tile, splits them in registers (5.5 vector operations per value, issued under the MFMAs) and
// feeds 90 MFMAs.  Ring of four quarter buffers as before: the two quarters of half H + 2 are fetched into the slots of half H as soon
// as its operands are in registers.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
rt_dw1_split_kernel(DevModel m, const float* __restrict__ tape, const float* __restrict__ tape2, long n_items,
                    float* __restrict__ slab_rows) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    float* buf = rt_smem + wave * RT_DW1_LDS;
    const long gw = (long)blockIdx.x * RT_WAVES + wave, GW = (long)gridDim.x * RT_WAVES;
    f32x16 gW1[5][3];
#pragma unroll
    for (int mt = 0; mt < 5; mt++)
#pragma unroll
        for (int q = 0; q < 3; q++) gW1[mt][q] = (f32x16)(0.0f);
    float b1acc[5] = {0, 0, 0, 0, 0};
    const int NQ = gw < n_items ? 4 * (int)((n_items - gw + GW - 1) / GW) : 0;       // this wave's quarters
    const int soff = ((lane >> 1) & 3) * 256 + ((lane & 1) * 32 + (lane >> 3)) * 4;  // the gather of rt_dw1_kernel
    auto issue = [&](int Q, int ring) {
        const int Qc = Q < NQ ? Q : NQ - 1;
        const size_t item = (size_t)(gw + (long)(Qc >> 2) * GW);
        const float* sx = tape + item * 3072 + soff + (Qc & 3) * 32;
        const float* sz = tape2 + item * RT_TAPE2 + soff + (Qc & 3) * 32;
#pragma unroll
        for (int T = 0; T < 3; T++) __builtin_amdgcn_global_load_lds(sx + T * 1024, buf + ring * 2048 + T * 256, 16, 0, RT_DMA_AUX);
#pragma unroll
        for (int T = 0; T < 5; T++) __builtin_amdgcn_global_load_lds(sz + T * 1024, buf + ring * 2048 + (3 + T) * 256, 16, 0, RT_DMA_AUX);
    };
    const float* rd = buf + h * 2048 + j;        // lane half kh reads the quarter buffer (slot0 + kh)
    float raw[2][8][8];                          // [set][tile][column of the quarter]
    auto read_half = [&](int set, int slot0) {
#pragma unroll
        for (int T = 0; T < 8; T++)
#pragma unroll
            for (int c = 0; c < 8; c++) raw[set][T][c] = rd[slot0 * 2048 + T * 256 + c * 32];
    };
    // Pipeline: while half H is multiplied, half H + 1 sits in registers (read at the top of H) and halves H + 2, H + 3 are in flight —
    // a half's two slots are refilled as soon as its operands have been read (the whole ring is in flight during the MFMAs).
    if (NQ > 0) {
        issue(0, 0);
        issue(1, 1);
        issue(2, 2);
        issue(3, 3);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        read_half(0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue(4, 0);
        issue(5, 1);
    }
    for (int Qb = 0; Qb < NQ; Qb += 4) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int cur = u;
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");       // half H + 1 has landed (H + 2 may still be in flight)
            read_half(cur ^ 1, 2 * (u ^ 1));
            __builtin_amdgcn_sched_barrier(0);
            Bf3 X[3];
#pragma unroll
            for (int q = 0; q < 3; q++) X[q] = bf3_split8(raw[cur][q]);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // ... and is in registers: its slots take half H + 3
            issue(Qb + 2 * u + 6, 2 * (u ^ 1));
            issue(Qb + 2 * u + 7, 2 * (u ^ 1) + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 5; mt++) {
                const Bf3 Z = bf3_split8(raw[cur][3 + mt]);
#pragma unroll
                for (int q = 0; q < 3; q++) gW1[mt][q] = mfma_bf3(Z, X[q], gW1[mt][q]);
                const float* z = raw[cur][3 + mt];
                b1acc[mt] += ((z[0] + z[1]) + (z[2] + z[3])) + ((z[4] + z[5]) + (z[6] + z[7]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* out = slab_rows + (size_t)gw * (m.n_params + 8);
#pragma unroll
    for (int mt = 0; mt < 5; mt++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int G = mt * 16 + r;
            if (G < 75) {
                const int n = G / 25, f = 2 * (G % 25) + h;
#pragma unroll
                for (int q = 0; q < 3; q++) out[n * m.net_size + m.w_off[0] + (q * 32 + j) * 50 + f] = gW1[mt][q][r];
            }
        }
        const float s1 = b1acc[mt] + swap32(b1acc[mt], h);
        const int r_j = (j & 3) + 4 * (j >> 3), h_j = (j >> 2) & 1;
        const int G = mt * 16 + r_j;
        if (h == 0 && G < 75) out[(G / 25) * m.net_size + m.b_off[0] + 2 * (G % 25) + h_j] = s1;
    }
}

// ------------------------------------------------------------------------------------------------
// forward solve on 16-column wave tiles (v_mfma_f32_16x16x4_f32), TWO wavefronts per SIMD
//
// Same idea as rt_forward_kernel with half the per-wave state: lane (j = lane & 15, g = lane >> 4) holds rows 4g..4g+3 of
// a 16-row x 16-column tile in a float4; the B operand of k-step r is element r (K quad = rows r, 4+r, 8+r, 12+r).  A
// 32-level variable is two tiles.  ~150 registers per wave, so eight waves (two per SIMD) share one LDS weight image and
// one wave's activations / physics / RK update hide under the other's MFMA chains.  The stage tape is written in the
// 32-column register-image format the adjoint and dW1 kernels read (two 16-column waves make one 32-column tile).
//
// Row placement: layer-1 outputs stacked into 10 tiles; quad Q = 4 t + r (rows 16t + 4g + r) carries features 4 (Q%13) + g
// of net Q / 13 (13 quads = 52 rows per net, 2 of them padding); layer-2 outputs: quad Q2 = 4u + r carries features
// 4 Q2 + g (Q2 < 5); layer-3 output row = face index.
// ------------------------------------------------------------------------------------------------
typedef float f32x4t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4t mfma16t(float a, float b, f32x4t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4t mfma16_bf(u32x4 a, u32x4 b, f32x4t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// c += A B over one 32-deep k-block from the exact three-way splits of both operands (see rt_dw1_split_kernel)
__device__ __forceinline__ f32x4t mfma16_bf3(const Bf3& a, const Bf3& b, f32x4t c) {
    c = mfma16_bf(a.m, b.m, c);
    c = mfma16_bf(a.l, b.h, c);
    c = mfma16_bf(a.h, b.l, c);
    c = mfma16_bf(a.m, b.h, c);
    c = mfma16_bf(a.h, b.m, c);
    c = mfma16_bf(a.h, b.h, c);
    return c;
}
__device__ __forceinline__ Bf3 rt16_ldA(const u32x4* simg, int G, int lane) {
    Bf3 a;
    a.h = simg[(G * 3 + 0) * 64 + lane];
    a.m = simg[(G * 3 + 1) * 64 + lane];
    a.l = simg[(G * 3 + 2) * 64 + lane];
    return a;
}

__device__ __forceinline__ float rot_dn16(float x, int lane) { return __shfl(x, (lane + 48) & 63); }   // from lane - 16
__device__ __forceinline__ float rot_up16(float x, int lane) { return __shfl(x, (lane + 16) & 63); }   // from lane + 16

struct V16 { f32x4t t[2]; };    // one 32-level variable of 16 columns: level = 16 tau + 4 g + r

__device__ __forceinline__ V16 shift_down16(const V16& T, int lane, float below) {
    const int g = lane >> 4;
    const float r0 = rot_dn16(T.t[0][3], lane), r1 = rot_dn16(T.t[1][3], lane);
    V16 o;
    o.t[0][0] = g == 0 ? below : r0;
    o.t[1][0] = g == 0 ? r0 : r1;
#pragma unroll
    for (int tau = 0; tau < 2; tau++) { o.t[tau][1] = T.t[tau][0]; o.t[tau][2] = T.t[tau][1]; o.t[tau][3] = T.t[tau][2]; }
    return o;
}

__device__ __forceinline__ V16 shift_up16(const V16& T, int lane, float above) {
    const int g = lane >> 4;
    const float r0 = rot_up16(T.t[0][0], lane), r1 = rot_up16(T.t[1][0], lane);
    V16 o;
    o.t[0][3] = g == 3 ? r1 : r0;
    o.t[1][3] = g == 3 ? above : r1;
#pragma unroll
    for (int tau = 0; tau < 2; tau++) { o.t[tau][0] = T.t[tau][1]; o.t[tau][1] = T.t[tau][2]; o.t[tau][2] = T.t[tau][3]; }
    return o;
}

// acc += sum_{s<N} A_s B_s on 16x16x4 MFMAs, A operands prefetched one chunk ahead (see rt_chain)
template <int N, int CH, class AF, class BF>
__device__ __forceinline__ f32x4t rt16_chain(const float* wl, f32x4t acc, AF aidx, BF bval) {
    float a[2][CH];
#pragma unroll
    for (int u = 0; u < CH; u++)
        if (u < N) a[0][u] = wl[aidx(u)];
    RT_SCHED_FENCE();
    __builtin_amdgcn_s_setprio(1);      // two waves share a SIMD in rt16_forward_kernel: the one inside a chain issues first (35.0 -> 34.3 ms)
#pragma unroll
    for (int c = 0; c * CH < N; c++) {
#pragma unroll
        for (int u = 0; u < CH; u++)
            if ((c + 1) * CH + u < N) a[(c + 1) & 1][u] = wl[aidx((c + 1) * CH + u)];
#pragma unroll
        for (int u = 0; u < CH; u++)
            if (c * CH + u < N) acc = mfma16t(a[c & 1][u], bval(c * CH + u), acc);
        RT_SCHED_FENCE();
    }
    __builtin_amdgcn_s_setprio(0);
    return acc;
}

// rt16_chain with independent vector work (`fill`: the previous tile's tape store and activation) issued BETWEEN the MFMAs.  One wave issues in
// order, and left to the instruction selector every activation of a layer sinks behind all of its chains (side-effect-free arithmetic is not
// ordered against the scheduling fences when the block is linearised); here the filler's inputs and results are pinned with empty asm
// statements and the region's issue order is prescribed: per MFMA one A-operand read (while any remain) and `NV` vector instructions.
template <int N, int CH, int NV, class AF, class BF, class FF>
__device__ __forceinline__ f32x4t rt16_chain_fill(const float* wl, f32x4t acc, AF aidx, BF bval, FF fill) {
    float a[2][CH];
#pragma unroll
    for (int u = 0; u < CH; u++)
        if (u < N) a[0][u] = wl[aidx(u)];
    RT_SCHED_HARD();
    fill();
#pragma unroll
    for (int c = 0; c * CH < N; c++) {
#pragma unroll
        for (int u = 0; u < CH; u++)
            if ((c + 1) * CH + u < N) a[(c + 1) & 1][u] = wl[aidx((c + 1) * CH + u)];
#pragma unroll
        for (int u = 0; u < CH; u++)
            if (c * CH + u < N) acc = mfma16t(a[c & 1][u], bval(c * CH + u), acc);
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // one LDS read (the next chunk's operands)
        __builtin_amdgcn_sched_group_barrier(0x402, NV, 0);      // NV vector / transcendental instructions of the filler
    }
    RT_SCHED_HARD();
    return acc;
}

// Z1 tape store of layer-1 tile tt of the 16-column forward kernels (the adjoint's per-net register-image format: feature f = 4 qq + g of net n is element
// 2 (qq & 1) + (g >> 1) of group qq >> 1 at lane32 = j + 16 half + 32 (g & 1)).  Round 5 A/B (-DRT_Z1_DWORD=0): 16-byte stores instead of dword stores.  The lanes g and g ^ 2
// (= lane ^ 32) hold the two halves of every group; two v_permlane32_swap per FOUR consecutive qq hand lanes g < 2 the complete group qq0 >> 1 and lanes
// g >= 2 the complete next group, so a net's twelve quads leave with three 16-byte stores per lane (all 64 lanes storing) instead of twelve dword stores;
// quad 12 (features 48, 49: lanes g < 2) keeps its dword store.  A chunk of four quads may straddle two tiles: the pre-activations wait in zpre (compile-time indices).
// MEASURED SLOWER: forward 28.0 -> 31.7 ms (bit-identical tapes; 39 dword stores become 9 16-byte stores + 3 dword stores + 18 swaps per lane and stage, but the swaps and the
// waiting pre-activations cost the 128-register kernel more than the store instructions did) — the dword stores stay.
#ifndef RT_Z1_DWORD
#define RT_Z1_DWORD 1         // 1: dword stores (shipped); 0: the 16-byte variant (A/B build)
#endif
__device__ __forceinline__ void rt16_tape_z1(float* oz /* tz + record offset (lane part and (g >> 1) folded in) */, float (&zpre)[40], const f32x4t z, int tt, int g) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int Q = 4 * tt + r, qq = Q % 13, net = Q / 13;
        if (Q >= 39) continue;
        if (RT_Z1_DWORD) {
            if (qq < 12 || g < 2) oz[(net * 7 + (qq >> 1)) * 256 + ((2 * qq) & 3)] = z[r];
            continue;
        }
        zpre[Q] = z[r];
        if (qq == 12) {
            if (g < 2) oz[(net * 7 + 6) * 256] = z[r];
        } else if ((qq & 3) == 3) {
            const int Q0 = Q - 3;
            const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(zpre[Q0]), __float_as_uint(zpre[Q0 + 2]), false, false);
            const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(zpre[Q0 + 1]), __float_as_uint(zpre[Q0 + 3]), false, false);
            const f32x4v v = {__uint_as_float(s02[0]), __uint_as_float(s02[1]), __uint_as_float(s13[0]), __uint_as_float(s13[1])};
            *reinterpret_cast<f32x4v*>(oz - (g >> 1) + (net * 7 + ((qq - 3) >> 1) + (g >> 1)) * 256) = v;
        }
    }
}

template <int ACT, bool SPLIT = false>
__global__ void __launch_bounds__(512)
rt16_forward_kernel(DevModel m, const float* __restrict__ wimg, const float* __restrict__ x0, const float* __restrict__ bcs,
                    const float* __restrict__ save_times, int n_save, int substeps, float* __restrict__ sol,
                    float* __restrict__ tape, float* __restrict__ tapez, int n_col) {
    float* wl = rt_smem;
    const float* bl = wl;                                 // biases: bl[RT_B1C ...]
    const u32x4* simg = reinterpret_cast<const u32x4*>(rt_smem);
    if constexpr (SPLIT) {
        // LDS: the bf16 operand image (RT_SIMG_WORDS words, behind the fp32 image in the same allocation) and the fp32 bias tail
        const unsigned* src = reinterpret_cast<const unsigned*>(wimg + RT_SIMG_OFF);
        unsigned* dst = reinterpret_cast<unsigned*>(rt_smem);
        for (int e = threadIdx.x; e < RT_SIMG_WORDS / 4; e += blockDim.x)
            reinterpret_cast<u32x4*>(dst)[e] = reinterpret_cast<const u32x4*>(src)[e];
        for (int e = threadIdx.x; e < RT_IMG_FLOATS - RT_B1C; e += blockDim.x) wl[RT_SIMG_WORDS + e] = wimg[RT_B1C + e];
        bl = wl + RT_SIMG_WORDS - RT_B1C;
    } else {
        for (int e = threadIdx.x; e < RT_IMG_FLOATS; e += blockDim.x) wl[e] = wimg[e];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int wt = blockIdx.x * (blockDim.x >> 6) + wave; // 16-column tile (1..RT16_WAVES wavefronts per workgroup)
    const int tile32 = wt >> 1, half = wt & 1;
    if (tile32 * 32 >= n_col) return;                     // (both halves of a live 32-column tile run: the tape must be whole)
    const int col = wt * 16 + j;
    const bool valid = col < n_col;
    const int colc = min(col, n_col - 1);
    // A-operand bases (this lane as output row i = 4 g_i + r_i of a 16-row tile; k index = g)
    const int i_ = lane & 15, g_i = i_ >> 2, r_i = i_ & 3;
    int a1b[10];
#pragma unroll
    for (int t = 0; t < 10; t++) {
        const int Q = 4 * t + r_i, f = 4 * (Q % 13) + g_i;
        const int row = (Q < 39 && f < 50) ? (Q / 13) * 50 + f : 0;       // padding rows read a valid row; never consumed
        a1b[t] = RT_W1C + row * RT_LD1 + 4 * g;
    }
    int a2b[2], a2l[2], a3b[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int Q2 = 4 * u + r_i;
        const int row2 = Q2 < 5 ? 4 * Q2 + g_i : 0;
        a2b[u] = RT_W2C + row2 * RT_LD2 + g;
        a2l[u] = RT_W2C + row2 * RT_LD2 + (g < 2 ? 48 + g : 50);           // last quad of a net: features 48, 49, then the zero column
        a3b[u] = RT_W3C + (16 * u + i_ - 1) * RT_LD3 + g;
    }
    RtBC bc;
    float bc5;
    {
        const float* bp = bcs + (size_t)colc * 6;
        bc.b[0] = bp[0]; bc.t[0] = bp[1]; bc.b[1] = bp[2]; bc.t[1] = bp[3]; bc.b[2] = bp[4]; bc5 = bp[5];
        bc.t[2] = bc5;
    }
    V16 Xn[3];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int tau = 0; tau < 2; tau++) {
            const f32x4v v = *reinterpret_cast<const f32x4v*>(x0 + (size_t)colc * 96 + q * 32 + 16 * tau + 4 * g);
            Xn[q].t[tau] = v;
            if (sol && valid) *reinterpret_cast<f32x4v*>(sol + ((size_t)col * n_save) * 96 + q * 32 + 16 * tau + 4 * g) = v;
        }
    const int n_steps = (n_save - 1) * substeps;
    RT_STAMP_DECL;
#ifdef COLNDE_STAMPS_FWD
    const unsigned long long rs_k0 = __builtin_amdgcn_s_memtime(), rs_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // 32-column register image: group (q*4 + 2 tau + (g>>1)), lane32 = j + 16 half + 32 (g & 1)
    float* tp = tape ? tape + (size_t)tile32 * n_steps * 4 * 3072 + ((g >> 1) * 64 + j + 16 * half + 32 * (g & 1)) * 4 : nullptr;
    // layer-1 pre-activations, taped in the adjoint kernel's per-net register-image format (same as the delta tape): feature
    // f = 4 qq + g of net n is element (2qq & 3) + (g >> 1) of group qq >> 1, lane32 = j + 16 half + 32 (g & 1)
    float* tz = tapez ? tapez + (size_t)tile32 * n_steps * 4 * RT_TAPEZ + (j + 16 * half + 32 * (g & 1)) * 4 + (g >> 1) : nullptr;
    const float Nz = 32.0f;
    struct { float cU, sU, cV, sV, cB, sB, kE, oE, cE, nA, nB, f0, f1, f2; } pc;
    {
        const float L2E = 1.4426950408889634f;
        pc.cU = m.sig_u * Nz; pc.sU = m.sig_u * m.eps; pc.cV = m.sig_v * Nz; pc.sV = m.sig_v * m.eps; pc.cB = m.B * Nz; pc.sB = m.B * m.eps;
        pc.kE = 2.0f * m.inv_dRi * L2E; pc.oE = -2.0f * m.Ric * m.inv_dRi * L2E; pc.cE = 30.0f * L2E;
        pc.nA = -0.5f * m.nu_minus; pc.nB = m.nu0 + 0.5f * m.nu_minus;
        pc.f0 = -m.cs[0] * Nz; pc.f1 = -m.cs[1] * Nz; pc.f2 = -m.cs[2] * m.inv_Pr * Nz;
    }
    int step = 0;
    for (int iv = 0; iv < n_save - 1; iv++) {
        const float t0 = save_times[iv];
        const float dt = (save_times[iv + 1] - t0) / (float)substeps;
        for (int s = 0; s < substeps; s++, step++) {
            const float ts = t0 + (float)s * dt;
            V16 Xs[3], Kacc[3];
#pragma unroll
            for (int q = 0; q < 3; q++) { Xs[q] = Xn[q]; Kacc[q].t[0] = (f32x4t)(0.0f); Kacc[q].t[1] = (f32x4t)(0.0f); }
#pragma nounroll
            for (int st = 0; st < 4; st++) {
                const float ca = st == 0 ? 0.0f : (st == 3 ? 1.0f : 0.5f);
                const float cb = (st == 0 || st == 3) ? 1.0f / 6.0f : 1.0f / 3.0f;
                RT_STAMP_BEGIN();
                if (tp) {
                    float* o = tp + ((size_t)step * 4 + st) * 3072;
#pragma unroll
                    for (int q = 0; q < 3; q++)
#pragma unroll
                        for (int tau = 0; tau < 2; tau++) *reinterpret_cast<f32x4v*>(o + (q * 4 + 2 * tau) * 256) = Xs[q].t[tau];
                }
                bc.t[2] = rt_top_flux(m, bc5, ts + ca * dt);
                RT_STAMP(0);
                // ---- three MLPs -------------------------------------------------------------------------------------
                f32x4t A1[10];
                float zpre[40];             // layer-1 pre-activations waiting for their 16-byte tape store (rt16_tape_z1): at most one tile's worth is live
                // operand addresses stay (lane base + immediate): left loop-invariant, all ~150 of them are computed outside the time loop and spilled
                int lz = lane;
                if constexpr (SPLIT) asm volatile("" : "+v"(lz));
                if constexpr (SPLIT) {
                    // k-block (= variable) outer, output tile inner: one variable's B planes live at a time, ten independent accumulation chains
#pragma unroll
                    for (int t = 0; t < 10; t++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int Q = 4 * t + r;
                            A1[t][r] = Q < 39 ? bl[RT_B1C + (Q / 13) * 50 + min(4 * (Q % 13) + g, 49)] : 0.0f;
                        }
                    Bf3 Apf = rt16_ldA(simg, 0, lz);
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        // B operand of the 32-deep k-block: the variable's eight levels of this lane, split exactly into three bf16 planes
                        const float x8[8] = {Xs[q].t[0][0], Xs[q].t[0][1], Xs[q].t[0][2], Xs[q].t[0][3], Xs[q].t[1][0], Xs[q].t[1][1], Xs[q].t[1][2], Xs[q].t[1][3]};
                        const Bf3 XB = bf3_split8(x8);
#pragma unroll
                        for (int t = 0; t < 10; t++) {
                            const Bf3 Ac = Apf;
                            const int nx = q * 10 + t + 1;                                        // the group after (q, t) in this order
                            if (nx < 30) Apf = rt16_ldA(simg, 3 * (nx % 10) + nx / 10, lz);
                            auto finish = [&](int tt) {        // tile tt is complete: tape store, activation
                                if (tz && !RT_ZRICH) rt16_tape_z1(tz + ((size_t)step * 4 + st) * RT_TAPEZ, zpre, A1[tt], tt, g);
                                // (side-effect-free arithmetic is not ordered against the scheduling fences when the block is linearised: the empty
                                //  asm statements pin the activation's inputs below the products' issue point and its results above the closing fence)
#pragma unroll
                                for (int r = 0; r < 4; r++) asm volatile("" : "+v"(A1[tt][r]));
                                if (RT_ZRICH && tz) {
                                    // value and derivative together (one exponential, one reciprocal), both taped
                                    f32x2v a0, d0, a1, d1;
                                    rt_act_pair4<ACT>((f32x2v){A1[tt][0], A1[tt][1]}, (f32x2v){A1[tt][2], A1[tt][3]}, a0, d0, a1, d1);
                                    A1[tt] = (f32x4t){a0.x, a0.y, a1.x, a1.y};
                                    const float dd[4] = {d0.x, d0.y, d1.x, d1.y};
                                    float* oz = tz + ((size_t)step * 4 + st) * RT_TAPEZ;
#pragma unroll
                                    for (int r = 0; r < 4; r++) {
                                        const int Q = 4 * tt + r, qq = Q % 13;
                                        if (Q < 39 && (qq < 12 || g < 2)) {
                                            oz[((Q / 13) * 7 + (qq >> 1)) * 256 + ((2 * qq) & 3)] = A1[tt][r];
                                            oz[RT_TAPEZ_HALF + ((Q / 13) * 7 + (qq >> 1)) * 256 + ((2 * qq) & 3)] = dd[r];
                                        }
                                    }
                                } else
                                A1[tt] = rt_act4<ACT>(A1[tt]);
#pragma unroll
                                for (int r = 0; r < 4; r++) asm volatile("" : "+v"(A1[tt][r]));
                            };
                            if (q == 2) RT_SCHED_HARD();
                            __builtin_amdgcn_s_setprio(1);
                            A1[t] = mfma16_bf3(Ac, XB, A1[t]);
                            __builtin_amdgcn_s_setprio(0);
                            if (q == 2) {
                                // the last k-block finishes the tiles one by one: tile t - 1's tape store and activation are issued BETWEEN tile t's
                                // six products (one wave issues in order: left to the scheduler, all activations sink behind all MFMAs)
                                if (t > 0) {
                                    finish(t - 1);
#pragma unroll
                                    for (int u = 0; u < 6; u++) {
                                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                        __builtin_amdgcn_sched_group_barrier(0x402, 9, 0);
                                    }
                                }
                                RT_SCHED_HARD();
                                if (t == 9) finish(9);
                            } else
                                RT_SCHED_FENCE();
                        }
                    }
                } else {
                auto finish1 = [&](int tt) {          // tile tt of layer 1 is complete: Z1 tape store, activation (pinned: see rt16_chain_fill)
                    if (tz && !RT_ZRICH) rt16_tape_z1(tz + ((size_t)step * 4 + st) * RT_TAPEZ, zpre, A1[tt], tt, g);
#pragma unroll
                    for (int r = 0; r < 4; r++) asm volatile("" : "+v"(A1[tt][r]));
                    if (RT_ZRICH && tz) {
                        f32x2v a0, d0, a1, d1;
                        rt_act_pair4<ACT>((f32x2v){A1[tt][0], A1[tt][1]}, (f32x2v){A1[tt][2], A1[tt][3]}, a0, d0, a1, d1);
                        A1[tt] = (f32x4t){a0.x, a0.y, a1.x, a1.y};
                        const float dd[4] = {d0.x, d0.y, d1.x, d1.y};
                        float* oz = tz + ((size_t)step * 4 + st) * RT_TAPEZ;
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int Q = 4 * tt + r, qq = Q % 13;
                            if (Q < 39 && (qq < 12 || g < 2)) {
                                oz[((Q / 13) * 7 + (qq >> 1)) * 256 + ((2 * qq) & 3)] = A1[tt][r];
                                oz[RT_TAPEZ_HALF + ((Q / 13) * 7 + (qq >> 1)) * 256 + ((2 * qq) & 3)] = dd[r];
                            }
                        }
                    } else
                    A1[tt] = rt_act4<ACT>(A1[tt]);
#pragma unroll
                    for (int r = 0; r < 4; r++) asm volatile("" : "+v"(A1[tt][r]));
                };
#pragma unroll
                for (int t = 0; t < 10; t++) {
                    f32x4t acc;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int Q = 4 * t + r;
                        acc[r] = Q < 39 ? wl[RT_B1C + (Q / 13) * 50 + min(4 * (Q % 13) + g, 49)] : 0.0f;
                    }
                    const int base = a1b[t];
#ifndef RT16_FWD_FILL     // (measured: 35.4 ms with the filler against 34.0-34.9 without — two waves per SIMD already overlap what can be overlapped; kept as an A/B switch)
                    acc = rt16_chain<24, 8>(wl, acc, [=](int k) { return base + (k >> 3) * 32 + ((k >> 2) & 1) * 16 + (k & 3); },
                                            [&](int k) { return Xs[k >> 3].t[(k >> 2) & 1][k & 3]; });
                    A1[t] = acc;
                    finish1(t);
#else
                    // tile t - 1's tape store and activation are issued between tile t's MFMAs
                    acc = rt16_chain_fill<24, 8, 2>(wl, acc, [=](int k) { return base + (k >> 3) * 32 + ((k >> 2) & 1) * 16 + (k & 3); },
                                                    [&](int k) { return Xs[k >> 3].t[(k >> 2) & 1][k & 3]; }, [&] { if (t > 0) finish1(t - 1); });
                    A1[t] = acc;
                    if (t == 9) finish1(9);
#endif
                }
                }
                RT_STAMP(1);
                V16 O[3];
#pragma unroll
                for (int n = 0; n < 3; n++) {
                    f32x4t A2[2];
                    Bf3 HB[2];
                    if constexpr (SPLIT) {
                        // net n's 13 quads of layer-1 activations as two 32-deep k-blocks (element e of block c: quad 8 c + e; three zero slots)
#pragma unroll
                        for (int c = 0; c < 2; c++) {
                            float a8[8];
#pragma unroll
                            for (int e = 0; e < 8; e++) a8[e] = 8 * c + e < 13 ? A1[(13 * n + 8 * c + e) >> 2][(13 * n + 8 * c + e) & 3] : 0.0f;
                            HB[c] = bf3_split8(a8);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        f32x4t acc;
#pragma unroll
                        for (int r = 0; r < 4; r++) acc[r] = (4 * u + r < 5) ? bl[RT_B2C + n * 20 + 4 * (4 * u + r) + g] : 0.0f;
                        if constexpr (SPLIT) {
                            const Bf3 Aa = rt16_ldA(simg, 30 + 4 * n + 2 * u, lz), Ab = rt16_ldA(simg, 30 + 4 * n + 2 * u + 1, lz);
                            __builtin_amdgcn_s_setprio(1);
                            acc = mfma16_bf3(Aa, HB[0], acc);
                            acc = mfma16_bf3(Ab, HB[1], acc);
                            __builtin_amdgcn_s_setprio(0);
                            RT_SCHED_FENCE();
                        } else {
                        const int base = a2b[u] + n * 20 * RT_LD2, basel = a2l[u] + n * 20 * RT_LD2;
                        acc = rt16_chain<13, 13>(wl, acc, [=](int k) { return k < 12 ? base + 4 * k : basel; },
                                                 [&](int k) { return A1[(13 * n + k) >> 2][(13 * n + k) & 3]; });
                        }
                        if (RT_Z2TAPE && tz) {
                            // layer-2 pre-activations in the adjoint's register-image format: feature f = 4 Q2 + g of net n is element (2 Q2 & 3) + (g >> 1) of group 3 n + (Q2 >> 1)
                            float* oz = tz + ((size_t)step * 4 + st) * RT_TAPEZ + RT_Z2OFF;
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const int Q2 = 4 * u + r;
                                if (Q2 < 5) oz[(3 * n + (Q2 >> 1)) * 256 + ((2 * Q2) & 3)] = acc[r];
                            }
                        }
                        A2[u] = rt_act4<ACT>(acc);
                    }
                    Bf3 GB;
                    if constexpr (SPLIT) {
                        const float a8[8] = {A2[0][0], A2[0][1], A2[0][2], A2[0][3], A2[1][0], 0.0f, 0.0f, 0.0f};
                        GB = bf3_split8(a8);
                    }
#pragma unroll
                    for (int v = 0; v < 2; v++) {
                        f32x4t acc;
#pragma unroll
                        for (int r = 0; r < 4; r++) acc[r] = bl[RT_B3C + n * 32 + 16 * v + 4 * g + r];
                        if constexpr (SPLIT) {
                            const Bf3 Aa = rt16_ldA(simg, 42 + 2 * n + v, lz);
                            __builtin_amdgcn_s_setprio(1);
                            O[n].t[v] = mfma16_bf3(Aa, GB, acc);
                            __builtin_amdgcn_s_setprio(0);
                            RT_SCHED_FENCE();
                        } else {
                        const int base = a3b[v] + n * 31 * RT_LD3;
                        O[n].t[v] = rt16_chain<5, 5>(wl, acc, [=](int k) { return base + 4 * k; },
                                                     [&](int k) { return A2[k >> 2][k & 3]; });
                        }
                    }
                }
                RT_STAMP(2);
                // ---- physics (predict_flux / predict_NDE), face index = level index ----------------------------------
                V16 F[3];
                {
                    V16 Ud, Vd, Td;
                    if (m.mpp || m.ca) {
                        Ud = shift_down16(Xs[0], lane, 0.0f); Vd = shift_down16(Xs[1], lane, 0.0f); Td = shift_down16(Xs[2], lane, 0.0f);
                    }
                    if (m.mpp) {
                        // the Richardson-number closure on pairs of adjacent levels (packed f32 arithmetic: two levels per issue slot);
                        // uniform factors folded (see rt_physics_vjp): d = level difference, a = σ (Nz d + ε), Ri = B (Nz d_T + ε) / S2,
                        // tanh via exp2, ν = nB + nA tanh, diffusive flux = -(c Nz) ν d
#pragma unroll
                        for (int tau = 0; tau < 2; tau++)
#pragma unroll
                            for (int r = 0; r < 4; r += 2) {
                                const f32x2v dU = {Xs[0].t[tau][r] - Ud.t[tau][r], Xs[0].t[tau][r + 1] - Ud.t[tau][r + 1]};
                                const f32x2v dV = {Xs[1].t[tau][r] - Vd.t[tau][r], Xs[1].t[tau][r + 1] - Vd.t[tau][r + 1]};
                                const f32x2v dT = {Xs[2].t[tau][r] - Td.t[tau][r], Xs[2].t[tau][r + 1] - Td.t[tau][r + 1]};
                                const f32x2v a1 = dU * pc.cU + pc.sU, a2 = dV * pc.cV + pc.sV;
                                const f32x2v s2 = a2 * a2 + a1 * a1;
                                f32x2v rS;
                                rS.x = __builtin_amdgcn_rcpf(s2.x);
                                rS.y = __builtin_amdgcn_rcpf(s2.y);
                                const f32x2v arg = ((dT * pc.cB + pc.sB) * rS) * pc.kE + pc.oE;
                                f32x2v e;
                                e.x = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(arg.x, -pc.cE, pc.cE));
                                e.y = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(arg.y, -pc.cE, pc.cE));
                                const f32x2v e1 = e + 1.0f;
                                f32x2v rc;
                                rc.x = __builtin_amdgcn_rcpf(e1.x);
                                rc.y = __builtin_amdgcn_rcpf(e1.y);
                                const f32x2v nu = (rc * -2.0f + 1.0f) * pc.nA + pc.nB;
                                const f32x2v o0 = {O[0].t[tau][r], O[0].t[tau][r + 1]}, o1 = {O[1].t[tau][r], O[1].t[tau][r + 1]},
                                             o2 = {O[2].t[tau][r], O[2].t[tau][r + 1]};
                                const f32x2v f0 = (nu * dU) * pc.f0 + o0, f1 = (nu * dV) * pc.f1 + o1, f2 = (nu * dT) * pc.f2 + o2;
                                F[0].t[tau][r] = f0.x; F[0].t[tau][r + 1] = f0.y;
                                F[1].t[tau][r] = f1.x; F[1].t[tau][r + 1] = f1.y;
                                F[2].t[tau][r] = f2.x; F[2].t[tau][r + 1] = f2.y;
                            }
                        if (g == 0) {                                              // face 0: the bottom boundary
#pragma unroll
                            for (int k = 0; k < 3; k++) F[k].t[0][0] = m.zero_w ? bc.b[k] - m.s0[k] : bc.b[k];
                        }
                    } else
#pragma unroll
                    for (int tau = 0; tau < 2; tau++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const bool in = !(tau == 0 && r == 0 && g == 0);       // face >= 1
                            float f0 = in ? O[0].t[tau][r] : 0.0f, f1 = in ? O[1].t[tau][r] : 0.0f, f2 = in ? O[2].t[tau][r] : 0.0f;
                            if (!m.zero_w && !in) { f0 = bc.b[0]; f1 = bc.b[1]; f2 = bc.b[2]; }
                            if (m.ca && in) {
                                const float gT = (Xs[2].t[tau][r] - Td.t[tau][r]) * Nz;
                                f2 -= m.cs[2] * m.kappa * fminf(0.0f, gT);
                            }
                            F[0].t[tau][r] = f0; F[1].t[tau][r] = f1; F[2].t[tau][r] = f2;
                        }
                }
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float top = m.zero_w ? bc.t[k] - m.s0[k] : bc.t[k];
                    const V16 Fu = shift_up16(F[k], lane, top);
#pragma unroll
                    for (int tau = 0; tau < 2; tau++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            float v = -m.A[k] * (Fu.t[tau][r] - F[k].t[tau][r]);
                            if (k == 0) v += m.cor_u * (m.sig_v * Xs[1].t[tau][r] + m.mu_v);
                            if (k == 1) v -= m.cor_v * (m.sig_u * Xs[0].t[tau][r] + m.mu_u);
                            F[k].t[tau][r] = v;                           // F now holds the tendency K
                        }
                }
                RT_STAMP(3);
                // ---- RK4 bookkeeping: accumulate, form the next stage input ------------------------------------------
                const float can = st == 2 ? 1.0f : 0.5f;                    // abscissa of the NEXT stage
#pragma unroll
                for (int q = 0; q < 3; q++)
#pragma unroll
                    for (int tau = 0; tau < 2; tau++) {
                        Kacc[q].t[tau] += cb * F[q].t[tau];
                        Xs[q].t[tau] = Xn[q].t[tau] + (can * dt) * F[q].t[tau];
                    }
                RT_STAMP(4);
            }
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int tau = 0; tau < 2; tau++) Xn[q].t[tau] += dt * Kacc[q].t[tau];
            if (s == substeps - 1 && sol && valid) {
#pragma unroll
                for (int q = 0; q < 3; q++)
#pragma unroll
                    for (int tau = 0; tau < 2; tau++)
                        *reinterpret_cast<f32x4v*>(sol + ((size_t)col * n_save + iv + 1) * 96 + q * 32 + 16 * tau + 4 * g) = Xn[q].t[tau];
            }
        }
    }
#ifdef COLNDE_STAMPS_FWD
    rs_acc[6] = __builtin_amdgcn_s_memtime() - rs_k0;          // whole kernel, shader ticks
    rs_acc[7] = __builtin_amdgcn_s_memrealtime() - rs_r0;      // whole kernel, 100 MHz reference ticks
    RT_STAMP_FLUSH();
#endif
}

// ------------------------------------------------------------------------------------------------
// forward solve of ONE 16-column tile by THREE wavefronts — the latency points (8 .. a few thousand simulations: fewer tiles than CUs,
// the shape the reference actually trains, NDE_training.jl:291).  The three flux nets are independent given the state, and so are the
// three variables' tendencies given the fluxes: wave n runs net n (4 + 2 + 2 output tiles: 132 MFMAs instead of 348 in a row) and advances
// variable n; the new stage input of each variable goes through a double-buffered LDS exchange (one barrier per stage).  Same
// arithmetic, layouts and chains as rt16_forward_kernel.  The tapes are written in tile16's formats, because at these sizes the gradient
// is taken by tile16's taped adjoint: stage inputs [tile][step][stage][column][3 Nz], hidden pre-activations [tile][step][stage][column][net][72].
// ------------------------------------------------------------------------------------------------
// value and derivative of the activation on a four-element tile
template <int ACT>
__device__ __forceinline__ void rt16_act_pair(const f32x4t z, f32x4t& a, f32x4t& d) {
    const f32x2v z0 = {z[0], z[1]}, z1 = {z[2], z[3]};
    f32x2v a0, d0, a1, d1;
    rt_act_pair4<ACT>(z0, z1, a0, d0, a1, d1);
    a = (f32x4t){a0.x, a0.y, a1.x, a1.y};
    d = (f32x4t){d0.x, d0.y, d1.x, d1.y};
}

// RICH (small problems, where tape bytes cost nothing): instead of the hidden pre-activations in tile16's format, the forward kernel tapes what the
// net-split adjoint would otherwise recompute on its critical path, as register images (1-KB coalesced stores):
//   [tile][step][stage][ net 0..2: a1 (4 tiles) | act'(z1) (4) | a2 (2) | act'(z2) (2) ][ physics pullback coefficients dn_k, nu_k, c_k (k = 0..2) x 2 tiles ][64 lanes][4]
// — the activation derivatives with the padding slots already zeroed, and the nine per-level coefficients of rt16_physics_apply (everything in the
// Richardson-number closure that depends on the stage input alone: all the transcendental work of the pullback).
#define RT16S_RREC ((3 * 12 + 18) * 256)    // floats per rich tape record
template <int ACT, bool RICH>
__global__ void __launch_bounds__(192)
rt16s_forward_kernel(DevModel m, const float* __restrict__ wimg, const float* __restrict__ x0, const float* __restrict__ bcs,
                     const float* __restrict__ save_times, int n_save, int substeps, float* __restrict__ sol,
                     float* __restrict__ t16_tape, float* __restrict__ t16_ztape, int n_col) {
    float* wl = rt_smem;
    for (int e = threadIdx.x; e < RT_IMG_FLOATS; e += 192) wl[e] = wimg[e];
    f32x4v* ex = reinterpret_cast<f32x4v*>(rt_smem + ((RT_IMG_FLOATS + 3) & ~3));          // [2 buffers][3 variables][2 tiles][64 lanes]
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int n = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                        // this wave's net = its variable
    const int j = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x;
    const int col = tile * 16 + j;
    const bool valid = col < n_col;
    const int colc = min(col, n_col - 1);
    const int i_ = lane & 15, g_i = i_ >> 2, r_i = i_ & 3;
    int a1b[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int Q = 4 * t + r_i, f = 4 * Q + g_i;
        a1b[t] = RT_W1C + (n * 50 + ((Q < 13 && f < 50) ? f : 0)) * RT_LD1 + 4 * g;       // padding rows read a valid row; never consumed
    }
    int a2b[2], a2l[2], a3b[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int Q2 = 4 * u + r_i;
        const int row2 = n * 20 + (Q2 < 5 ? 4 * Q2 + g_i : 0);
        a2b[u] = RT_W2C + row2 * RT_LD2 + g;
        a2l[u] = RT_W2C + row2 * RT_LD2 + (g < 2 ? 48 + g : 50);
        a3b[u] = RT_W3C + (n * 31 + 16 * u + i_ - 1) * RT_LD3 + g;
    }
    float bcb, bct, bc5;
    {
        const float* bp = bcs + (size_t)colc * 6;
        bcb = bp[2 * n];
        bct = bp[2 * n + 1];
        bc5 = bp[5];
    }
    V16 Xs[3], Xn, Kacc;
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int tau = 0; tau < 2; tau++) Xs[q].t[tau] = *reinterpret_cast<const f32x4v*>(x0 + (size_t)colc * 96 + q * 32 + 16 * tau + 4 * g);
    // (wave-uniform n: a select, not a dynamic register index)
#pragma unroll
    for (int tau = 0; tau < 2; tau++)
#pragma unroll
        for (int r = 0; r < 4; r++) Xn.t[tau][r] = n == 0 ? Xs[0].t[tau][r] : (n == 1 ? Xs[1].t[tau][r] : Xs[2].t[tau][r]);
    if (sol && valid)
#pragma unroll
        for (int tau = 0; tau < 2; tau++) *reinterpret_cast<f32x4v*>(sol + ((size_t)col * n_save) * 96 + n * 32 + 16 * tau + 4 * g) = Xn.t[tau];
    const int n_steps = (n_save - 1) * substeps;
    float* tp = t16_tape ? t16_tape + (size_t)tile * n_steps * 4 * 1536 + j * 96 + n * 32 + 4 * g : nullptr;
    float* tz = (t16_ztape && !RICH) ? t16_ztape + (size_t)tile * n_steps * 4 * (16 * 216) + j * 216 + n * 72 + g : nullptr;
    float* tr = (t16_ztape && RICH) ? t16_ztape + (size_t)tile * n_steps * 4 * RT16S_RREC + lane * 4 : nullptr;
    const float Nz = 32.0f;
    const float L2E = 1.4426950408889634f;
    const float cU = m.sig_u * Nz, sU = m.sig_u * m.eps, cV = m.sig_v * Nz, sV = m.sig_v * m.eps, cB = m.B * Nz, sB = m.B * m.eps;
    const float kE = 2.0f * m.inv_dRi * L2E, oE = -2.0f * m.Ric * m.inv_dRi * L2E, cE = 30.0f * L2E;
    const float nA = -0.5f * m.nu_minus, nB = m.nu0 + 0.5f * m.nu_minus;
    const float fn = n == 0 ? -m.cs[0] * Nz : (n == 1 ? -m.cs[1] * Nz : -m.cs[2] * m.inv_Pr * Nz);
    const float s0n = n == 0 ? m.s0[0] : (n == 1 ? m.s0[1] : m.s0[2]);
    const float An = n == 0 ? m.A[0] : (n == 1 ? m.A[1] : m.A[2]);
    const float rmn = n == 0 ? -m.cs[0] : (n == 1 ? -m.cs[1] : -m.cs[2] * m.inv_Pr);                 // m_n of rt16_physics_apply's coefficients
    int step = 0, buf = 0;
    RT_STAMP_DECL;
    for (int iv = 0; iv < n_save - 1; iv++) {
        const float t0 = save_times[iv];
        const float dt = (save_times[iv + 1] - t0) / (float)substeps;
        for (int s = 0; s < substeps; s++, step++) {
            const float ts = t0 + (float)s * dt;
            Kacc.t[0] = (f32x4t)(0.0f);
            Kacc.t[1] = (f32x4t)(0.0f);
#pragma nounroll
            for (int st = 0; st < 4; st++) {
                const float ca = st == 0 ? 0.0f : (st == 3 ? 1.0f : 0.5f);
                const float cb = (st == 0 || st == 3) ? 1.0f / 6.0f : 1.0f / 3.0f;
                RT_STAMP_BEGIN();
                V16 Xme;               // (element-wise selects on the wave-uniform n: a select between the aggregates becomes a scratch array)
#pragma unroll
                for (int tau = 0; tau < 2; tau++)
#pragma unroll
                    for (int r = 0; r < 4; r++) Xme.t[tau][r] = n == 0 ? Xs[0].t[tau][r] : (n == 1 ? Xs[1].t[tau][r] : Xs[2].t[tau][r]);
                if (tp) {
                    float* o = tp + ((size_t)step * 4 + st) * 1536;
#pragma unroll
                    for (int tau = 0; tau < 2; tau++) *reinterpret_cast<f32x4v*>(o + 16 * tau) = Xme.t[tau];
                }
                float* oz = tz ? tz + ((size_t)step * 4 + st) * (16 * 216) : nullptr;
                float* orr = tr ? tr + ((size_t)step * 4 + st) * RT16S_RREC : nullptr;
                const float top_raw = n == 2 ? rt_top_flux(m, bc5, ts + ca * dt) : bct;
                RT_STAMP(0);
                // ---- net n ----------------------------------------------------------------------------------------------
                f32x4t A1[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    f32x4t acc;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int Q = 4 * t + r;
                        acc[r] = Q < 13 ? wl[RT_B1C + n * 50 + min(4 * Q + g, 49)] : 0.0f;
                    }
                    const int base = a1b[t];
                    acc = rt16_chain<24, 8>(wl, acc, [=](int k) { return base + (k >> 3) * 32 + ((k >> 2) & 1) * 16 + (k & 3); },
                                            [&](int k) { return Xs[k >> 3].t[(k >> 2) & 1][k & 3]; });
                    if (oz) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int Q = 4 * t + r;
                            if (Q < 13 && (Q < 12 || g < 2)) oz[4 * Q] = acc[r];             // feature 4 Q + g of layer 1
                        }
                    }
                    if (RICH && orr) {
                        f32x4t dd;
                        rt16_act_pair<ACT>(acc, A1[t], dd);
                        if (t == 3) { dd[1] = 0.0f; dd[2] = 0.0f; dd[3] = 0.0f; if (g >= 2) dd[0] = 0.0f; }   // padding: quads >= 13, features 50, 51
                        *reinterpret_cast<f32x4v*>(orr + (n * 12 + t) * 256) = A1[t];
                        *reinterpret_cast<f32x4v*>(orr + (n * 12 + 4 + t) * 256) = dd;
                    } else
                        A1[t] = rt_act4<ACT>(acc);
                }
                RT_STAMP(1);
                f32x4t A2[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    f32x4t acc;
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[r] = (4 * u + r < 5) ? wl[RT_B2C + n * 20 + 4 * (4 * u + r) + g] : 0.0f;
                    const int base = a2b[u], basel = a2l[u];
                    acc = rt16_chain<13, 13>(wl, acc, [=](int k) { return k < 12 ? base + 4 * k : basel; },
                                             [&](int k) { return A1[k >> 2][k & 3]; });
                    if (oz) {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (4 * u + r < 5) oz[52 + 4 * (4 * u + r)] = acc[r];            // feature 4 Q2 + g of layer 2
                    }
                    if (RICH && orr) {
                        f32x4t dd;
                        rt16_act_pair<ACT>(acc, A2[u], dd);
                        if (u == 1) { dd[1] = 0.0f; dd[2] = 0.0f; dd[3] = 0.0f; }                             // padding: quads >= 5
                        *reinterpret_cast<f32x4v*>(orr + (n * 12 + 8 + u) * 256) = A2[u];
                        *reinterpret_cast<f32x4v*>(orr + (n * 12 + 10 + u) * 256) = dd;
                    } else
                        A2[u] = rt_act4<ACT>(acc);
                }
                V16 O;
#pragma unroll
                for (int v = 0; v < 2; v++) {
                    f32x4t acc;
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[r] = wl[RT_B3C + n * 32 + 16 * v + 4 * g + r];
                    const int base = a3b[v];
                    O.t[v] = rt16_chain<5, 5>(wl, acc, [=](int k) { return base + 4 * k; }, [&](int k) { return A2[k >> 2][k & 3]; });
                }
                RT_STAMP(2);
                // ---- physics: face flux and tendency of variable n (predict_flux / predict_NDE) -----------------------------
                V16 F, Pd, Pn, Pc;
                {
                    V16 Ud, Vd, Td;
                    if (m.mpp || m.ca) {
                        Ud = shift_down16(Xs[0], lane, 0.0f); Vd = shift_down16(Xs[1], lane, 0.0f); Td = shift_down16(Xs[2], lane, 0.0f);
                    }
                    if (m.mpp) {
#pragma unroll
                        for (int tau = 0; tau < 2; tau++)
#pragma unroll
                            for (int r = 0; r < 4; r += 2) {
                                const f32x2v dU = {Xs[0].t[tau][r] - Ud.t[tau][r], Xs[0].t[tau][r + 1] - Ud.t[tau][r + 1]};
                                const f32x2v dV = {Xs[1].t[tau][r] - Vd.t[tau][r], Xs[1].t[tau][r + 1] - Vd.t[tau][r + 1]};
                                const f32x2v dT = {Xs[2].t[tau][r] - Td.t[tau][r], Xs[2].t[tau][r + 1] - Td.t[tau][r + 1]};
                                const f32x2v a1 = dU * cU + sU, a2 = dV * cV + sV;
                                const f32x2v s2 = a2 * a2 + a1 * a1;
                                f32x2v rS;
                                rS.x = __builtin_amdgcn_rcpf(s2.x);
                                rS.y = __builtin_amdgcn_rcpf(s2.y);
                                const f32x2v arg = ((dT * cB + sB) * rS) * kE + oE;
                                f32x2v e;
                                e.x = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(arg.x, -cE, cE));
                                e.y = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(arg.y, -cE, cE));
                                const f32x2v e1 = e + 1.0f;
                                f32x2v rc;
                                rc.x = __builtin_amdgcn_rcpf(e1.x);
                                rc.y = __builtin_amdgcn_rcpf(e1.y);
                                const f32x2v th = rc * -2.0f + 1.0f;
                                const f32x2v nu = th * nA + nB;
                                const f32x2v dn = n == 0 ? dU : (n == 1 ? dV : dT);
                                const f32x2v on = {O.t[tau][r], O.t[tau][r + 1]};
                                const f32x2v f = (nu * dn) * fn + on;
                                F.t[tau][r] = f.x; F.t[tau][r + 1] = f.y;
                                if (RICH && orr) {
                                    // this wave's three of the nine pullback coefficients (rt16_physics_apply): dn_n, nu_n, c_n
                                    const f32x2v wf = (1.0f - th * th) * rS;                 // (1 - tanh²) / S2
                                    const f32x2v wR = wf * ((dT * cB + sB) * rS);            // ... times Ri
                                    const f32x2v cn = n == 0 ? wR * (a1 * (-2.0f * m.sig_u)) : (n == 1 ? wR * (a2 * (-2.0f * m.sig_v)) : wf * m.B);
                                    const f32x2v pd = dn * (rmn * (Nz * m.c_rib)), pn = nu * rmn;
                                    Pd.t[tau][r] = pd.x; Pd.t[tau][r + 1] = pd.y;
                                    Pn.t[tau][r] = pn.x; Pn.t[tau][r + 1] = pn.y;
                                    Pc.t[tau][r] = cn.x; Pc.t[tau][r + 1] = cn.y;
                                }
                            }
                        if (g == 0) F.t[0][0] = m.zero_w ? bcb - s0n : bcb;                 // face 0: the bottom boundary
                        if (RICH && g == 0) { Pn.t[0][0] = 0.0f; Pc.t[0][0] = 0.0f; }       // ... carries no diffusive flux
                    } else {
#pragma unroll
                        for (int tau = 0; tau < 2; tau++)
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const bool in = !(tau == 0 && r == 0 && g == 0);
                                float f = in ? O.t[tau][r] : (m.zero_w ? 0.0f : bcb);
                                if (RICH) { Pd.t[tau][r] = 0.0f; Pn.t[tau][r] = 0.0f; Pc.t[tau][r] = 0.0f; }
                                if (m.ca && in && n == 2) {
                                    const float gT = (Xs[2].t[tau][r] - Td.t[tau][r]) * Nz;
                                    f -= m.cs[2] * m.kappa * fminf(0.0f, gT);
                                    if (RICH && gT < 0.0f) Pn.t[tau][r] = -m.cs[2] * m.kappa;
                                }
                                F.t[tau][r] = f;
                            }
                    }
                }
                if (RICH && orr && (m.mpp || m.ca)) {
#pragma unroll
                    for (int tau = 0; tau < 2; tau++) {
                        *reinterpret_cast<f32x4v*>(orr + (36 + (0 * 3 + n) * 2 + tau) * 256) = Pd.t[tau];
                        *reinterpret_cast<f32x4v*>(orr + (36 + (1 * 3 + n) * 2 + tau) * 256) = Pn.t[tau];
                        *reinterpret_cast<f32x4v*>(orr + (36 + (2 * 3 + n) * 2 + tau) * 256) = Pc.t[tau];
                    }
                }
                {
                    const float top = m.zero_w ? top_raw - s0n : top_raw;
                    const V16 Fu = shift_up16(F, lane, top);
#pragma unroll
                    for (int tau = 0; tau < 2; tau++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            float v = -An * (Fu.t[tau][r] - F.t[tau][r]);
                            if (n == 0) v += m.cor_u * (m.sig_v * Xs[1].t[tau][r] + m.mu_v);
                            if (n == 1) v -= m.cor_v * (m.sig_u * Xs[0].t[tau][r] + m.mu_u);
                            F.t[tau][r] = v;                                                // F now holds the tendency of variable n
                        }
                }
                RT_STAMP(3);
                // ---- RK4 bookkeeping for variable n; the next stage input (or, after stage 3, the new state) is exchanged -------------
                V16 Xnext;
#pragma unroll
                for (int tau = 0; tau < 2; tau++) {
                    Kacc.t[tau] += cb * F.t[tau];
                    if (st < 3) {
                        const float can = st == 2 ? 1.0f : 0.5f;
                        Xnext.t[tau] = Xn.t[tau] + (can * dt) * F.t[tau];
                    } else {
                        Xn.t[tau] += dt * Kacc.t[tau];
                        Xnext.t[tau] = Xn.t[tau];
                    }
                }
                f32x4v* eb = ex + buf * 384;
#pragma unroll
                for (int tau = 0; tau < 2; tau++) eb[(n * 2 + tau) * 64 + lane] = Xnext.t[tau];
                // (a bare barrier behind the LDS writes: __syncthreads() would also drain vmcnt, i.e. wait for this stage's tape stores)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                for (int q = 0; q < 3; q++)
#pragma unroll
                    for (int tau = 0; tau < 2; tau++) Xs[q].t[tau] = eb[(q * 2 + tau) * 64 + lane];
                buf ^= 1;
                RT_STAMP(4);
            }
            if (s == substeps - 1 && sol && valid) {
#pragma unroll
                for (int tau = 0; tau < 2; tau++)
                    *reinterpret_cast<f32x4v*>(sol + ((size_t)col * n_save + iv + 1) * 96 + n * 32 + 16 * tau + 4 * g) = Xn.t[tau];
            }
        }
    }
#ifdef COLNDE_STAMPS_FWD
    RT_STAMP_FLUSH();
#endif
}

// The same solve with a FOURTH wavefront (the workgroup's idle SIMD) as helper: it evaluates the Richardson-number closure of all three variables
// once per stage — the diffusive face fluxes go to LDS, the rich tape's nine pullback coefficients to HBM — while the three net waves run their
// chains; a second bare barrier per stage (B) hands the fluxes over.  Every wave executes exactly the barriers (B) and (A) in every stage.
// SPLIT (COLNDE_MATRIX_BF16X3_EXACT; RK4 and, since round 4, RKC2): layers 1 and 2 on v_mfma_f32_16x16x32_bf16 from exact three-way operand splits (rt16_forward_kernel<ACT, true>;
// image RT_SIMG2_*: rt_pack_split_ns_kernel); layer 3 and the biases stay on the tail of the fp32 image
template <int ACT, bool RICH, bool RKC = false, bool SPLIT = false>
__global__ void __launch_bounds__(256)
rt16sh_forward_kernel(DevModel m, const float* __restrict__ wimg, const float* __restrict__ x0, const float* __restrict__ bcs,
                     const float* __restrict__ save_times, int n_save, int substeps, float* __restrict__ sol,
                     float* __restrict__ t16_tape, float* __restrict__ t16_ztape, int n_col) {
    float* wl = rt_smem;
    const u32x4* simg = reinterpret_cast<const u32x4*>(rt_smem);
    f32x4v* ex = reinterpret_cast<f32x4v*>(rt_smem + ((RT_IMG_FLOATS + 3) & ~3));          // [2 buffers][3 variables][2 tiles][64 lanes]
    if constexpr (SPLIT) {
        // LDS: [bf16 operand image RT_SIMG2_WORDS][fp32 image from RT_W3C on: W3 and the biases][exchange][closure]
        const u32x4* src = reinterpret_cast<const u32x4*>(wimg + RT_SIMG2_OFF);
        u32x4* dst = reinterpret_cast<u32x4*>(rt_smem);
        for (int e = threadIdx.x; e < RT_SIMG2_WORDS / 4; e += 256) dst[e] = src[e];
        for (int e = threadIdx.x; e < RT_IMG_FLOATS - RT_W3C; e += 256) rt_smem[RT_SIMG2_WORDS + e] = wimg[RT_W3C + e];
        wl = rt_smem + RT_SIMG2_WORDS - RT_W3C;                                            // wl[RT_W3C ...], wl[RT_B1C ...] as in the fp32 layout
        ex = reinterpret_cast<f32x4v*>(rt_smem + RT_SIMG2_WORDS + ((RT_IMG_FLOATS - RT_W3C + 3) & ~3));
    } else {
        for (int e = threadIdx.x; e < RT_IMG_FLOATS; e += 256) wl[e] = wimg[e];
    }
    f32x4v* cl = ex + 2 * 384;                                                             // the helper wave's closure fluxes [3 variables][2 tiles][64 lanes]
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                     // 0..2: net = variable; 3: the helper (fourth SIMD)
    const bool helper = role == 3;
    const int n = helper ? 2 : role;
    const int j = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x;
    const int col = tile * 16 + j;
    const bool valid = col < n_col;
    const int colc = min(col, n_col - 1);
    const int i_ = lane & 15, g_i = i_ >> 2, r_i = i_ & 3;
    const bool live3 = r_i == 0 && g_i < 2;            // (SPLIT) this lane holds a row of the net's fourth layer-1 tile
    const int l83 = g_i * 4 + (lane >> 4);
    int a1b[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int Q = 4 * t + r_i, f = 4 * Q + g_i;
        a1b[t] = RT_W1C + (n * 50 + ((Q < 13 && f < 50) ? f : 0)) * RT_LD1 + 4 * g;       // padding rows read a valid row; never consumed
    }
    int a2b[2], a2l[2], a3b[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int Q2 = 4 * u + r_i;
        const int row2 = n * 20 + (Q2 < 5 ? 4 * Q2 + g_i : 0);
        a2b[u] = RT_W2C + row2 * RT_LD2 + g;
        a2l[u] = RT_W2C + row2 * RT_LD2 + (g < 2 ? 48 + g : 50);
        a3b[u] = RT_W3C + (n * 31 + 16 * u + i_ - 1) * RT_LD3 + g;
    }
    float bcb, bct, bc5;
    {
        const float* bp = bcs + (size_t)colc * 6;
        bcb = bp[2 * n];
        bct = bp[2 * n + 1];
        bc5 = bp[5];
    }
    V16 Xs[3], Xn, Kacc;
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int tau = 0; tau < 2; tau++) Xs[q].t[tau] = *reinterpret_cast<const f32x4v*>(x0 + (size_t)colc * 96 + q * 32 + 16 * tau + 4 * g);
    // (wave-uniform n: a select, not a dynamic register index)
#pragma unroll
    for (int tau = 0; tau < 2; tau++)
#pragma unroll
        for (int r = 0; r < 4; r++) Xn.t[tau][r] = n == 0 ? Xs[0].t[tau][r] : (n == 1 ? Xs[1].t[tau][r] : Xs[2].t[tau][r]);
    if (sol && valid && !helper)
#pragma unroll
        for (int tau = 0; tau < 2; tau++) *reinterpret_cast<f32x4v*>(sol + ((size_t)col * n_save) * 96 + n * 32 + 16 * tau + 4 * g) = Xn.t[tau];
    const int n_steps = (n_save - 1) * substeps;
    // stepper: classical RK4 (nst = 4) or the s-stage RKC2 step of colnde_dev.h (m.rkc: coefficient table; increment form as in tile16's
    // forward_kernel) — a template parameter: as a run-time switch it cost the RK4 latency kernels 8 % (8 simulations: 16.6 -> 18.2 ms)
    const int nst = RKC ? m.nst : 4;
    constexpr bool rkc = RKC;
    const float* mu_t = m.rkc, *nu_t = m.rkc + RKC_LD, *mut_t = m.rkc + 2 * RKC_LD, *gat_t = m.rkc + 3 * RKC_LD, *c_t = m.rkc + 4 * RKC_LD;
    float* tp = t16_tape ? t16_tape + (size_t)tile * n_steps * nst * 1536 + j * 96 + n * 32 + 4 * g : nullptr;
    float* tz = (t16_ztape && !RICH) ? t16_ztape + (size_t)tile * n_steps * nst * (16 * 216) + j * 216 + n * 72 + g : nullptr;
    float* tr = (t16_ztape && RICH) ? t16_ztape + (size_t)tile * n_steps * nst * RT16S_RREC + lane * 4 : nullptr;
    const float Nz = 32.0f;
    const float L2E = 1.4426950408889634f;
    const float cU = m.sig_u * Nz, sU = m.sig_u * m.eps, cV = m.sig_v * Nz, sV = m.sig_v * m.eps, cB = m.B * Nz, sB = m.B * m.eps;
    const float kE = 2.0f * m.inv_dRi * L2E, oE = -2.0f * m.Ric * m.inv_dRi * L2E, cE = 30.0f * L2E;
    const float nA = -0.5f * m.nu_minus, nB = m.nu0 + 0.5f * m.nu_minus;
    const float s0n = n == 0 ? m.s0[0] : (n == 1 ? m.s0[1] : m.s0[2]);
    const float An = n == 0 ? m.A[0] : (n == 1 ? m.A[1] : m.A[2]);
    int step = 0, buf = 0;
    RT_STAMP_DECL;
    for (int iv = 0; iv < n_save - 1; iv++) {
        const float t0 = save_times[iv];
        const float dt = (save_times[iv + 1] - t0) / (float)substeps;
        for (int s = 0; s < substeps; s++, step++) {
            const float ts = t0 + (float)s * dt;
            Kacc.t[0] = (f32x4t)(0.0f);
            Kacc.t[1] = (f32x4t)(0.0f);
            V16 Ym1, Ym2, F0;                           // RKC2: d_{j-1}, d_{j-2} (increments Y_j - Y_0) and F_0 of variable n
#pragma unroll
            for (int tau = 0; tau < 2; tau++) { Ym1.t[tau] = (f32x4t)(0.0f); Ym2.t[tau] = (f32x4t)(0.0f); F0.t[tau] = (f32x4t)(0.0f); }
#pragma nounroll
            for (int st = 0; st < nst; st++) {
                const float ca = rkc ? c_t[st] : (st == 0 ? 0.0f : (st == 3 ? 1.0f : 0.5f));
                const float cb = (st == 0 || st == 3) ? 1.0f / 6.0f : 1.0f / 3.0f;
                RT_STAMP_BEGIN();
                f32x4v* eb = ex + buf * 384;
                if (helper) {
                    // ---- the Richardson-number closure of all three variables (predict_flux), once, on the fourth SIMD, while the net waves run
                    //      their chains: diffusive face fluxes to LDS, and (RICH) the nine pullback coefficients to the tape
                    float* orr = tr ? tr + ((size_t)step * nst + st) * RT16S_RREC : nullptr;
                    V16 C[3];
                    if (m.mpp) {
                        const V16 Ud = shift_down16(Xs[0], lane, 0.0f), Vd = shift_down16(Xs[1], lane, 0.0f), Td = shift_down16(Xs[2], lane, 0.0f);
                        const float f0 = -m.cs[0] * Nz, f1 = -m.cs[1] * Nz, f2 = -m.cs[2] * m.inv_Pr * Nz;
                        const float m0 = -m.cs[0], m1 = -m.cs[1], m2 = -m.cs[2] * m.inv_Pr, nr = Nz * m.c_rib;
#pragma unroll
                        for (int tau = 0; tau < 2; tau++)
#pragma unroll
                            for (int r = 0; r < 4; r += 2) {
                                const f32x2v dU = {Xs[0].t[tau][r] - Ud.t[tau][r], Xs[0].t[tau][r + 1] - Ud.t[tau][r + 1]};
                                const f32x2v dV = {Xs[1].t[tau][r] - Vd.t[tau][r], Xs[1].t[tau][r + 1] - Vd.t[tau][r + 1]};
                                const f32x2v dT = {Xs[2].t[tau][r] - Td.t[tau][r], Xs[2].t[tau][r + 1] - Td.t[tau][r + 1]};
                                const f32x2v a1 = dU * cU + sU, a2 = dV * cV + sV;
                                const f32x2v s2 = a2 * a2 + a1 * a1;
                                f32x2v rS;
                                rS.x = __builtin_amdgcn_rcpf(s2.x);
                                rS.y = __builtin_amdgcn_rcpf(s2.y);
                                const f32x2v Ri = (dT * cB + sB) * rS;
                                const f32x2v arg = Ri * kE + oE;
                                f32x2v e;
                                e.x = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(arg.x, -cE, cE));
                                e.y = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(arg.y, -cE, cE));
                                const f32x2v e1 = e + 1.0f;
                                f32x2v rc;
                                rc.x = __builtin_amdgcn_rcpf(e1.x);
                                rc.y = __builtin_amdgcn_rcpf(e1.y);
                                const f32x2v th = rc * -2.0f + 1.0f;
                                const f32x2v nu = th * nA + nB;
                                const f32x2v c0 = (nu * dU) * f0, c1 = (nu * dV) * f1, c2 = (nu * dT) * f2;
                                C[0].t[tau][r] = c0.x; C[0].t[tau][r + 1] = c0.y;
                                C[1].t[tau][r] = c1.x; C[1].t[tau][r + 1] = c1.y;
                                C[2].t[tau][r] = c2.x; C[2].t[tau][r + 1] = c2.y;
                                if (RICH && orr) {
                                    const bool z0 = tau == 0 && r == 0 && g == 0;            // face 0 carries no diffusive flux
                                    const f32x2v wf = (1.0f - th * th) * rS, wR = wf * Ri;
                                    f32x2v pn = nu, pc0 = wR * (a1 * (-2.0f * m.sig_u)), pc1 = wR * (a2 * (-2.0f * m.sig_v)), pc2 = wf * m.B;
                                    if (z0) { pn.x = 0.0f; pc0.x = 0.0f; pc1.x = 0.0f; pc2.x = 0.0f; }
                                    float* o2 = orr + (36 + tau) * 256 + r;
                                    const f32x2v d0 = dU * (m0 * nr), d1 = dV * (m1 * nr), d2 = dT * (m2 * nr), q0 = pn * m0, q1 = pn * m1, q2 = pn * m2;
                                    *reinterpret_cast<f32x2v*>(o2 + 0 * 512) = d0;  *reinterpret_cast<f32x2v*>(o2 + 1 * 512) = d1;  *reinterpret_cast<f32x2v*>(o2 + 2 * 512) = d2;
                                    *reinterpret_cast<f32x2v*>(o2 + 3 * 512) = q0;  *reinterpret_cast<f32x2v*>(o2 + 4 * 512) = q1;  *reinterpret_cast<f32x2v*>(o2 + 5 * 512) = q2;
                                    *reinterpret_cast<f32x2v*>(o2 + 6 * 512) = pc0; *reinterpret_cast<f32x2v*>(o2 + 7 * 512) = pc1; *reinterpret_cast<f32x2v*>(o2 + 8 * 512) = pc2;
                                }
                            }
                    } else {
                        const V16 Td = shift_down16(Xs[2], lane, 0.0f);
#pragma unroll
                        for (int tau = 0; tau < 2; tau++)
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const bool in = !(tau == 0 && r == 0 && g == 0);
                                const float gT = (Xs[2].t[tau][r] - Td.t[tau][r]) * Nz;
                                C[0].t[tau][r] = 0.0f;
                                C[1].t[tau][r] = 0.0f;
                                C[2].t[tau][r] = (m.ca && in) ? -m.cs[2] * m.kappa * fminf(0.0f, gT) : 0.0f;
                                if (RICH && orr && m.ca) {
                                    float* o2 = orr + (36 + tau) * 256 + r;
#pragma unroll
                                    for (int a9 = 0; a9 < 9; a9++) o2[a9 * 512] = (a9 == 5 && in && gT < 0.0f) ? -m.cs[2] * m.kappa : 0.0f;
                                }
                            }
                    }
#pragma unroll
                    for (int k = 0; k < 3; k++)
#pragma unroll
                        for (int tau = 0; tau < 2; tau++) cl[(k * 2 + tau) * 64 + lane] = C[k].t[tau];
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // (B) the closure is in LDS
                } else {
                V16 Xme;               // (element-wise selects on the wave-uniform n: a select between the aggregates becomes a scratch array)
#pragma unroll
                for (int tau = 0; tau < 2; tau++)
#pragma unroll
                    for (int r = 0; r < 4; r++) Xme.t[tau][r] = n == 0 ? Xs[0].t[tau][r] : (n == 1 ? Xs[1].t[tau][r] : Xs[2].t[tau][r]);
                if (tp) {
                    float* o = tp + ((size_t)step * nst + st) * 1536;
#pragma unroll
                    for (int tau = 0; tau < 2; tau++) *reinterpret_cast<f32x4v*>(o + 16 * tau) = Xme.t[tau];
                }
                float* oz = tz ? tz + ((size_t)step * nst + st) * (16 * 216) : nullptr;
                float* orr = tr ? tr + ((size_t)step * nst + st) * RT16S_RREC : nullptr;
                const float top_raw = n == 2 ? rt_top_flux(m, bc5, ts + ca * dt) : bct;
                RT_STAMP(0);
                // ---- net n ----------------------------------------------------------------------------------------------
                int lz = lane;
                Bf3 XB[3];
                if constexpr (SPLIT) {
                    asm volatile("" : "+v"(lz));          // operand addresses stay (lane base + immediate): see rt16_forward_kernel
                    // only the first k-block's split stands in front of the products: those of blocks 1 and 2 follow tile 0's MFMAs of the block before
                    // (one wave per SIMD: nothing else hides their 88 vector instructions)
                    const float x8[8] = {Xs[0].t[0][0], Xs[0].t[0][1], Xs[0].t[0][2], Xs[0].t[0][3], Xs[0].t[1][0], Xs[0].t[1][1], Xs[0].t[1][2], Xs[0].t[1][3]};
                    XB[0] = bf3_split8(x8);
                }
                f32x4t A1[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    f32x4t acc;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int Q = 4 * t + r;
                        acc[r] = Q < 13 ? wl[RT_B1C + n * 50 + min(4 * Q + g, 49)] : 0.0f;
                    }
                    if constexpr (SPLIT) {
                        __builtin_amdgcn_s_setprio(1);
#pragma unroll
                        for (int q = 0; q < 3; q++) {
                            Bf3 A;
                            if (t < 3) A = rt16_ldA(simg, (n * 3 + t) * 3 + q, lz);
                            else {          // the net's fourth tile holds quad 12 alone: rows 0 and 4 (features 48, 49); every other lane reads the zero operand
                                const int at = live3 ? RT_SIMG2_T3 / 4 + (n * 3 + q) * 24 + l83 + (lz - lane) : RT_SIMG2_ZERO / 4 + (lz - lane);
                                A.h = simg[at]; A.m = simg[at + (live3 ? 8 : 0)]; A.l = simg[at + (live3 ? 16 : 0)];
                            }
                            acc = mfma16_bf3(A, XB[q], acc);
                            if (t == 0 && q < 2) {
                                const int qn = q < 2 ? q + 1 : 2;
                                const float n8[8] = {Xs[qn].t[0][0], Xs[qn].t[0][1], Xs[qn].t[0][2], Xs[qn].t[0][3], Xs[qn].t[1][0], Xs[qn].t[1][1], Xs[qn].t[1][2], Xs[qn].t[1][3]};
                                XB[qn] = bf3_split8(n8);
                            }
                        }
                        __builtin_amdgcn_s_setprio(0);
                        RT_SCHED_FENCE();
                    } else {
                    const int base = a1b[t];
                    acc = rt16_chain<24, 8>(wl, acc, [=](int k) { return base + (k >> 3) * 32 + ((k >> 2) & 1) * 16 + (k & 3); },
                                            [&](int k) { return Xs[k >> 3].t[(k >> 2) & 1][k & 3]; });
                    }
                    if (oz) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int Q = 4 * t + r;
                            if (Q < 13 && (Q < 12 || g < 2)) oz[4 * Q] = acc[r];             // feature 4 Q + g of layer 1
                        }
                    }
                    if (RICH && orr) {
                        f32x4t dd;
                        rt16_act_pair<ACT>(acc, A1[t], dd);
                        if (t == 3) { dd[1] = 0.0f; dd[2] = 0.0f; dd[3] = 0.0f; if (g >= 2) dd[0] = 0.0f; }   // padding: quads >= 13, features 50, 51
                        *reinterpret_cast<f32x4v*>(orr + (n * 12 + t) * 256) = A1[t];
                        *reinterpret_cast<f32x4v*>(orr + (n * 12 + 4 + t) * 256) = dd;
                    } else
                        A1[t] = rt_act4<ACT>(acc);
                }
                RT_STAMP(1);
                f32x4t A2[2];
                Bf3 HB[2];
                if constexpr (SPLIT) {
                    // the net's 13 quads of layer-1 activations as two 32-deep k-blocks (element e of block c: quad 8 c + e; three zero slots)
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        float a8[8];
#pragma unroll
                        for (int e = 0; e < 8; e++) a8[e] = 8 * c + e < 13 ? A1[(8 * c + e) >> 2][(8 * c + e) & 3] : 0.0f;
                        HB[c] = bf3_split8(a8);
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    f32x4t acc;
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[r] = (4 * u + r < 5) ? wl[RT_B2C + n * 20 + 4 * (4 * u + r) + g] : 0.0f;
                    if constexpr (SPLIT) {
                        const Bf3 Aa = rt16_ldA(simg, 27 + (n * 2 + u) * 2, lz), Ab = rt16_ldA(simg, 27 + (n * 2 + u) * 2 + 1, lz);
                        __builtin_amdgcn_s_setprio(1);
                        acc = mfma16_bf3(Aa, HB[0], acc);
                        acc = mfma16_bf3(Ab, HB[1], acc);
                        __builtin_amdgcn_s_setprio(0);
                        RT_SCHED_FENCE();
                    } else {
                    const int base = a2b[u], basel = a2l[u];
                    acc = rt16_chain<13, 13>(wl, acc, [=](int k) { return k < 12 ? base + 4 * k : basel; },
                                             [&](int k) { return A1[k >> 2][k & 3]; });
                    }
                    if (oz) {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (4 * u + r < 5) oz[52 + 4 * (4 * u + r)] = acc[r];            // feature 4 Q2 + g of layer 2
                    }
                    if (RICH && orr) {
                        f32x4t dd;
                        rt16_act_pair<ACT>(acc, A2[u], dd);
                        if (u == 1) { dd[1] = 0.0f; dd[2] = 0.0f; dd[3] = 0.0f; }                             // padding: quads >= 5
                        *reinterpret_cast<f32x4v*>(orr + (n * 12 + 8 + u) * 256) = A2[u];
                        *reinterpret_cast<f32x4v*>(orr + (n * 12 + 10 + u) * 256) = dd;
                    } else
                        A2[u] = rt_act4<ACT>(acc);
                }
                V16 O;
#pragma unroll
                for (int v = 0; v < 2; v++) {
                    f32x4t acc;
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[r] = wl[RT_B3C + n * 32 + 16 * v + 4 * g + r];
                    const int base = a3b[v];
                    O.t[v] = rt16_chain<5, 5>(wl, acc, [=](int k) { return base + 4 * k; }, [&](int k) { return A2[k >> 2][k & 3]; });
                }
                RT_STAMP(2);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");           // (B) the helper's closure fluxes are in LDS
                // ---- physics: face flux = NN flux + closure flux, tendency of variable n (predict_flux / predict_NDE) ---------
                V16 F;
#pragma unroll
                for (int tau = 0; tau < 2; tau++) F.t[tau] = O.t[tau] + cl[(n * 2 + tau) * 64 + lane];
                if (g == 0) F.t[0][0] = m.mpp ? (m.zero_w ? bcb - s0n : bcb) : (m.zero_w ? 0.0f : bcb);      // face 0: the bottom boundary
                {
                    const float top = m.zero_w ? top_raw - s0n : top_raw;
                    const V16 Fu = shift_up16(F, lane, top);
#pragma unroll
                    for (int tau = 0; tau < 2; tau++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            float v = -An * (Fu.t[tau][r] - F.t[tau][r]);
                            if (n == 0) v += m.cor_u * (m.sig_v * Xs[1].t[tau][r] + m.mu_v);
                            if (n == 1) v -= m.cor_v * (m.sig_u * Xs[0].t[tau][r] + m.mu_u);
                            F.t[tau][r] = v;                                                // F now holds the tendency of variable n
                        }
                }
                RT_STAMP(3);
                // ---- RK4 bookkeeping for variable n; the next stage input (or, after stage 3, the new state) is exchanged -------------
                V16 Xnext;
                if (rkc) {
                    // Y_j, j = st + 1: d_1 = mu~_1 h F_0;  d_j = mu_j d_{j-1} + nu_j d_{j-2} + mu~_j h F_{j-1} + gamma~_j h F_0;  Y_s ends the step
                    const int jj = st + 1;
                    const float cmu = mu_t[jj], cnu = nu_t[jj], cmt = mut_t[jj] * dt, cga = gat_t[jj] * dt;
#pragma unroll
                    for (int tau = 0; tau < 2; tau++) {
                        if (st == 0) F0.t[tau] = F.t[tau];
                        const f32x4t dj = st == 0 ? cmt * F0.t[tau] : cmu * Ym1.t[tau] + cnu * Ym2.t[tau] + cmt * F.t[tau] + cga * F0.t[tau];
                        Xnext.t[tau] = Xn.t[tau] + dj;
                        Ym2.t[tau] = Ym1.t[tau];
                        Ym1.t[tau] = dj;
                        if (jj == nst) Xn.t[tau] = Xnext.t[tau];
                    }
                } else {
#pragma unroll
                for (int tau = 0; tau < 2; tau++) {
                    Kacc.t[tau] += cb * F.t[tau];
                    if (st < 3) {
                        const float can = st == 2 ? 1.0f : 0.5f;
                        Xnext.t[tau] = Xn.t[tau] + (can * dt) * F.t[tau];
                    } else {
                        Xn.t[tau] += dt * Kacc.t[tau];
                        Xnext.t[tau] = Xn.t[tau];
                    }
                }
                }
#pragma unroll
                for (int tau = 0; tau < 2; tau++) eb[(n * 2 + tau) * 64 + lane] = Xnext.t[tau];
                }
                // (a bare barrier behind the LDS writes: __syncthreads() would also drain vmcnt, i.e. wait for this stage's tape stores)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                for (int q = 0; q < 3; q++)
#pragma unroll
                    for (int tau = 0; tau < 2; tau++) Xs[q].t[tau] = eb[(q * 2 + tau) * 64 + lane];
                buf ^= 1;
                RT_STAMP(4);
            }
            if (s == substeps - 1 && sol && valid && !helper) {
#pragma unroll
                for (int tau = 0; tau < 2; tau++)
                    *reinterpret_cast<f32x4v*>(sol + ((size_t)col * n_save + iv + 1) * 96 + n * 32 + 16 * tau + 4 * g) = Xn.t[tau];
            }
        }
    }
#ifdef COLNDE_STAMPS_FWD
    RT_STAMP_FLUSH();
#endif
}

// ------------------------------------------------------------------------------------------------
// discrete adjoint of ONE 16-column tile by THREE wavefronts (the companion of rt16s_forward_kernel at the latency points).
//
// Wave n back-propagates flux net n: from the taped hidden pre-activations (no forward recomputation) through the three transposed
// chains W3^T, W2^T, W1^T (114 dependent 16x16x4 MFMAs instead of 342 in a row).  The physics pullback couples the three variables
// through the Richardson number and is cheap beside the chains, so every wave evaluates it in full — the three copies of λ, x̄ and
// the stage cotangent are bit-identical by construction (same operations, same order).  What a wave alone knows is its net's part of
// the state cotangent, W1_n^T δz1_n (96 rows): the three parts go through a double-buffered LDS exchange (one barrier per stage) and
// are summed in net order by everybody.
//
// The weight gradients are not accumulated here: the kernel writes tile16's delta-tape records
//   [tile][step][stage][column][ x (96) | a of net 0..2 (104 each: a1 at 0, a2 at 52) | δz of net 0..2 (104 each: δz1 at 0, δz2 at 52, δz3 at 72) ]
// and tile16's split-K dW GEMM contracts them (dw_gemm_lds_kernel) — no transposition, no accumulator registers, no flush code here.
// Bias gradients (column sums of the deltas) and the loss sums go to the tile's slab row, as in tile16's taped adjoint.
// ------------------------------------------------------------------------------------------------
#define RT16S_REC (16 * 720)       // floats per delta-tape record: 16 columns x (96 + 2 x 3 x 104)
#define RT16S_ZREC (16 * 216)      // floats per pre-activation tape record: 16 columns x 3 nets x 72
#define RT16S_STG (16 * 180)       // floats of one wave's record staging area in LDS

// NC independent chains of N k-steps advanced together, k-major, their A operands fetched as one group — early enough, by the caller, that the
// LDS latency is covered by other work.  (The grouping is for the operand fetch: a dependent fp32 MFMA chain issues at the full rate anyway,
// tools/probe/mfma_rate.hip.)
template <int NC, int N, class AF>
__device__ __forceinline__ void rt16_fetch_ops(const float* wl, float (&a)[NC][N], AF aidx) {
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int k = 0; k < N; k++) a[c][k] = wl[aidx(c, k)];
}
template <int NC, int N, class BF>
__device__ __forceinline__ void rt16_run_ops(const float (&a)[NC][N], f32x4t (&acc)[NC], BF bval) {
#pragma unroll
    for (int k = 0; k < N; k++)
#pragma unroll
        for (int c = 0; c < NC; c++) acc[c] = mfma16t(a[c][k], bval(k), acc[c]);
}

__device__ __forceinline__ void rt16_physics_vjp(const DevModel& m, const V16 (&X)[3], V16 (&kd)[3], int lane, V16 (&xb)[3]) {
    const float Nz = 32.0f;
    const int g = lane >> 4;
#pragma unroll
    for (int tau = 0; tau < 2; tau++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            xb[0].t[tau][r] = -m.cor_v * m.sig_u * kd[1].t[tau][r];
            xb[1].t[tau][r] = m.cor_u * m.sig_v * kd[0].t[tau][r];
            xb[2].t[tau][r] = 0.0f;
        }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const V16 dn = shift_down16(kd[k], lane, 0.0f);
#pragma unroll
        for (int tau = 0; tau < 2; tau++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                kd[k].t[tau][r] = (tau == 0 && r == 0 && g == 0) ? 0.0f : m.A[k] * (kd[k].t[tau][r] - dn.t[tau][r]);
    }
    if (!m.mpp && !m.ca) return;
    V16 gb[3];
    if (m.mpp) {
        const V16 Ud = shift_down16(X[0], lane, 0.0f), Vd = shift_down16(X[1], lane, 0.0f), Td = shift_down16(X[2], lane, 0.0f);
        // the arithmetic of rt_physics_vjp, uniform factors folded
        const float cU = m.sig_u * Nz, sU = m.sig_u * m.eps, cV = m.sig_v * Nz, sV = m.sig_v * m.eps, cB = m.B * Nz, sB = m.B * m.eps;
        const float L2E = 1.4426950408889634f;
        const float kE = 2.0f * m.inv_dRi * L2E, oE = -2.0f * m.Ric * m.inv_dRi * L2E, cE = 30.0f * L2E;
        const float nA = -0.5f * m.nu_minus, nB = m.nu0 + 0.5f * m.nu_minus;
        const float m0 = -m.cs[0], m1 = -m.cs[1], m2 = -m.cs[2] * m.inv_Pr;
        const float n0 = m0 * Nz * m.c_rib, n1 = m1 * Nz * m.c_rib, n2 = m2 * Nz * m.c_rib;
        const float q0 = -2.0f * m.sig_u, q1 = -2.0f * m.sig_v;
#pragma unroll
        for (int tau = 0; tau < 2; tau++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const bool in = !(tau == 0 && r == 0 && g == 0);
                const float dU = X[0].t[tau][r] - Ud.t[tau][r], dV = X[1].t[tau][r] - Vd.t[tau][r], dT = X[2].t[tau][r] - Td.t[tau][r];
                const float a1 = fmaf(dU, cU, sU), a2 = fmaf(dV, cV, sV);
                const float rS = __builtin_amdgcn_rcpf(fmaf(a2, a2, a1 * a1));
                const float Ri = fmaf(dT, cB, sB) * rS;
                const float e = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(fmaf(Ri, kE, oE), -cE, cE));
                const float th = fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
                const float nu = fmaf(th, nA, nB);
                const float k0 = kd[0].t[tau][r], k1 = kd[1].t[tau][r], k2 = kd[2].t[tau][r];
                const float t0 = k0 * nu, t1 = k1 * nu, t2 = k2 * nu;
                float nub = (k0 * dU) * n0;
                nub = fmaf(k1 * dV, n1, nub);
                nub = fmaf(k2 * dT, n2, nub);
                const float w = nub * (fmaf(-th, th, 1.0f) * rS);
                const float qq = w * Ri;
                const float g2 = fmaf(w, m.B, t2 * m2);
                const float g0 = fmaf(qq, a1 * q0, t0 * m0);
                const float g1 = fmaf(qq, a2 * q1, t1 * m1);
                gb[0].t[tau][r] = in ? g0 : 0.0f;
                gb[1].t[tau][r] = in ? g1 : 0.0f;
                gb[2].t[tau][r] = in ? g2 : 0.0f;
            }
    } else {
        const V16 Td = shift_down16(X[2], lane, 0.0f);
#pragma unroll
        for (int tau = 0; tau < 2; tau++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const bool in = !(tau == 0 && r == 0 && g == 0);
                const float gT = (X[2].t[tau][r] - Td.t[tau][r]) * Nz;
                gb[0].t[tau][r] = 0.0f;
                gb[1].t[tau][r] = 0.0f;
                gb[2].t[tau][r] = (in && gT < 0.0f) ? -kd[2].t[tau][r] * m.cs[2] * m.kappa : 0.0f;
            }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const V16 gu_ = shift_up16(gb[k], lane, 0.0f);
#pragma unroll
        for (int tau = 0; tau < 2; tau++)
#pragma unroll
            for (int r = 0; r < 4; r++) xb[k].t[tau][r] += (gb[k].t[tau][r] - gu_.t[tau][r]) * Nz;
    }
}

// The coupled (transcendental) half of the pullback read from the RICH tape instead of recomputed: nine coefficients per level,
//   face cotangent D_k = A_k (k̄_k[i] - k̄_k[i-1]);  g_k = nub c_k + D_k nu_k with nub = Σ_k D_k dn_k;  x̄_k += Nz (g_k[i] - g_k[i+1])
// (rt16_physics_vjp's arithmetic with the uniform factors and the level mask folded into the coefficients by the forward kernel).
// On entry kd holds k̄; on exit dO, and xb the physics part of the state cotangent.
struct PhysC { V16 dn0, dn1, dn2, nu0, nu1, nu2, c0, c1, c2; };

__device__ __forceinline__ void rt16_physics_apply(const DevModel& m, const PhysC& P, V16 (&kd)[3], int lane, V16 (&xb)[3]) {
    const float Nz = 32.0f;
    const int g = lane >> 4;
#pragma unroll
    for (int tau = 0; tau < 2; tau++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            xb[0].t[tau][r] = -m.cor_v * m.sig_u * kd[1].t[tau][r];
            xb[1].t[tau][r] = m.cor_u * m.sig_v * kd[0].t[tau][r];
            xb[2].t[tau][r] = 0.0f;
        }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const V16 dn = shift_down16(kd[k], lane, 0.0f);
#pragma unroll
        for (int tau = 0; tau < 2; tau++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                kd[k].t[tau][r] = (tau == 0 && r == 0 && g == 0) ? 0.0f : m.A[k] * (kd[k].t[tau][r] - dn.t[tau][r]);
    }
    if (!m.mpp && !m.ca) return;
    V16 gb[3];
#pragma unroll
    for (int tau = 0; tau < 2; tau++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float k0 = kd[0].t[tau][r], k1 = kd[1].t[tau][r], k2 = kd[2].t[tau][r];
            const float nub = fmaf(k2, P.dn2.t[tau][r], fmaf(k1, P.dn1.t[tau][r], k0 * P.dn0.t[tau][r]));
            gb[0].t[tau][r] = fmaf(nub, P.c0.t[tau][r], k0 * P.nu0.t[tau][r]);
            gb[1].t[tau][r] = fmaf(nub, P.c1.t[tau][r], k1 * P.nu1.t[tau][r]);
            gb[2].t[tau][r] = fmaf(nub, P.c2.t[tau][r], k2 * P.nu2.t[tau][r]);
        }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const V16 gu_ = shift_up16(gb[k], lane, 0.0f);
#pragma unroll
        for (int tau = 0; tau < 2; tau++)
#pragma unroll
            for (int r = 0; r < 4; r++) xb[k].t[tau][r] += (gb[k].t[tau][r] - gu_.t[tau][r]) * Nz;
    }
}

template <int ACT, bool RICH>
__global__ void __launch_bounds__(192)
rt16s_adjoint_kernel(DevModel m, const float* __restrict__ wimg, const float* __restrict__ save_times, int n_save, int substeps,
                     const float* __restrict__ sol, const float* __restrict__ truth, const float* __restrict__ t16_tape,
                     const float* __restrict__ t16_ztape, LossWeights lw, float* __restrict__ slab, int n_col,
                     float* __restrict__ dwtape) {
    float* wl = rt_smem;
    for (int e = threadIdx.x; e < RT_IMG_FLOATS; e += 192) wl[e] = wimg[e];
    f32x4v* ex = reinterpret_cast<f32x4v*>(rt_smem + ((RT_IMG_FLOATS + 3) & ~3));          // [2 buffers][3 nets][6 tiles][64 lanes]
    // staging of each wave's part of the delta-tape record: [16 columns][a1 52 | a2 20 | δz1 52 | δz2 20 | δz3 32] with row stride 180
    // (conflict-free for the element-wise writes, 16-byte aligned for the float4 read-back); the pad slots stay zero
    float* stg_all = rt_smem + ((RT_IMG_FLOATS + 3) & ~3) + 2 * (3 * 6 * 64) * 4;
    for (int e = threadIdx.x; e < 3 * RT16S_STG; e += 192) stg_all[e] = 0.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int n = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                        // this wave's net
    float* stg = stg_all + n * RT16S_STG;
    const int j = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x;
    const int col = tile * 16 + j;
    const bool valid = col < n_col;
    const int colc = min(col, n_col - 1);
    const int i_ = lane & 15, g_i = i_ >> 2, r_i = i_ & 3;
    // A-operand bases of the transposed products (output row i_ of a tile, k-lane g): W3^T (rows = a2 features), W2^T (rows = a1 features),
    // W1^T (rows = state features); rows that are padding read the image's zero column
    int b3T[2], b2T[4];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int Q2 = 4 * u + r_i;
        b3T[u] = RT_W3C + (n * 31 + 4 * g - 1) * RT_LD3 + (Q2 < 5 ? 4 * Q2 + g_i : 20);
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int Q = 4 * t + r_i, f = 4 * Q + g_i;
        b2T[t] = RT_W2C + (n * 20 + g) * RT_LD2 + ((Q < 13 && f < 50) ? f : 50);
    }
    const int b1T = RT_W1C + (n * 50 + g) * RT_LD1 + i_;

    V16 lam[3], xb[3], xbs[3];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int tau = 0; tau < 2; tau++) { lam[q].t[tau] = (f32x4t)(0.0f); xb[q].t[tau] = (f32x4t)(0.0f); xbs[q].t[tau] = (f32x4t)(0.0f); }
    f32x4t gb1[4], gb2[2], gb3[2];                         // bias gradients: this lane's column of every delta, summed over the stages
#pragma unroll
    for (int t = 0; t < 4; t++) gb1[t] = (f32x4t)(0.0f);
#pragma unroll
    for (int u = 0; u < 2; u++) { gb2[u] = (f32x4t)(0.0f); gb3[u] = (f32x4t)(0.0f); }
    float sum_d = 0.0f, sum_g = 0.0f;                      // loss sums of variable n (profile and gradient terms)
    RT_STAMP_DECL;

    const int n_steps = (n_save - 1) * substeps;
    const float* tp = t16_tape + (size_t)tile * n_steps * 4 * 1536 + j * 96 + 4 * g;
    const float* tz = RICH ? t16_ztape + (size_t)tile * n_steps * 4 * RT16S_RREC + lane * 4      // the rich tape (see rt16s_forward_kernel)
                           : t16_ztape + (size_t)tile * n_steps * 4 * RT16S_ZREC + j * 216 + n * 72 + g;
    const bool phys = m.mpp || m.ca;
    float* rec0 = dwtape + (size_t)tile * n_steps * 4 * RT16S_REC;
    // the record's 11 float4 pieces this lane copies from the staging area each stage: piece e = 64 i + lane = column e / 44, float4 e % 44
    // of that column's 72 a-floats and 104 δ-floats
    int stg_rd[11], rec_wr[11];
#pragma unroll
    for (int i = 0; i < 11; i++) {
        const int e = 64 * i + lane, c = e / 44, f4 = e - 44 * c;
        stg_rd[i] = c * 180 + 4 * f4;
        rec_wr[i] = c * 720 + 96 + (f4 < 18 ? n * 104 + 4 * f4 : (3 + n) * 104 + 4 * (f4 - 18));
    }
    float* sw = stg + j * 180 + g;                          // element-wise staging writes of this lane: feature 4 Q + g of column j

    // loss injection at save point sv: λ += ∂loss/∂sol[:, sv] (all three variables, every wave); the sums of squares of variable n only
    auto inject = [&](int sv, bool add) {
#pragma unroll
        for (int q = 0; q < 3; q++) {
            V16 d;
#pragma unroll
            for (int tau = 0; tau < 2; tau++) {
                const size_t o = ((size_t)colc * n_save + sv) * 96 + q * 32 + 16 * tau + 4 * g;
                const f32x4v a = *reinterpret_cast<const f32x4v*>(sol + o);
                const f32x4v b = *reinterpret_cast<const f32x4v*>(truth + o);
#pragma unroll
                for (int e = 0; e < 4; e++) d.t[tau][e] = valid ? a[e] - b[e] : 0.0f;
            }
            const V16 dd = shift_down16(d, lane, 0.0f);
            V16 gg;
            float sd = 0.0f, sg = 0.0f;
#pragma unroll
            for (int tau = 0; tau < 2; tau++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    gg.t[tau][r] = (tau == 0 && r == 0 && g == 0) ? 0.0f : (d.t[tau][r] - dd.t[tau][r]) * 32.0f;
                    sd += d.t[tau][r] * d.t[tau][r];
                    sg += gg.t[tau][r] * gg.t[tau][r];
                }
            if (q == n) { sum_d += sd; sum_g += sg; }
            if (add) {
                const V16 gu_ = shift_up16(gg, lane, 0.0f);
#pragma unroll
                for (int tau = 0; tau < 2; tau++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        lam[q].t[tau][r] += 2.0f * lw.w[q] * d.t[tau][r] + 2.0f * lw.w[3 + q] * 32.0f * (gg.t[tau][r] - gu_.t[tau][r]);
            }
        }
    };
    inject(0, false);

    // stage inputs and hidden pre-activations (RICH: variable n of the stage input, this net's activations and derivatives, the nine
    // physics coefficients), fetched one stage ahead of their use
    V16 Xp[3];
    float z1p[13], z2p[5];
    f32x4t adp[12];
    PhysC Pp;
    auto prefetch = [&](int qs) __attribute__((always_inline)) {
        const float* sx = tp + (size_t)qs * 1536;
        if (RICH) {
#pragma unroll
            for (int tau = 0; tau < 2; tau++) Xp[0].t[tau] = *reinterpret_cast<const f32x4v*>(sx + n * 32 + 16 * tau);
            const float* sr = tz + (size_t)qs * RT16S_RREC;
#pragma unroll
            for (int e = 0; e < 12; e++) adp[e] = *reinterpret_cast<const f32x4v*>(sr + (n * 12 + e) * 256);
            if (phys) {
#pragma unroll
                for (int tau = 0; tau < 2; tau++) {
                    Pp.dn0.t[tau] = *reinterpret_cast<const f32x4v*>(sr + (36 + 0 + tau) * 256);
                    Pp.dn1.t[tau] = *reinterpret_cast<const f32x4v*>(sr + (36 + 2 + tau) * 256);
                    Pp.dn2.t[tau] = *reinterpret_cast<const f32x4v*>(sr + (36 + 4 + tau) * 256);
                    Pp.nu0.t[tau] = *reinterpret_cast<const f32x4v*>(sr + (36 + 6 + tau) * 256);
                    Pp.nu1.t[tau] = *reinterpret_cast<const f32x4v*>(sr + (36 + 8 + tau) * 256);
                    Pp.nu2.t[tau] = *reinterpret_cast<const f32x4v*>(sr + (36 + 10 + tau) * 256);
                    Pp.c0.t[tau] = *reinterpret_cast<const f32x4v*>(sr + (36 + 12 + tau) * 256);
                    Pp.c1.t[tau] = *reinterpret_cast<const f32x4v*>(sr + (36 + 14 + tau) * 256);
                    Pp.c2.t[tau] = *reinterpret_cast<const f32x4v*>(sr + (36 + 16 + tau) * 256);
                }
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int tau = 0; tau < 2; tau++) Xp[q].t[tau] = *reinterpret_cast<const f32x4v*>(sx + q * 32 + 16 * tau);
        const float* sz = tz + (size_t)qs * RT16S_ZREC;
#pragma unroll
        for (int Q = 0; Q < 13; Q++) z1p[Q] = (Q < 12 || g < 2) ? sz[4 * Q] : 0.0f;
#pragma unroll
        for (int Q2 = 0; Q2 < 5; Q2++) z2p[Q2] = sz[52 + 4 * Q2];
    };
    prefetch(n_steps * 4 - 1);
    int buf = 0;

    for (int iv = n_save - 2; iv >= 0; iv--) {
        const float dt = (save_times[iv + 1] - save_times[iv]) / (float)substeps;
        inject(iv + 1, true);
        for (int s = substeps - 1; s >= 0; s--) {
            const int step = iv * substeps + s;
#pragma nounroll
            for (int st = 3; st >= 0; st--) {
                const float cwl = (st == 0 || st == 3) ? dt / 6.0f : dt / 3.0f;
                const float cwx = st == 3 ? 0.0f : (st == 2 ? dt : 0.5f * dt);
                const int qs = step * 4 + st;
                RT_STAMP_BEGIN();
                V16 X[3];
                f32x4t Z1[4], Z2[2];
                f32x4t A1[4], D1[4], A2[2], D2[2];
                V16 kb[3], xbp[3];
                // (1) stage cotangent k̄ = cwl λ + cwx x̄ (x̄: the state cotangent of the stage handled before), then the physics pullback
#pragma unroll
                for (int q = 0; q < 3; q++)
#pragma unroll
                    for (int tau = 0; tau < 2; tau++) kb[q].t[tau] = cwl * lam[q].t[tau] + cwx * xb[q].t[tau];
                if (RICH) {
                    X[0] = Xp[0];                                                        // variable n only
#pragma unroll
                    for (int t = 0; t < 4; t++) { A1[t] = adp[t]; D1[t] = adp[4 + t]; }
#pragma unroll
                    for (int u = 0; u < 2; u++) { A2[u] = adp[8 + u]; D2[u] = adp[10 + u]; }
                    rt16_physics_apply(m, Pp, kb, lane, xbp);                            // kb now holds dO
                    if (qs > 0) prefetch(qs - 1);
                } else {
#pragma unroll
                    for (int q = 0; q < 3; q++) X[q] = Xp[q];
#pragma unroll
                    for (int t = 0; t < 4; t++)
#pragma unroll
                        for (int r = 0; r < 4; r++) Z1[t][r] = (4 * t + r < 13) ? z1p[(4 * t + r) < 13 ? 4 * t + r : 0] : 0.0f;
#pragma unroll
                    for (int u = 0; u < 2; u++)
#pragma unroll
                        for (int r = 0; r < 4; r++) Z2[u][r] = (4 * u + r < 5) ? z2p[(4 * u + r) < 5 ? 4 * u + r : 0] : 0.0f;
                    if (qs > 0) prefetch(qs - 1);
                    rt16_physics_vjp(m, X, kb, lane, xbp);                               // kb now holds dO
                }
                RT_STAMP(0);
                V16 dO;                 // (element-wise selects on the wave-uniform n: a select between the aggregates becomes a scratch array)
#pragma unroll
                for (int tau = 0; tau < 2; tau++)
#pragma unroll
                    for (int r = 0; r < 4; r++) dO.t[tau][r] = n == 0 ? kb[0].t[tau][r] : (n == 1 ? kb[1].t[tau][r] : kb[2].t[tau][r]);
                // operands of the W3^T and W2^T products (fetched here: the activations below cover their LDS latency)
                float a3[2][8], a2[4][5];
                rt16_fetch_ops<2, 8>(wl, a3, [&](int u, int k) { return b3T[u] + (16 * (k >> 2) + (k & 3)) * RT_LD3; });
                rt16_fetch_ops<4, 5>(wl, a2, [&](int t, int k) { return b2T[t] + 4 * k * RT_LD2; });
                RT_SCHED_HARD();
                // (2) activations and their derivatives from the taped pre-activations (RICH: taped as such)
                if (!RICH) {
#pragma unroll
                    for (int t = 0; t < 4; t++) rt16_act_pair<ACT>(Z1[t], A1[t], D1[t]);
#pragma unroll
                    for (int u = 0; u < 2; u++) rt16_act_pair<ACT>(Z2[u], A2[u], D2[u]);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        if (r > 0) { D1[3][r] = 0.0f; D2[1][r] = 0.0f; }             // padding quads: Q >= 13, Q2 >= 5
                    }
                    if (g >= 2) D1[3][0] = 0.0f;                                         // features 50, 51 of quad 12
                }
                float* rec = rec0 + (size_t)qs * RT16S_REC;
                {
#pragma unroll
                    for (int tau = 0; tau < 2; tau++) {
                        f32x4v xv;
#pragma unroll
                        for (int r = 0; r < 4; r++) xv[r] = (RICH || n == 0) ? X[0].t[tau][r] : (n == 1 ? X[1].t[tau][r] : X[2].t[tau][r]);
                        *reinterpret_cast<f32x4v*>(rec + j * 720 + n * 32 + 16 * tau + 4 * g) = xv;
                    }
#pragma unroll
                    for (int Q = 0; Q < 13; Q++)
                        if (Q < 12 || g < 2) sw[4 * Q] = A1[Q >> 2][Q & 3];
#pragma unroll
                    for (int Q2 = 0; Q2 < 5; Q2++) sw[52 + 4 * Q2] = A2[Q2 >> 2][Q2 & 3];
                }
                RT_STAMP(1);
                // (3) δz2 = (W3^T dO) ∘ act'(z2): 8 k-steps = the faces (v, r), lane g holding face 16 v + 4 g + r
                f32x4t dZ2[2] = {(f32x4t)(0.0f), (f32x4t)(0.0f)};
                rt16_run_ops<2, 8>(a3, dZ2, [&](int k) { return dO.t[k >> 2][k & 3]; });
#pragma unroll
                for (int u = 0; u < 2; u++) dZ2[u] *= D2[u];
                // operands of the first two W1^T tiles, in flight under the W2^T products
                float a1[2][2][13];
                rt16_fetch_ops<2, 13>(wl, a1[0], [&](int c, int k) { return b1T + 16 * c + 4 * k * RT_LD1; });
                RT_SCHED_HARD();
                // (4) δz1 = (W2^T δz2) ∘ act'(z1): 5 k-steps = the quads of a2
                f32x4t dZ1[4] = {(f32x4t)(0.0f), (f32x4t)(0.0f), (f32x4t)(0.0f), (f32x4t)(0.0f)};
                rt16_run_ops<4, 5>(a2, dZ1, [&](int k) { return dZ2[k >> 2][k & 3]; });
#pragma unroll
                for (int t = 0; t < 4; t++) dZ1[t] *= D1[t];
                RT_STAMP(2);
                {
#pragma unroll
                    for (int Q = 0; Q < 13; Q++)
                        if (Q < 12 || g < 2) sw[72 + 4 * Q] = dZ1[Q >> 2][Q & 3];
#pragma unroll
                    for (int Q2 = 0; Q2 < 5; Q2++) sw[124 + 4 * Q2] = dZ2[Q2 >> 2][Q2 & 3];
                    float* s3 = stg + j * 180 + 144 + 4 * g - 1;                         // output o = face - 1
#pragma unroll
                    for (int v = 0; v < 2; v++)
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (!(v == 0 && r == 0 && g == 0)) s3[16 * v + r] = dO.t[v][r];
                    // the wave's 16 x 176 floats leave as 11 float4 stores per lane (the element-wise form costs 52 scattered dword stores)
#pragma unroll
                    for (int i = 0; i < 11; i++) *reinterpret_cast<f32x4v*>(rec + rec_wr[i]) = *reinterpret_cast<const f32x4v*>(stg + stg_rd[i]);
                }
#pragma unroll
                for (int t = 0; t < 4; t++) gb1[t] += dZ1[t];
#pragma unroll
                for (int u = 0; u < 2; u++) { gb2[u] += dZ2[u]; gb3[u] += dO.t[u]; }
                RT_STAMP(3);
                // (5) this net's part of the state cotangent, W1_n^T δz1 (6 tiles x 13 k-steps), exchanged and summed in net order
                f32x4v* eb = ex + buf * (3 * 6 * 64);
#pragma unroll
                for (int q = 0; q < 3; q++) {                                            // variable q: its two 16-level tiles together
                    if (q < 2) rt16_fetch_ops<2, 13>(wl, a1[(q + 1) & 1], [&](int c, int k) { return b1T + 32 * (q + 1) + 16 * c + 4 * k * RT_LD1; });
                    RT_SCHED_HARD();
                    f32x4t c2[2] = {(f32x4t)(0.0f), (f32x4t)(0.0f)};
                    rt16_run_ops<2, 13>(a1[q & 1], c2, [&](int k) { return dZ1[k >> 2][k & 3]; });
#pragma unroll
                    for (int tau = 0; tau < 2; tau++) eb[(n * 6 + q * 2 + tau) * 64 + lane] = c2[tau];
                    RT_SCHED_HARD();
                }
                RT_STAMP(4);
                // (a bare barrier behind the LDS writes: __syncthreads() would also drain vmcnt — this stage's tape stores and the next
                //  stage's prefetch)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                for (int q = 0; q < 3; q++)
#pragma unroll
                    for (int tau = 0; tau < 2; tau++) {
                        const f32x4v c0 = eb[(0 * 6 + q * 2 + tau) * 64 + lane], c1 = eb[(1 * 6 + q * 2 + tau) * 64 + lane],
                                     c2 = eb[(2 * 6 + q * 2 + tau) * 64 + lane];
                        xb[q].t[tau] = xbp[q].t[tau] + ((c0 + c1) + c2);
                        xbs[q].t[tau] += xb[q].t[tau];
                    }
                buf ^= 1;
                RT_STAMP(5);
            }
            // λ_n = λ_{n+1} + x̄_1 + x̄_2 + x̄_3 + x̄_4
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int tau = 0; tau < 2; tau++) { lam[q].t[tau] += xbs[q].t[tau]; xbs[q].t[tau] = (f32x4t)(0.0f); }
        }
    }

#ifndef COLNDE_STAMPS_FWD
    RT_STAMP_FLUSH();
#endif
    // ---- the tile's slab row: bias gradients of net n (sums over the 16 columns = the lanes of a g-group) and the loss sums ----
    float* out = slab + (size_t)tile * (m.n_params + 8);
    auto colsum = [&](float v) {
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
        return v;
    };
#pragma unroll
    for (int Q = 0; Q < 13; Q++) {
        const float v = colsum(gb1[Q >> 2][Q & 3]);
        if (j == 0 && (Q < 12 || g < 2)) out[n * m.net_size + m.b_off[0] + 4 * Q + g] = v;
    }
#pragma unroll
    for (int Q2 = 0; Q2 < 5; Q2++) {
        const float v = colsum(gb2[Q2 >> 2][Q2 & 3]);
        if (j == 0) out[n * m.net_size + m.b_off[1] + 4 * Q2 + g] = v;
    }
#pragma unroll
    for (int v_ = 0; v_ < 2; v_++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float v = colsum(gb3[v_][r]);
            const int face = 16 * v_ + 4 * g + r;
            if (j == 0 && face >= 1) out[n * m.net_size + m.b_off[2] + face - 1] = v;
        }
    float sd = sum_d, sg = sum_g;
    for (int off = 32; off > 0; off >>= 1) { sd += __shfl_down(sd, off); sg += __shfl_down(sg, off); }
    if (lane == 0) {
        out[m.n_params + n] = sd;
        out[m.n_params + 3 + n] = sg;
        if (n == 0) { out[m.n_params + 6] = 0.0f; out[m.n_params + 7] = 0.0f; }
    }
}

// ------------------------------------------------------------------------------------------------
// The net-split adjoint with a FOURTH wavefront.  In rt16s_adjoint_kernel every net wave carries the whole state of the adjoint — λ, x̄, the
// stage cotangent, the physics pullback — in three bit-identical copies.  Here ONE wave (the helper, on the workgroup's fourth SIMD) carries
// it: per stage it forms k̄ = cwl λ + cwx x̄, runs the physics pullback once and hands the three dO over through LDS (barrier B); the net waves
// back-propagate their nets from dO_n, write the delta-tape record and their W1_n^T δz1_n parts to LDS (barrier A); the helper sums the
// parts into x̄.  While the helper works (between A and B) the net waves do what depends on the tapes alone: next stage's prefetch, this
// stage's activations, the record's x and a parts.  Every wave executes exactly the barriers B and A in every stage.
// ------------------------------------------------------------------------------------------------
// SPLIT (COLNDE_MATRIX_BF16X3_EXACT, RK4): the net waves' W1_n^T δz1_n products — 78 of a stage's 114 fp32 MFMAs, 3.0 k of its 8.2 k cycles — on
// v_mfma_f32_16x16x32_bf16 from exact three-way operand splits: δz1 is already held as the B operand needs it (lane group kq holds the features
// 4 Q + kq of its column: two 32-deep k-blocks, quads 0..7 and 8..12 + padding), W1_n^T comes pre-split (RT_NSA_*), planes h and m from LDS in the
// fp32 W1's place (the rest of the fp32 image moves down), plane l from L2 into registers before barrier B.
template <int ACT, bool RICH, bool RKC = false, bool SPLIT = false>
__global__ void __launch_bounds__(256)
rt16sh_adjoint_kernel(DevModel m, const float* __restrict__ wimg, const float* __restrict__ save_times, int n_save, int substeps,
                      const float* __restrict__ sol, const float* __restrict__ truth, const float* __restrict__ t16_tape,
                      const float* __restrict__ t16_ztape, LossWeights lw, float* __restrict__ slab, int n_col,
                      float* __restrict__ dwtape) {
    // SPLIT: LDS holds the fp32 image from W2 on (addressed through wl as before: wl[RT_W2C + x]), no fp32 W1
    constexpr int IMG0 = SPLIT ? RT_W2C : 0;
    constexpr int IMGF = ((RT_IMG_FLOATS - IMG0) + 3) & ~3;
    float* wl = rt_smem - IMG0;
    for (int e = IMG0 + threadIdx.x; e < RT_IMG_FLOATS; e += 256) wl[e] = wimg[e];
    float* lbase = rt_smem + IMGF;
    f32x4v* ex = reinterpret_cast<f32x4v*>(lbase);                    // the nets' parts of x̄: [3 nets][6 tiles][64 lanes]
    f32x4v* dOl = ex + 3 * 6 * 64;                                     // the helper's dO: [3 variables][2 tiles][64 lanes]
    float* stg_all = lbase + (3 * 6 * 64 + 3 * 2 * 64) * 4;           // record staging: see rt16s_adjoint_kernel
    for (int e = threadIdx.x; e < 3 * RT16S_STG; e += 256) stg_all[e] = 0.0f;
    const u32x4* hm = reinterpret_cast<const u32x4*>(stg_all + 3 * RT16S_STG);      // SPLIT: [net][kb][tile][h, m][64 lanes] operand planes
    if constexpr (SPLIT) {
        const u32x4* src = reinterpret_cast<const u32x4*>(wimg + RT_NSA_OFF);
        u32x4* dst = reinterpret_cast<u32x4*>(stg_all + 3 * RT16S_STG);
        for (int e = threadIdx.x; e < RT_NSA_HM_WORDS / 4; e += 256) dst[e] = src[e];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool helper = role == 3;
    const int n = helper ? 0 : role;                                    // the net of a net wave
    const int j = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x;
    const int col = tile * 16 + j;
    const bool valid = col < n_col;
    const int colc = min(col, n_col - 1);
    const int n_steps = (n_save - 1) * substeps;
    const int nst = RKC ? m.nst : 4;                                    // RHS evaluations (= tape records) per step: 4 (RK4) or s (RKC2)
    constexpr bool rkc = RKC;
    const float* tp = t16_tape + (size_t)tile * n_steps * nst * 1536 + j * 96 + 4 * g;
    const bool phys = m.mpp || m.ca;
    float* out = slab + (size_t)tile * (m.n_params + 8);

    if (helper) {
        // ================================================= the helper wave: λ, x̄, loss injection, physics pullback ======================
        const float* tzr = t16_ztape + (size_t)tile * n_steps * nst * RT16S_RREC + lane * 4;             // RICH: the rich tape
        V16 lam[3], xb[3], xbs[3];
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int tau = 0; tau < 2; tau++) { lam[q].t[tau] = (f32x4t)(0.0f); xb[q].t[tau] = (f32x4t)(0.0f); xbs[q].t[tau] = (f32x4t)(0.0f); }
        float sum_d[3] = {0.0f, 0.0f, 0.0f}, sum_g[3] = {0.0f, 0.0f, 0.0f};
        auto inject = [&](int sv, bool add) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 3; q++) {
                V16 d;
#pragma unroll
                for (int tau = 0; tau < 2; tau++) {
                    const size_t o = ((size_t)colc * n_save + sv) * 96 + q * 32 + 16 * tau + 4 * g;
                    const f32x4v a = *reinterpret_cast<const f32x4v*>(sol + o);
                    const f32x4v b = *reinterpret_cast<const f32x4v*>(truth + o);
#pragma unroll
                    for (int e = 0; e < 4; e++) d.t[tau][e] = valid ? a[e] - b[e] : 0.0f;
                }
                const V16 dd = shift_down16(d, lane, 0.0f);
                V16 gg;
#pragma unroll
                for (int tau = 0; tau < 2; tau++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        gg.t[tau][r] = (tau == 0 && r == 0 && g == 0) ? 0.0f : (d.t[tau][r] - dd.t[tau][r]) * 32.0f;
                        sum_d[q] += d.t[tau][r] * d.t[tau][r];
                        sum_g[q] += gg.t[tau][r] * gg.t[tau][r];
                    }
                if (add) {
                    const V16 gu_ = shift_up16(gg, lane, 0.0f);
#pragma unroll
                    for (int tau = 0; tau < 2; tau++)
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            lam[q].t[tau][r] += 2.0f * lw.w[q] * d.t[tau][r] + 2.0f * lw.w[3 + q] * 32.0f * (gg.t[tau][r] - gu_.t[tau][r]);
                }
            }
        };
        // what the pullback reads from the tapes, one stage ahead: RICH the nine coefficients, otherwise the stage input
        PhysC Pp;
        V16 Xp[3];
        auto prefetch = [&](int qs) __attribute__((always_inline)) {
            if (RICH) {
                if (phys) {
                    const float* sr = tzr + (size_t)qs * RT16S_RREC;
#pragma unroll
                    for (int tau = 0; tau < 2; tau++) {
                        Pp.dn0.t[tau] = *reinterpret_cast<const f32x4v*>(sr + (36 + 0 + tau) * 256);
                        Pp.dn1.t[tau] = *reinterpret_cast<const f32x4v*>(sr + (36 + 2 + tau) * 256);
                        Pp.dn2.t[tau] = *reinterpret_cast<const f32x4v*>(sr + (36 + 4 + tau) * 256);
                        Pp.nu0.t[tau] = *reinterpret_cast<const f32x4v*>(sr + (36 + 6 + tau) * 256);
                        Pp.nu1.t[tau] = *reinterpret_cast<const f32x4v*>(sr + (36 + 8 + tau) * 256);
                        Pp.nu2.t[tau] = *reinterpret_cast<const f32x4v*>(sr + (36 + 10 + tau) * 256);
                        Pp.c0.t[tau] = *reinterpret_cast<const f32x4v*>(sr + (36 + 12 + tau) * 256);
                        Pp.c1.t[tau] = *reinterpret_cast<const f32x4v*>(sr + (36 + 14 + tau) * 256);
                        Pp.c2.t[tau] = *reinterpret_cast<const f32x4v*>(sr + (36 + 16 + tau) * 256);
                    }
                }
            } else {
                const float* sx = tp + (size_t)qs * 1536;
#pragma unroll
                for (int q = 0; q < 3; q++)
#pragma unroll
                    for (int tau = 0; tau < 2; tau++) Xp[q].t[tau] = *reinterpret_cast<const f32x4v*>(sx + q * 32 + 16 * tau);
            }
        };
        // RKC2 (discrete adjoint of the recurrence, as in tile16's adjoint_kernel): cotangents of Y_{j-1} (yb1), Y_{j-2} (yb2), Y_0 (yb0), F_0 (f0b);
        // lam is the cotangent of Y_j.  The convective-adjustment switch is pulled back with ONE pattern per step, that of Y_{s-1}:
        // the physics data of every stage of a step are then read from the record of its last stage.
        V16 yb0[3], yb1[3], yb2[3], f0b[3];
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int tau = 0; tau < 2; tau++) { yb0[q].t[tau] = (f32x4t)(0.0f); yb1[q].t[tau] = (f32x4t)(0.0f); yb2[q].t[tau] = (f32x4t)(0.0f); f0b[q].t[tau] = (f32x4t)(0.0f); }
        const float* mu_t = m.rkc, *nu_t = m.rkc + RKC_LD, *mut_t = m.rkc + 2 * RKC_LD, *gat_t = m.rkc + 3 * RKC_LD, *kap_t = m.rkc + 5 * RKC_LD;
        const bool one_pattern = rkc && m.ca && !m.mpp;
#define PHYS_Q(q) (one_pattern ? ((q) / nst) * nst + nst - 1 : (q))
        inject(0, false);
        prefetch(PHYS_Q(n_steps * nst - 1));
        for (int iv = n_save - 2; iv >= 0; iv--) {
            const float dt = (save_times[iv + 1] - save_times[iv]) / (float)substeps;
            inject(iv + 1, true);
            for (int s = substeps - 1; s >= 0; s--) {
                const int step = iv * substeps + s;
#pragma nounroll
                for (int st = nst - 1; st >= 0; st--) {
                    const float cwl = (st == 0 || st == 3) ? dt / 6.0f : dt / 3.0f;
                    const float cwx = st == 3 ? 0.0f : (st == 2 ? dt : 0.5f * dt);
                    const int qs = step * nst + st;
                    V16 kb[3], xbp[3];
                    if (rkc) {
                        // stage input Y_st feeds Y_j, j = st + 1, through mu~_j h F_st
                        const float cmu = mu_t[st + 1], cnu = nu_t[st + 1], cmt = mut_t[st + 1] * dt, cga = gat_t[st + 1] * dt, ck0 = kap_t[st + 1];
#pragma unroll
                        for (int q = 0; q < 3; q++)
#pragma unroll
                            for (int tau = 0; tau < 2; tau++) {
                                if (st < nst - 1) {        // lam = cotangent of Y_j, complete once the previous pullback (xb: J(Y_j)^T F̄_j) is added
                                    lam[q].t[tau] = yb1[q].t[tau] + xb[q].t[tau];
                                    yb1[q].t[tau] = yb2[q].t[tau];
                                    yb2[q].t[tau] = (f32x4t)(0.0f);
                                }
                                if (st >= 1) {
                                    yb0[q].t[tau] += ck0 * lam[q].t[tau];
                                    yb1[q].t[tau] += cmu * lam[q].t[tau];
                                    yb2[q].t[tau] += cnu * lam[q].t[tau];
                                    f0b[q].t[tau] += cga * lam[q].t[tau];
                                    kb[q].t[tau] = cmt * lam[q].t[tau];
                                } else {                   // Y_1 = Y_0 + mu~_1 h F_0: lam holds Ȳ_1, yb1 the nu_2 part of Ȳ_0
                                    yb0[q].t[tau] += lam[q].t[tau] + yb1[q].t[tau];
                                    kb[q].t[tau] = f0b[q].t[tau] + cmt * lam[q].t[tau];
                                    yb1[q].t[tau] = (f32x4t)(0.0f);
                                    f0b[q].t[tau] = (f32x4t)(0.0f);
                                }
                            }
                    } else {
#pragma unroll
                    for (int q = 0; q < 3; q++)
#pragma unroll
                        for (int tau = 0; tau < 2; tau++) kb[q].t[tau] = cwl * lam[q].t[tau] + cwx * xb[q].t[tau];
                    }
                    if (RICH) {
                        rt16_physics_apply(m, Pp, kb, lane, xbp);                        // kb now holds dO
                        if (qs > 0) prefetch(PHYS_Q(qs - 1));
                    } else {
                        V16 X[3];
#pragma unroll
                        for (int q = 0; q < 3; q++) X[q] = Xp[q];
                        if (qs > 0) prefetch(PHYS_Q(qs - 1));
                        rt16_physics_vjp(m, X, kb, lane, xbp);
                    }
#pragma unroll
                    for (int q = 0; q < 3; q++)
#pragma unroll
                        for (int tau = 0; tau < 2; tau++) dOl[(q * 2 + tau) * 64 + lane] = kb[q].t[tau];
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // (B) dO is in LDS
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // (A) the nets' parts of x̄ are in LDS
#pragma unroll
                    for (int q = 0; q < 3; q++)
#pragma unroll
                        for (int tau = 0; tau < 2; tau++) {
                            const f32x4v c0 = ex[(0 * 6 + q * 2 + tau) * 64 + lane], c1 = ex[(1 * 6 + q * 2 + tau) * 64 + lane],
                                         c2 = ex[(2 * 6 + q * 2 + tau) * 64 + lane];
                            xb[q].t[tau] = xbp[q].t[tau] + ((c0 + c1) + c2);
                            xbs[q].t[tau] += xb[q].t[tau];
                        }
                }
                // RK4: λ_n = λ_{n+1} + x̄_1 + x̄_2 + x̄_3 + x̄_4;  RKC2: λ_n = Ȳ_0 + J(Y_0)^T F̄_0
#pragma unroll
                for (int q = 0; q < 3; q++)
#pragma unroll
                    for (int tau = 0; tau < 2; tau++) {
                        if (rkc) { lam[q].t[tau] = yb0[q].t[tau] + xb[q].t[tau]; yb0[q].t[tau] = (f32x4t)(0.0f); }
                        else lam[q].t[tau] += xbs[q].t[tau];
                        xbs[q].t[tau] = (f32x4t)(0.0f);
                    }
            }
        }
#pragma unroll
        for (int q = 0; q < 3; q++) {
            float sd = sum_d[q], sg = sum_g[q];
            for (int off = 32; off > 0; off >>= 1) { sd += __shfl_down(sd, off); sg += __shfl_down(sg, off); }
            if (lane == 0) { out[m.n_params + q] = sd; out[m.n_params + 3 + q] = sg; }
        }
        if (lane == 0) { out[m.n_params + 6] = 0.0f; out[m.n_params + 7] = 0.0f; }
        return;
    }

    // ===================================================== the three net waves ========================================================
    float* stg = stg_all + n * RT16S_STG;
    const int i_ = lane & 15, g_i = i_ >> 2, r_i = i_ & 3;
    int b3T[2], b2T[4];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int Q2 = 4 * u + r_i;
        b3T[u] = RT_W3C + (n * 31 + 4 * g - 1) * RT_LD3 + (Q2 < 5 ? 4 * Q2 + g_i : 20);
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int Q = 4 * t + r_i, f = 4 * Q + g_i;
        b2T[t] = RT_W2C + (n * 20 + g) * RT_LD2 + ((Q < 13 && f < 50) ? f : 50);
    }
    const int b1T = RT_W1C + (n * 50 + g) * RT_LD1 + i_;
    f32x4t gb1[4], gb2[2], gb3[2];                         // bias gradients: this lane's column of every delta, summed over the stages
#pragma unroll
    for (int t = 0; t < 4; t++) gb1[t] = (f32x4t)(0.0f);
#pragma unroll
    for (int u = 0; u < 2; u++) { gb2[u] = (f32x4t)(0.0f); gb3[u] = (f32x4t)(0.0f); }
    const float* tz = RICH ? t16_ztape + (size_t)tile * n_steps * nst * RT16S_RREC + lane * 4
                           : t16_ztape + (size_t)tile * n_steps * nst * RT16S_ZREC + j * 216 + n * 72 + g;
    float* rec0 = dwtape + (size_t)tile * n_steps * nst * RT16S_REC;
    int stg_rd[11], rec_wr[11];
#pragma unroll
    for (int i = 0; i < 11; i++) {
        const int e = 64 * i + lane, c = e / 44, f4 = e - 44 * c;
        stg_rd[i] = c * 180 + 4 * f4;
        rec_wr[i] = c * 720 + 96 + (f4 < 18 ? n * 104 + 4 * f4 : (3 + n) * 104 + 4 * (f4 - 18));
    }
    float* sw = stg + j * 180 + g;
    // this net's tape data, one stage ahead: variable n of the stage input, and the activations with their derivatives (RICH) or the
    // pre-activations they are evaluated from
    V16 Xnp;
    float z1p[13], z2p[5];
    f32x4t adp[12];
    auto prefetch = [&](int qs) __attribute__((always_inline)) {
        const float* sx = tp + (size_t)qs * 1536;
#pragma unroll
        for (int tau = 0; tau < 2; tau++) Xnp.t[tau] = *reinterpret_cast<const f32x4v*>(sx + n * 32 + 16 * tau);
        if (RICH) {
            const float* sr = tz + (size_t)qs * RT16S_RREC;
#pragma unroll
            for (int e = 0; e < 12; e++) adp[e] = *reinterpret_cast<const f32x4v*>(sr + (n * 12 + e) * 256);
        } else {
            const float* sz = tz + (size_t)qs * RT16S_ZREC;
#pragma unroll
            for (int Q = 0; Q < 13; Q++) z1p[Q] = (Q < 12 || g < 2) ? sz[4 * Q] : 0.0f;
#pragma unroll
            for (int Q2 = 0; Q2 < 5; Q2++) z2p[Q2] = sz[52 + 4 * Q2];
        }
    };
    prefetch(n_steps * nst - 1);
    RT_STAMP_DECL;
    for (int iv = n_save - 2; iv >= 0; iv--) {
        for (int s = substeps - 1; s >= 0; s--) {
            const int step = iv * substeps + s;
#pragma nounroll
            for (int st = nst - 1; st >= 0; st--) {
                const int qs = step * nst + st;
                RT_STAMP_BEGIN();
                // ---- before (B), beside the helper's pullback: everything that depends on the tapes alone ----
                f32x4t A1[4], D1[4], A2[2], D2[2];
                const V16 Xme = Xnp;
                if (RICH) {
#pragma unroll
                    for (int t = 0; t < 4; t++) { A1[t] = adp[t]; D1[t] = adp[4 + t]; }
#pragma unroll
                    for (int u = 0; u < 2; u++) { A2[u] = adp[8 + u]; D2[u] = adp[10 + u]; }
                } else {
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        f32x4t z;
#pragma unroll
                        for (int r = 0; r < 4; r++) z[r] = (4 * t + r < 13) ? z1p[(4 * t + r) < 13 ? 4 * t + r : 0] : 0.0f;
                        rt16_act_pair<ACT>(z, A1[t], D1[t]);
                    }
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        f32x4t z;
#pragma unroll
                        for (int r = 0; r < 4; r++) z[r] = (4 * u + r < 5) ? z2p[(4 * u + r) < 5 ? 4 * u + r : 0] : 0.0f;
                        rt16_act_pair<ACT>(z, A2[u], D2[u]);
                    }
#pragma unroll
                    for (int r = 1; r < 4; r++) { D1[3][r] = 0.0f; D2[1][r] = 0.0f; }   // padding quads: Q >= 13, Q2 >= 5
                    if (g >= 2) D1[3][0] = 0.0f;                                         // features 50, 51 of quad 12
                }
                if (qs > 0) prefetch(qs - 1);
                // (SPLIT) the twelve l-plane operands of this net's W1^T products, from L2: requested here, consumed behind barrier B and two chains
                // ... and the 24 h / m plane fragments from LDS: all of them before barrier B too, so that behind it the products wait for nothing
                // (fetched tile by tile in front of their products they exposed an LDS round trip per tile: 8.03 instead of 8.2 ms, no more)
                u32x4 Lr[12], Hm[24];
                if constexpr (SPLIT) {
                    int lz = lane;                        // (opaque: left loop-invariant the loads are hoisted out of the time loop and their registers pinned)
                    asm volatile("" : "+v"(lz));
                    const u32x4* lg = reinterpret_cast<const u32x4*>(wimg + RT_NSA_OFF + RT_NSA_L) + n * 12 * 64 + lz;
#pragma unroll
                    for (int u = 0; u < 12; u++) Lr[u] = lg[u * 64];
                    const u32x4* hn = hm + n * 24 * 64 + lz;
#pragma unroll
                    for (int u = 0; u < 24; u++) Hm[u] = hn[u * 64];
                }
                float a3[2][8], a2[4][5];
                rt16_fetch_ops<2, 8>(wl, a3, [&](int u, int k) { return b3T[u] + (16 * (k >> 2) + (k & 3)) * RT_LD3; });
                rt16_fetch_ops<4, 5>(wl, a2, [&](int t, int k) { return b2T[t] + 4 * k * RT_LD2; });
                float* rec = rec0 + (size_t)qs * RT16S_REC;
#pragma unroll
                for (int tau = 0; tau < 2; tau++) *reinterpret_cast<f32x4v*>(rec + j * 720 + n * 32 + 16 * tau + 4 * g) = Xme.t[tau];
#pragma unroll
                for (int Q = 0; Q < 13; Q++)
                    if (Q < 12 || g < 2) sw[4 * Q] = A1[Q >> 2][Q & 3];
#pragma unroll
                for (int Q2 = 0; Q2 < 5; Q2++) sw[52 + 4 * Q2] = A2[Q2 >> 2][Q2 & 3];
                RT_STAMP(0);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");           // (B) the helper's dO is in LDS
                RT_STAMP(1);
                V16 dO;
#pragma unroll
                for (int tau = 0; tau < 2; tau++) dO.t[tau] = dOl[(n * 2 + tau) * 64 + lane];
                // (3) δz2 = (W3^T dO) ∘ act'(z2)
                f32x4t dZ2[2] = {(f32x4t)(0.0f), (f32x4t)(0.0f)};
                rt16_run_ops<2, 8>(a3, dZ2, [&](int k) { return dO.t[k >> 2][k & 3]; });
#pragma unroll
                for (int u = 0; u < 2; u++) dZ2[u] *= D2[u];
                float a1[2][2][13];
                if constexpr (!SPLIT) rt16_fetch_ops<2, 13>(wl, a1[0], [&](int c, int k) { return b1T + 16 * c + 4 * k * RT_LD1; });
                RT_SCHED_HARD();
                // (4) δz1 = (W2^T δz2) ∘ act'(z1)
                f32x4t dZ1[4] = {(f32x4t)(0.0f), (f32x4t)(0.0f), (f32x4t)(0.0f), (f32x4t)(0.0f)};
                rt16_run_ops<4, 5>(a2, dZ1, [&](int k) { return dZ2[k >> 2][k & 3]; });
#pragma unroll
                for (int t = 0; t < 4; t++) dZ1[t] *= D1[t];
                RT_STAMP(2);
                {
#pragma unroll
                    for (int Q = 0; Q < 13; Q++)
                        if (Q < 12 || g < 2) sw[72 + 4 * Q] = dZ1[Q >> 2][Q & 3];
#pragma unroll
                    for (int Q2 = 0; Q2 < 5; Q2++) sw[124 + 4 * Q2] = dZ2[Q2 >> 2][Q2 & 3];
                    float* s3 = stg + j * 180 + 144 + 4 * g - 1;                         // output o = face - 1
#pragma unroll
                    for (int v = 0; v < 2; v++)
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (!(v == 0 && r == 0 && g == 0)) s3[16 * v + r] = dO.t[v][r];
                    // (deferring this copy to the top of the next stage, beside the helper's pullback, was measured: 7.58 -> 7.63 ms in front of that
                    //  stage's loads, 7.89 ms behind them — the stores then sit in front of the l-plane loads in the in-order vmcnt queue)
#pragma unroll
                    for (int i = 0; i < 11; i++) *reinterpret_cast<f32x4v*>(rec + rec_wr[i]) = *reinterpret_cast<const f32x4v*>(stg + stg_rd[i]);
                }
#pragma unroll
                for (int t = 0; t < 4; t++) gb1[t] += dZ1[t];
#pragma unroll
                for (int u = 0; u < 2; u++) { gb2[u] += dZ2[u]; gb3[u] += dO.t[u]; }
                RT_STAMP(3);
                // (5) this net's part of the state cotangent, W1_n^T δz1 (6 tiles x 13 k-steps)
                if constexpr (SPLIT) {
                    // δz1 as the B operand of two 32-deep k-blocks: element e of k-block kb is quad 8 kb + e of this lane group (padding quads are zero)
                    const float d80[8] = {dZ1[0][0], dZ1[0][1], dZ1[0][2], dZ1[0][3], dZ1[1][0], dZ1[1][1], dZ1[1][2], dZ1[1][3]};
                    const float d81[8] = {dZ1[2][0], dZ1[2][1], dZ1[2][2], dZ1[2][3], dZ1[3][0], dZ1[3][1], dZ1[3][2], dZ1[3][3]};
                    // the first k-block over all six tiles, the second k-block's split issued in pieces between them (one wave per SIMD: nothing else hides
                    // its 44 vector instructions), then the second k-block
                    const Bf3 B0 = bf3_split8(d80);
                    Bf3 B1 = B0;
                    f32x4t c2[6];
#pragma unroll
                    for (int tl = 0; tl < 6; tl++) {
                        Bf3 A0;
                        A0.h = Hm[tl * 2 + 0]; A0.m = Hm[tl * 2 + 1]; A0.l = Lr[tl];
                        c2[tl] = mfma16_bf3(A0, B0, (f32x4t)(0.0f));
                        RT_SCHED_FENCE();
                        if (tl < 4) {
                            bf3_split_pair(d81[2 * tl], d81[2 * tl + 1], tl, B1);
                            RT_SCHED_FENCE();
                        }
                    }
#pragma unroll
                    for (int tl = 0; tl < 6; tl++) {
                        Bf3 A1_;
                        A1_.h = Hm[(6 + tl) * 2 + 0]; A1_.m = Hm[(6 + tl) * 2 + 1]; A1_.l = Lr[6 + tl];
                        c2[tl] = mfma16_bf3(A1_, B1, c2[tl]);
                        ex[(n * 6 + tl) * 64 + lane] = c2[tl];
                        RT_SCHED_FENCE();
                    }
                } else
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    if (q < 2) rt16_fetch_ops<2, 13>(wl, a1[(q + 1) & 1], [&](int c, int k) { return b1T + 32 * (q + 1) + 16 * c + 4 * k * RT_LD1; });
                    RT_SCHED_HARD();
                    f32x4t c2[2] = {(f32x4t)(0.0f), (f32x4t)(0.0f)};
                    rt16_run_ops<2, 13>(a1[q & 1], c2, [&](int k) { return dZ1[k >> 2][k & 3]; });
#pragma unroll
                    for (int tau = 0; tau < 2; tau++) ex[(n * 6 + q * 2 + tau) * 64 + lane] = c2[tau];
                    RT_SCHED_HARD();
                }
                RT_STAMP(4);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");           // (A) this net's part of x̄ is in LDS
                RT_STAMP(5);
            }
        }
    }
#ifndef COLNDE_STAMPS_FWD
    RT_STAMP_FLUSH();
#endif
    // ---- the tile's slab row: bias gradients of net n (sums over the 16 columns = the lanes of a g-group); the loss sums are the helper's ----
    auto colsum = [&](float v) {
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
        return v;
    };
#pragma unroll
    for (int Q = 0; Q < 13; Q++) {
        const float v = colsum(gb1[Q >> 2][Q & 3]);
        if (j == 0 && (Q < 12 || g < 2)) out[n * m.net_size + m.b_off[0] + 4 * Q + g] = v;
    }
#pragma unroll
    for (int Q2 = 0; Q2 < 5; Q2++) {
        const float v = colsum(gb2[Q2 >> 2][Q2 & 3]);
        if (j == 0) out[n * m.net_size + m.b_off[1] + 4 * Q2 + g] = v;
    }
#pragma unroll
    for (int v_ = 0; v_ < 2; v_++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float v = colsum(gb3[v_][r]);
            const int face = 16 * v_ + 4 * g + r;
            if (j == 0 && face >= 1) out[n * m.net_size + m.b_off[2] + face - 1] = v;
        }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool rt_supported(const DevModel& m) {
    return m.model == COLNDE_MODEL_WIND_MIXING && m.Nz == 32 && m.n_layers == 3 && m.sizes[0] == 96 && m.sizes[1] == 50 &&
           m.sizes[2] == 20 && m.sizes[3] == 31 && m.acts[0] == m.acts[1] && m.acts[2] == COLNDE_ACT_IDENTITY &&
           !m.smooth_NN && !m.smooth_Ri && !m.inplace;
}

size_t rt_forward_lds_bytes() { return (size_t)RT_IMG_FLOATS * sizeof(float); }
size_t rt_adjoint_lds_bytes() { return ((size_t)((RT_IMG_FLOATS + 3) & ~3) + RT_WAVES * (3072 + RT_TB)) * sizeof(float); }
size_t rt_tape_floats(int n_col, int n_steps) { return (size_t)((n_col + RT_COLS - 1) / RT_COLS) * n_steps * 4 * 3072; }
size_t rt_tape2_floats(int n_col, int n_steps) { return (size_t)((n_col + RT_COLS - 1) / RT_COLS) * n_steps * 4 * RT_TAPE2; }
size_t rt_split_rich_record_floats() { return RT16S_RREC; }
size_t rt_tapez_floats(int n_col, int n_steps) { return (size_t)((n_col + RT_COLS - 1) / RT_COLS) * n_steps * 4 * RT_TAPEZ; }
int rt_n_wtiles(int n_col) { return (n_col + RT_COLS - 1) / RT_COLS; }
int rt_dw1_waves(int n_col, int n_steps) {
    const long items = (long)rt_n_wtiles(n_col) * n_steps * 4;
    const long w = items < 1024 ? items : 1024;
    return (int)((w + RT_WAVES - 1) / RT_WAVES) * RT_WAVES;
}

hipError_t rt_set_attributes() {
    hipError_t e;
    const int v = 160 * 1024;
#define RT_SETATTR(K) if ((e = hipFuncSetAttribute((const void*)(K), hipFuncAttributeMaxDynamicSharedMemorySize, v)) != hipSuccess) return e
    RT_SETATTR(rt_forward_kernel<COLNDE_ACT_IDENTITY>);
    RT_SETATTR(rt_forward_kernel<COLNDE_ACT_RELU>);
    RT_SETATTR(rt_forward_kernel<COLNDE_ACT_MISH>);
    RT_SETATTR(rt_forward_kernel<COLNDE_ACT_SWISH>);
    RT_SETATTR(rt_forward_kernel<COLNDE_ACT_TANH>);
    RT_SETATTR(rt_forward_kernel<COLNDE_ACT_LEAKYRELU>);
    RT_SETATTR((rt16_forward_kernel<COLNDE_ACT_IDENTITY, false>));
    RT_SETATTR((rt16_forward_kernel<COLNDE_ACT_IDENTITY, true>));
    RT_SETATTR((rt16_forward_kernel<COLNDE_ACT_RELU, false>));
    RT_SETATTR((rt16_forward_kernel<COLNDE_ACT_RELU, true>));
    RT_SETATTR((rt16_forward_kernel<COLNDE_ACT_MISH, false>));
    RT_SETATTR((rt16_forward_kernel<COLNDE_ACT_MISH, true>));
    RT_SETATTR((rt16_forward_kernel<COLNDE_ACT_SWISH, false>));
    RT_SETATTR((rt16_forward_kernel<COLNDE_ACT_SWISH, true>));
    RT_SETATTR((rt16_forward_kernel<COLNDE_ACT_TANH, false>));
    RT_SETATTR((rt16_forward_kernel<COLNDE_ACT_TANH, true>));
    RT_SETATTR((rt16_forward_kernel<COLNDE_ACT_LEAKYRELU, false>));
    RT_SETATTR((rt16_forward_kernel<COLNDE_ACT_LEAKYRELU, true>));
    RT_SETATTR((rt16s_forward_kernel<COLNDE_ACT_IDENTITY, false>));
    RT_SETATTR((rt16s_forward_kernel<COLNDE_ACT_IDENTITY, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_IDENTITY, false>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_IDENTITY, false, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_IDENTITY, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_IDENTITY, true, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_IDENTITY, false, false, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_IDENTITY, true, false, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_IDENTITY, false, true, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_IDENTITY, true, true, true>));
    RT_SETATTR((rt16s_forward_kernel<COLNDE_ACT_RELU, false>));
    RT_SETATTR((rt16s_forward_kernel<COLNDE_ACT_RELU, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_RELU, false>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_RELU, false, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_RELU, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_RELU, true, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_RELU, false, false, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_RELU, true, false, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_RELU, false, true, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_RELU, true, true, true>));
    RT_SETATTR((rt16s_forward_kernel<COLNDE_ACT_MISH, false>));
    RT_SETATTR((rt16s_forward_kernel<COLNDE_ACT_MISH, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_MISH, false>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_MISH, false, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_MISH, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_MISH, true, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_MISH, false, false, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_MISH, true, false, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_MISH, false, true, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_MISH, true, true, true>));
    RT_SETATTR((rt16s_forward_kernel<COLNDE_ACT_SWISH, false>));
    RT_SETATTR((rt16s_forward_kernel<COLNDE_ACT_SWISH, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_SWISH, false>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_SWISH, false, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_SWISH, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_SWISH, true, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_SWISH, false, false, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_SWISH, true, false, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_SWISH, false, true, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_SWISH, true, true, true>));
    RT_SETATTR((rt16s_forward_kernel<COLNDE_ACT_TANH, false>));
    RT_SETATTR((rt16s_forward_kernel<COLNDE_ACT_TANH, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_TANH, false>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_TANH, false, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_TANH, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_TANH, true, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_TANH, false, false, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_TANH, true, false, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_TANH, false, true, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_TANH, true, true, true>));
    RT_SETATTR((rt16s_forward_kernel<COLNDE_ACT_LEAKYRELU, false>));
    RT_SETATTR((rt16s_forward_kernel<COLNDE_ACT_LEAKYRELU, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_LEAKYRELU, false>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_LEAKYRELU, false, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_LEAKYRELU, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_LEAKYRELU, true, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_LEAKYRELU, false, false, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_LEAKYRELU, true, false, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_LEAKYRELU, false, true, true>));
    RT_SETATTR((rt16sh_forward_kernel<COLNDE_ACT_LEAKYRELU, true, true, true>));
    RT_SETATTR(rt_dw1_kernel);
    RT_SETATTR(rt_dw1_split_kernel);
    RT_SETATTR((rt16s_adjoint_kernel<COLNDE_ACT_IDENTITY, false>));
    RT_SETATTR((rt16s_adjoint_kernel<COLNDE_ACT_IDENTITY, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_IDENTITY, false>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_IDENTITY, false, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_IDENTITY, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_IDENTITY, true, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_IDENTITY, false, false, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_IDENTITY, true, false, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_IDENTITY, false, true, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_IDENTITY, true, true, true>));
    RT_SETATTR((rt16s_adjoint_kernel<COLNDE_ACT_RELU, false>));
    RT_SETATTR((rt16s_adjoint_kernel<COLNDE_ACT_RELU, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_RELU, false>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_RELU, false, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_RELU, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_RELU, true, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_RELU, false, false, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_RELU, true, false, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_RELU, false, true, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_RELU, true, true, true>));
    RT_SETATTR((rt16s_adjoint_kernel<COLNDE_ACT_MISH, false>));
    RT_SETATTR((rt16s_adjoint_kernel<COLNDE_ACT_MISH, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_MISH, false>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_MISH, false, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_MISH, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_MISH, true, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_MISH, false, false, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_MISH, true, false, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_MISH, false, true, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_MISH, true, true, true>));
    RT_SETATTR((rt16s_adjoint_kernel<COLNDE_ACT_SWISH, false>));
    RT_SETATTR((rt16s_adjoint_kernel<COLNDE_ACT_SWISH, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_SWISH, false>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_SWISH, false, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_SWISH, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_SWISH, true, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_SWISH, false, false, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_SWISH, true, false, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_SWISH, false, true, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_SWISH, true, true, true>));
    RT_SETATTR((rt16s_adjoint_kernel<COLNDE_ACT_TANH, false>));
    RT_SETATTR((rt16s_adjoint_kernel<COLNDE_ACT_TANH, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_TANH, false>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_TANH, false, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_TANH, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_TANH, true, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_TANH, false, false, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_TANH, true, false, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_TANH, false, true, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_TANH, true, true, true>));
    RT_SETATTR((rt16s_adjoint_kernel<COLNDE_ACT_LEAKYRELU, false>));
    RT_SETATTR((rt16s_adjoint_kernel<COLNDE_ACT_LEAKYRELU, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_LEAKYRELU, false>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_LEAKYRELU, false, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_LEAKYRELU, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_LEAKYRELU, true, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_LEAKYRELU, false, false, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_LEAKYRELU, true, false, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_LEAKYRELU, false, true, true>));
    RT_SETATTR((rt16sh_adjoint_kernel<COLNDE_ACT_LEAKYRELU, true, true, true>));
    RT_SETATTR((rt_adjoint_kernel<COLNDE_ACT_IDENTITY, false>));
    RT_SETATTR((rt_adjoint_kernel<COLNDE_ACT_RELU, false>));
    RT_SETATTR((rt_adjoint_kernel<COLNDE_ACT_MISH, false>));
    RT_SETATTR((rt_adjoint_kernel<COLNDE_ACT_SWISH, false>));
    RT_SETATTR((rt_adjoint_kernel<COLNDE_ACT_TANH, false>));
    RT_SETATTR((rt_adjoint_kernel<COLNDE_ACT_LEAKYRELU, false>));
    RT_SETATTR((rt_adjoint_kernel<COLNDE_ACT_IDENTITY, true>));
    RT_SETATTR((rt_adjoint_kernel<COLNDE_ACT_IDENTITY, true, true>));
    RT_SETATTR((rt_adjoint_kernel<COLNDE_ACT_RELU, true>));
    RT_SETATTR((rt_adjoint_kernel<COLNDE_ACT_RELU, true, true>));
    RT_SETATTR((rt_adjoint_kernel<COLNDE_ACT_MISH, true>));
    RT_SETATTR((rt_adjoint_kernel<COLNDE_ACT_MISH, true, true>));
    RT_SETATTR((rt_adjoint_kernel<COLNDE_ACT_SWISH, true>));
    RT_SETATTR((rt_adjoint_kernel<COLNDE_ACT_SWISH, true, true>));
    RT_SETATTR((rt_adjoint_kernel<COLNDE_ACT_TANH, true>));
    RT_SETATTR((rt_adjoint_kernel<COLNDE_ACT_TANH, true, true>));
    RT_SETATTR((rt_adjoint_kernel<COLNDE_ACT_LEAKYRELU, true>));
    RT_SETATTR((rt_adjoint_kernel<COLNDE_ACT_LEAKYRELU, true, true>));
#undef RT_SETATTR
    return hipSuccess;
}

hipError_t rt_launch_pack(const DevModel& m, const float* w, float* wimg, hipStream_t stream) {
    hipLaunchKernelGGL(rt_pack_kernel, dim3((RT_IMG_FLOATS + 255) / 256), dim3(256), 0, stream, m, w, wimg);
    return hipGetLastError();
}

// COLNDE_RT_FWD=32 selects the one-wave-per-SIMD 32-column forward kernel (A/B aid; it does not tape Z1)
bool rt_forward_is32() {
    const char* e = getenv("COLNDE_RT_FWD");
    return e && atoi(e) == 32;
}

hipError_t rt_launch_forward(const DevModel& m, const float* wimg, const float* x0, const float* bcs,
                             const float* save_times, int n_save, int substeps, float* sol, float* tape, float* tapez,
                             int n_col, bool fwd32, bool split, hipStream_t stream) {
    const size_t lds = rt_forward_lds_bytes();
    // COLNDE_RT_FWD=32 selects the one-wave-per-SIMD 32-column kernel (A/B aid); default: 16-column tiles, two waves per SIMD
    if (fwd32) {
        const int n_wtiles = (n_col + RT_COLS - 1) / RT_COLS;
        const dim3 grid((n_wtiles + RT_WAVES - 1) / RT_WAVES), block(64 * RT_WAVES);
#define RT_FWD(A) hipLaunchKernelGGL(rt_forward_kernel<A>, grid, block, lds, stream, m, wimg, x0, bcs, save_times, n_save, substeps, sol, tape, n_col)
        switch (m.acts[0]) {
            case COLNDE_ACT_IDENTITY: RT_FWD(COLNDE_ACT_IDENTITY); break;
            case COLNDE_ACT_RELU: RT_FWD(COLNDE_ACT_RELU); break;
            case COLNDE_ACT_MISH: RT_FWD(COLNDE_ACT_MISH); break;
            case COLNDE_ACT_SWISH: RT_FWD(COLNDE_ACT_SWISH); break;
            case COLNDE_ACT_TANH: RT_FWD(COLNDE_ACT_TANH); break;
            case COLNDE_ACT_LEAKYRELU: RT_FWD(COLNDE_ACT_LEAKYRELU); break;
            default: return hipErrorInvalidValue;
        }
#undef RT_FWD
    } else {
        const int n_wt16 = 2 * ((n_col + RT_COLS - 1) / RT_COLS);
        // small problems spread over the CUs first (one wavefront per workgroup up to 256 tiles), then fill the SIMDs
        int wpw = (n_wt16 + 255) / 256;
        wpw = wpw < 1 ? 1 : (wpw > RT16_WAVES ? RT16_WAVES : wpw);
        const dim3 grid((n_wt16 + wpw - 1) / wpw), block(64 * wpw);
        // split: the nets on the bf16 pipe with exact three-way operand splitting (COLNDE_MATRIX_BF16X3_EXACT)
        const size_t lds_split = ((size_t)RT_SIMG_WORDS + (RT_IMG_FLOATS - RT_B1C)) * sizeof(float);
        if (split) hipLaunchKernelGGL(rt_pack_split_kernel, dim3(48), dim3(256), 0, stream, wimg, reinterpret_cast<unsigned*>(const_cast<float*>(wimg)) + RT_SIMG_OFF);
#define RT_FWD(A) do { if (split) hipLaunchKernelGGL((rt16_forward_kernel<A, true>), grid, block, lds_split, stream, m, wimg, x0, bcs, save_times, n_save, substeps, sol, tape, tapez, n_col); \
                       else hipLaunchKernelGGL((rt16_forward_kernel<A, false>), grid, block, lds, stream, m, wimg, x0, bcs, save_times, n_save, substeps, sol, tape, tapez, n_col); } while (0)
        switch (m.acts[0]) {
            case COLNDE_ACT_IDENTITY: RT_FWD(COLNDE_ACT_IDENTITY); break;
            case COLNDE_ACT_RELU: RT_FWD(COLNDE_ACT_RELU); break;
            case COLNDE_ACT_MISH: RT_FWD(COLNDE_ACT_MISH); break;
            case COLNDE_ACT_SWISH: RT_FWD(COLNDE_ACT_SWISH); break;
            case COLNDE_ACT_TANH: RT_FWD(COLNDE_ACT_TANH); break;
            case COLNDE_ACT_LEAKYRELU: RT_FWD(COLNDE_ACT_LEAKYRELU); break;
            default: return hipErrorInvalidValue;
        }
#undef RT_FWD
    }
    return hipGetLastError();
}

// the three-wavefronts-per-tile forward solve of the latency points; tapes (optional) in tile16's formats
hipError_t rt_launch_forward_split(const DevModel& m, const float* wimg, const float* x0, const float* bcs, const float* save_times,
                                   int n_save, int substeps, float* sol, float* t16_tape, float* t16_ztape, int n_col, bool rich, bool use_helper, bool want_split, hipStream_t stream) {
    const size_t lds = (((size_t)RT_IMG_FLOATS + 3) & ~(size_t)3) * sizeof(float) + 2 * 384 * 16;
    const dim3 grid((n_col + 15) / 16), block(192);
    const size_t ldsh = lds + 384 * 16;
    const dim3 blockh(256);
    if (m.nst != 4 && !(m.rkc && use_helper)) return hipErrorInvalidValue;      // RKC2 lives in the four-wave kernels only
    // layers 1 and 2 on the bf16 pipe with exact three-way operand splitting (COLNDE_MATRIX_BF16X3_EXACT; the four-wave kernels, RK4 and RKC2)
    const bool split = want_split && use_helper;
    const size_t ldss = ((size_t)RT_SIMG2_WORDS + ((RT_IMG_FLOATS - RT_W3C + 3) & ~3)) * sizeof(float) + 3 * 384 * 16;
    if (split) hipLaunchKernelGGL(rt_pack_split_ns_kernel, dim3(41), dim3(256), 0, stream, wimg, reinterpret_cast<unsigned*>(const_cast<float*>(wimg)) + RT_SIMG2_OFF);
#define RT_FWDS(A)                                                                                                                              \
    do {                                                                                                                                        \
        if (split && rich && m.rkc) hipLaunchKernelGGL((rt16sh_forward_kernel<A, true, true, true>), grid, blockh, ldss, stream, m, wimg, x0, bcs, save_times, n_save, substeps, sol, t16_tape, t16_ztape, n_col); \
        else if (split && m.rkc) hipLaunchKernelGGL((rt16sh_forward_kernel<A, false, true, true>), grid, blockh, ldss, stream, m, wimg, x0, bcs, save_times, n_save, substeps, sol, t16_tape, t16_ztape, n_col); \
        else if (split && rich) hipLaunchKernelGGL((rt16sh_forward_kernel<A, true, false, true>), grid, blockh, ldss, stream, m, wimg, x0, bcs, save_times, n_save, substeps, sol, t16_tape, t16_ztape, n_col); \
        else if (split) hipLaunchKernelGGL((rt16sh_forward_kernel<A, false, false, true>), grid, blockh, ldss, stream, m, wimg, x0, bcs, save_times, n_save, substeps, sol, t16_tape, t16_ztape, n_col); \
        else if (use_helper && rich && m.rkc) hipLaunchKernelGGL((rt16sh_forward_kernel<A, true, true>), grid, blockh, ldsh, stream, m, wimg, x0, bcs, save_times, n_save, substeps, sol, t16_tape, t16_ztape, n_col); \
        else if (use_helper && m.rkc) hipLaunchKernelGGL((rt16sh_forward_kernel<A, false, true>), grid, blockh, ldsh, stream, m, wimg, x0, bcs, save_times, n_save, substeps, sol, t16_tape, t16_ztape, n_col); \
        else if (use_helper && rich) hipLaunchKernelGGL((rt16sh_forward_kernel<A, true>), grid, blockh, ldsh, stream, m, wimg, x0, bcs, save_times, n_save, substeps, sol, t16_tape, t16_ztape, n_col); \
        else if (use_helper) hipLaunchKernelGGL((rt16sh_forward_kernel<A, false>), grid, blockh, ldsh, stream, m, wimg, x0, bcs, save_times, n_save, substeps, sol, t16_tape, t16_ztape, n_col); \
        else if (rich) hipLaunchKernelGGL((rt16s_forward_kernel<A, true>), grid, block, lds, stream, m, wimg, x0, bcs, save_times, n_save, substeps, sol, t16_tape, t16_ztape, n_col); \
        else hipLaunchKernelGGL((rt16s_forward_kernel<A, false>), grid, block, lds, stream, m, wimg, x0, bcs, save_times, n_save, substeps, sol, t16_tape, t16_ztape, n_col); \
    } while (0)
    switch (m.acts[0]) {
        case COLNDE_ACT_IDENTITY: RT_FWDS(COLNDE_ACT_IDENTITY); break;
        case COLNDE_ACT_RELU: RT_FWDS(COLNDE_ACT_RELU); break;
        case COLNDE_ACT_MISH: RT_FWDS(COLNDE_ACT_MISH); break;
        case COLNDE_ACT_SWISH: RT_FWDS(COLNDE_ACT_SWISH); break;
        case COLNDE_ACT_TANH: RT_FWDS(COLNDE_ACT_TANH); break;
        case COLNDE_ACT_LEAKYRELU: RT_FWDS(COLNDE_ACT_LEAKYRELU); break;
        default: return hipErrorInvalidValue;
    }
#undef RT_FWDS
    return hipGetLastError();
}

bool rt_adjoint_split_has_bf16(const DevModel& m, bool use_helper) { return use_helper && (m.rkc || m.nst == 4); }

hipError_t rt_launch_adjoint_split(const DevModel& m, const float* wimg, const float* save_times, int n_save, int substeps, const float* sol,
                                   const float* truth, const float* t16_tape, const float* t16_ztape, const LossWeights& lw, float* slab,
                                   int n_col, float* dwtape, bool rich, bool use_helper, bool want_split, hipStream_t stream) {
    // the record formats this kernel reads and writes are tile16's for exactly this shape
    if (!rt_supported(m) || dwtape_row_floats(m) * CT != RT16S_REC || t16_ztape_col_floats(m) * CT != RT16S_ZREC || (m.nst != 4 && !(m.rkc && use_helper)))
        return hipErrorInvalidValue;
    const size_t lds = (((size_t)RT_IMG_FLOATS + 3) & ~(size_t)3) * sizeof(float) + 2 * (3 * 6 * 64) * 16 + 3 * RT16S_STG * sizeof(float);
    const dim3 grid((n_col + 15) / 16), block(192);
    const size_t ldsh = (((size_t)RT_IMG_FLOATS + 3) & ~(size_t)3) * sizeof(float) + (3 * 6 * 64 + 3 * 2 * 64) * 16 + 3 * RT16S_STG * sizeof(float);
    const dim3 blockh(256);
    // COLNDE_MATRIX_BF16X3_EXACT: the W1^T products on the bf16 pipe (four-wave kernels, RK4 and RKC2); LDS: the fp32 image from W2 on, exchange, staging, the h / m planes
    const bool split = want_split && rt_adjoint_split_has_bf16(m, use_helper);
    const size_t ldss = ((((size_t)RT_IMG_FLOATS - RT_W2C) + 3) & ~(size_t)3) * sizeof(float) + (3 * 6 * 64 + 3 * 2 * 64) * 16 + 3 * RT16S_STG * sizeof(float) +
                        (size_t)RT_NSA_HM_WORDS * 4;
    if (split) hipLaunchKernelGGL(rt_pack_split_nsadj_kernel, dim3(36), dim3(256), 0, stream, wimg, reinterpret_cast<unsigned*>(const_cast<float*>(wimg)) + RT_NSA_OFF);
#define RT_ADJS(A)                                                                                                                              \
    do {                                                                                                                                        \
        if (split && rich && m.rkc) hipLaunchKernelGGL((rt16sh_adjoint_kernel<A, true, true, true>), grid, blockh, ldss, stream, m, wimg, save_times, n_save, substeps, sol, truth, t16_tape, t16_ztape, lw, slab, n_col, dwtape); \
        else if (split && m.rkc) hipLaunchKernelGGL((rt16sh_adjoint_kernel<A, false, true, true>), grid, blockh, ldss, stream, m, wimg, save_times, n_save, substeps, sol, truth, t16_tape, t16_ztape, lw, slab, n_col, dwtape); \
        else if (split && rich) hipLaunchKernelGGL((rt16sh_adjoint_kernel<A, true, false, true>), grid, blockh, ldss, stream, m, wimg, save_times, n_save, substeps, sol, truth, t16_tape, t16_ztape, lw, slab, n_col, dwtape); \
        else if (split) hipLaunchKernelGGL((rt16sh_adjoint_kernel<A, false, false, true>), grid, blockh, ldss, stream, m, wimg, save_times, n_save, substeps, sol, truth, t16_tape, t16_ztape, lw, slab, n_col, dwtape); \
        else if (use_helper && rich && m.rkc) hipLaunchKernelGGL((rt16sh_adjoint_kernel<A, true, true>), grid, blockh, ldsh, stream, m, wimg, save_times, n_save, substeps, sol, truth, t16_tape, t16_ztape, lw, slab, n_col, dwtape); \
        else if (use_helper && m.rkc) hipLaunchKernelGGL((rt16sh_adjoint_kernel<A, false, true>), grid, blockh, ldsh, stream, m, wimg, save_times, n_save, substeps, sol, truth, t16_tape, t16_ztape, lw, slab, n_col, dwtape); \
        else if (use_helper && rich) hipLaunchKernelGGL((rt16sh_adjoint_kernel<A, true>), grid, blockh, ldsh, stream, m, wimg, save_times, n_save, substeps, sol, truth, t16_tape, t16_ztape, lw, slab, n_col, dwtape); \
        else if (use_helper) hipLaunchKernelGGL((rt16sh_adjoint_kernel<A, false>), grid, blockh, ldsh, stream, m, wimg, save_times, n_save, substeps, sol, truth, t16_tape, t16_ztape, lw, slab, n_col, dwtape); \
        else if (rich) hipLaunchKernelGGL((rt16s_adjoint_kernel<A, true>), grid, block, lds, stream, m, wimg, save_times, n_save, substeps, sol, truth, t16_tape, t16_ztape, lw, slab, n_col, dwtape); \
        else hipLaunchKernelGGL((rt16s_adjoint_kernel<A, false>), grid, block, lds, stream, m, wimg, save_times, n_save, substeps, sol, truth, t16_tape, t16_ztape, lw, slab, n_col, dwtape); \
    } while (0)
    switch (m.acts[0]) {
        case COLNDE_ACT_IDENTITY: RT_ADJS(COLNDE_ACT_IDENTITY); break;
        case COLNDE_ACT_RELU: RT_ADJS(COLNDE_ACT_RELU); break;
        case COLNDE_ACT_MISH: RT_ADJS(COLNDE_ACT_MISH); break;
        case COLNDE_ACT_SWISH: RT_ADJS(COLNDE_ACT_SWISH); break;
        case COLNDE_ACT_TANH: RT_ADJS(COLNDE_ACT_TANH); break;
        case COLNDE_ACT_LEAKYRELU: RT_ADJS(COLNDE_ACT_LEAKYRELU); break;
        default: return hipErrorInvalidValue;
    }
#undef RT_ADJS
    return hipGetLastError();
}

hipError_t rt_launch_adjoint(const DevModel& m, const float* wimg, const float* bcs, const float* save_times, int n_save,
                             int substeps, const float* sol, const float* truth, const float* tape, float* tape2,
                             const float* tapez, const LossWeights& lw, float* slab, int n_col, bool want_split, hipStream_t stream) {
    const int n_wtiles = rt_n_wtiles(n_col);
    const dim3 grid((n_wtiles + RT_WAVES - 1) / RT_WAVES), block(64 * RT_WAVES);
    const size_t lds = rt_adjoint_lds_bytes();
    // with the Z1 tape: the W1^T products on the bf16 pipe with exact three-way operand splitting (COLNDE_MATRIX_BF16X3_EXACT)
    const bool split = want_split && tapez;
    if (split) hipLaunchKernelGGL(rt_pack_split_adj_kernel, dim3(30), dim3(256), 0, stream, wimg, reinterpret_cast<unsigned*>(const_cast<float*>(wimg)) + RT_ASIMG_OFF);
#define RT_ADJ(A)                                                                                                             \
    do {                                                                                                                      \
        if (split) hipLaunchKernelGGL((rt_adjoint_kernel<A, true, true>), grid, block, lds, stream, m, wimg, bcs, save_times, n_save, \
                                      substeps, sol, truth, tape, tape2, tapez, lw, slab, n_col);                      \
        else if (tapez) hipLaunchKernelGGL((rt_adjoint_kernel<A, true>), grid, block, lds, stream, m, wimg, bcs, save_times, n_save, \
                                      substeps, sol, truth, tape, tape2, tapez, lw, slab, n_col);                      \
        else hipLaunchKernelGGL((rt_adjoint_kernel<A, false>), grid, block, lds, stream, m, wimg, bcs, save_times, n_save,    \
                                substeps, sol, truth, tape, tape2, tapez, lw, slab, n_col);                            \
    } while (0)
    switch (m.acts[0]) {
        case COLNDE_ACT_IDENTITY: RT_ADJ(COLNDE_ACT_IDENTITY); break;
        case COLNDE_ACT_RELU: RT_ADJ(COLNDE_ACT_RELU); break;
        case COLNDE_ACT_MISH: RT_ADJ(COLNDE_ACT_MISH); break;
        case COLNDE_ACT_SWISH: RT_ADJ(COLNDE_ACT_SWISH); break;
        case COLNDE_ACT_TANH: RT_ADJ(COLNDE_ACT_TANH); break;
        case COLNDE_ACT_LEAKYRELU: RT_ADJ(COLNDE_ACT_LEAKYRELU); break;
        default: return hipErrorInvalidValue;
    }
#undef RT_ADJ
    return hipGetLastError();
}

hipError_t rt_launch_dw1(const DevModel& m, const float* tape, const float* tape2, int n_col, int n_steps, float* slab_rows,
                         bool split, hipStream_t stream) {
    const long items = (long)rt_n_wtiles(n_col) * n_steps * 4;
    const int waves = rt_dw1_waves(n_col, n_steps);
    if (split)      // dW1 on the bf16 pipe with exact three-way operand splitting (COLNDE_MATRIX_BF16X3_EXACT)
        hipLaunchKernelGGL(rt_dw1_split_kernel, dim3(waves / RT_WAVES), dim3(64 * RT_WAVES), RT_WAVES * RT_DW1_LDS * sizeof(float), stream, m,
                           tape, tape2, items, slab_rows);
    else
        hipLaunchKernelGGL(rt_dw1_kernel, dim3(waves / RT_WAVES), dim3(64 * RT_WAVES), RT_WAVES * RT_DW1_LDS * sizeof(float), stream, m, tape,
                           tape2, items, slab_rows);
    return hipGetLastError();
}

hipError_t rt_debug_read_stamps(unsigned long long* out8) {
#ifdef COLNDE_STAMPS
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_rt_stamps), sizeof(unsigned long long) * 16);
#else
    for (int i = 0; i < 8; i++) out8[i] = 0;
    return hipSuccess;
#endif
}
