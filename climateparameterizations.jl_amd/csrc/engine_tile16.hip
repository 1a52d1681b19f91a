// engine_tile16.hip — the MFMA tile engine (gfx950 / CDNA4 only).
//
// One workgroup owns a tile of CT = 16 columns for the whole time integration.  Every dense layer is a
// small GEMM on v_mfma_f32_16x16x4_f32 with N = the 16 columns:
//   forward     Z_l[c][j]   = b_l[j] + sum_k W_l[j][k] A_{l-1}[c][k]          (A = packed W, B = activations in LDS)
//   backward    dA_{l-1}[c][i] = sum_j W_l[j][i] dZ_l[c][j]                    (A = packed W^T)
//   weight grad dW_l[j][i] += sum_c A_{l-1}[c][i] dZ_l[c][j]                   (K = the 16 columns; accumulators stay in
//                                                                              registers for the whole kernel)
// Activations live in LDS as [column][feature] rows (row stride == 2 mod 4 keeps the B-operand reads conflict-free);
// weights are streamed from L2 in a pre-packed, zero-padded A-operand image (pack_weights_kernel), so the hot loops
// carry no bounds checks.  The physics (flux assembly, Richardson-number diffusivity, flux divergence, Coriolis) and
// its hand-written pullback run between the GEMMs, one thread per (column, face/cell).
//
// Reference arithmetic restated (paths relative to /root/reference):
//   wind_mixing/src/NDE_training.jl:46-165 (NDE, predict_flux, predict_NDE), training_postprocessing.jl:105-153 (NDE!),
//   free_convection/src/free_convection_nde.jl:29-38, convective_adjustment_nde.jl:33-48,
//   src/differentiation_operators.jl:6-29, wind_mixing/src/filtering_operators.jl:1-14, wind_mixing/src/loss.jl:1-9,
//   wind_mixing/src/NDE_training.jl:290-323 (losses), free_convection/double_gyre_nn.jl:149-168 (inference).
#include "colnde_dev.h"
#include <cstdlib>
#include "engine_tile16.h"
#include "split_bf16.h"
#include <algorithm>

#define FWD_MAXR 12   // owner-thread register items per state array in the forward kernel: CT*ns <= FWD_MAXR*blockDim
#define MAXB 4        // bias-gradient accumulators per thread: n_bias <= MAXB*blockDim

// Diagnostic build only (-DCOLNDE_STAMPS): per-phase cycle sums of workgroup 0 / wave 0, read back through
// colnde_debug_stamps().  No stamp executes in the shipped library.
#ifdef COLNDE_STAMPS
__device__ unsigned long long g_stamps[16];
#define STAMP_DECL unsigned long long st_t0 = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP_BEGIN() do { __builtin_amdgcn_sched_barrier(0); st_t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_t0; st_t0 = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_FLUSH() do { if (blockIdx.x == 0 && threadIdx.x == 0) for (int q = 0; q < 8; q++) g_stamps[q] = st_acc[q]; } while (0)
__device__ unsigned long long g_fine[16];
#define FINE_BEGIN() unsigned long long ft0_; do { __builtin_amdgcn_sched_barrier(0); ft0_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define FINE(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (blockIdx.x == 0 && threadIdx.x == 0) g_fine[i] += t_ - ft0_; ft0_ = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FINE_BEGIN()
#define FINE(i)
#define STAMP_DECL
#define STAMP_BEGIN()
#define STAMP(i)
#define STAMP_FLUSH()
#endif

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// acc += sum_{s<n} mfma(ap[s*a_step], bp[s*b_step]): operands are read U steps at a time ahead of their MFMAs.
// (The engine is instruction-issue-bound, so the loop carries no clamps or selects: full chunks, then a remainder.)
#ifndef GU
#define GU 32     // prefetch depth (k-steps) of the chains whose A operand streams from L2 (packed weight image); measured on the
                  // 64-256-256-63 taped adjoint: depth 4: 296 ms, 8: 413 ms, 16: 247 ms, 32: 233 ms (remainders halve the depth)
#endif
template <int U>
__device__ __forceinline__ f32x4 gemm_chain(const float* ap, int a_step, const float* bp, int b_step, int n, f32x4 acc) {
    int s = 0;
    for (; s + U <= n; s += U) {
        float a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            a[u] = ap[(s + u) * a_step];
            b[u] = bp[(s + u) * b_step];
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc = mfma16(a[u], b[u], acc);
    }
    if (U > 1) return s < n ? gemm_chain<(U > 1 ? U / 2 : 1)>(ap + s * a_step, a_step, bp + s * b_step, b_step, n - s, acc) : acc;
    for (; s < n; s++) acc = mfma16(ap[s * a_step], bp[s * b_step], acc);
    return acc;
}

// The same chain with both operands fetched FOUR k-steps at a time: the packed weight image holds, per lane, the A operands of four
// consecutive MFMAs contiguously (one 16-byte load from L2 instead of four dword loads, each of which cost a 64-bit address add), and
// the K index is dealt so that the matching four B operands are four consecutive floats of the activation row (two 8-byte LDS
// reads): k = 16 S + 4 q + j for group S, lane quarter q, MFMA j.  ~1.75 memory instructions per MFMA become ~0.75.
//   ap4: this lane's float4 of group 0 (groups are 64 lanes x 16 bytes apart); bp: the activation row + 4 q; n = groups of 16 k
template <int U>
__device__ __forceinline__ f32x4 gemm_chain4(const float4* ap4, const float* bp, int n, f32x4 acc) {
    int s = 0;
    for (; s + U <= n; s += U) {
        float4 a[U];
        float2 b0[U], b1[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            a[u] = ap4[(size_t)(s + u) * 64];
            b0[u] = *reinterpret_cast<const float2*>(bp + (s + u) * 16);
            b1[u] = *reinterpret_cast<const float2*>(bp + (s + u) * 16 + 2);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            acc = mfma16(a[u].x, b0[u].x, acc);
            acc = mfma16(a[u].y, b0[u].y, acc);
            acc = mfma16(a[u].z, b1[u].x, acc);
            acc = mfma16(a[u].w, b1[u].y, acc);
        }
    }
    if (U > 1) return s < n ? gemm_chain4<(U > 1 ? U / 2 : 1)>(ap4 + (size_t)s * 64, bp + s * 16, n - s, acc) : acc;
    for (; s < n; s++) {
        const float4 a = ap4[(size_t)s * 64];
        const float2 b0 = *reinterpret_cast<const float2*>(bp + s * 16), b1 = *reinterpret_cast<const float2*>(bp + s * 16 + 2);
        acc = mfma16(a.x, b0.x, acc);
        acc = mfma16(a.y, b0.y, acc);
        acc = mfma16(a.z, b1.x, acc);
        acc = mfma16(a.w, b1.y, acc);
    }
    return acc;
}
#ifndef GU4
#define GU4 8     // groups (of four k-steps) in flight in the L2-streamed chains: the same 32 k-steps of prefetch as GU
#endif

// ------------------------------------------------------------------------------------------------
// weight packing: raw Flux.destructure weights -> A-operand images (four floats per lane per group of four MFMAs)
//   forward image  Wf[net][l][mt][S][lane][j]: A[i = mt*16 + (lane&15)][k = 16 S + 4 (lane>>4) + j] = W_l[i][k]
//   backward image Wb[net][l][it][S][lane][j]: A[i = it*16 + (lane&15)][o = 16 S + 4 (lane>>4) + j] = W_l[o][i]
// (K padded with zeros to a multiple of 16.)  W_l[j][k] (out j, in k) sits at w_off[l] + k*no + j  (column-major out x in).
// ------------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(DevModel m, PackInfo pk, const float* __restrict__ w, float* __restrict__ wf,
                                    float* __restrict__ wb) {
    const int total_f = pk.pf_net * m.n_nets, total_b = pk.pb_net * m.n_nets;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total_f + total_b; idx += gridDim.x * blockDim.x) {
        const bool fwd = idx < total_f;
        int e = fwd ? idx : idx - total_f;
        const int pn = fwd ? pk.pf_net : pk.pb_net;
        const int net = e / pn;
        e -= net * pn;
        int l = 0;
        while (l + 1 < m.n_layers && e >= (fwd ? pk.pf_off[l + 1] : pk.pb_off[l + 1])) l++;
        e -= fwd ? pk.pf_off[l] : pk.pb_off[l];
        const int ni = m.sizes[l], no = m.sizes[l + 1];
        const int jj = e & 3, lane = (e >> 2) & 63;
        const int blk = e >> 8;
        const float* W = w + (size_t)net * m.net_size + m.w_off[l];
        float v = 0.0f;
        if (fwd) {
            const int nS = (ni + 15) >> 4;
            const int mt = blk / nS, S = blk - mt * nS;
            const int i = mt * 16 + (lane & 15), k = 16 * S + 4 * (lane >> 4) + jj;
            if (i < no && k < ni) v = W[(size_t)k * no + i];
            wf[idx] = v;
        } else {
            const int nS = (no + 15) >> 4;
            const int it = blk / nS, S = blk - it * nS;
            const int i = it * 16 + (lane & 15), j = 16 * S + 4 * (lane >> 4) + jj;
            if (i < ni && j < no) v = W[(size_t)i * no + j];
            wb[idx - total_f] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dense layers on MFMA
// ------------------------------------------------------------------------------------------------
// Forward pass of all nets.  xs: [CT][ld_x] state rows; A (and Z when STORE_Z): [net][CT][ld_a].
// WLDS: `w` points at the raw weights staged in LDS (A operand read in place, rows clamped so that a padded tile
// never leaves the array; out-of-range k meets a zero B operand); otherwise `w` is global and the A operand comes
// from the packed image `wf`.
template <bool STORE_Z, bool WLDS>
__device__ __forceinline__ void mlp_forward(const DevModel& m, const PackInfo& pk, const float* w,
                                            const float* __restrict__ wf, const float* xs, float* Z, float* A,
                                            int wave, int nwaves, int lane, float* __restrict__ zrec = nullptr, int zld = 0) {
    const int c = lane & 15, kq = lane >> 4;
    FINE_BEGIN();
    for (int l = 0; l < m.n_layers; l++) {
        const int ni = m.sizes[l], no = m.sizes[l + 1];
        const int nmt = (no + 15) >> 4, nk4 = (ni + 3) >> 2;
        const int act = m.acts[l];
        FINE(0);
        for (int job = wave; job < nmt * m.n_nets; job += nwaves) {
            const int net = job / nmt, mt = job - net * nmt;
            const float* bl = w + (size_t)net * m.net_size + m.b_off[l];
            const float* in = (l == 0) ? xs + c * m.ld_x : A + (net * CT + c) * m.ld_a + m.act_off[l - 1];
            // the bias is fetched now and added after the chain, so that its load latency hides under the MFMAs
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
            float bq[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = mt * 16 + 4 * kq + r;
                bq[r] = bl[min(row, no - 1)];
            }
            FINE(1);
            if (WLDS) {
                const float* ap = w + net * m.net_size + m.w_off[l] + kq * no + min(mt * 16 + (lane & 15), no - 1);
                acc = gemm_chain<4>(ap, 4 * no, in + kq, 4, nk4, acc);
            } else {
                const int nS = (ni + 15) >> 4;
                const float4* ap4 = reinterpret_cast<const float4*>(wf + (size_t)net * pk.pf_net + pk.pf_off[l]) + (size_t)mt * nS * 64 + lane;
                acc = gemm_chain4<GU4>(ap4, in + 4 * kq, nS, acc);
            }
            FINE(2);
            const int ro = (net * CT + c) * m.ld_a + m.act_off[l];
            float zv[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = mt * 16 + 4 * kq + r;
                zv[r] = acc[r] + bq[r];
                if (row < no) {
                    if (STORE_Z) Z[ro + row] = zv[r];
                    A[ro + row] = dev_act(act, zv[r]);
                }
            }
            // hidden-layer pre-activations taped for the adjoint: row [c][net][hidden features] of stride zld; the lane's four
            // rows are consecutive and 16-byte aligned (every segment starts at a multiple of 4 floats)
            if (zrec && l + 1 < m.n_layers) {
                float* zo = zrec + c * zld + net * m.act_off[m.n_layers - 1] + m.act_off[l] + mt * 16 + 4 * kq;
                if (mt * 16 + 4 * kq + 3 < no) *reinterpret_cast<float4*>(zo) = make_float4(zv[0], zv[1], zv[2], zv[3]);
                else
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (mt * 16 + 4 * kq + r < no) zo[r] = zv[r];
            }
            FINE(3);
        }
        __syncthreads();
        FINE(4);
    }
}

// Backward pass: on entry Z's last-layer slot holds dZ_L; on exit every Z slot holds dZ_l and xb += W_1^T dZ_1.
template <bool WLDS>
__device__ __forceinline__ void mlp_backward(const DevModel& m, const PackInfo& pk, const float* w,
                                             const float* __restrict__ wb, float* Z, float* xb, int wave, int nwaves,
                                             int lane) {
    const int c = lane & 15, jq = lane >> 4;
    for (int l = m.n_layers - 1; l >= 0; l--) {
        const int ni = m.sizes[l], no = m.sizes[l + 1];
        const int nit = (ni + 15) >> 4, nj4 = (no + 3) >> 2;
        if (l > 0) {
            const int actp = m.acts[l - 1];
            for (int job = wave; job < nit * m.n_nets; job += nwaves) {
                const int net = job / nit, it = job - net * nit;
                const float* dz = Z + (net * CT + c) * m.ld_a + m.act_off[l];
                f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
                if (WLDS) {
                    const float* ap = w + net * m.net_size + m.w_off[l] + min(it * 16 + (lane & 15), ni - 1) * no + jq;
                    acc = gemm_chain<4>(ap, 4, dz + jq, 4, nj4, acc);
                } else {
                    const int nS = (no + 15) >> 4;
                    const float4* ap4 = reinterpret_cast<const float4*>(wb + (size_t)net * pk.pb_net + pk.pb_off[l]) + (size_t)it * nS * 64 + lane;
                    acc = gemm_chain4<GU4>(ap4, dz + 4 * jq, nS, acc);
                }
                const int ro = (net * CT + c) * m.ld_a + m.act_off[l - 1];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = it * 16 + 4 * jq + r;
                    if (row < ni) Z[ro + row] = acc[r] * dev_act_grad(actp, Z[ro + row]);
                }
            }
        } else {
            for (int it = wave; it < nit; it += nwaves) {
                f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
                for (int net = 0; net < m.n_nets; net++) {
                    const float* dz = Z + (net * CT + c) * m.ld_a + m.act_off[0];
                    if (WLDS) {
                        const float* ap = w + net * m.net_size + m.w_off[0] + min(it * 16 + (lane & 15), ni - 1) * no + jq;
                        acc = gemm_chain<4>(ap, 4, dz + jq, 4, nj4, acc);
                    } else {
                        const int nS = (no + 15) >> 4;
                        const float4* ap4 = reinterpret_cast<const float4*>(wb + (size_t)net * pk.pb_net + pk.pb_off[0]) + (size_t)it * nS * 64 + lane;
                        acc = gemm_chain4<GU4>(ap4, dz + 4 * jq, nS, acc);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = it * 16 + 4 * jq + r;
                    if (row < ni) xb[c * m.ld_x + row] += acc[r];
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// The dense chains on the bf16 pipe (round 5; networks whose activation rows live in global memory — DevModel::ag — under COLNDE_MATRIX_BF16X3_EXACT): exact
// three-way operand split (split_bf16.h), six v_mfma_f32_16x16x32_bf16 per 32-deep k-block and 16-row tile.  The weights are split ONCE per call by
// pack_planes_kernel; the activation row's eight values of a lane are split once per k-block and shared by the TG row tiles a wave owns (the split costs
// as much vector time as six of these MFMAs cost matrix time: one tile per split would buy nothing).
// ------------------------------------------------------------------------------------------------
typedef float t16_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 t16_mfma_bf(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 t16_mfma_bf3(const Bf3& a, const Bf3& b, f32x4 c) {     // smallest products first
    c = t16_mfma_bf(a.m, b.m, c);
    c = t16_mfma_bf(a.l, b.h, c);
    c = t16_mfma_bf(a.h, b.l, c);
    c = t16_mfma_bf(a.m, b.h, c);
    c = t16_mfma_bf(a.h, b.m, c);
    c = t16_mfma_bf(a.h, b.h, c);
    return c;
}
#define T16_TG 5      // row tiles per wave and split (25 tiles of a 400-row layer: five jobs per net)

__global__ void pack_planes_kernel(DevModel m, const float* __restrict__ w, unsigned* __restrict__ sf, unsigned* __restrict__ sb) {
    const long total_f = (long)m.sf_net * m.n_nets, total_b = (long)m.sb_net * m.n_nets;         // 16-byte items
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total_f + total_b; idx += (long)gridDim.x * blockDim.x) {
        const bool fwd = idx < total_f;
        long e = fwd ? idx : idx - total_f;
        const int pn = fwd ? m.sf_net : m.sb_net;
        const int net = (int)(e / pn);
        e -= (long)net * pn;
        int l = 0;
        while (l + 1 < m.n_layers && e >= (fwd ? m.sf_off[l + 1] : m.sb_off[l + 1])) l++;
        e -= fwd ? m.sf_off[l] : m.sb_off[l];
        const int ni = m.sizes[l], no = m.sizes[l + 1];
        const int lane = (int)(e & 63), plane = (int)((e >> 6) % 3);
        const long blk = (e >> 6) / 3;
        const int nS = fwd ? (ni + 31) >> 5 : (no + 31) >> 5;
        const int tile = (int)(blk / nS), S = (int)(blk - (long)tile * nS);
        const float* W = w + (size_t)net * m.net_size + m.w_off[l];
        const int i = tile * 16 + (lane & 15);
        unsigned out[4];
#pragma unroll
        for (int pr = 0; pr < 4; pr++) {
            float v[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int k = 32 * S + 8 * (lane >> 4) + 2 * pr + q;
                float x = 0.0f;
                if (fwd) { if (i < no && k < ni) x = W[(size_t)k * no + i]; }            // A[i = out row][k = in]
                else { if (i < ni && k < no) x = W[(size_t)i * no + k]; }               // A[i = in row][k = out]
                const float hh = __uint_as_float(__float_as_uint(x) & 0xffff0000u), r1 = x - hh;
                const float mm = __uint_as_float(__float_as_uint(r1) & 0xffff0000u), ll = r1 - mm;
                v[q] = plane == 0 ? hh : (plane == 1 ? mm : ll);
            }
            out[pr] = (__float_as_uint(v[1]) & 0xffff0000u) | (__float_as_uint(v[0]) >> 16);
        }
        unsigned* dst = (fwd ? sf : sb) + (size_t)(fwd ? idx : idx - total_f) * 4;
        dst[0] = out[0]; dst[1] = out[1]; dst[2] = out[2]; dst[3] = out[3];
    }
}

// eight consecutive floats of an activation / delta row (8-byte aligned): fetched one k-block ahead of the split that makes them a B operand (the rows live in global memory)
struct T16Row8 { float2 v[4]; };
__device__ __forceinline__ T16Row8 t16_row_load8(const float* p) {
    T16Row8 r;
#pragma unroll
    for (int q = 0; q < 4; q++) r.v[q] = *reinterpret_cast<const float2*>(p + 2 * q);
    return r;
}
__device__ __forceinline__ Bf3 t16_row_split8(const T16Row8& r) {
    const float x[8] = {r.v[0].x, r.v[0].y, r.v[1].x, r.v[1].y, r.v[2].x, r.v[2].y, r.v[3].x, r.v[3].y};
    return bf3_split8(x);
}

// forward pass of all nets with the chains on the bf16 pipe (the f32 twin is mlp_forward<false, false>)
__device__ __forceinline__ void mlp_forward_split(const DevModel& m, const float* w, const float* xs, float* A, int wave, int nwaves, int lane,
                                                  float* __restrict__ zrec = nullptr, int zld = 0) {
    const int c = lane & 15, kq = lane >> 4;
    const u32x4* sf = reinterpret_cast<const u32x4*>(m.sf);
    for (int l = 0; l < m.n_layers; l++) {
        const int ni = m.sizes[l], no = m.sizes[l + 1];
        const int nmt = (no + 15) >> 4, nS = (ni + 31) >> 5, ngrp = (nmt + T16_TG - 1) / T16_TG;
        const int act = m.acts[l];
        for (int job = wave; job < ngrp * m.n_nets; job += nwaves) {
            const int net = job / ngrp, mt0 = (job - net * ngrp) * T16_TG;
            const float* bl = w + (size_t)net * m.net_size + m.b_off[l];
            const float* in = (l == 0) ? xs + c * m.ld_x : A + (net * CT + c) * m.ld_a + m.act_off[l - 1];
            f32x4 acc[T16_TG];
#pragma unroll
            for (int t = 0; t < T16_TG; t++) acc[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            const u32x4* ap = sf + (size_t)net * m.sf_net + m.sf_off[l] + (size_t)mt0 * nS * 3 * 64 + lane;
            // the A planes run TWO tiles ahead of their products (L2 latency under the other waves' and this wave's MFMAs); tiles past the layer's last
            // one read the last tile's planes again (never used)
            const int ntl = min(T16_TG, nmt - mt0);
            auto ldA = [&](int q) {                      // q = S * T16_TG + t, clamped
                const int S = min(q / T16_TG, nS - 1), t = min(q % T16_TG, ntl - 1);
                const u32x4* a = ap + ((size_t)t * nS + S) * 3 * 64;
                Bf3 r;
                r.h = a[0]; r.m = a[64]; r.l = a[128];
                return r;
            };
            Bf3 A0 = ldA(0), A1_ = ldA(1);
            T16Row8 rown = t16_row_load8(in + 8 * kq);
            for (int S = 0; S < nS; S++) {
                const Bf3 B = t16_row_split8(rown);
                rown = t16_row_load8(in + 32 * min(S + 1, nS - 1) + 8 * kq);
#pragma unroll
                for (int t = 0; t < T16_TG; t++) {
                    const Bf3 Aop = A0;
                    A0 = A1_;
                    A1_ = ldA(S * T16_TG + t + 2);
                    if (t < ntl) acc[t] = t16_mfma_bf3(Aop, B, acc[t]);
                }
            }
#pragma unroll
            for (int t = 0; t < T16_TG; t++) {
                const int mt = mt0 + t;
                if (mt >= nmt) continue;
                const int ro = (net * CT + c) * m.ld_a + m.act_off[l];
                float zv[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = mt * 16 + 4 * kq + r;
                    zv[r] = acc[t][r] + bl[min(row, no - 1)];
                    if (row < no) A[ro + row] = dev_act(act, zv[r]);
                }
                if (zrec && l + 1 < m.n_layers) {
                    float* zo = zrec + c * zld + net * m.act_off[m.n_layers - 1] + m.act_off[l] + mt * 16 + 4 * kq;
                    if (mt * 16 + 4 * kq + 3 < no) *reinterpret_cast<float4*>(zo) = make_float4(zv[0], zv[1], zv[2], zv[3]);
                    else
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (mt * 16 + 4 * kq + r < no) zo[r] = zv[r];
                }
            }
        }
        __syncthreads();
    }
}

// backward pass, layers L-1 .. 1 on the bf16 pipe (the layer-0 pullback into xb — 6 row tiles, a sum over the nets — stays on the f32 chain: 3 % of the work)
__device__ __forceinline__ void mlp_backward_split(const DevModel& m, const PackInfo& pk, const float* __restrict__ wb, float* Z, float* xb, int wave, int nwaves, int lane) {
    const int c = lane & 15, jq = lane >> 4;
    const u32x4* sb = reinterpret_cast<const u32x4*>(m.sb);
    for (int l = m.n_layers - 1; l >= 1; l--) {
        const int ni = m.sizes[l], no = m.sizes[l + 1];
        const int nit = (ni + 15) >> 4, nS = (no + 31) >> 5, ngrp = (nit + T16_TG - 1) / T16_TG;
        const int actp = m.acts[l - 1];
        for (int job = wave; job < ngrp * m.n_nets; job += nwaves) {
            const int net = job / ngrp, it0 = (job - net * ngrp) * T16_TG;
            const float* dz = Z + (net * CT + c) * m.ld_a + m.act_off[l];
            f32x4 acc[T16_TG];
#pragma unroll
            for (int t = 0; t < T16_TG; t++) acc[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            const u32x4* ap = sb + (size_t)net * m.sb_net + m.sb_off[l] + (size_t)it0 * nS * 3 * 64 + lane;
            const int ntl = min(T16_TG, nit - it0);
            auto ldA = [&](int q) {
                const int S = min(q / T16_TG, nS - 1), t = min(q % T16_TG, ntl - 1);
                const u32x4* a = ap + ((size_t)t * nS + S) * 3 * 64;
                Bf3 r;
                r.h = a[0]; r.m = a[64]; r.l = a[128];
                return r;
            };
            Bf3 A0 = ldA(0), A1_ = ldA(1);
            T16Row8 rown = t16_row_load8(dz + 8 * jq);
            for (int S = 0; S < nS; S++) {
                const Bf3 B = t16_row_split8(rown);
                rown = t16_row_load8(dz + 32 * min(S + 1, nS - 1) + 8 * jq);
#pragma unroll
                for (int t = 0; t < T16_TG; t++) {
                    const Bf3 Aop = A0;
                    A0 = A1_;
                    A1_ = ldA(S * T16_TG + t + 2);
                    if (t < ntl) acc[t] = t16_mfma_bf3(Aop, B, acc[t]);
                }
            }
            const int ro = (net * CT + c) * m.ld_a + m.act_off[l - 1];
#pragma unroll
            for (int t = 0; t < T16_TG; t++) {
                if (it0 + t >= nit) continue;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = (it0 + t) * 16 + 4 * jq + r;
                    if (row < ni) Z[ro + row] = acc[t][r] * dev_act_grad(actp, Z[ro + row]);
                }
            }
        }
        __syncthreads();
    }
    {   // l = 0: xb += sum over nets of W_1^T dZ_1 (f32 chain from the packed image)
        const int ni = m.sizes[0], no = m.sizes[1];
        const int nit = (ni + 15) >> 4, nS = (no + 15) >> 4;
        for (int it = wave; it < nit; it += nwaves) {
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int net = 0; net < m.n_nets; net++) {
                const float* dz = Z + (net * CT + c) * m.ld_a + m.act_off[0];
                const float4* ap4 = reinterpret_cast<const float4*>(wb + (size_t)net * pk.pb_net + pk.pb_off[0]) + (size_t)it * nS * 64 + lane;
                acc = gemm_chain4<GU4>(ap4, dz + 4 * jq, nS, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = it * 16 + 4 * jq + r;
                if (row < ni) xb[c * m.ld_x + row] += acc[r];
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// physics
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float filt3(const float* x, int i, int N) {   // (F x)[i], filtering_operators.jl:1-14
    if (i == 0) return 0.5f * (x[0] + x[1]);
    if (i == N - 1) return 0.5f * (x[N - 2] + x[N - 1]);
    return (x[i - 1] + x[i] + x[i + 1]) * (1.0f / 3.0f);
}
__device__ __forceinline__ float filt3T(const float* x, int j, int N) {  // (F^T x)[j]
    float s = 0.0f;
    for (int i = j - 1; i <= j + 1; i++)
        if (i >= 0 && i < N) s += x[i] * ((i == 0 || i == N - 1) ? 0.5f : (1.0f / 3.0f));
    return s;
}

__device__ __forceinline__ float wm_top_flux(const DevModel& m, const float* bc, float t) {
    if (!m.diurnal) return bc[5];
    // scalings.wT(Q sin(2π/86400 · tτ)/(αg)) — NDE_training.jl:73, data_containers.jl:135
    const float wq = bc[5] * sinf(6.283185307179586f / 86400.0f * (t * m.tau)) / m.alpha_g;
    return (wq - m.mu_wT) / m.sig_wT;
}

struct FaceGrad { float gu, gv, gT, S2, Ri; };

// tanh(y) = 1 - 2/(1 + e^{2y}) on the fast exp / reciprocal units (|rel err| ~ 1e-6)
__device__ __forceinline__ float fast_tanh(float y) {
    const float e = __expf(2.0f * fminf(fmaxf(y, -15.0f), 15.0f));
    return 1.0f - fast_div(2.0f, 1.0f + e);
}

__device__ __forceinline__ FaceGrad wm_face(const DevModel& m, const float* x, int f, float eps) {
    const int Nz = m.Nz;
    const bool in = f >= 1 && f < Nz;
    FaceGrad g;
    g.gu = in ? (x[f] - x[f - 1]) * (float)Nz : 0.0f;
    g.gv = in ? (x[Nz + f] - x[Nz + f - 1]) * (float)Nz : 0.0f;
    g.gT = in ? (x[2 * Nz + f] - x[2 * Nz + f - 1]) * (float)Nz : 0.0f;
    const float a1 = m.sig_u * (g.gu + eps), a2 = m.sig_v * (g.gv + eps);
    g.S2 = a1 * a1 + a2 * a2;
    g.Ri = fast_div(m.B * (g.gT + eps), g.S2);   // local_richardson, NDE_training.jl:46-52
    return g;
}

// k[c][:] = RHS(xs[c][:]).  A holds the nets' activations (last layer = interior fluxes).  Ends with a barrier.
__device__ void physics_forward(const DevModel& m, const float* xs, const float* A, float* F, float* Ri_l,
                                const float* bcl, float t, float* kk, int tid, int nth) {
    const int Nz = m.Nz, nf = Nz + 1, nout = Nz - 1;
    const int oo = m.act_off[m.n_layers - 1];
    if (m.model == COLNDE_MODEL_WIND_MIXING) {
        const float eps = m.inplace ? 0.0f : m.eps;
        if (m.mpp && m.smooth_Ri) {
            for (int it = tid; it < CT * nf; it += nth) {
                const int c = it / nf, f = it - c * nf;
                Ri_l[c * m.ld_f + f] = wm_face(m, xs + c * m.ld_x, f, eps).Ri;
            }
            __syncthreads();
        }
        for (int it = tid; it < CT * nf; it += nth) {
            const int c = it / nf, f = it - c * nf;
            const bool in = f >= 1 && f < Nz;
            const float* bc = bcl + c * 8;
            float Fk[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float* o = A + (k * CT + c) * m.ld_a + oo;
                float ov = 0.0f;
                if (in) ov = m.smooth_NN ? filt3(o, f - 1, nout) : o[f - 1];
                const float bb = bc[2 * k];
                const float bt = (k == 2) ? wm_top_flux(m, bc, t) : bc[2 * k + 1];
                Fk[k] = m.zero_w ? ov : (f == 0 ? bb : (f == Nz ? bt : ov));
                if (m.mpp && m.zero_w) {
                    if (f == 0) Fk[k] += bb - m.s0[k];
                    if (f == Nz) Fk[k] += (m.inplace && m.diurnal && k == 2) ? bt : bt - m.s0[k];
                }
            }
            if (m.mpp) {
                if (in) {
                    const FaceGrad g = wm_face(m, xs + c * m.ld_x, f, eps);
                    const float Ris = m.smooth_Ri ? filt3(Ri_l + c * m.ld_f, f, nf) : g.Ri;
                    const float th = fast_tanh((Ris - m.Ric) * m.inv_dRi);
                    const float nu = m.nu0 + m.nu_minus * (1.0f - th) * 0.5f;   // tanh_step, :54,:125
                    float nuT = nu / m.Pr;
                    if (m.inplace && m.ca) nuT = g.gu > 0.0f ? nu / m.Pr : m.kappa;  // training_postprocessing.jl:118-121
                    Fk[0] -= m.cs[0] * nu * g.gu;
                    Fk[1] -= m.cs[1] * nu * g.gv;
                    Fk[2] -= m.cs[2] * nuT * g.gT;
                }
            } else if (m.ca && in) {
                const float* x = xs + c * m.ld_x;
                const float gT = (x[2 * Nz + f] - x[2 * Nz + f - 1]) * (float)Nz;
                Fk[2] -= m.cs[2] * m.kappa * fminf(0.0f, gT);
            }
#pragma unroll
            for (int k = 0; k < 3; k++) F[(k * CT + c) * m.ld_f + f] = Fk[k];
        }
        __syncthreads();
        for (int it = tid; it < CT * Nz; it += nth) {
            const int c = it / Nz, i = it - c * Nz;
            const float* x = xs + c * m.ld_x;
            const float* F0 = F + (0 * CT + c) * m.ld_f;
            const float* F1 = F + (1 * CT + c) * m.ld_f;
            const float* F2 = F + (2 * CT + c) * m.ld_f;
            float* ko = kk + c * m.ld_x;
            ko[i] = -m.A[0] * (F0[i + 1] - F0[i]) + m.cor_u * (m.sig_v * x[Nz + i] + m.mu_v);
            ko[Nz + i] = -m.A[1] * (F1[i + 1] - F1[i]) - m.cor_v * (m.sig_u * x[i] + m.mu_u);
            ko[2 * Nz + i] = -m.A[2] * (F2[i + 1] - F2[i]);
        }
    } else {
        const bool ca = m.model == COLNDE_MODEL_CONV_ADJ_NDE;
        for (int it = tid; it < CT * nf; it += nth) {
            const int c = it / nf, f = it - c * nf;
            const bool in = f >= 1 && f < Nz;
            const float* x = xs + c * m.ld_x;
            const float* o = A + c * m.ld_a + oo;
            const float wv = f == 0 ? bcl[c * 8] : (f == Nz ? bcl[c * 8 + 1] : o[f - 1]);
            float q = 0.0f;
            if (ca && in) q = fminf(0.0f, m.ca_K * (x[f] - x[f - 1]) * (float)Nz);
            F[c * m.ld_f + f] = wv - q;     // dT = -C Nz d(w - q)
        }
        __syncthreads();
        const float CN = m.C_fc * (float)Nz;
        for (int it = tid; it < CT * Nz; it += nth) {
            const int c = it / Nz, i = it - c * Nz;
            const float* Fw = F + c * m.ld_f;
            kk[c * m.ld_x + i] = -CN * (Fw[i + 1] - Fw[i]);
        }
    }
    __syncthreads();
}

// Pullback of the physics for cotangent dbar: writes xb (physics part) and dZ_L into Z's last-layer slot.
// gb: [3][CT][ld_f] scratch, Ri_l / Rib_l: [CT][ld_f] scratch.  Ends with a barrier.
// sw_mode (RKC2 only): the pullback of the convective-adjustment switch min(0, K dT/dz).  Within one stabilised step the stage
// Jacobians must share ONE switch pattern: the stage polynomials rely on cancellations that hold only for a common Jacobian, and
// with per-stage patterns the exact discrete adjoint of the recurrence grows without bound (1e12 .. 1e57 measured on the oracle,
// DESIGN §2).  0: evaluate the switch at this stage's state (RK4: exact discrete adjoint); 1: evaluate it and record it in Ri_l
// (the first stage of a step the backward sweep meets, Y_{s-1}); 2: use the recorded pattern.
__device__ void physics_vjp(const DevModel& m, const float* xs, const float* dbar, float* Z, float* xb, float* gb,
                            float* Ri_l, float* Rib_l, int tid, int nth, int sw_mode = 0) {
    const int Nz = m.Nz, nf = Nz + 1, nout = Nz - 1;
    const int L = m.n_layers;
    const int oo = m.act_off[L - 1];
    const int actL = m.acts[L - 1];
    if (m.model == COLNDE_MODEL_WIND_MIXING) {
        const float eps = m.eps;
        if (m.mpp && m.smooth_Ri) {
            for (int it = tid; it < CT * nf; it += nth) {
                const int c = it / nf, f = it - c * nf;
                Ri_l[c * m.ld_f + f] = wm_face(m, xs + c * m.ld_x, f, eps).Ri;
            }
            __syncthreads();
        }
        for (int it = tid; it < CT * nf; it += nth) {
            const int c = it / nf, f = it - c * nf;
            const bool in = f >= 1 && f < Nz;
            const float* db = dbar + c * m.ld_x;
            float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f, ribs = 0.0f;
            if (in) {
                const float Fb0 = m.A[0] * (db[f] - db[f - 1]);
                const float Fb1 = m.A[1] * (db[Nz + f] - db[Nz + f - 1]);
                const float Fb2 = m.A[2] * (db[2 * Nz + f] - db[2 * Nz + f - 1]);
                if (m.mpp) {
                    const FaceGrad g = wm_face(m, xs + c * m.ld_x, f, eps);
                    const float Ris = m.smooth_Ri ? filt3(Ri_l + c * m.ld_f, f, nf) : g.Ri;
                    const float th = fast_tanh((Ris - m.Ric) * m.inv_dRi);
                    const float nu = m.nu0 + m.nu_minus * (1.0f - th) * 0.5f;
                    const float D0 = -Fb0, D1 = -Fb1, D2 = -Fb2;
                    g0 = D0 * m.cs[0] * nu;
                    g1 = D1 * m.cs[1] * nu;
                    g2 = D2 * m.cs[2] * nu / m.Pr;
                    const float nub = D0 * m.cs[0] * g.gu + D1 * m.cs[1] * g.gv + D2 * m.cs[2] * g.gT / m.Pr;
                    ribs = nub * (-m.nu_minus / (2.0f * m.dRi)) * (1.0f - th * th);
                    if (!m.smooth_Ri) {
                        g2 += fast_div(ribs * m.B, g.S2);
                        const float q = fast_div(ribs * -g.Ri, g.S2) * 2.0f;
                        g0 += q * m.sig_u * m.sig_u * (g.gu + eps);
                        g1 += q * m.sig_v * m.sig_v * (g.gv + eps);
                    }
                } else if (m.ca) {
                    const float* x = xs + c * m.ld_x;
                    const float gT = (x[2 * Nz + f] - x[2 * Nz + f - 1]) * (float)Nz;
                    bool on = gT < 0.0f;
                    if (sw_mode == 2) on = Ri_l[c * m.ld_f + f] != 0.0f;
                    if (sw_mode == 1) Ri_l[c * m.ld_f + f] = on ? 1.0f : 0.0f;
                    g2 = on ? -Fb2 * m.cs[2] * m.kappa : 0.0f;
                }
            }
            gb[(0 * CT + c) * m.ld_f + f] = g0;
            gb[(1 * CT + c) * m.ld_f + f] = g1;
            gb[(2 * CT + c) * m.ld_f + f] = g2;
            if (m.mpp && m.smooth_Ri) Rib_l[c * m.ld_f + f] = ribs;
        }
        __syncthreads();
        if (m.mpp && m.smooth_Ri) {
            for (int it = tid; it < CT * nf; it += nth) {
                const int c = it / nf, f = it - c * nf;
                if (f >= 1 && f < Nz) {
                    const FaceGrad g = wm_face(m, xs + c * m.ld_x, f, eps);
                    const float rib = filt3T(Rib_l + c * m.ld_f, f, nf);
                    gb[(2 * CT + c) * m.ld_f + f] += rib * m.B / g.S2;
                    const float q = rib * (-g.Ri / g.S2) * 2.0f;
                    gb[(0 * CT + c) * m.ld_f + f] += q * m.sig_u * m.sig_u * (g.gu + eps);
                    gb[(1 * CT + c) * m.ld_f + f] += q * m.sig_v * m.sig_v * (g.gv + eps);
                }
            }
            __syncthreads();
        }
        for (int it = tid; it < CT * Nz; it += nth) {
            const int c = it / Nz, i = it - c * Nz;
            const float* db = dbar + c * m.ld_x;
            float* xo = xb + c * m.ld_x;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float* g = gb + (k * CT + c) * m.ld_f;
                float v = (g[i] - g[i + 1]) * (float)Nz;          // transpose of Dᶠ; g[0] = g[Nz] = 0
                if (k == 0) v += -m.cor_v * m.sig_u * db[Nz + i];
                if (k == 1) v += m.cor_u * m.sig_v * db[i];
                xo[k * Nz + i] = v;
            }
            if (i < nout) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    float ob;
                    if (m.smooth_NN) {
                        ob = 0.0f;
                        for (int q = i - 1; q <= i + 1; q++)
                            if (q >= 0 && q < nout)
                                ob += m.A[k] * (db[k * Nz + q + 1] - db[k * Nz + q]) * ((q == 0 || q == nout - 1) ? 0.5f : (1.0f / 3.0f));
                    } else {
                        ob = m.A[k] * (db[k * Nz + i + 1] - db[k * Nz + i]);
                    }
                    const int zo = (k * CT + c) * m.ld_a + oo + i;
                    Z[zo] = ob * dev_act_grad(actL, Z[zo]);
                }
            }
        }
    } else {
        const bool ca = m.model == COLNDE_MODEL_CONV_ADJ_NDE;
        const float CN = m.C_fc * (float)Nz;
        for (int it = tid; it < CT * nf; it += nth) {
            const int c = it / nf, f = it - c * nf;
            float g = 0.0f;
            if (ca && f >= 1 && f < Nz) {
                const float* x = xs + c * m.ld_x;
                const float* db = dbar + c * m.ld_x;
                const float wbf = CN * (db[f] - db[f - 1]);
                const float gT = (x[f] - x[f - 1]) * (float)Nz;
                bool on = gT < 0.0f;
                if (sw_mode == 2) on = Ri_l[c * m.ld_f + f] != 0.0f;
                if (sw_mode == 1) Ri_l[c * m.ld_f + f] = on ? 1.0f : 0.0f;
                g = on ? -wbf * m.ca_K : 0.0f;
            }
            gb[c * m.ld_f + f] = g;
        }
        __syncthreads();
        for (int it = tid; it < CT * Nz; it += nth) {
            const int c = it / Nz, i = it - c * Nz;
            const float* db = dbar + c * m.ld_x;
            const float* g = gb + c * m.ld_f;
            xb[c * m.ld_x + i] = (g[i] - g[i + 1]) * (float)Nz;
            if (i < nout) {
                const int zo = c * m.ld_a + oo + i;
                Z[zo] = CN * (db[i + 1] - db[i]) * dev_act_grad(actL, Z[zo]);
            }
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// flux-MLP pre-training: `train_NN` — wind_mixing/src/NN_training.jl:207-249 with predict_uw/vw/wT (:25-169) —
// and the T -> wT pre-training of free_convection/train_free_convection_nde.jl:186-216.  `Flux.train!(NN_loss, params, data, opt)`
// makes ONE ADAM update per data point, in order: an inherently serial chain of tiny dense products, so one workgroup walks the
// whole (shuffled) data set with the weights, moments and activations on chip / in L2, and only the per-pass loss leaves the GPU.
//     NN_flux = [b; NN(x); t]-style face vector minus the diffusive / adjustment flux of the state (independent of the weights)
//     loss    = mse(NN_flux, flux) + gradient_scaling * mse(D^c flux, D^c NN_flux)
// update = 0 evaluates the mean loss at fixed weights (`total_loss(training_data)`, :234-236).
// LDS: act[act_total + ns] (a_0 = x, then every layer's activations), z[act_total] pre-activations, d[act_total] deltas,
//      F, y, c: [Nz + 1] face vectors, red[64].
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512)
pretrain_kernel(DevModel m, int flux_type, float* __restrict__ theta, float* __restrict__ mom, float* __restrict__ vel,
                const float* __restrict__ X, const float* __restrict__ BC, const float* __restrict__ Y, const int* __restrict__ order,
                int n_samples, float gs, float eta, float b1, float b2, float eps_adam, double bt1, double bt2, int update,
                float* __restrict__ loss_out, double* __restrict__ bt_out) {
    extern __shared__ __attribute__((aligned(16))) float pt_smem[];
    const int tid = threadIdx.x, nth = blockDim.x;
    const int Nz = m.Nz, nf = Nz + 1, L = m.n_layers, ns = m.ns;
    float* a0 = pt_smem;                       // x
    float* act = a0 + ((ns + 3) & ~3);         // a_l at act + act_off[l-1]... (act_off[l] = offset of layer l+1's outputs)
    float* zz = act + m.act_total + 4;
    float* dd = zz + m.act_total + 4;
    float* F = dd + m.act_total + 4;
    float* yv = F + nf + 3;
    float* cf = yv + nf + 3;
    float* red = cf + nf + 3;
    const bool wm = m.model == COLNDE_MODEL_WIND_MIXING;
    const int k = wm ? flux_type : 2;
    double p1 = bt1, p2 = bt2;                 // running powers beta^t (uniform over the workgroup)
    float loss_sum = 0.0f;
    for (int s = 0; s < n_samples; s++) {
        const int idx = order ? order[s] : s;
        for (int i = tid; i < ns; i += nth) a0[i] = X[(size_t)idx * ns + i];
        for (int f = tid; f < nf; f += nth) yv[f] = Y[(size_t)idx * nf + f];
        __syncthreads();
        // weight-independent part of the face flux: boundary faces and the diffusive / adjustment flux of the state
        for (int f = tid; f < nf; f += nth) {
            const float* bc = BC + (size_t)idx * m.n_bc;
            float c = 0.0f;
            if (wm) {
                const float bb = bc[2 * k], bt = bc[2 * k + 1];
                const bool in = f >= 1 && f < Nz;
                if (!in) c = m.zero_w ? (f == 0 ? bb : bt) - m.s0[k] : (f == 0 ? bb : bt);      // (MPP's end faces: NN_training.jl:62-66)
                else if (m.mpp) {
                    const FaceGrad g = wm_face(m, a0, f, m.eps);
                    const float nu = m.nu0 + m.nu_minus * (1.0f - fast_tanh((g.Ri - m.Ric) * m.inv_dRi)) * 0.5f;
                    c = -m.cs[k] * (k == 2 ? nu * m.inv_Pr : nu) * (k == 0 ? g.gu : (k == 1 ? g.gv : g.gT));
                } else if (m.ca && k == 2) {
                    const float gT = (a0[2 * Nz + f] - a0[2 * Nz + f - 1]) * (float)Nz;
                    c = -m.cs[2] * m.kappa * fminf(0.0f, gT);
                }
            } else {
                c = f == 0 ? bc[0] : (f == Nz ? bc[1] : 0.0f);
            }
            cf[f] = c;
        }
        // forward
        const float* th = theta + (wm ? k * m.net_size : 0);
        for (int l = 0; l < L; l++) {
            const int ni = m.sizes[l], no = m.sizes[l + 1];
            const float* ain = l == 0 ? a0 : act + m.act_off[l - 1];
            const float* W = th + m.w_off[l];
            for (int o = tid; o < no; o += nth) {
                float acc = th[m.b_off[l] + o];
                for (int i = 0; i < ni; i++) acc = fmaf(W[i * no + o], ain[i], acc);
                zz[m.act_off[l] + o] = acc;
                act[m.act_off[l] + o] = dev_act(m.acts[l], acc);
            }
            __syncthreads();
        }
        // face flux, loss and its cotangent on the network output
        const float* out = act + m.act_off[L - 1];
        for (int f = tid; f < nf; f += nth) F[f] = cf[f] + ((f >= 1 && f < Nz) ? out[f - 1] : 0.0f);
        __syncthreads();
        float part = 0.0f;
        for (int f = tid; f < nf; f += nth) {
            const float r = F[f] - yv[f];
            part += r * r * (1.0f / (float)nf);
            if (f < Nz) {
                const float dg = ((F[f + 1] - F[f]) - (yv[f + 1] - yv[f])) * (float)Nz;
                part += gs * dg * dg * (1.0f / (float)Nz);
            }
            if (f >= 1 && f < Nz) {
                const float dgm = ((F[f] - F[f - 1]) - (yv[f] - yv[f - 1])) * (float)Nz;
                const float dgp = ((F[f + 1] - F[f]) - (yv[f + 1] - yv[f])) * (float)Nz;
                const float g = 2.0f / (float)nf * r + gs * 2.0f * (dgm - dgp);
                dd[m.act_off[L - 1] + f - 1] = g * dev_act_grad(m.acts[L - 1], zz[m.act_off[L - 1] + f - 1]);
            }
        }
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off);
        if ((tid & 63) == 0) red[tid >> 6] = part;
        __syncthreads();
        if (tid == 0) {
            float t = 0.0f;
            for (int w = 0; w < (nth >> 6); w++) t += red[w];
            loss_sum += t;
        }
        if (!update) { __syncthreads(); continue; }
        // backward: every delta with the weights as they stood before this sample's update
        for (int l = L - 1; l >= 1; l--) {
            const int ni = m.sizes[l], no = m.sizes[l + 1];
            const float* W = th + m.w_off[l];
            for (int i = tid; i < ni; i += nth) {
                float acc = 0.0f;
                for (int o = 0; o < no; o++) acc = fmaf(W[i * no + o], dd[m.act_off[l] + o], acc);
                dd[m.act_off[l - 1] + i] = acc * dev_act_grad(m.acts[l - 1], zz[m.act_off[l - 1] + i]);
            }
            __syncthreads();
        }
        // Flux ADAM (apply! + update!) on every parameter of this net; gradient element = delta_out * input (bias: delta_out)
        const float c1 = (float)(1.0 / (1.0 - p1)), c2 = (float)(1.0 / (1.0 - p2));
        float* tw = theta + (wm ? k * m.net_size : 0);
        float* tm = mom + (wm ? k * m.net_size : 0);
        float* tv = vel + (wm ? k * m.net_size : 0);
        for (int l = 0; l < L; l++) {
            const int ni = m.sizes[l], no = m.sizes[l + 1];
            const float* ain = l == 0 ? a0 : act + m.act_off[l - 1];
            const int nw = ni * no;
            for (int e = tid; e < nw + no; e += nth) {
                const bool bias = e >= nw;
                const int o = bias ? e - nw : e % no, i = bias ? 0 : e / no;
                const float g = dd[m.act_off[l] + o] * (bias ? 1.0f : ain[i]);
                const int q = (bias ? m.b_off[l] - nw : m.w_off[l]) + e;
                const float mt = b1 * tm[q] + (1.0f - b1) * g;
                const float vt = b2 * tv[q] + (1.0f - b2) * g * g;
                tm[q] = mt;
                tv[q] = vt;
                tw[q] -= eta * (mt * c1) / (sqrtf(vt * c2) + eps_adam);
            }
        }
        p1 *= (double)b1;
        p2 *= (double)b2;
        __syncthreads();
    }
    if (tid == 0) {
        loss_out[0] = loss_sum;
        if (bt_out) { bt_out[0] = p1; bt_out[1] = p2; }
    }
}

hipError_t launch_pretrain(const DevModel& m, int flux_type, float* theta, float* mom, float* vel, const float* X, const float* BC,
                           const float* Y, const int* order, int n_samples, float gs, float eta, float b1, float b2, float eps,
                           double bt1, double bt2, int update, float* loss_out, double* bt_out, hipStream_t stream) {
    const size_t lds = (size_t)(((m.ns + 3) & ~3) + 3 * (m.act_total + 4) + 3 * (m.Nz + 4) + 64) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pretrain_kernel, dim3(1), dim3(512), lds, stream, m, flux_type, theta, mom, vel, X, BC, Y, order, n_samples, gs, eta,
                       b1, b2, eps, bt1, bt2, update, loss_out, bt_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// LDS carving (must match engine_tile16.h: lds_floats_*)
// ------------------------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) float smem[];

__device__ __forceinline__ void load_bcs(const DevModel& m, const float* __restrict__ bcs, float* bcl, int col0,
                                         int n_col, int tid) {
    if (tid < CT * 8) {
        const int c = tid >> 3, q = tid & 7;
        const int col = min(col0 + c, n_col - 1);
        bcl[tid] = q < m.n_bc ? bcs[(size_t)col * m.n_bc + q] : 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// single RHS evaluation (NDE / NDE! / ∂T∂t drop-in)
// ------------------------------------------------------------------------------------------------
// AG: the activation rows live in the workgroup's slab of m.ag (global memory) instead of LDS — see DevModel::ag
template <bool AG = false>
__global__ void __launch_bounds__(256) rhs_kernel(DevModel m, PackInfo pk, const float* __restrict__ w, const float* __restrict__ wf,
                           const float* __restrict__ x, const float* __restrict__ bcs, float t,
                           float* __restrict__ dx, float* __restrict__ flux, int n_col) {
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nth >> 6;
    float* xs = smem;
    float* kk = xs + CT * m.ld_x;
    float* A = AG ? m.ag + (size_t)blockIdx.x * m.n_nets * CT * m.ld_a : kk + CT * m.ld_x;
    float* F = AG ? kk + CT * m.ld_x : A + m.n_nets * CT * m.ld_a;
    float* Ri_l = F + 3 * CT * m.ld_f;
    float* bcl = Ri_l + CT * m.ld_f;
    const int total = (int)(bcl + CT * 8 - smem);
    for (int i = tid; i < total; i += nth) smem[i] = 0.0f;
    __syncthreads();
    const int col0 = blockIdx.x * CT;
    load_bcs(m, bcs, bcl, col0, n_col, tid);
    for (int it = tid; it < CT * m.ns; it += nth) {
        const int c = it / m.ns, i = it - c * m.ns;
        xs[c * m.ld_x + i] = x[(size_t)min(col0 + c, n_col - 1) * m.ns + i];
    }
    __syncthreads();
    if (AG && m.sf) mlp_forward_split(m, w, xs, A, wave, nwaves, lane);
    else mlp_forward<false, false>(m, pk, w, wf, xs, nullptr, A, wave, nwaves, lane);
    physics_forward(m, xs, A, F, Ri_l, bcl, t, kk, tid, nth);
    if (dx)
        for (int it = tid; it < CT * m.ns; it += nth) {
            const int c = it / m.ns, i = it - c * m.ns;
            if (col0 + c < n_col) dx[(size_t)(col0 + c) * m.ns + i] = kk[c * m.ld_x + i];
        }
    if (flux) {
        // predict_flux (NDE_training.jl:83-147): the face vectors the tendencies difference — NN flux minus the closure's diffusive flux, with the
        // boundary faces as the conditions say; free convection: [b; NN(T); t] (- min(0, K dT/dz): free_convection/src/solve.jl:32-46)
        const int nf = m.Nz + 1;
        for (int it = tid; it < CT * m.n_nets * nf; it += nth) {
            const int c = it / (m.n_nets * nf), r = it - c * m.n_nets * nf, k = r / nf, f = r - k * nf;
            if (col0 + c < n_col) flux[((size_t)(col0 + c) * m.n_nets + k) * nf + f] = F[(k * CT + c) * m.ld_f + f];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward solve: classical RK4, S sub-steps per save interval, state at save points -> sol,
// stage inputs of every step -> tape (read back by the adjoint kernel)
// ------------------------------------------------------------------------------------------------
template <bool WLDS, int NTH, bool RKC = false, bool AG = false>
__global__ void __launch_bounds__(NTH) forward_kernel(DevModel m, PackInfo pk, const float* __restrict__ w, const float* __restrict__ wf,
                               const float* __restrict__ x0, const float* __restrict__ bcs,
                               const float* __restrict__ save_times, int n_save, int substeps,
                               float* __restrict__ sol, float* __restrict__ tape, int n_col, float* __restrict__ ztape) {
    constexpr int FMAXR = 3072 / NTH;      // owner-thread register items per state array: CT * ns <= 3072 (ns <= 192)
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nth >> 6;
    float* wl = smem;                                   // raw weights (WLDS) + 128 floats of zero padding
    float* xs = smem + (WLDS ? ((m.n_params + 3) & ~3) + 128 : 0);
    float* kk = xs + CT * m.ld_x;
    const int zld = (m.n_nets * m.act_off[m.n_layers - 1] + 3) & ~3;      // floats per column of the hidden pre-activation tape
    float* A = AG ? m.ag + (size_t)blockIdx.x * m.n_nets * CT * m.ld_a : kk + CT * m.ld_x;
    float* F = AG ? kk + CT * m.ld_x : A + m.n_nets * CT * m.ld_a;
    float* Ri_l = F + 3 * CT * m.ld_f;
    float* bcl = Ri_l + CT * m.ld_f;
    const int total = (int)(bcl + CT * 8 - smem);
    for (int i = tid; i < total; i += nth) smem[i] = 0.0f;
    __syncthreads();
    if (WLDS) for (int i = tid; i < m.n_params; i += nth) wl[i] = w[i];
    const float* wsrc = WLDS ? wl : w;
    const int col0 = blockIdx.x * CT;
    const int n_items = CT * m.ns;
    load_bcs(m, bcs, bcl, col0, n_col, tid);

    float xn[FMAXR], acc[FMAXR];
#pragma unroll
    for (int r = 0; r < FMAXR; r++) {
        const int it = tid + r * nth;
        xn[r] = 0.0f;
        acc[r] = 0.0f;
        if (it < n_items) {
            const int c = it / m.ns, i = it - c * m.ns;
            xn[r] = x0[(size_t)min(col0 + c, n_col - 1) * m.ns + i];
            if (sol && col0 + c < n_col) sol[((size_t)(col0 + c) * n_save) * m.ns + i] = xn[r];
        }
    }
    const int n_steps = (n_save - 1) * substeps;
    const int nst = m.nst;                               // RHS evaluations (taped stage inputs) per step: 4 (RK4) or s (RKC2)
    float* tp = tape ? tape + (size_t)blockIdx.x * n_steps * nst * n_items : nullptr;
    int step = 0;
    if (RKC) {     // (a template parameter: the RK4 instantiation must not carry this branch's registers — 244 -> 262 VGPRs cost it a wave per SIMD)
        // ---- s-stage RKC2 steps (colnde_dev.h; coefficients from the host table): per owner item Y_0 = xn, Y_{j-1} = ym1,
        //      Y_{j-2} = ym2 and F_0 = f0 stay in registers; stage st evaluates F_st = f(Y_st), the step ends with Y_s
        const float* mu_t = m.rkc, *nu_t = m.rkc + RKC_LD, *mut_t = m.rkc + 2 * RKC_LD, *gat_t = m.rkc + 3 * RKC_LD, *c_t = m.rkc + 4 * RKC_LD;
        float ym1[FMAXR], ym2[FMAXR];                    // acc[] serves as f0
        for (int iv = 0; iv < n_save - 1; iv++) {
            const float t0 = save_times[iv];
            const float dt = (save_times[iv + 1] - t0) / (float)substeps;
            for (int s = 0; s < substeps; s++, step++) {
                const float ts = t0 + (float)s * dt;
#pragma nounroll
                for (int st = 0; st <= nst; st++) {      // st = nst: only the final combination Y_s
                    const float cmu = mu_t[st], cnu = nu_t[st], cmt = mut_t[st] * dt, cga = gat_t[st] * dt;
                    const bool last = st == nst;
                    const bool save = last && s == substeps - 1;
#pragma unroll
                    for (int r = 0; r < FMAXR; r++) {
                        const int it = tid + r * nth;
                        if (it < n_items) {
                            const int c = it / m.ns, i = it - c * m.ns;
                            const int o = c * m.ld_x + i;
                            // increment form d_j = Y_j - Y_0 (the weights of Y_0, Y_{j-1}, Y_{j-2} sum to one): the differences
                            // 2 d_{j-1} - d_{j-2} are taken between increments, not O(1) states — float32 stays accurate
                            float dj = 0.0f;
                            if (st == 1) {
                                acc[r] = kk[o];                                        // F_0
                                dj = cmt * acc[r];
                            } else if (st >= 2) {
                                dj = cmu * ym1[r] + cnu * ym2[r] + cmt * kk[o] + cga * acc[r];
                            }
                            const float v = xn[r] + dj;
                            ym2[r] = st == 0 ? 0.0f : ym1[r];
                            ym1[r] = dj;
                            if (last) {
                                xn[r] = v;
                                if (save && sol && col0 + c < n_col) sol[((size_t)(col0 + c) * n_save + iv + 1) * m.ns + i] = v;
                            } else {
                                xs[o] = v;
                                if (tp) tp[((size_t)step * nst + st) * n_items + it] = v;
                            }
                        }
                    }
                    __syncthreads();
                    if (last) break;
                    if (AG && m.sf) mlp_forward_split(m, wsrc, xs, A, wave, nwaves, lane,
                                             ztape ? ztape + ((size_t)blockIdx.x * n_steps * nst + (size_t)step * nst + st) * ((size_t)CT * zld) : nullptr, zld);
                    else
                    mlp_forward<false, WLDS>(m, pk, wsrc, wf, xs, nullptr, A, wave, nwaves, lane,
                                             ztape ? ztape + ((size_t)blockIdx.x * n_steps * nst + (size_t)step * nst + st) * ((size_t)CT * zld) : nullptr, zld);
                    physics_forward(m, xs, A, F, Ri_l, bcl, ts + c_t[st] * dt, kk, tid, nth);
                }
            }
        }
        return;
    }
    for (int iv = 0; iv < n_save - 1; iv++) {
        const float t0 = save_times[iv];
        const float dt = (save_times[iv + 1] - t0) / (float)substeps;
        for (int s = 0; s < substeps; s++, step++) {
            const float ts = t0 + (float)s * dt;
#pragma nounroll
            for (int st = 0; st < 4; st++) {
                const float ca = st == 0 ? 0.0f : (st == 3 ? 1.0f : 0.5f);            // stage abscissa
                const float cbp = (st == 1) ? 1.0f / 6.0f : 1.0f / 3.0f;               // weight of k_{st-1}
#pragma unroll
                for (int r = 0; r < FMAXR; r++) {
                    const int it = tid + r * nth;
                    if (it < n_items) {
                        const int c = it / m.ns, i = it - c * m.ns;
                        const int o = c * m.ld_x + i;
                        float v = xn[r];
                        if (st > 0) {
                            const float kv = kk[o];
                            acc[r] += cbp * kv;
                            v += ca * dt * kv;
                        }
                        xs[o] = v;
                        if (tp) tp[((size_t)step * 4 + st) * n_items + it] = v;
                    }
                }
                __syncthreads();
                if (AG && m.sf) mlp_forward_split(m, wsrc, xs, A, wave, nwaves, lane,
                                         ztape ? ztape + ((size_t)blockIdx.x * n_steps * 4 + (size_t)step * 4 + st) * ((size_t)CT * zld) : nullptr, zld);
                else
                mlp_forward<false, WLDS>(m, pk, wsrc, wf, xs, nullptr, A, wave, nwaves, lane,
                                         ztape ? ztape + ((size_t)blockIdx.x * n_steps * 4 + (size_t)step * 4 + st) * ((size_t)CT * zld) : nullptr, zld);
                physics_forward(m, xs, A, F, Ri_l, bcl, ts + ca * dt, kk, tid, nth);
            }
            const bool save = (s == substeps - 1);
#pragma unroll
            for (int r = 0; r < FMAXR; r++) {
                const int it = tid + r * nth;
                if (it < n_items) {
                    const int c = it / m.ns, i = it - c * m.ns;
                    acc[r] += (1.0f / 6.0f) * kk[c * m.ld_x + i];
                    xn[r] += dt * acc[r];
                    acc[r] = 0.0f;
                    if (save && sol && col0 + c < n_col)
                        sol[((size_t)(col0 + c) * n_save + iv + 1) * m.ns + i] = xn[r];
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// loss value only (no gradient): per-tile partial sums of the six squared-error terms
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void loss_inject(const DevModel& m, const float* __restrict__ sol,
                                            const float* __restrict__ truth, size_t base, int i,
                                            const float* w6, float& lam, float* sums) {
    // base = offset of (column, save point) row; i = index inside the row
    const int Nz = m.Nz;
    const float d = sol[base + i] - truth[base + i];
    if (m.model != COLNDE_MODEL_WIND_MIXING) {
        sums[2] += d * d;
        lam += 2.0f * w6[2] * d;
        return;
    }
    const int k = i / Nz, f = i - k * Nz;
    sums[k] += d * d;
    float add = 2.0f * w6[k] * d;
    // gradient terms: Dᶠ rows 1..Nz-1 (the two zero rows only enter the mean's denominator) — loss.jl:9
    float glo = 0.0f, ghi = 0.0f;
    if (f >= 1) glo = (d - (sol[base + i - 1] - truth[base + i - 1])) * (float)Nz;
    if (f + 1 < Nz) ghi = ((sol[base + i + 1] - truth[base + i + 1]) - d) * (float)Nz;
    sums[3 + k] += glo * glo;
    add += 2.0f * w6[3 + k] * (glo - ghi) * (float)Nz;
    lam += add;
}

__device__ __forceinline__ void block_reduce_sums(float* sums, int nq, float* red, float* out, int tid, int nth) {
    // red: LDS scratch [nwaves][8]; out: global [8]
    const int lane = tid & 63, wave = tid >> 6;
    for (int q = 0; q < nq; q++) {
        float v = sums[q];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) red[wave * 8 + q] = v;
    }
    __syncthreads();
    if (tid < nq) {
        float v = 0.0f;
        for (int wv = 0; wv < (nth >> 6); wv++) v += red[wv * 8 + tid];
        out[tid] = v;
    }
}

__global__ void loss_kernel(DevModel m, const float* __restrict__ sol, const float* __restrict__ truth, int n_save,
                            int n_col, float* __restrict__ partial /* [gridDim.x][8] */) {
    __shared__ float red[16 * 8];
    const int tid = threadIdx.x, nth = blockDim.x;
    float sums[6] = {0, 0, 0, 0, 0, 0};
    const float w6[6] = {0, 0, 0, 0, 0, 0};
    const size_t rows = (size_t)n_col * n_save;
    for (size_t row = blockIdx.x; row < rows; row += gridDim.x)
        for (int i = tid; i < m.ns; i += nth) {
            float dummy = 0.0f;
            loss_inject(m, sol, truth, row * m.ns, i, w6, dummy, sums);
        }
    block_reduce_sums(sums, 6, red, partial + (size_t)blockIdx.x * 8, tid, nth);
}

// ------------------------------------------------------------------------------------------------
// adjoint: back-propagation through the RK4 steps, replaying the stage-input tape
// ------------------------------------------------------------------------------------------------
// TAPEDW: the weight gradients are not accumulated here.  Each stage's layer inputs and deltas are written to `dwtape` as
// [tile][step][stage][CT columns][xs | A of every net | dZ of every net] and contracted by dw_gemm_kernel (networks whose
// weight-gradient tiles would not fit the register file: 64-256-256-63 has 384 of them).
template <int MAXT, int NTH, int MAXR, bool WLDS, bool TAPEDW = false, bool RKC = false, bool AG = false>
__global__ void __launch_bounds__(NTH, (TAPEDW && NTH == 512) ? 4 : 1)   // taped mode, 512 threads: 128 registers, so that TWO workgroups share a CU and one's GEMMs cover the other's tape traffic
adjoint_kernel(DevModel m_arg, PackInfo pk, const float* __restrict__ w, const float* __restrict__ wf,
               const float* __restrict__ wb, const TileDesc* __restrict__ tiles, const int* __restrict__ bias_zoff,
               const int* __restrict__ bias_goff, const float* __restrict__ bcs, const float* __restrict__ save_times,
               int n_save, int substeps, const float* __restrict__ sol, const float* __restrict__ truth,
               const float* __restrict__ tape, LossWeights lw, float* __restrict__ slab /* [grid][n_params+8] */,
               int n_col, float* __restrict__ dwtape = nullptr, const float* __restrict__ ztape = nullptr) {
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nth >> 6;
    // the model description lives in LDS: per-layer fields indexed with a runtime layer number would otherwise be
    // dependent scalar loads from the kernarg segment (each followed by a full lgkmcnt(0) drain)
    // (kept inside the dynamic region: a static __shared__ object would shift its base, guide G17)
    DevModel& m_lds = *reinterpret_cast<DevModel*>(smem);
    float* const base = smem + MODEL_FLOATS;
    float* wl = base;                                   // raw weights (WLDS) + 128 floats of zero padding
    float* xs = base + (WLDS ? ((m_arg.n_params + 3) & ~3) + 128 : 0);
    float* dbar = xs + CT * m_arg.ld_x;
    float* xb = dbar + CT * m_arg.ld_x;
    static_assert(!AG || TAPEDW, "rows in global memory: the taped-dW adjoint only (the in-register one addresses its operands relative to LDS)");
    // (AG: the delta rows in the workgroup's slab of m_arg.ag; only with the Z tape, so that no A array exists — the host guarantees it)
    float* Z = AG ? m_arg.ag + (size_t)blockIdx.x * m_arg.n_nets * CT * m_arg.ld_a : xb + CT * m_arg.ld_x;
    // taped mode with the hidden pre-activations taped: the activations go from the Z tape straight into the delta tape and are never
    // needed on chip — no A array, and two workgroups of a 64-level network fit in a CU's LDS
    const bool noA = TAPEDW && ztape != nullptr;
    float* A = AG ? Z : Z + m_arg.n_nets * CT * m_arg.ld_a;
    float* gb = AG ? xb + CT * m_arg.ld_x : (noA ? A : A + m_arg.n_nets * CT * m_arg.ld_a);
    float* Ri_l = gb + 3 * CT * m_arg.ld_f;
    float* Rib_l = Ri_l + CT * m_arg.ld_f;
    float* bcl = Rib_l + CT * m_arg.ld_f;
    float* red = bcl + CT * 8;
    const int total = (int)(red + 16 * 8 - smem);
    for (int i = tid; i < total; i += nth) smem[i] = 0.0f;
    __syncthreads();
    for (int i = tid; i < (int)(sizeof(DevModel) / 4); i += nth) ((int*)&m_lds)[i] = ((const int*)&m_arg)[i];
    if (WLDS) for (int i = tid; i < m_arg.n_params; i += nth) wl[i] = w[i];
    __syncthreads();
    const DevModel& m = m_lds;
    const float* wsrc = WLDS ? wl : w;
    const int col0 = blockIdx.x * CT;
    const int n_items = CT * m.ns;
    load_bcs(m, bcs, bcl, col0, n_col, tid);

    f32x4 gacc[MAXT];
#pragma unroll
    for (int s = 0; s < MAXT; s++) gacc[s] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    float dbias[MAXB];
    int bz[MAXB];
#pragma unroll
    for (int r = 0; r < MAXB; r++) {
        dbias[r] = 0.0f;
        const int b = tid + r * nth;
        bz[r] = b < m.n_bias ? bias_zoff[b] : -1;
    }
    float lam[MAXR], xbs[MAXR], xpre[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; r++) lam[r] = 0.0f;
    float sums[6] = {0, 0, 0, 0, 0, 0};
    STAMP_DECL;

    // per-slot LDS addresses (float indices into smem, lane part folded in) of this wave's weight-gradient tiles
    int a_ad[MAXT], d_ad[MAXT];
    unsigned long long a_is_act = 0ull;
    if (!TAPEDW) {
        const int cq = lane >> 4;
#pragma unroll
        for (int sl = 0; sl < MAXT; sl++) {
            const int t = wave + sl * nwaves;
            a_ad[sl] = 0;
            d_ad[sl] = 0;
            if (t < m.n_tiles) {
                const TileDesc d = tiles[t];
                const int stride = d.a_src ? m.ld_a : m.ld_x;
                a_ad[sl] = (int)((d.a_src ? A + d.net * CT * m.ld_a : xs) - smem) + d.a_off + (lane & 15) + cq * stride;
                d_ad[sl] = (int)(Z - smem) + d.net * CT * m.ld_a + d.d_off + (lane & 15) + cq * m.ld_a;
                if (d.a_src) a_is_act |= 1ull << sl;
            }
        }
    }

    const int n_steps = (n_save - 1) * substeps;
    const int nst = m.nst;                               // taped stage inputs per step: 4 (RK4) or s (RKC2)
    constexpr bool rkc = RKC;            // a template parameter: the RK4 instantiations do not carry the RKC cotangent registers
    const float* mu_t = m.rkc, *nu_t = m.rkc + RKC_LD, *mut_t = m.rkc + 2 * RKC_LD, *gat_t = m.rkc + 3 * RKC_LD, *kap_t = m.rkc + 5 * RKC_LD;
    const float* tp = tape + (size_t)blockIdx.x * n_steps * nst * n_items;
    // the tape is read one stage ahead of its use (xpre) so that its HBM latency hides under the previous stage
#pragma unroll
    for (int r = 0; r < MAXR; r++) {
        const int it = tid + r * nth;
        xpre[r] = it < n_items ? tp[((size_t)n_steps * nst - 1) * n_items + it] : 0.0f;
    }
    // RKC2 (discrete adjoint of the recurrence in colnde_dev.h): cotangents of Y_j (lam), Y_{j-1} (yb1), Y_{j-2} (yb2), Y_0 (yb0)
    // and F_0 (f0b) per owner item; unused by the RK4 path
    float yb1[MAXR], yb2[MAXR], yb0[MAXR], f0b[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; r++) { yb1[r] = 0.0f; yb2[r] = 0.0f; yb0[r] = 0.0f; f0b[r] = 0.0f; }

    // save point 0 enters the loss value only (x0 does not depend on the weights)
#pragma unroll
    for (int r = 0; r < MAXR; r++) {
        const int it = tid + r * nth;
        if (it < n_items) {
            const int c = it / m.ns, i = it - c * m.ns;
            if (col0 + c < n_col) {
                float dummy = 0.0f;
                loss_inject(m, sol, truth, ((size_t)(col0 + c) * n_save) * m.ns, i, lw.w, dummy, sums);
            }
        }
    }

    for (int iv = n_save - 2; iv >= 0; iv--) {
        const float t0 = save_times[iv];
        const float dt = (save_times[iv + 1] - t0) / (float)substeps;
        // λ += ∂loss/∂sol[:, iv+1]
#pragma unroll
        for (int r = 0; r < MAXR; r++) {
            const int it = tid + r * nth;
            if (it < n_items) {
                const int c = it / m.ns, i = it - c * m.ns;
                if (col0 + c < n_col)
                    loss_inject(m, sol, truth, ((size_t)(col0 + c) * n_save + iv + 1) * m.ns, i, lw.w, lam[r], sums);
            }
        }
        for (int s = substeps - 1; s >= 0; s--) {
            const int step = iv * substeps + s;
#pragma unroll
            for (int r = 0; r < MAXR; r++) xbs[r] = 0.0f;
#pragma nounroll
            for (int st = nst - 1; st >= 0; st--) {
                // k̄4 = dt/6 λ; k̄3 = dt/3 λ + dt x̄4; k̄2 = dt/3 λ + dt/2 x̄3; k̄1 = dt/6 λ + dt/2 x̄2
                const float cwl = (st == 0 || st == 3) ? dt / 6.0f : dt / 3.0f;
                const float cwx = st == 2 ? dt : 0.5f * dt;
                // RKC2, stage input Y_st feeds Y_j, j = st + 1, through mu~_j h F_st:
                const float cmu = rkc ? mu_t[st + 1] : 0.0f, cnu = rkc ? nu_t[st + 1] : 0.0f;
                const float cmt = rkc ? mut_t[st + 1] * dt : 0.0f, cga = rkc ? gat_t[st + 1] * dt : 0.0f, ck0 = rkc ? kap_t[st + 1] : 0.0f;
                STAMP_BEGIN();
                // stage input from the tape; stage cotangent k̄_st = wl λ + wx x̄_{st+1}
#pragma unroll
                for (int r = 0; r < MAXR; r++) {
                    const int it = tid + r * nth;
                    if (it < n_items) {
                        const int c = it / m.ns, i = it - c * m.ns;
                        const int o = c * m.ld_x + i;
                        xs[o] = xpre[r];
                        float kb;
                        if (rkc) {
                            // lam = cotangent of Y_j, complete once the previous iteration's pullback (xb: J(Y_j)^T F̄_j) is added
                            if (st < nst - 1) {
                                const float yj = yb1[r] + xb[o];
                                yb1[r] = yb2[r];
                                yb2[r] = 0.0f;
                                lam[r] = yj;
                            }
                            if (st >= 1) {
                                yb0[r] += ck0 * lam[r];
                                yb1[r] += cmu * lam[r];
                                yb2[r] += cnu * lam[r];
                                f0b[r] += cga * lam[r];
                                kb = cmt * lam[r];
                            } else {
                                // Y_1 = Y_0 + mu~_1 h F_0: lam holds Ȳ_1, yb1 the nu_2 part of Ȳ_0
                                yb0[r] += lam[r] + yb1[r];
                                kb = f0b[r] + cmt * lam[r];
                                yb1[r] = 0.0f;
                                f0b[r] = 0.0f;
                            }
                        } else {
                            kb = cwl * lam[r];
                            if (st < 3) kb += cwx * xb[o];
                        }
                        dbar[o] = kb;
                    }
                }
                {
                    const int qn = step * nst + st - 1;            // the stage handled next (tape order is time order)
                    if (qn >= 0) {
#pragma unroll
                        for (int r = 0; r < MAXR; r++) {
                            const int it = tid + r * nth;
                            if (it < n_items) xpre[r] = tp[(size_t)qn * n_items + it];
                        }
                    }
                }
                __syncthreads();
                STAMP(0);
                if (TAPEDW && ztape) {
                    // hidden-layer pre-activations taped by the forward kernel: Z into LDS, A = act(Z) straight into this stage's record of
                    // the delta tape (the output layer enters the pullback linearly: its values are not needed; its A slot is never read
                    // by the dW GEMM either)
                    const int hid = m.act_off[m.n_layers - 1], zld = (m.n_nets * hid + 3) & ~3;
                    const float* zr = ztape + ((size_t)blockIdx.x * n_steps * nst + (size_t)step * nst + st) * ((size_t)CT * zld);
                    const int ns4r = (m.ns + 3) & ~3, act4r = (m.act_total + 3) & ~3;
                    const int Rr = ns4r + 2 * m.n_nets * act4r;
                    float* recA = dwtape + ((size_t)blockIdx.x * n_steps * nst + (size_t)step * nst + st) * ((size_t)CT * Rr) + ns4r;
                    // (segments start at multiples of 4 floats: one float4 never straddles two layers; up to four float4 per lane
                    //  are fetched back to back so that one HBM latency covers them)
                    const int nq = (m.n_nets * hid) >> 2;                      // float4 items per column
                    for (int c = wave; c < CT; c += nwaves)
                        for (int q0 = lane; q0 < nq; q0 += 4 * 64) {
                            float4 v[4];
#pragma unroll
                            for (int u = 0; u < 4; u++)
                                if (q0 + 64 * u < nq) v[u] = *reinterpret_cast<const float4*>(zr + c * zld + 4 * (q0 + 64 * u));
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                const int q = q0 + 64 * u;
                                if (q < nq) {
                                    const int f = 4 * q, net = f / hid, o = f - net * hid;
                                    int l = 0;
                                    while (o >= m.act_off[l + 1]) l++;
                                    const int a = m.acts[l];
                                    // the pad slots behind a layer's last feature were never written by the forward kernel: they must read as
                                    // zero (a backward chain multiplies them with whatever weight sits behind the row's end)
                                    const int nvalid = m.sizes[l + 1] - (o - m.act_off[l]);
                                    if (nvalid < 4) {
                                        if (nvalid < 1) v[u].x = 0.0f;
                                        if (nvalid < 2) v[u].y = 0.0f;
                                        if (nvalid < 3) v[u].z = 0.0f;
                                        v[u].w = 0.0f;
                                    }
                                    float* zd = Z + (net * CT + c) * m.ld_a + o;
                                    zd[0] = v[u].x; zd[1] = v[u].y; zd[2] = v[u].z; zd[3] = v[u].w;
                                    *reinterpret_cast<float4*>(recA + (size_t)c * Rr + net * act4r + o) =
                                        make_float4(dev_act(a, v[u].x), dev_act(a, v[u].y), dev_act(a, v[u].z), dev_act(a, v[u].w));
                                }
                            }
                        }
                    __syncthreads();
                } else
                    mlp_forward<true, WLDS>(m, pk, wsrc, wf, xs, Z, A, wave, nwaves, lane);
                STAMP(1);
                physics_vjp(m, xs, dbar, Z, xb, gb, Ri_l, Rib_l, tid, nth, rkc ? (st == nst - 1 ? 1 : 2) : 0);
                STAMP(2);
                if (AG && m.sb) mlp_backward_split(m, pk, wb, Z, xb, wave, nwaves, lane);
                else mlp_backward<WLDS>(m, pk, wsrc, wb, Z, xb, wave, nwaves, lane);
                STAMP(3);
                // weight gradients: dW += A_{l-1}^T dZ_l over the tile's 16 columns; operands of slot sl+1 are read
                // while the MFMAs of slot sl run
                if (TAPEDW) {
                    // row = [xs (ns4) | A of every net (act4 each) | dZ of every net (act4 each)], every segment padded to a multiple of
                    // 4 floats (the pads copy LDS row padding: finite, never stored by dw_gemm).  LDS rows are 8-byte aligned (stride
                    // == 2 mod 4), tape rows 16-byte aligned: copied two floats at a time
                    const int ns4 = (m.ns + 3) & ~3, act4 = (m.act_total + 3) & ~3;
                    const int R = ns4 + 2 * m.n_nets * act4;
                    float* rec = dwtape + ((size_t)blockIdx.x * n_steps * nst + (size_t)step * nst + st) * ((size_t)CT * R);
                    const int hx = ns4 >> 1, ha = act4 >> 1;                                        // float2 items per segment
                    for (int c = wave; c < CT; c += nwaves) {                                       // one wave per column
                        float* row = rec + (size_t)c * R;
                        for (int q = lane; q < hx; q += 64)
                            *reinterpret_cast<float2*>(row + 2 * q) = *reinterpret_cast<const float2*>(xs + c * m.ld_x + 2 * q);
                        for (int seg = noA ? m.n_nets : 0; seg < 2 * m.n_nets; seg++) {               // A of every net (unless already written), then dZ of every net
                            const int net = seg < m.n_nets ? seg : seg - m.n_nets;
                            const float* src = (seg < m.n_nets ? A : Z) + (net * CT + c) * m.ld_a;
                            float* dst = row + ns4 + seg * act4;
                            for (int o = lane; o < ha; o += 64)
                                *reinterpret_cast<float2*>(dst + 2 * o) = *reinterpret_cast<const float2*>(src + 2 * o);
                        }
                    }
                } else {
                    float pa[2][4], pb[2][4];
                    const int zs4 = 4 * m.ld_a;
                    if (wave < m.n_tiles) {
                        const int as4 = 4 * ((a_is_act & 1ull) ? m.ld_a : m.ld_x);
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            pa[0][q] = smem[a_ad[0] + q * as4];
                            pb[0][q] = smem[d_ad[0] + q * zs4];
                        }
                    }
#pragma unroll
                    for (int sl = 0; sl < MAXT; sl++) {
                        if (sl + 1 < MAXT && wave + (sl + 1) * nwaves < m.n_tiles) {
                            const int as4 = 4 * (((a_is_act >> (sl + 1)) & 1ull) ? m.ld_a : m.ld_x);
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                pa[(sl + 1) & 1][q] = smem[a_ad[(sl + 1) % MAXT] + q * as4];
                                pb[(sl + 1) & 1][q] = smem[d_ad[(sl + 1) % MAXT] + q * zs4];
                            }
                        }
                        if (wave + sl * nwaves < m.n_tiles) {
#pragma unroll
                            for (int q = 0; q < 4; q++) gacc[sl] = mfma16(pa[sl & 1][q], pb[sl & 1][q], gacc[sl]);
                        }
                    }
                }
                STAMP(4);
#pragma unroll
                for (int r = 0; r < MAXB; r++)
                    if (bz[r] >= 0) {
                        float sacc = 0.0f;
#pragma unroll
                        for (int c = 0; c < CT; c++) sacc += Z[bz[r] + c * m.ld_a];
                        dbias[r] += sacc;
                    }
#pragma unroll
                for (int r = 0; r < MAXR; r++) {
                    const int it = tid + r * nth;
                    if (it < n_items) {
                        const int c = it / m.ns, i = it - c * m.ns;
                        xbs[r] += xb[c * m.ld_x + i];
                    }
                }
                __syncthreads();
                STAMP(5);
            }
            if (rkc) {
                // λ_n = Ȳ_0 + J(Y_0)^T F̄_0 (xb of the st = 0 pullback; visible after the stage's closing barrier)
#pragma unroll
                for (int r = 0; r < MAXR; r++) {
                    const int it = tid + r * nth;
                    if (it < n_items) {
                        const int c = it / m.ns, i = it - c * m.ns;
                        lam[r] = yb0[r] + xb[c * m.ld_x + i];
                    }
                    yb0[r] = 0.0f;
                }
            } else {
#pragma unroll
                for (int r = 0; r < MAXR; r++) lam[r] += xbs[r];
            }
        }
    }

    STAMP_FLUSH();
    // flush this tile's partial gradient and loss sums
    float* out = slab + (size_t)blockIdx.x * (m.n_params + 8);
#pragma unroll
    for (int sl = 0; sl < MAXT; sl++) {
        const int t = wave + sl * nwaves;
        if (!TAPEDW && t < m.n_tiles) {
            const TileDesc d = tiles[t];
            const int j = lane & 15;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = 4 * (lane >> 4) + r;
                if (i < d.ni_rem && j < d.no_rem) out[d.g_off + i * d.no + j] = gacc[sl][r];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < MAXB; r++) {
        const int b = tid + r * nth;
        if (b < m.n_bias) out[bias_goff[b]] = dbias[r];
    }
    block_reduce_sums(sums, 6, red, out + m.n_params, tid, nth);
    if (tid >= 6 && tid < 8) out[m.n_params + tid] = 0.0f;
}

// ------------------------------------------------------------------------------------------------
// taped weight gradients: dW[i][j] = Σ_{records, columns} a[c][i] · dz[c][j], a split-K GEMM over the rows taped by
// adjoint_kernel<..., TAPEDW>.  The contracted index (column, stage) is the row index of the tape, so both MFMA operands
// are read from HBM/L2 already in operand layout (lane = feature, 128-byte segments): no LDS, no transposition.
// One wavefront owns one 64x64 block (2 x 2 tiles of v_mfma_f32_32x32x2_f32) for one slice of the records; the four
// waves of a workgroup hold consecutive blocks (same layer: shared input rows hit L1).  Workgroups that share a slice
// are placed on the same XCD (id mod 8) so that the slice streams through that XCD's L2 once.
// ------------------------------------------------------------------------------------------------
typedef float dwf32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256)
dw_gemm_kernel(const float* __restrict__ dwtape, size_t n_records, int R, const DwMacro* __restrict__ macros, int n_macros,
               int n_groups, int n_slices, float* __restrict__ slab_rows, int stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L = blockIdx.x, xcd = L & 7, k = L >> 3;
    const int group = k % n_groups, slice = (k / n_groups) * 8 + xcd;
    const int mi = group * 4 + wave;
    if (slice >= n_slices || mi >= n_macros) return;
    const DwMacro mc = macros[mi];
    const size_t per = (n_records + n_slices - 1) / n_slices;
    const size_t r0 = (size_t)slice * per, r1 = r0 + per < n_records ? r0 + per : n_records;
    const int f = lane & 31, kk = lane >> 5;
    const bool a_hi = mc.ni_rem > 32, d_hi = mc.no_rem > 32;
    const int fa0 = min(mc.a_feat + f, R - 1), fa1 = min(mc.a_feat + 32 + f, R - 1);
    const int fd0 = min(mc.d_feat + f, R - 1), fd1 = min(mc.d_feat + 32 + f, R - 1);
    dwf32x16 acc00 = (dwf32x16)(0.0f), acc01 = acc00, acc10 = acc00, acc11 = acc00;
    for (size_t r = r0; r < r1; r++) {
        const float* rec = dwtape + r * ((size_t)CT * R) + (size_t)kk * R;
        float a0[8], a1[8], d0[8], d1[8];
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const float* row = rec + (size_t)(2 * s) * R;
            a0[s] = row[fa0];
            d0[s] = row[fd0];
            a1[s] = a_hi ? row[fa1] : 0.0f;
            d1[s] = d_hi ? row[fd1] : 0.0f;
        }
#pragma unroll
        for (int s = 0; s < 8; s++) {
            acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], d0[s], acc00, 0, 0, 0);
            if (d_hi) acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], d1[s], acc01, 0, 0, 0);
            if (a_hi) acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], d0[s], acc10, 0, 0, 0);
            if (a_hi && d_hi) acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], d1[s], acc11, 0, 0, 0);
        }
    }
    // D layout: lane (j = lane & 31, h = lane >> 5), register r -> row i = 8 (r / 4) + 4 h + (r % 4)
    float* out = slab_rows + (size_t)slice * stride + mc.g_off;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int i = 8 * (r >> 2) + 4 * kk + (r & 3);
        if (i < mc.ni_rem && f < mc.no_rem) out[(size_t)i * mc.no + f] = acc00[r];
        if (i < mc.ni_rem && f + 32 < mc.no_rem) out[(size_t)i * mc.no + f + 32] = acc01[r];
        if (i + 32 < mc.ni_rem && f < mc.no_rem) out[(size_t)(i + 32) * mc.no + f] = acc10[r];
        if (i + 32 < mc.ni_rem && f + 32 < mc.no_rem) out[(size_t)(i + 32) * mc.no + f + 32] = acc11[r];
    }
}

// LDS-staged variant: one workgroup streams its slice of records through two LDS buffers (global_load_lds, 16 bytes per lane: no
// register staging) and its DW_NW waves contract up to DW_MAXM blocks each from LDS, so the tape is read from HBM exactly once.
// Chosen when two records fit in LDS and the network has at most DW_NW * DW_MAXM blocks (64-256-256-63: 24 blocks, 77.7 KB records).
#define DW_MAXM 3          // blocks per wave: 3 x 64 accumulator registers leave room for two waves per SIMD
#define DW_NW 8            // waves per workgroup

// Round 3: (a) the operands of k-step t + 1 are read from LDS BEFORE the four MFMAs of step t are issued (the straight compiler schedule
// issued eight reads after each eight MFMAs and waited for them with the pipe draining: 72 % busy, profiles/r03g_cs_table.csv); (b) MAXM is a
// template parameter chosen from the number of blocks (a 32-level network has 8: one per wave, not three slots of which two are empty);
// (c) small records are staged RPB at a time, so that a barrier closes at least ~6 k cycles of MFMA work.
template <int MAXM, int NW, int RPB>
__global__ void __launch_bounds__(64 * NW)
dw_gemm_lds_kernel(const float* __restrict__ dwtape, size_t n_records, int R, const DwMacro* __restrict__ macros, int n_macros,
                   int n_slices, float* __restrict__ slab_rows, int stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slice = blockIdx.x;
    const int rec_floats = CT * R;                       // multiple of 4 (checked by the host)
    const int stage_floats = RPB * rec_floats;           // one LDS buffer: RPB consecutive records
    const size_t per = (n_records + n_slices - 1) / n_slices;
    const size_t r0 = (size_t)slice * per, r1 = r0 + per < n_records ? r0 + per : n_records;
    const int f = lane & 31, kk = lane >> 5;
    DwMacro mc[MAXM];
    int fa0[MAXM], fa1[MAXM], fd0[MAXM], fd1[MAXM];
    dwf32x16 acc[MAXM][4];
#pragma unroll
    for (int q = 0; q < MAXM; q++) {
        const int mi = wave + NW * q;
        if (mi < n_macros) mc[q] = macros[mi];
        else { mc[q].a_feat = 0; mc[q].d_feat = 0; mc[q].ni_rem = 0; mc[q].no_rem = 0; mc[q].g_off = 0; mc[q].no = 1; }
        fa0[q] = min(mc[q].a_feat + f, R - 1) + kk * R;
        fa1[q] = min(mc[q].a_feat + 32 + f, R - 1) + kk * R;
        fd0[q] = min(mc[q].d_feat + f, R - 1) + kk * R;
        fd1[q] = min(mc[q].d_feat + 32 + f, R - 1) + kk * R;
#pragma unroll
        for (int t = 0; t < 4; t++) acc[q][t] = (dwf32x16)(0.0f);
    }
    // asynchronous copy of up to RPB records into an LDS buffer: 1 KB per wave instruction, lanes contiguous
    // (issuing a stage's copy in slices between the blocks of the stage before it was measured slower: 56.4 -> 67.3 ms, profiles/r03s_gemm_spread.log)
    auto fetch = [&](size_t r, int dst /* float offset of the buffer inside smem */) {
        const size_t nrec = r1 - r < (size_t)RPB ? r1 - r : (size_t)RPB;
        const int nfl = (int)nrec * rec_floats;
        const float* src = dwtape + r * (size_t)rec_floats;
        for (int o = wave * 256; o < nfl; o += NW * 256)
            if (o + lane * 4 < nfl) __builtin_amdgcn_global_load_lds(src + o + lane * 4, smem + dst + o, 16, 0, 0);
    };
    if (r0 < r1) fetch(r0, 0);
    __syncthreads();
    int cur = 0;                                         // float offset of the buffer being contracted
    for (size_t r = r0; r < r1; r += RPB) {
        if (r + RPB < r1) fetch(r + RPB, stage_floats - cur);
        const int nv = (int)(r1 - r < (size_t)RPB ? r1 - r : (size_t)RPB);
#pragma unroll
        for (int rr = 0; rr < RPB; rr++) {
            if (rr >= nv) break;                         // wave-uniform: the last stage of a slice may hold fewer records
            const float* rec = smem + cur + rr * rec_floats;
            // straight-line over the wave's MAXM blocks x 8 k-steps (2 columns each); all four 32x32 tiles of a block are issued (a block
            // narrower than 64 reads clamped, in-range features whose products are never stored; an empty slot likewise)
            constexpr int T = MAXM * 8;
            float op[2][4];
            auto load = [&](int t, float (&o)[4]) {
                const int q = t >> 3, sgrp = t & 7;
                const float* row = rec + 2 * sgrp * R;
                o[0] = row[fa0[q]]; o[1] = row[fd0[q]]; o[2] = row[fa1[q]]; o[3] = row[fd1[q]];
            };
            load(0, op[0]);
#pragma unroll
            for (int t = 0; t < T; t++) {
                if (t + 1 < T) load(t + 1, op[(t + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const int q = t >> 3;
                const float a0 = op[t & 1][0], d0 = op[t & 1][1], a1 = op[t & 1][2], d1 = op[t & 1][3];
                acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, d0, acc[q][0], 0, 0, 0);
                acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, d1, acc[q][1], 0, 0, 0);
                acc[q][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, d0, acc[q][2], 0, 0, 0);
                acc[q][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, d1, acc[q][3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();          // every wave is done with this buffer, and (vmcnt drain) the next stage has landed
        cur = stage_floats - cur;
    }
#pragma unroll
    for (int q = 0; q < MAXM; q++) {
        if (mc[q].ni_rem == 0) continue;
        float* out = slab_rows + (size_t)slice * stride + mc[q].g_off;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int i = 8 * (r >> 2) + 4 * kk + (r & 3);
            if (i < mc[q].ni_rem && f < mc[q].no_rem) out[(size_t)i * mc[q].no + f] = acc[q][0][r];
            if (i < mc[q].ni_rem && f + 32 < mc[q].no_rem) out[(size_t)i * mc[q].no + f + 32] = acc[q][1][r];
            if (i + 32 < mc[q].ni_rem && f < mc[q].no_rem) out[(size_t)(i + 32) * mc[q].no + f] = acc[q][2][r];
            if (i + 32 < mc[q].ni_rem && f + 32 < mc[q].no_rem) out[(size_t)(i + 32) * mc[q].no + f + 32] = acc[q][3][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dW GEMM with exact three-way bf16 operand splitting (COLNDE_MATRIX_BF16X3_EXACT).  One workgroup of 8 waves per slice of records, as above, but the record
// does not go to LDS as floats: every thread loads its share (two adjacent features x eight columns, 8-byte loads, two records ahead in
// registers), splits the values and writes three bf16 planes [feature][column half][8 bf16] — MFMA-operand order, so a block's operands
// are 12 ds_read_b128 per record and nothing is split twice.  Two plane buffers, ONE bare barrier per record (a wave that passed the
// barrier of record r has finished the products of record r - 1, whose buffer record r + 1 overwrites).  A pass holds the operand
// features of some layers only (compact order, DwPassDesc), so the planes and 2 x 64 accumulator registers per wave fit; the passes
// together read each tape float once.
// ------------------------------------------------------------------------------------------------
#define DWS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
typedef float dws_f32x2 __attribute__((ext_vector_type(2)));

template <int MAXM, int NIT>
__global__ void __launch_bounds__(512)
dw_gemm_split_kernel(const float* __restrict__ dwtape, size_t n_records, int R, const DwMacro* __restrict__ macros, DwPassDesc ps,
                     int n_slices, float* __restrict__ slab_rows, int stride) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slice = blockIdx.x;
    const size_t per = (n_records + n_slices - 1) / n_slices;
    const size_t r0 = (size_t)slice * per, r1 = r0 + per < n_records ? r0 + per : n_records;
    if (r0 >= r1) return;                                  // (whole workgroup)
    const int Fc = ps.Fc, FcP = Fc + 64;
    u32x4* planes = reinterpret_cast<u32x4*>(smem);        // [2 buffers][3 planes][FcP features][2 column halves]
    // pad features (read by the clamped tiles of narrow blocks, never stored) must be finite
    for (int e = tid; e < 2 * 3 * 64 * 2; e += 512) {
        const int kh = e & 1, f = (e >> 1) & 63, bp = e >> 7;
        planes[((size_t)bp * FcP + Fc + f) * 2 + kh] = (u32x4)(0u);
    }
    // loader role: item `it` = (feature pair j, column half kh), j fastest (coalesced 8-byte loads)
    int src_off[NIT], cfeat[NIT], ikh[NIT];
    bool live[NIT];
#pragma unroll
    for (int k = 0; k < NIT; k++) {
        const int it = tid + 512 * k, np = Fc >> 1;
        live[k] = it < Fc;
        const int j = live[k] ? it % np : 0;
        ikh[k] = live[k] ? it / np : 0;
        cfeat[k] = 2 * j;
        int so = 0;
        for (int sg = 0; sg < ps.n_seg; sg++)
            if (cfeat[k] >= ps.seg[sg].dst && cfeat[k] < ps.seg[sg].dst + ps.seg[sg].len) so = ps.seg[sg].src + cfeat[k] - ps.seg[sg].dst;
        src_off[k] = so + 8 * ikh[k] * R;
    }
    const int f = lane & 31, kk = lane >> 5;
    DwMacro mc[MAXM];
    sp_f32x16 acc[MAXM][4];
#pragma unroll
    for (int q = 0; q < MAXM; q++) {
        const int mi = wave + 8 * q;
        if (mi < ps.n_macros) mc[q] = macros[ps.m0 + mi];
        else { mc[q].a_feat = 0; mc[q].d_feat = 0; mc[q].ni_rem = 0; mc[q].no_rem = 0; mc[q].g_off = 0; mc[q].no = 1; }
#pragma unroll
        for (int t = 0; t < 4; t++) acc[q][t] = (sp_f32x16)(0.0f);
    }
    dws_f32x2 pre[2][NIT][8];
    auto gl = [&](size_t r, dws_f32x2 (&dst)[NIT][8]) {
        const size_t rc = r < r1 ? r : r1 - 1;             // past the end: the last record again (loaded, never used)
        const float* rec = dwtape + rc * ((size_t)CT * R);
#pragma unroll
        for (int k = 0; k < NIT; k++)
#pragma unroll
            for (int c = 0; c < 8; c++) dst[k][c] = *reinterpret_cast<const dws_f32x2*>(rec + src_off[k] + c * R);
    };
    auto sp = [&](const dws_f32x2 (&src)[NIT][8], int buf) {
#pragma unroll
        for (int k = 0; k < NIT; k++) {
            if (!live[k]) continue;
#pragma unroll
            for (int z = 0; z < 2; z++) {
                float x8[8];
#pragma unroll
                for (int c = 0; c < 8; c++) x8[c] = src[k][c][z];
                const Bf3 b = bf3_split8(x8);
                u32x4* o = planes + ((size_t)(buf * 3) * FcP + cfeat[k] + z) * 2 + ikh[k];
                o[0] = b.h;
                o[(size_t)FcP * 2] = b.m;
                o[(size_t)FcP * 4] = b.l;
            }
        }
    };
    auto ld = [&](int buf, int cf) {
        const u32x4* o = planes + ((size_t)(buf * 3) * FcP + cf + f) * 2 + kk;
        Bf3 b;
        b.h = o[0];
        b.m = o[(size_t)FcP * 2];
        b.l = o[(size_t)FcP * 4];
        return b;
    };
    // (wave-uniform) the wave's two blocks are full 64 x 64 blocks of the same output columns: their d operands are read once
    const bool shared_d = MAXM == 2 && mc[0].ni_rem > 32 && mc[MAXM - 1].ni_rem > 32 && mc[0].no_rem > 32 && mc[MAXM - 1].no_rem > 32 &&
                          mc[0].d_feat == mc[MAXM - 1].d_feat;
    auto products = [&](int buf) {
        if (MAXM == 2 && shared_d) {
            // four a tiles against the same two d tiles: the next a tile is read while the current one is multiplied
            const Bf3 d0 = ld(buf, mc[0].d_feat), d1 = ld(buf, mc[0].d_feat + 32);
            Bf3 A = ld(buf, mc[0].a_feat);
#pragma unroll
            for (int t = 0; t < 2 * MAXM; t++) {
                const Bf3 Ac = A;
                if (t + 1 < 2 * MAXM) A = ld(buf, mc[(t + 1) >> 1].a_feat + 32 * ((t + 1) & 1));
                acc[t >> 1][2 * (t & 1)] = mfma_bf3(Ac, d0, acc[t >> 1][2 * (t & 1)]);
                acc[t >> 1][2 * (t & 1) + 1] = mfma_bf3(Ac, d1, acc[t >> 1][2 * (t & 1) + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < MAXM; q++) {
            if (mc[q].ni_rem == 0) continue;               // (wave-uniform) an empty slot
            const Bf3 a0 = ld(buf, mc[q].a_feat), d0 = ld(buf, mc[q].d_feat);
            acc[q][0] = mfma_bf3(a0, d0, acc[q][0]);
            if (mc[q].no_rem > 32) {
                const Bf3 d1 = ld(buf, mc[q].d_feat + 32);
                acc[q][1] = mfma_bf3(a0, d1, acc[q][1]);
                if (mc[q].ni_rem > 32) {
                    const Bf3 a1 = ld(buf, mc[q].a_feat + 32);
                    acc[q][2] = mfma_bf3(a1, d0, acc[q][2]);
                    acc[q][3] = mfma_bf3(a1, d1, acc[q][3]);
                }
            } else if (mc[q].ni_rem > 32) {
                const Bf3 a1 = ld(buf, mc[q].a_feat + 32);
                acc[q][2] = mfma_bf3(a1, d0, acc[q][2]);
            }
        }
    };
    // Between two barriers a wave multiplies record r (plane buffer r & 1) AND splits record r + 1 into the other buffer (free since the last
    // barrier: everybody has finished record r - 1).  The two waves of a SIMD (w, w + 4) take the two jobs in opposite order, so that one's
    // vector work and LDS round trips run under the other's MFMAs instead of both queueing for the matrix pipe at the same moment.
    const size_t n = r1 - r0;
    const bool split_first = (wave >> 2) & 1;
    gl(r0, pre[0]);
    gl(r0 + 1, pre[1]);
    sp(pre[0], 0);
    gl(r0 + 2, pre[0]);
    DWS_BARRIER();
    for (size_t i = 0; i < n; i += 2) {
        if (split_first) { sp(pre[1], 1); gl(r0 + i + 3, pre[1]); products(0); }
        else { products(0); sp(pre[1], 1); gl(r0 + i + 3, pre[1]); }
        DWS_BARRIER();
        if (i + 1 < n) {
            if (split_first) { sp(pre[0], 0); gl(r0 + i + 4, pre[0]); products(1); }
            else { products(1); sp(pre[0], 0); gl(r0 + i + 4, pre[0]); }
            DWS_BARRIER();
        }
    }
#pragma unroll
    for (int q = 0; q < MAXM; q++) {
        if (mc[q].ni_rem == 0) continue;
        float* out = slab_rows + (size_t)slice * stride + mc[q].g_off;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int i = 8 * (r >> 2) + 4 * kk + (r & 3);
            if (i < mc[q].ni_rem && f < mc[q].no_rem) out[(size_t)i * mc[q].no + f] = acc[q][0][r];
            if (i < mc[q].ni_rem && f + 32 < mc[q].no_rem) out[(size_t)i * mc[q].no + f + 32] = acc[q][1][r];
            if (i + 32 < mc[q].ni_rem && f < mc[q].no_rem) out[(size_t)(i + 32) * mc[q].no + f] = acc[q][2][r];
            if (i + 32 < mc[q].ni_rem && f + 32 < mc[q].no_rem) out[(size_t)(i + 32) * mc[q].no + f + 32] = acc[q][3][r];
        }
    }
}

// Deal the blocks to passes (first fit, largest layer first): a pass = the blocks of some (net, layer) groups whose operand features, merged
// into at most 8 even-aligned segments of the record row, are at most 768 (two plane buffers in LDS) and whose blocks are at most 16.
bool dw_split_build(const std::vector<DwMacro>& mac, const std::vector<int>& matrix_of, int R, DwSplitPlan& plan) {
    struct Group { std::vector<int> idx; int feats; };
    std::vector<Group> groups;                 // the blocks of one weight matrix (net, layer) stay together
    for (int i = 0; i < (int)mac.size(); i++) {
        if (groups.empty() || matrix_of[i] != matrix_of[groups.back().idx[0]]) groups.push_back(Group{{}, 0});
        groups.back().idx.push_back(i);
    }
    auto segments = [&](const std::vector<int>& idx, std::vector<DwSeg>& segs) {
        std::vector<std::pair<int, int>> iv;
        for (int i : idx) {
            iv.push_back({mac[i].a_feat & ~1, std::min(R, (mac[i].a_feat + mac[i].ni_rem + 1) & ~1)});
            iv.push_back({mac[i].d_feat & ~1, std::min(R, (mac[i].d_feat + mac[i].no_rem + 1) & ~1)});
        }
        std::sort(iv.begin(), iv.end());
        segs.clear();
        int dst = 0;
        for (auto& v : iv) {
            if (!segs.empty() && v.first <= segs.back().src + segs.back().len) {
                const int end = std::max(segs.back().src + segs.back().len, v.second);
                dst += end - (segs.back().src + segs.back().len);
                segs.back().len = end - segs.back().src;
            } else {
                segs.push_back(DwSeg{v.first, v.second - v.first, dst});
                dst += v.second - v.first;
            }
        }
        return dst;
    };
    // A matrix too large for one pass (the wide wind-mixing layers: 400 x 400 = 49 blocks, 800 features) is cut along its OUTPUT blocks into chunks that fit
    // (<= 16 blocks, <= 768 compact features): every chunk keeps all input features of the matrix and takes a run of output-block columns.
    {
        std::vector<Group> cut;
        for (auto& g : groups) {
            std::vector<DwSeg> sg;
            if ((int)g.idx.size() <= 16 && segments(g.idx, sg) <= 768 && sg.size() <= 8) { cut.push_back(g); continue; }
            std::vector<int> dcols;                                  // distinct output-block columns (d_feat) of this matrix, in order
            for (int i : g.idx)
                if (std::find(dcols.begin(), dcols.end(), mac[i].d_feat) == dcols.end()) dcols.push_back(mac[i].d_feat);
            std::sort(dcols.begin(), dcols.end());
            size_t c0 = 0;
            while (c0 < dcols.size()) {
                Group best{{}, 0};
                for (size_t c1 = c0 + 1; c1 <= dcols.size(); c1++) {
                    Group trial{{}, 0};
                    for (int i : g.idx)
                        if (mac[i].d_feat >= dcols[c0] && mac[i].d_feat <= dcols[c1 - 1]) trial.idx.push_back(i);
                    std::vector<DwSeg> st;
                    if ((int)trial.idx.size() > 16 || segments(trial.idx, st) > 768 || st.size() > 8) break;
                    best = trial;
                }
                if (best.idx.empty()) return false;                   // one output-block column alone does not fit
                const int last = mac[best.idx.back()].d_feat;
                cut.push_back(best);
                while (c0 < dcols.size() && dcols[c0] <= last) c0++;
            }
        }
        groups.swap(cut);
    }
    for (auto& g : groups) { std::vector<DwSeg> sg; g.feats = segments(g.idx, sg); }
    std::vector<int> order(groups.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return groups[a].feats > groups[b].feats; });
    std::vector<std::vector<int>> bins;        // macro indices per pass
    for (int gi : order) {
        bool placed = false;
        for (auto& b : bins) {
            std::vector<int> trial = b;
            trial.insert(trial.end(), groups[gi].idx.begin(), groups[gi].idx.end());
            std::vector<DwSeg> sg;
            const int fc = segments(trial, sg);
            if (fc <= 768 && sg.size() <= 8 && trial.size() <= 16) { b = trial; placed = true; break; }
        }
        if (!placed) {
            std::vector<DwSeg> sg;
            const int fc = segments(groups[gi].idx, sg);
            if (fc > 768 || sg.size() > 8 || groups[gi].idx.size() > 16) return false;
            bins.push_back(groups[gi].idx);
        }
    }
    std::vector<DwMacro> all;
    plan.passes.clear();
    for (auto& b : bins) {
        DwPassDesc pd{};
        std::vector<DwSeg> sg;
        pd.Fc = segments(b, sg);
        pd.n_seg = (int)sg.size();
        for (int i = 0; i < pd.n_seg; i++) pd.seg[i] = sg[i];
        pd.m0 = (int)all.size();
        pd.n_macros = (int)b.size();
        pd.maxm = (pd.n_macros + 7) / 8;
        pd.nit = (pd.Fc + 511) / 512;
        auto compact = [&](int feat) {
            for (auto& s : sg) if (feat >= s.src && feat < s.src + s.len) return s.dst + feat - s.src;
            return 0;
        };
        for (int i : b) { DwMacro d = mac[i]; d.a_feat = compact(d.a_feat); d.d_feat = compact(d.d_feat); all.push_back(d); }
        plan.passes.push_back(pd);
    }
    if ((R & 1) != 0) return false;            // 8-byte loads of feature pairs
    if (hipMalloc((void**)&plan.d_macros, all.size() * sizeof(DwMacro)) != hipSuccess) { (void)hipGetLastError(); plan.passes.clear(); return false; }
    if (hipMemcpy(plan.d_macros, all.data(), all.size() * sizeof(DwMacro), hipMemcpyHostToDevice) != hipSuccess) { dw_split_free(plan); return false; }
    return true;
}

void dw_split_free(DwSplitPlan& plan) {
    if (plan.d_macros) (void)hipFree(plan.d_macros);
    plan.d_macros = nullptr;
    plan.passes.clear();
}

hipError_t launch_dw_gemm_split(const float* dwtape, size_t n_records, int row_floats, const DwSplitPlan& plan, int n_slices,
                                float* slab_rows, int slab_stride, hipStream_t stream) {
    if (n_records == 0 || n_slices < 1 || plan.passes.empty()) return hipErrorInvalidValue;
    for (const DwPassDesc& pd : plan.passes) {
        const size_t lds = (size_t)2 * 3 * (pd.Fc + 64) * 2 * 16;
#define DWS_LAUNCH(M, N) hipLaunchKernelGGL((dw_gemm_split_kernel<M, N>), dim3(n_slices), dim3(512), lds, stream, dwtape, n_records, row_floats, plan.d_macros, pd, n_slices, slab_rows, slab_stride)
        if (pd.maxm <= 1) { if (pd.nit <= 1) DWS_LAUNCH(1, 1); else DWS_LAUNCH(1, 2); }
        else { if (pd.nit <= 1) DWS_LAUNCH(2, 1); else DWS_LAUNCH(2, 2); }
#undef DWS_LAUNCH
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

bool dw_gemm_lds_fits(int row_floats, int n_macros) {
    const char* e = getenv("COLNDE_T16_DWLDS");           // 0: always the L2-streaming kernel (testing aid)
    if (e && atoi(e) == 0) return false;
    return n_macros <= DW_NW * DW_MAXM && ((CT * row_floats) & 3) == 0 && (size_t)2 * CT * row_floats * sizeof(float) <= 160 * 1024;
}

// records staged per LDS buffer: as many as fit in half the LDS, at most 4
static int dw_gemm_rpb(int row_floats) {
    const size_t rec = (size_t)CT * row_floats * sizeof(float);
    const size_t n = (80 * 1024) / rec;
    return n >= 4 ? 4 : (n >= 2 ? 2 : 1);
}

hipError_t launch_dw_gemm(const float* dwtape, size_t n_records, int row_floats, const DwMacro* macros, int n_macros, int n_slices,
                          float* slab_rows, int slab_stride, hipStream_t stream) {
    if (dw_gemm_lds_fits(row_floats, n_macros)) {
        if (n_records == 0 || n_slices < 1) return hipErrorInvalidValue;
        const int maxm = (n_macros + DW_NW - 1) / DW_NW, rpb = dw_gemm_rpb(row_floats);
        const size_t lds = (size_t)2 * rpb * CT * row_floats * sizeof(float);
#define DW_LAUNCH(M, P) hipLaunchKernelGGL((dw_gemm_lds_kernel<M, DW_NW, P>), dim3(n_slices), dim3(64 * DW_NW), lds, stream, dwtape, n_records, row_floats, macros, n_macros, n_slices, slab_rows, slab_stride)
        if (maxm == 1) { if (rpb == 4) DW_LAUNCH(1, 4); else if (rpb == 2) DW_LAUNCH(1, 2); else DW_LAUNCH(1, 1); }
        else if (maxm == 2) { if (rpb == 4) DW_LAUNCH(2, 4); else if (rpb == 2) DW_LAUNCH(2, 2); else DW_LAUNCH(2, 1); }
        else { if (rpb == 4) DW_LAUNCH(3, 4); else if (rpb == 2) DW_LAUNCH(3, 2); else DW_LAUNCH(3, 1); }
#undef DW_LAUNCH
        return hipGetLastError();
    }
    if (n_records == 0 || n_macros < 1 || n_slices < 8 || (n_slices & 7)) return hipErrorInvalidValue;
    const int n_groups = (n_macros + 3) / 4;
    const int grid = n_groups * n_slices;          // = 8 * n_groups * (n_slices / 8): every (group, slice) pair once
    hipLaunchKernelGGL(dw_gemm_kernel, dim3(grid), dim3(256), 0, stream, dwtape, n_records, row_floats, macros, n_macros, n_groups,
                       n_slices, slab_rows, slab_stride);
    return hipGetLastError();
}

// grad[p] = Σ_rows slab[row][p] in a fixed order (deterministic): 64 parameters x 16 row lanes per workgroup, lane ry sums
// rows ry, ry + 16, ..., then the 16 partial sums are added in order; the 6 raw sums become scaled mean terms
__global__ void __launch_bounds__(1024) reduce_kernel(const float* __restrict__ slab, int n_tiles, int n_params, int stride, LossWeights lw,
                                                      float* __restrict__ out /* [n_params + 8] */) {
    __shared__ float part[16][65];
    const int px = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + px;
    float s = 0.0f;
    if (p < n_params + 6)
        for (int t = ry; t < n_tiles; t += 16) s += slab[(size_t)t * stride + p];
    part[ry][px] = s;
    __syncthreads();
    if (ry == 0 && p < n_params + 6) {
        float tot = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; q++) tot += part[q][px];
        if (p >= n_params) tot *= lw.w[p - n_params];
        out[p] = tot;
    }
}

__global__ void finish_loss_kernel(float* __restrict__ out8) {
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int q = 0; q < 6; q++) t += out8[q];
        out8[6] = t;
        out8[7] = 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// embedded inference: compute_neural_network_forcing! (double_gyre_nn.jl:149-168)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) infer_kernel(DevModel m, PackInfo pk, const float* __restrict__ w, const float* __restrict__ wf,
                             const float* __restrict__ T, const float* __restrict__ top_flux, float inv_dz,
                             float* __restrict__ out, int n_col) {
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nth >> 6;
    float* xs = smem;
    float* A = xs + CT * m.ld_x;
    const int total = CT * m.ld_x + m.n_nets * CT * m.ld_a;
    for (int i = tid; i < total; i += nth) smem[i] = 0.0f;
    __syncthreads();
    const int Nz = m.Nz;
    const int oo = m.act_off[m.n_layers - 1];
    for (int tile = blockIdx.x; tile * CT < n_col; tile += gridDim.x) {
        const int col0 = tile * CT;
        for (int it = tid; it < CT * Nz; it += nth) {
            const int c = it / Nz, i = it - c * Nz;
            const float Tm = T[(size_t)min(col0 + c, n_col - 1) * Nz + i];
            xs[c * m.ld_x + i] = ((19.65f + Tm / 20.0f) - m.mu_T) / m.sig_T;      // :156, T_scaling :158
        }
        __syncthreads();
        mlp_forward<false, false>(m, pk, w, wf, xs, nullptr, A, wave, nwaves, lane);
        for (int it = tid; it < CT * Nz; it += nth) {
            const int c = it / Nz, i = it - c * Nz;
            if (col0 + c < n_col) {
                const float* o = A + c * m.ld_a + oo;
                const float lo = i == 0 ? 0.0f : m.sig_wT * o[i - 1] + m.mu_wT;           // enforce_fluxes(·, 0, surface) :160
                const float hi = i == Nz - 1 ? top_flux[col0 + c] : m.sig_wT * o[i] + m.mu_wT;
                out[(size_t)(col0 + c) * Nz + i] = -(hi - lo) * inv_dz;                    // forcing = -∂z wT :135
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// host-callable launchers (declared in engine_tile16.h)
// ------------------------------------------------------------------------------------------------
#define LAUNCH_ADJ_K(MT, NT, MR, WL, TD, ...)                                                                                   \
    do {                                                                                                                        \
        if (m.rkc)                                                                                                              \
            hipLaunchKernelGGL((adjoint_kernel<MT, NT, MR, WL, TD, true>), dim3(n_tiles), dim3(NT), lds_bytes, stream, m, pk, w, wf, \
                               wb, tiles, bias_zoff, bias_goff, bcs, save_times, n_save, substeps, sol, truth, tape, lw, slab,  \
                               n_col, __VA_ARGS__);                                                                             \
        else                                                                                                                    \
            hipLaunchKernelGGL((adjoint_kernel<MT, NT, MR, WL, TD, false>), dim3(n_tiles), dim3(NT), lds_bytes, stream, m, pk, w, wf, \
                               wb, tiles, bias_zoff, bias_goff, bcs, save_times, n_save, substeps, sol, truth, tape, lw, slab,  \
                               n_col, __VA_ARGS__);                                                                             \
    } while (0)
#define LAUNCH_ADJ(MT, NT, MR, WL) LAUNCH_ADJ_K(MT, NT, MR, WL, false, (float*)nullptr, (const float*)nullptr)

// (threads, dW tiles per wave, state items per thread, weights in LDS) instantiations; the host picks the first that fits
static const AdjointGeom kGeoms[] = {{512, 16, 3, 1}, {256, 32, 6, 1}, {256, 32, 6, 0}, {256, 32, 12, 0},
                                     {512, 32, 6, 0}, {512, 48, 3, 0}};

size_t lds_floats_adjoint_geom(const DevModel& m, const AdjointGeom& g) {
    return MODEL_FLOATS + lds_floats_adjoint(m) + (g.wlds ? (size_t)((m.n_params + 3) & ~3) + 128 : 0);
}

bool pick_adjoint_geom(const DevModel& m, AdjointGeom* geo, int force) {
    const size_t cap = 160 * 1024;
    int idx = 0;
    for (const AdjointGeom& g : kGeoms) {
        const int nwaves = g.nthreads / 64;
        const bool fits = m.n_tiles <= g.maxt * nwaves && CT * m.ns <= g.maxr * g.nthreads && m.n_bias <= MAXB * g.nthreads &&
                          lds_floats_adjoint_geom(m, g) * sizeof(float) <= cap;
        if (fits && (force < 0 || force == idx)) {
            *geo = g;
            return true;
        }
        idx++;
    }
    return false;
}

hipError_t launch_pack(const DevModel& m, const PackInfo& pk, const float* w, float* wf, float* wb, hipStream_t stream) {
    const int total = (pk.pf_net + pk.pb_net) * m.n_nets;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, m, pk, w, wf, wb);
    return hipGetLastError();
}

hipError_t launch_pack_planes(const DevModel& m, const float* w, unsigned* sf, unsigned* sb, hipStream_t stream) {
    const long total = ((long)m.sf_net + m.sb_net) * m.n_nets;
    hipLaunchKernelGGL(pack_planes_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 65535)), dim3(256), 0, stream, m, w, sf, sb);
    return hipGetLastError();
}

hipError_t launch_rhs(const DevModel& m, const PackInfo& pk, const float* w, const float* wf, const float* x,
                      const float* bcs, float t, float* dx, int n_col, int nthreads, size_t lds_bytes, hipStream_t stream, float* flux) {
    if (m.ag) hipLaunchKernelGGL((rhs_kernel<true>), dim3((n_col + CT - 1) / CT), dim3(nthreads), lds_bytes, stream, m, pk, w, wf, x, bcs, t, dx, flux, n_col);
    else hipLaunchKernelGGL((rhs_kernel<false>), dim3((n_col + CT - 1) / CT), dim3(nthreads), lds_bytes, stream, m, pk, w, wf, x, bcs, t, dx, flux, n_col);
    return hipGetLastError();
}

hipError_t launch_forward(const DevModel& m, const PackInfo& pk, const float* w, const float* wf, const float* x0,
                          const float* bcs, const float* save_times, int n_save, int substeps, float* sol, float* tape,
                          int n_col, int nthreads, bool wlds, size_t lds_bytes, hipStream_t stream, float* ztape) {
    const dim3 grid((n_col + CT - 1) / CT);
#define LAUNCH_FWD(WL, NT)                                                                                                  \
    do {                                                                                                                    \
        if (m.rkc)                                                                                                          \
            hipLaunchKernelGGL((forward_kernel<WL, NT, true>), grid, dim3(NT), lds_bytes, stream, m, pk, w, wf, x0, bcs,    \
                               save_times, n_save, substeps, sol, tape, n_col, ztape);                                      \
        else                                                                                                                \
            hipLaunchKernelGGL((forward_kernel<WL, NT, false>), grid, dim3(NT), lds_bytes, stream, m, pk, w, wf, x0, bcs,   \
                               save_times, n_save, substeps, sol, tape, n_col, ztape);                                      \
    } while (0)
    if (m.ag) {        // rows in global memory (wide networks): weights streamed from L2; 1,024 threads (four waves per SIMD hide the row loads) or 256
        if (wlds || (nthreads != 256 && nthreads != 1024)) return hipErrorInvalidValue;
#define LAUNCH_FWD_AG(NT, RK) hipLaunchKernelGGL((forward_kernel<false, NT, RK, true>), grid, dim3(NT), lds_bytes, stream, m, pk, w, wf, x0, bcs, save_times, n_save, substeps, sol, tape, n_col, ztape)
        if (nthreads == 1024) { if (m.rkc) LAUNCH_FWD_AG(1024, true); else LAUNCH_FWD_AG(1024, false); }
        else { if (m.rkc) LAUNCH_FWD_AG(256, true); else LAUNCH_FWD_AG(256, false); }
#undef LAUNCH_FWD_AG
        return hipGetLastError();
    }
    if (wlds && nthreads == 512) LAUNCH_FWD(true, 512);
    else if (wlds && nthreads == 256) LAUNCH_FWD(true, 256);
    else if (!wlds && nthreads == 512) LAUNCH_FWD(false, 512);
    else if (!wlds && nthreads == 256) LAUNCH_FWD(false, 256);
    else if (wlds && nthreads == 1024) LAUNCH_FWD(true, 1024);
    else if (!wlds && nthreads == 1024) LAUNCH_FWD(false, 1024);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_loss(const DevModel& m, const float* sol, const float* truth, int n_save, int n_col, float* partial,
                       int n_blocks, hipStream_t stream) {
    hipLaunchKernelGGL(loss_kernel, dim3(n_blocks), dim3(256), 0, stream, m, sol, truth, n_save, n_col, partial);
    return hipGetLastError();
}

hipError_t launch_adjoint(const DevModel& m, const PackInfo& pk, const float* w, const float* wf, const float* wb,
                          const TileDesc* tiles, const int* bias_zoff, const int* bias_goff, const float* bcs,
                          const float* save_times, int n_save, int substeps, const float* sol, const float* truth,
                          const float* tape, const LossWeights& lw, float* slab, int n_col, const AdjointGeom& geo,
                          size_t lds_bytes, hipStream_t stream, float* dwtape, const float* ztape) {
    const int n_tiles = (n_col + CT - 1) / CT;
    if (dwtape && m.ag) {
        // rows in global memory (wide networks): the Z tape is required (no A array), 1,024 threads (n_bias <= MAXB * 1,024, CT * ns <= 2 * 1,024: checked by the host)
        if (!ztape || CT * m.ns > 2 * 1024 || m.n_bias > MAXB * 1024) return hipErrorInvalidValue;
        if (m.rkc)
            hipLaunchKernelGGL((adjoint_kernel<1, 1024, 2, false, true, true, true>), dim3(n_tiles), dim3(1024), lds_bytes, stream, m, pk, w, wf, wb, tiles, bias_zoff,
                               bias_goff, bcs, save_times, n_save, substeps, sol, truth, tape, lw, slab, n_col, dwtape, ztape);
        else
            hipLaunchKernelGGL((adjoint_kernel<1, 1024, 2, false, true, false, true>), dim3(n_tiles), dim3(1024), lds_bytes, stream, m, pk, w, wf, wb, tiles, bias_zoff,
                               bias_goff, bcs, save_times, n_save, substeps, sol, truth, tape, lw, slab, n_col, dwtape, ztape);
        return hipGetLastError();
    }
    if (dwtape) {
        // Two 512-thread workgroups per CU (128 registers each) when two fit in the LDS: one's GEMMs then cover the other's tape
        // traffic, activations and physics (32-128-128-31: adjoint 84.6 -> 61.6 ms); one 1,024-thread workgroup (four waves per SIMD)
        // otherwise.  COLNDE_T16_TAPE_THREADS=512|1024 forces either.
        const char* et = getenv("COLNDE_T16_TAPE_THREADS");
        const int nth_env = et ? atoi(et) : ((n_tiles > 256 && 2 * lds_bytes + 2048 <= 160 * 1024) ? 512 : 1024);   // (fewer tiles than CUs: a latency point, the wider workgroup finishes sooner)
        // latency points (fewer tiles than CUs) with a network that fits beside the tile's arrays: the raw weights staged in LDS, read in
        // place for W^T — no L2 round trip at the head of every backward chain (8 simulations: adjoint 29.3 -> 25.0 ms)
        const size_t wl_bytes = ((size_t)((m.n_params + 3) & ~3) + 128) * sizeof(float);
        const char* ew = getenv("COLNDE_T16_TAPE_WLDS");
        const bool wlds_ok = nth_env == 1024 && CT * m.ns <= 2 * 1024 && lds_bytes + wl_bytes <= 160 * 1024;
        if (wlds_ok && (ew ? atoi(ew) != 0 : n_tiles <= 256)) {
            const size_t lds_save = lds_bytes;
            lds_bytes += wl_bytes;
            LAUNCH_ADJ_K(1, 1024, 2, true, true, dwtape, ztape);
            lds_bytes = lds_save;
        } else if (nth_env == 1024 && CT * m.ns <= 2 * 1024)
            LAUNCH_ADJ_K(1, 1024, 2, false, true, dwtape, ztape);
        else if (CT * m.ns <= 3 * 512)
            LAUNCH_ADJ_K(1, 512, 3, false, true, dwtape, ztape);
        else if (CT * m.ns <= 6 * 512)
            LAUNCH_ADJ_K(1, 512, 6, false, true, dwtape, ztape);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (geo.nthreads == 512 && geo.maxt == 16 && geo.maxr == 3 && geo.wlds) LAUNCH_ADJ(16, 512, 3, true);
    else if (geo.nthreads == 256 && geo.maxt == 32 && geo.maxr == 6 && geo.wlds) LAUNCH_ADJ(32, 256, 6, true);
    else if (geo.nthreads == 256 && geo.maxt == 32 && geo.maxr == 6) LAUNCH_ADJ(32, 256, 6, false);
    else if (geo.nthreads == 256 && geo.maxt == 32 && geo.maxr == 12) LAUNCH_ADJ(32, 256, 12, false);
    else if (geo.nthreads == 512 && geo.maxt == 32 && geo.maxr == 6) LAUNCH_ADJ(32, 512, 6, false);
    else if (geo.nthreads == 512 && geo.maxt == 48 && geo.maxr == 3) LAUNCH_ADJ(48, 512, 3, false);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_reduce(const float* slab, int n_tiles, int n_params, int stride, const LossWeights& lw, float* out,
                         hipStream_t stream) {
    hipLaunchKernelGGL(reduce_kernel, dim3((n_params + 6 + 63) / 64), dim3(1024), 0, stream, slab, n_tiles, n_params, stride, lw, out);
    hipLaunchKernelGGL(finish_loss_kernel, dim3(1), dim3(64), 0, stream, out + n_params);
    return hipGetLastError();
}

hipError_t launch_infer(const DevModel& m, const PackInfo& pk, const float* w, const float* wf, const float* T,
                        const float* top_flux, float inv_dz, float* out, int n_col, int nthreads, size_t lds_bytes,
                        hipStream_t stream) {
    int n_tiles = (n_col + CT - 1) / CT;
    int grid = n_tiles < 2048 ? n_tiles : 2048;
    hipLaunchKernelGGL(infer_kernel, dim3(grid), dim3(nthreads), lds_bytes, stream, m, pk, w, wf, T, top_flux, inv_dz, out, n_col);
    return hipGetLastError();
}

hipError_t debug_read_stamps(unsigned long long* out16) {
#ifdef COLNDE_STAMPS
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8);
    if (e != hipSuccess) return e;
    return hipMemcpyFromSymbol(out16 + 8, HIP_SYMBOL(g_fine), sizeof(unsigned long long) * 8);
#else
    for (int i = 0; i < 16; i++) out16[i] = 0;
    return hipSuccess;
#endif
}

hipError_t set_kernel_attributes(size_t max_lds_bytes) {
    hipError_t e;
    const int v = (int)max_lds_bytes;
#define SETATTR(K) if ((e = hipFuncSetAttribute((const void*)(K), hipFuncAttributeMaxDynamicSharedMemorySize, v)) != hipSuccess) return e
    SETATTR((rhs_kernel<false>));
    SETATTR((rhs_kernel<true>));
    SETATTR((forward_kernel<false, 256, false, true>));
    SETATTR((forward_kernel<false, 256, true, true>));
    SETATTR((forward_kernel<false, 1024, false, true>));
    SETATTR((forward_kernel<false, 1024, true, true>));
    SETATTR((adjoint_kernel<1, 1024, 2, false, true, false, true>));
    SETATTR((adjoint_kernel<1, 1024, 2, false, true, true, true>));
    SETATTR(infer_kernel);
    SETATTR((forward_kernel<true, 512>));
    SETATTR((forward_kernel<true, 256>));
    SETATTR((forward_kernel<false, 512>));
    SETATTR((forward_kernel<false, 256>));
    SETATTR((forward_kernel<true, 1024>));
    SETATTR((forward_kernel<false, 1024>));
    SETATTR((adjoint_kernel<16, 512, 3, true>));
    SETATTR((adjoint_kernel<32, 256, 6, true>));
    SETATTR((adjoint_kernel<32, 256, 6, false>));
    SETATTR((adjoint_kernel<32, 256, 12, false>));
    SETATTR((adjoint_kernel<32, 512, 6, false>));
    SETATTR((adjoint_kernel<48, 512, 3, false>));
    SETATTR((adjoint_kernel<1, 512, 3, false, true>));
    SETATTR((adjoint_kernel<1, 512, 6, false, true>));
    SETATTR((adjoint_kernel<1, 1024, 2, false, true>));
    // the RKC2 instantiations
    SETATTR((forward_kernel<true, 512, true>));
    SETATTR((forward_kernel<true, 256, true>));
    SETATTR((forward_kernel<false, 512, true>));
    SETATTR((forward_kernel<false, 256, true>));
    SETATTR((forward_kernel<true, 1024, true>));
    SETATTR((forward_kernel<false, 1024, true>));
    SETATTR((adjoint_kernel<16, 512, 3, true, false, true>));
    SETATTR((adjoint_kernel<32, 256, 6, true, false, true>));
    SETATTR((adjoint_kernel<32, 256, 6, false, false, true>));
    SETATTR((adjoint_kernel<32, 256, 12, false, false, true>));
    SETATTR((adjoint_kernel<32, 512, 6, false, false, true>));
    SETATTR((adjoint_kernel<48, 512, 3, false, false, true>));
    SETATTR((adjoint_kernel<1, 512, 3, false, true, true>));
    SETATTR((adjoint_kernel<1, 512, 6, false, true, true>));
    SETATTR((adjoint_kernel<1, 1024, 2, false, true, true>));
    SETATTR((adjoint_kernel<1, 1024, 2, true, true, false>));
    SETATTR((adjoint_kernel<1, 1024, 2, true, true, true>));
    SETATTR((dw_gemm_split_kernel<1, 1>)); SETATTR((dw_gemm_split_kernel<1, 2>)); SETATTR((dw_gemm_split_kernel<2, 1>)); SETATTR((dw_gemm_split_kernel<2, 2>));
    SETATTR((dw_gemm_lds_kernel<1, DW_NW, 1>)); SETATTR((dw_gemm_lds_kernel<1, DW_NW, 2>)); SETATTR((dw_gemm_lds_kernel<1, DW_NW, 4>));
    SETATTR((dw_gemm_lds_kernel<2, DW_NW, 1>)); SETATTR((dw_gemm_lds_kernel<2, DW_NW, 2>)); SETATTR((dw_gemm_lds_kernel<2, DW_NW, 4>));
    SETATTR((dw_gemm_lds_kernel<3, DW_NW, 1>)); SETATTR((dw_gemm_lds_kernel<3, DW_NW, 2>)); SETATTR((dw_gemm_lds_kernel<3, DW_NW, 4>));
#undef SETATTR
    return hipSuccess;
}
