// engine_fc.hip — "fc32": the free-convection NDE on 32-column MFMA tiles with compile-time shapes (gfx950 only).
//
// Covers FreeConvectionNDE (free_convection/src/free_convection_nde.jl:29-38: dT/dt = -(σ_wT/σ_T)(τ/H) Dᶜ [b; NN(T); t]) with the network
// the reference trains, Dense(Nz,4Nz,relu) -> Dense(4Nz,4Nz,relu) -> Dense(4Nz,Nz-1) (train_free_convection_nde.jl:119-121), Nz = 32 or
// 64 (BASELINE configs[3] and its 32-level sibling), classical RK4.  Everything else stays on the generic tile16 engine.
//
// Why a second engine for this shape: 64-256-256-63 has 98,623 weights (394 KB) — they cannot live in a CU's LDS, so every stage of every
// column tile streams the whole A-operand image from L2.  tile16 feeds v_mfma_f32_16x16x4_f32 with 16-column tiles: 8 flop per streamed
// byte, i.e. 19.6 TB/s of L2->CU traffic at the fp32 MFMA peak, plus ≈50 address instructions per MFMA from runtime shapes (measured
// 44 % of peak on config 4).  Here a workgroup owns 32 columns and every dense layer runs on v_mfma_f32_32x32x2_f32 (M = 32 output
// rows, N = the 32 columns, 64 cycles): the same flops per cycle with HALF the operand bytes per flop for both operands, one 16-byte
// L2 load + one ds_read_b128 per FOUR MFMAs (256 cycles), all offsets compile-time immediates.
//
//  * 256 threads = 4 wavefronts, two workgroups per CU (LDS 75 KB / 67 KB): one's epilogues, physics and barriers hide under the
//    other's MFMA chains.  Layer l's row tiles are dealt to the waves (tiles w, w + 4); the last layer's two row tiles are split in K
//    so that all four waves work (partial sums to LDS, added in a fixed order by the physics).
//  * The A operand of a wave is ONE continuous stream — the same sequence of 16-byte groups every stage — fetched PF groups ahead
//    through a register ring that runs across layer boundaries, barriers and stages: L2 latency is exposed once per kernel.
//  * Workgroup barriers are bare `s_waitcnt lgkmcnt(0); s_barrier` (a __syncthreads() would drain the prefetch ring and the tape stores).
//  * Tapes: the forward kernel writes the stage input and the hidden activations a1, a2 STRAIGHT into tile16's delta-tape record
//    ([16 columns][xs | a1 a2 . | dz1 dz2 dz3], two records per 32-column tile and stage) and, separately, relu's derivative as one
//    bit per hidden unit in the accumulator layout (2 KB per tile and stage instead of 64 KB of pre-activations); the adjoint kernel
//    reads back nothing but those bits, back-propagates through W3ᵀ, W2ᵀ, W1ᵀ and fills the record's dz part; tile16's split-K dW GEMM
//    contracts the records unchanged.  Per column and stage: 4.9 KB of tape against tile16's 7.2 KB.
//  * Everything is summed in a fixed order: bit-reproducible gradients.
#include "engine_fc.h"

typedef float fc16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32;

extern __shared__ float fc_smem[];

#define FC_PF 8                                    // ring depth in groups of four MFMAs: 8 x 256 cycles of cover
#define FC_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// The group addresses of a stage are loop-invariant: left alone, the optimiser computes all ~100 of them once, outside the time loop, and
// keeps them in (spilled) registers.  An opaque zero added to the wave-uniform bases at the top of every stage keeps them what they should
// be: a scalar base, a compile-time offset and the lane.  (The pointers themselves stay kernel-argument-derived: laundering THEM loses the global
// address space and turns every load into a flat_load, which counts against lgkmcnt as well.)
#define FC_OPAQUE_ZERO(z) asm volatile("" : "+s"(z))

template <int NZ>
struct Fc {
    static constexpr int H = 4 * NZ, NO = NZ - 1;
    static constexpr int LDX = NZ + 4, LDH = H + 4;              // LDS row strides: 16-byte aligned rows, stride/4 odd => conflict-free ds_read_b128
    static constexpr int MT = H / 32, JH = MT / 4;               // row tiles of a hidden layer; jobs per wave
    static constexpr int S_IN = NZ / 8, S_H = H / 8;             // groups of 8 k per chain: K = NZ, K = 4 NZ
    static constexpr int MT3 = NZ / 32, KS3 = 4 / MT3, G3 = S_H / KS3;   // narrow layer (M = NZ): row tiles, K splits, groups per job
    static constexpr int F1 = 0, F1_SZ = MT * S_IN * 256;
    static constexpr int F2 = F1 + F1_SZ, F2_SZ = MT * S_H * 256;
    static constexpr int F3 = F2 + F2_SZ, F3_SZ = MT3 * S_H * 256;
    static constexpr int IMG = F3 + F3_SZ;                       // floats per operand image (forward and backward alike)
    static constexpr int BIAS = 2 * H + NZ;
    static constexpr int P = JH * S_IN + JH * S_H + G3;          // groups per stage and wave
    static constexpr int ACT4 = 2 * H + NZ;                      // dwtape_act4: (2H + NZ - 1) rounded up to 4
    static constexpr int R = NZ + 2 * ACT4;                      // floats per column of a delta-tape record (dwtape_row_floats)
    static constexpr int OWN = 32 * NZ / 256;                    // state items (column, level) per thread
    static_assert(P % FC_PF == 0, "the ring must close over one stage");
};

// stream position -> layer section (0: K = NZ hidden, 1: K = 4NZ hidden, 2: the narrow layer) and offset in float4 units from the wave's base
template <int NZ> __host__ __device__ constexpr int fc_sec(int p) {
    p %= Fc<NZ>::P;
    return p < Fc<NZ>::JH * Fc<NZ>::S_IN ? 0 : (p < Fc<NZ>::JH * (Fc<NZ>::S_IN + Fc<NZ>::S_H) ? 1 : 2);
}
template <int NZ> __host__ __device__ constexpr int fc_off(int p) {
    using S = Fc<NZ>;
    p %= S::P;
    if (p < S::JH * S::S_IN) return (4 * (p / S::S_IN) * S::S_IN + p % S::S_IN) * 64;
    p -= S::JH * S::S_IN;
    if (p < S::JH * S::S_H) return (4 * (p / S::S_H) * S::S_H + p % S::S_H) * 64;
    return (p - S::JH * S::S_H) * 64;
}

// One section of the wave's stream: NJ jobs (output row tiles) of NG groups each, starting at stream position P0.  Per group: four
// MFMAs fed by one ring slot (A: four k-steps of this lane's weight row) and one 16-byte LDS read (B: the same four k of column n);
// the slot is refilled with the group FC_PF positions ahead — possibly the next layer's or the next stage's.
template <int NZ, int P0, int NJ, int NG, class Epi>
__device__ __forceinline__ void fc_section(f32x4 (&ring)[FC_PF], const f32x4* const (&base)[3], int lane, const float* brow, Epi&& epi) {
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        fc16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        f32x4 b[2];
        b[0] = *reinterpret_cast<const f32x4*>(brow);
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int p = P0 + j * NG + g;
            if (g + 1 < NG) b[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(brow + 8 * (g + 1));
            const f32x4 a = ring[p % FC_PF];
            ring[p % FC_PF] = (base[fc_sec<NZ>(p + FC_PF)] + fc_off<NZ>(p + FC_PF))[lane];
            const f32x4 bv = b[g & 1];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bv.w, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        epi(j, acc);
    }
}

// ------------------------------------------------------------------------------------------------
// operand images.  Flux.destructure: W_l[o][i] (out o, in i) at w_off[l] + i*no + o, b_l[o] at b_off[l] + o.
//   forward  section (rows = outputs):  A[row = mt*32 + (lane&31)][k = 8S + 4(lane>>5) + j] = W_l[row][k]
//   backward section (rows = inputs):   A[row = it*32 + (lane&31)][k = 8S + 4(lane>>5) + j] = W_l[k][row]      (zero beyond the matrix)
// image[section][tile][S][lane][j]; backward sections in the order they are used: W3ᵀ (K = NZ), W2ᵀ, W1ᵀ (the narrow one).
// ------------------------------------------------------------------------------------------------
struct FcOffsets { int w[3], b[3]; };

template <int NZ>
__global__ void __launch_bounds__(256) fc_pack_kernel(FcOffsets o, const float* __restrict__ w, float* __restrict__ imgf, float* __restrict__ imgb,
                                                      float* __restrict__ bias) {
    using S = Fc<NZ>;
    const int total = 2 * S::IMG + S::BIAS;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        if (idx >= 2 * S::IMG) {
            const int q = idx - 2 * S::IMG;
            float v;
            if (q < S::H) v = w[o.b[0] + q];
            else if (q < 2 * S::H) v = w[o.b[1] + q - S::H];
            else v = q - 2 * S::H < S::NO ? w[o.b[2] + q - 2 * S::H] : 0.0f;
            bias[q] = v;
            continue;
        }
        const bool fwd = idx < S::IMG;
        const int e = fwd ? idx : idx - S::IMG;
        const int sec = e < S::F2 ? 0 : (e < S::F3 ? 1 : 2);
        const int r = e - (sec == 0 ? S::F1 : (sec == 1 ? S::F2 : S::F3));
        const int nS = sec == 0 ? S::S_IN : S::S_H;
        const int j = r & 3, lane = (r >> 2) & 63, blk = r >> 8;
        const int tile = blk / nS, Sg = blk - tile * nS;
        const int row = tile * 32 + (lane & 31), k = 8 * Sg + 4 * (lane >> 5) + j;
        float v = 0.0f;
        if (fwd) {
            // section 0: W1 (NZ -> H), 1: W2 (H -> H), 2: W3 (H -> NO)
            const int no = sec == 2 ? S::NO : S::H;
            if (row < no) v = w[o.w[sec] + k * no + row];
        } else {
            // section 0: W3ᵀ (rows = a2 features, k = outputs of layer 3), 1: W2ᵀ, 2: W1ᵀ (rows = state levels)
            const int l = 2 - sec;
            const int no = l == 2 ? S::NO : S::H;
            if (k < no) v = w[o.w[l] + row * no + k];
        }
        (fwd ? imgf : imgb)[e] = v;
    }
}

// state items owned by a thread: item it = tid + 256 r  ->  (column it / NZ, level it % NZ)
#define FC_OWNER_INDEX()                                                   \
    int oc[S::OWN];                                                        \
    const int oi = tid & (NZ - 1);                                         \
    _Pragma("unroll") for (int r = 0; r < S::OWN; r++) oc[r] = (tid + 256 * r) / NZ

// ------------------------------------------------------------------------------------------------
// forward solve (and, TAPE, the forward half of the tapes)
// ------------------------------------------------------------------------------------------------
template <int NZ, bool TAPE>
__global__ void __launch_bounds__(256, 2)
fc_forward_kernel(const float* __restrict__ imgf, const float* __restrict__ bias, const float* __restrict__ x0, const float* __restrict__ bcs,
                  const float* __restrict__ save_times, int n_save, int substeps, float CN, float* __restrict__ sol, float* __restrict__ dwtape,
                  u32* __restrict__ masks, int n_col) {
    using S = Fc<NZ>;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    float* X = fc_smem;                          // [32][LDX]   stage input
    float* A1 = X + 32 * S::LDX;                 // [32][LDH]   relu(W1 x + b1)
    float* A2 = A1 + 32 * S::LDH;                // [32][LDH]   relu(W2 a1 + b2)
    float* PART = A1;                            // [KS3][32][NZ] partial sums of the last layer (a1 is dead by then)
    float* BL = A2 + 32 * S::LDH;                // [2H + NZ] biases (a global load in an epilogue would be waited for with vmcnt(0): the ring too)
    for (int q = tid; q < S::BIAS; q += 256) BL[q] = bias[q];
    const int col0 = blockIdx.x * 32;
    FC_OWNER_INDEX();

    const f32x4* base[3];
    base[0] = reinterpret_cast<const f32x4*>(imgf + S::F1) + (w * S::S_IN) * 64;
    base[1] = reinterpret_cast<const f32x4*>(imgf + S::F2) + (w * S::S_H) * 64;
    base[2] = reinterpret_cast<const f32x4*>(imgf + S::F3) + ((w % S::MT3) * S::S_H + (w / S::MT3) * S::G3) * 64;
    f32x4 ring[FC_PF];
#pragma unroll
    for (int q = 0; q < FC_PF; q++) ring[q] = (base[fc_sec<NZ>(q)] + fc_off<NZ>(q))[lane];

    float xn[S::OWN], ac[S::OWN], kv[S::OWN], bcb[S::OWN], bct[S::OWN];
#pragma unroll
    for (int r = 0; r < S::OWN; r++) {
        const int col = min(col0 + oc[r], n_col - 1);
        xn[r] = x0[(size_t)col * NZ + oi];
        bcb[r] = bcs[(size_t)col * 2];
        bct[r] = bcs[(size_t)col * 2 + 1];
        ac[r] = 0.0f;
        kv[r] = 0.0f;
        if (sol && col0 + oc[r] < n_col) sol[((size_t)(col0 + oc[r]) * n_save) * NZ + oi] = xn[r];
    }
    const float b3v = oi < S::NO ? bias[2 * S::H + oi] : 0.0f;
    // every load issued so far is consumed HERE: a register still "in flight" at the loop header makes the wait-count pass put a
    // vmcnt(0) at the top of every stage, which would drain the prefetch ring each time
#pragma unroll
    for (int r = 0; r < S::OWN; r++) asm volatile("" :: "v"(xn[r]), "v"(bcb[r]), "v"(bct[r]));
    asm volatile("" :: "v"(b3v));
    const int n_steps = (n_save - 1) * substeps;
    int step = 0;
    for (int iv = 0; iv < n_save - 1; iv++) {
        const float dt = (save_times[iv + 1] - save_times[iv]) / (float)substeps;
        for (int s = 0; s < substeps; s++, step++) {
#pragma nounroll
            for (int st = 0; st < 4; st++) {
                int zero = 0;
                FC_OPAQUE_ZERO(zero);
                const f32x4* const sb[3] = {base[0] + zero, base[1] + zero, base[2] + zero};
                const float ca = st == 0 ? 0.0f : (st == 3 ? 1.0f : 0.5f);            // stage abscissa
                const float cbp = st == 1 ? 1.0f / 6.0f : 1.0f / 3.0f;                 // RK4 weight of k_{st-1}
                float* rec = TAPE ? dwtape + (((size_t)blockIdx.x * n_steps + step) * 4 + st) * ((size_t)32 * S::R) : nullptr;
                u32* mrec = TAPE ? masks + (((size_t)blockIdx.x * n_steps + step) * 4 + st) * 512 + w * 64 + lane : nullptr;
                // ---- stage input (owner layout) -> LDS rows, tape
#pragma unroll
                for (int r = 0; r < S::OWN; r++) {
                    float v = xn[r];
                    if (st > 0) {
                        ac[r] += cbp * kv[r];
                        v += ca * dt * kv[r];
                    }
                    X[oc[r] * S::LDX + oi] = v;
                    if (TAPE) __builtin_nontemporal_store(v, rec + (size_t)oc[r] * S::R + oi);
                }
                FC_BARRIER();
                // ---- hidden layers: z = W a + b on 32x32x2 MFMA, relu, rows to LDS (next layer's B operand) and to the tape
                auto hidden = [&](int l /* 1, 2 */, float* dstrows, int j, const fc16& acc) {
                    const int mt = w + 4 * j;
                    const float* bl = BL + (l - 1) * S::H + mt * 32 + 4 * h;
                    u32 bits = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const f32x4 bq = *reinterpret_cast<const f32x4*>(bl + 8 * q);
                        f32x4 a;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const float z = acc[4 * q + e] + bq[e];
                            a[e] = fmaxf(z, 0.0f);
                            bits |= (z > 0.0f ? 1u : 0u) << (4 * q + e);
                        }
                        const int f = mt * 32 + 8 * q + 4 * h;
                        *reinterpret_cast<f32x4*>(dstrows + n * S::LDH + f) = a;
                        if (TAPE) __builtin_nontemporal_store(a, reinterpret_cast<f32x4*>(rec + (size_t)n * S::R + NZ + (l - 1) * S::H + f));
                    }
                    return bits;
                };
                {
                    u32 mb = 0;
                    fc_section<NZ, 0, S::JH, S::S_IN>(ring, sb, lane, X + n * S::LDX + 4 * h,
                                                      [&](int j, const fc16& acc) { mb |= hidden(1, A1, j, acc) << (16 * j); });
                    if (TAPE) __builtin_nontemporal_store(mb, mrec);
                }
                FC_BARRIER();
                {
                    u32 mb = 0;
                    fc_section<NZ, S::JH * S::S_IN, S::JH, S::S_H>(ring, sb, lane, A1 + n * S::LDH + 4 * h,
                                                                   [&](int j, const fc16& acc) { mb |= hidden(2, A2, j, acc) << (16 * j); });
                    if (TAPE) __builtin_nontemporal_store(mb, mrec + 256);
                }
                FC_BARRIER();
                // ---- output layer: row tile w % MT3, K part w / MT3; partial sums to LDS
                fc_section<NZ, S::JH * (S::S_IN + S::S_H), 1, S::G3>(ring, sb, lane, A2 + n * S::LDH + (w / S::MT3) * S::G3 * 8 + 4 * h,
                    [&](int, const fc16& acc) {
                        float* pr = PART + ((w / S::MT3) * 32 + n) * NZ + (w % S::MT3) * 32 + 4 * h;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                            *reinterpret_cast<f32x4*>(pr + 8 * q) = v;
                        }
                    });
                FC_BARRIER();
                // ---- physics (free_convection_nde.jl:29-38): faces [b; NN(T); t], dT = -C Nz (w[i+1] - w[i])
#pragma unroll
                for (int r = 0; r < S::OWN; r++) {
                    float o = b3v;
#pragma unroll
                    for (int ks = 0; ks < S::KS3; ks++) o += PART[(ks * 32 + oc[r]) * NZ + oi];
                    const float olo = __shfl_up(o, 1);                                // NN output of face i (lane i - 1 holds it)
                    const float wlo = oi == 0 ? bcb[r] : olo;
                    const float whi = oi == NZ - 1 ? bct[r] : o;
                    kv[r] = -CN * (whi - wlo);
                }
            }
            const bool save = s == substeps - 1;
#pragma unroll
            for (int r = 0; r < S::OWN; r++) {
                ac[r] += (1.0f / 6.0f) * kv[r];
                xn[r] += dt * ac[r];
                ac[r] = 0.0f;
                if (save && sol && col0 + oc[r] < n_col) sol[((size_t)(col0 + oc[r]) * n_save + iv + 1) * NZ + oi] = xn[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// adjoint: back-propagation through the RK4 stages from the taped relu bits; fills the dz part of the delta-tape records
// ------------------------------------------------------------------------------------------------
struct FcGrad { int b[3]; int n_params; };

template <int NZ>
__global__ void __launch_bounds__(256, 2)
fc_adjoint_kernel(const float* __restrict__ imgb, const float* __restrict__ save_times, int n_save, int substeps, float CN,
                  const float* __restrict__ sol, const float* __restrict__ truth, float* __restrict__ dwtape, const u32* __restrict__ masks,
                  float w_loss, float* __restrict__ slab, FcGrad go, int n_col) {
    using S = Fc<NZ>;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    float* DZ2 = fc_smem;                        // [32][LDH]
    float* DZ1 = DZ2 + 32 * S::LDH;              // [32][LDH]
    float* DZ3 = DZ1;                            // [32][LDX]   dead before dz1 is written
    float* XBP = DZ2;                            // [KS3][32][NZ] partial sums of W1ᵀ dz1 (dz2 is dead by then)
    const int col0 = blockIdx.x * 32;
    FC_OWNER_INDEX();

    const f32x4* base[3];
    base[0] = reinterpret_cast<const f32x4*>(imgb + S::F1) + (w * S::S_IN) * 64;
    base[1] = reinterpret_cast<const f32x4*>(imgb + S::F2) + (w * S::S_H) * 64;
    base[2] = reinterpret_cast<const f32x4*>(imgb + S::F3) + ((w % S::MT3) * S::S_H + (w / S::MT3) * S::G3) * 64;
    f32x4 ring[FC_PF];
#pragma unroll
    for (int q = 0; q < FC_PF; q++) ring[q] = (base[fc_sec<NZ>(q)] + fc_off<NZ>(q))[lane];

    float lam[S::OWN], xb[S::OWN], xbs[S::OWN], db3[S::OWN];
    fc16 db2[S::JH], db1[S::JH];
#pragma unroll
    for (int j = 0; j < S::JH; j++) {
        db2[j] = (fc16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        db1[j] = db2[j];
    }
    float sumsq = 0.0f;
#pragma unroll
    for (int r = 0; r < S::OWN; r++) {
        lam[r] = 0.0f; xb[r] = 0.0f; db3[r] = 0.0f;
        if (col0 + oc[r] < n_col) {                                 // save point 0 enters the loss value only
            const size_t q = ((size_t)(col0 + oc[r]) * n_save) * NZ + oi;
            const float d = sol[q] - truth[q];
            sumsq += d * d;
        }
    }
    const int n_steps = (n_save - 1) * substeps;
    for (int iv = n_save - 2; iv >= 0; iv--) {
        const float dt = (save_times[iv + 1] - save_times[iv]) / (float)substeps;
        // λ += ∂loss/∂sol[:, iv+1]   (nde_loss = Flux.mse over every (level, save point, simulation): training.jl:55-62)
#pragma unroll
        for (int r = 0; r < S::OWN; r++)
            if (col0 + oc[r] < n_col) {
                const size_t q = ((size_t)(col0 + oc[r]) * n_save + iv + 1) * NZ + oi;
                const float d = sol[q] - truth[q];
                sumsq += d * d;
                lam[r] += 2.0f * w_loss * d;
            }
        for (int s = substeps - 1; s >= 0; s--) {
            const int step = iv * substeps + s;
#pragma unroll
            for (int r = 0; r < S::OWN; r++) xbs[r] = 0.0f;
#pragma nounroll
            for (int st = 3; st >= 0; st--) {
                // k̄4 = dt/6 λ; k̄3 = dt/3 λ + dt x̄4; k̄2 = dt/3 λ + dt/2 x̄3; k̄1 = dt/6 λ + dt/2 x̄2
                int zero = 0;
                FC_OPAQUE_ZERO(zero);
                const f32x4* const sb[3] = {base[0] + zero, base[1] + zero, base[2] + zero};
                const float cwl = (st == 0 || st == 3) ? dt / 6.0f : dt / 3.0f;
                const float cwx = st == 3 ? 0.0f : (st == 2 ? dt : 0.5f * dt);
                float* rec = dwtape + (((size_t)blockIdx.x * n_steps + step) * 4 + st) * ((size_t)32 * S::R);
                const u32* mrec = masks + (((size_t)blockIdx.x * n_steps + step) * 4 + st) * 512 + w * 64 + lane;
                const u32 m1 = __builtin_nontemporal_load(mrec), m2 = __builtin_nontemporal_load(mrec + 256);
                // ---- stage cotangent and the physics pullback: dz3[i] = C Nz (k̄[i+1] - k̄[i]) on the Nz-1 interior faces
#pragma unroll
                for (int r = 0; r < S::OWN; r++) {
                    const float kb = cwl * lam[r] + cwx * xb[r];
                    const float kn = __shfl_down(kb, 1);
                    const float dz = oi < S::NO ? CN * (kn - kb) : 0.0f;
                    DZ3[oc[r] * S::LDX + oi] = dz;
                    __builtin_nontemporal_store(dz, rec + (size_t)oc[r] * S::R + NZ + S::ACT4 + 2 * S::H + oi);
                    db3[r] += dz;
                }
                FC_BARRIER();
                auto hidden = [&](int l /* 2, 1: layer whose dz this is */, float* dstrows, u32 bits, fc16& dbacc, int j, const fc16& acc) {
                    const int mt = w + 4 * j;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        f32x4 d;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            d[e] = ((bits >> (16 * j + 4 * q + e)) & 1u) ? acc[4 * q + e] : 0.0f;
                            dbacc[4 * q + e] += d[e];
                        }
                        const int f = mt * 32 + 8 * q + 4 * h;
                        *reinterpret_cast<f32x4*>(dstrows + n * S::LDH + f) = d;
                        __builtin_nontemporal_store(d, reinterpret_cast<f32x4*>(rec + (size_t)n * S::R + NZ + S::ACT4 + (l - 1) * S::H + f));
                    }
                };
                // ---- dz2 = relu'(z2) ∘ W3ᵀ dz3
                fc_section<NZ, 0, S::JH, S::S_IN>(ring, sb, lane, DZ3 + n * S::LDX + 4 * h,
                                                  [&](int j, const fc16& acc) { hidden(2, DZ2, m2, db2[j], j, acc); });
                FC_BARRIER();
                // ---- dz1 = relu'(z1) ∘ W2ᵀ dz2
                fc_section<NZ, S::JH * S::S_IN, S::JH, S::S_H>(ring, sb, lane, DZ2 + n * S::LDH + 4 * h,
                                                               [&](int j, const fc16& acc) { hidden(1, DZ1, m1, db1[j], j, acc); });
                FC_BARRIER();
                // ---- x̄ = W1ᵀ dz1: row tile w % MT3, K part w / MT3
                fc_section<NZ, S::JH * (S::S_IN + S::S_H), 1, S::G3>(ring, sb, lane, DZ1 + n * S::LDH + (w / S::MT3) * S::G3 * 8 + 4 * h,
                    [&](int, const fc16& acc) {
                        float* pr = XBP + ((w / S::MT3) * 32 + n) * NZ + (w % S::MT3) * 32 + 4 * h;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                            *reinterpret_cast<f32x4*>(pr + 8 * q) = v;
                        }
                    });
                FC_BARRIER();
#pragma unroll
                for (int r = 0; r < S::OWN; r++) {
                    float v = 0.0f;
#pragma unroll
                    for (int ks = 0; ks < S::KS3; ks++) v += XBP[(ks * 32 + oc[r]) * NZ + oi];
                    xb[r] = v;
                    xbs[r] += v;
                }
                // (the next stage writes DZ3 = DZ1's rows: every wave's reads of DZ1 ended before the barrier above; XBP = DZ2's rows
                //  are next written two barriers from here)
            }
#pragma unroll
            for (int r = 0; r < S::OWN; r++) lam[r] += xbs[r];
        }
    }
    // ---- flush: bias gradients and the loss sum into this workgroup's slab row (weight gradients come from the dW GEMM)
    FC_BARRIER();
    float* out = slab + (size_t)blockIdx.x * (go.n_params + 8);
    float* scr = fc_smem;                                            // [4][NZ] + [4]
    {
        float s3 = 0.0f;
#pragma unroll
        for (int r = 0; r < S::OWN; r++) s3 += db3[r];              // this thread's columns, level oi
        if (NZ == 32) s3 += __shfl_down(s3, 32);                     // the wave's second column group
        if (lane < NZ) scr[w * NZ + lane] = s3;
        float v = sumsq;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) scr[4 * NZ + w] = v;
    }
    FC_BARRIER();
    if (tid < S::NO) out[go.b[2] + tid] = (scr[tid] + scr[NZ + tid]) + (scr[2 * NZ + tid] + scr[3 * NZ + tid]);
    if (tid == 0) out[go.n_params + 2] = (scr[4 * NZ] + scr[4 * NZ + 1]) + (scr[4 * NZ + 2] + scr[4 * NZ + 3]);
#pragma unroll
    for (int j = 0; j < S::JH; j++)
#pragma unroll
        for (int l = 0; l < 2; l++) {
            const fc16& acc = l == 0 ? db1[j] : db2[j];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                float v = acc[r];
                for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);      // over the 32 columns of this half (h fixed)
                if (n == 0) out[go.b[l] + (w + 4 * j) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = v;
            }
        }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool fc_supported(const DevModel& m, int stepper) {
    if (m.model != COLNDE_MODEL_FREE_CONVECTION || stepper != COLNDE_STEPPER_RK4) return false;
    if (m.Nz != 32 && m.Nz != 64) return false;
    if (m.n_layers != 3 || m.n_nets != 1) return false;
    if (m.sizes[0] != m.Nz || m.sizes[1] != 4 * m.Nz || m.sizes[2] != 4 * m.Nz || m.sizes[3] != m.Nz - 1) return false;
    return m.acts[0] == COLNDE_ACT_RELU && m.acts[1] == COLNDE_ACT_RELU && m.acts[2] == COLNDE_ACT_IDENTITY;
}

size_t fc_image_floats(int Nz) { return Nz == 64 ? Fc<64>::IMG : Fc<32>::IMG; }
size_t fc_bias_floats(int Nz) { return Nz == 64 ? Fc<64>::BIAS : Fc<32>::BIAS; }
size_t fc_record_row_floats(int Nz) { return Nz == 64 ? Fc<64>::R : Fc<32>::R; }

template <int NZ> static size_t fc_lds_fwd() { return (size_t)(32 * Fc<NZ>::LDX + 2 * 32 * Fc<NZ>::LDH + Fc<NZ>::BIAS) * sizeof(float); }
template <int NZ> static size_t fc_lds_adj() { return (size_t)(2 * 32 * Fc<NZ>::LDH) * sizeof(float); }

hipError_t fc_set_kernel_attributes() {
    hipError_t e;
#define FC_ATTR(K, B) if ((e = hipFuncSetAttribute((const void*)(K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(B))) != hipSuccess) return e
    FC_ATTR((fc_forward_kernel<64, true>), fc_lds_fwd<64>());
    FC_ATTR((fc_forward_kernel<64, false>), fc_lds_fwd<64>());
    FC_ATTR((fc_forward_kernel<32, true>), fc_lds_fwd<32>());
    FC_ATTR((fc_forward_kernel<32, false>), fc_lds_fwd<32>());
    FC_ATTR((fc_adjoint_kernel<64>), fc_lds_adj<64>());
    FC_ATTR((fc_adjoint_kernel<32>), fc_lds_adj<32>());
#undef FC_ATTR
    return hipSuccess;
}

hipError_t fc_launch_pack(const DevModel& m, const float* w, float* imgf, float* imgb, float* bias, hipStream_t stream) {
    FcOffsets o;
    for (int l = 0; l < 3; l++) { o.w[l] = m.w_off[l]; o.b[l] = m.b_off[l]; }
    if (m.Nz == 64) hipLaunchKernelGGL(fc_pack_kernel<64>, dim3(256), dim3(256), 0, stream, o, w, imgf, imgb, bias);
    else hipLaunchKernelGGL(fc_pack_kernel<32>, dim3(128), dim3(256), 0, stream, o, w, imgf, imgb, bias);
    return hipGetLastError();
}

hipError_t fc_launch_forward(const DevModel& m, const float* imgf, const float* bias, const float* x0, const float* bcs, const float* save_times,
                             int n_save, int substeps, float* sol, float* dwtape, unsigned int* masks, int n_col, hipStream_t stream) {
    if (n_col < 1) return hipErrorInvalidValue;
    const dim3 grid((n_col + 31) / 32), block(256);
    const float CN = m.C_fc * (float)m.Nz;
    const bool tape = dwtape != nullptr;
    if (tape && !masks) return hipErrorInvalidValue;
#define FC_FWD(N, T) hipLaunchKernelGGL((fc_forward_kernel<N, T>), grid, block, fc_lds_fwd<N>(), stream, imgf, bias, x0, bcs, save_times, n_save, substeps, CN, sol, dwtape, masks, n_col)
    if (m.Nz == 64) { if (tape) FC_FWD(64, true); else FC_FWD(64, false); }
    else { if (tape) FC_FWD(32, true); else FC_FWD(32, false); }
#undef FC_FWD
    return hipGetLastError();
}

hipError_t fc_launch_adjoint(const DevModel& m, const float* imgb, const float* save_times, int n_save, int substeps, const float* sol,
                             const float* truth, float* dwtape, const unsigned int* masks, float w_loss, float* slab, int n_col,
                             hipStream_t stream) {
    if (n_col < 1 || !dwtape || !masks) return hipErrorInvalidValue;
    const dim3 grid((n_col + 31) / 32), block(256);
    const float CN = m.C_fc * (float)m.Nz;
    FcGrad go;
    for (int l = 0; l < 3; l++) go.b[l] = m.b_off[l];
    go.n_params = m.n_params;
    if (m.Nz == 64) hipLaunchKernelGGL(fc_adjoint_kernel<64>, grid, block, fc_lds_adj<64>(), stream, imgb, save_times, n_save, substeps, CN, sol, truth,
                                       dwtape, masks, w_loss, slab, go, n_col);
    else hipLaunchKernelGGL(fc_adjoint_kernel<32>, grid, block, fc_lds_adj<32>(), stream, imgb, save_times, n_save, substeps, CN, sol, truth, dwtape,
                            masks, w_loss, slab, go, n_col);
    return hipGetLastError();
}
