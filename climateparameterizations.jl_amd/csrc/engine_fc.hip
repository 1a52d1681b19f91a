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
//  * COLNDE_MATRIX_BF16X3_EXACT (round 4): the 32-column tiles have kernels of their own (engine_fc_split.hip); the 16-column tiles of the latency sizes
//    (up to 4,096 columns) run THIS file's kernels with the operand stream swapped (FcStream<NZ, 16, true>): pre-split weight planes through the same kind
//    of ring, activations split by the wave that reads them (one split per 32-deep k-block, shared by its four / two row tiles, software-pipelined under
//    the previous block's MFMAs), v_mfma_f32_16x16x32_bf16.  Tapes, masks, epilogues, physics: unchanged.  8 simulations x 64 levels: 30.3 -> 24.6 ms.
#include <type_traits>
#include "engine_fc.h"
#include "split_bf16.h"

typedef float fc16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32;

extern __shared__ float fc_smem[];

#ifndef FC_PF
#define FC_PF 8                                    // ring depth in groups of four MFMAs: 8 x 256 cycles of cover
#endif
#ifndef FC_NT
#define FC_NT 0                                    // tape traffic with the non-temporal cache policy: measured SLOWER (A/B on one box, 16,384 columns:
#endif                                             // Nz = 32 forward 26.8 vs 16.7 ms, adjoint 25.4 vs 17.6; Nz = 64 61.0 vs 53.9 and 61.0 vs 54.0) — an
                                                   // nt store is acknowledged late, and every wait for a ring load (vmcnt is in order) waits behind it
#if FC_NT
#define FC_STORE(v, p) __builtin_nontemporal_store(v, p)
#define FC_LOAD(p) __builtin_nontemporal_load(p)
#else
#define FC_STORE(v, p) (*(p) = (v))
#define FC_LOAD(p) (*(p))
#endif
#define FC_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// The group addresses of a stage are loop-invariant: left alone, the optimiser computes all ~100 of them once, outside the time loop, and
// keeps them in (spilled) registers.  An opaque zero added to the wave-uniform bases at the top of every stage keeps them what they should
// be: a scalar base, a compile-time offset and the lane.  (The pointers themselves stay kernel-argument-derived: laundering THEM loses the global
// address space and turns every load into a flat_load, which counts against lgkmcnt as well.)
#define FC_OPAQUE_ZERO(z) asm volatile("" : "+s"(z))

// CW = columns per workgroup tile = N of the MFMA: 32 (v_mfma_f32_32x32x2_f32, the throughput shape) or 16 (v_mfma_f32_16x16x4_f32: half the
// matrix work per stage for problems that cannot fill 32-column tiles on every CU — the latency sizes, up to 4,096 columns).  Everything else is the
// same code: lane = (column n = lane % CW, k/row quad hq = lane / CW); a group is four MFMAs over KG = 256/CW consecutive k; an accumulator holds
// NQ = CW²/256 quads of four consecutive output rows: rows 8q + 4hq + e (CW = 32) or 4hq + e (CW = 16) of the row tile, column n.
template <int NZ, int CW = 32>
struct Fc {
    static_assert(CW == 32 || CW == 16, "tile width");
    static constexpr int H = 4 * NZ, NO = NZ - 1;
    static constexpr int LDX = NZ + 4, LDH = H + 4;              // LDS row strides: 16-byte aligned rows, stride/4 odd => conflict-free ds_read_b128
    static constexpr int KG = 256 / CW, NQ = CW * CW / 256, ACCN = 4 * NQ;   // k per group; accumulator quads (4 / 1); accumulator floats
    static constexpr int MT = H / CW, JH = MT / 4;               // row tiles of a hidden layer; jobs per wave
    static constexpr int S_IN = NZ / KG, S_H = H / KG;           // groups per chain: K = NZ, K = 4 NZ
    static constexpr int MT3 = NZ / CW, KS3 = MT3 >= 4 ? 1 : 4 / MT3, G3 = S_H / KS3;   // narrow layer (M = NZ): row tiles, K splits, groups per job
    static_assert(MT3 * KS3 == 4 && JH >= 1, "four jobs of the narrow layer, one per wave");
    static constexpr int F1 = 0, F1_SZ = MT * S_IN * 256;
    static constexpr int F2 = F1 + F1_SZ, F2_SZ = MT * S_H * 256;
    static constexpr int F3 = F2 + F2_SZ, F3_SZ = MT3 * S_H * 256;
    static constexpr int IMG = F3 + F3_SZ;                       // floats per operand image (forward and backward alike)
    static constexpr int BIAS = 2 * H + NZ;
    static constexpr int P = JH * S_IN + JH * S_H + G3;          // groups per stage and wave
    static constexpr int ACT4 = 2 * H + NZ;                      // dwtape_act4: (2H + NZ - 1) rounded up to 4
    static constexpr int R = NZ + 2 * ACT4;                      // floats per column of a delta-tape record (dwtape_row_floats)
    static constexpr int OWN = CW * NZ / 256;                    // state items (column, level) per thread
    static_assert(P % FC_PF == 0, "the ring must close over one stage");
    // ---- COLNDE_MATRIX_BF16X3_EXACT on the 16-column tiles (round 4; the 32-column tiles have their own kernels, engine_fc_split.hip): the same sections
    // on v_mfma_f32_16x16x32_bf16 from exact three-way operand splits (split_bf16.h).  The A operand is pre-split: per (16-row tile, 32-deep k-block) three
    // planes (h, m, l) of [64 lanes][8 bf16], lane (m = lane % 16, kq = lane / 16) holding k = 32 kb + 8 kq + i.  A wave's stream is contiguous per
    // section, k-block outer, its row tiles (jobs) inner, planes innermost: the B operand (8 consecutive floats of the column's activation row, split
    // in registers: 44 vector instructions) is shared by the wave's jobs.  Unit below: one SLOT = one plane fragment (64 lanes x 16 bytes).
    static constexpr int KB_IN = NZ / 32, KB_H = H / 32, KB3 = KB_H / KS3;          // k-blocks per chain: K = NZ, K = 4 NZ, the narrow layer's K part
    static constexpr int SF1 = 0, SF1_SZ = MT * KB_IN * 3 * 256;                    // u32 words
    static constexpr int SF2 = SF1 + SF1_SZ, SF2_SZ = MT * KB_H * 3 * 256;
    static constexpr int SF3 = SF2 + SF2_SZ, SF3_SZ = MT3 * KB_H * 3 * 256;
    static constexpr int SIMG = SF3 + SF3_SZ;                                       // words per split operand image (1.5 x IMG: engine_fc_split's size)
    static constexpr int PS0 = JH * 3 * KB_IN, PS1 = JH * 3 * KB_H, PS2 = 3 * KB3, PS = PS0 + PS1 + PS2;   // slots per stage and wave
    static constexpr int PFS = 12;                                                  // ring depth in slots
    static_assert(CW != 16 || (PS % PFS == 0 && KB_IN >= 1), "the split ring must close over one stage");
    typedef float acc_t __attribute__((ext_vector_type(ACCN)));
    // first row (within the row tile) of accumulator quad q for this lane's hq
    __device__ static constexpr int qrow(int q, int hq) { return (CW == 32 ? 8 * q : 0) + 4 * hq; }
};

// stream position -> layer section (0: K = NZ hidden, 1: K = 4NZ hidden, 2: the narrow layer) and offset in float4 units from the wave's base
template <int NZ, int CW> __host__ __device__ constexpr int fc_sec(int p) {
    p %= Fc<NZ, CW>::P;
    return p < Fc<NZ, CW>::JH * Fc<NZ, CW>::S_IN ? 0 : (p < Fc<NZ, CW>::JH * (Fc<NZ, CW>::S_IN + Fc<NZ, CW>::S_H) ? 1 : 2);
}
template <int NZ, int CW> __host__ __device__ constexpr int fc_off(int p) {
    using S = Fc<NZ, CW>;
    p %= S::P;
    if (p < S::JH * S::S_IN) return (4 * (p / S::S_IN) * S::S_IN + p % S::S_IN) * 64;
    p -= S::JH * S::S_IN;
    if (p < S::JH * S::S_H) return (4 * (p / S::S_H) * S::S_H + p % S::S_H) * 64;
    return (p - S::JH * S::S_H) * 64;
}

// One section of the wave's stream: NJ jobs (output row tiles) of NG groups each, starting at stream position P0.  Per group: four
// MFMAs fed by one ring slot (A: four k-steps of this lane's weight row) and one 16-byte LDS read (B: the same four k of column n);
// the slot is refilled with the group FC_PF positions ahead — possibly the next layer's or the next stage's.
template <int NZ, int CW, int P0, int NJ, int NG, class Epi>
__device__ __forceinline__ void fc_section(f32x4 (&ring)[FC_PF], const f32x4* const (&base)[3], int lane, const float* brow, Epi&& epi) {
    using S = Fc<NZ, CW>;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        typename S::acc_t acc = (typename S::acc_t)(0.0f);
        f32x4 b[2];
        b[0] = *reinterpret_cast<const f32x4*>(brow);
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int p = P0 + j * NG + g;
            if (g + 1 < NG) b[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(brow + S::KG * (g + 1));
            const f32x4 a = ring[p % FC_PF];
            ring[p % FC_PF] = (base[fc_sec<NZ, CW>(p + FC_PF)] + fc_off<NZ, CW>(p + FC_PF))[lane];
            const f32x4 bv = b[g & 1];
            if constexpr (CW == 32) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bv.w, acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bv.w, acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        epi(j, acc);
    }
}

// slot position of the split stream (16-column tiles) -> section, and offset in 16-byte units from the wave's base of that section
template <int NZ> __host__ __device__ constexpr int fc16s_sec(int p) {
    p %= Fc<NZ, 16>::PS;
    return p < Fc<NZ, 16>::PS0 ? 0 : (p < Fc<NZ, 16>::PS0 + Fc<NZ, 16>::PS1 ? 1 : 2);
}
template <int NZ> __host__ __device__ constexpr int fc16s_off(int p) {
    using S = Fc<NZ, 16>;
    p %= S::PS;
    return (p < S::PS0 ? p : (p < S::PS0 + S::PS1 ? p - S::PS0 : p - S::PS0 - S::PS1)) * 64;
}

__device__ __forceinline__ f32x4 fc_mfma16_bf(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// The split twin of fc_section on 16-column tiles: NJ jobs of NKB 32-deep k-blocks, stream position P0 (slots).  Per k-block: two 16-byte LDS reads
// (the 8 floats of column n this lane multiplies), ONE exact three-way split of them (shared by the jobs), and per job three ring slots (the
// pre-split planes of the weight fragment) and six bf16 MFMAs, smallest products first; every slot is refilled PFS positions ahead as it is consumed.
template <int NZ, int P0, int NJ, int NKB, int PFS, class Epi>
__device__ __forceinline__ void fc_section_bf(u32x4 (&ring)[PFS], const u32x4* const (&base)[3], int lane, const float* brow, Epi&& epi) {
    using S = Fc<NZ, 16>;
    static_assert(PFS == S::PFS, "ring depth");
    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[j] = (f32x4)(0.0f);
    // Software pipeline over the k-blocks: the split of block kb + 1 is issued in pieces BETWEEN the jobs' MFMA groups of block kb (one wave per SIMD at the
    // latency sizes: nothing else hides those 44 vector instructions), its eight floats were requested one block earlier still.
    f32x4 lo = *reinterpret_cast<const f32x4*>(brow), hi = *reinterpret_cast<const f32x4*>(brow + 4);
    f32x4 lo1 = lo, hi1 = hi;
    if (NKB > 1) {
        lo1 = *reinterpret_cast<const f32x4*>(brow + 32);
        hi1 = *reinterpret_cast<const f32x4*>(brow + 36);
    }
    Bf3 B;
    {
        const float x8[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        B = bf3_split8(x8);
    }
#pragma unroll
    for (int kb = 0; kb < NKB; kb++) {
        const float n8[8] = {lo1.x, lo1.y, lo1.z, lo1.w, hi1.x, hi1.y, hi1.z, hi1.w};      // block kb + 1 (landed during block kb - 1)
        if (kb + 2 < NKB) {
            lo1 = *reinterpret_cast<const f32x4*>(brow + 32 * (kb + 2));
            hi1 = *reinterpret_cast<const f32x4*>(brow + 32 * (kb + 2) + 4);
        }
        Bf3 Bn = B;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int p = P0 + (kb * NJ + j) * 3;
            const u32x4 Ah = ring[p % PFS], Am = ring[(p + 1) % PFS], Al = ring[(p + 2) % PFS];
            ring[p % PFS] = (base[fc16s_sec<NZ>(p + PFS)] + fc16s_off<NZ>(p + PFS))[lane];
            ring[(p + 1) % PFS] = (base[fc16s_sec<NZ>(p + 1 + PFS)] + fc16s_off<NZ>(p + 1 + PFS))[lane];
            ring[(p + 2) % PFS] = (base[fc16s_sec<NZ>(p + 2 + PFS)] + fc16s_off<NZ>(p + 2 + PFS))[lane];
            f32x4 c = acc[j];
            c = fc_mfma16_bf(Am, B.m, c);
            c = fc_mfma16_bf(Al, B.h, c);
            c = fc_mfma16_bf(Ah, B.l, c);
            c = fc_mfma16_bf(Am, B.h, c);
            c = fc_mfma16_bf(Ah, B.m, c);
            c = fc_mfma16_bf(Ah, B.h, c);
            acc[j] = c;
            __builtin_amdgcn_sched_barrier(0);
            if (kb + 1 < NKB) {
                // this job's share of the next block's split: pairs [4 j / NJ, 4 (j + 1) / NJ)
#pragma unroll
                for (int q = 4 * j / NJ; q < 4 * (j + 1) / NJ; q++) bf3_split_pair(n8[2 * q], n8[2 * q + 1], q, Bn);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        B = Bn;
    }
#pragma unroll
    for (int j = 0; j < NJ; j++) epi(j, acc[j]);
}

// The operand stream of one wave behind one interface: f32 MFMA (fc_section: ring of FC_PF float4 groups) or, SPLIT (16-column tiles), bf16 MFMA on exact
// three-way splits (fc_section_bf: ring of PFS plane fragments).  `rows` is the LDS row of column n (B operand); section 2 starts at the wave's K part.
template <int NZ, int CW, bool SPLIT>
struct FcStream {
    using S = Fc<NZ, CW>;
    static_assert(!SPLIT || CW == 16, "this file's split path is the 16-column one (32 columns: engine_fc_split.hip)");
    typedef typename std::conditional<SPLIT, u32x4, f32x4>::type slot_t;
    static constexpr int DEPTH = SPLIT ? S::PFS : FC_PF;
    slot_t ring[DEPTH];
    const slot_t* base[3];

    __device__ __forceinline__ void init(const void* img, int w, int lane) {
        if constexpr (SPLIT) {
            const u32x4* im = reinterpret_cast<const u32x4*>(img);
            base[0] = im + S::SF1 / 4 + w * S::PS0 * 64;
            base[1] = im + S::SF2 / 4 + w * S::PS1 * 64;
            base[2] = im + S::SF3 / 4 + w * S::PS2 * 64;
#pragma unroll
            for (int q = 0; q < DEPTH; q++) ring[q] = (base[fc16s_sec<NZ>(q)] + fc16s_off<NZ>(q))[lane];
        } else {
            const float* im = reinterpret_cast<const float*>(img);
            base[0] = reinterpret_cast<const f32x4*>(im + S::F1) + (w * S::S_IN) * 64;
            base[1] = reinterpret_cast<const f32x4*>(im + S::F2) + (w * S::S_H) * 64;
            base[2] = reinterpret_cast<const f32x4*>(im + S::F3) + ((w % S::MT3) * S::S_H + (w / S::MT3) * S::G3) * 64;
#pragma unroll
            for (int q = 0; q < DEPTH; q++) ring[q] = (base[fc_sec<NZ, CW>(q)] + fc_off<NZ, CW>(q))[lane];
        }
    }
    // SEC 0: K = NZ hidden section, 1: K = 4 NZ hidden section, 2: the narrow layer (this wave's K part); h = lane / CW
    template <int SEC, class Epi>
    __device__ __forceinline__ void section(const slot_t* const (&sb)[3], int lane, int h, int w, const float* rows, Epi&& epi) {
        if constexpr (SPLIT) {
            if constexpr (SEC == 0) fc_section_bf<NZ, 0, S::JH, S::KB_IN>(ring, sb, lane, rows + 8 * h, epi);
            else if constexpr (SEC == 1) fc_section_bf<NZ, S::PS0, S::JH, S::KB_H>(ring, sb, lane, rows + 8 * h, epi);
            else fc_section_bf<NZ, S::PS0 + S::PS1, 1, S::KB3>(ring, sb, lane, rows + (w / S::MT3) * S::KB3 * 32 + 8 * h, epi);
        } else {
            if constexpr (SEC == 0) fc_section<NZ, CW, 0, S::JH, S::S_IN>(ring, sb, lane, rows + 4 * h, epi);
            else if constexpr (SEC == 1) fc_section<NZ, CW, S::JH * S::S_IN, S::JH, S::S_H>(ring, sb, lane, rows + 4 * h, epi);
            else fc_section<NZ, CW, S::JH * (S::S_IN + S::S_H), 1, S::G3>(ring, sb, lane, rows + (w / S::MT3) * S::G3 * S::KG + 4 * h, epi);
        }
    }
};

// ------------------------------------------------------------------------------------------------
// operand images.  Flux.destructure: W_l[o][i] (out o, in i) at w_off[l] + i*no + o, b_l[o] at b_off[l] + o.
//   forward  section (rows = outputs):  A[row = mt*CW + lane%CW][k = KG*S + 4(lane/CW) + j] = W_l[row][k]
//   backward section (rows = inputs):   A[row = it*CW + lane%CW][k = KG*S + 4(lane/CW) + j] = W_l[k][row]      (zero beyond the matrix)
// image[section][tile][S][lane][j]; backward sections in the order they are used: W3ᵀ (K = NZ), W2ᵀ, W1ᵀ (the narrow one).
// ------------------------------------------------------------------------------------------------
struct FcOffsets { int w[3], b[3]; };

template <int NZ, int CW>
__global__ void __launch_bounds__(256) fc_pack_kernel(FcOffsets o, const float* __restrict__ w, float* __restrict__ imgf, float* __restrict__ imgb,
                                                      float* __restrict__ bias) {
    using S = Fc<NZ, CW>;
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
        const int row = tile * CW + (lane & (CW - 1)), k = S::KG * Sg + 4 * (lane / CW) + j;
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

// The split images of the 16-column tiles (COLNDE_MATRIX_BF16X3_EXACT): the same two operand matrices as fc_pack_kernel's, every weight split exactly into
// three bf16 (x = h + m + l by truncation: split_bf16.h), laid out as each WAVE streams them — image[section][wave][k-block][job][plane][lane][8 bf16]:
// lane (m = lane % 16, kq = lane / 16), element i <-> k = 32 kb + 8 kq + i of row 16 (wave + 4 job) + m (sections 0, 1); section 2: row tile
// wave % MT3, k-blocks (wave / MT3) KB3 + g.
template <int NZ>
__global__ void __launch_bounds__(256) fc_pack_split16_kernel(FcOffsets o, const float* __restrict__ w, u32* __restrict__ simgf, u32* __restrict__ simgb) {
    using S = Fc<NZ, 16>;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < 2 * S::SIMG; idx += gridDim.x * 256) {
        const bool fwd = idx < S::SIMG;
        const int e = fwd ? idx : idx - S::SIMG;
        const int sec = e < S::SF2 ? 0 : (e < S::SF3 ? 1 : 2);
        const int r = e - (sec == 0 ? S::SF1 : (sec == 1 ? S::SF2 : S::SF3));
        const int i2 = r & 3, lane = (r >> 2) & 63, slot = r >> 8;                  // word of the fragment, lane, plane fragment
        const int per_wave = sec == 0 ? S::PS0 : (sec == 1 ? S::PS1 : S::PS2);
        const int wv = slot / per_wave, q = slot - wv * per_wave;
        const int pl = q % 3;
        int tile, kb;
        if (sec < 2) {
            const int j = (q / 3) % S::JH;
            kb = q / (3 * S::JH);
            tile = wv + 4 * j;
        } else {
            tile = wv % S::MT3;
            kb = (wv / S::MT3) * S::KB3 + q / 3;
        }
        const int row = tile * 16 + (lane & 15);
        u32 word = 0;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int k = 32 * kb + 8 * (lane >> 4) + 2 * i2 + t;
            float v = 0.0f;
            if (fwd) {
                const int no = sec == 2 ? S::NO : S::H;                              // sections: W1 (NZ -> H), W2 (H -> H), W3 (H -> NO)
                if (row < no) v = w[o.w[sec] + k * no + row];
            } else {
                const int l = 2 - sec;                                               // sections: W3^T (k = layer-3 outputs), W2^T, W1^T (rows = state levels)
                const int no = l == 2 ? S::NO : S::H;
                if (k < no) v = w[o.w[l] + row * no + k];
            }
            const float vh = __uint_as_float(__float_as_uint(v) & 0xffff0000u);
            const float rr = v - vh;
            const float vm = __uint_as_float(__float_as_uint(rr) & 0xffff0000u);
            const float part = pl == 0 ? vh : (pl == 1 ? vm : rr - vm);
            word |= (__float_as_uint(part) >> 16) << (16 * t);                      // element 2 i2 in the low half (Bf3's order)
        }
        (fwd ? simgf : simgb)[e] = word;
    }
}

// state items owned by a thread: item it = tid + 256 r  ->  (column it / NZ, level it % NZ)
#define FC_OWNER_INDEX()                                                   \
    int oc[S::OWN];                                                        \
    const int oi = tid & (NZ - 1);                                         \
    _Pragma("unroll") for (int r = 0; r < S::OWN; r++) oc[r] = (tid + 256 * r) / NZ

// ------------------------------------------------------------------------------------------------
// forward solve (and, TAPE, the forward half of the tapes)
//   CA:  ConvectiveAdjustmentNDE (convective_adjustment_nde.jl:33-48): the face flux is [b; NN(T); t] - min(0, K dT/dz)
//   RKC: the s-stage RKC2 step of colnde_dev.h (coefficient table `rkc`, increment form: see tile16's forward_kernel) instead of classical RK4
// One record per right-hand-side evaluation: index step * nst + st, nst = 4 (RK4) or s.
// ------------------------------------------------------------------------------------------------
typedef unsigned long long u64;

template <int NZ, int CW, bool TAPE, bool CA, bool RKC, bool SPLIT = false>
__global__ void __launch_bounds__(256, 2)
fc_forward_kernel(const void* __restrict__ imgf, const float* __restrict__ bias, const float* __restrict__ x0, size_t x0_stride,
                  const float* __restrict__ bcs, const float* __restrict__ save_times, int n_save, int iv_begin, int iv_end, int tape_iv0, int substeps, float CN,
                  float caKN, int nst, const float* __restrict__ rkc, float* __restrict__ sol, float* __restrict__ dwtape, u32* __restrict__ masks,
                  u64* __restrict__ swtape, int n_col) {
    // Save intervals [iv_begin, iv_end) of the time axis, starting from x0 (column stride x0_stride: the initial state, or — a time SEGMENT
    // of the gradient path — the state the tape-less pass saved at save point iv_begin; restarting there is exact: the saved state IS xn).
    // Only the intervals from tape_iv0 on are taped (the records are numbered from its first step): the tape-less pass of a time-segmented
    // gradient tapes its LAST segment on the way, which that segment's own pass would otherwise have to repeat.
    using S = Fc<NZ, CW>;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & (CW - 1), h = lane / CW;               // column of the tile; k / row quad
    float* X = fc_smem;                          // [32][LDX]   stage input
    float* A1 = X + CW * S::LDX;                 // [32][LDH]   relu(W1 x + b1)
    float* A2 = A1 + CW * S::LDH;                // [32][LDH]   relu(W2 a1 + b2)
    float* PART = A1;                            // [KS3][32][NZ] partial sums of the last layer (a1 is dead by then)
    float* BL = A2 + CW * S::LDH;                // [2H + NZ] biases (a global load in an epilogue would be waited for with vmcnt(0): the ring too)
    for (int q = tid; q < S::BIAS; q += 256) BL[q] = bias[q];
    const int col0 = blockIdx.x * CW;
    FC_OWNER_INDEX();

    typedef FcStream<NZ, CW, SPLIT> Stream;
    Stream strm;
    strm.init(imgf, w, lane);

    float xn[S::OWN], vst[S::OWN], kv[S::OWN], bcb[S::OWN], bct[S::OWN];
#pragma unroll
    for (int r = 0; r < S::OWN; r++) {
        const int col = min(col0 + oc[r], n_col - 1);
        xn[r] = x0[(size_t)col * x0_stride + oi];
        bcb[r] = bcs[(size_t)col * 2];
        bct[r] = bcs[(size_t)col * 2 + 1];
        kv[r] = 0.0f;
        if (sol && iv_begin == 0 && col0 + oc[r] < n_col) sol[((size_t)(col0 + oc[r]) * n_save) * NZ + oi] = xn[r];
    }
    const float b3v = oi < S::NO ? bias[2 * S::H + oi] : 0.0f;
    // every load issued so far is consumed HERE: a register still "in flight" at the loop header makes the wait-count pass put a
    // vmcnt(0) at the top of every stage, which would drain the prefetch ring each time
#pragma unroll
    for (int r = 0; r < S::OWN; r++) asm volatile("" :: "v"(xn[r]), "v"(bcb[r]), "v"(bct[r]));
    asm volatile("" :: "v"(b3v));
    const int n_steps = (iv_end - tape_iv0) * substeps;          // taped steps (and, x nst, records per tile) of this launch
    const int step_t0 = (tape_iv0 - iv_begin) * substeps;         // first taped step

    // one right-hand-side evaluation: stage input vst[] (owner layout) -> kv[]; qi = record index step * nst + st
    auto rhs = [&](int qs) {
        const int qi = qs - step_t0 * nst;
        const bool tp = TAPE && qi >= 0;                              // wave-uniform
        int zero = 0;
        FC_OPAQUE_ZERO(zero);
        const typename Stream::slot_t* const sb[3] = {strm.base[0] + zero, strm.base[1] + zero, strm.base[2] + zero};
        const size_t ri = (size_t)blockIdx.x * n_steps * nst + qi;
        float* rec = tp ? dwtape + ri * ((size_t)CW * S::R) : nullptr;
        u32* mrec = tp ? masks + ri * 512 + w * 64 + lane : nullptr;
        // ---- stage input (owner layout) -> LDS rows, tape
#pragma unroll
        for (int r = 0; r < S::OWN; r++) {
            X[oc[r] * S::LDX + oi] = vst[r];
            if (tp) FC_STORE(vst[r], rec + (size_t)oc[r] * S::R + oi);
        }
        FC_BARRIER();
        // ---- hidden layers: z = W a + b on 32x32x2 MFMA, relu, rows to LDS (next layer's B operand) and to the tape
        auto hidden = [&](int l /* 1, 2 */, float* dstrows, int j, const typename S::acc_t& acc) {
            const int mt = w + 4 * j;
            u32 bits = 0;
#pragma unroll
            for (int q = 0; q < S::NQ; q++) {
                const int f = mt * CW + S::qrow(q, h);
                const f32x4 bq = *reinterpret_cast<const f32x4*>(BL + (l - 1) * S::H + f);
                f32x4 a;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float z = acc[4 * q + e] + bq[e];
                    a[e] = fmaxf(z, 0.0f);
                    bits |= (z > 0.0f ? 1u : 0u) << (4 * q + e);
                }
                *reinterpret_cast<f32x4*>(dstrows + n * S::LDH + f) = a;
                if (tp) FC_STORE(a, reinterpret_cast<f32x4*>(rec + (size_t)n * S::R + NZ + (l - 1) * S::H + f));
            }
            return bits;
        };
        {
            u32 mb = 0;
            strm.template section<0>(sb, lane, h, w, X + n * S::LDX,
                                     [&](int j, const typename S::acc_t& acc) { mb |= hidden(1, A1, j, acc) << (S::ACCN * j); });
            if (tp) FC_STORE(mb, mrec);
        }
        FC_BARRIER();
        {
            u32 mb = 0;
            strm.template section<1>(sb, lane, h, w, A1 + n * S::LDH,
                                     [&](int j, const typename S::acc_t& acc) { mb |= hidden(2, A2, j, acc) << (S::ACCN * j); });
            if (tp) FC_STORE(mb, mrec + 256);
        }
        FC_BARRIER();
        // ---- output layer: row tile w % MT3, K part w / MT3; partial sums to LDS
        strm.template section<2>(sb, lane, h, w, A2 + n * S::LDH,
            [&](int, const typename S::acc_t& acc) {
                float* pr = PART + ((w / S::MT3) * CW + n) * NZ + (w % S::MT3) * CW;
#pragma unroll
                for (int q = 0; q < S::NQ; q++) {
                    const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                    *reinterpret_cast<f32x4*>(pr + S::qrow(q, h)) = v;
                }
            });
        FC_BARRIER();
        // ---- physics: faces F = [b; NN(T); t] (free_convection_nde.jl:29-38) [- min(0, K dT/dz) on the interior faces,
        //      convective_adjustment_nde.jl:43-47], dT = -C Nz (F[i+1] - F[i])
#pragma unroll
        for (int r = 0; r < S::OWN; r++) {
            float o = b3v;
#pragma unroll
            for (int ks = 0; ks < S::KS3; ks++) o += PART[(ks * CW + oc[r]) * NZ + oi];
            const float olo = __shfl_up(o, 1);                                // NN output of face i (lane i - 1 holds it)
            float wlo = oi == 0 ? bcb[r] : olo;
            float whi = oi == NZ - 1 ? bct[r] : o;
            if (CA) {
                const float vlo = __shfl_up(vst[r], 1), vhi = __shfl_down(vst[r], 1);
                const float glo = (vst[r] - vlo) * (float)NZ, ghi = (vhi - vst[r]) * (float)NZ;     // dT/dz on faces i and i + 1
                const bool on = oi >= 1 && glo < 0.0f;
                if (oi >= 1) wlo -= fminf(0.0f, caKN * (vst[r] - vlo));
                if (oi <= NZ - 2) whi -= fminf(0.0f, caKN * (vhi - vst[r]));
                (void)ghi;
                if (tp) {
                    // the switch pattern of the stage, one bit per face, for the pullback
                    const u64 bal = __ballot(on);
                    const u64 mine = NZ == 64 ? bal : (lane < 32 ? (bal & 0xffffffffull) : (bal >> 32));
                    if (oi == 0) swtape[ri * CW + oc[r]] = mine;
                }
            }
            kv[r] = -CN * (whi - wlo);
        }
    };

    int step = 0;
    if constexpr (!RKC) {
        float ac[S::OWN];
#pragma unroll
        for (int r = 0; r < S::OWN; r++) ac[r] = 0.0f;
        for (int iv = iv_begin; iv < iv_end; iv++) {
            const float dt = (save_times[iv + 1] - save_times[iv]) / (float)substeps;
            for (int s = 0; s < substeps; s++, step++) {
#pragma nounroll
                for (int st = 0; st < 4; st++) {
                    const float ca = st == 0 ? 0.0f : (st == 3 ? 1.0f : 0.5f);            // stage abscissa
                    const float cbp = st == 1 ? 1.0f / 6.0f : 1.0f / 3.0f;                 // RK4 weight of k_{st-1}
#pragma unroll
                    for (int r = 0; r < S::OWN; r++) {
                        float v = xn[r];
                        if (st > 0) {
                            ac[r] += cbp * kv[r];
                            v += ca * dt * kv[r];
                        }
                        vst[r] = v;
                    }
                    rhs(step * 4 + st);
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
    } else {
        // Y_0 = xn, d_j = Y_j - Y_0 (increments: float32 stays accurate), F_0 = f0; stage st evaluates F_st = f(Y_st); Y_s ends the step
        const float* mu_t = rkc, *nu_t = rkc + RKC_LD, *mut_t = rkc + 2 * RKC_LD, *gat_t = rkc + 3 * RKC_LD;
        float ym1[S::OWN], ym2[S::OWN], f0[S::OWN];
#pragma unroll
        for (int r = 0; r < S::OWN; r++) { ym1[r] = 0.0f; ym2[r] = 0.0f; f0[r] = 0.0f; }
        for (int iv = iv_begin; iv < iv_end; iv++) {
            const float dt = (save_times[iv + 1] - save_times[iv]) / (float)substeps;
            for (int s = 0; s < substeps; s++, step++) {
#pragma nounroll
                for (int st = 0; st <= nst; st++) {      // st = nst: only the final combination Y_s
                    const float cmu = mu_t[st], cnu = nu_t[st], cmt = mut_t[st] * dt, cga = gat_t[st] * dt;
                    const bool last = st == nst;
                    const bool save = last && s == substeps - 1;
#pragma unroll
                    for (int r = 0; r < S::OWN; r++) {
                        float dj = 0.0f;
                        if (st == 1) {
                            f0[r] = kv[r];
                            dj = cmt * f0[r];
                        } else if (st >= 2) {
                            dj = cmu * ym1[r] + cnu * ym2[r] + cmt * kv[r] + cga * f0[r];
                        }
                        const float v = xn[r] + dj;
                        ym2[r] = st == 0 ? 0.0f : ym1[r];
                        ym1[r] = dj;
                        if (last) {
                            xn[r] = v;
                            if (save && sol && col0 + oc[r] < n_col) sol[((size_t)(col0 + oc[r]) * n_save + iv + 1) * NZ + oi] = v;
                        } else {
                            vst[r] = v;
                        }
                    }
                    if (last) break;
                    rhs(step * nst + st);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// embedded inference: compute_neural_network_forcing! (free_convection/double_gyre_nn.jl:149-168; BASELINE configs[4]) — one evaluation of
// the network per column and the divergence of the flux it predicts.  The same sections as the forward solve, a workgroup walking over
// tiles (gridDim.x workgroups, tile += gridDim.x) so that the A-operand ring keeps streaming from one tile to the next.
//   T̂ = T_scaling(19.65 + T/20) (:156-158), wT = enforce_fluxes(inv(wT_scaling)(NN(T̂)), 0, surface_flux) (:160), forcing = -∂z wT (:135)
// ------------------------------------------------------------------------------------------------
template <int NZ, int CW>
__global__ void __launch_bounds__(256, 2)
fc_infer_kernel(const float* __restrict__ imgf, const float* __restrict__ bias, const float* __restrict__ T, const float* __restrict__ top_flux,
                float mu_T, float inv_sig_T, float sig_wT, float mu_wT, float inv_dz, float* __restrict__ out, int n_col) {
    using S = Fc<NZ, CW>;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & (CW - 1), h = lane / CW;               // column of the tile; k / row quad
    float* X = fc_smem;
    float* A1 = X + CW * S::LDX;
    float* A2 = A1 + CW * S::LDH;
    float* PART = A1;
    float* BL = A2 + CW * S::LDH;
    for (int q = tid; q < S::BIAS; q += 256) BL[q] = bias[q];
    FC_OWNER_INDEX();
    const f32x4* base[3];
    base[0] = reinterpret_cast<const f32x4*>(imgf + S::F1) + (w * S::S_IN) * 64;
    base[1] = reinterpret_cast<const f32x4*>(imgf + S::F2) + (w * S::S_H) * 64;
    base[2] = reinterpret_cast<const f32x4*>(imgf + S::F3) + ((w % S::MT3) * S::S_H + (w / S::MT3) * S::G3) * 64;
    f32x4 ring[FC_PF];
#pragma unroll
    for (int q = 0; q < FC_PF; q++) ring[q] = (base[fc_sec<NZ, CW>(q)] + fc_off<NZ, CW>(q))[lane];
    const float b3v = oi < S::NO ? bias[2 * S::H + oi] : 0.0f;
    asm volatile("" :: "v"(b3v));
    const int n_tiles = (n_col + CW - 1) / CW;
#pragma nounroll
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int zero = 0;
        FC_OPAQUE_ZERO(zero);
        const f32x4* const sb[3] = {base[0] + zero, base[1] + zero, base[2] + zero};
        const int col0 = tile * CW;
        float tf[S::OWN];
#pragma unroll
        for (int r = 0; r < S::OWN; r++) {
            const int col = min(col0 + oc[r], n_col - 1);
            X[oc[r] * S::LDX + oi] = ((19.65f + T[(size_t)col * NZ + oi] / 20.0f) - mu_T) * inv_sig_T;
            tf[r] = top_flux[col];
        }
        FC_BARRIER();
        auto hidden = [&](int l, float* dstrows, int j, const typename S::acc_t& acc) {
            const int mt = w + 4 * j;
#pragma unroll
            for (int q = 0; q < S::NQ; q++) {
                const int f = mt * CW + S::qrow(q, h);
                const f32x4 bq = *reinterpret_cast<const f32x4*>(BL + (l - 1) * S::H + f);
                f32x4 a;
#pragma unroll
                for (int e = 0; e < 4; e++) a[e] = fmaxf(acc[4 * q + e] + bq[e], 0.0f);
                *reinterpret_cast<f32x4*>(dstrows + n * S::LDH + f) = a;
            }
        };
        fc_section<NZ, CW, 0, S::JH, S::S_IN>(ring, sb, lane, X + n * S::LDX + 4 * h, [&](int j, const typename S::acc_t& acc) { hidden(1, A1, j, acc); });
        FC_BARRIER();
        fc_section<NZ, CW, S::JH * S::S_IN, S::JH, S::S_H>(ring, sb, lane, A1 + n * S::LDH + 4 * h, [&](int j, const typename S::acc_t& acc) { hidden(2, A2, j, acc); });
        FC_BARRIER();
        fc_section<NZ, CW, S::JH * (S::S_IN + S::S_H), 1, S::G3>(ring, sb, lane, A2 + n * S::LDH + (w / S::MT3) * S::G3 * S::KG + 4 * h,
            [&](int, const typename S::acc_t& acc) {
                float* pr = PART + ((w / S::MT3) * CW + n) * NZ + (w % S::MT3) * CW;
#pragma unroll
                for (int q = 0; q < S::NQ; q++) {
                    const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                    *reinterpret_cast<f32x4*>(pr + S::qrow(q, h)) = v;
                }
            });
        FC_BARRIER();
#pragma unroll
        for (int r = 0; r < S::OWN; r++) {
            float o = b3v;
#pragma unroll
            for (int ks = 0; ks < S::KS3; ks++) o += PART[(ks * CW + oc[r]) * NZ + oi];
            const float wT = sig_wT * o + mu_wT;                                   // face oi + 1
            const float below = __shfl_up(wT, 1);                                  // face oi
            const float lo = oi == 0 ? 0.0f : below;
            const float hi = oi == NZ - 1 ? tf[r] : wT;
            if (col0 + oc[r] < n_col) out[(size_t)(col0 + oc[r]) * NZ + oi] = -(hi - lo) * inv_dz;
        }
        FC_BARRIER();                                                               // PART (= A1's rows) is rewritten by the next tile's first layer
    }
}

// ------------------------------------------------------------------------------------------------
// adjoint: back-propagation through the stages from the taped relu (and switch) bits; fills the dz part of the delta-tape records.
// RKC: the discrete adjoint of the RKC2 recurrence as in tile16's adjoint_kernel — cotangents of Y_j (lam), Y_{j-1}, Y_{j-2}, Y_0 and F_0
// per state item — with ONE convective-adjustment switch pattern per step (that of Y_{s-1}, the first stage the backward sweep meets:
// DESIGN §2 "a finding about discrete adjoints of stabilised steppers").
// ------------------------------------------------------------------------------------------------
struct FcGrad { int b[3]; int n_params; };

template <int NZ, int CW, bool CA, bool RKC, bool SPLIT = false>
__global__ void __launch_bounds__(256, 2)
fc_adjoint_kernel(const void* __restrict__ imgb, const float* __restrict__ save_times, int n_save, int iv_begin, int iv_end, int substeps, float CN,
                  float caKN, int nst, const float* __restrict__ rkc, const float* __restrict__ sol, const float* __restrict__ truth,
                  float* __restrict__ dwtape, const u32* __restrict__ masks, const u64* __restrict__ swtape, float w_loss, float* __restrict__ lam_io,
                  float* __restrict__ slab, FcGrad go, int n_col) {
    // Save intervals [iv_begin, iv_end), backwards.  lam_io [columns][NZ] (or null: one launch covers the axis) carries λ from one time
    // segment to the one before it: read unless this is the last segment of the axis, written unless it is the first.
    using S = Fc<NZ, CW>;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & (CW - 1), h = lane / CW;               // column of the tile; k / row quad
    float* DZ2 = fc_smem;                        // [32][LDH]
    float* DZ1 = DZ2 + CW * S::LDH;              // [32][LDH]
    float* DZ3 = DZ1;                            // [32][LDX]   dead before dz1 is written
    float* XBP = DZ2;                            // [KS3][32][NZ] partial sums of W1ᵀ dz1 (dz2 is dead by then)
    const int col0 = blockIdx.x * CW;
    FC_OWNER_INDEX();

    typedef FcStream<NZ, CW, SPLIT> Stream;
    Stream strm;
    strm.init(imgb, w, lane);

    float lam[S::OWN], xb[S::OWN], kb[S::OWN], db3[S::OWN];
    u32 swp = 0;                                 // switch bits of this thread's items: bit 2r = face oi, bit 2r + 1 = face oi + 1 of item r
    float db2 = 0.0f, db1 = 0.0f;                // bias gradients of hidden unit tid (< H): column sums of the dz rows, taken from LDS
    float sumsq = 0.0f;
#pragma unroll
    for (int r = 0; r < S::OWN; r++) {
        lam[r] = 0.0f; xb[r] = 0.0f; db3[r] = 0.0f; kb[r] = 0.0f;
        if (lam_io && iv_end < n_save - 1) lam[r] = lam_io[(size_t)(col0 + oc[r]) * NZ + oi];
        if (iv_begin == 0 && col0 + oc[r] < n_col) {                // save point 0 enters the loss value only
            const size_t q = ((size_t)(col0 + oc[r]) * n_save) * NZ + oi;
            const float d = sol[q] - truth[q];
            sumsq += d * d;
        }
    }
    const int n_steps = (iv_end - iv_begin) * substeps;          // steps (and, x nst, records per tile) of this launch

    // pullback of one right-hand-side evaluation: stage cotangent kb[] (owner layout) -> xb[] = J(Y)ᵀ kb; qi = record index
    auto pull = [&](int qi) {
        int zero = 0;
        FC_OPAQUE_ZERO(zero);
        const typename Stream::slot_t* const sb[3] = {strm.base[0] + zero, strm.base[1] + zero, strm.base[2] + zero};
        const size_t ri = (size_t)blockIdx.x * n_steps * nst + qi;
        float* rec = dwtape + ri * ((size_t)CW * S::R);
        const u32* mrec = masks + ri * 512 + w * 64 + lane;
        const u32 m1 = FC_LOAD(mrec), m2 = FC_LOAD(mrec + 256);
        // ---- physics pullback: dz3[i] = C Nz (k̄[i+1] - k̄[i]) on the Nz-1 interior faces; CA: x̄ += Dᶠᵀ(switch ∘ (-K) ∘ that)
        float xph[S::OWN];
#pragma unroll
        for (int r = 0; r < S::OWN; r++) {
            const float kn = __shfl_down(kb[r], 1);
            const float dz = oi < S::NO ? CN * (kn - kb[r]) : 0.0f;                 // face i + 1
            xph[r] = 0.0f;
            if (CA) {
                const float dlo = __shfl_up(dz, 1);                                 // face i
                const float ghi = (oi < S::NO && ((swp >> (2 * r + 1)) & 1u)) ? -dz * caKN : 0.0f;
                const float glo = (oi >= 1 && ((swp >> (2 * r)) & 1u)) ? -dlo * caKN : 0.0f;
                xph[r] = glo - ghi;
            }
            DZ3[oc[r] * S::LDX + oi] = dz;
            FC_STORE(dz, rec + (size_t)oc[r] * S::R + NZ + S::ACT4 + 2 * S::H + oi);
            db3[r] += dz;
        }
        FC_BARRIER();
        auto hidden = [&](int l /* 2, 1: layer whose dz this is */, float* dstrows, u32 bits, int j, const typename S::acc_t& acc) {
            const int mt = w + 4 * j;
#pragma unroll
            for (int q = 0; q < S::NQ; q++) {
                f32x4 d;
#pragma unroll
                for (int e = 0; e < 4; e++) d[e] = ((bits >> (S::ACCN * j + 4 * q + e)) & 1u) ? acc[4 * q + e] : 0.0f;
                const int f = mt * CW + S::qrow(q, h);
                *reinterpret_cast<f32x4*>(dstrows + n * S::LDH + f) = d;
                FC_STORE(d, reinterpret_cast<f32x4*>(rec + (size_t)n * S::R + NZ + S::ACT4 + (l - 1) * S::H + f));
            }
        };
        // ---- dz2 = relu'(z2) ∘ W3ᵀ dz3
        strm.template section<0>(sb, lane, h, w, DZ3 + n * S::LDX,
                                          [&](int j, const typename S::acc_t& acc) { hidden(2, DZ2, m2, j, acc); });
        FC_BARRIER();
        // bias gradients: hidden unit tid's column sum of the finished dz rows, straight from LDS (16 accumulator registers per row tile
        // and layer — 64 at Nz = 64 — would otherwise ride along in every lane)
        auto colsum = [&](const float* rows) {
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
            for (int c = 0; c < CW; c += 4) {
                a0 += rows[(c + 0) * S::LDH + tid];
                a1 += rows[(c + 1) * S::LDH + tid];
                a2 += rows[(c + 2) * S::LDH + tid];
                a3 += rows[(c + 3) * S::LDH + tid];
            }
            return (a0 + a1) + (a2 + a3);
        };
        if (S::H == 256 || tid < S::H) db2 += colsum(DZ2);
        // ---- dz1 = relu'(z1) ∘ W2ᵀ dz2
        strm.template section<1>(sb, lane, h, w, DZ2 + n * S::LDH,
                                                       [&](int j, const typename S::acc_t& acc) { hidden(1, DZ1, m1, j, acc); });
        FC_BARRIER();
        if (S::H == 256 || tid < S::H) db1 += colsum(DZ1);
        // ---- x̄ = W1ᵀ dz1: row tile w % MT3, K part w / MT3
        strm.template section<2>(sb, lane, h, w, DZ1 + n * S::LDH,
            [&](int, const typename S::acc_t& acc) {
                float* pr = XBP + ((w / S::MT3) * CW + n) * NZ + (w % S::MT3) * CW;
#pragma unroll
                for (int q = 0; q < S::NQ; q++) {
                    const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                    *reinterpret_cast<f32x4*>(pr + S::qrow(q, h)) = v;
                }
            });
        FC_BARRIER();
#pragma unroll
        for (int r = 0; r < S::OWN; r++) {
            float v = xph[r];
#pragma unroll
            for (int ks = 0; ks < S::KS3; ks++) v += XBP[(ks * CW + oc[r]) * NZ + oi];
            xb[r] = v;
        }
        // (the next evaluation writes DZ3 = DZ1's rows: every wave's reads of DZ1 ended before the barrier above; XBP = DZ2's rows
        //  are next written two barriers from here)
    };
    auto load_switch = [&](int qi) {
        if (CA) {
            const size_t ri = (size_t)blockIdx.x * n_steps * nst + qi;
            swp = 0;
#pragma unroll
            for (int r = 0; r < S::OWN; r++) swp |= (u32)((swtape[ri * CW + oc[r]] >> oi) & 3ull) << (2 * r);
        }
    };

    const float* mu_t = rkc, *nu_t = rkc + RKC_LD, *mut_t = rkc + 2 * RKC_LD, *gat_t = rkc + 3 * RKC_LD, *kap_t = rkc + 5 * RKC_LD;
    float xbs[S::OWN], yb1[S::OWN], yb2[S::OWN], yb0[S::OWN], f0b[S::OWN];
#pragma unroll
    for (int r = 0; r < S::OWN; r++) { xbs[r] = 0.0f; yb1[r] = 0.0f; yb2[r] = 0.0f; yb0[r] = 0.0f; f0b[r] = 0.0f; }
    for (int iv = iv_end - 1; iv >= iv_begin; iv--) {
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
            const int step = (iv - iv_begin) * substeps + s;
            if constexpr (!RKC) {
#pragma unroll
                for (int r = 0; r < S::OWN; r++) xbs[r] = 0.0f;
#pragma nounroll
                for (int st = 3; st >= 0; st--) {
                    // k̄4 = dt/6 λ; k̄3 = dt/3 λ + dt x̄4; k̄2 = dt/3 λ + dt/2 x̄3; k̄1 = dt/6 λ + dt/2 x̄2
                    const float cwl = (st == 0 || st == 3) ? dt / 6.0f : dt / 3.0f;
                    const float cwx = st == 3 ? 0.0f : (st == 2 ? dt : 0.5f * dt);
#pragma unroll
                    for (int r = 0; r < S::OWN; r++) kb[r] = cwl * lam[r] + cwx * xb[r];
                    load_switch(step * 4 + st);                 // RK4: every stage's own pattern (the exact discrete adjoint)
                    pull(step * 4 + st);
#pragma unroll
                    for (int r = 0; r < S::OWN; r++) xbs[r] += xb[r];
                }
#pragma unroll
                for (int r = 0; r < S::OWN; r++) lam[r] += xbs[r];
            } else {
                load_switch(step * nst + nst - 1);              // one switch pattern per step: that of Y_{s-1}
#pragma nounroll
                for (int st = nst - 1; st >= 0; st--) {
                    // stage input Y_st feeds Y_j, j = st + 1, through mu~_j h F_st
                    const float cmu = mu_t[st + 1], cnu = nu_t[st + 1], cmt = mut_t[st + 1] * dt, cga = gat_t[st + 1] * dt, ck0 = kap_t[st + 1];
#pragma unroll
                    for (int r = 0; r < S::OWN; r++) {
                        // lam = cotangent of Y_j, complete once the previous iteration's pullback (xb: J(Y_j)ᵀ F̄_j) is added
                        if (st < nst - 1) {
                            const float yj = yb1[r] + xb[r];
                            yb1[r] = yb2[r];
                            yb2[r] = 0.0f;
                            lam[r] = yj;
                        }
                        if (st >= 1) {
                            yb0[r] += ck0 * lam[r];
                            yb1[r] += cmu * lam[r];
                            yb2[r] += cnu * lam[r];
                            f0b[r] += cga * lam[r];
                            kb[r] = cmt * lam[r];
                        } else {
                            // Y_1 = Y_0 + mu~_1 h F_0: lam holds Ȳ_1, yb1 the nu_2 part of Ȳ_0
                            yb0[r] += lam[r] + yb1[r];
                            kb[r] = f0b[r] + cmt * lam[r];
                            yb1[r] = 0.0f;
                            f0b[r] = 0.0f;
                        }
                    }
                    pull(step * nst + st);
                }
                // λ_n = Ȳ_0 + J(Y_0)ᵀ F̄_0
#pragma unroll
                for (int r = 0; r < S::OWN; r++) {
                    lam[r] = yb0[r] + xb[r];
                    yb0[r] = 0.0f;
                }
            }
        }
    }
    if (lam_io && iv_begin > 0)
#pragma unroll
        for (int r = 0; r < S::OWN; r++) lam_io[(size_t)(col0 + oc[r]) * NZ + oi] = lam[r];
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
    if (tid < S::H) {
        out[go.b[0] + tid] = db1;
        out[go.b[1] + tid] = db2;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool fc_supported(const DevModel& m, int stepper) {
    const bool fc = m.model == COLNDE_MODEL_FREE_CONVECTION, ca = m.model == COLNDE_MODEL_CONV_ADJ_NDE;
    if (!(fc && stepper == COLNDE_STEPPER_RK4) && !(ca && (stepper == COLNDE_STEPPER_RK4 || stepper == COLNDE_STEPPER_RKC2))) return false;
    if (m.Nz != 32 && m.Nz != 64) return false;
    if (m.n_layers != 3 || m.n_nets != 1) return false;
    if (m.sizes[0] != m.Nz || m.sizes[1] != 4 * m.Nz || m.sizes[2] != 4 * m.Nz || m.sizes[3] != m.Nz - 1) return false;
    return m.acts[0] == COLNDE_ACT_RELU && m.acts[1] == COLNDE_ACT_RELU && m.acts[2] == COLNDE_ACT_IDENTITY;
}

size_t fc_image_floats(int Nz) { return Nz == 64 ? Fc<64>::IMG : Fc<32>::IMG; }        // (the same for both tile widths)
size_t fc_bias_floats(int Nz) { return Nz == 64 ? Fc<64>::BIAS : Fc<32>::BIAS; }
size_t fc_record_row_floats(int Nz) { return Nz == 64 ? Fc<64>::R : Fc<32>::R; }

template <int NZ, int CW> static size_t fc_lds_fwd() { return (size_t)(CW * Fc<NZ, CW>::LDX + 2 * CW * Fc<NZ, CW>::LDH + Fc<NZ, CW>::BIAS) * sizeof(float); }
template <int NZ, int CW> static size_t fc_lds_adj() { return (size_t)(2 * CW * Fc<NZ, CW>::LDH) * sizeof(float); }

// the instantiated (levels, tile width) x (model, stepper) combinations: FreeConvectionNDE x RK4; ConvectiveAdjustmentNDE x {RK4, RKC2}
#define FC_FOR_EACH_SHAPE(M, ...) M(64, 32, __VA_ARGS__) M(32, 32, __VA_ARGS__) M(64, 16, __VA_ARGS__) M(32, 16, __VA_ARGS__)
#define FC_FOR_EACH_FWD(M) FC_FOR_EACH_SHAPE(M, true, false, false) FC_FOR_EACH_SHAPE(M, false, false, false) \
                           FC_FOR_EACH_SHAPE(M, true, true, false) FC_FOR_EACH_SHAPE(M, false, true, false)   \
                           FC_FOR_EACH_SHAPE(M, true, true, true) FC_FOR_EACH_SHAPE(M, false, true, true)
#define FC_FOR_EACH_ADJ(M) FC_FOR_EACH_SHAPE(M, false, false) FC_FOR_EACH_SHAPE(M, true, false) FC_FOR_EACH_SHAPE(M, true, true)
// ... and their COLNDE_MATRIX_BF16X3_EXACT twins on 16-column tiles (this file; the 32-column ones: engine_fc_split.hip)
#define FC_FOR_EACH_SHAPE16(M, ...) M(64, 16, __VA_ARGS__) M(32, 16, __VA_ARGS__)
#define FC_FOR_EACH_FWD16(M) FC_FOR_EACH_SHAPE16(M, true, false, false) FC_FOR_EACH_SHAPE16(M, false, false, false) \
                             FC_FOR_EACH_SHAPE16(M, true, true, false) FC_FOR_EACH_SHAPE16(M, false, true, false)   \
                             FC_FOR_EACH_SHAPE16(M, true, true, true) FC_FOR_EACH_SHAPE16(M, false, true, true)
#define FC_FOR_EACH_ADJ16(M) FC_FOR_EACH_SHAPE16(M, false, false) FC_FOR_EACH_SHAPE16(M, true, false) FC_FOR_EACH_SHAPE16(M, true, true)

hipError_t fc_set_kernel_attributes() {
    hipError_t e;
#define FC_ATTR_F(N, W, T, C, K) if ((e = hipFuncSetAttribute((const void*)(fc_forward_kernel<N, W, T, C, K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fc_lds_fwd<N, W>())) != hipSuccess) return e;
#define FC_ATTR_A(N, W, C, K) if ((e = hipFuncSetAttribute((const void*)(fc_adjoint_kernel<N, W, C, K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fc_lds_adj<N, W>())) != hipSuccess) return e;
#define FC_ATTR_I(N, W, X) if ((e = hipFuncSetAttribute((const void*)(fc_infer_kernel<N, W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fc_lds_fwd<N, W>())) != hipSuccess) return e;
    FC_FOR_EACH_FWD(FC_ATTR_F)
    FC_FOR_EACH_ADJ(FC_ATTR_A)
    FC_FOR_EACH_SHAPE(FC_ATTR_I, 0)
#define FC_ATTR_FS(N, W, T, C, K) if ((e = hipFuncSetAttribute((const void*)(fc_forward_kernel<N, W, T, C, K, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fc_lds_fwd<N, W>())) != hipSuccess) return e;
#define FC_ATTR_AS(N, W, C, K) if ((e = hipFuncSetAttribute((const void*)(fc_adjoint_kernel<N, W, C, K, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fc_lds_adj<N, W>())) != hipSuccess) return e;
    FC_FOR_EACH_FWD16(FC_ATTR_FS)
    FC_FOR_EACH_ADJ16(FC_ATTR_AS)
#undef FC_ATTR_FS
#undef FC_ATTR_AS
#undef FC_ATTR_F
#undef FC_ATTR_A
#undef FC_ATTR_I
    return fcs_set_kernel_attributes();           // the split twins (engine_fc_split.hip)
}

// tile width for a problem of n_col columns: 16 while 32-column tiles could not put a workgroup on every CU twice over (COLNDE_FC_CW=16|32 forces)
int fc_tile_width(int n_col) {
    const char* e = getenv("COLNDE_FC_CW");
    if (e && (atoi(e) == 16 || atoi(e) == 32)) return atoi(e);
    return n_col <= 4096 ? 16 : 32;
}

hipError_t fc_launch_pack(const DevModel& m, int cw, const float* w, float* imgf, float* imgb, float* bias, unsigned int* simgf, unsigned int* simgb,
                          hipStream_t stream) {
    FcOffsets o;
    for (int l = 0; l < 3; l++) { o.w[l] = m.w_off[l]; o.b[l] = m.b_off[l]; }
    if (simgf && simgb && cw == 32) {                      // the split images of COLNDE_MATRIX_BF16X3_EXACT beside the f32 ones (the biases are shared)
        const hipError_t es = fcs_launch_pack(m, w, simgf, simgb, stream);
        if (es != hipSuccess) return es;
    } else if (simgf && simgb && cw == 16) {               // ... in the 16-column kernels' stream order (same size)
        if (m.Nz == 64) hipLaunchKernelGGL((fc_pack_split16_kernel<64>), dim3(256), dim3(256), 0, stream, o, w, simgf, simgb);
        else hipLaunchKernelGGL((fc_pack_split16_kernel<32>), dim3(128), dim3(256), 0, stream, o, w, simgf, simgb);
    }
    bool launched = false;
#define FC_PACK(N, W, X) if (!launched && m.Nz == N && cw == W) { hipLaunchKernelGGL((fc_pack_kernel<N, W>), dim3(N == 64 ? 256 : 128), dim3(256), 0, stream, o, w, imgf, imgb, bias); launched = true; }
    FC_FOR_EACH_SHAPE(FC_PACK, 0)
#undef FC_PACK
    return launched ? hipGetLastError() : hipErrorInvalidValue;
}

hipError_t fc_launch_infer(const DevModel& m, int cw, const float* imgf, const float* bias, const float* T, const float* top_flux, float inv_dz,
                           float* out, int n_col, hipStream_t stream) {
    if (n_col < 1) return hipErrorInvalidValue;
    const int n_tiles = (n_col + cw - 1) / cw;
    const dim3 grid(n_tiles < 512 ? n_tiles : 512), block(256);                  // two workgroups per CU, each walking over its tiles
    bool launched = false;
#define FC_INF(N, W, X) if (!launched && m.Nz == N && cw == W) { hipLaunchKernelGGL((fc_infer_kernel<N, W>), grid, block, (fc_lds_fwd<N, W>()), stream, imgf, bias, T, top_flux, m.mu_T, 1.0f / m.sig_T, m.sig_wT, m.mu_wT, inv_dz, out, n_col); launched = true; }
    FC_FOR_EACH_SHAPE(FC_INF, 0)
#undef FC_INF
    return launched ? hipGetLastError() : hipErrorInvalidValue;
}

hipError_t fc_launch_forward(const DevModel& m, int cw, const float* imgf, const unsigned int* simgf, const float* bias, const float* x0, size_t x0_stride,
                             const float* bcs, const float* save_times, int n_save, int iv_begin, int iv_end, int tape_iv0, int substeps, float* sol,
                             float* dwtape, unsigned int* masks, unsigned long long* swtape, int n_col, hipStream_t stream) {
    if (n_col < 1 || iv_begin < 0 || iv_end > n_save - 1 || iv_begin >= iv_end || tape_iv0 < iv_begin || tape_iv0 >= iv_end) return hipErrorInvalidValue;
    if (simgf && cw == 32) {                                // COLNDE_MATRIX_BF16X3_EXACT: the same solve on the bf16 pipe (engine_fc_split.hip)
        if (dwtape && (!masks || (m.model == COLNDE_MODEL_CONV_ADJ_NDE && !swtape))) return hipErrorInvalidValue;
        if (m.rkc && m.model != COLNDE_MODEL_CONV_ADJ_NDE) return hipErrorInvalidValue;
        return fcs_launch_forward(m, simgf, bias, x0, x0_stride, bcs, save_times, n_save, iv_begin, iv_end, tape_iv0, substeps, sol, dwtape, masks, swtape, n_col, stream);
    }
    const dim3 grid((n_col + cw - 1) / cw), block(256);
    const float CN = m.C_fc * (float)m.Nz, caKN = m.ca_K * (float)m.Nz;
    const bool tape = dwtape != nullptr, ca = m.model == COLNDE_MODEL_CONV_ADJ_NDE, rk = m.rkc != nullptr;
    if (tape && (!masks || (ca && !swtape))) return hipErrorInvalidValue;
    if (rk && !ca) return hipErrorInvalidValue;
    bool launched = false;
    if (simgf && cw == 16) {                                // ... and on 16-column tiles: the SPLIT instantiations of this file's kernels
#define FC_FWDS(N, W, T, C, K)                                                                                                                       \
    if (!launched && m.Nz == N && tape == T && ca == C && rk == K) {                                                                                 \
        hipLaunchKernelGGL((fc_forward_kernel<N, W, T, C, K, true>), grid, block, (fc_lds_fwd<N, W>()), stream, (const void*)simgf, bias, x0, x0_stride, bcs, save_times, n_save, \
                           iv_begin, iv_end, tape_iv0, substeps, CN, caKN, m.nst, m.rkc, sol, dwtape, masks, swtape, n_col);                         \
        launched = true;                                                                                                                             \
    }
        FC_FOR_EACH_FWD16(FC_FWDS)
#undef FC_FWDS
        return launched ? hipGetLastError() : hipErrorInvalidValue;
    }
#define FC_FWD(N, W, T, C, K)                                                                                                                        \
    if (!launched && m.Nz == N && cw == W && tape == T && ca == C && rk == K) {                                                                      \
        hipLaunchKernelGGL((fc_forward_kernel<N, W, T, C, K>), grid, block, (fc_lds_fwd<N, W>()), stream, (const void*)imgf, bias, x0, x0_stride, bcs, save_times, n_save, \
                           iv_begin, iv_end, tape_iv0, substeps, CN, caKN, m.nst, m.rkc, sol, dwtape, masks, swtape, n_col);                         \
        launched = true;                                                                                                                             \
    }
    FC_FOR_EACH_FWD(FC_FWD)
#undef FC_FWD
    return launched ? hipGetLastError() : hipErrorInvalidValue;
}

hipError_t fc_launch_adjoint(const DevModel& m, int cw, const float* imgb, const unsigned int* simgb, const float* save_times, int n_save, int iv_begin, int iv_end,
                             int substeps, const float* sol, const float* truth, float* dwtape, const unsigned int* masks, const unsigned long long* swtape,
                             float w_loss, float* lam_io, float* slab, int n_col, hipStream_t stream) {
    if (n_col < 1 || !dwtape || !masks || iv_begin < 0 || iv_end > n_save - 1 || iv_begin >= iv_end) return hipErrorInvalidValue;
    if ((iv_begin > 0 || iv_end < n_save - 1) && !lam_io) return hipErrorInvalidValue;
    if (simgb && cw == 32) {
        if ((m.model == COLNDE_MODEL_CONV_ADJ_NDE && !swtape) || (m.rkc && m.model != COLNDE_MODEL_CONV_ADJ_NDE)) return hipErrorInvalidValue;
        return fcs_launch_adjoint(m, simgb, save_times, n_save, iv_begin, iv_end, substeps, sol, truth, dwtape, masks, swtape, w_loss, lam_io, slab, n_col, stream);
    }
    const dim3 grid((n_col + cw - 1) / cw), block(256);
    const float CN = m.C_fc * (float)m.Nz, caKN = m.ca_K * (float)m.Nz;
    const bool ca = m.model == COLNDE_MODEL_CONV_ADJ_NDE, rk = m.rkc != nullptr;
    if ((ca && !swtape) || (rk && !ca)) return hipErrorInvalidValue;
    FcGrad go;
    for (int l = 0; l < 3; l++) go.b[l] = m.b_off[l];
    go.n_params = m.n_params;
    bool launched = false;
    if (simgb && cw == 16) {
#define FC_ADJS(N, W, C, K)                                                                                                                          \
    if (!launched && m.Nz == N && ca == C && rk == K) {                                                                                              \
        hipLaunchKernelGGL((fc_adjoint_kernel<N, W, C, K, true>), grid, block, (fc_lds_adj<N, W>()), stream, (const void*)simgb, save_times, n_save, iv_begin, iv_end, substeps, \
                           CN, caKN, m.nst, m.rkc, sol, truth, dwtape, masks, swtape, w_loss, lam_io, slab, go, n_col);                              \
        launched = true;                                                                                                                             \
    }
        FC_FOR_EACH_ADJ16(FC_ADJS)
#undef FC_ADJS
        return launched ? hipGetLastError() : hipErrorInvalidValue;
    }
#define FC_ADJ(N, W, C, K)                                                                                                                           \
    if (!launched && m.Nz == N && cw == W && ca == C && rk == K) {                                                                                   \
        hipLaunchKernelGGL((fc_adjoint_kernel<N, W, C, K>), grid, block, (fc_lds_adj<N, W>()), stream, (const void*)imgb, save_times, n_save, iv_begin, iv_end, substeps, \
                           CN, caKN, m.nst, m.rkc, sol, truth, dwtape, masks, swtape, w_loss, lam_io, slab, go, n_col);                              \
        launched = true;                                                                                                                             \
    }
    FC_FOR_EACH_ADJ(FC_ADJ)
#undef FC_ADJ
    return launched ? hipGetLastError() : hipErrorInvalidValue;
}
