// engine_fc_split.hip — the fc32 engine under COLNDE_MATRIX_BF16X3_EXACT: the free-convection NDE kernels of engine_fc.hip with every dense layer on
// v_mfma_f32_32x32x16_bf16 from EXACT three-way bf16 operand splits (split_bf16.h: x = h + m + l, six part-products per 16-deep k-block, f32 accumulate).
//
// Same model coverage, tapes, slab rows and results layout as engine_fc.hip (FreeConvectionNDE / ConvectiveAdjustmentNDE, the reference's
// Dense(Nz,4Nz,relu) -> Dense(4Nz,4Nz,relu) -> Dense(4Nz,Nz-1), Nz = 32 | 64; free_convection/src/free_convection_nde.jl:29-38,
// convective_adjustment_nde.jl:33-48, train_free_convection_nde.jl:119-121) — a handle can switch between the two kernel families between calls.
//
// What bounds these kernels is instruction issue, not the matrix pipe (measured on the first version of this file's kernels, which kept the f32
// activation rows of engine_fc.hip in LDS and split them in every wave that read them: 5.3 vector instructions per bf16 MFMA, 4x redundant
// splitting, a pipe 46 % busy).  So the operands are split ONCE, where they are produced:
//  * a workgroup owns 32 columns and has ONE WAVE PER 32-ROW TILE of a hidden layer (8 waves at Nz = 64, 4 at Nz = 32); the narrow layer's row
//    tiles are split in K over the waves, partial sums through LDS, added in a fixed order;
//  * activations live in LDS as three bf16 PLANES per value, rows [column][feature]: a wave's epilogue splits the 16 values per lane it has just
//    produced (88 vector instructions) and the next layer's B operand is three ds_read_b128 per k-block — no arithmetic between LDS and MFMA;
//  * weights are pre-split (fc_pack_split_kernel) into the order each wave streams them, [section][wave][k-block][plane][lane][8 bf16], fetched
//    through a register ring PFS fragments ahead that runs across layers, barriers and stages (as in engine_fc.hip);
//  * the f32 activations go to the delta-tape records, relu's derivative to the bit tape, exactly as engine_fc.hip writes them.
// LDS: 115-118 KB at Nz = 64 (one 512-thread workgroup per CU, two waves per SIMD), 60 KB at Nz = 32 (two 256-thread workgroups per CU).
#include "engine_fc.h"
#include "split_bf16.h"

typedef unsigned int u32;
typedef unsigned short u16;
typedef unsigned long long u64;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float fs16 __attribute__((ext_vector_type(16)));

extern __shared__ float fcs_smem[];

#ifndef FCS_PFS64
#define FCS_PFS64 12                               // Nz = 64: ring depth in plane fragments = four k-blocks ahead (18 and 24 measured no faster; 24 spills)
#endif
#define FCS_TSTORE(v, p) (*(p) = (v))              // tape stores: plain (non-temporal ones measured 43 -> 58 ms: they are acknowledged late and vmcnt is in order)
#define FCS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define FCS_OPAQUE_ZERO(z) asm volatile("" : "+s"(z))      // keeps the stream's group addresses "scalar base + immediate + lane" (see engine_fc.hip)

template <int NZ>
struct Fs {
    static constexpr int H = 4 * NZ, NO = NZ - 1;
    static constexpr int NW = H / 32, NT = 64 * NW;                       // waves = row tiles of a hidden layer; threads
    static constexpr int MT3 = NZ / 32, KS3 = NW / MT3;                   // narrow layer (M = NZ): row tiles, K parts
    static constexpr int KB_IN = NZ / 16, KB_H = H / 16, KB3 = KB_H / KS3;   // 16-deep k-blocks per chain
    static constexpr int PS0 = 3 * KB_IN, PS1 = 3 * KB_H, PS2 = 3 * KB3, PS = PS0 + PS1 + PS2;   // plane fragments (slots) per wave and stage
    static constexpr int SF1 = 0, SF2 = SF1 + NW * PS0 * 256, SF3 = SF2 + NW * PS1 * 256, SIMG = SF3 + NW * PS2 * 256;   // u32 words
    static constexpr int PFS = NZ == 64 ? FCS_PFS64 : 18;                 // ring depth in slots (three per k-block)
    static_assert(PS % PFS == 0, "the ring must close over one stage");
    static constexpr int LDXB = 2 * NZ + 16, LDHB = 2 * H + 16;           // bytes per column row of a plane: 16-byte aligned, an odd number of 16-byte units
    static constexpr int PXB = 32 * LDXB, PHB = 32 * LDHB;                // bytes per plane
    static constexpr int ACT4 = 2 * H + NZ, R = NZ + 2 * ACT4;            // the delta-tape record of tile16 / engine_fc.hip
    static constexpr int OWN = 32 * NZ / NT;                              // state items (column, level) per thread
    static constexpr int BIAS = 2 * H + NZ;
    static constexpr int STG = 32 * 36;                                   // floats of a wave's tape staging tile [32 columns][32 features + 4]
    static constexpr size_t LDS_FWD = 3 * PXB + 6 * PHB + BIAS * 4 + NW * STG * 4, LDS_ADJ = 3 * PXB + 6 * PHB + NW * STG * 4;
};

// slot position of the stream -> section (0: K = NZ hidden, 1: K = 4NZ hidden, 2: the narrow layer) and offset in 16-byte units from the wave's section base
template <int NZ> __host__ __device__ constexpr int fcs_sec(int p) {
    p %= Fs<NZ>::PS;
    return p < Fs<NZ>::PS0 ? 0 : (p < Fs<NZ>::PS0 + Fs<NZ>::PS1 ? 1 : 2);
}
template <int NZ> __host__ __device__ constexpr int fcs_off(int p) {
    using S = Fs<NZ>;
    p %= S::PS;
    return (p < S::PS0 ? p : (p < S::PS0 + S::PS1 ? p - S::PS0 : p - S::PS0 - S::PS1)) * 64;
}

// One section of a wave's stream: one job (a 32-row output tile) of NKB k-blocks starting at stream position P0.  Per k-block: three ring slots (the
// weight fragment's planes, refilled PFS positions ahead as they are consumed), three 16-byte LDS reads (column n's planes of the same 16 k, one
// k-block ahead) and six bf16 MFMAs.  bp: plane 0 of column n at this lane's k half; PB: bytes from one plane to the next.
// fill(kb): work that depends on nothing in this section (the previous tile's tape stores), placed in front of k-block kb's products: these kernels are
// bound by instruction issue in their epilogue phases — both waves of a SIMD are in the same phase — while the issue port is three quarters idle here.
template <int NZ, int P0, int NKB, int PB, class Fill, class Epi>
__device__ __forceinline__ void fcs_section(u32x4 (&ring)[Fs<NZ>::PFS], const u32x4* const (&base)[3], int lane, const char* bp, Fill&& fill, Epi&& epi) {
    using S = Fs<NZ>;
    constexpr int PFS = S::PFS;
    fs16 acc = (fs16)(0.0f);
    Bf3 Bn;
    Bn.h = *reinterpret_cast<const u32x4*>(bp);
    Bn.m = *reinterpret_cast<const u32x4*>(bp + PB);
    Bn.l = *reinterpret_cast<const u32x4*>(bp + 2 * PB);
#pragma unroll
    for (int kb = 0; kb < NKB; kb++) {
        const Bf3 B = Bn;
        if (kb + 1 < NKB) {
            Bn.h = *reinterpret_cast<const u32x4*>(bp + 32 * (kb + 1));
            Bn.m = *reinterpret_cast<const u32x4*>(bp + 32 * (kb + 1) + PB);
            Bn.l = *reinterpret_cast<const u32x4*>(bp + 32 * (kb + 1) + 2 * PB);
        }
        fill(kb);
        const int p = P0 + 3 * kb;
        Bf3 A;
        A.h = ring[p % PFS];
        A.m = ring[(p + 1) % PFS];
        A.l = ring[(p + 2) % PFS];
        ring[p % PFS] = (base[fcs_sec<NZ>(p + PFS)] + fcs_off<NZ>(p + PFS))[lane];
        ring[(p + 1) % PFS] = (base[fcs_sec<NZ>(p + 1 + PFS)] + fcs_off<NZ>(p + 1 + PFS))[lane];
        ring[(p + 2) % PFS] = (base[fcs_sec<NZ>(p + 2 + PFS)] + fcs_off<NZ>(p + 2 + PFS))[lane];
        acc = mfma_bf3(A, B, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    epi(acc);
}

// exact three-way split of four consecutive features: three planes of two packed bf16 pairs each (element 2p in the low half)
__device__ __forceinline__ void fcs_split4(const f32x4& a, u32x2& h, u32x2& m, u32x2& l) {
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const float x = a[2 * p], y = a[2 * p + 1];
        const float rx = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u), ry = y - __uint_as_float(__float_as_uint(y) & 0xffff0000u);
        const float lx = rx - __uint_as_float(__float_as_uint(rx) & 0xffff0000u), ly = ry - __uint_as_float(__float_as_uint(ry) & 0xffff0000u);
        h[p] = __builtin_amdgcn_perm(__float_as_uint(y), __float_as_uint(x), 0x07060302u);
        m[p] = __builtin_amdgcn_perm(__float_as_uint(ry), __float_as_uint(rx), 0x07060302u);
        l[p] = __builtin_amdgcn_perm(__float_as_uint(ly), __float_as_uint(lx), 0x07060302u);
    }
}
// ... of one value, written as three bf16 at byte address a of plane 0
template <int PB>
__device__ __forceinline__ void fcs_split1_store(float v, char* a) {
    const float r = v - __uint_as_float(__float_as_uint(v) & 0xffff0000u);
    const float lo = r - __uint_as_float(__float_as_uint(r) & 0xffff0000u);
    *reinterpret_cast<u16*>(a) = (u16)(__float_as_uint(v) >> 16);
    *reinterpret_cast<u16*>(a + PB) = (u16)(__float_as_uint(r) >> 16);
    *reinterpret_cast<u16*>(a + 2 * PB) = (u16)(__float_as_uint(lo) >> 16);
}
// A wave's 32 x 32 tile of f32 tape values (hidden activations, hidden deltas) leaves through a staging tile of its own in LDS: the accumulator layout
// has a lane hold 4 features of ONE column, and a record row is 2.3 KB long — stored from there, every 64-lane store would be 64 scattered 16-byte
// pieces.  Read back as [row = lane / 8 + 8 i][16-byte chunk lane % 8], a store instruction covers eight whole 128-byte lines.  (Measured, Nz = 64,
// 16,384 columns: taping costs the forward kernel 9 of its 43 ms whichever way the stores are issued — scattered or whole lines, in the epilogue or
// deferred into the next section's products; non-temporal: 58 ms.  The lines written are the 77 GB the dW GEMM needs.)
template <int R>
__device__ __forceinline__ void fcs_tape_tile(const float* stg, float* rec_tile /* record row of column 0 at the tile's first feature */, int lane) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = (lane >> 3) + 8 * i;
        const f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * 36 + 4 * (lane & 7));
        FCS_TSTORE(v, reinterpret_cast<f32x4*>(rec_tile + (size_t)row * R + 4 * (lane & 7)));
    }
}

// the planes of a quad of features written beside each other: plane 0 at byte address a
template <int PB>
__device__ __forceinline__ void fcs_split4_store(const f32x4& v, char* a) {
    u32x2 h, m, l;
    fcs_split4(v, h, m, l);
    *reinterpret_cast<u32x2*>(a) = h;
    *reinterpret_cast<u32x2*>(a + PB) = m;
    *reinterpret_cast<u32x2*>(a + 2 * PB) = l;
}

// ------------------------------------------------------------------------------------------------
// The split operand images: both operand matrices of engine_fc.hip's images (forward: W1, W2, W3; backward: W3ᵀ, W2ᵀ, W1ᵀ, zero beyond the matrix),
// every weight split exactly into three bf16, laid out as each wave streams them — image[section][wave][k-block][plane][lane][8 bf16]: lane (m = lane % 32,
// kh = lane / 32), element i <-> k = 16 kb + 8 kh + i of row 32 wave + m (sections 0, 1); section 2: row tile wave % MT3, k-blocks (wave / MT3) KB3 + g.
// Flux.destructure: W_l[o][i] (out o, in i) at w_off[l] + i*no + o.
// ------------------------------------------------------------------------------------------------
struct FcsOffsets { int w[3], b[3]; };

template <int NZ>
__global__ void __launch_bounds__(256) fcs_pack_kernel(FcsOffsets o, const float* __restrict__ w, u32* __restrict__ simgf, u32* __restrict__ simgb) {
    using S = Fs<NZ>;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < 2 * S::SIMG; idx += gridDim.x * 256) {
        const bool fwd = idx < S::SIMG;
        const int e = fwd ? idx : idx - S::SIMG;
        const int sec = e < S::SF2 ? 0 : (e < S::SF3 ? 1 : 2);
        const int r = e - (sec == 0 ? S::SF1 : (sec == 1 ? S::SF2 : S::SF3));
        const int i2 = r & 3, lane = (r >> 2) & 63, slot = r >> 8;                  // word of the fragment, lane, plane fragment
        const int per_wave = sec == 0 ? S::PS0 : (sec == 1 ? S::PS1 : S::PS2);
        const int wv = slot / per_wave, q = slot - wv * per_wave;
        const int pl = q % 3;
        const int tile = sec < 2 ? wv : wv % S::MT3;
        const int kb = sec < 2 ? q / 3 : (wv / S::MT3) * S::KB3 + q / 3;
        const int row = tile * 32 + (lane & 31);
        u32 word = 0;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int k = 16 * kb + 8 * (lane >> 5) + 2 * i2 + t;
            float v = 0.0f;
            if (fwd) {
                const int no = sec == 2 ? S::NO : S::H;                              // sections: W1 (NZ -> H), W2 (H -> H), W3 (H -> NO)
                if (row < no) v = w[o.w[sec] + k * no + row];
            } else {
                const int l = 2 - sec;                                               // sections: W3ᵀ (k = layer-3 outputs), W2ᵀ, W1ᵀ (rows = state levels)
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

// state items owned by a thread: item it = tid + NT r  ->  (column it / NZ, level it % NZ)
#define FCS_OWNER_INDEX()                                                  \
    int oc[S::OWN];                                                        \
    const int oi = tid & (NZ - 1);                                         \
    _Pragma("unroll") for (int r = 0; r < S::OWN; r++) oc[r] = (tid + S::NT * r) / NZ

// the wave's stream: section bases and the primed ring
#define FCS_STREAM(img)                                                                                       \
    const u32x4* base[3];                                                                                     \
    base[0] = reinterpret_cast<const u32x4*>(img) + S::SF1 / 4 + w * S::PS0 * 64;                             \
    base[1] = reinterpret_cast<const u32x4*>(img) + S::SF2 / 4 + w * S::PS1 * 64;                             \
    base[2] = reinterpret_cast<const u32x4*>(img) + S::SF3 / 4 + w * S::PS2 * 64;                             \
    u32x4 ring[S::PFS];                                                                                       \
    _Pragma("unroll") for (int q = 0; q < S::PFS; q++) ring[q] = (base[fcs_sec<NZ>(q)] + fcs_off<NZ>(q))[lane]

// ------------------------------------------------------------------------------------------------
// forward solve (and, TAPE, the forward half of the tapes): engine_fc.hip's fc_forward_kernel, argument for argument
// ------------------------------------------------------------------------------------------------
template <int NZ, bool TAPE, bool CA, bool RKC>
__global__ void __launch_bounds__(Fs<NZ>::NT, 2)
fcs_forward_kernel(const u32* __restrict__ simgf, const float* __restrict__ bias, const float* __restrict__ x0, size_t x0_stride,
                   const float* __restrict__ bcs, const float* __restrict__ save_times, int n_save, int iv_begin, int iv_end, int tape_iv0, int substeps, float CN,
                   float caKN, int nst, const float* __restrict__ rkc, float* __restrict__ sol, float* __restrict__ dwtape, u32* __restrict__ masks,
                   u64* __restrict__ swtape, int n_col) {
    using S = Fs<NZ>;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, kh = lane >> 5;                   // column of the tile; k half (operands) / row quad half (accumulators)
    char* XP = reinterpret_cast<char*>(fcs_smem);              // [3][32][LDXB]  stage input, planes
    char* A1P = XP + 3 * S::PXB;                               // [3][32][LDHB]  relu(W1 x + b1), planes
    char* A2P = A1P + 3 * S::PHB;                              // [3][32][LDHB]  relu(W2 a1 + b2), planes
    float* PART = reinterpret_cast<float*>(A1P);               // [KS3][32][NZ]  partial sums of the last layer (a1 is dead by then)
    float* BL = reinterpret_cast<float*>(A2P + 3 * S::PHB);    // [2H + NZ] biases
    float* STG = BL + S::BIAS + w * S::STG;                    // this wave's tape staging tile
    for (int q = tid; q < S::BIAS; q += S::NT) BL[q] = bias[q];
    const int col0 = blockIdx.x * 32;
    FCS_OWNER_INDEX();
    FCS_STREAM(simgf);

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
    // every load issued so far is consumed HERE (a register still in flight at the loop header puts a vmcnt(0) at the top of every stage: engine_fc.hip)
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
        FCS_OPAQUE_ZERO(zero);
        const u32x4* const sb[3] = {base[0] + zero, base[1] + zero, base[2] + zero};
        const size_t ri = (size_t)blockIdx.x * n_steps * nst + qi;
        float* rec = tp ? dwtape + ri * ((size_t)32 * S::R) : nullptr;
        // relu bits: engine_fc.hip's [layer 2][wave 4][lane 64] dwords hold row tiles w and w + 4 in their halves: this wave's tile is one u16 of them
        u16* mrec = tp ? reinterpret_cast<u16*>(masks + ri * 512 + (w & 3) * 64 + lane) + (w >> 2) : nullptr;
        // ---- stage input (owner layout) -> planes in LDS, tape
#pragma unroll
        for (int r = 0; r < S::OWN; r++) {
            fcs_split1_store<S::PXB>(vst[r], XP + oc[r] * S::LDXB + 2 * oi);
            if (tp) FCS_TSTORE(vst[r], rec + (size_t)oc[r] * S::R + oi);
        }
        FCS_BARRIER();
        // ---- hidden layers: z = W a + b, relu, planes to LDS (next layer's B operand), f32 rows and derivative bits to the tape
        u32 bits1 = 0, bits2 = 0;
        auto hidden = [&](int l /* 1, 2 */, char* dstP, u32& bits, const fs16& acc) {
            bits = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int f = 32 * w + 8 * q + 4 * kh;
                const f32x4 bq = *reinterpret_cast<const f32x4*>(BL + (l - 1) * S::H + f);
                f32x4 a;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float z = acc[4 * q + e] + bq[e];
                    a[e] = fmaxf(z, 0.0f);
                    bits |= (z > 0.0f ? 1u : 0u) << (4 * q + e);
                }
                fcs_split4_store<S::PHB>(a, dstP + n * S::LDHB + 2 * f);
                if (tp) *reinterpret_cast<f32x4*>(STG + n * 36 + 8 * q + 4 * kh) = a;
            }
        };
        // the tile a hidden layer's epilogue left in the staging area goes to the tape from inside the NEXT section (k-block 1), with its derivative bits
        auto tape_tile = [&](int l, u32 bits, int kb) {
            if (tp && kb == 1) {
                fcs_tape_tile<S::R>(STG, rec + NZ + (l - 1) * S::H + 32 * w, lane);
                FCS_TSTORE((u16)bits, mrec + (l - 1) * 512);            // (u16 units: 256 dwords per layer)
            }
        };
        fcs_section<NZ, 0, S::KB_IN, S::PXB>(ring, sb, lane, XP + n * S::LDXB + 16 * kh, [](int) {}, [&](const fs16& acc) { hidden(1, A1P, bits1, acc); });
        FCS_BARRIER();
        fcs_section<NZ, S::PS0, S::KB_H, S::PHB>(ring, sb, lane, A1P + n * S::LDHB + 16 * kh, [&](int kb) { tape_tile(1, bits1, kb); },
                                                 [&](const fs16& acc) { hidden(2, A2P, bits2, acc); });
        FCS_BARRIER();
        // ---- output layer: row tile w % MT3, K part w / MT3; partial sums to LDS
        fcs_section<NZ, S::PS0 + S::PS1, S::KB3, S::PHB>(ring, sb, lane, A2P + n * S::LDHB + (w / S::MT3) * S::KB3 * 32 + 16 * kh,
                                                         [&](int kb) { tape_tile(2, bits2, kb); }, [&](const fs16& acc) {
            float* pr = PART + ((w / S::MT3) * 32 + n) * NZ + (w % S::MT3) * 32;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                *reinterpret_cast<f32x4*>(pr + 8 * q + 4 * kh) = v;
            }
        });
        FCS_BARRIER();
        // ---- physics: faces F = [b; NN(T); t] (free_convection_nde.jl:29-38) [- min(0, K dT/dz) on the interior faces,
        //      convective_adjustment_nde.jl:43-47], dT = -C Nz (F[i+1] - F[i])
#pragma unroll
        for (int r = 0; r < S::OWN; r++) {
            float o = b3v;
#pragma unroll
            for (int ks = 0; ks < S::KS3; ks++) o += PART[(ks * 32 + oc[r]) * NZ + oi];
            const float olo = __shfl_up(o, 1);                                // NN output of face i (lane i - 1 holds it)
            float wlo = oi == 0 ? bcb[r] : olo;
            float whi = oi == NZ - 1 ? bct[r] : o;
            if (CA) {
                const float vlo = __shfl_up(vst[r], 1), vhi = __shfl_down(vst[r], 1);
                const float glo = (vst[r] - vlo) * (float)NZ;                                       // dT/dz on face i
                const bool on = oi >= 1 && glo < 0.0f;
                if (oi >= 1) wlo -= fminf(0.0f, caKN * (vst[r] - vlo));
                if (oi <= NZ - 2) whi -= fminf(0.0f, caKN * (vhi - vst[r]));
                if (tp) {
                    // the switch pattern of the stage, one bit per face, for the pullback
                    const u64 bal = __ballot(on);
                    const u64 mine = NZ == 64 ? bal : (lane < 32 ? (bal & 0xffffffffull) : (bal >> 32));
                    if (oi == 0) swtape[ri * 32 + oc[r]] = mine;
                }
            }
            kv[r] = -CN * (whi - wlo);
        }
        // (the next evaluation's first barrier stands between these reads of PART = A1P's rows and the next layer-1 epilogue's writes)
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
// adjoint: engine_fc.hip's fc_adjoint_kernel, argument for argument.  Bias gradients of the hidden layers are kept per lane (one register per
// accumulator element: this wave's 32 units x this lane's column) and summed over the columns once, at the end.
// ------------------------------------------------------------------------------------------------
struct FcsGrad { int b[3]; int n_params; };

template <int NZ, bool CA, bool RKC>
__global__ void __launch_bounds__(Fs<NZ>::NT, 2)
fcs_adjoint_kernel(const u32* __restrict__ simgb, const float* __restrict__ save_times, int n_save, int iv_begin, int iv_end, int substeps, float CN,
                   float caKN, int nst, const float* __restrict__ rkc, const float* __restrict__ sol, const float* __restrict__ truth,
                   float* __restrict__ dwtape, const u32* __restrict__ masks, const u64* __restrict__ swtape, float w_loss, float* __restrict__ lam_io,
                   float* __restrict__ slab, FcsGrad go, int n_col) {
    using S = Fs<NZ>;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, kh = lane >> 5;
    char* DZ3P = reinterpret_cast<char*>(fcs_smem);            // [3][32][LDXB]  dz3 (the Nz-1 interior faces; slot Nz-1 zero), planes
    char* DZ2P = DZ3P + 3 * S::PXB;                            // [3][32][LDHB]
    char* DZ1P = DZ2P + 3 * S::PHB;                            // [3][32][LDHB]
    float* XBP = reinterpret_cast<float*>(DZ2P);               // [KS3][32][NZ] partial sums of W1ᵀ dz1 (dz2 is dead by then)
    float* STG = reinterpret_cast<float*>(DZ1P + 3 * S::PHB) + w * S::STG;      // this wave's tape staging tile
    const int col0 = blockIdx.x * 32;
    FCS_OWNER_INDEX();
    FCS_STREAM(simgb);

    float lam[S::OWN], xb[S::OWN], kb[S::OWN], db3[S::OWN];
    float db1a[16], db2a[16];                    // bias-gradient sums of this wave's hidden units over time, for this lane's column
    u32 swp = 0;                                 // switch bits of this thread's items: bit 2r = face oi, bit 2r + 1 = face oi + 1 of item r
    float sumsq = 0.0f;
#pragma unroll
    for (int q = 0; q < 16; q++) { db1a[q] = 0.0f; db2a[q] = 0.0f; }
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
        FCS_OPAQUE_ZERO(zero);
        const u32x4* const sb[3] = {base[0] + zero, base[1] + zero, base[2] + zero};
        const size_t ri = (size_t)blockIdx.x * n_steps * nst + qi;
        float* rec = dwtape + ri * ((size_t)32 * S::R);
        const u16* mrec = reinterpret_cast<const u16*>(masks + ri * 512 + (w & 3) * 64 + lane) + (w >> 2);
        const u32 m1 = mrec[0], m2 = mrec[512];
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
            fcs_split1_store<S::PXB>(dz, DZ3P + oc[r] * S::LDXB + 2 * oi);
            FCS_TSTORE(dz, rec + (size_t)oc[r] * S::R + NZ + S::ACT4 + 2 * S::H + oi);
            db3[r] += dz;
        }
        FCS_BARRIER();
        auto hidden = [&](int l /* 2, 1: layer whose dz this is */, char* dstP, u32 bits, float (&dba)[16], const fs16& acc) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f32x4 d;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    d[e] = ((bits >> (4 * q + e)) & 1u) ? acc[4 * q + e] : 0.0f;
                    dba[4 * q + e] += d[e];
                }
                const int f = 32 * w + 8 * q + 4 * kh;
                fcs_split4_store<S::PHB>(d, dstP + n * S::LDHB + 2 * f);
                *reinterpret_cast<f32x4*>(STG + n * 36 + 8 * q + 4 * kh) = d;
            }
        };
        // the delta tile an epilogue left in the staging area goes to the tape from inside the NEXT section (k-block 1)
        auto tape_tile = [&](int l, int kb) {
            if (kb == 1) fcs_tape_tile<S::R>(STG, rec + NZ + S::ACT4 + (l - 1) * S::H + 32 * w, lane);
        };
        // ---- dz2 = relu'(z2) ∘ W3ᵀ dz3
        fcs_section<NZ, 0, S::KB_IN, S::PXB>(ring, sb, lane, DZ3P + n * S::LDXB + 16 * kh, [](int) {}, [&](const fs16& acc) { hidden(2, DZ2P, m2, db2a, acc); });
        FCS_BARRIER();
        // ---- dz1 = relu'(z1) ∘ W2ᵀ dz2
        fcs_section<NZ, S::PS0, S::KB_H, S::PHB>(ring, sb, lane, DZ2P + n * S::LDHB + 16 * kh, [&](int kb) { tape_tile(2, kb); },
                                                 [&](const fs16& acc) { hidden(1, DZ1P, m1, db1a, acc); });
        FCS_BARRIER();
        // ---- x̄ = W1ᵀ dz1: row tile w % MT3, K part w / MT3
        fcs_section<NZ, S::PS0 + S::PS1, S::KB3, S::PHB>(ring, sb, lane, DZ1P + n * S::LDHB + (w / S::MT3) * S::KB3 * 32 + 16 * kh,
                                                         [&](int kb) { tape_tile(1, kb); }, [&](const fs16& acc) {
            float* pr = XBP + ((w / S::MT3) * 32 + n) * NZ + (w % S::MT3) * 32;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                *reinterpret_cast<f32x4*>(pr + 8 * q + 4 * kh) = v;
            }
        });
        FCS_BARRIER();
#pragma unroll
        for (int r = 0; r < S::OWN; r++) {
            float v = xph[r];
#pragma unroll
            for (int ks = 0; ks < S::KS3; ks++) v += XBP[(ks * 32 + oc[r]) * NZ + oi];
            xb[r] = v;
        }
        // (XBP = DZ2P's rows are next written two barriers from here, by the next evaluation's first epilogue)
    };
    auto load_switch = [&](int qi) {
        if (CA) {
            const size_t ri = (size_t)blockIdx.x * n_steps * nst + qi;
            swp = 0;
#pragma unroll
            for (int r = 0; r < S::OWN; r++) swp |= (u32)((swtape[ri * 32 + oc[r]] >> oi) & 3ull) << (2 * r);
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
    FCS_BARRIER();
    float* out = slab + (size_t)blockIdx.x * (go.n_params + 8);
    float* scr = fcs_smem;                                            // [NW][NZ] + [NW]
    {
        float s3 = 0.0f;
#pragma unroll
        for (int r = 0; r < S::OWN; r++) s3 += db3[r];              // this thread's columns, level oi
        if (NZ == 32) s3 += __shfl_down(s3, 32);                     // the wave's second column group
        if (lane < NZ) scr[w * NZ + lane] = s3;
        float v = sumsq;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) scr[S::NW * NZ + w] = v;
    }
    // hidden-layer biases: sum over the 32 columns (the lanes of this k half), fixed order
#pragma unroll
    for (int q = 0; q < 16; q++) {
        float a = db1a[q], b = db2a[q];
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            a += __shfl_xor(a, off);
            b += __shfl_xor(b, off);
        }
        if (n == 0) {
            const int f = 32 * w + 8 * (q >> 2) + 4 * kh + (q & 3);
            out[go.b[0] + f] = a;
            out[go.b[1] + f] = b;
        }
    }
    FCS_BARRIER();
    if (tid < S::NO) {
        float a = 0.0f;
#pragma unroll
        for (int q = 0; q < S::NW; q++) a += scr[q * NZ + tid];
        out[go.b[2] + tid] = a;
    }
    if (tid == 0) {
        float a = 0.0f;
#pragma unroll
        for (int q = 0; q < S::NW; q++) a += scr[S::NW * NZ + q];
        out[go.n_params + 2] = a;
    }
}

// ------------------------------------------------------------------------------------------------
// host side (engine_fc.hip's launchers hand over when the split images are given)
// ------------------------------------------------------------------------------------------------
size_t fc_split_image_words(int Nz) { return Nz == 64 ? Fs<64>::SIMG : Fs<32>::SIMG; }
bool fc_split_supported(int cw) { return cw == 32 || cw == 16; }       // 32: this file; 16: the SPLIT instantiations of engine_fc.hip (same image size)

#define FCS_FOR_EACH_SHAPE(M, ...) M(64, __VA_ARGS__) M(32, __VA_ARGS__)
#define FCS_FOR_EACH_FWD(M) FCS_FOR_EACH_SHAPE(M, true, false, false) FCS_FOR_EACH_SHAPE(M, false, false, false) \
                            FCS_FOR_EACH_SHAPE(M, true, true, false) FCS_FOR_EACH_SHAPE(M, false, true, false)   \
                            FCS_FOR_EACH_SHAPE(M, true, true, true) FCS_FOR_EACH_SHAPE(M, false, true, true)
#define FCS_FOR_EACH_ADJ(M) FCS_FOR_EACH_SHAPE(M, false, false) FCS_FOR_EACH_SHAPE(M, true, false) FCS_FOR_EACH_SHAPE(M, true, true)

hipError_t fcs_set_kernel_attributes() {
    hipError_t e;
#define FCS_ATTR_F(N, T, C, K) if ((e = hipFuncSetAttribute((const void*)(fcs_forward_kernel<N, T, C, K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Fs<N>::LDS_FWD)) != hipSuccess) return e;
#define FCS_ATTR_A(N, C, K) if ((e = hipFuncSetAttribute((const void*)(fcs_adjoint_kernel<N, C, K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Fs<N>::LDS_ADJ)) != hipSuccess) return e;
    FCS_FOR_EACH_FWD(FCS_ATTR_F)
    FCS_FOR_EACH_ADJ(FCS_ATTR_A)
#undef FCS_ATTR_F
#undef FCS_ATTR_A
    return hipSuccess;
}

hipError_t fcs_launch_pack(const DevModel& m, const float* w, unsigned int* simgf, unsigned int* simgb, hipStream_t stream) {
    FcsOffsets o;
    for (int l = 0; l < 3; l++) { o.w[l] = m.w_off[l]; o.b[l] = m.b_off[l]; }
    if (m.Nz == 64) hipLaunchKernelGGL((fcs_pack_kernel<64>), dim3(512), dim3(256), 0, stream, o, w, simgf, simgb);
    else if (m.Nz == 32) hipLaunchKernelGGL((fcs_pack_kernel<32>), dim3(256), dim3(256), 0, stream, o, w, simgf, simgb);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t fcs_launch_forward(const DevModel& m, const unsigned int* simgf, const float* bias, const float* x0, size_t x0_stride, const float* bcs,
                              const float* save_times, int n_save, int iv_begin, int iv_end, int tape_iv0, int substeps, float* sol, float* dwtape,
                              unsigned int* masks, unsigned long long* swtape, int n_col, hipStream_t stream) {
    const dim3 grid((n_col + 31) / 32);
    const float CN = m.C_fc * (float)m.Nz, caKN = m.ca_K * (float)m.Nz;
    const bool tape = dwtape != nullptr, ca = m.model == COLNDE_MODEL_CONV_ADJ_NDE, rk = m.rkc != nullptr;
    bool launched = false;
#define FCS_FWD(N, T, C, K)                                                                                                                          \
    if (!launched && m.Nz == N && tape == T && ca == C && rk == K) {                                                                                 \
        hipLaunchKernelGGL((fcs_forward_kernel<N, T, C, K>), grid, dim3(Fs<N>::NT), Fs<N>::LDS_FWD, stream, simgf, bias, x0, x0_stride, bcs, save_times, n_save, \
                           iv_begin, iv_end, tape_iv0, substeps, CN, caKN, m.nst, m.rkc, sol, dwtape, masks, swtape, n_col);                         \
        launched = true;                                                                                                                             \
    }
    FCS_FOR_EACH_FWD(FCS_FWD)
#undef FCS_FWD
    return launched ? hipGetLastError() : hipErrorInvalidValue;
}

hipError_t fcs_launch_adjoint(const DevModel& m, const unsigned int* simgb, const float* save_times, int n_save, int iv_begin, int iv_end, int substeps,
                              const float* sol, const float* truth, float* dwtape, const unsigned int* masks, const unsigned long long* swtape,
                              float w_loss, float* lam_io, float* slab, int n_col, hipStream_t stream) {
    const dim3 grid((n_col + 31) / 32);
    const float CN = m.C_fc * (float)m.Nz, caKN = m.ca_K * (float)m.Nz;
    const bool ca = m.model == COLNDE_MODEL_CONV_ADJ_NDE, rk = m.rkc != nullptr;
    FcsGrad go;
    for (int l = 0; l < 3; l++) go.b[l] = m.b_off[l];
    go.n_params = m.n_params;
    bool launched = false;
#define FCS_ADJ(N, C, K)                                                                                                                             \
    if (!launched && m.Nz == N && ca == C && rk == K) {                                                                                              \
        hipLaunchKernelGGL((fcs_adjoint_kernel<N, C, K>), grid, dim3(Fs<N>::NT), Fs<N>::LDS_ADJ, stream, simgb, save_times, n_save, iv_begin, iv_end, substeps, \
                           CN, caKN, m.nst, m.rkc, sol, truth, dwtape, masks, swtape, w_loss, lam_io, slab, go, n_col);                              \
        launched = true;                                                                                                                             \
    }
    FCS_FOR_EACH_ADJ(FCS_ADJ)
#undef FCS_ADJ
    return launched ? hipGetLastError() : hipErrorInvalidValue;
}
