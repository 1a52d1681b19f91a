// engine_regtile.h — register-resident tile engine for the static wind-mixing shape
// (Nz = 32, three Chain(Dense(96,50,σ), Dense(50,20,σ), Dense(20,31)) nets: wind_mixing/train_NDE.jl:103).
#pragma once
#include "colnde_dev.h"
#include "engine_tile16.h"

#define RT_COLS 32                 // columns per wavefront tile (N of v_mfma_f32_32x32x2_f32)
#define RT_WAVES 4                 // wavefronts per workgroup (one per SIMD, up to 512 VGPRs each)

// compact weight image in LDS (floats): plain row-major matrices with odd row strides, so that both W (lane = output
// row) and W^T (lane = input column) A-operand reads are bank-conflict-free with a per-lane base + immediate offset
#define RT_LD1 97
#define RT_LD2 51
#define RT_LD3 21
#define RT_W1C 0                                   // [150][97]   rows n*50+f, cols = input feature (col 96 zero)
#define RT_W2C (150 * RT_LD1 + 2)                  // [60][51]    rows n*20+f, cols = a1 feature (col 50 zero)
#define RT_W3C (RT_W2C + 60 * RT_LD2 + 1)          // [93][21]    rows n*31+f, cols = a2 feature (col 20 zero)
#define RT_B1C (RT_W3C + 93 * RT_LD3 + 3)          // [160]       n*50+f, tail zero
#define RT_B2C (RT_B1C + 160)                      // [64]        n*20+f, tail zero
#define RT_B3C (RT_B2C + 64)                       // [96]        n*32+face (face 0 -> 0)
#define RT_IMG_FLOATS (RT_B3C + 96)
// behind it in the same allocation: the forward nets as bf16 A operands of v_mfma_f32_16x16x32_bf16, three planes (exact split) per group,
// [48 groups][3 planes][64 lanes][8 bf16] (COLNDE_MATRIX_BF16X3_EXACT; rt16_forward_kernel<ACT, true>)
#define RT_SIMG_OFF ((RT_IMG_FLOATS + 3) & ~3)
#define RT_SIMG_GROUPS 48
#define RT_SIMG_WORDS (RT_SIMG_GROUPS * 3 * 64 * 4)
// ... and behind that, the same for the net-split forward kernel (rt16sh_forward_kernel<ACT, RICH, false, true>): layers 1 and 2 in the per-net tile
// order of that kernel — [27 full layer-1 groups (net, tile 0..2, k-block)][12 layer-2 groups][the 8 live lanes of each net's quarter-filled
// fourth layer-1 tile: (net, k-block, plane, 8 lanes)][one zero operand]; layer 3 stays on the fp32 image
#define RT_SIMG2_OFF (RT_SIMG_OFF + RT_SIMG_WORDS)
#define RT_SIMG2_L2 (27 * 768)                       // word offsets inside the image
#define RT_SIMG2_T3 (RT_SIMG2_L2 + 12 * 768)
#define RT_SIMG2_ZERO (RT_SIMG2_T3 + 27 * 8 * 4)
#define RT_SIMG2_WORDS (RT_SIMG2_ZERO + 4)
// ... and for the regtile adjoint's W1^T products (rt_adjoint_kernel<ACT, true, true>, COLNDE_MATRIX_BF16X3_EXACT): group G = 9 n + 3 c + q (net, 16-deep k-block of
// the net's delta registers 8 c .. 8 c + 7, state tile), lane (state row m, kh), element i = W1[50 n + 2 (8 c + i) + kh][32 q + m]:
// [27 groups][planes h, m][64 lanes][8 bf16] and the fp32 rows of features 48, 49 ([net][kh][96]) go to LDS in place of the fp32 W1,
// [27 groups][plane l][64 lanes][8 bf16] stay in global memory (L2) and are fetched into registers
#define RT_ASIMG_OFF (RT_SIMG2_OFF + RT_SIMG2_WORDS)
#define RT_ASIMG_HM_WORDS (27 * 2 * 256)
#define RT_ASIMG_LEFT (RT_ASIMG_HM_WORDS)                 // 576 floats
#define RT_ASIMG_L (RT_ASIMG_LEFT + 576)                  // 27 * 256 words
#define RT_ASIMG_WORDS (RT_ASIMG_L + 27 * 256)
// ... and for the net-split adjoint's W1_n^T products (rt16sh_adjoint_kernel<ACT, RICH, false, true>, COLNDE_MATRIX_BF16X3_EXACT): per net n, 32-deep k-block kb
// (element e of lane (i, kq) <-> quad Q = 8 kb + e of the net's 13, hidden feature 4 Q + kq) and 16-row state tile (x index 16 tile + i):
// [net][kb][tile][planes h, m][64 lanes][8 bf16] go to LDS in the fp32 W1's place, [net][kb][tile][plane l][64 lanes][8 bf16] stay in global memory (L2)
#define RT_NSA_OFF (RT_ASIMG_OFF + RT_ASIMG_WORDS)
#define RT_NSA_HM_WORDS (3 * 2 * 6 * 2 * 256)
#define RT_NSA_L (RT_NSA_HM_WORDS)
#define RT_NSA_WORDS (RT_NSA_L + 3 * 2 * 6 * 256)
#define RT_IMG_ALLOC (RT_NSA_OFF + RT_NSA_WORDS)

bool rt_supported(const DevModel& m);
size_t rt_forward_lds_bytes();
hipError_t rt_set_attributes();
hipError_t rt_launch_pack(const DevModel& m, const float* w, float* wimg, hipStream_t stream);
hipError_t rt_launch_forward(const DevModel& m, const float* wimg, const float* x0, const float* bcs,
                             const float* save_times, int n_save, int substeps, float* sol, float* tape, float* tapez,
                             int n_col, bool fwd32, bool split, hipStream_t stream);
hipError_t rt_launch_forward_split(const DevModel& m, const float* wimg, const float* x0, const float* bcs, const float* save_times,
                                   int n_save, int substeps, float* sol, float* t16_tape, float* t16_ztape, int n_col, bool rich, bool use_helper, bool want_split, hipStream_t stream);
size_t rt_split_rich_record_floats();   // floats per (tile, step, stage) of the net-split kernels' rich tape (which then takes the place of t16_ztape)
hipError_t rt_launch_adjoint_split(const DevModel& m, const float* wimg, const float* save_times, int n_save, int substeps, const float* sol,
                                   const float* truth, const float* t16_tape, const float* t16_ztape, const LossWeights& lw, float* slab,
                                   int n_col, float* dwtape, bool rich, bool use_helper, bool want_split, hipStream_t stream);
bool rt_adjoint_split_has_bf16(const DevModel& m, bool use_helper);   // the net-split adjoint has a split (bf16-pipe) kernel for this configuration
bool rt_forward_is32();   // COLNDE_RT_FWD=32 in the environment (read when a handle is created)
size_t rt_adjoint_lds_bytes();
size_t rt_tape_floats(int n_col, int n_steps);
size_t rt_tape2_floats(int n_col, int n_steps);
size_t rt_tapez_floats(int n_col, int n_steps);
int rt_n_wtiles(int n_col);
int rt_dw1_waves(int n_col, int n_steps);
hipError_t rt_launch_adjoint(const DevModel& m, const float* wimg, const float* bcs, const float* save_times, int n_save,
                             int substeps, const float* sol, const float* truth, const float* tape, float* tape2,
                             const float* tapez, const LossWeights& lw, float* slab, int n_col, bool want_split, hipStream_t stream);
hipError_t rt_launch_dw1(const DevModel& m, const float* tape, const float* tape2, int n_col, int n_steps, float* slab_rows,
                         bool split, hipStream_t stream);
hipError_t rt_debug_read_stamps(unsigned long long* out8);
