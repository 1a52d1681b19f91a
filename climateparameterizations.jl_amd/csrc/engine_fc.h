// engine_fc.h — host interface of the 32-column free-convection engine (engine_fc.hip).
#pragma once
#include "colnde_dev.h"

// FreeConvectionNDE (classical RK4) or ConvectiveAdjustmentNDE (RK4 or the stabilised RKC2 stepper) with Dense(Nz,4Nz,relu) ->
// Dense(4Nz,4Nz,relu) -> Dense(4Nz,Nz-1), Nz = 32 | 64
bool fc_supported(const DevModel& m, int stepper);
size_t fc_image_floats(int Nz);          // floats of one A-operand image (forward and backward images have the same size)
size_t fc_bias_floats(int Nz);
size_t fc_record_row_floats(int Nz);     // = dwtape_row_floats(m): the records are tile16's delta-tape records
static inline size_t fc_mask_words() { return 512; }     // relu-derivative bits per 32-column tile and stage: [layer 2][wave 4][lane 64] dwords
static inline size_t fc_switch_words(int cw) { return (size_t)cw; }   // ConvectiveAdjustmentNDE: the switch pattern of a stage, one 64-bit word per column
hipError_t fc_set_kernel_attributes();
// cw = columns per workgroup tile (32: v_mfma_f32_32x32x2_f32, the throughput shape; 16: v_mfma_f32_16x16x4_f32, half the matrix work per stage for
// problems that cannot fill 32-column tiles on every CU); the operand images depend on it
int fc_tile_width(int n_col);
// simgf / simgb (fc_split_image_words(Nz) words each, or null): the operand images of COLNDE_MATRIX_BF16X3_EXACT — every weight split exactly into three
// bf16 planes, in each wave's stream order — packed beside the f32 images (fc_split_supported: tile width 32 -> engine_fc_split.hip's kernels, 16 -> the
// SPLIT instantiations of engine_fc.hip's; the two orders differ, the size does not)
size_t fc_split_image_words(int Nz);
bool fc_split_supported(int cw);
hipError_t fc_launch_pack(const DevModel& m, int cw, const float* w, float* imgf, float* imgb, float* bias, unsigned int* simgf, unsigned int* simgb,
                          hipStream_t stream);
// Save intervals [iv_begin, iv_end) from x0 (column stride x0_stride floats).  dwtape == nullptr: plain forward solve.  Otherwise, from interval
// tape_iv0 on (iv_begin <= tape_iv0 < iv_end; records numbered from its first step), the stage inputs
// and hidden activations go into the records [tile32][step of this launch][stage][cw/16 records of 16 columns][R], the relu bits into masks [..][512] and
// (ConvectiveAdjustmentNDE) the switch pattern into swtape [..][cw].  Stages per step: m.nst (4, or the RKC2 stage count when m.rkc is set).
// simgf != null (and cw == 32): the split kernels (v_mfma_f32_32x32x16_bf16 on exact three-way operand splits) instead of the f32-MFMA ones
hipError_t fc_launch_forward(const DevModel& m, int cw, const float* imgf, const unsigned int* simgf, const float* bias, const float* x0, size_t x0_stride,
                             const float* bcs, const float* save_times, int n_save, int iv_begin, int iv_end, int tape_iv0, int substeps, float* sol,
                             float* dwtape, unsigned int* masks, unsigned long long* swtape, int n_col, hipStream_t stream);
// slab: one row of n_params + 8 floats per tile (bias gradients and the squared-error sum; the weight gradients are the dW GEMM's).
// lam_io [columns padded to 32][Nz]: carries λ between the time segments of a segmented gradient pass (null when one launch covers the axis).
hipError_t fc_launch_adjoint(const DevModel& m, int cw, const float* imgb, const unsigned int* simgb, const float* save_times, int n_save, int iv_begin, int iv_end,
                             int substeps,
                             const float* sol, const float* truth, float* dwtape, const unsigned int* masks, const unsigned long long* swtape,
                             float w_loss, float* lam_io, float* slab, int n_col, hipStream_t stream);
// compute_neural_network_forcing! (double_gyre_nn.jl:149-168): T [n_col][Nz] model units, top_flux [n_col], out = -dz(wT) on cell centres
hipError_t fc_launch_infer(const DevModel& m, int cw, const float* imgf, const float* bias, const float* T, const float* top_flux, float inv_dz,
                           float* out, int n_col, hipStream_t stream);

// engine_fc_split.hip: the same kernels on v_mfma_f32_32x32x16_bf16 from exact three-way bf16 operand splits (COLNDE_MATRIX_BF16X3_EXACT; 32-column tiles):
// one wave per 32-row tile of a hidden layer, activations as bf16 planes in LDS, split once in the producing wave's epilogue.  Reached through the
// fc_launch_* calls above when the split images are handed over; tapes, slab rows and argument meaning are engine_fc.hip's.
hipError_t fcs_set_kernel_attributes();
hipError_t fcs_launch_pack(const DevModel& m, const float* w, unsigned int* simgf, unsigned int* simgb, hipStream_t stream);
hipError_t fcs_launch_forward(const DevModel& m, const unsigned int* simgf, const float* bias, const float* x0, size_t x0_stride, const float* bcs,
                              const float* save_times, int n_save, int iv_begin, int iv_end, int tape_iv0, int substeps, float* sol, float* dwtape,
                              unsigned int* masks, unsigned long long* swtape, int n_col, hipStream_t stream);
hipError_t fcs_launch_adjoint(const DevModel& m, const unsigned int* simgb, const float* save_times, int n_save, int iv_begin, int iv_end, int substeps,
                              const float* sol, const float* truth, float* dwtape, const unsigned int* masks, const unsigned long long* swtape,
                              float w_loss, float* lam_io, float* slab, int n_col, hipStream_t stream);
