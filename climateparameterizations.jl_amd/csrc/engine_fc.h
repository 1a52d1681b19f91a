// engine_fc.h — host interface of the 32-column free-convection engine (engine_fc.hip).
#pragma once
#include "colnde_dev.h"

// FreeConvectionNDE with Dense(Nz,4Nz,relu) -> Dense(4Nz,4Nz,relu) -> Dense(4Nz,Nz-1), Nz = 32 | 64, classical RK4
bool fc_supported(const DevModel& m, int stepper);
size_t fc_image_floats(int Nz);          // floats of one A-operand image (forward and backward images have the same size)
size_t fc_bias_floats(int Nz);
size_t fc_record_row_floats(int Nz);     // = dwtape_row_floats(m): the records are tile16's delta-tape records
static inline size_t fc_mask_words() { return 512; }     // relu-derivative bits per 32-column tile and stage: [layer 2][wave 4][lane 64] dwords
hipError_t fc_set_kernel_attributes();
hipError_t fc_launch_pack(const DevModel& m, const float* w, float* imgf, float* imgb, float* bias, hipStream_t stream);
// dwtape == nullptr: plain forward solve.  Otherwise the stage inputs and hidden activations go into the records
// [tile32][step][stage][2 x 16 columns][R] and the relu bits into masks [tile32][step][stage][512].
hipError_t fc_launch_forward(const DevModel& m, const float* imgf, const float* bias, const float* x0, const float* bcs, const float* save_times,
                             int n_save, int substeps, float* sol, float* dwtape, unsigned int* masks, int n_col, hipStream_t stream);
// slab: one row of n_params + 8 floats per 32-column tile (bias gradients and the squared-error sum; the weight gradients are the dW GEMM's)
hipError_t fc_launch_adjoint(const DevModel& m, const float* imgb, const float* save_times, int n_save, int substeps, const float* sol,
                             const float* truth, float* dwtape, const unsigned int* masks, float w_loss, float* slab, int n_col,
                             hipStream_t stream);
