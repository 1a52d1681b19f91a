// column_ops.h — the steps either side of the NDE hot path: the batched implicit convective-adjustment
// step of the ocean embeddings and the on-device ADAM update.
#pragma once
#include <hip/hip_runtime.h>

// T, out: [n_col][Nz] float32 (in place allowed), halo_bottom / halo_top: [n_col] or null.  2 <= Nz <= 128.
hipError_t launch_convective_adjustment(const float* T, const float* halo_bottom, const float* halo_top, float dt_over_dz2, float K,
                                        float* out, int Nz, int n_col, hipStream_t stream);
hipError_t launch_adam_step(float* w, const float* grad, float* m, float* v, float eta, float beta1, float beta2, float eps,
                            float beta1_t, float beta2_t, int n, hipStream_t stream);
