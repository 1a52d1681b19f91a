// column_ops.h — the steps either side of the NDE hot path: the batched implicit convective-adjustment
// step of the ocean embeddings and the on-device ADAM update.
#pragma once
#include <hip/hip_runtime.h>

// T, out: [n_col][Nz] float32 (in place allowed), halo_bottom / halo_top: [n_col] or null.  2 <= Nz <= 128.
hipError_t launch_convective_adjustment(const float* T, const float* halo_bottom, const float* halo_top, float dt_over_dz2, float K,
                                        float* out, int Nz, int n_col, hipStream_t stream);
// modified_pacanowski_philander! (wind_mixing/src/NDE_oceananigans.jl:61-101): u, v, T and the outputs [n_col][Nz] (in place allowed field by
// field), halo_bottom [3][n_col] (u, v, T halo cells below k = 0) or null, params = {nu0, nu_minus, dRi, Ric, Pr, alpha, g}.  2 <= Nz <= 128.
hipError_t launch_mpp_diffusion(const float* u, const float* v, const float* T, const float* halo_bottom, float dt, float dz,
                                const float params[7], int convective_adjustment, float* uo, float* vo, float* To, int Nz, int n_col,
                                hipStream_t stream);
hipError_t launch_adam_step(float* w, const float* grad, float* m, float* v, float eta, float beta1, float beta2, float eps,
                            float beta1_t, float beta2_t, int n, hipStream_t stream);
// data preparation: rows are profiles ([n_rows][N] -> [n_rows][n]); face = 0: block means, 1: linear interpolation keeping the end points
hipError_t launch_coarse_grain(const float* in, int n_rows, int N, int n, int face, float* out, hipStream_t stream);
hipError_t launch_zscore_stats(const float* x, long count, float* out2 /* mu, sigma */, hipStream_t stream);
hipError_t launch_zscore_scale(const float* x, long count, const float* mu_sigma, float* out, hipStream_t stream);
// loss_per_tstep (wind_mixing/src/loss.jl:44-46) for the six profile terms of every column: sol, truth [n_col][n_save][n_var Nz] (n_var = 3 | 1),
// out [n_col][6][n_save] = mse over the Nz levels (terms u, v, T) and over the Nz + 1 faces of the finite-difference gradient, its two zero
// boundary rows included (terms dudz, dvdz, dTdz: loss.jl:9, NDE_training.jl:308-317); T-only models fill terms 2 and 5, the others are 0
hipError_t launch_loss_per_tstep(const float* sol, const float* truth, int n_col, int n_save, int Nz, int n_var, float* out, hipStream_t stream);
// out[0] = max over the n_rows rows of rms_i((a_i - b_i) / (floor + |b_i|)) over the `row` floats of a row (+inf if any entry is not finite);
// partial: scratch of >= 1024 floats
hipError_t launch_rel_diff_max(const float* a, const float* b, long n_rows, int row, float floor, float* partial, float* out, hipStream_t stream);
