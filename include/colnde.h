/* colnde.h — C ABI of the MI355X-native NDE column-model hot path.
 *
 * The reference (CliMA/ClimateParameterizations.jl, pure Julia) has no FFI for this path: it is reached
 * through Julia closures handed to ODEProblem / OptimizationFunction / Flux.train!.  Each entry
 * point below names the reference closure it replaces (paths relative to /root/reference); INTEGRATION.md
 * shows the `ccall` stubs that give them the reference's names and arities.
 *
 * Conventions: every call returns 0 on success, non-zero on error with colnde_last_error() holding a
 * thread-local message; no exceptions cross the boundary.  One handle = one GPU = one host thread at a
 * time.  Host pointers are borrowed for the duration of the call; device memory is owned by the handle.
 * All arrays are float32, C order.  `weights` is in Flux.destructure order — per net, per Dense layer,
 * column-major vec(W[out x in]) then b — nets concatenated uw; vw; wT
 * (wind_mixing/src/NDE_training.jl:11-21,37).  There is NO CPU fallback: without a visible gfx950 device
 * colnde_create fails.
 */
#ifndef COLNDE_H
#define COLNDE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COLNDE_VERSION 106
#define COLNDE_MAX_LAYERS 8

enum { COLNDE_MODEL_WIND_MIXING = 0,        /* NDE / NDE!: wind_mixing/src/NDE_training.jl:56-165 */
       COLNDE_MODEL_FREE_CONVECTION = 1,    /* FreeConvectionNDE: free_convection/src/free_convection_nde.jl:29-38 */
       COLNDE_MODEL_CONV_ADJ_NDE = 2 };     /* ConvectiveAdjustmentNDE: free_convection/src/convective_adjustment_nde.jl:33-48 */

enum { COLNDE_ACT_IDENTITY = 0, COLNDE_ACT_RELU = 1, COLNDE_ACT_MISH = 2, COLNDE_ACT_SWISH = 3,
       COLNDE_ACT_TANH = 4, COLNDE_ACT_LEAKYRELU = 5 };

enum { COLNDE_ENGINE_AUTO = 0,      /* regtile when the configuration is one it is built for and there are > 8,192 columns, fc32 for its shape, else tile16 (with the net-split kernels where the shape is regtile's) */
       COLNDE_ENGINE_GENERIC = 1,   /* tile16: 16-column MFMA tiles staged through LDS, any layer sizes / model.  Networks whose per-tile activation rows exceed
                                       the CU's 160 KB of LDS — the reference's WIDE wind-mixing architectures, 3 x Chain(Dense(96,400,σ), Dense(400,400,σ),
                                       Dense(400,31)) of wind_mixing/train_NDE.jl:101-102 and train_NDE_args.jl:150-166 (their rows alone are 160 KB) — run
                                       with that one array in a per-workgroup slab of global memory (L2-resident) and the taped weight-gradient path
                                       (colnde_describe: "activation_rows=global_memory"); 3 x (96-400-31) fits the LDS.  Which pipe: for the networks with rows in
                                       global memory the dense chains of the forward and the taped adjoint AND the tape GEMM (dW) follow matrix_arithmetic
                                       (BF16X3_EXACT: pre-split weight planes, v_mfma_f32_16x16x32_bf16; colnde_plan info[7] bits 1-3); every other
                                       tile16 shape runs f32-MFMA forward / adjoint kernels under either arithmetic and only its tape GEMM follows it.
                                       4,096 columns x 32 steps of 3 x (96-400-400-31 swish), fwd + adjoint: 41 ms = 48 algorithmic TFLOP/s
                                       (52 ms under F32_MFMA; profiles/r05_wide_networks.log — bound by the weight stream from L2, not by the pipe). */
       COLNDE_ENGINE_MFMA = 2,      /* regtile: 32 columns per wavefront resident in registers (static 96-50-20-31 wind-mixing
                                       shape; colnde_create fails if the configuration is not covered) */
       COLNDE_ENGINE_FC32 = 3 };    /* fc32: 32-column v_mfma_f32_32x32x2_f32 tiles (16-column v_mfma_f32_16x16x4_f32 tiles up to 4,096 columns)
                                       with compile-time shapes for FreeConvectionNDE (RK4) and ConvectiveAdjustmentNDE (RK4, RKC2) with the
                                       reference's network Dense(Nz,4Nz,relu), Dense(4Nz,4Nz,relu), Dense(4Nz,Nz-1)
                                       (free_convection/train_free_convection_nde.jl:119-121), Nz = 32 | 64; AUTO picks it for that
                                       shape; colnde_create fails if it is requested for anything else */

enum { COLNDE_STEPPER_RK4 = 0,      /* classical RK4, `substeps` per save interval (what the bench measures) */
       COLNDE_STEPPER_RKC2 = 1 };   /* stabilised second-order Runge-Kutta-Chebyshev for the stiff variants: `substeps` steps per
                                       save interval of `rkc_stages` stages each (tile16; the net-split latency kernels; fc32 for
                                       ConvectiveAdjustmentNDE).  Stands where the reference uses ROCK4 (wind_mixing/train_NDE.jl:143,
                                       free_convection/test_free_convection_nde.jl:32-35).
                                       GRADIENT CAVEAT: for the models with a convective-adjustment switch (min(0, K dT/dz):
                                       ConvectiveAdjustmentNDE, the wind-mixing `convective_adjustment` branch) the RKC2 gradient is
                                       NOT the exact discrete adjoint — that one is unbounded, the stage Jacobians of one step
                                       differing in their switch patterns — but the pullback with ONE switch pattern per step (that
                                       of the last stage input).  It converges to the sub-stepped RK4 gradient as the step shrinks
                                       (6.5 % / cosine 0.998 at 8 steps per save interval of the 64-level model, 2 % at 16)
                                       and is exact for smooth closures (the Richardson-number branch).  colnde_plan reports it
                                       (info[7] bit 0). */

/* How the Float32 `Dense` products of the reference (W*x, NDE_training.jl:94-96; their transposes and outer products in the gradient) are
 * evaluated on the matrix pipe.  Both are f32 arithmetic: f32 operands, f32 accumulation, no operand rounded.
 *   BF16X3_EXACT (default, = 0): every f32 operand is split EXACTLY into three bf16 parts (x = x_h + x_m + x_l by truncation; 3 x 8 = 24
 *     significant bits) and a product is the six part-products down to 2^-16 on v_mfma_f32_*_bf16 with f32 accumulation; the three dropped
 *     part-products are below 2^-23 |a b| together — one f32 rounding of the product, which an f32 FMA chain commits at every step anyway
 *     (measured: profiles/r03ze_split_error_probe.txt).  Inf/NaN operands give NaN (Inf - Inf in the split), which the solve calls report
 *     as a non-finite loss.  Subnormals are not flushed (bf16 inputs, f32 results: measured, profiles/r04_split_edge_probe.txt), but the part of an
 *     operand below 2^-133, the smallest bf16 subnormal, is dropped: operands are exact for |x| >= 2^-110 and carry an ABSOLUTE error below 2^-133
 *     (9e-41) under that — F32_MFMA keeps full relative precision down to FLT_MIN.
 *     Used where a split kernel exists (colnde_plan info[7] says which kernels ran on it); the other kernels run F32_MFMA.
 *   F32_MFMA (= 1): v_mfma_f32_32x32x2_f32 / 16x16x4_f32 throughout (bitwise an fmaf chain; 1/16 of the bf16 pipe's rate).
 * Test overrides (read when the arithmetic is resolved: colnde_create, colnde_set_matrix_arithmetic): COLNDE_FWD_SPLIT / COLNDE_ADJ_SPLIT /
 * COLNDE_DW_SPLIT = 0 | 1 force the forward-solve / adjoint / weight-gradient-GEMM kernels individually. */
enum { COLNDE_MATRIX_BF16X3_EXACT = 0, COLNDE_MATRIX_F32_MFMA = 1 };

/* Mirrors the `constants`, `scalings`, `conditions` NamedTuples of prepare_parameters_NDE_training
 * (wind_mixing/src/NDE_training.jl:1-44, :205-207) and the parameter tail of the free-convection NDEs
 * (free_convection/src/free_convection_nde.jl:49-62). */
typedef struct colnde_config {
    int32_t model;                               /* COLNDE_MODEL_* */
    int32_t Nz;                                  /* cells per column; state = 3*Nz (wind mixing) or Nz */
    int32_t n_layers;                            /* Dense layers per net */
    int32_t layer_sizes[COLNDE_MAX_LAYERS + 1];  /* in, h1, ..., out (= Nz-1 interior faces) */
    int32_t activations[COLNDE_MAX_LAYERS];      /* COLNDE_ACT_* per layer */
    int32_t modified_pacanowski_philander;       /* conditions.* */
    int32_t convective_adjustment;
    int32_t zero_weights;
    int32_t smooth_NN;
    int32_t smooth_Ri;
    int32_t diurnal;                             /* bcs[5] then holds Q^b; wT_top(t) as NDE_training.jl:73 */
    int32_t train_gradient;
    int32_t inplace_variant;                     /* NDE! arithmetic of training_postprocessing.jl:105-153 */
    float H, tau, f, g, alpha, nu0, nu_minus, Ric, dRi, Pr, kappa, eps;
    float mu[6];                                 /* scalings u, v, T, uw, vw, wT */
    float sigma[6];
    float ca_K;                                  /* convective_adjustment_nde.jl:43 (10) */
    int32_t n_save;                              /* number of `saveat` times, >= 2 */
    int32_t substeps;                            /* steps per save interval (RK4; RKC2: steps of rkc_stages stages); 0 = chosen from `reltol` by the first solve
                                                    call of the handle (colnde_choose_substeps), then kept */
    const float* save_times;                     /* [n_save] nondimensional t_train ./ tau (borrowed during create) */
    int32_t n_columns;                           /* columns (simulations) held by this handle */
    int32_t device;                              /* HIP device ordinal */
    int32_t engine;                              /* COLNDE_ENGINE_* */
    int32_t stepper;                             /* COLNDE_STEPPER_* (0 = RK4) */
    int32_t rkc_stages;                          /* RKC2: stages per step, 2..256; 0 = automatic (colnde_rkc_stages) */
    int32_t matrix_arithmetic;                   /* COLNDE_MATRIX_* (0 = exact three-way bf16 split where a split kernel exists) */
    float reltol;                                /* the tolerance the reference hands its adaptive integrator (solve(...; reltol=1f-3): NDE_training.jl:291;
                                                    1e-4: free_convection/src/solve.jl:4); 0 = 1e-3.  Used when substeps = 0 and by colnde_choose_substeps:
                                                    see colnde_error_estimate for the norm */
} colnde_config;

typedef struct colnde_handle colnde_handle;

const char* colnde_last_error(void);
int colnde_version(void);

/* Least `substeps` for which the classical-RK4 step stays inside its stability region (|lambda dt| <= 2.785) for the stiffest
 * diffusive mode the configuration can switch on: lambda = -4 D Nz^2 with D = tau (nu0 + nu_minus) max(1, 1/Pr) / H^2
 * (Richardson closure), tau kappa / H^2 (convective-adjustment branches) or (sigma_wT/sigma_T)(tau/H) ca_K
 * (ConvectiveAdjustmentNDE).  The reference sidesteps this with ROCK4 (wind_mixing/train_NDE.jl:143).  colnde_forward / _loss /
 * _loss_grad refuse a configuration below it (instead of returning a blown-up solve with rc = 0) unless
 * COLNDE_ALLOW_UNSTABLE_DT=1; colnde_rhs is not affected.  Returns -1 on an invalid configuration.  No GPU needed. */
int  colnde_min_substeps(const colnde_config* cfg);

/* Stage count an RKC2 configuration runs with: cfg->rkc_stages when given, else the least s >= 2 whose real stability interval
 * beta(s) ~ 0.653 s^2 (damping 2/13), used to 90 %, covers lambda dt of the same stiffest mode.  A given stage count that does
 * not cover it is refused by the solve calls like an unstable RK4 step.  Returns -1 on an invalid configuration. */
int  colnde_rkc_stages(const colnde_config* cfg);

int  colnde_create(const colnde_config* cfg, colnde_handle** out);
void colnde_destroy(colnde_handle* h);
int  colnde_n_params(const colnde_handle* h);
int  colnde_engine(const colnde_handle* h);                  /* engine actually selected */
int  colnde_set_stream(colnde_handle* h, void* hip_stream);  /* default: the null stream */
/* Switch the matrix arithmetic of an existing handle (COLNDE_MATRIX_*): tapes, plans and results layout do not depend on it, so the same
 * handle can run both for an A/B on identical inputs.  colnde_matrix_arithmetic returns the configured value. */
int  colnde_set_matrix_arithmetic(colnde_handle* h, int matrix_arithmetic);
int  colnde_matrix_arithmetic(const colnde_handle* h);
/* Global column count when columns are sharded over ranks: losses and gradients are normalised by it so
 * that a SUM all-reduce of the per-rank results is the global mean (NDE_training.jl:312-317). */
int  colnde_set_global_columns(colnde_handle* h, int64_t n_columns_total);

/* x0 [n_col][n_state], bcs [n_col][n_bc] (wind mixing: uw_b,uw_t,vw_b,vw_t,wT_b,wT_t — the tail of `p`,
 * NDE_training.jl:60; free convection: bottom, top — free_convection_nde.jl:31), truth
 * [n_col][n_save][n_state] or NULL.  Replaces uvT₀s / BCs / uvT_trains (NDE_training.jl:220-243). */
int colnde_set_problem(colnde_handle* h, const float* x0, const float* bcs, const float* truth);

/* One RHS evaluation for n_columns columns — NDE(x,p,t) (NDE_training.jl:56-81), NDE!(dx,x,p,t)
 * (training_postprocessing.jl:131-153), ∂T∂t(T,p,t) (free_convection_nde.jl:29-38). */
int colnde_rhs(colnde_handle* h, const float* x, const float* weights, const float* bcs, float t,
               float* dx, int n_columns);

/* solve(prob, alg; p=[weights;BCs], saveat=t_train) — NDE_training.jl:291,403; training_postprocessing.jl:157;
 * free_convection/src/solve.jl:1-6.  sol [n_col][n_save][n_state] (NULL: keep on device only). */
int colnde_forward(colnde_handle* h, const float* weights, float* sol);

/* ---- what `reltol` means here.  The reference integrates with an ADAPTIVE stepper (solve(prob, ROCK4(); reltol=1f-3, saveat=...): wind_mixing/src/
 * NDE_training.jl:291,304,403; reltol=1e-4: free_convection/src/solve.jl:4); this path steps at a fixed `substeps` per save interval, and the tolerance is
 * enforced a posteriori.  colnde_error_estimate: one more forward solve at 2 x substeps and Richardson's estimate of the error of the solve at `substeps`,
 *   e = (u_S - u_2S) 2^p / (2^p - 1),   *max_rel_err = max over columns and save points of rms_i(e_i / (abstol / reltol + |u_i|))
 * (rms over the state's components) — the integrator's accept test rms(err / (abstol + reltol |u|)) <= 1 with OrdinaryDiffEq's default abstol = 1e-6,
 * divided through by reltol, applied to the whole save interval instead of one adaptive step; +inf when a solve is not finite.  The floor
 * abstol / reltol follows the tolerance in force (cfg.reltol for colnde_error_estimate, the argument for colnde_choose_substeps): 1e-3 at wind mixing's
 * reltol = 1e-3, 1e-2 at free convection's 1e-4.  p is the order the RIGHT-HAND SIDE lets the stepper reach: 4 (RK4) and 2 (RKC2) for smooth closures,
 * 1 (factor 2) for the switching ones — ConvectiveAdjustmentNDE's min(0, K dT/dz) and the wind-mixing convective-adjustment branch converge at about
 * first order through their kinks whatever the stepper (measured: profiles/r05_rkc2_conditioning.json).  colnde_choose_substeps: the least
 * power-of-two sub-step count, not below colnde_min_substeps, whose estimate is <= reltol (reltol <= 0: cfg.reltol); the handle keeps it (call it before
 * the first colnde_loss_grad: the sub-step count sizes the tapes).  A handle created with substeps = 0 does this by itself in its first solve call, with
 * the weights of that call; later calls reuse the count (a training loop re-checks with colnde_error_estimate when it wants to).  colnde_substeps: the
 * count in use.  The stability bound alone (colnde_min_substeps) knows only the closure's diffusion; a trained net's Jacobian shows up here.
 * float32 solves resolve about 1e-4 in this norm (it divides by 1e-3 + |u|; round-off grows with the step count): a tolerance below that floor is
 * refused by colnde_choose_substeps with the floor it found, and the reference's 1e-3 / 1e-4 sit at or above it. */
int colnde_error_estimate(colnde_handle* h, const float* weights, float* max_rel_err);
int colnde_error_estimate_dev(colnde_handle* h, const float* d_weights, float* max_rel_err /* host */);
int colnde_choose_substeps(colnde_handle* h, const float* weights, float reltol, int* substeps /* nullable */, float* estimate /* nullable */);
int colnde_substeps(const colnde_handle* h);
/* Impose a sub-step count.  The column-sharded recipe (one handle per GPU, colnde_set_global_columns): substeps = 0 would let every rank choose from its own
 * columns — the SUM-all-reduced gradient would mix discretisations and the tapes would differ per rank — so a handle that knows it holds a shard REFUSES the
 * automatic choice; instead every rank calls colnde_choose_substeps, the host takes the MAX over ranks (one MAX all-reduce of one integer) and every rank
 * calls colnde_set_substeps with it.  Refused once the gradient path has sized its tapes, and outside the stability bound (colnde_min_substeps).  With the
 * RKC2 stepper and rkc_stages = 0 the stage count follows the new step (the least s with 0.9 beta(s) >= lambda dt), as it does inside
 * colnde_choose_substeps / colnde_error_estimate.  Float32 and the stage count: at a FIXED step, 17 ... 136 stages stand equally far from float64 (the
 * increment-form recurrence is internally stable); what separates float32 from float64 on ConvectiveAdjustmentNDE is the switch, and it shrinks with the
 * STEP, not the stage count — see the table in DESIGN section 2 (tools/rkc_conditioning.py). */
int colnde_set_substeps(colnde_handle* h, int substeps);

/* predict_flux(uvT, BCs, ...) (wind_mixing/src/NDE_training.jl:83-147; exported at wind_mixing/src/WindMixing.jl:6): the face fluxes whose divergence the
 * RHS takes — uw, vw, wT on the Nz + 1 faces, NN output minus the closure's diffusive flux (MPP) or convective-adjustment flux, boundary faces as the
 * conditions say — for n_columns states: flux [n_col][3][Nz + 1], scaled units.  T-only models: flux [n_col][1][Nz + 1] = [bottom; NN(T); top]
 * (free_convection_nde.jl:33) minus min(0, K dT/dz) for ConvectiveAdjustmentNDE — the wT that the dataset-level solve_nde re-evaluates per saved step
 * (free_convection/src/solve.jl:32-46).  Same arguments as colnde_rhs. */
int colnde_flux(colnde_handle* h, const float* x, const float* weights, const float* bcs, float t, float* flux, int n_columns);
int colnde_flux_dev(colnde_handle* h, const float* d_x, const float* d_weights, const float* d_bcs, float t, float* d_flux, int n_columns);

/* loss_per_tstep(a, b) (wind_mixing/src/loss.jl:44-46) on the six profile matrices of every simulation, as NDE_profile calls it
 * (training_postprocessing.jl:311-316): out [n_col][6][n_save], term order u, v, T, dudz, dvdz, dTdz; entry = mse over the Nz levels (gradient terms:
 * over the Nz + 1 faces, the two zero boundary rows included) of solve(weights) against the truth at that save point, UNSCALED by the loss scalings.
 * T-only models fill terms 2 and 5. */
int colnde_loss_per_tstep(colnde_handle* h, const float* weights, float* out);
int colnde_loss_per_tstep_dev(colnde_handle* h, const float* d_weights, float* d_out);

/* loss_NDE / loss_gradient_NDE value (NDE_training.jl:290-323), nde_loss (training.jl:55-62).
 * scalings[6] = loss_scalings (u,v,T,dudz,dvdz,dTdz); terms[6] = scaled losses; total = their sum. */
int colnde_loss(colnde_handle* h, const float* weights, const float scalings[6], float terms[6], float* total);

/* Value and d(total)/d(weights): replaces Zygote through the sensitivity solve
 * (OptimizationFunction(loss_gradient_NDE, AutoZygote()), NDE_training.jl:327-333; Flux.train!, training.jl:71). */
int colnde_loss_grad(colnde_handle* h, const float* weights, const float scalings[6], float terms[6],
                     float* total, float* grad);

/* compute_neural_network_forcing! (free_convection/double_gyre_nn.jl:149-168): T [n_col][Nz] model units,
 * top_flux [n_col]; out [n_col][Nz] = -dz(wT) on cell centres with dz = Lz/Nz. */
int colnde_infer_forcing(colnde_handle* h, const float* weights, const float* T, const float* top_flux,
                         float Lz, float* out, int n_columns);
/* The same evaluation with the reference's storage convention: compute_neural_network_forcing! fills params.∂z_wT_NN with +dz(wT)
 * (double_gyre_nn.jl:165) and the forcing function negates it (:135).  out [n_col][Nz] = +dz(wT); colnde_infer_forcing returns the forcing itself. */
int colnde_infer_dz_wT(colnde_handle* h, const float* weights, const float* T, const float* top_flux,
                       float Lz, float* out, int n_columns);

/* ---- device-pointer twins: every pointer is device memory on cfg.device; work is enqueued on the
 * handle's stream and NOT synchronised.  d_out of loss_grad_dev has n_params + 8 floats:
 * [grad(n_params); scaled terms(6); total; 0] — the buffer a caller all-reduces (RCCL) when sharded. */
int colnde_set_problem_dev(colnde_handle* h, const float* d_x0, const float* d_bcs, const float* d_truth);
int colnde_rhs_dev(colnde_handle* h, const float* d_x, const float* d_weights, const float* d_bcs, float t,
                   float* d_dx, int n_columns);
int colnde_forward_dev(colnde_handle* h, const float* d_weights, float* d_sol);
int colnde_loss_dev(colnde_handle* h, const float* d_weights, const float scalings[6], float* d_out8);
int colnde_loss_grad_dev(colnde_handle* h, const float* d_weights, const float scalings[6], float* d_out);
int colnde_infer_forcing_dev(colnde_handle* h, const float* d_weights, const float* d_T, const float* d_top_flux,
                             float Lz, float* d_out, int n_columns);
int colnde_infer_dz_wT_dev(colnde_handle* h, const float* d_weights, const float* d_T, const float* d_top_flux,
                           float Lz, float* d_out, int n_columns);

/* ---- the steps either side of the hot path --------------------------------------------------------
 *
 * convective_adjustment!(model, Δt, K) — free_convection/double_gyre_nn.jl:27-62 (every (i,j) column of the 3-D model),
 * 1-D twin free_convection/src/oceananigans_nn.jl:13-40: κ_k = K where the centred ∂T/∂z of cell k is negative, 0 elsewhere,
 * then one backward-Euler step T' = L \ T with lower_k = -cκ_k, diag_k = 1 + c(κ_k + κ_{k+1}) (last row 1 + cκ_Nz),
 * upper_k = -cκ_{k+1}, c = Δt/Δz².  T, out: [n_col][Nz] (k = 0 deepest; out may alias T).  halo_bottom / halo_top: [n_col] values
 * of the halo cells below k = 0 / above k = Nz-1 as the ocean model filled them for the field's boundary conditions, or NULL for
 * the zero-gradient fill (nearest interior value).  Nz is the handle's. */
int colnde_convective_adjustment(colnde_handle* h, const float* T, const float* halo_bottom, const float* halo_top, float dt,
                                 float dz, float K, float* out, int n_columns);
int colnde_convective_adjustment_dev(colnde_handle* h, const float* d_T, const float* d_halo_bottom, const float* d_halo_top,
                                     float dt, float dz, float K, float* d_out, int n_columns);

/* modified_pacanowski_philander!(model, constants, Δt, p, convective_adjustment) — wind_mixing/src/NDE_oceananigans.jl:61-101, the implicit
 * diffusion step of the 1-D Oceananigans embedding of the wind-mixing NN (called every time step at :376,:402), with the face
 * diffusivities of modified_pacanowski_philander_diffusivity (:17-58):
 *   Ri_k  = ∂z b / ((∂z u)² + (∂z v)²) on faces, b = gαT (Oceanostics 0.3.2 richardson_number_ccf!, wind_mixing/Manifest.toml:1279)
 *   ν_k   = ν₀ + ν₋ tanh_step((Ri_k − Riᶜ)/ΔRi) on interior faces, 0 on the bottom face;  ν_T = ν/Pr, or under convective_adjustment
 *           Ri_k > 0 ? ν_k/Pr : 1 on every face (the bottom face sees the halo cells; NaN > 0 is false, as in Julia)
 *   u′ = L_ν \ u, v′ = L_ν \ v, T′ = L_νT \ T (backward Euler, the Tridiagonal of :69-83, c = Δt/Δz²), then T′[bottom] = T[bottom] (:94).
 * u, v, T and the outputs: [n_col][Nz] in the ocean model's units, k = 0 deepest; an output may alias its own input.  halo_bottom:
 * [3][n_col] = the u, v, T halo cells below k = 0 as the ocean model filled them for the fields' boundary conditions (only the
 * convective-adjustment switch of the bottom face reads them), or NULL for the zero-gradient fill.  params = {ν₀, ν₋, ΔRi, Riᶜ, Pr, α, g}
 * (the reference's `p` dictionary and `constants`).  Nz is the handle's (any model kind). */
int colnde_implicit_diffusion(colnde_handle* h, const float* u, const float* v, const float* T, const float* halo_bottom, float dt,
                              float dz, const float params[7], int convective_adjustment, float* u_out, float* v_out, float* T_out,
                              int n_columns);
int colnde_implicit_diffusion_dev(colnde_handle* h, const float* d_u, const float* d_v, const float* d_T, const float* d_halo_bottom,
                                  float dt, float dz, const float params[7], int convective_adjustment, float* d_u_out, float* d_v_out,
                                  float* d_T_out, int n_columns);

/* Flux.Optimise.ADAM apply! + update! (Flux 0.11.6 src/optimise/optimisers.jl; used at wind_mixing/src/NDE_training.jl:340-372,
 * free_convection/src/training.jl:71) on device vectors of n floats: m ← β₁m + (1-β₁)g, v ← β₂v + (1-β₂)g²,
 * w ← w - η·m/(1-β₁ᵗ)/(√(v/(1-β₂ᵗ)) + ϵ).  beta1_t / beta2_t are the running powers the optimiser state carries (β₁, β₂ on the
 * first call; the caller multiplies them by β after each call, as Flux does).  Enqueued on the handle's stream. */
int colnde_adam_step_dev(colnde_handle* h, float* d_weights, const float* d_grad, float* d_m, float* d_v, float eta, float beta1,
                         float beta2, float eps, float beta1_t, float beta2_t, int n);

/* Flux-MLP pre-training: one pass of `Flux.train!(NN_loss, Flux.params(NN), training_data, opt)` — `train_NN`,
 * wind_mixing/src/NN_training.jl:207-249 with the flux closures predict_uw / predict_vw / predict_wT (:25-169); T -> wT pre-training of
 * free_convection/train_free_convection_nde.jl:186-216 — i.e. ONE Flux-ADAM update per sample, in the order given:
 *   NN_flux = face vector of net `flux_type` (0 = uw, 1 = vw, 2 = wT; T-only models: 2) for the sample's profile and BCs, as the
 *             handle's conditions say (MPP / convective adjustment / zero_weights; the smoothing options are not covered);
 *   loss    = mse(NN_flux, flux) + gradient_scaling * mse(D^c flux, D^c NN_flux).
 * d_theta / d_m / d_v: the handle's full weight vector and its ADAM moments (n_params floats; only the net trained is touched);
 * d_profiles [n][n_state], d_bcs [n][n_bc], d_flux [n][Nz+1] scaled fluxes on faces, d_order [n] sample order (NULL: 0..n-1);
 * beta_t[2]: the optimiser's running powers (host, updated in place).  update = 0: no update, *mean_loss = mean loss at the given
 * weights (`total_loss(training_data)`, :234-236); update != 0: *mean_loss = mean of each sample's loss just before its update.
 * Synchronises the handle's stream. */
int colnde_pretrain_flux_dev(colnde_handle* h, int flux_type, float* d_theta, float* d_m, float* d_v, const float* d_profiles,
                             const float* d_bcs, const float* d_flux, const int32_t* d_order, int n_samples, float gradient_scaling,
                             float eta, float beta1, float beta2, float eps, double beta_t[2], int update, float* mean_loss);

/* Data preparation on device (wind_mixing/src/data_containers.jl:343-427).  d_in [n_rows][N] -> d_out [n_rows][n], one row per
 * profile.  location 0 = Center: coarse_grain(Φ, n, Center) (src/DataWrangling/coarse_graining.jl:8-16), block means, n divides N;
 * location 1 = Face: coarse_grain_linear_interpolation(Φ, n, Face) (:47-62), end points kept (the form data_containers.jl:357 uses). */
int colnde_coarse_grain_dev(colnde_handle* h, const float* d_in, int n_rows, int N, int n, int location, float* d_out);
/* ZeroMeanUnitVarianceScaling(data) (src/DataWrangling/feature_scaling.jl:17-20): d_mu_sigma[0] = mean, [1] = std (n-1 denominator)
 * of `count` device floats; colnde_scale_dev applies scale(x, s) = (x - μ)/σ (:22) with μ, σ read from device memory. */
int colnde_zscore_stats_dev(colnde_handle* h, const float* d_x, int64_t count, float* d_mu_sigma);
int colnde_scale_dev(colnde_handle* h, const float* d_x, int64_t count, const float* d_mu_sigma, float* d_out);

/* ---- multi-GPU: the path's one exchange step, over RCCL (xGMI inside a node) -----------------------------------
 * One process per GPU, columns sharded over the ranks (each handle normalises by colnde_set_global_columns); per optimiser
 * iteration every rank SUM-all-reduces its result buffer [grad(n_params); 6 terms; total; 0] and then applies the identical
 * ADAM step.  The reference has no distributed code: nothing is replaced here, the reference's serial comprehension over
 * simulations (NDE_training.jl:291,304) is what gets sharded.  RCCL is bound lazily (dlopen): these calls fail with a message
 * on a box without it.  Bootstrap: rank 0 calls colnde_comm_unique_id and hands the 128 bytes to every rank through any host
 * channel (MPI, a file, a TCP store); every rank then calls colnde_comm_create, which returns once all ranks have joined. */
typedef struct colnde_comm colnde_comm;
int  colnde_comm_unique_id(void* out128);
int  colnde_comm_create(int rank, int nranks, const void* unique_id128, int device, colnde_comm** out);
void colnde_comm_destroy(colnde_comm* c);
int  colnde_comm_rank(const colnde_comm* c);
int  colnde_comm_size(const colnde_comm* c);
/* in-place all-reduce of n device floats, enqueued on hip_stream (not synchronised); op 0 = sum, 1 = max */
int  colnde_comm_allreduce_dev(colnde_comm* c, float* d_buf, int64_t n, int op, void* hip_stream);
/* the result buffer of colnde_loss_grad_dev (n_params + 8 floats), summed over the communicator on the handle's stream */
int  colnde_allreduce_result_dev(colnde_handle* h, colnde_comm* comm, float* d_out);

/* How the handle runs its gradient path (filled in by the first colnde_loss_grad[_dev]; zeros before that):
 * info[0] engine (COLNDE_ENGINE_*), [1] columns per block of the gradient path (the tapes hold one block), [2] number of blocks, [3] regtile: layer-1
 * pre-activations taped (1) or recomputed (0); fc32: number of time segments the tapes are cut into (0: they hold the whole axis), [4] tile16: weight gradients taped (1) or accumulated in registers (0),
 * [5] tile16 taped mode: K-slices of the dW GEMM, [6] net-split kernels of the latency points (per 16-column tile one wavefront per flux net
 * plus a helper wavefront): bit 0 = forward solve, bit 1 = adjoint, bit 2 = with the rich tape (activations, their derivatives and the physics-pullback
 * coefficients taped by the forward kernel: blocks of at most 2,048 columns), [7] bit 0 = the gradient is the one-switch-pattern
 * RKC2 pullback, an approximation of the discrete adjoint (see COLNDE_STEPPER_RKC2); 0 = exact discrete adjoint of the stepper;
 * bits 1, 2, 3 = the forward-solve / adjoint / weight-gradient kernels of the handle's engine run BF16X3_EXACT (a cleared bit: F32_MFMA — the
 * configured arithmetic, a test override, or no split kernel for this engine and shape). */
int colnde_plan(const colnde_handle* h, int info[8]);

/* The same, spelled out, plus every tuning switch the library honours: one text line
 *   "engine=regtile stepper=rk4 substeps=2 matrix_arithmetic=bf16x3_exact forward=bf16x3 adjoint=bf16x3 dw=bf16x3 block=32768x1 ... | env COLNDE_RT_BLOCK=8192 ..."
 * The part after "| env" lists each COLNDE_* environment variable that is SET in this process and that the library reads (INTEGRATION.md has the
 * table): they are tuning and test aids, read when the handle is created or when it plans its tapes, never per call — and this is where a caller sees
 * that one of them shaped the handle.  buf may be NULL; returns the number of bytes the full line needs (including the terminating 0), or -1. */
int colnde_describe(const colnde_handle* h, char* buf, int capacity);

/* ---- measurement: HIP-event timing of the handle's kernels on its stream.
 * which: 0 = forward solve kernel, 1 = adjoint kernel, 2 = gradient reduce, 3 = rhs, 4 = inference,
 * 5 = streaming dW1 GEMM (regtile engine only), 6 = convective adjustment, 7 = ADAM step, 8 = implicit diffusion.
 * Returns accumulated milliseconds and launch count since the last reset (synchronises the stream). */
int colnde_set_profiling(colnde_handle* h, int enabled);
int colnde_kernel_time(colnde_handle* h, int which, float* ms_total, int* n_launches);
int colnde_reset_kernel_times(colnde_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* COLNDE_H */
