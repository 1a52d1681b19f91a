# ColumnNDE.jl — thin `ccall` layer over libcolnde.so (include/colnde.h) that keeps the reference's call signatures.
# NOT EXECUTED IN THIS REPOSITORY'S CI: the build image has no Julia (SURVEY §8c).  It mirrors, call for call, the ctypes front-end
# that IS tested (climateparameterizations.jl_amd/nde.py, wind_mixing.py, free_convection.py, distributed.py); the struct layout it
# relies on is checked field by field against include/colnde.h by tests/test_abi.py::test_julia_config_mirrors_the_header, and the
# same ABI is driven from plain C by tests/abi_smoke.c.
module ColumnNDE

using Flux, ChainRulesCore

const libcolnde = get(ENV, "COLNDE_LIB", "libcolnde.so")

# mirror of `colnde_config` (field order = include/colnde.h)
Base.@kwdef mutable struct Config
    model::Int32 = 0; Nz::Int32 = 32; n_layers::Int32 = 3
    layer_sizes::NTuple{9,Int32} = (96, 50, 20, 31, 0, 0, 0, 0, 0)
    activations::NTuple{8,Int32} = (2, 2, 0, 0, 0, 0, 0, 0)              # COLNDE_ACT_MISH, MISH, IDENTITY
    modified_pacanowski_philander::Int32 = 1; convective_adjustment::Int32 = 0; zero_weights::Int32 = 1
    smooth_NN::Int32 = 0; smooth_Ri::Int32 = 0; diurnal::Int32 = 0; train_gradient::Int32 = 1; inplace_variant::Int32 = 0
    H::Float32 = 256; tau::Float32 = 172800; f::Float32 = 1f-4; g::Float32 = 9.81f0; alpha::Float32 = 1.67f-4
    nu0::Float32 = 1f-4; nu_minus::Float32 = 1f-1; Ric::Float32 = 0.25f0; dRi::Float32 = 1f0; Pr::Float32 = 1f0
    kappa::Float32 = 10f0; eps::Float32 = 1f-7
    mu::NTuple{6,Float32} = (0, 0, 0, 0, 0, 0); sigma::NTuple{6,Float32} = (1, 1, 1, 1, 1, 1)
    ca_K::Float32 = 10f0
    n_save::Int32 = 2; substeps::Int32 = 2; save_times::Ptr{Float32} = C_NULL
    n_columns::Int32 = 1; device::Int32 = 0; engine::Int32 = 0
    stepper::Int32 = 0; rkc_stages::Int32 = 0                            # COLNDE_STEPPER_RK4 / _RKC2 (stands where the reference uses ROCK4)
    matrix_arithmetic::Int32 = 0                                         # COLNDE_MATRIX_BF16X3_EXACT (0, default) / COLNDE_MATRIX_F32_MFMA (1)
    reltol::Float32 = 0                                                  # solve(...; reltol=1f-3) (0 = 1f-3); with substeps = 0 the handle chooses the sub-step count from it
end

const MATRIX_BF16X3_EXACT = Int32(0)   # Float32 Dense products as six bf16 MFMA products of exact three-way operand splits, f32 accumulation
const MATRIX_F32_MFMA = Int32(1)       # v_mfma_f32_* throughout

check(rc) = rc == 0 || error(unsafe_string(ccall((:colnde_last_error, libcolnde), Cstring, ())))

mutable struct Handle
    ptr::Ptr{Cvoid}; n_params::Int; n_state::Int; n_save::Int; n_columns::Int; n_nets::Int
end

"least RK4 sub-steps per save interval inside the stability bound / stages of the RKC2 step (no GPU needed)"
function min_substeps(cfg::Config, save_times::Vector{Float32})
    cfg.n_save = length(save_times)
    GC.@preserve save_times begin
        cfg.save_times = pointer(save_times)
        ccall((:colnde_min_substeps, libcolnde), Cint, (Ref{Config},), cfg)
    end
end
function rkc_stages(cfg::Config, save_times::Vector{Float32})
    cfg.n_save = length(save_times)
    GC.@preserve save_times begin
        cfg.save_times = pointer(save_times)
        ccall((:colnde_rkc_stages, libcolnde), Cint, (Ref{Config},), cfg)
    end
end

"constants/scalings/conditions as built by prepare_parameters_NDE_training (NDE_training.jl:1-44); t_train ./ τ as save_times"
function Handle(cfg::Config, save_times::Vector{Float32})
    cfg.n_save = length(save_times)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve save_times begin
        cfg.save_times = pointer(save_times)
        check(ccall((:colnde_create, libcolnde), Cint, (Ref{Config}, Ref{Ptr{Cvoid}}), cfg, out))
    end
    h = Handle(out[], ccall((:colnde_n_params, libcolnde), Cint, (Ptr{Cvoid},), out[]),
               cfg.model == 0 ? 3cfg.Nz : cfg.Nz, cfg.n_save, cfg.n_columns, cfg.model == 0 ? 3 : 1)
    finalizer(x -> ccall((:colnde_destroy, libcolnde), Cvoid, (Ptr{Cvoid},), x.ptr), h)
end

"uvT₀s, BCs, uvT_trains of train_NDE (NDE_training.jl:220-243): one column per simulation"
set_problem!(h::Handle, uvT₀s::Matrix{Float32}, BCs::Matrix{Float32}, uvT_trains::Union{Nothing,Array{Float32,3}}) =
    check(ccall((:colnde_set_problem, libcolnde), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                h.ptr, uvT₀s, BCs, uvT_trains === nothing ? C_NULL : uvT_trains))   # Julia 96×n / 6×n / 96×Nt×n arrays ARE the C layouts

"NDE(x, p, t) with p = [weights; BCs] — wind_mixing/src/NDE_training.jl:56-66"
function NDE(h::Handle, x::Vector{Float32}, p::Vector{Float32}, t)
    dx = similar(x)
    NDE!(h, dx, x, p, t); dx
end

"diurnal overload NDE(x, p, t) — NDE_training.jl:68-81: `p` carries only 5 BCs, wT_top(t) comes from Qᵇ (parsed from the file name,
data_containers.jl:138-152); create the handle with diurnal = 1 and pass Qᵇ here (it fills the sixth BC slot of the C ABI)"
function NDE(h::Handle, x::Vector{Float32}, p::Vector{Float32}, t, Qᵇ::Real)
    length(p) == h.n_params + 5 || error("the diurnal NDE takes p = [weights; uw_b, uw_t, vw_b, vw_t, wT_b]")
    NDE(h, x, vcat(p, Float32(Qᵇ)), t)
end

"NDE!(dx, x, p, t) — wind_mixing/src/training_postprocessing.jl:131-153 (create the handle with inplace_variant = 1)"
function NDE!(h::Handle, dx, x, p, t)
    w = @view p[1:h.n_params]; bc = @view p[h.n_params+1:end]
    check(ccall((:colnde_rhs, libcolnde), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Cfloat, Ptr{Float32}, Cint),
                h.ptr, x, w, bc, Float32(t), dx, 1))
    nothing
end

"[Array(solve(prob_NDEs[i], …; p=[weights; BCs[i]], saveat=t_train)) for i in 1:n_simulations] — NDE_training.jl:291,403"
function solve_NDE(h::Handle, weights::Vector{Float32})
    sol = Array{Float32}(undef, h.n_state, h.n_save, h.n_columns)
    check(ccall((:colnde_forward, libcolnde), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}), h.ptr, weights, sol))
    [sol[:, :, i] for i in 1:h.n_columns]
end

const KEYS = (:u, :v, :T, :∂u∂z, :∂v∂z, :∂T∂z)

"loss_NDE / loss_gradient_NDE(weights, BCs) — NDE_training.jl:290-323: (total, scaled_losses, loss_scalings)"
function loss_gradient_NDE(h::Handle, weights, loss_scalings::NamedTuple)
    sc = Float32[loss_scalings[k] for k in KEYS]; terms = zeros(Float32, 6); total = Ref{Float32}(0)
    check(ccall((:colnde_loss, libcolnde), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ref{Float32}),
                h.ptr, weights, sc, terms, total))
    total[], NamedTuple{KEYS}(Tuple(terms)), loss_scalings
end

"∇loss: value and gradient in one call (replaces Zygote through InterpolatingAdjoint — NDE_training.jl:327-333)"
function ∇loss(h::Handle, weights, loss_scalings::NamedTuple)
    sc = Float32[loss_scalings[k] for k in KEYS]; terms = zeros(Float32, 6); total = Ref{Float32}(0)
    grad = similar(weights)
    check(ccall((:colnde_loss_grad, libcolnde), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ref{Float32}, Ptr{Float32}),
                h.ptr, weights, sc, terms, total, grad))
    total[], NamedTuple{KEYS}(Tuple(terms)), grad
end

# OptimizationFunction(loss, AutoZygote()) keeps working: the rrule routes the pullback to the HIP adjoint
function ChainRulesCore.rrule(::typeof(loss_gradient_NDE), h::Handle, weights, loss_scalings)
    total, losses, grad = ∇loss(h, weights, loss_scalings)
    pullback(ȳ) = (NoTangent(), NoTangent(), ȳ[1] .* grad, NoTangent())
    (total, losses, loss_scalings), pullback
end

# ---- free convection: FreeConvectionNDE / ConvectiveAdjustmentNDE (model = 1 / 2) -------------------------------------------

"∂T∂t(T, p, t) with p = [weights; bottom_flux, top_flux, σ_T, σ_wT, H, τ] — free_convection/src/free_convection_nde.jl:29-38,
convective_adjustment_nde.jl:33-48 (the four trailing scalars are constants of the handle's configuration)"
function ∂T∂t(h::Handle, T::Vector{Float32}, p::Vector{Float32}, t)
    dT = similar(T); w = @view p[1:h.n_params]; bc = @view p[h.n_params+1:h.n_params+2]
    check(ccall((:colnde_rhs, libcolnde), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Cfloat, Ptr{Float32}, Cint),
                h.ptr, T, w, bc, Float32(t), dT, 1))
    dT
end

"solve_nde(nde, NN, T₀, alg, nde_params) for every simulation of the handle — free_convection/src/solve.jl:1-6: Nz × Nt per simulation"
solve_nde(h::Handle, NN) = solve_NDE(h, first(Flux.destructure(NN)))

"nde_loss() = Flux.mse(cat(nde_sols...), true_sols) [+ causal_penalty(NN)] — free_convection/src/training.jl:55-62"
function nde_loss(h::Handle, NN; causal_penalty=nothing)
    θ = first(Flux.destructure(NN))
    terms = zeros(Float32, 6); total = Ref{Float32}(0)
    check(ccall((:colnde_loss, libcolnde), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ref{Float32}),
                h.ptr, θ, Float32[0, 0, 1, 0, 0, 0], terms, total))
    causal_penalty === nothing ? total[] : total[] + causal_penalty(NN)
end

"value of nde_loss and its gradient as a `Zygote.Grads` keyed by the arrays of `Flux.params(NN)` — what `Flux.train!`
(training.jl:71) obtains from `gradient(() -> nde_loss(), ps)`.  `Flux.destructure` concatenates exactly the arrays of
`Flux.params(NN)` in order (W₁, b₁, W₂, b₂, …), so the flat HIP gradient is cut back along the same sizes."
function nde_loss_gradient(h::Handle, NN; causal_penalty=nothing)
    θ = first(Flux.destructure(NN)); ps = Flux.params(NN)
    terms = zeros(Float32, 6); total = Ref{Float32}(0); grad = similar(θ)
    check(ccall((:colnde_loss_grad, libcolnde), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ref{Float32}, Ptr{Float32}),
                h.ptr, θ, Float32[0, 0, 1, 0, 0, 0], terms, total, grad))
    gs = IdDict{Any,Any}(); o = 0
    for p in ps
        gs[p] = reshape(grad[o+1:o+length(p)], size(p)); o += length(p)
    end
    loss = total[]
    if causal_penalty !== nothing                      # a function of the weights alone: Zygote differentiates it as before
        pen, back = Flux.Zygote.pullback(() -> causal_penalty(NN), ps)
        gp = back(one(pen)); loss += pen
        for p in ps
            gp[p] === nothing || (gs[p] = gs[p] .+ gp[p])
        end
    end
    loss, Flux.Zygote.Grads(gs, ps)
end

"train_neural_differential_equation! — training.jl:44-74: `Flux.train!(nde_loss, Flux.params(NN), repeated((), epochs), opt, cb)`
with the gradient from the HIP adjoint; `opt`'s state is keyed by the arrays of `Flux.params(NN)` and persists across calls, as in Flux"
function train_neural_differential_equation!(h::Handle, NN, opt, epochs; cb=() -> nothing, causal_penalty=nothing)
    ps = Flux.params(NN)
    for _ in 1:epochs
        _, gs = nde_loss_gradient(h, NN; causal_penalty)
        Flux.Optimise.update!(opt, ps, gs)
        cb()
    end
    nothing
end

"compute_neural_network_forcing!(params, model) — free_convection/double_gyre_nn.jl:149-168: T_interior is `interior(model.tracers.T)`
permuted to (Nz, Nx·Ny); surface_flux (Nx·Ny) is the relaxation flux of :163; writes −∂z wT into `forcing` (Nz, Nx·Ny)"
function compute_neural_network_forcing!(forcing::Matrix{Float32}, h::Handle, weights::Vector{Float32}, T_interior::Matrix{Float32},
                                         surface_flux::Vector{Float32}, Lz)
    check(ccall((:colnde_infer_forcing, libcolnde), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Cfloat, Ptr{Float32}, Cint),
                h.ptr, weights, T_interior, surface_flux, Float32(Lz), forcing, size(T_interior, 2)))
    forcing
end

"predict_flux(uvT, BCs, ...) — wind_mixing/src/NDE_training.jl:83-147 (exported at WindMixing.jl:6): (uw, vw, wT) on the Nz+1 faces for ONE state, as the
reference returns them; `p = [weights; BCs]` as for NDE"
function predict_flux(h::Handle, x::Vector{Float32}, p::Vector{Float32}, t=0)
    w = @view p[1:h.n_params]; bc = @view p[h.n_params+1:end]
    nn = h.n_nets                                                          # wind mixing: three nets on the 3 Nz state
    Nz = h.n_state ÷ nn
    fl = Matrix{Float32}(undef, Nz + 1, nn)                                 # column-major (Nz+1) x nn = C-order [nn][Nz+1]
    check(ccall((:colnde_flux, libcolnde), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Cfloat, Ptr{Float32}, Cint),
                h.ptr, x, w, bc, Float32(t), fl, 1))
    nn == 3 ? (fl[:, 1], fl[:, 2], fl[:, 3]) : fl[:, 1]
end

"loss_per_tstep (wind_mixing/src/loss.jl:44-46) of the six profile terms of every simulation: n_save x 6 x n_simulations"
function loss_per_tstep(h::Handle, weights::Vector{Float32})
    out = Array{Float32}(undef, h.n_save, 6, h.n_columns)
    check(ccall((:colnde_loss_per_tstep, libcolnde), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}), h.ptr, weights, out))
    out
end

"dataset-level solve_nde(ds, NN, NDEType, algorithm, T_scaling, wT_scaling) — free_convection/src/solve.jl:8-51: the solution of every simulation and the
flux wT re-evaluated at each saved step (min(0, 10 ∂T/∂z) included for ConvectiveAdjustmentNDE), both unscaled: (T = Nz x Nt x n, wT = (Nz+1) x Nt x n).
`BCs` (2 x n: scaled bottom and top flux) are the ones given to set_problem!."
function solve_nde(h::Handle, NN, BCs::Matrix{Float32}, T_scaling, wT_scaling)
    θ = first(Flux.destructure(NN))
    sols = solve_NDE(h, θ)
    h.n_nets == 1 || error("solve_nde(ds-level) is the free-convection driver: a T-only handle")
    Nz = h.n_state
    T = Array{Float32}(undef, Nz, h.n_save, h.n_columns); wT = Array{Float32}(undef, Nz + 1, h.n_save, h.n_columns)
    for i in 1:h.n_columns, n in 1:h.n_save
        T[:, n, i] .= sols[i][:, n]
        wT[:, n, i] .= predict_flux(h, sols[i][:, n], vcat(θ, BCs[:, i]), 0)
    end
    (T=inv(T_scaling).(T), wT=inv(wT_scaling).(wT))
end

"Richardson estimate of the solve's error at the current sub-step count (max |e| / (1f-3 + |u|): colnde_error_estimate) — what `reltol` bounds here"
function error_estimate(h::Handle, weights::Vector{Float32})
    est = Ref{Float32}(0)
    check(ccall((:colnde_error_estimate, libcolnde), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ref{Float32}), h.ptr, weights, est)); est[]
end
"the least power-of-two sub-step count meeting reltol (0: the handle's); kept by the handle — colnde_choose_substeps"
function choose_substeps!(h::Handle, weights::Vector{Float32}, reltol=0f0)
    s = Ref{Cint}(0); est = Ref{Float32}(0)
    check(ccall((:colnde_choose_substeps, libcolnde), Cint, (Ptr{Cvoid}, Ptr{Float32}, Cfloat, Ref{Cint}, Ref{Float32}), h.ptr, weights, reltol, s, est))
    Int(s[]), est[]
end
substeps(h::Handle) = ccall((:colnde_substeps, libcolnde), Cint, (Ptr{Cvoid},), h.ptr)
"impose a sub-step count (column shards: choose_substeps! on every rank, MAX over ranks, set_substeps! — colnde_set_substeps)"
set_substeps!(h::Handle, n::Integer) = (check(ccall((:colnde_set_substeps, libcolnde), Cint, (Ptr{Cvoid}, Cint), h.ptr, n)); h)

"+∂z wT as compute_neural_network_forcing! stores it in params.∂z_wT_NN (double_gyre_nn.jl:165; the forcing function negates it, :135)"
function compute_∂z_wT!(dz_wT::Matrix{Float32}, h::Handle, weights::Vector{Float32}, T_interior::Matrix{Float32}, surface_flux::Vector{Float32}, Lz)
    check(ccall((:colnde_infer_dz_wT, libcolnde), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Cfloat, Ptr{Float32}, Cint),
                h.ptr, weights, T_interior, surface_flux, Float32(Lz), dz_wT, size(T_interior, 2)))
    dz_wT
end

"switch an existing handle between MATRIX_BF16X3_EXACT and MATRIX_F32_MFMA (tapes and plans do not depend on it) — colnde_set_matrix_arithmetic"
set_matrix_arithmetic!(h::Handle, ma) = check(ccall((:colnde_set_matrix_arithmetic, libcolnde), Cint, (Ptr{Cvoid}, Cint), h.ptr, ma))
matrix_arithmetic(h::Handle) = ccall((:colnde_matrix_arithmetic, libcolnde), Cint, (Ptr{Cvoid},), h.ptr)

"how the gradient path runs: (engine, block_columns, n_blocks, z1_taped [regtile], time_segments [fc32], dw_taped, dw_slices, split_forward,
split_adjoint, split_rich_tape, approximate_gradient = the one-switch-pattern RKC2 pullback, see COLNDE_STEPPER_RKC2 in colnde.h,
bf16x3_forward / _adjoint / _dw = which kernel families run the exact three-way bf16 split) — colnde_plan (info[3] is the Z1-tape flag on
the regtile engine, engine 2, and the number of time segments on fc32, engine 3: the same mapping as nde.py's plan())"
function plan(h::Handle)
    info = zeros(Cint, 8)
    check(ccall((:colnde_plan, libcolnde), Cint, (Ptr{Cvoid}, Ptr{Cint}), h.ptr, info))
    engine = info[1]
    (engine=engine, block_columns=info[2], n_blocks=info[3], z1_taped=info[4] != 0 && engine == 2, time_segments=engine == 3 ? Int(info[4]) : 0,
     dw_taped=info[5] != 0, dw_slices=info[6],
     split_forward=(info[7] & 1) != 0, split_adjoint=(info[7] & 2) != 0, split_rich_tape=(info[7] & 4) != 0,
     approximate_gradient=(info[8] & 1) != 0, bf16x3_forward=(info[8] & 2) != 0, bf16x3_adjoint=(info[8] & 4) != 0, bf16x3_dw=(info[8] & 8) != 0)
end

"""The plan spelled out, plus every COLNDE_* tuning switch set in this process that the library reads — colnde_describe."""
function describe(h::Handle)
    need = ccall((:colnde_describe, libcolnde), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint), h.ptr, C_NULL, 0)
    need > 0 || error(unsafe_string(ccall((:colnde_last_error, libcolnde), Cstring, ())))
    buf = Vector{UInt8}(undef, need)
    ccall((:colnde_describe, libcolnde), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint), h.ptr, buf, need)
    return unsafe_string(pointer(buf))
end

"The reference's closures at the reference's ARITIES, closed over a handle: what `ODEProblem`, `OptimizationFunction` and `Flux.train!` are handed
today.  `NDE(x, p, t)` (NDE_training.jl:56), `NDE!(dx, x, p, t)` (training_postprocessing.jl:131; create that handle with inplace_variant = 1),
`loss_NDE(weights, BCs)` / `loss_gradient_NDE(weights, BCs)` (NDE_training.jl:290-323; BCs were given to set_problem! and are ignored here, as the
reference's closures capture everything but the weights), `∂T∂t(T, p, t)` (free_convection_nde.jl:29)."
function reference_closures(h::Handle, loss_scalings::NamedTuple=(u=1f0, v=1f0, T=1f0, ∂u∂z=5f-3, ∂v∂z=5f-3, ∂T∂z=5f-3))
    (NDE=(x, p, t) -> NDE(h, x, p, t),
     NDE! =(dx, x, p, t) -> NDE!(h, dx, x, p, t),
     ∂T∂t=(T, p, t) -> ∂T∂t(h, T, p, t),
     # loss_NDE (NDE_training.jl:290-301, the train_gradient = false objective) sets ∂u∂z = ∂v∂z = ∂T∂z = 0 before the scalings are applied
     # (and it returns the UNCHANGED loss_scalings as its third value, :300)
     loss_NDE=(weights, BCs) -> begin
         total, scaled, _ = loss_gradient_NDE(h, weights, merge(loss_scalings, (∂u∂z=0f0, ∂v∂z=0f0, ∂T∂z=0f0)))
         (total, scaled, loss_scalings)
     end,
     loss_gradient_NDE=(weights, BCs) -> loss_gradient_NDE(h, weights, loss_scalings))
end

# ---- multi-GPU: one Julia process per GPU, columns sharded, ONE exchange per optimiser iteration (include/colnde.h, colnde_comm_*) ----
mutable struct Comm
    ptr::Ptr{Cvoid}
end
"rank 0 makes the 128-byte RCCL id and hands it to every rank (MPI.Bcast!, a file, a socket) before `Comm(...)`"
function comm_unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:colnde_comm_unique_id, libcolnde), Cint, (Ptr{UInt8},), id)); id
end
function Comm(rank, nranks, unique_id::Vector{UInt8}, device=0)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:colnde_comm_create, libcolnde), Cint, (Cint, Cint, Ptr{UInt8}, Cint, Ref{Ptr{Cvoid}}), rank, nranks, unique_id, device, out))
    c = Comm(out[]); finalizer(x -> ccall((:colnde_comm_destroy, libcolnde), Cvoid, (Ptr{Cvoid},), x.ptr), c)
end
"set_global_columns!(h, N): losses and gradients of this rank are normalised by the GLOBAL simulation count (NDE_training.jl:312-317)"
set_global_columns!(h::Handle, N) = check(ccall((:colnde_set_global_columns, libcolnde), Cint, (Ptr{Cvoid}, Int64), h.ptr, N))
"SUM over the ranks of the device result buffer [grad; 6 terms; total; 0] of colnde_loss_grad_dev, on the handle's stream"
allreduce_result!(h::Handle, c::Comm, d_out::Ptr{Float32}) =
    check(ccall((:colnde_allreduce_result_dev, libcolnde), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float32}), h.ptr, c.ptr, d_out))

"convective_adjustment!(model, Δt, K) — free_convection/double_gyre_nn.jl:27-62: T is `interior(model.tracers.T)` permuted to
(Nz, Nx·Ny) column-major (= C-order [column][level]); halos are `nothing` for flux-bounded T (zero-gradient fill) or the two
halo planes Oceananigans filled"
function convective_adjustment!(h::Handle, T::Matrix{Float32}, Δt, Δz, K; halo_bottom=nothing, halo_top=nothing)
    hb = halo_bottom === nothing ? C_NULL : pointer(halo_bottom); ht = halo_top === nothing ? C_NULL : pointer(halo_top)
    GC.@preserve halo_bottom halo_top check(ccall((:colnde_convective_adjustment, libcolnde), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Cfloat, Cfloat, Cfloat, Ptr{Float32}, Cint),
        h.ptr, T, hb, ht, Δt, Δz, K, T, size(T, 2)))
    T
end

"modified_pacanowski_philander!(model, constants, Δt, p, convective_adjustment) — wind_mixing/src/NDE_oceananigans.jl:61-101: u, v, T are
`interior(model.velocities.u)[:]` … as (Nz, n_columns) column-major matrices (= C-order [column][level]), updated in place; `p` is the
reference's diffusivity dictionary, `constants` its NamedTuple; halo_bottom (n_columns, 3) column-major = C-order [3][n_columns] or nothing"
function modified_pacanowski_philander!(h::Handle, u::Matrix{Float32}, v::Matrix{Float32}, T::Matrix{Float32}, constants, Δt, Δz, p,
                                        convective_adjustment; halo_bottom=nothing)
    params = Float32[p["ν₀"], p["ν₋"], p["ΔRi"], p["Riᶜ"], p["Pr"], constants.α, constants.g]
    hb = halo_bottom === nothing ? C_NULL : pointer(halo_bottom)
    GC.@preserve halo_bottom check(ccall((:colnde_implicit_diffusion, libcolnde), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Cfloat, Cfloat, Ptr{Float32}, Cint,
         Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Cint),
        h.ptr, u, v, T, hb, Δt, Δz, params, convective_adjustment ? 1 : 0, u, v, T, size(T, 2)))
    nothing
end

"Flux.Optimise.ADAM apply!/update! on device pointers (θ, ∇, m, v resident on the GPU); βᵗ = running powers kept by the caller"
function adam_step!(h::Handle, dθ::Ptr{Float32}, dg::Ptr{Float32}, dm::Ptr{Float32}, dv::Ptr{Float32}, η, β, ϵ, βᵗ, n)
    check(ccall((:colnde_adam_step_dev, libcolnde), Cint,
        (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Cfloat, Cfloat, Cfloat, Cfloat, Cfloat, Cfloat, Cint),
        h.ptr, dθ, dg, dm, dv, η, β[1], β[2], ϵ, βᵗ[1], βᵗ[2], n))
    βᵗ .* β
end

"one `Flux.train!(NN_loss, Flux.params(NN), training_data, opt)` pass of train_NN (NN_training.jl:207-249) on device arrays: one ADAM
update per sample in `order`; returns (mean loss, βᵗ).  update = false evaluates `total_loss(training_data)` at fixed weights."
function pretrain_flux!(h::Handle, flux_type, dθ::Ptr{Float32}, dm::Ptr{Float32}, dv::Ptr{Float32}, dX::Ptr{Float32}, dBCs::Ptr{Float32},
                        dflux::Ptr{Float32}, dorder::Ptr{Int32}, n, gradient_scaling, η, β, ϵ, βᵗ::Vector{Float64}; update=true)
    loss = Ref{Float32}(0)
    check(ccall((:colnde_pretrain_flux_dev, libcolnde), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Int32}, Cint,
         Cfloat, Cfloat, Cfloat, Cfloat, Cfloat, Ptr{Float64}, Cint, Ref{Float32}),
        h.ptr, flux_type, dθ, dm, dv, dX, dBCs, dflux, dorder, n, gradient_scaling, η, β[1], β[2], ϵ, βᵗ, update ? 1 : 0, loss))
    loss[], βᵗ
end

end # module
