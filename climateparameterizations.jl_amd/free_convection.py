"""Host-side mirror of the reference's free-convection NDE interface (`FreeConvection` package).

    FreeConvectionNDE / ConvectiveAdjustmentNDE(NN, ds; iterations)     free_convection/src/free_convection_nde.jl:1-47,
                                                                        convective_adjustment_nde.jl:1-57   -> FreeConvectionNDE class
    FreeConvectionNDEParameters(ds, T_scaling, wT_scaling)               free_convection_nde.jl:49-62        -> nde_params rows [bottom, top]
    solve_nde(nde, NN, T₀, alg, nde_params)                              free_convection/src/solve.jl:1-6
    solve_nde(ds, NN, NDEType, algorithm, T_scaling, wT_scaling) -> (T, wT)  free_convection/src/solve.jl:8-51 -> solve_nde_dataset
    nde_loss()  and the Flux.train! loop                                  free_convection/src/training.jl:44-74
    compute_neural_network_forcing!                                       free_convection/double_gyre_nn.jl:149-168
    convective_adjustment!(model, Δt, K)                                  free_convection/double_gyre_nn.jl:27-62, src/oceananigans_nn.jl:13-40

The Oceananigans `FieldDataset` wrangling and DataDeps download stay outside (network + foreign types).
"""
from __future__ import annotations

from typing import Callable, List, Optional

import numpy as np

from .config import NDEConfig, FREE_CONVECTION, CONVECTIVE_ADJUSTMENT_NDE
from .flux_compat import ADAM
from .nde import ColumnNDE


class FreeConvectionNDE:
    """One NDE per simulation in the reference (`ndes[id]`); here all simulations are columns of one handle."""

    def __init__(self, cfg: NDEConfig, T0, nde_params, true_sols=None, device: int = 0,
                 causal_penalty: Optional[Callable] = None):
        """`causal_penalty`: the optional term of `nde_loss` (training.jl:44,57-58: `Flux.mse(...) + causal_penalty(NN)`), a
        function of the weights alone; here a callable θ -> (value, ∂value/∂θ) since there is no Zygote to differentiate it."""
        if cfg.model not in (FREE_CONVECTION, CONVECTIVE_ADJUSTMENT_NDE):
            raise ValueError("need a free-convection config")
        self.cfg = cfg
        self.causal_penalty = causal_penalty
        T0 = np.ascontiguousarray(T0, dtype=np.float32)
        self.n_simulations = T0.shape[0]
        self.engine = ColumnNDE(cfg, self.n_simulations, device=device)
        self.engine.set_problem(T0, np.ascontiguousarray(nde_params, dtype=np.float32), true_sols)

    def dTdt(self, T, p, t=0.0):
        """`∂T∂t(T, p, t)` with p = [weights; bottom_flux, top_flux, σ_T, σ_wT, H, τ] (free_convection_nde.jl:29-38);
        the four trailing scalars must equal the config's (they are compile-time constants of the handle)."""
        p = np.asarray(p, dtype=np.float32)
        n = self.cfg.n_params
        tail = p[n:]
        c = self.cfg
        expect = np.array([c.sigma[2], c.sigma[5], c.H, c.tau], dtype=np.float32)
        if tail.shape[0] != 6 or not np.allclose(tail[2:], expect, rtol=1e-6):
            raise ValueError("p tail must be [bottom, top, σ_T, σ_wT, H, τ] matching the handle's configuration")
        T2 = np.atleast_2d(np.asarray(T, dtype=np.float32))
        out = self.engine.rhs(T2, p[:n], np.broadcast_to(tail[None, :2], (T2.shape[0], 2)), float(t))
        return out[0] if np.ndim(T) == 1 else out

    def solve_nde(self, weights):
        """`solve(nde, alg; reltol=1e-4, u0=T₀, p=[w; nde_params])` per simulation → [n_sims, Nz, Nt]."""
        return np.transpose(self.engine.forward(weights), (0, 2, 1))

    def nde_loss(self, weights) -> float:
        """`Flux.mse(cat(nde_sols…), true_sols)` (training.jl:55-62)."""
        total, _ = self.engine.loss(weights, [0, 0, 1, 0, 0, 0])
        if self.causal_penalty is not None:
            total += float(self.causal_penalty(np.asarray(weights, dtype=np.float32))[0])
        return total

    def nde_loss_and_grad(self, weights):
        total, _, grad = self.engine.loss_grad(weights, [0, 0, 1, 0, 0, 0])
        if self.causal_penalty is not None:
            pv, pg = self.causal_penalty(np.asarray(weights, dtype=np.float32))
            total, grad = total + float(pv), grad + np.asarray(pg, dtype=grad.dtype)
        return total, grad

    def close(self):
        self.engine.close()


def train_neural_differential_equation(nde: FreeConvectionNDE, weights, opt: ADAM, epochs: int,
                                       cb: Optional[Callable] = None):
    """`Flux.train!(nde_loss, Flux.params(NN), repeated((), epochs), opt, cb)` (training.jl:71)."""
    theta = np.array(weights, dtype=np.float32)
    history: List[float] = []
    for _ in range(epochs):
        total, grad = nde.nde_loss_and_grad(theta)
        history.append(total)
        opt.update(theta, grad.astype(np.float64))
        if cb is not None:
            cb(theta, total)
    return theta, history


def solve_nde_dataset(engine: ColumnNDE, weights, nde_params):
    """Dataset-level `solve_nde(ds, NN, NDEType, algorithm, T_scaling, wT_scaling)` (free_convection/src/solve.jl:8-51) for every simulation of the
    handle (set_problem gave it T₀ and the scaled [bottom, top] fluxes `nde_params`): the solution at the save points and the flux re-evaluated at each
    of them — `wT_NN_n = [bottom; NN(T_n); top]`, minus `min(0, 10 ∂T/∂z)` for ConvectiveAdjustmentNDE (:32-46) — both UNSCALED
    (`inv(T_scaling).(T)`, `inv(wT_scaling).(wT)`).  Returns (T [n, Nt, Nz], wT [n, Nt, Nz + 1])."""
    c = engine.cfg
    sol = engine.forward(np.asarray(weights, np.float32))                       # [n, Nt, Nz], scaled
    n, nt, nz = sol.shape
    bc = np.repeat(np.asarray(nde_params, np.float32)[:, None, :2], nt, axis=1).reshape(n * nt, 2)
    wT = engine.flux(sol.reshape(n * nt, nz), weights, bc).reshape(n, nt, nz + 1)
    return c.sigma[2] * sol + c.mu[2], c.sigma[5] * wT + c.mu[5]


def compute_neural_network_dz_wT(engine: ColumnNDE, weights, T_interior, surface_flux, Lz: float):
    """What `compute_neural_network_forcing!` stores: `params.∂z_wT_NN .= ∂z_wT(wT)` (double_gyre_nn.jl:165), +∂z wT; the forcing function
    `neural_network_∂z_wT` negates it (:135) — `compute_neural_network_forcing` below returns that forcing."""
    T = np.asarray(T_interior, dtype=np.float32)
    nx, ny, nz = T.shape
    out = engine.infer_dz_wT(weights, T.reshape(nx * ny, nz), np.asarray(surface_flux, np.float32).reshape(-1), Lz)
    return out.reshape(nx, ny, nz)


def compute_neural_network_forcing(engine: ColumnNDE, weights, T_interior, surface_flux, Lz: float):
    """`compute_neural_network_forcing!` (double_gyre_nn.jl:149-168): T_interior [Nx, Ny, Nz] model units,
    surface_flux [Nx, Ny]; returns the T-forcing array −∂z wT of the same shape."""
    T = np.asarray(T_interior, dtype=np.float32)
    nx, ny, nz = T.shape
    out = engine.infer_forcing(weights, T.reshape(nx * ny, nz), np.asarray(surface_flux, np.float32).reshape(-1), Lz)
    return out.reshape(nx, ny, nz)


def convective_adjustment(engine: ColumnNDE, T_interior, dt: float, K: float, dz: float, halo_bottom=None, halo_top=None):
    """`convective_adjustment!(model, Δt, K)` (double_gyre_nn.jl:27-62; 1-D: src/oceananigans_nn.jl:13-40) on the T
    interior [Nx, Ny, Nz] (or [n, Nz]): returns Tⁿ⁺¹ of the same shape.  halo_bottom / halo_top [Nx, Ny]: the halo
    cells Oceananigans filled for T's boundary conditions (None: zero-gradient fill)."""
    T = np.asarray(T_interior, dtype=np.float32)
    shape = T.shape
    T2 = T.reshape(-1, shape[-1])
    hb = None if halo_bottom is None else np.asarray(halo_bottom, np.float32).reshape(-1)
    ht = None if halo_top is None else np.asarray(halo_top, np.float32).reshape(-1)
    return engine.convective_adjustment(T2, dt, dz, K, hb, ht).reshape(shape)


def train_neural_differential_equation_device(nde: FreeConvectionNDE, weights, opt: ADAM, epochs: int, process_group=None, comm=None):
    """`Flux.train!` (training.jl:71) with θ and the ADAM state resident on the GPU: per epoch one `colnde_loss_grad_dev`,
    [one SUM all-reduce when the simulations are sharded over `process_group`], one fused `colnde_adam_step_dev`.
    Returns (θ, loss history) like `train_neural_differential_equation`; the per-epoch callback is not available here."""
    import torch
    eng = nde.engine
    dev = torch.device("cuda", eng.device)
    n = eng.n_params
    theta = torch.as_tensor(np.asarray(weights, dtype=np.float32)).to(dev).contiguous()
    out = torch.empty(n + 8, dtype=torch.float32, device=dev)
    m = torch.zeros(n, dtype=torch.float32, device=dev)
    v = torch.zeros(n, dtype=torch.float32, device=dev)
    if opt.m is not None:
        m.copy_(torch.as_tensor(opt.m, dtype=torch.float32)); v.copy_(torch.as_tensor(opt.v, dtype=torch.float32))
    hist = []
    for _ in range(epochs):
        eng.loss_grad(theta, [0, 0, 1, 0, 0, 0], out=out)
        if comm is not None:
            comm.allreduce_result(eng, out)
        elif process_group is not None:
            import torch.distributed as dist
            dist.all_reduce(out, op=dist.ReduceOp.SUM, group=process_group)
        hist.append(out[n + 6].clone())
        eng.adam_step(theta, out, m, v, opt.eta, opt.beta, opt.eps, beta_t=tuple(opt.beta_t))
        opt.beta_t[0] *= opt.beta[0]
        opt.beta_t[1] *= opt.beta[1]
    opt.m, opt.v = m.double().cpu().numpy(), v.double().cpu().numpy()
    history = [float(x) for x in torch.stack(hist).cpu().numpy()] if hist else []
    return theta.cpu().numpy(), history
