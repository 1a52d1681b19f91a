"""Host-side mirror of the reference's wind-mixing NDE interface (`WindMixing` package), over the HIP engine.

Same names, argument meaning and return shapes as the Julia closures this path replaces:

    NDE(x, p, t)                     wind_mixing/src/NDE_training.jl:56-81      out-of-place RHS, p = [weights; BCs]
    NDE!(dx, x, p, t)  -> NDE_inplace  wind_mixing/src/training_postprocessing.jl:131-153
    solve_NDE_nonmutating / solve_NDE_mutating                                NDE_training.jl:376-406, training_postprocessing.jl:55-159
    loss_NDE(weights, BCs), loss_gradient_NDE(weights, BCs)                   NDE_training.jl:290-323
        -> (total, scaled_losses{u,v,T,∂u∂z,∂v∂z,∂T∂z}, loss_scalings)
    ∇loss  -> grad_loss(weights)      the pullback GalacticOptim obtains from Zygote (NDE_training.jl:327-333)
    calculate_loss_scalings, apply_loss_scalings                               wind_mixing/src/loss.jl:11-42
    train_NDE                                                                  NDE_training.jl:167-374 (optimiser loop :340-372)
    modified_pacanowski_philander!(model, constants, Δt, p, convective_adjustment)   wind_mixing/src/NDE_oceananigans.jl:61-101
        -> modified_pacanowski_philander_step   (implicit diffusion step of the 1-D ocean embedding)

Data loading, JLD2 logging and plotting stay outside (out of scope).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np

from .config import NDEConfig, WIND_MIXING
from .flux_compat import ADAM
from .nde import ColumnNDE

LOSS_KEYS = ("u", "v", "T", "dudz", "dvdz", "dTdz")


def calculate_loss_scalings(losses, fractions, train_gradient: bool):
    """wind_mixing/src/loss.jl:11-31.  `losses` in LOSS_KEYS order; `fractions` = dict(T, dTdz, profile)."""
    Lu, Lv, LT, Lgu, Lgv, LgT = [float(x) for x in losses]
    vel = (1 - fractions["T"]) / fractions["T"] * LT / (Lu + Lv)
    profile_loss = vel * (Lu + Lv) + LT
    if train_gradient:
        velg = (1 - fractions["dTdz"]) / fractions["dTdz"] * LgT / (Lgu + Lgv)
        gradient_loss = velg * (Lgu + Lgv) + LgT
        tot = (1 - fractions["profile"]) / fractions["profile"] * profile_loss / gradient_loss
    else:
        velg = tot = 0.0
    return np.array([vel, vel, 1.0, tot * velg, tot * velg, tot], dtype=np.float64)


def apply_loss_scalings(losses, scalings):
    """wind_mixing/src/loss.jl:33-42."""
    return {k: float(s) * float(l) for k, s, l in zip(LOSS_KEYS, scalings, losses)}


@dataclass
class TrainResult:
    weights: np.ndarray
    history: List[dict]


class WindMixingNDE:
    """The closures `train_NDE` builds (NDE_training.jl:252-323) for one set of simulations (columns)."""

    def __init__(self, cfg: NDEConfig, uvT0, BCs, uvT_trains=None, device: int = 0,
                 gradient_scaling: float = 5e-3, training_fractions: Optional[dict] = None,
                 weights0=None, matrix_arithmetic: str = "bf16x3_exact"):
        if cfg.model != WIND_MIXING:
            raise ValueError("WindMixingNDE needs a wind-mixing config")
        self.cfg = cfg
        uvT0 = np.ascontiguousarray(uvT0, dtype=np.float32)
        self.n_simulations = uvT0.shape[0]
        self.BCs = np.ascontiguousarray(BCs, dtype=np.float32)
        self.engine = ColumnNDE(cfg, self.n_simulations, device=device, matrix_arithmetic=matrix_arithmetic)
        self.engine.set_problem(uvT0, self.BCs, uvT_trains)
        self._rhs_engine = None
        # determine_loss_scalings (NDE_training.jl:256-288)
        if training_fractions is None:
            g = gradient_scaling if cfg.train_gradient else 0.0
            self.loss_scalings = np.array([1, 1, 1, g, g, g], dtype=np.float64)
        else:
            if weights0 is None:
                raise ValueError("training_fractions needs the initial weights (one forward solve, NDE_training.jl:260)")
            _, terms = self.engine.loss(weights0, [1, 1, 1, 1, 1, 1] if cfg.train_gradient else [1, 1, 1, 0, 0, 0])
            self.loss_scalings = calculate_loss_scalings(terms, training_fractions, cfg.train_gradient)

    # ---- RHS closures ---------------------------------------------------------------------------------
    def _split_p(self, p):
        p = np.asarray(p, dtype=np.float32)
        n = self.cfg.n_params
        if p.shape[-1] != n + 6:
            raise ValueError("p must be [weights(%d); uw_b, uw_t, vw_b, vw_t, wT_b, wT_t]" % n)
        return p[..., :n], p[..., n:]

    def NDE(self, x, p, t):
        """Out-of-place `NDE(x, p, t)`; x: [3Nz] (or [n, 3Nz] with p: [n_params+6] shared weights per row of BCs)."""
        w, bc = self._split_p(p)
        x2 = np.atleast_2d(np.asarray(x, dtype=np.float32))
        bc2 = np.broadcast_to(np.atleast_2d(bc), (x2.shape[0], 6))
        dx = self.engine.rhs(x2, w if w.ndim == 1 else w[0], bc2, float(t))
        return dx[0] if np.ndim(x) == 1 else dx

    def NDE_inplace(self, dx, x, p, t):
        """`NDE!(dx, x, p, t)`: the evaluation RHS arithmetic (no ϵ in Ri; training_postprocessing.jl:105-153)."""
        if self._rhs_engine is None:
            self._rhs_engine = ColumnNDE(self.cfg.with_(inplace_variant=True), 1, device=self.engine.device)
        w, bc = self._split_p(p)
        x2 = np.atleast_2d(np.asarray(x, dtype=np.float32))
        bc2 = np.broadcast_to(np.atleast_2d(bc), (x2.shape[0], 6))
        out = self._rhs_engine.rhs(x2, w, bc2, float(t))
        dx[...] = out[0] if np.ndim(x) == 1 else out
        return None

    # ---- solves ---------------------------------------------------------------------------------------
    def solve_NDE_nonmutating(self, weights):
        """`[Array(solve(prob_NDEs[i], …; p=[weights; BCs[i]], saveat=t_train)) for i in 1:n_simulations]`
        (NDE_training.jl:403).  Returns [n_sims, 3Nz, Nt] — each `sols[i]` is the reference's 96×Nt array."""
        return np.transpose(self.engine.forward(weights), (0, 2, 1))

    # ---- losses ---------------------------------------------------------------------------------------
    def loss_NDE(self, weights, BCs=None):
        sc = self.loss_scalings.copy()
        sc[3:] = 0.0                      # loss_NDE zeroes the gradient terms (NDE_training.jl:298)
        total, terms = self.engine.loss(weights, sc)
        return total, dict(zip(LOSS_KEYS, [float(t) for t in terms])), dict(zip(LOSS_KEYS, sc))

    def loss_gradient_NDE(self, weights, BCs=None):
        total, terms = self.engine.loss(weights, self.loss_scalings)
        return total, dict(zip(LOSS_KEYS, [float(t) for t in terms])), dict(zip(LOSS_KEYS, self.loss_scalings))

    def grad_loss(self, weights):
        """∇loss: value and gradient of `first(loss(θ, BCs))` w.r.t. θ — what AutoZygote hands GalacticOptim."""
        sc = self.loss_scalings.copy()
        if not self.cfg.train_gradient:
            sc[3:] = 0.0
        total, terms, grad = self.engine.loss_grad(weights, sc)
        return total, dict(zip(LOSS_KEYS, [float(t) for t in terms])), grad

    def close(self):
        self.engine.close()
        if self._rhs_engine is not None:
            self._rhs_engine.close()


def train_NDE(problem: WindMixingNDE, weights, optimizers: Sequence[ADAM], epochs: int = 1, maxiters: int = 500,
              cb: Optional[Callable] = None, continue_state: bool = False) -> TrainResult:
    """The optimiser loop of `train_NDE` (NDE_training.jl:340-372): for each optimiser and epoch one
    `res = solve(prob_loss, opt, cb=cb, maxiters=maxiters); weights .= res.minimizer`.

    That `solve` is GalacticOptim 1.2.0's Flux-optimiser `__solve` (third-party, pinned in wind_mixing/Manifest.toml, absent from
    /root/reference; restated from its published source, `tests/test_training_loops.py` holds the literal restatement):
      * it optimises `θ = copy(prob.u0)`, and Flux's ADAM keeps its state in an IdDict keyed by that array, so every
        (optimizer, epoch) solve starts from zero moments and βᵗ = β (`continue_state=True` carries them over instead);
      * per iteration: loss and gradient at θ, `cb(θ, total, losses, loss_scalings)` (true = stop), `update!(opt, θ, g)`,
        then `save_best`: if this iteration's loss is the lowest so far, `min_θ = copy(θ)` — taken AFTER the update, i.e. the
        point one ADAM step past the best-loss point; at `i == maxiters` θ reverts to `min_θ` and `cb(min_θ, min_err...)` is called once
        more, so a solve that is not halted issues maxiters + 1 callbacks.
    The reference's OWN callback ignores that extra call: its body is guarded by `if iter <= maxiters` with `iter += 1` behind it
    (NDE_training.jl:343-368), so it prints and `write_data_NDE_training`s exactly `maxiters` records per solve.  A `cb` that logs must carry
    the same guard to produce a reference-shaped log: `reference_logging_callback` below is that closure."""
    theta = np.array(weights, dtype=np.float32)
    history = []
    for opt in optimizers:
        for _ in range(epochs):
            if not continue_state:
                opt.reset()
            best, best_losses, best_theta = np.inf, None, theta.copy()
            halted = False
            for it in range(maxiters):
                total, losses, grad = problem.grad_loss(theta)
                history.append(dict(total=total, **losses))
                if cb is not None and cb(theta, total, losses, problem.loss_scalings):
                    halted = True
                    break
                opt.update(theta, grad.astype(np.float64))
                if total < best:
                    best, best_losses, best_theta = total, losses, theta.copy()
            theta = best_theta
            # `if i == maxiters ... θ = min_θ; cb(θ, x...); break` — GalacticOptim's extra callback on the reverted best point (its return value is
            # ignored).  The reference's cb does nothing on it (`if iter <= maxiters`, NDE_training.jl:344): see reference_logging_callback
            if cb is not None and not halted and maxiters > 0 and best_losses is not None:
                cb(theta, best, best_losses, problem.loss_scalings)
    return TrainResult(theta, history)


def reference_logging_callback(FILE_PATH, cfg: NDEConfig, stage, opt: ADAM, maxiters: int, log: Optional[Callable] = None):
    """The `cb(args...)` closure `train_NDE` builds per (optimizer, epoch) solve (NDE_training.jl:342-368): while `iter <= maxiters` it writes one
    record — losses, loss scalings, the three networks cut out of θ, the optimiser state — through `write_data_NDE_training`
    (wind_mixing/src/data_writing.jl:28-78), then `iter += 1` and returns false.  The guard is what makes the (maxiters + 1)-th call of
    GalacticOptim's solve (the reverted best point) a no-op, so a log holds exactly `maxiters` records per solve and `extract_NN`'s
    `N_data` / arg-min see what they see in a reference log.  `log(iter, total, losses)`: optional hook in place of the reference's `@info`."""
    from .checkpoint import network_record, write_data_NDE_training
    state = {"iter": 1}
    third = cfg.n_params // 3

    def cb(theta, total, losses, loss_scalings):
        if state["iter"] <= maxiters:
            if log is not None:
                log(state["iter"], total, losses)
            th = np.asarray(theta, dtype=np.float32)
            nets = [network_record(th[k * third:(k + 1) * third], cfg.layer_sizes, cfg.activations) for k in range(3)]
            sc = dict(zip(LOSS_KEYS, [float(s) for s in loss_scalings])) if not isinstance(loss_scalings, dict) else loss_scalings
            write_data_NDE_training(FILE_PATH, losses, sc, nets[0], nets[1], nets[2], stage, opt)
        state["iter"] += 1
        return False
    return cb


def _check_replicas(theta, comm, process_group, it):
    from .distributed import weights_in_sync
    if comm is not None:
        red = lambda t: comm.allreduce(t, "max")
    else:
        import torch.distributed as dist
        red = lambda t: dist.all_reduce(t, op=dist.ReduceOp.MAX, group=process_group)
    ok, spread = weights_in_sync(theta, red)
    if not ok:
        raise RuntimeError("train_NDE_device: the ranks' weight vectors differ at iteration %d (checksum spread %.3e): the replicas "
                           "no longer apply identical updates" % (it, spread))


def train_NDE_device(problem: WindMixingNDE, weights, optimizers: Sequence[ADAM], epochs: int = 1, maxiters: int = 500,
                     process_group=None, continue_state: bool = False, comm=None, sync_check_every: int = 0) -> TrainResult:
    """`train_NDE`'s optimiser loop (NDE_training.jl:340-372) with θ, the ADAM state and the best-loss copy resident on the
    GPU: per iteration one `colnde_loss_grad_dev`, [one SUM all-reduce of the gradient buffer when the columns are sharded
    over `process_group`], one fused `colnde_adam_step_dev`; nothing crosses PCIe until the end.  Same update rule, per-solve
    state reset and `save_best` selection as `train_NDE`; the per-iteration callback is not available here.
    sync_check_every = K > 0 (sharded runs): every K iterations, and before the first, the ranks compare checksums of their weight vectors
    with one 12-float MAX all-reduce (`distributed.weights_in_sync`) and raise if the replicas have drifted apart."""
    import torch
    eng = problem.engine
    dev = torch.device("cuda", eng.device)
    n = eng.n_params
    theta = torch.as_tensor(np.asarray(weights, dtype=np.float32)).to(dev).contiguous()
    out = torch.empty(n + 8, dtype=torch.float32, device=dev)
    sc = problem.loss_scalings.copy()
    if not problem.cfg.train_gradient:
        sc[3:] = 0.0
    hist = []
    for opt in optimizers:
        for _ in range(epochs):
            if not continue_state:
                opt.reset()
            m = torch.zeros(n, dtype=torch.float32, device=dev)
            v = torch.zeros(n, dtype=torch.float32, device=dev)
            if opt.m is not None:            # continue_state: the moments of the previous solve
                m.copy_(torch.as_tensor(opt.m, dtype=torch.float32)); v.copy_(torch.as_tensor(opt.v, dtype=torch.float32))
            best = torch.full((), float("inf"), dtype=torch.float32, device=dev)
            best_theta = theta.clone()
            for it in range(maxiters):
                if sync_check_every > 0 and it % sync_check_every == 0 and (comm is not None or process_group is not None):
                    _check_replicas(theta, comm, process_group, it)
                eng.loss_grad(theta, sc, out=out)
                if comm is not None:                       # colnde.distributed.Comm: RCCL behind the C ABI
                    comm.allreduce_result(eng, out)
                elif process_group is not None:
                    import torch.distributed as dist
                    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=process_group)
                total = out[n + 6]
                hist.append(out[n:n + 7].clone())
                eng.adam_step(theta, out, m, v, opt.eta, opt.beta, opt.eps, beta_t=tuple(opt.beta_t))
                opt.beta_t[0] *= opt.beta[0]
                opt.beta_t[1] *= opt.beta[1]
                better = total < best
                best = torch.where(better, total, best)
                best_theta = torch.where(better, theta, best_theta)      # min_θ = copy(θ) AFTER update! (GalacticOptim save_best)
            theta = best_theta.clone()
            opt.m, opt.v = m.double().cpu().numpy(), v.double().cpu().numpy()
    H = torch.stack(hist).cpu().numpy() if hist else np.zeros((0, 7), np.float32)
    history = [dict(total=float(r[6]), **{k: float(r[i]) for i, k in enumerate(LOSS_KEYS)}) for r in H]
    return TrainResult(theta.cpu().numpy(), history)


def train_NN(engine: ColumnNDE, NN_type: str, weights, profiles, BCs, fluxes, optimizers: Sequence[ADAM], train_epochs: Sequence[int],
             gradient_scaling: float = 1e-4, order=None, cb: Optional[Callable] = None):
    """`train_NN(NN, 𝒟train, optimizers, train_epochs, FILE_PATH, NN_type; …)` — wind_mixing/src/NN_training.jl:207-249: for each
    optimiser and epoch one `Flux.train!(NN_loss, Flux.params(NN), training_data, opt)` (ONE ADAM update per shuffled sample) followed by
    `total_loss(training_data)`, which `cb(total_loss, weights)` receives where the reference calls `write_data_NN_training`.
    The whole pass runs on the GPU (`colnde_pretrain_flux_dev`); `profiles` [n, 3Nz] = columns of 𝒟.uvT_scaled, `BCs` [n, 6],
    `fluxes` [n, Nz+1] = columns of 𝒟.<NN_type>.scaled, `order` = the shuffled sample order (default: as given).
    `weights`: the full [uw; vw; wT] vector — only the net being trained changes.  Returns (weights, [total loss per epoch])."""
    import torch
    k = {"uw": 0, "vw": 1, "wT": 2}[NN_type]
    dev = torch.device("cuda", engine.device)
    t = lambda a, dt=np.float32: torch.as_tensor(np.ascontiguousarray(a, dtype=dt)).to(dev)
    theta, X, B, Y = t(weights), t(profiles), t(BCs), t(fluxes)
    od = None if order is None else t(order, np.int32)
    hist = []
    for opt, n_ep in zip(optimizers, train_epochs):
        m, v = torch.zeros_like(theta), torch.zeros_like(theta)
        if opt.m is not None:
            m.copy_(t(opt.m)); v.copy_(t(opt.v))
        for _ in range(n_ep):
            engine.pretrain_flux(k, theta, m, v, X, B, Y, od, gradient_scaling, opt, update=True)
            total = engine.pretrain_flux(k, theta, None, None, X, B, Y, od, gradient_scaling, opt, update=False)
            hist.append(total)
            if cb is not None:
                cb(total, theta)
        opt.m, opt.v = m.double().cpu().numpy(), v.double().cpu().numpy()
    return theta.cpu().numpy(), hist


def modified_pacanowski_philander_step(engine: ColumnNDE, u, v, T, dt: float, dz: float, p: dict, constants, convective_adjustment: bool = False,
                                       halo_bottom=None):
    """`modified_pacanowski_philander!(model, constants, Δt, p, convective_adjustment)` (wind_mixing/src/NDE_oceananigans.jl:61-101;
    diffusivities :17-58) on `interior(u)`, `interior(v)`, `interior(T)` as [n, Nz] (or [Nz]) arrays: returns (u′, v′, T′).
    `p` is the reference's diffusivity dictionary (keys "ν₀", "ν₋", "ΔRi", "Riᶜ", "Pr"; ASCII "nu0", "nu_minus", "dRi", "Ric" accepted),
    `constants` carries α and g (attributes or keys `alpha`/`α`, `g`).  halo_bottom [3, n]: the u, v, T halo cells below the deepest
    cell (None: zero-gradient fill)."""
    def get(d, *names):
        for nm in names:
            if isinstance(d, dict) and nm in d:
                return float(d[nm])
            if not isinstance(d, dict) and hasattr(d, nm):
                return float(getattr(d, nm))
        raise KeyError(names[0])
    params = (get(p, "ν₀", "nu0"), get(p, "ν₋", "nu_minus"), get(p, "ΔRi", "dRi"), get(p, "Riᶜ", "Ric"), get(p, "Pr"),
              get(constants, "α", "alpha"), get(constants, "g"))
    u, v, T = (np.asarray(a, dtype=np.float32) for a in (u, v, T))
    shape = T.shape
    u2, v2, T2 = (a.reshape(-1, shape[-1]) for a in (u, v, T))
    hb = None if halo_bottom is None else np.asarray(halo_bottom, np.float32).reshape(3, -1)
    uo, vo, To = engine.implicit_diffusion(u2, v2, T2, dt, dz, params, convective_adjustment, hb)
    return uo.reshape(shape), vo.reshape(shape), To.reshape(shape)
