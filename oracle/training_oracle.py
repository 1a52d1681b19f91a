"""TEST INFRASTRUCTURE — the reference's optimiser loops in float64 over the float64 oracle (parity unpinned, like the oracle itself).

Only tests/ may import this.  It restates, on top of `oracle.nde_oracle.loss_and_grad`:

  * `train_NDE`'s loop (wind_mixing/src/NDE_training.jl:340-372): for each optimiser, for each epoch,
    `res = solve(prob_loss, opt, cb=cb, maxiters=maxiters); weights .= res.minimizer` — GalacticOptim 1.2.0's Flux-optimiser `__solve`
    with `save_best = true` (third-party, pinned in wind_mixing/Manifest.toml, absent from /root/reference; restated from its published
    source): θ = copy(u0) (so Flux's ADAM, whose state lives in an IdDict keyed by the array, starts every solve from zero
    moments and βᵗ = β); per iteration {x = loss(θ), g = Zygote gradient, update!(opt, θ, g), if x < min_err: min_θ = copy(θ) (AFTER the
    update)}; at i = maxiters θ reverts to min_θ.
  * `determine_loss_scalings` with `training_fractions` (NDE_training.jl:256-288, wind_mixing/src/loss.jl:11-31): the scalings are computed
    once from the unscaled losses at the initial weights.
  * `train_neural_differential_equation!` (free_convection/src/training.jl:44-74): `Flux.train!(nde_loss, Flux.params(NN),
    repeated((), epochs), opt)` — one ADAM step per epoch on a persistent state.

Flux 0.11.6 ADAM: m ← β₁m + (1−β₁)g; v ← β₂v + (1−β₂)g²; Δ = η·m/(1−β₁ᵗ)/(√(v/(1−β₂ᵗ)) + ϵ); θ ← θ − Δ, ϵ = 1e-8.
"""
import numpy as np

from . import nde_oracle as O


def initial_loss_scalings(cfg, x0, bcs, weights, truth, training_fractions=None, gradient_scaling=5e-3):
    """NDE_training.jl:256-288: `training_fractions === nothing` → (1, 1, 1, γ, γ, γ); else loss.jl:11-31 on the losses at `weights`."""
    if training_fractions is None:
        return O.default_loss_scalings(cfg, gradient_scaling)
    sol = O.solve(cfg, x0, bcs, weights)
    terms = O.loss_terms(cfg, sol, truth)
    return O.calculate_loss_scalings(terms, training_fractions, cfg.train_gradient)


def train_NDE(cfg, x0, bcs, truth, weights, scalings, etas, epochs=1, maxiters=5, beta=(0.9, 0.999), eps=1e-8):
    """Returns (θ float64, [per-iteration dict(total, terms[6], theta_before)]).  `etas`: one learning rate per optimiser of the list the
    reference passes (`train_optimizers[i]`, train_NDE.jl:141)."""
    theta = np.asarray(weights, np.float64).copy()
    sc = np.asarray(scalings, np.float64).copy()
    if not cfg.train_gradient:
        sc[3:] = 0.0
    hist = []
    for eta in etas:
        for _ in range(epochs):
            m, v, bt = np.zeros_like(theta), np.zeros_like(theta), (beta[0], beta[1])      # a fresh IdDict entry per solve
            min_err, min_theta = np.inf, theta.copy()
            for _ in range(maxiters):
                total, terms, g, _ = O.loss_and_grad(cfg, x0, bcs, theta, truth, sc)
                hist.append(dict(total=float(total), terms=np.asarray(terms, np.float64), theta=theta.copy(), grad=g.copy()))
                theta, m, v, bt = O.adam_step(theta, g, m, v, eta, beta, eps, bt)
                if total < min_err:
                    min_err, min_theta = total, theta.copy()
            theta = min_theta.copy()
    return theta, hist


def train_neural_differential_equation(cfg, x0, bcs, truth, weights, eta, epochs, beta=(0.9, 0.999), eps=1e-8, dtype=np.float64):
    """free_convection/src/training.jl:55-71: loss = Flux.mse over the concatenated solutions; `epochs` ADAM steps on one state.
    dtype = np.float32: the solve and its adjoint in float32 (the optimiser stays float64) — the scale of what float32 itself does to a trajectory
    through ConvectiveAdjustmentNDE's switch."""
    theta = np.asarray(weights, np.float64).copy()
    m, v, bt = np.zeros_like(theta), np.zeros_like(theta), (beta[0], beta[1])
    hist = []
    sc = np.array([0, 0, 1.0, 0, 0, 0])
    for _ in range(epochs):
        total, _, g, _ = O.loss_and_grad(cfg, x0, bcs, theta, truth, sc, dtype=dtype)
        hist.append(float(total))
        theta, m, v, bt = O.adam_step(theta, np.asarray(g, np.float64), m, v, eta, beta, eps, bt)
    return theta, hist
