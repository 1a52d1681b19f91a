"""Flux 0.11 conventions the reference's hot path relies on.

* `Flux.destructure(Chain(Dense…))` → flat θ = vcat(vec(W₁), b₁, vec(W₂), b₂, …) with
  `vec` column-major, `W` of shape out×in  (used at wind_mixing/src/NDE_training.jl:11-13,37;
  free_convection/src/free_convection_nde.jl:2).
* `Dense` default init glorot_uniform, zero bias.
* `ADAM(η, (0.9, 0.999))`, ϵ = 1e-8 (wind_mixing/train_NDE.jl:141; training.jl:71).
Pure numpy; used by host code, tests and bench.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def net_size(layer_sizes: Sequence[int]) -> int:
    return sum(layer_sizes[i] * layer_sizes[i + 1] + layer_sizes[i + 1] for i in range(len(layer_sizes) - 1))


def destructure(layers: Sequence[Tuple[np.ndarray, np.ndarray]]) -> np.ndarray:
    """[(W out×in, b out), …] → flat θ in Flux.destructure order (column-major `vec(W)`)."""
    parts = []
    for W, b in layers:
        parts.append(np.asarray(W).reshape(-1, order="F"))
        parts.append(np.asarray(b).reshape(-1))
    return np.concatenate(parts)


def restructure(theta: np.ndarray, layer_sizes: Sequence[int]) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Inverse of `destructure` for one net (`re(θ)` in the reference)."""
    out, o = [], 0
    for i in range(len(layer_sizes) - 1):
        n_in, n_out = layer_sizes[i], layer_sizes[i + 1]
        W = np.asarray(theta[o:o + n_in * n_out]).reshape((n_out, n_in), order="F")
        o += n_in * n_out
        b = np.asarray(theta[o:o + n_out])
        o += n_out
        out.append((W, b))
    if o != len(theta):
        raise ValueError("theta has %d entries, layer sizes need %d" % (len(theta), o))
    return out


def split_nets(theta: np.ndarray, layer_sizes: Sequence[int], n_nets: int):
    """weights = [uw; vw; wT] (NDE_training.jl:19-21,37) → per-net layer lists."""
    n = net_size(layer_sizes)
    if len(theta) != n * n_nets:
        raise ValueError("expected %d parameters, got %d" % (n * n_nets, len(theta)))
    return [restructure(theta[k * n:(k + 1) * n], layer_sizes) for k in range(n_nets)]


def glorot_uniform_net(rng: np.random.Generator, layer_sizes: Sequence[int], dtype=np.float32):
    """One `Chain(Dense(in,h,σ)…)` with Flux's default init: W ~ U(-r, r), r = sqrt(6/(in+out)), b = 0."""
    layers = []
    for i in range(len(layer_sizes) - 1):
        n_in, n_out = layer_sizes[i], layer_sizes[i + 1]
        r = np.sqrt(6.0 / (n_in + n_out))
        W = rng.uniform(-r, r, size=(n_out, n_in)).astype(dtype)
        layers.append((W, np.zeros(n_out, dtype=dtype)))
    return layers


class ADAM:
    """Flux.Optimise.ADAM: m←β₁m+(1−β₁)g; v←β₂v+(1−β₂)g²; Δ=η·m/(1−β₁ᵗ)/(√(v/(1−β₂ᵗ))+ϵ)."""

    def __init__(self, eta: float = 1e-3, beta=(0.9, 0.999), eps: float = 1e-8):
        self.eta, self.beta, self.eps = float(eta), (float(beta[0]), float(beta[1])), float(eps)
        self.m = None
        self.v = None
        self.beta_t = [self.beta[0], self.beta[1]]  # Flux keeps the running powers βᵗ

    def reset(self) -> None:
        """Forget the moments and the running powers.  Flux keeps ADAM's state in an IdDict keyed by the parameter ARRAY; every
        GalacticOptim `solve(prob, opt, …)` optimises a fresh `θ = copy(prob.u0)`, so each (optimizer, epoch) solve of
        `train_NDE` (NDE_training.jl:340-372) starts from m = v = 0, βᵗ = β — whereas `Flux.train!` on a persistent
        `Flux.params(NN)` (free_convection/src/training.jl:71) keeps one state across calls."""
        self.m = None
        self.v = None
        self.beta_t = [self.beta[0], self.beta[1]]

    def update(self, theta: np.ndarray, grad: np.ndarray) -> np.ndarray:
        if self.m is None:
            self.m = np.zeros_like(theta, dtype=np.float64)
            self.v = np.zeros_like(theta, dtype=np.float64)
        b1, b2 = self.beta
        self.m = b1 * self.m + (1 - b1) * grad
        self.v = b2 * self.v + (1 - b2) * grad * grad
        delta = self.m / (1 - self.beta_t[0]) / (np.sqrt(self.v / (1 - self.beta_t[1])) + self.eps) * self.eta
        self.beta_t[0] *= b1
        self.beta_t[1] *= b2
        theta -= delta.astype(theta.dtype)
        return theta
