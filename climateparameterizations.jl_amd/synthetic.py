"""Synthetic column problems of the shapes BASELINE.json names.

The reference's LES data files and trained networks are git-ignored and absent
(/root/reference/.gitignore:24-25,29,61), so every workload is generated here from a
counter-based RNG (Philox, seed 20261004) — identical arrays for oracle, C port and HIP.
Pure numpy; no oracle imports (bench.py and the tests both use this module).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from .config import (NDEConfig, WIND_MIXING, FREE_CONVECTION, CONVECTIVE_ADJUSTMENT_NDE)
from .flux_compat import glorot_uniform_net, destructure

SEED = 20261004


def _rng(seed: int, stream: int) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed, stream]))


@dataclass
class ColumnProblem:
    cfg: NDEConfig
    x0: np.ndarray          # [n_col, n_state] float32, scaled units
    bcs: np.ndarray         # [n_col, n_bc] float32, scaled fluxes (wind mixing: uw_b,uw_t,vw_b,vw_t,wT_b,wT_t)
    weights: np.ndarray     # [n_params] float32, Flux.destructure order, nets concatenated uw;vw;wT
    weights_truth: np.ndarray  # a perturbed weight set whose trajectory serves as "truth"
    truth: Optional[np.ndarray] = None  # [n_col, n_save, n_state]

    @property
    def n_columns(self) -> int:
        return self.x0.shape[0]


def make_weights(rng: np.random.Generator, cfg: NDEConfig, divisor: float, same_init: bool = True) -> np.ndarray:
    """Glorot-uniform nets, zero bias, divided as in wind_mixing/train_NDE.jl:103-107
    (`re(weights ./ 1f5)`; the three nets start from the same draw there)."""
    nets = []
    base = glorot_uniform_net(rng, cfg.layer_sizes)
    for k in range(cfg.n_nets):
        net = base if (same_init or k == 0) else glorot_uniform_net(rng, cfg.layer_sizes)
        nets.append(destructure(net))
    w = np.concatenate(nets).astype(np.float32)
    return (w / np.float32(divisor)).astype(np.float32)


def perturb_weights(rng: np.random.Generator, w: np.ndarray, rel: float = 0.5) -> np.ndarray:
    scale = np.float32(rel) * np.abs(w).max()
    return (w + scale * rng.standard_normal(w.shape).astype(np.float32)).astype(np.float32)


def wind_mixing_problem(n_columns: int, Nz: int = 32, n_frames: int = 289, substeps: int = 2,
                        frame_stride: int = 1, seed: int = SEED, weight_divisor: float = 1e5,
                        tau: float = 172800.0, n_frames_total: Optional[int] = None,
                        layer_sizes=None, activations=("mish", "mish", "identity"),
                        **cfg_overrides) -> ColumnProblem:
    """`train_NDE` 2-day-suite shape (289 frames, 10-minute spacing ⇒ Δt̂ = 1/288), MPP + zero_weights +
    train_gradient (wind_mixing/train_NDE.jl:114-131), three 96-50-20-31 mish nets (:103)."""
    if layer_sizes is None:
        layer_sizes = (3 * Nz, 50, 20, Nz - 1)
    if n_frames_total is None:
        n_frames_total = 289
    frames = np.arange(n_frames) * frame_stride
    save_times = tuple((frames / float(n_frames_total - 1)).tolist())
    alpha, g = 1.67e-4, 9.81
    sig = (0.05, 0.05, 0.3, 2e-4, 2e-4, 1e-5)
    mu = (0.0, 0.0, 19.5, -sig[3] / 2, -sig[4] / 2, -sig[5] / 2)
    kw = dict(model=WIND_MIXING, Nz=Nz, layer_sizes=tuple(layer_sizes), activations=tuple(activations),
              modified_pacanowski_philander=True, convective_adjustment=False, zero_weights=True,
              train_gradient=True, H=256.0, tau=tau, f=1e-4, g=g, alpha=alpha, nu0=1e-4, nu_minus=1e-1,
              Ric=0.25, dRi=1.0, Pr=1.0, kappa=10.0, mu=mu, sigma=sig, save_times=save_times,
              substeps=substeps)
    kw.update(cfg_overrides)
    cfg = NDEConfig(**kw)
    cfg.validate()

    r = _rng(seed, 1)
    k = np.arange(Nz, dtype=np.float64)
    That = np.linspace(-1.5, 1.5, Nz)[None, :] + 0.02 * r.standard_normal((n_columns, Nz))
    A_u = r.uniform(0.2, 1.0, size=(n_columns, 1))
    A_v = r.uniform(0.2, 1.0, size=(n_columns, 1))
    shear = np.tanh((k - 0.75 * Nz) / (Nz / 8.0))[None, :]
    uhat = A_u * shear + 0.02 * r.standard_normal((n_columns, Nz))
    vhat = A_v * shear + 0.02 * r.standard_normal((n_columns, Nz))
    x0 = np.concatenate([uhat, vhat, That], axis=1).astype(np.float32)

    Q_u = r.uniform(-1e-3, -2e-4, size=n_columns)
    Q_b = r.uniform(-5e-8, 5e-8, size=n_columns)
    bcs = np.zeros((n_columns, 6), dtype=np.float64)
    bcs[:, 0] = (0.0 - mu[3]) / sig[3]          # uw bottom
    bcs[:, 1] = (Q_u - mu[3]) / sig[3]          # uw top
    bcs[:, 2] = (0.0 - mu[4]) / sig[4]          # vw bottom
    bcs[:, 3] = (0.0 - mu[4]) / sig[4]          # vw top
    bcs[:, 4] = (0.0 - mu[5]) / sig[5]          # wT bottom
    if cfg.diurnal:
        # diurnal: p carries 5 BCs and Qᵇ defines wT_top(t) (NDE_training.jl:68-81); slot 6 holds Qᵇ
        bcs[:, 5] = np.abs(Q_b)
    else:
        bcs[:, 5] = (Q_b / (alpha * g) - mu[5]) / sig[5]
    bcs = bcs.astype(np.float32)

    rw = _rng(seed, 2)
    w = make_weights(rw, cfg, weight_divisor)
    w_truth = perturb_weights(rw, w)
    return ColumnProblem(cfg, x0, bcs, w, w_truth)


def free_convection_problem(n_columns: int, Nz: int = 32, n_save: int = 129, substeps: int = 4,
                            convective_adjustment: bool = False, seed: int = SEED,
                            weight_divisor: float = 1e2, t_end: float = 1.0,
                            layer_sizes=None, activations=("relu", "relu", "identity")) -> ColumnProblem:
    """free_convection/train_free_convection_nde.jl:119-121 — Dense(Nz,4Nz,relu)→Dense(4Nz,4Nz,relu)→Dense(4Nz,Nz-1);
    RHS free_convection_nde.jl:29-38 / convective_adjustment_nde.jl:33-48; (σ_wT/σ_T)(τ/H) fixed at 0.5."""
    if layer_sizes is None:
        layer_sizes = (Nz, 4 * Nz, 4 * Nz, Nz - 1)
    H, tau, s_wT = 128.0, 691200.0, 1e-5
    s_T = s_wT * tau / H / 0.5
    mu_wT = 0.0  # scaled zero flux ≡ 0, so an untrained (≈0) net is a quiescent interior
    cfg = NDEConfig(model=CONVECTIVE_ADJUSTMENT_NDE if convective_adjustment else FREE_CONVECTION,
                    Nz=Nz, layer_sizes=tuple(layer_sizes), activations=tuple(activations),
                    modified_pacanowski_philander=False, convective_adjustment=False, zero_weights=False,
                    train_gradient=False, H=H, tau=tau,
                    mu=(0.0, 0.0, 19.5, 0.0, 0.0, mu_wT), sigma=(1.0, 1.0, s_T, 1.0, 1.0, s_wT),
                    save_times=tuple(np.linspace(0.0, t_end, n_save).tolist()), substeps=substeps)
    cfg.validate()
    r = _rng(seed, 3)
    That = np.linspace(-1.5, 1.5, Nz)[None, :] + 0.02 * r.standard_normal((n_columns, Nz))
    x0 = That.astype(np.float32)
    Q = r.uniform(1e-6, 1e-5, size=n_columns)   # surface cooling (wT > 0 at the top face)
    bcs = np.zeros((n_columns, 2), dtype=np.float64)
    bcs[:, 0] = (0.0 - mu_wT) / s_wT
    bcs[:, 1] = (Q - mu_wT) / s_wT
    rw = _rng(seed, 4)
    w = make_weights(rw, cfg, weight_divisor)
    w_truth = perturb_weights(rw, w)
    return ColumnProblem(cfg, x0, bcs.astype(np.float32), w, w_truth)


def inference_problem(nx: int, ny: int, Nz: int = 32, seed: int = SEED):
    """double_gyre_nn.jl:149-168 — a T field of nx×ny columns, a relaxation surface flux, the wT MLP 32-128-128-31."""
    cfg = NDEConfig(model=FREE_CONVECTION, Nz=Nz, layer_sizes=(Nz, 4 * Nz, 4 * Nz, Nz - 1),
                    activations=("relu", "relu", "identity"), modified_pacanowski_philander=False,
                    zero_weights=False, train_gradient=False,
                    mu=(0.0, 0.0, 19.5, 0.0, 0.0, -5e-6), sigma=(1.0, 1.0, 0.3, 1.0, 1.0, 1e-5))
    cfg.validate()
    r = _rng(seed, 5)
    n = nx * ny
    T = (np.linspace(5.0, 25.0, Nz)[None, :] + 0.5 * r.standard_normal((n, Nz))).astype(np.float32)
    top_flux = (1e-5 * r.standard_normal(n)).astype(np.float32)
    w = make_weights(_rng(seed, 6), cfg, 1e2)
    return cfg, T, top_flux, w
