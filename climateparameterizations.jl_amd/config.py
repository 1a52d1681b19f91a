"""Host-side description of one NDE column problem.

Mirrors the bundle of NamedTuples the reference builds in
`prepare_parameters_NDE_training` (wind_mixing/src/NDE_training.jl:1-44:
`constants`, `scalings`, `conditions`, `NN_sizes`) and the parameter tail of the
free-convection NDEs (free_convection/src/free_convection_nde.jl:49-62).
Pure numpy/ctypes: no torch, no oracle imports.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field, replace
from typing import Sequence, Tuple

import numpy as np

# model kinds (colnde.h: COLNDE_MODEL_*)
WIND_MIXING = 0
FREE_CONVECTION = 1
CONVECTIVE_ADJUSTMENT_NDE = 2
MODEL_NAMES = {WIND_MIXING: "wind_mixing", FREE_CONVECTION: "free_convection",
               CONVECTIVE_ADJUSTMENT_NDE: "convective_adjustment_nde"}

# activations (colnde.h: COLNDE_ACT_*), NNlib 0.7 names
ACT_IDS = {"identity": 0, "relu": 1, "mish": 2, "swish": 3, "tanh": 4, "leakyrelu": 5}
ACT_NAMES = {v: k for k, v in ACT_IDS.items()}

COLNDE_MAX_LAYERS = 8
STEPPER_IDS = {"rk4": 0, "rkc2": 1}      # colnde.h: COLNDE_STEPPER_*
# colnde.h: COLNDE_MATRIX_* — how the Float32 Dense products run on the matrix pipe (both are f32 arithmetic): the exact three-way bf16
# operand split with f32 accumulation where a split kernel exists (default), or f32 MFMA throughout
MATRIX_ARITHMETIC_IDS = {"bf16x3_exact": 0, "f32_mfma": 1}
MATRIX_ARITHMETIC_NAMES = {v: k for k, v in MATRIX_ARITHMETIC_IDS.items()}


@dataclass(frozen=True)
class ZeroMeanUnitVarianceScaling:
    """`ZeroMeanUnitVarianceScaling{T}(μ, σ)` — src/DataWrangling/feature_scaling.jl:7-23."""
    mu: float = 0.0
    sigma: float = 1.0

    @classmethod
    def fit(cls, data) -> "ZeroMeanUnitVarianceScaling":
        d = np.asarray(data, dtype=np.float64)
        # Julia `std` is the corrected (n-1) sample standard deviation (feature_scaling.jl:18)
        return cls(float(d.mean()), float(d.std(ddof=1)))

    def scale(self, x):
        return (np.asarray(x) - self.mu) / self.sigma

    def unscale(self, y):
        return self.sigma * np.asarray(y) + self.mu

    __call__ = scale

    def inv(self):
        return self.unscale


@dataclass(frozen=True)
class NDEConfig:
    """Everything the column kernels need besides weights, x0, BCs and truth."""
    model: int = WIND_MIXING
    Nz: int = 32
    # one Flux `Chain(Dense…)`: (in, h1, ..., out); wind mixing uses three identical nets
    layer_sizes: Tuple[int, ...] = (96, 50, 20, 31)
    activations: Tuple[str, ...] = ("mish", "mish", "identity")
    # `conditions` NamedTuple (NDE_training.jl:205-207)
    modified_pacanowski_philander: bool = True
    convective_adjustment: bool = False
    zero_weights: bool = True
    smooth_NN: bool = False
    smooth_Ri: bool = False
    diurnal: bool = False
    train_gradient: bool = True
    # in-place `NDE!` arithmetic (training_postprocessing.jl:105-153): no ϵ in Ri,
    # ν_T switch on ∂u∂z, diurnal top flux not offset by scaling(0)
    inplace_variant: bool = False
    # `constants` (NDE_training.jl:23-33); defaults of test_nonmutating_NDE.jl:52
    H: float = 256.0
    tau: float = 172800.0
    f: float = 1e-4
    g: float = 9.81
    alpha: float = 1.67e-4
    nu0: float = 1e-4
    nu_minus: float = 1e-1
    Ric: float = 0.25
    dRi: float = 1.0
    Pr: float = 1.0
    kappa: float = 10.0
    eps: float = 1e-7
    # scalings u, v, T, uw, vw, wT (μ, σ)
    mu: Tuple[float, ...] = (0.0, 0.0, 19.5, -1e-4, -1e-4, -5e-6)
    sigma: Tuple[float, ...] = (0.05, 0.05, 0.3, 2e-4, 2e-4, 1e-5)
    # convective-adjustment NDE diffusivity (convective_adjustment_nde.jl:43: `10 * ∂T∂z`)
    ca_K: float = 10.0
    # time axis: nondimensional save times (t_train ./ τ, NDE_training.jl:235) and
    # classical-RK4 sub-steps per save interval
    save_times: Tuple[float, ...] = (0.0, 1.0)
    substeps: int = 2
    # time stepper: "rk4" (classical, `substeps` per save interval) or "rkc2" (stabilised second-order Runge-Kutta-Chebyshev for the
    # stiff variants, `substeps` steps of `rkc_stages` stages each; 0 = the least stage count whose stability interval covers the
    # stiffest diffusive mode, colnde_rkc_stages)
    stepper: str = "rk4"
    rkc_stages: int = 0
    # the tolerance the reference hands its adaptive integrator (reltol=1f-3: NDE_training.jl:291; 1e-4: free_convection/src/solve.jl:4): used when
    # substeps = 0 (the handle chooses the sub-step count in its first solve call) and by ColumnNDE.choose_substeps; see colnde_error_estimate
    reltol: float = 1e-3
    # oracle-only diagnostic (tests/test_oracle.py::test_rkc2_switch_pullback): pull the convective-adjustment switch back stage
    # by stage — the exact discrete adjoint of the RKC recurrence, unbounded on switching right-hand sides; the product never does
    rkc_exact_switch_pullback: bool = False

    # ---- derived -------------------------------------------------------
    @property
    def n_nets(self) -> int:
        return 3 if self.model == WIND_MIXING else 1

    @property
    def n_state(self) -> int:
        return 3 * self.Nz if self.model == WIND_MIXING else self.Nz

    @property
    def n_bc(self) -> int:
        return 6 if self.model == WIND_MIXING else 2

    @property
    def n_layers(self) -> int:
        return len(self.layer_sizes) - 1

    @property
    def net_size(self) -> int:
        s = self.layer_sizes
        return sum(s[i] * s[i + 1] + s[i + 1] for i in range(len(s) - 1))

    @property
    def n_params(self) -> int:
        return self.n_nets * self.net_size

    @property
    def n_save(self) -> int:
        return len(self.save_times)

    @property
    def n_steps(self) -> int:
        return (self.n_save - 1) * self.substeps

    def validate(self) -> None:
        s = self.layer_sizes
        if self.n_layers < 1 or self.n_layers > COLNDE_MAX_LAYERS:
            raise ValueError("1..%d dense layers supported" % COLNDE_MAX_LAYERS)
        if len(self.activations) != self.n_layers:
            raise ValueError("one activation per dense layer")
        if s[0] != self.n_state:
            raise ValueError("first layer input %d != state size %d" % (s[0], self.n_state))
        if s[-1] != self.Nz - 1:
            raise ValueError("last layer output %d != Nz-1 = %d interior faces" % (s[-1], self.Nz - 1))
        for a in self.activations:
            if a not in ACT_IDS:
                raise ValueError("unknown activation %r" % (a,))
        if self.model == WIND_MIXING:
            if self.modified_pacanowski_philander and self.convective_adjustment and not self.inplace_variant:
                # `@assert !modified_pacanowski_philander || !convective_adjustment` NDE_training.jl:171; the in-place `NDE!`
                # hard-wires MPP and reads `conditions.convective_adjustment` for its ν_T switch
                # (training_postprocessing.jl:118-121), so the pair is legal there
                raise ValueError("modified_pacanowski_philander and convective_adjustment are exclusive")
            if self.zero_weights and not self.modified_pacanowski_philander:
                # NDE_training.jl:192-194
                raise ValueError("zero_weights requires modified_pacanowski_philander")
        if self.stepper not in ("rk4", "rkc2"):
            raise ValueError("stepper must be 'rk4' or 'rkc2'")
        if self.rkc_stages != 0 and not (2 <= self.rkc_stages <= 256):
            raise ValueError("rkc_stages must be 0 (automatic) or 2..256")
        if self.n_save < 2 or self.substeps < 0:
            raise ValueError("need >= 2 save times and >= 1 substep (0: chosen from reltol)")
        if not (0.0 <= self.reltol < 1.0):
            raise ValueError("0 <= reltol < 1")
        if self.Nz < 4 or self.Nz > 128:
            raise ValueError("4 <= Nz <= 128")

    def with_(self, **kw) -> "NDEConfig":
        return replace(self, **kw)


class CConfig(ctypes.Structure):
    """ctypes twin of `colnde_config` (include/colnde.h)."""
    _fields_ = [
        ("model", ctypes.c_int32),
        ("Nz", ctypes.c_int32),
        ("n_layers", ctypes.c_int32),
        ("layer_sizes", ctypes.c_int32 * (COLNDE_MAX_LAYERS + 1)),
        ("activations", ctypes.c_int32 * COLNDE_MAX_LAYERS),
        ("modified_pacanowski_philander", ctypes.c_int32),
        ("convective_adjustment", ctypes.c_int32),
        ("zero_weights", ctypes.c_int32),
        ("smooth_NN", ctypes.c_int32),
        ("smooth_Ri", ctypes.c_int32),
        ("diurnal", ctypes.c_int32),
        ("train_gradient", ctypes.c_int32),
        ("inplace_variant", ctypes.c_int32),
        ("H", ctypes.c_float), ("tau", ctypes.c_float), ("f", ctypes.c_float),
        ("g", ctypes.c_float), ("alpha", ctypes.c_float), ("nu0", ctypes.c_float),
        ("nu_minus", ctypes.c_float), ("Ric", ctypes.c_float), ("dRi", ctypes.c_float),
        ("Pr", ctypes.c_float), ("kappa", ctypes.c_float), ("eps", ctypes.c_float),
        ("mu", ctypes.c_float * 6),
        ("sigma", ctypes.c_float * 6),
        ("ca_K", ctypes.c_float),
        ("n_save", ctypes.c_int32),
        ("substeps", ctypes.c_int32),
        ("save_times", ctypes.POINTER(ctypes.c_float)),
        ("n_columns", ctypes.c_int32),
        ("device", ctypes.c_int32),
        ("engine", ctypes.c_int32),
        ("stepper", ctypes.c_int32),
        ("rkc_stages", ctypes.c_int32),
        ("matrix_arithmetic", ctypes.c_int32),
        ("reltol", ctypes.c_float),
    ]


def matrix_arithmetic_id(ma) -> int:
    if isinstance(ma, str):
        if ma not in MATRIX_ARITHMETIC_IDS:
            raise ValueError("matrix_arithmetic must be one of %s" % sorted(MATRIX_ARITHMETIC_IDS))
        return MATRIX_ARITHMETIC_IDS[ma]
    if int(ma) not in MATRIX_ARITHMETIC_NAMES:
        raise ValueError("matrix_arithmetic must be one of %s" % sorted(MATRIX_ARITHMETIC_IDS))
    return int(ma)


def to_c_config(cfg: NDEConfig, n_columns: int, device: int = 0, engine: int = 0, matrix_arithmetic=0):
    """Build a `colnde_config`; returns (struct, keepalive) — keep both alive during the call."""
    cfg.validate()
    c = CConfig()
    c.model, c.Nz, c.n_layers = cfg.model, cfg.Nz, cfg.n_layers
    for i, s in enumerate(cfg.layer_sizes):
        c.layer_sizes[i] = s
    for i, a in enumerate(cfg.activations):
        c.activations[i] = ACT_IDS[a]
    for name in ("modified_pacanowski_philander", "convective_adjustment", "zero_weights",
                 "smooth_NN", "smooth_Ri", "diurnal", "train_gradient", "inplace_variant"):
        setattr(c, name, int(bool(getattr(cfg, name))))
    for name in ("H", "tau", "f", "g", "alpha", "nu0", "nu_minus", "Ric", "dRi", "Pr", "kappa", "eps", "ca_K"):
        setattr(c, name, float(getattr(cfg, name)))
    for i in range(6):
        c.mu[i] = cfg.mu[i]
        c.sigma[i] = cfg.sigma[i]
    times = np.ascontiguousarray(cfg.save_times, dtype=np.float32)
    c.n_save, c.substeps = cfg.n_save, cfg.substeps
    c.save_times = times.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    c.n_columns, c.device, c.engine = int(n_columns), int(device), int(engine)
    c.stepper, c.rkc_stages = STEPPER_IDS[cfg.stepper], int(cfg.rkc_stages)
    c.matrix_arithmetic = matrix_arithmetic_id(matrix_arithmetic)
    c.reltol = float(cfg.reltol)
    return c, times
