"""CPU-side checks of the drop-in boundary: libcolnde.so loads, exports every symbol include/colnde.h declares,
the ctypes struct mirrors the C struct, and — with no GPU — create() fails loudly instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest

import colnde
from colnde import _lib, synthetic
from colnde.config import CConfig, to_c_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "colnde.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(colnde_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 20
    bound = {name for name, _, _ in _lib.SYMBOLS}
    for name in declared:
        assert hasattr(L, name), "%s declared in colnde.h but not exported" % name
        assert name in bound, "%s declared in colnde.h but missing from the ctypes binding" % name
    header = open(os.path.join(ROOT, "include", "colnde.h")).read()
    assert L.colnde_version() == int(re.search(r"#define COLNDE_VERSION (\d+)", header).group(1)) == 101


def test_config_struct_layout_matches_header_order():
    text = open(os.path.join(ROOT, "include", "colnde.h")).read()
    body = text[text.index("typedef struct colnde_config {"):text.index("} colnde_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.replace("typedef struct colnde_config {", "").strip()
        if not decl:
            continue
        decl = re.sub(r"^(const\s+)?(int32_t|float)\s*\*?\s*", "", decl)
        for part in decl.split(","):
            names.append(re.sub(r"\[.*\]", "", part).strip())
    assert names == [f[0] for f in CConfig._fields_]


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(colnde.ColndeError, match="no CPU fallback"):
        _lib.lib()


def test_create_without_gpu_or_with_bad_config_reports_errors():
    import torch
    p = synthetic.wind_mixing_problem(4, n_frames=3)
    L = _lib.lib()
    h = ctypes.c_void_p()
    c, keep = to_c_config(p.cfg, 4)
    c.Nz = 2                                        # invalid on any machine
    assert L.colnde_create(ctypes.byref(c), ctypes.byref(h)) != 0
    assert b"Nz" in L.colnde_last_error()
    if not torch.cuda.is_available():
        with pytest.raises(colnde.ColndeError, match="no HIP device|no CPU fallback"):
            colnde.ColumnNDE(p.cfg, 4)


def test_python_validation_mirrors_reference_asserts():
    base = synthetic.wind_mixing_problem(1, n_frames=2).cfg
    with pytest.raises(ValueError):   # @assert !modified_pacanowski_philander || !convective_adjustment (NDE_training.jl:171)
        base.with_(convective_adjustment=True).validate()
    with pytest.raises(ValueError):   # zero_weights requires MPP (NDE_training.jl:192-194)
        base.with_(modified_pacanowski_philander=False).validate()
    with pytest.raises(ValueError):
        base.with_(layer_sizes=(96, 50, 20, 30)).validate()


def test_adam_matches_flux_update_rule():
    from colnde.flux_compat import ADAM
    th = np.array([1.0, -2.0], dtype=np.float64)
    g = np.array([0.5, -0.25])
    opt = ADAM(1e-2)
    opt.update(th, g)
    # first step of ADAM: Δ = η · g/(|g| + ϵ') ≈ η·sign(g)
    np.testing.assert_allclose(th, [1.0 - 1e-2, -2.0 + 1e-2], rtol=1e-6)


def test_min_substeps_is_the_rk4_diffusive_bound():
    """colnde_min_substeps (no GPU): lambda = 4 D Nz^2, dt = widest save interval / substeps, |lambda dt| <= 2.785."""
    import colnde
    from colnde import synthetic
    p = synthetic.wind_mixing_problem(1, n_frames=3)                         # bench workload: nu = 0.1001, tau = 172800, H = 256
    lam = 4 * 172800 * 0.1001 / 256 ** 2 * 32 ** 2
    assert colnde.min_substeps(p.cfg) == int(np.ceil(lam / 288 / 2.785)) == 2
    assert colnde.min_substeps(p.cfg.with_(nu_minus=0.0, nu0=1e-4)) == 1
    # kappa = 10 (the reference default) in the convective-adjustment branches: the ~130 sub-steps per frame ADVICE r1 quotes
    ca = p.cfg.with_(modified_pacanowski_philander=False, zero_weights=False, convective_adjustment=True)
    assert colnde.min_substeps(ca) == int(np.ceil(4 * 172800 * 10 / 256 ** 2 * 1024 / 288 / 2.785)) == 135
    assert colnde.min_substeps(p.cfg.with_(inplace_variant=True, convective_adjustment=True)) == 135
    # ConvectiveAdjustmentNDE, K = 10, C = 0.5, 64 levels, config 4's axis: 128 intervals over [0, 1]
    fc = synthetic.free_convection_problem(1, Nz=64, n_save=129, convective_adjustment=True).cfg
    assert colnde.min_substeps(fc) == int(np.ceil(4 * 0.5 * 10 * 64 ** 2 / 128 / 2.785)) == 230
    assert colnde.min_substeps(synthetic.free_convection_problem(1, Nz=64, n_save=129).cfg) == 1
    # a non-uniform axis is judged by its widest interval
    assert colnde.min_substeps(p.cfg.with_(save_times=(0.0, 0.001, 0.011))) == int(np.ceil(lam * 0.01 / 2.785))
