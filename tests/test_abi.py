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
    assert L.colnde_version() == int(re.search(r"#define COLNDE_VERSION (\d+)", header).group(1)) == 106


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


def _header_prototypes():
    text = open(os.path.join(ROOT, "include", "colnde.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(colnde_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return protos


def test_julia_config_mirrors_the_header():
    """julia/ColumnNDE.jl cannot run here (no Julia in the image), so its `Config` struct — the one thing a silent layout drift would
    break — is checked as text: same field names, order, element types and array lengths as `colnde_config` and its ctypes twin."""
    jl = open(os.path.join(ROOT, "julia", "ColumnNDE.jl")).read()
    body = jl[jl.index("Base.@kwdef mutable struct Config"):]
    body = body[:body.index("\nend")]
    body = re.sub(r"#.*", "", body)
    fields = re.findall(r"([A-Za-z_][A-Za-z_0-9]*)::((?:NTuple\{\d+,\s*(?:Int32|Float32)\})|Int32|Float32|Ptr\{Float32\})", body)
    cmap = {ctypes.c_int32: "Int32", ctypes.c_float: "Float32"}
    expect = []
    for name, ct in CConfig._fields_:
        if hasattr(ct, "_length_"):
            expect.append((name, "NTuple{%d,%s}" % (ct._length_, cmap[ct._type_])))
        elif ct in cmap:
            expect.append((name, cmap[ct]))
        else:
            expect.append((name, "Ptr{Float32}"))
    assert [(n, t.replace(" ", "")) for n, t in fields] == expect


def test_julia_ccalls_name_declared_symbols_with_the_right_arity():
    jl = open(os.path.join(ROOT, "julia", "ColumnNDE.jl")).read()
    protos = _header_prototypes()
    calls = []
    for m in re.finditer(r"ccall\(\(:(colnde_[a-z_0-9]+),\s*libcolnde\),\s*\w+,\s*\(", jl):
        i, depth = m.end(), 1                      # the balanced Julia argument-type tuple that follows
        while depth:
            depth += {"(": 1, ")": -1}.get(jl[i], 0)
            i += 1
        calls.append((m.group(1), jl[m.end():i - 1]))
    assert len(calls) >= 20
    for name, argtypes in calls:
        assert name in protos, name
        depth, parts, cur = 0, [], ""
        for ch in argtypes:                       # top-level commas only (Ptr{...}, Ref{...} nest braces)
            depth += {"{": 1, "(": 1, "}": -1, ")": -1}.get(ch, 0)
            if ch == "," and depth == 0:
                parts.append(cur)
                cur = ""
            else:
                cur += ch
        parts.append(cur)
        n = len([q for q in parts if q.strip()])
        assert n == protos[name], (name, n, protos[name])
