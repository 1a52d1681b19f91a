"""The two matrix arithmetics of include/colnde.h (COLNDE_MATRIX_BF16X3_EXACT, the default: f32 products as six bf16 MFMA products of exact
three-way operand splits with f32 accumulation; COLNDE_MATRIX_F32_MFMA: v_mfma_f32_* throughout) on every engine, switched from the test itself
(constructor argument / `set_matrix_arithmetic`; nothing here depends on the environment pytest was started in), and the edges of the split:
operands near FLT_MIN, where the low planes fall into or below the subnormal range, and non-finite operands.

Reference semantics being preserved: the Float32 `Dense` products of `predict_flux` (wind_mixing/src/NDE_training.jl:94-96) and of the
free-convection networks (free_convection/src/free_convection_nde.jl:33)."""
import os

import numpy as np
import pytest

import colnde
from colnde import synthetic
from colnde.nde import ENGINE_FC32, ENGINE_REGTILE, ENGINE_TILE16
from oracle import nde_oracle as O
from tests.test_gpu_parity import _record, _rel, SOL_ATOL, LOSS_RTOL, GRAD_REL, FC_SOL_ATOL, FC_LOSS_RTOL, FC_GRAD_REL

pytestmark = pytest.mark.gpu

MAS = ("bf16x3_exact", "f32_mfma")


def _wm(n, frames, **kw):
    p = synthetic.wind_mixing_problem(n, n_frames=frames, weight_divisor=1e2, **kw)
    return p, np.array([1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3]), (SOL_ATOL, LOSS_RTOL, GRAD_REL)


def _fc(n, Nz, **kw):
    p = synthetic.free_convection_problem(n, Nz=Nz, n_save=9, substeps=2, t_end=0.02, **kw)
    return p, O.default_loss_scalings(p.cfg), (FC_SOL_ATOL, FC_LOSS_RTOL, FC_GRAD_REL)


CASES = {
    # name: (problem factory, engine, kernel families that have a split kernel there: forward / adjoint / dW)
    "regtile": (lambda: _wm(200, 25), ENGINE_REGTILE, (True, True, True)),
    "net_split_auto": (lambda: _wm(40, 33), 0, (True, True, True)),
    "tile16_wind_mixing": (lambda: _wm(40, 9), ENGINE_TILE16, (False, False, True)),
    "tile16_smoothing": (lambda: _wm(21, 9, **{"smooth_NN": True}), 0, (False, False, True)),
    "fc32_32": (lambda: _fc(70, 32), 0, (True, True, True)),          # (16-column tiles at these sizes: the SPLIT instantiations of engine_fc.hip)
    "fc32_64": (lambda: _fc(45, 64), 0, (True, True, True)),
    "tile16_free_convection": (lambda: _fc(33, 32), ENGINE_TILE16, (False, False, True)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_both_matrix_arithmetics_on_every_engine(name):
    """Same inputs, one handle per arithmetic (the constructor argument — what a user of the boundary sets in `colnde_config.matrix_arithmetic`):
    both stand within the engine's stated tolerance of the float64 oracle, the default is as close as f32 MFMA (within 1.5x), they differ from
    each other by float32 round-off only, `colnde_plan` says which kernel families ran on the bf16 pipe, and both are run-to-run bit-identical."""
    make, engine, families = CASES[name]
    p, sc, (sol_atol, loss_rtol, grad_rel) = make()
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    out = {}
    for ma in MAS:
        with colnde.ColumnNDE(p.cfg, p.n_columns, engine=engine, matrix_arithmetic=ma) as nde:
            assert nde.matrix_arithmetic == ma
            nde.set_problem(p.x0, p.bcs, truth)
            s = nde.forward(p.weights)
            t1, terms1, g1 = nde.loss_grad(p.weights, sc)
            t2, _, g2 = nde.loss_grad(p.weights, sc)
            plan = nde.plan()
        assert t1 == t2 and np.array_equal(g1, g2)
        assert plan["matrix_arithmetic"] == ma
        fam = (plan["bf16x3_forward"], plan["bf16x3_adjoint"], plan["bf16x3_dw"])
        if ma == "f32_mfma":
            assert fam == (False, False, False)
        elif families is not None:
            assert fam == families
        else:
            assert plan["engine"] == ENGINE_FC32 and plan["bf16x3_dw"]
        assert np.abs(s - sol).max() < sol_atol
        assert np.isclose(t1, tot, rtol=loss_rtol)
        assert _rel(g1, g) < grad_rel
        out[ma] = (s, t1, g1)
    (ss, ts, gs), (s32, t32, g32) = out[MAS[0]], out[MAS[1]]
    d_sol, d_grad = np.abs(ss - s32).max(), _rel(gs, g32.astype(np.float64))
    _record("both_arithmetics/" + name, sol_split_vs_f32=d_sol, grad_split_vs_f32=d_grad, grad_split_vs_oracle=_rel(gs, g), grad_f32_vs_oracle=_rel(g32, g),
            sol_split_vs_oracle=np.abs(ss - sol).max(), sol_f32_vs_oracle=np.abs(s32 - sol).max())
    assert d_grad > 0.0                                                  # the switch did select other kernels
    assert d_sol < 0.25 * sol_atol and d_grad < 0.25 * grad_rel
    assert _rel(gs, g) < max(1.5 * _rel(g32, g), 0.1 * grad_rel)
    assert np.abs(ss - sol).max() < max(1.5 * np.abs(s32 - sol).max(), 0.1 * sol_atol)


def test_config_field_is_validated_and_switchable():
    p = synthetic.wind_mixing_problem(8, n_frames=3)
    with pytest.raises(ValueError):
        colnde.ColumnNDE(p.cfg, 8, matrix_arithmetic="bf16")
    from colnde.config import to_c_config
    import ctypes
    from colnde import _lib
    c, keep = to_c_config(p.cfg, 8)
    c.matrix_arithmetic = 7
    h = ctypes.c_void_p()
    assert _lib.lib().colnde_create(ctypes.byref(c), ctypes.byref(h)) != 0
    assert b"matrix_arithmetic" in _lib.lib().colnde_last_error()
    with colnde.ColumnNDE(p.cfg, 8) as nde:
        assert nde.matrix_arithmetic == "bf16x3_exact"                   # the default of a zero-initialised colnde_config field
        nde.set_matrix_arithmetic("f32_mfma")
        assert nde.matrix_arithmetic == "f32_mfma" and not nde.plan()["bf16x3_forward"]
        with pytest.raises(colnde.ColndeError, match="matrix_arithmetic"):
            _lib.check(nde._L.colnde_set_matrix_arithmetic(nde._h, 5))


def test_describe_spells_out_the_plan_and_lists_the_environment_switches_in_force(monkeypatch):
    """colnde_describe: the tuning switches are environment variables read at creation / planning — this is where a caller of the ABI sees them."""
    p = synthetic.wind_mixing_problem(40, n_frames=3)
    for k in list(os.environ):
        if k.startswith("COLNDE_"):
            monkeypatch.delenv(k)
    with colnde.ColumnNDE(p.cfg, 40, engine=ENGINE_REGTILE) as nde:
        nde.set_problem(p.x0, p.bcs)
        nde.set_problem(p.x0, p.bcs, nde.forward(p.weights_truth))
        nde.loss_grad(p.weights, [1, 1, 1, 0, 0, 0])
        d = nde.describe()
        assert "engine=regtile" in d and "matrix_arithmetic=bf16x3_exact" in d and "adjoint=bf16x3" in d and "z1_tape=1" in d and "block=" in d
        assert d.endswith("| env (none set)") and "exact discrete adjoint" in d
        # the truncating form: capacity smaller than the line
        import ctypes
        buf = ctypes.create_string_buffer(16)
        need = nde._L.colnde_describe(nde._h, buf, 16)
        assert need == len(d) + 1 and buf.value.decode() == d[:15]
    monkeypatch.setenv("COLNDE_RT_ZTAPE", "0")
    monkeypatch.setenv("COLNDE_ADJ_SPLIT", "0")
    with colnde.ColumnNDE(p.cfg, 40, engine=ENGINE_REGTILE) as nde:
        nde.set_problem(p.x0, p.bcs)
        nde.set_problem(p.x0, p.bcs, nde.forward(p.weights_truth))
        nde.loss_grad(p.weights, [1, 1, 1, 0, 0, 0])
        d = nde.describe()
        assert "z1_tape=0" in d and "adjoint=f32" in d and "forward=bf16x3" in d
        assert "COLNDE_RT_ZTAPE=0" in d.split("| env")[1] and "COLNDE_ADJ_SPLIT=0" in d.split("| env")[1]
    p = synthetic.free_convection_problem(33, Nz=32, n_save=3, convective_adjustment=True)
    with colnde.ColumnNDE(p.cfg.with_(stepper="rkc2", substeps=1), 33) as nde:
        nde.set_problem(p.x0, p.bcs)
        nde.set_problem(p.x0, p.bcs, nde.forward(p.weights_truth))
        nde.loss_grad(p.weights, [0, 0, 1, 0, 0, 0])
        d = nde.describe()
        assert "engine=fc32" in d and "stepper=rkc2" in d and "rkc_stages=" in d and "approximate" in d and "tile_width=" in d


@pytest.mark.parametrize("engine", [ENGINE_REGTILE, 0])
@pytest.mark.parametrize("log2_scale", [-100, -108, -118])
def test_split_with_operands_near_flt_min(engine, log2_scale):
    """Weights scaled to 2^log2_scale x O(0.1): below 2^-110 (= 2^16 FLT_MIN) the low plane of the three-way split is a subnormal float, below
    2^-118 the middle plane too.  What the hardware does with them (tools/probe/split_edge.hip -> profiles/r04_split_edge_probe.txt, measured):
    v_mfma_f32_32x32x16_bf16 honours subnormal bf16 INPUTS and keeps subnormal f32 RESULTS (nothing is flushed; a product below 2^-149 is 0), and
    the planes are formed by VALU subtractions that keep float32 subnormals — but a plane is the TOP HALF of a float word, so whatever part of an
    operand lies below 2^-133, the smallest bf16 subnormal, is dropped: the split represents x to within 2^-133 ≈ 9e-41 ABSOLUTE, exactly for
    |x| >= 2^-110 and with relative error 2^-133 / |x| below that (1.6e-6 at 2^-116, 1.6e-4 at 2^-124, 2^-7 at FLT_MIN).  f32 MFMA keeps full
    relative precision down to FLT_MIN.  Consequence tested here on whole solves: with every net operand that small the nets' outputs (~1e-31 …
    1e-36) vanish against the O(1) tendencies, so both arithmetics must give the same finite trajectory, loss and — for the blocks whose scale
    is set by O(1) factors (b3: column sums of the flux cotangent) — the same gradient as the float64 oracle; the remaining blocks are products
    with factors ~1e-33 and are compared between the two arithmetics with an absolute floor at the float32 subnormal scale."""
    p = synthetic.wind_mixing_problem(70 if engine == ENGINE_REGTILE else 24, n_frames=9, weight_divisor=1e2)
    w = (p.weights.astype(np.float64) * 2.0 ** log2_scale).astype(np.float32)
    assert 0 < np.abs(w[w != 0]).max() < 2.0 ** (log2_scale + 1) and np.abs(w[w != 0]).max() > 1e-40
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)          # a truth from O(1e-2) nets: the loss is not zero
    sc = np.array([1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3])
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, w, truth, sc)
    out = {}
    for ma in MAS:
        with colnde.ColumnNDE(p.cfg, p.n_columns, engine=engine, matrix_arithmetic=ma) as nde:
            nde.set_problem(p.x0, p.bcs, truth)
            out[ma] = (nde.forward(w),) + tuple(nde.loss_grad(w, sc))
    (ss, ts, _, gs), (s32, t32, _, g32) = out[MAS[0]], out[MAS[1]]
    assert np.isfinite(ss).all() and np.isfinite(gs).all() and np.isfinite(g32).all()
    assert np.abs(ss - sol).max() < SOL_ATOL and np.abs(s32 - sol).max() < SOL_ATOL
    assert np.abs(ss - s32).max() < 1e-6
    assert np.isclose(ts, tot, rtol=LOSS_RTOL) and np.isclose(ts, t32, rtol=1e-6)
    net = p.cfg.n_params // 3
    b3 = np.zeros(p.cfg.n_params, bool)
    for n in range(3):
        b3[(n + 1) * net - 31:(n + 1) * net] = True                       # Flux.destructure: b3 closes each net's block
    assert _rel(gs[b3], g[b3]) < GRAD_REL and _rel(g32[b3], g[b3]) < GRAD_REL
    floor = 64 * 2.0 ** -149                                             # a few float32 subnormal units: sums of flushed / denormal products
    assert np.all(np.abs(gs[~b3] - g32[~b3]) <= 1e-5 * np.abs(g32[~b3]).max() + floor)
    _record("near_flt_min/%d/%d" % (engine, log2_scale), sol_split_vs_f32=np.abs(ss - s32).max(), b3_split_vs_oracle=_rel(gs[b3], g[b3]),
            rest_abs_split_vs_f32=np.abs(gs[~b3] - g32[~b3]).max(), rest_max=np.abs(g32[~b3]).max())


@pytest.mark.parametrize("ma", MAS)
@pytest.mark.parametrize("engine", [ENGINE_REGTILE, 0])
@pytest.mark.parametrize("where", ["x0_nan", "x0_inf", "weights_inf", "weights_nan", "bcs_neg_inf"])
def test_non_finite_operands_are_reported_under_both_arithmetics(where, engine, ma):
    """±Inf / NaN anywhere in the operands: the split turns an Inf operand into NaN planes (Inf − Inf), f32 MFMA propagates Inf·0 = NaN —
    either way the loss is not finite and the host entry points say so ("not finite") instead of returning rc = 0 with poisoned numbers."""
    p = synthetic.wind_mixing_problem(70 if engine == ENGINE_REGTILE else 9, n_frames=5, weight_divisor=1e2)
    x0, w, bcs = p.x0.copy(), p.weights.copy(), p.bcs.copy()
    if where == "x0_nan":
        x0[3, 40] = np.nan
    elif where == "x0_inf":
        x0[5, 70] = np.inf
    elif where == "weights_inf":
        w[1234] = np.inf
    elif where == "weights_nan":
        w[p.cfg.n_params // 3 + 77] = np.nan
    else:
        bcs[2, 5] = -np.inf
    with colnde.ColumnNDE(p.cfg, p.n_columns, engine=engine, matrix_arithmetic=ma) as nde:
        nde.set_problem(x0, bcs, np.zeros((p.n_columns, 5, 96), np.float32))
        with pytest.raises(colnde.ColndeError, match="not finite"):
            nde.loss(w, [1] * 6)
        with pytest.raises(colnde.ColndeError, match="not finite"):
            nde.loss_grad(w, [1] * 6)
        # the handle survives: a clean problem afterwards gives a finite answer
        nde.set_problem(p.x0, p.bcs, np.zeros((p.n_columns, 5, 96), np.float32))
        tot, _, grad = nde.loss_grad(p.weights, [1] * 6)
        assert np.isfinite(tot) and np.isfinite(grad).all()
