"""The reference's WIDE wind-mixing architectures (VERDICT r4 task 6): `train_NDE.jl:101-102` (commented alternatives kept in the script) and
`train_NDE_args.jl:150-166` build 3 x Chain(Dense(96, 400, σ), Dense(400, 400, σ), Dense(400, 31)) and 3 x Chain(Dense(96, 400, σ), Dense(400, 31))
with σ in {swish, mish, leakyrelu, relu, tanh} (`hidden_units = 400`, train_NDE.jl:97-99).  These shapes run on the generic tile16 engine
(regtile and the net-split pair are specialised to 96-50-20-31): parity of the RHS, the solve, the six-term loss and the gradient against the
float64 oracle at 64 columns, under both matrix arithmetics, and the plan bits that say which pipe ran."""
import numpy as np
import pytest

import colnde
from colnde import synthetic
from colnde.nde import ENGINE_TILE16
from oracle import nde_oracle as O

from tests.test_gpu_parity import _record, SOL_ATOL, LOSS_RTOL, GRAD_REL

pytestmark = pytest.mark.gpu

WIDE = {
    "96-400-400-31_swish": dict(layer_sizes=(96, 400, 400, 31), activations=("swish", "swish", "identity")),
    "96-400-31_mish": dict(layer_sizes=(96, 400, 31), activations=("mish", "identity")),
    "96-400-400-31_leakyrelu": dict(layer_sizes=(96, 400, 400, 31), activations=("leakyrelu", "leakyrelu", "identity")),
}


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300)


@pytest.mark.parametrize("ma", ["bf16x3_exact", "f32_mfma"])
@pytest.mark.parametrize("name", sorted(WIDE))
def test_wide_wind_mixing_networks_against_the_oracle(name, ma):
    p = synthetic.wind_mixing_problem(64, n_frames=5, weight_divisor=1e2, **WIDE[name])
    n_hidden = sum(a * b + b for a, b in zip(WIDE[name]["layer_sizes"][:-1], WIDE[name]["layer_sizes"][1:]))
    assert p.cfg.n_params == 3 * n_hidden
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = O.default_loss_scalings(p.cfg)
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    dx = O.rhs(p.cfg, p.x0, p.bcs, p.weights)
    with colnde.ColumnNDE(p.cfg, 64, matrix_arithmetic=ma) as nde:
        assert nde.engine == ENGINE_TILE16
        assert _rel(nde.rhs(p.x0, p.weights, p.bcs, 0.0), dx) < 1e-6
        nde.set_problem(p.x0, p.bcs, truth)
        sol_g = nde.forward(p.weights)
        tot_g, terms_g, grad_g = nde.loss_grad(p.weights, sc)
        plan = nde.plan()
        desc = nde.describe()
    assert ("activation_rows=global_memory" in desc) == (len(WIDE[name]["layer_sizes"]) == 4)       # 400-400: rows in global memory; 400: they fit the LDS
    _record("wide/%s/%s" % (name, ma), sol_abs=np.abs(sol_g - sol).max(), loss_rel=abs(tot_g - tot) / tot, grad_rel=_rel(grad_g, g))
    assert np.abs(sol_g - sol).max() < SOL_ATOL
    assert np.isclose(tot_g, tot, rtol=LOSS_RTOL)
    np.testing.assert_allclose(terms_g, terms, rtol=10 * LOSS_RTOL, atol=0)
    assert _rel(grad_g, g) < GRAD_REL
    # every layer block of every net (Flux.destructure order: W1 b1 W2 b2 ...)
    sizes = WIDE[name]["layer_sizes"]
    off = 0
    for net in range(3):
        for a, b in zip(sizes[:-1], sizes[1:]):
            for n in (a * b, b):
                assert _rel(grad_g[off:off + n], g[off:off + n]) < 4 * GRAD_REL, (net, a, b, n)
                off += n
    assert off == p.cfg.n_params
    # which pipe ran (include/colnde.h says so): tile16's forward and adjoint are f32-MFMA kernels under either arithmetic; the tape GEMM (dW) follows the
    # arithmetic asked for — dw_gemm_split_kernel with the 400 x 400 matrices cut into chunks of output blocks, or the f32 L2-streaming dw_gemm_kernel
    assert plan["matrix_arithmetic"] == ma and plan["engine"] == ENGINE_TILE16 and plan["dw_taped"]
    wide2 = len(WIDE[name]["layer_sizes"]) == 4            # 400-400: rows in global memory, dense chains on the bf16 pipe under the default arithmetic
    assert plan["bf16x3_forward"] == plan["bf16x3_adjoint"] == (wide2 and ma == "bf16x3_exact") and plan["bf16x3_dw"] == (ma == "bf16x3_exact")


def test_wide_networks_other_paths_rkc2_inplace_rhs_and_column_blocks(monkeypatch):
    """The other kernels that keep the activation rows in global memory for 3 x (96-400-400-31): the RKC2 instantiations (the wind-mixing convective-adjustment
    branch, kappa = 10, `ROCK4` in the reference: train_NDE.jl:143) — solve and gradient against the oracle; the in-place `NDE!` arithmetic through `rhs_kernel`
    for MORE columns than the handle was created with (the slab grows); and the gradient path cut into column blocks (COLNDE_T16_BLOCK)."""
    kw = WIDE["96-400-400-31_swish"]
    p = synthetic.wind_mixing_problem(24, n_frames=3, weight_divisor=1e2, modified_pacanowski_philander=False, zero_weights=False, convective_adjustment=True,
                                      kappa=10.0, stepper="rkc2", substeps=1, **kw)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = O.default_loss_scalings(p.cfg)
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(p.cfg, 24) as nde:
        assert "activation_rows=global_memory" in nde.describe() and "rkc2" in nde.describe()
        nde.set_problem(p.x0, p.bcs, truth)
        sol_g = nde.forward(p.weights)
        tot_g, _, grad_g = nde.loss_grad(p.weights, sc)
        assert nde.plan()["approximate_gradient"]                       # one switch pattern per RKC2 step (include/colnde.h), as in the oracle
    _record("wide/rkc2_kappa10", sol_abs=np.abs(sol_g - sol).max(), loss_rel=abs(tot_g - tot) / tot, grad_rel=_rel(grad_g, g))
    assert np.abs(sol_g - sol).max() < 1e-4 and np.isclose(tot_g, tot, rtol=2e-3) and _rel(grad_g, g) < 2e-3

    q = synthetic.wind_mixing_problem(70, n_frames=3, weight_divisor=10.0, inplace_variant=True, **kw)
    with colnde.ColumnNDE(q.cfg, 8) as nde:                             # created for 8 columns, asked for 70: the slab of rows grows
        got = nde.rhs(q.x0, q.weights, q.bcs, 0.0)
    assert _rel(got, O.rhs(q.cfg, q.x0, q.bcs, q.weights)) < 1e-6

    r = synthetic.wind_mixing_problem(80, n_frames=4, weight_divisor=1e2, **kw)
    truth = O.solve(r.cfg, r.x0, r.bcs, r.weights_truth).astype(np.float32)
    tot, terms, g, sol = O.loss_and_grad(r.cfg, r.x0, r.bcs, r.weights, truth, sc)
    monkeypatch.setenv("COLNDE_T16_BLOCK", "32")
    with colnde.ColumnNDE(r.cfg, 80) as nde:
        nde.set_problem(r.x0, r.bcs, truth)
        # the diagnostics of round 4 on this shape: loss without gradient, predict_flux, loss_per_tstep, the Richardson error estimate
        tot_l, terms_l = nde.loss(r.weights, sc)
        assert np.isclose(tot_l, tot, rtol=LOSS_RTOL)
        fl = nde.flux(r.x0[:5], r.weights, r.bcs[:5], 0.0)
        assert _rel(fl, O.predict_flux(r.cfg, r.x0[:5], r.bcs[:5], r.weights)) < 1e-6
        lpt = nde.loss_per_tstep(r.weights)
        np.testing.assert_allclose(lpt.mean(axis=(0, 2)), O.loss_terms(r.cfg, sol, truth), rtol=10 * LOSS_RTOL)
        est = nde.error_estimate(r.weights)
        est64 = O.error_estimate(r.cfg, r.x0, r.bcs, r.weights)
        assert 0.5 * est64 < est < 2.0 * est64 or est < 2e-4                     # (float32's floor in this norm)
        tot_g, terms_g, grad_g = nde.loss_grad(r.weights, sc)
        assert nde.plan()["n_blocks"] == 3
    np.testing.assert_allclose(terms_g, terms, rtol=10 * LOSS_RTOL, atol=0)
    assert _rel(grad_g, g) < GRAD_REL
