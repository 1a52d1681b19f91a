"""oracle/colnde_ref.c (float32, what bench.py times as cpu_baseline) against oracle/nde_oracle.py (float64)."""
import numpy as np
import pytest

import colnde
from colnde import synthetic
from oracle import nde_oracle as O
from oracle import cref
from tests.test_oracle import VARIANTS, INPLACE_VARIANTS


def _rel(a, b):
    return np.linalg.norm(np.asarray(a, np.float64) - b) / (np.linalg.norm(b) + 1e-300)


@pytest.mark.parametrize("name", sorted(VARIANTS) + sorted(INPLACE_VARIANTS))
def test_c_rhs_matches_numpy(name):
    kw = dict(VARIANTS[name] if name in VARIANTS else INPLACE_VARIANTS[name])
    p = synthetic.wind_mixing_problem(5, n_frames=3, weight_divisor=10.0, **kw)
    ref = O.rhs(p.cfg, p.x0, p.bcs, p.weights, 0.02)
    got = cref.rhs(p.cfg, p.x0, p.bcs, p.weights, 0.02)
    assert _rel(got, ref) < 2e-5


@pytest.mark.parametrize("name", sorted(INPLACE_VARIANTS))
def test_c_forward_inplace_matches_numpy(name):
    """`solve_NDE_mutating` (training_postprocessing.jl:55-159): a forward solve under the in-place arithmetic; κ = 0.1 keeps
    the ν_T = κ faces inside RK4's stability region at two sub-steps per frame."""
    p = synthetic.wind_mixing_problem(4, n_frames=9, weight_divisor=1e2, kappa=0.1, **INPLACE_VARIANTS[name])
    sol = O.solve(p.cfg, p.x0, p.bcs, p.weights)
    got = cref.forward(p.cfg, p.x0, p.bcs, p.weights)
    assert np.abs(got - sol).max() < 5e-5


@pytest.mark.parametrize("ca", [False, True])
def test_c_free_convection_matches_numpy(ca):
    p = synthetic.free_convection_problem(4, Nz=32, n_save=5, substeps=20 if ca else 2, convective_adjustment=ca, t_end=0.01)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth)
    sc = O.default_loss_scalings(p.cfg)
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    tot_c, terms_c, g_c, sol_c = cref.loss_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc, want_sol=True)
    assert np.abs(sol_c - sol).max() < 1e-4
    assert np.isclose(tot_c, tot, rtol=1e-3)
    assert _rel(g_c, g) < 2e-3


@pytest.mark.parametrize("name", ["mpp_zero_weights", "mpp_bc_faces", "smooth_NN", "smooth_Ri", "diurnal", "conv_adj_branch"])
def test_c_loss_grad_matches_numpy(name):
    p = synthetic.wind_mixing_problem(4, n_frames=9, weight_divisor=1e2, **VARIANTS[name])
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth)
    sc = np.array([1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3])
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    tot_c, terms_c, g_c, sol_c = cref.loss_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc, want_sol=True)
    assert np.abs(sol_c - sol).max() < 5e-5
    np.testing.assert_allclose(terms_c, terms, rtol=2e-3, atol=1e-12)
    assert np.isclose(tot_c, tot, rtol=1e-3)
    assert _rel(g_c, g) < 2e-3
    # loss-only path and forward path agree with the loss+grad path
    tot_l, terms_l, _, _ = cref.loss_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc, want_grad=False)
    assert np.isclose(tot_l, tot_c, rtol=1e-6)
    np.testing.assert_array_equal(cref.forward(p.cfg, p.x0, p.bcs, p.weights), sol_c)


def test_c_infer_forcing_matches_numpy():
    cfg, T, top, w = synthetic.inference_problem(8, 4)
    ref = O.infer_forcing(cfg, T, top, w, 1000.0)
    got = cref.infer_forcing(cfg, T, top, w, 1000.0)
    assert _rel(got, ref) < 1e-4


def test_c_sharded_normalisation_sums_to_global():
    p = synthetic.wind_mixing_problem(6, n_frames=5, weight_divisor=1e2)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = O.default_loss_scalings(p.cfg)
    tot, terms, g, _ = cref.loss_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    parts = [cref.loss_grad(p.cfg, p.x0[s], p.bcs[s], p.weights, truth[s], sc, n_col_total=6) for s in (slice(0, 2), slice(2, 6))]
    assert np.isclose(parts[0][0] + parts[1][0], tot, rtol=1e-5)
    assert _rel(parts[0][2] + parts[1][2], g.astype(np.float64)) < 1e-4
