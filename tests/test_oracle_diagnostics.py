"""CPU checks of the oracle's restatements behind the round-4 entry points (test infrastructure, parity unpinned as nde_oracle.py says):
`predict_flux` (wind_mixing/src/NDE_training.jl:83-147), `loss_per_tstep` (wind_mixing/src/loss.jl:44-46) and the Richardson error estimate that
gives `reltol` (NDE_training.jl:291) a meaning for a fixed-step solve."""
import numpy as np
import pytest

from colnde import synthetic
from oracle import nde_oracle as O
from tests.test_oracle import VARIANTS


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_flux_divergence_is_the_rhs(name):
    """predict_NDE (NDE_training.jl:149-165) differences exactly the face vectors predict_flux returns: -(τ/H)(σ_flux/σ_q) Dᶜ F [+ Coriolis]."""
    p = synthetic.wind_mixing_problem(7, n_frames=3, weight_divisor=10.0, **VARIANTS[name])
    c, Nz = p.cfg, p.cfg.Nz
    F = O.predict_flux(c, p.x0, p.bcs, p.weights, 0.02)
    dx = O.rhs(c, p.x0, p.bcs, p.weights, 0.02)
    assert F.shape == (7, 3, Nz + 1)
    sg, mu = c.sigma, c.mu
    u, v = p.x0[:, :Nz].astype(np.float64), p.x0[:, Nz:2 * Nz].astype(np.float64)
    cor = (c.f * c.tau / sg[0] * (sg[1] * v + mu[1]), -c.f * c.tau / sg[1] * (sg[0] * u + mu[0]), 0.0)
    for k in range(3):
        A = c.tau / c.H * sg[3 + k] / sg[k] * Nz
        np.testing.assert_allclose(-A * (F[:, k, 1:] - F[:, k, :-1]) + cor[k], dx[:, k * Nz:(k + 1) * Nz], rtol=1e-12, atol=1e-12)
    if not c.zero_weights and not c.diurnal:           # the boundary faces carry the BCs themselves (:108-112)
        np.testing.assert_array_equal(F[:, 0, 0], p.bcs[:, 0].astype(np.float64))
        np.testing.assert_array_equal(F[:, 2, Nz], p.bcs[:, 5].astype(np.float64))


@pytest.mark.parametrize("ca", [False, True])
def test_free_convection_flux_is_what_solve_nde_reevaluates(ca):
    """free_convection/src/solve.jl:32-46: wT = [bottom; NN(T); top] (- min(0, 10 ∂T/∂z) for ConvectiveAdjustmentNDE)."""
    p = synthetic.free_convection_problem(5, Nz=32, n_save=3, substeps=40 if ca else 2, convective_adjustment=ca, t_end=0.01)
    x0 = p.x0.copy()
    x0[:, 10:16] = x0[:, 10:16][:, ::-1]
    F = O.predict_flux(p.cfg, x0, p.bcs, p.weights)
    assert F.shape == (5, 1, 33)
    dT = O.rhs(p.cfg, x0, p.bcs, p.weights)
    C = p.cfg.sigma[5] / p.cfg.sigma[2] * p.cfg.tau / p.cfg.H
    np.testing.assert_allclose(-C * 32 * (F[:, 0, 1:] - F[:, 0, :-1]), dT, rtol=1e-12, atol=1e-12)
    g = np.zeros((5, 33))
    g[:, 1:32] = (x0[:, 1:].astype(np.float64) - x0[:, :-1]) * 32
    plain = O.predict_flux(p.cfg.with_(model=1), x0, p.bcs, p.weights)
    if ca:
        np.testing.assert_allclose(F, plain - np.minimum(0.0, 10.0 * g)[:, None, :], rtol=1e-12, atol=1e-12)
        assert np.abs(F - plain).max() > 1.0            # the inverted layer switched the adjustment on
    else:
        np.testing.assert_array_equal(F, plain)


def test_loss_per_tstep_averages_to_the_loss_terms():
    """mean over time steps (and simulations) of loss_per_tstep = the unscaled terms of loss_NDE (NDE_training.jl:308-317): `mse` over a matrix is the
    mean of its columns' `mse`."""
    p = synthetic.wind_mixing_problem(6, n_frames=9, weight_divisor=1e2)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth)
    sol = O.solve(p.cfg, p.x0, p.bcs, p.weights)
    lp = O.loss_per_tstep(p.cfg, sol, truth)
    assert lp.shape == (6, 6, 9) and np.all(lp[:, :, 0] == 0.0)          # the solves start from the same state
    np.testing.assert_allclose(lp.mean(axis=(0, 2)), O.loss_terms(p.cfg, sol, truth), rtol=1e-12)


@pytest.mark.parametrize("divisor,expect_S", [(1e2, 2), (1.0, 4)])
def test_error_estimate_tracks_the_true_error_and_a_stiff_net_needs_more_substeps(divisor, expect_S):
    """Richardson's estimate from S and 2S sub-steps against the error measured on a 16x finer solve (same norm): within 25 % here (the test
    of the GPU path allows a factor 2).  Nets 100x larger than the 2-day-suite's initial ones (weights/1 instead of /1e2) make the right-hand side
    stiffer than the closure's diffusion: the diffusive stability bound still says 2 sub-steps (colnde_min_substeps), reltol = 1e-3 needs 4."""
    p = synthetic.wind_mixing_problem(8, n_frames=9, weight_divisor=divisor)
    fine = O.solve(p.cfg.with_(substeps=64), p.x0, p.bcs, p.weights)
    chosen = None
    for S in (2, 4, 8):
        cfg = p.cfg.with_(substeps=S)
        est = O.error_estimate(cfg, p.x0, p.bcs, p.weights)
        q = (O.solve(cfg, p.x0, p.bcs, p.weights) - fine) / (1e-3 + np.abs(fine))
        true = np.max(np.sqrt(np.mean(q * q, axis=-1)))
        assert 0.75 * true < est < 1.25 * true, (S, est, true)
        if chosen is None and est <= 1e-3:
            chosen = S
    assert chosen == expect_S
