"""SURVEY §8f rank 2, second half: `train_NN` flux-MLP pre-training (wind_mixing/src/NN_training.jl:25-169, 207-249;
free_convection/train_free_convection_nde.jl:186-216).  CPU: the oracle's flux closures against the RHS they are pieces of, and its
per-sample gradient against finite differences.  GPU: the one-workgroup pre-training kernel against the oracle's sequential ADAM."""
import numpy as np
import pytest

import colnde
from colnde import synthetic
from colnde.flux_compat import ADAM
from oracle import nde_oracle as O
from tests.test_oracle import VARIANTS


def _data(p, n, seed=3):
    """(profile, BCs, flux) samples as 𝒟train holds them: states along a trajectory and 'true' fluxes from a perturbed weight set."""
    cfg = p.cfg
    sol = O.solve(cfg, p.x0, p.bcs, p.weights_truth)
    X = sol.reshape(-1, cfg.n_state)[:n]
    B = np.repeat(p.bcs, cfg.n_save, axis=0)[:n].astype(np.float64)
    nets = O.unpack(p.weights_truth.astype(np.float64), cfg.layer_sizes, cfg.n_nets)
    rng = np.random.default_rng(seed)
    Y = [O.predict_single_flux(cfg, k, X, B, nets[min(k, cfg.n_nets - 1)]) + 0.05 * rng.standard_normal((n, cfg.Nz + 1)) for k in range(3)]
    return X, B, Y


@pytest.mark.parametrize("name", ["mpp_zero_weights", "mpp_bc_faces", "conv_adj_branch", "raw"])
def test_flux_closures_are_the_pieces_of_the_rhs(name):
    """predict_uw/vw/wT (NN_training.jl:25-169) and predict_flux (NDE_training.jl:94-147) restate the same arithmetic: assembling the
    three single-flux face vectors into tendencies (predict_NDE, :160-162) reproduces the oracle's RHS."""
    p = synthetic.wind_mixing_problem(3, n_frames=3, weight_divisor=10.0, **VARIANTS[name])
    cfg = p.cfg
    nets = O.unpack(p.weights.astype(np.float64), cfg.layer_sizes, 3)
    F = [O.predict_single_flux(cfg, k, p.x0, p.bcs, nets[k]) for k in range(3)]
    Nz, sg = cfg.Nz, cfg.sigma
    x = p.x0.astype(np.float64)
    A = [cfg.tau / cfg.H * sg[3 + k] / sg[k] * Nz for k in range(3)]
    du = -A[0] * (F[0][:, 1:] - F[0][:, :-1]) + cfg.f * cfg.tau / sg[0] * (sg[1] * x[:, Nz:2 * Nz] + cfg.mu[1])
    dv = -A[1] * (F[1][:, 1:] - F[1][:, :-1]) - cfg.f * cfg.tau / sg[1] * (sg[0] * x[:, :Nz] + cfg.mu[0])
    dT = -A[2] * (F[2][:, 1:] - F[2][:, :-1])
    np.testing.assert_allclose(np.concatenate([du, dv, dT], axis=1), O.rhs(cfg, p.x0, p.bcs, p.weights), rtol=1e-10, atol=1e-9)


def test_pretrain_gradient_matches_finite_differences():
    p = synthetic.wind_mixing_problem(2, n_frames=3, weight_divisor=10.0)
    cfg = p.cfg
    X, B, Y = _data(p, 4)
    th = p.weights[:cfg.net_size].astype(np.float64)
    loss, g = O.nn_pretrain_loss_and_grad(cfg, 2, X[:1], B[:1], O.unpack(th, cfg.layer_sizes, 1)[0], Y[2][:1], 1e-2)
    rng = np.random.default_rng(0)
    for _ in range(3):
        d = rng.standard_normal(th.shape)
        d /= np.linalg.norm(d)
        h = 1e-6
        f = lambda w: O.nn_pretrain_loss_and_grad(cfg, 2, X[:1], B[:1], O.unpack(w, cfg.layer_sizes, 1)[0], Y[2][:1], 1e-2)[0][0]
        assert np.isclose((f(th + h * d) - f(th - h * d)) / (2 * h), g[0] @ d, rtol=1e-5, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["uw_mpp", "vw_mpp_bc", "wT_mpp", "wT_ca", "fc"])
def test_device_pretraining_follows_the_sequential_adam_of_flux_train(case):
    from colnde.wind_mixing import train_NN
    if case == "fc":
        p = synthetic.free_convection_problem(3, Nz=32, n_save=9, substeps=2, t_end=0.05)
        k, gs = 2, 0.0
    else:
        kw = {"uw_mpp": {}, "vw_mpp_bc": VARIANTS["mpp_bc_faces"], "wT_mpp": {}, "wT_ca": VARIANTS["conv_adj_branch"]}[case]
        p = synthetic.wind_mixing_problem(3, n_frames=9, weight_divisor=10.0, **kw)
        k, gs = {"uw": 0, "vw": 1, "wT": 2}[case[:2]], 1e-2
    cfg = p.cfg
    n = 24
    X, B, Y = _data(p, n)
    order = np.random.default_rng(1).permutation(n)
    ns = cfg.net_size
    lo = k * ns if cfg.n_nets == 3 else 0
    ref, hist_ref = O.train_NN(cfg, k, p.weights[lo:lo + ns], X, B, Y[k], order, 1e-3, 2, gs)
    with colnde.ColumnNDE(cfg, 1) as eng:
        w, hist = train_NN(eng, ["uw", "vw", "wT"][k], p.weights, X, B, Y[k], [ADAM(1e-3)], [2], gradient_scaling=gs, order=order)
    assert hist_ref[1] < hist_ref[0]
    np.testing.assert_allclose(hist, hist_ref, rtol=2e-4)
    rel = np.linalg.norm(w[lo:lo + ns] - ref) / np.linalg.norm(ref)
    moved = np.linalg.norm(ref - p.weights[lo:lo + ns]) / np.linalg.norm(ref)
    assert rel < 2e-4 * max(1.0, moved / 1e-2) and moved > 1e-3          # 48 ADAM steps of float32 against float64
    other = np.ones(cfg.n_params, bool)
    other[lo:lo + ns] = False
    np.testing.assert_array_equal(w[other], p.weights[other])            # the other nets are untouched
