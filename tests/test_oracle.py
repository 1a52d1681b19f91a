"""Pins oracle/nde_oracle.py (CPU, no GPU): element-wise definitions, analytic known-answer cases
(SURVEY §8c (i)-(v)), a literal dense-matrix torch restatement with autograd, finite differences,
and a SciPy adaptive solve that *measures* the fixed-RK4-vs-adaptive gap.
The reference holds no golden vectors for this path (parity unpinned) — these are the substitutes."""
import numpy as np
import pytest
import torch

import colnde
from colnde import synthetic
from colnde.config import NDEConfig
from oracle import nde_oracle as O
from tests import literal_torch as LT


# ---------------------------------------------------------------- element-wise definitions (iv)
def test_difference_operators_match_definitions():
    N = 6
    dc, df = O.Dc(N, 1 / N), O.Df(N, 1 / N)
    assert dc.shape == (N, N + 1) and df.shape == (N + 1, N)
    w = np.arange(N + 1.0) ** 2
    np.testing.assert_allclose(dc @ w, (w[1:] - w[:-1]) * N)
    x = np.sin(np.arange(N))
    g = df @ x
    assert g[0] == 0 and g[-1] == 0
    np.testing.assert_allclose(g[1:-1], (x[1:] - x[:-1]) * N)
    # stencil helpers used by the oracle agree with the matrices and are mutual transposes
    np.testing.assert_allclose(O._face_grad(x[None], N)[0], g)
    gb = np.random.default_rng(0).standard_normal(N + 1)
    np.testing.assert_allclose(O._face_grad_T(gb[None], N)[0], df.T @ gb)


def test_smoothing_filter_rows():
    F = O.smoothing_filter(7, 3)
    np.testing.assert_allclose(F.sum(axis=1), 1.0)
    np.testing.assert_allclose(F[0, :3], [0.5, 0.5, 0])
    np.testing.assert_allclose(F[-1, -3:], [0, 0.5, 0.5])
    np.testing.assert_allclose(F[3, 2:5], [1 / 3] * 3)
    assert np.count_nonzero(F[3]) == 3


def test_activations_and_derivatives():
    z = np.linspace(-6, 6, 41)
    np.testing.assert_allclose(O.act("mish", z), z * np.tanh(np.log1p(np.exp(z))), rtol=1e-12)
    np.testing.assert_allclose(O.act("swish", z), z / (1 + np.exp(-z)), rtol=1e-12)
    np.testing.assert_allclose(O.act("leakyrelu", z), np.maximum(0.01 * z, z))
    h = 1e-6
    for name in ("mish", "swish", "tanh", "identity"):
        fd = (O.act(name, z + h) - O.act(name, z - h)) / (2 * h)
        np.testing.assert_allclose(O.act_grad(name, z), fd, rtol=1e-6, atol=1e-8)


def test_feature_scaling_roundtrip():
    # mirrors /root/reference/test/test_feature_scaling.jl:1-15 (zero mean, unit variance, inverse)
    rng = np.random.default_rng(1)
    data = rng.standard_normal(1000) * 3 + 7
    s = colnde.ZeroMeanUnitVarianceScaling.fit(data)
    y = s(data)
    assert abs(y.mean()) < 1e-12 and abs(y.std(ddof=1) - 1) < 1e-12
    np.testing.assert_allclose(s.inv()(y), data, rtol=1e-12)


def test_destructure_order():
    from colnde.flux_compat import destructure, restructure
    W = np.arange(6.0).reshape(2, 3)          # out=2, in=3
    b = np.array([10.0, 11.0])
    th = destructure([(W, b)])
    np.testing.assert_allclose(th, [0, 3, 1, 4, 2, 5, 10, 11])   # column-major vec(W), then b
    (W2, b2), = restructure(th, (3, 2))
    np.testing.assert_allclose(W2, W)
    nets = O.unpack(th, (3, 2), 1)
    np.testing.assert_allclose(nets[0][0][0], W)


def test_loss_scaling_ratio_identities():
    # the identities /root/reference/wind_mixing/test/test_training_scaling.jl:17-19 asserts
    L = np.array([0.3, 0.2, 0.05, 4.0, 3.0, 0.7])
    fr = dict(T=0.8, dTdz=0.8, profile=0.5)
    s = O.calculate_loss_scalings(L, fr, True)
    sc = s * L
    assert np.isclose(sc[2] / (sc[0] + sc[1]), fr["T"] / (1 - fr["T"]))
    assert np.isclose(sc[5] / (sc[3] + sc[4]), fr["dTdz"] / (1 - fr["dTdz"]))
    assert np.isclose(sc[:3].sum() / sc[3:].sum(), fr["profile"] / (1 - fr["profile"]))


# ---------------------------------------------------------------- literal restatement: RHS values
VARIANTS = {
    "mpp_zero_weights": {},
    "mpp_bc_faces": dict(zero_weights=False),
    # κ = 0.1 keeps the explicit RK4 step inside its stability region where ∂T∂z < 0 (κ = 10, the reference default,
    # needs ≈20 sub-steps per 10-minute frame: the reference reaches for ROCK4 there)
    "conv_adj_branch": dict(modified_pacanowski_philander=False, zero_weights=False, convective_adjustment=True, kappa=0.1),
    "raw": dict(modified_pacanowski_philander=False, zero_weights=False),
    "smooth_NN": dict(smooth_NN=True),
    "smooth_Ri": dict(smooth_Ri=True),
    "diurnal": dict(diurnal=True),
    "swish": dict(activations=("swish", "swish", "identity")),
    "relu": dict(activations=("relu", "relu", "identity")),
    "tanh": dict(activations=("tanh", "tanh", "identity")),
    "leakyrelu": dict(activations=("leakyrelu", "leakyrelu", "identity")),
    "dRi_small": dict(dRi=0.1),
}


def _wm(n, **kw):
    return synthetic.wind_mixing_problem(n, n_frames=3, weight_divisor=10.0, **kw)


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_wind_mixing_rhs_matches_literal(name):
    p = _wm(3, **VARIANTS[name])
    t = 0.013
    dx = O.rhs(p.cfg, p.x0, p.bcs, p.weights, t)
    th = torch.tensor(p.weights.astype(np.float64))
    for c in range(3):
        ref = LT.wm_rhs(p.cfg, torch.tensor(p.x0[c].astype(np.float64)),
                        torch.tensor(p.bcs[c].astype(np.float64)), th, t).numpy()
        np.testing.assert_allclose(dx[c], ref, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("ca", [False, True])
def test_free_convection_rhs_matches_literal(ca):
    p = synthetic.free_convection_problem(3, Nz=16, n_save=3, convective_adjustment=ca)
    x0 = p.x0.copy()
    x0[:, 8:12] = x0[:, 8:12][:, ::-1]      # force some unstable faces
    dx = O.rhs(p.cfg, x0, p.bcs, p.weights)
    th = torch.tensor(p.weights.astype(np.float64))
    for c in range(3):
        ref = LT.fc_rhs(p.cfg, torch.tensor(x0[c].astype(np.float64)),
                        torch.tensor(p.bcs[c].astype(np.float64)), th).numpy()
        np.testing.assert_allclose(dx[c], ref, rtol=1e-10, atol=1e-10)


def test_inplace_variant_differs_only_at_eps():
    p = _wm(2)
    a = O.rhs(p.cfg, p.x0, p.bcs, p.weights)
    b = O.rhs(p.cfg.with_(inplace_variant=True), p.x0, p.bcs, p.weights)
    assert np.abs(a - b).max() < 1e-4 * np.abs(a).max()


# the in-place `NDE!` arithmetic (training_postprocessing.jl:55-153): cases shared by the oracle, C-port and GPU tests
INPLACE_VARIANTS = {
    "inplace": dict(inplace_variant=True),
    "inplace_ca": dict(inplace_variant=True, convective_adjustment=True),                    # ν_T = κ where ∂u∂z <= 0 (:118-121)
    "inplace_diurnal": dict(inplace_variant=True, diurnal=True),                             # un-offset top flux (:142-144)
    "inplace_ca_diurnal": dict(inplace_variant=True, convective_adjustment=True, diurnal=True),
}


@pytest.mark.parametrize("name", sorted(INPLACE_VARIANTS))
def test_inplace_rhs_matches_literal(name):
    p = _wm(3, **INPLACE_VARIANTS[name])
    t = 0.013
    dx = O.rhs(p.cfg, p.x0, p.bcs, p.weights, t)
    th = torch.tensor(p.weights.astype(np.float64))
    for c in range(3):
        ref = LT.wm_rhs_inplace(p.cfg, torch.tensor(p.x0[c].astype(np.float64)),
                                torch.tensor(p.bcs[c].astype(np.float64)), th, t).numpy()
        np.testing.assert_allclose(dx[c], ref, rtol=1e-9, atol=1e-9)
    if "ca" in name:                       # the switch is live: both branches occur on these profiles, and it matters
        gu = O._face_grad(p.x0[:, :32].astype(np.float64), 32)[:, 1:32]
        assert (gu > 0).any() and (gu <= 0).any()
        plain = O.rhs(p.cfg.with_(convective_adjustment=False), p.x0, p.bcs, p.weights, t)
        assert np.abs(dx - plain).max() > 1e-2 * np.abs(plain).max()
    if "diurnal" in name:                  # the missing `- scaling(0)` offset is visible in the surface cell only
        off = O.rhs(p.cfg.with_(inplace_variant=False, convective_adjustment=False), p.x0, p.bcs, p.weights, t)
        base = O.rhs(p.cfg.with_(convective_adjustment=False), p.x0, p.bcs, p.weights, t)
        assert np.abs((base - off)[:, 95]).min() > 1e-3 and np.abs((base - off)[:, 64:95]).max() < 1e-3 * np.abs(off).max()


# ---------------------------------------------------------------- adjoint vs torch autograd
def _autograd_case(p, scal):
    cfg = p.cfg
    truth = O.solve(cfg, p.x0, p.bcs, p.weights_truth)
    tot, terms, g, sol = O.loss_and_grad(cfg, p.x0, p.bcs, p.weights, truth, scal)
    th = torch.tensor(p.weights.astype(np.float64), requires_grad=True)
    sols = [LT.solve_rk4(cfg, torch.tensor(p.x0[c].astype(np.float64)),
                         torch.tensor(p.bcs[c].astype(np.float64)), th) for c in range(p.n_columns)]
    L = LT.total_loss(cfg, sols, [torch.tensor(truth[c]) for c in range(p.n_columns)], scal)
    L.backward()
    np.testing.assert_allclose(sol, torch.stack(sols).detach().numpy(), rtol=1e-9, atol=1e-10)
    assert np.isclose(tot, L.item(), rtol=1e-10)
    gref = th.grad.numpy()
    assert np.linalg.norm(g - gref) <= 1e-8 * np.linalg.norm(gref) + 1e-14
    return tot, g


@pytest.mark.parametrize("name", ["mpp_zero_weights", "mpp_bc_faces", "conv_adj_branch", "smooth_NN",
                                  "smooth_Ri", "diurnal"])
def test_wind_mixing_adjoint_matches_autograd(name):
    p = _wm(2, **VARIANTS[name])
    tot, g = _autograd_case(p, np.array([1.0, 0.7, 1.3, 5e-3, 4e-3, 6e-3]))
    assert tot > 0 and np.linalg.norm(g) > 0


@pytest.mark.parametrize("ca", [False, True])
def test_free_convection_adjoint_matches_autograd(ca):
    p = synthetic.free_convection_problem(2, Nz=16, n_save=3, substeps=8 if ca else 2,
                                          convective_adjustment=ca, t_end=0.01)
    _autograd_case(p, np.array([0, 0, 1.0, 0, 0, 0]))


def test_adjoint_finite_differences_long_horizon():
    p = synthetic.wind_mixing_problem(3, n_frames=17, weight_divisor=1e2)
    cfg = p.cfg
    truth = O.solve(cfg, p.x0, p.bcs, p.weights_truth)
    sc = O.default_loss_scalings(cfg)
    tot, _, g, _ = O.loss_and_grad(cfg, p.x0, p.bcs, p.weights, truth, sc)
    rng = np.random.default_rng(3)
    w = p.weights.astype(np.float64)
    for _ in range(4):
        d = rng.standard_normal(w.shape)
        d /= np.linalg.norm(d)
        h = 1e-6
        lp = O.loss(cfg, O.solve(cfg, p.x0, p.bcs, w + h * d), truth, sc)[0]
        lm = O.loss(cfg, O.solve(cfg, p.x0, p.bcs, w - h * d), truth, sc)[0]
        assert np.isclose((lp - lm) / (2 * h), g @ d, rtol=2e-5, atol=1e-12)


# ---------------------------------------------------------------- analytic known answers
def _zero_nets(cfg):
    return np.zeros(cfg.n_params, dtype=np.float64)


def test_kat_inertial_oscillation():
    """(i) zero weights, ν₀=ν₋=0, zero boundary flux ⇒ (σ_u u+μ_u, σ_v v+μ_v) rotates by −fτ·t̂, T constant."""
    p = synthetic.wind_mixing_problem(2, n_frames=33, substeps=4, nu0=0.0, nu_minus=0.0)
    cfg = p.cfg
    s0 = [-cfg.mu[3 + k] / cfg.sigma[3 + k] for k in range(3)]
    bcs = np.tile(np.array([s0[0], s0[0], s0[1], s0[1], s0[2], s0[2]]), (2, 1))
    sol = O.solve(cfg, p.x0, bcs, _zero_nets(cfg))
    Nz = cfg.Nz
    U0 = cfg.sigma[0] * p.x0[:, :Nz] + cfg.mu[0]
    V0 = cfg.sigma[1] * p.x0[:, Nz:2 * Nz] + cfg.mu[1]
    th = cfg.f * cfg.tau * cfg.save_times[-1]
    U = U0 * np.cos(th) + V0 * np.sin(th)
    V = -U0 * np.sin(th) + V0 * np.cos(th)
    np.testing.assert_allclose(cfg.sigma[0] * sol[:, -1, :Nz] + cfg.mu[0], U, atol=2e-7)
    np.testing.assert_allclose(cfg.sigma[1] * sol[:, -1, Nz:2 * Nz] + cfg.mu[1], V, atol=2e-7)
    np.testing.assert_allclose(sol[:, -1, 2 * Nz:], p.x0[:, 2 * Nz:], atol=1e-13)


def test_kat_free_convection_zero_weights_linear():
    """(ii) zero weights ⇒ only cells 0 and Nz-1 change, linearly in t̂ (exact for any RK)."""
    p = synthetic.free_convection_problem(3, Nz=16, n_save=5, substeps=1, t_end=0.3)
    cfg = p.cfg
    sol = O.solve(cfg, p.x0, p.bcs, _zero_nets(cfg))
    C = cfg.sigma[5] / cfg.sigma[2] * cfg.tau / cfg.H
    t = np.asarray(cfg.save_times)
    x0 = p.x0.astype(np.float64)
    np.testing.assert_allclose(sol[:, :, 1:-1], np.repeat(x0[:, None, 1:-1], 5, 1), atol=1e-13)
    np.testing.assert_allclose(sol[:, :, -1], x0[:, -1:] - C * cfg.Nz * p.bcs[:, 1:2] * t[None], rtol=1e-12)
    np.testing.assert_allclose(sol[:, :, 0], x0[:, :1] + C * cfg.Nz * p.bcs[:, 0:1] * t[None], rtol=1e-12)


def test_kat_linear_diffusion_modes():
    """(iii) zero weights, ν≡ν₀+ν₋ (Riᶜ→∞), f=0, zero flux ⇒ cosine modes decay by the RK4 stability
    polynomial R(−κ̂λ_k Δt) per step, κ̂=τν/H², λ_k=4Nz² sin²(kπ/2Nz) — exact for the discrete scheme."""
    p = synthetic.wind_mixing_problem(1, n_frames=9, substeps=2, Ric=1e30, f=0.0)
    cfg = p.cfg
    Nz = cfg.Nz
    s0 = [-cfg.mu[3 + k] / cfg.sigma[3 + k] for k in range(3)]
    bcs = np.array([[s0[0], s0[0], s0[1], s0[1], s0[2], s0[2]]])
    kk = 5
    mode = np.cos(kk * np.pi * (np.arange(Nz) + 0.5) / Nz)
    x0 = np.concatenate([0.3 * mode, -0.2 * mode, 0.1 + 0.0 * mode])[None]
    sol = O.solve(cfg, x0, bcs, _zero_nets(cfg))
    kap = cfg.tau * (cfg.nu0 + cfg.nu_minus) / cfg.H ** 2
    lam = 4 * Nz ** 2 * np.sin(kk * np.pi / (2 * Nz)) ** 2
    dt = (cfg.save_times[1] - cfg.save_times[0]) / cfg.substeps
    z = -kap * lam * dt
    R = 1 + z + z ** 2 / 2 + z ** 3 / 6 + z ** 4 / 24
    amp = R ** cfg.n_steps
    np.testing.assert_allclose(sol[0, -1, :Nz], 0.3 * mode * amp, atol=1e-12)
    np.testing.assert_allclose(sol[0, -1, Nz:2 * Nz], -0.2 * mode * amp, atol=1e-12)
    assert abs(amp - np.exp(-kap * lam * cfg.save_times[-1])) < 1e-3


def test_kat_column_sum_conservation():
    """(v) zero boundary fluxes ⇒ Σ_k ∂T∂t = 0 (telescoping Dᶜ), with non-trivial nets."""
    p = _wm(4)
    cfg = p.cfg
    s0 = [-cfg.mu[3 + k] / cfg.sigma[3 + k] for k in range(3)]
    bcs = np.tile(np.array([s0[0], s0[0], s0[1], s0[1], s0[2], s0[2]]), (4, 1))
    dx = O.rhs(cfg, p.x0, bcs, p.weights)
    assert np.abs(dx[:, 2 * cfg.Nz:].sum(axis=1)).max() < 1e-9 * np.abs(dx).max()


def test_loss_terms_include_zero_rows():
    """∂/∂z terms average over Nz+1 rows including the two all-zero ones (loss.jl:9, NDE_training.jl:315-317)."""
    cfg = _wm(1).cfg
    Nz = cfg.Nz
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal((2, 2, 4, 3 * Nz))
    t = O.loss_terms(cfg, a, b)
    Dfm = O.Df(Nz, 1 / Nz)
    d = (b - a)[:, :, Nz:2 * Nz]
    ref = np.mean([np.mean((d[i] @ Dfm.T) ** 2) for i in range(2)])
    assert np.isclose(t[4], ref) and (d[0] @ Dfm.T).shape[1] == Nz + 1


# ---------------------------------------------------------------- fixed RK4 vs adaptive solver
def test_fixed_rk4_vs_adaptive_gap_is_below_reference_tolerance():
    """Measures the gap between 2-substep RK4 and an adaptive solve at the reference's reltol=1e-3
    (NDE_training.jl:291); the build's trajectories agree with "the reference CPU solve" to this level only."""
    from scipy.integrate import solve_ivp
    p = synthetic.wind_mixing_problem(1, n_frames=49, weight_divisor=1e2)
    cfg = p.cfg
    sol = O.solve(cfg, p.x0, p.bcs, p.weights)
    m = O.Model(cfg)
    nets = m.unpack(p.weights)
    bc = p.bcs.astype(np.float64)
    f = lambda t, x: m.rhs(x[None], bc, nets, t)[0]
    tight = solve_ivp(f, (0, cfg.save_times[-1]), p.x0[0].astype(np.float64), method="LSODA",
                      t_eval=cfg.save_times, rtol=1e-9, atol=1e-11).y.T
    loose = solve_ivp(f, (0, cfg.save_times[-1]), p.x0[0].astype(np.float64), method="RK45",
                      t_eval=cfg.save_times, rtol=1e-3, atol=1e-6).y.T
    gap_fixed = np.abs(sol[0] - tight).max()
    gap_adaptive = np.abs(loose - tight).max()
    print("max |RK4(S=2) - tight| = %.3e ; max |RK45(rtol=1e-3) - tight| = %.3e" % (gap_fixed, gap_adaptive))
    assert gap_fixed < 5e-3
    assert gap_fixed < 10 * max(gap_adaptive, 1e-4)


# ---------------------------------------------------------------- discrete RK4 adjoint vs the reference's kind of gradient
def test_discrete_vs_continuous_adjoint_gradient_gap():
    """north_star: "match the reference CPU DiffEqFlux solve within a stated tolerance".  The reference differentiates an adaptive
    solve (reltol 1e-3, abstol 1e-6) with a CONTINUOUS interpolating adjoint (NDE_training.jl:304,327-333); the
    product back-propagates through its own fixed RK4 steps.  oracle/continuous_adjoint.py restates the former with SciPy; here the
    gap between the two gradients is measured on a reduced config 3 (3 simulations x 49 frames) and bounded.  On the full config 3
    (8 x 289 frames, tools/adjoint_gap.py -> profiles/r02_adjoint_gap.json): 2.4e-5 relative L2, cosine 1 - 3e-10, and the
    discrete gradient is the closer of the two to a 16-sub-step reference (5.5e-7 vs 2.4e-5): the stated tolerance is the
    reference's own solver tolerance, not an error of the discrete adjoint."""
    from oracle import continuous_adjoint as CAd
    p = synthetic.wind_mixing_problem(3, n_frames=49, weight_divisor=1e2)
    cfg = p.cfg
    truth = O.solve(cfg.with_(substeps=16), p.x0, p.bcs, p.weights_truth)
    sc = O.default_loss_scalings(cfg)
    tot, _, g, sol = O.loss_and_grad(cfg, p.x0, p.bcs, p.weights, truth, sc)
    _, _, g_x, _ = O.loss_and_grad(cfg.with_(substeps=16), p.x0, p.bcs, p.weights, truth, sc)
    tot_c, _, g_c, sol_c, st = CAd.loss_and_grad_continuous(cfg, p.x0, p.bcs, p.weights, truth, sc, rtol=1e-3, atol=1e-6)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    gap, gap_d, gap_c = rel(g, g_c), rel(g, g_x), rel(g_c, g_x)
    print("gradient gaps: discrete vs continuous %.3e, discrete vs fine %.3e, continuous vs fine %.3e; loss gap %.3e; %s"
          % (gap, gap_d, gap_c, abs(tot - tot_c) / tot_c, st))
    assert gap < 1e-3                           # the stated gradient tolerance against a reltol-1e-3 reference solve
    assert gap_d < gap_c                        # ... all of which is the adaptive solve's own error
    assert g @ g_c > (1 - 1e-6) * np.linalg.norm(g) * np.linalg.norm(g_c)
    # tightening the reference's tolerances closes the gap: the two adjoints differentiate the same continuous problem
    _, _, g_t, _, _ = CAd.loss_and_grad_continuous(cfg, p.x0[:1], p.bcs[:1], p.weights, truth[:1], sc, rtol=1e-7, atol=1e-10)
    _, _, g_1, _ = O.loss_and_grad(cfg.with_(substeps=16), p.x0[:1], p.bcs[:1], p.weights, truth[:1], sc)
    assert rel(g_t, g_1) < 1e-5


# ---------------------------------------------------------------- stabilised stepper (RKC2) in the oracle
def test_rkc2_coefficients_and_linear_decay():
    """Consistency (c_s = 1, second order: sum of the quadrature conditions through the stability polynomial), the stability
    boundary beta(s) ~ 0.653 s^2, and the linear-diffusion known answer: a cosine mode decays by the RKC2 stability polynomial
    P_s(z) = a_s + b_s T_s(w0 + w1 z) per step (Sommeijer-Shampine-Verwer 1998, eq. 2.3)."""
    for s in (2, 3, 5, 12, 40):
        mu, nu, mut, gat, c, beta = O.rkc_coefficients(s)
        assert abs(c[s] - 1.0) < 1e-12
        assert 0.49 * s * s <= beta <= 0.66 * s * s
    p = synthetic.wind_mixing_problem(1, n_frames=9, substeps=1, Ric=1e30, f=0.0, stepper="rkc2", rkc_stages=6)
    cfg = p.cfg
    Nz = cfg.Nz
    s0 = [-cfg.mu[3 + k] / cfg.sigma[3 + k] for k in range(3)]
    bcs = np.array([[s0[0], s0[0], s0[1], s0[1], s0[2], s0[2]]])
    kk = 5
    mode = np.cos(kk * np.pi * (np.arange(Nz) + 0.5) / Nz)
    x0 = np.concatenate([0.3 * mode, -0.2 * mode, 0.1 + 0.0 * mode])[None]
    sol = O.solve(cfg, x0, bcs, np.zeros(cfg.n_params))
    kap = cfg.tau * (cfg.nu0 + cfg.nu_minus) / cfg.H ** 2
    lam = 4 * Nz ** 2 * np.sin(kk * np.pi / (2 * Nz)) ** 2
    z = -kap * lam * (cfg.save_times[1] - cfg.save_times[0])
    s = 6
    w0 = 1 + O.RKC_EPS / s ** 2
    T, dT, d2T = O._cheb(s, w0)
    w1, b = dT[s] / d2T[s], d2T[s] / dT[s] ** 2
    # T_s(w0 + w1 z) by the recurrence
    x = w0 + w1 * z
    Ta, Tb = 1.0, x
    for _ in range(2, s + 1):
        Ta, Tb = Tb, 2 * x * Tb - Ta
    P = (1 - b * T[s]) + b * Tb
    np.testing.assert_allclose(sol[0, -1, :Nz], 0.3 * mode * P ** 8, atol=1e-12)
    assert abs(P - np.exp(z)) < 0.1 * abs(z) ** 3             # second order: the error of one step is O(z^3)


def test_rkc2_adjoint_matches_finite_differences_on_a_stiff_case():
    """kappa = 10 in the convective-adjustment branch (lambda h = 375: RK4 would need 135 sub-steps per frame) with one 26-stage RKC2
    step per frame: the discrete adjoint of the recurrence against central differences of the loss."""
    p = synthetic.wind_mixing_problem(2, n_frames=5, weight_divisor=1e2, modified_pacanowski_philander=False, zero_weights=False,
                                      convective_adjustment=True, kappa=10.0, stepper="rkc2", substeps=1)
    cfg = p.cfg
    assert O.rkc_stages(cfg) == colnde.rkc_stages(cfg) == 26 and colnde.min_substeps(cfg) == 1
    assert colnde.min_substeps(cfg.with_(stepper="rk4")) == 135
    truth = O.solve(cfg, p.x0, p.bcs, p.weights_truth)
    assert np.isfinite(truth).all()
    sc = O.default_loss_scalings(cfg)
    tot, _, g, _ = O.loss_and_grad(cfg, p.x0, p.bcs, p.weights, truth, sc)
    rng = np.random.default_rng(5)
    w = p.weights.astype(np.float64)
    for _ in range(3):
        d = rng.standard_normal(w.shape)
        d /= np.linalg.norm(d)
        h = 1e-6
        lp = O.loss(cfg, O.solve(cfg, p.x0, p.bcs, w + h * d), truth, sc)[0]
        lm = O.loss(cfg, O.solve(cfg, p.x0, p.bcs, w - h * d), truth, sc)[0]
        assert np.isclose((lp - lm) / (2 * h), g @ d, rtol=2e-5, atol=1e-12)


def test_rkc2_stage_count_follows_the_stiffness_bound():
    for Nz, n_save, sub in ((64, 129, 1), (64, 129, 4), (32, 33, 1)):
        cfg = synthetic.free_convection_problem(1, Nz=Nz, n_save=n_save, substeps=sub, convective_adjustment=True).cfg.with_(stepper="rkc2")
        s = colnde.rkc_stages(cfg)
        assert s == O.rkc_stages(cfg)
        zh = O.stiff_lambda(cfg) / (n_save - 1) / sub
        assert 0.9 * O.rkc_coefficients(s)[5] >= zh and (s == 2 or 0.9 * O.rkc_coefficients(s - 1)[5] < zh)
    # an explicit stage count that does not cover the stiffest mode asks for more sub-steps
    cfg = synthetic.free_convection_problem(1, Nz=64, n_save=129, substeps=1, convective_adjustment=True).cfg.with_(stepper="rkc2", rkc_stages=10)
    assert colnde.min_substeps(cfg) == int(np.ceil(O.stiff_lambda(cfg) / 128 / (0.9 * O.rkc_coefficients(10)[5])))


def test_rkc2_accuracy_against_converged_rk4():
    """Where stability no longer sets the step, accuracy does: ConvectiveAdjustmentNDE at 64 levels from an inverted 24-cell layer,
    RKC2 against a converged RK4 solve (230 sub-steps per interval; 460 agree to 7e-6) over the first 16 of config 4's 128 save
    intervals — the error is made in the first adjustment and then stays.  Second-order convergence in the step count."""
    p = synthetic.free_convection_problem(1, Nz=64, n_save=17, substeps=1, convective_adjustment=True, t_end=0.125)
    x0 = p.x0.copy()
    x0[:, 20:44] = x0[:, 20:44][:, ::-1]
    ref = O.solve(p.cfg.with_(substeps=230), x0, p.bcs, p.weights)
    errs = {}
    for sub in (1, 4, 8):
        cfg = p.cfg.with_(stepper="rkc2", substeps=sub)
        errs[sub] = np.abs(O.solve(cfg, x0, p.bcs, p.weights) - ref)[:, -1].max()
        print("RKC2 %d x %d stages: |error| at t = 0.125: %.2e" % (sub, O.rkc_stages(cfg), errs[sub]))
    assert errs[1] < 0.3 and errs[4] < 2e-2 and errs[8] < 2e-4


def test_rkc2_switch_pullback():
    """The exact discrete adjoint of an s-stage stabilised step does not survive a switching right-hand side: with the switch of
    min(0, K dT/dz) evaluated stage by stage, the stage Jacobians differ, the Chebyshev cancellations fail, and the back-propagated
    cotangent grows without bound — while sub-stepped RK4's discrete gradient is benign.  The product (and this oracle) therefore
    pull the switch back with ONE pattern per step (that of Y_{s-1}); the resulting gradient converges to RK4's as the step shrinks.
    64-level ConvectiveAdjustmentNDE from an inverted layer, 8 save intervals of config 4's axis."""
    p = synthetic.free_convection_problem(1, Nz=64, n_save=9, substeps=1, convective_adjustment=True, t_end=8 / 128)
    x0 = p.x0.copy()
    x0[:, 20:44] = x0[:, 20:44][:, ::-1]
    sc = O.default_loss_scalings(p.cfg)
    truth = O.solve(p.cfg.with_(substeps=230), x0, p.bcs, p.weights_truth)
    _, _, g4, _ = O.loss_and_grad(p.cfg.with_(substeps=230), x0, p.bcs, p.weights, truth, sc)
    cos = lambda a, b: a @ b / np.linalg.norm(a) / np.linalg.norm(b)
    with np.errstate(all="ignore"):
        _, _, g_exact, _ = O.loss_and_grad(p.cfg.with_(stepper="rkc2", substeps=8, rkc_exact_switch_pullback=True), x0, p.bcs, p.weights, truth, sc)
    assert not np.isfinite(g_exact).all() or np.linalg.norm(g_exact) > 1e6 * np.linalg.norm(g4)
    res = {}
    for sub in (8, 16):
        _, _, g, _ = O.loss_and_grad(p.cfg.with_(stepper="rkc2", substeps=sub), x0, p.bcs, p.weights, truth, sc)
        res[sub] = (np.linalg.norm(g - g4) / np.linalg.norm(g4), cos(g, g4))
        print("RKC2 %d steps per interval, one switch pattern per step: |g - g_RK4|/|g_RK4| = %.3f, cosine %.5f" % (sub, *res[sub]))
    assert res[8][1] > 0.99 and res[16][1] > 0.999 and res[16][0] < res[8][0] < 0.15
