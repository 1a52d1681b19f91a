"""The steps either side of the hot path: the implicit convective-adjustment step
(free_convection/double_gyre_nn.jl:27-62) and Flux's ADAM update.  CPU tests pin the oracle's restatement by its
properties; GPU tests compare the HIP kernels with it through the C ABI."""
import numpy as np
import pytest

from oracle import nde_oracle as orc
from colnde import flux_compat, synthetic

RNG = np.random.default_rng(20261004)


def _columns(n, Nz, unstable=True):
    T = np.linspace(5.0, 25.0, Nz)[None, :] + (1.5 if unstable else 0.0) * RNG.standard_normal((n, Nz))
    return T.astype(np.float32)


# ---------------------------------------------------------------------------------------------- oracle (CPU)
def test_oracle_convadj_identity_on_stable_columns():
    T = np.linspace(5.0, 25.0, 32)[None, :].repeat(3, 0)
    out = orc.convective_adjustment(T, 600.0, 62.5, 10.0)
    np.testing.assert_allclose(out, T, rtol=0, atol=1e-12)


def test_oracle_convadj_matches_thomas_recurrence_and_bounds():
    T = _columns(16, 32).astype(np.float64)
    dt, dz, K = 1200.0, 62.5, 10.0
    out = orc.convective_adjustment(T, dt, dz, K)
    # independent Thomas elimination of the same rows
    c = dt / dz ** 2
    for i in range(T.shape[0]):
        x = T[i].copy()
        ext = np.concatenate([[x[0]], x, [x[-1]]])
        k = np.where(ext[2:] - ext[:-2] < 0, c * K, 0.0)
        Nz = x.size
        cp = np.zeros(Nz)
        b0 = 1 + k[0] + k[1]
        cp[0] = -k[1] / b0
        x[0] /= b0
        for r in range(1, Nz):
            a = -k[r]
            b = 1 + k[r] + (k[r + 1] if r < Nz - 1 else 0.0)
            den = b - a * cp[r - 1]
            cp[r] = (-k[r + 1] if r < Nz - 1 else 0.0) / den
            x[r] = (x[r] - a * x[r - 1]) / den
        for r in range(Nz - 2, -1, -1):
            x[r] -= cp[r] * x[r + 1]
        np.testing.assert_allclose(out[i], x, rtol=1e-12, atol=1e-12)
    # an M-matrix solve with row sums >= 1: the result stays inside the column's range scaled towards zero
    assert (out.max(axis=1) <= T.max(axis=1) + 1e-9).all()
    assert np.isfinite(out).all()


def test_oracle_convadj_halo_values_change_only_the_end_cells_kappa():
    T = np.linspace(5.0, 25.0, 16)[None, :].copy()
    base = orc.convective_adjustment(T, 600.0, 10.0, 5.0)
    cold_top = orc.convective_adjustment(T, 600.0, 10.0, 5.0, halo_top=np.array([0.0]))    # unstable top cell only
    assert np.allclose(base, T)
    assert not np.allclose(cold_top[:, -2:], T[:, -2:])
    assert np.allclose(cold_top[:, :-3], T[:, :-3], atol=1e-3)


def test_oracle_adam_step_matches_flux_compat():
    n = 257
    theta = RNG.standard_normal(n)
    opt = flux_compat.ADAM(1e-2)
    th_host = theta.astype(np.float64).copy()
    th, m, v, bt = theta.copy(), np.zeros(n), np.zeros(n), (0.9, 0.999)
    for _ in range(4):
        g = RNG.standard_normal(n)
        opt.update(th_host, g)
        th, m, v, bt = orc.adam_step(th, g, m, v, 1e-2, (0.9, 0.999), 1e-8, bt)
    np.testing.assert_allclose(th, th_host, rtol=1e-12, atol=1e-14)
    # first step moves every parameter by ~eta against the gradient sign
    th1, _, _, _ = orc.adam_step(np.zeros(3), np.array([1.0, -2.0, 0.5]), np.zeros(3), np.zeros(3), 1e-3, (0.9, 0.999), 1e-8, (0.9, 0.999))
    np.testing.assert_allclose(th1, [-1e-3, 1e-3, -1e-3], rtol=1e-6)


# ---------------------------------------------------------------------------------------------- HIP kernels
@pytest.mark.gpu
@pytest.mark.parametrize("Nz,n", [(32, 1), (32, 255), (32, 1000), (64, 513), (16, 300), (20, 77)])
def test_gpu_convective_adjustment_matches_oracle(Nz, n):
    import colnde
    cfg = synthetic.free_convection_problem(1, Nz=Nz, n_save=2).cfg
    T = _columns(n, Nz)
    dt, dz, K = 1200.0, 2000.0 / Nz, 10.0
    want = orc.convective_adjustment(T, dt, dz, K)
    with colnde.ColumnNDE(cfg, 1) as nde:
        got = nde.convective_adjustment(T, dt, dz, K)
        assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max()
        # caller-supplied halo cells (Value / Gradient boundary conditions)
        hb = (T[:, 0] + RNG.standard_normal(n)).astype(np.float32)
        ht = (T[:, -1] + RNG.standard_normal(n)).astype(np.float32)
        want_h = orc.convective_adjustment(T, dt, dz, K, hb, ht)
        got_h = nde.convective_adjustment(T, dt, dz, K, hb, ht)
        assert np.abs(got_h - want_h).max() <= 2e-5 * np.abs(want_h).max()
        # stable columns are untouched, bit for bit
        Ts = _columns(n, Nz, unstable=False)
        assert np.array_equal(nde.convective_adjustment(Ts, dt, dz, K), Ts)


@pytest.mark.gpu
def test_gpu_convective_adjustment_device_twin_in_place_and_bad_arguments():
    import torch
    import colnde
    cfg = synthetic.free_convection_problem(1, Nz=32, n_save=2).cfg
    T = _columns(4096, 32)
    want = orc.convective_adjustment(T, 600.0, 62.5, 10.0)
    with colnde.ColumnNDE(cfg, 1) as nde:
        Td = torch.from_numpy(T).cuda()
        nde.convective_adjustment(Td, 600.0, 62.5, 10.0, out=Td)          # in place
        torch.cuda.synchronize()
        assert np.abs(Td.cpu().numpy() - want).max() <= 2e-5 * np.abs(want).max()
        with pytest.raises(colnde.ColndeError):
            nde.convective_adjustment(T, -1.0, 62.5, 10.0)
        with pytest.raises(colnde.ColndeError):
            nde.convective_adjustment(T, 600.0, 0.0, 10.0)


@pytest.mark.gpu
def test_gpu_adam_step_matches_oracle_over_several_steps():
    import torch
    import colnde
    p = synthetic.wind_mixing_problem(2, n_frames=3)
    n = p.cfg.n_params
    with colnde.ColumnNDE(p.cfg, 2) as nde:
        theta = RNG.standard_normal(n).astype(np.float32)
        th, m, v, bt = theta.astype(np.float64), np.zeros(n), np.zeros(n), (0.9, 0.999)
        thd = torch.from_numpy(theta).cuda()
        md, vd = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        btd = [0.9, 0.999]
        for _ in range(5):
            g = (1e-3 * RNG.standard_normal(n)).astype(np.float32)
            th, m, v, bt = orc.adam_step(th, g, m, v, 1e-2, (0.9, 0.999), 1e-8, bt)
            nde.adam_step(thd, torch.from_numpy(g).cuda(), md, vd, 1e-2, (0.9, 0.999), 1e-8, beta_t=tuple(btd))
            btd = [btd[0] * 0.9, btd[1] * 0.999]
        torch.cuda.synchronize()
        np.testing.assert_allclose(thd.cpu().numpy(), th, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(md.cpu().numpy(), m, rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(vd.cpu().numpy(), v, rtol=1e-5, atol=1e-12)


@pytest.mark.gpu
def test_gpu_device_resident_training_follows_the_host_loop():
    from colnde import wind_mixing as wm
    p = synthetic.wind_mixing_problem(8, n_frames=9, weight_divisor=1e2)
    host = wm.WindMixingNDE(p.cfg, p.x0, p.bcs)
    truth = host.engine.forward(p.weights_truth)
    host.close()
    a = wm.WindMixingNDE(p.cfg, p.x0, p.bcs, truth)
    b = wm.WindMixingNDE(p.cfg, p.x0, p.bcs, truth)
    ra = wm.train_NDE(a, p.weights, [flux_compat.ADAM(3e-4)], epochs=1, maxiters=6)
    rb = wm.train_NDE_device(b, p.weights, [flux_compat.ADAM(3e-4)], epochs=1, maxiters=6)
    a.close(); b.close()
    la = np.array([h["total"] for h in ra.history]); lb = np.array([h["total"] for h in rb.history])
    np.testing.assert_allclose(lb, la, rtol=2e-3)
    assert lb[-1] < lb[0]
    np.testing.assert_allclose(rb.weights, ra.weights, rtol=1e-3, atol=2e-6)


@pytest.mark.gpu
def test_gpu_free_convection_device_training_follows_the_host_loop():
    from colnde import free_convection as fc
    p = synthetic.free_convection_problem(6, Nz=32, n_save=5, substeps=2, t_end=0.02)
    truth = orc.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    a = fc.FreeConvectionNDE(p.cfg, p.x0, p.bcs, truth)
    b = fc.FreeConvectionNDE(p.cfg, p.x0, p.bcs, truth)
    tha, ha = fc.train_neural_differential_equation(a, p.weights, flux_compat.ADAM(1e-3), epochs=6)
    thb, hb = fc.train_neural_differential_equation_device(b, p.weights, flux_compat.ADAM(1e-3), epochs=6)
    a.close(); b.close()
    np.testing.assert_allclose(hb, ha, rtol=2e-3)
    assert hb[-1] < hb[0]
    np.testing.assert_allclose(thb, tha, rtol=1e-3, atol=2e-6)


# ---------------------------------------------------------------------------------------------- implicit MPP diffusion step
# modified_pacanowski_philander! (wind_mixing/src/NDE_oceananigans.jl:61-101): the third cited site of SURVEY §8f rank 1
MPP = dict(nu0=1e-4, nu_minus=1e-1, dRi=1.0, Ric=0.25, Pr=1.0, alpha=1.67e-4, g=9.81)        # test_nonmutating_NDE.jl:52-53


def _uvT(n, Nz, rng, unstable=True):
    """Ocean-like columns in model units: sheared velocities (m/s), a stratified T (deg C) with inverted patches."""
    k = np.arange(Nz)[None, :]
    A = rng.uniform(0.02, 0.1, (n, 1))
    u = A * np.tanh((k - 0.75 * Nz) / (Nz / 8)) + 2e-3 * rng.standard_normal((n, Nz))
    v = 0.5 * A * np.tanh((k - 0.6 * Nz) / (Nz / 6)) + 2e-3 * rng.standard_normal((n, Nz))
    T = 19.6 + 0.4 * k / Nz + (0.02 if unstable else 0.0) * rng.standard_normal((n, Nz))
    return u.astype(np.float32), v.astype(np.float32), T.astype(np.float32)


def _mpp_thomas(u, v, T, dt, dz, P, ca, hb=None):
    """Independent restatement: scalar loops over faces, Thomas elimination (no LAPACK, no vectorised np.where)."""
    n, Nz = T.shape
    c = dt / dz ** 2
    outs = [np.empty((n, Nz)) for _ in range(3)]
    for i in range(n):
        nu, nuT = np.zeros(Nz + 1), np.zeros(Nz + 1)
        for f in range(Nz + 1):
            lo = lambda a, h: (a[i, f - 1] if f > 0 else (h if h is not None else a[i, 0]))
            hi = lambda a: a[i, f] if f < Nz else a[i, Nz - 1]
            du = hi(u) - lo(u, None if hb is None else hb[0][i])
            dv = hi(v) - lo(v, None if hb is None else hb[1][i])
            dT = hi(T) - lo(T, None if hb is None else hb[2][i])
            den = (du / dz) ** 2 + (dv / dz) ** 2
            num = P["g"] * P["alpha"] * dT / dz
            Ri = num / den if den != 0 else (np.nan if num == 0 else np.copysign(np.inf, num))
            if 1 <= f < Nz:
                nu[f] = P["nu0"] + P["nu_minus"] * (1 - np.tanh((Ri - P["Ric"]) / P["dRi"])) / 2
            nuT[f] = (nu[f] / P["Pr"] if Ri > 0 else 1.0) if ca else nu[f] / P["Pr"]
        for o, (x, kf) in zip(outs, ((u[i], nu), (v[i], nu), (T[i], nuT))):
            x = x.astype(np.float64).copy()
            kk = c * kf
            cp = np.zeros(Nz)
            b0 = 1 + kk[0] + kk[1]
            cp[0] = -kk[1] / b0
            x[0] /= b0
            for r in range(1, Nz):
                a = -kk[r]
                b = 1 + kk[r] + (kk[r + 1] if r < Nz - 1 else 0.0)
                den = b - a * cp[r - 1]
                cp[r] = (-kk[r + 1] if r < Nz - 1 else 0.0) / den
                x[r] = (x[r] - a * x[r - 1]) / den
            for r in range(Nz - 2, -1, -1):
                x[r] -= cp[r] * x[r + 1]
            o[i] = x
        outs[2][i, 0] = T[i, 0]
    return outs


@pytest.mark.parametrize("ca", [False, True])
def test_oracle_mpp_step_matches_scalar_thomas_restatement(ca):
    rng = np.random.default_rng(7)
    u, v, T = _uvT(12, 32, rng)
    hb = np.stack([u[:, 0] - 1e-3, v[:, 0] + 2e-3, T[:, 0] + np.where(np.arange(12) % 2, 0.01, -0.01)]).astype(np.float32)
    for halo in (None, hb):
        got = orc.modified_pacanowski_philander_step(u, v, T, 60.0, 8.0, convective_adjustment=ca, halo_bottom=halo, **MPP)
        want = _mpp_thomas(u.astype(np.float64), v.astype(np.float64), T.astype(np.float64), 60.0, 8.0, MPP, ca, halo)
        for g, w in zip(got, want):
            np.testing.assert_allclose(g, w, rtol=1e-11, atol=1e-13)


def test_oracle_mpp_step_properties():
    rng = np.random.default_rng(8)
    u, v, T = _uvT(6, 32, rng)
    dt, dz = 60.0, 8.0
    uo, vo, To = orc.modified_pacanowski_philander_step(u, v, T, dt, dz, **MPP)
    # `T′[1] = T_bottom` (:94), and the velocity operator conserves column momentum (zero-flux ends: column sums of L are 1)
    assert np.array_equal(To[:, 0], T[:, 0].astype(np.float64))
    np.testing.assert_allclose(uo.sum(1), u.astype(np.float64).sum(1), rtol=1e-12)
    np.testing.assert_allclose(vo.sum(1), v.astype(np.float64).sum(1), rtol=1e-12)
    # M-matrix with unit row sums: maximum principle
    assert (uo.max(1) <= u.max(1) + 1e-12).all() and (uo.min(1) >= u.min(1) - 1e-12).all()
    # ν₋ = ν₀ = 0: nothing diffuses
    P0 = dict(MPP, nu0=0.0, nu_minus=0.0)
    u0, v0, T0 = orc.modified_pacanowski_philander_step(u, v, T, dt, dz, **P0)
    assert np.array_equal(u0, u.astype(np.float64)) and np.array_equal(T0, T.astype(np.float64))
    # constant ν (Riᶜ → +∞ ⇒ tanh_step = 1): one backward-Euler step damps discrete cosine mode m by 1/(1 + c ν λ_m)
    Nz = 32
    Pc = dict(MPP, Ric=1e30, nu0=0.0, nu_minus=2e-2)
    m = 3
    mode = np.cos(np.pi * m * (np.arange(Nz) + 0.5) / Nz)
    uu = np.tile(mode, (2, 1)) * 0.05
    vv = 0.02 + 0 * uu
    TT = np.tile(19.6 + 0.01 * np.arange(Nz), (2, 1))
    u1, _, _ = orc.modified_pacanowski_philander_step(uu, vv, TT, dt, dz, **Pc)
    lam = 4 * np.sin(np.pi * m / (2 * Nz)) ** 2
    np.testing.assert_allclose(u1, uu / (1 + dt / dz ** 2 * 2e-2 * lam), rtol=1e-9, atol=1e-14)
    # convective adjustment: an inverted layer (Ri < 0) is mixed with ν_T = 1 while a stable one keeps ν/Pr
    Tinv = np.tile(19.6 + 0.01 * np.arange(Nz), (1, 1))
    Tinv[0, 20:24] -= 0.05
    ui = 0.05 * np.tanh((np.arange(Nz)[None, :] - 24) / 4.0)
    _, _, Tca = orc.modified_pacanowski_philander_step(ui, 0 * ui, Tinv, dt, dz, convective_adjustment=True, **MPP)
    _, _, Tno = orc.modified_pacanowski_philander_step(ui, 0 * ui, Tinv, dt, dz, convective_adjustment=False, **MPP)
    assert np.abs(Tca - Tinv).max() > 3 * np.abs(Tno - Tinv).max()


def _mpp_params():
    return (MPP["nu0"], MPP["nu_minus"], MPP["dRi"], MPP["Ric"], MPP["Pr"], MPP["alpha"], MPP["g"])


@pytest.mark.gpu
@pytest.mark.parametrize("Nz,n", [(32, 1), (32, 63), (32, 64), (32, 1000), (64, 257), (16, 300), (20, 77)])
@pytest.mark.parametrize("ca", [False, True])
def test_gpu_implicit_diffusion_matches_oracle(Nz, n, ca):
    import colnde
    rng = np.random.default_rng(100 + Nz + n)
    cfg = synthetic.free_convection_problem(1, Nz=Nz, n_save=2).cfg          # any model kind: only Nz is the handle's
    u, v, T = _uvT(n, Nz, rng)
    dt, dz = 60.0, 256.0 / Nz
    hb = np.stack([u[:, 0] - 1e-3, v[:, 0] + 2e-3, T[:, 0] + np.where(np.arange(n) % 2, 0.01, -0.01)]).astype(np.float32)
    with colnde.ColumnNDE(cfg, 1) as nde:
        for halo in (None, hb):
            want = orc.modified_pacanowski_philander_step(u, v, T, dt, dz, convective_adjustment=ca, halo_bottom=halo, **MPP)
            got = nde.implicit_diffusion(u, v, T, dt, dz, _mpp_params(), ca, halo)
            # float32 Thomas vs float64 LAPACK: velocities to 2e-5 of their range, T (|T| ~ 20, changes ~ 1e-2) to 2e-6 relative
            for g, w, tol in zip(got, want, (2e-5, 2e-5, 2e-6)):
                assert np.isfinite(g).all()
                assert np.abs(g - w).max() <= tol * np.abs(w).max(), (np.abs(g - w).max(), np.abs(w).max())
            assert np.array_equal(got[2][:, 0], T[:, 0])                                         # T′[1] = T_bottom, bit for bit
        # nothing to diffuse: bit-identical pass-through
        z = nde.implicit_diffusion(u, v, T, dt, dz, (0.0, 0.0, 1.0, 0.25, 1.0, MPP["alpha"], MPP["g"]), False)
        assert all(np.array_equal(a, b) for a, b in zip(z, (u, v, T)))


@pytest.mark.gpu
def test_gpu_implicit_diffusion_device_twin_in_place_mirror_and_bad_arguments():
    import torch
    import colnde
    from colnde import wind_mixing
    rng = np.random.default_rng(5)
    p = synthetic.wind_mixing_problem(2, n_frames=3)                          # a wind-mixing handle this time
    u, v, T = _uvT(8192 + 17, 32, rng)
    dt, dz = 60.0, 8.0
    want = orc.modified_pacanowski_philander_step(u[:256], v[:256], T[:256], dt, dz, convective_adjustment=True, **MPP)
    with colnde.ColumnNDE(p.cfg, 2) as nde:
        ud, vd, Td = (torch.from_numpy(a).cuda() for a in (u, v, T))
        nde.implicit_diffusion(ud, vd, Td, dt, dz, _mpp_params(), True, out=(ud, vd, Td))      # in place
        torch.cuda.synchronize()
        host = nde.implicit_diffusion(u, v, T, dt, dz, _mpp_params(), True)
        for d, hst, w, tol in zip((ud, vd, Td), host, want, (2e-5, 2e-5, 2e-6)):
            assert np.array_equal(d.cpu().numpy(), hst)                                        # device twin == host entry point
            assert np.abs(hst[:256] - w).max() <= tol * np.abs(w).max()
        # the reference-named mirror, with the reference's dictionary keys and a 1-D column
        pj = {"ν₀": MPP["nu0"], "ν₋": MPP["nu_minus"], "ΔRi": MPP["dRi"], "Riᶜ": MPP["Ric"], "Pr": MPP["Pr"]}
        u1, v1, T1 = wind_mixing.modified_pacanowski_philander_step(nde, u[3], v[3], T[3], dt, dz, pj, {"α": MPP["alpha"], "g": MPP["g"]}, True)
        assert u1.shape == (32,) and np.array_equal(T1, host[2][3]) and np.array_equal(u1, host[0][3])
        with pytest.raises(colnde.ColndeError):
            nde.implicit_diffusion(u, v, T, -1.0, dz, _mpp_params())
        with pytest.raises(colnde.ColndeError):
            nde.implicit_diffusion(u, v, T, dt, dz, (1e-4, 1e-1, 0.0, 0.25, 1.0, 1e-4, 9.81))   # ΔRi = 0
        with pytest.raises(colnde.ColndeError):
            nde.implicit_diffusion(u, v, T, dt, dz, (-1e-4, 1e-1, 1.0, 0.25, 1.0, 1e-4, 9.81))  # negative diffusivity
