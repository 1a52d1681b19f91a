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
