"""GPU tests of the host-side mirror (reference-named closures), the device-pointer twins, and size-independent
properties at BASELINE-scale column counts where the oracle would take too long."""
import numpy as np
import pytest

import colnde
from colnde import synthetic
from colnde.flux_compat import ADAM
from colnde.wind_mixing import WindMixingNDE, train_NDE, calculate_loss_scalings
from colnde.free_convection import FreeConvectionNDE, train_neural_differential_equation, compute_neural_network_forcing
from oracle import nde_oracle as O

from tests.test_gpu_parity import _record, SOL_ATOL, LOSS_RTOL

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(np.asarray(a, np.float64) - b) / (np.linalg.norm(b) + 1e-300)


def test_wind_mixing_closures_match_oracle():
    p = synthetic.wind_mixing_problem(5, n_frames=9, weight_divisor=1e2)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    wm = WindMixingNDE(p.cfg, p.x0, p.bcs, truth)
    # NDE(x, p, t) with p = [weights; BCs[i]]  (NDE_training.jl:56-66)
    pvec = np.concatenate([p.weights, p.bcs[2]])
    dx = wm.NDE(p.x0[2], pvec, 0.0)
    assert _rel(dx, O.rhs(p.cfg, p.x0[2:3], p.bcs[2:3], p.weights)[0]) < 1e-6
    # NDE!(dx, x, p, t) mutates dx and returns nothing (training_postprocessing.jl:131-153)
    out = np.zeros(96, np.float32)
    assert wm.NDE_inplace(out, p.x0[2], pvec, 0.0) is None
    assert _rel(out, O.rhs(p.cfg.with_(inplace_variant=True), p.x0[2:3], p.bcs[2:3], p.weights)[0]) < 1e-6
    sols = wm.solve_NDE_nonmutating(p.weights)
    assert sols.shape == (5, 96, 9)                                   # each sols[i] is the reference's 96 x Nt array
    ref = O.solve(p.cfg, p.x0, p.bcs, p.weights)
    assert np.abs(np.transpose(sols, (0, 2, 1)) - ref).max() < SOL_ATOL
    tot, losses, scal = wm.loss_gradient_NDE(p.weights)
    tref, terms_ref = O.loss(p.cfg, ref, truth, wm.loss_scalings)
    assert np.isclose(tot, tref, rtol=LOSS_RTOL) and set(losses) == {"u", "v", "T", "dudz", "dvdz", "dTdz"}
    tot0, losses0, _ = wm.loss_NDE(p.weights)
    assert losses0["dudz"] == 0 and np.isclose(tot0, sum(terms_ref[:3]), rtol=LOSS_RTOL)
    wm.close()


def test_training_fractions_scalings_and_train_loop_reduce_loss():
    p = synthetic.wind_mixing_problem(16, n_frames=9, weight_divisor=1e2)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    fr = dict(T=0.8, dTdz=0.8, profile=0.5)                          # train_NDE.jl:110
    wm = WindMixingNDE(p.cfg, p.x0, p.bcs, truth, training_fractions=fr, weights0=p.weights)
    tot, losses, _ = wm.loss_gradient_NDE(p.weights)
    # the identities of wind_mixing/test/test_training_scaling.jl:17-19 hold for the device-computed terms
    assert np.isclose(losses["T"] / (losses["u"] + losses["v"]), fr["T"] / (1 - fr["T"]), rtol=1e-3)
    assert np.isclose((losses["u"] + losses["v"] + losses["T"]) / (losses["dudz"] + losses["dvdz"] + losses["dTdz"]), 1.0, rtol=1e-3)
    res = train_NDE(wm, p.weights, [ADAM(3e-4)], epochs=1, maxiters=8)
    assert res.history[-1]["total"] < res.history[0]["total"]
    wm.close()


def test_free_convection_mirror_and_training():
    p = synthetic.free_convection_problem(6, Nz=32, n_save=5, substeps=2, t_end=0.02)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    nde = FreeConvectionNDE(p.cfg, p.x0, p.bcs, truth)
    c = p.cfg
    pvec = np.concatenate([p.weights, p.bcs[1], [c.sigma[2], c.sigma[5], c.H, c.tau]]).astype(np.float32)
    assert _rel(nde.dTdt(p.x0[1], pvec), O.rhs(c, p.x0[1:2], p.bcs[1:2], p.weights)[0]) < 2e-5
    assert nde.solve_nde(p.weights).shape == (6, 32, 5)
    l0 = nde.nde_loss(p.weights)
    assert np.isclose(l0, O.loss(c, O.solve(c, p.x0, p.bcs, p.weights), truth, [0, 0, 1, 0, 0, 0])[0], rtol=2e-3)
    theta, hist = train_neural_differential_equation(nde, p.weights, ADAM(1e-3), epochs=6)
    assert hist[-1] < hist[0]
    # `causal_penalty` (training.jl:44,57-58): an extra term of the weights alone, added to value and gradient
    _, g0 = nde.nde_loss_and_grad(p.weights)
    nde.causal_penalty = lambda th: (0.5 * 1e-3 * float(th @ th), 1e-3 * th)
    assert np.isclose(nde.nde_loss(p.weights), l0 + 0.5e-3 * float(p.weights @ p.weights), rtol=1e-6)
    l1, g1 = nde.nde_loss_and_grad(p.weights)
    np.testing.assert_allclose(g1 - g0, 1e-3 * p.weights, rtol=1e-4, atol=1e-9)
    nde.close()


def test_forcing_mirror_shape_and_values():
    cfg, T, top, w = synthetic.inference_problem(8, 6)
    with colnde.ColumnNDE(cfg, 48) as eng:
        out = compute_neural_network_forcing(eng, w, T.reshape(8, 6, 32), top.reshape(8, 6), 1000.0)
    assert out.shape == (8, 6, 32)
    assert _rel(out.reshape(48, 32), O.infer_forcing(cfg, T, top, w, 1000.0)) < 2e-6


def test_device_pointer_twins_match_host_entry_points():
    import torch
    p = synthetic.wind_mixing_problem(40, n_frames=5, weight_divisor=1e2)
    dev = torch.device("cuda", 0)
    with colnde.ColumnNDE(p.cfg, 40) as nde:
        nde.set_problem(p.x0, p.bcs)
        truth_h = nde.forward(p.weights_truth)
        nde.set_problem(p.x0, p.bcs, truth_h)
        sc = [1, 1, 1, 5e-3, 5e-3, 5e-3]
        tot_h, terms_h, grad_h = nde.loss_grad(p.weights, sc)
        x0, bcs, w = (torch.from_numpy(a).to(dev) for a in (p.x0, p.bcs, p.weights))
        nde.set_problem(x0, bcs, torch.from_numpy(truth_h).to(dev))
        sol_d = nde.forward(w)
        out = nde.loss_grad(w, sc)
        dx_d = nde.rhs(x0, w, bcs, 0.0)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(sol_d.cpu().numpy(), nde.forward(p.weights))
        np.testing.assert_array_equal(out[:nde.n_params].cpu().numpy(), grad_h)     # deterministic reduction order
        assert float(out[nde.n_params + 6]) == tot_h
        np.testing.assert_array_equal(dx_d.cpu().numpy(), nde.rhs(p.x0, p.weights, p.bcs, 0.0))


def test_gradient_is_run_to_run_bit_identical():
    p = synthetic.wind_mixing_problem(100, n_frames=9, weight_divisor=1e2)
    with colnde.ColumnNDE(p.cfg, 100) as nde:
        nde.set_problem(p.x0, p.bcs)
        nde.set_problem(p.x0, p.bcs, nde.forward(p.weights_truth))
        a = nde.loss_grad(p.weights, [1, 1, 1, 5e-3, 5e-3, 5e-3])
        b = nde.loss_grad(p.weights, [1, 1, 1, 5e-3, 5e-3, 5e-3])
    assert a[0] == b[0]
    np.testing.assert_array_equal(a[2], b[2])


# ---- size-independent properties at BASELINE scale (configs[1]: 4096 columns x 32 levels) -----------------
def test_inertial_oscillation_4096_columns():
    """Zero nets, nu = 0, zero boundary flux: (U, V) rotates by -f tau t, T constant — for every one of 4096 columns."""
    p = synthetic.wind_mixing_problem(4096, n_frames=33, substeps=4, nu0=0.0, nu_minus=0.0)
    cfg = p.cfg
    s0 = [-cfg.mu[3 + k] / cfg.sigma[3 + k] for k in range(3)]
    bcs = np.tile(np.array([s0[0], s0[0], s0[1], s0[1], s0[2], s0[2]], np.float32), (4096, 1))
    with colnde.ColumnNDE(cfg, 4096) as nde:
        nde.set_problem(p.x0, bcs)
        sol = nde.forward(np.zeros(cfg.n_params, np.float32))
    Nz = cfg.Nz
    th = cfg.f * cfg.tau * cfg.save_times[-1]
    U0, V0 = cfg.sigma[0] * p.x0[:, :Nz], cfg.sigma[1] * p.x0[:, Nz:2 * Nz]
    np.testing.assert_allclose(cfg.sigma[0] * sol[:, -1, :Nz], U0 * np.cos(th) + V0 * np.sin(th), atol=2e-6)
    np.testing.assert_allclose(cfg.sigma[1] * sol[:, -1, Nz:2 * Nz], -U0 * np.sin(th) + V0 * np.cos(th), atol=2e-6)
    np.testing.assert_array_equal(sol[:, -1, 2 * Nz:], p.x0[:, 2 * Nz:])


def test_heat_conservation_and_shard_additivity_4096_columns():
    p = synthetic.wind_mixing_problem(4096, n_frames=9, weight_divisor=1e2)
    cfg = p.cfg
    s0 = [-cfg.mu[3 + k] / cfg.sigma[3 + k] for k in range(3)]
    bcs = p.bcs.copy()
    bcs[:, 4] = s0[2]
    bcs[:, 5] = s0[2]                                   # zero heat flux at both boundaries
    sc = [1, 1, 1, 5e-3, 5e-3, 5e-3]
    with colnde.ColumnNDE(cfg, 4096) as nde:
        nde.set_problem(p.x0, bcs)
        sol = nde.forward(p.weights)
        # column heat content is conserved by the telescoping flux divergence
        assert np.abs(sol[:, -1, 64:].sum(axis=1) - p.x0[:, 64:].sum(axis=1)).max() < 2e-4
        truth = nde.forward(p.weights_truth)
        nde.set_problem(p.x0, bcs, truth)
        tot, terms, grad = nde.loss_grad(p.weights, sc)
    # two shards normalised by the global count add up to the global result (the multi-GPU contract)
    parts = []
    for lo, hi in ((0, 1500), (1500, 4096)):
        with colnde.ColumnNDE(cfg, hi - lo) as sh:
            sh.set_global_columns(4096)
            sh.set_problem(p.x0[lo:hi], bcs[lo:hi], truth[lo:hi])
            parts.append(sh.loss_grad(p.weights, sc))
    assert np.isclose(parts[0][0] + parts[1][0], tot, rtol=1e-4)
    assert _rel(parts[0][2] + parts[1][2], grad.astype(np.float64)) < 1e-4


# ---- BASELINE configs[3] and configs[4] at (near) full size: size-independent properties --------------------------
def test_config5_inference_65536_columns():
    """double_gyre_nn forcing at 256 x 256 columns x 32 levels (BASELINE configs[4]); oracle on a strided sample,
    linearity of the forcing in the surface flux, and column-integral identity sum_k dz * forcing = -(top - 0)."""
    cfg, T, top, w = synthetic.inference_problem(256, 256)
    with colnde.ColumnNDE(cfg, T.shape[0]) as nde:
        out = nde.infer_forcing(w, T, top, 1000.0)
        out2 = nde.infer_forcing(w, T, 2.0 * top, 1000.0)
    idx = np.arange(0, T.shape[0], 997)
    _record("config5_65536", rel=_rel(out[idx], O.infer_forcing(cfg, T[idx], top[idx], w, 1000.0)))
    assert _rel(out[idx], O.infer_forcing(cfg, T[idx], top[idx], w, 1000.0)) < 2e-6
    dz = 1000.0 / 32
    np.testing.assert_allclose(out.sum(axis=1) * dz, -top, rtol=2e-4, atol=1e-9)          # telescoping flux divergence
    # only the surface cell sees the top flux
    np.testing.assert_array_equal(out[:, :-1], out2[:, :-1])
    np.testing.assert_allclose((out2 - out)[:, -1] * dz, -top, rtol=1e-4, atol=1e-10)


def test_config4_free_convection_64_levels_properties():
    """Free convection, 64 levels, Dense(64,256) -> Dense(256,256) -> Dense(256,63) relu (BASELINE configs[3] shape, 2048
    columns here): zero-weight known answer, shard additivity of loss/gradient, run-to-run determinism."""
    p = synthetic.free_convection_problem(2048, Nz=64, n_save=9, substeps=2, t_end=0.05)
    cfg = p.cfg
    sc = [0, 0, 1.0, 0, 0, 0]
    with colnde.ColumnNDE(cfg, 2048) as nde:
        nde.set_problem(p.x0, p.bcs)
        sol0 = nde.forward(np.zeros(cfg.n_params, np.float32))
        # zero weights: only the two boundary cells change, linearly in time (exact for any RK)
        C = cfg.sigma[5] / cfg.sigma[2] * cfg.tau / cfg.H
        t = np.asarray(cfg.save_times, np.float32)
        np.testing.assert_array_equal(sol0[:, :, 1:-1], np.repeat(p.x0[:, None, 1:-1], 9, 1))
        np.testing.assert_allclose(sol0[:, :, -1], p.x0[:, -1:] - C * 64 * p.bcs[:, 1:2] * t[None], rtol=1e-5, atol=2e-6)
        truth = nde.forward(p.weights_truth)
        nde.set_problem(p.x0, p.bcs, truth)
        tot, terms, grad = nde.loss_grad(p.weights, sc)
        tot2, _, grad2 = nde.loss_grad(p.weights, sc)
    assert tot == tot2
    np.testing.assert_array_equal(grad, grad2)
    parts = []
    for lo, hi in ((0, 700), (700, 2048)):
        with colnde.ColumnNDE(cfg, hi - lo) as sh:
            sh.set_global_columns(2048)
            sh.set_problem(p.x0[lo:hi], p.bcs[lo:hi], truth[lo:hi])
            parts.append(sh.loss_grad(p.weights, sc))
    assert np.isclose(parts[0][0] + parts[1][0], tot, rtol=1e-4)
    assert _rel(parts[0][2] + parts[1][2], grad.astype(np.float64)) < 1e-4


def test_plan_reports_engine_and_tapes():
    from colnde.nde import ENGINE_REGTILE, ENGINE_TILE16
    p = synthetic.wind_mixing_problem(40, n_frames=3)
    for eng, key in ((ENGINE_REGTILE, "z1_taped"), (ENGINE_TILE16, None)):
        with colnde.ColumnNDE(p.cfg, 40, engine=eng) as nde:
            nde.set_problem(p.x0, p.bcs)
            truth = nde.forward(p.weights_truth)
            nde.set_problem(p.x0, p.bcs, truth)
            assert nde.plan()["n_blocks"] == 0                      # nothing planned before the first gradient
            nde.loss_grad(p.weights, [1, 1, 1, 5e-3, 5e-3, 5e-3])
            pl = nde.plan()
            assert pl["engine"] == eng
            if eng == ENGINE_REGTILE:
                assert pl["n_blocks"] == 1 and pl["block_columns"] == 64 and pl["z1_taped"]
            else:
                assert pl["dw_taped"] and pl["dw_slices"] >= 1           # tile16 default: taped deltas + split-K dW GEMM


def test_two_engines_agree_at_the_bench_horizon_4096_columns():
    """Size-independent check at BASELINE size: the two independent kernel families (register-resident regtile, LDS-staged
    tile16) integrate 4,096 columns over the full 2-day horizon (576 RK4 steps) and back-propagate to the same loss terms and
    gradient; the first columns are also compared with the float32 C port of the oracle."""
    from colnde.nde import ENGINE_REGTILE, ENGINE_TILE16
    from oracle import cref
    p = synthetic.wind_mixing_problem(4096, weight_divisor=1e2)
    sc = [1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3]
    res = {}
    for eng in (ENGINE_REGTILE, ENGINE_TILE16):
        with colnde.ColumnNDE(p.cfg, 4096, engine=eng) as nde:
            nde.set_problem(p.x0, p.bcs)
            truth = nde.forward(p.weights_truth)
            nde.set_problem(p.x0, p.bcs, truth)
            res[eng] = (truth, nde.loss_grad(p.weights, sc))
    ta, (tot_a, terms_a, ga) = res[ENGINE_REGTILE]
    tb, (tot_b, terms_b, gb) = res[ENGINE_TILE16]
    assert np.abs(ta - tb).max() < 5e-4                                # 576 steps of float32 round-off on O(1) profiles
    np.testing.assert_allclose(terms_a, terms_b, rtol=2e-3)
    assert np.linalg.norm(ga - gb) <= 2e-2 * np.linalg.norm(gb)
    ref = cref.forward(p.cfg, p.x0[:8], p.bcs[:8], p.weights_truth, n_threads=4)
    assert np.abs(ta[:8] - ref).max() < 5e-4


def test_non_finite_inputs_are_reported_not_returned():
    """A NaN in the initial state poisons the solve: the host entry points say so instead of handing back NaN losses with rc 0."""
    p = synthetic.wind_mixing_problem(9, n_frames=5, weight_divisor=1e2)
    x0 = p.x0.copy()
    x0[3, 40] = np.nan
    with colnde.ColumnNDE(p.cfg, 9) as nde:
        nde.set_problem(x0, p.bcs, np.zeros((9, 5, 96), np.float32))
        with pytest.raises(colnde.ColndeError, match="not finite"):
            nde.loss(p.weights, [1] * 6)
        with pytest.raises(colnde.ColndeError, match="not finite"):
            nde.loss_grad(p.weights, [1] * 6)


def test_two_handles_side_by_side_do_not_disturb_each_other():
    """Two handles of different models alive at once (own tapes, own slabs, one device): each gives what it gives alone."""
    pw = synthetic.wind_mixing_problem(40, n_frames=5, weight_divisor=1e2)
    pf = synthetic.free_convection_problem(19, Nz=32, n_save=5, substeps=2, t_end=0.01)
    scw, scf = [1, 1, 1, 5e-3, 5e-3, 5e-3], [0, 0, 1, 0, 0, 0]

    def alone(p, sc):
        with colnde.ColumnNDE(p.cfg, p.n_columns) as h:
            h.set_problem(p.x0, p.bcs)
            t = h.forward(p.weights_truth)
            h.set_problem(p.x0, p.bcs, t)
            return t, h.loss_grad(p.weights, sc)
    tw, gw = alone(pw, scw)
    tf, gf = alone(pf, scf)
    with colnde.ColumnNDE(pw.cfg, 40) as a, colnde.ColumnNDE(pf.cfg, 19) as b:
        a.set_problem(pw.x0, pw.bcs, tw)
        b.set_problem(pf.x0, pf.bcs, tf)
        ga1 = a.loss_grad(pw.weights, scw)
        gb1 = b.loss_grad(pf.weights, scf)
        ga2 = a.loss_grad(pw.weights, scw)
    for got, ref in ((ga1, gw), (ga2, gw), (gb1, gf)):
        assert got[0] == ref[0]
        np.testing.assert_array_equal(got[2], ref[2])


def test_config4_shard_size_16384_columns_known_answer_and_conservation():
    """One GPU's shard of BASELINE config 4 at full width (16,384 columns x 64 levels, 64-256-256-63): more tiles than CUs, so the
    taped adjoint runs two workgroups per CU and the forward streams its weights from L2.  Size-independent checks: the zero-weight
    known answer for every column (exact for any RK), the gradient of a loss that is exactly zero is exactly zero, determinism, and
    heat conservation with non-zero weights (the flux divergence telescopes: only the boundary fluxes change a column's heat)."""
    n = 16384
    p = synthetic.free_convection_problem(n, Nz=64, n_save=5, substeps=2, t_end=0.02)
    cfg = p.cfg
    sc = [0, 0, 1.0, 0, 0, 0]
    C = cfg.sigma[5] / cfg.sigma[2] * cfg.tau / cfg.H
    t = np.asarray(cfg.save_times, np.float32)
    with colnde.ColumnNDE(cfg, n) as nde:
        nde.set_problem(p.x0, p.bcs)
        sol0 = nde.forward(np.zeros(cfg.n_params, np.float32))
        np.testing.assert_array_equal(sol0[:, :, 1:-1], np.repeat(p.x0[:, None, 1:-1], 5, 1))
        np.testing.assert_allclose(sol0[:, :, -1], p.x0[:, -1:] - C * 64 * p.bcs[:, 1:2] * t[None], rtol=1e-5, atol=2e-6)
        sol = nde.forward(p.weights)
        heat = sol.sum(axis=2) / 64.0                                    # column mean of T-hat
        expect = heat[:, :1] + C * (p.bcs[:, 0:1] - p.bcs[:, 1:2]) * t[None]
        np.testing.assert_allclose(heat, expect, rtol=0, atol=5e-5)
        nde.set_problem(p.x0, p.bcs, sol)                                # truth = the model's own trajectory: loss and gradient vanish
        tot, terms, grad = nde.loss_grad(p.weights, sc)
        assert tot == 0.0 and not grad.any()
        truth = nde.forward(p.weights_truth)
        nde.set_problem(p.x0, p.bcs, truth)
        a = nde.loss_grad(p.weights, sc)
        b = nde.loss_grad(p.weights, sc)
        assert a[0] == b[0] and a[0] > 0 and np.array_equal(a[2], b[2]) and np.isfinite(a[2]).all()
        assert nde.plan()["dw_taped"]
