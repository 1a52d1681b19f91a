"""Parity at the sizes bench.py TIMES (VERDICT r3 task 4): the headline launch — 32,768 columns x 576 RK4 steps on the regtile engine, one
tape block of ~130 GB, 256 workgroups — and BASELINE configs[3]'s ConvectiveAdjustmentNDE half at shard size (16,384 columns x 64 levels,
RKC2, time-segmented tapes).  The oracle cannot run these sizes in seconds, so: a strided sample of the forward solve against the float32
C port (oracle/colnde_ref.c), and the size-independent properties of the path — shard additivity (the mean over simulations of
NDE_training.jl:312-317 is a sum over shards), run-to-run bit identity, replica invariance, heat conservation.

Reference shapes: wind_mixing/train_NDE.jl (2-day suite: 289 frames), free_convection/src/convective_adjustment_nde.jl:33-48 with the network of
train_free_convection_nde.jl:119-121 on the 129-point axis."""
import numpy as np
import pytest

import colnde
from colnde import synthetic
from colnde.nde import ENGINE_FC32, ENGINE_REGTILE

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300)


@pytest.mark.parametrize("ma", ["bf16x3_exact", "f32_mfma"])
def test_headline_launch_32768_columns_576_steps(ma):
    """bench.py's own launch, checked: (1) 64 columns strided over the whole batch (every 512th: all 256 workgroups' first wave tiles... and
    offsets 0..63 inside them) against the float32 C port over the full 576-step horizon — weights/1e2 so the nets matter; (2) loss and gradient
    of the whole batch equal the sum over two shards of 20,000 + 12,768 columns, each normalised by the global count (what N ranks all-reduce);
    (3) two calls give bit-identical results; (4) the plan is the bench's: regtile, one block of 32,768 columns, Z1 taped."""
    import torch
    from oracle import cref
    n = 32768
    dev = torch.device("cuda", 0)
    p = synthetic.wind_mixing_problem(n, weight_divisor=1e2)
    assert p.cfg.n_steps == 576
    sc = [1.0, 1.0, 1.0, 5e-3, 5e-3, 5e-3]
    x0, bcs = torch.from_numpy(p.x0).to(dev), torch.from_numpy(p.bcs).to(dev)
    w, wt = torch.from_numpy(p.weights).to(dev), torch.from_numpy(p.weights_truth).to(dev)
    with colnde.ColumnNDE(p.cfg, n, engine=ENGINE_REGTILE, matrix_arithmetic=ma) as nde:
        nde.set_problem(x0, bcs)
        truth = nde.forward(wt)
        sol = nde.forward(w)
        idx = np.arange(64) * 512 + np.arange(64)                          # column 513 k: tile offsets 0..63 across the batch
        ref = cref.forward(p.cfg, p.x0[idx], p.bcs[idx], p.weights, n_threads=8)
        got = sol[torch.from_numpy(idx).to(dev)].cpu().numpy()
        err = np.abs(got - ref).max()
        assert np.isfinite(got).all() and err < 5e-4, err                  # 576 steps of float32 round-off on O(1) profiles (tests/test_gpu_mirror.py)
        nde.set_problem(x0, bcs, truth)
        a = nde.loss_grad(w, sc).cpu().numpy()
        b = nde.loss_grad(w, sc).cpu().numpy()
        plan = nde.plan()
        assert np.array_equal(a, b) and np.isfinite(a).all() and a[p.cfg.n_params + 6] > 0
        assert plan["engine"] == ENGINE_REGTILE and plan["n_blocks"] == 1 and plan["block_columns"] == n and plan["z1_taped"]
        assert plan["bf16x3_forward"] == plan["bf16x3_adjoint"] == plan["bf16x3_dw"] == (ma == "bf16x3_exact")
    del sol
    torch.cuda.empty_cache()
    parts = np.zeros_like(a, dtype=np.float64)
    for lo, hi in ((0, 20000), (20000, n)):
        with colnde.ColumnNDE(p.cfg, hi - lo, engine=ENGINE_REGTILE, matrix_arithmetic=ma) as sh:
            sh.set_global_columns(n)
            sh.set_problem(x0[lo:hi].contiguous(), bcs[lo:hi].contiguous(), truth[lo:hi].contiguous())
            parts += sh.loss_grad(w, sc).cpu().numpy().astype(np.float64)
        torch.cuda.empty_cache()
    np_ = p.cfg.n_params
    np.testing.assert_allclose(parts[np_:np_ + 7], a[np_:np_ + 7], rtol=2e-5)
    assert _rel(parts[:np_], a[:np_]) < 2e-5                                # the same columns in other tiles and slab rows: summation order only


@pytest.mark.parametrize("ma", ["bf16x3_exact", "f32_mfma"])
def test_headline_gradient_against_float64_oracle_512_replicas_of_64_columns(ma):
    """VERDICT r4 task 2a — the headline launch's GRADIENT against the oracle, not against itself.  The loss is a mean over simulations
    (NDE_training.jl:303-317), so 512 replicas of 64 distinct columns have exactly the six loss terms and the gradient of the 64: the GPU runs the
    bench's own launch (32,768 columns x 576 RK4 steps, regtile, one ~130 GB tape block, 1,024 wave tiles, the dW1 streaming GEMM over 75 M
    records) and the float64 oracle evaluates the 64.  Weights/1e2 so the nets matter; tolerances are the long-horizon ones of
    tests/test_gpu_parity.py (LONG_*[1e2]: 2e-5 on the loss and on each term, 4e-5 relative L2 on the gradient).  Also: every replica's
    trajectory is bit-identical to the first's (same arithmetic in every tile, workgroup and XCD)."""
    import torch
    from oracle import nde_oracle as O
    from tests.test_gpu_parity import LONG_SOL_ATOL, LONG_LOSS_RTOL, LONG_TERMS_RTOL, LONG_GRAD_REL, _record
    n, reps = 32768, 512
    dev = torch.device("cuda", 0)
    q = synthetic.wind_mixing_problem(64, n_frames=289, weight_divisor=1e2)
    assert q.cfg.n_steps == 576
    truth64 = O.solve(q.cfg, q.x0, q.bcs, q.weights_truth).astype(np.float32)
    sc = O.default_loss_scalings(q.cfg)
    tot, terms, g, sol = O.loss_and_grad(q.cfg, q.x0, q.bcs, q.weights, truth64, sc)
    x0 = torch.from_numpy(np.tile(q.x0, (reps, 1))).to(dev)
    bcs = torch.from_numpy(np.tile(q.bcs, (reps, 1))).to(dev)
    truth = torch.from_numpy(np.tile(truth64, (reps, 1, 1))).to(dev)
    w = torch.from_numpy(q.weights).to(dev)
    with colnde.ColumnNDE(q.cfg, n, engine=ENGINE_REGTILE, matrix_arithmetic=ma) as nde:
        nde.set_problem(x0, bcs, truth)
        s = nde.forward(w).view(reps, 64, q.cfg.n_save, 96)
        assert bool((s == s[:1]).all())                                     # replica invariance, bit for bit
        first = s[0].cpu().numpy()
        del s
        torch.cuda.empty_cache()
        res = nde.loss_grad(w, list(sc)).cpu().numpy()
        plan = nde.plan()
    assert plan["engine"] == ENGINE_REGTILE and plan["n_blocks"] == 1 and plan["block_columns"] == n and plan["z1_taped"]
    assert plan["bf16x3_forward"] == plan["bf16x3_adjoint"] == plan["bf16x3_dw"] == (ma == "bf16x3_exact")
    np_ = q.cfg.n_params
    terms_g, tot_g, grad_g = res[np_:np_ + 6], res[np_ + 6], res[:np_]
    _record("bench_size/headline_32768x576_gradient/%s" % ma, sol_abs=np.abs(first - sol).max(), loss_rel=abs(tot_g - tot) / tot,
            terms_rel=np.abs(terms_g / terms - 1).max(), grad_rel=_rel(grad_g, g))
    assert np.abs(first - sol).max() < LONG_SOL_ATOL
    np.testing.assert_allclose(terms_g, terms, rtol=LONG_TERMS_RTOL[1e2], atol=0)
    assert np.isclose(tot_g, tot, rtol=LONG_LOSS_RTOL[1e2], atol=0)
    assert _rel(grad_g, g) < LONG_GRAD_REL[1e2]
    # each net's layer blocks separately (a wrong dW1 slab row or a wrong W3 block hides in the whole vector's norm)
    third = np_ // 3
    for net in range(3):
        lo = net * third
        for a, b in ((0, 4800), (4800, 4850), (4850, 5850), (5850, 5870), (5870, 6490), (6490, 6521)):      # W1 b1 W2 b2 W3 b3 (Flux.destructure order)
            assert _rel(grad_g[lo + a:lo + b], g[lo + a:lo + b]) < 4 * LONG_GRAD_REL[1e2], (net, a, b)


def test_conv_adj_nde_rkc2_16384x64_time_segments():
    """configs[3]'s stiff half as bench.py times it: ConvectiveAdjustmentNDE (K = 10), 16,384 columns x 64 levels, the 129-point axis, RKC2 with
    the automatic stage count, tapes cut into time segments.  The batch is 256 replicas of 64 distinct columns (one of them with an inverted
    layer, so the min(0, K dT/dz) switch is live), which makes three things checkable at full size:
      * replica invariance: every replica's trajectory is bit-identical to the first's (same arithmetic in every tile and workgroup);
      * the segmented full-size loss and gradient equal those of ONE 64-column handle that holds the whole axis in a single tape pass
        (the mean over simulations of 256 identical replicas is the mean over one) — to summation order;
      * column heat changes only through the boundary fluxes (both Dᶜ terms telescope: convective_adjustment_nde.jl:43-47);
    and the plan reports time segments and the one-switch-pattern (approximate) RKC2 gradient."""
    import torch
    n, reps = 16384, 256
    dev = torch.device("cuda", 0)
    p = synthetic.free_convection_problem(64, Nz=64, convective_adjustment=True)       # n_save = 129, t in [0, 1]
    cfg = p.cfg.with_(stepper="rkc2", substeps=4)
    x64 = p.x0.copy()
    x64[:8, 20:44] = x64[:8, 20:44][:, ::-1]                                            # inverted layers in 8 of the 64 columns
    sc = [0, 0, 1.0, 0, 0, 0]
    with colnde.ColumnNDE(cfg, 64) as one:
        assert one.engine == ENGINE_FC32
        one.set_problem(x64, p.bcs)
        truth64 = one.forward(p.weights_truth)
        sol64 = one.forward(p.weights)
        one.set_problem(x64, p.bcs, truth64)
        tot1, _, g1 = one.loss_grad(p.weights, sc)
        plan1 = one.plan()
    assert plan1["time_segments"] == 0 and plan1["approximate_gradient"] and np.isfinite(g1).all() and tot1 > 0
    x0 = torch.from_numpy(np.tile(x64, (reps, 1))).to(dev)
    bcs = torch.from_numpy(np.tile(p.bcs, (reps, 1))).to(dev)
    truth = torch.from_numpy(np.tile(truth64, (reps, 1, 1))).to(dev)
    w = torch.from_numpy(p.weights).to(dev)
    with colnde.ColumnNDE(cfg, n) as nde:
        nde.set_problem(x0, bcs, truth)
        sol = nde.forward(w)
        s = sol.view(reps, 64, cfg.n_save, 64)
        assert bool((s == s[:1]).all())                                                  # replica invariance, bit for bit
        first = s[0].cpu().numpy()
        out = nde.loss_grad(w, sc)
        out2 = nde.loss_grad(w, sc)
        plan = nde.plan()
        res = out.cpu().numpy()
        assert bool((out == out2).all())
    # (the 64-column handle runs 16-column tiles, the shard 32-column tiles: two MFMA shapes, float32 round-off apart — tests/test_gpu_fc.py)
    assert np.abs(first - sol64).max() < 2e-4
    assert plan["engine"] == ENGINE_FC32 and plan["time_segments"] >= 2 and plan["n_blocks"] == 1 and plan["approximate_gradient"]
    np_ = cfg.n_params
    assert np.isclose(res[np_ + 6], tot1, rtol=2e-4)
    assert _rel(res[:np_], g1) < 2e-3                                                    # relu kinks and the switch: tests/test_gpu_fc.py's fc32-vs-tile16 bound
    C = cfg.sigma[5] / cfg.sigma[2] * cfg.tau / cfg.H
    t = np.asarray(cfg.save_times, np.float64)
    heat = first.astype(np.float64).sum(axis=2) / 64.0
    expect = heat[:, :1] + C * (p.bcs[:, 0:1].astype(np.float64) - p.bcs[:, 1:2]) * t[None]
    np.testing.assert_allclose(heat, expect, rtol=0, atol=2e-4)


@pytest.mark.parametrize("ma", ["bf16x3_exact", "f32_mfma"])
def test_free_convection_rk4_16384x64_the_shard_bench_times(ma):
    """BASELINE configs[3]'s plain half exactly as bench.py times it: FreeConvectionNDE, 16,384 columns x 64 levels x 129 save points x 4 RK4 sub-steps
    (one tape pass of ~155 GB, 512 workgroups of 32 columns; under the default arithmetic fcs_forward_kernel / fcs_adjoint_kernel / dw_gemm_split_kernel,
    under f32_mfma their f32 twins), 256 replicas of 64 distinct columns:
      * replica invariance of the trajectories, bit for bit; two calls give bit-identical gradients;
      * the 64 distinct columns' trajectory, loss and gradient against the float64 oracle (free_convection/src/free_convection_nde.jl:29-38,
        training.jl:55-62) at the fc32 tolerances of tests/test_gpu_parity.py;
      * the plan: fc32, one block, no time segments, exact discrete adjoint, kernels on the arithmetic asked for;
      * column heat changes only through the boundary fluxes."""
    import torch
    from oracle import nde_oracle as O
    from tests.test_gpu_parity import FC_SOL_ATOL, FC_LOSS_RTOL, FC_GRAD_REL
    n, reps = 16384, 256
    dev = torch.device("cuda", 0)
    p = synthetic.free_convection_problem(64, Nz=64)                                    # n_save = 129, t in [0, 1], 4 sub-steps
    cfg = p.cfg
    sc = np.array([0, 0, 1.0, 0, 0, 0])
    truth64 = O.solve(cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    tot, terms, g, sol = O.loss_and_grad(cfg, p.x0, p.bcs, p.weights, truth64, sc)
    # over the full 512-step axis float32 itself stands 1.5e-4 (trajectory), 7e-5 (loss), 1.4e-4 (gradient) from float64: the float32 oracle sets the scale
    tot32, _, g32, sol32 = O.loss_and_grad(cfg, p.x0, p.bcs, p.weights, truth64, sc, dtype=np.float32)
    e32 = (np.abs(sol32 - sol).max(), abs(tot32 - tot) / tot, _rel(g32, g))
    x0 = torch.from_numpy(np.tile(p.x0, (reps, 1))).to(dev)
    bcs = torch.from_numpy(np.tile(p.bcs, (reps, 1))).to(dev)
    truth = torch.from_numpy(np.tile(truth64, (reps, 1, 1))).to(dev)
    w = torch.from_numpy(p.weights).to(dev)
    with colnde.ColumnNDE(cfg, n, matrix_arithmetic=ma) as nde:
        nde.set_problem(x0, bcs, truth)
        s = nde.forward(w).view(reps, 64, cfg.n_save, 64)
        assert bool((s == s[:1]).all())
        first = s[0].cpu().numpy()
        out = nde.loss_grad(w, list(sc))
        out2 = nde.loss_grad(w, list(sc))
        plan = nde.plan()
        res = out.cpu().numpy()
        assert bool((out == out2).all())
    split = ma == "bf16x3_exact"
    assert plan["engine"] == ENGINE_FC32 and plan["n_blocks"] == 1 and plan["time_segments"] == 0 and not plan["approximate_gradient"]
    assert (plan["bf16x3_forward"], plan["bf16x3_adjoint"], plan["bf16x3_dw"]) == (split, split, split)
    np_ = cfg.n_params
    assert np.abs(first - sol).max() < 2 * e32[0] + 0.25 * FC_SOL_ATOL
    assert abs(res[np_ + 6] - tot) / tot < 4 * e32[1] + 0.1 * FC_LOSS_RTOL
    assert _rel(res[:np_], g) < 4 * e32[2] + 0.25 * FC_GRAD_REL
    from tests.test_gpu_parity import _record
    _record("bench_size/free_convection_16384x64/%s" % ma, sol_abs=np.abs(first - sol).max(), loss_rel=abs(res[np_ + 6] - tot) / tot, grad_rel=_rel(res[:np_], g),
            sol_abs_oracle32_vs_64=e32[0], loss_rel_oracle32_vs_64=e32[1], grad_rel_oracle32_vs_64=e32[2])
    C = cfg.sigma[5] / cfg.sigma[2] * cfg.tau / cfg.H
    t = np.asarray(cfg.save_times, np.float64)
    heat = first.astype(np.float64).sum(axis=2) / 64.0
    expect = heat[:, :1] + C * (p.bcs[:, 0:1].astype(np.float64) - p.bcs[:, 1:2]) * t[None]
    np.testing.assert_allclose(heat, expect, rtol=0, atol=2e-4)
