"""The 32-column free-convection engine (csrc/engine_fc.hip, COLNDE_ENGINE_FC32: v_mfma_f32_32x32x2_f32 tiles with compile-time shapes for
FreeConvectionNDE with the reference's network, free_convection/train_free_convection_nde.jl:119-121) through the C ABI: against the
float64 oracle at tile16's free-convection tolerances, against tile16 itself (an independent kernel family behind the same dW GEMM) an
order tighter, bit-reproducible, on ragged tiles, in column blocks, and on config 4's 129-point axis."""
import numpy as np
import pytest

import colnde
from colnde import synthetic
from colnde.nde import ENGINE_FC32, ENGINE_TILE16
from oracle import nde_oracle as O
from tests.test_gpu_parity import _arith, _record, _rel, FC_SOL_ATOL, FC_LOSS_RTOL, FC_GRAD_REL

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cw", [16, 32])
@pytest.mark.parametrize("Nz,ncol", [(32, 1), (32, 19), (32, 64), (32, 97), (64, 5), (64, 32), (64, 45), (64, 130)])
def test_fc32_against_oracle_and_tile16(Nz, ncol, cw, monkeypatch):
    monkeypatch.setenv("COLNDE_FC_CW", str(cw))       # both tile widths at every size (default: 16 columns up to 4,096, 32 above)
    p = synthetic.free_convection_problem(ncol, Nz=Nz, n_save=5, substeps=2, t_end=0.01)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = O.default_loss_scalings(p.cfg)
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    res = {}
    for eng in (0, ENGINE_TILE16):
        with colnde.ColumnNDE(p.cfg, ncol, engine=eng) as nde:
            assert nde.engine == (ENGINE_FC32 if eng == 0 else ENGINE_TILE16)        # AUTO picks fc32 for this shape
            nde.set_problem(p.x0, p.bcs, truth)
            sol_g = nde.forward(p.weights)
            tot_l, _ = nde.loss(p.weights, sc)
            tot_g, terms_g, grad_g = nde.loss_grad(p.weights, sc)
            tot_2, _, grad_2 = nde.loss_grad(p.weights, sc)
            plan = nde.plan()
        assert tot_2 == tot_g and np.array_equal(grad_2, grad_g)                      # fixed-order sums: bit-reproducible
        assert np.isclose(tot_l, tot_g, rtol=1e-5)
        res[eng] = (sol_g, tot_g, grad_g)
        if eng == 0:
            assert plan["engine"] == ENGINE_FC32 and plan["dw_taped"] and plan["n_blocks"] == 1 and plan["block_columns"] == (ncol + 31) // 32 * 32
    sol_g, tot_g, grad_g = res[0]
    _record("fc32/cw%d/%d/%d" % (cw, Nz, ncol), sol_abs=np.abs(sol_g - sol).max(), loss_rel=abs(tot_g - tot) / abs(tot), grad_rel=_rel(grad_g, g),
            sol_abs_vs_tile16=np.abs(sol_g - res[ENGINE_TILE16][0]).max(), grad_rel_vs_tile16=_rel(grad_g, res[ENGINE_TILE16][2].astype(np.float64)))
    assert np.abs(sol_g - sol).max() < FC_SOL_ATOL
    assert np.isclose(tot_g, tot, rtol=FC_LOSS_RTOL)
    assert _rel(grad_g, g) < FC_GRAD_REL
    # the block structure of the gradient (Flux.destructure: W1, b1, W2, b2, W3, b3) — every block on its own, biases included
    H, off = 4 * Nz, 0
    for n in (Nz * H, H, H * H, H, H * (Nz - 1), Nz - 1):
        assert _rel(grad_g[off:off + n], g[off:off + n]) < 2 * FC_GRAD_REL, (off, n)
        off += n
    assert off == p.cfg.n_params
    assert np.abs(sol_g - res[ENGINE_TILE16][0]).max() < 0.25 * FC_SOL_ATOL
    assert _rel(grad_g, res[ENGINE_TILE16][2].astype(np.float64)) < 0.25 * FC_GRAD_REL


@pytest.mark.parametrize("cw", [16, 32])
@pytest.mark.parametrize("Nz", [32, 64])
def test_fc32_column_blocks_are_additive(Nz, cw, monkeypatch):
    """COLNDE_FC_BLOCK=32: 75 columns run forward -> adjoint -> dW GEMM in three passes through tapes sized for one 32-column tile."""
    p = synthetic.free_convection_problem(75, Nz=Nz, n_save=5, substeps=2, t_end=0.01)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = O.default_loss_scalings(p.cfg)
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(p.cfg, 75) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        one = nde.loss_grad(p.weights, sc)
    monkeypatch.setenv("COLNDE_FC_BLOCK", "32")
    monkeypatch.setenv("COLNDE_FC_CW", str(cw))
    with colnde.ColumnNDE(p.cfg, 75) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        blk = nde.loss_grad(p.weights, sc)
        plan = nde.plan()
    assert plan["block_columns"] == 32 and plan["n_blocks"] == 3
    assert np.isclose(blk[0], tot, rtol=FC_LOSS_RTOL) and _rel(blk[2], g) < FC_GRAD_REL
    assert np.isclose(blk[0], one[0], rtol=1e-5) and _rel(blk[2], one[2].astype(np.float64)) < 1e-5


def test_fc32_config4_time_axis_129_save_points():
    """BASELINE configs[3]'s axis: 64 levels, 129 save points over t in [0, 1], 4 RK4 sub-steps (512 steps), on 40 columns."""
    p = synthetic.free_convection_problem(40, Nz=64, n_save=129, substeps=4)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = O.default_loss_scalings(p.cfg)
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(p.cfg, 40) as nde:
        assert nde.engine == ENGINE_FC32
        nde.set_problem(p.x0, p.bcs, truth)
        sol_g = nde.forward(p.weights)
        tot_g, terms_g, grad_g = nde.loss_grad(p.weights, sc)
    _record("fc32/config4_axis", sol_abs=np.abs(sol_g - sol).max(), loss_rel=abs(tot_g - tot) / abs(tot), grad_rel=_rel(grad_g, g))
    assert np.abs(sol_g - sol).max() < 7e-4            # tests/test_gpu_parity.py::test_config4_time_axis_129_save_points: |T| to 14 over 512 steps
    assert np.isclose(tot_g, tot, rtol=FC_LOSS_RTOL)
    assert _rel(grad_g, g) < FC_GRAD_REL


def test_fc32_selection_and_refusals(monkeypatch):
    p = synthetic.free_convection_problem(8, Nz=32, n_save=3)
    with pytest.raises(colnde.ColndeError, match="fc32"):           # wind mixing is not covered: an explicit request fails loudly
        colnde.ColumnNDE(synthetic.wind_mixing_problem(8, n_frames=3).cfg, 8, engine=ENGINE_FC32)
    with pytest.raises(colnde.ColndeError, match="fc32"):           # another network shape
        colnde.ColumnNDE(synthetic.free_convection_problem(8, Nz=32, n_save=3, layer_sizes=(32, 48, 40, 31)).cfg, 8, engine=ENGINE_FC32)
    with colnde.ColumnNDE(synthetic.free_convection_problem(8, Nz=32, n_save=3, layer_sizes=(32, 48, 40, 31)).cfg, 8) as nde:
        assert nde.engine == ENGINE_TILE16                           # AUTO falls back
    with colnde.ColumnNDE(p.cfg.with_(stepper="rkc2"), 8) as nde:
        assert nde.engine == ENGINE_TILE16                           # FreeConvectionNDE under RKC2 (not stiff: nobody needs it) stays in tile16
    ca = synthetic.free_convection_problem(8, Nz=32, n_save=3, substeps=40, convective_adjustment=True).cfg
    for cfg in (ca, ca.with_(stepper="rkc2", substeps=2)):
        with colnde.ColumnNDE(cfg, 8) as nde:
            assert nde.engine == ENGINE_FC32                         # ConvectiveAdjustmentNDE: RK4 and RKC2
    monkeypatch.setenv("COLNDE_FC", "0")
    with colnde.ColumnNDE(p.cfg, 8) as nde:
        assert nde.engine == ENGINE_TILE16


@pytest.mark.parametrize("cw", [16, 32])
@pytest.mark.parametrize("Nz,ncol", [(32, 19), (32, 70), (64, 45)])
def test_fc32_conv_adj_nde_rk4_against_oracle_and_tile16(Nz, ncol, cw, monkeypatch):
    """ConvectiveAdjustmentNDE (convective_adjustment_nde.jl:33-48) on the fc32 engine, sub-stepped RK4, from a profile with an inverted
    layer so that the min(0, K dT/dz) switch is live: the exact discrete adjoint (every stage's own switch pattern, taped as bits)."""
    monkeypatch.setenv("COLNDE_FC_CW", str(cw))
    p = synthetic.free_convection_problem(ncol, Nz=Nz, n_save=5, substeps=20 * (Nz // 32) ** 2, convective_adjustment=True, t_end=0.01)
    x0 = p.x0.copy()
    x0[:, Nz // 2:Nz // 2 + 6] = x0[:, Nz // 2:Nz // 2 + 6][:, ::-1]
    truth = O.solve(p.cfg, x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = O.default_loss_scalings(p.cfg)
    tot, terms, g, sol = O.loss_and_grad(p.cfg, x0, p.bcs, p.weights, truth, sc)
    tot32, _, g32, sol32 = O.loss_and_grad(p.cfg, x0, p.bcs, p.weights, truth, sc, dtype=np.float32)
    res = {}
    for eng in (0, ENGINE_TILE16):
        with colnde.ColumnNDE(p.cfg, ncol, engine=eng) as nde:
            assert nde.engine == (ENGINE_FC32 if eng == 0 else ENGINE_TILE16)
            nde.set_problem(x0, p.bcs, truth)
            sol_g = nde.forward(p.weights)
            tot_g, terms_g, grad_g = nde.loss_grad(p.weights, sc)
            tot_2, _, grad_2 = nde.loss_grad(p.weights, sc)
        assert tot_2 == tot_g and np.array_equal(grad_2, grad_g)
        res[eng] = (sol_g, tot_g, grad_g)
    sol_g, tot_g, grad_g = res[0]
    e32 = (np.abs(sol32 - sol).max(), abs(tot32 - tot) / tot, _rel(g32, g))
    _record("fc32/ca_rk4/cw%d/%d/%d" % (cw, Nz, ncol), sol_abs=np.abs(sol_g - sol).max(), loss_rel=abs(tot_g - tot) / abs(tot), grad_rel=_rel(grad_g, g),
            grad_rel_f32=_rel(grad_g, g32.astype(np.float64)), grad_rel_oracle32_vs_64=e32[2], grad_rel_vs_tile16=_rel(grad_g, res[ENGINE_TILE16][2].astype(np.float64)))
    # layers sitting on the kink make float32 part from float64: the yardstick is the float32 oracle's own distance
    assert np.abs(sol_g - sol).max() < 3 * e32[0] + 2 * FC_SOL_ATOL
    assert abs(tot_g - tot) / abs(tot) < 3 * e32[1] + FC_LOSS_RTOL
    assert _rel(grad_g, g) < 3 * e32[2] + FC_GRAD_REL


@pytest.mark.parametrize("cw", [16, 32])
@pytest.mark.parametrize("Nz,ncol,case", [(32, 37, "stratified"), (64, 33, "stratified"), (64, 5, "inverted")])
def test_fc32_conv_adj_nde_rkc2_against_oracle_and_tile16(Nz, ncol, case, cw, monkeypatch):
    """ConvectiveAdjustmentNDE under the stabilised RKC2 stepper (the configs[3] half the reference integrates with ROCK4) on the fc32
    engine: the oracle's RKC2 recurrence and its one-switch-pattern-per-step pullback, as in tile16 (tests/test_gpu_rkc.py)."""
    monkeypatch.setenv("COLNDE_FC_CW", str(cw))
    p = synthetic.free_convection_problem(ncol, Nz=Nz, n_save=9, substeps=2, convective_adjustment=True, t_end=0.06)
    cfg = p.cfg.with_(stepper="rkc2")
    x0 = p.x0.copy()
    if case == "inverted":
        x0[:, 20:44] = x0[:, 20:44][:, ::-1]
    assert colnde.rkc_stages(cfg) >= 4
    truth = O.solve(cfg, x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = O.default_loss_scalings(cfg)
    tot, terms, g, sol = O.loss_and_grad(cfg, x0, p.bcs, p.weights, truth, sc)
    tot32, _, g32, sol32 = O.loss_and_grad(cfg, x0, p.bcs, p.weights, truth, sc, dtype=np.float32)
    res = {}
    for eng in (0, ENGINE_TILE16):
        with colnde.ColumnNDE(cfg, ncol, engine=eng) as nde:
            assert nde.engine == (ENGINE_FC32 if eng == 0 else ENGINE_TILE16)
            nde.set_problem(x0, p.bcs, truth)
            sol_g = nde.forward(p.weights)
            tot_g, terms_g, grad_g = nde.loss_grad(p.weights, sc)
        res[eng] = (sol_g, tot_g, grad_g)
    sol_g, tot_g, grad_g = res[0]
    e32 = (np.abs(sol32 - sol).max(), abs(tot32 - tot) / tot, _rel(g32, g))
    _record("fc32/ca_rkc2/cw%d/%d/%s" % (cw, Nz, case), sol_abs=np.abs(sol_g - sol).max(), loss_rel=abs(tot_g - tot) / abs(tot), grad_rel=_rel(grad_g, g),
            sol_abs_oracle32_vs_64=e32[0], loss_rel_oracle32_vs_64=e32[1], grad_rel_oracle32_vs_64=e32[2],
            sol_abs_vs_tile16=np.abs(sol_g - res[ENGINE_TILE16][0]).max(), grad_rel_vs_tile16=_rel(grad_g, res[ENGINE_TILE16][2].astype(np.float64)))
    assert np.isfinite(sol_g).all() and np.isfinite(grad_g).all()
    assert np.abs(sol_g - sol).max() < 5 * e32[0] + 2 * FC_SOL_ATOL
    assert abs(tot_g - tot) / abs(tot) < 5 * e32[1] + FC_LOSS_RTOL
    assert _rel(grad_g, g) < 5 * e32[2] + FC_GRAD_REL


@pytest.mark.parametrize("cw", [16, 32])
@pytest.mark.parametrize("nx,ny", [(1, 1), (16, 9), (33, 31), (256, 256)])
def test_fc32_inference_forcing_against_oracle_and_tile16(nx, ny, cw, monkeypatch):
    """`compute_neural_network_forcing!` (double_gyre_nn.jl:149-168; BASELINE configs[4]'s 256 x 256 grid included) on the fc32 sections: a
    workgroup walks over 32-column tiles with the A-operand ring streaming across them; against the float64 oracle and against tile16."""
    monkeypatch.setenv("COLNDE_FC_CW", str(cw))
    cfg, T, top, w = synthetic.inference_problem(nx, ny)
    n = nx * ny
    ref = O.infer_forcing(cfg, T[:4096], top[:4096], w, 1000.0)
    res = {}
    for eng in (0, ENGINE_TILE16):
        with colnde.ColumnNDE(cfg, n, engine=eng) as nde:
            assert nde.engine == (ENGINE_FC32 if eng == 0 else ENGINE_TILE16)
            res[eng] = nde.infer_forcing(w, T, top, 1000.0)
    _record("fc32/infer/%d" % n, rel=_rel(res[0][:4096], ref), rel_vs_tile16=_rel(res[0], res[ENGINE_TILE16].astype(np.float64)))
    assert _rel(res[0][:4096], ref) < 2e-6
    assert _rel(res[0], res[ENGINE_TILE16].astype(np.float64)) < 1e-6


@pytest.mark.parametrize("cw", [16, 32])
@pytest.mark.parametrize("model,seg,block", [("fc", 3, None), ("fc", 1, None), ("ca_rk4", 2, None), ("ca_rkc2", 3, None), ("fc", 2, 32)])
def test_fc32_time_segmented_tapes_equal_the_single_pass(model, seg, block, cw, monkeypatch):
    """When the tapes of all columns do not fit, fc32 cuts the TIME axis instead of the columns (every CU keeps its two workgroups): a tape-less
    forward pass saves the states at the save points, then each segment — from the last to the first — is re-run with tapes from its saved state,
    back-propagated (λ handed on through a device buffer) and contracted.  Restarting at a save point is exact, so the segmented gradient must
    equal the single-pass one to summation order, for every stepper and with column blocks on top (COLNDE_FC_SEG / COLNDE_FC_BLOCK force the cuts)."""
    monkeypatch.setenv("COLNDE_FC_CW", str(cw))
    ca = model != "fc"
    p = synthetic.free_convection_problem(75, Nz=32, n_save=8, substeps=24 if model == "ca_rk4" else 2, convective_adjustment=ca, t_end=0.02)
    cfg = p.cfg.with_(stepper="rkc2") if model == "ca_rkc2" else p.cfg
    x0 = p.x0.copy()
    if ca:
        x0[:, 14:20] = x0[:, 14:20][:, ::-1]
    truth = O.solve(cfg, x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = O.default_loss_scalings(cfg)
    tot, terms, g, sol = O.loss_and_grad(cfg, x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(cfg, 75) as nde:
        nde.set_problem(x0, p.bcs, truth)
        one = nde.loss_grad(p.weights, sc)
        assert nde.plan()["time_segments"] == 0
    monkeypatch.setenv("COLNDE_FC_SEG", str(seg))
    if block:
        monkeypatch.setenv("COLNDE_FC_BLOCK", str(block))
    with colnde.ColumnNDE(cfg, 75) as nde:
        nde.set_problem(x0, p.bcs, truth)
        cut = nde.loss_grad(p.weights, sc)
        again = nde.loss_grad(p.weights, sc)
        plan = nde.plan()
    assert plan["time_segments"] == -(-7 // seg) and plan["n_blocks"] == (3 if block else 1)
    assert again[0] == cut[0] and np.array_equal(again[2], cut[2])
    _record("fc32/segments/%s/%d" % (model, seg), loss_rel_vs_single=abs(cut[0] - one[0]) / abs(one[0]), grad_rel_vs_single=_rel(cut[2], one[2].astype(np.float64)),
            grad_rel=_rel(cut[2], g))
    assert np.isclose(cut[0], one[0], rtol=2e-6)
    np.testing.assert_allclose(cut[1], one[1], rtol=2e-6, atol=1e-12)
    assert _rel(cut[2], one[2].astype(np.float64)) < 2e-6
    if not ca:
        assert np.isclose(cut[0], tot, rtol=FC_LOSS_RTOL) and _rel(cut[2], g) < FC_GRAD_REL


@pytest.mark.parametrize("Nz,ncol,engine", [(64, 45, 0), (32, 70, 0), (64, 20, ENGINE_TILE16), (32, 33, ENGINE_TILE16)])
def test_dw_gemm_on_the_bf16_pipe_with_exact_operand_splitting_is_float32_grade(Nz, ncol, engine, monkeypatch):
    """The tape GEMM alone on the split arithmetic: `dw_gemm_split_kernel` contracts the same delta-tape records as `dw_gemm_lds_kernel`, in passes
    by layer, with six bf16 MFMA products of the exact three-way splits of both operands (split once per record into LDS planes) in place of
    v_mfma_f32_32x32x2_f32.  Same forward and adjoint kernels, same tapes: the weight gradient differs from the fp32 GEMM's by float32 round-off
    (stated: 2e-6 relative L2), the bias gradients and the loss are bit-identical, and the gap to the float64 oracle is the same (within 1.5x)."""
    p = synthetic.free_convection_problem(ncol, Nz=Nz, n_save=9, substeps=2, t_end=0.02)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = O.default_loss_scalings(p.cfg)
    tot, terms, g, _ = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    out = {}
    with colnde.ColumnNDE(p.cfg, ncol, engine=engine) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        for mode in ("0", "1"):
            _arith(monkeypatch, nde, dw=mode == "1")
            out[mode] = nde.loss_grad(p.weights, sc)
            again = nde.loss_grad(p.weights, sc)
            assert nde.plan()["bf16x3_dw"] == (mode == "1")
            assert np.array_equal(again[2], out[mode][2])               # bit-reproducible in either mode
    (t32, _, g32), (tsp, _, gsp) = out["0"], out["1"]
    assert t32 == tsp
    H, off, wmask = 4 * Nz, 0, np.zeros(p.cfg.n_params, bool)
    for n, is_w in ((Nz * H, True), (H, False), (H * H, True), (H, False), (H * (Nz - 1), True), (Nz - 1, False)):
        wmask[off:off + n] = is_w
        off += n
    np.testing.assert_array_equal(gsp[~wmask], g32[~wmask])            # biases: column sums taken by the adjoint kernels, not by the GEMM
    d = _rel(gsp[wmask], g32[wmask].astype(np.float64))
    e32, esp = _rel(g32, g), _rel(gsp, g)
    _record("dw_gemm_split/%d/%d/%d" % (Nz, ncol, engine), split_vs_fp32=d, fp32_vs_oracle=e32, split_vs_oracle=esp)
    assert 0.0 < d < 2e-6
    assert esp < max(1.5 * e32, 2e-6)
