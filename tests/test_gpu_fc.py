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
from tests.test_gpu_parity import _record, _rel, FC_SOL_ATOL, FC_LOSS_RTOL, FC_GRAD_REL

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("Nz,ncol", [(32, 1), (32, 19), (32, 64), (32, 97), (64, 5), (64, 32), (64, 45), (64, 130)])
def test_fc32_against_oracle_and_tile16(Nz, ncol):
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
    _record("fc32/%d/%d" % (Nz, ncol), sol_abs=np.abs(sol_g - sol).max(), loss_rel=abs(tot_g - tot) / abs(tot), grad_rel=_rel(grad_g, g),
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


@pytest.mark.parametrize("Nz", [32, 64])
def test_fc32_column_blocks_are_additive(Nz, monkeypatch):
    """COLNDE_FC_BLOCK=32: 75 columns run forward -> adjoint -> dW GEMM in three passes through tapes sized for one 32-column tile."""
    p = synthetic.free_convection_problem(75, Nz=Nz, n_save=5, substeps=2, t_end=0.01)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = O.default_loss_scalings(p.cfg)
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(p.cfg, 75) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        one = nde.loss_grad(p.weights, sc)
    monkeypatch.setenv("COLNDE_FC_BLOCK", "32")
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
    with pytest.raises(colnde.ColndeError, match="fc32"):           # ConvectiveAdjustmentNDE is not covered: explicit request fails loudly
        colnde.ColumnNDE(synthetic.free_convection_problem(8, Nz=32, n_save=3, substeps=40, convective_adjustment=True).cfg, 8, engine=ENGINE_FC32)
    with pytest.raises(colnde.ColndeError, match="fc32"):           # another network shape
        colnde.ColumnNDE(synthetic.free_convection_problem(8, Nz=32, n_save=3, layer_sizes=(32, 48, 40, 31)).cfg, 8, engine=ENGINE_FC32)
    with colnde.ColumnNDE(synthetic.free_convection_problem(8, Nz=32, n_save=3, layer_sizes=(32, 48, 40, 31)).cfg, 8) as nde:
        assert nde.engine == ENGINE_TILE16                           # AUTO falls back
    with colnde.ColumnNDE(p.cfg.with_(stepper="rkc2"), 8) as nde:
        assert nde.engine == ENGINE_TILE16                           # RKC2 lives in tile16
    monkeypatch.setenv("COLNDE_FC", "0")
    with colnde.ColumnNDE(p.cfg, 8) as nde:
        assert nde.engine == ENGINE_TILE16
