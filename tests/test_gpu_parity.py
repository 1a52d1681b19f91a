"""GPU parity tests proper: the HIP tile engine, called through the C ABI, against the oracle on the same
seeded inputs.  Floating-point path: tolerances are stated per test (float32 engine vs float64 oracle)."""
import numpy as np
import pytest

import colnde
from colnde import synthetic
from oracle import nde_oracle as O
from tests.test_oracle import VARIANTS, INPLACE_VARIANTS

pytestmark = pytest.mark.gpu

# Tolerances of the float32 HIP path against the float64 oracle: about 10x the largest error MEASURED on an MI355X for the
# family of cases (profiles/r03_parity_errors.json keeps the last measuring run; COLNDE_RECORD_ERRORS=1 re-measures).
# Wind mixing, 3..9 frames (smooth profiles, loss O(1e-3)): measured sol 2.1e-6, loss 8e-6, gradient 1.8e-5.
SOL_ATOL = 2e-5          # scaled units, O(1) profiles
LOSS_RTOL = 8e-5
GRAD_REL = 2e-4          # relative L2 error of the float32 gradient vs the float64 oracle
# Free convection (relu nets; the loss is a small difference of O(1) profiles, so float32 cancellation shows in it and relu
# kinks flip between float32 and float64): measured sol 7.4e-6 (stiff CA variant), loss 3.2e-4, gradient 3.9e-4.
FC_SOL_ATOL = 8e-5
FC_LOSS_RTOL = 3e-3
FC_GRAD_REL = 4e-3
# The 576-step bench horizon (round-off accumulates over the steps).  Loss and gradient tolerances are keyed by the weight
# divisor: with the bench's own weights/1e5 the net is near zero and the loss (3e-9) IS float32 round-off of the trajectories
# (measured: total 1.6e-3, the T term 1.8e-2, gradient 2.1e-3); with weights/1e2: sol 7.6e-6, loss 1.2e-6, terms 2e-6, gradient 3.2e-6.
LONG_SOL_ATOL = 8e-5
LONG_LOSS_RTOL = {1e2: 2e-5, 1e5: 1.5e-2}
LONG_TERMS_RTOL = {1e2: 2e-5, 1e5: 1.5e-1}
LONG_GRAD_REL = {1e2: 4e-5, 1e5: 2e-2}


def _rel(a, b):
    return np.linalg.norm(np.asarray(a, np.float64) - b) / (np.linalg.norm(b) + 1e-300)


def _arith(monkeypatch, nde, fwd=None, adj=None, dw=None, base="f32_mfma"):
    """Matrix arithmetic of an existing handle, per kernel family: `base` ("f32_mfma" | "bf16x3_exact", include/colnde.h COLNDE_MATRIX_*)
    for all three, then single families forced through the test overrides COLNDE_{FWD,ADJ,DW}_SPLIT (True: exact three-way bf16 split,
    False: f32 MFMA, None: as `base`).  The overrides are read when the arithmetic is resolved, i.e. by set_matrix_arithmetic here —
    every test that compares the two arithmetics switches them itself, nothing depends on the environment pytest was started in."""
    for k, v in (("COLNDE_FWD_SPLIT", fwd), ("COLNDE_ADJ_SPLIT", adj), ("COLNDE_DW_SPLIT", dw)):
        if v is None:
            monkeypatch.delenv(k, raising=False)
        else:
            monkeypatch.setenv(k, "1" if v else "0")
    nde.set_matrix_arithmetic(base)


def _record(case, **errs):
    """With COLNDE_RECORD_ERRORS set, append the measured errors to gpurun_out/parity_errors.jsonl: the tolerances in this
    file are set to about 10x what this records (profiles/r03_parity_errors.json keeps the last measuring run)."""
    import json
    import os
    if os.environ.get("COLNDE_RECORD_ERRORS"):
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/parity_errors.jsonl", "a") as f:
            f.write(json.dumps(dict(case=case, **{k: float(v) for k, v in errs.items()})) + "\n")


@pytest.mark.parametrize("name", sorted(VARIANTS) + sorted(INPLACE_VARIANTS))
def test_rhs_wind_mixing(name):
    kw = dict(VARIANTS[name] if name in VARIANTS else INPLACE_VARIANTS[name])
    p = synthetic.wind_mixing_problem(37, n_frames=3, weight_divisor=10.0, **kw)
    ref = O.rhs(p.cfg, p.x0, p.bcs, p.weights, 0.02)
    with colnde.ColumnNDE(p.cfg, p.n_columns) as nde:
        got = nde.rhs(p.x0, p.weights, p.bcs, 0.02)
    _record("rhs_wm/" + name, rhs_rel=_rel(got, ref))
    assert _rel(got, ref) < 1e-6                                            # measured 1.0e-7


@pytest.mark.parametrize("name", sorted(INPLACE_VARIANTS))
def test_forward_inplace_variant(name):
    """`solve_NDE_mutating` (training_postprocessing.jl:55-159): colnde_forward under the in-place `NDE!` arithmetic, including
    the ν_T = κ switch and the un-offset diurnal top flux (κ = 0.1: inside RK4's stability region at two sub-steps per frame)."""
    p = synthetic.wind_mixing_problem(21, n_frames=9, weight_divisor=1e2, kappa=0.1, **INPLACE_VARIANTS[name])
    sol = O.solve(p.cfg, p.x0, p.bcs, p.weights)
    with colnde.ColumnNDE(p.cfg, p.n_columns) as nde:
        nde.set_problem(p.x0, p.bcs)
        sol_g = nde.forward(p.weights)
        with pytest.raises(colnde.ColndeError, match="in-place"):          # an evaluation RHS: no gradient through it
            nde.set_problem(p.x0, p.bcs, sol_g)
            nde.loss_grad(p.weights, [1, 1, 1, 0, 0, 0])
    _record("forward_inplace/" + name, sol_abs=np.abs(sol_g - sol).max())
    assert np.abs(sol_g - sol).max() < SOL_ATOL
    if "ca" in name:                                                        # the switch changed the trajectory
        plain = O.solve(p.cfg.with_(convective_adjustment=False), p.x0, p.bcs, p.weights)
        assert np.abs(plain - sol).max() > 100 * SOL_ATOL


def test_time_step_outside_rk4_stability_is_refused():
    """kappa = 10 (the reference default) in the convective-adjustment branch needs 135 RK4 sub-steps per 10-minute frame; with 2
    the solve would blow up and come back with rc = 0.  colnde_forward / _loss / _loss_grad refuse it and say what is needed;
    colnde_rhs (no time stepping) still answers; COLNDE_ALLOW_UNSTABLE_DT=1 is the explicit override."""
    p = synthetic.wind_mixing_problem(5, n_frames=3, weight_divisor=1e2, **VARIANTS["conv_adj_branch"])
    cfg = p.cfg.with_(kappa=10.0)
    assert colnde.min_substeps(cfg) == 135
    with colnde.ColumnNDE(cfg, 5) as nde:
        nde.set_problem(p.x0, p.bcs, np.zeros((5, 3, 96), np.float32))
        assert np.isfinite(nde.rhs(p.x0, p.weights, p.bcs, 0.0)).all()
        for call in (lambda: nde.forward(p.weights), lambda: nde.loss(p.weights, [1] * 6), lambda: nde.loss_grad(p.weights, [1] * 6)):
            with pytest.raises(colnde.ColndeError, match="substeps >= 135"):
                call()
    with colnde.ColumnNDE(cfg.with_(substeps=135), 5) as nde:
        nde.set_problem(p.x0, p.bcs)
        assert np.isfinite(nde.forward(p.weights)).all()


@pytest.mark.parametrize("name", ["mpp_zero_weights", "mpp_bc_faces", "smooth_NN", "smooth_Ri", "diurnal",
                                  "conv_adj_branch", "swish"])
def test_forward_loss_grad_wind_mixing(name):
    p = synthetic.wind_mixing_problem(21, n_frames=9, weight_divisor=1e2, **VARIANTS[name])
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = np.array([1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3])
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(p.cfg, p.n_columns) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        sol_g = nde.forward(p.weights)
        tot_l, terms_l = nde.loss(p.weights, sc)
        tot_g, terms_g, grad_g = nde.loss_grad(p.weights, sc)
    _record("test_forward_loss_grad_wind_mixing" + "/" + str(name), sol_abs=np.abs(sol_g - sol).max(), loss_rel=abs(tot_g - tot) / abs(tot), grad_rel=_rel(grad_g, g))
    assert np.abs(sol_g - sol).max() < SOL_ATOL
    np.testing.assert_allclose(terms_l, terms, rtol=LOSS_RTOL, atol=1e-12)
    np.testing.assert_allclose(terms_g, terms, rtol=LOSS_RTOL, atol=1e-12)
    assert np.isclose(tot_l, tot, rtol=LOSS_RTOL) and np.isclose(tot_g, tot, rtol=LOSS_RTOL)
    assert _rel(grad_g, g) < GRAD_REL


@pytest.mark.parametrize("Nz,ca", [(32, False), (32, True), (64, False)])
def test_free_convection(Nz, ca):
    p = synthetic.free_convection_problem(19, Nz=Nz, n_save=5, substeps=20 if ca else 2,
                                          convective_adjustment=ca, t_end=0.01)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = O.default_loss_scalings(p.cfg)
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(p.cfg, p.n_columns) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        sol_g = nde.forward(p.weights)
        tot_g, terms_g, grad_g = nde.loss_grad(p.weights, sc)
    _record("test_free_convection" + "/" + str(Nz) + "/" + str(ca), sol_abs=np.abs(sol_g - sol).max(), loss_rel=abs(tot_g - tot) / abs(tot), grad_rel=_rel(grad_g, g))
    assert np.abs(sol_g - sol).max() < FC_SOL_ATOL
    assert np.isclose(tot_g, tot, rtol=FC_LOSS_RTOL)
    assert _rel(grad_g, g) < FC_GRAD_REL


@pytest.mark.parametrize("case", ["fc32", "fc64", "fc32_ca", "wind_mixing", "fc64_l2stream", "wind_mixing_l2stream", "fc64_512threads", "fc64_noztape",
                                  "fc32_inregister", "fc64_inregister", "wind_mixing_inregister", "fc64_blocked", "wind_mixing_blocked"])
def test_tile16_taped_weight_gradients(case, monkeypatch):
    """tile16's gradient modes: layer deltas and hidden pre-activations taped with every dW contracted by the split-K GEMM kernel
    (the default), its variants, and the in-register fallback."""
    from colnde.nde import ENGINE_TILE16
    monkeypatch.setenv("COLNDE_T16_DWTAPE", "1")
    if case.endswith("_blocked"):                    # tapes sized for one 16-column block: 21 / 37 columns run as 2 / 3 passes
        monkeypatch.setenv("COLNDE_T16_BLOCK", "16")
        case = case[:-len("_blocked")]
    if case.endswith("_inregister"):                 # the fallback when the tapes do not fit: accumulators resident in registers
        monkeypatch.setenv("COLNDE_T16_DWTAPE", "0")
        case = case[:-len("_inregister")]
    if case.endswith("_l2stream"):                   # the split-K kernel that reads its operands straight from L2 (no LDS staging)
        monkeypatch.setenv("COLNDE_T16_DWLDS", "0")
        case = case[:-len("_l2stream")]
    if case.endswith("_noztape"):                    # the adjoint recomputes the forward GEMMs instead of reading taped pre-activations
        monkeypatch.setenv("COLNDE_T16_ZTAPE", "0")
        case = case[:-len("_noztape")]
    if case.endswith("_512threads"):
        monkeypatch.setenv("COLNDE_T16_TAPE_THREADS", "512")
        case = case[:-len("_512threads")]
    if case == "wind_mixing":
        p = synthetic.wind_mixing_problem(21, n_frames=5, weight_divisor=1e2)
        sc = np.array([1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3])
    else:
        ca = case.endswith("_ca")
        p = synthetic.free_convection_problem(37, Nz=64 if "64" in case else 32, n_save=5, substeps=20 if ca else 2,
                                              convective_adjustment=ca, t_end=0.01)
        sc = O.default_loss_scalings(p.cfg)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(p.cfg, p.n_columns, engine=ENGINE_TILE16) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        tot_g, terms_g, grad_g = nde.loss_grad(p.weights, sc)
        tot_2, _, grad_2 = nde.loss_grad(p.weights, sc)
    _record("test_tile16_taped_weight_gradients" + "/" + str(case), loss_rel=abs(tot_g - tot) / abs(tot), grad_rel=_rel(grad_g, g))
    wm = case == "wind_mixing"
    assert np.isclose(tot_g, tot, rtol=LOSS_RTOL if wm else FC_LOSS_RTOL)
    assert _rel(grad_g, g) < (GRAD_REL if wm else FC_GRAD_REL)
    assert tot_2 == tot_g and np.array_equal(grad_2, grad_g)        # fixed-order reductions: bit-reproducible


def test_infer_forcing():
    cfg, T, top, w = synthetic.inference_problem(16, 9)
    ref = O.infer_forcing(cfg, T, top, w, 1000.0)
    with colnde.ColumnNDE(cfg, T.shape[0]) as nde:
        got = nde.infer_forcing(w, T, top, 1000.0)
    _record("infer_forcing", rel=_rel(got, ref))
    assert _rel(got, ref) < 2e-6                                            # measured 2e-7


def test_long_horizon_2day_suite_shape():
    """BASELINE config 3: 8 simulations x 32 levels x 289 frames, two RK4 sub-steps per frame."""
    p = synthetic.wind_mixing_problem(8, n_frames=289)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = O.default_loss_scalings(p.cfg)
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(p.cfg, p.n_columns) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        sol_g = nde.forward(p.weights)
        tot_g, terms_g, grad_g = nde.loss_grad(p.weights, sc)
    _record("test_long_horizon_2day_suite_shape", sol_abs=np.abs(sol_g - sol).max(), loss_rel=abs(tot_g - tot) / abs(tot), grad_rel=_rel(grad_g, g))
    assert np.abs(sol_g - sol).max() < LONG_SOL_ATOL
    assert np.isclose(tot_g, tot, rtol=LONG_LOSS_RTOL[1e5])               # synthetic default = the bench's weights/1e5
    assert _rel(grad_g, g) < LONG_GRAD_REL[1e5]


@pytest.mark.parametrize("n_col,divisor", [(8, 1e2), (64, 1e2), (64, 1e5)])
def test_regtile_long_horizon_against_oracle(n_col, divisor):
    """The headline engine over the bench horizon (289 frames x 2 RK4 sub-steps = 576 steps), named explicitly (AUTO sends
    problems this small to tile16): solution, six loss terms and gradient against the float64 oracle.  divisor 1e5 is the
    bench's own weight set (`re(weights ./ 1f5)`, train_NDE.jl:105-107: a near-zero net, loss ~1e-9)."""
    from colnde.nde import ENGINE_REGTILE
    p = synthetic.wind_mixing_problem(n_col, n_frames=289, weight_divisor=divisor)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = O.default_loss_scalings(p.cfg)
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(p.cfg, p.n_columns, engine=ENGINE_REGTILE) as nde:
        assert nde.engine == ENGINE_REGTILE
        nde.set_problem(p.x0, p.bcs, truth)
        sol_g = nde.forward(p.weights)
        tot_g, terms_g, grad_g = nde.loss_grad(p.weights, sc)
    _record("regtile_576/%d/%g" % (n_col, divisor), sol_abs=np.abs(sol_g - sol).max(), loss_rel=abs(tot_g - tot) / tot,
            terms_rel=np.abs(terms_g / terms - 1).max(), grad_rel=_rel(grad_g, g))
    assert np.abs(sol_g - sol).max() < LONG_SOL_ATOL
    np.testing.assert_allclose(terms_g, terms, rtol=LONG_TERMS_RTOL[divisor], atol=0)
    assert np.isclose(tot_g, tot, rtol=LONG_LOSS_RTOL[divisor], atol=0)
    assert _rel(grad_g, g) < LONG_GRAD_REL[divisor]


@pytest.mark.parametrize("engine", [0, 2])
def test_eight_day_suite_horizon(engine):
    """The reference's long training suites (wind_mixing 8DaySuite: 1,153 frames; two RK4 sub-steps per frame = 2,304 steps, four times
    the bench horizon) on 8 simulations: engine AUTO (the net-split kernels) and regtile named explicitly, against the float64 oracle.
    Round-off accumulates over 9,216 stage evaluations: measured sol 1.7e-5, loss 1.2e-6 / 5.2e-6, gradient 2.0e-6 / 2.4e-6
    (weights/1e2); the tolerances below are about 10x that."""
    p = synthetic.wind_mixing_problem(8, n_frames=1153, weight_divisor=1e2)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = O.default_loss_scalings(p.cfg)
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(p.cfg, p.n_columns, engine=engine) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        sol_g = nde.forward(p.weights)
        tot_g, terms_g, grad_g = nde.loss_grad(p.weights, sc)
        plan = nde.plan()
    assert plan["split_adjoint"] == (engine == 0)
    _record("eight_day/%d" % engine, sol_abs=np.abs(sol_g - sol).max(), loss_rel=abs(tot_g - tot) / tot,
            terms_rel=np.abs(terms_g / terms - 1).max(), grad_rel=_rel(grad_g, g))
    assert np.abs(sol_g - sol).max() < 1.7e-4
    assert np.isclose(tot_g, tot, rtol=5e-5, atol=0)
    assert _rel(grad_g, g) < 2.5e-5


def test_config1_shape_one_column_ten_frames():
    """BASELINE config 1 (`test_nonmutating_NDE.jl:49-58`): 1 simulation, 32 levels, MPP + zero_weights, nu0 = 1e-4, nu_minus = 0.1,
    dRi = 1, Ric = 0.25, Pr = 1, `tsteps = 1:1:10` of the 1,153-frame 8-day record (tau = 691,200 s, dt = 1/1152), both engines."""
    p = synthetic.wind_mixing_problem(1, n_frames=10, n_frames_total=1153, tau=691200.0, weight_divisor=1e2)
    assert len(p.cfg.save_times) == 10 and np.isclose(p.cfg.save_times[1], 1 / 1152)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = O.default_loss_scalings(p.cfg)
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    for eng in (1, 2):
        with colnde.ColumnNDE(p.cfg, 1, engine=eng) as nde:
            nde.set_problem(p.x0, p.bcs, truth)
            sol_g = nde.forward(p.weights)
            tot_g, terms_g, grad_g = nde.loss_grad(p.weights, sc)
        _record("config1/engine%d" % eng, sol_abs=np.abs(sol_g - sol).max(), loss_rel=abs(tot_g - tot) / tot, grad_rel=_rel(grad_g, g))
        assert np.abs(sol_g - sol).max() < SOL_ATOL
        np.testing.assert_allclose(terms_g, terms, rtol=LOSS_RTOL, atol=0)
        assert _rel(grad_g, g) < GRAD_REL


CA64_F32 = (2e-6, 5e-6, 6e-5)          # HIP vs the float32 oracle on the kinked case: measured 1.8e-7, 4.2e-7, 5.6e-6


def test_conv_adj_nde_64_levels():
    """`ConvectiveAdjustmentNDE` (convective_adjustment_nde.jl:33-48, K = 10) at the 64 levels of BASELINE config 4, on profiles
    with unstable faces so that the min(0, K dT/dz) term is live; RK4 needs dt <= 2.785 / (4 K C Nz^2) = 3.4e-5 here."""
    p = synthetic.free_convection_problem(19, Nz=64, n_save=5, substeps=96, convective_adjustment=True, t_end=0.01)
    x0 = p.x0.copy()
    x0[:, 40:48] = x0[:, 40:48][:, ::-1]                                    # an unstable layer besides the cooled surface
    truth = O.solve(p.cfg, x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = O.default_loss_scalings(p.cfg)
    tot, terms, g, sol = O.loss_and_grad(p.cfg, x0, p.bcs, p.weights, truth, sc)
    assert np.isfinite(sol).all() and np.abs(sol - x0[:, None]).max() > 0.05            # the adjustment acted
    with colnde.ColumnNDE(p.cfg, p.n_columns) as nde:
        nde.set_problem(x0, p.bcs, truth)
        sol_g = nde.forward(p.weights)
        tot_g, terms_g, grad_g = nde.loss_grad(p.weights, sc)
    # Once a layer is adjusted its dT/dz hovers at 0-, i.e. AT the kink of min(0, K dT/dz): which side a face sits on is
    # decided by round-off, so the float32 gradient differs from the float64 one by what those faces carry (179 of 1,197 faces
    # end with |K dT/dz| < 1e-3 here).  The float32 NumPy oracle shows the same 4.8e-2 gap to float64 as the HIP path does, and
    # the HIP path agrees with THAT oracle tightly: the gap is float32's, not the kernel's.
    tot32, _, g32, sol32 = O.loss_and_grad(p.cfg, x0, p.bcs, p.weights, truth, sc, dtype=np.float32)
    _record("ca_nde_64", sol_abs=np.abs(sol_g - sol).max(), loss_rel=abs(tot_g - tot) / tot, grad_rel=_rel(grad_g, g),
            sol_abs_f32=np.abs(sol_g - sol32).max(), loss_rel_f32=abs(tot_g - tot32) / tot32, grad_rel_f32=_rel(grad_g, g32.astype(np.float64)),
            grad_rel_oracle32_vs_64=_rel(g32, g))
    assert np.abs(sol_g - sol).max() < 2 * FC_SOL_ATOL                      # measured 1.4e-5
    assert np.isclose(tot_g, tot, rtol=FC_LOSS_RTOL)                        # measured 2.5e-4
    assert _rel(grad_g, g) < 3 * _rel(g32, g) + FC_GRAD_REL                 # measured 4.79e-2, the float32 oracle's own gap
    assert np.abs(sol_g - sol32).max() < CA64_F32[0] and abs(tot_g - tot32) <= CA64_F32[1] * tot32
    assert _rel(grad_g, g32.astype(np.float64)) < CA64_F32[2]


def test_config4_time_axis_129_save_points():
    """BASELINE config 4's real time axis: 129 save points over t in [0, 1] (`iterations = 1:9:1153`,
    train_free_convection_nde.jl:110-122), 64 levels, 64-256-256-63 relu, on a few columns.  (`ConvectiveAdjustmentNDE` over
    this axis needs 256 RK4 sub-steps per interval — dt <= 3.4e-5 — which the float64 oracle takes minutes for: it is covered
    at 64 levels by test_conv_adj_nde_64_levels and over the full axis by the stabilised stepper's tests.)"""
    ca = False
    p = synthetic.free_convection_problem(5, Nz=64, n_save=129, substeps=4, convective_adjustment=ca, t_end=1.0)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = O.default_loss_scalings(p.cfg)
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    assert np.isfinite(sol).all()
    with colnde.ColumnNDE(p.cfg, p.n_columns) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        sol_g = nde.forward(p.weights)
        tot_g, terms_g, grad_g = nde.loss_grad(p.weights, sc)
    _record("config4_axis/ca%d" % ca, sol_abs=np.abs(sol_g - sol).max(), loss_rel=abs(tot_g - tot) / tot, grad_rel=_rel(grad_g, g))
    assert np.abs(sol_g - sol).max() < 7e-4                                 # 512 steps, profiles reaching |T| = 14: measured 6.9e-5
    assert np.isclose(tot_g, tot, rtol=FC_LOSS_RTOL)                        # measured 5.9e-5
    assert _rel(grad_g, g) < FC_GRAD_REL                                    # measured 1.7e-4


# ---- engine selection: regtile (register-resident, static wind-mixing shape) vs tile16 (generic) -------------------
def test_engine_selection_and_cross_check():
    from colnde.nde import ENGINE_AUTO, ENGINE_TILE16, ENGINE_REGTILE
    p = synthetic.wind_mixing_problem(70, n_frames=9, weight_divisor=1e2)      # 70 columns: ragged 32- and 16-column tiles
    sc = [1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3]
    res = {}
    for eng in (ENGINE_TILE16, ENGINE_REGTILE, ENGINE_AUTO):
        with colnde.ColumnNDE(p.cfg, p.n_columns, engine=eng) as nde:
            # AUTO sends small problems (<= 8,192 columns: latency points) to tile16 and everything else regtile covers to regtile
            assert nde.engine == (ENGINE_REGTILE if eng == ENGINE_REGTILE else ENGINE_TILE16)
            nde.set_problem(p.x0, p.bcs)
            truth = nde.forward(p.weights_truth)
            nde.set_problem(p.x0, p.bcs, truth)
            res[eng] = (truth, nde.loss_grad(p.weights, sc))
    # two independent kernel families agree to float32 round-off
    assert np.abs(res[ENGINE_TILE16][0] - res[ENGINE_REGTILE][0]).max() < 2e-5
    a, b = res[ENGINE_TILE16][1], res[ENGINE_REGTILE][1]
    assert np.isclose(a[0], b[0], rtol=1e-4)
    assert _rel(a[2], b[2].astype(np.float64)) < 1e-4
    with colnde.ColumnNDE(p.cfg, 8192) as nde:
        assert nde.engine == ENGINE_TILE16
    with colnde.ColumnNDE(p.cfg, 8208) as nde:
        assert nde.engine == ENGINE_REGTILE
    # smoothing is outside the regtile engine's coverage: AUTO falls back, an explicit request fails loudly
    ps = synthetic.wind_mixing_problem(8, n_frames=3, smooth_NN=True)
    with colnde.ColumnNDE(ps.cfg, 8) as nde:
        assert nde.engine == ENGINE_TILE16
    with pytest.raises(colnde.ColndeError, match="regtile"):
        colnde.ColumnNDE(ps.cfg, 8, engine=ENGINE_REGTILE)


@pytest.mark.parametrize("name", ["mpp_zero_weights", "mpp_bc_faces", "diurnal", "conv_adj_branch", "swish", "raw", "dRi_small", "relu",
                                  "tanh", "leakyrelu"])
def test_regtile_engine_against_oracle(name):
    from colnde.nde import ENGINE_REGTILE
    p = synthetic.wind_mixing_problem(45, n_frames=9, weight_divisor=1e2, **VARIANTS[name])
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = np.array([1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3])
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(p.cfg, p.n_columns, engine=ENGINE_REGTILE) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        sol_g = nde.forward(p.weights)
        tot_g, terms_g, grad_g = nde.loss_grad(p.weights, sc)
    _record("test_regtile_engine_against_oracle" + "/" + str(name), sol_abs=np.abs(sol_g - sol).max(), loss_rel=abs(tot_g - tot) / abs(tot), grad_rel=_rel(grad_g, g))
    assert np.abs(sol_g - sol).max() < SOL_ATOL
    np.testing.assert_allclose(terms_g, terms, rtol=LOSS_RTOL, atol=1e-12)
    assert _rel(grad_g, g) < GRAD_REL


@pytest.mark.parametrize("ma", ["bf16x3_exact", "f32_mfma"])
@pytest.mark.parametrize("env", [{"COLNDE_RT_ZTAPE": "0"}, {"COLNDE_RT_FWD": "32"}, {"COLNDE_RT_BLOCK": "32"},
                                 {"COLNDE_RT_BLOCK": "64", "COLNDE_RT_ZTAPE": "0"}, {"COLNDE_DW_SPLIT": "0"},
                                 {"COLNDE_DW_SPLIT": "1", "COLNDE_RT_BLOCK": "32"}, {"COLNDE_FWD_SPLIT": "0"},
                                 {"COLNDE_FWD_SPLIT": "1", "COLNDE_DW_SPLIT": "0", "COLNDE_RT_BLOCK": "64"}, {"COLNDE_ADJ_SPLIT": "1"},
                                 {"COLNDE_ADJ_SPLIT": "0", "COLNDE_FWD_SPLIT": "1", "COLNDE_DW_SPLIT": "1", "COLNDE_RT_BLOCK": "32"}])
def test_regtile_alternative_paths_against_oracle(env, ma, monkeypatch):
    """The variants behind environment switches, under both matrix arithmetics (the handle's configured one, with single kernel families forced
    the other way by the test overrides): no Z1 tape (the adjoint recomputes layer 1; its split W1ᵀ products need the tape, so it runs f32 MFMA),
    the 32-column forward kernel (one wave per SIMD; f32 MFMA only; implies no Z1 tape), and the column-blocked gradient path that problems larger
    than the free HBM take (70 columns as blocks of 32 + 32 + 6 / 64 + 6 through one set of tapes)."""
    from colnde.nde import ENGINE_REGTILE
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    p = synthetic.wind_mixing_problem(70, n_frames=9, weight_divisor=1e2)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = np.array([1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3])
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(p.cfg, p.n_columns, engine=ENGINE_REGTILE, matrix_arithmetic=ma) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        sol_g = nde.forward(p.weights)
        tot_g, terms_g, grad_g = nde.loss_grad(p.weights, sc)
        plan = nde.plan()
    assert plan["matrix_arithmetic"] == ma
    want = {k: (env[e] == "1" if e in env else ma == "bf16x3_exact") for k, e in (("bf16x3_forward", "COLNDE_FWD_SPLIT"), ("bf16x3_adjoint", "COLNDE_ADJ_SPLIT"),
                                                                                   ("bf16x3_dw", "COLNDE_DW_SPLIT"))}
    if env.get("COLNDE_RT_FWD") == "32":
        want["bf16x3_forward"] = want["bf16x3_adjoint"] = False
    if env.get("COLNDE_RT_ZTAPE") == "0":
        want["bf16x3_adjoint"] = False
    assert {k: plan[k] for k in want} == want
    _record("test_regtile_alternative_paths_against_oracle" + "/" + ma + "/" + str(env), sol_abs=np.abs(sol_g - sol).max(), loss_rel=abs(tot_g - tot) / abs(tot), grad_rel=_rel(grad_g, g))
    assert np.abs(sol_g - sol).max() < SOL_ATOL
    np.testing.assert_allclose(terms_g, terms, rtol=LOSS_RTOL, atol=1e-12)
    assert _rel(grad_g, g) < GRAD_REL


def test_dw1_on_the_bf16_pipe_with_exact_operand_splitting_is_float32_grade(monkeypatch):
    """The dW1 GEMM alone on the split arithmetic: rt_dw1_split_kernel contracts the same two tapes with six v_mfma_f32_32x32x16_bf16 products of the exact
    three-way bf16 splits of both operands instead of v_mfma_f32_32x32x2_f32.  Same handle, same tapes: the layer-1 weight gradient of
    the two kernels differs by float32 round-off only (stated: 2e-6 relative L2 — the dropped cross terms are below 2^-23 per product),
    everything else in the gradient is bit-identical, and against the float64 oracle the split kernel is as close as the fp32 one
    (within 1.5x; both are recorded)."""
    from colnde.nde import ENGINE_REGTILE
    p = synthetic.wind_mixing_problem(200, n_frames=25, weight_divisor=1e2)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = np.array([1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3])
    tot, terms, g, _ = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(p.cfg, p.n_columns, engine=ENGINE_REGTILE) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        _arith(monkeypatch, nde)
        _, _, g32 = nde.loss_grad(p.weights, sc)
        _arith(monkeypatch, nde, dw=True)
        _, _, gsp = nde.loss_grad(p.weights, sc)
        assert nde.plan()["bf16x3_dw"] and not nde.plan()["bf16x3_forward"] and not nde.plan()["bf16x3_adjoint"]
    net = p.cfg.n_params // 3
    l1 = np.zeros(p.cfg.n_params, bool)
    for n in range(3):
        l1[n * net:n * net + 96 * 50 + 50] = True            # Flux.destructure: W1 (50 x 96), b1 (50) lead each net's block
    np.testing.assert_array_equal(gsp[~l1], g32[~l1])
    d = _rel(gsp[l1], g32[l1].astype(np.float64))
    e32, esp = _rel(g32[l1], g[l1]), _rel(gsp[l1], g[l1])
    _record("test_dw1_split", split_vs_fp32=d, fp32_vs_oracle=e32, split_vs_oracle=esp)
    assert d < 2e-6
    assert esp < max(1.5 * e32, 2e-6)
    assert _rel(gsp, g) < GRAD_REL


@pytest.mark.parametrize("name", [None, "diurnal", "relu", "mpp_bc_faces", "weights/4"])
def test_forward_nets_on_the_bf16_pipe_with_exact_operand_splitting_are_float32_grade(name, monkeypatch):
    """The forward solve alone on the split arithmetic: rt16_forward_kernel<ACT, true> evaluates the three flux nets with v_mfma_f32_16x16x32_bf16 on the exact three-way
    bf16 splits of weights (packed once) and activations (split in registers) — through the whole nonlinear solve, 2,304 stage evaluations at
    the bench horizon.  Same handle: the trajectory differs from the fp32-MFMA kernel's by float32 round-off (stated: 2e-5 in scaled units,
    the tolerance of the fp32 kernel itself against the oracle), and against the float64 oracle it is as close as the fp32 kernel (within 1.5x)."""
    from colnde.nde import ENGINE_REGTILE
    kw = VARIANTS[name] if name in VARIANTS else {}
    # "weights/4": nets 25x larger than in the other cases, so that their output is a leading term of the tendency
    p = synthetic.wind_mixing_problem(64, n_frames=289 if name is None else 33, weight_divisor=4.0 if name == "weights/4" else 1e2, **kw)
    sol = O.solve(p.cfg, p.x0, p.bcs, p.weights)
    assert np.isfinite(sol).all()
    with colnde.ColumnNDE(p.cfg, p.n_columns, engine=ENGINE_REGTILE) as nde:
        nde.set_problem(p.x0, p.bcs)
        _arith(monkeypatch, nde)
        s32 = nde.forward(p.weights)
        _arith(monkeypatch, nde, fwd=True)
        ssp = nde.forward(p.weights)
    d = np.abs(ssp - s32).max()
    e32, esp = np.abs(s32 - sol).max(), np.abs(ssp - sol).max()
    _record("test_fwd_split/" + str(name), split_vs_fp32=d, fp32_vs_oracle=e32, split_vs_oracle=esp)
    if name is None:
        assert d > 0.0                               # (the switch did select the other kernel; on 64 steps the two can round to the same trajectory)
    assert d < (LONG_SOL_ATOL if name is None else SOL_ATOL)
    assert esp < max(1.5 * e32, 2e-6)


@pytest.mark.parametrize("name", [None, "diurnal", "relu", "conv_adj_branch", "weights/4"])
def test_adjoint_w1t_products_on_the_bf16_pipe_with_exact_operand_splitting_are_float32_grade(name, monkeypatch):
    """The adjoint kernel alone on the split arithmetic: rt_adjoint_kernel<ACT, true, true> forms x̄ += W1ᵀ δz1 — 225 of the stage's 552 MFMAs — with v_mfma_f32_32x32x16_bf16 on the exact
    three-way splits (W1ᵀ planes h, m in LDS, l from L2; δz1 split in registers; features 48, 49 on one fp32 k-step).  x̄ feeds λ, so every later stage sees
    the difference.  Same handle, same tapes: the gradient differs from the fp32-MFMA kernel's by float32 round-off (stated: 2e-6 relative L2 on 24–288 frames),
    and against the float64 oracle it is as close (within 1.5x)."""
    from colnde.nde import ENGINE_REGTILE
    kw = VARIANTS[name] if name in VARIANTS else {}
    p = synthetic.wind_mixing_problem(96, n_frames=145 if name is None else 25, weight_divisor=4.0 if name == "weights/4" else 1e2, **kw)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = np.array([1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3])
    tot, terms, g, _ = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(p.cfg, p.n_columns, engine=ENGINE_REGTILE) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        _arith(monkeypatch, nde)
        t32, _, g32 = nde.loss_grad(p.weights, sc)
        _arith(monkeypatch, nde, adj=True)
        tsp, _, gsp = nde.loss_grad(p.weights, sc)
        again = nde.loss_grad(p.weights, sc)[2]
    assert t32 == tsp and np.array_equal(again, gsp)
    d = _rel(gsp, g32.astype(np.float64))
    e32, esp = _rel(g32, g), _rel(gsp, g)
    _record("test_adj_split/" + str(name), split_vs_fp32=d, fp32_vs_oracle=e32, split_vs_oracle=esp)
    assert 0.0 < d < 2e-6
    assert esp < max(1.5 * e32, 2e-6)


@pytest.mark.parametrize("rich", ["0", "1"])
def test_net_split_forward_on_the_bf16_pipe_with_exact_operand_splitting(rich, monkeypatch):
    """The split arithmetic under AUTO at a latency size, forward kernel alone: rt16sh_forward_kernel<ACT, RICH, false, true> (layers 1 and 2 of each net wave on
    v_mfma_f32_16x16x32_bf16 from exact three-way splits, the per-net operand image of rt_pack_split_ns_kernel with its quarter-filled fourth
    tile).  Same handle: trajectories within float32 round-off of the fp32-MFMA kernel and as close to the float64 oracle; the gradient taken
    from the split forward's tapes (plain and rich) stays within the net-split tolerances."""
    monkeypatch.setenv("COLNDE_T16_SPLIT_RICH", rich)
    p = synthetic.wind_mixing_problem(40, n_frames=33, weight_divisor=1e2)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = np.array([1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3])
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(p.cfg, p.n_columns) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        _arith(monkeypatch, nde)
        s32 = nde.forward(p.weights)
        t32, _, g32 = nde.loss_grad(p.weights, sc)
        _arith(monkeypatch, nde, fwd=True)
        ssp = nde.forward(p.weights)
        tsp, terms_sp, gsp = nde.loss_grad(p.weights, sc)
        plan = nde.plan()
    assert plan["split_forward"] and plan["split_adjoint"]
    d = np.abs(ssp - s32).max()
    e32, esp = np.abs(s32 - sol).max(), np.abs(ssp - sol).max()
    _record("test_net_split_fwd_split/rich" + rich, split_vs_fp32=d, fp32_vs_oracle=e32, split_vs_oracle=esp, grad_split_vs_fp32=_rel(gsp, g32.astype(np.float64)),
            grad_split_vs_oracle=_rel(gsp, g), grad_fp32_vs_oracle=_rel(g32, g))
    assert 0.0 < d < SOL_ATOL
    assert esp < max(1.5 * e32, 2e-6)
    np.testing.assert_allclose(terms_sp, terms, rtol=LOSS_RTOL, atol=1e-12)
    assert _rel(gsp, g) < GRAD_REL
    assert _rel(gsp, g) < max(1.5 * _rel(g32, g), 2e-6)


@pytest.mark.parametrize("name", [None, "diurnal", "relu", "conv_adj_branch", "weights/4"])
@pytest.mark.parametrize("rich", ["0", "1"])
def test_net_split_adjoint_on_the_bf16_pipe_with_exact_operand_splitting(rich, name, monkeypatch):
    """The adjoint kernel of the net-split pair alone on the split arithmetic: rt16sh_adjoint_kernel<ACT, RICH, false, true> forms every net wave's part
    of x̄, W1ₙᵀ δz1ₙ — 78 of a stage's 114 fp32 MFMAs — with v_mfma_f32_16x16x32_bf16 from exact three-way splits (W1ₙᵀ planes h, m in LDS, l from
    L2; δz1 split in registers, two 32-deep k-blocks with the padding quads zero).  x̄ feeds λ: every later stage sees the difference.  Same handle, same
    tapes (plain and rich): the gradient differs from the fp32-MFMA kernel's by float32 round-off (stated: 2e-6 relative L2), the loss is bit-identical
    (the forward is untouched), a repeat is bit-identical, and against the float64 oracle the split kernel is as close (within 1.5x)."""
    monkeypatch.setenv("COLNDE_T16_SPLIT_RICH", rich)
    kw = VARIANTS[name] if name in VARIANTS else {}
    p = synthetic.wind_mixing_problem(40, n_frames=73 if name is None else 25, weight_divisor=4.0 if name == "weights/4" else 1e2, **kw)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = np.array([1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3])
    tot, terms, g, _ = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(p.cfg, p.n_columns) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        _arith(monkeypatch, nde)
        t32, _, g32 = nde.loss_grad(p.weights, sc)
        plan32 = nde.plan()
        _arith(monkeypatch, nde, adj=True)
        tsp, _, gsp = nde.loss_grad(p.weights, sc)
        again = nde.loss_grad(p.weights, sc)[2]
        plan = nde.plan()
    assert plan["split_adjoint"] and plan["bf16x3_adjoint"] and not plan["bf16x3_forward"] and not plan["bf16x3_dw"] and not plan32["bf16x3_adjoint"]
    assert plan["split_rich_tape"] == (rich == "1")
    assert t32 == tsp and np.array_equal(again, gsp)
    d = _rel(gsp, g32.astype(np.float64))
    e32, esp = _rel(g32, g), _rel(gsp, g)
    _record("test_net_split_adj_split/rich%s/%s" % (rich, name), split_vs_fp32=d, fp32_vs_oracle=e32, split_vs_oracle=esp)
    assert 0.0 < d < 2e-6
    assert esp < max(1.5 * e32, 2e-6)


# ---- the net-split kernels of the latency points (engine AUTO up to 8,192 columns of the regtile shape) -----------------
@pytest.mark.parametrize("name", ["mpp_zero_weights", "mpp_bc_faces", "diurnal", "conv_adj_branch", "swish", "raw", "dRi_small", "relu",
                                  "tanh", "leakyrelu"])
def test_split_kernels_against_oracle_and_tile16(name, monkeypatch):
    """rt16s_forward_kernel + rt16s_adjoint_kernel (three wavefronts per 16-column tile, one per flux net) on 40 columns (two
    full tiles and a ragged one): against the float64 oracle at the usual tolerances, against pure tile16 (the same tapes and
    dW GEMM behind a different adjoint kernel) an order tighter, and bit-identical from one call to the next."""
    p = synthetic.wind_mixing_problem(40, n_frames=9, weight_divisor=1e2, **VARIANTS[name])
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = np.array([1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3])
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(p.cfg, p.n_columns) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        sol_s = nde.forward(p.weights)
        tot_s, terms_s, grad_s = nde.loss_grad(p.weights, sc)
        tot_2, terms_2, grad_2 = nde.loss_grad(p.weights, sc)
        plan = nde.plan()
    assert plan["split_forward"] and plan["split_adjoint"] and plan["dw_taped"] and plan["split_rich_tape"]
    monkeypatch.setenv("COLNDE_T16_FWD_SPLIT", "0")
    with colnde.ColumnNDE(p.cfg, p.n_columns) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        sol_t = nde.forward(p.weights)
        tot_t, terms_t, grad_t = nde.loss_grad(p.weights, sc)
        plan_t = nde.plan()
    assert not plan_t["split_forward"] and not plan_t["split_adjoint"]
    _record("test_split_kernels/" + str(name), sol_abs=np.abs(sol_s - sol).max(), loss_rel=abs(tot_s - tot) / abs(tot), grad_rel=_rel(grad_s, g),
            grad_rel_vs_tile16=_rel(grad_s, grad_t.astype(np.float64)), sol_abs_vs_tile16=np.abs(sol_s - sol_t).max())
    assert np.abs(sol_s - sol).max() < SOL_ATOL
    # (a term four orders below the total — the temperature terms of the convective-adjustment variant — is float32 cancellation:
    #  it gets the absolute slack of 1e-3 of the total's tolerance)
    np.testing.assert_allclose(terms_s, terms, rtol=LOSS_RTOL, atol=1e-3 * LOSS_RTOL * float(np.sum(terms)))
    assert np.isclose(tot_s, tot, rtol=LOSS_RTOL)
    assert _rel(grad_s, g) < GRAD_REL
    assert np.array_equal(grad_s, grad_2) and tot_s == tot_2
    assert np.abs(sol_s - sol_t).max() < 0.25 * SOL_ATOL
    assert _rel(grad_s, grad_t.astype(np.float64)) < 0.25 * GRAD_REL


@pytest.mark.parametrize("name", ["mpp_zero_weights", "conv_adj_branch", "raw", "relu"])
def test_split_kernels_plain_tape(name, monkeypatch):
    """COLNDE_T16_SPLIT_RICH=0: the net-split adjoint recomputes activation pairs and the physics closure from the pre-activation and stage
    tapes (what blocks above 2,048 columns take) instead of reading them from the rich tape; same oracle tolerances, and the two agree."""
    p = synthetic.wind_mixing_problem(40, n_frames=9, weight_divisor=1e2, **VARIANTS[name])
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = np.array([1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3])
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    res = {}
    for rich in ("1", "0"):
        monkeypatch.setenv("COLNDE_T16_SPLIT_RICH", rich)
        with colnde.ColumnNDE(p.cfg, p.n_columns) as nde:
            nde.set_problem(p.x0, p.bcs, truth)
            res[rich] = nde.loss_grad(p.weights, sc)
            plan = nde.plan()
            assert plan["split_adjoint"] and plan["split_rich_tape"] == (rich == "1")
        assert np.isclose(res[rich][0], tot, rtol=LOSS_RTOL)
        assert _rel(res[rich][2], g) < GRAD_REL
    assert _rel(res["1"][2], res["0"][2].astype(np.float64)) < 0.25 * GRAD_REL



# ---- config 2's and config 3's shipped kernels at THEIR size: AUTO at 4,096 columns = the net-split pair with the PLAIN tape --------------
def _assert_config2_plan(plan, grad=True):
    assert plan["engine"] == 1 and plan["split_forward"]
    if grad:
        assert plan["split_adjoint"] and plan["dw_taped"] and not plan["split_rich_tape"] and plan["block_columns"] > 2048


def test_auto_4096_columns_full_horizon_forward_against_oracle_on_sampled_columns():
    """BASELINE configs[1] as shipped: engine AUTO, 4,096 columns x 32 levels x 576 RK4 steps (`rt16sh_forward_kernel`, one 16-column tile
    per CU).  Columns are independent given the weights, so the float64 oracle integrates a sample of 64 of them — one from every 64th
    tile position, every lane of a tile covered — and the HIP trajectories of exactly those columns must match over the whole horizon."""
    p = synthetic.wind_mixing_problem(4096, n_frames=289, weight_divisor=1e2)
    idx = np.arange(64) * 64 + (np.arange(64) * 5) % 64              # 64 distinct tiles, all 16 in-tile positions four times
    sol = O.solve(p.cfg, p.x0[idx], p.bcs[idx], p.weights)
    with colnde.ColumnNDE(p.cfg, 4096) as nde:
        nde.set_problem(p.x0, p.bcs)
        sol_g = nde.forward(p.weights)
        _assert_config2_plan(nde.plan(), grad=False)
    _record("auto_4096x576/forward_sampled", sol_abs=np.abs(sol_g[idx] - sol).max())
    assert np.isfinite(sol_g).all()
    assert np.abs(sol_g[idx] - sol).max() < LONG_SOL_ATOL


def test_auto_4096_columns_full_horizon_gradient_against_oracle_on_replicated_suite():
    """The same kernels' gradient path at 4,096 columns x 576 steps (plain tape: the rich one stops at 2,048 columns) against the float64
    oracle.  The loss is a MEAN over simulations (NDE_training.jl:312-317): 64 distinct columns dealt round-robin 64 times over the 4,096
    slots have exactly the loss terms and gradient of the 64, which is what the oracle evaluates — while the GPU runs config-size grids,
    tapes and dW GEMM slices."""
    q = synthetic.wind_mixing_problem(64, n_frames=289, weight_divisor=1e2)
    truth = O.solve(q.cfg, q.x0, q.bcs, q.weights_truth).astype(np.float32)
    sc = O.default_loss_scalings(q.cfg)
    tot, terms, g, sol = O.loss_and_grad(q.cfg, q.x0, q.bcs, q.weights, truth, sc)
    rep = np.arange(4096) % 64
    with colnde.ColumnNDE(q.cfg, 4096) as nde:
        nde.set_problem(q.x0[rep], q.bcs[rep], truth[rep])
        tot_g, terms_g, grad_g = nde.loss_grad(q.weights, sc)
        _assert_config2_plan(nde.plan())
    _record("auto_4096x576/gradient_replicated", loss_rel=abs(tot_g - tot) / tot, terms_rel=np.abs(terms_g / terms - 1).max(), grad_rel=_rel(grad_g, g))
    np.testing.assert_allclose(terms_g, terms, rtol=LONG_TERMS_RTOL[1e2], atol=0)
    assert np.isclose(tot_g, tot, rtol=LONG_LOSS_RTOL[1e2], atol=0)
    assert _rel(grad_g, g) < LONG_GRAD_REL[1e2]


def test_auto_4096_distinct_columns_short_horizon_gradient_against_oracle():
    """4,096 DISTINCT columns (every tile different, > 2,048: plain tape) over 16 frames: the oracle evaluates all of them."""
    p = synthetic.wind_mixing_problem(4096, n_frames=17, weight_divisor=1e2)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = np.array([1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3])
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(p.cfg, 4096) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        sol_g = nde.forward(p.weights)
        tot_g, terms_g, grad_g = nde.loss_grad(p.weights, sc)
        _assert_config2_plan(nde.plan())
    _record("auto_4096x32/all_columns", sol_abs=np.abs(sol_g - sol).max(), loss_rel=abs(tot_g - tot) / tot, grad_rel=_rel(grad_g, g))
    assert np.abs(sol_g - sol).max() < SOL_ATOL
    np.testing.assert_allclose(terms_g, terms, rtol=LOSS_RTOL, atol=0)
    assert _rel(grad_g, g) < GRAD_REL


@pytest.mark.parametrize("switch", ["COLNDE_T16_FWD_HELPER", "COLNDE_T16_ADJ_HELPER"])
@pytest.mark.parametrize("rich", ["1", "0"])
def test_split_kernels_without_helper_wave(switch, rich, monkeypatch):
    """COLNDE_T16_FWD_HELPER=0 / COLNDE_T16_ADJ_HELPER=0: the three-wave kernels (every net wave evaluates the Richardson-number closure,
    resp. carries λ, x̄ and the physics pullback itself) give the trajectory and gradient of the default four-wave ones (a helper wave on
    the fourth SIMD does that once and hands the result over through LDS), with the rich and with the plain tape."""
    monkeypatch.setenv("COLNDE_T16_SPLIT_RICH", rich)
    for name in ("mpp_zero_weights", "conv_adj_branch", "raw", "diurnal"):
        p = synthetic.wind_mixing_problem(40, n_frames=9, weight_divisor=1e2, **VARIANTS[name])
        truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
        sc = np.array([1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3])
        res = {}
        for hw in ("1", "0"):
            monkeypatch.setenv(switch, hw)
            with colnde.ColumnNDE(p.cfg, p.n_columns) as nde:
                nde.set_problem(p.x0, p.bcs, truth)
                res[hw] = (nde.forward(p.weights), nde.loss_grad(p.weights, sc))
        assert np.abs(res["1"][0] - res["0"][0]).max() < 0.25 * SOL_ATOL, name
        assert _rel(res["1"][1][2], res["0"][1][2].astype(np.float64)) < 0.25 * GRAD_REL, name


def test_split_kernels_column_blocked(monkeypatch):
    """The column-blocked gradient path (tapes that hold one block: what a problem larger than the free HBM takes) through the net-split
    kernels: 40 columns as blocks of 16 + 16 + 8 give the gradient of the unblocked run (same kernels, same reduction order per row)."""
    p = synthetic.wind_mixing_problem(40, n_frames=9, weight_divisor=1e2)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = np.array([1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3])
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    res = []
    for blk in (None, "16"):
        if blk:
            monkeypatch.setenv("COLNDE_T16_BLOCK", blk)
        with colnde.ColumnNDE(p.cfg, p.n_columns) as nde:
            nde.set_problem(p.x0, p.bcs, truth)
            res.append(nde.loss_grad(p.weights, sc))
            plan = nde.plan()
            assert plan["split_adjoint"] and plan["n_blocks"] == (3 if blk else 1)
    assert _rel(res[1][2], g) < GRAD_REL and np.isclose(res[1][0], tot, rtol=LOSS_RTOL)
    assert _rel(res[1][2], res[0][2].astype(np.float64)) < 1e-6


def test_split_adjoint_behind_tile16_forward(monkeypatch):
    """COLNDE_T16_ADJ_SPLIT=0 keeps tile16's adjoint behind the split forward (the round-2 intermediate); both gradient paths read
    the same tapes, so they agree far inside the oracle tolerance."""
    p = synthetic.wind_mixing_problem(24, n_frames=9, weight_divisor=1e2)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = np.array([1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3])
    grads = []
    for adj in ("1", "0"):
        monkeypatch.setenv("COLNDE_T16_ADJ_SPLIT", adj)
        with colnde.ColumnNDE(p.cfg, p.n_columns) as nde:
            nde.set_problem(p.x0, p.bcs, truth)
            grads.append(nde.loss_grad(p.weights, sc))
            assert nde.plan()["split_forward"] and nde.plan()["split_adjoint"] == (adj == "1")
    assert np.isclose(grads[0][0], grads[1][0], rtol=1e-6)
    assert _rel(grads[0][2], grads[1][2].astype(np.float64)) < 0.25 * GRAD_REL


def test_three_gradient_paths_agree_on_random_cases(monkeypatch):
    """Differential fuzz (tools/fuzz_engines.py): engine AUTO (net-split kernels, rich or plain tape), tile16 and regtile on 32 random
    combinations of physics variant, activation, column count (1 .. 257), frame count, sub-step count and loss scalings, one truth for all
    three.  Measured worst disagreement with tile16 over 60 cases: solution 3.7e-6, loss 1.4e-6, gradient 2.6e-6 (relative L2)."""
    rng = np.random.default_rng(11)
    names = ["mpp_zero_weights", "mpp_bc_faces", "diurnal", "conv_adj_branch", "swish", "raw", "dRi_small", "relu", "tanh", "leakyrelu"]
    done = 0
    while done < 32:
        name = names[rng.integers(len(names))]
        ncol = int(rng.choice([1, 3, 8, 16, 17, 40, 64, 100, 257]))
        p = synthetic.wind_mixing_problem(ncol, n_frames=int(rng.choice([2, 3, 5, 9, 17])), weight_divisor=1e2, **VARIANTS[name])
        cfg = p.cfg.with_(substeps=max(int(rng.choice([2, 3, 5])), p.cfg.substeps))
        sc = np.concatenate([rng.uniform(0.5, 1.5, 3), rng.uniform(0, 1e-2, 3) * rng.integers(2)])
        monkeypatch.setenv("COLNDE_T16_SPLIT_RICH", str(int(rng.integers(2))))
        res, truth = {}, None
        for label, eng in (("tile16", 1), ("auto", 0), ("regtile", 2)):
            with colnde.ColumnNDE(cfg, ncol, engine=eng) as nde:
                nde.set_problem(p.x0, p.bcs)
                if truth is None:
                    truth = nde.forward(p.weights_truth)
                nde.set_problem(p.x0, p.bcs, truth)
                sol = nde.forward(p.weights)
                tot, terms, g = nde.loss_grad(p.weights, sc)
                res[label] = (sol, tot, g.astype(np.float64), nde.plan())
        assert res["auto"][3]["split_adjoint"] and not res["tile16"][3]["split_adjoint"]
        for label in ("auto", "regtile"):
            assert np.abs(res[label][0] - res["tile16"][0]).max() < SOL_ATOL, (name, ncol, label)
            assert np.isclose(res[label][1], res["tile16"][1], rtol=2e-5), (name, ncol, label)
            assert _rel(res[label][2], res["tile16"][2]) < 3e-5, (name, ncol, label)
        done += 1


# ---- edge cases: ragged/minimal shapes, non-uniform time axis, odd sub-step counts, both engines ---------------------
@pytest.mark.parametrize("engine", [0, 1, 2])          # 0 = AUTO: the net-split kernels at these sizes
@pytest.mark.parametrize("n_col", [1, 15, 31, 33])
def test_edge_columns_nonuniform_times(engine, n_col):
    p = synthetic.wind_mixing_problem(n_col, n_frames=5, weight_divisor=1e2)
    # non-uniform save times and 3 sub-steps per interval (train_tranges such as 1:20:200 give uniform spacing; the ABI
    # takes any increasing `saveat`)
    cfg = p.cfg.with_(save_times=(0.0, 0.004, 0.005, 0.011, 0.0125), substeps=3)
    truth = O.solve(cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = np.array([1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3])
    tot, terms, g, sol = O.loss_and_grad(cfg, p.x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(cfg, n_col, engine=engine) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        sol_g = nde.forward(p.weights)
        tot_g, terms_g, grad_g = nde.loss_grad(p.weights, sc)
        tot_l, terms_l = nde.loss(p.weights, sc)
    _record("test_edge_columns_nonuniform_times" + "/" + str(engine) + "/" + str(n_col), sol_abs=np.abs(sol_g - sol).max(), loss_rel=abs(tot_g - tot) / abs(tot), grad_rel=_rel(grad_g, g))
    assert np.abs(sol_g - sol).max() < SOL_ATOL
    assert np.isclose(tot_g, tot, rtol=LOSS_RTOL) and np.isclose(tot_l, tot, rtol=LOSS_RTOL)
    assert _rel(grad_g, g) < GRAD_REL


@pytest.mark.parametrize("engine", [0, 1, 2])
def test_zero_weights_give_zero_weight_gradient_blocks(engine):
    """With all weights zero every hidden activation of a mish net is 0, so dW2, dW3 vanish identically while db3 and
    (through mish'(0) = 0.6) the other gradients do not: checks the per-layer block placement in Flux.destructure order."""
    p = synthetic.wind_mixing_problem(40, n_frames=5)
    cfg = p.cfg
    w0 = np.zeros(cfg.n_params, np.float32)
    truth = O.solve(cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    sc = [1, 1, 1, 5e-3, 5e-3, 5e-3]
    _, _, g_ref, _ = O.loss_and_grad(cfg, p.x0, p.bcs, w0, truth, sc)
    with colnde.ColumnNDE(cfg, 40, engine=engine) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        _, _, g = nde.loss_grad(w0, sc)
    ns = cfg.net_size
    for n in range(3):
        blk = g[n * ns:(n + 1) * ns]
        W1, b1 = blk[:4800], blk[4800:4850]
        W2, b2 = blk[4850:5850], blk[5850:5870]
        W3, b3 = blk[5870:6490], blk[6490:6521]
        assert np.all(W1 == 0) and np.all(b1 == 0) and np.all(W2 == 0) and np.all(b2 == 0) and np.all(W3 == 0)
        assert np.abs(b3).max() > 0
    _record("zero_weights/engine%d" % engine, grad_rel=_rel(g, g_ref))
    # truth here is the trajectory of weights/1e5: the loss (and with it the gradient) is float32 round-off of the profiles,
    # as in the 576-step case with the bench's weights (measured 1.6e-3)
    assert _rel(g, g_ref) < LONG_GRAD_REL[1e5]


# ---- the generic engine on shapes nobody tuned for ----------------------------------------------------------------
_RANDOM_SHAPES = [
    # (model, Nz, hidden sizes, hidden activations, columns)
    ("wm", 8, (7,), ("tanh",), 5),
    ("wm", 16, (33, 9), ("swish", "relu"), 19),
    ("wm", 24, (17, 40, 5), ("mish", "leakyrelu", "tanh"), 33),
    ("wm", 48, (64,), ("relu",), 17),
    ("fc", 12, (20, 20), ("relu", "relu"), 16),
    ("fc", 40, (70, 11, 30), ("tanh", "mish", "swish"), 23),
    ("fc", 96, (48,), ("leakyrelu",), 9),
    ("ca", 20, (16, 16), ("relu", "tanh"), 31),
]


@pytest.mark.parametrize("mode", ["taped", "inregister", "l2stream"])
@pytest.mark.parametrize("model,Nz,hidden,acts,ncol", _RANDOM_SHAPES)
def test_tile16_untuned_shapes(model, Nz, hidden, acts, ncol, mode, monkeypatch):
    """Layer counts 2..4, widths that are not multiples of the 16-row MFMA tile, mixed activations, Nz from 8 to 96, ragged column
    counts: solution, loss terms and gradient of the generic engine (taped gradient path) against the float64 oracle."""
    if mode == "inregister":
        monkeypatch.setenv("COLNDE_T16_DWTAPE", "0")
    if mode == "l2stream":
        monkeypatch.setenv("COLNDE_T16_DWLDS", "0")
    if model == "wm":
        sizes = (3 * Nz,) + tuple(hidden) + (Nz - 1,)
        p = synthetic.wind_mixing_problem(ncol, Nz=Nz, n_frames=5, weight_divisor=1e2, layer_sizes=sizes,
                                          activations=tuple(acts) + ("identity",), substeps=4 if Nz > 32 else 2)  # 48 levels: min_substeps = 3
        sc = np.array([1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3])
    else:
        sizes = (Nz,) + tuple(hidden) + (Nz - 1,)
        p = synthetic.free_convection_problem(ncol, Nz=Nz, n_save=5, substeps=32 if model == "ca" else 2, t_end=0.01,
                                              convective_adjustment=(model == "ca"), layer_sizes=sizes,
                                              activations=tuple(acts) + ("identity",))
        sc = O.default_loss_scalings(p.cfg)
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    tot, terms, g, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    with colnde.ColumnNDE(p.cfg, p.n_columns) as nde:
        nde.set_problem(p.x0, p.bcs, truth)
        sol_g = nde.forward(p.weights)
        tot_g, terms_g, grad_g = nde.loss_grad(p.weights, sc)
        dx_g = nde.rhs(p.x0, p.weights, p.bcs, 0.01)
    _record("test_tile16_untuned_shapes" + "/" + str(model) + "/" + str(Nz) + "/" + str(hidden) + "/" + str(acts) + "/" + str(ncol) + "/" + str(mode), sol_abs=np.abs(sol_g - sol).max(), loss_rel=abs(tot_g - tot) / abs(tot), grad_rel=_rel(grad_g, g))
    assert np.abs(sol_g - sol).max() < (SOL_ATOL if model == "wm" else FC_SOL_ATOL)
    assert np.isclose(tot_g, tot, rtol=LOSS_RTOL if model == "wm" else FC_LOSS_RTOL)
    assert _rel(grad_g, g) < (GRAD_REL if model == "wm" else FC_GRAD_REL)
    assert _rel(dx_g, O.rhs(p.cfg, p.x0, p.bcs, p.weights, 0.01)) < 2e-6
