"""Training-TRAJECTORY parity (VERDICT r4 task 3): the reference's product is the optimiser loop, not one gradient.

The GPU loops — `train_NDE` (host-side ADAM over `colnde_loss_grad`), `train_NDE_device` (θ and the ADAM state on the card:
`colnde_loss_grad_dev` + `colnde_adam_step_dev`), `train_neural_differential_equation[_device]` — against oracle/training_oracle.py: the same
loops in float64 over the float64 oracle (GalacticOptim's save_best solve with Flux's ADAM, wind_mixing/src/NDE_training.jl:340-372;
`Flux.train!`, free_convection/src/training.jl:44-74).  Settings are the reference's own (wind_mixing/train_NDE.jl:110,138-141): 8 simulations,
`train_tranges = [1:20:200]` (10 save points 20 frames apart; 40 RK4 sub-steps per interval = the bench's step), `train_iterations = [5]`,
`ADAM(3e-4)`, `training_fractions = (T = 0.8, ∂T∂z = 0.8, profile = 0.5)`, networks initialised as `re(weights ./ 1f5)` (:105-107).

What is compared: the loss of every iteration (the sequence a training log holds), the loss scalings, and θ after the last iteration
(`res.minimizer`).  Tolerances are ~10x the measured disagreement (recorded with COLNDE_RECORD_ERRORS=1 like the other parity files).

Why θ is not held to float32 round-off: ADAM's step is η·m̂/(√v̂ + ϵ).  For |g| >> ϵ = 1e-8 it is η·sign(g) and blind to float32 error; for
|g| << ϵ it is η·g/ϵ, proportional to it; in between (|g| ~ ϵ) an absolute gradient error δ moves the step by up to η·δ/ϵ.  A float32
gradient's absolute error against float64 is ~1e-10 here, so the affected elements move by ~η·1e-2: the test bounds the relative L2 distance
of the whole vector and separately the largest single-element distance in units of η."""
import numpy as np
import pytest

import colnde
from colnde import synthetic
from colnde.flux_compat import ADAM
from colnde.wind_mixing import WindMixingNDE, train_NDE, train_NDE_device
from colnde.free_convection import (FreeConvectionNDE, train_neural_differential_equation,
                                    train_neural_differential_equation_device)
from oracle import nde_oracle as O
from oracle import training_oracle as TO

from tests.test_gpu_parity import _record

pytestmark = pytest.mark.gpu

FRACTIONS = dict(T=0.8, dTdz=0.8, profile=0.5)                  # train_NDE.jl:110
ETA, ITERS = 3e-4, 5                                            # train_NDE.jl:140-141
LOSS_SEQ_RTOL = 1e-4                                            # every iteration's loss against the float64 loop
SCALINGS_RTOL = 1e-4
THETA_REL_L2 = {"init_1e5": 4e-4, "init_1e2": 1.5e-4}           # ‖θ_K − θ_K^oracle‖ / ‖θ_K^oracle‖ (measured 4.2e-5 / 1.4e-5: profiles/r05_parity_errors.json)
THETA_MAX_STEP_UNITS = 0.05                                     # no single weight ends more than η/20 from the oracle's (measured 0.001 η / 0.003 η)


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300)


def _reference_setting(init):
    """8 simulations, tsteps 1:20:200, 40 sub-steps per save interval.  init_1e5: the reference's `weights ./ 1f5` start against a truth that
    is a finite distance away (the trajectory of a perturbed weights/1e2 set: a loss of O(1e-5), as LES truth gives the reference);
    init_1e2: nets that matter from the first iteration on."""
    div = 1e5 if init == "init_1e5" else 1e2
    p = synthetic.wind_mixing_problem(8, n_frames=10, frame_stride=20, substeps=40, weight_divisor=div)
    assert p.cfg.n_save == 10 and p.cfg.n_steps == 360
    wt = synthetic.wind_mixing_problem(8, n_frames=10, frame_stride=20, substeps=40, weight_divisor=1e2).weights_truth
    truth = O.solve(p.cfg, p.x0, p.bcs, wt).astype(np.float32)
    return p, truth


@pytest.mark.parametrize("init", ["init_1e5", "init_1e2"])
@pytest.mark.parametrize("ma", ["bf16x3_exact", "f32_mfma"])
def test_train_NDE_trajectory_matches_the_float64_loop(init, ma):
    p, truth = _reference_setting(init)
    sc_o = TO.initial_loss_scalings(p.cfg, p.x0, p.bcs, p.weights, truth, FRACTIONS)
    theta_o, hist_o = TO.train_NDE(p.cfg, p.x0, p.bcs, truth, p.weights, sc_o, [ETA], epochs=1, maxiters=ITERS)
    loss_o = np.array([h["total"] for h in hist_o])
    for name, loop in (("host", train_NDE), ("device", train_NDE_device)):
        wm = WindMixingNDE(p.cfg, p.x0, p.bcs, truth, training_fractions=FRACTIONS, weights0=p.weights, matrix_arithmetic=ma)
        try:
            assert wm.engine.matrix_arithmetic == ma
            np.testing.assert_allclose(wm.loss_scalings, sc_o, rtol=SCALINGS_RTOL)
            res = loop(wm, p.weights, [ADAM(ETA)], epochs=1, maxiters=ITERS)
        finally:
            wm.close()
        loss_g = np.array([h["total"] for h in res.history])
        assert len(loss_g) == ITERS
        e_loss = np.abs(loss_g / loss_o - 1).max()
        e_theta = _rel(res.weights, theta_o)
        e_max = np.abs(res.weights.astype(np.float64) - theta_o).max() / ETA
        _record("training/train_NDE_%s/%s/%s" % (name, init, ma), loss_seq_rel=e_loss, theta_rel_l2=e_theta, theta_max_in_eta=e_max,
                scalings_rel=np.abs(wm.loss_scalings / sc_o - 1).max())
        assert e_loss < LOSS_SEQ_RTOL, (name, loss_g, loss_o)
        assert e_theta < THETA_REL_L2[init], (name, e_theta)
        assert e_max < THETA_MAX_STEP_UNITS, (name, e_max)
        # the six scaled terms of every logged iteration (what write_data_NDE_training stores, data_writing.jl:28-78)
        for h, ho in zip(res.history, hist_o):
            got = np.array([h[k] for k in ("u", "v", "T", "dudz", "dvdz", "dTdz")])
            np.testing.assert_allclose(got, ho["terms"], rtol=10 * LOSS_SEQ_RTOL)
        # the training did something: θ moved by ~η per weight, and the save_best point is not the start
        assert np.abs(res.weights - p.weights).max() > 0.5 * ETA


def test_two_optimizers_two_epochs_follow_the_float64_loop():
    """`for opt in optimizers, epoch in 1:epochs` (NDE_training.jl:340-372): every solve restarts ADAM's moments and reverts to its best point."""
    p, truth = _reference_setting("init_1e2")
    sc = O.default_loss_scalings(p.cfg)
    theta_o, hist_o = TO.train_NDE(p.cfg, p.x0, p.bcs, truth, p.weights, sc, [3e-4, 1e-4], epochs=2, maxiters=3)
    wm = WindMixingNDE(p.cfg, p.x0, p.bcs, truth)
    try:
        rh = train_NDE(wm, p.weights, [ADAM(3e-4), ADAM(1e-4)], epochs=2, maxiters=3)
        rd = train_NDE_device(wm, p.weights, [ADAM(3e-4), ADAM(1e-4)], epochs=2, maxiters=3)
    finally:
        wm.close()
    lo = np.array([h["total"] for h in hist_o])
    for r in (rh, rd):
        assert len(r.history) == 12
        np.testing.assert_allclose([h["total"] for h in r.history], lo, rtol=LOSS_SEQ_RTOL)
        assert _rel(r.weights, theta_o) < THETA_REL_L2["init_1e2"]
        assert np.abs(r.weights.astype(np.float64) - theta_o).max() < 0.25 * 3e-4          # four solves, 12 updates: measured 0.06 η


@pytest.mark.parametrize("Nz,ca", [(32, False), (32, True)])
def test_flux_train_trajectory_matches_the_float64_loop(Nz, ca):
    """`train_neural_differential_equation!` (free_convection/src/training.jl:44-74): 6 epochs of ADAM(1e-3) on the single MSE of the
    concatenated solutions, FreeConvectionNDE and ConvectiveAdjustmentNDE, host loop and device loop."""
    p = synthetic.free_convection_problem(6, Nz=Nz, n_save=9, substeps=4, t_end=0.0625, convective_adjustment=ca)
    if ca:              # K = 10 is stiff (lambda dt = 40 here): the stabilised stepper, as the reference reaches for ROCK4 (test_free_convection_nde.jl:32-35)
        p.cfg = p.cfg.with_(stepper="rkc2")
        p.x0[:2, 10:22] = p.x0[:2, 10:22][:, ::-1].copy()       # two columns with an inverted layer: the min(0, K dT/dz) switch is live
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    theta_o, hist_o = TO.train_neural_differential_equation(p.cfg, p.x0, p.bcs, truth, p.weights, 1e-3, 6)
    gap_loss = gap_theta = 0.0
    if ca:
        # Through the live switch float32 ITSELF parts from float64 (the side of min(0, K dT/dz) is decided by round-off on the faces of an
        # adjusted layer): the same loop with a float32 solve + adjoint sets the scale, as in tests/test_gpu_fc.py (measured: 1.7e-2 on the loss
        # sequence, 8.7e-2 on θ after 6 epochs — and the HIP path lands on the float32 loop, 1e-4 / 1e-2 from it)
        theta_32, hist_32 = TO.train_neural_differential_equation(p.cfg, p.x0, p.bcs, truth, p.weights, 1e-3, 6, dtype=np.float32)
        gap_loss = np.abs(np.array(hist_32) / np.array(hist_o) - 1).max()
        gap_theta = _rel(theta_32, theta_o)
    for name, loop in (("host", train_neural_differential_equation), ("device", train_neural_differential_equation_device)):
        nde = FreeConvectionNDE(p.cfg, p.x0, p.bcs, truth)
        try:
            theta, hist = loop(nde, p.weights, ADAM(1e-3), 6)
        finally:
            nde.close()
        e_loss = np.abs(np.array(hist) / np.array(hist_o) - 1).max()
        e_theta = _rel(theta, theta_o)
        rec = dict(loss_seq_rel=e_loss, theta_rel_l2=e_theta)
        if ca:
            rec.update(oracle32_loss_seq_rel=gap_loss, oracle32_theta_rel_l2=gap_theta,
                       vs_oracle32_loss_seq_rel=np.abs(np.array(hist) / np.array(hist_32) - 1).max(), vs_oracle32_theta_rel_l2=_rel(theta, theta_32))
        _record("training/flux_train_%s/Nz%d_ca%d" % (name, Nz, ca), **rec)
        assert e_loss < 3e-3 + 2 * gap_loss, (name, hist, hist_o)  # the free-convection loss tolerance of tests/test_gpu_parity.py (FC_LOSS_RTOL) [+ float32's own gap]
        assert e_theta < 2e-3 + 2 * gap_theta, (name, e_theta)
        if ca:
            # against the float32 loop: the first epochs' losses (measured 8e-5, 3e-5) — later ones drift apart as each loop's own switch flips feed its ADAM state
            # (measured 8e-4 .. 4e-3 by epoch 6; θ 4e-2), both well inside float32's own distance from float64
            d32 = np.abs(np.array(hist) / np.array(hist_32) - 1)
            assert d32[:2].max() < 1e-3 and d32.max() < 2e-2, (name, hist, hist_32)
        assert hist_o[-1] < hist_o[0]
