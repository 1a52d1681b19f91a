"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the float64 oracle):
the oracle and its C port must keep reproducing them (CPU), and the HIP engine must match them (GPU)."""
import os

import numpy as np
import pytest

import colnde
from colnde import synthetic
from oracle import nde_oracle as O, cref

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = {
    "wind_mixing_mpp": lambda: synthetic.wind_mixing_problem(4, n_frames=9, weight_divisor=1e2),
    "wind_mixing_diurnal_smooth": lambda: synthetic.wind_mixing_problem(3, n_frames=5, weight_divisor=1e2, diurnal=True,
                                                                       smooth_NN=True, smooth_Ri=True),
    "free_convection_32": lambda: synthetic.free_convection_problem(3, Nz=32, n_save=5, substeps=2, t_end=0.02),
    "conv_adj_nde_32": lambda: synthetic.free_convection_problem(3, Nz=32, n_save=5, substeps=20, t_end=0.01,
                                                                 convective_adjustment=True),
}


def _rel(a, b):
    return np.linalg.norm(np.asarray(a, np.float64) - b) / (np.linalg.norm(b) + 1e-300)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_and_inputs_reproduce_golden(name):
    g = np.load(os.path.join(HERE, name + ".npz"))
    p = CASES[name]()
    np.testing.assert_array_equal(p.x0, g["x0"])            # the synthetic generator is part of the contract
    np.testing.assert_array_equal(p.weights, g["weights"])
    tot, terms, grad, sol = O.loss_and_grad(p.cfg, g["x0"], g["bcs"], g["weights"], g["truth"], g["scalings"])
    np.testing.assert_allclose(sol, g["sol"], atol=2e-6)
    np.testing.assert_allclose(terms, g["terms"], rtol=1e-9)
    assert _rel(grad, g["grad"].astype(np.float64)) < 1e-6


@pytest.mark.parametrize("name", sorted(CASES))
def test_c_port_matches_golden(name):
    g = np.load(os.path.join(HERE, name + ".npz"))
    p = CASES[name]()
    tot, terms, grad, sol = cref.loss_grad(p.cfg, g["x0"], g["bcs"], g["weights"], g["truth"], g["scalings"], want_sol=True)
    assert np.abs(sol - g["sol"]).max() < 1e-4
    assert np.isclose(tot, float(g["total"]), rtol=1e-3)
    assert _rel(grad, g["grad"].astype(np.float64)) < 2e-3
    assert _rel(cref.rhs(p.cfg, g["x0"], g["bcs"], g["weights"], float(g["rhs_t"])), g["rhs"].astype(np.float64)) < 2e-5


# the tolerances of tests/test_gpu_parity.py for the same families (about 10x the error measured on an MI355X, profiles/r03_parity_errors.json):
# (solution abs, loss/terms rel, gradient rel L2, one RHS rel)
GOLDEN_TOL = {
    "wind_mixing_mpp": (2e-5, 8e-5, 2e-4, 1e-6),
    "wind_mixing_diurnal_smooth": (2e-5, 8e-5, 2e-4, 1e-6),
    "free_convection_32": (8e-5, 3e-3, 4e-3, 1e-6),
    "conv_adj_nde_32": (8e-5, 3e-3, 4e-3, 1e-6),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_engine_matches_golden(name):
    g = np.load(os.path.join(HERE, name + ".npz"))
    p = CASES[name]()
    sol_atol, loss_rtol, grad_rel, rhs_rel = GOLDEN_TOL[name]
    with colnde.ColumnNDE(p.cfg, g["x0"].shape[0]) as nde:
        nde.set_problem(g["x0"], g["bcs"], g["truth"])
        sol = nde.forward(g["weights"])
        tot, terms, grad = nde.loss_grad(g["weights"], g["scalings"])
        dx = nde.rhs(g["x0"], g["weights"], g["bcs"], float(g["rhs_t"]))
    from tests.test_gpu_parity import _record
    _record("golden/" + name, sol_abs=np.abs(sol - g["sol"]).max(), loss_rel=abs(tot - float(g["total"])) / abs(float(g["total"])),
            grad_rel=_rel(grad, g["grad"].astype(np.float64)), rhs_rel=_rel(dx, g["rhs"].astype(np.float64)))
    assert np.abs(sol - g["sol"]).max() < sol_atol                  # float32 engine vs float64 oracle, O(1) profiles
    np.testing.assert_allclose(terms, g["terms"], rtol=loss_rtol, atol=1e-3 * loss_rtol * float(np.sum(g["terms"])))
    assert np.isclose(tot, float(g["total"]), rtol=loss_rtol)
    assert _rel(grad, g["grad"].astype(np.float64)) < grad_rel
    assert _rel(dx, g["rhs"].astype(np.float64)) < rhs_rel


@pytest.mark.gpu
def test_hip_infer_forcing_matches_golden():
    g = np.load(os.path.join(HERE, "infer_forcing_32.npz"))
    cfg, T, top, w = synthetic.inference_problem(6, 5)
    with colnde.ColumnNDE(cfg, T.shape[0]) as nde:
        out = nde.infer_forcing(g["weights"], g["T"], g["top_flux"], float(g["Lz"]))
    assert _rel(out, g["forcing"].astype(np.float64)) < 1e-4


def test_infer_golden_oracle():
    g = np.load(os.path.join(HERE, "infer_forcing_32.npz"))
    cfg, T, top, w = synthetic.inference_problem(6, 5)
    np.testing.assert_allclose(O.infer_forcing(cfg, g["T"], g["top_flux"], g["weights"], float(g["Lz"])), g["forcing"], rtol=1e-5, atol=1e-9)


def test_column_ops_golden_oracle():
    g = np.load(os.path.join(HERE, "column_ops.npz"))
    args = (g["T"], float(g["dt"]), float(g["dz"]), float(g["K"]))
    np.testing.assert_allclose(O.convective_adjustment(*args), g["T_adjusted"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(O.convective_adjustment(*args, g["halo_bottom"], g["halo_top"]), g["T_adjusted_halo"], rtol=1e-12, atol=1e-12)
    assert not np.allclose(g["T_adjusted"], g["T"])
    th, m, v, bt = g["adam_theta0"], np.zeros(101), np.zeros(101), (0.9, 0.999)
    for i in range(3):
        th, m, v, bt = O.adam_step(th, g["adam_grads"][i], m, v, 1e-3, (0.9, 0.999), 1e-8, bt)
        np.testing.assert_allclose(th, g["adam_thetas"][i], rtol=1e-13)


@pytest.mark.gpu
def test_hip_column_ops_match_golden():
    import torch
    g = np.load(os.path.join(HERE, "column_ops.npz"))
    cfg = synthetic.free_convection_problem(1, Nz=32, n_save=2).cfg
    with colnde.ColumnNDE(cfg, 1) as nde:
        out = nde.convective_adjustment(g["T"], float(g["dt"]), float(g["dz"]), float(g["K"]))
        assert _rel(out, g["T_adjusted"]) < 2e-5
        out = nde.convective_adjustment(g["T"], float(g["dt"]), float(g["dz"]), float(g["K"]), g["halo_bottom"], g["halo_top"])
        assert _rel(out, g["T_adjusted_halo"]) < 2e-5
        th = torch.from_numpy(g["adam_theta0"].astype(np.float32)).cuda()
        m, v, bt = torch.zeros(101, device="cuda"), torch.zeros(101, device="cuda"), [0.9, 0.999]
        for i in range(3):
            nde.adam_step(th, torch.from_numpy(g["adam_grads"][i].astype(np.float32)).cuda(), m, v, 1e-3, beta_t=tuple(bt))
            bt = [bt[0] * 0.9, bt[1] * 0.999]
            np.testing.assert_allclose(th.cpu().numpy(), g["adam_thetas"][i], rtol=2e-5, atol=2e-6)
