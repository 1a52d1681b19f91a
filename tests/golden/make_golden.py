"""Generates tests/golden/*.npz from the float64 NumPy oracle (oracle/nde_oracle.py).

The reference is Julia and cannot run in the build container, and its tests hold no vectors for
this path, so these are *self*-golden vectors: they freeze the pinned oracle's outputs (inputs + expected
outputs only — no reference source).  Regenerate with:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import colnde  # noqa: E402
from colnde import synthetic  # noqa: E402
from oracle import nde_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def cfg_dict(cfg):
    d = {k: getattr(cfg, k) for k in cfg.__dataclass_fields__}
    return {k: (np.array(v) if isinstance(v, tuple) and k != "activations" else v) for k, v in d.items()}


def save(name, p, scal, extra=None):
    truth = O.solve(p.cfg, p.x0, p.bcs, p.weights_truth).astype(np.float32)
    tot, terms, grad, sol = O.loss_and_grad(p.cfg, p.x0, p.bcs, p.weights, truth, scal)
    dx = O.rhs(p.cfg, p.x0, p.bcs, p.weights, 0.01)
    out = dict(x0=p.x0, bcs=p.bcs, weights=p.weights, truth=truth, scalings=np.asarray(scal, np.float64),
               sol=sol.astype(np.float32), terms=terms, total=np.float64(tot), grad=grad.astype(np.float32),
               rhs_t=np.float64(0.01), rhs=dx.astype(np.float32))
    out.update(extra or {})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "total", tot, "|grad|", np.linalg.norm(grad))


def main():
    # BASELINE configs[0]/[2] shape, shortened: MPP + zero_weights + train_gradient, 3 x (96-50-20-31 mish)
    save("wind_mixing_mpp", synthetic.wind_mixing_problem(4, n_frames=9, weight_divisor=1e2),
         [1.0, 0.8, 1.2, 5e-3, 4e-3, 6e-3])
    save("wind_mixing_diurnal_smooth", synthetic.wind_mixing_problem(3, n_frames=5, weight_divisor=1e2, diurnal=True,
                                                                     smooth_NN=True, smooth_Ri=True),
         [1.0, 1.0, 1.0, 5e-3, 5e-3, 5e-3])
    # free convection, Nz = 32 (32-128-128-31 relu) and the convective-adjustment NDE
    save("free_convection_32", synthetic.free_convection_problem(3, Nz=32, n_save=5, substeps=2, t_end=0.02), [0, 0, 1.0, 0, 0, 0])
    save("conv_adj_nde_32", synthetic.free_convection_problem(3, Nz=32, n_save=5, substeps=20, t_end=0.01,
                                                              convective_adjustment=True), [0, 0, 1.0, 0, 0, 0])
    cfg, T, top, w = synthetic.inference_problem(6, 5)
    np.savez_compressed(os.path.join(HERE, "infer_forcing_32.npz"), T=T, top_flux=top, weights=w, Lz=np.float64(1000.0),
                        forcing=O.infer_forcing(cfg, T, top, w, 1000.0).astype(np.float32))
    # the steps either side of the hot path: implicit convective adjustment, ADAM
    r = np.random.default_rng(synthetic.SEED + 7)
    Tc = (np.linspace(5.0, 25.0, 32)[None, :] + 1.5 * r.standard_normal((7, 32))).astype(np.float32)
    hb = (Tc[:, 0] + r.standard_normal(7)).astype(np.float32)
    ht = (Tc[:, -1] + r.standard_normal(7)).astype(np.float32)
    th, g = r.standard_normal(101), 1e-2 * r.standard_normal((3, 101))
    m, v, bt, ths = np.zeros(101), np.zeros(101), (0.9, 0.999), []
    th_run = th.copy()
    for i in range(3):
        th_run, m, v, bt = O.adam_step(th_run, g[i], m, v, 1e-3, (0.9, 0.999), 1e-8, bt)
        ths.append(th_run.copy())
    np.savez_compressed(os.path.join(HERE, "column_ops.npz"), T=Tc, halo_bottom=hb, halo_top=ht, dt=np.float64(1200.0),
                        dz=np.float64(62.5), K=np.float64(10.0), T_adjusted=O.convective_adjustment(Tc, 1200.0, 62.5, 10.0),
                        T_adjusted_halo=O.convective_adjustment(Tc, 1200.0, 62.5, 10.0, hb, ht),
                        adam_theta0=th, adam_grads=g, adam_thetas=np.array(ths), adam_m=m, adam_v=v)


if __name__ == "__main__":
    main()
