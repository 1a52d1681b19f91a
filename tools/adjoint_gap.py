#!/usr/bin/env python3
"""Measures, on the CPU oracle, the gap between the product's discrete RK4 adjoint and the reference's kind of gradient —
a continuous interpolating adjoint on an adaptive solve at the reference's tolerances (oracle/continuous_adjoint.py) — for
BASELINE config 3 (8 simulations x 32 levels x 289 frames, reltol 1e-3: NDE_training.jl:304) and a small config 4
(free convection, 64 levels, reltol 1e-4: free_convection/src/solve.jl).  Writes profiles/r02_adjoint_gap.json; the numbers
are quoted in DESIGN §2 and a reduced case is asserted in tests/test_oracle.py.  Never shipped, never timed."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import colnde  # noqa: E402
from colnde import synthetic  # noqa: E402
from oracle import nde_oracle as O  # noqa: E402
from oracle import continuous_adjoint as CA  # noqa: E402

rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))


def case(name, p, sc, rtol, tight_substeps):
    cfg = p.cfg
    truth = O.solve(cfg.with_(substeps=tight_substeps), p.x0, p.bcs, p.weights_truth)
    t0 = time.time()
    tot, terms, g, sol = O.loss_and_grad(cfg, p.x0, p.bcs, p.weights, truth, sc)
    # "exact" gradient of the continuous problem: the discrete adjoint on a much finer RK4 grid (4th-order convergent)
    tot_x, _, g_x, sol_x = O.loss_and_grad(cfg.with_(substeps=tight_substeps), p.x0, p.bcs, p.weights, truth, sc)
    t1 = time.time()
    tot_c, _, g_c, sol_c, st = CA.loss_and_grad_continuous(cfg, p.x0, p.bcs, p.weights, truth, sc, rtol=rtol, atol=1e-6)
    t2 = time.time()
    out = dict(case=name, columns=p.n_columns, n_save=cfg.n_save, substeps=cfg.substeps, rtol=rtol, atol=1e-6,
               loss_discrete=float(tot), loss_continuous=float(tot_c), loss_exact=float(tot_x),
               grad_gap_discrete_vs_continuous=rel(g, g_c),
               grad_gap_discrete_vs_exact=rel(g, g_x), grad_gap_continuous_vs_exact=rel(g_c, g_x),
               cosine_discrete_continuous=float(g @ g_c / np.linalg.norm(g) / np.linalg.norm(g_c)),
               sol_gap_discrete_vs_continuous=float(np.abs(sol - sol_c).max()),
               sol_gap_discrete_vs_exact=float(np.abs(sol - sol_x).max()), sol_gap_continuous_vs_exact=float(np.abs(sol_c - sol_x).max()),
               loss_rel_gap=float(abs(tot - tot_c) / tot_c), seconds_discrete=t1 - t0, seconds_continuous=t2 - t1, **st)
    print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    res = []
    n3 = int(os.environ.get("GAP_COLUMNS", "8"))
    p = synthetic.wind_mixing_problem(n3, n_frames=289, weight_divisor=1e2)
    res.append(case("config3: wind mixing 2DaySuite, 96-50-20-31 x3 mish, weights/1e2", p, O.default_loss_scalings(p.cfg), 1e-3, 16))
    p = synthetic.wind_mixing_problem(n3, n_frames=289, weight_divisor=1e5)
    res.append(case("config3 with the bench's weights/1e5", p, O.default_loss_scalings(p.cfg), 1e-3, 16))
    p = synthetic.free_convection_problem(4, Nz=64, n_save=33, substeps=4, t_end=0.25)
    res.append(case("config4-small: free convection 64 levels, 64-256-256-63 relu", p, O.default_loss_scalings(p.cfg), 1e-4, 32))
    with open(os.path.join(ROOT, "profiles", "r02_adjoint_gap.json"), "w") as f:
        json.dump(res, f, indent=1)
