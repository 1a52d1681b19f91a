#!/usr/bin/env python3
"""VERDICT r4 task 2b / ADVICE r4: what float32 costs the RKC2 stepper on the 64-level ConvectiveAdjustmentNDE as the stage count grows.

The same problem (free_convection/src/convective_adjustment_nde.jl:33-48, K = 10, 40 columns of which 8 carry an inverted layer, t in [0, 1],
5 save points, `solve.jl:4` reltol = 1e-4 is the accuracy the reference asks for) is solved with S RKC2 steps per save interval, S = 2 .. 128,
each with the automatic stage count s(S) (least s with 0.9 beta(s) >= lambda dt: fixed stability margin), so S x s sweeps
2 x 132, 4 x 94, 8 x 66, ..., 128 x 17 (the last is the step bench.py's configs[3] shard takes: dt = 1/512).  For each point:

  * float32 oracle vs float64 oracle (same discretisation): trajectory, loss, gradient — the round-off the stepper amplifies;
  * float64 at this (S, s) vs float64 at the finest point: the discretisation error, to see which of the two dominates;
  * with --gpu: the HIP path under both matrix arithmetics vs the float64 oracle at the same (S, s), and vs each other.

Writes one JSON (default profiles/r05_rkc2_conditioning.json) and prints the table.  Oracle part: CPU only, this container."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from colnde import synthetic          # noqa: E402
from oracle import nde_oracle as O    # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def problem(n_col=40, inverted=8):
    p = synthetic.free_convection_problem(n_col, Nz=64, n_save=5, convective_adjustment=True)
    x = p.x0.copy()
    x[:inverted, 20:44] = x[:inverted, 20:44][:, ::-1]
    return p, x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_rkc2_conditioning.json"))
    ap.add_argument("--substeps", type=int, nargs="*", default=[2, 4, 8, 16, 32, 64, 128])
    ap.add_argument("--fixed-dt", action="store_true", help="also the finest step with 2x, 4x, 8x its automatic stage count")
    ap.add_argument("--oracle-json", default=None, help="--gpu: reuse the oracle columns of an earlier run (the GPU box has few cores)")
    args = ap.parse_args()
    p, x0 = problem()
    sc = np.array([0, 0, 1.0, 0, 0, 0])
    rows = []
    # the truth every point is scored against: the finest discretisation's float64 trajectory of the perturbed weights
    fine = p.cfg.with_(stepper="rkc2", substeps=max(args.substeps))
    truth = O.solve(fine, x0, p.bcs, p.weights_truth).astype(np.float32)
    ref_tot = ref_g = ref_sol = None
    prev = {}
    if args.oracle_json and os.path.exists(args.oracle_json):
        prev = {(r["substeps"], r["stages"]): r for r in json.load(open(args.oracle_json))["rows"]}
    if args.gpu:
        import colnde
    points = [(S, 0) for S in sorted(args.substeps, reverse=True)]
    # ... and at the FINEST step, over-provisioned stage counts: the same dt with 2x, 4x, 8x the stages separates what the stage count does to
    # float32 (internal stability of the recurrence) from what the step size does (accuracy through live switches)
    points += [(max(args.substeps), k * O.rkc_stages(fine)) for k in (2, 4, 8)] if args.fixed_dt else []
    for S, forced in points:
        cfg = p.cfg.with_(stepper="rkc2", substeps=S, rkc_stages=forced)
        s = O.rkc_stages(cfg)
        t0 = time.time()
        tot64, _, g64, sol64 = O.loss_and_grad(cfg, x0, p.bcs, p.weights, truth, sc)
        if ref_tot is None:
            ref_tot, ref_g, ref_sol = tot64, g64, sol64
        row = {"substeps": S, "stages": s, "stages_forced": bool(forced), "rhs_per_interval": S * s, "lambda_dt": O.stiff_lambda(cfg) * 0.25 / S,
               "s2_eps32": s * s * 2.0 ** -24,
               "disc_sol_abs": float(np.abs(sol64 - ref_sol).max()), "disc_loss_rel": abs(tot64 - ref_tot) / ref_tot, "disc_grad_rel": rel(g64, ref_g)}
        if (S, s) in prev and "f32_sol_abs" in prev[(S, s)]:
            row.update({k: v for k, v in prev[(S, s)].items() if k.startswith("f32_")})
        else:
            tot32, _, g32, sol32 = O.loss_and_grad(cfg, x0, p.bcs, p.weights, truth, sc, dtype=np.float32)
            row.update({"f32_sol_abs": float(np.abs(sol32 - sol64).max()), "f32_loss_rel": abs(float(tot32) - tot64) / tot64, "f32_grad_rel": rel(g32, g64)})
        if args.gpu:
            res = {}
            for ma in ("bf16x3_exact", "f32_mfma"):
                with colnde.ColumnNDE(cfg, x0.shape[0], matrix_arithmetic=ma) as nde:
                    nde.set_problem(x0, p.bcs, truth)
                    sol = nde.forward(p.weights)
                    tot, _, g = nde.loss_grad(p.weights, list(sc))
                    if ma == "bf16x3_exact":
                        row["gpu_error_estimate"] = float(nde.error_estimate(p.weights))      # what colnde_error_estimate says about this step
                res[ma] = (tot, g, sol)
                row["gpu_%s_sol_abs" % ma] = float(np.abs(sol - sol64).max())
                row["gpu_%s_loss_rel" % ma] = abs(tot - tot64) / tot64
                row["gpu_%s_grad_rel" % ma] = rel(g, g64)
            a, b = res["bf16x3_exact"], res["f32_mfma"]
            row["gpu_split_vs_f32mfma_loss_rel"] = abs(a[0] - b[0]) / abs(b[0])
            row["gpu_split_vs_f32mfma_grad_rel"] = rel(a[1], b[1])
            row["gpu_split_vs_f32mfma_sol_abs"] = float(np.abs(a[2] - b[2]).max())
        row["seconds"] = time.time() - t0
        rows.append(row)
        print(json.dumps(row), flush=True)
    rows.sort(key=lambda r: (r["stages_forced"], r["substeps"], r["stages"]))
    out = {"problem": "ConvectiveAdjustmentNDE, 64 levels, 64-256-256-63 relu, K = 10, 40 columns (8 with an inverted layer), t in [0,1], 5 save points; "
                      "RKC2 with S steps per save interval and the automatic stage count; loss = MSE against the float64 trajectory of a perturbed weight set "
                      "at the finest S; gradient = one-switch-pattern RKC2 adjoint (include/colnde.h)",
           "columns": {"disc_*": "float64 oracle at (S, s) vs float64 oracle at the finest S", "f32_*": "float32 oracle vs float64 oracle, same (S, s)",
                       "gpu_<arithmetic>_*": "HIP path vs float64 oracle, same (S, s)", "s2_eps32": "s^2 * 2^-24: the textbook internal-stability amplification of RKC round-off",
                       "stages_forced": "rows at the finest step with 2x, 4x, 8x its automatic stage count (same dt, more stages)",
                       "gpu_error_estimate": "colnde_error_estimate at this step (Richardson, the integrator's own norm): what a caller who asks is told"},
           "rows": rows}
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    keys = ["substeps", "stages", "f32_sol_abs", "f32_loss_rel", "f32_grad_rel", "disc_loss_rel", "disc_grad_rel"] + \
           (["gpu_bf16x3_exact_loss_rel", "gpu_f32_mfma_loss_rel", "gpu_split_vs_f32mfma_loss_rel", "gpu_split_vs_f32mfma_grad_rel", "gpu_error_estimate"] if args.gpu else [])
    print(" ".join("%12s" % k[-12:] for k in keys))
    for r in rows:
        print(" ".join("%12.3g" % r[k] for k in keys))


if __name__ == "__main__":
    main()
