"""Differential fuzz of the three gradient paths on the regtile shape: engine AUTO (net-split kernels, rich or plain tape), tile16 and regtile
on random physics variants, activations, column counts, frame counts, sub-step counts and loss scalings; prints the worst disagreement.
Usage (GPU box): python tools/fuzz_engines.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import colnde
from colnde import synthetic
from tests.test_oracle import VARIANTS

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
names = ["mpp_zero_weights", "mpp_bc_faces", "diurnal", "conv_adj_branch", "swish", "raw", "dRi_small", "relu", "tanh", "leakyrelu"]
worst = dict(sol=0.0, loss=0.0, grad=0.0)
for c in range(cases):
    name = names[rng.integers(len(names))]
    ncol = int(rng.choice([1, 3, 8, 16, 17, 40, 64, 100, 257]))
    frames = int(rng.choice([2, 3, 5, 9, 17]))
    sub = int(rng.choice([1, 2, 3, 5]))
    rich = str(int(rng.integers(2)))
    arith = str(rng.choice(["bf16x3_exact", "bf16x3_exact", "f32_mfma"]))        # of the two engines under test (tile16, the reference here, runs f32 MFMA)
    p = synthetic.wind_mixing_problem(ncol, n_frames=frames, weight_divisor=1e2, **VARIANTS[name])      # (weights/1e3 and smaller: the loss sinks into float32 round-off of the trajectories and relative errors mean nothing)
    cfg = p.cfg.with_(substeps=max(sub, p.cfg.substeps if "conv_adj" in name else sub))
    rkc = bool(rng.integers(4) == 0)                                             # a quarter of the cases under the stabilised RKC2 stepper (net-split kernels and tile16; regtile refuses it)
    if rkc:
        cfg = cfg.with_(stepper="rkc2", substeps=int(rng.choice([1, 2])))
    sc = np.concatenate([rng.uniform(0.5, 1.5, 3), rng.uniform(0, 1e-2, 3) * (rng.integers(2))])
    res = {}
    truth = None
    for label, eng in (("tile16", 1), ("auto", 0), ("regtile", 2)):
        os.environ["COLNDE_T16_SPLIT_RICH"] = rich
        try:
            with colnde.ColumnNDE(cfg, ncol, engine=eng, matrix_arithmetic="f32_mfma" if label == "tile16" else arith) as nde:
                nde.set_problem(p.x0, p.bcs)
                if truth is None:
                    truth = nde.forward(p.weights_truth)          # one truth for all three engines
                nde.set_problem(p.x0, p.bcs, truth)
                sol = nde.forward(p.weights)
                tot, terms, g = nde.loss_grad(p.weights, sc)
                res[label] = (sol, tot, g.astype(np.float64), nde.plan())
        except colnde.ColndeError as e:
            res[label] = None
            msg = str(e)[:80]
    if res["auto"] is None or res["tile16"] is None:
        print("case %d %s ncol %d frames %d sub %d: refused (%s)" % (c, name, ncol, frames, sub, msg)); continue
    ref = res["tile16"]
    line = "case %2d %-16s ncol %3d frames %2d sub %d rich %s split %s %s%s loss %.1e:" % (c, name, ncol, frames, cfg.substeps, rich, res["auto"][3]["split_adjoint"], arith, " rkc2" if rkc else "", ref[1])
    for label in ("auto", "regtile"):
        r = res[label]
        if r is None: continue
        ds = np.abs(r[0] - ref[0]).max()
        dl = abs(r[1] - ref[1]) / max(abs(ref[1]), 1e-30)
        dg = np.linalg.norm(r[2] - ref[2]) / max(np.linalg.norm(ref[2]), 1e-30)
        worst["sol"] = max(worst["sol"], ds); worst["loss"] = max(worst["loss"], dl); worst["grad"] = max(worst["grad"], dg)
        line += "  %s vs tile16: sol %.1e loss %.1e grad %.1e" % (label, ds, dl, dg)
    print(line, flush=True)
print("worst disagreement with tile16 over %d cases: sol %.2e, loss %.2e (relative), gradient %.2e (relative L2)" % (cases, worst["sol"], worst["loss"], worst["grad"]))
