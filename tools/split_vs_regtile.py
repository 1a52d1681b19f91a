"""Engine AUTO (net-split kernels + tile16's tapes and dW GEMM) against regtile at the full 576-step horizon for a mid-size column count:
iteration time of both and the disagreement of their gradients.  Usage (GPU box): python tools/split_vs_regtile.py [columns]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, colnde
from colnde import synthetic
ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
p = synthetic.wind_mixing_problem(ncol, n_frames=289, weight_divisor=1e2)
sc = [1, 1, 1, 5e-3, 5e-3, 5e-3]
res = {}
truth = None
for label, eng in (("auto", 0), ("regtile", 2)):
    with colnde.ColumnNDE(p.cfg, ncol, engine=eng) as nde:
        nde.set_problem(p.x0, p.bcs)
        if truth is None:
            truth = nde.forward(p.weights_truth)
        nde.set_problem(p.x0, p.bcs, truth)
        nde.loss_grad(p.weights, sc)
        t0 = time.perf_counter(); tot, terms, g = nde.loss_grad(p.weights, sc); dt = time.perf_counter() - t0
        res[label] = (tot, g.astype(np.float64))
        print(label, "%d columns x %d steps: %.1f ms per iteration = %.1f M column-timesteps/s" % (ncol, p.cfg.n_steps, dt * 1e3, ncol * p.cfg.n_steps / dt / 1e6), nde.plan(), flush=True)
print("loss rel diff %.2e, gradient rel diff %.2e" % (abs(res["auto"][0] - res["regtile"][0]) / res["regtile"][0], np.linalg.norm(res["auto"][1] - res["regtile"][1]) / np.linalg.norm(res["regtile"][1])))
