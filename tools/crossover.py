"""Where the latency-point path (net-split kernels, 3 waves per 16-column tile) hands over to the throughput path (regtile, 32 columns
per wave): iteration time of both on the wind-mixing shape for a range of column counts.  Usage (GPU box): python tools/crossover.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import colnde
from colnde import synthetic

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 33
sc = [1, 1, 1, 5e-3, 5e-3, 5e-3]
for ncol in (8, 64, 256, 1024, 2048, 4096, 8192, 16384):
    p = synthetic.wind_mixing_problem(ncol, n_frames=frames)
    row = []
    for label, engine, env in (("split", 1, "1"), ("regtile", 2, None)):
        if env is None:
            os.environ.pop("COLNDE_T16_FWD_SPLIT", None)
        else:
            os.environ["COLNDE_T16_FWD_SPLIT"] = env
        with colnde.ColumnNDE(p.cfg, ncol, engine=engine) as nde:
            nde.set_problem(p.x0, p.bcs)
            truth = nde.forward(p.weights_truth)
            nde.set_problem(p.x0, p.bcs, truth)
            nde.loss_grad(p.weights, sc)
            t0 = time.perf_counter()
            for _ in range(3):
                nde.loss_grad(p.weights, sc)
            row.append((label, (time.perf_counter() - t0) / 3 * 1e3, nde.plan()))
    steps = p.cfg.n_steps
    print("%6d columns x %d steps: " % (ncol, steps) + "  ".join("%s %.2f ms (%.1f M col-steps/s)" % (l, t, ncol * steps / t / 1e3) for l, t, _ in row)
          + "  split plan: fwd %s adj %s" % (row[0][2]["split_forward"], row[0][2]["split_adjoint"]), flush=True)
