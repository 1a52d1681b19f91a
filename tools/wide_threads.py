"""Forward geometry of the wide networks (rows in global memory) at larger column counts: COLNDE_FWD_THREADS=256 | 1024, forward-only ms per 8-step solve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import colnde
from colnde import synthetic
dev = torch.device("cuda", 0)
for ncol in (4096, 8192, 16384, 32768):
    p = synthetic.wind_mixing_problem(ncol, n_frames=5, weight_divisor=1e2, layer_sizes=(96, 400, 400, 31), activations=("swish", "swish", "identity"))
    x0, bcs, w = (torch.from_numpy(a).to(dev) for a in (p.x0, p.bcs, p.weights))
    for th in ("256", "1024"):
        os.environ["COLNDE_FWD_THREADS"] = th
        nde = colnde.ColumnNDE(p.cfg, ncol)
        nde.set_problem(x0, bcs)
        sol = nde.forward(w)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(3): nde.forward(w, out=sol)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 3
        print("%6d columns, %4s threads: %.2f ms per 8-step forward = %.1f M column-timesteps/s" % (ncol, th, dt * 1e3, ncol * 8 / dt / 1e6), flush=True)
        nde.close()
