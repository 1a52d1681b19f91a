"""Diagnostic: the whole of BASELINE config 4 (65,536 columns x 64 levels) on ONE GPU, through the column-blocked taped path."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, colnde
from colnde import synthetic
ncol = 65536
p = synthetic.free_convection_problem(ncol, Nz=64)
dev = torch.device('cuda', 0)
nde = colnde.ColumnNDE(p.cfg, ncol)
x0, bcs, w, wt = (torch.from_numpy(a).to(dev) for a in (p.x0, p.bcs, p.weights, p.weights_truth))
nde.set_problem(x0, bcs); truth = nde.forward(wt); nde.set_problem(x0, bcs, truth)
out = torch.empty(p.cfg.n_params + 8, device=dev)
nde.loss_grad(w, [0, 0, 1, 0, 0, 0], out=out); torch.cuda.synchronize()
t0 = time.perf_counter(); nde.loss_grad(w, [0, 0, 1, 0, 0, 0], out=out); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("free convection 65536 columns x 64 levels on one GPU: %.1f ms -> %.2f M column-timesteps/s; plan %s" % (dt * 1e3, ncol * p.cfg.n_steps / dt / 1e6, nde.plan()))
