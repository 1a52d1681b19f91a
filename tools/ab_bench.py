"""A/B two builds of libcolnde in ONE process (interleaved rounds; guide rule 24).  usage: ab_bench.py libA.so libB.so [columns]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import colnde
from colnde import _lib, synthetic, nde as ndemod
libs = sys.argv[1:3]
ncol = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
p = synthetic.wind_mixing_problem(ncol, n_frames=289)
dev = torch.device("cuda", 0)
x0, bcs, w, wt = (torch.from_numpy(a).to(dev) for a in (p.x0, p.bcs, p.weights, p.weights_truth))
handles = []
for path in libs:
    _lib._lib = None
    _lib.LIB_PATH = os.path.join(ROOT, "climateparameterizations.jl_amd", path)
    h = colnde.ColumnNDE(p.cfg, ncol)
    h.set_problem(x0, bcs)
    truth = h.forward(wt)
    h.set_problem(x0, bcs, truth)
    h.set_profiling(True)
    handles.append(h)
sc = [1, 1, 1, 5e-3, 5e-3, 5e-3]
out = torch.empty(p.cfg.n_params + 8, device=dev)
for h in handles:
    h.loss_grad(w, sc, out=out)
torch.cuda.synchronize()
for h in handles: h.reset_kernel_times()
for rnd in range(3):
    for h in handles:
        h.loss_grad(w, sc, out=out)
torch.cuda.synchronize()
for path, h in zip(libs, handles):
    print(path, {k: round(h.kernel_time(k)[0] / max(h.kernel_time(k)[1], 1), 2) for k in ("forward", "adjoint", "dw1", "reduce")})
