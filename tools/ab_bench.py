"""A/B two builds of libcolnde in ONE process (interleaved rounds; guide rule 24).  usage: ab_bench.py libA.so libB.so ... [columns]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import colnde
from colnde import _lib, synthetic, nde as ndemod
libs = [a for a in sys.argv[1:] if a.endswith('.so')]
rest = [a for a in sys.argv[1:] if not a.endswith('.so')]
ncol = int(rest[0]) if rest else 32768
p = synthetic.wind_mixing_problem(ncol, n_frames=289)
dev = torch.device("cuda", 0)
x0, bcs, w, wt = (torch.from_numpy(a).to(dev) for a in (p.x0, p.bcs, p.weights, p.weights_truth))
sc = [1, 1, 1, 5e-3, 5e-3, 5e-3]
out = torch.empty(p.cfg.n_params + 8, device=dev)


def make(path):
    _lib._lib = None
    _lib.LIB_PATH = os.path.join(ROOT, "climateparameterizations.jl_amd", path)
    h = colnde.ColumnNDE(p.cfg, ncol)
    h.set_problem(x0, bcs)
    truth = h.forward(wt)
    h.set_problem(x0, bcs, truth)
    h.set_profiling(True)
    h.loss_grad(w, sc, out=out)
    torch.cuda.synchronize()
    h.reset_kernel_times()
    return h


def report(path, h):
    print(path, {k: round(h.kernel_time(k)[0] / max(h.kernel_time(k)[1], 1), 2) for k in ("forward", "adjoint", "dw1", "reduce")}, flush=True)


if len(libs) <= 2:          # both handles' tapes fit in HBM: interleave the rounds
    handles = [make(path) for path in libs]
    for rnd in range(3):
        for h in handles:
            h.loss_grad(w, sc, out=out)
    torch.cuda.synchronize()
    for path, h in zip(libs, handles):
        report(path, h)
else:                       # one handle at a time
    for path in libs:
        h = make(path)
        for rnd in range(3):
            h.loss_grad(w, sc, out=out)
        torch.cuda.synchronize()
        report(path, h)
        h.close()
        del h
