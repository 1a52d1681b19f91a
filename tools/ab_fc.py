"""A/B of libcolnde builds on the fc32 engine (one handle at a time: config 4's tapes take 165 GB).
usage: ab_fc.py libA.so [libB.so ...] [--nz 64,32] [--columns 16384]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import colnde
from colnde import _lib, synthetic

libs = [a for a in sys.argv[1:] if a.endswith('.so')]
args = sys.argv[1:]
nzs = [int(x) for x in (args[args.index('--nz') + 1] if '--nz' in args else '64,32').split(',')]
ncol = int(args[args.index('--columns') + 1]) if '--columns' in args else 16384
dev = torch.device("cuda", 0)
for rnd in range(2):
    for path in libs:
        _lib._lib = None
        _lib.LIB_PATH = os.path.join(ROOT, "climateparameterizations.jl_amd", path)
        for Nz in nzs:
            p = synthetic.free_convection_problem(ncol, Nz=Nz)
            h = colnde.ColumnNDE(p.cfg, ncol)
            x0, bcs, w, wt = (torch.from_numpy(a).to(dev) for a in (p.x0, p.bcs, p.weights, p.weights_truth))
            h.set_problem(x0, bcs)
            truth = h.forward(wt)
            h.set_profiling(True)
            h.reset_kernel_times()
            for _ in range(3):
                h.forward(w, out=truth)
            fwd_only = h.kernel_time("forward")[0] / 3
            h.forward(wt, out=truth)
            h.set_problem(x0, bcs, truth)
            out = torch.empty(p.cfg.n_params + 8, device=dev)
            h.loss_grad(w, [0, 0, 1, 0, 0, 0], out=out)
            torch.cuda.synchronize()
            h.reset_kernel_times()
            for _ in range(3):
                h.loss_grad(w, [0, 0, 1, 0, 0, 0], out=out)
            torch.cuda.synchronize()
            kt = {k: round(h.kernel_time(k)[0] / max(h.kernel_time(k)[1], 1), 2) for k in ("forward", "adjoint", "dw1")}
            print("%-28s Nz %d engine %d: forward-only %.2f ms | taped %s | sum %.1f ms" % (path, Nz, h.engine, fwd_only, kt, sum(kt.values())), flush=True)
            h.close()
            del h, truth, out
            torch.cuda.empty_cache()
