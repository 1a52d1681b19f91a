import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, colnde
from colnde import synthetic, _lib
if os.environ.get("COLNDE_LIB"):      # A/B aid: another build of the library
    _lib.LIB_PATH = os.path.join(os.getcwd(), "climateparameterizations.jl_amd", os.environ["COLNDE_LIB"])
for Nz in (32, 64):
  for ncol in (8, 64):
    p = synthetic.free_convection_problem(ncol, Nz=Nz)
    with colnde.ColumnNDE(p.cfg, ncol) as nde:
        nde.set_problem(p.x0, p.bcs)
        truth = nde.forward(p.weights_truth)
        nde.set_problem(p.x0, p.bcs, truth)
        sc = [0, 0, 1, 0, 0, 0]
        nde.set_profiling(True)
        nde.loss_grad(p.weights, sc)
        nde.reset_kernel_times()
        t0 = time.perf_counter()
        for _ in range(3): nde.loss_grad(p.weights, sc)
        dt = (time.perf_counter() - t0) / 3
        kt = {k: round(nde.kernel_time(k)[0] / max(nde.kernel_time(k)[1], 1), 2) for k in ("forward", "adjoint", "dw1", "reduce")}
        print("free convection Nz=%d, %d columns x %d steps: %.1f ms per iteration %s plan %s" % (Nz, ncol, p.cfg.n_steps, dt * 1e3, kt, nde.plan()), flush=True)
