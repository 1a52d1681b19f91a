"""Time the forward solve alone (HIP-event kernel time) for a column count; prints column-timesteps/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import colnde
from colnde import synthetic
ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 289
engine = int(sys.argv[3]) if len(sys.argv) > 3 else 0
p = synthetic.wind_mixing_problem(ncol, n_frames=frames)
dev = torch.device("cuda", 0)
nde = colnde.ColumnNDE(p.cfg, ncol, engine=engine)
x0, bcs, w = (torch.from_numpy(a).to(dev) for a in (p.x0, p.bcs, p.weights))
nde.set_problem(x0, bcs)
sol = nde.forward(w)
torch.cuda.synchronize()
nde.set_profiling(True); nde.reset_kernel_times()
for _ in range(3):
    nde.forward(w, out=sol)
torch.cuda.synchronize()
ms, n = nde.kernel_time("forward")
cs = ncol * p.cfg.n_steps
print("engine", nde._L.colnde_engine(nde._h), "cols", ncol, "forward %.2f ms/launch -> %.1f M column-timesteps/s (fwd only), %.1f TFLOP/s" % (ms / n, cs / (ms / n) / 1e3, cs * 154080 / (ms / n) / 1e9))
