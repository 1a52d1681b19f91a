import sys, time, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import colnde
from colnde import synthetic
dev = torch.device("cuda", 0)
for name, kw in (("400-400 swish", dict(layer_sizes=(96, 400, 400, 31), activations=("swish", "swish", "identity"))),
                 ("400 mish", dict(layer_sizes=(96, 400, 31), activations=("mish", "identity")))):
    p = synthetic.wind_mixing_problem(4096, n_frames=17, weight_divisor=1e2, **kw)
    for ma in ("bf16x3_exact", "f32_mfma"):
        nde = colnde.ColumnNDE(p.cfg, 4096, matrix_arithmetic=ma)
        x0, bcs, w, wt = (torch.from_numpy(a).to(dev) for a in (p.x0, p.bcs, p.weights, p.weights_truth))
        nde.set_problem(x0, bcs)
        truth = nde.forward(wt)
        torch.cuda.synchronize(); t0 = time.time(); nde.forward(w); torch.cuda.synchronize(); tf = time.time() - t0
        nde.set_problem(x0, bcs, truth)
        res = torch.empty(p.cfg.n_params + 8, device=dev)
        nde.loss_grad(w, [1, 1, 1, 5e-3, 5e-3, 5e-3], out=res)
        nde.set_profiling(True); nde.reset_kernel_times()
        torch.cuda.synchronize(); t0 = time.time(); nde.loss_grad(w, [1, 1, 1, 5e-3, 5e-3, 5e-3], out=res); torch.cuda.synchronize(); tg = time.time() - t0
        km = {k: nde.kernel_time(k)[0] / max(nde.kernel_time(k)[1], 1) for k in ("forward", "adjoint", "dw1", "reduce")}
        print(name, ma, "forward-only %.1f ms, loss_grad %.1f ms" % (tf * 1e3, tg * 1e3), km, nde.describe(), flush=True)
        nde.close()
