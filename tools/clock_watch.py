"""Shader clock and board power of this GPU (amdgpu sysfs, bench.GpuSensors) while one configuration's training step runs in a loop:
is the chip power-limited under a kernel mix, and at which clock does it settle?
usage: clock_watch.py [seconds per case]   -> one line per case: wind mixing (headline shape) and free convection (configs[3] shard, 32-level
sibling), each under both matrix arithmetics, plus the tape-less forward solve alone."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import colnde
from colnde import synthetic
from bench import GpuSensors

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
dev = torch.device("cuda", 0)


def watch(label, fn):
    fn()
    torch.cuda.synchronize()
    s = GpuSensors(dev).start()
    t0, n = time.time(), 0
    while time.time() - t0 < secs:
        fn()
        torch.cuda.synchronize()
        n += 1
    dt = (time.time() - t0) / n
    c = s.stop() or {}
    print("%-46s %8.2f ms/pass | clock %6.0f MHz (min %4.0f max %4.0f) | board %6.0f W | %d samples" % (
        label, dt * 1e3, c.get("shader_clock_MHz_mean", float("nan")), c.get("shader_clock_MHz_min", float("nan")),
        c.get("shader_clock_MHz_max", float("nan")), c.get("board_power_W_mean") or float("nan"), c.get("samples", 0)), flush=True)


for what in ("wm", "fc64", "fc32"):
    if what == "wm":
        ncol = 32768
        p = synthetic.wind_mixing_problem(ncol)
        sc = [1, 1, 1, 1, 1, 1]
    else:
        ncol = 16384
        p = synthetic.free_convection_problem(ncol, Nz=64 if what == "fc64" else 32)
        sc = [0, 0, 1, 0, 0, 0]
    h = colnde.ColumnNDE(p.cfg, ncol)
    x0, bcs, w, wt = (torch.from_numpy(a).to(dev) for a in (p.x0, p.bcs, p.weights, p.weights_truth))
    h.set_problem(x0, bcs)
    truth = h.forward(wt)
    sol = torch.empty_like(truth)
    h.set_problem(x0, bcs, truth)
    out = torch.empty(p.cfg.n_params + 8, device=dev)
    for ma in ("bf16x3_exact", "f32_mfma"):
        h.set_matrix_arithmetic(ma)
        watch("%s %s: training step" % (what, ma), lambda: h.loss_grad(w, sc, out=out))
        watch("%s %s: forward solve alone (no tapes)" % (what, ma), lambda: h.forward(w, out=sol))
    h.close()
    del h, truth, sol, out
    torch.cuda.empty_cache()
