"""Optimiser iterations per second of the device-resident training loop (train_NDE_device: loss_grad + fused ADAM per iteration, nothing
across PCIe) on the reference's own training shape — 8 simulations x 32 levels x 289 frames, 2 RK4 sub-steps per frame.
Usage (GPU box): python tools/train_rate.py [simulations] [iterations]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import colnde
from colnde import synthetic, wind_mixing
from colnde.flux_compat import ADAM

nsim = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
p = synthetic.wind_mixing_problem(nsim, n_frames=289)
with colnde.ColumnNDE(p.cfg, nsim) as e0:
    e0.set_problem(p.x0, p.bcs)
    truth = e0.forward(p.weights_truth)
prob = wind_mixing.WindMixingNDE(p.cfg, p.x0, p.bcs, truth)
wind_mixing.train_NDE_device(prob, p.weights, [ADAM(1e-3)], epochs=1, maxiters=3)          # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
res = wind_mixing.train_NDE_device(prob, p.weights, [ADAM(1e-3)], epochs=1, maxiters=iters)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("train_NDE_device, %d simulations x %d RK4 steps: %d ADAM iterations in %.3f s = %.1f ms per iteration (%.1f iterations/s); loss %.4e -> %.4e"
      % (nsim, p.cfg.n_steps, iters, dt, dt / iters * 1e3, iters / dt, res.history[0]["total"], min(h["total"] for h in res.history)))
prob.close()
