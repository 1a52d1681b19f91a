"""Throughput of the BASELINE configs other than the headline one (parity-test cases; this is a diagnostic, not bench.py).
usage: bench_configs.py [2|3|4|4ca|4s|4n32|5|ca|mpp|adam] ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import colnde
from colnde import synthetic, _lib
if os.environ.get("COLNDE_LIB"):      # A/B aid: another build of the library
    _lib.LIB_PATH = os.path.join(ROOT, "climateparameterizations.jl_amd", os.environ["COLNDE_LIB"])

dev = torch.device("cuda", 0)
MA = os.environ.get("COLNDE_MA", "bf16x3_exact")      # matrix arithmetic of every handle below (f32_mfma: the opt-out)
_ColumnNDE = colnde.ColumnNDE
colnde.ColumnNDE = lambda *a, **k: _ColumnNDE(*a, **dict(dict(matrix_arithmetic=MA), **k))
which = sys.argv[1:] or ["2", "4s", "5"]


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for c in which:
    if c == "2":      # forward only, 4,096 columns x 32 levels, 288 frames x 2 sub-steps
        p = synthetic.wind_mixing_problem(4096)
        nde = colnde.ColumnNDE(p.cfg, 4096)
        x0, bcs, w = (torch.from_numpy(a).to(dev) for a in (p.x0, p.bcs, p.weights))
        nde.set_problem(x0, bcs)
        sol = nde.forward(w)
        dt = timed(lambda: nde.forward(w, out=sol))
        cs = 4096 * p.cfg.n_steps
        print("config 2: forward 4096 columns: %.2f ms -> %.1f M column-timesteps/s, %.1f TFLOP/s (engine %d)" % (dt * 1e3, cs / dt / 1e6, cs * 154080 / dt / 1e12, nde.engine), flush=True)
        nde.close()
    if c in ("4", "4ca", "4s", "4n32"):   # free convection, 64 levels (4n32: the 32-level 32-128-128-31 network), 64-256-256-63 relu, 129 save points; 16,384 columns = one GPU's shard of 65,536
        ncol = 4096 if c == "4s" else 16384
        Nzc = 32 if c == "4n32" else 64
        p = synthetic.free_convection_problem(ncol, Nz=Nzc, convective_adjustment=(c == "4ca"))
        rhs_per_step = 4
        if c == "4ca":     # ConvectiveAdjustmentNDE is stiff (K = 10): the stabilised RKC2 stepper with its automatic stage count (tile16 engine)
            p.cfg = p.cfg.with_(stepper="rkc2")
            rhs_per_step = colnde.rkc_stages(p.cfg)
        nde = colnde.ColumnNDE(p.cfg, ncol)
        x0, bcs, w, wt = (torch.from_numpy(a).to(dev) for a in (p.x0, p.bcs, p.weights, p.weights_truth))
        nde.set_problem(x0, bcs)
        truth = nde.forward(wt)
        nde.set_problem(x0, bcs, truth)
        out = torch.empty(p.cfg.n_params + 8, device=dev)
        sc = [0, 0, 1, 0, 0, 0]
        nde.set_profiling(True)
        dt = timed(lambda: nde.loss_grad(w, sc, out=out), n=2)
        cs = ncol * p.cfg.n_steps
        mlp = 2 * (Nzc * 4 * Nzc + 16 * Nzc * Nzc + 4 * Nzc * (Nzc - 1))
        kt = {k: round(nde.kernel_time(k)[0] / max(nde.kernel_time(k)[1], 1), 2) for k in ("forward", "adjoint", "dw1", "reduce")}
        print("config %s: fwd+adjoint %d columns x %d levels x %d steps of %d RHS evaluations: %.1f ms -> %.2f M column-timesteps/s, %.1f TFLOP/s at 3x forward flops (engine %d) %s"
              % (c, ncol, Nzc, p.cfg.n_steps, rhs_per_step, dt * 1e3, cs / dt / 1e6, cs * rhs_per_step * 3 * mlp / dt / 1e12, nde.engine, kt), flush=True)
        nde.close()
    if c == "5":      # inference forcing, 256 x 256 columns x 32 levels, 32-128-128-31
        cfg, T, tf, w = synthetic.inference_problem(256, 256)
        nde = colnde.ColumnNDE(cfg, 65536)
        Td, tfd, wd = (torch.from_numpy(a).to(dev) for a in (T, tf, w))
        dt = timed(lambda: nde.infer_forcing(wd, Td, tfd, 1024.0), n=10)
        mlp = 2 * (32 * 128 + 128 * 128 + 128 * 31)
        print("config 5: inference 65536 columns: %.3f ms -> %.1f M columns/s, %.1f TFLOP/s, %.1f GB/s algorithmic (260 B/column)"
              % (dt * 1e3, 65536 / dt / 1e6, 65536 * mlp / dt / 1e12, 65536 * 260 / dt / 1e9), flush=True)
        nde.close()
    if c == "ca":     # implicit convective-adjustment step (double_gyre_nn.jl:27-62) on the config-5 grid and on a large batch
        cfg = synthetic.free_convection_problem(1, Nz=32, n_save=2).cfg
        nde = colnde.ColumnNDE(cfg, 1)
        for ncol in (65536, 4 * 1024 * 1024):
            g = torch.Generator(device="cpu").manual_seed(1)
            T = (torch.linspace(5.0, 25.0, 32)[None, :] + 1.5 * torch.randn(ncol, 32, generator=g)).to(dev)
            out = torch.empty_like(T)
            nde.set_profiling(True); nde.reset_kernel_times()
            dt = timed(lambda: nde.convective_adjustment(T, 1200.0, 62.5, 10.0, out=out), n=20)
            ms, nl = nde.kernel_time("convadj")
            kt = ms / nl * 1e-3
            print("convective adjustment: %d columns x 32 levels: kernel %.4f ms -> %.0f M columns/s, %.0f GB/s of 8000 (256 B/column algorithmic) = %.1f %% of HBM peak; wall %.4f ms"
                  % (ncol, kt * 1e3, ncol / kt / 1e6, ncol * 256 / kt / 1e9, ncol * 256 / kt / 1e9 / 80.0, dt * 1e3), flush=True)
        nde.close()
    if c == "mpp":    # implicit MPP diffusion step (NDE_oceananigans.jl:61-101): u, v, T of 32 levels, 768 B/column algorithmic
        cfg = synthetic.free_convection_problem(1, Nz=32, n_save=2).cfg
        nde = colnde.ColumnNDE(cfg, 1)
        prm = (1e-4, 1e-1, 1.0, 0.25, 1.0, 1.67e-4, 9.81)
        for ncol in (65536, 4 * 1024 * 1024):
            g = torch.Generator(device="cpu").manual_seed(2)
            k = torch.arange(32)[None, :]
            u = (0.05 * torch.tanh((k - 24) / 4.0) + 2e-3 * torch.randn(ncol, 32, generator=g)).to(dev)
            v = (0.02 * torch.tanh((k - 20) / 5.0) + 2e-3 * torch.randn(ncol, 32, generator=g)).to(dev)
            T = (19.6 + 0.4 * k / 32 + 0.02 * torch.randn(ncol, 32, generator=g)).to(dev)
            outs = tuple(torch.empty_like(T) for _ in range(3))
            for ca in (False, True):
                nde.set_profiling(True); nde.reset_kernel_times()
                dt = timed(lambda: nde.implicit_diffusion(u, v, T, 60.0, 8.0, prm, ca, out=outs), n=20)
                ms, nl = nde.kernel_time("impldiff")
                kt = ms / nl * 1e-3
                print("implicit MPP diffusion (ca=%d): %d columns x 32 levels: kernel %.4f ms -> %.0f M columns/s, %.0f GB/s of 8000 (768 B/column algorithmic) = %.1f %% of HBM peak; wall %.4f ms"
                      % (ca, ncol, kt * 1e3, ncol / kt / 1e6, ncol * 768 / kt / 1e9, ncol * 768 / kt / 1e9 / 80.0, dt * 1e3), flush=True)
        nde.close()
    if c == "adam":   # fused ADAM update on the 19,563-parameter vector and on a large one
        p = synthetic.wind_mixing_problem(2, n_frames=3)
        nde = colnde.ColumnNDE(p.cfg, 2)
        for n in (p.cfg.n_params, 64 * 1024 * 1024):
            w, g, m, v = (torch.randn(n, device=dev) for _ in range(4))
            v.abs_()
            nde.set_profiling(True); nde.reset_kernel_times()
            dt = timed(lambda: nde.adam_step(w, g, m, v, 1e-3), n=20)
            ms, nl = nde.kernel_time("adam")
            kt = ms / nl * 1e-3
            print("ADAM step: %d parameters: kernel %.4f ms -> %.0f GB/s (28 B/parameter); wall %.4f ms" % (n, kt * 1e3, n * 28 / kt / 1e9, dt * 1e3), flush=True)
        nde.close()
    if c == "3":      # latency point: 8 simulations x 32 levels x 289 frames, fwd+adjoint (BASELINE configs[2] as written), both engines
        for eng in (2, 1, 0):      # regtile, tile16, AUTO (tile16 gradient + three-wave split forward)
            p = synthetic.wind_mixing_problem(8)
            nde = colnde.ColumnNDE(p.cfg, 8, engine=eng)
            x0, bcs, w, wt = (torch.from_numpy(a).to(dev) for a in (p.x0, p.bcs, p.weights, p.weights_truth))
            nde.set_problem(x0, bcs)
            truth = nde.forward(wt)
            nde.set_problem(x0, bcs, truth)
            out = torch.empty(p.cfg.n_params + 8, device=dev)
            nde.set_profiling(True)
            dt = timed(lambda: nde.loss_grad(w, [1, 1, 1, 5e-3, 5e-3, 5e-3], out=out), n=3)
            kt = {k: round(nde.kernel_time(k)[0] / max(nde.kernel_time(k)[1], 1), 2) for k in ("forward", "adjoint", "dw1", "reduce")}
            print("config 3: 8 columns x 576 RK4 steps fwd+adjoint, engine %d: %.1f ms per iteration -> %.1f k column-timesteps/s %s"
                  % (nde.engine, dt * 1e3, 8 * p.cfg.n_steps / dt / 1e3, kt), flush=True)
            nde.close()
