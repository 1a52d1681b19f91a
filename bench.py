#!/usr/bin/env python3
"""bench.py — column-timesteps/s (forward + adjoint) of the 32-level wind-mixing NDE on N MI355X GPUs.

One "step" = one pass of the hot path over the resident batch: forward RK4 solve + discrete adjoint +
deterministic gradient reduce for every column of this rank (`colnde_loss_grad_dev`), then — for N > 1 —
one RCCL SUM all-reduce of [grad(19,563); 6 loss terms; total] (the only exchange step).
Columns shard across ranks with no other data-path collective: weak scaling, per-GPU columns fixed.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--columns C]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (metric/unit from BASELINE.json) carrying `roofline` (dominant kernel = the adjoint
kernel, HIP-event timed on its own stream through colnde_kernel_time) and `cpu_baseline` (the C port of the
oracle on the box's host cores, bounded sample, rank 0 at N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

# algorithmic figures per column-timestep (stated in DESIGN.md §4/§5): 2·Σ in·out per net, three nets.
# Each kernel is charged only the MLP flops it actually performs (padding, activations and physics are not counted):
L1, L2, L3 = 2 * 3 * 96 * 50, 2 * 3 * 50 * 20, 2 * 3 * 20 * 31      # 28,800 / 6,000 / 3,720 flop per RHS
MLP_FLOP_PER_RHS = L1 + L2 + L3                                     # 38,520
FWD_FLOP_PER_COLSTEP = 4 * MLP_FLOP_PER_RHS                         # forward kernel: 154,080
# adjoint pass per stage = dX (38,520) + dW (38,520) + whatever forward state is recomputed rather than taped
ADJ_FLOP_REGTILE = 4 * (MLP_FLOP_PER_RHS + (L2 + L3) + L2)          # rt_adjoint_kernel with Z1 taped: dX + dW2/dW3 + layer-2 recompute = 216,960
ADJ_FLOP_REGTILE_NOZ = ADJ_FLOP_REGTILE + 4 * L1                    # ... recomputing layer 1 as well: 332,160
DW1_FLOP_PER_COLSTEP = 4 * L1                                       # rt_dw1_kernel: 115,200
ADJ_FLOP_TILE16 = 4 * 3 * MLP_FLOP_PER_RHS                          # tile16 adjoint_kernel: recompute + dX + dW = 462,240 (SURVEY's "3x forward")
PEAK_FP32_MFMA_TFLOPS = 157.3                                      # MI355X_MICROARCH.md: FP32 matrix, dense
PEAK_HBM_GBPS = 8000.0


def algorithmic_bytes_per_colstep(Nz, substeps):
    bx = 4 * 3 * Nz
    return dict(forward=bx / substeps, adjoint=2 * bx / substeps, total=3 * bx / substeps)


def cpu_baseline(problem, scalings, budget_s=20.0):
    """Time oracle/colnde_ref.c (float32 port of the oracle) on the host cores: same workload shape, bounded sample."""
    from oracle import cref
    cfg = problem.cfg
    # the GPU box's CPU share for one GPU is 16 cores (more threads only oversubscribe the cgroup)
    threads = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16))
    steps = cfg.n_steps
    # single thread first: 4 columns
    n1 = 4
    truth1 = cref.forward(cfg, problem.x0[:n1], problem.bcs[:n1], problem.weights_truth, n_threads=1)
    t0 = time.perf_counter()
    cref.loss_grad(cfg, problem.x0[:n1], problem.bcs[:n1], problem.weights, truth1, scalings, n_threads=1)
    t1 = time.perf_counter() - t0
    rate1 = n1 * steps / t1
    # all cores: size the sample for ~budget_s/2 of wall time
    per_col = t1 / n1
    ncol = int(max(threads, min(problem.n_columns, (budget_s / 2) / per_col * threads * 0.5)))
    ncol = max(threads, (ncol // threads) * threads)
    ncol = min(ncol, problem.n_columns)
    truth = cref.forward(cfg, problem.x0[:ncol], problem.bcs[:ncol], problem.weights_truth, n_threads=threads)
    t0 = time.perf_counter()
    cref.loss_grad(cfg, problem.x0[:ncol], problem.bcs[:ncol], problem.weights, truth, scalings, n_threads=threads)
    tn = time.perf_counter() - t0
    return {
        "value": ncol * steps / tn, "unit": "column-timesteps/s", "cores": threads, "kind": "port",
        "sample": "oracle/colnde_ref.c (float32 C port of the oracle, OpenMP over columns): %d columns x %d RK4 steps "
                  "fwd+adjoint of the same workload in %.1f s on %d threads; 1 thread: %d columns in %.1f s = %.0f column-timesteps/s; "
                  "reference Julia path not runnable on this box" % (ncol, steps, tn, threads, n1, t1, rate1),
        "value_1thread": rate1,
    }


def launcher_command(n_gpus, argv, port):
    """The command `python bench.py --gpus N ...` turns itself into: one rank per GPU under torch.distributed.run."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n_gpus, argv):
    """Spawn the N ranks as children, relay their output (rank 0 prints the JSON line) and return their exit status."""
    import socket
    import subprocess
    import torch
    visible = torch.cuda.device_count()
    if visible < n_gpus:
        print("bench.py: --gpus %d but only %d HIP device(s) visible on this node" % (n_gpus, visible), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(launcher_command(n_gpus, argv, port), env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--columns", type=int, default=32768, help="columns per GPU (weak scaling); 32768 = one 32-column wavefront per SIMD")
    ap.add_argument("--frames", type=int, default=289, help="saved frames (2-day suite: 289)")
    ap.add_argument("--substeps", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # invoked plainly with N > 1: become the launcher.  Nothing in this process has touched the GPU (device_count() does not
        # initialise it on this image); the ranks are fresh children, never an exec of a process that holds a HIP context.
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import colnde
    from colnde import synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (colnde has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # The exchange step goes through the C ABI's own RCCL communicator (colnde_comm_*, what a Julia host would call);
    # COLNDE_BENCH_COMM=torch selects torch.distributed's backend "nccl" (= RCCL) instead.  COLNDE_BENCH_FORCE_DIST exercises the
    # path with one rank.
    dist = comm = None
    if world > 1 or os.environ.get("COLNDE_BENCH_FORCE_DIST"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on stdout when a communicator comes up; stdout carries the ONE JSON line, so the
        # banner goes to stderr (file-descriptor level: the print comes from native code)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if os.environ.get("COLNDE_BENCH_COMM", "colnde") == "torch":
                import torch.distributed as dist
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                dist.barrier()
            else:
                from colnde.distributed import bootstrap_comm
                comm = bootstrap_comm(rank, world, local_rank)
                warm = torch.zeros(1, dtype=torch.float32, device=dev)
                comm.allreduce(warm, "sum")
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    ncol = args.columns
    # each rank generates only its shard of the global synthetic suite (seeded per rank: independent columns)
    prob = synthetic.wind_mixing_problem(ncol, n_frames=args.frames, substeps=args.substeps,
                                         seed=synthetic.SEED + rank)
    # weights are replicated: every rank must start from the identical vector
    wprob = synthetic.wind_mixing_problem(1, n_frames=2, seed=synthetic.SEED)
    cfg = prob.cfg
    scal = np.array([1.0, 1.0, 1.0, 5e-3, 5e-3, 5e-3], dtype=np.float64)   # NDE_training.jl:257-258 defaults

    nde = colnde.ColumnNDE(cfg, ncol, device=local_rank)
    nde.set_global_columns(ncol * world)
    x0 = torch.from_numpy(prob.x0).to(dev)
    bcs = torch.from_numpy(prob.bcs).to(dev)
    w = torch.from_numpy(wprob.weights).to(dev)
    w_truth = torch.from_numpy(wprob.weights_truth).to(dev)
    nde.set_problem(x0, bcs)
    truth = nde.forward(w_truth)                       # synthetic "truth": trajectory of a perturbed weight set
    nde.set_problem(x0, bcs, truth)
    out = torch.empty(nde.n_params + 8, dtype=torch.float32, device=dev)

    sync_buf = torch.zeros(1, dtype=torch.float32, device=dev)

    def step():
        nde.loss_grad(w, scal, out=out)
        if comm is not None:
            comm.allreduce_result(nde, out)
        elif dist is not None:
            dist.all_reduce(out, op=dist.ReduceOp.SUM)

    def barrier():
        if comm is not None:
            comm.allreduce(sync_buf, "sum")          # every rank arrives before any leaves
        elif dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if comm is None and dist is None:
            return x
        te = torch.tensor([x], dtype=torch.float32, device=dev)
        if comm is not None:
            comm.allreduce(te, "max")
        else:
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        return float(te.item())

    for _ in range(args.warmup):
        step()
    barrier()
    nde.set_profiling(True)
    nde.reset_kernel_times()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    ms_fwd, n_fwd = nde.kernel_time("forward")
    ms_adj, n_adj = nde.kernel_time("adjoint")
    ms_red, n_red = nde.kernel_time("reduce")
    ms_dw1, n_dw1 = nde.kernel_time("dw1")
    nde.set_profiling(False)
    # the same K steps once more with the per-kernel HIP events off (reported beside the headline, never instead of it)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed_plain = max_over_ranks(time.perf_counter() - t0)
    res = out.cpu().numpy()

    if rank == 0:
        colsteps_per_step = ncol * world * cfg.n_steps
        value = colsteps_per_step * args.steps / elapsed
        units_per_launch = ncol * cfg.n_steps                     # one adjoint launch covers this rank's columns
        adj_s = ms_adj / max(n_adj, 1) * 1e-3
        fwd_s = ms_fwd / max(n_fwd, 1) * 1e-3
        ab = algorithmic_bytes_per_colstep(cfg.Nz, cfg.substeps)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("columns") == ncol and tj.get("frames") == args.frames and tj.get("substeps") == args.substeps:
                    traffic = tj.get("adjoint_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        regtile = nde.engine == 2
        plan = nde.plan()
        ztape = regtile and plan["z1_taped"]
        adj_flop = (ADJ_FLOP_REGTILE if ztape else ADJ_FLOP_REGTILE_NOZ) if regtile else ADJ_FLOP_TILE16
        dw1_s = ms_dw1 / max(n_dw1, 1) * 1e-3
        achieved_tf = adj_flop * units_per_launch / adj_s / 1e12
        step_flop = FWD_FLOP_PER_COLSTEP + adj_flop + (DW1_FLOP_PER_COLSTEP if regtile else 0)
        line = {
            "metric": "column-timesteps/sec (fwd+adjoint), 32-level wind-mixing NDE",
            "value": value, "unit": "column-timesteps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_without_kernel_events": elapsed_plain / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "wind_mixing train_NDE 2DaySuite shape (BASELINE configs[2]: 8 sims x 32 levels x 289 frames, fwd+adjoint): synthetic suite replicated to "
                            "%d columns/GPU x %d levels x %d frames, %d RK4 sub-steps/frame, MPP + zero_weights + train_gradient, "
                            "3 x (96-50-20-31 mish), six-term loss" % (ncol, cfg.Nz, args.frames, cfg.substeps),
                "columns_per_gpu": ncol, "levels": cfg.Nz, "frames": args.frames, "substeps": cfg.substeps,
                "rk4_steps": cfg.n_steps, "n_params": cfg.n_params, "parallelism": "columns sharded x%d" % world,
                "exchange": "none (one rank)" if (comm is None and dist is None) else
                            ("colnde_comm (RCCL behind the C ABI)" if comm is not None else "torch.distributed nccl (RCCL)"),
            },
            "roofline": {
                "kernel": "rt_adjoint_kernel" if regtile else "adjoint_kernel", "bound": "mfma",
                "achieved": achieved_tf, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tf / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                "algorithmic_flop_per_column_timestep": adj_flop,
                "whole_step": {"algorithmic_flop_per_column_timestep": step_flop,
                               "achieved": step_flop * colsteps_per_step / (elapsed / args.steps) / 1e12 / world, "unit": "TFLOP/s per GPU",
                               "frac": step_flop * colsteps_per_step / (elapsed / args.steps) / 1e12 / world / PEAK_FP32_MFMA_TFLOPS},
                "avg_launch_ms": adj_s * 1e3, "launches": n_adj,
                "hbm": {"achieved": ab["adjoint"] * units_per_launch / adj_s / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": ab["adjoint"] * units_per_launch / adj_s / 1e9 / PEAK_HBM_GBPS,
                        "algorithmic_bytes_per_column_timestep": ab["adjoint"]},
                "forward_kernel": {"avg_launch_ms": fwd_s * 1e3,
                                   "achieved": FWD_FLOP_PER_COLSTEP * units_per_launch / fwd_s / 1e12, "unit": "TFLOP/s",
                                   "hbm_GBps": ab["forward"] * units_per_launch / fwd_s / 1e9},
                "reduce_kernel_avg_ms": ms_red / max(n_red, 1),
                "dw1_kernel": {"avg_launch_ms": dw1_s * 1e3,
                               "achieved": (DW1_FLOP_PER_COLSTEP * units_per_launch / dw1_s / 1e12) if dw1_s > 0 else None,
                               "unit": "TFLOP/s"},
                "engine": {1: "tile16", 2: "regtile"}.get(nde.engine, str(nde.engine)),
                "plan": plan,
            },
            "loss_total": float(res[nde.n_params + 6]),
            "grad_l2": float(np.linalg.norm(res[:nde.n_params])),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(prob, scal)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    nde.close()
    if comm is not None:
        barrier()
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
