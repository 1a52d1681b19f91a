"""Two RANKS on the one GPU a test box has: each process holds a ragged column shard on the HIP engine (`colnde_set_global_columns`), computes its
[grad; terms; total] buffer on the device, and the exchange is ONE SUM all-reduce — over gloo on host copies here, because RCCL refuses two ranks on one
device; the product's exchange (`colnde_comm`, RCCL) has only ever run with one rank.  What this pins on real kernels: sharding, normalisation by the global
count, the divergence guard, `agree_substeps` (MAX over ranks) and a device-resident ADAM step applied identically on both ranks.  Reference: the serial
comprehension over simulations of NDE_training.jl:291,304 that the shards replace, and the mean over simulations of :312-317."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
N_COL = 37          # ragged: 19 + 18


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(kind="wind_mixing"):
    from colnde import synthetic
    if kind == "free_convection":        # BASELINE configs[3]'s model (64 levels, 64-256-256-63 relu) on the fc32 engine
        return synthetic.free_convection_problem(N_COL, Nz=64, n_save=9, t_end=8.0 / 128.0), [0, 0, 1, 0, 0, 0], 1e-4
    return synthetic.wind_mixing_problem(N_COL, n_frames=9, weight_divisor=1.5), [1, 1, 1, 5e-3, 5e-3, 5e-3], 1e-3      # nets large enough that reltol = 1e-3 needs more than the stability bound's 2 sub-steps


def _worker(rank, world, port, out_dir, truth_path, kind):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import colnde
    from colnde.distributed import shard_columns, weights_in_sync, agree_substeps
    p, sc, reltol = _problem(kind)
    truth = np.load(truth_path)
    lo, hi = shard_columns(N_COL, rank, world)
    dev = torch.device("cuda", 0)
    cpu_max = lambda t: (lambda c: (dist.all_reduce(c, op=dist.ReduceOp.MAX), t.copy_(c))[1])(t.detach().cpu())
    with colnde.ColumnNDE(p.cfg.with_(substeps=0, reltol=reltol), hi - lo) as nde:
        nde.set_global_columns(N_COL)
        nde.set_problem(p.x0[lo:hi], p.bcs[lo:hi], truth[lo:hi])
        w = torch.from_numpy(p.weights).to(dev)
        ok, spread = weights_in_sync(w, cpu_max)
        agreed = agree_substeps(lambda: nde.choose_substeps(p.weights, reltol)[0], nde.set_substeps, lambda t: dist.all_reduce(t, op=dist.ReduceOp.MAX))
        out = nde.loss_grad(w, sc)
        host = out.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)                               # the ONE exchange step
        out.copy_(host)
        m, v = torch.zeros_like(w), torch.zeros_like(w)
        nde.adam_step(w, out, m, v, 3e-4, (0.9, 0.999), 1e-8, beta_t=(0.9, 0.999))
        ok2, spread2 = weights_in_sync(w, cpu_max)
        np.save(os.path.join(out_dir, "r%d.npy" % rank), np.concatenate([host.numpy(), w.cpu().numpy(), [agreed, float(ok), spread, float(ok2), spread2]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["wind_mixing", "free_convection"])
def test_two_ranks_share_the_gpu_and_reproduce_the_single_process_step(tmp_path, kind):
    import colnde
    from oracle import nde_oracle as O
    p, sc, reltol = _problem(kind)
    with colnde.ColumnNDE(p.cfg, N_COL) as one:
        one.set_problem(p.x0, p.bcs)
        truth = one.forward(p.weights_truth)
    np.save(tmp_path / "truth.npy", truth)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), str(tmp_path / "truth.npy"), kind), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    n = p.cfg.n_params
    np.testing.assert_array_equal(r0[:2 * n + 8], r1[:2 * n + 8])              # identical reduced buffer and identical updated weights on both ranks
    agreed = int(r0[2 * n + 8])
    assert agreed == int(r1[2 * n + 8]) and agreed >= 1 and r0[2 * n + 9] == 1.0 and r0[2 * n + 10] == 0.0 and r0[2 * n + 11] == 1.0 and r0[2 * n + 12] == 0.0
    with colnde.ColumnNDE(p.cfg.with_(substeps=agreed), N_COL) as one:          # the single-process step at the agreed count
        one.set_problem(p.x0, p.bcs, truth)
        tot, terms, grad = one.loss_grad(p.weights, sc)
    assert np.isclose(r0[n + 6], tot, rtol=2e-5)
    np.testing.assert_allclose(r0[n:n + 6], terms, rtol=1e-4, atol=1e-30)
    assert np.linalg.norm(r0[:n] - grad) < (2e-5 if kind == "wind_mixing" else 1e-3) * np.linalg.norm(grad)          # other tiles, other summation order (fc: relu kinks, tests/test_gpu_fc.py)
    # ... and against the float64 oracle at that count
    g64 = O.loss_and_grad(p.cfg.with_(substeps=agreed), p.x0, p.bcs, p.weights, truth, np.array(sc, float))[2]
    assert np.linalg.norm(r0[:n] - g64) < (2e-3 if kind == "wind_mixing" else 4e-3) * np.linalg.norm(g64)
    # the ADAM step both ranks applied: eta * sign-like step of the first iteration
    w_new = r0[n + 8:2 * n + 8]
    assert 0 < np.abs(w_new - p.weights).max() <= 3e-4 * 1.0001
