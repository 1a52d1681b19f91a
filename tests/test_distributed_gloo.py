"""N > 1 path on CPU: two gloo ranks shard the columns, each normalises by the GLOBAL column count, one SUM
all-reduce of [grad; terms; total] reproduces the single-process result.  The per-rank compute stand-in is the C
port of the oracle (no GPU here); sharding, normalisation and the exchange are the product code under test
(colnde.distributed), exactly as bench.py uses them around the HIP engine."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import colnde
from colnde import synthetic
from colnde.distributed import shard_columns, allreduce_loss_grad, split_result, weights_in_sync, agree_substeps
from oracle import cref, nde_oracle as O

N_COL = 7   # deliberately not divisible by the world size (ragged shards)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    p = synthetic.wind_mixing_problem(N_COL, n_frames=5, weight_divisor=1e2)
    truth = cref.forward(p.cfg, p.x0, p.bcs, p.weights_truth)
    return p, truth


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p, truth = _problem()
    lo, hi = shard_columns(N_COL, rank, world)
    sc = O.default_loss_scalings(p.cfg)
    tot, terms, grad, _ = cref.loss_grad(p.cfg, p.x0[lo:hi], p.bcs[lo:hi], p.weights, truth[lo:hi], sc, n_col_total=N_COL)
    buf = torch.zeros(p.cfg.n_params + 8, dtype=torch.float32)
    buf[:p.cfg.n_params] = torch.from_numpy(grad)
    buf[p.cfg.n_params:p.cfg.n_params + 6] = torch.from_numpy(terms)
    buf[p.cfg.n_params + 6] = tot
    allreduce_loss_grad(buf)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), buf.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_columns_partition():
    for n, w in [(7, 2), (16, 4), (5, 8), (65536, 4)]:
        blocks = [shard_columns(n, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in blocks]
        assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_allreduce_matches_single_process(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    p, truth = _problem()
    sc = O.default_loss_scalings(p.cfg)
    tot, terms, grad, _ = cref.loss_grad(p.cfg, p.x0, p.bcs, p.weights, truth, sc)
    r0 = np.load(tmp_path / "rank0.npy")
    r1 = np.load(tmp_path / "rank1.npy")
    np.testing.assert_array_equal(r0, r1)                       # every rank ends with the identical buffer
    t, te, g = split_result(r0, p.cfg.n_params)
    assert np.isclose(t, tot, rtol=1e-5)
    np.testing.assert_allclose(te, terms, rtol=1e-4)
    assert np.linalg.norm(g - grad) < 1e-4 * np.linalg.norm(grad)


def _guard_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    red = lambda t: dist.all_reduce(t, op=dist.ReduceOp.MAX)
    theta = torch.linspace(-1.0, 1.0, 19563, dtype=torch.float32)
    same = weights_in_sync(theta, red)
    drift = theta.clone()
    if rank == 1:
        drift[777] = torch.nextafter(drift[777], torch.tensor(2.0))        # ONE ulp of ONE weight on one rank
    one_ulp = weights_in_sync(drift, red)
    swapped = theta.clone()
    if rank == 1:
        swapped[[10, 11]] = swapped[[11, 10]]                              # the same multiset: only the position-weighted checksum sees it
    np.save(os.path.join(out_dir, "guard%d.npy" % rank), np.array([same[0], same[1], one_ulp[0], one_ulp[1], weights_in_sync(swapped, red)[0]], np.float64))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_weight_divergence_guard(tmp_path):
    """ranks check that their replicated weight vectors are identical with ONE 12-float MAX all-reduce; a one-ulp
    drift of one weight on one rank is caught on EVERY rank (bench.py runs the check once, train_NDE_device every K iterations)."""
    world = 2
    mp.spawn(_guard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        same_ok, same_spread, ulp_ok, ulp_spread, swapped_ok = np.load(tmp_path / ("guard%d.npy" % r))
        assert same_ok == 1.0 and same_spread == 0.0
        assert ulp_ok == 0.0 and ulp_spread > 0.0
        assert swapped_ok == 0.0


def _substeps_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    imposed = []
    got = agree_substeps(lambda: (4, 16)[rank], imposed.append, lambda t: dist.all_reduce(t, op=dist.ReduceOp.MAX))
    np.save(os.path.join(out_dir, "sub%d.npy" % rank), np.array([got] + imposed))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_agree_on_one_substep_count(tmp_path):
    """shards that would choose 4 and 16 sub-steps from their own columns both end up imposing 16 (one MAX all-reduce of one number)."""
    world = 2
    mp.spawn(_substeps_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert np.load(tmp_path / ("sub%d.npy" % r)).tolist() == [16, 16]
