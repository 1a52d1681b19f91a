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
from colnde.distributed import shard_columns, allreduce_loss_grad, split_result
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
