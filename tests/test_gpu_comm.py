"""RCCL behind the C ABI (colnde_comm_*), exercised with the one rank a single-GPU box allows: bootstrap, all-reduce (sum and
max are the identity on one rank), the result-buffer exchange on the handle's stream, and the sharded training loop through it.
The sharding arithmetic itself (ragged shards, global normalisation, SUM) is covered on two ranks by tests/test_distributed_gloo.py."""
import numpy as np
import pytest

import colnde
from colnde import synthetic
from colnde.distributed import Comm
from colnde.flux_compat import ADAM
from colnde.wind_mixing import WindMixingNDE, train_NDE_device

pytestmark = pytest.mark.gpu


def test_comm_single_rank_allreduce_and_result_exchange():
    import torch
    uid = Comm.unique_id()
    assert len(uid) == 128 and uid != Comm.unique_id()
    comm = Comm(0, 1, uid, device=0)
    assert comm._L.colnde_comm_rank(comm._c) == 0 and comm._L.colnde_comm_size(comm._c) == 1
    x = torch.arange(1000, dtype=torch.float32, device="cuda") - 300.0
    ref = x.clone()
    comm.allreduce(x, "sum")
    comm.allreduce(x, "max")
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    p = synthetic.wind_mixing_problem(40, n_frames=5, weight_divisor=1e2)
    with colnde.ColumnNDE(p.cfg, 40) as nde:
        nde.set_problem(p.x0, p.bcs)
        nde.set_problem(p.x0, p.bcs, nde.forward(p.weights_truth))
        w = torch.from_numpy(p.weights).cuda()
        out = nde.loss_grad(w, [1, 1, 1, 5e-3, 5e-3, 5e-3])
        torch.cuda.synchronize()
        before = out.clone()
        comm.allreduce_result(nde, out)
        torch.cuda.synchronize()
        assert torch.equal(out, before)
    with pytest.raises(colnde.ColndeError, match="rank"):
        Comm(2, 1, uid)
    comm.close()


def test_device_training_loop_through_colnde_comm():
    p = synthetic.wind_mixing_problem(16, n_frames=5, weight_divisor=1e2)
    with colnde.ColumnNDE(p.cfg, 16) as nde:
        nde.set_problem(p.x0, p.bcs)
        truth = nde.forward(p.weights_truth)
    comm = Comm(0, 1, Comm.unique_id(), device=0)
    a = WindMixingNDE(p.cfg, p.x0, p.bcs, truth)
    ra = train_NDE_device(a, p.weights, [ADAM(3e-4)], epochs=1, maxiters=4, comm=comm)
    rb = train_NDE_device(a, p.weights, [ADAM(3e-4)], epochs=1, maxiters=4)
    np.testing.assert_array_equal(ra.weights, rb.weights)
    assert ra.history[-1]["total"] < ra.history[0]["total"]
    a.close()
    comm.close()
