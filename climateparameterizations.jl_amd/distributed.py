"""Column sharding and the one exchange step of the path.

Columns are independent given the weights, so they shard over ranks with no data-path collective; once per
optimiser iteration every rank contributes [grad(n_params); 6 loss terms; total; 0] — already normalised by the
GLOBAL column count (colnde_set_global_columns) — to one SUM all-reduce (RCCL on GPUs: backend "nccl"; gloo in
the CPU tests).  The reference has no distributed code; this is the MI355X-native addition.
"""
from __future__ import annotations

from typing import Tuple


def shard_columns(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of the global column list owned by `rank` (sim-major order preserved)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d not in 0..%d" % (rank, world - 1))
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def allreduce_loss_grad(buf, group=None):
    """SUM all-reduce of the [n_params + 8] result buffer (torch tensor, CPU or device); returns it."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


def split_result(buf, n_params: int):
    """(total, terms[6], grad) from the result buffer."""
    return float(buf[n_params + 6]), buf[n_params:n_params + 6], buf[:n_params]
