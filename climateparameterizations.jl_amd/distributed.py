"""Column sharding and the one exchange step of the path.

Columns are independent given the weights, so they shard over ranks with no data-path collective; once per
optimiser iteration every rank contributes [grad(n_params); 6 loss terms; total; 0] — already normalised by the
GLOBAL column count (colnde_set_global_columns) — to one SUM all-reduce (RCCL on GPUs: backend "nccl"; gloo in
the CPU tests).  The reference has no distributed code; this is the MI355X-native addition.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

from . import _lib


def shard_columns(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of the global column list owned by `rank` (sim-major order preserved)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d not in 0..%d" % (rank, world - 1))
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def visible_gpu_count(sysfs_root: str = "/sys/class/kfd/kfd/topology/nodes", env=None) -> Optional[int]:
    """How many GPUs a child process will see, WITHOUT touching HIP (a launcher that forks ranks must not hold a runtime): the KFD
    topology nodes with compute units (`simd_count` > 0: CPU nodes have 0), cut down by ROCR_VISIBLE_DEVICES and then by
    HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (comma-separated lists; an empty list hides everything).  None when the topology is
    not readable (no amdgpu driver in this container): the caller then lets the ranks find out."""
    env = os.environ if env is None else env
    n = None
    try:
        nodes = sorted(os.listdir(sysfs_root), key=lambda q: int(q) if q.isdigit() else 1 << 30)
        n = 0
        for node in nodes:
            try:
                with open(os.path.join(sysfs_root, node, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        pass
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if var in env:
            listed = [q for q in env[var].split(",") if q.strip() != ""]
            if not listed:
                return 0                   # an empty list hides every device, whatever the topology says
            if n is not None:
                n = min(n, len(listed))
            if var != "ROCR_VISIBLE_DEVICES":
                break                      # HIP_VISIBLE_DEVICES wins over CUDA_VISIBLE_DEVICES; both index into what ROCR left visible
    return n


def allreduce_loss_grad(buf, group=None):
    """SUM all-reduce of the [n_params + 8] result buffer (torch tensor, CPU or device); returns it."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


def split_result(buf, n_params: int):
    """(total, terms[6], grad) from the result buffer."""
    return float(buf[n_params + 6]), buf[n_params:n_params + 6], buf[:n_params]


class Comm:
    """`colnde_comm`: RCCL behind the C ABI (include/colnde.h), for hosts that do not carry torch.distributed — what a Julia
    deployment calls.  One per process (= per GPU)."""

    def __init__(self, rank: int, world: int, unique_id: bytes, device: int = 0):
        if len(unique_id) != 128:
            raise ValueError("the RCCL unique id is 128 bytes")
        self._L = _lib.lib()
        self._c = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(unique_id, 128)
        _lib.check(self._L.colnde_comm_create(int(rank), int(world), buf, int(device), ctypes.byref(self._c)))
        self.rank, self.world, self.device = int(rank), int(world), int(device)

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.lib().colnde_comm_unique_id(buf))
        return buf.raw

    def allreduce(self, tensor, op: str = "sum", stream_ptr: Optional[int] = None):
        """In place on a contiguous float32 device tensor, enqueued on torch's current stream (or `stream_ptr`)."""
        import torch
        if tensor.dtype != torch.float32 or not tensor.is_contiguous() or not tensor.is_cuda:
            raise ValueError("need a contiguous float32 device tensor")
        if stream_ptr is None:
            stream_ptr = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._L.colnde_comm_allreduce_dev(self._c, ctypes.c_void_p(tensor.data_ptr()), tensor.numel(),
                                                     {"sum": 0, "max": 1}[op], ctypes.c_void_p(stream_ptr)))
        return tensor

    def allreduce_result(self, nde, out):
        """The [n_params + 8] result buffer of `nde.loss_grad(..., out=out)`, summed over the ranks on the handle's stream."""
        _lib.check(self._L.colnde_allreduce_result_dev(nde._h, self._c, ctypes.c_void_p(out.data_ptr())))
        return out

    def close(self):
        if getattr(self, "_c", None) is not None and self._c.value:
            self._L.colnde_comm_destroy(self._c)
            self._c = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_BOOTSTRAPS = 0      # per-process count of exchange_unique_id calls: every rank makes the same sequence of calls, so the n-th call uses the same key everywhere


def exchange_unique_id(rank: int, world: int, make_id, key: str = "colnde_uid", store=None) -> bytes:
    """Host-side bootstrap: rank 0 calls `make_id()` and publishes the bytes through the launcher's TCP store (MASTER_ADDR /
    MASTER_PORT, as torch.distributed's env:// rendezvous uses it: every rank a client when torchrun hosts the store, rank 0 the server
    otherwise); every rank returns the same bytes.  No process group, no collective, no GPU.

    The store outlives one bootstrap (torchrun's agent keeps it for the whole job), so every call publishes under a key of its own —
    `<key>/<n>` with n this process's call count, identical on every rank — and rank 0 deletes it once all ranks have read: a second
    communicator in the same job (another problem, a retry) can never pick up the first one's id."""
    global _BOOTSTRAPS
    from datetime import timedelta
    import time
    if store is None:
        from torch.distributed import TCPStore
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("MASTER_PORT", "29500"))
        agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "False") == "True"
        store = TCPStore(addr, port, world, is_master=(rank == 0 and not agent), timeout=timedelta(seconds=300), multi_tenant=True)
    k = "%s/%d" % (key, _BOOTSTRAPS)
    _BOOTSTRAPS += 1
    if rank == 0:
        store.set(k, make_id())
    uid = bytes(store.get(k))                     # blocks until rank 0 has published THIS bootstrap's id
    store.add(k + "/read", 1)
    if rank == 0:
        # keep the store alive (rank 0 may host it) and the key in place until every rank has read it, then retire both keys
        t0 = time.time()
        while int(store.add(k + "/read", 0)) < world:
            if time.time() - t0 > 300:
                raise RuntimeError("exchange_unique_id: %d of %d ranks read the id within 300 s" % (int(store.add(k + "/read", 0)), world))
            time.sleep(0.01)
        for dead in (k, k + "/read"):
            try:
                store.delete_key(dead)
            except Exception:                      # a store without delete support: the per-call key already makes reuse impossible
                pass
    return uid


def bootstrap_comm(rank: int, world: int, device: int, key: str = "colnde_uid") -> Comm:
    """Rank 0 makes the RCCL unique id, `exchange_unique_id` hands it to every rank, every rank joins."""
    return Comm(rank, world, exchange_unique_id(rank, world, Comm.unique_id, key), device)


def weights_in_sync(theta, rank_allreduce_max, tol: float = 0.0):
    """Divergence guard for replicated weights.  Every rank forms three float64 checksums (Σθ, Σ|θ|, Σ (1 + i mod 65521) θᵢ — the last
    one sees permutations), splits each into a float32 head and tail, and contributes [c; −c] (12 floats) to ONE MAX all-reduce: the
    result holds max and −min of every component over the ranks.
    Returns (in_sync, spread), spread = the largest max − min (0.0 when the replicas are bit-identical, which identical ADAM steps on an
    identical all-reduced gradient guarantee).  `rank_allreduce_max(t)` reduces the tensor in place (Comm.allreduce(t, "max"), or
    torch.distributed.all_reduce with ReduceOp.MAX)."""
    import torch
    d = theta.detach().double()
    pos = (torch.arange(d.numel(), device=d.device, dtype=torch.float64) % 65521.0) + 1.0
    c64 = torch.stack([d.sum(), d.abs().sum(), (d.reshape(-1) * pos).sum()])
    hi = c64.float()
    lo = (c64 - hi.double()).float()
    c = torch.cat([hi, lo])
    t = torch.cat([c, -c]).contiguous()
    rank_allreduce_max(t)
    t = t.cpu()
    spread = float((t[:6] + t[6:]).abs().max())
    return spread <= tol, spread


def agree_substeps(choose_local, impose, rank_allreduce_max, device=None) -> int:
    """One sub-step count for every rank of a column-sharded run.  `substeps = 0` / `colnde_choose_substeps` pick the count from the
    columns a handle holds, so ranks would settle on different counts: the SUM-all-reduced gradient would mix discretisations, step times would be
    unbalanced and tape sizes differ per rank — a handle that holds a shard therefore refuses the automatic choice.  The recipe instead:

        agree_substeps(lambda: nde.choose_substeps(w, reltol)[0], nde.set_substeps, lambda t: comm.allreduce(t, "max"))

    every rank chooses from its shard (`choose_local() -> int`), ONE 1-element MAX all-reduce (`rank_allreduce_max(t)` reduces the float32
    tensor in place: Comm.allreduce(t, "max") or torch.distributed.all_reduce with ReduceOp.MAX), every rank imposes the maximum
    (`impose(count)` = `ColumnNDE.set_substeps`).  Before the first `loss_grad` of the handle.  Returns the agreed count."""
    import torch
    local = int(choose_local())
    t = torch.tensor([float(local)], dtype=torch.float32, device=device)
    rank_allreduce_max(t)
    agreed = int(round(float(t.cpu()[0])))
    if agreed < local:
        raise RuntimeError("agree_substeps: the MAX over ranks (%d) is below this rank's own count (%d)" % (agreed, local))
    impose(agreed)
    return agreed
