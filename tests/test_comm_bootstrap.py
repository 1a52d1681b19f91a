"""The host-side half of the colnde_comm bootstrap (no GPU): under `torch.distributed.run` with two ranks, rank 0's id bytes reach
rank 1 through the launcher's TCP store, whichever of the two store arrangements the launcher uses."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, %r)
from colnde.distributed import exchange_unique_id
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
uid = exchange_unique_id(rank, world, lambda: bytes(range(128)))
assert uid == bytes(range(128)), uid
print("rank %%d got the id" %% rank, flush=True)
"""


@pytest.mark.parametrize("agent_store", ["True", "False"])
def test_unique_id_reaches_every_rank(tmp_path, agent_store):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, TORCHELASTIC_USE_AGENT_STORE=agent_store)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "rank 0 got the id" in r.stdout and "rank 1 got the id" in r.stdout


WORKER_TWICE = r"""
import os, sys
sys.path.insert(0, %r)
from colnde.distributed import exchange_unique_id
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
first = exchange_unique_id(rank, world, lambda: bytes([1] * 128))
second = exchange_unique_id(rank, world, lambda: bytes([2] * 128))     # a second communicator in the same job: the store is still the first one's
assert first == bytes([1] * 128) and second == bytes([2] * 128), (first[:4], second[:4])
print("rank %%d got both ids" %% rank, flush=True)
"""


@pytest.mark.parametrize("agent_store", ["True", "False"])
def test_second_bootstrap_in_the_same_job_gets_its_own_id(tmp_path, agent_store):
    """with a fixed key a non-zero rank could `get` the FIRST bootstrap's id before rank 0 overwrote it and then hang in
    ncclCommInitRank on mismatched ids.  Every call now uses `<key>/<n>` and rank 0 retires the key once all ranks have read."""
    script = tmp_path / "worker2.py"
    script.write_text(WORKER_TWICE % ROOT)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, TORCHELASTIC_USE_AGENT_STORE=agent_store)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "rank 0 got both ids" in r.stdout and "rank 1 got both ids" in r.stdout


def test_exchange_unique_id_twice_on_one_store_in_process():
    """The same property on ONE store object shared by two ranks (threads): both ranks must receive the SECOND id from the second call."""
    import threading
    from datetime import timedelta
    import socket
    from torch.distributed import TCPStore
    from colnde import distributed as D
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    master = TCPStore("127.0.0.1", port, 2, is_master=True, timeout=timedelta(seconds=60), multi_tenant=True, wait_for_workers=False)
    client = TCPStore("127.0.0.1", port, 2, is_master=False, timeout=timedelta(seconds=60), multi_tenant=True)
    got = {}
    base = D._BOOTSTRAPS

    def rank1():
        # rank 1 runs its own call counter in a real job (its own process); here both ranks share the module, so rank 1 pins the keys by hand
        got[1] = []
        for n in range(2):
            k = "colnde_uid/%d" % (base + n)
            got[1].append(bytes(client.get(k)))
            client.add(k + "/read", 1)
    th = threading.Thread(target=rank1)
    th.start()
    got[0] = [D.exchange_unique_id(0, 2, lambda: bytes([7] * 128), store=master),
              D.exchange_unique_id(0, 2, lambda: bytes([9] * 128), store=master)]
    th.join(60)
    assert not th.is_alive()
    assert got[0] == [bytes([7] * 128), bytes([9] * 128)] and got[1] == got[0]
    # rank 0 retired the keys: nothing of the first bootstrap is left for a later reader to pick up
    assert not master.check(["colnde_uid/%d" % base])
