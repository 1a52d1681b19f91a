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
