"""bench.py invoked plainly with --gpus N > 1 becomes the launcher of N fresh rank processes (never an exec of a process
that has touched the GPU); with too few devices it fails with a message and a status, not a traceback."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_launcher_command_is_one_rank_per_gpu_on_loopback():
    cmd = _bench().launcher_command(4, ["--gpus", "4", "--steps", "3", "--warmup", "1"], 29517)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]


def test_plain_multi_gpu_invocation_without_devices_fails_cleanly():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""                     # no device visible, here and on a GPU box alike
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2
    assert "only 0 HIP device(s) visible" in r.stderr and "Traceback" not in r.stderr
    assert r.stdout.strip() == ""
