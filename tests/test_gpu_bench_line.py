"""bench.py end to end on the GPU (VERDICT r4 task 1): stdout is ONE line, it parses as strict JSON, it is far below 8 KB and carries `roofline` and
`cpu_baseline` — with the side configs on (the block that made round 4's line 25 KB), at a reduced size so that the test stays short."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_compact_parsable_line(tmp_path):
    env = dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--columns", "4096", "--frames", "33"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:3]
    assert len(lines[0]) < 8192
    line = json.loads(lines[0], parse_constant=lambda c: pytest.fail("non-strict JSON constant %s" % c))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["value"] > 0 and line["dtype"] == "f32"
    assert line["config"]["columns_per_gpu"] == 4096 and line["config"]["frames"] == 33
    rf, cb = line["roofline"], line["cpu_baseline"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] < 1 and rf["avg_launch_ms"] > 0 and rf["launches"] == 2 and rf["unit"] == "TFLOP/s"
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "oracle/colnde_ref.c" in cb["sample"]
    assert line["self_check"]["ok"] is True
    sc = line["side_configs_ms"]
    assert sc["errors"] is None and sc["config4_shard_ms"] > 0 and sc["config3_8sim_ms"] > 0 and sc["wide_wind_mixing_4096x32steps_ms"] > 0
    # the full record went to a file, not to stdout / stderr
    full = json.load(open(os.path.join(ROOT, "gpurun_out", "bench_full.json")))
    assert "configs" in full and len(json.dumps(full)) > len(lines[0])
    assert len(r.stderr) < 16384
