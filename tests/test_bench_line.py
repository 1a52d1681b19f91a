"""bench.py prints ONE compact JSON line on stdout (the 25.6 KB line of round 4 did not parse on the driver's side, so the
driver-held record carried no `roofline` and no `cpu_baseline`).  The compact line is a pure function of the full record: it is built here
from the canned full record of the last builder-run bench (profiles/r04c_bench.json) and from a worst case with every string at its longest."""
import copy
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                 "data", "config", "roofline", "cpu_baseline")


def _canned():
    return json.load(open(os.path.join(ROOT, "profiles", "r04c_bench.json")))


def test_compact_line_is_small_and_carries_the_contract():
    full = _canned()
    assert len(json.dumps(full)) > 20000                       # the record that did not parse
    text = json.dumps(bench.compact_line(full))
    assert len(text) < 8192
    line = json.loads(text.splitlines()[-1])
    for k in CONTRACT_KEYS:
        assert k in line, k
    assert line["metric"].startswith("column-timesteps/sec") and line["unit"] == "column-timesteps/s" and line["dtype"] == "f32"
    assert "model" not in line["config"] and line["config"]["workload"] and line["config"]["columns_per_gpu"] == 32768
    for k in ("levels", "frames", "substeps", "matrix_arithmetic"):
        assert k in line["config"], k
    rf = line["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches", "hbm", "clock"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and 0 < rf["frac"] < 1 and abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-3 * rf["frac"]
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["value_1thread"] > 0 and cb["sample"]
    assert line["opt_out"]["f32_mfma"]["value"] > 0 and line["self_check"]["ok"] is True
    # the two side configs VERDICT r4 wants driver-visible, as scalars
    assert line["side_configs_ms"]["config4_shard_ms"] > 0 and line["side_configs_ms"]["config3_8sim_ms"] > 0
    assert line["value"] == float("%.6g" % full["value"]) and line["ms_per_step"] == float("%.6g" % full["ms_per_step"])


def test_compact_line_stays_small_in_the_worst_case():
    full = copy.deepcopy(_canned())
    full["cpu_baseline"]["sample"] = "x" * 600
    full["config"]["workload"] = "y" * 600
    full["self_check"] = {"ok": False, "error": "z" * 300, "max_abs_error": float("inf"), "tolerance": 5e-4}
    full["multi_gpu"] = {"per_rank_ms_per_step_before_the_barrier": [99.123456789] * 8, "allreduce_alone_ms": 0.123456789,
                         "allreduce_floats": 19571, "weights_checksum_spread_over_ranks": 0.0}
    full["configs"] = {k: {"error": "e" * 300} for k in full["configs"]}
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < 8192
    assert json.loads(text)["self_check"]["max_abs_error"] is None           # non-finite floats never reach the line (strict JSON)
    assert "NaN" not in text and "Infinity" not in text


def test_compact_line_without_side_configs_or_cpu_baseline():
    full = _canned()
    for k in ("configs", "opt_out"):
        full.pop(k)
    full["cpu_baseline"] = None
    line = bench.compact_line(full)
    assert line["cpu_baseline"] is None and line["opt_out"] is None and "side_configs_ms" not in line
    assert len(json.dumps(line)) < 8192


def test_stdout_is_one_line_in_the_source():
    """The headline path prints exactly one thing on stdout: json.dumps(compact_line(line)); everything else goes to stderr or a file."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main_src = src[src.index("def main():"):]
    prints = [ln.strip() for ln in main_src.splitlines() if ln.strip().startswith("print(") and "file=sys.stderr" not in ln]
    assert prints == ["print(json.dumps(compact_line(line)), flush=True)"], prints
