"""Derive profiles/traffic.json (per-launch HBM bytes and SQ shares of each kernel of the headline step) from the rocprofv3 PMC passes.
usage: traffic_from_pmc.py <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> <tag> [<SQ counter_collection.csv>] [matrix arithmetic]
FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section), so it
is doubled.  The largest dispatch of each kernel is taken (the bench also launches a tape-less forward to make the truth).
SQ: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles per wave; SQ_VALU_MFMA_BUSY_CYCLES counts SIMD cycles with the matrix pipe busy;
SQ_BUSY_CYCLES is summed per shader engine (32 of them for 1,024 SIMDs), so the matrix pipe's busy share of the kernel is
SQ_VALU_MFMA_BUSY_CYCLES / (32 SQ_BUSY_CYCLES) — calibrated on the one-wave-per-SIMD kernels, where it equals SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_WAVE_CYCLES)."""
import collections, csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fetch_csv, write_csv, tag = sys.argv[1:4]
sq_csv = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] not in ("", "-") else None
arith = sys.argv[5] if len(sys.argv) > 5 else "bf16x3_exact"
KEYS = {"rt16_forward": "rt16_forward_kernel", "rt_forward": "rt_forward_kernel", "rt_adjoint": "rt_adjoint_kernel", "rt_dw1": "rt_dw1_",
        "reduce": "reduce_kernel"}


def per_dispatch(path, counter):
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: max(v.values()) for k, v in d.items()}


f, w = per_dispatch(fetch_csv, "FETCH_SIZE"), per_dispatch(write_csv, "WRITE_SIZE")
out = {"engine": "regtile", "columns": 32768, "frames": 289, "substeps": 2, "matrix_arithmetic": arith, "per_kernel": {}}
for short, sub in KEYS.items():
    fk = [v for k, v in f.items() if sub in k]
    wk = [v for k, v in w.items() if sub in k]
    if not fk and not wk:
        continue
    fr, wr = (max(fk) if fk else 0.0), (max(wk) if wk else 0.0)
    out["per_kernel"][short] = {"fetch_KB_raw": fr, "write_KB": wr, "hbm_bytes": (2.0 * fr + wr) * 1024.0}
out["adjoint_hbm_bytes_per_launch"] = out["per_kernel"]["rt_adjoint"]["hbm_bytes"]
if "rt16_forward" in out["per_kernel"]:
    out["forward_hbm_bytes_per_launch"] = out["per_kernel"]["rt16_forward"]["hbm_bytes"]
if "rt_dw1" in out["per_kernel"]:
    out["dw1_hbm_bytes_per_launch"] = out["per_kernel"]["rt_dw1"]["hbm_bytes"]
if sq_csv:
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(sq_csv)):
        d[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    sq = {}
    for short, sub in KEYS.items():
        c = max((v for (k, _), v in d.items() if sub in k), key=lambda v: v.get("SQ_WAVE_CYCLES", 0.0), default=None)
        if not c or not c.get("SQ_WAVE_CYCLES"):
            continue
        wc = c["SQ_WAVE_CYCLES"]
        sq[short] = {"issuing": c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, "parked_on_waitcnt_or_barrier": c.get("SQ_WAIT_ANY", 0.0) / wc,
                     "issue_stalled": c.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                     "matrix_pipe_busy": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (32.0 * c["SQ_BUSY_CYCLES"]) if c.get("SQ_BUSY_CYCLES") else None,
                     "valu_instructions_per_mfma": c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_MFMA"] if c.get("SQ_INSTS_MFMA") else None}
    out["sq"] = sq
out["correction"] = ("gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md HBM section); cross-check on known byte counts: rt16_forward WRITE_SIZE = stage "
                     "tape 28.99 GB + Z1 tape + sol 3.64 GB; rt_dw1 2*FETCH_SIZE = stage tape 28.99 GB + delta tape 48.32 GB (20 groups since r02b)")
out["source"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* (separate passes, --kernel-trace only) -- python3 bench.py --steps 1 --warmup 1 "
                 "--no-cpu-baseline --no-configs (%s)" % tag)
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
for k, v in out["per_kernel"].items():
    print("%-14s %8.2f GB per launch" % (k, v["hbm_bytes"] / 1e9))
for k, v in out.get("sq", {}).items():
    print("%-14s %s" % (k, {a: round(b, 4) if b is not None else None for a, b in v.items()}))
