"""Derive profiles/traffic.json (per-launch HBM bytes of each kernel) from the two rocprofv3 PMC passes.
usage: traffic_from_pmc.py <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> <tag>
FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section), so it
is doubled.  The largest dispatch of each kernel is taken (the bench also launches a tape-less forward to make the truth)."""
import collections, csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fetch_csv, write_csv, tag = sys.argv[1:4]
KEYS = {"rt16_forward": "rt16_forward_kernel", "rt_forward": "rt_forward_kernel", "rt_adjoint": "rt_adjoint_kernel", "rt_dw1": "rt_dw1_kernel",
        "reduce": "reduce_kernel"}


def per_dispatch(path, counter):
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: max(v.values()) for k, v in d.items()}


f, w = per_dispatch(fetch_csv, "FETCH_SIZE"), per_dispatch(write_csv, "WRITE_SIZE")
out = {"engine": "regtile", "columns": 32768, "frames": 289, "substeps": 2, "per_kernel": {}}
for short, sub in KEYS.items():
    fk = [v for k, v in f.items() if sub in k]
    wk = [v for k, v in w.items() if sub in k]
    if not fk and not wk:
        continue
    fr, wr = (max(fk) if fk else 0.0), (max(wk) if wk else 0.0)
    out["per_kernel"][short] = {"fetch_KB_raw": fr, "write_KB": wr, "hbm_bytes": (2.0 * fr + wr) * 1024.0}
out["adjoint_hbm_bytes_per_launch"] = out["per_kernel"]["rt_adjoint"]["hbm_bytes"]
out["correction"] = ("gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md HBM section); cross-check on known byte counts: rt16_forward WRITE_SIZE = stage "
                     "tape 28.99 GB + Z1 tape + sol 3.64 GB; rt_dw1 2*FETCH_SIZE = stage tape 28.99 GB + delta tape 48.32 GB (20 groups since r02b)")
out["source"] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline (%s)" % tag
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
for k, v in out["per_kernel"].items():
    print("%-14s %8.2f GB per launch" % (k, v["hbm_bytes"] / 1e9))
