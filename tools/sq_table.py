"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel (its largest dispatch) the counters, summed over XCDs, and the SQ split.
usage: sq_table.py <counter_collection.csv> [> profiles/<tag>_sq_counters.csv]
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts cycles (MI355X_MICROARCH.md)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(lambda: collections.defaultdict(float))
meta = {}
for r in rows:
    key = (r["Kernel_Name"], r["Dispatch_Id"])
    d[key][r["Counter_Name"]] += float(r["Counter_Value"])
    meta[key] = r
best = {}
for (k, disp), c in d.items():
    size = c.get("SQ_WAVE_CYCLES", 0.0) or c.get("FETCH_SIZE", 0.0) or c.get("WRITE_SIZE", 0.0) or sum(c.values())
    if k not in best or size > best[k][0]:
        best[k] = (size, disp, c)
names = sorted({n for _, _, c in best.values() for n in c})
print("Kernel,Grid_Size,Workgroup_Size,VGPR,Scratch,LDS," + ",".join(names) + ",frac_active,frac_wait_any,frac_wait_inst,mfma_busy_per_wave_quadcycle")
for k, (size, disp, c) in sorted(best.items(), key=lambda kv: -kv[1][0]):
    m = meta[(k, disp)]
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    fr = lambda n: ("%.4f" % (c.get(n, 0.0) / wc)) if wc else ""
    short = k.split("(")[0][:70]
    print('"%s",%s,%s,%s,%s,%s,' % (short, m.get("Grid_Size", ""), m.get("Workgroup_Size", ""), m.get("VGPR_Count", m.get("Arch_VGPR_Count", "")),
                                   m.get("Scratch_Size", m.get("Private_Segment_Size", "")), m.get("LDS_Block_Size", "")) +
          ",".join("%.6g" % c.get(n, 0.0) for n in names) + "," + fr("SQ_ACTIVE_INST_ANY") + "," + fr("SQ_WAIT_ANY") + "," + fr("SQ_WAIT_INST_ANY") + "," + fr("SQ_VALU_MFMA_BUSY_CYCLES"))
