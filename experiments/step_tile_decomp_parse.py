#!/usr/bin/env python
"""Groups the step_tile_kernel launches of ONE experiments/step_tile_decomp.py process by variant (launch order = the plan).
  step_tile_decomp_parse.py trace <kernel_trace.csv> <steps> <rounds>      -> kernel duration per variant (timed calls only)
  step_tile_decomp_parse.py pmc <counter_collection.csv> <steps> <rounds>  -> counters per launch per variant"""
import csv, sys, collections
mode, path, steps, rounds = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
WARM = 2
names = ["as_is", "grads_seq", "table_dense", "both_seq"]
per = WARM + steps
def variant_of(i):   # i-th step_tile_kernel launch of the process
    blk = i // per
    return names[blk % 4] if (i % per) >= WARM and blk < 4 * rounds else None
if mode == "trace":
    rows = [r for r in csv.DictReader(open(path)) if "step_tile_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    acc = collections.defaultdict(list)
    for i, r in enumerate(rows):
        v = variant_of(i)
        if v: acc[v].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for v in names:
        d = acc[v]
        if d: print("   %-12s step_tile_kernel launches %3d  avg %8.1f us  min %8.1f  max %8.1f" % (v, len(d), sum(d) / len(d), min(d), max(d)))
else:
    rows = [r for r in csv.DictReader(open(path)) if "step_tile_kernel" in r["Kernel_Name"]]
    disp = sorted({int(r["Dispatch_Id"]) for r in rows})
    idx = {d: i for i, d in enumerate(disp)}
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in rows:
        v = variant_of(idx[int(r["Dispatch_Id"])])
        if v:
            a = acc[v][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for v in names:
        if acc[v]: print("   %-12s %s" % (v, "  ".join("%s=%.6g" % (c, s / k) for c, (k, s) in sorted(acc[v].items()))))
