#!/usr/bin/env python3
"""timeline of the last step(s) of a rocprofv3 kernel trace: start offset, duration and gap to the previous kernel's end
usage: r05_timeline.py <kernel_trace.csv> <anchor kernel substring> [steps back = 1]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
if len(idx) < back + 2:
    print("anchor not found often enough"); sys.exit(1)
lo, hi = idx[-back - 1], idx[-back]
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = None
for r in rows[lo:hi + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("wm::(anonymous namespace)::", "").replace("wm::split::", "split::")[:48]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%9.1f us  dur %8.1f  gap %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r.get("Queue_Id", "?"), name))
    prev_end = e if prev_end is None else max(prev_end, e)
print("step period %.1f us" % ((int(rows[hi]["Start_Timestamp"]) - t0) / 1e3))
