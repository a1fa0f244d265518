"""Reduce one rocprofv3 --pmc pass over experiments/tables_in_one_process.py: the scatter (and gather) dispatches of
rows_batch_kernel come in blocks of 1 + 3 * REPS per (round, table); per block: mean kernel duration (kernel trace of the same
pass) and, per counter, the mean per launch (+ per-instance min / max / cv and sums by 8 groups when a counter has instances).
usage: tables_pmc_reduce.py <counter_collection.csv> <kernel_trace.csv> <block length>"""
import collections, csv, sys
import numpy as np
cc, kt, block = sys.argv[1], sys.argv[2], int(sys.argv[3])
dur = {}
names = {}
for r in csv.DictReader(open(kt)):
    dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    names[int(r["Dispatch_Id"])] = r["Kernel_Name"]
vals = collections.defaultdict(lambda: collections.defaultdict(list))   # counter -> dispatch -> [instances]
for r in csv.DictReader(open(cc)):
    vals[r["Counter_Name"]][int(r["Dispatch_Id"])].append(float(r["Counter_Value"]))
for op, tag in (("scatter", "rows_batch_kernel<long, false"), ("gather", "rows_batch_kernel<long, true")):
    ids = sorted(d for d, n in names.items() if tag in n)
    blocks = [ids[i:i + block] for i in range(0, len(ids), block)]
    for bi, b in enumerate(blocks):
        if len(b) < block:
            continue
        b = b[1:]   # the first launch of a block is the warm-up
        line = "%-7s table %d: %.4f ms" % (op, bi, sum(dur[d] for d in b) / len(b))
        for ctr, dd in sorted(vals.items()):
            inst = np.array([dd[d] for d in b if d in dd])
            if inst.size == 0:
                continue
            a = inst.mean(axis=0)
            line += " | %s %.0f" % (ctr, a.sum())
            if len(a) > 1:
                line += " (inst %d: min %.0f max %.0f cv %.4f" % (len(a), a.min(), a.max(), a.std() / max(a.mean(), 1e-9))
                if len(a) % 8 == 0:
                    line += "; by 8: " + " ".join("%.0f" % x for x in a.reshape(8, -1).sum(axis=1))
                    if len(a) // 8 <= 16:
                        line += "; by position: " + " ".join("%.0f" % x for x in a.reshape(8, -1).sum(axis=0))
                line += ")"
        print(line)
