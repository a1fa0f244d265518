"""Reduce a rocprofv3 counter_collection.csv of experiments/placement_pmc: per (kernel class tag, variant) and counter the mean
per launch. When a counter has several rows per dispatch (one per (XCC, channel) instance), the instance values are also
reported position by position: min / max / coefficient of variation over the instances and the per-XCC sums — the
per-channel histograms of a slow and a fast pair side by side."""
import collections
import csv
import re
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
if not rows:
    sys.exit("empty csv")
vals = collections.defaultdict(lambda: collections.defaultdict(list))   # (kernel key, counter) -> dispatch -> [instance values]
for r in rows:
    m = re.search(r"(gather512|stream_copy|stream_fill)<(\d+)(?:, (\d+))?>", r["Kernel_Name"])
    if not m:
        continue
    tag = {"0": "probe", "1": "fast", "2": "slow", "3": "contig"}[m.group(2)]
    key = "%s %s v%s" % (m.group(1), tag, m.group(3) or "-")
    vals[(key, r["Counter_Name"])][int(r["Dispatch_Id"])].append(float(r["Counter_Value"]))
print("%-34s %-44s %8s %16s" % ("kernel (class, variant)", "counter", "launches", "mean per launch"))
for (key, ctr), dd in sorted(vals.items()):
    per = [sum(v) for v in dd.values()]
    line = "%-34s %-44s %8d %16.1f" % (key, ctr, len(per), sum(per) / len(per))
    k = {len(v) for v in dd.values()}
    if k != {1} and len(k) == 1:
        a = np.array(list(dd.values())).mean(axis=0)      # mean over launches, per instance position
        line += "   instances %d: min %.0f max %.0f cv %.4f" % (len(a), a.min(), a.max(), a.std() / max(a.mean(), 1e-9))
        if len(a) % 8 == 0:
            line += "  | by 8 groups (xcc-major order assumed): " + " ".join("%.0f" % x for x in a.reshape(8, -1).sum(axis=1))
            line += "  | by position inside a group: " + " ".join("%.0f" % x for x in a.reshape(8, -1).sum(axis=0))
    print(line)
