#!/usr/bin/env python3
"""rocprofv3 --output-format json of experiments/tables_in_one_process.py: per-instance (16 TCC channels x 8 XCCs) values of the
collected counters for the SCATTER launches, averaged per table (launch order: per table, gathers then scatters)."""
import json, sys, collections
import numpy as np
d = json.load(open(sys.argv[1]))["rocprofiler-sdk-tool"][0]
names = {c["id"]["handle"]: c["name"] for c in d["counters"]}
ksym = {k["kernel_id"]: k.get("formatted_kernel_name") or k.get("demangled_kernel_name") or k.get("kernel_name") for k in d["kernel_symbols"]}
recs = d["callback_records"].get("counter_collection") or d["buffer_records"].get("counter_collection")
recs.sort(key=lambda r: r["dispatch_data"]["dispatch_info"]["dispatch_id"])
launches = []   # (is_gather, {counter: vector})
for r in recs:
    kn = ksym.get(r["dispatch_data"]["dispatch_info"]["kernel_id"], "")
    if "rows_batch_kernel" not in kn:
        continue
    vec = collections.defaultdict(list)
    for x in r["records"]:
        vec[names[x["counter_id"]["handle"]]].append(x["value"])
    launches.append(("true" in kn.split("rows_batch_kernel<")[1].split(",")[1], {k: np.array(v) for k, v in vec.items()}))
# split into tables: a table's block = its gather launches followed by its scatter launches
tables, cur, seen_scatter = [], [], False
for g, v in launches:
    if g and seen_scatter:
        tables.append(cur); cur, seen_scatter = [], False
    if not g:
        seen_scatter = True
    cur.append((g, v))
tables.append(cur)
for t, block in enumerate(tables):
    sc = [v for g, v in block if not g]
    print("table %d: %d scatter launches" % (t, len(sc)))
    for cname in sorted(sc[0]):
        m = np.mean([v[cname] for v in sc], axis=0)
        inst = m.reshape(-1, 8) if m.size == 128 else m.reshape(1, -1)   # guess: instance-major, XCC-minor
        print("   %-40s total %14.0f  per-slot min %10.0f max %10.0f  max/mean %.3f  cv %.3f" % (cname, m.sum(), m.min(), m.max(), m.max() / max(m.mean(), 1e-9), m.std() / max(m.mean(), 1e-9)))
        if cname == "TCC_EA0_WRREQ":
            print("      sorted slots (k):", " ".join("%d" % (x / 1e3) for x in sorted(m)[:6]), "...", " ".join("%d" % (x / 1e3) for x in sorted(m)[-6:]))
