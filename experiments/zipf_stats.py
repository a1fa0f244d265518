#!/usr/bin/env python
"""What a Zipf(1.05) batch of 10 M ids per rank looks like to the exchange (scripts/first_contact_report.py):
per sender, the copies and the DISTINCT ids that go to each owner, for the hashed and the clustered variant of bench.py
(make_indices) on W x 125 M rows. CPU only; prints the table the report embeds."""
import json
import numpy as np
n, rows_per_rank = 10_000_000, 125_000_000
out = {}
for W in (2, 4, 8):
    N = rows_per_rank * W
    for dist in ("zipf", "zipf_clustered"):
        cop, dis, nus, tops = [], [], [], []
        for r in range(min(W, 4)):
            k = np.random.default_rng(42 + r).zipf(1.05, n).astype(np.uint64)
            ids = ((k * np.uint64(2654435761)) % np.uint64(N) if dist == "zipf" else k % np.uint64(N)).astype(np.int64)
            u, c = np.unique(ids, return_counts=True)
            cop.append(np.bincount(np.minimum(ids // rows_per_rank, W - 1), minlength=W))
            dis.append(np.bincount(np.minimum(u // rows_per_rank, W - 1), minlength=W))
            nus.append(len(u)); tops.append(int(c.max()))
        cop, dis = np.mean(cop, axis=0), np.mean(dis, axis=0)
        out["%s_w%d" % (dist, W)] = {"distinct_per_rank": int(np.mean(nus)), "copies_of_hottest_id": int(np.mean(tops)),
                                     "max_copies_to_one_owner": int(cop.max()), "mean_copies_to_one_owner": int(cop.mean()),
                                     "max_distinct_to_one_owner": int(dis.max()), "mean_distinct_to_one_owner": int(dis.mean())}
print(json.dumps(out, indent=1))
