#!/usr/bin/env python
"""append_unique against the oracle: random sizes / universes / id widths, repeated targets, one hot id; every case three times
(a race in the hash-table route would show as a run that differs). 300 cases x 3 clean."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, oracle
import wholegraph_amd.torch as wgth
import wholegraph_amd.torch.graph_ops as gops
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_ERROR))
wgth.create_group_communicator(1)
rng = np.random.default_rng(123)
bad = 0
for it in range(300):
    dt = np.int32 if it % 2 else np.int64
    nt = int(rng.choice([0, 1, 7, 1000, 30000]))
    nn = int(rng.choice([0, 1, 64, 5000, 200000, 900000]))
    universe = int(rng.choice([3, 50, 5000, 10**6, 2**31 - 1]))
    t = rng.integers(0, universe, nt).astype(dt)      # targets may repeat
    n = rng.integers(0, universe, nn).astype(dt)
    if nn > 10 and rng.random() < 0.3:
        n[rng.integers(0, nn, nn // 3)] = n[0]        # one hot id
    ou, om = oracle.append_unique(t, n)
    for rep in range(3):                              # same inputs several times: any race would show as a differing run
        u, m = gops.append_unique(torch.from_numpy(t).cuda(), torch.from_numpy(n).cuda(), True)
        if not (np.array_equal(u.cpu().numpy(), ou) and np.array_equal(m.cpu().numpy(), om)):
            bad += 1
            print("MISMATCH it %d rep %d nt %d nn %d universe %d %s" % (it, rep, nt, nn, universe, np.dtype(dt).name), flush=True)
print("append_unique stress: 300 cases x 3 runs, failures %d" % bad)
