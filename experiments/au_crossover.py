#!/usr/bin/env python
"""append_unique: hash-table route against the sort route across sizes. One process per setting (the route limit is read once):
  WM_AU_TABLE_MAX=0 python au_crossover.py   # sort for everything
  WM_AU_TABLE_MAX=1000000000 python au_crossover.py   # table for everything"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wholegraph_amd.torch as wgth
import wholegraph_amd.torch.graph_ops as gops
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
g = torch.Generator(device="cuda").manual_seed(1)
for dt in (torch.int32, torch.int64):
    for nt, nn, universe in [(1024, 30000, 111_000_000), (31000, 920000, 111_000_000), (100000, 2_000_000, 111_000_000),
                             (200000, 4_000_000, 111_000_000), (400000, 8_000_000, 111_000_000), (800000, 16_000_000, 111_000_000),
                             (1_000_000, 22_400_000, 20_000_000)]:
        t = torch.randperm(universe, device="cuda", generator=g)[:nt].to(dt)
        n = torch.randint(0, universe, (nn,), device="cuda", generator=g).to(dt)
        for _ in range(3):
            gops.append_unique(t, n, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            u, m = gops.append_unique(t, n, True)
        torch.cuda.synchronize()
        print("%s nt %8d nn %9d -> %9d unique: %.3f ms  (WM_AU_TABLE_MAX=%s)" % (
            str(dt).split(".")[1], nt, nn, u.shape[0], (time.perf_counter() - t0) / 10 * 1e3, os.environ.get("WM_AU_TABLE_MAX", "default")), flush=True)
