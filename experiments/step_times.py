#!/usr/bin/env python
"""Per-step host times of a DISTRIBUTED gather through the full exchange route on one GPU (loopback): looks for outliers
(allocator growth, lazy initialisation) inside what bench.py would time. WM_FORCE_RCCL=1 WM_EXCHANGE_SELF=1 python experiments/step_times.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
rows, dim, n = 100_000_000, 128, 10_000_000
emb = wgth.create_embedding(comm, "distributed", "cuda", torch.float32, [rows, dim])
idx = torch.randint(0, rows, (n,), device="cuda")
out = torch.empty((n, dim), device="cuda")
ts = []
for i in range(40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    emb.gather(idx, out=out)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
    if i in (0, 4, 5, 39):
        print("step %d: reserved %.1f GB allocated %.1f GB" % (i, torch.cuda.memory_reserved() / 1e9, torch.cuda.memory_allocated() / 1e9))
print("synchronised steps:", " ".join("%.1f" % t for t in ts))
# the same without a synchronise between the steps (what bench.py times): host time of each call
for rep in range(3):
    torch.cuda.synchronize()
    ts, t_all = [], time.perf_counter()
    for i in range(30):
        t0 = time.perf_counter()
        emb.gather(idx, out=out)
        ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    print("back-to-back rep %d: total %.1f ms; per call:" % (rep, (time.perf_counter() - t_all) * 1e3), " ".join("%.1f" % t for t in ts))
    print("   reserved %.1f GB allocated %.1f GB" % (torch.cuda.memory_reserved() / 1e9, torch.cuda.memory_allocated() / 1e9))
