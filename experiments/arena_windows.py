#!/usr/bin/env python
"""Placement study, part 6: ONE 16 GiB allocation, the 5 GB output window moved through it in 1 GiB steps: how does the gather
level change along the arena?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
rows, dim, n = 100_000_000, 128, 10_000_000
def timed(fn, reps=12):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
e = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
idx = torch.randint(0, rows, (n,), device="cuda")
rows_per_gib = (1 << 30) // (dim * 4)
arena = torch.empty((16 * rows_per_gib, dim), device="cuda")
print("arena base 0x%x" % arena.data_ptr())
print("  ".join("%d GiB: %.3f" % (k, timed(lambda: e.gather(idx, out=arena[k * rows_per_gib:k * rows_per_gib + n]))) for k in range(0, 12)), flush=True)
