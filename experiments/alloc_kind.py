#!/usr/bin/env python
"""Does the allocation route matter? CHUNKED (hipMalloc) against CONTINUOUS (hipMemCreate + map) 51 GB tables in one process, same
ids, same output buffer: gather and scatter levels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
rows, dim, n = 100_000_000, 128, 10_000_000
idx = torch.randint(0, rows, (n,), device="cuda")
out = torch.empty((n, dim), device="cuda")
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for r in range(3):
    for mt in ("chunked", "continuous"):
        e = wgth.create_embedding(comm, mt, "cuda", torch.float32, [rows, dim])
        t = e.get_embedding_tensor()
        print("round %d %-10s: gather %.4f ms  scatter %.4f ms" % (r, mt, timed(lambda: e.gather(idx, out=out)), timed(lambda: t.scatter(out, idx))), flush=True)
        wgth.destroy_embedding(e)
