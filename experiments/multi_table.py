#!/usr/bin/env python
"""Several 51 GB tables alive at once in one process: does each allocation get its own gather level? (placement study)"""
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
sidx = torch.sort(idx).values
out = torch.empty((n, dim), device="cuda")
def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
embs = []
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for i in range(k):
    e = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
    local, _ = e.get_embedding_tensor().get_local_tensor()
    embs.append(e)
    print("table %d at 0x%x: gather %.4f ms, sorted ids %.4f ms, scatter %.4f ms" % (
        i, local.data_ptr(), timed(lambda: e.gather(idx, out=out)), timed(lambda: e.gather(sidx, out=out)),
        timed(lambda: e.get_embedding_tensor().scatter(out, idx))), flush=True)
print("again, all alive:")
for r in range(2):
    for i, e in enumerate(embs):
        print("  table %d: gather %.4f ms" % (i, timed(lambda: e.gather(idx, out=out))), flush=True)
# a torch-allocated table (hipMalloc through the caching allocator) through the same kernel
