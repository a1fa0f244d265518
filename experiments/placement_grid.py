#!/usr/bin/env python
"""Placement study, part 5: T tables x O output buffers alive at once in one process — the gather level of every pair"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
rows, dim, n = 100_000_000, 128, 10_000_000
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
O = int(sys.argv[2]) if len(sys.argv) > 2 else 4
idx = torch.randint(0, rows, (n,), device="cuda")
def timed(fn, reps=12):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
embs = [wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim]) for _ in range(T)]
outs = [torch.empty((n, dim), device="cuda") for _ in range(O)]
print("table addresses: " + " ".join("0x%x" % e.get_embedding_tensor().get_local_tensor()[0].data_ptr() for e in embs))
print("output addresses: " + " ".join("0x%x" % o.data_ptr() for o in outs))
print("gather ms, rows = tables, columns = output buffers")
for ti, e in enumerate(embs):
    print("  table %d: " % ti + "  ".join("%.3f" % timed(lambda: e.gather(idx, out=o)) for o in outs), flush=True)
print("scatter ms (source = output buffer 0): " + "  ".join("%.3f" % timed(lambda: e.get_embedding_tensor().scatter(outs[0], idx)) for e in embs), flush=True)
