#!/usr/bin/env python
"""One 51 GB table, several 5 GB output buffers alive at once: does the gather level follow the OUTPUT buffer? (placement study)"""
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
def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
first = len(sys.argv) > 2 and sys.argv[2] == "outs-first"   # allocate the output buffers BEFORE the table
if first:
    outs = [torch.empty((n, dim), device="cuda") for _ in range(k)]
e = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
if not first:
    outs = [torch.empty((n, dim), device="cuda") for _ in range(k)]
for r in range(2):
    for i, o in enumerate(outs):
        print("out %d at 0x%x: gather %.4f ms   fill (write only) %.4f ms" % (i, o.data_ptr(), timed(lambda: e.gather(idx, out=o)), timed(lambda: o.fill_(1.0))), flush=True)
# offsets inside one buffer: shift the output by k rows
big = torch.empty((n + 4096, dim), device="cuda")
for shift in (0, 1, 8, 64, 512, 4096):
    o = big[shift:shift + n]
    print("shift %5d rows (0x%x): gather %.4f ms" % (shift, o.data_ptr(), timed(lambda: e.gather(idx, out=o))), flush=True)
