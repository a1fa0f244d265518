#!/usr/bin/env python
"""Placement study, part 3: ONE big allocation, the output window shifted by multiples of 2 MiB (and the id array likewise):
does the gather level follow the ALIGNMENT of the buffers (virtual / physical set mapping) rather than the allocation?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
rows, dim, n = 100_000_000, 128, 10_000_000
def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
e = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
idx0 = torch.randint(0, rows, (n,), device="cuda")
rows_per_2m = (2 << 20) // (dim * 4)
extra = 1024 * rows_per_2m                       # 2 GiB of slack
big = torch.empty((n + extra, dim), device="cuda")
print("out window shifted inside one allocation (base 0x%x):" % big.data_ptr())
for k in (0, 1, 2, 3, 4, 5, 8, 16, 32, 64, 128, 256, 512, 1024, 0):
    o = big[k * rows_per_2m:k * rows_per_2m + n]
    print("  shift %5d x 2 MiB: gather %.4f ms" % (k, timed(lambda: e.gather(idx0, out=o))), flush=True)
ibig = torch.empty(n + 64 * (2 << 20) // 8, dtype=torch.int64, device="cuda")
o = big[:n]
print("id array shifted inside one allocation (base 0x%x):" % ibig.data_ptr())
for k in (0, 1, 2, 3, 4, 8, 16, 32, 64, 0):
    off = k * (2 << 20) // 8
    ix = ibig[off:off + n]
    ix.copy_(idx0)
    print("  shift %5d x 2 MiB: gather %.4f ms" % (k, timed(lambda: e.gather(ix, out=o))), flush=True)
