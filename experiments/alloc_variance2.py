#!/usr/bin/env python
"""alloc_variance.py, part 2: which of the two buffers matters? Table fixed / output re-allocated, then the reverse."""
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
def timed(fn, reps=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
emb = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
print("table fixed, output re-allocated:")
fillers = []
for rnd in range(6):
    out = torch.empty((n, dim), device="cuda")
    print("  out at 0x%x: gather %.4f ms" % (out.data_ptr(), timed(lambda: emb.gather(idx, out=out))), flush=True)
    fillers.append(torch.empty(int(1e9) + rnd * 12345 * 4096, dtype=torch.uint8, device="cuda"))
    del out
del fillers
torch.cuda.empty_cache()
out = torch.empty((n, dim), device="cuda")
print("output fixed at 0x%x, table re-created:" % out.data_ptr())
wgth.destroy_embedding(emb)
for rnd in range(6):
    emb = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
    local, _ = emb.get_embedding_tensor().get_local_tensor()
    print("  table at 0x%x: gather %.4f ms" % (local.data_ptr(), timed(lambda: emb.gather(idx, out=out))), flush=True)
    del local
    wgth.destroy_embedding(emb)
