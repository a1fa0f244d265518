#!/usr/bin/env python
"""Does the C2 gather time depend on WHERE the 51 GB table landed? One process: create the table, time the gather, destroy,
repeat (the allocator is emptied in between so that every round maps fresh memory)."""
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
for rnd in range(6):
    if rnd % 2 == 1:
        filler = torch.empty(int(3e9) * (rnd + 1), dtype=torch.uint8, device="cuda")   # shift where the next table lands
    emb = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
    out = torch.empty((n, dim), device="cuda")
    ts = []
    for rep in range(3):
        for _ in range(3):
            emb.gather(idx, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            emb.gather(idx, out=out)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 50 * 1e3)
    local, _ = emb.get_embedding_tensor().get_local_tensor()
    def timed(fn, reps=20):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    sidx = torch.sort(idx).values
    t_sorted = timed(lambda: emb.gather(sidx, out=out))
    flat = local.view(-1)
    t_stream = [timed(lambda: out.view(-1).copy_(flat[o:o + out.numel()])) for o in (0, flat.numel() // 2, flat.numel() - out.numel())]
    scat = timed(lambda: emb.get_embedding_tensor().scatter(out, idx))
    print("round %d: table at 0x%x, out at 0x%x: gather %s ms | sorted ids %.4f | stream copy of 5 GB table slices -> out %s | scatter %.4f" % (
        rnd, local.data_ptr(), out.data_ptr(), " ".join("%.4f" % t for t in ts), t_sorted, " ".join("%.4f" % t for t in t_stream), scat), flush=True)
    del sidx, flat
    wgth.destroy_embedding(emb)
    del out, local
    if rnd % 2 == 1:
        del filler
    torch.cuda.empty_cache()
