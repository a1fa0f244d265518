#!/usr/bin/env python
"""Latency of small calls: gather / scatter / gradient apply of n ids end to end (python -> C ABI -> kernel -> sync) and the
host time to queue one call, against torch indexing of the same tensors."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
rows, dim = 10_000_000, 128
for mt in ("chunked", "distributed"):
    emb = wgth.create_embedding(comm, mt, "cuda", torch.float32, [rows, dim])
    local, _ = emb.get_embedding_tensor().get_local_tensor()
    for n in (1, 1000, 100_000, 1_000_000):
        idx = torch.randint(0, rows, (n,), device="cuda")
        out = torch.empty((n, dim), device="cuda")
        def bench(fn, reps=200):
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            return (t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6
        hq, tot = bench(lambda: emb.gather(idx, out=out))
        hq2, tot2 = bench(lambda: emb.gather(idx))
        hq3, tot3 = bench(lambda: torch.index_select(local, 0, idx, out=out))
        print("%-11s n=%8d: gather(out=) host %6.1f us total %7.1f us | gather() host %6.1f total %7.1f | torch.index_select host %6.1f total %7.1f" % (
            mt, n, hq, tot, hq2, tot2, hq3, tot3), flush=True)
    wgth.destroy_embedding(emb)
