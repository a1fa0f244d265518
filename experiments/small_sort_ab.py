#!/usr/bin/env python
"""Gradient apply of SMALL batches (the gradients of one mini-batch): rocPRIM's merge sort (what radix_sort_pairs takes up to 2^20
items) against the onesweep passes for the id sort, interleaved in one process. 10 M x 128 fp32 table, SGD, uniform ids."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
rows, dim = 10_000_000, 128
emb = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
wgth.create_wholememory_optimizer(emb, "sgd", {})
for n in (8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576, 2097152):
    idx = torch.randint(0, rows, (n,), device="cuda")
    g = torch.randn((n, dim), device="cuda")
    def step():
        emb.add_gradients(idx, g); emb.need_apply = True; emb.apply_gradients(0.01)
    res = []
    for name, v in (("merge sort allowed", str(1 << 40)), ("radix passes", "1")):
        best = 1e9
        for r in range(3):
            os.environ["WM_SORT_RADIX_MIN"] = v; wmb.reload_knobs()
            for _ in range(5): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50): step()
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 50 * 1e6)
        res.append("%s %.1f us" % (name, best))
    print("n = %8d gradient rows: " % n + "   ".join(res), flush=True)
