#!/usr/bin/env python
"""Latency of SMALL ops (serving-sized batches): host enqueue time and end-to-end time per call of gather / scatter / gradient apply
for 64 ... 65536 rows of 512 bytes (chunked table, one GPU), against torch's own index_select / index_copy_ on a plain tensor of the
same shape (the floor a framework user compares with)."""
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
t = emb.get_embedding_tensor()
plain = torch.zeros((rows, dim), device="cuda")
def measure(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6
for n in (64, 1024, 16384, 65536):
    idx = torch.randint(0, rows, (n,), device="cuda")
    out = torch.empty((n, dim), device="cuda")
    g = torch.randn((n, dim), device="cuda")
    def grad():
        emb.add_gradients(idx, g); emb.need_apply = True; emb.apply_gradients(0.01)
    res = [("gather", measure(lambda: emb.gather(idx, out=out))), ("scatter", measure(lambda: t.scatter(out, idx))),
           ("gradient apply", measure(grad, 100)),
           ("torch index_select", measure(lambda: torch.index_select(plain, 0, idx, out=out))),
           ("torch index_copy_", measure(lambda: plain.index_copy_(0, idx, out)))]
    print("n = %6d rows: " % n + "   ".join("%s host %.1f / total %.1f us" % (k, h, tt) for k, (h, tt) in res), flush=True)
