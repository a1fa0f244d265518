#!/usr/bin/env python
"""gradient apply (SGD, 10 M uniform rows): step_tile_kernel at its natural 5 waves / SIMD against variants forced to 6 / 7 / 8
(register limits with spills), interleaved in one process. The 6 / 8 / other-width variants were compiled in for the experiment
only; the library keeps 7 for the 512-byte-row SGD kernel and WM_TILE_OCC=5 as its off switch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rows, n = int(51.2e9 // (dim * 4)), 10_000_000
emb = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
wgth.create_wholememory_optimizer(emb, "sgd", {})
idx = torch.randint(0, rows, (n,), device="cuda")
g = torch.randn((n, dim), device="cuda")
def step():
    emb.add_gradients(idx, g); emb.need_apply = True; emb.apply_gradients(0.01)
def timed(reps=20):
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for r in range(3):
    out = []
    for occ in ("5", "7"):
        os.environ["WM_TILE_OCC"] = occ
        out.append("occ %s: %.4f ms" % (occ, timed()))
    print("dim %d round %d: " % (dim, r) + "   ".join(out), flush=True)
