#!/usr/bin/env python
"""Round 6, review item 5: where do step_tile_kernel's 11 points below the scatter go? The SAME fused step (SGD, 10 M gradient
rows of 512 B on a 100 M-row fp32 table, one process) on four batches that differ only in which side is random:
  as_is        uniform random ids, gradient rows in the caller's order  (table walk ascending but scattered, gradient read random)
  grads_seq    the same ids sorted beforehand, gradient rows in that order (order[] = identity: gradient read sequential)
  table_dense  a random permutation of rows 0 .. n-1 (table walk contiguous, gradient read random)
  both_seq     rows 0 .. n-1 in order (table walk contiguous, gradient read sequential: the copy level of this kernel)
python experiments/step_tile_decomp.py <variant> [steps]   -> one line with the whole-call time (HIP events)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
variant, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows, dim, n = 100_000_000, 128, 10_000_000
emb = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
wgth.create_wholememory_optimizer(emb, "sgd", {})
g = torch.Generator(device="cuda"); g.manual_seed(42)
if variant == "as_is":
    idx = torch.randint(0, rows, (n,), device="cuda", generator=g)
elif variant == "grads_seq":
    idx = torch.sort(torch.randint(0, rows, (n,), device="cuda", generator=g)).values
elif variant == "table_dense":
    idx = torch.randperm(n, device="cuda", generator=g)
elif variant == "both_seq":
    idx = torch.arange(n, device="cuda")
else:
    raise SystemExit("variant?")
grads = torch.randn((n, dim), device="cuda")
def step():
    emb.add_gradients(idx, grads); emb.need_apply = True; emb.apply_gradients(0.01)
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps): step()
e1.record(); torch.cuda.synchronize()
nu = int(torch.unique(idx).numel())
ms = e0.elapsed_time(e1) / steps
alg = nu * 1544 + (n - nu) * 520
print("%-12s whole call %.4f ms  unique %d  algorithmic %.3f GB  -> %.1f %% of 8 TB/s" % (variant, ms, nu, alg / 1e9, alg / ms / 8e6 * 1e-3 * 100 / 1e0 if False else alg / (ms * 1e-3) / 8e12 * 100), flush=True)
