#!/usr/bin/env python
"""Round 6, review item 5: where do step_tile_kernel's 11 points below the scatter go? The SAME fused step (SGD, 10 M gradient
rows of 512 B on a 100 M-row fp32 table) on four batches that differ only in which side is random — all four in ONE process
(same table, same gradient buffer: the placement lottery of DESIGN 3.1b cannot tell them apart), interleaved over rounds:
  as_is        uniform random ids, gradient rows in the caller's order  (table walk ascending but scattered, gradient read random)
  grads_seq    the same ids sorted beforehand, gradient rows in that order (order[] = identity: gradient read sequential)
  table_dense  a random permutation of rows 0 .. n-1 (table walk contiguous, gradient read random)
  both_seq     rows 0 .. n-1 in order (table walk contiguous, gradient read sequential: the copy level of this kernel)
python experiments/step_tile_decomp.py [steps] [rounds]  -> per round and variant the whole-call time (HIP events); under
rocprofv3 the step_tile_kernel launches come in the order printed on the `plan` line (experiments/step_tile_decomp_parse.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
WARM = 2
rows, dim, n = 100_000_000, 128, 10_000_000
emb = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
wgth.create_wholememory_optimizer(emb, "sgd", {})
g = torch.Generator(device="cuda"); g.manual_seed(42)
uni = torch.randint(0, rows, (n,), device="cuda", generator=g)
batches = [("as_is", uni), ("grads_seq", torch.sort(uni).values), ("table_dense", torch.randperm(n, device="cuda", generator=g)),
           ("both_seq", torch.arange(n, device="cuda"))]
grads = torch.randn((n, dim), device="cuda")
print("plan: rounds %d x variants %s x (%d warm-up + %d timed) calls" % (rounds, ",".join(b[0] for b in batches), WARM, steps), flush=True)
def step(idx):
    emb.add_gradients(idx, grads); emb.need_apply = True; emb.apply_gradients(0.01)
for r in range(rounds):
    for name, idx in batches:
        for _ in range(WARM): step(idx)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps): step(idx)
        e1.record(); torch.cuda.synchronize()
        nu = n if name != "as_is" and name != "grads_seq" else int(torch.unique(idx).numel())
        ms = e0.elapsed_time(e1) / steps
        alg = nu * 1544 + (n - nu) * 520
        print("round %d %-12s whole call %.4f ms  unique %d  algorithmic %.3f GB  -> %.1f %% of 8 TB/s" % (
            r, name, ms, nu, alg / 1e9, alg / (ms * 1e-3) / 8e12 * 100), flush=True)
