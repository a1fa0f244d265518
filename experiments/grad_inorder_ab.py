#!/usr/bin/env python
"""gradient apply, whole call (10 M gradient rows): step_tile_kernel launched in order (one batch of runs per wave, round 3)
against round 2's persistent grid over tiles of 64 runs, and in-order tiles of 2 / 4 / 8 batches, interleaved in one process.
  python experiments/grad_inorder_ab.py [optimizer=sgd] [dist=uniform|zipf] [dim=128] [dtype=f32|f16]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
kind = sys.argv[1] if len(sys.argv) > 1 else "sgd"
dist = sys.argv[2] if len(sys.argv) > 2 else "uniform"
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dt = {"f32": torch.float32, "f16": torch.float16}[sys.argv[4] if len(sys.argv) > 4 else "f32"]
es = 4 if dt == torch.float32 else 2
rows, n = int(51.2e9 // (dim * es)) if kind == "sgd" else int(25.6e9 // (dim * es)), 10_000_000
emb = wgth.create_embedding(comm, "chunked", "cuda", dt, [rows, dim])
wgth.create_wholememory_optimizer(emb, kind, {})
if dist == "uniform":
    idx = torch.randint(0, rows, (n,), device="cuda")
else:
    k = np.random.default_rng(42).zipf(1.05, n).astype(np.uint64)
    idx = torch.from_numpy(((k * np.uint64(2654435761)) % np.uint64(rows)).astype(np.int64)).cuda()
g = torch.randn((n, dim), device="cuda").to(dt)
def step():
    emb.add_gradients(idx, g); emb.need_apply = True; emb.apply_gradients(0.01)
def timed(reps=20):
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
settings = [("persistent", {"WM_TILE_INORDER": "0"}), ("inorder", {}), ("inorder x2", {"WM_TILE_RUNS": "x2"}),
            ("inorder x4", {"WM_TILE_RUNS": "x4"}), ("inorder 64", {"WM_TILE_RUNS": "64"})]
batch = int(os.environ.get("AB_BATCH", "8"))
for r in range(3):
    out = []
    for name, env in settings:
        for k in ("WM_TILE_INORDER", "WM_TILE_RUNS"):
            os.environ.pop(k, None)
        for k, v in env.items():
            os.environ[k] = str(batch * int(v[1:])) if v.startswith("x") else v
        __import__("wholegraph_amd.binding").binding.reload_knobs()   # knobs are read once
        out.append("%s %.4f" % (name, timed()))
    print("%s %s dim %d %s round %d (ms per call): " % (kind, dist, dim, str(dt).split(".")[1], r) + "   ".join(out), flush=True)
