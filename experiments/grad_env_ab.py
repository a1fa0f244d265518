#!/usr/bin/env python
"""gradient apply, whole call (10 M gradient rows), under several environment settings interleaved in ONE process (the level of
a call depends on the process's table placement, so only in-process comparisons mean anything).
  python experiments/grad_env_ab.py <optimizer> <uniform|zipf> <dim> <f32|f16> "name:VAR=v,VAR2=v;name2:VAR=v;..." """
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("WHOLEGRAPH_AMD_VARIANT"):   # scripts/build_variant.sh NAME: A/B of compile-time variants
    sys.path.insert(0, os.path.join(ROOT, "experiments", "variants", os.environ["WHOLEGRAPH_AMD_VARIANT"]))
import numpy as np
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
kind, dist, dim = sys.argv[1], sys.argv[2], int(sys.argv[3])
dt = {"f32": torch.float32, "f16": torch.float16}[sys.argv[4]]
settings = []
for item in sys.argv[5].split(";"):
    name, _, envs = item.partition(":")
    settings.append((name, dict(kv.split("=") for kv in envs.split(",") if kv)))
es = 4 if dt == torch.float32 else 2
rows, n = (int(51.2e9 // (dim * es)) if kind == "sgd" else int(25.6e9 // (dim * es))), int(os.environ.get("AB_IDS", "10000000"))
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
def timed(reps=int(os.environ.get("AB_REPS", "20"))):
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
keys = sorted({k for _, e in settings for k in e})
for r in range(3):
    out = []
    for name, env in settings:
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(env)
        __import__("wholegraph_amd.binding").binding.reload_knobs()   # knobs are read once
        out.append("%s %.4f" % (name, timed()))
    print("%s %s dim %d %s round %d (ms per call): " % (kind, dist, dim, str(dt).split(".")[1], r) + "   ".join(out), flush=True)
