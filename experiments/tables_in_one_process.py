#!/usr/bin/env python
"""Round 3: K tables of 51.2 GB alive in ONE process (the library's own allocation path), the same ids and the same dense
buffer: scatter / gather / SGD gradient apply timed on each table, two rounds. Does the level follow the table's place in VRAM?
python experiments/tables_in_one_process.py [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
rows, dim, n = 100_000_000, 128, 10_000_000
blocker_gb = [float(x) for x in os.environ.get("BLOCKERS_GB", "").split(",") if x]
blockers = [torch.empty(int(g * 1e9), dtype=torch.uint8, device="cuda") for g in blocker_gb]
embs = []
for k in range(K):
    e = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
    embs.append(e)
idx = torch.randint(0, rows, (n,), device="cuda")
dense = torch.zeros((n, dim), dtype=torch.float32, device="cuda")


REPS = int(os.environ.get("REPS", "10"))
ROUNDS = int(os.environ.get("ROUNDS", "2"))


def timed(fn, reps=REPS):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best


for rnd in range(ROUNDS):
    for k, e in enumerate(embs):
        t = e.get_embedding_tensor()
        loc = t.get_local_tensor()
        loc = loc[0] if isinstance(loc, (tuple, list)) else loc
        base = loc.data_ptr()
        probes = []
        if rnd == 0 and os.environ.get("PROBE", "1") == "1":
            import ctypes
            for kind in (1, 2, 0):   # (kind 0 overwrites rows with zeros: nothing here reads the contents)
                ms = ctypes.c_float(0)
                wmb.check(wmb.lib().wholememory_ext_probe_memory(ctypes.c_void_p(base), ctypes.c_size_t(rows * dim * 4), kind, 5, ctypes.byref(ms)))
                probes.append(ms.value)
        g = timed(lambda: e.gather(idx, out=dense))
        s = timed(lambda: t.scatter(dense, idx))
        print("round %d table %d (local base 0x%x): gather %.4f ms  scatter %.4f ms   probe ms/GiB (read, read+write back, write): %s" % (
            rnd, k, base, g, s, " ".join("%.4f" % x for x in probes)), flush=True)
