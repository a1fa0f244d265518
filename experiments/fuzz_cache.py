#!/usr/bin/env python
"""Random op sequences on a cached embedding (HOST table + read-write device row cache) against an uncached twin: gathers
with and without cache adjustment, training steps (SGD / LazyAdam / AdaGrad) through the cache, write-backs, drops — the
cache must stay transparent: every gather equal bit for bit, and after a final write-back the tables and optimizer states
equal too. usage: fuzz_cache.py [sequences] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb

torch.cuda.set_device(0)
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_ERROR))
comm = wgth.create_group_communicator(1)
seqs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for s in range(seqs):
    n_rows = int(rng.integers(500, 40000))
    dim = int(rng.choice([4, 16, 33, 64, 100, 128]))
    kind = ["sgd", "adam", "adagrad"][rng.integers(3)]
    mt = ["chunked", "continuous", "distributed"][rng.integers(3)]
    ratio = float(rng.choice([0.02, 0.1, 0.3, 0.9]))
    desc = "seq %d: %s %s rows %d dim %d ratio %.2f" % (s, mt, kind, n_rows, dim, ratio)
    policy = wgth.create_wholememory_cache_policy(comm, memory_type=mt, memory_location="cuda", access_type="readwrite", ratio=ratio)
    cached = wgth.create_embedding(comm, mt, "cpu", torch.float32, [n_rows, dim], cache_policy=policy)
    plain = wgth.create_embedding(comm, mt, "cuda", torch.float32, [n_rows, dim])
    init = torch.randn(n_rows, dim)
    cached.get_embedding_tensor().get_local_tensor(host_view=True)[0].copy_(init)
    plain.get_embedding_tensor().get_local_tensor()[0].copy_(init.cuda())
    for e in (cached, plain):
        wgth.create_wholememory_optimizer(e, kind, {"weight_decay": 0.01})
    ok = True
    log = []
    try:
        for step in range(int(rng.integers(5, 25))):
            op = rng.choice(["gather", "gather", "train", "train", "writeback", "drop"])
            n = int(rng.choice([1, 50, 3000, 20000]))
            if rng.random() < 0.5:
                idx = (rng.zipf(1.2, n).astype(np.uint64) * np.uint64(2654435761) % np.uint64(n_rows)).astype(np.int64)
            else:
                idx = rng.integers(0, n_rows, n)
            if n > 10 and op == "gather":
                idx[::13] = -1
            t_idx = torch.from_numpy(idx).cuda()
            log.append("%s(%d)" % (op, n))
            if op == "gather":
                cached.set_adjust_cache(bool(rng.integers(2)))
                a = cached.gather(t_idx, out=torch.full((n, dim), 3.0, device="cuda"))
                b = plain.gather(t_idx, out=torch.full((n, dim), 3.0, device="cuda"))
                torch.cuda.synchronize()
                ok = ok and torch.equal(a, b)
            elif op == "train":
                g = torch.randn(n, dim, device="cuda")
                cached.set_adjust_cache(bool(rng.integers(2)))
                for e in (cached, plain):
                    e.add_gradients(t_idx, g)
                    e.need_apply = True
                    e.apply_gradients(0.05)
                torch.cuda.synchronize()
            elif op == "writeback":
                cached.writeback_all_cache()
            else:
                cached.drop_all_cache()
            if not ok:
                break
        cached.writeback_all_cache()
        torch.cuda.synchronize()
        ta = cached.get_embedding_tensor().get_local_tensor(host_view=True)[0]
        tb = plain.get_embedding_tensor().get_local_tensor()[0].cpu()
        ok = ok and torch.equal(ta, tb)
        for name in cached.get_optimizer_state_names():
            sa = cached.get_optimizer_state(name).get_local_tensor(host_view=True)[0] if name != "beta12t" else \
                cached.get_optimizer_state(name).get_local_tensor()[0].cpu()
            sb = plain.get_optimizer_state(name).get_local_tensor()[0].cpu()
            ok = ok and torch.equal(sa, sb)
    except Exception as ex:  # noqa
        ok = False
        print("ERROR", repr(ex)[:400], flush=True)
    if not ok:
        bad += 1
        print("MISMATCH", desc, " ".join(log), flush=True)
    wgth.destroy_embedding(cached)
    wgth.destroy_embedding(plain)
print("sequences %d, failures %d" % (seqs, bad))
