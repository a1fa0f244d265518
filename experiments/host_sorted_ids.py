"""Does the reference's sorted-ids trick for HOST tables (gather_op.cpp:116-120) pay on MI355X?  C1: 10 M x 64 fp32
host table, 1 M ids: gather time with the ids as they come vs pre-sorted ids (sort not timed: upper bound of the gain)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
torch.cuda.set_device(0)
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
for rows, dim, n in [(10_000_000, 64, 1_000_000), (10_000_000, 128, 1_000_000), (40_000_000, 64, 4_000_000)]:
    emb = wgth.create_embedding(comm, "chunked", "cpu", torch.float32, [rows, dim])
    local, _ = emb.get_embedding_tensor().get_local_tensor(host_view=True)
    local.fill_(1.0)
    idx = torch.from_numpy(np.random.default_rng(42).integers(0, rows, n, dtype=np.int64)).cuda()
    sidx = torch.sort(idx).values
    out = torch.empty((n, dim), device="cuda")
    for name, ids in (("random", idx), ("sorted", sidx), ("random", idx), ("sorted", sidx)):
        for _ in range(3):
            emb.gather(ids, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            emb.gather(ids, out=out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 100
        print("%d x %d, %d ids, %s: %.3f ms  %.1f GB/s" % (rows, dim, n, name, ms, n * dim * 4 / ms / 1e6), flush=True)
    t0 = time.perf_counter()
    for _ in range(10):
        torch.sort(idx)
    torch.cuda.synchronize()
    print("   torch.sort of the ids: %.3f ms" % ((time.perf_counter() - t0) * 100), flush=True)
    wgth.destroy_embedding(emb)
