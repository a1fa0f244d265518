#!/usr/bin/env python
"""Placement study, part 4: a TLB-bound probe per 256 MiB chunk of a buffer (one 4-byte write per 4 KiB page, and one per
64 KiB) — do the chunks of an allocation differ in how they are mapped (fragment / page size)? Compared with the gather level
of output windows placed at the same offsets."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
rows, dim, n = 100_000_000, 128, 10_000_000
def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
e = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
idx0 = torch.randint(0, rows, (n,), device="cuda")
chunk_rows = (256 << 20) // (dim * 4)
n_chunks = 28
big = torch.empty((n_chunks * chunk_rows, dim), device="cuda")   # 7 GiB
flat = big.view(-1)
print("per 256 MiB chunk: 4 KiB-stride probe / 64 KiB-stride probe (us), then gather into a 5 GB window starting there")
for c in range(n_chunks):
    piece = flat[c * (64 << 20):(c + 1) * (64 << 20)]          # 64 Mi floats = 256 MiB
    p4k = piece.view(-1, 1024)[:, 0]
    p64k = piece.view(-1, 16384)[:, 0]
    t4 = timed(lambda: p4k.fill_(1.0), 50) * 1e3
    t64 = timed(lambda: p64k.fill_(1.0), 50) * 1e3
    g = ""
    if (c + 1) * chunk_rows + n <= big.shape[0] + chunk_rows and c * chunk_rows + n <= big.shape[0]:
        o = big[c * chunk_rows:c * chunk_rows + n]
        g = "gather %.4f ms" % timed(lambda: e.gather(idx0, out=o))
    print("  chunk %2d: %7.1f / %7.1f us   %s" % (c, t4, t64, g), flush=True)
