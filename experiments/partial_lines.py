#!/usr/bin/env python
"""Why rows that are not whole 128-byte lines scatter (and gather) below the power-of-two shapes: the SAME table (stride 256
floats = 1 KiB, every row starts on a line boundary), scattered / gathered through column views of different widths — the only
thing that changes is whether the last line of a row is written (read) in part."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
rows, stride, n = 8_000_000, 256, 5_000_000
t = wgth.create_wholememory_tensor(comm, "chunked", "cuda", [rows, stride], torch.float32, [stride, 1])
idx = torch.randperm(rows, device="cuda")[:n].contiguous()
def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print("row stride 1 KiB, %d random rows; width = columns moved per row" % n)
for width in (64, 96, 100, 104, 112, 128, 160, 192, 200, 224, 256):
    sub = t.get_sub_tensor([0, 0], [-1, width])
    buf = torch.empty((n, width), device="cuda")
    ts = timed(lambda: sub.scatter(buf, idx))
    tg = timed(lambda: wgth.wholememory_ops.gather(sub, idx, buf) if hasattr(wgth, "wholememory_ops") and hasattr(wgth.wholememory_ops, "gather") else sub.gather(idx))
    nb = width * 4
    lines = (nb + 127) // 128
    print("width %3d (%4d B = %d lines%s): scatter %.3f ms = %5.2f ns/row, %4.1f %% of 8 TB/s | gather %.3f ms" % (
        width, nb, lines, "" if nb % 128 == 0 else ", last one partial", ts, ts * 1e6 / n, n * (8 + 2 * nb) / ts / 8e9 * 100, tg), flush=True)
