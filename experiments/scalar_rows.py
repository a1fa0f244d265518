#!/usr/bin/env python
"""Side measurement: gather / scatter of SCALAR rows (1-D WholeMemory tensors: csr_row_ptr, csr_col, edge weights — what the
sampling path on DISTRIBUTED graphs gathers) and of very narrow 2-D rows. 1 GPU, chunked."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb

wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
n = 10_000_000
for dt, shape in [(torch.int64, [400_000_000]), (torch.int32, [400_000_000]), (torch.float32, [200_000_000, 2]),
                  (torch.float32, [100_000_000, 4]), (torch.float32, [50_000_000, 8])]:
    es = torch.empty((), dtype=dt).element_size()
    t = wgth.create_wholememory_tensor(comm, "chunked", "cuda", shape, dt, [1] if len(shape) == 1 else [shape[1], 1])
    rows = shape[0]
    row_b = es * (shape[1] if len(shape) > 1 else 1)
    idx = torch.randint(0, rows, (n,), device="cuda")
    src = torch.zeros([n] + shape[1:], dtype=dt, device="cuda")
    for op in ("gather", "scatter"):
        fn = (lambda: t.gather(idx)) if op == "gather" else (lambda: t.scatter(src, idx))
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print("%-7s %-8s rows of %2d B, %d ids: %.3f ms  %.1f G rows/s  (torch index: see below)" % (
            op, str(dt).split(".")[1], row_b, n, ms, n / ms / 1e6), flush=True)
    local, _ = t.get_local_tensor()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        local[idx]
    torch.cuda.synchronize()
    print("        torch local[idx]: %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3), flush=True)
    wgth.destroy_wholememory_tensor(t)
