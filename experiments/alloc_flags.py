#!/usr/bin/env python
"""Placement study, part 7: output buffers from hipExtMallocWithFlags — default against physically CONTIGUOUS VRAM
(hipDeviceMallocContiguous) — behind one table: do contiguous allocations land on one level?"""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
from wholegraph_amd.torch.wholegraph_env import torch_tensor_from_pointer
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
rows, dim, n = 100_000_000, 128, 10_000_000
idx = torch.randint(0, rows, (n,), device="cuda")
def timed(fn, reps=12):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
e = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
for name, flag in (("default", 0x0), ("contiguous", 0x4), ("default", 0x0), ("contiguous", 0x4)):
    res = []
    keep = []
    for k in range(6):
        p = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(p), n * dim * 4, flag)
        if rc != 0:
            res.append("alloc failed (%d)" % rc)
            continue
        keep.append(p)
        o = torch_tensor_from_pointer(p.value, [n, dim], torch.float32, [dim, 1], True)
        res.append("%.3f" % timed(lambda: e.gather(idx, out=o)))
    print("%-10s outputs: %s" % (name, "  ".join(res)), flush=True)
    for p in keep:
        hip.hipFree(p)
