#!/usr/bin/env python
"""Placement study, part 10: the bench's situation — one 51 GB table, then six torch.empty (hipMalloc) output buffers and six
buffers built from 512 MiB HIP VMM chunks, all alive: which kind yields the fast pairs?"""
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
vmm = C.CDLL(os.path.join(ROOT, "experiments", "libvmm_alloc.so"))
vmm.vmm_alloc.restype = C.c_void_p
vmm.vmm_alloc.argtypes = [C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
rows, dim, n = 100_000_000, 128, 10_000_000
order = sys.argv[1] if len(sys.argv) > 1 else "torch-first"
e = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
idx = torch.randint(0, rows, (n,), device="cuda")
def timed(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
chunk = int(sys.argv[2]) << 20 if len(sys.argv) > 2 else 512 << 20
stride = int(sys.argv[3]) if len(sys.argv) > 3 else 1
vmm.vmm_set_stride(stride)
def vmm_tensor():
    base = C.c_void_p()
    h = vmm.vmm_alloc(n * dim * 4, chunk, 0, C.byref(base))
    assert h
    return torch_tensor_from_pointer(base.value, [n, dim], torch.float32, [dim, 1], True)
if order == "torch-first":
    t_outs = [torch.empty((n, dim), device="cuda") for _ in range(6)]
    v_outs = [vmm_tensor() for _ in range(6)]
else:
    v_outs = [vmm_tensor() for _ in range(6)]
    t_outs = [torch.empty((n, dim), device="cuda") for _ in range(6)]
print("%s chunk %d MiB stride %d | torch.empty: %s | VMM chunks: %s" % (order, chunk >> 20, stride, " ".join("%.3f" % timed(lambda: e.gather(idx, out=o)) for o in t_outs),
                                                 " ".join("%.3f" % timed(lambda: e.gather(idx, out=o)) for o in v_outs)), flush=True)
