#!/usr/bin/env python
"""Placement study, part 9: table and output buffer each from hipMalloc or from HIP VMM chunks of 512 MiB
(experiments/vmm_alloc.hip): gather and scatter levels of the four combinations (plain-pointer tables through
wholememory_gather / wholememory_scatter), two instances each"""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
from wholegraph_amd.torch.wholegraph_env import torch_tensor_from_pointer, wrap_torch_tensor, get_wholegraph_env_fns, get_stream
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
vmm = C.CDLL(os.path.join(ROOT, "experiments", "libvmm_alloc.so"))
vmm.vmm_alloc.restype = C.c_void_p
vmm.vmm_alloc.argtypes = [C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
rows, dim, n = 100_000_000, 128, 10_000_000
idx = torch.randint(0, rows, (n,), device="cuda")
def timed(fn, reps=12):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
def vmm_tensor(r, chunk=512 << 20):
    base = C.c_void_p()
    h = vmm.vmm_alloc(r * dim * 4, chunk, 0, C.byref(base))
    assert h, "vmm_alloc failed"
    return torch_tensor_from_pointer(base.value, [r, dim], torch.float32, [dim, 1], True)
tables = {"hipMalloc": [torch.empty((rows, dim), device="cuda") for _ in range(2)], "VMM chunks": [vmm_tensor(rows) for _ in range(2)]}
outs = {"hipMalloc": [torch.empty((n, dim), device="cuda") for _ in range(2)], "VMM chunks": [vmm_tensor(n) for _ in range(2)]}
L = wmb.lib()
wi = wrap_torch_tensor(idx)
def gather(t, o):
    wt, wo = wrap_torch_tensor(t), wrap_torch_tensor(o)
    wmb.check(L.wholememory_gather(wt.handle, wi.handle, wo.handle, get_wholegraph_env_fns(), C.c_void_p(get_stream()), -1))
def scatter(t, o):
    wt, wo = wrap_torch_tensor(t), wrap_torch_tensor(o)
    wmb.check(L.wholememory_scatter(wo.handle, wi.handle, wt.handle, get_wholegraph_env_fns(), C.c_void_p(get_stream()), -1))
for tk, tl in tables.items():
    for ok, ol in outs.items():
        g = ["%.3f" % timed(lambda: gather(t, o)) for t in tl for o in ol]
        s = ["%.3f" % timed(lambda: scatter(t, o)) for t in tl for o in ol]
        print("table %-10s buffer %-10s: gather %s | scatter %s" % (tk, ok, " ".join(g), " ".join(s)), flush=True)
