#!/usr/bin/env python
"""Placement study, part 8: output buffers built from separately created physical chunks (HIP VMM, experiments/vmm_alloc.hip),
mapped in creation order or shuffled, against hipMalloc (default) and physically contiguous allocations"""
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
vmm.vmm_free.argtypes = [C.c_void_p]
vmm.vmm_granularity.restype = C.c_size_t
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
print("VMM granularity %d KiB" % (vmm.vmm_granularity() >> 10))
nbytes = n * dim * 4
def run(label, make, free, count=4):
    res = []
    for k in range(count):
        h, ptr = make(k)
        if not ptr:
            res.append("failed"); continue
        o = torch_tensor_from_pointer(ptr, [n, dim], torch.float32, [dim, 1], True)
        res.append("%.3f" % timed(lambda: e.gather(idx, out=o)))
        del o
        free(h)
    print("%-34s %s" % (label, "  ".join(res)), flush=True)
def mk_hip(flag):
    def f(k):
        p = C.c_void_p()
        return (p, p.value) if hip.hipExtMallocWithFlags(C.byref(p), nbytes, flag) == 0 else (None, None)
    return f
def mk_vmm(chunk, shuffled):
    def f(k):
        base = C.c_void_p()
        h = vmm.vmm_alloc(nbytes, chunk, (k + 1) if shuffled else 0, C.byref(base))
        return (h, base.value) if h else (None, None)
    return f
run("hipMalloc (default)", mk_hip(0), lambda p: hip.hipFree(p))
run("hipExtMalloc contiguous", mk_hip(4), lambda p: hip.hipFree(p))
for chunk in (2 << 20, 512 << 20, 1 << 30, 2 << 30, 3 << 30, 6 << 30):
    run("VMM %4d MiB chunks" % (chunk >> 20), mk_vmm(chunk, False), lambda h: vmm.vmm_free(h))
vmm.vmm_set_exportable(1)
for chunk in (32 << 20, 6 << 30):
    run("VMM %4d MiB chunks, exportable" % (chunk >> 20), mk_vmm(chunk, False), lambda h: vmm.vmm_free(h))
vmm.vmm_set_exportable(0)
run("hipMalloc (default) again", mk_hip(0), lambda p: hip.hipFree(p))
