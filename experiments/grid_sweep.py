#!/usr/bin/env python
"""C2 gather / scatter time against the workgroup cap (`gather_sms` / `scatter_sms` argument of the C API)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
from wholegraph_amd.torch.wholegraph_env import wrap_torch_tensor, get_wholegraph_env_fns, get_stream
torch.cuda.set_device(0)
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
rows, dim, n = 100_000_000, 128, 10_000_000
t = wgth.create_wholememory_tensor(comm, "chunked", "cuda", [rows, dim], torch.float32, [dim, 1])
idx = torch.randint(0, rows, (n,), device="cuda")
out = torch.zeros((n, dim), device="cuda")
wi, wo = wrap_torch_tensor(idx), wrap_torch_tensor(out)
env = get_wholegraph_env_fns()
for op in ("gather", "scatter"):
    for cap in (-1, 1024, 2048, 4096, 8192, 12288, 16384, 32768, 65536):
        def call():
            if op == "gather":
                wmb.check(wmb.lib().wholememory_gather(t.wmb_tensor, wi, wo, env, C.c_void_p(get_stream()), cap))
            else:
                wmb.check(wmb.lib().wholememory_scatter(wo, wi, t.wmb_tensor, env, C.c_void_p(get_stream()), cap))
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        print("%s cap %6d: %.4f ms" % (op, cap, e0.elapsed_time(e1) / 20), flush=True)
