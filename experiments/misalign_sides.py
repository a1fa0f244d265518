#!/usr/bin/env python
"""Which side of a 4000 B row copy pays for rows that start on 32-byte instead of 4 KiB multiples? dim 1000 fp32 rows, the table
and the dense side each either packed (stride 1000) or padded (stride 1024), gather and scatter, kernels interleaved."""
import os, re, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
from wholegraph_amd.torch.wholegraph_env import wrap_torch_tensor, get_wholegraph_env_fns, get_stream
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
pad = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rows, n = int(8e9 // (pad * 4)), 1_000_000
idx = torch.randint(0, rows, (n,), device="cuda")
knobs = ("WM_ROWS_SPAN",)
for ts in (dim, pad):
    t = wgth.create_wholememory_tensor(comm, "chunked", "cuda", [rows, dim], torch.float32, [ts, 1])
    for os_ in (dim, pad):
        big = torch.empty((n, os_), dtype=torch.float32, device="cuda")
        out = big[:, :dim]
        wi, wo = wrap_torch_tensor(idx), wrap_torch_tensor(out)
        def gather():
            wmb.check(wmb.lib().wholememory_gather(t.wmb_tensor, wi.handle, wo.handle, get_wholegraph_env_fns(), C.c_void_p(get_stream()), -1))
        def scatter():
            wmb.check(wmb.lib().wholememory_scatter(wo.handle, wi.handle, t.wmb_tensor, get_wholegraph_env_fns(), C.c_void_p(get_stream()), -1))
        for op, fn in (("gather", gather), ("scatter", scatter)):
            res = []
            for name, env in (("default", {}), ("span", {"WM_ROWS_SPAN": "1", "WM_ROWS_STAGED_SCATTER": "0", "WM_ROWS_FLAT": "1", "WM_ROWS_INORDER": "1"})):
                best = 1e9
                for r in range(3):
                    for k in ("WM_ROWS_SPAN", "WM_ROWS_STAGED_SCATTER", "WM_ROWS_FLAT", "WM_ROWS_INORDER"): os.environ.pop(k, None)
                    os.environ.update(env); wmb.reload_knobs()
                    for _ in range(3): fn()
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in range(10): fn()
                    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
                kern = re.search(r"(rows_\w+<[^(]*>)\(", wmb.lib().wholememory_ext_last_rows_kernel().decode()).group(1)
                res.append("%s %.3f ms %.1f%% [%s]" % (name, best, n * (8 + 8 * dim) / best / 8e9 * 100, kern[:36]))
            print("%-7s table stride %4d  dense stride %4d : %s" % (op, ts, os_, "   ".join(res)), flush=True)
        del big, out
    del t
