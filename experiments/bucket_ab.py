#!/usr/bin/env python
"""owner bucketing of 10 M ids over 8 owners (wholememory_ext_bucket_ids): time per call, uniform and Zipf ids.
WHOLEGRAPH_AMD_VARIANT=oldbucket: the library before the id loads were hoisted."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("WHOLEGRAPH_AMD_VARIANT"):
    sys.path.insert(0, os.path.join(ROOT, "experiments", "variants", os.environ["WHOLEGRAPH_AMD_VARIANT"]))
sys.path.insert(0, ROOT)
import numpy as np, torch
from wholegraph_amd import binding as wmb
from wholegraph_amd.torch.wholegraph_env import get_wholegraph_env_fns, get_stream
import ctypes as C
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
n, world, total = 10_000_000, 8, 1_000_000_000
offs = torch.tensor([total // world * r for r in range(world)] + [total], dtype=torch.int64, device="cuda")
for dist in ("uniform", "zipf"):
    for idt, wdt in ((torch.int64, wmb.DT_INT64), (torch.int32, wmb.DT_INT)):
        if dist == "uniform":
            idx = torch.randint(0, total, (n,), device="cuda").to(idt)
        else:
            k = np.random.default_rng(42).zipf(1.05, n).astype(np.uint64)
            idx = torch.from_numpy(((k * np.uint64(2654435761)) % np.uint64(total)).astype(np.int64)).cuda().to(idt)
        cnt = torch.zeros(world, dtype=torch.int64, device="cuda")
        ids = torch.zeros(n, dtype=idt, device="cuda"); raw = torch.zeros(n, dtype=torch.int64, device="cuda")
        def call():
            wmb.check(wmb.lib().wholememory_ext_bucket_ids(idx.data_ptr(), wdt, n, offs.data_ptr(), world, cnt.data_ptr(), ids.data_ptr(),
                                                           raw.data_ptr(), get_wholegraph_env_fns(), C.c_void_p(get_stream())))
        for _ in range(5): call()
        torch.cuda.synchronize(); best = 1e9
        for r in range(3):
            t0 = time.perf_counter()
            for _ in range(20): call()
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 20 * 1e6)
        print("%s bucket_ids %s %s: %.1f us per 10 M ids" % (os.environ.get("WHOLEGRAPH_AMD_VARIANT") or "product", dist, str(idt).split(".")[1], best), flush=True)
