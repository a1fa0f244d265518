#!/usr/bin/env python
"""C5's feature gather runs over the outermost frontier's UPPER-BOUND array (valid ids ++ -1 up to the room): what do the skipped
entries cost? One process, the same 51 GB table: gather of n_valid ids ++ pad x (-1) against gather of the n_valid ids alone."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("WHOLEGRAPH_AMD_VARIANT"):   # scripts/build_variant.sh validprefix "-DWM_EXP_VALID_PREFIX=1": dead tiles leave on a scalar load
    sys.path.insert(0, os.path.join(ROOT, "experiments", "variants", os.environ["WHOLEGRAPH_AMD_VARIANT"]))
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
rows, dim = 100_000_000, 128
emb = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
room = 1024 * 31 * 31
for n_valid in (600_000, 400_000, 984_064):
    ids = torch.full((room,), -1, dtype=torch.int32, device="cuda")
    ids[:n_valid] = torch.randint(0, rows, (n_valid,), device="cuda", dtype=torch.int32)
    out = torch.empty((room, dim), device="cuda")
    nv = torch.tensor([n_valid], dtype=torch.int32, device="cuda")
    def run(idx, o, reps=200):
        for _ in range(10): emb.gather(idx, out=o)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): emb.gather(idx, out=o)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6
    for r in range(2):
        a = run(ids, out)
        b = run(ids[:n_valid].contiguous(), out[:n_valid])
        c = float("nan")
        if os.environ.get("WHOLEGRAPH_AMD_VARIANT"):
            os.environ["WM_EXP_VALID_PTR"] = hex(nv.data_ptr())
            c = run(ids, out)
            os.environ.pop("WM_EXP_VALID_PTR")
        print("valid %7d of %d: padded %.1f us   trimmed %.1f us   tail costs %.1f us   padded + device-side bound %.1f us" % (n_valid, room, a, b, a - b, c), flush=True)
