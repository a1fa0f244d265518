#!/usr/bin/env python
"""rows of whole 16-byte pieces that are not a power of two: tile sizes of the flat-stream kernel (the span-kernel variant this
script first compared is in profiles/r04_misaligned_rows.txt), settings interleaved in one process, min of 5 rounds of 10 launches.  python experiments/span_ab.py [dims ...]"""
import os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("WHOLEGRAPH_AMD_VARIANT"):
    sys.path.insert(0, os.path.join(ROOT, "experiments", "variants", os.environ["WHOLEGRAPH_AMD_VARIANT"]))
sys.path.insert(0, ROOT)
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
dims = [int(x) for x in sys.argv[1:]] or [132, 136, 160, 200, 240, 300, 400, 500, 1000]
SCAT = {"WM_ROWS_STAGED_SCATTER": "0", "WM_ROWS_FLAT": "1", "WM_ROWS_INORDER": "1"}
settings = {"gather": [("default", {}), ("flat nt loads", {"WM_ROWS_FLAT_NT": "1"})], "scatter": []}
knobs = ("WM_ROWS_FLAT_NT", "WM_ROWS_COPY_NT", "WM_ROWS_STAGED", "WM_ROWS_STAGED_ALIGN_STORES", "WM_ROWS_FLAT_TILE8", "WM_ROWS_STAGED_SCATTER", "WM_ROWS_FLAT", "WM_ROWS_INORDER", "WM_ROWS_TILE")
for dim in dims:
    rows = int(8e9 // (dim * 4)); n = int(min(10_000_000, 4e9 // (dim * 4)))
    emb = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
    t = emb.get_embedding_tensor()
    idx = torch.randint(0, rows, (n,), device="cuda")
    out = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    for op in ("gather", "scatter"):
        fn = (lambda: emb.gather(idx, out=out)) if op == "gather" else (lambda: t.scatter(out, idx))
        times = {s: [] for s, _ in settings[op]}; kern = {}
        for r in range(5):
            for name, env in settings[op]:
                for k in knobs: os.environ.pop(k, None)
                os.environ.update(env); wmb.reload_knobs()
                for _ in range(3): fn()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10): fn()
                torch.cuda.synchronize(); times[name].append((time.perf_counter() - t0) / 10 * 1e3)
                kern[name] = re.search(r"(rows_\w+<[^(]*>)\(", wmb.lib().wholememory_ext_last_rows_kernel().decode()).group(1)
        for k in knobs: os.environ.pop(k, None)
        wmb.reload_knobs()
        gb = n * (8 + 2 * dim * 4) / 1e9
        print("%-7s %5d B rows: " % (op, dim * 4) + "   ".join("%s %.3f ms %.1f%% [%s]" % (s, min(times[s]), gb / min(times[s]) / 8 * 100, kern[s][5:14]) for s, _ in settings[op]), flush=True)
    del emb, t, idx, out
    torch.cuda.empty_cache()
