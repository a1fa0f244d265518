#!/usr/bin/env python
"""Side measurement: gather / scatter WITH a dtype cast (table dtype != plain dtype), 8 GB table, 10 M ids.
python experiments/cast_sweep.py            one setting (the library's defaults)
python experiments/cast_sweep.py --ab       interleaved A/B of the launch shapes of rows_convert_kernel / rows_copy_kernel:
                                            in-order ~4 KiB tiles (default), in-order 64-row tiles, persistent grid"""
import os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("WHOLEGRAPH_AMD_VARIANT"):   # scripts/build_variant.sh NAME: A/B of compile-time variants
    sys.path.insert(0, os.path.join(ROOT, "experiments", "variants", os.environ["WHOLEGRAPH_AMD_VARIANT"]))
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
es = {torch.float32: 4, torch.float16: 2, torch.bfloat16: 2}
SETTINGS = [("default", {})]
if "--ab" in sys.argv:
    SETTINGS = [("default", {}), ("inorder 64-row tiles", {"WM_ROWS_SMALL_TILE": "0"}), ("persistent", {"WM_ROWS_INORDER": "0"}),
                ("nt", {"WM_ROWS_CONVERT_NT": "1"})]
if os.environ.get("CAST_SWEEP_SETTINGS"):
    SETTINGS = [x for x in SETTINGS if x[0] in os.environ["CAST_SWEEP_SETTINGS"].split(",")]
KEYS = sorted({k for _, e in SETTINGS for k in e})


def apply(env):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    __import__("wholegraph_amd.binding").binding.reload_knobs()   # knobs are read once


for tdt, odt, dim in [(torch.float16, torch.float32, 128), (torch.float16, torch.float32, 256), (torch.float32, torch.float16, 128),
                      (torch.bfloat16, torch.float32, 128), (torch.float16, torch.float16, 256), (torch.float16, torch.float32, 100),
                      (torch.float16, torch.float32, 512), (torch.float32, torch.float32, 24), (torch.float16, torch.float16, 100)]:
    rows = int(8e9 // (dim * es[tdt]))
    n = 10_000_000
    emb = wgth.create_embedding(comm, "chunked", "cuda", tdt, [rows, dim])
    t = emb.get_embedding_tensor()
    idx = torch.randint(0, rows, (n,), device="cuda")
    out = torch.empty((n, dim), dtype=odt, device="cuda")
    for op in ("gather", "scatter"):
        fn = (lambda: emb.gather(idx, force_dtype=odt, out=out)) if op == "gather" else (lambda: t.scatter(out, idx))
        best = {name: 1e9 for name, _ in SETTINGS}
        kern = {}
        for r in range(3):
            for name, env in SETTINGS:
                apply(env)
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()
                best[name] = min(best[name], (time.perf_counter() - t0) / 10 * 1e3)
                raw = wmb.lib().wholememory_ext_last_rows_kernel().decode()
                m = re.search(r"(rows_\w+)", raw)
                kern[name] = raw[m.start():][:70] if m else raw[:70]
        apply({})
        gb = n * (8 + dim * es[tdt] + dim * es[odt]) / 1e9
        for name, _ in SETTINGS:
            print("%-7s %s -> %s dim %d  %-22s %.3f ms  %.1f%% of 8 TB/s algorithmic  [%s]" % (
                op, str(tdt).split(".")[1], str(odt).split(".")[1], dim, name, best[name], gb / best[name] / 8.0 * 100, kern[name]))
        sys.stdout.flush()
    wgth.destroy_embedding(emb)
