#!/usr/bin/env python
"""Side measurement: gather / scatter algorithmic bandwidth across row shapes (1 GPU, chunked table ~ 8 GB).
  python experiments/dim_sweep.py [--csv=file] [--ab] [dims ...]
--ab: every (shape, op) is measured under several kernel / launch-shape switches (WM_ROWS_INORDER, WM_ROWS_BLOCK, WM_ROWS_FLAT,
WM_ROWS_STAGED; DIM_SWEEP_SETTINGS=default,inorder=0 keeps a subset) as well as the default rule, interleaved over 5 rounds of 10
launches, and the MIN over rounds is reported (a shared box drifts by 10-20 % between back-to-back runs of one setting)."""
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    import wholegraph_amd.torch as wgth
    from wholegraph_amd import binding as wmb
    wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
    comm = wgth.create_group_communicator(1)
    cases = [(torch.float32, d) for d in (8, 16, 32, 64, 100, 128, 200, 256, 300, 512, 602, 1024)] + \
            [(torch.float16, d) for d in (64, 128, 256, 768, 1024)]
    csv_out, ab = None, False
    args = []
    for x in sys.argv[1:]:
        if x.startswith("--csv="):
            csv_out = open(x[6:], "w")
            csv_out.write("op,dtype,dim,row_bytes,stride_elems,n_ids,setting,kernel,ms_min,ms_median,algorithmic_GB,frac_of_8TBps\n")
        elif x == "--ab":
            ab = True
        else:
            args.append(x)
    if args:
        cases = [(torch.float32, int(x)) for x in args]
    knobs = ("WM_ROWS_FLAT", "WM_ROWS_TILE", "WM_ROWS_STAGED", "WM_ROWS_INORDER", "WM_ROWS_BLOCK", "WM_ROWS_BATCH", "WM_ROWS_PIECES", "WM_ROWS_LDS", "WM_ROWS_STAGED_SCATTER", "WM_ROWS_STAGED_MAXROW", "WM_ROWS_STAGED_ALIGNED", "WM_ROWS_STAGED_MINROW")
    settings = [("default", {}), ("inorder=0", {"WM_ROWS_INORDER": "0"}), ("block=64", {"WM_ROWS_BLOCK": "64"}), ("batch=0", {"WM_ROWS_BATCH": "0"}), ("pieces=1", {"WM_ROWS_PIECES": "1"}), ("lds=6.6k", {"WM_ROWS_LDS": "6800"}), ("lds=10k", {"WM_ROWS_LDS": "10240"}), ("lds=20k", {"WM_ROWS_LDS": "20480"}), ("pieces+nostage", {"WM_ROWS_PIECES": "1", "WM_ROWS_STAGED": "0"}), ("inorder=1", {"WM_ROWS_INORDER": "1"}),
                ("flat=0", {"WM_ROWS_FLAT": "0"}), ("flat=1", {"WM_ROWS_FLAT": "1"}),
                ("staged=0", {"WM_ROWS_STAGED": "0"}), ("sscatter=0", {"WM_ROWS_STAGED_SCATTER": "0"}),
                ("maxrow=5120", {"WM_ROWS_STAGED_MAXROW": "5120"}), ("maxrow=1024", {"WM_ROWS_STAGED_MAXROW": "1024"}), ("aligned=1", {"WM_ROWS_STAGED_ALIGNED": "1"}), ("minrow=16", {"WM_ROWS_STAGED_MINROW": "16"}), ("minrow=1M", {"WM_ROWS_STAGED_MINROW": "1000000"}), ("maxrow=5120+inorder=0", {"WM_ROWS_STAGED_MAXROW": "5120", "WM_ROWS_INORDER": "0"})] if ab else [("default", {})]
    if ab and os.environ.get("DIM_SWEEP_SETTINGS"):   # e.g. "default,inorder=0"
        keep = os.environ["DIM_SWEEP_SETTINGS"].split(",")
        settings = [x for x in settings if x[0] in keep]
    rounds = 5 if ab else 1
    for dt, dim in cases:
        es = 4 if dt == torch.float32 else 2
        rows = int(8e9 // (dim * es))
        n = int(min(10_000_000, 4e9 // (dim * es)))
        emb = wgth.create_embedding(comm, "chunked", "cuda", dt, [rows, dim])
        t = emb.get_embedding_tensor()
        idx = torch.randint(0, rows, (n,), device="cuda")
        out = torch.empty((n, dim), dtype=dt, device="cuda")
        for op in ("gather", "scatter"):
            fn = (lambda: emb.gather(idx, out=out)) if op == "gather" else (lambda: t.scatter(out, idx))
            times = {s: [] for s, _ in settings}
            kernels = {}
            for r in range(rounds):
                for name, val in settings:
                    for k in knobs:
                        os.environ.pop(k, None)
                    os.environ.update(val)
                    __import__("wholegraph_amd.binding").binding.reload_knobs()   # knobs are read once
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(10):
                        fn()
                    torch.cuda.synchronize()
                    times[name].append((time.perf_counter() - t0) / 10 * 1e3)
                    kernels[name] = re.search(r"(rows_\w+<[^(]*>)\(", wmb.lib().wholememory_ext_last_rows_kernel().decode()).group(1)
            for k in knobs:
                os.environ.pop(k, None)
            wmb.reload_knobs()
            gb = n * (8 + 2 * dim * es) / 1e9
            for name, _ in settings:
                ts = sorted(times[name])
                mn, md = ts[0], ts[len(ts) // 2]
                if csv_out:
                    csv_out.write("%s,%s,%d,%d,%d,%d,%s,\"%s\",%.4f,%.4f,%.4f,%.4f\n" % (
                        op, str(dt).split(".")[1], dim, dim * es, t.stride()[0], n, name, kernels[name], mn, md, gb, gb / mn / 8.0))
                    csv_out.flush()
                # rows under 128 B: a random access moves whole 64-byte sectors of the table side whatever the row holds, so the
                # algorithmic bytes are the wrong yardstick there — the sector-level roofline counts id + ceil(row / 64) x 64 bytes
                # on the random (table) side + the row on the dense side
                row_b, stride_b = dim * es, t.stride()[0] * es
                sect = (row_b + 63) // 64 * 64 if stride_b % 64 == 0 else row_b + 48   # rows start at 16-byte steps: (row + 48) / 64 sectors on average
                gb_sector = n * (8 + sect + row_b) / 1e9
                extra = "" if row_b >= 128 else "  | sector-level: %.1f%% of 8 TB/s (%d B per lookup)" % (gb_sector / mn / 8.0 * 100, 8 + sect + row_b)
                print("%-7s %s dim %4d (%4d B rows, stride %d) n=%8d %-8s: min %.3f ms median %.3f  %.1f%% of 8 TB/s algorithmic  [%s]%s" % (
                    op, str(dt).split(".")[1], dim, dim * es, t.stride()[0], n, name, mn, md, gb / mn / 8.0 * 100, kernels[name], extra))
        wgth.destroy_embedding(emb)


if __name__ == "__main__":
    main()
