#!/usr/bin/env python
"""Side measurement: gather / scatter algorithmic bandwidth across row shapes (1 GPU, chunked table ~ 8 GB)."""
import os
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
    if len(sys.argv) > 1:
        cases = [(torch.float32, int(x)) for x in sys.argv[1:]]
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
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            algo = n * (8 + 2 * dim * es) / ms / 1e6
            print("%-7s %s dim %4d (%4d B rows, stride %d) n=%8d : %.3f ms  %.0f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
                op, str(dt).split(".")[1], dim, dim * es, t.stride()[0] if hasattr(t, "stride") else -1, n, ms, algo, algo / 80.0))
        wgth.destroy_embedding(emb)


if __name__ == "__main__":
    main()
