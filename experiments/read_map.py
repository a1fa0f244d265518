#!/usr/bin/env python
"""Read-bandwidth map of a fresh 51 GB device allocation: GB/s of a read-only pass (torch sum) over every 1 GiB piece.
Looks for physical regions that read slower than others (the C2 gather is bimodal between allocations, alloc_variance.py)."""
import sys, time
import torch
for rnd in range(3):
    buf = torch.empty(51 * (1 << 30), dtype=torch.uint8, device="cuda")
    v = buf.view(torch.float32)
    v[:1024].zero_()
    per = (1 << 30) // 4
    rates = []
    for c in range(51):
        x = v[c * per:(c + 1) * per]
        x.zero_()
        torch.cuda.synchronize()
        for _ in range(2):
            x.sum()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            x.sum()
        torch.cuda.synchronize()
        rates.append((1 << 30) / ((time.perf_counter() - t0) / 10) / 1e9)
    print("round %d base 0x%x: GB/s per GiB piece: %s" % (rnd, buf.data_ptr(), " ".join("%.0f" % r for r in rates)), flush=True)
    print("   min %.0f max %.0f mean %.0f" % (min(rates), max(rates), sum(rates) / len(rates)), flush=True)
    if rnd == 0:
        keep = torch.empty(7 * (1 << 30), dtype=torch.uint8, device="cuda")
    del buf, v, x
    torch.cuda.empty_cache()
