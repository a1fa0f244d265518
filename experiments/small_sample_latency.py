#!/usr/bin/env python
"""Latency of a serving-sized C5 request: S seeds, fan-out [10, 10], 2-hop sample + feature gather of the sampled sub-graph, end to end
(synchronised) per request; synthetic graph of 2 M nodes, average degree 16, [nodes, 128] fp32 features, one GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
nodes, avg = 2_000_000, 16
gen = torch.Generator(device="cuda").manual_seed(1)
row = torch.zeros(nodes + 1, dtype=torch.int64, device="cuda")
torch.cumsum(torch.randint(0, 2 * avg + 1, (nodes,), device="cuda", generator=gen), 0, out=row[1:])
edges = int(row[-1])
wrow = wgth.create_wholememory_tensor(comm, "chunked", "cuda", [nodes + 1], torch.int64, [1])
wcol = wgth.create_wholememory_tensor(comm, "chunked", "cuda", [edges], torch.int32, [1])
wrow.get_local_tensor()[0].copy_(row)
wcol.get_local_tensor()[0].copy_(torch.randint(0, nodes, (edges,), device="cuda", dtype=torch.int32, generator=gen))
feat = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [nodes, 128])
g = wgth.GraphStructure(); g.set_csr_graph(wrow, wcol)
for S in [int(x) for x in os.environ.get("SEEDS", "1,16,256,4096").split(",")]:
    seeds = torch.randint(0, nodes, (S,), device="cuda", dtype=torch.int32)
    def deferred():
        h = g.multilayer_sample_begin(seeds, [10, 10])
        xp = feat.gather(h.padded_frontier)
        tg, ei, rp, ci = h.result()
        return xp[:tg[0].numel()]
    def reference():
        tg, ei, rp, ci = g.multilayer_sample_without_replacement(seeds, [10, 10])
        return feat.gather(tg[0])
    out = []
    for name, fn in (("deferred flow", deferred), ("reference flow", reference)):
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): fn()
        torch.cuda.synchronize()
        out.append("%s %.1f us" % (name, (time.perf_counter() - t0) / 200 * 1e6))
    print("%5d seeds x [10, 10] + feature gather: " % S + "   ".join(out), flush=True)
