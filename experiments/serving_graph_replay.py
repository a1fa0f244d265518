#!/usr/bin/env python
"""A serving-sized C5 request (S seeds, fan-out [10, 10], 2-hop sample + feature gather) queued call by call against the same step
captured ONCE into a hipGraph (GraphStructure.multilayer_sample_begin + the gather on the padded frontier enqueue kernels only) and
replayed per request: the seeds are copied into the captured buffer, the counts read from pinned memory after the replay."""
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
fan, hop_seeds = [10, 10], [5, 6]
for S in (1, 64, 1024):
    seeds = torch.randint(0, nodes, (S,), device="cuda", dtype=torch.int32)
    room = S * 11 * 11
    out = torch.empty((room, 128), device="cuda")
    def eager():
        h = g.multilayer_sample_begin(seeds, fan, random_seeds=hop_seeds)
        feat.gather(h.padded_frontier, out=out)
        return h.result()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ref = eager()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        h = g.multilayer_sample_begin(seeds, fan, random_seeds=hop_seeds)
        feat.gather(h.padded_frontier, out=out)
    new_seeds = torch.randint(0, nodes, (S,), device="cuda", dtype=torch.int32)
    def replay():
        seeds.copy_(new_seeds, non_blocking=True)      # the request's seeds into the captured buffer
        graph.replay()
        torch.cuda.synchronize()                        # counts are in pinned memory now; outputs are views of the captured buffers
    def timed(fn, reps=300):
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6
    te = timed(lambda: (seeds.copy_(new_seeds, non_blocking=True), eager()))
    tr = timed(replay)
    got = h.result()
    n = got[0][0].shape[0]
    ok = bool(torch.equal(out[:n, 0], feat.gather(got[0][0])[:, 0]))
    print("%5d seeds x [10, 10] + gather: call by call %.1f us per request, replayed from a hipGraph %.1f us (frontier %d nodes, rows correct: %s)" % (S, te, tr, n, ok), flush=True)
