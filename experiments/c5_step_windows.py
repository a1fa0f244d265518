"""(diagnostic) C5 step time by window of 25 steps over the first 600 steps of a process — C5 step (2-hop [30, 30] sample from 1024 seeds + append_unique + feature gather on the papers100M-shaped synthetic graph):
the reference flow (sampling call, host round trip, feat.gather(target_gids[0])) against the gather_features_from extension
(features fetched inside the sampling call from the device-side count), interleaved in ONE process, several rounds.
usage: c5_fused_ab.py [nodes] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
torch.cuda.set_device(0)
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 111_059_956
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
deg, dim = 29, 128
gen = torch.Generator(device="cuda").manual_seed(1)
# power-law-ish degrees with the papers100M mean, as bench.py builds them
d = (torch.rand(nodes, device="cuda", generator=gen).clamp_min(1e-6) ** -0.5)
d = (d * (deg / d.mean())).clamp(max=20000).long()
row_ptr = torch.zeros(nodes + 1, dtype=torch.int64, device="cuda")
row_ptr[1:] = torch.cumsum(d, 0)
edges = int(row_ptr[-1])
del d
wrow = wgth.create_wholememory_tensor(comm, "chunked", "cuda", [nodes + 1], torch.int64, [1])
wrow.get_local_tensor()[0].copy_(row_ptr); del row_ptr
wcol = wgth.create_wholememory_tensor(comm, "chunked", "cuda", [edges], torch.int32, [1])
lc = wcol.get_local_tensor()[0]
for s in range(0, edges, 1 << 28):
    e = min(edges, s + (1 << 28))
    lc[s:e] = torch.randint(0, nodes, (e - s,), device="cuda", generator=gen, dtype=torch.int32)
feat = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [nodes, dim])
lf = feat.get_embedding_tensor().get_local_tensor()[0]
for s in range(0, nodes, 1 << 24):
    e = min(nodes, s + (1 << 24))
    lf[s:e] = (torch.arange(s, e, device="cuda") & 0xFFFFFF).float().unsqueeze(1)
g = wgth.GraphStructure(); g.set_csr_graph(wrow, wcol)
seeds = torch.randint(0, nodes, (1024,), device="cuda", generator=gen, dtype=torch.int32)
fan = [30, 30]


import gc
def step():
    tg, ei, rp, ci = g.multilayer_sample_without_replacement(seeds, fan)
    x = feat.gather(tg[0])
    return x, tg[0]

for _ in range(3):
    step()
torch.cuda.synchronize()
mode = os.environ.get("C5_WINDOWS_MODE", "plain")
if mode == "nogc":
    gc.disable()
win = []
t_prev = time.perf_counter()
for i in range(600):
    step()
    if (i + 1) % 25 == 0:
        torch.cuda.synchronize()
        t = time.perf_counter()
        win.append((t - t_prev) / 25 * 1e3)
        t_prev = t
print("mode %s: ms per step by window of 25 steps: %s" % (mode, " ".join("%.3f" % w for w in win)), flush=True)
print("gc counts", gc.get_count(), "torch allocs", torch.cuda.memory_stats().get("num_device_alloc", 0), flush=True)
