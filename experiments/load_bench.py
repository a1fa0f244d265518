#!/usr/bin/env python
"""Side measurement: WholeMemoryEmbedding.save / .load throughput (page-cache warm files in /tmp)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
rows, dim = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000, 128
emb = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
local, _ = emb.get_embedding_tensor().get_local_tensor()
local.normal_()
torch.cuda.synchronize()
ref = local[::100003].clone()
gb = rows * dim * 4 / 1e9
t0 = time.perf_counter(); emb.save("/tmp/wm_load_bench"); t1 = time.perf_counter()
print("save %.2f GB: %.2f s (%.2f GB/s)" % (gb, t1 - t0, gb / (t1 - t0)))
for th in (1, 4, 8, 16):
    os.environ["WG_LOAD_THREADS_PER_RANK"] = str(th)
    local.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter(); emb.load("/tmp/wm_load_bench"); torch.cuda.synchronize(); t1 = time.perf_counter()
    assert torch.equal(local[::100003], ref)
    print("load %.2f GB with %2d threads: %.2f s (%.2f GB/s)" % (gb, th, t1 - t0, gb / (t1 - t0)))
for f in os.listdir("/tmp"):
    if f.startswith("wm_load_bench"):
        os.remove(os.path.join("/tmp", f))
