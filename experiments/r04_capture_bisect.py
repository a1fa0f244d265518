"""which part of the C5 step survives a hipGraph capture + replay? (run each mode in its own process)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
from test_graph_ops_gpu import make_csr
mode = sys.argv[1]
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
n_nodes, dim, fanouts, hop_seeds = 20011, 32, [30, 30], [5, 6]
row_ptr, col = make_csr(n_nodes, 40, 11, np.int32, heavy=[(3, 3000), (4, 0)])
def wm_array(arr):
    t = wgth.create_wholememory_tensor(comm, "chunked", "cuda", [arr.shape[0]], torch.from_numpy(arr[:1]).dtype, [1])
    l, _ = t.get_local_tensor(); l.copy_(torch.from_numpy(arr).cuda()); return t
wrow, wcol = wm_array(row_ptr), wm_array(col)
g = wgth.GraphStructure(); g.set_csr_graph(wrow, wcol)
emb = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [n_nodes, dim])
local, _ = emb.get_embedding_tensor().get_local_tensor()
local.copy_(torch.arange(n_nodes, device="cuda", dtype=torch.float32).unsqueeze(1).expand(n_nodes, dim))
seeds = torch.from_numpy(np.concatenate([[3, 4], np.random.default_rng(5).permutation(n_nodes)[:200]]).astype(np.int32)).cuda()
room = seeds.shape[0] * 31 * 31
out = torch.empty((room, dim), device="cuda")
if mode.startswith("test"):
    ref = g.multilayer_sample_without_replacement(seeds, fanouts, random_seeds=hop_seeds)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    h0 = g.multilayer_sample_begin(seeds, fanouts, random_seeds=hop_seeds)
    emb.gather(h0.padded_frontier, out=out); h0.result()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    if mode in ("c3", "c3_old", "chain", "both", "test", "test_nofill", "test3", "test3_out", "test3_pad", "test3_neg", "test3_sleep", "test3_none"):
        h = g.multilayer_sample_begin(seeds, fanouts, random_seeds=hop_seeds)
    if mode in ("gather", "g3"):
        emb.gather(h0.padded_frontier, out=out)
    if mode in ("both", "test", "test_nofill", "test3", "test3_out", "test3_pad", "test3_neg", "test3_sleep", "test3_none"):
        emb.gather(h.padded_frontier, out=out)
    if mode == "onehop":
        h = g.multilayer_sample_begin(seeds, [30], random_seeds=[5])
print(mode, "captured", flush=True)
if mode in ("g3", "c3", "c3_old"):
    for it in range(3):
        torch.cuda._sleep(1000000)
        graph.replay(); torch.cuda.synchronize()
        print("iteration", it, "ok", flush=True)
    sys.exit(0)
if mode == "test":
    out.fill_(-7.0); h.padded_frontier.fill_(123)
    print("filled", flush=True)
if mode.startswith("test3"):
    for it in range(3):
        if mode in ("test3", "test3_out"): out.fill_(-7.0)
        if mode in ("test3", "test3_pad"): h.padded_frontier.fill_(123)
        if mode == "test3_neg": h.padded_frontier.fill_(-1)
        if mode == "test3_sleep": torch.cuda._sleep(10000000)
        graph.replay(); torch.cuda.synchronize()
        print("iteration", it, "ok", flush=True)
graph.replay(); torch.cuda.synchronize()
print(mode, "replayed", flush=True)
graph.replay(); torch.cuda.synchronize()
print(mode, "replayed twice OK", flush=True)

if mode.startswith("test"):
    got = h.result()
    print("result ok", [tuple(t.shape) for t in got[0]], torch.equal(got[0][0], ref[0][0]), flush=True)
