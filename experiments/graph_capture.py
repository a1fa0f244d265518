import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, time
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
torch.cuda.set_device(0)
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
rows, dim = 2_000_000, 128
emb = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
local, _ = emb.get_embedding_tensor().get_local_tensor()
local.copy_((torch.arange(rows, device="cuda") & 0xFFFFFF).float().unsqueeze(1).expand(rows, dim))
n = 4096
idxs = [torch.randint(0, rows, (n,), device="cuda") for _ in range(16)]
outs = [torch.zeros((n, dim), device="cuda") for _ in range(16)]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(16): emb.gather(idxs[i], out=outs[i])
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for i in range(16): emb.gather(idxs[i], out=outs[i])
for o in outs: o.zero_()
g.replay(); torch.cuda.synchronize()
ok = all(bool((outs[i] == (idxs[i] & 0xFFFFFF).float().unsqueeze(1)).all()) for i in range(16))
print("graph replay correct:", ok)
def t(fn, reps=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
def eager():
    for i in range(16): emb.gather(idxs[i], out=outs[i])
print("16 small gathers: eager %.1f us, hipGraph replay %.1f us" % (t(eager), t(g.replay)))
