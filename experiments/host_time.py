import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
rows, dim, n = 100_000_000, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
emb = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [rows, dim])
wgth.create_wholememory_optimizer(emb, "sgd", {})
idx = torch.randint(0, rows, (n,), device="cuda")
g = torch.randn((n, dim), device="cuda")
def step():
    emb.add_gradients(idx, g); emb.need_apply = True; emb.apply_gradients(0.01)
for _ in range(5): step()
torch.cuda.synchronize()
for reps in (1, 1, 1, 3, 20):
    t0 = time.perf_counter()
    for _ in range(reps): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("n=%d reps %2d: host enqueue %.3f ms/call, total %.3f ms/call" % (n, reps, (t1 - t0) / reps * 1e3, (t2 - t0) / reps * 1e3), flush=True)
