import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
torch.cuda.set_device(0)
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
free0 = torch.cuda.mem_get_info()[0]
for i in range(40):
    emb = wgth.create_embedding(comm, ["chunked", "continuous", "distributed"][i % 3], "cuda", torch.float32, [16_000_000, 128])
    t = emb.get_embedding_tensor()
    idx = torch.randint(0, 16_000_000, (100000,), device="cuda")
    out = emb.gather(idx)
    t.scatter(out, idx)
    torch.cuda.synchronize()
    wgth.destroy_embedding(emb)
    del emb, t, out, idx
torch.cuda.empty_cache()
free1 = torch.cuda.mem_get_info()[0]
print("free before %.2f GB after %.2f GB" % (free0 / 1e9, free1 / 1e9))
assert free0 - free1 < 1e9
print("LEAK_TEST_OK")
