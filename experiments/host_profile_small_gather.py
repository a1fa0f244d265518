import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, ctypes as C
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
from wholegraph_amd.torch.wholegraph_env import wrap_torch_tensor, get_wholegraph_env_fns, get_stream
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
emb = wgth.create_embedding(comm, "chunked", "cuda", torch.float32, [1_000_000, 128])
idx = torch.randint(0, 1_000_000, (64,), device="cuda"); out = torch.empty((64, 128), device="cuda")
def t(fn, reps=20000):
    for _ in range(200): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6
print("emb.gather            %.2f us" % t(lambda: emb.gather(idx, out=out), 5000))
print("wrap idx + out        %.2f us" % t(lambda: (wrap_torch_tensor(idx), wrap_torch_tensor(out))))
print("get_env_fns           %.2f us" % t(get_wholegraph_env_fns))
print("get_stream            %.2f us" % t(get_stream))
wi, wo = wrap_torch_tensor(idx), wrap_torch_tensor(out)
env, st = get_wholegraph_env_fns(), get_stream()
lib = wmb.lib()
print("C call only           %.2f us" % t(lambda: lib.wholememory_embedding_gather(emb.wmb_embedding, wi.handle, wo.handle, False, env, st), 5000))
print("get_embedding_tensor  %.2f us" % t(emb.get_embedding_tensor))
print("idx.dim/shape/stride  %.2f us" % t(lambda: (idx.dim(), tuple(idx.shape), tuple(idx.stride()), idx.dtype, idx.data_ptr())))
torch.cuda.synchronize()
