import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
import wholegraph_amd.torch as wgth
from wholegraph_amd import binding as wmb
from wholegraph_amd.torch.wholegraph_env import wrap_torch_tensor, get_wholegraph_env_fns, get_stream
torch.cuda.set_device(0)
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_WARN))
comm = wgth.create_group_communicator(1)
# dirty the HBM so freshly allocated blocks hold garbage, not zeros
junk = [torch.full((1<<30,), 0x7f7f7f7f, dtype=torch.int32, device='cuda') for _ in range(40)]
torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()
n_rows, dim, n_idx = 100003, 32, 100000
for trial in range(6):
    root = wgth.create_wholememory_tensor(comm, "continuous", "cuda", [n_rows, dim], torch.float32, [dim, 1])
    full = oracle.fill_closed_form(np.float32, 0, n_rows, dim, dim)
    local, start = root.get_local_tensor()
    local.copy_(torch.from_numpy(full).cuda())
    torch.cuda.synchronize()
    tab_back = local.cpu().numpy()
    print("trial", trial, "table ok:", np.array_equal(tab_back, full), "ptr", hex(local.data_ptr()))
    rng = np.random.default_rng(1234 + n_rows + dim + n_idx)
    idx = rng.integers(0, n_rows, n_idx).astype(np.int64)
    idx[rng.integers(0, n_idx, n_idx // 50)] = -1
    idx[:64] = idx[0]
    out_np = rng.integers(-3, 3, (n_idx, dim)).astype(np.float32)
    out_t = torch.from_numpy(out_np.copy()).cuda()
    wi, wo = wrap_torch_tensor(torch.from_numpy(idx).cuda()), wrap_torch_tensor(out_t)
    wmb.check(wmb.lib().wholememory_gather(root.wmb_tensor, wi.handle, wo.handle, get_wholegraph_env_fns(), C.c_void_p(get_stream()), -1))
    torch.cuda.synchronize()
    exp = out_np.copy()
    tab = oracle.ShardedTable.from_full(full, 1)
    oracle.gather(tab, idx, exp)
    got = out_t.cpu().numpy()
    bad_rows = np.where((got != exp).any(axis=1))[0]
    print("  mismatching rows:", len(bad_rows), bad_rows[:20], "idx there:", idx[bad_rows[:10]])
    if len(bad_rows):
        r = bad_rows[0]
        print("   got", got[r][:8], "exp", exp[r][:8], "prefill", out_np[r][:8])
        badcols = np.where(got[r] != exp[r])[0]
        print("   bad cols", badcols)
    wgth.destroy_wholememory_tensor(root)
