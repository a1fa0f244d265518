"""Sorted-ids gather of HOST-located tables (reference gather_op.cpp:116-120 + functions/sort_indices_func.cu:41-91): rows of at
most 512 bytes are fetched in ascending row order, every id carrying its output row as the row map. Results must not depend
on the route: exact copies (and the same round-to-nearest casts) of the table rows, rows of negative ids left untouched."""
import numpy as np
import pytest


def _torch():
    import torch
    return torch


@pytest.mark.gpu
@pytest.mark.parametrize("mt", ["chunked", "continuous"])
@pytest.mark.parametrize("dim,tdt,odt,taken", [
    (64, "float32", "float32", True),      # C1's shape: 256-byte rows
    (128, "float32", "float32", True),     # 512 bytes: the last size the reference sorts for
    (129, "float32", "float32", False),    # 516 bytes: past the limit
    (100, "float16", "float32", True),     # ragged 200-byte rows with a cast
    (7, "int64", "int64", True),
])
@pytest.mark.parametrize("idt", ["int32", "int64"])
def test_host_gather_takes_the_sorted_route_and_matches(gpu_env, wm_lib, knobs, mt, dim, tdt, odt, taken, idt):
    torch = _torch()
    knobs.set("WM_HOST_SORTED_MIN", "16384")   # (the default, 2^19 ids, is where the route starts to pay)
    import ctypes as C
    import wholegraph_amd.torch as wgth
    from wholegraph_amd import binding as wmb
    from wholegraph_amd.torch.wholegraph_env import wrap_torch_tensor, get_wholegraph_env_fns, get_stream
    rows, n = 300_007, 50_021
    tt, ot, it = getattr(torch, tdt), getattr(torch, odt), getattr(torch, idt)
    wm = wgth.create_wholememory_tensor(gpu_env, mt, "cpu", [rows, dim], tt, [dim, 1])
    local, _ = wm.get_local_tensor(host_view=True)
    rng = np.random.default_rng(5)
    if tt.is_floating_point:
        table = torch.from_numpy(rng.standard_normal((rows, dim)).astype(np.float32) * 100).to(tt)
    else:
        table = torch.from_numpy(rng.integers(-2 ** 40, 2 ** 40, (rows, dim)))
    local.copy_(table)
    idx_np = rng.integers(0, rows, n)
    idx_np[::11] = -1 - (idx_np[::11] % 5)          # "skip me" ids of several values
    idx_np[1000:2000] = idx_np[0]                   # a run of duplicates
    idx = torch.from_numpy(idx_np).to(it).cuda()
    out = torch.full((n, dim), 7, dtype=ot, device="cuda")
    before = wm_lib.wholememory_ext_host_sorted_gathers()
    wi, wo = wrap_torch_tensor(idx), wrap_torch_tensor(out)      # a prefilled output: rows of negative ids must stay as they are
    wmb.check(wm_lib.wholememory_gather(wm.wmb_tensor, wi.handle, wo.handle, get_wholegraph_env_fns(), C.c_void_p(get_stream()), -1))
    torch.cuda.synchronize()
    took = wm_lib.wholememory_ext_host_sorted_gathers() - before
    assert took == (1 if taken else 0)
    want = torch.full((n, dim), 7, dtype=ot)
    ok = torch.from_numpy(idx_np >= 0)
    want[ok] = table[torch.from_numpy(idx_np[idx_np >= 0])].to(ot)
    assert out.cpu().view(torch.uint8).numpy().tobytes() == want.view(torch.uint8).numpy().tobytes()
    wgth.destroy_wholememory_tensor(wm)


@pytest.mark.gpu
def test_default_rule_small_batches_and_device_tables_keep_the_plain_route(gpu_env, wm_lib, knobs):
    torch = _torch()
    knobs.unset("WM_HOST_SORTED_MIN")
    import wholegraph_amd.torch as wgth
    rows, dim = 50_000, 64      # (float32 holds every row number exactly)
    for loc, n, taken in (("cpu", 100_000, 0), ("cuda", 600_000, 0), ("cpu", 600_000, 1)):
        wm = wgth.create_wholememory_tensor(gpu_env, "chunked", loc, [rows, dim], torch.float32, [dim, 1])
        local, _ = wm.get_local_tensor(host_view=(loc == "cpu"))
        local.copy_(torch.arange(rows, dtype=torch.float32).unsqueeze(1).expand(rows, dim))
        idx = torch.randint(0, rows, (n,), device="cuda")
        before = wm_lib.wholememory_ext_host_sorted_gathers()
        got = wm.gather(idx)
        torch.cuda.synchronize()
        assert wm_lib.wholememory_ext_host_sorted_gathers() - before == taken, (loc, n)
        assert torch.equal(got[:, 0].cpu(), idx.cpu().float()) and torch.equal(got[:, -1].cpu(), idx.cpu().float())
        wgth.destroy_wholememory_tensor(wm)
