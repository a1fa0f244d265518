"""BASELINE.json's full single-GPU sizes, checked through size-independent properties (the oracle would need minutes and
tens of GB for these): closed-form tables (every gathered element is known from its id), scatter -> gather round trips,
skipped negative ids, and integer-valued gradients whose fp32 sums are exact in any order.

  C2  CHUNKED  100 M x 128 fp32 (51.2 GB), 10 M int64 ids            — the contract workload of bench.py
  C1  HOST     10 M x 64 fp32 (2.56 GB pinned/shared), 1 M int64 ids
  C4' DISTRIBUTED (one rank) 100 M x 128 fp32, 10 M Zipf ids with heavy duplicates, gradient apply as scatter-add
      (SGD, lr = -1, weight decay 0), and the fp16 x 256 extension at 50 M rows (25.6 GB)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MASK = 0xFFFFFF


def _fill_closed_form(local, row_start, torch):
    rows = local.shape[0]
    for s in range(0, rows, 4 << 20):
        e = min(rows, s + (4 << 20))
        r = torch.arange(row_start + s, row_start + e, device="cuda", dtype=torch.int64) & MASK
        local[s:e] = r.to(torch.float32).unsqueeze(1).to(local.dtype).to(local.device)
    torch.cuda.synchronize()


def _need_hbm(torch, gib):
    free, _ = torch.cuda.mem_get_info()
    if free < gib * (1 << 30):
        pytest.skip("needs %d GiB of free HBM" % gib)


def _zipf_ids(n, rows, seed):
    k = np.random.default_rng(seed).zipf(1.05, n).astype(np.uint64)
    return ((k * np.uint64(2654435761)) % np.uint64(rows)).astype(np.int64)


def test_c2_gather_every_element_and_scatter_round_trip(gpu_env):
    import torch
    import wholegraph_amd.torch as wgth
    _need_hbm(torch, 75)
    rows, dim, n = 100_000_000, 128, 10_000_000
    emb = wgth.create_embedding(gpu_env, "chunked", "cuda", torch.float32, [rows, dim])
    table = emb.get_embedding_tensor()
    local, start = table.get_local_tensor()
    assert start == 0 and tuple(local.shape) == (rows, dim)
    _fill_closed_form(local, 0, torch)
    idx_np = np.random.default_rng(42).integers(0, rows, n, dtype=np.int64)
    idx_np[::1000] = -1                      # skipped: their output rows must stay as they were
    idx_np[1::1000] = rows - 1               # the last row, many times
    idx_np[2::1000] = 0
    idx = torch.from_numpy(idx_np).cuda()
    out = torch.full((n, dim), -7.0, device="cuda")
    emb.gather(idx, out=out)
    torch.cuda.synchronize()
    want = torch.where(idx >= 0, (idx & MASK).to(torch.float32), torch.full((), -7.0, device="cuda"))
    assert bool((out == want.unsqueeze(1)).all()), "gather at C2 size: some element differs from the closed form"
    # gather is idempotent and does not touch the table
    out2 = torch.full((n, dim), -7.0, device="cuda")
    emb.gather(idx, out=out2)
    assert torch.equal(out, out2)
    del out2, want
    # scatter -> gather round trip on 10 M DISTINCT rows (a random arithmetic progression mod a prime-ish stride), payload
    # depends on the position so a misplaced row cannot go unnoticed
    pos = torch.arange(n, device="cuda", dtype=torch.int64)
    uniq = (pos * 7 + 12345) % rows          # 7 * n < rows: all distinct
    uniq[5::997] = -1                        # skipped on both sides
    payload = (pos.to(torch.float32) * 0.5).unsqueeze(1) + torch.arange(dim, device="cuda", dtype=torch.float32)
    table.scatter(payload, uniq)
    back = torch.zeros((n, dim), device="cuda")
    emb.gather(uniq, out=back)
    torch.cuda.synchronize()
    valid = (uniq >= 0).unsqueeze(1)
    assert bool((torch.where(valid, back, payload) == payload).all()), "scatter -> gather round trip lost or moved a row"
    assert bool((torch.where(valid, torch.zeros((), device="cuda"), back) == 0).all()), "a skipped id produced output"
    # rows that no id addressed are untouched: 10 M probes of the complement (ids not congruent to 12345 mod 7)
    probe = torch.from_numpy(np.random.default_rng(1).integers(0, rows, n, dtype=np.int64)).cuda()
    probe = probe[(probe - 12345) % 7 != 0]
    got = torch.empty((probe.numel(), dim), device="cuda")
    emb.gather(probe, out=got)
    assert bool((got == (probe & MASK).to(torch.float32).unsqueeze(1)).all()), "scatter touched a row nobody addressed"
    wgth.destroy_embedding(emb)


def test_c1_host_table_gather(gpu_env):
    import torch
    import wholegraph_amd.torch as wgth
    rows, dim, n = 10_000_000, 64, 1_000_000
    emb = wgth.create_embedding(gpu_env, "chunked", "cpu", torch.float32, [rows, dim])
    local, start = emb.get_embedding_tensor().get_local_tensor(host_view=True)
    assert start == 0 and tuple(local.shape) == (rows, dim) and not local.is_cuda
    r = (torch.arange(rows, dtype=torch.int64) & MASK).to(torch.float32)
    local.copy_(r.unsqueeze(1).expand(rows, dim))
    idx_np = np.random.default_rng(42).integers(0, rows, n, dtype=np.int64)
    idx_np[::777] = -1
    idx = torch.from_numpy(idx_np).cuda()
    out = torch.full((n, dim), 3.0, device="cuda")
    emb.gather(idx, out=out)
    torch.cuda.synchronize()
    want = torch.where(idx >= 0, (idx & MASK).to(torch.float32), torch.full((), 3.0, device="cuda"))
    assert bool((out == want.unsqueeze(1)).all())
    # fp16 output of the fp32 host table: cast at gather time, values < 2048 survive exactly
    small = torch.from_numpy(np.random.default_rng(2).integers(0, 2048, n, dtype=np.int32)).cuda()
    half = emb.gather(small, force_dtype=torch.float16)
    assert half.dtype == torch.float16 and bool((half == small.to(torch.float16).unsqueeze(1)).all())
    wgth.destroy_embedding(emb)


@pytest.mark.parametrize("dtype_name,rows,dim,memory_type", [("float32", 100_000_000, 128, "distributed"),
                                                             ("float16", 50_000_000, 256, "distributed"),
                                                             ("float16", 50_000_000, 256, "continuous"),   # C4 as BASELINE names it
                                                             ("float32", 100_000_000, 128, "continuous")])
def test_c4_scatter_add_with_heavy_duplicates(gpu_env, dtype_name, rows, dim, memory_type):
    """Gradient apply as scatter-add (SGD, lr = -1, wd = 0) of 10 M Zipf(1.05) ids — about half of them duplicates, the
    hottest row hit ~10^5 times. Gradients are small integers, so every partial sum is an exactly representable integer
    whatever the summation order: table'[r] = table[r] + sum of the gradient rows addressed to r, bit for bit, checked
    against torch.index_add_ in fp64 for every touched row and a probe of untouched ones."""
    import torch
    import wholegraph_amd.torch as wgth
    dt = getattr(torch, dtype_name)
    _need_hbm(torch, 80)
    n = 10_000_000
    emb = wgth.create_embedding(gpu_env, memory_type, "cuda", dt, [rows, dim])
    local, start = emb.get_embedding_tensor().get_local_tensor()
    local.zero_()
    opt = wgth.create_wholememory_optimizer(emb, "sgd", {"weight_decay": 0.0})
    idx_np = _zipf_ids(n, rows, 42)
    idx = torch.from_numpy(idx_np).cuda()
    # g[i, c] in {-1, 0, 1} (fp16 tables: the hottest row's sum stays well inside fp16's exact-integer range only if it
    # cancels, so use a per-column sign pattern that sums to small integers: +1/-1 alternating by occurrence parity)
    sign = torch.where((torch.arange(n, device="cuda") & 1) == 0, 1.0, -1.0)
    cols = ((torch.arange(dim, device="cuda") % 3) - 1).to(torch.float32)            # -1, 0, 1, -1, ...
    grads = (sign.unsqueeze(1) * cols.unsqueeze(0)).to(dt)
    emb.add_gradients(idx, grads)
    emb.need_apply = True
    emb.apply_gradients(-1.0)
    torch.cuda.synchronize()
    # expectation per distinct id: column pattern times (#even-position hits - #odd-position hits)
    uniq, inv = torch.unique(idx, return_inverse=True)
    net = torch.zeros(uniq.numel(), dtype=torch.float64, device="cuda").index_add_(0, inv, sign.to(torch.float64))
    # partial sums in fp32 are exact; for fp16 tables the single final rounding is exact while |net| <= 2048, so the few
    # hottest rows whose net count may exceed that are compared after the same rounding
    step = 1 << 20
    for s in range(0, uniq.numel(), step):
        rows_got = local[uniq[s:s + step]].to(torch.float64)
        rows_want = (net[s:s + step].unsqueeze(1) * cols.to(torch.float64).unsqueeze(0)).to(dt).to(torch.float64)
        assert bool((rows_got == rows_want).all()), "scatter-add result differs from the exact integer sum"
    touched = torch.zeros(rows, dtype=torch.bool, device="cuda")
    touched[uniq] = True
    probe = torch.from_numpy(np.random.default_rng(3).integers(0, rows, 2_000_000, dtype=np.int64)).cuda()
    probe = probe[~touched[probe]]
    assert bool((local[probe] == 0).all()), "gradient apply touched a row no id addressed"
    assert uniq.numel() < 0.6 * n            # the batch really is duplicate-heavy
    wgth.destroy_embedding(emb)
