"""GPU parity of the distributed-path stages through the C ABI, against the CPU oracle:

 * wholememory_ext_bucket_ids  — counts bit-exact vs bucket_ids_func.cu:51-87 restatement; grouped ids /
   raw_indices: a stable owner-partition whose segments, stably sorted by id, reproduce the reference's
   full stable sort (exchange_ids_nccl_func.cu:42-92) bit for bit;
 * wholememory_ext_dedup_apply — sorted-order duplicate sum + SGD / LazyAdam(W) / AdaGrad / RMSProp step,
   bit-exact vs oracle (wm_oracle.c: dedup + *_step), which itself is pinned to the reference tests'
   CPUOptimizer within 1e-5 (tests/test_oracle_pinning.py);
 * WholeMemoryEmbedding end to end on one GPU (DISTRIBUTED runs the full exchange path with world_size 1),
   including the reference test shapes 400001 x {127,129} (wholememory_embedding_gradient_apply_tests.cu:30-39).
"""
import ctypes as C

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _env():
    from wholegraph_amd.torch.wholegraph_env import get_wholegraph_env_fns, get_stream
    return get_wholegraph_env_fns(), C.c_void_p(get_stream())


@pytest.mark.parametrize("idt", [np.int32, np.int64])
@pytest.mark.parametrize("n,world,kind", [(0, 4, "equal"), (1, 1, "equal"), (63, 2, "equal"), (4097, 8, "equal"),
                                          (100000, 8, "equal"), (250000, 3, "custom"), (99999, 8, "empty_ranks"),
                                          (300000, 64, "equal"), (1000003, 8, "zipf"),
                                          # 16 ranks = the last world size of the one-ballot-per-owner kernels, 17 the first of the peel loop
                                          (200001, 16, "equal"), (200001, 15, "zipf"), (200001, 17, "equal")])
def test_bucket_ids(gpu_env, idt, n, world, kind):
    import torch
    from wholegraph_amd import binding as wmb
    rng = np.random.default_rng(n + world)
    total = 1_000_003
    if kind == "custom":
        offs, _ = oracle.custom_partition([400000, 3, 600000])
    elif kind == "empty_ranks":
        offs = np.array([0, 10, 10, 10, 500000, 500000, 900000, total, total], dtype=np.uint64)
    else:
        _, offs = oracle.equal_partition(total, world)
    if kind == "zipf":
        idx = (rng.zipf(1.05, n).astype(np.uint64) * np.uint64(2654435761) % np.uint64(total)).astype(idt)
    else:
        idx = rng.integers(0, total, n).astype(idt)
    if n > 10:
        idx[rng.integers(0, n, n // 20)] = -1
        idx[rng.integers(0, n, 5)] = -(2 ** 20)
        idx[:100] = idx[50]
    d_idx = torch.from_numpy(idx).cuda()
    d_off = torch.from_numpy(offs.astype(np.int64)).cuda()
    d_cnt = torch.full((world,), -1, dtype=torch.int64, device="cuda")
    d_ids = torch.zeros(max(n, 1), dtype=d_idx.dtype, device="cuda")
    d_raw = torch.zeros(max(n, 1), dtype=torch.int64, device="cuda")
    env, stream = _env()
    wmb.check(wmb.lib().wholememory_ext_bucket_ids(d_idx.data_ptr(), wmb.DT_INT if idt == np.int32 else wmb.DT_INT64, n,
                                                   d_off.data_ptr(), world, d_cnt.data_ptr(), d_ids.data_ptr(),
                                                   d_raw.data_ptr(), env, stream))
    torch.cuda.synchronize()
    counts = d_cnt.cpu().numpy()
    assert np.array_equal(counts, oracle.bucket_counts(idx, offs)), "per-owner counts differ from the oracle"
    ids, raw = d_ids.cpu().numpy()[:n], d_raw.cpu().numpy()[:n]
    assert sorted(raw.tolist()) == list(range(n)), "raw_indices is not a permutation"
    assert np.array_equal(ids, idx[raw]), "bucketed ids are not idx[raw_indices]"
    # segments: owner-major, stable
    seg = np.concatenate([[0], np.cumsum(counts), [n]])
    for r in range(world + 1):
        s, e = int(seg[r]), int(seg[r + 1])
        assert np.all(np.diff(raw[s:e]) > 0), "segment %d is not in original order (not stable)" % r
        if r < world and e > s:
            assert ids[s:e].min() >= int(offs[r]) and ids[s:e].max() < int(offs[r + 1])
        if r == world:
            assert np.all(ids[s:e] < 0)
    # canonicalise to the reference order: stable sort by id inside each valid segment
    ref_sorted, ref_raw = oracle.sort_ids(idx)
    canon_ids, canon_raw = ids.copy(), raw.copy()
    nvalid = int(seg[world])
    o = np.argsort(ids[:nvalid].astype(np.int64), kind="stable")
    canon_ids[:nvalid], canon_raw[:nvalid] = ids[:nvalid][o], raw[:nvalid][o]
    assert np.array_equal(canon_ids[:nvalid], ref_sorted[:nvalid])
    assert np.array_equal(canon_raw[:nvalid], ref_raw[:nvalid])
    # counts-only call
    d_cnt2 = torch.full((world,), -1, dtype=torch.int64, device="cuda")
    wmb.check(wmb.lib().wholememory_ext_bucket_ids(d_idx.data_ptr(), wmb.DT_INT if idt == np.int32 else wmb.DT_INT64, n,
                                                   d_off.data_ptr(), world, d_cnt2.data_ptr(), None, None, env, stream))
    torch.cuda.synchronize()
    assert np.array_equal(d_cnt2.cpu().numpy(), counts)


OPTS = [("sgd", 1, {}), ("sgd", 1, {"weight_decay": 0.05}), ("adam", 2, {}), ("adam", 2, {"weight_decay": 0.01}),
        ("adam", 2, {"weight_decay": 0.02, "adam_w": 1.0}), ("rmsprop", 3, {"alpha": 0.9, "weight_decay": 0.01}),
        ("adagrad", 4, {"weight_decay": 0.01})]


@pytest.mark.parametrize("idt", [np.int32, np.int64])
@pytest.mark.parametrize("n,owners,buckets", [(0, 8, 2), (777, 4, 2), (100000, 16, 8), (250001, 24, 3), (400000, 512, 8),
                                              (50000, 1024, 1)])
def test_bucket_ids_folded(gpu_env, idt, n, owners, buckets):
    """First hop of the HIERARCHY gather: `owners` row ranges folded onto `buckets` ranks by owner % buckets — stable,
    negatives last, counts exact (numpy restatement: searchsorted over the range offsets)."""
    import torch
    from wholegraph_amd import binding as wmb
    rng = np.random.default_rng(n + owners)
    total = 1_000_003
    _, offs = oracle.equal_partition(total, owners)
    if owners == 24:   # unequal ranges with empty ones in between
        cuts = np.sort(rng.integers(0, total, owners - 1))
        cuts[5] = cuts[4]
        offs = np.concatenate([[0], cuts, [total]]).astype(np.uint64)
    idx = rng.integers(0, total, n).astype(idt)
    if n > 10:
        idx[rng.integers(0, n, n // 20)] = -1
        idx[:100] = idx[50]
    d_idx = torch.from_numpy(idx).cuda()
    d_off = torch.from_numpy(offs.astype(np.int64)).cuda()
    d_cnt = torch.full((buckets,), -1, dtype=torch.int64, device="cuda")
    d_ids = torch.zeros(max(n, 1), dtype=d_idx.dtype, device="cuda")
    d_raw = torch.zeros(max(n, 1), dtype=torch.int64, device="cuda")
    env, stream = _env()
    wmb.check(wmb.lib().wholememory_ext_bucket_ids_folded(
        d_idx.data_ptr(), wmb.DT_INT if idt == np.int32 else wmb.DT_INT64, n, d_off.data_ptr(), owners, buckets,
        d_cnt.data_ptr(), d_ids.data_ptr(), d_raw.data_ptr(), env, stream))
    torch.cuda.synchronize()
    owner = np.searchsorted(offs[1:].astype(np.int64), idx.astype(np.int64), side="right")   # the r with off[r] <= id < off[r+1]
    bucket = np.where(idx >= 0, owner % buckets, buckets)
    want_raw = np.argsort(bucket, kind="stable")
    assert np.array_equal(d_cnt.cpu().numpy(), np.bincount(bucket, minlength=buckets + 1)[:buckets])
    assert np.array_equal(d_raw.cpu().numpy()[:n], want_raw)
    assert np.array_equal(d_ids.cpu().numpy()[:n], idx[want_raw])


@pytest.mark.parametrize("kind,code,params", OPTS, ids=lambda x: str(x))
@pytest.mark.parametrize("dim,n_recv,idt", [(127, 20005, np.int64), (129, 20005, np.int32), (128, 50001, np.int64),
                                            (392, 5001, np.int64), (4, 3000, np.int64), (128, 0, np.int64), (256, 100003, np.int64),
                                            (64, 41000, np.int32)])
def test_dedup_apply_bit_exact(gpu_env, kind, code, params, dim, n_recv, idt):
    import torch
    from wholegraph_amd import binding as wmb
    if isinstance(kind, tuple):
        pytest.skip()
    rng = np.random.default_rng(dim * 7 + n_recv + code)
    local_rows, local_off = 4001, 123456
    stride = int(oracle.align_embedding_dim(dim, 4))
    table = np.zeros((local_rows, stride), np.float32)
    table[:, :dim] = rng.standard_normal((local_rows, dim)).astype(np.float32)
    ids = (local_off + rng.integers(0, local_rows, n_recv)).astype(idt)
    if n_recv:
        ids[::5] = ids[0]  # a long run of duplicates
    grads = rng.standard_normal((max(n_recv, 1), dim)).astype(np.float32)[:n_recv]
    p = dict(weight_decay=0.0, epsilon=1e-8, beta1=0.9, beta2=0.999, alpha=0.99, adam_w=0.0)
    p.update(params)
    ref_opt = oracle.Optimizer(kind, local_rows, stride, **params)
    d_table = torch.from_numpy(table.copy()).cuda()
    d_pe = d_pr = None
    if kind == "adam":
        d_pe = torch.zeros((local_rows, 2 * stride), device="cuda")
        d_pr = torch.ones((local_rows, 2), device="cuda")
    elif kind in ("adagrad", "rmsprop"):
        d_pe = torch.zeros((local_rows, stride), device="cuda")
    d_ids = torch.from_numpy(ids).cuda()
    d_grads = torch.from_numpy(grads).cuda() if n_recv else torch.zeros((1, dim), device="cuda")
    arr = (C.c_float * 6)(p["weight_decay"], p["epsilon"], p["beta1"], p["beta2"], p["alpha"], p["adam_w"])
    env, stream = _env()
    ref_table = table.copy()
    for step in range(3):
        nu = C.c_int64(-1)
        wmb.check(wmb.lib().wholememory_ext_dedup_apply(
            d_ids.data_ptr(), wmb.DT_INT if idt == np.int32 else wmb.DT_INT64, n_recv, d_grads.data_ptr(), dim, dim,
            d_table.data_ptr(), stride, local_off, local_rows, code, arr, 0.03,
            d_pe.data_ptr() if d_pe is not None else None, d_pr.data_ptr() if d_pr is not None else None, C.byref(nu),
            env, stream))
        torch.cuda.synchronize()
        uniq, dg = oracle.dedup_grads(ids, grads) if n_recv else (ids[:0], grads[:0])
        assert nu.value == len(uniq)
        ref_opt.step(uniq, dg, ref_table, stride, local_off, dim, 0.03)
        assert d_table.cpu().numpy().tobytes() == ref_table.tobytes(), "%s step %d: table differs from the oracle" % (kind, step)
    if kind != "sgd":
        assert d_pe.cpu().numpy().tobytes() == ref_opt.per_element.tobytes()
    if kind == "adam":
        assert d_pr.cpu().numpy().tobytes() == ref_opt.per_row.tobytes()


@pytest.mark.parametrize("mt", ["continuous", "chunked", "distributed"])
@pytest.mark.parametrize("kind,params", [("sgd", {"weight_decay": 0.01}), ("adam", {}), ("adagrad", {}), ("rmsprop", {})])
def test_embedding_training_flow(gpu_env, mt, kind, params):
    """WholeMemoryEmbeddingModule forward + autograd backward + WholeMemoryOptimizer.step, reference shapes
    (400001 rows x 127, 100005 ids; wholememory_embedding_gradient_apply_tests.cu:30-39), 2 steps."""
    import torch
    import wholegraph_amd.torch as wgth
    n_rows, dim, n_idx = 400001, 127, 100005
    emb = wgth.create_embedding(gpu_env, mt, "cuda", torch.float32, [n_rows, dim])
    assert emb.get_embedding_tensor().stride() == (128, 1) and emb.shape == (n_rows, dim)
    rng = np.random.default_rng(3)
    init = rng.standard_normal((n_rows, dim)).astype(np.float32)
    local, start = emb.get_embedding_tensor().get_local_tensor()
    local.copy_(torch.from_numpy(init).cuda())
    opt = wgth.create_wholememory_optimizer(emb, kind, params)
    module = wgth.WholeMemoryEmbeddingModule(emb)
    module.train()
    padded = np.zeros((n_rows, 128), np.float32)
    padded[:, :dim] = init
    tab = oracle.ShardedTable.from_full(padded, 1)
    tab.dim = dim
    ref_opt = oracle.Optimizer(kind, n_rows, 128, **params)
    for step in range(2):
        idx = rng.integers(0, n_rows, n_idx).astype(np.int64)
        idx[::3] = idx[0]
        w = rng.standard_normal((n_idx, dim)).astype(np.float32)
        out = module(torch.from_numpy(idx).cuda())
        exp_out = np.zeros((n_idx, dim), np.float32)
        oracle.gather(tab, idx, exp_out)
        assert out.detach().cpu().numpy().tobytes() == exp_out.tobytes()
        loss = (out * torch.from_numpy(w).cuda()).sum()
        loss.backward()
        opt.step(0.02)
        oracle.gradient_apply(tab, [ref_opt], [idx], [w], 0.02)
        torch.cuda.synchronize()
        assert local.cpu().numpy().tobytes() == tab.shards[0][:, :dim].tobytes(), "step %d" % step
    wgth.destroy_wholememory_optimizer(opt)
    wgth.destroy_embedding(emb)


@pytest.mark.parametrize("mt", ["chunked", "distributed"])
@pytest.mark.parametrize("tdt_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("fold", ["ordered", "default"])
@pytest.mark.parametrize("dim,lr,wd", [(256, -1.0, 0.0), (256, 0.05, 0.01), (100, -1.0, 0.0), (33, -1.0, 0.0), (64, 0.1, 0.0)])
def test_sgd_on_16bit_tables(gpu_env, knobs, mt, tdt_name, dim, lr, wd, fold):
    """Extension (BASELINE config 4, "fp16 scatter-add"): HALF / BF16 embeddings trained with SGD; lr = -1, wd = 0 is
    scatter-add. The reference trains fp32 tables only (embedding.cpp:61-63), so the semantics are this repo's:
    duplicates summed in fp32 in receive order, e' = e - lr (g + wd e) in fp32 from fp32(e), ONE rounding to the table
    dtype. The oracle is the fp32 oracle wrapped in exact widenings and that one rounding. dim 256 = 512 B rows (C4
    shape), a hot id with ~17 k duplicates exercises the LDS-DMA long-run kernel, dim 100 / 33 the shapes without it.
    fold = "ordered" (WM_GRAD_FOLD=ordered): the receive-order sum, bit for bit. fold = "default": since round 3 the 16-bit
    extension folds long runs as a tree (no reference bits to keep: parity unpinned) — the fp32 sum then differs from the
    ordered one in its last bits, so the rounded result may land on the neighbouring 16-bit value: every element within one
    unit in the last place of the ordered result, all but a few equal."""
    import torch
    import wholegraph_amd.torch as wgth
    if fold == "ordered":
        knobs.set("WM_GRAD_FOLD", "ordered")
    else:
        knobs.unset("WM_GRAD_FOLD")
    tdt = getattr(torch, tdt_name)
    n_rows, n_idx = 20011, 50001
    emb = wgth.create_embedding(gpu_env, mt, "cuda", tdt, [n_rows, dim])
    stride = emb.get_embedding_tensor().stride()[0]
    rng = np.random.default_rng(1000 + dim + int(lr * 100))
    init16 = torch.from_numpy(rng.standard_normal((n_rows, dim)).astype(np.float32)).to(tdt)
    local, _ = emb.get_embedding_tensor().get_local_tensor()
    local.copy_(init16.cuda())
    opt = wgth.create_wholememory_optimizer(emb, "sgd", {"weight_decay": wd})
    padded = np.zeros((n_rows, stride), np.float32)
    padded[:, :dim] = init16.float().numpy()
    tab = oracle.ShardedTable.from_full(padded, 1)
    tab.dim = dim
    ref_opt = oracle.Optimizer("sgd", n_rows, stride, weight_decay=wd)
    for step in range(2):
        idx = rng.integers(0, n_rows, n_idx).astype(np.int64)
        idx[::3] = idx[1]                       # one id with ~17 k duplicates
        g16 = torch.from_numpy(rng.standard_normal((n_idx, dim)).astype(np.float32)).to(tdt)
        emb.add_gradients(torch.from_numpy(idx).cuda(), g16.cuda())
        emb.need_apply = True
        opt.step(lr)
        oracle.gradient_apply(tab, [ref_opt], [idx], [g16.float().numpy()], lr)
        rounded = torch.from_numpy(tab.shards[0][:, :dim].copy()).to(tdt)     # the one rounding
        tab.shards[0][:, :dim] = rounded.float().numpy()
        torch.cuda.synchronize()
        got = local.cpu()
        if fold == "ordered":
            assert torch.equal(got.view(torch.int16), rounded.view(torch.int16)), "step %d" % step
        else:
            gb, rb = got.view(torch.int16).to(torch.int32), rounded.view(torch.int16).to(torch.int32)
            # neighbouring 16-bit values of one sign are neighbouring bit patterns (results next to zero are far from the hot row)
            off = (gb - rb).abs()
            assert int(off.max()) <= 1, "step %d: more than one unit in the last place from the ordered result" % step
            assert float((off != 0).float().mean()) < 1e-3, "step %d: too many elements differ" % step
            tab.shards[0][:, :dim] = got.float().numpy()          # the next step starts from what the device holds
    # every other optimizer is refused on 16-bit tables
    emb2 = wgth.create_embedding(gpu_env, mt, "cuda", tdt, [128, 8])
    from wholegraph_amd import binding as wmb
    with pytest.raises(wmb.WholeMemoryError):
        wgth.create_wholememory_optimizer(emb2, "adam", {})
    wgth.destroy_wholememory_optimizer(opt)
    wgth.destroy_embedding(emb)
    wgth.destroy_embedding(emb2)


def test_round_robin_embedding_gather(gpu_env):
    """round_robin_size != 0: padded row count (embedding.cpp:467-484) and index remap
    (map_indices_func.cu:26-45) — at world_size 1 the remap is the identity on [0, N)."""
    import torch
    import wholegraph_amd.torch as wgth
    n_rows, dim, rr = 10007, 32, 16
    emb = wgth.create_embedding(gpu_env, "chunked", "cuda", torch.float32, [n_rows, dim], round_robin_size=rr)
    assert emb.shape[0] == oracle.round_robin_total_entries(n_rows, 1, rr)
    local, _ = emb.get_embedding_tensor().get_local_tensor()
    full = oracle.fill_closed_form(np.float32, 0, emb.shape[0], dim)
    local.copy_(torch.from_numpy(full).cuda())
    idx = np.random.default_rng(0).integers(0, n_rows, 5000).astype(np.int64)
    out = emb.gather(torch.from_numpy(idx).cuda())
    mapped = oracle.round_robin_map(idx, 0, 1, rr)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), full[mapped])
    wgth.destroy_embedding(emb)


def test_save_load_roundtrip(gpu_env, tmp_path):
    """WholeMemoryEmbedding.save/load: "%s_part_%d_of_%d" raw row-major shards (torch/embedding.py:358-377)."""
    import torch
    import wholegraph_amd.torch as wgth
    n_rows, dim = 5003, 11
    emb = wgth.create_embedding(gpu_env, "distributed", "cuda", torch.float32, [n_rows, dim])
    opt = wgth.create_wholememory_optimizer(emb, "adam", {})
    local, _ = emb.get_embedding_tensor().get_local_tensor()
    data = np.random.default_rng(1).standard_normal((n_rows, dim)).astype(np.float32)
    local.copy_(torch.from_numpy(data).cuda())
    m, _ = emb.get_optimizer_state("m").get_local_tensor()
    m.copy_(torch.from_numpy(data * 2).cuda())
    torch.cuda.synchronize()
    prefix = str(tmp_path / "ckpt")
    emb.save(prefix)
    raw = np.fromfile(prefix + "_embedding_tensor_part_0_of_1", dtype=np.float32).reshape(n_rows, dim)
    assert np.array_equal(raw, data)  # file rows are dim wide, not stride wide
    local.zero_()
    m.zero_()
    emb.load(prefix)
    torch.cuda.synchronize()
    assert np.array_equal(local.cpu().numpy(), data) and np.array_equal(m.cpu().numpy(), data * 2)
    wgth.destroy_wholememory_optimizer(opt)
    wgth.destroy_embedding(emb)


@pytest.mark.parametrize("idt", [np.int64, np.int32])
@pytest.mark.parametrize("with_negatives", [False, True, "past_end", "only_junk"])
def test_gradient_apply_ignores_negative_ids(gpu_env, with_negatives, idt):
    """Negative ids are dropped by the bucketing (bucket_ids_func.cu:73): a batch with such entries must leave the table
    exactly as the same batch without them does. A single rank uses the caller's ids and gradient rows in place and drops
    the ids that address no row inside the owner-side sort (they read as one marker key behind every row) — negative ones,
    and ("past_end") ids beyond the table; "only_junk": a batch of nothing else leaves the table untouched."""
    import torch
    import wholegraph_amd.torch as wgth
    n_rows, dim, n = 50021, 64, 30000
    rng = np.random.default_rng(9)
    init = rng.standard_normal((n_rows, dim)).astype(np.float32)
    ids = rng.integers(0, n_rows, n).astype(idt)
    ids[:500] = ids[7]                                   # a long run too
    ids[500:520] = n_rows - 1                            # the last row: next to the marker key
    grads = rng.standard_normal((n, dim)).astype(np.float32)

    def run(batch_ids, batch_grads):
        emb = wgth.create_embedding(gpu_env, "distributed", "cuda", torch.float32, [n_rows, dim])
        local, _ = emb.get_embedding_tensor().get_local_tensor()
        local.copy_(torch.from_numpy(init).cuda())
        wgth.create_wholememory_optimizer(emb, "sgd", {"weight_decay": 0.01})
        emb.add_gradients(torch.from_numpy(batch_ids).cuda(), torch.from_numpy(batch_grads).cuda())
        emb.need_apply = True
        emb.apply_gradients(0.05)
        torch.cuda.synchronize()
        out = local.cpu().numpy().copy()
        wgth.destroy_embedding(emb)
        return out

    if with_negatives == "only_junk":
        junk = np.array([-1, -7, n_rows, n_rows + 3, np.iinfo(idt).max, np.iinfo(idt).min] * 50, dtype=idt)
        got = run(junk, np.full((len(junk), dim), 1e9, dtype=np.float32))
        assert got.tobytes() == init.tobytes()
        return
    want = run(ids, grads)
    if with_negatives:
        pos = np.sort(rng.choice(n + 700, 700, replace=False))
        # interleave 700 junk entries (with junk gradient rows) at random positions, order of the rest unchanged
        keep = np.ones(n + 700, dtype=bool)
        keep[pos] = False
        mixed_ids = np.full(n + 700, -1, dtype=idt)
        if with_negatives == "past_end":
            mixed_ids[pos[::3]] = n_rows
            mixed_ids[pos[1::3]] = np.iinfo(idt).max
            mixed_ids[pos[2::3]] = n_rows + 12345
        mixed_ids[keep] = ids
        mixed_grads = np.full((n + 700, dim), 1e9, dtype=np.float32)
        mixed_grads[keep] = grads
        got = run(mixed_ids, mixed_grads)
    else:
        got = run(ids.copy(), grads.copy())
    assert got.tobytes() == want.tobytes()
    uniq, dg = oracle.dedup_grads(ids.astype(np.int64), grads)
    ref = init.copy()
    oracle.Optimizer("sgd", n_rows, dim, weight_decay=0.01).step(uniq, dg, ref, dim, 0, dim, 0.05)
    assert want.tobytes() == ref.tobytes()


# ---- relaxed-order ("tree") fold of long duplicate runs: WM_GRAD_FOLD=tree / wm_optimizer_args::fold_mode = 1 -------------
def _dedup_apply_sgd(wmb, torch, ids, grads, table, stride, dim, lr, fold):
    """one SGD step (weight decay 0) through the raw stage, with the given fold order"""
    import os
    env, stream = _env()
    d_table = torch.from_numpy(table.copy()).cuda()
    d_ids, d_grads = torch.from_numpy(ids).cuda(), torch.from_numpy(grads).cuda()
    arr = (C.c_float * 6)(0.0, 1e-8, 0.9, 0.999, 0.99, 0.0)
    nu = C.c_int64(-1)
    os.environ["WM_GRAD_FOLD"] = fold
    wmb.reload_knobs()
    try:
        wmb.check(wmb.lib().wholememory_ext_dedup_apply(
            d_ids.data_ptr(), wmb.DT_INT64, len(ids), d_grads.data_ptr(), grads.shape[1], dim, d_table.data_ptr(), stride, 0,
            table.shape[0], 1, arr, lr, None, None, C.byref(nu), env, stream))
        torch.cuda.synchronize()
    finally:
        del os.environ["WM_GRAD_FOLD"]
        wmb.reload_knobs()
    return d_table.cpu().numpy(), nu.value


def _run_mix(rng, n, rows, hot):
    """ids with run lengths on both sides of every threshold of the fold: singles, short runs, runs of 33 ... 1025 rows
    (one segment), and `hot` ids with thousands of rows (several segments)"""
    ids = rng.integers(0, rows, n).astype(np.int64)
    pos = 0
    for k, length in enumerate([2, 31, 32, 33, 40, 127, 128, 129, 255, 256, 257, 511, 512, 513, 600, 1025] + list(hot)):
        ids[pos:pos + length] = rows - 1 - k
        pos += length
    rng.shuffle(ids)
    return ids


@pytest.mark.parametrize("dim", [128, 32, 100, 260])
def test_tree_fold_exact_on_integer_gradients(gpu_env, dim):
    """Integer-valued gradients: every partial sum is exactly representable, so ANY association order gives the same bits —
    the tree fold must equal the ordered oracle exactly (which also proves that no row is dropped, doubled or misplaced)."""
    import torch
    from wholegraph_amd import binding as wmb
    rng = np.random.default_rng(dim)
    rows, n = 5003, 120_000
    stride = int(oracle.align_embedding_dim(dim, 4))
    ids = _run_mix(rng, n, rows, hot=[5000, 20000, 3000])
    grads = rng.integers(-3, 4, (n, dim)).astype(np.float32)
    table = np.zeros((rows, stride), np.float32)
    table[:, :dim] = rng.integers(-8, 9, (rows, dim)).astype(np.float32)
    got, nu = _dedup_apply_sgd(wmb, torch, ids, grads, table, stride, dim, -1.0, "tree")     # lr -1, wd 0: scatter-add
    uniq, dg = oracle.dedup_grads(ids, grads)
    ref = table.copy()
    oracle.Optimizer("sgd", rows, stride).step(uniq, dg, ref, stride, 0, dim, -1.0)
    assert nu == len(uniq)
    assert got.tobytes() == ref.tobytes(), "tree fold differs from the exact integer sums"
    ordered, _ = _dedup_apply_sgd(wmb, torch, ids, grads, table, stride, dim, -1.0, "ordered")
    assert ordered.tobytes() == ref.tobytes()


@pytest.mark.parametrize("case", ["normal", "cancellation"])
def test_tree_fold_within_rounding_of_the_ordered_sum(gpu_env, case):
    """Real-valued gradients: the tree fold is not the reference's association order (exchange_embeddings_nccl_func.cu:76-103),
    so it is held to the forward-error bound of fp32 summation instead of to its bits: per element
      |tree - exact| <= 2^-24 x (longest chain of the tree) x sum |g_i|          (chains: 512 / 8 rows + 8 slots + segments)
    and it must not be LESS accurate than the ordered fold (whose chain is the whole run) beyond noise; rows whose runs stay
    under the threshold are bit-identical to the ordered oracle. `cancellation`: pairs of +/-1e4 around values of order 1."""
    import torch
    from wholegraph_amd import binding as wmb
    rng = np.random.default_rng(11)
    rows, n, dim = 4001, 200_000, 128
    ids = _run_mix(rng, n, rows, hot=[60_000, 9000])
    grads = rng.standard_normal((n, dim)).astype(np.float32)
    if case == "cancellation":
        big = (rng.integers(0, 2, (n // 2, dim)) * 2 - 1).astype(np.float32) * 1e4
        grads[0:2 * (n // 2):2] += big
        grads[1:2 * (n // 2):2] -= big
    table = np.zeros((rows, dim), np.float32)
    table[:] = rng.standard_normal((rows, dim)).astype(np.float32)
    tree, _ = _dedup_apply_sgd(wmb, torch, ids, grads, table, dim, dim, -1.0, "tree")
    ordered, _ = _dedup_apply_sgd(wmb, torch, ids, grads, table, dim, dim, -1.0, "ordered")
    uniq, dg = oracle.dedup_grads(ids, grads)
    ref = table.copy()
    oracle.Optimizer("sgd", rows, dim).step(uniq, dg, ref, dim, 0, dim, -1.0)
    assert ordered.tobytes() == ref.tobytes()                       # the default stays the reference order, bit for bit
    exact = table.astype(np.float64)
    np.add.at(exact, ids, grads.astype(np.float64))
    sum_abs = np.abs(table).astype(np.float64)
    np.add.at(sum_abs, ids, np.abs(grads).astype(np.float64))
    counts = np.bincount(ids, minlength=rows)
    short = counts <= 128                                           # (kTreeMin in optim.hip)
    assert tree[short].tobytes() == ref[short].tobytes(), "runs under the threshold must keep the reference bits"
    chain = 512 // 8 + 8 + (counts.max() + 511) // 512 + 2
    bound = 2.0 ** -24 * chain * sum_abs
    err_tree, err_ord = np.abs(tree - exact), np.abs(ordered - exact)
    assert np.all(err_tree <= bound + 1e-30), "tree fold outside the forward-error bound: max ratio %g" % (err_tree / bound).max()
    long_rows = counts > 128
    assert err_tree[long_rows].mean() <= 1.5 * err_ord[long_rows].mean() + 1e-12, (err_tree[long_rows].mean(), err_ord[long_rows].mean())
    # and relative to the magnitude of what was summed the two folds agree to ~1e-6
    rel = np.abs(tree - ordered) / np.maximum(sum_abs, 1e-30)
    assert rel.max() < 2e-6, rel.max()


def test_tree_fold_is_the_default_of_the_16bit_extension_and_deterministic(gpu_env):
    """HALF tables (the C4 extension, parity unpinned by the reference): the tree fold is the default; two runs give the same
    bits, integer-valued gradients give the exact sums."""
    import torch
    import wholegraph_amd.torch as wgth
    rows, dim, n = 3001, 256, 90_000
    rng = np.random.default_rng(5)
    ids = _run_mix(rng, n, rows, hot=[30_000, 4000])
    grads = rng.integers(-1, 2, (n, dim)).astype(np.float16)
    results = []
    for _ in range(2):
        emb = wgth.create_embedding(gpu_env, "continuous", "cuda", torch.float16, [rows, dim])
        local, _ = emb.get_embedding_tensor().get_local_tensor()
        local.zero_()
        opt = wgth.create_wholememory_optimizer(emb, "sgd", {"weight_decay": 0.0})
        emb.add_gradients(torch.from_numpy(ids).cuda(), torch.from_numpy(grads).cuda())
        emb.need_apply = True
        opt.step(-1.0)
        torch.cuda.synchronize()
        results.append(local.cpu().numpy().copy())
        wgth.destroy_wholememory_optimizer(opt)
        wgth.destroy_embedding(emb)
    want = np.zeros((rows, dim), np.float64)
    np.add.at(want, ids, grads.astype(np.float64))
    assert np.array_equal(results[0], results[1])
    assert np.array_equal(results[0].astype(np.float64), want.astype(np.float16).astype(np.float64))


def test_grad_fold_is_an_optimizer_parameter(gpu_env):
    """create_wholememory_optimizer(emb, "sgd", {"grad_fold": "tree"}) (-> wholememory_optimizer_set_parameter(opt, "grad_fold",
    1.0)): a user selects the tree fold without touching the environment. One id with 20 000 duplicate gradient rows:
    "ordered" (and the default of an fp32 table) reproduces the receive-order sum bit for bit, "tree" differs from it in the
    last bits on real-valued gradients (so the parameter did reach the kernels), stays within the fp32 forward-error bound of
    the exact sum, and is exact — equal to the ordered result — on integer-valued gradients."""
    import torch
    import wholegraph_amd.torch as wgth
    n_rows, dim, n = 5003, 128, 30_000
    rng = np.random.default_rng(77)
    ids = rng.integers(0, n_rows, n).astype(np.int64)
    ids[:20_000] = 1234
    results = {}
    for grads_kind in ("real", "integer"):
        g = rng.standard_normal((n, dim)).astype(np.float32) if grads_kind == "real" else rng.integers(-3, 4, (n, dim)).astype(np.float32)
        for fold in (None, "ordered", "tree"):
            emb = wgth.create_embedding(gpu_env, "chunked", "cuda", torch.float32, [n_rows, dim])
            local, _ = emb.get_embedding_tensor().get_local_tensor()
            local.zero_()
            opt = wgth.create_wholememory_optimizer(emb, "sgd", {} if fold is None else {"grad_fold": fold})
            emb.add_gradients(torch.from_numpy(ids).cuda(), torch.from_numpy(g).cuda())
            emb.need_apply = True
            emb.apply_gradients(-1.0)                       # lr = -1, wd = 0: scatter-add
            torch.cuda.synchronize()
            results[(grads_kind, fold)] = local[:, :dim].cpu().numpy().copy()
            wgth.destroy_wholememory_optimizer(opt)
            wgth.destroy_embedding(emb)
        exact = np.zeros((n_rows, dim), np.float64)
        np.add.at(exact, ids, g.astype(np.float64))
        ordered, tree, default = results[(grads_kind, "ordered")], results[(grads_kind, "tree")], results[(grads_kind, None)]
        assert default.tobytes() == ordered.tobytes()                      # fp32 tables: the reference's order unless asked
        uniq, dg = oracle.dedup_grads(ids, g)
        assert ordered[uniq].tobytes() == dg[:, :dim].astype(np.float32).tobytes()   # the receive-order sum, bit for bit
        if grads_kind == "integer":
            assert tree.tobytes() == ordered.tobytes() and np.array_equal(tree.astype(np.float64), exact)
        else:
            assert tree.tobytes() != ordered.tobytes(), "the grad_fold parameter did not reach the kernels"
            bound = 20_000 * np.finfo(np.float32).eps * np.abs(g[:20_000]).sum(axis=0).max()
            assert np.abs(tree.astype(np.float64) - exact).max() <= bound


@pytest.mark.parametrize("kind,code,params", [("adam", 2, {"weight_decay": 0.01}), ("adam", 2, {"weight_decay": 0.02, "adam_w": 1.0}),
                                              ("rmsprop", 3, {"alpha": 0.9}), ("adagrad", 4, {"weight_decay": 0.01})])
@pytest.mark.parametrize("dim,idt", [(128, np.int64), (36, np.int32)])
def test_tree_fold_stateful_optimizers_exact_on_integer_gradients(gpu_env, knobs, kind, code, params, dim, idt):
    """The tree fold in front of the stateful optimizers: with integer-valued gradients the folded sums are exact, so table
    AND optimizer states must equal the ordered oracle bit for bit over several steps (long runs of one and of several
    segments, LazyAdam's per-row beta powers advanced once per listed run)."""
    import torch
    from wholegraph_amd import binding as wmb
    knobs.set("WM_GRAD_FOLD", "tree")
    rng = np.random.default_rng(dim + code)
    local_rows, local_off, n_recv = 3001, 777, 60_000
    stride = int(oracle.align_embedding_dim(dim, 4))
    table = np.zeros((local_rows, stride), np.float32)
    table[:, :dim] = rng.standard_normal((local_rows, dim)).astype(np.float32)
    ids = (local_off + _run_mix(rng, n_recv, local_rows, hot=[9000, 2500])).astype(idt)
    p = dict(weight_decay=0.0, epsilon=1e-8, beta1=0.9, beta2=0.999, alpha=0.99, adam_w=0.0)
    p.update(params)
    ref_opt = oracle.Optimizer(kind, local_rows, stride, **params)
    d_table = torch.from_numpy(table.copy()).cuda()
    d_pr = torch.ones((local_rows, 2), device="cuda") if kind == "adam" else None
    d_pe = torch.zeros((local_rows, (2 if kind == "adam" else 1) * stride), device="cuda")
    arr = (C.c_float * 6)(p["weight_decay"], p["epsilon"], p["beta1"], p["beta2"], p["alpha"], p["adam_w"])
    env, stream = _env()
    ref_table = table.copy()
    d_ids = torch.from_numpy(ids).cuda()
    for step in range(3):
        grads = rng.integers(-2, 3, (n_recv, dim)).astype(np.float32)
        d_grads = torch.from_numpy(grads).cuda()
        nu = C.c_int64(-1)
        wmb.check(wmb.lib().wholememory_ext_dedup_apply(
            d_ids.data_ptr(), wmb.DT_INT if idt == np.int32 else wmb.DT_INT64, n_recv, d_grads.data_ptr(), dim, dim,
            d_table.data_ptr(), stride, local_off, local_rows, code, arr, 0.03, d_pe.data_ptr(),
            d_pr.data_ptr() if d_pr is not None else None, C.byref(nu), env, stream))
        torch.cuda.synchronize()
        uniq, dg = oracle.dedup_grads(ids, grads)
        assert nu.value == len(uniq)
        ref_opt.step(uniq, dg, ref_table, stride, local_off, dim, 0.03)
        assert d_table.cpu().numpy().tobytes() == ref_table.tobytes(), "%s step %d: table differs" % (kind, step)
    assert d_pe.cpu().numpy().tobytes() == ref_opt.per_element.tobytes()
    if kind == "adam":
        assert d_pr.cpu().numpy().tobytes() == ref_opt.per_row.tobytes()


@pytest.mark.parametrize("mt", ["chunked", "distributed"])
@pytest.mark.parametrize("kind,params", [("sgd", {}), ("adam", {"weight_decay": 0.01})])
def test_embedding_row_align_knob(gpu_env, knobs, tmp_path, mt, kind, params):
    """Extension, opt-in: WM_EMBEDDING_ROW_ALIGN=128 pads the row stride of embeddings (and of their optimizer states) to whole
    128-byte lines instead of the reference's 16 bytes (embedding.cpp:43-50). Everything a user sees is unchanged: the logical
    shape, gather / training results (bit for bit against the oracle on the reference's 16-byte stride) and the files."""
    import torch
    import wholegraph_amd.torch as wgth
    knobs.set("WM_EMBEDDING_ROW_ALIGN", 128)
    n_rows, dim, n_idx = 40001, 50, 30005          # 200 B rows -> stride 64 elements
    emb = wgth.create_embedding(gpu_env, mt, "cuda", torch.float32, [n_rows, dim])
    assert emb.get_embedding_tensor().stride() == (64, 1) and emb.shape == (n_rows, dim)
    rng = np.random.default_rng(5)
    init = rng.standard_normal((n_rows, dim)).astype(np.float32)
    local, _ = emb.get_embedding_tensor().get_local_tensor()
    local.copy_(torch.from_numpy(init).cuda())
    opt = wgth.create_wholememory_optimizer(emb, kind, params)
    module = wgth.WholeMemoryEmbeddingModule(emb)
    module.train()
    padded = np.zeros((n_rows, 52), np.float32)     # the oracle keeps the reference's stride
    padded[:, :dim] = init
    tab = oracle.ShardedTable.from_full(padded, 1)
    tab.dim = dim
    ref_opt = oracle.Optimizer(kind, n_rows, 52, **params)
    for step in range(2):
        idx = rng.integers(0, n_rows, n_idx).astype(np.int64)
        idx[::3] = idx[0]
        w = rng.standard_normal((n_idx, dim)).astype(np.float32)
        out = module(torch.from_numpy(idx).cuda())
        exp_out = np.zeros((n_idx, dim), np.float32)
        oracle.gather(tab, idx, exp_out)
        assert out.detach().cpu().numpy().tobytes() == exp_out.tobytes()
        (out * torch.from_numpy(w).cuda()).sum().backward()
        opt.step(0.02)
        oracle.gradient_apply(tab, [ref_opt], [idx], [w], 0.02)
        torch.cuda.synchronize()
        assert local.cpu().numpy().tobytes() == tab.shards[0][:, :dim].tobytes(), "step %d" % step
    # files hold logical rows: written with the wide stride, read back into a table with the reference's stride
    prefix = str(tmp_path / "aligned")
    emb.save(prefix)
    raw = np.fromfile(prefix + "_embedding_tensor_part_0_of_1", dtype=np.float32).reshape(n_rows, dim)
    assert raw.tobytes() == tab.shards[0][:, :dim].tobytes()
    knobs.unset("WM_EMBEDDING_ROW_ALIGN")
    emb2 = wgth.create_embedding(gpu_env, mt, "cuda", torch.float32, [n_rows, dim])
    assert emb2.get_embedding_tensor().stride() == (52, 1)
    emb2.load(prefix, ignore_embedding=False) if kind == "sgd" else emb2.get_embedding_tensor().from_file_prefix(prefix + "_embedding_tensor")
    torch.cuda.synchronize()
    local2, _ = emb2.get_embedding_tensor().get_local_tensor()
    assert local2.cpu().numpy().tobytes() == raw.tobytes()
    wgth.destroy_wholememory_optimizer(opt)
    wgth.destroy_embedding(emb)
    wgth.destroy_embedding(emb2)


@pytest.mark.parametrize("n", [150_000, 300_000, 1_000_000])
@pytest.mark.parametrize("idt", [np.int64, np.int32])
def test_id_sort_routes_agree(gpu_env, knobs, n, idt):
    """The id sort of a mid-sized batch takes rocPRIM's merge sort below WM_SORT_RADIX_MIN (196608) items and the onesweep
    passes from there up (rocPRIM alone would merge up to 2^20 items). A stable sort has one answer: scatter-add of the same
    real-valued gradients through either route leaves the same bytes in the table, and they are the receive-order sums of the
    oracle (reference exchange_embeddings_nccl_func.cu:93-206)."""
    import torch
    import wholegraph_amd.torch as wgth
    n_rows, dim = 200_003, 32
    rng = np.random.default_rng(n)
    ids = rng.integers(0, n_rows, n).astype(idt)
    ids[::7] = 4242                                                 # one long run among the short ones
    g = rng.standard_normal((n, dim)).astype(np.float32)
    tables = []
    for route in ("1", str(1 << 40)):                               # always the radix passes / the merge sort wherever rocPRIM allows it
        knobs.set("WM_SORT_RADIX_MIN", route)
        emb = wgth.create_embedding(gpu_env, "chunked", "cuda", torch.float32, [n_rows, dim])
        local, _ = emb.get_embedding_tensor().get_local_tensor()
        local.zero_()
        opt = wgth.create_wholememory_optimizer(emb, "sgd", {})
        emb.add_gradients(torch.from_numpy(ids).cuda(), torch.from_numpy(g).cuda())
        emb.need_apply = True
        emb.apply_gradients(-1.0)                                   # lr = -1, wd = 0: scatter-add
        torch.cuda.synchronize()
        tables.append(local[:, :dim].cpu().numpy().copy())
        wgth.destroy_wholememory_optimizer(opt)
        wgth.destroy_embedding(emb)
    assert tables[0].tobytes() == tables[1].tobytes()
    uniq, dg = oracle.dedup_grads(ids.astype(np.int64), g)
    assert tables[0][uniq].tobytes() == dg[:, :dim].astype(np.float32).tobytes()
