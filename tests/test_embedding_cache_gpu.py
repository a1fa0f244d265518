"""Cached embeddings (device row cache): the cache is transparent — every gather returns exactly what the uncached
table holds — and does its job: hot rows become resident, lookups hit.

Reference: cpp/src/wholememory/embedding.cpp:564-892 (device_cached_host_embedding, local_cached_global_readonly_
embedding), embedding_cache.{hpp,cpp}; parameter shapes follow cpp/tests/wholememory_ops/wholememory_embedding_tests.cu
(cache ratios 0.1-0.5, HOST raw tables with DEVICE caches, local device caches of chunked tables)."""
import ctypes as C

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _info(emb):
    from wholegraph_amd import binding as wmb
    v = [C.c_int64() for _ in range(5)]
    wmb.check(wmb.lib().wholememory_ext_embedding_cache_info(emb.wmb_embedding, *[C.byref(x) for x in v]))
    return dict(zip(("slots", "occupied", "dirty", "hits", "lookups"), [x.value for x in v]))


def _zipf(rng, n, n_rows, dtype=np.int64):
    k = rng.zipf(1.2, n).astype(np.uint64)
    return ((k * np.uint64(2654435761)) % np.uint64(n_rows)).astype(dtype)


def _fill(emb, n_rows, dim, np_dt, host):
    import torch
    full = oracle.fill_closed_form(np_dt, 0, n_rows, dim)
    local, start = emb.get_embedding_tensor().get_local_tensor(host_view=host)
    local.copy_(torch.from_numpy(full))
    torch.cuda.synchronize()
    return full


@pytest.mark.parametrize("idt", [np.int64, np.int32])
@pytest.mark.parametrize("mt,loc", [("chunked", "cpu"), ("distributed", "cpu"), ("continuous", "cpu"), ("chunked", "cuda")])
def test_device_cache_of_own_shard_readonly(gpu_env, mt, loc, idt):
    import torch
    import wholegraph_amd.torch as wgth
    n_rows, dim = 60013, 64
    policy = wgth.create_wholememory_cache_policy(gpu_env, memory_type=mt, memory_location="cuda", access_type="readonly",
                                                  ratio=0.1)
    emb = wgth.create_embedding(gpu_env, mt, loc, torch.float32, [n_rows, dim], cache_policy=policy)
    full = _fill(emb, n_rows, dim, np.float32, loc == "cpu")
    rng = np.random.default_rng(7)
    info0 = _info(emb)
    assert info0["slots"] % 64 == 0 and 0.09 * n_rows <= info0["slots"] <= 0.11 * n_rows + 64 and info0["occupied"] == 0
    hits_per_batch = []
    for b in range(6):
        idx = _zipf(rng, 20000, n_rows, idt)
        idx[::97] = -1
        out = torch.full((len(idx), dim), -7.0, device="cuda")
        before = _info(emb)["hits"]
        got = emb.gather(torch.from_numpy(idx).cuda(), out=out)
        torch.cuda.synchronize()
        want = np.full((len(idx), dim), -7.0, np.float32)
        want[idx >= 0] = full[idx[idx >= 0]]
        assert got.cpu().numpy().tobytes() == want.tobytes(), "batch %d" % b
        hits_per_batch.append(_info(emb)["hits"] - before)
    info = _info(emb)
    assert 0 < info["occupied"] <= info["slots"] and info["dirty"] == 0
    # a Zipf stream: from the second batch on most lookups are served from the cache
    assert hits_per_batch[-1] > 0.5 * 20000, hits_per_batch
    # the hottest rows are resident
    hot = _zipf(np.random.default_rng(1), 200000, n_rows)
    top = np.argsort(-np.bincount(hot, minlength=n_rows))[:20].astype(idt)
    emb.set_adjust_cache(False)
    before = _info(emb)["hits"]
    got = emb.gather(torch.from_numpy(top).cuda())
    assert np.array_equal(got.cpu().numpy(), full[top]) and _info(emb)["hits"] - before >= 18
    # dropping the cache empties it; lookups keep returning the table
    emb.drop_all_cache()
    assert _info(emb)["occupied"] == 0
    got = emb.gather(torch.from_numpy(top).cuda())
    assert np.array_equal(got.cpu().numpy(), full[top])
    wgth.destroy_embedding(emb)
    wgth.destroy_wholememory_cache_policy(policy)


@pytest.mark.parametrize("mt,loc", [("chunked", "cuda"), ("continuous", "cpu"), ("distributed", "cuda")])
def test_local_cache_of_global_table(gpu_env, mt, loc):
    """cache communicator != embedding communicator: each rank keeps its own read-only cache of the whole table."""
    import torch
    import wholegraph_amd.torch as wgth
    local_comm = wgth.create_group_communicator(1)   # a different communicator object
    n_rows, dim = 30011, 100
    policy = wgth.create_wholememory_cache_policy(local_comm, memory_type="continuous", memory_location="cuda",
                                                  access_type="readonly", ratio=0.25)
    emb = wgth.create_embedding(gpu_env, mt, loc, torch.float16, [n_rows, dim], cache_policy=policy)
    full = _fill(emb, n_rows, dim, np.float16, loc == "cpu")
    rng = np.random.default_rng(3)
    for b in range(4):
        idx = _zipf(rng, 15000, n_rows)
        got = emb.gather(torch.from_numpy(idx).cuda())
        assert got.cpu().numpy().tobytes() == full[idx].tobytes()
    info = _info(emb)
    assert info["occupied"] > 0 and info["hits"] > 0.3 * info["lookups"]
    wgth.destroy_embedding(emb)
    wgth.destroy_wholememory_cache_policy(policy)


def test_cache_policy_validation(gpu_env):
    import torch
    import wholegraph_amd.torch as wgth
    from wholegraph_amd import binding as wmb
    other = wgth.create_group_communicator(1)
    bad = [
        (wgth.create_wholememory_cache_policy(gpu_env, memory_type="chunked", memory_location="cpu", access_type="readonly", ratio=0.2), "chunked"),
        (wgth.create_wholememory_cache_policy(gpu_env, memory_type="continuous", memory_location="cuda", access_type="readonly", ratio=0.2), "distributed"),
        (wgth.create_wholememory_cache_policy(other, memory_type="chunked", memory_location="cuda", access_type="readwrite", ratio=0.2), "chunked"),
        (wgth.create_wholememory_cache_policy(other, memory_type="distributed", memory_location="cuda", access_type="readonly", ratio=0.2), "chunked"),
    ]
    for policy, mt in bad:
        with pytest.raises(wmb.WholeMemoryError):
            wgth.create_embedding(gpu_env, mt, "cpu", torch.float32, [1000, 16], cache_policy=policy)
    with pytest.raises(wmb.WholeMemoryError):
        wgth.create_wholememory_cache_policy(gpu_env, ratio=0.0001)


@pytest.mark.parametrize("kind,params", [("sgd", {"weight_decay": 0.01}), ("adam", {})])
@pytest.mark.parametrize("mt", ["chunked", "distributed"])
def test_readwrite_cache_training_and_writeback(gpu_env, mt, kind, params):
    """Training a HOST embedding through its read-write device cache (reference device_cached_host_embedding with
    WHOLEMEMORY_AT_READWRITE): resident rows are updated in the cache and marked modified, gathers see the new values at
    once, the raw table catches up at write-back. Every value is compared bit for bit with the uncached oracle."""
    import torch
    import wholegraph_amd.torch as wgth
    n_rows, dim, n_idx = 40009, 32, 30000
    policy = wgth.create_wholememory_cache_policy(gpu_env, memory_type=mt, memory_location="cuda", access_type="readwrite",
                                                  ratio=0.2)
    emb = wgth.create_embedding(gpu_env, mt, "cpu", torch.float32, [n_rows, dim], cache_policy=policy)
    rng = np.random.default_rng(11)
    init = rng.standard_normal((n_rows, dim)).astype(np.float32)
    local, _ = emb.get_embedding_tensor().get_local_tensor(host_view=True)
    local.copy_(torch.from_numpy(init))
    opt = wgth.create_wholememory_optimizer(emb, kind, params)
    tab = oracle.ShardedTable.from_full(init.copy(), 1)
    ref_opt = oracle.Optimizer(kind, n_rows, dim, **params)
    for step in range(4):
        idx = _zipf(rng, n_idx, n_rows)
        got = emb.gather(torch.from_numpy(idx).cuda(), is_training=True)       # also warms the cache (adjust_cache)
        exp = np.zeros((n_idx, dim), np.float32)
        oracle.gather(tab, idx, exp)
        assert got.detach().cpu().numpy().tobytes() == exp.tobytes(), "gather before step %d" % step
        g = rng.standard_normal((n_idx, dim)).astype(np.float32)
        emb.add_gradients(torch.from_numpy(idx).cuda(), torch.from_numpy(g).cuda())
        opt.step(0.05)
        oracle.gradient_apply(tab, [ref_opt], [idx], [g], 0.05)
        probe = _zipf(np.random.default_rng(step), 5000, n_rows)
        emb.set_adjust_cache(False)
        got = emb.gather(torch.from_numpy(probe).cuda())
        emb.set_adjust_cache(True)
        exp = np.zeros((len(probe), dim), np.float32)
        oracle.gather(tab, probe, exp)
        assert got.cpu().numpy().tobytes() == exp.tobytes(), "gather after step %d" % step
    info = _info(emb)
    assert info["dirty"] > 0 and info["occupied"] >= info["dirty"]
    torch.cuda.synchronize()
    assert local.numpy().tobytes() != tab.shards[0].tobytes(), "raw table should lag behind the modified cache lines"
    if kind == "adam":   # the per-element states of resident rows live in the cache too and lag in the raw state table
        m_local, _ = emb.get_optimizer_state("m").get_local_tensor(host_view=True)
        assert m_local.numpy().tobytes() != ref_opt.per_element[:, :dim].tobytes()
    emb.writeback_all_cache()
    assert _info(emb)["dirty"] == 0
    assert local.numpy().tobytes() == tab.shards[0].tobytes(), "raw table after write-back differs from the oracle"
    if kind == "adam":
        v_local, _ = emb.get_optimizer_state("v").get_local_tensor(host_view=True)
        assert m_local.numpy().tobytes() == ref_opt.per_element[:, :dim].tobytes(), "state m after write-back"
        assert v_local.numpy().tobytes() == ref_opt.per_element[:, dim:2 * dim].tobytes(), "state v after write-back"
    emb.drop_all_cache()
    assert _info(emb)["occupied"] == 0
    probe = rng.integers(0, n_rows, 4000)
    got = emb.gather(torch.from_numpy(probe).cuda())
    assert np.array_equal(got.cpu().numpy(), tab.shards[0][probe])
    wgth.destroy_wholememory_optimizer(opt)
    wgth.destroy_embedding(emb)
    wgth.destroy_wholememory_cache_policy(policy)


def test_readonly_cache_refuses_training(gpu_env):
    import torch
    import wholegraph_amd.torch as wgth
    from wholegraph_amd import binding as wmb
    policy = wgth.create_wholememory_cache_policy(gpu_env, memory_type="chunked", memory_location="cuda",
                                                  access_type="readonly", ratio=0.2)
    emb = wgth.create_embedding(gpu_env, "chunked", "cpu", torch.float32, [1000, 16], cache_policy=policy)
    with pytest.raises(wmb.WholeMemoryError):
        wgth.create_wholememory_optimizer(emb, "sgd", {})
    wgth.destroy_embedding(emb)
