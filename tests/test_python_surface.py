"""The Python surface mirrors pylibwholegraph.torch for the embedding path (names + signatures), and the
view helpers of WholeMemoryTensor alias the real memory."""
import inspect

import pytest


def test_names_and_signatures():
    import wholegraph_amd.torch as wgth
    # reference python/pylibwholegraph/pylibwholegraph/torch/__init__.py:14-78, embedding-path subset
    for name in ["WholeMemoryCommunicator", "create_group_communicator", "destroy_communicator", "get_global_communicator",
                 "get_local_node_communicator", "get_local_device_communicator", "split_communicator",
                 "get_local_mnnvl_communicator", "WholeMemoryOptimizer", "create_wholememory_optimizer",
                 "destroy_wholememory_optimizer", "WholeMemoryCachePolicy", "create_builtin_cache_policy",
                 "create_wholememory_cache_policy", "destroy_wholememory_cache_policy", "WholeMemoryEmbedding",
                 "create_embedding", "create_embedding_from_filelist", "destroy_embedding", "WholeMemoryEmbeddingModule",
                 "init", "init_torch_env", "init_torch_env_and_create_wm_comm", "finalize", "WholeMemoryTensor",
                 "create_wholememory_tensor", "create_wholememory_tensor_from_filelist", "destroy_wholememory_tensor",
                 "get_part_file_name", "get_part_file_list", "wholememory_dtype_to_torch_dtype",
                 "torch_dtype_to_wholememory_dtype"]:
        assert hasattr(wgth, name), name
    sig = inspect.signature(wgth.create_embedding)
    # the reference's parameters, in its order; one keyword-only extension behind them (placement_probe, round 5)
    assert list(sig.parameters) == ["comm", "memory_type", "memory_location", "dtype", "sizes", "cache_policy",
                                    "embedding_entry_partition", "random_init", "gather_sms", "round_robin_size",
                                    "placement_probe"]
    assert sig.parameters["placement_probe"].kind is inspect.Parameter.KEYWORD_ONLY and sig.parameters["placement_probe"].default is None
    for kw in ("cache_policy", "embedding_entry_partition", "random_init", "gather_sms", "round_robin_size"):
        assert sig.parameters[kw].kind is inspect.Parameter.KEYWORD_ONLY
    g = inspect.signature(wgth.WholeMemoryEmbedding.gather)
    assert list(g.parameters)[:4] == ["self", "indice", "is_training", "force_dtype"]
    assert list(inspect.signature(wgth.create_wholememory_tensor).parameters) == [
        "comm", "memory_type", "memory_location", "sizes", "dtype", "strides", "tensor_entry_partition"]
    assert wgth.get_part_file_name("p", 1, 4) == "p_part_1_of_4"
    assert wgth.get_part_file_list("p", 2) == ["p_part_0_of_2", "p_part_1_of_2"]
    import pylibwholegraph.torch as alias
    assert alias is wgth


def test_string_enums():
    from wholegraph_amd.torch import utils as u
    from wholegraph_amd import binding as B
    assert [u.str_to_wmb_wholememory_memory_type(s) for s in ("continuous", "chunked", "distributed", "hierarchy")] == \
        [B.MT_CONTINUOUS, B.MT_CHUNKED, B.MT_DISTRIBUTED, B.MT_HIERARCHY]
    assert [u.str_to_wmb_wholememory_location(s) for s in ("cuda", "cpu")] == [B.ML_DEVICE, B.ML_HOST]
    assert [u.str_to_wmb_wholememory_optimizer_type(s) for s in ("sgd", "adam", "rmsprop", "adagrad")] == \
        [B.OPT_SGD, B.OPT_LAZY_ADAM, B.OPT_RMSPROP, B.OPT_ADAGRAD]
    for fn, bad in [(u.str_to_wmb_wholememory_memory_type, "x"), (u.str_to_wmb_wholememory_location, "gpu"),
                    (u.str_to_wmb_wholememory_optimizer_type, "adamw"), (u.str_to_wmb_wholememory_access_type, "w")]:
        with pytest.raises(ValueError):
            fn(bad)


@pytest.mark.gpu
@pytest.mark.parametrize("loc", ["cuda", "cpu"])
@pytest.mark.parametrize("mt", ["continuous", "chunked", "distributed"])
def test_tensor_views_alias_memory(gpu_env, mt, loc):
    import torch
    import wholegraph_amd.torch as wgth
    n, dim = 3001, 16
    wm = wgth.create_wholememory_tensor(gpu_env, mt, loc, [n, dim], torch.float32, None)
    rows = torch.arange(n * dim, dtype=torch.float32, device="cuda").reshape(n, dim)
    wm.scatter(rows, torch.arange(n, dtype=torch.int64, device="cuda"))
    torch.cuda.synchronize()
    dev_view, start = wm.get_local_tensor(host_view=False)   # device view, also for host-located memory
    assert start == 0 and dev_view.is_cuda and torch.equal(dev_view, rows)
    if loc == "cpu":
        host_view, _ = wm.get_local_tensor(host_view=True)
        assert not host_view.is_cuda and torch.equal(host_view, rows.cpu())
        host_view[5, 3] = -1.0     # writes through the host view are seen by a gather
        got = wm.gather(torch.tensor([5], device="cuda"))
        torch.cuda.synchronize()
        assert got[0, 3].item() == -1.0
    else:
        with pytest.raises(ValueError):
            wm.get_local_tensor(host_view=True)
    if mt == "continuous" or (mt == "chunked" and loc == "cpu"):
        g, off = wm.get_global_tensor(host_view=(loc == "cpu"))
        assert off == 0 and tuple(g.shape) == (n, dim)
    if mt != "distributed":
        views, offs = wm.get_all_chunked_tensor(host_view=(loc == "cpu"))
        assert offs == [0] and tuple(views[0].shape) == (n, dim)
    sub = wm.get_sub_tensor([10, 2], [20, 9])
    assert sub.shape == (10, 7) and sub.stride() == (dim, 1) and sub.storage_offset() == 10 * dim + 2
    wgth.destroy_wholememory_tensor(sub)
    wgth.destroy_wholememory_tensor(wm)


def test_wrapped_tensor_handles_are_cached_per_thread_and_outlive_the_cache(wm_lib):
    """wrap_torch_tensor keeps the wholememory_tensor_t of (pointer, shape, strides, dtype) per thread (an op wraps its index and
    output tensors on every call): the same tensor wrapped twice yields the same handle, a different view a different one, a
    wrapper that is still alive keeps its handle when the cache is emptied by 600 other wraps, and the description the library
    reads back is the tensor's. Another thread has its own cache. No GPU needed: wrapping never touches the memory."""
    import ctypes as C
    import threading
    import torch
    from wholegraph_amd import binding as wmb
    from wholegraph_amd.torch import wholegraph_env as E
    t = torch.zeros((12, 5), dtype=torch.float32)
    w1, w2 = E.wrap_torch_tensor(t), E.wrap_torch_tensor(t)
    assert w1.handle.value == w2.handle.value
    assert E.wrap_torch_tensor(t[:6]).handle.value != w1.handle.value            # same pointer, another shape
    assert E.wrap_torch_tensor(t[1:]).handle.value != w1.handle.value            # same shape class, another pointer

    def described(w):
        d = wm_lib.wholememory_tensor_get_tensor_description(w.handle).contents
        return d.dim, d.sizes[0], d.sizes[1], d.strides[0], d.dtype
    assert described(w1) == (2, 12, 5, 5, wmb.DT_FLOAT)
    keep = [torch.zeros(3 + (i % 11), 2 + i // 11) for i in range(600)]          # > the cache's 512 entries: it starts over
    for k in keep:
        E.wrap_torch_tensor(k)
    assert len(E._tls.handles) < 600
    assert described(w1) == (2, 12, 5, 5, wmb.DT_FLOAT)                          # w1's handle was not destroyed under it
    w3 = E.wrap_torch_tensor(t)
    assert described(w3) == (2, 12, 5, 5, wmb.DT_FLOAT)
    other = {}
    th = threading.Thread(target=lambda: other.setdefault("h", E.wrap_torch_tensor(t).handle.value))
    th.start(); th.join()
    assert other["h"] != w3.handle.value
    assert isinstance(E.get_stream(), int)
