"""wholememory_tensor_* host logic against the known answers the reference's own tests hold:
cpp/tests/wholememory/wholememory_tensor_tests.cpp:48-120 (SubTensorTest) and
python/.../tests/pylibwholegraph/test_wholememory_tensor.py:39-85 (array / matrix sub-tensor cases).
Pointer-backed tensors need no device, so these run on the CPU box; the handle-backed variants run with -m gpu."""
import ctypes as C

import pytest

from wholegraph_amd import binding as B


def desc_of(lib, t):
    d = lib.wholememory_tensor_get_tensor_description(t).contents
    return d.dim, d.dtype, d.storage_offset, [d.sizes[i] for i in range(d.dim)], [d.strides[i] for i in range(d.dim)]


def sub(lib, t, starts, ends):
    n = len(starts)
    out = C.c_void_p()
    rc = lib.wholememory_tensor_get_subtensor(t, (C.c_int64 * n)(*starts), (C.c_int64 * n)(*ends), C.byref(out))
    return rc, out


def check_reference_subtensor_kat(lib, root, row, col):
    # wholememory_tensor_tests.cpp:77-108
    rc, s0 = sub(lib, root, [1, 10], [-1, 100])
    assert rc == 0
    assert desc_of(lib, s0) == (2, B.DT_INT, col * 1 + 10, [row - 1, 90], [256, 1])
    rc, s1 = sub(lib, s0, [2, -1], [10000, 80])
    assert rc == 0
    assert desc_of(lib, s1) == (2, B.DT_INT, col * 3 + 10, [10000 - 2, 80], [256, 1])
    assert lib.wholememory_tensor_get_root(s1) == root.value
    assert lib.wholememory_destroy_tensor(s0) == 0
    assert lib.wholememory_destroy_tensor(s1) == 0


def test_subtensor_known_answers_pointer_backed(wm_lib):
    row, col = 256 * 128, 256
    before = wm_lib.get_wholememory_tensor_count()
    d = B.make_tensor_desc([row, col], B.DT_INT, [col, 1])
    buf = (C.c_char * 64)()
    root = C.c_void_p()
    assert wm_lib.wholememory_make_tensor_from_pointer(C.byref(root), C.cast(buf, C.c_void_p), C.byref(d)) == 0
    assert not wm_lib.wholememory_tensor_has_handle(root)
    check_reference_subtensor_kat(wm_lib, root, row, col)
    # data pointer = storage + storage_offset * elt_size (wholememory_tensor.cpp:274-294)
    rc, s = sub(wm_lib, root, [3, 7], [-1, -1])
    assert rc == 0
    assert wm_lib.wholememory_tensor_get_data_pointer(s) == C.addressof(buf) + (3 * col + 7) * 4
    assert wm_lib.wholememory_destroy_tensor(s) == 0
    # python reference cases (test_wholememory_tensor.py:39-85)
    size = 128 * 1024 * 1024 + 1
    a = B.make_tensor_desc([size], B.DT_FLOAT)
    arr = C.c_void_p()
    assert wm_lib.wholememory_make_tensor_from_pointer(C.byref(arr), C.cast(buf, C.c_void_p), C.byref(a)) == 0
    rc, sa = sub(wm_lib, arr, [size // 4], [-1])
    assert rc == 0 and desc_of(wm_lib, sa) == (1, B.DT_FLOAT, size // 4, [size - size // 4], [1])
    m0, m1 = 1024 * 1024 + 131, 256
    md = B.make_tensor_desc([m0, m1], B.DT_FLOAT)
    mat = C.c_void_p()
    assert wm_lib.wholememory_make_tensor_from_pointer(C.byref(mat), C.cast(buf, C.c_void_p), C.byref(md)) == 0
    rc, sm = sub(wm_lib, mat, [m0 // 3, m1 // 5], [-1, m1 // 5 * 3])
    assert rc == 0
    assert desc_of(wm_lib, sm) == (2, B.DT_FLOAT, m1 // 5 + m0 // 3 * m1, [m0 - m0 // 3, m1 // 5 * 3 - m1 // 5], [m1, 1])
    # invalid ranges (wholememory_tensor.cpp:439-444)
    for st, en in [([5, 0], [5, -1]), ([m0, 0], [-1, -1]), ([0, 0], [0, 4]), ([0, 9], [4, 3])]:
        rc, bad = sub(wm_lib, mat, st, en)
        assert rc == 6
    # entry partition of a plain tensor: one "rank" holding every row (wholememory_tensor.cpp:322-325)
    offs = (C.c_size_t * 2)()
    assert wm_lib.wholememory_tensor_get_entry_offsets(offs, mat) == 0 and list(offs) == [0, m0]
    cnt, start = C.c_size_t(), C.c_size_t(7)
    assert wm_lib.wholememory_tensor_get_local_entry_count(C.byref(cnt), mat) == 0 and cnt.value == m0
    assert wm_lib.wholememory_tensor_get_local_entry_start(C.byref(start), mat) == 0 and start.value == 0
    for t in (sa, sm, arr, mat, root):
        assert wm_lib.wholememory_destroy_tensor(t) == 0
    assert wm_lib.get_wholememory_tensor_count() == before  # no leaked tensor objects


def test_create_tensor_argument_checks(wm_lib):
    """wholememory_create_tensor rejects bad descriptions before touching the device
    (wholememory_tensor.cpp:59-82)."""
    t = C.c_void_p()
    comm = C.c_void_p(1)  # never dereferenced: the checks fail first
    for d in (B.make_tensor_desc([4, 4, 4], B.DT_FLOAT), B.make_tensor_desc([4, 4], B.DT_FLOAT, [4, 2]),
              B.make_tensor_desc([4, 4], B.DT_UNKNOWN), B.make_tensor_desc([4, 4], B.DT_FLOAT, [4, 1], 3)):
        assert wm_lib.wholememory_create_tensor(C.byref(t), C.byref(d), comm, B.MT_CHUNKED, B.ML_DEVICE, None) == 6
    assert wm_lib.wholememory_create_tensor(None, None, comm, B.MT_CHUNKED, B.ML_DEVICE, None) == 6


def test_equal_partition_plan_and_optimizer_objects(wm_lib):
    v = C.c_size_t()
    for n, w, exp in [(1003, 8, 126), (8, 8, 1), (0, 4, 0), (1000000007, 8, 125000001)]:
        assert wm_lib.wholememory_equal_entry_partition_plan(C.byref(v), n, w) == 0 and v.value == exp
    # optimizer objects: parameter names per type (embedding_optimizer.cpp:114-119,176-190,300-307,401-410)
    for typ, good, bad in [(B.OPT_SGD, ["weight_decay"], ["epsilon", "alpha"]),
                           (B.OPT_LAZY_ADAM, ["weight_decay", "epsilon", "beta1", "beta2", "adam_w"], ["alpha"]),
                           (B.OPT_ADAGRAD, ["weight_decay", "epsilon"], ["beta1"]),
                           (B.OPT_RMSPROP, ["weight_decay", "epsilon", "alpha"], ["beta2"])]:
        o = C.c_void_p()
        assert wm_lib.wholememory_create_embedding_optimizer(C.byref(o), typ) == 0
        val = C.c_float(0.5)
        for g in good:
            assert wm_lib.wholememory_optimizer_set_parameter(o, g.encode(), C.byref(val)) == 0
        for b in bad:
            assert wm_lib.wholememory_optimizer_set_parameter(o, b.encode(), C.byref(val)) == 6
        # extension: the order of the duplicate sum is a parameter of every optimizer (-1 default of the dtype, 0 ordered, 1 tree)
        for fold in (-1.0, 0.0, 1.0):
            assert wm_lib.wholememory_optimizer_set_parameter(o, b"grad_fold", C.byref(C.c_float(fold))) == 0
        wm_lib.wholememory_destroy_embedding_optimizer(o)
    o = C.c_void_p()
    assert wm_lib.wholememory_create_embedding_optimizer(C.byref(o), B.OPT_NONE) == 2
    # cache policy objects exist, ratio range checked (embedding.cpp:908-912)
    p = C.c_void_p()
    assert wm_lib.wholememory_create_embedding_cache_policy(C.byref(p), None, B.MT_CHUNKED, B.ML_DEVICE, B.AT_READONLY, 0.5) == 0
    assert wm_lib.wholememory_destroy_embedding_cache_policy(p) == 0
    assert wm_lib.wholememory_create_embedding_cache_policy(C.byref(p), None, B.MT_CHUNKED, B.ML_DEVICE, B.AT_READONLY, 1.5) == 7


@pytest.mark.gpu
def test_subtensor_known_answers_handle_backed(gpu_env):
    """the reference test itself: CONTINUOUS HOST tensor of int32, fill through the global pointer, take views."""
    import torch
    lib = B.lib()
    row, col = 256 * 128, 256
    d = B.make_tensor_desc([row, col], B.DT_INT, [col, 1])
    root = C.c_void_p()
    assert lib.wholememory_create_tensor(C.byref(root), C.byref(d), gpu_env.wmb_comm, B.MT_CONTINUOUS, B.ML_HOST, None) == 0
    h = C.c_void_p(lib.wholememory_tensor_get_memory_handle(root))
    gp = C.c_void_p()
    assert lib.wholememory_get_global_pointer(C.byref(gp), h) == 0
    host = torch.frombuffer((C.c_char * (row * col * 4)).from_address(gp.value), dtype=torch.int32)
    host.copy_(torch.arange(row * col, dtype=torch.int32))
    check_reference_subtensor_kat(lib, root, row, col)
    assert torch.equal(host, torch.arange(row * col, dtype=torch.int32))
    # partition queries on a handle-backed tensor
    offs = (C.c_size_t * 2)()
    assert lib.wholememory_tensor_get_entry_offsets(offs, root) == 0 and list(offs) == [0, row]
    assert lib.wholememory_get_total_size(h) == row * col * 4 and lib.wholememory_get_data_granularity(h) == col * 4
    assert lib.wholememory_destroy_tensor(root) == 0


@pytest.mark.gpu
def test_env_test_op(gpu_env):
    """wholememory_env_test_op (reference wholememory_test_op.cu): out[i, :] = T(float(i)) + input[:], delivered to a
    fixed tensor and to device / pinned / host tensors allocated through the env output functions."""
    import ctypes as C
    import torch
    from wholegraph_amd import binding as wmb
    from wholegraph_amd.torch.wholegraph_env import (TorchMemoryContext, get_stream, get_wholegraph_env_fns,
                                                     wrap_torch_tensor)
    for dt in (torch.float32, torch.int64, torch.float16):
        inp = (torch.arange(37, device="cuda") % 5).to(dt)
        n = 11
        fixed = torch.zeros((n, 37), dtype=dt, device="cuda")
        ctxs = [TorchMemoryContext() for _ in range(3)]
        wi, wf = wrap_torch_tensor(inp), wrap_torch_tensor(fixed)
        wmb.check(wmb.lib().wholememory_env_test_op(wi.handle, wf.handle, *[C.c_void_p(c.get_c_context()) for c in ctxs], n,
                                                    get_wholegraph_env_fns(), C.c_void_p(get_stream())))
        want = torch.arange(n, device="cuda").to(dt).unsqueeze(1) + inp.unsqueeze(0)
        assert torch.equal(fixed, want)
        dev, pinned, host = [c.get_tensor() for c in ctxs]
        assert dev.is_cuda and not pinned.is_cuda and not host.is_cuda
        for t in (dev, pinned, host):
            assert torch.equal(t.cuda(), want)


def test_wrapper_can_be_passed_as_a_temporary(wm_lib):
    """`wrap_torch_tensor(t)` owns its C handle; passing the wrapper itself (not `.handle`) keeps a temporary alive for
    the duration of the call."""
    import torch
    from wholegraph_amd import binding as wmb
    from wholegraph_amd.torch.wholegraph_env import wrap_torch_tensor
    t = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    desc = wm_lib.wholememory_tensor_get_tensor_description(wrap_torch_tensor(t))
    assert bool(desc)
    w = wrap_torch_tensor(t)
    td = C.cast(wm_lib.wholememory_tensor_get_tensor_description(w), C.POINTER(wmb.TensorDescription)).contents
    assert td.dim == 2 and list(td.sizes[:2]) == [3, 4] and list(td.strides[:2]) == [4, 1]


@pytest.mark.gpu
def test_native_env_functions_are_in_use(gpu_env):
    """On a GPU box the default env table takes scratch memory natively from torch's caching allocator
    (libwg_torch_env.so); WG_NATIVE_ENV=0 would select the all-Python table."""
    import os
    import torch
    from wholegraph_amd.torch import wholegraph_env as we
    we.get_wholegraph_env_fns()
    assert os.path.exists(os.path.join(os.path.dirname(B.LIB_PATH), "libwg_torch_env.so"))
    assert isinstance(we._default_env, we._NativeEnvTable)
    env = we._default_env.env
    # a scratch allocation through the table comes out of torch's allocator: reserved memory does not grow on reuse
    ctx = C.c_void_p()
    desc = B.make_tensor_desc([1 << 20], B.DT_FLOAT)
    env.temporary_fns.create_memory_context_fn(C.byref(ctx), None)
    p1 = env.temporary_fns.malloc_fn(C.byref(desc), B.MA_DEVICE, ctx, None)
    assert p1
    env.temporary_fns.free_fn(ctx, None)
    reserved = torch.cuda.memory_reserved()
    p2 = env.temporary_fns.malloc_fn(C.byref(desc), B.MA_DEVICE, ctx, None)
    assert p2 and torch.cuda.memory_reserved() == reserved
    env.temporary_fns.destroy_memory_context_fn(ctx, None)
