"""The ORDERED fold of very long duplicate runs through their dense transposed copies (csrc/kernels/long_dense.cuh, optim.hip:
step_dense_kernel; round 6) — reference exchange_embeddings_nccl_func.cu:76-103 (DedupIndiceAndGradientsKernel: the gradient rows
of one id summed one by one in receive order) + embedding_optimizer_func.cu (the four update rules).

A batch whose hottest ids repeat thousands of times (and one a few hundred times, and a tail of runs of 1-6 rows) on RANDOM fp32
gradients, where the summation order shows in the last bits: with the threshold lowered (WM_DENSE_FOLD_MIN) the hottest runs go
through copy + fold in one launch, the next ones stay step_long4_kernel's — because they are below the threshold or because the
dense buffer (3 n / 8 rows per call) is full —, the rest is the tile kernel's. Every path must give the oracle's bits: table,
per-element states, per-row beta powers; and the same bits with the dense route switched off (WM_DENSE_FOLD=0). Row shapes: the
tile kernel's 128 floats, 100 floats (25 sixteen-byte pieces: a last 8-column slice of 4 columns), 36 floats (short rows), 200 floats
and 256 floats (the widest rows the route takes: the long-run kernel's resident round then has just 16 workgroups per slice to spare). Two calls in a row on the same workspace addresses (a stale line of the first call's dense copy in
another XCD's L2 would show in the second)."""
import ctypes as C

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

OPTS = [("sgd", 1, {"weight_decay": 0.05}), ("adam", 2, {"weight_decay": 0.01})]


def _env():
    from wholegraph_amd.torch.wholegraph_env import get_wholegraph_env_fns, get_stream
    return get_wholegraph_env_fns(), C.c_void_p(get_stream())


def _batch(rng, local_rows, local_off, n_recv, idt):
    ids = (local_off + rng.integers(0, local_rows, n_recv)).astype(idt)
    hot = local_off + rng.choice(local_rows, 4, replace=False)
    u = rng.random(n_recv)
    ids[u < 0.20] = hot[0]                       # ~20 % of the batch, ~12 k rows: the dense route
    ids[(u >= 0.20) & (u < 0.32)] = hot[1]        # ~12 %: dense while the buffer lasts (3 n / 8 rows per call: not all three fit)
    ids[(u >= 0.32) & (u < 0.40)] = hot[2]        # ~8 %: dense or step_long4_kernel, by the threshold and by the room left
    ids[(u >= 0.40) & (u < 0.405)] = hot[3]       # ~0.5 %, ~300 rows
    return ids


@pytest.mark.parametrize("kind,code,params", OPTS, ids=lambda x: str(x))
@pytest.mark.parametrize("dim,idt,dense_min,n_recv", [(128, np.int64, 300, 60007), (100, np.int32, 300, 60007), (36, np.int64, 2000, 60007),
                                                      (256, np.int64, 1000, 60007), (200, np.int32, 1000, 60007),
                                                      # at least 65536 ids: the split sort (its control words hold the counters),
                                                      # whose buckets these hot ids overflow: the gated generic path / the adaptive route
                                                      (128, np.int64, 2000, 200003)])
def test_dense_fold_bit_exact(gpu_env, knobs, kind, code, params, dim, idt, dense_min, n_recv):
    import torch
    from wholegraph_amd import binding as wmb
    rng = np.random.default_rng(dim * 7 + code)
    local_rows, local_off = 20011, 333
    stride = int(oracle.align_embedding_dim(dim, 4))
    table = np.zeros((local_rows, stride), np.float32)
    table[:, :dim] = rng.standard_normal((local_rows, dim)).astype(np.float32)
    arr_p = dict(weight_decay=0.0, epsilon=1e-8, beta1=0.9, beta2=0.999, alpha=0.99, adam_w=0.0)
    arr_p.update(params)
    arr = (C.c_float * 6)(arr_p["weight_decay"], arr_p["epsilon"], arr_p["beta1"], arr_p["beta2"], arr_p["alpha"], arr_p["adam_w"])
    env, stream = _env()
    results = {}
    for route in ("dense", "off"):
        if route == "dense":
            knobs.unset("WM_DENSE_FOLD")
            knobs.set("WM_DENSE_FOLD_MIN", dense_min)
        else:
            knobs.set("WM_DENSE_FOLD", 0)
        rng_b = np.random.default_rng(dim * 13 + code)
        ref_opt = oracle.Optimizer(kind, local_rows, stride, **params)
        ref_table = table.copy()
        d_table = torch.from_numpy(table.copy()).cuda()
        d_pe = d_pr = None
        if kind == "adam":
            d_pe = torch.zeros((local_rows, 2 * stride), device="cuda")
            d_pr = torch.ones((local_rows, 2), device="cuda")
        for step in range(2):
            ids = _batch(rng_b, local_rows, local_off, n_recv, idt)
            grads = rng_b.standard_normal((n_recv, dim)).astype(np.float32)
            d_ids, d_grads = torch.from_numpy(ids).cuda(), torch.from_numpy(grads).cuda()
            uniq, dg = oracle.dedup_grads(ids, grads)
            nu = C.c_int64(-1)
            wmb.check(wmb.lib().wholememory_ext_dedup_apply(
                d_ids.data_ptr(), wmb.DT_INT if idt == np.int32 else wmb.DT_INT64, n_recv, d_grads.data_ptr(), dim, dim,
                d_table.data_ptr(), stride, local_off, local_rows, code, arr, 0.03,
                d_pe.data_ptr() if d_pe is not None else None, d_pr.data_ptr() if d_pr is not None else None, C.byref(nu),
                env, stream))
            torch.cuda.synchronize()
            assert nu.value == len(uniq)
            ref_opt.step(uniq, dg, ref_table, stride, local_off, dim, 0.03)
            assert d_table.cpu().numpy().tobytes() == ref_table.tobytes(), "%s %s step %d: table differs from the oracle" % (kind, route, step)
        if kind == "adam":
            assert d_pe.cpu().numpy().tobytes() == ref_opt.per_element.tobytes()
            assert d_pr.cpu().numpy().tobytes() == ref_opt.per_row.tobytes()
        results[route] = d_table.cpu().numpy().tobytes()
    assert results["dense"] == results["off"]


def test_dense_fold_takes_the_hot_runs(gpu_env, knobs):
    """the route is really taken — and not past its buffer: the count of runs the last step folded through a dense copy"""
    import torch
    from wholegraph_amd import binding as wmb
    lib = wmb.lib()
    rng = np.random.default_rng(5)
    local_rows, n_recv, dim = 20011, 60007, 128
    ids = _batch(rng, local_rows, 0, n_recv, np.int64)
    grads = rng.standard_normal((n_recv, dim)).astype(np.float32)
    d_ids, d_grads = torch.from_numpy(ids).cuda(), torch.from_numpy(grads).cuda()
    arr = (C.c_float * 6)(0.0, 1e-8, 0.9, 0.999, 0.99, 0.0)
    env, stream = _env()

    def step():
        d_table = torch.zeros((local_rows, dim), device="cuda")
        nu = C.c_int64(-1)
        wmb.check(lib.wholememory_ext_dedup_apply(d_ids.data_ptr(), wmb.DT_INT64, n_recv, d_grads.data_ptr(), dim, dim,
                                                  d_table.data_ptr(), dim, 0, local_rows, 1, arr, 1.0, None, None, C.byref(nu), env, stream))
        torch.cuda.synchronize()
        return lib.wholememory_ext_dense_fold_last()

    knobs.unset("WM_DENSE_FOLD")
    knobs.set("WM_DENSE_FOLD_MIN", 300)
    # the buffer holds 3 n / 8 = 22.5 k rows; the runs are ~12 k, ~7.2 k, ~4.8 k and ~300 rows: not all of them fit.
    # (the room for the copies is only part of a step's scratch while the device's recent steps listed long runs: the first call
    # after a series without any runs them through step_long4_kernel and tells the next one)
    step()
    took = step()
    assert 2 <= took <= 3, took
    knobs.set("WM_DENSE_FOLD_MIN", 10000)
    step()
    assert step() == 1
    knobs.set("WM_DENSE_FOLD", 0)
    assert step() == 0
