"""Pins the CPU oracle (oracle/wm_oracle.c) against independent statements of the reference's own test
rules, written here in numpy / pure Python (NOT through the oracle):

 * closed-form tables + exact-compare rule of the reference gather/scatter tests
   (cpp/tests/wholememory_ops/embedding_test_utils.cu:197-238,401-431,467-520);
 * the Python reference oracle value(r, c) = float(int32(r) + c)
   (python/.../tests/wholegraph_torch/ops/test_wholegraph_gather_scatter.py:26-37);
 * the reference tests' host CPUOptimizer + first-seen-order dedup, tolerance 1e-5
   (cpp/tests/wholememory_ops/wholememory_embedding_gradient_apply_tests.cu:169-371,437-466,481-501);
 * partition plans (memory_handle.cpp:1618-1635; host_random_partition embedding_test_utils.cu:531-546;
   python random_partition test_comm.py:188-195) via the committed golden fixtures.
"""
import json
import os

import numpy as np
import pytest

import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def closed_form(np_dtype, rows, dim):
    """embedding_test_utils.cu:197-238: T(r & (2^(M+1)-1)), column independent; ints: plain cast."""
    r = np.arange(rows, dtype=np.int64)
    mant = {np.float32: 23, np.float16: 10, np.float64: 52}
    if np_dtype in mant:
        v = (r & ((1 << (mant[np_dtype] + 1)) - 1)).astype(np.float32 if np_dtype != np.float64 else np.float64)
        v = v.astype(np_dtype)
    else:
        v = r.astype(np_dtype)
    return np.repeat(v[:, None], dim, axis=1)


@pytest.mark.parametrize("dt", [np.float32, np.float16, np.float64, np.int8, np.int16, np.int32, np.int64])
def test_closed_form_fill(dt):
    got = oracle.fill_closed_form(dt, 0, 5000, 7)
    assert got.tobytes() == closed_form(dt, 5000, 7).tobytes()
    got2 = oracle.fill_closed_form(dt, 4090, 20, 3, stride=4)
    assert np.array_equal(got2[:, :3], closed_form(dt, 4110, 3)[4090:])


@pytest.mark.parametrize("tdt,odt", [(np.float32, np.float32), (np.float16, np.float32), (np.float32, np.float16),
                                     (np.float64, np.float16), (np.float16, np.float64), (np.int64, np.int8),
                                     (np.int8, np.int64), (np.int32, np.int16)])
@pytest.mark.parametrize("idt", [np.int32, np.int64])
@pytest.mark.parametrize("world", [1, 3, 8])
def test_gather_matches_reference_test_rule(tdt, odt, idt, world):
    """expected output of the reference test = closed form generated in the TABLE dtype, then cast to the
    output dtype (device_get_expected_embedding + device_matrix_type_cast, embedding_test_utils.cu:401-431)."""
    rng = np.random.default_rng(5)
    n_rows, dim, n = 1003, 11, 700
    full = closed_form(tdt, n_rows, dim)
    tab = oracle.ShardedTable.from_full(full, world)
    idx = rng.integers(0, n_rows, n).astype(idt)
    idx[::13] = -1
    out = np.full((n, dim), 9, dtype=odt)
    oracle.gather(tab, idx, out)
    valid = idx >= 0
    src = full[idx[valid].astype(np.int64)]
    if tdt == np.float64 and odt == np.float16:
        exp_valid = src.astype(np.float32).astype(np.float16)  # double -> float -> half (two roundings)
    else:
        exp_valid = src.astype(odt)
    assert out[valid].tobytes() == exp_valid.tobytes()
    assert np.all(out[~valid] == 9)  # negative ids: row untouched (gather_scatter_func.cuh:296)


def test_python_reference_int_embedding_rule():
    """test_wholegraph_gather_scatter.py:26-37: value(r, c) = float(int32(r) + c), scatter rank-strided
    rows then gather; here all 'ranks' are simulated by the sharded oracle table."""
    world, dim = 4, 16
    n_rows = 1024 * world + 3
    tab = oracle.ShardedTable.from_full(np.zeros((n_rows, dim), np.float32), world)
    for rank in range(world):
        ids = np.arange(rank, n_rows, world, dtype=np.int64)
        rows = (ids.astype(np.int32)[:, None] + np.arange(dim, dtype=np.int32)).astype(np.float32)
        oracle.scatter(rows, ids, tab)
    gidx = np.random.default_rng(42).integers(0, n_rows, 5001).astype(np.int32)
    out = np.zeros((5001, dim), np.float32)
    oracle.gather(tab, gidx, out)
    assert np.array_equal(out, (gidx[:, None] + np.arange(dim, dtype=np.int32)).astype(np.float32))


def test_half_and_bf16_conversions_match_numpy_and_torch():
    L = oracle.lib()
    rng = np.random.default_rng(0)
    f = (rng.standard_normal(20000) * rng.choice([1e-8, 1e-6, 1e-4, 1, 300, 7e4], 20000)).astype(np.float32)
    with np.errstate(over="ignore"):
        ref = f.astype(np.float16).view(np.uint16)
    mine = np.array([L.wmo_float_to_half(float(x)) for x in f], dtype=np.uint16)
    assert np.array_equal(mine, ref)
    allh = np.arange(65536, dtype=np.uint16)
    back = np.array([L.wmo_half_to_float(int(h)) for h in allh], dtype=np.float32)
    refb = allh.view(np.float16).astype(np.float32)
    assert np.array_equal(back[~np.isnan(refb)], refb[~np.isnan(refb)])
    import torch
    tb = torch.from_numpy(f).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    mb = np.array([L.wmo_float_to_bf16(float(x)) for x in f], dtype=np.uint16)
    assert np.array_equal(mb, tb)


def test_sort_ids_is_stable_unsigned_sort():
    """exchange_ids_nccl_func.cu:60-92: keys reinterpreted unsigned (negatives last), payload iota, stable."""
    rng = np.random.default_rng(3)
    for dt in (np.int32, np.int64):
        idx = rng.integers(0, 50, 4000).astype(dt)
        idx[::17] = -1
        idx[5::31] = -7
        s, raw = oracle.sort_ids(idx)
        key = idx.astype(np.int64).view(np.uint64) if dt == np.int64 else idx.view(np.uint32).astype(np.uint64)
        order = np.argsort(key, kind="stable")
        assert np.array_equal(raw, order.astype(np.int64))
        assert np.array_equal(s, idx[order])


def test_bucket_counts_rule():
    """bucket_ids_func.cu:31-87 incl. empty ranks and negatives."""
    offs = np.array([0, 3, 3, 10, 10, 12], dtype=np.uint64)  # ranks 1 and 3 are empty
    idx = np.array([0, 2, 3, 9, 10, 11, -1, 5, 5, -4, 2], dtype=np.int64)
    counts = oracle.bucket_counts(idx, offs)
    assert counts.tolist() == [3, 0, 4, 0, 2]


def cpu_optimizer_reference(kind, table, idx, grads, lr, steps, params):
    """Independent restatement of the reference tests' CPUOptimizer flow
    (wholememory_embedding_gradient_apply_tests.cu:169-371,437-466): per step, first-seen-order dedup with
    += accumulation, then per-row update in float32 scalar arithmetic."""
    f = np.float32
    wd, eps = f(params.get("weight_decay", 0.0)), f(params.get("epsilon", 1e-8))
    alpha, b1, b2 = f(params.get("alpha", 0.99)), f(params.get("beta1", 0.9)), f(params.get("beta2", 0.999))
    adam_w = params.get("adam_w", 0.0) > 0.5
    lr = f(lr)
    n, dim = table.shape
    st0, st1 = np.zeros_like(table), np.zeros_like(table)
    pe0, pe1 = np.ones(n, f), np.ones(n, f)
    for _ in range(steps):
        first, acc = {}, []
        for i, ix in enumerate(idx):
            ix = int(ix)
            if ix not in first:
                first[ix] = len(acc)
                acc.append(grads[i].astype(f).copy())
            else:
                acc[first[ix]] = (acc[first[ix]] + grads[i]).astype(f)
        for ix, k in first.items():
            g, e = acc[k].astype(f), table[ix].astype(f)
            if kind == "sgd":
                g = g + wd * e
                e = e - lr * g
            elif kind == "adam":
                pe0[ix] = pe0[ix] * b1
                pe1[ix] = pe1[ix] * b2
                if adam_w:
                    e = e - lr * wd * e
                else:
                    g = g + wd * e
                m = b1 * st0[ix] + (f(1) - b1) * g
                v = b2 * st1[ix] + (f(1) - b2) * g * g
                mhat = m / (f(1) - pe0[ix])
                vhat = v / (f(1) - pe1[ix])
                e = e - lr * mhat / (np.sqrt(vhat, dtype=f) + eps)
                st0[ix], st1[ix] = m, v
            elif kind == "adagrad":
                g = g + wd * e
                s = st0[ix] + g * g
                e = e - lr * g / (np.sqrt(s, dtype=f) + eps)
                st0[ix] = s
            elif kind == "rmsprop":
                g = g + wd * e
                v = alpha * st0[ix] + (f(1) - alpha) * g * g
                e = e - lr * g / (np.sqrt(v, dtype=f) + eps)
                st0[ix] = v
            table[ix] = e.astype(f)
    return table


@pytest.mark.parametrize("kind,params", [("sgd", {}), ("sgd", {"weight_decay": 0.1}), ("adam", {}),
                                         ("adam", {"weight_decay": 0.05, "adam_w": 1.0}),
                                         ("adagrad", {"weight_decay": 0.01}), ("rmsprop", {"alpha": 0.9})])
@pytest.mark.parametrize("world", [1, 3])
def test_gradient_apply_matches_cpu_optimizer_rule(kind, params, world):
    """oracle.gradient_apply (exchange -> sorted-order dedup -> step) vs the reference tests' CPUOptimizer with
    first-seen-order dedup: duplicates are summed in different orders, so the reference's own tolerance
    applies (atol = rtol = 1e-5, host_expect_all_close :481-501)."""
    rng = np.random.default_rng(11)
    n_rows, dim, steps = 301, 13, 3
    table0 = rng.standard_normal((n_rows, dim)).astype(np.float32)
    rank_idx = [rng.integers(0, n_rows, 120).astype(np.int64) for _ in range(world)]
    rank_grads = [rng.standard_normal((120, dim)).astype(np.float32) for _ in range(world)]
    tab = oracle.ShardedTable.from_full(table0.copy(), world)
    opts = [oracle.Optimizer(kind, int(tab.entry_offsets[r + 1] - tab.entry_offsets[r]), dim, **params)
            for r in range(world)]
    for _ in range(steps):
        oracle.gradient_apply(tab, opts, rank_idx, rank_grads, 0.1)
    got = np.concatenate([tab.shards[r][: int(tab.entry_offsets[r + 1] - tab.entry_offsets[r])] for r in range(world)])
    ref = cpu_optimizer_reference(kind, table0.copy(), np.concatenate(rank_idx), np.concatenate(rank_grads), 0.1, steps,
                                  params)
    aerr = np.abs(got - ref)
    rerr = aerr / np.maximum(np.maximum(np.abs(got), np.abs(ref)), 1e-30)
    assert np.all((aerr < 1e-5) | (rerr < 1e-5))


def test_dedup_sums_in_receive_order():
    """exchange_embeddings_nccl_func.cu:76-103: first occurrence copied, later ones added one by one."""
    ids = np.array([7, 3, 7, 7, 3, 9], dtype=np.int64)
    g = np.array([[1e8], [1.0], [1.0], [-1e8], [2.0], [5.0]], dtype=np.float32)
    u, dg = oracle.dedup_grads(ids, g)
    assert u.tolist() == [3, 7, 9]
    f = np.float32
    assert dg[:, 0].tolist() == [float(f(1.0) + f(2.0)), float((f(1e8) + f(1.0)) + f(-1e8)), 5.0]


def test_partition_golden():
    with open(os.path.join(GOLDEN, "partitions.json")) as fh:
        cases = json.load(fh)
    for c in cases["equal"]:
        sizes, offs = oracle.equal_partition(c["n"], c["world"])
        assert sizes.tolist() == c["sizes"] and offs.tolist() == c["offsets"]
    for c in cases["host_random_partition"] + cases["python_random_partition"]:
        offs, same = oracle.custom_partition(c["sizes"])
        assert offs.tolist() == c["offsets"] and same == c["same_chunk"]
        assert sum(c["sizes"]) == c["n"]


def test_round_robin_rules():
    """embedding.cpp:467-484 padded row count and map_indices_func.cu:34-43 remap (quirk included)."""
    assert oracle.round_robin_total_entries(1003, 4, 16) == 4 * (1003 // 64 * 16 + 16)
    assert oracle.round_robin_total_entries(1000, 8, 0) == 1000
    assert oracle.round_robin_total_entries(130, 4, 32) == 4 * (32 + 2)
    idx = np.arange(0, 200, dtype=np.int64)
    got = oracle.round_robin_map(idx, entry_start=1000, world=4, rr=8)
    exp = 1000 + 8 * ((idx // 8) // 4) + idx % 8
    assert np.array_equal(got, exp)


def test_align_embedding_dim():
    """embedding.cpp:43-50"""
    assert [oracle.align_embedding_dim(d, 4) for d in (1, 4, 11, 127, 128, 129)] == [4, 4, 12, 128, 128, 132]
    assert [oracle.align_embedding_dim(d, 2) for d in (1, 8, 9, 256)] == [8, 8, 16, 256]
    assert oracle.align_embedding_dim(5, 8) == 6
