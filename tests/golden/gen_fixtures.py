"""Generates the frozen hot-path fixtures under tests/golden/ — numpy only: nothing here imports the oracle or the
product, so the files are a third, independent statement of the reference's rules that BOTH are compared with
(tests/test_golden_fixtures.py on the CPU for the oracle, tests/test_golden_fixtures_gpu.py on the GPU for the HIP
path through the C ABI).

 bucketing.npz       owner bucketing + the exchange's id grouping: counts per owner (negative ids skipped,
                     bucket_ids_func.cu:51-87), ids grouped by owner in the reference's order = stable UNSIGNED sort of
                     the ids with an iota payload (exchange_ids_nccl_func.cu:42-92: cub::DeviceRadixSort::SortPairs over
                     unsigned keys; negatives land last), raw_indices = the payload. Cases: int32 / int64 ids, 4 k ids,
                     duplicates, negatives, empty ranks, custom partitions, W in {1, 2, 3, 8}.
 gather_scatter.npz  the reference gather / scatter tests' rule: table value(r, c) = T(r & (2^(M+1) - 1))
                     (embedding_test_utils.cu:197-238), expected output = that closed form IN THE TABLE DTYPE cast to the
                     output dtype (device_get_expected_embedding + device_matrix_type_cast, :401-431), 64 ids into a
                     257-row table, dims {1, 11, 32, 127, 128, 129, 513}, the dtype-cast matrix; plus random-valued tables
                     (so that the f32 -> f16 / f64 -> f16 roundings matter), negative ids (row left untouched,
                     gather_scatter_func.cuh:296), padded strides; scatter: expected table after scattering closed-form
                     rows of the INPUT dtype at distinct ids into a zeroed table.
 optimizers.npz      dedup + optimizer step on a 1000 x 127 fp32 table (stride 128 in memory): duplicates of an id are summed
                     one by one in order of arrival, first occurrence copied (exchange_embeddings_nccl_func.cu:76-103; at
                     one rank arrival order = caller order), then the kernels' statement sequence in fp32 with every *
                     and + rounded on its own (embedding_optimizer_func.cu:212-223 SGD, :385-418 LazyAdam / AdamW,
                     :644-657 AdaGrad, :842-855 RMSProp; the tests' CPUOptimizer is the same statements,
                     wholememory_embedding_gradient_apply_tests.cu:169-371). Two steps of 400 gradient rows; expected
                     table rows / states / beta powers of every touched row after the last step (after each step for SGD);
                     untouched rows must equal the initial table.

Run: python tests/golden/gen_fixtures.py      (deterministic: seeds are fixed, output is byte-stable)
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
F = np.float32


# ---------------------------------------------------------------------------------------------------- bucketing
def bucket_case(rng, idt, n, offsets, neg_every, dup_every):
    total = int(offsets[-1])
    ids = rng.integers(0, max(total, 1), n).astype(idt)
    if dup_every:
        ids[::dup_every] = ids[0]
    if neg_every:
        ids[3::neg_every] = -1
        ids[5::neg_every * 3] = -77
    if total == 0:
        ids[:] = -1
    W = len(offsets) - 1
    counts = np.zeros(W, dtype=np.int64)
    for r in range(W):
        counts[r] = int(np.sum((ids >= 0) & (ids.astype(np.int64) >= int(offsets[r])) & (ids.astype(np.int64) < int(offsets[r + 1]))))
    key = ids.astype(np.int64).view(np.uint64) if idt == np.int64 else ids.view(np.uint32).astype(np.uint64)
    order = np.argsort(key, kind="stable")
    return {"ids": ids, "offsets": np.asarray(offsets, dtype=np.uint64), "counts": counts, "sorted_ids": ids[order],
            "raw_indices": order.astype(np.int64)}


def gen_bucketing():
    rng = np.random.default_rng(20260928)
    out, k = {}, 0
    plans = [
        [0, 5000],                                            # W = 1
        [0, 2500, 5000],                                      # W = 2 equal
        [0, 1667, 3334, 5000],                                # W = 3
        [0, 625, 1250, 1875, 2500, 3125, 3750, 4375, 5000],   # W = 8 equal
        [0, 3, 3, 10, 10, 12],                                # empty ranks 1 and 3 (tiny table, many duplicates)
        [0, 0, 4096, 4096],                                   # empty first and last rank
        [0, 1731, 1900, 4000, 4001, 4001, 4950, 4999, 5000],  # custom partition, one empty
    ]
    for offsets in plans:
        for idt in (np.int32, np.int64):
            for neg_every, dup_every in ((0, 0), (29, 7)):
                c = bucket_case(rng, idt, 4096, offsets, neg_every, dup_every)
                for name, v in c.items():
                    out["c%d_%s" % (k, name)] = v
                k += 1
    c = bucket_case(rng, np.int64, 0, [0, 10, 20], 0, 0)      # no ids at all
    for name, v in c.items():
        out["c%d_%s" % (k, name)] = v
    k += 1
    out["n_cases"] = np.array(k)
    np.savez_compressed(os.path.join(HERE, "bucketing.npz"), **out)
    return k


# ------------------------------------------------------------------------------------------------ gather / scatter
MANT = {np.float32: 23, np.float16: 10, np.float64: 52}


def closed_form(dt, rows, dim, row_start=0):
    r = np.arange(row_start, row_start + rows, dtype=np.int64)
    if dt in MANT:
        m = r & ((1 << (MANT[dt] + 1)) - 1)
        v = m.astype(np.float64 if dt == np.float64 else np.float32).astype(dt)   # static_cast<T>(float(data))
    else:
        v = r.astype(dt)                                                        # integer tables: plain cast
    return np.repeat(v[:, None], dim, axis=1)


def cast(a, odt):
    """device_matrix_type_cast: static_cast per element; double -> half goes through float (the reference's
    convert_type<double, __half>, gather_scatter_func.cuh:44-99)"""
    if a.dtype == np.float64 and odt == np.float16:
        return a.astype(np.float32).astype(np.float16)
    with np.errstate(over="ignore"):
        return a.astype(odt)


PAIRS = [(np.float32, np.float32), (np.float16, np.float16), (np.float16, np.float32), (np.float32, np.float16),
         (np.float64, np.float32), (np.float32, np.float64), (np.float64, np.float16), (np.int64, np.int64),
         (np.int32, np.int64), (np.int64, np.int32), (np.int8, np.int32), (np.int16, np.int8)]
DIMS = [1, 11, 32, 127, 128, 129, 513]


def gen_gather_scatter():
    rng = np.random.default_rng(20260929)
    out, k = {}, 0
    n_rows, n_idx = 257, 64

    def put(kind, tdt, odt, dim, stride, table, idx, expected):
        nonlocal k
        out["c%d_meta" % k] = np.array([kind, np.dtype(tdt).name, np.dtype(odt).name, str(dim), str(stride)])
        out["c%d_table" % k] = table
        out["c%d_idx" % k] = idx
        out["c%d_expected" % k] = expected
        k += 1

    for tdt, odt in PAIRS:
        for dim in DIMS:
            # the reference test's own shape: closed-form table, random ids, expected = cast(closed form)
            idt = np.int64 if (dim + len(np.dtype(tdt).name)) % 2 else np.int32
            idx = rng.integers(0, n_rows, n_idx).astype(idt)
            table = closed_form(tdt, n_rows, dim)
            put("gather", tdt, odt, dim, dim, table, idx, cast(table[idx.astype(np.int64)], odt))
    # negative ids leave their output row untouched (prefill value 9), padded table stride (dim + 3), duplicates
    for tdt, odt in [(np.float32, np.float32), (np.float16, np.float32), (np.int64, np.int32)]:
        for dim in (11, 128, 129):
            stride = dim + 3
            table = np.zeros((n_rows, stride), dtype=tdt)
            table[:, :dim] = closed_form(tdt, n_rows, dim)
            table[:, dim:] = 5                                   # padding columns must never be read into the output
            idx = rng.integers(0, n_rows, n_idx).astype(np.int64)
            idx[::9] = -1
            idx[1::13] = idx[1]
            exp = np.full((n_idx, dim), 9, dtype=odt)
            v = idx >= 0
            exp[v] = cast(table[idx[v], :dim], odt)
            put("gather_neg_pad", tdt, odt, dim, stride, table, idx, exp)
    # random-valued tables: the cast roundings matter
    for tdt, odt in [(np.float32, np.float16), (np.float64, np.float16), (np.float64, np.float32), (np.float16, np.float64)]:
        for dim in (11, 128, 129):
            scale = rng.choice([1e-6, 1e-3, 1.0, 300.0, 1e4], (64, 1))   # |x| stays below the f16 range
            table = (rng.standard_normal((64, dim)) * scale).astype(tdt)
            idx = rng.integers(0, 64, n_idx).astype(np.int32)
            put("gather_random", tdt, odt, dim, dim, table, idx, cast(table[idx.astype(np.int64)], odt))
    # scatter: rows of the INPUT dtype (closed form of the destination row id) written at distinct ids into zeros
    for tdt, idt_ in [(np.float32, np.float32), (np.float16, np.float32), (np.float32, np.float16), (np.int64, np.int32),
                      (np.int32, np.int64), (np.float64, np.float32)]:
        for dim in DIMS:
            idx = rng.permutation(n_rows)[:n_idx].astype(np.int64 if dim % 2 else np.int32)
            if dim in (11, 128):
                idx[::10] = -1                                    # skipped
            rows = np.zeros((n_idx, dim), dtype=idt_)
            v = idx >= 0
            rows[v] = closed_form(idt_, n_rows, dim)[idx[v].astype(np.int64)]
            exp = np.zeros((n_rows, dim), dtype=tdt)
            exp[idx[v].astype(np.int64)] = cast(rows[v], tdt)
            put("scatter", tdt, idt_, dim, dim, rows, idx, exp)   # table slot = input rows; expected = table afterwards
    out["n_cases"] = np.array(k)
    np.savez_compressed(os.path.join(HERE, "gather_scatter.npz"), **out)
    return k


# ------------------------------------------------------------------------------------------------ dedup + optimizers
def dedup_first_seen(ids, grads):
    """first occurrence copied, later ones added one at a time in order of arrival (fp32); unique ids ascending"""
    acc = {}
    for i, ix in enumerate(ids.tolist()):
        if ix in acc:
            acc[ix] = (acc[ix] + grads[i]).astype(F)
        else:
            acc[ix] = grads[i].astype(F).copy()
    u = np.array(sorted(acc), dtype=ids.dtype)
    return u, np.stack([acc[int(x)] for x in u]) if len(u) else np.zeros((0, grads.shape[1]), F)


def step(kind, p, lr, table, u, g, st0, st1, per_row):
    """one optimizer step on rows `u` with de-duplicated gradients g; every statement one fp32 rounding"""
    wd, eps, b1, b2, alpha = F(p["weight_decay"]), F(p["epsilon"]), F(p["beta1"]), F(p["beta2"]), F(p["alpha"])
    lr = F(lr)
    e = table[u]
    one = F(1)
    if kind == "sgd":
        g = g + wd * e
        e = e - lr * g
    elif kind == "adam":
        b1t = per_row[u, 0] * b1
        b2t = per_row[u, 1] * b2
        if p["adam_w"]:
            e = e - (lr * wd) * e
        else:
            g = g + wd * e
        m = b1 * st0[u] + (one - b1) * g
        v = b2 * st1[u] + ((one - b2) * g) * g
        mhat = m / (one - b1t)[:, None]
        vhat = v / (one - b2t)[:, None]
        e = e - (lr * mhat) / (np.sqrt(vhat) + eps)
        st0[u], st1[u] = m, v
        per_row[u, 0], per_row[u, 1] = b1t, b2t
    elif kind == "adagrad":
        g = g + wd * e
        s = st0[u] + g * g
        e = e - (lr * g) / (np.sqrt(s) + eps)
        st0[u] = s
    elif kind == "rmsprop":
        g = g + wd * e
        v = alpha * st0[u] + ((one - alpha) * g) * g
        e = e - (lr * g) / (np.sqrt(v) + eps)
        st0[u] = v
    assert e.dtype == F
    table[u] = e


OPT_CASES = [("sgd", {"weight_decay": 0.1}), ("adam", {"weight_decay": 0.05}), ("adam", {"weight_decay": 0.02, "adam_w": 1.0}),
             ("adagrad", {"weight_decay": 0.01}), ("rmsprop", {"alpha": 0.9})]
OPT_DEFAULTS = {"weight_decay": 0.0, "epsilon": 1e-8, "beta1": 0.9, "beta2": 0.999, "alpha": 0.99, "adam_w": 0.0}


def gen_optimizers():
    rng = np.random.default_rng(20260930)
    n_rows, dim, n_ids, steps = 1000, 127, 400, 2
    table0 = rng.standard_normal((n_rows, dim)).astype(F)
    ids, grads = [], []
    for s in range(steps):
        ix = rng.integers(0, n_rows, n_ids).astype(np.int64)
        ix[::5] = ix[0]              # one id with 80 duplicates
        ix[1::11] = ix[1]
        ids.append(ix)
        grads.append(rng.standard_normal((n_ids, dim)).astype(F))
    out = {"table0": table0, "lr": np.array(0.05, F), "n_steps": np.array(steps)}
    for s in range(steps):
        out["ids_%d" % s] = ids[s]
        out["grads_%d" % s] = grads[s]
        u, dg = dedup_first_seen(ids[s], grads[s])
        out["unique_%d" % s] = u
        out["dedup_grads_%d" % s] = dg
    # rows no step touches must come out bit-identical to table0 (states: zero, beta powers: one): only the touched rows'
    # expectations are stored, which keeps the file small
    touched = np.unique(np.concatenate(ids))
    out["touched"] = touched
    for k, (kind, over) in enumerate(OPT_CASES):
        p = dict(OPT_DEFAULTS, **over)
        table = table0.copy()
        st0, st1 = np.zeros_like(table), np.zeros_like(table)
        per_row = np.ones((n_rows, 2), F)
        out["o%d_kind" % k] = np.array(kind)
        out["o%d_params" % k] = np.array([p["weight_decay"], p["epsilon"], p["beta1"], p["beta2"], p["alpha"], p["adam_w"]], F)
        for s in range(steps):
            u, dg = dedup_first_seen(ids[s], grads[s])
            step(kind, p, 0.05, table, u, dg, st0, st1, per_row)
            if s == steps - 1 or kind == "sgd":
                out["o%d_table_%d" % (k, s)] = table[touched].copy()
        untouched = np.setdiff1d(np.arange(n_rows), touched)
        assert np.array_equal(table[untouched], table0[untouched])
        if kind in ("adam", "adagrad", "rmsprop"):
            out["o%d_state0" % k] = st0[touched]
        if kind == "adam":
            out["o%d_state1" % k] = st1[touched]
            out["o%d_per_row" % k] = per_row[touched]
    out["n_cases"] = np.array(len(OPT_CASES))
    np.savez_compressed(os.path.join(HERE, "optimizers.npz"), **out)
    return len(OPT_CASES)


if __name__ == "__main__":
    print("bucketing cases:", gen_bucketing())
    print("gather / scatter cases:", gen_gather_scatter())
    print("optimizer cases:", gen_optimizers())
    for f in ("bucketing.npz", "gather_scatter.npz", "optimizers.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
