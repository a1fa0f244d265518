"""CPU tests of the graph-op oracle (oracle/wm_graph_oracle.c) and of the product library's HOST random helpers.

Pinning:
  * the PCG-XSH-RR 64/32 core against the published pcg32 known-answer vector (pcg32-demo, seed 42, stream 54);
  * skip-ahead against stepping;
  * the sampler against the reference's own host statement of index sampling (Q = iota; a[i] = Q[r[i]];
    Q[r[i]] = Q[N-1-i], tests/wholegraph_ops/graph_sampling_test_utils.cu:306-321) re-derived here in numpy from
    the same draws;
  * append_unique / add_self_loop against independent numpy statements (reference test oracles
    tests/graph_ops/append_unique_test_utils.cu:27-80, csr_add_self_loop_utils.cu).
raft's wrapping of the generator stays "parity unpinned" (oracle/wm_graph_oracle.c header).
"""
import numpy as np
import pytest

import oracle


def test_pcg32_known_answer():
    # O'Neill's pcg32-demo: pcg32_srandom_r(&rng, 42u, 54u) -> first six outputs
    want = [0xA15C02B7, 0x7B47F409, 0xBA1D3330, 0x83D2F293, 0xBFA4784B, 0xCBED606E]
    assert [int(x) for x in oracle.pcg_raw(42, 54, 0, 6)] == want


@pytest.mark.parametrize("skip", [1, 2, 7, 64, 1000, 12345])
def test_pcg_skipahead_equals_stepping(skip):
    full = oracle.pcg_raw(2024, 77, 0, skip + 8)
    assert np.array_equal(oracle.pcg_raw(2024, 77, skip, 8), full[skip:])


def test_random_positive_int_layout():
    # generator(seed, subsequence) = init(seed, subsequence, offset = subsequence); int32 = output & 0x7fffffff,
    # int64 = (lo | hi << 32) & (2^63 - 1)
    raw = oracle.pcg_raw(9, 5, 5, 16).astype(np.uint64)
    assert np.array_equal(oracle.random_positive_int(9, 5, 16), (raw & 0x7FFFFFFF).astype(np.int32))
    want64 = ((raw[0::2] | (raw[1::2] << np.uint64(32))) & np.uint64(0x7FFFFFFFFFFFFFFF)).astype(np.int64)
    assert np.array_equal(oracle.random_positive_int(9, 5, 8, np.int64), want64)


def _geometry(m):
    warps = [1, 1, 1, 2, 2, 2, 4, 4, 4, 4, 4, 4] + [8] * 20
    items = [1, 2, 3, 2, 3, 3, 2, 2, 3, 3, 3, 3, 2, 2, 2, 2] + [3] * 8 + [4] * 8
    f = (m - 1) // 32
    return warps[f] * 32, items[f]


def _numpy_sample(row_ptr, col, centers, m, seed):
    """The reference host statement, from draws obtained one virtual thread at a time."""
    out, lid, egid = [], [], []
    for c, nid in enumerate(centers):
        s, e = int(row_ptr[nid]), int(row_ptr[nid + 1])
        n = e - s
        if m <= 0 or n <= m:
            a = list(range(n))
        elif m > 1024:
            a = list(range(m))
            for t in range(32):
                idxs = list(range(m + t, n, 32))
                draws = oracle.random_positive_int(seed, c * 32 + t, len(idxs))
                for idx, d in zip(idxs, draws):
                    r = int(d) % (idx + 1)
                    if r < m:
                        a[r] = max(a[r], idx)
        else:
            T, items = _geometry(m)
            r = [0] * m
            for t in range(min(T, m)):
                draws = oracle.random_positive_int(seed, c * T + t, items)
                for k in range(items):
                    idx = k * T + t
                    if idx < m:
                        r[idx] = int(draws[k]) % (n - idx)
            q = list(range(n))
            a = []
            for i in range(m):
                a.append(q[r[i]])
                q[r[i]] = q[n - 1 - i]
        out += [int(col[s + x]) for x in a]
        lid += [c] * len(a)
        egid += [s + x for x in a]
    return np.array(out, dtype=col.dtype), np.array(lid, dtype=np.int32), np.array(egid, dtype=np.int64)


def make_csr(n_nodes, max_degree, seed, col_dtype=np.int64, heavy=()):
    rng = np.random.default_rng(seed)
    deg = rng.integers(0, max_degree + 1, n_nodes)
    for node, d in heavy:
        deg[node] = d
    row_ptr = np.zeros(n_nodes + 1, dtype=np.int64)
    np.cumsum(deg, out=row_ptr[1:])
    col = rng.integers(0, n_nodes, int(row_ptr[-1])).astype(col_dtype)
    return row_ptr, col


@pytest.mark.parametrize("m", [-1, 1, 5, 30, 32, 33, 97, 200, 1024, 1025, 1500])
def test_oracle_sampler_matches_reference_host_statement(m):
    row_ptr, col = make_csr(60, 70, 3 + max(m, 0), heavy=[(3, 2100), (17, 1300), (40, 1024), (41, 1025)])
    centers = np.array([3, 17, 0, 40, 41, 5, 3, 59, 22], dtype=np.int64)
    off, ids, lid, egid = oracle.sample_unweighted(row_ptr, col, centers, m, 0xDEADBEEFCAFE)
    want_ids, want_lid, want_egid = _numpy_sample(row_ptr, col, centers, m, 0xDEADBEEFCAFE)
    deg = row_ptr[centers + 1] - row_ptr[centers]
    cnt = deg if m <= 0 else np.minimum(deg, m)
    assert np.array_equal(off, np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32))
    assert np.array_equal(ids, want_ids) and np.array_equal(lid, want_lid) and np.array_equal(egid, want_egid)
    # without replacement: edge ids of one center node are distinct and inside its row
    for c in range(len(centers)):
        e = egid[off[c]:off[c + 1]]
        assert len(set(e.tolist())) == len(e)
        assert np.all((e >= row_ptr[centers[c]]) & (e < row_ptr[centers[c] + 1]))


def test_oracle_append_unique():
    rng = np.random.default_rng(5)
    targets = rng.permutation(5000)[:700].astype(np.int64)
    neighbors = rng.integers(0, 5000, 9000).astype(np.int64)
    uniq, mapping = oracle.append_unique(targets, neighbors)
    assert np.array_equal(uniq[:700], targets)
    tail = uniq[700:]
    # first-seen order of the neighbour ids that are not targets
    seen, want = set(targets.tolist()), []
    for v in neighbors.tolist():
        if v not in seen:
            seen.add(v)
            want.append(v)
    assert tail.tolist() == want
    assert np.array_equal(uniq[mapping], neighbors)
    # the docstring example of the reference (python/.../torch/graph_ops.py:29-35)
    uniq, mapping = oracle.append_unique(np.array([3, 11, 2, 10]), np.array([4, 5, 2, 11, 6, 9, 10, 5]))
    assert sorted(uniq.tolist()) == sorted([3, 11, 2, 10, 6, 4, 9, 5]) and uniq[:4].tolist() == [3, 11, 2, 10]
    assert np.array_equal(uniq[mapping], [4, 5, 2, 11, 6, 9, 10, 5])


def test_oracle_add_self_loop():
    row_ptr, col = make_csr(200, 9, 11, np.int32)
    out_row, out_col = oracle.csr_add_self_loop(row_ptr.astype(np.int32), col)
    assert np.array_equal(out_row, row_ptr + np.arange(201))
    for r in range(200):
        seg = out_col[out_row[r]:out_row[r + 1]]
        assert seg[0] == r and np.array_equal(seg[1:], col[row_ptr[r]:row_ptr[r + 1]])


# ---- the product library's host helpers (no GPU needed: they only touch host tensors) ----
def test_library_random_positive_int_matches_oracle(wm_lib):
    import wholegraph_amd.torch.wholegraph_ops as wops
    for seed, sub, n in [(1, 0, 10), (12345, 7, 100), (2 ** 40 + 3, 1000, 33)]:
        got = wops.generate_random_positive_int_cpu(seed, sub, n).numpy()
        assert np.array_equal(got, oracle.random_positive_int(seed, sub, n))
        assert got.min() >= 0


def test_library_exponential_helper(wm_lib):
    import wholegraph_amd.torch.wholegraph_ops as wops
    v = wops.generate_exponential_distribution_negative_float_cpu(99, 3, 20000).numpy().astype(np.float64)
    assert np.all(v < 0) and np.all(np.isfinite(v))
    # log2 of a uniform(0,1): mean -1/ln2, P(v < -1) = 1/2
    assert abs(v.mean() + 1 / np.log(2)) < 0.05
    assert abs((v < -1).mean() - 0.5) < 0.02
    again = wops.generate_exponential_distribution_negative_float_cpu(99, 3, 20000).numpy()
    assert np.array_equal(again.astype(np.float64), v)


def test_samplers_reject_null_arguments(wm_lib):
    from wholegraph_amd import binding as wmb
    rc = wm_lib.wholegraph_csr_weighted_sample_without_replacement(None, None, None, None, 1, None, None, None, None, 0,
                                                                   None, None)
    assert wmb.ERROR_NAMES[rc] == "WHOLEMEMORY_INVALID_INPUT"
    rc = wm_lib.wholegraph_csr_unweighted_sample_without_replacement(None, None, None, 1, None, None, None, None, 0,
                                                                     None, None)
    assert wmb.ERROR_NAMES[rc] == "WHOLEMEMORY_INVALID_INPUT"


# ---- weighted sampling ----
def test_det_log2_1p_matches_libm():
    rng = np.random.default_rng(0)
    # the sampler only ever passes x = -(24-bit mantissa in [0.5, 1]) * 2^-k, for which 1 + x is exact
    m24 = (0.5 + 0.5 * rng.random(6000).astype(np.float32)).astype(np.float32).astype(np.float64)
    xs = np.concatenate([-m24[:2000], -m24[2000:] * 2.0 ** -rng.integers(1, 80, 4000).astype(np.float64),
                         [-0.5, -0.25, -2.0 ** -27, -2.0 ** -28, -(1 - 2.0 ** -24) * 2.0 ** -3]])
    for x in xs:
        want = np.log1p(x) / np.log(2.0)
        got = oracle.det_log2_1p(x)
        assert abs(got - want) <= 1e-15 * abs(want) + 1e-300, (x, got, want)
    assert oracle.det_log2_1p(-1.0) == -np.inf


def test_weighted_keys_are_log2_uniform_over_weight():
    w = np.ones(200000, dtype=np.float32)
    k = oracle.weighted_keys(5, 11, w).astype(np.float64)
    assert np.all(k < 0)
    # -ln(2) * key ~ Exp(1): mean 1, P(> 1) = 1/e
    x = -np.log(2.0) * k
    assert abs(x.mean() - 1.0) < 0.01 and abs((x > 1.0).mean() - np.exp(-1)) < 0.005
    # weight w scales the key by 1/w (same stream, same uniforms)
    k4 = oracle.weighted_keys(5, 11, np.full(1000, 4.0, dtype=np.float32))
    assert np.array_equal(k4, (k[:1000].astype(np.float32) * np.float32(0.25)))


@pytest.mark.parametrize("m", [1, 7, 30, 128, 256, 257, 300])
@pytest.mark.parametrize("wdtype", [np.float32, np.float64])
def test_oracle_weighted_sampler_matches_reference_host_statement(m, wdtype):
    row_ptr, col = make_csr(40, 60, 77 + m, heavy=[(2, 900), (9, 257), (10, 256), (11, 1)])
    weights = (np.random.default_rng(m).random(col.shape[0]) + 0.05).astype(wdtype)
    centers = np.array([2, 9, 10, 11, 0, 2, 33], dtype=np.int32)
    seed = 0x1234567 + m
    off, ids, lid, egid = oracle.sample_weighted(row_ptr, col, weights, centers, m, seed)
    block = 256 if m > 256 else 128
    want = []
    for c, nid in enumerate(centers):
        s, e = int(row_ptr[nid]), int(row_ptr[nid + 1])
        n = e - s
        if n <= m:
            want += list(range(s, e))
            continue
        keys = np.empty(n, dtype=np.float32)
        for t in range(min(block, n)):   # virtual thread t keys neighbours t, t + block, ... consecutively
            sel = np.arange(t, n, block)
            keys[sel] = oracle.weighted_keys(seed, c * block + t, weights[s + sel].astype(np.float32))
        order = np.lexsort((np.arange(n), -keys.astype(np.float64)))[:m]
        want += [s + int(i) for i in order]
    assert np.array_equal(egid, np.array(want, dtype=np.int64))
    assert np.array_equal(ids, col[egid]) and np.array_equal(lid, np.repeat(np.arange(len(centers)), np.diff(off)))


def test_sampling_recurrence_closed_form():
    """The closed form sample_small_kernel evaluates (csrc/kernels/graph.hip) against the sequential recurrence of the reference
    (unweighted_sample_without_replacement_func.cuh: a[i] = Q[r_i]; Q[r_i] = Q[N-1-i] over a sparse image of Q): lane i looks for
    the last earlier step whose draw equals its own draw (px) or its own tail position N-1-i (py), the value chains through py
    are resolved by pointer jumping, a[i] = N-1-root(px) or r_i. Pure Python on random and collision-heavy draws, both group
    sizes of the kernel (32 and 64 lanes per centre)."""
    import random

    def sequential(r, n):
        q, a = {}, []
        for i, x in enumerate(r):
            y = n - 1 - i
            vx, vy = q.get(x, x), q.get(y, y)
            a.append(vx)
            q[x] = vy
        return a

    def closed_form(r, n, group):
        m = len(r)
        xs = r + [0] * (group - m)
        px, py = [-1] * group, [-1] * group
        for lane in range(group):
            for k in range(min(m, lane)):
                if xs[k] == xs[lane]:
                    px[lane] = k
                if xs[k] == n - 1 - lane:
                    py[lane] = k
        root = [lane if py[lane] < 0 else py[lane] for lane in range(group)]
        for _ in range(6 if group == 64 else 5):
            root = [root[root[lane]] for lane in range(group)]
        return [xs[i] if px[i] < 0 else n - 1 - root[px[i]] for i in range(m)]

    rnd = random.Random(5)
    for case in range(30000):
        group = 32 if case % 2 else 64
        m = rnd.choice([1, 2, 3, 5, 7, 15, 30, 31, 32] if group == 32 else [1, 2, 30, 33, 48, 63, 64])
        n = m + rnd.choice([1, 1, 2, 3, 5, 10, 40, 1000, 100000])
        r = [rnd.randrange(0, n - i) for i in range(m)]
        if case % 3 == 0:   # many collisions: draws on the tail positions and on each other
            r = [max(0, min(n - 1 - i, rnd.choice([0, 1, n - 1 - i, n - 2 - i, r[0]]))) for i in range(m)]
        assert closed_form(r, n, group) == sequential(r, n), (case, group, m, n, r)
