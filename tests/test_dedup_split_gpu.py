"""The owner-side id sort as a split sort (csrc/kernels/split_sort.cuh) — reference exchange_embeddings_nccl_func.cu:76-174
(stable radix sort of the received ids, payload = receive position, then unique_by_key and a SEQUENTIAL sum per id).

The raw stage wholememory_ext_dedup_apply is run as an SGD step with lr = -1, weight decay 0 (row += ordered sum of its
gradients) on RANDOM fp32 gradients: the result depends on the order in which duplicates are summed, so it is bit-identical to
the oracle (oracle.dedup_grads: stable sort, sums in receive order) only if the sort is stable — on every path:
  * map path      buckets whose rows appear <= 8 times (uniform ids)
  * radix path    buckets with a longer run (LDS radix passes inside the same kernel)
  * generic path  a bucket that does not fit LDS (> 8192 ids): the device-side overflow word switches the split kernels off and
                  the gated onesweep + run detection on
each forced to be taken by small batches through WM_DEDUP_SPLIT_MIN=1, and compared with WM_DEDUP_SPLIT=0 (rocPRIM)."""
import ctypes as C

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _env():
    from wholegraph_amd.torch import wholegraph_env
    import torch
    return wholegraph_env.get_wholegraph_env_fns(), torch.cuda.current_stream().cuda_stream


def _apply(ids, grads, rows, row_offset, idt):
    import torch
    from wholegraph_amd import binding as wmb
    env, stream = _env()
    dim = grads.shape[1]
    d_table = torch.zeros((rows, dim), device="cuda")
    d_ids, d_grads = torch.from_numpy(ids.astype(idt)).cuda(), torch.from_numpy(grads).cuda()
    arr = (C.c_float * 6)(0.0, 1e-8, 0.9, 0.999, 0.99, 0.0)
    nu = C.c_int64(-1)
    wmb.check(wmb.lib().wholememory_ext_dedup_apply(
        d_ids.data_ptr(), wmb.DT_INT if idt == np.int32 else wmb.DT_INT64, len(ids), d_grads.data_ptr(), dim, dim,
        d_table.data_ptr(), dim, row_offset, rows, 1, arr, -1.0, None, None, C.byref(nu), env, stream))
    torch.cuda.synchronize()
    return d_table.cpu().numpy(), nu.value


def _expect(ids, grads, rows, row_offset):
    keep = (ids >= row_offset) & (ids < row_offset + rows)
    uniq, dg = oracle.dedup_grads(ids[keep].astype(np.int64), grads[keep])
    t = np.zeros((rows, grads.shape[1]), np.float32)
    t[uniq - row_offset] = dg          # 0 - (-1) * sum: exact
    return t, len(uniq)


CASES = {
    # name: (n, rows, row_offset, generator of ids in [0, rows))
    "uniform_sparse": (60000, 5_000_000, 0, lambda r, n, rows: r.integers(0, rows, n)),
    "uniform_offset_int32": (70001, 3_000_000, 40_000_000, lambda r, n, rows: r.integers(0, rows, n)),
    "runs_up_to_8": (50000, 400_000, 0, lambda r, n, rows: np.repeat(r.integers(0, rows, n // 4), r.integers(1, 9, n // 4))[:n]),
    "long_runs_radix_path": (40000, 4_000_000, 1000, lambda r, n, rows: np.where(r.random(n) < 0.3, r.integers(0, 40, n) * 70001 % rows,
                                                                                r.integers(0, rows, n))),
    "dense_small_table": (30000, 300, 0, lambda r, n, rows: r.integers(0, rows, n)),
    "hot_id_overflows_bucket": (90000, 6_000_000, 0, lambda r, n, rows: np.where(r.random(n) < 0.25, 4242, r.integers(0, rows, n))),
    "clustered_overflow": (120000, 100_000_000, 0, lambda r, n, rows: r.integers(0, 30000, n)),
    "single_id": (20000, 1_000_000, 0, lambda r, n, rows: np.full(n, 777)),
    "ascending": (50000, 50_000_000, 0, lambda r, n, rows: np.arange(n) * 997),
    "two_to_27_rows": (65536, 1 << 27, 0, lambda r, n, rows: r.integers(0, rows, n)),
}


@pytest.mark.parametrize("split", ["forced", "off", "hot"])
@pytest.mark.parametrize("name", list(CASES))
def test_split_sort_keeps_the_reference_order(gpu_env, knobs, name, split):
    n, rows, off, gen = CASES[name]
    rng = np.random.default_rng(abs(hash(name)) % (1 << 31))
    ids = np.asarray(gen(rng, n, rows)).astype(np.int64)[:n] + off
    # a few ids that address no row of the owner ("skip me" negatives, ids past its range): dropped, not counted
    ids[rng.integers(0, n, 7)] = -1
    ids[rng.integers(0, n, 5)] = off + rows + 3
    grads = rng.standard_normal((n, 8)).astype(np.float32)
    if split == "forced":
        knobs.set("WM_DEDUP_SPLIT_MIN", 1)
    elif split == "hot":   # round 6: the split sort with the batch's hot ids peeled into buckets of their own (split::launch_hot)
        knobs.set("WM_DEDUP_SPLIT_MIN", 1)
        knobs.set("WM_DEDUP_HOT", 2)
    else:
        knobs.set("WM_DEDUP_SPLIT", 0)
    idt = np.int32 if "int32" in name else np.int64
    rows_alloc = min(rows, 6_000_000)   # the table only needs the rows that are hit ...
    if rows > rows_alloc:               # ... so big row ranges are folded into the head of the range (same keys, same order)
        ids_eff = np.where((ids >= off) & (ids < off + rows), off + (ids - off) % rows_alloc, ids)
        want, nu_want = _expect(ids_eff, grads, rows_alloc, off)
        # sort keys differ after folding: run the device on the folded ids too, but with the FULL row range as its bound
        got, nu = _apply_bounded(ids_eff, grads, rows_alloc, off, idt, rows)
    else:
        want, nu_want = _expect(ids, grads, rows, off)
        got, nu = _apply(ids, grads, rows, off, idt)
    assert nu == nu_want
    assert got.tobytes() == want.tobytes()


def _apply_bounded(ids, grads, rows_alloc, row_offset, idt, rows_bound):
    """as _apply, but the owner's row range (what bounds the sort keys) is rows_bound while only rows_alloc rows are hit"""
    import torch
    from wholegraph_amd import binding as wmb
    env, stream = _env()
    dim = grads.shape[1]
    d_table = torch.zeros((rows_alloc, dim), device="cuda")
    d_ids, d_grads = torch.from_numpy(ids.astype(idt)).cuda(), torch.from_numpy(grads).cuda()
    arr = (C.c_float * 6)(0.0, 1e-8, 0.9, 0.999, 0.99, 0.0)
    nu = C.c_int64(-1)
    # local_entry_count = rows_bound: the kernels only touch rows that occur, all of them < rows_alloc
    wmb.check(wmb.lib().wholememory_ext_dedup_apply(
        d_ids.data_ptr(), wmb.DT_INT if idt == np.int32 else wmb.DT_INT64, len(ids), d_grads.data_ptr(), dim, dim,
        d_table.data_ptr(), dim, row_offset, rows_bound, 1, arr, -1.0, None, None, C.byref(nu), env, stream))
    torch.cuda.synchronize()
    return d_table.cpu().numpy(), nu.value


def test_default_threshold_takes_the_split_sort_for_big_batches(gpu_env, knobs):
    """2 M uniform ids on a 100 M-row range, defaults: the split sort (checked through its counter of launches)"""
    from wholegraph_amd import binding as wmb
    rng = np.random.default_rng(11)
    n, rows_alloc, bound = 2_000_000, 4_000_000, 100_000_000
    ids = rng.integers(0, rows_alloc, n).astype(np.int64)
    grads = rng.integers(-3, 4, (n, 8)).astype(np.float32)
    before = wmb.lib().wholememory_ext_split_sorts()
    got, nu = _apply_bounded(ids, grads, rows_alloc, 0, np.int64, bound)
    assert wmb.lib().wholememory_ext_split_sorts() == before + 1
    want = np.zeros((rows_alloc, 8), np.float32)
    np.add.at(want, ids, grads)        # integer-valued: exact in any order
    assert nu == len(np.unique(ids)) and got.tobytes() == want.tobytes()


def test_long_run_side_follows_the_batches(gpu_env, knobs):
    """The long-run side of the step is queued on the caller's stream while the previous calls listed no long run, beside the tile
    kernel once one did (a count that lags a call or two, optim.hip: long_lane::expect_long), and its listing kernel is gated
    on what the split sort saw. Whatever the sequence of batches — no long run, a 30 k-row run, none again — every call's
    result is the oracle's, bit for bit."""
    knobs.set("WM_DEDUP_SPLIT_MIN", 1)
    rng = np.random.default_rng(99)
    rows, n = 3_000_000, 120_000
    for kind in ("uniform", "hot", "uniform", "uniform", "hot", "hot", "radix", "uniform"):
        ids = rng.integers(0, rows, n).astype(np.int64)
        if kind == "hot":
            ids[rng.random(n) < 0.25] = 123_457            # one run of ~30 k rows: the bucket overflows, generic path, long-run fold
        if kind == "radix":
            hot = rng.integers(0, rows, 60)
            sel = rng.random(n) < 0.3
            ids[sel] = hot[rng.integers(0, 60, int(sel.sum()))]   # 60 runs of ~600 rows: radix path of their buckets, long-run fold
        grads = rng.standard_normal((n, 8)).astype(np.float32)
        want, nu_want = _expect(ids, grads, rows, 0)
        got, nu = _apply(ids, grads, rows, 0, np.int64)
        assert nu == nu_want, kind
        assert got.tobytes() == want.tobytes(), kind


@pytest.mark.parametrize("kind", ["uniform", "hot_id"])
def test_gradient_apply_replays_from_a_hipgraph(gpu_env, knobs, kind):
    """One rank's gradient apply has no host synchronisation, so it can be captured. While a stream is being captured the
    split sort forks and joins its side stream with EVENTS only (optim.hip: a wave that waits for a word needs the other branch
    to be running, and the branches of a graph may be replayed one after the other): the replay must give the eager call's
    bits on the map path and on the generic path (a hot id that overflows its bucket)."""
    import torch
    import wholegraph_amd.torch as wgth
    knobs.set("WM_DEDUP_SPLIT_MIN", 1)
    rows, dim, n = 300_000, 32, 120_000
    rng = np.random.default_rng(5)
    ids = rng.integers(0, rows, n)
    if kind == "hot_id":
        ids = np.where(rng.random(n) < 0.3, 4242, ids)
    idx = torch.from_numpy(ids.astype(np.int64)).cuda()
    grads = torch.from_numpy(rng.standard_normal((n, dim)).astype(np.float32)).cuda()

    def table():
        emb = wgth.create_embedding(gpu_env, "chunked", "cuda", torch.float32, [rows, dim])
        wgth.create_wholememory_optimizer(emb, "sgd", {})
        local, _ = emb.get_embedding_tensor().get_local_tensor()
        local.zero_()
        return emb, local

    def step(emb):
        emb.add_gradients(idx, grads)
        emb.need_apply = True
        emb.apply_gradients(0.5)

    eager, eager_rows = table()
    step(eager)
    step(eager)
    torch.cuda.synchronize()
    captured, captured_rows = table()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):       # (warm-up outside the capture: lanes, attributes, allocator pools)
        step(captured)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    captured_rows.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step(captured)
    torch.cuda.synchronize()
    captured_rows.zero_()
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    assert captured_rows.cpu().numpy().tobytes() == eager_rows.cpu().numpy().tobytes()


MODES = {
    # how the sort's side stream is forked / joined and where the step's long-run side runs (optim.hip); the default is
    # "word fork + deferred join + detached long-run side", every other combination stays selectable for A/B runs
    "event_fork": {"WM_DEDUP_FORK_EVENT": 1},
    "join_in_front": {"WM_DEDUP_DEFER_JOIN": 0},
    "long_side_inline": {"WM_STEP_DETACH": 0},
    "all_on_one_stream": {"WM_DEDUP_SERIAL": 1, "WM_STEP_SERIAL": 1},
    "high_priority_side": {"WM_DEDUP_LANE_PRIO": "h"},
    # no kernel may wait for another stream's kernel: what the library picks by itself under rocprofv3's counter collection
    # (one kernel runs at a time there; it exports ROCPROF_COUNTER_COLLECTION) — the --pmc passes hang otherwise
    "no_device_waits": {"WM_DEVICE_WAITS": 0},
    "under_counter_collection": {"ROCPROF_COUNTER_COLLECTION": 1},
}


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("name", ["uniform_sparse", "long_runs_radix_path", "hot_id_overflows_bucket"])
def test_every_fork_and_join_arrangement_gives_the_same_bits(gpu_env, knobs, name, mode):
    n, rows, off, gen = CASES[name]
    rng = np.random.default_rng(abs(hash(name)) % (1 << 31) + 17)
    ids = np.asarray(gen(rng, n, rows)).astype(np.int64)[:n] + off
    grads = rng.standard_normal((n, 8)).astype(np.float32)
    knobs.set("WM_DEDUP_SPLIT_MIN", 1)
    for k, v in MODES[mode].items():
        knobs.set(k, v)
    want, nu_want = _expect(ids, grads, rows, off)
    for _ in range(3):     # (the long-run side follows the previous calls: let it settle in either arrangement)
        got, nu = _apply(ids, grads, rows, off, np.int64)
        assert nu == nu_want
        assert got.tobytes() == want.tobytes()


def test_route_follows_the_batches(gpu_env, knobs):
    """Adaptive route (optim.hip: run_dedup): once a split sort overflowed a bucket, the next batches of the same row range go
    straight to rocPRIM's sort (a series of skewed batches is the usual case, and the gated generic path is the slower of the two);
    every fourth such call probes with the split sort's first two kernels, and the first batch that would not overflow switches
    back. Whatever the route, every call's result is the oracle's, bit for bit; the route shows in the split sort counter."""
    from wholegraph_amd import binding as wmb
    knobs.set("WM_DEDUP_SPLIT_MIN", 1)
    knobs.unset("WM_DEDUP_HOT")                # (the default: round 5's two routes; the opt-in hot-id route of round 6: next test)
    rng = np.random.default_rng(2024)
    rows, n = 2_000_000, 80_000
    uniform = rng.integers(0, rows, n).astype(np.int64)
    skewed = np.where(rng.random(n) < 0.4, 31337, rng.integers(0, rows, n)).astype(np.int64)
    grads = rng.standard_normal((n, 8)).astype(np.float32)
    want_u, want_s = _expect(uniform, grads, rows, 0), _expect(skewed, grads, rows, 0)
    count = wmb.lib().wholememory_ext_split_sorts

    def call(ids, want):
        before = count()
        got, nu = _apply(ids, grads, rows, 0, np.int64)
        assert nu == want[1] and got.tobytes() == want[0].tobytes()
        return count() - before

    assert call(uniform, want_u) == 1          # nothing known about this row range: the split sort
    assert call(skewed, want_s) == 1           # ... which overflows (the generic path sorts the batch) and says so
    assert call(skewed, want_s) == 0           # the series continues on rocPRIM's sort
    assert call(skewed, want_s) == 0
    routes = [call(uniform, want_u) for _ in range(8)]   # the batches stop being skewed: a probe notices within four calls
    assert routes[0] == 0 and routes[-1] == 1 and sum(routes) >= 3, routes
    knobs.set("WM_DEDUP_ADAPT", 0)             # (a reload forgets what was learnt; ADAPT=0: always the split sort)
    assert call(skewed, want_s) == 1 and call(skewed, want_s) == 1


def test_hot_route_follows_the_batches(gpu_env, knobs):
    """Round 6, three routes per row range: after a plain split sort overflowed, the next batches take the split sort with their
    hot ids peeled into buckets of their own (counted by wholememory_ext_hot_split_sorts) — a skewed series stays on the
    hand-written sort; when that overflows too (ids clustered in a few thousand rows) the series goes to rocPRIM's sort and is
    probed every fourth call; a hot-mode sort that peeled nothing hands the range back to the plain split sort. Every call's
    result is the oracle's, bit for bit, whatever the route."""
    from wholegraph_amd import binding as wmb
    knobs.set("WM_DEDUP_SPLIT_MIN", 1)
    knobs.set("WM_DEDUP_HOT", 1)               # (opt-in: the hot-mode sort is correct but not faster than rocPRIM's yet)
    rng = np.random.default_rng(77)
    rows, n = 3_000_000, 90_000
    uniform = rng.integers(0, rows, n).astype(np.int64)
    skewed = np.where(rng.random(n) < 0.3, 31337, rng.integers(0, rows, n)).astype(np.int64)
    skewed[rng.random(n) < 0.05] = 31338
    clustered = rng.integers(0, 20000, n).astype(np.int64)
    grads = rng.standard_normal((n, 8)).astype(np.float32)
    want = {k: _expect(v, grads, rows, 0) for k, v in (("u", uniform), ("s", skewed), ("c", clustered))}
    ids = {"u": uniform, "s": skewed, "c": clustered}
    splits, hots = wmb.lib().wholememory_ext_split_sorts, wmb.lib().wholememory_ext_hot_split_sorts

    def call(k):
        b_s, b_h = splits(), hots()
        got, nu = _apply(ids[k], grads, rows, 0, np.int64)
        assert nu == want[k][1] and got.tobytes() == want[k][0].tobytes(), k
        return splits() - b_s, hots() - b_h

    assert call("u") == (1, 0)                 # nothing known: the plain split sort
    assert call("s") == (1, 0)                 # ... which overflows (the generic path sorts the batch) and says so
    assert call("s") == (1, 1)                 # the series continues on the split sort with the hot ids peeled
    assert call("s") == (1, 1)
    assert call("c") == (1, 1)                 # clustered ids overflow a regular bucket whatever is peeled (generic path)
    assert call("c") == (0, 0)                 # ... and the series goes to rocPRIM's sort
    routes = [call("u") for _ in range(10)]    # uniform again: a probe notices within four calls, the hot mode finds nothing to peel
    assert routes[0] == (0, 0) and routes[-1] == (1, 0), routes
    assert any(r == (1, 1) for r in routes), routes


@pytest.mark.parametrize("gate", ["lookback", "join"])
def test_a_device_side_wait_that_gives_up_is_an_error_code_not_a_wrong_table(gpu_env, knobs, gate):
    """The split sort's waiting waves (bucket look-back, join of the generic path) give up after a bounded number of polls.
    Round 5 wrote a word nobody read and carried on — a stalled wait ended in a silently wrong gradient step. Now the sort's last
    kernel turns a sort with a timeout into "no runs" (the step behind it does nothing) and leaves the code in pinned memory,
    which the host reports as WHOLEMEMORY_CUDA_ERROR at its next synchronise / entry. Forced here by a short poll limit
    (WM_DEBUG_SPIN_LIMIT) and a gate that never opens (WM_DEBUG_STALL): 'lookback' = a stage-2 bucket that never publishes its
    run count (uniform ids, map path); 'join' = a generic path that never says done (a hot id overflows a bucket)."""
    import torch
    from wholegraph_amd import binding as wmb
    rng = np.random.default_rng(99)
    n, rows, dim = 90000, 6_000_000, 8
    ids = rng.integers(0, rows, n).astype(np.int64)
    if gate == "join":
        ids = np.where(rng.random(n) < 0.25, 4242, ids)
    grads = rng.standard_normal((n, dim)).astype(np.float32)
    knobs.set("WM_DEDUP_SPLIT_MIN", 1)
    knobs.set("WM_DEDUP_ADAPT", 0)
    knobs.set("WM_DEBUG_SPIN_LIMIT", 3000)
    knobs.set("WM_DEBUG_STALL", gate)
    env, stream = _env()
    d_table = torch.zeros((rows, dim), device="cuda")
    d_ids, d_grads = torch.from_numpy(ids).cuda(), torch.from_numpy(grads).cuda()
    arr = (C.c_float * 6)(0.0, 1e-8, 0.9, 0.999, 0.99, 0.0)
    nu = C.c_int64(-1)
    rc = wmb.lib().wholememory_ext_dedup_apply(d_ids.data_ptr(), wmb.DT_INT64, n, d_grads.data_ptr(), dim, dim, d_table.data_ptr(),
                                               dim, 0, rows, 1, arr, -1.0, None, None, C.byref(nu), env, stream)
    torch.cuda.synchronize()
    assert rc == 4, "expected WHOLEMEMORY_CUDA_ERROR from the call whose sort timed out, got %d" % rc
    assert int(torch.count_nonzero(d_table)) == 0, "the step of a sort that reported a timeout touched the table"
    # the error is reported once; without the stall the same call is served and right
    knobs.unset("WM_DEBUG_STALL")
    knobs.unset("WM_DEBUG_SPIN_LIMIT")
    want, nu_want = _expect(ids, grads, rows, 0)
    got, nu2 = _apply(ids, grads, rows, 0, np.int64)
    assert nu2 == nu_want and got.tobytes() == want.tobytes()


def test_a_timeout_of_an_unsynchronised_call_is_reported_by_the_next_one(gpu_env, knobs):
    """Through the embedding API on one rank nothing synchronises (the reference returns with its kernels queued too): the call
    whose sort timed out returns success, its step is NOT applied, and the next entry into the gradient path returns the error."""
    import torch
    import wholegraph_amd.torch as wgth
    from wholegraph_amd import binding as wmb
    rng = np.random.default_rng(5)
    n_rows, dim, n = 4_000_000, 32, 80000
    emb = wgth.create_embedding(gpu_env, "distributed", "cuda", torch.float32, [n_rows, dim])
    opt = wgth.create_wholememory_optimizer(emb, "sgd", {})
    local, _ = emb.get_embedding_tensor().get_local_tensor()
    local.zero_()
    ids = torch.from_numpy(rng.integers(0, n_rows, n).astype(np.int64)).cuda()
    grads = torch.from_numpy(rng.standard_normal((n, dim)).astype(np.float32)).cuda()
    knobs.set("WM_DEDUP_SPLIT_MIN", 1)
    knobs.set("WM_DEDUP_ADAPT", 0)
    knobs.set("WM_DEBUG_SPIN_LIMIT", 3000)
    knobs.set("WM_DEBUG_STALL", "lookback")
    emb.add_gradients(ids, grads)
    emb.need_apply = True
    opt.step(1.0)                      # queued; the timeout happens on the device after the call has returned
    torch.cuda.synchronize()
    assert int(torch.count_nonzero(local)) == 0, "the step of a sort that reported a timeout touched the table"
    knobs.unset("WM_DEBUG_STALL")
    knobs.unset("WM_DEBUG_SPIN_LIMIT")
    emb.add_gradients(ids, grads)
    emb.need_apply = True
    with pytest.raises(wmb.WholeMemoryError):
        opt.step(1.0)
    torch.cuda.synchronize()
    # reported once: the step after that is applied (row -= 1.0 * sum of its gradients; -1 * g exact, one gradient per row mostly)
    emb.add_gradients(ids, grads)
    emb.need_apply = True
    opt.step(1.0)
    torch.cuda.synchronize()
    assert int(torch.count_nonzero(local)) > 0
    wgth.destroy_wholememory_optimizer(opt)
    wgth.destroy_embedding(emb)
