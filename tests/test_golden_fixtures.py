"""CPU: the oracle (oracle/wm_oracle.c) against the frozen numpy fixtures of tests/golden/gen_fixtures.py — bucketing
(bucket_ids_func.cu:51-87, exchange_ids_nccl_func.cu:42-92), gather / scatter with the dtype-cast matrix
(embedding_test_utils.cu:197-238,401-431), dedup + optimizer steps (exchange_embeddings_nccl_func.cu:76-103,
embedding_optimizer_func.cu:212-223,385-418,644-657,842-855). Bit-exact. The GPU twin
(tests/test_golden_fixtures_gpu.py) compares the HIP path with the same files WITHOUT the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def test_fixture_files_are_what_the_generator_writes(tmp_path):
    """the committed .npz files are exactly gen_fixtures.py's output (numpy only — it must not import the oracle)"""
    src = open(os.path.join(GOLDEN, "gen_fixtures.py")).read()
    assert "import oracle" not in src and "wholegraph_amd" not in src.split('"""')[2]
    import shutil
    work = tmp_path / "golden"
    work.mkdir()
    shutil.copy(os.path.join(GOLDEN, "gen_fixtures.py"), work / "gen_fixtures.py")
    subprocess.check_call([sys.executable, "-W", "ignore", str(work / "gen_fixtures.py")], stdout=subprocess.DEVNULL)
    for f in ("bucketing.npz", "gather_scatter.npz", "optimizers.npz"):
        a, b = np.load(work / f), load(f)
        assert sorted(a.files) == sorted(b.files), f
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].tobytes() == b[k].tobytes(), (f, k)


def test_oracle_bucketing_matches_fixtures():
    z = load("bucketing.npz")
    for k in range(int(z["n_cases"])):
        ids, offs = z["c%d_ids" % k], z["c%d_offsets" % k]
        assert np.array_equal(oracle.bucket_counts(ids, offs), z["c%d_counts" % k]), k
        s, raw = oracle.sort_ids(ids)
        assert np.array_equal(s, z["c%d_sorted_ids" % k]) and np.array_equal(raw, z["c%d_raw_indices" % k]), k


def test_oracle_gather_scatter_matches_fixtures():
    z = load("gather_scatter.npz")
    seen = set()
    for k in range(int(z["n_cases"])):
        kind, tdt, odt, dim, stride = z["c%d_meta" % k]
        dim, stride = int(dim), int(stride)
        seen.add(kind)
        table, idx, exp = z["c%d_table" % k], z["c%d_idx" % k], z["c%d_expected" % k]
        if kind == "scatter":
            tab = oracle.ShardedTable.from_full(np.zeros(exp.shape, dtype=exp.dtype), 3 if k % 2 else 1)
            oracle.scatter(table, idx, tab)
            got = np.concatenate([tab.shards[r][: int(tab.entry_offsets[r + 1] - tab.entry_offsets[r])]
                                  for r in range(len(tab.shards))])
            assert got.tobytes() == exp.tobytes(), (k, kind, tdt, odt, dim)
        else:
            tab = oracle.ShardedTable.from_full(table, 3 if k % 2 else 1)
            tab.dim = dim
            out = np.full(exp.shape, 9, dtype=exp.dtype)
            oracle.gather(tab, idx, out)
            assert out.tobytes() == exp.tobytes(), (k, kind, tdt, odt, dim)
    assert seen == {"gather", "gather_neg_pad", "gather_random", "scatter"}


def test_oracle_dedup_matches_fixtures():
    z = load("optimizers.npz")
    for s in range(int(z["n_steps"])):
        u, dg = oracle.dedup_grads(z["ids_%d" % s], z["grads_%d" % s])
        assert np.array_equal(u, z["unique_%d" % s]) and dg.tobytes() == z["dedup_grads_%d" % s].tobytes()


@pytest.mark.parametrize("world", [1, 3])
def test_oracle_optimizers_match_fixtures(world):
    """world 3 too: one requester, three owners — the per-id arrival order is unchanged, so the same bits must come out"""
    z = load("optimizers.npz")
    table0, lr, touched = z["table0"], float(z["lr"]), z["touched"]
    n_rows, dim = table0.shape
    for k in range(int(z["n_cases"])):
        kind = str(z["o%d_kind" % k])
        wd, eps, b1, b2, alpha, adam_w = [float(x) for x in z["o%d_params" % k]]
        params = {"weight_decay": wd, "epsilon": eps, "beta1": b1, "beta2": b2, "alpha": alpha, "adam_w": adam_w}
        tab = oracle.ShardedTable.from_full(table0.copy(), world)
        cnt = [int(tab.entry_offsets[r + 1] - tab.entry_offsets[r]) for r in range(world)]
        opts = [oracle.Optimizer(kind, cnt[r], dim, **params) for r in range(world)]
        full = lambda: np.concatenate([tab.shards[r][:cnt[r]] for r in range(world)])
        for s in range(int(z["n_steps"])):
            rank_idx = [z["ids_%d" % s]] + [np.zeros(0, np.int64)] * (world - 1)
            rank_g = [z["grads_%d" % s]] + [np.zeros((0, dim), np.float32)] * (world - 1)
            oracle.gradient_apply(tab, opts, rank_idx, rank_g, lr)
            key = "o%d_table_%d" % (k, s)
            if key in z.files:
                assert full()[touched].tobytes() == z[key].tobytes(), (kind, s)
        untouched = np.setdiff1d(np.arange(n_rows), touched)
        assert np.array_equal(full()[untouched], table0[untouched])
        if kind in ("adam", "adagrad", "rmsprop"):
            pe = np.concatenate([opts[r].per_element[:cnt[r]] for r in range(world)])
            assert pe[touched, :dim].tobytes() == z["o%d_state0" % k].tobytes(), kind
        if kind == "adam":
            assert pe[touched, dim:2 * dim].tobytes() == z["o%d_state1" % k].tobytes()
            pr = np.concatenate([opts[r].per_row[:cnt[r]] for r in range(world)])
            assert pr[touched].tobytes() == z["o%d_per_row" % k].tobytes()
