"""GPU: the HIP path through the C ABI compared DIRECTLY with the frozen numpy fixtures (tests/golden/*.npz, written by
tests/golden/gen_fixtures.py) — no oracle in the loop. Bit-exact.
  bucketing       wholememory_ext_bucket_ids vs counts / the reference's sorted order (bucket_ids_func.cu:51-87,
                  exchange_ids_nccl_func.cu:42-92)
  gather/scatter  wholememory_gather / wholememory_scatter on CONTINUOUS, CHUNKED and DISTRIBUTED tensors vs the reference
                  tests' closed forms and cast rule (embedding_test_utils.cu:197-238,401-431)
  optimizers      wholememory_ext_dedup_apply and the WholeMemoryEmbedding training surface vs the numpy restatement of
                  the reference kernels (embedding_optimizer_func.cu:212-223,385-418,644-657,842-855)"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def _env():
    from wholegraph_amd.torch.wholegraph_env import get_wholegraph_env_fns, get_stream
    return get_wholegraph_env_fns(), C.c_void_p(get_stream())


def _tt(np_dtype):
    import torch
    return {np.dtype(np.float32): torch.float32, np.dtype(np.float16): torch.float16, np.dtype(np.float64): torch.float64,
            np.dtype(np.int8): torch.int8, np.dtype(np.int16): torch.int16, np.dtype(np.int32): torch.int32,
            np.dtype(np.int64): torch.int64}[np.dtype(np_dtype)]


def test_bucketing_matches_fixtures(gpu_env):
    import torch
    from wholegraph_amd import binding as wmb
    z = load("bucketing.npz")
    env, stream = _env()
    for k in range(int(z["n_cases"])):
        ids, offs, counts = z["c%d_ids" % k], z["c%d_offsets" % k], z["c%d_counts" % k]
        n, W = len(ids), len(offs) - 1
        d_idx = torch.from_numpy(ids).cuda() if n else torch.zeros(1, dtype=_tt(ids.dtype), device="cuda")
        d_off = torch.from_numpy(offs.astype(np.int64)).cuda()
        d_cnt = torch.full((W,), -1, dtype=torch.int64, device="cuda")
        d_ids = torch.zeros(max(n, 1), dtype=d_idx.dtype, device="cuda")
        d_raw = torch.zeros(max(n, 1), dtype=torch.int64, device="cuda")
        wmb.check(wmb.lib().wholememory_ext_bucket_ids(d_idx.data_ptr(), wmb.DT_INT if ids.dtype == np.int32 else wmb.DT_INT64,
                                                       n, d_off.data_ptr(), W, d_cnt.data_ptr(), d_ids.data_ptr(),
                                                       d_raw.data_ptr(), env, stream))
        torch.cuda.synchronize()
        assert np.array_equal(d_cnt.cpu().numpy(), counts), "case %d: counts" % k
        got_ids, got_raw = d_ids.cpu().numpy()[:n], d_raw.cpu().numpy()[:n]
        # the product groups by owner and keeps the original order inside a group; a stable sort by id
        # of the valid part must reproduce the reference's order — ids AND payload — bit for bit
        nvalid = int(counts.sum())
        o = np.argsort(got_ids[:nvalid].astype(np.int64), kind="stable")
        assert np.array_equal(got_ids[:nvalid][o], z["c%d_sorted_ids" % k][:nvalid]), "case %d: grouped ids" % k
        assert np.array_equal(got_raw[:nvalid][o], z["c%d_raw_indices" % k][:nvalid]), "case %d: raw_indices" % k
        # owner segments are contiguous and in owner order
        seg = np.concatenate([[0], np.cumsum(counts)])
        for r in range(W):
            s = got_ids[int(seg[r]):int(seg[r + 1])].astype(np.int64)
            assert s.size == 0 or (s.min() >= int(offs[r]) and s.max() < int(offs[r + 1])), "case %d owner %d" % (k, r)
        assert np.all(got_ids[nvalid:] < 0) and sorted(got_raw.tolist()) == list(range(n))


@pytest.mark.parametrize("mt", ["continuous", "chunked", "distributed"])
def test_gather_scatter_matches_fixtures(gpu_env, mt):
    import torch
    import wholegraph_amd.torch as wgth
    from wholegraph_amd import binding as wmb
    from wholegraph_amd.torch.wholegraph_env import wrap_torch_tensor
    z = load("gather_scatter.npz")
    env, stream = _env()
    n_done = 0
    for k in range(int(z["n_cases"])):
        kind, tdt, odt, dim, stride = z["c%d_meta" % k]
        dim, stride = int(dim), int(stride)
        table, idx, exp = z["c%d_table" % k], z["c%d_idx" % k], z["c%d_expected" % k]
        d_idx = wrap_torch_tensor(torch.from_numpy(idx).cuda())
        if kind == "scatter":
            n_rows = exp.shape[0]
            root = wgth.create_wholememory_tensor(gpu_env, mt, "cuda", [n_rows, dim], _tt(exp.dtype), [dim, 1])
            local, _ = root.get_local_tensor()
            local.zero_()
            d_in = wrap_torch_tensor(torch.from_numpy(table).cuda())
            wmb.check(wmb.lib().wholememory_scatter(d_in.handle, d_idx.handle, root.wmb_tensor, env, stream, -1))
            torch.cuda.synchronize()
            assert local.cpu().numpy().tobytes() == exp.tobytes(), "case %d %s %s<-%s dim %d (%s)" % (k, kind, tdt, odt, dim, mt)
        else:
            n_rows = table.shape[0]
            root = wgth.create_wholememory_tensor(gpu_env, mt, "cuda", [n_rows, stride], _tt(table.dtype), [stride, 1])
            local, _ = root.get_local_tensor()
            local.copy_(torch.from_numpy(table).cuda())
            view = root.get_sub_tensor([0, 0], [n_rows, dim]) if stride != dim else root
            out = torch.full(exp.shape, 9, dtype=_tt(exp.dtype), device="cuda")
            d_out = wrap_torch_tensor(out)
            wmb.check(wmb.lib().wholememory_gather(view.wmb_tensor, d_idx.handle, d_out.handle, env, stream, -1))
            torch.cuda.synchronize()
            assert out.cpu().numpy().tobytes() == exp.tobytes(), "case %d %s %s->%s dim %d (%s)" % (k, kind, tdt, odt, dim, mt)
            if view is not root:
                wgth.destroy_wholememory_tensor(view)
        wgth.destroy_wholememory_tensor(root)
        n_done += 1
    assert n_done == int(z["n_cases"]) >= 140


CODE = {"sgd": 1, "adam": 2, "rmsprop": 3, "adagrad": 4}


def test_dedup_apply_matches_fixtures(gpu_env):
    """the owner-side stage on raw pointers: sorted-order duplicate sum fused with the optimizer statement sequence"""
    import torch
    from wholegraph_amd import binding as wmb
    z = load("optimizers.npz")
    table0, lr, touched = z["table0"], float(z["lr"]), z["touched"]
    n_rows, dim = table0.shape
    stride, off = 128, 77000            # rows padded to 16 B; the shard starts at global row 77000
    env, stream = _env()
    for k in range(int(z["n_cases"])):
        kind = str(z["o%d_kind" % k])
        params = (C.c_float * 6)(*[float(x) for x in z["o%d_params" % k]])
        padded = np.zeros((n_rows, stride), np.float32)
        padded[:, :dim] = table0
        d_table = torch.from_numpy(padded).cuda()
        d_pe = d_pr = None
        if kind == "adam":
            d_pe, d_pr = torch.zeros((n_rows, 2 * stride), device="cuda"), torch.ones((n_rows, 2), device="cuda")
        elif kind != "sgd":
            d_pe = torch.zeros((n_rows, stride), device="cuda")
        for s in range(int(z["n_steps"])):
            ids, grads = z["ids_%d" % s], z["grads_%d" % s]
            d_ids, d_g = torch.from_numpy(ids + off).cuda(), torch.from_numpy(grads).cuda()
            nu = C.c_int64(-1)
            wmb.check(wmb.lib().wholememory_ext_dedup_apply(
                d_ids.data_ptr(), wmb.DT_INT64, len(ids), d_g.data_ptr(), dim, dim, d_table.data_ptr(), stride, off, n_rows,
                CODE[kind], params, lr, d_pe.data_ptr() if d_pe is not None else None,
                d_pr.data_ptr() if d_pr is not None else None, C.byref(nu), env, stream))
            torch.cuda.synchronize()
            assert nu.value == len(z["unique_%d" % s])
            key = "o%d_table_%d" % (k, s)
            if key in z.files:
                assert d_table.cpu().numpy()[touched, :dim].tobytes() == z[key].tobytes(), (kind, s)
        got = d_table.cpu().numpy()
        untouched = np.setdiff1d(np.arange(n_rows), touched)
        assert np.array_equal(got[untouched, :dim], table0[untouched]) and not got[:, dim:].any()
        if kind != "sgd":
            pe = d_pe.cpu().numpy()
            assert pe[touched, :dim].tobytes() == z["o%d_state0" % k].tobytes(), kind
            assert not pe[untouched].any()
        if kind == "adam":
            assert pe[touched, stride:stride + dim].tobytes() == z["o%d_state1" % k].tobytes()
            assert d_pr.cpu().numpy()[touched].tobytes() == z["o%d_per_row" % k].tobytes()


@pytest.mark.parametrize("mt", ["chunked", "distributed"])
def test_embedding_training_matches_fixtures(gpu_env, mt):
    """the same fixtures through the product surface: WholeMemoryEmbedding.add_gradients + WholeMemoryOptimizer.step
    (wholememory_embedding_gather_gradient_apply: bucketing, dedup, step) and the optimizer-state tensors"""
    import torch
    import wholegraph_amd.torch as wgth
    z = load("optimizers.npz")
    table0, lr, touched = z["table0"], float(z["lr"]), z["touched"]
    n_rows, dim = table0.shape
    names = {"sgd": "sgd", "adam": "adam", "adagrad": "adagrad", "rmsprop": "rmsprop"}
    for k in range(int(z["n_cases"])):
        kind = str(z["o%d_kind" % k])
        wd, eps, b1, b2, alpha, adam_w = [float(x) for x in z["o%d_params" % k]]
        emb = wgth.create_embedding(gpu_env, mt, "cuda", torch.float32, [n_rows, dim])
        local, _ = emb.get_embedding_tensor().get_local_tensor()
        local.copy_(torch.from_numpy(table0).cuda())
        allp = {"weight_decay": wd, "epsilon": eps, "beta1": b1, "beta2": b2, "alpha": alpha, "adam_w": adam_w}
        accepted = {"sgd": ["weight_decay"], "adam": ["weight_decay", "epsilon", "beta1", "beta2", "adam_w"],
                    "adagrad": ["weight_decay", "epsilon"], "rmsprop": ["weight_decay", "epsilon", "alpha"]}[kind]
        opt = wgth.create_wholememory_optimizer(emb, names[kind], {n: allp[n] for n in accepted})
        for s in range(int(z["n_steps"])):
            emb.add_gradients(torch.from_numpy(z["ids_%d" % s]).cuda(), torch.from_numpy(z["grads_%d" % s]).cuda())
            emb.need_apply = True
            opt.step(lr)
            torch.cuda.synchronize()
            key = "o%d_table_%d" % (k, s)
            if key in z.files:
                assert local.cpu().numpy()[touched].tobytes() == z[key].tobytes(), (kind, s, mt)
        untouched = np.setdiff1d(np.arange(n_rows), touched)
        assert np.array_equal(local.cpu().numpy()[untouched], table0[untouched])
        state0 = {"adam": "m", "adagrad": "state_sum", "rmsprop": "v"}.get(kind)
        if state0:
            t, _ = emb.get_optimizer_state(state0).get_local_tensor()
            assert t.cpu().numpy()[touched].tobytes() == z["o%d_state0" % k].tobytes(), (kind, mt)
        if kind == "adam":
            t, _ = emb.get_optimizer_state("v").get_local_tensor()
            assert t.cpu().numpy()[touched].tobytes() == z["o%d_state1" % k].tobytes()
            t, _ = emb.get_optimizer_state("beta12t").get_local_tensor()
            assert t.cpu().numpy()[touched].tobytes() == z["o%d_per_row" % k].tobytes()
        wgth.destroy_wholememory_optimizer(opt)
        wgth.destroy_embedding(emb)
