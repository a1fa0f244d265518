"""The fused duplicate-sum / optimizer step on rows that are NOT a whole number of the tile kernel's wave steps
(csrc/kernels/optim.hip: step_tile_kernel with repeated tail lanes, step_short_kernel on 8- and 4-byte pieces) — reference
exchange_embeddings_nccl_func.cu:76-103 (duplicates summed sequentially in receive order) + embedding_optimizer_func.cu:178-329,
331-421, 583-686, 781-884 (the four update rules).

Round 6: rows of dim % 4 != 0 floats (513, 301 ..., 129, 127: the reference's own test dims; 602, 130, 66) take step_tile_kernel's
RAGGED instantiation — the first dim / 4 sixteen-byte pieces in the usual straight-line batches (gradient rows only 4-byte
aligned), the last dim % 4 floats in a lane-per-run pass; WM_STEP_RAGGED=0 keeps the routes it replaced: the 8-byte-piece
instantiation (global_load_dwordx2 batches, ISA-gated) for dim = 2 mod 4 and the wave-per-run kernel for odd dims.
Row shapes: 75 / 25 / 50 / 250 sixteen-byte pieces (GloVe / word2vec 300 / 100 / 200, 1000 floats), 2408-byte rows (Reddit's
602 floats: 8-byte pieces, on a 16-byte and on a 128-byte row stride), 513 floats (4-byte pieces), 36 and 130 floats (a row
shorter / a little longer than a wave step) and the tile kernel's own 128 / 64 floats. Each must give the oracle's bits: tables,
per-element states, per-row beta powers. Ids: a table of 30 k rows hit by 40 k gradient rows (runs of 1-6 rows in almost every
tile), one id with ~2 k duplicates (the long-run side: its tile takes the predicated path) and a last tile that is not full.
(Round 5 tried a flat-stream kernel for these shapes — 64 runs as one stream of pieces, every lane busy; this file was its
parity test, all green, and it lost 5-19 % to the kernels above: profiles/r05_flat_step_ab_not_kept.txt.)"""
import ctypes as C

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

OPTS = [("sgd", 1, {"weight_decay": 0.05}), ("adam", 2, {"weight_decay": 0.01}), ("rmsprop", 3, {"alpha": 0.9, "weight_decay": 0.01}),
        ("adagrad", 4, {"weight_decay": 0.01})]


def _env():
    from wholegraph_amd.torch.wholegraph_env import get_wholegraph_env_fns, get_stream
    return get_wholegraph_env_fns(), C.c_void_p(get_stream())


@pytest.mark.parametrize("kind,code,params", OPTS, ids=lambda x: str(x))
@pytest.mark.parametrize("dim,align,idt", [(300, 4, np.int64), (100, 4, np.int32), (200, 4, np.int64), (1000, 4, np.int64),
                                           (602, 32, np.int64), (602, 4, np.int32), (513, 4, np.int64), (36, 4, np.int64),
                                           (130, 32, np.int64), (128, 4, np.int64), (64, 4, np.int32),
                                           # the reference's own gradient-apply test dims (wholememory_embedding_gradient_apply_tests.cu:
                                           # 481-501): 127 (stride 128), 129 (stride 132), 392; and the smallest row the 8-byte-piece
                                           # tile kernel takes (66 floats; 602 and 130 above take it too, round 6)
                                           (127, 4, np.int64), (129, 4, np.int32), (392, 4, np.int64), (66, 4, np.int64)])
def test_odd_row_step_bit_exact(gpu_env, kind, code, params, dim, align, idt):
    _run(kind, code, params, dim, align, idt)


@pytest.mark.parametrize("kind,code,params", OPTS, ids=lambda x: str(x))
@pytest.mark.parametrize("dim,align,idt", [(602, 32, np.int64), (130, 32, np.int32), (66, 4, np.int64), (513, 4, np.int64), (129, 4, np.int32)])
def test_odd_row_step_without_the_ragged_kernel(gpu_env, knobs, kind, code, params, dim, align, idt):
    """WM_STEP_RAGGED=0: the routes the ragged tile kernel replaced — the 8-byte-piece tile kernel (dim = 2 mod 4) and the
    wave-per-run kernel on 4-byte pieces (odd dims) — stay reachable and bit-exact"""
    knobs.set("WM_STEP_RAGGED", 0)
    _run(kind, code, params, dim, align, idt)


def _run(kind, code, params, dim, align, idt):
    import torch
    from wholegraph_amd import binding as wmb
    rng = np.random.default_rng(dim * 11 + code)
    local_rows, local_off, n_recv = 30011, 777, 40003
    stride = int(oracle.align_embedding_dim(dim, 4)) if align == 4 else (dim + align - 1) // align * align
    table = np.zeros((local_rows, stride), np.float32)
    table[:, :dim] = rng.standard_normal((local_rows, dim)).astype(np.float32)
    ids = (local_off + rng.integers(0, local_rows, n_recv)).astype(idt)
    ids[::19] = ids[3]  # ~2 k duplicates of one id: the long-run side's
    grads = rng.standard_normal((n_recv, dim)).astype(np.float32)
    p = dict(weight_decay=0.0, epsilon=1e-8, beta1=0.9, beta2=0.999, alpha=0.99, adam_w=0.0)
    p.update(params)
    ref_opt = oracle.Optimizer(kind, local_rows, stride, **params)
    d_table = torch.from_numpy(table.copy()).cuda()
    d_pe = d_pr = None
    if kind == "adam":
        d_pe = torch.zeros((local_rows, 2 * stride), device="cuda")
        d_pr = torch.ones((local_rows, 2), device="cuda")
    elif kind in ("adagrad", "rmsprop"):
        d_pe = torch.zeros((local_rows, stride), device="cuda")
    d_ids, d_grads = torch.from_numpy(ids).cuda(), torch.from_numpy(grads).cuda()
    arr = (C.c_float * 6)(p["weight_decay"], p["epsilon"], p["beta1"], p["beta2"], p["alpha"], p["adam_w"])
    env, stream = _env()
    ref_table = table.copy()
    uniq, dg = oracle.dedup_grads(ids, grads)
    for step in range(2):
        nu = C.c_int64(-1)
        wmb.check(wmb.lib().wholememory_ext_dedup_apply(
            d_ids.data_ptr(), wmb.DT_INT if idt == np.int32 else wmb.DT_INT64, n_recv, d_grads.data_ptr(), dim, dim,
            d_table.data_ptr(), stride, local_off, local_rows, code, arr, 0.03,
            d_pe.data_ptr() if d_pe is not None else None, d_pr.data_ptr() if d_pr is not None else None, C.byref(nu),
            env, stream))
        torch.cuda.synchronize()
        assert nu.value == len(uniq)
        ref_opt.step(uniq, dg, ref_table, stride, local_off, dim, 0.03)
        assert d_table.cpu().numpy().tobytes() == ref_table.tobytes(), "%s step %d: table differs from the oracle" % (kind, step)
    if kind != "sgd":
        assert d_pe.cpu().numpy().tobytes() == ref_opt.per_element.tobytes()
    if kind == "adam":
        assert d_pr.cpu().numpy().tobytes() == ref_opt.per_row.tobytes()
