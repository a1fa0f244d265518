#!/usr/bin/env python
"""Randomised owner-side "dedup + optimizer step" (wholememory_ext_dedup_apply) against the CPU oracle, bit for bit:
optimizer kinds, dims 1..520 (odd, padded strides), int32 / int64 ids, run-length mixes that put runs on every path of
optim.hip (in-wave folds up to 256 rows, the LDS-DMA long-run kernel across tile and order-chunk boundaries, its fallback
for rows that are not whole 16-byte pieces), two steps each. usage: fuzz_optim.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
import oracle
from wholegraph_amd import binding as wmb
import wholegraph_amd.torch as wgth
from wholegraph_amd.torch.wholegraph_env import get_wholegraph_env_fns, get_stream

torch.cuda.set_device(0)
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_ERROR))
comm = wgth.create_group_communicator(1)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
KINDS = [("sgd", 1, {}), ("sgd", 1, {"weight_decay": 0.05}), ("adam", 2, {"weight_decay": 0.01}),
         ("adam", 2, {"weight_decay": 0.02, "adam_w": 1.0}), ("rmsprop", 3, {"alpha": 0.9}), ("adagrad", 4, {})]
bad = 0
for case in range(cases):
    kind, code, params = KINDS[rng.integers(len(KINDS))]
    dim = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 31, 32, 33, 48, 64, 96, 100, 127, 128, 129, 160, 256, 300, 513, 520]))
    stride = int(oracle.align_embedding_dim(dim, 4))
    local_rows = int(rng.integers(1, 6000))
    local_off = int(rng.integers(0, 1 << 20))
    idt = np.int32 if rng.random() < 0.5 else np.int64
    # run-length mix: background ids + a few hot ids of chosen lengths
    n_bg = int(rng.choice([0, 10, 1000, 20000]))
    hot = [int(x) for x in rng.choice([2, 33, 200, 256, 257, 300, 1000, 2048, 2049, 4100, 9000], size=rng.integers(0, 4))]
    parts = [rng.integers(0, local_rows, n_bg)] + [np.full(h, rng.integers(0, local_rows)) for h in hot]
    ids = np.concatenate(parts)
    rng.shuffle(ids)
    ids = (ids + local_off).astype(idt)
    # ids that address no row of this shard (negative, below it, past it) with junk gradient rows, in a third of the cases:
    # the sort drops them (backend.hpp: dedup_ids), the oracle never sees them
    junk_pos = None
    if len(ids) and rng.random() < 0.33:
        n_junk = int(rng.integers(1, 40))
        info = np.iinfo(idt)
        pool = [-1, -5, info.min, info.max, local_off + local_rows, local_off + local_rows + 7]
        if local_off > 0:
            pool += [local_off - 1, 0]
        junk = np.array([pool[i] for i in rng.integers(0, len(pool), n_junk)], dtype=idt)
        total = len(ids) + n_junk
        junk_pos = np.sort(rng.choice(total, n_junk, replace=False))
        keep = np.ones(total, dtype=bool)
        keep[junk_pos] = False
        mixed = np.empty(total, dtype=idt)
        mixed[keep], mixed[junk_pos] = ids, junk
        ids = mixed
    n = len(ids)
    grad_stride = dim + int(rng.choice([0, 0, 0, 4]))
    grads_buf = rng.standard_normal((max(n, 1), grad_stride)).astype(np.float32)
    if junk_pos is not None:
        grads_buf[junk_pos] = 1e30
    good = np.ones(n, dtype=bool)
    if junk_pos is not None:
        good[junk_pos] = False
    grads = np.ascontiguousarray(grads_buf[:n, :dim][good])
    good_ids = ids[good]
    desc = "case %d: %s %s dim %d stride %d grad_stride %d rows %d n %d hot %s %s" % (
        case, kind, params, dim, stride, grad_stride, local_rows, n, hot, np.dtype(idt).name) + (" +junk" if junk_pos is not None else "")
    table = np.zeros((local_rows, stride), np.float32)
    table[:, :dim] = rng.standard_normal((local_rows, dim)).astype(np.float32)
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
    d_ids = torch.from_numpy(ids).cuda() if n else torch.zeros(1, dtype=torch.int64, device="cuda")
    d_grads = torch.from_numpy(grads_buf).cuda()
    arr = (C.c_float * 6)(p["weight_decay"], p["epsilon"], p["beta1"], p["beta2"], p["alpha"], p["adam_w"])
    ref_table = table.copy()
    ok = True
    try:
        for step in range(2):
            nu = C.c_int64(-1)
            wmb.check(wmb.lib().wholememory_ext_dedup_apply(
                d_ids.data_ptr(), wmb.DT_INT if idt == np.int32 else wmb.DT_INT64, n, d_grads.data_ptr(), grad_stride, dim,
                d_table.data_ptr(), stride, local_off, local_rows, code, arr, 0.03,
                d_pe.data_ptr() if d_pe is not None else None, d_pr.data_ptr() if d_pr is not None else None, C.byref(nu),
                get_wholegraph_env_fns(), C.c_void_p(get_stream())))
            torch.cuda.synchronize()
            uniq, dg = oracle.dedup_grads(good_ids, grads) if len(good_ids) else (good_ids[:0], grads[:0])
            ref_opt.step(uniq, dg, ref_table, stride, local_off, dim, 0.03)
            ok = ok and nu.value == len(uniq) and d_table.cpu().numpy().tobytes() == ref_table.tobytes()
        if kind != "sgd":
            ok = ok and d_pe.cpu().numpy().tobytes() == ref_opt.per_element.tobytes()
        if kind == "adam":
            ok = ok and d_pr.cpu().numpy().tobytes() == ref_opt.per_row.tobytes()
    except Exception as ex:  # noqa
        ok = False
        print("ERROR", repr(ex)[:300], flush=True)
    if not ok:
        bad += 1
        print("MISMATCH", desc, flush=True)
print("cases %d, failures %d" % (cases, bad))
