"""Worker for tests/test_distributed_cpu.py: one rank of a world_size-N gloo job driving the PRODUCT's
multi-rank orchestration (libwholegraph.so ops.cpp / embedding.cpp, Python wholegraph_amd.torch) with the
device seam replaced by the CPU test backend (oracle/test_backend.cpp) and collectives provided by
torch.distributed/gloo. Every rank recomputes the all-rank expectation with the oracle and checks its own part."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MODE = sys.argv[4] if len(sys.argv) > 4 else "cpu"
HIP_MODE = MODE.startswith("hip")   # real kernels on cuda:0, collectives still over gloo
HIER_MODE = MODE.endswith("-hier")  # HIERARCHY tables on a pretended multi-node layout (WM_LOCAL_SIZE ranks per node)
FUZZ_MODE = MODE.endswith("-fuzz")  # random shapes / partitions / dtypes (FUZZ_SEED, FUZZ_CASES), same on every rank
# "hip-rccl": one rank per GPU, torch.distributed backend nccl, the library's own RCCL communicator underneath (the product
# configuration). On a one-GPU box it runs at world 1 with WM_FORCE_RCCL=1 + WM_EXCHANGE_SELF=1 (set by the test), which
# sends the rank's own segment through rccl_provider like a peer's.
RCCL_MODE = MODE == "hip-rccl"
if not HIP_MODE:
    os.environ["WHOLEGRAPH_AMD_TESTING"] = "1"

import numpy as np
import torch
import torch.distributed as dist

import oracle
from wholegraph_amd import binding as wmb
import wholegraph_amd.torch as wgth


def _reload_knobs():
    """the library reads each WM_* / WG_* variable once: say so after changing one mid-process"""
    wmb.reload_knobs()


def install_test_backend():
    tb = C.CDLL(os.path.join(ROOT, "oracle", "libwm_test_backend.so"))
    tb.wm_test_backend.restype = C.c_void_p
    wmb.check(wmb.lib().wm_testing_install_backend(C.c_void_p(tb.wm_test_backend())))
    assert wmb.lib().wholememory_ext_backend_name().startswith(b"oracle-test-backend")
    return tb


def dev(t):
    """op inputs/outputs live where the installed backend keeps device memory"""
    return t.cuda() if HIP_MODE else t


def host(t):
    return t.cpu() if HIP_MODE else t


def shard_views(emb_tensor, rank):
    local, start = emb_tensor.get_local_tensor(host_view=False)
    return local, start


def gather_all_local(emb_tensor):
    """this rank's rows as numpy (through the local view)"""
    local, start = emb_tensor.get_local_tensor()
    return local.numpy().copy(), start


def scenario_gather_scatter(comm, rank, world, mt, n_rows, dim, tdt, odt, idt, entries, loc="cuda"):
    tt = {np.float32: torch.float32, np.float16: torch.float16, np.int32: torch.int32, np.int64: torch.int64}
    stride = dim + (3 if dim % 4 else 0)
    wm = wgth.create_wholememory_tensor(comm, mt, loc, [n_rows, stride], tt[tdt], [stride, 1], entries)
    hv = loc == "cpu"   # host-located tables are filled / checked through their host view
    view = wm.get_sub_tensor([0, 0], [n_rows, dim]) if stride != dim else wm
    full = oracle.fill_closed_form(tdt, 0, n_rows, dim, stride)
    tab = oracle.ShardedTable.from_full(full, world, entries)
    tab.dim = dim
    local, start = wm.get_local_tensor(host_view=hv)
    assert start == int(tab.entry_offsets[rank])
    cnt = int(tab.entry_offsets[rank + 1] - tab.entry_offsets[rank])
    assert local.shape[0] == cnt
    if cnt:
        local.copy_(torch.from_numpy(full[start:start + cnt]) if hv else dev(torch.from_numpy(full[start:start + cnt])))
    if HIP_MODE:
        torch.cuda.synchronize()
    comm.barrier()
    rngs = [np.random.default_rng(100 + r) for r in range(world)]
    rank_idx = []
    for r in range(world):
        n = 0 if (r == 1 and n_rows % 2 == 0) else 500 + 37 * r   # one rank may have nothing to ask
        ix = rngs[r].integers(0, n_rows, n).astype(idt)
        if n:
            ix[::29] = -1
            ix[:8] = ix[0]
        rank_idx.append(ix)
    exp = oracle.distributed_gather(tab, rank_idx, odt, out_init=[np.full((len(ix), dim), 5, dtype=odt) for ix in rank_idx])
    out = dev(torch.full((len(rank_idx[rank]), dim), 5, dtype=tt[odt]))
    from wholegraph_amd.torch.wholegraph_env import wrap_torch_tensor, get_wholegraph_env_fns, get_stream
    wi, wo = wrap_torch_tensor(dev(torch.from_numpy(rank_idx[rank]))), wrap_torch_tensor(out)
    launches0 = wmb.lib().wholememory_ext_distributed_gather_launches()
    wmb.check(wmb.lib().wholememory_gather(view.wmb_tensor, wi.handle, wo.handle, get_wholegraph_env_fns(),
                                           C.c_void_p(get_stream()), -1))
    if HIP_MODE:
        torch.cuda.synchronize()
    assert host(out).numpy().tobytes() == exp[rank].tobytes(), "%s gather mismatch on rank %d" % (mt, rank)
    if mt == "distributed" and world > 1 and world <= 16 and loc == "cuda":
        # one row kernel per exchange chunk and side (+ the rank's own rows, + the two chunk-major copies), whatever the
        # number of ranks — and the per-peer launches of rounds 2-4 (WM_EXCHANGE_PER_PEER=1) give the same rows
        chunks = int(os.environ.get("WM_EXCHANGE_CHUNKS", "1"))
        folded = wmb.lib().wholememory_ext_distributed_gather_launches() - launches0
        assert folded <= 2 * chunks + 3, "distributed gather queued %d kernels with %d chunks" % (folded, chunks)
        os.environ["WM_EXCHANGE_PER_PEER"] = "1"
        _reload_knobs()
        out2 = dev(torch.full((len(rank_idx[rank]), dim), 5, dtype=tt[odt]))
        wo2 = wrap_torch_tensor(out2)
        launches1 = wmb.lib().wholememory_ext_distributed_gather_launches()
        wmb.check(wmb.lib().wholememory_gather(view.wmb_tensor, wi.handle, wo2.handle, get_wholegraph_env_fns(),
                                               C.c_void_p(get_stream()), -1))
        if HIP_MODE:
            torch.cuda.synchronize()
        per_peer = wmb.lib().wholememory_ext_distributed_gather_launches() - launches1
        del os.environ["WM_EXCHANGE_PER_PEER"]
        _reload_knobs()
        assert host(out2).numpy().tobytes() == exp[rank].tobytes(), "per-peer gather mismatch on rank %d" % rank
        if n_rows >= 500 and min(len(ix) for ix in rank_idx) >= 100:   # (tiny batches: most (peer, chunk) pairs are empty)
            assert per_peer >= folded, (per_peer, folded)
            if world >= 3 and chunks > 1 and os.environ.get("WM_EXCHANGE_SELF") != "1":
                # (the second figures: the de-duplicating route, whose rows are received in place — no reorder side)
                assert per_peer in (1 + 2 * (world - 1) * chunks, 1 + (world - 1) * chunks), (per_peer, folded)
                assert folded in (3 + 2 * chunks, 2 + chunks), (per_peer, folded)
    # scatter: every rank writes rows of its own ids (closed-form rows: duplicates agree), then everyone checks
    comm.barrier()
    if cnt:
        local.zero_()
    if HIP_MODE:
        torch.cuda.synchronize()
    comm.barrier()
    ref = oracle.ShardedTable.from_full(np.zeros((n_rows, stride), dtype=tdt), world, entries)
    ref.dim = dim
    src = oracle.fill_closed_form(odt, 0, n_rows, dim)
    for r in range(world):
        rows = np.zeros((len(rank_idx[r]), dim), dtype=odt)
        v = rank_idx[r] >= 0
        rows[v] = src[rank_idx[r][v].astype(np.int64)]
        oracle.scatter(rows, rank_idx[r], ref)
        if r == rank:
            wr, wx = wrap_torch_tensor(dev(torch.from_numpy(rows))), wrap_torch_tensor(dev(torch.from_numpy(rank_idx[r])))
            s0 = wmb.lib().wholememory_ext_distributed_scatter_launches()
            wmb.check(wmb.lib().wholememory_scatter(wr.handle, wx.handle, view.wmb_tensor, get_wholegraph_env_fns(),
                                                    C.c_void_p(get_stream()), -1))
            if HIP_MODE:
                torch.cuda.synchronize()
            if mt == "distributed" and world > 1 and world <= 16 and loc == "cuda":
                # one row kernel per exchange chunk and side (+ the rank's own rows, + the two chunk-major copies), whatever
                # the number of ranks (round 6; the per-peer launches of rounds 1-5 are checked against it below)
                chunks = int(os.environ.get("WM_EXCHANGE_CHUNKS", "1"))
                folded = wmb.lib().wholememory_ext_distributed_scatter_launches() - s0
                assert folded <= 2 * chunks + 3, "distributed scatter queued %d kernels with %d chunks" % (folded, chunks)
                if (world >= 3 and chunks > 1 and os.environ.get("WM_EXCHANGE_SELF") != "1" and n_rows >= 500
                        and min(len(ix) for ix in rank_idx) >= 100 and entries is None):
                    assert folded == 2 * chunks + 3, (folded, chunks)
    comm.barrier()
    if cnt:
        assert (local if hv else host(local)).numpy().tobytes() == ref.shards[rank][:cnt].tobytes(), \
            "%s/%s scatter mismatch on rank %d" % (mt, loc, rank)
    comm.barrier()
    if mt == "distributed" and world > 2 and world <= 16 and loc == "cuda":
        # the same scatter with the per-peer launches of rounds 1-5: the same table, 2 (W - 1) C + 1 kernels
        if cnt:
            local.zero_()
        if HIP_MODE:
            torch.cuda.synchronize()
        comm.barrier()
        os.environ["WM_EXCHANGE_PER_PEER"] = "1"
        _reload_knobs()
        rows = np.zeros((len(rank_idx[rank]), dim), dtype=odt)
        v = rank_idx[rank] >= 0
        rows[v] = src[rank_idx[rank][v].astype(np.int64)]
        wr, wx = wrap_torch_tensor(dev(torch.from_numpy(rows))), wrap_torch_tensor(dev(torch.from_numpy(rank_idx[rank])))
        s1 = wmb.lib().wholememory_ext_distributed_scatter_launches()
        wmb.check(wmb.lib().wholememory_scatter(wr.handle, wx.handle, view.wmb_tensor, get_wholegraph_env_fns(),
                                                C.c_void_p(get_stream()), -1))
        if HIP_MODE:
            torch.cuda.synchronize()
        per_peer = wmb.lib().wholememory_ext_distributed_scatter_launches() - s1
        del os.environ["WM_EXCHANGE_PER_PEER"]
        _reload_knobs()
        comm.barrier()
        if cnt:
            assert (local if hv else host(local)).numpy().tobytes() == ref.shards[rank][:cnt].tobytes(), \
                "%s/%s per-peer scatter mismatch on rank %d" % (mt, loc, rank)
        chunks = int(os.environ.get("WM_EXCHANGE_CHUNKS", "1"))
        assert per_peer <= 1 + 2 * world * chunks, (per_peer, world, chunks)   # (loopback: the own segment travels too)
        if (chunks > 1 and os.environ.get("WM_EXCHANGE_SELF") != "1" and n_rows >= 500
                and min(len(ix) for ix in rank_idx) >= 100 and entries is None):
            assert per_peer == 1 + 2 * (world - 1) * chunks, (per_peer, world, chunks)
        comm.barrier()
    if view is not wm:
        wgth.destroy_wholememory_tensor(view)
    wgth.destroy_wholememory_tensor(wm)


def scenario_gather_skewed(comm, rank, world, idt, entries=None):
    """A batch dominated by a few hot rows (Zipf): with WM_GATHER_DEDUP unset the ranks estimate the duplicate share, agree
    through the counts exchange and take the de-duplicating route (distinct ids sorted, owner segments by binary search,
    rows received in place, local expansion); results are those of the plain route."""
    n_rows, dim = 3001, 24
    wm = wgth.create_wholememory_tensor(comm, "distributed", "cuda", [n_rows, dim], torch.float32, [dim, 1], entries)
    full = np.random.default_rng(12).standard_normal((n_rows, dim)).astype(np.float32)
    tab = oracle.ShardedTable.from_full(full, world, entries)
    local, start = wm.get_local_tensor(host_view=False)
    if local.shape[0]:
        local.copy_(dev(torch.from_numpy(full[start:start + local.shape[0]])))
    if HIP_MODE:
        torch.cuda.synchronize()
    comm.barrier()
    rank_idx = []
    for r in range(world):
        g = np.random.default_rng(300 + r)
        n = 0 if (r == 1 and world > 2) else 4000 + 13 * r          # one rank may ask for nothing
        k = g.zipf(1.2, n).astype(np.uint64)
        ix = ((k * np.uint64(2654435761)) % np.uint64(n_rows)).astype(idt)
        if n:
            ix[::31] = -1
        rank_idx.append(ix)
    exp = oracle.distributed_gather(tab, rank_idx, np.float32, out_init=[np.full((len(ix), dim), 7, np.float32) for ix in rank_idx])
    from wholegraph_amd.torch.wholegraph_env import wrap_torch_tensor, get_wholegraph_env_fns, get_stream
    for _ in range(2):
        out = dev(torch.full((len(rank_idx[rank]), dim), 7.0))
        wi, wo = wrap_torch_tensor(dev(torch.from_numpy(rank_idx[rank]))), wrap_torch_tensor(out)
        wmb.check(wmb.lib().wholememory_gather(wm.wmb_tensor, wi.handle, wo.handle, get_wholegraph_env_fns(),
                                               C.c_void_p(get_stream()), -1))
        if HIP_MODE:
            torch.cuda.synchronize()
        assert host(out).numpy().tobytes() == exp[rank].tobytes(), "skewed gather mismatch on rank %d" % rank
    comm.barrier()
    wgth.destroy_wholememory_tensor(wm)


def scenario_gradient_apply(comm, rank, world, kind, params, idt, entries, mt="distributed", loc="cuda"):
    """Training steps on a table of any memory type (reference embedding_base::gather_gradient_apply, embedding.cpp:146-323,
    runs over every type): after each step the owner's shard is compared with the oracle, and EVERY rank reads the whole
    table back — for CHUNKED / CONTINUOUS through its own mappings of the peers' shards (right behind the optimizer's
    barrier: the owners must have drained their streams), for DISTRIBUTED through the collective gather."""
    n_rows, dim, steps = 1201, 13, 3
    emb = wgth.create_embedding(comm, mt, loc, torch.float32, [n_rows, dim],
                                embedding_entry_partition=entries)
    hv = loc == "cpu"
    stride = emb.get_embedding_tensor().stride()[0]
    assert stride == 16
    init = np.random.default_rng(5).standard_normal((n_rows, dim)).astype(np.float32)
    padded = np.zeros((n_rows, stride), dtype=np.float32)
    padded[:, :dim] = init
    tab = oracle.ShardedTable.from_full(padded, world, entries)
    tab.dim = dim
    local, start = emb.get_embedding_tensor().get_local_tensor(host_view=hv)
    cnt = int(tab.entry_offsets[rank + 1] - tab.entry_offsets[rank])
    assert tuple(local.shape) == (cnt, dim) and local.stride(0) == stride
    local.copy_(torch.from_numpy(init[start:start + cnt]) if hv else dev(torch.from_numpy(init[start:start + cnt])))
    if HIP_MODE:
        torch.cuda.synchronize()
    comm.barrier()
    all_ids = dev(torch.arange(n_rows, dtype=torch.int64))
    opt = wgth.create_wholememory_optimizer(emb, kind, params)
    assert emb.get_optimizer_state_names() == {"sgd": [], "adam": ["m", "v", "beta12t"], "adagrad": ["state_sum"],
                                               "rmsprop": ["v"]}[kind]
    ref_opts = [oracle.Optimizer(kind, int(tab.entry_offsets[r + 1] - tab.entry_offsets[r]), stride, **params)
                for r in range(world)]
    for step in range(steps):
        rank_idx, rank_grads = [], []
        for r in range(world):
            g = np.random.default_rng(1000 * step + r)
            ix = g.integers(0, n_rows, 400 + 11 * r).astype(idt)
            ix[::7] = ix[0]  # heavy duplicates
            gr = g.standard_normal((len(ix), dim)).astype(np.float32)
            if step == 1:    # "skip me" ids with junk gradient rows: dropped by the bucketing (one rank: inside the sort)
                ix[3::29] = -1
                gr[3::29] = 1e30
            rank_idx.append(ix)
            rank_grads.append(gr)
        emb.add_gradients(dev(torch.from_numpy(rank_idx[rank])), dev(torch.from_numpy(rank_grads[rank])))
        emb.need_apply = True
        g0 = wmb.lib().wholememory_ext_gradient_exchange_launches()
        if step == 2 and world > 2:
            os.environ["WM_EXCHANGE_PER_PEER"] = "1"   # the last step with the per-peer line-ups of rounds 1-5: the same bits
            _reload_knobs()
        opt.step(0.05)
        queued = wmb.lib().wholememory_ext_gradient_exchange_launches() - g0
        chunks = int(os.environ.get("WM_EXCHANGE_CHUNKS", "1"))
        if step == 2 and world > 2:
            del os.environ["WM_EXCHANGE_PER_PEER"]
            _reload_knobs()
            assert queued <= world * chunks + 1, (queued, world, chunks)   # (loopback: the own segment travels too)
            if chunks > 1 and os.environ.get("WM_EXCHANGE_SELF") != "1" and entries is None:   # (equal partition: no empty pair)
                assert queued in ((world - 1) * chunks, (world - 1) * chunks + 1), (queued, world, chunks)
        elif world > 1 and world <= 16:
            # one line-up kernel per chunk whatever the number of ranks (+ the chunk-major copy of the positions, + a copy of
            # the rank's own rows when they are not read in place); one chunk: the two ranges around the rank's own segment
            assert queued <= max(chunks + 2, 3), "gradient apply queued %d row kernels in front of %d chunks" % (queued, chunks)
            if world >= 3 and chunks > 1 and os.environ.get("WM_EXCHANGE_SELF") != "1" and entries is None:
                assert queued in (chunks + 1, chunks + 2), (queued, chunks)
        oracle.gradient_apply(tab, ref_opts, rank_idx, rank_grads, 0.05)
        if HIP_MODE:
            torch.cuda.synchronize()
        got = (local if hv else host(local)).numpy()
        if os.environ.get("WM_TEST_DIAG") and kind in ("rmsprop", "adagrad"):
            st, _ = emb.get_optimizer_state("v" if kind == "rmsprop" else "state_sum").get_local_tensor()
            st_got, st_want = host(st).numpy(), ref_opts[rank].per_element[:cnt, :dim]
            if st_got.tobytes() != st_want.tobytes():
                sb = np.nonzero((st_got != st_want).any(axis=1))[0]
                print("DIAG-STATE rank %d step %d %s %s: %d bad state rows (first %s -> global %s), got %s want %s" % (
                    rank, step, kind, mt, len(sb), sb[:8], sb[:8] + start, st_got[sb[0]][:3], st_want[sb[0]][:3]), flush=True)
        if got.tobytes() != tab.shards[rank][:cnt, :dim].tobytes() and os.environ.get("WM_TEST_DIAG"):
            want = tab.shards[rank][:cnt, :dim]
            bad = np.nonzero((got != want).any(axis=1))[0]
            touched = np.unique(np.concatenate([ix[ix >= 0] for ix in rank_idx]).astype(np.int64))
            print("DIAG rank %d step %d %s %s/%s: %d bad rows of %d (first %s), global ids %s; touched this step: %s; max |diff| %g" % (
                rank, step, kind, mt, loc, len(bad), cnt, bad[:8], (bad[:8] + start), np.isin(bad + start, touched)[:8],
                np.abs(got[bad].astype(np.float64) - want[bad]).max()), flush=True)
            for b in bad[:3]:
                print("   row %d got %s want %s" % (b + start, got[b][:4], want[b][:4]), flush=True)
        assert got.tobytes() == tab.shards[rank][:cnt, :dim].tobytes(), \
            "gradient apply (%s, %s/%s) mismatch on rank %d step %d" % (kind, mt, loc, rank, step)
        # the whole table as THIS rank sees it right after the step (opt.step ended with the communicator's barrier)
        seen = emb.gather(all_ids)
        if HIP_MODE:
            torch.cuda.synchronize()
        whole = np.concatenate([tab.shards[r][:int(tab.entry_offsets[r + 1] - tab.entry_offsets[r]), :dim] for r in range(world)])
        assert host(seen).numpy().tobytes() == whole.tobytes(), \
            "table read back after the step (%s, %s/%s) differs on rank %d step %d" % (kind, mt, loc, rank, step)
        comm.barrier()   # nobody starts the next step while a peer still reads
    if kind == "adam":
        m, _ = emb.get_optimizer_state("m").get_local_tensor()
        assert host(m).numpy().tobytes() == ref_opts[rank].per_element[:cnt, :dim].tobytes()
        b, _ = emb.get_optimizer_state("beta12t").get_local_tensor()
        assert host(b).numpy().tobytes() == ref_opts[rank].per_row[:cnt].tobytes()
    comm.barrier()
    wgth.destroy_wholememory_optimizer(opt)
    wgth.destroy_embedding(emb)


def scenario_sgd16(comm, rank, world, tdt, dim, lr, wd, mt="distributed"):
    """HALF / BF16 table trained with SGD over several ranks (HIP mode): gradient rows travel in the table dtype, the
    owner sums duplicates in fp32 in rank-major receive order and rounds once. Oracle = the fp32 multi-rank oracle
    wrapped in exact widenings and that one rounding."""
    n_rows, steps = 1501, 2
    os.environ["WM_GRAD_FOLD"] = "ordered"   # the bits of the receive-order sum (the 16-bit default is the tree fold, round 3)
    _reload_knobs()
    emb = wgth.create_embedding(comm, mt, "cuda", tdt, [n_rows, dim])
    stride = emb.get_embedding_tensor().stride()[0]
    init16 = torch.from_numpy(np.random.default_rng(8).standard_normal((n_rows, dim)).astype(np.float32)).to(tdt)
    padded = np.zeros((n_rows, stride), dtype=np.float32)
    padded[:, :dim] = init16.float().numpy()
    tab = oracle.ShardedTable.from_full(padded, world, None)
    tab.dim = dim
    local, start = emb.get_embedding_tensor().get_local_tensor()
    cnt = int(tab.entry_offsets[rank + 1] - tab.entry_offsets[rank])
    local.copy_(dev(init16[start:start + cnt]))
    opt = wgth.create_wholememory_optimizer(emb, "sgd", {"weight_decay": wd})
    ref_opts = [oracle.Optimizer("sgd", int(tab.entry_offsets[r + 1] - tab.entry_offsets[r]), stride, weight_decay=wd)
                for r in range(world)]
    for step in range(steps):
        rank_idx, rank_g16 = [], []
        for r in range(world):
            g = np.random.default_rng(500 * step + r)
            ix = g.integers(0, n_rows, 6000 + 13 * r).astype(np.int64)
            ix[::2] = ix[0]  # ~3000 duplicates per rank of one id: a long run at its owner
            rank_idx.append(ix)
            rank_g16.append(torch.from_numpy(g.standard_normal((len(ix), dim)).astype(np.float32)).to(tdt))
        emb.add_gradients(dev(torch.from_numpy(rank_idx[rank])), dev(rank_g16[rank]))
        emb.need_apply = True
        opt.step(lr)
        oracle.gradient_apply(tab, ref_opts, rank_idx, [g.float().numpy() for g in rank_g16], lr)
        for r in range(world):
            sh = tab.shards[r]
            sh[:, :dim] = torch.from_numpy(sh[:, :dim].copy()).to(tdt).float().numpy()   # the one rounding
        torch.cuda.synchronize()
        want = torch.from_numpy(tab.shards[rank][:cnt, :dim].copy()).to(tdt)
        assert torch.equal(host(local).view(torch.int16), want.view(torch.int16)), \
            "16-bit SGD (%s) mismatch on rank %d step %d" % (mt, rank, step)
        if mt != "distributed":   # every rank reads the whole table through its mappings right behind the step's barrier
            seen = emb.gather(dev(torch.arange(n_rows, dtype=torch.int64)))
            torch.cuda.synchronize()
            whole = torch.from_numpy(np.concatenate(
                [tab.shards[r][:int(tab.entry_offsets[r + 1] - tab.entry_offsets[r]), :dim] for r in range(world)])).to(tdt)
            assert torch.equal(host(seen).view(torch.int16), whole.view(torch.int16)), \
                "16-bit table read back (%s) differs on rank %d step %d" % (mt, rank, step)
            comm.barrier()
    comm.barrier()
    del os.environ["WM_GRAD_FOLD"]
    _reload_knobs()
    wgth.destroy_wholememory_optimizer(opt)
    wgth.destroy_embedding(emb)


def scenario_tree_fold(comm, rank, world, mt, kind, params):
    """WM_GRAD_FOLD=tree over several ranks (HIP mode): duplicates of a hot id arrive from every rank (a run of several
    segments at its owner, self rows read in place beside received ones); gradients are integer-valued, so the tree's sums are
    exact and table + states must equal the ordered multi-rank oracle bit for bit."""
    n_rows, dim, steps = 2003, 64, 2
    entries = None   # (equal partition)
    os.environ["WM_GRAD_FOLD"] = "tree"
    _reload_knobs()
    emb = wgth.create_embedding(comm, mt, "cuda", torch.float32, [n_rows, dim])
    stride = emb.get_embedding_tensor().stride()[0]
    init = np.random.default_rng(31).standard_normal((n_rows, dim)).astype(np.float32)
    padded = np.zeros((n_rows, stride), dtype=np.float32)
    padded[:, :dim] = init
    tab = oracle.ShardedTable.from_full(padded, world, None)
    tab.dim = dim
    local, start = emb.get_embedding_tensor().get_local_tensor()
    cnt = int(tab.entry_offsets[rank + 1] - tab.entry_offsets[rank])
    local.copy_(dev(torch.from_numpy(init[start:start + cnt])))
    torch.cuda.synchronize()
    comm.barrier()
    opt = wgth.create_wholememory_optimizer(emb, kind, params)
    ref_opts = [oracle.Optimizer(kind, int(tab.entry_offsets[r + 1] - tab.entry_offsets[r]), stride, **params) for r in range(world)]
    for step in range(steps):
        rank_idx, rank_grads = [], []
        for r in range(world):
            g = np.random.default_rng(900 * step + r)
            ix = g.integers(0, n_rows, 5000 + 17 * r).astype(np.int64)
            ix[::2] = 5 + step              # ~2500 duplicates per rank of one id
            ix[1::10] = n_rows - 3          # ~500 per rank of another (one segment at world 1, several at world 3)
            rank_idx.append(ix)
            rank_grads.append(g.integers(-2, 3, (len(ix), dim)).astype(np.float32))
        emb.add_gradients(dev(torch.from_numpy(rank_idx[rank])), dev(torch.from_numpy(rank_grads[rank])))
        emb.need_apply = True
        g0 = wmb.lib().wholememory_ext_gradient_exchange_launches()
        if step == 2 and world > 2:
            os.environ["WM_EXCHANGE_PER_PEER"] = "1"   # the last step with the per-peer line-ups of rounds 1-5: the same bits
            _reload_knobs()
        opt.step(0.05)
        queued = wmb.lib().wholememory_ext_gradient_exchange_launches() - g0
        chunks = int(os.environ.get("WM_EXCHANGE_CHUNKS", "1"))
        if step == 2 and world > 2:
            del os.environ["WM_EXCHANGE_PER_PEER"]
            _reload_knobs()
            assert queued <= world * chunks + 1, (queued, world, chunks)   # (loopback: the own segment travels too)
            if chunks > 1 and os.environ.get("WM_EXCHANGE_SELF") != "1" and entries is None:   # (equal partition: no empty pair)
                assert queued in ((world - 1) * chunks, (world - 1) * chunks + 1), (queued, world, chunks)
        elif world > 1 and world <= 16:
            # one line-up kernel per chunk whatever the number of ranks (+ the chunk-major copy of the positions, + a copy of
            # the rank's own rows when they are not read in place); one chunk: the two ranges around the rank's own segment
            assert queued <= max(chunks + 2, 3), "gradient apply queued %d row kernels in front of %d chunks" % (queued, chunks)
            if world >= 3 and chunks > 1 and os.environ.get("WM_EXCHANGE_SELF") != "1" and entries is None:
                assert queued in (chunks + 1, chunks + 2), (queued, chunks)
        oracle.gradient_apply(tab, ref_opts, rank_idx, rank_grads, 0.05)
        torch.cuda.synchronize()
        assert host(local).numpy().tobytes() == tab.shards[rank][:cnt, :dim].tobytes(), \
            "tree fold (%s, %s) mismatch on rank %d step %d" % (kind, mt, rank, step)
    comm.barrier()
    del os.environ["WM_GRAD_FOLD"]
    _reload_knobs()
    wgth.destroy_wholememory_optimizer(opt)
    wgth.destroy_embedding(emb)


def scenario_combined_gradients(comm, rank, world, mt, kind, params, tdt=torch.float32, idt=np.int64, overflow=False):
    """Sender-side combination of duplicate gradient rows (embedding.cpp: combined_gradient_apply; not in the reference, which
    ships every copy: embedding.cpp:193-247). Tree fold + WM_GRAD_COMBINE=1: every rank folds ITS duplicates of an id into one
    partial row before the exchange, the owner folds at most `world` partial rows per id. Gradients are integer-valued with
    partial sums that the exchange dtype represents exactly, so table (and states) must equal the ORDERED multi-rank oracle bit
    for bit; the bytes handed to the all-to-all-v must shrink (a hot id makes up half of every rank's batch); with
    WM_GRAD_COMBINE=0 the same step ships every copy and gives the same bits. overflow (float16): one rank's partial sum
    leaves the float16 range -> its veto rides in the counts exchange and EVERY rank takes the uncombined route for that step."""
    n_rows, dim, steps = 3001, 32, 3
    f32 = tdt == torch.float32
    os.environ["WM_GRAD_FOLD"] = "tree"
    _reload_knobs()
    emb = wgth.create_embedding(comm, mt, "cuda", tdt, [n_rows, dim])
    stride = emb.get_embedding_tensor().stride()[0]
    init_t = torch.from_numpy(np.random.default_rng(61).integers(-8, 9, (n_rows, dim)).astype(np.float32)).to(tdt)
    padded = np.zeros((n_rows, stride), dtype=np.float32)
    padded[:, :dim] = init_t.float().numpy()
    tab = oracle.ShardedTable.from_full(padded, world, None)
    tab.dim = dim
    local, start = emb.get_embedding_tensor().get_local_tensor()
    cnt = int(tab.entry_offsets[rank + 1] - tab.entry_offsets[rank])
    local.copy_(dev(init_t[start:start + cnt]))
    if HIP_MODE:
        torch.cuda.synchronize()
    comm.barrier()
    opt = wgth.create_wholememory_optimizer(emb, kind, params)
    ref_opts = [oracle.Optimizer(kind, int(tab.entry_offsets[r + 1] - tab.entry_offsets[r]), stride, **params) for r in range(world)]
    exchanging = world > 1 or os.environ.get("WM_EXCHANGE_SELF") == "1"
    lr = 0.5 if f32 else 2.0 ** -6
    sent = {}
    for step in range(steps):
        combine = step != 1                      # step 1 ships every copy: the same bits, more bytes
        os.environ["WM_GRAD_COMBINE"] = "1" if combine else "0"
        _reload_knobs()
        rank_idx, rank_grads = [], []
        for r in range(world):
            g = np.random.default_rng(4000 + 10 * (step % 2) + r)   # (steps 0 and 1 use different batches, 2 repeats 0's)
            ix = g.integers(0, n_rows, 4000 + 9 * r).astype(idt)
            ix[::2] = 17                         # half of every rank's batch is one hot id: one run per sender, `world` rows at its owner
            ix[1::8] = n_rows - 2 - (r % 2)      # ~500 per rank of two warm ids
            ix[5::97] = -1                       # "skip me"
            gr = g.choice(np.array([-1.0, 0.0, 1.0], np.float32), size=(len(ix), dim), p=[0.02, 0.96, 0.02]).astype(np.float32)
            if overflow and step == 2 and r == world - 1:
                gr[ix == 17] = 0.0
                gr[np.nonzero(ix == 17)[0][:3]] = 30000.0      # three copies: each fits float16, their sum does not
            rank_idx.append(ix)
            rank_grads.append(gr)
        if not f32 and not (overflow and step == 2):   # the test's premise: every sender's partial sums are exact in the table dtype
            for r in range(world):
                for hot in (17, n_rows - 2, n_rows - 3):
                    ps = rank_grads[r][rank_idx[r] == hot].sum(axis=0)
                    assert np.abs(ps).max() <= 128, "partial sums too large for an exact 16-bit test: %g" % np.abs(ps).max()
        b0, c0 = wmb.lib().wholememory_ext_alltoallv_bytes(), wmb.lib().wholememory_ext_combined_gradient_calls()
        emb.add_gradients(dev(torch.from_numpy(rank_idx[rank])), dev(torch.from_numpy(rank_grads[rank]).to(tdt)))
        emb.need_apply = True
        opt.step(lr)
        if HIP_MODE:
            torch.cuda.synchronize()
        sent[step] = wmb.lib().wholememory_ext_alltoallv_bytes() - b0
        took = wmb.lib().wholememory_ext_combined_gradient_calls() - c0
        vetoed = overflow and step == 2
        assert took == (1 if combine and exchanging and not vetoed else 0), \
            "step %d: combined route taken %d times (combine %s, exchanging %s, veto %s)" % (step, took, combine, exchanging, vetoed)
        oracle.gradient_apply(tab, ref_opts, rank_idx, rank_grads, lr)
        if not f32:
            for r in range(world):
                sh = tab.shards[r]
                sh[:, :dim] = torch.from_numpy(sh[:, :dim].copy()).to(tdt).float().numpy()   # the table's one rounding
        want = torch.from_numpy(tab.shards[rank][:cnt, :dim].copy()).to(tdt)
        got = host(local)
        assert got.numpy().tobytes() == want.numpy().tobytes() if f32 else torch.equal(got.view(torch.int16), want.view(torch.int16)), \
            "combined gradient apply (%s, %s, %s) mismatch on rank %d step %d" % (kind, mt, tdt, rank, step)
        comm.barrier()
    if exchanging and not overflow:
        # steps 0 and 2 are the same batch with combination, step 1 a batch of the same shape without: half of every batch is one
        # id, so the combined exchange moves well under two thirds of the bytes
        assert sent[0] == sent[2] and sent[0] * 3 < sent[1] * 2, sent
    if kind == "adam":
        m, _ = emb.get_optimizer_state("m").get_local_tensor()
        assert host(m).numpy().tobytes() == ref_opts[rank].per_element[:cnt, :dim].tobytes()
    comm.barrier()
    del os.environ["WM_GRAD_FOLD"]
    del os.environ["WM_GRAD_COMBINE"]
    _reload_knobs()
    wgth.destroy_wholememory_optimizer(opt)
    wgth.destroy_embedding(emb)


def scenario_cached_embedding(comm, rank, world, mt):
    """HOST embedding with a read-write device cache on every rank (HIP mode): owners serve lookups cache-first through
    the exchange, train through the cache, write back. Bit-exact vs the uncached multi-rank oracle."""
    n_rows, dim, steps = 9001, 24, 3
    policy = wgth.create_wholememory_cache_policy(comm, memory_type=mt, memory_location="cuda", access_type="readwrite",
                                                  ratio=0.15)
    emb = wgth.create_embedding(comm, mt, "cpu", torch.float32, [n_rows, dim], cache_policy=policy)
    stride = emb.get_embedding_tensor().stride()[0]
    init = np.random.default_rng(21).standard_normal((n_rows, dim)).astype(np.float32)
    padded = np.zeros((n_rows, stride), dtype=np.float32)
    padded[:, :dim] = init
    tab = oracle.ShardedTable.from_full(padded, world, None)
    tab.dim = dim
    local, start = emb.get_embedding_tensor().get_local_tensor(host_view=True)
    cnt = int(tab.entry_offsets[rank + 1] - tab.entry_offsets[rank])
    local.copy_(torch.from_numpy(init[start:start + cnt]))
    comm.barrier()
    opt = wgth.create_wholememory_optimizer(emb, "adam", {})
    ref_opts = [oracle.Optimizer("adam", int(tab.entry_offsets[r + 1] - tab.entry_offsets[r]), stride) for r in range(world)]
    for step in range(steps):
        rank_idx, rank_grads = [], []
        for r in range(world):
            g = np.random.default_rng(77 * step + r)
            k = g.zipf(1.3, 3000 + 50 * r).astype(np.uint64)
            ix = ((k * np.uint64(2654435761)) % np.uint64(n_rows)).astype(np.int64)
            rank_idx.append(ix)
            rank_grads.append(g.standard_normal((len(ix), dim)).astype(np.float32))
        exp = oracle.distributed_gather(tab, rank_idx, np.float32)
        got = emb.gather(dev(torch.from_numpy(rank_idx[rank])))
        torch.cuda.synchronize()
        assert host(got).numpy().tobytes() == exp[rank].tobytes(), "cached gather mismatch rank %d step %d" % (rank, step)
        emb.add_gradients(dev(torch.from_numpy(rank_idx[rank])), dev(torch.from_numpy(rank_grads[rank])))
        emb.need_apply = True
        opt.step(0.03)
        oracle.gradient_apply(tab, ref_opts, rank_idx, rank_grads, 0.03)
    emb.writeback_all_cache()
    comm.barrier()
    assert local.numpy().tobytes() == tab.shards[rank][:cnt, :dim].tobytes(), "table after write-back, rank %d" % rank
    m_local, _ = emb.get_optimizer_state("m").get_local_tensor(host_view=True)
    assert m_local.numpy().tobytes() == ref_opts[rank].per_element[:cnt, :dim].tobytes(), "state m after write-back"
    comm.barrier()
    wgth.destroy_wholememory_optimizer(opt)
    wgth.destroy_embedding(emb)


def scenario_local_cache(comm, rank, world, mt):
    """Every rank keeps its own read-only cache (cache communicator of size 1) of a table spread over all ranks. For a
    DISTRIBUTED table the cache fills and the misses travel through the collective exchange."""
    n_rows, dim = 7001, 40
    local_comm = wgth.create_group_communicator(1)
    policy = wgth.create_wholememory_cache_policy(local_comm, memory_type="continuous", memory_location="cuda",
                                                  access_type="readonly", ratio=0.2)
    emb = wgth.create_embedding(comm, mt, "cuda", torch.float32, [n_rows, dim], cache_policy=policy)
    stride = emb.get_embedding_tensor().stride()[0]
    full = oracle.fill_closed_form(np.float32, 0, n_rows, dim, stride)
    tab = oracle.ShardedTable.from_full(full, world, None)
    tab.dim = dim
    local, start = emb.get_embedding_tensor().get_local_tensor()
    cnt = int(tab.entry_offsets[rank + 1] - tab.entry_offsets[rank])
    local.copy_(dev(torch.from_numpy(full[start:start + cnt, :dim])))
    torch.cuda.synchronize()
    comm.barrier()
    for step in range(4):
        rank_idx = []
        for r in range(world):
            g = np.random.default_rng(31 * step + r)
            k = g.zipf(1.3, 2500 + 100 * r).astype(np.uint64)
            ix = ((k * np.uint64(2654435761)) % np.uint64(n_rows)).astype(np.int64)
            ix[::41] = -1
            rank_idx.append(ix)
        exp = oracle.distributed_gather(tab, rank_idx, np.float32, out_init=[np.full((len(ix), dim), 3, np.float32) for ix in rank_idx])
        out = dev(torch.full((len(rank_idx[rank]), dim), 3.0))
        got = emb.gather(dev(torch.from_numpy(rank_idx[rank])), out=out)
        torch.cuda.synchronize()
        assert host(got).numpy().tobytes() == exp[rank].tobytes(), "local cache (%s) gather mismatch rank %d step %d" % (mt, rank, step)
    v = [C.c_int64() for _ in range(5)]
    wmb.check(wmb.lib().wholememory_ext_embedding_cache_info(emb.wmb_embedding, *[C.byref(x) for x in v]))
    assert v[1].value > 0 and v[3].value > 0, "cache should hold rows and serve hits: %s" % [x.value for x in v]
    comm.barrier()
    wgth.destroy_embedding(emb)


def scenario_file_io(comm, rank, world, tmpdir):
    """wholememory_load_from_file / store_to_file through the Python surface: files re-sharded over ranks (3 files of
    uneven size -> W shards), padded rows (file rows are dim wide, memory rows stride wide), round-robin placement,
    and a save -> load round trip of "%s_part_%d_of_%d" shards (torch/embedding.py:358-377, file_io.cpp:1860,2059)."""
    n_rows, dim = 1003, 7
    full = (np.arange(n_rows * dim, dtype=np.float32).reshape(n_rows, dim) * 0.5).astype(np.float32)
    files = [os.path.join(tmpdir, "feat_%d.bin" % i) for i in range(3)]
    cuts = [0, 400, 401, n_rows]
    if rank == 0:
        os.makedirs(tmpdir, exist_ok=True)
        for i, f in enumerate(files):
            full[cuts[i]:cuts[i + 1]].tofile(f)
    comm.barrier()
    wm = wgth.create_wholememory_tensor_from_filelist(comm, "distributed", "cuda", files, torch.float32, last_dim_size=dim,
                                                      last_dim_strides=8)
    local, start = wm.get_local_tensor()
    assert local.stride(0) == 8
    exp = full[start:start + local.shape[0]]
    assert host(local).numpy().tobytes() == np.ascontiguousarray(exp).tobytes(), "plain load mismatch on rank %d" % rank
    # store every rank's shard, reload into a fresh tensor through the part files
    prefix = os.path.join(tmpdir, "ckpt")
    wm.to_file_prefix(prefix)
    comm.barrier()
    wm2 = wgth.create_wholememory_tensor(comm, "distributed", "cuda", [n_rows, dim], torch.float32, [8, 1])
    wm2.from_file_prefix(prefix)
    l2, s2 = wm2.get_local_tensor()
    assert s2 == start and host(l2).numpy().tobytes() == np.ascontiguousarray(exp).tobytes()
    wgth.destroy_wholememory_tensor(wm2)
    wgth.destroy_wholememory_tensor(wm)
    # round-robin placement: local row l of rank r <- file entry ((l // rr) * W + r) * rr + l % rr
    rr = 4
    emb = wgth.create_embedding_from_filelist(comm, "distributed", "cuda", files, torch.float32, dim, round_robin_size=rr)
    assert emb.shape[0] == oracle.round_robin_total_entries(n_rows, world, rr)
    lt, st = emb.get_embedding_tensor().get_local_tensor()
    got = host(lt).numpy()
    for l in range(lt.shape[0]):
        e = ((l // rr) * world + rank) * rr + l % rr
        if e < n_rows:
            assert np.array_equal(got[l], full[e]), "round-robin row %d on rank %d" % (l, rank)
    comm.barrier()
    # ... and a gather by STORAGE id returns the file's rows whoever owns them (the reference's remap only finds the rows
    # the caller owns itself, map_indices_func.cu:34-43; the product maps to the owner's shard)
    ids = np.random.default_rng(70 + rank).integers(0, n_rows, 300 + 11 * rank).astype(np.int64)
    ids[::17] = -1
    out = dev(torch.full((len(ids), dim), -3.0))
    got_rows = emb.gather(dev(torch.from_numpy(ids)), out=out)
    if HIP_MODE:
        torch.cuda.synchronize()
    exp_rows = np.full((len(ids), dim), -3.0, np.float32)
    exp_rows[ids >= 0] = full[ids[ids >= 0].astype(np.int64)]
    assert np.array_equal(host(got_rows).numpy(), exp_rows), "round-robin gather by storage id, rank %d" % rank
    rank_rows = emb.shape[0] // world
    assert np.array_equal(oracle.round_robin_map(ids[ids >= 0], 0, world, rr, rank_rows=rank_rows) % rank_rows,
                          rr * ((ids[ids >= 0] // rr) // world) + ids[ids >= 0] % rr)
    comm.barrier()
    wgth.destroy_embedding(emb)


def scenario_sampling(comm, rank, world, mt, col_dt, loc="cuda"):
    """Neighbour sampling on a CSR spread over the ranks (HIP mode): DISTRIBUTED goes through two collective gathers
    (row bounds, sampled columns), CHUNKED / CONTINUOUS read peers' shards directly. Bit-exact vs the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_graph_oracle import make_csr
    row_ptr, col = make_csr(1500, 60, 4242, col_dt, heavy=[(5, 1400), (6, 0), (1499, 300)])
    tensors = []
    for arr in (row_ptr, col):
        t = wgth.create_wholememory_tensor(comm, mt, loc, [arr.shape[0]], torch.from_numpy(arr).dtype, [1])
        local, start = t.get_local_tensor(host_view=(loc == "cpu"))
        local.copy_(torch.from_numpy(arr[start:start + local.shape[0]]))
        tensors.append(t)
    if HIP_MODE:
        torch.cuda.synchronize()
    comm.barrier()
    g = wgth.GraphStructure()
    g.set_csr_graph(tensors[0], tensors[1])
    rng = np.random.default_rng(900 + rank)
    n_center = 0 if rank == world - 1 else 400 + 17 * rank          # the last rank asks for nothing
    centers = np.concatenate([[5, 6, 1499], rng.integers(0, 1500, n_center)])[:n_center + (3 if n_center else 0)]
    centers = centers.astype(np.int64 if rank % 2 == 0 else np.int32)
    for m in (5, 40, -1, 1100):
        seed = 31337 * (m + 2) + rank
        off, ids, lid, egid = g.unweighted_sample_without_replacement_one_hop(
            dev(torch.from_numpy(centers)), m, random_seed=seed, need_center_local_output=True, need_edge_output=True)
        o_off, o_ids, o_lid, o_egid = oracle.sample_unweighted(row_ptr, col, centers, m, seed)
        assert np.array_equal(host(off).numpy(), o_off), (mt, m)
        assert np.array_equal(host(egid).numpy(), o_egid), (mt, m)
        assert np.array_equal(host(ids).numpy(), o_ids) and np.array_equal(host(lid).numpy(), o_lid), (mt, m)
        off2, ids2 = g.unweighted_sample_without_replacement_one_hop(dev(torch.from_numpy(centers)), m, random_seed=seed)
        assert torch.equal(ids2, ids) and torch.equal(off2, off)
    comm.barrier()
    for t in tensors:
        wgth.destroy_wholememory_tensor(t)


def hierarchy_scenarios(comm, rank, world):
    """HIERARCHY tables: rows owned as in DISTRIBUTED, gathered in two hops (inside the node to the relay with the
    owner's local rank, then between nodes along that rail). The node layout is pretended through WM_LOCAL_SIZE."""
    L = int(os.environ["WM_LOCAL_SIZE"])
    local_size = C.c_int(0)
    wmb.check(wmb.lib().wholememory_communicator_get_local_size(C.byref(local_size), comm.wmb_comm))
    assert local_size.value == L
    S = wmb.lib().wholememory_communicator_support_type_location
    assert S(comm.wmb_comm, wmb.MT_HIERARCHY, wmb.ML_DEVICE) == 0 and S(comm.wmb_comm, wmb.MT_DISTRIBUTED, wmb.ML_HOST) == 0
    # peer mappings stop at the node boundary
    assert (S(comm.wmb_comm, wmb.MT_CHUNKED, wmb.ML_DEVICE) == 0) == (L == world)
    # the handle carries the two sub-communicators: my node, my rail
    t = wgth.create_wholememory_tensor(comm, "hierarchy", "cuda", [64, 4], torch.float32, [4, 1])
    h = wmb.lib().wholememory_tensor_get_memory_handle(t.wmb_tensor)
    lc, cc, r, n = C.c_void_p(), C.c_void_p(), C.c_int(), C.c_int()
    wmb.check(wmb.lib().wholememory_get_local_communicator(C.byref(lc), C.c_void_p(h)))
    wmb.check(wmb.lib().wholememory_get_cross_communicator(C.byref(cc), C.c_void_p(h)))
    for sub, want_rank, want_size in ((lc, rank % L, L), (cc, rank // L, world // L)):
        wmb.check(wmb.lib().wholememory_communicator_get_rank(C.byref(r), sub))
        wmb.check(wmb.lib().wholememory_communicator_get_size(C.byref(n), sub))
        assert (r.value, n.value) == (want_rank, want_size)
    wgth.destroy_wholememory_tensor(t)
    w8 = np.random.default_rng(42).uniform(90, 100, world)
    w8[world // 2] *= 0.3
    ent = [int(x) for x in (w8 / w8.sum() * 997).astype(int)]
    ent[0] += 997 - sum(ent)
    scenario_gather_scatter(comm, rank, world, "hierarchy", 1003, 11, np.float32, np.float32, np.int64, None)
    scenario_gather_scatter(comm, rank, world, "hierarchy", 2000, 32, np.float16, np.float32, np.int32, None)
    scenario_gather_scatter(comm, rank, world, "hierarchy", 997, 8, np.int64, np.int32, np.int64, ent)
    scenario_gather_scatter(comm, rank, world, "hierarchy", 5, 4, np.float32, np.float32, np.int64, None)  # empty ranks
    scenario_gather_scatter(comm, rank, world, "hierarchy", 2003, 32, np.float32, np.float32, np.int64, None, loc="cpu")
    scenario_gradient_apply(comm, rank, world, "adam", {"weight_decay": 0.01}, np.int64, None, mt="hierarchy")
    # plain DISTRIBUTED is unaffected by the node layout
    scenario_gather_scatter(comm, rank, world, "distributed", 1003, 11, np.float32, np.float32, np.int64, None)


def fuzz_scenarios(comm, rank, world):
    """Random table shapes, dtype pairs, id dtypes and row partitions (every rank draws the same ones from FUZZ_SEED) through
    the gather / scatter scenario; HIERARCHY tables too when a node layout is pretended (WM_LOCAL_SIZE)."""
    rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
    types = ["distributed"] + (["hierarchy"] if "WM_LOCAL_SIZE" in os.environ else [])
    pairs = [(np.float32, np.float32), (np.float16, np.float32), (np.float32, np.float16), (np.int64, np.int32),
             (np.int32, np.int64), (np.float16, np.float16)]
    for case in range(int(os.environ.get("FUZZ_CASES", "10"))):
        n_rows = int(rng.integers(1, 5000))
        dim = int(rng.choice([1, 3, 4, 8, 11, 32, 33, 64, 100]))
        tdt, odt = pairs[rng.integers(len(pairs))]
        idt = np.int32 if rng.random() < 0.5 else np.int64
        mt = types[rng.integers(len(types))]
        entries = None
        if rng.random() < 0.5 and n_rows >= world:
            cuts = np.sort(rng.choice(np.arange(1, n_rows), world - 1, replace=False)) if world > 1 else np.array([], int)
            entries = [int(x) for x in np.diff(np.concatenate([[0], cuts, [n_rows]]))]
        try:
            scenario_gather_scatter(comm, rank, world, mt, n_rows, dim, tdt, odt, idt, entries)
            if case % 3 == 0:   # a training flow too: random optimizer on a randomly partitioned 1201-row table
                kind, params = [("sgd", {"weight_decay": 0.1}), ("adam", {"weight_decay": 0.01}), ("adagrad", {}),
                                ("rmsprop", {"alpha": 0.95})][rng.integers(4)]
                cuts = np.sort(rng.choice(np.arange(1, 1201), world - 1, replace=False))
                gent = [int(x) for x in np.diff(np.concatenate([[0], cuts, [1201]]))] if rng.random() < 0.5 else None
                scenario_gradient_apply(comm, rank, world, kind, params, np.int64 if rng.random() < 0.5 else np.int32, gent, mt=mt)
        except BaseException:
            print("FUZZ CASE %d: %s rows %d dim %d %s->%s %s entries %s" % (
                case, mt, n_rows, dim, np.dtype(tdt).name, np.dtype(odt).name, np.dtype(idt).name, entries), flush=True)
            raise


def rccl_scenarios(comm, rank, world):
    """The distributed ops with the library's RCCL transport underneath (reference nccl_comms.cpp:82-86 barrier,
    :383-437 host_alltoall / alltoallv, communicator.cpp:703-752 create): transport identity, barrier, split, then
    DISTRIBUTED gather / scatter / gradient apply bit-exact against the oracle's multi-rank simulation."""
    assert comm.transport() == ("rccl", world), comm.transport()
    comm.barrier()
    # split: colour = rank parity -> communicators of ceil/floor(world / 2) ranks, each with its own ncclComm
    sub = wgth.split_communicator(comm, rank % 2, rank)
    assert sub.get_size() == (world + 1 - rank % 2) // 2 and sub.get_rank() == rank // 2
    if sub.get_size() > 1 or os.environ.get("WM_EXCHANGE_SELF") == "1":
        assert sub.transport() == ("rccl", sub.get_size()), sub.transport()
    sub.barrier()
    if os.environ.get("WM_EXCHANGE_SELF") == "1" or sub.get_size() > 1:
        scenario_gather_scatter(sub, sub.get_rank(), sub.get_size(), "distributed", 1003, 11, np.float32, np.float32,
                                np.int64, None)
    wgth.destroy_communicator(sub)
    comm.barrier()
    scenario_gather_scatter(comm, rank, world, "distributed", 1003, 11, np.float32, np.float32, np.int64, None)
    scenario_gather_scatter(comm, rank, world, "distributed", 2000, 32, np.float16, np.float32, np.int32, None)
    scenario_gather_scatter(comm, rank, world, "distributed", 300001, 128, np.float32, np.float32, np.int64, None)
    w8 = np.random.default_rng(42).uniform(90, 100, world)
    ent = [int(x) for x in (w8 / w8.sum() * 997).astype(int)]
    ent[0] += 997 - sum(ent)
    scenario_gather_scatter(comm, rank, world, "distributed", 997, 8, np.int64, np.int32, np.int64, ent)
    os.environ["WM_GATHER_DEDUP"] = "2"
    _reload_knobs()
    scenario_gather_scatter(comm, rank, world, "distributed", 1003, 11, np.float32, np.float32, np.int64, None)
    del os.environ["WM_GATHER_DEDUP"]
    _reload_knobs()
    scenario_gather_skewed(comm, rank, world, np.int64)     # automatic decision (duplicate estimate through the exchange)
    scenario_gather_skewed(comm, rank, world, np.int32)
    scenario_gather_scatter(comm, rank, world, "distributed", 2003, 32, np.float32, np.float32, np.int64, None, loc="cpu")
    for kind, params in [("sgd", {"weight_decay": 0.1}), ("adam", {"weight_decay": 0.01}), ("adagrad", {}),
                         ("rmsprop", {"alpha": 0.95})]:
        scenario_gradient_apply(comm, rank, world, kind, params, np.int64 if kind != "adagrad" else np.int32, None)
    scenario_sgd16(comm, rank, world, torch.float16, 256, -1.0, 0.0)
    # C4 as BASELINE names it: training on mapped tables (several GPUs: hipIpc / HIP-VMM mappings between the ranks)
    for mt5 in ("continuous", "chunked"):
        for kind, params in [("sgd", {"weight_decay": 0.1}), ("adam", {"weight_decay": 0.01}), ("adagrad", {}),
                             ("rmsprop", {"alpha": 0.95})]:
            scenario_gradient_apply(comm, rank, world, kind, params, np.int64, None, mt=mt5)
        scenario_sgd16(comm, rank, world, torch.float16, 256, -1.0, 0.0, mt=mt5)
    scenario_tree_fold(comm, rank, world, "distributed", "rmsprop", {"alpha": 0.95})
    scenario_combined_gradients(comm, rank, world, "distributed", "sgd", {"weight_decay": 0.0})
    scenario_combined_gradients(comm, rank, world, "distributed", "sgd", {"weight_decay": 0.0}, tdt=torch.float16)
    scenario_combined_gradients(comm, rank, world, "distributed", "sgd", {"weight_decay": 0.0}, tdt=torch.float16, overflow=True)
    scenario_sampling(comm, rank, world, "distributed", np.int64)
    scenario_cached_embedding(comm, rank, world, "distributed")
    scenario_file_io(comm, rank, world, "/tmp/wgamd_test_rccl_%s" % os.environ["MASTER_PORT"])


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    if RCCL_MODE:
        assert torch.cuda.device_count() >= world, "hip-rccl needs one GPU per rank"
        torch.cuda.set_device(rank)
    elif HIP_MODE:
        torch.cuda.set_device(0)   # every rank shares the one GPU of the test box
    dist.init_process_group(backend="nccl" if RCCL_MODE else "gloo", init_method="env://", rank=rank, world_size=world)
    if HIP_MODE:
        assert wmb.lib().wholememory_ext_backend_name() == b"hip-gfx950"
    else:
        install_test_backend()
    wgth.init(rank, world, rank, world, os.environ.get("WM_TEST_LOG", "warn"))
    comm = wgth.get_global_communicator()
    assert comm.get_rank() == rank and comm.get_size() == world
    if RCCL_MODE:
        rccl_scenarios(comm, rank, world)
        comm.barrier()
        dist.barrier()
        print("RANK %d OK" % rank)
        wgth.finalize()
        return
    if FUZZ_MODE:
        fuzz_scenarios(comm, rank, world)
        comm.barrier()
        dist.barrier()
        print("RANK %d OK" % rank)
        wgth.finalize()
        return
    if HIER_MODE:
        hierarchy_scenarios(comm, rank, world)
        comm.barrier()
        dist.barrier()
        print("RANK %d OK" % rank)
        wgth.finalize()
        return
    if os.environ.get("WM_TEST_ONLY") == "mapped_training":   # debugging aid: only the mapped-table training block, repeated
        gw = np.random.default_rng(7).uniform(60, 100, world)
        gent = [int(x) for x in (gw / gw.sum() * 1201).astype(int)]
        gent[-1] += 1201 - sum(gent)
        for rep in range(int(os.environ.get("WM_TEST_REPS", "3"))):
            for mt5 in ("continuous", "chunked"):
                for kind, params in [("rmsprop", {"alpha": 0.95}), ("sgd", {"weight_decay": 0.1}), ("adam", {"weight_decay": 0.01}), ("adagrad", {})]:
                    for ent5, idt5 in ((gent, np.int64), (None, np.int32), (gent, np.int32), (None, np.int64)):
                        try:
                            scenario_gradient_apply(comm, rank, world, kind, params, idt5, ent5, mt=mt5)
                            print("ok   rep %d %s %s %s %s" % (rep, mt5, kind, "custom" if ent5 else "equal", np.dtype(idt5).name), flush=True)
                        except AssertionError as ex:
                            print("FAIL rep %d %s %s %s %s: %s" % (rep, mt5, kind, "custom" if ent5 else "equal", np.dtype(idt5).name, ex), flush=True)
                            os._exit(3)
        comm.barrier()
        dist.barrier()
        print("RANK %d OK" % rank)
        wgth.finalize()
        return
    # (1) gather / scatter, equal plan, padded rows, fp32, int64 ids
    scenario_gather_scatter(comm, rank, world, "distributed", 1003, 11, np.float32, np.float32, np.int64, None)
    # (2) dtype cast at the owner + int32 ids + an asking-nothing rank
    scenario_gather_scatter(comm, rank, world, "distributed", 2000, 32, np.float16, np.float32, np.int32, None)
    # (3) custom partition (python random_partition shape) + ints
    w8 = np.random.default_rng(42).uniform(90, 100, world)   # reference python random_partition shape
    w8[world // 2] *= 0.3                                       # plus one clearly smaller shard
    ent = [int(x) for x in (w8 / w8.sum() * 997).astype(int)]
    ent[0] += 997 - sum(ent)
    ent2 = [2 * e for e in ent]
    ent2[0] += 2003 - sum(ent2)
    ent3 = [3 * e for e in ent]
    ent3[0] += 3001 - sum(ent3)
    scenario_gather_scatter(comm, rank, world, "distributed", 997, 8, np.int64, np.int32, np.int64, ent)
    # (4) trailing ranks empty under the equal plan: N < W * ceil(N / W) only when W > N … use tiny N
    if world >= 3:
        scenario_gather_scatter(comm, rank, world, "distributed", 4, 4, np.float32, np.float32, np.int64, None)
    # (4') skewed batches: the automatic de-duplication decision (and the same batch with the decision forced off)
    scenario_gather_skewed(comm, rank, world, np.int64)
    scenario_gather_skewed(comm, rank, world, np.int32, ent3 if world > 1 else None)
    os.environ["WM_GATHER_DEDUP"] = "0"
    _reload_knobs()
    scenario_gather_skewed(comm, rank, world, np.int64)
    del os.environ["WM_GATHER_DEDUP"]
    _reload_knobs()
    # (4a) the same gathers with request de-duplication before the exchange (WM_GATHER_DEDUP=1)
    os.environ["WM_GATHER_DEDUP"] = "1"
    _reload_knobs()
    scenario_gather_scatter(comm, rank, world, "distributed", 1003, 11, np.float32, np.float32, np.int64, None)
    scenario_gather_scatter(comm, rank, world, "distributed", 2000, 32, np.float16, np.float32, np.int32, None)
    scenario_gather_scatter(comm, rank, world, "distributed", 997, 8, np.int64, np.int32, np.int64, ent)
    del os.environ["WM_GATHER_DEDUP"]
    _reload_knobs()
    if HIP_MODE:
        # (4b) CHUNKED over two processes: peers' shards mapped with hipIpc, kernels read them directly;
        #      host-located tables: one POSIX shm segment registered with HIP on every rank
        scenario_gather_scatter(comm, rank, world, "chunked", 3001, 128, np.float32, np.float32, np.int64, None)
        scenario_gather_scatter(comm, rank, world, "chunked", 997, 8, np.int64, np.int32, np.int64, ent)
        scenario_gather_scatter(comm, rank, world, "continuous", 3001, 128, np.float32, np.float32, np.int64, None)
        scenario_gather_scatter(comm, rank, world, "continuous", 997, 8, np.int64, np.int32, np.int64, ent)
        scenario_gather_scatter(comm, rank, world, "continuous", 900001, 32, np.float32, np.float32, np.int32, None)
        scenario_gather_scatter(comm, rank, world, "chunked", 2003, 32, np.float32, np.float32, np.int64, None, loc="cpu")
        scenario_gather_scatter(comm, rank, world, "continuous", 2003, 32, np.float32, np.float16, np.int32, ent2, loc="cpu")
        scenario_gather_scatter(comm, rank, world, "distributed", 2003, 32, np.float32, np.float32, np.int64, None, loc="cpu")
        # (4b') the same CHUNKED / CONTINUOUS tables served through the explicit all-to-all-v route
        os.environ["WM_MAPPED_VIA_EXCHANGE"] = "1"
        _reload_knobs()
        scenario_gather_scatter(comm, rank, world, "chunked", 3001, 128, np.float32, np.float32, np.int64, None)
        scenario_gather_scatter(comm, rank, world, "continuous", 997, 8, np.int64, np.int32, np.int64, ent)
        scenario_gather_scatter(comm, rank, world, "chunked", 2003, 32, np.float32, np.float32, np.int64, None, loc="cpu")
        del os.environ["WM_MAPPED_VIA_EXCHANGE"]
        _reload_knobs()
        # (4c) neighbour sampling on a CSR spread over the ranks
        scenario_sampling(comm, rank, world, "distributed", np.int64)
        scenario_sampling(comm, rank, world, "distributed", np.int32)
        scenario_sampling(comm, rank, world, "chunked", np.int32)
        scenario_sampling(comm, rank, world, "continuous", np.int64)
        scenario_sampling(comm, rank, world, "distributed", np.int64, loc="cpu")
    scenario_file_io(comm, rank, world, "/tmp/wgamd_test_%s" % port)
    # the same with the loader's threads forced on and a chunk of a single row (WG_LOAD_* as in the reference)
    os.environ["WG_LOAD_THREADS_PER_RANK"] = "3"
    scenario_file_io(comm, rank, world, "/tmp/wgamd_test_t_%s" % port)
    del os.environ["WG_LOAD_THREADS_PER_RANK"]
    # ... and past the page cache (O_DIRECT through aligned bounce buffers; falls back where the file system refuses it)
    os.environ["WG_LOAD_USE_DIRECTIO"] = "1"
    _reload_knobs()
    scenario_file_io(comm, rank, world, "/var/tmp/wgamd_test_d_%s" % port)
    os.environ["WG_LOAD_THREADS_PER_RANK"] = "2"
    scenario_file_io(comm, rank, world, "/tmp/wgamd_test_dt_%s" % port)
    del os.environ["WG_LOAD_THREADS_PER_RANK"]
    del os.environ["WG_LOAD_USE_DIRECTIO"]
    _reload_knobs()
    # (5) gradient apply, all optimizers
    for kind, params in [("sgd", {"weight_decay": 0.1}), ("adam", {"weight_decay": 0.01}),
                         ("adam", {"adam_w": 1.0, "weight_decay": 0.02}), ("adagrad", {}), ("rmsprop", {"alpha": 0.95})]:
        scenario_gradient_apply(comm, rank, world, kind, params, np.int64 if kind != "adagrad" else np.int32, None)
    # (5') BASELINE config C4 as named — training on MAPPED tables shared by several ranks (reference embedding.cpp:146-323 over
    #      memory_handle.cpp:633-1054): every optimizer, equal and custom partitions, skip-me ids (step 1 of the scenario).
    #      HOST-located mapped tables (one shm segment) run on the CPU backend too; device CHUNKED (hipIpc) and CONTINUOUS
    #      (HIP VMM) need the HIP backend.
    gw = np.random.default_rng(7).uniform(60, 100, world)
    gent = [int(x) for x in (gw / gw.sum() * 1201).astype(int)]
    gent[-1] += 1201 - sum(gent)
    mapped = [("chunked", "cpu"), ("continuous", "cpu")]
    if HIP_MODE:
        mapped = [("chunked", "cuda"), ("continuous", "cuda")] + mapped
    for k, (mt5, loc5) in enumerate(mapped):
        for j, (kind, params) in enumerate([("sgd", {"weight_decay": 0.1}), ("adam", {"weight_decay": 0.01}),
                                            ("adagrad", {}), ("rmsprop", {"alpha": 0.95})]):
            scenario_gradient_apply(comm, rank, world, kind, params, np.int32 if (j + k) % 2 else np.int64,
                                    gent if (j + k) % 2 == 0 else None, mt=mt5, loc=loc5)
    # (5'') sender-side combination of duplicate gradient rows (tree fold): fp32 on either backend, 16-bit tables on the HIP one
    scenario_combined_gradients(comm, rank, world, "distributed", "sgd", {"weight_decay": 0.0})
    scenario_combined_gradients(comm, rank, world, "distributed", "adam", {"weight_decay": 0.01}, idt=np.int32)
    if HIP_MODE:
        scenario_combined_gradients(comm, rank, world, "continuous", "rmsprop", {"alpha": 0.95})
        scenario_combined_gradients(comm, rank, world, "distributed", "sgd", {"weight_decay": 0.0}, tdt=torch.float16)
        scenario_combined_gradients(comm, rank, world, "chunked", "sgd", {"weight_decay": 0.0}, tdt=torch.bfloat16, idt=np.int32)
        scenario_combined_gradients(comm, rank, world, "distributed", "sgd", {"weight_decay": 0.0}, tdt=torch.float16, overflow=True)
    if HIP_MODE:
        scenario_tree_fold(comm, rank, world, "distributed", "sgd", {"weight_decay": 0.0})
        scenario_tree_fold(comm, rank, world, "continuous", "adam", {"weight_decay": 0.01})
        scenario_sgd16(comm, rank, world, torch.float16, 256, -1.0, 0.0, mt="continuous")
        scenario_sgd16(comm, rank, world, torch.float16, 256, -1.0, 0.0, mt="chunked")
        scenario_sgd16(comm, rank, world, torch.bfloat16, 40, 0.05, 0.01, mt="continuous")
    if HIP_MODE:
        # (7) device row caches: HOST tables served and trained through per-rank caches
        scenario_cached_embedding(comm, rank, world, "distributed")
        scenario_cached_embedding(comm, rank, world, "chunked")
        scenario_local_cache(comm, rank, world, "distributed")
        scenario_local_cache(comm, rank, world, "chunked")
        # (6) extension: SGD on 16-bit tables (fp16 scatter-add = lr -1, wd 0)
        scenario_sgd16(comm, rank, world, torch.float16, 256, -1.0, 0.0)
        scenario_sgd16(comm, rank, world, torch.bfloat16, 40, 0.05, 0.01)
    comm.barrier()
    dist.barrier()
    print("RANK %d OK" % rank)
    wgth.finalize()


if __name__ == "__main__":
    main()
