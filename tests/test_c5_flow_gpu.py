"""BASELINE config 5 as ONE flow, hop by hop against the CPU oracle (bit-exact): unweighted 2-hop neighbour sampling on a
CSR graph held in WholeMemory -> append_unique per hop -> feature gather of every node of the sampled sub-graph.
Reference callers: python/pylibwholegraph/pylibwholegraph/torch/graph_structure.py:140-196 (multilayer sampling: one-hop
sample + append_unique per layer), torch/embedding.py gather of the frontier's features (examples/node_classfication.py).
The per-hop sampler seeds are fixed through the `random_seeds` extension so the oracle can replay the chain:
oracle/wm_graph_oracle.c (sampling: tests/wholegraph_ops/graph_sampling_test_utils.cu:306-440; append_unique:
append_unique_test_utils.cu:27-80) and oracle/wm_oracle.c (gather)."""
import numpy as np
import pytest

import oracle
from test_graph_oracle import make_csr

pytestmark = pytest.mark.gpu


def _wm_array(comm, mt, arr, loc="cuda"):
    import torch
    import wholegraph_amd.torch as wgth
    t = wgth.create_wholememory_tensor(comm, mt, loc, [arr.shape[0]], torch.from_numpy(arr).dtype, [1])
    local, _ = t.get_local_tensor(host_view=(loc == "cpu"))
    local.copy_(torch.from_numpy(arr))
    torch.cuda.synchronize()
    return t


@pytest.mark.parametrize("mt", ["chunked", "distributed"])
@pytest.mark.parametrize("id_dtype,fanouts", [(np.int64, [30, 30]), (np.int32, [7, 5]), (np.int64, [15, 10, 5])])
def test_sample_append_unique_gather_chain(gpu_env, mt, id_dtype, fanouts):
    import torch
    import wholegraph_amd.torch as wgth
    n_nodes, dim = 20011, 128
    row_ptr, col = make_csr(n_nodes, 70, 99, id_dtype, heavy=[(3, 4000), (4, 0), (5, 1500), (6, 31)])
    feats = np.random.default_rng(5).standard_normal((n_nodes, dim)).astype(np.float32)
    wrow, wcol = _wm_array(gpu_env, mt, row_ptr), _wm_array(gpu_env, mt, col)
    emb = wgth.create_embedding(gpu_env, mt, "cuda", torch.float32, [n_nodes, dim])
    emb.get_embedding_tensor().get_local_tensor()[0].copy_(torch.from_numpy(feats).cuda())
    torch.cuda.synchronize()
    g = wgth.GraphStructure()
    g.set_csr_graph(wrow, wcol)
    g.set_node_attribute("feat", emb.get_embedding_tensor())

    seeds_np = np.concatenate([[3, 4, 5, 6], np.random.default_rng(2).permutation(n_nodes)[:508]]).astype(id_dtype)
    hop_seeds = [1000 + 17 * i for i in range(len(fanouts))]
    target_gids, edge_indice, csr_row_ptr, csr_col_ind = g.multilayer_sample_without_replacement(
        torch.from_numpy(seeds_np).cuda(), fanouts, random_seeds=hop_seeds)
    x = emb.gather(target_gids[0])
    torch.cuda.synchronize()

    # the oracle's chain, hop by hop from the seeds outwards
    hops = len(fanouts)
    frontier = seeds_np
    assert torch.equal(target_gids[hops].cpu(), torch.from_numpy(seeds_np))
    for depth, fanout in enumerate(fanouts):
        layer = hops - 1 - depth
        off, ids, lid, _ = oracle.sample_unweighted(row_ptr, col, frontier, fanout, hop_seeds[depth], need_egid=False)
        widened, mapping = oracle.append_unique(frontier, ids)
        assert np.array_equal(csr_row_ptr[layer].cpu().numpy(), off), "hop %d: offsets" % depth
        assert np.array_equal(csr_col_ind[layer].cpu().numpy(), mapping), "hop %d: neighbour positions" % depth
        assert np.array_equal(edge_indice[layer][0].cpu().numpy(), mapping), "hop %d: edge index row 0" % depth
        assert np.array_equal(edge_indice[layer][1].cpu().numpy(), lid), "hop %d: centre local ids" % depth
        assert np.array_equal(target_gids[layer].cpu().numpy(), widened), "hop %d: widened frontier" % depth
        assert target_gids[layer].dtype == torch.from_numpy(seeds_np).dtype
        # sampled neighbour = widened[mapping]: every sampled edge exists in the graph
        src = frontier[lid]
        dst = widened[mapping]
        assert np.array_equal(dst, ids)
        for s, d in list(zip(src, dst))[:500]:
            assert d in col[row_ptr[s]:row_ptr[s + 1]]
        frontier = widened
    # features of the outermost frontier: the gather of the hot path, against the oracle's gather
    tab = oracle.ShardedTable.from_full(feats, 1)
    exp = np.zeros((len(frontier), dim), np.float32)
    oracle.gather(tab, frontier, exp)
    assert x.cpu().numpy().tobytes() == exp.tobytes(), "feature rows of the sampled sub-graph"
    assert len(frontier) > len(seeds_np) * 5          # the sub-graph really grew
    wgth.destroy_embedding(emb)
    wgth.destroy_wholememory_tensor(wrow)
    wgth.destroy_wholememory_tensor(wcol)


def _two_ops(wops, gops, wrow, wcol, frontier, fanout, seed):
    off, ids, lid = wops.unweighted_sample_without_replacement(wrow.wmb_tensor, wcol.wmb_tensor, frontier, fanout, seed, True, False)
    uniq, pos = gops.append_unique(frontier, ids, need_neighbor_raw_to_unique=True)
    return off, uniq, pos, lid


@pytest.mark.parametrize("mt", ["chunked", "continuous"])
@pytest.mark.parametrize("id_dtype", [np.int32, np.int64])
@pytest.mark.parametrize("fanout", [1, 30, 200])
def test_fused_hop_equals_the_two_ops(gpu_env, mt, id_dtype, fanout):
    """wholememory_ext_sample_append_unique (one call, one host round trip) against the sampler followed by append_unique:
    the four outputs are equal element for element — frontiers with duplicates, zero-degree and heavy nodes included."""
    import torch
    import wholegraph_amd.torch.wholegraph_ops as wops
    import wholegraph_amd.torch.graph_ops as gops
    n_nodes = 9001
    row_ptr, col = make_csr(n_nodes, 40, 7, id_dtype, heavy=[(3, 3000), (4, 0), (5, 700), (6, 31)])
    wrow, wcol = _wm_array(gpu_env, mt, row_ptr), _wm_array(gpu_env, mt, col)
    rng = np.random.default_rng(fanout)
    for n_front in (1, 4, 777):
        front = np.concatenate([[3, 4, 5, 6, 4], rng.integers(0, n_nodes, n_front)])[:max(n_front, 1)].astype(id_dtype)
        frontier = torch.from_numpy(front).cuda()
        fused = wops.sample_append_unique(wrow.wmb_tensor, wcol.wmb_tensor, frontier, fanout, 99 + n_front)
        assert fused is not None
        want = _two_ops(wops, gops, wrow, wcol, frontier, fanout, 99 + n_front)
        for name, a, b in zip(("offsets", "unique", "neighbour positions", "centre ids"), fused, want):
            assert a.dtype == b.dtype and torch.equal(a, b), "%s differ (frontier %d, fan-out %d)" % (name, n_front, fanout)
        o_off, o_ids, o_lid, _ = oracle.sample_unweighted(row_ptr, col, front, fanout, 99 + n_front, need_egid=False)
        o_uniq, o_map = oracle.append_unique(front, o_ids)
        assert np.array_equal(fused[1].cpu().numpy(), o_uniq) and np.array_equal(fused[2].cpu().numpy(), o_map)
    # nothing to sample at all: a frontier of zero-degree nodes
    lonely = torch.from_numpy(np.array([4, 4, 4], dtype=id_dtype)).cuda()
    off, uniq, pos, lid = wops.sample_append_unique(wrow.wmb_tensor, wcol.wmb_tensor, lonely, fanout, 5)
    assert off.tolist() == [0, 0, 0, 0] and uniq.tolist() == [4, 4, 4] and pos.numel() == 0 and lid.numel() == 0


def test_fused_hop_on_a_host_located_graph(gpu_env):
    """the CSR in pinned host memory (memory_location "cpu"), read by the kernels over PCIe: the fused hop still applies"""
    import torch
    import wholegraph_amd.torch.wholegraph_ops as wops
    import wholegraph_amd.torch.graph_ops as gops
    row_ptr, col = make_csr(3001, 25, 5, np.int64, heavy=[(3, 900), (4, 0)])
    wrow, wcol = _wm_array(gpu_env, "chunked", row_ptr, "cpu"), _wm_array(gpu_env, "chunked", col, "cpu")
    frontier = torch.from_numpy(np.random.default_rng(3).integers(0, 3001, 400).astype(np.int64)).cuda()
    fused = wops.sample_append_unique(wrow.wmb_tensor, wcol.wmb_tensor, frontier, 10, 123)
    assert fused is not None
    for a, b in zip(fused, _two_ops(wops, gops, wrow, wcol, frontier, 10, 123)):
        assert torch.equal(a, b)


def test_fused_hop_declines_what_it_does_not_cover(gpu_env):
    """None (WHOLEMEMORY_NOT_SUPPORTED, nothing queued) for a DISTRIBUTED CSR, for column ids of another dtype than the
    frontier's, for an empty frontier and for max_sample_count <= 0: GraphStructure then runs the two ops."""
    import torch
    import wholegraph_amd.torch.wholegraph_ops as wops
    row_ptr, col = make_csr(500, 9, 3, np.int32)
    wrow, wcol = _wm_array(gpu_env, "chunked", row_ptr), _wm_array(gpu_env, "chunked", col)
    f32 = torch.arange(10, dtype=torch.int32, device="cuda")
    assert wops.sample_append_unique(wrow.wmb_tensor, wcol.wmb_tensor, f32.long(), 5, 1) is None      # int64 frontier, int32 columns
    assert wops.sample_append_unique(wrow.wmb_tensor, wcol.wmb_tensor, f32[:0], 5, 1) is None
    assert wops.sample_append_unique(wrow.wmb_tensor, wcol.wmb_tensor, f32, 0, 1) is None
    assert wops.sample_append_unique(wrow.wmb_tensor, wcol.wmb_tensor, f32, 5, 1) is not None
    drow, dcol = _wm_array(gpu_env, "distributed", row_ptr), _wm_array(gpu_env, "distributed", col)
    assert wops.sample_append_unique(drow.wmb_tensor, dcol.wmb_tensor, f32, 5, 1) is None


def test_fused_hop_big_frontier_route():
    """A frontier whose upper bound exceeds what the table route of append_unique serves (forced here with
    WM_AU_TABLE_MAX=0): the fused op learns the sample count first and takes the sort route — same outputs."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch, oracle
from test_graph_oracle import make_csr
import wholegraph_amd.torch as wgth
import wholegraph_amd.torch.wholegraph_ops as wops
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_ERROR))
comm = wgth.create_group_communicator(1)
for dt in (np.int32, np.int64):
    row_ptr, col = make_csr(4000, 30, 11, dt, heavy=[(3, 900), (4, 0)])
    ws = []
    for arr in (row_ptr, col):
        t = wgth.create_wholememory_tensor(comm, "chunked", "cuda", [arr.shape[0]], torch.from_numpy(arr).dtype, [1])
        t.get_local_tensor()[0].copy_(torch.from_numpy(arr)); ws.append(t)
    front = np.random.default_rng(1).integers(0, 4000, 600).astype(dt)
    off, uniq, pos, lid = wops.sample_append_unique(ws[0].wmb_tensor, ws[1].wmb_tensor, torch.from_numpy(front).cuda(), 12, 77)
    o_off, o_ids, o_lid, _ = oracle.sample_unweighted(row_ptr, col, front, 12, 77, need_egid=False)
    o_uniq, o_map = oracle.append_unique(front, o_ids)
    assert np.array_equal(off.cpu().numpy(), o_off) and np.array_equal(uniq.cpu().numpy(), o_uniq)
    assert np.array_equal(pos.cpu().numpy(), o_map) and np.array_equal(lid.cpu().numpy(), o_lid)
print("BIG_ROUTE_OK")
""" % (root, os.path.join(root, "tests"))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, WM_AU_TABLE_MAX="0"), capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0 and "BIG_ROUTE_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


@pytest.mark.parametrize("mt,loc", [("chunked", "cuda"), ("continuous", "cuda"), ("chunked", "cpu")])
@pytest.mark.parametrize("id_dtype,fanouts", [(np.int32, [30, 30]), (np.int64, [15, 10, 5]), (np.int32, [1]), (np.int64, [200, 3])])
def test_chain_with_one_host_round_trip_equals_hop_by_hop(gpu_env, monkeypatch, mt, loc, id_dtype, fanouts):
    """wholememory_ext_multilayer_sample (upper-bound-sized buffers, counts kept on the device between hops, ONE stream
    synchronise for the whole chain) against the hop-by-hop route (WM_MULTILAYER_CHAIN=0: one fused call and one host round
    trip per hop) with the same per-hop seeds: every returned tensor equal, bit for bit."""
    import torch
    import wholegraph_amd.torch as wgth
    n_nodes = 30011
    row_ptr, col = make_csr(n_nodes, 60, 7, id_dtype, heavy=[(3, 5000), (4, 0), (5, 1500), (6, 31), (7, 201)])
    wrow, wcol = _wm_array(gpu_env, mt, row_ptr, loc), _wm_array(gpu_env, mt, col, loc)
    g = wgth.GraphStructure()
    g.set_csr_graph(wrow, wcol)
    seeds = torch.from_numpy(np.concatenate([[3, 4, 5, 6, 7, 7], np.random.default_rng(9).permutation(n_nodes)[:700]]).astype(id_dtype)).cuda()
    hop_seeds = [77 + 5 * i for i in range(len(fanouts))]
    monkeypatch.setenv("WM_MULTILAYER_CHAIN", "0")
    ref = g.multilayer_sample_without_replacement(seeds, fanouts, random_seeds=hop_seeds)
    monkeypatch.setenv("WM_MULTILAYER_CHAIN", "1")
    got = g.multilayer_sample_without_replacement(seeds, fanouts, random_seeds=hop_seeds)
    torch.cuda.synchronize()
    for name, a_list, b_list in zip(("target_gids", "edge_indice", "csr_row_ptr", "csr_col_ind"), got, ref):
        assert len(a_list) == len(b_list)
        for layer, (a, b) in enumerate(zip(a_list, b_list)):
            assert a.dtype == b.dtype and tuple(a.shape) == tuple(b.shape), "%s[%d]: %s vs %s" % (name, layer, a.shape, b.shape)
            assert torch.equal(a, b), "%s[%d] differs" % (name, layer)
    # the chain really ran as one call: its outputs are views of upper-bound buffers
    assert got[0][0].untyped_storage().nbytes() >= ref[0][0].untyped_storage().nbytes()


@pytest.mark.parametrize("mt", ["chunked", "distributed"])
@pytest.mark.parametrize("id_dtype,fanouts", [(np.int32, [30, 30]), (np.int64, [7, 5, 3])])
def test_deferred_chain_feeds_the_gather_before_the_host_knows_the_counts(gpu_env, mt, id_dtype, fanouts):
    """GraphStructure.multilayer_sample_begin: the chain is queued, the outermost frontier is handed to the feature gather at
    its upper-bound size with the entries behind the sampled nodes set to -1 (skipped by the gather), and result() — the one
    host round trip — returns exactly what multilayer_sample_without_replacement returns with the same seeds. On a
    DISTRIBUTED CSR the chain does not apply: the handle then holds an eager hop-by-hop sample, same contract."""
    import torch
    import wholegraph_amd.torch as wgth
    n_nodes, dim = 20011, 32
    row_ptr, col = make_csr(n_nodes, 40, 11, id_dtype, heavy=[(3, 3000), (4, 0), (5, 31)])
    wrow, wcol = _wm_array(gpu_env, mt, row_ptr), _wm_array(gpu_env, mt, col)
    g = wgth.GraphStructure()
    g.set_csr_graph(wrow, wcol)
    emb = wgth.create_embedding(gpu_env, "chunked", "cuda", torch.float32, [n_nodes, dim])
    local, _ = emb.get_embedding_tensor().get_local_tensor()
    local.copy_(torch.arange(n_nodes, device="cuda", dtype=torch.float32).unsqueeze(1) + torch.arange(dim, device="cuda") / 64.0)
    seeds = torch.from_numpy(np.concatenate([[3, 4, 5], np.random.default_rng(5).permutation(n_nodes)[:300]]).astype(id_dtype)).cuda()
    hop_seeds = [11 + 3 * i for i in range(len(fanouts))]
    ref = g.multilayer_sample_without_replacement(seeds, fanouts, random_seeds=hop_seeds)
    h = g.multilayer_sample_begin(seeds, fanouts, random_seeds=hop_seeds)
    padded = h.padded_frontier
    out = torch.full((padded.shape[0], dim), -7.0, device="cuda")
    emb.gather(padded, out=out)                      # queued behind the sampling kernels, before result()
    got = h.result()
    torch.cuda.synchronize()
    for name, a_list, b_list in zip(("target_gids", "edge_indice", "csr_row_ptr", "csr_col_ind"), got, ref):
        assert len(a_list) == len(b_list)
        for layer, (a, b) in enumerate(zip(a_list, b_list)):
            assert a.dtype == b.dtype and torch.equal(a, b), "%s[%d] differs" % (name, layer)
    n = got[0][0].shape[0]
    assert torch.equal(padded[:n], got[0][0]) and bool((padded[n:] == -1).all())
    assert torch.equal(out[:n], emb.gather(got[0][0])) and bool((out[n:] == -7.0).all())   # rows behind the frontier untouched
    if mt == "chunked":
        assert padded.shape[0] > n                   # the chain really handed out its upper-bound array
    wgth.destroy_embedding(emb)


def test_deferred_chain_and_gather_replay_from_a_graph(gpu_env):
    """The queued chain + the gather on its padded frontier enqueue kernels on the caller's stream and nothing else (scratch
    from torch's allocator through the env functions, the scans' state in a buffer the library keeps, counts left in pinned
    memory by the last kernel): the whole C5 step can be captured into a hipGraph once and replayed."""
    import torch
    import wholegraph_amd.torch as wgth
    n_nodes, dim, fanouts, hop_seeds = 20011, 32, [30, 30], [5, 6]
    row_ptr, col = make_csr(n_nodes, 40, 11, np.int32, heavy=[(3, 3000), (4, 0)])
    wrow, wcol = _wm_array(gpu_env, "chunked", row_ptr), _wm_array(gpu_env, "chunked", col)
    g = wgth.GraphStructure()
    g.set_csr_graph(wrow, wcol)
    emb = wgth.create_embedding(gpu_env, "chunked", "cuda", torch.float32, [n_nodes, dim])
    local, _ = emb.get_embedding_tensor().get_local_tensor()
    local.copy_(torch.arange(n_nodes, device="cuda", dtype=torch.float32).unsqueeze(1).expand(n_nodes, dim))
    seeds = torch.from_numpy(np.concatenate([[3, 4], np.random.default_rng(5).permutation(n_nodes)[:200]]).astype(np.int32)).cuda()
    ref = g.multilayer_sample_without_replacement(seeds, fanouts, random_seeds=hop_seeds)
    room = seeds.shape[0] * 31 * 31
    out = torch.empty((room, dim), device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # warm-up outside the capture (allocator, scan state, pinned counts)
        h = g.multilayer_sample_begin(seeds, fanouts, random_seeds=hop_seeds)
        emb.gather(h.padded_frontier, out=out)
        h.result()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        h = g.multilayer_sample_begin(seeds, fanouts, random_seeds=hop_seeds)
        emb.gather(h.padded_frontier, out=out)
    for _ in range(3):
        out.fill_(-7.0)
        h.padded_frontier.fill_(123)
        graph.replay()
        torch.cuda.synchronize()
    got = h.result()
    for a_list, b_list in zip(got, ref):
        for a, b in zip(a_list, b_list):
            assert torch.equal(a, b)
    n = got[0][0].shape[0]
    assert torch.equal(out[:n, 0], got[0][0].float()) and bool((out[n:] == -7.0).all())
    wgth.destroy_embedding(emb)


def test_chain_declines_upper_bounds_beyond_the_table_route(gpu_env):
    """65536 seeds x [30, 30, 30]: the third hop's upper bound (63 M centres + 1.9 G samples) is past what append_unique's hash
    table takes from device-side counts: the library answers NOT_SUPPORTED to the query, before any buffer is allocated or
    anything queued, and the caller goes hop by hop. 65536 x [30, 30] (63 M keys at most) is taken."""
    import torch
    import wholegraph_amd.torch as wgth
    from wholegraph_amd.torch import wholegraph_ops
    row_ptr, col = make_csr(5003, 20, 3, np.int32)
    wrow, wcol = _wm_array(gpu_env, "chunked", row_ptr), _wm_array(gpu_env, "chunked", col)
    seeds = torch.randint(0, 5003, (65536,), dtype=torch.int32, device="cuda")
    free0 = torch.cuda.mem_get_info()[0]
    assert wholegraph_ops.multilayer_sample(wrow.wmb_tensor, wcol.wmb_tensor, seeds, [30, 30, 30], [1, 2, 3]) is None
    assert torch.cuda.mem_get_info()[0] >= free0 - (64 << 20), "a declined chain must not have allocated its buffers"
    assert wholegraph_ops.multilayer_sample(wrow.wmb_tensor, wcol.wmb_tensor, seeds[:0], [5], [1]) is None
    assert wholegraph_ops.multilayer_sample(wrow.wmb_tensor, wcol.wmb_tensor, seeds[:4096], [30, 30], [1, 2]) is not None
