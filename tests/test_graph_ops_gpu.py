"""GPU parity of neighbour sampling / append_unique / add_csr_self_loop through the C ABI vs the CPU oracle.

Parameter sets follow the reference's tests:
  cpp/tests/wholegraph_ops/wholegraph_csr_unweighted_sample_without_replacement_tests.cu:290-375
      (memory types continuous/chunked x host/device, max_sample_count 10/30/40/128/500/1025..., int32/int64 ids)
  python/.../tests/wholegraph_torch/ops/test_wholegraph_unweighted_sample.py:196-245
  cpp/tests/graph_ops/append_unique_tests.cu, csr_add_self_loop_tests.cu, python test_graph_append_unique.py
Bit-exact: samples are a pure function of (seed, center position, max_sample_count, CSR row).
"""
import os

import numpy as np
import pytest

import oracle
from test_graph_oracle import make_csr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _wm_csr(comm, mt, loc, row_ptr, col):
    import torch
    import wholegraph_amd.torch as wgth
    tensors = []
    for arr in (row_ptr, col):
        t = wgth.create_wholememory_tensor(comm, mt, loc, [arr.shape[0]], torch.from_numpy(arr).dtype, [1])
        local, start = t.get_local_tensor(host_view=(loc == "cpu"))
        local.copy_(torch.from_numpy(arr))
        tensors.append(t)
    torch.cuda.synchronize()
    return tensors


GRAPH = None


def _graph(col_dtype):
    global GRAPH
    if GRAPH is None:
        GRAPH = make_csr(3000, 90, 123, np.int64,
                         heavy=[(7, 5000), (8, 1024), (9, 1025), (10, 1026), (11, 2048), (12, 0), (13, 1), (14, 30), (15, 31)])
    return GRAPH[0], GRAPH[1].astype(col_dtype)


@pytest.mark.parametrize("mt,loc", [("continuous", "cuda"), ("chunked", "cuda"), ("continuous", "cpu"), ("chunked", "cpu"),
                                    ("distributed", "cuda"), ("distributed", "cpu")])
@pytest.mark.parametrize("center_dtype,col_dtype", [(np.int64, np.int64), (np.int32, np.int32), (np.int64, np.int32),
                                                    (np.int32, np.int64)])
def test_unweighted_sample_parity(gpu_env, mt, loc, center_dtype, col_dtype):
    import torch
    import wholegraph_amd.torch as wgth
    row_ptr, col = _graph(col_dtype)
    wrow, wcol = _wm_csr(gpu_env, mt, loc, row_ptr, col)
    rng = np.random.default_rng(17)
    centers = np.concatenate([np.arange(0, 20), rng.integers(0, 3000, 1500)]).astype(center_dtype)
    g = wgth.GraphStructure()
    g.set_csr_graph(wrow, wcol)
    for m in (-1, 1, 10, 30, 32, 33, 40, 64, 65, 128, 500, 1024, 1025, 1500):
        seed = 1000003 * (m + 5) + 12345678901234567
        off, ids, lid, egid = g.unweighted_sample_without_replacement_one_hop(
            torch.from_numpy(centers).cuda(), m, random_seed=seed, need_center_local_output=True, need_edge_output=True)
        o_off, o_ids, o_lid, o_egid = oracle.sample_unweighted(row_ptr, col, centers, m, seed)
        assert off.dtype == torch.int32 and lid.dtype == torch.int32 and egid.dtype == torch.int64
        assert ids.dtype == torch.from_numpy(col).dtype
        assert np.array_equal(off.cpu().numpy(), o_off), "offsets differ at max_sample_count=%d" % m
        assert np.array_equal(egid.cpu().numpy(), o_egid), "edge ids differ at max_sample_count=%d" % m
        assert np.array_equal(ids.cpu().numpy(), o_ids), "sampled ids differ at max_sample_count=%d" % m
        assert np.array_equal(lid.cpu().numpy(), o_lid)
    wgth.destroy_wholememory_tensor(wrow)
    wgth.destroy_wholememory_tensor(wcol)


def test_unweighted_sample_output_variants_and_plain_tensors(gpu_env):
    """Optional outputs; CSR held in plain device tensors (no WholeMemory handle) wrapped on the fly."""
    import torch
    import wholegraph_amd.torch.wholegraph_ops as wops
    from wholegraph_amd.torch.wholegraph_env import wrap_torch_tensor
    row_ptr, col = _graph(np.int64)
    drow, dcol = torch.from_numpy(row_ptr).cuda(), torch.from_numpy(col).cuda()
    wr, wc = wrap_torch_tensor(drow), wrap_torch_tensor(dcol)
    centers = np.array([7, 8, 12, 13, 100, 200, 7], dtype=np.int64)
    o_off, o_ids, o_lid, o_egid = oracle.sample_unweighted(row_ptr, col, centers, 25, 42)
    dc = torch.from_numpy(centers).cuda()
    r = wops.unweighted_sample_without_replacement(wr.handle, wc.handle, dc, 25, 42)
    assert len(r) == 2 and np.array_equal(r[1].cpu().numpy(), o_ids) and np.array_equal(r[0].cpu().numpy(), o_off)
    r = wops.unweighted_sample_without_replacement(wr.handle, wc.handle, dc, 25, 42, need_center_local_output=True)
    assert len(r) == 3 and np.array_equal(r[2].cpu().numpy(), o_lid)
    r = wops.unweighted_sample_without_replacement(wr.handle, wc.handle, dc, 25, 42, need_edge_output=True)
    assert len(r) == 3 and np.array_equal(r[2].cpu().numpy(), o_egid)
    # no center nodes, and center nodes without neighbours
    r = wops.unweighted_sample_without_replacement(wr.handle, wc.handle, dc[:0], 25, 42, need_edge_output=True)
    assert r[0].cpu().tolist() == [0] and r[1].numel() == 0 and r[2].numel() == 0
    r = wops.unweighted_sample_without_replacement(wr.handle, wc.handle, torch.tensor([12, 12], device="cuda"), 25, 42)
    assert r[0].cpu().tolist() == [0, 0, 0] and r[1].numel() == 0
    # a random seed is drawn when none is given: two calls differ (5000-neighbour node, 25 samples)
    a = wops.unweighted_sample_without_replacement(wr.handle, wc.handle, dc[:1], 25)[1]
    b = wops.unweighted_sample_without_replacement(wr.handle, wc.handle, dc[:1], 25)[1]
    assert not torch.equal(a, b)


def test_unweighted_sample_is_uniform(gpu_env):
    """Every neighbour of a node is picked with probability M/N (chi-square over 4000 seeds)."""
    import torch
    import wholegraph_amd.torch.wholegraph_ops as wops
    from wholegraph_amd.torch.wholegraph_env import wrap_torch_tensor
    n_nb, m, trials = 50, 10, 4000
    row_ptr = np.arange(0, (trials + 1) * n_nb, n_nb, dtype=np.int64)   # `trials` nodes with the same 50 neighbours
    col = np.tile(np.arange(n_nb, dtype=np.int32), trials)
    wr, wc = wrap_torch_tensor(torch.from_numpy(row_ptr).cuda()), wrap_torch_tensor(torch.from_numpy(col).cuda())
    centers = torch.arange(trials, device="cuda", dtype=torch.int32)
    off, ids = wops.unweighted_sample_without_replacement(wr.handle, wc.handle, centers, m, 20260928)
    ids = ids.cpu().numpy().reshape(trials, m)
    assert all(len(set(r.tolist())) == m for r in ids)
    counts = np.bincount(ids.ravel(), minlength=n_nb).astype(np.float64)
    expect = trials * m / n_nb
    chi2 = ((counts - expect) ** 2 / expect).sum()
    assert chi2 < 100.0, "chi-square %.1f over 49 dof" % chi2   # p(chi2 > 100) ~ 2e-5
    # position i of the sample is itself uniform
    first = np.bincount(ids[:, 0], minlength=n_nb).astype(np.float64)
    assert ((first - trials / n_nb) ** 2 / (trials / n_nb)).sum() < 100.0


def test_sample_argument_errors(gpu_env):
    import torch
    from wholegraph_amd import binding as wmb
    import wholegraph_amd.torch.wholegraph_ops as wops
    from wholegraph_amd.torch.wholegraph_env import wrap_torch_tensor
    row = torch.tensor([0, 2, 4], device="cuda", dtype=torch.int64)
    col = torch.tensor([1, 0, 1, 0], device="cuda", dtype=torch.int64)
    centers = torch.tensor([0, 1], device="cuda")
    wr, wc = wrap_torch_tensor(row), wrap_torch_tensor(col)
    wr32 = wrap_torch_tensor(row.int())
    with pytest.raises(wmb.WholeMemoryError):   # row_ptr must be int64
        wops.unweighted_sample_without_replacement(wr32.handle, wc.handle, centers, 2, 1)
    with pytest.raises(wmb.WholeMemoryError):   # float ids
        wops.unweighted_sample_without_replacement(wr.handle, wc.handle, centers.float(), 2, 1)
    ww = wrap_torch_tensor(torch.ones(4, device="cuda"))
    with pytest.raises(wmb.WholeMemoryError) as e:     # the in-LDS weighted selection covers 1..8192
        wops.weighted_sample_without_replacement(wr.handle, wc.handle, ww.handle, centers, 8193, 1)
    assert "NOT_IMPLEMENTED" in str(e.value)
    with pytest.raises(wmb.WholeMemoryError):          # integer weights
        wops.weighted_sample_without_replacement(wr.handle, wc.handle, wc.handle, centers, 2, 1)
    wshort = wrap_torch_tensor(torch.ones(3, device="cuda"))
    with pytest.raises(wmb.WholeMemoryError):          # one weight per edge
        wops.weighted_sample_without_replacement(wr.handle, wc.handle, wshort.handle, centers, 2, 1)


@pytest.mark.parametrize("mt,loc", [("continuous", "cuda"), ("chunked", "cuda"), ("chunked", "cpu")])
@pytest.mark.parametrize("center_dtype,col_dtype,wdtype", [(np.int64, np.int64, np.float32), (np.int32, np.int32, np.float64),
                                                           (np.int64, np.int32, np.float32)])
def test_weighted_sample_parity(gpu_env, mt, loc, center_dtype, col_dtype, wdtype):
    """cpp/tests/wholegraph_ops/wholegraph_csr_weighted_sample_without_replacement_tests.cu:438-470 (memory types,
    max_sample_count 10 / 300, int32 / int64 centers); keys are bit-identical on device and in the oracle, so the
    comparison is exact and ordered (the reference compares per-center sorted sets)."""
    import torch
    import wholegraph_amd.torch as wgth
    row_ptr, col = _graph(col_dtype)
    weights = (np.random.default_rng(99).random(col.shape[0]) * 3 + 0.01).astype(wdtype)
    wrow, wcol = _wm_csr(gpu_env, mt, loc, row_ptr, col)
    wwgt = wgth.create_wholememory_tensor(gpu_env, mt, loc, [weights.shape[0]], torch.from_numpy(weights).dtype, [1])
    wwgt.get_local_tensor(host_view=(loc == "cpu"))[0].copy_(torch.from_numpy(weights))
    torch.cuda.synchronize()
    g = wgth.GraphStructure()
    g.set_csr_graph(wrow, wcol)
    g.set_edge_attribute("w", wwgt)
    rng = np.random.default_rng(3)
    centers = np.concatenate([np.arange(0, 20), rng.integers(0, 3000, 700)]).astype(center_dtype)
    for m in (-1, 1, 10, 30, 64, 65, 128, 256, 257, 300, 1000, 1024, 1025, 3000, 4999):
        seed = 77 * (m + 3) + 987654321987
        off, ids, lid, egid = g.weighted_sample_without_replacement_one_hop(
            "w", torch.from_numpy(centers).cuda(), m, random_seed=seed, need_center_local_output=True, need_edge_output=True)
        o_off, o_ids, o_lid, o_egid = oracle.sample_weighted(row_ptr, col, weights, centers, m, seed)
        assert np.array_equal(off.cpu().numpy(), o_off)
        assert np.array_equal(egid.cpu().numpy(), o_egid), "edge ids differ at max_sample_count=%d" % m
        assert np.array_equal(ids.cpu().numpy(), o_ids) and np.array_equal(lid.cpu().numpy(), o_lid)
    for t in (wrow, wcol, wwgt):
        wgth.destroy_wholememory_tensor(t)


def test_weighted_sample_follows_the_weights(gpu_env):
    """With one sample per node, neighbour i is picked with probability w_i / sum(w)."""
    import torch
    import wholegraph_amd.torch.wholegraph_ops as wops
    from wholegraph_amd.torch.wholegraph_env import wrap_torch_tensor
    n_nb, trials = 8, 40000
    w = np.array([1, 2, 3, 4, 5, 6, 7, 12], dtype=np.float32)
    row_ptr = np.arange(0, (trials + 1) * n_nb, n_nb, dtype=np.int64)
    col = np.tile(np.arange(n_nb, dtype=np.int32), trials)
    wts = np.tile(w, trials)
    hold = [torch.from_numpy(a).cuda() for a in (row_ptr, col, wts)]
    wr, wc, ww = [wrap_torch_tensor(t) for t in hold]
    centers = torch.arange(trials, device="cuda", dtype=torch.int32)
    off, ids = wops.weighted_sample_without_replacement(wr.handle, wc.handle, ww.handle, centers, 1, 424242)
    counts = np.bincount(ids.cpu().numpy(), minlength=n_nb).astype(np.float64)
    expect = trials * w / w.sum()
    chi2 = ((counts - expect) ** 2 / expect).sum()
    assert chi2 < 40.0, "chi-square %.1f over 7 dof" % chi2   # p ~ 1e-6
    # and without replacement: 3 of 8, all distinct
    off, ids = wops.weighted_sample_without_replacement(wr.handle, wc.handle, ww.handle, centers, 3, 7)
    ids = ids.cpu().numpy().reshape(trials, 3)
    assert all(len(set(r.tolist())) == 3 for r in ids[:5000])


@pytest.mark.parametrize("np_dtype", [np.int32, np.int64])
@pytest.mark.parametrize("nt,nn,universe", [(700, 9000, 5000), (1, 1, 10), (0, 5000, 300), (3000, 0, 10 ** 6),
                                            (50000, 400000, 150000), (1000, 20000, 2 ** 31 - 1),
                                            (100000, 2200000, 3000000)])   # the last one: > 2 M 32-bit keys = the sort route
def test_append_unique_parity(gpu_env, np_dtype, nt, nn, universe):
    import torch
    import wholegraph_amd.torch.graph_ops as gops
    rng = np.random.default_rng(nt + nn)
    if universe > 4 * nt:
        targets = np.unique(rng.integers(0, universe, 2 * nt + 8))
        rng.shuffle(targets)
        targets = targets[:nt].astype(np_dtype)
    else:
        targets = rng.permutation(universe)[:nt].astype(np_dtype)
    neighbors = rng.integers(0, universe, nn).astype(np_dtype)
    o_uniq, o_map = oracle.append_unique(targets, neighbors)
    uniq, mapping = gops.append_unique(torch.from_numpy(targets).cuda(), torch.from_numpy(neighbors).cuda(), True)
    assert uniq.dtype == torch.from_numpy(targets).dtype and mapping.dtype == torch.int32
    assert np.array_equal(uniq.cpu().numpy(), o_uniq)
    assert np.array_equal(mapping.cpu().numpy(), o_map)
    only = gops.append_unique(torch.from_numpy(targets).cuda(), torch.from_numpy(neighbors).cuda())
    assert torch.equal(only, uniq)


@pytest.mark.parametrize("np_dtype", [np.int32, np.int64])
@pytest.mark.parametrize("case", ["all_ones_id", "extremes", "hot_id", "duplicate_targets", "colliding_hashes"])
def test_append_unique_table_edges(gpu_env, np_dtype, case):
    """The hash-table append_unique at its edges: the id whose bits are all ones (the table's "empty" pattern, it has a slot
    of its own), the extreme values of the dtype, one id offered by every neighbour position (the atomic-min hot spot),
    duplicates among the targets (the first one wins, as in the oracle), and ids a multiple of a large power of two apart."""
    import torch
    import wholegraph_amd.torch.graph_ops as gops
    rng = np.random.default_rng(5)
    info = np.iinfo(np_dtype)
    if case == "all_ones_id":
        targets = np.array([7, -1, 3], dtype=np_dtype)
        neighbors = np.concatenate([rng.integers(-3, 12, 5000), np.full(100, -1)]).astype(np_dtype)
        rng.shuffle(neighbors)
    elif case == "extremes":
        targets = np.array([info.max, 0], dtype=np_dtype)
        neighbors = rng.choice(np.array([info.min, info.max, info.min + 1, info.max - 1, -1, 0, 1], dtype=np_dtype), 4000)
    elif case == "hot_id":
        targets = np.arange(10, 20, dtype=np_dtype)
        neighbors = np.full(300000, 123456, dtype=np_dtype)
        neighbors[::1000] = 15
        neighbors[7::5000] = 99
    elif case == "duplicate_targets":
        targets = np.array([5, 9, 5, 5, 2, 9], dtype=np_dtype)
        neighbors = rng.integers(0, 12, 3000).astype(np_dtype)
    else:
        step = 1 << 20
        targets = (np.arange(50) * step).astype(np_dtype)
        neighbors = (rng.integers(0, 1500, 60000) * step).astype(np_dtype)
    o_uniq, o_map = oracle.append_unique(targets, neighbors)
    uniq, mapping = gops.append_unique(torch.from_numpy(targets).cuda(), torch.from_numpy(neighbors).cuda(), True)
    assert np.array_equal(uniq.cpu().numpy(), o_uniq)
    assert np.array_equal(mapping.cpu().numpy(), o_map)
    assert np.array_equal(o_uniq[o_map], neighbors)


@pytest.mark.parametrize("np_dtype", [np.int32, np.int64])
@pytest.mark.parametrize("variant", ["merged_look", "merged_direct", "round4", "round4_direct"])
@pytest.mark.parametrize("dist", ["uniform", "zipf", "one_id_and_all_ones"])
def test_append_unique_insert_variants(gpu_env, knobs, np_dtype, variant, dist):
    """Round 5: the table insert merges a workgroup's keys in LDS before it goes to memory, with or without a look at the slot
    before the compare-and-swap (graph.hip: au_insert_merged_kernel, au_direct_cas). Every variant — and rounds 3-4's kernel,
    kept for A/B — gives the oracle's unique list and mapping on uniform ids, on Zipf ids (one id in ~5 % of the positions:
    every workgroup merges it) and on the degenerate inputs (one id everywhere; the id whose bits are all ones among them)."""
    import torch
    import wholegraph_amd.torch.graph_ops as gops
    env = {"merged_look": {"WM_AU_DIRECT_CAS": 0}, "merged_direct": {"WM_AU_DIRECT_CAS": 1},
           "round4": {"WM_AU_MERGE": 0, "WM_AU_DIRECT_CAS": 0}, "round4_direct": {"WM_AU_MERGE": 0, "WM_AU_DIRECT_CAS": 2}}[variant]
    for k, v in env.items():
        knobs.set(k, v)
    rng = np.random.default_rng(17)
    nt, nn, universe = 3001, 200003, 5_000_000
    targets = rng.permutation(universe)[:nt].astype(np_dtype)
    if dist == "uniform":
        neighbors = rng.integers(0, universe, nn).astype(np_dtype)
    elif dist == "zipf":
        neighbors = ((rng.zipf(1.05, nn).astype(np.uint64) * np.uint64(2654435761)) % np.uint64(universe)).astype(np_dtype)
        neighbors[::7] = targets[rng.integers(0, nt, len(neighbors[::7]))]      # and ids the targets hold
    else:
        neighbors = np.full(nn, 424242, dtype=np_dtype)
        neighbors[5::1000] = -1
        neighbors[9::3000] = targets[0]
    o_uniq, o_map = oracle.append_unique(targets, neighbors)
    uniq, mapping = gops.append_unique(torch.from_numpy(targets).cuda(), torch.from_numpy(neighbors).cuda(), True)
    assert np.array_equal(uniq.cpu().numpy(), o_uniq)
    assert np.array_equal(mapping.cpu().numpy(), o_map)


@pytest.mark.parametrize("limit", ["0", "1000000000"])
def test_append_unique_both_routes(limit):
    """Both routes of append_unique (hash table / radix sort; the library picks by size and id width) forced over the same
    inputs through WM_AU_TABLE_MAX, in a process of their own (the switch is read once): equal to the oracle either way."""
    import subprocess
    import sys
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np, torch, oracle
import wholegraph_amd.torch as wgth
import wholegraph_amd.torch.graph_ops as gops
from wholegraph_amd import binding as wmb
wmb.check(wmb.lib().wholememory_init(0, wmb.LEVEL_ERROR))
wgth.create_group_communicator(1)
rng = np.random.default_rng(11)
for dt in (np.int32, np.int64):
    for nt, nn, universe in [(0, 0, 10), (5, 0, 10), (0, 7, 3), (700, 9000, 5000), (30000, 600000, 200000)]:
        t = rng.permutation(universe)[:nt].astype(dt)
        n = rng.integers(0, universe, nn).astype(dt)
        if nn > 4:
            n[:3] = -1
        ou, om = oracle.append_unique(t, n)
        u, m = gops.append_unique(torch.from_numpy(t).cuda(), torch.from_numpy(n).cuda(), True)
        assert np.array_equal(u.cpu().numpy(), ou) and np.array_equal(m.cpu().numpy(), om), (dt, nt, nn)
print("ROUTES_OK")
""" % ROOT
    env = dict(os.environ, WM_AU_TABLE_MAX=limit)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ROUTES_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_append_unique_reference_docstring_example(gpu_env):
    import torch
    import wholegraph_amd.torch.graph_ops as gops
    t = torch.tensor([3, 11, 2, 10], device="cuda")
    n = torch.tensor([4, 5, 2, 11, 6, 9, 10, 5], device="cuda")
    uniq, mapping = gops.append_unique(t, n, True)
    assert uniq.tolist() == [3, 11, 2, 10, 4, 5, 6, 9]          # a valid order of the reference's "may be" answer
    assert mapping.tolist() == [4, 5, 2, 1, 6, 7, 3, 5]
    assert torch.equal(uniq[mapping.long()], n)


@pytest.mark.parametrize("n_nodes,max_degree", [(1, 0), (1, 5), (200, 9), (5000, 300)])
def test_add_csr_self_loop_parity(gpu_env, n_nodes, max_degree):
    import torch
    import wholegraph_amd.torch.graph_ops as gops
    row_ptr, col = make_csr(n_nodes, max_degree, n_nodes + max_degree, np.int32)
    o_row, o_col = oracle.csr_add_self_loop(row_ptr, col)
    row, colo = gops.add_csr_self_loop(torch.from_numpy(row_ptr.astype(np.int32)).cuda(), torch.from_numpy(col).cuda())
    assert np.array_equal(row.cpu().numpy(), o_row) and np.array_equal(colo.cpu().numpy(), o_col)


def test_multilayer_sample(gpu_env):
    """3-hop sampling = chained one-hop + append_unique with the samplers' default (random) seeds: structural properties only
    (frontiers nest, ids are distinct, per-centre counts = min(degree, fan-out), every sampled edge exists). The same chain
    with fixed seeds is compared with the oracle bit for bit in tests/test_c5_flow_gpu.py."""
    import torch
    import wholegraph_amd.torch as wgth
    row_ptr, col = _graph(np.int64)
    wrow, wcol = _wm_csr(gpu_env, "chunked", "cuda", row_ptr, col)
    g = wgth.GraphStructure()
    g.set_csr_graph(wrow, wcol)
    seeds = torch.from_numpy(np.random.default_rng(1).permutation(3000)[:256]).cuda()
    target_gids, edge_indice, csr_row_ptr, csr_col_ind = g.multilayer_sample_without_replacement(seeds, [5, 4, 3])
    assert len(target_gids) == 4 and torch.equal(target_gids[3], seeds)
    for i in (2, 1, 0):
        inner, outer = target_gids[i + 1], target_gids[i]
        assert torch.equal(outer[:inner.numel()], inner)                    # targets stay in front
        assert outer.unique().numel() == outer.numel()
        off, colind = csr_row_ptr[i], csr_col_ind[i]
        assert off.numel() == inner.numel() + 1 and int(off[-1]) == colind.numel()
        fan = [5, 4, 3][2 - i]
        deg = torch.from_numpy(row_ptr[1:] - row_ptr[:-1]).cuda()[inner]
        assert torch.equal((off[1:] - off[:-1]).long(), torch.clamp(deg, max=fan))
        # every sampled edge (center -> outer[colind]) exists in the graph
        src = inner[edge_indice[i][1].long()].cpu().numpy()
        dst = outer[edge_indice[i][0].long()].cpu().numpy()
        for s, d in list(zip(src, dst))[:2000]:
            assert d in col[row_ptr[s]:row_ptr[s + 1]]
    wgth.destroy_wholememory_tensor(wrow)
    wgth.destroy_wholememory_tensor(wcol)


def test_append_unique_scan_hands_out_tiles_by_ticket(gpu_env, knobs):
    """The ranking scan of append_unique is a single-launch chained scan (graph.hip: chain_scan_kernel). Up to 8 tiles per CU
    it runs one tile per block; beyond that blocks loop over tiles and take their tile numbers from a ticket counter, so that
    a block never waits for a tile nobody has started (advisor, round 4). WM_SCAN_ITEMS=1 makes the tiles 256 values small:
    600 k neighbours are ~2400 tiles — past the one-tile-per-block bound on a 256-CU part — and the result must not change."""
    import torch
    import wholegraph_amd.torch.graph_ops as gops
    rng = np.random.default_rng(77)
    targets = rng.permutation(1 << 20)[:5000].astype(np.int32)
    neighbors = rng.integers(0, 1 << 20, 600_000).astype(np.int32)
    o_uniq, o_map = oracle.append_unique(targets, neighbors)
    for items in ("1", "4", None):
        if items is None:
            knobs.unset("WM_SCAN_ITEMS")
        else:
            knobs.set("WM_SCAN_ITEMS", items)
        uniq, mapping = gops.append_unique(torch.from_numpy(targets).cuda(), torch.from_numpy(neighbors).cuda(), True)
        assert np.array_equal(uniq.cpu().numpy(), o_uniq), items
        assert np.array_equal(mapping.cpu().numpy(), o_map), items
