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


def _wm_array(comm, mt, arr):
    import torch
    import wholegraph_amd.torch as wgth
    t = wgth.create_wholememory_tensor(comm, mt, "cuda", [arr.shape[0]], torch.from_numpy(arr).dtype, [1])
    local, _ = t.get_local_tensor()
    local.copy_(torch.from_numpy(arr))
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
