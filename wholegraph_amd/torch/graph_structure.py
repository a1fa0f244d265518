"""One relation of a graph held in WholeMemory as CSR arrays, and the samplers that walk it.

Public surface = ``pylibwholegraph.torch.graph_structure.GraphStructure`` (reference
``python/pylibwholegraph/pylibwholegraph/torch/graph_structure.py:21-228``: same method names, arguments and return
layouts, so cuGraph-DGL / PyG call sites keep working). The bodies are this package's own: the CSR pair is validated once
in a helper, attributes live in one registry keyed by kind, and the multi-hop sampler is a loop over a per-hop record
instead of four parallel lists. Extensions: ``multilayer_sample_without_replacement(..., random_seeds=[...])`` fixes the
per-hop sampler seeds (the reference draws them from the global RNG), which is what lets tests/test_c5_flow_gpu.py replay
the whole chain on the CPU oracle; an unweighted multi-hop sample on a CSR mapped into this rank runs as ONE library call
with ONE host round trip for all hops (``wholegraph_ops.multilayer_sample``: upper-bound-sized buffers, counts kept on the
device, views trimmed at the end; ``WM_MULTILAYER_CHAIN=0`` switches it off), and where that does not apply a hop still runs as
one call (``wholegraph_ops.sample_append_unique``: sampler + append_unique with a single host round trip, same outputs).
"""
import os
from collections import namedtuple
from typing import List, Optional, Sequence, Union

import torch

from . import graph_ops, wholegraph_ops
from .tensor import WholeMemoryTensor

_INDEX_DTYPES = (torch.int32, torch.int64)
_Hop = namedtuple("_Hop", "targets edge_index row_ptr col_ind")


def _layer_lists(layers, node_ids):
    return ([hop.targets for hop in layers] + [node_ids], [hop.edge_index for hop in layers],
            [hop.row_ptr for hop in layers], [hop.col_ind for hop in layers])


def _chain_layers(chain, hops):
    layers = [None] * hops
    for depth, (offsets, widened, neighbour_pos, centre_lid, edge_index) in enumerate(chain):
        layers[hops - 1 - depth] = _Hop(widened, edge_index, offsets, neighbour_pos)
    return layers


class _DeferredSample(object):
    """handle of GraphStructure.multilayer_sample_begin"""

    def __init__(self, graph, node_ids, max_neighbors, random_seeds, pending):
        self._node_ids, self._hops, self._pending, self._lists = node_ids, len(max_neighbors), pending, None
        if pending is None:      # not queued as one chain: sampled now, hop by hop
            self._lists = graph._sample_hop_by_hop(node_ids, max_neighbors, None, random_seeds)
            self.padded_frontier = self._lists[0][0]
        else:
            self.padded_frontier = pending.padded_frontier

    def result(self):
        if self._lists is None:
            self._lists = _layer_lists(_chain_layers(self._pending.finish(), self._hops), self._node_ids)
        return self._lists


def _checked_csr(row_ptr: WholeMemoryTensor, col_ind: WholeMemoryTensor):
    """(node count, edge count) of a valid CSR pair; raises AssertionError like the reference on a malformed one"""
    problems = []
    if row_ptr.dim() != 1 or col_ind.dim() != 1:
        problems.append("csr_row_ptr and csr_col_ind must be 1-D")
    if row_ptr.dtype != torch.int64:
        problems.append("csr_row_ptr must be int64")
    if col_ind.dtype not in _INDEX_DTYPES:
        problems.append("csr_col_ind must be int32 or int64")
    if row_ptr.dim() == 1 and row_ptr.shape[0] < 2:
        problems.append("csr_row_ptr needs at least two entries")
    assert not problems, "; ".join(problems)
    return row_ptr.shape[0] - 1, col_ind.shape[0]


class GraphStructure(object):
    """CSR structure of one relation + per-node / per-edge attribute tensors (all WholeMemory tensors)."""

    def __init__(self):
        self.csr_row_ptr = self.csr_col_ind = None
        self.node_count = self.edge_count = 0
        self._attributes = {"node": {}, "edge": {}}

    # the reference exposes the two registries as plain dict attributes
    @property
    def node_attributes(self):
        return self._attributes["node"]

    @property
    def edge_attributes(self):
        return self._attributes["edge"]

    def set_csr_graph(self, csr_row_ptr: WholeMemoryTensor, csr_col_ind: WholeMemoryTensor):
        self.node_count, self.edge_count = _checked_csr(csr_row_ptr, csr_col_ind)
        self.csr_row_ptr, self.csr_col_ind = csr_row_ptr, csr_col_ind

    def _register(self, kind: str, name: str, tensor: WholeMemoryTensor, expected_rows: int):
        registry = self._attributes[kind]
        assert name not in registry, "%s attribute %r is already set" % (kind, name)
        assert tensor.shape[0] == expected_rows, "%s attribute %r has %d rows, the graph has %d %ss" % (
            kind, name, tensor.shape[0], expected_rows, kind)
        registry[name] = tensor

    def set_node_attribute(self, attr_name: str, attr_tensor: WholeMemoryTensor):
        self._register("node", attr_name, attr_tensor, self.node_count)

    def set_edge_attribute(self, attr_name: str, attr_tensor: WholeMemoryTensor):
        self._register("edge", attr_name, attr_tensor, self.edge_count)

    # ------------------------------------------------------------------------------------------- one hop
    def _one_hop(self, centers: torch.Tensor, fanout: int, weight_name: Optional[str], seed, want_lid: bool, want_eid: bool):
        csr = (self.csr_row_ptr.wmb_tensor, self.csr_col_ind.wmb_tensor)
        if weight_name is None:
            return wholegraph_ops.unweighted_sample_without_replacement(*csr, centers, fanout, seed, want_lid, want_eid)
        assert weight_name in self.edge_attributes, "no edge attribute named %r" % weight_name
        weights = self.edge_attributes[weight_name].wmb_tensor
        return wholegraph_ops.weighted_sample_without_replacement(*csr, weights, centers, fanout, seed, want_lid, want_eid)

    def unweighted_sample_without_replacement_one_hop(self, center_nodes_tensor: torch.Tensor, max_sample_count: int, *,
                                                      random_seed: Union[int, None] = None,
                                                      need_center_local_output: bool = False,
                                                      need_edge_output: bool = False):
        """-> csr_row_ptr, sampled_nodes[, center_node_local_id][, edge_index]"""
        return self._one_hop(center_nodes_tensor, max_sample_count, None, random_seed, need_center_local_output,
                             need_edge_output)

    def weighted_sample_without_replacement_one_hop(self, weight_name: str, center_nodes_tensor: torch.Tensor,
                                                    max_sample_count: int, *, random_seed: Union[int, None] = None,
                                                    need_center_local_output: bool = False,
                                                    need_edge_output: bool = False):
        """the same, neighbours drawn with probability proportional to the named edge attribute"""
        return self._one_hop(center_nodes_tensor, max_sample_count, weight_name, random_seed, need_center_local_output,
                             need_edge_output)

    # ------------------------------------------------------------------------------------------ several hops
    def multilayer_sample_without_replacement(self, node_ids: torch.Tensor, max_neighbors: List[int],
                                              weight_name: Union[str, None] = None, *,
                                              random_seeds: Optional[Sequence[int]] = None):
        """Sample len(max_neighbors) hops outwards from node_ids (max_neighbors[0] is the fan-out of the hop next to the
        seeds). Returns four lists indexed by layer, OUTERMOST layer first, as the reference does:
          target_gids  (hops + 1 entries; the last one is node_ids, entry i holds entry i + 1 followed by its new neighbours)
          edge_indice  [2, n_edges]: row 0 = position of the neighbour in target_gids[i], row 1 = position of its centre
          csr_row_ptr / csr_col_ind of the sampled block (col_ind = row 0 of edge_indice)"""
        hops = len(max_neighbors)
        if random_seeds is not None:
            assert len(random_seeds) == hops, "one seed per hop"
        layers = [None] * hops
        if weight_name is None and hops > 0 and os.environ.get("WM_MULTILAYER_CHAIN", "1") != "0":
            # the whole chain as one library call with a single host round trip (extension); None = not applicable to this
            # graph or these sizes: hop by hop below then
            chain = wholegraph_ops.multilayer_sample(self.csr_row_ptr.wmb_tensor, self.csr_col_ind.wmb_tensor, node_ids,
                                                     max_neighbors, random_seeds)
            if chain is not None:
                return _layer_lists(_chain_layers(chain, hops), node_ids)
        return self._sample_hop_by_hop(node_ids, max_neighbors, weight_name, random_seeds)

    def multilayer_sample_begin(self, node_ids: torch.Tensor, max_neighbors: List[int], *,
                                random_seeds: Optional[Sequence[int]] = None):
        """Extension: the unweighted multi-layer sample QUEUED, the host not waiting for it. Returns a handle with
          .padded_frontier   the outermost frontier (what target_gids[0] will be) at its upper-bound size, the entries behind the
                             sampled nodes set to -1 — hand it to WholeMemoryEmbedding.gather right away: negative ids are
                             skipped, so rows [0, n) of that gather's output are the features of target_gids[0];
          .result()          one stream synchronise, then exactly what multilayer_sample_without_replacement returns.
        The feature gather of a mini-batch then runs back to back with the sampling kernels instead of behind a host round
        trip. When the one-call chain does not apply (see wholegraph_ops.multilayer_sample_begin) the sample is taken here and
        now, hop by hop, and padded_frontier is target_gids[0] itself."""
        hops = len(max_neighbors)
        if random_seeds is not None:
            assert len(random_seeds) == hops, "one seed per hop"
        pending = None
        if hops > 0 and os.environ.get("WM_MULTILAYER_CHAIN", "1") != "0":
            pending = wholegraph_ops.multilayer_sample_begin(self.csr_row_ptr.wmb_tensor, self.csr_col_ind.wmb_tensor, node_ids,
                                                             max_neighbors, random_seeds)
        return _DeferredSample(self, node_ids, max_neighbors, random_seeds, pending)

    def _sample_hop_by_hop(self, node_ids, max_neighbors, weight_name, random_seeds):
        hops = len(max_neighbors)
        layers = [None] * hops
        frontier = node_ids
        for depth, fanout in enumerate(max_neighbors):          # depth 0 = next to the seeds = layer hops - 1
            seed = None if random_seeds is None else random_seeds[depth]
            fused = None
            if weight_name is None:
                # sampler + append_unique as one library call with one host round trip (extension); None = not applicable
                # to this graph (CSR not mapped into this rank, id dtypes differ ...): the two ops below then
                fused = wholegraph_ops.sample_append_unique(self.csr_row_ptr.wmb_tensor, self.csr_col_ind.wmb_tensor,
                                                            frontier, fanout, seed)
            if fused is not None:
                offsets, widened, neighbour_pos, centre_lid = fused
            else:
                offsets, neighbours, centre_lid = self._one_hop(frontier, fanout, weight_name, seed, True, False)
                widened, neighbour_pos = graph_ops.append_unique(frontier, neighbours, need_neighbor_raw_to_unique=True)
            layers[hops - 1 - depth] = _Hop(widened, torch.stack([neighbour_pos, centre_lid]), offsets, neighbour_pos)
            frontier = widened
        return _layer_lists(layers, node_ids)
