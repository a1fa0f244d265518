"""GraphStructure — one relation of a graph in CSR form held in WholeMemory, with one-hop and multi-layer
sampling. Mirror of ``python/pylibwholegraph/pylibwholegraph/torch/graph_structure.py:21-228``."""
from typing import List, Union

import torch

from . import graph_ops, wholegraph_ops
from .tensor import WholeMemoryTensor


class GraphStructure(object):
    def __init__(self):
        self.node_count = 0
        self.edge_count = 0
        self.csr_row_ptr = None
        self.csr_col_ind = None
        self.node_attributes = {}
        self.edge_attributes = {}

    def set_csr_graph(self, csr_row_ptr: WholeMemoryTensor, csr_col_ind: WholeMemoryTensor):
        assert csr_row_ptr.dim() == 1
        assert csr_row_ptr.dtype == torch.int64
        assert csr_row_ptr.shape[0] > 1
        assert csr_col_ind.dim() == 1
        assert csr_col_ind.dtype in (torch.int32, torch.int64)
        self.node_count = csr_row_ptr.shape[0] - 1
        self.edge_count = csr_col_ind.shape[0]
        self.csr_row_ptr = csr_row_ptr
        self.csr_col_ind = csr_col_ind

    def set_node_attribute(self, attr_name: str, attr_tensor: WholeMemoryTensor):
        assert attr_name not in self.node_attributes
        assert attr_tensor.shape[0] == self.node_count
        self.node_attributes[attr_name] = attr_tensor

    def set_edge_attribute(self, attr_name: str, attr_tensor: WholeMemoryTensor):
        assert attr_name not in self.edge_attributes
        assert attr_tensor.shape[0] == self.edge_count
        self.edge_attributes[attr_name] = attr_tensor

    def unweighted_sample_without_replacement_one_hop(self, center_nodes_tensor: torch.Tensor, max_sample_count: int, *,
                                                      random_seed: Union[int, None] = None,
                                                      need_center_local_output: bool = False,
                                                      need_edge_output: bool = False):
        """-> csr_row_ptr, sampled_nodes[, center_node_local_id][, edge_index]"""
        return wholegraph_ops.unweighted_sample_without_replacement(
            self.csr_row_ptr.wmb_tensor, self.csr_col_ind.wmb_tensor, center_nodes_tensor, max_sample_count,
            random_seed, need_center_local_output, need_edge_output)

    def weighted_sample_without_replacement_one_hop(self, weight_name: str, center_nodes_tensor: torch.Tensor,
                                                    max_sample_count: int, *, random_seed: Union[int, None] = None,
                                                    need_center_local_output: bool = False,
                                                    need_edge_output: bool = False):
        assert weight_name in self.edge_attributes
        return wholegraph_ops.weighted_sample_without_replacement(
            self.csr_row_ptr.wmb_tensor, self.csr_col_ind.wmb_tensor, self.edge_attributes[weight_name].wmb_tensor,
            center_nodes_tensor, max_sample_count, random_seed, need_center_local_output, need_edge_output)

    def multilayer_sample_without_replacement(self, node_ids: torch.Tensor, max_neighbors: List[int],
                                              weight_name: Union[str, None] = None):
        """Sample len(max_neighbors) hops outwards from node_ids.
        -> target_gids (hops + 1 id lists, outermost first), edge_indice, csr_row_ptr, csr_col_ind per hop."""
        hops = len(max_neighbors)
        edge_indice, csr_row_ptr, csr_col_ind = [None] * hops, [None] * hops, [None] * hops
        target_gids = [None] * (hops + 1)
        target_gids[hops] = node_ids
        for i in range(hops - 1, -1, -1):
            fanout = max_neighbors[hops - i - 1]
            if weight_name is None:
                offsets, neighbors, src_lids = self.unweighted_sample_without_replacement_one_hop(
                    target_gids[i + 1], fanout, need_center_local_output=True)
            else:
                offsets, neighbors, src_lids = self.weighted_sample_without_replacement_one_hop(
                    weight_name, target_gids[i + 1], fanout, need_center_local_output=True)
            unique_gids, raw_to_unique = graph_ops.append_unique(target_gids[i + 1], neighbors,
                                                                 need_neighbor_raw_to_unique=True)
            csr_row_ptr[i] = offsets
            csr_col_ind[i] = raw_to_unique
            edge_indice[i] = torch.stack([raw_to_unique, src_lids])
            target_gids[i] = unique_gids
        return target_gids, edge_indice, csr_row_ptr, csr_col_ind
