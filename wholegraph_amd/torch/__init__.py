"""wholegraph_amd.torch — the ``pylibwholegraph.torch`` surface of the embedding path on MI355X.

Same names and signatures as reference ``python/pylibwholegraph/pylibwholegraph/torch/__init__.py:14-78`` for
communicators, initialisation, WholeMemory tensors, embeddings / optimizers / cache policies, the gather / scatter
functors, neighbour sampling and GraphStructure. The GNN model zoo, data loaders, launch helpers and option parsers of
the reference are outside this build's scope.
"""
from . import comm, embedding, graph_ops, graph_structure, initialize, tensor, utils, wholegraph_ops, wholememory_ops

_PUBLIC = {
    comm: ("WholeMemoryCommunicator create_group_communicator destroy_communicator get_global_communicator "
           "get_local_node_communicator get_local_device_communicator split_communicator get_local_mnnvl_communicator"),
    embedding: ("WholeMemoryOptimizer create_wholememory_optimizer destroy_wholememory_optimizer WholeMemoryCachePolicy "
                "create_builtin_cache_policy create_wholememory_cache_policy destroy_wholememory_cache_policy "
                "WholeMemoryEmbedding create_embedding create_embedding_from_filelist destroy_embedding "
                "WholeMemoryEmbeddingModule"),
    initialize: "init init_torch_env init_torch_env_and_create_wm_comm finalize",
    tensor: ("WholeMemoryTensor create_wholememory_tensor create_wholememory_tensor_from_filelist "
             "destroy_wholememory_tensor"),
    utils: "get_part_file_name get_part_file_list wholememory_dtype_to_torch_dtype torch_dtype_to_wholememory_dtype",
    wholememory_ops: "wholememory_gather_forward_functor wholememory_scatter_functor",
    graph_structure: "GraphStructure",
}
__all__ = ["graph_ops", "wholegraph_ops"]
for _module, _names in _PUBLIC.items():
    for _name in _names.split():
        globals()[_name] = getattr(_module, _name)
        __all__.append(_name)
del _module, _names, _name
