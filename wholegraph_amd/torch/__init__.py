"""wholegraph_amd.torch — drop-in for the comm / initialize / tensor / embedding / ops subset of
``pylibwholegraph.torch`` (reference ``python/pylibwholegraph/pylibwholegraph/torch/__init__.py:14-78``).
plus unweighted neighbour sampling / append_unique / add_csr_self_loop and GraphStructure. The GNN-model and
launcher helpers of the reference are outside this build's scope."""
from .comm import (
    WholeMemoryCommunicator,
    create_group_communicator,
    destroy_communicator,
    get_global_communicator,
    get_local_node_communicator,
    get_local_device_communicator,
    split_communicator,
    get_local_mnnvl_communicator,
)
from .embedding import (
    WholeMemoryOptimizer,
    create_wholememory_optimizer,
    destroy_wholememory_optimizer,
    WholeMemoryCachePolicy,
    create_builtin_cache_policy,
    create_wholememory_cache_policy,
    destroy_wholememory_cache_policy,
    WholeMemoryEmbedding,
    create_embedding,
    create_embedding_from_filelist,
    destroy_embedding,
    WholeMemoryEmbeddingModule,
)
from .initialize import init, init_torch_env, init_torch_env_and_create_wm_comm, finalize
from .tensor import (
    WholeMemoryTensor,
    create_wholememory_tensor,
    create_wholememory_tensor_from_filelist,
    destroy_wholememory_tensor,
)
from .utils import get_part_file_name, get_part_file_list
from .utils import wholememory_dtype_to_torch_dtype, torch_dtype_to_wholememory_dtype
from .wholememory_ops import wholememory_gather_forward_functor, wholememory_scatter_functor
from .graph_structure import GraphStructure
from . import graph_ops, wholegraph_ops
