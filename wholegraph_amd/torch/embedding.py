"""WholeMemoryEmbedding / WholeMemoryOptimizer — the user-facing embedding table.

Mirrors reference ``python/pylibwholegraph/pylibwholegraph/torch/embedding.py`` (:33-67 optimizer,
:214-238 autograd function, :241-377 embedding, :380-524 factories, :537-577 module + optimizer
factories): same names, keyword arguments, autograd contract (backward only stashes (indice, grad);
``WholeMemoryOptimizer.step(lr)`` applies them through the C ABI and then barriers).
"""
import ctypes as C
from typing import List, Union

import torch

from .. import binding as wmb
from .comm import WholeMemoryCommunicator, get_global_communicator, get_local_node_communicator, \
    get_local_device_communicator
from .tensor import WholeMemoryTensor
from .utils import (
    torch_dtype_to_wholememory_dtype,
    str_to_wmb_wholememory_memory_type,
    str_to_wmb_wholememory_location,
    str_to_wmb_wholememory_access_type,
    str_to_wmb_wholememory_optimizer_type,
    get_file_size,
)
from .wholegraph_env import wrap_torch_tensor, get_wholegraph_env_fns, get_stream, op_device


class WholeMemoryOptimizer(object):
    """Sparse optimizer shared by any number of embeddings; create with create_wholememory_optimizer."""

    def __init__(self, global_comm: WholeMemoryCommunicator):
        super().__init__()
        self.wmb_opt = C.c_void_p()
        self.embeddings = []
        self.global_comm = global_comm

    def add_embedding(self, wm_embedding):
        assert isinstance(wm_embedding, WholeMemoryEmbedding)
        if wm_embedding.wmb_optimizer is not None:
            raise ValueError("optimizer can only be set once.")
        wm_embedding.wmb_optimizer = self.wmb_opt
        wm_embedding.dummy_input.requires_grad_(True)
        wmb.check(wmb.lib().wholememory_embedding_set_optimizer(wm_embedding.wmb_embedding, self.wmb_opt))
        self.embeddings.append(wm_embedding)

    def step(self, lr: float):
        for wm_embedding in self.embeddings:
            if wm_embedding.need_apply:
                wm_embedding.apply_gradients(lr)
        self.global_comm.barrier()


class WholeMemoryCachePolicy(object):
    def __init__(self, wmb_cache_policy):
        super().__init__()
        self.wmb_cache_policy = wmb_cache_policy


def create_wholememory_cache_policy(cache_comm, *, memory_type="chunked", memory_location="cuda",
                                    access_type="readonly", ratio=0.5):
    p = C.c_void_p()
    wmb.check(wmb.lib().wholememory_create_embedding_cache_policy(
        C.byref(p), cache_comm.wmb_comm, str_to_wmb_wholememory_memory_type(memory_type),
        str_to_wmb_wholememory_location(memory_location), str_to_wmb_wholememory_access_type(access_type), ratio))
    return WholeMemoryCachePolicy(p)


def destroy_wholememory_cache_policy(cache_policy):
    wmb.check(wmb.lib().wholememory_destroy_embedding_cache_policy(cache_policy.wmb_cache_policy))
    cache_policy.wmb_cache_policy = None


# builtin cache flavour -> (communicator the cache lives on, memory type of the cache when the caller names none;
# None = "same as the embedding", a fixed string = always that type)
_BUILTIN_CACHES = {
    "all_devices": (get_global_communicator, None),
    "local_node": (get_local_node_communicator, "chunked"),
    "local_device": (get_local_device_communicator, "continuous"),
}
_MEMORY_TYPES = frozenset(("continuous", "chunked", "distributed", "hierarchy"))
_LOCATIONS = frozenset(("cpu", "cuda"))


def create_builtin_cache_policy(builtin_cache_type, embedding_memory_type, embedding_memory_location, access_type,
                                cache_ratio, *, cache_memory_type="", cache_memory_location=""):
    """Cache policy by name ("none" | "local_device" | "local_node" | "all_devices"), arguments and defaults of the
    reference (embedding.py:139-209). What stands behind the policy here is a device row cache."""
    for what, value, allowed in (("embedding_memory_type", embedding_memory_type, _MEMORY_TYPES),
                                 ("embedding_memory_location", embedding_memory_location, _LOCATIONS),
                                 ("cache_memory_location", cache_memory_location, _LOCATIONS | {""})):
        if value not in allowed:
            raise ValueError("%s=%r is not valid (one of %s)" % (what, value, sorted(allowed)))
    if builtin_cache_type == "none":
        return None
    if builtin_cache_type not in _BUILTIN_CACHES:
        raise ValueError("builtin_cache_type=%r not supported, should be none, local_device, local_node or all_devices"
                         % (builtin_cache_type,))
    communicator_of, default_type = _BUILTIN_CACHES[builtin_cache_type]
    if builtin_cache_type == "local_device":
        memory_type = default_type                       # one device: a flat block, whatever was asked
    else:
        memory_type = cache_memory_type or default_type or embedding_memory_type
    return create_wholememory_cache_policy(communicator_of(), memory_type=memory_type,
                                           memory_location=cache_memory_location or "cuda", access_type=access_type,
                                           ratio=cache_ratio)


class EmbeddingLookupFn(torch.autograd.Function):
    """Lookup with a sparse backward: forward gathers rows, backward does NOT build a dense gradient — it hands (ids,
    row gradients) to the embedding, whose optimizer applies them at WholeMemoryOptimizer.step(lr). `dummy_input` only
    exists so that autograd has a differentiable input to call backward for (contract of the reference, embedding.py:214-238)."""

    @staticmethod
    def forward(ctx, indice, dummy_input, wm_embedding, is_training=False, force_dtype=None):
        rows = wm_embedding.gather(indice, is_training=is_training, force_dtype=force_dtype)
        ctx.target = wm_embedding if (is_training and wm_embedding.need_grad()) else None
        # the gradient handed back for dummy_input must look like the tensor the caller passed (its own anchor, maybe of
        # another shape or on another device), not like the embedding's
        ctx.dummy_like = (tuple(dummy_input.shape), dummy_input.dtype, dummy_input.device)
        if ctx.target is not None:
            ctx.save_for_backward(indice)
        return rows

    @staticmethod
    def backward(ctx, grad_outputs):
        target, ctx.target = ctx.target, None
        if target is None:   # a forward outside training (or without optimizer) recorded nothing: no gradient for anybody
            return None, None, None, None, None
        (indice,) = ctx.saved_tensors
        target.add_gradients(indice, grad_outputs)
        # one entry per forward input; only dummy_input is differentiable and its gradient carries no information
        shape, dtype, device = ctx.dummy_like
        return None, torch.zeros(shape, dtype=dtype, device=device), None, None, None


class WholeMemoryEmbedding(object):
    """Handle of a wholememory_embedding_t plus the Python-side training state: the gradient batches waiting for the next
    optimizer step and the autograd anchor. Create with create_embedding / create_embedding_from_filelist."""

    def __init__(self, wmb_embedding, wmb_cache_policy: Union[WholeMemoryCachePolicy, None]):
        # native objects
        self.wmb_embedding = wmb_embedding          # c_void_p (wholememory_embedding_t)
        self.wmb_cache_policy = wmb_cache_policy
        self.wmb_optimizer = None                   # set by WholeMemoryOptimizer.add_embedding
        # lazily wrapped views of native tensors
        self.embedding_tensor = None
        self.optimizer_states = {}
        # training state
        self.adjust_cache = wmb_cache_policy is not None
        self.need_apply = False
        self.sparse_indices, self.sparse_grads = [], []
        self.dummy_input = torch.nn.Parameter(torch.zeros(1), requires_grad=False)

    def dim(self):
        return self.get_embedding_tensor().dim()

    @property
    def shape(self):
        return self.get_embedding_tensor().shape

    def set_adjust_cache(self, adjust_cache: bool):
        self.adjust_cache = adjust_cache if self.wmb_cache_policy is not None else False

    def need_grad(self):
        return self.wmb_embedding is not None

    def gather(self, indice: torch.Tensor, *, is_training: bool = False, force_dtype: Union[torch.dtype, None] = None,
               out: Union[torch.Tensor, None] = None):
        """reference embedding.py:280-311. `out` (extension, not in the reference): gather into a caller-owned
        [n, dim] cuda tensor instead of allocating one (what the reference's C++ bench does)."""
        assert indice.dim() == 1
        emb = self.get_embedding_tensor()
        output_dtype = force_dtype if force_dtype is not None else emb.dtype
        need_grad = self.need_grad() and is_training
        if out is not None:
            assert out.dim() == 2 and out.shape[0] >= indice.shape[0] and out.shape[1] == emb.shape[1]
            output_tensor = out
        else:
            output_tensor = torch.empty([indice.shape[0], emb.shape[1]],
                                        device=op_device(), dtype=output_dtype,
                                        requires_grad=need_grad)
        if need_grad:
            self.need_apply = True
        wi, wo = wrap_torch_tensor(indice), wrap_torch_tensor(output_tensor)
        wmb.check(wmb.lib().wholememory_embedding_gather(self.wmb_embedding, wi.handle, wo.handle, self.adjust_cache,
                                                         get_wholegraph_env_fns(), get_stream()))
        return output_tensor

    def add_gradients(self, indice: torch.Tensor, grad_outputs: torch.Tensor):
        self.sparse_indices.append(indice)
        self.sparse_grads.append(grad_outputs)

    def apply_gradients(self, lr: float):
        """one wholememory_embedding_gather_gradient_apply over everything add_gradients collected since the last step"""
        def merged(parts):  # a single pending batch is used as it is: torch.cat would copy the whole matrix
            return parts[0] if len(parts) == 1 else torch.cat(parts)
        ids, grads = merged(self.sparse_indices), merged(self.sparse_grads).contiguous()
        self.sparse_indices, self.sparse_grads, self.need_apply = [], [], False
        wi, wg = wrap_torch_tensor(ids), wrap_torch_tensor(grads)
        wmb.check(wmb.lib().wholememory_embedding_gather_gradient_apply(
            self.wmb_embedding, wi.handle, wg.handle, self.adjust_cache, lr, get_wholegraph_env_fns(), get_stream()))

    def writeback_all_cache(self):
        wmb.check(wmb.lib().wholememory_embedding_writeback_cache(self.wmb_embedding, get_stream(False)))

    def drop_all_cache(self):
        wmb.check(wmb.lib().wholememory_embedding_drop_all_cache(self.wmb_embedding, get_stream(False)))

    def get_embedding_tensor(self):
        if self.embedding_tensor is None:
            self.embedding_tensor = WholeMemoryTensor(
                C.c_void_p(wmb.lib().wholememory_embedding_get_embedding_tensor(self.wmb_embedding)))
        return self.embedding_tensor

    def get_optimizer_state_names(self):
        names = wmb.lib().wholememory_embedding_get_optimizer_state_names(self.wmb_embedding)
        out, i = [], 0
        while names and names[i] is not None:
            out.append(names[i].decode())
            i += 1
        return out

    def get_optimizer_state(self, state_name):
        if state_name not in self.optimizer_states:
            p = wmb.lib().wholememory_embedding_get_optimizer_state(self.wmb_embedding, state_name.encode())
            if not p:
                raise ValueError("no optimizer state named %s" % state_name)
            self.optimizer_states[state_name] = WholeMemoryTensor(C.c_void_p(p))
        return self.optimizer_states[state_name]

    def save(self, file_prefix: str):
        self.get_embedding_tensor().to_file_prefix(file_prefix + "_embedding_tensor")
        for state_name in self.get_optimizer_state_names():
            self.get_optimizer_state(state_name).to_file_prefix(file_prefix + "_" + state_name)

    def load(self, file_prefix: str, *, ignore_embedding: bool = False, part_count: Union[int, None] = None):
        if ignore_embedding is False:
            self.get_embedding_tensor().from_file_prefix(file_prefix + "_embedding_tensor", part_count)
        for state_name in self.get_optimizer_state_names():
            self.get_optimizer_state(state_name).from_file_prefix(file_prefix + "_" + state_name, part_count)


def _reconcile_sharding(embedding_entry_partition, cache_policy, round_robin_size):
    """The three ways to say where rows live exclude each other: a cache policy decides for itself, an explicit per-rank
    partition wins over round-robin sharding. Returns (partition, round_robin_size) after dropping what is overridden, with
    the reference's notices."""
    if embedding_entry_partition is not None and cache_policy is not None:
        print("embedding_entry_partition is ignored because cache_policy is specified")
        embedding_entry_partition = None
    if embedding_entry_partition is not None and round_robin_size:
        print("round_robin_size is ignored because embedding_entry_partition is specified")
        round_robin_size = 0
    return embedding_entry_partition, round_robin_size


def create_embedding(comm: WholeMemoryCommunicator, memory_type: str, memory_location: str, dtype: torch.dtype,
                     sizes: List[int], *, cache_policy: Union[WholeMemoryCachePolicy, None] = None,
                     embedding_entry_partition: Union[List[int], None] = None, random_init: bool = False,
                     gather_sms: int = -1, round_robin_size: int = 0, placement_probe: Union[str, int, None] = None):
    """Collective over `comm`: a [rows, dim] embedding table (reference embedding.py:380-459, same arguments).
    embedding_entry_partition: rows per rank (default: equal shares); round_robin_size: rows per round-robin block when the
    table is filled from files in round-robin order; gather_sms: cap on the workgroups of the gather kernels (-1: default).
    One extension: placement_probe = "auto" | 2 ... 8 | "off" | None (None: what WM_MALLOC_PROBE says, off by default) — for
    tables that will be WRITTEN at random (trained, scattered into): the device shard is chosen among a few candidate
    allocations by a short write probe, because the speed of random row writes follows where the allocation sits in HBM
    (about 20 %, for the table's lifetime). Holds the candidates transiently (at most a quarter of the
    free memory)."""
    rows, dim = sizes                                            # exactly two dimensions
    previous = None
    if placement_probe is not None:
        buf = C.create_string_buffer(16)
        wmb.check(wmb.lib().wholememory_ext_get_malloc_probe(buf, len(buf)))
        previous = buf.value                     # what the application had set before (or b"env"): put back afterwards
        wmb.check(wmb.lib().wholememory_ext_set_malloc_probe(str(placement_probe).encode()))
    try:
        return _create_embedding(comm, memory_type, memory_location, dtype, rows, dim, cache_policy, embedding_entry_partition,
                                 random_init, gather_sms, round_robin_size)
    finally:
        if previous is not None:                 # (never raises: an error here would mask the one from the creation)
            wmb.lib().wholememory_ext_set_malloc_probe(previous)


def _create_embedding(comm, memory_type, memory_location, dtype, rows, dim, cache_policy, embedding_entry_partition, random_init,
                      gather_sms, round_robin_size):
    partition, round_robin_size = _reconcile_sharding(embedding_entry_partition, cache_policy, round_robin_size)
    description = wmb.make_tensor_desc([rows, dim], torch_dtype_to_wholememory_dtype(dtype), [dim, 1], 0)
    handle = C.c_void_p()
    wmb.check(wmb.lib().wholememory_create_embedding(
        C.byref(handle), C.byref(description), comm.wmb_comm, str_to_wmb_wholememory_memory_type(memory_type),
        str_to_wmb_wholememory_location(memory_location), cache_policy.wmb_cache_policy if cache_policy else None,
        wmb.size_t_array(partition), gather_sms, round_robin_size))
    embedding = WholeMemoryEmbedding(handle, cache_policy)
    if random_init:
        shard, _ = embedding.get_embedding_tensor().get_local_tensor()
        if shard.numel():
            torch.nn.init.xavier_uniform_(shard)
    comm.barrier()
    return embedding


def create_embedding_from_filelist(comm, memory_type, memory_location, filelist, dtype, last_dim_size, *,
                                   cache_policy=None, embedding_entry_partition=None, gather_sms=-1,
                                   round_robin_size=0):
    """An embedding sized after, and filled from, raw row-major files of `last_dim_size`-wide rows (one file or a list)."""
    files = [filelist] if isinstance(filelist, str) else list(filelist)
    if last_dim_size <= 0:
        raise ValueError("last_dim_size must be positive")
    partition, round_robin_size = _reconcile_sharding(embedding_entry_partition, None, round_robin_size)
    row_bytes = torch.empty(0, dtype=dtype).element_size() * last_dim_size
    sizes = [get_file_size(name) for name in files]   # per list entry: a file named twice is loaded twice
    for name, nbytes in zip(files, sizes):
        if nbytes % row_bytes:
            raise ValueError("File %s size is %d not mutlple of %d" % (name, nbytes, row_bytes))
    embedding = create_embedding(comm, memory_type, memory_location, dtype, [sum(sizes) // row_bytes, last_dim_size],
                                 cache_policy=cache_policy, embedding_entry_partition=partition, gather_sms=gather_sms,
                                 round_robin_size=round_robin_size)
    embedding.get_embedding_tensor().from_filelist(files, round_robin_size)
    return embedding


def destroy_embedding(wm_embedding: WholeMemoryEmbedding):
    wmb.check(wmb.lib().wholememory_destroy_embedding(wm_embedding.wmb_embedding))
    wm_embedding.wmb_embedding = None
    wm_embedding.embedding_tensor = None
    wm_embedding.optimizer_states = dict()


class WholeMemoryEmbeddingModule(torch.nn.Module):
    """torch.nn.Module wrapper of WholeMemoryEmbedding."""

    def __init__(self, wm_embedding: WholeMemoryEmbedding):
        super().__init__()
        self.wm_embedding = wm_embedding
        self.embedding_gather_fn = EmbeddingLookupFn.apply

    def forward(self, indice: torch.Tensor, force_dtype: Union[torch.dtype, None] = None):
        return self.embedding_gather_fn(indice, self.wm_embedding.dummy_input, self.wm_embedding, self.training,
                                        force_dtype)


def create_wholememory_optimizer(embeddings, optimizer_type: str, param_dict: dict):
    """One sparse optimizer ("sgd" | "adam" | "adagrad" | "rmsprop") for an embedding or a list of them; param_dict holds the
    hyper-parameters by the reference's names (weight_decay, epsilon, beta1, beta2, alpha, adam_w). One extension:
    "grad_fold": "ordered" | "tree" | "default" — the order in which the fp32 sum of a run of duplicate gradient rows is taken
    ("ordered" = the reference's receive order, bit-identical results, the default for fp32 tables; "tree" = fixed-shape
    partial sums, deterministic, equal within rounding and much faster when a few ids carry most of the batch)."""
    optimizer = WholeMemoryOptimizer(get_global_communicator())
    kind = str_to_wmb_wholememory_optimizer_type(optimizer_type)
    wmb.check(wmb.lib().wholememory_create_embedding_optimizer(C.byref(optimizer.wmb_opt), kind))
    for name, value in (param_dict or {}).items():
        if name == "grad_fold" and isinstance(value, str):
            value = {"default": -1.0, "ordered": 0.0, "tree": 1.0}[value]
        boxed = C.c_float(float(value))
        wmb.check(wmb.lib().wholememory_optimizer_set_parameter(optimizer.wmb_opt, name.encode(), C.byref(boxed)))
    targets = [embeddings] if isinstance(embeddings, WholeMemoryEmbedding) else list(embeddings)
    for target in targets:
        optimizer.add_embedding(target)
    return optimizer


def destroy_wholememory_optimizer(optimizer: WholeMemoryOptimizer):
    wmb.lib().wholememory_destroy_embedding_optimizer(optimizer.wmb_opt)
    optimizer.wmb_opt = None
