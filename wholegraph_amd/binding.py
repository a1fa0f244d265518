"""ctypes binding of the wholegraph_amd C ABI (``include/wholememory/*.h``).

Plays the role of the reference's Cython module
``python/pylibwholegraph/pylibwholegraph/binding/wholememory_binding.pyx`` (error-code -> exception map
:255-277, handle wrappers, EmbeddingGatherForward/GradientApply :920-948, gather/scatter ops
:1897-1917): thin, no compute, opaque handles carried as ``c_void_p``.

The shared library is built in-tree by ``wholegraph_amd/csrc/Makefile`` (``__graft_entry__.build()``)
and MUST be present: there is no Python/CPU fallback for any op.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwholegraph.so")

# ---- enums (values are ABI; include/wholememory/*.h) -------------------------------------------
WHOLEMEMORY_SUCCESS = 0
NOT_SUPPORTED = 9
OUT_OF_MEMORY = 8
ERROR_NAMES = {
    0: "WHOLEMEMORY_SUCCESS", 1: "WHOLEMEMORY_UNKNOW_ERROR", 2: "WHOLEMEMORY_NOT_IMPLEMENTED",
    3: "WHOLEMEMORY_LOGIC_ERROR", 4: "WHOLEMEMORY_CUDA_ERROR", 5: "WHOLEMEMORY_COMMUNICATION_ERROR",
    6: "WHOLEMEMORY_INVALID_INPUT", 7: "WHOLEMEMORY_INVALID_VALUE", 8: "WHOLEMEMORY_OUT_OF_MEMORY",
    9: "WHOLEMEMORY_NOT_SUPPORTED", 10: "WHOLEMEMORY_SYSTEM_ERROR",
}
MT_NONE, MT_CONTINUOUS, MT_CHUNKED, MT_DISTRIBUTED, MT_HIERARCHY = range(5)
ML_NONE, ML_DEVICE, ML_HOST = range(3)
(DT_UNKNOWN, DT_FLOAT, DT_HALF, DT_DOUBLE, DT_BF16, DT_INT, DT_INT64, DT_INT16, DT_INT8, DT_COUNT) = range(10)
MA_NONE, MA_DEVICE, MA_HOST, MA_PINNED = range(4)
OPT_NONE, OPT_SGD, OPT_LAZY_ADAM, OPT_RMSPROP, OPT_ADAGRAD = range(5)
AT_NONE, AT_READONLY, AT_READWRITE = range(3)
LEVEL_FATAL, LEVEL_ERROR, LEVEL_WARN, LEVEL_INFO, LEVEL_DEBUG, LEVEL_TRACE = range(6)
MAX_TENSOR_DIM = 8
UNIQUE_ID_BYTES = 128


class WholeMemoryError(RuntimeError):
    """Raised for every non-zero wholememory_error_code_t (reference wholememory_binding.pyx:255-277
    raises a family of builtin exceptions; the mapping below keeps the same builtin bases)."""

    def __init__(self, code, what=""):
        self.code = code
        super().__init__("%s (%d)%s" % (ERROR_NAMES.get(code, "?"), code, (": " + what) if what else ""))


class _NotImpl(WholeMemoryError, NotImplementedError):
    pass


class _Value(WholeMemoryError, ValueError):
    pass


class _Memory(WholeMemoryError, MemoryError):
    pass


class _System(WholeMemoryError, SystemError):
    pass


_EXC = {2: _NotImpl, 6: _Value, 7: _Value, 8: _Memory, 9: _NotImpl, 10: _System}


def check(code, what=""):
    if code != WHOLEMEMORY_SUCCESS:
        raise _EXC.get(code, WholeMemoryError)(code, what)


# ---- structs ----------------------------------------------------------------------------------
class TensorDescription(C.Structure):
    _fields_ = [("sizes", C.c_int64 * MAX_TENSOR_DIM), ("strides", C.c_int64 * MAX_TENSOR_DIM),
                ("storage_offset", C.c_int64), ("dim", C.c_int), ("dtype", C.c_int)]


class ArrayDescription(C.Structure):
    _fields_ = [("size", C.c_int64), ("storage_offset", C.c_int64), ("dtype", C.c_int)]


class MatrixDescription(C.Structure):
    _fields_ = [("sizes", C.c_int64 * 2), ("stride", C.c_int64), ("storage_offset", C.c_int64), ("dtype", C.c_int)]


class GRef(C.Structure):
    _fields_ = [("pointer", C.c_void_p), ("rank_memory_offsets", C.c_void_p), ("world_size", C.c_int),
                ("stride", C.c_size_t), ("same_chunk", C.c_bool)]


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * UNIQUE_ID_BYTES)]


CREATE_CTX_FN = C.CFUNCTYPE(None, C.POINTER(C.c_void_p), C.c_void_p)
DESTROY_CTX_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
MALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.POINTER(TensorDescription), C.c_int, C.c_void_p, C.c_void_p)
FREE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)


class TempMemoryFns(C.Structure):
    _fields_ = [("create_memory_context_fn", CREATE_CTX_FN), ("destroy_memory_context_fn", DESTROY_CTX_FN),
                ("malloc_fn", MALLOC_FN), ("free_fn", FREE_FN), ("global_context", C.c_void_p)]


class OutputMemoryFns(C.Structure):
    _fields_ = [("malloc_fn", MALLOC_FN), ("free_fn", FREE_FN), ("global_context", C.c_void_p)]


class EnvFunc(C.Structure):
    _fields_ = [("temporary_fns", TempMemoryFns), ("output_fns", OutputMemoryFns)]


BARRIER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
ALLGATHER_HOST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_void_p,
                           C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_void_p)


class ExtCollectives(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("barrier", BARRIER_FN), ("allgather_host", ALLGATHER_HOST_FN),
                ("alltoallv_device", ALLTOALLV_FN)]


# ---- prototypes -------------------------------------------------------------------------------
_vp, _i, _i64, _sz, _f, _b = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float, C.c_bool
_P = C.POINTER

# name -> (restype, argtypes). Every symbol declared in include/wholememory/*.h appears here;
# tests/test_abi_symbols.py checks the list against the headers and against the built library.
PROTOTYPES = {
    # wholememory.h
    "wholememory_init": (_i, [C.c_uint, _i]),
    "wholememory_finalize": (_i, []),
    "wholememory_create_unique_id": (_i, [_P(UniqueId)]),
    "wholememory_create_communicator": (_i, [_P(_vp), UniqueId, _i, _i]),
    "wholememory_split_communicator": (_i, [_P(_vp), _vp, _i, _i]),
    "wholememory_destroy_communicator": (_i, [_vp]),
    "wholememory_communicator_support_type_location": (_i, [_vp, _i, _i]),
    "wholememory_communicator_get_rank": (_i, [_P(_i), _vp]),
    "wholememory_communicator_get_size": (_i, [_P(_i), _vp]),
    "wholememory_communicator_get_local_size": (_i, [_P(_i), _vp]),
    "wholememory_communicator_get_clique_info": (_i, [_vp, _vp]),
    "wholememory_communicator_is_bind_to_nvshmem": (_b, [_vp]),
    "wholememory_communicator_set_distributed_backend": (_i, [_vp, _i]),
    "wholememory_communicator_get_distributed_backend": (_i, [_vp]),
    "wholememory_communicator_barrier": (_i, [_vp]),
    "wholememory_is_intranode_communicator": (_b, [_vp]),
    "wholememory_is_intra_mnnvl_communicator": (_b, [_vp]),
    "wholememory_is_build_with_nvshmem": (_b, []),
    "wholememory_malloc": (_i, [_P(_vp), _sz, _vp, _i, _i, _sz, _P(_sz)]),
    "wholememory_free": (_i, [_vp]),
    "wholememory_get_communicator": (_i, [_P(_vp), _vp]),
    "wholememory_get_local_communicator": (_i, [_P(_vp), _vp]),
    "wholememory_get_cross_communicator": (_i, [_P(_vp), _vp]),
    "wholememory_get_memory_type": (_i, [_vp]),
    "wholememory_get_memory_location": (_i, [_vp]),
    "wholememory_get_distributed_backend": (_i, [_vp]),
    "wholememory_get_total_size": (_sz, [_vp]),
    "wholememory_get_data_granularity": (_sz, [_vp]),
    "wholememory_get_local_memory": (_i, [_P(_vp), _P(_sz), _P(_sz), _vp]),
    "wholememory_get_local_size": (_i, [_P(_sz), _vp]),
    "wholememory_get_local_offset": (_i, [_P(_sz), _vp]),
    "wholememory_get_rank_memory": (_i, [_P(_vp), _P(_sz), _P(_sz), _i, _vp]),
    "wholememory_equal_entry_partition_plan": (_i, [_P(_sz), _sz, _i]),
    "wholememory_get_global_pointer": (_i, [_P(_vp), _vp]),
    "wholememory_get_global_reference": (_i, [_P(GRef), _vp]),
    "wholememory_get_rank_partition_sizes": (_i, [_P(_sz), _vp]),
    "wholememory_get_rank_partition_offsets": (_i, [_P(_sz), _vp]),
    "fork_get_device_count": (_i, []),
    "wholememory_load_from_file": (_i, [_vp, _sz, _sz, _sz, _P(C.c_char_p), _i, _i]),
    "wholememory_store_to_file": (_i, [_vp, _sz, _sz, _sz, C.c_char_p]),
    # global_reference.h
    "wholememory_create_continuous_global_reference": (GRef, [_vp]),
    # tensor_description.h
    "wholememory_dtype_get_element_size": (_sz, [_i]),
    "wholememory_dtype_is_floating_number": (_b, [_i]),
    "wholememory_dtype_is_integer_number": (_b, [_i]),
    "wholememory_create_array_desc": (ArrayDescription, [_i64, _i64, _i]),
    "wholememory_create_matrix_desc": (MatrixDescription, [_P(_i64), _i64, _i64, _i]),
    "wholememory_initialize_tensor_desc": (None, [_P(TensorDescription)]),
    "wholememory_copy_array_desc_to_matrix": (None, [_P(MatrixDescription), _P(ArrayDescription)]),
    "wholememory_copy_array_desc_to_tensor": (None, [_P(TensorDescription), _P(ArrayDescription)]),
    "wholememory_copy_matrix_desc_to_tensor": (None, [_P(TensorDescription), _P(MatrixDescription)]),
    "wholememory_convert_tensor_desc_to_array": (_b, [_P(ArrayDescription), _P(TensorDescription)]),
    "wholememory_convert_tensor_desc_to_matrix": (_b, [_P(MatrixDescription), _P(TensorDescription)]),
    "wholememory_get_memory_element_count_from_array": (_i64, [_P(ArrayDescription)]),
    "wholememory_get_memory_size_from_array": (_i64, [_P(ArrayDescription)]),
    "wholememory_get_memory_element_count_from_matrix": (_i64, [_P(MatrixDescription)]),
    "wholememory_get_memory_size_from_matrix": (_i64, [_P(MatrixDescription)]),
    "wholememory_get_memory_element_count_from_tensor": (_i64, [_P(TensorDescription)]),
    "wholememory_get_memory_size_from_tensor": (_i64, [_P(TensorDescription)]),
    "wholememory_squeeze_tensor": (_b, [_P(TensorDescription), _i]),
    "wholememory_unsqueeze_tensor": (_b, [_P(TensorDescription), _i]),
    # wholememory_tensor.h
    "wholememory_create_tensor": (_i, [_P(_vp), _P(TensorDescription), _vp, _i, _i, _P(_sz)]),
    "wholememory_destroy_tensor": (_i, [_vp]),
    "wholememory_make_tensor_from_pointer": (_i, [_P(_vp), _vp, _P(TensorDescription)]),
    "wholememory_make_tensor_from_handle": (_i, [_P(_vp), _vp, _P(TensorDescription)]),
    "wholememory_tensor_has_handle": (_b, [_vp]),
    "wholememory_tensor_get_memory_handle": (_vp, [_vp]),
    "wholememory_tensor_get_tensor_description": (_P(TensorDescription), [_vp]),
    "wholememory_tensor_get_global_reference": (_i, [_vp, _P(GRef)]),
    "wholememory_tensor_map_local_tensor": (_i, [_vp, _P(_vp)]),
    "wholememory_tensor_get_data_pointer": (_vp, [_vp]),
    "wholememory_tensor_get_entry_offsets": (_i, [_P(_sz), _vp]),
    "wholememory_tensor_get_entry_partition_sizes": (_i, [_P(_sz), _vp]),
    "wholememory_tensor_get_local_entry_count": (_i, [_P(_sz), _vp]),
    "wholememory_tensor_get_local_entry_start": (_i, [_P(_sz), _vp]),
    "wholememory_tensor_get_subtensor": (_i, [_vp, _P(_i64), _P(_i64), _P(_vp)]),
    "wholememory_tensor_get_root": (_vp, [_vp]),
    "get_wholememory_tensor_count": (_i64, []),
    # env_func_ptrs.h
    "get_device_prop": (_vp, [_i]),
    "wholememory_get_default_env_func": (_P(EnvFunc), []),
    "wholememory_get_cached_env_func": (_P(EnvFunc), []),
    "wholememory_drop_cached_env_func_cache": (None, []),
    # wholememory_op.h
    "wholememory_gather": (_i, [_vp, _vp, _vp, _P(EnvFunc), _vp, _i]),
    "wholememory_scatter": (_i, [_vp, _vp, _vp, _P(EnvFunc), _vp, _i]),
    # embedding.h
    "wholememory_create_embedding_optimizer": (_i, [_P(_vp), _i]),
    "wholememory_optimizer_set_parameter": (_i, [_vp, C.c_char_p, _vp]),
    "wholememory_destroy_embedding_optimizer": (None, [_vp]),
    "wholememory_create_embedding_cache_policy": (_i, [_P(_vp), _vp, _i, _i, _i, _f]),
    "wholememory_destroy_embedding_cache_policy": (_i, [_vp]),
    "wholememory_create_embedding": (_i, [_P(_vp), _P(TensorDescription), _vp, _i, _i, _vp, _P(_sz), _i, _i]),
    "wholememory_destroy_embedding": (_i, [_vp]),
    "wholememory_embedding_get_embedding_tensor": (_vp, [_vp]),
    "wholememory_embedding_set_optimizer": (_i, [_vp, _vp]),
    "wholememory_embedding_gather": (_i, [_vp, _vp, _vp, _b, _P(EnvFunc), _i64]),
    "wholememory_embedding_gather_gradient_apply": (_i, [_vp, _vp, _vp, _b, _f, _P(EnvFunc), _i64]),
    "wholememory_embedding_get_optimizer_state_names": (_P(C.c_char_p), [_vp]),
    "wholememory_embedding_get_optimizer_state": (_vp, [_vp, C.c_char_p]),
    "wholememory_embedding_writeback_cache": (_i, [_vp, _i64]),
    "wholememory_embedding_drop_all_cache": (_i, [_vp, _i64]),
    "wholememory_env_test_op": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _P(EnvFunc), _vp]),
    # wholegraph_op.h
    "wholegraph_csr_unweighted_sample_without_replacement": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, C.c_ulonglong,
                                                                 _P(EnvFunc), _vp]),
    "wholegraph_csr_weighted_sample_without_replacement": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp,
                                                               C.c_ulonglong, _P(EnvFunc), _vp]),
    "generate_random_positive_int_cpu": (_i, [_i64, _i64, _vp]),
    "generate_exponential_distribution_negative_float_cpu": (_i, [_i64, _i64, _vp]),
    # graph_op.h
    "graph_append_unique": (_i, [_vp, _vp, _vp, _vp, _P(EnvFunc), _vp]),
    "csr_add_self_loop": (_i, [_vp, _vp, _vp, _vp, _vp]),
    # wholegraph_amd_ext.h
    "wholememory_create_communicator_ext": (_i, [_P(_vp), _i, _i, _P(ExtCollectives)]),
    "wholememory_ext_communicator_transport": (_i, [_vp, _P(C.c_char_p), _P(_i)]),
    "wholememory_ext_bucket_ids": (_i, [_vp, _i, _i64, _vp, _i, _vp, _vp, _vp, _P(EnvFunc), _vp]),
    "wholememory_ext_bucket_ids_folded": (_i, [_vp, _i, _i64, _vp, _i, _i, _vp, _vp, _vp, _P(EnvFunc), _vp]),
    "wholememory_ext_dedup_apply": (_i, [_vp, _i, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i, _P(_f), _f, _vp,
                                        _vp, _P(_i64), _P(EnvFunc), _vp]),
    "wholememory_ext_round_robin_map": (_i, [_vp, _vp, _i, _i64, _i64, _i, _i, _vp]),
    "wholememory_ext_embedding_cache_info": (_i, [_vp, _P(_i64), _P(_i64), _P(_i64), _P(_i64), _P(_i64)]),
    "wholememory_ext_backend_name": (C.c_char_p, []),
    "wholememory_ext_last_rows_kernel": (C.c_char_p, []),
    "wholememory_ext_sample_append_unique": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_ulonglong, C.c_void_p,
                                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "wholememory_ext_set_async_completion": (_i, [_i]),
    "wholememory_ext_reload_knobs": (_i, []),
    "wholememory_ext_probe_memory": (_i, [_vp, C.c_size_t, _i, _i, _P(_f)]),
    "wholememory_ext_host_sorted_gathers": (_i64, []),
    "wholememory_ext_split_sorts": (_i64, []),
    "wholememory_ext_hot_split_sorts": (_i64, []),
    "wholememory_ext_dense_fold_last": (_i64, []),
    "wholememory_ext_distributed_gather_launches": (_i64, []),
    "wholememory_ext_distributed_scatter_launches": (_i64, []),
    "wholememory_ext_gradient_exchange_launches": (_i64, []),
    "wholememory_ext_alltoallv_bytes": (_i64, []),
    "wholememory_ext_combined_gradient_calls": (_i64, []),
    "wholememory_ext_set_malloc_probe": (_i, [C.c_char_p]),
    "wholememory_ext_get_malloc_probe": (_i, [C.c_char_p, C.c_size_t]),
    "wholememory_ext_handle_was_probed": (_i, [_vp]),
    "wholememory_ext_multilayer_sample": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "wm_testing_install_backend": (_i, [_vp]),
}

_lib = None


def lib():
    """Load libwholegraph.so (after torch, so both share torch's HIP runtime in-process)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "wholegraph_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C wholegraph_amd/csrc`). There is no fallback implementation." % LIB_PATH)
        try:
            import torch  # noqa: F401  (loads torch's libamdhip64 first when torch is installed)
        except Exception:  # pragma: no cover
            pass
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


# ---- small helpers ----------------------------------------------------------------------------
def reload_knobs():
    """The library reads its WM_* / WG_* environment knobs once; call this after changing one mid-process."""
    check(lib().wholememory_ext_reload_knobs())


def make_tensor_desc(sizes, dtype, strides=None, storage_offset=0):
    d = TensorDescription()
    lib().wholememory_initialize_tensor_desc(C.byref(d))
    d.dim = len(sizes)
    if strides is None:
        strides = [1] * len(sizes)
        for i in range(len(sizes) - 2, -1, -1):
            strides[i] = strides[i + 1] * sizes[i + 1]
    for i, (s, st) in enumerate(zip(sizes, strides)):
        d.sizes[i] = s
        d.strides[i] = st
    d.storage_offset = storage_offset
    d.dtype = dtype
    return d


def size_t_array(values):
    return (C.c_size_t * len(values))(*values) if values is not None else None
