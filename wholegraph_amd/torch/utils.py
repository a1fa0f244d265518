"""dtype / string-enum helpers. Mirrors the helper set of reference
``python/pylibwholegraph/pylibwholegraph/torch/utils.py:25-193`` (same function names, same accepted
strings, same error type) on top of the ctypes binding."""
import os

import torch

from .. import binding as wmb

_T2W = {
    torch.float: wmb.DT_FLOAT, torch.half: wmb.DT_HALF, torch.double: wmb.DT_DOUBLE, torch.bfloat16: wmb.DT_BF16,
    torch.int: wmb.DT_INT, torch.int64: wmb.DT_INT64, torch.int16: wmb.DT_INT16, torch.int8: wmb.DT_INT8,
}
_W2T = {v: k for k, v in _T2W.items()}


def torch_dtype_to_wholememory_dtype(torch_dtype):
    try:
        return _T2W[torch_dtype]
    except KeyError:
        raise ValueError("torch_dtype: %s not supported" % (torch_dtype,))


def wholememory_dtype_to_torch_dtype(wm_dtype):
    try:
        return _W2T[int(wm_dtype)]
    except KeyError:
        raise ValueError("WholeMemoryMemory: %s not supported" % (int(wm_dtype),))


def get_file_size(filename):
    if not os.path.isfile(filename):
        raise ValueError("File %s not found or not file" % (filename,))
    if not os.access(filename, os.R_OK):
        raise ValueError("File %s not readable" % (filename,))
    return os.path.getsize(filename)


def _lookup(table, key, what, allowed):
    if key in table:
        return table[key]
    raise ValueError("WholeMemory %s %s not supported, should be (%s)" % (what, key, allowed))


def str_to_wmb_wholememory_memory_type(s):
    return _lookup({"continuous": wmb.MT_CONTINUOUS, "chunked": wmb.MT_CHUNKED, "distributed": wmb.MT_DISTRIBUTED,
                    "hierarchy": wmb.MT_HIERARCHY}, s, "type", "continuous, chunked, distributed, hierarchy")


def str_to_wmb_wholememory_location(s):
    return _lookup({"cuda": wmb.ML_DEVICE, "cpu": wmb.ML_HOST}, s, "location", "cuda, cpu")


def str_to_wmb_wholememory_log_level(s):
    return _lookup({"error": wmb.LEVEL_ERROR, "warn": wmb.LEVEL_WARN, "info": wmb.LEVEL_INFO,
                    "debug": wmb.LEVEL_DEBUG, "trace": wmb.LEVEL_TRACE}, s, "log level",
                   "error, warn, info, debug, trace")


def str_to_wmb_wholememory_access_type(s):
    return _lookup({"readonly": wmb.AT_READONLY, "ro": wmb.AT_READONLY, "readwrite": wmb.AT_READWRITE,
                    "rw": wmb.AT_READWRITE}, s, "access", "readonly, ro, readwrite, rw")


def str_to_wmb_wholememory_optimizer_type(s):
    return _lookup({"sgd": wmb.OPT_SGD, "adam": wmb.OPT_LAZY_ADAM, "adagrad": wmb.OPT_ADAGRAD,
                    "rmsprop": wmb.OPT_RMSPROP}, s, "optimizer", "sgd, adam, adagrad, rmsprop")


def str_to_wmb_wholememory_distributed_backend_type(s):
    return _lookup({"nccl": 1, "nvshmem": 2}, s, "str_wmb_distributed_backend", "nccl, nvshmem")


def wholememory_distributed_backend_type_to_str(v):
    if v == 1:
        return "nccl"
    if v == 2:
        return "nvshmem"
    raise ValueError("WholeMemory distributed_backend  not supported, should be (DbNCCL, DbNVSHMEM)")


def get_part_file_name(prefix, part_id, part_count):
    return "%s_part_%d_of_%d" % (prefix, part_id, part_count)


def get_part_file_list(prefix, part_count):
    return [get_part_file_name(prefix, i, part_count) for i in range(part_count)]
