"""Functional gather / scatter wrappers. Mirrors reference
``python/pylibwholegraph/pylibwholegraph/torch/wholememory_ops.py:24-78``."""
import ctypes as C

import torch

from .. import binding as wmb
from .wholegraph_env import get_stream, get_wholegraph_env_fns, wrap_torch_tensor, op_device


def _raw(wholememory_tensor):
    return wholememory_tensor.wmb_tensor if hasattr(wholememory_tensor, "wmb_tensor") else wholememory_tensor


def wholememory_gather_forward_functor(wholememory_tensor, indices_tensor, requires_grad=False, torch_output_dtype=None):
    assert indices_tensor.dim() == 1
    assert indices_tensor.dtype == torch.int32 or indices_tensor.dtype == torch.int64
    from .tensor import WholeMemoryTensor
    wt = wholememory_tensor if isinstance(wholememory_tensor, WholeMemoryTensor) else WholeMemoryTensor(_raw(wholememory_tensor))
    if torch_output_dtype is None:
        torch_output_dtype = wt.dtype
    output_tensor = torch.empty([indices_tensor.shape[0], wt.shape[1]], device=op_device(), dtype=torch_output_dtype,
                                requires_grad=requires_grad)
    wi, wo = wrap_torch_tensor(indices_tensor), wrap_torch_tensor(output_tensor)
    wmb.check(wmb.lib().wholememory_gather(wt.wmb_tensor, wi.handle, wo.handle, get_wholegraph_env_fns(),
                                           C.c_void_p(get_stream()), -1))
    return output_tensor


def wholememory_scatter_functor(input_tensor, indices_tensor, wholememory_tensor):
    assert indices_tensor.dim() == 1
    assert indices_tensor.dtype == torch.int32 or indices_tensor.dtype == torch.int64
    wi, wt = wrap_torch_tensor(indices_tensor), wrap_torch_tensor(input_tensor)
    wmb.check(wmb.lib().wholememory_scatter(wt.handle, wi.handle, _raw(wholememory_tensor), get_wholegraph_env_fns(),
                                            C.c_void_p(get_stream()), -1))
