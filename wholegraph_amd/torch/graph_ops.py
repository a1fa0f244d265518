"""append_unique / add_csr_self_loop — mirror of ``python/pylibwholegraph/pylibwholegraph/torch/graph_ops.py:24-99``
over ``include/wholememory/graph_op.h``."""
import ctypes as C

import torch

from .. import binding as wmb
from .wholegraph_env import TorchMemoryContext, get_stream, get_wholegraph_env_fns, op_device, wrap_torch_tensor


def append_unique(target_node_tensor: torch.Tensor, neighbor_node_tensor: torch.Tensor,
                  need_neighbor_raw_to_unique: bool = False):
    """unique(target ++ neighbor) with the targets kept first and unchanged; the new ids follow in first-occurrence
    order (the reference leaves that order unspecified). Optionally the position of every neighbour in the result."""
    assert target_node_tensor.dim() == 1
    assert neighbor_node_tensor.dim() == 1
    assert target_node_tensor.is_cuda
    assert neighbor_node_tensor.is_cuda
    ctx = TorchMemoryContext()
    mapping = None
    if need_neighbor_raw_to_unique:
        mapping = torch.empty(neighbor_node_tensor.shape[0], device=op_device(), dtype=torch.int)
    wt, wn, wm = wrap_torch_tensor(target_node_tensor), wrap_torch_tensor(neighbor_node_tensor), None
    if mapping is not None:
        wm = wrap_torch_tensor(mapping)
    wmb.check(wmb.lib().graph_append_unique(wt.handle, wn.handle, C.c_void_p(ctx.get_c_context()),
                                            wm.handle if wm is not None else None, get_wholegraph_env_fns(),
                                            C.c_void_p(get_stream())))
    if need_neighbor_raw_to_unique:
        return ctx.get_tensor(), mapping
    return ctx.get_tensor()


def add_csr_self_loop(csr_row_ptr_tensor: torch.Tensor, csr_col_ptr_tensor: torch.Tensor):
    """CSR (int32) with every node's own id inserted in front of its neighbours; existing loops are not checked."""
    assert csr_row_ptr_tensor.dim() == 1
    assert csr_col_ptr_tensor.dim() == 1
    assert csr_row_ptr_tensor.is_cuda
    assert csr_col_ptr_tensor.is_cuda
    out_row = torch.empty((csr_row_ptr_tensor.shape[0],), device=op_device(), dtype=csr_row_ptr_tensor.dtype)
    out_col = torch.empty((csr_col_ptr_tensor.shape[0] + csr_row_ptr_tensor.shape[0] - 1,), device=op_device(),
                          dtype=csr_col_ptr_tensor.dtype)
    w = [wrap_torch_tensor(t) for t in (csr_row_ptr_tensor, csr_col_ptr_tensor, out_row, out_col)]
    wmb.check(wmb.lib().csr_add_self_loop(w[0].handle, w[1].handle, w[2].handle, w[3].handle, C.c_void_p(get_stream())))
    return out_row, out_col
