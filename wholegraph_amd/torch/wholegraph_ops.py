"""Neighbour sampling on CSR graphs stored in WholeMemory.

Mirror of ``python/pylibwholegraph/pylibwholegraph/torch/wholegraph_ops.py:27-209`` over the C ABI of
``include/wholememory/wholegraph_op.h``. `wm_csr_*_tensor` arguments are `wholememory_tensor_t` handles
(``WholeMemoryTensor.wmb_tensor``), exactly what the reference functions take."""
import ctypes as C
import random
from typing import Union

import torch

from .. import binding as wmb
from .wholegraph_env import TorchMemoryContext, get_stream, get_wholegraph_env_fns, op_device, wrap_torch_tensor


def _handle(t):
    return t.wmb_tensor if hasattr(t, "wmb_tensor") else t


def _tensor_dim(h):
    return int(wmb.lib().wholememory_tensor_get_tensor_description(h).contents.dim)


def _sample_outputs(offset, dest, lid, egid, need_center_local_output, need_edge_output):
    if need_edge_output and need_center_local_output:
        return offset, dest.get_tensor(), lid.get_tensor(), egid.get_tensor()
    if need_center_local_output:
        return offset, dest.get_tensor(), lid.get_tensor()
    if need_edge_output:
        return offset, dest.get_tensor(), egid.get_tensor()
    return offset, dest.get_tensor()


def unweighted_sample_without_replacement(wm_csr_row_ptr_tensor, wm_csr_col_ptr_tensor, center_nodes_tensor: torch.Tensor,
                                          max_sample_count: int, random_seed: Union[int, None] = None,
                                          need_center_local_output: bool = False, need_edge_output: bool = False):
    """For every center node, min(degree, max_sample_count) distinct neighbours (all when max_sample_count <= 0).
    Returns (sample_offset int32 [n + 1], sampled node ids[, center local id int32][, edge id int64])."""
    row, col = _handle(wm_csr_row_ptr_tensor), _handle(wm_csr_col_ptr_tensor)
    assert _tensor_dim(row) == 1
    assert _tensor_dim(col) == 1
    assert center_nodes_tensor.dim() == 1
    if random_seed is None:
        random_seed = random.getrandbits(64)
    offset = torch.empty(center_nodes_tensor.shape[0] + 1, device=op_device(), dtype=torch.int)
    dest = TorchMemoryContext()
    lid = TorchMemoryContext() if need_center_local_output else None
    egid = TorchMemoryContext() if need_edge_output else None
    wc, wo = wrap_torch_tensor(center_nodes_tensor), wrap_torch_tensor(offset)
    wmb.check(wmb.lib().wholegraph_csr_unweighted_sample_without_replacement(
        row, col, wc.handle, int(max_sample_count), wo.handle, C.c_void_p(dest.get_c_context()),
        C.c_void_p(lid.get_c_context() if lid else 0), C.c_void_p(egid.get_c_context() if egid else 0),
        C.c_ulonglong(random_seed & 0xFFFFFFFFFFFFFFFF), get_wholegraph_env_fns(), C.c_void_p(get_stream())))
    return _sample_outputs(offset, dest, lid, egid, need_center_local_output, need_edge_output)


def sample_append_unique(wm_csr_row_ptr_tensor, wm_csr_col_ptr_tensor, center_nodes_tensor: torch.Tensor, max_sample_count: int,
                         random_seed: Union[int, None] = None):
    """Extension (include/wholememory/wholegraph_amd_ext.h): one hop = unweighted sampling + append_unique(center nodes,
    sampled neighbours) in ONE call with one host round trip. Returns (sample_offset int32 [n + 1], unique nodes, position of
    every sampled neighbour in `unique` int32, center local id int32) — or None when the library declines (CSR not mapped
    into this rank, dtypes differ, empty frontier, max_sample_count <= 0): run the two ops then."""
    row, col = _handle(wm_csr_row_ptr_tensor), _handle(wm_csr_col_ptr_tensor)
    assert center_nodes_tensor.dim() == 1
    if random_seed is None:
        random_seed = random.getrandbits(64)
    offset = torch.empty(center_nodes_tensor.shape[0] + 1, device=op_device(), dtype=torch.int)
    uniq, pos, lid = TorchMemoryContext(), TorchMemoryContext(), TorchMemoryContext()
    wc, wo = wrap_torch_tensor(center_nodes_tensor), wrap_torch_tensor(offset)
    rc = wmb.lib().wholememory_ext_sample_append_unique(
        row, col, wc.handle, int(max_sample_count), C.c_ulonglong(random_seed & 0xFFFFFFFFFFFFFFFF), wo.handle,
        C.c_void_p(uniq.get_c_context()), C.c_void_p(pos.get_c_context()), C.c_void_p(lid.get_c_context()),
        get_wholegraph_env_fns(), C.c_void_p(get_stream()))
    if rc == wmb.NOT_SUPPORTED:
        return None
    wmb.check(rc)
    return offset, uniq.get_tensor(), pos.get_tensor(), lid.get_tensor()


_pinned_pool = {}     # hops -> idle pinned count buffers (one per chain in flight, handed back by finish())
_event_pool = []


def _multilayer_budget_bytes():
    """upper-bound buffers of a chain that the one-call route may allocate (WM_MULTILAYER_MAX_BYTES, default 8 GiB): beyond it
    the chain is declined and the caller samples hop by hop with exactly sized outputs"""
    import os
    try:
        return int(os.environ.get("WM_MULTILAYER_MAX_BYTES", str(8 << 30)))
    except ValueError:
        return 8 << 30


def _compact(t):
    """a trimmed view keeps its whole upper-bound buffer alive: copy out of big, mostly empty ones"""
    room = t.untyped_storage().nbytes()
    return t.clone() if room > (16 << 20) and t.numel() * t.element_size() * 2 < room else t


class PendingMultilayerSample:
    """A multi-hop sample whose kernels are queued and whose counts the host has not read yet (multilayer_sample_begin).
    `padded_frontier` is the outermost frontier at its full upper-bound size, the entries behind the sampled nodes set to -1:
    it can be handed to a gather right away (negative ids are skipped). `finish()` synchronises the stream once, trims every
    output and returns what multilayer_sample returns."""

    def __init__(self, hops, n0, offsets, uniques, edges, counts, stream):
        self._hops, self._n0, self._offsets, self._uniques, self._edges = hops, n0, offsets, uniques, edges
        self._counts, self._result = counts, None
        self.padded_frontier = uniques[-1]
        # finish() waits for THIS chain, not for whatever the caller queues behind it (the feature gather on padded_frontier
        # keeps running while the host trims the outputs)
        # (a chain queued while the stream is being captured into a graph has no event of its own: after the replay the
        # caller's stream is waited for instead)
        self._stream, self._done = stream, None
        if not torch.cuda.is_current_stream_capturing():
            self._done = _event_pool.pop() if _event_pool else torch.cuda.Event()
            self._done.record(stream)

    def finish(self):
        if self._result is not None:
            return self._result
        if self._done is not None:
            self._done.synchronize()      # the one host round trip of the whole chain
            _event_pool.append(self._done)
            self._done = None
        else:
            self._stream.synchronize()
        got = self._counts.tolist()
        _pinned_pool.setdefault(self._hops, []).append(self._counts)
        self._counts = None
        out, n_c = [], self._n0
        for h in range(self._hops):
            n_samples, n_new = got[2 * h], got[2 * h + 1]
            edge = _compact(self._edges[h][:, :n_samples])
            # (the outermost frontier stays a view of the padded array a gather may still be reading)
            uniq = self._uniques[h][:n_c + n_new] if h == self._hops - 1 else _compact(self._uniques[h][:n_c + n_new])
            out.append((_compact(self._offsets[h][:n_c + 1]), uniq, edge[0], edge[1], edge))
            n_c += n_new
        self.n_frontier = n_c
        self._result = out
        return out


def multilayer_sample_begin(wm_csr_row_ptr_tensor, wm_csr_col_ptr_tensor, seed_nodes_tensor: torch.Tensor, max_sample_counts,
                            random_seeds=None):
    """Queues every hop of an unweighted multi-layer sample (wholememory_ext_multilayer_sample: one library call, counts kept
    on the device between hops, no host round trip) and returns a PendingMultilayerSample WITHOUT waiting — or None when the
    library declines (CSR not mapped into this rank, dtypes differ, empty seeds, upper bounds beyond append_unique's hash-table
    route or beyond the memory budget, allocation failure): run hop by hop then."""
    row, col = _handle(wm_csr_row_ptr_tensor), _handle(wm_csr_col_ptr_tensor)
    assert seed_nodes_tensor.dim() == 1
    hops = len(max_sample_counts)
    n0 = seed_nodes_tensor.shape[0]
    if hops == 0 or n0 == 0 or any(int(m) <= 0 for m in max_sample_counts):
        return None
    if random_seeds is None:
        random_seeds = [random.getrandbits(64) for _ in range(hops)]
    cap_c, cap_s = [n0], []
    for m in max_sample_counts:
        cap_s.append(cap_c[-1] * int(m))
        cap_c.append(cap_c[-1] + cap_s[-1])
    if cap_c[-1] >= (1 << 31) - 1:
        return None
    idt = seed_nodes_tensor.dtype
    # outputs + the library's scratch (ids, hash table of 2 slots per key, positions ...): ~ 10 words per sampled neighbour
    need = sum(4 * (cap_c[h] + 1) + idt.itemsize * cap_c[h + 1] + 8 * cap_s[h] + 40 * (cap_c[h] + cap_s[h]) for h in range(hops))
    if need > _multilayer_budget_bytes():
        return None
    fan = (C.c_int * hops)(*[int(m) for m in max_sample_counts])
    ws = wrap_torch_tensor(seed_nodes_tensor)
    # ask first (no buffers yet); anything but SUCCESS is a decline
    if wmb.lib().wholememory_ext_multilayer_sample(row, col, ws.handle, hops, fan, None, None, None, None, None, None, None,
                                                   None) != wmb.WHOLEMEMORY_SUCCESS:
        return None
    dev = op_device()
    try:
        offsets = [torch.empty(cap_c[h] + 1, device=dev, dtype=torch.int) for h in range(hops)]
        uniques = [torch.empty(cap_c[h + 1], device=dev, dtype=idt) for h in range(hops)]
        edges = [torch.empty((2, max(cap_s[h], 1)), device=dev, dtype=torch.int) for h in range(hops)]   # row 0 positions, row 1 centre ids
    except torch.OutOfMemoryError:
        return None
    pool = _pinned_pool.setdefault(hops, [])
    counts = pool.pop() if pool else torch.zeros(2 * hops, dtype=torch.int32).pin_memory()
    rng = (C.c_ulonglong * hops)(*[int(s) & 0xFFFFFFFFFFFFFFFF for s in random_seeds])
    ptrs = lambda ts: (C.c_void_p * hops)(*[t.data_ptr() for t in ts])
    try:
        rc = wmb.lib().wholememory_ext_multilayer_sample(
            row, col, ws.handle, hops, fan, rng, ptrs(offsets), ptrs(uniques), ptrs([e[0] for e in edges]),
            ptrs([e[1] for e in edges]), C.c_void_p(counts.data_ptr()), get_wholegraph_env_fns(), C.c_void_p(get_stream()))
    except torch.OutOfMemoryError:      # the library's scratch comes from torch's allocator through the env functions
        rc = wmb.NOT_SUPPORTED
    if rc != wmb.WHOLEMEMORY_SUCCESS:
        pool.append(counts)
        if rc in (wmb.NOT_SUPPORTED, wmb.OUT_OF_MEMORY):
            return None
        wmb.check(rc)
    return PendingMultilayerSample(hops, n0, offsets, uniques, edges, counts, torch.cuda.current_stream())


def multilayer_sample(wm_csr_row_ptr_tensor, wm_csr_col_ptr_tensor, seed_nodes_tensor: torch.Tensor, max_sample_counts,
                      random_seeds=None):
    """Extension (wholememory_ext_multilayer_sample): every hop of an unweighted multi-layer sample in ONE library call with no
    host round trip inside — buffers sized for their upper bounds, counts kept on the device between hops, ONE stream
    synchronise here at the end. Returns a list with one (sample_offset, unique, neighbor_pos, center_lid, edge_index) tuple per
    hop, hop 0 next to the seeds, each tensor trimmed to its size and equal to what `sample_append_unique` returns hop by hop
    with the same seeds — or None when the library declines (see multilayer_sample_begin): run hop by hop then."""
    pending = multilayer_sample_begin(wm_csr_row_ptr_tensor, wm_csr_col_ptr_tensor, seed_nodes_tensor, max_sample_counts,
                                      random_seeds)
    return None if pending is None else pending.finish()


def weighted_sample_without_replacement(wm_csr_row_ptr_tensor, wm_csr_col_ptr_tensor, wm_csr_weight_ptr_tensor,
                                        center_nodes_tensor: torch.Tensor, max_sample_count: int,
                                        random_seed: Union[int, None] = None, need_center_local_output: bool = False,
                                        need_edge_output: bool = False):
    """A-Res weighted sampling: neighbours are kept with probability proportional to their edge weight
    (wm_csr_weight_ptr_tensor: float32/float64, one per edge). max_sample_count <= 8192."""
    row, col, wgt = _handle(wm_csr_row_ptr_tensor), _handle(wm_csr_col_ptr_tensor), _handle(wm_csr_weight_ptr_tensor)
    assert center_nodes_tensor.dim() == 1
    if random_seed is None:
        random_seed = random.getrandbits(64)
    offset = torch.empty(center_nodes_tensor.shape[0] + 1, device=op_device(), dtype=torch.int)
    dest = TorchMemoryContext()
    lid = TorchMemoryContext() if need_center_local_output else None
    egid = TorchMemoryContext() if need_edge_output else None
    wc, wo = wrap_torch_tensor(center_nodes_tensor), wrap_torch_tensor(offset)
    wmb.check(wmb.lib().wholegraph_csr_weighted_sample_without_replacement(
        row, col, wgt, wc.handle, int(max_sample_count), wo.handle, C.c_void_p(dest.get_c_context()),
        C.c_void_p(lid.get_c_context() if lid else 0), C.c_void_p(egid.get_c_context() if egid else 0),
        C.c_ulonglong(random_seed & 0xFFFFFFFFFFFFFFFF), get_wholegraph_env_fns(), C.c_void_p(get_stream())))
    return _sample_outputs(offset, dest, lid, egid, need_center_local_output, need_edge_output)


def generate_random_positive_int_cpu(random_seed, sub_sequence, output_random_value_count):
    output = torch.empty((output_random_value_count,), dtype=torch.int)
    w = wrap_torch_tensor(output)
    wmb.check(wmb.lib().generate_random_positive_int_cpu(int(random_seed), int(sub_sequence), w.handle))
    return output


def generate_exponential_distribution_negative_float_cpu(random_seed: int, sub_sequence: int,
                                                         output_random_value_count: int):
    output = torch.empty((output_random_value_count,), dtype=torch.float)
    w = wrap_torch_tensor(output)
    wmb.check(wmb.lib().generate_exponential_distribution_negative_float_cpu(int(random_seed), int(sub_sequence),
                                                                             w.handle))
    return output
