"""WholeMemoryTensor — the Python face of a wholememory_tensor_t.

Mirrors reference ``python/pylibwholegraph/pylibwholegraph/torch/tensor.py:33-308`` (same methods,
arguments and return shapes). Where the reference hands out torch views through DLPack capsules made by
its Cython module, this build aliases the raw pointers returned by the C ABI (``__cuda_array_interface__``
for HBM, ``torch.frombuffer`` for host memory).
"""
import ctypes as C

import torch

from .. import binding as wmb
from .utils import (
    torch_dtype_to_wholememory_dtype,
    wholememory_dtype_to_torch_dtype,
    get_file_size,
    str_to_wmb_wholememory_memory_type,
    str_to_wmb_wholememory_location,
    get_part_file_name,
    get_part_file_list,
)
from .comm import WholeMemoryCommunicator
from .wholegraph_env import wrap_torch_tensor, get_wholegraph_env_fns, get_stream, torch_tensor_from_pointer, \
    op_device


class WholeMemoryTensor(object):
    def __init__(self, wmb_tensor, owns=False):
        self.wmb_tensor = wmb_tensor  # c_void_p (wholememory_tensor_t)
        self._owns = owns

    # -- description -------------------------------------------------------------------------
    def _desc(self):
        return wmb.lib().wholememory_tensor_get_tensor_description(self.wmb_tensor).contents

    def _layout(self):
        """(dtype, shape, strides, storage_offset): fixed for the life of the tensor, read from the C side once"""
        lay = getattr(self, "_cached_layout", None)
        if lay is None:
            d = self._desc()
            n = d.dim
            lay = (wholememory_dtype_to_torch_dtype(d.dtype), tuple(int(d.sizes[i]) for i in range(n)),
                   tuple(int(d.strides[i]) for i in range(n)), int(d.storage_offset))
            self._cached_layout = lay
        return lay

    @property
    def dtype(self):
        return self._layout()[0]

    def dim(self):
        return len(self._layout()[1])

    @property
    def shape(self):
        return self._layout()[1]

    def stride(self):
        return self._layout()[2]

    def storage_offset(self):
        return self._layout()[3]

    def _handle(self):
        return C.c_void_p(wmb.lib().wholememory_tensor_get_memory_handle(self.wmb_tensor))

    def get_comm(self):
        comm = C.c_void_p()
        wmb.check(wmb.lib().wholememory_get_communicator(C.byref(comm), self._handle()))
        return WholeMemoryCommunicator(comm)

    # -- ops -----------------------------------------------------------------------------------
    def gather(self, indice, *, force_dtype=None):
        assert indice.dim() == 1
        out_dtype = force_dtype if force_dtype is not None else self.dtype
        shape = [indice.shape[0]] + ([self.shape[1]] if self.dim() == 2 else [])
        output = torch.empty(shape, device=op_device(), dtype=out_dtype, requires_grad=False)
        wi, wo = wrap_torch_tensor(indice), wrap_torch_tensor(output)
        wmb.check(wmb.lib().wholememory_gather(self.wmb_tensor, wi.handle, wo.handle, get_wholegraph_env_fns(),
                                               C.c_void_p(get_stream()), -1))
        return output

    def scatter(self, input_tensor, indice):
        assert indice.dim() == 1
        assert input_tensor.dim() == self.dim()
        assert indice.shape[0] == input_tensor.shape[0]
        if self.dim() == 2:
            assert input_tensor.shape[1] == self.shape[1]
        wi, wt = wrap_torch_tensor(indice), wrap_torch_tensor(input_tensor)
        wmb.check(wmb.lib().wholememory_scatter(wt.handle, wi.handle, self.wmb_tensor, get_wholegraph_env_fns(),
                                                C.c_void_p(get_stream()), -1))

    def get_sub_tensor(self, starts, ends):
        """starts/ends per dim; -1 in ends = to the last element."""
        n = self.dim()
        s = (C.c_int64 * n)(*starts)
        e = (C.c_int64 * n)(*ends)
        sub = C.c_void_p()
        wmb.check(wmb.lib().wholememory_tensor_get_subtensor(self.wmb_tensor, s, e, C.byref(sub)))
        return WholeMemoryTensor(sub)

    # -- views ---------------------------------------------------------------------------------
    def _location_is_device(self):
        return wmb.lib().wholememory_get_memory_location(self._handle()) == wmb.ML_DEVICE

    def _view(self, ptr, rows, host_view):
        d = self._desc()
        shape = [rows] + [int(d.sizes[i]) for i in range(1, d.dim)]
        strides = [int(d.strides[i]) for i in range(d.dim)]
        device_mem = self._location_is_device()
        if host_view and device_mem:
            raise ValueError("host view of device-located WholeMemory is not available")
        if not host_view and not device_mem:
            # host-located memory is registered with HIP: the same address is valid on the device
            return torch_tensor_from_pointer(ptr, shape, self.dtype, strides, True, owner=self)
        return torch_tensor_from_pointer(ptr, shape, self.dtype, strides, device_mem, owner=self)

    def get_local_tensor(self, host_view=False):
        """(torch view of this rank's rows, first row index). reference tensor.py:107-121."""
        L = wmb.lib()
        local = C.c_void_p()
        wmb.check(L.wholememory_tensor_map_local_tensor(self.wmb_tensor, C.byref(local)))
        try:
            ld = L.wholememory_tensor_get_tensor_description(local).contents
            rows = int(ld.sizes[0])
            ptr = L.wholememory_tensor_get_data_pointer(local)
        finally:
            L.wholememory_destroy_tensor(local)
        start = C.c_size_t()
        wmb.check(L.wholememory_tensor_get_local_entry_start(C.byref(start), self.wmb_tensor))
        return self._view(ptr, rows, host_view), int(start.value)

    def get_global_tensor(self, host_view=False):
        """(torch view of the whole tensor, 0): CONTINUOUS (or host CHUNKED) only. reference tensor.py:123-137."""
        ptr = wmb.lib().wholememory_tensor_get_data_pointer(self.wmb_tensor)
        mt = wmb.lib().wholememory_get_memory_type(self._handle())
        if not ptr and mt == wmb.MT_CHUNKED and not self._location_is_device():
            gp = C.c_void_p()
            wmb.check(wmb.lib().wholememory_get_global_pointer(C.byref(gp), self._handle()))
            ptr = gp.value + self.storage_offset() * torch.tensor([], dtype=self.dtype).element_size()
        if not ptr:
            raise ValueError("global tensor is only available for continuous (or host chunked) WholeMemory")
        return self._view(ptr, self.shape[0], host_view), 0

    def get_all_chunked_tensor(self, host_view=False):
        """([one view per rank], [first row per rank]) for mapped types. reference tensor.py:139-153."""
        L = wmb.lib()
        comm = self.get_comm()
        W = comm.get_size()
        offs = (C.c_size_t * (W + 1))()
        wmb.check(L.wholememory_tensor_get_entry_offsets(offs, self.wmb_tensor))
        es = torch.tensor([], dtype=self.dtype).element_size()
        views = []
        for r in range(W):
            p, sz, off = C.c_void_p(), C.c_size_t(), C.c_size_t()
            wmb.check(L.wholememory_get_rank_memory(C.byref(p), C.byref(sz), C.byref(off), r, self._handle()))
            rows = min(int(offs[r + 1]), self.shape[0]) - min(int(offs[r]), self.shape[0])
            views.append(self._view(p.value + self.storage_offset() * es, max(rows, 0), host_view))
        return views, [int(offs[r]) for r in range(W)]

    # -- files ---------------------------------------------------------------------------------
    def from_filelist(self, filelist, round_robin_size=0):
        if isinstance(filelist, str):
            filelist = [filelist]
        d = self._desc()
        es = torch.tensor([], dtype=self.dtype).element_size()
        mem_entry = es * (int(d.strides[0]) if d.dim == 2 else 1)
        file_entry = es * (int(d.sizes[1]) if d.dim == 2 else 1)
        mem_off = es * int(d.storage_offset)
        names = (C.c_char_p * len(filelist))(*[f.encode() for f in filelist])
        wmb.check(wmb.lib().wholememory_load_from_file(self._handle(), mem_off, mem_entry, file_entry, names,
                                                       len(filelist), round_robin_size))

    def from_file_prefix(self, file_prefix, part_count=None):
        if part_count is None:
            part_count = self.get_comm().get_size()
        self.from_filelist(get_part_file_list(file_prefix, part_count))

    def local_to_file(self, filename):
        d = self._desc()
        es = torch.tensor([], dtype=self.dtype).element_size()
        mem_entry = es * (int(d.strides[0]) if d.dim == 2 else 1)
        file_entry = es * (int(d.sizes[1]) if d.dim == 2 else 1)
        wmb.check(wmb.lib().wholememory_store_to_file(self._handle(), es * int(d.storage_offset), mem_entry, file_entry,
                                                      filename.encode()))

    def to_file_prefix(self, file_prefix):
        c = self.get_comm()
        self.local_to_file(get_part_file_name(file_prefix, c.get_rank(), c.get_size()))


def create_wholememory_tensor(comm, memory_type, memory_location, sizes, dtype, strides,
                              tensor_entry_partition=None):
    """Collective. 1-D or 2-D; strides None = dense. reference tensor.py:200-243."""
    dim = len(sizes)
    if dim < 1 or dim > 2:
        raise ValueError("Only dim 1 or 2 is supported now.")
    if strides is None:
        strides = [1] * dim
        strides[0] = sizes[1] if dim == 2 else 1
    else:
        assert len(strides) == dim
        assert strides[-1] == 1
        if dim == 2:
            assert strides[0] >= sizes[1]
    desc = wmb.make_tensor_desc(list(sizes), torch_dtype_to_wholememory_dtype(dtype), list(strides), 0)
    t = C.c_void_p()
    wmb.check(wmb.lib().wholememory_create_tensor(C.byref(t), C.byref(desc), comm.wmb_comm,
                                                  str_to_wmb_wholememory_memory_type(memory_type),
                                                  str_to_wmb_wholememory_location(memory_location),
                                                  wmb.size_t_array(tensor_entry_partition)))
    return WholeMemoryTensor(t, owns=True)


def create_wholememory_tensor_from_filelist(comm, memory_type, memory_location, filelist, dtype, last_dim_size=0,
                                            last_dim_strides=-1, tensor_entry_partition=None):
    if isinstance(filelist, str):
        filelist = [filelist]
    element_size = torch.tensor([], dtype=dtype).element_size()
    if last_dim_strides == -1:
        last_dim_strides = last_dim_size if last_dim_size > 0 else 1
    file_entry_size = element_size * last_dim_size if last_dim_size > 0 else element_size
    total_file_size = 0
    for filename in filelist:
        file_size = get_file_size(filename)
        if file_size % file_entry_size != 0:
            raise ValueError("File %s size is %d not mutlple of %d" % (filename, file_size, file_entry_size))
        total_file_size += file_size
    total_entry_count = total_file_size // file_entry_size
    if last_dim_size == 0:
        sizes, strides = [total_entry_count], [1]
    else:
        sizes, strides = [total_entry_count, last_dim_size], [last_dim_strides, 1]
    wm_tensor = create_wholememory_tensor(comm, memory_type, memory_location, sizes, dtype, strides,
                                          tensor_entry_partition)
    wm_tensor.from_filelist(filelist)
    return wm_tensor


def destroy_wholememory_tensor(wm_tensor):
    wmb.check(wmb.lib().wholememory_destroy_tensor(wm_tensor.wmb_tensor))
    wm_tensor.wmb_tensor = None
