"""Glue between torch-ROCm and the C ABI: the env-function table (scratch memory comes from torch's
caching allocator), the current HIP stream, and torch.Tensor -> wholememory_tensor_t wrapping.

Reference counterpart: ``python/pylibwholegraph/pylibwholegraph/torch/wholegraph_env.py:27-182`` (Python
callback env fns) and ``torch_cpp_ext/torch_env_func_ptrs.cpp`` (native env fns: ``csrc/torch_env.cpp`` here).
"""
import ctypes as C
import threading
import weakref

import torch

from .. import binding as wmb
from .utils import torch_dtype_to_wholememory_dtype, wholememory_dtype_to_torch_dtype


def device_memory_is_host():
    """True only under the CPU test backend (oracle/test_backend.cpp, WHOLEGRAPH_AMD_TESTING=1): what the
    library calls device memory is then plain host memory. Always False in the product."""
    return wmb.lib().wholememory_ext_backend_name() != b"hip-gfx950"


def op_device():
    """torch device op inputs/outputs live on."""
    return "cpu" if device_memory_is_host() else "cuda:%d" % torch.cuda.current_device()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_have_gpu = None


def get_stream(use_default=True):
    """Current torch HIP stream as an integer (0 = the null stream). (The raw getter skips the Stream object torch builds for
    `current_stream()`: 0.4 instead of 3.3 us on the path of every op.)"""
    global _have_gpu
    if _have_gpu is None:
        _have_gpu = bool(torch.cuda.is_available())
    if not _have_gpu:
        return 0
    if _raw_stream is not None:
        return int(_raw_stream(torch.cuda.current_device()))
    s = torch.cuda.current_stream().cuda_stream
    return int(s) if s is not None else 0


# Output memory contexts (variable-size results of sampling / append_unique): the caller creates one, passes its
# integer key as the `void* memory_context` of the C call, and reads the tensor the library allocated through
# output_fns.malloc_fn afterwards. reference wholegraph_env.py:42-82
_OUTPUT_KEY_BASE = 1 << 40
_output_contexts = weakref.WeakValueDictionary()
_output_key_lock = threading.Lock()
_output_next_key = [_OUTPUT_KEY_BASE]


class TorchMemoryContext(object):
    def __init__(self):
        self.tensor = None
        with _output_key_lock:
            self.handle = _output_next_key[0]
            _output_next_key[0] += 1
        _output_contexts[self.handle] = self

    def get_c_context(self):
        return self.handle

    def get_handle(self):
        return self.handle

    def set_tensor(self, t):
        self.tensor = t

    def get_tensor(self):
        return self.tensor

    def free(self):
        self.tensor = None

    def free_data(self):
        self.tensor = None


class _EnvTable(object):
    """Owns the ctypes callbacks + every live scratch tensor (keyed by a small integer context)."""

    def __init__(self):
        self._slots = {}
        self._next = 1
        self._lock = threading.Lock()
        self._create = wmb.CREATE_CTX_FN(self._create_ctx)
        self._destroy = wmb.DESTROY_CTX_FN(self._destroy_ctx)
        self._malloc = wmb.MALLOC_FN(self._malloc_fn)
        self._free = wmb.FREE_FN(self._free_fn)
        self.env = wmb.EnvFunc()
        self.env.temporary_fns.create_memory_context_fn = self._create
        self.env.temporary_fns.destroy_memory_context_fn = self._destroy
        self.env.temporary_fns.malloc_fn = self._malloc
        self.env.temporary_fns.free_fn = self._free
        self.env.temporary_fns.global_context = None
        self.env.output_fns.malloc_fn = self._malloc
        self.env.output_fns.free_fn = self._free
        self.env.output_fns.global_context = None

    def _create_ctx(self, p_ctx, _global):
        with self._lock:
            key = self._next
            self._next += 1
            self._slots[key] = None
        p_ctx[0] = key

    def _destroy_ctx(self, ctx, _global):
        with self._lock:
            self._slots.pop(int(ctx or 0), None)

    def _malloc_fn(self, p_desc, alloc_type, ctx, _global):
        d = p_desc.contents
        shape = [int(d.sizes[i]) for i in range(d.dim)]
        dtype = wholememory_dtype_to_torch_dtype(d.dtype)
        if alloc_type == wmb.MA_DEVICE:
            t = torch.empty(shape, dtype=dtype, device=op_device())
        elif alloc_type == wmb.MA_PINNED:
            t = torch.empty(shape, dtype=dtype, device="cpu", pin_memory=torch.cuda.is_available())
        else:
            t = torch.empty(shape, dtype=dtype, device="cpu")
        key = int(ctx or 0)
        if key >= _OUTPUT_KEY_BASE:
            out = _output_contexts.get(key)
            if out is not None:
                out.set_tensor(t)
            return t.data_ptr()
        with self._lock:
            self._slots[key] = t
        return t.data_ptr()

    def _free_fn(self, ctx, _global):
        if int(ctx or 0) >= _OUTPUT_KEY_BASE:
            out = _output_contexts.get(int(ctx))
            if out is not None:
                out.set_tensor(None)
            return
        with self._lock:
            if int(ctx or 0) in self._slots:
                self._slots[int(ctx or 0)] = None

    def tensor_of(self, ctx):
        return self._slots.get(int(ctx))


class _NativeEnvTable(_EnvTable):
    """Scratch buffers straight from torch's HIP caching allocator in C++ (csrc/torch_env.cpp, libwg_torch_env.so — the
    counterpart of the reference's torch C++ extension); only the variable-size OUTPUT buffers, which must become torch
    tensors, still come through the Python callbacks of the base class."""

    def __init__(self, lib):
        super().__init__()
        lib.wg_torch_env_init.restype = None
        lib.wg_torch_env_init.argtypes = [C.POINTER(wmb.EnvFunc), wmb.MALLOC_FN, wmb.FREE_FN, C.c_void_p]
        self._lib = lib
        self.env = wmb.EnvFunc()
        lib.wg_torch_env_init(C.byref(self.env), self._malloc, self._free, None)


def _load_native_env():
    import os
    if os.environ.get("WG_NATIVE_ENV", "1") == "0" or device_memory_is_host():
        return None
    path = os.path.join(os.path.dirname(wmb.LIB_PATH), "libwg_torch_env.so")
    if not os.path.exists(path):
        return None
    try:
        return C.CDLL(path)
    except OSError:
        return None


_default_env = None


def get_wholegraph_env_fns(use_default=True):
    """ctypes pointer to a wholememory_env_func_t backed by torch allocations: the native table when libwg_torch_env.so
    was built (WG_NATIVE_ENV=0 forces the all-Python one), else Python callbacks around torch.empty."""
    global _default_env
    if _default_env is None or not use_default:
        lib = _load_native_env()
        table = _NativeEnvTable(lib) if lib is not None else _EnvTable()
        if use_default:
            _default_env = table
        # both tables hand out memory of torch's caching allocator on the current stream, which is also the stream every
        # op of this layer is called with: stream-ordered scratch and outputs, so the sampling ops may return with their
        # last kernels queued (the library's default keeps the reference's drain for other env functions)
        wmb.check(wmb.lib().wholememory_ext_set_async_completion(1))
    else:
        table = _default_env
    return C.pointer(table.env)


_desc_cache = {}
# Handles of wrapped tensors, per thread: a wholememory_tensor_t made from a pointer is nothing but (pointer, description), so the
# handle of (data_ptr, shape, strides, dtype) can be used again by whoever passes a live tensor with that key — a serving or
# training loop wraps the same few buffers call after call (2 library calls per wrapped tensor and op otherwise: 3 of the 12 us
# a small gather spends on the host). Thread-local, so that emptying a full cache never destroys a handle another thread is
# passing to the library.
_tls = threading.local()
_HANDLE_CACHE_MAX = 512


class _HandleBox(object):
    """owns one wholememory_tensor_t made from a pointer; destroyed with the last reference (the cache's or a wrapper's), so
    emptying the cache never pulls a handle from under a wrapper that is still alive"""
    __slots__ = ("h",)

    def __init__(self, h):
        self.h = h

    def __del__(self):
        try:
            if self.h:
                wmb.lib().wholememory_destroy_tensor(self.h)
                self.h = None
        except Exception:
            pass


class WrappedLocalTensor(object):
    """A torch tensor wrapped as a (non-owning) wholememory_tensor_t for the duration of one call."""

    def __init__(self, t):
        self.torch_tensor = t  # keep the storage alive
        self.handle = C.c_void_p()
        self._owned = True
        if t is None:
            desc = wmb.make_tensor_desc([], wmb.DT_UNKNOWN)
            wmb.check(wmb.lib().wholememory_make_tensor_from_pointer(C.byref(self.handle), None, C.byref(desc)))
            return
        # shape/stride/dtype + data_ptr(), storage_offset 0 (reference wholegraph_env.py:173-182)
        # (an empty torch tensor may report stride 0: describe it as dense instead)
        # The description only depends on (shape, strides, dtype): a training loop wraps the same few layouts over and over,
        # so finished descriptions are kept (the library copies the struct, it is never written through)
        shape = tuple(t.shape)
        strides = tuple(t.stride()) if t.numel() > 0 else None
        key = (shape, strides, t.dtype)
        cache = getattr(_tls, "handles", None)
        if cache is None:
            cache = _tls.handles = {}
        hkey = (t.data_ptr(), key)
        cached = cache.get(hkey)
        if cached is not None:
            self._box, self.handle, self._owned = cached, cached.h, False
            return
        desc = _desc_cache.get(key)
        if desc is None:
            desc = wmb.make_tensor_desc(list(shape), torch_dtype_to_wholememory_dtype(t.dtype),
                                        list(strides) if strides is not None else None, 0)
            if len(_desc_cache) < 4096:
                _desc_cache[key] = desc
        wmb.check(wmb.lib().wholememory_make_tensor_from_pointer(C.byref(self.handle), C.c_void_p(t.data_ptr()),
                                                                 C.byref(desc)))
        if len(cache) >= _HANDLE_CACHE_MAX:      # start over: boxes nobody else holds go now, the others with their wrappers
            cache.clear()
        self._box = cache[hkey] = _HandleBox(self.handle)
        self._owned = False

    @property
    def _as_parameter_(self):
        """ctypes passes the wrapper itself: `lib.wholememory_gather(table, wrap_torch_tensor(idx), ...)` keeps the temporary
        alive for the call, which `wrap_torch_tensor(idx).handle` does not (the wrapper owns the C handle)."""
        return self.handle

    def __del__(self):
        try:
            if self.handle and self._owned:
                wmb.lib().wholememory_destroy_tensor(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass


def wrap_torch_tensor(t):
    return WrappedLocalTensor(t)


class _PointerView(object):
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can alias it."""

    def __init__(self, ptr, shape, torch_dtype, strides_elems, owner):
        import numpy as np
        np_dtype = {torch.float: "<f4", torch.half: "<f2", torch.double: "<f8", torch.int: "<i4", torch.int64: "<i8",
                    torch.int16: "<i2", torch.int8: "|i1", torch.bfloat16: "<i2"}[torch_dtype]
        itemsize = np.dtype(np_dtype).itemsize
        self.__cuda_array_interface__ = {
            "shape": tuple(shape), "typestr": np_dtype, "data": (int(ptr), False), "version": 3,
            "strides": tuple(int(s) * itemsize for s in strides_elems),
        }
        self._owner = owner


def torch_tensor_from_pointer(ptr, shape, torch_dtype, strides_elems, device_memory, owner=None):
    """Alias `ptr` as a torch tensor (no copy). Device memory -> cuda tensor; host memory -> cpu tensor."""
    if device_memory and device_memory_is_host():
        device_memory = False
    numel = 1
    for s in shape:
        numel *= int(s)
    if numel == 0:
        return torch.empty(list(shape), dtype=torch_dtype, device="cuda" if device_memory else "cpu")
    if device_memory:
        view = _PointerView(ptr, shape, torch_dtype, strides_elems, owner)
        t = torch.as_tensor(view, device="cuda")
        if torch_dtype == torch.bfloat16:
            t = t.view(torch.bfloat16)
        t._wm_owner = owner
        return t
    span = 1 + sum((int(s) - 1) * int(st) for s, st in zip(shape, strides_elems))
    itemsize = torch.tensor([], dtype=torch_dtype).element_size()
    buf = (C.c_char * (span * itemsize)).from_address(int(ptr))
    flat = torch.frombuffer(buf, dtype=torch_dtype, count=span)
    t = torch.as_strided(flat, list(shape), list(strides_elems))
    t._wm_owner = owner
    return t
