"""CPU oracle for the WholeMemory gather / scatter / gradient-apply path.

TEST INFRASTRUCTURE ONLY (see oracle/wm_oracle.c header): numpy-facing wrappers over
``oracle/libwm_oracle.so`` plus multi-rank simulations of the reference's distributed flows.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` import
this package; nothing under ``wholegraph_amd/`` does.

Reference citations are relative to /root/reference/cpp/src.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libwm_oracle.so")

# wholememory_dtype_t values (include/wholememory/tensor_description.h)
DT_FLOAT, DT_HALF, DT_DOUBLE, DT_BF16, DT_INT, DT_INT64, DT_INT16, DT_INT8 = 1, 2, 3, 4, 5, 6, 7, 8

_NP2DT = {
    np.dtype(np.float32): DT_FLOAT,
    np.dtype(np.float16): DT_HALF,
    np.dtype(np.float64): DT_DOUBLE,
    np.dtype(np.int32): DT_INT,
    np.dtype(np.int64): DT_INT64,
    np.dtype(np.int16): DT_INT16,
    np.dtype(np.int8): DT_INT8,
}
_DT2NP = {v: k for k, v in _NP2DT.items()}
_DT2NP[DT_BF16] = np.dtype(np.uint16)  # bf16 carried as raw bits


def build(force=False):
    """Compile the C oracle (and oracle/_ref when the reference tree is present)."""
    srcs = [os.path.join(_HERE, f) for f in ("wm_oracle.c", "wm_graph_oracle.c")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "libwm_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/cpp/src/wholememory/tensor_description.cpp"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        c = ctypes
        L.wmo_dtype_size.restype = c.c_size_t
        L.wmo_dtype_size.argtypes = [c.c_int]
        L.wmo_align_embedding_dim.restype = c.c_int64
        L.wmo_align_embedding_dim.argtypes = [c.c_int64, c.c_int64]
        L.wmo_round_robin_total_entries.restype = c.c_int64
        L.wmo_round_robin_total_entries.argtypes = [c.c_int64, c.c_int, c.c_int]
        L.wmo_equal_partition.restype = None
        L.wmo_equal_partition.argtypes = [c.c_uint64, c.c_int, c.c_void_p, c.c_void_p]
        L.wmo_custom_partition.restype = None
        L.wmo_custom_partition.argtypes = [c.c_void_p, c.c_int, c.c_void_p, c.POINTER(c.c_int)]
        L.wmo_gather.restype = c.c_int
        L.wmo_gather.argtypes = [c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_int64, c.c_int64, c.c_int64,
                                 c.c_void_p, c.c_int, c.c_int64, c.c_void_p, c.c_void_p, c.c_int, c.c_int64,
                                 c.c_int64]
        L.wmo_scatter.restype = c.c_int
        L.wmo_scatter.argtypes = [c.c_void_p, c.c_int, c.c_int64, c.c_int64, c.c_void_p, c.c_int, c.c_int64,
                                  c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_int64, c.c_int64, c.c_int64]
        L.wmo_bucket_counts.restype = None
        L.wmo_bucket_counts.argtypes = [c.c_void_p, c.c_int, c.c_int64, c.c_void_p, c.c_int, c.c_void_p]
        L.wmo_sort_ids.restype = None
        L.wmo_sort_ids.argtypes = [c.c_void_p, c.c_int, c.c_int64, c.c_void_p, c.c_void_p]
        L.wmo_round_robin_map.restype = None
        L.wmo_round_robin_map.argtypes = [c.c_void_p, c.c_int, c.c_int64, c.c_int64, c.c_int, c.c_int, c.c_void_p]
        L.wmo_round_robin_map_ex.restype = None
        L.wmo_round_robin_map_ex.argtypes = [c.c_void_p, c.c_int, c.c_int64, c.c_int64, c.c_int, c.c_int, c.c_int64, c.c_void_p]
        L.wmo_dedup_grads.restype = c.c_int64
        L.wmo_dedup_grads.argtypes = [c.c_void_p, c.c_int, c.c_int64, c.c_void_p, c.c_int64, c.c_int64,
                                      c.c_void_p, c.c_void_p]
        f = c.c_float
        L.wmo_sgd_step.restype = None
        L.wmo_sgd_step.argtypes = [c.c_void_p, c.c_int, c.c_int64, c.c_void_p, c.c_int64, c.c_void_p, c.c_int64,
                                   c.c_int64, c.c_int64, f, f]
        L.wmo_lazy_adam_step.restype = None
        L.wmo_lazy_adam_step.argtypes = [c.c_void_p, c.c_int, c.c_int64, c.c_void_p, c.c_int64, c.c_void_p,
                                         c.c_void_p, c.c_void_p, c.c_int64, c.c_int64, c.c_int64, f, f, f, f,
                                         c.c_int, f]
        L.wmo_adagrad_step.restype = None
        L.wmo_adagrad_step.argtypes = [c.c_void_p, c.c_int, c.c_int64, c.c_void_p, c.c_int64, c.c_void_p,
                                       c.c_void_p, c.c_int64, c.c_int64, c.c_int64, f, f, f]
        L.wmo_rmsprop_step.restype = None
        L.wmo_rmsprop_step.argtypes = [c.c_void_p, c.c_int, c.c_int64, c.c_void_p, c.c_int64, c.c_void_p,
                                       c.c_void_p, c.c_int64, c.c_int64, c.c_int64, f, f, f, f]
        L.wmo_fill_closed_form.restype = None
        L.wmo_fill_closed_form.argtypes = [c.c_void_p, c.c_int, c.c_int64, c.c_int64, c.c_int64, c.c_int64]
        L.wmo_half_to_float.restype = c.c_float
        L.wmo_half_to_float.argtypes = [c.c_uint16]
        L.wmo_float_to_half.restype = c.c_uint16
        L.wmo_float_to_half.argtypes = [c.c_float]
        L.wmo_float_to_bf16.restype = c.c_uint16
        L.wmo_float_to_bf16.argtypes = [c.c_float]
        L.wmo_bf16_to_float.restype = c.c_float
        L.wmo_bf16_to_float.argtypes = [c.c_uint16]
        L.wmo_num_threads.restype = c.c_int
        L.wmo_set_num_threads.argtypes = [c.c_int]
        # wm_graph_oracle.c
        L.wmo_pcg_raw.restype = None
        L.wmo_pcg_raw.argtypes = [c.c_uint64, c.c_uint64, c.c_uint64, c.c_int64, c.c_void_p]
        L.wmo_random_positive_int.restype = None
        L.wmo_random_positive_int.argtypes = [c.c_int64, c.c_int64, c.c_int64, c.c_int, c.c_void_p]
        L.wmo_sample_offsets.restype = c.c_int64
        L.wmo_sample_offsets.argtypes = [c.c_void_p, c.c_void_p, c.c_int, c.c_int64, c.c_int, c.c_void_p]
        L.wmo_sample_unweighted.restype = None
        L.wmo_sample_unweighted.argtypes = [c.c_void_p, c.c_void_p, c.c_int, c.c_void_p, c.c_int, c.c_int64, c.c_int,
                                            c.c_uint64, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]
        L.wmo_det_log2_1p.restype = c.c_double
        L.wmo_det_log2_1p.argtypes = [c.c_double]
        L.wmo_weighted_keys.restype = None
        L.wmo_weighted_keys.argtypes = [c.c_uint64, c.c_uint64, c.c_int64, c.c_void_p, c.c_void_p]
        L.wmo_sample_weighted.restype = None
        L.wmo_sample_weighted.argtypes = [c.c_void_p, c.c_void_p, c.c_int, c.c_void_p, c.c_int, c.c_void_p, c.c_int,
                                          c.c_int64, c.c_int, c.c_uint64, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]
        L.wmo_append_unique.restype = c.c_int64
        L.wmo_append_unique.argtypes = [c.c_void_p, c.c_int64, c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p]
        L.wmo_csr_add_self_loop.restype = None
        L.wmo_csr_add_self_loop.argtypes = [c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


def np_to_dt(dtype):
    return _NP2DT[np.dtype(dtype)]


def dt_to_np(dt):
    return _DT2NP[dt]


# ------------------------------------------------------------------ layout / partition
def align_embedding_dim(dim, elt_size):
    return lib().wmo_align_embedding_dim(dim, elt_size)


def round_robin_total_entries(n, world, rr):
    return lib().wmo_round_robin_total_entries(n, world, rr)


def equal_partition(total_entries, world):
    sizes = np.zeros(world, dtype=np.uint64)
    offs = np.zeros(world + 1, dtype=np.uint64)
    lib().wmo_equal_partition(total_entries, world, _p(sizes), _p(offs))
    return sizes, offs


def custom_partition(entries):
    entries = np.ascontiguousarray(entries, dtype=np.uint64)
    offs = np.zeros(len(entries) + 1, dtype=np.uint64)
    same = ctypes.c_int(0)
    lib().wmo_custom_partition(_p(entries), len(entries), _p(offs), ctypes.byref(same))
    return offs, bool(same.value)


class ShardedTable:
    """A row-sharded table held as one numpy array per rank (the oracle's WholeMemory tensor)."""

    def __init__(self, shards, entry_offsets, dim, stride=None, storage_offset=0, dt=None):
        self.shards = [np.ascontiguousarray(s) for s in shards]
        self.entry_offsets = np.ascontiguousarray(entry_offsets, dtype=np.uint64)
        self.world = len(shards)
        self.dim = dim
        self.stride = stride if stride is not None else dim
        self.storage_offset = storage_offset
        self.dt = dt if dt is not None else np_to_dt(self.shards[0].dtype)
        self._ptrs = (ctypes.c_void_p * self.world)(*[s.ctypes.data for s in self.shards])

    @classmethod
    def from_full(cls, full2d, world=1, entries=None, dt=None):
        """Split a dense [N, stride] array by the equal plan (or a custom entry partition)."""
        n = full2d.shape[0]
        if entries is None:
            _, offs = equal_partition(n, world)
        else:
            offs, _ = custom_partition(entries)
        shards = [np.array(full2d[int(offs[r]):int(offs[r + 1])], copy=True) for r in range(world)]
        # keep zero-row shards addressable
        shards = [s if s.size else np.zeros((1, full2d.shape[1]), dtype=full2d.dtype) for s in shards]
        return cls(shards, offs, full2d.shape[1], full2d.shape[1], 0, dt)


def gather(table, indices, out, *, dim=None, out_stride=None, out_storage_offset=0, raw_indices=None, out_dt=None):
    indices = np.ascontiguousarray(indices)
    dim = table.dim if dim is None else dim
    if out_stride is None:
        out_stride = out.shape[1] if out.ndim == 2 else 1
    if raw_indices is not None:
        raw_indices = np.ascontiguousarray(raw_indices, dtype=np.int64)
    rc = lib().wmo_gather(table._ptrs, _p(table.entry_offsets), table.world, table.dt, dim, table.stride,
                          table.storage_offset, _p(indices), np_to_dt(indices.dtype), indices.size,
                          _p(raw_indices), _p(out), out_dt if out_dt is not None else np_to_dt(out.dtype),
                          out_stride, out_storage_offset)
    if rc != 0:
        raise ValueError("oracle gather: unsupported dtype combination")
    return out


def scatter(inp, indices, table, *, dim=None, in_stride=None, in_storage_offset=0, in_dt=None):
    indices = np.ascontiguousarray(indices)
    dim = table.dim if dim is None else dim
    if in_stride is None:
        in_stride = inp.shape[1] if inp.ndim == 2 else 1
    rc = lib().wmo_scatter(_p(inp), in_dt if in_dt is not None else np_to_dt(inp.dtype), in_stride,
                           in_storage_offset, _p(indices), np_to_dt(indices.dtype), indices.size, table._ptrs,
                           _p(table.entry_offsets), table.world, table.dt, dim, table.stride,
                           table.storage_offset)
    if rc != 0:
        raise ValueError("oracle scatter: unsupported dtype combination")


# ------------------------------------------------------------------ bucketing / exchange
def bucket_counts(indices, entry_offsets):
    indices = np.ascontiguousarray(indices)
    offs = np.ascontiguousarray(entry_offsets, dtype=np.uint64)
    counts = np.zeros(len(offs) - 1, dtype=np.int64)
    lib().wmo_bucket_counts(_p(indices), np_to_dt(indices.dtype), indices.size, _p(offs), len(offs) - 1, _p(counts))
    return counts


def sort_ids(indices):
    """(sorted_ids, raw_indices) exactly as exchange_ids_nccl_func.cu:42-92."""
    indices = np.ascontiguousarray(indices)
    sorted_out = np.empty_like(indices)
    raw = np.empty(indices.size, dtype=np.int64)
    lib().wmo_sort_ids(_p(indices), np_to_dt(indices.dtype), indices.size, _p(sorted_out), _p(raw))
    return sorted_out, raw


def round_robin_map(indices, entry_start, world, rr, rank_rows=0):
    """rank_rows == 0: the reference statement (map_indices_func.cu:34-43). rank_rows > 0: the owner's shard (product)."""
    indices = np.ascontiguousarray(indices)
    out = np.empty_like(indices)
    if rank_rows:
        lib().wmo_round_robin_map_ex(_p(indices), np_to_dt(indices.dtype), indices.size, entry_start, world, rr, rank_rows, _p(out))
    else:
        lib().wmo_round_robin_map(_p(indices), np_to_dt(indices.dtype), indices.size, entry_start, world, rr, _p(out))
    return out


def exchange_ids(rank_indices, entry_offsets):
    """Every rank's bucket_and_exchange_ids_func (exchange_ids_nccl_func.cu:157-226) at once.

    Returns per rank: send_counts[W], recv_counts[W], sorted ids, raw_indices, recv ids (rank-major).
    """
    world = len(rank_indices)
    send_counts = [bucket_counts(ix, entry_offsets) for ix in rank_indices]
    sorted_raw = [sort_ids(ix) for ix in rank_indices]
    out = []
    for r in range(world):
        recv_counts = np.array([send_counts[s][r] for s in range(world)], dtype=np.int64)
        pieces = []
        for s in range(world):
            off = int(send_counts[s][:r].sum())
            pieces.append(sorted_raw[s][0][off:off + int(send_counts[s][r])])
        recv = np.concatenate(pieces) if pieces else np.empty(0, dtype=rank_indices[r].dtype)
        out.append(dict(send_counts=send_counts[r], recv_counts=recv_counts, sorted_ids=sorted_raw[r][0],
                        raw_indices=sorted_raw[r][1], recv_ids=recv.astype(rank_indices[r].dtype)))
    return out


def distributed_gather(table, rank_indices, out_np_dtype, out_init=None):
    """wholememory_gather_nccl (wholememory_ops/gather_op_impl_nccl.cu:100-168) for every rank:
    exchange ids -> owner-side gather WITH the dtype cast (:118-121) -> rows all-to-all-v ->
    reorder by raw_indices. Returns one [n_r, dim] array per rank."""
    world = table.world
    ex = exchange_ids(rank_indices, table.entry_offsets)
    local_rows = []
    for r in range(world):
        # owner gathers its received ids from its own shard ("fake" continuous gref, :122-140)
        buf = np.zeros((ex[r]["recv_ids"].size, table.dim), dtype=out_np_dtype)
        gather(table, ex[r]["recv_ids"], buf)
        local_rows.append(buf)
    outs = []
    for r in range(world):
        n = rank_indices[r].size
        recv = np.zeros((n, table.dim), dtype=out_np_dtype)
        pos = 0
        for s in range(world):  # rows come back rank-major, same order the ids were sent in
            cnt = int(ex[r]["send_counts"][s])
            src_off = int(ex[s]["recv_counts"][:r].sum())
            recv[pos:pos + cnt] = local_rows[s][src_off:src_off + cnt]
            pos += cnt
        out = np.zeros((n, table.dim), dtype=out_np_dtype) if out_init is None else np.array(out_init[r], copy=True)
        valid = pos
        out[ex[r]["raw_indices"][:valid]] = recv[:valid]  # scatter_func reorder (:151-168)
        outs.append(out)
    return outs


# ------------------------------------------------------------------ gradient apply
def dedup_grads(indices, grads):
    indices = np.ascontiguousarray(indices)
    grads = np.ascontiguousarray(grads, dtype=np.float32)
    n, dim = grads.shape if grads.ndim == 2 else (0, 0)
    uniq = np.empty_like(indices)
    dg = np.zeros((max(n, 1), dim), dtype=np.float32)
    cnt = lib().wmo_dedup_grads(_p(indices), np_to_dt(indices.dtype), indices.size, _p(grads), dim, dim, _p(uniq),
                                _p(dg))
    return uniq[:cnt].copy(), dg[:cnt].copy()


class Optimizer:
    """Per-rank optimizer state + step, following embedding_optimizer.cpp:100-538 for state shapes
    and the kernels cited in wm_oracle.c for arithmetic."""

    def __init__(self, kind, local_rows, stride, **params):
        self.kind = kind
        self.p = dict(weight_decay=0.0, epsilon=1e-8, alpha=0.99, beta1=0.9, beta2=0.999, adam_w=0.0)
        self.p.update(params)
        rows = max(local_rows, 1)
        if kind == "adam":
            self.per_element = np.zeros((rows, 2 * stride), dtype=np.float32)
            self.per_row = np.ones((rows, 2), dtype=np.float32)
        elif kind in ("adagrad", "rmsprop"):
            self.per_element = np.zeros((rows, stride), dtype=np.float32)

    def step(self, ids, grads, local_table, stride, local_entry_offset, dim, lr):
        ids = np.ascontiguousarray(ids)
        grads = np.ascontiguousarray(grads, dtype=np.float32)
        L, p = lib(), self.p
        gs = grads.shape[1] if grads.ndim == 2 and grads.shape[0] else dim
        a = (_p(ids), np_to_dt(ids.dtype), ids.size, _p(grads), gs)
        if self.kind == "sgd":
            L.wmo_sgd_step(*a, _p(local_table), stride, local_entry_offset, dim, p["weight_decay"], lr)
        elif self.kind == "adam":
            L.wmo_lazy_adam_step(*a, _p(local_table), _p(self.per_element), _p(self.per_row), stride,
                                 local_entry_offset, dim, p["weight_decay"], p["epsilon"], p["beta1"], p["beta2"],
                                 int(p["adam_w"] > 0.5), lr)
        elif self.kind == "adagrad":
            L.wmo_adagrad_step(*a, _p(local_table), _p(self.per_element), stride, local_entry_offset, dim,
                               p["weight_decay"], p["epsilon"], lr)
        elif self.kind == "rmsprop":
            L.wmo_rmsprop_step(*a, _p(local_table), _p(self.per_element), stride, local_entry_offset, dim,
                               p["weight_decay"], p["epsilon"], p["alpha"], lr)
        else:
            raise ValueError(self.kind)


def gradient_apply(table, optimizers, rank_indices, rank_grads, lr):
    """embedding_base::gather_gradient_apply (wholememory/embedding.cpp:146-323) for every rank:
    exchange ids -> grads gathered by raw_indices -> all-to-all-v to owners -> dedup (sorted-order
    sequential sum) -> optimizer step on the owner's shard. Mutates table.shards in place."""
    world = table.world
    ex = exchange_ids(rank_indices, table.entry_offsets)
    for r in range(world):
        pieces = []
        for s in range(world):
            off = int(ex[s]["send_counts"][:r].sum())
            cnt = int(ex[s]["send_counts"][r])
            send_rows = np.asarray(rank_grads[s], dtype=np.float32)[ex[s]["raw_indices"][off:off + cnt]]
            pieces.append(send_rows.reshape(cnt, table.dim))
        recv_grads = np.concatenate(pieces) if pieces else np.zeros((0, table.dim), np.float32)
        uniq, dg = dedup_grads(ex[r]["recv_ids"], recv_grads)
        optimizers[r].step(uniq, dg, table.shards[r], table.stride, int(table.entry_offsets[r]), table.dim, lr)


# ------------------------------------------------------------------ closed-form tables
def fill_closed_form(np_dtype_or_dt, row_start, rows, dim, stride=None):
    dt = np_dtype_or_dt if isinstance(np_dtype_or_dt, int) else np_to_dt(np_dtype_or_dt)
    stride = dim if stride is None else stride
    arr = np.zeros((max(rows, 1), stride), dtype=dt_to_np(dt))
    lib().wmo_fill_closed_form(_p(arr), dt, row_start, rows, dim, stride)
    return arr[:rows] if rows else arr[:0]


def num_threads():
    return lib().wmo_num_threads()


def set_num_threads(n):
    lib().wmo_set_num_threads(n)


# ---------------------------------------------------------------------------------------------- graph ops
def pcg_raw(seed, subsequence, offset, n):
    """n raw 32-bit outputs of PCG-XSH-RR 64/32 after init(seed, subsequence) + skip-ahead by offset."""
    out = np.empty(n, dtype=np.uint32)
    lib().wmo_pcg_raw(seed, subsequence, offset, n, _p(out))
    return out


def random_positive_int(seed, subsequence, n, np_dtype=np.int32):
    out = np.empty(n, dtype=np_dtype)
    lib().wmo_random_positive_int(seed, subsequence, n, 1 if np.dtype(np_dtype) == np.int64 else 0, _p(out))
    return out


def sample_unweighted(row_ptr, col, centers, max_sample, seed, need_lid=True, need_egid=True):
    """-> offsets int32 [n + 1], ids (col dtype), lid int32 | None, egid int64 | None"""
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
    col = np.ascontiguousarray(col)
    centers = np.ascontiguousarray(centers)
    n = centers.shape[0]
    offsets = np.empty(n + 1, dtype=np.int32)
    c64 = 1 if centers.dtype == np.int64 else 0
    total = lib().wmo_sample_offsets(_p(row_ptr), _p(centers), c64, n, max_sample, _p(offsets))
    ids = np.empty(total, dtype=np.int64)
    lid = np.empty(total, dtype=np.int32) if need_lid else None
    egid = np.empty(total, dtype=np.int64) if need_egid else None
    lib().wmo_sample_unweighted(_p(row_ptr), _p(col), 1 if col.dtype == np.int64 else 0, _p(centers), c64, n,
                                max_sample, seed & 0xFFFFFFFFFFFFFFFF, _p(offsets), _p(ids),
                                _p(lid) if need_lid else None, _p(egid) if need_egid else None)
    return offsets, ids.astype(col.dtype), lid, egid


def det_log2_1p(x):
    return lib().wmo_det_log2_1p(float(x))


def weighted_keys(seed, subsequence, weights):
    """consecutive A-Res keys log2(u)/w of one PCG stream"""
    w = np.ascontiguousarray(weights, dtype=np.float32)
    keys = np.empty(w.shape[0], dtype=np.float32)
    lib().wmo_weighted_keys(seed & 0xFFFFFFFFFFFFFFFF, subsequence, w.shape[0], _p(w), _p(keys))
    return keys


def sample_weighted(row_ptr, col, weights, centers, max_sample, seed, need_lid=True, need_egid=True):
    """-> offsets, ids (key-descending per center), lid, egid"""
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
    col = np.ascontiguousarray(col)
    weights = np.ascontiguousarray(weights)
    assert weights.dtype in (np.float32, np.float64)
    centers = np.ascontiguousarray(centers)
    n = centers.shape[0]
    offsets = np.empty(n + 1, dtype=np.int32)
    c64 = 1 if centers.dtype == np.int64 else 0
    total = lib().wmo_sample_offsets(_p(row_ptr), _p(centers), c64, n, max_sample, _p(offsets))
    ids = np.empty(total, dtype=np.int64)
    lid = np.empty(total, dtype=np.int32) if need_lid else None
    egid = np.empty(total, dtype=np.int64) if need_egid else None
    lib().wmo_sample_weighted(_p(row_ptr), _p(col), 1 if col.dtype == np.int64 else 0, _p(weights),
                              1 if weights.dtype == np.float64 else 0, _p(centers), c64, n, max_sample,
                              seed & 0xFFFFFFFFFFFFFFFF, _p(offsets), _p(ids), _p(lid) if need_lid else None,
                              _p(egid) if need_egid else None)
    return offsets, ids.astype(col.dtype), lid, egid


def append_unique(targets, neighbors):
    """-> unique (targets first, then new neighbour ids in first-seen order), mapping int32 [n_neighbor]"""
    dt = np.asarray(targets).dtype
    t = np.ascontiguousarray(targets, dtype=np.int64)
    nb = np.ascontiguousarray(neighbors, dtype=np.int64)
    out = np.empty(t.shape[0] + nb.shape[0], dtype=np.int64)
    mapping = np.empty(nb.shape[0], dtype=np.int32)
    cnt = lib().wmo_append_unique(_p(t), t.shape[0], _p(nb), nb.shape[0], _p(out), _p(mapping))
    return out[:cnt].astype(dt), mapping


def csr_add_self_loop(row_ptr, col):
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    n = row_ptr.shape[0] - 1
    out_row = np.empty(n + 1, dtype=np.int32)
    out_col = np.empty(col.shape[0] + n, dtype=np.int32)
    lib().wmo_csr_add_self_loop(_p(row_ptr), _p(col), n, _p(out_row), _p(out_col))
    return out_row, out_col
