// wholegraph_amd — device row cache of an embedding: sizing, allocation, update / lookup / write-back drivers.
// See embedding_cache.hpp and kernels/cache.hip.
#include "embedding_cache.hpp"

#include <algorithm>
#include <cstring>

namespace wm {

#define WM_BK(call)                                                                                  \
  do {                                                                                               \
    int rc__ = (call);                                                                               \
    if (rc__ != 0) throw ::wm::hip_error(::wm::format_string("%s failed with code %d", #call, rc__)); \
  } while (0)

row_cache::~row_cache()
{
  const auto* bk = backend();
  if (args.slot_of) bk->free_device(args.slot_of);
  if (args.count) bk->free_device(args.count);
  if (args.row_of) bk->free_device(args.row_of);
  if (args.dirty) bk->free_device(args.dirty);
  if (args.data) bk->free_device(args.data);
  if (args.data2) bk->free_device(args.data2);
  if (counters_dev) bk->free_device(counters_dev);
}

wholememory_error_code_t create_row_cache(row_cache** out, const wholememory_embedding_cache_policy_* policy,
                                          wholememory_tensor_t raw, wholememory_comm_t embedding_comm)
{
  const auto* bk = backend();
  if (bk->cache_update == nullptr) return WHOLEMEMORY_NOT_SUPPORTED;
  auto* desc = wholememory_tensor_get_tensor_description(raw);
  auto h     = wholememory_tensor_get_memory_handle(raw);
  std::unique_ptr<row_cache> c(new row_cache());
  c->same_comm = policy->cache_comm == embedding_comm;
  c->writable  = policy->access_type == WHOLEMEMORY_AT_READWRITE;
  c->dtype     = desc->dtype;
  c->row_elems = desc->strides[0];
  c->raw       = raw;
  const size_t es = wholememory_dtype_get_element_size(desc->dtype);
  auto& a         = c->args;
  a.row_bytes     = c->row_elems * static_cast<int64_t>(es);
  if (a.row_bytes % 16 != 0) return WHOLEMEMORY_LOGIC_ERROR;  // embedding rows are padded to 16 bytes
  a.raw_row_stride_bytes = a.row_bytes;
  a.raw_row_offset_bytes = desc->storage_offset * static_cast<int64_t>(es);
  if (c->same_comm) {
    // every rank caches the rows it owns; the raw shard is addressed by GLOBAL row through a flat base
    std::vector<size_t> off = entry_offsets_of(h, static_cast<size_t>(a.row_bytes));
    a.cover_start           = static_cast<int64_t>(off[embedding_comm->world_rank]);
    a.cover_rows            = static_cast<int64_t>(off[embedding_comm->world_rank + 1]) - a.cover_start;
    a.raw_gref              = local_shard_gref(h);
  } else {
    // each rank caches rows of the whole table for its own lookups: the raw table must be addressable from here
    a.cover_start = 0;
    a.cover_rows  = desc->sizes[0];
    if (wholememory_get_memory_type(h) == WHOLEMEMORY_MT_DISTRIBUTED && embedding_comm->world_size > 1) {
      c->raw_addressable = false;  // rows arrive through the exchange (ops.cpp:gather_cached)
      a.raw_gref         = wholememory_create_continuous_global_reference(nullptr);
    } else {
      WHOLEMEMORY_RETURN_ON_FAIL(tensor_mapped_gref(raw, &a.raw_gref));
    }
  }
  if (a.cover_rows >= (INT64_C(1) << 31)) return WHOLEMEMORY_NOT_SUPPORTED;
  // reference embedding_cache.cpp: cache_ratio of the covered rows, in whole sets
  int64_t slots = static_cast<int64_t>(static_cast<double>(a.cover_rows) * policy->cache_ratio);
  slots         = std::min(a.cover_rows + 63, std::max<int64_t>(slots, a.cover_rows > 0 ? 64 : 0));
  a.n_sets      = (slots + 63) / 64;
  a.set_cover   = a.n_sets > 0 ? (a.cover_rows + a.n_sets - 1) / a.n_sets : 0;
  const int64_t n_slots = a.n_sets * 64;
  auto dev_alloc = [&](void** p, size_t bytes) { WM_BK(bk->malloc_device(p, std::max<size_t>(bytes, 16))); };
  dev_alloc(reinterpret_cast<void**>(&a.slot_of), sizeof(int32_t) * a.cover_rows);
  dev_alloc(reinterpret_cast<void**>(&a.count), sizeof(int32_t) * a.cover_rows);
  dev_alloc(reinterpret_cast<void**>(&a.row_of), sizeof(int64_t) * n_slots);
  dev_alloc(reinterpret_cast<void**>(&a.dirty), n_slots);
  dev_alloc(reinterpret_cast<void**>(&a.data), static_cast<size_t>(n_slots) * a.row_bytes);
  dev_alloc(reinterpret_cast<void**>(&c->counters_dev), 32);
  WM_BK(bk->memset_async(a.slot_of, 0xff, sizeof(int32_t) * a.cover_rows, nullptr));
  WM_BK(bk->memset_async(a.count, 0, sizeof(int32_t) * a.cover_rows, nullptr));
  WM_BK(bk->memset_async(a.row_of, 0xff, sizeof(int64_t) * n_slots, nullptr));
  WM_BK(bk->memset_async(a.dirty, 0, n_slots, nullptr));
  WM_BK(bk->memset_async(c->counters_dev, 0, 32, nullptr));
  WM_BK(bk->stream_sync(nullptr));
  *out = c.release();
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t row_cache_update(row_cache* c, const void* ids, wholememory_dtype_t index_dtype, int64_t n,
                                          int64_t key_upper_bound, wholememory_env_func_t* env, void* stream)
{
  const auto* bk = backend();
  if (n == 0 || c->args.n_sets == 0) return WHOLEMEMORY_SUCCESS;
  if (n >= (INT64_C(1) << 31)) return WHOLEMEMORY_INVALID_INPUT;
  temp_mem unique_ids(env), run_starts(env), order(env), n_unique(env), ws(env);
  void* d_unique  = unique_ids.device(n, index_dtype);
  auto* d_starts  = static_cast<int32_t*>(run_starts.device(n + 1, WHOLEMEMORY_DT_INT));
  auto* d_order   = static_cast<int32_t*>(order.device(n, WHOLEMEMORY_DT_INT));
  auto* d_nunique = static_cast<int64_t*>(n_unique.device(1, WHOLEMEMORY_DT_INT64));
  void* d_ws      = ws.device(static_cast<int64_t>(bk->dedup_workspace_bytes(n, index_dtype)), WHOLEMEMORY_DT_INT8);
  int rc = bk->dedup_ids(ids, index_dtype, n, key_upper_bound, 0, d_unique, d_starts, d_order, d_nunique, d_ws, stream);
  if (rc != 0) return rc == -1 ? WHOLEMEMORY_INVALID_INPUT : WHOLEMEMORY_CUDA_ERROR;
  rc = bk->cache_update(&c->args, d_unique, index_dtype, d_starts, d_nunique, n, nullptr, nullptr, nullptr, stream);
  if (rc != 0) return rc == -1 ? WHOLEMEMORY_INVALID_INPUT : WHOLEMEMORY_CUDA_ERROR;
  WM_BK(bk->stream_sync(stream));  // scratch buffers return to the caller's allocator
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t row_cache_plan(row_cache* c, const void* ids, wholememory_dtype_t index_dtype, int64_t n, int64_t key_upper_bound,
                                        wholememory_env_func_t* env, void* stream, temp_mem* rows_mem, temp_mem* slots_mem,
                                        int64_t* n_fill)
{
  const auto* bk = backend();
  *n_fill        = 0;
  auto* fill_rows  = static_cast<int64_t*>(rows_mem->device(n, WHOLEMEMORY_DT_INT64));
  auto* fill_slots = static_cast<int64_t*>(slots_mem->device(n, WHOLEMEMORY_DT_INT64));
  if (n == 0 || c->args.n_sets == 0) return WHOLEMEMORY_SUCCESS;
  if (n >= (INT64_C(1) << 31) || c->writable) return WHOLEMEMORY_INVALID_INPUT;
  temp_mem unique_ids(env), run_starts(env), order(env), n_unique(env), ws(env), count(env), host_n(env);
  void* d_unique  = unique_ids.device(n, index_dtype);
  auto* d_starts  = static_cast<int32_t*>(run_starts.device(n + 1, WHOLEMEMORY_DT_INT));
  auto* d_order   = static_cast<int32_t*>(order.device(n, WHOLEMEMORY_DT_INT));
  auto* d_nunique = static_cast<int64_t*>(n_unique.device(1, WHOLEMEMORY_DT_INT64));
  void* d_ws      = ws.device(static_cast<int64_t>(bk->dedup_workspace_bytes(n, index_dtype)), WHOLEMEMORY_DT_INT8);
  auto* d_count   = static_cast<int*>(count.device(1, WHOLEMEMORY_DT_INT));
  WM_BK(bk->memset_async(d_count, 0, sizeof(int), stream));
  int rc = bk->dedup_ids(ids, index_dtype, n, key_upper_bound, 0, d_unique, d_starts, d_order, d_nunique, d_ws, stream);
  if (rc != 0) return rc == -1 ? WHOLEMEMORY_INVALID_INPUT : WHOLEMEMORY_CUDA_ERROR;
  rc = bk->cache_update(&c->args, d_unique, index_dtype, d_starts, d_nunique, n, fill_rows, fill_slots, d_count, stream);
  if (rc != 0) return rc == -1 ? WHOLEMEMORY_INVALID_INPUT : WHOLEMEMORY_CUDA_ERROR;
  auto* h = static_cast<int*>(host_n.pinned(1, WHOLEMEMORY_DT_INT));
  WM_BK(bk->memcpy_async(h, d_count, sizeof(int), stream));
  WM_BK(bk->stream_sync(stream));
  *n_fill = *h;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t row_cache_install(row_cache* c, const void* rows_data, const int64_t* slots, int64_t n_fill,
                                           void* stream)
{
  const auto* bk = backend();
  if (n_fill == 0) return WHOLEMEMORY_SUCCESS;
  wm_rows_args a{};  // scatter of dense rows into the cache lines
  a.gref          = wholememory_create_continuous_global_reference(c->args.data);
  a.table_dtype   = c->dtype;
  a.dim           = c->row_elems;
  a.table_stride  = c->row_elems;
  a.indices       = slots;
  a.index_dtype   = WHOLEMEMORY_DT_INT64;
  a.n             = n_fill;
  a.plain         = const_cast<void*>(rows_data);
  a.plain_dtype   = c->dtype;
  a.plain_stride  = c->row_elems;
  a.max_blocks    = -1;
  WM_BK(bk->scatter_rows(&a, stream));
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t row_cache_split(row_cache* c, const void* ids, wholememory_dtype_t index_dtype, int64_t n,
                                         int64_t* cache_idx, void* raw_idx, void* stream)
{
  int rc = backend()->cache_split(&c->args, ids, index_dtype, n, cache_idx, raw_idx, c->counters_dev, stream);
  if (rc != 0) return rc == -1 ? WHOLEMEMORY_INVALID_INPUT : WHOLEMEMORY_CUDA_ERROR;
  c->lookups += n;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t row_cache_gather(row_cache* c, const wm_rows_args& a, wholememory_env_func_t* env, void* stream)
{
  const auto* bk = backend();
  if (a.n == 0) return WHOLEMEMORY_SUCCESS;
  temp_mem cache_idx_mem(env), raw_idx_mem(env);
  auto* cache_idx = static_cast<int64_t*>(cache_idx_mem.device(a.n, WHOLEMEMORY_DT_INT64));
  void* raw_idx   = raw_idx_mem.device(a.n, a.index_dtype);
  WHOLEMEMORY_RETURN_ON_FAIL(row_cache_split(c, a.indices, a.index_dtype, a.n, cache_idx, raw_idx, stream));
  // hits: out of the cache lines (a dense [slots, row_elems] table of the raw dtype)
  wm_rows_args hit        = a;
  hit.gref                = wholememory_create_continuous_global_reference(c->args.data);
  hit.table_stride        = c->row_elems;
  hit.table_storage_offset = 0;
  hit.indices             = cache_idx;
  hit.index_dtype         = WHOLEMEMORY_DT_INT64;
  WM_BK(bk->gather_rows(&hit, stream));
  // misses (and nothing for negative ids): out of the raw table
  wm_rows_args miss = a;
  miss.indices      = raw_idx;
  WM_BK(bk->gather_rows(&miss, stream));
  WM_BK(bk->stream_sync(stream));  // the two index lists return to the caller's allocator
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t row_cache_attach_states(row_cache* c, wholememory_tensor_t state_local)
{
  const auto* bk = backend();
  if (!c->same_comm || !c->writable || c->args.data2 != nullptr) return WHOLEMEMORY_INVALID_INPUT;
  WHOLEMEMORY_RETURN_ON_FAIL(row_cache_writeback(c, true, nullptr));
  auto* d = wholememory_tensor_get_tensor_description(state_local);
  if (d->dtype != WHOLEMEMORY_DT_FLOAT || d->dim != 2) return WHOLEMEMORY_INVALID_INPUT;
  auto& a                 = c->args;
  a.row_bytes2            = d->strides[0] * static_cast<int64_t>(sizeof(float));
  a.raw2_row_stride_bytes = a.row_bytes2;
  if (a.row_bytes2 % 16 != 0) return WHOLEMEMORY_LOGIC_ERROR;
  // flat base through which GLOBAL row ids address this rank's state shard
  char* local = static_cast<char*>(wholememory_tensor_get_data_pointer(state_local));
  a.raw2_gref = wholememory_create_continuous_global_reference(local - a.cover_start * a.row_bytes2);
  void* p     = nullptr;
  WM_BK(bk->malloc_device(&p, std::max<size_t>(static_cast<size_t>(a.n_sets) * 64 * a.row_bytes2, 16)));
  a.data2 = static_cast<char*>(p);
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t row_cache_writeback(row_cache* c, bool drop, void* stream)
{
  const auto* bk = backend();
  int rc         = bk->cache_writeback(&c->args, drop ? 1 : 0, stream);
  if (rc != 0) return WHOLEMEMORY_CUDA_ERROR;
  return bk->stream_sync(stream) == 0 ? WHOLEMEMORY_SUCCESS : WHOLEMEMORY_CUDA_ERROR;
}

wholememory_error_code_t row_cache_info(row_cache* c, int64_t* slots, int64_t* occupied, int64_t* dirty, int64_t* hits,
                                        int64_t* lookups, void* stream)
{
  const auto* bk = backend();
  if (bk->cache_info(&c->args, c->counters_dev + 1, stream) != 0) return WHOLEMEMORY_CUDA_ERROR;
  unsigned long long h[3] = {0, 0, 0};
  WM_BK(bk->memcpy_async(h, c->counters_dev, sizeof(h), stream));
  WM_BK(bk->stream_sync(stream));
  if (slots) *slots = c->args.n_sets * 64;
  if (occupied) *occupied = static_cast<int64_t>(h[1]);
  if (dirty) *dirty = static_cast<int64_t>(h[2]);
  if (hits) *hits = static_cast<int64_t>(h[0]);
  if (lookups) *lookups = c->lookups;
  return WHOLEMEMORY_SUCCESS;
}

}  // namespace wm
