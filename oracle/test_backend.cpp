// oracle/test_backend.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A host-memory implementation of the device seam (wholegraph_amd/csrc/backend.hpp) built from the CPU
// oracle (wm_oracle.c). It exists for exactly one purpose: letting tests/ drive the product's multi-rank
// HOST ORCHESTRATION (ops.cpp / embedding.cpp: partition plan, count exchange, all-to-all-v layout,
// reorder, dedup hand-off) at world_size 2 over torch.distributed/gloo on a CPU-only box. It is installed
// through wm_testing_install_backend(), which refuses to act unless WHOLEGRAPH_AMD_TESTING=1.
// Nothing under wholegraph_amd/ links, loads or references this file; kernels are validated separately on a
// real MI355X against the same oracle (tests -m gpu).
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../wholegraph_amd/csrc/backend.hpp"
#include <wholememory/embedding.h>

extern "C" {
size_t wmo_dtype_size(int dtype);
int wmo_gather(const void* const* shard_ptrs, const uint64_t* entry_offsets, int world_size, int table_dtype,
               int64_t dim, int64_t stride, int64_t storage_offset, const void* indices, int idx_dtype, int64_t n,
               const void* raw_indices, void* out, int out_dtype, int64_t out_stride, int64_t out_storage_offset);
int wmo_scatter_mapped(const void* in, int in_dtype, int64_t in_stride, int64_t in_storage_offset, const void* indices,
                       int idx_dtype, int64_t n, const int64_t* raw_indices, void* const* shard_ptrs,
                       const uint64_t* entry_offsets, int world_size, int table_dtype, int64_t dim, int64_t stride,
                       int64_t storage_offset);
void wmo_bucket_counts(const void* indices, int idx_dtype, int64_t n, const uint64_t* entry_offsets, int world_size,
                       int64_t* counts);
void wmo_round_robin_map_ex(const void* indices, int idx_dtype, int64_t n, int64_t entry_start, int world_size,
                            int round_robin_size, int64_t rank_rows, void* mapped);
void wmo_sgd_step(const void* ids, int idx_dtype, int64_t count, const float* grads, int64_t grad_stride,
                  float* local_table, int64_t table_stride, int64_t local_entry_offset, int64_t dim, float weight_decay,
                  float lr);
void wmo_lazy_adam_step(const void* ids, int idx_dtype, int64_t count, const float* grads, int64_t grad_stride,
                        float* local_table, float* per_element, float* per_row, int64_t table_stride,
                        int64_t local_entry_offset, int64_t dim, float weight_decay, float epsilon, float beta1,
                        float beta2, int adam_w, float lr);
void wmo_adagrad_step(const void* ids, int idx_dtype, int64_t count, const float* grads, int64_t grad_stride,
                      float* local_table, float* state_sum_tbl, int64_t table_stride, int64_t local_entry_offset,
                      int64_t dim, float weight_decay, float epsilon, float lr);
void wmo_rmsprop_step(const void* ids, int idx_dtype, int64_t count, const float* grads, int64_t grad_stride,
                      float* local_table, float* v_tbl, int64_t table_stride, int64_t local_entry_offset, int64_t dim,
                      float weight_decay, float epsilon, float alpha, float lr);
}

namespace {

int64_t idx_at(const void* p, wholememory_dtype_t dt, int64_t i)
{
  return dt == WHOLEMEMORY_DT_INT ? static_cast<const int32_t*>(p)[i] : static_cast<const int64_t*>(p)[i];
}

int t_device_count() { return 1; }
int t_malloc(void** p, size_t bytes)
{
  *p = malloc(bytes ? bytes : 16);
  return *p ? 0 : 1;
}
int t_free(void* p)
{
  free(p);
  return 0;
}
int t_memcpy(void* d, const void* s, size_t n, void*)
{
  memmove(d, s, n);
  return 0;
}
int t_memset(void* d, int v, size_t n, void*)
{
  memset(d, v, n);
  return 0;
}
int t_sync(void*) { return 0; }
int t_stream_create(void** s)
{
  *s = nullptr;  // everything is synchronous here: one "stream"
  return 0;
}
int t_noop1(void*) { return 0; }
int t_event_create(void** e)
{
  *e = nullptr;
  return 0;
}
int t_noop2(void*, void*) { return 0; }
int t_ipc_get(void*, void*) { return 1; }  // no cross-process mapping of malloc'ed "device" memory
int t_ipc_open(void**, const void*) { return 1; }
int t_ipc_close(void*) { return 1; }
int t_host_register(void* h, size_t, void** d)
{
  *d = h;
  return 0;
}
int t_host_unregister(void*) { return 0; }

// flat view of a gref for the oracle: continuous -> one shard at row 0; chunked -> per-rank shards
struct shard_view {
  std::vector<const void*> ptrs;
  std::vector<uint64_t> offs;
  int world;
};
bool make_view(const wm_rows_args* a, shard_view* v)
{
  const size_t es = wmo_dtype_size(a->table_dtype);
  if (a->gref.stride == 0) {
    v->world = 1;
    v->ptrs  = {a->gref.pointer};
    v->offs  = {0, UINT64_MAX};
    return true;
  }
  const size_t row = static_cast<size_t>(a->table_stride) * es;
  v->world         = a->gref.world_size;
  v->ptrs.assign(static_cast<void**>(a->gref.pointer), static_cast<void**>(a->gref.pointer) + v->world);
  v->offs.resize(v->world + 1);
  for (int r = 0; r <= v->world; r++) {
    if (a->gref.rank_memory_offsets[r] % row != 0) return false;
    v->offs[r] = a->gref.rank_memory_offsets[r] / row;
  }
  return true;
}

int t_gather(const wm_rows_args* a, void*)
{
  if (a->n == 0 || a->dim == 0) return 0;
  shard_view v;
  if (!make_view(a, &v)) return -1;
  int rc = wmo_gather(v.ptrs.data(), v.offs.data(), v.world, a->table_dtype, a->dim, a->table_stride,
                      a->table_storage_offset, a->indices, a->index_dtype, a->n, a->row_map, a->plain, a->plain_dtype,
                      a->plain_stride, a->plain_storage_offset);
  return rc == 0 ? 0 : -1;
}
int t_scatter(const wm_rows_args* a, void*)
{
  if (a->n == 0 || a->dim == 0) return 0;
  shard_view v;
  if (!make_view(a, &v)) return -1;
  int rc = wmo_scatter_mapped(a->plain, a->plain_dtype, a->plain_stride, a->plain_storage_offset, a->indices,
                              a->index_dtype, a->n, static_cast<const int64_t*>(a->row_map),
                              const_cast<void* const*>(v.ptrs.data()), v.offs.data(), v.world, a->table_dtype, a->dim,
                              a->table_stride, a->table_storage_offset);
  return rc == 0 ? 0 : -1;
}

size_t t_bucket_ws(int64_t, int) { return 64; }
int t_bucket(const wm_bucket_args* a, void*)
{
  const int owners = a->owner_count > 0 ? a->owner_count : a->world_size;
  // bucket of id i: its owner (the oracle's range search), or owner % world_size in the owner_count mode; negatives go to
  // the trailing bucket `world_size`
  auto bucket_of = [&](int64_t i) {
    int64_t id = idx_at(a->indices, a->index_dtype, i);
    if (id < 0) return a->world_size;
    int owner = 0;
    for (int k = 1; k < owners; k++)
      if (static_cast<uint64_t>(id) >= a->entry_offsets[k]) owner = k;
    return owners == a->world_size ? owner : owner % a->world_size;
  };
  if (a->owner_count <= 0) {
    wmo_bucket_counts(a->indices, a->index_dtype, a->n, a->entry_offsets, a->world_size, a->counts);
  } else {
    for (int r = 0; r < a->world_size; r++) a->counts[r] = 0;
    for (int64_t i = 0; i < a->n; i++) {
      int b = bucket_of(i);
      if (b < a->world_size) a->counts[b]++;
    }
  }
  if (a->bucketed_ids == nullptr) return 0;
  // stable partition by bucket, negatives last (kernels/bucket.hip contract)
  int64_t pos = 0;
  for (int r = 0; r <= a->world_size; r++) {
    for (int64_t i = 0; i < a->n; i++) {
      if (bucket_of(i) != r) continue;
      int64_t id = idx_at(a->indices, a->index_dtype, i);
      if (a->index_dtype == WHOLEMEMORY_DT_INT)
        static_cast<int32_t*>(a->bucketed_ids)[pos] = static_cast<int32_t>(id);
      else
        static_cast<int64_t*>(a->bucketed_ids)[pos] = id;
      a->raw_indices[pos] = i;
      pos++;
    }
  }
  return 0;
}

size_t t_dedup_ws(int64_t, wholememory_dtype_t) { return 64; }
int t_dedup(const void* ids, wholememory_dtype_t dt, int64_t n, int64_t upper, int64_t lower, void* unique_ids, int32_t* run_starts,
            int32_t* order, int64_t* n_unique_out, void*, void*)
{
  std::vector<int32_t> ord(n);
  for (int64_t i = 0; i < n; i++) ord[i] = static_cast<int32_t>(i);
  if (upper <= 0 || lower < 0 || lower >= upper) lower = 0;
  // a bounded range narrower than 2^32 - 1 rows: ids outside it are dropped from the runs (backend.hpp)
  const bool drops = upper > 0 && upper - lower < INT64_C(0xFFFFFFFF);
  auto outside     = [&](int64_t id) { return drops && (id < lower || id >= upper); };
  // the kernels' order: the ids' two's-complement bits as UNSIGNED keys — negative ("skip me") ids after every valid id;
  // dropped ids behind everything, in their original order
  std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) {
    const int64_t a = idx_at(ids, dt, x), b = idx_at(ids, dt, y);
    if (outside(a) || outside(b)) return !outside(a) && outside(b);
    return static_cast<uint64_t>(a) < static_cast<uint64_t>(b);
  });
  int64_t nu = 0, kept = 0;
  for (int64_t i = 0; i < n; i++) {
    order[i] = ord[i];
    if (outside(idx_at(ids, dt, ord[i]))) continue;
    kept = i + 1;
    if (i == 0 || idx_at(ids, dt, ord[i]) != idx_at(ids, dt, ord[i - 1])) {
      if (dt == WHOLEMEMORY_DT_INT)
        static_cast<int32_t*>(unique_ids)[nu] = static_cast<int32_t>(idx_at(ids, dt, ord[i]));
      else
        static_cast<int64_t*>(unique_ids)[nu] = idx_at(ids, dt, ord[i]);
      run_starts[nu] = static_cast<int32_t>(i);
      nu++;
    }
  }
  run_starts[nu] = static_cast<int32_t>(kept);
  *n_unique_out  = nu;
  return 0;
}

int t_step(const wm_optimizer_args* a, const int64_t* n_unique_dev, void*)
{
  if (a->value_dtype != WHOLEMEMORY_DT_FLOAT && a->value_dtype != WHOLEMEMORY_DT_UNKNOWN) return -1;  // fp32 only here
  const int64_t count = n_unique_dev ? *n_unique_dev : a->count;
  const float* grads      = static_cast<const float*>(a->grads);
  const float* self_grads = static_cast<const float*>(a->self_grads);
  float* local_table      = static_cast<float*>(a->local_table);
  std::vector<float> g(a->dim);
  for (int64_t u = 0; u < count; u++) {
    for (int64_t d = 0; d < a->dim; d++) {
      auto row = [&](int32_t o) {  // wm_optimizer_args::self_grads: negative entries address the caller's own rows
        return o >= 0 ? grads + static_cast<int64_t>(o) * a->grad_stride
                      : self_grads + (-(static_cast<int64_t>(o) + 1)) * a->self_grad_stride;
      };
      float acc = row(a->order[a->run_starts[u]])[d];
      for (int32_t j = a->run_starts[u] + 1; j < a->run_starts[u + 1]; j++) acc += row(a->order[j])[d];
      g[d] = acc;
    }
    int64_t id64     = idx_at(a->ids, a->index_dtype, u);
    const void* idp  = &id64;
    const int idx_dt = WHOLEMEMORY_DT_INT64;
    switch (a->type) {
      case WHOLEMEMORY_OPT_SGD:
        wmo_sgd_step(idp, idx_dt, 1, g.data(), a->dim, local_table, a->table_stride, a->local_entry_offset, a->dim,
                     a->weight_decay, a->lr);
        break;
      case WHOLEMEMORY_OPT_LAZY_ADAM:
        if (a->per_element_stride != 2 * a->table_stride) return -1;
        wmo_lazy_adam_step(idp, idx_dt, 1, g.data(), a->dim, local_table, a->per_element_state, a->per_row_state,
                           a->table_stride, a->local_entry_offset, a->dim, a->weight_decay, a->epsilon, a->beta1,
                           a->beta2, a->adam_w, a->lr);
        break;
      case WHOLEMEMORY_OPT_ADAGRAD:
        if (a->per_element_stride != a->table_stride) return -1;
        wmo_adagrad_step(idp, idx_dt, 1, g.data(), a->dim, local_table, a->per_element_state, a->table_stride,
                         a->local_entry_offset, a->dim, a->weight_decay, a->epsilon, a->lr);
        break;
      case WHOLEMEMORY_OPT_RMSPROP:
        if (a->per_element_stride != a->table_stride) return -1;
        wmo_rmsprop_step(idp, idx_dt, 1, g.data(), a->dim, local_table, a->per_element_state, a->table_stride,
                         a->local_entry_offset, a->dim, a->weight_decay, a->epsilon, a->alpha, a->lr);
        break;
      default: return -1;
    }
  }
  return 0;
}

size_t t_long_ws(int64_t, int64_t) { return 64; }
int t_run_inverse(const int32_t* run_starts, const int32_t* order, const void* unique_ids, wholememory_dtype_t dt,
                  const int64_t* n_unique, int64_t n, int64_t id_limit, int64_t* inverse, void*)
{
  for (int64_t u = 0; u < *n_unique; u++) {
    const int64_t id = idx_at(unique_ids, dt, u);
    for (int32_t j = run_starts[u]; j < run_starts[u + 1]; j++) inverse[order[j]] = (id < 0 || (id_limit > 0 && id >= id_limit)) ? -1 : u;
  }
  (void)n;
  return 0;
}
int t_remap_self(int32_t* order, int64_t n, int64_t self_begin, int64_t self_count, const int64_t* self_rows, void*)
{
  for (int64_t i = 0; i < n; i++) {
    const int64_t pos = order[i] - self_begin;
    if (pos >= 0 && pos < self_count) order[i] = static_cast<int32_t>(-(self_rows[pos] + 1));
  }
  return 0;
}
int t_rr(const void* ids, void* mapped, wholememory_dtype_t dt, int64_t n, int64_t entry_start, int world, int rr,
         int64_t rank_rows, void*)
{
  wmo_round_robin_map_ex(ids, dt, n, entry_start, world, rr, rank_rows, mapped);
  return 0;
}
int t_fill(float* p, float v, int64_t n, void*)
{
  for (int64_t i = 0; i < n; i++) p[i] = v;
  return 0;
}

template <typename T>
void env_test_rows(const void* in, void* out, int64_t dim, int64_t entries, int64_t stride)
{
  for (int64_t i = 0; i < entries; i++)
    for (int64_t c = 0; c < dim; c++)
      static_cast<T*>(out)[stride * i + c] = static_cast<T>(static_cast<T>(static_cast<float>(i)) + static_cast<const T*>(in)[c]);
}
// wholememory_env_test_op arithmetic (reference wholememory_test_op.cu:25-37) for the dtypes a CPU can do natively
int t_env_test(const void* in, void* out, wholememory_dtype_t dt, int64_t dim, int64_t entries, int64_t stride, void*)
{
  switch (dt) {
    case WHOLEMEMORY_DT_FLOAT: env_test_rows<float>(in, out, dim, entries, stride); return 0;
    case WHOLEMEMORY_DT_DOUBLE: env_test_rows<double>(in, out, dim, entries, stride); return 0;
    case WHOLEMEMORY_DT_INT: env_test_rows<int32_t>(in, out, dim, entries, stride); return 0;
    case WHOLEMEMORY_DT_INT64: env_test_rows<int64_t>(in, out, dim, entries, stride); return 0;
    default: return -1;
  }
}

size_t t_dup_ws(int64_t) { return 64; }
int t_dup_estimate(const void* ids, wholememory_dtype_t dt, int64_t n, void*, int64_t* permille, void*)
{
  // exact duplicate share of the whole batch (the product samples and estimates; only the decision it feeds must agree
  // between ranks, and that is taken from exchanged values)
  std::vector<int64_t> v(static_cast<size_t>(n));
  for (int64_t i = 0; i < n; i++) v[i] = idx_at(ids, dt, i);
  std::sort(v.begin(), v.end());
  const int64_t distinct = static_cast<int64_t>(std::unique(v.begin(), v.end()) - v.begin());
  *permille              = n > 0 ? (1000 * (n - distinct) + n / 2) / n : 0;
  return 0;
}
int t_sorted_counts(const void* ids, wholememory_dtype_t dt, const int64_t* n_dev, int64_t, const uint64_t* off, int world,
                    int64_t* counts, void*)
{
  for (int r = 0; r < world; r++) counts[r] = 0;
  for (int64_t i = 0; i < *n_dev; i++) {
    const int64_t id = idx_at(ids, dt, i);
    if (id < 0) continue;
    for (int r = 0; r < world; r++)
      if (static_cast<uint64_t>(id) >= off[r] && static_cast<uint64_t>(id) < off[r + 1]) counts[r]++;
  }
  return 0;
}

// chunk-major copy of per-peer segments (backend.hpp: permute_chunks)
int t_permute(const void* src, void* dst, int elt_bytes, const int64_t* off, const int64_t* cnt, int n_segs, int n_chunks, void*)
{
  int64_t k = 0;
  for (int c = 0; c < n_chunks; c++)
    for (int p = 0; p < n_segs; p++) {
      const int64_t a = cnt[p] * c / n_chunks, b = cnt[p] * (c + 1) / n_chunks;
      memcpy(static_cast<char*>(dst) + k * elt_bytes, static_cast<const char*>(src) + (off[p] + a) * elt_bytes,
             static_cast<size_t>(b - a) * elt_bytes);
      k += b - a;
    }
  return 0;
}

int t_iota(void* p, wholememory_dtype_t dt, int64_t n, int64_t first, void*)
{
  for (int64_t i = 0; i < n; i++) {
    if (dt == WHOLEMEMORY_DT_INT) static_cast<int32_t*>(p)[i] = static_cast<int32_t>(first + i);
    else static_cast<int64_t*>(p)[i] = first + i;
  }
  return 0;
}

const wm_device_backend kTestBackend = {
  "oracle-test-backend (CPU, tests only)",
  t_device_count, t_malloc, t_free, t_malloc, t_free, t_memcpy, t_memset, t_sync,
  t_stream_create, t_noop1, t_event_create, t_noop1, t_noop2, t_noop2,
  t_ipc_get, t_ipc_open, t_ipc_close, t_host_register, t_host_unregister,
  t_gather, t_scatter, t_bucket_ws, t_bucket, t_dedup_ws, t_dedup, t_step, t_long_ws, t_run_inverse, t_remap_self, t_rr, t_fill,
  t_dup_ws, t_dup_estimate, t_sorted_counts,
  // graph ops: not provided by the CPU backend
  nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
  t_env_test,
  // device cache, placement probe, id sort, memory info, append_unique extras, 0xFF fill: not provided
  nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
  t_permute,
  nullptr, nullptr,   // no side stream of its own: nothing to join later
  nullptr,            // ... and no device-side waits that could give up
  t_iota,
  nullptr,            // float32 tables only: no float16 partial sums to check
};

}  // namespace

extern "C" const void* wm_test_backend() { return &kTestBackend; }
