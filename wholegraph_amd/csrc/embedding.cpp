// wholegraph_amd — WholeMemory embedding object + sparse optimizers (host orchestration).
//
// Reference: cpp/src/wholememory/embedding.cpp:43-50 (row padding), :87-144 (allocate), :146-323
// (gather_gradient_apply), :325-448 (optimizer state tensors), :467-484 (round-robin shard size),
// :946-1121 (C API); embedding_optimizer.cpp:100-538 (optimizer objects, parameters, states).
//
// Gradient apply on MI355X: ids are bucketed by owner (one multisplit pass), gradient rows are
// lined up in send order and exchanged by RCCL all-to-all-v, and the owner runs ONE fused kernel per
// step (duplicate-sum in the reference's order + optimizer update, kernels/optim.hip) — there is no
// intermediate de-duplicated gradient buffer and no host sync to learn the unique count.
// Cached embeddings: embedding_cache.{hpp,cpp} + kernels/cache.hip (a device row cache with a direct row -> slot map).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <atomic>
#include <wholememory/embedding.h>
#include <wholememory/wholegraph_amd_ext.h>

#include "knobs.hpp"
#include "embedding_cache.hpp"
#include "ops_internal.hpp"

struct wholememory_embedding_optimizer_ {
  wholememory_optimizer_type_t type = WHOLEMEMORY_OPT_NONE;
  // defaults: reference embedding_optimizer.cpp (class member initialisers)
  float weight_decay = 0.0f;
  float epsilon      = 1e-8f;
  float beta1        = 0.9f;
  float beta2        = 0.999f;
  float alpha        = 0.99f;
  float adam_w       = 0.0f;
  // extension (not in the reference): order of the fp32 sum of a run of duplicate gradient rows, see
  // wm_optimizer_args::fold_mode — -1 the default of the table dtype (ordered for fp32: the reference's bits), 0 ordered,
  // 1 tree (deterministic, equal within rounding, 30 % faster under heavy skew). Parameter name "grad_fold".
  float grad_fold    = -1.0f;
  std::vector<const char*> state_names;  // nullptr-terminated
  std::map<std::string, float*> params;
};

struct wholememory_embedding_ {
  wholememory_tensor_t allocated = nullptr;  // padded table [N, stride]
  wholememory_tensor_t user      = nullptr;  // [N, dim] view handed to callers
  wholememory_comm_t comm        = nullptr;
  wholememory_dtype_t dtype      = WHOLEMEMORY_DT_UNKNOWN;
  int gather_sms                 = -1;
  int round_robin_size           = 0;
  wholememory_embedding_optimizer_t optimizer = nullptr;
  // optimizer state (reference optimizer_state_t)
  wholememory_embedding_t state_embedding = nullptr;   // per-element states packed side by side
  wholememory_tensor_t state_local        = nullptr;   // local shard of the packed states
  std::map<std::string, wholememory_tensor_t> state_views;
  wholememory_tensor_t per_row_padded = nullptr;       // "beta12t" [N_alloc, 2], DISTRIBUTED
  wholememory_tensor_t per_row_view   = nullptr;
  wholememory_tensor_t per_row_local  = nullptr;
  int64_t state_row_elems             = 0;             // columns of the packed state table
  wm::row_cache* cache                = nullptr;       // device row cache (cache policy given at creation)
};

namespace wm {
extern std::atomic<int64_t> g_grad_exchange_launches;   // row kernels queued in front of the gradient exchange (ops.cpp)
namespace {

#define WM_BK(call)                                                                                  \
  do {                                                                                               \
    int rc__ = (call);                                                                               \
    if (rc__ != 0) throw ::wm::hip_error(::wm::format_string("%s failed with code %d", #call, rc__)); \
  } while (0)

int64_t align_embedding_dim(int64_t dim, size_t element_size)
{  // rows padded to 16 bytes: reference embedding.cpp:43-50 — and, round 5, to whole 128-byte lines when that is cheap.
  // A row that does not start on a line boundary is WRITTEN with a partial line at either end; scatter and gradient apply of
  // such rows run 15-20 points under rows of whole lines (800 B rows 56 -> 74 % of the HBM peak at stride 1024, 4000 B rows
  // 60 -> 74 %: profiles/r04_misaligned_rows.txt). GloVe / word2vec (100 / 200 / 300 floats) and Reddit (602) tables are such
  // rows. WM_EMBEDDING_ROW_ALIGN:
  //   auto (default)  pad the stride to a multiple of 128 bytes when that costs at most 8 % more memory than the reference's
  //                   16-byte padding (602 floats -> 608, 300 -> 320, 513 -> 544; 100 and 200 floats stay at 400 / 800 bytes)
  //   16              the reference's padding, always
  //   32 ... 4096     (a power of two) pad to that many bytes, whatever it costs
  // Only the STRIDE of tables the library allocates itself changes: the user still sees [N, dim], files are written and read
  // per logical row (file_io.cpp: file entry size = dim x element size), so they stay interchangeable between alignments.
  const int64_t row = dim * static_cast<int64_t>(element_size);
  auto padded = [&](int64_t bytes) { return (row + bytes - 1) / bytes * bytes; };
  int64_t bytes = 0;   // 0 = auto
  if (const char* e = WM_KNOB("WM_EMBEDDING_ROW_ALIGN")) {
    const int64_t v = atoll(e);
    if (strcmp(e, "auto") == 0 || strcmp(e, "AUTO") == 0) bytes = 0;   // (exactly: "a..." anything used to pass, advisor)
    else if (v >= 16 && v <= 4096 && (v & (v - 1)) == 0) bytes = v;
    else WM_WARN("WM_EMBEDDING_ROW_ALIGN=%s ignored: auto or a power of two between 16 and 4096 is expected", e);
  }
  const bool by_rule = bytes == 0;
  if (bytes == 0) bytes = padded(128) * 100 <= padded(16) * 108 ? 128 : 16;
  const int64_t a      = std::max<int64_t>(bytes / static_cast<int64_t>(element_size), 1);
  const int64_t stride = dim % a == 0 ? dim : (dim / a + 1) * a;
  // not a silent change of a default (advisor, round 5): the first table of a process whose stride differs from the
  // reference's says so, once, with the way back
  const int64_t ra = std::max<int64_t>(16 / static_cast<int64_t>(element_size), 1);
  const int64_t reference_stride = dim % ra == 0 ? dim : (dim / ra + 1) * ra;
  static std::atomic<bool> said{false};
  if (by_rule && stride != reference_stride && !said.exchange(true))
    WM_INFO("embedding rows of %ld elements are stored with a stride of %ld elements (whole 128-byte lines, +%.1f %% memory) "
            "instead of the reference's %ld (16-byte padding): scatter and gradient apply write whole lines. Files are "
            "unaffected. WM_EMBEDDING_ROW_ALIGN=16 restores the reference stride.",
            static_cast<long>(dim), static_cast<long>(stride), 100.0 * (stride - reference_stride) / reference_stride,
            static_cast<long>(reference_stride));
  return stride;
}

int per_element_state_count(wholememory_optimizer_type_t t)
{
  switch (t) {
    case WHOLEMEMORY_OPT_LAZY_ADAM: return 2;  // m, v
    case WHOLEMEMORY_OPT_ADAGRAD: return 1;    // state_sum
    case WHOLEMEMORY_OPT_RMSPROP: return 1;    // v
    default: return 0;
  }
}

void destroy_states(wholememory_embedding_* e)
{
  for (auto& kv : e->state_views) {
    if (kv.second != e->per_row_view) wholememory_destroy_tensor(kv.second);
  }
  e->state_views.clear();
  if (e->state_local) wholememory_destroy_tensor(e->state_local);
  if (e->state_embedding) wholememory_destroy_embedding(e->state_embedding);
  if (e->per_row_local) wholememory_destroy_tensor(e->per_row_local);
  if (e->per_row_view) wholememory_destroy_tensor(e->per_row_view);
  if (e->per_row_padded) wholememory_destroy_tensor(e->per_row_padded);
  e->state_local = e->per_row_local = e->per_row_view = e->per_row_padded = nullptr;
  e->state_embedding                                                       = nullptr;
}

// reference embedding.cpp:325-428 (create_optimizer_states) + init_optimizer_states of each optimizer
wholememory_error_code_t create_states(wholememory_embedding_* e)
{
  const auto* bk   = backend();
  auto* opt        = e->optimizer;
  auto* alloc_desc = wholememory_tensor_get_tensor_description(e->allocated);
  auto* user_desc  = wholememory_tensor_get_tensor_description(e->user);
  const int64_t dim = user_desc->sizes[1];
  int W;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_communicator_get_size(&W, e->comm));
  std::vector<size_t> alloc_part(W), user_part(W);
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_tensor_get_entry_partition_sizes(alloc_part.data(), e->allocated));
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_tensor_get_entry_partition_sizes(user_part.data(), e->user));

  const int n_states = per_element_state_count(opt->type);
  if (n_states > 0) {
    const size_t es       = wholememory_dtype_get_element_size(user_desc->dtype);
    const int64_t aligned = align_embedding_dim(dim, es);
    e->state_row_elems    = aligned * n_states;
    wholememory_tensor_description_t sd = *user_desc;
    sd.sizes[1]                         = e->state_row_elems;
    sd.strides[0]                       = e->state_row_elems;
    auto h  = wholememory_tensor_get_memory_handle(e->allocated);
    WHOLEMEMORY_RETURN_ON_FAIL(wholememory_create_embedding(&e->state_embedding, &sd, e->comm,
                                                            wholememory_get_memory_type(h),
                                                            wholememory_get_memory_location(h), nullptr,
                                                            user_part.data(), -1, 0));
    auto state_user = wholememory_embedding_get_embedding_tensor(e->state_embedding);
    WHOLEMEMORY_RETURN_ON_FAIL(wholememory_tensor_map_local_tensor(state_user, &e->state_local));
    const char* names_adam[] = {"m", "v"};
    for (int i = 0; i < n_states; i++) {
      const char* nm = opt->type == WHOLEMEMORY_OPT_LAZY_ADAM ? names_adam[i]
                       : opt->type == WHOLEMEMORY_OPT_ADAGRAD ? "state_sum"
                                                              : "v";
      int64_t starts[2] = {0, aligned * i};
      int64_t ends[2]   = {-1, aligned * i + dim};
      wholememory_tensor_t view;
      WHOLEMEMORY_RETURN_ON_FAIL(wholememory_tensor_get_subtensor(state_user, starts, ends, &view));
      e->state_views[nm] = view;
    }
    // zero the local shard of the packed states (reference zero_local_state_tensor)
    auto* ld = wholememory_tensor_get_tensor_description(e->state_local);
    if (WM_AB_KNOB("WM_STATE_ZERO_MEMSET") != nullptr)
      WM_BK(bk->memset_async(wholememory_tensor_get_data_pointer(e->state_local), 0,
                             static_cast<size_t>(ld->sizes[0]) * ld->strides[0] * sizeof(float), nullptr));
    else
      WM_BK(bk->fill_float(static_cast<float*>(wholememory_tensor_get_data_pointer(e->state_local)), 0.0f,
                           ld->sizes[0] * ld->strides[0], nullptr));
    WM_BK(bk->stream_sync(nullptr));
    // a read-write device cache also holds the states of its resident rows (reference: cachable optimizer states)
    if (e->cache != nullptr && e->cache->writable) WHOLEMEMORY_RETURN_ON_FAIL(wm::row_cache_attach_states(e->cache, e->state_local));
  }
  if (opt->type == WHOLEMEMORY_OPT_LAZY_ADAM) {
    // per-row [beta1^t, beta2^t]: DISTRIBUTED device tensor partitioned like the table, init 1.0
    wholememory_tensor_description_t pd = *alloc_desc;
    pd.dtype                            = WHOLEMEMORY_DT_FLOAT;
    pd.sizes[1] = pd.strides[0] = 2;
    WHOLEMEMORY_RETURN_ON_FAIL(wholememory_create_tensor(&e->per_row_padded, &pd, e->comm, WHOLEMEMORY_MT_DISTRIBUTED,
                                                         WHOLEMEMORY_ML_DEVICE, alloc_part.data()));
    int64_t starts[2] = {0, 0};
    int64_t ends[2]   = {user_desc->sizes[0], 2};
    WHOLEMEMORY_RETURN_ON_FAIL(wholememory_tensor_get_subtensor(e->per_row_padded, starts, ends, &e->per_row_view));
    WHOLEMEMORY_RETURN_ON_FAIL(wholememory_tensor_map_local_tensor(e->per_row_view, &e->per_row_local));
    e->state_views["beta12t"] = e->per_row_view;
    auto* ld = wholememory_tensor_get_tensor_description(e->per_row_local);
    WM_BK(bk->fill_float(static_cast<float*>(wholememory_tensor_get_data_pointer(e->per_row_local)), 1.0f,
                         ld->sizes[0] * 2, nullptr));
  }
  WM_BK(bk->stream_sync(nullptr));
  return WHOLEMEMORY_SUCCESS;
}

void fill_optimizer_args(wm_optimizer_args* a, const wholememory_embedding_optimizer_* o, float lr)
{
  a->type         = o->type;
  a->weight_decay = o->weight_decay;
  a->epsilon      = o->epsilon;
  a->beta1        = o->beta1;
  a->beta2        = o->beta2;
  a->alpha        = o->alpha;
  a->adam_w       = o->adam_w > 0.5f ? 1 : 0;
  a->lr           = lr;
  // the caller's choice (optimizer parameter "grad_fold"), else the backend's default for the table dtype (ordered for fp32);
  // WM_GRAD_FOLD in the environment overrides both
  a->fold_mode    = o->grad_fold < 0.0f ? -1 : (o->grad_fold > 0.5f ? 1 : 0);
}

// owner side: sort received ids, then the fused duplicate-sum + optimizer kernel
// `rows_ready` (optional event): recorded on another stream when recv_grads is complete; the id sort does not need
// the rows, so it is issued first and only the step kernel waits for the event.
// gradient rows of ids this rank owns itself, left in the caller's tensor: receive positions
// [begin, begin + count) stand for rows `rows[i]` of `grads` (see wm_optimizer_args::self_grads)
struct self_rows_ref {
  int64_t begin = 0, count = 0;
  const int64_t* rows = nullptr;
  const void* grads   = nullptr;
  int64_t stride      = 0;
};

// sorted view of a batch of received ids (unique ids, run starts, sorted order): the scratch lives as long as the object
struct dedup_result {
  explicit dedup_result(wholememory_env_func_t* env) : unique_ids(env), run_starts(env), order(env), n_unique(env), ws(env) {}
  // join_later: the caller queues the optimizer step behind the sort and calls join() after it (the sort's side stream is
  // then joined behind the step instead of in front of it: backend.hpp, dedup_defer_join); the destructor joins in any case
  ~dedup_result() { (void)join(); }
  // 0, or the backend's error: the join is also where a device-side wait of the sort that gave up is reported (backend.hpp:
  // device_error) — by then the sort has turned itself into "no runs", so the step behind it changed nothing
  int join()
  {
    int rc = 0;
    if (join_owed) {
      const auto* bk = backend();
      if (bk->dedup_join != nullptr) rc = bk->dedup_join(deferred_on);
      join_owed = false;
    }
    return rc;
  }
  void run(const void* ids, wholememory_dtype_t index_dtype, int64_t n, int64_t key_upper_bound, void* stream,
           int64_t key_lower_bound = 0, bool join_later = false)
  {
    const auto* bk = backend();
    struct defer_scope {
      const wm_device_backend* bk;
      bool on;
      defer_scope(const wm_device_backend* b, bool o) : bk(b), on(o && b->dedup_defer_join != nullptr && b->dedup_join != nullptr)
      {
        if (on) bk->dedup_defer_join(1);
      }
      ~defer_scope()
      {
        if (on) bk->dedup_defer_join(0);
      }
    } scope(bk, join_later);
    if (scope.on) deferred_on = stream, join_owed = true;
    d_unique  = unique_ids.device(n, index_dtype);
    d_starts  = static_cast<int32_t*>(run_starts.device(n + 1, WHOLEMEMORY_DT_INT));
    d_order   = static_cast<int32_t*>(order.device(n, WHOLEMEMORY_DT_INT));
    d_nunique = static_cast<int64_t*>(n_unique.device(1, WHOLEMEMORY_DT_INT64));
    void* d_ws = ws.device(static_cast<int64_t>(bk->dedup_workspace_bytes(n, index_dtype)), WHOLEMEMORY_DT_INT8);
    int rc = bk->dedup_ids(ids, index_dtype, n, key_upper_bound, key_lower_bound, d_unique, d_starts, d_order, d_nunique, d_ws, stream);
    if (rc == -1) throw logic_error("dedup_ids: unsupported index dtype or more than 2^31 received ids");
    if (rc != 0) throw hip_error("dedup_ids failed");
  }
  temp_mem unique_ids, run_starts, order, n_unique, ws;
  void* deferred_on  = nullptr;   // stream a deferred join is owed on (the null stream is a stream like any other)
  bool join_owed     = false;
  void* d_unique     = nullptr;
  int32_t* d_starts  = nullptr;
  int32_t* d_order   = nullptr;
  int64_t* d_nunique = nullptr;
};

// the fused duplicate-sum + optimizer kernel over an already sorted batch
void step_sorted(dedup_result& r, wholememory_dtype_t index_dtype, int64_t n_recv, const void* recv_grads, int64_t grad_stride,
                 wm_optimizer_args* oa, wholememory_env_func_t* env, void* stream, int64_t* n_unique_host, void* rows_ready,
                 const self_rows_ref* self)
{
  const auto* bk = backend();
  temp_mem host_n(env), long_ws(env);
  oa->ids         = r.d_unique;
  oa->index_dtype = index_dtype;
  oa->run_starts  = r.d_starts;
  oa->order       = r.d_order;
  oa->grads       = recv_grads;
  oa->grad_stride = grad_stride;
  if (self != nullptr && self->count > 0) {
    WM_BK(bk->remap_self_order(r.d_order, n_recv, self->begin, self->count, self->rows, stream));
    oa->self_grads       = self->grads;
    oa->self_grad_stride = self->stride;
  }
  oa->count       = n_recv;  // upper bound; the kernel reads the true count from d_nunique
  oa->long_run_ws_bytes = bk->long_run_workspace_bytes(n_recv, oa->dim);
  oa->long_run_ws       = long_ws.device(static_cast<int64_t>(oa->long_run_ws_bytes), WHOLEMEMORY_DT_INT8);
  if (rows_ready != nullptr) WM_BK(bk->stream_wait_event(stream, rows_ready));
  int rc = bk->optimizer_step(oa, r.d_nunique, stream);
  const int join_rc = r.join();   // the sort's side stream, if it left one running: joined behind the step
  if (rc != 0) throw hip_error("optimizer_step failed");
  if (join_rc != 0) throw hip_error("the id sort of an earlier gradient step reported a device-side timeout (see the ERROR line above)");
  if (n_unique_host != nullptr) {
    auto* h = static_cast<int64_t*>(host_n.pinned(1, WHOLEMEMORY_DT_INT64));
    WM_BK(bk->memcpy_async(h, r.d_nunique, sizeof(int64_t), stream));
    WM_BK(bk->stream_sync(stream));
    *n_unique_host = *h;
    if (bk->device_error != nullptr && bk->device_error() != 0)   // (synchronised: this call's own sort has reported by now)
      throw hip_error("the id sort of this gradient step reported a device-side timeout (see the ERROR line above)");
  }
}

void dedup_and_step(const void* recv_ids, wholememory_dtype_t index_dtype, int64_t n_recv, const void* recv_grads,
                    int64_t grad_stride, wm_optimizer_args* oa, int64_t key_upper_bound, wholememory_env_func_t* env,
                    void* stream, int64_t* n_unique_host, void* rows_ready = nullptr, const self_rows_ref* self = nullptr,
                    int64_t key_lower_bound = 0)
{
  const auto* bk = backend();
  if (n_recv == 0) {
    if (rows_ready != nullptr) WM_BK(bk->stream_wait_event(stream, rows_ready));  // fence the scratch buffers anyway
    if (n_unique_host) *n_unique_host = 0;
    return;
  }
  dedup_result r(env);
  r.run(recv_ids, index_dtype, n_recv, key_upper_bound, stream, key_lower_bound, /*join_later=*/true);
  step_sorted(r, index_dtype, n_recv, recv_grads, grad_stride, oa, env, stream, n_unique_host, rows_ready, self);
}

// What the owner was given: ids and gradient rows of every requester in rank-major receive order (that order defines the fp32
// sum of duplicates), or — `sorted` — a batch that was sorted already (one rank: the caller's arrays are the receive buffers)
struct owner_input {
  const void* recv_ids            = nullptr;
  wholememory_dtype_t index_dtype = WHOLEMEMORY_DT_INT64;
  int64_t n_recv                  = 0;
  const void* rows                = nullptr;   // [n_recv, row_stride] gradient rows (the table's dtype)
  int64_t row_stride              = 0;
  const self_rows_ref* self       = nullptr;   // receive positions that stand for rows of the caller's own tensor
  void* rows_arrived              = nullptr;   // event: `rows` is complete (the id sort does not wait for it)
  dedup_result* sorted            = nullptr;
};

// owner side of gather_gradient_apply (reference embedding.cpp:248-318): fused dedup + optimizer step on the local shard
wholememory_error_code_t owner_apply(wholememory_embedding_* e, const owner_input& in, const std::vector<size_t>& entry_offsets,
                                     float lr, wholememory_env_func_t* env, void* stream, bool adjust_cache)
{
  const auto* bk    = backend();
  auto* adesc       = wholememory_tensor_get_tensor_description(e->allocated);
  const int rank    = e->comm->world_rank;
  const int64_t dim = wholememory_tensor_get_tensor_description(e->user)->sizes[1];
  wholememory_tensor_t local_table;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_tensor_map_local_tensor(e->user, &local_table));
  wm_optimizer_args oa{};
  fill_optimizer_args(&oa, e->optimizer, lr);
  oa.local_table        = wholememory_tensor_get_data_pointer(local_table);
  oa.value_dtype        = e->dtype;
  oa.table_stride       = adesc->strides[0];
  oa.local_entry_offset = static_cast<int64_t>(entry_offsets[rank]);
  oa.dim                = dim;
  if (e->state_local != nullptr) {
    oa.per_element_state  = static_cast<float*>(wholememory_tensor_get_data_pointer(e->state_local));
    oa.per_element_stride = wholememory_tensor_get_tensor_description(e->state_local)->strides[0];
  }
  if (e->per_row_local != nullptr)
    oa.per_row_state = static_cast<float*>(wholememory_tensor_get_data_pointer(e->per_row_local));
  wholememory_destroy_tensor(local_table);
  if (e->cache != nullptr) {
    // read-write device cache of this rank's shard (reference: the optimizer kernels work through the cache,
    // embedding_optimizer_func.cu + embedding.cpp:146-323): resident rows are updated in their cache line and marked
    // modified, the others in the raw table; the packed per-element states of resident rows live in companion cache lines
    if (adjust_cache)
      WHOLEMEMORY_RETURN_ON_FAIL(wm::row_cache_update(e->cache, in.recv_ids, in.index_dtype, in.n_recv,
                                                      static_cast<int64_t>(entry_offsets[e->comm->world_size]), env, stream));
    oa.cache_slot_of   = e->cache->args.slot_of;
    oa.cache_data      = e->cache->args.data;
    oa.cache_dirty     = e->cache->args.dirty;
    oa.cache_row_elems = e->cache->row_elems;
    if (e->cache->args.data2 != nullptr) {
      oa.cache_state_data      = reinterpret_cast<float*>(e->cache->args.data2);
      oa.cache_state_row_elems = e->cache->args.row_bytes2 / static_cast<int64_t>(sizeof(float));
    }
  }
  if (in.sorted != nullptr)
    step_sorted(*in.sorted, in.index_dtype, in.n_recv, in.rows, in.row_stride, &oa, env, stream, nullptr, in.rows_arrived, nullptr);
  else
    dedup_and_step(in.recv_ids, in.index_dtype, in.n_recv, in.rows, in.row_stride, &oa,
                   static_cast<int64_t>(entry_offsets[rank + 1]), env, stream, nullptr, in.rows_arrived, in.self,
                   static_cast<int64_t>(entry_offsets[rank]));
  // The reference returns with the optimizer kernels still queued (embedding.cpp:318-323: its scratch goes back to the env
  // allocator, which is stream-ordered — include/wholememory/env_func_ptrs.h states that contract). The same here on ONE
  // rank: no trailing synchronise, so a training loop's next step is prepared while this one runs (the end-of-call bubble
  // was ~0.1 ms of a 3.2 ms step). With several ranks the stream is drained before returning: a peer may read this shard
  // through its own mapping (CHUNKED / CONTINUOUS) right after the barrier that follows the step, and that barrier orders
  // hosts, not this stream.
  if (e->comm->world_size > 1 || debug_sync_enabled()) {
    WM_BK(bk->stream_sync(stream));
    // everything of this call has finished: a device-side wait that gave up is reported by THIS call (without the
    // synchronise — one rank — by the next entry into the sort or its join)
    if (bk->device_error != nullptr && bk->device_error() != 0) return WHOLEMEMORY_CUDA_ERROR;
  }
  return WHOLEMEMORY_SUCCESS;
}

// is the fp32 sum of a run's duplicate gradients free of the reference's order? (wm_optimizer_args::fold_mode: WM_GRAD_FOLD
// overrides, then the optimizer's "grad_fold", then the default of the value dtype — ordered for fp32, tree for 16-bit tables)
bool fold_is_tree(const wholememory_embedding_optimizer_* o, wholememory_dtype_t vdt)
{
  const char* env = WM_KNOB("WM_GRAD_FOLD");
  if (env != nullptr && (env[0] == 't' || env[0] == 'T')) return true;
  if (env != nullptr && (env[0] == 'o' || env[0] == 'O')) return false;
  if (o->grad_fold >= 0.0f) return o->grad_fold > 0.5f;
  return vdt == WHOLEMEMORY_DT_HALF || vdt == WHOLEMEMORY_DT_BF16;
}

std::atomic<int64_t> g_grad_combined_calls{0};

// SENDER-SIDE COMBINATION of duplicate gradient rows (round 6; not in the reference, which ships every copy:
// embedding.cpp:193-247). Under skew every rank holds thousands of gradient rows for the same hot ids — Zipf(1.05), 10 M ids:
// 49 % distinct, 527 k copies of the hottest — and every copy would cross ONE xGMI link to its owner, who then folds W x 527 k
// rows (in the reference's order: a dependent chain of ~11 ms at W = 8). When the order of the fp32 sum is not bound to the
// reference's (fold_is_tree), each rank first folds ITS OWN duplicates:
//   1 runs of equal ids in the batch (the owner-side id sort, here on the sender: ids sorted, run starts, positions)
//   2 one partial sum per distinct id: the fused fold + step kernels run as "row u += sum of run u" on a dense zeroed buffer
//     (SGD with lr = -1, no weight decay: 0 - (-1) x sum, exact) — duplicates summed in fp32 in a fixed order, long runs by
//     the tree kernels, rounded ONCE to the exchange dtype (= the table's)
//   3 the distinct ids are sorted, so the owner segments are contiguous pieces of ids and partial rows as they stand: counts by
//     binary search, no multisplit, no line-up kernel — ids and rows go out by all-to-all-v straight from those arrays
//   4 the owner folds at most W partial rows per id (rank-major order) and applies the optimizer: runs of <= W rows, no long
//     dependent chain whatever the skew.
// Results: a fixed order (deterministic), equal to the ordered sum within the usual bound of fp32 summation — exact whenever the
// partial sums are exactly representable (integer-valued gradients) — for 16-bit tables plus one rounding of each sender's
// partial sum to the table's dtype. float16 partial sums can leave the float16 range where single gradients cannot: every
// rank checks its partial rows and the verdict rides in the counts exchange (slot W, like the duplicate estimate of the
// gather), so ALL ranks fall back to the uncombined route together for such a batch.
// Whether to combine is decided per call by all ranks alike: WM_GRAD_COMBINE=0 never, =1 whenever the fold is free, unset:
// when the mean duplicate estimate that rides in the counts exchange reaches WM_GRAD_COMBINE_PERMILLE (default 100 = 10 %).
// Returns 1 = the step is done (*rc holds its code), 0 = declined by every rank (nothing changed: take the plain route).
int combined_gradient_apply(wholememory_embedding_* e, const char* idx_ptr, const wholememory_array_description_t& iarr,
                            const void* grads_ptr, const wholememory_matrix_description_t& gmat,
                            const std::vector<size_t>& entry_offsets, float lr, wholememory_env_func_t* env, void* stream,
                            bool adjust_cache, wholememory_error_code_t* rc)
{
  const auto* bk          = backend();
  const int W             = e->comm->world_size;
  const int rank          = e->comm->world_rank;
  const int64_t n         = iarr.size;
  const int64_t dim       = gmat.sizes[1];
  const auto vdt          = e->dtype;
  const size_t ves        = wholememory_dtype_get_element_size(vdt);
  const size_t ies        = wholememory_dtype_get_element_size(iarr.dtype);
  const size_t row_bytes  = static_cast<size_t>(dim) * ves;
  const bool self_local   = !e->comm->loopback;
  const int64_t all_rows  = static_cast<int64_t>(entry_offsets[W]);
  // (1) runs of equal ids. Keys bounded by the table's rows when they fit 32 bits (ids that address no row drop out of the runs);
  // wider tables sort full-width keys: ids outside every owner's range then form runs nobody asks for
  dedup_result r(env);
  r.run(idx_ptr, iarr.dtype, n, all_rows < INT64_C(0xFFFFFFFF) ? all_rows : 0, stream, 0, /*join_later=*/false);
  temp_mem host_n(env), flag_mem(env);
  auto* h_nu = static_cast<int64_t*>(host_n.pinned(1, WHOLEMEMORY_DT_INT64));
  WM_BK(bk->memcpy_async(h_nu, r.d_nunique, sizeof(int64_t), stream));
  WM_BK(bk->stream_sync(stream));   // (the dense buffer of partial sums is sized and zeroed for exactly the distinct ids)
  const int64_t nu = *h_nu;
  // (2) partial sums: row u of `partial` = fp32 sum of the gradient rows of run u, rounded once
  temp_mem partial_mem(env), iota_mem(env), long_ws(env);
  char* partial = static_cast<char*>(partial_mem.device(dim * nu, vdt));
  void* iota    = iota_mem.device(nu, iarr.dtype);
  auto* d_flag  = static_cast<int64_t*>(flag_mem.device(1, WHOLEMEMORY_DT_INT64));
  WM_BK(bk->memset_async(d_flag, 0, sizeof(int64_t), stream));
  if (nu > 0) {
    WM_BK(bk->memset_async(partial, 0, row_bytes * static_cast<size_t>(nu), stream));
    WM_BK(bk->fill_iota(iota, iarr.dtype, nu, 0, stream));
    wm_optimizer_args fa{};
    fa.type = WHOLEMEMORY_OPT_SGD, fa.lr = -1.0f, fa.weight_decay = 0.0f;
    fa.ids = iota, fa.index_dtype = iarr.dtype, fa.run_starts = r.d_starts, fa.order = r.d_order;
    fa.value_dtype = vdt, fa.grads = grads_ptr, fa.grad_stride = gmat.stride;
    fa.count = nu, fa.local_table = partial, fa.table_stride = dim, fa.local_entry_offset = 0, fa.dim = dim;
    fa.fold_mode   = 1;
    fa.long_run_ws_bytes = bk->long_run_workspace_bytes(n, dim);
    fa.long_run_ws       = long_ws.device(static_cast<int64_t>(fa.long_run_ws_bytes), WHOLEMEMORY_DT_INT8);
    if (bk->optimizer_step(&fa, r.d_nunique, stream) != 0) throw hip_error("folding the duplicate gradient rows failed");
    if (vdt == WHOLEMEMORY_DT_HALF && bk->partials_nonfinite != nullptr)
      WM_BK(bk->partials_nonfinite(r.d_starts, r.d_nunique, nu, partial, dim, dim, d_flag, stream));
  }
  // (3) owner segments of the sorted distinct ids + their exchange; the verdict on the partial rows travels with the counts
  sorted_unique su{r.d_nunique, d_flag};
  id_exchange x(env);
  bucket_and_exchange_ids(e->comm, r.d_unique, iarr.dtype, nu, entry_offsets, env, stream, &x, self_local, false, &su);
  if (x.dup_permille < 0) return 0;   // some rank's float16 partial sums left the range: everybody takes the plain route
  std::vector<int64_t> full_recv_counts = x.recv_counts, full_recv_offsets(W + 1, 0);
  full_recv_counts[rank]                = x.self_count;
  for (int i = 0; i < W; i++) full_recv_offsets[i + 1] = full_recv_offsets[i] + full_recv_counts[i];
  const int64_t n_recv = full_recv_offsets[W];
  temp_mem recv_rows(env), recv_ids_mem(env);
  char* recv_buf = static_cast<char*>(recv_rows.device(dim * n_recv, vdt));
  char* recv_ids = static_cast<char*>(recv_ids_mem.device(n_recv, iarr.dtype));
  for (int q = 0; q < W; q++) {   // ids: the peers' segments were received compactly (self cut out) — around the self slot
    const char* src = (q == rank && self_local) ? static_cast<const char*>(x.bucketed_ids) + ies * x.self_offset
                                                : static_cast<const char*>(x.recv_ids) + ies * x.recv_offsets[q];
    if (full_recv_counts[q] > 0) WM_BK(bk->memcpy_async(recv_ids + ies * full_recv_offsets[q], src, ies * full_recv_counts[q], stream));
  }
  if (self_local && x.self_count > 0)   // this rank's own partial rows: one copy into their rank-major slot
    WM_BK(bk->memcpy_async(recv_buf + row_bytes * full_recv_offsets[rank], partial + row_bytes * x.self_offset,
                           row_bytes * static_cast<size_t>(x.self_count), stream));
  {
    std::vector<int64_t> ro(full_recv_offsets.begin(), full_recv_offsets.end() - 1);
    exchange_segments(e->comm, partial, x.send_counts, x.bucket_offsets, recv_buf, x.recv_counts, ro, row_bytes, stream);
  }
  g_grad_combined_calls.fetch_add(1, std::memory_order_relaxed);
  // (4) the owner's step over at most W partial rows per id
  owner_input in;
  in.recv_ids = recv_ids, in.index_dtype = iarr.dtype, in.n_recv = n_recv, in.rows = recv_buf, in.row_stride = dim;
  *rc = owner_apply(e, in, entry_offsets, lr, env, stream, adjust_cache);
  return 1;
}

// reference embedding.cpp:146-323
wholememory_error_code_t gather_gradient_apply(wholememory_embedding_* e, wholememory_tensor_t indices,
                                               wholememory_tensor_t grads, float lr, wholememory_env_func_t* env,
                                               void* stream, bool adjust_cache = false)
{
  const auto* bk  = backend();
  auto* idesc     = wholememory_tensor_get_tensor_description(indices);
  auto* gdesc     = wholememory_tensor_get_tensor_description(grads);
  WM_CHECK_ABORT(idesc->dim == 1, "indices must be 1-D");
  if (e->optimizer == nullptr) {
    WM_ERROR("gather_gradient_apply: no optimizer set on this embedding");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  wholememory_array_description_t iarr;
  wholememory_matrix_description_t gmat;
  if (!wholememory_convert_tensor_desc_to_array(&iarr, idesc) || !wholememory_convert_tensor_desc_to_matrix(&gmat, gdesc))
    return WHOLEMEMORY_INVALID_INPUT;
  // reference: float32 tables and gradients only (exchange_embeddings_nccl_func.cu:192, embedding.cpp:61-63).
  // Extension: HALF / BF16 tables trained with SGD take gradients of the table's own dtype.
  const wholememory_dtype_t vdt = e->dtype;
  if (gmat.dtype != vdt) {
    WM_ERROR("gradients must have the embedding's dtype (float32; float16 / bfloat16 for SGD on 16-bit tables)");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  const size_t ves = wholememory_dtype_get_element_size(vdt);
  if (gmat.sizes[0] != iarr.size) return WHOLEMEMORY_INVALID_INPUT;
  auto* udesc       = wholememory_tensor_get_tensor_description(e->user);
  const int64_t dim = gmat.sizes[1];
  if (dim != udesc->sizes[1]) return WHOLEMEMORY_INVALID_INPUT;

  std::vector<size_t> entry_offsets(e->comm->world_size + 1);
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_tensor_get_entry_offsets(entry_offsets.data(), e->allocated));
  const size_t ies    = wholememory_dtype_get_element_size(iarr.dtype);
  const char* idx_ptr = static_cast<const char*>(wholememory_tensor_get_data_pointer(indices));  // data ptr already offset

  // this rank's own rows are not copied at all: the step kernels read them where the caller left them (the receive
  // positions of the self segment are remapped to caller rows after the sort). WM_GRAD_SELF_COPY=1 restores the copy.
  const char* self_copy_env = WM_AB_KNOB("WM_GRAD_SELF_COPY");
  const bool self_local     = !e->comm->loopback;  // loopback: the self segment is exchanged like a peer's
  const bool self_in_place  = self_local && bk->remap_self_order != nullptr &&
                             !(self_copy_env != nullptr && self_copy_env[0] == '1');
  // One rank, nothing to exchange: the batch IS the receive buffer, the caller's gradient tensor IS the row buffer. Ids that
  // address no row (negative "skip me" ids, ids past the table) are dropped by the sort itself — they read as one marker key
  // that sorts behind every row and whose run is not counted (backend.hpp: dedup_ids) — so no pass over the ids and no look
  // at a count by the host stands in front of the sort (that was a histogram, two small copies and a host synchronise:
  // ~45 us of a 3.2 ms call, and the host can now queue the next call while this one runs).
  std::unique_ptr<dedup_result> early;
  if (self_in_place && e->comm->single_rank_direct() && iarr.size > 0 && iarr.size < (INT64_C(1) << 31) &&
      entry_offsets[1] - entry_offsets[0] < UINT64_C(0xFFFFFFFF)) {
    early.reset(new dedup_result(env));
    early->run(idx_ptr, iarr.dtype, iarr.size, static_cast<int64_t>(entry_offsets[1]), stream,
               static_cast<int64_t>(entry_offsets[0]), /*join_later=*/true);
  }
  // several ranks and a fold that is free of the reference's order: combine this rank's duplicates before they travel?
  // (combined_gradient_apply; every condition is the same on all ranks — the op is collective)
  id_exchange x(env);
  bool ids_deferred = false;
  if (!early && !e->comm->single_rank_direct() && iarr.size < (INT64_C(1) << 31) && bk->fill_iota != nullptr &&
      bk->sorted_owner_counts != nullptr && fold_is_tree(e->optimizer, vdt)) {
    const char* sw = WM_KNOB("WM_GRAD_COMBINE");
    const int mode = sw == nullptr ? -1 : atoi(sw);   // -1 auto, 0 never, 1 always
    bool combine   = mode >= 1;
    if (mode < 0) {
      const int64_t threshold = [] {
        const char* t = WM_KNOB("WM_GRAD_COMBINE_PERMILLE");
        return t != nullptr && atoi(t) > 0 ? static_cast<int64_t>(atoi(t)) : INT64_C(100);
      }();
      // counts + duplicate estimate only; the grouping pass and the ids exchange follow once the route is known
      bucket_and_exchange_ids(e->comm, idx_ptr, iarr.dtype, iarr.size, entry_offsets, env, stream, &x, self_local, false, nullptr,
                              true, true);
      ids_deferred = true;
      combine      = x.dup_permille >= threshold;
    }
    if (combine) {
      wholememory_error_code_t rc = WHOLEMEMORY_SUCCESS;
      if (combined_gradient_apply(e, idx_ptr, iarr, wholememory_tensor_get_data_pointer(grads), gmat, entry_offsets, lr, env,
                                  stream, adjust_cache, &rc) == 1)
        return rc;
    }
  }
  if (ids_deferred) {
    finish_id_exchange(e->comm, idx_ptr, iarr.dtype, iarr.size, entry_offsets, env, stream, &x);
  } else if (early) {
    x.identity       = true;
    x.bucketed_ids   = const_cast<char*>(idx_ptr);
    x.raw_indices    = nullptr;
    x.send_counts    = {0};
    x.recv_counts    = {0};
    x.send_offsets   = {0, 0};
    x.recv_offsets   = {0, 0};
    x.bucket_offsets = {0, iarr.size};
    x.total_valid = x.self_count = iarr.size;   // upper bounds: the true number of runs stays on the device
  } else {
    bucket_and_exchange_ids(e->comm, idx_ptr, iarr.dtype, iarr.size, entry_offsets, env, stream, &x, self_local, self_in_place);
  }
  (void)ies;
  const int rank = e->comm->world_rank;

  // The owner needs ids and gradient rows of ALL requesters in rank-major receive order (that order defines
  // the fp32 summation order of duplicates). Peers' rows arrive by all-to-all-v; this rank's own rows are
  // written straight into their slot of the receive buffers (no send staging, no self copy).
  std::vector<int64_t> full_recv_counts = x.recv_counts, full_recv_offsets(e->comm->world_size + 1, 0);
  full_recv_counts[rank]                = x.self_count;
  for (int i = 0; i < e->comm->world_size; i++) full_recv_offsets[i + 1] = full_recv_offsets[i] + full_recv_counts[i];
  const int64_t n_recv = full_recv_offsets[e->comm->world_size];

  temp_mem send_rows(env), recv_rows(env), recv_ids_mem(env);
  // (x.identity: one rank, nothing dropped — the caller's ids and gradient rows are used where they are: no staging
  // buffers at all, which at 10 M x 512 B rows is 10 GB not taken from the allocator)
  auto* send_buf = static_cast<char*>(send_rows.device(x.identity ? 0 : dim * x.total_valid, vdt));
  auto* recv_buf = static_cast<char*>(recv_rows.device(x.identity ? 0 : dim * n_recv, vdt));
  const size_t row_bytes = static_cast<size_t>(dim) * ves;
  char* recv_ids = x.identity ? const_cast<char*>(idx_ptr) : static_cast<char*>(recv_ids_mem.device(n_recv, iarr.dtype));
  // ids: peers' segments were received compactly (self cut out) — place them around the self slot
  for (int r = 0; r < e->comm->world_size && !x.identity; r++) {
    const char* src = r == rank ? static_cast<const char*>(x.bucketed_ids) + ies * x.self_offset
                                : static_cast<const char*>(x.recv_ids) + ies * x.recv_offsets[r];
    if (full_recv_counts[r] > 0)
      WM_BK(bk->memcpy_async(recv_ids + ies * full_recv_offsets[r], src, ies * full_recv_counts[r], stream));
  }
  // gradient rows in bucketed order: remote segments into the send buffer, the self segment into recv_buf
  const auto grads_gref = wholememory_create_continuous_global_reference(wholememory_tensor_get_data_pointer(grads));
  auto launch_rows = [&](const int64_t* raw, int64_t s0, int64_t s1, char* dst) {   // grads[raw[s0 .. s1)] -> dst rows 0 ..
    if (s1 <= s0) return;
    wm_rows_args ga{};
    ga.gref         = grads_gref;
    ga.table_dtype  = vdt;
    ga.dim          = dim;
    ga.table_stride = gmat.stride;
    ga.indices      = raw + s0;
    ga.index_dtype  = WHOLEMEMORY_DT_INT64;
    ga.n            = s1 - s0;
    ga.plain        = dst;
    ga.plain_dtype  = vdt;
    ga.plain_stride = dim;
    ga.max_blocks   = -1;
    WM_BK(bk->gather_rows(&ga, stream));
    g_grad_exchange_launches.fetch_add(1, std::memory_order_relaxed);
  };
  const bool self_direct = x.self_count > 0 && self_in_place;
  self_rows_ref self_ref;
  if (self_direct) {
    self_ref.begin  = full_recv_offsets[rank];
    self_ref.count  = x.self_count;
    self_ref.rows   = x.raw_indices + x.self_offset;
    self_ref.grads  = wholememory_tensor_get_data_pointer(grads);
    self_ref.stride = gmat.stride;
  } else if (self_local) {
    launch_rows(x.raw_indices, x.self_offset, x.self_offset + x.self_count, recv_buf + full_recv_offsets[rank] * row_bytes);
  }
  // peers' rows: line-up (HBM) and all-to-all-v (xGMI, side stream) pipelined in C row-chunks; the id sort that
  // follows on the caller's stream overlaps with the tail of the exchange.
  // ONE line-up kernel per chunk whatever the number of ranks (round 6; the distributed gather since round 5): the positions
  // of the rows to send are brought into chunk-major order once (ops_internal.hpp: chunk_layout, backend: permute_chunks)
  // and the send buffer is laid out chunk-major — C + 1 kernels in front of the exchange instead of (W - 1) C (28 -> 5 at
  // W = 8, C = 4). The RECEIVE side keeps the rank-major order: it defines the order of the fp32 sum of duplicates.
  const int W = e->comm->world_size;
  const int C = exchange_chunks(W, x.global_moved);
  const bool per_peer = bk->permute_chunks == nullptr || W > 16 || W <= 2 /* one peer: a chunk is one range already */ ||
                        (WM_KNOB("WM_EXCHANGE_PER_PEER") != nullptr && WM_KNOB("WM_EXCHANGE_PER_PEER")[0] == '1');
  const bool folded = !per_peer && C > 1 && !x.identity;
  const chunk_layout want(x.send_counts, C), serve(x.recv_counts, C);
  temp_mem raw_cm_mem(env);
  const int64_t* send_raw = x.raw_indices;
  if (folded && x.total_send > 0) {
    auto* cm = static_cast<int64_t*>(raw_cm_mem.device(x.total_send, WHOLEMEMORY_DT_INT64));
    WM_BK(bk->permute_chunks(x.raw_indices, cm, 8, x.bucket_offsets.data(), x.send_counts.data(), W, C, stream));
    g_grad_exchange_launches.fetch_add(1, std::memory_order_relaxed);
    send_raw = cm;
  }
  void* side = C > 1 ? e->comm->get_side_stream() : stream;
  event_set lined_up(C > 1 ? C : 0), arrived(C > 1 ? 1 : 0);
  for (int c = 0; c < C; c++) {
    std::vector<int64_t> sc(W), so(W), rc(W), ro(W);
    for (int p = 0; p < W; p++) {
      sc[p] = want.count(c, p), so[p] = folded ? want.pos(c, p) : x.bucket_offsets[p] + want.first(c, p);
      rc[p] = serve.count(c, p), ro[p] = full_recv_offsets[p] + serve.first(c, p);
    }
    if (folded) {
      launch_rows(send_raw, want.start(c), want.start(c + 1), send_buf + want.start(c) * row_bytes);
    } else if (!per_peer && !x.identity) {
      // one chunk: the rows to send are the bucketed order minus this rank's own segment — the range before it and the one after
      const int64_t self_b = self_local ? x.self_offset : x.total_valid, self_e = self_local ? x.self_offset + x.self_count : x.total_valid;
      launch_rows(x.raw_indices, 0, self_b, send_buf);
      launch_rows(x.raw_indices, self_e, x.total_valid, send_buf + self_e * row_bytes);
    } else {
      for (int p = 0; p < W; p++)
        if (p != rank || !self_local) launch_rows(x.raw_indices, so[p], so[p] + sc[p], send_buf + so[p] * row_bytes);
    }
    if (C > 1) {
      WM_BK(bk->event_record(lined_up[c], stream));
      WM_BK(bk->stream_wait_event(side, lined_up[c]));
    }
    exchange_segments(e->comm, send_buf, sc, so, recv_buf, rc, ro, row_bytes, side);
  }
  if (C > 1) WM_BK(bk->event_record(arrived[0], side));
  void* rows_arrived = C > 1 ? arrived[0] : nullptr;

  // Everything this rank was given is its own (one rank; ids that address no row are dropped inside the sort, see above):
  // the receive order IS the caller's order and the caller's gradient tensor IS the receive buffer — no remapping pass
  const bool whole_input_is_self = self_direct && x.self_count == iarr.size && n_recv == iarr.size;
  owner_input in;
  in.recv_ids = recv_ids, in.index_dtype = iarr.dtype, in.n_recv = n_recv, in.rows_arrived = rows_arrived;
  if (whole_input_is_self) {
    in.rows = self_ref.grads, in.row_stride = self_ref.stride, in.sorted = early.get();
  } else {
    in.rows = recv_buf, in.row_stride = dim, in.self = self_direct ? &self_ref : nullptr;
  }
  return owner_apply(e, in, entry_offsets, lr, env, stream, adjust_cache);
}

wholememory_error_code_t remap_round_robin(wholememory_embedding_* e, wholememory_tensor_t indices, temp_mem* mapped_mem,
                                           wholememory_tensor_t* mapped, void* stream)
{
  auto* idesc = wholememory_tensor_get_tensor_description(indices);
  void* mp    = mapped_mem->device(idesc->sizes[0], idesc->dtype);
  wholememory_tensor_description_t md = *idesc;
  md.storage_offset                   = 0;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_make_tensor_from_pointer(mapped, mp, &md));
  size_t entry_start = 0;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_tensor_get_local_entry_start(&entry_start, e->allocated));
  // Deviation from the reference, on purpose: map_indices_func.cu:34-43 adds the CALLER's first row whoever owns the
  // entry (it computes the owner and drops it), so with more than one rank an id whose block lives on another rank hits
  // the wrong row. Here the row is the one wholememory_load_from_file(round_robin_size) put the entry in: every rank
  // holds the same number of rows (create_embedding pads to that), owner = (id / rr) % world. One rank: identical.
  // WM_RR_REFERENCE=1 restores the reference's caller-relative statement (rank_rows = 0 below) for call sites that were
  // written against it: ids are then only right for entries the CALLER owns ("round-robin remap").
  const bool reference_statement = [] {
    const char* v = WM_KNOB("WM_RR_REFERENCE");
    return v != nullptr && v[0] == '1';
  }();
  const int64_t rank_rows =
    reference_statement ? 0 : wholememory_tensor_get_tensor_description(e->allocated)->sizes[0] / e->comm->world_size;
  int rc = backend()->round_robin_map(wholememory_tensor_get_data_pointer(indices), mp, idesc->dtype, idesc->sizes[0],
                                      static_cast<int64_t>(entry_start), e->comm->world_size, e->round_robin_size,
                                      rank_rows, stream);
  if (rc != 0) return WHOLEMEMORY_CUDA_ERROR;
  // reference map_indices_func.cu:56-63 synchronises after the remap
  return backend()->stream_sync(stream) == 0 ? WHOLEMEMORY_SUCCESS : WHOLEMEMORY_CUDA_ERROR;
}

}  // namespace
}  // namespace wm

extern "C" int64_t wholememory_ext_combined_gradient_calls(void) { return wm::g_grad_combined_calls.load(std::memory_order_relaxed); }

extern "C" {

// ---------------------------------------------------------------- optimizers
wholememory_error_code_t wholememory_create_embedding_optimizer(wholememory_embedding_optimizer_t* optimizer,
                                                                wholememory_optimizer_type_t optimizer_type)
{
  WM_API_BEGIN
  if (optimizer == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  auto* o = new wholememory_embedding_optimizer_();
  o->type = optimizer_type;
  o->params["weight_decay"] = &o->weight_decay;
  o->params["grad_fold"]    = &o->grad_fold;
  switch (optimizer_type) {
    case WHOLEMEMORY_OPT_SGD: o->state_names = {nullptr}; break;
    case WHOLEMEMORY_OPT_LAZY_ADAM:
      o->params["epsilon"] = &o->epsilon;
      o->params["beta1"]   = &o->beta1;
      o->params["beta2"]   = &o->beta2;
      o->params["adam_w"]  = &o->adam_w;
      o->state_names       = {"m", "v", "beta12t", nullptr};
      break;
    case WHOLEMEMORY_OPT_ADAGRAD:
      o->params["epsilon"] = &o->epsilon;
      o->state_names       = {"state_sum", nullptr};
      break;
    case WHOLEMEMORY_OPT_RMSPROP:
      o->params["epsilon"] = &o->epsilon;
      o->params["alpha"]   = &o->alpha;
      o->state_names       = {"v", nullptr};
      break;
    default:
      delete o;
      return WHOLEMEMORY_NOT_IMPLEMENTED;  // reference embedding_optimizer.cpp:509-516
  }
  *optimizer = o;
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

wholememory_error_code_t wholememory_optimizer_set_parameter(wholememory_embedding_optimizer_t optimizer,
                                                             const char* parameter_name,
                                                             void* value)
{
  if (optimizer == nullptr || parameter_name == nullptr || value == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  auto it = optimizer->params.find(parameter_name);
  if (it == optimizer->params.end()) {
    WM_ERROR("optimizer has no parameter named '%s'", parameter_name);
    return WHOLEMEMORY_INVALID_INPUT;
  }
  *it->second = *static_cast<float*>(value);
  return WHOLEMEMORY_SUCCESS;
}

void wholememory_destroy_embedding_optimizer(wholememory_embedding_optimizer_t optimizer) { delete optimizer; }

// ---------------------------------------------------------------- cache policy (object only)
wholememory_error_code_t wholememory_create_embedding_cache_policy(wholememory_embedding_cache_policy_t* cache_policy,
                                                                   wholememory_comm_t cache_level_comm,
                                                                   wholememory_memory_type_t memory_type,
                                                                   wholememory_memory_location_t memory_location,
                                                                   wholememory_access_type_t access_type,
                                                                   float cache_ratio)
{
  if (cache_policy == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (cache_ratio > 1.0F || cache_ratio < 1.0F / 512) {  // reference embedding.cpp:908-912
    WM_ERROR("cache_ratio should in range [1/512, 1.0]");
    return WHOLEMEMORY_INVALID_VALUE;
  }
  auto* p                  = new wholememory_embedding_cache_policy_();
  p->cache_comm            = cache_level_comm;
  p->cache_memory_type     = memory_type;
  p->cache_memory_location = memory_location;
  p->access_type           = access_type;
  p->cache_ratio           = cache_ratio;
  *cache_policy            = p;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_destroy_embedding_cache_policy(wholememory_embedding_cache_policy_t cache_policy)
{
  delete cache_policy;
  return WHOLEMEMORY_SUCCESS;
}

// ---------------------------------------------------------------- embedding lifetime
wholememory_error_code_t wholememory_create_embedding(wholememory_embedding_t* wholememory_embedding,
                                                      wholememory_tensor_description_t* embedding_description,
                                                      wholememory_comm_t comm,
                                                      wholememory_memory_type_t memory_type,
                                                      wholememory_memory_location_t memory_location,
                                                      wholememory_embedding_cache_policy_t cache_policy,
                                                      size_t* embedding_entry_partition,
                                                      int user_defined_sms,
                                                      int round_robin_size)
{
  WM_API_BEGIN
  if (wholememory_embedding == nullptr || embedding_description == nullptr || comm == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  wholememory_matrix_description_t md;
  if (!wholememory_convert_tensor_desc_to_matrix(&md, embedding_description) || embedding_description->dim != 2) {
    WM_ERROR("wholememory_create_embedding input description must be 2D matrix");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  if (cache_policy != nullptr) {
    // reference embedding.cpp:966-1016
    if (memory_type == WHOLEMEMORY_MT_HIERARCHY) {
      WM_ERROR("Cache is not supported now in hierarchy memory type.");
      return WHOLEMEMORY_NOT_SUPPORTED;
    }
    if (cache_policy->cache_comm == comm) {
      if (cache_policy->cache_memory_location != WHOLEMEMORY_ML_DEVICE) {
        WM_ERROR("Cache has same communicator with raw embedding, should be device cached host embedding, but cache memory "
                 "location is not WHOLEMEMORY_ML_DEVICE.");
        return WHOLEMEMORY_INVALID_INPUT;
      }
      if (cache_policy->cache_memory_type < memory_type) {
        WM_ERROR("For device cached host memory, raw embedding should cover cache's address modes.");
        return WHOLEMEMORY_INVALID_INPUT;
      }
    } else {
      if (cache_policy->cache_memory_type == WHOLEMEMORY_MT_DISTRIBUTED) {
        WM_ERROR("For local cached global readonly embedding, cache_memory_type should be chunked or continuous.");
        return WHOLEMEMORY_INVALID_INPUT;
      }
      if (cache_policy->access_type != WHOLEMEMORY_AT_READONLY) {
        WM_ERROR("Only ReadOnly access type supported for local cached global readonly embedding.");
        return WHOLEMEMORY_INVALID_INPUT;
      }
    }
    embedding_entry_partition = nullptr;
  }
  if (embedding_entry_partition != nullptr) {
    if (round_robin_size != 0) WM_WARN("Parameter 'round_robin_size' is ignored.");
    round_robin_size = 0;
  }
  int W;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_communicator_get_size(&W, comm));
  std::unique_ptr<wholememory_embedding_> e(new wholememory_embedding_());
  e->comm             = comm;
  e->dtype            = md.dtype;
  e->round_robin_size = round_robin_size;
  if (round_robin_size != 0) {
    // every rank holds the same, rr-aligned share: reference embedding.cpp:467-484
    int64_t total = md.sizes[0];
    int extra     = static_cast<int>(total % (static_cast<int64_t>(W) * round_robin_size));
    if (extra > round_robin_size) extra = round_robin_size;
    int64_t first = total / (static_cast<int64_t>(W) * round_robin_size) * round_robin_size + extra;
    md.sizes[0]   = first * W;
  }
  // reference embedding.cpp:450-463
  if (user_defined_sms != -1 && (user_defined_sms <= 0 || user_defined_sms > 1568)) {
    WM_WARN("Illegal SM number for gather/scatter! Will use default size.");
    user_defined_sms = -1;
  }
  e->gather_sms = user_defined_sms;

  wholememory_tensor_description_t padded;
  wholememory_copy_matrix_desc_to_tensor(&padded, &md);
  padded.storage_offset = 0;
  padded.strides[0]     = wm::align_embedding_dim(md.sizes[1], wholememory_dtype_get_element_size(md.dtype));
  padded.strides[1]     = 1;
  WHOLEMEMORY_RETURN_ON_FAIL(
    wholememory_create_tensor(&e->allocated, &padded, comm, memory_type, memory_location, embedding_entry_partition));
  int64_t starts[2] = {0, 0};
  int64_t ends[2]   = {md.sizes[0], md.sizes[1]};
  auto rc           = wholememory_tensor_get_subtensor(e->allocated, starts, ends, &e->user);
  if (rc != WHOLEMEMORY_SUCCESS) {
    wholememory_destroy_tensor(e->allocated);
    return rc;
  }
  if (cache_policy != nullptr) {
    rc = wm::create_row_cache(&e->cache, cache_policy, e->allocated, comm);
    if (rc != WHOLEMEMORY_SUCCESS) {
      wholememory_destroy_tensor(e->user);
      wholememory_destroy_tensor(e->allocated);
      return rc;
    }
  }
  *wholememory_embedding = e.release();
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

wholememory_error_code_t wholememory_destroy_embedding(wholememory_embedding_t e)
{
  WM_API_BEGIN
  if (e == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  wm::destroy_states(e);
  delete e->cache;
  if (e->user) wholememory_destroy_tensor(e->user);
  if (e->allocated) wholememory_destroy_tensor(e->allocated);
  delete e;
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

wholememory_tensor_t wholememory_embedding_get_embedding_tensor(wholememory_embedding_t e) { return e ? e->user : nullptr; }

wholememory_error_code_t wholememory_embedding_set_optimizer(wholememory_embedding_t e,
                                                             wholememory_embedding_optimizer_t optimizer)
{
  WM_API_BEGIN
  if (e == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (e->optimizer != nullptr) {
    WM_ERROR("optimizer can only be set once.");
    return WHOLEMEMORY_NOT_SUPPORTED;
  }
  if (optimizer == nullptr) return WHOLEMEMORY_SUCCESS;
  if (e->cache != nullptr && !e->cache->writable) {
    WM_ERROR("a read-only cached embedding cannot be trained");
    return WHOLEMEMORY_NOT_SUPPORTED;
  }
  // reference embedding.cpp:61-63: "Only float embedding supports training." Extension (BASELINE config 4, fp16
  // scatter-add): HALF / BF16 embeddings accept the stateless optimizer, SGD (see backend.hpp value_dtype).
  const bool sgd16 = (e->dtype == WHOLEMEMORY_DT_HALF || e->dtype == WHOLEMEMORY_DT_BF16) &&
                     optimizer->type == WHOLEMEMORY_OPT_SGD;
  if (e->dtype != WHOLEMEMORY_DT_FLOAT && !sgd16) {
    WM_ERROR("Only float embedding supports training (float16 / bfloat16: SGD only).");
    return WHOLEMEMORY_NOT_IMPLEMENTED;
  }
  e->optimizer = optimizer;
  {
    // A trained table is written at random: the one kind of access whose speed follows WHERE the shard sits in HBM (scatter /
    // gradient apply 81 % or 67 % of the peak by placement). The probe that picks a well placed shard
    // is opt-in; say so once per process when a big device table was made without it. (No "this table is in the slow class"
    // claim: telling the classes apart needs a second candidate of the same size to compare with — that IS the probe.)
    static std::atomic<bool> told{false};
    auto h = wholememory_tensor_get_memory_handle(e->allocated);
    size_t local_bytes = 0, local_off = 0;
    void* local_ptr = nullptr;
    if (h != nullptr && wholememory_get_memory_location(h) == WHOLEMEMORY_ML_DEVICE && !wholememory_ext_handle_was_probed(h) &&
        wholememory_get_local_memory(&local_ptr, &local_bytes, &local_off, h) == WHOLEMEMORY_SUCCESS &&
        local_bytes >= (static_cast<size_t>(1) << 30) && !told.exchange(true))
      WM_WARN("an optimizer is attached to a %.1f GiB device table that was allocated without the placement probe: scatter and "
              "gradient apply on it run at either of two levels (about 20 %% apart) depending on where the allocation landed. "
              "create_embedding(..., placement_probe=\"auto\") / wholememory_ext_set_malloc_probe(\"auto\") / "
              "WM_MALLOC_PROBE=auto picks a well placed shard (costs a few transient candidate allocations)",
              local_bytes / 1073741824.0);
  }
  auto rc      = wm::create_states(e);
  if (rc != WHOLEMEMORY_SUCCESS) {
    wm::destroy_states(e);
    e->optimizer = nullptr;
  }
  return rc;
  WM_API_END
}

// ---------------------------------------------------------------- hot path
wholememory_error_code_t wholememory_embedding_gather(wholememory_embedding_t e,
                                                      wholememory_tensor_t indices,
                                                      wholememory_tensor_t output,
                                                      bool adjust_cache,
                                                      wholememory_env_func_t* p_env_fns,
                                                      int64_t stream_int)
{
  WM_API_BEGIN
  if (e == nullptr || indices == nullptr || output == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  void* stream = reinterpret_cast<void*>(stream_int);
  auto do_gather = [&](wholememory_tensor_t ids) {
    return e->cache != nullptr
             ? wm::gather_cached(e->allocated, ids, output, p_env_fns, stream, e->gather_sms, e->cache, adjust_cache)
             : wholememory_gather(e->allocated, ids, output, p_env_fns, stream, e->gather_sms);
  };
  if (e->round_robin_size == 0) return do_gather(indices);
  wm::temp_mem mapped_mem(p_env_fns);
  wholememory_tensor_t mapped = nullptr;
  WHOLEMEMORY_RETURN_ON_FAIL(wm::remap_round_robin(e, indices, &mapped_mem, &mapped, stream));
  auto rc = do_gather(mapped);
  wholememory_destroy_tensor(mapped);
  return rc;
  WM_API_END
}

wholememory_error_code_t wholememory_embedding_gather_gradient_apply(wholememory_embedding_t e,
                                                                     wholememory_tensor_t indices,
                                                                     wholememory_tensor_t grads,
                                                                     bool adjust_cache,
                                                                     float lr,
                                                                     wholememory_env_func_t* p_env_fns,
                                                                     int64_t stream_int)
{
  WM_API_BEGIN
  if (e == nullptr || indices == nullptr || grads == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  void* stream = reinterpret_cast<void*>(stream_int);
  if (e->cache != nullptr && !e->cache->same_comm) return WHOLEMEMORY_NOT_SUPPORTED;  // local caches are read-only
  if (e->round_robin_size == 0) return wm::gather_gradient_apply(e, indices, grads, lr, p_env_fns, stream, adjust_cache);
  wm::temp_mem mapped_mem(p_env_fns);
  wholememory_tensor_t mapped = nullptr;
  WHOLEMEMORY_RETURN_ON_FAIL(wm::remap_round_robin(e, indices, &mapped_mem, &mapped, stream));
  auto rc = wm::gather_gradient_apply(e, mapped, grads, lr, p_env_fns, stream, adjust_cache);
  wholememory_destroy_tensor(mapped);
  return rc;
  WM_API_END
}

const char* const* wholememory_embedding_get_optimizer_state_names(wholememory_embedding_t e)
{
  static const char* const kNone[] = {nullptr};
  if (e == nullptr || e->optimizer == nullptr) return kNone;  // reference returns nullptr-terminated list
  return e->optimizer->state_names.data();
}

wholememory_tensor_t wholememory_embedding_get_optimizer_state(wholememory_embedding_t e, const char* name)
{
  if (e == nullptr || name == nullptr) return nullptr;
  auto it = e->state_views.find(name);
  return it == e->state_views.end() ? nullptr : it->second;
}

wholememory_error_code_t wholememory_embedding_writeback_cache(wholememory_embedding_t e, int64_t stream_int)
{
  WM_API_BEGIN
  if (e == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (e->cache == nullptr) return WHOLEMEMORY_SUCCESS;  // reference embedding.cpp:1090-1098: nothing to do without a cache
  return wm::row_cache_writeback(e->cache, false, reinterpret_cast<void*>(stream_int));
  WM_API_END
}
wholememory_error_code_t wholememory_embedding_drop_all_cache(wholememory_embedding_t e, int64_t stream_int)
{
  WM_API_BEGIN
  if (e == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (e->cache == nullptr) return WHOLEMEMORY_SUCCESS;
  return wm::row_cache_writeback(e->cache, true, reinterpret_cast<void*>(stream_int));
  WM_API_END
}

wholememory_error_code_t wholememory_ext_embedding_cache_info(wholememory_embedding_t e, int64_t* slots, int64_t* occupied,
                                                              int64_t* dirty, int64_t* hits, int64_t* lookups)
{
  WM_API_BEGIN
  if (e == nullptr || e->cache == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  return wm::row_cache_info(e->cache, slots, occupied, dirty, hits, lookups, nullptr);
  WM_API_END
}

// ---------------------------------------------------------------- raw stage for parity tests
wholememory_error_code_t wholememory_ext_dedup_apply(const void* recv_ids,
                                                     wholememory_dtype_t index_dtype,
                                                     int64_t n_recv,
                                                     const float* recv_grads,
                                                     int64_t grad_stride,
                                                     int64_t dim,
                                                     float* local_table,
                                                     int64_t table_stride,
                                                     int64_t local_entry_offset,
                                                     int64_t local_entry_count,
                                                     wholememory_optimizer_type_t opt_type,
                                                     const float* opt_params,
                                                     float lr,
                                                     float* per_element_state,
                                                     float* per_row_state,
                                                     int64_t* n_unique_host,
                                                     wholememory_env_func_t* p_env_fns,
                                                     void* stream)
{
  WM_API_BEGIN
  if (opt_params == nullptr || local_table == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  wholememory_embedding_optimizer_ o;
  o.type         = opt_type;
  o.weight_decay = opt_params[0];
  o.epsilon      = opt_params[1];
  o.beta1        = opt_params[2];
  o.beta2        = opt_params[3];
  o.alpha        = opt_params[4];
  o.adam_w       = opt_params[5];
  wm_optimizer_args oa{};
  wm::fill_optimizer_args(&oa, &o, lr);
  oa.local_table        = local_table;
  oa.table_stride       = table_stride;
  oa.local_entry_offset = local_entry_offset;
  oa.dim                = dim;
  oa.per_element_state  = per_element_state;
  oa.per_element_stride = table_stride * wm::per_element_state_count(opt_type);
  oa.per_row_state      = per_row_state;
  int64_t nu            = 0;
  wm::dedup_and_step(recv_ids, index_dtype, n_recv, recv_grads, grad_stride, &oa, local_entry_offset + local_entry_count,
                     p_env_fns, stream, &nu, nullptr, nullptr, local_entry_offset);
  if (n_unique_host) *n_unique_host = nu;
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

}  // extern "C"
