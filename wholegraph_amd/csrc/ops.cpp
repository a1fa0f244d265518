// wholegraph_amd — wholememory_gather / wholememory_scatter: validation, dispatch and the
// DISTRIBUTED orchestration (host side). Kernels live behind backend.hpp.
//
// Reference: cpp/src/wholememory_ops/gather_op.cpp:23-131, gather_op_impl_mapped.cu:29-78,
// gather_op_impl_nccl.cu:34-182, scatter_op.cpp:23-110, scatter_op_impl_nccl.cu:34-181,
// functions/exchange_ids_nccl_func.cu:157-226.
//
// DISTRIBUTED gather on MI355X (one process per GPU, RCCL over xGMI):
//   1 bucket ids by owner (stable multisplit kernel: counts + grouped ids + raw positions)
//   2 counts -> pinned host, ONE stream sync, counts all-to-all (the reference pays two syncs and a
//     staged host_alltoall, exchange_ids_nccl_func.cu:194-207)
//   3 ids all-to-all-v (grouped ncclSend/ncclRecv on the caller's stream)
//   4 owner gathers its rows straight into the send buffer, casting to the output dtype there
//   5 rows all-to-all-v
//   6 reorder-on-receive: out[raw_indices[j]] = recv[j]
// DISTRIBUTED scatter mirrors it (rows travel in the input dtype, the owner casts on write) and ends
// with a stream synchronise like the reference (scatter_op_impl_nccl.cu:168).
#include "knobs.hpp"
#include "ops_internal.hpp"
#include "embedding_cache.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>

namespace wm {

#define WM_BK(call)                                                                                  \
  do {                                                                                               \
    int rc__ = (call);                                                                               \
    if (rc__ != 0) throw ::wm::hip_error(::wm::format_string("%s failed with code %d", #call, rc__)); \
  } while (0)

// ------------------------------------------------------------------------------------------------
temp_mem::temp_mem(wholememory_env_func_t* env) : env_(env)
{
  WM_CHECK(env != nullptr, "p_env_fns must not be null");
  env_->temporary_fns.create_memory_context_fn(&ctx_, env_->temporary_fns.global_context);
}
temp_mem::~temp_mem()
{
  if (ptr_ != nullptr) env_->temporary_fns.free_fn(ctx_, env_->temporary_fns.global_context);
  env_->temporary_fns.destroy_memory_context_fn(ctx_, env_->temporary_fns.global_context);
}
void* temp_mem::alloc(int64_t elt_count, wholememory_dtype_t dtype, wholememory_memory_allocation_type_t type)
{
  WM_CHECK(ptr_ == nullptr, "temp_mem slot reused");
  wholememory_tensor_description_t d;
  wholememory_initialize_tensor_desc(&d);
  d.dim      = 1;
  d.sizes[0] = elt_count > 0 ? elt_count : 1;  // never hand a zero-size request to the host allocator
  d.dtype    = dtype;
  ptr_       = env_->temporary_fns.malloc_fn(&d, type, ctx_, env_->temporary_fns.global_context);
  if (ptr_ == nullptr) throw std::bad_alloc();
  return ptr_;
}

// ------------------------------------------------------------------------------------------------
void bucket_and_exchange_ids(wholememory_comm_t comm, const void* indices, wholememory_dtype_t index_dtype, int64_t n,
                             const std::vector<size_t>& entry_offsets, wholememory_env_func_t* env, void* stream,
                             id_exchange* x, bool keep_self_local, bool allow_identity, const sorted_unique* sorted,
                             bool estimate_duplicates, bool defer_ids)
{
  const auto* bk = backend();
  const int W    = comm->world_size;
  const size_t ies = wholememory_dtype_get_element_size(index_dtype);
  x->send_counts.assign(W, 0);
  x->recv_counts.assign(W, 0);
  x->send_offsets.assign(W + 1, 0);
  x->recv_offsets.assign(W + 1, 0);

  // entry_offsets describes `owners` row ranges. Normally owners == W (one bucket per rank of `comm`); with more
  // owners than ranks an id of owner o travels to rank o % W (first hop of the HIERARCHY gather)
  const int owners = static_cast<int>(entry_offsets.size()) - 1;
  WM_CHECK(owners >= W, "bucket_and_exchange_ids: fewer row ranges than ranks");
  temp_mem dev_offsets(env), dev_counts(env), workspace(env), workspace2(env), host_counts(env), est_ws(env);
  // defer_ids: the device copy of the offsets and the bucketing workspace (which keeps the scanned block counts of the
  // counts-only pass) outlive this call inside `x`, so that finish_id_exchange() neither uploads the offsets a second time
  // (its pinned staging buffer used to be released with the copy still queued) nor repeats the histogram pass
  const bool keep_for_finish = defer_ids && sorted == nullptr;
  auto* d_off = static_cast<uint64_t*>((keep_for_finish ? x->aux_offsets : dev_offsets).device(owners + 1, WHOLEMEMORY_DT_INT64));
  auto* d_cnt = static_cast<int64_t*>(dev_counts.device(W + 1, WHOLEMEMORY_DT_INT64));  // [W] = the duplicate estimate
  auto* h_cnt = static_cast<int64_t*>(host_counts.pinned(W + 1 + owners + 1, WHOLEMEMORY_DT_INT64));
  // stage the offsets through pinned memory so the H2D copy is truly asynchronous
  uint64_t* h_off = reinterpret_cast<uint64_t*>(h_cnt + W + 1);
  for (int i = 0; i <= owners; i++) h_off[i] = entry_offsets[i];
  WM_BK(bk->memcpy_async(d_off, h_off, sizeof(uint64_t) * (owners + 1), stream));

  if (allow_identity && keep_self_local && W == 1 && owners == 1) {
    // one rank: the only thing bucketing could do is drop negative ids — count first (histogram only), and when there is
    // none leave the caller's array where it is
    wm_bucket_args ca{};
    ca.indices       = indices;
    ca.index_dtype   = index_dtype;
    ca.n             = n;
    ca.entry_offsets = d_off;
    ca.world_size    = 1;
    ca.counts        = d_cnt;
    ca.workspace     = workspace.device(static_cast<int64_t>(bk->bucket_workspace_bytes(n, 1)), WHOLEMEMORY_DT_INT8);
    WM_BK(bk->bucket_ids(&ca, stream));
    WM_BK(bk->memcpy_async(h_cnt, d_cnt, sizeof(int64_t), stream));
    WM_BK(bk->stream_sync(stream));
    if (h_cnt[0] == n) {
      x->identity       = true;
      x->bucketed_ids   = const_cast<void*>(indices);
      x->raw_indices    = nullptr;
      x->bucket_offsets = {0, n};
      x->total_valid = x->self_count = n;
      x->self_offset = x->total_send = x->total_recv = x->global_moved = 0;
      return;
    }
  }
  // the duplicate estimate of this rank's ids travels with the counts (slot W), so that every rank takes the same
  // de-duplication decision from the same W numbers
  // (a rank whose batch cannot be de-duplicated — 2^31 ids or more — publishes -1, which vetoes the decision for everybody)
  // Small batches are latency-bound and a de-duplicated exchange could not pay for its sort: they skip the estimate (a
  // 16 MiB flag array to clear and count) and publish 0 — a vote for "as they are", not a veto; the other ranks' estimates
  // still decide. WM_GATHER_DEDUP_MIN_IDS moves the limit.
  const int64_t estimate_min_ids = [] {
    const char* e = WM_KNOB("WM_GATHER_DEDUP_MIN_IDS");
    return e != nullptr && atoll(e) >= 0 ? static_cast<int64_t>(atoll(e)) : (INT64_C(1) << 18);
  }();
  const bool estimate = estimate_duplicates && bk->dup_estimate != nullptr && n >= std::max<int64_t>(estimate_min_ids, 2) &&
                        n < (INT64_C(1) << 31);
  if (sorted != nullptr && sorted->vote_dev != nullptr) {
    WM_BK(bk->memcpy_async(d_cnt + W, sorted->vote_dev, sizeof(int64_t), stream));   // the caller's vote rides in the spare slot
  } else if (estimate) {
    void* ws = est_ws.device(static_cast<int64_t>(bk->dup_estimate_workspace_bytes(n)), WHOLEMEMORY_DT_INT8);
    WM_BK(bk->dup_estimate(indices, index_dtype, n, ws, d_cnt + W, stream));
  } else {
    WM_BK(bk->memset_async(d_cnt + W, estimate_duplicates && n >= (INT64_C(1) << 31) ? 0xff : 0, sizeof(int64_t), stream));
  }
  if (sorted != nullptr) {
    // sorted, distinct ids: owners hold contiguous id ranges, so the owner segments are contiguous pieces of the array as
    // it stands — W pairs of binary searches instead of a multisplit pass, no copy, no position array
    WM_CHECK(owners == W && bk->sorted_owner_counts != nullptr, "sorted ids need one range per rank");
    x->presorted    = true;
    x->bucketed_ids = const_cast<void*>(indices);
    x->raw_indices  = nullptr;
    WM_BK(bk->sorted_owner_counts(indices, index_dtype, sorted->n_dev, n, d_off, W, d_cnt, stream));
  } else {
    // defer_ids: only the per-owner counts for now (histogram pass); the grouping pass and the ids exchange follow in
    // finish_id_exchange() once the caller has looked at the counts / the duplicate estimate
    x->bucketed_ids = defer_ids ? nullptr : x->bucketed_mem.device(n, index_dtype);
    x->raw_indices  = defer_ids ? nullptr : static_cast<int64_t*>(x->raw_mem.device(n, WHOLEMEMORY_DT_INT64));
    wm_bucket_args ba{};
    ba.indices       = indices;
    ba.index_dtype   = index_dtype;
    ba.n             = n;
    ba.entry_offsets = d_off;
    ba.world_size    = W;
    ba.owner_count   = owners == W ? 0 : owners;
    ba.counts        = d_cnt;
    ba.bucketed_ids  = x->bucketed_ids;
    ba.raw_indices   = x->raw_indices;
    ba.workspace     = (keep_for_finish ? x->aux_ws : workspace2).device(static_cast<int64_t>(bk->bucket_workspace_bytes(n, W)), WHOLEMEMORY_DT_INT8);
    WM_BK(bk->bucket_ids(&ba, stream));
  }
  // counts: with RCCL the W x W matrix is all-gathered on the caller's stream straight from the device counters and the
  // host waits ONCE (the reference pays a sync for its own counts and a second one inside the staged host_alltoall,
  // exchange_ids_nccl_func.cu:194-207); providers without a device all-gather take the host route
  temp_mem dev_matrix(env), host_matrix(env);
  auto* d_mat = static_cast<int64_t*>(dev_matrix.device(static_cast<int64_t>(W) * (W + 1), WHOLEMEMORY_DT_INT64));
  auto* h_mat = static_cast<int64_t*>(host_matrix.pinned(static_cast<int64_t>(W) * (W + 1), WHOLEMEMORY_DT_INT64));
  std::vector<int64_t> estimates(W, 0);
  if (!comm->alltoall_counts_device(d_cnt, d_mat, h_mat, stream, x->send_counts.data(), x->recv_counts.data(),
                                    &x->global_moved, estimates.data())) {
    WM_BK(bk->memcpy_async(h_cnt, d_cnt, sizeof(int64_t) * (W + 1), stream));
    WM_BK(bk->stream_sync(stream));
    for (int i = 0; i < W; i++) x->send_counts[i] = h_cnt[i];
    estimates[0] = h_cnt[W];
    comm->alltoall_host_i64(x->send_counts.data(), x->recv_counts.data(), &x->global_moved, estimates.data());
  }
  x->dup_permille = 0;
  bool veto       = false;
  for (int i = 0; i < W; i++) {
    x->dup_permille += estimates[i];
    veto = veto || estimates[i] < 0;
  }
  x->dup_permille = veto ? -1 : x->dup_permille / W;
  // bucketed layout (all owners, self included) — positions into bucketed_ids / raw_indices
  x->bucket_offsets.assign(W + 1, 0);
  for (int i = 0; i < W; i++) x->bucket_offsets[i + 1] = x->bucket_offsets[i] + x->send_counts[i];
  x->total_valid = x->bucket_offsets[W];
  x->self_count  = x->send_counts[comm->world_rank];
  x->self_offset = x->bucket_offsets[comm->world_rank];
  if (keep_self_local) {
    // ids this rank owns itself never enter the exchange: the caller serves them straight from / to
    // the local shard. The wire layout below is the bucketed layout with the self segment cut out.
    x->send_counts[comm->world_rank] = 0;
    x->recv_counts[comm->world_rank] = 0;
  }
  for (int i = 0; i < W; i++) {
    x->send_offsets[i + 1] = x->send_offsets[i] + x->send_counts[i];
    x->recv_offsets[i + 1] = x->recv_offsets[i] + x->recv_counts[i];
  }
  x->total_send = x->send_offsets[W];
  x->total_recv = x->recv_offsets[W];
  if (defer_ids && sorted == nullptr) return;
  x->recv_ids   = x->recv_mem.device(x->total_recv, index_dtype);
  exchange_segments(comm, x->bucketed_ids, x->send_counts, x->bucket_offsets, x->recv_ids, x->recv_counts,
                    x->recv_offsets, ies, stream);
  if (debug_sync_enabled()) WM_BK(bk->stream_sync(stream));
}

// second half of a bucket_and_exchange_ids(..., defer_ids = true): group the ids by owner and exchange them (the counts
// are known and exchanged already)
void finish_id_exchange(wholememory_comm_t comm, const void* indices, wholememory_dtype_t index_dtype, int64_t n,
                        const std::vector<size_t>& entry_offsets, wholememory_env_func_t* env, void* stream, id_exchange* x)
{
  const auto* bk   = backend();
  const int W      = comm->world_size;
  const size_t ies = wholememory_dtype_get_element_size(index_dtype);
  const int owners = static_cast<int>(entry_offsets.size()) - 1;
  (void)env;
  (void)owners;
  // the offsets on the device and the scanned block counts of the counts-only pass are still in `x`
  auto* d_off = static_cast<uint64_t*>(x->aux_offsets.get());
  WM_CHECK(d_off != nullptr && x->aux_ws.get() != nullptr, "finish_id_exchange without a deferred bucket_and_exchange_ids");
  x->bucketed_ids = x->bucketed_mem.device(n, index_dtype);
  x->raw_indices  = static_cast<int64_t*>(x->raw_mem.device(n, WHOLEMEMORY_DT_INT64));
  wm_bucket_args ba{};
  ba.indices       = indices;
  ba.index_dtype   = index_dtype;
  ba.n             = n;
  ba.entry_offsets = d_off;
  ba.world_size    = W;
  ba.owner_count   = owners == W ? 0 : owners;
  ba.counts        = static_cast<int64_t*>(x->aux_counts.device(W + 1, WHOLEMEMORY_DT_INT64));
  ba.bucketed_ids  = x->bucketed_ids;
  ba.raw_indices   = x->raw_indices;
  ba.workspace     = x->aux_ws.get();
  ba.reuse_scan    = 1;
  WM_BK(bk->bucket_ids(&ba, stream));
  x->recv_ids = x->recv_mem.device(x->total_recv, index_dtype);
  exchange_segments(comm, x->bucketed_ids, x->send_counts, x->bucket_offsets, x->recv_ids, x->recv_counts,
                    x->recv_offsets, ies, stream);
}

void exchange_segments(wholememory_comm_t comm, const void* send, const std::vector<int64_t>& send_counts,
                       const std::vector<int64_t>& send_offsets, void* recv, const std::vector<int64_t>& recv_counts,
                       const std::vector<int64_t>& recv_offsets, size_t row_bytes, void* stream)
{
  const int W = comm->world_size;
  std::vector<size_t> sb(W), sd(W), rb(W), rd(W);
  for (int i = 0; i < W; i++) {
    sb[i] = static_cast<size_t>(send_counts[i]) * row_bytes;
    rb[i] = static_cast<size_t>(recv_counts[i]) * row_bytes;
    sd[i] = static_cast<size_t>(send_offsets[i]) * row_bytes;
    rd[i] = static_cast<size_t>(recv_offsets[i]) * row_bytes;
  }
  comm->alltoallv_device(send, sb.data(), sd.data(), recv, rb.data(), rd.data(), stream);
}

void exchange_rows(wholememory_comm_t comm, const void* send, const std::vector<int64_t>& send_counts, void* recv,
                   const std::vector<int64_t>& recv_counts, size_t row_bytes, void* stream)
{
  const int W = comm->world_size;
  std::vector<size_t> sb(W), sd(W), rb(W), rd(W);
  size_t so = 0, ro = 0;
  for (int i = 0; i < W; i++) {
    sb[i] = static_cast<size_t>(send_counts[i]) * row_bytes;
    rb[i] = static_cast<size_t>(recv_counts[i]) * row_bytes;
    sd[i] = so, rd[i] = ro;
    so += sb[i], ro += rb[i];
  }
  comm->alltoallv_device(send, sb.data(), sd.data(), recv, rb.data(), rd.data(), stream);
}

std::vector<size_t> entry_offsets_of(wholememory_handle_t handle, size_t entry_bytes)
{
  wholememory_comm_t comm;
  WM_CHECK(wholememory_get_communicator(&comm, handle) == WHOLEMEMORY_SUCCESS, "handle has no communicator");
  std::vector<size_t> off(comm->world_size + 1);
  WM_CHECK(wholememory_get_rank_partition_offsets(off.data(), handle) == WHOLEMEMORY_SUCCESS, "partition offsets");
  for (auto& o : off) {
    // reference gather_op_impl_nccl.cu:83-93 (NOTHROW check -> abort)
    WM_CHECK_ABORT(o % entry_bytes == 0, "embedding memory offset %zu is not a multiple of the row size %zu", o, entry_bytes);
    o /= entry_bytes;
  }
  return off;
}

wholememory_gref_t local_shard_gref(wholememory_handle_t handle)
{
  void* p = nullptr;
  size_t sz, off;
  WM_CHECK(wholememory_get_local_memory(&p, &sz, &off, handle) == WHOLEMEMORY_SUCCESS, "local memory");
  // "fake" flat base so that GLOBAL row ids address the local shard (reference gather_op_impl_nccl.cu:122-126)
  return wholememory_create_continuous_global_reference(static_cast<char*>(p) - off);
}

namespace {
wholememory_error_code_t mapped_gref(wholememory_tensor_t t, wholememory_gref_t* gref);
}
wholememory_error_code_t tensor_mapped_gref(wholememory_tensor_t t, wholememory_gref_t* gref) { return mapped_gref(t, gref); }

// WM_MAPPED_VIA_EXCHANGE=1: CHUNKED / CONTINUOUS tables shared by several ranks are served by the explicit
// bucket -> all-to-all-v -> owner-side kernel route of the DISTRIBUTED type instead of loads / stores through the peer
// mappings. The op then becomes COLLECTIVE over the table's communicator (every rank must call), which is why it is
// opt-in: by default mapped gathers / scatters stay rank-local and asynchronous like the reference's.
bool mapped_via_exchange(wholememory_tensor_t t, wholememory_memory_type_t mt)
{
  if (mt != WHOLEMEMORY_MT_CHUNKED && mt != WHOLEMEMORY_MT_CONTINUOUS) return false;
  const char* e = WM_KNOB("WM_MAPPED_VIA_EXCHANGE");
  if (e == nullptr || e[0] != '1') return false;
  wholememory_comm_t comm;
  if (wholememory_get_communicator(&comm, wholememory_tensor_get_memory_handle(t)) != WHOLEMEMORY_SUCCESS) return false;
  return comm->world_size > 1;
}

// Every rank of the exchange must arrive at the same number (each chunk is one collective call), so the decision may only
// use what all ranks know alike: the world size, the environment and id_exchange::global_moved.
extern std::atomic<int64_t> g_dist_gather_launches;   // kernels queued by gather_distributed_rows (defined with the other counters)
extern std::atomic<int64_t> g_dist_scatter_launches;  // ... by scatter_distributed

int exchange_chunks(int world_size, int64_t global_moved)
{
  const char* e = WM_KNOB("WM_EXCHANGE_CHUNKS");
  if (e != nullptr && atoi(e) >= 1) return std::min(atoi(e), 16);
  if (world_size <= 1) return 1;
  // below ~256 k rows in and out of the average rank the exchange is latency-bound and extra launches only add overhead
  return 2 * global_moved / world_size >= (1 << 18) ? 4 : 1;
}

event_set::event_set(int n) : events_(n, nullptr)
{
  for (auto& e : events_)
    if (backend()->event_create(&e) != 0) throw hip_error("event_create failed");
}
event_set::~event_set()
{
  for (auto e : events_)
    if (e != nullptr) backend()->event_destroy(e);
}

namespace {

struct op_descs {
  wholememory_matrix_description_t table;
  wholememory_array_description_t indices;
  wholememory_matrix_description_t plain;
  void* indices_ptr;
  void* plain_ptr;
};

// shared argument validation of gather_op.cpp:36-86 / scatter_op.cpp:36-88
wholememory_error_code_t check_args(wholememory_tensor_t wm_tensor, wholememory_tensor_t indices_tensor,
                                    wholememory_tensor_t plain_tensor, const char* plain_name, op_descs* d)
{
  if (wm_tensor == nullptr || indices_tensor == nullptr || plain_tensor == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  auto td = *wholememory_tensor_get_tensor_description(wm_tensor);
  if (td.dim != 1 && td.dim != 2) {
    WM_ERROR("wholememory_tensor should be 1D or 2D tensor.");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  if (td.dim == 1 && !wholememory_unsqueeze_tensor(&td, 1)) return WHOLEMEMORY_LOGIC_ERROR;
  if (!wholememory_convert_tensor_desc_to_matrix(&d->table, &td)) return WHOLEMEMORY_LOGIC_ERROR;
  if (wholememory_tensor_get_tensor_description(indices_tensor)->dim != 1) {
    WM_ERROR("indices tensor should be 1D tensor");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  auto pd = *wholememory_tensor_get_tensor_description(plain_tensor);
  if (pd.dim != wholememory_tensor_get_tensor_description(wm_tensor)->dim) {
    WM_ERROR("%s tensor should be same dim as wholememory_tensor.", plain_name);
    return WHOLEMEMORY_INVALID_INPUT;
  }
  if (pd.dim == 1 && !wholememory_unsqueeze_tensor(&pd, 1)) return WHOLEMEMORY_LOGIC_ERROR;
  if (!wholememory_convert_tensor_desc_to_array(&d->indices, wholememory_tensor_get_tensor_description(indices_tensor))) {
    WM_ERROR("Convert indices tensor to array failed.");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  if (!wholememory_convert_tensor_desc_to_matrix(&d->plain, &pd)) {
    WM_ERROR("Convert %s tensor to matrix failed.", plain_name);
    return WHOLEMEMORY_INVALID_INPUT;
  }
  d->indices_ptr = wholememory_tensor_get_data_pointer(indices_tensor);
  d->plain_ptr   = wholememory_tensor_get_data_pointer(plain_tensor);
  // functions/gather_func.cu:72-105 / scatter_func.cu: dtype rules
  if (d->indices.dtype != WHOLEMEMORY_DT_INT && d->indices.dtype != WHOLEMEMORY_DT_INT64) {
    WM_ERROR("indices must be int32 or int64");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  if (wholememory_dtype_is_floating_number(d->table.dtype) != wholememory_dtype_is_floating_number(d->plain.dtype)) {
    WM_ERROR("embedding and %s must both be floating point or both be integer", plain_name);
    return WHOLEMEMORY_LOGIC_ERROR;  // reference gather_func.cu:79-81 throws logic_error
  }
  if (d->plain.sizes[0] < d->indices.size) {
    WM_ERROR("%s rows (%ld) < indices count (%ld)", plain_name, static_cast<long>(d->plain.sizes[0]),
             static_cast<long>(d->indices.size));
    return WHOLEMEMORY_INVALID_INPUT;
  }
  if (d->plain.sizes[1] != d->table.sizes[1]) {
    WM_ERROR("%s columns (%ld) != embedding columns (%ld)", plain_name, static_cast<long>(d->plain.sizes[1]),
             static_cast<long>(d->table.sizes[1]));
    return WHOLEMEMORY_INVALID_INPUT;
  }
  return WHOLEMEMORY_SUCCESS;
}

void fill_rows_args(wm_rows_args* a, const wholememory_gref_t& gref, const wholememory_matrix_description_t& table,
                    const void* indices, wholememory_dtype_t index_dtype, int64_t n, void* plain,
                    const wholememory_matrix_description_t& plain_desc, int max_blocks)
{
  a->gref                 = gref;
  a->table_dtype          = table.dtype;
  a->dim                  = table.sizes[1];
  a->table_stride         = table.stride;
  a->table_storage_offset = table.storage_offset;
  a->indices              = indices;
  a->index_dtype          = index_dtype;
  a->n                    = n;
  a->row_map              = nullptr;
  a->plain                = plain;
  a->plain_dtype          = plain_desc.dtype;
  a->plain_stride         = plain_desc.stride;
  // `plain` comes from wholememory_tensor_get_data_pointer(), which has ALREADY applied the
  // descriptor's storage_offset. The reference adds it a second time inside its kernels
  // (gather_scatter_func.cuh:293,564 on top of wholememory_tensor.cpp:290-293) — harmless there only
  // because every caller passes offset 0 (wholegraph_env.py:173-182). Applied once here.
  a->plain_storage_offset = 0;
  a->max_blocks           = max_blocks;
}

// gref a kernel should use for a mapped tensor: with a single rank everything is one flat block
wholememory_error_code_t mapped_gref(wholememory_tensor_t t, wholememory_gref_t* gref)
{
  if (wholememory_tensor_has_handle(t)) {
    auto h = wholememory_tensor_get_memory_handle(t);
    wholememory_comm_t comm;
    WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_communicator(&comm, h));
    if (comm->world_size == 1) {
      void* p;
      size_t sz, off;
      WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_local_memory(&p, &sz, &off, h));
      *gref = wholememory_create_continuous_global_reference(p);
      return WHOLEMEMORY_SUCCESS;
    }
  }
  return wholememory_tensor_get_global_reference(t, gref);
}

// the exchange normally runs over the handle's communicator with its per-rank row ranges; a `route` substitutes another
// communicator and another set of ranges over the same local shard (second hop of the HIERARCHY gather)
struct route {
  wholememory_comm_t comm;
  std::vector<size_t> offsets;  // comm->world_size + 1 row offsets
};
wholememory_error_code_t gather_distributed_rows(wholememory_handle_t handle, const op_descs& d,
                                                 wholememory_env_func_t* env, void* stream, int gather_sms,
                                                 row_cache* cache = nullptr, bool adjust_cache = false,
                                                 const route* via = nullptr, id_exchange* prepared = nullptr);

// Request de-duplication (not in the reference): a skewed batch asks for the same hot rows over and over — Zipf(1.05),
// 10 M ids: 49 % unique — and every copy would cross xGMI. The requester then sorts its ids (radix sort + run detection,
// the gradient path's primitive), fetches each DISTINCT row once and expands locally: out[i] = fetched[run of i], one more
// pass over the output in HBM, which a link-bound multi-GPU step hides many times over.
//  * sorted distinct ids are already grouped by owner (owners hold contiguous id ranges): no multisplit pass, the owner
//    segments are found by W pairs of binary searches, and the rows are received straight into their place of the dense
//    [distinct, dim] buffer — no reorder-on-receive pass either;
//  * WHETHER to do it is decided per call from the batch itself: every rank estimates the duplicate share of a sample of its
//    ids (linear counting over 16 MiB of byte flags, ~25 us; batches under 2^18 ids skip it and vote 0) next to the bucketing pass it runs anyway, the estimates ride along with
//    the counts exchange, and all ranks apply the same rule to the same W numbers (mean >= WM_GATHER_DEDUP_PERMILLE, default
//    100 = 10 % duplicates in the sample) — so the collectives stay matched. Uniform batches pay only for the estimate.
//    WM_GATHER_DEDUP=0 / 1 forces the choice (2: also on a single rank, to measure).
wholememory_error_code_t gather_distributed_dedup(wholememory_handle_t handle, const op_descs& d, wholememory_env_func_t* env,
                                                  void* stream, int gather_sms, wholememory_comm_t comm,
                                                  const std::vector<size_t>& entry_offsets)
{
  const auto* bk    = backend();
  const int64_t n   = d.indices.size;
  const int64_t dim = d.table.sizes[1];
  temp_mem unique_ids(env), run_starts(env), order(env), n_unique(env), ws(env), rows(env), inverse(env);
  void* d_unique  = unique_ids.device(n, d.indices.dtype);
  auto* d_starts  = static_cast<int32_t*>(run_starts.device(n + 1, WHOLEMEMORY_DT_INT));
  auto* d_order   = static_cast<int32_t*>(order.device(n, WHOLEMEMORY_DT_INT));
  auto* d_nunique = static_cast<int64_t*>(n_unique.device(1, WHOLEMEMORY_DT_INT64));
  void* d_ws = ws.device(static_cast<int64_t>(bk->dedup_workspace_bytes(n, d.indices.dtype)), WHOLEMEMORY_DT_INT8);
  // full-width keys: negative ("skip me") ids must stay distinct from every valid id; as unsigned keys they sort last
  int rc = bk->dedup_ids(d.indices_ptr, d.indices.dtype, n, 0, 0, d_unique, d_starts, d_order, d_nunique, d_ws, stream);
  if (rc != 0) throw hip_error("dedup of the requested ids failed");  // not a return: the peers are committed to the exchange

  // (1) owner segments of the distinct ids + the ids exchange; the host learns the counts in the exchange's one sync
  sorted_unique su{d_nunique};
  id_exchange x(env);
  bucket_and_exchange_ids(comm, d_unique, d.indices.dtype, n, entry_offsets, env, stream, &x, !comm->loopback, false, &su);
  const int64_t nu = x.total_valid;  // distinct NON-NEGATIVE ids
  // (2) each of them once, through the exchange, straight into its row of a dense [nu, dim] buffer of the output dtype
  char* uniq_rows = static_cast<char*>(rows.device(dim * nu, d.plain.dtype));
  op_descs du     = d;
  du.indices_ptr  = d_unique;
  du.indices      = wholememory_create_array_desc(nu, 0, d.indices.dtype);
  du.plain_ptr    = uniq_rows;
  int64_t usz[2]  = {nu, dim};
  du.plain        = wholememory_create_matrix_desc(usz, dim, 0, d.plain.dtype);
  WHOLEMEMORY_RETURN_ON_FAIL(gather_distributed_rows(handle, du, env, stream, gather_sms, nullptr, false, nullptr, &x));
  // (3) expand: out[i] = uniq_rows[run of i]; positions of negative ids get -1 and stay untouched
  auto* inv = static_cast<int64_t*>(inverse.device(n, WHOLEMEMORY_DT_INT64));
  // (an id past the last row belongs to no owner segment and was not fetched: its output row stays untouched, like a
  // negative id's — the plain route reads past the last owner's shard for such ids, undefined in the reference too)
  WM_BK(bk->run_inverse(d_starts, d_order, d_unique, d.indices.dtype, d_nunique, n, static_cast<int64_t>(entry_offsets.back()),
                        inv, stream));
  wm_rows_args ea{};
  fill_rows_args(&ea, wholememory_create_continuous_global_reference(uniq_rows), du.plain, inv, WHOLEMEMORY_DT_INT64, n,
                 d.plain_ptr, d.plain, gather_sms);
  WM_BK(bk->gather_rows(&ea, stream));
  // like the plain route, this one returns with its last kernels queued: the scratch goes back to the caller's
  // stream-ordered allocator (env_func_ptrs.h), and the side stream of the pipelined exchange was fenced by its last event
  if (debug_sync_enabled()) WM_BK(bk->stream_sync(stream));
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t gather_distributed(wholememory_handle_t handle, const op_descs& d,
                                            wholememory_env_func_t* env, void* stream, int gather_sms)
{
  const auto* bk = backend();
  wholememory_comm_t comm;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_communicator(&comm, handle));
  const int64_t n = d.indices.size;
  const char* sw  = WM_KNOB("WM_GATHER_DEDUP");
  const int mode  = sw == nullptr ? -1 : atoi(sw);  // -1 auto, 0 never, 1 always (several ranks), 2 always
  // every condition here is the same on all ranks (the op is collective: a rank with nothing to ask, or with too much,
  // must still walk the same sequence of collectives as the others)
  const bool can  = bk->run_inverse != nullptr && bk->sorted_owner_counts != nullptr &&
                   d.table.storage_offset >= 0 && d.table.storage_offset + d.table.sizes[1] <= d.table.stride;
  const bool exchanging = !comm->single_rank_direct();  // one rank and nothing to exchange: de-duplication only costs
  if (!can || mode == 0 || (!exchanging && mode != 2)) return gather_distributed_rows(handle, d, env, stream, gather_sms);
  const size_t tes   = wholememory_dtype_get_element_size(d.table.dtype);
  auto entry_offsets = entry_offsets_of(handle, tes * static_cast<size_t>(d.table.stride));
  if (mode >= 1) {
    if (n >= (INT64_C(1) << 31)) return WHOLEMEMORY_INVALID_INPUT;  // forced, and this batch cannot be (int32 positions)
    return gather_distributed_dedup(handle, d, env, stream, gather_sms, comm, entry_offsets);
  }

  // auto: bucket and exchange the ids as they are, with the duplicate estimate riding along; keep going on that exchange
  // when the batch is not worth de-duplicating (the common case), else start over on the distinct ids
  const int64_t threshold = [] {
    const char* e = WM_KNOB("WM_GATHER_DEDUP_PERMILLE");
    return e != nullptr && atoi(e) > 0 ? static_cast<int64_t>(atoi(e)) : INT64_C(100);
  }();
  id_exchange x(env);
  bucket_and_exchange_ids(comm, d.indices_ptr, d.indices.dtype, n, entry_offsets, env, stream, &x, !comm->loopback, false,
                          nullptr, true, true);   // counts + estimate only: what follows depends on the estimate
  const bool trace = WM_KNOB("WM_GATHER_DEDUP_TRACE") != nullptr;
  if (trace)
    WM_WARN("gather of %ld ids: duplicate estimate %ld permille (mean over %d ranks), threshold %ld -> %s",
            static_cast<long>(n), static_cast<long>(x.dup_permille), comm->world_size, static_cast<long>(threshold),
            x.dup_permille >= threshold ? "de-duplicate" : "as they are");
  if (x.dup_permille < threshold) {  // (-1: some rank cannot de-duplicate its batch)
    finish_id_exchange(comm, d.indices_ptr, d.indices.dtype, n, entry_offsets, env, stream, &x);
    return gather_distributed_rows(handle, d, env, stream, gather_sms, nullptr, false, nullptr, &x);
  }
  return gather_distributed_dedup(handle, d, env, stream, gather_sms, comm, entry_offsets);
}

// `cache` (optional): this rank's device row cache of its own shard — the owner-side gathers then read resident rows
// from the cache lines and only the others from the raw shard (reference device_cached_host_embedding::gather,
// embedding.cpp:576-760); with adjust_cache the ids that arrive at this owner update the cache first.
wholememory_error_code_t gather_distributed_rows(wholememory_handle_t handle, const op_descs& d,
                                                 wholememory_env_func_t* env, void* stream, int gather_sms,
                                                 row_cache* cache, bool adjust_cache, const route* via, id_exchange* prepared)
{
  const auto* bk = backend();
  if (d.table.storage_offset < 0 || d.table.storage_offset + d.table.sizes[1] > d.table.stride)
    return WHOLEMEMORY_INVALID_INPUT;  // gather_op_impl_nccl.cu:45-48
  wholememory_comm_t comm;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_communicator(&comm, handle));
  const size_t tes         = wholememory_dtype_get_element_size(d.table.dtype);
  const size_t oes         = wholememory_dtype_get_element_size(d.plain.dtype);
  const size_t ies         = wholememory_dtype_get_element_size(d.indices.dtype);
  const int64_t dim        = d.table.sizes[1];
  auto entry_offsets       = entry_offsets_of(handle, tes * static_cast<size_t>(d.table.stride));
  if (via != nullptr) comm = via->comm, entry_offsets = via->offsets;
  const char* indices      = static_cast<const char*>(d.indices_ptr);  // data pointer: offset already applied
  if (comm->single_rank_direct() && cache == nullptr && prepared == nullptr) {
    // one rank owns every row: nothing to bucket or exchange, the gather kernel itself skips negative ids
    wm_rows_args a{};
    fill_rows_args(&a, local_shard_gref(handle), d.table, indices, d.indices.dtype, d.indices.size, d.plain_ptr, d.plain,
                   gather_sms);
    WM_BK(bk->gather_rows(&a, stream));
    if (debug_sync_enabled()) WM_BK(bk->stream_sync(stream));
    return WHOLEMEMORY_SUCCESS;
  }

  // loopback (WM_EXCHANGE_SELF=1): this rank's own segment travels through the transport like a peer's
  const bool self_local = !comm->loopback;
  id_exchange own(env);
  if (prepared == nullptr)
    bucket_and_exchange_ids(comm, indices, d.indices.dtype, d.indices.size, entry_offsets, env, stream, &own, self_local);
  id_exchange& x = prepared != nullptr ? *prepared : own;
  // presorted ids: the output IS the bucketed layout (dense rows, one per id, in id order) — rows are gathered and
  // received straight into place
  const bool in_place = x.presorted;
  if (in_place) WM_CHECK(d.plain.stride == d.plain.sizes[1] && d.plain.storage_offset == 0, "presorted gather needs a dense output");
  const auto local_gref = local_shard_gref(handle);
  // owner-side row gather: through the cache when there is one
  auto local_gather = [&](const wm_rows_args& ga) {
    if (cache == nullptr) {
      WM_BK(bk->gather_rows(&ga, stream));
    } else if (row_cache_gather(cache, ga, env, stream) != WHOLEMEMORY_SUCCESS) {
      throw hip_error("cached gather failed");
    }
  };
  if (cache != nullptr && adjust_cache) {
    const int64_t total_rows = static_cast<int64_t>(entry_offsets[comm->world_size]);
    if (self_local)
      WHOLEMEMORY_RETURN_ON_FAIL(row_cache_update(cache, static_cast<const char*>(x.bucketed_ids) + ies * x.self_offset,
                                                  d.indices.dtype, x.self_count, total_rows, env, stream));
    WHOLEMEMORY_RETURN_ON_FAIL(row_cache_update(cache, x.recv_ids, d.indices.dtype, x.total_recv, total_rows, env, stream));
  }

  // (a) ids this rank owns itself: straight from the local shard into their final output rows
  //     (row_map = raw_indices) — no staging buffer, no copy, no reorder pass for them
  if (x.self_count > 0 && self_local) {
    wm_rows_args sa{};
    fill_rows_args(&sa, local_gref, d.table, static_cast<const char*>(x.bucketed_ids) + ies * x.self_offset,
                   d.indices.dtype, x.self_count, d.plain_ptr, d.plain, gather_sms);
    if (in_place)
      sa.plain = static_cast<char*>(d.plain_ptr) + static_cast<size_t>(x.self_offset) * static_cast<size_t>(dim) * oes;
    else
      sa.row_map = x.raw_indices + x.self_offset;
    local_gather(sa);
    g_dist_gather_launches.fetch_add(1, std::memory_order_relaxed);
  }

  // (b)-(d) the peers' rows, pipelined in C row-chunks so the three legs overlap:
  //   G_c  owner side: gather chunk c of every peer's requested rows into the send buffer, already cast to the
  //        output dtype (gather_op_impl_nccl.cu:115-140)                                — HBM, caller's stream
  //   A_c  rows all-to-all-v of chunk c (gather_op_impl_nccl.cu:141-150)                 — xGMI, side stream
  //   R_c  reorder on receive: out[raw_indices[j]] = recv[j] (gather_op_impl_nccl.cu:151-168) — HBM, caller's stream
  // issue order on the caller's stream: G_0 G_1 R_0 G_2 R_1 ... so that G_{c+1} and R_{c-1} run while A_c is on the
  // links. Chunk c of a segment of n rows is [n*c/C, n*(c+1)/C) on both ends of a pair, so sizes always match.
  temp_mem local_rows(env), recv_rows(env), ids_cm_mem(env), raw_cm_mem(env);
  char* local_buf = static_cast<char*>(local_rows.device(dim * x.total_recv, d.plain.dtype));
  char* recv_buf  = in_place ? static_cast<char*>(d.plain_ptr)   // bucketed layout = the output itself
                             : static_cast<char*>(recv_rows.device(dim * x.total_valid, d.plain.dtype));
  const size_t row_bytes = static_cast<size_t>(dim) * oes;
  const int W            = comm->world_size;
  const int rank         = comm->world_rank;
  const int C            = exchange_chunks(W, x.global_moved);
  const auto out_gref    = wholememory_create_continuous_global_reference(d.plain_ptr);
  auto chunk_of = [C](int64_t n, int c, int64_t* a, int64_t* b) {
    *a = n * c / C;
    *b = n * (c + 1) / C;
  };
  // ONE launch per chunk and side (round 5; rounds 2-4 launched per peer: 2 (W - 1) C + 1 row kernels per call — 57 at
  // W = 8, C = 4 — a measurable tax on mini-batch-sized gathers). The received ids (serving side) and the original positions
  // of the requested rows (requesting side) are brought into CHUNK-MAJOR order once, by one small kernel each (backend:
  // permute_chunks): chunk c of every peer's segment then lies in one contiguous range of the ids, of the send buffer, of the
  // receive buffer and of the positions, and the chunk's owner gather / reorder is one row kernel over that range. With one
  // chunk the peer-major order already is contiguous (and the rows before / after this rank's own segment are two ranges).
  // Backends without permute_chunks, more than 16 ranks, or WM_EXCHANGE_PER_PEER=1: the per-peer launches as before.
  const bool per_peer = bk->permute_chunks == nullptr || W > 16 || W <= 2 /* one peer: a chunk is one range already */ ||
                        (WM_KNOB("WM_EXCHANGE_PER_PEER") != nullptr && WM_KNOB("WM_EXCHANGE_PER_PEER")[0] == '1');
  const bool fold_serve = !per_peer;                          // serving side: ids -> send buffer
  const bool fold_recv  = !per_peer && !in_place;            // requesting side: receive buffer -> output rows
  // chunk-major starts: serve_start[c] over recv_counts, want_start[c] over send_counts (self travels as 0 when kept local)
  std::vector<int64_t> serve_start(C + 1, 0), want_start(C + 1, 0);
  for (int c = 0; c < C; c++) {
    int64_t s1 = 0, s2 = 0;
    for (int p = 0; p < W; p++) {
      int64_t a, b;
      chunk_of(x.recv_counts[p], c, &a, &b), s1 += b - a;
      chunk_of(x.send_counts[p], c, &a, &b), s2 += b - a;
    }
    serve_start[c + 1] = serve_start[c] + s1;
    want_start[c + 1]  = want_start[c] + s2;
  }
  auto serve_pos = [&](int c, int p) {   // where chunk c of peer p's requests starts in the chunk-major order
    int64_t pos = serve_start[c];
    for (int q = 0; q < p; q++) {
      int64_t a, b;
      chunk_of(x.recv_counts[q], c, &a, &b), pos += b - a;
    }
    return pos;
  };
  auto want_pos = [&](int c, int p) {
    int64_t pos = want_start[c];
    for (int q = 0; q < p; q++) {
      int64_t a, b;
      chunk_of(x.send_counts[q], c, &a, &b), pos += b - a;
    }
    return pos;
  };
  const char* serve_ids  = static_cast<const char*>(x.recv_ids);
  const int64_t* want_raw = x.raw_indices;
  if (C > 1 && fold_serve && x.total_recv > 0) {
    void* cm = ids_cm_mem.device(x.total_recv, d.indices.dtype);
    WM_BK(bk->permute_chunks(x.recv_ids, cm, static_cast<int>(ies), x.recv_offsets.data(), x.recv_counts.data(), W, C, stream));
    g_dist_gather_launches.fetch_add(1, std::memory_order_relaxed);
    serve_ids = static_cast<const char*>(cm);
  }
  if (C > 1 && fold_recv && x.total_send > 0) {
    auto* cm = static_cast<int64_t*>(raw_cm_mem.device(x.total_send, WHOLEMEMORY_DT_INT64));
    WM_BK(bk->permute_chunks(x.raw_indices, cm, 8, x.bucket_offsets.data(), x.send_counts.data(), W, C, stream));
    g_dist_gather_launches.fetch_add(1, std::memory_order_relaxed);
    want_raw = cm;
  }
  auto gather_range = [&](const char* ids, int64_t first, int64_t count) {   // ids[first ...] -> send buffer rows first ...
    if (count <= 0) return;
    int64_t lsz[2]  = {count, dim};
    auto local_desc = wholememory_create_matrix_desc(lsz, dim, 0, d.plain.dtype);
    wm_rows_args ga{};
    fill_rows_args(&ga, local_gref, d.table, ids + ies * first, d.indices.dtype, count, local_buf + row_bytes * first, local_desc,
                   gather_sms);
    local_gather(ga);
    g_dist_gather_launches.fetch_add(1, std::memory_order_relaxed);
  };
  auto reorder_range = [&](const int64_t* raw, int64_t first, int64_t count) {   // receive buffer rows first ... -> out[raw[...]]
    if (count <= 0) return;
    int64_t rsz[2]  = {count, dim};
    auto recv_desc  = wholememory_create_matrix_desc(rsz, dim, 0, d.plain.dtype);
    wm_rows_args ra{};
    fill_rows_args(&ra, out_gref, d.plain, raw + first, WHOLEMEMORY_DT_INT64, count, recv_buf + row_bytes * first, recv_desc, -1);
    WM_BK(bk->scatter_rows(&ra, stream));
    g_dist_gather_launches.fetch_add(1, std::memory_order_relaxed);
  };
  auto gather_chunk = [&](int c) {
    if (fold_serve) {
      // (C == 1: the peer-major arrays are contiguous over the peers as they are)
      gather_range(serve_ids, serve_start[c], serve_start[c + 1] - serve_start[c]);
      return;
    }
    for (int p = 0; p < W; p++) {
      int64_t a, b;
      chunk_of(x.recv_counts[p], c, &a, &b);
      gather_range(serve_ids, x.recv_offsets[p] + a, b - a);
    }
  };
  auto exchange_chunk = [&](int c, void* on_stream) {
    std::vector<int64_t> sc(W), so(W), rc(W), ro(W);
    for (int p = 0; p < W; p++) {
      int64_t a, b;
      chunk_of(x.recv_counts[p], c, &a, &b);  // what this rank serves to p
      sc[p] = b - a, so[p] = (fold_serve && C > 1) ? serve_pos(c, p) : x.recv_offsets[p] + a;
      chunk_of(x.send_counts[p], c, &a, &b);  // what p serves to this rank
      rc[p] = b - a, ro[p] = (fold_recv && C > 1) ? want_pos(c, p) : x.bucket_offsets[p] + a;
    }
    exchange_segments(comm, local_buf, sc, so, recv_buf, rc, ro, row_bytes, on_stream);
  };
  auto reorder_chunk = [&](int c) {
    if (in_place) return;  // received where they belong
    if (fold_recv && C > 1) {
      reorder_range(want_raw, want_start[c], want_start[c + 1] - want_start[c]);
      return;
    }
    if (fold_recv) {
      // one chunk: the peers' rows are the bucketed order minus this rank's own segment — the range before it and the one after
      const int64_t self_b = self_local ? x.self_offset : x.total_valid, self_e = self_local ? x.self_offset + x.self_count : x.total_valid;
      reorder_range(x.raw_indices, 0, self_b);
      reorder_range(x.raw_indices, self_e, x.total_valid - self_e);
      return;
    }
    for (int p = 0; p < W; p++) {
      if (p == rank && self_local) continue;
      int64_t a, b;
      chunk_of(x.send_counts[p], c, &a, &b);
      reorder_range(x.raw_indices, x.bucket_offsets[p] + a, b - a);
    }
  };

  if (C == 1) {
    gather_chunk(0);
    if (debug_sync_enabled()) WM_BK(bk->stream_sync(stream));
    exchange_chunk(0, stream);
    reorder_chunk(0);
  } else {
    void* side = comm->get_side_stream();
    event_set gathered(C), arrived(C);
    for (int c = 0; c < C; c++) {
      gather_chunk(c);
      WM_BK(bk->event_record(gathered[c], stream));
      WM_BK(bk->stream_wait_event(side, gathered[c]));
      exchange_chunk(c, side);
      WM_BK(bk->event_record(arrived[c], side));
      if (c >= 1) {
        WM_BK(bk->stream_wait_event(stream, arrived[c - 1]));
        reorder_chunk(c - 1);
      }
    }
    WM_BK(bk->stream_wait_event(stream, arrived[C - 1]));  // also orders every side-stream access to the
    reorder_chunk(C - 1);                                   // scratch buffers before anything later on `stream`
  }
  if (debug_sync_enabled()) WM_BK(bk->stream_sync(stream));
  return WHOLEMEMORY_SUCCESS;
}

// HIERARCHY tables (reference wholememory_gather_hierarchy, gather_op_impl_hierarchy.cu:117-352): rows are owned as in
// DISTRIBUTED, but a request reaches a remote node in two hops so that traffic between nodes only ever flows between
// equal local ranks ("rails") and every distinct row crosses the network once per node-local relay:
//   A  ids -> the rank of MY node whose local rank equals the owner's (all-to-all-v inside the node, xGMI)
//   B  the relay de-duplicates what it received and fetches each distinct row from its owner — same local rank, another
//      node — by the ordinary exchange over the cross-node communicator (ids out, rows back), then expands the
//      duplicates again
//   A' rows return to the requesters inside the node and are placed by their original positions
// Results are those of a DISTRIBUTED gather; only the route differs. Collective over the table's communicator: every
// rank must call, also one that asks for nothing (it still relays and serves).
wholememory_error_code_t gather_hierarchy(wholememory_handle_t handle, const op_descs& d, wholememory_env_func_t* env,
                                          void* stream, int gather_sms)
{
  const auto* bk = backend();
  if (d.table.storage_offset < 0 || d.table.storage_offset + d.table.sizes[1] > d.table.stride)
    return WHOLEMEMORY_INVALID_INPUT;
  wholememory_comm_t comm, local_comm, cross_comm;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_communicator(&comm, handle));
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_local_communicator(&local_comm, handle));
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_cross_communicator(&cross_comm, handle));
  const int L = local_comm->world_size, X = cross_comm->world_size;
  WM_CHECK(L * X == comm->world_size, "HIERARCHY: local size x cross size != world size");
  const size_t tes   = wholememory_dtype_get_element_size(d.table.dtype);
  const size_t oes   = wholememory_dtype_get_element_size(d.plain.dtype);
  const int64_t dim  = d.table.sizes[1];
  const int64_t n    = d.indices.size;
  auto entry_offsets = entry_offsets_of(handle, tes * static_cast<size_t>(d.table.stride));
  route rail{cross_comm, std::vector<size_t>(X + 1)};  // hop B: node k holds the rows of ranks [k*L, (k+1)*L)
  for (int k = 0; k <= X; k++) rail.offsets[k] = entry_offsets[static_cast<size_t>(k) * L];
  if (L == 1)  // one rank per node: no relay, the rail exchange is the whole gather
    return gather_distributed_rows(handle, d, env, stream, gather_sms, nullptr, false, &rail);

  // ---- hop A: ids to the relay (all W row ranges, folded onto the L ranks of this node by owner % L) ----
  id_exchange xa(env);
  bucket_and_exchange_ids(local_comm, d.indices_ptr, d.indices.dtype, n, entry_offsets, env, stream, &xa, false);
  const int64_t n_relay = xa.total_recv;

  // ---- relay: distinct ids only (reference sort_unique_ids_for_hierarchy_func) ----
  temp_mem unique_ids(env), run_starts(env), order(env), n_unique(env), ws(env), host_n(env), inverse(env);
  const bool dedup = n_relay > 0 && n_relay < (INT64_C(1) << 31) && bk->run_inverse != nullptr;
  void* fetch_ids  = xa.recv_ids;
  int64_t n_fetch  = n_relay;
  int64_t* inv     = nullptr;
  if (dedup) {
    void* d_unique  = unique_ids.device(n_relay, d.indices.dtype);
    auto* d_starts  = static_cast<int32_t*>(run_starts.device(n_relay + 1, WHOLEMEMORY_DT_INT));
    auto* d_order   = static_cast<int32_t*>(order.device(n_relay, WHOLEMEMORY_DT_INT));
    auto* d_nunique = static_cast<int64_t*>(n_unique.device(1, WHOLEMEMORY_DT_INT64));
    void* d_ws = ws.device(static_cast<int64_t>(bk->dedup_workspace_bytes(n_relay, d.indices.dtype)), WHOLEMEMORY_DT_INT8);
    int rc = bk->dedup_ids(xa.recv_ids, d.indices.dtype, n_relay, 0, 0, d_unique, d_starts, d_order, d_nunique, d_ws, stream);
    if (rc != 0) throw hip_error("dedup of relayed ids failed");  // not a return: the peers are already committed to hop B
    inv = static_cast<int64_t*>(inverse.device(n_relay, WHOLEMEMORY_DT_INT64));
    WM_BK(bk->run_inverse(d_starts, d_order, d_unique, d.indices.dtype, d_nunique, n_relay, 0, inv, stream));
    auto* h_n = static_cast<int64_t*>(host_n.pinned(1, WHOLEMEMORY_DT_INT64));
    WM_BK(bk->memcpy_async(h_n, d_nunique, sizeof(int64_t), stream));
    WM_BK(bk->stream_sync(stream));
    fetch_ids = d_unique;
    n_fetch   = *h_n;
  }

  // ---- hop B: each distinct row from its owner over the rail, already in the output dtype ----
  temp_mem fetched(env), relay_rows(env), back_rows(env);
  char* fetched_buf = static_cast<char*>(fetched.device(dim * n_fetch, d.plain.dtype));
  op_descs db       = d;
  db.indices_ptr    = fetch_ids;
  db.indices        = wholememory_create_array_desc(n_fetch, 0, d.indices.dtype);
  db.plain_ptr      = fetched_buf;
  int64_t fsz[2]    = {n_fetch, dim};
  db.plain          = wholememory_create_matrix_desc(fsz, dim, 0, d.plain.dtype);
  WHOLEMEMORY_RETURN_ON_FAIL(gather_distributed_rows(handle, db, env, stream, gather_sms, nullptr, false, &rail));

  // ---- expand the duplicates: relay_rows[j] = fetched[run of j], in the order the ids arrived in hop A ----
  char* send_buf = fetched_buf;
  WM_DEBUG("HIERARCHY gather: rank %d (local %d of %d, node %d of %d) asked %ld ids, relays %ld, fetches %ld distinct",
           comm->world_rank, local_comm->world_rank, L, cross_comm->world_rank, X, static_cast<long>(n),
           static_cast<long>(n_relay), static_cast<long>(n_fetch));
  if (dedup) {
    send_buf       = static_cast<char*>(relay_rows.device(dim * n_relay, d.plain.dtype));
    int64_t rsz[2] = {n_relay, dim};
    auto relay_desc = wholememory_create_matrix_desc(rsz, dim, 0, d.plain.dtype);
    wm_rows_args ea{};
    fill_rows_args(&ea, wholememory_create_continuous_global_reference(fetched_buf), db.plain, inv, WHOLEMEMORY_DT_INT64,
                   n_relay, send_buf, relay_desc, gather_sms);
    WM_BK(bk->gather_rows(&ea, stream));
  }

  // ---- hop A': rows back inside the node, then out[raw_indices[j]] = row j of the bucketed layout ----
  const size_t row_bytes = static_cast<size_t>(dim) * oes;
  char* back_buf = static_cast<char*>(back_rows.device(dim * xa.total_valid, d.plain.dtype));
  exchange_segments(local_comm, send_buf, xa.recv_counts, xa.recv_offsets, back_buf, xa.send_counts, xa.bucket_offsets,
                    row_bytes, stream);
  if (xa.total_valid > 0) {
    int64_t bsz[2] = {xa.total_valid, dim};
    auto back_desc = wholememory_create_matrix_desc(bsz, dim, 0, d.plain.dtype);
    wm_rows_args ra{};
    fill_rows_args(&ra, wholememory_create_continuous_global_reference(d.plain_ptr), d.plain, xa.raw_indices,
                   WHOLEMEMORY_DT_INT64, xa.total_valid, back_buf, back_desc, -1);
    WM_BK(bk->scatter_rows(&ra, stream));
  }
  WM_BK(bk->stream_sync(stream));  // scratch buffers return to the caller's allocator
  return WHOLEMEMORY_SUCCESS;
}

}  // namespace

// gather of an embedding that has a device row cache (embedding.cpp)
wholememory_error_code_t gather_cached(wholememory_tensor_t table, wholememory_tensor_t indices_tensor,
                                       wholememory_tensor_t output_tensor, wholememory_env_func_t* env, void* stream,
                                       int gather_sms, row_cache* cache, bool adjust_cache)
{
  op_descs d;
  WHOLEMEMORY_RETURN_ON_FAIL(check_args(table, indices_tensor, output_tensor, "output", &d));
  auto handle = wholememory_tensor_get_memory_handle(table);
  if (cache->same_comm)  // owners serve their shard (cache first), rows travel by all-to-all-v — for every memory type
    return gather_distributed_rows(handle, d, env, stream, gather_sms, cache, adjust_cache);
  if (cache->raw_addressable) {  // local read-only cache of a table that is addressable from here
    if (adjust_cache)
      // (the table's row count bounds the sort keys: 24 bits / 3 radix passes for a 10 M-row table instead of all 64; ids
      // outside [0, rows) are left out of the runs, and they have no business in the cache)
      WHOLEMEMORY_RETURN_ON_FAIL(row_cache_update(cache, d.indices_ptr, d.indices.dtype, d.indices.size, d.table.sizes[0], env, stream));
    wm_rows_args a{};
    fill_rows_args(&a, cache->args.raw_gref, d.table, d.indices_ptr, d.indices.dtype, d.indices.size, d.plain_ptr, d.plain,
                   gather_sms);
    return row_cache_gather(cache, a, env, stream);
  }
  // local read-only cache of a DISTRIBUTED table (reference local_cached_global_readonly_embedding over NCCL,
  // embedding.cpp:762-892): rows only reach this rank through the exchange, so every step that touches the raw table is
  // a collective distributed gather — the fill of newly chosen cache lines and the lookups that miss. Every rank of the
  // embedding's communicator calls this together (as for any gather of a DISTRIBUTED embedding), even with nothing to do.
  const auto* bk     = backend();
  const int64_t n    = d.indices.size;
  const int64_t dim  = d.table.sizes[1];
  if (adjust_cache) {
    temp_mem rows_mem(env), slots_mem(env), staging(env);
    int64_t n_fill = 0;
    WHOLEMEMORY_RETURN_ON_FAIL(row_cache_plan(cache, d.indices_ptr, d.indices.dtype, n, d.table.sizes[0], env, stream, &rows_mem, &slots_mem, &n_fill));
    char* rows_data = static_cast<char*>(staging.device(n_fill * cache->row_elems, d.table.dtype));
    op_descs df     = d;
    df.indices_ptr  = rows_mem.get();
    df.indices      = wholememory_create_array_desc(n_fill, 0, WHOLEMEMORY_DT_INT64);
    df.plain_ptr    = rows_data;
    int64_t fsz[2]  = {n_fill, dim};
    df.plain        = wholememory_create_matrix_desc(fsz, cache->row_elems, 0, d.table.dtype);
    WHOLEMEMORY_RETURN_ON_FAIL(gather_distributed_rows(handle, df, env, stream, gather_sms));
    WHOLEMEMORY_RETURN_ON_FAIL(row_cache_install(cache, rows_data, static_cast<const int64_t*>(slots_mem.get()), n_fill, stream));
    WM_BK(bk->stream_sync(stream));
  }
  temp_mem cache_idx_mem(env), raw_idx_mem(env);
  auto* cache_idx = static_cast<int64_t*>(cache_idx_mem.device(n, WHOLEMEMORY_DT_INT64));
  void* raw_idx   = raw_idx_mem.device(n, d.indices.dtype);
  WHOLEMEMORY_RETURN_ON_FAIL(row_cache_split(cache, d.indices_ptr, d.indices.dtype, n, cache_idx, raw_idx, stream));
  if (n > 0) {  // hits: out of the cache lines
    wm_rows_args hit{};
    fill_rows_args(&hit, wholememory_create_continuous_global_reference(cache->args.data), d.table, cache_idx,
                   WHOLEMEMORY_DT_INT64, n, d.plain_ptr, d.plain, gather_sms);
    hit.table_stride         = cache->row_elems;
    hit.table_storage_offset = 0;
    WM_BK(bk->gather_rows(&hit, stream));
  }
  op_descs dm    = d;  // misses (hits and negative ids are -1 in raw_idx and skipped): through the exchange
  dm.indices_ptr = raw_idx;
  dm.indices.storage_offset = 0;
  WHOLEMEMORY_RETURN_ON_FAIL(gather_distributed(handle, dm, env, stream, gather_sms));
  WM_BK(bk->stream_sync(stream));
  return WHOLEMEMORY_SUCCESS;
}

namespace {

wholememory_error_code_t scatter_distributed(wholememory_handle_t handle, const op_descs& d,
                                             wholememory_env_func_t* env, void* stream, int scatter_sms)
{
  const auto* bk = backend();
  if (d.table.storage_offset < 0 || d.table.storage_offset + d.table.sizes[1] > d.table.stride)
    return WHOLEMEMORY_INVALID_INPUT;  // scatter_op_impl_nccl.cu:45-48
  wholememory_comm_t comm;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_communicator(&comm, handle));
  const size_t tes    = wholememory_dtype_get_element_size(d.table.dtype);
  const size_t pes    = wholememory_dtype_get_element_size(d.plain.dtype);
  const size_t ies    = wholememory_dtype_get_element_size(d.indices.dtype);
  const int64_t dim   = d.table.sizes[1];
  auto entry_offsets  = entry_offsets_of(handle, tes * static_cast<size_t>(d.table.stride));
  const char* indices = static_cast<const char*>(d.indices_ptr);  // data pointer: offset already applied
  if (comm->single_rank_direct()) {
    // one rank owns every row: a plain scatter (negative ids are skipped by the kernel), then the reference's sync
    wm_rows_args a{};
    fill_rows_args(&a, local_shard_gref(handle), d.table, indices, d.indices.dtype, d.indices.size, d.plain_ptr, d.plain,
                   scatter_sms);
    WM_BK(bk->scatter_rows(&a, stream));
    WM_BK(bk->stream_sync(stream));
    return WHOLEMEMORY_SUCCESS;
  }

  const bool self_local = !comm->loopback;  // loopback: the self segment is exchanged like a peer's
  id_exchange x(env);
  bucket_and_exchange_ids(comm, indices, d.indices.dtype, d.indices.size, entry_offsets, env, stream, &x, self_local);
  const auto local_gref = local_shard_gref(handle);

  // (a) rows this rank owns itself: input row raw_indices[j] -> local table row, directly
  if (x.self_count > 0 && self_local) {
    wm_rows_args sa{};
    fill_rows_args(&sa, local_gref, d.table, static_cast<const char*>(x.bucketed_ids) + ies * x.self_offset,
                   d.indices.dtype, x.self_count, d.plain_ptr, d.plain, scatter_sms);
    sa.row_map = x.raw_indices + x.self_offset;
    WM_BK(bk->scatter_rows(&sa, stream));
    g_dist_scatter_launches.fetch_add(1, std::memory_order_relaxed);
  }

  // (b)-(d) rows for the peers, pipelined in C row-chunks over two streams (same scheme as the gather):
  //   L_c  line up chunk c of every peer's input rows in bucketed order (scatter_op_impl_nccl.cu:118-133) — HBM
  //   A_c  rows all-to-all-v of chunk c                                                              — xGMI, side stream
  //   S_c  owner writes (and casts) chunk c into its shard (scatter_op_impl_nccl.cu:145-166)          — HBM
  // ONE launch per chunk and side (round 6, as the gather since round 5): the positions of the rows to send and the received
  // ids are brought into chunk-major order once (ops_internal.hpp: chunk_layout), the send and receive buffers are laid out
  // chunk-major, so L_c and S_c are one row kernel each over a contiguous range: 2 C + 3 kernels per call instead of
  // 2 (W - 1) C + 1 (57 -> 11 at W = 8, C = 4). A scatter overwrites, duplicates are unordered in the reference
  // (gather_scatter_func.cuh:519-598), so the order in which the owner writes received rows is free.
  const size_t row_bytes = static_cast<size_t>(dim) * pes;
  const int W            = comm->world_size;
  const int rank         = comm->world_rank;
  const int C            = exchange_chunks(W, x.global_moved);
  const auto in_gref     = wholememory_create_continuous_global_reference(d.plain_ptr);
  const bool per_peer    = bk->permute_chunks == nullptr || W > 16 || W <= 2 /* one peer: a chunk is one range already */ ||
                        (WM_KNOB("WM_EXCHANGE_PER_PEER") != nullptr && WM_KNOB("WM_EXCHANGE_PER_PEER")[0] == '1');
  const bool folded = !per_peer && C > 1;   // chunk-major buffers
  const chunk_layout want(x.send_counts, C), serve(x.recv_counts, C);
  temp_mem send_rows(env), recv_rows(env), raw_cm_mem(env), ids_cm_mem(env);
  // (peer-major: the send buffer keeps the bucketed layout, this rank's own segment stays unused)
  char* send_buf = static_cast<char*>(send_rows.device(dim * (folded ? x.total_send : x.total_valid), d.plain.dtype));
  char* recv_buf = static_cast<char*>(recv_rows.device(dim * x.total_recv, d.plain.dtype));
  const int64_t* send_raw = x.raw_indices;
  const char* write_ids   = static_cast<const char*>(x.recv_ids);
  if (folded && x.total_send > 0) {
    auto* cm = static_cast<int64_t*>(raw_cm_mem.device(x.total_send, WHOLEMEMORY_DT_INT64));
    WM_BK(bk->permute_chunks(x.raw_indices, cm, 8, x.bucket_offsets.data(), x.send_counts.data(), W, C, stream));
    g_dist_scatter_launches.fetch_add(1, std::memory_order_relaxed);
    send_raw = cm;
  }
  if (folded && x.total_recv > 0) {
    void* cm = ids_cm_mem.device(x.total_recv, d.indices.dtype);
    WM_BK(bk->permute_chunks(x.recv_ids, cm, static_cast<int>(ies), x.recv_offsets.data(), x.recv_counts.data(), W, C, stream));
    g_dist_scatter_launches.fetch_add(1, std::memory_order_relaxed);
    write_ids = static_cast<const char*>(cm);
  }
  auto lineup_range = [&](const int64_t* raw, int64_t first, int64_t count) {   // in[raw[first ...]] -> send buffer rows first ...
    if (count <= 0) return;
    int64_t ssz[2] = {count, dim};
    auto send_desc = wholememory_create_matrix_desc(ssz, dim, 0, d.plain.dtype);
    wm_rows_args ga{};
    fill_rows_args(&ga, in_gref, d.plain, raw + first, WHOLEMEMORY_DT_INT64, count, send_buf + row_bytes * first, send_desc, -1);
    WM_BK(bk->gather_rows(&ga, stream));
    g_dist_scatter_launches.fetch_add(1, std::memory_order_relaxed);
  };
  auto write_range = [&](const char* ids, int64_t first, int64_t count) {   // receive buffer rows first ... -> table[ids[first ...]]
    if (count <= 0) return;
    int64_t rsz[2] = {count, dim};
    auto recv_desc = wholememory_create_matrix_desc(rsz, dim, 0, d.plain.dtype);
    wm_rows_args wa{};
    fill_rows_args(&wa, local_gref, d.table, ids + ies * first, d.indices.dtype, count, recv_buf + row_bytes * first, recv_desc,
                   scatter_sms);
    WM_BK(bk->scatter_rows(&wa, stream));
    g_dist_scatter_launches.fetch_add(1, std::memory_order_relaxed);
  };
  auto lineup_chunk = [&](int c) {
    if (folded) {
      lineup_range(send_raw, want.start(c), want.size(c));
    } else if (!per_peer) {
      // one chunk: the rows to send are the bucketed order minus this rank's own segment — the range before it and the one after
      const int64_t self_b = self_local ? x.self_offset : x.total_valid, self_e = self_local ? x.self_offset + x.self_count : x.total_valid;
      lineup_range(x.raw_indices, 0, self_b);
      lineup_range(x.raw_indices, self_e, x.total_valid - self_e);
    } else {
      for (int p = 0; p < W; p++) {
        if (p == rank && self_local) continue;
        lineup_range(x.raw_indices, x.bucket_offsets[p] + want.first(c, p), want.count(c, p));
      }
    }
  };
  auto exchange_chunk = [&](int c, void* on_stream) {
    std::vector<int64_t> sc(W), so(W), rc(W), ro(W);
    for (int p = 0; p < W; p++) {
      sc[p] = want.count(c, p), so[p] = folded ? want.pos(c, p) : x.bucket_offsets[p] + want.first(c, p);
      rc[p] = serve.count(c, p), ro[p] = folded ? serve.pos(c, p) : x.recv_offsets[p] + serve.first(c, p);
    }
    exchange_segments(comm, send_buf, sc, so, recv_buf, rc, ro, row_bytes, on_stream);
  };
  auto write_chunk = [&](int c) {
    if (folded) {
      write_range(write_ids, serve.start(c), serve.size(c));
    } else if (!per_peer) {
      write_range(write_ids, 0, x.total_recv);   // one chunk: everything received is one contiguous range
    } else {
      for (int p = 0; p < W; p++) write_range(write_ids, x.recv_offsets[p] + serve.first(c, p), serve.count(c, p));
    }
  };
  if (C == 1) {
    lineup_chunk(0);
    exchange_chunk(0, stream);
    write_chunk(0);
  } else {
    void* side = comm->get_side_stream();
    event_set lined_up(C), arrived(C);
    for (int c = 0; c < C; c++) {
      lineup_chunk(c);
      WM_BK(bk->event_record(lined_up[c], stream));
      WM_BK(bk->stream_wait_event(side, lined_up[c]));
      exchange_chunk(c, side);
      WM_BK(bk->event_record(arrived[c], side));
      if (c >= 1) {
        WM_BK(bk->stream_wait_event(stream, arrived[c - 1]));
        write_chunk(c - 1);
      }
    }
    WM_BK(bk->stream_wait_event(stream, arrived[C - 1]));
    write_chunk(C - 1);
  }
  WM_BK(bk->stream_sync(stream));  // scatter_op_impl_nccl.cu:168
  return WHOLEMEMORY_SUCCESS;
}

}  // namespace
}  // namespace wm

namespace wm {
std::atomic<int64_t> g_host_sorted_gathers{0};
std::atomic<int64_t> g_dist_gather_launches{0};
std::atomic<int64_t> g_dist_scatter_launches{0};
std::atomic<int64_t> g_grad_exchange_launches{0};
std::atomic<int64_t> g_alltoallv_bytes{0};
}

extern "C" {

int64_t wholememory_ext_host_sorted_gathers(void) { return wm::g_host_sorted_gathers.load(std::memory_order_relaxed); }
int64_t wholememory_ext_distributed_gather_launches(void) { return wm::g_dist_gather_launches.load(std::memory_order_relaxed); }
int64_t wholememory_ext_distributed_scatter_launches(void) { return wm::g_dist_scatter_launches.load(std::memory_order_relaxed); }
int64_t wholememory_ext_gradient_exchange_launches(void) { return wm::g_grad_exchange_launches.load(std::memory_order_relaxed); }
int64_t wholememory_ext_alltoallv_bytes(void) { return wm::g_alltoallv_bytes.load(std::memory_order_relaxed); }

wholememory_error_code_t wholememory_gather(wholememory_tensor_t wholememory_tensor,
                                            wholememory_tensor_t indices_tensor,
                                            wholememory_tensor_t output_tensor,
                                            wholememory_env_func_t* p_env_fns,
                                            void* stream,
                                            int gather_sms)
{
  WM_API_BEGIN
  wm::op_descs d;
  WHOLEMEMORY_RETURN_ON_FAIL(wm::check_args(wholememory_tensor, indices_tensor, output_tensor, "output", &d));
  const bool has_handle = wholememory_tensor_has_handle(wholememory_tensor);
  auto mt = has_handle ? wholememory_get_memory_type(wholememory_tensor_get_memory_handle(wholememory_tensor))
                       : WHOLEMEMORY_MT_NONE;
  if (has_handle && (mt == WHOLEMEMORY_MT_DISTRIBUTED || wm::mapped_via_exchange(wholememory_tensor, mt)))
    return wm::gather_distributed(wholememory_tensor_get_memory_handle(wholememory_tensor), d, p_env_fns, stream, gather_sms);
  if (has_handle && mt == WHOLEMEMORY_MT_HIERARCHY)  // gather_op.cpp:96-107
    return wm::gather_hierarchy(wholememory_tensor_get_memory_handle(wholememory_tensor), d, p_env_fns, stream, gather_sms);
  if (has_handle && mt != WHOLEMEMORY_MT_CHUNKED && mt != WHOLEMEMORY_MT_CONTINUOUS) return WHOLEMEMORY_NOT_SUPPORTED;
  wholememory_gref_t gref;
  WHOLEMEMORY_RETURN_ON_FAIL(wm::mapped_gref(wholememory_tensor, &gref));
  wm_rows_args a{};
  wm::fill_rows_args(&a, gref, d.table, d.indices_ptr, d.indices.dtype, d.indices.size, d.plain_ptr, d.plain, gather_sms);
  // HOST-located tables with rows of at most 512 bytes are gathered in ascending row order (gather_op.cpp:116-120,
  // sort_indices_func.cu:41-91): the rows cross PCIe, and neighbouring rows requested together are served faster than the
  // same rows in random order (C1: 10 M x 64 fp32, 1 M ids). The ids are sorted over their significant bits only and the
  // output row of every id travels with it as the row map. Measured on MI355X (profiles/r03_host_sorted_ab.txt, one process
  // per setting): 1 M ids 4.81 -> 4.71 ms (53.3 -> 54.4 GB/s of 256-byte rows), 4 M ids of 128-byte rows 12.59 -> 11.86 ms;
  // the sort costs ~60 us whatever the batch (histogram + 3 passes + the expansion), so 100 k ids LOSE 0.06 ms (0.518 -> 0.578):
  // the route starts at WM_HOST_SORTED_MIN ids (default 2^19; the reference sorts every batch). WM_HOST_SORTED_GATHER=0
  // switches it off. Ignoring the low id bits in the sort (WM_HOST_SORTED_LOW_BIT) buys nothing: 4 bits equal, 8 / 12 slower.
  std::unique_ptr<wm::temp_mem> sorted_ids_mem, sorted_raw_mem, sorted_ws_mem;   // (alive until the kernels are queued)
  if (has_handle && p_env_fns != nullptr && wm::backend()->sort_ids != nullptr && wm::host_sorted_gather_min() > 0 &&
      d.indices.size >= wm::host_sorted_gather_min() &&
      wholememory_get_memory_location(wholememory_tensor_get_memory_handle(wholememory_tensor)) == WHOLEMEMORY_ML_HOST &&
      d.table.sizes[1] * static_cast<int64_t>(wholememory_dtype_get_element_size(d.table.dtype)) <= 512) {
    const auto* bk    = wm::backend();
    const int64_t n   = d.indices.size;
    sorted_ids_mem.reset(new wm::temp_mem(p_env_fns));
    sorted_raw_mem.reset(new wm::temp_mem(p_env_fns));
    sorted_ws_mem.reset(new wm::temp_mem(p_env_fns));
    void* sorted      = sorted_ids_mem->device(n, d.indices.dtype);
    int64_t* raw      = static_cast<int64_t*>(sorted_raw_mem->device(n, WHOLEMEMORY_DT_INT64));
    void* ws          = sorted_ws_mem->device(static_cast<int64_t>(bk->sort_ids_workspace_bytes(n)), WHOLEMEMORY_DT_INT8);
    // ids address rows of the VIEW that was passed in (gather_scatter_func.cuh:297-298): its row count bounds the keys
    const int src = bk->sort_ids(d.indices_ptr, d.indices.dtype, n, d.table.sizes[0], wm::host_sorted_gather_low_bit(), sorted, raw,
                                 ws, stream);
    if (src == 0) {
      a.indices = sorted;
      a.row_map = raw;
      wm::g_host_sorted_gathers.fetch_add(1, std::memory_order_relaxed);
    } else if (src != -3) {
      return src == -1 ? WHOLEMEMORY_INVALID_INPUT : WHOLEMEMORY_CUDA_ERROR;
    }
  }
  int rc = wm::backend()->gather_rows(&a, stream);
  if (rc == -1) return WHOLEMEMORY_INVALID_INPUT;
  if (rc != 0) return WHOLEMEMORY_CUDA_ERROR;
  if (wm::debug_sync_enabled() && wm::backend()->stream_sync(stream) != 0) return WHOLEMEMORY_CUDA_ERROR;
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

wholememory_error_code_t wholememory_scatter(wholememory_tensor_t input_tensor,
                                             wholememory_tensor_t indices_tensor,
                                             wholememory_tensor_t wholememory_tensor,
                                             wholememory_env_func_t* p_env_fns,
                                             void* stream,
                                             int scatter_sms)
{
  WM_API_BEGIN
  wm::op_descs d;
  WHOLEMEMORY_RETURN_ON_FAIL(wm::check_args(wholememory_tensor, indices_tensor, input_tensor, "input", &d));
  const bool has_handle = wholememory_tensor_has_handle(wholememory_tensor);
  auto mt = has_handle ? wholememory_get_memory_type(wholememory_tensor_get_memory_handle(wholememory_tensor))
                       : WHOLEMEMORY_MT_NONE;
  // HIERARCHY rows are owned exactly as DISTRIBUTED rows, and a scatter has nothing to de-duplicate on the way: it takes
  // the direct exchange (the reference rejects scatter on HIERARCHY tables, scatter_op.cpp:94-97)
  if (has_handle && (mt == WHOLEMEMORY_MT_DISTRIBUTED || mt == WHOLEMEMORY_MT_HIERARCHY ||
                     wm::mapped_via_exchange(wholememory_tensor, mt)))
    return wm::scatter_distributed(wholememory_tensor_get_memory_handle(wholememory_tensor), d, p_env_fns, stream, scatter_sms);
  if (has_handle && mt != WHOLEMEMORY_MT_CHUNKED && mt != WHOLEMEMORY_MT_CONTINUOUS) return WHOLEMEMORY_NOT_SUPPORTED;
  wholememory_gref_t gref;
  WHOLEMEMORY_RETURN_ON_FAIL(wm::mapped_gref(wholememory_tensor, &gref));
  wm_rows_args a{};
  wm::fill_rows_args(&a, gref, d.table, d.indices_ptr, d.indices.dtype, d.indices.size, d.plain_ptr, d.plain, scatter_sms);
  int rc = wm::backend()->scatter_rows(&a, stream);
  if (rc == -1) return WHOLEMEMORY_INVALID_INPUT;
  if (rc != 0) return WHOLEMEMORY_CUDA_ERROR;
  if (wm::debug_sync_enabled() && wm::backend()->stream_sync(stream) != 0) return WHOLEMEMORY_CUDA_ERROR;
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

wholememory_error_code_t wholememory_ext_bucket_ids(const void* indices,
                                                    wholememory_dtype_t index_dtype,
                                                    int64_t n,
                                                    const void* entry_offsets_dev,
                                                    int world_size,
                                                    int64_t* counts_dev,
                                                    void* bucketed_ids_dev,
                                                    int64_t* raw_indices_dev,
                                                    wholememory_env_func_t* p_env_fns,
                                                    void* stream)
{
  return wholememory_ext_bucket_ids_folded(indices, index_dtype, n, entry_offsets_dev, world_size, world_size, counts_dev,
                                           bucketed_ids_dev, raw_indices_dev, p_env_fns, stream);
}

wholememory_error_code_t wholememory_ext_bucket_ids_folded(const void* indices,
                                                           wholememory_dtype_t index_dtype,
                                                           int64_t n,
                                                           const void* entry_offsets_dev,
                                                           int owner_count,
                                                           int world_size,
                                                           int64_t* counts_dev,
                                                           void* bucketed_ids_dev,
                                                           int64_t* raw_indices_dev,
                                                           wholememory_env_func_t* p_env_fns,
                                                           void* stream)
{
  WM_API_BEGIN
  if (n < 0 || world_size < 1 || counts_dev == nullptr || entry_offsets_dev == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (owner_count < world_size) return WHOLEMEMORY_INVALID_INPUT;
  if ((bucketed_ids_dev == nullptr) != (raw_indices_dev == nullptr)) return WHOLEMEMORY_INVALID_INPUT;
  const auto* bk = wm::backend();
  wm::temp_mem ws(p_env_fns);
  wm_bucket_args ba{};
  ba.indices       = indices;
  ba.index_dtype   = index_dtype;
  ba.n             = n;
  ba.entry_offsets = static_cast<const uint64_t*>(entry_offsets_dev);
  ba.world_size    = world_size;
  ba.owner_count   = owner_count == world_size ? 0 : owner_count;
  ba.counts        = counts_dev;
  ba.bucketed_ids  = bucketed_ids_dev;
  ba.raw_indices   = raw_indices_dev;
  ba.workspace     = ws.device(static_cast<int64_t>(bk->bucket_workspace_bytes(n, world_size)), WHOLEMEMORY_DT_INT8);
  int rc           = bk->bucket_ids(&ba, stream);
  if (rc == -1) return WHOLEMEMORY_INVALID_INPUT;
  // the workspace goes back to the caller's allocator when `ws` leaves scope: it must not be in use any more (an env
  // allocator is only required to be stream-ordered on ITS OWN current stream, see env_func_ptrs.h)
  if (rc == 0 && bk->stream_sync(stream) != 0) rc = -2;
  return rc == 0 ? WHOLEMEMORY_SUCCESS : WHOLEMEMORY_CUDA_ERROR;
  WM_API_END
}

wholememory_error_code_t wholememory_ext_round_robin_map(const void* ids,
                                                         void* mapped,
                                                         wholememory_dtype_t index_dtype,
                                                         int64_t n,
                                                         int64_t entry_start,
                                                         int world_size,
                                                         int round_robin_size,
                                                         void* stream)
{
  WM_API_BEGIN
  if (round_robin_size <= 0 || world_size <= 0) return WHOLEMEMORY_INVALID_INPUT;
  int rc = wm::backend()->round_robin_map(ids, mapped, index_dtype, n, entry_start, world_size, round_robin_size, 0, stream);
  if (rc == -1) return WHOLEMEMORY_INVALID_INPUT;
  return rc == 0 ? WHOLEMEMORY_SUCCESS : WHOLEMEMORY_CUDA_ERROR;
  WM_API_END
}

}  // extern "C"
