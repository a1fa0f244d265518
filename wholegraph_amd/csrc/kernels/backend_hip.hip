// wholegraph_amd — the HIP implementation of the device seam (backend.hpp): raw HIP memory /
// stream calls plus the launchers of the hand-written gfx950 kernels in this directory.
#include <hip/hip_runtime.h>

#include "../backend.hpp"
#include "../knobs.hpp"

namespace wm {

int hip_gather_rows(const wm_rows_args* a, void* stream);
int hip_scatter_rows(const wm_rows_args* a, void* stream);
size_t hip_bucket_workspace_bytes(int64_t n, int world_size);
int hip_bucket_ids(const wm_bucket_args* a, void* stream);
void hip_dedup_defer_join(int on);
int hip_dedup_join(void* stream);
int hip_device_error();
int hip_fill_iota(void* p, wholememory_dtype_t index_dtype, int64_t n, int64_t first, void* stream);
int hip_partials_nonfinite(const int32_t* run_starts, const int64_t* n_unique_dev, int64_t n_upper, const void* rows, int64_t dim,
                           int64_t stride, int64_t* flag_dev, void* stream);
int hip_permute_chunks(const void* src, void* dst, int elt_bytes, const int64_t* seg_offsets, const int64_t* seg_counts, int n_segs,
                       int n_chunks, void* stream);
size_t hip_dedup_workspace_bytes(int64_t n, wholememory_dtype_t index_dtype);
int hip_dedup_ids(const void* ids, wholememory_dtype_t index_dtype, int64_t n, int64_t key_upper_bound, int64_t key_lower_bound,
                  void* unique_ids, int32_t* run_starts, int32_t* order, int64_t* n_unique_out, void* workspace,
                  void* stream);
int hip_optimizer_step_dev(const wm_optimizer_args* a, const int64_t* n_unique_dev, void* stream);
size_t hip_long_run_ws_bytes(int64_t n_recv, int64_t dim);
int hip_run_inverse(const int32_t* run_starts, const int32_t* order, const void* unique_ids, wholememory_dtype_t index_dtype,
                    const int64_t* n_unique_dev, int64_t n, int64_t id_limit, int64_t* inverse, void* stream);
int hip_remap_self_order(int32_t* order, int64_t n, int64_t self_begin, int64_t self_count, const int64_t* self_rows,
                         void* stream);
int hip_round_robin_map(const void* ids, void* mapped, wholememory_dtype_t index_dtype, int64_t n, int64_t entry_start,
                        int world_size, int round_robin_size, int64_t rank_rows, void* stream);
int hip_fill_float(float* p, float value, int64_t count, void* stream);
size_t hip_dup_estimate_workspace_bytes(int64_t n);
int hip_dup_estimate(const void* ids, wholememory_dtype_t index_dtype, int64_t n, void* workspace, int64_t* permille_dev,
                     void* stream);
int hip_sorted_owner_counts(const void* sorted_ids, wholememory_dtype_t index_dtype, const int64_t* n_dev, int64_t n_upper,
                            const uint64_t* entry_offsets, int world_size, int64_t* counts, void* stream);
int hip_sample_counts(const wholememory_gref_t* row_gref, int64_t row_off, const int64_t* row_pairs, const void* centers,
                      wholememory_dtype_t id_dtype, int n, const int* n_dev, int max_sample, int* counts, void* stream);
int hip_sample_pair_ids(const void* centers, wholememory_dtype_t id_dtype, int n, int64_t* ids, void* stream);
size_t hip_scan_i32_ws_bytes(int64_t n);
int hip_exclusive_scan_i32(const int* in, int* out, int64_t n, void* ws, size_t ws_bytes, void* stream);
int hip_sample_offsets(const wholememory_gref_t* row_gref, int64_t row_off, const void* centers, wholememory_dtype_t id_dtype,
                       int n, const int* n_dev, int max_sample, int* offsets, void* ws, size_t ws_bytes, int ws_is_ones, void* stream);
int hip_sample_unweighted(const wm_sample_args* a, void* stream);
int hip_sample_weighted(const wm_sample_args* a, void* stream);
size_t hip_append_unique_ws_bytes(int nt, int nn, wholememory_dtype_t dt);
int hip_append_unique_table_region(int nt, int nn, wholememory_dtype_t dt, void* ws, void** ptr, size_t* bytes);
int hip_fill_ff(void* ptr, size_t bytes, void* stream);
int hip_append_unique_phase1(const void* targets, int nt, const void* neighbors, int nn, const int* nn_dev,
                             wholememory_dtype_t dt, void* ws, int* new_count_dev, int* publish_host, const wm_au_bounds* bounds,
                             void* stream);
int hip_append_unique_phase2(const void* targets, int nt, int nn, int nn_used, wholememory_dtype_t dt, void* ws,
                             void* out_unique, int* mapping, const int* copy_src, int* copy_dst, const wm_au_bounds* bounds,
                             void* stream);
int hip_csr_add_self_loop(const int* row_ptr, const int* col, int* out_row, int* out_col, int n_rows, void* stream);
bool hip_append_unique_takes_bounds(int nt, int nn, wholememory_dtype_t dt);

int hip_env_test_fill(const void* in, void* out, wholememory_dtype_t dt, int64_t dim, int64_t entries, int64_t stride, void* stream);
int hip_cache_update(const wm_cache_args* c, const void* unique_rows, wholememory_dtype_t dt, const int32_t* run_starts,
                     const int64_t* n_unique_dev, int64_t n_upper, int64_t* fill_rows, int64_t* fill_slots, int* fill_count,
                     void* stream);
int hip_cache_split(const wm_cache_args* c, const void* ids, wholememory_dtype_t dt, int64_t n, int64_t* cache_idx, void* raw_idx,
                    unsigned long long* hits_dev, void* stream);
int hip_cache_writeback(const wm_cache_args* c, int drop, void* stream);
int hip_cache_info(const wm_cache_args* c, unsigned long long* out2_dev, void* stream);
int hip_probe_memory(void* ptr, size_t bytes, int kind, int reps, float* ms_per_gib, void* stream);
size_t hip_sort_ids_workspace_bytes(int64_t n);
int hip_sort_ids(const void* ids, wholememory_dtype_t index_dtype, int64_t n, int64_t key_upper_bound, int low_bit,
                 void* sorted_ids, int64_t* raw, void* workspace, void* stream);

namespace {

int rc(hipError_t e) { return e == hipSuccess ? 0 : static_cast<int>(e); }

int h_device_count()
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
// a failed allocation must not leave its error behind for the next, unrelated hipGetLastError() check (a candidate shard of the
// placement probe that does not fit simply ends the search, memory_handle.cpp:alloc_local)
int h_malloc_device(void** p, size_t bytes)
{
  // WM_MALLOC_CONTIGUOUS=1 (opt-in, round 4): device blocks of WM_MALLOC_CONTIGUOUS_MIN bytes and more (default 1 GiB: table
  // shards, optimizer states) as PHYSICALLY CONTIGUOUS memory (hipDeviceMallocContiguous), falling back to the plain
  // allocation when the driver finds no such block. Measured on the 51 GB C2 table: a contiguous table's good level is the
  // best seen (gather 1.651-1.661 ms = 77.7-78.2 % of peak, scatter 1.56-1.59 ms = 81-82.5 %), but WHICH level a process gets
  // still varies with where the block lands (gather 4 of 6 processes at the good level, scatter 2 of 6; plain allocations:
  // 5 of 6 and 0 of 6), creating the table takes 1.5-3.2 s, and the TLB side is unchanged (one UTCL1 miss per random row):
  // profiles/r04_contiguous_table_ab.txt, r04_six_fresh_processes_contiguous.txt, r04_random_read_pmc.txt. hipIpc export of
  // such blocks works (the 2- / 3-process tests pass with it). Not the default: the gain is not reliable.
  const char* c  = WM_KNOB("WM_MALLOC_CONTIGUOUS");
  const char* cm = WM_KNOB("WM_MALLOC_CONTIGUOUS_MIN");
  const size_t contiguous_from = cm != nullptr && atoll(cm) > 0 ? static_cast<size_t>(atoll(cm)) : (size_t(1) << 30);
  if (c != nullptr && c[0] == '1' && bytes >= contiguous_from) {
    const hipError_t ec = hipExtMallocWithFlags(p, bytes, hipDeviceMallocContiguous);
    if (ec == hipSuccess) return 0;
    (void)hipGetLastError();
  }
  const hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) (void)hipGetLastError();
  return rc(e);
}
int h_free_device(void* p) { return rc(hipFree(p)); }
int h_malloc_pinned(void** p, size_t bytes)
{
  const hipError_t e = hipHostMalloc(p, bytes, hipHostMallocDefault);
  if (e != hipSuccess) (void)hipGetLastError();
  return rc(e);
}
int h_free_pinned(void* p) { return rc(hipHostFree(p)); }
int h_memcpy_async(void* dst, const void* src, size_t bytes, void* stream)
{
  return rc(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, static_cast<hipStream_t>(stream)));
}
int h_memset_async(void* dst, int value, size_t bytes, void* stream)
{
  return rc(hipMemsetAsync(dst, value, bytes, static_cast<hipStream_t>(stream)));
}
int h_stream_sync(void* stream) { return rc(hipStreamSynchronize(static_cast<hipStream_t>(stream))); }
int h_stream_create(void** s)
{
  hipStream_t st;
  hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  *s           = st;
  return rc(e);
}
int h_stream_destroy(void* s) { return rc(hipStreamDestroy(static_cast<hipStream_t>(s))); }
int h_event_create(void** ev)
{
  hipEvent_t e;
  hipError_t r = hipEventCreateWithFlags(&e, hipEventDisableTiming);
  *ev          = e;
  return rc(r);
}
int h_event_destroy(void* ev) { return rc(hipEventDestroy(static_cast<hipEvent_t>(ev))); }
int h_event_record(void* ev, void* s) { return rc(hipEventRecord(static_cast<hipEvent_t>(ev), static_cast<hipStream_t>(s))); }
int h_stream_wait_event(void* s, void* ev)
{
  return rc(hipStreamWaitEvent(static_cast<hipStream_t>(s), static_cast<hipEvent_t>(ev), 0));
}
static_assert(sizeof(hipIpcMemHandle_t) <= 64, "ipc handle does not fit the 64-byte slot");
int h_ipc_get(void* handle64, void* dev_ptr)
{
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, dev_ptr);
  if (e == hipSuccess) {
    __builtin_memset(handle64, 0, 64);
    __builtin_memcpy(handle64, &h, sizeof(h));
  }
  return rc(e);
}
int h_ipc_open(void** dev_ptr, const void* handle64)
{
  hipIpcMemHandle_t h;
  __builtin_memcpy(&h, handle64, sizeof(h));
  return rc(hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess));
}
int h_ipc_close(void* dev_ptr) { return rc(hipIpcCloseMemHandle(dev_ptr)); }
int h_host_register(void* host_ptr, size_t bytes, void** dev_ptr)
{
  hipError_t e = hipHostRegister(host_ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
  if (e != hipSuccess) return rc(e);
  return rc(hipHostGetDevicePointer(dev_ptr, host_ptr, 0));
}
int h_host_unregister(void* host_ptr) { return rc(hipHostUnregister(host_ptr)); }
int h_mem_info(size_t* free_bytes, size_t* total_bytes) { return rc(hipMemGetInfo(free_bytes, total_bytes)); }
int h_get_device(int* device) { return rc(hipGetDevice(device)); }

const wm_device_backend kHipBackend = {
  "hip-gfx950",
  h_device_count,
  h_malloc_device,
  h_free_device,
  h_malloc_pinned,
  h_free_pinned,
  h_memcpy_async,
  h_memset_async,
  h_stream_sync,
  h_stream_create,
  h_stream_destroy,
  h_event_create,
  h_event_destroy,
  h_event_record,
  h_stream_wait_event,
  h_ipc_get,
  h_ipc_open,
  h_ipc_close,
  h_host_register,
  h_host_unregister,
  hip_gather_rows,
  hip_scatter_rows,
  hip_bucket_workspace_bytes,
  hip_bucket_ids,
  hip_dedup_workspace_bytes,
  hip_dedup_ids,
  hip_optimizer_step_dev,
  hip_long_run_ws_bytes,
  hip_run_inverse,
  hip_remap_self_order,
  hip_round_robin_map,
  hip_fill_float,
  hip_dup_estimate_workspace_bytes,
  hip_dup_estimate,
  hip_sorted_owner_counts,
  hip_sample_counts,
  hip_sample_pair_ids,
  hip_sample_offsets,
  hip_scan_i32_ws_bytes,
  hip_exclusive_scan_i32,
  hip_sample_unweighted,
  hip_sample_weighted,
  hip_append_unique_ws_bytes,
  hip_append_unique_phase1,
  hip_append_unique_phase2,
  hip_csr_add_self_loop,
  hip_append_unique_takes_bounds,
  hip_env_test_fill,
  hip_cache_update,
  hip_cache_split,
  hip_cache_writeback,
  hip_cache_info,
  hip_probe_memory,
  hip_sort_ids_workspace_bytes,
  hip_sort_ids,
  h_mem_info,
  h_get_device,
  hip_append_unique_table_region,
  hip_fill_ff,
  hip_permute_chunks,
  hip_dedup_defer_join,
  hip_dedup_join,
  hip_device_error,
  hip_fill_iota,
  hip_partials_nonfinite,
};

}  // namespace

const wm_device_backend* hip_backend() { return &kHipBackend; }

}  // namespace wm
