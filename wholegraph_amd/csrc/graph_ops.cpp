// wholegraph_amd — host side of neighbour sampling and the small graph utilities (the C ABI of
// include/wholememory/wholegraph_op.h and graph_op.h). Kernels: kernels/graph.hip.
//
// Reference call stacks: wholegraph_csr_unweighted_sample_without_replacement
//   cpp/src/wholegraph_ops/unweighted_sample_without_replacement.cpp:24-190 (validation, dispatch on memory type)
//   -> ..._impl_mapped.cu -> ..._func.cuh:283-470 (count, scan, D2H of the total, output allocation, sample kernel);
// graph_append_unique cpp/src/graph_ops/append_unique.cpp:23-83 -> append_unique_func.cuh:300-353;
// csr_add_self_loop cpp/src/graph_ops/csr_add_self_loop.cpp:22-80.
#include <cmath>
#include <cstring>
#include <new>

#include <wholememory/graph_op.h>
#include <wholememory/wholegraph_op.h>
#include <wholememory/wholememory_op.h>

#include "knobs.hpp"
#include "ops_internal.hpp"
#include "pcg.hpp"

namespace {

using namespace wm;

#define WM_BK(call)                                                                   \
  do {                                                                                \
    int rc__ = (call);                                                                \
    if (rc__ != 0) throw wm::hip_error(wm::format_string("%s failed: %d", #call, rc__)); \
  } while (0)

// one variable-size result handed to the caller's allocator (reference output_memory_handle.hpp:23-93)
void* output_alloc(wholememory_env_func_t* env, void* memory_context, int64_t count, wholememory_dtype_t dtype)
{
  wholememory_tensor_description_t d;
  wholememory_initialize_tensor_desc(&d);
  d.dim            = 1;
  d.sizes[0]       = count;
  d.strides[0]     = 1;
  d.dtype          = dtype;
  d.storage_offset = 0;
  return env->output_fns.malloc_fn(&d, WHOLEMEMORY_MA_DEVICE, memory_context, env->output_fns.global_context);
}

bool array_of(wholememory_tensor_t t, const char* what, wholememory_array_description_t* out, wholememory_error_code_t* err)
{
  if (t == nullptr) {
    *err = WHOLEMEMORY_INVALID_INPUT;
    return false;
  }
  auto desc = *wholememory_tensor_get_tensor_description(t);
  auto* td  = &desc;
  if (td->dim != 1) {
    WM_ERROR("%s should be 1D tensor.", what);
    *err = WHOLEMEMORY_INVALID_INPUT;
    return false;
  }
  if (!wholememory_convert_tensor_desc_to_array(out, td)) {
    WM_ERROR("%s convert to array failed.", what);
    *err = WHOLEMEMORY_LOGIC_ERROR;
    return false;
  }
  return true;
}

bool is_index_dtype(wholememory_dtype_t d) { return d == WHOLEMEMORY_DT_INT || d == WHOLEMEMORY_DT_INT64; }

wholememory_memory_type_t memory_type_of(wholememory_tensor_t t)
{
  if (!wholememory_tensor_has_handle(t)) return WHOLEMEMORY_MT_NONE;
  return wholememory_get_memory_type(wholememory_tensor_get_memory_handle(t));
}

// a caller-side device array wrapped as a wholememory_tensor_t for the duration of one call
struct local_tensor {
  wholememory_tensor_t handle = nullptr;
  local_tensor(void* ptr, int64_t count, wholememory_dtype_t dtype)
  {
    wholememory_tensor_description_t d;
    wholememory_initialize_tensor_desc(&d);
    d.dim            = 1;
    d.sizes[0]       = count;
    d.strides[0]     = 1;
    d.dtype          = dtype;
    d.storage_offset = 0;
    if (wholememory_make_tensor_from_pointer(&handle, ptr, &d) != WHOLEMEMORY_SUCCESS) throw std::bad_alloc();
  }
  ~local_tensor()
  {
    if (handle != nullptr) wholememory_destroy_tensor(handle);
  }
  local_tensor(const local_tensor&)            = delete;
  local_tensor& operator=(const local_tensor&) = delete;
};

const wm_device_backend* graph_backend()
{
  const auto* bk = backend();
  if (bk->sample_unweighted == nullptr || bk->sample_weighted == nullptr || bk->append_unique_phase1 == nullptr) return nullptr;
  return bk;
}

// shared body of the two samplers; wm_csr_weight_ptr_tensor == nullptr selects the unweighted one
wholememory_error_code_t sample_without_replacement(
  wholememory_tensor_t wm_csr_row_ptr_tensor, wholememory_tensor_t wm_csr_col_ptr_tensor,
  wholememory_tensor_t wm_csr_weight_ptr_tensor, bool weighted, wholememory_tensor_t center_nodes_tensor,
  int max_sample_count, wholememory_tensor_t output_sample_offset_tensor, void* output_dest_memory_context,
  void* output_center_localid_memory_context, void* output_edge_gid_memory_context, unsigned long long random_seed,
  wholememory_env_func_t* p_env_fns, void* stream)
{
  const auto* bk = graph_backend();
  if (bk == nullptr) return WHOLEMEMORY_NOT_SUPPORTED;
  if (p_env_fns == nullptr || output_dest_memory_context == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  wholememory_error_code_t err = WHOLEMEMORY_SUCCESS;
  wholememory_array_description_t row_desc, col_desc, center_desc, offset_desc;
  if (!array_of(wm_csr_row_ptr_tensor, "wm_csr_row_ptr_tensor", &row_desc, &err)) return err;
  if (!array_of(wm_csr_col_ptr_tensor, "wm_csr_col_ptr_tensor", &col_desc, &err)) return err;
  if (!array_of(center_nodes_tensor, "center_nodes_tensor", &center_desc, &err)) return err;
  if (!array_of(output_sample_offset_tensor, "output_sample_offset_tensor", &offset_desc, &err)) return err;
  const auto row_mt = memory_type_of(wm_csr_row_ptr_tensor), col_mt = memory_type_of(wm_csr_col_ptr_tensor);
  if (row_mt == WHOLEMEMORY_MT_HIERARCHY || col_mt == WHOLEMEMORY_MT_HIERARCHY) {
    WM_ERROR("Memory type not supported.");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  // DISTRIBUTED CSR (reference ..._impl_nccl.cu / ..._nccl_func.cuh:196-390): the arrays are not addressable from this
  // rank, so row bounds and sampled columns travel through wholememory_gather (collective over the CSR's communicator)
  const bool via_gather = row_mt == WHOLEMEMORY_MT_DISTRIBUTED || col_mt == WHOLEMEMORY_MT_DISTRIBUTED;
  if (via_gather && weighted) {
    WM_ERROR("weighted sampling on DISTRIBUTED CSR tensors is not implemented (the reference has no such path either)");
    return WHOLEMEMORY_NOT_IMPLEMENTED;
  }
  // dtype rules of ..._func.cuh:304-314 and the dispatch table of ..._impl_mapped.cu (logic_error there)
  if (row_desc.dtype != WHOLEMEMORY_DT_INT64) {
    WM_ERROR("wm_csr_row_ptr_tensor must be int64, got %d", static_cast<int>(row_desc.dtype));
    return WHOLEMEMORY_LOGIC_ERROR;
  }
  if (offset_desc.dtype != WHOLEMEMORY_DT_INT) {
    WM_ERROR("output_sample_offset_tensor must be int32, got %d", static_cast<int>(offset_desc.dtype));
    return WHOLEMEMORY_LOGIC_ERROR;
  }
  if (!is_index_dtype(col_desc.dtype) || !is_index_dtype(center_desc.dtype)) {
    WM_ERROR("center nodes and csr_col_ptr must be int32 or int64");
    return WHOLEMEMORY_LOGIC_ERROR;
  }
  const int64_t n = center_desc.size;
  if (n >= (INT64_C(1) << 31) - 1) return WHOLEMEMORY_INVALID_INPUT;
  if (offset_desc.size < n + 1) {
    WM_ERROR("output_sample_offset_tensor needs %ld entries, has %ld", static_cast<long>(n + 1),
             static_cast<long>(offset_desc.size));
    return WHOLEMEMORY_INVALID_INPUT;
  }
  wm_sample_args a{};
  if (weighted) {
    wholememory_array_description_t weight_desc;
    if (!array_of(wm_csr_weight_ptr_tensor, "wm_csr_weight_ptr_tensor", &weight_desc, &err)) return err;
    const auto wmt = memory_type_of(wm_csr_weight_ptr_tensor);
    if (wmt == WHOLEMEMORY_MT_HIERARCHY) return WHOLEMEMORY_INVALID_INPUT;
    if (wmt == WHOLEMEMORY_MT_DISTRIBUTED) {
      WM_ERROR("WEIGHTED sampling with a DISTRIBUTED weight tensor is not implemented (the reference has no such path either; "
               "unweighted sampling on a DISTRIBUTED CSR is: see via_gather above)");
      return WHOLEMEMORY_NOT_IMPLEMENTED;
    }
    if (weight_desc.dtype != WHOLEMEMORY_DT_FLOAT && weight_desc.dtype != WHOLEMEMORY_DT_DOUBLE) {
      WM_ERROR("wm_csr_weight_ptr_tensor must be float or double");
      return WHOLEMEMORY_LOGIC_ERROR;
    }
    if (weight_desc.size != col_desc.size) {
      WM_ERROR("wm_csr_weight_ptr_tensor must have one weight per edge (%ld vs %ld)", static_cast<long>(weight_desc.size),
               static_cast<long>(col_desc.size));
      return WHOLEMEMORY_INVALID_INPUT;
    }
    if (max_sample_count > 8192) {
      // reference: key generation + cub segmented sort for > 256 samples (func.cuh:470-560); the in-LDS selection
      // built here covers 1..8192 (a candidate list of 2 M keys must fit the 160 KiB of LDS)
      WM_ERROR("weighted sampling with max_sample_count > 8192 is not implemented in this build");
      return WHOLEMEMORY_NOT_IMPLEMENTED;
    }
    WHOLEMEMORY_RETURN_ON_FAIL(tensor_mapped_gref(wm_csr_weight_ptr_tensor, &a.weight_gref));
    a.weight_storage_offset = weight_desc.storage_offset;
    a.weight_dtype          = weight_desc.dtype;
  }
  if (!via_gather) {
    WHOLEMEMORY_RETURN_ON_FAIL(tensor_mapped_gref(wm_csr_row_ptr_tensor, &a.row_gref));
    WHOLEMEMORY_RETURN_ON_FAIL(tensor_mapped_gref(wm_csr_col_ptr_tensor, &a.col_gref));
  }
  a.row_storage_offset = row_desc.storage_offset;
  a.col_storage_offset = col_desc.storage_offset;
  a.col_dtype          = col_desc.dtype;
  a.centers            = wholememory_tensor_get_data_pointer(center_nodes_tensor);
  a.center_dtype       = center_desc.dtype;
  a.n_center           = static_cast<int>(n);
  a.max_sample_count   = max_sample_count;
  a.random_seed        = random_seed;
  int* offsets         = static_cast<int*>(wholememory_tensor_get_data_pointer(output_sample_offset_tensor));
  a.sample_offsets     = offsets;

  // per-center counts -> exclusive scan into the caller's offset tensor -> total on the host
  temp_mem counts_mem(p_env_fns), scan_mem(p_env_fns);
  int* counts           = static_cast<int*>(counts_mem.device(n + 1, WHOLEMEMORY_DT_INT));
  const size_t scan_ws  = bk->scan_i32_workspace_bytes(n + 1);
  void* scan_ws_ptr     = scan_mem.device(static_cast<int64_t>(scan_ws), WHOLEMEMORY_DT_INT8);
  temp_mem pair_ids_mem(p_env_fns), pairs_mem(p_env_fns), egid_mem(p_env_fns);
  if (via_gather) {
    // (row_ptr[c], row_ptr[c + 1]) of every center node in one gather
    int64_t* pair_ids = static_cast<int64_t*>(pair_ids_mem.device(2 * n, WHOLEMEMORY_DT_INT64));
    int64_t* pairs    = static_cast<int64_t*>(pairs_mem.device(2 * n, WHOLEMEMORY_DT_INT64));
    WM_BK(bk->sample_pair_ids(a.centers, a.center_dtype, a.n_center, pair_ids, stream));
    local_tensor ids_t(pair_ids, 2 * n, WHOLEMEMORY_DT_INT64), pairs_t(pairs, 2 * n, WHOLEMEMORY_DT_INT64);
    WHOLEMEMORY_RETURN_ON_FAIL(wholememory_gather(wm_csr_row_ptr_tensor, ids_t.handle, pairs_t.handle, p_env_fns, stream, -1));
    a.row_pairs = pairs;
  }
  WM_BK(bk->sample_counts(via_gather ? nullptr : &a.row_gref, a.row_storage_offset, a.row_pairs, a.centers, a.center_dtype,
                          a.n_center, nullptr, max_sample_count, counts, stream));
  WM_BK(bk->exclusive_scan_i32(counts, offsets, n + 1, scan_ws_ptr, scan_ws, stream));
  int total = 0;
  WM_BK(bk->memcpy_async(&total, offsets + n, sizeof(int), stream));
  WM_BK(bk->stream_sync(stream));

  a.out_ids = output_alloc(p_env_fns, output_dest_memory_context, total, col_desc.dtype);
  if (output_center_localid_memory_context != nullptr)
    a.out_center_lid = static_cast<int*>(output_alloc(p_env_fns, output_center_localid_memory_context, total, WHOLEMEMORY_DT_INT));
  if (output_edge_gid_memory_context != nullptr)
    a.out_edge_gid = static_cast<int64_t*>(output_alloc(p_env_fns, output_edge_gid_memory_context, total, WHOLEMEMORY_DT_INT64));
  if (total > 0 && a.out_ids == nullptr) return WHOLEMEMORY_OUT_OF_MEMORY;
  if (via_gather) {
    // positions first (edge ids), then the columns by one more gather; every rank takes part even with nothing to sample
    void* out_ids = a.out_ids;
    a.out_ids     = nullptr;
    if (a.out_edge_gid == nullptr) a.out_edge_gid = static_cast<int64_t*>(egid_mem.device(total, WHOLEMEMORY_DT_INT64));
    if (total > 0) WM_BK(bk->sample_unweighted(&a, stream));
    local_tensor egid_t(a.out_edge_gid, total, WHOLEMEMORY_DT_INT64), out_t(out_ids, total, col_desc.dtype);
    WHOLEMEMORY_RETURN_ON_FAIL(wholememory_gather(wm_csr_col_ptr_tensor, egid_t.handle, out_t.handle, p_env_fns, stream, -1));
  } else if (total > 0) {
    WM_BK(weighted ? bk->sample_weighted(&a, stream) : bk->sample_unweighted(&a, stream));
  }
  // The reference returns with the samples complete (:385,:404), and so does this by default. A host framework whose env
  // allocator is ordered on `stream`, like every consumer of the outputs, may declare that
  // (wholememory_ext_set_async_completion): the call then returns with the kernels queued — one host round trip per call
  // (the count) instead of two. The DISTRIBUTED route keeps the drain: its gathers are
  // collectives and peers read this rank's buffers.
  if (via_gather || !async_completion_enabled() || debug_sync_enabled()) WM_BK(bk->stream_sync(stream));
  return WHOLEMEMORY_SUCCESS;
}

}  // namespace

extern "C" {

wholememory_error_code_t wholegraph_csr_unweighted_sample_without_replacement(
  wholememory_tensor_t wm_csr_row_ptr_tensor, wholememory_tensor_t wm_csr_col_ptr_tensor,
  wholememory_tensor_t center_nodes_tensor, int max_sample_count, wholememory_tensor_t output_sample_offset_tensor,
  void* output_dest_memory_context, void* output_center_localid_memory_context, void* output_edge_gid_memory_context,
  unsigned long long random_seed, wholememory_env_func_t* p_env_fns, void* stream)
{
  WM_API_BEGIN
  return sample_without_replacement(wm_csr_row_ptr_tensor, wm_csr_col_ptr_tensor, nullptr, false, center_nodes_tensor,
                                    max_sample_count, output_sample_offset_tensor, output_dest_memory_context,
                                    output_center_localid_memory_context, output_edge_gid_memory_context, random_seed,
                                    p_env_fns, stream);
  WM_API_END
}

wholememory_error_code_t wholegraph_csr_weighted_sample_without_replacement(
  wholememory_tensor_t wm_csr_row_ptr_tensor, wholememory_tensor_t wm_csr_col_ptr_tensor,
  wholememory_tensor_t wm_csr_weight_ptr_tensor, wholememory_tensor_t center_nodes_tensor, int max_sample_count,
  wholememory_tensor_t output_sample_offset_tensor, void* output_dest_memory_context,
  void* output_center_localid_memory_context, void* output_edge_gid_memory_context, unsigned long long random_seed,
  wholememory_env_func_t* p_env_fns, void* stream)
{
  WM_API_BEGIN
  return sample_without_replacement(wm_csr_row_ptr_tensor, wm_csr_col_ptr_tensor, wm_csr_weight_ptr_tensor, true,
                                    center_nodes_tensor, max_sample_count, output_sample_offset_tensor,
                                    output_dest_memory_context, output_center_localid_memory_context,
                                    output_edge_gid_memory_context, random_seed, p_env_fns, stream);
  WM_API_END
}

wholememory_error_code_t generate_random_positive_int_cpu(int64_t random_seed, int64_t subsequence, wholememory_tensor_t output)
{
  WM_API_BEGIN
  if (output == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  const auto d = *wholememory_tensor_get_tensor_description(output);
  if (d.dim != 1) {
    WM_ERROR("output should be 1D tensor.");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  if (!is_index_dtype(d.dtype)) {
    WM_ERROR("output should be int64 or int32 tensor.");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  void* p = wholememory_tensor_get_data_pointer(output);
  pcg32 rng(static_cast<uint64_t>(random_seed), 0, static_cast<uint64_t>(subsequence));
  for (int64_t i = 0; i < d.sizes[0]; i++) {
    if (d.dtype == WHOLEMEMORY_DT_INT)
      static_cast<int32_t*>(p)[i] = rng.next_i32();
    else
      static_cast<int64_t*>(p)[i] = rng.next_i64();
  }
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

wholememory_error_code_t generate_exponential_distribution_negative_float_cpu(int64_t random_seed, int64_t subsequence,
                                                                              wholememory_tensor_t output)
{
  WM_API_BEGIN
  if (output == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  const auto d = *wholememory_tensor_get_tensor_description(output);
  if (d.dim != 1) {
    WM_ERROR("output should be 1D tensor.");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  if (d.dtype != WHOLEMEMORY_DT_FLOAT) {
    WM_ERROR("output should be float.");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  float* p = static_cast<float*>(wholememory_tensor_get_data_pointer(output));
  pcg32 rng(static_cast<uint64_t>(random_seed), 0, static_cast<uint64_t>(subsequence));
  // log2 of a uniform in (0,1) built from a mantissa in [0.5,1) and a geometric exponent = the number of leading zero
  // bits of a 64-bit stream (redrawn while it is all zeros): raft_random_gen.cu:83-105
  for (int64_t i = 0; i < d.sizes[0]; i++) {
    float u = rng.next_float();
    u       = static_cast<float>(-(0.5 + 0.5 * static_cast<double>(u)));
    uint64_t bits;
    int redraws = -1;
    do {
      bits = rng.next_u64();
      redraws++;
    } while (bits == 0);
    const int zeros = __builtin_clzll(bits) + redraws * 64;
    u               = static_cast<float>(static_cast<double>(u) * std::pow(2.0, -zeros));
    p[i]            = static_cast<float>(std::log1p(static_cast<double>(u)) / std::log(2.0));
  }
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

wholememory_error_code_t graph_append_unique(wholememory_tensor_t target_nodes_tensor,
                                             wholememory_tensor_t neighbor_nodes_tensor,
                                             void* output_unique_node_memory_context,
                                             wholememory_tensor_t output_neighbor_raw_to_unique_mapping_tensor,
                                             wholememory_env_func_t* p_env_fns, void* stream)
{
  WM_API_BEGIN
  const auto* bk = graph_backend();
  if (bk == nullptr) return WHOLEMEMORY_NOT_SUPPORTED;
  if (p_env_fns == nullptr || output_unique_node_memory_context == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  wholememory_error_code_t err = WHOLEMEMORY_SUCCESS;
  wholememory_array_description_t target_desc, neighbor_desc;
  if (!array_of(target_nodes_tensor, "target_nodes_tensor", &target_desc, &err)) return err;
  if (!array_of(neighbor_nodes_tensor, "neighbor_nodes_tensor", &neighbor_desc, &err)) return err;
  if (target_desc.dtype != neighbor_desc.dtype) {  // append_unique.cpp:45-49
    WM_ERROR("target_nodes_dtype should be the same with neighbor_nodes_dtype");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  if (!is_index_dtype(target_desc.dtype)) {
    WM_ERROR("node ids must be int32 or int64");
    return WHOLEMEMORY_LOGIC_ERROR;
  }
  int* mapping = nullptr;
  if (output_neighbor_raw_to_unique_mapping_tensor != nullptr) {
    wholememory_array_description_t map_desc;
    if (!array_of(output_neighbor_raw_to_unique_mapping_tensor, "output_neighbor_raw_to_unique_mapping_tensor", &map_desc, &err))
      return err;
    if (map_desc.size != neighbor_desc.size) {  // append_unique.cpp:63-68
      WM_ERROR("output_neighbor_raw_to_unique_mapping size should be the same as neighbor_nodes");
      return WHOLEMEMORY_INVALID_INPUT;
    }
    if (map_desc.dtype != WHOLEMEMORY_DT_INT) {
      WM_ERROR("output_neighbor_raw_to_unique_mapping must be int32");
      return WHOLEMEMORY_LOGIC_ERROR;
    }
    mapping = static_cast<int*>(wholememory_tensor_get_data_pointer(output_neighbor_raw_to_unique_mapping_tensor));
  }
  if (target_desc.size + neighbor_desc.size >= (INT64_C(1) << 31) - 1) return WHOLEMEMORY_INVALID_INPUT;
  const int nt = static_cast<int>(target_desc.size), nn = static_cast<int>(neighbor_desc.size);
  const void* targets   = wholememory_tensor_get_data_pointer(target_nodes_tensor);
  const void* neighbors = wholememory_tensor_get_data_pointer(neighbor_nodes_tensor);
  if (nt + nn == 0) {
    (void)output_alloc(p_env_fns, output_unique_node_memory_context, 0, target_desc.dtype);
    return WHOLEMEMORY_SUCCESS;
  }
  temp_mem ws_mem(p_env_fns), host_mem(p_env_fns);
  void* ws  = ws_mem.device(static_cast<int64_t>(bk->append_unique_workspace_bytes(nt, nn, target_desc.dtype)), WHOLEMEMORY_DT_INT8);
  int* host = static_cast<int*>(host_mem.pinned(2, WHOLEMEMORY_DT_INT));   // written by the phase's last kernel
  WM_BK(bk->append_unique_phase1(targets, nt, neighbors, nn, nullptr, target_desc.dtype, ws, nullptr, host, nullptr, stream));
  WM_BK(bk->stream_sync(stream));
  const int new_count = host[1];
  void* out = output_alloc(p_env_fns, output_unique_node_memory_context, static_cast<int64_t>(nt) + new_count, target_desc.dtype);
  if (out == nullptr) return WHOLEMEMORY_OUT_OF_MEMORY;
  WM_BK(bk->append_unique_phase2(targets, nt, nn, nn, target_desc.dtype, ws, out, mapping, nullptr, nullptr, nullptr, stream));
  if (!async_completion_enabled() || debug_sync_enabled()) WM_BK(bk->stream_sync(stream));  // (reference append_unique_func.cuh:351)
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

namespace {
// offsets[0 .. n] of a hop: one launch (counts inside the scan, kernels/graph.hip: chain_scan_kernel) where the backend has
// it and takes the size, else count kernel + scan. WM_SAMPLE_FUSED_SCAN=0 forces the two steps (A/B).
int hop_offsets(const wm_device_backend* bk, const wm_sample_args& a, const int* n_dev, int* counts, int* offsets, void* scan_ws,
                size_t scan_ws_bytes, void* stream, int ws_is_ones = 0)
{
  const char* sw = WM_AB_KNOB("WM_SAMPLE_FUSED_SCAN");
  if (bk->sample_offsets != nullptr && !(sw != nullptr && sw[0] == '0')) {
    const int rc = bk->sample_offsets(&a.row_gref, a.row_storage_offset, a.centers, a.center_dtype, a.n_center, n_dev,
                                      a.max_sample_count, offsets, scan_ws, scan_ws_bytes, ws_is_ones, stream);
    if (rc != -3) return rc;
  }
  int rc = bk->sample_counts(&a.row_gref, a.row_storage_offset, nullptr, a.centers, a.center_dtype, a.n_center, n_dev,
                             a.max_sample_count, counts, stream);
  if (rc != 0) return rc;
  return bk->exclusive_scan_i32(counts, offsets, static_cast<int64_t>(a.n_center) + 1, scan_ws, scan_ws_bytes, stream);
}
}  // namespace

// One hop of multi-layer sampling as ONE call (extension; the reference runs the sampler and append_unique as two ops with a
// host round trip each to size their outputs — wholegraph_ops/unweighted_sample_without_replacement + graph_ops/append_unique,
// driven by python/.../torch/graph_structure.py:140-196). Here the sampled ids go to scratch sized for the upper bound
// n_center * max_sample_count, append_unique reads the number in use on the device, and the host learns both counts —
// samples and new unique ids — with a single synchronise. Outputs are bit-identical to the two-op sequence:
//   output_sample_offset_tensor  int32 [n_center + 1]   (caller-allocated, as in the sampler)
//   unique                       frontier ++ new neighbour ids in first-occurrence order   (append_unique's output)
//   neighbor_pos                 int32 [n_samples]: position of every sampled neighbour in `unique`
//   center_lid                   int32 [n_samples]: position of its centre in the frontier
// Mapped CSR (CONTINUOUS / CHUNKED / plain) with column ids of the frontier's dtype only; anything else answers
// WHOLEMEMORY_NOT_SUPPORTED before touching the stream and the caller takes the two-op route.
wholememory_error_code_t wholememory_ext_sample_append_unique(
  wholememory_tensor_t wm_csr_row_ptr_tensor, wholememory_tensor_t wm_csr_col_ptr_tensor,
  wholememory_tensor_t center_nodes_tensor, int max_sample_count, unsigned long long random_seed,
  wholememory_tensor_t output_sample_offset_tensor, void* output_unique_memory_context,
  void* output_neighbor_pos_memory_context, void* output_center_localid_memory_context, wholememory_env_func_t* p_env_fns,
  void* stream)
{
  WM_API_BEGIN
  const auto* bk = graph_backend();
  if (bk == nullptr) return WHOLEMEMORY_NOT_SUPPORTED;
  if (p_env_fns == nullptr || output_unique_memory_context == nullptr || output_neighbor_pos_memory_context == nullptr ||
      output_center_localid_memory_context == nullptr)
    return WHOLEMEMORY_INVALID_INPUT;
  wholememory_error_code_t err = WHOLEMEMORY_SUCCESS;
  wholememory_array_description_t row_desc, col_desc, center_desc, offset_desc;
  if (!array_of(wm_csr_row_ptr_tensor, "wm_csr_row_ptr_tensor", &row_desc, &err)) return err;
  if (!array_of(wm_csr_col_ptr_tensor, "wm_csr_col_ptr_tensor", &col_desc, &err)) return err;
  if (!array_of(center_nodes_tensor, "center_nodes_tensor", &center_desc, &err)) return err;
  if (!array_of(output_sample_offset_tensor, "output_sample_offset_tensor", &offset_desc, &err)) return err;
  const auto row_mt = memory_type_of(wm_csr_row_ptr_tensor), col_mt = memory_type_of(wm_csr_col_ptr_tensor);
  const bool mapped = row_mt != WHOLEMEMORY_MT_HIERARCHY && col_mt != WHOLEMEMORY_MT_HIERARCHY &&
                      row_mt != WHOLEMEMORY_MT_DISTRIBUTED && col_mt != WHOLEMEMORY_MT_DISTRIBUTED;
  const int64_t n = center_desc.size;
  const int64_t room = n * static_cast<int64_t>(std::max(max_sample_count, 0));
  if (!mapped || max_sample_count <= 0 || row_desc.dtype != WHOLEMEMORY_DT_INT64 || offset_desc.dtype != WHOLEMEMORY_DT_INT ||
      !is_index_dtype(center_desc.dtype) || col_desc.dtype != center_desc.dtype || offset_desc.size < n + 1 || n == 0 ||
      n + room >= (INT64_C(1) << 31) - 1)
    return WHOLEMEMORY_NOT_SUPPORTED;
  wm_sample_args a{};
  WHOLEMEMORY_RETURN_ON_FAIL(tensor_mapped_gref(wm_csr_row_ptr_tensor, &a.row_gref));
  WHOLEMEMORY_RETURN_ON_FAIL(tensor_mapped_gref(wm_csr_col_ptr_tensor, &a.col_gref));
  a.row_storage_offset = row_desc.storage_offset;
  a.col_storage_offset = col_desc.storage_offset;
  a.col_dtype          = col_desc.dtype;
  a.centers            = wholememory_tensor_get_data_pointer(center_nodes_tensor);
  a.center_dtype       = center_desc.dtype;
  a.n_center           = static_cast<int>(n);
  a.max_sample_count   = max_sample_count;
  a.random_seed        = random_seed;
  int* offsets         = static_cast<int*>(wholememory_tensor_get_data_pointer(output_sample_offset_tensor));
  a.sample_offsets     = offsets;

  const int nt = static_cast<int>(n), nn_room = static_cast<int>(room);
  temp_mem counts_mem(p_env_fns), scan_mem(p_env_fns), ids_mem(p_env_fns), lid_mem(p_env_fns), ws_mem(p_env_fns),
    host_mem(p_env_fns);
  int* counts          = static_cast<int*>(counts_mem.device(n + 1, WHOLEMEMORY_DT_INT));
  const size_t scan_ws = bk->scan_i32_workspace_bytes(n + 1);
  void* scan_ws_ptr    = scan_mem.device(static_cast<int64_t>(scan_ws), WHOLEMEMORY_DT_INT8);
  void* ids            = ids_mem.device(room, col_desc.dtype);
  int* lid             = static_cast<int*>(lid_mem.device(room, WHOLEMEMORY_DT_INT));
  void* ws = ws_mem.device(static_cast<int64_t>(bk->append_unique_workspace_bytes(nt, nn_room, center_desc.dtype)), WHOLEMEMORY_DT_INT8);
  // {samples, new unique ids}: left in pinned memory by the last kernel before the host looks (no copy commands)
  int* host = static_cast<int*>(host_mem.pinned(2, WHOLEMEMORY_DT_INT));

  WM_BK(hop_offsets(bk, a, nullptr, counts, offsets, scan_ws_ptr, scan_ws, stream));
  a.out_ids        = ids;
  a.out_center_lid = lid;
  WM_BK(bk->sample_unweighted(&a, stream));   // writes exactly offsets[n] entries of the scratch arrays
  int rc = bk->append_unique_phase1(a.centers, nt, ids, nn_room, offsets + n, center_desc.dtype, ws, nullptr, host, nullptr, stream);
  int total = 0, n_new = 0;
  if (rc == -3) {
    // a frontier too big for the route that works from a device-side count: learn the sample count first
    WM_BK(bk->memcpy_async(host, offsets + n, sizeof(int), stream));
    WM_BK(bk->stream_sync(stream));
    total = host[0];
    temp_mem ws2_mem(p_env_fns);
    void* ws2 = ws2_mem.device(static_cast<int64_t>(bk->append_unique_workspace_bytes(nt, total, center_desc.dtype)), WHOLEMEMORY_DT_INT8);
    WM_BK(bk->append_unique_phase1(a.centers, nt, ids, total, nullptr, center_desc.dtype, ws2, nullptr, host, nullptr, stream));
    WM_BK(bk->stream_sync(stream));
    n_new = host[1];
    void* uniq = output_alloc(p_env_fns, output_unique_memory_context, static_cast<int64_t>(nt) + n_new, center_desc.dtype);
    int* pos   = static_cast<int*>(output_alloc(p_env_fns, output_neighbor_pos_memory_context, total, WHOLEMEMORY_DT_INT));
    int* olid  = static_cast<int*>(output_alloc(p_env_fns, output_center_localid_memory_context, total, WHOLEMEMORY_DT_INT));
    if (uniq == nullptr || (total > 0 && (pos == nullptr || olid == nullptr))) return WHOLEMEMORY_OUT_OF_MEMORY;
    WM_BK(bk->append_unique_phase2(a.centers, nt, total, total, center_desc.dtype, ws2, uniq, pos, lid, olid, nullptr, stream));
    WM_BK(bk->stream_sync(stream));   // ws2 goes out of scope here
    return WHOLEMEMORY_SUCCESS;
  }
  if (rc != 0) return rc == -1 ? WHOLEMEMORY_LOGIC_ERROR : WHOLEMEMORY_CUDA_ERROR;
  WM_BK(bk->stream_sync(stream));             // the only host round trip of the hop
  total = host[0], n_new = host[1];
  void* uniq = output_alloc(p_env_fns, output_unique_memory_context, static_cast<int64_t>(nt) + n_new, center_desc.dtype);
  int* pos   = static_cast<int*>(output_alloc(p_env_fns, output_neighbor_pos_memory_context, total, WHOLEMEMORY_DT_INT));
  int* olid  = static_cast<int*>(output_alloc(p_env_fns, output_center_localid_memory_context, total, WHOLEMEMORY_DT_INT));
  if (uniq == nullptr || (total > 0 && (pos == nullptr || olid == nullptr))) return WHOLEMEMORY_OUT_OF_MEMORY;
  WM_BK(bk->append_unique_phase2(a.centers, nt, nn_room, total, center_desc.dtype, ws, uniq, pos, lid, olid, nullptr, stream));
  if (!async_completion_enabled() || debug_sync_enabled()) WM_BK(bk->stream_sync(stream));   // else: outputs and scratch are ordered on `stream`
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

// The whole chain of hops of GraphStructure.multilayer_sample_without_replacement as ONE call with NO host round trip inside
// (extension; the reference pays two per hop, the fused hop above one): every array is sized by the CALLER for its upper
// bound — hop h has at most cap_c[h] centres (cap_c[0] = seeds, cap_c[h + 1] = cap_c[h] + cap_s[h]) and cap_s[h] = cap_c[h] x
// fan-out samples — the counts stay on the device from hop to hop (wm_sample_args::n_center_dev, wm_au_bounds), and the last
// kernel of every hop leaves {samples, new unique ids} in counts_host[2h], counts_host[2h + 1] (pinned memory). The caller
// synchronises the stream ONCE, reads the counts and trims:
//   sample_offsets[h]  int32 [cap_c[h] + 1]          first n_c[h] + 1 entries are the hop's csr_row_ptr
//   unique[h]          ids   [cap_c[h] + cap_s[h]]    first n_c[h] + new[h] entries = centres ++ new neighbours = hop h + 1's centres
//                                                      (last hop: the entries behind them are -1 up to the array's room, so the
//                                                      whole array can feed a gather before the host has read the counts)
//   neighbor_pos[h]    int32 [cap_s[h]]               first samples[h] entries
//   center_lid[h]      int32 [cap_s[h]]               first samples[h] entries
// with n_c[0] = seeds, n_c[h + 1] = n_c[h] + new[h]. Outputs equal those of `hops` fused-hop calls bit for bit (same kernels,
// same per-hop seeds). WHOLEMEMORY_NOT_SUPPORTED (nothing queued): CSR not mapped into this rank, dtypes differ, an empty seed
// array, a fan-out <= 0, or a hop whose upper bounds are too big for the hash-table route of append_unique.
wholememory_error_code_t wholememory_ext_multilayer_sample(
  wholememory_tensor_t wm_csr_row_ptr_tensor, wholememory_tensor_t wm_csr_col_ptr_tensor, wholememory_tensor_t seed_nodes_tensor,
  int hops, const int* max_sample_counts, const unsigned long long* random_seeds, void* const* sample_offsets, void* const* unique,
  int* const* neighbor_pos, int* const* center_lid, int* counts_host, wholememory_env_func_t* p_env_fns, void* stream)
{
  WM_API_BEGIN
  const auto* bk = graph_backend();
  if (bk == nullptr || bk->append_unique_takes_bounds == nullptr) return WHOLEMEMORY_NOT_SUPPORTED;
  // sample_offsets == nullptr: a QUERY — would this chain be taken? (SUCCESS / NOT_SUPPORTED, nothing queued, no buffer needed:
  // the caller asks before it allocates the upper-bound buffers)
  const bool query = sample_offsets == nullptr;
  if (hops <= 0 || hops > 16 || max_sample_counts == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (!query && (p_env_fns == nullptr || random_seeds == nullptr || unique == nullptr || neighbor_pos == nullptr ||
                 center_lid == nullptr || counts_host == nullptr))
    return WHOLEMEMORY_INVALID_INPUT;
  wholememory_error_code_t err = WHOLEMEMORY_SUCCESS;
  wholememory_array_description_t row_desc, col_desc, seed_desc;
  if (!array_of(wm_csr_row_ptr_tensor, "wm_csr_row_ptr_tensor", &row_desc, &err)) return err;
  if (!array_of(wm_csr_col_ptr_tensor, "wm_csr_col_ptr_tensor", &col_desc, &err)) return err;
  if (!array_of(seed_nodes_tensor, "seed_nodes_tensor", &seed_desc, &err)) return err;
  const auto row_mt = memory_type_of(wm_csr_row_ptr_tensor), col_mt = memory_type_of(wm_csr_col_ptr_tensor);
  const bool mapped = row_mt != WHOLEMEMORY_MT_HIERARCHY && col_mt != WHOLEMEMORY_MT_HIERARCHY &&
                      row_mt != WHOLEMEMORY_MT_DISTRIBUTED && col_mt != WHOLEMEMORY_MT_DISTRIBUTED;
  if (!mapped || row_desc.dtype != WHOLEMEMORY_DT_INT64 || !is_index_dtype(seed_desc.dtype) || col_desc.dtype != seed_desc.dtype ||
      seed_desc.size == 0)
    return WHOLEMEMORY_NOT_SUPPORTED;
  std::vector<int64_t> cap_c(hops + 1), cap_s(hops);
  cap_c[0] = seed_desc.size;
  for (int h = 0; h < hops; h++) {
    if (max_sample_counts[h] <= 0) return WHOLEMEMORY_NOT_SUPPORTED;
    cap_s[h]     = cap_c[h] * max_sample_counts[h];
    cap_c[h + 1] = cap_c[h] + cap_s[h];
    if (cap_c[h + 1] >= (INT64_C(1) << 31) - 1 ||
        !bk->append_unique_takes_bounds(static_cast<int>(cap_c[h]), static_cast<int>(cap_s[h]), seed_desc.dtype))
      return WHOLEMEMORY_NOT_SUPPORTED;
  }
  if (query) return WHOLEMEMORY_SUCCESS;
  wm_sample_args a{};
  WHOLEMEMORY_RETURN_ON_FAIL(tensor_mapped_gref(wm_csr_row_ptr_tensor, &a.row_gref));
  WHOLEMEMORY_RETURN_ON_FAIL(tensor_mapped_gref(wm_csr_col_ptr_tensor, &a.col_gref));
  a.row_storage_offset = row_desc.storage_offset;
  a.col_storage_offset = col_desc.storage_offset;
  a.col_dtype          = col_desc.dtype;
  a.center_dtype       = seed_desc.dtype;

  // scratch of all hops at once (the kernels are queued back to back; nothing may be reused before the last one has run)
  temp_mem n_dev_mem(p_env_fns);
  int* n_dev = static_cast<int*>(n_dev_mem.device(hops, WHOLEMEMORY_DT_INT));   // centres of hop h + 1 = unique ids after hop h
  std::vector<std::unique_ptr<temp_mem>> keep;
  auto scratch = [&](int64_t count, wholememory_dtype_t dt) {
    keep.emplace_back(new temp_mem(p_env_fns));
    return keep.back()->device(count, dt);
  };
  // the offsets scans' workspaces of all hops in one block, initialised (0xFF: the scans' state) by ONE fill command
  std::vector<size_t> scan_bytes(hops), scan_at(hops);
  size_t scan_total = 0;
  for (int h = 0; h < hops; h++) {
    scan_bytes[h] = bk->scan_i32_workspace_bytes(cap_c[h] + 1);
    scan_at[h]    = scan_total;
    scan_total += scan_bytes[h];
  }
  char* scan_block = static_cast<char*>(scratch(static_cast<int64_t>(scan_total), WHOLEMEMORY_DT_INT8));
  if (bk->fill_ff_async != nullptr)
    WM_BK(bk->fill_ff_async(scan_block, scan_total, stream));
  else
    WM_BK(bk->memset_async(scan_block, 0xFF, scan_total, stream));
  for (int h = 0; h < hops; h++) {
    const int nc = static_cast<int>(cap_c[h]), ns = static_cast<int>(cap_s[h]);
    const int* centres_in_use = h == 0 ? nullptr : n_dev + (h - 1);
    a.centers          = h == 0 ? wholememory_tensor_get_data_pointer(seed_nodes_tensor) : unique[h - 1];
    a.n_center         = nc;
    a.n_center_dev     = centres_in_use;
    a.max_sample_count = max_sample_counts[h];
    a.random_seed      = random_seeds[h];
    int* offsets       = static_cast<int*>(sample_offsets[h]);
    a.sample_offsets   = offsets;
    int* counts        = static_cast<int*>(scratch(nc + 1, WHOLEMEMORY_DT_INT));
    const size_t scan_ws = scan_bytes[h];
    void* scan_ws_ptr  = scan_block + scan_at[h];
    void* ids          = scratch(ns, col_desc.dtype);
    void* ws = scratch(static_cast<int64_t>(bk->append_unique_workspace_bytes(nc, ns, seed_desc.dtype)), WHOLEMEMORY_DT_INT8);
    WM_BK(hop_offsets(bk, a, centres_in_use, counts, offsets, scan_ws_ptr, scan_ws, stream, 1));   // offsets[nc] = samples of the hop
    a.out_ids        = ids;
    a.out_center_lid = center_lid[h];
    // the sampling kernel empties the hop's hash table on the side (one fill command fewer per hop)
    a.fill_ff_ptr = nullptr, a.fill_ff_bytes = 0;
    const bool side_fill = bk->append_unique_table_region != nullptr &&
                           bk->append_unique_table_region(nc, ns, seed_desc.dtype, ws, &a.fill_ff_ptr, &a.fill_ff_bytes) == 0;
    if (!side_fill) a.fill_ff_ptr = nullptr, a.fill_ff_bytes = 0;
    WM_BK(bk->sample_unweighted(&a, stream));
    // (the hop's counts are published by phase 2's emitting kernel: one tiny launch fewer per hop)
    // (the outermost frontier is padded with -1 up to its room: a gather can be queued on the whole array before the host
    // has read the counts — negative ids are skipped, gather_scatter_func.cuh:296)
    wm_au_bounds b{centres_in_use, offsets + nc, n_dev + h, counts_host + 2 * h, side_fill ? 1 : 0, h == hops - 1 ? 1 : 0};
    int rc = bk->append_unique_phase1(a.centers, nc, ids, ns, offsets + nc, seed_desc.dtype, ws, nullptr, nullptr, &b, stream);
    if (rc != 0) return rc == -1 ? WHOLEMEMORY_LOGIC_ERROR : WHOLEMEMORY_CUDA_ERROR;
    WM_BK(bk->append_unique_phase2(a.centers, nc, ns, ns, seed_desc.dtype, ws, unique[h], neighbor_pos[h], nullptr, nullptr, &b, stream));
  }
  // by default the call returns complete, like every op of the reference; a host framework that declared stream-ordered
  // allocators (wholememory_ext_set_async_completion) gets it back with everything queued and synchronises when it reads
  // counts_host
  if (!async_completion_enabled() || debug_sync_enabled()) WM_BK(bk->stream_sync(stream));
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

wholememory_error_code_t csr_add_self_loop(wholememory_tensor_t csr_row_ptr_tensor, wholememory_tensor_t csr_col_ptr_tensor,
                                           wholememory_tensor_t output_csr_row_ptr_tensor,
                                           wholememory_tensor_t output_csr_col_ptr_tensor, void* stream)
{
  WM_API_BEGIN
  const auto* bk = graph_backend();
  if (bk == nullptr) return WHOLEMEMORY_NOT_SUPPORTED;
  wholememory_error_code_t err = WHOLEMEMORY_SUCCESS;
  wholememory_array_description_t row_desc, col_desc, out_row_desc, out_col_desc;
  if (!array_of(csr_row_ptr_tensor, "csr_row_ptr_tensor", &row_desc, &err)) return err;
  if (!array_of(csr_col_ptr_tensor, "csr_col_ptr_tensor", &col_desc, &err)) return err;
  if (!array_of(output_csr_row_ptr_tensor, "output_csr_row_ptr_tensor", &out_row_desc, &err)) return err;
  if (!array_of(output_csr_col_ptr_tensor, "output_csr_col_ptr_tensor", &out_col_desc, &err)) return err;
  // int32 only: csr_add_self_loop.cpp:32-63
  if (row_desc.dtype != WHOLEMEMORY_DT_INT || col_desc.dtype != WHOLEMEMORY_DT_INT || out_row_desc.dtype != WHOLEMEMORY_DT_INT ||
      out_col_desc.dtype != WHOLEMEMORY_DT_INT) {
    WM_ERROR("csr_add_self_loop works on int32 CSR arrays");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  // the reference launches without looking at the output sizes; writing past a short output is refused here
  if (row_desc.size < 1 || out_row_desc.size < row_desc.size || out_col_desc.size < col_desc.size + row_desc.size - 1) {
    WM_ERROR("csr_add_self_loop outputs need %ld row and %ld col entries", static_cast<long>(row_desc.size),
             static_cast<long>(col_desc.size + row_desc.size - 1));
    return WHOLEMEMORY_INVALID_INPUT;
  }
  WM_BK(bk->csr_add_self_loop(static_cast<const int*>(wholememory_tensor_get_data_pointer(csr_row_ptr_tensor)),
                              static_cast<const int*>(wholememory_tensor_get_data_pointer(csr_col_ptr_tensor)),
                              static_cast<int*>(wholememory_tensor_get_data_pointer(output_csr_row_ptr_tensor)),
                              static_cast<int*>(wholememory_tensor_get_data_pointer(output_csr_col_ptr_tensor)),
                              static_cast<int>(row_desc.size - 1), stream));
  WM_BK(bk->stream_sync(stream));
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

// Self-test of the env functions (reference wholememory_op.h:58-79, wholememory_test_op.cu:60-160): computes
// out[i, :] = T(float(i)) + input[:] into a scratch buffer obtained through p_env_fns->temporary_fns, then copies it to
// the fixed output tensor and to device / pinned / host outputs allocated through p_env_fns->output_fns for every
// non-null memory context.
wholememory_error_code_t wholememory_env_test_op(wholememory_tensor_t input_tensor, wholememory_tensor_t output_fixed_tensor,
                                                 void* output_variable_device_tensor_handle,
                                                 void* output_variable_pinned_tensor_handle,
                                                 void* output_variable_host_tensor_handle, int64_t output_variable_entry_count,
                                                 wholememory_env_func_t* p_env_fns, void* stream)
{
  WM_API_BEGIN
  const auto* bk = backend();
  if (bk->env_test_fill == nullptr) return WHOLEMEMORY_NOT_SUPPORTED;
  if (input_tensor == nullptr || output_fixed_tensor == nullptr || p_env_fns == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  auto in_desc  = *wholememory_tensor_get_tensor_description(input_tensor);
  auto out_desc = *wholememory_tensor_get_tensor_description(output_fixed_tensor);
  if (in_desc.dim != 1 || out_desc.dim != 2 || out_desc.sizes[0] != output_variable_entry_count ||
      out_desc.sizes[1] != in_desc.sizes[0] || in_desc.dtype != out_desc.dtype)
    return WHOLEMEMORY_INVALID_INPUT;  // the reference aborts on these (WHOLEMEMORY_CHECK_NOTHROW)
  const int64_t dim = in_desc.sizes[0], n = output_variable_entry_count;
  temp_mem scratch(p_env_fns);
  void* tmp = scratch.device(n * dim, in_desc.dtype);
  int rc    = bk->env_test_fill(wholememory_tensor_get_data_pointer(input_tensor), tmp, in_desc.dtype, dim, n, dim, stream);
  if (rc == -1) return WHOLEMEMORY_INVALID_INPUT;
  if (rc != 0) return WHOLEMEMORY_CUDA_ERROR;
  const size_t bytes = static_cast<size_t>(n) * dim * wholememory_dtype_get_element_size(in_desc.dtype);
  auto out_alloc = [&](void* ctx, wholememory_memory_allocation_type_t type) -> void* {
    if (ctx == nullptr) return nullptr;
    auto d = out_desc;
    d.strides[0] = dim, d.strides[1] = 1, d.storage_offset = 0;
    return p_env_fns->output_fns.malloc_fn(&d, type, ctx, p_env_fns->output_fns.global_context);
  };
  void* dsts[4] = {wholememory_tensor_get_data_pointer(output_fixed_tensor),
                   out_alloc(output_variable_device_tensor_handle, WHOLEMEMORY_MA_DEVICE),
                   out_alloc(output_variable_pinned_tensor_handle, WHOLEMEMORY_MA_PINNED),
                   out_alloc(output_variable_host_tensor_handle, WHOLEMEMORY_MA_HOST)};
  for (void* dst : dsts)
    if (dst != nullptr && bytes > 0) WM_BK(bk->memcpy_async(dst, tmp, bytes, stream));
  WM_BK(bk->stream_sync(stream));  // the scratch buffer returns to the caller's allocator
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

}  // extern "C"
