// wholegraph_amd — WholeMemory tensor objects (host only): a descriptor over a handle or a caller
// pointer, subtensor views, and the row-partition queries the ops layer builds on.
// Behaviour follows reference cpp/src/wholememory/wholememory_tensor.cpp:38-464.
#include <atomic>
#include <algorithm>

#include <wholememory/wholememory_tensor.h>

#include "wm_common.hpp"

struct wholememory_tensor_ {
  wholememory_handle_t handle = nullptr;  // handle-backed ...
  void* storage_ptr           = nullptr;  // ... or pointer-backed
  wholememory_tensor_description_t desc;
  wholememory_tensor_t root = nullptr;
  bool is_wholememory       = false;
  bool owns_handle          = false;
};

namespace {
std::atomic<int64_t> g_live_tensors{0};

// row size in bytes of the ROOT allocation (1-D: one element)
size_t root_entry_bytes(wholememory_tensor_t t)
{
  const auto& rd = t->root->desc;
  size_t es      = wholememory_dtype_get_element_size(t->desc.dtype);
  return rd.dim == 2 ? es * static_cast<size_t>(rd.strides[0]) : es;
}
}  // namespace

extern "C" {

int64_t get_wholememory_tensor_count() { return g_live_tensors.load(); }

wholememory_error_code_t wholememory_create_tensor(wholememory_tensor_t* p_tensor,
                                                   wholememory_tensor_description_t* desc,
                                                   wholememory_comm_t comm,
                                                   wholememory_memory_type_t memory_type,
                                                   wholememory_memory_location_t memory_location,
                                                   size_t* tensor_entry_partition)
{
  if (p_tensor == nullptr || desc == nullptr) {
    WM_ERROR("wholememory_create_tensor: null argument");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  if (desc->dim <= 0 || desc->dim > 2 || desc->storage_offset != 0 || desc->strides[desc->dim - 1] != 1 ||
      desc->dtype <= WHOLEMEMORY_DT_UNKNOWN || desc->dtype >= WHOLEMEMORY_DT_COUNT) {
    WM_ERROR("wholememory_create_tensor: need a 1-D/2-D, offset-0, unit-inner-stride tensor of a known dtype");
    return WHOLEMEMORY_INVALID_INPUT;
  }
  const size_t es          = wholememory_dtype_get_element_size(desc->dtype);
  const size_t malloc_size = static_cast<size_t>(wholememory_get_memory_element_count_from_tensor(desc)) * es;
  const size_t granularity = es * static_cast<size_t>(desc->strides[0]);  // a row is never split
  auto* t                  = new wholememory_tensor_();
  t->desc                  = *desc;
  t->is_wholememory        = true;
  t->owns_handle           = true;
  t->root                  = t;
  auto rc = wholememory_malloc(&t->handle, malloc_size, comm, memory_type, memory_location, granularity, tensor_entry_partition);
  if (rc != WHOLEMEMORY_SUCCESS) {
    delete t;
    return rc;
  }
  g_live_tensors++;
  *p_tensor = t;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_destroy_tensor(wholememory_tensor_t t)
{
  if (t == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (t->owns_handle && t->is_wholememory) WHOLEMEMORY_RETURN_ON_FAIL(wholememory_free(t->handle));
  g_live_tensors--;
  delete t;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_make_tensor_from_pointer(wholememory_tensor_t* p_tensor,
                                                              void* storage_ptr,
                                                              wholememory_tensor_description_t* desc)
{
  if (p_tensor == nullptr || desc == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  // a null pointer or a 0-dim description is wrapped unchecked, exactly like the reference
  // (wholememory_tensor.cpp:128-139); everything else must be unit-inner-stride and of a known dtype
  if (storage_ptr != nullptr && desc->dim != 0) {
    if (desc->dim < 0 || desc->dim > WHOLEMEMORY_MAX_TENSOR_DIM) return WHOLEMEMORY_INVALID_INPUT;
    if (desc->strides[desc->dim - 1] != 1) return WHOLEMEMORY_INVALID_INPUT;
    if (desc->dtype <= WHOLEMEMORY_DT_UNKNOWN || desc->dtype >= WHOLEMEMORY_DT_COUNT) return WHOLEMEMORY_INVALID_INPUT;
  }
  auto* t           = new wholememory_tensor_();
  t->storage_ptr    = storage_ptr;
  t->desc           = *desc;
  t->is_wholememory = false;
  t->owns_handle    = false;
  t->root           = t;
  g_live_tensors++;
  *p_tensor = t;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_make_tensor_from_handle(wholememory_tensor_t* p_tensor,
                                                             wholememory_handle_t handle,
                                                             wholememory_tensor_description_t* desc)
{
  if (p_tensor == nullptr || handle == nullptr || desc == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (desc->dim <= 0 || desc->dim > 2 || desc->strides[desc->dim - 1] != 1) return WHOLEMEMORY_INVALID_INPUT;
  if (desc->dtype <= WHOLEMEMORY_DT_UNKNOWN || desc->dtype >= WHOLEMEMORY_DT_COUNT) return WHOLEMEMORY_INVALID_INPUT;
  auto* t           = new wholememory_tensor_();
  t->handle         = handle;
  t->desc           = *desc;
  t->is_wholememory = true;
  t->owns_handle    = false;
  t->root           = t;
  g_live_tensors++;
  *p_tensor = t;
  return WHOLEMEMORY_SUCCESS;
}

bool wholememory_tensor_has_handle(wholememory_tensor_t t) { return t->is_wholememory; }
wholememory_handle_t wholememory_tensor_get_memory_handle(wholememory_tensor_t t)
{
  return t->is_wholememory ? t->handle : nullptr;
}
wholememory_tensor_description_t* wholememory_tensor_get_tensor_description(wholememory_tensor_t t) { return &t->desc; }

wholememory_error_code_t wholememory_tensor_get_global_reference(wholememory_tensor_t t, wholememory_gref_t* gref)
{
  if (t == nullptr || gref == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (t->is_wholememory) return wholememory_get_global_reference(gref, t->handle);
  *gref = wholememory_create_continuous_global_reference(t->storage_ptr);
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_tensor_map_local_tensor(wholememory_tensor_t t, wholememory_tensor_t* local_tensor)
{
  // the view may drop rows at the tail but not at the front (reference wholememory_tensor.cpp:238)
  if (t == nullptr || local_tensor == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (!t->is_wholememory) return WHOLEMEMORY_INVALID_VALUE;
  const auto& d = t->desc;
  if (d.dim != 1 && d.dim != 2) return WHOLEMEMORY_INVALID_VALUE;
  if (d.dim == 1 && d.storage_offset != 0) return WHOLEMEMORY_INVALID_VALUE;
  if (d.dim == 2 && d.storage_offset + d.sizes[1] > d.strides[0]) return WHOLEMEMORY_INVALID_VALUE;
  void* local_ptr;
  size_t local_size, local_offset;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_local_memory(&local_ptr, &local_size, &local_offset, t->handle));
  const size_t es   = wholememory_dtype_get_element_size(d.dtype);
  const size_t gran = d.dim == 1 ? es : es * static_cast<size_t>(d.strides[0]);
  const size_t view_bytes = static_cast<size_t>(d.sizes[0]) * gran;
  local_size = view_bytes > local_offset ? std::min(local_size, view_bytes - local_offset) : 0;
  if (local_size % gran != 0) return WHOLEMEMORY_LOGIC_ERROR;
  wholememory_tensor_description_t ld = d;
  ld.sizes[0]                         = static_cast<int64_t>(local_size / gran);
  return wholememory_make_tensor_from_pointer(local_tensor, local_ptr, &ld);
}

void* wholememory_tensor_get_data_pointer(wholememory_tensor_t t)
{
  char* base = nullptr;
  if (!t->is_wholememory) {
    base = static_cast<char*>(t->storage_ptr);
  } else {
    if (wholememory_get_memory_type(t->handle) != WHOLEMEMORY_MT_CONTINUOUS) return nullptr;
    if (wholememory_get_global_pointer(reinterpret_cast<void**>(&base), t->handle) != WHOLEMEMORY_SUCCESS) return nullptr;
  }
  return base + wholememory_dtype_get_element_size(t->desc.dtype) * static_cast<size_t>(t->desc.storage_offset);
}

wholememory_error_code_t wholememory_tensor_get_entry_offsets(size_t* entry_offsets, wholememory_tensor_t t)
{
  if (entry_offsets == nullptr || t == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  WM_CHECK_ABORT(t->root->desc.dim == 1 || t->root->desc.dim == 2, "root tensor must be 1-D or 2-D");
  if (!t->is_wholememory) {
    entry_offsets[0] = 0;
    entry_offsets[1] = static_cast<size_t>(t->root->desc.sizes[0]);
    return WHOLEMEMORY_SUCCESS;
  }
  wholememory_comm_t comm;
  int W;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_communicator(&comm, t->handle));
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_communicator_get_size(&W, comm));
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_rank_partition_offsets(entry_offsets, t->handle));
  const size_t eb = root_entry_bytes(t);
  for (int i = 0; i <= W; i++) {
    WM_CHECK_ABORT(entry_offsets[i] % eb == 0, "partition offset %zu is not a whole number of rows (%zu B)", entry_offsets[i], eb);
    entry_offsets[i] /= eb;
  }
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_tensor_get_entry_partition_sizes(size_t* entry_partition, wholememory_tensor_t t)
{
  if (entry_partition == nullptr || t == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  WM_CHECK_ABORT(t->root->desc.dim == 1 || t->root->desc.dim == 2, "root tensor must be 1-D or 2-D");
  if (!t->is_wholememory) {
    entry_partition[0] = static_cast<size_t>(t->root->desc.sizes[0]);
    return WHOLEMEMORY_SUCCESS;
  }
  wholememory_comm_t comm;
  int W;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_communicator(&comm, t->handle));
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_communicator_get_size(&W, comm));
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_rank_partition_sizes(entry_partition, t->handle));
  const size_t eb = root_entry_bytes(t);
  for (int i = 0; i < W; i++) {
    WM_CHECK_ABORT(entry_partition[i] % eb == 0, "partition size %zu is not a whole number of rows (%zu B)", entry_partition[i], eb);
    entry_partition[i] /= eb;
  }
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_tensor_get_local_entry_count(size_t* local_entry_count, wholememory_tensor_t t)
{
  if (local_entry_count == nullptr || t == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (!t->is_wholememory) {
    *local_entry_count = static_cast<size_t>(t->root->desc.sizes[0]);
    return WHOLEMEMORY_SUCCESS;
  }
  size_t bytes;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_local_size(&bytes, t->handle));
  const size_t eb = root_entry_bytes(t);
  WM_CHECK_ABORT(bytes % eb == 0, "local size is not a whole number of rows");
  *local_entry_count = bytes / eb;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_tensor_get_local_entry_start(size_t* local_entry_start, wholememory_tensor_t t)
{
  if (local_entry_start == nullptr || t == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (!t->is_wholememory) {
    *local_entry_start = 0;
    return WHOLEMEMORY_SUCCESS;
  }
  size_t bytes;
  WHOLEMEMORY_RETURN_ON_FAIL(wholememory_get_local_offset(&bytes, t->handle));
  const size_t eb = root_entry_bytes(t);
  WM_CHECK_ABORT(bytes % eb == 0, "local offset is not a whole number of rows");
  *local_entry_start = bytes / eb;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_tensor_get_subtensor(wholememory_tensor_t t,
                                                          int64_t* starts,
                                                          int64_t* ends,
                                                          wholememory_tensor_t* p_sub)
{
  if (t == nullptr || starts == nullptr || ends == nullptr || p_sub == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  const int dim = t->desc.dim;
  if (dim > 2) return WHOLEMEMORY_NOT_IMPLEMENTED;
  int64_t new_offset = t->desc.storage_offset;
  int64_t new_sizes[2] = {0, 0};
  for (int i = 0; i < dim; i++) {
    const int64_t s = starts[i] == -1 ? 0 : starts[i];
    const int64_t e = ends[i] == -1 ? t->desc.sizes[i] : ends[i];
    if (e <= s || s >= t->desc.sizes[i] || e <= 0) return WHOLEMEMORY_INVALID_INPUT;
    new_offset += t->desc.strides[i] * s;
    new_sizes[i] = e - s;
  }
  auto* sub        = new wholememory_tensor_(*t);
  sub->owns_handle = false;
  sub->desc.storage_offset = new_offset;
  for (int i = 0; i < dim; i++) sub->desc.sizes[i] = new_sizes[i];
  g_live_tensors++;
  *p_sub = sub;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_tensor_t wholememory_tensor_get_root(wholememory_tensor_t t) { return t->root; }

}  // extern "C"
