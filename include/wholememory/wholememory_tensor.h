/*
 * wholegraph_amd — WholeMemory tensors: a descriptor bound to a WholeMemory handle or to a plain
 * caller pointer, plus row ("entry") partition queries. Replaces reference
 * cpp/include/wholememory/wholememory_tensor.h:30-194.
 */
#ifndef WHOLEMEMORY_WHOLEMEMORY_TENSOR_H_
#define WHOLEMEMORY_WHOLEMEMORY_TENSOR_H_

#include <wholememory/tensor_description.h>
#include <wholememory/wholememory.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wholememory_tensor_* wholememory_tensor_t;

/* Collective. 1-D or 2-D, storage_offset 0, innermost stride 1; the row stride is the partition
 * granularity. reference wholememory_tensor.h:43-49 */
enum wholememory_error_code_t wholememory_create_tensor(
  wholememory_tensor_t* wholememory_tensor,
  struct wholememory_tensor_description_t* tensor_description,
  wholememory_comm_t comm,
  enum wholememory_memory_type_t memory_type,
  enum wholememory_memory_location_t memory_location,
  size_t* tensor_entry_partition WM_DEFAULT(nullptr));
enum wholememory_error_code_t wholememory_destroy_tensor(wholememory_tensor_t wholememory_tensor);

/* Non-owning wrappers. reference wholememory_tensor.h:65-81 */
enum wholememory_error_code_t wholememory_make_tensor_from_pointer(
  wholememory_tensor_t* wholememory_tensor,
  void* storage_ptr,
  struct wholememory_tensor_description_t* tensor_description);
enum wholememory_error_code_t wholememory_make_tensor_from_handle(
  wholememory_tensor_t* wholememory_tensor,
  wholememory_handle_t wholememory_handle,
  struct wholememory_tensor_description_t* tensor_description);

/* reference wholememory_tensor.h:88-121 */
bool wholememory_tensor_has_handle(wholememory_tensor_t wholememory_tensor);
wholememory_handle_t wholememory_tensor_get_memory_handle(wholememory_tensor_t wholememory_tensor);
struct wholememory_tensor_description_t* wholememory_tensor_get_tensor_description(
  wholememory_tensor_t wholememory_tensor);
enum wholememory_error_code_t wholememory_tensor_get_global_reference(
  wholememory_tensor_t wholememory_tensor, struct wholememory_gref_t* wholememory_gref);
/* this rank's shard as a plain (pointer-backed) tensor; caller destroys it */
enum wholememory_error_code_t wholememory_tensor_map_local_tensor(
  wholememory_tensor_t wholememory_tensor, wholememory_tensor_t* local_tensor);
/* pointer to element [storage_offset]; NULL for handle-backed tensors that are not CONTINUOUS */
void* wholememory_tensor_get_data_pointer(wholememory_tensor_t wholememory_tensor);

/* Row partition in ENTRIES (rows of the root tensor). reference wholememory_tensor.h:137-171 */
enum wholememory_error_code_t wholememory_tensor_get_entry_offsets(
  size_t* entry_offsets /* [world_size + 1] */, wholememory_tensor_t wholememory_tensor);
enum wholememory_error_code_t wholememory_tensor_get_entry_partition_sizes(
  size_t* entry_partition /* [world_size] */, wholememory_tensor_t wholememory_tensor);
enum wholememory_error_code_t wholememory_tensor_get_local_entry_count(
  size_t* local_entry_count, wholememory_tensor_t wholememory_tensor);
enum wholememory_error_code_t wholememory_tensor_get_local_entry_start(
  size_t* local_entry_start, wholememory_tensor_t wholememory_tensor);

/* View [starts, ends) per dim, -1 = open end. reference wholememory_tensor.h:181-185 */
enum wholememory_error_code_t wholememory_tensor_get_subtensor(
  wholememory_tensor_t wholememory_tensor,
  int64_t* starts,
  int64_t* ends,
  wholememory_tensor_t* sub_wholememory_tensor);
wholememory_tensor_t wholememory_tensor_get_root(wholememory_tensor_t wholememory_tensor);

/* live tensor objects (leak check used by the reference tests, wholememory_tensor.h:193-194) */
#define WM_TENSOR_COUNT_DEBUG
int64_t get_wholememory_tensor_count();

#ifdef __cplusplus
}
#endif
#endif
