/*
 * wholegraph_amd — core C ABI: library init, communicators, WholeMemory handles.
 * Replaces reference cpp/include/wholememory/wholememory.h:32-485 for the embedding
 * gather/scatter path. Same symbols, same enum values, same argument meaning; the implementation
 * underneath is HIP + RCCL (one process per MI355X, xGMI peers), not CUDA + NCCL.
 *
 * Default arguments of the reference's C++ view of this header are reproduced under __cplusplus.
 */
#ifndef WHOLEMEMORY_WHOLEMEMORY_H_
#define WHOLEMEMORY_WHOLEMEMORY_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdio.h>

#include <wholememory/global_reference.h>

#ifdef __cplusplus
#define WM_DEFAULT(x) = x
extern "C" {
#else
#define WM_DEFAULT(x)
#endif

/* reference wholememory.h:32-44 — returned by every entry point; 0 == success */
enum wholememory_error_code_t {
  WHOLEMEMORY_SUCCESS = 0,
  WHOLEMEMORY_UNKNOW_ERROR,
  WHOLEMEMORY_NOT_IMPLEMENTED,
  WHOLEMEMORY_LOGIC_ERROR,
  WHOLEMEMORY_CUDA_ERROR, /* name kept for ABI; on this build it means "HIP runtime error" */
  WHOLEMEMORY_COMMUNICATION_ERROR,
  WHOLEMEMORY_INVALID_INPUT,
  WHOLEMEMORY_INVALID_VALUE,
  WHOLEMEMORY_OUT_OF_MEMORY,
  WHOLEMEMORY_NOT_SUPPORTED,
  WHOLEMEMORY_SYSTEM_ERROR,
};

/* reference wholememory.h:46-55 */
#define WHOLEMEMORY_RETURN_ON_FAIL(X)                                                    \
  do {                                                                                   \
    enum wholememory_error_code_t wm_err__ = (X);                                        \
    if (wm_err__ != WHOLEMEMORY_SUCCESS) {                                               \
      fprintf(stderr, "File %s line %d %s failed.\n", __FILE__, __LINE__, #X);           \
      return wm_err__;                                                                   \
    }                                                                                    \
  } while (0)

/* reference wholememory.h:62-68 — how other ranks' rows are addressed */
enum wholememory_memory_type_t {
  WHOLEMEMORY_MT_NONE = 0,
  WHOLEMEMORY_MT_CONTINUOUS,  /* all shards mapped into one flat VA range (HIP VMM) */
  WHOLEMEMORY_MT_CHUNKED,     /* one mapped base pointer per rank (hipIpc) */
  WHOLEMEMORY_MT_DISTRIBUTED, /* peers not mapped; RCCL all-to-all-v moves ids and rows */
  WHOLEMEMORY_MT_HIERARCHY,   /* owned as DISTRIBUTED; gathers go node-local relay -> cross-node rail (multi-node) */
};

/* reference wholememory.h:75-79 */
enum wholememory_memory_location_t {
  WHOLEMEMORY_ML_NONE = 0,
  WHOLEMEMORY_ML_DEVICE, /* HBM */
  WHOLEMEMORY_ML_HOST,   /* pinned host memory, device-mapped */
};

/* reference wholememory.h:81-85 — only the collective (RCCL) backend exists on this build */
enum wholememory_distributed_backend_t {
  WHOLEMEMORY_DB_NONE = 0,
  WHOLEMEMORY_DB_NCCL,
  WHOLEMEMORY_DB_NVSHMEM,
};

/* reference wholememory.h:86-93 */
enum LogLevel { LEVEL_FATAL = 0, LEVEL_ERROR, LEVEL_WARN, LEVEL_INFO, LEVEL_DEBUG, LEVEL_TRACE };

#ifndef __cplusplus
typedef enum wholememory_error_code_t wholememory_error_code_t;
typedef enum wholememory_memory_type_t wholememory_memory_type_t;
typedef enum wholememory_memory_location_t wholememory_memory_location_t;
typedef enum wholememory_distributed_backend_t wholememory_distributed_backend_t;
typedef enum LogLevel LogLevel;
#endif

#define WHOLEMEMORY_SPILT_NO_COLOR -1 /* (sic) reference wholememory.h:95 */

/* ---- library lifetime: reference wholememory.h:102-108 ---- */
enum wholememory_error_code_t wholememory_init(unsigned int flags,
                                               enum LogLevel log_level WM_DEFAULT(LEVEL_INFO));
enum wholememory_error_code_t wholememory_finalize();

/* ---- communicator: reference wholememory.h:115-252 ---- */
typedef struct wholememory_comm_* wholememory_comm_t;

struct clique_info_t { /* MNNVL clique description; always "not in a clique" on MI355X */
  int is_in_clique;
  int clique_first_rank;
  int clique_rank;
  int clique_rank_num;
  int clique_id;
  int clique_num;
};

#define WHOLEMEMORY_UNIQUE_ID_BYTES (128)
struct wholememory_unique_id_t { /* carries an RCCL ncclUniqueId */
  char internal[WHOLEMEMORY_UNIQUE_ID_BYTES];
};
#ifndef __cplusplus
typedef struct clique_info_t clique_info_t;
typedef struct wholememory_unique_id_t wholememory_unique_id_t;
#endif

enum wholememory_error_code_t wholememory_create_unique_id(
  struct wholememory_unique_id_t* unique_id);
enum wholememory_error_code_t wholememory_create_communicator(
  wholememory_comm_t* comm, struct wholememory_unique_id_t unique_id, int rank, int size);
enum wholememory_error_code_t wholememory_split_communicator(wholememory_comm_t* new_comm,
                                                             wholememory_comm_t comm,
                                                             int color,
                                                             int key);
enum wholememory_error_code_t wholememory_destroy_communicator(wholememory_comm_t comm);
enum wholememory_error_code_t wholememory_communicator_support_type_location(
  wholememory_comm_t comm,
  enum wholememory_memory_type_t memory_type,
  enum wholememory_memory_location_t memory_location);
enum wholememory_error_code_t wholememory_communicator_get_rank(int* rank, wholememory_comm_t comm);
enum wholememory_error_code_t wholememory_communicator_get_size(int* size, wholememory_comm_t comm);
enum wholememory_error_code_t wholememory_communicator_get_local_size(int* local_size,
                                                                      wholememory_comm_t comm);
enum wholememory_error_code_t wholememory_communicator_get_clique_info(
  struct clique_info_t* clique_info, wholememory_comm_t comm);
bool wholememory_communicator_is_bind_to_nvshmem(wholememory_comm_t comm);
enum wholememory_error_code_t wholememory_communicator_set_distributed_backend(
  wholememory_comm_t comm, enum wholememory_distributed_backend_t distributed_backend);
enum wholememory_distributed_backend_t wholememory_communicator_get_distributed_backend(
  wholememory_comm_t comm);
enum wholememory_error_code_t wholememory_communicator_barrier(wholememory_comm_t comm);
bool wholememory_is_intranode_communicator(wholememory_comm_t comm);
bool wholememory_is_intra_mnnvl_communicator(wholememory_comm_t comm);
bool wholememory_is_build_with_nvshmem();

/* ---- WholeMemory handle: reference wholememory.h:259-430 ---- */
typedef struct wholememory_handle_* wholememory_handle_t;

/* Collective over comm. total_size bytes split at data_granularity boundaries; rank_entry_partition
 * (entries of data_granularity bytes per rank, world_size values) overrides the equal plan. */
enum wholememory_error_code_t wholememory_malloc(wholememory_handle_t* wholememory_handle_ptr,
                                                 size_t total_size,
                                                 wholememory_comm_t comm,
                                                 enum wholememory_memory_type_t memory_type,
                                                 enum wholememory_memory_location_t memory_location,
                                                 size_t data_granularity,
                                                 size_t* rank_entry_partition WM_DEFAULT(nullptr));
enum wholememory_error_code_t wholememory_free(wholememory_handle_t wholememory_handle);

enum wholememory_error_code_t wholememory_get_communicator(wholememory_comm_t* comm,
                                                           wholememory_handle_t handle);
enum wholememory_error_code_t wholememory_get_local_communicator(wholememory_comm_t* comm,
                                                                 wholememory_handle_t handle);
enum wholememory_error_code_t wholememory_get_cross_communicator(wholememory_comm_t* comm,
                                                                 wholememory_handle_t handle);
enum wholememory_memory_type_t wholememory_get_memory_type(wholememory_handle_t handle);
enum wholememory_memory_location_t wholememory_get_memory_location(wholememory_handle_t handle);
enum wholememory_distributed_backend_t wholememory_get_distributed_backend(
  wholememory_handle_t handle);
size_t wholememory_get_total_size(wholememory_handle_t handle);
size_t wholememory_get_data_granularity(wholememory_handle_t handle);

enum wholememory_error_code_t wholememory_get_local_memory(void** local_ptr,
                                                           size_t* local_size,
                                                           size_t* local_offset,
                                                           wholememory_handle_t handle);
enum wholememory_error_code_t wholememory_get_local_size(size_t* local_size,
                                                         wholememory_handle_t handle);
enum wholememory_error_code_t wholememory_get_local_offset(size_t* local_offset,
                                                           wholememory_handle_t handle);
enum wholememory_error_code_t wholememory_get_rank_memory(void** rank_memory_ptr,
                                                          size_t* rank_memory_size,
                                                          size_t* rank_memory_offset,
                                                          int rank,
                                                          wholememory_handle_t handle);
/* entries per rank of the default plan = ceil(total / world) — reference wholememory.h:391-393 */
enum wholememory_error_code_t wholememory_equal_entry_partition_plan(size_t* entry_per_rank,
                                                                     size_t total_entry_count,
                                                                     int world_size);
enum wholememory_error_code_t wholememory_get_global_pointer(void** global_ptr,
                                                             wholememory_handle_t handle);
enum wholememory_error_code_t wholememory_get_global_reference(struct wholememory_gref_t* gref,
                                                               wholememory_handle_t handle);
/* bytes per rank [world_size] / byte offsets [world_size + 1] */
enum wholememory_error_code_t wholememory_get_rank_partition_sizes(size_t* rank_mem_sizes,
                                                                   wholememory_handle_t handle);
enum wholememory_error_code_t wholememory_get_rank_partition_offsets(size_t* rank_mem_offsets,
                                                                     wholememory_handle_t handle);

/* reference wholememory.h:436 — device count probed in a forked child (no HIP state in parent) */
int fork_get_device_count();

/* ---- raw-binary table I/O ("%s_part_%d_of_%d" shards): reference wholememory.h:449-471 ---- */
enum wholememory_error_code_t wholememory_load_from_file(wholememory_handle_t handle,
                                                         size_t memory_offset,
                                                         size_t memory_entry_size,
                                                         size_t file_entry_size,
                                                         const char** file_names,
                                                         int file_count,
                                                         int round_robin_size);
enum wholememory_error_code_t wholememory_store_to_file(wholememory_handle_t handle,
                                                        size_t memory_offset,
                                                        size_t memory_entry_stride,
                                                        size_t file_entry_size,
                                                        const char* local_file_name);

#ifdef __cplusplus
}
#endif
#endif
