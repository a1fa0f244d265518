/*
 * wholegraph_amd — WholeMemory embedding table: padded row-sharded table + sparse optimizer.
 * Replaces reference cpp/include/wholememory/embedding.h:30-244 (what pylibwholegraph's
 * EmbeddingGatherForward / EmbeddingGatherGradientApply bind, wholememory_binding.pyx:920-948).
 */
#ifndef WHOLEMEMORY_EMBEDDING_H_
#define WHOLEMEMORY_EMBEDDING_H_

#include <wholememory/env_func_ptrs.h>
#include <wholememory/wholememory_tensor.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wholememory_embedding_cache_policy_* wholememory_embedding_cache_policy_t;
typedef struct wholememory_embedding_optimizer_* wholememory_embedding_optimizer_t;
typedef struct wholememory_embedding_* wholememory_embedding_t;

enum wholememory_access_type_t { /* reference embedding.h:50-54 */
  WHOLEMEMORY_AT_NONE = 0,
  WHOLEMEMORY_AT_READONLY,
  WHOLEMEMORY_AT_READWRITE,
};
enum wholememory_optimizer_type_t { /* reference embedding.h:60-66 */
  WHOLEMEMORY_OPT_NONE = 0,
  WHOLEMEMORY_OPT_SGD,
  WHOLEMEMORY_OPT_LAZY_ADAM,
  WHOLEMEMORY_OPT_RMSPROP,
  WHOLEMEMORY_OPT_ADAGRAD,
};
#ifndef __cplusplus
typedef enum wholememory_access_type_t wholememory_access_type_t;
typedef enum wholememory_optimizer_type_t wholememory_optimizer_type_t;
#endif

/* ---- sparse optimizers: reference embedding.h:74-98 ----
 * parameters (float*, by name): "weight_decay" for all; "epsilon","beta1","beta2","adam_w" (lazy
 * adam); "epsilon","alpha" (rmsprop); "epsilon" (adagrad). */
enum wholememory_error_code_t wholememory_create_embedding_optimizer(wholememory_embedding_optimizer_t* out,
                                                                     enum wholememory_optimizer_type_t type);
enum wholememory_error_code_t wholememory_optimizer_set_parameter(wholememory_embedding_optimizer_t opt,
                                                                  const char* name, void* float_value);
void wholememory_destroy_embedding_optimizer(wholememory_embedding_optimizer_t opt);

/* ---- cache policy: reference embedding.h:110-124. cache_comm == the embedding's communicator: every rank keeps a
 * device cache of ITS OWN shard (read-only or read-write); another communicator: every rank keeps a read-only cache
 * of the whole table for its own lookups. ratio in [1/512, 1]: cache lines per covered row. */
enum wholememory_error_code_t wholememory_create_embedding_cache_policy(
  wholememory_embedding_cache_policy_t* out, wholememory_comm_t cache_comm, enum wholememory_memory_type_t cache_memory_type,
  enum wholememory_memory_location_t cache_location, enum wholememory_access_type_t access, float ratio);
enum wholememory_error_code_t wholememory_destroy_embedding_cache_policy(wholememory_embedding_cache_policy_t policy);

/* ---- table lifetime: reference embedding.h:138-165. `desc` is 2-D (dtype + sizes used); rows are padded to a
 * 16-byte multiple. Collective over comm. entry_partition: rows per rank (NULL = equal plan); sms: grid cap of the
 * gather kernels (-1 = default); round_robin_size: rows dealt to the ranks in turns of this many (0 = range split). */
enum wholememory_error_code_t wholememory_create_embedding(
  wholememory_embedding_t* out, struct wholememory_tensor_description_t* desc, wholememory_comm_t comm,
  enum wholememory_memory_type_t memory_type, enum wholememory_memory_location_t location,
  wholememory_embedding_cache_policy_t cache_policy, size_t* entry_partition WM_DEFAULT(nullptr), int sms WM_DEFAULT(-1),
  int round_robin_size WM_DEFAULT(0));
enum wholememory_error_code_t wholememory_destroy_embedding(wholememory_embedding_t emb);
/* borrowed [N, dim] view of the padded table */
wholememory_tensor_t wholememory_embedding_get_embedding_tensor(wholememory_embedding_t emb);
/* once per embedding; allocates + initialises the optimizer state tensors. fp32 tables (HALF / BF16: SGD only) */
enum wholememory_error_code_t wholememory_embedding_set_optimizer(wholememory_embedding_t emb,
                                                                  wholememory_embedding_optimizer_t opt);

/* ---- the hot path: reference embedding.h:184-209. stream_int is a hipStream_t as int64. ---- */
enum wholememory_error_code_t wholememory_embedding_gather(wholememory_embedding_t emb, wholememory_tensor_t indices,
                                                           wholememory_tensor_t output, bool adjust_cache,
                                                           struct wholememory_env_func_t* env, int64_t stream_int);
/* Collective: ids + gradient rows go to the owning rank, duplicates are summed in a defined order (fp32), then the
 * optimizer updates the local shard. */
enum wholememory_error_code_t wholememory_embedding_gather_gradient_apply(
  wholememory_embedding_t emb, wholememory_tensor_t indices, wholememory_tensor_t grads, bool adjust_cache, float lr,
  struct wholememory_env_func_t* env, int64_t stream_int);

/* ---- optimizer state access: reference embedding.h:217-228 (names array is NULL-terminated) */
const char* const* wholememory_embedding_get_optimizer_state_names(wholememory_embedding_t emb);
wholememory_tensor_t wholememory_embedding_get_optimizer_state(wholememory_embedding_t emb, const char* name);

/* ---- cache maintenance: reference embedding.h:236-244 (no-ops without a cache) ---- */
enum wholememory_error_code_t wholememory_embedding_writeback_cache(wholememory_embedding_t emb, int64_t stream_int);
enum wholememory_error_code_t wholememory_embedding_drop_all_cache(wholememory_embedding_t emb, int64_t stream_int);

#ifdef __cplusplus
}
#endif
#endif
