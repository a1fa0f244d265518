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
enum wholememory_error_code_t wholememory_create_embedding_optimizer(
  wholememory_embedding_optimizer_t* optimizer, enum wholememory_optimizer_type_t optimizer_type);
enum wholememory_error_code_t wholememory_optimizer_set_parameter(
  wholememory_embedding_optimizer_t optimizer, const char* parameter_name, void* value);
void wholememory_destroy_embedding_optimizer(wholememory_embedding_optimizer_t optimizer);

/* ---- cache policy: reference embedding.h:110-124. The device LFU cache is out of this build's
 * scope: policies can be created/destroyed so call sites link, but creating an
 * embedding WITH a policy returns WHOLEMEMORY_NOT_IMPLEMENTED. */
enum wholememory_error_code_t wholememory_create_embedding_cache_policy(
  wholememory_embedding_cache_policy_t* cache_policy,
  wholememory_comm_t cache_level_comm,
  enum wholememory_memory_type_t memory_type,
  enum wholememory_memory_location_t memory_location,
  enum wholememory_access_type_t access_type,
  float cache_ratio);
enum wholememory_error_code_t wholememory_destroy_embedding_cache_policy(
  wholememory_embedding_cache_policy_t cache_policy);

/* ---- table lifetime: reference embedding.h:138-165. Description is 2-D (dtype + sizes used);
 * rows are padded to a 16-byte multiple. Collective over comm. */
enum wholememory_error_code_t wholememory_create_embedding(
  wholememory_embedding_t* wholememory_embedding,
  struct wholememory_tensor_description_t* embedding_tensor_description,
  wholememory_comm_t comm,
  enum wholememory_memory_type_t memory_type,
  enum wholememory_memory_location_t memory_location,
  wholememory_embedding_cache_policy_t cache_policy,
  size_t* embedding_entry_partition WM_DEFAULT(nullptr),
  int user_defined_sms WM_DEFAULT(-1),
  int round_robin_size WM_DEFAULT(0));
enum wholememory_error_code_t wholememory_destroy_embedding(
  wholememory_embedding_t wholememory_embedding);
/* borrowed [N, dim] view of the padded table */
wholememory_tensor_t wholememory_embedding_get_embedding_tensor(
  wholememory_embedding_t wholememory_embedding);
/* once per embedding, fp32 tables only; allocates + initialises optimizer state tensors */
enum wholememory_error_code_t wholememory_embedding_set_optimizer(
  wholememory_embedding_t wholememory_embedding, wholememory_embedding_optimizer_t optimizer);

/* ---- the hot path: reference embedding.h:184-209. stream_int is a hipStream_t as int64. ---- */
enum wholememory_error_code_t wholememory_embedding_gather(
  wholememory_embedding_t wholememory_embedding,
  wholememory_tensor_t indices,
  wholememory_tensor_t output,
  bool adjust_cache,
  struct wholememory_env_func_t* p_env_fns,
  int64_t stream_int);
/* Collective: ids+grads go to the owning rank, duplicates are summed in a defined order (fp32),
 * then the optimizer updates the local shard. */
enum wholememory_error_code_t wholememory_embedding_gather_gradient_apply(
  wholememory_embedding_t wholememory_embedding,
  wholememory_tensor_t indices,
  wholememory_tensor_t grads,
  bool adjust_cache,
  float lr,
  struct wholememory_env_func_t* p_env_fns,
  int64_t stream_int);

/* ---- optimizer state access: reference embedding.h:217-228 (names array is NULL-terminated) */
const char* const* wholememory_embedding_get_optimizer_state_names(
  wholememory_embedding_t wholememory_embedding);
wholememory_tensor_t wholememory_embedding_get_optimizer_state(
  wholememory_embedding_t wholememory_embedding, const char* name);

/* ---- cache maintenance: reference embedding.h:236-244 (no-ops without a cache) ---- */
enum wholememory_error_code_t wholememory_embedding_writeback_cache(
  wholememory_embedding_t wholememory_embedding, int64_t stream_int);
enum wholememory_error_code_t wholememory_embedding_drop_all_cache(
  wholememory_embedding_t wholememory_embedding, int64_t stream_int);

#ifdef __cplusplus
}
#endif
#endif
