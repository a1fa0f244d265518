/*
 * wholegraph_amd — caller-supplied allocators ("env functions").
 * Every scratch buffer an op needs is requested through these callbacks so the host framework's
 * caching allocator (torch-ROCm) owns it. Replaces reference
 * cpp/include/wholememory/env_func_ptrs.h:25-76; the one CUDA type in that header
 * (cudaDeviceProp* get_device_prop, :76) becomes hipDeviceProp_t*.
 */
#ifndef WHOLEMEMORY_ENV_FUNC_PTRS_H_
#define WHOLEMEMORY_ENV_FUNC_PTRS_H_

#include <wholememory/tensor_description.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference env_func_ptrs.h:33-38 */
enum wholememory_memory_allocation_type_t {
  WHOLEMEMORY_MA_NONE = 0,
  WHOLEMEMORY_MA_DEVICE,
  WHOLEMEMORY_MA_HOST,
  WHOLEMEMORY_MA_PINNED,
};
#ifndef __cplusplus
typedef enum wholememory_memory_allocation_type_t wholememory_memory_allocation_type_t;
#endif

/* reference env_func_ptrs.h:43-56: a memory_context is one allocation slot */
typedef void (*wholememory_create_memory_context_func_t)(void** slot_out, void* global);
typedef void (*wholememory_destroy_memory_context_func_t)(void* slot, void* global);
/* allocates what `shape` describes in the given kind of memory, remembers it in `slot`, returns the data pointer */
typedef void* (*wholememory_malloc_func_t)(struct wholememory_tensor_description_t* shape,
                                           enum wholememory_memory_allocation_type_t kind, void* slot, void* global);
typedef void (*wholememory_free_func_t)(void* slot, void* global);

/* Stream ordering contract of temporary_fns (the reference leaves it implicit): the ops call malloc_fn / free_fn from
 * the host while kernels that use the memory may still be queued on the op's `stream`. An allocator must therefore
 * either be stream-ordered on that stream (torch's caching allocator on the current stream is, and the ops are always
 * called with the current stream by the torch layer) or free synchronously (hipFree / the built-in default env). Ops
 * that also work on the communicator's side stream fence it with an event on `stream` before they return, and the
 * raw-stage entry points of wholegraph_amd_ext.h synchronise `stream` before their scratch is released. */
struct wholememory_temp_memory_func_t { /* reference env_func_ptrs.h:58-64 */
  wholememory_create_memory_context_func_t create_memory_context_fn;
  wholememory_destroy_memory_context_func_t destroy_memory_context_fn;
  wholememory_malloc_func_t malloc_fn;
  wholememory_free_func_t free_fn;
  void* global_context;
};
struct wholememory_output_memory_func_t { /* reference env_func_ptrs.h:65-69 */
  wholememory_malloc_func_t malloc_fn;
  wholememory_free_func_t free_fn;
  void* global_context;
};
struct wholememory_env_func_t { /* reference env_func_ptrs.h:71-74 */
  struct wholememory_temp_memory_func_t temporary_fns;
  struct wholememory_output_memory_func_t output_fns;
};
#ifndef __cplusplus
typedef struct wholememory_temp_memory_func_t wholememory_temp_memory_func_t;
typedef struct wholememory_output_memory_func_t wholememory_output_memory_func_t;
typedef struct wholememory_env_func_t wholememory_env_func_t;
#endif

/* Opaque here so this header needs no HIP include: points at a hipDeviceProp_t. dev_id -1 = current.
 * reference env_func_ptrs.h:76 */
void* get_device_prop(int dev_id);

/* Built-in env tables for C/C++ callers, tests and the bench (reference
 * cpp/src/wholememory/env_func_ptrs.hpp: get_default_env_func / get_cached_env_func /
 * drop_cached_env_func_cache): plain hipMalloc-backed, and a size-class pooled variant. */
struct wholememory_env_func_t* wholememory_get_default_env_func();
struct wholememory_env_func_t* wholememory_get_cached_env_func();
void wholememory_drop_cached_env_func_cache();

#ifdef __cplusplus
}
#endif
#endif
