// wholegraph_amd — host side of the device row cache of an embedding (kernels/cache.hip has the design notes).
// Replaces reference cpp/src/wholememory/embedding_cache.{hpp,cpp} (cache objects, sizing) and the cache halves of
// embedding.cpp:564-892 (device_cached_host_embedding / local_cached_global_readonly_embedding).
#pragma once

#include <wholememory/embedding.h>

#include "backend.hpp"
#include "ops_internal.hpp"

struct wholememory_embedding_cache_policy_ {
  wholememory_comm_t cache_comm;
  wholememory_memory_type_t cache_memory_type;
  wholememory_memory_location_t cache_memory_location;
  wholememory_access_type_t access_type;
  float cache_ratio;
};

namespace wm {

struct row_cache {
  wm_cache_args args{};
  bool same_comm = false;  // cache communicator == embedding communicator: every rank caches ITS OWN shard and serves it
                           // (reference device_cached_host_embedding); otherwise each rank caches any row of the global
                           // table for its own lookups, read-only (reference local_cached_global_readonly_embedding)
  bool writable  = false;
  bool raw_addressable = true;  // false: the raw table is DISTRIBUTED and this is a local cache of it — rows are fetched
                                // through the exchange (collectively) and only then installed
  wholememory_tensor_t raw = nullptr;  // the embedding's padded table (borrowed)
  wholememory_dtype_t dtype = WHOLEMEMORY_DT_UNKNOWN;
  int64_t row_elems         = 0;  // elements per cache line (= padded row)
  unsigned long long* counters_dev = nullptr;  // [0] hits, [1..2] scratch of cache_info
  int64_t lookups                  = 0;
  ~row_cache();
};

// sizes and allocates the cache of `raw` (the embedding's padded [N, stride] tensor) per `policy`
wholememory_error_code_t create_row_cache(row_cache** out, const wholememory_embedding_cache_policy_* policy,
                                          wholememory_tensor_t raw, wholememory_comm_t embedding_comm);

// adds a batch of ids (device array) to the counters and lets frequently used missing rows replace the least used
// residents. key_upper_bound as in dedup_ids (0 = ids may be negative: full-width keys)
wholememory_error_code_t row_cache_update(row_cache* c, const void* ids, wholememory_dtype_t index_dtype, int64_t n,
                                          int64_t key_upper_bound, wholememory_env_func_t* env, void* stream);

// out rows of `ids` from the cache where resident, from the raw table (through `raw_gref`, GLOBAL row ids) otherwise;
// a: rows args prepared for the RAW table (indices, row_map, plain side, dtypes) — see ops.cpp:fill_rows_args
wholememory_error_code_t row_cache_gather(row_cache* c, const wm_rows_args& a, wholememory_env_func_t* env, void* stream);

// plan mode of row_cache_update (raw table not addressable): counts the batch, decides the replacements and returns
// them as device lists (global rows / cache slots, *n_fill of them) WITHOUT moving data; the caller fetches the rows and
// calls row_cache_install. The lists live in `rows_mem` / `slots_mem`.
wholememory_error_code_t row_cache_plan(row_cache* c, const void* ids, wholememory_dtype_t index_dtype, int64_t n,
                                        int64_t key_upper_bound, wholememory_env_func_t* env, void* stream, temp_mem* rows_mem, temp_mem* slots_mem,
                                        int64_t* n_fill);
// cache_line[slots[k]] = rows_data[k] for k < n_fill (rows_data: dense [n_fill, row_elems] of the raw dtype)
wholememory_error_code_t row_cache_install(row_cache* c, const void* rows_data, const int64_t* slots, int64_t n_fill,
                                           void* stream);
// the two index lists of a lookup (see wm_device_backend::cache_split); counts the lookups
wholememory_error_code_t row_cache_split(row_cache* c, const void* ids, wholememory_dtype_t index_dtype, int64_t n,
                                         int64_t* cache_idx, void* raw_idx, void* stream);

// Lets the packed per-element optimizer states of the cached rows share the cache: `state_local` is this rank's shard
// of the state table ([local rows, state_row_elems] fp32, same row partition as the embedding). The cache is emptied
// first, so every line that becomes resident afterwards carries its state line with it.
wholememory_error_code_t row_cache_attach_states(row_cache* c, wholememory_tensor_t state_local);

wholememory_error_code_t row_cache_writeback(row_cache* c, bool drop, void* stream);
wholememory_error_code_t row_cache_info(row_cache* c, int64_t* slots, int64_t* occupied, int64_t* dirty, int64_t* hits,
                                        int64_t* lookups, void* stream);

}  // namespace wm
