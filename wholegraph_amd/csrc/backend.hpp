// wholegraph_amd — the device seam.
//
// Host orchestration (ops.cpp, embedding.cpp, memory_handle.cpp) reaches the GPU only through this
// table: raw memory management plus one launcher per hand-written gfx950 kernel. The product
// installs exactly one implementation, the HIP one (kernels/*.hip → hip_backend()); a missing or
// failing HIP runtime is a hard error, there is no CPU fallback in this library.
//
// The table is also the seam that lets tests/ drive the multi-rank orchestration at world_size 2 on
// a CPU-only box: oracle/test_backend.c (test infrastructure) can be installed through
// wm_testing_install_backend(), which refuses to act unless WHOLEGRAPH_AMD_TESTING=1 is set.
#pragma once

#include <cstddef>
#include <cstdint>

#include <wholememory/global_reference.h>
#include <wholememory/tensor_description.h>

extern "C" {

// gather: out[i,:] = cast(table[idx[i],:]), idx < 0 skipped. scatter: the mirror image.
// `gref` addresses the table (continuous or chunked); dims/strides/offsets in ELEMENTS.
struct wm_rows_args {
  wholememory_gref_t gref;            // table
  wholememory_dtype_t table_dtype;
  int64_t dim;                        // columns moved per row
  int64_t table_stride;               // elements
  int64_t table_storage_offset;       // elements
  const void* indices;                // [n] int32/int64, device
  wholememory_dtype_t index_dtype;
  int64_t n;
  const void* row_map;                // optional [n] int64: plain-side row for entry i; nullptr = i
  void* plain;                        // output (gather) / input (scatter), device
  wholememory_dtype_t plain_dtype;
  int64_t plain_stride;               // elements
  int64_t plain_storage_offset;       // elements
  int max_blocks;                     // "gather_sms"/"scatter_sms": -1 = default
};

// per-rank bucketing of ids (see kernels/bucket.hip)
struct wm_bucket_args {
  const void* indices;  // [n] device
  wholememory_dtype_t index_dtype;
  int64_t n;
  const uint64_t* entry_offsets;  // [world+1] device, row offsets
  int world_size;
  int64_t* counts;       // [world] device out: ids per owner (negatives not counted)
  void* bucketed_ids;    // [n] device out (index dtype): ids grouped by owner, stable; may be nullptr
  int64_t* raw_indices;  // [n] device out: original position of each bucketed id; may be nullptr
  void* workspace;       // device scratch of bucket_workspace_bytes(n, world)
  // 0: one bucket per owner (entry_offsets has world+1 entries). > 0: entry_offsets describes owner_count >= world_size
  // owners (owner_count+1 entries) and an id of owner o goes to bucket o % world_size — first hop of the HIERARCHY
  // gather, where the buckets are the ranks of this node and the owners the ranks of every node
  int owner_count;
  // 1: `workspace` still holds the scanned block offsets an earlier counts-only call (bucketed_ids == nullptr) over the very
  // same ids / offsets left there, and `counts` is not needed again: only the grouping pass runs. A backend may ignore it
  // and recompute everything (the results are the same).
  int reuse_scan;
};

struct wm_optimizer_args {
  int type;                    // wholememory_optimizer_type_t
  const void* ids;             // [count] device: unique global row ids
  wholememory_dtype_t index_dtype;
  const int32_t* run_starts;   // [count+1] device: segment starts into order[]
  const int32_t* order;        // [n_recv] device: receive-buffer positions sorted by (id, position)
  // dtype of the table AND of the gradient rows: FLOAT (every optimizer; 0 / UNKNOWN is read as FLOAT) or HALF / BF16
  // (SGD only — an extension, the reference trains fp32 tables only: duplicates are summed in fp32 in receive order,
  // the update is computed in fp32 from fp32(e) and rounded once to the table dtype)
  wholememory_dtype_t value_dtype;
  const void* grads;           // [n_recv, grad_stride] device: received gradient rows
  int64_t grad_stride;
  // order[] entries >= 0 address rows of `grads`; an entry < 0 addresses row -(entry + 1) of `self_grads` — gradient rows
  // of ids this rank owns itself, read where the caller left them instead of being copied into the receive buffer
  const void* self_grads;
  int64_t self_grad_stride;
  int64_t count;               // number of unique ids (= grid size)
  // optional device row cache of the local shard (kernels/cache.hip): row `local` lives in cache line cache_slot_of[local]
  // when that is >= 0 (the update then goes there and the line is marked modified), else in local_table
  const int32_t* cache_slot_of;
  void* cache_data;
  uint8_t* cache_dirty;
  int64_t cache_row_elems;
  float* cache_state_data;       // companion lines of per_element_state (same slots) or nullptr
  int64_t cache_state_row_elems;
  void* local_table;           // this rank's first row
  int64_t table_stride;        // elements
  int64_t local_entry_offset;  // global id of local row 0
  int64_t dim;
  float* per_element_state;    // adam: [rows, 2*stride] (m|v); adagrad/rmsprop: [rows, stride]; else nullptr
  int64_t per_element_stride;
  float* per_row_state;        // adam: [rows, 2] (beta1^t, beta2^t); else nullptr
  float weight_decay, epsilon, beta1, beta2, alpha, lr;
  int adam_w;
  void* long_run_ws;           // device scratch of long_run_workspace_bytes(n_recv, dim) or nullptr (then every run is
                               // folded by one wave)
  size_t long_run_ws_bytes;    // its size as long_run_workspace_bytes() returned it (round 6: the answer may include room for
                               // dense copies of the longest runs, or not — it follows what earlier steps saw; 0 = no such room)
  // Order of the fp32 sum of a run's duplicate gradient rows. 0 = the reference's (receive order, one chain per element:
  // exchange_embeddings_nccl_func.cu:76-103) — bit-identical results, the default for fp32 tables. 1 = "tree": runs of more
  // than a few dozen rows are cut into segments of rows that are summed side by side and combined afterwards — a fixed
  // order too (results are deterministic), but not the reference's: equal within rounding (and exact whenever the partial
  // sums are exactly representable). -1 = the backend's default for the value dtype (ordered for fp32, tree for the 16-bit
  // extension, WM_GRAD_FOLD=ordered|tree overrides).
  int fold_mode;
};

// neighbour sampling on a CSR graph whose arrays are mapped (flat or chunked) in this process
struct wm_sample_args {
  wholememory_gref_t row_gref, col_gref;  // csr_row_ptr (int64), csr_col_ptr (col_dtype)
  int64_t row_storage_offset, col_storage_offset;  // elements
  wholememory_dtype_t col_dtype;
  const int64_t* row_pairs;  // optional [2 * n_center] device: (row_ptr[c], row_ptr[c + 1]) fetched beforehand (DISTRIBUTED
                             // CSR); then out_ids must be nullptr and only out_edge_gid / out_center_lid are written
  const void* centers;  // [n_center] device
  wholememory_dtype_t center_dtype;
  int n_center;
  int max_sample_count;  // <= 0: every neighbour
  uint64_t random_seed;
  const int* sample_offsets;  // [n_center + 1] device, exclusive prefix of the per-center counts
  void* out_ids;              // [total] col_dtype
  int* out_center_lid;        // [total] or nullptr
  int64_t* out_edge_gid;      // [total] or nullptr
  // weighted sampling only: per-edge weights (float / double), same indexing as csr_col_ptr
  wholememory_gref_t weight_gref;
  int64_t weight_storage_offset;
  wholememory_dtype_t weight_dtype;
  // optional (device): the number of centres in use when `centers` / `sample_offsets` are sized for an upper bound n_center
  // (the frontier of the previous hop of a bounded multi-hop call); nullptr = n_center
  const int* n_center_dev;
  // optional side job of the sampling kernel: fill_ff_bytes bytes at fill_ff_ptr (both multiples of 16) are set to 0xFF —
  // the empty hash table of the append_unique that follows in a fused hop (append_unique_table_region), which then skips
  // its own fill command (wm_au_bounds::table_is_clear): one launch fewer per hop
  void* fill_ff_ptr;
  size_t fill_ff_bytes;
};

// device-side sizes of an append_unique over upper-bound-sized arrays (the hops of wholememory_ext_multilayer_sample: nothing
// leaves the device between hops). nullptr instead of the struct = every size is the host's.
struct wm_au_bounds {
  const int* n_target_dev;    // targets in use (<= n_target)
  const int* n_neighbor_dev;  // neighbours in use (<= n_neighbor)
  int* n_unique_dev;          // out: targets in use + new ids = what the output holds (the next hop's centre count)
  // optional (pinned host memory, 2 ints): when set, {neighbours in use, new unique ids} and n_unique_dev are written by
  // phase 2's emitting kernel instead of a kernel of their own at the end of phase 1 (the caller runs both phases back to
  // back without looking at the counts in between, and passes no publish_host to phase 1): one tiny launch fewer per hop
  int* publish_host_late;
  // 1: the hash table region of the workspace (append_unique_table_region) is already all-ones
  int table_is_clear;
  // 1: phase 2 also sets the entries of the output array BEHIND the unique ids (up to its room of n_target + n_neighbor) to
  // -1 ("skip me" for a gather): the array can then be handed to the next op at its full size before the host knows the count
  int pad_unique_tail;
};

// device row cache of an embedding (kernels/cache.hip): direct map row -> slot, 64-slot LFU sets
struct wm_cache_args {
  int32_t* slot_of;     // [cover_rows] slot of a covered row or -1
  int32_t* count;       // [cover_rows] access counter
  int64_t* row_of;      // [64 * n_sets] covered-row index resident in a slot or -1
  uint8_t* dirty;       // [64 * n_sets]
  char* data;           // [64 * n_sets, row_bytes] cache lines
  int64_t cover_start;  // first GLOBAL row this cache may hold
  int64_t cover_rows;
  int64_t n_sets;
  int64_t set_cover;    // rows per set: set s covers [s * set_cover, (s + 1) * set_cover) of the covered range
  int64_t row_bytes;    // bytes per cache line = padded row (multiple of 16)
  wholememory_gref_t raw_gref;  // the raw table, addressed by GLOBAL row
  int64_t raw_row_stride_bytes;
  int64_t raw_row_offset_bytes;
  // optional companion table that shares the slots (the packed per-element optimizer states of the same rows): its
  // lines move in and out together with the embedding's
  char* data2;                   // [64 * n_sets, row_bytes2] or nullptr
  int64_t row_bytes2;
  wholememory_gref_t raw2_gref;  // addressed by GLOBAL row like raw_gref
  int64_t raw2_row_stride_bytes;
};

struct wm_device_backend {
  const char* name;
  // memory / stream
  int (*device_count)();
  int (*malloc_device)(void** p, size_t bytes);
  int (*free_device)(void* p);
  int (*malloc_pinned)(void** p, size_t bytes);
  int (*free_pinned)(void* p);
  int (*memcpy_async)(void* dst, const void* src, size_t bytes, void* stream);  // any direction (UVA)
  int (*memset_async)(void* dst, int value, size_t bytes, void* stream);
  int (*stream_sync)(void* stream);
  // side stream + events (overlap of the all-to-all-v with local kernels)
  int (*stream_create)(void** stream);
  int (*stream_destroy)(void* stream);
  int (*event_create)(void** event);
  int (*event_destroy)(void* event);
  int (*event_record)(void* event, void* stream);
  int (*stream_wait_event)(void* stream, void* event);
  // cross-process mapping of device allocations (hipIpc*): handle is 64 opaque bytes
  int (*ipc_get_handle)(void* handle64, void* dev_ptr);
  int (*ipc_open_handle)(void** dev_ptr, const void* handle64);
  int (*ipc_close_handle)(void* dev_ptr);
  // make a host range (shared-memory segment) device-visible; *dev_ptr = address kernels may use
  int (*host_register)(void* host_ptr, size_t bytes, void** dev_ptr);
  int (*host_unregister)(void* host_ptr);
  // kernels (all asynchronous on `stream`)
  int (*gather_rows)(const wm_rows_args* a, void* stream);
  int (*scatter_rows)(const wm_rows_args* a, void* stream);
  size_t (*bucket_workspace_bytes)(int64_t n, int world_size);
  int (*bucket_ids)(const wm_bucket_args* a, void* stream);
  // stable sort of the ids by their two's-complement bits as UNSIGNED keys (valid ids ascending, negative ids after all
  // of them), emit unique ids, run starts and the sorted order. n_unique_out is a device int64. workspace from
  // dedup_workspace_bytes(n).
  size_t (*dedup_workspace_bytes)(int64_t n, wholememory_dtype_t index_dtype);
  // key_upper_bound > 0 (key_lower_bound defaults to 0): the ids of interest are those in [lower, upper) — the owner's own
  // row range; a backend may sort id - lower and so need fewer key bits (a 125 M-row shard of a 1 B-row table: 27 instead
  // of 30 bits, 3 radix passes instead of 4). The outputs are the ids themselves either way. Ids OUTSIDE the range
  // (negative "skip me" ids, ids past the table) are left out of the runs when upper - lower < 2^32 - 1: they are not
  // counted in n_unique_out, and their positions fill the tail of `order` behind the last run (run_starts[n_unique] = where
  // that tail starts). For a wider range every id must lie inside it.
  int (*dedup_ids)(const void* ids, wholememory_dtype_t index_dtype, int64_t n, int64_t key_upper_bound, int64_t key_lower_bound,
                   void* unique_ids, int32_t* run_starts, int32_t* order, int64_t* n_unique_out, void* workspace,
                   void* stream);
  // fused duplicate-sum + optimizer update. a->count bounds the launch; when n_unique_dev != nullptr the true
  // number of unique ids is read from that device scalar (no host sync to learn it).
  int (*optimizer_step)(const wm_optimizer_args* a, const int64_t* n_unique_dev, void* stream);
  size_t (*long_run_workspace_bytes)(int64_t n_recv, int64_t dim);
  // inverse of a dedup: inverse[order[j]] = index of the run that sorted position j belongs to, or -1 when that run's
  // id is negative or, with id_limit > 0, not below id_limit (n_unique_dev: device scalar written by dedup_ids)
  int (*run_inverse)(const int32_t* run_starts, const int32_t* order, const void* unique_ids, wholememory_dtype_t index_dtype,
                     const int64_t* n_unique_dev, int64_t n, int64_t id_limit, int64_t* inverse, void* stream);
  // order[i] in [self_begin, self_begin + self_count)  ->  -(self_rows[order[i] - self_begin] + 1)   (see self_grads)
  int (*remap_self_order)(int32_t* order, int64_t n, int64_t self_begin, int64_t self_count, const int64_t* self_rows,
                          void* stream);
  // storage id -> WholeMemory row of a round-robin-loaded table: t = id / rr, off = id % rr, owner = t % world.
  // rank_rows > 0: row = owner * rank_rows + rr * (t / world) + off — the row file_io.cpp put that entry in (every rank
  // holds rank_rows rows). rank_rows == 0: the reference's statement as it stands, entry_start + rr * (t / world) + off
  // (map_indices_func.cu:34-43 computes the owner and never uses it), which is only right for ids the caller owns.
  int (*round_robin_map)(const void* ids, void* mapped, wholememory_dtype_t index_dtype, int64_t n,
                         int64_t entry_start, int world_size, int round_robin_size, int64_t rank_rows, void* stream);
  int (*fill_float)(float* p, float value, int64_t count, void* stream);
  // Duplicate estimate of a batch of lookup ids (the requester-side decision whether to de-duplicate before the exchange):
  // every k-th id is hashed into a bitmap; *permille_dev = 1000 * (1 - distinct / sampled) of the SAMPLE, by linear
  // counting. A heuristic: backends need not agree on the value. workspace from dup_estimate_workspace_bytes(n).
  size_t (*dup_estimate_workspace_bytes)(int64_t n);
  int (*dup_estimate)(const void* ids, wholememory_dtype_t index_dtype, int64_t n, void* workspace, int64_t* permille_dev,
                      void* stream);
  // per-owner counts of ids that are already SORTED (as unsigned keys, the order dedup_ids leaves them in; negative ids
  // come last and are not counted): counts[r] = #{ids in [entry_offsets[r], entry_offsets[r + 1])}. The number of ids is
  // read from *n_dev (<= n_upper).
  int (*sorted_owner_counts)(const void* sorted_ids, wholememory_dtype_t index_dtype, const int64_t* n_dev, int64_t n_upper,
                             const uint64_t* entry_offsets, int world_size, int64_t* counts, void* stream);
  // ---- graph ops (kernels/graph.hip); nullptr in a backend that does not provide them ----
  // counts[i] = min(degree(center i), max_sample) for i < n, counts[n] = 0
  // (row bounds from row_pairs when it is not nullptr, else through row_gref)
  // n_dev (optional, device): centres in use of the n the arrays are sized for; counts past them are 0
  int (*sample_counts)(const wholememory_gref_t* row_gref, int64_t row_storage_offset, const int64_t* row_pairs,
                       const void* centers, wholememory_dtype_t center_dtype, int n, const int* n_dev, int max_sample,
                       int* counts, void* stream);
  // ids[2i] = center i, ids[2i + 1] = center i + 1
  int (*sample_pair_ids)(const void* centers, wholememory_dtype_t center_dtype, int n, int64_t* ids, void* stream);
  // sample_counts + exclusive scan as ONE launch (mapped CSR only): offsets[0 .. n]; n_dev as in sample_counts; workspace of
  // scan_i32_workspace_bytes(n + 1). workspace_is_ones = 1: the caller has filled the workspace with 0xFF bytes (the scan's
  // state; it is left that way) — a chain of hops fills all its workspaces with one command; 0: the call fills it itself.
  // Returns -3 with nothing queued when it does not take the size. Optional: nullptr = run the two steps
  int (*sample_offsets)(const wholememory_gref_t* row_gref, int64_t row_storage_offset, const void* centers,
                        wholememory_dtype_t center_dtype, int n_center, const int* n_dev, int max_sample_count, int* offsets,
                        void* workspace, size_t workspace_bytes, int workspace_is_ones, void* stream);
  size_t (*scan_i32_workspace_bytes)(int64_t n);
  int (*exclusive_scan_i32)(const int* in, int* out, int64_t n, void* workspace, size_t workspace_bytes, void* stream);
  int (*sample_unweighted)(const wm_sample_args* a, void* stream);
  int (*sample_weighted)(const wm_sample_args* a, void* stream);  // max_sample_count in [1, 8192]
  // append_unique in two phases around the host learning the output size: phase 1 leaves the number of neighbour ids
  // that are not targets in *new_count_dev, phase 2 writes the unique array and the raw->unique mapping
  // n_neighbor_dev (optional): the neighbour array holds n_neighbor entries of ROOM, the number in use is read on the
  // device (the fused sample + append_unique op learns both counts with one host round trip). A backend route that
  // cannot work from a device count returns -3 before queueing anything; the caller then passes the exact count.
  // phase 2 takes the same n_neighbor as phase 1 (the scratch layout depends on it) and the number of entries in use.
  size_t (*append_unique_workspace_bytes)(int n_target, int n_neighbor, wholememory_dtype_t dtype);
  // publish_host (optional, PINNED host memory the device can write, 2 ints): the phase's last kernel leaves
  // {neighbours in use, new unique ids} there, so that the caller learns both with a stream synchronise and no copy
  // command (a D2H / D2D copy of 4 bytes is its own ~13 us entry in the GPU's queue)
  int (*append_unique_phase1)(const void* targets, int n_target, const void* neighbors, int n_neighbor,
                              const int* n_neighbor_dev, wholememory_dtype_t dtype, void* workspace, int* new_count_dev,
                              int* publish_host, const wm_au_bounds* bounds, void* stream);
  // copy_src / copy_dst (optional): n_neighbor_used ints moved by the emitting kernel on the side (the fused hop's centre
  // local ids, from their upper-bound scratch to the exactly sized output)
  int (*append_unique_phase2)(const void* targets, int n_target, int n_neighbor, int n_neighbor_used,
                              wholememory_dtype_t dtype, void* workspace, void* out_unique, int* mapping,
                              const int* copy_src, int* copy_dst, const wm_au_bounds* bounds, void* stream);
  int (*csr_add_self_loop)(const int* row_ptr, const int* col, int* out_row, int* out_col, int n_rows, void* stream);
  // can append_unique over arrays with room for n_target + n_neighbor ids run from device-side counts (wm_au_bounds)?
  // (the hash-table route can, the sort route needs the counts on the host)
  bool (*append_unique_takes_bounds)(int n_target, int n_neighbor, wholememory_dtype_t dtype);
  // out[i, c] = T(float(i)) + in[c] (wholememory_env_test_op)
  int (*env_test_fill)(const void* in, void* out, wholememory_dtype_t dtype, int64_t dim, int64_t entries, int64_t stride,
                       void* stream);
  // ---- embedding row cache (kernels/cache.hip); nullptr in a backend that does not provide it ----
  // unique_rows / run_starts / n_unique_dev: output of dedup_ids on the batch's ids (full-width keys); adds the batch to
  // the access counters and replaces least-frequently-used residents by more frequently used missing rows
  // fill_rows != nullptr ("plan" mode, read-only caches whose raw table is not addressable from this rank): no row is
  // moved; each decided (global row, slot) pair is appended to fill_rows / fill_slots (room for n_upper), *fill_count
  // (device, zeroed by the caller) counts them, and the caller fetches and installs the rows
  int (*cache_update)(const wm_cache_args* c, const void* unique_rows, wholememory_dtype_t index_dtype,
                      const int32_t* run_starts, const int64_t* n_unique_dev, int64_t n_upper, int64_t* fill_rows,
                      int64_t* fill_slots, int* fill_count, void* stream);
  // cache_idx[i] = slot of ids[i] or -1; raw_idx[i] = ids[i] if it misses (or is negative) else -1; *hits_dev += hits
  int (*cache_split)(const wm_cache_args* c, const void* ids, wholememory_dtype_t index_dtype, int64_t n, int64_t* cache_idx,
                     void* raw_idx, unsigned long long* hits_dev, void* stream);
  int (*cache_writeback)(const wm_cache_args* c, int drop, void* stream);
  int (*cache_info)(const wm_cache_args* c, unsigned long long* occupied_dirty_dev, void* stream);
  // placement probe (kernels/probe.hip): milliseconds per GiB of pseudo-random 512-byte rows of [ptr, ptr + bytes) touched,
  // averaged over `reps` launches. kind 0 = write zeros (fresh allocations only), 1 = read, 2 = read and write back.
  // Synchronises `stream`. nullptr in a backend that does not provide it.
  int (*probe_memory)(void* ptr, size_t bytes, int kind, int reps, float* ms_per_gib, void* stream);
  // Ids put in ascending row order for LOCALITY (the sorted-ids gather of HOST tables, gather_op.cpp:116-120 /
  // sort_indices_func.cu:41-91): sorted_ids[i] = ids[raw[i]] with the ids ascending as unsigned numbers over the bits
  // [low_bit, bits of key_upper_bound) — negative ("skip me") ids keep their value and come last; raw[] is int64 like every row
  // map. Any permutation would be correct for the gather that follows; only pairs must stay together. Returns -3 (nothing
  // queued) when the keys do not fit the 32-bit sort (key_upper_bound <= 0 or >= 2^32 - 1). Workspace from
  // sort_ids_workspace_bytes(n). nullptr in a backend that does not provide it.
  size_t (*sort_ids_workspace_bytes)(int64_t n);
  int (*sort_ids)(const void* ids, wholememory_dtype_t index_dtype, int64_t n, int64_t key_upper_bound, int low_bit,
                  void* sorted_ids, int64_t* raw, void* workspace, void* stream);
  // free / total bytes of the current device's memory right now. nullptr in a backend that does not provide it.
  int (*mem_info)(size_t* free_bytes, size_t* total_bytes);
  // ordinal of the device the calling thread works on (per-device locks). nullptr in a backend that does not provide it.
  int (*get_device)(int* device);
  // the part of an append_unique workspace that must be all-ones before phase 1 (the empty hash table); -3 when the route
  // these sizes take has none. nullptr in a backend that does not provide it.
  int (*append_unique_table_region)(int n_target, int n_neighbor, wholememory_dtype_t dtype, void* workspace, void** ptr,
                                    size_t* bytes);
  // bytes set to 0xFF by a KERNEL launch (what memset_async(.., 0xFF, ..) does as a memset command: inside a captured
  // hipGraph such a command was seen to run out of order with the kernels around it). nullptr: use memset_async.
  int (*fill_ff_async)(void* ptr, size_t bytes, void* stream);
  // dst = src in CHUNK-MAJOR order (ops.cpp: gather_distributed_rows). src holds n_segs segments, segment p = seg_counts[p]
  // elements from seg_offsets[p]; chunk c of a segment of n elements is [n*c/C, n*(c+1)/C) of it, C = n_chunks. dst lists
  // chunk 0 of every segment (in segment order), then chunk 1 of every segment ... — each chunk of the exchange pipeline
  // becomes ONE contiguous range, one kernel launch instead of one per peer. elt_bytes 4 or 8, n_segs <= 16 (the tables travel
  // as kernel arguments). nullptr in a backend that does not provide it: callers launch per segment.
  int (*permute_chunks)(const void* src, void* dst, int elt_bytes, const int64_t* seg_offsets, const int64_t* seg_counts,
                        int n_segs, int n_chunks, void* stream);
  // dedup_ids may leave work on a side stream of its own (a generic sort it usually does not need, kernels/optim.hip). By
  // default it joins that stream before it returns; a caller that has more work of its own to queue behind the sort — the
  // optimizer step — may ask (dedup_defer_join(1), calling thread only) for the join to be left to dedup_join(stream), which it
  // then MUST call on the same stream before it frees the sort's workspace or returns to its caller. The outputs of dedup_ids
  // are valid for work queued on `stream` either way. Both nullptr in a backend without such a side stream.
  void (*dedup_defer_join)(int on);
  int (*dedup_join)(void* stream);
  // Device-side waits of the id sort that gave up (kernels/split_sort.cuh: wait_cfg) leave a code in pinned memory after
  // turning their sort into "no runs". Returns and clears that code for the current device (0 = none; logs one ERROR line
  // otherwise). Non-blocking: it reports what has FINISHED, so callers ask after they synchronise and when they are entered
  // (dedup_ids and dedup_join ask too and fail with -2). nullptr in a backend without device-side waits.
  int (*device_error)(void);
  // p[i] = first + i for i < n (index dtype): the dense row numbers of a batch's runs (embedding.cpp: sender-side combination
  // of duplicate gradient rows folds run u into row u of a dense buffer)
  int (*fill_iota)(void* p, wholememory_dtype_t index_dtype, int64_t n, int64_t first, void* stream);
  // *flag_dev = -1 when row u of `rows` (HALF, [*, stride] elements, dim used) holds a value that is not finite for some run u
  // < *n_unique_dev of more than one id (run_starts as dedup_ids leaves them): a partial SUM of float16 gradients that left
  // the float16 range (a single gradient cannot). Leaves the flag alone otherwise. nullptr in a backend without 16-bit tables.
  int (*partials_nonfinite)(const int32_t* run_starts, const int64_t* n_unique_dev, int64_t n_upper, const void* rows,
                            int64_t dim, int64_t stride, int64_t* flag_dev, void* stream);
};

}  // extern "C"

namespace wm {
// Host copies of a chunked handle's per-rank tables, keyed by the DEVICE pointer array its gref carries (gref.pointer).
// memory_handle.cpp registers them when it uploads the device arrays; the row kernels of a table of up to
// kOwnersByValue ranks then get bases and bounds as kernel arguments instead of loading them per row. A gref somebody
// built by hand is simply not found and takes the device arrays.
constexpr int kOwnersByValue = 8;
struct gref_host_tables {
  int world_size;
  void* rank_ptrs[kOwnersByValue];
  size_t rank_offsets[kOwnersByValue + 1];   // bytes
};
void register_gref_tables(const void* dev_rank_ptrs, int world_size, void* const* rank_ptrs, const size_t* rank_offsets);
void unregister_gref_tables(const void* dev_rank_ptrs);
bool lookup_gref_tables(const void* dev_rank_ptrs, gref_host_tables* out);

const wm_device_backend* backend();      // the installed backend (HIP unless a test replaced it)
const wm_device_backend* hip_backend();  // kernels/backend_hip.hip
}  // namespace wm
