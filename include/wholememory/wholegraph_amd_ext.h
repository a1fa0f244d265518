/*
 * wholegraph_amd — entry points that do NOT exist in the reference ABI.
 *
 *  (1) external-collective communicators: a communicator whose collectives are supplied by the
 *      host framework instead of RCCL. The product path never needs it on MI355X (RCCL over xGMI
 *      is the transport); it exists so the SAME orchestration code can be driven at world_size > 1
 *      over torch.distributed/gloo in the CPU test-suite, and for hosts that already own a
 *      communicator.
 *  (2) the raw device stages of the distributed path on plain device pointers (id bucketing,
 *      owner-side dedup + optimizer step), so parity tests can check each stage against the oracle
 *      through the C ABI.
 *  (3) the testing seam for the device backend (see wholegraph_amd/csrc/backend.hpp).
 */
#ifndef WHOLEMEMORY_WHOLEGRAPH_AMD_EXT_H_
#define WHOLEMEMORY_WHOLEGRAPH_AMD_EXT_H_

#include <wholememory/embedding.h>
#include <wholememory/wholememory_op.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- (1) external collectives ------------------------------------------------------------ */
/* All byte counts/displacements are in BYTES; every function returns 0 on success. `stream` is the
 * hipStream_t the library is working on; a provider that stages through the host must order itself
 * after/before that stream. */
struct wm_ext_collectives_t {
  void* ctx;
  int (*barrier)(void* ctx);
  /* recv[r*bytes .. (r+1)*bytes) = rank r's send; HOST buffers */
  int (*allgather_host)(void* ctx, const void* send, void* recv, size_t bytes);
  /* DEVICE buffers (whatever the installed device backend calls device memory) */
  int (*alltoallv_device)(void* ctx,
                          const void* send,
                          const size_t* send_bytes,
                          const size_t* send_disp,
                          void* recv,
                          const size_t* recv_bytes,
                          const size_t* recv_disp,
                          void* stream);
};
#ifndef __cplusplus
typedef struct wm_ext_collectives_t wm_ext_collectives_t;
#endif

enum wholememory_error_code_t wholememory_create_communicator_ext(
  wholememory_comm_t* comm, int rank, int size, const struct wm_ext_collectives_t* collectives);

/* Which transport carries this communicator's collectives: name = "rccl" | "external" | "external (sub-group)" | "none"
 * (a single-rank communicator needs none); ranks = what the transport itself reports (ncclCommCount for RCCL, -1 when
 * it cannot tell, 0 for "none"). bench.py prints it so that an N-GPU line proves N RCCL ranks. */
enum wholememory_error_code_t wholememory_ext_communicator_transport(wholememory_comm_t comm, const char** name, int* ranks);

/* ---- (2) raw stages ------------------------------------------------------------------------ */
/* Owner bucketing of ids (reference bucket_ids_func.cu:51-87 + the grouping effect of
 * exchange_ids_nccl_func.cu:42-92). entry_offsets: DEVICE uint64[world+1] row offsets. counts:
 * DEVICE int64[world]. bucketed_ids (index dtype) / raw_indices (int64): DEVICE [n] or both NULL for
 * counts only. Scratch comes from p_env_fns. Asynchronous on stream. */
enum wholememory_error_code_t wholememory_ext_bucket_ids(const void* indices,
                                                         enum wholememory_dtype_t index_dtype,
                                                         int64_t n,
                                                         const void* entry_offsets_dev,
                                                         int world_size,
                                                         int64_t* counts_dev,
                                                         void* bucketed_ids_dev,
                                                         int64_t* raw_indices_dev,
                                                         struct wholememory_env_func_t* p_env_fns,
                                                         void* stream);

/* The same with more row ranges than buckets: entry_offsets has owner_count+1 entries (owner_count >= bucket_count) and
 * an id of range o goes to bucket o % bucket_count — the first hop of the HIERARCHY gather (ranges = ranks of every node,
 * buckets = ranks of this node; reference bucket_and_reorder_ids_for_hierarchy_func, gather_op_impl_hierarchy.cu:194-205).
 * counts: DEVICE int64[bucket_count]. */
enum wholememory_error_code_t wholememory_ext_bucket_ids_folded(const void* indices,
                                                                enum wholememory_dtype_t index_dtype,
                                                                int64_t n,
                                                                const void* entry_offsets_dev,
                                                                int owner_count,
                                                                int bucket_count,
                                                                int64_t* counts_dev,
                                                                void* bucketed_ids_dev,
                                                                int64_t* raw_indices_dev,
                                                                struct wholememory_env_func_t* p_env_fns,
                                                                void* stream);

/* Owner-side "dedup + optimizer step" on received (ids, grads): reference
 * exchange_embeddings_nccl_func.cu:76-174 followed by embedding_optimizer_func.cu steps, fused.
 * opt_params: float[6] = {weight_decay, epsilon, beta1, beta2, alpha, adam_w}. State pointers may be
 * NULL when the optimizer has none. Synchronises `stream` before returning the unique count. */
enum wholememory_error_code_t wholememory_ext_dedup_apply(const void* recv_ids,
                                                          enum wholememory_dtype_t index_dtype,
                                                          int64_t n_recv,
                                                          const float* recv_grads,
                                                          int64_t grad_stride,
                                                          int64_t dim,
                                                          float* local_table,
                                                          int64_t table_stride,
                                                          int64_t local_entry_offset,
                                                          int64_t local_entry_count,
                                                          enum wholememory_optimizer_type_t opt_type,
                                                          const float* opt_params,
                                                          float lr,
                                                          float* per_element_state,
                                                          float* per_row_state,
                                                          int64_t* n_unique_host,
                                                          struct wholememory_env_func_t* p_env_fns,
                                                          void* stream);

/* round-robin id remap (reference map_indices_func.cu:26-106) on raw device pointers */
enum wholememory_error_code_t wholememory_ext_round_robin_map(const void* ids,
                                                              void* mapped,
                                                              enum wholememory_dtype_t index_dtype,
                                                              int64_t n,
                                                              int64_t entry_start,
                                                              int world_size,
                                                              int round_robin_size,
                                                              void* stream);

/* name of the installed device backend ("hip-gfx950" in the product) */
const char* wholememory_ext_backend_name();

/* demangled name the HIP runtime holds (hipKernelNameRefByPtr) for the gather / scatter kernel instantiation the calling
 * thread launched last; "" before the first launch */
const char* wholememory_ext_last_rows_kernel();

/* One hop of multi-layer neighbour sampling as one call: unweighted sampling without replacement of every node of
 * `center_nodes_tensor` followed by graph_append_unique(center nodes, sampled neighbours), with ONE host round trip to size
 * the outputs instead of the two of the separate ops (reference: wholegraph_csr_unweighted_sample_without_replacement,
 * include/wholememory/wholegraph_op.h, + graph_append_unique, graph_op.h, driven per hop by
 * python/pylibwholegraph/pylibwholegraph/torch/graph_structure.py:140-196). Outputs, bit-identical to that sequence:
 *   output_sample_offset_tensor  int32 [n_center + 1], caller-allocated (csr_row_ptr of the sampled block)
 *   unique        (memory context)  center nodes ++ new neighbour ids in first-occurrence order
 *   neighbor_pos  (memory context)  int32 [n_samples]: position of each sampled neighbour in `unique`
 *   center_lid    (memory context)  int32 [n_samples]: position of its centre in center_nodes_tensor
 * WHOLEMEMORY_NOT_SUPPORTED (nothing queued, nothing allocated) when the CSR is not mapped into this rank (DISTRIBUTED /
 * HIERARCHY), when column ids and center ids differ in dtype, when max_sample_count <= 0 or the center array is empty:
 * the caller then runs the two ops. */
wholememory_error_code_t wholememory_ext_sample_append_unique(
  wholememory_tensor_t wm_csr_row_ptr_tensor, wholememory_tensor_t wm_csr_col_ptr_tensor,
  wholememory_tensor_t center_nodes_tensor, int max_sample_count, unsigned long long random_seed,
  wholememory_tensor_t output_sample_offset_tensor, void* output_unique_memory_context,
  void* output_neighbor_pos_memory_context, void* output_center_localid_memory_context, wholememory_env_func_t* p_env_fns,
  void* stream);

/* Every hop of a multi-layer unweighted sample (hop = neighbour sample + graph_append_unique, as in
 * wholememory_ext_sample_append_unique) in ONE call with NO host round trip inside. The caller sizes every array for its upper
 * bound: hop h (h = 0 next to the seeds) has at most cap_c[h] centres and cap_s[h] = cap_c[h] * max_sample_counts[h] samples,
 * cap_c[0] = number of seeds, cap_c[h + 1] = cap_c[h] + cap_s[h]:
 *   sample_offsets[h]  int32 [cap_c[h] + 1]        unique[h]       id dtype [cap_c[h] + cap_s[h]]
 *   neighbor_pos[h]    int32 [cap_s[h]]            center_lid[h]   int32 [cap_s[h]]
 * all DEVICE memory; counts_host: PINNED host memory the device can write, 2 * hops ints — the last kernel of hop h leaves
 * {samples, new unique ids} of that hop in counts_host[2h], counts_host[2h + 1]. After ONE stream synchronise the caller
 * trims: with n_c[0] = seeds and n_c[h + 1] = n_c[h] + new[h], hop h's csr_row_ptr is sample_offsets[h][0 .. n_c[h]], its
 * widened frontier unique[h][0 .. n_c[h + 1]), neighbor_pos / center_lid the first samples[h] entries. Bit-identical to `hops`
 * calls of wholememory_ext_sample_append_unique with the same seeds. WHOLEMEMORY_NOT_SUPPORTED (nothing queued) when the CSR is
 * not mapped into this rank, the id dtypes differ, there is no seed, a fan-out is <= 0 or a hop's upper bounds exceed what
 * graph_append_unique's hash-table route takes (the caller then runs hop by hop). With sample_offsets == NULL the call is a QUERY:
 * it answers SUCCESS / NOT_SUPPORTED for these tensors, hops and fan-outs and touches nothing else — ask before allocating the
 * upper-bound buffers. Reference call sequence:
 * python/pylibwholegraph/pylibwholegraph/torch/graph_structure.py:140-196. */
enum wholememory_error_code_t wholememory_ext_multilayer_sample(
  wholememory_tensor_t wm_csr_row_ptr_tensor, wholememory_tensor_t wm_csr_col_ptr_tensor, wholememory_tensor_t seed_nodes_tensor,
  int hops, const int* max_sample_counts, const unsigned long long* random_seeds, void* const* sample_offsets,
  void* const* unique, int* const* neighbor_pos, int* const* center_lid, int* counts_host,
  struct wholememory_env_func_t* p_env_fns, void* stream);

/* Completion semantics of the ops whose reference versions drain the stream before returning (neighbour sampling,
 * graph_append_unique and the fused hop above). Default 0 = the reference's: outputs complete and scratch idle at return,
 * safe with any env functions. 1 = the ops return with their last kernels queued on `stream` (one host round trip fewer
 * per call); only legal when every allocator behind p_env_fns is stream-ordered on that stream and every consumer of the
 * outputs is ordered on it — wholegraph_amd.torch declares it for torch's caching allocator. Process-wide;
 * WM_ASYNC_OPS=0/1 in the environment overrides the call. */
enum wholememory_error_code_t wholememory_ext_set_async_completion(int on);

/* Environment knobs (WM_* / WG_* variables) are read ONCE, the first time the code that consults one runs; no op calls
 * getenv afterwards. A program that changes such a variable later (A/B experiments, tests) calls this to make every knob
 * read its variable again at its next use. Not to be called while ops run on other threads. */
enum wholememory_error_code_t wholememory_ext_reload_knobs(void);

/* Placement probe: milliseconds per GiB of pseudo-random 512-byte rows of [ptr, ptr + bytes) touched by a fixed kernel
 * (kind 0 = zeros written: destroys the contents, 1 = read, 2 = read and written back), averaged over `reps` launches after a
 * warm-up; blocks until done. The level the memory system serves random row accesses at depends on where a large allocation
 * sits in HBM and stays with it for its lifetime. wholememory_malloc makes ONE allocation per device
 * shard, like the reference; WM_MALLOC_PROBE=auto (self-calibrating: candidates are added until two agree with the best seen
 * within 3 %) or =K (exactly K candidates) opts into choosing the shard with this probe — the candidates are alive together,
 * capped at a quarter of the free memory, one prober per device at a time. */
enum wholememory_error_code_t wholememory_ext_probe_memory(void* ptr, size_t bytes, int kind, int reps, float* ms_per_gib);

/* The placement probe of wholememory_malloc (above) without the environment: "auto", "2" ... "8" (that many candidates), "off",
 * or "env" / NULL (follow WM_MALLOC_PROBE again, the initial state). Applies to the device allocations this process makes from
 * now on, until changed; WHOLEMEMORY_INVALID_INPUT for anything else. pylibwholegraph: create_embedding(...,
 * placement_probe="auto") sets it around the one creation. Tables that are written at random — scatter targets, trained
 * embeddings — are what it is for: their speed follows the placement of the shard by up to 20 %. */
enum wholememory_error_code_t wholememory_ext_set_malloc_probe(const char* mode);
/* The mode set by the call above ("env" when none is), written to mode[capacity >= 8]: for callers that set a mode around ONE
 * allocation and restore what was there before. */
enum wholememory_error_code_t wholememory_ext_get_malloc_probe(char* mode, size_t capacity);
/* 1 when the local shard of the handle was chosen among several probed candidates, else 0 */
int wholememory_ext_handle_was_probed(wholememory_handle_t handle);

/* Number of wholememory_gather calls of this process that took the sorted-ids route of HOST-located tables (rows of at most
 * 512 bytes, batches of at least WM_HOST_SORTED_MIN ids: wholememory_op.h / gather_op.cpp:116-120 of the reference).
 * A counter for tests and benchmarks. */
int64_t wholememory_ext_host_sorted_gathers(void);

/* Number of owner-side id sorts (gradient apply, cached gather) of this process that were queued as the two-stage split sort
 * (csrc/kernels/split_sort.cuh: bounded ids, at least WM_DEDUP_SPLIT_MIN of them, a bucket plan that fits) rather than as the
 * generic radix sort. Whether a queued split sort then found a bucket too large and handed the batch to the gated generic path
 * is decided on the device and not visible here. A counter for tests and benchmarks. */
int64_t wholememory_ext_split_sorts(void);

/* ... and how many of those ran in HOT mode (round 6): a batch whose plain split sort overflowed a bucket is followed by split
 * sorts that first sample the batch and give its hot ids buckets of their own (csrc/kernels/split_sort.cuh: launch_hot), so a
 * Zipf-skewed series no longer goes to the generic sort. Opt-in (WM_DEDUP_HOT=1): correct, not yet faster than rocPRIM's sort on
 * such batches. A counter for tests and benchmarks. */
int64_t wholememory_ext_hot_split_sorts(void);

/* Duplicate runs of the last finished ORDERED gradient step on the current device that were summed through a dense transposed
 * copy of their gradient rows (csrc/kernels/long_dense.cuh: runs of at least WM_DENSE_FOLD_MIN rows, default 131072, while
 * 3 n / 8 rows of copies last; WM_DENSE_FOLD=0 switches the route off). Read after a synchronise. A counter for tests. */
int64_t wholememory_ext_dense_fold_last(void);

/* Kernels queued so far by the DISTRIBUTED gather route of this process (owner-side row gathers, reorder-on-receive
 * scatters, the two chunk-major copies): with C exchange chunks a call costs at most 2 C + 3 of them whatever the number of
 * ranks (rounds 2-4: 2 (W - 1) C + 1). A counter for tests. */
int64_t wholememory_ext_distributed_gather_launches(void);

/* The same for the DISTRIBUTED scatter route (line-up gathers of the rows to send, owner-side writes, this rank's own rows,
 * the two chunk-major copies): at most 2 C + 3 per call (rounds 1-5: 2 (W - 1) C + 1). A counter for tests. */
int64_t wholememory_ext_distributed_scatter_launches(void);

/* Row kernels queued so far in FRONT of the gradient exchange of wholememory_embedding_gather_gradient_apply (the chunk-major
 * copy of the positions, the line-up gathers of the gradient rows to send, a copy of this rank's own rows when they are not
 * read in place): at most C + 2 per call whatever the number of ranks (rounds 1-5: (W - 1) C + 1). The receive side launches
 * nothing per chunk: rows arrive in rank-major order, which is the order of the fp32 sum of duplicates. A counter for tests. */
int64_t wholememory_ext_gradient_exchange_launches(void);

/* Bytes this process has handed to the all-to-all-v transport for OTHER ranks so far (ids and rows of every distributed op;
 * this rank's own segment counts where it travels like a peer's: WM_EXCHANGE_SELF=1). A counter for tests and for the
 * first-contact report: the sender-side combination of duplicate gradient rows halves it on a Zipf(1.05) batch. */
int64_t wholememory_ext_alltoallv_bytes(void);

/* Calls of wholememory_embedding_gather_gradient_apply that took the COMBINED route: several ranks, a fold that is not bound to
 * the reference's order (optimizer parameter grad_fold = tree / WM_GRAD_FOLD=tree / a 16-bit table) and enough duplicates
 * (decided by all ranks from the duplicate estimates in the counts exchange; WM_GRAD_COMBINE=0|1 forces): every rank folds ITS
 * duplicates of an id into one partial sum before the exchange, the owner folds at most world_size partial rows per id. */
int64_t wholememory_ext_combined_gradient_calls(void);

/* ---- (3) testing seam ---------------------------------------------------------------------- */
/* Replaces the device backend. Refuses (WHOLEMEMORY_NOT_SUPPORTED) unless the environment has
 * WHOLEGRAPH_AMD_TESTING=1. `backend` is a const wm_device_backend* (wholegraph_amd/csrc/backend.hpp);
 * NULL restores the HIP backend. Never called by product code. */
enum wholememory_error_code_t wm_testing_install_backend(const void* backend);

/* Occupancy / effectiveness of an embedding's device row cache: slots, occupied slots, modified slots, lookups served
 * from the cache and lookups seen so far (this rank). INVALID_INPUT for an embedding without cache. */
enum wholememory_error_code_t wholememory_ext_embedding_cache_info(wholememory_embedding_t embedding, int64_t* slots,
                                                                   int64_t* occupied, int64_t* dirty, int64_t* hits,
                                                                   int64_t* lookups);

#ifdef __cplusplus
}
#endif
#endif
