/*
 * wholegraph_amd — neighbour sampling on CSR graphs held in WholeMemory (the step before the feature gather in
 * BASELINE config 5). Replaces reference cpp/include/wholememory/wholegraph_op.h:39-105.
 *
 * Built: unweighted sampling without replacement on every memory type (DISTRIBUTED CSR tensors are read through
 * collective wholememory_gather calls: every rank of the CSR's communicator must take part), weighted sampling
 * (max_sample_count <= 8192, mapped CSR tensors) and the two host random helpers. Weighted sampling of more than 8192
 * neighbours or on DISTRIBUTED tensors returns WHOLEMEMORY_NOT_IMPLEMENTED.
 * Random streams: see wholegraph_amd/csrc/pcg.hpp (parity with raft unpinned).
 */
#ifndef WHOLEMEMORY_WHOLEGRAPH_OP_H_
#define WHOLEMEMORY_WHOLEGRAPH_OP_H_

#include <wholememory/env_func_ptrs.h>
#include <wholememory/wholememory.h>
#include <wholememory/wholememory_tensor.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * For every center node: min(degree, fanout) distinct neighbours (all of them when fanout <= 0).
 *   row_ptr  int64 [n_nodes + 1], col_idx int32/int64 [n_edges]   (WholeMemory or plain tensors)
 *   centers  int32/int64 device tensor [n]
 *   offsets  int32 device tensor [n + 1], written: exclusive prefix of the per-center sample counts
 * The variable-size results are allocated through env->output_fns with the caller's memory contexts:
 *   dest_ctx  sampled neighbour ids (dtype of col_idx)                       — required
 *   lid_ctx   index of the center node each sample belongs to (int32)        — optional (NULL)
 *   egid_ctx  global edge id of each sample (int64)                          — optional (NULL)
 * Synchronises `stream`. reference wholegraph_op.h:39-50
 */
enum wholememory_error_code_t wholegraph_csr_unweighted_sample_without_replacement(
  wholememory_tensor_t row_ptr, wholememory_tensor_t col_idx, wholememory_tensor_t centers, int fanout,
  wholememory_tensor_t offsets, void* dest_ctx, void* lid_ctx, void* egid_ctx, unsigned long long seed,
  struct wholememory_env_func_t* env, void* stream);

/*
 * Weighted variant (A-Res: key = log2(u) / weight per neighbour, the `fanout` largest keys win).
 * weights: float32/float64 [n_edges]. Samples of one center node come out key-descending.
 * fanout > 8192 -> WHOLEMEMORY_NOT_IMPLEMENTED. reference wholegraph_op.h:70-82
 */
enum wholememory_error_code_t wholegraph_csr_weighted_sample_without_replacement(
  wholememory_tensor_t row_ptr, wholememory_tensor_t col_idx, wholememory_tensor_t weights, wholememory_tensor_t centers,
  int fanout, wholememory_tensor_t offsets, void* dest_ctx, void* lid_ctx, void* egid_ctx, unsigned long long seed,
  struct wholememory_env_func_t* env, void* stream);

/*
 * Host helpers the reference's Python tests use to re-derive expected samples (wholegraph_op.h:91-105): `out` is a HOST
 * tensor (int32/int64, resp. float32) filled from stream `subsequence` of generator `seed`.
 */
enum wholememory_error_code_t generate_random_positive_int_cpu(int64_t seed, int64_t subsequence, wholememory_tensor_t out);
enum wholememory_error_code_t generate_exponential_distribution_negative_float_cpu(int64_t seed, int64_t subsequence,
                                                                                   wholememory_tensor_t out);

#ifdef __cplusplus
}
#endif
#endif
