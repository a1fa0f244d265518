/*
 * wholegraph_amd — neighbour sampling on CSR graphs held in WholeMemory (the step before the feature gather in
 * BASELINE config 5). Replaces reference cpp/include/wholememory/wholegraph_op.h:39-105.
 *
 * Built: unweighted sampling without replacement on every memory type (DISTRIBUTED CSR tensors are read through
 * collective wholememory_gather calls: every rank of the CSR's communicator must take part), weighted sampling
 * (max_sample_count <= 1024, mapped CSR tensors) and the two host random helpers. Weighted sampling of more than 1024
 * neighbours or on DISTRIBUTED tensors returns WHOLEMEMORY_NOT_IMPLEMENTED. Random streams: see wholegraph_amd/csrc/pcg.hpp (parity with raft unpinned).
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
 * For every center node: min(degree, max_sample_count) distinct neighbours (all of them when
 * max_sample_count <= 0). csr_row_ptr: int64 [n_nodes + 1]; csr_col_ptr: int32/int64 [n_edges]; center nodes:
 * int32/int64 device tensor; output_sample_offset: int32 device tensor [n_center + 1] (exclusive prefix of the
 * per-center counts). The variable-size outputs are allocated through p_env_fns->output_fns with the caller's
 * memory contexts: sampled neighbour ids (col dtype), optional center-local index per sample (int32), optional
 * global edge id per sample (int64). Synchronises `stream`. reference wholegraph_op.h:39-50
 */
enum wholememory_error_code_t wholegraph_csr_unweighted_sample_without_replacement(
  wholememory_tensor_t wm_csr_row_ptr_tensor,
  wholememory_tensor_t wm_csr_col_ptr_tensor,
  wholememory_tensor_t center_nodes_tensor,
  int max_sample_count,
  wholememory_tensor_t output_sample_offset_tensor,
  void* output_dest_memory_context,
  void* output_center_localid_memory_context,
  void* output_edge_gid_memory_context,
  unsigned long long random_seed,
  struct wholememory_env_func_t* p_env_fns,
  void* stream);

/*
 * Weighted variant (A-Res: key = log2(u) / weight per neighbour, the max_sample_count largest keys win).
 * wm_csr_weight_ptr_tensor: float32/float64 [n_edges]. Samples of one center node come out key-descending.
 * max_sample_count > 1024 -> WHOLEMEMORY_NOT_IMPLEMENTED. reference wholegraph_op.h:70-82
 */
enum wholememory_error_code_t wholegraph_csr_weighted_sample_without_replacement(
  wholememory_tensor_t wm_csr_row_ptr_tensor,
  wholememory_tensor_t wm_csr_col_ptr_tensor,
  wholememory_tensor_t wm_csr_weight_ptr_tensor,
  wholememory_tensor_t center_nodes_tensor,
  int max_sample_count,
  wholememory_tensor_t output_sample_offset_tensor,
  void* output_dest_memory_context,
  void* output_center_localid_memory_context,
  void* output_edge_gid_memory_context,
  unsigned long long random_seed,
  struct wholememory_env_func_t* p_env_fns,
  void* stream);

/* host helpers the reference's Python tests use to re-derive expected samples (wholegraph_op.h:91-105):
 * `output` is a HOST tensor (int32/int64, resp. float32) filled from stream `subsequence` of seed `random_seed` */
enum wholememory_error_code_t generate_random_positive_int_cpu(int64_t random_seed,
                                                               int64_t subsequence,
                                                               wholememory_tensor_t output);
enum wholememory_error_code_t generate_exponential_distribution_negative_float_cpu(
  int64_t random_seed, int64_t subsequence, wholememory_tensor_t output);

#ifdef __cplusplus
}
#endif
#endif
