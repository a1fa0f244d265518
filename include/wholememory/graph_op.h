/*
 * wholegraph_amd — small graph utilities used between sampling hops. Replaces reference
 * cpp/include/wholememory/graph_op.h:36-59.
 */
#ifndef WHOLEMEMORY_GRAPH_OP_H_
#define WHOLEMEMORY_GRAPH_OP_H_

#include <wholememory/env_func_ptrs.h>
#include <wholememory/wholememory.h>
#include <wholememory/wholememory_tensor.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * unique(targets ++ neighbors) with the targets kept first and in order: result[0:T] = targets, followed by the neighbour
 * ids that are not targets, each once. The reference leaves the order of that tail unspecified (hash-slot order); here
 * it is first-occurrence order, which is what the reference's host test oracle produces.
 *   unique_ctx   memory context of the result (dtype of the inputs), allocated through env->output_fns
 *   raw_to_unique  optional int32 [n_neighbors]: position of every neighbour in the result
 * Synchronises `stream`.
 */
enum wholememory_error_code_t graph_append_unique(wholememory_tensor_t targets, wholememory_tensor_t neighbors,
                                                  void* unique_ctx, wholememory_tensor_t raw_to_unique,
                                                  struct wholememory_env_func_t* env, void* stream);

/*
 * int32 CSR -> CSR with node i's own id inserted in front of its neighbours (no check for existing loops):
 * out_row_ptr[i] = row_ptr[i] + i; out_col_idx has n_edges + n_rows entries.
 */
enum wholememory_error_code_t csr_add_self_loop(wholememory_tensor_t row_ptr, wholememory_tensor_t col_idx,
                                                wholememory_tensor_t out_row_ptr, wholememory_tensor_t out_col_idx,
                                                void* stream);

#ifdef __cplusplus
}
#endif
#endif
