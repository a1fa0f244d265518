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
 * unique(target ++ neighbor) with the targets kept first and in order: output[0:T] = targets, followed by the
 * neighbour ids that are not targets, each once. The reference leaves the order of that tail unspecified (hash-slot
 * order); here it is first-occurrence order, which is what the reference's host test oracle produces.
 * output_neighbor_raw_to_unique_mapping (optional int32 [n_neighbor]): position of every neighbour in the output.
 * The output array (dtype of the inputs) is allocated through p_env_fns->output_fns. Synchronises `stream`.
 */
enum wholememory_error_code_t graph_append_unique(
  wholememory_tensor_t target_nodes_tensor,
  wholememory_tensor_t neighbor_nodes_tensor,
  void* output_unique_node_memory_context,
  wholememory_tensor_t output_neighbor_raw_to_unique_mapping_tensor,
  struct wholememory_env_func_t* p_env_fns,
  void* stream);

/* int32 CSR -> CSR with node i's own id inserted in front of its neighbours (no check for existing loops):
 * out_row[i] = row[i] + i, out_col has n_edges + n_rows entries. */
enum wholememory_error_code_t csr_add_self_loop(wholememory_tensor_t csr_row_ptr_tensor,
                                                wholememory_tensor_t csr_col_ptr_tensor,
                                                wholememory_tensor_t output_csr_row_ptr_tensor,
                                                wholememory_tensor_t output_csr_col_ptr_tensor,
                                                void* stream);

#ifdef __cplusplus
}
#endif
#endif
