/*
 * wholegraph_amd — row gather / scatter on WholeMemory tensors. Replaces reference
 * cpp/include/wholememory/wholememory_op.h:36-58.
 *
 *   gather : output[i, :] = cast(table[indices[i], :])       rows with indices[i] < 0 are skipped
 *   scatter: table[indices[i], :] = cast(input[i, :])        (overwrite; duplicate ids unordered)
 *
 * indices: 1-D int32/int64 device tensor. table/input/output: 1-D or 2-D, both floating or both
 * integer. `stream` is a hipStream_t. CHUNKED/CONTINUOUS tables: rank-local and asynchronous.
 * DISTRIBUTED tables: collective over the table's communicator (ids and rows travel by RCCL
 * all-to-all-v over xGMI) and the stream is synchronised inside the call, as in the reference.
 * gather_sms / scatter_sms cap the number of workgroups (reference: "SMs"); -1 = library default.
 */
#ifndef WHOLEMEMORY_WHOLEMEMORY_OP_H_
#define WHOLEMEMORY_WHOLEMEMORY_OP_H_

#include <wholememory/env_func_ptrs.h>
#include <wholememory/wholememory_tensor.h>

#ifdef __cplusplus
extern "C" {
#endif

enum wholememory_error_code_t wholememory_gather(wholememory_tensor_t wholememory_tensor,
                                                 wholememory_tensor_t indices_tensor,
                                                 wholememory_tensor_t output_tensor,
                                                 struct wholememory_env_func_t* p_env_fns,
                                                 void* stream,
                                                 int gather_sms WM_DEFAULT(-1));

enum wholememory_error_code_t wholememory_scatter(wholememory_tensor_t input_tensor,
                                                  wholememory_tensor_t indices_tensor,
                                                  wholememory_tensor_t wholememory_tensor,
                                                  struct wholememory_env_func_t* p_env_fns,
                                                  void* stream,
                                                  int scatter_sms WM_DEFAULT(-1));

/*
 * Self-test of the env functions: output[i, :] = T(float(i)) + input[:], computed in scratch memory from
 * p_env_fns->temporary_fns, copied to output_fixed_tensor ([output_variable_entry_count, len(input)], dense) and to
 * device / pinned / host tensors allocated through p_env_fns->output_fns for each non-null memory context.
 * reference wholememory_op.h:58-79
 */
enum wholememory_error_code_t wholememory_env_test_op(wholememory_tensor_t input_tensor,
                                                      wholememory_tensor_t output_fixed_tensor,
                                                      void* output_variable_device_tensor_handle,
                                                      void* output_variable_pinned_tensor_handle,
                                                      void* output_variable_host_tensor_handle,
                                                      int64_t output_variable_entry_count,
                                                      struct wholememory_env_func_t* p_env_fns,
                                                      void* stream);

#ifdef __cplusplus
}
#endif
#endif
