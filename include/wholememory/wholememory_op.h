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

/* output[i, :] = cast(table[indices[i], :]); negative indices leave their output row untouched. `table` is a
 * WholeMemory tensor of any memory type (DISTRIBUTED: collective) or a plain device tensor; sms caps the grid (-1 =
 * default). reference wholememory_op.h:36-43 */
enum wholememory_error_code_t wholememory_gather(wholememory_tensor_t table, wholememory_tensor_t indices,
                                                 wholememory_tensor_t output, struct wholememory_env_func_t* env,
                                                 void* stream, int sms WM_DEFAULT(-1));

/* table[indices[i], :] = cast(input[i, :]); overwrite, unordered for duplicate indices. reference :47-54 */
enum wholememory_error_code_t wholememory_scatter(wholememory_tensor_t input, wholememory_tensor_t indices,
                                                  wholememory_tensor_t table, struct wholememory_env_func_t* env,
                                                  void* stream, int sms WM_DEFAULT(-1));

/*
 * Self-test of the env functions: fixed_out[i, :] = T(float(i)) + input[:], computed in scratch memory from
 * env->temporary_fns, copied to fixed_out ([entries, len(input)], dense) and to device / pinned / host tensors
 * allocated through env->output_fns for each non-null memory context. reference wholememory_op.h:58-79
 */
enum wholememory_error_code_t wholememory_env_test_op(wholememory_tensor_t input, wholememory_tensor_t fixed_out,
                                                      void* device_ctx, void* pinned_ctx, void* host_ctx, int64_t entries,
                                                      struct wholememory_env_func_t* env, void* stream);

#ifdef __cplusplus
}
#endif
#endif
