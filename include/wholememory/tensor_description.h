/*
 * wholegraph_amd — tensor / matrix / array descriptors of the WholeMemory C ABI.
 * Replaces reference cpp/include/wholememory/tensor_description.h:29-242. Enum values and struct
 * layouts are ABI: the Cython/ctypes side passes them by value.
 */
#ifndef WHOLEMEMORY_TENSOR_DESCRIPTION_H_
#define WHOLEMEMORY_TENSOR_DESCRIPTION_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference tensor_description.h:29-40 */
enum wholememory_dtype_t {
  WHOLEMEMORY_DT_UNKNOWN = 0,
  WHOLEMEMORY_DT_FLOAT   = 1, /* f32 */
  WHOLEMEMORY_DT_HALF    = 2, /* f16 */
  WHOLEMEMORY_DT_DOUBLE  = 3, /* f64 */
  WHOLEMEMORY_DT_BF16    = 4,
  WHOLEMEMORY_DT_INT     = 5, /* i32 */
  WHOLEMEMORY_DT_INT64   = 6,
  WHOLEMEMORY_DT_INT16   = 7,
  WHOLEMEMORY_DT_INT8    = 8,
  WHOLEMEMORY_DT_COUNT   = 9,
};

#define WHOLEMEMORY_MAX_TENSOR_DIM (8)

/* 1-D: reference tensor_description.h:64-68. Offsets are in ELEMENTS everywhere. */
struct wholememory_array_description_t {
  int64_t size;
  int64_t storage_offset;
  enum wholememory_dtype_t dtype;
};

/* 2-D row-major, row stride in elements: reference tensor_description.h:73-78 */
struct wholememory_matrix_description_t {
  int64_t sizes[2];
  int64_t stride;
  int64_t storage_offset;
  enum wholememory_dtype_t dtype;
};

/* N-D: reference tensor_description.h:83-90 */
struct wholememory_tensor_description_t {
  int64_t sizes[WHOLEMEMORY_MAX_TENSOR_DIM];
  int64_t strides[WHOLEMEMORY_MAX_TENSOR_DIM];
  int64_t storage_offset;
  int dim;
  enum wholememory_dtype_t dtype;
};

#ifndef __cplusplus
typedef enum wholememory_dtype_t wholememory_dtype_t;
typedef struct wholememory_array_description_t wholememory_array_description_t;
typedef struct wholememory_matrix_description_t wholememory_matrix_description_t;
typedef struct wholememory_tensor_description_t wholememory_tensor_description_t;
#endif

/* dtype queries — reference tensor_description.h:44-58; (size_t)-1 for an invalid dtype */
size_t wholememory_dtype_get_element_size(enum wholememory_dtype_t dtype);
bool wholememory_dtype_is_floating_number(enum wholememory_dtype_t dtype);
bool wholememory_dtype_is_integer_number(enum wholememory_dtype_t dtype);

/* constructors — reference tensor_description.h:99-119 */
struct wholememory_array_description_t wholememory_create_array_desc(
  int64_t size, int64_t storage_offset, enum wholememory_dtype_t dtype);
struct wholememory_matrix_description_t wholememory_create_matrix_desc(
  int64_t sizes[2], int64_t stride, int64_t storage_offset, enum wholememory_dtype_t dtype);
void wholememory_initialize_tensor_desc(struct wholememory_tensor_description_t* desc);

/* conversions — reference tensor_description.h:126-172 */
void wholememory_copy_array_desc_to_matrix(struct wholememory_matrix_description_t* dst,
                                           struct wholememory_array_description_t* src);
void wholememory_copy_array_desc_to_tensor(struct wholememory_tensor_description_t* dst,
                                           struct wholememory_array_description_t* src);
void wholememory_copy_matrix_desc_to_tensor(struct wholememory_tensor_description_t* dst,
                                            struct wholememory_matrix_description_t* src);
bool wholememory_convert_tensor_desc_to_array(struct wholememory_array_description_t* dst,
                                              struct wholememory_tensor_description_t* src);
bool wholememory_convert_tensor_desc_to_matrix(struct wholememory_matrix_description_t* dst,
                                               struct wholememory_tensor_description_t* src);

/* storage extents — reference tensor_description.h:179-224 */
int64_t wholememory_get_memory_element_count_from_array(struct wholememory_array_description_t* d);
int64_t wholememory_get_memory_size_from_array(struct wholememory_array_description_t* d);
int64_t wholememory_get_memory_element_count_from_matrix(struct wholememory_matrix_description_t* d);
int64_t wholememory_get_memory_size_from_matrix(struct wholememory_matrix_description_t* d);
int64_t wholememory_get_memory_element_count_from_tensor(struct wholememory_tensor_description_t* d);
int64_t wholememory_get_memory_size_from_tensor(struct wholememory_tensor_description_t* d);

/* shape edits — reference tensor_description.h:232-238 */
bool wholememory_squeeze_tensor(struct wholememory_tensor_description_t* desc, int dim);
bool wholememory_unsqueeze_tensor(struct wholememory_tensor_description_t* desc, int dim);

#ifdef __cplusplus
}
#endif
#endif
