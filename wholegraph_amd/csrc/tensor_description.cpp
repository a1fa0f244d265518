// wholegraph_amd — descriptor helpers of the C ABI (pure host code).
// Behaviour follows reference cpp/src/wholememory/tensor_description.cpp:25-237 call for call;
// tests/test_tensor_description.py compares every function here against the reference TU itself
// (compiled in place into oracle/_ref/).
#include <wholememory/tensor_description.h>

extern "C" {

size_t wholememory_dtype_get_element_size(wholememory_dtype_t dtype)
{
  static const size_t kSize[WHOLEMEMORY_DT_COUNT] = {0, 4, 2, 8, 2, 4, 8, 2, 1};
  if (dtype < WHOLEMEMORY_DT_UNKNOWN || dtype >= WHOLEMEMORY_DT_COUNT) return static_cast<size_t>(-1);
  return kSize[dtype];
}

bool wholememory_dtype_is_floating_number(wholememory_dtype_t dtype)
{
  return dtype == WHOLEMEMORY_DT_FLOAT || dtype == WHOLEMEMORY_DT_HALF || dtype == WHOLEMEMORY_DT_DOUBLE ||
         dtype == WHOLEMEMORY_DT_BF16;
}

bool wholememory_dtype_is_integer_number(wholememory_dtype_t dtype)
{
  return dtype == WHOLEMEMORY_DT_INT || dtype == WHOLEMEMORY_DT_INT64 || dtype == WHOLEMEMORY_DT_INT16 ||
         dtype == WHOLEMEMORY_DT_INT8;
}

wholememory_array_description_t wholememory_create_array_desc(int64_t size,
                                                              int64_t storage_offset,
                                                              wholememory_dtype_t dtype)
{
  wholememory_array_description_t d;
  d.size = size, d.storage_offset = storage_offset, d.dtype = dtype;
  return d;
}

wholememory_matrix_description_t wholememory_create_matrix_desc(int64_t sizes[2],
                                                                int64_t stride,
                                                                int64_t storage_offset,
                                                                wholememory_dtype_t dtype)
{
  wholememory_matrix_description_t d;
  d.sizes[0] = sizes[0], d.sizes[1] = sizes[1];
  d.stride = stride, d.storage_offset = storage_offset, d.dtype = dtype;
  return d;
}

void wholememory_initialize_tensor_desc(wholememory_tensor_description_t* t)
{
  for (int i = 0; i < WHOLEMEMORY_MAX_TENSOR_DIM; i++) t->sizes[i] = t->strides[i] = 1;
  t->storage_offset = 0;
  t->dim            = 0;
  t->dtype          = WHOLEMEMORY_DT_UNKNOWN;
}

void wholememory_copy_array_desc_to_matrix(wholememory_matrix_description_t* m,
                                           wholememory_array_description_t* a)
{
  m->sizes[0] = a->size, m->sizes[1] = 1, m->stride = 1;
  m->storage_offset = a->storage_offset;
  m->dtype          = a->dtype;
}

void wholememory_copy_array_desc_to_tensor(wholememory_tensor_description_t* t,
                                           wholememory_array_description_t* a)
{
  wholememory_initialize_tensor_desc(t);
  t->dim = 1, t->sizes[0] = a->size, t->strides[0] = 1;
  t->storage_offset = a->storage_offset;
  t->dtype          = a->dtype;
}

void wholememory_copy_matrix_desc_to_tensor(wholememory_tensor_description_t* t,
                                            wholememory_matrix_description_t* m)
{
  wholememory_initialize_tensor_desc(t);
  t->dim      = 2;
  t->sizes[0] = m->sizes[0], t->sizes[1] = m->sizes[1];
  t->strides[0] = m->stride, t->strides[1] = 1;
  t->storage_offset = m->storage_offset;
  t->dtype          = m->dtype;
}

static bool dtype_valid(wholememory_dtype_t d) { return d > WHOLEMEMORY_DT_UNKNOWN && d < WHOLEMEMORY_DT_COUNT; }

bool wholememory_convert_tensor_desc_to_array(wholememory_array_description_t* a,
                                              wholememory_tensor_description_t* t)
{
  if (!dtype_valid(t->dtype) || t->dim != 1 || t->strides[0] != 1) return false;
  a->size = t->sizes[0], a->storage_offset = t->storage_offset, a->dtype = t->dtype;
  return true;
}

bool wholememory_convert_tensor_desc_to_matrix(wholememory_matrix_description_t* m,
                                               wholememory_tensor_description_t* t)
{
  if (!dtype_valid(t->dtype) || t->dim <= 0 || t->dim > 2) return false;
  if (t->dim == 2 && t->strides[1] != 1) return false;
  m->dtype = t->dtype, m->storage_offset = t->storage_offset, m->sizes[0] = t->sizes[0];
  m->sizes[1] = t->dim == 2 ? t->sizes[1] : 1;
  m->stride   = t->dim == 2 ? t->strides[0] : 1;
  return true;
}

int64_t wholememory_get_memory_element_count_from_array(wholememory_array_description_t* a) { return a->size; }
int64_t wholememory_get_memory_size_from_array(wholememory_array_description_t* a)
{
  return a->size * static_cast<int64_t>(wholememory_dtype_get_element_size(a->dtype));
}
int64_t wholememory_get_memory_element_count_from_matrix(wholememory_matrix_description_t* m)
{
  return m->sizes[0] * m->stride;
}
int64_t wholememory_get_memory_size_from_matrix(wholememory_matrix_description_t* m)
{
  return m->sizes[0] * m->stride * static_cast<int64_t>(wholememory_dtype_get_element_size(m->dtype));
}
int64_t wholememory_get_memory_element_count_from_tensor(wholememory_tensor_description_t* t)
{
  if (t->dim == 0) return 1;
  if (t->dim < 0 || t->dim >= WHOLEMEMORY_MAX_TENSOR_DIM) return -1;
  return t->strides[0] * t->sizes[0];
}
int64_t wholememory_get_memory_size_from_tensor(wholememory_tensor_description_t* t)
{
  return wholememory_get_memory_element_count_from_tensor(t) *
         static_cast<int64_t>(wholememory_dtype_get_element_size(t->dtype));
}

bool wholememory_squeeze_tensor(wholememory_tensor_description_t* t, int dim)
{
  if (t == nullptr || dim < 0 || dim >= t->dim || t->sizes[dim] != 1) return false;
  if (dim != t->dim - 1 && t->strides[dim] != t->strides[dim + 1]) return false;
  for (int i = dim; i + 1 < t->dim; i++) {
    t->sizes[i]   = t->sizes[i + 1];
    t->strides[i] = t->strides[i + 1];
  }
  t->dim--;
  return true;
}

bool wholememory_unsqueeze_tensor(wholememory_tensor_description_t* t, int dim)
{
  if (t == nullptr || dim < 0 || dim > t->dim) return false;
  // the new unit dimension inherits the stride of the dimension it displaces (or of the last
  // dimension when appended) — reference tensor_description.cpp:221-234
  int64_t stride = t->dim > 0 ? t->strides[t->dim - 1] : 1;
  for (int i = t->dim; i > dim; i--) {
    t->sizes[i] = t->sizes[i - 1];
    stride = t->strides[i] = t->strides[i - 1];
  }
  t->sizes[dim]   = 1;
  t->strides[dim] = stride;
  t->dim++;
  return true;
}

}  // extern "C"
