/*
 * wm_oracle.c — CPU restatement of the reference WholeMemory embedding gather / scatter /
 * gradient-apply path (rapidsai/wholegraph 24.12, paths relative to /root/reference/cpp).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE. Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the shipped library (wholegraph_amd/csrc) never links
 * or calls anything in this directory.
 *
 * Parity status: the reference hot path cannot be compiled or imported in this image (CUDA, NCCL,
 * raft), so this restatement is pinned against
 *   (1) the closed-form tables + exact-compare rule of the reference's own gather/scatter tests
 *       (tests/wholememory_ops/embedding_test_utils.cu:197-238,401-431,467-520 and
 *        python/.../tests/wholegraph_torch/ops/test_wholegraph_gather_scatter.py:26-37),
 *   (2) the reference tests' host CPUOptimizer + first-seen dedup
 *       (tests/wholememory_ops/wholememory_embedding_gradient_apply_tests.cu:169-371,437-466),
 *       which tests/test_oracle_pinning.py re-derives independently in numpy,
 *   (3) the one reference TU that does build here, src/wholememory/tensor_description.cpp,
 *       compiled in place into oracle/_ref/ and compared call-for-call with the product library.
 * Every function cites the reference lines it follows.
 *
 * dtype codes are the values of wholememory_dtype_t (include/wholememory/tensor_description.h:29-40).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#ifdef _OPENMP
#include <omp.h>
#endif

enum {
  DT_UNKNOWN = 0,
  DT_FLOAT   = 1,
  DT_HALF    = 2,
  DT_DOUBLE  = 3,
  DT_BF16    = 4,
  DT_INT     = 5,
  DT_INT64   = 6,
  DT_INT16   = 7,
  DT_INT8    = 8,
};

/* src/wholememory/tensor_description.cpp:25-39 */
size_t wmo_dtype_size(int dtype)
{
  switch (dtype) {
    case DT_INT8: return 1;
    case DT_INT16:
    case DT_BF16:
    case DT_HALF: return 2;
    case DT_INT:
    case DT_FLOAT: return 4;
    case DT_INT64:
    case DT_DOUBLE: return 8;
    case DT_UNKNOWN: return 0;
    default: return (size_t)-1;
  }
}

static int is_float_dtype(int d) { return d == DT_FLOAT || d == DT_HALF || d == DT_DOUBLE || d == DT_BF16; }
static int is_int_dtype(int d) { return d == DT_INT || d == DT_INT64 || d == DT_INT16 || d == DT_INT8; }

/* ---------------- IEEE binary16 / bfloat16 <-> binary32, round-to-nearest-even ----------------
 * The reference converts through float with CUDA's static_cast<__half>(float) / static_cast<float>
 * (__half) (functions/gather_scatter_func.cuh:175-208), i.e. RN-even. */
float wmo_half_to_float(uint16_t h)
{
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp  = (h >> 10) & 0x1f;
  uint32_t man  = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else { /* subnormal: normalise */
      int e = -1;
      do {
        e++;
        man <<= 1;
      } while ((man & 0x400u) == 0);
      man &= 0x3ffu;
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

uint16_t wmo_float_to_half(float f)
{
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t absx = x & 0x7fffffffu;
  if (absx >= 0x7f800000u) { /* inf / nan */
    if (absx > 0x7f800000u) return (uint16_t)(sign | 0x7e00u | ((absx >> 13) & 0x3ffu) | 0x200u);
    return (uint16_t)(sign | 0x7c00u);
  }
  if (absx >= 0x477ff000u) { /* >= 65520 rounds to inf */
    return (uint16_t)(sign | 0x7c00u);
  }
  if (absx < 0x33000001u) { /* < 2^-25 (or == 2^-25 tie to even 0) */
    return (uint16_t)sign;
  }
  int32_t exp  = (int32_t)(absx >> 23) - 127;
  uint32_t man = (absx & 0x7fffffu) | 0x800000u;
  uint32_t shift;
  uint32_t hexp;
  if (exp < -14) { /* subnormal half */
    shift = (uint32_t)(13 + (-14 - exp));
    hexp  = 0;
  } else {
    shift = 13;
    hexp  = (uint32_t)(exp + 15);
  }
  uint32_t halfman = man >> shift;
  uint32_t rem     = man & ((1u << shift) - 1u);
  uint32_t halfway = 1u << (shift - 1);
  if (rem > halfway || (rem == halfway && (halfman & 1u))) halfman++;
  uint32_t out;
  if (hexp == 0) {
    out = halfman; /* may carry into exponent 1: correct */
  } else {
    out = ((hexp - 1) << 10) + halfman; /* halfman includes the implicit 1 at bit 10 */
  }
  return (uint16_t)(sign | out);
}

float wmo_bf16_to_float(uint16_t b)
{
  uint32_t bits = (uint32_t)b << 16;
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

uint16_t wmo_float_to_bf16(float f)
{
  uint32_t x;
  memcpy(&x, &f, 4);
  if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40u); /* quiet nan */
  uint32_t lsb = (x >> 16) & 1u;
  x += 0x7fffu + lsb;
  return (uint16_t)(x >> 16);
}

/* Element load/convert/store with the reference's conversion chain
 * convert_type<From,To> = To::convert_store_data(From::convert_load_data(x))
 * (functions/gather_scatter_func.cuh:161-208): half/bf16 load as float and are stored from float,
 * so double->half is double->float->half; integers are plain static_casts. */
static inline double load_fp(int dtype, const void* p, int64_t i)
{
  switch (dtype) {
    case DT_FLOAT: return (double)((const float*)p)[i];
    case DT_DOUBLE: return ((const double*)p)[i];
    case DT_HALF: return (double)wmo_half_to_float(((const uint16_t*)p)[i]);
    default: return (double)wmo_bf16_to_float(((const uint16_t*)p)[i]);
  }
}
static inline void store_fp(int dtype, void* p, int64_t i, double v)
{
  switch (dtype) {
    case DT_FLOAT: ((float*)p)[i] = (float)v; break;
    case DT_DOUBLE: ((double*)p)[i] = v; break;
    case DT_HALF: ((uint16_t*)p)[i] = wmo_float_to_half((float)v); break;
    default: ((uint16_t*)p)[i] = wmo_float_to_bf16((float)v); break;
  }
}
static inline int64_t load_int(int dtype, const void* p, int64_t i)
{
  switch (dtype) {
    case DT_INT8: return ((const int8_t*)p)[i];
    case DT_INT16: return ((const int16_t*)p)[i];
    case DT_INT: return ((const int32_t*)p)[i];
    default: return ((const int64_t*)p)[i];
  }
}
static inline void store_int(int dtype, void* p, int64_t i, int64_t v)
{
  switch (dtype) {
    case DT_INT8: ((int8_t*)p)[i] = (int8_t)v; break;
    case DT_INT16: ((int16_t*)p)[i] = (int16_t)v; break;
    case DT_INT: ((int32_t*)p)[i] = (int32_t)v; break;
    default: ((int64_t*)p)[i] = v; break;
  }
}

static inline void convert_row(int from_dtype, const void* from, int64_t from_off, int to_dtype, void* to,
                               int64_t to_off, int64_t count)
{
  if (from_dtype == to_dtype) {
    size_t es = wmo_dtype_size(from_dtype);
    memcpy((char*)to + to_off * es, (const char*)from + from_off * es, (size_t)count * es);
    return;
  }
  if (is_float_dtype(from_dtype)) {
    for (int64_t c = 0; c < count; c++) store_fp(to_dtype, to, to_off + c, load_fp(from_dtype, from, from_off + c));
  } else {
    for (int64_t c = 0; c < count; c++) store_int(to_dtype, to, to_off + c, load_int(from_dtype, from, from_off + c));
  }
}

static inline int64_t load_index(int idx_dtype, const void* indices, int64_t i)
{
  return idx_dtype == DT_INT ? (int64_t)((const int32_t*)indices)[i] : ((const int64_t*)indices)[i];
}

/* ------------------------------- layout / partition plan -------------------------------------- */

/* src/wholememory/embedding.cpp:43-50 — row stride padded to a 16-byte multiple (in elements) */
int64_t wmo_align_embedding_dim(int64_t dim, int64_t element_size)
{
  int64_t align_count = 16 / element_size;
  return dim % align_count == 0 ? dim : (dim / align_count + 1) * align_count;
}

/* src/wholememory/embedding.cpp:467-484 — padded row count under round-robin sharding */
int64_t wmo_round_robin_total_entries(int64_t total_entry_count, int world_size, int round_robin_size)
{
  if (round_robin_size == 0) return total_entry_count;
  int first_rank_extra_entry = (int)(total_entry_count % ((int64_t)world_size * round_robin_size));
  if (first_rank_extra_entry > round_robin_size) first_rank_extra_entry = round_robin_size;
  int64_t first_rank_entry_size =
    total_entry_count / ((int64_t)world_size * round_robin_size) * round_robin_size;
  first_rank_entry_size += first_rank_extra_entry;
  return first_rank_entry_size * world_size;
}

/* src/wholememory/memory_handle.cpp:1618-1635 + :2122-2128 — equal plan: ceil(N/W) entries per
 * rank, clipped at N (trailing ranks may be empty). offsets has world_size + 1 entries. */
void wmo_equal_partition(uint64_t total_entries, int world_size, uint64_t* sizes, uint64_t* offsets)
{
  uint64_t per_rank = (total_entries + (uint64_t)world_size - 1) / (uint64_t)world_size;
  for (int i = 0; i < world_size; i++) {
    uint64_t s = (uint64_t)i * per_rank;
    uint64_t e = (uint64_t)(i + 1) * per_rank;
    if (s > total_entries) s = total_entries;
    if (e > total_entries) e = total_entries;
    sizes[i]   = e - s;
    offsets[i] = s;
  }
  offsets[world_size] = total_entries;
}

/* src/wholememory/memory_handle.cpp:69-79 (prefix sums of the user partition) and :1607-1616
 * (same_chunk: sizes[0..W-2] all equal — the loop bound `i < W-2` compares pairs (i, i+1)). */
void wmo_custom_partition(const uint64_t* entries, int world_size, uint64_t* offsets, int* same_chunk)
{
  offsets[0] = 0;
  for (int i = 0; i < world_size; i++) offsets[i + 1] = offsets[i] + entries[i];
  int same = 1;
  for (int i = 0; i < world_size - 2; i++) {
    if (entries[i] != entries[i + 1]) {
      same = 0;
      break;
    }
  }
  *same_chunk = same;
}

/* Owner of a table row. "the r with offsets[r] <= idx < offsets[r+1]"
 * (functions/bucket_ids_func.cu:31-49; include/wholememory/device_reference.cuh:41-61). Empty
 * ranks (offsets[r] == offsets[r+1]) never own anything. */
static inline int owner_rank(int64_t idx, const uint64_t* entry_offsets, int world_size)
{
  for (int r = 0; r < world_size; r++) {
    if ((uint64_t)idx < entry_offsets[r + 1]) return r;
  }
  return world_size - 1;
}

/* ------------------------------------ gather / scatter --------------------------------------- */

/*
 * functions/gather_scatter_func.cuh:253-316 (gather_func_kernel) with the global reference of
 * include/wholememory/device_reference.cuh:41-61 restated per row:
 *   out[storage_off_out + i*out_stride + c] = cast(table[storage_off + idx[i]*stride + c]),
 *   c in [0, dim); rows with idx[i] < 0 are skipped (output row left untouched, :296).
 * The table is given as per-rank shard base pointers + row offsets (world_size == 1 with
 * entry_offsets {0, N} is the CONTINUOUS / plain-pointer case). raw_indices != NULL (int64) reproduces the
 * raw_output_idx indirection of gather_with_sorted_ids (:287-288): row i is written to output row
 * raw_indices[i] (the reference types that array like the indices; here it is always int64).
 * Returns 0, or -1 on an unsupported dtype pair (functions/gather_func.cu:79-81).
 */
int wmo_gather(const void* const* shard_ptrs, const uint64_t* entry_offsets, int world_size, int table_dtype,
               int64_t dim, int64_t stride, int64_t storage_offset, const void* indices, int idx_dtype,
               int64_t n, const void* raw_indices, void* out, int out_dtype, int64_t out_stride,
               int64_t out_storage_offset)
{
  if (!((is_float_dtype(table_dtype) && is_float_dtype(out_dtype)) ||
        (is_int_dtype(table_dtype) && is_int_dtype(out_dtype))))
    return -1;
  if (idx_dtype != DT_INT && idx_dtype != DT_INT64) return -1;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    int64_t idx = load_index(idx_dtype, indices, i);
    if (idx < 0) continue;
    int64_t out_row = raw_indices ? ((const int64_t*)raw_indices)[i] : i;
    int r           = owner_rank(idx, entry_offsets, world_size);
    int64_t local   = idx - (int64_t)entry_offsets[r];
    convert_row(table_dtype, shard_ptrs[r], storage_offset + local * stride, out_dtype, out,
                out_storage_offset + out_row * out_stride, dim);
  }
  return 0;
}

/*
 * functions/gather_scatter_func.cuh:519-598 (scatter_func_kernel):
 *   table[storage_off + idx[i]*stride + c] = cast(in[in_storage_off + i*in_stride + c]);
 * idx[i] < 0 skipped (:576). Plain overwrite: for duplicate ids the reference leaves the winner
 * unordered; this restatement is sequential, so the LAST occurrence wins — tests that use
 * duplicates must make equal ids carry equal rows (as the reference tests do).
 */
int wmo_scatter_mapped(const void* in, int in_dtype, int64_t in_stride, int64_t in_storage_offset, const void* indices,
                       int idx_dtype, int64_t n, const int64_t* raw_indices, void* const* shard_ptrs,
                       const uint64_t* entry_offsets, int world_size, int table_dtype, int64_t dim, int64_t stride,
                       int64_t storage_offset)
{
  if (!((is_float_dtype(table_dtype) && is_float_dtype(in_dtype)) ||
        (is_int_dtype(table_dtype) && is_int_dtype(in_dtype))))
    return -1;
  if (idx_dtype != DT_INT && idx_dtype != DT_INT64) return -1;
  for (int64_t i = 0; i < n; i++) {
    int64_t idx = load_index(idx_dtype, indices, i);
    if (idx < 0) continue;
    int r         = owner_rank(idx, entry_offsets, world_size);
    int64_t local = idx - (int64_t)entry_offsets[r];
    int64_t in_row = raw_indices ? raw_indices[i] : i; /* input row feeding entry i */
    convert_row(in_dtype, in, in_storage_offset + in_row * in_stride, table_dtype, shard_ptrs[r],
                storage_offset + local * stride, dim);
  }
  return 0;
}

int wmo_scatter(const void* in, int in_dtype, int64_t in_stride, int64_t in_storage_offset, const void* indices,
                int idx_dtype, int64_t n, void* const* shard_ptrs, const uint64_t* entry_offsets, int world_size,
                int table_dtype, int64_t dim, int64_t stride, int64_t storage_offset)
{
  return wmo_scatter_mapped(in, in_dtype, in_stride, in_storage_offset, indices, idx_dtype, n, 0, shard_ptrs,
                            entry_offsets, world_size, table_dtype, dim, stride, storage_offset);
}

/* ------------------------------ index bucketing / exchange ------------------------------------ */

/* functions/bucket_ids_func.cu:51-87: count[r] = #{i : idx[i] >= 0, offsets[r] <= idx[i] < offsets[r+1]} */
void wmo_bucket_counts(const void* indices, int idx_dtype, int64_t n, const uint64_t* entry_offsets,
                       int world_size, int64_t* counts)
{
  for (int r = 0; r < world_size; r++) counts[r] = 0;
  for (int64_t i = 0; i < n; i++) {
    int64_t idx = load_index(idx_dtype, indices, i);
    if (idx < 0) continue;
    counts[owner_rank(idx, entry_offsets, world_size)]++;
  }
}

/* stable LSD radix sort of (u64 key, i64 payload) — the order cub::DeviceRadixSort::SortPairs
 * produces: ascending key, ties in input order. */
static void radix_sort_pairs_u64(uint64_t* keys, int64_t* vals, int64_t n)
{
  if (n <= 1) return;
  uint64_t* k2 = (uint64_t*)malloc((size_t)n * 8);
  int64_t* v2  = (int64_t*)malloc((size_t)n * 8);
  uint64_t *ka = keys, *kb = k2;
  int64_t *va = vals, *vb = v2;
  for (int pass = 0; pass < 8; pass++) {
    int shift = pass * 8;
    size_t hist[257];
    memset(hist, 0, sizeof(hist));
    for (int64_t i = 0; i < n; i++) hist[((ka[i] >> shift) & 0xff) + 1]++;
    if (hist[((ka[0] >> shift) & 0xff) + 1] == (size_t)n) continue; /* all same digit: skip */
    for (int d = 0; d < 256; d++) hist[d + 1] += hist[d];
    for (int64_t i = 0; i < n; i++) {
      size_t pos = hist[(ka[i] >> shift) & 0xff]++;
      kb[pos]    = ka[i];
      vb[pos]    = va[i];
    }
    uint64_t* tk = ka;
    ka           = kb;
    kb           = tk;
    int64_t* tv = va;
    va          = vb;
    vb          = tv;
  }
  if (ka != keys) {
    memcpy(keys, ka, (size_t)n * 8);
    memcpy(vals, va, (size_t)n * 8);
  }
  free(k2);
  free(v2);
}

/*
 * functions/exchange_ids_nccl_func.cu:42-92 (exchange_ids_temp_func, sort part):
 * keys = indices reinterpreted as UNSIGNED of the same width (negatives sort last, :66-69),
 * payload = 0..n-1 as int64 (:60-65), full-width stable radix sort.
 * sorted_out has the index dtype; raw_out is always int64.
 */
void wmo_sort_ids(const void* indices, int idx_dtype, int64_t n, void* sorted_out, int64_t* raw_out)
{
  uint64_t* keys = (uint64_t*)malloc((size_t)(n > 0 ? n : 1) * 8);
  for (int64_t i = 0; i < n; i++) {
    keys[i]    = idx_dtype == DT_INT ? (uint64_t)(uint32_t)((const int32_t*)indices)[i]
                                     : (uint64_t)((const int64_t*)indices)[i];
    raw_out[i] = i;
  }
  radix_sort_pairs_u64(keys, raw_out, n);
  for (int64_t i = 0; i < n; i++) {
    if (idx_dtype == DT_INT)
      ((int32_t*)sorted_out)[i] = (int32_t)(uint32_t)keys[i];
    else
      ((int64_t*)sorted_out)[i] = (int64_t)keys[i];
  }
  free(keys);
}

/* functions/map_indices_func.cu:26-45 (storage_idx2wm_emb_idx_kernel), quirk included: the rank
 * that owns the index is computed but unused (:39) and `entry_start` is the CALLER's local first
 * row (:88-90):
 *   t = idx / rr; off = idx % rr; wm = entry_start + rr * (t / W) + off               */
/* rank_rows > 0 (NOT the reference): the owner's first row replaces entry_start — owner = t % W holds rows
 * [owner * rank_rows, (owner + 1) * rank_rows), which is where the round-robin file loader (file_io.cpp:145-190 of the
 * product; reference file_io.cpp round_robin path) puts entry ((l / rr) * W + r) * rr + l % rr of rank r. The reference
 * statement is right only for ids the caller owns; the product maps to the owner (deviations). */
void wmo_round_robin_map_ex(const void* indices, int idx_dtype, int64_t n, int64_t entry_start, int world_size,
                            int round_robin_size, int64_t rank_rows, void* mapped)
{
  for (int64_t i = 0; i < n; i++) {
    int64_t idx = load_index(idx_dtype, indices, i);
    int64_t wm  = idx;
    if (idx >= 0) {
      int64_t t    = idx / round_robin_size;
      int64_t off  = idx % round_robin_size;
      int64_t base = rank_rows > 0 ? (t % world_size) * rank_rows : entry_start;
      wm           = base + (int64_t)round_robin_size * (t / world_size) + off;
    }
    if (idx_dtype == DT_INT)
      ((int32_t*)mapped)[i] = (int32_t)wm;
    else
      ((int64_t*)mapped)[i] = wm;
  }
}

void wmo_round_robin_map(const void* indices, int idx_dtype, int64_t n, int64_t entry_start, int world_size,
                         int round_robin_size, void* mapped)
{
  for (int64_t i = 0; i < n; i++) {
    int64_t idx = load_index(idx_dtype, indices, i);
    int64_t t   = idx / round_robin_size;
    int64_t off = idx % round_robin_size;
    int64_t wm  = entry_start + (int64_t)round_robin_size * (t / world_size) + off;
    if (idx_dtype == DT_INT)
      ((int32_t*)mapped)[i] = (int32_t)wm;
    else
      ((int64_t*)mapped)[i] = wm;
  }
}

/* ------------------------------- gradient dedup + optimizers ---------------------------------- */

/*
 * functions/exchange_embeddings_nccl_func.cu:76-174 (dedup_indice_and_gradients):
 * stable radix sort of the received ids as SIGNED keys with payload = position (int), then
 * unique_by_key, then for each unique id the fp32 rows of its occurrences are summed
 * SEQUENTIALLY in sorted (= receive-buffer) order: first row copied, the rest `+=` (:92-101).
 * Returns the number of unique ids; unique_ids has the index dtype, dedup_grads is [count, dim]
 * with row stride = dim.
 */
int64_t wmo_dedup_grads(const void* indices, int idx_dtype, int64_t n, const float* grads, int64_t dim,
                        int64_t grad_stride, void* unique_ids, float* dedup_grads)
{
  if (n == 0) return 0;
  uint64_t* keys = (uint64_t*)malloc((size_t)n * 8);
  int64_t* pos   = (int64_t*)malloc((size_t)n * 8);
  for (int64_t i = 0; i < n; i++) {
    int64_t v = load_index(idx_dtype, indices, i);
    keys[i]   = (uint64_t)v ^ 0x8000000000000000ull; /* signed order as unsigned */
    pos[i]    = i;
  }
  radix_sort_pairs_u64(keys, pos, n);
  int64_t count = 0;
  for (int64_t i = 0; i < n;) {
    int64_t j = i;
    float* dst = dedup_grads + count * dim;
    while (j < n && keys[j] == keys[i]) {
      const float* src = grads + pos[j] * grad_stride;
      if (j == i) {
        for (int64_t d = 0; d < dim; d++) dst[d] = src[d];
      } else {
        for (int64_t d = 0; d < dim; d++) dst[d] += src[d];
      }
      j++;
    }
    int64_t id = (int64_t)(keys[i] ^ 0x8000000000000000ull);
    if (idx_dtype == DT_INT)
      ((int32_t*)unique_ids)[count] = (int32_t)id;
    else
      ((int64_t*)unique_ids)[count] = id;
    count++;
    i = j;
  }
  free(keys);
  free(pos);
  return count;
}

/* Optimizer steps on the local shard. `local_table` points at this rank's first row; ids are
 * GLOBAL row ids and local = id - local_entry_offset (functions/embedding_optimizer_func.cu:195).
 * Arithmetic order follows the kernels and the reference tests' host CPUOptimizer
 * (tests/wholememory_ops/wholememory_embedding_gradient_apply_tests.cu:213-292) statement for
 * statement; compiled with -ffp-contract=off so every * and + rounds separately. */

/* functions/embedding_optimizer_func.cu:212-223; CPUOptimizer::ApplySGD (:280-292) */
void wmo_sgd_step(const void* ids, int idx_dtype, int64_t count, const float* grads, int64_t grad_stride,
                  float* local_table, int64_t table_stride, int64_t local_entry_offset, int64_t dim,
                  float weight_decay, float lr)
{
  for (int64_t i = 0; i < count; i++) {
    int64_t local  = load_index(idx_dtype, ids, i) - local_entry_offset;
    const float* g = grads + i * grad_stride;
    float* e       = local_table + local * table_stride;
    for (int64_t d = 0; d < dim; d++) {
      float grad_value      = g[d];
      float embedding_value = e[d];
      grad_value += weight_decay * embedding_value;
      embedding_value -= lr * grad_value;
      e[d] = embedding_value;
    }
  }
}

/* functions/embedding_optimizer_func.cu:331-421; CPUOptimizer::ApplyLazyAdam (:213-245).
 * state layout: per_element = [m(0..stride) | v(0..stride)] per row, row stride 2*table_stride
 * (embedding_optimizer_func.cu:382-383); per_row = [beta1^t, beta2^t] per row (:360). */
void wmo_lazy_adam_step(const void* ids, int idx_dtype, int64_t count, const float* grads, int64_t grad_stride,
                        float* local_table, float* per_element, float* per_row, int64_t table_stride,
                        int64_t local_entry_offset, int64_t dim, float weight_decay, float epsilon, float beta1,
                        float beta2, int adam_w, float lr)
{
  for (int64_t i = 0; i < count; i++) {
    int64_t local  = load_index(idx_dtype, ids, i) - local_entry_offset;
    const float* g = grads + i * grad_stride;
    float* e       = local_table + local * table_stride;
    float* m_ptr   = per_element + local * table_stride * 2;
    float* v_ptr   = m_ptr + table_stride;
    float beta1t   = per_row[local * 2 + 0];
    float beta2t   = per_row[local * 2 + 1];
    beta1t *= beta1;
    beta2t *= beta2;
    for (int64_t d = 0; d < dim; d++) {
      float grad_value      = g[d];
      float embedding_value = e[d];
      if (adam_w) {
        embedding_value -= lr * weight_decay * embedding_value;
      } else {
        grad_value = grad_value + weight_decay * embedding_value;
      }
      float m         = m_ptr[d];
      float v         = v_ptr[d];
      m               = beta1 * m + (1 - beta1) * grad_value;
      v               = beta2 * v + (1 - beta2) * grad_value * grad_value;
      float mhat      = m / (1 - beta1t);
      float vhat      = v / (1 - beta2t);
      embedding_value = embedding_value - lr * mhat / (sqrtf(vhat) + epsilon);
      m_ptr[d]        = m;
      v_ptr[d]        = v;
      e[d]            = embedding_value;
    }
    per_row[local * 2 + 0] = beta1t;
    per_row[local * 2 + 1] = beta2t;
  }
}

/* functions/embedding_optimizer_func.cu:594-658; CPUOptimizer::ApplyAdaGrad (:246-262) */
void wmo_adagrad_step(const void* ids, int idx_dtype, int64_t count, const float* grads, int64_t grad_stride,
                      float* local_table, float* state_sum_tbl, int64_t table_stride, int64_t local_entry_offset,
                      int64_t dim, float weight_decay, float epsilon, float lr)
{
  for (int64_t i = 0; i < count; i++) {
    int64_t local  = load_index(idx_dtype, ids, i) - local_entry_offset;
    const float* g = grads + i * grad_stride;
    float* e       = local_table + local * table_stride;
    float* s       = state_sum_tbl + local * table_stride;
    for (int64_t d = 0; d < dim; d++) {
      float grad_value      = g[d];
      float embedding_value = e[d];
      grad_value            = grad_value + weight_decay * embedding_value;
      float state_sum       = s[d];
      state_sum             = state_sum + grad_value * grad_value;
      embedding_value       = embedding_value - lr * grad_value / (sqrtf(state_sum) + epsilon);
      s[d]                  = state_sum;
      e[d]                  = embedding_value;
    }
  }
}

/* functions/embedding_optimizer_func.cu:791-856; CPUOptimizer::ApplyRMSProp (:263-279) */
void wmo_rmsprop_step(const void* ids, int idx_dtype, int64_t count, const float* grads, int64_t grad_stride,
                      float* local_table, float* v_tbl, int64_t table_stride, int64_t local_entry_offset,
                      int64_t dim, float weight_decay, float epsilon, float alpha, float lr)
{
  for (int64_t i = 0; i < count; i++) {
    int64_t local  = load_index(idx_dtype, ids, i) - local_entry_offset;
    const float* g = grads + i * grad_stride;
    float* e       = local_table + local * table_stride;
    float* vp      = v_tbl + local * table_stride;
    for (int64_t d = 0; d < dim; d++) {
      float grad_value      = g[d];
      float embedding_value = e[d];
      grad_value            = grad_value + weight_decay * embedding_value;
      float v               = vp[d];
      v                     = alpha * v + (1 - alpha) * grad_value * grad_value;
      embedding_value       = embedding_value - lr * grad_value / (sqrtf(v) + epsilon);
      vp[d]                 = v;
      e[d]                  = embedding_value;
    }
  }
}

/* ------------------------------------ test-table closed forms --------------------------------- */

/* tests/wholememory_ops/embedding_test_utils.cu:197-238: value(row r, any col) =
 * T(r & (2^(M+1) - 1)), M = mantissa bits of T (float 23, half 10, double 52, bf16 7); integer
 * tables: plain cast of r. Fills rows [row_start, row_start + rows). */
void wmo_fill_closed_form(void* table, int dtype, int64_t row_start, int64_t rows, int64_t dim, int64_t stride)
{
  for (int64_t r = 0; r < rows; r++) {
    int64_t g = row_start + r;
    for (int64_t c = 0; c < dim; c++) {
      int64_t o = r * stride + c;
      switch (dtype) {
        case DT_FLOAT: ((float*)table)[o] = (float)(g & ((1ll << 24) - 1)); break;
        case DT_HALF: ((uint16_t*)table)[o] = wmo_float_to_half((float)(g & ((1ll << 11) - 1))); break;
        case DT_BF16: ((uint16_t*)table)[o] = wmo_float_to_bf16((float)(g & ((1ll << 8) - 1))); break;
        case DT_DOUBLE: ((double*)table)[o] = (double)(g & ((1ll << 53) - 1)); break;
        case DT_INT8: ((int8_t*)table)[o] = (int8_t)g; break;
        case DT_INT16: ((int16_t*)table)[o] = (int16_t)g; break;
        case DT_INT: ((int32_t*)table)[o] = (int32_t)g; break;
        default: ((int64_t*)table)[o] = g; break;
      }
    }
  }
}

int wmo_num_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void wmo_set_num_threads(int n)
{
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}
