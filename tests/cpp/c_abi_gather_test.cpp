// A C++ caller of the C ABI, written the way the reference's gtests and bench call libwholegraph
// (cpp/tests/wholememory_ops/wholememory_gather_tests.cu:137-286, cpp/bench/wholememory_ops/gather_scatter_bench.cu):
// create communicator -> create embedding -> fill the closed-form table (embedding_test_utils.cu:197-238) through the
// local pointer -> wholememory_embedding_gather with the library's cached env functions -> exact compare on the host.
// Plain HIP runtime calls + <wholememory/*.h> only. Build + run: tests/test_cpp_abi_gpu.py.
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <wholememory/embedding.h>
#include <wholememory/wholememory_op.h>

#define REQUIRE(cond)                                                          \
  do {                                                                         \
    if (!(cond)) {                                                             \
      fprintf(stderr, "REQUIRE failed: %s (line %d)\n", #cond, __LINE__);      \
      return 1;                                                                \
    }                                                                          \
  } while (0)

static int run_case(wholememory_comm_t comm, wholememory_memory_type_t mt, wholememory_memory_location_t ml, int64_t rows,
                    int64_t dim, int64_t n_idx)
{
  wholememory_tensor_description_t desc;
  int64_t sizes[2] = {rows, dim};
  auto mdesc       = wholememory_create_matrix_desc(sizes, dim, 0, WHOLEMEMORY_DT_FLOAT);
  wholememory_copy_matrix_desc_to_tensor(&desc, &mdesc);
  wholememory_embedding_t emb;
  REQUIRE(wholememory_create_embedding(&emb, &desc, comm, mt, ml, nullptr) == WHOLEMEMORY_SUCCESS);
  wholememory_tensor_t table = wholememory_embedding_get_embedding_tensor(emb);
  auto* tdesc                = wholememory_tensor_get_tensor_description(table);
  const int64_t stride       = tdesc->strides[0];
  REQUIRE(stride % 4 == 0 && stride >= dim);

  wholememory_tensor_t local;
  REQUIRE(wholememory_tensor_map_local_tensor(table, &local) == WHOLEMEMORY_SUCCESS);
  float* local_ptr = static_cast<float*>(wholememory_tensor_get_data_pointer(local));
  std::vector<float> host(static_cast<size_t>(rows) * stride, -1.0f);
  for (int64_t r = 0; r < rows; r++)
    for (int64_t c = 0; c < dim; c++) host[r * stride + c] = static_cast<float>(r & 0xFFFFFF);
  REQUIRE(hipMemcpy(local_ptr, host.data(), host.size() * sizeof(float), hipMemcpyDefault) == hipSuccess);
  REQUIRE(wholememory_destroy_tensor(local) == WHOLEMEMORY_SUCCESS);

  std::vector<int64_t> idx(n_idx);
  uint64_t s = 88172645463325252ull;
  for (auto& v : idx) {
    s ^= s << 13, s ^= s >> 7, s ^= s << 17;
    v = static_cast<int64_t>(s % static_cast<uint64_t>(rows));
  }
  if (n_idx > 10) idx[3] = -1, idx[n_idx - 1] = -1;
  int64_t* d_idx;
  float* d_out;
  REQUIRE(hipMalloc(&d_idx, sizeof(int64_t) * (n_idx + 1)) == hipSuccess);
  REQUIRE(hipMalloc(&d_out, sizeof(float) * (n_idx * dim + 1)) == hipSuccess);
  REQUIRE(hipMemcpy(d_idx, idx.data(), sizeof(int64_t) * n_idx, hipMemcpyHostToDevice) == hipSuccess);
  REQUIRE(hipMemset(d_out, 0xFF, sizeof(float) * n_idx * dim) == hipSuccess);  // NaN pattern = "untouched"

  wholememory_tensor_t idx_t, out_t;
  auto idesc = wholememory_create_array_desc(n_idx, 0, WHOLEMEMORY_DT_INT64);
  wholememory_tensor_description_t it, ot;
  wholememory_copy_array_desc_to_tensor(&it, &idesc);
  int64_t osz[2] = {n_idx, dim};
  auto odesc     = wholememory_create_matrix_desc(osz, dim, 0, WHOLEMEMORY_DT_FLOAT);
  wholememory_copy_matrix_desc_to_tensor(&ot, &odesc);
  REQUIRE(wholememory_make_tensor_from_pointer(&idx_t, d_idx, &it) == WHOLEMEMORY_SUCCESS);
  REQUIRE(wholememory_make_tensor_from_pointer(&out_t, d_out, &ot) == WHOLEMEMORY_SUCCESS);

  hipStream_t stream;
  REQUIRE(hipStreamCreate(&stream) == hipSuccess);
  for (int rep = 0; rep < 3; rep++) {
    REQUIRE(wholememory_embedding_gather(emb, idx_t, out_t, false, wholememory_get_cached_env_func(),
                                         reinterpret_cast<int64_t>(stream)) == WHOLEMEMORY_SUCCESS);
  }
  REQUIRE(hipStreamSynchronize(stream) == hipSuccess);
  std::vector<float> out(static_cast<size_t>(n_idx) * dim);
  REQUIRE(hipMemcpy(out.data(), d_out, out.size() * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
  int64_t bad = 0;
  for (int64_t i = 0; i < n_idx; i++)
    for (int64_t c = 0; c < dim; c++) {
      const float got = out[i * dim + c];
      if (idx[i] < 0) {
        uint32_t bits;
        __builtin_memcpy(&bits, &got, 4);
        bad += bits != 0xFFFFFFFFu;  // skipped rows must stay untouched
      } else {
        bad += got != static_cast<float>(idx[i] & 0xFFFFFF);
      }
    }
  REQUIRE(bad == 0);
  REQUIRE(wholememory_destroy_tensor(idx_t) == WHOLEMEMORY_SUCCESS);
  REQUIRE(wholememory_destroy_tensor(out_t) == WHOLEMEMORY_SUCCESS);
  REQUIRE(wholememory_destroy_embedding(emb) == WHOLEMEMORY_SUCCESS);
  (void)hipFree(d_idx);
  (void)hipFree(d_out);
  (void)hipStreamDestroy(stream);
  return 0;
}

int main()
{
  REQUIRE(wholememory_init(0, LEVEL_WARN) == WHOLEMEMORY_SUCCESS);
  REQUIRE(hipSetDevice(0) == hipSuccess);
  wholememory_unique_id_t uid{};
  wholememory_comm_t comm;
  REQUIRE(wholememory_create_communicator(&comm, uid, 0, 1) == WHOLEMEMORY_SUCCESS);
  int rank = -1, size = -1;
  REQUIRE(wholememory_communicator_get_rank(&rank, comm) == WHOLEMEMORY_SUCCESS && rank == 0);
  REQUIRE(wholememory_communicator_get_size(&size, comm) == WHOLEMEMORY_SUCCESS && size == 1);
  const int64_t tensors_before = get_wholememory_tensor_count();
  const wholememory_memory_type_t types[] = {WHOLEMEMORY_MT_CONTINUOUS, WHOLEMEMORY_MT_CHUNKED, WHOLEMEMORY_MT_DISTRIBUTED};
  const wholememory_memory_location_t locs[] = {WHOLEMEMORY_ML_DEVICE, WHOLEMEMORY_ML_HOST};
  int failures = 0;
  for (auto mt : types)
    for (auto ml : locs) {
      failures += run_case(comm, mt, ml, 100003, 128, 50000);
      failures += run_case(comm, mt, ml, 20011, 11, 5000);  // padded rows (stride 12)
      failures += run_case(comm, mt, ml, 20011, 32, 0);     // indices_count == 0
    }
  REQUIRE(get_wholememory_tensor_count() == tensors_before);
  REQUIRE(wholememory_communicator_barrier(comm) == WHOLEMEMORY_SUCCESS);
  REQUIRE(wholememory_destroy_communicator(comm) == WHOLEMEMORY_SUCCESS);
  REQUIRE(wholememory_finalize() == WHOLEMEMORY_SUCCESS);
  if (failures) {
    fprintf(stderr, "%d case(s) failed\n", failures);
    return 1;
  }
  printf("C ABI GATHER OK\n");
  return 0;
}
