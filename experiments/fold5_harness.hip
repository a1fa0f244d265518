// Round 6, review item 4: the ordered fold of a very long run through a dense transposed copy (kernels/long_dense.cuh), alone:
// copy kernel and fold kernel timed separately for R = 64 / 128 / 192 rows per turn, bit-compared with a sequential CPU sum.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I wholegraph_amd/csrc/kernels experiments/fold5_harness.hip -o experiments/fold5_harness
//   experiments/fold5_harness [n_recv = 10000000] [hot rows = 527000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "long_dense.cuh"
using namespace wm::dense_fold;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__host__ __device__ inline float value_of(int64_t row, int col)
{
  uint64_t h = static_cast<uint64_t>(row) * 128u + static_cast<uint64_t>(col);
  h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
  return static_cast<float>(static_cast<uint32_t>(h & 0xFFFFFFu)) * (1.0f / 16777216.0f) - 0.5f;
}
__global__ void fill_kernel(float* g, int64_t rows, int dim)
{
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < rows * dim; i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    g[i] = value_of(i / dim, static_cast<int>(i % dim));
}
struct row_of_t {
  const float* grads; int64_t stride;
  __device__ __forceinline__ const float* operator()(int32_t o) const { return grads + static_cast<int64_t>(o) * stride; }
};
struct store_ep {
  float* out; int dim;
  __device__ __forceinline__ void operator()(const job& jb, int col, float acc) const { out[static_cast<int64_t>(jb.user) * dim + col] = acc; }
};

// a bandwidth hog for the "beside the tile kernel" runs: in-place update of a 5 GB buffer from another, `rounds` times
typedef float hf4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void hog_kernel(hf4* a, const hf4* b, int64_t n4, int rounds)
{
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int r = 0; r < rounds; r++)
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += 4 * stride) {
      hf4 x[4], y[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int64_t j = min(i + u * stride, n4 - 1);
        x[u] = __builtin_nontemporal_load(a + j), y[u] = __builtin_nontemporal_load(b + j);
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (i + u * stride < n4) __builtin_nontemporal_store(x[u] - 0.01f * y[u], a + i + u * stride);
    }
}
struct hog_t { hf4* a; const hf4* b; int64_t n4; hipStream_t stream; };

// hog != nullptr: the fold is launched first (its workgroups get their CUs), the hog right behind it on another stream
template <int R, int S>
float time_fold(const job* jobs, const int32_t* n_jobs, int n_host, int dim, const float* dense, float* out, int reps, const hog_t* hog = nullptr)
{
  const void* kfn = reinterpret_cast<const void*>(&fold_kernel<R, S, store_ep>);
  const int lds_bytes = static_cast<int>(shape<R, S>::kLdsBytes);
  CK(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipStream_t fs; int least, greatest; CK(hipDeviceGetStreamPriorityRange(&least, &greatest)); CK(hipStreamCreateWithPriority(&fs, hipStreamNonBlocking, greatest));
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, fs));
    hipLaunchKernelGGL((fold_kernel<R, S, store_ep>), dim3(n_host * slices_of(dim, S)), dim3(kBlock), lds_bytes, fs, jobs, n_jobs, dim, dense, store_ep{out, dim});
    CK(hipEventRecord(e1, fs));
    if (hog != nullptr) hipLaunchKernelGGL(hog_kernel, dim3(8192), dim3(256), 0, hog->stream, hog->a, hog->b, hog->n4, 2);
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  CK(hipDeviceSynchronize());
  return best;
}

int main(int argc, char** argv)
{
  const int64_t n_recv = argc > 1 ? atoll(argv[1]) : 10000000;
  const int hot        = argc > 2 ? atoi(argv[2]) : 527000;
  const int dim        = 128;
  float* grads; CK(hipMalloc(&grads, n_recv * dim * 4));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, grads, n_recv, dim);
  // runs: the hot one (every ~n_recv / hot-th position), then 100 k, 20 k, 4097, 300 and 257 rows at random positions
  std::vector<int> lens = {hot, 100000, 20000, 4097, 300, 257};
  std::vector<int32_t> order;
  std::vector<job> jobs;
  uint64_t s = 12345;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return static_cast<uint32_t>(s >> 33); };
  int64_t dense_cursor = 0;
  for (size_t j = 0; j < lens.size(); j++) {
    const int L = lens[j];
    std::vector<int32_t> pos(L);
    const double step = static_cast<double>(n_recv) / L;
    for (int i = 0; i < L; i++) pos[i] = static_cast<int32_t>(std::min<double>(n_recv - 1, i * step + (rnd() % 1000) * step / 1000.0));
    std::sort(pos.begin(), pos.end());
    job jb{}; jb.s0 = static_cast<int32_t>(order.size()); jb.rows = L; jb.dense_off = dense_cursor; jb.user = static_cast<int32_t>(j);
    dense_cursor += dense_floats(L, dim);
    order.insert(order.end(), pos.begin(), pos.end());
    jobs.push_back(jb);
  }
  int32_t* d_order; CK(hipMalloc(&d_order, order.size() * 4)); CK(hipMemcpy(d_order, order.data(), order.size() * 4, hipMemcpyHostToDevice));
  job* d_jobs; CK(hipMalloc(&d_jobs, jobs.size() * sizeof(job))); CK(hipMemcpy(d_jobs, jobs.data(), jobs.size() * sizeof(job), hipMemcpyHostToDevice));
  int32_t n_jobs = static_cast<int32_t>(jobs.size()), *d_n; CK(hipMalloc(&d_n, 4)); CK(hipMemcpy(d_n, &n_jobs, 4, hipMemcpyHostToDevice));
  float* dense; CK(hipMalloc(&dense, dense_cursor * 4));
  float* out; CK(hipMalloc(&out, jobs.size() * dim * 4));
  CK(hipDeviceSynchronize());
  // CPU reference: sequential sums in receive order
  std::vector<float> ref(jobs.size() * dim);
  for (size_t j = 0; j < jobs.size(); j++)
    for (int c = 0; c < dim; c++) {
      float acc = value_of(order[jobs[j].s0], c);
      for (int i = 1; i < jobs[j].rows; i++) acc += value_of(order[jobs[j].s0 + i], c);
      ref[j * dim + c] = acc;
    }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int grid : {256, 1024, 4096}) {
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((copy_kernel<row_of_t>), dim3(grid), dim3(256), 0, 0, d_jobs, d_n, d_order, row_of_t{grads, dim}, dim, dense);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    int64_t rows_total = 0; for (int L : lens) rows_total += L;
    printf("copy_kernel grid %5d: %.4f ms  (%.1f MB in + out -> %.0f GB/s)\n", grid, best, rows_total * 1024.0 / 1e6, rows_total * 1024.0 / best / 1e6);
  }
  auto check = [&](const char* name, float ms) {
    std::vector<float> got(jobs.size() * dim);
    CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
    const bool ok = memcmp(got.data(), ref.data(), got.size() * 4) == 0;
    int bad = 0; for (size_t i = 0; i < got.size(); i++) bad += memcmp(&got[i], &ref[i], 4) != 0;
    printf("%-18s %.4f ms = %.2f cycles per row of the hot run at 2.4 GHz   bit-exact vs CPU: %s (%d of %zu differ)\n", name, ms, ms * 1e-3 * 2.4e9 / lens[0], ok ? "yes" : "NO", bad, got.size());
    CK(hipMemset(out, 0, got.size() * 4));
  };
  const int nj = static_cast<int>(jobs.size());
  check("fold R 128 S 32", time_fold<128, 32>(d_jobs, d_n, nj, dim, dense, out, 5));
  check("fold R 128 S 16", time_fold<128, 16>(d_jobs, d_n, nj, dim, dense, out, 5));
  check("fold R 128 S 8", time_fold<128, 8>(d_jobs, d_n, nj, dim, dense, out, 5));
  // beside a kernel that saturates the memory system (2 x 5 GB read + 5 GB written per round, ~5.5 ms for two rounds)
  hf4 *ha, *hb; const int64_t hn4 = 10000000ll * 32;
  CK(hipMalloc(&ha, hn4 * 16)); CK(hipMemset(ha, 0, hn4 * 16));
  hb = reinterpret_cast<hf4*>(grads);
  hipStream_t hs; CK(hipStreamCreateWithFlags(&hs, hipStreamNonBlocking));
  hog_t hog{ha, hb, hn4, hs};
  check("loaded: R 128 S 32", time_fold<128, 32>(d_jobs, d_n, nj, dim, dense, out, 4, &hog));
  check("loaded: R 128 S 16", time_fold<128, 16>(d_jobs, d_n, nj, dim, dense, out, 4, &hog));
  check("loaded: R 128 S 8", time_fold<128, 8>(d_jobs, d_n, nj, dim, dense, out, 4, &hog));
  return 0;
}
