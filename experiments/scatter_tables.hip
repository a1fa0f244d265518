// Experiment harness (not product code), round 3: the 512 B-row SCATTER (streamed input, random row writes) runs at 1.56 ms in
// one process and 1.85-1.90 ms in the next (profiles/r03_batch_kernel_cpp_ab.txt) whatever source buffer it reads — the level
// follows the TABLE. Here: three 51.2 GB tables in one process, the in-order single-batch scatter on each with different
// store policies, the gather on each for comparison; kernels carry the table number in their name for rocprofv3 --pmc.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/scatter_tables.hip -o experiments/scatter_tables
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define GAS __attribute__((address_space(1)))

__global__ void gen_idx(int64_t* idx, int64_t n, int64_t rows, uint64_t seed)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  idx[i] = (int64_t)(x % (uint64_t)rows);
}
__device__ __forceinline__ char* readlane_ptr(char* p, int lane)
{
  uint64_t v = (uint64_t)p;
  uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, lane), hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), lane);
  return (char*)(((uint64_t)hi << 32) | lo);
}

// POLICY of the random-side access: 0 nt, 1 plain, 2 sc1 (agent-scope write-through), 3 sc0 sc1 (system scope)
template <int POLICY>
__device__ __forceinline__ void store16(char* p, u32x4 v)
{
  if (POLICY == 0) __builtin_nontemporal_store(v, (GAS u32x4*)p);
  else if (POLICY == 1) *(GAS u32x4*)p = v;
  else if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// one 4 KiB tile (8 rows of 512 B) per wave, in order; SCATTER: plain -> table rows, else table rows -> plain
template <int TAB, bool SCATTER, int POLICY>
__global__ __launch_bounds__(64) void rows8(char* tab, const int64_t* idx, char* plain, int64_t n)
{
  const int lane = threadIdx.x & 63;
  const int64_t tile = blockIdx.x;
  const int col = lane & 31;
  const bool upper = lane >= 32;
  const int64_t e = tile * 8 + lane;
  char* my = nullptr;
  if (lane < 8 && e < n) my = tab + idx[e] * 512;
  char* pbase = plain + tile * 4096 + (upper ? 512 : 0);
  u32x4 d[4];
  char* dst[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    char* a = readlane_ptr(my, 2 * u);
    char* b = readlane_ptr(my, 2 * u + 1);
    char* t = upper ? b : a;
    char* q = pbase + u * 1024 + col * 16;
    dst[u] = t ? (SCATTER ? t + col * 16 : q) : nullptr;
    if (t) d[u] = __builtin_nontemporal_load((const GAS u32x4*)(SCATTER ? q : t + col * 16));
  }
#pragma unroll
  for (int u = 0; u < 4; u++)
    if (dst[u]) {
      if (SCATTER) store16<POLICY>(dst[u], d[u]);
      else __builtin_nontemporal_store(d[u], (GAS u32x4*)dst[u]);
    }
}

// every 512 B row of a window of the table written once, in a pseudo-random order private to each wave: random row writes
// with no streamed side at all (is it the write side alone?)
template <int TAB>
__global__ __launch_bounds__(64) void fill_rows_random(char* tab, const int64_t* idx, int64_t n)
{
  const int lane = threadIdx.x & 63;
  const int64_t tile = blockIdx.x;
  const int col = lane & 31;
  const bool upper = lane >= 32;
  const int64_t e = tile * 8 + lane;
  char* my = nullptr;
  if (lane < 8 && e < n) my = tab + idx[e] * 512;
  u32x4 v = {1u, 2u, 3u, 4u};
#pragma unroll
  for (int u = 0; u < 4; u++) {
    char* a = readlane_ptr(my, 2 * u);
    char* b = readlane_ptr(my, 2 * u + 1);
    char* t = upper ? b : a;
    if (t) __builtin_nontemporal_store(v, (GAS u32x4*)(t + col * 16));
  }
}

static hipEvent_t e0, e1;
template <typename F>
float timed(F launch, int iters)
{
  launch();
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; i++) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

template <int TAB>
void run_table(char* tab, const int64_t* idx, char* src, int64_t n, int iters)
{
  const dim3 g((unsigned)((n + 7) / 8)), b(64);
  printf("table %d at %p: gather %.4f", TAB, (void*)tab, timed([&] { hipLaunchKernelGGL((rows8<TAB, false, 0>), g, b, 0, 0, tab, idx, src, n); }, iters));
  printf("  scatter nt %.4f", timed([&] { hipLaunchKernelGGL((rows8<TAB, true, 0>), g, b, 0, 0, tab, idx, src, n); }, iters));
  printf("  plain %.4f", timed([&] { hipLaunchKernelGGL((rows8<TAB, true, 1>), g, b, 0, 0, tab, idx, src, n); }, iters));
  printf("  sc1 %.4f", timed([&] { hipLaunchKernelGGL((rows8<TAB, true, 2>), g, b, 0, 0, tab, idx, src, n); }, iters));
  printf("  sc0sc1 %.4f", timed([&] { hipLaunchKernelGGL((rows8<TAB, true, 3>), g, b, 0, 0, tab, idx, src, n); }, iters));
  printf("  random row fill %.4f ms\n", timed([&] { hipLaunchKernelGGL((fill_rows_random<TAB>), g, b, 0, 0, tab, idx, n); }, iters));
}

int main(int argc, char** argv)
{
  const int T        = argc > 1 ? atoi(argv[1]) : 3;
  const int iters    = argc > 2 ? atoi(argv[2]) : 6;
  const int64_t rows = 100000000ll, n = 10000000ll;
  std::vector<char*> tabs(T);
  for (auto& t : tabs) { CK(hipMalloc(&t, (size_t)rows * 512)); CK(hipMemsetAsync(t, 0, (size_t)rows * 512, 0)); }
  int64_t* idx; char* src;
  CK(hipMalloc(&idx, n * 8)); CK(hipMalloc(&src, (size_t)n * 512));
  hipLaunchKernelGGL(gen_idx, dim3((n + 255) / 256), dim3(256), 0, 0, idx, n, rows, 42ull);
  CK(hipMemsetAsync(src, 1, (size_t)n * 512, 0));
  CK(hipDeviceSynchronize());
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int round = 0; round < 2; round++) {
    if (T > 0) run_table<0>(tabs[0], idx, src, n, iters);
    if (T > 1) run_table<1>(tabs[1], idx, src, n, iters);
    if (T > 2) run_table<2>(tabs[2], idx, src, n, iters);
    if (T > 3) run_table<3>(tabs[3], idx, src, n, iters);
  }
  return 0;
}
