// wholegraph_amd — placement probe (round 3). The speed of random 512-byte row accesses to a large device allocation depends
// on WHERE the allocation sits in HBM (profiles/r03_tables_in_one_process.txt: four 51 GB tables alive in one process scatter
// at 1.57 ... 1.92 ms per 10 M rows, reproducibly per table; the slow tables show 3-4 x the TCC_EA0_WRREQ_DRAM_CREDIT_STALL
// cycles of the fast ones at identical request counts, profiles/r03_tables_pmc_summary.txt). Nothing a kernel does changes that,
// so wholememory_malloc can be asked (WM_MALLOC_PROBE=K, memory_handle.cpp:alloc_local) to allocate K candidates, time this
// probe on each and keep the best. The probe is the scatter's table side alone: pseudo-random rows, one 4 KiB batch (8 rows) per
// wave, launched in order. kind 0 writes zeros (fresh allocations only), kind 1 reads, kind 2 reads each row and writes it back.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "../backend.hpp"
#include "device_common.cuh"

namespace wm {
namespace {

constexpr int kRowBytes = 512;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ char* shfl_ptr(char* p, int src_lane)
{
  const uint64_t v  = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __shfl(static_cast<uint32_t>(v), src_lane, 64);
  const uint32_t hi = __shfl(static_cast<uint32_t>(v >> 32), src_lane, 64);
  return reinterpret_cast<char*>((static_cast<uint64_t>(hi) << 32) | lo);
}

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

template <int KIND>
__global__ __launch_bounds__(64) void probe_rows_kernel(char* base, uint64_t rows, uint64_t n, uint64_t seed, uint32_t* sink)
{
  const int lane      = threadIdx.x;
  const uint64_t tile = blockIdx.x;
  const uint64_t e    = tile * 8 + (lane & 7);
  char* mine          = base + (mix64(e ^ seed) % rows) * kRowBytes;   // lanes 0-7 hold the tile's 8 rows (repeated above)
  const int col       = lane & 31;
  u32x4 acc           = {0u, 0u, 0u, 0u};
  u32x4 d[4];
  char* dst[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int r = 2 * u + (lane >> 5);
    char* t     = shfl_ptr(mine, r);
    dst[u]      = tile * 8 + r < n ? t + col * 16 : nullptr;
    if (KIND != 0 && dst[u] != nullptr) d[u] = ld_global_nt<u32x4>(dst[u]);
  }
#pragma unroll
  for (int u = 0; u < 4; u++) {
    if (dst[u] == nullptr) continue;
    if (KIND == 0) st_global_nt<u32x4>(dst[u], acc);
    if (KIND == 1) acc[0] ^= d[u][0] ^ d[u][1] ^ d[u][2] ^ d[u][3];
    if (KIND == 2) st_global_nt<u32x4>(dst[u], d[u]);
  }
  if (KIND == 1 && acc[0] == 0x9E3779B9u && sink != nullptr) *sink = acc[0];   // keeps the loads; practically never taken
}

}  // namespace

// average milliseconds per GiB of rows touched (so that allocations of different sizes compare), over `reps` launches after
// one warm-up; each launch touches min(rows, 4 Mi) pseudo-random 512-byte rows of [ptr, ptr + bytes)
int hip_probe_memory(void* ptr, size_t bytes, int kind, int reps, float* ms_per_gib, void* stream_v)
{
  hipStream_t stream  = static_cast<hipStream_t>(stream_v);
  const uint64_t rows = bytes / kRowBytes;
  if (ptr == nullptr || rows == 0 || ms_per_gib == nullptr || kind < 0 || kind > 2) return -1;
  const uint64_t n  = std::min<uint64_t>(rows, UINT64_C(4) << 20);
  const dim3 grid(static_cast<unsigned>((n + 7) / 8)), block(64);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -2;
  reps = std::max(reps, 1);
  for (int i = -1; i < reps; i++) {
    if (i == 0) (void)hipEventRecord(e0, stream);
    const uint64_t seed = 0x1234567ull * static_cast<uint64_t>(i + 2);
    if (kind == 0) hipLaunchKernelGGL(probe_rows_kernel<0>, grid, block, 0, stream, static_cast<char*>(ptr), rows, n, seed, nullptr);
    else if (kind == 1) hipLaunchKernelGGL(probe_rows_kernel<1>, grid, block, 0, stream, static_cast<char*>(ptr), rows, n, seed, nullptr);
    else hipLaunchKernelGGL(probe_rows_kernel<2>, grid, block, 0, stream, static_cast<char*>(ptr), rows, n, seed, nullptr);
  }
  (void)hipEventRecord(e1, stream);
  const hipError_t rc = hipEventSynchronize(e1);
  float ms            = 0;
  if (rc == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc != hipSuccess || hipGetLastError() != hipSuccess) return -2;
  *ms_per_gib = ms / reps / (static_cast<float>(n) * kRowBytes / (1024.f * 1024.f * 1024.f));
  return 0;
}

}  // namespace wm
