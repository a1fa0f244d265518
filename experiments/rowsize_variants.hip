// Experiment harness: flat-stream gather / scatter for arbitrary 16-byte-multiple row sizes.
// usage: rowsize_variants <row_bytes> [n] [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void gen_idx(int64_t* idx, int64_t n, int64_t rows, uint64_t seed)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
  x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29; x *= 0x94D049BB133111EBull; x ^= x >> 32;
  idx[i] = (int64_t)(x % (uint64_t)rows);
}
__global__ void fill_tab(uint32_t* t, int64_t n_words, int words_per_row)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n_words; i += (int64_t)gridDim.x * blockDim.x) t[i] = (uint32_t)((i / words_per_row) * 31 + (i % words_per_row));
}
__device__ __forceinline__ char* shfl_ptr(char* p, int src)
{
  uint64_t v = (uint64_t)p;
  uint32_t lo = __shfl((uint32_t)v, src, 64), hi = __shfl((uint32_t)(v >> 32), src, 64);
  return (char*)(((uint64_t)hi << 32) | lo);
}
// rows of row_bytes (multiple of 4) stored with a 16-byte-padded stride; plain side dense (stride = row_bytes, rows only
// 4-byte aligned): full 16-byte vectors at whatever alignment the row has + a dword tail
template <int KU, bool GATHER>
__global__ __launch_bounds__(256) void k_flat_ragged(const char* tab, const int64_t* idx, char* plain, int64_t n, int rv, float rcp,
                                                     int64_t row_bytes, int64_t tab_stride)
{
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * 256) >> 6;
  const int64_t tiles = (n + 63) / 64;
  const int n_vec = 64 * rv;
  const int tail = (int)(row_bytes - 16 * (rv - 1));  // 4, 8, 12 or 16 bytes in the last slot
  for (int64_t tile = wave; tile < tiles; tile += n_waves) {
    const int64_t e = tile * 64 + lane;
    char* my_tab = nullptr;
    if (e < n) { int64_t id = idx[e]; if (id >= 0) my_tab = const_cast<char*>(tab) + id * tab_stride; }
    char* plain_tile = plain + tile * 64 * row_bytes;
#pragma unroll 1
    for (int v0 = 0; v0 < n_vec; v0 += 64 * KU) {
      u32x4 data[KU]; char* dst[KU]; bool part[KU];
#pragma unroll
      for (int u = 0; u < KU; u++) {
        const int v = v0 + u * 64 + lane;
        int row = (int)((float)v * rcp);
        int col = v - row * rv;
        if (col < 0) { row--; col += rv; }
        if (col >= rv) { row++; col -= rv; }
        char* t = shfl_ptr(my_tab, row & 63);
        char* q = plain_tile + (int64_t)row * row_bytes + col * 16;
        const bool ok = v < n_vec && t != nullptr;
        part[u] = col == rv - 1 && tail != 16;
        const char* src = GATHER ? t + col * 16 : q;
        dst[u] = ok ? (GATHER ? q : t + col * 16) : nullptr;
        if (ok) {
          if (!part[u]) data[u] = *(const u32x4*)src;
          else { for (int w = 0; w < 3; w++) if (w * 4 < tail) data[u][w] = ((const uint32_t*)src)[w]; }
        }
      }
#pragma unroll
      for (int u = 0; u < KU; u++)
        if (dst[u]) {
          if (!part[u]) __builtin_nontemporal_store(data[u], (u32x4*)dst[u]);
          else { for (int w = 0; w < 3; w++) if (w * 4 < tail) ((uint32_t*)dst[u])[w] = data[u][w]; }
        }
    }
  }
}
__global__ void check_out(const uint32_t* out, const int64_t* idx, int64_t n, int words_per_row, unsigned long long* bad)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n * words_per_row; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / words_per_row; int c = i % words_per_row;
    if (out[i] != (uint32_t)(idx[r] * 31 + c)) atomicAdd(bad, 1ull);
  }
}


// flat stream: vector v of a 64-row tile -> row v / RV, column v % RV
template <int KU, bool GATHER, bool NT_LOAD, bool NT_STORE>
__global__ __launch_bounds__(256) void k_flat(const char* tab, const int64_t* idx, char* plain, int64_t n, int rv, float rcp, int64_t row_bytes)
{
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * 256) >> 6;
  const int64_t tiles = (n + 63) / 64;
  const int n_vec = 64 * rv;
  for (int64_t tile = wave; tile < tiles; tile += n_waves) {
    const int64_t e = tile * 64 + lane;
    char* my_tab = nullptr;
    if (e < n) { int64_t id = idx[e]; if (id >= 0) my_tab = const_cast<char*>(tab) + id * row_bytes; }
    char* plain_tile = plain + tile * 64 * row_bytes;
#pragma unroll 1
    for (int v0 = 0; v0 < n_vec; v0 += 64 * KU) {
      u32x4 data[KU]; char* dst[KU];
#pragma unroll
      for (int u = 0; u < KU; u++) {
        const int v = v0 + u * 64 + lane;
        int row = (int)((float)v * rcp);
        int col = v - row * rv;
        if (col < 0) { row--; col += rv; }
        if (col >= rv) { row++; col -= rv; }
        char* t = shfl_ptr(my_tab, row & 63);
        char* q = plain_tile + (int64_t)v * 16;   // contiguous plain side
        const bool ok = v < n_vec && t != nullptr;
        const char* src = (GATHER ? t + col * 16 : q);
        dst[u] = ok ? (GATHER ? q : t + col * 16) : nullptr;
        if (ok) data[u] = NT_LOAD ? __builtin_nontemporal_load((const u32x4*)src) : *(const u32x4*)src;
      }
#pragma unroll
      for (int u = 0; u < KU; u++)
        if (dst[u]) { if (NT_STORE) __builtin_nontemporal_store(data[u], (u32x4*)dst[u]); else *(u32x4*)dst[u] = data[u]; }
    }
  }
}

template <typename F>
float time_it(F f, int iters)
{
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  f(); f(); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int i = 0; i < iters; i++) f();
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}

int main(int argc, char** argv)
{
  int64_t row_bytes = argc > 1 ? atoll(argv[1]) : 400;
  int64_t n = argc > 2 ? atoll(argv[2]) : 0;
  int iters = argc > 3 ? atoi(argv[3]) : 10;
  if (row_bytes % 16 != 0) {
    int64_t ts = (row_bytes + 15) / 16 * 16;
    int64_t rows = 8000000000ll / ts;
    if (n == 0) n = std::min<int64_t>(10000000, 4000000000ll / row_bytes);
    int rv = (int)(ts / 16);
    char *tab, *out; int64_t* idx; unsigned long long* bad;
    CHECK(hipMalloc(&tab, rows * ts)); CHECK(hipMalloc(&out, n * row_bytes + 64)); CHECK(hipMalloc(&idx, n * 8)); CHECK(hipMalloc(&bad, 8));
    hipLaunchKernelGGL(fill_tab, dim3(8192), dim3(256), 0, 0, (uint32_t*)tab, rows * ts / 4, (int)(ts / 4));
    hipLaunchKernelGGL(gen_idx, dim3((n + 255) / 256), dim3(256), 0, 0, idx, n, rows, 42);
    CHECK(hipDeviceSynchronize());
    const int64_t tiles = (n + 63) / 64;
    const double algo = (double)n * (8 + 2 * row_bytes);
    for (int grid : {4096, 8192}) {
      int g = (int)std::min<int64_t>(grid, (tiles + 3) / 4);
      CHECK(hipMemset(out, 0, n * row_bytes));
      float ms = time_it([&] { hipLaunchKernelGGL((k_flat_ragged<4, true>), dim3(g), dim3(256), 0, 0, tab, idx, out, n, rv, 1.0f / rv, row_bytes, ts); }, iters);
      CHECK(hipMemset(bad, 0, 8));
      hipLaunchKernelGGL(check_out, dim3(8192), dim3(256), 0, 0, (const uint32_t*)out, idx, n, (int)(row_bytes / 4), bad);
      unsigned long long h; CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
      printf("gather ragged ku4 grid %d: %.3f ms %.0f GB/s (%.1f%%) bad=%llu\n", g, ms, algo / ms / 1e6, algo / ms / 1e6 / 80.0, h);
      float ms2 = time_it([&] { hipLaunchKernelGGL((k_flat_ragged<4, false>), dim3(g), dim3(256), 0, 0, tab, idx, out, n, rv, 1.0f / rv, row_bytes, ts); }, iters);
      printf("scatter ragged ku4 grid %d: %.3f ms %.0f GB/s (%.1f%%)\n", g, ms2, algo / ms2 / 1e6, algo / ms2 / 1e6 / 80.0);
    }
    return 0;
  }
  int64_t rows = 8000000000ll / row_bytes;
  if (n == 0) n = std::min<int64_t>(10000000, 4000000000ll / row_bytes);
  int rv = (int)(row_bytes / 16);
  char *tab, *out; int64_t* idx; unsigned long long* bad;
  CHECK(hipMalloc(&tab, rows * row_bytes)); CHECK(hipMalloc(&out, n * row_bytes)); CHECK(hipMalloc(&idx, n * 8)); CHECK(hipMalloc(&bad, 8));
  hipLaunchKernelGGL(fill_tab, dim3(8192), dim3(256), 0, 0, (uint32_t*)tab, rows * row_bytes / 4, (int)(row_bytes / 4));
  hipLaunchKernelGGL(gen_idx, dim3((n + 255) / 256), dim3(256), 0, 0, idx, n, rows, 42);
  CHECK(hipDeviceSynchronize());
  const int64_t tiles = (n + 63) / 64;
  const double algo = (double)n * (8 + 2 * row_bytes);
  auto report = [&](const char* name, int grid, float ms, bool check) {
    unsigned long long h = 0;
    if (check) {
      CHECK(hipMemset(bad, 0, 8));
      hipLaunchKernelGGL(check_out, dim3(8192), dim3(256), 0, 0, (const uint32_t*)out, idx, n, (int)(row_bytes / 4), bad);
      CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    }
    printf("%-28s grid %6d : %.3f ms  %.0f GB/s (%.1f%%)%s\n", name, grid, ms, algo / ms / 1e6, algo / ms / 1e6 / 80.0, h ? "  MISMATCH" : "");
  };
  float rcp = 1.0f / rv;
  for (int grid : {2048, 4096, 8192, 16384}) {
    int g = (int)std::min<int64_t>(grid, (tiles + 3) / 4);
#define RUN(KU, G, NL, NS, name) { CHECK(hipMemset(out, 0, n * row_bytes)); \
    float ms = time_it([&] { hipLaunchKernelGGL((k_flat<KU, G, NL, NS>), dim3(g), dim3(256), 0, 0, tab, idx, out, n, rv, rcp, row_bytes); }, iters); \
    report(name, g, ms, G); }
    RUN(2, true, true, true, "gather flat ku2 nt/nt");
    RUN(4, true, true, true, "gather flat ku4 nt/nt");
    RUN(8, true, true, true, "gather flat ku8 nt/nt");
    RUN(4, true, false, true, "gather flat ku4 ld/nt");
    RUN(4, true, true, false, "gather flat ku4 nt/st");
    RUN(4, false, true, true, "scatter flat ku4 nt/nt");
    RUN(8, false, true, true, "scatter flat ku8 nt/nt");
    RUN(4, false, true, false, "scatter flat ku4 nt/st");
  }
  return 0;
}
