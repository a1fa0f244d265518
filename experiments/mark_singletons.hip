// Experiment harness (not product code), round 3: can the owner-side gradient apply skip the id SORT for the ids that occur once?
// (uniform 10 M-in-100 M batch: 95 % of the ids are singletons; the sort + run detection cost 0.3-0.4 ms beside a 2.6 ms step kernel)
//  1. mark: every id sets its bit in a bitmap of the shard's rows with a RETURNING device-scope atomicOr; an id that finds its bit
//     set marks the row in a second bitmap ("this row has duplicates"). What do 10 M random returning atomics cost?
//  2. direct step: gradient rows walked IN RECEIVE ORDER (dense stream), table row read + written at ids[i]; rows whose id is
//     marked are left to the sorted route. Compared in the same process with the sorted shape (ascending ids, gradient row
//     through order[]).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/mark_singletons.hip -o experiments/mark_singletons
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GAS __attribute__((address_space(1)))

__global__ void gen_idx(int64_t* idx, int64_t n, int64_t rows, uint64_t seed)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  idx[i] = (int64_t)(x % (uint64_t)rows);
}

__global__ void fill_f(float* p, int64_t n, float v)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// ---- 1. mark ----
// VAR 0: one returning atomicOr per id + a non-returning one for repeats
// VAR 1: ids of a wave that share a 32-bit word are not merged either, but the atomic is 64-bit (half the words)
// VAR 2: non-returning atomicOr only (what the returning form costs on top)
template <int VAR>
__global__ __launch_bounds__(256) void mark_kernel(const int64_t* ids, int64_t n, uint32_t* seen, uint32_t* dup)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t id = ids[i];
  if (id < 0) return;
  if (VAR == 1) {
    const unsigned long long bit = 1ull << (id & 63);
    const unsigned long long old = atomicOr(reinterpret_cast<unsigned long long*>(seen) + (id >> 6), bit);
    if (old & bit) atomicOr(reinterpret_cast<unsigned long long*>(dup) + (id >> 6), bit);
  } else {
    const uint32_t bit = 1u << (id & 31);
    if (VAR == 2) {
      __hip_atomic_fetch_or(seen + (id >> 5), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const uint32_t old = __hip_atomic_fetch_or(seen + (id >> 5), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old & bit) __hip_atomic_fetch_or(dup + (id >> 5), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ void popc_kernel(const uint32_t* w, int64_t nw, unsigned long long* out)
{
  unsigned long long c = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += (int64_t)gridDim.x * blockDim.x) c += __popc(w[i]);
  atomicAdd(out, c);
}

// ---- 2. step (SGD, 512 B fp32 rows) ----
__device__ __forceinline__ char* readlane_ptr(char* p, int lane)
{
  uint64_t v = (uint64_t)p;
  uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, lane), hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), lane);
  return (char*)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ f32x4 ldnt(const char* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
__device__ __forceinline__ void stnt(char* p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); }

// one 8-row tile per wave, one wave per workgroup, in order (the shape of rows_batch_kernel)
// MODE 0: direct — gradient row i is row i of grads, table row ids[i], rows marked in dup are skipped
// MODE 1: sorted — ids ascending, gradient row order[i]
// MODE 2: direct without the dup test
template <int MODE, int OCC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, OCC))) void step8_kernel(char* tab, const char* grads, const int64_t* ids, const int32_t* order,
                                                             const uint32_t* dup, int64_t n, float lr, float wd)
{
  const int lane = threadIdx.x;
  const int64_t tile = blockIdx.x;
  const int64_t e = tile * 8 + lane;
  char* my_tab = nullptr;
  const char* my_g = nullptr;
  if (lane < 8 && e < n) {
    const int64_t id = ids[e];
    bool ok = id >= 0;
    if (MODE == 0 && ok) ok = !((dup[id >> 5] >> (id & 31)) & 1);
    if (ok) {
      my_tab = tab + id * 512;
      my_g   = grads + (MODE == 1 ? (int64_t)order[e] : e) * 512;
    }
  }
  const int col = lane & 31;
  const bool upper = lane >= 32;
  f32x4 g[4], t[4];
  char* dst[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    char* a = readlane_ptr(my_tab, 2 * u);
    char* b = readlane_ptr(my_tab, 2 * u + 1);
    char* ga = readlane_ptr(const_cast<char*>(my_g), 2 * u);
    char* gb = readlane_ptr(const_cast<char*>(my_g), 2 * u + 1);
    char* tr = upper ? b : a;
    const char* gr = upper ? gb : ga;
    dst[u] = tr ? tr + col * 16 : nullptr;
    if (tr) {
      g[u] = ldnt(gr + col * 16);
      t[u] = ldnt(tr + col * 16);
    }
  }
#pragma unroll
  for (int u = 0; u < 4; u++) {
    if (dst[u]) {
      f32x4 gv = g[u] + wd * t[u];
      stnt(dst[u], t[u] - lr * gv);
    }
  }
}

// persistent 64-row tiles (the shape of step_tile_kernel): 8192 workgroups of 256
template <int MODE>
__global__ __launch_bounds__(256) void step64_kernel(char* tab, const char* grads, const int64_t* ids, const int32_t* order, const uint32_t* dup,
                                                      int64_t n, float lr, float wd)
{
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * 256) >> 6;
  const int64_t tiles = (n + 63) / 64;
  const int col = lane & 31;
  const bool upper = lane >= 32;
  for (int64_t tile = wave; tile < tiles; tile += nw) {
    const int64_t e = tile * 64 + lane;
    char* my_tab = nullptr;
    const char* my_g = nullptr;
    if (e < n) {
      const int64_t id = ids[e];
      bool ok = id >= 0;
      if (MODE == 0 && ok) ok = !((dup[id >> 5] >> (id & 31)) & 1);
      if (ok) {
        my_tab = tab + id * 512;
        my_g   = grads + (MODE == 1 ? (int64_t)order[e] : e) * 512;
      }
    }
#pragma unroll 1
    for (int s = 0; s < 64; s += 8) {
      f32x4 g[4], t[4];
      char* dst[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        char* a = readlane_ptr(my_tab, s + 2 * u);
        char* b = readlane_ptr(my_tab, s + 2 * u + 1);
        char* ga = readlane_ptr(const_cast<char*>(my_g), s + 2 * u);
        char* gb = readlane_ptr(const_cast<char*>(my_g), s + 2 * u + 1);
        char* tr = upper ? b : a;
        const char* gr = upper ? gb : ga;
        dst[u] = tr ? tr + col * 16 : nullptr;
        if (tr) {
          g[u] = ldnt(gr + col * 16);
          t[u] = ldnt(tr + col * 16);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (dst[u]) {
          f32x4 gv = g[u] + wd * t[u];
          stnt(dst[u], t[u] - lr * gv);
        }
      }
    }
  }
}

template <typename F>
float time_ms(F&& f, int iters = 10)
{
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; i++) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}

int main(int argc, char** argv)
{
  const int64_t rows = argc > 1 ? atoll(argv[1]) : 100000000;
  const int64_t n    = argc > 2 ? atoll(argv[2]) : 10000000;
  char *tab, *grads;
  int64_t* ids;
  CK(hipMalloc(&tab, rows * 512));
  CK(hipMalloc(&grads, n * 512));
  CK(hipMalloc(&ids, n * 8));
  fill_f<<<8192, 256>>>((float*)tab, rows * 128, 1.0f);
  fill_f<<<8192, 256>>>((float*)grads, n * 128, 0.5f);
  gen_idx<<<(n + 255) / 256, 256>>>(ids, n, rows, 42);
  CK(hipDeviceSynchronize());
  const int64_t words = (rows + 63) / 64 * 2;
  uint32_t *seen, *dup;
  CK(hipMalloc(&seen, words * 4));
  CK(hipMalloc(&dup, words * 4));
  unsigned long long* cnt;
  CK(hipMalloc(&cnt, 16));

  // --- mark ---
  for (int var = 0; var < 3; var++) {
    float ms = time_ms([&] {
      CK(hipMemsetAsync(seen, 0, words * 4));
      CK(hipMemsetAsync(dup, 0, words * 4));
      if (var == 0) mark_kernel<0><<<(n + 255) / 256, 256>>>(ids, n, seen, dup);
      if (var == 1) mark_kernel<1><<<(n + 255) / 256, 256>>>(ids, n, seen, dup);
      if (var == 2) mark_kernel<2><<<(n + 255) / 256, 256>>>(ids, n, seen, dup);
    });
    CK(hipMemset(cnt, 0, 16));
    popc_kernel<<<1024, 256>>>(seen, words, cnt);
    popc_kernel<<<1024, 256>>>(dup, words, cnt + 1);
    unsigned long long h[2];
    CK(hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost));
    float ms0 = time_ms([&] {
      CK(hipMemsetAsync(seen, 0, words * 4));
      CK(hipMemsetAsync(dup, 0, words * 4));
    });
    printf("mark var %d: %.1f us with the two memsets (memsets alone %.1f us): distinct ids %llu, rows with duplicates %llu\n", var, ms * 1e3,
           ms0 * 1e3, h[0], h[1]);
  }
  // the dup bitmap of variant 0 for the step kernels
  CK(hipMemset(seen, 0, words * 4));
  CK(hipMemset(dup, 0, words * 4));
  mark_kernel<0><<<(n + 255) / 256, 256>>>(ids, n, seen, dup);
  CK(hipDeviceSynchronize());

  // --- sorted shape: ascending ids + order ---
  std::vector<int64_t> h_ids(n);
  CK(hipMemcpy(h_ids.data(), ids, n * 8, hipMemcpyDeviceToHost));
  std::vector<int32_t> h_ord(n);
  std::iota(h_ord.begin(), h_ord.end(), 0);
  std::stable_sort(h_ord.begin(), h_ord.end(), [&](int32_t a, int32_t b) { return h_ids[a] < h_ids[b]; });
  std::vector<int64_t> h_sorted(n);
  for (int64_t i = 0; i < n; i++) h_sorted[i] = h_ids[h_ord[i]];
  // keep one entry per id in the sorted shape (the others would race; the product folds them)
  for (int64_t i = 1; i < n; i++) if (h_sorted[i] == h_sorted[i - 1]) h_sorted[i] = -1;
  for (int64_t i = 1; i < n; i++) if (h_sorted[i] == -1) { /* skip */ }
  int64_t* sids;
  int32_t* ord;
  CK(hipMalloc(&sids, n * 8));
  CK(hipMalloc(&ord, n * 4));
  CK(hipMemcpy(sids, h_sorted.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(ord, h_ord.data(), n * 4, hipMemcpyHostToDevice));

  const double gb = (double)n * (8 + 3 * 512) / 1e9;
  const unsigned tiles8 = (unsigned)((n + 7) / 8);
  for (int rep = 0; rep < 3; rep++) {
    float a = time_ms([&] { step8_kernel<0, 6><<<tiles8, 64>>>(tab, grads, ids, ord, dup, n, 1e-6f, 0.f); });
    float a8 = time_ms([&] { step8_kernel<0, 8><<<tiles8, 64>>>(tab, grads, ids, ord, dup, n, 1e-6f, 0.f); });
    float b = time_ms([&] { step8_kernel<1, 6><<<tiles8, 64>>>(tab, grads, sids, ord, dup, n, 1e-6f, 0.f); });
    float c = time_ms([&] { step8_kernel<2, 6><<<tiles8, 64>>>(tab, grads, ids, ord, dup, n, 1e-6f, 0.f); });
    float d = time_ms([&] { step64_kernel<0><<<8192, 256>>>(tab, grads, ids, ord, dup, n, 1e-6f, 0.f); });
    float e = time_ms([&] { step64_kernel<1><<<8192, 256>>>(tab, grads, sids, ord, dup, n, 1e-6f, 0.f); });
    printf("rep %d  in-order 8-row tiles: direct %.3f ms (%.1f %%; 8 waves %.3f)  sorted %.3f ms (%.1f %%)  direct, no dup test %.3f ms |"
           " persistent 64-row tiles: direct %.3f ms (%.1f %%)  sorted %.3f ms (%.1f %%)\n",
           rep, a, gb / a / 8 * 100, a8, b, gb / b / 8 * 100, c, d, gb / d / 8 * 100, e, gb / e / 8 * 100);
  }
  return 0;
}
