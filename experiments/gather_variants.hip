// Experiment harness (not product code): variants of the 512 B-row gather kernel, timed with HIP events.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -ffp-contract=off experiments/gather_variants.hip wholegraph_amd/csrc/tensor_description.cpp -o experiments/gather_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>
#include "../wholegraph_amd/csrc/kernels/rows.hip"   // the product kernels, launched through wm::hip_gather_rows

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void gen_idx(int64_t* idx, int64_t n, int64_t rows, uint64_t seed)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  idx[i] = (int64_t)(x % (uint64_t)rows);
}
__global__ void fill_tab(float* t, int64_t n4)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    reinterpret_cast<float4*>(t)[i] = make_float4((float)((i >> 5) & 0xFFFFFF), 1, 2, 3);
}

__device__ __forceinline__ const char* shfl_ptr(const char* p, int src)
{
  uint64_t v = (uint64_t)p;
  uint32_t lo = __shfl((uint32_t)v, src, 64), hi = __shfl((uint32_t)(v >> 32), src, 64);
  return (const char*)(((uint64_t)hi << 32) | lo);
}

// V0: product kernel shape: tile of 64 entries per wave, bpermute broadcast, UNROLL steps
template <int UNROLL, bool NT_LOAD, bool NT_STORE>
__global__ __launch_bounds__(256) void k_tile(const char* tab, const int64_t* idx, char* out, int64_t n)
{
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * 256) >> 6;
  const int sub = lane >> 5, col = lane & 31;
  const int64_t tiles = (n + 63) / 64;
  for (int64_t tile = wave; tile < tiles; tile += nw) {
    int64_t e = tile * 64 + lane;
    const char* my = nullptr;
    if (e < n) my = tab + idx[e] * 512;
    char* obase = out + tile * 64 * 512;
    for (int s0 = 0; s0 < 64; s0 += 2 * UNROLL) {
      u32x4 d[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
        int ee = s0 + 2 * u + sub;
        const char* src = shfl_ptr(my, ee);
        const u32x4* p = (const u32x4*)(src + col * 16);
        if (src) d[u] = NT_LOAD ? __builtin_nontemporal_load(p) : *p;
      }
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
        int ee = s0 + 2 * u + sub;
        if (tile * 64 + ee < n) {
          u32x4* q = (u32x4*)(obase + ee * 512 + col * 16);
          if (NT_STORE) __builtin_nontemporal_store(d[u], q); else *q = d[u];
        }
      }
    }
  }
}

// V1: scalar-base variant: each wave handles one row pair per step using readlane (SGPR base), 2 rows/step
template <int UNROLL, bool NT_LOAD, bool NT_STORE>
__global__ __launch_bounds__(256) void k_readlane(const char* tab, const int64_t* idx, char* out, int64_t n)
{
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * 256) >> 6;
  const int col = lane & 31;
  const bool hi = lane >= 32;
  const int64_t tiles = (n + 63) / 64;
  for (int64_t tile = wave; tile < tiles; tile += nw) {
    int64_t e = tile * 64 + lane;
    int64_t my = (e < n) ? idx[e] : 0;
    uint32_t mlo = (uint32_t)my, mhi = (uint32_t)((uint64_t)my >> 32);
    char* obase = out + tile * 64 * 512 + (hi ? 512 : 0) + col * 16;
#pragma unroll 1
    for (int s0 = 0; s0 < 64; s0 += 2 * UNROLL) {
      u32x4 d[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
        int e0 = s0 + 2 * u;
        uint64_t i0 = ((uint64_t)__builtin_amdgcn_readlane(mhi, e0) << 32) | __builtin_amdgcn_readlane(mlo, e0);
        uint64_t i1 = ((uint64_t)__builtin_amdgcn_readlane(mhi, e0 + 1) << 32) | __builtin_amdgcn_readlane(mlo, e0 + 1);
        uint64_t r = hi ? i1 : i0;
        const u32x4* p = (const u32x4*)(tab + r * 512 + col * 16);
        d[u] = NT_LOAD ? __builtin_nontemporal_load(p) : *p;
      }
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
        u32x4* q = (u32x4*)(obase + (s0 + 2 * u) * 512);
        if (NT_STORE) __builtin_nontemporal_store(d[u], q); else *q = d[u];
      }
    }
  }
}

// V2: one row per HALF-wave pair but 16 lanes x 32B (2 x dwordx4 per lane): 4 rows per step
template <int UNROLL, bool NT_LOAD, bool NT_STORE>
__global__ __launch_bounds__(256) void k_4rows(const char* tab, const int64_t* idx, char* out, int64_t n)
{
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * 256) >> 6;
  const int sub = lane >> 4, col = lane & 15;
  const int64_t tiles = (n + 63) / 64;
  for (int64_t tile = wave; tile < tiles; tile += nw) {
    int64_t e = tile * 64 + lane;
    const char* my = nullptr;
    if (e < n) my = tab + idx[e] * 512;
    char* obase = out + tile * 64 * 512;
    for (int s0 = 0; s0 < 64; s0 += 4 * UNROLL) {
      u32x4 d[UNROLL][2];
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
        int ee = s0 + 4 * u + sub;
        const char* src = shfl_ptr(my, ee);
        const u32x4* p = (const u32x4*)(src + col * 16);
        if (src) { d[u][0] = NT_LOAD ? __builtin_nontemporal_load(p) : *p; d[u][1] = NT_LOAD ? __builtin_nontemporal_load(p + 16) : p[16]; }
      }
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
        int ee = s0 + 4 * u + sub;
        if (tile * 64 + ee < n) {
          u32x4* q = (u32x4*)(obase + ee * 512 + col * 16);
          if (NT_STORE) { __builtin_nontemporal_store(d[u][0], q); __builtin_nontemporal_store(d[u][1], q + 16); } else { q[0] = d[u][0]; q[16] = d[u][1]; }
        }
      }
    }
  }
}

// streaming copy of the same byte volume (upper bound for read+write)
__global__ __launch_bounds__(256) void k_copy(const u32x4* in, u32x4* out, int64_t n16)
{
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256)
    __builtin_nontemporal_store(in[i], out + i);
}
// read-only random gather (no stores): isolates the random-read side
__global__ __launch_bounds__(256) void k_readonly(const char* tab, const int64_t* idx, char* out, int64_t n)
{
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * 256) >> 6;
  const int sub = lane >> 5, col = lane & 31;
  const int64_t tiles = (n + 63) / 64;
  u32x4 acc = {0, 0, 0, 0};
  for (int64_t tile = wave; tile < tiles; tile += nw) {
    int64_t e = tile * 64 + lane;
    const char* my = nullptr;
    if (e < n) my = tab + idx[e] * 512;
    for (int s0 = 0; s0 < 64; s0 += 16) {
      u32x4 d[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const char* src = shfl_ptr(my, s0 + 2 * u + sub);
        if (src) d[u] = *(const u32x4*)(src + col * 16);
      }
#pragma unroll
      for (int u = 0; u < 8; u++) acc ^= d[u];
    }
  }
  if (acc.x == 0x12345678 && acc.y == 0x9abcdef) *(u32x4*)out = acc;
}

struct Variant { const char* name; void (*launch)(const char*, const int64_t*, char*, int64_t, int, hipStream_t); };
#define LAUNCHER(fn) [](const char* t, const int64_t* i, char* o, int64_t n, int blocks, hipStream_t s) { hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, s, t, i, o, n); }

int main(int argc, char** argv)
{
  int64_t rows = argc > 1 ? atoll(argv[1]) : 100000000ll;
  int64_t n    = argc > 2 ? atoll(argv[2]) : 10000000ll;
  int iters    = argc > 3 ? atoi(argv[3]) : 10;
  char *tab, *out; int64_t* idx;
  CK(hipMalloc(&tab, rows * 512)); CK(hipMalloc(&out, n * 512)); CK(hipMalloc(&idx, n * 8));
  hipLaunchKernelGGL(fill_tab, dim3(4096), dim3(256), 0, 0, (float*)tab, rows * 32);
  hipLaunchKernelGGL(gen_idx, dim3((n + 255) / 256), dim3(256), 0, 0, idx, n, rows, 42ull);
  CK(hipDeviceSynchronize());
  std::vector<Variant> vs = {
    {"tile_u8_ld_ntst", LAUNCHER((k_tile<8, false, true>))},
    {"tile_u8_ld_st", LAUNCHER((k_tile<8, false, false>))},
    {"tile_u8_ntld_ntst", LAUNCHER((k_tile<8, true, true>))},
    {"tile_u4_ld_ntst", LAUNCHER((k_tile<4, false, true>))},
    {"tile_u16_ld_ntst", LAUNCHER((k_tile<16, false, true>))},
    {"readlane_u8_ld_ntst", LAUNCHER((k_readlane<8, false, true>))},
    {"readlane_u16_ld_ntst", LAUNCHER((k_readlane<16, false, true>))},
    {"readlane_u8_ntld_ntst", LAUNCHER((k_readlane<8, true, true>))},
    {"4rows_u4_ld_ntst", LAUNCHER((k_4rows<4, false, true>))},
    {"4rows_u8_ld_ntst", LAUNCHER((k_4rows<8, false, true>))},
    {"readonly_u8", LAUNCHER(k_readonly)},
  };
  vs.clear();
  vs.push_back({"readlane_u8_ntld_ntst", LAUNCHER((k_readlane<8, true, true>))});
  vs.push_back({"PRODUCT_gather_rows", [](const char* t, const int64_t* i, char* o, int64_t n, int blocks, hipStream_t s) {
    wm_rows_args a{};
    a.gref = wholememory_gref_t{(void*)t, nullptr, 1, 0, true};
    a.table_dtype = WHOLEMEMORY_DT_FLOAT; a.dim = 128; a.table_stride = 128; a.indices = i; a.index_dtype = WHOLEMEMORY_DT_INT64;
    a.n = n; a.plain = o; a.plain_dtype = WHOLEMEMORY_DT_FLOAT; a.plain_stride = 128; a.max_blocks = blocks;
    if (wm::hip_gather_rows(&a, s) != 0) { printf("product launch failed\n"); exit(1); } }});
  vs.push_back({"PRODUCT_scatter_rows", [](const char* t, const int64_t* i, char* o, int64_t n, int blocks, hipStream_t s) {
    wm_rows_args a{};
    a.gref = wholememory_gref_t{(void*)t, nullptr, 1, 0, true};
    a.table_dtype = WHOLEMEMORY_DT_FLOAT; a.dim = 128; a.table_stride = 128; a.indices = i; a.index_dtype = WHOLEMEMORY_DT_INT64;
    a.n = n; a.plain = o; a.plain_dtype = WHOLEMEMORY_DT_FLOAT; a.plain_stride = 128; a.max_blocks = blocks;
    if (wm::hip_scatter_rows(&a, s) != 0) { printf("product launch failed\n"); exit(1); } }});
  const bool pmc_mode = argc > 4 && std::string(argv[4]) == "pmc";
  std::vector<int> grids = {2048, 4096, 8192, 16384};
  if (pmc_mode) { vs.erase(vs.begin()); grids = {8192}; iters = 3; }
  if (argc > 4 && std::string(argv[4]) == "product") { vs.erase(vs.begin()); grids = {4096, 8192}; }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("rows=%lld n=%lld iters=%d  (algorithmic bytes per launch = %.3f GB)\n", (long long)rows, (long long)n, iters, n * 1032.0 / 1e9);
  for (auto& v : vs) {
    for (int g : grids) {
      v.launch(tab, idx, out, n, g, 0);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; i++) v.launch(tab, idx, out, n, g, 0);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
      printf("%-24s grid=%5d  %.4f ms  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)\n", v.name, g, ms, n * 1032.0 / ms / 1e6, n * 1032.0 / ms / 1e6 / 80.0);
    }
  }
  // streaming copy of n*512 bytes
  {
    int64_t n16 = n * 32;
    for (int g : {8192}) {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, (const u32x4*)tab, (u32x4*)out, n16);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
      printf("stream_copy grid=%d %.4f ms  %.1f GB/s (read+write)\n", g, ms, 2.0 * n * 512 / ms / 1e6);
    }
  }
  return 0;
}
