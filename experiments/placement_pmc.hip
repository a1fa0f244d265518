// Experiment harness (not product code), round 3: why does the 512 B-row gather run at 1.70 ... 1.95 ms depending on the
// physical placement of the (table, output) pair?
//  * one process: one 51.2 GB table, K hipMalloc output buffers + one physically contiguous one; every buffer probed, the
//    fastest / slowest hipMalloc buffer and the contiguous one become the classes "fast", "slow", "contig";
//  * every kernel launched on a class carries the class in its NAME (template tag), so that rocprofv3 --pmc rows of one
//    process can be attributed: `rocprofv3 --pmc <counters> --kernel-trace -- experiments/placement_pmc`
//  * kernel variants that change WHICH addresses are in flight together (the decorrelation candidates) A/B'd on all three.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/placement_pmc.hip -o experiments/placement_pmc
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

__device__ __forceinline__ const char* readlane_ptr(const char* p, int lane)
{
  uint64_t v = (uint64_t)p;
  uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, lane), hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), lane);
  return (const char*)(((uint64_t)hi << 32) | lo);
}

// The product's fast kernel specialised for 512 B rows (2 rows per wave step, v_readlane row bases, 4 steps in flight,
// nt loads + nt stores). VAR selects the order in which output addresses are touched:
//  0 product: wave w of the grid takes tiles w, w + W, ...; rows of a tile front to back
//  1 tile order permuted over the WHOLE output (odd multiplier mod tiles): the tiles in flight are spread over all of it
//  2 rows of a tile walked from a per-tile pseudo-random start pair (rotation)
//  3 odd tiles walked back to front
//  4 8 steps in flight
//  5 stores not nt
//  6 tile = 16 rows (8 KiB) with the permuted order of 1
template <int TAG, int VAR>
__global__ __launch_bounds__(256) void gather512(const char* tab, const int64_t* idx, char* out, int64_t n)
{
  constexpr int TR = VAR == 6 ? 16 : 64;
  constexpr int KU = VAR == 4 ? 8 : 4;
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * 256) >> 6;
  const int col = lane & 31;
  const bool upper = lane >= 32;
  const int64_t tiles = (n + TR - 1) / TR;
  for (int64_t t0 = wave; t0 < tiles; t0 += nw) {
    int64_t tile = t0;
    if (VAR == 1 || VAR == 6) tile = (int64_t)(((unsigned __int128)(uint64_t)t0 * 0x9E3779B1ull) % (uint64_t)tiles);  // (bijective when gcd = 1; tiles is not a multiple of the prime-ish constant)
    const int64_t e = tile * TR + lane;
    const char* my = nullptr;
    if (lane < TR && e < n) my = tab + idx[e] * 512;
    char* obase = out + tile * TR * 512 + (upper ? 512 : 0);
    int rot = 0;
    if (VAR == 2) rot = (int)((uint64_t)tile * 2654435761ull >> 20) & (TR / 2 - 1);
    const bool back = VAR == 3 && (tile & 1);
#pragma unroll 1
    for (int s0 = 0; s0 < TR / 2; s0 += KU) {
      u32x4 d[KU];
      char* dst[KU];
#pragma unroll
      for (int u = 0; u < KU; u++) {
        int s = s0 + u;
        if (VAR == 2) s = (s + rot) & (TR / 2 - 1);
        if (back) s = TR / 2 - 1 - s;
        const char* a = readlane_ptr(my, 2 * s);
        const char* b = readlane_ptr(my, 2 * s + 1);
        const char* src = upper ? b : a;
        dst[u] = src ? obase + (int64_t)s * 1024 + col * 16 : nullptr;
        if (src) d[u] = __builtin_nontemporal_load((const GAS u32x4*)(src + col * 16));
      }
#pragma unroll
      for (int u = 0; u < KU; u++) {
        if (dst[u]) {
          if (VAR == 5) *(GAS u32x4*)dst[u] = d[u];
          else __builtin_nontemporal_store(d[u], (GAS u32x4*)dst[u]);
        }
      }
    }
  }
}

// streaming copy of n16 16-byte pieces (nt loads, nt stores), grid-stride by wave-sized 4 KiB pieces
template <int TAG>
__global__ __launch_bounds__(256) void stream_copy(const u32x4* src, u32x4* dst, int64_t n16)
{
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256)
    __builtin_nontemporal_store(__builtin_nontemporal_load((const GAS u32x4*)(src + i)), (GAS u32x4*)(dst + i));
}

template <int TAG>
__global__ __launch_bounds__(256) void stream_fill(u32x4* dst, int64_t n16)
{
  u32x4 v = {1u, 2u, 3u, 4u};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256)
    __builtin_nontemporal_store(v, (GAS u32x4*)(dst + i));
}


// ---- round 3, second batch: the SHAPE OF THE WRITE FRONT --------------------------------------------------------------
// MODE 0 persistent grid-stride (tile = wave, wave + n_waves, ...: with 8192 workgroups and 2048 resident the output is
//        swept four times, a quarter of the tiles each time)
// MODE 1 one tile per wave, grid = tiles / 4 workgroups: the hardware dispatcher hands tiles out in order, so the tiles in
//        flight form one compact, strictly advancing window
// MODE 2 persistent waves, tiles handed out in order by an atomic counter (the same window, no relaunch cost)
template <int TAG, int TR, int MODE>
__global__ __launch_bounds__(256) void gather_front(const char* tab, const int64_t* idx, char* out, int64_t n, unsigned long long* counter)
{
  constexpr int KU = 4;
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * 256) >> 6;
  const int col = lane & 31;
  const bool upper = lane >= 32;
  const int64_t tiles = (n + TR - 1) / TR;
  int64_t tile = wave;
  if (MODE == 2) {
    unsigned long long t = 0;
    if (lane == 0) t = atomicAdd(counter, 1ull);
    tile = (int64_t)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(t >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)t));
  }
  while (tile < tiles) {
    unsigned long long nxt = 0;
    if (MODE == 2 && lane == 0) nxt = atomicAdd(counter, 1ull);   // the next tile's number travels while this one is moved
    const int64_t e = tile * TR + lane;
    const char* my = nullptr;
    if (lane < TR && e < n) my = tab + idx[e] * 512;
    char* obase = out + tile * TR * 512 + (upper ? 512 : 0);
#pragma unroll 1
    for (int s0 = 0; s0 < TR / 2; s0 += KU) {
      u32x4 d[KU];
      char* dst[KU];
#pragma unroll
      for (int u = 0; u < KU; u++) {
        const int s = s0 + u;
        const char* a = readlane_ptr(my, 2 * s);
        const char* b = readlane_ptr(my, 2 * s + 1);
        const char* src = upper ? b : a;
        dst[u] = src ? obase + (int64_t)s * 1024 + col * 16 : nullptr;
        if (src) d[u] = __builtin_nontemporal_load((const GAS u32x4*)(src + col * 16));
      }
#pragma unroll
      for (int u = 0; u < KU; u++)
        if (dst[u]) __builtin_nontemporal_store(d[u], (GAS u32x4*)dst[u]);
    }
    if (MODE == 0) tile += nw;
    else if (MODE == 1) break;
    else tile = (int64_t)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(nxt >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)nxt));
  }
}

// fill with the same three front shapes: PIECE bytes per wave visit
template <int TAG, int MODE>
__global__ __launch_bounds__(256) void fill_front(u32x4* dst, int64_t n16)
{
  u32x4 v = {1u, 2u, 3u, 4u};
  if (MODE == 0) {       // grid-stride by 16 B per thread (stream_fill above)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256)
      __builtin_nontemporal_store(v, (GAS u32x4*)(dst + i));
  } else {               // block b fills the contiguous 16 KiB piece b (4 x 16 B per thread, 4 KiB per instruction), blocks in order
    const int64_t base = (int64_t)blockIdx.x * 1024;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int64_t i = base + k * 256 + threadIdx.x;
      if (i < n16) __builtin_nontemporal_store(v, (GAS u32x4*)(dst + i));
    }
  }
}


// ---- round 3, third batch: in-order dispatch (one tile per wave), the tile / batch / workgroup shape ---------------------
// TR rows per wave moved as ONE batch of TR / 2 steps (all loads, then all stores); BLOCK threads per workgroup;
// SCATTER: the mirrored op (streamed input read in order, table rows written at random)
template <int TAG, int TR, int BLOCK, bool SCATTER, bool NT_LOAD>
__global__ __launch_bounds__(BLOCK) void rows_inorder(char* tab, const int64_t* idx, char* plain, int64_t n)
{
  constexpr int KU = TR / 2;
  const int lane = threadIdx.x & 63;
  const int64_t tile = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int col = lane & 31;
  const bool upper = lane >= 32;
  const int64_t e = tile * TR + lane;
  if (tile * TR >= n) return;
  const char* my = nullptr;
  if (lane < TR && e < n) my = tab + idx[e] * 512;
  char* pbase = plain + tile * TR * 512 + (upper ? 512 : 0);
  u32x4 d[KU];
  char* dst[KU];
#pragma unroll
  for (int u = 0; u < KU; u++) {
    const char* a = readlane_ptr(my, 2 * u);
    const char* b = readlane_ptr(my, 2 * u + 1);
    const char* t = upper ? b : a;
    char* q = pbase + (int64_t)u * 1024 + col * 16;
    const char* src = SCATTER ? q : t + col * 16;
    dst[u] = t ? (SCATTER ? (char*)t + col * 16 : q) : nullptr;
    if (t) d[u] = NT_LOAD ? __builtin_nontemporal_load((const GAS u32x4*)src) : *(const GAS u32x4*)src;
  }
#pragma unroll
  for (int u = 0; u < KU; u++)
    if (dst[u]) __builtin_nontemporal_store(d[u], (GAS u32x4*)dst[u]);
}

typedef void (*gather_fn)(const char*, const int64_t*, char*, int64_t);

template <int TAG>
gather_fn variant(int var)
{
  switch (var) {
    case 0: return gather512<TAG, 0>;
    case 1: return gather512<TAG, 1>;
    case 2: return gather512<TAG, 2>;
    case 3: return gather512<TAG, 3>;
    case 4: return gather512<TAG, 4>;
    case 5: return gather512<TAG, 5>;
    default: return gather512<TAG, 6>;
  }
}
gather_fn pick(int tag, int var)
{
  switch (tag) {
    case 0: return variant<0>(var);
    case 1: return variant<1>(var);
    case 2: return variant<2>(var);
    default: return variant<3>(var);
  }
}

static hipEvent_t e0, e1;
float time_gather(gather_fn f, const char* tab, const int64_t* idx, char* out, int64_t n, int iters)
{
  hipLaunchKernelGGL(f, dim3(8192), dim3(256), 0, 0, tab, idx, out, n);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL(f, dim3(8192), dim3(256), 0, 0, tab, idx, out, n);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

int main(int argc, char** argv)
{
  const int K        = argc > 1 ? atoi(argv[1]) : 8;
  const int iters    = argc > 2 ? atoi(argv[2]) : 6;
  const int64_t rows = 100000000ll, n = 10000000ll;
  const size_t out_bytes = (size_t)n * 512;
  char* tab; int64_t* idx;
  CK(hipMalloc(&tab, (size_t)rows * 512));
  CK(hipMalloc(&idx, n * 8));
  std::vector<char*> outs(K);
  for (auto& o : outs) CK(hipMalloc(&o, out_bytes));
  char* contig = nullptr;
  if (hipExtMallocWithFlags((void**)&contig, out_bytes, hipDeviceMallocContiguous) != hipSuccess) { contig = nullptr; (void)hipGetLastError(); }
  hipLaunchKernelGGL(gen_idx, dim3((n + 255) / 256), dim3(256), 0, 0, idx, n, rows, 42ull);
  CK(hipMemsetAsync(tab, 0, (size_t)rows * 512, 0));
  CK(hipDeviceSynchronize());
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  printf("== probe: product-order kernel on every output buffer (ms per 10 M rows)\n");
  std::vector<float> ms(K);
  for (int k = 0; k < K; k++) { ms[k] = time_gather(pick(0, 0), tab, idx, outs[k], n, iters); printf("out %d at %p: %.4f ms\n", k, (void*)outs[k], ms[k]); }
  int fast = (int)(std::min_element(ms.begin(), ms.end()) - ms.begin());
  int slow = (int)(std::max_element(ms.begin(), ms.end()) - ms.begin());
  printf("classes: fast = out %d (%.4f ms)  slow = out %d (%.4f ms)  contig %s\n", fast, ms[fast], slow, ms[slow], contig ? "allocated" : "NOT available");
  struct cls { const char* name; int tag; char* buf; };
  std::vector<cls> classes = {{"fast", 1, outs[fast]}, {"slow", 2, outs[slow]}};
  if (contig) classes.push_back({"contig", 3, contig});
  const char* vnames[] = {"product", "tiles_permuted", "row_rotation", "odd_tiles_backwards", "8_steps_in_flight", "plain_stores", "16row_tiles_permuted"};
  printf("== variants x classes (ms), two rounds\n");
  for (int round = 0; round < 2; round++)
    for (int v = 0; v < 7; v++) {
      printf("round %d  %-22s", round, vnames[v]);
      for (auto& c : classes) printf("  %s %.4f", c.name, time_gather(pick(c.tag, v), tab, idx, c.buf, n, iters));
      printf("\n");
    }
  printf("== streaming copy table[0 : 5.12 GB] -> buffer, and pure fill (ms)\n");
  for (auto& c : classes) {
    float t[2];
    for (int which = 0; which < 2; which++) {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; i++) {
        if (which == 0) {
          if (c.tag == 1) hipLaunchKernelGGL(stream_copy<1>, dim3(8192), dim3(256), 0, 0, (const u32x4*)tab, (u32x4*)c.buf, n * 32);
          else if (c.tag == 2) hipLaunchKernelGGL(stream_copy<2>, dim3(8192), dim3(256), 0, 0, (const u32x4*)tab, (u32x4*)c.buf, n * 32);
          else hipLaunchKernelGGL(stream_copy<3>, dim3(8192), dim3(256), 0, 0, (const u32x4*)tab, (u32x4*)c.buf, n * 32);
        } else {
          if (c.tag == 1) hipLaunchKernelGGL(stream_fill<1>, dim3(8192), dim3(256), 0, 0, (u32x4*)c.buf, n * 32);
          else if (c.tag == 2) hipLaunchKernelGGL(stream_fill<2>, dim3(8192), dim3(256), 0, 0, (u32x4*)c.buf, n * 32);
          else hipLaunchKernelGGL(stream_fill<3>, dim3(8192), dim3(256), 0, 0, (u32x4*)c.buf, n * 32);
        }
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&t[which], e0, e1)); t[which] /= iters;
    }
    printf("%-7s copy %.4f ms (%.0f GB/s r+w)   fill %.4f ms (%.0f GB/s)\n", c.name, t[0], 2.0 * out_bytes / t[0] / 1e6, t[1], out_bytes / t[1] / 1e6);
  }

  // ---- the shape of the write front
  {
    unsigned long long* counter; CK(hipMalloc(&counter, 8));
    struct cfg { const char* name; int tr, mode, grid; };
    const cfg cfgs[] = {{"persistent_tr64_g8192", 64, 0, 8192}, {"persistent_tr64_g2048", 64, 0, 2048}, {"persistent_tr16_g2048", 16, 0, 2048},
                        {"inorder_tr64", 64, 1, 0}, {"inorder_tr32", 32, 1, 0}, {"inorder_tr16", 16, 1, 0}, {"inorder_tr8", 8, 1, 0},
                        {"atomic_tr64_g2048", 64, 2, 2048}};
    printf("== write-front shapes x classes (ms), two rounds\n");
    for (int round = 0; round < 2; round++)
      for (const cfg& c : cfgs) {
        printf("round %d  %-24s", round, c.name);
        for (auto& cl : classes) {
          const int64_t tiles = (n + c.tr - 1) / c.tr;
          const int grid = c.mode == 1 ? (int)((tiles + 3) / 4) : c.grid;
          auto launch = [&]() {
            if (c.mode == 2) CK(hipMemsetAsync(counter, 0, 8, 0));
#define GF(TAGV, TRV, MODEV) hipLaunchKernelGGL((gather_front<TAGV, TRV, MODEV>), dim3(grid), dim3(256), 0, 0, tab, idx, cl.buf, n, counter)
#define GF_TAG(TRV, MODEV) do { if (cl.tag == 1) GF(1, TRV, MODEV); else if (cl.tag == 2) GF(2, TRV, MODEV); else GF(3, TRV, MODEV); } while (0)
#define GF_TR(MODEV) do { if (c.tr == 64) GF_TAG(64, MODEV); else if (c.tr == 32) GF_TAG(32, MODEV); else if (c.tr == 16) GF_TAG(16, MODEV); else GF_TAG(8, MODEV); } while (0)
            if (c.mode == 0) GF_TR(0); else if (c.mode == 1) GF_TR(1); else GF_TR(2);
          };
          launch();
          CK(hipEventRecord(e0, 0));
          for (int i = 0; i < iters; i++) launch();
          CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
          float t; CK(hipEventElapsedTime(&t, e0, e1));
          printf("  %s %.4f", cl.name, t / iters);
        }
        printf("\n");
      }
    printf("== fill: grid-stride (8192 blocks) vs block-contiguous in-order pieces (ms)\n");
    for (auto& cl : classes) {
      float t[2];
      for (int mode = 0; mode < 2; mode++) {
        const int grid = mode == 0 ? 8192 : (int)((n * 32 + 1023) / 1024);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; i++) {
          if (mode == 0) { if (cl.tag == 1) hipLaunchKernelGGL((fill_front<1, 0>), dim3(grid), dim3(256), 0, 0, (u32x4*)cl.buf, n * 32); else if (cl.tag == 2) hipLaunchKernelGGL((fill_front<2, 0>), dim3(grid), dim3(256), 0, 0, (u32x4*)cl.buf, n * 32); else hipLaunchKernelGGL((fill_front<3, 0>), dim3(grid), dim3(256), 0, 0, (u32x4*)cl.buf, n * 32); }
          else { if (cl.tag == 1) hipLaunchKernelGGL((fill_front<1, 1>), dim3(grid), dim3(256), 0, 0, (u32x4*)cl.buf, n * 32); else if (cl.tag == 2) hipLaunchKernelGGL((fill_front<2, 1>), dim3(grid), dim3(256), 0, 0, (u32x4*)cl.buf, n * 32); else hipLaunchKernelGGL((fill_front<3, 1>), dim3(grid), dim3(256), 0, 0, (u32x4*)cl.buf, n * 32); }
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t[mode], e0, e1)); t[mode] /= iters;
      }
      printf("%-7s grid-stride %.4f ms   in-order pieces %.4f ms\n", cl.name, t[0], t[1]);
    }
  }

  {
    printf("== in-order shapes x classes (ms), two rounds  [gather unless marked scatter]\n");
    struct shp { const char* name; int id; };
    const shp shps[] = {{"tr8_b256", 0}, {"tr16_b256", 1}, {"tr4_b256", 2}, {"tr8_b64", 3}, {"tr8_b128", 4}, {"tr8_b512", 5}, {"tr8_b1024", 6},
                        {"tr8_b256_plainload", 7}, {"tr32_b256", 8}, {"scatter_tr8_b256", 9}, {"scatter_tr16_b256", 10}, {"scatter_persistent_product_like", 11}};
    for (int round = 0; round < 2; round++)
      for (const shp& sh : shps) {
        printf("round %d  %-32s", round, sh.name);
        for (auto& cl : classes) {
          auto launch = [&]() {
#define RI(TAGV, TRV, BLK, SC, NTL) hipLaunchKernelGGL((rows_inorder<TAGV, TRV, BLK, SC, NTL>), dim3((unsigned)((((n + TRV - 1) / TRV) * 64 + BLK - 1) / BLK)), dim3(BLK), 0, 0, tab, idx, cl.buf, n)
#define RI_TAG(TRV, BLK, SC, NTL) do { if (cl.tag == 1) RI(1, TRV, BLK, SC, NTL); else if (cl.tag == 2) RI(2, TRV, BLK, SC, NTL); else RI(3, TRV, BLK, SC, NTL); } while (0)
            switch (sh.id) {
              case 0: RI_TAG(8, 256, false, true); break;
              case 1: RI_TAG(16, 256, false, true); break;
              case 2: RI_TAG(4, 256, false, true); break;
              case 3: RI_TAG(8, 64, false, true); break;
              case 4: RI_TAG(8, 128, false, true); break;
              case 5: RI_TAG(8, 512, false, true); break;
              case 6: RI_TAG(8, 1024, false, true); break;
              case 7: RI_TAG(8, 256, false, false); break;
              case 8: RI_TAG(32, 256, false, true); break;
              case 9: RI_TAG(8, 256, true, true); break;
              case 10: RI_TAG(16, 256, true, true); break;
              default: break;
            }
          };
          if (sh.id == 11) { printf("  %s n/a", cl.name); continue; }
          launch();
          CK(hipEventRecord(e0, 0));
          for (int i = 0; i < iters; i++) launch();
          CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
          float t; CK(hipEventElapsedTime(&t, e0, e1));
          printf("  %s %.4f", cl.name, t / iters);
        }
        printf("\n");
      }
  }
  // the same on every probe buffer: does a plain copy see the gather's per-buffer levels?
  printf("== streaming copy per probe buffer (ms):");
  for (int k = 0; k < K; k++) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(stream_copy<0>, dim3(8192), dim3(256), 0, 0, (const u32x4*)tab, (u32x4*)outs[k], n * 32);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    printf(" %.4f", t / iters);
  }
  printf("\n== gather probe again:");
  for (int k = 0; k < K; k++) printf(" %.4f", time_gather(pick(0, 0), tab, idx, outs[k], n, iters));
  printf("\n");
  return 0;
}
