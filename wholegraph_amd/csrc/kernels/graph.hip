// wholegraph_amd — neighbour sampling and small graph utilities (gfx950 HIP).
//
// Reference behaviour (cpp/src/wholegraph_ops/unweighted_sample_without_replacement_func.cuh:40-300,
// cpp/src/graph_ops/append_unique_func.cuh:212-353, csr_add_self_loop_func.cuh:24-58):
//   * per center node, min(degree, M) distinct neighbours. For degree > M the reference draws r[i] =
//     rand_i % (N - i), i < M, from per-"thread" PCG streams and resolves the sequential index-sampling recurrence
//         Q = iota(N);  a[i] = Q[r[i]];  Q[r[i]] = Q[N - 1 - i]
//     (its own host statement: tests/wholegraph_ops/graph_sampling_test_utils.cu:306-321) with a block radix sort
//     + pointer-jumping chain in shared memory; for M > 1024 it switches to a reservoir with atomicMax.
//   * the random stream of draw i is fixed by the reference's LAUNCH geometry: virtual thread j = i % T of block b
//     uses PCG stream b*T + j and its (i / T)-th draw (T, items per thread from max_sample_count).
// MI355X design: ONE WAVE PER CENTER NODE, no block-wide sort. The recurrence touches at most 2M positions of Q, so Q
// is kept as a sparse (position -> value) list in LDS and each of the M sequential steps looks its two positions up
// with a 64-lane parallel scan + ballot (M = 30 fan-out: 30 steps x 1 scan). Draws are produced in parallel, each
// lane building the PCG stream of the virtual thread that owns its draw. Results are a pure function of
// (seed, center index, M, CSR row) and identical to the recurrence above; no atomics, no sort.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "../knobs.hpp"
#include "../backend.hpp"
#include "../pcg.hpp"

#include <mutex>
#include <unordered_map>

namespace wm {
namespace {

constexpr int kBlock        = 256;
constexpr int kWavesPerBlk  = kBlock / 64;
constexpr int kMaxSparse    = 1024;
constexpr int kMaxWeighted  = 8192;  // weighted selection: candidate list of 2 M composite keys in LDS (<= 128 KiB)  // M <= 1024 on the sparse-recurrence path (the reference's small-sample limit)

struct gref_view {
  char* base;
  char* const* rank_ptrs;
  const size_t* rank_offsets;
  size_t chunk_stride;
  int world_size;
  int same_chunk;
};

inline gref_view make_view(const wholememory_gref_t& g)
{
  gref_view v{};
  v.chunk_stride = g.stride;
  v.world_size   = g.world_size;
  v.same_chunk   = g.same_chunk ? 1 : 0;
  if (g.stride == 0) {
    v.base = static_cast<char*>(g.pointer);
  } else {
    v.rank_ptrs    = static_cast<char* const*>(g.pointer);
    v.rank_offsets = g.rank_memory_offsets;
  }
  return v;
}

template <typename T>
__device__ __forceinline__ T gref_load(const gref_view& v, int64_t elem_index)
{
  const size_t off = static_cast<size_t>(elem_index) * sizeof(T);
  if (v.chunk_stride == 0) return *reinterpret_cast<const T*>(v.base + off);
  int rank;
  size_t start;
  if (v.same_chunk) {
    rank  = static_cast<int>(off / v.chunk_stride);
    start = static_cast<size_t>(rank) * v.chunk_stride;
  } else {
    rank = 0;
    for (int r = 1; r < v.world_size; r++)
      if (off >= v.rank_offsets[r]) rank = r;
    start = v.rank_offsets[rank];
  }
  return *reinterpret_cast<const T*>(v.rank_ptrs[rank] + (off - start));
}

// ------------------------------------------------------------------------------------------------ single-launch scan
// Exclusive int32 scan of v[i] = fn(i), i in [0, n), in ONE launch with no workspace to prepare (round 4; rocPRIM's look-back
// scan is two launches — state initialisation + scan — and a separate kernel has to materialise its input first: on the C5
// step that was 6 of 20 launches, each at the ~5 us floor of a tiny launch). Decoupled look-back over tiles of
// kChainTile values:
//  * tiles are dealt to blocks STATICALLY — block b takes tiles b, b + G, b + 2 G ... of a grid of G <= 1 x the CU count
//    blocks, all of which are resident together (256 threads, < 64 VGPRs: a CU holds eight of them), so a block only ever
//    waits for tiles whose blocks are running. (The first version took tile numbers from an atomic ticket, which needs no
//    residency argument but serialises: ~27 ns per ticket at the memory-side atomic unit — 100 us for the 3720 tiles of the
//    952 k-value scan with 256-value tiles, 6 us of the 25 us with 4096-value tiles. profiles/r04_c5_flows.txt);
//  * the values of a thread (1, 4 or 16) are evaluated before anything else (fn may be a chain of random loads: they are all in flight
//    together), then thread / wave / block scans, the tile's aggregate is published, wave 0 looks back over the published
//    words of its predecessors (status in the top two bits: 1 = aggregate, 2 = inclusive prefix), publishes the inclusive
//    prefix, and the block writes its outputs. A status word carries its whole message (flag + value in one 64-bit atomic),
//    so the accesses are RELAXED agent-scope atomics: an acquire / release pair at agent scope is an L2 invalidate / write-back
//    on this part (the XCDs' L2s are not coherent with each other) — with those the 465-tile scan of the C5 hop took 153 us;
//  * the state (one word per tile behind a 16-byte header) is caller-provided scratch that must read ALL-ONES on entry — every
//    word is stored complemented, so that the 0xFF fill which empties append_unique's hash table initialises a scan state
//    lying next to it in the same stroke — and is left used: the caller fills it again before the next scan (the chain of
//    hops fills all its scan states with one kernel, the single-hop calls fill their own). Nothing is allocated and nothing
//    on the host waits: the launch can be captured into a hipGraph and replayed.
// `tail(total)` runs once, on the thread that owns value n - 1 (total = the sum of all n values): the place for what used
// to be a one-thread publishing kernel.
constexpr int kChainThreads = 256;   // values per tile: 256 x ITEMS, ITEMS = 1 / 4 / 16 by the size of the scan (chain_scan)
constexpr int kChainMaxTiles = 16384;   // 67 M values at 16 per thread; bigger scans take rocPRIM
struct chain_state {   // (stored complemented: all-ones = no status)
  unsigned int pad[4];
  unsigned long long status[1];   // one word per tile
};
inline size_t chain_state_bytes(int64_t n) { return 16 + 8 * static_cast<size_t>((n + 255) / 256) + 16; }   // room for the smallest tile size
struct no_tail {
  __device__ void operator()(int) const {}
};

// TICKETS: a block takes the number of its next tile from a counter in the state header (pad[0], all-ones like the rest: the
// k-th atomicSub returns ~k), so a tile only ever waits for tiles whose blocks already RUN — forward progress needs no
// assumption about how many blocks the device holds at once (a CU-masked stream, a persistent kernel such as RCCL's or the
// optimizer's long-run side holding CUs: advisor, round 4). false = round 4's static mapping (tile = block + k x grid), which
// completes only if all blocks of the grid are resident together; kept for A/B (WM_SCAN_STATIC=1).
template <typename Fn, typename Tail, int kChainItems, bool TICKETS>
__global__ __launch_bounds__(kChainThreads) void chain_scan_kernel(Fn fn, int n, int* out, chain_state* st, Tail tail)
{
  constexpr int kChainTile = kChainThreads * kChainItems;
  __shared__ int s_excl;
  __shared__ int s_tile;
  __shared__ int s_wave[kChainThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n_tiles = (n + kChainTile - 1) / kChainTile;
  // the functor reads its device-side counts ONCE, here; `live` false = every value is 0 and nothing may be read
  const bool live = fn.prepare();
  for (int round = 0;; round++) {
  int tile;
  if (TICKETS) {
    if (tid == 0) s_tile = static_cast<int>(~atomicSub(&st->pad[0], 1u));
    __syncthreads();
    tile = s_tile;
  } else {
    tile = blockIdx.x + round * gridDim.x;
  }
  if (tile >= n_tiles) break;
  const int i0   = tile * kChainTile + tid * kChainItems;
  int v[kChainItems];
#pragma unroll
  for (int k = 0; k < kChainItems; k++) v[k] = 0;
  if (live) {   // ONE guard around the thread's values: inside it the index is clamped and the value masked, nothing is predicated,
                // so the load chains of all values are in flight together (a guard per value: one waited load after the other)
#pragma unroll
    for (int k = 0; k < kChainItems; k++) {
      const int x = fn(min(i0 + k, n - 1));
      v[k]        = x & -static_cast<int>(i0 + k < n);   // (arithmetic mask, not a select: hipcc sinks a load into the branch that needs it)
    }
  }
  int sum = 0;
#pragma unroll
  for (int k = 0; k < kChainItems; k++) {   // exclusive within the thread
    const int x = v[k];
    v[k]        = sum;
    sum += x;
  }
  int incl = sum;   // inclusive over the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(incl, d, 64);
    if (lane >= d) incl += y;
  }
  if (lane == 63) s_wave[wv] = incl;
  __syncthreads();
  int wave_off = 0, tile_total = 0;
#pragma unroll
  for (int w = 0; w < kChainThreads / 64; w++) {
    if (w < wv) wave_off += s_wave[w];
    tile_total += s_wave[w];
  }
  if (wv == 0) {
    constexpr unsigned long long kAggregate = 1ull << 62, kInclusive = 2ull << 62;
    if (lane == 0 && tile > 0)
      __hip_atomic_store(&st->status[tile], ~(kAggregate | static_cast<uint32_t>(tile_total)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int excl = 0;
    for (int j = tile - 1; j >= 0; j -= 64) {   // wave-uniform; lane l looks at tile j - l
      const int idx        = j - lane;
      unsigned long long w = kInclusive;        // "in front of tile 0": inclusive prefix 0
      if (idx >= 0) {
        do {
          w = ~__hip_atomic_load(&st->status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while ((w >> 62) == 0);
      }
      const unsigned long long incl_mask = __ballot((w >> 62) == 2);
      const int first = incl_mask != 0 ? __ffsll(static_cast<long long>(incl_mask)) - 1 : 64;   // nearest tile with a full prefix
      int part        = lane <= first ? static_cast<int>(static_cast<uint32_t>(w)) : 0;
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
      excl += part;
      if (incl_mask != 0) break;
    }
    if (lane == 0) {
      __hip_atomic_store(&st->status[tile], ~(kInclusive | static_cast<uint32_t>(excl + tile_total)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_excl = excl;
    }
  }
  __syncthreads();
  const int base = s_excl + wave_off + incl - sum;   // exclusive prefix of this thread's first value
  if (kChainItems >= 4 && i0 + kChainItems <= n && (reinterpret_cast<uintptr_t>(out + i0) & 15) == 0) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int k = 0; k + 3 < kChainItems; k += 4) {
      i32x4 a;
      a.x = base + v[k], a.y = base + v[k + 1], a.z = base + v[k + 2], a.w = base + v[k + 3];
      *reinterpret_cast<i32x4*>(out + i0 + k) = a;
    }
  } else {
#pragma unroll
    for (int k = 0; k < kChainItems; k++)
      if (i0 + k < n) out[i0 + k] = base + v[k];
  }
  if (i0 <= n - 1 && n - 1 < i0 + kChainItems) tail(base + sum);
  __syncthreads();   // s_excl / s_wave / s_tile are reused by the block's next tile
  }
}

// 0xFF fill as a KERNEL (16-byte pieces; ptr and bytes multiples of 16). Not hipMemsetAsync: inside a captured hipGraph the
// memset command of a chain was seen to run out of order with the scan kernels that depend on it (replays with other work
// between them faulted; experiments/r04_capture_bisect.py) — a kernel node is ordered like its neighbours.
__global__ void fill_ff_kernel(void* ptr, size_t vecs)
{
  typedef uint32_t fill4 __attribute__((ext_vector_type(4)));
  const fill4 ones    = {~0u, ~0u, ~0u, ~0u};
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < vecs; i += stride) static_cast<fill4*>(ptr)[i] = ones;
}
inline int fill_ff(void* ptr, size_t bytes, hipStream_t stream)
{
  if (bytes == 0) return 0;
  if ((reinterpret_cast<uintptr_t>(ptr) | bytes) & 15) return hipMemsetAsync(ptr, 0xFF, bytes, stream) == hipSuccess ? 0 : -2;
  const size_t vecs = bytes / 16;
  const int blocks  = static_cast<int>(std::min<size_t>((vecs + 255) / 256, 4096));
  hipLaunchKernelGGL(fill_ff_kernel, dim3(blocks), dim3(256), 0, stream, ptr, vecs);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

inline bool chain_scan_fits(int64_t n) { return n > 0 && n <= static_cast<int64_t>(kChainMaxTiles) * kChainThreads * 16; }

// Tile size by the size of the scan: a small scan wants MANY small tiles (fn is a chain of random loads: 31 k values in 8
// tiles of 4096 keep 8 CUs busy and take 18 us, in 124 tiles of 256 they spread over the chip), a big one fewer, larger tiles
// (every 64 predecessors are one look-back round trip of ~1 us for the last tile). WM_SCAN_ITEMS=1|4|16 forces (A/B).
template <typename Fn, typename Tail>
int chain_scan(Fn fn, int n, int* out, Tail tail, void* state, hipStream_t stream)
{
  chain_state* st = static_cast<chain_state*>(state);   // chain_state_bytes(n) bytes, all-ones
  int items = n <= (64 << 10) ? 1 : n <= (512 << 10) ? 4 : 16;
  if (const char* e = WM_KNOB("WM_SCAN_ITEMS")) {
    const int v = atoi(e);
    if ((v == 1 || v == 4 || v == 16) && (static_cast<int64_t>(n) + 256 * v - 1) / (256 * v) <= kChainMaxTiles) items = v;
  }
  const int tile = kChainThreads * items;
  // one block per CU of the CURRENT device (per device ordinal: a process may drive several)
  static int cus_of[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (cus_of[dev] == 0) {
    hipDeviceProp_t prop;
    cus_of[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 64;
  }
  // One tile per block (tile = block index) while the scan has at most 8 tiles per CU: a block then waits only for blocks
  // with a smaller index, and workgroups start in index order on every XCD, so the tile with the smallest unfinished index
  // always runs or is next in line for a slot held by finished-or-running smaller ones — no assumption about how many blocks
  // are resident together. Bigger scans loop over tiles, and a looping block must not wait for a tile nobody has started:
  // there the tile numbers come from the ticket counter (see the kernel). WM_SCAN_STATIC=1: round 4's mapping (A/B only).
  const int n_tiles = (n + tile - 1) / tile;
  const bool one_to_one = n_tiles <= 8 * cus_of[dev];
  const bool static_ab = WM_AB_KNOB("WM_SCAN_STATIC") != nullptr && WM_AB_KNOB("WM_SCAN_STATIC")[0] == '1';
  const dim3 grid(one_to_one && !static_ab ? n_tiles : std::min(n_tiles, cus_of[dev])), block(kChainThreads);
  const bool tickets = !one_to_one && !static_ab;
#define WM_CHAIN(ITEMS)                                                                                              \
  do {                                                                                                               \
    if (tickets) hipLaunchKernelGGL((chain_scan_kernel<Fn, Tail, ITEMS, true>), grid, block, 0, stream, fn, n, out, st, tail); \
    else hipLaunchKernelGGL((chain_scan_kernel<Fn, Tail, ITEMS, false>), grid, block, 0, stream, fn, n, out, st, tail);        \
  } while (0)
  if (items == 1) WM_CHAIN(1);
  else if (items == 4) WM_CHAIN(4);
  else WM_CHAIN(16);
#undef WM_CHAIN
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ------------------------------------------------------------------------------------------------ counts
// ids[2i] = center i, ids[2i+1] = center i + 1: the row_ptr entries a DISTRIBUTED CSR has to fetch per center node
template <typename IdT>
__global__ void pair_ids_kernel(const IdT* centers, int n, int64_t* ids)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t nid = static_cast<int64_t>(centers[i]);
  ids[2 * i]        = nid;
  ids[2 * i + 1]    = nid + 1;
}

template <typename IdT>
__global__ void sample_count_kernel(gref_view row_ptr, int64_t row_off, const int64_t* pairs, const IdT* centers, int n,
                                    const int* n_dev, int max_sample, int* counts)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  const int used = n_dev != nullptr ? min(n, *n_dev) : n;   // arrays sized for n, `used` centres in them
  if (i >= used) {
    counts[i] = 0;  // the scan runs over n + 1 entries (reference :334-338)
    return;
  }
  int64_t s, e;
  if (pairs != nullptr) {
    s = pairs[2 * i], e = pairs[2 * i + 1];
  } else {
    const int64_t nid = static_cast<int64_t>(centers[i]);
    s                 = gref_load<int64_t>(row_ptr, row_off + nid);
    e                 = gref_load<int64_t>(row_ptr, row_off + nid + 1);
  }
  int deg           = static_cast<int>(e - s);
  if (max_sample > 0) deg = min(deg, max_sample);  // <= 0 means "all neighbours"
  counts[i] = deg;
}

// ------------------------------------------------------------------------------------------------ sampling
struct sample_params {
  gref_view row_ptr, col_ptr;
  int64_t row_off, col_off;  // storage offsets (elements)
  const int64_t* pairs;  // optional [2n]: (row_ptr[c], row_ptr[c+1]) already fetched (DISTRIBUTED CSR); then
                         // out_ids must be nullptr and the kernels only emit positions (edge ids)
  const void* centers;
  int n_center;
  int max_sample;
  uint64_t seed;
  const int* offsets;  // [n + 1]
  void* out_ids;       // ColT, or nullptr
  int* out_lid;        // optional
  int64_t* out_egid;   // optional
  const int* n_center_dev;  // optional: centres in use (<= n_center)
  // side job (wm_sample_args::fill_ff_ptr): 16-byte pieces set to all-ones by the sampling kernel's threads before they sample
  void* fill_ptr;
  size_t fill_vecs;
};
__device__ __forceinline__ void side_fill(const sample_params& p)
{
  if (p.fill_ptr == nullptr) return;
  typedef uint32_t fill4 __attribute__((ext_vector_type(4)));
  const fill4 ones      = {~0u, ~0u, ~0u, ~0u};
  const size_t stride   = static_cast<size_t>(gridDim.x) * blockDim.x;
  fill4* dst            = static_cast<fill4*>(p.fill_ptr);
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < p.fill_vecs; i += stride) dst[i] = ones;
}
__device__ __forceinline__ int centers_in_use(const sample_params& p)
{
  return p.n_center_dev != nullptr ? min(p.n_center, *p.n_center_dev) : p.n_center;
}

template <typename IdT>
__device__ __forceinline__ void row_bounds(const sample_params& p, int center, int64_t* s, int64_t* e)
{
  if (p.pairs != nullptr) {
    *s = p.pairs[2 * center];
    *e = p.pairs[2 * center + 1];
    return;
  }
  const int64_t nid = static_cast<int64_t>(static_cast<const IdT*>(p.centers)[center]);
  *s                = gref_load<int64_t>(p.row_ptr, p.row_off + nid);
  *e                = gref_load<int64_t>(p.row_ptr, p.row_off + nid + 1);
}

template <typename IdT, typename ColT>
__global__ __launch_bounds__(kBlock) void sample_sparse_kernel(sample_params p)
{
  // per wave: draws r[], sampled positions a[] and the sparse image of Q
  extern __shared__ int lds[];
  const int M       = p.max_sample;
  const int wave_in = threadIdx.x >> 6;
  const int lane    = threadIdx.x & 63;
  int* r_s          = lds + wave_in * (4 * M);
  int* a_s          = r_s + M;
  int* qpos         = a_s + M;
  int* qval         = qpos + M;
  const int center  = blockIdx.x * kWavesPerBlk + wave_in;
  if (center >= centers_in_use(p)) return;
  ColT* out          = static_cast<ColT*>(p.out_ids);
  int64_t s, e;
  row_bounds<IdT>(p, center, &s, &e);
  const int N        = static_cast<int>(e - s);
  if (N <= 0) return;
  const int off = p.offsets[center];
  if (M <= 0 || N <= M) {  // every neighbour (reference sample_all_kernel / the `neighbor_count <= max` branches)
    for (int i = lane; i < N; i += 64) {
      if (out) out[off + i] = gref_load<ColT>(p.col_ptr, p.col_off + s + i);
      if (p.out_lid) p.out_lid[off + i] = center;
      if (p.out_egid) p.out_egid[off + i] = s + i;
    }
    return;
  }
  // draws: id i belongs to virtual thread j = i % T (stream center*T + j), its (i / T)-th draw
  const sample_geometry g = sample_geometry_for(M);
  for (int i = lane; i < M; i += 64) {
    const int j = i % g.threads, k = i / g.threads;
    pcg32 rng(p.seed, 0, static_cast<uint64_t>(center) * g.threads + j);
    int32_t v = 0;
    for (int q = 0; q <= k; q++) v = rng.next_i32();
    r_s[i] = v % (N - i);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  // the recurrence, sequential in i, each lookup a parallel scan of the sparse list
  int cnt = 0;  // wave-uniform
  auto lookup = [&](int pos, int* idx_out) -> int {  // value of Q[pos]; *idx_out = list slot or -1
    int found = -1;
    for (int b = 0; b < cnt; b += 64) {
      const int t         = b + lane;
      const bool hit      = t < cnt && qpos[t] == pos;
      const uint64_t mask = __ballot(hit);
      if (mask) {
        found = b + (__ffsll(static_cast<long long>(mask)) - 1);
        break;
      }
    }
    *idx_out = found;
    return found >= 0 ? qval[found] : pos;
  };
  for (int i = 0; i < M; i++) {
    const int x = r_s[i];
    int ix, iy;
    const int vx = lookup(x, &ix);
    const int vy = lookup(N - 1 - i, &iy);
    if (lane == 0) {
      a_s[i] = vx;
      if (ix >= 0) {
        qval[ix] = vy;
      } else {
        qpos[cnt] = x;
        qval[cnt] = vy;
      }
    }
    if (ix < 0) cnt++;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  for (int i = lane; i < M; i += 64) {
    const int ai = a_s[i];
    if (out) out[off + i] = gref_load<ColT>(p.col_ptr, p.col_off + s + ai);
    if (p.out_lid) p.out_lid[off + i] = center;
    if (p.out_egid) p.out_egid[off + i] = s + ai;
  }
}

// The same recurrence for max_sample <= 64 (the fan-outs GNN samplers use: 5 ... 30) with NOTHING in LDS: lane i holds draw
// r[i], sampled position a[i] and entry i of the sparse image of Q (position, value) in registers; a lookup is one ballot
// and one v_readlane, an update one predicated move. The LDS version pays two LDS round trips, a fence and a wave barrier
// per step of the sequential recurrence (~70 us for the 24 k centres of a papers100M-shaped second hop).
// G = lanes per centre: 64 (one centre per wave, max_sample <= 64) or 32 (TWO centres per wave, max_sample <= 32 — the fan-outs
// GNN samplers use). With one centre per wave the 31.7 k short waves of a papers100M-shaped second hop took 24 us whatever was
// done to their load chain or their vector instruction count (round 3, profiles/r03_sample_pmc.txt); two centres per wave
// halve the waves and the per-centre instruction stream: 19.5 us.
template <typename IdT, typename ColT, int G>
__global__ __launch_bounds__(kBlock) void sample_small_kernel(sample_params p)
{
  constexpr int kPerWave = 64 / G;
  side_fill(p);
  const int M      = p.max_sample;   // 1 ... G
  const int lane   = threadIdx.x & 63;
  const int gl     = lane & (G - 1);          // lane within its centre's group
  const int gbase  = lane & ~(G - 1);         // first lane of the group
  const int wave   = blockIdx.x * kWavesPerBlk + (threadIdx.x >> 6);
  const int center = wave * kPerWave + lane / G;
  // every lane of a group reads its centre's bounds itself (one address per group); a group past the end, an isolated node
  // or a node with N <= M neighbours just sits out the parts it does not need — no early return, the groups of a wave
  // share the shuffles below
  const bool live = center < centers_in_use(p);
  int64_t s = 0, e = 0;
  if (live) row_bounds<IdT>(p, center, &s, &e);
  const int N     = static_cast<int>(e - s);
  const int off   = live && N > 0 ? p.offsets[center] : 0;
  ColT* out       = static_cast<ColT*>(p.out_ids);
  const bool all  = live && N > 0 && N <= M;      // every neighbour
  const bool draw = live && N > M;                // M distinct positions out of N
  if (!__any(live && N > 0)) return;
  // draw i belongs to virtual thread j = i % T (stream center * T + j), its (i / T)-th draw — as in sample_sparse_kernel
  const sample_geometry g = sample_geometry_for(M);
  int my_r = 0;
  if (draw && gl < M) {
    const int j = gl % g.threads, k = gl / g.threads;
    pcg32 rng(p.seed, 0, static_cast<uint64_t>(center) * g.threads + j);
    int32_t v = 0;
    for (int q = 0; q <= k; q++) v = rng.next_i32();
    my_r = v % (N - gl);
  }
  // The recurrence  for i = 0 .. M-1:  x = r[i], y = N-1-i;  a[i] = Q[x];  Q[x] = Q[y]   (Q = identity at the start)
  // is sequential as written — one lookup of the sparse image of Q per step, ~22 dependent vector / scalar instructions each
  // — but its result has a closed form the lanes can evaluate side by side. Step k ASSIGNS key x_k the value
  // v_k = Q_k[y_k] (Q_k = the state before step k), so
  //     Q_i[p] = v_k for the LARGEST k < i with x_k == p, else p,
  //     a[i]   = Q_i[x_i],        v_k = Q_k[y_k] = v_m for the largest m < k with x_m == y_k, else y_k.
  // Lane i finds its two predecessors px = max{k < i : x_k == x_i} and py = max{k < i : x_k == y_i} in one pass over k (x_k
  // broadcast inside the group); the v-chain k -> py(k) -> py(py(k)) ... ends in a step that moved an untouched position,
  // whose value is that position: v_k = y_root(k), the root found by pointer jumping (log2 G doublings cover G steps).
  // Same integers as the sequential loop, a third of the instructions.
  const int my_x = my_r, my_y = N - 1 - gl;
  int px = -1, py = -1;
  for (int k = 0; k < M; k++) {
    const int xk      = G == 64 ? __builtin_amdgcn_readlane(my_x, k) : __shfl(my_x, gbase + k, 64);
    const bool before = k < gl;
    if (before && xk == my_x) px = k;
    if (before && xk == my_y) py = k;
  }
  int root = py < 0 ? gl : py;                       // (group-relative step numbers)
#pragma unroll
  for (int it = 0; it < (G == 64 ? 6 : 5); it++) root = __shfl(root, gbase + root, 64);
  const int root_of_px = __shfl(root, gbase + (px < 0 ? gl : px), 64);
  const int my_a       = px < 0 ? my_x : N - 1 - root_of_px;
  const int pos        = all ? gl : my_a;            // neighbour position this lane emits
  if ((all && gl < N) || (draw && gl < M)) {
    if (out) out[off + gl] = gref_load<ColT>(p.col_ptr, p.col_off + s + pos);
    if (p.out_lid) p.out_lid[off + gl] = center;
    if (p.out_egid) p.out_egid[off + gl] = s + pos;
  }
}

// M > 1024 (reference large_sample_kernel :62-130): reservoir of M slots; candidate idx >= M replaces slot
// rand % (idx + 1) when that is < M, the LARGEST idx wins a slot. 32 virtual threads per center node (streams
// center*32 + t), thread t visits idx = M + t, M + t + 32, ... with consecutive draws. One wave per center node,
// lanes 0..31 play the virtual threads; like the reference, the slots live in the (>= 4-byte) output elements.
template <typename IdT, typename ColT>
__global__ __launch_bounds__(64) void sample_large_kernel(sample_params p)
{
  const int center = blockIdx.x;
  if (center >= centers_in_use(p)) return;   // (block-uniform)
  const int lane   = threadIdx.x;
  const int M      = p.max_sample;
  ColT* out          = static_cast<ColT*>(p.out_ids);
  int64_t s, e;
  row_bounds<IdT>(p, center, &s, &e);
  const int N        = static_cast<int>(e - s);
  if (N <= 0) return;
  const int off = p.offsets[center];
  if (N <= M) {
    for (int i = lane; i < N; i += 64) {
      if (out) out[off + i] = gref_load<ColT>(p.col_ptr, p.col_off + s + i);
      if (p.out_lid) p.out_lid[off + i] = center;
      if (p.out_egid) p.out_egid[off + i] = s + i;
    }
    return;
  }
  // the reservoir slots live in the output elements (>= 4 bytes each); in positions-only mode in the edge-id elements
  auto slot = [&](int i) { return out ? reinterpret_cast<int*>(out + off + i) : reinterpret_cast<int*>(p.out_egid + off + i); };
  for (int i = lane; i < M; i += 64) {
    *slot(i) = i;
    if (p.out_lid) p.out_lid[off + i] = center;
  }
  __syncthreads();
  if (lane < 32) {
    pcg32 rng(p.seed, 0, static_cast<uint64_t>(center) * 32 + lane);
    for (int idx = M + lane; idx < N; idx += 32) {
      const int32_t rnd = rng.next_i32() % (idx + 1);
      if (rnd < M) atomicMax(slot(rnd), idx);
    }
  }
  __syncthreads();
  for (int i = lane; i < M; i += 64) {
    const int ai = __hip_atomic_load(slot(i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (out) out[off + i] = gref_load<ColT>(p.col_ptr, p.col_off + s + ai);
    if (p.out_egid) p.out_egid[off + i] = s + ai;
  }
}

template <typename IdT, typename ColT>
int launch_sample(sample_params p, hipStream_t stream)
{
  const bool small = p.max_sample >= 1 && p.max_sample <= 64 && WM_AB_KNOB("WM_SAMPLE_LDS") == nullptr;
  if (p.fill_ptr != nullptr && (p.n_center == 0 || !small)) {   // only the small-sample kernels carry the side job
    if (fill_ff(p.fill_ptr, p.fill_vecs * 16, stream) != 0) return -2;
    p.fill_ptr = nullptr;
  }
  if (p.n_center == 0) return 0;
  if (p.max_sample > kMaxSparse) {
    hipLaunchKernelGGL((sample_large_kernel<IdT, ColT>), dim3(p.n_center), dim3(64), 0, stream, p);
  } else if (p.max_sample >= 1 && p.max_sample <= 32 && WM_AB_KNOB("WM_SAMPLE_LDS") == nullptr && WM_AB_KNOB("WM_SAMPLE_ONE_PER_WAVE") == nullptr) {
    const int per_block = 2 * kWavesPerBlk;   // two centres per wave
    hipLaunchKernelGGL((sample_small_kernel<IdT, ColT, 32>), dim3((p.n_center + per_block - 1) / per_block), dim3(kBlock), 0, stream, p);
  } else if (p.max_sample >= 1 && p.max_sample <= 64 && WM_AB_KNOB("WM_SAMPLE_LDS") == nullptr) {
    const int blocks = (p.n_center + kWavesPerBlk - 1) / kWavesPerBlk;
    hipLaunchKernelGGL((sample_small_kernel<IdT, ColT, 64>), dim3(blocks), dim3(kBlock), 0, stream, p);
  } else {
    const int M       = std::max(p.max_sample, 1);
    const size_t lds  = static_cast<size_t>(kWavesPerBlk) * 4 * M * sizeof(int);
    const int blocks  = (p.n_center + kWavesPerBlk - 1) / kWavesPerBlk;
    hipLaunchKernelGGL((sample_sparse_kernel<IdT, ColT>), dim3(blocks), dim3(kBlock), lds, stream, p);
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ------------------------------------------------------------------------------------------------ weighted sampling
// A-Res (weighted_sample_without_replacement_func.cuh:183-300): every neighbour gets key = log2(u) / weight, the
// max_sample_count largest keys win. Which stream feeds which neighbour follows the reference geometry: virtual thread
// j of a block of `vthreads` (128, or 256 when max_sample_count > 256) visits neighbours j, j + vthreads, ... with
// consecutive keys of stream center * vthreads + j. The reference keeps the winners in raft's warp-sort queues; here
// ONE WAVE per center node keeps a candidate list of composite keys (orderable key bits << 32 | ~neighbour index —
// unique, so ties break towards the smaller index) in LDS, filters new keys against the running M-th largest and
// compacts with an in-LDS bitonic sort whenever the list fills up. Output order: key descending (deterministic; the
// reference leaves the order to its queues and its tests compare per-center sets).
struct weighted_params {
  sample_params sp;
  gref_view weight_ptr;
  int64_t weight_off;
  int vthreads;
  int capacity;  // P: power of two >= 2 * max_sample, >= 128
};

__device__ __forceinline__ void bitonic_sort_desc(uint64_t* a, int P, int lane)
{
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = lane; t < (P >> 1); t += 64) {
        const int i    = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l    = i | j;
        const uint64_t x = a[i], y = a[l];
        const bool desc  = (i & k) == 0;
        if ((x < y) == desc) {
          a[i] = y;
          a[l] = x;
        }
      }
      __syncthreads();
    }
  }
}

template <typename IdT, typename ColT, typename WT>
__global__ __launch_bounds__(64) void sample_weighted_kernel(weighted_params w)
{
  extern __shared__ uint64_t cand[];
  const sample_params& p = w.sp;
  const int center       = blockIdx.x;
  const int lane         = threadIdx.x;
  const int M            = p.max_sample;
  ColT* out              = static_cast<ColT*>(p.out_ids);
  int64_t s, e;
  row_bounds<IdT>(p, center, &s, &e);
  const int N            = static_cast<int>(e - s);
  if (N <= 0) return;
  const int off = p.offsets[center];
  if (M <= 0 || N <= M) {
    for (int i = lane; i < N; i += 64) {
      if (out) out[off + i] = gref_load<ColT>(p.col_ptr, p.col_off + s + i);
      if (p.out_lid) p.out_lid[off + i] = center;
      if (p.out_egid) p.out_egid[off + i] = s + i;
    }
    return;
  }
  const int P          = w.capacity;
  int cnt              = 0;  // wave-uniform
  uint64_t threshold   = 0;  // composite keys are > 0; accept only keys above the current M-th largest
  const uint64_t lt    = (1ull << lane) - 1;
  auto compact = [&]() {
    for (int i = cnt + lane; i < P; i += 64) cand[i] = 0;
    __syncthreads();
    bitonic_sort_desc(cand, P, lane);
    if (cnt >= M) {
      cnt       = M;
      threshold = cand[M - 1];
    }
  };
  for (int vbase = 0; vbase < w.vthreads; vbase += 64) {
    const int vt = vbase + lane;
    pcg32 rng(p.seed, 0, static_cast<uint64_t>(center) * w.vthreads + vt);
    for (int id0 = vbase; id0 < N; id0 += w.vthreads) {
      const int id = id0 + lane;
      bool accept  = false;
      uint64_t comp = 0;
      if (id < N) {
        const float wt  = static_cast<float>(gref_load<WT>(w.weight_ptr, w.weight_off + s + id));
        const float key = weighted_sample_key(rng, wt);
        comp            = (static_cast<uint64_t>(orderable_float(key)) << 32) | (0xffffffffu - static_cast<uint32_t>(id));
        accept          = comp > threshold;
      }
      const uint64_t mask = __ballot(accept);
      if (accept) cand[cnt + __popcll(mask & lt)] = comp;
      cnt += __popcll(mask);
      __syncthreads();
      if (cnt > P - 64) compact();
    }
  }
  compact();
  for (int i = lane; i < M; i += 64) {
    const int ai = static_cast<int>(0xffffffffu - static_cast<uint32_t>(cand[i]));
    if (out) out[off + i] = gref_load<ColT>(p.col_ptr, p.col_off + s + ai);
    if (p.out_lid) p.out_lid[off + i] = center;
    if (p.out_egid) p.out_egid[off + i] = s + ai;
  }
}

template <typename IdT, typename ColT, typename WT>
int launch_weighted(const weighted_params& w, hipStream_t stream)
{
  if (w.sp.n_center == 0) return 0;
  const size_t lds = static_cast<size_t>(w.capacity) * sizeof(uint64_t);
  if (lds > 64 * 1024) {  // above the default dynamic-LDS limit: raise it once per instantiation
    static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&sample_weighted_kernel<IdT, ColT, WT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kMaxWeighted * 8) == hipSuccess;
    if (!ok) return -2;
  }
  hipLaunchKernelGGL((sample_weighted_kernel<IdT, ColT, WT>), dim3(w.sp.n_center), dim3(64), lds, stream, w);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <typename WT>
int dispatch_weighted(const weighted_params& w, bool id32, bool col32, hipStream_t stream)
{
  if (id32 && col32) return launch_weighted<int32_t, int32_t, WT>(w, stream);
  if (id32) return launch_weighted<int32_t, int64_t, WT>(w, stream);
  if (col32) return launch_weighted<int64_t, int32_t, WT>(w, stream);
  return launch_weighted<int64_t, int64_t, WT>(w, stream);
}

// ------------------------------------------------------------------------------------------------ append_unique
// unique = targets (as they stand) ++ the neighbour ids that are not targets, each once, in the order of their FIRST
// occurrence in the neighbour array; mapping[p] = position of neighbours[p] in that array.
// Round 1 sorted (id, position) pairs; on a 950 k-key hop rocPRIM picks its merge sort for that size — a block sort and
// 19 merge passes, ~30 launches per call and 250 us of the C5 step. Here: an open-addressing table in scratch
// (2 slots per key, linear probing) whose slot holds the id and the SMALLEST position it occurs at (targets are positions
// 0 .. nt-1, neighbours nt ..), built with one compare-and-swap + one atomic min per key. A neighbour is "new" when the
// smallest position of its id is its own; an exclusive scan over those flags ranks the new ids in neighbour order. Six
// launches before the host learns the count, two after. Every step is an order-independent reduction (min) or a scan, so
// the result is deterministic and equals the oracle's bit for bit.
template <typename KeyT>
struct au_layout {
  void* scan_state;   // chain_state of the ranking scan: in front of the table, inside the region the 0xFF fill covers
  KeyT* slots;        // cap + 1 ids (all-ones = empty; slot `cap` is reserved for the id that IS all-ones)
  uint32_t* min_pos;  // cap + 1 smallest positions (0xFFFFFFFF = none yet)
  uint32_t* slot_of;  // nt + nn: where each key landed (phase 2 and the flag pass do not probe again)
  int *first_flag, *new_rank;  // nn + 1 each
  void* temp;
  size_t temp_bytes, table_bytes, total;
  uint32_t cap;
};

template <typename KeyT>
au_layout<KeyT> au_plan(void* ws, int nt, int nn)
{
  const size_t n = static_cast<size_t>(nt) + nn;
  auto al        = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  size_t scan_b  = 0;
  (void)rocprim::exclusive_scan(nullptr, scan_b, static_cast<const int*>(nullptr), static_cast<int*>(nullptr), 0,
                                static_cast<size_t>(nn) + 1, rocprim::plus<int>(), nullptr);
  au_layout<KeyT> l;
  size_t cap = 64;
  while (cap < 2 * n) cap <<= 1;   // n < 2^31: cap <= 2^32 would not fit uint32 — n is an int sum the host keeps below 2^30
  l.cap    = static_cast<uint32_t>(cap);
  char* p  = static_cast<char*>(ws);
  size_t o = 0;
  // [scan state | slots | min_pos] are contiguous: one fill of 0xFF bytes initialises the scan state and empties the table
  l.scan_state = p + o, o += al(chain_state_bytes(static_cast<int64_t>(nn) + 1));
  l.slots = reinterpret_cast<KeyT*>(p + o), o += al(sizeof(KeyT) * (cap + 1));
  l.min_pos = reinterpret_cast<uint32_t*>(p + o), o += al(4 * (cap + 1));
  l.table_bytes = o;
  l.slot_of = reinterpret_cast<uint32_t*>(p + o), o += al(4 * std::max<size_t>(n, 1));
  l.first_flag = reinterpret_cast<int*>(p + o), o += al(4 * (static_cast<size_t>(nn) + 1));
  l.new_rank = reinterpret_cast<int*>(p + o), o += al(4 * (static_cast<size_t>(nn) + 1));
  l.temp       = p + o;
  l.temp_bytes = scan_b;
  l.total      = o + al(l.temp_bytes) + 256;
  return l;
}

__device__ __forceinline__ uint32_t au_hash(uint64_t k)
{
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 29;
  return static_cast<uint32_t>(k);
}

__device__ __forceinline__ uint32_t au_cas(uint32_t* a, uint32_t expect, uint32_t v) { return atomicCAS(a, expect, v); }
__device__ __forceinline__ uint64_t au_cas(uint64_t* a, uint64_t expect, uint64_t v)
{
  return atomicCAS(reinterpret_cast<unsigned long long*>(a), static_cast<unsigned long long>(expect), static_cast<unsigned long long>(v));
}

template <typename KeyT>
__global__ __launch_bounds__(kBlock) void au_insert_kernel(const KeyT* targets, int nt, const KeyT* neighbors, int nn,
                                                           const int* nn_dev, KeyT* slots, uint32_t* min_pos,
                                                           uint32_t* slot_of, uint32_t cap, const int* nt_dev, int direct_cas)
{
  // the arrays hold room for nt targets and nn neighbours; nt_use / nn_use of them are in use (device-side counts of a
  // bounded call, else all). Thread i serves array entry i of targets ++ neighbours; the POSITION of a key — what
  // "first occurrence" is decided on — counts the entries in use: targets 0 .. nt_use - 1, neighbours nt_use ...
  const int i      = blockIdx.x * blockDim.x + threadIdx.x;
  const int nt_use = nt_dev != nullptr ? min(nt, *nt_dev) : nt;
  const int nn_use = nn_dev != nullptr ? min(nn, *nn_dev) : nn;
  if (i >= nt + nn) return;
  const bool is_target = i < nt;
  const int entry      = is_target ? i : i - nt;
  if (entry >= (is_target ? nt_use : nn_use)) return;
  const int pos         = is_target ? entry : nt_use + entry;
  constexpr KeyT kEmpty = ~static_cast<KeyT>(0);
  const KeyT key        = is_target ? targets[entry] : neighbors[entry];
  if constexpr (sizeof(KeyT) == 4) {
    // 32-bit ids: (id << 32 | smallest position) in ONE 64-bit word per slot — the [slots | min_pos] region read as uint64 —
    // so a new id costs one compare-and-swap and a repeated one usually nothing (its position is larger than what is there)
    // instead of a compare-and-swap plus an atomic min on a second array
    unsigned long long* table = reinterpret_cast<unsigned long long*>(slots);
    const unsigned long long mine = (static_cast<unsigned long long>(key) << 32) | static_cast<uint32_t>(pos);
    uint32_t s = au_hash(static_cast<uint64_t>(key)) & (cap - 1);
    for (;;) {
      unsigned long long cur = direct_cas ? ~0ull : table[s];   // direct_cas: no look first — one round trip per new id, not two
      if (cur == ~0ull) {
        cur = atomicCAS(&table[s], ~0ull, mine);
        if (cur == ~0ull) break;                       // the slot is mine
      }
      if (static_cast<uint32_t>(cur >> 32) == key) {   // equal ids: the words compare by position
        if (cur > mine) atomicMin(&table[s], mine);
        break;
      }
      s = (s + 1) & (cap - 1);
    }
    slot_of[i] = s;
    return;
  }
  uint32_t s            = cap;  // the id that looks like "empty" has a slot of its own
  if (key != kEmpty) {
    s = au_hash(static_cast<uint64_t>(key)) & (cap - 1);
    for (;;) {
      KeyT cur = direct_cas ? kEmpty : slots[s];
      if (cur == kEmpty) cur = au_cas(&slots[s], kEmpty, key);   // returns what was there: empty = the slot is mine now
      if (cur == kEmpty || cur == key) break;
      s = (s + 1) & (cap - 1);
    }
  }
  slot_of[i] = s;
  // (a plain look first: a hot id is offered by thousands of positions, most of them larger than what is already there)
  if (min_pos[s] > static_cast<uint32_t>(pos)) atomicMin(&min_pos[s], static_cast<uint32_t>(pos));
}

// The insert, round 5: the keys of a workgroup are first merged in an LDS table of the same shape (id -> smallest position among
// the workgroup's 256 entries); only the entry that holds an id's smallest position in its workgroup goes to the table in
// memory, the others take the slot number from it through LDS — so an id offered by many positions (a hub of a power-law graph)
// costs one same-address operation per workgroup, not per position. look_first: the leader looks at a slot before its
// compare-and-swap, or not (see au_direct_cas). Results are the same words as au_insert_kernel's (every step is a min or an
// insert-if-absent), so everything downstream is bit-identical.
constexpr int kAuLdsSlots = 2 * kBlock;
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void au_insert_merged_kernel(const KeyT* targets, int nt, const KeyT* neighbors, int nn,
                                                                  const int* nn_dev, KeyT* slots, uint32_t* min_pos,
                                                                  uint32_t* slot_of, uint32_t cap, const int* nt_dev,
                                                                  int look_first)
{
  __shared__ unsigned long long l_key[kAuLdsSlots];   // 32-bit ids: (id << 32 | smallest position); 64-bit ids: the id
  __shared__ uint32_t l_pos[kAuLdsSlots];             // 64-bit ids: smallest position
  __shared__ uint32_t l_slot[kAuLdsSlots];            // where the id landed in the table in memory
  __shared__ int l_any_repeat;                        // an id occurs twice among the workgroup's entries
  for (int k = threadIdx.x; k < kAuLdsSlots; k += kBlock) l_key[k] = ~0ull, l_pos[k] = ~0u;
  if (threadIdx.x == 0) l_any_repeat = 0;
  const int i          = blockIdx.x * blockDim.x + threadIdx.x;
  const int nt_use     = nt_dev != nullptr ? min(nt, *nt_dev) : nt;
  const int nn_use     = nn_dev != nullptr ? min(nn, *nn_dev) : nn;
  const bool is_target = i < nt;
  const int entry      = is_target ? i : i - nt;
  const bool active    = i < nt + nn && entry < (is_target ? nt_use : nn_use);
  const int pos        = is_target ? entry : nt_use + entry;
  constexpr KeyT kEmpty = ~static_cast<KeyT>(0);
  const KeyT key        = active ? (is_target ? targets[entry] : neighbors[entry]) : static_cast<KeyT>(0);
  // 64-bit ids: the id that looks like "empty" has a slot of its own in memory and does not go through LDS
  const bool odd_one = sizeof(KeyT) == 8 && key == kEmpty;
  const unsigned long long mine =
    sizeof(KeyT) == 4 ? (static_cast<unsigned long long>(key) << 32) | static_cast<uint32_t>(pos) : static_cast<unsigned long long>(key);
  __syncthreads();
  uint32_t ls = 0;
  if (active && !odd_one) {
    ls = (au_hash(static_cast<uint64_t>(key)) >> 11) & (kAuLdsSlots - 1);
    for (;;) {
      unsigned long long cur = l_key[ls];
      if (cur == ~0ull) {
        cur = atomicCAS(&l_key[ls], ~0ull, mine);
        if (cur == ~0ull) break;
      }
      if (sizeof(KeyT) == 4) {
        if (static_cast<uint32_t>(cur >> 32) == static_cast<uint32_t>(key)) {
          if (cur > mine) atomicMin(&l_key[ls], mine);
          l_any_repeat = 1;
          break;
        }
      } else if (cur == mine) {
        l_any_repeat = 1;
        break;
      }
      ls = (ls + 1) & (kAuLdsSlots - 1);
    }
    if (sizeof(KeyT) == 8) atomicMin(&l_pos[ls], static_cast<uint32_t>(pos));
  }
  __syncthreads();
  const bool leader = active && !odd_one && (sizeof(KeyT) == 4 ? l_key[ls] == mine : l_pos[ls] == static_cast<uint32_t>(pos));
  if (leader || (active && odd_one)) {
    uint32_t s = cap;
    if constexpr (sizeof(KeyT) == 4) {
      unsigned long long* table = reinterpret_cast<unsigned long long*>(slots);
      s = au_hash(static_cast<uint64_t>(key)) & (cap - 1);
      for (;;) {
        unsigned long long cur = look_first ? table[s] : ~0ull;
        if (cur == ~0ull) {
          cur = atomicCAS(&table[s], ~0ull, mine);
          if (cur == ~0ull) break;                     // the slot is mine
        }
        if (static_cast<uint32_t>(cur >> 32) == static_cast<uint32_t>(key)) {   // equal ids: the words compare by position
          if (cur > mine) atomicMin(&table[s], mine);
          break;
        }
        s = (s + 1) & (cap - 1);
      }
    } else {
      if (!odd_one) {
        s = au_hash(static_cast<uint64_t>(key)) & (cap - 1);
        for (;;) {
          KeyT cur = look_first ? slots[s] : kEmpty;
          if (cur == kEmpty) cur = au_cas(&slots[s], kEmpty, key);   // returns what was there: empty = the slot is mine now
          if (cur == kEmpty || cur == key) break;
          s = (s + 1) & (cap - 1);
        }
      }
      if (!look_first || min_pos[s] > static_cast<uint32_t>(pos)) atomicMin(&min_pos[s], static_cast<uint32_t>(pos));
    }
    slot_of[i] = s;
    if (!odd_one) l_slot[ls] = s;
  }
  // (the usual workgroup of a mini-batch hop holds no id twice: every entry led, nothing to hand over, no third barrier —
  // l_any_repeat was written before the second barrier and is read by everybody after it: the branch is uniform)
  if (!l_any_repeat) return;
  __syncthreads();
  if (active && !odd_one && !leader) slot_of[i] = l_slot[ls];
}

__global__ __launch_bounds__(kBlock) void au_flag_kernel(const uint32_t* min_pos, const uint32_t* slot_of, int nt, int nn,
                                                         const int* nn_dev, int* first_flag, int stride, const int* nt_dev)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > nn) return;
  const int used   = nn_dev != nullptr ? min(nn, *nn_dev) : nn;
  const int nt_use = nt_dev != nullptr ? min(nt, *nt_dev) : nt;
  // first occurrence of an id that no target holds (targets sit at positions < nt); the entries past the ones in use are
  // zero, so the exclusive scan's last entry (index nn) is the number of new ids whatever `used` is
  // (stride 2: 32-bit ids keep (id, position) words, min_pos then points at their low halves)
  first_flag[p] = p < used && min_pos[static_cast<size_t>(slot_of[nt + p]) * stride] == static_cast<uint32_t>(nt_use + p) ? 1 : 0;
}

// one new-count word written where the caller wants it (device and / or pinned host memory): replaces 4-byte copy commands
__global__ void au_publish_kernel(const int* new_rank_end, const int* nn_dev, int nn, int* new_count_dev, int* publish_host,
                                  const int* nt_dev, int nt, int* n_unique_dev)
{
  const int c = *new_rank_end;
  if (new_count_dev != nullptr) *new_count_dev = c;
  if (publish_host != nullptr) {
    publish_host[0] = nn_dev != nullptr ? min(nn, *nn_dev) : nn;
    publish_host[1] = c;
  }
  if (n_unique_dev != nullptr) *n_unique_dev = (nt_dev != nullptr ? min(nt, *nt_dev) : nt) + c;
}

// (the grid covers max(nt, nn): the same launch copies the targets to the head of the output and, for the fused hop, the
// centre local ids from their scratch to the exactly sized output — both were copy commands of their own)
struct au_flag_fn {   // au_flag_kernel as a function of the neighbour position: the input of the ranking scan (chain_scan_kernel)
  const uint32_t* min_pos;
  const uint32_t* slot_of;
  int nt, nn, stride;
  const int *nn_dev, *nt_dev;
  int used, nt_use;   // filled by prepare()
  __device__ bool prepare()
  {
    used   = nn_dev != nullptr ? min(nn, *nn_dev) : nn;
    nt_use = nt_dev != nullptr ? min(nt, *nt_dev) : nt;
    return used > 0;
  }
  // (clamp the position, mask the result: a load behind a per-lane `p < used` is a conditionally defined value, and the 16
  // values of a thread would be fetched one after the other — the same trap as in the row kernels)
  __device__ int operator()(int p) const
  {
    const int pc     = min(p, used - 1);
    const uint32_t m = min_pos[static_cast<size_t>(slot_of[nt + pc]) * stride];
    return static_cast<int>(p < used) & static_cast<int>(m == static_cast<uint32_t>(nt_use + p));   // (no short circuit: see chain_scan_kernel)
  }
};
struct au_publish_fn {   // au_publish_kernel as the scan's tail: `total` = the number of new unique neighbours
  const int *nn_dev, *nt_dev;
  int nn, nt;
  int *new_count_dev, *publish_host, *n_unique_dev;
  __device__ void operator()(int total) const
  {
    if (new_count_dev != nullptr) *new_count_dev = total;
    if (publish_host != nullptr) {
      publish_host[0] = nn_dev != nullptr ? min(nn, *nn_dev) : nn;
      publish_host[1] = total;
    }
    if (n_unique_dev != nullptr) *n_unique_dev = (nt_dev != nullptr ? min(nt, *nt_dev) : nt) + total;
  }
};

template <typename KeyT>
__global__ __launch_bounds__(kBlock) void au_emit_kernel(const KeyT* slots, const uint32_t* min_pos, const uint32_t* slot_of,
                                                         const int* new_rank, int nt, int nn, KeyT* out_unique, int* mapping,
                                                         const KeyT* targets, const int* copy_src, int* copy_dst,
                                                         const int* nt_dev, const int* nn_dev, int* publish_late, int* n_unique_late,
                                                         int nn_room, int pad_room)
{
  const int p      = blockIdx.x * blockDim.x + threadIdx.x;
  const int nt_use = nt_dev != nullptr ? min(nt, *nt_dev) : nt;
  const int nn_use = nn_dev != nullptr ? min(nn, *nn_dev) : nn;
  if (pad_room > 0) {   // wm_au_bounds::pad_unique_tail: everything behind the unique ids reads "skip me"
    const int stride = gridDim.x * blockDim.x;
    for (int q = nt_use + new_rank[nn_room] + p; q < pad_room; q += stride) out_unique[q] = ~static_cast<KeyT>(0);
  }
  if (p == 0 && publish_late != nullptr) {   // what au_publish_kernel writes, here (wm_au_bounds::publish_host_late)
    const int c     = new_rank[nn_room];     // the scan ran over the ROOM of the neighbour array: its last entry is the total
    publish_late[0] = nn_use;
    publish_late[1] = c;
    if (n_unique_late != nullptr) *n_unique_late = nt_use + c;
  }
  if (p < nt_use) out_unique[p] = targets[p];
  if (p >= nn_use) return;
  if (copy_dst != nullptr) copy_dst[p] = copy_src[p];
  const uint32_t s = slot_of[nt + p];
  int m;
  KeyT key;
  if constexpr (sizeof(KeyT) == 4) {
    const unsigned long long w = reinterpret_cast<const unsigned long long*>(slots)[s];
    m   = static_cast<int>(static_cast<uint32_t>(w));
    key = static_cast<KeyT>(w >> 32);
  } else {
    m   = static_cast<int>(min_pos[s]);
    key = slots[s];
  }
  const int uid = m < nt_use ? m : nt_use + new_rank[m - nt_use];
  if (m == nt_use + p) out_unique[uid] = key;
  if (mapping != nullptr) mapping[p] = uid;
}

// The table insert of a key: look at the slot first and compare-and-swap only when it reads empty (two dependent round trips for
// a new id, none of them an atomic for a repeated one), or compare-and-swap straight away (one round trip, an atomic per key).
// Either way the keys of a workgroup are merged in LDS first (au_insert_merged_kernel): an id offered by many positions then
// costs one operation on the table per WORKGROUP that holds it, not one per position — whole append_unique calls of 31.7 k +
// 952 k keys: uniform ids 132 -> 131 us, Zipf(1.05) ids (the hottest one x 50 k) 357 -> 122 us, Zipf(1.3) 1378 -> 99 us, one id
// 1.8 ms -> 58 us; 8 M keys: uniform 747 -> 753 us, Zipf(1.05) 740 -> 498 us, Zipf(1.3) 1.52 ms -> 0.22 ms
// (profiles/r05_au_insert_skew.txt; real graphs are skewed, bench.py's synthetic one is not).
// Without the look: the hops of the sampling chain (the table was emptied by the sampler on the side, long before) of up to 2 M
// keys — C5's step 0.243-0.245 -> 0.237-0.240 ms (0.230-0.235 without the merge, which a skewed hop would pay for dearly:
// Zipf(1.05) 357 -> 652 us). A call that has just filled its table finds the freshly written lines close by and is 10 us
// faster WITH the look (uniform 131 vs 141 us), and a 37 M-key hop is bound by the memory-side atomic units, most of its keys
// being repeats the look filters out (10.43 vs 10.95 ms per step at 65 536 seeds): those keep the look
// (profiles/r05_c5_direct_cas_ab.txt). WM_AU_DIRECT_CAS=0 / 1 forces; WM_AU_MERGE=0 (A/B only): rounds 3-4's kernel.
inline int au_direct_cas(int64_t keys, bool table_cleared_long_ago = true)
{
  const char* e = WM_KNOB("WM_AU_DIRECT_CAS");
  if (e != nullptr && e[0] == '0') return 0;
  if (e != nullptr && (e[0] == '1' || e[0] == '2')) return 1;
  return table_cleared_long_ago && keys <= (INT64_C(2) << 20);
}
inline bool au_merged()
{
  const char* e = WM_KNOB("WM_AU_MERGE");
  return !(e != nullptr && e[0] == '0');
}
inline int au_direct_cas_plain()
{
  const char* e = WM_KNOB("WM_AU_DIRECT_CAS");
  return e != nullptr && e[0] == '2';
}

template <typename KeyT>
int au_phase1(const void* targets, int nt, const void* neighbors, int nn, const int* nn_dev, void* ws, int* new_count_dev,
              int* publish_host, const wm_au_bounds* bounds, hipStream_t stream)
{
  const int* nt_dev = bounds != nullptr ? bounds->n_target_dev : nullptr;
  if (bounds != nullptr && bounds->n_neighbor_dev != nullptr) nn_dev = bounds->n_neighbor_dev;
  using UKey  = typename std::make_unsigned<KeyT>::type;
  auto l      = au_plan<UKey>(ws, nt, nn);
  const int n = nt + nn;
  if (!(bounds != nullptr && bounds->table_is_clear) && fill_ff(l.scan_state, l.table_bytes, stream) != 0) return -2;
  if (n > 0 && au_merged())
    hipLaunchKernelGGL((au_insert_merged_kernel<UKey>), dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                       static_cast<const UKey*>(targets), nt, static_cast<const UKey*>(neighbors), nn, nn_dev, l.slots,
                       l.min_pos, l.slot_of, l.cap, nt_dev, au_direct_cas(n, bounds != nullptr && bounds->table_is_clear) ? 0 : 1);
  else if (n > 0)
    hipLaunchKernelGGL((au_insert_kernel<UKey>), dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                       static_cast<const UKey*>(targets), nt, static_cast<const UKey*>(neighbors), nn, nn_dev, l.slots,
                       l.min_pos, l.slot_of, l.cap, nt_dev, au_direct_cas_plain());
  // 32-bit ids: the positions are the low halves of the (id, position) words that start where `slots` starts
  const uint32_t* positions = sizeof(UKey) == 4 ? reinterpret_cast<const uint32_t*>(l.slots) : l.min_pos;
  const bool late = bounds != nullptr && bounds->publish_host_late != nullptr && new_count_dev == nullptr && publish_host == nullptr && nt + nn > 0;
  // flags, ranking scan and the publishing of the count as ONE launch (round 4: chain_scan_kernel evaluates the flag — two
  // dependent random loads — for a thread's values up front; rocPRIM's look-back scan over the same functor took 19.9 us
  // against 6.8 + 5.9 us for flag kernel + plain scan, which is why round 3 kept four launches here). WM_AU_FUSED_SCAN=0: A/B.
  const char* fused_sw = WM_AB_KNOB("WM_AU_FUSED_SCAN");
  if (chain_scan_fits(static_cast<int64_t>(nn) + 1) && !(fused_sw != nullptr && fused_sw[0] == '0')) {
    au_flag_fn fn{positions, l.slot_of, nt, nn, sizeof(UKey) == 4 ? 2 : 1, nn_dev, nt_dev, 0, 0};
    if (late) return chain_scan(fn, nn + 1, l.new_rank, no_tail{}, l.scan_state, stream);   // phase 2's kernel publishes
    au_publish_fn pub{nn_dev, nt_dev, nn, nt, new_count_dev, publish_host, bounds != nullptr ? bounds->n_unique_dev : static_cast<int*>(nullptr)};
    return chain_scan(fn, nn + 1, l.new_rank, pub, l.scan_state, stream);
  }
  hipLaunchKernelGGL(au_flag_kernel, dim3((nn + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, positions, l.slot_of, nt, nn,
                     nn_dev, l.first_flag, sizeof(UKey) == 4 ? 2 : 1, nt_dev);
  size_t tb = l.temp_bytes;
  if (rocprim::exclusive_scan(l.temp, tb, l.first_flag, l.new_rank, 0, static_cast<size_t>(nn) + 1, rocprim::plus<int>(),
                              stream) != hipSuccess)
    return -2;
  // new_rank[nn] = number of new unique neighbours
  if (bounds != nullptr && bounds->publish_host_late != nullptr && new_count_dev == nullptr && publish_host == nullptr && nt + nn > 0)
    return hipGetLastError() == hipSuccess ? 0 : -2;   // phase 2's kernel publishes (its grid is never empty then)
  hipLaunchKernelGGL(au_publish_kernel, dim3(1), dim3(1), 0, stream, l.new_rank + nn, nn_dev, nn, new_count_dev, publish_host,
                     nt_dev, nt, bounds != nullptr ? bounds->n_unique_dev : static_cast<int*>(nullptr));
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <typename KeyT>
int au_phase2(const void* targets, int nt, int nn, int nn_used, void* ws, void* out_unique, int* mapping, const int* copy_src,
              int* copy_dst, const wm_au_bounds* bounds, hipStream_t stream)
{
  const int* nt_dev = bounds != nullptr ? bounds->n_target_dev : nullptr;
  const int* nn_dev = bounds != nullptr ? bounds->n_neighbor_dev : nullptr;
  using UKey  = typename std::make_unsigned<KeyT>::type;
  auto l      = au_plan<UKey>(ws, nt, nn);   // the layout phase 1 used
  const int g = std::max(nt, nn_used);
  int* late   = bounds != nullptr ? bounds->publish_host_late : nullptr;
  if (g > 0)
    hipLaunchKernelGGL((au_emit_kernel<UKey>), dim3((g + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, l.slots, l.min_pos,
                       l.slot_of, l.new_rank, nt, nn_used, static_cast<UKey*>(out_unique), mapping,
                       static_cast<const UKey*>(targets), copy_src, copy_dst, nt_dev, nn_dev, late,
                       late != nullptr ? bounds->n_unique_dev : static_cast<int*>(nullptr), nn,
                       bounds != nullptr && bounds->pad_unique_tail ? nt + nn : 0);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// -- the sort route (round 1), kept for BIG inputs: a table for tens of millions of ids no longer sits in the caches and its
// random atomics lose against the radix sort's streaming passes (1 M + 22.4 M ids: sort 2.44 ms, table 2.92 ms): stable
// radix sort of (id, position) over targets ++ neighbours, run heads by max-scan, "first occurrence is a neighbour" flags,
// exclusive scan -> unique index
template <typename KeyT>
__global__ void concat_keys_kernel(const KeyT* targets, int nt, const KeyT* neighbors, int nn, KeyT* keys, int* pos)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nt + nn) return;
  keys[i] = i < nt ? targets[i] : neighbors[i - nt];
  pos[i]  = i;
}

template <typename KeyT>
__global__ void run_heads_kernel(const KeyT* sorted, int n, int* head_index)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  head_index[i] = (i == 0 || sorted[i] != sorted[i - 1]) ? i : 0;  // max-scan turns this into "index of my run head"
}

// first_flag[p] = 1 for a neighbour position p (0-based in the neighbour array) that is the first occurrence of an id
// which is not a target
__global__ void mark_new_kernel(const int* head_of, const int* sorted_pos, int n, int nt, int* first_flag)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (head_of[i] == i) {  // run head: the smallest original position of this id (stable sort)
    const int p0 = sorted_pos[i];
    if (p0 >= nt) first_flag[p0 - nt] = 1;
  }
}

template <typename KeyT>
__global__ void emit_unique_kernel(const KeyT* sorted, const int* head_of, const int* sorted_pos, const int* new_rank,
                                   int n, int nt, KeyT* out_unique, int* mapping)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int h  = head_of[i];
  const int p0 = sorted_pos[h];
  const int uid = p0 < nt ? p0 : nt + new_rank[p0 - nt];
  if (h == i && p0 >= nt) out_unique[uid] = sorted[i];
  const int me = sorted_pos[i];
  if (mapping != nullptr && me >= nt) mapping[me - nt] = uid;
}

struct max_op {
  __host__ __device__ int operator()(int a, int b) const { return a > b ? a : b; }
};

template <typename KeyT>
struct aus_layout {
  KeyT *keys, *sorted;
  int *pos, *sorted_pos, *head, *first_flag, *new_rank;
  void* temp;
  size_t temp_bytes, total;
};

template <typename KeyT>
aus_layout<KeyT> aus_plan(void* ws, int nt, int nn)
{
  const size_t n = static_cast<size_t>(nt) + nn;
  auto al        = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  size_t sort_b = 0, scan_b = 0, scan2_b = 0;
  (void)rocprim::radix_sort_pairs(nullptr, sort_b, static_cast<const KeyT*>(nullptr), static_cast<KeyT*>(nullptr),
                                  static_cast<const int*>(nullptr), static_cast<int*>(nullptr), n, 0, 8 * sizeof(KeyT),
                                  nullptr);
  (void)rocprim::inclusive_scan(nullptr, scan_b, static_cast<const int*>(nullptr), static_cast<int*>(nullptr), n, max_op(),
                                nullptr);
  (void)rocprim::exclusive_scan(nullptr, scan2_b, static_cast<const int*>(nullptr), static_cast<int*>(nullptr), 0,
                                static_cast<size_t>(nn) + 1, rocprim::plus<int>(), nullptr);
  aus_layout<KeyT> l;
  char* p  = static_cast<char*>(ws);
  size_t o = 0;
  l.keys = reinterpret_cast<KeyT*>(p + o), o += al(sizeof(KeyT) * n);
  l.sorted = reinterpret_cast<KeyT*>(p + o), o += al(sizeof(KeyT) * n);
  l.pos = reinterpret_cast<int*>(p + o), o += al(4 * n);
  l.sorted_pos = reinterpret_cast<int*>(p + o), o += al(4 * n);
  l.head = reinterpret_cast<int*>(p + o), o += al(4 * n);
  l.first_flag = reinterpret_cast<int*>(p + o), o += al(4 * (static_cast<size_t>(nn) + 1));
  l.new_rank = reinterpret_cast<int*>(p + o), o += al(4 * (static_cast<size_t>(nn) + 1));
  l.temp       = p + o;
  l.temp_bytes = std::max(sort_b, std::max(scan_b, scan2_b));
  l.total      = o + al(l.temp_bytes) + 256;
  return l;
}

template <typename KeyT>
int aus_phase1(const void* targets, int nt, const void* neighbors, int nn, void* ws, int* new_count_dev, int* publish_host,
               hipStream_t stream)
{
  using UKey     = typename std::make_unsigned<KeyT>::type;
  auto l         = aus_plan<UKey>(ws, nt, nn);
  const int n    = nt + nn;
  const int blks = (n + kBlock - 1) / kBlock;
  hipLaunchKernelGGL((concat_keys_kernel<UKey>), dim3(blks), dim3(kBlock), 0, stream, static_cast<const UKey*>(targets), nt,
                     static_cast<const UKey*>(neighbors), nn, l.keys, l.pos);
  size_t tb = l.temp_bytes;
  if (rocprim::radix_sort_pairs(l.temp, tb, l.keys, l.sorted, l.pos, l.sorted_pos, static_cast<size_t>(n), 0,
                                8 * sizeof(KeyT), stream) != hipSuccess)
    return -2;
  hipLaunchKernelGGL((run_heads_kernel<UKey>), dim3(blks), dim3(kBlock), 0, stream, l.sorted, n, l.head);
  tb = l.temp_bytes;
  if (rocprim::inclusive_scan(l.temp, tb, l.head, l.head, static_cast<size_t>(n), max_op(), stream) != hipSuccess) return -2;
  if (hipMemsetAsync(l.first_flag, 0, sizeof(int) * (static_cast<size_t>(nn) + 1), stream) != hipSuccess) return -2;
  hipLaunchKernelGGL(mark_new_kernel, dim3(blks), dim3(kBlock), 0, stream, l.head, l.sorted_pos, n, nt, l.first_flag);
  tb = l.temp_bytes;
  if (rocprim::exclusive_scan(l.temp, tb, l.first_flag, l.new_rank, 0, static_cast<size_t>(nn) + 1, rocprim::plus<int>(),
                              stream) != hipSuccess)
    return -2;
  // new_rank[nn] = number of new unique neighbours
  hipLaunchKernelGGL(au_publish_kernel, dim3(1), dim3(1), 0, stream, l.new_rank + nn, static_cast<const int*>(nullptr), nn,
                     new_count_dev, publish_host, static_cast<const int*>(nullptr), nt, static_cast<int*>(nullptr));
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <typename KeyT>
int aus_phase2(const void* targets, int nt, int nn, void* ws, void* out_unique, int* mapping, const int* copy_src, int* copy_dst,
               hipStream_t stream)
{
  if (copy_dst != nullptr && nn > 0 &&
      hipMemcpyAsync(copy_dst, copy_src, sizeof(int) * static_cast<size_t>(nn), hipMemcpyDeviceToDevice, stream) != hipSuccess)
    return -2;
  using UKey     = typename std::make_unsigned<KeyT>::type;
  auto l         = aus_plan<UKey>(ws, nt, nn);
  const int n    = nt + nn;
  const int blks = (n + kBlock - 1) / kBlock;
  if (nt > 0 && hipMemcpyAsync(out_unique, targets, sizeof(KeyT) * nt, hipMemcpyDeviceToDevice, stream) != hipSuccess) return -2;
  hipLaunchKernelGGL((emit_unique_kernel<UKey>), dim3(blks), dim3(kBlock), 0, stream, l.sorted, l.head, l.sorted_pos,
                     l.new_rank, n, nt, static_cast<UKey*>(out_unique), mapping);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}


// which route (experiments/au_crossover.py, whole calls). 64-bit ids: the sort moves twice the key bytes through twice the
// passes and the table wins or ties up to the largest size measured (23 M: 3.02 vs 3.12 ms): the table serves up to 24 M keys.
// 32-bit ids, round 2 (id and smallest position in two arrays, two atomics per new id): the table won up to ~2 M keys only.
// Round 3 (one 64-bit word per slot, one atomic per new id): it wins at every size measured — 950 k keys 0.139 vs 0.260 ms,
// 4.2 M 0.42 vs 0.54, 16.8 M 1.52 vs 2.05, 23.4 M 2.01 vs 2.46 ms, and the 65 536-seed C5 step (a 37 M-key second hop) 10.15
// vs 11.0 ms — so it serves up to 128 M keys (a 2 GiB table). WM_AU_TABLE_MAX overrides the limit for both widths.
inline bool au_use_table(int nt, int nn, wholememory_dtype_t dt)
{
  const int64_t forced = [] {
    const char* e = WM_AB_KNOB("WM_AU_TABLE_MAX");
    return e != nullptr ? static_cast<int64_t>(atoll(e)) : INT64_C(-1);
  }();
  const int64_t limit = forced >= 0 ? forced : (dt == WHOLEMEMORY_DT_INT ? INT64_C(128) << 20 : INT64_C(24) << 20);
  return static_cast<int64_t>(nt) + nn <= limit;
}

__global__ void add_self_loop_kernel(const int* row_ptr, const int* col, int* out_row, int* out_col, int n_rows)
{
  // reference csr_add_self_loop_func.cuh:24-44: block per row, the node itself first, then its neighbours
  const int row = blockIdx.x;
  const int s = row_ptr[row], e = row_ptr[row + 1];
  if (threadIdx.x == 0) {
    out_row[row] = s + row;
    if (row == n_rows - 1) out_row[row + 1] = e + row + 1;
  }
  for (int k = threadIdx.x; k <= e - s; k += blockDim.x) out_col[s + row + k] = k == 0 ? row : col[s + k - 1];
}

// out[i, c] = T(float(i)) + in[c]: the arithmetic of the env-function self-test op (wholememory_test_op.cu:25-37)
template <typename T>
__global__ void env_test_kernel(const T* in, T* out, int64_t dim, int64_t stride)
{
  const int id = blockIdx.x;
  const float f_id = static_cast<float>(id);
  for (int c = threadIdx.x; c < dim; c += blockDim.x) out[stride * id + c] = static_cast<T>(static_cast<T>(f_id) + in[c]);
}

}  // namespace

int hip_env_test_fill(const void* in, void* out, wholememory_dtype_t dt, int64_t dim, int64_t entries, int64_t stride, void* stream_v)
{
  if (entries == 0 || dim == 0) return 0;
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  const int threads  = static_cast<int>(std::min<int64_t>(dim, 512));
#define WM_ENV_TEST(T) \
  hipLaunchKernelGGL((env_test_kernel<T>), dim3(entries), dim3(threads), 0, stream, static_cast<const T*>(in), static_cast<T*>(out), dim, stride)
  switch (dt) {
    case WHOLEMEMORY_DT_FLOAT: WM_ENV_TEST(float); break;
    case WHOLEMEMORY_DT_DOUBLE: WM_ENV_TEST(double); break;
    case WHOLEMEMORY_DT_HALF: WM_ENV_TEST(_Float16); break;
    case WHOLEMEMORY_DT_INT8: WM_ENV_TEST(int8_t); break;
    case WHOLEMEMORY_DT_INT16: WM_ENV_TEST(int16_t); break;
    case WHOLEMEMORY_DT_INT: WM_ENV_TEST(int32_t); break;
    case WHOLEMEMORY_DT_INT64: WM_ENV_TEST(int64_t); break;
    default: return -1;
  }
#undef WM_ENV_TEST
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- launchers exported to backend_hip ----
int hip_sample_counts(const wholememory_gref_t* row_gref, int64_t row_off, const int64_t* row_pairs, const void* centers,
                      wholememory_dtype_t id_dtype, int n, const int* n_dev, int max_sample, int* counts, void* stream_v)
{
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  gref_view rv{};
  if (row_pairs == nullptr) rv = make_view(*row_gref);
  const int blocks = (n + 1 + 127) / 128;
  if (id_dtype == WHOLEMEMORY_DT_INT)
    hipLaunchKernelGGL((sample_count_kernel<int32_t>), dim3(blocks), dim3(128), 0, stream, rv, row_off, row_pairs,
                       static_cast<const int32_t*>(centers), n, n_dev, max_sample, counts);
  else if (id_dtype == WHOLEMEMORY_DT_INT64)
    hipLaunchKernelGGL((sample_count_kernel<int64_t>), dim3(blocks), dim3(128), 0, stream, rv, row_off, row_pairs,
                       static_cast<const int64_t*>(centers), n, n_dev, max_sample, counts);
  else
    return -1;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int hip_sample_pair_ids(const void* centers, wholememory_dtype_t id_dtype, int n, int64_t* ids, void* stream_v)
{
  if (n <= 0) return 0;
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  const int blocks   = (n + 255) / 256;
  if (id_dtype == WHOLEMEMORY_DT_INT)
    hipLaunchKernelGGL((pair_ids_kernel<int32_t>), dim3(blocks), dim3(256), 0, stream, static_cast<const int32_t*>(centers), n, ids);
  else if (id_dtype == WHOLEMEMORY_DT_INT64)
    hipLaunchKernelGGL((pair_ids_kernel<int64_t>), dim3(blocks), dim3(256), 0, stream, static_cast<const int64_t*>(centers), n, ids);
  else
    return -1;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// offsets[i] = sum of min(degree(center j), max_sample) over j < i, i = 0 .. n: the per-centre counts are computed inside the
// scan's input iterator (no count kernel, no count array)
template <typename IdT>
struct degree_fn {
  gref_view row_ptr;
  int64_t row_off;
  const IdT* centers;
  int n, max_sample;
  const int* n_dev;   // optional: centres in use of the n the arrays are sized for
  int used;           // filled by prepare()
  __device__ bool prepare()
  {
    used = n_dev != nullptr ? min(n, *n_dev) : n;
    return used > 0;
  }
  __device__ int operator()(int i) const
  {
    // clamped index, masked result (see au_flag_fn): the scan runs over n + 1 entries, the ones past `used` count 0 (reference :334-338)
    const int64_t nid = static_cast<int64_t>(centers[min(i, used - 1)]);
    const int64_t s   = gref_load<int64_t>(row_ptr, row_off + nid);
    const int64_t e   = gref_load<int64_t>(row_ptr, row_off + nid + 1);
    int deg           = static_cast<int>(e - s);
    if (max_sample > 0) deg = min(deg, max_sample);
    return deg & -static_cast<int>(i < used);
  }
};
int hip_sample_offsets(const wholememory_gref_t* row_gref, int64_t row_off, const void* centers, wholememory_dtype_t id_dtype,
                       int n, const int* n_dev, int max_sample, int* offsets, void* ws, size_t ws_bytes, int ws_is_ones,
                       void* stream_v)
{
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  const gref_view rv = make_view(*row_gref);
  const size_t need  = chain_state_bytes(static_cast<int64_t>(n) + 1);
  if (!chain_scan_fits(static_cast<int64_t>(n) + 1) || ws == nullptr || ws_bytes < need) return -3;   // nothing queued: count kernel + scan
  if (id_dtype != WHOLEMEMORY_DT_INT && id_dtype != WHOLEMEMORY_DT_INT64) return -1;
  if (!ws_is_ones && fill_ff(ws, (need + 15) & ~size_t(15), stream) != 0) return -2;
  if (id_dtype == WHOLEMEMORY_DT_INT) {
    degree_fn<int32_t> fn{rv, row_off, static_cast<const int32_t*>(centers), n, max_sample, n_dev, 0};
    return chain_scan(fn, n + 1, offsets, no_tail{}, ws, stream);
  }
  degree_fn<int64_t> fn{rv, row_off, static_cast<const int64_t*>(centers), n, max_sample, n_dev, 0};
  return chain_scan(fn, n + 1, offsets, no_tail{}, ws, stream);
}

size_t hip_scan_i32_ws_bytes(int64_t n)
{
  size_t b = 0;
  (void)rocprim::exclusive_scan(nullptr, b, static_cast<const int*>(nullptr), static_cast<int*>(nullptr), 0,
                                static_cast<size_t>(n), rocprim::plus<int>(), nullptr);
  return ((std::max(b, chain_state_bytes(n)) + 255) & ~static_cast<size_t>(255)) + 256;   // (either route's scratch; a multiple of 256)
}
int hip_fill_ff(void* ptr, size_t bytes, void* stream) { return fill_ff(ptr, bytes, static_cast<hipStream_t>(stream)); }
int hip_exclusive_scan_i32(const int* in, int* out, int64_t n, void* ws, size_t ws_bytes, void* stream)
{
  size_t b = ws_bytes;
  return rocprim::exclusive_scan(ws, b, in, out, 0, static_cast<size_t>(n), rocprim::plus<int>(),
                                 static_cast<hipStream_t>(stream)) == hipSuccess ? 0 : -2;
}

int hip_sample_unweighted(const wm_sample_args* a, void* stream_v)
{
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  sample_params p{};
  p.centers = a->centers, p.n_center = a->n_center, p.max_sample = a->max_sample_count;
  p.seed = a->random_seed, p.offsets = a->sample_offsets;
  p.out_ids = a->out_ids, p.out_lid = a->out_center_lid, p.out_egid = a->out_edge_gid;
  p.n_center_dev = a->n_center_dev;
  if (a->fill_ff_ptr != nullptr && a->fill_ff_bytes > 0) {
    if ((reinterpret_cast<uintptr_t>(a->fill_ff_ptr) | a->fill_ff_bytes) & 15) return -1;
    p.fill_ptr = a->fill_ff_ptr, p.fill_vecs = a->fill_ff_bytes / 16;
  }
  if (a->row_pairs != nullptr) {
    // positions only: row bounds come from the fetched pairs, the caller gathers the columns by edge id afterwards
    if (a->out_ids != nullptr || a->out_edge_gid == nullptr) return -1;
    p.pairs = a->row_pairs;
    return launch_sample<int64_t, int64_t>(p, stream);
  }
  p.row_ptr = make_view(a->row_gref), p.col_ptr = make_view(a->col_gref);
  p.row_off = a->row_storage_offset, p.col_off = a->col_storage_offset;
  const bool id32 = a->center_dtype == WHOLEMEMORY_DT_INT, col32 = a->col_dtype == WHOLEMEMORY_DT_INT;
  if ((!id32 && a->center_dtype != WHOLEMEMORY_DT_INT64) || (!col32 && a->col_dtype != WHOLEMEMORY_DT_INT64)) return -1;
  if (id32 && col32) return launch_sample<int32_t, int32_t>(p, stream);
  if (id32) return launch_sample<int32_t, int64_t>(p, stream);
  if (col32) return launch_sample<int64_t, int32_t>(p, stream);
  return launch_sample<int64_t, int64_t>(p, stream);
}

int hip_sample_weighted(const wm_sample_args* a, void* stream_v)
{
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (a->max_sample_count > kMaxWeighted) return -3;  // host side reports NOT_IMPLEMENTED before getting here
  weighted_params w{};
  sample_params& p = w.sp;
  p.row_ptr = make_view(a->row_gref), p.col_ptr = make_view(a->col_gref);
  p.row_off = a->row_storage_offset, p.col_off = a->col_storage_offset;
  p.centers = a->centers, p.n_center = a->n_center, p.max_sample = a->max_sample_count;
  p.seed = a->random_seed, p.offsets = a->sample_offsets;
  p.out_ids = a->out_ids, p.out_lid = a->out_center_lid, p.out_egid = a->out_edge_gid;
  w.weight_ptr = make_view(a->weight_gref);
  w.weight_off = a->weight_storage_offset;
  w.vthreads   = a->max_sample_count > 256 ? 256 : 128;  // reference block sizes (func.cuh:540-560, test utils :597-598)
  int P        = 128;
  while (P < 2 * a->max_sample_count) P <<= 1;
  w.capacity      = P;
  const bool id32 = a->center_dtype == WHOLEMEMORY_DT_INT, col32 = a->col_dtype == WHOLEMEMORY_DT_INT;
  if ((!id32 && a->center_dtype != WHOLEMEMORY_DT_INT64) || (!col32 && a->col_dtype != WHOLEMEMORY_DT_INT64)) return -1;
  if (a->weight_dtype == WHOLEMEMORY_DT_FLOAT) return dispatch_weighted<float>(w, id32, col32, stream);
  if (a->weight_dtype == WHOLEMEMORY_DT_DOUBLE) return dispatch_weighted<double>(w, id32, col32, stream);
  return -1;
}

// the part of an append_unique workspace that has to be all-ones before phase 1 (the empty hash table); -3: the sort route
int hip_append_unique_table_region(int nt, int nn, wholememory_dtype_t dt, void* ws, void** ptr, size_t* bytes)
{
  if (!au_use_table(nt, nn, dt)) return -3;
  if (dt == WHOLEMEMORY_DT_INT) {
    const auto l = au_plan<uint32_t>(ws, nt, nn);
    *ptr = l.scan_state, *bytes = l.table_bytes;
  } else {
    const auto l = au_plan<uint64_t>(ws, nt, nn);
    *ptr = l.scan_state, *bytes = l.table_bytes;
  }
  return 0;
}

size_t hip_append_unique_ws_bytes(int nt, int nn, wholememory_dtype_t dt)
{
  if (au_use_table(nt, nn, dt)) return dt == WHOLEMEMORY_DT_INT ? au_plan<uint32_t>(nullptr, nt, nn).total : au_plan<uint64_t>(nullptr, nt, nn).total;
  return dt == WHOLEMEMORY_DT_INT ? aus_plan<uint32_t>(nullptr, nt, nn).total : aus_plan<uint64_t>(nullptr, nt, nn).total;
}
int hip_append_unique_phase1(const void* targets, int nt, const void* neighbors, int nn, const int* nn_dev,
                             wholememory_dtype_t dt, void* ws, int* new_count_dev, int* publish_host, const wm_au_bounds* bounds,
                             void* stream_v)
{
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  const bool table   = au_use_table(nt, nn, dt);
  if (!table && (nn_dev != nullptr || bounds != nullptr)) return -3;   // the sort route needs the exact counts on the host
  if (dt == WHOLEMEMORY_DT_INT)
    return table ? au_phase1<int32_t>(targets, nt, neighbors, nn, nn_dev, ws, new_count_dev, publish_host, bounds, stream)
                 : aus_phase1<int32_t>(targets, nt, neighbors, nn, ws, new_count_dev, publish_host, stream);
  if (dt == WHOLEMEMORY_DT_INT64)
    return table ? au_phase1<int64_t>(targets, nt, neighbors, nn, nn_dev, ws, new_count_dev, publish_host, bounds, stream)
                 : aus_phase1<int64_t>(targets, nt, neighbors, nn, ws, new_count_dev, publish_host, stream);
  return -1;
}
int hip_append_unique_phase2(const void* targets, int nt, int nn, int nn_used, wholememory_dtype_t dt, void* ws,
                             void* out_unique, int* mapping, const int* copy_src, int* copy_dst, const wm_au_bounds* bounds,
                             void* stream_v)
{
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  const bool table   = au_use_table(nt, nn, dt);
  if (!table && (nn_used != nn || bounds != nullptr)) return -3;
  if (dt == WHOLEMEMORY_DT_INT)
    return table ? au_phase2<int32_t>(targets, nt, nn, nn_used, ws, out_unique, mapping, copy_src, copy_dst, bounds, stream)
                 : aus_phase2<int32_t>(targets, nt, nn, ws, out_unique, mapping, copy_src, copy_dst, stream);
  if (dt == WHOLEMEMORY_DT_INT64)
    return table ? au_phase2<int64_t>(targets, nt, nn, nn_used, ws, out_unique, mapping, copy_src, copy_dst, bounds, stream)
                 : aus_phase2<int64_t>(targets, nt, nn, ws, out_unique, mapping, copy_src, copy_dst, stream);
  return -1;
}
bool hip_append_unique_takes_bounds(int nt, int nn, wholememory_dtype_t dt)
{
  return (dt == WHOLEMEMORY_DT_INT || dt == WHOLEMEMORY_DT_INT64) && au_use_table(nt, nn, dt);
}
int hip_csr_add_self_loop(const int* row_ptr, const int* col, int* out_row, int* out_col, int n_rows, void* stream)
{
  if (n_rows <= 0) return 0;
  hipLaunchKernelGGL(add_self_loop_kernel, dim3(n_rows), dim3(64), 0, static_cast<hipStream_t>(stream), row_ptr, col, out_row,
                     out_col, n_rows);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace wm
