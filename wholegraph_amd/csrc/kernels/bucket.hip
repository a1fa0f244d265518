// wholegraph_amd — per-owner bucketing of lookup ids for the DISTRIBUTED exchange (gfx950 HIP).
//
// Reference behaviour being replaced (cpp/src/wholememory_ops/functions/):
//   bucket_ids_func.cu:51-87            counts[r] = #{ids owned by rank r}, negatives not counted
//   exchange_ids_nccl_func.cu:42-92     full-width stable radix sort of (unsigned id, position) so
//                                       that ids come out grouped by owner (owners hold contiguous
//                                       id ranges) with `raw_indices` = original positions
// MI355X design: grouping by owner does not need a sort. A stable W+1-way multisplit (W owners +
// one trailing bucket for negative ids, which the reference's unsigned sort also parks last and
// never sends) produces, in two streaming passes over the ids,
//   bucketed_ids : ids grouped by owner, ORIGINAL ORDER preserved inside each owner's segment
//   raw_indices  : original position of each bucketed id (int64, as gather_op_impl_nccl.cu:69-70)
//   counts       : bit-identical to the reference histogram
// Within an owner segment the reference has ids ascending (ties by position) where this has plain
// position order; both are stable w.r.t. equal ids, so every user-visible result (gathered rows,
// and the fp32 summation order of duplicate gradient rows at the owner) is identical.
// tests/ pin that: sorting each segment of (bucketed_ids, raw_indices) by id, stably, reproduces
// the oracle's sort_ids() output bit for bit.
//
// Kernels: (1) per-block histogram over a contiguous chunk, wave-ballot "peel" loop (one ballot
// per DISTINCT owner present in the wave, __popcll for the count); (2) one-block exclusive scan of
// the bucket-major [bucket][block] count matrix; (3) scatter: the same peel loop gives each lane
// its rank among equal-owner lanes via __popcll(mask & lanemask_lt); wave bases come from a tiny
// LDS prefix over the 4 waves of the block. Index traffic only: HBM-bound, no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "../backend.hpp"
#include "../knobs.hpp"

namespace wm {
namespace {

constexpr int kBlock      = 256;
constexpr int kWaves      = kBlock / 64;
constexpr int kItems      = 4;                       // 64-id groups per wave per block iteration
constexpr int kIterItems  = kBlock * kItems;         // 1024 ids per block iteration
constexpr int kMaxBlocks  = 2048;                    // (1024: the scan is quicker, the ballot-bound kernels lose more: 46 vs 33 us)
constexpr int kMaxBuckets = 257;                     // world_size <= 256
constexpr int kMaxOwners  = 1024;                    // ranges searched per id (== buckets unless owner_count is set)

struct bucket_geom {
  int64_t chunk;  // ids per block (multiple of kIterItems)
  int blocks;
};

inline bucket_geom geometry(int64_t n)
{
  bucket_geom g;
  int64_t want = std::max<int64_t>(1, std::min<int64_t>(kMaxBlocks, (n + 4 * kIterItems - 1) / (4 * kIterItems)));
  g.chunk      = ((n + want - 1) / want + kIterItems - 1) / kIterItems * kIterItems;
  if (g.chunk < kIterItems) g.chunk = kIterItems;
  g.blocks = static_cast<int>(std::max<int64_t>(1, (n + g.chunk - 1) / g.chunk));
  return g;
}

// owner of a non-negative id: the r with off[r] <= id < off[r+1]; branch-free count of passed
// boundaries (empty ranks share a boundary and are skipped naturally)
__device__ __forceinline__ int owner_of(uint64_t id, const uint64_t* s_off, int owners)
{
  if (owners <= 16) {
    int r = 0;
    for (int k = 1; k < owners; k++) r += (id >= s_off[k]) ? 1 : 0;
    return r;
  }
  int lo = 0, hi = owners;  // largest lo with s_off[lo] <= id: the same r as the count above (offsets are monotone)
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (id >= s_off[mid]) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ int bucket_of_owner(int o, int world, int owners) { return owners == world ? o : o % world; }
// a value every lane holds alike, moved to scalar registers
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v)
{
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
  return (static_cast<uint64_t>(hi) << 32) | lo;
}

// bucket of an id: its owner, or — when there are more owners than buckets (wm_bucket_args::owner_count) — owner % world
template <typename IdxT>
__device__ __forceinline__ int bucket_of(const IdxT* ids, int64_t i, int64_t n, const uint64_t* s_off, int world,
                                         int owners, IdxT& id_out)
{
  if (i >= n) return -1;  // padding lane: belongs to no bucket
  IdxT id = ids[i];
  id_out  = id;
  if (id < 0) return world;  // trailing "never sent" bucket
  int o = owner_of(static_cast<uint64_t>(id), s_off, owners);
  return owners == world ? o : o % world;
}

// DENSE (world + 1 <= kDenseBuckets, i.e. up to 16 ranks — one node): no peel loop. One ballot per BUCKET and 64-id group,
// the count of bucket b accumulates in a register of lane b; LDS is touched once per wave, at the end. The peel loop pays
// an LDS round trip (and in the scatter two) per distinct bucket and group, back to back: 40 / 98 us per 10 M ids over 8
// owners against 16 / 52 with this (profiles/r04_bucket_dense.txt).
constexpr int kDenseBuckets = 17;
template <typename IdxT, bool DENSE>
__global__ __launch_bounds__(kBlock) void bucket_hist_kernel(const IdxT* ids, int64_t n, const uint64_t* entry_offsets,
                                                             int world, int owners, int64_t chunk,
                                                             int64_t* block_counts)
{
  __shared__ uint64_t s_off[kMaxOwners + 1];
  __shared__ int s_cnt[kMaxBuckets];
  const int nb = world + 1;
  for (int i = threadIdx.x; i <= owners; i += kBlock) s_off[i] = entry_offsets[i];
  for (int i = threadIdx.x; i < nb; i += kBlock) s_cnt[i] = 0;
  __syncthreads();
  const int lane      = threadIdx.x & 63;
  const int64_t begin = static_cast<int64_t>(blockIdx.x) * chunk;
  const int64_t end   = min(begin + chunk, n);
  if constexpr (DENSE) {
    constexpr int kU = 4;   // ids per thread in flight, loaded unconditionally (clamped into the chunk; 8: 32 us against 28)
    int mine = 0;           // lane b: ids of bucket b this wave has seen
    if (owners == world) {
      // one bucket per owner: no owner lookup at all. The owners' first rows sit in SGPRs, "ids at or past boundary k" is one
      // 64-bit compare and a ballot per boundary, counted in scalar registers; bucket r = [boundary r, boundary r + 1)
      uint64_t bnd[kDenseBuckets - 1];
#pragma unroll
      for (int k = 0; k < kDenseBuckets - 1; k++) bnd[k] = uniform_u64(k >= 1 && k < owners ? s_off[k] : ~0ull);
      int at_or_past[kDenseBuckets - 1];   // [0]: every non-negative id
#pragma unroll
      for (int k = 0; k < kDenseBuckets - 1; k++) at_or_past[k] = 0;
      int negative = 0;
      for (int64_t base = begin; base < end; base += kBlock * kU) {
        IdxT id[kU];
#pragma unroll
        for (int u = 0; u < kU; u++) id[u] = ids[min(base + u * kBlock + threadIdx.x, end - 1)];
#pragma unroll
        for (int u = 0; u < kU; u++) {
          const bool in      = base + u * kBlock + threadIdx.x < end;
          const bool valid   = in && id[u] >= 0;
          const uint64_t uid = static_cast<uint64_t>(static_cast<int64_t>(id[u]));
          negative += __popcll(__ballot(in && id[u] < 0));
          at_or_past[0] += __popcll(__ballot(valid));
#pragma unroll
          for (int k = 1; k < kDenseBuckets - 1; k++)
            if (k < owners) at_or_past[k] += __popcll(__ballot(valid && uid >= bnd[k]));   // (wave-uniform guard)
        }
      }
#pragma unroll
      for (int r = 0; r < kDenseBuckets - 1; r++) {
        const int next = r + 1 < kDenseBuckets - 1 ? at_or_past[r + 1] : 0;   // (0 for boundaries that do not exist)
        if (lane == r) mine = at_or_past[r] - (r + 1 < owners ? next : 0);
      }
      if (lane == world) mine = negative;
    } else
    for (int64_t base = begin; base < end; base += kBlock * kU) {
      IdxT id[kU];
#pragma unroll
      for (int u = 0; u < kU; u++) id[u] = ids[min(base + u * kBlock + threadIdx.x, end - 1)];
#pragma unroll
      for (int u = 0; u < kU; u++) {
        const bool in = base + u * kBlock + threadIdx.x < end;
        const int b   = !in ? -1 : id[u] < 0 ? world : bucket_of_owner(owner_of(static_cast<uint64_t>(id[u]), s_off, owners), world, owners);
        for (int k = 0; k < nb; k++) {   // (wave-uniform trip count)
          const int c = __popcll(__ballot(b == k));
          mine += lane == k ? c : 0;
        }
      }
    }
    if (lane < nb && mine != 0) atomicAdd(&s_cnt[lane], mine);
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += kBlock)
      block_counts[static_cast<int64_t>(i) * gridDim.x + blockIdx.x] = s_cnt[i];
    return;
  }
  for (int64_t base = begin; base < end; base += kBlock) {
    IdxT id;
    int b            = bucket_of(ids, base + threadIdx.x, end, s_off, world, owners, id);
    uint64_t pending = __ballot(b >= 0);
    while (pending) {  // one trip per distinct bucket present in this wave
      int leader    = __ffsll(static_cast<long long>(pending)) - 1;
      int lb        = __shfl(b, leader, 64);
      uint64_t mask = __ballot(b == lb);
      if (lane == leader) atomicAdd(&s_cnt[lb], __popcll(mask));
      pending &= ~mask;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += kBlock)
    block_counts[static_cast<int64_t>(i) * gridDim.x + blockIdx.x] = s_cnt[i];
}

// exclusive scan of block_counts (bucket-major) in place; totals of the first `world` buckets to counts[]
// (round 4: a thread's span is read 8 values at a time, unconditionally — the one-load-per-trip loops of the first version
// paid a memory latency per value, twice: 31 us for the 18 k counts of 8 owners x 2048 blocks — and the 1024 partial sums meet
// through wave shuffles and two barriers instead of twenty)
__global__ __launch_bounds__(1024) void bucket_scan_kernel(int64_t* block_counts, int blocks, int world, int64_t* counts)
{
  __shared__ int64_t s_wave[16];
  constexpr int kV    = 24;   // 8 owners x 2048 blocks: 18 counts per thread, all in registers
  const int64_t total = static_cast<int64_t>(world + 1) * blocks;
  const int64_t per   = (total + 1023) / 1024;
  const int64_t b0    = min(static_cast<int64_t>(threadIdx.x) * per, total);
  const int64_t b1    = min(b0 + per, total);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int64_t sum         = 0;
  int64_t first[kV];   // the first kV values of the span stay in registers for the second pass (the whole span when per <= kV)
#pragma unroll
  for (int u = 0; u < kV; u++) first[u] = block_counts[min(b0 + u, total - 1)];
#pragma unroll
  for (int u = 0; u < kV; u++) sum += b0 + u < b1 ? first[u] : 0;
  for (int64_t i = b0 + kV; i < b1; i += kV) {
    int64_t v[kV];
#pragma unroll
    for (int u = 0; u < kV; u++) v[u] = block_counts[min(i + u, total - 1)];
#pragma unroll
    for (int u = 0; u < kV; u++) sum += i + u < b1 ? v[u] : 0;
  }
  int64_t incl = sum;   // inclusive scan over the wave, then over the 16 wave totals
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int64_t o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  if (lane == 63) s_wave[wv] = incl;
  __syncthreads();
  int64_t before = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) before += w < wv ? s_wave[w] : 0;
  int64_t run = before + incl - sum;  // exclusive prefix of this thread's span
#pragma unroll
  for (int u = 0; u < kV; u++) {
    if (b0 + u < b1) block_counts[b0 + u] = run;
    run += b0 + u < b1 ? first[u] : 0;
  }
  for (int64_t i = b0 + kV; i < b1; i += kV) {
    int64_t v[kV];
#pragma unroll
    for (int u = 0; u < kV; u++) v[u] = block_counts[min(i + u, total - 1)];
#pragma unroll
    for (int u = 0; u < kV; u++) {
      if (i + u < b1) block_counts[i + u] = run;
      run += i + u < b1 ? v[u] : 0;
    }
  }
  __syncthreads();
  // bucket totals: offset of next bucket's first block minus own first block
  for (int r = threadIdx.x; r < world; r += 1024) {
    int64_t start = block_counts[static_cast<int64_t>(r) * blocks];
    int64_t next  = block_counts[static_cast<int64_t>(r + 1) * blocks];
    counts[r]     = next - start;
  }
}

template <typename IdxT, bool DENSE>
__global__ __launch_bounds__(kBlock) void bucket_scatter_kernel(const IdxT* ids, int64_t n,
                                                                const uint64_t* entry_offsets, int world,
                                                                int owners, int64_t chunk,
                                                                const int64_t* block_offsets, IdxT* bucketed_ids,
                                                                int64_t* raw_indices)
{
  __shared__ uint64_t s_off[kMaxOwners + 1];
  __shared__ int64_t s_run[kMaxBuckets];          // next free slot per bucket for this block
  __shared__ volatile int s_wcnt[kWaves][kMaxBuckets];  // per-wave counts of the current iteration
  __shared__ int64_t s_wbase[kWaves][kMaxBuckets];
  const int nb   = world + 1;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i <= owners; i += kBlock) s_off[i] = entry_offsets[i];
  for (int i = threadIdx.x; i < nb; i += kBlock)
    s_run[i] = block_offsets[static_cast<int64_t>(i) * gridDim.x + blockIdx.x];
  __syncthreads();
  const int64_t begin    = static_cast<int64_t>(blockIdx.x) * chunk;
  const int64_t end      = min(begin + chunk, n);
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

  for (int64_t it = begin; it < end; it += kIterItems) {
    if (!DENSE) {
      for (int i = threadIdx.x; i < kWaves * nb; i += kBlock) s_wcnt[i / nb][i % nb] = 0;
    }
    __syncthreads();
    // wave `wave` owns ids [it + wave*256, it + wave*256 + 256) as kItems consecutive 64-groups
    IdxT id[kItems];
    int bkt[kItems];
    int rank[kItems];
    if constexpr (DENSE) {
      // ids loaded back to back (clamped into the chunk), then per group one ballot per bucket: lane b carries the wave's
      // running count of bucket b (what s_wcnt[wave][b] is in the peel loop), read with v_readlane — no LDS inside the loop
#pragma unroll
      for (int j = 0; j < kItems; j++) id[j] = ids[min(it + static_cast<int64_t>(wave) * (64 * kItems) + j * 64 + lane, end - 1)];
      int run = 0;
#pragma unroll
      for (int j = 0; j < kItems; j++) {
        const int64_t pos = it + static_cast<int64_t>(wave) * (64 * kItems) + j * 64 + lane;
        bkt[j]  = pos >= end ? -1 : id[j] < 0 ? world : bucket_of_owner(owner_of(static_cast<uint64_t>(id[j]), s_off, owners), world, owners);
        rank[j] = 0;
        for (int k = 0; k < nb; k++) {   // (wave-uniform trip count)
          const uint64_t mask = __ballot(bkt[j] == k);
          const int prior     = __builtin_amdgcn_readlane(run, k);
          if (bkt[j] == k) rank[j] = prior + __popcll(mask & lt_mask);
          run += lane == k ? __popcll(mask) : 0;
        }
      }
      if (lane < nb) s_wcnt[wave][lane] = run;
    }
#pragma unroll
    for (int j = 0; j < (DENSE ? 0 : kItems); j++) {
      const int64_t pos = it + static_cast<int64_t>(wave) * (64 * kItems) + j * 64 + lane;
      bkt[j]            = bucket_of(ids, pos, end, s_off, world, owners, id[j]);
      rank[j]           = 0;
      uint64_t pending  = __ballot(bkt[j] >= 0);
      while (pending) {
        int leader    = __ffsll(static_cast<long long>(pending)) - 1;
        int lb        = __shfl(bkt[j], leader, 64);
        uint64_t mask = __ballot(bkt[j] == lb);
        int prior     = s_wcnt[wave][lb];  // ids of this bucket in earlier groups of this wave
        if (bkt[j] == lb) rank[j] = prior + __popcll(mask & lt_mask);
        if (lane == leader) s_wcnt[wave][lb] = prior + __popcll(mask);
        pending &= ~mask;
      }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += kBlock) {
      int64_t run = s_run[b];
#pragma unroll
      for (int w = 0; w < kWaves; w++) {
        s_wbase[w][b] = run;
        run += s_wcnt[w][b];
      }
      s_run[b] = run;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kItems; j++) {
      if (bkt[j] >= 0) {
        const int64_t pos  = it + static_cast<int64_t>(wave) * (64 * kItems) + j * 64 + lane;
        const int64_t dest = s_wbase[wave][bkt[j]] + rank[j];
        bucketed_ids[dest] = id[j];
        raw_indices[dest]  = pos;
      }
    }
    // s_wcnt is re-zeroed behind the barrier at the top of the next iteration; s_wbase is only
    // rewritten after that barrier too
  }
}

template <typename IdxT>
int run_bucket(const wm_bucket_args* a, hipStream_t stream)
{
  bucket_geom g         = geometry(a->n);
  int64_t* block_counts = static_cast<int64_t*>(a->workspace);
  const IdxT* ids       = static_cast<const IdxT*>(a->indices);
  const int owners      = a->owner_count > 0 ? a->owner_count : a->world_size;
  const char* de        = WM_AB_KNOB("WM_BUCKET_DENSE");   // 0: the peel loop for every world size (A/B)
  const bool dense      = a->world_size + 1 <= kDenseBuckets && !(de != nullptr && de[0] == '0');
  if (!(a->reuse_scan && a->bucketed_ids != nullptr)) {  // (reuse: the counts-only call over the same ids left the scan behind)
    if (dense)
      hipLaunchKernelGGL((bucket_hist_kernel<IdxT, true>), dim3(g.blocks), dim3(kBlock), 0, stream, ids, a->n, a->entry_offsets,
                         a->world_size, owners, g.chunk, block_counts);
    else
      hipLaunchKernelGGL((bucket_hist_kernel<IdxT, false>), dim3(g.blocks), dim3(kBlock), 0, stream, ids, a->n, a->entry_offsets,
                         a->world_size, owners, g.chunk, block_counts);
    hipLaunchKernelGGL(bucket_scan_kernel, dim3(1), dim3(1024), 0, stream, block_counts, g.blocks, a->world_size,
                       a->counts);
  }
  if (a->bucketed_ids != nullptr && a->raw_indices != nullptr) {
    if (dense)
      hipLaunchKernelGGL((bucket_scatter_kernel<IdxT, true>), dim3(g.blocks), dim3(kBlock), 0, stream, ids, a->n,
                         a->entry_offsets, a->world_size, owners, g.chunk, block_counts,
                         static_cast<IdxT*>(a->bucketed_ids), a->raw_indices);
    else
      hipLaunchKernelGGL((bucket_scatter_kernel<IdxT, false>), dim3(g.blocks), dim3(kBlock), 0, stream, ids, a->n,
                         a->entry_offsets, a->world_size, owners, g.chunk, block_counts,
                         static_cast<IdxT*>(a->bucketed_ids), a->raw_indices);
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- duplicate estimate (linear counting over a sample) ---------------------------------------------------------
// One BYTE per hash slot, set with a plain store: a hot id (Zipf: one id is 5 % of the batch) hits the same slot tens of
// thousands of times, and atomics to one word serialise (the bitmap + atomicOr version took 0.48 ms on the skewed batch,
// 25 us on the uniform one); racing plain stores of the same value just merge.
constexpr int kDupSlots = 1 << 24;          // 16 MiB of flags: stays in the Infinity Cache; <= ~1 M sampled ids (6 % full)
constexpr int64_t kDupSampleTarget = 1 << 19;

__device__ __forceinline__ uint32_t mix64(uint64_t x)
{  // murmur3 finaliser
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return static_cast<uint32_t>(x);
}

template <typename IdxT>
__global__ void dup_sample_kernel(const IdxT* ids, int64_t n, int64_t step, int64_t sampled, uint8_t* flags)
{
  const int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= sampled) return;
  const int64_t id = static_cast<int64_t>(ids[min(j * step, n - 1)]);
  flags[mix64(static_cast<uint64_t>(id)) & (kDupSlots - 1)] = 1;
}

__global__ __launch_bounds__(1024) void dup_count_kernel(const uint32_t* flag_words, int64_t sampled, unsigned long long* set_slots,
                                                         unsigned int* blocks_done, int64_t* permille)
{
  __shared__ int s_sum[16];
  int local = 0;  // flags are 0 / 1 bytes: the population count of a word is the number of set slots in it
  for (int i = blockIdx.x * 1024 + threadIdx.x; i < kDupSlots / 4; i += gridDim.x * 1024) local += __popc(flag_words[i]);
  for (int d = 32; d > 0; d >>= 1) local += __shfl_down(local, d, 64);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    int total = 0;
    for (int w = 0; w < 16; w++) total += s_sum[w];
    atomicAdd(set_slots, static_cast<unsigned long long>(total));
    __threadfence();
    if (atomicAdd(blocks_done, 1u) == gridDim.x - 1) {  // last block: linear counting, distinct ~ -M ln(1 - set / M)
      const double M        = static_cast<double>(kDupSlots);
      const double set      = static_cast<double>(atomicAdd(set_slots, 0ull));
      const double distinct = set >= M ? M : -M * log(1.0 - set / M);
      double dup            = 1.0 - distinct / static_cast<double>(sampled);
      dup                   = dup < 0.0 ? 0.0 : dup;
      *permille             = static_cast<int64_t>(dup * 1000.0 + 0.5);
    }
  }
}

template <typename IdxT>
__global__ void sorted_owner_counts_kernel(const IdxT* sorted, const int64_t* n_dev, const uint64_t* entry_offsets, int world,
                                           int64_t* counts)
{
  // thread r: lower bounds of entry_offsets[r] and entry_offsets[r + 1] in the ids, compared as UNSIGNED keys (the order
  // they were sorted in; a negative id is a huge key and stays outside every range)
  using UT = typename std::make_unsigned<IdxT>::type;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= world) return;
  const int64_t n = *n_dev;
  auto lower      = [&](uint64_t key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (static_cast<uint64_t>(static_cast<UT>(sorted[mid])) < key) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  counts[r] = lower(entry_offsets[r + 1]) - lower(entry_offsets[r]);
}

}  // namespace

size_t hip_dup_estimate_workspace_bytes(int64_t) { return static_cast<size_t>(kDupSlots) + 64; }

int hip_dup_estimate(const void* ids, wholememory_dtype_t index_dtype, int64_t n, void* workspace, int64_t* permille_dev,
                     void* stream_v)
{
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (index_dtype != WHOLEMEMORY_DT_INT && index_dtype != WHOLEMEMORY_DT_INT64) return -1;
  if (n < 2) return hipMemsetAsync(permille_dev, 0, sizeof(int64_t), stream) == hipSuccess ? 0 : -2;
  auto* flags       = static_cast<uint8_t*>(workspace);
  auto* set_slots   = reinterpret_cast<unsigned long long*>(flags + kDupSlots);
  auto* blocks_done = reinterpret_cast<unsigned int*>(set_slots + 1);
  if (hipMemsetAsync(workspace, 0, static_cast<size_t>(kDupSlots) + 64, stream) != hipSuccess) return -2;
  const int64_t step    = std::max<int64_t>(1, n / kDupSampleTarget);
  const int64_t sampled = (n + step - 1) / step;
  const int blocks      = static_cast<int>((sampled + 255) / 256);
  if (index_dtype == WHOLEMEMORY_DT_INT)
    hipLaunchKernelGGL((dup_sample_kernel<int32_t>), dim3(blocks), dim3(256), 0, stream, static_cast<const int32_t*>(ids), n, step,
                       sampled, flags);
  else
    hipLaunchKernelGGL((dup_sample_kernel<int64_t>), dim3(blocks), dim3(256), 0, stream, static_cast<const int64_t*>(ids), n, step,
                       sampled, flags);
  hipLaunchKernelGGL(dup_count_kernel, dim3(64), dim3(1024), 0, stream, reinterpret_cast<const uint32_t*>(flags), sampled,
                     set_slots, blocks_done, permille_dev);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int hip_sorted_owner_counts(const void* sorted_ids, wholememory_dtype_t index_dtype, const int64_t* n_dev, int64_t,
                            const uint64_t* entry_offsets, int world_size, int64_t* counts, void* stream_v)
{
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  const int blocks   = (world_size + 63) / 64;
  if (index_dtype == WHOLEMEMORY_DT_INT)
    hipLaunchKernelGGL((sorted_owner_counts_kernel<int32_t>), dim3(blocks), dim3(64), 0, stream,
                       static_cast<const int32_t*>(sorted_ids), n_dev, entry_offsets, world_size, counts);
  else if (index_dtype == WHOLEMEMORY_DT_INT64)
    hipLaunchKernelGGL((sorted_owner_counts_kernel<int64_t>), dim3(blocks), dim3(64), 0, stream,
                       static_cast<const int64_t*>(sorted_ids), n_dev, entry_offsets, world_size, counts);
  else
    return -1;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

namespace {
}  // namespace

size_t hip_bucket_workspace_bytes(int64_t n, int world_size)
{
  bucket_geom g = geometry(std::max<int64_t>(n, 1));
  return static_cast<size_t>(world_size + 2) * static_cast<size_t>(g.blocks) * sizeof(int64_t) + 256;
}

int hip_bucket_ids(const wm_bucket_args* a, void* stream_v)
{
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (a->world_size < 1 || a->world_size + 1 > kMaxBuckets) return -1;
  if (a->owner_count > kMaxOwners || (a->owner_count > 0 && a->owner_count < a->world_size)) return -1;
  if (a->n == 0) {  // reference bucket_ids_func.cu:121-122: counts zeroed, nothing launched
    return hipMemsetAsync(a->counts, 0, sizeof(int64_t) * a->world_size, stream) == hipSuccess ? 0 : -2;
  }
  if (a->index_dtype == WHOLEMEMORY_DT_INT) return run_bucket<int32_t>(a, stream);
  if (a->index_dtype == WHOLEMEMORY_DT_INT64) return run_bucket<int64_t>(a, stream);
  return -1;
}


// ---- chunk-major copy of per-peer segments (backend.hpp: permute_chunks) ------------------------------------------------
namespace {
constexpr int kPermuteSegs = 16;
struct permute_table {
  int64_t off[kPermuteSegs], cnt[kPermuteSegs];
  int n_segs, n_chunks;
};
template <typename T>
__global__ __launch_bounds__(256) void permute_chunks_kernel(const T* src, T* dst, permute_table t, int64_t total)
{
  const int64_t k = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (k >= total) return;
  // which chunk, which segment: at most n_chunks + n_segs steps over kernel arguments (scalar registers)
  int64_t base = 0;
  for (int c = 0; c < t.n_chunks; c++) {
    for (int p = 0; p < t.n_segs; p++) {
      const int64_t a = t.cnt[p] * c / t.n_chunks, b = t.cnt[p] * (c + 1) / t.n_chunks;
      if (k < base + (b - a)) {
        dst[k] = src[t.off[p] + a + (k - base)];
        return;
      }
      base += b - a;
    }
  }
}
}  // namespace

int hip_permute_chunks(const void* src, void* dst, int elt_bytes, const int64_t* seg_offsets, const int64_t* seg_counts, int n_segs,
                       int n_chunks, void* stream_v)
{
  if (n_segs < 1 || n_segs > kPermuteSegs || n_chunks < 1 || (elt_bytes != 4 && elt_bytes != 8)) return -1;
  permute_table t{};
  int64_t total = 0;
  for (int p = 0; p < n_segs; p++) t.off[p] = seg_offsets[p], t.cnt[p] = seg_counts[p], total += seg_counts[p];
  t.n_segs = n_segs, t.n_chunks = n_chunks;
  if (total == 0) return 0;
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  const dim3 grid(static_cast<unsigned>((total + 255) / 256)), block(256);
  if (elt_bytes == 4)
    hipLaunchKernelGGL(permute_chunks_kernel<uint32_t>, grid, block, 0, stream, static_cast<const uint32_t*>(src),
                       static_cast<uint32_t*>(dst), t, total);
  else
    hipLaunchKernelGGL(permute_chunks_kernel<uint64_t>, grid, block, 0, stream, static_cast<const uint64_t*>(src),
                       static_cast<uint64_t*>(dst), t, total);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace wm
