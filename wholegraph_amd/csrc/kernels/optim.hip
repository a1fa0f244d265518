// wholegraph_amd — owner-side gradient de-duplication and sparse optimizer steps (gfx950 HIP).
//
// Reference behaviour (cpp/src/wholememory_ops/functions/):
//   exchange_embeddings_nccl_func.cu:76-174   stable radix sort of received ids (payload = receive
//        position) -> unique_by_key -> per unique id a SEQUENTIAL fp32 sum of its gradient rows in
//        sorted order, written to a dedup buffer;
//   embedding_optimizer_func.cu:178-224 (SGD), :331-421 (lazy Adam/AdamW), :594-658 (AdaGrad),
//        :791-856 (RMSProp): one block per unique id reads the dedup row and updates the local shard.
// MI355X design: the dedup buffer never exists. After the sort, ONE kernel walks each run of equal
// ids, accumulates the duplicates' rows in registers in exactly the reference's order (first row
// copied, the rest added one by one in receive order) and applies the optimizer to the table row
// in the same pass: per unique row the HBM traffic is grad rows + 1 table read + 1 table write
// (+ state), instead of grad rows + dedup write + dedup read + table RMW. Each wave owns one unique
// id at a time (grid-stride over runs; the run count is read from device memory, so the host
// never synchronises to learn it). All arithmetic is separate fp32 multiplies and adds in the
// statement order of the reference kernels (this TU is compiled with -ffp-contract=off); results
// are bit-identical to oracle/wm_oracle.c. Element-wise math: VALU work, HBM-bound; no MFMA.
//
// The id sort itself is rocPRIM's radix sort restricted to the significant key bits
// (ids at the owner are non-negative and < table rows): same stable order as the reference's
// full-width signed sort, fewer passes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

// rocPRIM's radix sort copies its inputs to scratch first whenever input and output "can alias", and answers "yes" for
// every iterator that is not a plain pointer (detail/various.hpp) — 26 us per 10 M (key, position) pairs for the narrowing
// key iterator and the counting payload used below. The exact answers for those two, declared before the sort's templates
// are defined (the call there is a qualified name: only overloads visible at that point take part):
#include <iterator>
#include <rocprim/config.hpp>
#include <rocprim/detail/various.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
namespace wm {
// random-access iterator over ids[i] - base, narrowed to 32 bits (rocprim::transform_iterator keeps its pointer private).
// An id outside [base, base + span) — negative ("skip me") or past the range — reads as the key `span`: all such ids sort
// behind every real key, as ONE run that the run detection drops (see run_dedup).
template <typename InT>
struct narrow_key_iterator {
  using value_type        = uint32_t;
  using reference         = uint32_t;
  using pointer           = const uint32_t*;
  using difference_type   = std::ptrdiff_t;
  using iterator_category = std::random_access_iterator_tag;
  const InT* ptr;  // InT is the UNSIGNED index type: a negative id is a huge offset
  InT base;
  uint32_t span;
  __host__ __device__ uint32_t key(InT v) const
  {
    const InT off = v - base;
    return off < static_cast<InT>(span) ? static_cast<uint32_t>(off) : span;
  }
  __host__ __device__ uint32_t operator*() const { return key(*ptr); }
  __host__ __device__ uint32_t operator[](difference_type i) const { return key(ptr[i]); }
  __host__ __device__ narrow_key_iterator operator+(difference_type d) const { return {ptr + d, base, span}; }
  __host__ __device__ narrow_key_iterator operator-(difference_type d) const { return {ptr - d, base, span}; }
  __host__ __device__ difference_type operator-(const narrow_key_iterator& o) const { return ptr - o.ptr; }
  __host__ __device__ narrow_key_iterator& operator+=(difference_type d) { ptr += d; return *this; }
  __host__ __device__ narrow_key_iterator& operator-=(difference_type d) { ptr -= d; return *this; }
  __host__ __device__ narrow_key_iterator& operator++() { ++ptr; return *this; }
  __host__ __device__ narrow_key_iterator operator++(int) { narrow_key_iterator t = *this; ++ptr; return t; }
  __host__ __device__ narrow_key_iterator& operator--() { --ptr; return *this; }
  __host__ __device__ narrow_key_iterator operator--(int) { narrow_key_iterator t = *this; --ptr; return t; }
  __host__ __device__ bool operator==(const narrow_key_iterator& o) const { return ptr == o.ptr; }
  __host__ __device__ bool operator!=(const narrow_key_iterator& o) const { return ptr != o.ptr; }
  __host__ __device__ bool operator<(const narrow_key_iterator& o) const { return ptr < o.ptr; }
};
}  // namespace wm
BEGIN_ROCPRIM_NAMESPACE
namespace detail {
template <class InT, class Out>
inline bool can_iterators_alias(::wm::narrow_key_iterator<InT> it, Out* out, const size_t size)
{
  return can_iterators_alias(it.ptr, out, size);  // the ids array against the output array
}
template <class I, class D, class Out>
inline bool can_iterators_alias(counting_iterator<I, D>, Out*, const size_t)
{
  return false;  // generates its values, reads no memory
}
}  // namespace detail
END_ROCPRIM_NAMESPACE
#include <rocprim/rocprim.hpp>

#include "../knobs.hpp"
#include "../backend.hpp"
#include "../wm_common.hpp"
#include "device_common.cuh"
#include "onesweep.cuh"
#include "split_sort.cuh"
#include "long_dense.cuh"

#include <atomic>
#include <mutex>
#include <wholememory/embedding.h>
#include <wholememory/wholegraph_amd_ext.h>

namespace wm {
namespace {

constexpr int kBlock = 256;

// Run detection over the sorted keys, three small launches instead of head-flags + a device-wide scan + compaction
// (10 M keys: 185 us -> ~65 us): a tile is kRunTile consecutive sorted keys;
//   run_count_kernel   heads per tile (a head = first key, or a key that differs from its predecessor)
//   run_scan_kernel    one workgroup: exclusive prefix of the tile counts (a few thousand numbers) + the total
//   run_compact_kernel recomputes its tile's heads, ranks them with a block scan on top of the tile's prefix and writes
//                      unique_ids[rank] (widened to the caller's index type) and run_starts[rank]; the last tile adds the
//                      closing run_starts entry and the count.
// No tile waits for another one (a chained look-back scan serialises on the prefix hand-over while thousands of tiles are
// resident), and the keys are read twice out of L2 / Infinity Cache rather than flags and ranks written and re-read.
constexpr int kRunItems = 8;
constexpr int kRunTile  = kBlock * kRunItems;  // 2048 keys

template <typename KeyT>
__device__ __forceinline__ int tile_heads(const KeyT* sorted, int64_t n, int64_t base, bool head[kRunItems], KeyT key[kRunItems])
{
  // thread t owns keys base + t * kRunItems + [0, kRunItems): contiguous, so its heads are already in rank order
  const int64_t first = base + static_cast<int64_t>(threadIdx.x) * kRunItems;
  KeyT prev           = first > 0 && first <= n ? sorted[first - 1] : KeyT(0);
  int heads           = 0;
  if (first + kRunItems <= n && (reinterpret_cast<uint64_t>(sorted) & 15) == 0) {
    // the thread's 8 keys as whole 16-byte loads (the sorted array starts on a 256-byte boundary and `first` is a multiple of
    // 8 keys): element-wise guarded loads at a 32-byte lane stride made the two run-detection kernels 23 + 33 us per 10 M keys
    typedef uint32_t raw4 __attribute__((ext_vector_type(4)));
    constexpr int kVecs = static_cast<int>(sizeof(KeyT)) * kRunItems / 16;
    raw4 raw[kVecs];
#pragma unroll
    for (int v = 0; v < kVecs; v++) raw[v] = reinterpret_cast<const raw4*>(sorted + first)[v];
    __builtin_memcpy(key, raw, sizeof(KeyT) * kRunItems);
#pragma unroll
    for (int i = 0; i < kRunItems; i++) {
      head[i] = (first + i == 0) || key[i] != prev;
      prev    = key[i];
      heads += head[i] ? 1 : 0;
    }
    return heads;
  }
#pragma unroll
  for (int i = 0; i < kRunItems; i++) {
    const int64_t g = first + i;
    key[i]          = g < n ? sorted[g] : KeyT(0);
    head[i]         = g < n && (g == 0 || key[i] != prev);
    prev            = key[i];
    heads += head[i] ? 1 : 0;
  }
  return heads;
}

__device__ __forceinline__ int block_exclusive_sum(int v, int* total)
{
  __shared__ int wave_sums[kBlock / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wave_sums[wv] = incl;
  __syncthreads();
  int before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; w++) {
    if (w < wv) before += wave_sums[w];
    all += wave_sums[w];
  }
  __syncthreads();
  *total = all;
  return before + incl - v;
}

// `gate` (all three kernels): device word, 0 = the split sort has already written the runs, return at once (nullptr: always run)
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void run_count_kernel(const KeyT* sorted, int64_t n, int32_t* tile_counts, const uint32_t* gate)
{
  if (gate != nullptr && *gate == 0u) return;
  // (a bounded grid that walks the tiles: gated off, a launch of 4883 workgroups that only read the gate still takes ~7 us of
  // the machine, of 1024 about 3)
  const int n_tiles = static_cast<int>((n + kRunTile - 1) / kRunTile);
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    bool head[kRunItems];
    KeyT key[kRunItems];
    const int heads = tile_heads(sorted, n, static_cast<int64_t>(tile) * kRunTile, head, key);
    int total;
    (void)block_exclusive_sum(heads, &total);
    if (threadIdx.x == 0) tile_counts[tile] = total;
  }
}

// `last_key` / `drop_key`: when the LAST sorted key equals drop_key (the out-of-range marker of narrow_key_iterator), its run —
// the last one — does not count (last_key == nullptr: nothing is dropped)
__global__ __launch_bounds__(1024) void run_scan_kernel(int32_t* tile_counts, int n_tiles, int64_t* n_unique,
                                                        const uint32_t* last_key, uint32_t drop_key, const uint32_t* gate)
{
  if (gate != nullptr && *gate == 0u) return;
  // one workgroup walks the tile counts in chunks of 1024, carrying the running total
  __shared__ int wave_sums[16];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int base = 0; base < n_tiles; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < n_tiles ? tile_counts[i] : 0;
    int incl    = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wave_sums[wv] = incl;
    __syncthreads();
    int before = carry_s;
    for (int w = 0; w < wv; w++) before += wave_sums[w];
    if (i < n_tiles) tile_counts[i] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_unique = carry_s - (last_key != nullptr && *last_key == drop_key ? 1 : 0);
}

template <typename KeyT, typename OutT>
__global__ __launch_bounds__(kBlock) void run_compact_kernel(const KeyT* sorted, int64_t n, const int32_t* tile_prefix,
                                                             const int64_t* n_unique, OutT* unique_ids, int32_t* run_starts,
                                                             OutT key_base, bool drop_last, KeyT drop_key, const uint32_t* gate)
{
  if (gate != nullptr && *gate == 0u) return;
  // heads are ranked inside the tile, parked in LDS at their rank and written out as two coalesced streams (a thread's
  // own heads are kRunItems apart in rank order: written directly they cost a scattered store per item — 82 us vs ~35)
  __shared__ KeyT s_key[kRunTile];
  __shared__ int32_t s_pos[kRunTile];
  const int n_tiles = static_cast<int>((n + kRunTile - 1) / kRunTile);
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    bool head[kRunItems];
    KeyT key[kRunItems];
    const int64_t base = static_cast<int64_t>(tile) * kRunTile;
    const int heads    = tile_heads(sorted, n, base, head, key);
    int total;
    int rank = block_exclusive_sum(heads, &total);
#pragma unroll
    for (int i = 0; i < kRunItems; i++) {
      if (head[i]) {
        s_key[rank] = key[i];
        s_pos[rank] = static_cast<int32_t>(base + static_cast<int64_t>(threadIdx.x) * kRunItems + i);
        rank++;
      }
    }
    __syncthreads();
    const int64_t out0 = tile_prefix[tile];
    for (int i = threadIdx.x; i < total; i += kBlock) {
      // ids are stored in the caller's (signed) index type: the keys are its two's-complement bits, possibly narrowed to
      // 32 bits when the caller bounded them (then they are non-negative and the widening is exact)
      const KeyT k         = s_key[i];
      // (key_base: the keys were sorted relative to the first row of the owner's range, see run_dedup)
      unique_ids[out0 + i] = (sizeof(KeyT) == sizeof(OutT) ? static_cast<OutT>(k) : static_cast<OutT>(static_cast<uint64_t>(k))) + key_base;
      run_starts[out0 + i] = s_pos[i];
    }
    // end marker — unless the out-of-range run was dropped: then its own start (written above, at index *n_unique) ends the
    // last real run
    if (tile == n_tiles - 1 && threadIdx.x == 0 && !(drop_last && sorted[n - 1] == drop_key))
      run_starts[*n_unique] = static_cast<int32_t>(n);
    __syncthreads();   // s_key / s_pos are reused by the block's next tile
  }
}

inline unsigned significant_bits(int64_t upper_bound, unsigned full)
{
  if (upper_bound <= 0) return full;
  unsigned b = 1;
  while (b < full && (static_cast<uint64_t>(upper_bound - 1) >> b) != 0) b++;
  return b;
}

// keys narrowed on the fly: ids the caller bounded to a range of less than 2^32 rows are sorted as 32-bit keys RELATIVE to
// the start of the range (8 + 4 bytes per element and pass instead of 8 + 8, and only the bits of the range's width: a
// 125 M-row shard of a 1 B-row table sorts 27 bits in 3 passes, not 30 in 4) ... the first pass reads the ids through this
// iterator (narrow_key_iterator, top of the file), no conversion pass

// rocPRIM's onesweep with 9 radix bits per pass and 1024 x 8 keys per workgroup: ids of a 100 M-row shard (27 bits) sort
// in 3 passes instead of the tuned default's 4 x 8 bits (10 M (key, position) pairs: 398 -> 251 us;
// experiments/sort_variants.hip has the sweep). 64-bit keys keep the library default.
constexpr int64_t kSortRadixMin = 3 << 16;   // 196608 (crossover between 131072 and 262144 items: profiles/r04_small_sort.txt)
template <typename SortKeyT>
struct sort_config {
  using type = rocprim::default_config;
};
template <>
struct sort_config<uint32_t> {
  using type = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 12>, rocprim::kernel_config<1024, 8>, 9,
                                        rocprim::block_radix_rank_algorithm::match>>;
};

// rocPRIM sorts up to 2^20 items with a MERGE sort (radix_sort_config's MergeSortLimit): block sort + 10 merge passes of two
// launches each for 1 M (key, position) pairs — 160 us where three onesweep passes over 24 bits take ~70 (the batch of a cached
// C1 gather, the gradients of a 1024-seed mini-batch: profiles/r04_small_sort.txt). The second configuration never merges;
// sort_pairs32 picks by size (WM_SORT_RADIX_MIN, items from which the radix passes are taken).
struct sort_config_radix32 {
  using type = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 12>, rocprim::kernel_config<1024, 8>, 9,
                                        rocprim::block_radix_rank_algorithm::match>,
    0>;
};
inline int64_t sort_radix_min()
{
  const char* e = WM_KNOB("WM_SORT_RADIX_MIN");
  return e != nullptr && atoll(e) > 0 ? atoll(e) : kSortRadixMin;
}
template <typename KeysIn, typename ValsIn>
hipError_t sort_pairs32(void* temp, size_t& temp_bytes, KeysIn keys, uint32_t* keys_out, ValsIn vals, int32_t* vals_out, size_t n,
                        unsigned lo, unsigned hi, hipStream_t stream)
{
  if (temp == nullptr) {   // size query: room for either configuration
    size_t a = 0, b = 0;
    hipError_t e = rocprim::radix_sort_pairs<sort_config<uint32_t>::type>(nullptr, a, keys, keys_out, vals, vals_out, n, lo, hi, stream);
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_pairs<sort_config_radix32::type>(nullptr, b, keys, keys_out, vals, vals_out, n, lo, hi, stream);
    temp_bytes = std::max(a, b);
    return e;
  }
  if (static_cast<int64_t>(n) >= sort_radix_min())
    return rocprim::radix_sort_pairs<sort_config_radix32::type>(temp, temp_bytes, keys, keys_out, vals, vals_out, n, lo, hi, stream);
  return rocprim::radix_sort_pairs<sort_config<uint32_t>::type>(temp, temp_bytes, keys, keys_out, vals, vals_out, n, lo, hi, stream);
}

template <typename SortKeyT>
struct dedup_layout {
  SortKeyT* sorted;
  int32_t* tile_counts;
  void* temp;
  size_t temp_bytes;
  size_t total;
};

// ---- the split sort (split_sort.cuh) in front of the generic sort --------------------------------------------------------
// Batches of bounded ids (the owner's row range is known) from kSplitMin ids up take the two-stage split sort; what it cannot
// take — a bucket that does not fit LDS — it finds out on the device, so the generic path (onesweep.cuh + the run detection
// above, every kernel gated on the split sort's overflow word) is enqueued behind it either way: ~7 launches that return at
// once in the usual case. WM_DEDUP_SPLIT=0 switches the split sort off, WM_DEDUP_SPLIT_MIN moves the threshold.
constexpr int64_t kSplitMin = 1 << 16;
constexpr int kOswBlock = 512, kOswIpt = 16;   // profiles/r03_onesweep_ab.txt
inline int64_t split_min()
{
  const char* off = WM_KNOB("WM_DEDUP_SPLIT");
  if (off != nullptr && off[0] == '0') return INT64_MAX;
  const char* e = WM_KNOB("WM_DEDUP_SPLIT_MIN");
  return e != nullptr && atoll(e) > 0 ? atoll(e) : kSplitMin;
}
// The generic path's ~7 launches return at once in the usual case, but each still costs 5-8 us of the stream's time (47 us per
// call in profiles/r05_grad_timeline_serial.txt). They depend on nothing but the overflow word (known after the split sort's
// SECOND kernel), so they go to a side stream that forks there and joins after the split sort's last kernel: idle, they hide
// under its two long kernels; when the batch overflowed, those two return at once and the caller's stream waits for the side.
constexpr int kMaxDevices = 64;
template <typename Lane>
Lane& per_device()
{
  static std::mutex mu;
  static Lane* lanes[kMaxDevices] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
  std::lock_guard<std::mutex> lk(mu);
  if (lanes[dev] == nullptr) lanes[dev] = new Lane();
  return *lanes[dev];
}
__global__ void waits_probe_set_kernel(uint32_t* word) { __hip_atomic_store(word, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
struct sort_lane {
  std::mutex mu;   // one fork .. join sequence at a time: the events are shared
  // Do two kernels on two streams of this device RUN side by side? Every device-side wait of the gradient path needs that (a wave
  // polls a word another stream's kernel sets). Measured once per device, when the lane is made: a one-wave kernel waits ~10 ms
  // at most for a word that a kernel on the other stream sets. A tool that executes one kernel at a time (any tool: round 5
  // only knew rocprofv3's counter collection by its environment variables) lets the wait give up -> events only, for the
  // life of the process, one WARN line.
  bool waits_work = true;
  hipStream_t stream = nullptr, stream_high = nullptr;
  hipEvent_t forked = nullptr, joined = nullptr;
  bool ok = false;
  hipStream_t side() const
  {
    const char* pe = WM_KNOB("WM_DEDUP_LANE_PRIO");
    return pe != nullptr && pe[0] == 'h' ? stream_high : stream;
  }
  // Fork without an event on the caller's stream: the split sort's scan kernel publishes "verdict final" = the sort's sequence
  // number in one word of this ring (zero at the start, values only grow, a word comes round again after kRing sorts), and the
  // side stream's first kernel waits for it (split_wait_kernel). An event recorded between the scan and the scatter kernel
  // delayed the scatter kernel by ~7 us on every call (profiles/r05_grad_timeline_split_sort.txt: "gap 7.1").
  static constexpr uint32_t kRing = 4096;
  uint32_t* ring = nullptr;
  uint32_t seq   = 0;
  // Which sort a batch gets follows the batches before it (run_dedup: "adaptive route"). Per row range (lower, upper) of the
  // sorted ids: did the last split sort (or probe) overflow a bucket? The word is copied to pinned memory behind the kernels
  // that decide it and read by the host without synchronising, so it lags a call.
  static constexpr int kAdapt = 8, kProbeEvery = 4;
  struct adapt_entry {
    int64_t lower = -1, upper = -1;
    unsigned calls = 0;
    int mode = 0;   // round 6: 0 the plain split sort, 1 the split sort with hot ids peeled (split::launch_hot), 2 rocPRIM's sort
  };
  adapt_entry adapt[kAdapt];
  // pinned, three words per row range: [i] the overflow word of its last PLAIN split sort, [kAdapt + i] of its last HOT split
  // sort or probe, [2 kAdapt + i] how many ids that one peeled
  volatile int32_t* adapt_flags = nullptr;
  unsigned adapt_generation     = ~0u;       // a knob reload forgets what was learnt (tests, A/B runs)
  int adapt_next                = 0;
  void* probe_ws                = nullptr;   // split::probe_workspace_bytes(), allocated at the first probe
  int adapt_slot(int64_t lower, int64_t upper)
  {
    if (adapt_flags == nullptr) {
      void* h = nullptr;
      if (hipHostMalloc(&h, 3 * kAdapt * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) return -1;
      adapt_flags = static_cast<volatile int32_t*>(h);
      for (int i = 0; i < 3 * kAdapt; i++) adapt_flags[i] = 0;
    }
    const unsigned g = g_knob_generation.load(std::memory_order_acquire);
    if (g != adapt_generation) {
      for (int i = 0; i < kAdapt; i++) adapt[i] = adapt_entry{}, adapt_flags[i] = adapt_flags[kAdapt + i] = adapt_flags[2 * kAdapt + i] = 0;
      adapt_generation = g;
    }
    for (int i = 0; i < kAdapt; i++)
      if (adapt[i].lower == lower && adapt[i].upper == upper) return i;
    const int i = adapt_next;
    adapt_next  = (adapt_next + 1) % kAdapt;
    adapt[i]    = adapt_entry{lower, upper, 0, 0};
    adapt_flags[i] = adapt_flags[kAdapt + i] = adapt_flags[2 * kAdapt + i] = 0;
    return i;
  }
  // Device-side waits that gave up (split_sort.cuh: wait_cfg) leave their code in this word of pinned, device-mapped host
  // memory — split_join_kernel ORs it in after it has turned the failed sort into "no runs", split_wait_kernel when it stops
  // waiting. The host looks at it (take_error) whenever it enters the sort or the join and after every synchronise of the
  // gradient path (backend: device_error): a stalled wait becomes WHOLEMEMORY_CUDA_ERROR with one ERROR line, never a step
  // applied over runs that were not final.
  volatile uint32_t* host_err = nullptr;   // pinned
  uint32_t* host_err_dev      = nullptr;   // the same word as the device addresses it
  uint32_t take_error()
  {
    if (host_err == nullptr) return 0;
    const uint32_t e = *host_err;
    if (e != 0) *host_err = 0;
    return e;
  }
  sort_lane()
  {
    void* h = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess) {
      void* d = nullptr;
      if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
        host_err     = static_cast<volatile uint32_t*>(h);
        *host_err    = 0;
        host_err_dev = static_cast<uint32_t*>(d);
      } else {
        (void)hipHostFree(h);
      }
    }
    // two streams, plain and highest priority; WM_DEDUP_LANE_PRIO=n|h picks one per call (see side())
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    ok = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess &&
         hipStreamCreateWithPriority(&stream_high, hipStreamNonBlocking, greatest) == hipSuccess &&
         hipEventCreateWithFlags(&forked, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&joined, hipEventDisableTiming) == hipSuccess;
    if (ok) {
      void* r = nullptr;
      if (hipMalloc(&r, kRing * sizeof(uint32_t)) == hipSuccess) {
        if (hipMemsetAsync(r, 0, kRing * sizeof(uint32_t), stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess)
          ring = static_cast<uint32_t*>(r);
        else
          (void)hipFree(r);
      }
    }
    if (ok) {
      uint32_t* probe = nullptr;   // [0] the word waited for, [1] the waiter's error word
      if (hipMalloc(reinterpret_cast<void**>(&probe), 2 * sizeof(uint32_t)) == hipSuccess) {
        uint32_t err = 1;
        if (hipMemsetAsync(probe, 0, 2 * sizeof(uint32_t), stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess) {
          hipLaunchKernelGGL(split::split_wait_kernel, dim3(1), dim3(64), 0, stream, probe, 1u, probe + 1, 6000u, static_cast<uint32_t*>(nullptr));
          hipLaunchKernelGGL(waits_probe_set_kernel, dim3(1), dim3(1), 0, stream_high, probe);
          if (hipStreamSynchronize(stream_high) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess &&
              hipMemcpy(&err, probe + 1, sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess && err != 0) {
            waits_work = false;
            WM_WARN("kernels on two streams of this device do not run side by side (a tool that executes one kernel at a time?): "
                    "the gradient path synchronises its side streams with events only");
          }
        }
        (void)hipFree(probe);
      }
    }
  }
  // one lane per DEVICE (round 6; a process-wide one put device 0's streams, ring and events under device 1's kernels): created
  // on the device that is current at the first call that needs it there, kept for the life of the process
  static sort_lane& get() { return per_device<sort_lane>(); }
};
// Kernels that wait for a word another stream's kernel sets (split_wait_kernel, split_join_kernel) need that other kernel to be
// able to RUN beside them. A tool that lets one kernel execute at a time — rocprofv3's counter collection does, and not in
// submission order: the --pmc passes of scripts/collect_profiles.sh sat in a waiter until their timeout — turns every such
// wait into a hang. So: events only (the round's first arrangement, ~20 us slower per call) when rocprofv3 collects counters
// (it exports ROCPROF_COUNTER_COLLECTION to the profiled process) or when WM_DEVICE_WAITS=0 says so.
inline bool device_waits_allowed()
{
  const char* e = WM_KNOB("WM_DEVICE_WAITS");
  if (e != nullptr) return e[0] != '0';
  if (WM_KNOB("ROCPROF_COUNTER_COLLECTION") != nullptr || WM_KNOB("ROCPROF_COUNTERS") != nullptr) return false;
  return sort_lane::get().waits_work;   // (round 6: measured per device, whatever the tool is called)
}
// limits of the device-side waits (split_sort.cuh: wait_cfg). WM_DEBUG_SPIN_LIMIT=n shortens every one of them to n polls and
// WM_DEBUG_STALL=lookback|join keeps a gate shut (a stage-2 bucket that never publishes / a generic path that never reports
// done): tests/test_dedup_split_gpu.py forces each timeout and sees the error code and an untouched table.
inline split::wait_cfg wait_limits(const split::plan* sp = nullptr)
{
  split::wait_cfg wc;
  if (const char* e = WM_KNOB("WM_DEBUG_SPIN_LIMIT")) {
    const long long v = atoll(e);
    if (v > 0 && v < (1ll << 31)) wc.look_back_polls = wc.join_polls = wc.wait_polls = static_cast<uint32_t>(v);
  }
  const char* st = WM_KNOB("WM_DEBUG_STALL");
  if (st != nullptr && st[0] == 'l' && sp != nullptr) wc.stall_bucket = sp->buckets / 2;
  return wc;
}
inline bool stall_join() { const char* st = WM_KNOB("WM_DEBUG_STALL"); return st != nullptr && st[0] == 'j'; }
std::atomic<int64_t> g_split_sorts{0};
std::atomic<int64_t> g_hot_split_sorts{0};   // ... of them with hot ids peeled (split::launch_hot)
// The optimizer step that follows a split sort on the same thread finds the sort's control words through the run_starts array
// both were given: its long-run counters live there (zeroed by the sort's first kernel: no fill in front of the step), and
// the listing kernels return at once when the sort saw neither an overflow nor a bucket with a run of more than kMaxDup ids —
// then no run is longer than kMaxDup, far below any long-run threshold (fill 5.7 + listing 13 us per call otherwise).
// The record is cleared by EVERY id sort of the thread (whatever path it takes) and consumed by the next step, and it has to
// match the three arrays a sort hands to a step — run starts, unique ids, run count — so a step can only pick up the control
// words of the sort that produced exactly its inputs, whose workspace its caller still holds.
struct last_split_record {
  const int32_t* run_starts = nullptr;
  const void* unique_ids    = nullptr;
  const int64_t* n_unique   = nullptr;
  uint32_t* ctl             = nullptr;
};
thread_local last_split_record g_last_split;
// Deferred join (backend: dedup_defer_join / dedup_join). The generic path's launches on the side stream take ~35 us even when
// they have nothing to do, longer than the split sort of a mini-batch (profiles/r05_grad_timeline_small_batch.txt). A caller
// that goes on to the optimizer step asks for the join to be split in two: a one-wave kernel on its stream that waits ON THE
// DEVICE, and only when the batch overflowed; and the event wait, enqueued after the step (dedup_join) — by then the idle
// launches have long drained under the tile kernel. The event wait still comes before anything else the caller queues, so
// the sort's workspace is not handed back while a side-stream kernel may still look at it.
thread_local bool g_defer_join     = false;
thread_local bool g_join_pending   = false;
struct split_layout {
  void* split_ws;        // split::plan offsets apply; its first two arrays double as the generic sort's second (key, position) pair
  uint32_t* sorted;      // generic path: sorted keys
  int32_t* tile_counts;  // generic path: run detection
  uint32_t* osw_ctrl;    // generic path: control words, zeroed by split_hist_kernel
  size_t osw_ctrl_words;
  size_t total;
};
inline split_layout split_carve(void* ws, int64_t n)
{
  auto align = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  split_layout l;
  char* p  = static_cast<char*>(ws);
  size_t o = 0;
  l.split_ws       = p + o, o += align(split::workspace_bound(n));
  l.sorted         = reinterpret_cast<uint32_t*>(p + o), o += align(4 * static_cast<size_t>(n));
  l.tile_counts    = reinterpret_cast<int32_t*>(p + o), o += align(4 * static_cast<size_t>((n + kBlock * 8 - 1) / (kBlock * 8) + 1));
  l.osw_ctrl_words = osw::ctrl_words_bound<kOswBlock, kOswIpt>(n);
  l.osw_ctrl        = reinterpret_cast<uint32_t*>(p + o), o += align(4 * l.osw_ctrl_words);
  l.total           = o + 256;
  return l;
}

inline int run_tiles(int64_t n) { return static_cast<int>((n + kRunTile - 1) / kRunTile); }

template <typename SortKeyT>
dedup_layout<SortKeyT> layout(void* ws, int64_t n)
{
  auto align = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  size_t sort_bytes = 0;
  if constexpr (sizeof(SortKeyT) == 4)
    (void)sort_pairs32(nullptr, sort_bytes, static_cast<const uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
                       rocprim::counting_iterator<int32_t>(0), static_cast<int32_t*>(nullptr), static_cast<size_t>(n), 0, 32, nullptr);
  else
    (void)rocprim::radix_sort_pairs<typename sort_config<SortKeyT>::type>(
      nullptr, sort_bytes, static_cast<const SortKeyT*>(nullptr), static_cast<SortKeyT*>(nullptr),
      rocprim::counting_iterator<int32_t>(0), static_cast<int32_t*>(nullptr), static_cast<size_t>(n), 0, 8 * sizeof(SortKeyT),
      nullptr);
  dedup_layout<SortKeyT> l;
  char* p       = static_cast<char*>(ws);
  size_t o      = 0;
  l.sorted      = reinterpret_cast<SortKeyT*>(p + o), o += align(sizeof(SortKeyT) * n);
  l.tile_counts = reinterpret_cast<int32_t*>(p + o), o += align(4 * static_cast<size_t>(run_tiles(n) + 1));
  l.temp        = p + o;
  l.temp_bytes  = sort_bytes;
  l.total       = o + align(l.temp_bytes) + 256;
  return l;
}

// closing kernel of the generic path behind a split sort: "done" for split_join_kernel — after the onesweep passes' error word
// (a look-back that gave up) has been folded into the split sort's, which is the one the join kernel reports
__global__ void set_word_kernel(uint32_t* word, uint32_t value, const uint32_t* osw_error, uint32_t* ctl_error)
{
  if (osw_error != nullptr && *osw_error != 0u) atomicOr(ctl_error, static_cast<uint32_t>(split::kErrOnesweep));
  __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename SortKeyT, typename OutT>
int detect_runs(const SortKeyT* sorted, int32_t* tile_counts, int64_t n, OutT* unique_ids, int32_t* run_starts,
                int64_t* n_unique_out, hipStream_t stream, OutT key_base = 0, bool drop_last = false, SortKeyT drop_key = 0,
                const uint32_t* gate = nullptr, uint32_t* done_word = nullptr, const uint32_t* osw_error = nullptr,
                uint32_t* ctl_error = nullptr)
{
  const int tiles = run_tiles(n);
  const uint32_t* last_key = nullptr;
  if constexpr (sizeof(SortKeyT) == 4) {
    if (drop_last) last_key = reinterpret_cast<const uint32_t*>(sorted + (n - 1));
  }
  // (one block per tile: looping 1024 blocks over the tiles made an ACTIVE run_compact_kernel 77 us instead of 18)
  const int run_grid = tiles;
  hipLaunchKernelGGL((run_count_kernel<SortKeyT>), dim3(run_grid), dim3(kBlock), 0, stream, sorted, n, tile_counts, gate);
  hipLaunchKernelGGL(run_scan_kernel, dim3(1), dim3(1024), 0, stream, tile_counts, tiles, n_unique_out, last_key,
                     static_cast<uint32_t>(drop_key), gate);
  hipLaunchKernelGGL((run_compact_kernel<SortKeyT, OutT>), dim3(run_grid), dim3(kBlock), 0, stream, sorted, n, tile_counts,
                     n_unique_out, unique_ids, run_starts, key_base, last_key != nullptr, drop_key, gate);
  // the generic path behind a split sort, joined on the device: one more (tiny) kernel says so when everything above has
  // finished — the end of a kernel makes its writes visible; a fence + counter per block of the kernel above made that kernel
  // 302 us instead of 18
  if (done_word != nullptr)
    hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, stream, done_word, 1u, osw_error, ctl_error);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <typename KeyT>
int run_dedup(const void* ids, int64_t n, int64_t key_upper_bound, int64_t key_lower_bound, void* unique_ids,
              int32_t* run_starts, int32_t* order, int64_t* n_unique_out, void* workspace, hipStream_t stream)
{
  using UKey = typename std::make_unsigned<KeyT>::type;
  if (key_upper_bound <= 0 || key_lower_bound < 0 || key_lower_bound >= key_upper_bound) key_lower_bound = 0;
  // the payload 0, 1, 2 ... is generated by the sort's first pass (counting iterator): no iota array
  rocprim::counting_iterator<int32_t> positions(0);
  const int64_t span = key_upper_bound > 0 ? key_upper_bound - key_lower_bound : 0;
  // Adaptive route. A batch that overflows a bucket of the split sort (hot ids of a skewed batch) is sorted by the gated generic
  // path instead — correct, but 0.11-0.13 ms slower per 10 M ids than rocPRIM's sort on the caller's stream would have been
  // (its passes take 61-69 us on such ids, the hand-written ones 98-121: profiles/r05_grad_timeline_zipf_*.txt), and skewed
  // batches come in series (the same power-law rows every step). So while the last split sort of this row range overflowed, the
  // batch goes straight to rocPRIM (below), and every kProbeEvery-th such call runs the split sort's first two kernels as a
  // probe in front (30 us / 4); the first batch that would not overflow switches back. A wrong guess costs time, once.
  // WM_DEDUP_ADAPT=0: always the split sort. Not while a stream is captured (a graph replays ONE route).
  // Round 6: three routes per row range. 0 = the plain split sort; when it overflowed, 1 = the split sort with the batch's hot ids
  // peeled into buckets of their own (split::launch_hot: a Zipf batch then fits; taken only with WM_DEDUP_HOT=1, see below); when THAT
  // overflowed too (ids clustered in a few thousand rows), 2 = rocPRIM's sort, probed every kProbeEvery-th call with the first
  // kernels of the hot-mode sort. Back: a hot-mode sort that peeled nothing returns the range to route 0.
  bool expect_overflow = false;
  bool hot_route       = false;
  int adapt_slot       = -1;
  // (OPT-IN, WM_DEDUP_HOT=1: correct on every case of the harness and the parity tests, but on the Zipf(1.05) batch of 10 M ids the
  // hot-mode sort takes 0.54-0.70 ms where rocPRIM's takes 0.36 — putting the ~90 k listed segments into receive order costs more
  // than the radix passes it avoids: profiles/r06_split_sort_hot_harness.txt. Route 2 follows a plain overflow by default.)
  const bool hot_allowed = WM_KNOB("WM_DEDUP_HOT") != nullptr && (WM_KNOB("WM_DEDUP_HOT")[0] == '1' || WM_KNOB("WM_DEDUP_HOT")[0] == '2');
  const bool hot_forced  = WM_KNOB("WM_DEDUP_HOT") != nullptr && WM_KNOB("WM_DEDUP_HOT")[0] == '2';   // (tests: every batch takes route 1)
  if (hot_forced) {
    hot_route = span > 0 && span < INT64_C(0xFFFFFFFF) && n >= split_min() && split::make_plan(n, span, 0, 0, true).ok;
  } else if (span > 0 && span < INT64_C(0xFFFFFFFF) && n >= split_min() && WM_KNOB("WM_DEDUP_SERIAL") == nullptr &&
      !(WM_KNOB("WM_DEDUP_ADAPT") != nullptr && WM_KNOB("WM_DEDUP_ADAPT")[0] == '0')) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone && sort_lane::get().ok) {
      sort_lane& lane = sort_lane::get();
      std::lock_guard<std::mutex> lk(lane.mu);
      adapt_slot = lane.adapt_slot(key_lower_bound, key_upper_bound);
      if (adapt_slot >= 0) {
        auto& e = lane.adapt[adapt_slot];
        volatile int32_t* f_plain = lane.adapt_flags + adapt_slot;
        volatile int32_t* f_hot   = lane.adapt_flags + sort_lane::kAdapt + adapt_slot;
        volatile int32_t* f_nhot  = lane.adapt_flags + 2 * sort_lane::kAdapt + adapt_slot;
        const split::plan hp = hot_allowed ? split::make_plan(n, span, 0, 0, true) : split::plan{};
        const bool can_hot   = hot_allowed && hp.ok;
        if (e.mode == 0 && *f_plain != 0) {
          e.mode = can_hot ? 1 : 2, e.calls = 0;
          *f_hot = can_hot ? 0 : 1, *f_nhot = 1;
        } else if (e.mode == 1 && (!can_hot || *f_hot != 0)) {
          e.mode = 2, e.calls = 0;
        } else if (e.mode == 1 && *f_nhot == 0) {
          e.mode = 0, *f_plain = 0;   // (nothing was peeled: the batches are no longer skewed)
        }
        if (e.mode == 2) {
          const split::plan sp = can_hot ? hp : split::make_plan(n, span);
          if (!sp.ok) {
            e.mode = 0, *f_plain = 0;   // (a batch the split sort would not take anyway)
          } else {
            volatile int32_t* verdict = can_hot ? f_hot : f_plain;
            if (*verdict == 0) {
              e.mode = can_hot ? 1 : 0;   // the last probe found room: back to the split sort
              if (e.mode == 1) *f_nhot = 1;
            } else if (++e.calls % sort_lane::kProbeEvery == 0) {
              if (lane.probe_ws == nullptr && hipMalloc(&lane.probe_ws, split::probe_workspace_bytes()) != hipSuccess) lane.probe_ws = nullptr;
              if (lane.probe_ws != nullptr) {
                if (split::launch_probe<UKey>(sp, static_cast<const UKey*>(ids), n, static_cast<UKey>(key_lower_bound),
                                              static_cast<uint32_t>(span), lane.probe_ws, stream) != 0)
                  return -2;
                (void)hipMemcpyAsync(const_cast<int32_t*>(verdict), split::probe_overflow_word(sp, lane.probe_ws), sizeof(int32_t),
                                     hipMemcpyDeviceToHost, stream);
              }
            }
          }
        }
        expect_overflow = e.mode == 2;
        hot_route       = e.mode == 1;
      }
    }
  }
  if (!expect_overflow && span > 0 && span < INT64_C(0xFFFFFFFF) && n >= split_min()) {
    const split::plan sp = split::make_plan(n, span, 0, 0, hot_route);
    if (sp.ok) {
      const split_layout sl = split_carve(workspace, n);
      const unsigned bits   = significant_bits(span + 1, 32);
      const size_t ctrl     = osw::ctrl_words<kOswBlock, kOswIpt>(n, bits);
      const int64_t zero_n  = static_cast<int64_t>(ctrl);
      // the generic path, gated on the overflow word, between the split sort's second and third kernel — on the side stream
      // (WM_DEDUP_SERIAL=1: on the caller's stream, for measurements)
      const uint32_t* gate = split::overflow_word(sp, sl.split_ws);
      char* sw             = static_cast<char*>(sl.split_ws);
      uint32_t* ctl        = reinterpret_cast<uint32_t*>(sw + sp.off_ctl);
      const split::wait_cfg wc = wait_limits(&sp);
      narrow_key_iterator<UKey> keys{static_cast<const UKey*>(ids), static_cast<UKey>(key_lower_bound), static_cast<uint32_t>(span)};
      const bool serial = WM_KNOB("WM_DEDUP_SERIAL") != nullptr;
      std::unique_lock<std::mutex> lane_lock;
      if (!serial) lane_lock = std::unique_lock<std::mutex>(sort_lane::get().mu);
      int generic_rc = 0;
      auto generic = [&](hipStream_t gs) {
        generic_rc = osw::sort_pairs<kOswBlock, kOswIpt>(keys, sl.sorted, reinterpret_cast<uint32_t*>(order), n, bits,
                                                         reinterpret_cast<uint32_t*>(sw + sp.off_keys),
                                                         reinterpret_cast<uint32_t*>(sw + sp.off_pos), sl.osw_ctrl, gate, gs,
                                                         WM_KNOB("WM_DEBUG_SPIN_LIMIT") != nullptr ? wc.look_back_polls : 1u << 26);
        if (generic_rc == 0)
          generic_rc = detect_runs<uint32_t, UKey>(sl.sorted, sl.tile_counts, n, static_cast<UKey*>(unique_ids), run_starts,
                                                   n_unique_out, gs, static_cast<UKey>(key_lower_bound), true,
                                                   static_cast<uint32_t>(span), gate,
                                                   stall_join() ? nullptr : ctl + split::kCtlGenericDone,
                                                   osw::error_word<kOswBlock, kOswIpt>(sl.osw_ctrl, n, bits), ctl + split::kCtlError);
      };
      bool forked = false;
      uint32_t* verdict_word = nullptr;
      uint32_t verdict_value = 0;
      // (a stream that is being captured gets events only: a wave that waits for a word needs the other side to be RUNNING,
      // and the branches of a graph may be replayed one after the other)
      hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
      const bool capturing = hipStreamIsCapturing(stream, &capture) != hipSuccess || capture != hipStreamCaptureStatusNone;
      const bool by_event = WM_KNOB("WM_DEDUP_FORK_EVENT") != nullptr && WM_KNOB("WM_DEDUP_FORK_EVENT")[0] == '1';   // (A/B switch)
      const bool waits_ok = !capturing && device_waits_allowed();
      if (!serial && waits_ok && !by_event && sort_lane::get().ok && sort_lane::get().ring != nullptr) {
        sort_lane& lane = sort_lane::get();
        verdict_value   = ++lane.seq;
        verdict_word    = lane.ring + (verdict_value % sort_lane::kRing);
      }
      auto between = [&]() {
        sort_lane& lane = sort_lane::get();
        if (verdict_word != nullptr) {
          hipLaunchKernelGGL(split::split_wait_kernel, dim3(1), dim3(64), 0, lane.side(), verdict_word, verdict_value,
                             ctl + split::kCtlError, wc.wait_polls, lane.host_err_dev);
          forked = hipGetLastError() == hipSuccess;
          if (!forked) {   // (the caller's kernels and the join kernel are queued already: nothing of the generic path behind them)
            generic_rc = -2;
            return;
          }
        } else {
          forked = !serial && lane.ok && hipEventRecord(lane.forked, stream) == hipSuccess &&
                   hipStreamWaitEvent(lane.side(), lane.forked, 0) == hipSuccess;
        }
        generic(forked ? lane.side() : stream);
        if (forked && adapt_slot >= 0) {   // (what this batch did decides the route of the next: see "adaptive route")
          (void)hipMemcpyAsync(const_cast<int32_t*>(lane.adapt_flags) + (hot_route ? sort_lane::kAdapt : 0) + adapt_slot, gate,
                               sizeof(int32_t), hipMemcpyDeviceToHost, lane.side());
          if (hot_route)   // ... and how many ids it peeled (none: the range goes back to the plain split sort)
            (void)hipMemcpyAsync(const_cast<int32_t*>(lane.adapt_flags) + 2 * sort_lane::kAdapt + adapt_slot,
                                 split::hot_view(sp, sl.split_ws).n_hot, sizeof(int32_t), hipMemcpyDeviceToHost, lane.side());
        }
        if (forked) forked = hipEventRecord(lane.joined, lane.side()) == hipSuccess;
      };
      // With the fork by a word nothing ties the side stream's launches to a place in the caller's queue: the caller's kernels
      // — the split sort's four — are enqueued FIRST (a mini-batch is bound by the host's launch rate — the nine side launches in the middle delayed the
      // scatter kernel by as many launch times), then the side stream, then the join kernel.
      const bool side_last = verdict_word != nullptr && !(WM_AB_KNOB("WM_SIDE_FIRST") != nullptr && WM_AB_KNOB("WM_SIDE_FIRST")[0] == '1');
      auto nothing         = []() {};
      const bool after_scatter = WM_AB_KNOB("WM_DEDUP_FORK") != nullptr && WM_AB_KNOB("WM_DEDUP_FORK")[0] == '3';
      const int launched =
        hot_route ? (side_last ? split::launch_hot<UKey>(sp, static_cast<const UKey*>(ids), n, static_cast<UKey>(key_lower_bound),
                                                         static_cast<uint32_t>(span), unique_ids, run_starts, order, n_unique_out,
                                                         sl.split_ws, sl.osw_ctrl, zero_n, stream, nothing, verdict_word, verdict_value, wc)
                               : split::launch_hot<UKey>(sp, static_cast<const UKey*>(ids), n, static_cast<UKey>(key_lower_bound),
                                                         static_cast<uint32_t>(span), unique_ids, run_starts, order, n_unique_out,
                                                         sl.split_ws, sl.osw_ctrl, zero_n, stream, between, verdict_word, verdict_value, wc)) :
        side_last ? split::launch<UKey>(sp, static_cast<const UKey*>(ids), n, static_cast<UKey>(key_lower_bound),
                                        static_cast<uint32_t>(span), unique_ids, run_starts, order, n_unique_out, sl.split_ws,
                                        sl.osw_ctrl, zero_n, stream, nothing, false, verdict_word, verdict_value, wc)
                  : split::launch<UKey>(sp, static_cast<const UKey*>(ids), n, static_cast<UKey>(key_lower_bound),
                                        static_cast<uint32_t>(span), unique_ids, run_starts, order, n_unique_out, sl.split_ws,
                                        sl.osw_ctrl, zero_n, stream, between, after_scatter, verdict_word, verdict_value, wc);
      if (launched != 0) return -2;
      // (every wave that waits is enqueued BEHIND the kernel it waits for — the side stream's behind the split sort's kernels,
      // the join kernel behind the side stream's last — so that even one in-order hardware queue makes progress)
      if (side_last) between();
      const bool defer = g_defer_join && waits_ok && forked;
      if (!defer && forked && hipStreamWaitEvent(stream, sort_lane::get().joined, 0) != hipSuccess) return -2;
      // the sort's LAST kernel on the caller's stream, whichever way the side stream is joined: waits for the generic path when
      // the join is deferred and the batch overflowed (1 = what detect_runs' closing kernel sets), and turns any wait of this
      // sort that gave up into "no runs" + an error word the host will see (split_sort.cuh: split_join_kernel)
      hipLaunchKernelGGL(split::split_join_kernel, dim3(1), dim3(64), 0, stream, ctl, 1u, wc.join_polls, n_unique_out,
                         sort_lane::get().host_err_dev);
      if (defer) g_join_pending = true;
      g_split_sorts.fetch_add(1, std::memory_order_relaxed);
      if (hot_route) g_hot_split_sorts.fetch_add(1, std::memory_order_relaxed);
      g_last_split.run_starts = run_starts;
      g_last_split.unique_ids = unique_ids;
      g_last_split.n_unique   = n_unique_out;
      g_last_split.ctl        = ctl;
      return generic_rc;
    }
  }
  if (span > 0 && span < INT64_C(0xFFFFFFFF)) {
    // a bounded range of fewer than 2^32 - 1 rows: 32-bit keys relative to its start, the value `span` marks ids outside it
    const unsigned bits = significant_bits(span + 1, 32);
    auto l    = layout<uint32_t>(workspace, n);
    size_t tb = l.temp_bytes;
    narrow_key_iterator<UKey> keys{static_cast<const UKey*>(ids), static_cast<UKey>(key_lower_bound), static_cast<uint32_t>(span)};
    if (sort_pairs32(l.temp, tb, keys, l.sorted, positions, order, static_cast<size_t>(n), 0, bits, stream) != hipSuccess) return -2;
    return detect_runs<uint32_t, UKey>(l.sorted, l.tile_counts, n, static_cast<UKey*>(unique_ids), run_starts, n_unique_out,
                                       stream, static_cast<UKey>(key_lower_bound), true, static_cast<uint32_t>(span));
  }
  const unsigned bits = significant_bits(key_upper_bound > 0 ? key_upper_bound : 0, 8 * sizeof(KeyT));
  auto l    = layout<UKey>(workspace, n);
  size_t tb = l.temp_bytes;
  if (rocprim::radix_sort_pairs<typename sort_config<UKey>::type>(l.temp, tb, static_cast<const UKey*>(ids), l.sorted, positions,
                                                                  order, static_cast<size_t>(n), 0, bits, stream) != hipSuccess)
    return -2;
  return detect_runs<UKey, UKey>(l.sorted, l.tile_counts, n, static_cast<UKey*>(unique_ids), run_starts, n_unique_out, stream);
}

// ---------------------------------------------------------------------------------------------
// fused duplicate-sum + optimizer step
// ---------------------------------------------------------------------------------------------
// Two kernels share the work by run length (a "run" = all received rows of one unique id, in receive order):
//   step_short_kernel : one wave per run, for runs of <= kLongRun rows (the common case). Lanes own columns;
//                       the duplicates are folded sequentially (first copied, the rest added one by one — the
//                       reference order), up to 4 duplicate rows prefetched at a time. Longer runs are skipped here:
//                       mark_long_runs_kernel lists them (with their LazyAdam beta powers) for the long-run kernel,
//                       which runs on a second stream next to this one.
//   step_long_kernel  : one workgroup per (long run, 32-column slice). 256 threads stream the run's rows through
//                       double-buffered LDS tiles (256 rows x 128 B, the next tile already in registers while the
//                       current one is folded), 32 lanes fold their column sequentially out of LDS. Summation
//                       order per element is still exactly the receive order, so results stay bit-identical; a
//                       500 k-duplicate hot row (Zipf s = 1.05) costs milliseconds instead of ~0.5 s of dependent
//                       global loads.
// Runs of up to kLongRun rows are folded by the wave that owns them in step_short_kernel (4 rows prefetched per
// dependent round: ~60 us for 256 rows, and a Zipf batch has only one or two such runs per wave). The long-run kernel
// pays ~6 us per run before its first row is folded and keeps 255 VGPRs x 6 waves busy on a CU while it walks its list —
// with the threshold at 32 it was given 9600 runs of the Zipf(1.05) batch, held every CU for over a millisecond and
// halved the speed of step_short_kernel beside it; at 256 it gets 1300 (whole call, SGD / LazyAdam / fp16 x 256:
// 32 -> 4.45 / 6.7 / 5.2 ms, 256 -> 3.85 / 6.1 / 4.2 ms, 1024 -> 3.85 / 6.1 / 4.3 ms, 4096 -> 4.2 / 6.2 / 4.2 ms).
constexpr int kLongRun   = 256;
constexpr int kDenseR    = 128;   // dense ordered fold (kernels/long_dense.cuh): rows a folding wave holds per turn (experiments/fold5_harness.hip)
constexpr int kDenseS    = 8;     // ... and columns per folding workgroup
constexpr int kSliceCols = 32;    // 128 B of every row per long-run workgroup
constexpr int kTileRows  = 256;   // rows per LDS tile (32 KiB), double buffered

struct long_run_entry {
  int32_t run;
  float beta1t, beta2t;
  int32_t dense;   // 1: folded through its dense copy (dense_fold), step_long4_kernel skips it
};

struct opt_params {
  wm_optimizer_args a;
  const int64_t* n_unique;  // device scalar (or nullptr -> a.count)
  long_run_entry* long_list;
  int32_t* long_count;
  const uint32_t* split_ctl;   // control words of the split sort that produced the runs (or nullptr): see last_split_record
  // step_tile_kernel: runs per wave tile. 64 with the persistent grid of round 2; one batch (RPS x kU runs) when the tiles
  // are handed out in order, one per wave (round 3, see launch_step_opt)
  int tile_runs;
  // runs of more rows than this are not folded by step_tile_kernel / step_short_kernel but listed for the long-run side
  // (kLongRun with the ordered fold, tree_threshold() with the tree fold)
  int long_threshold;
  int detached_side;   // 1 / 2: the long-run side runs on a side stream the caller's stream does NOT wait for (hip_optimizer_step_dev)
  int fold_tree;   // 1: the long-run side is the tree fold (tree_fold_kernel), 0: the ordered fold (step_long4_kernel)
  // ordered fold of the very long runs through a dense transposed copy (kernels/long_dense.cuh; round 6): listed runs of at least
  // dense_min rows get a job and room in dense_buf while it lasts (dense_cap floats), the others stay step_long4_kernel's.
  // Counters beside long_count[0]: [1] jobs, [2..3] the buffer cursor (64 bit)
  dense_fold::job* dense_jobs;
  float* dense_buf;
  int64_t dense_cap;
  int dense_min, dense_max_jobs;
  int dense_x;   // step_long4_kernel: its first dense_x workgroups (per slice) fold the runs with a dense copy
};

// optimizer statement sequences of the reference kernels (embedding_optimizer_func.cu:212-223, 392-415,
// 644-657, 842-855), one element. Loading and updating are separate so callers can put the loads of several
// independent rows in flight before the first dependent arithmetic.
struct opt_elem {
  float e, s0, s1;  // table value, first / second per-element state (m, v | state_sum | v)
};

// gradient row addressed by an order[] entry (wm_optimizer_args::self_grads)
template <typename T = float>
__device__ __forceinline__ const T* grad_row(const wm_optimizer_args& a, int32_t o)
{
  return o >= 0 ? static_cast<const T*>(a.grads) + static_cast<int64_t>(o) * a.grad_stride
                : static_cast<const T*>(a.self_grads) + (-(static_cast<int64_t>(o) + 1)) * a.self_grad_stride;
}

// V consecutive elements of a gradient row as fp32 (16-bit types are widened exactly)
template <typename T, int V>
__device__ __forceinline__ auto load_vec(const T* p)
{
  typedef float fvec __attribute__((ext_vector_type(V)));
  if constexpr (std::is_same<T, float>::value) {
    return *reinterpret_cast<const fvec*>(p);
  } else {
    typedef uint16_t rvec __attribute__((ext_vector_type(V)));
    const rvec raw = *reinterpret_cast<const rvec*>(p);
    fvec out;
#pragma unroll
    for (int i = 0; i < V; i++) {
      const uint16_t bits = raw[i];
      out[i]              = load_wide<T>(__builtin_bit_cast(T, bits));
    }
    return out;
  }
}
template <typename T>
__device__ __forceinline__ float load_vec1(const T* p)
{
  return load_wide<T>(*p);
}

// where row `local` of the shard lives: its cache line when it is resident in the device row cache, else the table
template <typename T = float>
__device__ __forceinline__ T* table_row_at(const wm_optimizer_args& a, int64_t local, int32_t slot)
{
  return slot >= 0 ? static_cast<T*>(a.cache_data) + static_cast<int64_t>(slot) * a.cache_row_elems
                   : static_cast<T*>(a.local_table) + local * a.table_stride;
}
__device__ __forceinline__ int32_t cache_slot(const wm_optimizer_args& a, int64_t local)
{
  return a.cache_slot_of != nullptr ? a.cache_slot_of[local] : -1;
}

// per-element optimizer state of row `local`: the companion cache line of its slot when the row is resident in the
// device row cache (and the states are cached with it), else the state table
__device__ __forceinline__ float* state_row_at(const wm_optimizer_args& a, int64_t local, int32_t slot)
{
  if (a.per_element_state == nullptr) return nullptr;
  return slot >= 0 && a.cache_state_data != nullptr ? a.cache_state_data + static_cast<int64_t>(slot) * a.cache_state_row_elems
                                                    : a.per_element_state + local * a.per_element_stride;
}

template <int OPT, typename T = float>
__device__ __forceinline__ opt_elem load_elem(const wm_optimizer_args& a, const T* row, const float* st, int64_t d)
{
  opt_elem x;
  x.e  = load_wide<T>(row[d]);
  x.s0 = 0.f;
  x.s1 = 0.f;
  if (OPT != WHOLEMEMORY_OPT_SGD) {
    x.s0            = st[d];
    if (OPT == WHOLEMEMORY_OPT_LAZY_ADAM) x.s1 = st[a.table_stride + d];
  }
  return x;
}

// the statement sequences themselves, on values: x.e / x.s0 / x.s1 in, updated in place (compiled with -ffp-contract=off)
template <int OPT>
__device__ __forceinline__ void opt_math(const wm_optimizer_args& a, opt_elem& x, float grad_value, float beta1t, float beta2t)
{
  float embedding_value = x.e;
  if (OPT == WHOLEMEMORY_OPT_SGD) {
    grad_value += a.weight_decay * embedding_value;
    embedding_value -= a.lr * grad_value;
  } else if (OPT == WHOLEMEMORY_OPT_LAZY_ADAM) {
    if (a.adam_w) {
      embedding_value -= a.lr * a.weight_decay * embedding_value;
    } else {
      grad_value = grad_value + a.weight_decay * embedding_value;
    }
    float m         = x.s0;
    float v         = x.s1;
    m               = a.beta1 * m + (1 - a.beta1) * grad_value;
    v               = a.beta2 * v + (1 - a.beta2) * grad_value * grad_value;
    float mhat      = m / (1 - beta1t);
    float vhat      = v / (1 - beta2t);
    embedding_value = embedding_value - a.lr * mhat / (sqrtf(vhat) + a.epsilon);
    x.s0            = m;
    x.s1            = v;
  } else if (OPT == WHOLEMEMORY_OPT_ADAGRAD) {
    grad_value      = grad_value + a.weight_decay * embedding_value;
    float state_sum = x.s0;
    state_sum       = state_sum + grad_value * grad_value;
    embedding_value = embedding_value - a.lr * grad_value / (sqrtf(state_sum) + a.epsilon);
    x.s0            = state_sum;
  } else if (OPT == WHOLEMEMORY_OPT_RMSPROP) {
    grad_value      = grad_value + a.weight_decay * embedding_value;
    float v         = x.s0;
    v               = a.alpha * v + (1 - a.alpha) * grad_value * grad_value;
    embedding_value = embedding_value - a.lr * grad_value / (sqrtf(v) + a.epsilon);
    x.s0            = v;
  }
  x.e = embedding_value;
}

template <int OPT, typename T = float>
__device__ __forceinline__ void update_elem(const wm_optimizer_args& a, T* row, float* st, int64_t d, opt_elem x,
                                            float grad_value, float beta1t, float beta2t)
{
  opt_math<OPT>(a, x, grad_value, beta1t, beta2t);
  if (OPT != WHOLEMEMORY_OPT_SGD) st[d] = x.s0;
  if (OPT == WHOLEMEMORY_OPT_LAZY_ADAM) st[a.table_stride + d] = x.s1;
  // the updated row is not read again in this pass: non-temporal store (merged into one wide store per lane)
  if constexpr (std::is_same<T, float>::value)
    __builtin_nontemporal_store(x.e, &row[d]);
  else  // 16-bit tables (SGD extension): one rounding from the fp32 result
    row[d] = store_narrow<T>(x.e);
}

template <int OPT, typename T = float>
__device__ __forceinline__ void apply_optimizer(const wm_optimizer_args& a, int64_t local, int64_t d, float grad_value,
                                                float beta1t, float beta2t)
{
  const int32_t slot = cache_slot(a, local);
  T* row             = table_row_at<T>(a, local, slot);
  float* st          = state_row_at(a, local, slot);
  if (slot >= 0) a.cache_dirty[slot] = 1;
  update_elem<OPT, T>(a, row, st, d, load_elem<OPT, T>(a, row, st, d), grad_value, beta1t, beta2t);
}

// The work per run is a chain of dependent loads (ids / run_starts -> order -> gradient row; ids -> table row), and the
// memory system is NOT saturated by this kernel (rocprofv3: TCC_EA0_WRREQ_STALL and TCC_TAG_STALL ~ 0, against heavy
// stalls in the gather kernel) — it is bound by how many row loads are in flight. So the loop is software-pipelined in
// three stages that all issue at the top of an iteration without waiting on each other:
//   stage 1  ids / run_starts of iteration i + 2
//   stage 2  order[run start] (+ LazyAdam beta powers) of iteration i + 1   (its stage 1 was issued an iteration ago)
//   stage 3  gradient rows + table rows of iteration i                       (its stage 2 was issued an iteration ago)
// with WM_STEP_K runs per stage. Measured on 10 M gradient rows (9.5 M unique, 512 B rows, SGD): 3.0-3.1 ms = 15 GB of
// HBM traffic (PMC: 1.02 x algorithmic) at ~5 TB/s. Runs in flight (K = 2 / 4 / 8), vector width (8 / 16 B per lane),
// occupancy (5 / 6 / 8 waves per SIMD) and non-temporal table stores all land within 5 % of that figure, and removing
// any one of the three row streams (gradient read, table read, table write) removes its share of the time: the kernel
// sits on what the memory system delivers for two random row reads + one row write-back per unique id.
#ifndef WM_STEP_K
#define WM_STEP_K 4
#endif
// CACHED: the table has a device row cache (wm_optimizer_args::cache_slot_of) and every row is addressed through its
// slot map. Compiled out otherwise: with the per-row select between two bases in the way, hipcc keeps the row addresses
// in VGPR pairs instead of scalar base + lane offset, and the plain kernel lost a quarter of its speed (3.1 -> 3.9 ms).
template <typename IdxT, int OPT, int V, typename T = float, bool CACHED = true>
__global__ __launch_bounds__(kBlock) void step_short_kernel(opt_params p)
{
  typedef float vec_t __attribute__((ext_vector_type(V)));
  constexpr int K            = WM_STEP_K;
  const wm_optimizer_args& a = p.a;
  const int lane             = threadIdx.x & 63;
  // the wave index as a SCALAR: everything derived from it (run numbers, ids, run_starts, order entries, row bases) is
  // then wave-uniform for the compiler too — scalar loads and SGPRs instead of 64 identical vector loads and VGPRs
  const int64_t wave   = __builtin_amdgcn_readfirstlane(static_cast<int>((blockIdx.x * kBlock + threadIdx.x) >> 6));
  const int64_t stride = ((static_cast<int64_t>(gridDim.x) * kBlock) >> 6) * K;
  const int64_t count  = p.n_unique ? *p.n_unique : a.count;
  const IdxT* ids      = static_cast<const IdxT*>(a.ids);

  struct stage1 {
    int64_t local[K];
    int32_t s0[K], s1[K];
  };
  struct stage2 {
    int32_t o0[K];
    int32_t slot[K];  // cache line of the row, or -1
    float beta1t[K], beta2t[K];
  };
  auto load1 = [&](int64_t u0) {
    stage1 m;
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int64_t u = min(u0 + k, count - 1);  // clamped: loads stay unconditional (the values of dead slots are unused)
      m.local[k]      = static_cast<int64_t>(ids[u]) - a.local_entry_offset;
      m.s0[k]         = a.run_starts[u];
      m.s1[k]         = a.run_starts[u + 1];
    }
    return m;
  };
  auto load2 = [&](const stage1& m) {
    stage2 r;
#pragma unroll
    for (int k = 0; k < K; k++) {
      r.o0[k]     = a.order[m.s0[k]];
      r.slot[k]   = CACHED ? cache_slot(a, m.local[k]) : -1;
      r.beta1t[k] = r.beta2t[k] = 0.f;
      if (OPT == WHOLEMEMORY_OPT_LAZY_ADAM) {
        r.beta1t[k] = a.per_row_state[m.local[k] * 2 + 0] * a.beta1;
        r.beta2t[k] = a.per_row_state[m.local[k] * 2 + 1] * a.beta2;
      }
    }
    return r;
  };

  int64_t u0 = wave * K;
  if (u0 >= count) return;
  stage1 m_cur = load1(u0);
  stage1 m_nxt = load1(u0 + stride);
  stage2 r_cur = load2(m_cur);
  for (; u0 < count; u0 += stride) {
    const stage1 m_nn  = load1(u0 + 2 * stride);
    const stage2 r_nxt = load2(m_nxt);
    bool live[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      live[k] = u0 + k < count;
      // runs of more than kLongRun rows belong to the long-run kernel (listed by mark_long_runs_kernel, which also
      // advances their LazyAdam beta powers): nothing of them is touched here
      if (p.long_list != nullptr && m_cur.s1[k] - m_cur.s0[k] > p.long_threshold) live[k] = false;
      if (OPT == WHOLEMEMORY_OPT_LAZY_ADAM && live[k] && lane == 0) {
        // every lane has read the old values (same wave, program order) before this store
        a.per_row_state[m_cur.local[k] * 2 + 0] = r_cur.beta1t[k];
        a.per_row_state[m_cur.local[k] * 2 + 1] = r_cur.beta2t[k];
      }
    }
    T* row[K];
    float* st[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      if (CACHED) {
        row[k] = table_row_at<T>(a, m_cur.local[k], r_cur.slot[k]);
        st[k]  = state_row_at(a, m_cur.local[k], r_cur.slot[k]);
        if (live[k] && r_cur.slot[k] >= 0 && lane == 0) a.cache_dirty[r_cur.slot[k]] = 1;
      } else {
        row[k] = static_cast<T*>(a.local_table) + m_cur.local[k] * a.table_stride;
        st[k]  = OPT != WHOLEMEMORY_OPT_SGD ? a.per_element_state + m_cur.local[k] * a.per_element_stride : nullptr;
      }
    }
    for (int64_t d = static_cast<int64_t>(lane) * V; d < a.dim; d += 64 * V) {
      vec_t acc[K];
      opt_elem x[K][V];
#pragma unroll
      for (int k = 0; k < K; k++) {
        // first occurrence copied (DedupIndiceAndGradientsKernel); table / state values loaded alongside
        acc[k] = load_vec<T, V>(grad_row<T>(a, r_cur.o0[k]) + d);
#pragma unroll
        for (int v = 0; v < V; v++) x[k][v] = load_elem<OPT, T>(a, row[k], st[k], d + v);
      }
#pragma unroll
      for (int k = 0; k < K; k++) {
        if (!live[k]) continue;
        // later occurrences added in receive order, 4 rows prefetched at a time (index clamped into the run)
        for (int32_t j = m_cur.s0[k] + 1; j < m_cur.s1[k]; j += 4) {
          int32_t o[4];
          vec_t g[4];
#pragma unroll
          for (int q = 0; q < 4; q++) o[q] = a.order[min(j + q, m_cur.s1[k] - 1)];
#pragma unroll
          for (int q = 0; q < 4; q++)
            g[q] = load_vec<T, V>(grad_row<T>(a, o[q]) + d);
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (j + q < m_cur.s1[k]) acc[k] += g[q];
        }
#pragma unroll
        for (int v = 0; v < V; v++)
          update_elem<OPT, T>(a, row[k], st[k], d + v, x[k][v], acc[k][v], r_cur.beta1t[k], r_cur.beta2t[k]);
      }
    }
    m_cur = m_nxt;
    m_nxt = m_nn;
    r_cur = r_nxt;
  }
}


// ---------------------------------------------------------------------------------------------
// step_tile_kernel: the gather kernel's shape applied to the fused step (fp32 rows of whole 16-byte pieces).
// A wave owns a TILE of 64 consecutive runs (= 64 ascending unique ids): ids / run_starts / order[run start] arrive as
// coalesced vector loads (lane l holds run 64 t + l), each lane resolves ITS run's gradient-row and table-row (and
// state-row) addresses once, and the tile is then streamed kU steps of RPS rows at a time with the bases broadcast by
// v_readlane — per step one 16-byte load per lane from the gradient row and one from the table row (two state rows
// for the stateful optimizers), kU steps in flight before the first dependent arithmetic, updated rows written back
// non-temporally. step_short_kernel walks the same data with scalar loads per run (ids -> run_starts -> order ->
// rows: three dependent latencies per K runs per wave); here the metadata of 64 runs costs one latency.
// Duplicates (run length > 1; 5 % of the runs of a uniform 10 M-in-100 M batch) are folded in the step that holds their
// first row: the later rows are added one by one in receive order, 4 prefetched at a time — the reference order, so
// results stay bit-identical. Runs of more than kLongRun rows belong to the long-run kernel as before.
template <int RPS>
__device__ __forceinline__ uint32_t tile_bcast32(uint32_t v, int e0, int sub)
{
  if (RPS == 1) return __builtin_amdgcn_readlane(v, e0);
  if (RPS == 2) {
    const uint32_t lo = __builtin_amdgcn_readlane(v, e0);
    const uint32_t hi = __builtin_amdgcn_readlane(v, e0 + 1);
    return sub ? hi : lo;
  }
  return __shfl(v, e0 + sub, 64);
}
// base pointer of run e0 + sub for this lane: RPS == 1: v_readlane (wave-uniform, SGPRs); else one ds_bpermute per half
// (2 LDS-crossbar instructions per pointer instead of the 4 v_readlane + 4 v_mov + 2 v_cndmask of tile_bcast_ptr<2>)
template <int RPS, typename P>
__device__ __forceinline__ P* tile_lane_ptr(P* p, int e0, int sub)
{
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  uint32_t lo, hi;
  if (RPS == 1) {
    lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(v), e0);
    hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(v >> 32), e0);
  } else {
    lo = __builtin_amdgcn_ds_bpermute(4 * (e0 + sub), static_cast<uint32_t>(v));
    hi = __builtin_amdgcn_ds_bpermute(4 * (e0 + sub), static_cast<uint32_t>(v >> 32));
  }
  return reinterpret_cast<P*>((static_cast<uint64_t>(hi) << 32) | lo);
}
template <int RPS, typename P>
__device__ __forceinline__ P* tile_bcast_ptr(P* p, int e0, int sub)
{
  const uint64_t v  = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = tile_bcast32<RPS>(static_cast<uint32_t>(v), e0, sub);
  const uint32_t hi = tile_bcast32<RPS>(static_cast<uint32_t>(v >> 32), e0, sub);
  return reinterpret_cast<P*>((static_cast<uint64_t>(hi) << 32) | lo);
}

#ifndef WM_TILE_KU_SGD
#define WM_TILE_KU_SGD 4
#endif
#ifndef WM_TILE_KU_STATE
#define WM_TILE_KU_STATE 2
#endif
#ifndef WM_TILE_DUP
#define WM_TILE_DUP 2
#endif
// 16 bytes per lane, KU = steps in flight (0: the default for the optimizer). Swept on the 10 M-row SGD / LazyAdam call
// (profiles/r02_tile_sweep.txt): 16 B x 4 steps (SGD) and 16 B x 2 (stateful) are the defaults; 8 B per lane (a 512-byte row
// per wave, every per-row quantity wave-uniform) is 5 % slower, 8 steps in flight no faster, grids of 2048 / 4096
// workgroups 5-8 % slower than 8192.
// T = element type of the table AND of the gradient rows: float, or half / bf16 (SGD only: duplicates are summed in fp32
// in receive order, the update is computed in fp32 from fp32(e) and rounded once — 8 elements per 16-byte piece).
// PIECE = bytes a lane moves per step: 16 (rows of whole 16-byte pieces), or — round 6, fp32 only — 8: rows of whole 8-byte
// pieces that are not whole 16-byte ones (602 floats = 2408 B, the Reddit feature width; any dim = 2 mod 4) used to miss this
// kernel altogether and took the wave-per-run kernel (global_load_dwordx2, one run at a time).
typedef uint32_t tile_raw4 __attribute__((ext_vector_type(4)));
typedef uint32_t tile_raw2 __attribute__((ext_vector_type(2)));
// 16 bytes at an address that is only 4-byte aligned (RAGGED: packed gradient rows of dim % 4 != 0 floats): still ONE
// global_load_dwordx4 (the hardware takes dword-aligned wide accesses; one that straddles a line costs a second request)
typedef tile_raw4 tile_raw4_u __attribute__((aligned(4)));
template <int PIECE>
struct tile_raw_of {
  typedef tile_raw4 type;
};
template <>
struct tile_raw_of<8> {
  typedef tile_raw2 type;
};
template <int N>
struct tile_vals {
  float v[N];
};
template <typename T>
__device__ __forceinline__ tile_vals<2> tile_unpack(tile_raw2 r)
{
  static_assert(std::is_same<T, float>::value, "8-byte pieces: fp32 rows only");
  tile_vals<2> out;
  out.v[0] = __builtin_bit_cast(float, static_cast<uint32_t>(r.x));
  out.v[1] = __builtin_bit_cast(float, static_cast<uint32_t>(r.y));
  return out;
}
template <typename T>
__device__ __forceinline__ tile_raw2 tile_pack(const tile_vals<2>& in)
{
  static_assert(std::is_same<T, float>::value, "8-byte pieces: fp32 rows only");
  tile_raw2 r;
  r.x = __builtin_bit_cast(uint32_t, in.v[0]);
  r.y = __builtin_bit_cast(uint32_t, in.v[1]);
  return r;
}
template <typename T>
__device__ __forceinline__ tile_vals<16 / sizeof(T)> tile_unpack(tile_raw4 r)
{
  tile_vals<16 / sizeof(T)> out;
  if constexpr (std::is_same<T, float>::value) {
    out.v[0] = __builtin_bit_cast(float, static_cast<uint32_t>(r.x));
    out.v[1] = __builtin_bit_cast(float, static_cast<uint32_t>(r.y));
    out.v[2] = __builtin_bit_cast(float, static_cast<uint32_t>(r.z));
    out.v[3] = __builtin_bit_cast(float, static_cast<uint32_t>(r.w));
  } else {
    T e[8];
    __builtin_memcpy(e, &r, 16);
#pragma unroll
    for (int i = 0; i < 8; i++) out.v[i] = load_wide<T>(e[i]);
  }
  return out;
}
template <typename T>
__device__ __forceinline__ tile_raw4 tile_pack(const tile_vals<16 / sizeof(T)>& in)
{
  tile_raw4 r;
  if constexpr (std::is_same<T, float>::value) {
    r.x = __builtin_bit_cast(uint32_t, in.v[0]);
    r.y = __builtin_bit_cast(uint32_t, in.v[1]);
    r.z = __builtin_bit_cast(uint32_t, in.v[2]);
    r.w = __builtin_bit_cast(uint32_t, in.v[3]);
  } else {
    T e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = store_narrow<T>(in.v[i]);
    __builtin_memcpy(&r, e, 16);
  }
  return r;
}

// RAGGED (round 6, fp32, 16-byte pieces): rows of dim % 4 != 0 floats — 513, 129, 127: the reference's own gradient-apply test
// dims, wholememory_embedding_gradient_apply_tests.cu:481-501 — used to take the wave-per-run kernel on 4-byte pieces (36-47 %
// of peak). Here the first dim / 4 pieces of every row go through the same straight-line batches (the packed gradient rows are
// only 4-byte aligned: tile_raw4_u; table and state rows sit on a 16-byte stride), and the last dim % 4 floats of the tile's
// 64 rows are a pass of their own, one LANE per run (4-byte accesses, duplicates added in receive order as everywhere).
template <typename IdxT, int OPT, int RPS, bool CACHED, typename T = float, int KU = 0, int OCC = 0, int PIECE = 16, bool RAGGED = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(OCC > 0 ? OCC : 1, OCC > 0 ? OCC : 8)))
void step_tile_kernel(opt_params p)
{
  static_assert(PIECE == 16 || (PIECE == 8 && sizeof(T) == 4), "pieces of 16 bytes, or of 8 bytes of fp32 rows");
  static_assert(!RAGGED || (PIECE == 16 && sizeof(T) == 4 && !CACHED), "ragged rows: fp32, 16-byte pieces, no row cache");
  typedef typename tile_raw_of<PIECE>::type raw_t;
  constexpr int kSV = PIECE / 4;   // fp32 state values per piece
  // optimizer states move non-temporally like the rows (round 4, interleaved in one process, 10 M uniform rows of 128 floats:
  // LazyAdam 6.69 -> 6.38 ms, Zipf 4.83 -> 4.56; AdaGrad 4.53 -> 4.49; RMSProp 4.47 -> 4.45)
  auto ld_state = [](const void* q) { return ld_global_nt<raw_t>(q); };
  auto ld_grad  = [](const void* q) -> raw_t {
    if constexpr (RAGGED) return ld_global_nt<tile_raw4_u>(q);
    else return ld_global_nt<raw_t>(q);
  };
  auto st_state = [](void* q, raw_t v) { st_global_nt<raw_t>(q, v); };
  constexpr int kVE          = PIECE / static_cast<int>(sizeof(T));  // elements per lane
  constexpr bool k16         = sizeof(T) == 2;
  constexpr int kU           = KU > 0 ? KU : ((OPT == WHOLEMEMORY_OPT_SGD && !k16) ? WM_TILE_KU_SGD : WM_TILE_KU_STATE);
  constexpr int kLpr         = 64 / RPS;
  constexpr bool kState      = OPT != WHOLEMEMORY_OPT_SGD;
  constexpr bool kAdam       = OPT == WHOLEMEMORY_OPT_LAZY_ADAM;
  constexpr int kDup         = WM_TILE_DUP;   // later occurrences of a duplicated id prefetched at a time
  static_assert(!k16 || OPT == WHOLEMEMORY_OPT_SGD, "16-bit tables are trained with SGD only");
  const wm_optimizer_args& a = p.a;
  const int lane             = threadIdx.x & 63;
  const int64_t wave         = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) >> 6;
  const int64_t n_waves      = (static_cast<int64_t>(gridDim.x) * kBlock) >> 6;
  const int64_t count        = p.n_unique ? *p.n_unique : a.count;
  const int tile_runs        = p.tile_runs;   // a multiple of RPS * kU, <= 64
  const int64_t tiles        = (count + tile_runs - 1) / tile_runs;
  const IdxT* ids            = static_cast<const IdxT*>(a.ids);
  const int col              = lane & (kLpr - 1);
  const int sub              = lane / kLpr;
  const int row_vecs         = static_cast<int>(a.dim / kVE);

  for (int64_t tile = wave; tile < tiles; tile += n_waves) {
    const int64_t u     = tile * tile_runs + min(lane, tile_runs - 1);
    const int64_t uc    = min(u, count - 1);  // clamped: the loads stay unconditional, dead lanes get run length 0
    const int64_t local = static_cast<int64_t>(ids[uc]) - a.local_entry_offset;
    const int32_t my_s0 = a.run_starts[uc];
    int32_t my_len      = (u < count && lane < tile_runs) ? a.run_starts[uc + 1] - my_s0 : 0;
    if (p.long_list != nullptr && my_len > p.long_threshold) my_len = 0;  // the long-run side's (mark_long_runs_kernel / tree_mark_kernel lists it)
    const T* my_grad = grad_row<T>(a, a.order[my_s0]);
    T* my_row;
    float* my_st = nullptr;
    if (CACHED) {
      const int32_t slot = cache_slot(a, local);
      my_row             = table_row_at<T>(a, local, slot);
      my_st              = state_row_at(a, local, slot);
      if (my_len > 0 && slot >= 0) a.cache_dirty[slot] = 1;
    } else {
      my_row = static_cast<T*>(a.local_table) + local * a.table_stride;
      if (kState) my_st = a.per_element_state + local * a.per_element_stride;
    }
    float my_b1 = 0.f, my_b2 = 0.f;
    if (kAdam) {
      my_b1 = a.per_row_state[local * 2 + 0] * a.beta1;
      my_b2 = a.per_row_state[local * 2 + 1] * a.beta2;
      if (my_len > 0) {  // ids of a tile are distinct; dead lanes (same address as the last live one) do not write
        a.per_row_state[local * 2 + 0] = my_b1;
        a.per_row_state[local * 2 + 1] = my_b2;
      }
    }
    // Every run of the tile is this kernel's (none left to the long-run side, none past the end): the FAST PATH — a batch
    // of kU steps is straight-line code, 2 kU unconditional row loads (gradient + table) issued back to back, then the
    // arithmetic, then the stores, nothing predicated. Lanes past the end of a row repeat its last piece: same loads, same
    // result, same store as the lane that owns it. (Round 3 loaded under `if (ln[k] > 0)`: a conditionally defined
    // register per load, and hipcc waited for each load before issuing the next.) Duplicates (run length > 1) are folded
    // behind a wave-uniform test per step.
    const bool whole = __ballot(lane < tile_runs && my_len <= 0) == 0;
    for (int cbase = 0; cbase < row_vecs && whole; cbase += kLpr) {
      const int64_t coff = static_cast<int64_t>(min(cbase + col, row_vecs - 1)) * kVE;
#pragma unroll 1
      for (int s = 0; s < tile_runs; s += RPS * kU) {
        raw_t gv[kU], ev[kU], s0v[kU], s1v[kU];
        T* trow[kU];
        float* srow[kU];
#pragma unroll
        for (int k = 0; k < kU; k++) {
          const int e0 = s + RPS * k;
          const T* g   = tile_lane_ptr<RPS>(my_grad, e0, sub);
          trow[k]      = tile_lane_ptr<RPS>(my_row, e0, sub);
          if (kState) srow[k] = tile_lane_ptr<RPS>(my_st, e0, sub);
          gv[k]        = ld_grad(g + coff);
          ev[k]        = ld_global_nt<raw_t>(trow[k] + coff);
          if (kState) s0v[k] = ld_state(srow[k] + coff);
          if (kAdam) s1v[k] = ld_state(srow[k] + a.table_stride + coff);
        }
#pragma unroll
        for (int k = 0; k < kU; k++) {
          tile_vals<kVE> acc = tile_unpack<T>(gv[k]);   // first occurrence copied (DedupIndiceAndGradientsKernel)
          // longest run of this step's RPS rows, wave-uniform (the rows of a step differ only by `sub`)
          int32_t longest = 0;
#pragma unroll
          for (int r = 0; r < RPS; r++) longest = max(longest, static_cast<int32_t>(__builtin_amdgcn_readlane(my_len, s + RPS * k + r)));
          if (longest > 1) {
            // later occurrences added in receive order, 4 rows prefetched at a time (index clamped into the run)
            const int32_t ln = static_cast<int32_t>(tile_bcast32<RPS>(static_cast<uint32_t>(my_len), s + RPS * k, sub));
            const int32_t rs = static_cast<int32_t>(tile_bcast32<RPS>(static_cast<uint32_t>(my_s0), s + RPS * k, sub));
            for (int32_t j = 1; j < longest; j += kDup) {
              raw_t gq[kDup];
#pragma unroll
              for (int q = 0; q < kDup; q++) {
                const int32_t o = a.order[rs + min(j + q, ln - 1)];
                gq[q]           = ld_grad(grad_row<T>(a, o) + coff);
              }
#pragma unroll
              for (int q = 0; q < kDup; q++) {
                if (j + q < ln) {
                  const tile_vals<kVE> gx = tile_unpack<T>(gq[q]);
#pragma unroll
                  for (int v = 0; v < kVE; v++) acc.v[v] += gx.v[v];
                }
              }
            }
          }
          const tile_vals<kVE> e_in = tile_unpack<T>(ev[k]);
          tile_vals<kSV> s0_in{}, s1_in{}, so0{}, so1{};
          tile_vals<kVE> eo;
          if (kState) s0_in = tile_unpack<float>(s0v[k]);
          if (kAdam) s1_in = tile_unpack<float>(s1v[k]);
          float b1 = 0.f, b2 = 0.f;
          if (kAdam) {
            b1 = __builtin_bit_cast(float, tile_bcast32<RPS>(__builtin_bit_cast(uint32_t, my_b1), s + RPS * k, sub));
            b2 = __builtin_bit_cast(float, tile_bcast32<RPS>(__builtin_bit_cast(uint32_t, my_b2), s + RPS * k, sub));
          }
#pragma unroll
          for (int v = 0; v < kVE; v++) {
            opt_elem x;
            x.e  = e_in.v[v];
            x.s0 = kState ? s0_in.v[v & (kSV - 1)] : 0.f;
            x.s1 = kAdam ? s1_in.v[v & (kSV - 1)] : 0.f;
            opt_math<OPT>(a, x, acc.v[v], b1, b2);
            eo.v[v] = x.e;
            if (kState) so0.v[v & (kSV - 1)] = x.s0;
            if (kAdam) so1.v[v & (kSV - 1)] = x.s1;
          }
          if (kState) st_state(srow[k] + coff, tile_pack<float>(so0));
          if (kAdam) st_state(srow[k] + a.table_stride + coff, tile_pack<float>(so1));
          st_global_nt<raw_t>(trow[k] + coff, tile_pack<T>(eo));
        }
      }
    }
    // a tile with a run that is not this kernel's (left to the long-run side, or past the end): RPS runs at a time, predicated —
    // rare (the last tile; under skew the tiles that hold a hot id), kept small so that it does not set the register budget
    for (int cbase = 0; cbase < row_vecs && !whole; cbase += kLpr) {  // > 1 trip only when a row has more pieces than a wave step covers
      const int c        = cbase + col;
      const int64_t coff = static_cast<int64_t>(min(c, row_vecs - 1)) * kVE;
#pragma unroll 1
      for (int s = 0; s < tile_runs; s += RPS) {
        // (the broadcasts need every lane: outside the guard)
        const T* g       = tile_lane_ptr<RPS>(my_grad, s, sub);
        T* trow          = tile_lane_ptr<RPS>(my_row, s, sub);
        float* srow      = kState ? tile_lane_ptr<RPS>(my_st, s, sub) : nullptr;
        const int32_t rs = static_cast<int32_t>(tile_bcast32<RPS>(static_cast<uint32_t>(my_s0), s, sub));
        int32_t ln       = static_cast<int32_t>(tile_bcast32<RPS>(static_cast<uint32_t>(my_len), s, sub));
        float b1 = 0.f, b2 = 0.f;
        if (kAdam) {
          b1 = __builtin_bit_cast(float, tile_bcast32<RPS>(__builtin_bit_cast(uint32_t, my_b1), s, sub));
          b2 = __builtin_bit_cast(float, tile_bcast32<RPS>(__builtin_bit_cast(uint32_t, my_b2), s, sub));
        }
        if (c >= row_vecs) ln = 0;
        if (ln <= 0) continue;
        tile_vals<kVE> acc = tile_unpack<T>(ld_grad(g + coff));   // first occurrence copied
        const raw_t ev = ld_global_nt<raw_t>(trow + coff);
        raw_t s0v{}, s1v{};
        if (kState) s0v = ld_state(srow + coff);
        if (kAdam) s1v = ld_state(srow + a.table_stride + coff);
        for (int32_t j = 1; j < ln; j++) {   // later occurrences added in receive order
          const tile_vals<kVE> gx = tile_unpack<T>(ld_grad(grad_row<T>(a, a.order[rs + j]) + coff));
#pragma unroll
          for (int v = 0; v < kVE; v++) acc.v[v] += gx.v[v];
        }
        const tile_vals<kVE> e_in = tile_unpack<T>(ev);
        tile_vals<kSV> s0_in{}, s1_in{}, so0{}, so1{};
        tile_vals<kVE> eo;
        if (kState) s0_in = tile_unpack<float>(s0v);
        if (kAdam) s1_in = tile_unpack<float>(s1v);
#pragma unroll
        for (int v = 0; v < kVE; v++) {
          opt_elem x;
          x.e  = e_in.v[v];
          x.s0 = kState ? s0_in.v[v & (kSV - 1)] : 0.f;
          x.s1 = kAdam ? s1_in.v[v & (kSV - 1)] : 0.f;
          opt_math<OPT>(a, x, acc.v[v], b1, b2);
          eo.v[v] = x.e;
          if (kState) so0.v[v & (kSV - 1)] = x.s0;
          if (kAdam) so1.v[v & (kSV - 1)] = x.s1;
        }
        if (kState) st_state(srow + coff, tile_pack<float>(so0));
        if (kAdam) st_state(srow + a.table_stride + coff, tile_pack<float>(so1));
        st_global_nt<raw_t>(trow + coff, tile_pack<T>(eo));
      }
    }
    if constexpr (RAGGED) {
      // the last dim % 4 floats of the tile's rows: lane = run, 4-byte accesses, duplicates in receive order
      if (lane < tile_runs && my_len > 0) {
        for (int c = row_vecs * kVE; c < static_cast<int>(a.dim); c++) {
          float acc = my_grad[c];   // first occurrence copied
          for (int32_t j = 1; j < my_len; j++) acc += grad_row<T>(a, a.order[my_s0 + j])[c];
          update_elem<OPT, T>(a, my_row, my_st, c, load_elem<OPT, T>(a, my_row, my_st, c), acc, my_b1, my_b2);
        }
      }
    }
  }
}

template <typename IdxT, int OPT, bool VEC4>
__global__ __launch_bounds__(kBlock) void step_long_kernel(opt_params p)
{
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) float tile[2][kTileRows * kSliceCols];
  const wm_optimizer_args& a = p.a;
  const int n_long           = *p.long_count;
  const IdxT* ids            = static_cast<const IdxT*>(a.ids);
  const int col0             = blockIdx.y * kSliceCols;
  const int cols             = min(kSliceCols, static_cast<int>(a.dim) - col0);
  // a tile is kTileRows x 32 floats. Scalar mapping: thread -> (row t/32 + 8 i, column t%32), 32 loads per thread.
  // Vector mapping (16-byte aligned rows): thread -> (row t/8 + 32 i, float4 t%8), 8 loads per thread.
  constexpr int kLoads  = VEC4 ? kTileRows * kSliceCols / 4 / kBlock : kTileRows * kSliceCols / kBlock;
  constexpr int kRowStep = VEC4 ? kBlock / (kSliceCols / 4) : kBlock / kSliceCols;
  const int c_ld = VEC4 ? (threadIdx.x & 7) * 4 : (threadIdx.x & (kSliceCols - 1));
  const int r_ld = VEC4 ? (threadIdx.x >> 3) : (threadIdx.x >> 5);
  const int c_cl = VEC4 ? c_ld : min(c_ld, cols - 1);  // VEC4 is only chosen when every slice is full

  for (int li = blockIdx.x; li < n_long; li += gridDim.x) {
    const long_run_entry ent = p.long_list[li];
    const int64_t u          = ent.run;
    const int64_t local      = static_cast<int64_t>(ids[u]) - a.local_entry_offset;
    const int32_t s0         = a.run_starts[u];
    const int32_t s1         = a.run_starts[u + 1];
    float acc                = 0.f;
    typename std::conditional<VEC4, f4, float>::type stage[kLoads];
    // unconditional loads (row index clamped into the run): all row addresses first, then all rows
    auto load_tile = [&](int32_t base) {
      int32_t o[kLoads];
#pragma unroll
      for (int i = 0; i < kLoads; i++) o[i] = a.order[min(base + r_ld + kRowStep * i, s1 - 1)];
#pragma unroll
      for (int i = 0; i < kLoads; i++) {
        const float* src = grad_row(a, o[i]) + col0 + c_cl;
        if constexpr (VEC4)
          stage[i] = *reinterpret_cast<const f4*>(src);
        else
          stage[i] = *src;
      }
    };
    load_tile(s0);
    int buf = 0;
    for (int32_t base = s0; base < s1; base += kTileRows) {
#pragma unroll
      for (int i = 0; i < kLoads; i++) {
        float* dst = &tile[buf][(r_ld + kRowStep * i) * kSliceCols + c_ld];
        if constexpr (VEC4)
          *reinterpret_cast<f4*>(dst) = stage[i];
        else
          *dst = stage[i];
      }
      __syncthreads();
      if (base + kTileRows < s1) load_tile(base + kTileRows);  // in flight while this tile is folded
      if (threadIdx.x < cols) {
        const int32_t rows = min(kTileRows, s1 - base);
        int32_t r          = 0;
        if (base == s0) {
          acc = tile[buf][threadIdx.x];  // first occurrence is copied, not added to 0
          r   = 1;
        }
        // 16 LDS reads are issued back to back, then folded one by one: the add chain (not the LDS
        // latency) is what paces a long run
        for (; r < rows; r += 16) {
          float v[16];
#pragma unroll
          for (int k = 0; k < 16; k++) v[k] = tile[buf][min(r + k, kTileRows - 1) * kSliceCols + threadIdx.x];
#pragma unroll
          for (int k = 0; k < 16; k++)
            if (r + k < rows) acc += v[k];
        }
      }
      buf ^= 1;  // the other buffer was last read before the previous barrier: safe to overwrite
    }
    if (threadIdx.x < cols) apply_optimizer<OPT>(a, local, col0 + threadIdx.x, acc, ent.beta1t, ent.beta2t);
    __syncthreads();
  }
}

// step_long_kernel restructured for rows of whole 16-byte pieces. What paces a long run (measured on the Zipf(1.05) batch
// whose hottest id has 527 k duplicates, 8.8 ms in the kernel above; in-kernel cycle counters, experiments/*):
//   (1) one CU pulls only ~25-40 GB/s out of HBM whatever it keeps in flight (a 128-column, whole-row variant with a
//       3-tile LDS-DMA ring ran at 23 GB/s per workgroup and was SLOWER): the fetch of one run has to be spread over
//       CUs, so a workgroup takes a column slice of 128 bytes per row, one workgroup per (run, slice);
//   (2) the kernel above keeps ONE tile in flight and pays two dependent HBM latencies per tile (order[] entries, then
//       rows): 7 GB/s per workgroup;
//   (3) a wave pays ~6 cycles per row for its LDS read and ~6 for the dependent add and overlaps neither with the other,
//       however the reads are scheduled (experiments/fold_microbench.hip) — per ROW, whatever the width of the slice.
// So: tiles travel global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave instruction = 8 rows of the slice)
// into a ring of tile buffers — no staging registers, and because the compiler does not track LDS-DMA results the waits
// are hand-counted `s_waitcnt vmcnt(N)` + raw `s_barrier`, so the whole ring really stays in flight (a register ring
// does not survive hipcc's s_waitcnt placement: it drains at every loop trip). Loads return in issue order (vmcnt), so
// order[] reads interleaved with row reads would drain the ring: row addresses come out of LDS, where the order[]
// entries are kept a chunk of 2048 rows at a time in two alternating buffers — wave 0 fetches chunk c + 1 into
// registers while chunk c is folded and drops it into the free buffer, so the ring runs through a whole run without
// draining (refilling one buffer between chunks cost ~8 us per 4096 rows, 1.1 of 4.5 ms on the hottest run). Two waves
// fold alternate tiles: while one adds the 128 rows it holds in registers to the running sums, the other reads its next
// tile out of LDS; the sums change hands through LDS at the per-tile barrier. Summation order per element is unchanged
// (receive order; starting from -0.0f is the same as copying the first row, and rows past the end of a run are read as
// -0.0f, which adds nothing): results stay bit-identical. History on the Zipf batch, kernel alone: 8.8 ms (kernel
// above) -> 4.5 (LDS-DMA ring, 256-byte slices) -> 3.5 (order[] double-buffered) -> 3.4 (two folding waves) -> 3.0
// (128-byte slices, which only pay once the fold is off the critical path; 64-byte slices: no faster alone and they
// take twice the CUs away from step_short_kernel running beside it).

constexpr int kOrdChunk  = 2048;   // order[] entries per LDS buffer (two buffers)
constexpr int kRingBytes = 131072; // LDS given to the tile ring
constexpr int kSlice4Bytes = 128;  // bytes of every row per workgroup: 32 fp32 / 64 16-bit columns. One CU pulls a limited
                                   // number of bytes per second out of HBM whatever it keeps in flight, so a run is
                                   // spread over dim * elt / 128 workgroups
template <typename T>
constexpr int slice4_cols() { return kSlice4Bytes / static_cast<int>(sizeof(T)); }
constexpr int kTile4Rows = 128;    // rows per LDS tile = registers of a folding wave (32 KiB of fp32 slices, 16 KiB of 16-bit ones)
constexpr size_t kLong4LdsBytes = static_cast<size_t>(kRingBytes) + 2 * kOrdChunk * 4 + 64 * 4;  // + the partial sums

constexpr int kLongProducers = 4;                          // waves that only fetch
constexpr int kLongFolders   = 2;                          // waves 0 and 1 fold alternate tiles
constexpr int kLongBlock     = 64 * (kLongProducers + kLongFolders);
static_assert(kLongBlock == dense_fold::kBlock && kLong4LdsBytes >= dense_fold::shape<kDenseR, kDenseS>::kLdsBytes,
              "the first workgroups of step_long4_kernel fold the dense copies (kernels/long_dense.cuh): same shape, enough LDS");

template <typename IdxT, int OPT, typename T = float>
__global__ __launch_bounds__(kLongBlock) void step_long4_kernel(opt_params p)
{
  extern __shared__ __attribute__((aligned(16))) float lds4[];
  constexpr int S          = slice4_cols<T>();                          // columns per slice: 32 (fp32) / 64 (16-bit)
  constexpr int kRowBytes  = kSlice4Bytes;                              // bytes of every row a workgroup handles
  constexpr int kRing      = kRingBytes / (kTile4Rows * kRowBytes) > 8 ? 8 : kRingBytes / (kTile4Rows * kRowBytes);  // tiles in the ring
  constexpr int kE16       = 16 / static_cast<int>(sizeof(T));          // elements per 16-byte piece
  T* const tiles       = reinterpret_cast<T*>(lds4);                                        // [kRing][kTile4Rows][S]
  int32_t* const ord_s = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(lds4) + kRingBytes);  // [2][kOrdChunk]
  const wm_optimizer_args& a = p.a;
  const int n_long           = *p.long_count;
  const IdxT* ids            = static_cast<const IdxT*>(a.ids);
  const int col0             = blockIdx.y * S;
  const int cols             = min(S, static_cast<int>(a.dim) - col0);  // a whole number of 16-byte pieces
  constexpr int kLpr         = kRowBytes / 16;                                // lanes per row (16-byte pieces)
  constexpr int kRpp         = 64 / kLpr;                                     // rows per wave instruction
  constexpr int kLoads       = kTile4Rows / (kRpp * kLongProducers);          // pieces per producer lane per tile (8)
  static_assert(kLoads * kRing < 64, "vmcnt field");
  const int lane             = threadIdx.x & 63;
  // Waves 0 and 1 fold, waves 2 .. fetch. One wave pays ~6 cycles per row for the LDS read and ~6 for the dependent add
  // and overlaps neither with the other (experiments/fold_microbench.hip), so two waves take turns: while the owner of
  // tile t adds its 128 (256) rows out of REGISTERS to the running sums, the other one reads tile t + 1 from LDS into
  // its registers; the sums change hands through 256 bytes of LDS at the barrier that ends every tile anyway.
  const int wave_id   = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const bool producer = wave_id >= kLongFolders;
  const int wv        = wave_id - kLongFolders;
  const int c_safe    = min((lane % kLpr) * kE16, cols - kE16);  // lanes past a narrow slice re-read its last piece
  const int r_lane    = lane / kLpr;
  const bool folder   = lane < cols;  // one column per lane of a folding wave
  float* const acc_s  = reinterpret_cast<float*>(ord_s + 2 * kOrdChunk);  // [64] running sums between the two folders

  constexpr int kTpc   = kOrdChunk / kTile4Rows;  // tiles per chunk of order[] entries
  constexpr int kPre   = kOrdChunk / 64;          // entries per lane of wave 0 when it carries a whole chunk in registers
  static_assert((kOrdChunk & (kOrdChunk - 1)) == 0 && kOrdChunk % kTile4Rows == 0 && kTpc > kRing + 1 && kTpc % 2 == 0, "a chunk is a whole number of tiles, more than the ring holds");

  int first_li = blockIdx.x, li_step = gridDim.x;
  if constexpr (std::is_same<T, float>::value) {
    if (p.dense_jobs != nullptr) {
      // the first dense_x workgroups of every slice: the runs with a dense copy (kernels/long_dense.cuh), started first
      if (static_cast<int>(blockIdx.x) < p.dense_x) {
        auto ep = [&](const dense_fold::job& jb, int col, float acc) {
          const long_run_entry ent = p.long_list[jb.user];
          const int64_t local      = static_cast<int64_t>(ids[ent.run]) - a.local_entry_offset;
          apply_optimizer<OPT, float>(a, local, col, acc, ent.beta1t, ent.beta2t);
        };
        // (8-column slices: the ring then holds 4096 rows, ~10 us of the chain — beside the tile kernel a load takes ~7 us;
        // with 32-column slices the 527 k-row run took 2.2 ms here against 1.28 ms alone: profiles/r06_fold5_harness_slices.txt)
        dense_fold::fold_part<kDenseR, kDenseS>(p.dense_jobs, min(p.long_count[1], p.dense_max_jobs), static_cast<int>(a.dim), p.dense_buf,
                                                ep, static_cast<int>(blockIdx.y) * p.dense_x + static_cast<int>(blockIdx.x),
                                                p.dense_x * static_cast<int>(gridDim.y), lds4);
        return;
      }
      first_li -= p.dense_x, li_step -= p.dense_x;
    }
  }
  for (int li = first_li; li < n_long; li += li_step) {
    const long_run_entry ent = p.long_list[li];
    if (ent.dense != 0) continue;   // folded through its dense copy by the first workgroups
    const int64_t u          = ent.run;
    const int64_t local      = static_cast<int64_t>(ids[u]) - a.local_entry_offset;
    const int32_t s0         = a.run_starts[u];
    const int32_t s1         = a.run_starts[u + 1];
    const int32_t run_rows   = s1 - s0;
    const int n_tiles        = (run_rows + kTile4Rows - 1) / kTile4Rows;
    const int n_chunks       = (run_rows + kOrdChunk - 1) / kOrdChunk;
    // order[] entries reach the producers through two LDS buffers of one chunk each (chunk c in buffer c % 2), so the
    // tile ring never drains inside a run: chunk 0 is staged by everybody, every later chunk is fetched by wave 0 into
    // registers a whole chunk ahead (plain loads — wave 0 has no LDS-DMA of its own to keep count of) and dropped into
    // its buffer while chunk c - 1 is being folded. Entries past the end of the run repeat its last row.
    __syncthreads();  // previous run: every tile folded, every LDS-DMA piece landed (vmcnt(0) below)
    for (int i = threadIdx.x; i < min(kOrdChunk, run_rows); i += kLongBlock) ord_s[i] = a.order[s0 + i];
    int32_t pre[kPre];
    auto prefetch_chunk = [&](int c) {  // wave 0 only
#pragma unroll
      for (int k = 0; k < kPre; k++) pre[k] = a.order[min(s0 + c * kOrdChunk + k * 64 + lane, s1 - 1)];
    };
    auto drop_chunk = [&](int c) {
#pragma unroll
      for (int k = 0; k < kPre; k++) ord_s[(c & 1) * kOrdChunk + k * 64 + lane] = pre[k];
    };
    if (wave_id == 0 && n_chunks > 1) prefetch_chunk(1);
    __syncthreads();
    // tile t -> ring slot t % kRing: kLoads LDS-DMA pieces per producer lane; producer w, piece i lands as the kRpp
    // adjacent tile rows starting at kRpp (w + kLongProducers i) (wave-uniform LDS base + lane * 16 B). Rows are
    // clamped into the run: tiles past its end re-read the last row, harmlessly — so every producer always has
    // the same number of pieces in flight.
    auto issue = [&](int t) {
      T* slot = tiles + (t % kRing) * (kTile4Rows * S);
      int32_t o[kLoads];
#pragma unroll
      for (int i = 0; i < kLoads; i++) {
        const int32_t row = min(t * kTile4Rows + kRpp * (wv + kLongProducers * i) + r_lane, run_rows - 1);
        o[i]              = ord_s[row & (2 * kOrdChunk - 1)];  // chunk c lives in buffer c % 2
      }
#pragma unroll
      for (int i = 0; i < kLoads; i++) {
        const T* src = grad_row<T>(a, o[i]) + col0 + c_safe;
        typedef __attribute__((address_space(1))) void gvoid;
        typedef __attribute__((address_space(3))) void lvoid;
        __builtin_amdgcn_global_load_lds((gvoid*)src, (lvoid*)(slot + kRpp * (wv + kLongProducers * i) * S), 16, 0, 0);
      }
    };
    // the rows of one tile, in the registers of the wave that owns it
    float v[kTile4Rows];
    // Rows past the end of the run become -0.0f: x + (-0.0f) == x bit for bit for every x, so the owner adds all
    // kTile4Rows registers without predicates (and a register array cannot be indexed by a loop counter anyway).
    auto read_tile = [&](int t) {
      const T* src = tiles + (t % kRing) * (kTile4Rows * S) + lane;
#pragma clang loop unroll(full)
      for (int k = 0; k < kTile4Rows; k++) v[k] = load_wide<T>(src[k * S]);
      const int32_t rows = run_rows - t * kTile4Rows;
      if (rows < kTile4Rows) {
#pragma clang loop unroll(full)
        for (int k = 0; k < kTile4Rows; k++) v[k] = k < rows ? v[k] : -0.0f;
      }
    };
    // Schedule. Before barrier t every slot is in use: tile t sits in its owner's registers (its slot is free), tile
    // t + 1 has landed, tiles t + 2 .. t + kRing - 1 are in flight. After barrier t the producers refill the slot of
    // tile t with tile t + kRing, the owner of tile t (wave t % 2) takes the sums from LDS, adds its rows and puts the
    // sums back, the other folder reads tile t + 1. A lane past the slice (lane >= cols) folds rubbish nobody reads.
    // (two separate loops with the same number of barriers: in one loop the registers of a tile would stay reserved in
    // the producers' code too and hipcc spills to scratch — vector memory traffic that would also wreck the vmcnt count)
    float acc = 0.f;
    if (producer) {
#pragma unroll
      for (int j = 0; j < kRing; j++) issue(j);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoads * (kRing - 1)) : "memory");  // tile 0 has landed
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      for (int tt = 0; tt < n_tiles; tt++) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoads * (kRing - 2)) : "memory");  // tile tt + 1 has landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(tt + kRing);
      }
    } else {
      // one barrier per tile for everybody; a folding wave alternates between adding its own tile and reading its next
      auto tile_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // sums / order entries written, tile in registers
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      };
      auto add_tile = [&](int t) {
        // the first occurrence is copied, not added to 0 — which is what starting from -0.0f does: -0.0f + x == x
        acc = t > 0 ? acc_s[lane] : -0.0f;
#pragma clang loop unroll(full)
        for (int k = 0; k < kTile4Rows; k++) acc += v[k];
        if (t + 1 < n_tiles) acc_s[lane] = acc;
      };
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (wave_id == 0) {  // even tiles
        read_tile(0);
        for (int tt = 0; tt < n_tiles; tt += 2) {
          tile_barrier();
          // first tile of chunk c: buffer (c + 1) % 2 was last read for the tiles of chunk c - 1, all issued long before
          // the barrier just passed (kTpc > kRing + 1); its new content is first read kTpc - kRing barriers from now
          if (tt % kTpc == 0) {
            const int c = tt / kTpc;
            if (c + 1 < n_chunks) {
              drop_chunk(c + 1);
              if (c + 2 < n_chunks) prefetch_chunk(c + 2);
            }
          }
          add_tile(tt);
          if (tt + 1 < n_tiles) {
            tile_barrier();
            if (tt + 2 < n_tiles) read_tile(tt + 2);
          }
        }
      } else {  // odd tiles
        for (int tt = 0; tt < n_tiles; tt += 2) {
          tile_barrier();
          if (tt + 1 < n_tiles) {
            read_tile(tt + 1);
            tile_barrier();
            add_tile(tt + 1);
          }
        }
      }
    }
    if (producer) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // clamped tail tiles must not land in the next run
    if (wave_id == ((n_tiles - 1) & 1) && folder)
      apply_optimizer<OPT, T>(a, local, col0 + lane, acc, ent.beta1t, ent.beta2t);
  }
}

// Lists the runs of more than kLongRun rows for the long-run kernel, with their LazyAdam beta powers (advanced here, the
// same two multiplications step_short_kernel does for the rows it owns). A kernel of its own so that the long-run kernel
// does not have to wait for step_short_kernel: the two then run side by side on two streams (see long_lane) — one is a
// handful of workgroups chewing through a few very long dependent chains, the other wants the whole memory system.
template <typename IdxT>
__global__ void mark_long_runs_kernel(opt_params p)
{
  if (p.split_ctl != nullptr && p.split_ctl[split::kCtlOverflow] == 0 && p.split_ctl[split::kCtlRadixBuckets] == 0 &&
      p.long_threshold >= split::kMaxDup)
    return;
  // detached side (its stream is ordered behind the sort only by a wave that waits for "runs are final"): list nothing unless
  // that word says FINAL — not after a wait that gave up (0) and not after a sort that reported a timeout (2). The long-run
  // kernels behind this one then find an empty list.
  if (p.detached_side != 0 && p.split_ctl != nullptr &&
      __hip_atomic_load(&p.split_ctl[split::kCtlSortDone], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u)
    return;
  const wm_optimizer_args& a = p.a;
  const int64_t count        = p.n_unique ? *p.n_unique : a.count;
  const IdxT* ids            = static_cast<const IdxT*>(a.ids);
  for (int64_t u = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; u < count;
       u += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    if (a.run_starts[u + 1] - a.run_starts[u] <= kLongRun) continue;
    float beta1t = 0.f, beta2t = 0.f;
    if (a.type == WHOLEMEMORY_OPT_LAZY_ADAM) {
      const int64_t local             = static_cast<int64_t>(ids[u]) - a.local_entry_offset;
      beta1t                          = a.per_row_state[local * 2 + 0] * a.beta1;
      beta2t                          = a.per_row_state[local * 2 + 1] * a.beta2;
      a.per_row_state[local * 2 + 0] = beta1t;
      a.per_row_state[local * 2 + 1] = beta2t;
    }
    const int slot    = atomicAdd(p.long_count, 1);
    int32_t dense     = 0;
    const int32_t len = a.run_starts[u + 1] - a.run_starts[u];
    if (p.dense_jobs != nullptr && len >= p.dense_min) {
      // room in the dense buffer while it lasts (the cursor may overshoot: whoever finds no room leaves the run to step_long4_kernel)
      const int64_t want = dense_fold::dense_floats(len, a.dim);
      const int64_t off  = static_cast<int64_t>(atomicAdd(reinterpret_cast<unsigned long long*>(p.long_count + 2), static_cast<unsigned long long>(want)));
      if (off + want <= p.dense_cap) {
        const int j = atomicAdd(p.long_count + 1, 1);
        if (j < p.dense_max_jobs) {
          p.dense_jobs[j] = dense_fold::job{a.run_starts[u], len, off, slot, 0};
          dense           = 1;
        } else {
          atomicSub(p.long_count + 1, 1);   // (cannot happen: max_jobs covers cap / dense_min)
        }
      }
    }
    p.long_list[slot] = long_run_entry{static_cast<int32_t>(u), beta1t, beta2t, dense};
  }
}

// The very long runs of an ordered fp32 fold through their dense transposed copies (kernels/long_dense.cuh): the copy is a launch
// of its own on the long-run side's stream (the whole chip, at memory speed, no LDS); the fold is the FIRST dense_x workgroups
// per slice of step_long4_kernel (same workgroup shape: six waves, a 128 KiB ring), so that the hottest chains start first and
// run beside the other long runs — as a launch of its own in front of step_long4_kernel, that kernel only started when the
// 1.3 ms chain had finished, behind a tile kernel that by then filled the chip (3.32 -> 3.41 ms, profiles/r06_dense_fold_ab.txt).
// The folding workgroups have to be ON the machine before the tile kernel fills it (launched behind it they never find a CU:
// see long_lane), and their input is the copy's output — so the caller's stream waits for the COPY (the event that used to
// follow the listing kernel follows it now): the tile kernel starts ~0.15 ms later and then has the memory system to itself. (First version, measured and dropped: copy and fold as ONE launch, every workgroup copying its share and
// the folding ones waiting for a counter — every workgroup of that grid carries the fold's 128 KiB of LDS, so the copy ran one
// workgroup per CU: Zipf call 3.13 -> 4.44 ms, profiles/r06_dense_fold_ab.txt.)
template <typename IdxT>
__global__ __launch_bounds__(256) void step_dense_copy_kernel(opt_params p)
{
  const wm_optimizer_args& a = p.a;
  auto row_of = [&](int32_t o) { return grad_row<float>(a, o); };
  const int n = min(p.long_count[1], p.dense_max_jobs);
  dense_fold::copy_part(p.dense_jobs, n, a.order, row_of, static_cast<int>(a.dim), p.dense_buf,
                        static_cast<int64_t>(blockIdx.x) * 8 + (threadIdx.x >> 5), static_cast<int64_t>(gridDim.x) * 8,
                        static_cast<int>(threadIdx.x & 31));
}
// Side stream + fork/join events for the long-run kernels, created once per process. fork(): the side stream waits for
// everything the caller's stream has queued so far; join(): the caller's stream waits for the side stream.
struct long_lane {
  std::mutex mu;  // one fork .. join sequence at a time: the events are shared
  hipStream_t stream = nullptr;
  hipEvent_t forked = nullptr, marked = nullptr, joined = nullptr;
  bool ok = false;
  long_lane()
  {
    int least = 0, greatest = 0;  // the side stream gets the highest priority: its few workgroups must not queue up behind
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);  // the thousands of step_short_kernel
    ok = hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, greatest) == hipSuccess &&
         hipEventCreateWithFlags(&forked, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&marked, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&joined, hipEventDisableTiming) == hipSuccess;
  }
  static long_lane& get() { return per_device<long_lane>(); }   // one per device, as sort_lane
  bool fork(hipStream_t from)
  {
    return ok && hipEventRecord(forked, from) == hipSuccess && hipStreamWaitEvent(stream, forked, 0) == hipSuccess;
  }
  bool join(hipStream_t into)
  {
    return hipEventRecord(joined, stream) == hipSuccess && hipStreamWaitEvent(into, joined, 0) == hipSuccess;
  }
  // How many long runs the last finished call listed (pinned; written by a device-to-host copy behind the long-run kernels,
  // read by the host without synchronising, so it lags a call or two): starts at 1 = "expect long runs".
  // The side stream costs the USUAL call — the one whose list stays empty — two cross-stream hand-overs in front of the tile
  // kernel (fork -> fill + listing -> `marked` -> tile: ~40 us, profiles/r05_grad_timeline_serial.txt). While the previous
  // calls listed nothing, everything is queued on the caller's stream instead (fill, listing, the long-run kernels finding an
  // empty list, tile kernel: ~20 us). A wrong guess costs time only, once: the long runs are then folded BEFORE the tile
  // kernel instead of beside it, and the next call sees the count. WM_STEP_SERIAL=1 / 0 forces the choice.
  volatile int32_t* listed = nullptr;
  bool expect_long()
  {
    if (listed == nullptr) {
      void* h = nullptr;
      if (hipHostMalloc(&h, 64, hipHostMallocDefault) != hipSuccess) return true;
      listed  = static_cast<volatile int32_t*>(h);
      *listed = 1;
    }
    return *listed != 0;
  }
  // (two words: [0] the listed runs, [1] — ordered fold — how many of them went through a dense copy: wholememory_ext_dense_fold_last)
  void report(const int32_t* long_count_dev, hipStream_t s)
  {
    if (listed != nullptr)
      (void)hipMemcpyAsync(const_cast<int32_t*>(listed), long_count_dev, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s);
  }
};

// step_short_kernel fills every wave slot of the chip with persistent waves: whatever is to run NEXT to it has to be on
// the machine first. So the caller's stream waits for the (tiny) listing kernel; the long-run kernel is then queued on
// the high-priority side stream at the moment step_short_kernel is queued on the caller's.
// detached long-run side: its stream is not ordered behind the sort by anything but this wave, which waits for the sort's
// "runs are final" word (split_join_kernel sets it on the caller's stream)
inline void wait_for_final_runs(const opt_params& p, hipStream_t lstream)
{
  if (p.detached_side)
    hipLaunchKernelGGL(split::split_wait_kernel, dim3(1), dim3(64), 0, lstream, p.split_ctl + split::kCtlSortDone, 1u,
                       const_cast<uint32_t*>(p.split_ctl) + split::kCtlError, wait_limits().wait_polls,
                       sort_lane::get().host_err_dev);
}

template <typename IdxT>
void launch_mark_long_runs(const opt_params& p, hipStream_t stream, hipStream_t lstream)
{
  wait_for_final_runs(p, lstream);
  // two coalesced reads of run_starts[] per run, on the caller's critical path: a grid-stride loop over at most 4096 workgroups
  // (one run per thread and 37 k workgroups for 9.5 M runs took 15 us — the time to hand out 148 k one-load waves)
  const int blocks = static_cast<int>(std::min<int64_t>((p.a.count + 255) / 256, 4096));
  hipLaunchKernelGGL((mark_long_runs_kernel<IdxT>), dim3(std::max(blocks, 1)), dim3(256), 0, lstream, p);
  // the dense copies of the very long runs (step_dense_copy_kernel: leaves at once when none was listed): in front of the event
  // the caller's stream waits for — see step_dense_copy_kernel
  if (p.dense_jobs != nullptr) hipLaunchKernelGGL((step_dense_copy_kernel<IdxT>), dim3(1024), dim3(256), 0, lstream, p);
  if (lstream != stream && !p.detached_side) {
    (void)hipEventRecord(long_lane::get().marked, lstream);
    (void)hipStreamWaitEvent(stream, long_lane::get().marked, 0);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// "Tree" fold of long duplicate runs (wm_optimizer_args::fold_mode = 1; WM_GRAD_FOLD=tree) — round 3.
// The ordered fold above is one fp32 chain per element and run: the 527 k duplicates of the hottest row of a Zipf(1.05)
// batch are 527 k dependent adds, ~8 cycles each, whatever feeds them (2.2-2.9 ms beside a 2.6 ms step_tile_kernel). When the
// caller does not need the reference's association order, a run is cut by ROWS instead: segments of kTreeSeg rows, one
// workgroup each, whose 256 threads sum kTreeSeg / row-slots rows apiece in registers (4 row loads in flight per thread) and
// meet in LDS; runs of one segment apply the optimizer right there, longer ones park one fp32 partial row per segment and a
// second small kernel adds the partials of a run in segment order and applies the optimizer. Every sum has a fixed order, so
// results are deterministic — and exact whenever every partial sum is exactly representable (integer-valued gradients) —
// but they are not the reference's bits: |tree - ordered| is bounded by the usual forward error of fp32 summation,
// (number of terms) x 2^-24 x sum |g_i|, and the tree's own error is the smaller of the two (shorter chains).
// Because the fold is bandwidth-bound, not latency-bound, the threshold between "the tile kernel folds it in its lanes" and
// "listed for the long-run side" drops from 256 rows to kTreeMin.
constexpr int kTreeSeg     = 512;   // rows per segment, at least
constexpr int kTreeMaxSegs = 256;   // segments per run, at most (a longer run gets longer segments): bounds the combine chain
constexpr int kTreeMin     = 128;   // runs of more rows than this are listed (WM_GRAD_FOLD_MIN; swept 16 ... 256 on the Zipf batch:
                                    // 2.58 / 2.34 / 2.23 / 2.19 / 2.25 ms per call at 16 / 32 / 64 / 128 / 256)

struct tree_run {
  int32_t run;       // index into the unique ids
  float beta1t, beta2t;
  int32_t pbase;     // first partial row of the run, or -1 for a run of one segment
  int32_t seg_rows;  // rows per segment
  int32_t nseg;
  int32_t pad[2];
};
struct tree_seg {
  int32_t run_slot;  // index into the listed runs
  int32_t seg;       // segment number inside the run
};
struct tree_ws_view {
  int32_t* counters;       // [0] listed runs, [1] segments, [2] partial rows, [3] runs of several segments
  tree_run* runs;
  int32_t* multi;          // slots of the runs of several segments (what tree_combine_kernel walks)
  tree_seg* segs;
  float* partials;         // [partial rows][dim rounded up to 4]
  int64_t max_runs, max_segs, max_partials;
};
__host__ __device__ inline int64_t tree_dim_pad(int64_t dim) { return (dim + 3) / 4 * 4; }
inline void tree_bounds(int64_t n_recv, int threshold, int64_t* max_runs, int64_t* max_segs, int64_t* max_partials)
{
  *max_runs     = n_recv / (threshold + 1) + 2;
  *max_partials = 2 * (n_recv / kTreeSeg) + 2;            // only runs of >= 2 segments park partials
  *max_segs     = *max_runs + n_recv / kTreeSeg + 2;
}
inline size_t tree_ws_bytes(int64_t n_recv, int64_t dim, int threshold)
{
  int64_t r, s, q;
  tree_bounds(n_recv, threshold, &r, &s, &q);
  return 64 + static_cast<size_t>(r) * (sizeof(tree_run) + sizeof(int32_t)) + static_cast<size_t>(s) * sizeof(tree_seg) + 128 +
         static_cast<size_t>(q) * static_cast<size_t>(tree_dim_pad(dim)) * sizeof(float);
}
inline tree_ws_view tree_ws_carve(void* ws, int64_t n_recv, int threshold)
{
  tree_ws_view v;
  tree_bounds(n_recv, threshold, &v.max_runs, &v.max_segs, &v.max_partials);
  char* p    = static_cast<char*>(ws);
  v.counters = reinterpret_cast<int32_t*>(p);
  p += 64;
  v.runs = reinterpret_cast<tree_run*>(p);
  p += static_cast<size_t>(v.max_runs) * sizeof(tree_run);
  v.multi = reinterpret_cast<int32_t*>(p);
  p += (static_cast<size_t>(v.max_runs) * sizeof(int32_t) + 15) & ~size_t(15);
  v.segs = reinterpret_cast<tree_seg*>(p);
  p += static_cast<size_t>(v.max_segs) * sizeof(tree_seg);
  p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 63) & ~uintptr_t(63));
  v.partials = reinterpret_cast<float*>(p);
  return v;
}

// lists the runs of more than `threshold` rows with their segments. One lane per run finds and registers it (the beta powers of
// LazyAdam are advanced here as in mark_long_runs_kernel); the segment entries of every listed run of a wave are then written
// by the 64 lanes together — the hottest run of a Zipf batch has hundreds, and one lane writing them alone was 0.25 ms of
// the call's critical path.
template <typename IdxT>
__global__ __launch_bounds__(256) void tree_mark_kernel(opt_params p, tree_ws_view w, int threshold)
{
  // (no bucket of the split sort held a run of more than kMaxDup ids: nothing to list)
  if (p.split_ctl != nullptr && p.split_ctl[split::kCtlOverflow] == 0 && p.split_ctl[split::kCtlRadixBuckets] == 0 &&
      threshold >= split::kMaxDup)
    return;
  // detached side (its stream is ordered behind the sort only by a wave that waits for "runs are final"): list nothing unless
  // that word says FINAL — not after a wait that gave up (0) and not after a sort that reported a timeout (2). The long-run
  // kernels behind this one then find an empty list.
  if (p.detached_side != 0 && p.split_ctl != nullptr &&
      __hip_atomic_load(&p.split_ctl[split::kCtlSortDone], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u)
    return;
  const wm_optimizer_args& a = p.a;
  const int64_t count        = p.n_unique ? *p.n_unique : a.count;
  const IdxT* ids            = static_cast<const IdxT*>(a.ids);
  const int64_t u            = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int lane             = threadIdx.x & 63;
  int32_t len                = 0;
  if (u < count) len = a.run_starts[u + 1] - a.run_starts[u];
  int slot = -1, sbase = 0, nseg = 0, seg_rows = 0;
  float beta1t = 0.f, beta2t = 0.f;
  const bool listed = len > threshold;
  if (listed) {
    if (a.type == WHOLEMEMORY_OPT_LAZY_ADAM) {
      const int64_t local            = static_cast<int64_t>(ids[u]) - a.local_entry_offset;
      beta1t                         = a.per_row_state[local * 2 + 0] * a.beta1;
      beta2t                         = a.per_row_state[local * 2 + 1] * a.beta2;
      a.per_row_state[local * 2 + 0] = beta1t;
      a.per_row_state[local * 2 + 1] = beta2t;
    }
    nseg     = min((len + kTreeSeg - 1) / kTreeSeg, kTreeMaxSegs);
    seg_rows = (len + nseg - 1) / nseg;
    nseg     = (len + seg_rows - 1) / seg_rows;   // no empty segment at the end
  }
  // one atomic per counter and WAVE (a Zipf batch lists ~10^5 runs; one atomic each on four addresses serialised)
  const uint64_t lmask = __ballot(listed);
  if (lmask != 0) {   // wave-uniform
    const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const bool is_multi  = nseg > 1;
    const uint64_t mmask = __ballot(is_multi);
    // inclusive wave scans of nseg (all listed) and of nseg over the multi-segment runs
    int incl_all = nseg, incl_multi = is_multi ? nseg : 0;
    for (int d = 1; d < 64; d <<= 1) {
      const int va = __shfl_up(incl_all, d, 64), vm = __shfl_up(incl_multi, d, 64);
      if (lane >= d) incl_all += va, incl_multi += vm;
    }
    const int tot_all = __shfl(incl_all, 63, 64), tot_multi = __shfl(incl_multi, 63, 64);
    int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    if (lane == 0) {
      b0 = atomicAdd(&w.counters[0], __popcll(lmask));
      b1 = atomicAdd(&w.counters[1], tot_all);
      if (mmask != 0) {
        b2 = atomicAdd(&w.counters[2], tot_multi);
        b3 = atomicAdd(&w.counters[3], __popcll(mmask));
      }
    }
    b0 = __shfl(b0, 0, 64), b1 = __shfl(b1, 0, 64), b2 = __shfl(b2, 0, 64), b3 = __shfl(b3, 0, 64);
    if (listed) {
      slot            = b0 + __popcll(lmask & below);
      sbase           = b1 + incl_all - nseg;
      const int pbase = is_multi ? b2 + incl_multi - nseg : -1;
      w.runs[slot]    = tree_run{static_cast<int32_t>(u), beta1t, beta2t, pbase, seg_rows, nseg, {0, 0}};
      if (is_multi) w.multi[b3 + __popcll(mmask & below)] = slot;
    }
  }
  uint64_t pending = __ballot(slot >= 0);
  while (pending) {   // wave-uniform
    const int src   = __ffsll(static_cast<long long>(pending)) - 1;
    const int r_sl  = __shfl(slot, src, 64);
    const int r_sb  = __shfl(sbase, src, 64);
    const int r_ns  = __shfl(nseg, src, 64);
    for (int g = lane; g < r_ns; g += 64) w.segs[r_sb + g] = tree_seg{r_sl, g};
    pending &= pending - 1;
  }
}

// one segment per workgroup trip: thread t = (row slot t / LPR, 16-byte piece t % LPR); rows of a slot are added in order,
// four loads in flight, the row slots meet in LDS in slot order
#ifndef WM_TREE_DEPTH
#define WM_TREE_DEPTH 4
#endif
template <typename IdxT, int OPT, typename T>
__global__ __launch_bounds__(kBlock) void tree_fold_kernel(opt_params p, tree_ws_view w)
{
  constexpr int kTreeDepth = WM_TREE_DEPTH;
  constexpr int kVE = 16 / static_cast<int>(sizeof(T));
  __shared__ float red[kBlock * kVE];
  const wm_optimizer_args& a = p.a;
  const IdxT* ids            = static_cast<const IdxT*>(a.ids);
  const int n_seg            = w.counters[1];
  const int pieces           = static_cast<int>(a.dim / kVE);          // 16-byte pieces per row (the launcher checks dim % kVE == 0)
  int lpr                    = 1;
  while (lpr < pieces && lpr < kBlock) lpr <<= 1;                       // pieces handled per pass, a power of two <= 256
  const int slots            = kBlock / lpr;                            // row slots
  const int c                = threadIdx.x & (lpr - 1);
  const int rs               = threadIdx.x / lpr;
  const int64_t dpad         = tree_dim_pad(a.dim);
  for (int si = blockIdx.x; si < n_seg; si += gridDim.x) {
    const tree_seg sg   = w.segs[si];
    const tree_run ent  = w.runs[sg.run_slot];
    const int64_t u     = ent.run;
    const int32_t s0    = a.run_starts[u] + sg.seg * ent.seg_rows;
    const int32_t rows  = min(ent.seg_rows, a.run_starts[u + 1] - s0);
    const int64_t local = static_cast<int64_t>(ids[u]) - a.local_entry_offset;
    for (int pbase = 0; pbase < pieces; pbase += lpr) {
      const int piece = pbase + c;
      const bool live = piece < pieces;
      float acc[kVE];
#pragma unroll
      for (int v = 0; v < kVE; v++) acc[v] = 0.f;
      // kTreeDepth rows per thread in flight: the positions and then the rows are loaded UNCONDITIONALLY (a row slot past the
      // end of the segment repeats its last row, a thread past the end of the row its last piece) and only the adds are
      // predicated — a load under `if (rr < rows)` is a conditionally defined register, and hipcc then waited for every
      // position and every row before issuing the next one (round 3's shape: one 16-byte load in flight per thread)
      const int64_t poff = static_cast<int64_t>(min(piece, pieces - 1)) * kVE;
      for (int r0 = 0; r0 < rows; r0 += kTreeDepth * slots) {   // (block-uniform trip count)
        int32_t pos[kTreeDepth];
        tile_raw4 g[kTreeDepth];
#pragma unroll
        for (int q = 0; q < kTreeDepth; q++) pos[q] = a.order[s0 + min(r0 + rs + q * slots, rows - 1)];
#pragma unroll
        for (int q = 0; q < kTreeDepth; q++) g[q] = ld_global_nt<tile_raw4>(grad_row<T>(a, pos[q]) + poff);
#pragma unroll
        for (int q = 0; q < kTreeDepth; q++) {
          const bool in = r0 + rs + q * slots < rows;
          const tile_vals<kVE> gv = tile_unpack<T>(g[q]);
#pragma unroll
          for (int v = 0; v < kVE; v++) acc[v] = in ? acc[v] + gv.v[v] : acc[v];
        }
      }
      __syncthreads();   // (the previous pass / segment has read red[])
#pragma unroll
      for (int v = 0; v < kVE; v++) red[threadIdx.x * kVE + v] = acc[v];
      __syncthreads();
      if (rs == 0 && live) {
        float tot[kVE];
#pragma unroll
        for (int v = 0; v < kVE; v++) tot[v] = red[c * kVE + v];
        for (int s = 1; s < slots; s++) {
#pragma unroll
          for (int v = 0; v < kVE; v++) tot[v] += red[(s * lpr + c) * kVE + v];
        }
        if (ent.pbase < 0) {   // the whole run: apply the optimizer here
#pragma unroll
          for (int v = 0; v < kVE; v++)
            apply_optimizer<OPT, T>(a, local, static_cast<int64_t>(piece) * kVE + v, tot[v], ent.beta1t, ent.beta2t);
        } else {
          float* dst = w.partials + (static_cast<int64_t>(ent.pbase) + sg.seg) * dpad + static_cast<int64_t>(piece) * kVE;
#pragma unroll
          for (int v = 0; v < kVE; v++) dst[v] = tot[v];
        }
      }
    }
  }
}

// runs of several segments: the partial rows are added in segment order (8 loads in flight per thread), then the optimizer
// statement. One workgroup per run, one column per thread.
template <typename IdxT, int OPT, typename T>
__global__ __launch_bounds__(kBlock) void tree_combine_kernel(opt_params p, tree_ws_view w)
{
  const wm_optimizer_args& a = p.a;
  const IdxT* ids            = static_cast<const IdxT*>(a.ids);
  const int n_multi          = w.counters[3];
  const int64_t dpad         = tree_dim_pad(a.dim);
  for (int mi = blockIdx.x; mi < n_multi; mi += gridDim.x) {
    const tree_run ent = w.runs[w.multi[mi]];
    const int64_t local = static_cast<int64_t>(ids[ent.run]) - a.local_entry_offset;
    for (int64_t d = threadIdx.x; d < a.dim; d += kBlock) {
      const float* src = w.partials + static_cast<int64_t>(ent.pbase) * dpad + d;
      float acc        = src[0];
      for (int g = 1; g < ent.nseg; g += 8) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = g + q < ent.nseg ? src[static_cast<int64_t>(g + q) * dpad] : 0.f;
#pragma unroll
        for (int q = 0; q < 8; q++)
          if (g + q < ent.nseg) acc += v[q];
      }
      apply_optimizer<OPT, T>(a, local, d, acc, ent.beta1t, ent.beta2t);
    }
  }
}

inline int tree_threshold()
{
  const char* e = WM_AB_KNOB("WM_GRAD_FOLD_MIN");
  const int v   = e != nullptr ? atoi(e) : 0;
  return v >= 4 && v <= kLongRun ? v : kTreeMin;
}
// ordered (0) or tree (1): WM_GRAD_FOLD overrides, then the caller's fold_mode, then the default of the value dtype
inline int resolve_fold_mode(const wm_optimizer_args& a)
{
  const char* e = WM_KNOB("WM_GRAD_FOLD");
  if (e != nullptr && (e[0] == 't' || e[0] == 'T')) return 1;
  if (e != nullptr && (e[0] == 'o' || e[0] == 'O')) return 0;
  if (a.fold_mode >= 0) return a.fold_mode;
  return (a.value_dtype == WHOLEMEMORY_DT_HALF || a.value_dtype == WHOLEMEMORY_DT_BF16) ? 1 : 0;
}

template <typename IdxT, int OPT, typename T>
void launch_tree(const opt_params& p, hipStream_t stream, hipStream_t lstream)
{
  wait_for_final_runs(p, lstream);
  tree_ws_view w = tree_ws_carve(p.a.long_run_ws, p.a.count, p.long_threshold);
  if (p.split_ctl != nullptr) w.counters = p.long_count;   // (the split sort's control words: already zero)
  // one run per thread, no grid-stride loop: the grid must cover every run (callers keep count below 2^31 -> at most 2^23 blocks)
  const int mblocks    = static_cast<int>(std::min<int64_t>((p.a.count + 255) / 256, INT64_C(1) << 23));
  hipLaunchKernelGGL((tree_mark_kernel<IdxT>), dim3(std::max(mblocks, 1)), dim3(256), 0, lstream, p, w, p.long_threshold);
  if (lstream != stream && !p.detached_side) {
    (void)hipEventRecord(long_lane::get().marked, lstream);
    (void)hipStreamWaitEvent(stream, long_lane::get().marked, 0);
  }
  int fgrid = 2048;
  if (const char* e = WM_AB_KNOB("WM_TREE_GRID")) fgrid = std::max(1, atoi(e));
  hipLaunchKernelGGL((tree_fold_kernel<IdxT, OPT, T>), dim3(fgrid), dim3(kBlock), 0, lstream, p, w);
  hipLaunchKernelGGL((tree_combine_kernel<IdxT, OPT, T>), dim3(256), dim3(kBlock), 0, lstream, p, w);
}

// Launch shape of step_tile_kernel. Default: round 2's persistent grid of 8192 workgroups over tiles of 64 runs.
// WM_TILE_INORDER=1: the in-order shape of the row kernels (rows.hip: rows_op) — one tile of ONE batch of runs per wave
// (RPS x kU runs: 8 for SGD on 512-byte rows), as many workgroups as the upper bound of the run count takes (the true count is
// on the device: waves past it leave at once). Measured on whole 10 M-row calls, interleaved in one process
// (experiments/grad_inorder_ab.py, profiles/r03_grad_inorder_ab.txt): SGD uniform 3.281 vs 3.268 ms, Zipf 3.12-3.23 vs 3.09-3.10,
// LazyAdam 6.295 vs 6.305, fp16 x 256 3.143 vs 3.189, 256-byte rows 1.856 vs 1.877 — a wash: this kernel has no dense streamed
// side whose DRAM pages an ordered window could keep open (gradient rows arrive through order[], table rows are 1 in 10 of a
// sorted sweep), so the default stays. WM_TILE_RUNS forces the tile (a multiple of the batch).
inline void tile_launch_shape(int64_t count_bound, int vecs, int ku, int* tile_runs, int* tblocks)
{
  const int rps      = vecs > 32 ? 1 : vecs > 16 ? 2 : vecs > 8 ? 4 : 8;
  const int batch    = rps * ku;
  const char* io_env = WM_AB_KNOB("WM_TILE_INORDER");
  const bool inorder = io_env != nullptr && io_env[0] == '1';
  *tile_runs         = inorder ? std::min(64, batch) : 64;
  if (const char* e = WM_AB_KNOB("WM_TILE_RUNS")) {
    const int v = atoi(e);
    if (v >= batch && v <= 64 && v % batch == 0) *tile_runs = v;
  }
  const int64_t tiles = (count_bound + *tile_runs - 1) / *tile_runs;
  int b = static_cast<int>(std::min<int64_t>((tiles + 3) / 4, inorder ? INT64_C(0x7fffffff) : INT64_C(256 * 32)));
  if (const char* e = WM_AB_KNOB("WM_STEP_BLOCKS")) b = std::min(b, std::max(1, atoi(e)));
  *tblocks = std::max(b, 1);
}

template <typename IdxT, int OPT>
int launch_step_opt(const opt_params& p, int blocks, hipStream_t stream, hipStream_t lstream)
{
  const uint64_t gaddr = reinterpret_cast<uint64_t>(p.a.grads) | reinterpret_cast<uint64_t>(p.a.self_grads);
  const bool self_ok2  = p.a.self_grads == nullptr || p.a.self_grad_stride % 2 == 0;
  const bool self_ok4  = p.a.self_grads == nullptr || p.a.self_grad_stride % 4 == 0;
  const bool vec2      = p.a.dim % 2 == 0 && p.a.grad_stride % 2 == 0 && gaddr % 8 == 0 && self_ok2;
  // the long runs first, on their own stream: they are listed, then folded while step_short_kernel does the rest — unless the
  // side is detached (nothing expected there, the caller's stream does not wait for it): then the tile kernel is enqueued first
  auto long_side = [&]() -> int {
    if (p.long_list != nullptr && p.fold_tree) {
      launch_tree<IdxT, OPT, float>(p, stream, lstream);
    } else if (p.long_list != nullptr) {
      launch_mark_long_runs<IdxT>(p, stream, lstream);
      const int slices = static_cast<int>((p.a.dim + kSliceCols - 1) / kSliceCols);
      const bool long4 = p.a.dim % kSliceCols == 0 && p.a.grad_stride % 4 == 0 && gaddr % 16 == 0 && self_ok4;
      const bool rows4 = p.a.dim % 4 == 0 && p.a.grad_stride % 4 == 0 && gaddr % 16 == 0 && self_ok4 &&
                         p.a.dim <= 65535 * slice4_cols<float>();
      const bool old_long = WM_AB_KNOB("WM_STEP_LONG_OLD") != nullptr;
      if (rows4 && !old_long) {
        static const bool lds_ok =
          hipFuncSetAttribute(reinterpret_cast<const void*>(&step_long4_kernel<IdxT, OPT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLong4LdsBytes)) == hipSuccess;
        if (!lds_ok) return -2;
        // 144 KiB of LDS = one workgroup per CU: launch one resident wave of workgroups (256 CUs) and let each walk the
        // run list — with more workgroups than CUs the hottest run may only START after several rounds of others
        const int slices4 = static_cast<int>((p.a.dim + slice4_cols<float>() - 1) / slice4_cols<float>());
        int gx            = std::max(1, 256 / slices4);
        if (const char* e = WM_AB_KNOB("WM_LONG_GRID")) gx = std::max(1, atoi(e));
        // (the first workgroups per slice fold the dense copies — 16 x slices4 of them take 4 runs of 128 columns in 8-column
        // slices —, the others walk the list as before; TOGETHER one resident round: with 16 workgroups per slice on top of
        // the 256 / slices4 the last ones only started when the tile kernel had filled the chip, and finished after it —
        // profiles/r06_grad_timeline_zipf_dense_second.txt)
        opt_params lp = p;
        lp.dense_x    = p.dense_jobs != nullptr ? 16 : 0;   // (dense_jobs is only set for dim <= 256: gx >= 32, hip_optimizer_step_dev)
        hipLaunchKernelGGL((step_long4_kernel<IdxT, OPT>), dim3(gx, slices4), dim3(kLongBlock), kLong4LdsBytes, lstream, lp);
      }
      else if (long4)
        hipLaunchKernelGGL((step_long_kernel<IdxT, OPT, true>), dim3(1024, slices), dim3(kBlock), 0, lstream, p);
      else
        hipLaunchKernelGGL((step_long_kernel<IdxT, OPT, false>), dim3(1024, slices), dim3(kBlock), 0, lstream, p);
    }
    return 0;
  };
  if (p.detached_side != 1 && long_side() != 0) return -2;
  // (a float4-per-lane variant, two runs per wave instruction, was measured too: no gain for SGD, 10-15 % slower for the
  // stateful optimizers through register pressure — 8 bytes per lane stay)
  const bool cached = p.a.cache_slot_of != nullptr;
  // rows of whole 16-byte pieces on every side: the tile kernel (WM_STEP_TILE=0 keeps the wave-per-run kernel)
  const bool tile_off = WM_AB_KNOB("WM_STEP_TILE") != nullptr && WM_AB_KNOB("WM_STEP_TILE")[0] == '0';
  const bool st_ok = p.a.per_element_state == nullptr ||
                     (p.a.per_element_stride % 4 == 0 && reinterpret_cast<uint64_t>(p.a.per_element_state) % 16 == 0 &&
                      (p.a.cache_state_data == nullptr || (p.a.cache_state_row_elems % 4 == 0 &&
                                                           reinterpret_cast<uint64_t>(p.a.cache_state_data) % 16 == 0)));
  const bool tb_ok = p.a.table_stride % 4 == 0 && reinterpret_cast<uint64_t>(p.a.local_table) % 16 == 0 &&
                     (!cached || (p.a.cache_row_elems % 4 == 0 && reinterpret_cast<uint64_t>(p.a.cache_data) % 16 == 0));
  const bool tile_ok = !tile_off && p.a.dim % 4 == 0 && p.a.dim >= 32 && p.a.grad_stride % 4 == 0 && gaddr % 16 == 0 &&
                       self_ok4 && st_ok && tb_ok;
  if (tile_ok) {
    const int vecs = static_cast<int>(p.a.dim / 4);
    opt_params tp = p;
    int tblocks   = 1;
    tile_launch_shape(p.a.count, vecs, OPT == WHOLEMEMORY_OPT_SGD ? WM_TILE_KU_SGD : WM_TILE_KU_STATE, &tp.tile_runs, &tblocks);
    // (Round 3 ran SGD on 512-byte rows through a copy of the kernel forced to 7 waves / SIMD, 10 values spilled to scratch:
    // its loads were serialised and occupancy was the only source of loads in flight. With the straight-line batches the
    // natural register budget — 84 VGPRs, 5 waves, no scratch — is the faster one: 2.92 vs 2.98 ms per 10 M rows,
    // profiles/r04_grad_apply_ab_occupancy_launch_shape.txt; the forced build is gone and with it the 1.12 x write traffic.)
#define WM_TILE(RPS)                                                                                                    \
  do {                                                                                                                  \
    if (cached)                                                                                                         \
      hipLaunchKernelGGL((step_tile_kernel<IdxT, OPT, RPS, true>), dim3(tblocks), dim3(kBlock), 0, stream, tp);          \
    else                                                                                                                \
      hipLaunchKernelGGL((step_tile_kernel<IdxT, OPT, RPS, false>), dim3(tblocks), dim3(kBlock), 0, stream, tp);         \
  } while (0)
    if (vecs > 32) WM_TILE(1);
    else if (vecs > 16) WM_TILE(2);
    else if (vecs > 8) WM_TILE(4);
    else WM_TILE(8);
#undef WM_TILE
    if (p.detached_side == 1 && long_side() != 0) return -2;
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  // rows of whole 8-byte pieces that are not whole 16-byte ones (dim = 2 mod 4: 602 floats, the Reddit feature width): the tile
  // kernel with 8-byte pieces (round 6) — one 512-byte wave step per row instead of the wave-per-run kernel below.
  // WM_STEP_TILE8=0 keeps the wave-per-run kernel (A/B).
  // rows of dim % 4 != 0 floats: the tile kernel on the first dim / 4 sixteen-byte pieces + a lane-per-run pass over the last
  // dim % 4 floats (RAGGED). Gradient rows may start anywhere (4-byte aligned). Whole calls, 10 M rows, probed placement
  // (profiles/r06_dim_sweep_ragged.txt): 513 floats 47.8 -> 62-64 % of peak, 301 floats 46 -> 55 %; 129 / 127 floats 36-44 % either
  // way (rows of ~512 B that start inside a line on every side); dim = 2 mod 4: 602 floats 60.0 -> 61.7 %, 130 floats 41.8 -> 43.7 %
  // against the 8-byte-piece kernel below, which WM_STEP_RAGGED=0 keeps (with the wave-per-run kernel for odd dims).
  const char* ragged_env = WM_KNOB("WM_STEP_RAGGED");
  const bool ragged_ok = !tile_off && !cached && p.a.dim >= 36 && p.a.dim % 4 != 0 && st_ok && tb_ok &&
                         !(ragged_env != nullptr && ragged_env[0] == '0');
  if (ragged_ok) {
    const int vecs = static_cast<int>(p.a.dim / 4);
    opt_params tp = p;
    int tblocks   = 1;
    tile_launch_shape(p.a.count, vecs, OPT == WHOLEMEMORY_OPT_SGD ? WM_TILE_KU_SGD : WM_TILE_KU_STATE, &tp.tile_runs, &tblocks);
#define WM_TILE_RAGGED(RPS) \
    hipLaunchKernelGGL((step_tile_kernel<IdxT, OPT, RPS, false, float, 0, 0, 16, true>), dim3(tblocks), dim3(kBlock), 0, stream, tp)
    if (vecs > 32) WM_TILE_RAGGED(1);
    else if (vecs > 16) WM_TILE_RAGGED(2);
    else if (vecs > 8) WM_TILE_RAGGED(4);
    else WM_TILE_RAGGED(8);
#undef WM_TILE_RAGGED
    if (p.detached_side == 1 && long_side() != 0) return -2;
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  const bool tile8_off = WM_AB_KNOB("WM_STEP_TILE8") != nullptr && WM_AB_KNOB("WM_STEP_TILE8")[0] == '0';
  const bool tile8_ok  = !tile_off && !tile8_off && !cached && vec2 && p.a.dim >= 66 && p.a.table_stride % 2 == 0 &&
                        reinterpret_cast<uint64_t>(p.a.local_table) % 8 == 0 &&
                        (p.a.per_element_state == nullptr ||
                         (p.a.per_element_stride % 2 == 0 && reinterpret_cast<uint64_t>(p.a.per_element_state) % 8 == 0));
  if (tile8_ok) {
    opt_params tp = p;
    int tblocks   = 1;
    constexpr int kKu = OPT == WHOLEMEMORY_OPT_SGD ? WM_TILE_KU_SGD : WM_TILE_KU_STATE;
    tile_launch_shape(p.a.count, static_cast<int>(p.a.dim / 2), kKu, &tp.tile_runs, &tblocks);   // (> 32 pieces: one row per wave step)
    hipLaunchKernelGGL((step_tile_kernel<IdxT, OPT, 1, false, float, 0, 0, 8>), dim3(tblocks), dim3(kBlock), 0, stream, tp);
    if (p.detached_side == 1 && long_side() != 0) return -2;
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  if (vec2 && !cached)
    hipLaunchKernelGGL((step_short_kernel<IdxT, OPT, 2, float, false>), dim3(blocks), dim3(kBlock), 0, stream, p);
  else if (vec2)
    hipLaunchKernelGGL((step_short_kernel<IdxT, OPT, 2, float, true>), dim3(blocks), dim3(kBlock), 0, stream, p);
  else if (!cached)
    hipLaunchKernelGGL((step_short_kernel<IdxT, OPT, 1, float, false>), dim3(blocks), dim3(kBlock), 0, stream, p);
  else
    hipLaunchKernelGGL((step_short_kernel<IdxT, OPT, 1, float, true>), dim3(blocks), dim3(kBlock), 0, stream, p);
  if (p.detached_side == 1 && long_side() != 0) return -2;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// HALF / BF16 tables with gradients of the same dtype: SGD only (see wm_optimizer_args::value_dtype)
template <typename IdxT, typename T>
int launch_step_sgd16(opt_params p, int blocks, hipStream_t stream, hipStream_t lstream)
{
  constexpr int kOpt   = WHOLEMEMORY_OPT_SGD;
  constexpr int kS     = slice4_cols<T>();  // columns per long-run slice
  const uint64_t gaddr = reinterpret_cast<uint64_t>(p.a.grads) | reinterpret_cast<uint64_t>(p.a.self_grads);
  const int64_t sstr   = p.a.self_grads == nullptr ? 0 : p.a.self_grad_stride;
  const bool vec4      = p.a.dim % 4 == 0 && p.a.grad_stride % 4 == 0 && sstr % 4 == 0 && p.a.table_stride % 4 == 0 &&
                    gaddr % 8 == 0 && reinterpret_cast<uint64_t>(p.a.local_table) % 8 == 0;
  const bool rows16    = p.a.dim % 8 == 0 && p.a.grad_stride % 8 == 0 && sstr % 8 == 0 && gaddr % 16 == 0 &&
                      p.a.dim <= 65535 * kS;
  if (!rows16) p.long_list = nullptr;  // no LDS-DMA path for this shape: the wave-per-run kernel folds every run itself
  auto long_side = [&]() -> int {   // (first, or behind the tile kernel when detached: launch_step_opt)
    if (p.long_list != nullptr && p.fold_tree) {
      launch_tree<IdxT, kOpt, T>(p, stream, lstream);
    } else if (p.long_list != nullptr) {
      static const bool lds_ok =
        hipFuncSetAttribute(reinterpret_cast<const void*>(&step_long4_kernel<IdxT, kOpt, T>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLong4LdsBytes)) == hipSuccess;
      if (!lds_ok) return -2;
      launch_mark_long_runs<IdxT>(p, stream, lstream);
      const int slices = static_cast<int>((p.a.dim + kS - 1) / kS);
      const int gx     = std::max(1, 256 / slices);
      hipLaunchKernelGGL((step_long4_kernel<IdxT, kOpt, T>), dim3(gx, slices), dim3(kLongBlock), kLong4LdsBytes, lstream, p);
    }
    return 0;
  };
  if (p.detached_side != 1 && long_side() != 0) return -2;
  const bool cached = p.a.cache_slot_of != nullptr;
  // rows of whole 16-byte pieces (8 elements) on every side: the tile kernel, as for fp32 tables
  const bool tile_off = WM_AB_KNOB("WM_STEP_TILE") != nullptr && WM_AB_KNOB("WM_STEP_TILE")[0] == '0';
  const bool tile_ok = !tile_off && rows16 && p.a.dim >= 64 && p.a.table_stride % 8 == 0 &&
                       reinterpret_cast<uint64_t>(p.a.local_table) % 16 == 0 &&
                       (!cached || (p.a.cache_row_elems % 8 == 0 && reinterpret_cast<uint64_t>(p.a.cache_data) % 16 == 0));
  if (tile_ok) {
    const int vecs = static_cast<int>(p.a.dim / 8);
    int tblocks    = 1;
    tile_launch_shape(p.a.count, vecs, WM_TILE_KU_STATE, &p.tile_runs, &tblocks);
#define WM_TILE16(RPS)                                                                                                       \
  do {                                                                                                                       \
    if (cached)                                                                                                              \
      hipLaunchKernelGGL((step_tile_kernel<IdxT, kOpt, RPS, true, T>), dim3(tblocks), dim3(kBlock), 0, stream, p);           \
    else                                                                                                                     \
      hipLaunchKernelGGL((step_tile_kernel<IdxT, kOpt, RPS, false, T>), dim3(tblocks), dim3(kBlock), 0, stream, p);          \
  } while (0)
    if (vecs > 32) WM_TILE16(1);
    else if (vecs > 16) WM_TILE16(2);
    else if (vecs > 8) WM_TILE16(4);
    else WM_TILE16(8);
#undef WM_TILE16
    if (p.detached_side == 1 && long_side() != 0) return -2;
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  if (vec4 && !cached)
    hipLaunchKernelGGL((step_short_kernel<IdxT, kOpt, 4, T, false>), dim3(blocks), dim3(kBlock), 0, stream, p);
  else if (vec4)
    hipLaunchKernelGGL((step_short_kernel<IdxT, kOpt, 4, T, true>), dim3(blocks), dim3(kBlock), 0, stream, p);
  else if (!cached)
    hipLaunchKernelGGL((step_short_kernel<IdxT, kOpt, 1, T, false>), dim3(blocks), dim3(kBlock), 0, stream, p);
  else
    hipLaunchKernelGGL((step_short_kernel<IdxT, kOpt, 1, T, true>), dim3(blocks), dim3(kBlock), 0, stream, p);
  if (p.detached_side == 1 && long_side() != 0) return -2;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <typename IdxT>
int launch_step(const opt_params& p, int blocks, hipStream_t stream, hipStream_t lstream)
{
  if (p.a.value_dtype == WHOLEMEMORY_DT_HALF || p.a.value_dtype == WHOLEMEMORY_DT_BF16) {
    if (p.a.type != WHOLEMEMORY_OPT_SGD) return -1;
    return p.a.value_dtype == WHOLEMEMORY_DT_HALF ? launch_step_sgd16<IdxT, half_t>(p, blocks, stream, lstream)
                                                  : launch_step_sgd16<IdxT, bf16_t>(p, blocks, stream, lstream);
  }
  if (p.a.value_dtype != WHOLEMEMORY_DT_FLOAT && p.a.value_dtype != WHOLEMEMORY_DT_UNKNOWN) return -1;
  switch (p.a.type) {
    case WHOLEMEMORY_OPT_SGD: return launch_step_opt<IdxT, WHOLEMEMORY_OPT_SGD>(p, blocks, stream, lstream);
    case WHOLEMEMORY_OPT_LAZY_ADAM: return launch_step_opt<IdxT, WHOLEMEMORY_OPT_LAZY_ADAM>(p, blocks, stream, lstream);
    case WHOLEMEMORY_OPT_ADAGRAD: return launch_step_opt<IdxT, WHOLEMEMORY_OPT_ADAGRAD>(p, blocks, stream, lstream);
    case WHOLEMEMORY_OPT_RMSPROP: return launch_step_opt<IdxT, WHOLEMEMORY_OPT_RMSPROP>(p, blocks, stream, lstream);
    default: return -1;
  }
}

template <typename IdxT>
__global__ void round_robin_map_kernel(const IdxT* ids, IdxT* mapped, int64_t n, int64_t entry_start, int world,
                                       int rr, int64_t rank_rows)
{
  // reference functions/map_indices_func.cu:26-45. rank_rows == 0 is the reference statement (the caller's first row
  // plus the position inside a shard); rank_rows > 0 puts the OWNER's first row there (see backend.hpp)
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t idx = static_cast<int64_t>(ids[i]);
  if (idx < 0) {  // "skip me" ids stay what they are
    mapped[i] = ids[i];
    return;
  }
  int64_t t    = idx / rr;
  int64_t off  = idx % rr;
  int64_t base = rank_rows > 0 ? (t % world) * rank_rows : entry_start;
  mapped[i]    = static_cast<IdxT>(base + static_cast<int64_t>(rr) * (t / world) + off);
}

__global__ void fill_float_kernel(float* p, float v, int64_t n)
{
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    p[i] = v;
}

}  // namespace

}  // namespace wm
extern "C" int64_t wholememory_ext_split_sorts(void) { return wm::g_split_sorts.load(std::memory_order_relaxed); }
extern "C" int64_t wholememory_ext_hot_split_sorts(void) { return wm::g_hot_split_sorts.load(std::memory_order_relaxed); }
// runs of the last finished ORDERED optimizer step on the current device that were folded through a dense transposed copy
// (kernels/long_dense.cuh); read after a synchronise. A counter for tests and benchmarks.
extern "C" int64_t wholememory_ext_dense_fold_last(void)
{
  auto& lane = wm::long_lane::get();
  return lane.listed != nullptr ? static_cast<int64_t>(lane.listed[1]) : 0;
}
namespace wm {

void hip_dedup_defer_join(int on)
{
  const char* e = WM_KNOB("WM_DEDUP_DEFER_JOIN");   // =0: the side stream is joined in front of the step again (A/B switch)
  g_defer_join  = on != 0 && !(e != nullptr && e[0] == '0');
}
// A device-side wait of an earlier sort gave up (sort_lane::host_err): say so ONCE, as an error. Non-blocking — what it sees is
// what has finished; callers that synchronise (the multi-rank gradient apply, WM_DEBUG_SYNC=1) ask again afterwards.
int hip_device_error()
{
  const uint32_t e = sort_lane::get().take_error();
  if (e == 0) return 0;
  WM_ERROR("a device-side wait of the gradient path's id sort timed out (code 0x%x:%s%s%s%s): the optimizer step of that call was "
           "NOT applied (its run count was zeroed on the device). A tool that runs one kernel at a time (counter collection, some "
           "debuggers) stalls these waits: set WM_DEVICE_WAITS=0 for event-only synchronisation.",
           e, (e & split::kErrLookBack) ? " bucket look-back" : "", (e & split::kErrJoin) ? " join of the generic sort" : "",
           (e & split::kErrWait) ? " side-stream wait" : "", (e & split::kErrOnesweep) ? " radix-pass look-back" : "");
  return static_cast<int>(e);
}
int hip_dedup_join(void* stream_v)
{
  if (!g_join_pending) return hip_device_error() != 0 ? -2 : 0;
  g_join_pending = false;
  if (hipStreamWaitEvent(static_cast<hipStream_t>(stream_v), sort_lane::get().joined, 0) != hipSuccess) return -2;
  return hip_device_error() != 0 ? -2 : 0;
}

size_t hip_dedup_workspace_bytes(int64_t n, wholememory_dtype_t index_dtype)
{
  if (n <= 0) return 256;
  // the bounded-key path carves the 32-bit layout whatever the index type, and the split sort its own (advisor, round 4:
  // the 64-bit layout alone can be the smaller one)
  size_t most = layout<uint32_t>(nullptr, n).total;
  if (index_dtype != WHOLEMEMORY_DT_INT) most = std::max(most, layout<uint64_t>(nullptr, n).total);
  // (only batches that can take the split sort: its layout has a fixed ~17 MB term — kMaxTiles x kMaxPitch counters — that a
  // mini-batch of a few hundred ids would otherwise ask the caller's allocator for on every call; run_dedup tests the same
  // two conditions before it carves that layout: advisor, round 5)
  if (n < (INT64_C(1) << 30) && n >= split_min()) most = std::max(most, split_carve(nullptr, n).total);
  return most;
}

int hip_dedup_ids(const void* ids, wholememory_dtype_t index_dtype, int64_t n, int64_t key_upper_bound, int64_t key_lower_bound,
                  void* unique_ids, int32_t* run_starts, int32_t* order, int64_t* n_unique_out, void* workspace,
                  void* stream_v)
{
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  g_last_split = last_split_record{};   // (set again below when this sort is a split sort)
  if (hip_dedup_join(stream_v) != 0) return -2;   // (a deferred join nobody collected: before this sort touches anything)
  if (n >= (1ll << 31)) return -1;  // reference casts the receive count to int (exchange_embeddings_nccl_func.cu:118)
  if (n == 0) return hipMemsetAsync(n_unique_out, 0, sizeof(int64_t), stream) == hipSuccess ? 0 : -2;
  if (index_dtype == WHOLEMEMORY_DT_INT)
    return run_dedup<int32_t>(ids, n, key_upper_bound, key_lower_bound, unique_ids, run_starts, order, n_unique_out, workspace, stream);
  if (index_dtype == WHOLEMEMORY_DT_INT64)
    return run_dedup<int64_t>(ids, n, key_upper_bound, key_lower_bound, unique_ids, run_starts, order, n_unique_out, workspace, stream);
  return -1;
}

// ---- ids in ascending row order for locality (HOST-table gather, backend.hpp: sort_ids) ----
namespace {
template <typename KeyT>
__global__ void expand_sorted_ids_kernel(const KeyT* ids, const int32_t* order, int64_t n, KeyT* sorted_ids, int64_t* raw)
{
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t o = order[i];
  sorted_ids[i]   = ids[o];   // the id as the caller wrote it (a negative one stays negative: the gather skips it)
  raw[i]          = o;
}

struct sort_ids_layout {
  uint32_t* keys;
  int32_t* order;
  void* temp;
  size_t temp_bytes, total;
};
sort_ids_layout sort_ids_carve(void* ws, int64_t n)
{
  auto align = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  size_t sort_bytes = 0;
  (void)sort_pairs32(nullptr, sort_bytes, static_cast<const uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
                     rocprim::counting_iterator<int32_t>(0), static_cast<int32_t*>(nullptr), static_cast<size_t>(n), 0, 32, nullptr);
  sort_ids_layout l;
  char* p  = static_cast<char*>(ws);
  size_t o = 0;
  l.keys   = reinterpret_cast<uint32_t*>(p + o), o += align(4 * static_cast<size_t>(n));
  l.order  = reinterpret_cast<int32_t*>(p + o), o += align(4 * static_cast<size_t>(n));
  l.temp   = p + o;
  l.temp_bytes = sort_bytes;
  l.total  = o + align(sort_bytes) + 256;
  return l;
}

template <typename KeyT>
int run_sort_ids(const void* ids, int64_t n, int64_t key_upper_bound, int low_bit, void* sorted_ids, int64_t* raw, void* ws,
                 hipStream_t stream)
{
  using UKey = typename std::make_unsigned<KeyT>::type;
  auto l     = sort_ids_carve(ws, n);
  size_t tb  = l.temp_bytes;
  // ids outside [0, key_upper_bound) — negative ones above all — read as the key `key_upper_bound` and land behind every row
  narrow_key_iterator<UKey> keys{static_cast<const UKey*>(ids), static_cast<UKey>(0), static_cast<uint32_t>(key_upper_bound)};
  const unsigned bits = significant_bits(key_upper_bound + 1, 32);
  const unsigned lo   = static_cast<unsigned>(std::max(0, std::min<int>(low_bit, static_cast<int>(bits) - 1)));
  if (sort_pairs32(l.temp, tb, keys, l.keys, rocprim::counting_iterator<int32_t>(0), l.order, static_cast<size_t>(n), lo, bits, stream) != hipSuccess)
    return -2;
  hipLaunchKernelGGL((expand_sorted_ids_kernel<KeyT>), dim3(static_cast<unsigned>((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     stream, static_cast<const KeyT*>(ids), l.order, n, static_cast<KeyT*>(sorted_ids), raw);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
}  // namespace

size_t hip_sort_ids_workspace_bytes(int64_t n) { return n <= 0 ? 256 : sort_ids_carve(nullptr, n).total; }

int hip_sort_ids(const void* ids, wholememory_dtype_t index_dtype, int64_t n, int64_t key_upper_bound, int low_bit,
                 void* sorted_ids, int64_t* raw, void* workspace, void* stream_v)
{
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (n <= 0) return 0;
  if (key_upper_bound <= 0 || key_upper_bound >= INT64_C(0xFFFFFFFF) || n >= (INT64_C(1) << 31)) return -3;
  if (index_dtype == WHOLEMEMORY_DT_INT) return run_sort_ids<int32_t>(ids, n, key_upper_bound, low_bit, sorted_ids, raw, workspace, stream);
  if (index_dtype == WHOLEMEMORY_DT_INT64) return run_sort_ids<int64_t>(ids, n, key_upper_bound, low_bit, sorted_ids, raw, workspace, stream);
  return -1;
}

// Dense copies of the very long runs of an ORDERED fp32 fold (kernels/long_dense.cuh): runs of at least dense_min_rows() rows,
// at most 3 / 8 of the batch's rows per call (what does not fit stays step_long4_kernel's), none for batches without room for
// one such run. WM_DENSE_FOLD=0 switches it off,
// WM_DENSE_FOLD_MIN=rows moves the threshold (tests).
inline int dense_min_rows(int64_t n_recv)
{
  if (const char* e = WM_KNOB("WM_DENSE_FOLD")) if (e[0] == '0') return 0;
  // A run needs the dense route only when its chain would outlast the tile kernel beside it, and every copied row delays that
  // kernel (the caller's stream waits for the copy): runs of at least 1 / 32 of the batch (131072 rows or more). Zipf(1.05),
  // 10 M ids: the 527 k-row run alone — the 255 k- and 166 k-row runs take 1.3 and 0.9 ms in step_long4_kernel's ring, well
  // inside the tile kernel's 2 ms (copying them too measured equal: 2.69 ms either way, profiles/r06_dense_fold_ab_v6_threshold.txt)
  int v = static_cast<int>(std::min<int64_t>(std::max<int64_t>(131072, n_recv / 32), INT64_C(1) << 30));
  if (const char* e = WM_KNOB("WM_DENSE_FOLD_MIN")) v = std::max(kLongRun + 1, atoi(e));
  return v;
}
struct dense_carve {
  size_t off_jobs, off_buf, total;
  int max_jobs;
  int64_t cap_floats;
};
inline dense_carve dense_layout(size_t base, int64_t n_recv, int64_t dim)
{
  dense_carve c{base, base, base, 0, 0};
  const int dmin = dense_min_rows(n_recv);
  const int64_t cap_rows = n_recv / 8 * 3;   // (3 / 8 of the batch: at 8 ranks the owner of a Zipf(1.05) batch's hottest id receives 8 x 527 k
                                             // copies among its 13.7 M rows — 31 %; what does not fit stays step_long4_kernel's)
  if (dmin <= 0 || dim % 4 != 0 || cap_rows < dmin) return c;
  auto align = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  c.max_jobs   = static_cast<int>(cap_rows / dmin + 2);
  c.cap_floats = (cap_rows + 4 * c.max_jobs) * dim;   // (every run rounds up to a whole group of four rows)
  c.off_jobs   = align(base);
  c.off_buf    = align(c.off_jobs + sizeof(dense_fold::job) * static_cast<size_t>(c.max_jobs));
  c.total      = c.off_buf + sizeof(float) * static_cast<size_t>(c.cap_floats);
  return c;
}
inline size_t ordered_list_bytes(int64_t n_recv) { return 32 + sizeof(long_run_entry) * static_cast<size_t>(n_recv / (kLongRun + 1) + 2); }

size_t hip_long_run_ws_bytes(int64_t n_recv, int64_t dim)
{   // the larger of the two layouts (the fold order is only resolved at the step: it may depend on the value dtype). The room for
    // dense copies (3 n / 8 rows: 1.9 GB for 10 M rows of 128 floats) only while the device's recent steps listed long runs at all
    // (long_lane::expect_long: a uniform series pays nothing for it after its first calls); the step takes the dense route only
    // when the workspace it is handed is that big (wm_optimizer_args::long_run_ws_bytes)
  bool expect;
  {
    std::lock_guard<std::mutex> lk(long_lane::get().mu);
    expect = long_lane::get().expect_long();
  }
  const size_t ordered = expect ? dense_layout(ordered_list_bytes(n_recv), n_recv, dim).total : ordered_list_bytes(n_recv);
  return std::max(ordered, tree_ws_bytes(n_recv, dim, std::min(tree_threshold(), kTreeMin)));
}

// a->count is an UPPER BOUND for the launch geometry; the true run count is read on the device from
// a->run_starts' companion scalar when `n_unique_dev` is non-null (ids == unique ids from hip_dedup_ids).
int hip_optimizer_step_dev(const wm_optimizer_args* a, const int64_t* n_unique_dev, void* stream_v)
{
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (a->count == 0) return 0;
  opt_params p{*a, n_unique_dev, nullptr, nullptr};
  p.long_threshold = kLongRun;
  if (a->long_run_ws != nullptr && a->dim <= 65535 * kSliceCols) {
    // [int32 counter | pad to 16 B | entries]; at most count / (kLongRun + 1) long runs can exist
    p.long_count = static_cast<int32_t*>(a->long_run_ws);
    p.long_list  = reinterpret_cast<long_run_entry*>(static_cast<char*>(a->long_run_ws) + 32);
    // tree fold: 16-byte pieces of the gradient rows on every side (else the ordered kernels take the call as before)
    const bool f32     = a->value_dtype != WHOLEMEMORY_DT_HALF && a->value_dtype != WHOLEMEMORY_DT_BF16;
    const int ve       = f32 ? 4 : 8;
    const uint64_t ga  = reinterpret_cast<uint64_t>(a->grads) | reinterpret_cast<uint64_t>(a->self_grads);
    const bool pieces  = a->dim % ve == 0 && a->grad_stride % ve == 0 && ga % 16 == 0 &&
                        (a->self_grads == nullptr || a->self_grad_stride % ve == 0);
    if (resolve_fold_mode(*a) == 1 && pieces && a->count < (INT64_C(1) << 31)) {
      p.fold_tree      = 1;
      p.long_threshold = tree_threshold();
      p.long_list      = reinterpret_cast<long_run_entry*>(static_cast<char*>(a->long_run_ws) + 64);   // (non-null marker; the tree kernels carve the workspace themselves)
    } else if (f32 && pieces && a->dim <= 256 && WM_AB_KNOB("WM_STEP_LONG_OLD") == nullptr) {
      // (rows of up to 256 columns: the long-run kernel's one resident round — 256 / (dim / 32) workgroups per 32-column slice —
      // then has 16 per slice to spare for the dense copies, launch_step_opt)
      const dense_carve dc = dense_layout(ordered_list_bytes(a->count), a->count, a->dim);
      if (dc.max_jobs > 0 && a->long_run_ws_bytes >= dc.total) {
        p.dense_jobs     = reinterpret_cast<dense_fold::job*>(static_cast<char*>(a->long_run_ws) + dc.off_jobs);
        p.dense_buf      = reinterpret_cast<float*>(static_cast<char*>(a->long_run_ws) + dc.off_buf);
        p.dense_cap      = dc.cap_floats;
        p.dense_min      = dense_min_rows(a->count);
        p.dense_max_jobs = dc.max_jobs;
      }
    }
  }
  // runs that a split sort of this thread has just written: its control words hold the long-run counters (already zero) and
  // say whether a long run can exist at all (last_split_record)
  if (g_last_split.run_starts == a->run_starts && g_last_split.unique_ids == a->ids && g_last_split.n_unique == n_unique_dev &&
      n_unique_dev != nullptr && g_last_split.ctl != nullptr && p.long_list != nullptr &&
      WM_AB_KNOB("WM_STEP_OWN_COUNTERS") == nullptr) {
    p.split_ctl  = g_last_split.ctl;
    p.long_count = reinterpret_cast<int32_t*>(g_last_split.ctl + split::kCtlLongCounters);
  }
  g_last_split = last_split_record{};
  // the long-run side goes to its own stream (WM_STEP_SERIAL=1: everything on the caller's stream, for measurements)
  hipStream_t lstream = stream;
  const char* serial_env = WM_KNOB("WM_STEP_SERIAL");
  std::unique_lock<std::mutex> lane_lock;
  if (p.long_list != nullptr) lane_lock = std::unique_lock<std::mutex>(long_lane::get().mu);
  const bool by_guess = serial_env == nullptr && p.long_list != nullptr && !long_lane::get().expect_long();
  const bool serial   = serial_env != nullptr ? serial_env[0] != '0' : by_guess;
  // No long run expected AND the runs come from a split sort whose side stream the caller joins behind this step (deferred
  // join): listing + long-run kernels + the count's copy — three launches that find nothing to do, 17 us in line — go to that
  // side stream behind a wave that waits for the sort's "runs are final" word; the caller's stream gets the tile kernel and
  // nothing else. A wrong guess costs time once, as before: the fold then starts beside a tile kernel that already fills the chip.
  std::unique_lock<std::mutex> sort_lock;
  bool detached = false;
  if (by_guess && p.split_ctl != nullptr && g_join_pending && sort_lane::get().ok &&
      !(WM_KNOB("WM_STEP_DETACH") != nullptr && WM_KNOB("WM_STEP_DETACH")[0] == '0')) {
    sort_lock = std::unique_lock<std::mutex>(sort_lane::get().mu);
    lstream   = sort_lane::get().side();
    detached  = true;
    // (launch_mark_long_runs / launch_tree queue the waiting wave in front of their first kernel; 1: the side is enqueued
    // behind the tile kernel, 2 = WM_SIDE_FIRST=1: in front of it, the order of the first version, for A/B runs)
    p.detached_side = WM_AB_KNOB("WM_SIDE_FIRST") != nullptr && WM_AB_KNOB("WM_SIDE_FIRST")[0] == '1' ? 2 : 1;
  }
  if (p.long_list != nullptr && !serial && long_lane::get().fork(stream)) lstream = long_lane::get().stream;
  // (the counters are cleared on the side stream: only the long-run kernels read them)
  if (p.long_list != nullptr && p.split_ctl == nullptr &&
      hipMemsetAsync(p.long_count, 0, p.fold_tree ? 64 : 32, lstream) != hipSuccess)
    return -2;
  int64_t waves = a->count;
  int max_blocks = 256 * 8;
  if (const char* e = WM_AB_KNOB("WM_STEP_BLOCKS")) max_blocks = std::max(1, atoi(e));
  int blocks = static_cast<int>(std::min<int64_t>((waves + 3) / 4, max_blocks));
  if (blocks < 1) blocks = 1;
  int rc = -1;
  if (a->index_dtype == WHOLEMEMORY_DT_INT) rc = launch_step<int32_t>(p, blocks, stream, lstream);
  if (a->index_dtype == WHOLEMEMORY_DT_INT64) rc = launch_step<int64_t>(p, blocks, stream, lstream);
  if (p.long_list != nullptr) long_lane::get().report(p.long_count, lstream);
  if (detached) {   // (the caller's deferred join now waits for the long-run side as well)
    if (hipEventRecord(sort_lane::get().joined, lstream) != hipSuccess) return -2;
  } else if (lstream != stream && !long_lane::get().join(stream)) {
    return -2;  // the caller's stream continues after both sides
  }
  return rc;
}

template <typename IdxT>
__global__ void run_inverse_kernel(const int32_t* run_starts, const int32_t* order, const IdxT* unique_ids,
                                   const int64_t* n_unique, int64_t n, int64_t id_limit, int64_t* inverse)
{
  // the run of sorted position j: last u with run_starts[u] <= j. The 256 positions of a workgroup are consecutive, so
  // their runs lie between the run of the first and the run of the last one: two full-range searches per workgroup, then
  // every thread searches a window of at most 256 runs (8 steps instead of 23 on 5 M runs: 369 -> ~120 us per 10 M ids)
  __shared__ int64_t s_lo, s_hi;
  const int64_t j0 = static_cast<int64_t>(blockIdx.x) * blockDim.x;
  const int64_t j  = j0 + threadIdx.x;
  auto search      = [&](int64_t pos, int64_t lo, int64_t hi) {
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (run_starts[mid] <= pos) lo = mid; else hi = mid - 1;
    }
    return lo;
  };
  if (threadIdx.x == 0) s_lo = search(j0, 0, *n_unique - 1);
  if (threadIdx.x == 64) s_hi = search(min(j0 + static_cast<int64_t>(blockDim.x) - 1, n - 1), 0, *n_unique - 1);
  __syncthreads();
  if (j >= n) return;
  const int64_t u   = search(j, s_lo, s_hi);
  const int64_t id  = static_cast<int64_t>(unique_ids[u]);
  inverse[order[j]] = (id < 0 || (id_limit > 0 && id >= id_limit)) ? -1 : u;
}

int hip_run_inverse(const int32_t* run_starts, const int32_t* order, const void* unique_ids, wholememory_dtype_t index_dtype,
                    const int64_t* n_unique_dev, int64_t n, int64_t id_limit, int64_t* inverse, void* stream)
{
  if (n == 0) return 0;
  const int blocks = static_cast<int>((n + kBlock - 1) / kBlock);
  if (index_dtype == WHOLEMEMORY_DT_INT)
    hipLaunchKernelGGL((run_inverse_kernel<int32_t>), dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       run_starts, order, static_cast<const int32_t*>(unique_ids), n_unique_dev, n, id_limit, inverse);
  else if (index_dtype == WHOLEMEMORY_DT_INT64)
    hipLaunchKernelGGL((run_inverse_kernel<int64_t>), dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       run_starts, order, static_cast<const int64_t*>(unique_ids), n_unique_dev, n, id_limit, inverse);
  else
    return -1;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

__global__ void remap_self_order_kernel(int32_t* order, int64_t n, int64_t self_begin, int64_t self_count,
                                        const int64_t* self_rows)
{
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t pos = order[i] - self_begin;
  if (pos >= 0 && pos < self_count) order[i] = static_cast<int32_t>(-(self_rows[pos] + 1));
}

int hip_remap_self_order(int32_t* order, int64_t n, int64_t self_begin, int64_t self_count, const int64_t* self_rows,
                         void* stream)
{
  if (n == 0 || self_count == 0) return 0;
  const int blocks = static_cast<int>((n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(remap_self_order_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), order, n,
                     self_begin, self_count, self_rows);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int hip_round_robin_map(const void* ids, void* mapped, wholememory_dtype_t index_dtype, int64_t n, int64_t entry_start,
                        int world_size, int round_robin_size, int64_t rank_rows, void* stream_v)
{
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (n == 0) return 0;
  int blocks = static_cast<int>((n + kBlock - 1) / kBlock);
  if (index_dtype == WHOLEMEMORY_DT_INT)
    hipLaunchKernelGGL((round_robin_map_kernel<int32_t>), dim3(blocks), dim3(kBlock), 0, stream,
                       static_cast<const int32_t*>(ids), static_cast<int32_t*>(mapped), n, entry_start, world_size,
                       round_robin_size, rank_rows);
  else if (index_dtype == WHOLEMEMORY_DT_INT64)
    hipLaunchKernelGGL((round_robin_map_kernel<int64_t>), dim3(blocks), dim3(kBlock), 0, stream,
                       static_cast<const int64_t*>(ids), static_cast<int64_t*>(mapped), n, entry_start, world_size,
                       round_robin_size, rank_rows);
  else
    return -1;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

namespace {
template <typename IdxT>
__global__ void fill_iota_kernel(IdxT* p, int64_t n, int64_t first)
{
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    p[i] = static_cast<IdxT>(first + i);
}
// one wave per run of more than one id: is its row of partial sums finite?
__global__ __launch_bounds__(256) void partials_nonfinite_kernel(const int32_t* run_starts, const int64_t* n_unique, const half_t* rows,
                                                                int64_t dim, int64_t stride, int64_t* flag)
{
  const int lane     = threadIdx.x & 63;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t step = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  const int64_t nu   = *n_unique;
  for (int64_t u0 = wave * 64; u0 < nu; u0 += step * 64) {
    const int64_t u  = u0 + lane;
    const bool multi = u < nu && run_starts[u + 1] - run_starts[u] > 1;
    uint64_t todo    = __ballot(multi);
    while (todo != 0) {
      const int l = __ffsll(static_cast<long long>(todo)) - 1;
      todo &= todo - 1;
      const half_t* row = rows + (u0 + l) * stride;
      bool bad          = false;
      for (int64_t d = lane; d < dim; d += 64) {
        const float v = load_wide<half_t>(row[d]);
        bad           = bad || !(fabsf(v) <= 65504.f);   // inf or NaN
      }
      if (__ballot(bad) != 0 && lane == 0) *flag = -1;
    }
  }
}
}  // namespace

int hip_fill_iota(void* p, wholememory_dtype_t index_dtype, int64_t n, int64_t first, void* stream_v)
{
  if (n <= 0) return 0;
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  const int blocks   = static_cast<int>(std::min<int64_t>((n + 255) / 256, 4096));
  if (index_dtype == WHOLEMEMORY_DT_INT)
    hipLaunchKernelGGL(fill_iota_kernel<int32_t>, dim3(blocks), dim3(256), 0, stream, static_cast<int32_t*>(p), n, first);
  else if (index_dtype == WHOLEMEMORY_DT_INT64)
    hipLaunchKernelGGL(fill_iota_kernel<int64_t>, dim3(blocks), dim3(256), 0, stream, static_cast<int64_t*>(p), n, first);
  else
    return -1;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int hip_partials_nonfinite(const int32_t* run_starts, const int64_t* n_unique_dev, int64_t n_upper, const void* rows, int64_t dim,
                           int64_t stride, int64_t* flag_dev, void* stream_v)
{
  if (n_upper <= 0) return 0;
  const int blocks = static_cast<int>(std::min<int64_t>((n_upper + 255) / 256, 2048));   // 4 waves x 64 runs per block and trip
  hipLaunchKernelGGL(partials_nonfinite_kernel, dim3(std::max(blocks, 1)), dim3(256), 0, static_cast<hipStream_t>(stream_v), run_starts,
                     n_unique_dev, static_cast<const half_t*>(rows), dim, stride, flag_dev);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int hip_fill_float(float* p, float value, int64_t count, void* stream_v)
{
  if (count == 0) return 0;
  int blocks = static_cast<int>(std::min<int64_t>((count + kBlock - 1) / kBlock, 256 * 16));
  hipLaunchKernelGGL(fill_float_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream_v), p, value,
                     count);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace wm
