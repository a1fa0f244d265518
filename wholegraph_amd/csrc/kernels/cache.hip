// wholegraph_amd — device row cache for embeddings (gfx950 HIP).
//
// What the reference does (cpp/src/wholememory/embedding_cache.{hpp,cpp}, wholememory_ops/functions/
// embedding_cache_func.cuh, gather_cached_func.cu): a 32-way set-associative cache of embedding rows in device memory
// (16-bit tags, 14-bit scaled LFU counters packed per 32-thread warp), updated by a block per cache set after a sort of
// the batch, probed by a warp per lookup (32 tags loaded, ballot, compare).
// MI355X design — same observable behaviour (the cache is transparent: gathers and write-backs return exactly what
// the uncached table would), different structure, sized for 288 GB of HBM:
//   * a DIRECT MAP row -> slot (int32 per covered row) instead of tag probing: a lookup costs one 4-byte read before
//     the row read, not a 64-tag scan; 4 bytes per covered row is 0.8 % of a 512 B-row table;
//   * exact int32 access counters per covered row (the reference scales 14-bit counters);
//   * 64-slot sets, one WAVE per set for replacement: lane l holds slot l's resident row and its counter, the victim
//     is a wave-wide min, a row is moved by the 64 lanes together. Set s covers the contiguous row range
//     [s * set_cover, (s + 1) * set_cover) so that the sorted unique ids of a batch are already grouped by set.
// Replacement policy (LFU): after the batch's accesses are added to the counters, every missing row of the batch
// whose counter exceeds the smallest counter resident in its set replaces that resident (empty slots first);
// evicted modified rows are written back to the raw table first.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "../backend.hpp"

namespace wm {
namespace {

constexpr int kBlock = 256;

struct raw_view {
  char* base;
  char* const* rank_ptrs;
  const size_t* rank_offsets;
  size_t chunk_stride;
  int world_size;
  int same_chunk;
  int64_t row_stride_bytes;   // of the raw table
  int64_t row_offset_bytes;   // storage offset of the raw table view
};

// the companion table (optimizer states): flat per-rank base addressed by GLOBAL row
struct raw2_view {
  char* base;
  int64_t row_stride_bytes;
};
inline raw2_view make_raw2(const wm_cache_args& c)
{
  return raw2_view{c.data2 != nullptr ? static_cast<char*>(c.raw2_gref.pointer) : nullptr, c.raw2_row_stride_bytes};
}

inline raw_view make_raw(const wm_cache_args& c)
{
  raw_view v{};
  v.chunk_stride = c.raw_gref.stride;
  v.world_size   = c.raw_gref.world_size;
  v.same_chunk   = c.raw_gref.same_chunk ? 1 : 0;
  if (c.raw_gref.stride == 0) {
    v.base = static_cast<char*>(c.raw_gref.pointer);
  } else {
    v.rank_ptrs    = static_cast<char* const*>(c.raw_gref.pointer);
    v.rank_offsets = c.raw_gref.rank_memory_offsets;
  }
  v.row_stride_bytes = c.raw_row_stride_bytes;
  v.row_offset_bytes = c.raw_row_offset_bytes;
  return v;
}

// address of GLOBAL row `row` in the raw table
__device__ __forceinline__ char* raw_row(const raw_view& v, int64_t row)
{
  const size_t off = static_cast<size_t>(v.row_offset_bytes) + static_cast<size_t>(row) * static_cast<size_t>(v.row_stride_bytes);
  if (v.chunk_stride == 0) return v.base + off;
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
  return v.rank_ptrs[rank] + (off - start);
}

// a row moved by the whole wave (row_bytes is a multiple of 16: embedding rows are padded to 16 bytes)
__device__ __forceinline__ void wave_copy_row(char* dst, const char* src, int row_bytes, int lane)
{
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  for (int o = lane * 16; o < row_bytes; o += 64 * 16) *reinterpret_cast<u32x4*>(dst + o) = *reinterpret_cast<const u32x4*>(src + o);
}

struct cache_dev {
  int32_t* slot_of;
  int32_t* count;
  int64_t* row_of;
  uint8_t* dirty;
  char* data;
  int64_t cover_start, cover_rows, n_sets, set_cover;
  int row_bytes;
  char* data2;
  int row_bytes2;
};

inline cache_dev make_dev(const wm_cache_args& c)
{
  return cache_dev{c.slot_of, c.count, c.row_of, c.dirty, c.data, c.cover_start, c.cover_rows, c.n_sets, c.set_cover,
                   static_cast<int>(c.row_bytes), c.data2, static_cast<int>(c.row_bytes2)};
}

// counters: count[row] += multiplicity of the row in the batch (unique sorted rows + run starts from dedup_ids)
template <typename IdxT>
__global__ void cache_count_kernel(cache_dev c, const IdxT* unique_rows, const int32_t* run_starts, const int64_t* n_unique)
{
  const int64_t u = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= *n_unique) return;
  const int64_t r = static_cast<int64_t>(unique_rows[u]) - c.cover_start;
  if (r < 0 || r >= c.cover_rows) return;  // negative ("skip") ids and rows outside this cache
  c.count[r] += run_starts[u + 1] - run_starts[u];
}

// one wave per set: bring in the batch's missing rows that beat the set's least frequently used residents
// fill list of the PLAN mode (raw table not addressable from this rank, e.g. DISTRIBUTED): the kernel only decides —
// which row goes into which slot — and the host fetches the rows through the exchange and installs them afterwards
struct fill_list {
  int64_t* rows;   // GLOBAL rows to fetch
  int64_t* slots;  // the cache line each of them goes to
  int* count;      // device counter
};

template <typename IdxT>
__global__ __launch_bounds__(kBlock) void cache_update_kernel(cache_dev c, raw_view raw, raw2_view raw2,
                                                              const IdxT* unique_rows, const int64_t* n_unique_p,
                                                              fill_list fill)
{
  const int lane    = threadIdx.x & 63;
  const int64_t set = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) >> 6;
  if (set >= c.n_sets) return;
  const int64_t nu  = *n_unique_p;
  const int64_t lo_row = c.cover_start + set * c.set_cover;
  const int64_t hi_row = min(c.cover_start + c.cover_rows, lo_row + c.set_cover);
  // [first, last) = the batch's unique rows that fall into this set (unique_rows is sorted as UNSIGNED keys: negative
  // ids sit at the end, beyond every valid row)
  auto lower = [&](int64_t key) {
    int64_t a = 0, b = nu;
    while (a < b) {
      const int64_t m = (a + b) >> 1;
      const IdxT v    = unique_rows[m];
      if (v >= 0 && static_cast<int64_t>(v) < key)
        a = m + 1;
      else
        b = m;
    }
    return a;
  };
  const int64_t first = lower(lo_row), last = lower(hi_row);
  if (first >= last) return;
  const int64_t slot  = set * 64 + lane;
  int64_t my_row      = c.row_of[slot];                        // covered-row index or -1
  int32_t my_cnt      = my_row >= 0 ? c.count[my_row] : -1;    // empty slots lose against everything
  for (int64_t j = first; j < last; j++) {
    const int64_t g = static_cast<int64_t>(unique_rows[j]);
    const int64_t r = g - c.cover_start;
    if (c.slot_of[r] >= 0) continue;                           // already resident (wave-uniform)
    const int32_t cand = c.count[r];
    // victim = smallest (counter, lane)
    uint64_t key = (static_cast<uint64_t>(static_cast<uint32_t>(my_cnt + 1)) << 8) | static_cast<uint32_t>(lane);
    for (int off = 32; off > 0; off >>= 1) {
      const uint64_t o = __shfl_xor(key, off, 64);
      key              = o < key ? o : key;
    }
    const int victim      = static_cast<int>(key & 0xff);
    const int32_t min_cnt = static_cast<int32_t>(key >> 8) - 1;
    if (cand <= min_cnt) continue;                             // not more frequently used than anything resident
    const int64_t vslot = set * 64 + victim;
    char* line          = c.data + vslot * c.row_bytes;
    const int64_t old   = __shfl(my_row, victim, 64);
    if (old >= 0) {
      if (fill.rows == nullptr && c.dirty[vslot]) {
        wave_copy_row(raw_row(raw, c.cover_start + old), line, c.row_bytes, lane);
        if (c.data2 != nullptr)
          wave_copy_row(raw2.base + (c.cover_start + old) * raw2.row_stride_bytes, c.data2 + vslot * c.row_bytes2, c.row_bytes2, lane);
      }
      if (lane == 0) c.slot_of[old] = -1;
    }
    if (fill.rows == nullptr) {
      wave_copy_row(line, raw_row(raw, g), c.row_bytes, lane);
      if (c.data2 != nullptr)
        wave_copy_row(c.data2 + vslot * c.row_bytes2, raw2.base + g * raw2.row_stride_bytes, c.row_bytes2, lane);
    } else if (lane == 0) {  // plan mode (read-only caches: nothing to write back): record, the host installs
      const int k   = atomicAdd(fill.count, 1);
      fill.rows[k]  = g;
      fill.slots[k] = vslot;
    }
    if (lane == 0) {
      c.slot_of[r]    = static_cast<int32_t>(vslot);
      c.row_of[vslot] = r;
      c.dirty[vslot]  = 0;
    }
    if (lane == victim) {
      my_row = r;
      my_cnt = cand;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // slot_of[] written by lane 0 is read by every lane next trip
  }
}

// ids -> (slot or -1, id or -1): the two index lists of the split lookup. Grid-stride with at most 1024 workgroups and ONE
// atomic per workgroup for the hit statistics: an atomicAdd per wave on the single counter serialised 15 k atomics per
// 1 M ids and made this 20 us kernel take 190 us.
template <typename IdxT>
__global__ __launch_bounds__(kBlock) void cache_split_kernel(cache_dev c, const IdxT* ids, int64_t n, int64_t* cache_idx,
                                                             IdxT* raw_idx, unsigned long long* hits)
{
  __shared__ unsigned int block_hits;
  if (threadIdx.x == 0) block_hits = 0;
  __syncthreads();
  unsigned int wave_hits = 0;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * kBlock; base < n; base += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t i = base + threadIdx.x;
    bool hit        = false;
    if (i < n) {
      const IdxT id   = ids[i];
      const int64_t r = static_cast<int64_t>(id) - c.cover_start;
      int32_t slot    = -1;
      if (id >= 0 && r >= 0 && r < c.cover_rows) slot = c.slot_of[r];
      hit          = slot >= 0;
      cache_idx[i] = slot;
      raw_idx[i]   = hit ? static_cast<IdxT>(-1) : id;
    }
    wave_hits += static_cast<unsigned int>(__popcll(__ballot(hit)));
  }
  if (hits != nullptr) {
    if ((threadIdx.x & 63) == 0 && wave_hits) atomicAdd(&block_hits, wave_hits);
    __syncthreads();
    if (threadIdx.x == 0 && block_hits) atomicAdd(hits, static_cast<unsigned long long>(block_hits));
  }
}

// write modified lines back to the raw table; with `drop` also empty the cache and clear the counters
__global__ __launch_bounds__(kBlock) void cache_writeback_kernel(cache_dev c, raw_view raw, raw2_view raw2, int drop)
{
  const int lane     = threadIdx.x & 63;
  const int64_t slot = (static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x) >> 6;
  if (slot >= c.n_sets * 64) return;
  const int64_t r = c.row_of[slot];
  if (r < 0) return;
  if (c.dirty[slot]) {
    wave_copy_row(raw_row(raw, c.cover_start + r), c.data + slot * c.row_bytes, c.row_bytes, lane);
    if (c.data2 != nullptr)
      wave_copy_row(raw2.base + (c.cover_start + r) * raw2.row_stride_bytes, c.data2 + slot * c.row_bytes2, c.row_bytes2, lane);
  }
  if (lane == 0) {
    c.dirty[slot] = 0;
    if (drop) {
      c.slot_of[r]   = -1;
      c.row_of[slot] = -1;
    }
  }
}

// occupied / modified cache lines: grid-stride, one pair of atomics per workgroup
__global__ __launch_bounds__(kBlock) void cache_info_kernel(cache_dev c, unsigned long long* out)
{
  __shared__ unsigned int block_occ, block_dirty;
  if (threadIdx.x == 0) block_occ = block_dirty = 0;
  __syncthreads();
  unsigned int occ_n = 0, dirty_n = 0;
  const int64_t slots = c.n_sets * 64;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * kBlock; base < slots; base += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t slot = base + threadIdx.x;
    const bool occ     = slot < slots && c.row_of[slot] >= 0;
    const bool dirt    = occ && c.dirty[slot];
    occ_n += static_cast<unsigned int>(__popcll(__ballot(occ)));
    dirty_n += static_cast<unsigned int>(__popcll(__ballot(dirt)));
  }
  if ((threadIdx.x & 63) == 0) {
    if (occ_n) atomicAdd(&block_occ, occ_n);
    if (dirty_n) atomicAdd(&block_dirty, dirty_n);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (block_occ) atomicAdd(out, static_cast<unsigned long long>(block_occ));
    if (block_dirty) atomicAdd(out + 1, static_cast<unsigned long long>(block_dirty));
  }
}

}  // namespace

int hip_cache_update(const wm_cache_args* c, const void* unique_rows, wholememory_dtype_t dt, const int32_t* run_starts,
                     const int64_t* n_unique_dev, int64_t n_upper, int64_t* fill_rows, int64_t* fill_slots, int* fill_count,
                     void* stream_v)
{
  const fill_list fill{fill_rows, fill_slots, fill_count};
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (n_upper == 0 || c->n_sets == 0) return 0;
  const cache_dev d = make_dev(*c);
  const raw_view rv = make_raw(*c);
  const int cb      = static_cast<int>((n_upper + kBlock - 1) / kBlock);
  const int ub      = static_cast<int>((c->n_sets * 64 + kBlock - 1) / kBlock);
  if (dt == WHOLEMEMORY_DT_INT) {
    hipLaunchKernelGGL((cache_count_kernel<int32_t>), dim3(cb), dim3(kBlock), 0, stream, d, static_cast<const int32_t*>(unique_rows), run_starts, n_unique_dev);
    hipLaunchKernelGGL((cache_update_kernel<int32_t>), dim3(ub), dim3(kBlock), 0, stream, d, rv, make_raw2(*c), static_cast<const int32_t*>(unique_rows), n_unique_dev, fill);
  } else if (dt == WHOLEMEMORY_DT_INT64) {
    hipLaunchKernelGGL((cache_count_kernel<int64_t>), dim3(cb), dim3(kBlock), 0, stream, d, static_cast<const int64_t*>(unique_rows), run_starts, n_unique_dev);
    hipLaunchKernelGGL((cache_update_kernel<int64_t>), dim3(ub), dim3(kBlock), 0, stream, d, rv, make_raw2(*c), static_cast<const int64_t*>(unique_rows), n_unique_dev, fill);
  } else {
    return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int hip_cache_split(const wm_cache_args* c, const void* ids, wholememory_dtype_t dt, int64_t n, int64_t* cache_idx, void* raw_idx,
                    unsigned long long* hits_dev, void* stream_v)
{
  if (n == 0) return 0;
  const cache_dev d = make_dev(*c);
  const int blocks  = static_cast<int>(std::min<int64_t>((n + kBlock - 1) / kBlock, 1024));
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (dt == WHOLEMEMORY_DT_INT)
    hipLaunchKernelGGL((cache_split_kernel<int32_t>), dim3(blocks), dim3(kBlock), 0, stream, d, static_cast<const int32_t*>(ids), n, cache_idx, static_cast<int32_t*>(raw_idx), hits_dev);
  else if (dt == WHOLEMEMORY_DT_INT64)
    hipLaunchKernelGGL((cache_split_kernel<int64_t>), dim3(blocks), dim3(kBlock), 0, stream, d, static_cast<const int64_t*>(ids), n, cache_idx, static_cast<int64_t*>(raw_idx), hits_dev);
  else
    return -1;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int hip_cache_writeback(const wm_cache_args* c, int drop, void* stream_v)
{
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (c->n_sets == 0) return 0;
  const int blocks = static_cast<int>((c->n_sets * 64 * 64 + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(cache_writeback_kernel, dim3(blocks), dim3(kBlock), 0, stream, make_dev(*c), make_raw(*c), make_raw2(*c), drop);
  if (drop && hipMemsetAsync(c->count, 0, sizeof(int32_t) * c->cover_rows, stream) != hipSuccess) return -2;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

int hip_cache_info(const wm_cache_args* c, unsigned long long* out2_dev, void* stream_v)
{
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (hipMemsetAsync(out2_dev, 0, 16, stream) != hipSuccess) return -2;
  if (c->n_sets == 0) return 0;
  const int blocks = static_cast<int>(std::min<int64_t>((c->n_sets * 64 + kBlock - 1) / kBlock, 2048));
  hipLaunchKernelGGL(cache_info_kernel, dim3(blocks), dim3(kBlock), 0, stream, make_dev(*c), out2_dev);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace wm
