// wholegraph_amd — stable least-significant-digit radix sort of (32-bit key, position) pairs in the "onesweep" form, hand-written
// for gfx950 (one read and one write of the pairs per digit; the tile prefixes are handed from workgroup to workgroup through a
// look-back chain instead of a device-wide scan between passes).
//
// Role in the product: the GENERIC path behind the split sort (split_sort.cuh) of the owner-side id sort — reference counterpart
// cub::DeviceRadixSort::SortPairs in exchange_embeddings_nccl_func.cu:93-117 (stable, ascending, payload = receive position).
// The split sort decides ON THE DEVICE that a batch does not suit it (a bucket too large for LDS: the hot ids of a Zipf batch,
// clustered ids); what sorts the batch then has to be enqueued beforehand and must cost next to nothing when it is not needed.
// rocPRIM's sort cannot be switched off by a device word, these kernels can: each one reads `gate` first and returns when the
// word is zero. (Round 3 measured this sort against the tuned library call on 10 M 27-bit keys: 275 us against 260 us,
// bit-exact on every case — profiles/r03_onesweep_ab.txt; as the always-on sort the library stays, run_dedup's other branch.)
//
// A pass, per workgroup (BLOCK threads, IPT keys per thread, TILE = BLOCK x IPT keys):
//   1 take a tile ticket (atomic counter: tiles are numbered in the order workgroups START, so every tile a workgroup
//     waits for below belongs to a workgroup that is already running — forward progress without co-residency assumptions);
//   2 load the tile wave-striped (wave w owns keys [w, w + 1) x 64 x IPT of the tile, item j of lane l is key j x 64 + l:
//     every load instruction is one contiguous 256-byte read, and (j, l) order is memory order);
//   3 rank each key among the keys of its wave with the same digit: the lanes holding the same digit find each other with
//     one ballot per digit bit, the lowest of them adds the group size to the wave's LDS counter of the digit and hands
//     the old value to the group — no LDS contention whatever the key distribution (a Zipf batch's hot id is one group);
//   4 per digit: prefix over the waves, tile count -> published as AGGREGATE, exclusive scan over the digits (tile-local
//     start of each digit), exclusive scan of the digit's global histogram (start of the digit in the output);
//   5 look back over the earlier tiles' words of the digit (AGGREGATE: add and go on, PREFIX: add and stop), publish PREFIX;
//   6 reorder through LDS (keys, then the payloads through the same buffer) so that each digit's keys of the tile leave as
//     one contiguous segment.
// Keys are u32, payloads u32 positions, n < 2^30 (the look-back word is 2 flag bits + 30 count bits).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

namespace wm {
namespace osw {

constexpr int kMaxPasses   = 4;
constexpr int kMaxRadix    = 10;
constexpr uint32_t kFlagAggregate = 1u << 30, kFlagPrefix = 2u << 30, kValueMask = (1u << 30) - 1;
constexpr int kHistBlock = 512, kHistGrid = 512;

struct plan {
  int passes, radix_bits, bins, tiles, tile;
  size_t ctrl_words;   // u32 words to zero before the histogram kernel: hist | tickets + error | look-back state
  size_t hist_off, ticket_off, state_off;   // in words
};

// digits of ceil(bits / passes) bits each, passes = ceil(bits / 10) but never fewer than what 9-bit digits of a <= 27-bit
// key need (9 bits x 3 measured best for the 27-bit ids of a 100 M - 125 M row shard)
inline plan make_plan(int64_t n, unsigned bits, int tile)
{
  plan p{};
  if (bits < 1) bits = 1;
  p.passes     = static_cast<int>((bits + kMaxRadix - 1) / kMaxRadix);
  p.radix_bits = static_cast<int>((bits + p.passes - 1) / p.passes);
  if (p.radix_bits < 4) p.radix_bits = 4;
  p.bins       = 1 << p.radix_bits;
  p.tile       = tile;
  p.tiles      = static_cast<int>((n + tile - 1) / tile);
  p.hist_off   = 0;
  p.ticket_off = static_cast<size_t>(p.passes) * p.bins;
  p.state_off  = p.ticket_off + 64;
  p.ctrl_words = p.state_off + static_cast<size_t>(p.passes) * p.tiles * p.bins;
  return p;
}

template <int BLOCK>
__device__ __forceinline__ uint32_t block_exclusive_sum_u32(uint32_t v, uint32_t* s_waves /* BLOCK / 64 + 1 words */)
{
  constexpr int WAVES = BLOCK / 64;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  __syncthreads();   // s_waves may still be read from a previous call
  if (lane == 63) s_waves[wv] = incl;
  __syncthreads();
  uint32_t before = 0;
#pragma unroll
  for (int w = 0; w < WAVES; w++)
    if (w < wv) before += s_waves[w];
  return before + incl - v;
}

// histograms of every digit in one read of the keys
template <typename KeyIt>
__global__ __launch_bounds__(kHistBlock) void hist_kernel(KeyIt keys, int64_t n, int passes, int rb, uint32_t* hist, const uint32_t* gate)
{
  if (gate != nullptr && *gate == 0u) return;
  __shared__ uint32_t s_hist[kMaxPasses << kMaxRadix];
  const int bins = 1 << rb, total = passes * bins;
  const uint32_t mask = static_cast<uint32_t>(bins - 1);
  for (int i = threadIdx.x; i < total; i += kHistBlock) s_hist[i] = 0;
  __syncthreads();
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kHistBlock;
  int64_t i            = static_cast<int64_t>(blockIdx.x) * kHistBlock + threadIdx.x;
  constexpr int U      = 8;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    uint32_t k[U];
#pragma unroll
    for (int u = 0; u < U; u++) k[u] = keys[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++)
      for (int p = 0; p < passes; p++) atomicAdd(&s_hist[p * bins + ((k[u] >> (p * rb)) & mask)], 1u);
  }
  for (; i < n; i += stride) {
    const uint32_t k = keys[i];
    for (int p = 0; p < passes; p++) atomicAdd(&s_hist[p * bins + ((k >> (p * rb)) & mask)], 1u);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < total; j += kHistBlock)
    if (s_hist[j] != 0) atomicAdd(&hist[j], s_hist[j]);
}

struct pass_args {
  const uint32_t* vals_in;   // unused by the first pass (payload = position)
  uint32_t* keys_out;
  uint32_t* vals_out;
  int64_t n;
  int pass, rb, tiles;
  const uint32_t* gate;      // device word: 0 = this sort is not needed, return at once (nullptr: always run)
  const uint32_t* hist;      // [passes][bins]
  uint32_t* ticket;          // [passes], then the error word at ticket[60]
  uint32_t look_back_polls;  // how long a digit's look-back polls before it reports (error word) instead of hanging
  uint32_t* state;           // [passes][tiles][bins]
};

template <int BLOCK, int IPT, bool FIRST, typename KeyIt>
__global__ __launch_bounds__(BLOCK) void pass_kernel(KeyIt keys_in, pass_args a)
{
  if (a.gate != nullptr && *a.gate == 0u) return;
  constexpr int WAVES = BLOCK / 64, TILE = BLOCK * IPT;
  extern __shared__ uint32_t s_mem[];
  const int bins       = 1 << a.rb;
  uint32_t* s_buf      = s_mem;                   // [TILE]
  uint32_t* s_cnt      = s_buf + TILE;            // [WAVES][bins]
  uint32_t* s_off      = s_cnt + WAVES * bins;    // [bins] global position of the digit's segment minus its tile-local start
  uint32_t* s_waves    = s_off + bins;            // [WAVES + 1] scan scratch, then the tile number
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t mask  = static_cast<uint32_t>(bins - 1);
  const int shift      = a.pass * a.rb;

  // a bounded grid: a workgroup takes tickets until the tiles are handed out (a gated-off launch of thousands of workgroups
  // that only read the gate still costs ~7 us of a stream's time; of a few hundred, ~3)
  for (;;) {
  __syncthreads();   // the previous tile's LDS reads are done
  if (threadIdx.x == 0) s_waves[WAVES] = atomicAdd(&a.ticket[a.pass], 1u);
  for (int b = lane; b < bins; b += 64) s_cnt[wv * bins + b] = 0;
  __syncthreads();
  const int tile       = static_cast<int>(s_waves[WAVES]);
  if (tile >= a.tiles) break;
  const int64_t base   = static_cast<int64_t>(tile) * TILE;
  const int valid      = static_cast<int>(a.n - base < TILE ? a.n - base : TILE);

  // 2 load
  uint32_t key[IPT], val[IPT], slot[IPT];
  const int64_t first = base + wv * (64 * IPT) + lane;
#pragma unroll
  for (int j = 0; j < IPT; j++) {
    const int64_t g = first + j * 64;
    key[j]          = g < a.n ? static_cast<uint32_t>(keys_in[g]) : 0xFFFFFFFFu;
    if (FIRST) val[j] = static_cast<uint32_t>(g);
    else val[j] = g < a.n ? a.vals_in[g] : 0u;
  }
  // 3 rank inside the wave
  const uint64_t lt = (1ull << lane) - 1ull;
  uint32_t* my_cnt  = s_cnt + wv * bins;
#pragma unroll
  for (int j = 0; j < IPT; j++) {
    const uint32_t d = (key[j] >> shift) & mask;
    uint64_t m       = ~0ull;
    for (int b = 0; b < a.rb; b++) {
      const bool bit     = (d >> b) & 1u;
      const uint64_t bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    const int before = __popcll(m & lt);
    uint32_t old     = 0;
    if (before == 0) old = atomicAdd(&my_cnt[d], static_cast<uint32_t>(__popcll(m)));
    old     = __shfl(old, __ffsll(static_cast<long long>(m)) - 1, 64);
    slot[j] = old + before;
  }
  __syncthreads();

  // 4 per digit: prefix over the waves, tile count, scans; 5 look-back. A thread owns BPT consecutive digits.
  static_assert(BLOCK * 4 >= (1 << kMaxRadix), "a thread owns at most 4 digits");
  const int bpt = (bins + BLOCK - 1) / BLOCK;
  const int b0  = threadIdx.x * bpt;
  uint32_t tile_count[4] = {0, 0, 0, 0}, ghist[4] = {0, 0, 0, 0};
  uint32_t mine = 0, gmine = 0;
  uint32_t* st  = a.state + (static_cast<size_t>(a.pass) * a.tiles + tile) * bins;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int b = b0 + i;
    if (i < bpt && b < bins) {
      uint32_t run = 0;
#pragma unroll
      for (int w = 0; w < WAVES; w++) {
        const uint32_t c  = s_cnt[w * bins + b];
        s_cnt[w * bins + b] = run;
        run += c;
      }
      tile_count[i] = run;
      ghist[i]      = a.hist[a.pass * bins + b];
      __hip_atomic_store(&st[b], (tile == 0 ? kFlagPrefix : kFlagAggregate) | run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      mine += run;
      gmine += ghist[i];
    }
  }
  uint32_t local_start = block_exclusive_sum_u32<BLOCK>(mine, s_waves);
  uint32_t global_base = block_exclusive_sum_u32<BLOCK>(gmine, s_waves);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int b = b0 + i;
    if (i < bpt && b < bins) {
      uint32_t excl = 0;
      if (tile > 0) {
        const uint32_t* prev = st;
        for (int t = tile - 1; t >= 0; t--) {
          prev -= bins;
          uint32_t v;
          unsigned spins = 0;
          while (((v = __hip_atomic_load(&prev[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 30) == 0u) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > a.look_back_polls) {   // never in a healthy run: report instead of hanging the device
              a.ticket[60] = 1u;
              v            = kFlagPrefix;
              break;
            }
          }
          excl += v & kValueMask;
          if ((v >> 30) == 2u) break;
        }
        __hip_atomic_store(&st[b], kFlagPrefix | (excl + tile_count[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      s_off[b] = global_base + excl - local_start;
      // fold the digit's tile-local start into the wave prefixes: slot = s_cnt[w][d] + rank in the wave
#pragma unroll
      for (int w = 0; w < WAVES; w++) s_cnt[w * bins + b] += local_start;
      local_start += tile_count[i];
      global_base += ghist[i];
    }
  }
  __syncthreads();

  // 6 reorder: keys through LDS, out as one segment per digit; then the payloads through the same buffer
#pragma unroll
  for (int j = 0; j < IPT; j++) {
    const uint32_t d = (key[j] >> shift) & mask;
    slot[j] += my_cnt[d];
    s_buf[slot[j]] = key[j];
  }
  __syncthreads();
  uint32_t pos[IPT];
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const int s      = k * BLOCK + threadIdx.x;
    const uint32_t x = s_buf[s];
    pos[k]           = s_off[(x >> shift) & mask] + static_cast<uint32_t>(s);
    if (s < valid) a.keys_out[pos[k]] = x;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < IPT; j++) s_buf[slot[j]] = val[j];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const int s = k * BLOCK + threadIdx.x;
    if (s < valid) a.vals_out[pos[k]] = s_buf[s];
  }
  }   // next ticket
}

template <int BLOCK, int IPT>
constexpr size_t pass_lds_bytes(int bins)
{
  return sizeof(uint32_t) * (static_cast<size_t>(BLOCK) * IPT + static_cast<size_t>(BLOCK / 64) * bins + bins + BLOCK / 64 + 1);
}

// control words (u32) that must read zero before the first kernel: histograms | tickets + error word | look-back state
template <int BLOCK, int IPT>
inline size_t ctrl_words(int64_t n, unsigned bits)
{
  return make_plan(n, bits, BLOCK * IPT).ctrl_words;
}
// most control words any key width needs for n keys (workspaces are sized before the width is known)
template <int BLOCK, int IPT>
inline size_t ctrl_words_bound(int64_t n)
{
  size_t most = 0;
  for (unsigned bits = 1; bits <= 32; bits++) most = std::max(most, make_plan(n, bits, BLOCK * IPT).ctrl_words);
  return most;
}

// sorts (keys[i], i) for i in [0, n) by the low `bits` bits of the key, stable; keys read through KeyIt (anything with
// operator[] returning a 32-bit key). tk / tv: a second (keys, payloads) pair of n words each; ctrl: ctrl_words() words that
// READ ZERO (the caller's business: one fill, or a kernel that runs before anyway). Enqueues 1 + passes kernels on `stream`,
// every one gated on `gate` (see the file comment); returns 0 or a negative error.
// the word (of ctrl) a pass sets when its look-back gave up: whoever consumes the sorted pairs must look at it (optim.hip
// folds it into the split sort's error word in the generic path's closing kernel)
template <int BLOCK, int IPT>
inline uint32_t* error_word(uint32_t* ctrl, int64_t n, unsigned bits)
{
  return ctrl + make_plan(n, bits, BLOCK * IPT).ticket_off + 60;
}

template <int BLOCK, int IPT, typename KeyIt>
int sort_pairs(KeyIt keys, uint32_t* keys_sorted, uint32_t* order, int64_t n, unsigned bits, uint32_t* tk, uint32_t* tv,
               uint32_t* ctrl, const uint32_t* gate, hipStream_t stream, uint32_t look_back_polls = 1u << 26)
{
  if (n <= 0) return 0;
  if (n >= (INT64_C(1) << 30)) return -1;
  const plan p = make_plan(n, bits, BLOCK * IPT);
  hipLaunchKernelGGL((hist_kernel<KeyIt>), dim3(kHistGrid), dim3(kHistBlock), 0, stream, keys, n, p.passes, p.radix_bits,
                     ctrl + p.hist_off, gate);
  const size_t lds = pass_lds_bytes<BLOCK, IPT>(p.bins);
  static bool attr_set = [] {
    const size_t most = pass_lds_bytes<BLOCK, IPT>(1 << kMaxRadix);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pass_kernel<BLOCK, IPT, true, KeyIt>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(most));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pass_kernel<BLOCK, IPT, false, const uint32_t*>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(most));
    return true;
  }();
  (void)attr_set;
  pass_args a{};
  a.n = n, a.rb = p.radix_bits, a.tiles = p.tiles;
  a.gate = gate;
  a.look_back_polls = look_back_polls;
  a.hist = ctrl + p.hist_off, a.ticket = ctrl + p.ticket_off, a.state = ctrl + p.state_off;
  const uint32_t* kin = nullptr;
  const uint32_t* vin = nullptr;
  // three workgroups per CU, looping over tickets (an ACTIVE pass over 10 M pairs: 79-91 us; one workgroup per tile, 1221 of
  // them: 99-121 us — the fourth workgroup per CU's worth of tiles runs as a ragged second round)
  const int grid      = p.tiles < 768 ? p.tiles : 768;
  for (int pass = 0; pass < p.passes; pass++) {
    const bool to_final = ((p.passes - 1 - pass) & 1) == 0;
    a.pass     = pass;
    a.vals_in  = vin;
    a.keys_out = to_final ? keys_sorted : tk;
    a.vals_out = to_final ? order : tv;
    if (pass == 0)
      hipLaunchKernelGGL((pass_kernel<BLOCK, IPT, true, KeyIt>), dim3(grid), dim3(BLOCK), lds, stream, keys, a);
    else
      hipLaunchKernelGGL((pass_kernel<BLOCK, IPT, false, const uint32_t*>), dim3(grid), dim3(BLOCK), lds, stream, kin, a);
    kin = a.keys_out;
    vin = a.vals_out;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace osw
}  // namespace wm
