// wholegraph_amd — the owner-side id sort of the gradient apply as a two-stage "split sort" (gfx950 HIP, wave64).
//
// What it computes (reference: exchange_embeddings_nccl_func.cu:76-174 — cub::DeviceRadixSort::SortPairs of the received ids
// with the receive position as payload, then unique_by_key): the stable ascending order of the ids, the start of every run of
// equal ids in that order, and the unique ids. A stable sort has exactly one answer, so any algorithm that is stable gives the
// reference's bits; this one exploits what the owner knows — keys are row numbers of ITS shard (< 2^27 for a 125 M-row shard),
// the payload is a position < 2^31 — instead of running a generic radix sort three times over the pairs:
//
//   stage 1 (three launches, no workgroup waits for another):
//     split_hist_kernel     per tile of the ids: a histogram over BUCKETS = the top key bits (<= 2048 buckets of 2^shift rows
//                           each — 4096 for the biggest batches — + one "drop" bucket for ids outside the owner's range), written as one row of a
//                           tiles x buckets matrix
//     split_scan_kernel     column-wise exclusive prefix of the matrix (a tile's start inside each bucket), bucket totals,
//                           the overflow verdict (a bucket that would not fit stage 2's LDS), and the zeroing of the control
//                           words of what follows
//     split_scatter_kernel  re-reads its tile, gives every id a slot in its bucket's segment of the tile from one LDS counter
//                           per bucket (arrival order: the segment is NOT in receive order, stage 2 orders equal ids by
//                           their position), brings the tile into bucket order in LDS and writes keys and positions bucket
//                           by bucket, each bucket's share of the tile as one contiguous segment. (Written straight from the
//                           registers, one 8-byte word per lane and bucket, the 10 M write requests alone cost 75 us.)
//   stage 2 (one launch):
//     split_sort_kernel     one workgroup per bucket (bucket = workgroup index: a workgroup only ever waits for smaller
//                           indices, which start first on every XCD) brings the bucket's words — low key bits << 13 |
//                           index in the bucket — into order in LDS. A bucket spans <= 2^16 rows and holds a few thousand ids, so the usual case
//                           needs no radix pass at all: a 65536-bit map of the rows present (8 KB of LDS) and its prefix
//                           popcounts give every id the RANK OF ITS ROW among the bucket's rows — which is the run it belongs
//                           to — an LDS counter per run gives it a slot in the run, an exclusive scan of the counters gives
//                           the run starts, and runs of more than one id (5 % for 10 M ids on 100 M rows) are sorted by
//                           position by the thread that owns the run. Buckets with a run of more than kMaxDup ids or
//                           more than 16 low bits take stable least-significant-digit passes of the ballot ranking
//                           instead (position digits first, then the low key bits). Either way the runs are known THERE:
//                           order[], run_starts[], unique_ids[] are written
//                           by the same kernel; the global rank of a bucket's first run comes from a decoupled look-back
//                           over the buckets' run counts (one status word per bucket, published as soon as the map is
//                           counted); the last workgroup publishes the number of runs and the closing run_starts entry.
//
// Traffic for 10 M ids: ids read twice (0.16 GB) + words written and read once (0.16 GB) + outputs (0.16 GB), against
// three read + write passes over (key, position) pairs plus histogram plus three run-detection passes before (~0.75 GB), and
// 4 launches instead of 11 + 8 fills.
//
// When a bucket does not fit (more than 4096 / 8192 ids, by the size of stage 2: a hot id of a Zipf batch, ids clustered in a few
// thousand rows), the scan
// kernel raises the overflow word, stages 1c / 2 return at once, and the caller's generic path — gated on the same word, see
// optim.hip: run_dedup — sorts the batch instead. The decision is taken on the device: no host synchronisation, capturable.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wm {
namespace split {

#ifdef WM_SPLIT_DEBUG
__device__ int g_split_debug = 0;   // experiments only: kernels with parts switched off (results are wrong on purpose)
#define WM_SPLIT_DBG(x) (g_split_debug == (x))
__device__ unsigned long long g_split_times[2][4096][12];   // [kernel][workgroup][phase] wall_clock64 of thread 0
#define WM_SPLIT_T(kern, wg, ph) do { if (threadIdx.x == 0 && (wg) < 4096) g_split_times[kern][wg][ph] = wall_clock64(); } while (0)
#else
#define WM_SPLIT_DBG(x) false
#define WM_SPLIT_T(kern, wg, ph) do { } while (0)
#endif

constexpr int kBlock      = 1024;                 // threads of a stage-1 workgroup
constexpr int kWaves      = kBlock / 64;
constexpr int kMaxBuckets = 4096;                 // real buckets (the drop bucket comes on top)
constexpr int kMaxPitch   = kMaxBuckets + 32;
// stage 2 comes in two sizes: up to 4096 ids per bucket with 512 threads (four workgroups per CU: 32 KiB of LDS each) — what a
// 10 M-id batch on a 100 M-row shard gets, 3052 buckets of 2^15 rows — or up to 8192 ids with 1024 threads (two per CU)
constexpr int kCapBitsSmall = 12, kCapBitsBig = 13;
constexpr int kMaxLowBits = 32 - kCapBitsBig;     // low key bits that fit the sort word beside the index
constexpr int kMapBits    = 16;                   // low key bits the row map of stage 2 covers (2^16 bits = 8 KB)
constexpr int kMaxDup     = 8;                    // runs of more ids than this send their bucket to the radix passes
constexpr int kMaxIpt     = 24;                   // ids per thread of a stage-1 tile, at most
constexpr int kMaxTiles   = 1024;
constexpr int kSortIpt    = 8;                    // ids per thread of a stage-2 workgroup
constexpr size_t kLdsBytes = 160 * 1024;
constexpr uint32_t kFlagAggregate = 1u << 30, kFlagPrefix = 2u << 30, kValueMask = (1u << 30) - 1;

// control words (u32), zeroed by split_hist_kernel's first workgroup
// [kCtlLongCounters, +16): the counters of the optimizer step's long-run side (optim.hip), zeroed here with the rest so that
// the step needs no fill of its own when it follows a split sort
// kCtlGenericDone: set by the generic path's closing kernel (optim.hip: detect_runs)
// kCtlScanCount: workgroups of split_scan_kernel that have finished (the last one publishes the caller's verdict word)
// kCtlSortDone: set by split_join_kernel — the runs are final (whichever path wrote them); side-stream work that needs them
// waits for this word instead of an event on the caller's stream (optim.hip: the detached long-run side)
enum { kCtlOverflow = 0, kCtlTicket = 1, kCtlError = 2, kCtlRadixBuckets = 3, kCtlGenericDone = 4, kCtlSortDone = 5, kCtlScanCount = 6, kCtlLongCounters = 16, kCtlWords = 32 };

// How long the waiting waves of this file poll before they give up and REPORT instead of hanging the device, in polls. A
// timeout sets ctl[kCtlError]; split_join_kernel (always the last kernel of a sort on the caller's stream) turns that into
// "no runs" (*n_unique = 0: the optimizer step that follows leaves the table alone) and a word in pinned host memory that the
// host looks at when it next synchronises or enters the library (optim.hip: sort_lane::take_error): a stall is an ERROR CODE,
// never a silently wrong step. stall_bucket (tests only, WM_DEBUG_STALL=lookback): the bucket of stage 2 that never publishes.
struct wait_cfg {
  uint32_t look_back_polls = 1u << 24;   // x ~0.3 us: a bucket takes 20 us, its predecessors a few rounds of that
  uint32_t join_polls      = 1u << 24;   // x 3.4 us = about a minute (the generic path takes well under a second)
  uint32_t wait_polls      = 1u << 25;   // x 1.7 us = about a minute (round 5: 2^31 = an hour)
  int stall_bucket         = -1;
};
enum { kErrLookBack = 1u, kErrJoin = 2u, kErrWait = 4u, kErrOnesweep = 8u };

struct plan {
  bool ok;            // false: the batch does not suit the split sort (too many ids per bucket, too many key bits)
  int shift;          // low key bits (ordered in LDS); bucket = key >> shift
  int buckets;        // real buckets; the drop bucket has index `buckets`
  int pitch;          // row pitch of the counts matrix (multiple of 32, >= buckets + 2)
  int ipt, tile, tiles;
  int passes, digit_bits;   // of the radix passes of stage 2 over the low key bits
  int pos_passes, pos_digit_bits;   // ... and over the positions, which come first (a bucket is not in receive order)
  int bucket_bits;    // bits of `buckets`
  int cap_bits;       // stage 2: ids per bucket = 1 << cap_bits (kCapBitsSmall or kCapBitsBig)
  // workspace carve (bytes from the workspace start)
  size_t off_keys, off_pos, off_counts, off_totals, off_starts, off_state, off_ctl, total;
};

inline size_t scatter_lds_bytes(int pitch, int ipt);
inline plan make_plan(int64_t n, int64_t span, int ipt_override = 0, int cap_bits_override = 0)
{
  plan p{};
  p.ok = false;
  if (n <= 0 || span <= 0 || span >= INT64_C(0xFFFFFFFF) || n >= (INT64_C(1) << 30)) return p;
  auto buckets_at = [&](int s) { return ((span - 1) >> s) + 1; };
  // Fewer, larger buckets are the faster ones (a bucket costs stage 2 ~20 us of barrier-separated phases whatever it holds,
  // and stage 1's segments grow with the bucket: 3052 buckets of 2^15 rows measured 197 us against 184 for 1526 of 2^16 on the
  // 10 M-id batch), so: at most kPreferBuckets buckets while the average bucket fits the big stage-2 size with headroom, up to
  // kMaxBuckets otherwise (batches of 14-28 M ids), and for small batches fewer, larger buckets still, as long as the row map
  // of stage 2 covers them.
  constexpr int kPreferBuckets = 2048;
  int s = 0;
  while (buckets_at(s) > kPreferBuckets) s++;
  if (s > 0 && n / buckets_at(s) > (1 << kCapBitsBig) * 85 / 100 && buckets_at(s - 1) <= kMaxBuckets) s--;
  const int64_t want = n / 1536 > 1 ? n / 1536 : 1;
  while (s < kMapBits && buckets_at(s) > want) s++;
  if (s > kMaxLowBits) return p;
  p.shift   = s;
  p.buckets = static_cast<int>(buckets_at(s));
  // uniform ids must leave headroom in a bucket (the overflow path is correct but it is the slow one)
  const int64_t mean = n / p.buckets;
  if (mean <= (1 << kCapBitsSmall) * 85 / 100) p.cap_bits = kCapBitsSmall;
  else if (mean <= (1 << kCapBitsBig) * 85 / 100) p.cap_bits = kCapBitsBig;
  else return p;
  if (cap_bits_override == kCapBitsBig) p.cap_bits = kCapBitsBig;   // experiments
  p.pitch = (p.buckets + 2 + 31) / 32 * 32;
  // tiles: about two rounds of the workgroups the chip holds (two per CU while a tile's LDS stays under 80 KB, else one)
  const bool wide = p.pitch > 2080;
  int ipt = static_cast<int>((n + (wide ? 500 : 1000) * kBlock - 1) / ((wide ? 500 : 1000) * kBlock));
  if (ipt < 4) ipt = 4;
  if (ipt > kMaxIpt) ipt = kMaxIpt;
  if (ipt_override > 0 && ipt_override <= kMaxIpt) ipt = ipt_override;   // experiments
  while (ipt > 4 && scatter_lds_bytes(p.pitch, ipt) > kLdsBytes - 1024) ipt--;
  p.ipt   = ipt;
  p.tile  = ipt * kBlock;
  p.tiles = static_cast<int>((n + p.tile - 1) / p.tile);
  if (p.tiles > kMaxTiles) return p;
  {
    int pb = 1;
    while (pb < 32 && ((n - 1) >> pb) != 0) pb++;
    p.pos_passes     = (pb + 7) / 8;
    p.pos_digit_bits = (pb + p.pos_passes - 1) / p.pos_passes;
  }
  p.passes      = s == 0 ? 0 : (s + 7) / 8;
  p.digit_bits  = p.passes == 0 ? 0 : (s + p.passes - 1) / p.passes;
  p.bucket_bits = 1;
  while ((1 << p.bucket_bits) <= p.buckets) p.bucket_bits++;
  auto align = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  size_t o = 0;
  p.off_keys   = o, o += align(4 * static_cast<size_t>(n));
  p.off_pos    = o, o += align(4 * static_cast<size_t>(n));
  p.off_counts = o, o += align(4 * static_cast<size_t>(p.tiles) * p.pitch);
  p.off_totals = o, o += align(4 * static_cast<size_t>(p.pitch));
  p.off_starts = o, o += align(4 * static_cast<size_t>(p.pitch + 2));
  p.off_state  = o, o += align(4 * static_cast<size_t>(p.pitch + 2));
  p.off_ctl    = o, o += align(4 * kCtlWords);
  p.total      = o;
  p.ok         = true;
  return p;
}

// upper bound of plan::total over every span (the workspace is sized before the span is known)
inline size_t workspace_bound(int64_t n)
{
  auto align = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  if (n <= 0) return 256;
  return 2 * align(4 * static_cast<size_t>(n)) + align(4 * static_cast<size_t>(kMaxTiles) * kMaxPitch) +
         3 * align(4 * (kMaxPitch + 2)) + align(4 * kCtlWords) + 256;
}

// ids as 32-bit keys relative to the owner's first row; an id outside [base, base + span) reads as the key `span`
template <typename UKey>
struct key_source {
  const UKey* ids;
  UKey base;
  uint32_t span;
  __device__ __forceinline__ uint32_t narrow(UKey id) const
  {
    const UKey off = id - base;
    return off < static_cast<UKey>(span) ? static_cast<uint32_t>(off) : span;
  }
  __device__ __forceinline__ uint32_t key(int64_t i) const { return narrow(ids[i]); }
};

// consecutive tiles on the same XCD (workgroups go to the 8 XCDs round-robin): the segments two neighbouring tiles write
// into a bucket are neighbours in memory and meet in one L2 instead of in two
__device__ __forceinline__ int tile_of_block(int bid, int tiles)
{
  const int per = (tiles + 7) / 8;
  return (bid & 7) * per + (bid >> 3);
}
inline int tile_grid(int tiles) { return (tiles + 7) / 8 * 8; }

template <int WAVES = kWaves>
__device__ __forceinline__ uint32_t block_exclusive_sum(uint32_t v, uint32_t* s_waves /* WAVES words */, uint32_t* total)
{
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t incl  = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  __syncthreads();   // s_waves may still be read from an earlier call
  if (lane == 63) s_waves[wv] = incl;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < WAVES; w++) {
    const uint32_t t = s_waves[w];
    if (w < wv) before += t;
    all += t;
  }
  if (total != nullptr) *total = all;
  return before + incl - v;
}

// lanes of the wave that hold the same `value` (low `bits` bits compared) among the lanes in `among`
__device__ __forceinline__ uint64_t match_lanes(uint32_t value, int bits, uint64_t among)
{
  uint64_t m = among;
  for (int b = 0; b < bits; b++) {
    const bool bit     = (value >> b) & 1u;
    const uint64_t bal = __ballot(bit);
    m &= bit ? bal : ~bal;
  }
  return m;
}

// ---- stage 1a: per-tile bucket histogram -------------------------------------------------------------------------------
template <typename UKey>
__global__ __launch_bounds__(kBlock) void split_hist_kernel(key_source<UKey> src, int64_t n, int tile, int tiles, int shift,
                                                            int buckets, int pitch, uint32_t* counts, uint32_t* ctl,
                                                            uint32_t* zero_words, int64_t n_zero_words)
{
  __shared__ uint32_t s_hist[kMaxPitch];
  if (blockIdx.x == 0 && threadIdx.x < kCtlWords) ctl[threadIdx.x] = 0;
  // the control words of the caller's generic path (look-back state of its passes): zeroed here, on the way, not by a fill
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n_zero_words;
       i += static_cast<int64_t>(gridDim.x) * kBlock)
    zero_words[i] = 0;
  const int t = tile_of_block(blockIdx.x, tiles);
  if (t >= tiles) return;
  for (int i = threadIdx.x; i < pitch; i += kBlock) s_hist[i] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(t) * tile;
  const int64_t end  = base + tile < n ? base + tile : n;
  int64_t i          = base + threadIdx.x;
  constexpr int U    = 8;
  for (; i + (U - 1) * kBlock < end; i += U * kBlock) {
    uint32_t k[U];
#pragma unroll
    for (int u = 0; u < U; u++) k[u] = src.key(i + u * kBlock);
#pragma unroll
    for (int u = 0; u < U; u++) atomicAdd(&s_hist[k[u] >= src.span ? buckets : static_cast<int>(k[u] >> shift)], 1u);
  }
  for (; i < end; i += kBlock) {
    const uint32_t k = src.key(i);
    atomicAdd(&s_hist[k >= src.span ? buckets : static_cast<int>(k >> shift)], 1u);
  }
  __syncthreads();
  uint32_t* row = counts + static_cast<size_t>(t) * pitch;
  for (int j = threadIdx.x; j < pitch; j += kBlock) row[j] = s_hist[j];
}

// ---- stage 1b: column-wise exclusive prefix of the counts matrix --------------------------------------------------------
// one workgroup = 32 columns x all tiles: thread (row group rg, column c) owns rows_per consecutive tiles of its column
__global__ __launch_bounds__(kBlock) void split_scan_kernel(uint32_t* counts, int tiles, int pitch, int buckets, uint32_t* totals,
                                                            uint32_t* ctl, int cap, uint32_t* state, int state_words,
                                                            uint32_t* verdict_word, uint32_t verdict_value)
{
  __shared__ uint32_t s_g[32][33];
  const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int col      = blockIdx.x * 32 + c;
  const int rows_per = (tiles + 31) / 32;   // <= kMaxTiles / 32 = 32
  const int r0       = rg * rows_per;
  uint32_t v[kMaxTiles / 32];
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < kMaxTiles / 32; k++) {
    const int r = r0 + k;
    v[k]        = (k < rows_per && r < tiles && col < pitch) ? counts[static_cast<size_t>(r) * pitch + col] : 0u;
  }
#pragma unroll
  for (int k = 0; k < kMaxTiles / 32; k++) {
    const uint32_t t = v[k];
    v[k]             = sum;
    sum += t;
  }
  s_g[rg][c] = sum;
  __syncthreads();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (int g = 0; g < 32; g++) {
    const uint32_t t = s_g[g][c];
    if (g < rg) before += t;
    total += t;
  }
#pragma unroll
  for (int k = 0; k < kMaxTiles / 32; k++) {
    const int r = r0 + k;
    if (k < rows_per && r < tiles && col < pitch) counts[static_cast<size_t>(r) * pitch + col] = before + v[k];
  }
  if (rg == 0 && col < pitch) {
    totals[col] = total;
    if (col < buckets && total > static_cast<uint32_t>(cap)) atomicOr(&ctl[kCtlOverflow], 1u);
  }
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < state_words; i += gridDim.x * kBlock) state[i] = 0;
  // "the overflow word is final": the last workgroup to get here says so in a word of the caller's (a side stream's first
  // kernel waits for it — split_wait_kernel — instead of an event recorded behind this kernel, which held up the NEXT kernel
  // of this stream by ~7 us)
  if (verdict_word != nullptr) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      if (atomicAdd(&ctl[kCtlScanCount], 1u) == gridDim.x - 1)
        __hip_atomic_store(verdict_word, verdict_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- stage 1c: stable multisplit --------------------------------------------------------------------------------------
inline size_t scatter_lds_bytes(int pitch, int ipt)
{
  // global segment starts (4 B per bucket) + tile-local segment starts and slot counters (2 B each per bucket) + the tile's
  // keys (4 B) and indices in the tile (2 B)
  return 4 * static_cast<size_t>(pitch) + 2 * 2 * static_cast<size_t>(pitch + 2) + 6 * static_cast<size_t>(ipt) * kBlock + 16;
}

// Multisplit of one tile: every id gets a slot in its bucket's segment of the tile from ONE LDS counter per bucket, in
// whatever order the lanes arrive — so a bucket's share of a tile is NOT in receive order. It does not have to be: stage 2
// orders equal ids by their POSITION (carried beside every key), not by where stage 1 left them, and nothing else in a bucket
// depends on the order inside a tile's segment. (The first version ranked every id stably among its bucket's ids of the tile —
// per-wave counters, their prefix over the waves, a lane-order fix per step: 31 us per tile of 10 k ids, profiles/
// r05_split_sort_harness.txt history; this is one atomic per id and two barriers: ~12 us.)
// PER: buckets per thread in the per-bucket loops (3 up to 3072 buckets + pitch slack, 5 up to kMaxBuckets)
template <typename UKey, int MAXIPT, int PER>
__global__ __launch_bounds__(kBlock, MAXIPT <= 12 ? 8 : 4) void split_scatter_kernel(key_source<UKey> src, int64_t n, int ipt, int tiles,
                                                                                   int shift, int buckets, int bucket_bits, int pitch,
                                                                                   const uint32_t* counts, const uint32_t* totals,
                                                                                   uint32_t* bucket_start, uint32_t* keys_out,
                                                                                   uint32_t* pos_out, const uint32_t* ctl)
{
  if (ctl[kCtlOverflow] != 0) return;
  const int t = tile_of_block(blockIdx.x, tiles);
  if (t >= tiles) return;
  extern __shared__ uint32_t s_mem[];
  const int tile   = ipt * kBlock;
  uint32_t* s_off  = s_mem;                                                 // [pitch]     global position of this tile's segment of a bucket MINUS its start in the tile
  uint16_t* s_lst  = reinterpret_cast<uint16_t*>(s_off + pitch);            // [pitch + 2] start of the bucket's segment in the tile
  uint16_t* s_one  = s_lst + pitch + 2;                                     // [pitch + 2] slots handed out (two buckets share a 32-bit word)
  uint32_t* s_keys = reinterpret_cast<uint32_t*>(s_one + pitch + 2);        // [tile]      the tile in bucket order
  uint16_t* s_idx  = reinterpret_cast<uint16_t*>(s_keys + tile);            // [tile]      index in the tile of the id at that place
  __shared__ uint32_t s_waves[kWaves];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  WM_SPLIT_T(0, blockIdx.x, 0);

  // load, wave-striped: wave w owns ids [w, w + 1) x 64 x ipt of the tile, item j of lane l is id j x 64 + l of that chunk:
  // every load instruction is one contiguous 512-byte read (issued first: the scans below run under their latency)
  const int64_t base  = static_cast<int64_t>(t) * tile;
  const int local0    = wv * (64 * ipt) + lane;
  const int valid_n   = static_cast<int>(n - base < tile ? n - base : tile);
  uint32_t key[MAXIPT];
  // UNCONDITIONAL loads, in batches of up to 8 with nothing between them (a load under a condition is a load the compiler waits
  // for before it issues the next: scripts/check_isa.py counted ONE load in flight here): lanes past the tile's end and items
  // past ipt re-read the tile's last id
#pragma unroll
  for (int j0 = 0; j0 < MAXIPT; j0 += 8) {
    UKey raw[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int li  = local0 + (j0 + u) * 64;
      const bool in = j0 + u < MAXIPT && j0 + u < ipt && li < valid_n;
      raw[u]        = src.ids[base + (in ? li : valid_n - 1)];
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (j0 + u < MAXIPT) key[j0 + u] = (j0 + u < ipt && local0 + (j0 + u) * 64 < valid_n) ? src.narrow(raw[u]) : 0xFFFFFFFFu;
  }

  // bucket starts = exclusive scan of the totals (every workgroup redoes these ~2 k additions rather than wait for a kernel);
  // this tile's ids per bucket = the difference of two rows of the scanned matrix; their exclusive scan = the segment starts
  // inside the tile
  const int per = (pitch + kBlock - 1) / kBlock;   // <= PER consecutive buckets per thread
  const int b0  = threadIdx.x * per;
  {
    uint32_t tt[PER] = {}, row[PER] = {}, cnt[PER] = {}, mine = 0, mine_t = 0;
#pragma unroll
    for (int i = 0; i < PER; i++)
      if (i < per && b0 + i < pitch) {
        tt[i]  = totals[b0 + i];
        row[i] = counts[static_cast<size_t>(t) * pitch + b0 + i];
        cnt[i] = (t + 1 < tiles ? counts[static_cast<size_t>(t + 1) * pitch + b0 + i] : tt[i]) - row[i];
        mine += tt[i];
        mine_t += cnt[i];
      }
    uint32_t run    = block_exclusive_sum(mine, s_waves, nullptr);
    uint32_t lstart = block_exclusive_sum(mine_t, s_waves, nullptr);
#pragma unroll
    for (int i = 0; i < PER; i++)
      if (i < per && b0 + i < pitch) {
        s_off[b0 + i] = run + row[i] - lstart;
        s_lst[b0 + i] = static_cast<uint16_t>(lstart);
        if (t == 0) bucket_start[b0 + i] = run;   // [buckets] = number of ids inside the range, [buckets + 1] = n
        run += tt[i];
        lstart += cnt[i];
      }
    uint32_t* one32 = reinterpret_cast<uint32_t*>(s_one);
    for (int i = threadIdx.x; i < (pitch + 2) / 2; i += kBlock) one32[i] = 0;
  }
  __syncthreads();
  WM_SPLIT_T(0, blockIdx.x, 1);
  {
    uint32_t* one32 = reinterpret_cast<uint32_t*>(s_one);
#pragma unroll
    for (int j = 0; j < MAXIPT; j++)
      if (j < ipt && local0 + j * 64 < valid_n) {
        const uint32_t b  = key[j] >= src.span ? static_cast<uint32_t>(buckets) : key[j] >> shift;
        const int sh      = (b & 1u) * 16;
        const uint32_t o  = (atomicAdd(&one32[b >> 1], 1u << sh) >> sh) & 0xFFFFu;
        const uint32_t lp = s_lst[b] + o;
        s_keys[lp]        = key[j];
        s_idx[lp]         = static_cast<uint16_t>(local0 + j * 64);
      }
  }
  __syncthreads();
  WM_SPLIT_T(0, blockIdx.x, 2);
#pragma unroll
  for (int k = 0; k < MAXIPT; k++) {
    const int sl = k * kBlock + threadIdx.x;
    if (k < ipt && sl < valid_n && !WM_SPLIT_DBG(1)) {
      const uint32_t x  = s_keys[sl];
      const uint32_t gp = s_off[x >= src.span ? static_cast<uint32_t>(buckets) : x >> shift] + static_cast<uint32_t>(sl);
      keys_out[gp]      = x;
      pos_out[gp]       = static_cast<uint32_t>(base) + s_idx[sl];
    }
  }
  WM_SPLIT_T(0, blockIdx.x, 6);
}

// ---- stage 2: per-bucket order in LDS + run detection --------------------------------------------------------------------
// wave-wide decoupled look-back over the run counts of the buckets before `b`; called by wave 0, returns the exclusive prefix
__device__ __forceinline__ uint32_t look_back(uint32_t* state, int b, uint32_t* ctl, const wait_cfg& wc)
{
  const int lane = threadIdx.x & 63;
  uint32_t excl  = 0;
  int look       = b - 1;
  unsigned spins = 0;
  constexpr int U = 4;   // windows of 64 buckets loaded together (the buckets of one round of workgroups are all "aggregate")
  while (look >= 0) {
    uint32_t v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int idx = look - 64 * u - lane;
      v[u]          = idx >= 0 ? __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kFlagPrefix;
    }
    bool done = false, stalled = false;
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (!done && !stalled) {
        const uint32_t f     = v[u] >> 30;
        const uint64_t empty = __ballot(f == 0u), prefix = __ballot(f == 2u);
        // lanes up to (and with) the nearest bucket that knows its prefix
        const uint64_t need = prefix != 0 ? ((2ull << (__ffsll(static_cast<long long>(prefix)) - 1)) - 1ull) : ~0ull;
        if ((empty & need) != 0) {
          stalled = true;   // a bucket in reach has not published yet: come back to this window
        } else {
          uint32_t part = ((need >> lane) & 1ull) ? (v[u] & kValueMask) : 0u;
#pragma unroll
          for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
          excl += part;
          look -= 64;
          done = prefix != 0;
        }
      }
    }
    if (done) break;
    if (stalled) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > wc.look_back_polls) {   // never in a healthy run: report instead of hanging the device
        if (lane == 0) atomicOr(&ctl[kCtlError], static_cast<uint32_t>(kErrLookBack));
        break;
      }
    }
  }
  return excl;
}

__device__ __forceinline__ void publish(const wait_cfg& wc, uint32_t* state, int b, uint32_t flag, uint32_t value)
{
  if (b == wc.stall_bucket) return;   // (tests: a gate that never opens)
  __hip_atomic_store(&state[b], flag | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (second launch bound = waves per SIMD: 8 = as many workgroups per CU as 32 waves make: two of 1024 threads, four of 512)
template <typename OutT, int CAPBITS>
__global__ __launch_bounds__((1 << CAPBITS) / kSortIpt, 8) void split_sort_kernel(const uint32_t* keys, const uint32_t* pos,
                                                               const uint32_t* bucket_start, int buckets, int shift, int passes,
                                                               int digit_bits, int pos_passes, int pos_digit_bits, OutT key_base, OutT* unique_ids, int32_t* run_starts,
                                                               int32_t* order, int64_t* n_unique, uint32_t* ctl, uint32_t* state,
                                                               wait_cfg wc)
{
  if (ctl[kCtlOverflow] != 0) return;
  constexpr int CAP = 1 << CAPBITS, BLOCK = CAP / kSortIpt, WAVES = BLOCK / 64;
  __shared__ uint32_t s_buf[CAP];   // the bucket's words in order; before that: the row map (2048 words) + its prefix (2048)
  __shared__ uint32_t s_run[CAP];   // map path: ids per run, then run starts; radix path: per-wave digit counters
  __shared__ uint32_t s_waves[WAVES];
  __shared__ uint32_t s_misc[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  WM_SPLIT_T(1, blockIdx.x, 0);
  // The bucket IS the workgroup index: a workgroup waits (in look_back) only for smaller indices, and workgroups start in
  // index order on every XCD, so the smallest unfinished bucket always runs or is next in line for a slot held by finished-or-
  // running smaller ones — no co-residency assumption (the argument of graph.hip's chain scan). A ticket counter here
  // serialised the workgroups' starts at ~27 ns each: 14 us per round of 512 workgroups (profiles/r05_split_sort_harness.txt).
  if (threadIdx.x == 0) s_misc[2] = 0;
  __syncthreads();
  const int b          = static_cast<int>(blockIdx.x);
  const uint32_t start = bucket_start[b];
  const int m          = static_cast<int>(bucket_start[b + 1] - start);

  if (b == buckets) {
    // the drop bucket: positions of the ids outside the range fill the tail of order[]; then the totals
    for (int i = threadIdx.x; i < m; i += BLOCK) order[start + i] = static_cast<int32_t>(pos[start + i]);
    if (wv == 0) {
      const uint32_t total = look_back(state, buckets, ctl, wc);
      if (lane == 0) {
        *n_unique         = static_cast<int64_t>(total);
        run_starts[total] = static_cast<int32_t>(start);
      }
    }
    return;
  }
  if (m == 0) {
    // an empty bucket still takes its place in the chain
    if (wv == 0) {
      if (lane == 0) publish(wc, state, b, b == 0 ? kFlagPrefix : kFlagAggregate, 0u);
      if (b > 0) {
        const uint32_t excl = look_back(state, b, ctl, wc);
        if (lane == 0) publish(wc, state, b, kFlagPrefix, excl);
      }
    }
    return;
  }

  // wave w owns positions [w, w + 1) x chunk of the bucket, chunk = a multiple of 64 with 16 chunks covering m
  const int steps = (m + BLOCK - 1) / BLOCK;   // <= kSortIpt
  const int chunk = steps * 64;
  const int p0    = wv * chunk + lane;
  const uint32_t low_mask = (1u << shift) - 1u;   // shift <= kMaxLowBits
  uint32_t w[kSortIpt], slot[kSortIpt];
  {
    uint32_t raw[kSortIpt];   // unconditional loads, all in flight together (positions past the bucket re-read its last key)
#pragma unroll
    for (int j = 0; j < kSortIpt; j++) {
      const int p = p0 + j * 64;
      raw[j]      = keys[start + (j < steps && p < m ? p : m - 1)];
    }
#pragma unroll
    for (int j = 0; j < kSortIpt; j++) {
      const int p = p0 + j * 64;
      w[j]        = (j < steps && p < m) ? ((raw[j] & low_mask) << CAPBITS) | static_cast<uint32_t>(p) : 0xFFFFFFFFu;
    }
  }
  const OutT bucket_key = (static_cast<OutT>(b) << shift) + key_base;
  bool radix            = shift > kMapBits;
  bool published        = false;
  uint32_t heads_total  = 0;

  if (!radix) {
    const int map_words = shift > 5 ? 1 << (shift - 5) : 1;   // <= 2048 (shift <= kMapBits)
    uint32_t* s_map = s_buf;               // [map_words] bit r: row r of the bucket is present
    uint32_t* s_pre = s_buf + map_words;   // [map_words] present rows before the word
    WM_SPLIT_T(1, blockIdx.x, 1);
    for (int i = threadIdx.x; i < map_words; i += BLOCK) s_map[i] = 0;
    for (int i = threadIdx.x; i < m; i += BLOCK) s_run[i] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kSortIpt; j++)
      if (j < steps && p0 + j * 64 < m) atomicOr(&s_map[w[j] >> (CAPBITS + 5)], 1u << ((w[j] >> CAPBITS) & 31u));
    __syncthreads();
    WM_SPLIT_T(1, blockIdx.x, 2);
    {
      // a thread owns wpt consecutive words of the map (2 for the usual shapes: 2048 words / 1024 threads, 1024 / 512)
      const int wpt = map_words >= BLOCK ? map_words / BLOCK : 1;   // <= 4
      const int w0  = threadIdx.x * wpt;
      uint32_t c[4] = {0, 0, 0, 0}, mine = 0;
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (q < wpt && w0 + q < map_words) c[q] = static_cast<uint32_t>(__popc(s_map[w0 + q])), mine += c[q];
      uint32_t ex = block_exclusive_sum<WAVES>(mine, s_waves, &heads_total);
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (q < wpt && w0 + q < map_words) s_pre[w0 + q] = ex, ex += c[q];
    }
    // the bucket's run count is known: tell the buckets behind this one now, look back later
    if (threadIdx.x == 0) publish(wc, state, b, b == 0 ? kFlagPrefix : kFlagAggregate, heads_total);
    published = true;
    __syncthreads();
    WM_SPLIT_T(1, blockIdx.x, 3);
    bool long_run = false;
#pragma unroll
    for (int j = 0; j < kSortIpt; j++)
      if (j < steps && p0 + j * 64 < m) {
        const uint32_t k  = w[j] >> CAPBITS;
        const uint32_t r  = s_pre[k >> 5] + static_cast<uint32_t>(__popc(s_map[k >> 5] & ((1u << (k & 31u)) - 1u)));
        const uint32_t o  = atomicAdd(&s_run[r], 1u);
        slot[j]           = r | (o << 16);
        long_run |= o >= static_cast<uint32_t>(kMaxDup);
      }
    if (long_run) s_misc[2] = 1;
    __syncthreads();   // (also: the map has been read, its memory becomes the word buffer)
    radix = s_misc[2] != 0;
    WM_SPLIT_T(1, blockIdx.x, 4);
    if (!radix) {
      // run starts: exclusive scan of the ids per run, 8 consecutive runs per thread
      {
        const int r0 = threadIdx.x * kSortIpt;
        uint32_t c[kSortIpt], mine = 0;
#pragma unroll
        for (int i = 0; i < kSortIpt; i++) c[i] = r0 + i < static_cast<int>(heads_total) ? s_run[r0 + i] : 0u, mine += c[i];
        uint32_t run = block_exclusive_sum<WAVES>(mine, s_waves, nullptr);
#pragma unroll
        for (int i = 0; i < kSortIpt; i++) {
          if (r0 + i < static_cast<int>(heads_total)) s_run[r0 + i] = run;
          run += c[i];
        }
      }
      __syncthreads();
      WM_SPLIT_T(1, blockIdx.x, 5);
#pragma unroll
      for (int j = 0; j < kSortIpt; j++)
        if (j < steps && p0 + j * 64 < m) s_buf[s_run[slot[j] & 0xFFFFu] + (slot[j] >> 16)] = w[j];
      __syncthreads();
      WM_SPLIT_T(1, blockIdx.x, 6);
      // a run of several ids is in the order its ids reached the counter, and the bucket itself is not in receive order
      // (stage 1 places a tile's ids of a bucket in arrival order): put the run into receive order = ascending POSITION.
      // Runs of one id — 95 % of them for 10 M ids on 100 M rows — are left alone; the others (2 ... kMaxDup ids) are sorted
      // by the thread that owns the run, positions read through the index in the word.
      for (int r = threadIdx.x; r < static_cast<int>(heads_total); r += BLOCK) {
        const int i0 = static_cast<int>(s_run[r]);
        const int i1 = r + 1 < static_cast<int>(heads_total) ? static_cast<int>(s_run[r + 1]) : m;
        for (int i = i0 + 1; i < i1; i++) {
          const uint32_t x  = s_buf[i];
          const uint32_t px = pos[start + (x & (CAP - 1))];
          int q             = i;
          while (q > i0 && pos[start + (s_buf[q - 1] & (CAP - 1))] > px) {
            s_buf[q] = s_buf[q - 1];
            q--;
          }
          s_buf[q] = x;
        }
      }
      __syncthreads();
      WM_SPLIT_T(1, blockIdx.x, 7);
      if (wv == 0) {
        const uint32_t excl = look_back(state, b, ctl, wc);
        if (lane == 0) {
          if (b > 0) publish(wc, state, b, kFlagPrefix, excl + heads_total);
          s_misc[1] = excl;
        }
      }
      WM_SPLIT_T(1, blockIdx.x, 8);
      {
        uint32_t pv[kSortIpt];   // the positions of the bucket's ids in their new order: 8 gathers in flight, then 8 stores
#pragma unroll
        for (int k = 0; k < kSortIpt; k++) {
          const int i = k * BLOCK + threadIdx.x;
          pv[k]       = pos[start + (i < m ? (s_buf[i] & (CAP - 1)) : 0u)];
        }
#pragma unroll
        for (int k = 0; k < kSortIpt; k++) {
          const int i = k * BLOCK + threadIdx.x;
          if (i < m) order[start + i] = static_cast<int32_t>(pv[k]);
        }
      }
      __syncthreads();
      WM_SPLIT_T(1, blockIdx.x, 9);
      const uint32_t run_base = s_misc[1];
      for (int r = threadIdx.x; r < static_cast<int>(heads_total); r += BLOCK) {
        const uint32_t i0        = s_run[r];
        run_starts[run_base + r] = static_cast<int32_t>(start + i0);
        unique_ids[run_base + r] = bucket_key + static_cast<OutT>(s_buf[i0] >> CAPBITS);
      }
      WM_SPLIT_T(1, blockIdx.x, 10);
      return;
    }
    if (threadIdx.x == 0) atomicAdd(&ctl[kCtlRadixBuckets], 1u);   // statistics: buckets the map path handed over
  }

  // ---- radix path: stable least-significant-digit passes over the low key bits --------------------------------------------
  if (shift > kMapBits && threadIdx.x == 0) atomicAdd(&ctl[kCtlRadixBuckets], 1u);   // (a bucket the map never saw: it may hold runs of any length)
  {
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t* s_cnt   = s_run;   // [WAVES][256]
    uint32_t* my_cnt  = s_cnt + wv * 256;
    // The bucket arrives grouped by tile but in arrival order inside a tile's segment, and a stable sort by the key alone would
    // keep that order among equal ids: so the positions are sorted first (pos_passes digits of them, read through the index in
    // the word: 8 gathers in flight per pass), then the low key bits — least significant digit first over (key, position).
    const int all_passes = pos_passes + passes;
    for (int pass = 0; pass < all_passes; pass++) {
      const bool on_pos    = pass < pos_passes;
      const int dbits      = on_pos ? pos_digit_bits : digit_bits;
      const int bins       = 1 << dbits;
      const int sh         = on_pos ? pass * pos_digit_bits : CAPBITS + (pass - pos_passes) * digit_bits;
      const uint32_t dmask = static_cast<uint32_t>(bins - 1);
      uint32_t dig[kSortIpt];
      if (on_pos) {
#pragma unroll
        for (int j = 0; j < kSortIpt; j++) dig[j] = pos[start + ((j < steps && p0 + j * 64 < m) ? (w[j] & (CAP - 1)) : 0u)];
      }
#pragma unroll
      for (int j = 0; j < kSortIpt; j++) dig[j] = ((on_pos ? dig[j] : w[j]) >> sh) & dmask;
      for (int i = lane; i < bins; i += 64) my_cnt[i] = 0;
      // (a wave's counters are its own until the scan below: no barrier between the zeroing and the ranking)
#pragma unroll
      for (int j = 0; j < kSortIpt; j++) {
        if (j < steps && wv * chunk + j * 64 < m) {
          const bool valid = p0 + j * 64 < m;
          const uint32_t d = dig[j];
          const uint64_t g = match_lanes(d, dbits, __ballot(valid));
          const int before = __popcll(g & lt);
          uint32_t old     = 0;
          if (valid && before == 0) old = atomicAdd(&my_cnt[d], static_cast<uint32_t>(__popcll(g)));
          old     = __shfl(old, __ffsll(static_cast<long long>(g | (1ull << 63))) - 1, 64);
          slot[j] = old + before;
        }
      }
      __syncthreads();
      // per digit: prefix over the waves, then the exclusive scan over the digits
      uint32_t mine = 0;
      if (threadIdx.x < bins) {
        uint32_t run = 0;
#pragma unroll
        for (int ww = 0; ww < WAVES; ww++) {
          const uint32_t c              = s_cnt[ww * 256 + threadIdx.x];
          s_cnt[ww * 256 + threadIdx.x] = run;
          run += c;
        }
        mine = run;
      }
      const uint32_t dbase = block_exclusive_sum<WAVES>(mine, s_waves, nullptr);
      if (threadIdx.x < bins) {
#pragma unroll
        for (int ww = 0; ww < WAVES; ww++) s_cnt[ww * 256 + threadIdx.x] += dbase;
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < kSortIpt; j++) {
        if (j < steps && p0 + j * 64 < m) s_buf[my_cnt[dig[j]] + slot[j]] = w[j];
      }
      __syncthreads();
      if (pass + 1 < all_passes) {
#pragma unroll
        for (int j = 0; j < kSortIpt; j++) {
          const int p = p0 + j * 64;
          if (j < steps && p < m) w[j] = s_buf[p];
        }
        __syncthreads();   // the counters are zeroed and the buffer rewritten by the next pass
      }
    }
    if (all_passes == 0) {
      // (never: a position always has a digit)
#pragma unroll
      for (int j = 0; j < kSortIpt; j++) {
        const int p = p0 + j * 64;
        if (j < steps && p < m) s_buf[p] = w[j];
      }
      __syncthreads();
    }
    // runs: position i (striped: i = k x 1024 + thread) is a head when its low key differs from its predecessor's
    uint64_t head_mask[kSortIpt];
    uint32_t* s_hc = s_run;   // [kSortIpt][WAVES] head counts, then their exclusive prefix
#pragma unroll
    for (int k = 0; k < kSortIpt; k++) {
      const int i = k * BLOCK + threadIdx.x;
      bool head   = false;
      w[k]        = 0;
      if (k < steps && i < m) {
        w[k]                = s_buf[i];
        const uint32_t prev = i > 0 ? s_buf[i - 1] : ~w[k];
        head                = (w[k] >> CAPBITS) != (prev >> CAPBITS);
        order[start + i]    = static_cast<int32_t>(pos[start + (w[k] & (CAP - 1))]);
      }
      head_mask[k] = __ballot(head);
      if (lane == 0) s_hc[k * WAVES + wv] = static_cast<uint32_t>(__popcll(head_mask[k]));
    }
    __syncthreads();
    const uint32_t hv = threadIdx.x < kSortIpt * WAVES ? s_hc[threadIdx.x] : 0u;
    uint32_t ht;
    const uint32_t hx = block_exclusive_sum<WAVES>(hv, s_waves, &ht);
    if (threadIdx.x < kSortIpt * WAVES) s_hc[threadIdx.x] = hx;
    __syncthreads();
    if (wv == 0) {
      if (lane == 0 && !published) publish(wc, state, b, b == 0 ? kFlagPrefix : kFlagAggregate, ht);
      const uint32_t excl = look_back(state, b, ctl, wc);
      if (lane == 0) {
        if (b > 0) publish(wc, state, b, kFlagPrefix, excl + ht);
        s_misc[1] = excl;
      }
    }
    __syncthreads();
    const uint32_t run_base = s_misc[1];
#pragma unroll
    for (int k = 0; k < kSortIpt; k++) {
      const int i = k * BLOCK + threadIdx.x;
      if (k < steps && ((head_mask[k] >> lane) & 1ull)) {
        const uint32_t r = run_base + s_hc[k * WAVES + wv] + static_cast<uint32_t>(__popcll(head_mask[k] & ((1ull << lane) - 1ull)));
        run_starts[r]    = static_cast<int32_t>(start + i);
        unique_ids[r]    = bucket_key + static_cast<OutT>(w[k] >> CAPBITS);
      }
    }
  }
}

// enqueues the four launches. `zero_words`: control words of the caller's generic path that have to read zero before it
// runs (may be null). `between`: called after the second launch, when the overflow word is final for whatever is enqueued
// from then on (the caller forks its generic path there). Returns 0 or -2.
struct no_hook {
  void operator()() const {}
};
template <typename UKey, typename Hook = no_hook>
int launch(const plan& p, const UKey* ids, int64_t n, UKey key_lower_bound, uint32_t span, void* unique_ids, int32_t* run_starts,
           int32_t* order, int64_t* n_unique, void* workspace, uint32_t* zero_words, int64_t n_zero_words, hipStream_t stream,
           Hook between = Hook(), bool hook_after_scatter = false, uint32_t* verdict_word = nullptr, uint32_t verdict_value = 0,
           const wait_cfg& wc = wait_cfg())
{
  char* ws         = static_cast<char*>(workspace);
  uint32_t* keys   = reinterpret_cast<uint32_t*>(ws + p.off_keys);
  uint32_t* pos    = reinterpret_cast<uint32_t*>(ws + p.off_pos);
  uint32_t* counts = reinterpret_cast<uint32_t*>(ws + p.off_counts);
  uint32_t* totals = reinterpret_cast<uint32_t*>(ws + p.off_totals);
  uint32_t* starts = reinterpret_cast<uint32_t*>(ws + p.off_starts);
  uint32_t* state  = reinterpret_cast<uint32_t*>(ws + p.off_state);
  uint32_t* ctl    = reinterpret_cast<uint32_t*>(ws + p.off_ctl);
  key_source<UKey> src{ids, key_lower_bound, span};
  static bool attr_set = [] {
    const int most = static_cast<int>(kLdsBytes - 1024);   // (static LDS comes on top; make_plan keeps a tile below this)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&split_scatter_kernel<UKey, 12, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, most);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&split_scatter_kernel<UKey, kMaxIpt, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, most);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&split_scatter_kernel<UKey, kMaxIpt, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, most);
    return true;
  }();
  (void)attr_set;
  const int grid = tile_grid(p.tiles);
  hipLaunchKernelGGL((split_hist_kernel<UKey>), dim3(grid), dim3(kBlock), 0, stream, src, n, p.tile, p.tiles, p.shift, p.buckets,
                     p.pitch, counts, ctl, zero_words, n_zero_words);
  hipLaunchKernelGGL(split_scan_kernel, dim3(p.pitch / 32), dim3(kBlock), 0, stream, counts, p.tiles, p.pitch, p.buckets, totals, ctl,
                     1 << p.cap_bits, state, p.pitch + 2, verdict_word, verdict_value);
  if (!hook_after_scatter) between();
  const size_t lds = scatter_lds_bytes(p.pitch, p.ipt);
#define WM_SPLIT_SCATTER(MAXI, PERB)                                                                                             \
  hipLaunchKernelGGL((split_scatter_kernel<UKey, MAXI, PERB>), dim3(grid), dim3(kBlock), lds, stream, src, n, p.ipt, p.tiles, p.shift, \
                     p.buckets, p.bucket_bits, p.pitch, counts, totals, starts, keys, pos, ctl)
  if (p.pitch > 3 * kBlock) WM_SPLIT_SCATTER(kMaxIpt, 5);
  else if (p.ipt <= 12 && lds <= kLdsBytes / 2) WM_SPLIT_SCATTER(12, 3);
  else WM_SPLIT_SCATTER(kMaxIpt, 3);
#undef WM_SPLIT_SCATTER
  if (hook_after_scatter) between();
  if (p.cap_bits == kCapBitsSmall)
    hipLaunchKernelGGL((split_sort_kernel<UKey, kCapBitsSmall>), dim3(p.buckets + 1), dim3((1 << kCapBitsSmall) / kSortIpt), 0, stream,
                       keys, pos, starts, p.buckets, p.shift, p.passes, p.digit_bits, p.pos_passes, p.pos_digit_bits, key_lower_bound,
                       static_cast<UKey*>(unique_ids), run_starts, order, n_unique, ctl, state, wc);
  else
    hipLaunchKernelGGL((split_sort_kernel<UKey, kCapBitsBig>), dim3(p.buckets + 1), dim3((1 << kCapBitsBig) / kSortIpt), 0, stream,
                       keys, pos, starts, p.buckets, p.shift, p.passes, p.digit_bits, p.pos_passes, p.pos_digit_bits, key_lower_bound,
                       static_cast<UKey*>(unique_ids), run_starts, order, n_unique, ctl, state, wc);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// The LAST kernel of a sort on the caller's stream: one wave that returns at once in the usual case. When the generic path
// runs on a side stream and is not joined by an event before the step (optim.hip: deferred join) and the batch overflowed, it
// waits until the generic path's closing kernel has counted itself in. Then it publishes "the runs are final" — or, when any
// wait of this sort timed out (ctl[kCtlError], which the generic path's closing kernel also folds the onesweep passes' word
// into), "the sort FAILED": kCtlSortDone = 2, *n_unique = 0 (every consumer reads its run count there: the step that follows
// does nothing) and the code in *host_err (pinned host memory; optim.hip reports it at the next synchronise / entry).
__global__ void split_join_kernel(uint32_t* ctl, uint32_t expected_blocks, uint32_t join_polls, int64_t* n_unique, uint32_t* host_err)
{
  if (ctl[kCtlOverflow] != 0) {
    // (RELAXED polls, far apart: an acquire at agent scope invalidates cache lines on every poll, and the kernels this wave is
    // waiting for ran 30 % slower beside it; one acquire fence at the end is all the ordering needed)
    unsigned spins = 0;
    while (__hip_atomic_load(&ctl[kCtlGenericDone], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected_blocks) {
      __builtin_amdgcn_s_sleep(127);   // ~8 k cycles = 3.4 us between polls
      if (++spins > join_polls) {   // never in a healthy run: report, do not hang
        if (threadIdx.x == 0) atomicOr(&ctl[kCtlError], static_cast<uint32_t>(kErrJoin));
        break;
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  if (threadIdx.x == 0) {
    const uint32_t err = __hip_atomic_load(&ctl[kCtlError], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (err != 0) {
      if (n_unique != nullptr) *n_unique = 0;
      if (host_err != nullptr) __hip_atomic_fetch_or(host_err, err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // the runs are final: everything that wrote them has finished before this kernel (stream order, or the wait above)
    __hip_atomic_store(&ctl[kCtlSortDone], err != 0 ? 2u : 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// A side stream's way to wait for a word another stream's kernel sets (no event on that stream: an event record between two
// kernels of the caller's stream costs ~7 us of its critical path). One wave, polling far apart. A timeout (about a minute: the
// kernel waited for was SUBMITTED before this one, so only a tool that runs one kernel at a time, out of submission order, can
// get here — optim.hip: device_waits_allowed) goes to error_word and to *host_err.
__global__ void split_wait_kernel(const uint32_t* word, uint32_t value, uint32_t* error_word, uint32_t wait_polls, uint32_t* host_err)
{
  unsigned spins = 0;
  // (signed distance: the words of a ring are re-used with growing values, a stale one is "before")
  while (static_cast<int32_t>(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - value) < 0) {
    __builtin_amdgcn_s_sleep(63);
    if (++spins > wait_polls) {
      if (threadIdx.x == 0) {
        atomicOr(error_word, static_cast<uint32_t>(kErrWait));
        if (host_err != nullptr) __hip_atomic_fetch_or(host_err, static_cast<uint32_t>(kErrWait), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      break;
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

// The first two kernels alone, as a PROBE: "would a split sort of this batch overflow a bucket?" — for a caller that has routed
// a run of skewed batches to another sort and wants to know when the batches stop being skewed (optim.hip: run_dedup).
// probe_ws: probe_workspace_bytes() bytes of the caller's; the answer is the word probe_overflow_word() points at.
inline size_t probe_workspace_bytes()
{
  auto align = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  return align(4 * static_cast<size_t>(kMaxTiles) * kMaxPitch) + 2 * align(4 * (kMaxPitch + 2)) + align(4 * kCtlWords) + 256;
}
inline plan probe_plan(plan p)
{
  auto align = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  size_t o = 0;
  p.off_counts = o, o += align(4 * static_cast<size_t>(p.tiles) * p.pitch);
  p.off_totals = o, o += align(4 * static_cast<size_t>(p.pitch));
  p.off_state  = o, o += align(4 * static_cast<size_t>(p.pitch + 2));
  p.off_ctl    = o, o += align(4 * kCtlWords);
  p.total      = o;
  return p;
}
template <typename UKey>
int launch_probe(const plan& full, const UKey* ids, int64_t n, UKey key_lower_bound, uint32_t span, void* probe_ws, hipStream_t stream)
{
  const plan p = probe_plan(full);
  char* ws     = static_cast<char*>(probe_ws);
  key_source<UKey> src{ids, key_lower_bound, span};
  uint32_t* counts = reinterpret_cast<uint32_t*>(ws + p.off_counts);
  uint32_t* ctl    = reinterpret_cast<uint32_t*>(ws + p.off_ctl);
  hipLaunchKernelGGL((split_hist_kernel<UKey>), dim3(tile_grid(p.tiles)), dim3(kBlock), 0, stream, src, n, p.tile, p.tiles, p.shift,
                     p.buckets, p.pitch, counts, ctl, static_cast<uint32_t*>(nullptr), static_cast<int64_t>(0));
  hipLaunchKernelGGL(split_scan_kernel, dim3(p.pitch / 32), dim3(kBlock), 0, stream, counts, p.tiles, p.pitch, p.buckets,
                     reinterpret_cast<uint32_t*>(ws + p.off_totals), ctl, 1 << p.cap_bits, reinterpret_cast<uint32_t*>(ws + p.off_state),
                     p.pitch + 2, static_cast<uint32_t*>(nullptr), 0u);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
inline const uint32_t* probe_overflow_word(const plan& full, void* probe_ws)
{
  return reinterpret_cast<const uint32_t*>(static_cast<char*>(probe_ws) + probe_plan(full).off_ctl) + kCtlOverflow;
}

inline const uint32_t* overflow_word(const plan& p, void* workspace)
{
  return reinterpret_cast<const uint32_t*>(static_cast<char*>(workspace) + p.off_ctl) + kCtlOverflow;
}

}  // namespace split
}  // namespace wm
