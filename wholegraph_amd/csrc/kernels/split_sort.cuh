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
enum { kErrLookBack = 1u, kErrJoin = 2u, kErrWait = 4u, kErrOnesweep = 8u, kErrTasks = 16u };

// ---- hot ids (round 6) ---------------------------------------------------------------------------------------------------
// A skewed batch overflows a stage-2 bucket (the hot id of a Zipf batch alone is 5 % of it) and used to be handed to the generic
// sort. In HOT mode a one-workgroup kernel first samples the batch and names up to kHotMax ids that would fill a good part of a
// bucket by themselves; each of them becomes a SPLIT POINT of stage 1's buckets — the regular bucket B of row range
// [B << shift, (B + 1) << shift) that holds hot ids h0 < h1 < ... is cut into [.., h0) {h0} (h0, h1) {h1} ... — so the split
// buckets stay in ascending key order (unique ids come out sorted as before) and a hot id's occurrences form a bucket of their
// own that is one RUN. Stage 1 places ids in arrival order as before; what has to be in receive order afterwards — the tile
// segments of a hot bucket, and the runs of more than kMaxDup ids of a regular bucket, which no longer send their bucket to
// the radix passes — is listed as (start, length) segments of order[] and sorted by position by split_fix_*_kernel.
constexpr int kHotMax       = 512;   // hot ids, at most
constexpr int kHotPerBucket = 63;    // ... of one regular bucket (6 bits of the per-bucket word)
constexpr uint32_t kInfoHot = 1u << 31;
constexpr int kHotInTile    = 8;    // a hot bucket's share of a tile of up to this many ids is put in order inside stage 1's scatter kernel
constexpr int kFixSmall     = 2048;  // positions of a listed segment a single wave sorts in its LDS (8 KiB); longer: a workgroup
constexpr unsigned long long kTaskTile = 1ull << 63;   // fix task: the segment's positions lie inside ONE tile of stage 1
constexpr int kSelSamples   = 32;    // ids per thread the selection kernel looks at (32768 of the batch; 56: 62 us, mostly load latency)

struct hot_tables {        // device arrays in the sort's workspace, written by hot_select_kernel
  uint32_t* n_hot;         // [0] hot ids H, [1] the sample count an id needed, [2] fix tasks listed so far, [3] unused
  uint32_t* keys;          // [kHotMax] the hot ids, ascending (narrowed keys)
  uint16_t* per_bucket;    // [buckets + 1] hot ids in regular buckets before B (low 10 bits) | hot ids of B << 10
  uint32_t* info;          // [pitch] per split bucket: regular bucket (low 16 bits) | hot slot << 16 | kInfoHot
  unsigned long long* tasks;   // [max_tasks] segments of order[] to be sorted by position: start | length << 32
  uint32_t max_tasks;
};

// split bucket of a key: regular bucket + 2 x (hot ids below the key) + 1 if the key is hot itself
__device__ __forceinline__ uint32_t split_bucket_of(uint32_t key, int shift, const uint16_t* hb, const uint32_t* hot, bool* is_hot)
{
  const uint32_t B = key >> shift;
  const uint32_t v = hb[B];
  uint32_t i       = v & 0x3FFu;
  uint32_t idx     = B + 2u * i;
  *is_hot          = false;
  for (uint32_t nh = v >> 10; nh > 0; nh--, i++) {
    const uint32_t h = hot[i];
    if (key > h) {
      idx += 2;
    } else {
      if (key == h) idx += 1, *is_hot = true;
      break;
    }
  }
  return idx;
}

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
  int hot_max;        // 0, or kHotMax: the pitch has room for 2 x hot_max split buckets more (HOT mode)
  // workspace carve (bytes from the workspace start)
  size_t off_keys, off_pos, off_counts, off_totals, off_starts, off_state, off_ctl, total;
  size_t off_hot_n, off_hot_keys, off_hot_pb, off_info, off_tasks;   // HOT mode only
  uint32_t max_tasks;
};

inline size_t scatter_lds_bytes(int pitch, int ipt, bool hot = false);
inline plan make_plan(int64_t n, int64_t span, int ipt_override = 0, int cap_bits_override = 0, bool hot = false)
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
  // HOT mode: room for two more split buckets per hot id (split_bucket_of); a plan without that room is not taken in HOT mode
  p.hot_max = hot && p.buckets + 2 * kHotMax + 2 <= kMaxPitch && s <= kMapBits ? kHotMax : 0;
  if (hot && p.hot_max == 0) return p;
  p.pitch = (p.buckets + 2 * p.hot_max + 2 + 31) / 32 * 32;
  // tiles: about two rounds of the workgroups the chip holds (two per CU while a tile's LDS stays under 80 KB, else one)
  const bool wide = p.pitch > 2080;
  int ipt = static_cast<int>((n + (wide ? 500 : 1000) * kBlock - 1) / ((wide ? 500 : 1000) * kBlock));
  if (ipt < 4) ipt = 4;
  if (ipt > kMaxIpt) ipt = kMaxIpt;
  if (ipt_override > 0 && ipt_override <= kMaxIpt) ipt = ipt_override;   // experiments
  while (ipt > 4 && scatter_lds_bytes(p.pitch, ipt, p.hot_max > 0) > kLdsBytes - 1024) ipt--;
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
  if (p.hot_max > 0) {
    // hot tables + the list of segments of order[] that have to be sorted by position afterwards: a hot bucket's share of a
    // tile with at least two ids (at most n / 2 of them) and every run of more than kMaxDup ids of a regular bucket (n / 9)
    p.max_tasks    = static_cast<uint32_t>(n / 2 + n / 9 + 64);
    p.off_hot_n    = o, o += align(16);
    p.off_hot_keys = o, o += align(4 * kHotMax);
    p.off_hot_pb   = o, o += align(2 * static_cast<size_t>(p.buckets + 2));
    p.off_info     = o, o += align(4 * static_cast<size_t>(p.pitch));
    p.off_tasks    = o, o += align(8 * static_cast<size_t>(p.max_tasks));
  }
  p.total      = o;
  p.ok         = true;
  return p;
}
inline hot_tables hot_view(const plan& p, void* workspace)
{
  char* ws = static_cast<char*>(workspace);
  hot_tables h{};
  if (p.hot_max == 0) return h;
  h.n_hot      = reinterpret_cast<uint32_t*>(ws + p.off_hot_n);
  h.keys       = reinterpret_cast<uint32_t*>(ws + p.off_hot_keys);
  h.per_bucket = reinterpret_cast<uint16_t*>(ws + p.off_hot_pb);
  h.info       = reinterpret_cast<uint32_t*>(ws + p.off_info);
  h.tasks      = reinterpret_cast<unsigned long long*>(ws + p.off_tasks);
  h.max_tasks  = p.max_tasks;
  return h;
}

// upper bound of plan::total over every span (the workspace is sized before the span is known)
inline size_t workspace_bound(int64_t n)
{
  auto align = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  if (n <= 0) return 256;
  return 2 * align(4 * static_cast<size_t>(n)) + align(4 * static_cast<size_t>(kMaxTiles) * kMaxPitch) +
         3 * align(4 * (kMaxPitch + 2)) + align(4 * kCtlWords) + 256 +
         // HOT mode: tables + task list
         align(16) + align(4 * kHotMax) + align(2 * (kMaxPitch + 2)) + align(4 * kMaxPitch) +
         align(8 * static_cast<size_t>(n / 2 + n / 9 + 64));
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

// ---- stage 0 (HOT mode): which ids get a bucket of their own --------------------------------------------------------------
// One workgroup. 32 k ids of the batch (32 windows of 1024 consecutive ids, evenly spread: every load instruction is one
// contiguous read) are counted in a sketch of 65536 16-bit counters (the whole LDS: a uniform batch puts ~1 sample into a
// counter); a sample whose counter reached the threshold is a candidate and is counted exactly in a 2048-slot table; candidates
// with at least `thr` samples — an estimated cap / 8 ids of the batch, at least 6 samples — are hot, at most kHotMax of them (the
// threshold doubles until they fit) and at most kHotPerBucket per regular bucket. Any choice gives a correct sort: an id that is
// missed stays in its regular bucket (which may then overflow: the generic path, as before), one that is picked needlessly
// costs two empty-ish split buckets. The tables for stages 1 and 2 are written here: hot ids ascending, per regular bucket
// how many hot ids lie before it and in it, per split bucket its regular bucket / hot slot.
template <typename UKey>
__global__ __launch_bounds__(kBlock) void hot_select_kernel(key_source<UKey> src, int64_t n, int shift, int buckets, int pitch, int cap,
                                                            hot_tables ht)
{
  extern __shared__ uint32_t s_sel[];
  constexpr int kCounters = 32768;   // words: two 16-bit counters each
  constexpr int kCand     = 2048;
  uint32_t* s_cnt   = s_sel;                       // [kCounters]
  uint32_t* s_ckey  = s_cnt + kCounters;           // [kCand] candidate ids (0xFFFFFFFF: empty)
  uint32_t* s_ccnt  = s_ckey + kCand;              // [kCand] their exact sample counts
  uint32_t* s_pick  = s_ccnt + kCand;              // [kHotMax] picked ids, any order
  uint32_t* s_sort  = s_pick + kHotMax;            // [kHotMax] ... ascending
  uint32_t* s_keep  = s_sort + kHotMax;            // [kHotMax] ... after the per-bucket limit
  __shared__ uint32_t s_waves[kWaves];
  __shared__ uint32_t s_n[4];
  const int tid = threadIdx.x;
  for (int i = tid; i < kCounters; i += kBlock) s_cnt[i] = 0;
  for (int i = tid; i < kCand; i += kBlock) s_ckey[i] = 0xFFFFFFFFu, s_ccnt[i] = 0;
  if (tid < 4) s_n[tid] = 0;
  // samples: window w of 1024 ids starts at base_w; thread t reads id base_w + t
  const bool spread     = n >= static_cast<int64_t>(kSelSamples) * kBlock;
  const int64_t gap     = spread ? (n - kBlock) / (kSelSamples - 1) : kBlock;
  uint32_t smp[kSelSamples];
  constexpr int kSelBatch = 16;   // loads in flight per thread
  static_assert(kSelSamples % kSelBatch == 0, "whole batches");
#pragma unroll
  for (int w0 = 0; w0 < kSelSamples; w0 += kSelBatch) {
    UKey raw[kSelBatch];
#pragma unroll
    for (int u = 0; u < kSelBatch; u++) {
      const int64_t pos = static_cast<int64_t>(w0 + u) * gap + tid;
      raw[u]            = src.ids[pos < n ? pos : n - 1];
    }
#pragma unroll
    for (int u = 0; u < kSelBatch; u++) {
      const int64_t pos = static_cast<int64_t>(w0 + u) * gap + tid;
      const uint32_t k  = src.narrow(raw[u]);
      smp[w0 + u]       = (pos < n && k < src.span) ? k : 0xFFFFFFFFu;   // (ids outside the range are the drop bucket's)
    }
  }
  __syncthreads();
  const int64_t sampled = spread ? static_cast<int64_t>(kSelSamples) * kBlock : n;
#pragma unroll
  for (int j = 0; j < kSelSamples; j++)
    if (smp[j] != 0xFFFFFFFFu) {
      const uint32_t h = (smp[j] * 2654435761u) >> 16;
      atomicAdd(&s_cnt[h >> 1], 1u << ((h & 1u) * 16));
    }
  __syncthreads();
  // an id is worth a bucket of its own from about an eighth of a stage-2 bucket: that many ids of the batch, in samples
  uint32_t thr = static_cast<uint32_t>((static_cast<int64_t>(cap / 8) * sampled + n - 1) / n);
  if (thr < 6) thr = 6;
#pragma unroll
  for (int j = 0; j < kSelSamples; j++)
    if (smp[j] != 0xFFFFFFFFu) {
      const uint32_t h = (smp[j] * 2654435761u) >> 16;
      if (((s_cnt[h >> 1] >> ((h & 1u) * 16)) & 0xFFFFu) >= thr) {
        uint32_t slot = (smp[j] * 0x9E3779B1u) >> 21;   // 11 bits
        for (int probe = 0; probe < 64; probe++, slot = (slot + 1) & (kCand - 1)) {
          const uint32_t old = atomicCAS(&s_ckey[slot], 0xFFFFFFFFu, smp[j]);
          if (old == 0xFFFFFFFFu || old == smp[j]) {
            atomicAdd(&s_ccnt[slot], 1u);
            break;
          }
        }
      }
    }
  __syncthreads();
  // how many candidates reach the threshold; double it until they fit
  for (int round = 0; round < 12; round++) {
    uint32_t mine = 0;
    for (int i = tid; i < kCand; i += kBlock) mine += s_ccnt[i] >= thr ? 1u : 0u;
    uint32_t total;
    (void)block_exclusive_sum(mine, s_waves, &total);
    if (total <= static_cast<uint32_t>(kHotMax)) break;
    thr *= 2;
    __syncthreads();
  }
  __syncthreads();
  for (int i = tid; i < kCand; i += kBlock)
    if (s_ccnt[i] >= thr) {
      const uint32_t at = atomicAdd(&s_n[0], 1u);
      if (at < static_cast<uint32_t>(kHotMax)) s_pick[at] = s_ckey[i];
    }
  __syncthreads();
  const int picked = static_cast<int>(s_n[0] < static_cast<uint32_t>(kHotMax) ? s_n[0] : kHotMax);
  // ascending: rank by counting (ids are distinct)
  if (tid < picked) {
    const uint32_t x = s_pick[tid];
    int r            = 0;
    for (int j = 0; j < picked; j++) r += s_pick[j] < x ? 1 : 0;
    s_sort[r] = x;
  }
  __syncthreads();
  // at most kHotPerBucket per regular bucket: an id keeps its place when fewer than that many picked ids of its bucket precede it
  uint32_t keep = 0;
  if (tid < picked) {
    const uint32_t B = s_sort[tid] >> shift;
    int before       = 0;
    for (int j = tid - 1; j >= 0 && (s_sort[j] >> shift) == B; j--) before++;
    keep = before < kHotPerBucket ? 1u : 0u;
  }
  uint32_t H;
  const uint32_t at = block_exclusive_sum(keep, s_waves, &H);
  if (keep) s_keep[at] = s_sort[tid];
  __syncthreads();
  if (tid < static_cast<int>(H)) ht.keys[tid] = s_keep[tid];
  if (tid == 0) ht.n_hot[0] = H, ht.n_hot[1] = thr, ht.n_hot[2] = 0, ht.n_hot[3] = 0;
  // per regular bucket: hot ids before it and in it (lower bounds in the ascending list)
  auto lower = [&](uint32_t key_lo) {   // first hot id >= key_lo
    int lo = 0, hi = static_cast<int>(H);
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (s_keep[mid] < key_lo) lo = mid + 1;
      else hi = mid;
    }
    return lo;
  };
  for (int B = tid; B <= buckets; B += kBlock) {
    const int first = B < buckets ? lower(static_cast<uint32_t>(B) << shift) : static_cast<int>(H);
    const int next  = B + 1 < buckets ? lower(static_cast<uint32_t>(B + 1) << shift) : static_cast<int>(H);
    const int nh    = B < buckets ? next - first : 0;
    ht.per_bucket[B] = static_cast<uint16_t>(first | (nh << 10));
    if (B < buckets) {
      const int base = B + 2 * first;
      ht.info[base]  = static_cast<uint32_t>(B);
      for (int i = 0; i < nh; i++) {
        ht.info[base + 2 * i + 1] = kInfoHot | (static_cast<uint32_t>(first + i) << 16) | static_cast<uint32_t>(B);
        ht.info[base + 2 * i + 2] = static_cast<uint32_t>(B);
      }
    }
  }
  // (split buckets from buckets + 2 H on — the drop bucket and the unused columns — are never looked up in info[])
}
inline size_t hot_select_lds_bytes() { return 4 * static_cast<size_t>(32768 + 2 * 2048 + 3 * kHotMax); }

// ---- stage 1a: per-tile bucket histogram -------------------------------------------------------------------------------
template <typename UKey, bool HOT = false>
__global__ __launch_bounds__(kBlock) void split_hist_kernel(key_source<UKey> src, int64_t n, int tile, int tiles, int shift,
                                                            int buckets, int pitch, uint32_t* counts, uint32_t* ctl,
                                                            uint32_t* zero_words, int64_t n_zero_words, hot_tables ht = hot_tables{})
{
  __shared__ uint32_t s_hist[kMaxPitch];
  __shared__ uint16_t s_hb[HOT ? kMaxBuckets + 2 : 2];
  __shared__ uint32_t s_hot[HOT ? kHotMax : 1];
  int drop = buckets;   // index of the drop bucket
  if constexpr (HOT) {
    const int H = static_cast<int>(ht.n_hot[0]);
    drop        = buckets + 2 * H;
    for (int i = threadIdx.x; i <= buckets; i += kBlock) s_hb[i] = ht.per_bucket[i];
    for (int i = threadIdx.x; i < H; i += kBlock) s_hot[i] = ht.keys[i];
  }
  auto column = [&](uint32_t k) -> int {
    if (k >= src.span) return drop;
    if constexpr (HOT) {
      bool hot;
      return static_cast<int>(split_bucket_of(k, shift, s_hb, s_hot, &hot));
    } else {
      return static_cast<int>(k >> shift);
    }
  };
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
    for (int u = 0; u < U; u++) atomicAdd(&s_hist[column(k[u])], 1u);
  }
  for (; i < end; i += kBlock) {
    const uint32_t k = src.key(i);
    atomicAdd(&s_hist[column(k)], 1u);
  }
  __syncthreads();
  uint32_t* row = counts + static_cast<size_t>(t) * pitch;
  for (int j = threadIdx.x; j < pitch; j += kBlock) row[j] = s_hist[j];
}

// ---- stage 1b: column-wise exclusive prefix of the counts matrix --------------------------------------------------------
// one workgroup = 32 columns x all tiles: thread (row group rg, column c) owns rows_per consecutive tiles of its column
__global__ __launch_bounds__(kBlock) void split_scan_kernel(uint32_t* counts, int tiles, int pitch, int buckets, uint32_t* totals,
                                                            uint32_t* ctl, int cap, uint32_t* state, int state_words,
                                                            uint32_t* verdict_word, uint32_t verdict_value,
                                                            hot_tables ht = hot_tables{})
{
  // HOT mode: `buckets` regular + 2 H split buckets; a hot id's own bucket may hold any number of ids (it is one run whose
  // positions stage 1 writes straight into order[])
  const int real = ht.n_hot != nullptr ? buckets + 2 * static_cast<int>(ht.n_hot[0]) : buckets;
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
    if (col < real && total > static_cast<uint32_t>(cap) && !(ht.n_hot != nullptr && (ht.info[col] & kInfoHot) != 0))
      atomicOr(&ctl[kCtlOverflow], 1u);
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
inline size_t scatter_lds_bytes(int pitch, int ipt, bool hot)
{
  // global segment starts (4 B per bucket) + tile-local segment starts and slot counters (2 B each per bucket) + the tile's
  // keys (4 B) and indices in the tile (2 B) [+ HOT: the per-bucket words (2 B) and the hot ids]
  return 4 * static_cast<size_t>(pitch) + 2 * 2 * static_cast<size_t>(pitch + 2) + 6 * static_cast<size_t>(ipt) * kBlock + 16 +
         (hot ? 2 * static_cast<size_t>(pitch + 2) + 4 * static_cast<size_t>(kHotMax) + 16 : 0);
}

// Multisplit of one tile: every id gets a slot in its bucket's segment of the tile from ONE LDS counter per bucket, in
// whatever order the lanes arrive — so a bucket's share of a tile is NOT in receive order. It does not have to be: stage 2
// orders equal ids by their POSITION (carried beside every key), not by where stage 1 left them, and nothing else in a bucket
// depends on the order inside a tile's segment. (The first version ranked every id stably among its bucket's ids of the tile —
// per-wave counters, their prefix over the waves, a lane-order fix per step: 31 us per tile of 10 k ids, profiles/
// r05_split_sort_harness.txt history; this is one atomic per id and two barriers: ~12 us.)
// PER: buckets per thread in the per-bucket loops (3 up to 3072 buckets + pitch slack, 5 up to kMaxBuckets)
// HOT: split buckets by split_bucket_of; the ids of a hot bucket (one run) go straight into order[] — in arrival order like
// everything else here; a tile's share of a hot bucket that holds two ids or more is listed for split_fix_*_kernel
template <typename UKey, int MAXIPT, int PER, bool HOT = false>
__global__ __launch_bounds__(kBlock, MAXIPT <= 12 ? 8 : 4) void split_scatter_kernel(key_source<UKey> src, int64_t n, int ipt, int tiles,
                                                                                   int shift, int buckets, int bucket_bits, int pitch,
                                                                                   const uint32_t* counts, const uint32_t* totals,
                                                                                   uint32_t* bucket_start, uint32_t* keys_out,
                                                                                   uint32_t* pos_out, const uint32_t* ctl,
                                                                                   hot_tables ht = hot_tables{}, int32_t* order = nullptr)
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
  uint32_t* s_hot  = reinterpret_cast<uint32_t*>(s_idx + tile);             // HOT: [kHotMax] the hot ids
  uint16_t* s_hb   = reinterpret_cast<uint16_t*>(s_hot + kHotMax);          // HOT: [buckets + 1] per regular bucket: hot ids before / in it
  __shared__ uint32_t s_waves[kWaves];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  WM_SPLIT_T(0, blockIdx.x, 0);
  int drop = buckets;   // index of the drop bucket
  if constexpr (HOT) {
    const int H = static_cast<int>(ht.n_hot[0]);
    drop        = buckets + 2 * H;
    for (int i = threadIdx.x; i <= buckets; i += kBlock) s_hb[i] = ht.per_bucket[i];
    for (int i = threadIdx.x; i < H; i += kBlock) s_hot[i] = ht.keys[i];
  }
  auto column = [&](uint32_t k, bool* hot) -> uint32_t {
    *hot = false;
    if (k >= src.span) return static_cast<uint32_t>(drop);
    if constexpr (HOT) return split_bucket_of(k, shift, s_hb, s_hot, hot);
    else return k >> shift;
  };

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
    unsigned long long task[HOT ? PER : 1];   // HOT: segments this thread lists (one slot of the global list per WORKGROUP and
    uint32_t n_task = 0;                      // round trip: 200 k single appends to one counter took a millisecond)
#pragma unroll
    for (int i = 0; i < PER; i++)
      if (i < per && b0 + i < pitch) {
        s_off[b0 + i] = run + row[i] - lstart;
        s_lst[b0 + i] = static_cast<uint16_t>(lstart);
        if (t == 0) bucket_start[b0 + i] = run;   // [buckets] = number of ids inside the range, [buckets + 1] = n
        if constexpr (HOT) {
          // this tile's share of a hot bucket arrives in any order: two ids or more -> to be sorted by position afterwards
          // (up to kHotInTile ids: put in order right here, in LDS, behind the placement — see below; more: listed, flagged as
          // a TILE segment — its positions lie inside one tile, split_fix_small_kernel ranks them with a bitmap of the tile)
          if (cnt[i] > static_cast<uint32_t>(kHotInTile) && b0 + i < drop && (ht.info[b0 + i] & kInfoHot) != 0)
            task[n_task++] = static_cast<unsigned long long>(run + row[i]) | (static_cast<unsigned long long>(cnt[i]) << 32) | kTaskTile;
        }
        run += tt[i];
        lstart += cnt[i];
      }
    uint32_t* one32 = reinterpret_cast<uint32_t*>(s_one);
    for (int i = threadIdx.x; i < (pitch + 2) / 2; i += kBlock) one32[i] = 0;
    if constexpr (HOT) {
      __shared__ uint32_t s_task_base;
      uint32_t all;
      const uint32_t at = block_exclusive_sum(n_task, s_waves, &all);
      if (threadIdx.x == 0) s_task_base = all > 0 ? atomicAdd(&ht.n_hot[2], all) : 0u;
      __syncthreads();
      for (uint32_t q = 0; q < n_task; q++)
        if (s_task_base + at + q < ht.max_tasks) ht.tasks[s_task_base + at + q] = task[q];
    }
  }
  __syncthreads();
  WM_SPLIT_T(0, blockIdx.x, 1);
  {
    uint32_t* one32 = reinterpret_cast<uint32_t*>(s_one);
#pragma unroll
    for (int j = 0; j < MAXIPT; j++)
      if (j < ipt && local0 + j * 64 < valid_n) {
        bool hot_id;
        const uint32_t b  = column(key[j], &hot_id);
        const int sh      = (b & 1u) * 16;
        const uint32_t o  = (atomicAdd(&one32[b >> 1], 1u << sh) >> sh) & 0xFFFFu;
        const uint32_t lp = s_lst[b] + o;
        s_keys[lp]        = key[j];
        s_idx[lp]         = static_cast<uint16_t>(local0 + j * 64);
      }
  }
  __syncthreads();
  if constexpr (HOT) {
    // a hot bucket's share of this tile, 2 .. kHotInTile ids: into receive order now (the ids are equal: only the 16-bit
    // indices in the tile move), by the thread that owns the bucket
    for (int q = 0; q < per; q++) {
      const int b = b0 + q;
      if (b >= drop || b >= pitch) break;
      const int lo = s_lst[b], c = (b + 1 < pitch ? s_lst[b + 1] : valid_n) - lo;   // (segments are contiguous in bucket order)
      if (c < 2 || c > kHotInTile || (ht.info[b] & kInfoHot) == 0) continue;
      for (int i = lo + 1; i < lo + c; i++) {
        const uint16_t x = s_idx[i];
        int j            = i;
        while (j > lo && s_idx[j - 1] > x) s_idx[j] = s_idx[j - 1], j--;
        s_idx[j] = x;
      }
    }
    __syncthreads();   // (the write-out below reads what was sorted)
  }
  WM_SPLIT_T(0, blockIdx.x, 2);
#pragma unroll
  for (int k = 0; k < MAXIPT; k++) {
    const int sl = k * kBlock + threadIdx.x;
    if (k < ipt && sl < valid_n && !WM_SPLIT_DBG(1)) {
      const uint32_t x  = s_keys[sl];
      bool hot_id;
      const uint32_t gp = s_off[column(x, &hot_id)] + static_cast<uint32_t>(sl);
      if (HOT && hot_id) {
        order[gp] = static_cast<int32_t>(static_cast<uint32_t>(base) + s_idx[sl]);
      } else {
        keys_out[gp] = x;
        pos_out[gp]  = static_cast<uint32_t>(base) + s_idx[sl];
      }
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
// HOT: the workgroup index is a SPLIT bucket (split_bucket_of): a hot id's bucket is one run whose positions stage 1 has written
// into order[] already — it only takes its place in the chain of run counts —, a regular one is ordered as before except that a
// run of more than kMaxDup ids no longer sends the bucket to the radix passes: it is left in arrival order and listed for
// split_fix_*_kernel.
template <typename OutT, int CAPBITS, bool HOT = false>
__global__ __launch_bounds__((1 << CAPBITS) / kSortIpt, 8) void split_sort_kernel(const uint32_t* keys, const uint32_t* pos,
                                                               const uint32_t* bucket_start, int buckets, int shift, int passes,
                                                               int digit_bits, int pos_passes, int pos_digit_bits, OutT key_base, OutT* unique_ids, int32_t* run_starts,
                                                               int32_t* order, int64_t* n_unique, uint32_t* ctl, uint32_t* state,
                                                               wait_cfg wc, hot_tables ht = hot_tables{})
{
  if (ctl[kCtlOverflow] != 0) return;
  int drop = buckets;   // index of the drop bucket = number of real split buckets
  if constexpr (HOT) {
    drop = buckets + 2 * static_cast<int>(ht.n_hot[0]);
    if (static_cast<int>(blockIdx.x) > drop) return;
  }
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

  if (b == drop) {
    // the drop bucket: positions of the ids outside the range fill the tail of order[]; then the totals
    for (int i = threadIdx.x; i < m; i += BLOCK) order[start + i] = static_cast<int32_t>(pos[start + i]);
    if (wv == 0) {
      const uint32_t total = look_back(state, drop, ctl, wc);
      if (lane == 0) {
        *n_unique         = static_cast<int64_t>(total);
        run_starts[total] = static_cast<int32_t>(start);
      }
    }
    return;
  }
  uint32_t regular = static_cast<uint32_t>(b);   // the regular bucket (key >> shift) of this split bucket
  if constexpr (HOT) {
    const uint32_t inf = ht.info[b];
    regular            = inf & 0xFFFFu;
    if ((inf & kInfoHot) != 0 && m > 0) {
      // a hot id: one run, its positions are in order[start, start + m) already (receive order after split_fix_*_kernel)
      if (wv == 0) {
        if (lane == 0) publish(wc, state, b, b == 0 ? kFlagPrefix : kFlagAggregate, 1u);
        const uint32_t excl = look_back(state, b, ctl, wc);
        if (lane == 0) {
          if (b > 0) publish(wc, state, b, kFlagPrefix, excl + 1u);
          run_starts[excl] = static_cast<int32_t>(start);
          unique_ids[excl] = key_base + static_cast<OutT>(ht.keys[(inf >> 16) & 0x3FFu]);
          // ("a run of more than kMaxDup ids exists": what the optimizer step's listing kernels look at, optim.hip)
          if (m > kMaxDup) atomicAdd(&ctl[kCtlRadixBuckets], 1u);
        }
      }
      return;
    }
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
  const OutT bucket_key = (static_cast<OutT>(regular) << shift) + key_base;
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
        if constexpr (!HOT) long_run |= o >= static_cast<uint32_t>(kMaxDup);   // (HOT: such a run is put in order afterwards)
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
      if constexpr (HOT) {
        // long runs (more than kMaxDup ids) are left as they arrived; their pieces of order[] are listed for split_fix_*_kernel —
        // one slot of the global list per workgroup and round trip, the entries written side by side
        uint32_t mine = 0;
        bool any_big  = false;
        for (int r = threadIdx.x; r < static_cast<int>(heads_total); r += BLOCK) {
          const int i0 = static_cast<int>(s_run[r]);
          const int i1 = r + 1 < static_cast<int>(heads_total) ? static_cast<int>(s_run[r + 1]) : m;
          mine += i1 - i0 > kMaxDup && i1 - i0 <= kFixSmall ? 1u : 0u;
          if (i1 - i0 > kFixSmall) {   // (an id the selection missed: rare — its own list, from the END of the array, for split_fix_big_kernel)
            const uint32_t k = atomicAdd(&ht.n_hot[3], 1u);
            if (k < ht.max_tasks)
              ht.tasks[ht.max_tasks - 1 - k] = static_cast<unsigned long long>(start + static_cast<uint32_t>(i0)) | (static_cast<unsigned long long>(i1 - i0) << 32);
            any_big = true;
          }
        }
        uint32_t all;
        uint32_t at = block_exclusive_sum<WAVES>(mine, s_waves, &all);
        if (any_big) atomicAdd(&ctl[kCtlRadixBuckets], 1u);   // "a run of more than kMaxDup ids exists" (the optimizer step's listing kernels)
        if (all > 0) {   // (uniform)
          if (threadIdx.x == 0) {
            s_misc[3] = atomicAdd(&ht.n_hot[2], all);
            atomicAdd(&ctl[kCtlRadixBuckets], 1u);
          }
          __syncthreads();
          at += s_misc[3];
          for (int r = threadIdx.x; r < static_cast<int>(heads_total); r += BLOCK) {
            const int i0 = static_cast<int>(s_run[r]);
            const int i1 = r + 1 < static_cast<int>(heads_total) ? static_cast<int>(s_run[r + 1]) : m;
            if (i1 - i0 > kMaxDup && i1 - i0 <= kFixSmall) {
              if (at < ht.max_tasks)
                ht.tasks[at] = static_cast<unsigned long long>(start + static_cast<uint32_t>(i0)) | (static_cast<unsigned long long>(i1 - i0) << 32);
              at++;
            }
          }
        }
      }
      for (int r = threadIdx.x; r < static_cast<int>(heads_total); r += BLOCK) {
        const int i0 = static_cast<int>(s_run[r]);
        const int i1 = r + 1 < static_cast<int>(heads_total) ? static_cast<int>(s_run[r + 1]) : m;
        if (HOT && i1 - i0 > kMaxDup) continue;
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

// ---- HOT mode: the listed segments of order[] into receive order = ascending position ------------------------------------------
// (positions are distinct and < 2^31.) Segments of up to 64 positions: one wave ranks them by counting, out of registers;
// up to 1024: a bitonic sort in the wave's 4 KiB of LDS; longer ones (a tile's share of a very hot id, up to the tile; a regular
// bucket's run of up to 8192 ids) take a workgroup each, split_fix_big_kernel. Both kernels walk the whole list and take what
// is theirs by length; they return at once when the batch overflowed (the generic path writes order[]) .
constexpr int kFixMapWords = kMaxIpt * kBlock / 32;   // bitmap of a tile: 768 words
// Listed segments of order[] into ascending position, one WAVE per segment:
//  * a TILE segment (a hot bucket's share of one tile, any length): its positions lie in [t x tile, (t + 1) x tile) — a bitmap of the
//    tile, its prefix popcounts, and the sorted segment is written back straight from the set bits: O(length + tile / 32), no compare;
//  * a run of a regular bucket, up to 64 positions: ranked by counting, out of registers; up to 1024: bitonic in the wave's LDS;
//    longer (an id the selection missed): split_fix_big_kernel.
__global__ __launch_bounds__(256) void split_fix_small_kernel(const hot_tables ht, int32_t* order, uint32_t* ctl, int tile)
{
  if (ctl[kCtlOverflow] != 0) return;
#ifdef WM_SPLIT_DEBUG
  const int only = g_split_debug;   // experiments: 11 = tile segments only, 12 = runs of up to 64 only, 13 = bitonic runs only (results wrong)
#else
  constexpr int only = 0;
#endif
  __shared__ uint32_t s_buf[4][kFixSmall];   // per wave: the bitmap of a tile (3 KiB), or the 8 KiB of a bitonic sort
  static_assert(kFixSmall >= kFixMapWords, "the bitmap fits");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t n_tasks = ht.n_hot[2];
  if (n_tasks + ht.n_hot[3] > ht.max_tasks) {   // (cannot happen: the list is sized for every segment a batch can produce)
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&ctl[kCtlError], static_cast<uint32_t>(kErrTasks));
    n_tasks = ht.max_tasks;
  }
  auto wave_sync = []() {   // one wave: its LDS operations complete in order; this keeps the compiler's order too
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  for (uint32_t t = blockIdx.x * 4 + wv; t < n_tasks; t += gridDim.x * 4) {
    const unsigned long long task = ht.tasks[t];
    const uint32_t start = static_cast<uint32_t>(task), len = static_cast<uint32_t>(task >> 32) & 0x7FFFFFFFu;
    if ((task & kTaskTile) != 0) {
      if (only != 0 && only != 11) continue;
      uint32_t* map = s_buf[wv];
      const int words = (tile + 31) >> 5;   // <= kFixMapWords
      const uint32_t base = static_cast<uint32_t>(order[start]) / static_cast<uint32_t>(tile) * static_cast<uint32_t>(tile);
      for (int w = lane; w < words; w += 64) map[w] = 0;
      wave_sync();
      for (uint32_t i0 = 0; i0 < len; i0 += 64 * 8) {   // eight loads in flight per lane
        uint32_t x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const uint32_t i = i0 + u * 64 + lane;
          x[u]             = static_cast<uint32_t>(order[start + (i < len ? i : len - 1)]);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const uint32_t off = x[u] - base;   // (a repeated last element sets the same bit again)
          atomicOr(&map[off >> 5], 1u << (off & 31u));
        }
      }
      wave_sync();
      // lane l owns words [l x wpl, (l + 1) x wpl): popcounts, exclusive scan over the lanes, then its set bits in ascending order
      const int wpl = (words + 63) / 64;   // <= 12
      uint32_t mine = 0;
      for (int q = 0; q < wpl; q++) {
        const int w = lane * wpl + q;
        mine += w < words ? static_cast<uint32_t>(__popc(map[w])) : 0u;
      }
      uint32_t incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
      }
      uint32_t at = incl - mine;
      for (int q = 0; q < wpl; q++) {
        const int w = lane * wpl + q;
        uint32_t bits = w < words ? map[w] : 0u;
        while (bits != 0) {
          const int bit = __ffs(static_cast<int>(bits)) - 1;
          bits &= bits - 1;
          order[start + at++] = static_cast<int32_t>(base + static_cast<uint32_t>(w) * 32u + static_cast<uint32_t>(bit));
        }
      }
      wave_sync();
      continue;
    }
    if (len > static_cast<uint32_t>(kFixSmall)) continue;
    if (only != 0 && only != (len <= 64 ? 12 : 13)) continue;
    if (len <= 64) {
      const int32_t x = lane < static_cast<int>(len) ? order[start + lane] : 0x7FFFFFFF;
      int rank        = 0;
      for (int j = 0; j < static_cast<int>(len); j++) rank += __builtin_amdgcn_readlane(x, j) < x ? 1 : 0;
      if (lane < static_cast<int>(len)) order[start + rank] = x;
      continue;
    }
    int32_t* seg = reinterpret_cast<int32_t*>(s_buf[wv]);
    int P = 128;
    while (P < static_cast<int>(len)) P <<= 1;
    for (int i = lane; i < P; i += 64) seg[i] = i < static_cast<int>(len) ? order[start + i] : 0x7FFFFFFF;
    for (int k = 2; k <= P; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        wave_sync();
        for (int i = lane; i < P / 2; i += 64) {
          const int l = ((i & ~(j - 1)) << 1) | (i & (j - 1)), r = l + j;   // (j is a power of two)
          const int32_t a = seg[l], b = seg[r];
          const bool up = (l & k) == 0;
          if ((a > b) == up) seg[l] = b, seg[r] = a;
        }
      }
    wave_sync();
    for (int i = lane; i < static_cast<int>(len); i += 64) order[start + i] = seg[i];
    wave_sync();   // the next task reuses the buffer
  }
}
constexpr int kFixBigMax = 32768;   // positions a workgroup sorts in LDS (128 KiB): a tile holds at most kMaxIpt x 1024 = 24576 ids
__global__ __launch_bounds__(kBlock) void split_fix_big_kernel(const hot_tables ht, int32_t* order, uint32_t* ctl)
{
  if (ctl[kCtlOverflow] != 0) return;
  extern __shared__ int32_t s_big[];
  uint32_t n_tasks = ht.n_hot[3];   // (the big segments are listed from the END of the array: this kernel does not walk the small ones)
  if (n_tasks > ht.max_tasks) n_tasks = ht.max_tasks;
  for (uint32_t t = blockIdx.x; t < n_tasks; t += gridDim.x) {
    const unsigned long long task = ht.tasks[ht.max_tasks - 1 - t];
    const uint32_t start = static_cast<uint32_t>(task), len = static_cast<uint32_t>(task >> 32) & 0x7FFFFFFFu;
    if (len <= static_cast<uint32_t>(kFixSmall)) continue;
    if (len > static_cast<uint32_t>(kFixBigMax)) {   // (no such segment exists)
      if (threadIdx.x == 0) atomicOr(&ctl[kCtlError], static_cast<uint32_t>(kErrTasks));
      continue;
    }
    int P = 4096;
    while (P < static_cast<int>(len)) P <<= 1;
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += kBlock) s_big[i] = i < static_cast<int>(len) ? order[start + i] : 0x7FFFFFFF;
    for (int k = 2; k <= P; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        __syncthreads();
        for (int i = threadIdx.x; i < P / 2; i += kBlock) {
          const int l = ((i & ~(j - 1)) << 1) | (i & (j - 1)), r = l + j;   // (j is a power of two)
          const int32_t a = s_big[l], b = s_big[r];
          const bool up = (l & k) == 0;
          if ((a > b) == up) s_big[l] = b, s_big[r] = a;
        }
      }
    __syncthreads();
    for (int i = threadIdx.x; i < static_cast<int>(len); i += kBlock) order[start + i] = s_big[i];
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

// HOT mode (plan made with hot = true): selection, the three stage-1 kernels with the split points, stage 2, the two fix kernels.
// Same hooks and outputs as launch(); the overflow word still decides on the device whether the generic path runs instead (ids
// clustered in a few thousand rows overflow a regular bucket whatever is peeled).
template <typename UKey, typename Hook = no_hook>
int launch_hot(const plan& p, const UKey* ids, int64_t n, UKey key_lower_bound, uint32_t span, void* unique_ids, int32_t* run_starts,
               int32_t* order, int64_t* n_unique, void* workspace, uint32_t* zero_words, int64_t n_zero_words, hipStream_t stream,
               Hook between = Hook(), uint32_t* verdict_word = nullptr, uint32_t verdict_value = 0, const wait_cfg& wc = wait_cfg())
{
  if (p.hot_max == 0) return -2;
  char* ws         = static_cast<char*>(workspace);
  uint32_t* keys   = reinterpret_cast<uint32_t*>(ws + p.off_keys);
  uint32_t* pos    = reinterpret_cast<uint32_t*>(ws + p.off_pos);
  uint32_t* counts = reinterpret_cast<uint32_t*>(ws + p.off_counts);
  uint32_t* totals = reinterpret_cast<uint32_t*>(ws + p.off_totals);
  uint32_t* starts = reinterpret_cast<uint32_t*>(ws + p.off_starts);
  uint32_t* state  = reinterpret_cast<uint32_t*>(ws + p.off_state);
  uint32_t* ctl    = reinterpret_cast<uint32_t*>(ws + p.off_ctl);
  const hot_tables ht = hot_view(p, workspace);
  key_source<UKey> src{ids, key_lower_bound, span};
  static bool attr_set = [] {
    const int most = static_cast<int>(kLdsBytes - 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&split_scatter_kernel<UKey, kMaxIpt, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, most);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&split_scatter_kernel<UKey, kMaxIpt, 5, true>), hipFuncAttributeMaxDynamicSharedMemorySize, most);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hot_select_kernel<UKey>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(hot_select_lds_bytes()));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&split_fix_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kFixBigMax);
    return true;
  }();
  (void)attr_set;
  const int grid = tile_grid(p.tiles);
  hipLaunchKernelGGL((hot_select_kernel<UKey>), dim3(1), dim3(kBlock), hot_select_lds_bytes(), stream, src, n, p.shift, p.buckets, p.pitch,
                     1 << p.cap_bits, ht);
  hipLaunchKernelGGL((split_hist_kernel<UKey, true>), dim3(grid), dim3(kBlock), 0, stream, src, n, p.tile, p.tiles, p.shift, p.buckets,
                     p.pitch, counts, ctl, zero_words, n_zero_words, ht);
  hipLaunchKernelGGL(split_scan_kernel, dim3(p.pitch / 32), dim3(kBlock), 0, stream, counts, p.tiles, p.pitch, p.buckets, totals, ctl,
                     1 << p.cap_bits, state, p.pitch + 2, verdict_word, verdict_value, ht);
  between();
  const size_t lds = scatter_lds_bytes(p.pitch, p.ipt, true);
  if (p.pitch > 3 * kBlock)
    hipLaunchKernelGGL((split_scatter_kernel<UKey, kMaxIpt, 5, true>), dim3(grid), dim3(kBlock), lds, stream, src, n, p.ipt, p.tiles, p.shift,
                       p.buckets, p.bucket_bits, p.pitch, counts, totals, starts, keys, pos, ctl, ht, order);
  else
    hipLaunchKernelGGL((split_scatter_kernel<UKey, kMaxIpt, 3, true>), dim3(grid), dim3(kBlock), lds, stream, src, n, p.ipt, p.tiles, p.shift,
                       p.buckets, p.bucket_bits, p.pitch, counts, totals, starts, keys, pos, ctl, ht, order);
  const int sort_grid = p.buckets + 2 * p.hot_max + 1;
  if (p.cap_bits == kCapBitsSmall)
    hipLaunchKernelGGL((split_sort_kernel<UKey, kCapBitsSmall, true>), dim3(sort_grid), dim3((1 << kCapBitsSmall) / kSortIpt), 0, stream,
                       keys, pos, starts, p.buckets, p.shift, p.passes, p.digit_bits, p.pos_passes, p.pos_digit_bits, key_lower_bound,
                       static_cast<UKey*>(unique_ids), run_starts, order, n_unique, ctl, state, wc, ht);
  else
    hipLaunchKernelGGL((split_sort_kernel<UKey, kCapBitsBig, true>), dim3(sort_grid), dim3((1 << kCapBitsBig) / kSortIpt), 0, stream,
                       keys, pos, starts, p.buckets, p.shift, p.passes, p.digit_bits, p.pos_passes, p.pos_digit_bits, key_lower_bound,
                       static_cast<UKey*>(unique_ids), run_starts, order, n_unique, ctl, state, wc, ht);
  hipLaunchKernelGGL(split_fix_small_kernel, dim3(2048), dim3(256), 0, stream, ht, order, ctl, p.tile);
  hipLaunchKernelGGL(split_fix_big_kernel, dim3(256), dim3(kBlock), 4 * kFixBigMax, stream, ht, order, ctl);
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
  return align(4 * static_cast<size_t>(kMaxTiles) * kMaxPitch) + 2 * align(4 * (kMaxPitch + 2)) + align(4 * kCtlWords) + 256 +
         align(16) + align(4 * kHotMax) + align(2 * (kMaxPitch + 2)) + align(4 * kMaxPitch);   // (HOT: the tables, no task list)
}
inline plan probe_plan(plan p)
{
  auto align = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  size_t o = 0;
  p.off_counts = o, o += align(4 * static_cast<size_t>(p.tiles) * p.pitch);
  p.off_totals = o, o += align(4 * static_cast<size_t>(p.pitch));
  p.off_state  = o, o += align(4 * static_cast<size_t>(p.pitch + 2));
  p.off_ctl    = o, o += align(4 * kCtlWords);
  if (p.hot_max > 0) {
    p.max_tasks    = 0;
    p.off_hot_n    = o, o += align(16);
    p.off_hot_keys = o, o += align(4 * kHotMax);
    p.off_hot_pb   = o, o += align(2 * static_cast<size_t>(p.buckets + 2));
    p.off_info     = o, o += align(4 * static_cast<size_t>(p.pitch));
    p.off_tasks    = o;
  }
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
  if (p.hot_max > 0) {   // a hot-mode plan: the verdict of a sort with the batch's hot ids peeled
    const hot_tables ht = hot_view(p, probe_ws);
    static bool attr_set = [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hot_select_kernel<UKey>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(hot_select_lds_bytes()));
      return true;
    }();
    (void)attr_set;
    hipLaunchKernelGGL((hot_select_kernel<UKey>), dim3(1), dim3(kBlock), hot_select_lds_bytes(), stream, src, n, p.shift, p.buckets, p.pitch,
                       1 << p.cap_bits, ht);
    hipLaunchKernelGGL((split_hist_kernel<UKey, true>), dim3(tile_grid(p.tiles)), dim3(kBlock), 0, stream, src, n, p.tile, p.tiles, p.shift,
                       p.buckets, p.pitch, counts, ctl, static_cast<uint32_t*>(nullptr), static_cast<int64_t>(0), ht);
    hipLaunchKernelGGL(split_scan_kernel, dim3(p.pitch / 32), dim3(kBlock), 0, stream, counts, p.tiles, p.pitch, p.buckets,
                       reinterpret_cast<uint32_t*>(ws + p.off_totals), ctl, 1 << p.cap_bits, reinterpret_cast<uint32_t*>(ws + p.off_state),
                       p.pitch + 2, static_cast<uint32_t*>(nullptr), 0u, ht);
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
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
