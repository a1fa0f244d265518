// wholegraph_amd — row gather / scatter kernels for gfx950 (MI355X), hand-written HIP.
//
// Semantics (reference cpp/src/wholememory_ops/functions/gather_scatter_func.cuh:253-316 gather,
// :519-598 scatter; address rule include/wholememory/device_reference.cuh:41-61):
//   gather : plain[row_map(i), c] = cast(table[idx[i], c])   c in [0, dim)   (idx[i] < 0: skipped)
//   scatter: table[idx[i], c]     = cast(plain[row_map(i), c])
//
// MI355X design (NOT the reference's warp-per-row + smem staging):
//  * a wave owns a TILE of consecutive entries: one coalesced index load, each lane resolves ITS
//    entry's owner rank / byte address once (the reference re-derives the rank with a 64-bit divide
//    per element access, device_reference.cuh:47);
//  * the tile is moved in wave "steps" of 16 bytes per lane (global_load_dwordx4 /
//    global_store_dwordx4): a 512 B fp32 row is 32 lanes x 16 B, so a step moves two whole rows as
//    one 1 KiB wave transaction; row base addresses travel between lanes with v_readlane (fast /
//    batch kernels) or ds_bpermute (generic and flat kernels): no LDS allocation, no barriers;
//  * all loads of a batch (4 steps = 4 KiB per wave) are issued before its first store — in the ISA, not only in the source:
//    every kernel tests once per tile whether all its entries move (wave-uniform) and then runs a straight-line batch with
//    nothing predicated (lanes without work of their own repeat the last piece); a conditionally defined load result or a
//    predicated store is enough for hipcc to wait for every load (round 3 shipped that). scripts/check_isa.py disassembles
//    the shipped library and gates the shape; tests/test_isa_gate.py runs it;
//  * the owner of a row of a chunked table is found from KERNEL ARGUMENTS (bases and bounds of up to 8 ranks, host copies of
//    the gref's device arrays) — no 64-bit divide, no dependent load of the peer pointer; hand-built grefs take the device
//    arrays and a multiply-high by a host-computed magic number;
//  * the streamed side (gather output / scatter input) is touched exactly once: non-temporal on
//    both sides, which keeps L2 / Infinity Cache for table rows (which DO repeat under skew);
//  * launch shape (round 3, see rows_op): IN ORDER — one ~4 KiB tile per wave and as many workgroups
//    as that takes, handed out in order by the dispatcher, so that the tiles in flight are one
//    compact window advancing through the streamed side. Rounds 1-2 ran a persistent grid
//    (min(tiles/4, 32 x CUs) workgroups, grid-stride over 64-row tiles), whose scattered write front
//    made the kernel's speed depend on the physical placement of the buffers; that shape remains
//    for callers that cap the grid (gather_sms / scatter_sms) and for the flat-stream scatter;
//  * rows that are not powers of two: the flat-stream kernel (slots of 16 bytes over a tile's rows), and where the dense
//    side is contiguous the LDS-staged kernels (rows_staged_gather_kernel / rows_staged_scatter_kernel): the dense side of
//    a chunk of rows moves as one aligned 16-byte stream through the wave's own LDS region.
// The path is HBM-bound byte movement: there is no contraction here and MFMA is not used.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cxxabi.h>

#include <algorithm>
#include <cstdlib>
#include <string>

#include "../knobs.hpp"
#include "../backend.hpp"
#include "device_common.cuh"

namespace wm {
namespace {

#ifndef WM_BATCH_RPS2_BPERMUTE
#define WM_BATCH_RPS2_BPERMUTE 1
#endif
#ifndef WM_BATCH_MAX_WAVES
#define WM_BATCH_MAX_WAVES 6
#endif
constexpr int kBlock  = 256;   // threads per workgroup of the persistent (capped-grid) launches; upper bound for all
constexpr int kWave   = 64;

template <int BYTES>
struct vec_of;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <>
struct vec_of<16> {
  using type = u32x4;
};
template <>
struct vec_of<8> {
  using type = u32x2;
};
template <>
struct vec_of<4> {
  using type = uint32_t;
};
template <>
struct vec_of<2> {
  using type = uint16_t;
};
template <>
struct vec_of<1> {
  using type = uint8_t;
};

struct rows_params {
  // table side
  char* base;                      // continuous: flat base; chunked: unused
  char* const* rank_ptrs;          // chunked: per-rank bases (device array)
  const size_t* rank_offsets;      // chunked, !same_chunk: byte offsets [world+1] (device array)
  size_t chunk_stride;             // 0 = continuous; else bytes per rank
  int world_size;
  int same_chunk;
  // same_chunk: rank = off / chunk_stride as a multiply-high and a shift (fixed per table: found once on the host)
  uint64_t chunk_magic;            // ceil(2^(63 + l) / chunk_stride), l = ceil(log2(chunk_stride)); 0: chunk_stride == 1
  int chunk_shift;                 // l - 1
  // chunked tables of up to kOwnersByValue ranks whose handle registered host copies of its tables (memory_handle.cpp):
  // the owner of a row is found by comparing against kernel arguments — no dependent load of a pointer, no division
  int owners_by_value;
  uint64_t owner_bound[8];         // byte offset where rank r's memory starts (r >= world_size: ~0)
  uint64_t owner_delta[8];         // rank r's base address - owner_bound[r]: address = delta[owner] + off
  int64_t table_stride_bytes;
  int64_t table_offset_bytes;
  // index side
  const void* indices;
  const int64_t* row_map;
  int64_t n;
  // plain side (already offset by storage_offset)
  char* plain;
  int64_t plain_stride_bytes;
  // geometry
  int row_vecs;   // vectors per row (row_bytes / VB  or  dim / V)
  int lpr_log2;   // log2(lanes per row)
  // flat-stream geometry (rows_flat_kernel): 16-byte slots per row, bytes in the last slot (4..16), 1 / slots
  int flat_slots;
  int flat_tail;
  float flat_rcp;
  // entries per wave tile (64 / 32 / 16 / 8; the fast and flat kernels): big rows get smaller tiles, so that a launch has
  // many more tiles than the chip has wave slots (64 rows of 2 KiB are 128 KiB per wave and 10 M KiB-rows only ~30 k tiles:
  // under four rounds of resident waves, and the last round runs half empty)
  int tile_rows;
  // LDS-staged gather (rows_staged_gather_kernel): rows per chunk (a power of two, chunk bytes a multiple of 16), 0 = not used
  int stage_rows;
  int stage_align;                 // staged gather: wave stores shifted onto the 128-byte lines of the output (1) or begun with the chunk (0)
  // host side only: threads per workgroup of this launch (a multiple of 64, <= kBlock; the kernels read blockDim)
  int launch_threads;
  // host side only: 32 / 64 / 128 / 256 = this launch takes rows_batch_kernel<..., that many 16-byte pieces per row>
  int batch_vecs;
};

// owner of byte offset `off` of a chunked table from the kernel arguments: bounds ascend, the last rank whose start is <= off
// owns it (ranks that do not exist: ~0). The table entries are pinned in SGPRs — left alone, hipcc turns the chain of
// selects into a select of kernarg ADDRESSES and one dependent global load.
__device__ __forceinline__ char* resolve_by_value(const rows_params& p, size_t off)
{
  uint64_t delta = p.owner_delta[0];
  asm volatile("" : "+s"(delta));
#pragma unroll
  for (int r = 1; r < 8; r++) {
    uint64_t d = p.owner_delta[r], b = p.owner_bound[r];
    asm volatile("" : "+s"(d), "+s"(b));
    delta = off >= b ? d : delta;
  }
  return reinterpret_cast<char*>(delta + off);
}

// byte address of the first moved element of table row `idx`
__device__ __forceinline__ char* resolve_row(const rows_params& p, int64_t idx)
{
  size_t off = static_cast<size_t>(p.table_offset_bytes) + static_cast<size_t>(idx) * static_cast<size_t>(p.table_stride_bytes);
  if (p.chunk_stride == 0) return p.base + off;
  if (p.owners_by_value) return resolve_by_value(p, off);
  int rank;
  size_t rank_start;
  if (p.same_chunk) {
    // off / chunk_stride for off < 2^63 (the reference divides, device_reference.cuh:47)
    rank       = static_cast<int>(p.chunk_magic == 0 ? off : (__umul64hi(off, p.chunk_magic) >> p.chunk_shift));
    rank_start = static_cast<size_t>(rank) * p.chunk_stride;
  } else {
    rank = 0;
    for (int r = 1; r < p.world_size; r++) {
      if (off >= p.rank_offsets[r]) rank = r;
    }
    rank_start = p.rank_offsets[rank];
  }
  return p.rank_ptrs[rank] + (off - rank_start);
}

// OWNERS: -1 = whatever the gref says (run-time branches), 0 = continuous, 1 = chunked with the owner tables by value
template <typename IdxT, int OWNERS = -1>
__device__ __forceinline__ void load_tile_entry(const rows_params& p, int64_t entry, char*& tab, char*& pl)
{
  tab = nullptr;
  pl  = nullptr;
  if (entry < p.n) {
#ifndef WM_IDX_NT
#define WM_IDX_NT 0
#endif
    int64_t idx = WM_IDX_NT ? static_cast<int64_t>(__builtin_nontemporal_load(static_cast<const IdxT*>(p.indices) + entry))
                            : static_cast<int64_t>(static_cast<const IdxT*>(p.indices)[entry]);
    if (idx >= 0) {
      if constexpr (OWNERS < 0) {
        tab = resolve_row(p, idx);
      } else {
        const size_t off = static_cast<size_t>(p.table_offset_bytes) + static_cast<size_t>(idx) * static_cast<size_t>(p.table_stride_bytes);
        tab              = OWNERS == 0 ? p.base + off : resolve_by_value(p, off);
      }
      int64_t row = p.row_map ? p.row_map[entry] : entry;  // row map is always int64 (raw_indices)
      pl          = p.plain + row * p.plain_stride_bytes;
    }
  }
  // both bases are complete HERE (the waits for the index / row-map loads sit in front of the tile's batches, not between
  // the row loads of its first batch, where hipcc otherwise lets a late use of the row map drag them)
  asm volatile("" : "+v"(tab), "+v"(pl));
}

__device__ __forceinline__ char* shfl_ptr(char* p, int src_lane)
{
  uint64_t v  = reinterpret_cast<uint64_t>(p);
  uint32_t lo = __shfl(static_cast<uint32_t>(v), src_lane, kWave);
  uint32_t hi = __shfl(static_cast<uint32_t>(v >> 32), src_lane, kWave);
  return reinterpret_cast<char*>((static_cast<uint64_t>(hi) << 32) | lo);
}

// ------------------------------------------------------------------------------------------------
// same-dtype path: pure byte movement with VB-byte vectors
// ------------------------------------------------------------------------------------------------
// generic geometry (any power-of-two lanes-per-row): row bases travel with ds_bpermute
template <typename IdxT, int VB, bool GATHER>
__global__ __launch_bounds__(kBlock) void rows_copy_kernel(rows_params p)
{
  using vec_t         = typename vec_of<VB>::type;
  constexpr int kU    = 4;
  const int lane      = threadIdx.x & (kWave - 1);
  const int64_t wave  = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  const int lpr       = 1 << p.lpr_log2;
  const int rps       = kWave >> p.lpr_log2;  // rows per step
  const int sub       = lane >> p.lpr_log2;   // which row of the step this lane serves
  const int col       = lane & (lpr - 1);
  const int tile_rows = p.tile_rows;   // 64, or fewer (a multiple of rps * kU) for in-order launches of big rows
  const int64_t tiles = (p.n + tile_rows - 1) / tile_rows;

  for (int64_t tile = wave; tile < tiles; tile += n_waves) {
    char *my_tab, *my_plain;
    load_tile_entry<IdxT>(p, lane < tile_rows ? tile * tile_rows + lane : p.n, my_tab, my_plain);
    // every entry of the tile moves (all tiles but the last one of a batch without negative ids): the batch is straight-line
    // code — loads issued back to back, then the stores, nothing predicated. Lanes without work of their own (columns past
    // the end of the row, steps past the end of the tile) repeat the row's last vector / the tile's last row: they load and
    // store the same bytes as the lane that owns them. (A predicated store is enough for hipcc to sink "its" load into the
    // predicate, behind a wait for the loads before it.)
    const bool whole = __ballot(lane < tile_rows && my_tab == nullptr) == 0;
    for (int cbase = 0; cbase < p.row_vecs; cbase += lpr) {  // >1 trip only when a row needs > 64 vectors
      const int c        = cbase + col;
      const bool col_ok  = c < p.row_vecs;
      const int64_t coff = static_cast<int64_t>(min(c, p.row_vecs - 1)) * VB;
      for (int s0 = 0; s0 < tile_rows; s0 += rps * kU) {
        vec_t data[kU];
        char* dst[kU];
        if (whole) {
#pragma unroll
          for (int u = 0; u < kU; u++) {
            const int el = min(s0 + u * rps + sub, tile_rows - 1);
            char* t      = shfl_ptr(my_tab, el);
            char* q      = shfl_ptr(my_plain, el);
            dst[u]       = (GATHER ? q : t) + coff;
            // gather: the table row is read once (non-temporal: 16 / 32 / 64 B rows 18.1 -> 19.8, 30.2 -> 32.6, 48.1 -> 51.8 % of
            // peak); scatter: the dense rows stay cached loads and the table stores cached stores (non-temporal loads 100 B
            // 23.2 -> 22.3, 132 B 26.1 -> 24.5, stores another -0.5: partial lines of neighbouring tiles meet in L2)
            data[u]      = GATHER ? ld_global_nt<vec_t>(t + coff) : ld_global<vec_t>(q + coff);
          }
#pragma unroll
          for (int u = 0; u < kU; u++) {
            if constexpr (GATHER)
              st_global_nt<vec_t>(dst[u], data[u]);
            else
              st_global<vec_t>(dst[u], data[u]);
          }
          continue;
        }
#pragma unroll 1
        for (int u = 0; u < kU; u++) {   // a tile with a skipped entry: one step at a time
          const int e = s0 + u * rps + sub;
          char* t     = shfl_ptr(my_tab, e & (kWave - 1));
          char* q     = shfl_ptr(my_plain, e & (kWave - 1));
          if (col_ok && e < kWave && t != nullptr) {
            const vec_t d = ld_global<vec_t>((GATHER ? t : q) + coff);
            if constexpr (GATHER)
              st_global_nt<vec_t>(q + coff, d);
            else
              st_global<vec_t>(t + coff, d);
          }
        }
      }
    }
  }
}

// 64-bit lane broadcast through v_readlane_b32 (lane index is wave-uniform): the row base lands in
// SGPRs, no LDS crossbar traffic at all
__device__ __forceinline__ char* readlane_ptr(char* p, int src_lane)
{
  const uint64_t v  = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(v), src_lane);
  const uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(v >> 32), src_lane);
  return reinterpret_cast<char*>((static_cast<uint64_t>(hi) << 32) | lo);
}

// The hot geometry: 16-byte vectors and rows of >= 512 B (RPS = 2 rows per wave step, 32 lanes each)
// or >= 1 KiB (RPS = 1). Row bases are broadcast with v_readlane (scalar), kU steps (kU KiB per wave)
// of random row reads are in flight before the first store (kU = 4 measured best: 4 -> 73.1 %, 8 -> 72.5 %,
// 16 -> 68.9 % of HBM peak on the 10 M-id gather; occupancy, not per-wave depth, provides the parallelism), table rows are read with the
// non-temporal hint in gather (each row is used once per batch; under skew the Infinity Cache
// still serves repeats), the streamed side is written non-temporally. Scatter mirrors it: the streamed
// input is READ non-temporally (measured: 2.05 ms -> 1.71 ms per 10 M rows) and the table rows are written
// non-temporally (-> 1.67 ms, 77 % of HBM peak). Prefetching the NEXT tile's indices before streaming the current tile was
// measured too: no change (1.772 vs 1.774 ms) — occupancy already hides that latency.
template <typename IdxT, bool GATHER, int RPS, bool HAS_MAP>
__global__ __launch_bounds__(kBlock) void rows_copy16_fast_kernel(rows_params p)
{
#ifndef WM_FAST_KU
#define WM_FAST_KU 4
#endif
  constexpr int kU      = WM_FAST_KU;
  constexpr int kLpr    = kWave / RPS;
  const int lane        = threadIdx.x & (kWave - 1);
  const int64_t wave    = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  const int col         = lane & (kLpr - 1);
  const bool upper      = RPS == 2 && lane >= kLpr;
  const int tile_rows   = p.tile_rows;
  const int64_t tiles   = (p.n + tile_rows - 1) / tile_rows;

  for (int64_t tile = wave; tile < tiles; tile += n_waves) {
    char *my_tab, *my_plain;
    load_tile_entry<IdxT>(p, lane < tile_rows ? tile * tile_rows + lane : p.n, my_tab, my_plain);
    // without a row map the plain side is affine in the entry number: no broadcast needed
    char* const plain_tile = p.plain + (tile * tile_rows + (upper ? 1 : 0)) * p.plain_stride_bytes;
    // The tile is a sequence of (row, 1 KiB chunk) steps, row-major: a row of more than 1 KiB is read front to back by
    // consecutive wave instructions (kU of them in flight), not chunk 0 of every row first and chunk 1 a pass later —
    // the second half of a row then finds its DRAM page still open (scatter of 2 KiB rows 65.8 -> 69.8 % of HBM peak, 4 KiB rows 63.7 -> 70.5 %; gather unchanged).
    const int chunks = RPS == 1 ? (p.row_vecs + kLpr - 1) / kLpr : 1;
    const int total  = (tile_rows / RPS) * chunks;   // a multiple of kU: tile_rows / RPS is (tile_rows >= 8)
    // every entry of the tile moves: straight-line batches — loads back to back, then the stores, nothing predicated (lanes
    // past the end of a row repeat its last piece; see rows_copy_kernel)
    const bool whole = __ballot(lane < tile_rows && my_tab == nullptr) == 0;
    int e = 0, cb = 0;
#pragma unroll 1
    for (int q0 = 0; q0 < total; q0 += kU) {
      u32x4 data[kU];
      char* dst[kU];
      if (whole) {
#pragma unroll
        for (int u = 0; u < kU; u++) {
          const int e0 = RPS * e;
          char* t      = readlane_ptr(my_tab, e0);
          if (RPS == 2) {
            char* t1 = readlane_ptr(my_tab, e0 + 1);
            t        = upper ? t1 : t;
          }
          char* q;
          if (HAS_MAP) {
            q = readlane_ptr(my_plain, e0);
            if (RPS == 2) {
              char* q1 = readlane_ptr(my_plain, e0 + 1);
              q        = upper ? q1 : q;
            }
          } else {
            q = plain_tile + static_cast<int64_t>(e0) * p.plain_stride_bytes;
          }
          const int c        = cb * kLpr + col;
          const int64_t coff = static_cast<int64_t>(min(c, p.row_vecs - 1)) * 16;
          dst[u]             = (GATHER ? q : t) + coff;
          data[u]            = ld_global_nt<u32x4>((GATHER ? t : q) + coff);
          if (++cb == chunks) {  // wave-uniform: stays in SGPRs
            cb = 0;
            e++;
          }
        }
#pragma unroll
        for (int u = 0; u < kU; u++) st_global_nt<u32x4>(dst[u], data[u]);
        continue;
      }
#pragma unroll 1
      for (int u = 0; u < kU; u++) {   // a tile with a skipped entry: one step at a time
        const int e0 = RPS * e;
        char* t      = readlane_ptr(my_tab, e0);
        if (RPS == 2) {
          char* t1 = readlane_ptr(my_tab, e0 + 1);
          t        = upper ? t1 : t;
        }
        char* q;
        if (HAS_MAP) {
          q = readlane_ptr(my_plain, e0);
          if (RPS == 2) {
            char* q1 = readlane_ptr(my_plain, e0 + 1);
            q        = upper ? q1 : q;
          }
        } else {
          q = plain_tile + static_cast<int64_t>(e0) * p.plain_stride_bytes;
        }
        const int c        = cb * kLpr + col;
        const int64_t coff = static_cast<int64_t>(c) * 16;
        if (c < p.row_vecs && t != nullptr) {  // entries past n and negative ids carry a null base
          const u32x4 d = ld_global_nt<u32x4>((GATHER ? t : q) + coff);
          st_global_nt<u32x4>((GATHER ? q : t) + coff, d);
        }
        if (++cb == chunks) {  // wave-uniform: stays in SGPRs
          cb = 0;
          e++;
        }
      }
    }
  }
}

// The in-order shape of the hot geometry, fully specialised: rows of 512 B, 1, 2 or 4 KiB (ROW_VECS 16-byte pieces), one
// tile of 4 KiB per wave moved as ONE batch and the wave is done; the dispatcher hands the next tile to whichever wave slot
// frees up first, in order. A tile whose entries all move (the wave-uniform test on the ballot below; every tile but the
// last one of a batch without negative ids) takes the FAST PATH: four unconditional 1 KiB wave loads into four distinct
// register quads, issued back to back, then the four stores behind s_waitcnt vmcnt(3..0) — one basic block, no per-step
// predicate, so nothing makes the compiler merge registers or wait between the loads (round 3 shipped
// `if (t != nullptr) data[u] = ld(...)`: hipcc then waited for every load before issuing the next and spent 133 of 294 VALU
// instructions on v_mov; scripts/check_isa.py now gates the shape of this code). Row bases reach the wave through
// v_readlane into SGPRs; without a row map the streamed side is affine in the tile number (scalar base + a 32-bit lane offset).
// A tile with a skipped entry takes the predicated path, one step at a time.
// At most 6 waves per SIMD: with 24 instead of 32 waves per CU the window of tiles in flight is a quarter narrower and the
// kernel a little faster on every shape (profiles/r03_dim_sweep_occupancy.csv).
template <typename IdxT, bool GATHER, int ROW_VECS, bool HAS_MAP, int OWNERS>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(1, WM_BATCH_MAX_WAVES))) void rows_batch_kernel(rows_params p)
{
  constexpr int kSteps  = 4;                                        // 4 x 1 KiB
  constexpr int RPS     = ROW_VECS == 32 ? 2 : 1;                   // rows per wave instruction
  constexpr int kChunks = ROW_VECS <= 64 ? 1 : ROW_VECS / 64;       // wave instructions per row
  constexpr int kRows   = 256 / ROW_VECS;                           // rows per tile: 8, 4, 2, 1
  const int lane        = threadIdx.x & (kWave - 1);
  const int64_t tile    = static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t first   = tile * kRows;                             // wave-uniform: lives in SGPRs
  if (first >= p.n) return;
  char *my_tab, *my_plain;
  load_tile_entry<IdxT, OWNERS>(p, lane < kRows ? first + lane : p.n, my_tab, my_plain);
  const bool upper          = RPS == 2 && lane >= 32;
  const uint32_t col16      = static_cast<uint32_t>(RPS == 2 ? (lane & 31) : lane) * 16u;
  // the streamed side without a row map: row (first + e) starts at plain_tile + e * stride
  char* const plain_tile    = p.plain + first * p.plain_stride_bytes;
  const uint32_t plain_lane = col16 + (upper ? static_cast<uint32_t>(p.plain_stride_bytes) : 0u);   // stride < 2^31: rows_op

  // RPS == 2: the two halves of the wave serve different rows. WM_BATCH_RPS2_BPERMUTE=1: each lane fetches ITS row's base
  // with ds_bpermute (2 LDS-crossbar instructions per 64-bit base and step); 0: both bases come through v_readlane and a
  // per-lane select (4 v_readlane + 4 v_mov + 2 v_cndmask per base and step)
  const int half_lane4 = (lane >> 5) * 4;                           // byte index of lane (lane >= 32 ? 1 : 0) for ds_bpermute
  auto lane_ptr = [&](char* mine, int e0) -> char* {                // base of row e0 (lower half) / e0 + 1 (upper half)
#if WM_BATCH_RPS2_BPERMUTE
    const uint64_t v  = reinterpret_cast<uint64_t>(mine);
    const uint32_t lo = __builtin_amdgcn_ds_bpermute(half_lane4 + 4 * e0, static_cast<uint32_t>(v));
    const uint32_t hi = __builtin_amdgcn_ds_bpermute(half_lane4 + 4 * e0, static_cast<uint32_t>(v >> 32));
    return reinterpret_cast<char*>((static_cast<uint64_t>(hi) << 32) | lo);
#else
    char* a = readlane_ptr(mine, e0);
    char* b = readlane_ptr(mine, e0 + 1);
    return upper ? b : a;
#endif
  };
  auto tab_addr = [&](int u) -> char* {                             // u is a compile-time constant after unrolling
    const int e0 = RPS == 2 ? 2 * u : u / kChunks;
    const int cb = RPS == 2 ? 0 : u % kChunks;
    char* t      = RPS == 2 ? lane_ptr(my_tab, e0) : readlane_ptr(my_tab, e0);
    return t + (cb * 1024 + col16);
  };
  auto plain_addr = [&](int u) -> char* {
    const int e0 = RPS == 2 ? 2 * u : u / kChunks;
    const int cb = RPS == 2 ? 0 : u % kChunks;
    if (HAS_MAP) {
      char* q = RPS == 2 ? lane_ptr(my_plain, e0) : readlane_ptr(my_plain, e0);
      return q + (cb * 1024 + col16);
    }
    return plain_tile + static_cast<int64_t>(e0) * p.plain_stride_bytes + (cb * 1024 + plain_lane);
  };

  constexpr uint64_t kAll = (1ull << kRows) - 1;
  const uint64_t moving   = __ballot(my_tab != nullptr);            // entries past n and negative ids carry a null base
  if ((moving & kAll) == kAll) {
    u32x4 data[kSteps];
#pragma unroll
    for (int u = 0; u < kSteps; u++) data[u] = ld_global_nt<u32x4>(GATHER ? tab_addr(u) : plain_addr(u));
#pragma unroll
    for (int u = 0; u < kSteps; u++) st_global_nt<u32x4>(GATHER ? plain_addr(u) : tab_addr(u), data[u]);
    return;
  }
#pragma unroll 1
  for (int r = 0; r < kRows; r++) {                                 // a tile with a skipped entry: row by row
    if (((moving >> r) & 1) == 0) continue;                         // wave-uniform
    char* t = readlane_ptr(my_tab, r);
    char* q = HAS_MAP ? readlane_ptr(my_plain, r) : plain_tile + static_cast<int64_t>(r) * p.plain_stride_bytes;
    for (int c = lane; c < ROW_VECS; c += kWave) {
      const u32x4 d = ld_global_nt<u32x4>((GATHER ? t : q) + c * 16);
      st_global_nt<u32x4>((GATHER ? q : t) + c * 16, d);
    }
  }
}



// Rows whose size is not a power of two (400 B, 1200 B, 2408 B ... : dims 100 / 300 / 602 of common GNN datasets), and
// 1 KiB rows. The pow-of-two lane mapping above leaves lanes idle and walks a big row in several passes; here the tile is
// a FLAT STREAM of 16-byte slots: slot v of the tile belongs to row v / S, column v % S (S slots per row), lane l of
// step k owns slot 64 k + l. Every lane is busy whatever S is, a row is walked front to back once (DRAM page locality),
// and a dense output is written as one contiguous stream. Row bases travel with ds_bpermute. Rows whose byte count is
// only a multiple of 4 use full 16-byte accesses at their natural (4-byte) alignment — legal on gfx950 global memory —
// and the LAST slot of such a row is shifted back so that it ENDS with the row (bytes [row_bytes - 16, row_bytes)): it
// overlaps its neighbour by 16 - tail bytes, which both slots read from the same source and write with the same values.
// No dword tail, no per-lane special case: every slot is one 16-byte load and one 16-byte store (round 3 carried the tail
// as conditional dword accesses inside the slot loop: 100-104 VGPRs and a wait after every load).
// A batch is kU x 64 slots: in a tile whose entries all move it is straight-line code, loads issued back to back and then
// the stores (slots past the end of the tile repeat its last slot).
// (A variant without the per-slot division and ds_bpermute — row bases in SGPRs, ceil(S / 64) wave instructions per row with
// idle lanes — measured equal within +-2 points on 528 B ... 4000 B rows, profiles/r04_misaligned_rows.txt: what these rows
// lose against the powers of two is partial cache lines, not instructions.)
template <typename IdxT, bool GATHER, bool HAS_MAP>
__global__ __launch_bounds__(kBlock) void rows_flat_kernel(rows_params p)
{
  constexpr int kU      = 4;
  const int lane        = threadIdx.x & (kWave - 1);
  const int64_t wave    = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  const int tile_rows   = p.tile_rows;
  const int64_t tiles   = (p.n + tile_rows - 1) / tile_rows;
  const int S           = p.flat_slots;
  const int n_slots     = tile_rows * S;
  const int last_off    = (S - 1) * 16 + p.flat_tail - 16;   // row_bytes - 16: where the last slot of a row starts

  // slot v -> (row of the tile, byte offset in the row)
  auto locate = [&](int v, int& row, int& off) {
    row     = static_cast<int>(static_cast<float>(v) * p.flat_rcp);  // v / S, fixed up below
    int col = v - row * S;
    if (col < 0) row--, col += S;
    if (col >= S) row++, col -= S;
    off = col == S - 1 ? last_off : col * 16;
  };

  for (int64_t tile = wave; tile < tiles; tile += n_waves) {
    char *my_tab, *my_plain;
    load_tile_entry<IdxT>(p, lane < tile_rows ? tile * tile_rows + lane : p.n, my_tab, my_plain);
    char* const plain_tile = p.plain + tile * tile_rows * p.plain_stride_bytes;
    const bool whole       = __ballot(lane < tile_rows && my_tab == nullptr) == 0;
#pragma unroll 1
    for (int v0 = 0; v0 < n_slots; v0 += kWave * kU) {
      if (whole) {
        u32x4 data[kU];
        char* dst[kU];
#pragma unroll
        for (int u = 0; u < kU; u++) {
          const int v = v0 + u * kWave + lane;
          int row, off;
          locate(min(v, n_slots - 1), row, off);
          char* t = shfl_ptr(my_tab, row);
          char* q = HAS_MAP ? shfl_ptr(my_plain, row) : plain_tile + row * p.plain_stride_bytes;
          dst[u]  = (GATHER ? q : t) + off;   // a slot past the end of the tile repeats its last slot: same bytes, same place
          if constexpr (GATHER)
            // rows share cache lines with their neighbours: kept (non-temporal: 528-640 B rows +1.3 ... +2.6 points, 800 B - 4000 B
            // -0.9 ... -1.6 in one session and +3.0 / +1.9 / -0.8 at 800 / 1200 / 4000 B in another: no rule, round 4)
            data[u] = ld_global<u32x4>(t + off);
          else
            data[u] = ld_global_nt<u32x4>(q + off);
        }
#pragma unroll
        for (int u = 0; u < kU; u++) st_global_nt<u32x4>(dst[u], data[u]);
        continue;
      }
#pragma unroll 1
      for (int u = 0; u < kU; u++) {   // a tile with a skipped entry: one step at a time
        const int v = v0 + u * kWave + lane;
        int row, off;
        locate(min(v, n_slots - 1), row, off);
        char* t = shfl_ptr(my_tab, row);   // (ds_bpermute needs every lane: outside the guard)
        char* q = HAS_MAP ? shfl_ptr(my_plain, row) : plain_tile + row * p.plain_stride_bytes;
        if (v < n_slots && t != nullptr) {  // entries past n and negative ids carry a null base
          const u32x4 d = GATHER ? ld_global<u32x4>(t + off) : ld_global_nt<u32x4>(q + off);
          st_global_nt<u32x4>((GATHER ? q : t) + off, d);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LDS-staged gather for a DENSE output whose rows are only 4- or 8-byte aligned (dim 129 -> 516 B, dim 602 -> 2408 B ...)
// ------------------------------------------------------------------------------------------------
// The table rows of such an embedding start on 16-byte boundaries (the stride is padded), the rows of the dense [n, dim]
// output do not: every 16-byte store of the flat kernel straddles 64-byte sectors on the output side. But the output of R
// consecutive entries is ONE contiguous piece of R * row_bytes bytes, 16-byte aligned as a whole when R * row_bytes is a
// multiple of 16 — so the rows of a chunk are assembled in LDS at their dense offsets (aligned 16-byte global loads from
// the table, dword LDS writes) and leave as aligned 16-byte non-temporal stores of the contiguous stream. Each wave has
// its own LDS region: no workgroup barrier, only the wave's own LDS ordering. A chunk with a skipped entry (negative id, or
// past the end of the batch) must not touch that entry's output row: such chunks take the per-row path (dword stores).
// kStageIters = 16-byte slots per lane per chunk: chunks of up to 5 KiB (5 x 64 x 16 B), or 10 KiB for the rows whose
// smallest aligned group (2 or 4 rows) does not fit 5 KiB
template <typename IdxT, int kStageIters>
__global__ __launch_bounds__(kBlock) void rows_staged_gather_kernel(rows_params p)
{
  extern __shared__ __attribute__((aligned(16))) char staged_lds[];
  const int lane        = threadIdx.x & (kWave - 1);
  const int wave_in_blk = threadIdx.x >> 6;
  const int64_t wave    = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  const int tile_rows   = p.tile_rows;                        // 64 (persistent launches) or one chunk of R rows (in-order launches)
  const int64_t tiles   = (p.n + tile_rows - 1) / tile_rows;
  const int R           = p.stage_rows;
  const int S           = p.flat_slots;                       // 16-byte slots per row, the last one holds flat_tail bytes
  const int row_bytes   = (S - 1) * 16 + p.flat_tail;
  const int chunk_bytes = R * row_bytes;                      // multiple of 16
  const int chunk_slots = R * S;
  const int chunk_vecs  = chunk_bytes >> 4;
  char* const lds       = staged_lds + wave_in_blk * ((chunk_bytes + 15) & ~15);
  const int tail_words  = p.flat_tail >> 2;

  for (int64_t tile = wave; tile < tiles; tile += n_waves) {
    char *my_tab, *my_plain;
    load_tile_entry<IdxT>(p, lane < tile_rows ? tile * tile_rows + lane : p.n, my_tab, my_plain);
    const uint64_t present = __ballot(my_tab != nullptr);
    for (int r0 = 0; r0 < tile_rows; r0 += R) {
      const uint64_t chunk_mask = (R == 64 ? ~0ull : ((1ull << R) - 1)) << r0;
      if ((present & chunk_mask) == 0) continue;              // nothing to move (tail of the last tile, all skipped)
      char* const out = p.plain + (tile * tile_rows + r0) * static_cast<int64_t>(row_bytes);
      if ((present & chunk_mask) == chunk_mask) {
        // ---- table -> LDS: slot v of the chunk = row v / S, piece v % S; every load of the chunk is issued before the
        // first LDS write (up to kStageIters x 1 KiB per wave in flight)
        u32x4 d[kStageIters];
        int lds_off[kStageIters], nw[kStageIters];
#pragma unroll
        for (int i = 0; i < kStageIters; i++) {
          const int v = min(lane + i * kWave, chunk_slots - 1);   // clamped: the broadcast below needs every lane
          const int r = static_cast<int>(static_cast<float>(v) * p.flat_rcp);
          int row = r, col = v - r * S;
          if (col < 0) row--, col += S;
          if (col >= S) row++, col -= S;
          // (ds_bpermute returns 0 for an inactive SOURCE lane: the broadcast must not sit inside the guard)
          const char* t = shfl_ptr(my_tab, r0 + row);
          // unconditional: a lane past the end of the chunk re-reads its last slot (and writes nothing to LDS) — a load under
          // a per-lane guard is a conditionally defined register, and hipcc then waits for each load before the next
          d[i]          = ld_global_nt<u32x4>(t + col * 16);   // the padded table row holds 16 bytes here even in the tail slot
          lds_off[i]    = row * row_bytes + col * 16;
          nw[i]         = lane + i * kWave < chunk_slots ? (col == S - 1 ? tail_words : 4) : 0;
        }
#pragma unroll
        for (int i = 0; i < kStageIters; i++) {
          uint32_t* dst = reinterpret_cast<uint32_t*>(lds + lds_off[i]);
#pragma unroll
          for (int w = 0; w < 4; w++)
            if (w < nw[i]) dst[w] = d[i][w];
        }
        // the rows of the chunk sit in this wave's LDS region only; LDS operations of one wave complete in order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- LDS -> the contiguous, 16-byte aligned output stream. The lanes are shifted so that every wave store begins on a
        // 128-byte line of the output (the chunk itself begins wherever R rows put it): whole-line writes except at the two
        // ends of the chunk (what writes that begin inside a line cost: profiles/r04_misaligned_rows.txt)
        const int head = p.stage_align ? __builtin_amdgcn_readfirstlane(static_cast<int>((reinterpret_cast<uint64_t>(out) >> 4) & 7)) : 0;
        for (int v = lane - head; v < chunk_vecs; v += kWave)
          if (v >= 0) st_global_nt<u32x4>(out + v * 16, *reinterpret_cast<const u32x4*>(lds + v * 16));
        __builtin_amdgcn_wave_barrier();                      // the next chunk overwrites the region
      } else {
        // ---- a skipped entry inside the chunk: row by row, dword stores at the rows' natural alignment
        for (int r = 0; r < R; r++) {
          const char* t = shfl_ptr(my_tab, r0 + r);
          if (t == nullptr) continue;                         // wave-uniform
          char* q = out + r * static_cast<int64_t>(row_bytes);
          for (int w = lane; w < (row_bytes >> 2); w += kWave) st_global<uint32_t>(q + 4 * w, ld_global<uint32_t>(t + 4 * w));
        }
      }
    }
  }
}

// The mirror for SCATTER (round 3): the dense INPUT rows are the misaligned side. A chunk of R consecutive input rows is one
// contiguous, 16-byte aligned piece: it enters LDS with aligned 16-byte non-temporal loads, and every table row is then
// written as whole aligned 16-byte pieces (+ its dword tail) assembled from dword LDS reads at the row's dense offset.
// Entries with a negative id are simply not written (the dense side is only read); a chunk that reaches past the end of
// the batch takes the per-row path so that nothing is read behind the input.
template <typename IdxT, int kStageIters>
__global__ __launch_bounds__(kBlock) void rows_staged_scatter_kernel(rows_params p)
{
  extern __shared__ __attribute__((aligned(16))) char staged_lds[];
  const int lane        = threadIdx.x & (kWave - 1);
  const int wave_in_blk = threadIdx.x >> 6;
  const int64_t wave    = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  const int tile_rows   = p.tile_rows;
  const int64_t tiles   = (p.n + tile_rows - 1) / tile_rows;
  const int R           = p.stage_rows;
  const int S           = p.flat_slots;
  const int row_bytes   = (S - 1) * 16 + p.flat_tail;
  const int chunk_bytes = R * row_bytes;
  const int chunk_slots = R * S;
  const int chunk_vecs  = chunk_bytes >> 4;
  char* const lds       = staged_lds + wave_in_blk * ((chunk_bytes + 15) & ~15);
  const int tail_words  = p.flat_tail >> 2;

  for (int64_t tile = wave; tile < tiles; tile += n_waves) {
    char *my_tab, *my_plain;
    load_tile_entry<IdxT>(p, lane < tile_rows ? tile * tile_rows + lane : p.n, my_tab, my_plain);
    const uint64_t present = __ballot(my_tab != nullptr);
    for (int r0 = 0; r0 < tile_rows; r0 += R) {
      const uint64_t chunk_mask = (R == 64 ? ~0ull : ((1ull << R) - 1)) << r0;
      if ((present & chunk_mask) == 0) continue;
      const int64_t e0     = tile * tile_rows + r0;
      const char* const in = p.plain + e0 * static_cast<int64_t>(row_bytes);
      if (e0 + R <= p.n) {
        // ---- dense input -> LDS, aligned 16-byte pieces of the contiguous stream
        u32x4 d[kStageIters];
#pragma unroll
        for (int i = 0; i < kStageIters; i++)   // unconditional (a lane past the end re-reads the chunk's last piece): see the gather
          d[i] = ld_global_nt<u32x4>(in + min(lane + i * kWave, chunk_vecs - 1) * 16);
#pragma unroll
        for (int i = 0; i < kStageIters; i++) {
          const int v = lane + i * kWave;
          if (v < chunk_vecs) *reinterpret_cast<u32x4*>(lds + v * 16) = d[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- LDS -> table rows: slot v of the chunk = row v / S, piece v % S
#pragma unroll
        for (int i = 0; i < kStageIters; i++) {
          const int v = min(lane + i * kWave, chunk_slots - 1);   // clamped: the broadcast below needs every lane
          const int r = static_cast<int>(static_cast<float>(v) * p.flat_rcp);
          int row = r, col = v - r * S;
          if (col < 0) row--, col += S;
          if (col >= S) row++, col -= S;
          char* t = shfl_ptr(my_tab, r0 + row);
          if (lane + i * kWave < chunk_slots && t != nullptr) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(lds + row * row_bytes + col * 16);
            if (col < S - 1 || tail_words == 4) {
              u32x4 x;
              x[0] = src[0], x[1] = src[1], x[2] = src[2], x[3] = src[3];
              st_global_nt<u32x4>(t + col * 16, x);
            } else {
#pragma unroll
              for (int w = 0; w < 3; w++)
                if (w < tail_words) st_global<uint32_t>(t + col * 16 + 4 * w, src[w]);
            }
          }
        }
        __builtin_amdgcn_wave_barrier();                      // the next chunk overwrites the region
      } else {
        for (int r = 0; r < R && e0 + r < p.n; r++) {
          char* t = shfl_ptr(my_tab, r0 + r);
          if (t == nullptr) continue;                         // wave-uniform
          const char* q = in + r * static_cast<int64_t>(row_bytes);
          for (int w = lane; w < (row_bytes >> 2); w += kWave) st_global<uint32_t>(t + 4 * w, ld_global<uint32_t>(q + 4 * w));
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// converting path: V elements per lane, FromT -> ToT through the reference's conversion chain
// ------------------------------------------------------------------------------------------------
template <typename T, int V>
struct alignas(sizeof(T) * V) elt_vec {
  T v[V];
};

// struct-typed accesses through an address-space pointer fall back to FLAT (the copy goes through a generic reference):
// move the bytes as a plain vector of the same size and reinterpret them in registers
template <typename S, bool NT = false>
__device__ __forceinline__ S ld_global_pod(const void* p)
{
  constexpr size_t N = sizeof(S);
  static_assert(N == 1 || N == 2 || N == 4 || N == 8 || N == 16 || N == 32, "element vector size");
  S out;
  if constexpr (N == 32) {
    u32x4 r[2] = {NT ? ld_global_nt<u32x4>(p) : ld_global<u32x4>(p),
                  NT ? ld_global_nt<u32x4>(static_cast<const char*>(p) + 16) : ld_global<u32x4>(static_cast<const char*>(p) + 16)};
    __builtin_memcpy(&out, r, N);
  } else {
    using V = typename vec_of<N>::type;
    V r     = NT ? ld_global_nt<V>(p) : ld_global<V>(p);
    __builtin_memcpy(&out, &r, N);
  }
  return out;
}
template <typename S, bool NT = false>
__device__ __forceinline__ void st_global_pod(void* p, const S& v)
{
  constexpr size_t N = sizeof(S);
  if constexpr (N == 32) {
    u32x4 r[2];
    __builtin_memcpy(r, &v, N);
    if (NT) {
      st_global_nt<u32x4>(p, r[0]);
      st_global_nt<u32x4>(static_cast<char*>(p) + 16, r[1]);
    } else {
      st_global<u32x4>(p, r[0]);
      st_global<u32x4>(static_cast<char*>(p) + 16, r[1]);
    }
  } else {
    using V = typename vec_of<N>::type;
    V r;
    __builtin_memcpy(&r, &v, N);
    if (NT) st_global_nt<V>(p, r); else st_global<V>(p, r);
  }
}

template <typename TabT, typename PlainT, typename IdxT, int V, bool GATHER>
__global__ __launch_bounds__(kBlock) void rows_convert_kernel(rows_params p)
{
  using FromT         = typename std::conditional<GATHER, TabT, PlainT>::type;
  using ToT           = typename std::conditional<GATHER, PlainT, TabT>::type;
  const int lane      = threadIdx.x & (kWave - 1);
  const int64_t wave    = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  const int lpr       = 1 << p.lpr_log2;
  const int rps       = kWave >> p.lpr_log2;
  const int sub       = lane >> p.lpr_log2;
  const int col       = lane & (lpr - 1);
  const int tile_rows = p.tile_rows;   // 64, or fewer (a multiple of rps * kU) for in-order launches of big rows
  const int64_t tiles = (p.n + tile_rows - 1) / tile_rows;
  constexpr int kU    = 4;

  for (int64_t tile = wave; tile < tiles; tile += n_waves) {
    char *my_tab, *my_plain;
    load_tile_entry<IdxT>(p, lane < tile_rows ? tile * tile_rows + lane : p.n, my_tab, my_plain);
    // every entry of the tile moves: straight-line batches, nothing predicated (idle lanes repeat the row's last vector, steps
    // past the tile its last row; see rows_copy_kernel)
    const bool whole = __ballot(lane < tile_rows && my_tab == nullptr) == 0;
    for (int cbase = 0; cbase < p.row_vecs; cbase += lpr) {
      const int c       = cbase + col;
      const bool col_ok = c < p.row_vecs;
      const int64_t cc  = min(c, p.row_vecs - 1);
      for (int s0 = 0; s0 < tile_rows; s0 += rps * kU) {
        if (whole) {
          elt_vec<FromT, V> data[kU];
          char* dst[kU];
#pragma unroll
          for (int u = 0; u < kU; u++) {
            const int el = min(s0 + u * rps + sub, tile_rows - 1);
            char* t      = shfl_ptr(my_tab, el);
            char* q      = shfl_ptr(my_plain, el);
            dst[u]       = (GATHER ? q : t) + cc * V * sizeof(ToT);
            // non-temporal on both sides, like the copying kernels (round 4, interleaved A/B on 16 <-> 32-bit pairs: gather
            // 67.3 -> 70.8, 69.5 -> 75.7, 67.8 -> 74.0, 67.3 -> 72.7 % of peak, scatter 76.2 -> 82.4, 78.9 -> 81.0, 74.7 -> 80.8,
            // 100-element rows +2.4 / +6.5: profiles/r04_cast_sweep_nontemporal_ab.txt)
            data[u]      = ld_global_pod<elt_vec<FromT, V>, true>((GATHER ? t : q) + cc * V * sizeof(FromT));
          }
#pragma unroll
          for (int u = 0; u < kU; u++) {
            elt_vec<ToT, V> o;
#pragma unroll
            for (int k = 0; k < V; k++) o.v[k] = convert_elt<FromT, ToT>(data[u].v[k]);
            st_global_pod<elt_vec<ToT, V>, true>(dst[u], o);
          }
          continue;
        }
#pragma unroll 1
        for (int u = 0; u < kU; u++) {   // a tile with a skipped entry: one step at a time
          const int e = s0 + u * rps + sub;
          char* t     = shfl_ptr(my_tab, e & (kWave - 1));
          char* q     = shfl_ptr(my_plain, e & (kWave - 1));
          if (col_ok && e < kWave && t != nullptr) {
            const elt_vec<FromT, V> d = ld_global_pod<elt_vec<FromT, V>>((GATHER ? t : q) + static_cast<int64_t>(c) * V * sizeof(FromT));
            elt_vec<ToT, V> o;
#pragma unroll
            for (int k = 0; k < V; k++) o.v[k] = convert_elt<FromT, ToT>(d.v[k]);
            st_global_pod<elt_vec<ToT, V>>((GATHER ? q : t) + static_cast<int64_t>(c) * V * sizeof(ToT), o);
          }
        }
      }
    }
  }
}

// every launch goes through here so the library can tell which kernel instantiation served the last call of this thread
// (wholememory_ext_last_rows_kernel: bench.py reports the name the HIP runtime holds for it, not a hard-coded string)
thread_local const void* t_last_rows_kernel = nullptr;
template <typename K>
inline void launch_rows_kernel(K kernel, int blocks, hipStream_t stream, const rows_params& p)
{
  t_last_rows_kernel = reinterpret_cast<const void*>(kernel);
  // WM_ROWS_LDS=bytes (experiments): dynamic LDS nobody uses, to cap the workgroups resident per CU
  const char* le   = WM_AB_KNOB("WM_ROWS_LDS");
  const size_t lds = le != nullptr ? static_cast<size_t>(atoi(le)) : 0;
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(p.launch_threads), lds, stream, p);
}

// rank = off / chunk_stride as multiply-high + shift (Granlund / Montgomery, dividends below 2^63): with l = ceil(log2 d)
// and m = ceil(2^(63 + l) / d) (< 2^64), floor(n / d) = (n * m) >> (63 + l) = umulhi64(n, m) >> (l - 1) for every n < 2^63
inline void magic_for(uint64_t d, uint64_t* m, int* shift)
{
  if (d <= 1) {
    *m = 0, *shift = 0;
    return;
  }
  int l = 0;
  while (l < 64 && (l == 63 ? false : (uint64_t(1) << l) < d)) l++;
  if (l >= 63) l = 63;   // d > 2^62: quotients are 0 or 1; m below still satisfies the bound for n < 2^63
  const unsigned __int128 num = static_cast<unsigned __int128>(1) << (63 + l);
  *m     = static_cast<uint64_t>((num + d - 1) / d);
  *shift = l - 1;
}

// chunked table: the division constants, and — when the handle registered host copies of its tables — bases and bounds
// by value (wm::lookup_gref_tables, backend.hpp)
void fill_owner_tables(rows_params* p)
{
  magic_for(p->chunk_stride, &p->chunk_magic, &p->chunk_shift);
  const char* off = WM_KNOB("WM_ROWS_OWNERS_BY_VALUE");   // =0: device arrays only (A/B, tests)
  gref_host_tables t;
  if ((off != nullptr && off[0] == '0') || !lookup_gref_tables(p->rank_ptrs, &t) || t.world_size != p->world_size) return;
  p->owners_by_value = 1;
  for (int r = 0; r < kOwnersByValue; r++) {
    const bool exists  = r < t.world_size;
    p->owner_bound[r]  = exists ? t.rank_offsets[r] : ~uint64_t(0);
    p->owner_delta[r]  = exists ? reinterpret_cast<uint64_t>(t.rank_ptrs[r]) - t.rank_offsets[r] : 0;
  }
  // (a rank without memory shares its start with the next one: the chain of comparisons lands on the LAST rank with that
  // start, the real owner — the same answer as the search over rank_offsets in resolve_row)
}

inline int ilog2_ceil(int x)
{
  int l = 0;
  while ((1 << l) < x) l++;
  return l;
}

inline int64_t pow2_divisor(int64_t v, int64_t cap)
{  // largest power of two <= cap dividing v (v == 0 divides everything)
  int64_t a = cap;
  while (a > 1 && (v % a) != 0) a >>= 1;
  return a;
}

int default_max_blocks()
{
  static int cus = 0;
  if (cus == 0) {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    else
      cus = 256;
  }
  return cus * 32;  // measured: 8192 workgroups (32 per CU) beat 2048 by ~2% on the 10 M-id gather
}

// WM_ROWS_INORDER=0: the persistent grid-stride launches of rounds 1-2 everywhere; 1: in-order launches everywhere;
// unset (-1): the measured rule in rows_op
int inorder_setting()
{
  const char* e = WM_KNOB("WM_ROWS_INORDER");
  return (e == nullptr || e[0] == '\0') ? -1 : (e[0] == '0' ? 0 : 1);
}
// threads per workgroup of the in-order launches (WM_ROWS_BLOCK=64 / 128 / 256; measured: 64 and 256 within 1 %, 512 and
// 1024 5-12 % slower — the finer the unit the dispatcher hands out, the tighter the window)
int inorder_block_threads()
{
  const char* e = WM_AB_KNOB("WM_ROWS_BLOCK");
  const int v   = e != nullptr ? atoi(e) : 0;
  return (v == 64 || v == 128 || v == 256) ? v : 256;
}

// rows per wave tile of rows_copy_kernel / rows_convert_kernel when they are launched in order: about 4 KiB of the (wider) row
// side, a power of two between one batch of the kernel (rps x 4 rows) and 64. WM_ROWS_SMALL_TILE=0 keeps 64-row tiles (A/B).
int small_tile_rows(int lpr_log2, int64_t row_bytes)
{
  const char* e = WM_AB_KNOB("WM_ROWS_SMALL_TILE");
  if (e != nullptr && e[0] == '0') return kWave;
  const int batch = (kWave >> lpr_log2) * 4;
  int t           = kWave;
  while (t > batch && static_cast<int64_t>(t) * row_bytes > 4096) t >>= 1;
  return std::max(t, std::min(batch, kWave));
}

// 0 = never, 1 = always when legal, -1 (default) = by the measured rule in want_flat()
int flat_override()
{
  const char* e = WM_KNOB("WM_ROWS_FLAT");
  return e == nullptr ? -1 : atoi(e);
}

// WM_ROWS_STAGED_SCATTER=0 switches the LDS-staged scatter off (A/B: the flat-stream kernel then)
bool staged_scatter_enabled()
{
  const char* e = WM_AB_KNOB("WM_ROWS_STAGED_SCATTER");
  return e == nullptr || e[0] != '0';
}
// longest row the staged kernels take (WM_ROWS_STAGED_MAXROW overrides, A/B)
int64_t staged_max_row(bool gather)
{
  const char* e = WM_AB_KNOB("WM_ROWS_STAGED_MAXROW");
  if (e != nullptr && atoll(e) > 0) return atoll(e);
  (void)gather;
  return 5120;
}
// shortest row the staged kernels are tried on (WM_ROWS_STAGED_MINROW overrides, A/B). Measured against rows_copy_kernel
// (profiles/r03_dim_sweep_flat_forced.csv): gather of ragged rows 132 B 29.8 -> 49.5 % of peak, 164 B 34.3 -> 56.5, 200 B
// 46.8 -> 57.0, 260 B 29.5 -> 58.9, and r03_dim_sweep_staged_small_rows.csv: 36 B 23.4 -> 29.5, 52 B 34.0 -> 45.1, 68 B 28.6 ->
// 40.2, 100 B 33.8 -> 47.6, 120 B 60.9 -> 70.1; scatter 164 B +3.2, 200 B +4.1, 260 B +8.2 points, 132 B and below equal
// or -1; rows of whole 16-byte pieces, scatter: 144 B +0.8, 176 B +2.2, 208 B +3.4, 240 B +5.2, 80 B equal.
bool staged_row_wanted(bool gather, int64_t row_bytes)
{
  const char* e = WM_AB_KNOB("WM_ROWS_STAGED_MINROW");
  if (e != nullptr && atoll(e) > 0) return row_bytes >= atoll(e);
  if (gather) return row_bytes >= 16;
  // (scatter of 64 / 128 / 256 B rows: +1 / +1.2 / +6.6 although 96 B and 112 B lose 1-2: r03_dim_sweep_pow2_small.csv)
  if ((row_bytes & (row_bytes - 1)) == 0 && row_bytes >= 64) return true;
  return row_bytes >= (row_bytes % 16 == 0 ? 144 : 160);
}
// rows of whole 16-byte pieces (no tail) that the flat-stream kernel would take: through the staged kernel too? Measured
// (profiles/r03_dim_sweep_staged_aligned.csv): scatter +1.3 ... +4.3 points on every shape from 400 B to 4000 B, gather mixed
// (+4.7 at 400 B and 1600 B, -0.9 at 544 B and 1200 B, -6.3 at 4000 B; under 512 B always ahead: 48 B +1.8, 112 B +3.5, 176 B
// +5.7, 240 B +5.5, 304 B +3.1, r03_dim_sweep_staged_aligned_small.csv) -> yes for the scatter, under 512 B for the gather.
// WM_ROWS_STAGED_ALIGNED=0 / 1 forces.
bool staged_aligned_rows(bool gather, int64_t row_bytes)
{
  const char* e = WM_AB_KNOB("WM_ROWS_STAGED_ALIGNED");
  if (e != nullptr && (e[0] == '0' || e[0] == '1')) return e[0] == '1';
  return !gather || row_bytes < 512;
}
// WM_ROWS_STAGED=0 switches the LDS-staged gather off (A/B)
bool staged_enabled()
{
  const char* e = WM_AB_KNOB("WM_ROWS_STAGED");
  return e == nullptr || e[0] != '0';
}

// flat-stream kernel or the pow-of-two lane mappings? (rule from experiments/dim_sweep.py, see rows_flat_kernel)
bool want_flat(bool gather, int vb, int64_t row_bytes)
{
  const int ov = flat_override();
  if (ov == 0) return false;
  if (ov == 1) return true;
  const bool pow2 = (row_bytes & (row_bytes - 1)) == 0;
  // 16-byte-multiple rows (interleaved A/B, min of 5 rounds, profiles/r02_dim_sweep_ab*.csv): gather — flat wins for every
  // row that is not a power of two (400 B: 63 vs 57 %, 800 B: 67.5 vs 62 %, 1200 B: 63 vs 52 %); the powers of two,
  // 1 KiB included, go to the readlane kernel with its smaller tiles (1 KiB: 71.4 vs 69.6 %, 2 KiB: 71.7 vs 70.0 %).
  // scatter — flat wins for 400 B (50.7 vs 44.9 %), 416, 448 B (68 vs 56 %) and from 1200 B up (67.7 vs 63.2 %), loses at
  // 800 B (50.6 vs 54.5 %)
  if (vb == 16) return row_bytes > 256 && !pow2 && (gather || row_bytes < 640 || row_bytes > 1024);
  // 4- / 8-byte-multiple rows: 16-byte accesses at 4-byte alignment beat 4- / 8-byte vectors from ~320 B up
  // (508 B: 37 -> 46 %, 516 B: 27 -> 54 %, 2408 B: 42 -> 59 %; 200 B: 44 -> 41 %, so small rows stay on the old path)
  return row_bytes >= 320;
}

template <typename IdxT, bool GATHER>
void launch_flat(const rows_params& p, int blocks, hipStream_t stream)
{
  if (p.row_map != nullptr)
    launch_rows_kernel(rows_flat_kernel<IdxT, GATHER, true>, blocks, stream, p);
  else
    launch_rows_kernel(rows_flat_kernel<IdxT, GATHER, false>, blocks, stream, p);
}

template <typename IdxT, bool GATHER>
void launch_copy(const rows_params& p, int vb, int blocks, hipStream_t stream)
{
  if (p.stage_rows > 0) {
    const int row_bytes   = (p.flat_slots - 1) * 16 + p.flat_tail;
    const size_t lds      = static_cast<size_t>(p.launch_threads / kWave) * ((static_cast<size_t>(p.stage_rows) * row_bytes + 15) & ~size_t(15));
    const bool big = static_cast<size_t>(p.stage_rows) * row_bytes > 5120;   // chunks of up to 10 KiB
#define WM_STAGED(KERNEL)                                                                                   \
  do {                                                                                                      \
    if (big) {                                                                                              \
      t_last_rows_kernel = reinterpret_cast<const void*>(KERNEL<IdxT, 10>);                                 \
      hipLaunchKernelGGL((KERNEL<IdxT, 10>), dim3(blocks), dim3(p.launch_threads), lds, stream, p);         \
    } else {                                                                                                \
      t_last_rows_kernel = reinterpret_cast<const void*>(KERNEL<IdxT, 5>);                                  \
      hipLaunchKernelGGL((KERNEL<IdxT, 5>), dim3(blocks), dim3(p.launch_threads), lds, stream, p);          \
    }                                                                                                       \
  } while (0)
    if constexpr (GATHER)
      WM_STAGED(rows_staged_gather_kernel);
    else
      WM_STAGED(rows_staged_scatter_kernel);
#undef WM_STAGED
    return;
  }
  if (p.flat_slots > 0) {
    launch_flat<IdxT, GATHER>(p, blocks, stream);
    return;
  }
  if (vb == 16 && p.batch_vecs > 0) {  // in-order launch of 512 B / 1 / 2 / 4 KiB rows: the single-batch kernel
    const bool has_map = p.row_map != nullptr;
    const bool by_value = p.chunk_stride != 0;   // rows_op takes this kernel for continuous tables and by-value owner tables only
#define WM_BATCH(RV)                                                                                                  \
  do {                                                                                                                \
    if (has_map) {                                                                                                    \
      if (by_value) launch_rows_kernel(rows_batch_kernel<IdxT, GATHER, RV, true, 1>, blocks, stream, p);              \
      else launch_rows_kernel(rows_batch_kernel<IdxT, GATHER, RV, true, 0>, blocks, stream, p);                       \
    } else {                                                                                                          \
      if (by_value) launch_rows_kernel(rows_batch_kernel<IdxT, GATHER, RV, false, 1>, blocks, stream, p);             \
      else launch_rows_kernel(rows_batch_kernel<IdxT, GATHER, RV, false, 0>, blocks, stream, p);                      \
    }                                                                                                                 \
  } while (0)
    switch (p.batch_vecs) {
      case 32: WM_BATCH(32); break;
      case 64: WM_BATCH(64); break;
      case 128: WM_BATCH(128); break;
      default: WM_BATCH(256); break;
    }
#undef WM_BATCH
    return;
  }
  if (vb == 16 && p.row_vecs >= 32) {  // rows of >= 512 B: readlane fast path
    const bool one_row = p.row_vecs > 32;  // > 512 B: a full wave per row
    const bool has_map = p.row_map != nullptr;
#define WM_FAST(RPS, MAP) \
  launch_rows_kernel(rows_copy16_fast_kernel<IdxT, GATHER, RPS, MAP>, blocks, stream, p)
    if (one_row) {
      if (has_map) WM_FAST(1, true); else WM_FAST(1, false);
    } else {
      if (has_map) WM_FAST(2, true); else WM_FAST(2, false);
    }
#undef WM_FAST
    return;
  }
  switch (vb) {
    case 16: launch_rows_kernel(rows_copy_kernel<IdxT, 16, GATHER>, blocks, stream, p); break;
    case 8: launch_rows_kernel(rows_copy_kernel<IdxT, 8, GATHER>, blocks, stream, p); break;
    case 4: launch_rows_kernel(rows_copy_kernel<IdxT, 4, GATHER>, blocks, stream, p); break;
    case 2: launch_rows_kernel(rows_copy_kernel<IdxT, 2, GATHER>, blocks, stream, p); break;
    default: launch_rows_kernel(rows_copy_kernel<IdxT, 1, GATHER>, blocks, stream, p); break;
  }
}

template <typename TabT, typename PlainT, typename IdxT, bool GATHER>
void launch_convert_v(const rows_params& p, int v, int blocks, hipStream_t stream)
{
  switch (v) {
    case 4:
      launch_rows_kernel(rows_convert_kernel<TabT, PlainT, IdxT, 4, GATHER>, blocks, stream, p);
      break;
    case 2:
      launch_rows_kernel(rows_convert_kernel<TabT, PlainT, IdxT, 2, GATHER>, blocks, stream, p);
      break;
    default:
      launch_rows_kernel(rows_convert_kernel<TabT, PlainT, IdxT, 1, GATHER>, blocks, stream, p);
      break;
  }
}

template <typename TabT, typename PlainT, bool GATHER>
void launch_convert_idx(const rows_params& p, wholememory_dtype_t idx_dt, int v, int blocks, hipStream_t stream)
{
  if (idx_dt == WHOLEMEMORY_DT_INT)
    launch_convert_v<TabT, PlainT, int32_t, GATHER>(p, v, blocks, stream);
  else
    launch_convert_v<TabT, PlainT, int64_t, GATHER>(p, v, blocks, stream);
}

template <typename TabT, bool GATHER>
bool launch_convert_plain_fp(const rows_params& p, wholememory_dtype_t plain_dt, wholememory_dtype_t idx_dt, int v,
                             int blocks, hipStream_t stream)
{
  switch (plain_dt) {
    case WHOLEMEMORY_DT_FLOAT: launch_convert_idx<TabT, float, GATHER>(p, idx_dt, v, blocks, stream); return true;
    case WHOLEMEMORY_DT_HALF: launch_convert_idx<TabT, half_t, GATHER>(p, idx_dt, v, blocks, stream); return true;
    case WHOLEMEMORY_DT_DOUBLE: launch_convert_idx<TabT, double, GATHER>(p, idx_dt, v, blocks, stream); return true;
    case WHOLEMEMORY_DT_BF16: launch_convert_idx<TabT, bf16_t, GATHER>(p, idx_dt, v, blocks, stream); return true;
    default: return false;
  }
}
template <typename TabT, bool GATHER>
bool launch_convert_plain_int(const rows_params& p, wholememory_dtype_t plain_dt, wholememory_dtype_t idx_dt, int v,
                              int blocks, hipStream_t stream)
{
  switch (plain_dt) {
    case WHOLEMEMORY_DT_INT8: launch_convert_idx<TabT, int8_t, GATHER>(p, idx_dt, v, blocks, stream); return true;
    case WHOLEMEMORY_DT_INT16: launch_convert_idx<TabT, int16_t, GATHER>(p, idx_dt, v, blocks, stream); return true;
    case WHOLEMEMORY_DT_INT: launch_convert_idx<TabT, int32_t, GATHER>(p, idx_dt, v, blocks, stream); return true;
    case WHOLEMEMORY_DT_INT64: launch_convert_idx<TabT, int64_t, GATHER>(p, idx_dt, v, blocks, stream); return true;
    default: return false;
  }
}

template <bool GATHER>
bool launch_convert(const rows_params& p, wholememory_dtype_t tab_dt, wholememory_dtype_t plain_dt,
                    wholememory_dtype_t idx_dt, int v, int blocks, hipStream_t s)
{
  switch (tab_dt) {
    case WHOLEMEMORY_DT_FLOAT: return launch_convert_plain_fp<float, GATHER>(p, plain_dt, idx_dt, v, blocks, s);
    case WHOLEMEMORY_DT_HALF: return launch_convert_plain_fp<half_t, GATHER>(p, plain_dt, idx_dt, v, blocks, s);
    case WHOLEMEMORY_DT_DOUBLE: return launch_convert_plain_fp<double, GATHER>(p, plain_dt, idx_dt, v, blocks, s);
    case WHOLEMEMORY_DT_BF16: return launch_convert_plain_fp<bf16_t, GATHER>(p, plain_dt, idx_dt, v, blocks, s);
    case WHOLEMEMORY_DT_INT8: return launch_convert_plain_int<int8_t, GATHER>(p, plain_dt, idx_dt, v, blocks, s);
    case WHOLEMEMORY_DT_INT16: return launch_convert_plain_int<int16_t, GATHER>(p, plain_dt, idx_dt, v, blocks, s);
    case WHOLEMEMORY_DT_INT: return launch_convert_plain_int<int32_t, GATHER>(p, plain_dt, idx_dt, v, blocks, s);
    case WHOLEMEMORY_DT_INT64: return launch_convert_plain_int<int64_t, GATHER>(p, plain_dt, idx_dt, v, blocks, s);
    default: return false;
  }
}

template <bool GATHER>
int rows_op(const wm_rows_args* a, void* stream_v)
{
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  if (a->n == 0 || a->dim == 0) return 0;  // reference gather_func.cu:82
  const int64_t tes = static_cast<int64_t>(wholememory_dtype_get_element_size(a->table_dtype));
  const int64_t pes = static_cast<int64_t>(wholememory_dtype_get_element_size(a->plain_dtype));
  const bool tab_fp = wholememory_dtype_is_floating_number(a->table_dtype);
  const bool pl_fp  = wholememory_dtype_is_floating_number(a->plain_dtype);
  if (tab_fp != pl_fp) return -1;  // reference gather_func.cu:79-81
  if (a->index_dtype != WHOLEMEMORY_DT_INT && a->index_dtype != WHOLEMEMORY_DT_INT64) return -1;

  rows_params p{};
  p.chunk_stride = a->gref.stride;
  p.world_size   = a->gref.world_size;
  p.same_chunk   = a->gref.same_chunk ? 1 : 0;
  if (a->gref.stride == 0) {
    p.base = static_cast<char*>(a->gref.pointer);
  } else {
    p.rank_ptrs    = static_cast<char* const*>(a->gref.pointer);
    p.rank_offsets = a->gref.rank_memory_offsets;
    fill_owner_tables(&p);
  }
  p.table_stride_bytes = a->table_stride * tes;
  p.table_offset_bytes = a->table_storage_offset * tes;
  p.indices            = a->indices;
  p.row_map            = static_cast<const int64_t*>(a->row_map);
  p.n                  = a->n;
  p.plain              = static_cast<char*>(a->plain) + a->plain_storage_offset * pes;
  p.plain_stride_bytes = a->plain_stride * pes;

  // Launch shape (round 3). IN ORDER (the default): one tile per wave and as many workgroups as that takes — the hardware
  // dispatcher hands workgroups out in order, so the tiles in flight are one compact window that advances through the
  // streamed side (the gather's output, the scatter's input), and a tile is ~4 KiB moved as ONE batch (all loads, then all
  // stores). PERSISTENT (when the caller caps the grid: gather_sms / scatter_sms, or WM_ROWS_INORDER=0): round 1-2's
  // grid-stride loop over tiles of up to 64 rows. Measured on the 10 M x 512 B gather (experiments/placement_pmc.hip,
  // profiles/r03_placement_*): the persistent shape runs at 1.70 ... 1.95 ms depending on the PHYSICAL PLACEMENT of the
  // output buffer (its 8192 resident waves each own a 32 KiB tile and the 8192 workgroups sweep the output four times, so
  // the 64-byte write requests of one DRAM page arrive spread over microseconds; a physically contiguous buffer: always
  // slow), the in-order shape at 1.64-1.73 ms on every buffer of every process, the contiguous one included.
  // WM_ROWS_INORDER: 0 = never, 1 = every kernel, unset = the measured rule: every kernel except the flat-stream SCATTER, whose
  // per-tile set-up is too heavy for 4 KiB tiles (profiles/r03_dim_sweep_inorder_ab.csv: 1200 B rows 62 vs 70 % of peak,
  // 516 B 46 vs 51 %, 2408 B 57 vs 60 %; the flat gather gains: 800 B 69 vs 65 %, 2408 B 66 vs 61 %) — it keeps the persistent
  // launch where it is still used. The LDS-staged kernels run in order, one chunk per wave (measured twice on different
  // boxes, profiles/r03_dim_sweep_staged_scatter.csv: gather +6 ... +10 points, scatter +2 ... +3.5 over the persistent grid).
  const int inorder_mode = inorder_setting();
  // (a batch of 2^31 entries or more could need more workgroups than a grid has: the persistent launch loops)
  bool inorder           = a->max_blocks <= 0 && inorder_mode != 0 && a->n < (INT64_C(1) << 31);
  p.launch_threads       = inorder ? inorder_block_threads() : kBlock;
  p.tile_rows            = kWave;
  auto grid_for = [&](int tile_rows) {
    const int64_t tiles = (a->n + tile_rows - 1) / tile_rows;
    const int wpb       = p.launch_threads / kWave;
    // in order = one tile per wave, no loop in the kernel (n < 2^31 there, so the grid always covers the tiles)
    if (inorder) return static_cast<int>(std::min<int64_t>((tiles + wpb - 1) / wpb, INT64_C(0x7fffffff)));
    int b = static_cast<int>(std::min<int64_t>((tiles + wpb - 1) / wpb, default_max_blocks()));
    if (a->max_blocks > 0) b = std::min(b, a->max_blocks);
    return std::max(b, 1);
  };
  int blocks = grid_for(kWave);

  // widest power-of-two access every address on both sides is aligned to — the reference's
  // alignment pick (gather_scatter_func.cuh:215-251) plus the base pointers themselves
  const uint64_t tab_base = a->gref.stride == 0 ? reinterpret_cast<uint64_t>(a->gref.pointer) : 0;
  if (a->table_dtype == a->plain_dtype) {
    const int64_t row_bytes = a->dim * tes;
    int64_t vb              = 16;
    vb                      = pow2_divisor(row_bytes, vb);
    vb                      = pow2_divisor(p.table_stride_bytes, vb);
    vb                      = pow2_divisor(p.table_offset_bytes, vb);
    vb                      = pow2_divisor(static_cast<int64_t>(tab_base & 15), vb);
    vb                      = pow2_divisor(p.plain_stride_bytes, vb);
    vb                      = pow2_divisor(static_cast<int64_t>(reinterpret_cast<uint64_t>(p.plain) & 15), vb);
    p.row_vecs              = static_cast<int>(row_bytes / vb);
    p.lpr_log2              = std::min(6, ilog2_ceil(p.row_vecs));
    // flat-stream kernel: needs every address 4-byte aligned (then 16-byte accesses are legal at any such address)
    const bool dword_ok = vb >= 4 && row_bytes >= 16 && row_bytes < (INT64_C(1) << 24);
    const bool flat     = dword_ok && want_flat(GATHER, static_cast<int>(vb), row_bytes);
    // the staged kernels share the flat kernel's slot geometry; they are also tried on rows the flat kernel does not take
    // (rows of whole 16-byte pieces above 256 B: where the flat rule says no, the readlane kernel is at least as good)
    // and the powers of two from 512 B up have the single-batch kernel: r03_dim_sweep_pow2_staged.csv — 64 / 128 / 256 B rows
    // gain +2.3 / +4.5 / +1.7 (gather) and +3.5 / +1.7 / +4.8 (scatter), 512 B - 4 KiB lose 0 ... 2.3)
    const bool stage_try = dword_ok && staged_row_wanted(GATHER, row_bytes) && (row_bytes % 16 != 0 || row_bytes <= 256);
    if (flat || stage_try) {
      p.flat_slots = static_cast<int>((row_bytes + 15) / 16);
      p.flat_tail  = static_cast<int>(row_bytes - 16 * static_cast<int64_t>(p.flat_slots - 1));
      p.flat_rcp   = 1.0f / static_cast<float>(p.flat_slots);
    }
    // LDS-staged kernels: dense rows only 4 / 8-byte aligned (for the scatter also rows of whole 16-byte pieces), table rows on
    // 16-byte boundaries with room for whole 16-byte accesses (padded stride), no row map (the dense side of consecutive
    // entries must be one contiguous piece). Rows up to 5120 B whose smallest aligned group (1 / 2 / 4 rows) fits a 10 KiB
    // chunk. Measured against the flat-stream kernel (same file): scatter 516 B 39.4 -> 46.1 % of peak on a slow box and
    // 49.9 -> 61.0 on a fast one, 1000 B 42.9 -> 55.0, 1204 B 51.9 -> 63.2, 2408 B 51.4 -> 60.0, 4120 B 56.5 -> 58.7; gather
    // 1032 B 52.0 -> 64.1, 2408 B 60.4 -> 66.6, 4120 B 61.8 -> 68.6.
    if ((GATHER ? staged_enabled() : staged_scatter_enabled()) && p.flat_slots > 0 && (p.flat_tail != 16 || staged_aligned_rows(GATHER, row_bytes)) && (flat || stage_try) &&
        p.row_map == nullptr &&
        p.plain_stride_bytes == row_bytes &&
        (reinterpret_cast<uint64_t>(p.plain) & 15) == 0 && p.table_stride_bytes % 16 == 0 && p.table_offset_bytes % 16 == 0 &&
        (tab_base & 15) == 0 && a->gref.stride % 16 == 0 && p.table_stride_bytes >= static_cast<int64_t>(p.flat_slots) * 16 &&
        row_bytes <= staged_max_row(GATHER)) {
      const int need    = row_bytes % 16 == 0 ? 1 : row_bytes % 8 == 0 ? 2 : 4;   // rows per 16-byte-aligned piece of the dense stream
      const int64_t cap = static_cast<int64_t>(need) * row_bytes <= 5120 ? 5120 : 10240;   // 5 or 10 x 1 KiB per wave
      int R             = 64;
      while (R > need && static_cast<int64_t>(R) * row_bytes > cap) R >>= 1;
      if (static_cast<int64_t>(R) * row_bytes <= cap) {   // bigger rows stay on the flat kernel
        p.stage_rows = R;
        {
          const char* sa = WM_AB_KNOB("WM_ROWS_STAGED_ALIGN_STORES");
          p.stage_align  = (sa != nullptr && sa[0] == '0') ? 0 : 1;
        }
        if (inorder_mode == 0 || a->max_blocks > 0 || a->n >= (INT64_C(1) << 31)) {
          inorder          = false;
          p.launch_threads = kBlock;
        } else {          // in order: one chunk per wave (the flat branch above may have switched it off for the scatter)
          inorder          = true;
          p.launch_threads = inorder_block_threads();
        }
        p.tile_rows = inorder ? R : kWave;
        blocks      = grid_for(p.tile_rows);
      }
    }
    if (p.stage_rows == 0 && p.flat_slots > 0) {
      if (!flat) {
        p.flat_slots = 0, p.flat_tail = 0;            // not staged after all: the generic kernels
      } else if (!GATHER && inorder_mode < 0 && inorder) {   // flat-stream scatter: persistent unless forced
        inorder          = false;
        p.launch_threads = kBlock;
      }
    }
    if (p.stage_rows == 0 && (p.flat_slots > 0 || (vb == 16 && p.row_vecs >= 32))) {  // the two kernels that take tile_rows
      const char* te    = WM_AB_KNOB("WM_ROWS_TILE");  // experiment switch
      const int forced  = te != nullptr ? atoi(te) : 0;
      p.tile_rows = row_bytes <= 768 ? 64 : row_bytes <= 1536 ? 32 : row_bytes <= 3072 ? 16 : 8;
      if (inorder) {
        if (p.flat_slots > 0) {   // flat stream: a batch is 4 x 64 slots of 16 bytes; the rows that fill one batch, or two when
          const int S     = p.flat_slots;   // one would stay under 80 % full (2408 B rows = 151 slots: 1 row 59 %, 3 rows 88 % of two)
          const int r1    = std::max(1, 256 / S), r2 = std::max(1, 512 / S);
          const double f1 = r1 * S / 256.0, f2 = r2 * S / 512.0;
          p.tile_rows     = std::min(kWave, (S > 256 || f1 >= 0.8 || f2 <= f1) ? r1 : r2);
          // gather of rows of whole 16-byte pieces: 8 rows per tile whatever S is. The dense side of a tile is then a multiple
          // of 128 bytes and every wave store a whole, aligned 1 KiB of it; with tiles that begin on 32-byte multiples each
          // wave writes two partial lines (4000 B rows on a packed output 64 % against 71 % on an output padded to 4 KiB,
          // profiles/r04_misaligned_rows.txt). Measured: 528 B +2.3, 640 B +2.0, 800 B +2.2, 2000 B +3.0, 4000 B +2.0 points,
          // 960 / 1200 / 1600 B within +-0.7. (WM_ROWS_FLAT_TILE8=0: the batch-filling rule above, A/B)
          const char* t8 = WM_AB_KNOB("WM_ROWS_FLAT_TILE8");
          if (GATHER && vb == 16 && p.row_map == nullptr && !(t8 != nullptr && t8[0] == '0')) p.tile_rows = 8;
        } else {                  // readlane kernel: (tile_rows / RPS) x chunks steps, a multiple of its 4-step batch
          const int chunks = p.row_vecs > 32 ? (p.row_vecs + kWave - 1) / kWave : 1;
          p.tile_rows      = p.row_vecs == 32 ? 8 : chunks == 1 ? 4 : chunks == 2 ? 2 : chunks % 4 == 0 ? 1 : 4;
          // 512 B / 1 / 2 / 4 KiB rows: exactly one 4 KiB batch per tile -> the specialised kernel (WM_ROWS_BATCH=0: A/B)
          const char* be = WM_AB_KNOB("WM_ROWS_BATCH");
          // (continuous tables and owner tables by value; plain rows less than 2 GiB apart: 32-bit lane offsets)
          if ((p.row_vecs == 32 || p.row_vecs == 64 || p.row_vecs == 128 || p.row_vecs == 256) && forced == 0 &&
              (p.chunk_stride == 0 || p.owners_by_value) && p.plain_stride_bytes < (INT64_C(1) << 31) &&
              !(be != nullptr && be[0] == '0')) {
            p.batch_vecs = p.row_vecs;
            // one wave per workgroup for this kernel: the finest unit the dispatcher can hand out (measured against 256
            // threads on 512 B - 4 KiB rows: gather +0.2 ... +1.4 %, scatter +0.5 ... +1 %; the flat kernel loses 2-3 % with it)
            if (WM_AB_KNOB("WM_ROWS_BLOCK") == nullptr) p.launch_threads = kWave;
          }
        }
      }
      if ((forced == 8 || forced == 16 || forced == 32 || forced == 64) && (p.flat_slots > 0 || forced % 8 == 0)) p.tile_rows = forced;
      blocks = grid_for(p.tile_rows);
    }
    if (inorder && p.stage_rows == 0 && p.flat_slots == 0 && !(vb == 16 && p.row_vecs >= 32)) {   // rows_copy_kernel
      p.tile_rows = small_tile_rows(p.lpr_log2, row_bytes);
      blocks      = grid_for(p.tile_rows);
    }
    if (a->index_dtype == WHOLEMEMORY_DT_INT)
      launch_copy<int32_t, GATHER>(p, static_cast<int>(vb), blocks, stream);
    else
      launch_copy<int64_t, GATHER>(p, static_cast<int>(vb), blocks, stream);
  } else {
    // (measured in round 4 on the 16 <-> 32-bit pairs, interleaved on one box: 8 elements per lane (16 / 32 bytes) 62-63 % of
    // peak against 65 % with 4; 8 steps of 8-byte loads in flight instead of 4: equal within the process spread. 4 x 4 stay.)
    int64_t v = 4;
    v         = pow2_divisor(a->dim, v);
    v         = pow2_divisor(a->table_stride, v);
    v         = pow2_divisor(a->table_storage_offset, v);
    v         = pow2_divisor(a->plain_stride, v);
    while (v > 1 && ((tab_base % (v * tes)) != 0 || (reinterpret_cast<uint64_t>(p.plain) % (v * pes)) != 0)) v >>= 1;
    p.row_vecs = static_cast<int>(a->dim / v);
    p.lpr_log2 = std::min(6, ilog2_ceil(p.row_vecs));
    if (inorder) {   // rows_convert_kernel: ~4 KiB of the wider side per wave
      p.tile_rows = small_tile_rows(p.lpr_log2, a->dim * std::max(tes, pes));
      blocks      = grid_for(p.tile_rows);
    }
    if (!launch_convert<GATHER>(p, a->table_dtype, a->plain_dtype, a->index_dtype, static_cast<int>(v), blocks, stream))
      return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

int hip_gather_rows(const wm_rows_args* a, void* stream) { return rows_op<true>(a, stream); }
int hip_scatter_rows(const wm_rows_args* a, void* stream) { return rows_op<false>(a, stream); }

}  // namespace wm

// name (demangled) the HIP runtime holds for the row kernel this thread launched last; "" before the first launch
extern "C" const char* wholememory_ext_last_rows_kernel()
{
  thread_local std::string name;
  name.clear();
  if (wm::t_last_rows_kernel == nullptr) return name.c_str();
  const char* mangled = hipKernelNameRefByPtr(wm::t_last_rows_kernel, nullptr);
  if (mangled == nullptr) return name.c_str();
  int status       = 0;
  char* demangled  = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
  name             = status == 0 && demangled != nullptr ? demangled : mangled;
  if (demangled != nullptr) free(demangled);
  return name.c_str();
}
