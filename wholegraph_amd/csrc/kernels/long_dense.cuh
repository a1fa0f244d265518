// wholegraph_amd — the ORDERED fold of very long duplicate runs through a dense, transposed copy (gfx950 HIP, wave64). Round 6.
//
// What it computes (reference: exchange_embeddings_nccl_func.cu:76-103, DedupIndiceAndGradientsKernel): the gradient rows of
// one id summed one by one IN RECEIVE ORDER, per column a chain of dependent fp32 adds — 527 k of them for the hottest id of a
// Zipf(1.05) batch of 10 M ids. A dependent v_add_f32 issues every 6.25 cycles on this part (profiles/r05_fold_floor.txt), so
// the chain cannot take less than 1.38 ms; step_long4_kernel (optim.hip) pays 10.5 cycles per row: its folding wave issues
// one LDS read per TWO rows (ds_read2st64_b32, rows are 128 bytes apart in LDS) at ~12 cycles per instruction, its rows come
// at random from the gradient buffer through one CU (order[] entries staged in LDS chunk by chunk), and the chain changes
// waves every 128 rows. Here the run is first COPIED — by the whole chip, at memory speed — into a dense buffer laid out for
// the fold:
//
//   dense[g][c][k] = gradient row order[s0 + 4 g + k], column c        (k = 0 .. 3; rows past the end of the run read -0.0f)
//
// i.e. the four consecutive rows of a column are 16 adjacent bytes. A workgroup of the fold owns a slice of 32 columns; its
// producer waves stream the slice's 512 contiguous bytes per row group into an LDS ring (global_load_lds_dwordx4: a
// sequential fetch with no order[] look-ups, no clamping — the copy did that), and a folding lane gets FOUR rows of its column
// with one ds_read_b128 — a quarter of the LDS instructions per row. Two folding waves take turns tile by tile (one adds the
// rows it holds in registers while the other reads its next tile), R rows per turn. Summation order per element is the
// receive order, starting from -0.0f (== copying the first row; x + (-0.0f) == x bit for bit for every x), so results are
// bit-identical to the reference's and to step_long4_kernel's.
//
// Measured alone (experiments/fold5_harness.hip, profiles/r06_fold5_harness.txt): copy of 0.65 M rows 0.15 ms (4.4 TB/s in + out),
// fold of the 527 k-row run 1.28 ms at R = 128 (5.8 cycles per row at 2.4 GHz; step_long4_kernel: 2.32 ms), 1.20 ms at R = 192
// (spills), 1.52 ms at R = 64.
// This header holds the two parts free of the optimizer's types, so that the harness times them alone: the row address of an
// order[] entry and what happens to a finished column come in as functors (optim.hip: step_dense_copy_kernel / _fold_kernel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wm {
namespace dense_fold {

struct job {           // one listed long run
  int32_t s0;          // first index into order[]
  int32_t rows;        // run length
  int64_t dense_off;   // first float of the run's dense copy (groups x dim x 4 floats), or -1: no room, another kernel folds it
  int32_t user;        // the caller's (optim.hip: index into the long-run list)
  int32_t pad;
};

typedef float f4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline int64_t groups_of(int64_t rows) { return (rows + 3) / 4; }
// floats of the dense copy of a run (dim = a multiple of 4)
__host__ __device__ inline int64_t dense_floats(int64_t rows, int64_t dim) { return groups_of(rows) * dim * 4; }

// ---- the copy: rows of the listed runs -> dense[g][c][k] ---------------------------------------------------------------------
// A half-wave (32 lanes) owns one row group at a time: lane c4 loads 16 bytes (columns 4 c4 .. 4 c4 + 3) of each of the
// group's four rows — four whole-row reads of 512 B for dim = 128 —, transposes 4 x 4 in registers and writes 64 contiguous
// bytes (its four columns' k = 0 .. 3): 2 KiB contiguous per half-wave. Groups are dealt out to the half-waves of the grid in
// turn, run after run, so one giant run spreads over the whole chip.
template <typename RowOf>
__device__ __forceinline__ void copy_part(const job* jobs, int n, const int32_t* order, const RowOf& row_of, int dim, float* dense,
                                          int64_t hw /* this half-wave among all */, int64_t nhw, int c4l /* lane of the half-wave */)
{
  const int quads    = dim >> 2;
  int64_t first      = 0;                                                 // groups of the runs before job j (dealing position)
  constexpr int U    = 2;                                                 // row groups a half-wave has in flight (8 row loads per lane)
  for (int j = 0; j < n; j++) {
    const job jb = jobs[j];
    const int64_t G = groups_of(jb.rows);
    if (jb.dense_off >= 0) {
      // the first group of this run that is this half-wave's: (first + g) % nhw == hw; then every nhw-th, U at a time
      int64_t g = (hw - first % nhw + nhw) % nhw;
      for (; g < G; g += U * nhw) {
        int32_t o[U][4];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int64_t r = 4 * (g + u * nhw) + k;
            o[u][k]         = order[jb.s0 + (r < jb.rows ? r : jb.rows - 1)];   // (groups past the end re-read the last row: not stored)
          }
        for (int c4 = c4l; c4 < quads; c4 += 32) {
          f4 v[U][4];
#pragma unroll
          for (int u = 0; u < U; u++)
#pragma unroll
            for (int k = 0; k < 4; k++) v[u][k] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(row_of(o[u][k]) + 4 * c4));
#pragma unroll
          for (int u = 0; u < U; u++) {
            const int64_t gu = g + u * nhw;
            if (gu >= G) continue;
#pragma unroll
            for (int k = 0; k < 4; k++)
              if (4 * gu + k >= jb.rows) v[u][k] = f4{-0.0f, -0.0f, -0.0f, -0.0f};
            f4* dst = reinterpret_cast<f4*>(dense + jb.dense_off) + (gu * dim + 4 * c4);
            dst[0]  = f4{v[u][0].x, v[u][1].x, v[u][2].x, v[u][3].x};
            dst[1]  = f4{v[u][0].y, v[u][1].y, v[u][2].y, v[u][3].y};
            dst[2]  = f4{v[u][0].z, v[u][1].z, v[u][2].z, v[u][3].z};
            dst[3]  = f4{v[u][0].w, v[u][1].w, v[u][2].w, v[u][3].w};
          }
        }
      }
    }
    first += G;
  }
}
template <typename RowOf>
__global__ __launch_bounds__(256) void copy_kernel(const job* jobs, const int32_t* n_jobs, const int32_t* order, RowOf row_of,
                                                   int dim, float* dense)
{
  copy_part(jobs, *n_jobs, order, row_of, dim, dense, static_cast<int64_t>(blockIdx.x) * 8 + (threadIdx.x >> 5),
            static_cast<int64_t>(gridDim.x) * 8, static_cast<int>(threadIdx.x & 31));
}

// ---- the fold ---------------------------------------------------------------------------------------------------------------
constexpr int kProducers = 4;                          // waves that only fetch
constexpr int kFolders   = 2;                          // waves 0 and 1 fold alternate tiles
constexpr int kBlock     = 64 * (kProducers + kFolders);
constexpr int kRingBytes = 128 * 1024;

// R = rows a folding wave holds in registers per turn, S = columns of a workgroup's slice (32, 16 or 8: 16 S bytes per row
// group). The ring is what covers the memory latency — 128 KiB are 8 tiles of 128 rows x 32 columns, 2.5 us of the chain: enough
// alone, NOT beside the optimizer step's tile kernel (the fold of the 527 k-row run: 1.28 ms alone, 2.2 ms beside it,
// profiles/r06_grad_timeline_zipf_dense_first.txt) — so the product folds 8-column slices: a quarter of the bytes per row and
// workgroup, four times the rows in flight (10 us), four times the workgroups per run.
template <int R, int S>
struct shape {
  static_assert(S == 32 || S == 16 || S == 8, "a row group of a slice is 512, 256 or 128 bytes");
  static constexpr int kRowsPerInstr = 256 / S;                            // rows one LDS-DMA wave instruction brings (1 KiB)
  static_assert(R % (kRowsPerInstr * kProducers) == 0, "a tile is a whole number of instructions per producer");
  static constexpr int kTileBytes = R * S * 4;
  static constexpr int kLoads     = R / kRowsPerInstr / kProducers;        // LDS-DMA pieces per producer lane per tile
  static constexpr int kRingFit   = kRingBytes / kTileBytes;
  static constexpr int kRing      = kRingFit * kLoads < 64 ? kRingFit : 63 / kLoads;   // (vmcnt counts at most 63 pieces in flight)
  static_assert(kRing >= 3, "tile t in registers, t + 1 landed, t + 2 in flight");
  static constexpr size_t kLdsBytes = static_cast<size_t>(kRing) * kTileBytes + 64 * 4;
};
__host__ __device__ inline int slices_of(int dim, int s) { return (dim + s - 1) / s; }

// Epilogue: ep(job, column, sum) for every column of every listed run with a dense copy. Work items = (run, slice) pairs,
// dealt to the workgroups `first_item`, `first_item + item_stride`, ...
template <int R, int S, typename Epilogue>
__device__ __forceinline__ void fold_part(const job* jobs, int n, int dim, const float* dense, const Epilogue& ep, int first_item,
                                          int item_stride, float* lds5)
{
  typedef shape<R, S> sh;
  float* const tiles = lds5;                                                    // [kRing][R / 4][S][4]
  float* const acc_s = lds5 + sh::kRing * (sh::kTileBytes / 4);                 // [64] running sums between the two folders
  const int lane     = threadIdx.x & 63;
  const int wave_id  = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const bool producer = wave_id >= kFolders;
  const int wv       = wave_id - kFolders;
  const int slices   = slices_of(dim, S);
  constexpr int kGpi = 64 / S;                     // row groups per LDS-DMA instruction
  const int p_sub    = lane / S;                   // producer lane: row group p_sub of its piece, column lane % S of the slice

  for (int item = first_item; item < n * slices; item += item_stride) {
    const job jb = jobs[item / slices];
    if (jb.dense_off < 0) continue;
    const int col0    = (item % slices) * S;
    const int cols    = min(S, dim - col0);                                      // a multiple of 4
    const int p_col   = min(lane & (S - 1), cols - 1);                           // lanes past a narrow slice re-read its last column
    const bool folder = lane < cols;
    const int64_t G   = groups_of(jb.rows);
    const int n_tiles = static_cast<int>((4 * G + R - 1) / R);
    const f4* src0    = reinterpret_cast<const f4*>(dense + jb.dense_off) + col0 + p_col;   // group g: + g * dim
    __syncthreads();   // previous run: every tile folded, every LDS-DMA piece landed (vmcnt(0) below)
    auto issue = [&](int t) {
      float* slot = tiles + (t % sh::kRing) * (sh::kTileBytes / 4);
#pragma unroll
      for (int i = 0; i < sh::kLoads; i++) {
        const int gi     = kGpi * (wv + kProducers * i);                                    // first row group of this wave's piece inside the tile
        const int64_t g  = min(static_cast<int64_t>(t) * (R / 4) + gi + p_sub, G - 1);       // tiles past the end re-read the last group
        const f4* src    = src0 + g * dim;
        typedef __attribute__((address_space(1))) void gvoid;
        typedef __attribute__((address_space(3))) void lvoid;
        __builtin_amdgcn_global_load_lds((gvoid*)src, (lvoid*)(slot + gi * (S * 4)), 16, 0, 0);
      }
    };
    float v[R];
    auto read_tile = [&](int t) {
      const f4* src = reinterpret_cast<const f4*>(tiles + (t % sh::kRing) * (sh::kTileBytes / 4)) + (lane & (S - 1));
#pragma clang loop unroll(full)
      for (int q = 0; q < R / 4; q++) {
        const f4 x = src[q * S];
        v[4 * q + 0] = x.x, v[4 * q + 1] = x.y, v[4 * q + 2] = x.z, v[4 * q + 3] = x.w;
      }
      const int64_t rows_left = 4 * G - static_cast<int64_t>(t) * R;   // (the copy padded the run to whole groups with -0.0f)
      if (rows_left < R) {
#pragma clang loop unroll(full)
        for (int k = 0; k < R; k++) v[k] = k < rows_left ? v[k] : -0.0f;
      }
    };
    float acc = 0.f;
    if (producer) {
#pragma unroll
      for (int j = 0; j < sh::kRing; j++) issue(j);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(sh::kLoads * (sh::kRing - 1)) : "memory");   // tile 0 has landed
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      for (int tt = 0; tt < n_tiles; tt++) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(sh::kLoads * (sh::kRing - 2)) : "memory");  // tile tt + 1 has landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(tt + sh::kRing);
      }
    } else {
      auto tile_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      };
      auto add_tile = [&](int t) {
        acc = t > 0 ? acc_s[lane] : -0.0f;
#pragma clang loop unroll(full)
        for (int k = 0; k < R; k++) acc += v[k];
        if (t + 1 < n_tiles) acc_s[lane] = acc;
      };
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (wave_id == 0) {   // even tiles
        read_tile(0);
        for (int tt = 0; tt < n_tiles; tt += 2) {
          tile_barrier();
          add_tile(tt);
          if (tt + 1 < n_tiles) {
            tile_barrier();
            if (tt + 2 < n_tiles) read_tile(tt + 2);
          }
        }
      } else {              // odd tiles
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
    if (producer) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // clamped tail tiles must not land in the next run
    if (wave_id == ((n_tiles - 1) & 1) && folder) ep(jb, col0 + lane, acc);
  }
}
template <int R, int S, typename Epilogue>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(1, 2)))
void fold_kernel(const job* jobs, const int32_t* n_jobs, int dim, const float* dense, Epilogue ep)
{
  extern __shared__ __attribute__((aligned(16))) float lds5[];
  fold_part<R, S>(jobs, *n_jobs, dim, dense, ep, blockIdx.x, gridDim.x, lds5);
}

}  // namespace dense_fold
}  // namespace wm
