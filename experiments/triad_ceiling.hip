// Round 6, review item 5: what does this memory system give a kernel with step_tile_kernel's traffic mix — two row reads and one
// row write per 512-byte row, the write going back to one of the rows read — when the access pattern is ideal (sequential)?
// Plain grid-stride streaming kernels, 16 bytes per lane, non-temporal on every side, 10 M rows of 512 B per stream:
//   copy      : out[i] = a[i]                      (1 read + 1 write: the level the row kernels reach, ~80 %)
//   read2     : sink += a[i] + b[i]                (2 reads)
//   triad     : out[i] = a[i] - lr * b[i]          (2 reads + 1 write, three buffers)
//   update    : a[i]   = a[i] - lr * b[i]          (2 reads + 1 write, in place: the optimizer step's mix)
//   update_t  : the same, a wave owning a TILE of 64 consecutive rows and streaming it 4 rows x 2 loads at a time, all loads of
//               a batch before its arithmetic and stores — step_tile_kernel's inner loop without its metadata
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 experiments/triad_ceiling.hip -o experiments/triad_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void stream_kernel(f4* a, const f4* b, f4* out, int64_t n4, float lr, float* sink)
{
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  f4 acc{0, 0, 0, 0};
  constexpr int U = 4;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += U * stride) {
    f4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t j = min(i + u * stride, n4 - 1);
      x[u] = __builtin_nontemporal_load(a + j);
      if (MODE != 0) y[u] = __builtin_nontemporal_load(b + j);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t j = i + u * stride;
      if (MODE == 1) acc += x[u] + y[u];
      if (j < n4) {
        if (MODE == 0) __builtin_nontemporal_store(x[u], out + j);
        if (MODE == 2) __builtin_nontemporal_store(x[u] - lr * y[u], out + j);
        if (MODE == 3) __builtin_nontemporal_store(x[u] - lr * y[u], a + j);
      }
    }
  }
  if (MODE == 1 && acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

// a wave owns tiles of 64 consecutive 512-byte rows; per batch: 4 steps x (2 rows per step: half-waves) = 8 rows, 8 loads in flight
__global__ __launch_bounds__(256) void update_tile_kernel(f4* a, const f4* b, int64_t rows, float lr, int in_order)
{
  const int lane = threadIdx.x & 63;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) >> 6;
  const int64_t n_waves = (static_cast<int64_t>(gridDim.x) * 256) >> 6;
  const int64_t tiles = (rows + 63) / 64;
  const int sub = lane >> 5, col = lane & 31;
  for (int64_t t = wave; t < tiles; t += n_waves) {
    for (int s = 0; s < 64; s += 8) {
      f4 x[4], y[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int64_t r = min(t * 64 + s + 2 * k + sub, rows - 1);
        x[k] = __builtin_nontemporal_load(a + r * 32 + col);
        y[k] = __builtin_nontemporal_load(b + r * 32 + col);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int64_t r = min(t * 64 + s + 2 * k + sub, rows - 1);
        __builtin_nontemporal_store(x[k] - lr * y[k], a + r * 32 + col);
      }
    }
  }
}

// the launch shape of the row kernels (rows.hip): one-wave workgroups IN ORDER, each 8 consecutive rows (4 KiB per stream)
template <int MODE>   // 0: copy a -> out, 3: update in place
__global__ __launch_bounds__(64) void inorder_kernel(f4* a, const f4* b, f4* out, int64_t rows, float lr)
{
  const int lane = threadIdx.x, sub = lane >> 5, col = lane & 31;
  f4 x[4], y[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int64_t r = min(static_cast<int64_t>(blockIdx.x) * 8 + 2 * k + sub, rows - 1);
    x[k] = __builtin_nontemporal_load(a + r * 32 + col);
    if (MODE == 3) y[k] = __builtin_nontemporal_load(b + r * 32 + col);
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int64_t r = min(static_cast<int64_t>(blockIdx.x) * 8 + 2 * k + sub, rows - 1);
    if (MODE == 0) __builtin_nontemporal_store(x[k], out + r * 32 + col);
    if (MODE == 3) __builtin_nontemporal_store(x[k] - lr * y[k], a + r * 32 + col);
  }
}

int main(int argc, char** argv)
{
  const int64_t rows = argc > 1 ? atoll(argv[1]) : 10000000;
  const int64_t n4 = rows * 32;
  f4 *a, *b, *c; float* sink;
  CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16)); CK(hipMalloc(&c, n4 * 16)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 0, n4 * 16)); CK(hipMemset(b, 0, n4 * 16)); CK(hipMemset(c, 0, n4 * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, double bytes, auto launch) {
    float best = 1e30f, sum = 0;
    for (int r = 0; r < 12; r++) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) { best = std::min(best, ms); sum += ms; }
    }
    printf("%-52s avg %.4f ms  best %.4f ms  -> %.0f GB/s = %.1f %% of 8 TB/s (avg)\n", name, sum / 10, best, bytes / (sum / 10) / 1e6, bytes / (sum / 10) / 1e6 / 80);
  };
  const double rb = rows * 512.0;
  run("copy      in order, 8 rows per one-wave workgroup", 2 * rb, [&] { hipLaunchKernelGGL(inorder_kernel<0>, dim3((rows + 7) / 8), dim3(64), 0, 0, a, b, c, rows, 0.01f); });
  run("update    in order, 8 rows per one-wave workgroup", 3 * rb, [&] { hipLaunchKernelGGL(inorder_kernel<3>, dim3((rows + 7) / 8), dim3(64), 0, 0, a, b, c, rows, 0.01f); });
  for (int grid : {2048, 8192, 32768}) {
    char nm[96];
    snprintf(nm, sizeof nm, "copy      (1R + 1W)            grid %5d", grid);
    run(nm, 2 * rb, [&] { hipLaunchKernelGGL(stream_kernel<0>, dim3(grid), dim3(256), 0, 0, a, b, c, n4, 0.01f, sink); });
    snprintf(nm, sizeof nm, "read2     (2R)                 grid %5d", grid);
    run(nm, 2 * rb, [&] { hipLaunchKernelGGL(stream_kernel<1>, dim3(grid), dim3(256), 0, 0, a, b, c, n4, 0.01f, sink); });
    snprintf(nm, sizeof nm, "triad     (2R + 1W, 3 buffers) grid %5d", grid);
    run(nm, 3 * rb, [&] { hipLaunchKernelGGL(stream_kernel<2>, dim3(grid), dim3(256), 0, 0, a, b, c, n4, 0.01f, sink); });
    snprintf(nm, sizeof nm, "update    (2R + 1W, in place)  grid %5d", grid);
    run(nm, 3 * rb, [&] { hipLaunchKernelGGL(stream_kernel<3>, dim3(grid), dim3(256), 0, 0, a, b, c, n4, 0.01f, sink); });
    snprintf(nm, sizeof nm, "update_t  (tiles of 64 rows)   grid %5d", grid);
    run(nm, 3 * rb, [&] { hipLaunchKernelGGL(update_tile_kernel, dim3(grid), dim3(256), 0, 0, a, b, rows, 0.01f, 0); });
  }
  return 0;
}
