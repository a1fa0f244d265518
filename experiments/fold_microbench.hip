// What does the long-run folder wave (optim.hip, step_long4_kernel) cost per row? One wave per workgroup:
//   (a) N dependent v_add_f32 from registers                 -> cycles per dependent add
//   (b) 64 LDS reads (hipcc pairs them into ds_read2st64_b32) + wait + 64 dependent adds per 64 rows: the fold loop
//   (c) the reads alone
// clock64() (shader clock) and wall_clock64() (100 MHz) per row are printed next to the event time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k_adds(float* out, int n, long long* t)
{
  float acc    = threadIdx.x;
  float v      = out[threadIdx.x];
  long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int k = 0; k < 16; k++) acc += v;
    asm volatile("" : "+v"(acc));
  }
  long long c1 = clock64(), w1 = wall_clock64();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) t[blockIdx.x * 2] = c1 - c0, t[blockIdx.x * 2 + 1] = w1 - w0;
}

// hand-pipelined fold: DEPTH two-row reads in flight, each pair of adds waits for the oldest only
template <int DEPTH>
__global__ void k_fold_pipe(float* out, int tiles, long long* t)
{
  __shared__ float lds[128 * 64];
  for (int i = threadIdx.x; i < 128 * 64; i += 64) lds[i] = i * 1e-9f;
  __syncthreads();
  float acc = 0.f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) const float lds_t;
  const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_t*)(lds + threadIdx.x)));
  long long c0 = clock64(), w0 = wall_clock64();
  for (int tt = 0; tt < tiles; tt++) {
    f2 b[DEPTH + 1];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define RD2(slot, pair) \
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(b[slot]) : "v"(base), "n"(2 * (pair)), "n"(2 * (pair) + 1))
#pragma unroll
    for (int j = 0; j < DEPTH; j++) RD2(j, j);
#pragma unroll
    for (int j = 0; j < 64; j++) {
      const int left = 64 - j < DEPTH ? 64 - j : DEPTH;
      switch (left - 1) {
        case 14: asm volatile("s_waitcnt lgkmcnt(14)" : "+v"(b[j % (DEPTH + 1)])); break;
        case 13: asm volatile("s_waitcnt lgkmcnt(13)" : "+v"(b[j % (DEPTH + 1)])); break;
        case 12: asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(b[j % (DEPTH + 1)])); break;
        case 11: asm volatile("s_waitcnt lgkmcnt(11)" : "+v"(b[j % (DEPTH + 1)])); break;
        case 10: asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(b[j % (DEPTH + 1)])); break;
        case 9: asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(b[j % (DEPTH + 1)])); break;
        case 8: asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(b[j % (DEPTH + 1)])); break;
        case 7: asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(b[j % (DEPTH + 1)])); break;
        case 6: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(b[j % (DEPTH + 1)])); break;
        case 5: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(b[j % (DEPTH + 1)])); break;
        case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(b[j % (DEPTH + 1)])); break;
        case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(b[j % (DEPTH + 1)])); break;
        case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(b[j % (DEPTH + 1)])); break;
        case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(b[j % (DEPTH + 1)])); break;
        default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[j % (DEPTH + 1)])); break;
      }
      acc += b[j % (DEPTH + 1)].x;
      acc += b[j % (DEPTH + 1)].y;
      if (j + DEPTH < 64) RD2((j + DEPTH) % (DEPTH + 1), j + DEPTH);
    }
#undef RD2
    asm volatile("" : "+v"(acc));
  }
  long long c1 = clock64(), w1 = wall_clock64();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) t[blockIdx.x * 2] = c1 - c0, t[blockIdx.x * 2 + 1] = w1 - w0;
}

template <int MODE>
__global__ void k_fold(float* out, int tiles, long long* t)
{
  __shared__ float lds[128 * 64];
  for (int i = threadIdx.x; i < 128 * 64; i += 64) lds[i] = i * 1e-9f;
  __syncthreads();
  float acc        = 0.f;
  const float* src = lds + threadIdx.x;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int tt = 0; tt < tiles; tt++) {
#pragma unroll 1
    for (int r = 0; r < 128; r += 64) {
      float v[64];
#pragma unroll
      for (int k = 0; k < 64; k++) v[k] = src[(r + k) * 64];
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 64; k++) acc += v[k];
      } else {
        float m = 0;
#pragma unroll
        for (int k = 0; k < 64; k += 16) m = fmaxf(m, v[k]);  // keep the reads alive, almost no dependent work
        acc += m;
#pragma unroll
        for (int k = 0; k < 64; k++) asm volatile("" ::"v"(v[k]));
      }
    }
    asm volatile("" : "+v"(acc));
  }
  long long c1 = clock64(), w1 = wall_clock64();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) t[blockIdx.x * 2] = c1 - c0, t[blockIdx.x * 2 + 1] = w1 - w0;
}

int main()
{
  float* out;
  long long* t;
  hipMalloc(&out, 4096);
  hipMemset(out, 0, 4096);
  hipMalloc(&t, 8192);
  long long h[4];
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto report = [&](const char* name, double rows) {
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-34s %8.3f ms  %6.2f ns/row  clock64 %6.2f /row  wall_clock64 %6.2f /row\n", name, ms, ms * 1e6 / rows,
           h[0] / rows, h[1] / rows);
  };
  for (int rep = 0; rep < 2; rep++) {
    const int n = 1 << 22;
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_adds, dim3(2), dim3(64), 0, 0, out, n, t);
    report("dependent v_add_f32", n);
    const int tiles = 1 << 15;
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_fold<0>, dim3(2), dim3(64), 0, 0, out, tiles, t);
    report("fold: 64 LDS reads + 64 adds", tiles * 128.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_fold<1>, dim3(2), dim3(64), 0, 0, out, tiles, t);
    report("reads only", tiles * 128.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_fold_pipe<15>, dim3(2), dim3(64), 0, 0, out, tiles, t);
    report("pipelined fold, 15 in flight", tiles * 128.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_fold_pipe<8>, dim3(2), dim3(64), 0, 0, out, tiles, t);
    report("pipelined fold, 8 in flight", tiles * 128.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_fold_pipe<4>, dim3(2), dim3(64), 0, 0, out, tiles, t);
    report("pipelined fold, 4 in flight", tiles * 128.0);
  }
  return 0;
}
