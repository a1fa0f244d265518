// What a dependent launch costs on this part, whatever the kernel does: N empty kernels (1 workgroup / 256 workgroups / 4096
// workgroups of 256 threads) back to back on one stream, wall time per launch; then the same with a 4-byte store per thread.
// hipcc --offload-arch=gfx950 -O2 experiments/launch_floor.hip -o experiments/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void empty_kernel() {}
__global__ void store_kernel(int* p) { p[blockIdx.x * blockDim.x + threadIdx.x] = threadIdx.x; }
int main()
{
  int* d = nullptr;
  hipMalloc(&d, 4096 * 256 * 4);
  hipStream_t s;
  hipStreamCreate(&s);
  const int grids[3] = {1, 256, 4096};
  for (int mode = 0; mode < 2; mode++)
    for (int g : grids) {
      for (int rep = 0; rep < 3; rep++) {
        const int n = 2000;
        for (int i = 0; i < 50; i++) hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(256), 0, s);
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; i++) {
          if (mode == 0) hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(256), 0, s);
          else hipLaunchKernelGGL(store_kernel, dim3(g), dim3(256), 0, s, d);
        }
        hipStreamSynchronize(s);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
        printf("%s kernel, %4d workgroups: %.2f us per back-to-back launch\n", mode == 0 ? "empty" : "store", g, us);
      }
    }
  return 0;
}
