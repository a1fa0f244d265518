// Reproducer (pure HIP, no torch, no libwholegraph): a buffer from the HIP virtual-memory API that is re-created every iteration
// (reserve / create / map / set access ... synchronise / unmap / release / free / synchronise) starts to LOSE WRITES after some
// tens of cycles — fill kernel followed by a check kernel, 1-5 M wrong words of 400 M, 34 of 150 cycles in one run — while the
// same loop over a buffer mapped ONCE (mode 0) or over hipMalloc memory (mode 1) never fails. ROCm 7.2 / gfx950.
//   hipcc --offload-arch=gfx950 -O2 vmm_cycle.hip -o vmm_cycle;  ./vmm_cycle <mode 0|1|2> [chunks per buffer]
// libwholegraph's own VMM route (CONTINUOUS WholeMemory: one exportable handle per rank, recommended granularity, barrier +
// synchronise around the teardown) passed 150 create / fill / check / destroy cycles of the same size; it got a second
// synchronise after hipMemAddressFree all the same (csrc/memory_vmm.cpp). A chunk-stitched output-buffer allocator built on
// this API was dropped again because of it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(uint32_t* p, size_t n, uint32_t seed) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = seed + (uint32_t)i; }
__global__ void check(const uint32_t* p, size_t n, uint32_t seed, unsigned long long* bad) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) if (p[i] != seed + (uint32_t)i) atomicAdd(bad, 1ull); }
int main(int argc, char** argv)
{
  const int mode = argc > 1 ? atoi(argv[1]) : 0;   // 0: VMM allocated once; 1: hipMalloc once; 2: VMM re-created every iteration with a sync after the free
  hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  unsigned long long* bad; CK(hipMalloc(&bad, 8));
  const size_t n_chunks = argc > 2 ? atoi(argv[2]) : 3; const size_t chunk = (1536ull << 20) / n_chunks, bytes = n_chunks * chunk, n = bytes / 4;
  void* base = nullptr; std::vector<hipMemGenericAllocationHandle_t> hs(n_chunks);
  auto make = [&]() -> int {
    CK(hipMemAddressReserve(&base, bytes, 0, nullptr, 0));
    for (size_t i = 0; i < n_chunks; i++) { CK(hipMemCreate(&hs[i], chunk, &prop, 0)); CK(hipMemMap((char*)base + i * chunk, chunk, 0, hs[i], 0)); }
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(base, bytes, &acc, 1));
    return 0;
  };
  auto drop = [&]() -> int {
    CK(hipDeviceSynchronize());
    for (size_t i = 0; i < n_chunks; i++) CK(hipMemUnmap((char*)base + i * chunk, chunk));
    for (auto h : hs) CK(hipMemRelease(h));
    CK(hipMemAddressFree(base, bytes));
    CK(hipDeviceSynchronize());
    return 0;
  };
  if (mode == 1) CK(hipMalloc(&base, bytes)); else if (mode == 0) { if (make()) return 1; }
  int failures = 0;
  for (int it = 0; it < 150; it++) {
    if (mode == 2 && make()) return 1;
    CK(hipMemsetAsync(bad, 0, 8, nullptr));
    fill<<<4096, 256>>>((uint32_t*)base, n, 17u * it);
    check<<<4096, 256>>>((const uint32_t*)base, n, 17u * it, bad);
    unsigned long long h = 0; CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    if (h) { failures++; if (failures < 4) printf("iter %d: %llu words wrong\n", it, h); }
    if (mode == 2 && drop()) return 1;
  }
  printf("mode %d failures %d of 150\n", mode, failures);
  return 0;
}
