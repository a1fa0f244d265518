// Round 4: which teardown / re-map recipes of the HIP virtual-memory API survive re-creation of a buffer at a REUSED virtual
// range? (vmm_cycle.hip showed: reserve / map / unmap / free every iteration starts to lose writes after tens of cycles.)
//   ./vmm_cycle2 <mode> [iterations] [chunks]
// mode 2: reserve+map ... unmap+release+AddressFree every iteration (the failing baseline)
// mode 3: mode 2 with hipMemSetAccess(PROT_NONE) + synchronise before the unmap
// mode 4: the range is reserved ONCE and kept (a pool): map new handles into it every iteration, unmap + release after
// mode 5: mode 4 with PROT_NONE + synchronise before the unmap
// mode 6: range over-reserved 8 x, the mapping rotates through 8 offsets (a retired range is re-used every 8th cycle)
// mode 7: mode 4, and after every re-map a kernel touches one word per 4 KiB of a separate 16 GiB hipMalloc buffer
//         (pushes the old translations out of the TLBs by capacity)
// mode 8: mode 4, and between the unmap and the next map ONE dummy map / set-access / kernel / unmap cycle of a 2 MiB handle
//         at another reserved address (does any later page-table operation bring the missing invalidation with it?)
// mode 9: mode 4 with a 20 ms sleep after the teardown
// mode 1x (12, 13, 14, 18): mode 6 with x offsets instead of 8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <unistd.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(uint32_t* p, size_t n, uint32_t seed) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = seed + (uint32_t)i; }
__global__ void check(const uint32_t* p, size_t n, uint32_t seed, unsigned long long* bad) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) if (p[i] != seed + (uint32_t)i) atomicAdd(bad, 1ull); }
__global__ void touch(uint32_t* p, size_t pages) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < pages; i += (size_t)gridDim.x * blockDim.x) p[i * 1024] += 1; }
int main(int argc, char** argv)
{
  const int mode = argc > 1 ? atoi(argv[1]) : 2;
  const int iters = argc > 2 ? atoi(argv[2]) : 150;
  hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  unsigned long long* bad; CK(hipMalloc(&bad, 8));
  const size_t n_chunks = argc > 3 ? atoi(argv[3]) : 3; const size_t chunk = (1536ull << 20) / n_chunks, bytes = n_chunks * chunk, n = bytes / 4;
  int mode_in = mode; const bool keep_range = mode >= 4;
  const int rot = mode == 6 ? 8 : mode > 10 ? mode - 10 : 1;
  void* dummy_va = nullptr; const size_t dummy_bytes = 2ull << 20;
  if (mode == 8) CK(hipMemAddressReserve(&dummy_va, dummy_bytes, 0, nullptr, 0));
  void* range = nullptr; char* base = nullptr; std::vector<hipMemGenericAllocationHandle_t> hs(n_chunks);
  uint32_t* big = nullptr; const size_t big_bytes = 16ull << 30;
  if (mode == 7) CK(hipMalloc(&big, big_bytes));
  if (keep_range) CK(hipMemAddressReserve(&range, bytes * rot, 0, nullptr, 0));
  hipMemAccessDesc acc{}; acc.location = prop.location;
  auto make = [&](int it) -> int {
    if (!keep_range) CK(hipMemAddressReserve(&range, bytes, 0, nullptr, 0));
    base = (char*)range + (size_t)(it % rot) * bytes;
    for (size_t i = 0; i < n_chunks; i++) { CK(hipMemCreate(&hs[i], chunk, &prop, 0)); CK(hipMemMap(base + i * chunk, chunk, 0, hs[i], 0)); }
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(base, bytes, &acc, 1));
    if (mode == 7) { touch<<<4096, 256>>>(big, big_bytes / 4096); CK(hipDeviceSynchronize()); }
    return 0;
  };
  auto drop = [&]() -> int {
    CK(hipDeviceSynchronize());
    if (mode == 3 || mode == 5) { acc.flags = hipMemAccessFlagsProtNone; hipError_t e = hipMemSetAccess(base, bytes, &acc, 1); if (e != hipSuccess) { static int once = 0; if (!once++) printf("PROT_NONE -> %s\n", hipGetErrorString(e)); (void)hipGetLastError(); } CK(hipDeviceSynchronize()); }
    for (size_t i = 0; i < n_chunks; i++) CK(hipMemUnmap(base + i * chunk, chunk));
    for (auto h : hs) CK(hipMemRelease(h));
    if (!keep_range) CK(hipMemAddressFree(range, bytes));
    CK(hipDeviceSynchronize());
    if (mode == 9) usleep(20000);
    if (mode == 8) {
      hipMemGenericAllocationHandle_t dh; CK(hipMemCreate(&dh, dummy_bytes, &prop, 0)); CK(hipMemMap(dummy_va, dummy_bytes, 0, dh, 0));
      acc.flags = hipMemAccessFlagsProtReadWrite; CK(hipMemSetAccess(dummy_va, dummy_bytes, &acc, 1));
      fill<<<16, 256>>>((uint32_t*)dummy_va, dummy_bytes / 4, 1u); CK(hipDeviceSynchronize());
      CK(hipMemUnmap(dummy_va, dummy_bytes)); CK(hipMemRelease(dh)); CK(hipDeviceSynchronize());
    }
    return 0;
  };
  int failures = 0, first = -1;
  for (int it = 0; it < iters; it++) {
    if (make(it)) return 1;
    CK(hipMemsetAsync(bad, 0, 8, nullptr));
    fill<<<4096, 256>>>((uint32_t*)base, n, 17u * it);
    check<<<4096, 256>>>((const uint32_t*)base, n, 17u * it, bad);
    unsigned long long h = 0; CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    if (h) { failures++; if (first < 0) first = it; if (failures < 3) printf("  iter %d: %llu words wrong\n", it, h); }
    if (drop()) return 1;
  }
  (void)mode_in; printf("mode %d: %d failures of %d cycles (first at %d)\n", mode, failures, iters, first);
  return 0;
}
