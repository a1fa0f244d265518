// experiment helper: a VA-contiguous device buffer built from separately created physical chunks (HIP VMM), mapped in
// creation order or in a shuffled order — to see whether the gather level of an output buffer follows the physical contiguity
// of its backing (experiments/vmm_scatter.py). hipcc --offload-arch=gfx950 -shared -fPIC vmm_alloc.hip -o libvmm_alloc.so
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <random>
#include <vector>

struct vmm_buf {
  void* base;
  size_t bytes, chunk;
  std::vector<hipMemGenericAllocationHandle_t> handles;
};

static int g_exportable = 0;
extern "C" void vmm_set_exportable(int on) { g_exportable = on; }
// stride k > 1: k x as many chunks are created, every k-th one is mapped, the others are released afterwards — the mapped chunks
// then cannot be physical neighbours of each other
static int g_stride = 1;
extern "C" void vmm_set_stride(int k) { g_stride = k < 1 ? 1 : k; }

extern "C" void* vmm_alloc(size_t bytes, size_t chunk_bytes, int shuffle_seed, void** base_out)
{
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  hipMemAllocationProp prop{};
  prop.type          = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id   = dev;
  if (g_exportable) prop.requestedHandleTypes = hipMemHandleTypePosixFileDescriptor;
  size_t gran        = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess) return nullptr;
  const size_t chunk = ((chunk_bytes + gran - 1) / gran) * gran;
  const size_t n     = (bytes + chunk - 1) / chunk;
  auto* b            = new vmm_buf{nullptr, n * chunk, chunk, {}};
  if (hipMemAddressReserve(&b->base, b->bytes, 0, nullptr, 0) != hipSuccess) return nullptr;
  b->handles.resize(n);
  std::vector<hipMemGenericAllocationHandle_t> spacers;
  for (size_t i = 0; i < n; i++) {
    if (hipMemCreate(&b->handles[i], chunk, &prop, 0) != hipSuccess) return nullptr;
    for (int k = 1; k < g_stride; k++) {
      hipMemGenericAllocationHandle_t sp;
      if (hipMemCreate(&sp, chunk, &prop, 0) != hipSuccess) return nullptr;
      spacers.push_back(sp);
    }
  }
  for (auto sp : spacers) (void)hipMemRelease(sp);
  std::vector<size_t> order(n);
  for (size_t i = 0; i < n; i++) order[i] = i;
  if (shuffle_seed != 0) {
    std::mt19937_64 rng(static_cast<uint64_t>(shuffle_seed));
    std::shuffle(order.begin(), order.end(), rng);
  }
  for (size_t i = 0; i < n; i++)
    if (hipMemMap(static_cast<char*>(b->base) + i * chunk, chunk, 0, b->handles[order[i]], 0) != hipSuccess) return nullptr;
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags    = hipMemAccessFlagsProtReadWrite;
  if (hipMemSetAccess(b->base, b->bytes, &acc, 1) != hipSuccess) return nullptr;
  *base_out = b->base;
  return b;
}

extern "C" void vmm_free(void* handle)
{
  auto* b = static_cast<vmm_buf*>(handle);
  if (b == nullptr) return;
  for (size_t i = 0; i < b->handles.size(); i++) (void)hipMemUnmap(static_cast<char*>(b->base) + i * b->chunk, b->chunk);   // one per mapping
  for (auto h : b->handles) (void)hipMemRelease(h);
  (void)hipMemAddressFree(b->base, b->bytes);
  (void)hipDeviceSynchronize();   // teardown is not finished when hipMemAddressFree returns (experiments/vmm_cycle.hip)
  delete b;
}

extern "C" size_t vmm_granularity()
{
  int dev = 0;
  (void)hipGetDevice(&dev);
  hipMemAllocationProp prop{};
  prop.type          = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id   = dev;
  size_t gran        = 0;
  (void)hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
  return gran;
}
