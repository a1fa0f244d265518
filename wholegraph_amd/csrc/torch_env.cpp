// wholegraph_amd — native wholememory_env_func_t for processes that run PyTorch-ROCm: scratch memory of the ops comes
// straight from torch's HIP caching allocator (on the current torch stream), without a round trip through Python for
// every buffer. Counterpart of the reference's torch C++ extension (python/pylibwholegraph/pylibwholegraph/
// torch_cpp_ext/torch_env_func_ptrs.cpp:24-52, torch_utils.cpp:62-98). Built as a separate small library
// (libwg_torch_env.so) so that libwholegraph.so itself stays free of torch.
//
// Temporary buffers (temporary_fns) are native. Output buffers (output_fns: results whose size the caller cannot know,
// e.g. sampled neighbours) must become torch tensors the Python caller can hold, so those two entries are supplied by
// the caller (wholegraph_amd/torch/wholegraph_env.py passes its Python callbacks).
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <mutex>
#include <vector>

#include <c10/hip/HIPCachingAllocator.h>
#include <c10/hip/HIPStream.h>

#include <wholememory/env_func_ptrs.h>
#include <wholememory/tensor_description.h>

namespace {

struct scratch {
  void* ptr                                  = nullptr;
  wholememory_memory_allocation_type_t type = WHOLEMEMORY_MA_NONE;
  size_t bytes                               = 0;
};

// pinned staging buffers (a few dozen bytes of counts per op) are recycled: hipHostMalloc costs ~100 us
struct pinned_pool {
  std::mutex mu;
  std::vector<std::pair<void*, size_t>> idle;
  void* take(size_t bytes)
  {
    {
      std::lock_guard<std::mutex> g(mu);
      for (size_t i = 0; i < idle.size(); i++)
        if (idle[i].second >= bytes && idle[i].second <= 4 * bytes + 4096) {
          void* p = idle[i].first;
          idle.erase(idle.begin() + static_cast<long>(i));
          return p;
        }
    }
    void* p = nullptr;
    return hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr;
  }
  size_t capacity_of(size_t bytes) { return bytes; }
  void give(void* p, size_t bytes)
  {
    std::lock_guard<std::mutex> g(mu);
    if (idle.size() < 64) {
      idle.emplace_back(p, bytes);
    } else {
      (void)hipHostFree(p);
    }
  }
};
pinned_pool& pool()
{
  static pinned_pool p;
  return p;
}

void release(scratch* s)
{
  if (s->ptr == nullptr) return;
  switch (s->type) {
    case WHOLEMEMORY_MA_DEVICE: c10::hip::HIPCachingAllocator::raw_delete(s->ptr); break;
    case WHOLEMEMORY_MA_PINNED: pool().give(s->ptr, s->bytes); break;
    default: free(s->ptr); break;
  }
  s->ptr = nullptr;
}

void create_ctx(void** ctx, void*) { *ctx = new scratch(); }

void destroy_ctx(void* ctx, void*)
{
  auto* s = static_cast<scratch*>(ctx);
  if (s == nullptr) return;
  release(s);
  delete s;
}

void* scratch_malloc(wholememory_tensor_description_t* desc, wholememory_memory_allocation_type_t type, void* ctx, void*)
{
  auto* s = static_cast<scratch*>(ctx);
  release(s);
  size_t bytes = static_cast<size_t>(wholememory_get_memory_element_count_from_tensor(desc)) *
                 wholememory_dtype_get_element_size(desc->dtype);
  if (bytes == 0) bytes = 16;
  s->type  = type;
  s->bytes = bytes;
  try {
    switch (type) {
      case WHOLEMEMORY_MA_DEVICE:
        // freed memory is reused by later work on the same stream only (the allocator's stream-ordered reuse): exactly
        // the contract of a torch.empty() made while that stream is current
        s->ptr = c10::hip::HIPCachingAllocator::raw_alloc_with_stream(bytes, c10::hip::getCurrentHIPStream().stream());
        break;
      case WHOLEMEMORY_MA_PINNED: s->ptr = pool().take(bytes); break;
      default: s->ptr = malloc(bytes); break;
    }
  } catch (...) {
    s->ptr = nullptr;  // out of memory: the op reports it through its own error path
  }
  return s->ptr;
}

void scratch_free(void* ctx, void*) { release(static_cast<scratch*>(ctx)); }

}  // namespace

extern "C" {

// Fills `env`: native temporary_fns; output_fns as given (may be null when the caller never runs an op with
// variable-size outputs).
void wg_torch_env_init(wholememory_env_func_t* env, wholememory_malloc_func_t output_malloc,
                       wholememory_free_func_t output_free, void* output_global_context)
{
  env->temporary_fns.create_memory_context_fn  = create_ctx;
  env->temporary_fns.destroy_memory_context_fn = destroy_ctx;
  env->temporary_fns.malloc_fn                 = scratch_malloc;
  env->temporary_fns.free_fn                   = scratch_free;
  env->temporary_fns.global_context            = nullptr;
  env->output_fns.malloc_fn                    = output_malloc;
  env->output_fns.free_fn                      = output_free;
  env->output_fns.global_context               = output_global_context;
}

// recycled pinned buffers back to the driver (tests; process exit does it anyway)
void wg_torch_env_trim()
{
  auto& p = pool();
  std::lock_guard<std::mutex> g(p.mu);
  for (auto& e : p.idle) (void)hipHostFree(e.first);
  p.idle.clear();
}

}  // extern "C"
