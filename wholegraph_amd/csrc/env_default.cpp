// wholegraph_amd — built-in env allocators for C/C++ callers, tests and the bench.
// Counterpart of reference cpp/src/wholememory/env_func_ptrs.cpp:29-105 (plain) and :107-373
// (cached). The cached variant keeps freed blocks in power-of-two size classes (>= 256 B; >= 1 GiB
// rounded to 1 GiB) so the steady-state op path performs no hipMalloc/hipFree at all.
#include <map>
#include <mutex>
#include <vector>

#include <wholememory/env_func_ptrs.h>

#include "backend.hpp"
#include "wm_common.hpp"

namespace {

struct mem_slot {
  void* ptr                                 = nullptr;
  size_t bytes                              = 0;
  wholememory_memory_allocation_type_t type = WHOLEMEMORY_MA_NONE;
};

void* raw_alloc(size_t bytes, wholememory_memory_allocation_type_t type)
{
  void* p        = nullptr;
  const auto* bk = wm::backend();
  if (bytes == 0) bytes = 16;
  int rc = 0;
  switch (type) {
    case WHOLEMEMORY_MA_DEVICE: rc = bk->malloc_device(&p, bytes); break;
    case WHOLEMEMORY_MA_PINNED: rc = bk->malloc_pinned(&p, bytes); break;
    case WHOLEMEMORY_MA_HOST: p = malloc(bytes); break;
    default: rc = -1;
  }
  if (rc != 0 || p == nullptr) {
    WM_ERROR("env allocator: allocation of %zu bytes (type %d) failed", bytes, static_cast<int>(type));
    return nullptr;
  }
  return p;
}

void raw_free(void* p, wholememory_memory_allocation_type_t type)
{
  if (p == nullptr) return;
  const auto* bk = wm::backend();
  switch (type) {
    case WHOLEMEMORY_MA_DEVICE: bk->free_device(p); break;
    case WHOLEMEMORY_MA_PINNED: bk->free_pinned(p); break;
    case WHOLEMEMORY_MA_HOST: free(p); break;
    default: break;
  }
}

// ---- plain ----
void plain_create(void** ctx, void*) { *ctx = new mem_slot(); }
void plain_free(void* ctx, void*)
{
  auto* s = static_cast<mem_slot*>(ctx);
  raw_free(s->ptr, s->type);
  *s = mem_slot();
}
void plain_destroy(void* ctx, void* g)
{
  plain_free(ctx, g);
  delete static_cast<mem_slot*>(ctx);
}
void* plain_malloc(wholememory_tensor_description_t* desc, wholememory_memory_allocation_type_t type, void* ctx, void*)
{
  auto* s  = static_cast<mem_slot*>(ctx);
  s->bytes = static_cast<size_t>(wholememory_get_memory_size_from_tensor(desc));
  s->type  = type;
  s->ptr   = raw_alloc(s->bytes, type);
  return s->ptr;
}

// ---- cached ----
struct pool {
  std::mutex mu;
  std::map<std::pair<int, size_t>, std::vector<void*>> free_blocks;  // (type, class bytes) -> blocks
};
pool& the_pool()
{
  static pool p;
  return p;
}
size_t size_class(size_t bytes)
{
  constexpr size_t kGiB = 1ull << 30;
  if (bytes >= kGiB) return (bytes + kGiB - 1) / kGiB * kGiB;
  size_t c = 256;
  while (c < bytes) c <<= 1;
  return c;
}
void* cached_malloc(wholememory_tensor_description_t* desc, wholememory_memory_allocation_type_t type, void* ctx, void*)
{
  auto* s  = static_cast<mem_slot*>(ctx);
  s->bytes = size_class(static_cast<size_t>(wholememory_get_memory_size_from_tensor(desc)));
  s->type  = type;
  {
    auto& p = the_pool();
    std::lock_guard<std::mutex> g(p.mu);
    auto it = p.free_blocks.find({static_cast<int>(type), s->bytes});
    if (it != p.free_blocks.end() && !it->second.empty()) {
      s->ptr = it->second.back();
      it->second.pop_back();
      return s->ptr;
    }
  }
  s->ptr = raw_alloc(s->bytes, type);
  return s->ptr;
}
void cached_free(void* ctx, void*)
{
  auto* s = static_cast<mem_slot*>(ctx);
  if (s->ptr != nullptr) {
    auto& p = the_pool();
    std::lock_guard<std::mutex> g(p.mu);
    p.free_blocks[{static_cast<int>(s->type), s->bytes}].push_back(s->ptr);
  }
  *s = mem_slot();
}
void cached_destroy(void* ctx, void* g)
{
  cached_free(ctx, g);
  delete static_cast<mem_slot*>(ctx);
}

wholememory_env_func_t g_plain  = {{plain_create, plain_destroy, plain_malloc, plain_free, nullptr},
                                   {plain_malloc, plain_free, nullptr}};
wholememory_env_func_t g_cached = {{plain_create, cached_destroy, cached_malloc, cached_free, nullptr},
                                   {cached_malloc, cached_free, nullptr}};

}  // namespace

extern "C" {
wholememory_env_func_t* wholememory_get_default_env_func() { return &g_plain; }
wholememory_env_func_t* wholememory_get_cached_env_func() { return &g_cached; }
void wholememory_drop_cached_env_func_cache()
{
  auto& p = the_pool();
  std::lock_guard<std::mutex> g(p.mu);
  for (auto& kv : p.free_blocks)
    for (void* b : kv.second) raw_free(b, static_cast<wholememory_memory_allocation_type_t>(kv.first.first));
  p.free_blocks.clear();
}
}
