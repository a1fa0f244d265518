// wholegraph_amd — HIP virtual-memory stitching for multi-rank CONTINUOUS device memory (memory_vmm.cpp).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <vector>

#include <wholememory/wholememory.h>

namespace wm {

struct vmm_mapping {
  void* base         = nullptr;  // start of the stitched VA range
  size_t total_alloc = 0;        // page-rounded size of the range
  size_t page        = 0;
  std::vector<size_t> alloc_offsets, alloc_sizes;          // physical page runs per rank
  std::vector<hipMemGenericAllocationHandle_t> handles;    // own + imported
};

// Collective over comm. Throws on failure.
void vmm_continuous_create(wholememory_comm_t comm, size_t total_size, vmm_mapping* m);
void vmm_continuous_destroy(wholememory_comm_t comm, vmm_mapping* m) noexcept;

}  // namespace wm
