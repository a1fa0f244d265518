// wholegraph_amd — WholeMemory handles: per-rank allocation, the row partition plan and the
// cross-rank mapping that backs wholememory_gref_t.
//
// Reference: cpp/src/wholememory/memory_handle.cpp (:53-231 base + partition strategy, :312-407
// distributed, :412-628 host shm, :633-1054 continuous, :1060-1219 chunked, :1603-1636 plan).
// MI355X layout decisions:
//  * DEVICE shards are single hipMalloc blocks in HBM (288 GB per GPU: a 1 B x 128 fp32 table is
//    64 GB per rank on 8 GPUs, one allocation, no sub-chunking);
//  * CHUNKED over ranks = hipIpc-mapped peer bases; kernels then issue plain global loads that
//    travel over xGMI (fine-grained peer access), one base pointer per rank in a device table;
//  * DISTRIBUTED = local shard only; rows move by RCCL all-to-all-v (ops.cpp);
//  * HOST = one POSIX shared-memory segment mapped by every rank and registered with HIP, so the
//    same gref shapes work for host-resident tables;
//  * with ONE rank every type degenerates to a flat allocation and the ops layer hands kernels a
//    continuous gref (no per-row owner lookup).
//  * multi-rank CONTINUOUS/DEVICE = HIP VMM: each rank creates its page run (hipMemCreate), exports it as
//    a dmabuf fd, fds are passed over AF_UNIX sockets, and every rank maps all runs into one reserved VA
//    range (memory_vmm.cpp).
#include <fcntl.h>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <mutex>
#include <string>
#include <cstring>
#include <random>
#include <vector>

#include <wholememory/wholememory.h>

#include "knobs.hpp"
#include "backend.hpp"
#include "communicator.hpp"
#include "memory_vmm.hpp"
#include "wm_common.hpp"

struct wholememory_handle_ {
  wholememory_comm_t comm                = nullptr;
  wholememory_memory_type_t type         = WHOLEMEMORY_MT_NONE;
  wholememory_memory_location_t location = WHOLEMEMORY_ML_NONE;
  size_t total_size                      = 0;
  size_t granularity                     = 0;
  // row partition, in BYTES
  std::vector<size_t> part_sizes;    // [W]
  std::vector<size_t> part_offsets;  // [W+1]
  size_t mem_stride = 0;             // bytes per rank when same_chunk
  bool same_chunk   = true;
  // local shard
  void* local_ptr       = nullptr;
  size_t local_alloc    = 0;
  bool local_is_pinned  = false;
  bool placement_probed = false;   // the local shard was chosen among several candidates (WM_MALLOC_PROBE / ext_set_malloc_probe)
  // mapped views
  void* global_base = nullptr;        // CONTINUOUS (and HOST chunked): flat pointer usable on device
  std::vector<void*> rank_ptrs;       // CHUNKED: per-rank bases as seen from this process
  void** dev_rank_ptrs     = nullptr; // device copy of rank_ptrs
  size_t* dev_rank_offsets = nullptr; // device copy of part_offsets
  // host shm
  void* shm_host_ptr = nullptr;
  size_t shm_bytes   = 0;
  // multi-rank CONTINUOUS device memory (HIP VMM)
  wm::vmm_mapping vmm;
  // HIERARCHY: ranks of this node / ranks with this local rank on every node (owned by the handle;
  // reference hierarchy_wholememory_impl, memory_handle.cpp:1756-1790,1899-1912)
  wholememory_comm_t local_comm = nullptr;
  wholememory_comm_t cross_comm = nullptr;
};

namespace wm {
namespace {

#define WM_BK(call)                                                                                \
  do {                                                                                             \
    int rc__ = (call);                                                                             \
    if (rc__ != 0) throw ::wm::hip_error(::wm::format_string("%s failed with code %d", #call, rc__)); \
  } while (0)

void plan_partition(wholememory_handle_* h, const size_t* rank_entry_partition)
{
  const int W = h->comm->world_size;
  h->part_sizes.assign(W, 0);
  h->part_offsets.assign(W + 1, 0);
  if (rank_entry_partition != nullptr) {
    // reference memory_handle.cpp:69-79 + :1605-1616
    for (int i = 0; i < W; i++) {
      h->part_sizes[i]       = rank_entry_partition[i] * h->granularity;
      h->part_offsets[i + 1] = h->part_offsets[i] + h->part_sizes[i];
    }
    // same_chunk lets kernels find the owner as byte_offset / stride. The reference sets
    // stride = total / W and same_chunk = "sizes[0..W-2] all equal" (memory_handle.cpp:1605-1616), which
    // mis-addresses peers whenever that common size is not total / W (always true for W == 2 with an
    // uneven split). Here the flag is only raised when the division is actually right: every rank but the
    // last holds exactly sizes[0] bytes and the last holds no more than that.
    h->mem_stride = h->part_sizes[0];
    h->same_chunk = h->part_sizes[W - 1] <= h->part_sizes[0];
    for (int i = 0; i + 1 < W && h->same_chunk; i++) {
      if (h->part_sizes[i] != h->part_sizes[0]) h->same_chunk = false;
    }
    return;
  }
  // equal plan: ceil(slots / W) per rank, clipped (reference memory_handle.cpp:1618-1635)
  const size_t slots    = h->total_size / h->granularity;
  const size_t per_rank = (slots + W - 1) / W;
  for (int i = 0; i < W; i++) {
    size_t s           = std::min<size_t>(static_cast<size_t>(i) * per_rank, slots);
    size_t e           = std::min<size_t>(static_cast<size_t>(i + 1) * per_rank, slots);
    h->part_sizes[i]   = (e - s) * h->granularity;
    h->part_offsets[i] = s * h->granularity;
  }
  h->part_offsets[W] = slots * h->granularity;
  h->mem_stride      = per_rank * h->granularity;
  h->same_chunk      = true;
}

void upload_tables(wholememory_handle_* h)
{
  const auto* bk = backend();
  const int W    = h->comm->world_size;
  WM_BK(bk->malloc_device(reinterpret_cast<void**>(&h->dev_rank_ptrs), sizeof(void*) * W));
  WM_BK(bk->malloc_device(reinterpret_cast<void**>(&h->dev_rank_offsets), sizeof(size_t) * (W + 1)));
  WM_BK(bk->memcpy_async(h->dev_rank_ptrs, h->rank_ptrs.data(), sizeof(void*) * W, nullptr));
  WM_BK(bk->memcpy_async(h->dev_rank_offsets, h->part_offsets.data(), sizeof(size_t) * (W + 1), nullptr));
  WM_BK(bk->stream_sync(nullptr));
  // host copies for the row kernels (by-value owner tables, kernels/rows.hip)
  wm::register_gref_tables(h->dev_rank_ptrs, W, h->rank_ptrs.data(), h->part_offsets.data());
}

std::mutex g_probe_mode_mutex;
std::string g_probe_mode;   // "" = follow WM_MALLOC_PROBE

void alloc_local(wholememory_handle_* h)
{
  const auto* bk  = backend();
  const int rank  = h->comm->world_rank;
  h->local_alloc  = std::max<size_t>(h->part_sizes[rank], 16);
  h->local_is_pinned = h->location == WHOLEMEMORY_ML_HOST;
  if (h->local_is_pinned) {
    WM_BK(bk->malloc_pinned(&h->local_ptr, h->local_alloc));
    return;
  }
  // Placement probe — OPT-IN since round 4 (the reference makes one allocation; so does this library
  // unless asked). The level the memory system serves random row WRITES at (scatter, gradient apply) depends on where a big
  // allocation sits in HBM — by up to 20 %, for the allocation's lifetime — so a device shard of at least
  // WM_MALLOC_PROBE_MIN_BYTES (default 1 GiB) can be chosen among several candidate allocations: each is timed with the probe
  // (kernels/probe.hip: pseudo-random 512-byte row writes, a few ms), the fastest is kept, the others are released. The
  // candidates have to be alive together (an allocation that is freed comes back at the same place).
  //   WM_MALLOC_PROBE unset / 0 / 1   off: the first allocation is the shard
  //   WM_MALLOC_PROBE=auto            self-calibrating: candidates are added until TWO of them agree with the best seen within
  //                                   WM_MALLOC_PROBE_REL (default 3 %) — the level this device serves well placed memory at is
  //                                   learnt from the candidates themselves, no absolute threshold — or 4 have been tried
  //   WM_MALLOC_PROBE=K (2 ... 8)     exactly K candidates
  // Bounded: the losers alive at any time never exceed a quarter of the memory that was free after the first allocation, and
  // the whole section runs under a per-device file lock (ranks or processes sharing a GPU probe one after the other instead of
  // pushing each other, or torch's allocator, into a transient OOM). Why it is not the default: the placement that serves
  // random WRITES best is not the one that serves the gather's random READS best (profiles/r04_six_fresh_processes_probe_off_
  // vs_default.txt: gather 77 % of peak on the first allocation in 5 of 6 processes, 74 % on the probe's choice), and a library
  // call that transiently holds several shards is a behaviour change against the reference. CONTINUOUS tables of more than
  // one rank (HIP VMM handles, memory_vmm.cpp) are never probed: a candidate would have to be mapped, probed and unmapped,
  // and unmapped ranges are exactly what memory_vmm.cpp avoids re-using.
  // A candidate that cannot be allocated ends the search. Every rank decides for its own shard; no collective is involved.
  // wholememory_ext_set_malloc_probe() (Python: create_embedding(..., placement_probe="auto")) takes precedence over the
  // environment for the allocations made while it is set
  std::string api_setting;
  {
    std::lock_guard<std::mutex> lk(g_probe_mode_mutex);
    api_setting = g_probe_mode;
  }
  const char* setting = api_setting.empty() ? WM_KNOB("WM_MALLOC_PROBE") : api_setting.c_str();
  const bool auto_mode = setting != nullptr && (setting[0] == 'a' || setting[0] == 'A');
  const int k_fixed    = (setting == nullptr || auto_mode) ? 1 : std::min(std::max(atoi(setting), 1), 8);
  const size_t min_bytes = [] {
    const char* e = WM_KNOB("WM_MALLOC_PROBE_MIN_BYTES");
    return e != nullptr && atoll(e) > 0 ? static_cast<size_t>(atoll(e)) : (static_cast<size_t>(1) << 30);
  }();
  const float rel = [] {
    const char* e = WM_KNOB("WM_MALLOC_PROBE_REL");
    return e != nullptr && atof(e) > 0 ? static_cast<float>(atof(e)) : 0.03f;
  }();
  const bool probing = (auto_mode || k_fixed > 1) && bk->probe_memory != nullptr && h->local_alloc >= min_bytes;
  if (!probing) {
    WM_BK(bk->malloc_device(&h->local_ptr, h->local_alloc));
    return;
  }
  const bool verbose = WM_KNOB("WM_MALLOC_PROBE_VERBOSE") != nullptr;
  // one prober per device at a time, across processes
  int dev_id = 0;
  if (bk->get_device != nullptr) (void)bk->get_device(&dev_id);
  // the lock file: in the user's runtime directory when there is one, else /tmp with the uid in the name; never through a
  // symlink, never inherited by children; waited for at most WM_MALLOC_PROBE_LOCK_WAIT_S seconds (default 120 — a candidate
  // probe of a 64 GB shard takes ~25 ms, so a longer wait means a stopped or hung holder) and then probed without it, loudly
  char lock_path[256];
  const char* run_dir = WM_KNOB("XDG_RUNTIME_DIR");
  if (run_dir != nullptr && run_dir[0] == '/' && strlen(run_dir) < 200)
    snprintf(lock_path, sizeof(lock_path), "%s/wholegraph_amd_probe_dev%d.lock", run_dir, dev_id);
  else
    snprintf(lock_path, sizeof(lock_path), "/tmp/wholegraph_amd_probe_uid%u_dev%d.lock", static_cast<unsigned>(getuid()), dev_id);
  int lock_fd = open(lock_path, O_CREAT | O_RDWR | O_NOFOLLOW | O_CLOEXEC, 0600);
  if (lock_fd < 0) {
    WM_WARN("wholememory_malloc: cannot open the probe lock %s (%s): probing without it", lock_path, strerror(errno));
  } else {
    const char* we    = WM_KNOB("WM_MALLOC_PROBE_LOCK_WAIT_S");
    const int wait_s  = we != nullptr && atoi(we) >= 0 ? atoi(we) : 120;
    bool locked       = false;
    int lock_errno    = 0;   // a real flock failure (ENOLCK, EBADF ...), as opposed to "somebody else holds it"
    for (int tenth = 0; tenth <= wait_s * 10; tenth++) {
      if (flock(lock_fd, LOCK_EX | LOCK_NB) == 0) {
        locked = true;
        break;
      }
      if (errno != EWOULDBLOCK && errno != EINTR) {
        lock_errno = errno;
        break;
      }
      usleep(100000);
    }
    if (!locked) {
      if (lock_errno != 0)
        WM_WARN("wholememory_malloc: flock(%s) failed: %s — probing without the lock", lock_path, strerror(lock_errno));
      else
        WM_WARN("wholememory_malloc: the probe lock %s stayed busy for %d s (a stopped process?): probing without it", lock_path,
                wait_s);
      close(lock_fd);
      lock_fd = -1;
    }
  }
  struct unlock {
    int fd;
    ~unlock()
    {
      if (fd >= 0) {
        (void)flock(fd, LOCK_UN);
        close(fd);
      }
    }
  } guard{lock_fd};
  WM_BK(bk->malloc_device(&h->local_ptr, h->local_alloc));
  size_t free_b = 0, total_b = 0;
  if (bk->mem_info == nullptr || bk->mem_info(&free_b, &total_b) != 0) return;
  const int max_losers = static_cast<int>(std::min<size_t>(7, free_b / 4 / h->local_alloc));   // alive at once
  if (max_losers < 1) {
    WM_INFO("wholememory_malloc: placement probe skipped, a second %zu-byte candidate would exceed a quarter of the free memory",
            h->local_alloc);
    return;
  }
  const int k_candidates = auto_mode ? 4 : k_fixed;
  float best_ms = 0;
  if (bk->probe_memory(h->local_ptr, h->local_alloc, 0, 3, &best_ms, nullptr) != 0) return;
  if (verbose) fprintf(stderr, "[wholegraph_amd] malloc probe: candidate 0 at %p: %.4f ms per GiB\n", h->local_ptr, best_ms);
  std::vector<void*> losers;
  std::vector<float> seen{best_ms};
  size_t peak_losers = 0;
  int tried = 1;
  for (int k = 1; k < k_candidates; k++) {
    if (static_cast<int>(losers.size()) >= max_losers) break;
    void* cand = nullptr;
    if (bk->malloc_device(&cand, h->local_alloc) != 0 || cand == nullptr) break;
    float ms = 0;
    const int prc = bk->probe_memory(cand, h->local_alloc, 0, 3, &ms, nullptr);
    tried++;
    if (verbose) fprintf(stderr, "[wholegraph_amd] malloc probe: candidate %d at %p: %.4f ms per GiB\n", k, cand, ms);
    if (prc == 0 && ms < best_ms) {
      losers.push_back(h->local_ptr);
      h->local_ptr = cand;
      best_ms      = ms;
    } else {
      losers.push_back(cand);
    }
    if (prc == 0) seen.push_back(ms);
    peak_losers = std::max(peak_losers, losers.size());
    if (auto_mode) {   // two candidates at the best level: that IS this device's level
      int at_best = 0;
      for (float v : seen) at_best += v <= best_ms * (1.0f + rel) ? 1 : 0;
      if (at_best >= 2) break;
    }
  }
  for (void* l : losers) (void)bk->free_device(l);
  h->placement_probed = tried > 1;
  WM_INFO("wholememory_malloc: kept the best of %d probed device allocations of %zu bytes (%.4f ms per GiB of random rows; up to "
          "%zu bytes of candidates were held while probing)",
          tried, h->local_alloc, best_ms, peak_losers * h->local_alloc);
}

void map_chunked_device(wholememory_handle_* h)
{
  const auto* bk = backend();
  const int W    = h->comm->world_size;
  const int rank = h->comm->world_rank;
  h->rank_ptrs.assign(W, nullptr);
  h->rank_ptrs[rank] = h->local_ptr;
  if (W > 1) {
    std::vector<char> handles(static_cast<size_t>(W) * 64);
    char mine[64];
    WM_BK(bk->ipc_get_handle(mine, h->local_ptr));
    h->comm->allgather_host(mine, handles.data(), 64);
    for (int r = 0; r < W; r++) {
      if (r == rank) continue;
      WM_BK(bk->ipc_open_handle(&h->rank_ptrs[r], handles.data() + static_cast<size_t>(r) * 64));
    }
    h->comm->barrier();
  }
  upload_tables(h);
}

// one shm segment holding the whole table; rank 0 names it, everyone maps + registers it
void map_host_shared(wholememory_handle_* h)
{
  const auto* bk = backend();
  const int W    = h->comm->world_size;
  const int rank = h->comm->world_rank;
  char name[64]  = {0};
  if (rank == 0) {
    std::random_device rd;
    snprintf(name, sizeof(name), "/wgamd_%d_%08x%08x", static_cast<int>(getpid()), rd(), rd());
  }
  std::vector<char> names(static_cast<size_t>(W) * 64);
  h->comm->allgather_host(name, names.data(), 64);
  memcpy(name, names.data(), 64);
  h->shm_bytes = std::max<size_t>(h->total_size, 16);
  // Every step a single rank can fail on is followed by an exchange of success flags instead of a throw before the next
  // collective (a rank that threw used to leave the others in a barrier for good), and the name is unlinked on every
  // path once rank 0 has created it.
  auto all_ok = [&](bool mine) {
    char flag = mine ? 1 : 0;
    std::vector<char> flags(static_cast<size_t>(W), 0);
    h->comm->allgather_host(&flag, flags.data(), 1);   // doubles as the barrier between the steps
    for (char f : flags)
      if (f == 0) return false;
    return true;
  };
  int fd       = -1;
  bool created = false;
  if (rank == 0) {
    fd      = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    created = fd >= 0;
    if (created && ftruncate(fd, static_cast<off_t>(h->shm_bytes)) != 0) {
      close(fd);
      fd = -1;
    }
  }
  if (!all_ok(rank != 0 || fd >= 0)) {
    if (created) shm_unlink(name);
    if (fd >= 0) close(fd);
    throw logic_error("shm_open / ftruncate of the shared host segment failed on rank 0");
  }
  if (rank != 0) fd = shm_open(name, O_RDWR, 0600);
  void* p = fd >= 0 ? mmap(nullptr, h->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : MAP_FAILED;
  if (fd >= 0) close(fd);
  const bool mapped = all_ok(p != MAP_FAILED);   // everybody has attached (or given up): the name can go
  if (rank == 0) shm_unlink(name);
  if (!mapped) {
    if (p != MAP_FAILED) munmap(p, h->shm_bytes);
    throw logic_error("attaching the shared host segment failed on at least one rank");
  }
  h->shm_host_ptr = p;
  void* dev       = nullptr;
  WM_BK(bk->host_register(p, h->shm_bytes, &dev));
  h->global_base = dev;
  h->local_ptr   = static_cast<char*>(dev) + h->part_offsets[rank];
  h->rank_ptrs.assign(W, nullptr);
  for (int r = 0; r < W; r++) h->rank_ptrs[r] = static_cast<char*>(dev) + h->part_offsets[r];
  upload_tables(h);
}

void create_memory(wholememory_handle_* h)
{
  const int W = h->comm->world_size;
  if (h->type == WHOLEMEMORY_MT_DISTRIBUTED) {
    alloc_local(h);
    return;
  }
  if (h->type == WHOLEMEMORY_MT_HIERARCHY) {
    // stored like DISTRIBUTED (every rank holds its own range, nothing mapped); what differs is the route of a gather:
    // within the node first, then between nodes along the "rail" of equal local ranks (ops.cpp gather_hierarchy)
    const int L = h->comm->local_size;
    if (wholememory_split_communicator(&h->local_comm, h->comm, h->comm->world_rank / L, h->comm->world_rank % L) !=
          WHOLEMEMORY_SUCCESS ||
        wholememory_split_communicator(&h->cross_comm, h->comm, h->comm->world_rank % L, h->comm->world_rank / L) !=
          WHOLEMEMORY_SUCCESS)
      throw logic_error("cannot split the communicator for a HIERARCHY allocation");
    alloc_local(h);
    return;
  }
  if (W == 1) {  // flat allocation serves every mapped type
    alloc_local(h);
    h->global_base = h->local_ptr;
    h->rank_ptrs.assign(1, h->local_ptr);
    upload_tables(h);
    return;
  }
  if (h->location == WHOLEMEMORY_ML_HOST) {
    map_host_shared(h);
    return;
  }
  if (h->type == WHOLEMEMORY_MT_CHUNKED) {
    alloc_local(h);
    map_chunked_device(h);
    return;
  }
  // multi-rank CONTINUOUS in HBM: every rank's pages stitched into one VA range (memory_vmm.cpp)
  if (backend() != hip_backend()) throw logic_error("multi-rank CONTINUOUS device memory needs the HIP backend");
  vmm_continuous_create(h->comm, h->total_size, &h->vmm);
  h->global_base = h->vmm.base;
  h->local_ptr   = static_cast<char*>(h->vmm.base) + h->part_offsets[h->comm->world_rank];
  h->rank_ptrs.assign(W, nullptr);
  for (int r = 0; r < W; r++) h->rank_ptrs[r] = static_cast<char*>(h->vmm.base) + h->part_offsets[r];
  upload_tables(h);
}

void destroy_memory(wholememory_handle_* h) noexcept
{
  const auto* bk = backend();
  const int W    = h->comm->world_size;
  const int rank = h->comm->world_rank;
  if (h->local_comm != nullptr) wholememory_destroy_communicator(h->local_comm);
  if (h->cross_comm != nullptr) wholememory_destroy_communicator(h->cross_comm);
  h->local_comm = h->cross_comm = nullptr;
  if (h->dev_rank_ptrs) {
    wm::unregister_gref_tables(h->dev_rank_ptrs);
    bk->free_device(h->dev_rank_ptrs);
  }
  if (h->dev_rank_offsets) bk->free_device(h->dev_rank_offsets);
  if (h->vmm.base != nullptr) {
    vmm_continuous_destroy(h->comm, &h->vmm);
  } else if (h->shm_host_ptr != nullptr) {
    bk->host_unregister(h->shm_host_ptr);
    munmap(h->shm_host_ptr, h->shm_bytes);
  } else {
    if (h->type == WHOLEMEMORY_MT_CHUNKED && W > 1) {
      for (int r = 0; r < W; r++)
        if (r != rank && h->rank_ptrs.size() == static_cast<size_t>(W) && h->rank_ptrs[r]) bk->ipc_close_handle(h->rank_ptrs[r]);
      try {
        h->comm->barrier();  // nobody frees a shard a peer still maps
      } catch (...) {
      }
    }
    if (h->local_ptr) {
      if (h->local_is_pinned)
        bk->free_pinned(h->local_ptr);
      else
        bk->free_device(h->local_ptr);
    }
  }
}

}  // namespace
}  // namespace wm

extern "C" {

wholememory_error_code_t wholememory_malloc(wholememory_handle_t* wholememory_handle_ptr,
                                            size_t total_size,
                                            wholememory_comm_t comm,
                                            wholememory_memory_type_t memory_type,
                                            wholememory_memory_location_t memory_location,
                                            size_t data_granularity,
                                            size_t* rank_entry_partition)
{
  if (wholememory_handle_ptr == nullptr || comm == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  // argument checks of reference memory_handle.cpp:1793-1830
  if (total_size == 0) {
    WM_ERROR("wholememory_malloc: total_size must be > 0");
    return WHOLEMEMORY_INVALID_VALUE;
  }
  if (data_granularity == 0 || total_size % data_granularity != 0) {
    WM_ERROR("wholememory_malloc: total_size=%zu is not a multiple of data_granularity=%zu", total_size, data_granularity);
    return WHOLEMEMORY_INVALID_VALUE;
  }
  if (memory_type != WHOLEMEMORY_MT_CONTINUOUS && memory_type != WHOLEMEMORY_MT_CHUNKED &&
      memory_type != WHOLEMEMORY_MT_DISTRIBUTED && memory_type != WHOLEMEMORY_MT_HIERARCHY) {
    WM_ERROR("wholememory_malloc: unknown memory type %d", static_cast<int>(memory_type));
    return WHOLEMEMORY_INVALID_INPUT;
  }
  if (memory_location != WHOLEMEMORY_ML_DEVICE && memory_location != WHOLEMEMORY_ML_HOST) return WHOLEMEMORY_INVALID_INPUT;
  if (wholememory_communicator_support_type_location(comm, memory_type, memory_location) != WHOLEMEMORY_SUCCESS) {
    WM_ERROR("wholememory_malloc: memory type %d is not available on this communicator (%d ranks, %d on this node%s)",
             static_cast<int>(memory_type), comm->world_size, comm->local_size,
             comm->regular_nodes ? "" : ", nodes of unequal size or interleaved ranks");
    return WHOLEMEMORY_NOT_SUPPORTED;
  }
  if (rank_entry_partition != nullptr) {
    size_t sum = 0;
    for (int i = 0; i < comm->world_size; i++) {
      if (rank_entry_partition[i] == 0) return WHOLEMEMORY_INVALID_VALUE;  // reference memory_handle.cpp:1808
      sum += rank_entry_partition[i];
    }
    if (sum * data_granularity != total_size) {
      WM_ERROR("wholememory_malloc: rank_entry_partition sums to %zu entries, expected %zu", sum, total_size / data_granularity);
      return WHOLEMEMORY_INVALID_VALUE;
    }
  }
  WM_API_BEGIN
  std::lock_guard<std::mutex> guard(comm->mu);
  auto* h        = new wholememory_handle_();
  h->comm        = comm;
  h->type        = memory_type;
  h->location    = memory_location;
  h->total_size  = total_size;
  h->granularity = data_granularity;
  wm::plan_partition(h, rank_entry_partition);
  try {
    wm::create_memory(h);
  } catch (...) {
    wm::destroy_memory(h);
    delete h;
    throw;
  }
  comm->live_handles++;
  *wholememory_handle_ptr = h;
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

wholememory_error_code_t wholememory_free(wholememory_handle_t h)
{
  WM_API_BEGIN
  if (h == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  std::lock_guard<std::mutex> guard(h->comm->mu);
  wm::destroy_memory(h);
  h->comm->live_handles--;
  delete h;
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

wholememory_error_code_t wholememory_get_communicator(wholememory_comm_t* comm, wholememory_handle_t h)
{
  if (comm == nullptr || h == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  *comm = h->comm;
  return WHOLEMEMORY_SUCCESS;
}
// HIERARCHY handles only (reference memory_handle.cpp:1988-2017: other memory types answer NOT_SUPPORTED)
wholememory_error_code_t wholememory_get_local_communicator(wholememory_comm_t* comm, wholememory_handle_t h)
{
  if (comm == nullptr || h == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (h->type != WHOLEMEMORY_MT_HIERARCHY) return WHOLEMEMORY_NOT_SUPPORTED;
  *comm = h->local_comm;
  return WHOLEMEMORY_SUCCESS;
}
wholememory_error_code_t wholememory_get_cross_communicator(wholememory_comm_t* comm, wholememory_handle_t h)
{
  if (comm == nullptr || h == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (h->type != WHOLEMEMORY_MT_HIERARCHY) return WHOLEMEMORY_NOT_SUPPORTED;
  *comm = h->cross_comm;
  return WHOLEMEMORY_SUCCESS;
}
wholememory_memory_type_t wholememory_get_memory_type(wholememory_handle_t h) { return h ? h->type : WHOLEMEMORY_MT_NONE; }
wholememory_memory_location_t wholememory_get_memory_location(wholememory_handle_t h)
{
  return h ? h->location : WHOLEMEMORY_ML_NONE;
}
wholememory_distributed_backend_t wholememory_get_distributed_backend(wholememory_handle_t h)
{
  return h ? h->comm->distributed_backend : WHOLEMEMORY_DB_NONE;
}
size_t wholememory_get_total_size(wholememory_handle_t h) { return h ? h->total_size : 0; }
size_t wholememory_get_data_granularity(wholememory_handle_t h) { return h ? h->granularity : 0; }

wholememory_error_code_t wholememory_get_local_memory(void** local_ptr,
                                                      size_t* local_size,
                                                      size_t* local_offset,
                                                      wholememory_handle_t h)
{
  if (h == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  const int rank = h->comm->world_rank;
  if (local_ptr) *local_ptr = h->local_ptr;
  if (local_size) *local_size = h->part_sizes[rank];
  if (local_offset) *local_offset = h->part_offsets[rank];
  return WHOLEMEMORY_SUCCESS;
}
wholememory_error_code_t wholememory_get_local_size(size_t* local_size, wholememory_handle_t h)
{
  return wholememory_get_local_memory(nullptr, local_size, nullptr, h);
}
wholememory_error_code_t wholememory_get_local_offset(size_t* local_offset, wholememory_handle_t h)
{
  return wholememory_get_local_memory(nullptr, nullptr, local_offset, h);
}

wholememory_error_code_t wholememory_get_rank_memory(void** rank_memory_ptr,
                                                     size_t* rank_memory_size,
                                                     size_t* rank_memory_offset,
                                                     int rank,
                                                     wholememory_handle_t h)
{
  if (h == nullptr || rank < 0 || rank >= h->comm->world_size) return WHOLEMEMORY_INVALID_INPUT;
  void* p = nullptr;
  if (rank == h->comm->world_rank)
    p = h->local_ptr;
  else if (h->rank_ptrs.size() == static_cast<size_t>(h->comm->world_size))
    p = h->rank_ptrs[rank];
  if (p == nullptr) return WHOLEMEMORY_INVALID_INPUT;  // DISTRIBUTED: peers are not mapped
  if (rank_memory_ptr) *rank_memory_ptr = p;
  if (rank_memory_size) *rank_memory_size = h->part_sizes[rank];
  if (rank_memory_offset) *rank_memory_offset = h->part_offsets[rank];
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_equal_entry_partition_plan(size_t* entry_per_rank,
                                                                size_t total_entry_count,
                                                                int world_size)
{
  if (entry_per_rank == nullptr || world_size <= 0) return WHOLEMEMORY_INVALID_INPUT;
  *entry_per_rank = (total_entry_count + world_size - 1) / world_size;  // reference memory_handle.cpp:2122-2128
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_get_global_pointer(void** global_ptr, wholememory_handle_t h)
{
  if (global_ptr == nullptr || h == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  // CONTINUOUS, or host CHUNKED (one mapped segment) — reference wholememory.h:399-403
  bool ok = h->type == WHOLEMEMORY_MT_CONTINUOUS || (h->type == WHOLEMEMORY_MT_CHUNKED && h->location == WHOLEMEMORY_ML_HOST);
  if (!ok || h->global_base == nullptr) {
    *global_ptr = nullptr;
    return WHOLEMEMORY_INVALID_INPUT;
  }
  *global_ptr = h->global_base;
  return WHOLEMEMORY_SUCCESS;
}

wholememory_error_code_t wholememory_get_global_reference(wholememory_gref_t* gref, wholememory_handle_t h)
{
  if (gref == nullptr || h == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (h->type == WHOLEMEMORY_MT_CONTINUOUS) {
    *gref = wholememory_create_continuous_global_reference(h->global_base);
    gref->world_size = h->comm->world_size;
    return h->global_base ? WHOLEMEMORY_SUCCESS : WHOLEMEMORY_INVALID_INPUT;
  }
  if (h->type == WHOLEMEMORY_MT_CHUNKED) {  // reference memory_handle.cpp:1174-1185
    gref->pointer             = h->dev_rank_ptrs;
    gref->rank_memory_offsets = h->dev_rank_offsets;
    gref->world_size          = h->comm->world_size;
    gref->stride              = h->mem_stride;
    gref->same_chunk          = h->same_chunk;
    return WHOLEMEMORY_SUCCESS;
  }
  return WHOLEMEMORY_NOT_SUPPORTED;  // DISTRIBUTED has no global reference
}

wholememory_error_code_t wholememory_get_rank_partition_sizes(size_t* rank_mem_sizes, wholememory_handle_t h)
{
  if (rank_mem_sizes == nullptr || h == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  for (int i = 0; i < h->comm->world_size; i++) rank_mem_sizes[i] = h->part_sizes[i];
  return WHOLEMEMORY_SUCCESS;
}
wholememory_error_code_t wholememory_get_rank_partition_offsets(size_t* rank_mem_offsets, wholememory_handle_t h)
{
  if (rank_mem_offsets == nullptr || h == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  for (int i = 0; i <= h->comm->world_size; i++) rank_mem_offsets[i] = h->part_offsets[i];
  return WHOLEMEMORY_SUCCESS;
}

wholememory_gref_t wholememory_create_continuous_global_reference(void* ptr)
{
  wholememory_gref_t g;
  g.pointer             = ptr;
  g.rank_memory_offsets = nullptr;
  g.world_size          = 1;
  g.stride              = 0;
  g.same_chunk          = true;
  return g;
}

}  // extern "C"

extern "C" wholememory_error_code_t wholememory_ext_set_malloc_probe(const char* mode)
{
  std::string m = mode == nullptr ? "" : mode;
  if (m == "env") m.clear();
  if (m == "off") m = "0";
  if (!m.empty() && !(m == "auto" || m == "0" || (m.size() == 1 && m[0] >= '1' && m[0] <= '8'))) return WHOLEMEMORY_INVALID_INPUT;
  std::lock_guard<std::mutex> lk(wm::g_probe_mode_mutex);
  wm::g_probe_mode = m;
  return WHOLEMEMORY_SUCCESS;
}

// the mode wholememory_ext_set_malloc_probe left ("env" when none is set), so that a caller that sets one for a single
// allocation can put back what the application had chosen before (advisor, round 5)
extern "C" wholememory_error_code_t wholememory_ext_get_malloc_probe(char* mode, size_t capacity)
{
  if (mode == nullptr || capacity < 8) return WHOLEMEMORY_INVALID_INPUT;
  std::lock_guard<std::mutex> lk(wm::g_probe_mode_mutex);
  snprintf(mode, capacity, "%s", wm::g_probe_mode.empty() ? "env" : wm::g_probe_mode.c_str());
  return WHOLEMEMORY_SUCCESS;
}

// was the local shard of this handle chosen by the placement probe?
extern "C" int wholememory_ext_handle_was_probed(wholememory_handle_t h) { return h != nullptr && h->placement_probed ? 1 : 0; }

// placement probe on caller memory (experiments, tests; the same probe WM_MALLOC_PROBE uses inside wholememory_malloc)
extern "C" wholememory_error_code_t wholememory_ext_probe_memory(void* ptr, size_t bytes, int kind, int reps, float* ms_per_gib)
{
  const auto* bk = wm::backend();
  if (bk->probe_memory == nullptr) return WHOLEMEMORY_NOT_SUPPORTED;
  const int rc = bk->probe_memory(ptr, bytes, kind, reps, ms_per_gib, nullptr);
  return rc == 0 ? WHOLEMEMORY_SUCCESS : (rc == -1 ? WHOLEMEMORY_INVALID_INPUT : WHOLEMEMORY_CUDA_ERROR);
}
