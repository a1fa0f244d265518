// wholegraph_amd — multi-rank CONTINUOUS device memory: every rank's physical pages stitched into ONE
// virtual address range with the HIP virtual-memory API, so any rank can dereference any row with flat
// pointer arithmetic (loads to peer pages travel over xGMI).
//
// Reference counterpart: cpp/src/wholememory/memory_handle.cpp:633-1054 (cuMemCreate + unix-socket fd
// passing + cuMemMap) with the page split of each_rank_multiple_page_strategy (:1684-1704): the physical
// pages of the padded range are dealt to ranks in equal page runs, independent of the logical row partition.
#include <hip/hip_runtime_api.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "knobs.hpp"
#include "communicator.hpp"
#include "memory_vmm.hpp"
#include "wm_common.hpp"

namespace wm {

#define WM_HIP_TRY(expr)                                                                                  \
  do {                                                                                                    \
    hipError_t e__ = (expr);                                                                              \
    if (e__ != hipSuccess) throw ::wm::hip_error(::wm::format_string("%s -> %s", #expr, hipGetErrorString(e__))); \
  } while (0)

namespace {

// Virtual address space that destroyed CONTINUOUS tables keep (see vmm_continuous_destroy): counted, announced, and
// bounded — a process that would run the 47-bit space low fails with a clear error at table creation instead of with an
// obscure reservation failure (or worse) later. WM_VMM_VA_BUDGET_TIB moves the limit (default 64 TiB = half of 2^47).
std::atomic<unsigned long long> g_retired_va_bytes{0};
unsigned long long va_budget_bytes()
{
  const char* e = WM_KNOB("WM_VMM_VA_BUDGET_TIB");
  const unsigned long long tib = e != nullptr && atoll(e) > 0 ? static_cast<unsigned long long>(atoll(e)) : 64ull;
  return tib << 40;
}

std::string sock_name(const char* token, int rank) { return std::string("wgamd_vmm_") + token + "_" + std::to_string(rank); }

sockaddr_un abstract_addr(const std::string& name, socklen_t* len)
{
  sockaddr_un a{};
  a.sun_family = AF_UNIX;
  a.sun_path[0] = '\0';  // abstract namespace: nothing to unlink, vanishes with the socket
  memcpy(a.sun_path + 1, name.data(), std::min(name.size(), sizeof(a.sun_path) - 2));
  *len = static_cast<socklen_t>(offsetof(sockaddr_un, sun_path) + 1 + std::min(name.size(), sizeof(a.sun_path) - 2));
  return a;
}

void send_fd(int sock, const sockaddr_un& to, socklen_t to_len, int fd, int from_rank)
{
  msghdr msg{};
  char ctrl[CMSG_SPACE(sizeof(int))] = {0};
  iovec io{&from_rank, sizeof(from_rank)};
  msg.msg_name       = const_cast<sockaddr_un*>(&to);
  msg.msg_namelen    = to_len;
  msg.msg_iov        = &io;
  msg.msg_iovlen     = 1;
  msg.msg_control    = ctrl;
  msg.msg_controllen = sizeof(ctrl);
  cmsghdr* c         = CMSG_FIRSTHDR(&msg);
  c->cmsg_level      = SOL_SOCKET;
  c->cmsg_type       = SCM_RIGHTS;
  c->cmsg_len        = CMSG_LEN(sizeof(int));
  memcpy(CMSG_DATA(c), &fd, sizeof(int));
  if (sendmsg(sock, &msg, 0) < 0) throw logic_error("sendmsg(SCM_RIGHTS) failed while sharing a VMM handle");
}

int recv_fd(int sock, int* from_rank)
{
  msghdr msg{};
  char ctrl[CMSG_SPACE(sizeof(int))] = {0};
  iovec io{from_rank, sizeof(*from_rank)};
  msg.msg_iov        = &io;
  msg.msg_iovlen     = 1;
  msg.msg_control    = ctrl;
  msg.msg_controllen = sizeof(ctrl);
  if (recvmsg(sock, &msg, 0) < 0) throw logic_error("recvmsg(SCM_RIGHTS) failed while sharing a VMM handle");
  cmsghdr* c = CMSG_FIRSTHDR(&msg);
  if (c == nullptr || c->cmsg_type != SCM_RIGHTS) throw logic_error("no file descriptor received");
  int fd;
  memcpy(&fd, CMSG_DATA(c), sizeof(int));
  return fd;
}

}  // namespace

void vmm_continuous_create(wholememory_comm_t comm, size_t total_size, vmm_mapping* m)
{
  const int W = comm->world_size, rank = comm->world_rank;
  int dev = 0;
  WM_HIP_TRY(hipGetDevice(&dev));
  hipMemAllocationProp prop{};
  prop.type                 = hipMemAllocationTypePinned;
  prop.requestedHandleTypes = hipMemHandleTypePosixFileDescriptor;
  prop.location.type        = hipMemLocationTypeDevice;
  prop.location.id          = dev;
  size_t page = 0;
  WM_HIP_TRY(hipMemGetAllocationGranularity(&page, &prop, hipMemAllocationGranularityRecommended));
  if (page == 0) page = 2u << 20;
  // ranks may report different granularities only on heterogeneous nodes; agree on the maximum
  std::vector<size_t> pages(W);
  comm->allgather_host(&page, pages.data(), sizeof(size_t));
  for (size_t p : pages) page = std::max(page, p);

  m->page        = page;
  m->total_alloc = round_up<size_t>(total_size, page);
  // the budget is a per-process count (ranks may have different create / destroy histories, or different
  // WM_VMM_VA_BUDGET_TIB): the ranks agree before anybody acts, so that all of them fail together instead of one throwing
  // while the others wait in the fd exchange below (advisor, round 4)
  const int over_mine = g_retired_va_bytes.load() + m->total_alloc > va_budget_bytes() ? 1 : 0;
  std::vector<int> over_all(W);
  comm->allgather_host(&over_mine, over_all.data(), sizeof(int));
  int over_rank = -1;
  for (int i = 0; i < W; i++)
    if (over_all[i] != 0 && over_rank < 0) over_rank = i;
  if (over_rank >= 0 && !over_mine)
    throw logic_error(format_string(
      "CONTINUOUS table not created: rank %d has used up its virtual address budget for destroyed CONTINUOUS tables "
      "(WM_VMM_VA_BUDGET_TIB, memory_vmm.cpp); every rank of the communicator gives up together", over_rank));
  if (over_mine)
    throw logic_error(format_string(
      "CONTINUOUS tables destroyed by this process keep %.1f TiB of virtual address space (ranges are never handed back: "
      "stale GPU translations, memory_vmm.cpp); another %.1f GiB would pass the budget of %llu TiB (WM_VMM_VA_BUDGET_TIB). "
      "Create long-lived tables once, or use CHUNKED / DISTRIBUTED tables for create-destroy cycles",
      g_retired_va_bytes.load() / 1099511627776.0, m->total_alloc / 1073741824.0, va_budget_bytes() >> 40));
  const size_t n_pages = m->total_alloc / page;
  m->alloc_offsets.resize(W);
  m->alloc_sizes.resize(W);
  for (int i = 0; i < W; i++) {  // equal page runs (reference each_rank_multiple_page_strategy)
    size_t p0           = static_cast<size_t>(i) * n_pages / W;
    size_t p1           = static_cast<size_t>(i + 1) * n_pages / W;
    m->alloc_offsets[i] = p0 * page;
    m->alloc_sizes[i]   = (p1 - p0) * page;
  }
  m->handles.assign(W, nullptr);

  // 1. own physical pages + exportable fd
  int my_fd = -1;
  if (m->alloc_sizes[rank] > 0) {
    WM_HIP_TRY(hipMemCreate(&m->handles[rank], m->alloc_sizes[rank], &prop, 0));
    WM_HIP_TRY(hipMemExportToShareableHandle(&my_fd, m->handles[rank], hipMemHandleTypePosixFileDescriptor, 0));
  }
  // 2. pass fds around: one abstract datagram socket per rank, named after a token minted by rank 0
  char token[32] = {0};
  if (rank == 0) {
    std::random_device rd;
    snprintf(token, sizeof(token), "%d_%08x", static_cast<int>(getpid()), rd());
  }
  std::vector<char> tokens(static_cast<size_t>(W) * 32);
  comm->allgather_host(token, tokens.data(), 32);
  memcpy(token, tokens.data(), 32);
  int sock = socket(AF_UNIX, SOCK_DGRAM, 0);
  if (sock < 0) throw logic_error("socket(AF_UNIX) failed");
  socklen_t my_len;
  sockaddr_un my_addr = abstract_addr(sock_name(token, rank), &my_len);
  if (bind(sock, reinterpret_cast<sockaddr*>(&my_addr), my_len) != 0) {
    close(sock);
    throw logic_error("bind of the VMM handle-passing socket failed");
  }
  comm->barrier();  // every socket is bound
  std::vector<int> peer_fds(W, -1);
  try {
    for (int r = 0; r < W; r++) {
      if (r == rank || my_fd < 0) continue;
      socklen_t len;
      sockaddr_un to = abstract_addr(sock_name(token, r), &len);
      send_fd(sock, to, len, my_fd, rank);
    }
    int expected = 0;
    for (int r = 0; r < W; r++)
      if (r != rank && m->alloc_sizes[r] > 0) expected++;
    for (int k = 0; k < expected; k++) {
      int from = -1;
      int fd   = recv_fd(sock, &from);
      if (from < 0 || from >= W) throw logic_error("VMM handle from an unknown rank");
      peer_fds[from] = fd;
    }
  } catch (...) {
    close(sock);
    throw;
  }
  comm->barrier();
  close(sock);

  // 3. one VA range, every rank's pages mapped at its page offset, access granted to this device
  WM_HIP_TRY(hipMemAddressReserve(&m->base, m->total_alloc, page, nullptr, 0));
  for (int r = 0; r < W; r++) {
    if (m->alloc_sizes[r] == 0) continue;
    if (r != rank) {
      // osHandle convention differs between HIP runtimes: 7.0 (bundled with torch 2.10+rocm7.0) takes a
      // POINTER to the fd, 7.2 takes the fd VALUE cast to void* (CUDA style). The pointer form is safe to try
      // first on both (a value-style runtime just sees a bogus fd number and returns an error).
      int fd       = peer_fds[r];
      hipError_t e = hipMemImportFromShareableHandle(&m->handles[r], &fd, hipMemHandleTypePosixFileDescriptor);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipMemImportFromShareableHandle(&m->handles[r], reinterpret_cast<void*>(static_cast<intptr_t>(fd)),
                                            hipMemHandleTypePosixFileDescriptor);
      }
      if (e != hipSuccess) throw hip_error(format_string("hipMemImportFromShareableHandle -> %s", hipGetErrorString(e)));
      close(peer_fds[r]);
    }
    WM_HIP_TRY(hipMemMap(static_cast<char*>(m->base) + m->alloc_offsets[r], m->alloc_sizes[r], 0, m->handles[r], 0));
  }
  if (my_fd >= 0) close(my_fd);
  hipMemAccessDesc access{};
  access.location.type = hipMemLocationTypeDevice;
  access.location.id   = dev;
  access.flags         = hipMemAccessFlagsProtReadWrite;
  WM_HIP_TRY(hipMemSetAccess(m->base, m->total_alloc, &access, 1));
  comm->barrier();
}

void vmm_continuous_destroy(wholememory_comm_t comm, vmm_mapping* m) noexcept
{
  if (m->base == nullptr) return;
  (void)hipDeviceSynchronize();
  try {
    comm->barrier();  // nobody unmaps pages a peer kernel may still be reading
  } catch (...) {
  }
  for (size_t r = 0; r < m->handles.size(); r++) {
    if (m->alloc_sizes[r] == 0) continue;
    (void)hipMemUnmap(static_cast<char*>(m->base) + m->alloc_offsets[r], m->alloc_sizes[r]);
    if (m->handles[r] != nullptr) (void)hipMemRelease(m->handles[r]);
  }
  // The virtual range is NOT handed back (WM_VMM_FREE_VA=1 restores hipMemAddressFree). On this ROCm a range that is
  // freed is usually handed out again by the next reservation, and translations of the OLD mapping survive in the GPU's
  // TLBs: kernels on the new mapping then read and write 4 KiB pages of the released physical memory. Round 2 saw it as
  // lost writes in one process (experiments/vmm_cycle.hip) and papered over it with a second synchronise; round 3's
  // multi-process training tests on CONTINUOUS tables (tests/_dist_worker.py: scenario_gradient_apply) still hit it — the
  // wrong rows always filled whole 4 KiB pages (rows 576-639 of a 64-byte-row table ...), the values were the previous
  // table's. A range that is never reserved again cannot be hit by a stale translation: what leaks is address space only
  // (the physical pages are released above), 2^47 bytes of it are there, a table takes its padded size.
  const bool free_va = [] {
    const char* e = WM_KNOB("WM_VMM_FREE_VA");
    return e != nullptr && e[0] == '1';
  }();
  if (free_va) {
    (void)hipMemAddressFree(m->base, m->total_alloc);
  } else {
    const unsigned long long before = g_retired_va_bytes.fetch_add(m->total_alloc);
    const unsigned long long after  = before + m->total_alloc;
    // one line per TiB crossed (a test suite that cycles small tables stays silent)
    if ((after >> 40) != (before >> 40))
      WM_WARN("destroyed CONTINUOUS tables keep %.1f TiB of virtual address space in this process (budget %llu TiB, "
              "WM_VMM_VA_BUDGET_TIB); physical memory IS released", after / 1099511627776.0, va_budget_bytes() >> 40);
  }
  (void)hipDeviceSynchronize();
  m->base = nullptr;
}

}  // namespace wm

