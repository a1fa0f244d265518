// wholegraph_amd — communicator implementation (see communicator.hpp).
//
// RCCL provider: one ncclComm_t per communicator, created from the 128-byte unique id that the host
// framework broadcasts (reference flow: python torch/comm.py:152-167 -> communicator.cpp:703-752).
// All-to-all-v is a grouped ncclSend/ncclRecv per peer on the CALLER's stream (xGMI is
// point-to-point: 7 links per GPU, one per peer, so one grouped exchange drives every link at once);
// the self segment never touches RCCL — it is a device-to-device copy on the same stream.
// Host-side helpers (barrier, small allgathers) run on a private stream through a staging buffer,
// like the reference's host_* family (nccl_comms.cpp:48-53,99-120), but there is exactly one
// host round trip per call.
#include "knobs.hpp"
#include "communicator.hpp"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <vector>

#include "backend.hpp"
#include "wm_common.hpp"

namespace wm {

#define WM_HIP_TRY(expr)                                                                                  \
  do {                                                                                                    \
    hipError_t e__ = (expr);                                                                              \
    if (e__ != hipSuccess) throw ::wm::hip_error(::wm::format_string("%s -> %s", #expr, hipGetErrorString(e__))); \
  } while (0)

#define WM_NCCL_TRY(expr)                                                                                   \
  do {                                                                                                      \
    ncclResult_t r__ = (expr);                                                                              \
    if (r__ != ncclSuccess) throw ::wm::comm_error(::wm::format_string("%s -> %s", #expr, ncclGetErrorString(r__))); \
  } while (0)

namespace {

// order ranks of one color by (key, old rank); shared by every provider's split()
struct split_entry {
  int color, key, rank;
};

void plan_split(const std::vector<split_entry>& all, int color, int my_rank, int* new_rank, int* new_size,
                std::vector<int>* members)
{
  std::vector<split_entry> mine;
  for (auto& e : all)
    if (e.color == color) mine.push_back(e);
  std::stable_sort(mine.begin(), mine.end(), [](const split_entry& a, const split_entry& b) {
    return a.key != b.key ? a.key < b.key : a.rank < b.rank;
  });
  *new_size = static_cast<int>(mine.size());
  *new_rank = -1;
  if (members) members->clear();
  for (int i = 0; i < *new_size; i++) {
    if (mine[i].rank == my_rank) *new_rank = i;
    if (members) members->push_back(mine[i].rank);
  }
}

class rccl_provider : public collective_provider {
 public:
  rccl_provider(ncclComm_t comm, int rank, int size) : comm_(comm), rank_(rank), size_(size)
  {
    WM_HIP_TRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    WM_HIP_TRY(hipMalloc(&dev_stage_, kStageBytes));
    WM_HIP_TRY(hipHostMalloc(&host_stage_, kStageBytes, hipHostMallocDefault));
    // WM_RCCL_SELF_SENDRECV=1 (bring-up / tests): the self segment of an all-to-all-v also travels as an
    // ncclSend/ncclRecv pair inside the group instead of a device-to-device copy, so the grouped point-to-point
    // path runs on a box with a single GPU
    const char* e  = WM_KNOB("WM_RCCL_SELF_SENDRECV");
    self_sendrecv_ = e != nullptr && e[0] == '1';
  }
  ~rccl_provider() override
  {
    if (comm_ != nullptr) ncclCommDestroy(comm_);
    if (dev_stage_) (void)hipFree(dev_stage_);
    if (host_stage_) (void)hipHostFree(host_stage_);
    if (stream_) (void)hipStreamDestroy(stream_);
  }
  const char* name() const override { return "rccl"; }
  int transport_ranks() const override
  {
    int n = -1;
    return ncclCommCount(comm_, &n) == ncclSuccess ? n : -1;
  }

  void barrier() override
  {  // reference nccl_comms.cpp:82-86: 1-int allreduce + sync
    WM_NCCL_TRY(ncclAllReduce(dev_stage_, dev_stage_, 1, ncclInt32, ncclSum, comm_, stream_));
    WM_HIP_TRY(hipStreamSynchronize(stream_));
  }

  void allgather_host(const void* send, void* recv, size_t bytes) override
  {
    if (bytes * (size_ + 1) > kStageBytes) throw comm_error("allgather_host payload too large for the staging buffer");
    char* dsend = static_cast<char*>(dev_stage_);
    char* drecv = dsend + bytes;
    memcpy(host_stage_, send, bytes);
    WM_HIP_TRY(hipMemcpyAsync(dsend, host_stage_, bytes, hipMemcpyHostToDevice, stream_));
    WM_NCCL_TRY(ncclAllGather(dsend, drecv, bytes, ncclInt8, comm_, stream_));
    WM_HIP_TRY(hipMemcpyAsync(static_cast<char*>(host_stage_) + bytes, drecv, bytes * size_, hipMemcpyDeviceToHost, stream_));
    WM_HIP_TRY(hipStreamSynchronize(stream_));
    memcpy(recv, static_cast<char*>(host_stage_) + bytes, bytes * size_);
  }

  bool allgather_device(const void* send, void* recv, size_t bytes, void* stream_v) override
  {
    WM_NCCL_TRY(ncclAllGather(send, recv, bytes, ncclInt8, comm_, static_cast<hipStream_t>(stream_v)));
    return true;
  }

  void alltoallv_device(const void* send, const size_t* send_bytes, const size_t* send_disp, void* recv,
                        const size_t* recv_bytes, const size_t* recv_disp, void* stream_v) override
  {
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    const char* s      = static_cast<const char*>(send);
    char* r            = static_cast<char*>(recv);
    if (send_bytes[rank_] != recv_bytes[rank_]) throw comm_error("alltoallv: self send/recv size mismatch");
    if (send_bytes[rank_] > 0 && !self_sendrecv_)
      WM_HIP_TRY(hipMemcpyAsync(r + recv_disp[rank_], s + send_disp[rank_], send_bytes[rank_], hipMemcpyDeviceToDevice, stream));
    WM_NCCL_TRY(ncclGroupStart());
    for (int step = self_sendrecv_ ? 0 : 1; step < size_; step++) {  // skewed order: rank r talks to r+step / r-step
      int to   = (rank_ + step) % size_;
      int from = (rank_ - step + size_) % size_;
      if (recv_bytes[from] > 0) WM_NCCL_TRY(ncclRecv(r + recv_disp[from], recv_bytes[from], ncclInt8, from, comm_, stream));
      if (send_bytes[to] > 0) WM_NCCL_TRY(ncclSend(s + send_disp[to], send_bytes[to], ncclInt8, to, comm_, stream));
    }
    WM_NCCL_TRY(ncclGroupEnd());
  }

  std::unique_ptr<collective_provider> split(int color, int key, int my_rank, int* new_rank, int* new_size,
                                             std::vector<int>* members) override
  {
    std::vector<split_entry> all(size_);
    split_entry me{color, key, my_rank};
    allgather_host(&me, all.data(), sizeof(split_entry));
    ncclComm_t sub = nullptr;
    WM_NCCL_TRY(ncclCommSplit(comm_, color < 0 ? NCCL_SPLIT_NOCOLOR : color, key, &sub, nullptr));
    if (color < 0) {
      *new_rank = -1, *new_size = 0;
      return nullptr;
    }
    plan_split(all, color, my_rank, new_rank, new_size, members);
    return std::unique_ptr<collective_provider>(new rccl_provider(sub, *new_rank, *new_size));
  }

 private:
  static constexpr size_t kStageBytes = 1 << 20;
  ncclComm_t comm_                    = nullptr;
  int rank_, size_;
  hipStream_t stream_ = nullptr;
  void* dev_stage_    = nullptr;
  void* host_stage_   = nullptr;
  bool self_sendrecv_ = false;
};

// Sub-group of an external-collectives communicator: every collective is carried by the ROOT provider with the member
// list applied (non-members get zero-byte segments). This only works when all groups of one split run the same sequence
// of collectives at the same time — true for the one user, the HIERARCHY gather, where every node (every rail) runs the
// same steps — and it makes a sub-group barrier a root barrier. The root provider must outlive the sub-group.
class ext_group_provider : public collective_provider {
 public:
  ext_group_provider(collective_provider* root, int root_size, std::vector<int> members)
    : root_(root), root_size_(root_size), members_(std::move(members))
  {
  }
  const char* name() const override { return "external (sub-group)"; }
  void barrier() override { root_->barrier(); }
  void allgather_host(const void* send, void* recv, size_t bytes) override
  {
    std::vector<char> all(static_cast<size_t>(root_size_) * bytes);
    root_->allgather_host(send, all.data(), bytes);
    for (size_t i = 0; i < members_.size(); i++)
      memcpy(static_cast<char*>(recv) + i * bytes, all.data() + static_cast<size_t>(members_[i]) * bytes, bytes);
  }
  void alltoallv_device(const void* send, const size_t* send_bytes, const size_t* send_disp, void* recv,
                        const size_t* recv_bytes, const size_t* recv_disp, void* stream) override
  {
    std::vector<size_t> sb(root_size_, 0), sd(root_size_, 0), rb(root_size_, 0), rd(root_size_, 0);
    for (size_t i = 0; i < members_.size(); i++) {
      const int r = members_[i];
      sb[r] = send_bytes[i], sd[r] = send_disp[i], rb[r] = recv_bytes[i], rd[r] = recv_disp[i];
    }
    root_->alltoallv_device(send, sb.data(), sd.data(), recv, rb.data(), rd.data(), stream);
  }
  std::unique_ptr<collective_provider> split(int color, int key, int my_rank, int* new_rank, int* new_size,
                                             std::vector<int>* members) override
  {
    std::vector<split_entry> all(members_.size());
    split_entry me{color, key, my_rank};
    allgather_host(&me, all.data(), sizeof(split_entry));
    if (color < 0) {
      *new_rank = -1, *new_size = 0;
      return nullptr;
    }
    std::vector<int> sub;
    plan_split(all, color, my_rank, new_rank, new_size, &sub);
    if (members) *members = sub;
    for (auto& r : sub) r = members_[r];  // ranks of this group -> ranks of the root
    return std::unique_ptr<collective_provider>(new ext_group_provider(root_, root_size_, std::move(sub)));
  }

 private:
  collective_provider* root_;
  int root_size_;
  std::vector<int> members_;
};

// Collectives supplied by the host framework (wholegraph_amd_ext.h). split() yields ext_group_provider sub-groups.
class ext_provider : public collective_provider {
 public:
  ext_provider(const wm_ext_collectives_t& c, int, int size) : c_(c), size_(size) {}
  const char* name() const override { return "external"; }
  void barrier() override
  {
    if (c_.barrier(c_.ctx) != 0) throw comm_error("external barrier failed");
  }
  void allgather_host(const void* send, void* recv, size_t bytes) override
  {
    if (c_.allgather_host(c_.ctx, send, recv, bytes) != 0) throw comm_error("external allgather_host failed");
  }
  void alltoallv_device(const void* send, const size_t* send_bytes, const size_t* send_disp, void* recv,
                        const size_t* recv_bytes, const size_t* recv_disp, void* stream) override
  {
    if (c_.alltoallv_device(c_.ctx, send, send_bytes, send_disp, recv, recv_bytes, recv_disp, stream) != 0)
      throw comm_error("external alltoallv_device failed");
  }
  std::unique_ptr<collective_provider> split(int color, int key, int my_rank, int* new_rank, int* new_size,
                                             std::vector<int>* members) override
  {
    std::vector<split_entry> all(size_);
    split_entry me{color, key, my_rank};
    allgather_host(&me, all.data(), sizeof(split_entry));
    if (color < 0) {
      *new_rank = -1, *new_size = 0;
      return nullptr;
    }
    std::vector<int> sub;
    plan_split(all, color, my_rank, new_rank, new_size, &sub);
    if (members) *members = sub;
    return std::unique_ptr<collective_provider>(new ext_group_provider(this, size_, std::move(sub)));
  }

 private:
  wm_ext_collectives_t c_;
  int size_;
};

bool loopback_requested()
{
  const char* e = WM_KNOB("WM_EXCHANGE_SELF");
  return e != nullptr && e[0] == '1';
}

int next_comm_id()
{
  static std::mutex m;
  static int id = 0;
  std::lock_guard<std::mutex> g(m);
  return id++;
}

}  // namespace
}  // namespace wm

// ------------------------------------------------------------------------------------------------
void* wholememory_comm_::get_side_stream()
{
  if (side_stream == nullptr) {
    if (wm::backend()->stream_create(&side_stream) != 0) throw wm::hip_error("cannot create the exchange side stream");
  }
  return side_stream;
}

wholememory_comm_::~wholememory_comm_()
{
  if (side_stream != nullptr) wm::backend()->stream_destroy(side_stream);
}

// Which ranks share a node: every rank publishes (host name, boot id) and nodes are numbered in order of their first
// rank (the reference exchanges the same identity, communicator.cpp:405-500,548-580). WM_LOCAL_SIZE=n overrides the detection
// with "n consecutive ranks per node" (bring-up of the multi-node paths on one box; must divide the world size).
void wholememory_comm_::detect_nodes()
{
  node_of_rank.assign(world_size, 0);
  local_size    = world_size;
  regular_nodes = true;
  if (world_size == 1) return;
  const char* forced = WM_KNOB("WM_LOCAL_SIZE");
  if (forced != nullptr && atoi(forced) > 0) {
    const int n = atoi(forced);
    if (world_size % n != 0) throw wm::logic_error("WM_LOCAL_SIZE must divide the communicator size");
    for (int r = 0; r < world_size; r++) node_of_rank[r] = r / n;
    local_size = n;
    return;
  }
  constexpr size_t kIdBytes = 128;
  char mine[kIdBytes]       = {0};
  gethostname(mine, 63);
  mine[63] = 0;
  if (FILE* f = fopen("/proc/sys/kernel/random/boot_id", "r")) {
    size_t got = fread(mine + 64, 1, 63, f);
    (void)got;
    fclose(f);
  }
  std::vector<char> all(static_cast<size_t>(world_size) * kIdBytes);
  allgather_host(mine, all.data(), kIdBytes);
  int nodes = 0;
  for (int r = 0; r < world_size; r++) {
    int found = -1;
    for (int q = 0; q < r && found < 0; q++)
      if (memcmp(&all[q * kIdBytes], &all[r * kIdBytes], kIdBytes) == 0) found = node_of_rank[q];
    node_of_rank[r] = found >= 0 ? found : nodes++;
  }
  local_size = 0;
  for (int r = 0; r < world_size; r++) local_size += node_of_rank[r] == node_of_rank[world_rank] ? 1 : 0;
  for (int r = 0; r < world_size; r++)
    if (world_size % local_size != 0 || node_of_rank[r] != r / local_size) regular_nodes = false;
}

// node layout of a sub-communicator: `members` lists the parent ranks in their new order
void wholememory_comm_::adopt_nodes(const wholememory_comm_& parent, const std::vector<int>& members)
{
  const int n = static_cast<int>(members.size());
  node_of_rank.assign(n, 0);
  int nodes = 0;
  for (int i = 0; i < n; i++) {
    int found = -1;
    for (int j = 0; j < i && found < 0; j++)
      if (parent.node_of_rank[members[j]] == parent.node_of_rank[members[i]]) found = node_of_rank[j];
    node_of_rank[i] = found >= 0 ? found : nodes++;
  }
  local_size    = 0;
  regular_nodes = true;
  for (int i = 0; i < n; i++) local_size += node_of_rank[i] == node_of_rank[world_rank] ? 1 : 0;
  for (int i = 0; i < n; i++)
    if (n % local_size != 0 || node_of_rank[i] != i / local_size) regular_nodes = false;
}

void wholememory_comm_::barrier()
{
  if (transport) transport->barrier();
}

void wholememory_comm_::allgather_host(const void* send, void* recv, size_t bytes)
{
  if (!transport) {
    memcpy(recv, send, bytes);
    return;
  }
  transport->allgather_host(send, recv, bytes);
}

void wholememory_comm_::alltoall_host_i64(const int64_t* send, int64_t* recv, int64_t* between_ranks, int64_t* extra)
{
  if (between_ranks) *between_ranks = 0;
  if (!transport) {
    recv[0] = send[0];
    return;  // extra[0] already holds this rank's value
  }
  // every rank learns the whole W x (W + 1) matrix in one collective and reads its column
  const size_t W = static_cast<size_t>(world_size), L = W + 1;
  std::vector<int64_t> mine(L), all(W * L);
  for (size_t r = 0; r < W; r++) mine[r] = send[r];
  mine[W] = extra != nullptr ? extra[0] : 0;
  transport->allgather_host(mine.data(), all.data(), sizeof(int64_t) * L);
  for (size_t r = 0; r < W; r++) {
    recv[r] = all[r * L + static_cast<size_t>(world_rank)];
    if (extra != nullptr) extra[r] = all[r * L + W];
  }
  if (between_ranks)
    for (size_t s = 0; s < W; s++)
      for (size_t r = 0; r < W; r++)
        if (s != r) *between_ranks += all[s * L + r];
}

bool wholememory_comm_::alltoall_counts_device(const int64_t* dev_counts, int64_t* dev_matrix, int64_t* pinned_matrix,
                                               void* stream, int64_t* send, int64_t* recv, int64_t* between_ranks,
                                               int64_t* extra)
{
  if (!transport) return false;
  const size_t W = static_cast<size_t>(world_size), L = W + 1;
  if (!transport->allgather_device(dev_counts, dev_matrix, sizeof(int64_t) * L, stream)) return false;
  if (wm::backend()->memcpy_async(pinned_matrix, dev_matrix, sizeof(int64_t) * W * L, stream) != 0 ||
      wm::backend()->stream_sync(stream) != 0)
    throw wm::hip_error("counts exchange: copy of the count matrix failed");
  if (between_ranks) *between_ranks = 0;
  for (size_t r = 0; r < W; r++) {
    send[r] = pinned_matrix[static_cast<size_t>(world_rank) * L + r];
    recv[r] = pinned_matrix[r * L + static_cast<size_t>(world_rank)];
    if (extra != nullptr) extra[r] = pinned_matrix[r * L + W];
    if (between_ranks)
      for (size_t q = 0; q < W; q++)
        if (q != r) *between_ranks += pinned_matrix[r * L + q];
  }
  return true;
}

namespace wm {
extern std::atomic<int64_t> g_alltoallv_bytes;   // ops.cpp, with the other counters
}
void wholememory_comm_::alltoallv_device(const void* send, const size_t* send_bytes, const size_t* send_disp,
                                         void* recv, const size_t* recv_bytes, const size_t* recv_disp, void* stream)
{
  {   // bytes handed to the transport for OTHER ranks (and for this one where it travels like a peer: loopback)
    int64_t out = 0;
    for (int p = 0; p < world_size; p++)
      if (p != world_rank || loopback) out += static_cast<int64_t>(send_bytes[p]);
    wm::g_alltoallv_bytes.fetch_add(out, std::memory_order_relaxed);
  }
  if (!transport) {
    if (send_bytes[0] > 0) {
      int rc = wm::backend()->memcpy_async(static_cast<char*>(recv) + recv_disp[0],
                                           static_cast<const char*>(send) + send_disp[0], send_bytes[0], stream);
      if (rc != 0) throw wm::hip_error("self copy failed in alltoallv");
    }
    return;
  }
  transport->alltoallv_device(send, send_bytes, send_disp, recv, recv_bytes, recv_disp, stream);
}

// ------------------------------------------------------------------------------------------------
extern "C" {

wholememory_error_code_t wholememory_create_unique_id(wholememory_unique_id_t* unique_id)
{
  WM_API_BEGIN
  if (unique_id == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  static_assert(sizeof(ncclUniqueId) <= WHOLEMEMORY_UNIQUE_ID_BYTES, "unique id does not fit");
  ncclUniqueId id;
  WM_NCCL_TRY(ncclGetUniqueId(&id));
  memset(unique_id->internal, 0, WHOLEMEMORY_UNIQUE_ID_BYTES);
  memcpy(unique_id->internal, &id, sizeof(id));
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

wholememory_error_code_t wholememory_create_communicator(wholememory_comm_t* comm,
                                                         wholememory_unique_id_t unique_id,
                                                         int rank,
                                                         int size)
{
  WM_API_BEGIN
  if (comm == nullptr || size < 1 || rank < 0 || rank >= size) return WHOLEMEMORY_INVALID_INPUT;
  auto* c       = new wholememory_comm_();
  c->world_rank = rank;
  c->world_size = size;
  c->local_size = size;
  c->comm_id    = wm::next_comm_id();
  c->loopback   = wm::loopback_requested();
  // WM_FORCE_RCCL=1 builds the RCCL transport even for a single rank (bring-up / smoke testing of the RCCL
  // plumbing on a one-GPU box); normally a single-rank communicator needs no transport at all.
  const char* force = WM_KNOB("WM_FORCE_RCCL");
  if (size > 1 || (force != nullptr && force[0] == '1')) {
    ncclUniqueId id;
    memcpy(&id, unique_id.internal, sizeof(id));
    ncclComm_t nc = nullptr;
    try {
      WM_NCCL_TRY(ncclCommInitRank(&nc, size, id, rank));
      c->transport.reset(new wm::rccl_provider(nc, rank, size));
      c->detect_nodes();
    } catch (...) {
      delete c;
      throw;
    }
  }
  *comm = c;
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

wholememory_error_code_t wholememory_create_communicator_ext(wholememory_comm_t* comm,
                                                             int rank,
                                                             int size,
                                                             const wm_ext_collectives_t* collectives)
{
  WM_API_BEGIN
  if (comm == nullptr || size < 1 || rank < 0 || rank >= size) return WHOLEMEMORY_INVALID_INPUT;
  if (size > 1 && (collectives == nullptr || collectives->barrier == nullptr ||
                   collectives->allgather_host == nullptr || collectives->alltoallv_device == nullptr))
    return WHOLEMEMORY_INVALID_INPUT;
  auto* c       = new wholememory_comm_();
  c->world_rank = rank;
  c->world_size = size;
  c->local_size = size;
  c->comm_id    = wm::next_comm_id();
  c->loopback   = wm::loopback_requested();
  if (size > 1) {
    c->transport.reset(new wm::ext_provider(*collectives, rank, size));
    try {
      c->detect_nodes();
    } catch (...) {
      delete c;
      throw;
    }
  }
  *comm = c;
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

wholememory_error_code_t wholememory_split_communicator(wholememory_comm_t* new_comm,
                                                        wholememory_comm_t comm,
                                                        int color,
                                                        int key)
{
  WM_API_BEGIN
  if (new_comm == nullptr || comm == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  *new_comm = nullptr;
  if (comm->world_size == 1 && comm->transport == nullptr) {
    if (color < 0) return WHOLEMEMORY_SUCCESS;
    auto* c    = new wholememory_comm_();
    c->comm_id = wm::next_comm_id();
    *new_comm  = c;
    return WHOLEMEMORY_SUCCESS;
  }
  int nr = -1, ns = 0;
  std::vector<int> members;
  auto sub = comm->transport->split(color, key, comm->world_rank, &nr, &ns, &members);
  if (color < 0) return WHOLEMEMORY_SUCCESS;
  auto* c       = new wholememory_comm_();
  c->world_rank = nr;
  c->world_size = ns;
  c->comm_id    = wm::next_comm_id();
  c->loopback   = comm->loopback;
  c->adopt_nodes(*comm, members);
  if (ns > 1 || comm->loopback) c->transport = std::move(sub);
  *new_comm = c;
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

wholememory_error_code_t wholememory_destroy_communicator(wholememory_comm_t comm)
{
  WM_API_BEGIN
  if (comm == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (comm->live_handles != 0) WM_WARN("destroying communicator %d with %d live WholeMemory handles", comm->comm_id, comm->live_handles);
  delete comm;
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}

wholememory_error_code_t wholememory_communicator_support_type_location(wholememory_comm_t comm,
                                                                        wholememory_memory_type_t memory_type,
                                                                        wholememory_memory_location_t memory_location)
{
  if (comm == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (memory_location != WHOLEMEMORY_ML_DEVICE && memory_location != WHOLEMEMORY_ML_HOST) return WHOLEMEMORY_NOT_SUPPORTED;
  const bool one_node = comm->local_size == comm->world_size;
  switch (memory_type) {
    case WHOLEMEMORY_MT_CONTINUOUS:
    case WHOLEMEMORY_MT_CHUNKED:  // peer mappings (hipIpc / VMM / shared segment) do not cross nodes (communicator.cpp:357-373)
      return one_node ? WHOLEMEMORY_SUCCESS : WHOLEMEMORY_NOT_SUPPORTED;
    case WHOLEMEMORY_MT_DISTRIBUTED: return WHOLEMEMORY_SUCCESS;
    case WHOLEMEMORY_MT_HIERARCHY:  // needs nodes of equal size holding consecutive ranks (memory_handle.cpp:1780-1783)
      return comm->regular_nodes ? WHOLEMEMORY_SUCCESS : WHOLEMEMORY_NOT_SUPPORTED;
    default: return WHOLEMEMORY_NOT_SUPPORTED;
  }
}

wholememory_error_code_t wholememory_communicator_get_rank(int* rank, wholememory_comm_t comm)
{
  if (rank == nullptr || comm == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  *rank = comm->world_rank;
  return WHOLEMEMORY_SUCCESS;
}
wholememory_error_code_t wholememory_communicator_get_size(int* size, wholememory_comm_t comm)
{
  if (size == nullptr || comm == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  *size = comm->world_size;
  return WHOLEMEMORY_SUCCESS;
}
wholememory_error_code_t wholememory_communicator_get_local_size(int* local_size, wholememory_comm_t comm)
{
  if (local_size == nullptr || comm == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  *local_size = comm->local_size;
  return WHOLEMEMORY_SUCCESS;
}
wholememory_error_code_t wholememory_communicator_get_clique_info(clique_info_t* clique_info, wholememory_comm_t comm)
{
  if (clique_info == nullptr || comm == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  *clique_info                  = clique_info_t{};
  clique_info->is_in_clique     = 0;
  clique_info->clique_first_rank = -1;
  clique_info->clique_rank      = -1;
  clique_info->clique_rank_num  = 0;
  clique_info->clique_id        = -1;
  clique_info->clique_num       = 0;
  return WHOLEMEMORY_SUCCESS;
}
bool wholememory_communicator_is_bind_to_nvshmem(wholememory_comm_t) { return false; }
wholememory_error_code_t wholememory_communicator_set_distributed_backend(wholememory_comm_t comm,
                                                                          wholememory_distributed_backend_t db)
{
  if (comm == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (db != WHOLEMEMORY_DB_NCCL) return WHOLEMEMORY_NOT_SUPPORTED;  // NVSHMEM has no counterpart here
  comm->distributed_backend = db;
  return WHOLEMEMORY_SUCCESS;
}
wholememory_distributed_backend_t wholememory_communicator_get_distributed_backend(wholememory_comm_t comm)
{
  return comm ? comm->distributed_backend : WHOLEMEMORY_DB_NONE;
}
wholememory_error_code_t wholememory_communicator_barrier(wholememory_comm_t comm)
{
  WM_API_BEGIN
  if (comm == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  comm->barrier();
  return WHOLEMEMORY_SUCCESS;
  WM_API_END
}
wholememory_error_code_t wholememory_ext_communicator_transport(wholememory_comm_t comm, const char** name, int* ranks)
{
  if (comm == nullptr) return WHOLEMEMORY_INVALID_INPUT;
  if (name != nullptr) *name = comm->transport ? comm->transport->name() : "none";
  if (ranks != nullptr) *ranks = comm->transport ? comm->transport->transport_ranks() : 0;
  return WHOLEMEMORY_SUCCESS;
}
bool wholememory_is_intranode_communicator(wholememory_comm_t comm) { return comm != nullptr && comm->local_size == comm->world_size; }
bool wholememory_is_intra_mnnvl_communicator(wholememory_comm_t) { return false; }
bool wholememory_is_build_with_nvshmem() { return false; }

}  // extern "C"
